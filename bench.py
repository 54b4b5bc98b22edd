#!/usr/bin/env python3
"""bench.py -- Mbases/s through the NanoSpring hot path (sketch + overlap + align + consensus edits) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path over one batch of synthetic reads that is already
resident (2-bit packed) in HBM:
    MinHash sketch of every read -> n bucket tables                      (MinHashReadFilter::initialize)
    -> contig stage: window queries against the tables, batched alignRead of every candidate
       against its contig's consensus (all banded DP on the GPU), consensus-DAG update,
       consensus-edit emission into the seven streams                    (Consensus::generateAndWriteConsensus)
Workload = BASELINE.json configs[1]: 100 000 synthetic ONT-like reads, mean 8 kb, 20x of an iid
genome, 1% sub + 1% ins + 1% del, k=23 n=60 thr=6, minimap k=20 w=50, per GPU.

Multi-GPU (one process per GPU, RCCL): reads shard by id (rank r generates reads [r*R, (r+1)*R) of ONE read set over
a genome N times larger: weak scaling).  Exchange mode (default) runs the C++ driver of csrc/dist.hip on the library's own
RCCL communicator: the read shards are replicated by all-gather at load time; every step each rank sketches its own id
range, the (slot, key, id) tuples go by RCCL all-to-all to the bucket-table owners (table j on rank j % N), the sorted
tables are all-gathered (window queries stay local), and the contig builders with gid % N == rank run on each rank with
ONE small all-gather of claim / seed request lists per pipeline slot (resolved in global builder order).
--dist-mode replicate: all-gather of the sketch rows instead.  --no-exchange: independent shards, no collective.

Prints ONE JSON line on rank 0, including
  roofline     : dominant kernel (ksw_extd2 wavefront DP) algorithmic bytes / HIP-event kernel time vs HBM peak
  cpu_baseline : the reference's hot path as oracle/consensus_oracle.cpp restates it, with the reference's own minimap2
                 (oracle/_ref) answering the alignments, on ALL host cores (-t N, OpenMP like the reference): on the SAME full input on
                 this host (`value`; 1 GPU at full cfg2 size, --cpu-full 0 to skip), on a bounded sample and at -t 1; core count and CPU model stated.
  reference_legal_schedule : one step of the fastest schedule found that is an interleaving the reference's -t N can itself produce
                 (Consensus::getRead's seed rule, no bucket policy) with streams within 5 % of the reference's -t N.
  nonideal     : one step on a genome with planted repeats (duplications, tandem repeats, homopolymer / (AT)n runs).
  compression  : stream bytes per base of the timed schedule beside the reference's own -t <cores> and -t 1 runs on the SAME input
                 (oracle/consensus_oracle.cpp, committed measurements under profiles/): the default schedule is chosen so that
                 the streams stay within 5 % of the reference's -t N (iso-compression); `throughput_schedule` times one step of the
                 1024-builder pipelined schedule, which is faster and is NOT iso-compression (its ratio is stated).
  config.consensus_graph : where the contigs' consensus graphs lived in the timed steps (--graph: on the host as a pointer graph, or in HBM with
                 one workgroup per accepted read; by default the library decides by the host threads of the process: in HBM with at most 5)
                 and the graph kernels' counters (updates, launches, splitPath calls, sequential fall-backs: n_sequential_updates / n_full_walks).
  consensus_graph_other_placement : ONE first step with the graphs in the other placement (same schedule; the streams must be the same bytes).
  host_threads_sweep : ONE first step with 4 and with 2 host threads (what a rank of a shared node gets), each in a child process.
  cfg3         : ONE first step of BASELINE configs[2]'s shape (1.0 Gbase at 217x of a 4.6 Mb genome), the automatic schedule.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before anything initialises HIP (see nanospring_amd/__init__.py)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def host_cores():
    """CPUs this process may use: the cgroup quota when there is one (cpu.max), else the affinity mask."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(float(q) / float(per) + 0.5))
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def full_input_reference():
    """The reference's -t N on the FULL cfg2 input (oracle, reference minimap2): a committed measurement of the build container (8 cores),
    the only place an hour-scale CPU run fits -- rate, contigs and stream size on the very input bench.py times."""
    pth = os.path.join(ROOT, "profiles", "r03_oracle_t8_cfg2.json")
    if not os.path.exists(pth):
        return None
    oj = json.load(open(pth))
    return {"value": round(oj["mbases_per_s"], 3), "unit": "Mbases/s", "cores": oj["threads"], "seconds": round(oj["seconds"], 1),
            "host": oj.get("host", "build container, %s CPUs" % oj.get("host_cpus", "?")),
            "contigs": oj["stats"]["n_contigs"], "lone_reads": oj["stats"]["n_lone"], "stream_bytes_per_base": round(oj["stream_bytes_per_base"], 4),
            "source": "profiles/r03_oracle_t8_cfg2.json"}


def cpu_full_input(bases, off, k, n, thr, salts):
    """The reference's -t <cores> on the very input of the timed steps, on THIS host: oracle/consensus_oracle.cpp (the reference's OpenMP loop
    and try_lock claiming, the reference's own minimap2 answering every alignRead).  Checker code, timed after the timed region."""
    from tests import oracle_lib
    cores = host_cores()
    t0 = time.perf_counter()
    sm, st = oracle_lib.cons_oracle_run(bases, off, salts, k=k, n=n, thr=thr, checks=False, num_thr=cores)
    dt = time.perf_counter() - t0
    nb = int(off[-1])
    stream = sum(len(t[x]) for t in (sm["threads"] if "threads" in sm else [sm]) for x in oracle_lib.CONS_STREAMS)
    return {"value": round(nb / 1e6 / dt, 3), "unit": "Mbases/s", "cores": cores, "cpu": cpu_model(), "seconds": round(dt, 1),
            "contigs": st["n_contigs"], "lone_reads": st["n_lone"], "reads_aligned": st["count_aligner"], "bad_roundtrip": st["n_bad_roundtrip"],
            "stream_bytes_per_base": round(stream / nb, 4),
            "note": "timing-dependent for -t > 1: one sample of the reference's own distribution on this input"}


def cpu_baseline(n_reads, mean_len, k, n, thr, salts, t1_reads=1500):
    """The reference's hot path on this host's cores, as oracle/consensus_oracle.cpp restates it (kind "port": literal
    Consensus / ConsensusGraph + ns_oracle.c's MinHash filter; every alignRead is answered by the REFERENCE's own minimap2,
    oracle/_ref/libmm2ref.so, SSE ksw2): (a) -t <all host cores>, the reference's OpenMP schedule with its optimistic read
    claiming, on a bounded sample; (b) -t 1 on a smaller sample.  Checker code, timed -- never on the product's path."""
    import nanospring_amd as ns
    from tests import oracle_lib
    cores = host_cores()
    bases, off = ns.synth_reads(11, int(n_reads * mean_len / 20), n_reads, mean_len)
    t0 = time.perf_counter()
    sm, st = oracle_lib.cons_oracle_run(bases, off, salts, k=k, n=n, thr=thr, checks=False, num_thr=cores)
    dt = time.perf_counter() - t0
    nb = int(off[-1])
    sample_stream = sum(len(t[x]) for t in (sm["threads"] if "threads" in sm else [sm]) for x in oracle_lib.CONS_STREAMS)
    assert st["n_bad_roundtrip"] == 0
    b1, o1 = ns.synth_reads(11, int(t1_reads * mean_len / 20), t1_reads, mean_len)
    t0 = time.perf_counter()
    _, s1 = oracle_lib.cons_oracle_run(b1, o1, salts, k=k, n=n, thr=thr, checks=False, num_thr=1)
    d1 = time.perf_counter() - t0
    assert s1["n_bad_roundtrip"] == 0
    return {"value": round(nb / 1e6 / dt, 3), "unit": "Mbases/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"{n_reads} reads / {nb / 1e6:.1f} Mbases, same generator and parameters (20x of a {n_reads * mean_len / 20 / 1e6:.2f} Mb genome), "
                      f"-t {cores} in {dt:.1f} s (sketch + tables {st['sketch_ms'] / 1e3:.1f} s, contig stage {st['consensus_ms'] / 1e3:.1f} s): "
                      f"{st['count_aligner']} reads aligned into {st['n_contigs']} contigs ({st['n_lone']} lone reads)",
            "sample_stream_bytes_per_base": round(sample_stream / nb, 4),
            "full_input": full_input_reference(),
            "t1": {"value": round(int(o1[-1]) / 1e6 / d1, 3), "cores": 1,
                   "sample": f"{t1_reads} reads / {int(o1[-1]) / 1e6:.1f} Mbases in {d1:.1f} s; the -t 1 rate falls with the input size (contigs get longer and the "
                             f"reference re-indexes the whole consensus per candidate): 1.10 Mbases/s on the full cfg2 input (profiles/r01_parity_full.txt)"}}


def threads_sweep(args):
    """ONE first step with 4 and with 2 host threads (what a rank gets of a CPU quota it shares with the other ranks of its node), each in a child
    process (the thread count is read once per process): this very script without its other legs.  Run BEFORE this process touches the GPU: a
    child beside a process that holds the GPU shares the hardware queues with it and measures that, not the thread count."""
    import subprocess
    ts = {"note": "one first step per thread count, each in a child process (NSGPU_THREADS) before this process initialised the GPU, consensus graphs where the library puts them by itself (in HBM with at most 5 host threads); the timed steps above ran with host_threads = %d.  The pointer graph on the host with 2 threads: profiles/r06_graph_placement_by_threads.txt", "runs": []}
    for nthr in (4, 2):
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--throughput-leg", "0", "--cpu-sample", "0", "--cpu-full", "0", "--legal-leg", "0",
               "--nonideal-leg", "0", "--threads-sweep", "0", "--graph-leg", "0", "--cfg3-leg", "0", "--reads", str(args.reads), "--mean-len", str(args.mean_len), "--depth", str(args.depth), "--genome", args.genome]
        try:
            rr = subprocess.run(cmd, env=dict(os.environ, NSGPU_THREADS=str(nthr)), capture_output=True, text=True, timeout=600)
            lines = [ln for ln in rr.stdout.splitlines() if ln.startswith("{")]
            if not lines: raise RuntimeError("the child printed no result (exit code %d): %s" % (rr.returncode, rr.stderr[-600:]))
            cj = json.loads(lines[-1])
            ts["runs"].append({"host_threads": nthr, "value": cj["value"], "unit": "Mbases/s", "ms_per_step": cj["ms_per_step"], "consensus_graphs": cj["config"]["consensus_graph"]["placement"],
                               "graph_host_wall_ms": cj["config"]["stage_ms_per_step"]["graph_host_wall"], "builder_steps_host_cpu_ms": cj["config"]["stage_ms_per_step"].get("builder_steps_host_cpu"),
                               "lossless_roundtrip_bad_reads": cj["config"]["lossless_roundtrip_bad_reads"], "streams_identical_to_the_fixture": (cj.get("parity") or {}).get("all_identical")})
            if cj["config"]["lossless_roundtrip_bad_reads"] or (cj.get("parity") or {}).get("all_identical") is False: ts["runs"][-1]["stderr_tail"] = rr.stderr[-3000:]
        except Exception as ex:                 # (a leg, not the measurement: report and go on)
            ts["runs"].append({"host_threads": nthr, "error": str(ex)[:900]})
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU (cfg2: 100000)")
    ap.add_argument("--mean-len", type=float, default=8000.0)
    ap.add_argument("--builders", type=int, default=0, help="virtual contig builders per GPU (default 0: the library's own choice, nsgpu_set_schedule_auto -- on cfg2: 80)")
    ap.add_argument("--groups", type=int, default=0, choices=[0, 1, 2, 4], help="pipeline groups of the contig stage: a builder steps once per `groups` slots (nsgpu_set_schedule); default 0: the library derives the whole schedule from the input (nsgpu_set_schedule_auto; on cfg2: one group, buckets of depth 3, 5 rings, 3 in the tail)")
    ap.add_argument("--seed-depth", type=int, default=3, help="with --groups: conflict-aware seeds: bucket depth (0 = the reference's getRead rule)")
    ap.add_argument("--seed-rings", type=int, default=5, help="with --groups: conflict-aware seeds: adjacency rings around occupied buckets that a seed must keep clear of")
    ap.add_argument("--threads-sweep", type=int, default=-1, help="also time ONE step with 4 and with 2 host threads (what a rank gets of a shared CPU quota), each in a child process (default: only with 1 GPU at full cfg2 size; 0 = skip)")
    ap.add_argument("--depth", type=float, default=20.0, help="sequencing depth of the synthetic read set (cfg2: 20; cfg3's E. coli regime: ~200)")
    ap.add_argument("--genome", choices=["iid", "repeats"], default="iid", help="synthetic genome: iid (BASELINE cfg2) or with planted duplications / tandem repeats / homopolymer and (AT)n runs")
    ap.add_argument("--throughput-leg", type=int, default=-1, help="also time ONE step of the 1024-builder pipelined schedule, which is not iso-compression (default: only with 1 GPU at full cfg2 size; 0 = skip)")
    ap.add_argument("--seed-tail-rings", type=int, default=3, help="conflict-aware seeds: the radius in a seed round in which more than half of ALL builders ask (one round carries one group's requests, so this only acts with --groups 1; default 3, negative = --seed-rings)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads in the all-cores CPU-baseline sample (0 = skip; default: 2000 per host core)")
    ap.add_argument("--cpu-full", type=int, default=-1, help="also time the reference's -t <cores> (oracle) on the FULL input of the timed steps on this host, ~1-2 min (default: only with 1 GPU at full cfg2 size and a CPU sample; 0 = skip)")
    ap.add_argument("--legal-leg", type=int, default=-1, help="also time ONE step of the fastest reference-legal schedule within 5 %% of the reference's streams: 32 builders, one group, Consensus::getRead's seed rule (default: only with 1 GPU at full cfg2 size; 0 = skip)")
    ap.add_argument("--nonideal-leg", type=int, default=-1, help="also time ONE step on a genome with planted repeats (default: only with 1 GPU at full cfg2 size; 0 = skip)")
    ap.add_argument("--graph", choices=["auto", "host", "device"], default="auto", help="where the contigs' consensus graphs live (nsgpu_set_graph): in HBM (one workgroup per accepted read), on the host (pointer graph), or by the host threads this process has (auto: in HBM with at most 5)")
    ap.add_argument("--graph-leg", type=int, default=-1, help="also time ONE first step with the consensus graphs in the other placement (default: only with 1 GPU at full cfg2 size; 0 = skip)")
    ap.add_argument("--cfg3-leg", type=int, default=-1, help="also time ONE first step of BASELINE configs[2]'s shape (1.0 Gbase at 217x of a 4.6 Mb genome, automatic schedule) (default: only with 1 GPU at full cfg2 size; 0 = skip)")
    ap.add_argument("--no-exchange", action="store_true", help="multi-GPU: independent shards, no collective")
    ap.add_argument("--dist-mode", choices=["alltoall", "replicate"], default="alltoall",
                    help="multi-GPU bucket tables: owners of an RCCL all-to-all of (slot, key, id) tuples, or all-gathered sketch rows")
    args = ap.parse_args()

    pre_tsweep = None
    _full = int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.reads == 100000 and args.depth == 20.0 and args.genome == "iid" and args.mean_len == 8000.0
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and (args.threads_sweep if args.threads_sweep >= 0 else int(_full)):
        pre_tsweep = threads_sweep(args)

    import torch
    import nanospring_amd as ns

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the product has no CPU fallback)")
    local %= max(torch.cuda.device_count(), 1)      # (test rigs with fewer GPUs than ranks share devices; needs the gloo override)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("NSGPU_BENCH_BACKEND", "nccl")      # "nccl" = RCCL; "gloo" only for single-GPU test rigs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    k, n, thr = 23, 60, 6
    salts = ns.mt19937_64_salts(n, 12345)
    exchange = world > 1 and not args.no_exchange
    genome_len = int(world * args.reads * args.mean_len / args.depth)  # 20x depth by default; one genome for the whole job
    # rank r owns read ids [r*R, (r+1)*R) of one read set
    bases, off = ns.synth_reads(11, genome_len, args.reads, args.mean_len, first=rank * args.reads, genome=args.genome)
    n_bases = int(off[-1])

    stream = torch.cuda.Stream()
    g = ns.NsGpu(k=k, n=n, overlap_sketch_thr=thr, device=local, stream=stream.cuda_stream)
    if args.seed_tail_rings < 0:
        args.seed_tail_rings = args.seed_rings
    auto_sched = args.groups == 0
    def apply_schedule(ctx):
        if auto_sched:
            ns.filter.check(ctx.lib, ctx.lib.nsgpu_set_schedule_auto(ctx.ctx))       # builders, bucket depth and radii from the input (include/nsgpu.h)
        else:
            ns.set_schedule(ctx, args.groups, args.seed_depth, args.seed_rings, args.seed_tail_rings)
    apply_schedule(g)
    graph_mode = {"auto": ns.GRAPH_AUTO, "host": ns.GRAPH_HOST, "device": ns.GRAPH_DEVICE}[args.graph]
    if args.graph != "auto":
        ns.set_graph(g, graph_mode)          # (auto: the library's own choice -- NSGPU_GRAPH, else by the host threads of this process)
    job = None
    if exchange:
        # the C++ driver (csrc/dist.hip): the library's own RCCL communicator; Python only calls three entry points
        from nanospring_amd import dist as nd
        job = nd.DistJob(g, dist, backend=os.environ.get("NSGPU_BENCH_BACKEND", "nccl"))
        lo, hi = job.load_reads(bases, off)                             # all-gather of the shards (load time, untimed)
        dmode = nd.ALLTOALL if args.dist_mode == "alltoall" else nd.REPLICATE
    else:
        g.load_reads((bases, off))      # host -> HBM + 2-bit pack: outside the timed region
        ns.filter.check(g.lib, g.lib.nsgpu_set_read_id_base(g.ctx, rank * args.reads))   # global read ids of this shard

    def step():
        if exchange:
            job.sketch_index(salts, dmode)                     # own rows; all-to-all of tuples to the table owners (or all-gather of rows)
            return job.consensus_run(args.builders * world, 8)
        g.sketch(salts, fetch=False)
        g.build_index()
        return ns.consensus_run(g, args.builders, 8)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ns.align_stats(g, reset=True)
    sk_ms = idx_ms = 0.0
    st = None
    waits0, rounds_sum = int(g.lib.nsgpu_host_wait_count()), 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        st = step()
        if rank == 0:
            print("step %d: %.2f s (graph %.1f s, align %.1f s)" % (i, time.perf_counter() - ts, st["graph_ms"] / 1e3, st["align_ms"] / 1e3), file=sys.stderr, flush=True)
        tm = g.timing()
        sk_ms += tm["sketch_kernel_ms"]
        idx_ms += tm["index_ms"]
        rounds_sum += int(st["n_rounds"])
    barrier()
    dt = time.perf_counter() - t0
    waits = int(g.lib.nsgpu_host_wait_count()) - waits0
    dt_local = dt
    if dist is not None:
        cdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tb = torch.tensor([n_bases], device=cdev, dtype=torch.int64)
        dist.all_reduce(tb)
        total_bases = int(tb.item())
    else:
        total_bases = n_bases

    per_rank = None
    if dist is not None:
        # what every rank did, so that a scaling curve can be read: host threads (the node's CPU quota is divided by LOCAL_WORLD_SIZE),
        # collective bytes received, host memory of the replicated read copy, the rank's own step time
        mine = {"rank": rank, "host_threads": ns.align_stats(g)["host_threads"], "s_per_step": round(dt_local / max(args.steps, 1), 3), "bases": n_bases,
                "contigs": st["n_contigs"] if st else 0, "rounds": st["n_rounds"] if st else 0, "consensus_graphs": ns.graph_stats(g)["placement"]}
        if job is not None:
            mine.update(job.comm_stats())
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    hbm_copy = None
    if rank == 0:
        # the HBM roof as this box delivers it to a plain device-to-device copy ("confirm with a copy benchmark on the box")
        try:
            n_copy = 1 << 30
            src = torch.empty(n_copy, dtype=torch.uint8, device="cuda")
            dst = torch.empty_like(src)
            src.fill_(1)
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            hbm_copy = round(10 * 2 * n_copy / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)      # read + write
            del src, dst
        except Exception:
            hbm_copy = None
    # the schedule the timed steps ran in (derived by the library unless --groups was given)
    used = ns.filter.get_schedule(g)
    args.groups, args.seed_depth, args.seed_rings, args.seed_tail_rings = used[0], used[1], used[2], used[3]
    builders_used = used[4] // world if world > 1 else used[4]
    if rank == 0:
        steps = max(args.steps, 1)
        a = ns.align_stats(g)
        bad = ns.consensus_verify(g)         # rank 0's share of the reads (all of them when n_gpus == 1)
        stream_bytes = sum(len(ns.consensus_stream(g, t, kk)) for t in range(8) for kk in ns.filter.STREAMS)
        # dominant kernel: the ksw_extd2 wavefront DP.  Algorithmic bytes per DP problem = its two
        # sequences (1 B/base as coded) + its CIGAR (4 B/op) + the 44-byte result; the traceback matrix
        # is scratch.  A launch = one ksw_extd2 kernel launch (one LDS size class of one DP round);
        # achieved = mean algorithmic bytes per launch / mean launch duration (HIP events on the launch stream).
        launches = max(a["dp_launches"], 1)
        dp_ms = a["dp_kernel_sum_ms"] / launches          # mean duration of one ksw_extd2 launch (launches of a batch overlap on streams)
        alg = a.get("dp_alg_bytes", 0.0) / launches
        achieved = alg / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
        # HBM traffic of the same kernel: PMC counters cannot be read from inside this process; the committed figure
        # comes from rocprofv3 --pmc passes over this very command (profiles/r01_pmc_ksw_traffic.json) and is only
        # reported for the workload it was measured on.
        traffic, traffic_src = None, None
        for pmc_name in (("r05_pmc_ksw_traffic.json", "r04_pmc_ksw_traffic.json", "r03_pmc_ksw_traffic.json") if args.groups == 1 else ("r02_pmc_ksw_traffic.json", "r01_pmc_ksw_traffic.json")):
            pmc = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc) and args.reads == 100000 and world == 1:
                pj = json.load(open(pmc))
                traffic, traffic_src = round(pj["traffic_bytes_per_launch"]), "profiles/%s (rocprofv3 --pmc passes over bench.py at %s: 2*FETCH_SIZE + WRITE_SIZE per launch; traceback scratch dominates)" % (pmc_name, "the default one-group schedule" if args.groups == 1 else "the 1024-builder schedule")
                break
        # what the schedule trades: contigs that grow at the same time compete for reads, so more concurrent builders mean more, shorter
        # contigs and larger streams.  The yard-sticks for THIS input are committed measurements of the oracle (the reference's own
        # OpenMP loop): -t 8 (profiles/r03_oracle_t8_cfg2.json; iso_compression is judged against this, the smallest, one), -t 16 (the thread
        # count of cpu_baseline; 16 threads on the container's 8 cores: profiles/r03_oracle_t16_cfg2.json) and -t 1 (profiles/r02_one_builder_cfg2.json).
        def compression_of(stream_bytes_per_base, stats):
            out = {"builders": stats["n_builders"], "stream_bytes_per_base": round(stream_bytes_per_base, 4), "contigs": stats["n_contigs"], "lone_reads": stats["n_lone"]}
            if args.reads == 100000 and args.mean_len == 8000.0 and world == 1 and args.genome == "iid" and args.depth == 20.0:
                for key, name in (("reference_tN", "r03_oracle_t8_cfg2.json"), ("reference_t16", "r03_oracle_t16_cfg2.json"), ("reference_t1", "r02_one_builder_cfg2.json")):
                    pth = os.path.join(ROOT, "profiles", name)
                    if os.path.exists(pth):
                        oj = json.load(open(pth))
                        out[key] = {"threads": oj.get("threads", 1), "stream_bytes_per_base": round(oj["stream_bytes_per_base"], 4), "contigs": oj["stats"]["n_contigs"],
                                    "lone_reads": oj["stats"]["n_lone"], "source": "profiles/%s (oracle/consensus_oracle.cpp, reference minimap2)" % name}
                        out["ratio_to_" + key] = round(stream_bytes_per_base / oj["stream_bytes_per_base"], 4)
                if "ratio_to_reference_tN" in out:
                    out["iso_compression"] = out["ratio_to_reference_tN"] <= 1.05
            return out
        def parity_of(fixture, stats):
            """The run's streams against what the ORACLE's lock-step virtual threads recorded for this very input and schedule
            (profiles/<fixture>, tools/oracle_lockstep_cfg2.py): sha256 per stream type over all output sets in order.  (.id is left out: the
            8 merged output sets of this run carry the lone-read id deltas of several builders in one list; the GPU test compares it set by set.)"""
            import hashlib
            pth = os.path.join(ROOT, "profiles", fixture)
            if not os.path.exists(pth) or not (args.reads == 100000 and args.mean_len == 8000.0 and world == 1 and args.genome == "iid" and args.depth == 20.0):
                return None
            want = json.load(open(pth))
            same = {}
            for kk in ns.filter.STREAMS:
                if kk == "id":
                    continue
                h = hashlib.sha256()
                for t in range(8):
                    h.update(ns.consensus_stream(g, t, kk))
                same[kk] = h.hexdigest() == want["sha256_over_threads_in_order"][kk]
            return {"fixture": "profiles/" + fixture, "schedule_of_fixture": want["schedule"], "checked_against": "this repository's oracle (lock-step virtual threads, the same seed policy): parity-unpinned against reference bytes",
                    "streams_identical_to_the_oracle_6_of_7": same, "all_identical": all(same.values()),
                    "contigs_slots_equal": stats["n_contigs"] == want["stats"]["n_contigs"] and stats["n_rounds"] == want["stats"]["slots"],
                    "test": "tests/test_consensus_gpu.py::test_cfg2_full_default_schedule_equals_lockstep_oracle_hashes"}
        default_sched = (builders_used, args.groups, args.seed_depth, args.seed_rings, args.seed_tail_rings) == (80, 1, 3, 5, 3)
        parity = parity_of("r03_lockstep_cfg2.json", st) if default_sched else None
        penalty = compression_of(stream_bytes / n_bases, st)
        penalty["schedule"] = {"groups": args.groups, "seed_bucket_depth": args.seed_depth, "seed_rings": args.seed_rings, "seed_tail_rings": args.seed_tail_rings,
                               "derived_by": "the library (nsgpu_set_schedule_auto: from reads, bases and the whole-read filter results per read)" if auto_sched else "the command line"}
        dfr = ns.get_defer(g)
        penalty["schedule"]["deferred_alignments"] = {"anchors_above": dfr[0], "more_slots": dfr[1], "in_the_last_step": dfr[2]}      # nsgpu_set_defer (the automatic schedule: 4096 / 2)
        # three steps (after one untimed) of the 1024-builder, four-group pipelined schedule with the reference's seed rule (the round-2 headline): faster, larger streams
        tleg = None
        want_leg = args.throughput_leg if args.throughput_leg >= 0 else int(world == 1 and args.reads == 100000 and args.depth == 20.0 and args.genome == "iid")
        if want_leg and world == 1 and not (builders_used == 1024 and args.groups == 4 and args.seed_depth == 0):
            ns.set_schedule(g, 4, 0, 1)
            g.sketch(salts, fetch=False); g.build_index()
            ns.consensus_run(g, 1024, 8)                                   # warm-up (buffers of this batch size)
            torch.cuda.synchronize()
            leg_steps, leg_ms = 2, []
            for _ in range(leg_steps):
                tt = time.perf_counter()
                g.sketch(salts, fetch=False); g.build_index()
                st2 = ns.consensus_run(g, 1024, 8)
                torch.cuda.synchronize()
                leg_ms.append((time.perf_counter() - tt) * 1e3)
            dt2 = sum(leg_ms) / 1e3 / leg_steps
            sb2 = sum(len(ns.consensus_stream(g, t, kk)) for t in range(8) for kk in ns.filter.STREAMS)
            tleg = {"value": round(n_bases / 1e6 / dt2, 2), "unit": "Mbases/s", "ms_per_step": round(dt2 * 1e3, 1), "steps": leg_steps, "ms_of_each_step": [round(x, 1) for x in leg_ms],
                    "schedule": {"builders": 1024, "groups": 4, "seed_bucket_depth": 0}, "lossless_roundtrip_bad_reads": ns.consensus_verify(g),
                    "compression": compression_of(sb2 / n_bases, st2), "parity": parity_of("r03_lockstep_cfg2_1024.json", st2),
                    "note": "NOT iso-compression: 1024 contigs grow at once on a 40 Mb genome and cut each other short"}
            apply_schedule(g)
        full_size = world == 1 and args.reads == 100000 and args.depth == 20.0 and args.genome == "iid" and args.mean_len == 8000.0
        # the fastest schedule found that the reference's -t N can itself produce (d = 0: no bucket policy, Consensus::getRead's rule; one group) with
        # streams within 5 % of its -t N: 32 builders (24 / 32 / 40 builders: x1.027 / x1.046 / x1.054 of the -t 8 streams, 47.6 / 58.4 / 69.3 Mbases/s)
        lleg = None
        want_legal = args.legal_leg if args.legal_leg >= 0 else int(full_size)
        if want_legal and world == 1:
            ns.set_schedule(g, 1, 0, 1, 1)
            g.sketch(salts, fetch=False); g.build_index()
            tt = time.perf_counter()
            g.sketch(salts, fetch=False); g.build_index()
            st3 = ns.consensus_run(g, 32, 8)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - tt
            sb3 = sum(len(ns.consensus_stream(g, t, kk)) for t in range(8) for kk in ns.filter.STREAMS)
            lleg = {"value": round(n_bases / 1e6 / dt3, 2), "unit": "Mbases/s", "ms_per_step": round(dt3 * 1e3, 1), "steps": 1,
                    "schedule": {"builders": 32, "groups": 1, "seed_bucket_depth": 0}, "lossless_roundtrip_bad_reads": ns.consensus_verify(g),
                    "compression": compression_of(sb3 / n_bases, st3), "slots": st3["n_rounds"],
                    "note": "an interleaving the reference's own -t N can produce (no try_lock ever fails, its getRead seed rule); the headline schedule's seed rule is this repository's extension"}
            apply_schedule(g)
        # one step on data that is not the best case: a genome with planted repeats
        nleg = None
        want_non = args.nonideal_leg if args.nonideal_leg >= 0 else int(full_size)
        if want_non and world == 1:
            rb, ro = ns.synth_reads(11, genome_len, args.reads, args.mean_len, genome="repeats")
            g2 = ns.NsGpu(k=k, n=n, overlap_sketch_thr=thr, device=local, stream=stream.cuda_stream)
            apply_schedule(g2)
            g2.load_reads((rb, ro))
            ns.align_stats(g2, reset=True)
            tt = time.perf_counter()
            g2.sketch(salts, fetch=False); g2.build_index()
            st4 = ns.consensus_run(g2, args.builders, 8)
            torch.cuda.synchronize()
            dt4 = time.perf_counter() - tt
            a4 = ns.align_stats(g2)
            sb4 = sum(len(ns.consensus_stream(g2, t, kk)) for t in range(8) for kk in ns.filter.STREAMS)
            nleg = {"value": round(int(ro[-1]) / 1e6 / dt4, 2), "unit": "Mbases/s", "ms_per_step": round(dt4 * 1e3, 1), "steps": 1, "first_step": True,
                    "genome": "planted repeats: a 4 kb duplication, a 1.5 kb tandem repeat, a homopolymer run, an (AT)n / (ACGT)n run per ~150 kb",
                    "seed_pairs": {"gpu": a4["seed_pairs_gpu"], "host": a4["seed_pairs_host"]},
                    "device_plan": {"alignments_planned_on_device": a4["plan_pairs_dev"], "left_to_host": a4["plan_pairs_host"], "problems_found": a4["plan_hits"], "not_found": a4["plan_misses"]},
                    "stream_bytes_per_base": round(sb4 / int(ro[-1]), 4), "contigs": st4["n_contigs"], "slots": st4["n_rounds"], "lossless_roundtrip_bad_reads": ns.consensus_verify(g2)}
            g2.close()
            del rb, ro
        # what a rank gets of a CPU quota it shares with the other ranks of its node: ONE step with 4 and with 2 host threads (the thread count is
        # read once per process: a child each, this very script without its other legs; the children's first step includes their allocations)
        tsweep = pre_tsweep
        if tsweep is not None:
            tsweep["note"] = tsweep["note"] % a["host_threads"]
        gstats = ns.graph_stats(g)               # of the last timed step
        # ONE first step with the consensus graphs in the other placement (same schedule, same streams): what the choice costs / buys on this host
        gleg = None
        want_gl = args.graph_leg if args.graph_leg >= 0 else int(full_size)
        if want_gl and world == 1:
            other = ns.GRAPH_HOST if gstats["placement"] == "device" else ns.GRAPH_DEVICE
            ns.set_graph(g, other)
            tt = time.perf_counter()
            g.sketch(salts, fetch=False); g.build_index()
            st5 = ns.consensus_run(g, args.builders, 8)
            torch.cuda.synchronize()
            dt5 = time.perf_counter() - tt
            sb5 = sum(len(ns.consensus_stream(g, t, kk)) for t in range(8) for kk in ns.filter.STREAMS)
            gs5 = ns.graph_stats(g)
            gleg = {"consensus_graphs": gs5["placement"], "value": round(n_bases / 1e6 / dt5, 2), "unit": "Mbases/s", "ms_per_step": round(dt5 * 1e3, 1), "steps": 1, "first_step": True, "host_threads": a["host_threads"],
                    "same_stream_bytes_as_the_timed_steps": sb5 == stream_bytes, "contigs": st5["n_contigs"], "slots": st5["n_rounds"], "lossless_roundtrip_bad_reads": ns.consensus_verify(g),
                    "parity": parity_of("r03_lockstep_cfg2.json", st5) if default_sched else None, "graph_host_wall_ms": round(st5["graph_ms"], 1), "kernels": gs5 if gs5["placement"] == "device" else None}
            ns.set_graph(g, graph_mode)
        # BASELINE configs[2]'s shape: ONE first step of 1.0 Gbase at 217x of a 4.6 Mb genome in the automatic schedule (the E. coli regime: depth, not size)
        c3leg = None
        want_c3 = args.cfg3_leg if args.cfg3_leg >= 0 else int(full_size)
        if want_c3 and world == 1:
            b3, o3 = ns.synth_reads(11, 4600000, 125000, 8000.0)      # (the input of profiles/r04_oracle_t8_cfg3.json)
            g3 = ns.NsGpu(k=k, n=n, overlap_sketch_thr=thr, device=local, stream=stream.cuda_stream)
            ns.filter.check(g3.lib, g3.lib.nsgpu_set_schedule_auto(g3.ctx))
            g3.load_reads((b3, o3))
            tt = time.perf_counter()
            g3.sketch(salts, fetch=False); g3.build_index()
            st6 = ns.consensus_run(g3, 0, 8)
            torch.cuda.synchronize()
            dt6 = time.perf_counter() - tt
            sb6 = sum(len(ns.consensus_stream(g3, t, kk)) for t in range(8) for kk in ns.filter.STREAMS)
            u6 = ns.filter.get_schedule(g3)
            c3leg = {"value": round(int(o3[-1]) / 1e6 / dt6, 2), "unit": "Mbases/s", "ms_per_step": round(dt6 * 1e3, 1), "steps": 1, "first_step": True,
                     "workload": "cfg3's shape: 125 000 synthetic ONT reads, mean 8 kb, 217x of a 4.6 Mb iid genome (1.0 Gbase)", "consensus_graphs": ns.graph_stats(g3)["placement"],
                     "schedule": {"builders": u6[4], "groups": u6[0], "seed_bucket_depth": u6[1], "seed_rings": u6[2], "seed_tail_rings": u6[3], "derived_by": "the library"},
                     "stream_bytes_per_base": round(sb6 / int(o3[-1]), 4), "reference_t8_stream_bytes_per_base": 0.1062, "ratio_to_reference_t8": round(sb6 / int(o3[-1]) / 0.1062, 4),
                     "reference_source": "profiles/r04_oracle_t8_cfg3.json (oracle/consensus_oracle.cpp -t 8 on this input)", "contigs": st6["n_contigs"], "slots": st6["n_rounds"], "lossless_roundtrip_bad_reads": ns.consensus_verify(g3)}
            g3.close()
            del b3, o3
        comp = None
        pv = os.path.join(ROOT, "profiles", "r02_pmc_ksw_issue.json")
        if args.groups == 1:
            pv = next((x for x in (os.path.join(ROOT, "profiles", nm) for nm in ("r05_pmc_ksw_issue.json", "r04_pmc_ksw_issue.json", "r03_pmc_ksw_issue.json")) if os.path.exists(x)), pv)
        if os.path.exists(pv):
            pj2 = json.load(open(pv))
            comp = pj2.get("summary") or {"source": pj2.get("source"), "reading": pj2.get("reading"), "kernels": {k: {"valu_utilisation": v["valu_utilisation"], "waves_per_simd": v["waves_per_simd"], "wave_time_share": v["wave_time_share"]} for k, v in pj2.get("kernels", {}).items()}}
        out = {
            "metric": "Mbases/sec sketch+overlap+align, 8kb ONT reads",
            "value": round(total_bases * steps / 1e6 / dt, 2),
            "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i8", "data": "synthetic",
            "config": {"workload": f"cfg2: {args.reads} synthetic ONT reads/GPU, mean {args.mean_len:.0f} b, {args.depth:g}x of an iid genome, "
                                   f"1% sub + 1% ins + 1% del, k=23 n=60 thr=6, minimap k=20 w=50 max_chain_iter=400, salts mt19937_64(12345)"
                                   + ("" if args.genome == "iid" else "; genome with planted repeats (a 4 kb duplication, a 1.5 kb tandem repeat, a homopolymer run, an (AT)n / (ACGT)n run per ~150 kb): NOT the cfg2 genome"),
                       "genome": args.genome,
                       "stages": ["sketch", "bucket-tables", "overlap (window queries)", "align (batched alignRead, DP on GPU)",
                                  "consensus graph (%s) + edit emission (host)" % ("structure of arrays in HBM, one workgroup per accepted read" if gstats["placement"] == "device" else "pointer graph on the host")],
                       # where the contigs' consensus graphs lived in the timed steps (nsgpu_set_graph; auto = by host threads) and, in HBM, what their kernels did in
                       # the last step: n_sequential_updates / n_full_walks / n_long_reports are the slow paths taken (there is no host fall-back)
                       "consensus_graph": gstats,
                       "bases_per_gpu": n_bases, "builders": st["n_builders"], "schedule": {"groups": args.groups, "seed_bucket_depth": args.seed_depth, "seed_rings": args.seed_rings, "seed_tail_rings": args.seed_tail_rings},
                       "host_threads": a["host_threads"],
                       "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1048576.0, 1),
                       "lossless_roundtrip_bad_reads": bad, "stream_bytes_per_base": round(stream_bytes / n_bases, 4),
                       "contigs": st["n_contigs"], "lone_reads": st["n_lone"], "reads_aligned": st["count_aligner"], "align_calls": st["n_align_calls"],
                       "rounds": st["n_rounds"],
                       # index + seeds + chaining scores of the alignments: pairs done by the kernels (seeds.hip, chain.hip) / handed back
                       # to the host code (anchors sharing a reference position, oversize lists), over the timed steps
                       "seed_pairs": {"gpu": a["seed_pairs_gpu"], "host": a["seed_pairs_host"]},
                       # the alignment plan on the device (plan.hip): alignments whose DP problems were planned and launched behind the chaining kernel
                       # without a host round trip / left to the host's plan; problems the host's own plan asked for and found / did not find
                       "device_plan": {"alignments_planned_on_device": a["plan_pairs_dev"], "left_to_host": a["plan_pairs_host"], "problems_found": a["plan_hits"], "not_found": a["plan_misses"], "unasked": a["plan_extra"]},
                       # host waits for GPU work (stream / event waits inside the library) per slot of the contig stage, over the timed steps
                       "host_waits_per_slot": round(waits / max(rounds_sum, 1), 2),
                       "stage_ms_per_step": {"sketch": round(sk_ms / steps, 2), "tables": round(idx_ms / steps, 2),
                                             "contig_stage_total": round(st["total_ms"], 1), "window_queries": round(st["filter_ms"], 1),
                                             "consensus_index": round(st["index_ms"], 1), "align_total": round(st["align_ms"], 1),
                                             "align_dp_kernel_wall": round(a["dp_kernel_ms"] / steps, 1), "align_dp_kernel_sum": round(a["dp_kernel_sum_ms"] / steps, 1), "graph_host_wall": round(st["graph_ms"], 1), "builder_steps_host_cpu": round(st["graph_cpu_ms"], 1),
                                             "note": "graph_host_wall = wall of the phases in which the host applies alignments (it waits for the DP results there); builder_steps_host_cpu = CPU time of the builders' own steps in sum over the host threads, last step (the pointer graph's updates are in it; with the graphs in HBM what is left is the hand-over). With four groups the contig-stage parts overlap (host phase | batches part 1 | DP in flight | batches part 2) and do not add up to the total; with one group they run one after the other"},
                       "parallelism": (f"x{world}: reads sharded by id, replicated by all-gather at load; per step "
                                       + ("RCCL all-to-all of (slot, key, id) tuples to the bucket-table owners (table j on rank j % world) + all-gather of the sorted tables"
                                          if args.dist_mode == "alltoall" else "all-gather of sketch rows") +
                                       f" + {st.get('n_collectives', 0)} small all-gathers of claim lists (global builder order); C++ driver, library-owned RCCL communicator") if exchange
                       else f"reads sharded by id x{world}, no collective"},
            "per_rank": per_rank,
            "compression": penalty,
            "parity": parity,
            "throughput_schedule": tleg,
            "reference_legal_schedule": lleg,
            "nonideal": nleg,
            "consensus_graph_other_placement": gleg,
            "cfg3": c3leg,
            "host_threads_sweep": tsweep,
            "roofline": {"kernel": "ksw_extd2 (ksw_extd2_reg_kernel<NW,NCH>: DP state in registers)", "bound": "latency" if args.groups == 1 else "valu-issue", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 7), "hbm_frac": round(achieved / HBM_PEAK_GBS, 7), "hbm_copy_measured_gbs": hbm_copy,
                         "traffic": None, "traffic_from_profile": traffic, "traffic_source": traffic_src,
                         "launches": int(a["dp_launches"]), "avg_launch_ms": round(dp_ms, 3),
                         # the kernel is integer DP bound by instruction issue, not by HBM: its real ceiling as first-class fields
                         "compute": {"bound": "latency of one wave per problem at the default schedule (a launch holds ~80 alignments' problems; a round waits for its slowest ALIGNMENT -- the DP kernels hand each one over when its last problem is done); VALU + SALU instruction issue at the 1024-builder schedule", "cells": a["dp_cells"],
                                     "gcups_over_dp_wall": round(a["dp_cells"] / (a["dp_kernel_ms"] * 1e-3) / 1e9, 1) if a["dp_kernel_ms"] else 0,
                                     "gcups_over_kernel_sum": round(a["dp_cells"] / (a["dp_kernel_sum_ms"] * 1e-3) / 1e9, 1) if a["dp_kernel_sum_ms"] else 0,
                                     "pmc": comp},
                         "note": "achieved / peak / frac are the HBM figures the contract asks for (algorithmic bytes per launch / launch time; ~1e-5 by construction: 1 B of sequence per ~250 DP cells); "
                                 "neither the HBM nor the MFMA roof applies: at the one-group schedule an alignment is as long as the dependent chain of its longest problem, 1 000-1 800 anti-diagonals of ~1 us (bound = latency: "
                                 "compute.pmc shows the waves waiting, not issuing), at the 1024-builder schedule the kernels are bound by VALU + SALU instruction issue; traffic is not "
                                 "measured inside this run, traffic_from_profile is the committed rocprofv3 --pmc figure for the workload it names"},
        }
        if args.cpu_sample != 0:
            want_full = args.cpu_full if args.cpu_full >= 0 else int(full_size)
            # (with the full input timed below the bounded sample is only a side figure: a quarter of it, and a short -t 1 run)
            cb = cpu_baseline(args.cpu_sample if args.cpu_sample > 0 else (500 if want_full and world == 1 else 2000) * host_cores(), args.mean_len, k, n, thr, salts, t1_reads=600 if want_full and world == 1 else 1500)
            if want_full and world == 1:
                # the reference's -t <cores> on the very input and host of the timed steps: the baseline proper, and the yard-stick of `compression`
                fi = cpu_full_input(bases, off, k, n, thr, salts)
                cb["bounded_sample"] = {"value": cb["value"], "sample": cb["sample"], "stream_bytes_per_base": cb["sample_stream_bytes_per_base"]}
                cb["value"], cb["sample"] = fi["value"], "the FULL input of the timed steps (%d reads / %.1f Mbases) on this host, -t %d in %.1f s: %d reads aligned into %d contigs (%d lone reads), streams %.4f B/base" % (
                    args.reads, n_bases / 1e6, fi["cores"], fi["seconds"], fi["reads_aligned"], fi["contigs"], fi["lone_reads"], fi["stream_bytes_per_base"])
                cb["full_input_same_host"] = fi
                out["compression"]["reference_tN_same_run"] = {"threads": fi["cores"], "stream_bytes_per_base": fi["stream_bytes_per_base"], "contigs": fi["contigs"], "lone_reads": fi["lone_reads"],
                                                               "source": "oracle/consensus_oracle.cpp -t %d on this host in this run (cpu_baseline.full_input_same_host)" % fi["cores"]}
                out["compression"]["ratio_to_reference_tN_same_run"] = round(out["compression"]["stream_bytes_per_base"] / fi["stream_bytes_per_base"], 4)
                out["speedup_over_cpu_same_input_same_host"] = round(out["value"] / fi["value"], 2)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if job is not None:
        job.close()
    g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
