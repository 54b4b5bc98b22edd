#!/usr/bin/env python3
"""bench.py -- Mbases/s through the NanoSpring hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic reads that is
already resident (2-bit packed) in HBM: MinHash sketch of every read -> the n
bucket tables -> overlap candidates for every read (forward + reverse-complement
whole-read queries) [-> alignment stages as they land; config.stages lists what
the timed region contains].  Workload = BASELINE.json configs[1]: 100 000
synthetic ONT-like reads, mean 8 kb, k=23, n=60 per GPU.

Multi-GPU: reads shard by id, one process per GPU, no data-path collective in the
stages timed so far (weak scaling: every rank holds its own 100 k reads).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     : dominant kernel's algorithmic bytes / its HIP-event duration vs HBM peak
  cpu_baseline : the CPU oracle (bit-exact restatement, oracle/ns_oracle.c) timed on
                 this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(bases, off, k, n, thr, salts, sample_reads):
    """Oracle (kind "port") on the host cores, bounded sample, same stages as the GPU step."""
    from tests import oracle_lib
    orc = oracle_lib.Oracle()
    ns_ = min(sample_reads, len(off) - 1)
    sb = bases[:int(off[ns_])]
    so = off[:ns_ + 1]
    t0 = time.perf_counter()
    sk = orc.sketch_reads(sb, so, k, n, salts)
    t1 = time.perf_counter()
    idx = orc.index_build(sk)
    t2 = time.perf_counter()
    # overlap queries: forward sketches are the reads' own; RC sketches need the RC strings
    b = bytes(sb)
    comp = bytes.maketrans(b"ATCG", b"TAGC")
    nq = 0
    for r in range(ns_):
        s = b[int(so[r]):int(so[r + 1])]
        orc.filter_sketch(sk[r], idx, thr)
        orc.filter_string(s[::-1].translate(comp), k, salts, idx, thr)
        nq += 2
    t3 = time.perf_counter()
    nb = int(so[-1])
    return {"value": round(nb / 1e6 / (t3 - t0), 3), "unit": "Mbases/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{ns_} reads / {nb / 1e6:.1f} Mbases of the same workload; sketch {t1 - t0:.2f}s (OpenMP), "
                      f"tables {t2 - t1:.2f}s (OpenMP), overlap queries {t3 - t2:.2f}s (1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU (cfg2: 100000)")
    ap.add_argument("--mean-len", type=float, default=8000.0)
    ap.add_argument("--cpu-sample", type=int, default=3000, help="reads in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import nanospring_amd as ns

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    k, n, thr = 23, 60, 6
    salts = ns.mt19937_64_salts(n, 12345)
    genome_len = int(args.reads * args.mean_len / 20)          # 20x depth
    # shard = its own slice of the read-id space: rank r draws reads with seed 11 + r
    bases, off = ns.synth_reads(11 + rank, genome_len, args.reads, args.mean_len)
    n_bases = int(off[-1])

    stream = torch.cuda.Stream()
    g = ns.NsGpu(k=k, n=n, overlap_sketch_thr=thr, device=local, stream=stream.cuda_stream)
    g.load_reads((bases, off))          # host -> HBM + 2-bit pack: outside the timed region

    def step():
        g.sketch(salts, fetch=False)
        g.build_index()
        return g.filter_all_reads(fetch=False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    k_sketch = k_filter = k_index = 0.0
    cands = 0
    for _ in range(args.steps):
        cands = step()
        tm = g.timing()
        k_sketch += tm["sketch_kernel_ms"]
        k_index += tm["index_ms"]
        k_filter += tm["filter_ms"]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tb = torch.tensor([n_bases], device="cuda", dtype=torch.int64)
        dist.all_reduce(tb)
        total_bases = int(tb.item())
    else:
        total_bases = n_bases

    if rank == 0:
        tm = g.timing()
        steps = max(args.steps, 1)
        sk_ms = k_sketch / steps
        # dominant kernel so far: the xor-min sketch kernel.  Algorithmic bytes per launch
        # (SURVEY 8d): 0.25 B/base (2-bit read) + 8n B per read (sketch row).
        alg_bytes = 0.25 * n_bases + 8.0 * n * args.reads
        achieved = alg_bytes / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else 0.0
        out = {
            "metric": "Mbases/sec sketch+overlap+align, 8kb ONT reads",
            "value": round(total_bases * steps / 1e6 / dt, 2),
            "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"cfg2: {args.reads} synthetic ONT reads/GPU, mean {args.mean_len:.0f} b, 20x of an iid genome, "
                                   f"1% sub + 1% ins + 1% del, k=23 n=60 thr=6, salts mt19937_64(12345)",
                       "stages": ["sketch", "bucket-tables", "overlap(fwd+rc whole-read queries)"],
                       "stages_missing": ["align", "consensus-edit"],
                       "bases_per_gpu": n_bases, "candidates_per_step": int(cands),
                       "stage_ms": {"sketch": round(sk_ms, 3), "tables": round(k_index / steps, 3), "overlap": round(k_filter / steps, 3)},
                       "parallelism": f"reads sharded by id x{world}"},
            "roofline": {"kernel": "sketch_kernel<1>", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "note": "integer-ALU bound by construction: %.1f G xor-min/s" % (n * n_bases / (sk_ms * 1e-3) / 1e9 if sk_ms else 0)},
        }
        if args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(bases, off, k, n, thr, salts, args.cpu_sample)
        print(json.dumps(out), flush=True)
    g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
