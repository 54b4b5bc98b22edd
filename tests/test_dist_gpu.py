"""Exchange-mode multi-GPU contig stage (nanospring_amd/dist.py) with 2 and 3 ranks sharing the one GPU of the
test box (collectives over gloo): the result must not depend on the number of ranks -- same contigs, same
counters as the single-process run with the same total number of builders -- and be lossless."""
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import nanospring_amd as ns
from nanospring_amd import dist as nd
from nanospring_amd.filter import STREAMS
from tests.stream_decode import decode

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Several ranks on ONE GPU (these tests; a deployment has one process per GPU): with the package's eight hardware queues per process the ranks' queues
# outnumber what the GPU runs at once and take turns -- a rank's DP kernels can then sit unscheduled while its graph workgroups wait for the orders
# those kernels' results lead to.  Two queues per process keep every queue on the GPU.
SHARED_GPU = {"GPU_MAX_HW_QUEUES": "2"}

N_READS, N_BUILDERS = 500, 24


def single():
    bases, off = ns.synth_reads(41, 120000, N_READS, 3000.0)
    g = ns.NsGpu()
    g.load_reads((bases, off))
    g.sketch(ns.mt19937_64_salts(60), fetch=False)
    g.build_index()
    st = ns.consensus_run(g, N_BUILDERS, 1)
    s = {k: ns.consensus_stream(g, 0, k) for k in STREAMS}
    assert ns.consensus_verify(g) == 0
    g.close()
    return bases, off, st, s


@pytest.mark.parametrize("world,driver", [(2, "py"), (3, "py"), (2, "replicate"), (3, "replicate"), (2, "alltoall"), (3, "alltoall")])
def test_result_is_independent_of_rank_count(world, driver):
    """"py": the phase calls driven from Python; "replicate" / "alltoall": the C++ driver (nsgpu_dist_load_reads / _sketch_index /
    _consensus_run) with the bucket tables built from all-gathered sketch rows, or by the owners of an all-to-all of (slot, key,
    id) tuples (table j on rank j % world) -- the same tables, the same contigs as one process."""
    bases, off, st1, s1 = single()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.pkl")
        port = str(29600 + world + 10 * ["py", "replicate", "alltoall"].index(driver))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, NSGPU_THREADS="4", **SHARED_GPU)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                            "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(N_READS),
                            str(N_BUILDERS), out, driver], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res = pickle.load(open(out, "rb"))
    assert len(res) == world
    got, genomes, lones = {}, [], []
    for streams, md, st, bad in res:
        assert bad == 0
        d = decode(streams)
        assert not set(d) & set(got)
        got.update(d)
        genomes += streams["genome"].split(b"\n")[:-1]
        lones += streams["lone"].split(b"\n")[:-1]
    b = bytes(bases)
    assert sorted(got) == list(range(N_READS))
    for i in range(N_READS):
        assert got[i] == b[int(off[i]):int(off[i + 1])]
    # same contigs and counters as one process with the same number of builders
    assert sorted(genomes) == sorted(s1["genome"].split(b"\n")[:-1])
    assert sorted(lones) == sorted(s1["lone"].split(b"\n")[:-1])
    for k in ("n_contigs", "n_lone", "count_minhash", "count_minhash_not_in_graph", "count_aligner", "n_align_calls"):
        assert sum(x[2][k] for x in res) == st1[k], k
    m = nd.parse_meta(nd.merge_meta([x[1] for x in res]))
    assert m["numReads"] == N_READS and m["numThr"] == world and sum(m["numReadsInContig"]) == N_READS


@pytest.mark.parametrize("driver", ["py", "replicate", "alltoall"])
def test_exchange_path_over_rccl_world_size_one(driver):
    """The box has one GPU, so RCCL cannot run two ranks here; a world of one still drives every collective of the
    exchange path (all-gathers of read shards, sketch rows and claim lists, on device tensors) through the real "nccl"
    backend, and must reproduce the single-process run bit for bit.  NSGPU_TEST_FORCE_EXCHANGE makes the world of one take the
    device get -> all_gather_into_tensor -> set path of exchange_sketch_rows (re-importing its own rows) instead of skipping it.
    With the C++ drivers the library's OWN RCCL communicator (dlopen of librccl, ncclCommInitRank from a unique id made by rank 0)
    carries the collectives -- including, in "alltoall" mode with one rank, the degenerate send-to-self."""
    bases, off, st1, s1 = single()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.pkl")
        port = str(29641 + ["py", "replicate", "alltoall"].index(driver))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, NSGPU_TEST_BACKEND="nccl", NSGPU_TEST_FORCE_EXCHANGE="1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", port, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(N_READS), str(N_BUILDERS), out, driver],
                           env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res = pickle.load(open(out, "rb"))
    assert len(res) == 1
    streams, md, st, bad = res[0]
    assert bad == 0
    for k in STREAMS:
        assert streams[k] == s1[k], k
    for k in ("n_contigs", "n_lone", "count_minhash", "count_aligner", "n_align_calls"):
        assert st[k] == st1[k], k


@pytest.mark.parametrize("world,groups,depth", [(2, 4, 0), (3, 1, 3), (2, 2, 2)])
def test_cxx_driver_over_callback_communicator(tmp_path, world, groups, depth):
    """INTEGRATION.md section 3b executed, not only described: a C++11 program (tests/integration/dist_stage.cpp: no Python, W forked
    processes on the one GPU, collectives as host callbacks over shared memory -- where a C++ host would plug MPI) drives
    nsgpu_dist_load_reads / _sketch_index / _consensus_run; its contigs are those of one process with the same number of builders and
    the same schedule, every read decodes, and a rank's host copy of ALL reads is the 2-bit rows (<= 0.3 B/base + 24 B/read), never
    the ASCII text."""
    exe = tmp_path / "dist_stage"
    lib_dir = os.path.dirname(ns.lib_path())
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "integration", "dist_stage.cpp"),
                        "-o", str(exe), "-L", lib_dir, "-lnsgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    bases, off = ns.synth_reads(41, 120000, N_READS, 3000.0)
    b = bytes(bases)
    reads = [b[int(off[i]):int(off[i + 1])] for i in range(N_READS)]
    fq = tmp_path / "reads.fastq"
    with open(fq, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
    out = tmp_path / "tmp"
    out.mkdir()
    for rk in range(world):
        (out / ("rank%d" % rk)).mkdir()
    r = subprocess.run([str(exe), str(fq), str(out) + "/", str(world), str(N_BUILDERS), str(groups), str(depth), "1"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, NSGPU_THREADS="4"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "lossless check: 0 bad reads" in r.stdout
    # one process, same builders, same schedule
    g = ns.NsGpu()
    g.load_reads((bases, off))
    g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
    g.build_index()
    st1 = ns.consensus_run(g, N_BUILDERS, 1, schedule=(groups, depth, 1))
    want_genomes = sorted(ns.consensus_stream(g, 0, "genome").split(b"\n")[:-1])
    g.close()
    tot = {ln.split(" = ")[0]: int(ln.split(" = ")[1]) for ln in r.stdout.splitlines() if " = " in ln}
    assert tot["numContigs"] == st1["n_contigs"] and tot["#LoneReads"] == st1["n_lone"]
    got, genomes = {}, []
    for rk in range(world):
        streams = {e: open(out / ("rank%d" % rk) / ("Stream.tid.0." + e), "rb").read() for e in STREAMS}
        d = decode(streams)
        assert not set(d) & set(got)
        got.update(d)
        genomes += streams["genome"].split(b"\n")[:-1]
    assert sorted(got) == list(range(N_READS)) and all(got[i] == reads[i] for i in range(N_READS))
    assert sorted(genomes) == want_genomes
    n_bases = int(off[-1])
    for ln in r.stdout.splitlines():
        if ln.startswith("rank "):
            f = ln.split()
            host_bytes, ag, aa = int(f[f.index("copy") + 1]), int(f[f.index("all-gather") + 2]), int(f[f.index("all-to-all") + 2])
            assert host_bytes <= 0.3 * n_bases + 24 * N_READS + (1 << 16), ln
            assert ag > 0.25 * n_bases and aa > 0, ln                      # the packed rows and the bucket tuples did travel


@pytest.mark.parametrize("mode", ["alltoall", "replicate", "alltoall-auto"])
def test_bench_two_ranks_on_one_gpu_default_schedule(mode):
    """`bench.py --gpus 2` itself, launched the way the driver launches it (torch.distributed.run, one process per rank; the launcher starts
    before anything touches the GPU), the two ranks sharing the box's one GPU with the collectives over gloo (NSGPU_BENCH_BACKEND): the
    line is the contract's, every rank reports its host threads, step time and collective bytes, and in the DEFAULT schedule (one group,
    conflict-aware seeds of depth 3) the job's contigs are those of one process holding all reads with the same number of builders."""
    import json
    R, B, L = 600, 12, 3000.0
    auto_count = mode.endswith("-auto")             # no --builders either: the count is the library's too (1 per 10 Mbases of the whole job, at least 32)
    mode = mode.split("-")[0]
    port = "29673" if auto_count else "29671" if mode == "alltoall" else "29672"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, NSGPU_BENCH_BACKEND="gloo", NSGPU_THREADS="4", **SHARED_GPU)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", str(R), "--mean-len", str(L)] + ([] if auto_count else ["--builders", str(B)]) +
                       ["--cpu-sample", "0", "--dist-mode", mode], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["warmup"] == 1 and j["scaling"] == "weak" and j["higher_is_better"] is True and j["value"] > 0
    assert j["config"]["schedule"] == {"groups": 1, "seed_bucket_depth": 3, "seed_rings": 5, "seed_tail_rings": 3}
    pr = j["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1]
    for p in pr:
        assert p["host_threads"] >= 1 and p["s_per_step"] > 0 and p["bases"] > 0
        assert p["all_gather_bytes"] > 0 and p["host_bytes_of_the_read_copy"] > 0
        assert (p["all_to_all_bytes"] > 0) == (mode == "alltoall")
    # one process, all reads, the same number of builders, the same schedule
    bases, off = ns.synth_reads(11, int(2 * R * L / 20), 2 * R, L)
    assert sum(p["bases"] for p in pr) == int(off[-1])
    g = ns.NsGpu()
    g.load_reads((bases, off))
    g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
    g.build_index()
    st1 = ns.consensus_run(g, 0, 8) if auto_count else ns.consensus_run(g, 2 * B, 8, schedule=(1, 3, 5, 3))
    if auto_count:
        assert ns.filter.get_schedule(g) == (1, 3, 5, 3, 32) and j["config"]["builders"] == 32
    assert ns.consensus_verify(g) == 0
    g.close()
    assert sum(p["contigs"] for p in pr) == st1["n_contigs"] and all(p["rounds"] == st1["n_rounds"] for p in pr)


def test_bench_four_ranks_on_one_gpu_two_host_threads_each_graphs_in_hbm():
    """What a rank of a full node gets: four ranks (one GPU here, collectives over gloo) with TWO host threads each -- the library then keeps the
    contigs' consensus graphs in HBM (nsgpu_set_graph's automatic rule) -- in the default schedule: every rank reports it, its step time is
    printed, and the job's contigs and slots are those of one process holding all reads (the result does not depend on the rank count or on
    where the graphs live)."""
    import json
    R, B, L = 400, 8, 3000.0
    port = "29677"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, NSGPU_BENCH_BACKEND="gloo", NSGPU_THREADS="2", **SHARED_GPU)
    env.pop("NSGPU_GRAPH", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", port,
                        os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--reads", str(R), "--mean-len", str(L), "--builders", str(B),
                        "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    pr = j["per_rank"]
    assert j["n_gpus"] == 4 and [p["rank"] for p in pr] == [0, 1, 2, 3]
    for p in pr:
        assert p["host_threads"] == 2 and p["consensus_graphs"] == "device" and p["s_per_step"] > 0
    print("per-rank step times with 2 host threads each:", [p["s_per_step"] for p in pr])
    bases, off = ns.synth_reads(11, int(4 * R * L / 20), 4 * R, L)
    g = ns.NsGpu()
    g.load_reads((bases, off))
    g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
    g.build_index()
    st1 = ns.consensus_run(g, 4 * B, 8, schedule=(1, 3, 5, 3))
    assert ns.consensus_verify(g) == 0
    g.close()
    assert sum(p["contigs"] for p in pr) == st1["n_contigs"] and all(p["rounds"] == st1["n_rounds"] for p in pr)
