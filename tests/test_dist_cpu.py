"""world_size-2 gloo test of the multi-GPU plumbing (nanospring_amd/dist.py) on CPU: shard planning by bases,
global read-id base, gathering rank outputs as extra "threads", merged metaData.  The per-rank engine here is the
sequential CPU restatement (tests/host_harness.cpp) standing in for the GPU engine, which has the same contract
(nsgpu_set_read_id_base + nsgpu_consensus_run); the GPU engine itself is covered by tests/test_consensus_gpu.py."""
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

import nanospring_amd as ns
from nanospring_amd import dist as nd
from tests.stream_decode import decode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
import nanospring_amd as ns
from nanospring_amd import dist as nd
from tests import host_lib
from tests.host_lib import STREAMS

dist.init_process_group("gloo")
bases, off = ns.synth_reads(77, 50000, 220, 2500.0)
salts = ns.mt19937_64_salts(60)

def engine(sb, so, id_base, n_out):
    out, st = host_lib.consensus(sb, so, salts, checks=False, id_base=id_base)
    assert st["n_bad_roundtrip"] == 0
    return [{k: out[k] for k in STREAMS if k != "metaData"}], out["metaData"], st

res = nd.run_sharded(engine, bases, off, dist)
if dist.get_rank() == 0:
    pickle.dump(res, open(sys.argv[1], "wb"))
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_bounds_balance_and_cover():
    rng = np.random.RandomState(0)
    lens = rng.randint(100, 20000, size=1000)
    off = np.zeros(1001, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    for world in (1, 2, 3, 8):
        b = nd.shard_bounds(off, world)
        assert b[0][0] == 0 and b[-1][1] == 1000 and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [int(off[hi] - off[lo]) for lo, hi in b]
        assert max(sizes) - min(sizes) <= 2 * 20000
    assert nd.shard_bounds(np.zeros(1, dtype=np.uint64), 4) == [(0, 0)] * 4


def test_two_ranks_gloo_roundtrip():
    with tempfile.TemporaryDirectory() as td:
        script, out = os.path.join(td, "w.py"), os.path.join(td, "out.pkl")
        open(script, "w").write(WORKER % {"root": ROOT})
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", OMP_NUM_THREADS="2")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29577", script, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        streams, md, stats = pickle.load(open(out, "rb"))
    bases, off = ns.synth_reads(77, 50000, 220, 2500.0)
    b = bytes(bases)
    assert len(streams) == 2 and len(stats) == 2
    got = {}
    for s in streams:
        d = decode(s)
        assert not set(d) & set(got)          # global ids: shards do not collide
        got.update(d)
    assert sorted(got) == list(range(220))
    for i in range(220):
        assert got[i] == b[int(off[i]):int(off[i + 1])]
    m = nd.parse_meta(md)
    assert m["numReads"] == 220 and m["numThr"] == 2 and sum(m["numReadsInContig"]) == 220
    assert m["numContigs"] == len(m["numReadsInContig"]) == sum(s["n_contigs"] for s in stats)
    lo1 = nd.shard_bounds(off, 2)[1][0]
    assert min(decode(streams[1])) == lo1
