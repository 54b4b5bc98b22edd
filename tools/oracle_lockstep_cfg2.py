#!/usr/bin/env python3
"""The bench's DEFAULT schedule (B builders in one group, conflict-aware seeds) on bench.py's full cfg2 input, computed on the CPU by the
oracle's lock-step virtual threads (oracle/consensus_oracle.cpp struct LockStep: the literal thread body of the reference under the
schedule DESIGN.md 2 documents, the reference's minimap2 answering every alignRead).  Records sizes, counters, the slot count and one
sha256 per stream type over the B thread files in thread order -- the fixtures of
tests/test_consensus_gpu.py::test_cfg2_full_default_schedule_equals_lockstep_oracle_hashes.

    python tools/oracle_lockstep_cfg2.py [builders depth rings tail_rings] [out.json] [n_reads] [genome_len] [every k-th read] [name]

cfg3's schedule on every 5th read of the cfg3 input (BASELINE configs[2]; 25 000 reads over the 4.6 Mb genome):
    python tools/oracle_lockstep_cfg2.py 256 1 4 3 profiles/r04_lockstep_cfg3_fifth.json 125000 4600000 5 cfg3

Full size: 8 minutes on 8 cores and 54 GB of memory at the peak (80 naive graphs, the whole genome in flight half way through).
"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from tests import oracle_lib

B, depth, rings, tail = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (80, 3, 5, 3)))
groups = int(os.environ.get("NS_ORACLE_GROUPS", "1"))
out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "r03_lockstep_cfg2.json")
n_reads = int(sys.argv[6]) if len(sys.argv) > 6 else 100000
genome = int(sys.argv[7]) if len(sys.argv) > 7 else int(n_reads * 8000 / 20)
stride = int(sys.argv[8]) if len(sys.argv) > 8 else 1
wname = sys.argv[9] if len(sys.argv) > 9 else "cfg2"
bases, off = ns.synth_reads(11, genome, n_reads, 8000.0)
if stride > 1:
    import numpy as np
    keep = np.arange(0, n_reads, stride)
    parts = [bases[int(off[i]):int(off[i + 1])] for i in keep]
    off = np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.uint64)
    bases = np.concatenate(parts)
n_hash = int(os.environ.get("NS_ORACLE_NHASH", "60"))          # --num-hash (BASELINE configs[4] sweeps 60 / 128)
salts = ns.mt19937_64_salts(n_hash, 12345)
t0 = time.time()
streams, st = oracle_lib.cons_oracle_run(bases, off, salts, n=n_hash, num_thr=B, checks=False, lock_step=True, groups=groups, seed_hops=depth, seed_rings=rings, seed_tail_rings=tail)
dt = time.time() - t0
names = oracle_lib.CONS_STREAMS
sha, size = {}, {}
for n in names:
    h = hashlib.sha256()
    tot = 0
    for t in streams["threads"]:
        h.update(t[n]); tot += len(t[n])
    sha[n], size[n] = h.hexdigest(), tot
sha["metaData"], size["metaData"] = hashlib.sha256(streams["metaData"]).hexdigest(), len(streams["metaData"])
tot7 = sum(size[n] for n in names)
rec = {"workload": "%s (bench.py's generator: seed 11, %d reads, mean 8000, genome %d%s)" % (wname, n_reads, genome, ", every %dth read" % stride if stride > 1 else ""),
       "num_hash": n_hash, "schedule": {"builders": B, "groups": groups, "seed_bucket_depth": depth, "seed_rings": rings, "seed_tail_rings": tail},
       "computed_by": "oracle/consensus_oracle.cpp lock-step virtual threads, reference minimap2 (oracle/_ref/libmm2ref.so)",
       "seconds": dt, "bases": int(off[-1]), "stream_bytes": size, "sha256_over_threads_in_order": sha, "stats": st,
       "stream_bytes_total_7": tot7, "stream_bytes_per_base": tot7 / int(off[-1])}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec)[:1500])
