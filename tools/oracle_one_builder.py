#!/usr/bin/env python3
"""The reference's -t 1 schedule on bench.py's cfg2-shaped input with other knobs (BASELINE configs[4]: --num-hash 128, --edge-thr 4M), computed
once on the CPU by oracle/consensus_oracle.cpp with the reference's own minimap2: sizes and sha256 of the eight streams, the fixture of
tests/test_consensus_gpu.py::test_cfg5_knobs_at_size_one_builder_equals_oracle_hashes.  About an hour on one core.

    python tools/oracle_one_builder.py <num_hash> <edge_thr> <out.json> [n_reads]
"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib
import nanospring_amd as ns

n_hash, edge_thr, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n_reads = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
os.environ["OMP_NUM_THREADS"] = "1"
bases, off = ns.synth_reads(11, int(n_reads * 8000 / 20), n_reads, 8000.0)
salts = ns.mt19937_64_salts(n_hash, 12345)
t0 = time.time()
streams, st = oracle_lib.cons_oracle_run(bases, off, salts, n=n_hash, edge_thr=edge_thr, num_thr=1, checks=False)
dt = time.time() - t0
names = oracle_lib.CONS_STREAMS + ["metaData"]
rec = {"workload": "cfg2-shaped input (seed 11, %d reads, mean 8000, 20x) with cfg5's knobs: --num-hash %d, --edge-thr %d" % (n_reads, n_hash, edge_thr),
       "schedule": "oracle/consensus_oracle.cpp -t 1 (reference minimap2)", "num_hash": n_hash, "edge_thr": edge_thr, "n_reads": n_reads, "seconds": dt, "bases": int(off[-1]),
       "stream_bytes": {k: len(streams[k]) for k in names}, "sha256": {k: hashlib.sha256(streams[k]).hexdigest() for k in names}, "stats": st}
rec["stream_bytes_total_7"] = sum(rec["stream_bytes"][k] for k in oracle_lib.CONS_STREAMS)
rec["stream_bytes_per_base"] = rec["stream_bytes_total_7"] / rec["bases"]
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
