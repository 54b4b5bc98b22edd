#!/usr/bin/env python3
"""BASELINE configs[4]'s knobs at cfg2's size against the INDEPENDENT oracle: the GPU engine with ONE builder on bench.py's input with
--num-hash 128 (edge threshold 4 M) must produce the streams whose sizes and sha256 profiles/r05_one_builder_cfg5knobs.json records for
oracle/consensus_oracle.cpp at -t 1 (an hour of CPU, tools/oracle_one_builder.py).  ~5 minutes on the GPU box; its output is committed as
profiles/r05_parity_cfg5_one_builder.txt (the 80-builder schedule at these knobs is a test of the suite: r05_lockstep_cfg5knobs.json)."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS

want = json.load(open(os.path.join(ROOT, "profiles", "r05_one_builder_cfg5knobs.json")))
n_hash = want["num_hash"]
bases, off = ns.synth_reads(11, int(want["n_reads"] * 8000 / 20), want["n_reads"], 8000.0)
g = ns.NsGpu(n=n_hash, edge_threshold=want["edge_thr"])
g.load_reads((bases, off))
g.sketch(ns.mt19937_64_salts(n_hash, 12345), fetch=False)
g.build_index()
t0 = time.time()
st = ns.consensus_run(g, 1, 1)
dt = time.time() - t0
ok = True
for k in STREAMS + ["metaData"]:
    b = ns.consensus_stream(g, 0, k)
    same = hashlib.sha256(b).hexdigest() == want["sha256"][k] and len(b) == want["stream_bytes"][k]
    ok &= same
    print(f"{k:12s} {len(b):10d} bytes  {'identical to the oracle' if same else 'DIFFERS'}")
print(f"GPU engine, 1 builder, --num-hash {n_hash}: {dt:.0f} s ({int(off[-1]) / 1e6 / dt:.1f} Mbases/s), contigs {st['n_contigs']} (oracle {want['stats']['n_contigs']}), "
      f"aligned {st['count_aligner']} (oracle {want['stats']['count_aligner']}), lossless round trip: {ns.consensus_verify(g)} bad reads")
print("PARITY", "OK" if ok else "FAILED")
g.close()
sys.exit(0 if ok else 1)
