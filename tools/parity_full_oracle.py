#!/usr/bin/env python3
"""Whole-path byte parity at BASELINE cfg2's full size against the INDEPENDENT oracle: the GPU engine with ONE builder on bench.py's
input (100 000 reads, 801 Mbases) must produce streams whose sha256 are the ones profiles/r02_one_builder_cfg2.json records for
oracle/consensus_oracle.cpp at -t 1 (an hour of CPU, computed once; the oracle shares no code with the product and had the reference's
own minimap2 answering every alignRead).  ~4 minutes on the GPU box."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS

want = json.load(open(os.path.join(ROOT, "profiles", "r02_one_builder_cfg2.json")))
bases, off = ns.synth_reads(11, int(100000 * 8000 / 20), 100000, 8000.0)
g = ns.NsGpu()
g.load_reads((bases, off))
g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
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
print(f"GPU engine, 1 builder: {dt:.0f} s ({int(off[-1]) / 1e6 / dt:.1f} Mbases/s), contigs {st['n_contigs']} (oracle {want['stats']['n_contigs']}), "
      f"aligned {st['count_aligner']} (oracle {want['stats']['count_aligner']}), lossless round trip: {ns.consensus_verify(g)} bad reads")
print("PARITY", "OK" if ok else "FAILED")
g.close()
sys.exit(0 if ok else 1)
