#!/usr/bin/env python3
"""BASELINE configs[2] at size (1.0 Gbase of 8 kb reads over a 4.6 Mb genome, 217x): throughput and stream size of a list of schedules,
beside the reference's -t 8 streams on the same input (profiles/r04_oracle_t8_cfg3.json).

    python tools/cfg3_sweep.py "B,G,depth,rings,tail" ...      e.g. "80,1,3,5,3" "256,1,3,2,1" "1024,4,0,1,1"
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS

ref = json.load(open(os.path.join(ROOT, "profiles", "r04_oracle_t8_cfg3.json")))
bases, off = ns.synth_reads(11, 4600000, 125000, 8000.0)
nb = int(off[-1])
assert nb == ref["bases"]
g = ns.NsGpu()
g.load_reads((bases, off))
salts = ns.mt19937_64_salts(60, 12345)
for spec in sys.argv[1:]:
    B, G, d, r, t = (int(x) for x in spec.split(","))
    for rep in range(1):
        t0 = time.perf_counter()
        g.sketch(salts, fetch=False)
        g.build_index()
        st = ns.consensus_run(g, B, 8, schedule=(G, d, r, t))
        dt = time.perf_counter() - t0
    sb = sum(len(ns.consensus_stream(g, th, k)) for th in range(8) for k in STREAMS)
    print(json.dumps({"builders": B, "groups": G, "depth": d, "rings": r, "tail": t, "mbases_per_s": round(nb / 1e6 / dt, 2), "s": round(dt, 2), "B_per_base": round(sb / nb, 4),
                      "ratio_to_ref_t8": round(sb / nb / ref["stream_bytes_per_base"], 4), "contigs": st["n_contigs"], "lone": st["n_lone"], "slots": st["n_rounds"], "bad": ns.consensus_verify(g)}), flush=True)
g.close()
