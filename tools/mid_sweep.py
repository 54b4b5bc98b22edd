#!/usr/bin/env python3
"""A third point for the automatic schedule: 0.8 Gbase of 8 kb reads at 60x (13.3 Mb genome), between cfg2 (20x) and cfg3 (217x): throughput
and stream size of a list of schedules beside the reference's -t 8 streams on the same input (profiles/r05_oracle_t8_mid60x.json).

    python tools/mid_sweep.py auto "B,G,depth,rings,tail" ...
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS, get_schedule

ref = json.load(open(os.path.join(ROOT, "profiles", "r05_oracle_t8_mid60x.json")))
bases, off = ns.synth_reads(11, 13333333, 100000, 8000.0)
nb = int(off[-1])
assert nb == ref["bases"]
g = ns.NsGpu()
g.load_reads((bases, off))
salts = ns.mt19937_64_salts(60, 12345)
for spec in sys.argv[1:]:
    t0 = time.perf_counter()
    g.sketch(salts, fetch=False)
    g.build_index()
    if spec == "auto":
        st = ns.consensus_run(g, 0, 8, schedule="auto")
    else:
        B, G, d, r, t = (int(x) for x in spec.split(","))
        st = ns.consensus_run(g, B, 8, schedule=(G, d, r, t))
    dt = time.perf_counter() - t0
    sb = sum(len(ns.consensus_stream(g, th, k)) for th in range(8) for k in STREAMS)
    print(json.dumps({"spec": spec, "schedule_used": get_schedule(g), "mbases_per_s": round(nb / 1e6 / dt, 2), "s": round(dt, 2), "B_per_base": round(sb / nb, 4),
                      "ratio_to_ref_t8": round(sb / nb / ref["stream_bytes_per_base"], 4), "contigs": st["n_contigs"], "lone": st["n_lone"], "slots": st["n_rounds"], "bad": ns.consensus_verify(g)}), flush=True)
g.close()
