#!/usr/bin/env python3
"""Repeated contig-stage runs in the schedules the tests pin (cfg3's 256-builder schedule on the cfg3 input, the default schedule on the repeats
genome): every run must be lossless and give the same stream hash as the first one of its kind.  NSGPU_SEGV_TRACE=1 prints a backtrace
should the process die.    python tools/stress_engine.py [runs]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
salts = ns.mt19937_64_salts(60, 12345)
for name, args, builders, sched in (("cfg3", (11, 4600000, 125000, 8000.0), 256, (1, 1, 4, 3)), ("repeats", (11, 8000000, 20000, 8000.0), 80, (1, 3, 5, 3))):
    bases, off = ns.synth_reads(*args, genome="repeats" if name == "repeats" else "iid")
    g = ns.NsGpu()
    g.load_reads((bases, off))
    first = None
    for i in range(runs):
        g.sketch(salts, fetch=False)
        g.build_index()
        st = ns.consensus_run(g, builders, 8, schedule=sched)
        h = hashlib.sha256()
        for t in range(8):
            for k in STREAMS:
                h.update(ns.consensus_stream(g, t, k))
        bad = ns.consensus_verify(g)
        first = first or h.hexdigest()
        print(name, i, st["n_contigs"], bad, h.hexdigest()[:16], "same" if h.hexdigest() == first else "DIFFERENT", flush=True)
        assert bad == 0 and h.hexdigest() == first
    g.close()
print("stress ok")
