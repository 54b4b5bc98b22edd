#!/usr/bin/env python3
"""Runs the whole hot path once on synthetic reads and prints stage timings / counters."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_builders = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mean = float(sys.argv[3]) if len(sys.argv) > 3 else 8000.0
verify = "--verify" in sys.argv
t0 = time.time()
bases, off = ns.synth_reads(11, int(n_reads * mean / 20), n_reads, mean)
print("synth %.1fs, %.1f Mbases" % (time.time() - t0, off[-1] / 1e6), flush=True)
g = ns.NsGpu()
t0 = time.time(); g.load_reads((bases, off)); print("load %.2fs" % (time.time() - t0), flush=True)
t0 = time.time(); g.sketch(ns.mt19937_64_salts(60), fetch=False); g.build_index(); print("sketch+index %.3fs" % (time.time() - t0), flush=True)
ns.align_stats(g, reset=True)
t0 = time.time(); st = ns.consensus_run(g, n_builders, 8); dt = time.time() - t0
print("consensus %.2fs -> %.1f Mbases/s" % (dt, off[-1] / 1e6 / dt))
print(json.dumps(st))
print(json.dumps(ns.align_stats(g)))
if verify:
    t0 = time.time(); print("verify bad =", ns.consensus_verify(g), "%.1fs" % (time.time() - t0))
tot = sum(len(ns.consensus_stream(g, t, k)) for t in range(8) for k in ns.filter.STREAMS)
print("stream bytes %d = %.3f bytes/base" % (tot, tot / off[-1]))
