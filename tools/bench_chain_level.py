#!/usr/bin/env python3
"""Time of the chaining kernels on tandem-repeat seed lists (tests/test_chain_gpu.py tandem_lists + a 5-mer unit: ~80 000 anchors):
the level kernel (a workgroup per list) against the ring kernel (a wave per list).  NSGPU_CHAIN_LEVEL_MIN=0 selects the ring kernel.
    python tools/bench_chain_level.py"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nanospring_amd as ns
from tests import host_lib
from tests.test_chain_gpu import tandem_lists, gpu_scores

rnd = random.Random(8)
unit = "".join(rnd.choice("ACGT") for _ in range(5))
ref = "".join(rnd.choice("ACGT") for _ in range(2500)) + unit * 300 + "".join(rnd.choice("ACGT") for _ in range(3000))
qry = "".join(ch for ch in ref if rnd.random() > 0.01)
lists = tandem_lists() + [host_lib.seeds(ref, qry)]
g = ns.NsGpu()
for a in lists:
    gpu_scores(g, [a])
    t0 = time.perf_counter()
    for _ in range(5):
        gpu_scores(g, [a])
    dt = (time.perf_counter() - t0) / 5
    r = a[:, 0].astype(np.int64) & 0xffffffff
    print("anchors %6d, distinct reference positions %5d: %.3f ms per call (%.0f ns per anchor)" % (len(a), len(np.unique(r)), dt * 1e3, dt * 1e9 / max(len(a), 1)), flush=True)
g.close()
