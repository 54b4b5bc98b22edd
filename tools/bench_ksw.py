#!/usr/bin/env python3
"""Micro-benchmark of the ksw_extd2 kernel on production-shaped batches (SURVEY 3.5:
94% of calls are ~240x240 gap fills).  Prints GCUPS from the HIP-event kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.RandomState(1)
probs = []
for i in range(n):
    ql = int(rng.randint(200, 330))
    q, t = oracle_lib.ksw_random_problem(rng, ql, ql + int(rng.randint(-8, 9)), err=0.04)
    probs.append((q, t, 751, 400, -1, 0x08))
g = ns.NsGpu()
cells = sum(len(q) * len(t) for q, t, *_ in probs)
for it in range(3):
    t0 = time.perf_counter()
    ns.ksw_extd2_batch(g, probs)
    dt = time.perf_counter() - t0
    print(f"batch {n}: wall {dt*1e3:.1f} ms  ({cells/dt/1e9:.1f} GCUPS incl. host packing + copies)")
st = ns.align_stats(g)
print(f"kernels: {st['dp_launches']} launches, sum {st['dp_kernel_sum_ms']:.2f} ms -> {3 * cells / st['dp_kernel_sum_ms'] / 1e6:.1f} GCUPS (kernel only)")
