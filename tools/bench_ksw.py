#!/usr/bin/env python3
"""Micro-benchmark of the ksw_extd2 kernels on production-shaped batches.
    python tools/bench_ksw.py [n]            gap fills of ~265 x 265, band 751, flag 0x08 (94% of the calls)
    python tools/bench_ksw.py --long [n]     extensions of 1500..5000 x same, band 751, zdrop 400, flag 0x40 (exact max, the long class)
Prints GCUPS from the HIP-event kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib

args = [a for a in sys.argv[1:] if not a.startswith("--")]
long_set = "--long" in sys.argv
n = int(args[0]) if args else (600 if long_set else 20000)
rng = np.random.RandomState(1)
probs = []
for i in range(n):
    if long_set:
        ql = int(rng.randint(1500, 5000))
        q, t = oracle_lib.ksw_random_problem(rng, ql, ql + int(rng.randint(-60, 61)), err=0.04)
        probs.append((q, t, 751, 400, -1, 0x40 | (0x82 if i & 1 else 0)))
    else:
        ql = int(rng.randint(200, 330))
        q, t = oracle_lib.ksw_random_problem(rng, ql, ql + int(rng.randint(-8, 9)), err=0.04)
        probs.append((q, t, 751, 400, -1, 0x08))
g = ns.NsGpu()
cells = sum(len(q) * len(t) for q, t, *_ in probs)
band_cells = sum(min(len(q) * len(t), (len(q) + len(t)) * 752 // 2) for q, t, *_ in probs)
for it in range(3):
    t0 = time.perf_counter()
    ns.ksw_extd2_batch(g, probs)
    dt = time.perf_counter() - t0
    print(f"batch {n}: wall {dt*1e3:.1f} ms  ({cells/dt/1e9:.1f} GCUPS incl. host packing + copies)")
st = ns.align_stats(g)
print(f"kernels: {st['dp_launches']} launches, sum {st['dp_kernel_sum_ms']:.2f} ms -> {3 * cells / st['dp_kernel_sum_ms'] / 1e6:.1f} GCUPS (kernel only, qlen x tlen)"
      f"; {3 * band_cells / st['dp_kernel_sum_ms'] / 1e6:.1f} GCUPS counting only the cells inside the band")
