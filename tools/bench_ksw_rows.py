#!/usr/bin/env python3
"""Latency per anti-diagonal of the register DP kernels: 64 equal square problems per size (the chip is nearly empty, as in a launch of the
default contig schedule), kernel time / (qlen + tlen - 1).   python tools/bench_ksw_rows.py [flag]      flag 0x08 = gap fill, 0x40 = extension"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib

flag = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x08
g = ns.NsGpu()
rng = np.random.RandomState(3)
for L in (120, 200, 250, 300, 400, 500, 600, 700, 930, 1200, 1500, 2000):
    probs = []
    for i in range(64):
        q, t = oracle_lib.ksw_random_problem(rng, L, L, err=0.10)
        probs.append((q, t, 751, 400, -1, flag))
    ns.ksw_extd2_batch(g, probs)
    ns.align_stats(g, reset=True)
    for _ in range(3):
        ns.ksw_extd2_batch(g, probs)
    st = ns.align_stats(g)
    ms = st["dp_kernel_sum_ms"] / 3
    print(f"L {L:5d}: kernel {ms:7.3f} ms, {1e3 * ms / (2 * L - 1):6.3f} us per anti-diagonal, active blocks per row <= {min(L, 751) / 128:.1f}")
