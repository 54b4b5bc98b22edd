#!/usr/bin/env python3
"""Debugging aid: the ksw_extd2 kernels against oracle/ksw2_oracle.c on a diverse seeded problem set, mismatches summarised per
shape class (which kernel served the problem is a function of tlen: <= 256, <= 512, <= 1536, <= 5120 cells per anti-diagonal)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib
from tests.ksw_cases import diverse_cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
probs = diverse_cases(seed, n)
g = ns.NsGpu()
orc = oracle_lib.Oracle()
t0 = time.time()
ezs, cigs = ns.ksw_extd2_batch(g, probs)
print("gpu batch %.2f s" % (time.time() - t0))
bad = {}
tot = {}
for i, (q, t, w, zd, eb, fl) in enumerate(probs):
    cls = 0 if len(t) <= 256 else 1 if len(t) <= 512 else 2 if len(t) <= 1536 else 3 if len(t) <= 5120 else 4
    key = (cls, "approx" if fl & 8 else "exact", "right" if fl & 2 else "left")
    tot[key] = tot.get(key, 0) + 1
    we, wc = oracle_lib.oracle_ksw(orc, q, t, w, zd, eb, fl)
    if ezs[i] != we or not np.array_equal(cigs[i], wc):
        bad.setdefault(key, []).append((i, len(q), len(t), w, zd, hex(fl), ezs[i], we, len(cigs[i]), len(wc)))
for k in sorted(tot):
    b = bad.get(k, [])
    print(k, "problems", tot[k], "bad", len(b))
    for x in b[:3]:
        print("   ", x)
print("TOTAL bad", sum(len(v) for v in bad.values()), "of", len(probs))
