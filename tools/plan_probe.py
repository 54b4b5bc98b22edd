#!/usr/bin/env python3
"""The device-side alignment plan (plan.hip) on the alignment test pairs: how many alignments the kernel planned, how many of the DP
problems the host's plan asked for were found among the device-planned results.   python tools/plan_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nanospring_amd as ns
from tests.align_cases import pairs
from tests.align_util import load_align_golden

g = ns.NsGpu()
gold = load_align_golden()
ns.align_stats(g, reset=True)
ns.align_batch(g, gold["refs"], gold["qrys"], gold["pair_ref"])
print("golden pairs:", {k: v for k, v in ns.align_stats(g, reset=True).items() if k.startswith("plan") or k in ("pairs", "dp_tasks", "dp_rounds")})
ps = pairs(2025, 320, big=True)
refs, rid = [], []
for r, _ in ps:
    if r not in refs:
        refs.append(r)
    rid.append(refs.index(r))
ns.align_batch(g, refs, [q for _, q in ps], rid)
print("random pairs:", {k: v for k, v in ns.align_stats(g, reset=True).items() if k.startswith("plan") or k in ("pairs", "dp_tasks", "dp_rounds")})
