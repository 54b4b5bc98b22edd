#!/usr/bin/env python3
"""Builders vs compression on the CPU: the oracle's lock-step virtual threads (the product's schedule, oracle/consensus_oracle.cpp struct
LockStep) on a 1/s-scale cfg2 input, for builder counts and seed policies.  Prints contigs, lone reads, stream B/base and slots.

    python tools/seed_policy_sim.py <n_reads> <builders,builders,...> <hops,hops,...>
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from tests import oracle_lib

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
builders = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,16,128").split(",")]
hops = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2").split(",")]
groups = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "4").split(",")]
bases, off = ns.synth_reads(11, int(n_reads * 8000 / 20), n_reads, 8000.0)
salts = ns.mt19937_64_salts(60, 12345)
nb = int(off[-1])
for B in builders:
    for h, G in [(h, G) for h in hops for G in groups]:
        if B == 1 and (h != hops[0] or G != groups[0]):
            continue
        t0 = time.time()
        streams, st = oracle_lib.cons_oracle_run(bases, off, salts, num_thr=B, checks=False, lock_step=True, seed_hops=h & 255, seed_rings=(h >> 8) or 1, groups=G)
        dt = time.time() - t0
        th = [streams] if B == 1 else streams["threads"]
        tot = sum(len(t[n]) for t in th for n in oracle_lib.CONS_STREAMS)
        assert st["n_bad_roundtrip"] == 0
        print(json.dumps({"reads": n_reads, "builders": B, "seed_hops": h, "groups": G, "contigs": st["n_contigs"], "lone": st["n_lone"], "aligned": st["count_aligner"],
                          "align_calls": st["n_align_calls"], "stream_B_per_base": round(tot / nb, 5), "slots": st["slots"], "idle_seed_rounds": st["idle_seed_rounds"],
                          "seconds": round(dt, 1)}), flush=True)
