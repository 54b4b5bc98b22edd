#!/usr/bin/env python3
"""Repeated contig-stage runs in one process: resident memory must level off (graph slabs circulate between threads)."""
import sys, os, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nanospring_amd as ns


def rss_gb():
    with open("/proc/self/status") as f:
        for l in f:
            if l.startswith("VmRSS"):
                return int(l.split()[1]) / 1e6
    return 0.0


n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bases, off = ns.synth_reads(11, int(n * 8000 / 20), n, 8000.0)
g = ns.NsGpu()
g.load_reads((bases, off))
g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
g.build_index()
print(f"after load: RSS {rss_gb():.2f} GB")
for i in range(reps):
    t = time.time()
    st = ns.consensus_run(g, 1024, 20)
    print(f"run {i}: {time.time() - t:.2f} s, RSS {rss_gb():.2f} GB", flush=True)
g.close()
