#!/usr/bin/env python3
"""BASELINE configs[3]'s per-GPU share on ONE GPU: 625 000 reads of mean 10 kb (6.25 Gbases: beyond 2^32 bases, so every offset on the path
is exercised past 32 bits) over a 312.5 Mb genome (20x), the automatic schedule (nsgpu_consensus_run with 0 builders), `runs` steps: every
read decodes, repeated runs give one stream hash, peak host memory is printed (and its ratio to the input's bases).

    python tools/cfg4_share.py [runs = 1] [reads = 625000] [mean = 10000]
"""
import hashlib, json, os, resource, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS, get_schedule

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 625000
mean = float(sys.argv[3]) if len(sys.argv) > 3 else 10000.0
t0 = time.perf_counter()
bases, off = ns.synth_reads(11, int(reads * mean / 20), reads, mean)
nb = int(off[-1])
print("INPUT", reads, nb, "%.1f s to generate" % (time.perf_counter() - t0), flush=True)
g = ns.NsGpu()
g.load_reads((bases, off))
del bases
salts = ns.mt19937_64_salts(60, 12345)
for i in range(runs):
    t0 = time.perf_counter()
    g.sketch(salts, fetch=False)
    g.build_index()
    st = ns.consensus_run(g, 0, 16)
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    tot = 0
    for t in range(16):
        for k in STREAMS:
            b = ns.consensus_stream(g, t, k)
            h.update(b)
            tot += len(b)
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0
    print("RUN", i, json.dumps({"schedule": get_schedule(g), "s": round(dt, 2), "mbases_per_s": round(nb / 1e6 / dt, 1), "contigs": st["n_contigs"], "lone": st["n_lone"],
                                "aligned": st["count_aligner"], "slots": st["n_rounds"], "bad_reads": ns.consensus_verify(g), "stream_bytes_per_base": round(tot / nb, 4),
                                "sha256": h.hexdigest(), "peak_rss_gb": round(rss, 1), "rss_bytes_per_base": round(rss * (1 << 30) / nb, 2)}), flush=True)
g.close()
