"""Child process of tests/test_stress_gpu.py: repeated contig-stage runs on one input in one schedule; prints one line per run
(contigs, reads that do not decode, sha256 over all streams).    python tests/stress_worker.py <iid|repeats> <reads> <builders> <runs>"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS

genome, reads, builders, runs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
salts = ns.mt19937_64_salts(60, 12345)
bases, off = ns.synth_reads(11, reads * 400, reads, 8000.0, genome=genome)       # 20x
g = ns.NsGpu()
g.load_reads((bases, off))
for i in range(runs):
    g.sketch(salts, fetch=False)
    g.build_index()
    st = ns.consensus_run(g, builders, 8, schedule=(1, 3, 5, 3))
    h = hashlib.sha256()
    for t in range(8):
        for k in STREAMS:
            h.update(ns.consensus_stream(g, t, k))
    print("RUN", i, st["n_contigs"], ns.consensus_verify(g), h.hexdigest(), flush=True)
g.close()
