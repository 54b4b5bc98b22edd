"""Contig stage at other depths / read lengths than cfg2 (robustness + throughput): run_depth.py <reads> <depth> [builders] [mean_len]."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns
n, depth = int(sys.argv[1]), float(sys.argv[2])
mean = float(sys.argv[4]) if len(sys.argv) > 4 else 8000.0
bases, off = ns.synth_reads(11, int(n * mean / depth), n, mean)
g = ns.NsGpu()
g.load_reads((bases, off))
g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
g.build_index()
t = time.time()
st = ns.consensus_run(g, int(sys.argv[3]) if len(sys.argv) > 3 else 1024, 4)
dt = time.time() - t
bad = ns.consensus_verify(g)
nb = sum(len(ns.consensus_stream(g, t, k)) for t in range(4) for k in ns.filter.STREAMS if k != "metaData")
print(f"depth {depth}: {n} reads {int(off[-1])/1e6:.0f} Mbases in {dt:.2f}s = {int(off[-1])/dt/1e6:.1f} Mbases/s; contigs {st['n_contigs']} lone {st['n_lone']} aligned {st['count_aligner']} calls {st['n_align_calls']} minhash {st['count_minhash']} bad {bad} bytes/base {nb/int(off[-1]):.3f}")
g.close()
