import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns
from tests.align_cases import make_genome, mutate, revcomp
rng = np.random.RandomState(3)
g0 = make_genome(rng, 1200000)
reads = []
for i in range(14):
    ln = int(rng.randint(150000, 400000))
    st = rng.randint(0, len(g0) - ln)
    s = mutate(rng, g0[st:st + ln], 0.02)
    reads.append(revcomp(s) if rng.randint(2) else s)
for i in range(400):
    ln = int(max(500, rng.gamma(2.0, 4000.0)))
    st = rng.randint(0, len(g0) - ln)
    s = mutate(rng, g0[st:st + ln], 0.03)
    reads.append(revcomp(s) if rng.randint(2) else s)
order = rng.permutation(len(reads))
reads = [reads[i] for i in order]
g = ns.NsGpu()
g.load_reads(reads)
g.sketch(ns.mt19937_64_salts(60), fetch=False)
g.build_index()
for nb in (1, 32):
    t = time.time()
    st = ns.consensus_run(g, nb, 2)
    print("builders", nb, "time %.1fs" % (time.time() - t), {k: st[k] for k in ("n_contigs", "n_lone", "count_aligner", "n_align_calls")}, "bad", ns.consensus_verify(g), flush=True)
g.close()
