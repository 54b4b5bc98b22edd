import sys, os
sys.path.insert(0, '.')
import numpy as np
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS
def run(mode):
    bases, off = ns.synth_reads(31, 150000, 600, 4000.0)
    g = ns.NsGpu()
    g.load_reads((bases, off)); g.sketch(ns.mt19937_64_salts(60), fetch=False); g.build_index()
    ns.set_graph(g, mode)
    ns.set_schedule(g, 1, 0, 1)
    st = ns.consensus_run(g, 40, 40)
    out = [{k: ns.consensus_stream(g, t, k) for k in STREAMS} for t in range(40)]
    g.close()
    return out, st
a, sa = run(ns.GRAPH_HOST)
b, sb = run(ns.GRAPH_DEVICE)
print(sa["count_aligner"], sb["count_aligner"], sa["n_contigs"], sb["n_contigs"])
for t in range(40):
    for k in STREAMS:
        if a[t][k] != b[t][k]:
            x, y = a[t][k], b[t][k]
            d = next((i for i in range(min(len(x), len(y))) if x[i] != y[i]), min(len(x), len(y)))
            print("DIFF thread", t, k, "len", len(x), len(y), "first diff", d, x[max(0,d-8):d+8], y[max(0,d-8):d+8])
