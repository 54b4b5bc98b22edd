#!/usr/bin/env python3
"""sha256 over all output streams of one contig-stage run (cfg2 shape): the result is a deterministic function of
(reads, salts, n_builders), so two builds / two settings of an exact shortcut must print the same hash.
stream_hash.py [reads = 100000] [builders = 1024] [threads_out = 8]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 8
bases, off = ns.synth_reads(11, int(n * 8000 / 20), n, 8000.0)
g = ns.NsGpu()
g.load_reads((bases, off))
g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
g.build_index()
st = ns.consensus_run(g, nb, nt)
h = hashlib.sha256()
for t in range(nt):
    for k in STREAMS:
        h.update(ns.consensus_stream(g, t, k))
print(f"{n} reads, {nb} builders: contigs {st['n_contigs']} aligned {st['count_aligner']} sha256 {h.hexdigest()} bad {ns.consensus_verify(g)}")
g.close()
