#!/usr/bin/env python3
"""Large-sample parity of the whole path in the reference's deterministic schedule: the GPU engine with ONE builder
(= NanoSpring -t 1) against the sequential restatement whose alignments are answered by the reference's own minimap2
(oracle/_ref/libmm2ref.so; falls back to the CPU oracle DP when that object is absent).  All eight streams must be
byte-identical.  parity_full.py [reads = 20000] [mean_len = 8000]   (20x depth; 20 000 reads = 160 Mbases: ~5 min)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS
from tests import host_lib, oracle_lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
mean = float(sys.argv[2]) if len(sys.argv) > 2 else 8000.0
bases, off = ns.synth_reads(11, int(n * mean / 20), n, mean)
salts = ns.mt19937_64_salts(60, 12345)
have_ref = oracle_lib.mm2ref() is not None
t = time.time()
want, wst = host_lib.consensus(bases, off, salts, checks=False, ref_aligner=have_ref)
t_cpu = time.time() - t
g = ns.NsGpu()
g.load_reads((bases, off))
g.sketch(salts, fetch=False)
g.build_index()
t = time.time()
st = ns.consensus_run(g, 1, 1)
t_gpu = time.time() - t
bad = [k for k in STREAMS if ns.consensus_stream(g, 0, k) != want[k]]
print(f"{n} reads, {int(off[-1]) / 1e6:.0f} Mbases; sequential restatement ({'reference minimap2' if have_ref else 'oracle DP'}) {t_cpu:.0f} s, "
      f"GPU engine with 1 builder {t_gpu:.0f} s; contigs {st['n_contigs']} / {wst['n_contigs']}, aligned {st['count_aligner']} / {wst['count_aligner']}; "
      f"streams differing: {bad if bad else 'none'}; lossless round trip bad reads: {ns.consensus_verify(g)}")
g.close()
sys.exit(1 if bad else 0)
