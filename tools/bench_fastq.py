#!/usr/bin/env python3
"""FASTQ ingest: text already in host memory -> 2-bit rows in HBM (nsgpu_load_fastq).
Prints the wall time (PCIe H2D of the text included), the GPU time of the parse + pack kernels, and the CPU port
(oracle getline loop + pack, 1 core) on the same text."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bases, off = ns.synth_reads(11, int(n * 8000 / 20), n, 8000.0)
t0 = time.perf_counter()
parts = []
for r in range(n):
    s = bases[int(off[r]):int(off[r + 1])].tobytes()
    parts += [b"@read%d" % r, s, b"+", b"I" * len(s)]
text = np.frombuffer(b"\n".join(parts) + b"\n", dtype=np.uint8)
print(f"FASTQ text {text.size / 1e9:.3f} GB, {n} reads, {int(off[-1]) / 1e6:.1f} Mbases (built in {time.perf_counter() - t0:.1f} s)")
g = ns.NsGpu()
for it in range(3):
    t0 = time.perf_counter()
    nr = g.load_fastq(text)
    dt = time.perf_counter() - t0
    tm = g.timing()
    print(f"load_fastq: {nr} reads, wall {dt * 1e3:.1f} ms = {text.size / dt / 1e9:.2f} GB/s of text ({int(off[-1]) / dt / 1e6:.0f} Mbases/s); "
          f"GPU parse+pack {tm['pack_ms']:.2f} ms = {text.size / (tm['pack_ms'] * 1e-3) / 1e9:.0f} GB/s of text")
assert nr == n and g.num_bases == int(off[-1])
orc = oracle_lib.Oracle()
t0 = time.perf_counter()
st, ln = orc.fastq_index(text)
dt = time.perf_counter() - t0
print(f"oracle getline loop (1 core): {dt * 1e3:.0f} ms = {text.size / dt / 1e9:.2f} GB/s of text")
