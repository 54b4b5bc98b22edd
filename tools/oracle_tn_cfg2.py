#!/usr/bin/env python3
"""The reference's own -t N schedule on bench.py's cfg2 input, computed on the CPU by the oracle (oracle/consensus_oracle.cpp: the
reference's OpenMP loop with optimistic try_lock claiming, the reference's minimap2 answering every alignRead).  Records contigs,
lone reads and the stream sizes -- the iso-compression yard-stick `bench.py` prints beside its own stream size.  Timing-dependent
for N > 1 (SURVEY 0 trap 3), so the figure is one sample of the reference's distribution; run it a few times to see the spread.

    python tools/oracle_tn_cfg2.py <threads> [out.json] [n_reads] [genome_len] [workload name]

cfg3 (BASELINE configs[2], ~1 Gbase at 217x): python tools/oracle_tn_cfg2.py 8 profiles/r04_oracle_t8_cfg3.json 125000 4600000 cfg3
"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests import oracle_lib
import nanospring_amd as ns

T = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r03_oracle_t%d_cfg2.json" % T)
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
os.environ["OMP_NUM_THREADS"] = str(T)
genome = int(sys.argv[4]) if len(sys.argv) > 4 else int(n_reads * 8000 / 20)
wname = sys.argv[5] if len(sys.argv) > 5 else "cfg2"
bases, off = ns.synth_reads(11, genome, n_reads, 8000.0)
salts = ns.mt19937_64_salts(60, 12345)
t0 = time.time()
streams, st = oracle_lib.cons_oracle_run(bases, off, salts, num_thr=T, checks=False)
dt = time.time() - t0
names = oracle_lib.CONS_STREAMS
th = streams["threads"] if "threads" in streams else [streams]
tot = {n: sum(len(t[n]) for t in th) for n in names}
tot7 = sum(tot.values())
rec = {
    "workload": "%s (bench.py's generator: seed 11, %d reads, mean 8000, %.0fx of a %.2f Mb genome)" % (wname, n_reads, n_reads * 8000.0 / genome, genome / 1e6),
    "schedule": "oracle/consensus_oracle.cpp -t %d (the reference's OpenMP loop and try_lock claiming; reference minimap2)" % T,
    "threads": T, "host_cpus": os.cpu_count(), "seconds": dt, "bases": int(off[-1]), "mbases_per_s": int(off[-1]) / 1e6 / dt,
    "stream_bytes": dict(tot, metaData=len(streams["metaData"])), "stats": st,
    "stream_bytes_total_7": tot7, "stream_bytes_per_base": tot7 / int(off[-1]),
    "note": "timing-dependent for -t > 1: one sample of the reference's own distribution on this input",
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
