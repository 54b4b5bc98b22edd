#!/usr/bin/env python3
"""Overhead of the multi-GPU driver loop: the cfg2 contig stage through nsgpu_consensus_run (C++ slot loop) and through
nanospring_amd.dist.consensus_exchange (Python slot loop + one fixed-size RCCL all-gather per request list) on ONE rank
(world size 1, backend nccl = RCCL).  Same data, same schedule, same streams; the difference is the driver's cost per slot."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

import nanospring_amd as ns
from nanospring_amd import dist as nd

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bases, off = ns.synth_reads(11, int(n * 8000 / 20), n, 8000.0)
g = ns.NsGpu()
g.load_reads((bases, off))
salts = ns.mt19937_64_salts(60, 12345)
for rep in range(3):
    g.sketch(salts, fetch=False)
    g.build_index()
    t = time.time()
    a = ns.consensus_run(g, 1024, 8)
    ta = time.time() - t
    sa = ns.consensus_stream(g, 0, "pos")
    nd.exchange_sketch_rows(g, salts, 0, n, dist)
    g.build_index()
    t = time.time()
    b = nd.consensus_exchange(g, 1024, dist, 8)
    tb = time.time() - t
    sb = ns.consensus_stream(g, 0, "pos")
    print(f"rep {rep}: C++ loop {ta:.2f} s, exchange loop {tb:.2f} s ({b['n_collectives']} collectives), same streams: {sa == sb}", flush=True)
g.close()
dist.destroy_process_group()
