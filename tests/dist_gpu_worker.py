"""Worker of tests/test_dist_gpu.py: one rank of the multi-GPU contig stage (all ranks share cuda:0 here; the
collectives run over gloo so that the test needs only one GPU).  argv[4] picks the driver: "py" = the phase calls driven from
Python (nanospring_amd/dist.py), "replicate" / "alltoall" = the C++ entry points nsgpu_dist_* with that bucket-table mode."""
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist

import nanospring_amd as ns
from nanospring_amd import dist as nd
from nanospring_amd.filter import STREAMS

BACKEND = os.environ.get("NSGPU_TEST_BACKEND", "gloo")      # "nccl" (= RCCL) needs one GPU per rank: used with world size 1
if BACKEND == "nccl":
    import torch
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_reads, n_builders, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
driver = sys.argv[4] if len(sys.argv) > 4 else "py"
bases, off = ns.synth_reads(41, 120000, n_reads, 3000.0)
lo, hi = nd.shard_bounds(off, world)[rank]
sb, so = nd.take_shard(bases, off, lo, hi)                  # what this rank "owns" before the exchange
salts = ns.mt19937_64_salts(60)
g = ns.NsGpu()
if driver == "py":
    all_b, all_o, lo2, hi2 = nd.replicate_reads(sb, so, dist)
    assert (lo2, hi2) == (lo, hi) and np.array_equal(all_o, off) and np.array_equal(all_b, bases)
    g.load_reads((all_b, all_o))
    nd.exchange_sketch_rows(g, salts, lo, hi, dist)
    g.build_index()
    st = nd.consensus_exchange(g, n_builders, dist, 1)
else:
    job = nd.DistJob(g, dist, backend=BACKEND)
    assert job.load_reads(sb, so) == (lo, hi)
    job.sketch_index(salts, nd.ALLTOALL if driver == "alltoall" else nd.REPLICATE)
    # the tables every rank ends up with are the ones a single process builds
    ref = ns.NsGpu()
    ref.load_reads((bases, off))
    ref.sketch(salts, fetch=False)
    ref.build_index()
    for j in (0, 1, 7, 33, 59):
        a, b = g.index_export(j), ref.index_export(j)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), ("bucket table", j)
    ref.close()
    st = job.consensus_run(n_builders, 1)
    job.close()
streams = {k: ns.consensus_stream(g, 0, k) for k in STREAMS}
res = nd.gather_to_rank0((streams, ns.consensus_stream(g, 0, "metaData"), st, ns.consensus_verify(g)), dist)
if rank == 0:
    pickle.dump(res, open(out, "wb"))
dist.barrier()
g.close()
dist.destroy_process_group()
