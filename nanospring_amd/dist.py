"""Multi-GPU plumbing: reads shard by id over the ranks of one node, one process per GPU
(torch.distributed; "nccl" = RCCL on ROCm, "gloo" on CPU for the tests).  Every rank runs the whole hot
path on its shard with a global read-id base; no data-path collective.  Rank outputs are gathered as
additional "threads" of one archive: metaData is merged, stream sets are kept per rank/thread."""
import numpy as np


def shard_bounds(off, world):
    """Contiguous read-id ranges balanced by bases.  off: N+1 cumulative base offsets.  Returns [(lo, hi)] * world."""
    off = np.asarray(off, dtype=np.uint64)
    n = len(off) - 1
    total = int(off[-1] - off[0])
    cuts = [0]
    for r in range(1, world):
        target = int(off[0]) + total * r // world
        cuts.append(int(np.searchsorted(off, np.uint64(target), side="left")))
    cuts.append(n)
    for i in range(1, len(cuts)):
        cuts[i] = min(max(cuts[i], cuts[i - 1]), n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def take_shard(bases, off, lo, hi):
    off = np.asarray(off, dtype=np.uint64)
    b0, b1 = int(off[lo]), int(off[hi])
    return np.ascontiguousarray(bases[b0:b1]), (off[lo:hi + 1] - off[lo]).astype(np.uint64)


def parse_meta(md):
    d = {}
    for line in md.decode().splitlines():
        k, _, v = line.partition("=")
        d[k] = v
    return {"numReads": int(d["numReads"]), "numContigs": int(d["numContigs"]), "numThr": int(d["numThr"]),
            "numReadsInContig": [int(x) for x in d["numReadsInContig"].split(":") if x]}


def merge_meta(metas):
    """metaData of the shards -> metaData of the whole run (finishWriteConsensus layout, src/Consensus.cpp:370-386):
    threads of rank 0 first, then rank 1, ..."""
    ps = [parse_meta(m) for m in metas]
    s = "numReads=%d\nnumContigs=%d\nnumThr=%d\nnumReadsInContig=" % (
        sum(p["numReads"] for p in ps), sum(p["numContigs"] for p in ps), sum(p["numThr"] for p in ps))
    for p in ps:
        s += "".join("%d:" % c for c in p["numReadsInContig"])
    return (s + "\n").encode()


def gather_to_rank0(obj, dist):
    """all ranks -> list on rank 0 (None elsewhere); dist = torch.distributed (already initialised)."""
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def run_sharded(engine, bases, off, dist, n_threads_out=1):
    """engine(shard_bases, shard_off, id_base, n_threads_out) -> (list of stream dicts, metaData bytes, stats dict).
    Returns on rank 0: (streams of all ranks in rank order, merged metaData, [stats per rank]); None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(off, world)[rank]
    sb, so = take_shard(bases, off, lo, hi)
    streams, md, stats = engine(sb, so, lo, n_threads_out)
    got = gather_to_rank0((streams, md, stats), dist)
    if rank != 0:
        return None
    all_streams = [s for g in got for s in g[0]]
    return all_streams, merge_meta([g[1] for g in got]), [g[2] for g in got]


def gpu_engine(device=0, n_builders=1024, k=23, n=60, thr=6, salts=None, **kw):
    """The product engine for run_sharded (one nsgpu context on `device`)."""
    from . import filter as F

    def engine(sb, so, id_base, n_threads_out):
        g = F.NsGpu(k=k, n=n, overlap_sketch_thr=thr, device=device, **kw)
        g.load_reads((sb, so))
        F.check(g.lib, g.lib.nsgpu_set_read_id_base(g.ctx, id_base))
        g.sketch(F.mt19937_64_salts(n) if salts is None else salts, fetch=False)
        g.build_index()
        st = F.consensus_run(g, n_builders, n_threads_out)
        streams = [{s: F.consensus_stream(g, t, s) for s in F.STREAMS} for t in range(n_threads_out)]
        md = F.consensus_stream(g, 0, "metaData")
        st["bad"] = F.consensus_verify(g)
        g.close()
        return streams, md, st
    return engine
