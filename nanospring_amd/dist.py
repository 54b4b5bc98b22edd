"""Multi-GPU plumbing: reads shard by id over the ranks of one node, one process per GPU
(torch.distributed; "nccl" = RCCL on ROCm, "gloo" on CPU for the tests).  Every rank runs the whole hot
path on its shard with a global read-id base; no data-path collective.  Rank outputs are gathered as
additional "threads" of one archive: metaData is merged, stream sets are kept per rank/thread."""
import os

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


# ---------------------------------------------------------------------------
# Exchange mode: every rank ends up with ALL reads and the WHOLE bucket index, so contigs
# recruit reads across shards; claims are resolved on a replicated table from all-gathered request
# lists in global builder order, which makes the result independent of the number of ranks.
#   bulk data   : all-gather of the read shards (at load time) and, every step, of the sketch rows each
#                 rank computed for its own id range  (RCCL when the backend is "nccl")
#   per round   : two small all-gathers of (builder, read) request lists
# ---------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _dev(dist):
    torch = _torch()
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def all_gather_bytes(arr, dist):
    """arr: 1-D uint8 numpy array (different length per rank) -> list of numpy arrays in rank order."""
    torch = _torch()
    dev = _dev(dist)
    world = dist.get_world_size()
    n = torch.tensor([arr.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if arr.size:
        buf[:arr.size] = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    out = torch.empty(mx * world, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy()
    return [out[r * mx:r * mx + sizes[r]].copy() for r in range(world)]


def all_gather_u32_lists(a, b, dist):
    """Two equally long uint32 lists per rank -> the concatenation over all ranks (rank order)."""
    world = dist.get_world_size()
    if world == 1:
        return np.asarray(a, dtype=np.uint32), np.asarray(b, dtype=np.uint32)
    packed = np.concatenate([np.asarray(a, dtype=np.uint32), np.asarray(b, dtype=np.uint32)]).view(np.uint8)
    parts = all_gather_bytes(packed, dist)
    aa, bb = [], []
    for p in parts:
        v = p.view(np.uint32)
        h = v.size // 2
        aa.append(v[:h]); bb.append(v[h:])
    return np.concatenate(aa), np.concatenate(bb)


class U32ListGatherer:
    """all_gather_u32_lists for `n_lists` (a, b) list pairs at once with a known bound on the list length: ONE collective
    per call (no size exchange) on buffers allocated once -- the claim and seed lists of the contig engine are exchanged
    together, once per pipeline slot."""

    def __init__(self, cap, dist, n_lists=1):
        torch = _torch()
        self.dist, self.cap, self.world, self.n_lists = dist, int(cap), dist.get_world_size(), int(n_lists)
        dev = _dev(dist)
        self.words = self.n_lists * (1 + 2 * self.cap)
        self.mine = torch.zeros(self.words, dtype=torch.int32, device=dev)
        self.all = torch.zeros(self.words * self.world, dtype=torch.int32, device=dev)
        self.stage = np.zeros(self.words, dtype=np.uint32)

    def __call__(self, *lists):
        """lists = a0, b0, a1, b1, ...; returns the concatenation over ranks of every list, in the same order."""
        assert len(lists) == 2 * self.n_lists
        torch = _torch()
        st = self.stage
        blk = 1 + 2 * self.cap
        for i in range(self.n_lists):
            a = np.asarray(lists[2 * i], dtype=np.uint32)
            b = np.asarray(lists[2 * i + 1], dtype=np.uint32)
            n = a.size                               # (a world of one still goes through the collective: the RCCL test relies on it)
            assert n == b.size and n <= self.cap, (n, self.cap)
            o = i * blk
            st[o] = n
            st[o + 1:o + 1 + n] = a
            st[o + 1 + self.cap:o + 1 + self.cap + n] = b
        self.mine.copy_(torch.from_numpy(st.view(np.int32)))
        self.dist.all_gather_into_tensor(self.all, self.mine)
        v = self.all.cpu().numpy().view(np.uint32).reshape(self.world, self.words)
        out = []
        for i in range(self.n_lists):
            o = i * blk
            out.append(np.concatenate([v[r, o + 1:o + 1 + int(v[r, o])] for r in range(self.world)]))
            out.append(np.concatenate([v[r, o + 1 + self.cap:o + 1 + self.cap + int(v[r, o])] for r in range(self.world)]))
        return tuple(out)


def replicate_reads(bases, off, dist):
    """Each rank passes its own shard (reads in global id order across ranks).  Returns (all_bases, all_off, lo, hi)
    with [lo, hi) = this rank's id range in the replicated set."""
    off = np.asarray(off, dtype=np.uint64)
    lens = np.diff(off).astype(np.uint32)
    shards = all_gather_bytes(np.ascontiguousarray(bases[int(off[0]):int(off[-1])]), dist)
    lens_all = [p.view(np.uint32) for p in all_gather_bytes(lens.view(np.uint8), dist)]
    rank = dist.get_rank()
    lo = int(sum(len(x) for x in lens_all[:rank]))
    hi = lo + len(lens_all[rank])
    all_lens = np.concatenate(lens_all) if lens_all else np.zeros(0, np.uint32)
    all_off = np.zeros(all_lens.size + 1, dtype=np.uint64)
    all_off[1:] = np.cumsum(all_lens, dtype=np.uint64)
    return np.concatenate(shards), all_off, lo, hi


def exchange_sketch_rows(gpu, salts, lo, hi, dist):
    """Sketch the own id range, all-gather the rows, import the other ranks' rows (device buffers with nccl)."""
    import ctypes as C
    from . import filter as F
    torch = _torch()
    lib, ctx = gpu.lib, gpu.ctx
    n = gpu.n
    salts = np.ascontiguousarray(salts, dtype=np.uint64)
    F.check(lib, lib.nsgpu_sketch_range(ctx, salts.ctypes.data_as(C.c_void_p), lo, hi))
    world = dist.get_world_size()
    if world > 1 or os.environ.get("NSGPU_TEST_FORCE_EXCHANGE"):     # the flag lets a world of one drive the device get / all-gather / set path
        dev = _dev(dist)
        on_dev = int(dev.type == "cuda")
        rows = torch.tensor([hi - lo], dtype=torch.int64, device=dev)
        cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(cnts, rows)
        cnts = [int(x.item()) for x in cnts]
        mx = max(max(cnts), 1)
        # The library copies the rows on ITS stream; torch fills / gathers on torch's current stream.  Order the two explicitly:
        # the buffer is allocated uninitialised, the library's copy is host-synchronised before it returns (rows_get), only the
        # padding tail is zeroed by torch afterwards, and torch's stream is drained before the library reads the gathered rows.
        mine = torch.empty(mx * n, dtype=torch.int64, device=dev)
        if on_dev:
            torch.cuda.current_stream().synchronize()
        if hi > lo:
            F.check(lib, lib.nsgpu_sketch_rows_get(ctx, lo, hi, C.c_void_p(mine.data_ptr()), on_dev))
        if (hi - lo) * n < mine.numel():
            mine[(hi - lo) * n:].zero_()
        allrows = torch.empty(mx * n * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allrows, mine)
        if on_dev:
            torch.cuda.synchronize()
        start = 0
        me = dist.get_rank()
        for r in range(world):
            if cnts[r] and (r != me or os.environ.get("NSGPU_TEST_FORCE_EXCHANGE")):
                ptr = allrows.data_ptr() + r * mx * n * 8
                F.check(lib, lib.nsgpu_sketch_rows_set(ctx, start, start + cnts[r], C.c_void_p(ptr), on_dev))
            start += cnts[r]
    F.check(lib, lib.nsgpu_sketch_mark_complete(ctx))


def consensus_exchange(gpu, n_builders_total, dist, n_threads_out=1):
    """The contig stage over all ranks (nsgpu_cons_* phases with an all-gather between requests and resolve)."""
    import ctypes as C
    from . import filter as F
    lib, ctx = gpu.lib, gpu.ctx
    rank, world = dist.get_rank(), dist.get_world_size()
    F.check(lib, lib.nsgpu_cons_begin(ctx, n_builders_total, rank, world))

    def take(fn, group):
        pa, pb, n = C.c_void_p(), C.c_void_p(), C.c_uint32()
        F.check(lib, fn(ctx, group, C.byref(pa), C.byref(pb), C.byref(n)))
        a = np.ctypeslib.as_array(C.cast(pa, C.POINTER(C.c_uint32)), shape=(max(n.value, 1),))[:n.value].copy()
        b = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint32)), shape=(max(n.value, 1),))[:n.value].copy()
        lib.nsgpu_free(pa); lib.nsgpu_free(pb)
        return a, b

    def ptr(a):
        return a.ctypes.data_as(C.c_void_p) if a.size else None

    # a rank never has more requests than local builders
    gather = U32ListGatherer((n_builders_total + world - 1) // world + 1, dist, n_lists=2)

    # the slot schedule of include/nsgpu.h (nsgpu_consensus_run runs the same one with world = 1): one collective per slot
    n_coll = 0
    slot = 0
    g_ = C.c_uint32()
    F.check(lib, lib.nsgpu_get_schedule(ctx, C.byref(g_), None, None))      # 4 (default) or 2; the one-group schedule is the C++ drivers'
    n_groups = int(g_.value)
    while True:
        h, b = slot % n_groups, (slot + 1) % n_groups
        F.check(lib, lib.nsgpu_cons_slot(ctx, slot))
        ca, cb = take(lib.nsgpu_cons_claim_requests, b)
        sa, sb = take(lib.nsgpu_cons_seed_requests, h)
        ca, cb, sa, sb = (np.ascontiguousarray(x) for x in gather(ca, cb, sa, sb))
        n_coll += 1
        done = C.c_uint32()
        F.check(lib, lib.nsgpu_cons_claim_resolve(ctx, ptr(ca), ptr(cb), ca.size, C.byref(done)))
        started = C.c_uint32()
        F.check(lib, lib.nsgpu_cons_seed_resolve(ctx, ptr(sa), ptr(sb), sa.size, C.byref(started), C.byref(done)))
        if started.value:
            F.check(lib, lib.nsgpu_cons_advance(ctx, 1, h))
        if done.value:
            break
        slot += 1
    st = F.ConsensusStats()
    F.check(lib, lib.nsgpu_cons_finish(ctx, n_threads_out, C.byref(st)))
    out = {k: getattr(st, k) for k, _ in F.ConsensusStats._fields_}
    out["n_collectives"] = n_coll
    return out


# ---------------------------------------------------------------------------
# The C++ multi-GPU path (csrc/dist.hip, nsgpu_dist_*): Python only creates the communicator and calls three entry points.
#   backend "nccl": the library's own RCCL communicator (rank 0's unique id travels through torch.distributed's store)
#   backend "gloo": host callbacks into torch.distributed (tests: several ranks on one GPU, or no RCCL at all)
# ---------------------------------------------------------------------------
REPLICATE, ALLTOALL = 0, 1


class _Callbacks:
    """nsgpu_comm_callbacks over torch.distributed (host tensors)."""

    def __init__(self, dist):
        import ctypes as C
        torch = _torch()
        self.dist, self.world = dist, dist.get_world_size()

        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
        A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))

        def view(ptr, nbytes):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),))[:int(nbytes)]

        def all_gather(_user, send, recv, nbytes):
            try:
                src = torch.from_numpy(view(send, nbytes).copy())
                out = torch.empty(int(nbytes) * self.world, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, src)
                view(recv, int(nbytes) * self.world)[:] = out.numpy()
                return 0
            except Exception as e:            # never let an exception cross the C boundary
                print("nsgpu comm callback all_gather:", e, flush=True)
                return 1

        def all_to_all(_user, send, sb, recv, rb):
            try:
                sbs = [int(sb[p]) for p in range(self.world)]
                rbs = [int(rb[p]) for p in range(self.world)]
                sv = view(send, sum(sbs))
                ins, o = [], 0
                for n in sbs:
                    ins.append(torch.from_numpy(sv[o:o + n].copy()))
                    o += n
                outs = [torch.empty(n, dtype=torch.uint8) for n in rbs]
                # gloo has no ragged all_to_all: one gather per destination (test transport, not the product's)
                for dst in range(self.world):
                    got = [None] * self.world if dist.get_rank() == dst else None
                    dist.gather_object(ins[dst].numpy().tobytes(), got, dst=dst)
                    if dist.get_rank() == dst:
                        for p in range(self.world):
                            assert len(got[p]) == rbs[p], (p, len(got[p]), rbs[p])
                            outs[p] = torch.from_numpy(np.frombuffer(got[p], dtype=np.uint8).copy()) if rbs[p] else outs[p]
                rv = view(recv, sum(rbs))
                o = 0
                for p, n in enumerate(rbs):
                    if n:
                        rv[o:o + n] = outs[p].numpy()
                    o += n
                return 0
            except Exception as e:
                print("nsgpu comm callback all_to_all:", e, flush=True)
                return 1

        class CB(C.Structure):
            _fields_ = [("user", C.c_void_p), ("all_gather", AG), ("all_to_all", A2A)]
        self._keep = (AG(all_gather), A2A(all_to_all))
        self.struct = CB(None, self._keep[0], self._keep[1])


class DistJob:
    """One rank of a multi-GPU run through the C++ entry points (include/nsgpu.h, multi-GPU section)."""

    def __init__(self, gpu, dist, backend=None):
        import ctypes as C
        from . import filter as F
        self.gpu, self.dist, self.F = gpu, dist, F
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        lib = gpu.lib
        self.comm = C.c_void_p()
        backend = backend or dist.get_backend()
        if backend == "nccl":
            uid = (C.c_uint8 * 128)()
            if self.rank == 0:
                F.check(lib, lib.nsgpu_comm_unique_id(uid))
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
            F.check(lib, lib.nsgpu_comm_init_rccl(gpu.ctx, uid, self.rank, self.world, C.byref(self.comm)))
            self._cb = None
        else:
            self._cb = _Callbacks(dist)
            F.check(lib, lib.nsgpu_comm_init_callbacks(gpu.ctx, C.byref(self._cb.struct), self.rank, self.world, C.byref(self.comm)))

    def load_reads(self, bases, off):
        """this rank's shard (reads in global id order across ranks) -> every rank holds all reads; returns (lo, hi)"""
        import ctypes as C
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lo, hi = C.c_uint32(), C.c_uint32()
        self.F.check(self.gpu.lib, self.gpu.lib.nsgpu_dist_load_reads(self.gpu.ctx, self.comm, bases.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                                                     C.c_uint32(len(off) - 1), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def sketch_index(self, salts, mode=REPLICATE):
        import ctypes as C
        salts = np.ascontiguousarray(salts, dtype=np.uint64)
        self.F.check(self.gpu.lib, self.gpu.lib.nsgpu_dist_sketch_index(self.gpu.ctx, self.comm, salts.ctypes.data_as(C.c_void_p), int(mode)))

    def consensus_run(self, n_builders_total, n_threads_out=1):
        import ctypes as C
        st = self.F.ConsensusStats()
        self.F.check(self.gpu.lib, self.gpu.lib.nsgpu_dist_consensus_run(self.gpu.ctx, self.comm, n_builders_total, n_threads_out, C.byref(st)))
        out = {k: getattr(st, k) for k, _ in self.F.ConsensusStats._fields_}
        out["n_collectives"] = out.get("reserved", 0)
        return out

    def comm_stats(self):
        """bytes this rank received in all-gathers / all-to-alls so far, and the host memory of its copy of all reads"""
        import ctypes as C
        a, b, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.F.check(self.gpu.lib, self.gpu.lib.nsgpu_comm_stats(self.comm, C.byref(a), C.byref(b), C.byref(h)))
        return {"all_gather_bytes": int(a.value), "all_to_all_bytes": int(b.value), "host_bytes_of_the_read_copy": int(h.value)}

    def close(self):
        if self.comm:
            self.gpu.lib.nsgpu_comm_destroy(self.comm)
            self.comm = None
