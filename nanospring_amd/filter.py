"""Host-side mirror of the reference's operator interface for the MinHash half of
the path: `MinHashReadFilter` has the reference's public fields and methods
(include/ReadFilter.h:15-30, 33-111) and forwards to libnsgpu.so."""
import ctypes as C
import os

import numpy as np

from ._lib import NsGpuError, Params, Timing, check, load_library


def mt19937_64_salts(n, seed=12345):
    """First n outputs of std::mt19937_64(seed): what generateRandomNumbers
    (src/ReadFilter.cpp:49-63) yields once its random_device seed is pinned
    (uniform_int_distribution<unsigned long long> over the full range is the
    identity map in libstdc++)."""
    nn, mm = 312, 156
    mask = (1 << 64) - 1
    mt = [0] * nn
    mt[0] = seed & mask
    for i in range(1, nn):
        mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & mask
    out = []
    idx = nn
    um, lm = 0xFFFFFFFF80000000, 0x7FFFFFFF
    while len(out) < n:
        if idx >= nn:
            for i in range(nn):
                x = (mt[i] & um) | (mt[(i + 1) % nn] & lm)
                xa = x >> 1
                if x & 1:
                    xa ^= 0xB5026F5AA96619E9
                mt[i] = mt[(i + mm) % nn] ^ xa
            idx = 0
        x = mt[idx]
        idx += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        out.append(x & mask)
    return np.array(out, dtype=np.uint64)


def synth_reads(seed, genome_len, n_reads, mean_len=8000.0, p_sub=0.01, p_ins=0.01, p_del=0.01, first=0, genome="iid"):
    """SURVEY 8d synthetic reads.  Returns (bases: np.uint8 array of ASCII, off: np.uint64[n+1]).
    first > 0: reads [first, first + n_reads) of the same read set (a multi-GPU rank's id range).
    genome="repeats": the iid genome with planted interspersed duplications, tandem repeats, homopolymer and (AT)n / (ACGT)n runs."""
    lib = load_library()
    pb, po = C.c_void_p(), C.c_void_p()
    rc = lib.nsgpu_synth_reads_kind(seed, genome_len, first, n_reads, mean_len, p_sub, p_ins, p_del, {"iid": 0, "repeats": 1}[genome], C.byref(pb), C.byref(po))
    if rc != 0:
        raise NsGpuError(f"nsgpu_synth_reads failed ({rc})")
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n_reads + 1,)).copy()
    total = int(off[-1])
    bases = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), shape=(max(total, 1),))[:total].copy()
    lib.nsgpu_free(pb)
    lib.nsgpu_free(po)
    return bases, off


def _concat(strings):
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs and off[-1] else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(buf), off


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class NsGpu:
    """One nsgpu_ctx (one GPU, one stream)."""

    def __init__(self, k=23, n=60, overlap_sketch_thr=6, m_k=20, m_w=50, max_chain_iter=400, edge_threshold=4000000,
                 device=0, stream=None):
        self.lib = load_library()
        p = Params()
        self.lib.nsgpu_default_params(C.byref(p))
        p.k, p.n, p.overlap_sketch_thr = k, n, overlap_sketch_thr
        p.m_k, p.m_w, p.max_chain_iter, p.edge_threshold, p.device = m_k, m_w, max_chain_iter, edge_threshold, device
        self.params = p
        self.ctx = C.c_void_p()
        check(self.lib, self.lib.nsgpu_create(C.byref(p), C.byref(self.ctx)))
        if stream is not None:
            check(self.lib, self.lib.nsgpu_set_stream(self.ctx, C.c_void_p(stream)))
        self.k, self.n = k, n

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.nsgpu_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- read store ----
    def load_reads(self, reads):
        """reads: list of str/bytes, or (bases uint8 array, off uint64 array)."""
        if isinstance(reads, tuple):
            bases, off = reads
        else:
            bases, off = _concat(reads)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self._keep = (bases, off)
        check(self.lib, self.lib.nsgpu_load_reads_ascii(self.ctx, _ptr(bases), _ptr(off), len(off) - 1))

    def load_fastq(self, text):
        """Plain FASTQ text (bytes) -> reads, parsed on the GPU with the reference's getline rules (ReadData::loadFromFastqFile)."""
        buf = np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
        self._keep = buf
        n = C.c_uint32()
        check(self.lib, self.lib.nsgpu_load_fastq(self.ctx, _ptr(buf) if buf.size else None, buf.size, C.byref(n)))
        return n.value

    def load_fastq_chunks(self, pieces):
        """The same from consecutive pieces of the text, cut anywhere (nsgpu_load_fastq_begin / _chunk / _end)."""
        check(self.lib, self.lib.nsgpu_load_fastq_begin(self.ctx))
        for piece in pieces:
            buf = np.frombuffer(bytes(piece), dtype=np.uint8)
            check(self.lib, self.lib.nsgpu_load_fastq_chunk(self.ctx, _ptr(buf) if buf.size else None, buf.size))
        n = C.c_uint32()
        check(self.lib, self.lib.nsgpu_load_fastq_end(self.ctx, C.byref(n)))
        return n.value

    def load_fastq_file(self, path, gzip_flag=-1):
        """ReadData::loadFromFile(path, FASTQ, gzip_flag): the file itself, plain or gzip (-1: by its first two bytes)."""
        n = C.c_uint32()
        check(self.lib, self.lib.nsgpu_load_fastq_file(self.ctx, os.fsencode(path), int(gzip_flag), C.byref(n)))
        return n.value

    def load_reads_packed(self, packed, byte_off, lens):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        byte_off = np.ascontiguousarray(byte_off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        check(self.lib, self.lib.nsgpu_load_reads_packed(self.ctx, _ptr(packed), _ptr(byte_off), _ptr(lens), len(lens)))

    @property
    def num_reads(self):
        return int(self.lib.nsgpu_num_reads(self.ctx))

    @property
    def num_bases(self):
        return int(self.lib.nsgpu_num_bases(self.ctx))

    def get_read(self, r, maxlen=1 << 24):
        ln = C.c_uint32()
        buf = np.zeros(maxlen, dtype=np.uint8)
        check(self.lib, self.lib.nsgpu_get_read(self.ctx, r, _ptr(buf), C.byref(ln)))
        return buf[:ln.value].tobytes().decode()

    def get_read_packed(self, r, maxlen=1 << 24):
        ln = C.c_uint32()
        buf = np.zeros(maxlen // 4 + 4, dtype=np.uint8)
        check(self.lib, self.lib.nsgpu_get_read_packed(self.ctx, r, _ptr(buf), C.byref(ln)))
        return buf[:(ln.value + 3) // 4].copy(), ln.value

    # ---- sketch / index / filter ----
    def sketch(self, salts, fetch=True):
        salts = np.ascontiguousarray(salts, dtype=np.uint64)
        assert salts.size == self.n
        out = np.zeros((self.num_reads, self.n), dtype=np.uint64) if fetch else None
        check(self.lib, self.lib.nsgpu_sketch(self.ctx, _ptr(salts), _ptr(out) if fetch else None))
        return out

    def build_index(self):
        check(self.lib, self.lib.nsgpu_build_index(self.ctx))

    def index_export(self, j):
        N = self.num_reads
        keys = np.zeros(max(N, 1), dtype=np.uint64)
        start = np.zeros(N + 1, dtype=np.uint32)
        ids = np.zeros(max(N, 1), dtype=np.uint32)
        nk = C.c_uint32()
        check(self.lib, self.lib.nsgpu_index_export(self.ctx, j, _ptr(keys), _ptr(start), _ptr(ids), C.byref(nk)))
        u = nk.value
        return keys[:u].copy(), start[:u + 1].copy(), ids[:N].copy()

    def filter(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        ids, n = C.c_void_p(), C.c_size_t()
        check(self.lib, self.lib.nsgpu_filter(self.ctx, b, len(b), C.byref(ids), C.byref(n)))
        out = np.ctypeslib.as_array(C.cast(ids, C.POINTER(C.c_uint32)), shape=(max(n.value, 1),))[:n.value].copy()
        self.lib.nsgpu_free(ids)
        return out

    def filter_batch(self, strings):
        bases, off = _concat(strings)
        po, pi = C.c_void_p(), C.c_void_p()
        check(self.lib, self.lib.nsgpu_filter_batch(self.ctx, _ptr(bases), _ptr(off), len(strings), C.byref(po), C.byref(pi)))
        q = len(strings)
        o = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(q + 1,)).copy()
        tot = int(o[-1])
        ids = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint32)), shape=(max(tot, 1),))[:tot].copy()
        self.lib.nsgpu_free(po)
        self.lib.nsgpu_free(pi)
        return o, ids

    def filter_all_reads(self, fetch=True):
        tot = C.c_uint64()
        check(self.lib, self.lib.nsgpu_filter_all_reads(self.ctx, C.byref(tot)))
        if not fetch:
            return int(tot.value)
        off = np.zeros(2 * self.num_reads + 1, dtype=np.uint64)
        ids = np.zeros(max(int(tot.value), 1), dtype=np.uint32)
        check(self.lib, self.lib.nsgpu_filter_all_fetch(self.ctx, _ptr(off), _ptr(ids)))
        return off, ids[:int(tot.value)]

    def check_repetitive(self):
        f = np.zeros(max(self.num_reads, 1), dtype=np.uint8)
        check(self.lib, self.lib.nsgpu_check_repetitive(self.ctx, _ptr(f)))
        return f[:self.num_reads]

    def timing(self):
        t = Timing()
        check(self.lib, self.lib.nsgpu_get_timing(self.ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def sync(self):
        check(self.lib, self.lib.nsgpu_sync(self.ctx))


class MinHashReadFilter:
    """Same surface as the reference's MinHashReadFilter (include/ReadFilter.h:33-111):
    public fields k, n, overlapSketchThreshold; initialize(reads); getFilteredReads(s).
    The salts are explicit (`randNumbers`) instead of std::random_device."""

    def __init__(self, k=23, n=60, overlapSketchThreshold=6, randNumbers=None, device=0):
        self.k, self.n, self.overlapSketchThreshold = k, n, overlapSketchThreshold
        self.randNumbers = mt19937_64_salts(n) if randNumbers is None else np.asarray(randNumbers, dtype=np.uint64)
        self.gpu = NsGpu(k=k, n=n, overlap_sketch_thr=overlapSketchThreshold, device=device)
        self.sketches = None

    def initialize(self, reads, keep_sketches=False):
        self.gpu.load_reads(reads)
        self.sketches = self.gpu.sketch(self.randNumbers, fetch=keep_sketches)
        self.gpu.build_index()

    def getFilteredReads(self, s):
        return self.gpu.filter(s)


class KswParams(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("sc_ambi", C.c_int32), ("q", C.c_int32), ("e", C.c_int32),
                ("q2", C.c_int32), ("e2", C.c_int32)]


class KswEz(C.Structure):
    _fields_ = [("max", C.c_uint32), ("zdropped", C.c_int32), ("max_q", C.c_int32), ("max_t", C.c_int32), ("mqe", C.c_int32),
                ("mqe_t", C.c_int32), ("mte", C.c_int32), ("mte_q", C.c_int32), ("score", C.c_int32), ("n_cigar", C.c_int32),
                ("reach_end", C.c_int32)]


def bsc_aux_rate(n):
    """The sampling rate of the auxiliary indexes bsc_bwt_encode asks libsais for (libbsc/bwt/bwt.cpp:50-56): the largest power of two
    <= n / 8 (halved once more by the bit trick there), at least 1."""
    mod = n // 8
    for sh in (1, 2, 4, 8, 16):
        mod |= mod >> sh
    return (mod >> 1) + 1


def bwt_block(gpu, data, aux_rate=None):
    """nsgpu_bwt_block: the block sorter of the back end (libbsc's bsc_bwt_encode, libbsc/bwt/bwt.cpp:46-79) on one block.
    Returns (bwt bytes, primary index, indexes, gpu_ms, rounds); `indexes` are bsc_bwt_encode's (rank of suffix (t + 1) * rate, 0-based),
    rate = bsc_aux_rate(n) unless given (0 = none)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    n = int(buf.size)
    rate = bsc_aux_rate(n) if aux_rate is None else int(aux_rate)
    out = np.empty(max(n, 1), dtype=np.uint8)
    n_aux = (n - 1) // rate + 1 if rate and n else 0
    aux = np.zeros(max(n_aux, 1), dtype=np.int32)
    prim, na, ms, rounds = C.c_int32(), C.c_uint32(), C.c_double(), C.c_uint32()
    check(gpu.lib, gpu.lib.nsgpu_bwt_block(gpu.ctx, buf.ctypes.data if n else None, n, out.ctypes.data, C.byref(prim), rate, aux.ctypes.data, C.byref(na), C.byref(ms), C.byref(rounds)))
    assert int(na.value) == n_aux
    return out[:n].tobytes(), int(prim.value), [int(x) - 1 for x in aux[1:n_aux]], float(ms.value), int(rounds.value)


def ksw_extd2_batch(gpu, problems, a=2, b=4, sc_ambi=1, q=4, e=2, q2=24, e2=1):
    """problems: list of (query codes uint8, target codes uint8, w, zdrop, end_bonus, flag).
    Returns (list of ez tuples, list of CIGAR uint32 arrays) -- ksw_extd2_sse semantics."""
    n = len(problems)
    parts, qoff, toff, ql, tl = [], [], [], [], []
    pos = 0
    for (qq, tt, *_r) in problems:
        qq = np.ascontiguousarray(qq, dtype=np.uint8)
        tt = np.ascontiguousarray(tt, dtype=np.uint8)
        qoff.append(pos); pos += len(qq); toff.append(pos); pos += len(tt)
        ql.append(len(qq)); tl.append(len(tt))
        parts += [qq, tt]
    seqs = np.concatenate(parts) if parts else np.zeros(1, np.uint8)
    if seqs.size == 0:
        seqs = np.zeros(1, np.uint8)
    arr = lambda v, dt: np.ascontiguousarray(np.array(v, dtype=dt))
    qoff, toff = arr(qoff, np.uint64), arr(toff, np.uint64)
    ql, tl = arr(ql, np.int32), arr(tl, np.int32)
    w, zd, eb, fl = (arr([p[i] for p in problems], np.int32) for i in (2, 3, 4, 5))
    prm = KswParams(a, b, sc_ambi, q, e, q2, e2)
    ez = (KswEz * max(n, 1))()
    po, pc = C.c_void_p(), C.c_void_p()
    check(gpu.lib, gpu.lib.nsgpu_ksw_extd2_batch(gpu.ctx, n, _ptr(seqs), _ptr(qoff), _ptr(ql), _ptr(toff), _ptr(tl), _ptr(w), _ptr(zd),
                                                 _ptr(eb), _ptr(fl), C.byref(prm), ez, C.byref(po), C.byref(pc)))
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    tot = int(off[-1])
    cig = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint32)), shape=(max(tot, 1),))[:tot].copy()
    gpu.lib.nsgpu_free(po)
    gpu.lib.nsgpu_free(pc)
    ezs = [tuple(getattr(ez[i], f) for f, _ in KswEz._fields_) for i in range(n)]
    return ezs, [cig[int(off[i]):int(off[i + 1])] for i in range(n)]


class Aln(C.Structure):
    _fields_ = [("ok", C.c_int32), ("hits", C.c_int32), ("rel_pos", C.c_int64), ("begin_offset", C.c_int64), ("end_offset", C.c_int64)] + \
               [(n, C.c_int32) for n in ("rs", "re", "qs", "qe", "blen", "mlen", "n_ambi", "dp_max")] + \
               [("n_cigar", C.c_uint32), ("n_edits", C.c_uint32), ("cigar_off", C.c_uint64), ("edit_off", C.c_uint64)]


class AlignStats(C.Structure):
    _fields_ = [("pairs", C.c_uint64), ("dp_tasks", C.c_uint64), ("dp_rounds", C.c_uint64), ("dp_cells", C.c_double),
                ("index_ms", C.c_double), ("host_ms", C.c_double), ("dp_ms", C.c_double), ("dp_kernel_ms", C.c_double),
                ("dp_kernel_sum_ms", C.c_double), ("dp_alg_bytes", C.c_double), ("dp_launches", C.c_uint64), ("host_threads", C.c_uint32), ("reserved", C.c_uint32),
                ("seed_pairs_gpu", C.c_uint64), ("seed_pairs_host", C.c_uint64),
                ("plan_pairs_dev", C.c_uint64), ("plan_pairs_host", C.c_uint64), ("plan_hits", C.c_uint64), ("plan_misses", C.c_uint64), ("plan_extra", C.c_uint64)]


EDIT_DT = np.dtype([("type", np.uint8), ("base", np.uint8), ("reserved", np.uint16), ("num", np.uint32)])


def mm_sketch_batch(gpu, seqs, w=50, k=20):
    """mm_sketch (minimap2/sketch.c:77-143, rid 0) of a batch of sequences on the GPU.  seqs: list of str or a
    (bases, off) tuple.  Returns a list of (n_i, 2) uint64 arrays (x, y) in the reference's output order."""
    sb, so = seqs if isinstance(seqs, tuple) else _concat(seqs)
    n = len(so) - 1
    px, po = C.c_void_p(), C.c_void_p()
    check(gpu.lib, gpu.lib.nsgpu_mm_sketch_batch(gpu.ctx, _ptr(sb), _ptr(so), n, w, k, C.byref(px), C.byref(po)))
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    tot = int(off[n])
    xy = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint64)), shape=(max(tot, 1) * 2,))[:tot * 2].copy().reshape(-1, 2)
    gpu.lib.nsgpu_free(px)
    gpu.lib.nsgpu_free(po)
    return [xy[int(off[i]):int(off[i + 1])] for i in range(n)]


def align_batch(gpu, refs, queries, pair_ref):
    """ConsensusGraph::alignRead for a batch: refs/queries are lists of str (or (bases, off) tuples),
    pair_ref[i] = index of the reference of query i.  Returns a list of dicts with alignRead's outputs
    (ok, rel_pos, begin_offset, end_offset, edits) plus reg[0]'s coordinates and CIGAR."""
    rb, ro = refs if isinstance(refs, tuple) else _concat(refs)
    qb, qo = queries if isinstance(queries, tuple) else _concat(queries)
    pr = np.ascontiguousarray(pair_ref, dtype=np.uint32)
    n = len(qo) - 1
    assert len(pr) == n
    out = (Aln * max(n, 1))()
    pc, pe = C.c_void_p(), C.c_void_p()
    check(gpu.lib, gpu.lib.nsgpu_align_batch(gpu.ctx, _ptr(rb), _ptr(ro), len(ro) - 1, _ptr(qb), _ptr(qo), _ptr(pr), n, out,
                                             C.byref(pc), C.byref(pe)))
    nc = sum(out[i].n_cigar for i in range(n))
    ne = sum(out[i].n_edits for i in range(n))
    cig = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint32)), shape=(max(nc, 1),))[:nc].copy()
    raw = np.ctypeslib.as_array(C.cast(pe, C.POINTER(C.c_uint8)), shape=(max(ne, 1) * 8,))[:ne * 8].copy()
    ed = raw.view(EDIT_DT)
    gpu.lib.nsgpu_free(pc)
    gpu.lib.nsgpu_free(pe)
    res = []
    for i in range(n):
        a = out[i]
        d = {f: getattr(a, f) for f, _ in Aln._fields_}
        d["cigar"] = cig[a.cigar_off:a.cigar_off + a.n_cigar]
        d["edits"] = ed[a.edit_off:a.edit_off + a.n_edits]
        res.append(d)
    return res


def align_stats(gpu, reset=False):
    s = AlignStats()
    check(gpu.lib, gpu.lib.nsgpu_get_align_stats(gpu.ctx, C.byref(s)))
    if reset:
        check(gpu.lib, gpu.lib.nsgpu_reset_align_stats(gpu.ctx))
    return {k: getattr(s, k) for k, _ in AlignStats._fields_}


class ConsensusStats(C.Structure):
    _fields_ = [("n_builders", C.c_uint32), ("reserved", C.c_uint32)] + \
               [(n, C.c_uint64) for n in ("n_rounds", "n_filter_rounds", "n_align_rounds", "n_windows", "n_contigs", "n_lone", "count_minhash",
                                          "count_minhash_not_in_graph", "count_aligner", "n_align_calls")] + \
               [(n, C.c_double) for n in ("total_ms", "graph_ms", "filter_ms", "index_ms", "align_ms", "graph_cpu_ms", "graph_max_ms", "graph_crit_ms", "write_cpu_ms")]


STREAMS = ["genome", "lone", "id", "pos", "type", "base", "complement"]


def set_schedule(gpu, groups=4, seed_bucket_depth=0, seed_rings=1, seed_tail_rings=None):
    """nsgpu_set_schedule(2): pipeline groups (1, 2, 4) and the conflict-aware seed rule (bucket depth 0 = the reference's getRead rule;
    seed_tail_rings: the smaller radius while more than half of all builders wait for a seed)."""
    check(gpu.lib, gpu.lib.nsgpu_set_schedule2(gpu.ctx, groups, seed_bucket_depth, seed_rings, seed_rings if seed_tail_rings is None else seed_tail_rings))


def get_schedule(gpu):
    """(groups, bucket depth, rings, tail rings, builders) of the context: what the last run used when it derived them itself."""
    v = [C.c_uint32() for _ in range(5)]
    check(gpu.lib, gpu.lib.nsgpu_get_schedule2(gpu.ctx, *[C.byref(x) for x in v]))
    return tuple(int(x.value) for x in v)


def set_defer(gpu, anchors, slots):
    """alignments with more than `anchors` anchors take `slots` more slots (nsgpu_set_defer; 0 slots = off)"""
    check(gpu.lib, gpu.lib.nsgpu_set_defer(gpu.ctx, int(anchors), int(slots)))


def get_defer(gpu):
    """(anchors, slots, alignments deferred in the last contig stage)"""
    a, s, n = C.c_uint32(), C.c_uint32(), C.c_uint64()
    check(gpu.lib, gpu.lib.nsgpu_get_defer(gpu.ctx, C.byref(a), C.byref(s), C.byref(n)))
    return int(a.value), int(s.value), int(n.value)


GRAPH_AUTO, GRAPH_HOST, GRAPH_DEVICE, GRAPH_CHECK = 0, 1, 2, 0x100


class GraphStats(C.Structure):
    _fields_ = [("placement", C.c_uint32), ("checked", C.c_uint32)] + \
               [(n, C.c_uint64) for n in ("n_updates", "n_launches", "n_array_growths", "n_long_reports", "n_sequential_updates", "n_full_walks", "n_split_calls")] + \
               [("kernel_ms", C.c_double * 8), ("report_ms", C.c_double), ("host_wait_first_ms", C.c_double), ("host_wait_second_ms", C.c_double), ("by_duration", C.c_uint64 * 8)] + \
               [(n, C.c_double) for n in ("gb_copied_back", "hbm_peak_gb", "hbm_mapped_gb", "pinned_peak_gb", "pinned_mapped_gb")]


def set_graph(gpu, mode):
    """where the contigs' consensus graphs live: GRAPH_AUTO / GRAPH_HOST / GRAPH_DEVICE, optionally | GRAPH_CHECK (nsgpu_set_graph)"""
    check(gpu.lib, gpu.lib.nsgpu_set_graph(gpu.ctx, int(mode)))


def graph_stats(gpu):
    """what the last contig stage used and what its graph kernels did (nsgpu_get_graph_stats)"""
    s = GraphStats()
    check(gpu.lib, gpu.lib.nsgpu_get_graph_stats(gpu.ctx, C.byref(s)))
    d = {k: getattr(s, k) for k, _ in GraphStats._fields_}
    d["kernel_ms"] = [float(x) for x in s.kernel_ms]
    d["by_duration"] = [int(x) for x in s.by_duration]
    d["placement"] = {1: "host", 2: "device"}.get(int(s.placement), "none")
    return d


def consensus_run(gpu, n_builders=256, n_threads_out=1, schedule=None, defer=None):
    """schedule: (groups, depth, rings[, tail rings]), or "auto" (nsgpu_set_schedule_auto; n_builders = 0 lets the library choose the count too);
    defer: (anchors, slots) for nsgpu_set_defer"""
    if defer is not None:
        set_defer(gpu, *defer)
    if isinstance(schedule, str):
        assert schedule == "auto"
        check(gpu.lib, gpu.lib.nsgpu_set_schedule_auto(gpu.ctx))
    elif schedule is not None:
        set_schedule(gpu, *schedule)
    s = ConsensusStats()
    check(gpu.lib, gpu.lib.nsgpu_consensus_run(gpu.ctx, n_builders, n_threads_out, C.byref(s)))
    return {k: getattr(s, k) for k, _ in ConsensusStats._fields_}


def consensus_stream(gpu, thread, which):
    idx = 7 if which == "metaData" else STREAMS.index(which)
    p, n = C.c_void_p(), C.c_size_t()
    check(gpu.lib, gpu.lib.nsgpu_consensus_stream(gpu.ctx, thread, idx, C.byref(p), C.byref(n)))
    b = C.string_at(p, n.value)
    gpu.lib.nsgpu_free(p)
    return b


def consensus_verify(gpu):
    bad = C.c_uint64()
    check(gpu.lib, gpu.lib.nsgpu_consensus_verify(gpu.ctx, C.byref(bad)))
    return int(bad.value)


def consensus_write(gpu, temp_dir, temp_file_name="Stream"):
    check(gpu.lib, gpu.lib.nsgpu_consensus_write(gpu.ctx, temp_dir.encode(), temp_file_name.encode()))
