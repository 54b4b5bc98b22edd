"""ctypes wrappers around the checkers: oracle/libnsoracle.so (our CPU
restatement) and, when present, oracle/_ref (the reference's own objects).
Test infrastructure only -- never imported by nanospring_amd."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libnsoracle.so")
NSREF = os.path.join(ORACLE_DIR, "_ref", "nsref")
MM2REF = os.path.join(ORACLE_DIR, "_ref", "libmm2ref.so")


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "libnsoracle.so"], check=True, capture_output=True)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def concat(strings):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if off[-1] else np.zeros(1, dtype=np.uint8)
    return buf, off


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(
                os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in os.listdir(ORACLE_DIR) if f.endswith("_oracle.c")):
            build_oracle()
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.oracle_filter_sketch.restype = C.c_uint64
        L.oracle_filter_string.restype = C.c_uint64
        L.oracle_check_repetitive.restype = C.c_int
        L.oracle_num_threads.restype = C.c_int

    def pack2bit(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        out = np.zeros((len(b) + 3) // 4 + 1, dtype=np.uint8)
        self.lib.oracle_pack2bit(b, C.c_uint64(len(b)), _p(out))
        return out[:(len(b) + 3) // 4]

    def unpack2bit(self, packed, n):
        out = np.zeros(n + 1, dtype=np.uint8)
        packed = np.ascontiguousarray(packed)
        self.lib.oracle_unpack2bit(_p(packed), C.c_uint64(n), _p(out))
        return out[:n].tobytes().decode()

    def revcomp(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        out = np.zeros(len(b) + 1, dtype=np.uint8)
        self.lib.oracle_revcomp(b, C.c_uint64(len(b)), _p(out))
        return out[:len(b)].tobytes().decode()

    def sketch(self, s, k, n, salts):
        b = s.encode() if isinstance(s, str) else bytes(s)
        out = np.zeros(n, dtype=np.uint64)
        salts = np.ascontiguousarray(salts, dtype=np.uint64)
        self.lib.oracle_sketch(b, C.c_uint64(len(b)), C.c_uint32(k), C.c_uint32(n), _p(salts), _p(out))
        return out

    def sketch_reads(self, bases, off, k, n, salts):
        N = len(off) - 1
        out = np.zeros((N, n), dtype=np.uint64)
        salts = np.ascontiguousarray(salts, dtype=np.uint64)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self.lib.oracle_sketch_reads(_p(bases), _p(off), C.c_uint32(N), C.c_uint32(k), C.c_uint32(n), _p(salts), _p(out))
        return out

    def index_build(self, sketches):
        N, n = sketches.shape
        sk = np.ascontiguousarray(sketches, dtype=np.uint64)
        keys = np.zeros((n, max(N, 1)), dtype=np.uint64)
        start = np.zeros((n, N + 1), dtype=np.uint32)
        ids = np.zeros((n, max(N, 1)), dtype=np.uint32)
        nkeys = np.zeros(n, dtype=np.uint32)
        self.lib.oracle_index_build(_p(sk), C.c_uint32(N), C.c_uint32(n), _p(keys), _p(start), _p(ids), _p(nkeys))
        return {"keys": keys, "start": start, "ids": ids, "nkeys": nkeys, "N": N, "n": n}

    def filter_sketch(self, q, idx, thr):
        N, n = idx["N"], idx["n"]
        q = np.ascontiguousarray(q, dtype=np.uint64)
        out = np.zeros(max(N, 1), dtype=np.uint32)
        m = C.c_uint64()
        c = self.lib.oracle_filter_sketch(_p(q), C.c_uint32(N), C.c_uint32(n), C.c_uint32(thr), _p(idx["keys"]), _p(idx["start"]),
                                          _p(idx["ids"]), _p(idx["nkeys"]), _p(out), C.c_uint64(N), C.byref(m))
        return out[:c].copy(), int(m.value)

    def filter_string(self, s, k, salts, idx, thr):
        return self.filter_sketch(self.sketch(s, k, idx["n"], salts), idx, thr)

    def check_repetitive(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        return int(self.lib.oracle_check_repetitive(b, C.c_uint64(len(b))))

    def fastq_index(self, text):
        """(start, len) of every read's base line under the reference's getline loop (oracle_fastq_index)."""
        b = np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        cap = b.size // 2 + 4
        st = np.zeros(cap, dtype=np.uint64)
        ln = np.zeros(cap, dtype=np.uint32)
        f = self.lib.oracle_fastq_index
        f.restype = C.c_uint64
        n = int(f(_p(b) if b.size else None, C.c_uint64(b.size), _p(st), _p(ln), C.c_uint64(cap)))
        return st[:n].copy(), ln[:n].copy()

    def num_threads(self):
        return int(self.lib.oracle_num_threads())


def have_nsref():
    return os.path.exists(NSREF)


def run_nsref(reads, queries, k, n, thr, salts):
    """Runs the reference's own MinHashReadFilter objects (oracle/_ref/nsref)."""
    rb, roff = concat(reads)
    qb, qoff = concat(queries)
    N, Q = len(reads), len(queries)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([k, n, thr, N, Q], dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(salts, dtype=np.uint64).tobytes())
            f.write(roff.tobytes())
            f.write(rb[:int(roff[-1])].tobytes())
            f.write(qoff.tobytes())
            f.write(qb[:int(qoff[-1])].tobytes())
        subprocess.run([NSREF, fin, fout, td], check=True, capture_output=True)
        raw = open(fout, "rb").read()
    p = 0

    def take(dtype, cnt):
        nonlocal p
        a = np.frombuffer(raw, dtype=dtype, count=cnt, offset=p).copy()
        p += a.nbytes
        return a
    res = {}
    res["sketches"] = take(np.uint64, N * n).reshape(N, n)
    tot = int(take(np.uint64, 1)[0])
    res["packed"] = take(np.uint8, tot)
    res["qsketch"] = take(np.uint64, Q * n).reshape(Q, n)
    fl = []
    for _ in range(Q):
        c = int(take(np.uint64, 1)[0])
        fl.append(take(np.uint32, c))
    res["filter"] = fl
    tabs = []
    for j in range(n):
        row = []
        for r in range(N):
            c = int(take(np.uint64, 1)[0])
            row.append(take(np.uint32, c))
        tabs.append(row)
    res["tables"] = tabs
    res["unpack_ok"] = take(np.uint8, N)
    return res


# ---------------------------------------------------------------------------
# ksw2 (dual-affine banded DP): oracle restatement and the reference's own SSE kernel
# ---------------------------------------------------------------------------
class EzT(C.Structure):
    _fields_ = [("max", C.c_uint32), ("zdropped", C.c_int32), ("max_q", C.c_int32), ("max_t", C.c_int32), ("mqe", C.c_int32),
                ("mqe_t", C.c_int32), ("mte", C.c_int32), ("mte_q", C.c_int32), ("score", C.c_int32), ("n_cigar", C.c_int32),
                ("reach_end", C.c_int32)]

    def astuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


KSW_DEFAULT = dict(a=2, b=4, sc_ambi=1, q=4, e=2, q2=24, e2=1)   # minimap2/options.c:43-44


def oracle_ksw(orc, query, target, w, zdrop, end_bonus, flag, prm=KSW_DEFAULT):
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(target, dtype=np.uint8)
    ez = EzT()
    cap = len(q) + len(t) + 4
    cig = np.zeros(cap, dtype=np.uint32)
    f = orc.lib.oracle_ksw_extd2
    f.restype = C.c_int
    n = f(C.c_int(len(q)), _p(q), C.c_int(len(t)), _p(t), C.c_int8(prm["a"]), C.c_int8(-prm["b"]), C.c_int8(-prm["sc_ambi"]),
          C.c_int8(prm["q"]), C.c_int8(prm["e"]), C.c_int8(prm["q2"]), C.c_int8(prm["e2"]), C.c_int(w), C.c_int(zdrop),
          C.c_int(end_bonus), C.c_int(flag), C.byref(ez), _p(cig), C.c_int(cap))
    return ez.astuple(), cig[:max(n, 0)].copy()


_MM2 = None


def mm2ref():
    global _MM2
    if _MM2 is None:
        if not os.path.exists(MM2REF):
            return None
        _MM2 = C.CDLL(MM2REF)
    return _MM2


def ref_ksw(query, target, w, zdrop, end_bonus, flag, prm=KSW_DEFAULT):
    lib = mm2ref()
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(target, dtype=np.uint8)
    ez = EzT()
    cap = len(q) + len(t) + 4
    cig = np.zeros(cap, dtype=np.uint32)
    lib.ref_ksw_extd2(C.c_int(len(q)), _p(q), C.c_int(len(t)), _p(t), C.c_int(prm["a"]), C.c_int(prm["b"]), C.c_int(prm["sc_ambi"]),
                      C.c_int(prm["q"]), C.c_int(prm["e"]), C.c_int(prm["q2"]), C.c_int(prm["e2"]), C.c_int(w), C.c_int(zdrop),
                      C.c_int(end_bonus), C.c_int(flag), C.byref(ez), _p(cig), C.c_int(cap))
    return ez.astuple(), cig[:max(ez.n_cigar, 0)].copy()


def ksw_random_problem(rng, qlen, tlen, err=0.06, n_frac=0.0, diverge_at=None):
    """target = random; query = noisy copy (sub/ins/del), cut/padded to qlen; codes 0..3 (4 = N)."""
    tgt = rng.randint(0, 4, size=tlen).astype(np.uint8)
    out = []
    i = 0
    while len(out) < qlen:
        if i >= tlen or (diverge_at is not None and len(out) >= diverge_at):
            out.append(rng.randint(0, 4))
            continue
        u = rng.random_sample()
        if u < err / 3:
            out.append((int(tgt[i]) + 1 + rng.randint(3)) % 4); i += 1
        elif u < 2 * err / 3:
            out.append(rng.randint(0, 4))
        elif u < err:
            i += 1
        else:
            out.append(int(tgt[i])); i += 1
    qry = np.array(out[:qlen], dtype=np.uint8)
    if n_frac > 0:
        qry[rng.random_sample(qlen) < n_frac] = 4
        tgt[rng.random_sample(tlen) < n_frac] = 4
    return qry, tgt


# ---------------------------------------------------------------------------
# the reference's minimap2 (oracle/_ref/libmm2ref.so)
# ---------------------------------------------------------------------------
class RefAln(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hits", "rs", "re", "qs", "qe", "blen", "mlen", "n_ambi", "dp_max", "dp_score", "score", "cnt",
                                         "rev", "mid_occ", "n_cigar")]


def ref_mm2_align(ref, qry, k=20, w=50, max_chain_iter=400):
    lib = mm2ref()
    rb, qb = ref.encode(), qry.encode()
    out = RefAln()
    cap = len(rb) + len(qb) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    lib.ref_mm2_align(rb, len(rb), qb, len(qb), k, w, max_chain_iter, C.byref(out), _p(cig), cap)
    d = {f: getattr(out, f) for f, _ in RefAln._fields_}
    d["cigar"] = cig[:max(out.n_cigar, 0)].copy()
    return d


def ref_mm_sketch(s, w, k):
    lib = mm2ref()
    b = s.encode()
    xy = np.zeros(2 * (len(b) + 8), dtype=np.uint64)
    n = lib.ref_mm_sketch(b, len(b), w, k, 0, 0, _p(xy), len(b) + 8)
    return xy[:2 * n].reshape(n, 2).copy()


def ref_mm_seeds(ref, qry, k=20, w=50, max_chain_iter=400):
    """The anchors the REFERENCE's mm_map_frag hands to mm_chain_dp (collect_seed_hits + radix sort, minimap2/map.c:293-303) in
    ConsensusGraph::alignRead's call sequence, read off the library's own seed dump: (xy [n, 2], mid_occ, rep_len)."""
    lib = mm2ref()
    lib.ref_mm_seeds.restype = C.c_int64
    rb, qb = ref.encode(), qry.encode()
    cap = 4 * (len(rb) + len(qb)) // max(w, 1) * 8 + 4096
    xy = np.zeros(2 * cap, dtype=np.uint64)
    mid, rep = C.c_int32(), C.c_int32()
    n = lib.ref_mm_seeds(rb, len(rb), qb, len(qb), k, w, max_chain_iter, _p(xy), C.c_int64(cap), C.byref(mid), C.byref(rep))
    assert 0 <= n <= cap, (n, cap)
    return xy[:2 * n].reshape(n, 2).copy(), int(mid.value), int(rep.value)


def ref_mm_chain_dp(xy, max_chain_iter=400, max_dist=5000, bw=500, max_skip=25, min_cnt=3, min_sc=40):
    """The REFERENCE's mm_chain_dp (minimap2/chain.c:22-164) on a sorted anchor list with the parameters of mm_map_frag (map.c:316):
    (u [n_u], reordered anchors [n_a, 2])."""
    lib = mm2ref()
    xy = np.ascontiguousarray(xy, dtype=np.uint64)
    n = len(xy)
    u = np.zeros(max(n, 1), dtype=np.uint64)
    a = np.zeros((max(n, 1), 2), dtype=np.uint64)
    na = C.c_int64()
    nu = lib.ref_mm_chain_dp(max_dist, max_dist, bw, max_skip, max_chain_iter, min_cnt, min_sc, C.c_float(1.0), 0, 1, C.c_int64(n), _p(xy), _p(u), _p(a), C.byref(na))
    return u[:nu].copy(), a[:int(na.value)].copy()


# ---------------------------------------------------------------------------
# the contig stage (oracle/consensus_oracle.cpp -> oracle/libconsoracle.so)
# ---------------------------------------------------------------------------
CONS_SO = os.path.join(ORACLE_DIR, "libconsoracle.so")
CONS_STREAMS = ["genome", "lone", "id", "pos", "type", "base", "complement"]
_CONS = None


class ConsOracleStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_contigs", "n_lone", "count_minhash", "count_minhash_not_in_graph", "count_aligner", "n_align_calls",
                                          "n_bad_roundtrip", "n_check_fail")] + [(n, C.c_double) for n in ("sketch_ms", "consensus_ms")]


def cons_lib():
    global _CONS
    if _CONS is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("consensus_oracle.cpp", "ns_oracle.c")]
        if not os.path.exists(CONS_SO) or os.path.getmtime(CONS_SO) < max(os.path.getmtime(s) for s in srcs):
            subprocess.run(["make", "-C", ORACLE_DIR, "libconsoracle.so"], check=True, capture_output=True)
        _CONS = C.CDLL(CONS_SO)
        for f in ("cons_oracle_convert_hit", "cons_oracle_optimize_edits", "cons_oracle_decode"):
            getattr(_CONS, f).restype = C.c_int64
    return _CONS


def ref_align_fn():
    """The reference's minimap2 behind ConsensusGraph::alignRead's call sequence (oracle/_ref/libmm2ref.so), as a C function pointer."""
    lib = mm2ref()
    assert lib is not None, "oracle/_ref/libmm2ref.so is missing: run `make -C oracle ref` where /root/reference exists"
    return C.cast(lib.ref_mm2_align, C.c_void_p)


def auto_schedule(n_reads, n_bases, n_filter_results):
    """The library's automatic schedule (csrc/consensus_driver.hip auto_schedule; nsgpu_set_schedule_auto) restated: (builders, bucket depth,
    rings, tail rings) in ONE group, from the read count, the bases and the whole-read filter results of all reads, both strands."""
    r = n_filter_results / n_reads if n_reads else 0.0
    if r < 30.0:
        depth, rings, tail, b_min = 3, 5, 3, 32
    elif r < 70.0:
        depth, rings, tail, b_min = 2, 4, 3, 96
    else:
        depth, rings, tail, b_min = 1, 4, 3, 128
    b = min(1024, max(b_min, n_bases // 10000000))
    return max(1, min(b, n_reads if n_reads else 1)), depth, rings, tail


def cons_oracle_run(bases, off, salts, k=23, n=60, thr=6, m_k=20, m_w=50, mci=400, edge_thr=4000000, num_thr=1, checks=True, id_base=0, align_fn=None,
                    lock_step=False, seed_hops=0, groups=4, seed_rings=1, seed_tail_rings=None, defer=None):
    """The reference's hot path (sketch + tables + Consensus::generateAndWriteConsensus) on the CPU with the reference's own minimap2
    answering alignRead.  Returns (streams, stats): streams[name] for num_thr == 1, else streams['threads'][t][name]; streams['metaData'].
    lock_step=True: num_thr LOCK-STEP virtual threads (the product's deterministic schedule restated around the literal thread body,
    oracle/consensus_oracle.cpp struct LockStep; groups 1 / 2 / 4; seed_hops >= 1: conflict-aware seeds with buckets of that depth and
    seed_rings rings); stats gains 'slots' and 'idle_seed_rounds'.  defer=(anchors, slots), one group: an alignment whose anchor list before
    chaining (the reference library's own index and sketch, ref_mm_count_seeds) is longer than `anchors` takes `slots` more slots
    (include/nsgpu.h nsgpu_set_defer)."""
    L = cons_lib()
    if defer is not None and defer[1]:
        assert lock_step and groups == 1
        L.cons_oracle_set_defer(C.cast(mm2ref().ref_mm_count_seeds, C.c_void_p), C.c_uint32(int(defer[0])), C.c_uint32(int(defer[1])))
    else:
        L.cons_oracle_set_defer(None, 0, 0)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    salts = np.ascontiguousarray(salts, dtype=np.uint64)
    ns = 7 * num_thr + 1
    ptrs = (C.c_void_p * ns)()
    lens = (C.c_uint64 * ns)()
    st = ConsOracleStats()
    ls_out = (C.c_uint64 * 2)()
    if lock_step:
        rc = L.cons_oracle_run_lockstep(_p(bases), _p(off), C.c_uint32(len(off) - 1), C.c_uint32(k), C.c_uint32(n), C.c_uint32(thr), _p(salts), m_k, m_w, mci,
                                        C.c_uint64(edge_thr), num_thr, int(checks), align_fn or ref_align_fn(), C.c_uint32(id_base), ptrs, lens, C.byref(st),
                                        int(seed_hops) + ((256 * (int(seed_rings) + 1) + 65536 * (int(seed_rings if seed_tail_rings is None else seed_tail_rings) + 1)) if seed_hops else 0), int(groups), ls_out)
    else:
        rc = L.cons_oracle_run(_p(bases), _p(off), C.c_uint32(len(off) - 1), C.c_uint32(k), C.c_uint32(n), C.c_uint32(thr), _p(salts), m_k, m_w, mci,
                               C.c_uint64(edge_thr), num_thr, int(checks), align_fn or ref_align_fn(), C.c_uint32(id_base), ptrs, lens, C.byref(st))
    assert rc == 0, "cons_oracle_run failed: %d" % rc
    per = []
    for t in range(num_thr):
        per.append({name: C.string_at(ptrs[7 * t + i], lens[7 * t + i]) for i, name in enumerate(CONS_STREAMS)})
    meta = C.string_at(ptrs[7 * num_thr], lens[7 * num_thr])
    for i in range(ns):
        L.cons_oracle_free(C.c_void_p(ptrs[i]))
    stats = {f: getattr(st, f) for f, _ in ConsOracleStats._fields_}
    if lock_step:
        stats["slots"], stats["idle_seed_rounds"] = int(ls_out[0]), int(ls_out[1])
    if num_thr == 1:
        out = dict(per[0])
        out["metaData"] = meta
        return out, stats
    return {"threads": per, "metaData": meta}, stats


def cons_oracle_convert_hit(hit, ref, qry):
    """alignRead's conversion (src/ConsensusGraph.cpp:218-397) of reg[0] as ref_mm2_align() returns it -> dict(ok, rel_pos, begin_offset, end_offset, edits)."""
    L = cons_lib()
    h = RefAln(**{f: int(hit[f]) for f, _ in RefAln._fields_})
    cig = np.ascontiguousarray(hit["cigar"], dtype=np.uint32)
    if cig.size == 0:
        cig = np.zeros(1, np.uint32)
    rb, qb = ref.encode(), qry.encode()
    cap = 2 * (len(rb) + len(qb)) + 8
    ed = np.zeros(cap, dtype=np.uint64)
    ok = C.c_int32()
    rp, bo, eo = C.c_int64(), C.c_int64(), C.c_int64()
    n = L.cons_oracle_convert_hit(C.byref(h), _p(cig), rb, C.c_uint64(len(rb)), qb, C.c_uint64(len(qb)), C.byref(ok), C.byref(rp), C.byref(bo), C.byref(eo),
                                  _p(ed), C.c_uint64(cap))
    assert n >= 0, n
    return {"ok": ok.value, "rel_pos": rp.value, "begin_offset": bo.value, "end_offset": eo.value, "edits": ed[:n].copy()}


def cons_oracle_optimize_edits(types, chars, nums):
    L = cons_lib()
    t = np.ascontiguousarray(types, dtype=np.uint8)
    b = np.ascontiguousarray(chars, dtype=np.uint8)
    m = np.ascontiguousarray(nums, dtype=np.uint32)
    cap = len(t) + 4
    ot, ob, om = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(cap, np.uint32)
    dis = C.c_uint64()
    n = L.cons_oracle_optimize_edits(_p(t), _p(b), _p(m), C.c_uint32(len(t)), _p(ot), _p(ob), _p(om), C.c_uint32(cap), C.byref(dis))
    assert n >= 0
    return int(dis.value), ot[:n].copy(), ob[:n].copy(), om[:n].copy()


def cons_oracle_check_repetitive(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    return int(cons_lib().cons_oracle_check_repetitive(b, C.c_uint64(len(b))))


def cons_oracle_decode(streams):
    """Decompressor's loop (src/Decompressor.cpp:105-172, 252-314) over one thread's seven streams -> [(id, read bytes)] or None if malformed."""
    L = cons_lib()
    bufs = [np.frombuffer(streams[nm], dtype=np.uint8) if len(streams[nm]) else np.zeros(1, np.uint8) for nm in CONS_STREAMS]
    ptrs = (C.c_void_p * 7)(*[b.ctypes.data for b in bufs])
    lens = (C.c_uint64 * 7)(*[len(streams[nm]) for nm in CONS_STREAMS])
    rcap = len(streams["complement"]) + streams["lone"].count(b"\n") + 4
    bcap = 1 << 16
    while True:
        ids = np.zeros(rcap, np.uint32)
        off = np.zeros(rcap + 1, np.uint64)
        out = np.zeros(bcap, np.uint8)
        n = L.cons_oracle_decode(ptrs, lens, _p(ids), _p(off), C.c_uint64(rcap), _p(out), C.c_uint64(bcap))
        if n == -2 and bcap < (1 << 34):
            bcap *= 8
            continue
        break
    if n < 0:
        return None
    raw = out.tobytes()
    return [(int(ids[i]), raw[int(off[i]):int(off[i + 1])]) for i in range(n)]


# ---- the block sorter ------------------------------------------------------------------------------------------------
BACKENDREF = os.path.join(ORACLE_DIR, "_ref", "backendref")


def bwt_naive(data):
    """What bsc_bwt_encode (libbsc/bwt/bwt.cpp:46-79 -> libsais_bwt_aux) returns, restated: suffixes sorted with the end of the block below
    every byte; B = T[n-1] + T[SA[k]-1] without the row of suffix 0; primary index = rank of suffix 0 + 1; indexes[t] = rank of suffix
    (t + 1) * r, r = the largest power of two <= n / 16 ... (the bit trick of bwt.cpp:50-56).  Pure Python: small inputs only.
    Pinned against the reference's own libbsc (oracle/_ref/backendref bwt) in tests/test_bwt_oracle.py."""
    T = bytes(data)
    n = len(T)
    if n == 0:
        return b"", 0, []
    sa = sorted(range(n), key=lambda i: T[i:])
    rank = [0] * n
    for k, s_ in enumerate(sa):
        rank[s_] = k
    out = bytearray([T[n - 1]])
    for s_ in sa:
        if s_:
            out.append(T[s_ - 1])
    mod = n // 8
    for sh in (1, 2, 4, 8, 16):
        mod |= mod >> sh
    r = (mod >> 1) + 1
    return bytes(out), rank[0] + 1, [rank[(t + 1) * r] for t in range((n - 1) // r)]


def ref_bwt(data):
    """libbsc's own block sorter on `data` as one block (oracle/_ref/backendref bwt: bsc_bwt_encode); n >= 16 (below that libsais rejects
    the sampling rate and bsc stores the block).  Returns (bwt bytes, primary index, indexes)."""
    import struct, tempfile
    assert os.path.exists(BACKENDREF), "oracle/_ref/backendref is missing: run `make -C oracle ref` where /root/reference exists"
    with tempfile.TemporaryDirectory() as td:
        a, b = os.path.join(td, "in"), os.path.join(td, "out")
        open(a, "wb").write(bytes(data))
        r = subprocess.run([BACKENDREF, "bwt", a, b], capture_output=True)
        assert r.returncode == 0, (r.returncode, r.stderr[-500:])
        d = open(b, "rb").read()
    index, num = struct.unpack("<ii", d[:8])
    return d[8 + 4 * num:], index, list(struct.unpack("<%di" % num, d[8:8 + 4 * num]))
