"""Builds and wraps tests/_build/libhostharness.so: the product's host-side aligner logic
(nanospring_amd/csrc/mm2.cpp) driven by the CPU oracle DP.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libhostharness.so")
SRCS = [os.path.join(ROOT, "tests", "host_harness.cpp"), os.path.join(ROOT, "nanospring_amd", "csrc", "mm2.cpp"),
        os.path.join(ROOT, "nanospring_amd", "csrc", "consensus.cpp"), os.path.join(ROOT, "nanospring_amd", "csrc", "consensus_soa.cpp")]
CSRC = [os.path.join(ROOT, "oracle", "ksw2_oracle.c"), os.path.join(ROOT, "oracle", "ns_oracle.c")]
DEPS = SRCS + CSRC + [os.path.join(ROOT, "nanospring_amd", "csrc", "mm2.hpp"), os.path.join(ROOT, "nanospring_amd", "csrc", "consensus.hpp"),
                      os.path.join(ROOT, "nanospring_amd", "csrc", "consensus_soa.hpp"), os.path.join(ROOT, "nanospring_amd", "csrc", "dgraph.hpp")]


class HarnessAln(C.Structure):
    _fields_ = [("ok", C.c_int32), ("hits", C.c_int32), ("rel_pos", C.c_int64), ("begin_offset", C.c_int64), ("end_offset", C.c_int64)] + \
               [(n, C.c_int32) for n in ("rs", "re", "qs", "qe", "blen", "mlen", "n_ambi", "dp_max", "n_cigar", "mid_occ", "n_rounds", "n_dp")]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    objs = []
    for c in CSRC:
        obj = os.path.join(os.path.dirname(OUT), os.path.basename(c) + ".o")
        subprocess.run(["gcc", "-O2", "-fPIC", "-fopenmp", "-c", c, "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", "-Wall", "-DNSGPU_HOST_CHAIN"] + SRCS + objs + ["-o", OUT], check=True)
    return OUT


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def align(ref, qry, k=20, w=50, max_chain_iter=400):
    L = lib()
    rb, qb = ref.encode(), qry.encode()
    out = HarnessAln()
    ccap, ecap = len(qb) + len(rb) + 8, 2 * (len(qb) + len(rb)) + 8
    cig = np.zeros(ccap, dtype=np.uint32)
    ed = np.zeros(ecap, dtype=np.uint64)
    n = L.harness_align(rb, len(rb), qb, len(qb), k, w, max_chain_iter, C.byref(out), _p(cig), ccap, _p(ed), ecap)
    d = {f: getattr(out, f) for f, _ in HarnessAln._fields_}
    d["cigar"] = cig[:max(out.n_cigar, 0)].copy()
    d["edits"] = ed[:n].copy()
    return d


def seeds(ref, qry, k=20, w=50):
    """The sorted anchor list (x, y) that the chaining sees for one pair."""
    L = lib()
    L.harness_seeds.restype = C.c_int64
    rb, qb = ref.encode(), qry.encode()
    cap = 4 * (len(qb) + 64)
    while True:
        xy = np.zeros(2 * cap, dtype=np.uint64)
        n = L.harness_seeds(rb, len(rb), qb, len(qb), k, w, _p(xy), C.c_int64(cap))
        if n <= cap:
            return xy[:2 * n].reshape(n, 2).copy()
        cap = n


def seeds_full(ref, qry, k=20, w=50):
    """(anchors, mid_occ of the reference's index, mean query span) as the host code computes them for one pair."""
    L = lib()
    L.harness_seeds2.restype = C.c_int64
    rb, qb = ref.encode(), qry.encode()
    cap = 4 * (len(qb) + 64)
    mid, avg = C.c_int32(), C.c_float()
    while True:
        xy = np.zeros(2 * cap, dtype=np.uint64)
        n = L.harness_seeds2(rb, len(rb), qb, len(qb), k, w, _p(xy), C.c_int64(cap), C.byref(mid), C.byref(avg))
        if n <= cap:
            return xy[:2 * n].reshape(n, 2).copy(), mid.value, avg.value
        cap = n


def chain_forward(xy, max_chain_iter=400):
    """chain.c's f[] / p[] for one sorted anchor list, by the plain loop."""
    L = lib()
    xy = np.ascontiguousarray(xy, dtype=np.uint64)
    n = len(xy)
    f = np.zeros(max(n, 1), dtype=np.int32)
    p = np.zeros(max(n, 1), dtype=np.int32)
    L.harness_chain_forward(_p(xy), C.c_int64(n), max_chain_iter, _p(f), _p(p))
    return f[:n], p[:n]


def chain_finish(xy, f, p, max_chain_iter=400):
    """The product's chain_finish (backtracking + chain order, mm2.cpp) on forward-pass scores computed elsewhere: (u, reordered anchors)."""
    L = lib()
    L.harness_chain_finish.restype = C.c_int64
    a = np.ascontiguousarray(xy, dtype=np.uint64).copy()
    n = len(a)
    u = np.zeros(max(n, 1), dtype=np.uint64)
    na = C.c_int64()
    nu = L.harness_chain_finish(_p(a), C.c_int64(n), max_chain_iter, _p(np.ascontiguousarray(f, dtype=np.int32)), _p(np.ascontiguousarray(p, dtype=np.int32)), _p(u), C.byref(na))
    return u[:nu].copy(), a[:int(na.value)].copy()


def sketch(s, w, k):
    L = lib()
    b = s.encode()
    xy = np.zeros(2 * (len(b) + 8), dtype=np.uint64)
    n = L.harness_sketch(b, len(b), w, k, _p(xy), len(b) + 8)
    return xy[:2 * n].reshape(n, 2).copy()


def optimize_edits(types, bases, nums):
    """consensus.cpp's optimize_edit_script: (types, bases, nums) of the raw script -> (editDis, types, bases, nums)."""
    L = lib()
    t = np.ascontiguousarray(types, dtype=np.uint8)
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    m = np.ascontiguousarray(nums, dtype=np.uint32)
    cap = len(t) + 4
    ot, ob, om = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(cap, np.uint32)
    dis = C.c_uint64()
    L.harness_optimize_edits.restype = C.c_int64
    n = L.harness_optimize_edits(_p(t), _p(b), _p(m), C.c_uint32(len(t)), _p(ot), _p(ob), _p(om), C.c_uint32(cap), C.byref(dis))
    assert n >= 0
    return int(dis.value), ot[:n].copy(), ob[:n].copy(), om[:n].copy()


class HarnessConsStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_contigs", "n_lone", "count_minhash", "count_minhash_not_in_graph", "count_aligner", "n_align_calls",
                                          "n_bad_roundtrip", "n_graph_check_fail")] + \
               [(n, C.c_double) for n in ("update_ms", "mainpath_ms", "write_ms")]


STREAMS = ["genome", "lone", "id", "pos", "type", "base", "complement", "metaData"]


def consensus(bases, off, salts, k=23, n=60, thr=6, m_k=20, m_w=50, mci=400, edge_thr=4000000, checks=True, ref_aligner=False, id_base=0):
    """The reference's -t 1 contig loop as plain nested loops over the product's host graph code, CPU oracle filter and DP.
    ref_aligner=True answers every alignment with the reference's own minimap2 (oracle/_ref/libmm2ref.so) instead."""
    L = lib()
    if ref_aligner:
        from tests import oracle_lib
        fn = C.cast(oracle_lib.mm2ref().ref_mm2_align, C.c_void_p)
        L.harness_set_ref_align(fn)
    else:
        L.harness_set_ref_align(None)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    salts = np.ascontiguousarray(salts, dtype=np.uint64)
    ptrs = (C.c_void_p * 8)()
    lens = (C.c_uint64 * 8)()
    st = HarnessConsStats()
    L.harness_consensus(_p(bases), _p(off), C.c_uint32(len(off) - 1), C.c_uint32(k), C.c_uint32(n), C.c_uint32(thr), _p(salts), m_k, m_w, mci,
                        C.c_uint64(edge_thr), int(checks), ptrs, lens, C.byref(st), C.c_uint32(id_base))
    out = {}
    for i, name in enumerate(STREAMS):
        out[name] = C.string_at(ptrs[i], lens[i])
        L.harness_free(C.c_void_p(ptrs[i]))
    return out, {f: getattr(st, f) for f, _ in HarnessConsStats._fields_}
