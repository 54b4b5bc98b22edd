"""Builds and wraps tests/_build/libhostharness.so: the product's host-side aligner logic
(nanospring_amd/csrc/mm2.cpp) driven by the CPU oracle DP.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libhostharness.so")
SRCS = [os.path.join(ROOT, "tests", "host_harness.cpp"), os.path.join(ROOT, "nanospring_amd", "csrc", "mm2.cpp")]
CSRC = [os.path.join(ROOT, "oracle", "ksw2_oracle.c")]
DEPS = SRCS + CSRC + [os.path.join(ROOT, "nanospring_amd", "csrc", "mm2.hpp")]


class HarnessAln(C.Structure):
    _fields_ = [("ok", C.c_int32), ("hits", C.c_int32), ("rel_pos", C.c_int64), ("begin_offset", C.c_int64), ("end_offset", C.c_int64)] + \
               [(n, C.c_int32) for n in ("rs", "re", "qs", "qe", "blen", "mlen", "n_ambi", "dp_max", "n_cigar", "mid_occ", "n_rounds", "n_dp")]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    obj = os.path.join(os.path.dirname(OUT), "ksw2_oracle.o")
    subprocess.run(["gcc", "-O2", "-fPIC", "-c", CSRC[0], "-o", obj], check=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wall"] + SRCS + [obj, "-o", OUT], check=True)
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


def sketch(s, w, k):
    L = lib()
    b = s.encode()
    xy = np.zeros(2 * (len(b) + 8), dtype=np.uint64)
    n = L.harness_sketch(b, len(b), w, k, _p(xy), len(b) + 8)
    return xy[:2 * n].reshape(n, 2).copy()
