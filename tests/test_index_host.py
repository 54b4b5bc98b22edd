"""Host-side pieces of the single-sequence minimizer index (mm_idx_str for one sequence, index.c:386-434): the word-wise
ASCII -> nt4 conversion and RefIndex::build_from_sketch (radix-sorted 64-bit keys, open-addressing get) against plain
numpy restatements of "hash -> positions ascending" and of mm_idx_cal_max_occ (index.c:164-185)."""
import ctypes as C

import numpy as np

from tests import host_lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_nt4_codes_match_the_table_for_every_byte_and_alignment():
    L = host_lib.lib()
    table = np.full(256, 4, dtype=np.uint8)
    for ch, v in zip("ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        table[ord(ch)] = v
    table[ord("U")] = table[ord("u")] = 3                   # seq_nt4_table maps U like T ...
    table[:4] = [0, 1, 2, 3]                                # ... and its first row maps raw codes onto themselves (sketch.c:10)
    rng = np.random.RandomState(5)
    cases = [np.arange(256, dtype=np.uint8), np.frombuffer(b"ACGT" * 9 + b"N" + b"acgt" * 3 + b"ACGTACG", dtype=np.uint8)]
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 1000):
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, n)].copy()
        if n > 3:
            a[rng.randint(0, n, max(1, n // 50))] = rng.randint(0, 256, max(1, n // 50)).astype(np.uint8)
        cases.append(a)
    for a in cases:
        for shift in range(0, 9):                           # every alignment of the 8-byte words
            buf = np.concatenate([np.zeros(shift, np.uint8), a])
            out = np.zeros(len(buf) + 1, dtype=np.uint8)
            L.harness_nt4(C.c_void_p(buf.ctypes.data + shift), C.c_int64(len(a)), _p(out))
            assert np.array_equal(out[:len(a)], table[a]), (len(a), shift)
            assert out[len(a)] == 0                         # nothing written past the end


def _expect(xy):
    h = xy[:, 0] >> np.uint64(8)
    order = np.lexsort((xy[:, 1], h))
    hs, ys = h[order], xy[order, 1]
    keys, start = np.unique(hs, return_index=True)
    occ = np.diff(np.append(start, len(hs)))
    if len(keys) == 0:
        mid = 1
    else:
        kk = int(np.uint32((1.0 - float(np.float32(2e-4))) * len(keys)))
        mid = int(np.sort(occ)[kk]) + 1
    return keys, np.append(start, len(hs)).astype(np.uint32), ys, occ, mid


def _index(seq, xy, w=50, k=20):
    L = host_lib.lib()
    L.harness_index.restype = C.c_int64
    cap = len(xy) + 8
    keys, start, pos = np.zeros(cap, np.uint64), np.zeros(cap + 1, np.uint32), np.zeros(cap, np.uint64)
    first, gn = np.zeros(cap, np.uint64), np.zeros(cap, np.int32)
    mid = C.c_int32()
    flat = np.ascontiguousarray(xy.reshape(-1), dtype=np.uint64)
    nk = L.harness_index(seq.encode(), len(seq), w, k, _p(flat), C.c_int64(len(xy)), _p(keys), _p(start), _p(pos), C.c_int64(cap), C.byref(mid), _p(first), _p(gn))
    assert nk >= 0, nk
    return keys[:nk], start[:nk + 1], pos[:len(xy)], first[:nk], gn[:nk], mid.value


def test_index_of_real_sketches():
    rng = np.random.RandomState(9)
    for n in (60, 400, 8000, 30000, 120000):                # below and above the radix threshold (512 minimizers)
        seq = "".join(rng.choice(list("ACGT"), n))
        if n == 8000:
            seq = seq[:3000] + seq[1000:2500] + seq[3000:]  # a repeat: minimizers that occur twice
        xy = host_lib.sketch(seq, 50, 20)
        keys, start, pos, first, gn, mid = _index(seq, xy)
        ek, es, ep, occ, emid = _expect(xy)
        assert np.array_equal(keys, ek) and np.array_equal(start, es) and np.array_equal(pos, ep) and mid == emid
        assert np.array_equal(gn, occ) and np.array_equal(first, ep[es[:-1]])


def test_index_falls_back_when_the_key_trick_does_not_apply():
    """Minimizers out of position order, or hashes wider than 64 - log2(n) bits (k = 28), take the pair sort."""
    rng = np.random.RandomState(10)
    seq = "".join(rng.choice(list("ACGT"), 40000))
    xy = host_lib.sketch(seq, 50, 20)
    shuffled = xy[rng.permutation(len(xy))]
    keys, start, pos, first, gn, mid = _index(seq, shuffled)
    ek, es, ep, occ, emid = _expect(shuffled)
    assert np.array_equal(keys, ek) and np.array_equal(start, es) and np.array_equal(pos, ep) and mid == emid
    xy28 = host_lib.sketch(seq, 10, 28)                     # 56-bit hashes, ~7000 minimizers: 56 + 13 bits do not fit
    keys, start, pos, first, gn, mid = _index(seq, xy28, w=10, k=28)
    ek, es, ep, occ, emid = _expect(xy28)
    assert np.array_equal(keys, ek) and np.array_equal(start, es) and np.array_equal(pos, ep) and mid == emid
    assert np.array_equal(gn, occ)


def test_host_seeds_and_chains_against_reference_dumps():
    """The host code that redoes what the seeding kernel hands back (collect_seeds + the radix sort port) and the chain backtracking
    (chain_finish) against the REFERENCE directly: the anchor array of the library's own seed dump (oracle_lib.ref_mm_seeds) -- order
    included, ties too -- and the chains of the reference's mm_chain_dp on it."""
    from tests import align_cases, oracle_lib
    low = np.uint64(0xFFFFFFFF)
    n_ties = n_chains = 0
    for r, q in align_cases.pairs(5, 60):
        if not r or not q:
            continue
        want, wmid, _ = oracle_lib.ref_mm_seeds(r, q)
        h, hmid, _ = host_lib.seeds_full(r, q, 20, 50)
        assert hmid == wmid and len(h) == len(want)
        assert np.array_equal(h[:, 0] & low, want[:, 0] & low) and np.array_equal(h[:, 1] & low, want[:, 1] & low)
        assert np.array_equal((h[:, 1] >> np.uint64(32)) & np.uint64(0xFF), want[:, 1] >> np.uint64(32))
        n_ties += int(len(want) > 1 and bool((want[1:, 0] == want[:-1, 0]).any()))
        f, p = host_lib.chain_forward(h, 400)
        u, ra = host_lib.chain_finish(h, f, p, 400)
        wu, wa = oracle_lib.ref_mm_chain_dp(h, 400)
        assert np.array_equal(u, wu) and np.array_equal(ra, wa)
        n_chains += len(wu)
    assert n_ties > 3 and n_chains > 50
