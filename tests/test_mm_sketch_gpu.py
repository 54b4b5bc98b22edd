"""Batched GPU mm_sketch (nanospring_amd/csrc/mm_sketch.hip, nsgpu_mm_sketch_batch) against the reference's own
mm_sketch (minimap2/sketch.c:77-143): committed golden vectors generated from the reference object
(tests/golden/make_golden.py sketch), the live reference object when it travelled (oracle/_ref/libmm2ref.so), and the
product's host implementation (mm2.cpp) on larger random batches."""
import os

import numpy as np
import pytest

from tests import host_lib, oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    z = np.load(os.path.join(HERE, "golden", "mm_sketch_cases.npz"))
    bases, off = z["bases"], z["off"]
    seqs = [bytes(bases[int(off[i]):int(off[i + 1])]).decode("latin-1") for i in range(len(off) - 1)]
    return seqs, z["wk"], z["xy"], z["xy_off"]


def test_host_sketch_equals_golden():
    """the product's host mm_sketch (used by the single-pair paths and as the GPU kernel's template)"""
    seqs, wk, xy, xo = golden()
    j = 0
    for w, k in wk:
        for s in seqs:
            want = xy[2 * int(xo[j]):2 * int(xo[j + 1])].reshape(-1, 2)
            j += 1
            got = host_lib.sketch(s, int(w), int(k)) if s else np.zeros((0, 2), dtype=np.uint64)
            assert np.array_equal(got, want), (w, k, len(s))


@pytest.mark.gpu
def test_gpu_sketch_equals_golden():
    import nanospring_amd as ns
    from nanospring_amd.filter import mm_sketch_batch
    seqs, wk, xy, xo = golden()
    g = ns.NsGpu()
    j = 0
    for w, k in wk:
        got = mm_sketch_batch(g, seqs, int(w), int(k))
        assert len(got) == len(seqs)
        for i, s in enumerate(seqs):
            want = xy[2 * int(xo[j]):2 * int(xo[j + 1])].reshape(-1, 2)
            j += 1
            assert np.array_equal(got[i], want), (w, k, i, len(s))
    g.close()


@pytest.mark.gpu
def test_gpu_sketch_large_batch_and_capacity_rerun():
    """2000 sequences of ragged lengths in one launch (several waves, lanes finishing at different times) and a batch
    whose tie-rich sequences overflow the first-pass capacity (exercises the exact-capacity rerun)."""
    import nanospring_amd as ns
    from nanospring_amd.filter import mm_sketch_batch
    rng = np.random.RandomState(5)
    seqs = []
    for i in range(2000):
        ln = int(rng.choice([0, 5, 40, 300, 2000, 9000])) + int(rng.randint(0, 50))
        seqs.append("".join("ACGT"[c] for c in rng.randint(0, 4, size=ln)))
    seqs.append("ACGTTGCA" * 4000)            # period-8 repeat: every window is full of equal hashes
    seqs.append("A" * 5000)
    seqs.append(("ACGGTCA" * 3 + "N") * 900)
    g = ns.NsGpu()
    for w, k in ((50, 20), (10, 15)):
        got = mm_sketch_batch(g, seqs, w, k)
        for i in list(range(0, 2000, 37)) + [2000, 2001, 2002]:
            want = host_lib.sketch(seqs[i], w, k) if seqs[i] else np.zeros((0, 2), dtype=np.uint64)
            assert np.array_equal(got[i], want), (w, k, i)
    if oracle_lib.mm2ref() is not None:
        for i in (2000, 2001, 2002, 7, 100):
            if seqs[i]:
                assert np.array_equal(mm_sketch_batch(g, [seqs[i]], 50, 20)[0], oracle_lib.ref_mm_sketch(seqs[i], 50, 20))
    g.close()


@pytest.mark.gpu
def test_gpu_sketch_fused_path_symmetric_kmers_and_tile_seams():
    """The fused kernels (pure-ACGT batches: one tile-local pass to count, one to write) on what makes them hard: k-mers equal to their
    own reverse complement (they push nothing and do not count in `run`) alone, in pairs, inside the first k bases, right at the
    1024-position tile seams and at the very end of a sequence; sequence lengths around k, w + k and the tile size; and an (AT)n run,
    which has so many of them that the batch must fall back to the general passes -- all against the reference's own mm_sketch."""
    import nanospring_amd as ns
    from nanospring_amd.filter import mm_sketch_batch
    rng = np.random.RandomState(77)

    def rnd(n):
        return "".join("ACGT"[c] for c in rng.randint(0, 4, size=n))

    def pal(k):                                  # a k-mer equal to its reverse complement (k even)
        h = rnd(k // 2)
        return h + h[::-1].translate(str.maketrans("ACGT", "TGCA"))

    def with_pals(n, at, k=20):
        s = list(rnd(n))
        for a in at:
            if 0 <= a and a + k <= n:
                s[a:a + k] = pal(k)
        return "".join(s)

    k, w = 20, 50
    seqs = [rnd(n) for n in (19, 20, 21, 69, 70, 71, 1023, 1024, 1025, 1043, 1044, 1045, 2047, 2048, 2049, 3000, 20000)]
    seqs += [with_pals(5000, [0]), with_pals(5000, [3]), with_pals(5000, [0, 1000, 1004, 1005, 1024, 2040, 2048, 4980]),
             with_pals(3000, [2980]), with_pals(1024, [1004]), with_pals(1025, [1005]), with_pals(2100, [1003 + i * 20 for i in range(4)]),
             with_pals(6000, list(range(100, 5900, 97))), "AT" * 700 + rnd(3000), rnd(3000) + "ACGT" * 300, "AT" * 40 + rnd(200)]
    g = ns.NsGpu()
    batches = [seqs, seqs[:23], [seqs[i] for i in (17, 18, 19, 20, 21, 22, 23, 24)]]      # with and without the sequences that force the fallback
    for batch in batches:
        got = mm_sketch_batch(g, batch, w, k)
        for i, s_ in enumerate(batch):
            want = oracle_lib.ref_mm_sketch(s_, w, k) if oracle_lib.mm2ref() is not None else host_lib.sketch(s_, w, k)
            assert np.array_equal(got[i], want), (i, len(s_))
    # other (w, k): odd k has no symmetric k-mers at all
    for w2, k2 in ((10, 15), (5, 28), (200, 12)):
        got = mm_sketch_batch(g, seqs[:20], w2, k2)
        for i, s_ in enumerate(seqs[:20]):
            assert np.array_equal(got[i], host_lib.sketch(s_, w2, k2)), (w2, k2, i)
    g.close()
