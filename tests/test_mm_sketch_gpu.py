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
