"""The chaining-score kernel (nanospring_amd/csrc/chain.hip, nsgpu_chain_scores: the forward pass of mm_chain_dp,
minimap2/chain.c:43-92) against the same recurrence as a plain sequential loop (tests/host_harness.cpp ->
chain_forward_host, itself checked through whole alignments against the reference's minimap2 in test_align_host.py):
f[] and p[] must be identical for every anchor.  Anchor lists: what the seeding produces for the alignment cases
(tandem repeats, chimeras, junk), and synthetic lists built to hit the parts real lists rarely reach -- the max_skip
break, the max_chain_iter clamp, dense diagonals with ties, lists beyond the LDS variant's capacity."""
import ctypes as C

import numpy as np
import pytest

from tests import align_cases, host_lib


def gpu_scores(g, lists):
    from nanospring_amd._lib import check
    off = np.zeros(len(lists) + 1, dtype=np.uint64)
    for i, a in enumerate(lists):
        off[i + 1] = off[i] + len(a)
    tot = int(off[-1])
    xy = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.uint64).reshape(-1, 2) for a in lists] + [np.zeros((0, 2), dtype=np.uint64)]))
    f = np.zeros(max(tot, 1), dtype=np.int32)
    p = np.zeros(max(tot, 1), dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(g.lib, g.lib.nsgpu_chain_scores(g.ctx, vp(xy), vp(off), len(lists), vp(f), vp(p)))
    return [(f[int(off[i]):int(off[i + 1])], p[int(off[i]):int(off[i + 1])]) for i in range(len(lists))]


def compare(g, lists, mci):
    got = gpu_scores(g, lists)
    n_pred = 0
    for i, a in enumerate(lists):
        wf, wp = host_lib.chain_forward(a, mci)
        gf, gp = got[i]
        bad = np.flatnonzero((wf != gf) | (wp != gp))
        assert len(bad) == 0, (i, len(a), int(bad[0]), wf[bad[:4]], gf[bad[:4]], wp[bad[:4]], gp[bad[:4]])
        n_pred += int((wp >= 0).sum())
    return n_pred


def synthetic(rng, n, kind):
    """sorted anchors (x = reference position, y = span << 32 | query position) with controlled structure"""
    span = 20
    if kind == "diagonals":          # a few long diagonals interleaved + noise: many equally good predecessors, marks everywhere
        nd = rng.randint(2, 6)
        shift = rng.randint(-300, 300, size=nd)
        r = np.sort(rng.choice(np.arange(50, 50 + 12 * n), size=n, replace=False))
        d = rng.randint(0, nd, size=n)
        q = r + shift[d] + rng.randint(-2, 3, size=n)
    elif kind == "dense":            # a repeat: every reference position hit from many query positions
        r = np.sort(rng.randint(100, 100 + max(4, n // 6), size=n))
        q = rng.randint(50, 50 + max(4, n // 4), size=n)
    elif kind == "random":
        r = np.sort(rng.randint(0, 40 * n + 10, size=n))
        q = rng.randint(0, 40 * n + 10, size=n)
    else:                            # "colinear": one diagonal with growing gaps, some beyond max_gap / bw
        step = rng.choice([3, 30, 300, 3000, 6000], size=n, p=[0.5, 0.3, 0.15, 0.04, 0.01])
        r = 100 + np.cumsum(step)
        q = r + np.cumsum(rng.choice([0, 1, -1, 40, 700], size=n, p=[0.8, 0.08, 0.08, 0.03, 0.01]))
    q = np.maximum(q, span)
    spans = rng.choice([span, span - 1, span + 3], size=n, p=[0.9, 0.05, 0.05]).astype(np.uint64)
    xy = np.stack([r.astype(np.uint64), spans << np.uint64(32) | q.astype(np.uint64)], axis=1)
    return xy[np.argsort(xy[:, 0], kind="stable")]


@pytest.mark.gpu
def test_chain_scores_of_seeded_pairs():
    import nanospring_amd as ns
    g = ns.NsGpu()
    lists = [host_lib.seeds(r, q) for r, q in align_cases.pairs(11, 96)]
    assert sum(len(a) > 64 for a in lists) > 40 and any(len(a) == 0 for a in lists)
    assert compare(g, lists, 400) > 5000


@pytest.mark.gpu
@pytest.mark.parametrize("mci", [400, 37, 5000])
def test_chain_scores_of_synthetic_lists(mci):
    import nanospring_amd as ns
    g = ns.NsGpu(max_chain_iter=mci)
    rng = np.random.RandomState(1000 + mci)
    lists = []
    for kind in ("diagonals", "dense", "random", "colinear"):
        for n in (0, 1, 2, 63, 64, 65, 129, 700, 3000):
            lists.append(synthetic(rng, n, kind) if n else np.zeros((0, 2), dtype=np.uint64))
    lists.append(synthetic(rng, 9000, "dense"))          # the skip counter reaches max_skip all the time
    lists.append(synthetic(rng, 14000, "diagonals"))     # longer than the LDS variant takes: f / p / marks in global memory
    lists.append(synthetic(rng, 13500, "dense"))
    wide = synthetic(rng, 900, "diagonals")               # coordinates above 2^31 (a sequence id in the upper half of x): the general kernel
    wide[:, 0] += np.uint64(5 << 32)
    lists.append(wide)
    far = synthetic(rng, 300, "colinear")
    far[:, 0] += np.uint64(3000000000)
    lists.append(far)
    compare(g, lists, mci)


@pytest.mark.gpu
@pytest.mark.parametrize("mci", [400, 37])
def test_chain_kernel_against_the_reference_mm_chain_dp(mci):
    """Pinned directly against the REFERENCE: the kernel's f / p through the product's backtracking give the chains (u[]) and the
    reordered anchor array that the reference's own mm_chain_dp (oracle/_ref/libmm2ref.so, minimap2/chain.c:22-164, called with
    mm_map_frag's parameters) returns for the same sorted anchors -- seeded lists of the alignment cases and synthetic lists."""
    import nanospring_amd as ns
    from tests import oracle_lib
    g = ns.NsGpu(max_chain_iter=mci)
    rng = np.random.RandomState(77 + mci)
    lists = [host_lib.seeds(r, q) for r, q in align_cases.pairs(23, 60)]
    for kind in ("diagonals", "dense", "random", "colinear"):
        for n in (1, 2, 65, 700, 3000):
            lists.append(synthetic(rng, n, kind))
    lists.append(synthetic(rng, 9000, "dense"))
    # what a read across a tandem repeat looks like: tens of thousands of anchors, every reference position hit from many query positions
    # (beyond the LDS kernel's capacity: the ring kernel, anchors streaming through rings of 1024 entries)
    lists.append(synthetic(rng, 40000, "dense"))
    lists.append(synthetic(rng, 25000, "diagonals"))
    got = gpu_scores(g, lists)
    n_chains = 0
    for a, (f, p) in zip(lists, got):
        u, ra = host_lib.chain_finish(a, f, p, mci)
        wu, wa = oracle_lib.ref_mm_chain_dp(a, mci)
        assert np.array_equal(u, wu) and np.array_equal(ra, wa), (len(a), len(u), len(wu))
        n_chains += len(wu)
    assert n_chains > 60
    g.close()
