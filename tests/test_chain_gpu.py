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


def tandem_lists():
    """what the seeding gives for a read across a tandem repeat (1.5 kb of a 7 / 11 / 23 / 40-mer unit between unique flanks, the read with 3 %
    errors): every reference minimizer of the repeat is hit from every copy in the read -- hundreds of anchors per reference position"""
    import random
    rnd = random.Random(4)
    out = []
    for unit_len, tlen in ((7, 1500), (11, 900), (23, 1500), (40, 1500)):
        unit = "".join(rnd.choice("ACGT") for _ in range(unit_len))
        ref = "".join(rnd.choice("ACGT") for _ in range(2500)) + (unit * (tlen // unit_len + 1))[:tlen] + "".join(rnd.choice("ACGT") for _ in range(3000))
        q = []
        for ch in ref:
            x = rnd.random()
            if x < 0.01:
                q.append(rnd.choice("ACGT"))
            elif x < 0.02:
                q.append(ch + rnd.choice("ACGT"))
            elif x >= 0.03:
                q.append(ch)
        out.append(host_lib.seeds(ref, "".join(q)))
    return out


LEVEL_WORKER = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import nanospring_amd as ns
from tests import host_lib
from tests.test_chain_gpu import tandem_lists, synthetic, gpu_scores
from tests import oracle_lib
g = ns.NsGpu()
rng = np.random.RandomState(12)
lists = tandem_lists()
assert max(len(a) for a in lists) > 15000, [len(a) for a in lists]
# batches that end exactly at / one short of / one past a block of 64 results, a level longer than the sixteen waves, levels of one
for n in (2048, 2111, 2112, 2113, 4096):
    lists.append(synthetic(rng, n, "dense"))
lists.append(synthetic(rng, 3000, "colinear"))
lists.append(synthetic(rng, 9000, "diagonals"))
one = synthetic(rng, 2500, "dense")
one[:, 0] = 777                                           # ONE reference position: nobody has a predecessor
lists.append(one)
got = gpu_scores(g, lists)
for a, (f, p) in zip(lists, got):
    wf, wp = host_lib.chain_forward(a, 400)
    assert np.array_equal(wf, f) and np.array_equal(wp, p), len(a)
    u, ra = host_lib.chain_finish(a, f, p, 400)
    wu, wa = oracle_lib.ref_mm_chain_dp(a, 400)
    assert np.array_equal(u, wu) and np.array_equal(ra, wa), len(a)
print("OK", len(lists), sum(len(a) for a in lists))
g.close()
'''


@pytest.mark.gpu
@pytest.mark.parametrize("level_min", ["768", "1", "0"])
def test_level_kernel_on_tandem_repeat_lists(level_min):
    """chain_forward_level_kernel (a workgroup per long list: the anchors of one reference position on sixteen waves at once): f / p identical to
    the sequential loop, and the chains through the product's backtracking identical to the REFERENCE's mm_chain_dp, on real tandem-repeat
    seed lists (20 000+ anchors) and on synthetic lists around the kernel's block and batch boundaries.  NSGPU_CHAIN_LEVEL_MIN=1: every
    list through it, also the ones without a repeated position; 0: the ring / LDS kernels as before (A/B switch)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", LEVEL_WORKER % {"root": root}], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, NSGPU_CHAIN_LEVEL_MIN=level_min, NSGPU_WAIT_TIMEOUT_S="120"))
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
