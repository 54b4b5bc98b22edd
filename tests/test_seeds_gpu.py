"""Index + seeds on the GPU (nanospring_amd/csrc/seeds.hip, nsgpu_seed_anchors: mm_idx_str's lookup structure,
mm_idx_cal_max_occ and collect_seed_hits with MM_F_FOR_ONLY, minimap2/index.c:164-248, map.c:215-247) against the host
code (mm2.cpp through tests/host_harness.cpp, which the CPU suite checks against the live reference minimap2 in whole
alignments): per pair the same sorted anchor list, the same mid_occ, the same mean span -- or a flag where the kernel
declines (anchors sharing a reference position: the order there is the reference's unstable radix sort's)."""
import ctypes as C

import numpy as np
import pytest

from tests import align_cases, host_lib


def gpu_seeds(g, ref_lists, qry_lists, pair_ref):
    from nanospring_amd._lib import check

    def cat(lists):
        off = np.zeros(len(lists) + 1, dtype=np.uint64)
        for i, a in enumerate(lists):
            off[i + 1] = off[i] + len(a)
        xy = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.uint64).reshape(-1, 2) for a in lists] + [np.zeros((0, 2), dtype=np.uint64)]))
        return xy, off
    rxy, roff = cat(ref_lists)
    qxy, qoff = cat(qry_lists)
    n = len(qry_lists)
    pr = np.ascontiguousarray(pair_ref, dtype=np.uint32)
    mid = np.zeros(max(n, 1), dtype=np.int32)
    flags = np.zeros(max(n, 1), dtype=np.uint32)
    avg = np.zeros(max(n, 1), dtype=np.float32)
    px, po = C.c_void_p(), C.c_void_p()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(g.lib, g.lib.nsgpu_seed_anchors(g.ctx, vp(rxy), vp(roff), len(ref_lists), vp(qxy), vp(qoff), vp(pr), n, C.byref(px), C.byref(po), vp(mid), vp(flags), vp(avg)))
    off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    tot = int(off[n])
    xy = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint64)), shape=(max(tot, 1) * 2,))[:tot * 2].copy().reshape(-1, 2)
    g.lib.nsgpu_free(px)
    g.lib.nsgpu_free(po)
    return [xy[int(off[i]):int(off[i + 1])] for i in range(n)], mid[:n], flags[:n], avg[:n]


def check_pairs(g, pairs, k=20, w=50):
    refs = {}
    ref_lists, pair_ref, qry_lists = [], [], []
    for r, q in pairs:
        if r not in refs:
            refs[r] = len(ref_lists)
            ref_lists.append(host_lib.sketch(r, w, k) if r else np.zeros((0, 2), dtype=np.uint64))
        pair_ref.append(refs[r])
        qry_lists.append(host_lib.sketch(q, w, k) if q else np.zeros((0, 2), dtype=np.uint64))
    got, mid, flags, avg = gpu_seeds(g, ref_lists, qry_lists, pair_ref)
    n_ties = n_anchors = 0
    for i, (r, q) in enumerate(pairs):
        want, wmid, wavg = host_lib.seeds_full(r, q, k, w)
        assert mid[i] == wmid, (i, mid[i], wmid)
        ties = len(want) > 1 and bool((want[1:, 0] == want[:-1, 0]).any())
        if ties:
            n_ties += 1
            assert flags[i] & 1 or (flags[i] & 2 and len(want) > 4096), (i, flags[i], len(want))
            continue
        assert flags[i] == 0 or (flags[i] & 2 and len(want) > 4096), (i, flags[i], len(want))
        if flags[i]:
            continue
        assert np.array_equal(got[i], want), (i, len(got[i]), len(want))
        assert avg[i] == np.float32(wavg), (i, avg[i], wavg)
        n_anchors += len(want)
    return n_ties, n_anchors


@pytest.mark.gpu
def test_seeds_of_alignment_cases():
    import nanospring_amd as ns
    g = ns.NsGpu()
    n_ties, n_anchors = check_pairs(g, align_cases.pairs(5, 120))
    assert n_anchors > 3000 and n_ties < 60, (n_anchors, n_ties)


@pytest.mark.gpu
def test_seeds_repeats_ties_and_frequent_minimizers():
    import nanospring_amd as ns
    g = ns.NsGpu()
    rng = np.random.RandomState(3)
    pairs = []
    unit = align_cases.rand_seq(rng, 700)
    for copies in (2, 5, 30):
        # a reference made of repeats: every hash occurs `copies` times; the query covers two units (tandem flags, ties)
        ref = "".join(align_cases.mutate(rng, unit, 0.002) for _ in range(copies)) + align_cases.rand_seq(rng, 3000)
        pairs.append((ref, align_cases.mutate(rng, unit + unit, 0.01)))
        pairs.append((ref, align_cases.mutate(rng, ref[300:2500], 0.03)))
    big = align_cases.make_genome(rng, 300000)            # > 5000 distinct minimizers: mid_occ below the largest count
    big = big[:150000] + (unit * 40) + big[150000:]
    for a in (1000, 149000, 152000, 200000):
        pairs.append((big, align_cases.mutate(rng, big[a:a + 6000], 0.02)))
    pairs.append(("", "ACGT" * 50))
    pairs.append((align_cases.rand_seq(rng, 5000), ""))
    pairs.append((align_cases.rand_seq(rng, 30), align_cases.rand_seq(rng, 30)))
    for k, w in ((20, 50), (15, 10), (28, 100)):
        n_ties, _ = check_pairs(g, pairs, k, w)
        assert (k, w) != (20, 50) or n_ties > 0


@pytest.mark.gpu
def test_empty_batches():
    """no pairs / no lists: both entry points return at once with empty results"""
    import nanospring_amd as ns
    from tests.test_chain_gpu import gpu_scores
    g = ns.NsGpu()
    got, mid, flags, avg = gpu_seeds(g, [np.zeros((0, 2), dtype=np.uint64)], [], [])
    assert got == [] and len(mid) == 0
    assert gpu_scores(g, []) == []
    got, mid, flags, avg = gpu_seeds(g, [np.zeros((0, 2), dtype=np.uint64)], [np.zeros((0, 2), dtype=np.uint64)], [0])
    assert len(got) == 1 and len(got[0]) == 0 and flags[0] == 0 and mid[0] == 1


@pytest.mark.gpu
def test_seed_kernel_against_the_reference_seed_dump():
    """Pinned directly against the REFERENCE: the anchors mm_map_frag hands to mm_chain_dp (collect_seed_hits + radix sort, read off the
    library's own MM_DBG_PRINT_SEED dump through oracle/_ref/libmm2ref.so) and its mid_occ, for the alignment cases and for repeat-rich /
    long-consensus pairs -- the kernel gives the same list in the same order, or flags the pair (two anchors on one reference position:
    the order there is the radix sort's and the host code redoes the pair)."""
    import nanospring_amd as ns
    from tests import oracle_lib
    g = ns.NsGpu()
    pairs = [(r, q) for r, q in align_cases.pairs(5, 90) if r and q]
    ref_lists, qry_lists, pair_ref = [], [], []
    for r, q in pairs:
        ref_lists.append(oracle_lib.ref_mm_sketch(r, 50, 20))
        qry_lists.append(oracle_lib.ref_mm_sketch(q, 50, 20))
        pair_ref.append(len(ref_lists) - 1)
    got, mid, flags, avg = gpu_seeds(g, ref_lists, qry_lists, pair_ref)
    n_same = n_flag = n_anchors = 0
    low = np.uint64(0xFFFFFFFF)
    for i, (r, q) in enumerate(pairs):
        want, wmid, _ = oracle_lib.ref_mm_seeds(r, q)
        assert mid[i] == wmid, (i, mid[i], wmid)
        if flags[i]:
            n_flag += 1
            continue
        a = got[i]
        assert len(a) == len(want), (i, len(a), len(want))
        if len(a):
            assert np.array_equal(a[:, 0] & low, want[:, 0] & low) and np.array_equal(a[:, 0] >> np.uint64(63), want[:, 0] >> np.uint64(63)), i
            assert np.array_equal(a[:, 1] & low, want[:, 1] & low) and np.array_equal((a[:, 1] >> np.uint64(32)) & np.uint64(0xFF), want[:, 1] >> np.uint64(32)), i
        n_same += 1
        n_anchors += len(a)
    assert n_same > 50 and n_anchors > 3000, (n_same, n_flag, n_anchors)
    g.close()
