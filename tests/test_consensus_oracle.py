"""The contig-stage oracle (oracle/consensus_oracle.cpp: a literal, independent restatement of src/Consensus.cpp,
src/ConsensusGraph.cpp and Decompressor::generateRead that shares no code with the product) --
  1. its pieces that CAN be pinned are pinned: optimizeEditScript against the vectors emitted by the reference's own
     src/Edits.cpp object; alignRead's conversion by the reference's -DCHECKS invariant on hits of the reference's minimap2;
     checkRepetitive against hand-derived expectations; the decoder against the independent Python decoder;
  2. the PRODUCT's host-side contig code (nanospring_amd/csrc/consensus.cpp + the alignRead conversion in mm2.cpp, driven on the CPU
     by tests/host_harness.cpp) must give the oracle's streams byte for byte: iid, deep, repeat-rich, edge-case, edge-threshold
     and n = 128 inputs.  (The GPU engine is compared with the same oracle in tests/test_consensus_gpu.py.)"""
import os

import numpy as np
import pytest

import nanospring_amd as ns
from tests import host_lib, oracle_lib
from tests.align_cases import make_genome, mutate, revcomp
from tests.align_util import load_align_golden, check_alignread_invariant
from tests.stream_decode import decode, fold

HERE = os.path.dirname(os.path.abspath(__file__))
STREAMS = oracle_lib.CONS_STREAMS + ["metaData"]
needs_ref = pytest.mark.skipif(oracle_lib.mm2ref() is None, reason="oracle/_ref/libmm2ref.so not built")


def pack(reads):
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8) if sum(map(len, reads)) else np.zeros(1, np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    return bases, off


def repeat_rich_reads(seed, n_reads=220):
    rng = np.random.RandomState(seed)
    g = make_genome(rng, 30000)
    g = g + g[5000:9000] + make_genome(rng, 15000)
    reads = []
    for _ in range(n_reads):
        ln = int(max(400, rng.gamma(2.0, 1500.0)))
        st = rng.randint(0, max(1, len(g) - ln))
        s = mutate(rng, g[st:st + ln], 0.04)
        reads.append(revcomp(s) if rng.randint(2) else s)
    return reads


# ----------------------------------------------------------------------------------------------------- 1. pins of the oracle
def test_oracle_optimize_edit_script_equals_reference_object():
    z = np.load(os.path.join(HERE, "golden", "edit_cases.npz"))
    io, oo = z["in_off"], z["out_off"]
    for c in range(len(io) - 1):
        t, b, m = z["in_types"][io[c]:io[c + 1]], z["in_bases"][io[c]:io[c + 1]], z["in_nums"][io[c]:io[c + 1]]
        dis, ot, ob, om = oracle_lib.cons_oracle_optimize_edits(t, b, m)
        wt, wb, wm = z["out_types"][oo[c]:oo[c + 1]], z["out_bases"][oo[c]:oo[c + 1]], z["out_nums"][oo[c]:oo[c + 1]]
        assert dis == int(z["dis"][c]) and np.array_equal(ot, wt), c
        assert np.array_equal(om[ot == 0], wm[wt == 0]), c
        assert np.array_equal(ob[(ot == 1) | (ot == 3)], wb[(wt == 1) | (wt == 3)]), c
        assert (ob[ot == 2] == ord("-")).all(), c                           # src/Edits.cpp:49 surplus deletes carry '-'


def test_oracle_check_repetitive_hand_cases(oracle):
    rep = oracle_lib.cons_oracle_check_repetitive
    rng = np.random.RandomState(0)
    iid = "".join("ACGT"[i] for i in rng.randint(0, 4, size=1000))
    cases = [("A" * 100, 1),                       # shift 1: 100 matches > 70
             ("AC" * 100, 1),                      # shift 2 (and 4, 6): 200 > 140
             ("ACG" * 40, 1),                      # shift 3 and 6
             ("ACGTTGCA" * 50, 0),                 # period 8: shifts 1..6 match 100, 0, 100, 0, 100, 0 of 400 positions (<= 280)
             ("ACGTAC" * 30, 1),                   # period 6 = the last shift tried
             ("ACGTACG" * 30, 0),                  # period 7 is outside the six shifts: 0, 0, 60, 60, 0, 0 matches of 210
             (iid, 0), ("", 0), ("A", 1),          # len 1: (j+i) % 1 == j -> 1 match > 0.7
             ("AAAAAAACGT", 0),                    # 10 bases, shift 1: 6 same + wrap T/A no -> 6; 6 > 7.0 false
             ("AAAAAAAACG", 0),                    # shift 1: 7 matches; 7 > 7.0 is false (strict compare with a double)
             ("AAAAAAAAAC", 1)]                    # shift 1: 8 matches > 7.0
    for s, want in cases:
        assert rep(s) == want, (s[:16], len(s))
        if s:
            assert oracle.check_repetitive(s) == want, (s[:16], len(s))     # ns_oracle.c's copy (row a10's GPU checker)
    # the two restatements agree on reads with planted low-complexity stretches
    for i in range(200):
        ln = int(rng.randint(1, 400))
        per = int(rng.randint(1, 9))
        unit = "".join("ACGT"[j] for j in rng.randint(0, 4, size=per))
        s = list((unit * (ln // per + 1))[:ln])
        for j in rng.randint(0, ln, size=int(rng.randint(0, ln // 3 + 1))):
            s[j] = "ACGT"[rng.randint(4)]
        s = "".join(s)
        assert rep(s) == oracle.check_repetitive(s), s


@needs_ref
def test_oracle_convert_hit_checks_invariant_and_equals_product_conversion():
    """alignRead's CIGAR -> Edit conversion: on the reference minimap2's hits for the golden pairs (multi-hit, clipped, failing,
    overhanging cases), the oracle's script satisfies the reference's own CHECKS block, and the product's conversion
    (mm2.cpp behind host_lib.align) gives the same (ok, relPos, offsets, script)."""
    g = load_align_golden()
    n_ok = n_clip = n_fail = 0
    for i, q in enumerate(g["qrys"]):
        ref = g["refs"][int(g["pair_ref"][i])]
        hit = oracle_lib.ref_mm2_align(ref, q)
        o = oracle_lib.cons_oracle_convert_hit(hit, ref, q)
        p = host_lib.align(ref, q)
        assert bool(p["ok"]) == bool(o["ok"]), i
        if not o["ok"]:
            n_fail += 1
            continue
        ed = [(int(e & 0xff), int(e >> 8 & 0xff), int(e >> 16)) for e in o["edits"]]
        d = dict(hit, **o)
        check_alignread_invariant(ref, q, d, ed)
        assert (p["rel_pos"], p["begin_offset"], p["end_offset"]) == (o["rel_pos"], o["begin_offset"], o["end_offset"]), i
        assert np.array_equal(p["edits"], o["edits"]), i
        n_ok += 1
        n_clip += int(hit["rs"] > 0 and hit["qs"] > 0) + int(hit["re"] < len(ref) and hit["qe"] < len(q))
    assert n_ok > 30 and n_clip > 0 and n_fail > 0


def test_oracle_decoder_equals_python_decoder():
    bases, off = ns.synth_reads(5, 30000, 90, 2500.0)
    if oracle_lib.mm2ref() is None:
        pytest.skip("oracle/_ref/libmm2ref.so not built")
    out, st = oracle_lib.cons_oracle_run(bases, off, ns.mt19937_64_salts(60))
    assert st["n_bad_roundtrip"] == 0 and st["n_check_fail"] == 0
    got = oracle_lib.cons_oracle_decode(out)
    py = decode(out)
    assert got is not None and len(got) == 90 and {i: r for i, r in got} == py
    b = bytes(bases)
    for i, r in got:
        assert r == fold(b[int(off[i]):int(off[i + 1])])
    bad = dict(out, pos=out["pos"][:-3])
    assert oracle_lib.cons_oracle_decode(bad) is None


# ------------------------------------------------------------------ 2. the product's host contig code against the oracle
def both(bases, off, **kw):
    salts = ns.mt19937_64_salts(kw.get("n", 60))
    want, wst = oracle_lib.cons_oracle_run(bases, off, salts, checks=True, **kw)
    got, gst = host_lib.consensus(bases, off, salts, checks=False, ref_aligner=True, **kw)
    assert wst["n_bad_roundtrip"] == 0 and wst["n_check_fail"] == 0
    for s in STREAMS:
        assert got[s] == want[s], s
    for f in ("n_contigs", "n_lone", "count_minhash", "count_minhash_not_in_graph", "count_aligner", "n_align_calls"):
        assert gst[f] == wst[f], f
    return want, wst


@needs_ref
@pytest.mark.parametrize("seed,glen,n_reads,mean", [(5, 40000, 160, 2500.0), (3, 60000, 260, 3500.0), (1, 120000, 300, 8000.0)])
def test_product_graph_code_equals_oracle_iid(seed, glen, n_reads, mean):
    bases, off = ns.synth_reads(seed, glen, n_reads, mean)
    _, st = both(bases, off)
    assert st["count_aligner"] > 0.7 * n_reads


@needs_ref
@pytest.mark.parametrize("seed", [4, 21])
def test_product_graph_code_equals_oracle_repeat_rich(seed):
    """exact 4 kb duplication + both strands: removeCycles / splitPath have work all the time"""
    bases, off = pack(repeat_rich_reads(seed))
    _, st = both(bases, off)
    assert st["count_aligner"] > 100


@needs_ref
def test_product_graph_code_equals_oracle_deep_coverage():
    """cfg3's regime (BASELINE configs[2]: ~200x of a small genome): every window query returns most of the read set,
    contigs hold hundreds of reads, edges carry long read lists"""
    bases, off = ns.synth_reads(17, 12000, 400, 3000.0)          # 100x of 12 kb
    _, st = both(bases, off)
    assert st["n_contigs"] - st["n_lone"] <= 6 and st["count_aligner"] > 350


@needs_ref
def test_product_graph_code_equals_oracle_edge_cases():
    rng = np.random.RandomState(2)
    g = "".join("ACGT"[i] for i in rng.randint(0, 4, size=6000))
    reads = [g[0:3000], g[1000:4000], g[2000:5500], g[500:2500], "A" * 400, "ACGT", "", g[100:131], g[100:132], g[0:3000], g[3000:6000][::-1],
             g[1500:3500].replace("A", "N", 5), "AC" * 300, g[4000:6000], revcomp(g[2500:5000]), g[0:6000]]
    bases, off = pack(reads)
    want, st = both(bases, off)
    assert st["n_lone"] >= 4
    assert [r for _, r in sorted(oracle_lib.cons_oracle_decode(want))] == [fold(r.encode()) for r in reads]


@needs_ref
def test_product_graph_code_equals_oracle_with_binding_edge_threshold():
    """--edge-thr small enough to bind (src/Consensus.cpp:73, 86, 200): contigs are cut short in all three places"""
    bases, off = ns.synth_reads(9, 30000, 120, 2500.0)
    free, fst = both(bases, off)
    cut, cst = both(bases, off, edge_thr=20000)
    assert cst["n_contigs"] > fst["n_contigs"] and cut["genome"] != free["genome"]


@needs_ref
def test_product_graph_code_equals_oracle_num_hash_128():
    """--num-hash 128 (cfg5's sweep) through the whole contig stage"""
    bases, off = ns.synth_reads(23, 50000, 200, 3000.0)
    w128, s128 = both(bases, off, n=128, thr=6)
    w60, s60 = both(bases, off)
    assert s128["count_minhash"] > s60["count_minhash"]           # more tables, same threshold: more candidates pass


@needs_ref
def test_product_host_aligner_and_graph_code_equal_oracle_on_long_consensus():
    """A ~190 kb consensus with a tandem repeat: mid_occ leaves its trivial value.  Here the PRODUCT's whole host
    chain runs on the CPU (mm2.cpp index / seeds / chaining / skeleton with the CPU DP oracle + consensus.cpp) against the oracle
    with the reference's minimap2."""
    from tests.align_cases import long_consensus_reads
    bases, off = pack(long_consensus_reads())
    salts = ns.mt19937_64_salts(60)
    want, wst = oracle_lib.cons_oracle_run(bases, off, salts, checks=False)
    got, gst = host_lib.consensus(bases, off, salts, checks=False, ref_aligner=False)
    for s in STREAMS:
        assert got[s] == want[s], s
    longest = max(want["genome"].split(b"\n"), key=len).decode()
    mz = oracle_lib.ref_mm_sketch(longest, 50, 20)
    _, cnt = np.unique(mz[:, 0] >> np.uint64(8), return_counts=True)
    mid = int(oracle_lib.mm2ref().ref_mm_mid_occ(longest.encode(), 20, 50))
    assert len(longest) > 150000 and len(cnt) > 5000 and cnt.max() > mid


@needs_ref
def test_oracle_threads_race_but_stay_lossless():
    """-t 4: the reference's optimistic claiming; output is timing dependent but always lossless (SURVEY 8c 'Determinism')"""
    bases, off = ns.synth_reads(5, 40000, 160, 2500.0)
    out, st = oracle_lib.cons_oracle_run(bases, off, ns.mt19937_64_salts(60), num_thr=4, checks=False)
    assert st["n_bad_roundtrip"] == 0
    md = out["metaData"].decode().splitlines()
    assert md[0] == "numReads=160" and md[2] == "numThr=4"
    got = {}
    for t in out["threads"]:
        d = oracle_lib.cons_oracle_decode(t)
        assert d is not None
        got.update(dict(d))
    b = bytes(bases)
    assert sorted(got) == list(range(160)) and all(got[i] == fold(b[int(off[i]):int(off[i + 1])]) for i in range(160))


def test_lockstep_oracle_one_thread_is_t1_and_many_are_lossless_and_deterministic():
    """The oracle's lock-step virtual threads (struct LockStep: the engine's documented schedule around the literal thread body): one
    thread gives the -t 1 streams whatever the schedule; many threads are lossless, deterministic (unlike -t N) and every schedule
    terminates; conflict-aware seeds never make more contigs than the reference's rule on this input."""
    import nanospring_amd as ns
    bases, off = ns.synth_reads(31, 150000, 600, 4000.0)
    salts = ns.mt19937_64_salts(60)
    t1, s1 = oracle_lib.cons_oracle_run(bases, off, salts, checks=False)
    for groups, depth in ((4, 0), (1, 0), (2, 3)):
        a, sa = oracle_lib.cons_oracle_run(bases, off, salts, checks=False, num_thr=1, lock_step=True, groups=groups, seed_hops=depth)
        assert a == t1 and sa["n_contigs"] == s1["n_contigs"]
    runs = {}
    for groups, depth in ((4, 0), (1, 0), (1, 3), (2, 3)):
        a, sa = oracle_lib.cons_oracle_run(bases, off, salts, checks=True, num_thr=24, lock_step=True, groups=groups, seed_hops=depth)
        b, sb = oracle_lib.cons_oracle_run(bases, off, salts, checks=False, num_thr=24, lock_step=True, groups=groups, seed_hops=depth)
        assert a == b and sa["n_bad_roundtrip"] == 0 and sa["n_check_fail"] == 0
        runs[(groups, depth)] = sa
    assert runs[(1, 3)]["n_contigs"] <= runs[(1, 0)]["n_contigs"]
    assert runs[(1, 0)]["slots"] * 3 < runs[(4, 0)]["slots"] * 1.2      # a step per slot instead of one per four


def test_lockstep_oracle_deferred_alignments_rule():
    """The lock-step oracle's rule for deferred alignments (LockStep::VT::extra; include/nsgpu.h nsgpu_set_defer): the anchor count it is stated on
    -- ref_mm_count_seeds, the reference library's own index and sketch through its public calls -- equals the number of seeds the library's own
    debug dump prints for the pair (ref_mm_seeds), repeats included; with one thread the rule changes nothing but the slot count (-t 1 streams);
    with many threads the run stays lossless and deterministic, takes more slots, and a threshold nothing reaches is the run without the rule."""
    import ctypes as C
    import nanospring_amd as ns
    from tests.align_cases import make_genome, mutate
    lib = oracle_lib.mm2ref()
    lib.ref_mm_count_seeds.restype = C.c_int64
    rng = np.random.RandomState(5)
    g0 = make_genome(rng, 6000)
    unit = make_genome(rng, 23)
    g1 = g0[:2500] + unit * 50 + g0[2500:]
    for ref, qry in ((g0, mutate(rng, g0[300:4000], 0.05)), (g1, mutate(rng, g1[1800:5200], 0.03)), (g1, g1[2000:4500])):
        rb, qb = ref.encode(), qry.encode()
        n = lib.ref_mm_count_seeds(rb, len(rb), qb, len(qb), 20, 50)
        want = oracle_lib.ref_mm_seeds(ref, qry)
        assert n == len(want[0]), (n, len(want[0]))
    bases, off = ns.synth_reads(3, 200000, 330, 6000.0, genome="repeats")
    salts = ns.mt19937_64_salts(60)
    t1, s1 = oracle_lib.cons_oracle_run(bases, off, salts, checks=False)
    a, sa = oracle_lib.cons_oracle_run(bases, off, salts, checks=False, num_thr=1, lock_step=True, groups=1, seed_hops=3, defer=(150, 2))
    assert a == t1 and sa["slots"] > 0
    base, sb = oracle_lib.cons_oracle_run(bases, off, salts, checks=True, num_thr=10, lock_step=True, groups=1, seed_hops=3, seed_rings=2)
    off_, so = oracle_lib.cons_oracle_run(bases, off, salts, checks=False, num_thr=10, lock_step=True, groups=1, seed_hops=3, seed_rings=2, defer=(10 ** 9, 2))
    assert off_ == base and so["slots"] == sb["slots"]
    d1, sd1 = oracle_lib.cons_oracle_run(bases, off, salts, checks=True, num_thr=10, lock_step=True, groups=1, seed_hops=3, seed_rings=2, defer=(150, 2))
    d2, sd2 = oracle_lib.cons_oracle_run(bases, off, salts, checks=False, num_thr=10, lock_step=True, groups=1, seed_hops=3, seed_rings=2, defer=(150, 2))
    assert d1 == d2 and sd1["n_bad_roundtrip"] == 0 and sd1["n_check_fail"] == 0
    assert sd1["slots"] > sb["slots"]
