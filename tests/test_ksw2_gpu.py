"""GPU parity of the ksw_extd2 wavefront kernel through the C-ABI:
bit-exact ez fields + CIGAR versus the golden vectors emitted by the reference's own
ksw_extd2_sse and versus oracle/ksw2_oracle.c on seeded random problems, for every flag
combination NanoSpring reaches, band-limited / Z-dropped / empty problems, and the
HBM-state variant used for very long problems."""
import os

import numpy as np
import pytest

import nanospring_amd as ns
from tests import oracle_lib
from tests.test_ksw2_oracle import cases, GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    g = ns.NsGpu()
    yield g
    g.close()


def test_golden_calls(gpu):
    z = np.load(GOLD)
    n = int(z["n"])
    seqs, so = z["seqs"], z["seq_off"]
    probs = []
    for i in range(n):
        q = seqs[int(so[2 * i]):int(so[2 * i + 1])]
        t = seqs[int(so[2 * i + 1]):int(so[2 * i + 2])]
        w, zdrop, eb, flag = map(int, z["params"][i])
        probs.append((q, t, w, zdrop, eb, flag))
    ezs, cigs = ns.ksw_extd2_batch(gpu, probs)
    co = z["cigar_off"]
    for i in range(n):
        assert ezs[i] == tuple(int(v) for v in z["ez"][i]), (i, probs[i][2:], ezs[i], tuple(z["ez"][i]))
        assert np.array_equal(cigs[i], z["cigar"][int(co[i]):int(co[i + 1])]), i


def test_random_vs_oracle(gpu, oracle):
    probs = cases(31337, 400)
    ezs, cigs = ns.ksw_extd2_batch(gpu, probs)
    nzd = 0
    for i, (q, t, w, zdrop, eb, flag) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(oracle, q, t, w, zdrop, eb, flag)
        assert ezs[i] == we, (i, len(q), len(t), w, zdrop, eb, hex(flag), ezs[i], we)
        assert np.array_equal(cigs[i], wc), i
        nzd += we[1]
    assert nzd > 5


def test_empty_and_degenerate(gpu, oracle):
    q = np.array([0, 1, 2, 3], dtype=np.uint8)
    e = np.zeros(0, dtype=np.uint8)
    probs = [(e, q, 751, 400, -1, 0x08), (q, e, 751, 400, -1, 0x40), (q, q, 0, 400, -1, 0x08), (q[:1], q[:1], 751, 400, -1, 0x00)]
    ezs, cigs = ns.ksw_extd2_batch(gpu, probs)
    for i, (qq, tt, w, zd, eb, fl) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(oracle, qq, tt, w, zd, eb, fl)
        assert ezs[i] == we and np.array_equal(cigs[i], wc), i
    assert ns.ksw_extd2_batch(gpu, []) == ([], [])


def test_long_problems_use_hbm_state(gpu, oracle):
    """LONG_JOIN-sized gap fills (bw = max(len)) do not fit the 64 KiB LDS class."""
    rng = np.random.RandomState(9)
    probs = []
    for ql, tl, flag, w in [(7000, 6900, 0x08, 7000), (6800, 7100, 0x00, 7100), (5000, 9000, 0x40, 751), (9000, 6000, 0xC2, 751)]:
        q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.05)
        probs.append((q, t, w, 400, -1 if flag < 0x40 else 0, flag))
    ezs, cigs = ns.ksw_extd2_batch(gpu, probs)
    for i, (q, t, w, zd, eb, fl) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(oracle, q, t, w, zd, eb, fl)
        assert ezs[i] == we, (i, ezs[i], we)
        assert np.array_equal(cigs[i], wc), i


def test_cigar_consumes_both_sequences_at_scale(gpu):
    """Size-independent property on a production-sized batch (typical 240 x 240 gap fills):
    a global (non-extension) alignment that is not Z-dropped consumes exactly qlen / tlen
    bases, and its score equals the score recomputed from the CIGAR."""
    rng = np.random.RandomState(4)
    probs = []
    for i in range(4000):
        ql = int(rng.randint(200, 330))
        q, t = oracle_lib.ksw_random_problem(rng, ql, ql + int(rng.randint(-8, 9)), err=0.04)
        probs.append((q, t, 751, 400, -1, 0x08))
    ezs, cigs = ns.ksw_extd2_batch(gpu, probs)
    for (q, t, *_), ez, c in zip(probs, ezs, cigs):
        assert not ez[1]
        ops, lens = c & 0xf, c >> 4
        assert int(lens[ops != 2].sum()) == len(q) and int(lens[ops != 1].sum()) == len(t)
        sc, i, j = 0, 0, 0
        for op, ln in zip(ops, lens):
            ln = int(ln)
            if op == 0:
                eq = q[j:j + ln] == t[i:i + ln]
                sc += 2 * int(eq.sum()) - 4 * int((~eq).sum()); i += ln; j += ln
            else:
                sc -= min(4 + 2 * ln, 24 + ln)
                if op == 1: j += ln
                else: i += ln
        assert sc == ez[8]


VARIANT_WORKER = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib
from tests.test_ksw2_oracle import cases
orc = oracle_lib.Oracle()
g = ns.NsGpu()
probs = cases(4711, 300)
rng = np.random.RandomState(3)
for ql, tl, flag, w in [(900, 950, 0x08, 751), (1500, 1400, 0x40, 751), (2100, 2300, 0x42, 751), (700, 40, 0x40, 751), (60, 900, 0xC2, 751)]:
    q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.06)
    probs.append((q, t, w, 400, 0 if flag >= 0x40 else -1, flag))
bad = 0
for prm in (oracle_lib.KSW_DEFAULT, dict(a=10, b=30, sc_ambi=5, q=30, e=20, q2=80, e2=10), dict(a=1, b=9, sc_ambi=1, q=16, e=2, q2=41, e2=1)):
    ezs, cigs = ns.ksw_extd2_batch(g, probs, **prm)
    for i, (q, t, w, zd, eb, fl) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(orc, q, t, w, zd, eb, fl, prm)
        if ezs[i] != we or not np.array_equal(cigs[i], wc):
            bad += 1
print("RESULT", bad)
'''


@pytest.mark.parametrize("env", [{}, {"NSGPU_KSW_NO_WG": "1"}])
def test_scores_that_wrap_int8_in_earnest(env):
    """minimap2's scores keep every in-band value small; the int8 wrap-around arithmetic of the reference only shows in
    the out-of-band cells of 16-cell blocks.  With large scores (q2 + e2 = 90) in-band values wrap too -- the reference
    wraps, the oracle wraps, the kernels must wrap identically -- for the workgroup kernels and (NSGPU_KSW_NO_WG) for the
    one-wave kernel alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", VARIANT_WORKER % {"root": root}], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 0, (env, line)


def test_register_kernels_diverse_vs_oracle(gpu, oracle):
    """ksw2_reg.hip (state in registers, packed int16, one wave or a workgroup per problem) on problems of every class: all flag
    combinations, thin bands that bind (out-of-band garbage cells with int8 wrap-around feed in-band cells), Z-drops, N bases."""
    from tests.ksw_cases import diverse_cases
    probs = diverse_cases(7, 640)
    ezs, cigs = ns.ksw_extd2_batch(gpu, probs)
    n_long = n_zd = 0
    for i, (q, t, w, zdrop, eb, flag) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(oracle, q, t, w, zdrop, eb, flag)
        assert ezs[i] == we, (i, len(q), len(t), w, zdrop, hex(flag), ezs[i], we)
        assert np.array_equal(cigs[i], wc), (i, len(q), len(t), w, hex(flag))
        n_long += len(t) > 1536
        n_zd += we[1]
    assert n_long > 40 and n_zd > 40


def overhang_cases(seed, n):
    """Extensions that run off the target's end (a read hanging over the end of its contig): a query of thousands of bases against a
    short target, band exhausted long before the query's end -- the shape the register kernels' exact early exit serves (ksw2_reg.hip)
    -- with related, unrelated and low-complexity sequences, all extension flags, thin and default bands, Z-drop on and off, plus
    near-misses of the exit's precondition (band exhaustion row around the query's end)."""
    rng = np.random.RandomState(seed)
    out = []
    for it in range(n):
        w = int([751, 751, 751, 300, 100, 50][rng.randint(6)])
        tl = int(rng.randint(1, 1300))
        kind = it % 6
        if kind == 5:                                   # precondition boundary: 2 (tl - 1) + w + 1 around ql - 1
            ql = max(1, 2 * (tl - 1) + w + 1 + int(rng.randint(-3, 4)) + 1)
        else:
            ql = int(rng.randint(2 * tl + w + 2, 2 * tl + w + 3000))
        ql = min(ql, 5000)
        flag = int([0x40, 0xC2, 0x42, 0x00, 0x40, 0xC2][rng.randint(6)])
        zdrop = int([400, 400, -1, 50, 200][rng.randint(5)])
        if kind in (0, 5):                              # related: the target is the start of the query, with errors
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=[0.03, 0.1][rng.randint(2)])
        elif kind == 1:                                 # unrelated
            q = rng.randint(0, 4, size=ql).astype(np.uint8); t = rng.randint(0, 4, size=tl).astype(np.uint8)
        elif kind == 2:                                 # homopolymer / short tandem repeat on both sides
            u = rng.randint(0, 4, size=int(rng.randint(1, 5))).astype(np.uint8)
            q = np.resize(u, ql).copy(); t = np.resize(u, tl).copy()
            q[rng.randint(0, ql, size=ql // 50 + 1)] = rng.randint(0, 4); t[rng.randint(0, tl, size=tl // 50 + 1)] = rng.randint(0, 4)
        elif kind == 3:                                 # the target re-appears deep in the query's overhang
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.03)
            at = int(rng.randint(tl, max(tl + 1, ql - tl)))
            q[at:at + tl] = t[:min(tl, ql - at)]
        else:                                           # N bases
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.05, n_frac=0.03)
        out.append((np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8), w, zdrop, 0 if flag >= 0x40 else -1, flag))
    return out


def test_overhang_extensions_with_early_exit_vs_oracle(gpu, oracle):
    probs = overhang_cases(2024, 720)
    ezs, cigs = ns.ksw_extd2_batch(gpu, probs)
    n_zd = n_cls = 0
    for i, (q, t, w, zdrop, eb, flag) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(oracle, q, t, w, zdrop, eb, flag)
        assert ezs[i] == we, (i, len(q), len(t), w, zdrop, hex(flag), ezs[i], we)
        assert np.array_equal(cigs[i], wc), (i, len(q), len(t), w, hex(flag))
        n_zd += we[1]
        n_cls += len(t) > 512
    assert n_zd > 500 and n_cls > 200


LATENCY_WORKER = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib
from tests.ksw_cases import diverse_cases
from tests.test_ksw2_gpu import overhang_cases
orc = oracle_lib.Oracle()
g = ns.NsGpu()
probs = diverse_cases(17, 320) + overhang_cases(99, 360)
ezs, cigs = ns.ksw_extd2_batch(g, probs)
bad = 0
for i, (q, t, w, zd, eb, fl) in enumerate(probs):
    we, wc = oracle_lib.oracle_ksw(orc, q, t, w, zd, eb, fl)
    if ezs[i] != we or not np.array_equal(cigs[i], wc):
        bad += 1
print("RESULT", bad, len(probs))
'''


@pytest.mark.parametrize("env", [{"NSGPU_KSW_NO_EARLY_EXIT": "1"}, {"NSGPU_KSW_SERIAL_BACKTRACK": "1"}, {"NSGPU_KSW_ALL_BOOKS": "1"}, {"NSGPU_KSW_PROMOTE_ROWS": "0"}])
def test_latency_classes_and_early_exit_switches(env):
    """The A/B switches of the register DP kernels that are still in the code -- the exact early exit, the one-lane traceback walk, books in
    every wave, the promotion rule of long narrow problems -- each against the oracle on problems with many
    anti-diagonals: bit-exact either way."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", LATENCY_WORKER % {"root": root}], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 0 and int(line[2]) == 680, (env, line)


def test_no_score_flag_changes_nothing_but_the_score(gpu, oracle):
    """KSW_EZ_NS_NO_SCORE (0x80000, ksw2_reg.hip): what the aligner sets on its gap fills -- approximate mode without KSW_EZ_APPROX_DROP, whose
    per-row books only produce ez.score, which nothing on NanoSpring's path reads.  Every other field and the CIGAR stay bit-exact against the
    oracle: one-wave classes (the path without books), the multi-wave class (books skipped), banded problems, and a band that runs out
    (zdropped = 1, no CIGAR)."""
    rng = np.random.RandomState(77)
    probs = []
    for ql, tl, w in [(240, 250, 751), (300, 310, 751), (420, 400, 751), (500, 512, 100), (700, 690, 751), (1000, 980, 751), (1500, 1400, 751), (900, 300, 64), (2500, 200, 100), (256, 255, 5)]:
        for _ in range(3):
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.08)
            probs.append((q, t, w, 400, -1, 0x08))
            probs.append((q, t, w, 400, -1, 0x08 | 0x02))
    flagged = [(q, t, w, zd, eb, fl | 0x80000) for q, t, w, zd, eb, fl in probs]
    ezs, cigs = ns.ksw_extd2_batch(gpu, flagged)
    n_drop = 0
    for i, (q, t, w, zd, eb, fl) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(oracle, q, t, w, zd, eb, fl)
        got = ezs[i]
        assert got[:8] == we[:8] and got[9:] == we[9:], (i, len(q), len(t), w, hex(fl), got, we)
        assert got[8] in (0, we[8]), (i, got, we)              # (0: a register kernel skipped the books; the first-generation kernels ignore the flag)
        assert np.array_equal(cigs[i], wc), i
        n_drop += we[1]
    assert n_drop > 0


def target_longer_cases(seed, n):
    """Extensions whose target window is longer than the query (what minimap2 hands the DP for every read end inside its contig: about twice the
    query's length of target, align.c:617-677) -- the shape of the register kernels' second early exit (KSW_EZ_NS_NO_MTE, ksw2_reg.hip): related,
    unrelated and low-complexity sequences, the query re-appearing further along the target (the one way a far cell can score), N bases, all
    extension flags, Z-drop on / off / tight, default and thin bands incl. bands that run out before the target does and the boundary of that."""
    rng = np.random.RandomState(seed)
    out = []
    for it in range(n):
        w = int([751, 751, 751, 300, 100, 50][rng.randint(6)])
        ql = int(rng.randint(2, 760))
        kind = it % 7
        if kind == 6:                                   # the band runs out about where the target ends: c2 ~ (c1 + c2 + w) / 2, i.e. tl ~ ql + w
            tl = max(ql + 1, ql + w + int(rng.randint(-3, 4)))
        else:
            tl = int(rng.randint(ql + 1, int(2.6 * ql) + 40))
        tl = min(tl, 1536 if it % 11 else 2400)
        flag = int([0x40, 0xC2, 0x42, 0x40, 0xC2][rng.randint(5)])
        zdrop = int([400, 400, -1, 50, 200][rng.randint(5)])
        eb = int([0, 0, -1, 10][rng.randint(4)])
        if kind in (0, 6):                              # related: the query is the start of the target, with errors
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=[0.03, 0.1, 0.2][rng.randint(3)])
        elif kind == 1:                                 # unrelated
            q = rng.randint(0, 4, size=ql).astype(np.uint8); t = rng.randint(0, 4, size=tl).astype(np.uint8)
        elif kind == 2:                                 # homopolymer / short tandem repeat on both sides
            u = rng.randint(0, 4, size=int(rng.randint(1, 5))).astype(np.uint8)
            q = np.resize(u, ql).copy(); t = np.resize(u, tl).copy()
            q[rng.randint(0, ql, size=ql // 50 + 1)] = rng.randint(0, 4); t[rng.randint(0, tl, size=tl // 50 + 1)] = rng.randint(0, 4)
        elif kind == 3:                                 # the query re-appears further along the target
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.03)
            at = int(rng.randint(min(ql, tl - 1), tl))
            t[at:at + ql] = q[:min(ql, tl - at)]
        elif kind == 4:                                 # related only over the first part: the alignment Z-drops or ends early
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.05, diverge_at=int(rng.randint(1, ql)))
        else:                                           # N bases
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.05, n_frac=0.03)
        out.append((np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8), w, zdrop, eb, flag))
    return out


def test_target_longer_extensions_with_the_second_early_exit_vs_oracle(gpu, oracle):
    """KSW_EZ_NS_NO_MTE (0x100000): the aligner's extensions declare ez.mte / mte_q / score unread, and the register kernels stop the sweep of an
    extension whose target is longer than its query once no later row can change max, mqe or the Z-drop decision.  Everything else -- max, max_t,
    max_q, mqe, mqe_t, zdropped, reach_end and the CIGAR -- stays bit-exact against the oracle (which sweeps to the end, as the reference does); and
    without the flag the very same problems are bit-exact in every field."""
    probs = target_longer_cases(515, 840)
    flagged = [(q, t, w, zd, eb, fl | 0x100000) for q, t, w, zd, eb, fl in probs]
    ezs, cigs = ns.ksw_extd2_batch(gpu, flagged)
    ezp, cigp = ns.ksw_extd2_batch(gpu, probs)
    n_exit = n_zd = n_end = 0
    for i, (q, t, w, zd, eb, fl) in enumerate(probs):
        we, wc = oracle_lib.oracle_ksw(oracle, q, t, w, zd, eb, fl)
        assert ezp[i] == we and np.array_equal(cigp[i], wc), (i, len(q), len(t), w, zd, hex(fl), ezp[i], we)
        got = ezs[i]
        assert got[:6] == we[:6] and got[9:] == we[9:], (i, len(q), len(t), w, zd, eb, hex(fl), got, we)
        assert np.array_equal(cigs[i], wc), (i, len(q), len(t), w, hex(fl))
        n_exit += got[6:9] != we[6:9]
        n_zd += we[1]
        n_end += we[10]
    assert n_exit > 150 and n_zd > 100 and n_end > 50, (n_exit, n_zd, n_end)
