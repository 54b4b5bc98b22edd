"""Pins oracle/ksw2_oracle.c against the reference's own ksw_extd2_sse
(oracle/_ref/libmm2ref.so, compiled from /root/reference/minimap2) on seeded
random problems for the four flag combinations NanoSpring reaches
(align.c:690-778): 0x08 gap fill (approx max), 0x00 second pass (exact),
0x40 right extension, 0xC2 left extension; and against the committed
golden vectors.  CPU only."""
import os

import numpy as np
import pytest

from tests import oracle_lib

FLAGS = [0x08, 0x00, 0x40, 0xC2]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ksw2_cases.npz")


def cases(seed, n):
    rng = np.random.RandomState(seed)
    out = []
    shapes = [(1, 1), (1, 40), (37, 1), (15, 16), (16, 16), (17, 16), (16, 32), (33, 31), (64, 48), (100, 100), (236, 240), (240, 236),
              (256, 256), (321, 300), (465, 470), (50, 100), (142, 262), (70, 116)]
    for i in range(n):
        ql, tl = shapes[i % len(shapes)]
        flag = FLAGS[(i // len(shapes)) % 4]
        w = [751, 751, 30, 10, -1][i % 5] if (i % 7) else max(ql, tl)
        zdrop = [400, 400, 50, 200][i % 4]
        end_bonus = -1 if flag in (0x08, 0x00) or i % 3 else 5
        dv = None if i % 6 else max(1, ql // 2)
        q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=[0.03, 0.1, 0.25][i % 3], n_frac=0.02 if i % 11 == 0 else 0.0, diverge_at=dv)
        out.append((q, t, w, zdrop, end_bonus, flag))
    # long extensions where the 751 band binds, and a long Z-dropping one
    for i, (ql, tl) in enumerate([(3000, 900), (900, 2500), (2000, 2000), (5000, 1170), (1600, 1601)]):
        for flag in (0x40, 0xC2, 0x00):
            q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=0.06, diverge_at=None if i % 2 else ql * 2 // 3)
            out.append((q, t, 751, 400, -1 if flag == 0 else 0, flag))
    return out


@pytest.mark.skipif(oracle_lib.mm2ref() is None, reason="oracle/_ref/libmm2ref.so not built")
def test_oracle_equals_reference_sse(oracle):
    n_zd = n_band = 0
    for ci, (q, t, w, zdrop, eb, flag) in enumerate(cases(2024, 360)):
        want_ez, want_c = oracle_lib.ref_ksw(q, t, w, zdrop, eb, flag)
        got_ez, got_c = oracle_lib.oracle_ksw(oracle, q, t, w, zdrop, eb, flag)
        assert got_ez == want_ez, (ci, len(q), len(t), w, zdrop, eb, hex(flag), got_ez, want_ez)
        assert np.array_equal(got_c, want_c), (ci, hex(flag))
        n_zd += want_ez[1]
        n_band += (0 <= w < abs(len(q) - len(t)) + 40)
    assert n_zd > 5 and n_band > 5


def test_oracle_equals_golden(oracle):
    z = np.load(GOLD)
    n = int(z["n"])
    seqs, so = z["seqs"], z["seq_off"]
    prm, ezs, cig, co = z["params"], z["ez"], z["cigar"], z["cigar_off"]
    for i in range(n):
        q = seqs[int(so[2 * i]):int(so[2 * i + 1])]
        t = seqs[int(so[2 * i + 1]):int(so[2 * i + 2])]
        w, zdrop, eb, flag = map(int, prm[i])
        got_ez, got_c = oracle_lib.oracle_ksw(oracle, q, t, w, zdrop, eb, flag)
        assert got_ez == tuple(int(v) for v in ezs[i]), i
        assert np.array_equal(got_c, cig[int(co[i]):int(co[i + 1])]), i
