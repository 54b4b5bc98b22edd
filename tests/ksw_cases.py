"""Seeded ksw_extd2 problems that reach every kernel class and every branch the aligner uses: tiny, gap-fill sized, mid, long banded
extensions (up to 5000), thin bands, all flag combinations NanoSpring reaches plus APPROX_DROP, Z-drop on and off, N bases, reads that
diverge from the target part way through."""
import numpy as np

from tests import oracle_lib


def diverse_cases(seed, n):
    rng = np.random.RandomState(seed)
    out = []
    for it in range(n):
        kind = it % 8
        if kind == 0:
            ql, tl = rng.randint(1, 40), rng.randint(1, 40)
        elif kind == 1:
            ql = rng.randint(10, 400); tl = max(1, ql + rng.randint(-30, 31))
        elif kind == 2:
            ql = rng.randint(300, 1500); tl = max(1, ql + rng.randint(-200, 201))
        elif kind == 3:
            ql, tl = rng.randint(1, 60), rng.randint(100, 900)
        elif kind == 4:
            ql, tl = rng.randint(100, 900), rng.randint(1, 60)
        elif kind == 5:
            ql = rng.randint(500, 2500); tl = max(1, ql + rng.randint(-50, 51))
        elif kind == 6:
            ql = rng.randint(2500, 5000); tl = max(1, ql + rng.randint(-300, 101))
        else:
            ql = rng.randint(120, 260); tl = max(1, ql + rng.randint(-12, 13))
        w = [751, 751, -1, 10, 50, 100, 3, 200][rng.randint(8)]
        if kind == 6:
            w = [751, 300, 100][rng.randint(3)]
        flag = [0x08, 0x00, 0x40, 0xC2, 0x48, 0x42, 0x18, 0x08][rng.randint(8)]
        zdrop = [400, 200, -1, 50][rng.randint(4)]
        div = None if rng.randint(3) else rng.randint(0, ql + 1)
        q, t = oracle_lib.ksw_random_problem(rng, ql, tl, err=[0.04, 0.15, 0.4][rng.randint(3)], n_frac=[0, 0, 0.02][rng.randint(3)], diverge_at=div)
        out.append((q, t, int(w), int(zdrop), -1, int(flag)))
    return out
