"""Seeded (reference, query) pairs that exercise the branches of the aligner's decision
chain: single chains, tandem / interspersed repeats, paralogs, large indels
(long join, Z-drop, second DP pass, region split), junk inserts, overhanging ends, queries
without a hit, short queries, N bases, and a > 125 kb consensus (mid_occ off its trivial value)."""
import numpy as np

COMP = str.maketrans("ACGT", "TGCA")


def revcomp(s):
    return s[::-1].translate(COMP)


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.randint(0, 4, size=n))


def mutate(rng, s, p):
    if p <= 0:
        return s
    arr = np.frombuffer(s.encode(), dtype=np.uint8)
    u = rng.random_sample(len(arr))
    out = []
    bases = b"ACGT"
    for c, x in zip(arr, u):
        if x < p / 3:
            out.append(bases[rng.randint(4)])
        elif x < 2 * p / 3:
            out.append(bases[rng.randint(4)]); out.append(c)
        elif x < p:
            continue
        else:
            out.append(c)
    return bytes(out).decode()


def make_genome(rng, n):
    g = rand_seq(rng, n)
    # tandem repeat, interspersed repeat family, a diverged paralog, a homopolymer run
    unit = rand_seq(rng, rng.randint(20, 300))
    p = rng.randint(0, len(g))
    g = g[:p] + unit * rng.randint(3, 40) + g[p:]
    fam = rand_seq(rng, rng.randint(300, 1500))
    for _ in range(rng.randint(2, 8)):
        p = rng.randint(0, len(g))
        g = g[:p] + mutate(rng, fam, 0.03) + g[p:]
    p, q = rng.randint(0, len(g) - 3000), rng.randint(0, len(g))
    g = g[:q] + mutate(rng, g[p:p + rng.randint(500, 3000)], 0.05) + g[q:]
    p = rng.randint(0, len(g))
    g = g[:p] + "A" * rng.randint(10, 60) + g[p:]
    return g


def pairs(seed, n, big=True):
    rng = np.random.RandomState(seed)
    out = []
    g = make_genome(rng, 60000)
    for it in range(n):
        if it % 40 == 0:
            g = make_genome(rng, rng.randint(20000, 80000))
        kind = it % 16
        rl = rng.randint(1500, 40000)
        st = rng.randint(0, max(1, len(g) - rl))
        ref = g[st:st + rl]
        ql = int(max(60, rng.gamma(2.0, 3000.0)))
        off = rng.randint(-ql // 2, max(1, len(ref) - ql // 2))
        a = max(0, st + off)
        q = g[a:a + ql]
        err = [0.0, 0.01, 0.03, 0.06, 0.10][it % 5]
        if kind == 1 and len(q) > 800:            # large deletion in the query
            c = rng.randint(200, len(q) - 200)
            q = q[:c] + q[c + rng.randint(50, 3000):]
        elif kind == 2 and len(q) > 800:          # large insertion
            c = rng.randint(200, len(q) - 200)
            q = q[:c] + rand_seq(rng, rng.randint(50, 2500)) + q[c:]
        elif kind == 3 and len(q) > 1500:         # junk in the middle (Z-drop / split)
            c = rng.randint(400, len(q) - 400)
            w = rng.randint(100, 600)
            q = q[:c] + rand_seq(rng, w) + q[c + w:]
        elif kind == 4:                           # unrelated
            q = rand_seq(rng, ql)
        elif kind == 5:                           # opposite strand: MM_F_FOR_ONLY must drop it
            q = revcomp(q)
        elif kind == 6:                           # short
            q = q[:rng.randint(20, 200)]
        elif kind == 7 and len(q) > 2000:         # chimera of two distant loci
            b = rng.randint(0, max(1, len(g) - ql))
            q = q[:len(q) // 2] + g[b:b + ql // 2]
        elif kind == 8 and len(q) > 600:          # tandem duplication inside the query
            c = rng.randint(100, len(q) - 300)
            q = q[:c] + q[c:c + 200] * rng.randint(2, 6) + q[c + 200:]
        q = mutate(rng, q, err)
        if kind == 9 and len(q) > 100:
            qa = list(q)
            for i in rng.randint(0, len(q), size=max(1, len(q) // 200)):
                qa[i] = "N"
            q = "".join(qa)
        if not q:
            q = "ACGT"
        out.append((ref, q))
    if big:
        gg = make_genome(rng, 260000)             # > 125 kb: mid_occ leaves its trivial value
        for _ in range(4):
            a = rng.randint(0, len(gg) - 9000)
            out.append((gg, mutate(rng, gg[a:a + rng.randint(3000, 9000)], 0.03)))
    return out


def long_consensus_reads(seed=42, half=150000, cov=30, mean=8000):
    """Reads whose -t 1 contig stage grows one consensus of ~190 kb (> 5000 distinct minimizers) through a 23-bp-period
    tandem repeat: minimap2's mid_occ percentile (index.c:164-185) then drops the repeat's minimizer.
    Read 0 is a 40 kb read across the repeat so that the first contig starts there with a long window."""
    rng = np.random.RandomState(seed)
    unit = rand_seq(rng, 23)
    g0 = rand_seq(rng, half) + unit * 60 + rand_seq(rng, half)
    reads = [mutate(rng, g0[half - 20000:half + 20000], 0.03)]
    for _ in range(int(len(g0) * cov / mean)):
        ln = int(max(500, rng.gamma(2.0, mean / 2)))
        st = rng.randint(0, max(1, len(g0) - ln))
        reads.append(mutate(rng, g0[st:st + ln], 0.03))
    return reads
