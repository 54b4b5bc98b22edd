"""Pins oracle/ns_oracle.c (our CPU restatement of the MinHash filter) against the
reference: (1) the committed golden vectors the reference's own objects emitted
(tests/golden/minhash_small.npz), (2) when oracle/_ref/nsref is present, the
reference objects run live on fresh random inputs.  CPU only."""
import numpy as np
import pytest

from tests import oracle_lib
from tests.golden_util import load_minhash
from nanospring_amd.filter import mt19937_64_salts


def test_salts_are_mt19937_64():
    # first outputs of std::mt19937_64(12345) (checked against libstdc++)
    s = mt19937_64_salts(3, 12345)
    assert list(map(int, s)) == [6597103971274460346, 7386862472818278521, 12716877617435052285]
    # the C++11 standard's check value: 10000th output of the default-seeded engine
    assert int(mt19937_64_salts(10000, 5489)[-1]) == 9981545732273789042


def test_pack_matches_dnabitset(oracle):
    g = load_minhash()
    p = 0
    for r in g["reads"]:
        nb = (len(r) + 3) // 4
        assert np.array_equal(oracle.pack2bit(r), g["packed"][p:p + nb])
        # round trip folds N / lowercase exactly like DnaBitset::to_string
        back = oracle.unpack2bit(g["packed"][p:p + nb], len(r))
        assert back == "".join("ATCG"[(ord(c) & 2) | ((ord(c) & 4) >> 2)] for c in r)
        p += nb
    assert p == len(g["packed"])


def test_sketch_matches_reference(oracle):
    g = load_minhash()
    sk = oracle.sketch_reads(g["read_bases"], g["read_off"], g["k"], g["n"], g["salts"])
    assert np.array_equal(sk, g["sketches"])
    for q, want in zip(g["queries"], g["qsketch"]):
        assert np.array_equal(oracle.sketch(q, g["k"], g["n"], g["salts"]), want)
    # edge cases called out in SURVEY A2
    k = g["k"]
    lens = [len(r) for r in g["reads"]]
    assert any(l < k - 1 for l in lens) and (k - 1) in lens and k in lens
    for r, row in zip(g["reads"], sk):
        if len(r) < k - 1:
            assert not row.any()
        elif len(r) == k - 1:
            assert (row == np.uint64(0xFFFFFFFFFFFFFFFF)).all()


def test_tables_match_reference(oracle):
    g = load_minhash()
    idx = oracle.index_build(g["sketches"])
    N, n = g["sketches"].shape
    toff, tids = g["table_off"], g["table_ids"]
    for j in range(n):
        keys = idx["keys"][j][:idx["nkeys"][j]]
        assert (np.diff(keys.astype(object)) > 0).all()
        for r in range(N):
            e = j * N + r
            want = tids[int(toff[e]):int(toff[e + 1])]
            pos = int(np.searchsorted(keys, g["sketches"][r, j]))
            a, b = idx["start"][j][pos], idx["start"][j][pos + 1]
            assert np.array_equal(idx["ids"][j][a:b], want)


def test_filter_matches_reference(oracle):
    g = load_minhash()
    idx = oracle.index_build(g["sketches"])
    hits = 0
    for qi, q in enumerate(g["queries"]):
        want = g["filter_ids"][int(g["filter_off"][qi]):int(g["filter_off"][qi + 1])]
        got, _ = oracle.filter_string(q, g["k"], g["salts"], idx, g["thr"])
        assert np.array_equal(got, want), qi
        hits += len(want)
    assert hits > 20


@pytest.mark.skipif(not oracle_lib.have_nsref(), reason="oracle/_ref/nsref not built (no /root/reference)")
@pytest.mark.parametrize("k,n,thr,seed", [(23, 60, 6, 1), (15, 20, 2, 2), (31, 128, 1, 3), (8, 64, 3, 4)])
def test_oracle_vs_live_reference(oracle, k, n, thr, seed):
    rng = np.random.RandomState(seed)
    g = "".join("ACGT"[i] for i in rng.randint(0, 4, size=20000))
    reads = []
    for _ in range(40):
        ln = rng.randint(1, 3000)
        st = rng.randint(0, len(g) - ln)
        s = list(g[st:st + ln])
        for i in rng.randint(0, ln, size=ln // 50):
            s[i] = "ACGTN"[rng.randint(5)]
        reads.append("".join(s))
    queries = reads[:10] + [oracle.revcomp(r) for r in reads[:10]] + [g[1000:3000], g[:k - 1], g[:k]]
    salts = rng.randint(0, 2 ** 63, size=n).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, size=n).astype(np.uint64)
    ref = oracle_lib.run_nsref(reads, queries, k, n, thr, salts)
    rb, roff = oracle_lib.concat(reads)
    sk = oracle.sketch_reads(rb, roff, k, n, salts)
    assert np.array_equal(sk, ref["sketches"])
    idx = oracle.index_build(sk)
    for qi, q in enumerate(queries):
        got, _ = oracle.filter_string(q, k, salts, idx, thr)
        assert np.array_equal(got, ref["filter"][qi]), qi


def test_check_repetitive(oracle):
    assert oracle.check_repetitive("A" * 100) == 1
    assert oracle.check_repetitive("AC" * 100) == 1
    assert oracle.check_repetitive("ACGTTGCA" * 50) == 0    # period 8: shifts 1..6 match 100, 0, 100, 0, 100, 0 of 400 positions, none > 280
    assert oracle.check_repetitive("ACGTACG" * 30) == 0     # period 7 is outside the six shifts (src/Consensus.cpp:411)
    assert oracle.check_repetitive("AAAAAAAACG") == 0       # 7 of 10 match at shift 1: 7 > 0.7 * 10 is false (:419, strict)
    assert oracle.check_repetitive("AAAAAAAAAC") == 1       # 8 of 10
    rng = np.random.RandomState(0)
    assert oracle.check_repetitive("".join("ACGT"[i] for i in rng.randint(0, 4, size=1000))) == 0
