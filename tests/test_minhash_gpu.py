"""GPU parity tests for the MinHash half of the path:
libnsgpu.so through its C-ABI versus the golden vectors emitted by the reference's
own objects and versus oracle/ns_oracle.c on seeded inputs.  Bit-exact."""
import numpy as np
import pytest

import nanospring_amd as ns
from tests import oracle_lib
from tests.golden_util import load_minhash

pytestmark = pytest.mark.gpu


def revcomp(s):
    return s[::-1].translate(str.maketrans("ATCG", "TAGC"))


def fold(s):
    return "".join("ATCG"[(ord(c) & 2) | ((ord(c) & 4) >> 2)] for c in s)


@pytest.fixture(scope="module")
def gold():
    return load_minhash()


@pytest.fixture(scope="module")
def gold_gpu(gold):
    g = ns.NsGpu(k=gold["k"], n=gold["n"], overlap_sketch_thr=gold["thr"])
    g.load_reads(gold["reads"])
    yield g
    g.close()


def test_pack_bytes_equal_dnabitset(gold, gold_gpu):
    p = 0
    for r, s in enumerate(gold["reads"]):
        nb = (len(s) + 3) // 4
        got, ln = gold_gpu.get_read_packed(r, maxlen=20000)
        assert ln == len(s)
        assert np.array_equal(got, gold["packed"][p:p + nb]), r
        assert gold_gpu.get_read(r, maxlen=20000) == fold(s)
        p += nb


def test_sketch_index_filter_equal_reference(gold, gold_gpu):
    g = gold_gpu
    sk = g.sketch(gold["salts"])
    assert np.array_equal(sk, gold["sketches"])
    g.build_index()
    N, n = sk.shape
    toff, tids = gold["table_off"], gold["table_ids"]
    for j in range(n):
        keys, start, ids = g.index_export(j)
        assert (np.diff(keys.astype(object)) > 0).all()
        for r in range(N):
            e = j * N + r
            want = tids[int(toff[e]):int(toff[e + 1])]
            pos = int(np.searchsorted(keys, sk[r, j]))
            assert keys[pos] == sk[r, j]
            assert np.array_equal(ids[start[pos]:start[pos + 1]], want)
    off, ids = g.filter_batch(gold["queries"])
    assert np.array_equal(off, gold["filter_off"])
    assert np.array_equal(ids, gold["filter_ids"])
    for qi in (0, 1, 5, 14, 20):
        want = gold["filter_ids"][int(gold["filter_off"][qi]):int(gold["filter_off"][qi + 1])]
        assert np.array_equal(g.filter(gold["queries"][qi]), want)


def test_packed_load_path(gold):
    g = ns.NsGpu(k=gold["k"], n=gold["n"], overlap_sketch_thr=gold["thr"])
    lens = np.array([len(r) for r in gold["reads"]], dtype=np.uint32)
    boff = np.zeros(len(lens), dtype=np.uint64)
    boff[1:] = np.cumsum((lens[:-1].astype(np.uint64) + 3) // 4)
    g.load_reads_packed(gold["packed"], boff, lens)
    assert np.array_equal(g.sketch(gold["salts"]), gold["sketches"])
    g.close()


def test_repetitive_flags(gold, gold_gpu, oracle):
    want = np.array([oracle.check_repetitive(fold(r)) if len(r) else 0 for r in gold["reads"]], dtype=np.uint8)
    got = gold_gpu.check_repetitive()
    assert np.array_equal(got, want)
    assert want.sum() >= 2


@pytest.mark.parametrize("k,n,thr,nreads,mean", [(23, 60, 6, 1500, 3000.0), (15, 20, 2, 300, 1500.0), (31, 128, 3, 300, 2000.0),
                                                 (9, 200, 4, 200, 1000.0), (23, 64, 1, 200, 1200.0)])
def test_random_reads_vs_oracle(oracle, k, n, thr, nreads, mean):
    bases, off = ns.synth_reads(100 + k, 200000, nreads, mean)
    salts = ns.mt19937_64_salts(n, 999 + n)
    g = ns.NsGpu(k=k, n=n, overlap_sketch_thr=thr)
    g.load_reads((bases, off))
    sk = g.sketch(salts)
    want = oracle.sketch_reads(bases, off, k, n, salts)
    assert np.array_equal(sk, want)
    g.build_index()
    idx = oracle.index_build(want)
    for j in (0, n // 2, n - 1):
        keys, start, ids = g.index_export(j)
        u = idx["nkeys"][j]
        assert np.array_equal(keys, idx["keys"][j][:u])
        assert np.array_equal(start, idx["start"][j][:u + 1])
        assert np.array_equal(ids, idx["ids"][j][:nreads])
    foff, fids = g.filter_all_reads()
    assert len(foff) == 2 * nreads + 1
    b = bytes(bases)
    total = 0
    for r in range(0, nreads, max(1, nreads // 120)):
        s = b[int(off[r]):int(off[r + 1])].decode()
        for strand, q in ((0, s), (1, revcomp(s))):
            w, _ = oracle.filter_string(q, k, salts, idx, thr)
            qi = 2 * r + strand
            assert np.array_equal(fids[int(foff[qi]):int(foff[qi + 1])], w), (r, strand)
            total += len(w)
        # the forward whole-read query always finds the read itself (multiplicity n >= thr)
        assert r in fids[int(foff[2 * r]):int(foff[2 * r + 1])]
    assert total > 0
    g.close()


def test_heavy_queries_use_counter_path(oracle):
    """> 2048 matches per query (repeat-rich data) leaves the LDS sort for the HBM counter path."""
    rng = np.random.RandomState(5)
    base = "".join("ACGT"[i] for i in rng.randint(0, 4, size=600))
    other = "".join("ACGT"[i] for i in rng.randint(0, 4, size=600))
    reads = [base] * 150 + [other] * 3 + [base[:300] + other[300:]] * 5
    k, n, thr = 23, 60, 6
    salts = ns.mt19937_64_salts(n)
    g = ns.NsGpu(k=k, n=n, overlap_sketch_thr=thr)
    g.load_reads(reads)
    sk = g.sketch(salts)
    g.build_index()
    rb, roff = oracle_lib.concat(reads)
    idx = oracle.index_build(oracle.sketch_reads(rb, roff, k, n, salts))
    qs = [base, other, revcomp(base), base[:300] + other[300:], base[100:500]]
    off, ids = g.filter_batch(qs)
    for qi, q in enumerate(qs):
        w, m = oracle.filter_string(q, k, salts, idx, thr)
        assert np.array_equal(ids[int(off[qi]):int(off[qi + 1])], w), qi
    _, m0 = oracle.filter_string(base, k, salts, idx, thr)
    assert m0 > 2048
    assert g.timing()["filter_matches"] >= m0
    g.close()


def test_empty_and_tiny_inputs():
    g = ns.NsGpu()
    g.load_reads([])
    assert g.sketch(ns.mt19937_64_salts(60)).shape == (0, 60)
    g.build_index()
    off, ids = g.filter_batch(["ACGT" * 20])
    assert list(off) == [0, 0] and len(ids) == 0
    g.load_reads(["ACGTACGTAC"])
    sk = g.sketch(ns.mt19937_64_salts(60))
    assert not sk.any()
    g.build_index()
    assert list(g.filter("ACGTACGTAC")) == [0]    # both sketches are all-zero rows: 60 matches >= 6
    with pytest.raises(ns.NsGpuError):
        g.get_read(5)
    g.close()


def test_linearity_property_at_scale():
    """Size-independent property at a larger size: the sketch of a read set does not depend on
    what else is loaded, and min-sketches compose: sketch(whole) = min(sketch of overlapping halves)."""
    bases, off = ns.synth_reads(11, 2000000, 20000, 8000.0)
    salts = ns.mt19937_64_salts(60)
    g = ns.NsGpu()
    g.load_reads((bases, off))
    sk = g.sketch(salts)
    b = bytes(bases)
    k = 23
    pieces, owner = [], []
    for r in range(0, 20000, 997):
        s = b[int(off[r]):int(off[r + 1])]
        h = len(s) // 2
        pieces += [s[:h + k - 1], s[h:]]
        owner.append(r)
    g2 = ns.NsGpu()
    g2.load_reads(pieces)
    sp = g2.sketch(salts)
    for i, r in enumerate(owner):
        assert np.array_equal(np.minimum(sp[2 * i], sp[2 * i + 1]), sk[r])
    g.close()
    g2.close()


def test_window_query_kernel_equals_oracle_on_ragged_windows(oracle):
    """ReadFilter::getFilteredReads of window strings through the one-kernel path (window_query_kernel: sketch + table search + sort + count in
    one workgroup per query): windows of every awkward length -- below k - 1, k - 1 (all-ones row), k, around the 1024-k-mer chunk seams, a whole
    read, lower case / N (folded like DnaBitset) -- in one batch with empty strings between them, against oracle/ns_oracle.c."""
    k, n, thr = 23, 60, 6
    bases, off = ns.synth_reads(21, 400000, 1500, 4000.0)
    salts = ns.mt19937_64_salts(n)
    g = ns.NsGpu(k=k, n=n, overlap_sketch_thr=thr)
    g.load_reads((bases, off))
    g.sketch(salts, fetch=False)
    g.build_index()
    idx = oracle.index_build(oracle.sketch_reads(bases, off, k, n, salts))
    b = bytes(bases).decode()
    rng = np.random.RandomState(9)
    qs = []
    for L in (0, 1, k - 2, k - 1, k, k + 1, 64, 1023 + k - 1, 1024 + k - 1, 1025 + k - 1, 2048 + k - 1, 3000):
        r = int(rng.randint(0, 1500))
        s = b[int(off[r]):int(off[r + 1])]
        qs += [s[:L], revcomp(s[:L]), s[max(0, len(s) - L):]]
    for r in (3, 700, 1499):
        s = b[int(off[r]):int(off[r + 1])]
        qs += [s, revcomp(s), s.lower(), s[:500] + "N" * 30 + s[530:]]
    qs.insert(5, "")
    o, ids = g.filter_batch(qs)
    assert len(o) == len(qs) + 1
    hits = 0
    for qi, q in enumerate(qs):
        w, _ = oracle.filter_string(fold(q), k, salts, idx, thr)
        assert np.array_equal(ids[int(o[qi]):int(o[qi + 1])], w), (qi, len(q))
        hits += len(w)
    assert hits > 50
    # one query at a time gives the same lists
    for qi in (0, 7, len(qs) - 4, len(qs) - 1):
        assert np.array_equal(g.filter(qs[qi]), ids[int(o[qi]):int(o[qi + 1])])
    g.close()
