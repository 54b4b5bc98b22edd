"""GPU parity of the batched alignRead path: nsgpu_align_batch (host
decision chain + every banded DP on the HIP ksw_extd2 kernel) against the golden vectors the
reference's minimap2 emitted, against the reference run live when oracle/_ref travelled with the
snapshot, and against the reference's own CHECKS invariant for the Edit scripts."""
import numpy as np
import pytest

import nanospring_amd as ns
from tests import oracle_lib
from tests.align_cases import pairs
from tests.align_util import FIELDS, load_align_golden, check_alignread_invariant

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    g = ns.NsGpu()
    yield g
    g.close()


def edits_of(d):
    return [(int(e["type"]), int(e["base"]), int(e["num"])) for e in d["edits"]]


def test_golden_pairs(gpu):
    g = load_align_golden()
    res = ns.align_batch(gpu, g["refs"], g["qrys"], g["pair_ref"])
    n_ok = 0
    for i, d in enumerate(res):
        want = dict(zip(g["fields"], map(int, g["values"][i])))
        assert d["hits"] == want["hits"], i
        if want["hits"] == 0:
            assert not d["ok"] and d["n_edits"] == 0
            continue
        for f in FIELDS:
            assert d[f] == want[f], (i, f, d[f], want[f])
        assert np.array_equal(d["cigar"], g["cigar"][int(g["cigar_off"][i]):int(g["cigar_off"][i + 1])]), i
        if d["ok"]:
            check_alignread_invariant(g["refs"][int(g["pair_ref"][i])], g["qrys"][i], d, edits_of(d))
            n_ok += 1
    assert n_ok > 30


@pytest.mark.skipif(oracle_lib.mm2ref() is None, reason="oracle/_ref/libmm2ref.so not present")
def test_random_pairs_equal_live_reference(gpu):
    ps = pairs(2025, 320, big=True)
    refs, rid = [], []
    for r, _ in ps:
        if r not in refs:
            refs.append(r)
        rid.append(refs.index(r))
    res = ns.align_batch(gpu, refs, [q for _, q in ps], rid)
    n_ok = multi = 0
    for i, ((ref, q), d) in enumerate(zip(ps, res)):
        b = oracle_lib.ref_mm2_align(ref, q)
        assert d["hits"] == b["hits"], i
        if b["hits"] == 0:
            continue
        multi += b["hits"] > 1
        for f in FIELDS:
            assert d[f] == b[f], (i, f, d[f], b[f])
        assert np.array_equal(d["cigar"], b["cigar"]), i
        if d["ok"]:
            check_alignread_invariant(ref, q, d, edits_of(d))
            n_ok += 1
    assert n_ok > 150 and multi > 10
    st = ns.align_stats(gpu)
    assert st["dp_tasks"] > 1000 and st["dp_rounds"] <= 12
    # both seeding paths ran: the kernels' (seeds.hip + chain.hip) for most pairs, the host code for the pairs the kernel hands
    # back (tandem duplications: two anchors on one reference position)
    assert st["seed_pairs_gpu"] > 200 and st["seed_pairs_host"] > 0 and st["seed_pairs_gpu"] + st["seed_pairs_host"] >= len(ps), st


def test_many_queries_share_one_reference_at_scale(gpu):
    """cfg2-shaped batch: reads of one region against one consensus-like reference; every accepted
    alignment satisfies the reference's CHECKS invariant (size-independent property)."""
    bases, off = ns.synth_reads(21, 60000, 400, 8000.0)
    b = bytes(bases)
    reads = [b[int(off[i]):int(off[i + 1])].decode() for i in range(400)]
    ref = max(reads, key=len)
    comp = str.maketrans("ACGT", "TGCA")
    qs = reads + [r[::-1].translate(comp) for r in reads]
    res = ns.align_batch(gpu, [ref], qs, [0] * len(qs))
    n_ok = 0
    for q, d in zip(qs, res):
        if d["ok"]:
            check_alignread_invariant(ref, q, d, edits_of(d))
            n_ok += 1
    assert n_ok > 40
    assert not ns.align_batch(gpu, [ref], [], [])


OVERFLOW_WORKER = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import nanospring_amd as ns
from tests import oracle_lib
from tests.align_cases import pairs
g = ns.NsGpu()
ps = pairs(77, 48, big=False)
refs, rid = [], []
for r, _ in ps:
    if r not in refs:
        refs.append(r)
    rid.append(refs.index(r))
for rnd in range(3):
    res = ns.align_batch(g, refs, [q for _, q in ps], rid)
    for i, ((ref, q), d) in enumerate(zip(ps, res)):
        b = oracle_lib.ref_mm2_align(ref, q)
        assert d["hits"] == b["hits"], (rnd, i)
        if b["hits"]:
            assert all(d[f] == b[f] for f in ("rs", "re", "qs", "qe", "blen", "mlen", "dp_max")) and np.array_equal(d["cigar"], b["cigar"]), (rnd, i)
    st = ns.align_stats(g)
    print("ROUND", rnd, st["seed_pairs_gpu"], st["seed_pairs_host"])
'''


@pytest.mark.skipif(oracle_lib.mm2ref() is None, reason="oracle/_ref/libmm2ref.so not present")
def test_anchor_capacity_overflow_goes_through_the_host_code_and_grows():
    """The seeding kernel's anchor buffer too small (test switches): the pairs that do not fit are flagged and redone by the host
    code, results unchanged; the next launch has the room."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NSGPU_SEED_CAP_FACTOR="0", NSGPU_SEED_CAP_SLACK="0")
    r = subprocess.run([sys.executable, "-c", OVERFLOW_WORKER % {"root": root}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rounds = [list(map(int, l.split()[2:])) for l in r.stdout.splitlines() if l.startswith("ROUND")]
    assert len(rounds) == 3
    assert rounds[0][1] > 20                                  # first call: (nearly) everything handed back
    assert rounds[2][0] - rounds[1][0] > 20                   # later calls: the kernels' lists again


def test_device_plan_finds_every_problem_the_host_asks_for(gpu):
    """plan.hip: the alignment plan on the device is a prefetch -- the host's own plan looks the DP results up by key.  On the golden pairs
    (tandem duplications, several chains, N runs: many are left to the host on purpose) every problem of an alignment the device planned must
    be there, none unasked for, and the results are the reference's (test_golden_pairs runs the same batch)."""
    g = load_align_golden()
    ns.align_stats(gpu, reset=True)
    ns.align_batch(gpu, g["refs"], g["qrys"], g["pair_ref"])
    st = ns.align_stats(gpu, reset=True)
    assert st["plan_pairs_dev"] > 10 and st["plan_pairs_dev"] + st["plan_pairs_host"] == st["pairs"], st
    assert st["plan_hits"] > 100 and st["plan_misses"] == 0 and st["plan_extra"] == 0, st
