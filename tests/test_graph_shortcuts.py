"""The two exact shortcuts in the consensus DAG maintenance (skip of no-op cycle pruning, re-use of every
stretch of the main path whose greedy choices the last update cannot have changed) must not change a single output byte: run the sequential contig loop with and
without them -- and against the independent oracle (oracle/consensus_oracle.cpp), which has no shortcut at all -- (NSGPU_NO_CYCLE_SKIP / NSGPU_NO_TAIL_SPLICE make the code take the reference's literal route)
on iid and on repeat-rich genomes and compare all streams."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys, hashlib
sys.path.insert(0, %(root)r)
import numpy as np
import nanospring_amd as ns
from tests import host_lib, oracle_lib
from tests.align_cases import make_genome, mutate, revcomp
kind, seed = sys.argv[1], int(sys.argv[2])
rng = np.random.RandomState(seed)
if kind == "iid":
    bases, off = ns.synth_reads(seed, 60000, 260, 3500.0)
elif kind == "homopolymer":                               # low-complexity consensus at depth: at most forks a side branch starts with the consensus's next base,
    g = "".join(c * int(rng.randint(1, 4)) for c in make_genome(rng, 12000))      # and the lists of reads on such branches grow long (the emission's tables give up: kAmbComplex)
    reads = []
    for _ in range(420):
        ln = int(max(400, rng.gamma(2.0, 1200.0)))
        st = rng.randint(0, max(1, len(g) - ln))
        s = mutate(rng, g[st:st + ln], 0.05)
        reads.append(revcomp(s) if rng.randint(2) else s)
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(r) for r in reads])
elif kind == "long":                                       # cfg2-like: 8 kb reads at 20x, contigs of a dozen reads and more
    bases, off = ns.synth_reads(seed, 800 * 8000 // 20, 800, 8000.0)   # seed 1: holds a left-hanging read that once broke the bookkeeping
else:
    g = make_genome(rng, 30000)
    g = g + g[5000:9000] + make_genome(rng, 15000)          # a 4 kb exact duplication and more repeats
    reads = []
    for _ in range(220):
        ln = int(max(400, rng.gamma(2.0, 1500.0)))
        st = rng.randint(0, max(1, len(g) - ln))
        s = mutate(rng, g[st:st + ln], 0.04)
        reads.append(revcomp(s) if rng.randint(2) else s)
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(r) for r in reads])
out, st = host_lib.consensus(bases, off, ns.mt19937_64_salts(60), checks=False, ref_aligner=oracle_lib.mm2ref() is not None)
assert st["n_bad_roundtrip"] == 0
h = hashlib.sha256()
for k in sorted(out):
    h.update(out[k])
print("HASH", h.hexdigest(), st["count_aligner"], st["n_contigs"])
_hl = host_lib.lib(); _hl.harness_soa_stat.restype = __import__("ctypes").c_uint64
print("SOASTAT", *[int(_hl.harness_soa_stat(i)) for i in range(13)])      # (the structure-of-arrays graph's counters, when it ran: tests/test_soa_graph.py)
if len(sys.argv) > 3 and oracle_lib.mm2ref() is not None:    # the independent oracle (oracle/consensus_oracle.cpp) on the same reads
    want, wst = oracle_lib.cons_oracle_run(bases, off, ns.mt19937_64_salts(60), checks=False)
    h = hashlib.sha256()
    for k in sorted(want):
        h.update(want[k])
    print("ORACLE", h.hexdigest(), wst["count_aligner"], wst["n_contigs"])
'''


def run(kind, seed, oracle=False, **env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, kind, str(seed)] + (["oracle"] if oracle else []), env=e, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    # NSGPU_SPLICE_CHECK=1: every re-used stretch of the main path is compared with a plain greedy walk, and the
    # "consistent from" bookkeeping with the graph, after every update
    assert "MISMATCH" not in r.stderr and "INVARIANT" not in r.stderr, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0].split()
    for o in [l.split() for l in r.stdout.splitlines() if l.startswith("ORACLE")]:
        assert o[1:] == line[1:], "product host graph code differs from oracle/consensus_oracle.cpp"
    return line[1], int(line[2]), int(line[3])


@pytest.mark.parametrize("kind,seed", [("iid", 3), ("repeats", 4), ("repeats", 5), ("long", 1), ("homopolymer", 7)])
def test_shortcuts_change_nothing(kind, seed):
    fast = run(kind, seed, oracle=True, NSGPU_SPLICE_CHECK="1")
    literal = run(kind, seed, NSGPU_NO_CYCLE_SKIP="1", NSGPU_NO_TAIL_SPLICE="1", NSGPU_NO_RUN_FASTPATH="1")
    assert fast == literal
    assert fast[1] > 100
    # edit emission: walks guided by the reads' own bases and the per-contig tables (default) / along the edges' read lists
    assert fast == run(kind, seed, NSGPU_EMIT_NO_SOURCE="1")


@pytest.mark.parametrize("kind,seed", [("repeats", 21), ("repeats", 28), ("long", 3)])
def test_cycle_pruning_from_the_noted_nodes_equals_the_full_walk(kind, seed):
    """remove_cycles served from the list of noted multi-in side nodes (default) against the reference's walk over every
    side branch (NSGPU_CYCLE_FULLSCAN=1), everything else equal."""
    assert run(kind, seed) == run(kind, seed, NSGPU_CYCLE_FULLSCAN="1")
