"""CPU tests of the structure-of-arrays consensus DAG (nanospring_amd/csrc/dgraph.hpp: the code the GPU runs with one workgroup per update, here
with a team of one) against the pointer graph (consensus.cpp) and the independent oracle (oracle/consensus_oracle.cpp).
  NSGPU_HARNESS_GRAPH=both  runs the two graphs side by side in the sequential -t 1 contig loop (tests/host_harness.cpp): after EVERY
  update + recompute the consensus, the contig's span and the graph's size must agree, and the emission of every contig must give the same bytes;
  NSGPU_SOA_DEBUG_FLAGS     takes the rare branches on every update: excursions one at a time (1), removeCycles by the reference's full walk (2),
  the left part of the path moved instead of its tail (4), splitPath's chain runs and the probes with a team of one (8), splitPath step by step and by
  stretches for its first 24 copies and only then by the reads' routes (128; by default by routes from the first edge), no routes at all (64);
  with 8 the routes are walked as sixteen lanes' work per read in turn, the way a workgroup's groups try edge ids ahead.
  NSGPU_HARNESS_SOA2_FLAGS  a second structure-of-arrays graph with these flags beside the first: whichever way the splits are taken, the nodes, the
  edges and every list must come out the same, id for id.
  NSGPU_SOA_SLACK           so little spare room in the arrays that removeCycles keeps stopping in front of splits that do not fit (ERR_ROOM) and is
  run again after the arrays have grown."""
import os
import subprocess
import sys

import pytest

from tests.test_graph_shortcuts import ROOT, WORKER


def run(kind, seed, oracle=False, **env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, kind, str(seed)] + (["oracle"] if oracle else []), env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "MISMATCH" not in r.stderr, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0].split()
    for o in [l.split() for l in r.stdout.splitlines() if l.startswith("ORACLE")]:
        assert o[1:] == line[1:], "the structure-of-arrays graph differs from oracle/consensus_oracle.cpp"
    return line[1], int(line[2]), int(line[3])


@pytest.mark.parametrize("kind,seed", [("iid", 3), ("repeats", 4), ("long", 1), ("homopolymer", 7)])
def test_soa_graph_equals_pointer_graph_after_every_update_and_the_oracle(kind, seed):
    ptr = run(kind, seed)
    both = run(kind, seed, oracle=True, NSGPU_HARNESS_GRAPH="both")          # (the streams of this run are the SoA graph's emission)
    assert both == ptr and ptr[1] > 100


@pytest.mark.parametrize("kind,seed,flags", [("repeats", 28, "3"), ("long", 1, "12"), ("repeats", 5, "15"), ("long", 3, "128"), ("homopolymer", 7, "136"), ("long", 1, "8"), ("repeats", 4, "64")])
def test_soa_graph_rare_branches_change_nothing(kind, seed, flags):
    assert run(kind, seed, NSGPU_HARNESS_GRAPH="both", NSGPU_SOA_DEBUG_FLAGS=flags) == run(kind, seed)


@pytest.mark.parametrize("kind,seed,flags2", [("long", 1, "64"), ("homopolymer", 7, "128"), ("repeats", 5, "8")])
def test_every_way_of_taking_a_split_builds_the_same_arrays(kind, seed, flags2):
    assert run(kind, seed, NSGPU_HARNESS_GRAPH="both", NSGPU_HARNESS_SOA2_FLAGS=flags2) == run(kind, seed)


@pytest.mark.parametrize("kind,seed", [("long", 3), ("repeats", 28)])
def test_remove_cycles_stops_in_front_of_a_split_that_does_not_fit_and_comes_again(kind, seed):
    assert run(kind, seed, NSGPU_HARNESS_GRAPH="both", NSGPU_SOA_SLACK="24") == run(kind, seed)
