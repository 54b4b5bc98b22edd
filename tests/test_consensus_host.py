"""CPU tests of the product's host-side contig code (nanospring_amd/csrc/consensus.cpp: consensus DAG,
main-path maintenance, cycle pruning, edit emission, the seven streams) driven by a plain sequential
restatement of the reference's -t 1 loop with the CPU oracles (tests/host_harness.cpp):
  - the reference's own CHECKS invariants hold after every graph update (checkRead / checkNoCycle),
  - the streams decode -- with an independent Python restatement of Decompressor::generateRead -- to
    exactly the input reads (the reference's only pinned property, util/test_script.sh:7-9)."""
import numpy as np

import nanospring_amd as ns
from tests import host_lib
from tests.stream_decode import decode, fold


def reads_of(bases, off):
    b = bytes(bases)
    return [b[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


def check_roundtrip(streams, reads):
    got = decode(streams)
    assert len(got) == len(reads)
    for i, r in enumerate(reads):
        assert got[i] == fold(r), i


def test_synthetic_reads_round_trip_and_graph_invariants():
    bases, off = ns.synth_reads(5, 40000, 160, 2500.0)
    out, st = host_lib.consensus(bases, off, ns.mt19937_64_salts(60))
    assert st["n_bad_roundtrip"] == 0 and st["n_graph_check_fail"] == 0
    assert st["count_aligner"] > 100 and st["n_contigs"] < 40
    check_roundtrip(out, reads_of(bases, off))
    md = out["metaData"].decode().splitlines()
    assert md[0] == "numReads=160" and md[2] == "numThr=1" and md[1] == "numContigs=%d" % st["n_contigs"]
    counts = [int(x) for x in md[3].split("=")[1].split(":") if x]
    assert sum(counts) == 160 and len(counts) == st["n_contigs"]
    assert len(out["id"]) == 4 * 160


def test_edge_cases_short_repetitive_duplicate_and_n_reads():
    rng = np.random.RandomState(2)
    g = "".join("ACGT"[i] for i in rng.randint(0, 4, size=6000))
    reads = [g[0:3000], g[1000:4000], g[2000:5500], g[500:2500], "A" * 400, "ACGT", "", g[100:131], g[100:132], g[0:3000], g[3000:6000][::-1],
             g[1500:3500].replace("A", "N", 5), "AC" * 300, g[4000:6000]]
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    out, st = host_lib.consensus(bases, off, ns.mt19937_64_salts(60))
    assert st["n_bad_roundtrip"] == 0 and st["n_graph_check_fail"] == 0
    check_roundtrip(out, [r.encode() for r in reads])
    assert st["n_lone"] >= 4          # empty, 4-mer, homopolymer, dinucleotide repeat


def test_edge_threshold_cuts_contigs():
    bases, off = ns.synth_reads(9, 30000, 120, 2500.0)
    out, st = host_lib.consensus(bases, off, ns.mt19937_64_salts(60), edge_thr=20000, checks=False)
    out2, st2 = host_lib.consensus(bases, off, ns.mt19937_64_salts(60), checks=False)
    assert st["n_bad_roundtrip"] == 0 and st["n_contigs"] > st2["n_contigs"]
    check_roundtrip(out, reads_of(bases, off))
