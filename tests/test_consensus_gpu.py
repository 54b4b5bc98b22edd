"""GPU tests of the contig stage: nsgpu_consensus_run (virtual
builders in lock-step, window queries and alignments batched on the GPU).
  - ONE builder must produce byte-identical streams to the plain sequential restatement of the
    reference's -t 1 loop over the CPU oracles (tests/host_harness.cpp);
  - any number of builders: lossless (library decoder and the independent Python decoder), deterministic;
  - file names / metaData as Compressor::compress expects them."""
import os

import numpy as np
import pytest

import nanospring_amd as ns
from tests import host_lib
from tests.stream_decode import decode, fold
from nanospring_amd.filter import STREAMS

pytestmark = pytest.mark.gpu


def run(bases, off, n_builders, n_out=1, **kw):
    g = ns.NsGpu(**kw)
    g.load_reads((bases, off))
    g.sketch(ns.mt19937_64_salts(60), fetch=False)
    g.build_index()
    st = ns.consensus_run(g, n_builders, n_out)
    streams = [{k: ns.consensus_stream(g, t, k) for k in STREAMS} for t in range(n_out)]
    md = ns.consensus_stream(g, 0, "metaData")
    return g, st, streams, md


def test_one_builder_equals_sequential_reference_loop():
    bases, off = ns.synth_reads(5, 40000, 160, 2500.0)
    want, wst = host_lib.consensus(bases, off, ns.mt19937_64_salts(60), checks=False)
    g, st, streams, md = run(bases, off, 1)
    for k in STREAMS:
        assert streams[0][k] == want[k], k
    assert md == want["metaData"]
    for a, b in (("count_minhash", "count_minhash"), ("count_minhash_not_in_graph", "count_minhash_not_in_graph"), ("count_aligner", "count_aligner"),
                 ("n_contigs", "n_contigs"), ("n_lone", "n_lone"), ("n_align_calls", "n_align_calls")):
        assert st[a] == wst[b], a
    assert ns.consensus_verify(g) == 0
    g.close()


def test_cfg1_one_builder_equals_sequential_reference_loop():
    """BASELINE configs[0] shape (the reference's CPU-runnable plumbing case: 1 235 reads of mean 8 kb, 20x of a 0.5 Mb
    genome, -k 23 -n 60 -t 1): the GPU path with one builder must give the streams of the sequential -t 1 restatement
    byte for byte -- with the reference's own minimap2 answering the alignments there when its object travelled."""
    from tests import oracle_lib
    bases, off = ns.synth_reads(7, 500000, 1235, 8000.0)
    want, wst = host_lib.consensus(bases, off, ns.mt19937_64_salts(60), checks=False, ref_aligner=oracle_lib.mm2ref() is not None)
    assert wst["n_bad_roundtrip"] == 0
    g, st, streams, md = run(bases, off, 1)
    for k in STREAMS:
        assert streams[0][k] == want[k], k
    assert md == want["metaData"]
    assert st["count_aligner"] == wst["count_aligner"] > 1000 and st["n_contigs"] == wst["n_contigs"]
    assert ns.consensus_verify(g) == 0
    g.close()


@pytest.mark.parametrize("n_builders,n_out", [(16, 1), (64, 3)])
def test_many_builders_lossless_and_deterministic(n_builders, n_out):
    bases, off = ns.synth_reads(31, 150000, 600, 4000.0)
    g, st, streams, md = run(bases, off, n_builders, n_out)
    assert ns.consensus_verify(g) == 0
    b = bytes(bases)
    got = {}
    for s in streams:
        d = decode(s)
        assert not (set(d) & set(got))
        got.update(d)
    assert len(got) == 600
    for i in range(600):
        assert got[i] == b[int(off[i]):int(off[i + 1])]
    assert st["count_aligner"] > 400 and st["n_rounds"] < 400
    lines = md.decode().splitlines()
    assert lines[0] == "numReads=600" and lines[2] == "numThr=%d" % n_out
    assert sum(int(x) for x in lines[3].split("=")[1].split(":") if x) == 600
    g.close()
    g2, st2, streams2, md2 = run(bases, off, n_builders, n_out)
    assert streams2 == streams and md2 == md
    g2.close()


def test_edge_cases_and_files(tmp_path):
    rng = np.random.RandomState(2)
    gs = "".join("ACGT"[i] for i in rng.randint(0, 4, size=6000))
    reads = [gs[0:3000], gs[1000:4000], gs[2000:5500], gs[500:2500], "A" * 400, "ACGT", "", gs[100:131], gs[100:132], gs[0:3000],
             gs[1500:3500].replace("A", "N", 5), "AC" * 300, gs[4000:6000]]
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    want, _ = host_lib.consensus(bases, off, ns.mt19937_64_salts(60), checks=False)
    g, st, streams, md = run(bases, off, 1)
    for k in STREAMS:
        assert streams[0][k] == want[k], k
    assert ns.consensus_verify(g) == 0
    d = str(tmp_path) + "/"
    ns.consensus_write(g, d, "Stream")
    for ext in STREAMS:
        assert open(d + "Stream.tid.0." + ext, "rb").read() == streams[0][ext]
    assert open(d + "metaData", "rb").read() == md
    g.close()
    g4, st4, streams4, _ = run(bases, off, 4, 2)
    assert ns.consensus_verify(g4) == 0
    got = {}
    for s in streams4:
        got.update(decode(s))
    assert [got[i] for i in range(len(reads))] == [fold(r.encode()) for r in reads]
    g4.close()


def test_shard_engine_with_global_id_base():
    """The per-rank engine of nanospring_amd/dist.py: stream ids are global (id base of the shard)."""
    from nanospring_amd import dist as nd
    bases, off = ns.synth_reads(8, 60000, 240, 3000.0)
    lo, hi = nd.shard_bounds(off, 2)[1]
    sb, so = nd.take_shard(bases, off, lo, hi)
    streams, md, st = nd.gpu_engine(n_builders=8)(sb, so, lo, 2)
    assert st["bad"] == 0
    got = {}
    for s in streams:
        got.update(decode(s))
    assert sorted(got) == list(range(lo, hi))
    b = bytes(bases)
    for i in range(lo, hi):
        assert got[i] == b[int(off[i]):int(off[i + 1])]
    assert nd.parse_meta(md)["numReads"] == hi - lo


def test_repeat_rich_genome_many_builders_lossless():
    """A genome with an exact 4 kb duplication, tandem repeats and reads of both strands: the consensus graphs get cycles
    to prune and paths to split (removeCycles / splitPath) all the time; any number of builders must stay lossless and
    deterministic."""
    from tests.align_cases import make_genome, mutate, revcomp
    rng = np.random.RandomState(12)
    g0 = make_genome(rng, 30000)
    g0 = g0 + g0[5000:9000] + make_genome(rng, 15000) + "ACGGT" * 300 + make_genome(rng, 8000)
    reads = []
    for _ in range(320):
        ln = int(max(400, rng.gamma(2.0, 1500.0)))
        st = rng.randint(0, max(1, len(g0) - ln))
        s = mutate(rng, g0[st:st + ln], 0.04)
        reads.append(revcomp(s) if rng.randint(2) else s)
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    outs = []
    for n_builders in (1, 7, 48):
        g, st, streams, md = run(bases, off, n_builders, 2)
        assert ns.consensus_verify(g) == 0, n_builders
        got = {}
        for s2 in streams:
            got.update(decode(s2))
        assert len(got) == len(reads)
        for i, r in enumerate(reads):
            assert got[i] == r.encode(), (n_builders, i)
        assert st["count_aligner"] > 150
        outs.append((streams, md))
        g.close()
    g, st, streams, md = run(bases, off, 48, 2)
    assert (streams, md) == outs[2]
    g.close()
