"""GPU tests of the consensus graphs in HBM (graph_dev.hip over dgraph.hpp -- updateGraph, calculateMainPathGreedy,
removeCycles / splitPath with one workgroup per accepted read, one launch per slot).
  - NSGPU_GRAPH_CHECK: every update is also run on the host by the SAME code with a team of one and the arrays are compared entry by entry
    (ids are handed out from prefix sums, so 512 threads and one thread must build identical arrays);
  - the streams must be the oracle's (one builder = -t 1; many builders = the lock-step oracle's virtual threads), and the same as with
    the pointer graph on the host, byte for byte -- also at BASELINE cfg2's full size against the recorded lock-step fixture."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import nanospring_amd as ns
from nanospring_amd.filter import STREAMS
from tests import test_consensus_gpu as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def device_graphs_checked(monkeypatch):
    monkeypatch.setenv("NSGPU_GRAPH", "device")
    monkeypatch.setenv("NSGPU_GRAPH_CHECK", "1")


def test_one_builder_in_hbm_equals_oracle_every_update_checked(device_graphs_checked):
    bases, off = ns.synth_reads(5, 40000, 160, 2500.0)
    T.one_builder_equals_oracle(bases, off)
    bases, off = ns.synth_reads(7, 200000, 500, 8000.0)              # cfg1's shape: 8 kb reads, contigs of dozens of reads, both phases of the walk
    T.one_builder_equals_oracle(bases, off)


@pytest.mark.parametrize("B,groups,depth,rings", [(40, 1, 0, 1), (24, 1, 2, 1), (40, 4, 3, 1)])
def test_many_builders_in_hbm_equal_lockstep_oracle_every_update_checked(device_graphs_checked, B, groups, depth, rings):
    bases, off = ns.synth_reads(31, 150000, 600, 4000.0)
    T.many_builders_equal_lockstep_oracle(bases, off, B, groups, depth, rings)


def test_workgroups_that_start_late_take_no_other_update(monkeypatch):
    """A launch whose workgroups come to life when the slot is over (NSGPU_GRAPH_LATE_START_US: its kernel queued behind others, as beside a
    second process on the GPU) finds the slot records re-armed for the NEXT update: the order word carries the ticket of the prepare() it
    belongs to, and a workgroup takes nothing else (without that: the same script applied twice -- reads that do not round-trip)."""
    monkeypatch.setenv("NSGPU_GRAPH", "device")
    monkeypatch.setenv("NSGPU_GRAPH_LATE_START_US", "2500")
    bases, off = ns.synth_reads(31, 150000, 600, 4000.0)
    T.many_builders_equal_lockstep_oracle(bases, off, 40, 1, 0, 1)


def test_remove_cycles_that_runs_out_of_room_is_run_again(device_graphs_checked, monkeypatch):
    """next to no spare room in the arrays (NSGPU_GRAPH_SLACK): removeCycles stops in front of every split whose copies do not fit -- nothing of it
    done, the graph whole --, the host grows the arrays and runs it again (dg_finish_kernel); every update checked against the host's arrays"""
    monkeypatch.setenv("NSGPU_GRAPH_SLACK", "48")
    bases, off = _repeats_reads()
    T.many_builders_equal_lockstep_oracle(bases, off, 20, 1, 2, 1)
    bases, off = ns.synth_reads(7, 200000, 500, 8000.0)
    T.one_builder_equals_oracle(bases, off)


def _repeats_reads():
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
    return T.pack(reads)


@pytest.mark.parametrize("flags", ["0", "1", "2", "4"])
def test_repeat_rich_genome_in_hbm_with_the_rare_branches(device_graphs_checked, monkeypatch, flags):
    """removeCycles / splitPath have work on this genome; the debug flags take the rare branches on every update
    (1: excursions one at a time, 2: the reference's full walk, 4: the path's left part moves)"""
    monkeypatch.setenv("NSGPU_SOA_DEBUG_FLAGS", flags)
    bases, off = _repeats_reads()
    T.many_builders_equal_lockstep_oracle(bases, off, 20, 1, 2, 1)
    st = None
    g, st, streams, md = T.run(bases, off, 1)
    gs = ns.graph_stats(g)
    assert gs["placement"] == "device" and gs["checked"] == 1 and gs["n_updates"] == st["count_aligner"] and gs["n_split_calls"] > 0
    if flags == "1":
        assert gs["n_sequential_updates"] > 0
    if flags == "2":
        assert gs["n_full_walks"] > 0
    g.close()


def test_both_placements_give_the_same_streams_and_the_api_reports_them():
    bases, off = ns.synth_reads(9, 300000, 900, 6000.0)
    out = {}
    for mode in (ns.GRAPH_HOST, ns.GRAPH_DEVICE):
        g = ns.NsGpu()
        g.load_reads((bases, off))
        g.sketch(ns.mt19937_64_salts(60), fetch=False)
        g.build_index()
        ns.set_graph(g, mode)
        st = ns.consensus_run(g, 48, 2, schedule=(1, 3, 2))
        h = hashlib.sha256()
        for t in range(2):
            for k in STREAMS:
                h.update(ns.consensus_stream(g, t, k))
        gs = ns.graph_stats(g)
        assert gs["placement"] == ("host" if mode == ns.GRAPH_HOST else "device")
        if mode == ns.GRAPH_DEVICE:
            assert gs["n_updates"] == st["count_aligner"] and 0 < gs["n_launches"] <= st["n_rounds"] * 2 + 8 and sum(gs["by_duration"]) == gs["n_updates"]
            assert gs["n_long_reports"] == 0 and gs["n_full_walks"] == 0
        assert ns.consensus_verify(g) == 0
        out[mode] = (h.hexdigest(), st["n_contigs"], st["count_aligner"])
        g.close()
    assert out[ns.GRAPH_HOST] == out[ns.GRAPH_DEVICE]


def test_cfg2_full_default_schedule_in_hbm_equals_lockstep_oracle_hashes(monkeypatch):
    """BASELINE cfg2 at full size, the bench's default schedule, consensus graphs in HBM: the recorded lock-step fixture (profiles/r03_lockstep_cfg2.json)"""
    monkeypatch.setenv("NSGPU_GRAPH", "device")
    T.test_cfg2_full_default_schedule_equals_lockstep_oracle_hashes("auto:r03_lockstep_cfg2.json")


def test_automatic_placement_follows_the_host_threads():
    """NSGPU_GRAPH unset: in HBM with at most 5 host threads (a rank of a shared node), on the host otherwise; same streams"""
    code = ("import sys, hashlib; sys.path.insert(0, %r); import nanospring_amd as ns; from nanospring_amd.filter import STREAMS\n"
            "bases, off = ns.synth_reads(5, 60000, 240, 2500.0)\n"
            "g = ns.NsGpu(); g.load_reads((bases, off)); g.sketch(ns.mt19937_64_salts(60), fetch=False); g.build_index()\n"
            "ns.consensus_run(g, 16, 1, schedule=(1, 2, 1))\n"
            "h = hashlib.sha256(); [h.update(ns.consensus_stream(g, 0, k)) for k in STREAMS]\n"
            "print('R', ns.graph_stats(g)['placement'], h.hexdigest())\n") % ROOT
    res = {}
    for thr in ("2", "8"):
        env = dict(os.environ, NSGPU_THREADS=thr)
        env.pop("NSGPU_GRAPH", None)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[thr] = [l for l in r.stdout.splitlines() if l.startswith("R ")][0].split()[1:]
    assert res["2"][0] == "device" and res["8"][0] == "host" and res["2"][1] == res["8"][1]
