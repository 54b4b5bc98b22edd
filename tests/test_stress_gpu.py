"""The hand-over of DP results per alignment (csrc/ksw_collect.hpp: the DP kernels write into pinned memory while they run, the host believes a
status word once its check word adds up) is timing-dependent by construction.  What guards it here: repeated runs of the contig stage in the
bench's kind of schedule, under different host timings -- the default, no spinning in the pool (workers block at once), four host threads (every
thread watches many builders) -- must all be lossless and give ONE stream hash per input.  Every configuration is a fresh child process (the
thread count is read once per process; a child that dies fails the test, nothing is ever re-executed in a process that holds the GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(genome, reads, builders, runs, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_worker.py"), genome, str(reads), str(builders), str(runs)],
                       capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, "stress child failed (%s %s):\n%s\n%s" % (genome, env_extra, r.stdout[-2000:], r.stderr[-4000:])
    rows = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("RUN ")]
    assert len(rows) == runs
    for row in rows:
        assert int(row[3]) == 0, "reads that do not decode: %s" % row
    return [row[4] for row in rows], [int(row[2]) for row in rows]


@pytest.mark.gpu
@pytest.mark.parametrize("genome,reads", [("iid", 24000), ("repeats", 10000)])
def test_repeated_runs_under_different_host_timings_are_lossless_and_identical(genome, reads):
    hashes, contigs = run_child(genome, reads, 80, 3, {})
    assert len(set(hashes)) == 1, "three runs of one process differ: %s" % hashes
    for extra in ({"NSGPU_POOL_SPIN_US": "0"}, {"NSGPU_THREADS": "4"}):
        h2, c2 = run_child(genome, reads, 80, 1, extra)
        assert h2[0] == hashes[0] and c2[0] == contigs[0], "%s changes the streams (%s vs %s)" % (extra, h2[0][:16], hashes[0][:16])
