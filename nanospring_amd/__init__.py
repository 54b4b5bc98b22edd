"""nanospring_amd -- MI355X (gfx950) implementation of NanoSpring's read-clustering
+ reference-encode hot path.

The product is libnsgpu.so (hand-written HIP behind the C-ABI of include/nsgpu.h).
This package is the thin host-side mirror used by the tests and bench.py; it has
no CPU fallback: importing works anywhere, but every operator raises
NsGpuError when the library or a gfx950 device is missing.
"""
import os as _os

# the HIP runtime's stream -> hardware-queue multiplexing (default 4 queues) serialises the contig stage's ~30 streams; must be in
# the environment before the runtime initialises (csrc/api.hip nsgpu_create, profiles/r02_stream_priority_ab.txt)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import NsGpuError, lib_path, load_library, Params, Timing  # noqa: F401
from .filter import NsGpu, MinHashReadFilter, mt19937_64_salts, synth_reads, ksw_extd2_batch, align_batch, align_stats, consensus_run, consensus_stream, consensus_verify, consensus_write, set_schedule, set_defer, get_defer, bwt_block, bsc_aux_rate, set_graph, graph_stats, GRAPH_AUTO, GRAPH_HOST, GRAPH_DEVICE, GRAPH_CHECK  # noqa: F401

__all__ = ["NsGpuError", "lib_path", "load_library", "Params", "Timing", "NsGpu", "MinHashReadFilter",
           "mt19937_64_salts", "synth_reads", "ksw_extd2_batch", "align_batch", "align_stats", "consensus_run", "consensus_stream", "consensus_verify", "consensus_write", "set_schedule"]
