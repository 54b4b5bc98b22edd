#!/usr/bin/env python3
"""SURVEY 8(f4), measured: what the back-end coders cost behind the GPU hot path, and what the builder count costs AFTER them.

The drop-in keeps the reference's coders (BSC -b48 -p -e2 per stream file, LZMA2 preset 6 for .base; src/Compressor.cpp:111-143):
they are the reference's own code, built by oracle/Makefile into oracle/_ref/backendref and only MEASURED here.  For one contig-stage
run (cfg2 shape) per builder count the script writes the stream files the way Compressor expects them and times
  (a) the reference's schedule: extensions one after the other, `numThr` files of an extension in parallel;
  (b) all files of all extensions in one pool over the host cores (a two-line change on the reference side);
and reports bytes per base before / after the coders.

    python tools/backend_coders.py [reads = 100000] [threads_out = 16] [builders ...= 1024 256 64]
"""
import json, os, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanospring_amd as ns
from nanospring_amd.filter import STREAMS
import bench

BIN = os.path.join(ROOT, "oracle", "_ref", "backendref")
BIN_GPU = os.path.join(ROOT, "oracle", "_ref", "backendref_gpu")      # the same front end and coders, libbsc's block sorter served by nsgpu_bwt_block (oracle/bwt_gpu_binding.cpp)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_out = int(sys.argv[2]) if len(sys.argv) > 2 else 16
builders = [int(x) for x in sys.argv[3:]] or [1024, 256, 64]
cores = bench.host_cores()
bases, off = ns.synth_reads(11, int(n_reads * 8000 / 20), n_reads, 8000.0)
n_bases = int(off[-1])
g = ns.NsGpu()
g.load_reads((bases, off))
g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
g.build_index()


def code(path, ext):
    t = time.perf_counter()
    subprocess.run([BIN, "lzma2" if ext == "base" else "bsc", path, path + "Compressed"], check=True, capture_output=True)
    return time.perf_counter() - t


res = {"workload": f"cfg2 shape, {n_reads} reads / {n_bases / 1e6:.0f} Mbases, {n_out} output thread sets", "host_cores": cores, "cpu": bench.cpu_model(), "runs": []}
for nb in builders:
    t0 = time.perf_counter()
    st = ns.consensus_run(g, nb, n_out)
    t_stage = time.perf_counter() - t0
    with tempfile.TemporaryDirectory() as td:
        ns.consensus_write(g, td + "/", "Stream")
        files = {ext: [f"{td}/Stream.tid.{t}.{ext}" for t in range(n_out)] for ext in STREAMS}
        raw = {ext: sum(os.path.getsize(f) for f in fs) for ext, fs in files.items()}
        # (a) the reference's loop: per extension, numThr files in parallel
        t0 = time.perf_counter()
        cpu = {}
        for ext, fs in files.items():
            with ThreadPoolExecutor(max_workers=min(n_out, cores)) as ex:
                cpu[ext] = sum(ex.map(lambda f: code(f, ext), fs))
        wall_ref = time.perf_counter() - t0
        comp = {ext: sum(os.path.getsize(f + "Compressed") for f in fs) for ext, fs in files.items()}
        for fs in files.values():
            for f in fs:
                os.remove(f + "Compressed")
        # (b) one pool over everything, longest files first
        jobs = sorted(((os.path.getsize(f), f, ext) for ext, fs in files.items() for f in fs), reverse=True)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(lambda j: code(j[1], j[2]), jobs))
        wall_pool = time.perf_counter() - t0
        # (c) the BSC files once more, one after the other, with the reference's sorter and with the GPU's (round 5: the .bsc files are
        # byte-identical, tests/test_bwt_gpu.py): CPU seconds of the whole coder with either, and the device's share
        gpu_sorter = None
        if os.path.exists(BIN_GPU):
            import re
            bsc_files = [f for ext, fs in files.items() if ext != "base" for f in fs]
            t0 = time.perf_counter()
            for f in bsc_files:
                subprocess.run([BIN, "bsc", f, f + "Compressed"], check=True, capture_output=True)
            cpu_ref = time.perf_counter() - t0
            t0 = time.perf_counter()
            dev_ms, n_blocks, startup = 0.0, 0, 0.0
            for f in bsc_files:
                r = subprocess.run([BIN_GPU, "bsc", f, f + "CompressedGpu"], check=True, capture_output=True, text=True)
                m = re.search(r"(\d+) blocks, (\d+) bytes through nsgpu_bwt_block, ([0-9.]+) ms", r.stderr)
                if m:
                    n_blocks += int(m.group(1)); dev_ms += float(m.group(3))
                assert open(f + "Compressed", "rb").read() == open(f + "CompressedGpu", "rb").read()
            wall_gpu = time.perf_counter() - t0
            t0 = time.perf_counter()
            subprocess.run([BIN_GPU, "bsc", "/dev/null", td + "/null.bsc"], capture_output=True)
            startup = time.perf_counter() - t0
            gpu_sorter = {"bsc_files": len(bsc_files), "serial_s_reference_sorter": round(cpu_ref, 2), "serial_s_gpu_sorter": round(wall_gpu, 2),
                          "of_it_process_start_and_context_s": round(startup * len(bsc_files), 2), "blocks": n_blocks, "device_ms_in_the_sorter": round(dev_ms, 1),
                          "files_identical": True}
    res["runs"].append({"builders": nb, "contig_stage_s": round(t_stage, 2), "contigs": st["n_contigs"], "lone_reads": st["n_lone"],
                        "raw_bytes": raw, "coded_bytes": comp, "raw_bytes_per_base": round(sum(raw.values()) / n_bases, 4),
                        "coded_bytes_per_base": round(sum(comp.values()) / n_bases, 4), "coder_cpu_s": {k: round(v, 2) for k, v in cpu.items()},
                        "coder_cpu_s_total": round(sum(cpu.values()), 2), "coder_rate_MB_per_core_s": round(sum(raw.values()) / 1e6 / sum(cpu.values()), 2),
                        "wall_reference_schedule_s": round(wall_ref, 2), "wall_one_pool_s": round(wall_pool, 2), "gpu_block_sorter": gpu_sorter})
    print(json.dumps(res["runs"][-1]), flush=True)
g.close()
print(json.dumps(res))
