"""The C-ABI used the way INTEGRATION.md section 1 shows it: a C++11 program (tests/integration/compress_stage.cpp, no
Python, no ctypes) links libnsgpu.so, reads a FASTQ file, runs the stage and leaves the file set that the reference's
BSC/LZMA2 loop and Decompressor expect (Stream.tid.<i>.{genome,lone,id,pos,type,base,complement} + metaData); the files
are then decoded with the independent Python restatement of Decompressor::generateRead."""
import os
import subprocess

import numpy as np
import pytest

import nanospring_amd as ns
from tests.stream_decode import decode

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx11_adaptor_writes_the_reference_file_set(tmp_path):
    exe = tmp_path / "compress_stage"
    lib_dir = os.path.dirname(ns.lib_path())
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "integration", "compress_stage.cpp"),
                        "-o", str(exe), "-L", lib_dir, "-lnsgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    bases, off = ns.synth_reads(21, 200000, 500, 6000.0)
    b = bytes(bases)
    reads = [b[int(off[i]):int(off[i + 1])] for i in range(500)]
    fq = tmp_path / "reads.fastq"
    with open(fq, "wb") as f:
        for i, s in enumerate(reads):
            f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
    out = tmp_path / "tmp"
    out.mkdir()
    num_thr = 3
    r = subprocess.run([str(exe), str(fq), str(out) + "/", str(num_thr)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "numReads 500" in r.stdout and "lossless check: 0 bad reads" in r.stdout
    names = sorted(os.listdir(out))
    want = sorted(["metaData"] + ["Stream.tid.%d.%s" % (t, e) for t in range(num_thr) for e in ("genome", "lone", "id", "pos", "type", "base", "complement")])
    assert names == want
    md = open(out / "metaData", "rb").read().decode().splitlines()
    assert md[0] == "numReads=500" and md[2] == "numThr=%d" % num_thr
    got = {}
    for t in range(num_thr):
        streams = {e: open(out / ("Stream.tid.%d.%s" % (t, e)), "rb").read() for e in ("genome", "lone", "id", "pos", "type", "base", "complement")}
        d = decode(streams)
        assert not set(d) & set(got)
        got.update(d)
    assert sorted(got) == list(range(500))
    for i in range(500):
        assert got[i] == reads[i]
