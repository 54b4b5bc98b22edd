"""SURVEY 8 f4, the block sorter on the GPU (nsgpu_bwt_block, csrc/bwt.hip) against the bytes the reference's own libbsc produces for
the same block (oracle/_ref/backendref bwt = bsc_bwt_encode, libbsc/bwt/bwt.cpp:46-79, the call bsc::BSC_compress makes per 48 MB
block of a stream file, src/bsc.cpp:1045-1057): BWT bytes, primary index and the auxiliary indexes, bit-exact."""
import numpy as np
import pytest

import nanospring_amd as ns
from nanospring_amd.filter import STREAMS
from tests import oracle_lib

pytestmark = pytest.mark.gpu


def check(g, t):
    b, idx, aux, ms, rounds = ns.bwt_block(g, t)
    if len(t) >= 16:
        wb, widx, waux = oracle_lib.ref_bwt(t)
    else:
        wb, widx, waux = oracle_lib.bwt_naive(t)
    assert idx == widx and aux == waux and b == wb, (len(t), idx, widx)
    return ms, rounds


def test_small_and_edge_blocks():
    """1 .. 40 bytes (below 16 libsais rejects bsc's sampling rate: the restatement is the checker there), runs of one byte (every suffix a
    prefix of the next: log2(n) doubling rounds), periodic text, all 256 byte values, newline-separated records"""
    g = ns.NsGpu()
    rng = np.random.RandomState(9)
    assert ns.bwt_block(g, b"")[:3] == (b"", 0, [])
    for n in list(range(1, 41)):
        check(g, bytes(rng.choice(list(b"AC"), n).tolist()))
    check(g, bytes(70000))
    check(g, bytes([255]) * 5000 + bytes([0]) * 5000 + bytes([255]) * 5000)
    check(g, b"ACGT" * 40000)
    check(g, bytes(rng.randint(0, 256, 300000).astype(np.uint8).tolist()))
    check(g, (b"ACGTTGCA" * 1000 + b"\n") * 30)
    _, rounds = check(g, bytes(rng.choice(list(b"ACGT"), 1 << 20).tolist()))
    assert rounds <= 4                               # random DNA: no repeat beyond ~2 log4(n) = 20 characters
    g.close()


def test_stream_files_of_a_contig_stage():
    """the blocks the back end really sees: the seven stream files of a contig stage (genome text, positions, edit types, bases ...)"""
    bases, off = ns.synth_reads(7, 300000, 800, 6000.0)
    g = ns.NsGpu()
    g.load_reads((bases, off))
    g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
    g.build_index()
    ns.consensus_run(g, 16, 1)
    n = 0
    for k in STREAMS:
        s = ns.consensus_stream(g, 0, k)
        if len(s) >= 16:
            check(g, s)
            n += 1
    assert n >= 6
    g.close()


def test_full_size_block():
    """48 MB, the block size bsc::BSC_compress cuts stream files into (src/bsc.cpp:1045): consensus-like text -- random ACGT with 4 kb
    duplications and a long homopolymer run -- and its time on the device"""
    rng = np.random.RandomState(3)
    n = 48 << 20
    t = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n)
    for _ in range(200):
        a, b = rng.randint(0, n - 4096, 2)
        t[b:b + 4096] = t[a:a + 4096]
    t[1000000:1003000] = ord("A")
    t[::9973] = ord("\n")
    g = ns.NsGpu()
    ms, rounds = check(g, t)
    print("bwt of a 48 MB block: %.1f ms on the device (%.2f GB/s of input), %d doubling rounds" % (ms, n / ms / 1e6, rounds))
    assert rounds <= 12
    g.close()


def test_bsc_files_with_the_gpu_block_sorter_are_the_reference_s_files(tmp_path):
    """The drop-in proved on whole files: oracle/_ref/backendref_gpu is the reference's own BSC front end and libbsc (src/bsc.cpp,
    libbsc/libbsc/libbsc.cpp: detectors, LZP off under -p, QLFC) with ONE function replaced -- bsc_bwt_encode, by the binding INTEGRATION.md
    section 3b shows over nsgpu_bwt_block (oracle/bwt_gpu_binding.cpp).  The .bsc file of every stream of a contig stage, and of a
    consensus-like file of several 48 MB blocks, must equal the file the unmodified reference writes (oracle/_ref/backendref) byte for byte and
    decode with the reference's decoder to the input."""
    import os, subprocess
    ref = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "backendref")
    gpu = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "backendref_gpu")
    assert os.path.exists(ref) and os.path.exists(gpu), "oracle/_ref/backendref(_gpu) missing: run `make -C oracle ref` where /root/reference exists"
    bases, off = ns.synth_reads(7, 600000, 1600, 6000.0)
    g = ns.NsGpu()
    g.load_reads((bases, off))
    g.sketch(ns.mt19937_64_salts(60, 12345), fetch=False)
    g.build_index()
    ns.consensus_run(g, 16, 1)
    files = {k: ns.consensus_stream(g, 0, k) for k in STREAMS}
    g.close()
    rng = np.random.RandomState(5)
    n = 110 << 20                                   # 48 + 48 + 14 MB: three blocks
    t = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n)
    for _ in range(300):
        a, b = rng.randint(0, n - 4096, 2)
        t[b:b + 4096] = t[a:a + 4096]
    t[::9973] = ord("\n")
    files["three_blocks"] = t.tobytes()
    files["tiny"] = b"ACGTACGTAC"                   # below 16 bytes: bsc stores the block, the sorter is never asked
    files["empty"] = b""
    n_sorted = 0
    for name, data in files.items():
        src = tmp_path / name
        src.write_bytes(data)
        a, b, back = str(src) + ".ref.bsc", str(src) + ".gpu.bsc", str(src) + ".back"
        subprocess.run([ref, "bsc", str(src), a], check=True, capture_output=True)
        r = subprocess.run([gpu, "bsc", str(src), b], check=True, capture_output=True, text=True)
        assert open(a, "rb").read() == open(b, "rb").read(), "%s: the .bsc file written with the GPU block sorter differs from the reference's" % name
        if "blocks" in r.stderr and len(data) >= 16:
            n_sorted += 1
        subprocess.run([ref, "unbsc", b, back], check=True, capture_output=True)
        assert open(back, "rb").read() == data, "%s: does not decode to the input" % name
        for f in (a, b, back):
            os.remove(f)
    assert n_sorted >= 6, "the GPU block sorter was not the one that ran"
