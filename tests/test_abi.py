"""CPU-side checks of the drop-in boundary: libnsgpu.so loads, exports every
symbol include/nsgpu.h declares, and fails loudly (no CPU fallback) without a GPU."""
import os
import re

import numpy as np
import pytest

import nanospring_amd as ns
from nanospring_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    out = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            out |= set(re.findall(r"\b(nsgpu_[a-z0-9_]+)\s*\(", txt))
    return out


def test_library_exports_every_declared_symbol():
    lib = ns.load_library()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libnsgpu.so does not export {s}"
    # and the Python binding covers the whole header
    assert syms == set(_lib.SIGNATURES), syms ^ set(_lib.SIGNATURES)


def test_default_params_are_the_reference_defaults():
    lib = ns.load_library()
    p = ns.Params()
    lib.nsgpu_default_params(p)
    # src/main.cpp:46-78
    assert (p.k, p.n, p.overlap_sketch_thr, p.m_k, p.m_w, p.max_chain_iter, p.edge_threshold) == (23, 60, 6, 20, 50, 400, 4000000)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ns.NsGpuError, match="no CPU fallback"):
        ns.NsGpu()


def test_synth_reads_model():
    bases, off = ns.synth_reads(7, 500000, 400, 8000.0)
    lens = np.diff(off.astype(np.int64))
    assert lens.min() >= 400 and 6000 < lens.mean() < 10000
    assert set(np.unique(bases)) <= set(b"ACGT")
    b2, o2 = ns.synth_reads(7, 500000, 400, 8000.0)
    assert np.array_equal(bases, b2) and np.array_equal(off, o2)      # deterministic
    b3, _ = ns.synth_reads(8, 500000, 400, 8000.0)
    assert not np.array_equal(bases[:1000], b3[:1000])


def test_header_is_plain_c_and_cxx11(tmp_path):
    """The boundary is a C ABI: include/nsgpu.h must compile as C99 and as C++11 (the reference's language level) with
    warnings on, nothing but standard headers."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "nsgpu.h"\nint main(void) { nsgpu_params p; nsgpu_default_params(&p); return (int)p.k; }\n')
    inc = os.path.join(root, "include")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror"], ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-x", "c++"]):
        r = subprocess.run(cmd + ["-I", inc, "-c", str(src), "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
