"""Builds libnsgpu.so (hand-written HIP for gfx950 + the C-ABI) in-tree.

    python -m nanospring_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so stays in the tree
(nanospring_amd/lib/) so that it travels with gpurun snapshots.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libnsgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# no fast-math anywhere: chaining and the repetitive test compare floats/doubles
# exactly as the reference does.
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-result",
         "-ffp-contract=off", "-fno-fast-math", "-I/opt/rocm/include"] + os.environ.get("NSGPU_HOST_FLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(HERE, "..", "include", "nsgpu.h"))
    jobs = []
    objs = []
    for f in _sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJDIR, f + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if warn and verbose:
                    sys.stderr.write(warn)
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lpthread", "-lz"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
