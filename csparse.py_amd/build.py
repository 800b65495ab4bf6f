#!/usr/bin/env python3
"""Build libcsx.so (hand-written HIP for gfx950 + the C ABI) in-tree.

    python csparse.py_amd/build.py [--force] [--jobs N]

hipcc cross-compiles without a GPU.  Objects go to csparse.py_amd/build/, the
shared library to csparse.py_amd/libcsx.so (git-ignored; it travels to the GPU
box with the gpurun snapshot)."""
import argparse
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcsx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast-honor-pragmas", "-munsafe-fp-atomics"]


def newest_header():
    hs = glob.glob(os.path.join(SRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def compile_one(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header()):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed on " + src)
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(SRC, "*.hip")) + glob.glob(os.path.join(SRC, "*.cpp")))
    objs, rebuilt = [], False
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for obj, did in ex.map(lambda s: compile_one(s, force), srcs):
            objs.append(obj)
            rebuilt |= did
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    build(a.force, a.jobs)
