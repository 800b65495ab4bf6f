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
FLAGS = ["--offload-arch=gfx950", "-I/opt/rocm/include", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast-honor-pragmas", "-munsafe-fp-atomics"]


def newest_header():
    hs = glob.glob(os.path.join(SRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def compile_one(src, force, objdir=None, extra=()):
    obj = os.path.join(objdir or OBJ, os.path.basename(src) + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header()):
        return obj, False
    cmd = [HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed on " + src)
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, jobs=4, verbose=True, ablation=False):
    """ablation=True builds libcsx_ablation.so with -DCSX_ABLATION: the timing variants of the kernels (some
    compute wrong results on purpose) exist only there and are selected by environment variables only there.
    It is never loaded unless CSX_LIB points at it (profiles/*_ablation.md say how each number was taken)."""
    objdir = OBJ + "_ablation" if ablation else OBJ
    lib = os.path.join(HERE, "libcsx_ablation.so") if ablation else LIB
    extra = ("-DCSX_ABLATION",) if ablation else ()
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(SRC, "*.hip")) + glob.glob(os.path.join(SRC, "*.cpp")))
    objs, rebuilt = [], False
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for obj, did in ex.map(lambda s: compile_one(s, force, objdir, extra), srcs):
            objs.append(obj)
            rebuilt |= did
    if rebuilt or not os.path.exists(lib):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        subprocess.check_call(cmd)
        if verbose:
            print("built", lib)
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--ablation", action="store_true", help="build libcsx_ablation.so (-DCSX_ABLATION) instead")
    a = ap.parse_args()
    build(a.force, a.jobs, ablation=a.ablation)
