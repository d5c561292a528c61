"""Build libmelspec_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmelspec_hip.so")
import glob

LAB_LIB_PATH = os.path.join(PKG_DIR, "libmelspec_hip_lab.so")   # -DMELSPEC_LAB: tuning switches for tools/, never loaded by default
SOURCES = ["melspec_hip.hip"]


def _inputs():
    return ([os.path.join(CSRC, s) for s in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.hpp")))
            + sorted(glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h"))))


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in _inputs())


def build(force: bool = False, verbose: bool = False, lab: bool = False, defines=()) -> str:
    """Compile the HIP kernels + C ABI for gfx950. Returns the path of the shared library.
    lab=True builds libmelspec_hip_lab.so with the tuning switches of tools/ compiled in (load it with MELSPEC_LIB)."""
    out = LAB_LIB_PATH if lab else LIB_PATH
    if not lab and not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + (["-DMELSPEC_LAB"] if lab else []) + [f"-D{d}" for d in defines] + [
           # no SLP packing: v_pk_*_f32 issues at half the rate of the plain op on gfx950 (measured,
           # tools/valu_rate.hip) and pairing registers costs ~250 v_mov per kernel
           "-fno-slp-vectorize",
           "-Wall", "-Wno-unused-function",
           "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, lab="--lab" in sys.argv))
