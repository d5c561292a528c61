"""Build libmelspec_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmelspec_hip.so")
SOURCES = ["melspec_hip.hip"]
HEADERS = ["melspec_kernels.hpp", "whisper_fast.hpp", "device_fft.hpp", "fast_tables.hpp", "tables.hpp",
           os.path.join("..", "..", "include", "melspec_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP kernels + C ABI for gfx950. Returns the path of the shared library."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           # no SLP packing: v_pk_*_f32 issues at half the rate of the plain op on gfx950 (measured,
           # tools/valu_rate.hip) and pairing registers costs ~250 v_mov per kernel
           "-fno-slp-vectorize",
           "-Wall", "-Wno-unused-function",
           "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
