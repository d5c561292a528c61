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
# translation units (one kernel family each, csrc/host_common.hpp has the map) and the flags only they get: melspec_runs.hip holds the
# run-per-wave f32 Whisper kernels, scheduled for ILP (csrc/melspec_runs.hip says why; the default strategy is the better one for
# everything else)
SOURCES = ["host_api.hip", "whisper400.hip", "fbank512.hip", "pow2.hip", "aux.hip", "melspec_runs.hip"]
UNIT_FLAGS = {"melspec_runs.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _inputs():
    # (the flags of every unit are part of what the library is: hashed through this file's own text would be too broad, so they are
    # appended to the hash explicitly in source_hash)
    return ([os.path.join(CSRC, s) for s in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc")))
            + sorted(glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h"))))


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_hash() -> str:
    """sha256 over the sources the library is built from (csrc/*.hip, csrc/*.hpp, include/*.h); baked into the library at build time
    (-DMELSPEC_SOURCE_HASH) and returned by melspec_source_hash(), so a stale prebuilt library is detected whatever its mtime."""
    import hashlib
    h = hashlib.sha256()
    for f in _inputs():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(repr(sorted(UNIT_FLAGS.items())).encode())
    return h.hexdigest()[:32]


HASH_MARKER = b"@melspec-source-hash:"


def built_hash(path: str = LIB_PATH):
    """The source hash a built library carries (what its melspec_source_hash() returns), read from the file without loading it -- the
    caller may be about to overwrite it; None if the file has none."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(HASH_MARKER)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(HASH_MARKER):j].decode(errors="replace")


def needs_build() -> bool:
    return not os.path.exists(LIB_PATH) or built_hash() != source_hash()


INFO_PATH = os.path.join(PKG_DIR, "build_info.json")


def build(force: bool = False, verbose: bool = False, lab: bool = False, defines=()) -> str:
    """Compile the HIP kernels + C ABI for gfx950. Returns the path of the shared library.
    lab=True builds libmelspec_hip_lab.so with the tuning switches of tools/ compiled in (load it with MELSPEC_LIB)."""
    out = LAB_LIB_PATH if lab else LIB_PATH
    if not lab and not force and not needs_build():
        return LIB_PATH
    import json, time
    src_hash = source_hash()
    common = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f'-DMELSPEC_SOURCE_HASH="{src_hash}"'] + (["-DMELSPEC_LAB"] if lab else []) + [f"-D{d}" for d in defines] + [
              # no SLP packing: v_pk_*_f32 issues at half the rate of the plain op on gfx950 (measured,
              # tools/valu_rate.hip) and pairing registers costs ~250 v_mov per kernel
              "-fno-slp-vectorize",
              "-Wall", "-Wno-unused-function"]
    import tempfile
    t0 = time.time()
    cmds = []
    with tempfile.TemporaryDirectory(prefix="melspec_build_") as tmp:
        objs, procs = [], []
        for src in SOURCES:                                   # one object per translation unit, compiled side by side
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            c = common + UNIT_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
            cmds.append(" ".join(c))
            if verbose:
                print(cmds[-1])
            objs.append(obj)
            procs.append(subprocess.Popen(c))
        if any(p.wait() != 0 for p in procs):
            raise subprocess.CalledProcessError(1, cmds[0])
        link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        cmds.append(" ".join(link[:6] + [os.path.basename(o) for o in objs]))
        if verbose:
            print(cmds[-1])
        subprocess.check_call(link)
    cmd = [common[0]]
    if not lab:
        # what was compiled, by what, where: printed by __graft_entry__.smoke() so that the record of a run shows which build it used
        import hashlib, platform
        try:
            ver = subprocess.run([cmd[0], "--version"], capture_output=True, text=True).stdout.strip().splitlines()[0]
        except Exception:
            ver = "?"
        with open(out, "rb") as fh:
            so_hash = hashlib.sha256(fh.read()).hexdigest()[:32]
        with open(INFO_PATH, "w") as fh:
            json.dump({"library": os.path.basename(out), "source_hash": src_hash, "library_sha256_32": so_hash, "hipcc": ver,
                       "seconds": round(time.time() - t0, 1), "host": platform.node(), "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                       "gpu_visible_at_build": os.path.exists("/dev/kfd"), "commands": cmds}, fh, indent=1)
    return out


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, lab="--lab" in sys.argv))
