#!/usr/bin/env python3
"""Compiles the translation units of the library (mel_spec_amd/build.py: SOURCES, UNIT_FLAGS) with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs / SGPRs / scratch bytes / occupancy.  Usage: tools/kernel_resources.py [substring filter] [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
sys.path.insert(0, ROOT)
from mel_spec_amd.build import SOURCES, UNIT_FLAGS          # every translation unit with the flags the library build gives it
err = ""
for src in SOURCES:
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-c", "-Rpass-analysis=kernel-resource-usage",
           os.path.join(ROOT, "mel_spec_amd", "csrc", src), "-o", "/dev/null"] + UNIT_FLAGS.get(src, []) + sys.argv[2:]
    err += subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = dem.replace("melspec::", "").replace("LensIntervalStatic", "LensI").replace("void ", "")
    dem = re.sub(r"\(.*\)$", "", dem)
    if flt and flt not in dem:
        continue
    print(f"{dem[:110]:110s} VGPR {r.get('VGPRs', -1):4d}  SGPR {r.get('TotalSGPRs', -1):4d}  scratch {r.get('ScratchSize [bytes/lane]', -1):4d}"
          f"  spill {r.get('VGPRs Spill', -1):3d}  occ {r.get('Occupancy [waves/SIMD]', -1)}")
