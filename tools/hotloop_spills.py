#!/usr/bin/env python3
"""Compiles every unit of the library (mel_spec_amd.build.SOURCES) to gfx950 assembly and reports, per kernel, the scratch (spill) instructions that sit INSIDE the unit
loop -- between the first `s_setprio 0` (phase 1 of a unit) and the loop's back edge -- as opposed to the recompute tail behind it.
A spill in the tail is cheap; one in the loop is paid per unit (the mel-major kernel once lost 20 % that way).
Kernels without phase priorities (pow2_frame_kernel, the normalisers, ...): every scratch instruction inside ANY loop, by the compiler's own
block annotations ("in Loop: Header=... Depth=n") -- reported as "loop" in lower case and not counted into the exit status: the n_fft = 2048
instances of pow2_frame_kernel are known to spill there, and the spill-free form of that kernel was measured 11-14 % slower
(profiles/r05_fb512_twelve_waves.txt)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "melspec_hotloop.s")
sys.path.insert(0, ROOT)
from mel_spec_amd.build import SOURCES, UNIT_FLAGS          # every translation unit with the flags the library build gives it
with open(out, "w") as cat:
    for src in SOURCES:
        part = out + "." + src
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", part,
                        os.path.join(ROOT, "mel_spec_amd", "csrc", src)] + UNIT_FLAGS.get(src, []), check=True, stderr=subprocess.DEVNULL)
        cat.write(open(part).read())
name, body, bad = None, [], 0
def report(name, body):
    prio = [i for i, l in enumerate(body) if "s_setprio" in l]
    if not prio:
        depth, inside, total = 0, 0, 0
        for l in body:
            m = re.match(r"^\.LBB\d+_\d+:\s*;?(.*)", l)
            if m:
                d = re.search(r"Depth[= ](\d)", m.group(1))
                depth = int(d.group(1)) if d else 0
            if "scratch_" in l:
                total += 1
                inside += depth > 0
        if total:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:100]
            print(f"{'loop ' if inside else '     '}{inside:3d} inside a loop,   {total:3d} in all   {dem}")
        return 0
    first = prio[0]
    # the back edge of the unit loop: the last branch to a label defined before the first s_setprio
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    back = first
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if m and i > prio[-1] and labels.get(m.group(1) or m.group(2), 10 ** 9) <= first:
            back = i
            break
    inside = [i for i, l in enumerate(body) if "scratch_" in l and first <= i <= back]
    total = sum("scratch_" in l for l in body)
    if total:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:100]
        print(f"{'LOOP ' if inside else '     '}{len(inside):3d} in the unit loop, {total:3d} in all   {dem}")
    return len(inside)
for line in open(out):
    m = re.match(r"^(_ZN7melspec\w+):", line)
    if m:
        if name: bad += report(name, body)
        name, body = m.group(1), []
    elif ".amdhsa_kernel" in line and name:
        bad += report(name, body)
        name, body = None, []
    elif name is not None:
        body.append(line)
sys.exit(1 if bad else 0)
