#!/usr/bin/env python3
"""How uneven is the end of the headline kernel?  Lab build with -DMELSPEC_LAB_STAMPS (tools/ab_build.sh stamps:"-DMELSPEC_LAB -DMELSPEC_LAB_STAMPS"):
every wave of whisper400_six_runs_kernel leaves the constant-clock time (10 ns ticks) at which it finished, every workgroup the time it
started; the 200th launch of config 2 is printed.  Reports the spread: if the last wave ends long after the median one, a dynamic
hand-out of the last units (instead of one static run per wave) has that much to win.
usage (GPU box): MELSPEC_LIB=mel_spec_amd/ab/lib_stamps.so MELSPEC_LAB_STAMPS=1 python tools/tail_probe.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1])
import mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
out = M.DeviceBuffer(n_clips * m.num_frames(clip_len) * 80 * 4)
for _ in range(260): m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
m.synchronize()
'''
p = subprocess.run([sys.executable, "-c", WORKER, ROOT], capture_output=True, text=True, env=dict(os.environ, MELSPEC_LAB_STAMPS="1"))
rows = [l.split() for l in p.stderr.splitlines() if l.startswith("STAMP")]
if not rows:
    print(p.stderr[-2000:]); raise SystemExit("no stamps (is MELSPEC_LIB a -DMELSPEC_LAB_STAMPS build?)")
start = np.array([int(r[4]) for r in rows]) * 0.01                    # us
ends = np.array([[int(x) for x in r[6:]] for r in rows]) * 0.01       # [wg][wave] us
print(f"{len(rows)} workgroups x {ends.shape[1]} waves; times in us from the first workgroup's start")
print(f"workgroup starts: min {start.min():.2f}  median {np.median(start):.2f}  max {start.max():.2f}")
e = ends.reshape(-1)
print(f"wave ends:        min {e.min():.2f}  p10 {np.percentile(e, 10):.2f}  median {np.median(e):.2f}  p90 {np.percentile(e, 90):.2f}  p99 {np.percentile(e, 99):.2f}  max {e.max():.2f}")
print(f"kernel end (last wave) - mean wave end = {e.max() - e.mean():.2f} us of {e.max():.2f}: the idle tail a perfect hand-out would remove is {100 * (e.max() - e.mean()) / e.max():.1f} %")
wg_end = ends.max(axis=1)
print(f"workgroup ends:   min {wg_end.min():.2f}  median {np.median(wg_end):.2f}  max {wg_end.max():.2f}")
xcd = np.arange(len(rows)) % 8
for x in range(8):
    print(f"  XCD {x}: mean wave end {ends[xcd == x].mean():.2f}  last {ends[xcd == x].max():.2f}")
