#!/usr/bin/env python3
"""BASELINE config 3 (Kaldi fbank 80 bins + CMN, 1024 x 10 s) with the CMN inside (Fbank::compute's contract) and as the split output of
round 6 (melspec_fbank_compute_uniform_device_split: rows before CMN + the means): event-timed ms per call, same box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
fb = M.Fbank()
nf = fb.num_frames(clip_len)
out = M.DeviceBuffer(n_clips * nf * 80 * 4); means = M.DeviceBuffer(n_clips * 80 * 4)
import time
def timed(fn, iters=200):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10): fn()
        fb.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(iters): fn()
        fb.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode in ("both", "cmn"):
    print(f"CMN inside (default contract):            {timed(lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)):.4f} ms")
if mode in ("both", "split"):
    print(f"split output (rows before CMN + means):   {timed(lambda: fb.compute_uniform_device_split(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, means.ptr)):.4f} ms")
