#!/usr/bin/env python3
"""Uniform batch through the uniform kernel (units dealt round-robin) and through the ragged kernel (a contiguous run of units per wave)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
n_clips, clip_len, n_mels = int(os.environ.get("L_CLIPS", "1024")), int(os.environ.get("L_LEN", "160000")), int(os.environ.get("L_MELS", "80"))
reps = int(os.environ.get("L_REPS", "200"))
m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
fpc = m.num_frames(clip_len)
pcm = M.DeviceBuffer(n_clips * clip_len * 4); out = M.DeviceBuffer(n_clips * fpc * n_mels * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
offs = np.arange(n_clips, dtype=np.uint64) * np.uint64(clip_len)
lens = np.full(n_clips, clip_len, np.uint64)
ooff = np.arange(n_clips, dtype=np.uint64) * np.uint64(fpc * n_mels)
def bench(fn):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(max(2, reps // 10)): fn()
        m.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    m.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for _ in range(2):
    a = bench(lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr))
    b = bench(lambda: m.compute_ragged_device(pcm.ptr, offs, lens, out.ptr, ooff))
    print(f"{n_clips} x {clip_len / 16000:.0f} s, {n_mels} mels: dealt {a:.4f} ms   runs {b:.4f} ms   ({n_clips * fpc / a / 1e6:.3f} vs {n_clips * fpc / b / 1e6:.3f} G frames/s)")
