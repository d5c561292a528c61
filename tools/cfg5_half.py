#!/usr/bin/env python3
"""Half of BASELINE config 5 on one GPU: 32768 x 30 s clips (62.9 GB of PCM, 31.4 GB of mel) in one launch,
with parity on clips sampled across the 64-bit offset range."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O
n_clips, clip_len, n_mels = int(os.environ.get("C5_CLIPS", "32768")), 480000, 80
m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
fpc = m.num_frames(clip_len)
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
out = M.DeviceBuffer(n_clips * fpc * n_mels * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
for _ in range(2):
    m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
m.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
m.synchronize()
dt = (time.perf_counter() - t0) / 5
worst = 0.0
for c in (0, 1, n_clips // 3, n_clips // 2 + 1, n_clips - 2, n_clips - 1):
    got = out.download((fpc, n_mels), offset_bytes=c * fpc * n_mels * 4)
    worst = max(worst, float(np.abs(got - O.compute_mel_spectrogram_cpu(O.synth_pcm(c, clip_len), 400, 160, n_mels)).max()))
frames = n_clips * fpc
print(f"{n_clips} x 30 s: {frames} frames in {dt * 1e3:.2f} ms = {frames / dt / 1e9:.3f} G frames/s, {frames * 960 / dt / 1e12:.3f} TB/s algorithmic, parity {worst:.2e}")
