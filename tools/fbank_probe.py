#!/usr/bin/env python3
"""Config 3 (Kaldi fbank + CMN, 1024 x 10 s, 80 bins): time and parity of whatever path the loaded library takes.
Lab library: MELSPEC_FB_CLIP=0 keeps the fused kernel + cmn_kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (HIP runtime load order)
import mel_spec_amd as M
from oracle import oracle as O
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
clip_len = 160000
fb = M.Fbank()
fpc = fb.num_frames(clip_len)
pcm = M.DeviceBuffer(n_clips * clip_len * 4); out = M.DeviceBuffer(n_clips * fpc * 80 * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
run = lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(5): run()
    fb.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(40): run()
    fb.synchronize()
    best = min(best, (time.perf_counter() - t0) / 40)
worst = 0.0
for c in (0, n_clips // 2, n_clips - 1):
    got = out.download((fpc, 80), offset_bytes=c * fpc * 80 * 4)
    worst = max(worst, float(np.abs(got - O.fbank_compute(O.synth_pcm(c, clip_len))).max()))
print(f"fbank {n_clips} x 10 s: {best * 1e3:.4f} ms  {n_clips * fpc / best / 1e9:.3f} G frames/s  {n_clips * fpc * 960 / best / 1e12:.3f} TB/s  parity {worst:.2e}  "
      f"FB_CLIP={os.environ.get('MELSPEC_FB_CLIP', '-')} lib={os.environ.get('MELSPEC_LIB', 'product')}", flush=True)
