#!/usr/bin/env python3
"""Kaldi fbank + CMN on a ragged batch (1024 clips of 5..15 s, 80 bins), one launch per call (melspec_fbank_compute_ragged_device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import mel_spec_amd as M
from oracle import oracle as O
rng = np.random.default_rng(3)
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lens = rng.integers(80000, 240001, n_clips).astype(np.uint64)
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
total = int(lens.sum())
pcm = M.DeviceBuffer(total * 4)
M.synth_pcm_device(pcm.ptr, total, total, 0, 1); M.device_synchronize()
fb = M.Fbank()
frames = np.array([fb.num_frames(int(n)) for n in lens], dtype=np.uint64)
out = M.DeviceBuffer(int(frames.sum()) * 80 * 4)
run = lambda: fb.compute_ragged_device(pcm.ptr, offs, lens, out.ptr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(3): run()
    fb.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(20): run()
    fb.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
# parity of three clips against the oracle (the synthetic PCM is one long "clip 0")
whole = O.synth_pcm(0, total)
worst = 0.0
oo = np.concatenate([[0], np.cumsum(frames * 80)[:-1]])
for c in (0, n_clips // 2, n_clips - 1):
    got = out.download((int(frames[c]), 80), offset_bytes=int(oo[c]) * 4)
    want = O.fbank_compute(whole[int(offs[c]):int(offs[c] + lens[c])])
    worst = max(worst, float(np.abs(got - want).max()))
print(f"ragged fbank {n_clips} clips of 5..15 s: {best * 1e3:.3f} ms  {int(frames.sum()) / best / 1e9:.3f} G frames/s  parity {worst:.2e}", flush=True)
