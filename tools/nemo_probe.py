#!/usr/bin/env python3
"""Timing of the NeMo/Parakeet frontend under a few configurations (1024 x 10 s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
CASES = (dict(n_mels=128, preemphasis=0.97), dict(n_mels=80, preemphasis=0.97), dict(n_mels=128, preemphasis=0.0),
         dict(n_mels=128, preemphasis=0.97, center=False), dict(n_mels=128, preemphasis=0.97, normalize_per_feature=True))
if os.environ.get("NEMO_ONLY") == "norm": CASES = (CASES[0], CASES[-1])     # plain and normalised, 128 mels
for kw in CASES:
    fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(log_zero_guard=2.0 ** -24, **kw))
    cols = fe.padded_frames(clip_len)
    out = M.DeviceBuffer(n_clips * cols * kw["n_mels"] * 4)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5): fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
        fe.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
    fe.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(kw, f"{dt * 1e3:.3f} ms  {n_clips * cols / dt / 1e9:.3f} G frames/s", flush=True)
    out.free(); fe.close()
