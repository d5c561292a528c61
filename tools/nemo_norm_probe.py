#!/usr/bin/env python3
"""30 launches of the NeMo frontend with per-feature normalisation (1024 x 10 s, 128 mels): the workload for rocprofv3 runs of blm_normalize_kernel."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24, normalize_per_feature=True))
out = M.DeviceBuffer(n_clips * fe.padded_frames(clip_len) * 128 * 4)
for _ in range(30):
    fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
fe.synchronize()
