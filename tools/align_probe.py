#!/usr/bin/env python3
"""Does the alignment of the PCM matter?  Uniform 1024 x 10 s batch with the base pointer shifted by 0 / 8 / 32 / 64 bytes and with an odd clip stride."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer((n_clips * (clip_len + 64) + 64) * 4)
M.synth_pcm_device(pcm.ptr, clip_len + 64, clip_len + 64, 0, n_clips); M.device_synchronize()
m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
out = M.DeviceBuffer(n_clips * m.num_frames(clip_len) * 80 * 4)
for shift, stride in ((0, clip_len), (8, clip_len), (32, clip_len), (64, clip_len), (0, clip_len + 2), (0, clip_len + 18), (0, clip_len + 32)):
    ts = [m.time_uniform_device(pcm.ptr + shift, stride, clip_len, n_clips, out.ptr, warmup=30, iters=100) for _ in range(3)]
    print(f"base + {shift:2d} B, clip stride {stride} samples: {min(ts):.4f} ms")
