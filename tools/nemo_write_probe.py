#!/usr/bin/env python3
"""Two launches of the NeMo frontend and of the Kaldi fbank (for rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24))
out = M.DeviceBuffer(n_clips * fe.padded_frames(clip_len) * 128 * 4)
for _ in range(2):
    fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); fe.synchronize()
fb = M.Fbank()        # CMN on: the clip kernel (or, MELSPEC_FB_CLIP=0 in lab builds, the fused kernel + cmn_kernel)
for _ in range(2):
    fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); fb.synchronize()
