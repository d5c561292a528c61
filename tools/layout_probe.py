#!/usr/bin/env python3
"""Three launches each of the frame-major and the mel-major store (for rocprofv3 --pmc WRITE_SIZE FETCH_SIZE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
n_clips, clip_len, n_mels = 1024, 160000, 80
m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
out = M.DeviceBuffer(n_clips * 1000 * n_mels * 4 + 4096)
for _ in range(3):
    m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); m.synchronize()
for _ in range(3):
    m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, False, 0); m.synchronize()
