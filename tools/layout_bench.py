#!/usr/bin/env python3
"""Frame-major vs mel-major (interleave_frames) store of the fused Whisper kernel, 1024 x 10 s, 80 mels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
n_clips, clip_len, n_mels = int(os.environ.get("L_CLIPS", "1024")), int(os.environ.get("L_LEN", "160000")), int(os.environ.get("L_MELS", "80"))
m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
fpc = m.num_frames(clip_len)
W = m.interleaved_width(clip_len, 0)
out = M.DeviceBuffer(n_clips * max(W, fpc) * n_mels * 4 + 4096)
REPS = int(os.environ.get("L_REPS", "200"))
def bench(fn):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20): fn()
        m.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS): fn()
    m.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3
print("frame-major            %.4f ms" % bench(lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)))
print("mel-major   W=%d     %.4f ms" % (W, bench(lambda: m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, False, 0))))
print("frame-major padded     %.4f ms" % bench(lambda: m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, True, 0)))
