#!/usr/bin/env python3
"""Plain and mel-major launches of a compile-time bank of the six-frame family at 1024 x 10 s (profiles/r06_wide_layouts.txt):
tools/nm64_probe.py [n_mels = 64]   (MELSPEC_LIB selects a library variant)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O
n_clips, clip_len, nm = 1024, 160000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
m = M.HipMelSpectrogram(400, 160, 16000.0, nm)
nf = m.num_frames(clip_len)
out = M.DeviceBuffer(n_clips * (nf + 8) * nm * 4)
for name, fn in (("plain", lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)),
                 ("mel-major", lambda: m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, False, 2))):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(20): fn()
        m.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(200): fn()
        m.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
    print(name, "%.4f ms" % best, m.plain_kernel_name()[:60])
got = None
m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); m.synchronize()
g = out.download((nf, nm), offset_bytes=0)
print("parity", float(np.abs(g - O.compute_mel_spectrogram_cpu(O.synth_pcm(0, clip_len), 400, 160, nm)).max()))
