#!/usr/bin/env python3
"""Geometries off the fused kernels (generic_frame_kernel): Whisper-style log-mel at other n_fft, Kaldi fbank at 8 kHz.
Lab library: MELSPEC_GENERIC_FFT=0 keeps the direct O(N^2) transform for power-of-two sizes too."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import mel_spec_amd as M
from oracle import oracle as O
n_clips, clip_len = 256, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
for n_fft, hop, n_mels in ((256, 64, 40), (1024, 256, 80), (2048, 512, 128), (4096, 1024, 128), (320, 160, 80), (800, 200, 80), (1200, 300, 128), (1000, 250, 80),
                           (400, 160, 160), (441, 160, 80)):
    m = M.HipMelSpectrogram(n_fft, hop, 16000.0, n_mels)
    nf = m.num_frames(clip_len)
    out = M.DeviceBuffer(n_clips * nf * n_mels * 4)
    run = lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
    run(); m.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run()
    m.synchronize()
    dt = (time.perf_counter() - t0) / 3
    got = out.download((nf, n_mels))
    d = float(np.abs(got - O.compute_mel_spectrogram_cpu(O.synth_pcm(0, clip_len), n_fft, hop, n_mels)).max())
    print(f"n_fft {n_fft:5d} hop {hop:4d} mels {n_mels:3d}: {dt * 1e3:9.3f} ms  {n_clips * nf / dt / 1e6:9.1f} M frames/s  parity {d:.2e}  fused={m.uses_fast_path}", flush=True)
    out.free(); m.close()
