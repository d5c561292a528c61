#!/usr/bin/env python3
"""NeMo frontend on a ragged batch (1024 clips of 5..15 s, 128 mels), plain and with per-feature normalisation: one launch per call
(melspec_blm_compute_ragged_device), device-resident."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import mel_spec_amd as M
from mel_spec_amd._lib import lib
rng = np.random.default_rng(3)
n_clips = 1024
lens = rng.integers(80000, 240001, n_clips).astype(np.uint64)
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
total = int(lens.sum())
pcm = M.DeviceBuffer(total * 4)
M.synth_pcm_device(pcm.ptr, total, total, 0, 1); M.device_synchronize()
u64p = C.POINTER(C.c_uint64)
for norm in (False, True):
    fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24, normalize_per_feature=norm))
    cols = [fe.padded_frames(int(n)) for n in lens]
    out = M.DeviceBuffer(sum(cols) * 128 * 4)
    run = lambda: M.hip._check(lib().melspec_blm_compute_ragged_device(fe._h, C.c_void_p(pcm.ptr), offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p),
                                                                  n_clips, C.c_void_p(out.ptr), None, None))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(3): run()
        fe.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20): run()
        fe.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print(f"ragged NeMo 128 mels, normalise={norm}: {best * 1e3:.3f} ms  {sum(cols) / best / 1e9:.3f} G frames/s ({sum(cols)} columns)", flush=True)
    out.free(); fe.close()
