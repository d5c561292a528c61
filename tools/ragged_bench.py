#!/usr/bin/env python3
"""Ragged batch (1024 clips of 5..15 s, packed back to back: the same 655 MB of PCM as config 2) against the uniform batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O
n_clips, n_mels = int(os.environ.get("L_CLIPS", "1024")), int(os.environ.get("L_MELS", "80"))
base_len = int(os.environ.get("L_LEN", "160000"))     # mean clip length; lengths are drawn from 0.5x .. 1.5x
rng = np.random.default_rng(7)
lens = rng.integers(base_len // 2, base_len * 3 // 2 + 1, n_clips).astype(np.uint64)
lens = (lens * (n_clips * base_len / lens.sum())).astype(np.uint64) & ~np.uint64(1)
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
total = int(lens.sum())
m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
frames = np.array([m.num_frames(int(l)) for l in lens], dtype=np.uint64)
ooff = np.concatenate([[0], np.cumsum(frames * n_mels)[:-1]]).astype(np.uint64)
pcm = M.DeviceBuffer(n_clips * base_len * 4)
M.synth_pcm_device(pcm.ptr, base_len, base_len, 0, n_clips); M.device_synchronize()
out = M.DeviceBuffer(max(int(frames.sum()), n_clips * m.num_frames(base_len)) * n_mels * 4 + 4096)     # the uniform run of the same buffer can be the larger one
REPS = int(os.environ.get("L_REPS", "200"))
def bench(fn, reps=None):
    reps = reps or REPS
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20): fn()
        m.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    m.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
tu = bench(lambda: m.compute_uniform_device(pcm.ptr, base_len, base_len, n_clips, out.ptr))
tr = bench(lambda: m.compute_ragged_device(pcm.ptr, offs, lens, out.ptr, ooff))
print(f"uniform {n_clips} x {base_len / 16000:.0f} s        {tu:.4f} ms  {n_clips * m.num_frames(base_len) / tu / 1e6:.3f} G frames/s")
print(f"ragged  {n_clips} x {base_len / 32000:.0f}..{base_len * 3 / 32000:.0f} s     {tr:.4f} ms  {int(frames.sum()) / tr / 1e6:.3f} G frames/s   ({int(frames.sum())} frames)")
# parity of two clips of the ragged run (the packed PCM is the uniform synthetic buffer read at other offsets)
host = np.concatenate([O.synth_pcm(c, base_len) for c in range(3)])
for c in (0, 1):
    got = out.download((int(frames[c]), n_mels), offset_bytes=int(ooff[c]) * 4)
    want = O.compute_mel_spectrogram_cpu(host[int(offs[c]):int(offs[c] + lens[c])], 400, 160, n_mels)
    print("clip", c, "max |diff|", float(np.abs(got - want).max()))
