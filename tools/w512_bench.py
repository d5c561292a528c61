#!/usr/bin/env python3
"""n_fft = 512 Whisper-style log-mel (the reference's RingBuffer golden geometry, 512/160/80) on 1024 x 10 s:
fused f64 512-point kernel vs the generic kernel (MELSPEC_W512=0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
n_clips, clip_len = 1024, 160000
m = M.HipMelSpectrogram(512, 160, 16000.0, 80)
fpc = m.num_frames(clip_len)
pcm = M.DeviceBuffer(n_clips * clip_len * 4); out = M.DeviceBuffer(n_clips * fpc * 80 * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); m.synchronize()
iters = 20 if m.uses_fast_path else 3
t0 = time.perf_counter()
for _ in range(iters):
    m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
m.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"fused={m.uses_fast_path}: {dt * 1e3:.3f} ms  {n_clips * fpc / dt / 1e9:.4f} G frames/s")
m.set_precision("f32")                      # round 5: the f32 instantiation (no guard: noise-like input only)
for _ in range(30): m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
m.synchronize()
t0 = time.perf_counter()
for _ in range(iters): m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
m.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"precision {m.precision}: {dt * 1e3:.3f} ms  {n_clips * fpc / dt / 1e9:.4f} G frames/s")
