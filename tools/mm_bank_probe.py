#!/usr/bin/env python3
"""The mel-major launch of any bank at 1024 x 10 s (lab library + MELSPEC_MM_SYNC sweeps the sub-group barrier): tools/mm_bank_probe.py <n_mels>.
Round 6: run-time banks of 96 / 100 mels (five-frame kernel) are best at their default (pairs 4 apart: 0.472-0.492 ms; fours +5 %, none +2...+8 %)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mel_spec_amd as M
nm = int(sys.argv[1])
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
m = M.HipMelSpectrogram(400, 160, 16000.0, nm)
out = M.DeviceBuffer(n_clips * (m.num_frames(clip_len) + 8) * nm * 4)
fn = lambda: m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, False, 2)
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
print(nm, "mel-major %.4f ms" % best, m.plain_kernel_name()[:50])
