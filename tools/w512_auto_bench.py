#!/usr/bin/env python3
"""MELSPEC_PRECISION_AUTO at n_fft = 512 (round 6) on 1024 x 10 s: hash noise and the reference's speech fixture through AUTO (the voting f32
launch + the gated f64 launch), F64 and F32; 80 and 128 mels; event-timed, ms per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O
n_clips, clip_len = 1024, 160000
noise = M.DeviceBuffer(n_clips * clip_len * 4); speech = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(noise.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
jfk = O.load_wav_f32(os.path.join(ROOT, "tests", "golden", "jfk_f32le.wav"))
x = np.stack([np.resize(np.roll(jfk, -1237 * c), clip_len) for c in range(64)])
speech.upload(np.tile(x, (n_clips // 64, 1)).reshape(-1))
for nm in (80, 128):
    m = M.HipMelSpectrogram(512, 160, 16000.0, nm)
    out = M.DeviceBuffer(n_clips * m.num_frames(clip_len) * nm * 4)
    for mode in ("auto", "f64", "f32"):
        m.set_precision(mode)
        row = []
        for name, pcm in (("noise", noise), ("speech", speech)):
            ms = min(m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=60, iters=200) for _ in range(3))
            row.append(f"{name} {ms:.4f} ms")
        print(f"n_mels {nm:3d}  {mode:4s} ({m.precision}):  " + "   ".join(row) + f"   [{m.plain_kernel_name()[:60]}]", flush=True)
    m.close(); out.free()
