#!/usr/bin/env python3
"""What MELSPEC_PRECISION_AUTO costs on input that trips the precision guard: 1024 x 10 s clips of (a) hash noise, (b) speech
(jfk_f32le.wav tiled, per-clip offsets), (c) a tone over a -70 dB floor -- auto / f64 / f32, ms per launch and frames recomputed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O

n_clips, clip_len = 1024, 160000
jfk = O.load_wav_f32(os.path.join(ROOT, "tests", "golden", "jfk_f32le.wav"))
rng = np.random.default_rng(0)
t = np.arange(clip_len) / 16000.0
sets = {
    "hash noise": np.stack([O.synth_pcm(c % 8, clip_len) for c in range(64)]),
    "speech (jfk tiled)": np.stack([np.resize(np.roll(jfk, -1237 * c), clip_len) for c in range(64)]),
    "tone over -70 dB floor": np.stack([(0.9 * np.sin(2 * np.pi * (300 + 97 * c) * t) + 10 ** (-70 / 20) * rng.standard_normal(clip_len)).astype(np.float32) for c in range(64)]),
}
for n_mels in (80, 128):
    m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
    nf = m.num_frames(clip_len)
    pcm, out = M.DeviceBuffer(n_clips * clip_len * 4), M.DeviceBuffer(n_clips * nf * n_mels * 4)
    for name, x in sets.items():
        for r in range(n_clips // 64):
            pcm.upload(x, offset_bytes=r * x.nbytes)
        for mode in ("f32", "auto", "auto-pinned", "f64"):
            m.set_precision(mode.split("-")[0])
            m.set_auto_adaptive(mode != "auto-pinned")      # "auto-pinned": the f32 regime whatever the input (round 2's AUTO)
            m.guard_last_count()
            for _ in range(2):      # the first call's statistics decide the regime of the second
                m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); m.synchronize()
            flagged = m.guard_last_count() / 2
            ms = m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=20, iters=100)
            heavy, frac = m.auto_state()
            print(f"{n_mels:3d} mels  {name:24s} {mode:11s}  {ms:7.4f} ms  {n_clips * nf / ms / 1e6:7.2f} G frames/s   guard tripped on {flagged / (n_clips * nf) * 100:5.1f} % of the frames"
                  f"   regime {'f64 kernel' if heavy and mode == 'auto' else '-'}  ({m.plain_kernel_name()})", flush=True)
    pcm.free(); out.free(); m.close()
