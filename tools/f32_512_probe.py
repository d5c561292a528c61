#!/usr/bin/env python3
"""What an f32 build of the fused 512-point kernels (NeMo-128, Whisper-512) does to the results (GPU box; MELSPEC_LIB selects the library):
per input class the largest |difference| from the f64 evaluation of the reference's definition, the share of VALUES past 1e-4, the share
of FRAMES holding such a value, and -- for NeMo -- the same figures for the reference's own literal f32 arithmetic (oracle, f64=False).
tools/f32_512_probe.py [name]   ->  one line per (flavour, input)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O

SR = 16000
rng = np.random.default_rng(5)
jfk = O.load_wav_f32(os.path.join(ROOT, "tests", "golden", "jfk_f32le.wav"))
n = 160000
t = np.arange(n) / SR
inputs = {
    "hash noise (the bench's clips)": O.synth_pcm(3, n),
    "gaussian noise 0.1": (0.1 * rng.standard_normal(n)).astype(np.float32),
    "speech (jfk)": np.resize(jfk, n).astype(np.float32),
    "tone 440 Hz over -70 dB floor": (0.5 * np.sin(2 * np.pi * 440 * t) + 10 ** (-70 / 20) * rng.standard_normal(n)).astype(np.float32),
    "chirp": (0.3 * np.sin(2 * np.pi * (100 + 3000 * t / t[-1]) * t)).astype(np.float32),
    "quiet speech (jfk x 1e-3)": (1e-3 * np.resize(jfk, n)).astype(np.float32),
}


def stats(got, want, axis_frames):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    bad = d > 1e-4
    return d.max(), bad.mean(), bad.any(axis=axis_frames).mean()


tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("MELSPEC_LIB", "default")
fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97))
w5 = M.HipMelSpectrogram(512, 160, float(SR), 80)
for name, x in inputs.items():
    want, valid = O.blm_compute(x, O.blm_default_config(n_mels=128, preemphasis=0.97), True)
    lit, _ = O.blm_compute(x, O.blm_default_config(n_mels=128, preemphasis=0.97), False)
    got = np.asarray(fe.compute(x))[:, :valid]
    mx, fv, ff = stats(got, want[:, :valid], 0)
    lmx, lfv, lff = stats(lit[:, :valid], want[:, :valid], 0)
    print(f"{tag:10s} NeMo-128     {name:32s} max {mx:9.2e}  values>1e-4 {100 * fv:6.2f} %  frames {100 * ff:6.2f} %   | reference's f32: max {lmx:9.2e} values {100 * lfv:6.2f} % frames {100 * lff:6.2f} %")
    want5 = O.compute_mel_spectrogram_cpu(x, 512, 160, 80, float(SR))
    got5 = np.asarray(w5.compute_mel_spectrogram(x))
    mx, fv, ff = stats(got5, want5, 1)
    print(f"{tag:10s} Whisper-512  {name:32s} max {mx:9.2e}  values>1e-4 {100 * fv:6.2f} %  frames {100 * ff:6.2f} %")
