#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.npz from the CPU oracle (oracle/melspec_oracle.c).

The reference is Rust and cannot be executed here; the oracle is pinned to the reference's own
fixtures first (tests/test_oracle.py: mel_filters.npz @1e-7, rust_jfk_golden.npy @1e-6 with
max-abs-diff 0.0), and only then used to emit these extra vectors for configurations the
reference's fixtures do not cover (400/160/80, 400/160/128, fbank values, edge lengths).
Inputs are either the reference's own jfk_f32le.wav or deterministic generators, so only the
expected outputs are stored.   Run:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def four_tone():
    sr = np.float32(16000.0)
    t = np.arange(16000, dtype=np.float32) / sr
    two_pi = np.float32(2.0) * np.float32(np.pi)
    return (np.float32(0.6) * np.sin(two_pi * np.float32(220.0) * t)
            + np.float32(0.25) * np.sin(two_pi * np.float32(440.0) * t)
            + np.float32(0.10) * np.sin(two_pi * np.float32(880.0) * t)
            + np.float32(0.05) * np.sin(two_pi * np.float32(1760.0) * t)).astype(np.float32)


def main():
    jfk = O.load_wav_f32(os.path.join(HERE, "jfk_f32le.wav"))
    g = np.load(os.path.join(HERE, "rust_jfk_golden.npy"))
    assert np.abs(O.stream_mel(jfk, 512, 160, 80).T - g).max() <= 1e-6, "oracle no longer matches the reference golden"
    out = {
        "jfk_w80": O.compute_mel_spectrogram_cpu(jfk, 400, 160, 80),
        "jfk_w128": O.compute_mel_spectrogram_cpu(jfk, 400, 160, 128),
        "jfk_fbank_cmn": O.fbank_compute(jfk),
        "tone_w80": O.compute_mel_spectrogram_cpu(four_tone(), 400, 160, 80),
    }
    cfg = O.fbank_default_config(); cfg.apply_cmn = 0
    out["jfk_fbank_nocmn"] = O.fbank_compute(jfk, cfg)
    for c in range(8):
        x = O.synth_pcm(c, 16000)
        out[f"noise{c}_w80"] = O.compute_mel_spectrogram_cpu(x, 400, 160, 80)
    out["noise3_fbank"] = O.fbank_compute(O.synth_pcm(3, 16000))
    out["zeros_w80"] = O.compute_mel_spectrogram_cpu(np.zeros(1000, np.float32), 400, 160, 80)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
