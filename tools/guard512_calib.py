#!/usr/bin/env python3
"""Calibration of the precision guard for the f32 512-point Whisper kernel (GPU box): tools/flag_calib.py's zoo through
MELSPEC_PRECISION_F32 (the bare f32 kernel) against the oracle, per frame e = max_m |f32 - oracle| and r_min = the depth of the lowest
unclamped band under the frame maximum in decades (what w512_phase4's guard sees).  Prints, per guard band B (flag when a band sits
within B decades of the clamp, i.e. r_min < -(8 - B)), the worst error among UNFLAGGED frames and the flagged fraction per class.
usage: tools/guard512_calib.py [n_mels]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
from oracle import oracle as O
n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
sr, hop = 16000.0, 160
rng = np.random.default_rng(5)
N = 16000 * 4
t = np.arange(N) / sr
def tone_floor(f, level_db, amp=0.9):
    return (amp * np.sin(2 * np.pi * f * t) + 10 ** (level_db / 20) * rng.standard_normal(N)).astype(np.float32)
zoo = {}
zoo["jfk"] = [O.load_wav_f32(os.path.join(ROOT, "tests", "golden", "jfk_f32le.wav"))]
zoo["noise"] = [rng.standard_normal(N).astype(np.float32) * np.float32(10.0 ** rng.uniform(-4, 0)) for _ in range(4)] + [O.synth_pcm(c, N) for c in range(4)]
zoo["tone"] = [(np.sin(np.arange(N) * rng.uniform(0.01, 3.0)) * rng.uniform(0.01, 1.0)).astype(np.float32) for _ in range(12)]
zoo["tone+floor"] = [tone_floor(f, lv, a) for f in (200.0, 1000.0, 3333.3, 5000.0, 7000.0, 7800.0) for lv in (-40, -50, -60, -65, -70, -75, -80, -90) for a in (0.9, 0.05)]
imp = []
for _ in range(12):
    x = rng.standard_normal(N).astype(np.float32) * np.float32(10.0 ** rng.uniform(-5, -2))
    x[:: int(rng.integers(50, 500))] += 0.7
    imp.append(x)
zoo["impulses"] = imp
ch = []
for lv in (-50, -70, -90):
    f = 100 + 7800 * (t / t[-1])
    ch.append((0.8 * np.sin(2 * np.pi * np.cumsum(f) / sr) + 10 ** (lv / 20) * rng.standard_normal(N)).astype(np.float32))
zoo["chirp+floor"] = ch
zoo["two tones"] = [(0.9 * np.sin(2 * np.pi * 440.0 * t) + 10 ** (lv / 20) * np.sin(2 * np.pi * 6000.0 * t)).astype(np.float32) for lv in (-60, -75, -85, -95)]
j = zoo["jfk"][0].astype(np.float64)
J = np.fft.rfft(j); fr = np.fft.rfftfreq(len(j), 1 / 16000.0)
zoo["jfk lowpass"] = [np.fft.irfft(J * np.where(fr < cut, 1.0, att), len(j)).astype(np.float32) for cut, att in ((3400, 1e-4), (4000, 1e-5), (3400, 0.0))]
zoo["quiet jfk"] = [(zoo["jfk"][0] * s).astype(np.float32) for s in (1e-2, 1e-3)]

m = M.HipMelSpectrogram(512, hop, sr, n_mels)
m.set_precision("f32")
assert m.precision == "f32"
rows = []
for k, sigs in zoo.items():
    for x in sigs:
        got = m.compute_mel_spectrogram(x)
        want = O.compute_mel_spectrogram_cpu(x, 512, hop, n_mels, sr)
        e = np.abs(got - want).max(axis=1)
        v = 4.0 * got.astype(np.float64) - 4.0
        r = v - v.max(axis=1, keepdims=True)
        r_min = r.min(axis=1)                       # clamped bands sit at exactly -8: they count (the kernel's guard counts them too)
        silent = (v.max(axis=1) <= -10.0 + 1e-9)     # every band on the 1e-10 floor: lo is negative, nothing is within reach
        for a, b, s in zip(e, r_min, silent):
            rows.append((k, a, 0.0 if s else b))
cls = np.array([r[0] for r in rows]); e = np.array([r[1] for r in rows]); rm = np.array([r[2] for r in rows])
print(f"f32 512-point Whisper kernel, n_mels={n_mels}: {len(e)} frames, worst f32 error {e.max():.3e}")
for k in zoo:
    q = cls == k
    print(f"  {k:12s} {q.sum():6d} frames  worst e {e[q].max():.2e}  frames with e > 5e-5 / 1e-4: {(e[q] > 5e-5).sum():5d} {(e[q] > 1e-4).sum():5d}")
print("guard band B (flag when r_min < -(8 - B)): worst error among unflagged frames | flagged fraction per class")
for B in (2.0, 2.25, 2.5, 2.75, 3.0, 3.5, 4.0):
    fl = rm < -(8.0 - B)
    worst = e[~fl].max() if (~fl).any() else 0.0
    print(f"  B={B:4.2f}  worst unflagged {worst:.2e}   " + "  ".join(f"{k}:{(fl & (cls == k)).sum() / max(1, (cls == k).sum()):.3f}" for k in zoo))
print("max error by r_min bin:")
for lo in np.arange(-8, 0, 0.5):
    q = (rm >= lo) & (rm < lo + 0.5)
    if q.any():
        print(f"  [{lo:5.1f},{lo + 0.5:5.1f})  n={q.sum():6d}  max e {e[q].max():.2e}  p99 {np.quantile(e[q], 0.99):.2e}")
