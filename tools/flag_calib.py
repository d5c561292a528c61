#!/usr/bin/env python3
"""Calibration of the per-frame precision flag (CPU only: host emulation of the f32 kernel vs the f64 oracle).

For every frame of a zoo of signals: e = max_m |f32 kernel - oracle| and the candidate statistics the kernel can compute for
free in phase 4 from the log-mel values v_m and the frame maximum mx:
    r_min  = min over unclamped mels of (v_m - mx)          (in [-8, 0])
Prints, per threshold T (flag when r_min < -T), the worst error among UNFLAGGED frames and the flagged fraction per signal class.
Usage: tools/flag_calib.py [n_mels] [hop] [sr]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
hop = int(sys.argv[2]) if len(sys.argv) > 2 else 160
sr = float(sys.argv[3]) if len(sys.argv) > 3 else 16000.0

d = os.path.join(ROOT, "tests", "emu")
subprocess.check_call(["make", "-C", d, "-s"])
L = C.CDLL(os.path.join(d, "libmelspec_emu.so"))
f32p = C.POINTER(C.c_float)
L.emu_whisper_six.restype = C.c_longlong
L.emu_whisper_six.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p]
L.emu_whisper_wave.restype = C.c_longlong
L.emu_whisper_wave.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p]


def f32_kernel(x):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    if n_mels <= 80:
        got = L.emu_whisper_six(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, 0, out.ctypes.data_as(f32p))
    else:
        got = L.emu_whisper_wave(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, 4, out.ctypes.data_as(f32p))
    assert got == nf, (got, nf)
    return out


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
two = []
for lv in (-60, -75, -85, -95):
    two.append((0.9 * np.sin(2 * np.pi * 440.0 * t) + 10 ** (lv / 20) * np.sin(2 * np.pi * 6000.0 * t)).astype(np.float32))
zoo["two tones"] = two
# speech over shaped floors: jfk low-passed (a band-limited recording resampled to 16 kHz)
j = zoo["jfk"][0].astype(np.float64)
J = np.fft.rfft(j)
fr = np.fft.rfftfreq(len(j), 1 / 16000.0)
lp = []
for cut, att in ((3400, 1e-4), (4000, 1e-5), (3400, 0.0)):
    g = np.where(fr < cut, 1.0, att)
    lp.append(np.fft.irfft(J * g, len(j)).astype(np.float32))
zoo["jfk lowpass"] = lp

rows = []   # (class, e, r_min, r_min_all)
for k, sigs in zoo.items():
    for x in sigs:
        got = f32_kernel(x)
        want = O.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)
        e = np.abs(got - want).max(axis=1)
        v = 4.0 * got.astype(np.float64) - 4.0
        mx = v.max(axis=1, keepdims=True)
        r = v - mx
        uncl = r > -8.0 + 1e-6
        r_min = np.where(uncl, r, 0.0).min(axis=1)
        for a, b in zip(e, r_min):
            rows.append((k, a, b))

cls = np.array([r[0] for r in rows])
e = np.array([r[1] for r in rows])
rm = np.array([r[2] for r in rows])
print(f"n_mels={n_mels} hop={hop} sr={sr}: {len(e)} frames, worst f32 error {e.max():.3e}")
print("per class: frames, worst e, frames with e > 5e-5 / 1e-4")
for k in zoo:
    m = cls == k
    print(f"  {k:12s} {m.sum():6d}  {e[m].max():.2e}  {(e[m] > 5e-5).sum():5d} {(e[m] > 1e-4).sum():5d}   r_min quantiles {np.quantile(rm[m], [0, .01, .1, .5])}")
print("threshold T: worst error among unflagged frames | flagged fraction per class")
for T in (4.0, 4.5, 5.0, 5.5, 6.0, 6.5, 7.0, 7.5):
    fl = rm < -T
    worst = e[~fl].max() if (~fl).any() else 0.0
    fr_ = "  ".join(f"{k}:{(fl & (cls == k)).sum() / max(1, (cls == k).sum()):.3f}" for k in zoo)
    print(f"  T={T:.1f}  worst unflagged {worst:.2e}   {fr_}")
# error as a function of r_min (binned)
print("max error by r_min bin:")
for lo in np.arange(-8, 0, 0.5):
    m = (rm >= lo) & (rm < lo + 0.5)
    if m.any():
        print(f"  [{lo:5.1f},{lo + 0.5:5.1f})  n={m.sum():6d}  max e {e[m].max():.2e}  p99 {np.quantile(e[m], 0.99):.2e}")

# ---- energy-based statistic: s = log10(min unclamped E_m) - log10(||x_w||^2 * 400) -----------------------------------
w_hann = 0.5 * (1 - np.cos(2 * np.pi * np.arange(400) / 400))
rows2 = []
for k, sigs in zoo.items():
    for x in sigs:
        got = f32_kernel(x)
        want = O.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)
        e = np.abs(got - want).max(axis=1)
        v = 4.0 * got.astype(np.float64) - 4.0
        mx = v.max(axis=1, keepdims=True)
        uncl = (v - mx) > -8.0 + 1e-6
        vmin = np.where(uncl, v, 99.0).min(axis=1)
        nf = got.shape[0]
        idx = np.arange(nf)[:, None] * hop + np.arange(400)[None, :]
        fe = ((x.astype(np.float64)[idx] * w_hann) ** 2).sum(axis=1) * 400.0
        s = vmin - np.log10(np.maximum(fe, 1e-300))
        for a, b in zip(e, s):
            rows2.append((k, a, b))
cls2 = np.array([r[0] for r in rows2]); e2 = np.array([r[1] for r in rows2]); s2 = np.array([r[2] for r in rows2])
print("\nenergy statistic s = log10(Emin_unclamped / ||X||^2): max error by bin")
for lo in np.arange(-14, -2, 0.5):
    m = (s2 >= lo) & (s2 < lo + 0.5)
    if m.any():
        print(f"  [{lo:5.1f},{lo + 0.5:5.1f})  n={m.sum():6d}  max e {e2[m].max():.2e}  p99 {np.quantile(e2[m], 0.99):.2e}")
for T in (-9.5, -10.0, -10.5, -11.0, -11.5, -12.0):
    fl = s2 < T
    worst = e2[~fl].max() if (~fl).any() else 0.0
    fr_ = "  ".join(f"{k}:{(fl & (cls2 == k)).sum() / max(1, (cls2 == k).sum()):.3f}" for k in zoo)
    print(f"  flag s<{T:.1f}  worst unflagged {worst:.2e}   {fr_}")
print("\nper class, by s bin: n / max e / median e")
for k in zoo:
    print(" ", k)
    for lo in np.arange(-12, -8, 0.5):
        m = (s2 >= lo) & (s2 < lo + 0.5) & (cls2 == k)
        if m.any():
            print(f"    [{lo:5.1f},{lo + 0.5:5.1f})  n={m.sum():6d}  max e {e2[m].max():.2e}  med {np.median(e2[m]):.2e}")
