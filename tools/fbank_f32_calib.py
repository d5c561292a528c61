#!/usr/bin/env python3
"""VERDICT r04 "next" 1, first step: what would an f32 Kaldi-fbank kernel cost in accuracy, and what can a guard see?  (CPU only.)

Fbank::compute (src/fbank.rs:141-236) is f64 inside; the fused kernel computes it in f64 and is gated at 1e-4 against the oracle.  This
script runs the kernel's own arithmetic instantiated in f32 (tests/emu: emu_fbank_wave(.., f64 = 0), the float instantiation of
fbank_wave.hpp) against the oracle's Fbank::compute without CMN -- ln(max(E, f32::EPSILON)) rows -- on BASELINE config 3's own input
(the hash-noise clips of SURVEY section 8(d)), jfk_f32le.wav and the signal zoo of tools/flag_calib2.py, and reports per class

  * the worst error of the f32 build and the fraction of FRAMES with a band off by more than 5e-5 / 1e-4 (what a PERFECT guard would
    hand to the f64 kernel),
  * for a guard the kernel can evaluate -- per band E (its own value), the frame's total power SP, the band's peak weight:
        d ln E ~ K * eps * sqrt(SP * wpeak / E),   flag when K * eps * sqrt(SP * wpeak / E) > bound
    -- the smallest K that leaves no unflagged band above the bound anywhere in the zoo, and the fraction of frames it flags.
Usage: tools/fbank_f32_calib.py [n_mels] [K ...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
d = os.path.join(ROOT, "tests", "emu")
subprocess.check_call(["make", "-C", d, "-s"])
L = C.CDLL(os.path.join(d, "libmelspec_emu.so"))
f32p = C.POINTER(C.c_float)
L.emu_fbank_wave.restype = C.c_longlong
L.emu_fbank_wave.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_float, C.c_int, C.c_int, C.c_int, f32p]
cfg = O.fbank_default_config()
cfg.num_mel_bins = n_mels
cfg.apply_cmn = 0
HIGH = cfg.high_freq if cfg.high_freq > 0 else cfg.sample_rate / 2.0
fb = O.kaldi_mel_filterbank(cfg.sample_rate, 512, n_mels, cfg.low_freq, HIGH)       # [n_mels][257]
wpeak = fb.max(axis=1)
EPS32 = float(np.finfo(np.float32).eps)


def kernel(x, f64):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else 1 + (len(x) - 400) // 160
    out = np.zeros((nf, n_mels), np.float32)
    got = L.emu_fbank_wave(x.ctypes.data_as(f32p), len(x), 160, n_mels, cfg.sample_rate, cfg.low_freq, HIGH, cfg.preemphasis,
                           EPS32, 1, 1, 1 if f64 else 0, out.ctypes.data_as(f32p))
    assert got == nf, (got, nf)
    return out


def frame_powers(x):
    """The f64 power spectrum of every Kaldi frame (src/fbank.rs:164-203), for SP."""
    x = np.asarray(x, np.float64)
    nf = 1 + (len(x) - 400) // 160
    idx = np.arange(400)[None, :] + 160 * np.arange(nf)[:, None]
    fr = x[idx]
    m = fr.mean(axis=1, keepdims=True)
    prev = np.where(idx > 0, x[np.maximum(idx - 1, 0)], 0.0)
    y = fr - m
    pe = y - cfg.preemphasis * (prev - m)
    pe[0, 0] = y[0, 0]
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 399.0)) ** 0.85
    X = np.fft.rfft(pe * w, 512, axis=1)
    return np.abs(X) ** 2


sys.argv = sys.argv[:1]
src = open(os.path.join(ROOT, "tools", "flag_calib2.py")).read()
ns = {"__name__": "fc2", "__file__": os.path.join(ROOT, "tools", "flag_calib2.py")}
exec(compile(src.split("def zoo_signals():")[0], "fc2-head", "exec"), ns)
exec(compile("def zoo_signals():" + src.split("def zoo_signals():")[1].split("\n\n\n")[0], "fc2-zoo", "exec"), ns)
zoo = ns["zoo_signals"]()
# BASELINE config 3's own workload: the hash-noise clips (every amplitude class, clip & 7), 10 s each
zoo = {"cfg3 synth": [O.synth_pcm(c, 160000) for c in range(16)], **zoo}

print(f"Kaldi fbank, {n_mels} bins, 25 ms / 10 ms @ 16 kHz, 512-point FFT, pre-emphasis {cfg.preemphasis}, ln(max(E, eps)), no CMN")
print(f"{'class':14s} {'frames':>7s}  {'f64 build worst':>15s}  {'f32 build worst':>15s}  {'frames > 5e-5':>13s}  {'frames > 1e-4':>13s}  {'frames > 1e-3':>13s}")
recs = []
for name, sigs in zoo.items():
    w64 = w32 = 0.0
    nfr = n5 = n4 = n3 = 0
    for x in sigs:
        x = x[: 16000 * 2] if name not in ("jfk", "cfg3 synth") else x
        want = O.fbank_compute(x, cfg)
        g64 = kernel(x, True)
        g32 = kernel(x, False)
        e64 = np.abs(g64.astype(np.float64) - want)
        e32 = np.abs(g32.astype(np.float64) - want)
        w64 = max(w64, float(e64.max())); w32 = max(w32, float(e32.max()))
        pf = e32.max(axis=1)
        nfr += pf.size; n5 += int((pf > 5e-5).sum()); n4 += int((pf > 1e-4).sum()); n3 += int((pf > 1e-3).sum())
        P = frame_powers(x)
        SP = P.sum(axis=1)
        E = P @ fb.T                                                     # [frames][n_mels]
        pred = EPS32 * np.sqrt(SP[:, None] * wpeak[None, :] / np.maximum(E, 1e-300))
        pred = np.where(E > 0, pred, 0.0)
        recs.append((name, pred, e32))
    print(f"{name:14s} {nfr:7d}  {w64:15.2e}  {w32:15.2e}  {n5 / nfr * 100:12.1f}%  {n4 / nfr * 100:12.1f}%  {n3 / nfr * 100:12.1f}%", flush=True)

for bound in (5e-5, 1e-4):
    K = max(float((e[e > bound] / np.maximum(p[e > bound], 1e-300)).max()) if (e > bound).any() else 0.0 for _, p, e in recs)
    for KK in [K] + [float(a) for a in []]:
        print(f"\nrealisable guard, bound {bound:g}: K = {KK:.2f} (the smallest that flags every band above the bound)")
        agg = {}
        for name, p, e in recs:
            fl = (KK * p > bound).any(axis=1)
            a = agg.setdefault(name, [0, 0]); a[0] += fl.size; a[1] += int(fl.sum())
        for name, (n, f) in agg.items():
            print(f"  {name:14s} {f / n * 100:6.2f} % of {n} frames flagged")
# what fixed K values would do (worst unflagged error, frames flagged)
print("\nfixed K: worst error among unflagged bands | frames flagged per class (bound 5e-5)")
for KK in (0.25, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0):
    worst = 0.0
    agg = {}
    for name, p, e in recs:
        flb = KK * p > 5e-5
        fl = flb.any(axis=1)
        safe = ~fl[:, None] & np.ones_like(flb)
        if safe.any():
            worst = max(worst, float(e[safe].max()))
        a = agg.setdefault(name, [0, 0]); a[0] += fl.size; a[1] += int(fl.sum())
    print(f"  K = {KK:4.2f}: worst unflagged {worst:.2e} | " + "  ".join(f"{k}:{f / n * 100:.2f}%" for k, (n, f) in agg.items()))
