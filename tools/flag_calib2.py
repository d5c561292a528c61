#!/usr/bin/env python3
"""Per-band calibration of the refined precision flag (CPU only: host emulation of the f32 kernel vs the f64 oracle).

The f32 FFT leaves rounding noise of about eps * sqrt(sum_k P_k) in every bin, so the relative error of a mel band is
    relE_m ~ K * eps * sqrt(SP * sum_k w_m(k)^2 P_k) / E_m  <=  K * eps * sqrt(SP * wpeak_m / E_m),     SP = sum_k P_k.
Statistic per band: q_m = log10(E_m) - log10(SP * wpeak_m).  A band is hazardous when it is above the clamp and q_m < Q.
Prints, per Q, the worst error among bands that are not flagged and the fraction of FRAMES with at least one flagged band,
per signal class; the same for the round-2 rule (any band within 2 decades of the clamp).
Usage: tools/flag_calib2.py [n_mels] [hop]"""
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
sr = 16000.0

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


def zoo_signals():
    rng = np.random.default_rng(5)
    N = 16000 * 4
    t = np.arange(N) / sr

    def tone_floor(f, level_db, amp=0.9):
        return (amp * np.sin(2 * np.pi * f * t) + 10 ** (level_db / 20) * rng.standard_normal(N)).astype(np.float32)

    zoo = {}
    zoo["jfk"] = [O.load_wav_f32(os.path.join(ROOT, "tests", "golden", "jfk_f32le.wav"))]
    zoo["noise"] = [rng.standard_normal(N).astype(np.float32) * np.float32(10.0 ** rng.uniform(-4, 0)) for _ in range(4)] + [O.synth_pcm(c, N) for c in range(4)]
    zoo["tone"] = [(np.sin(np.arange(N) * rng.uniform(0.01, 3.0)) * rng.uniform(0.01, 1.0)).astype(np.float32) for _ in range(12)]
    zoo["tone+floor"] = [tone_floor(f, lv, a) for f in (60.0, 200.0, 1000.0, 3333.3, 5000.0, 7000.0, 7800.0) for lv in (-40, -50, -60, -65, -70, -75, -80, -90) for a in (0.9, 0.05)]
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
        two.append((0.9 * np.sin(2 * np.pi * 6000.0 * t) + 10 ** (lv / 20) * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32))
    zoo["two tones"] = two
    dc = []
    for lv in (-40, -60, -80):
        dc.append((0.5 + 10 ** (lv / 20) * rng.standard_normal(N)).astype(np.float32))
        dc.append((0.5 + 0.3 * np.sin(2 * np.pi * 7900.0 * t) + 10 ** (lv / 20) * rng.standard_normal(N)).astype(np.float32))
    zoo["dc+floor"] = dc
    j = zoo["jfk"][0].astype(np.float64)
    J = np.fft.rfft(j)
    fr = np.fft.rfftfreq(len(j), 1 / 16000.0)
    lp = []
    for cut, att in ((3400, 1e-4), (4000, 1e-5), (3400, 0.0)):
        g = np.where(fr < cut, 1.0, att)
        lp.append(np.fft.irfft(J * g, len(j)).astype(np.float32))
    zoo["jfk lowpass"] = lp
    zoo["jfk quiet"] = [(zoo["jfk"][0] * np.float32(s)).astype(np.float32) for s in (0.01, 1e-4)]
    return zoo


def main():
    fb = O.mel_filterbank(sr, 400, n_mels)             # [n_mels][201]
    wpeak = fb.max(axis=1)
    zoo = zoo_signals()
    cls, E_all, Q_all, R_all, F_all = [], [], [], [], []
    fid = 0
    for k, sigs in zoo.items():
        for x in sigs:
            got = f32_kernel(x)
            want = O.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)
            X = O.compute_all_cpu(x, 400, hop)        # [frames][400] complex
            P = (np.abs(X[:, :201]) ** 2)
            SP = P[:, :200].sum(axis=1) + 1e-300
            Em = P[:, :200] @ fb[:, :200].T           # [frames][n_mels]
            v = np.log10(np.maximum(Em, 1e-10))
            mx = v.max(axis=1, keepdims=True)
            e = np.abs(got.astype(np.float64) - want)
            q = v - np.log10(SP[:, None] * wpeak[None, :])
            r = v - mx
            nf = got.shape[0]
            cls.append(np.full(nf * n_mels, k))
            E_all.append(e.ravel()); Q_all.append(q.ravel()); R_all.append(r.ravel())
            F_all.append((fid + np.repeat(np.arange(nf), n_mels)))
            fid += nf
    cls = np.concatenate(cls); e = np.concatenate(E_all); q = np.concatenate(Q_all); r = np.concatenate(R_all); f = np.concatenate(F_all)
    nfr = fid
    fcls = np.empty(nfr, dtype=cls.dtype); fcls[f] = cls
    uncl = r > -8.0 - 0.05          # bands at or above the clamp (a hair below counts: the decision itself is noisy)
    print(f"n_mels={n_mels} hop={hop}: {nfr} frames, {len(e)} bands, worst f32 error {e.max():.3e}")

    def report(name, flag_band):
        fl_frame = np.zeros(nfr, bool)
        np.logical_or.at(fl_frame, f, flag_band)
        safe = ~fl_frame[f]          # bands of frames that are not recomputed
        worst = e[safe].max() if safe.any() else 0.0
        frs = "  ".join(f"{k}:{fl_frame[fcls == k].mean():.3f}" for k in zoo)
        print(f"  {name:28s} worst unflagged {worst:.2e} | {frs}")

    print("round-2 rule (band within T decades of the clamp):")
    for T in (2.0, 1.5, 1.25, 1.0):
        report(f"r < -8 + {T}", uncl & (r < -8.0 + T))
    print("refined rule (q = log10 E_m - log10(SP * wpeak_m) < Q, band at or above the clamp):")
    for Q in (-5.5, -6.0, -6.25, -6.5, -6.75, -7.0, -7.25, -7.5):
        report(f"q < {Q}", uncl & (q < Q))
    print("both (band within 2 decades of the clamp AND q < Q):")
    for Q in (-6.0, -6.5, -7.0):
        report(f"r < -6 and q < {Q}", uncl & (r < -6.0) & (q < Q))
    print("max error by q bin (bands at or above the clamp):")
    for lo in np.arange(-9, -3, 0.25):
        m = uncl & (q >= lo) & (q < lo + 0.25)
        if m.any():
            print(f"  [{lo:6.2f},{lo + 0.25:6.2f})  n={m.sum():8d}  max e {e[m].max():.2e}  p99.9 {np.quantile(e[m], 0.999):.2e}  median {np.median(e[m]):.2e}")


if __name__ == "__main__":
    main()
