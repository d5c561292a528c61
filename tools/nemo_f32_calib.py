#!/usr/bin/env python3
"""VERDICT r03 "next" 6: would an f32 NeMo kernel be worth building?  (CPU only.)

The reference computes BatchLogMelSpectrogram in f32 (src/mel.rs:251-252, 356-357) and pins only its shape; this library computes the
512-point FFT in f64 and is gated at 1e-4 against the f64 evaluation of the same definition.  An f32 twin of the kernel would run at
about twice the rate -- IF its output stayed inside the gate, or if the frames where it does not were few enough to hand to the f64
kernel (the AUTO scheme of the Whisper path pays off below ~14 % of the frames).

This script runs the kernel's own source instantiated in f32 (tests/emu: emu_blm_wave_f32 = fbank512_wave_kernel<float, ..., kFlavorNemo>)
and in f64 against the oracle's f64 evaluation, on jfk_f32le.wav and the signal zoo of tools/flag_calib2.py, and reports per class:
the worst |ln(E + 2^-24)| error of either build, and the fraction of FRAMES in which at least one band of the f32 build is off by more
than 5e-5 / 1e-4 -- the frames an f32 kernel would have to hand over under a PERFECT guard (any realisable guard flags more).
Usage: tools/nemo_f32_calib.py [n_mels]"""
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
for fn in (L.emu_blm_wave, L.emu_blm_wave_f32):
    fn.restype = C.c_longlong
    fn.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                   C.c_longlong, f32p]
cfg = O.blm_default_config(n_mels=n_mels)


def kernel(x, f64):
    x = np.ascontiguousarray(x, np.float32)
    want, valid = O.blm_compute(x, cfg, True)
    out = np.zeros_like(want)
    fn = L.emu_blm_wave if f64 else L.emu_blm_wave_f32
    got = fn(x.ctypes.data_as(f32p), len(x), cfg.hop_length, n_mels, cfg.sample_rate, cfg.f_min, cfg.f_max, cfg.htk, cfg.norm, cfg.preemphasis,
             cfg.center, cfg.log_zero_guard, want.shape[1], out.ctypes.data_as(f32p))
    assert got == valid, (got, valid)
    return out[:, :valid], want[:, :valid]


sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("fc2", os.path.join(ROOT, "tools", "flag_calib2.py"))
src = open(os.path.join(ROOT, "tools", "flag_calib2.py")).read()
ns = {"__name__": "fc2", "__file__": os.path.join(ROOT, "tools", "flag_calib2.py")}
exec(compile(src.split("def zoo_signals():")[0], "fc2-head", "exec"), ns)          # its imports and emu handle
exec(compile("def zoo_signals():" + src.split("def zoo_signals():")[1].split("\n\n\n")[0], "fc2-zoo", "exec"), ns)
zoo = ns["zoo_signals"]()

print(f"NeMo / Parakeet frontend, {n_mels} mels, 512/400/160, pre-emphasis {cfg.preemphasis}, ln(E + 2^-24), not normalised")
print(f"{'class':14s} {'frames':>7s}  {'f64 build worst':>15s}  {'f32 build worst':>15s}  {'frames > 5e-5':>13s}  {'frames > 1e-4':>13s}  {'frames > 1e-3':>13s}")
tot = [0, 0, 0, 0]
for name, sigs in zoo.items():
    w64 = w32 = 0.0
    nfr = n5 = n4 = n3 = 0
    for x in sigs:
        x = x[: 16000 * 2] if name != "jfk" else x
        g64, want = kernel(x, True)
        g32, _ = kernel(x, False)
        e64 = np.abs(g64 - want); e32 = np.abs(g32 - want)
        w64 = max(w64, float(e64.max())); w32 = max(w32, float(e32.max()))
        per_frame = e32.max(axis=0)
        nfr += per_frame.size; n5 += int((per_frame > 5e-5).sum()); n4 += int((per_frame > 1e-4).sum()); n3 += int((per_frame > 1e-3).sum())
    tot = [tot[0] + nfr, tot[1] + n5, tot[2] + n4, tot[3] + n3]
    print(f"{name:14s} {nfr:7d}  {w64:15.2e}  {w32:15.2e}  {n5 / nfr * 100:12.1f}%  {n4 / nfr * 100:12.1f}%  {n3 / nfr * 100:12.1f}%", flush=True)
print(f"{'all':14s} {tot[0]:7d}  {'':15s}  {'':15s}  {tot[1] / tot[0] * 100:12.1f}%  {tot[2] / tot[0] * 100:12.1f}%  {tot[3] / tot[0] * 100:12.1f}%")

# ---- a realisable guard -----------------------------------------------------------------------------------------------------------
# What a kernel can know per band without the f64 answer: E (its own value), SP = the frame's total power, the band's peak weight.
# Model of the f32 FFT's error (tools/flag_calib2.py, fitted there): dE ~ K * eps * sqrt(SP * wpeak * E)  =>  d ln(E + g) ~ dE / (E + g).
# K is calibrated on the whole zoo as the smallest value for which no unflagged band is off by more than the bound; the table then
# says how many FRAMES such a guard flags (at least one flagged band) -- the frames that would go to the f64 kernel.
fb = O.mel_filterbank(float(cfg.sample_rate), cfg.n_fft, n_mels, cfg.f_min, cfg.f_max if cfg.f_max > 0 else cfg.sample_rate / 2.0, bool(cfg.htk), bool(cfg.norm)) \
    if hasattr(O, "mel_filterbank") else None
if fb is not None:
    wpeak = fb.max(axis=1)
    eps, g = 2.0 ** -24, float(cfg.log_zero_guard)
    recs = []
    for name, sigs in zoo.items():
        for x in sigs:
            x = x[: 16000 * 2] if name != "jfk" else x
            g32, want = kernel(x, False)
            E = np.maximum(np.exp(want.astype(np.float64)) - g, 0.0)          # [mel][frame]
            # total power of the frame as the kernel would sum it: through the bank (sum of band energies / mean weight) is enough for a model
            SP = (E / np.maximum(wpeak[:, None], 1e-30)).sum(axis=0)
            pred = eps * np.sqrt(SP[None, :] * wpeak[:, None] * E) / (E + g)
            err = np.abs(g32.astype(np.float64) - want)
            recs.append((name, pred, err))
    for bound in (5e-5, 1e-4):
        K = max(float((e[e > bound] / np.maximum(p[e > bound], 1e-300)).max()) if (e > bound).any() else 0.0 for _, p, e in recs)
        # flag when K * pred > bound  (then every band with err > bound is flagged by construction)
        print(f"\nrealisable guard, bound {bound:g}: K = {K:.2f}")
        agg = {}
        for name, p, e in recs:
            fl = (K * p > bound).any(axis=0)
            a = agg.setdefault(name, [0, 0]); a[0] += fl.size; a[1] += int(fl.sum())
        for name, (n, f) in agg.items():
            print(f"  {name:14s} {f / n * 100:6.1f} % of {n} frames flagged")
