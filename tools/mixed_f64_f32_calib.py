#!/usr/bin/env python3
"""VERDICT r03 "next" 1(c): mixed precision INSIDE the transform -- f64 through window, first DFT-20 and twiddle, f32 for the second
DFT-10 and the Hermitian split (a dominant line then leaks rounding noise into its own residue class mod 20 only).  Would it hold 5e-5?
CPU only: tests/emu's emu_whisper_mixed (the precise kernel's phase 1 + an f32 phase 2) against the oracle over jfk_f32le.wav and the
signal zoo of tools/flag_calib2.py, next to the pure-f32 kernel and the f64 kernel.  Usage: tools/mixed_f64_f32_calib.py [n_mels]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
hop, sr = 160, 16000.0
d = os.path.join(ROOT, "tests", "emu")
subprocess.check_call(["make", "-C", d, "-s"])
L = C.CDLL(os.path.join(d, "libmelspec_emu.so"))
f32p = C.POINTER(C.c_float)
for fn in (L.emu_whisper_mixed, L.emu_whisper_precise):
    fn.restype = C.c_longlong
    fn.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, f32p]
L.emu_whisper_six.restype = C.c_longlong
L.emu_whisper_six.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p]

def run(fn, x, extra=()):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    got = fn(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, *extra, out.ctypes.data_as(f32p))
    assert got == nf
    return out

sys.argv = sys.argv[:1]
src = open(os.path.join(ROOT, "tools", "flag_calib2.py")).read()
ns = {"__name__": "fc2", "__file__": os.path.join(ROOT, "tools", "flag_calib2.py")}
exec(compile(src.split("def zoo_signals():")[0], "fc2-head", "exec"), ns)
exec(compile("def zoo_signals():" + src.split("def zoo_signals():")[1].split("\n\n\n")[0], "fc2-zoo", "exec"), ns)
zoo = ns["zoo_signals"]()
print(f"Whisper 400/160/{n_mels}: worst |difference| against the oracle and the fraction of FRAMES with a band off by more than 5e-5")
print(f"{'class':14s} {'frames':>7s}   {'f64 kernel':>10s}   {'mixed: f64 stage 1, f32 stage 2':>34s}   {'f32 kernel':>22s}")
tot = [0, 0, 0]
for name, sigs in zoo.items():
    w = [0.0, 0.0, 0.0]; nfr = 0; bad = [0, 0]
    for x in sigs:
        x = x[: 16000 * 2] if name != "jfk" else x
        want = O.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)
        e64 = np.abs(run(L.emu_whisper_precise, x) - want)
        emx = np.abs(run(L.emu_whisper_mixed, x) - want)
        e32 = np.abs(run(L.emu_whisper_six, x, (0,)) - want) if n_mels <= 80 else emx * np.nan
        w = [max(w[0], e64.max()), max(w[1], emx.max()), max(w[2], np.nanmax(e32))]
        nfr += want.shape[0]; bad[0] += int((emx.max(axis=1) > 5e-5).sum()); bad[1] += int((e32.max(axis=1) > 5e-5).sum())
    tot = [tot[0] + nfr, tot[1] + bad[0], tot[2] + bad[1]]
    print(f"{name:14s} {nfr:7d}   {w[0]:10.2e}   {w[1]:12.2e}  {bad[0] / nfr * 100:6.1f} % of the frames   {w[2]:10.2e}  {bad[1] / nfr * 100:6.1f} %", flush=True)
print(f"{'all':14s} {tot[0]:7d}   {'':10s}   {'':12s}  {tot[1] / tot[0] * 100:6.1f} % of the frames   {'':10s}  {tot[2] / tot[0] * 100:6.1f} %")
