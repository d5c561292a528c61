#!/usr/bin/env python3
"""Padded layouts of the six-frame kernel, both orders (major_column_order True = [W][mel] rows, False = mel-major), wall ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mel_spec_amd as M
n, cl = 1024, 160000
pcm = M.DeviceBuffer(n * cl * 4); M.synth_pcm_device(pcm.ptr, cl, cl, 0, n); M.device_synchronize()
m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
if os.environ.get("AB_NOVOTE"): m.set_auto_adaptive(False)
W = m.interleaved_width(cl, 1200)
out = M.DeviceBuffer(n * W * 80 * 4)
for mco in (True, False):
    f = lambda: m.compute_uniform_device_interleaved(pcm.ptr, cl, cl, n, out.ptr, mco, 1200)
    for _ in range(50): f()
    m.synchronize(); t = time.perf_counter()
    for _ in range(200): f()
    m.synchronize(); print(os.path.basename(os.environ.get("MELSPEC_LIB", "")), "major_column_order", mco, f"{(time.perf_counter() - t) / 200 * 1e3:.4f} ms")
