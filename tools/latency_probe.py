#!/usr/bin/env python3
"""Latency of the drop-in host call (CudaMelSpectrogram::compute_mel_spectrogram twin) for one clip, PCIe and synchronisation included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O
for n_mels in (80, 128):
    m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
    for secs in (1, 10, 30, 300):
        x = O.synth_pcm(1, 16000 * secs)
        for _ in range(5): m.compute_mel_spectrogram(x)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); y = m.compute_mel_spectrogram(x); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{n_mels} mels, {secs:4d} s clip: median {ts[len(ts)//2]*1e3:.3f} ms  min {ts[0]*1e3:.3f} ms  ({y.shape[0]} frames, {y.shape[0]/ts[len(ts)//2]/1e6:.1f} M frames/s, {secs/ts[len(ts)//2]:.0f}x realtime)")
    m.close()
