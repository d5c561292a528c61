#!/usr/bin/env python3
"""Latency / throughput of the streaming bank: S live streams, one chunk per stream per push (device producer)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M

m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
for n_streams, chunk in ((1, 160), (256, 160), (4096, 160), (4096, 1600), (16384, 1600), (65536, 160)):
    bank = M.StreamBank(m, n_streams, chunk)
    ids = np.arange(n_streams, dtype=np.uint32)
    lens = np.full(n_streams, chunk, np.uint32)
    out = M.DeviceBuffer(n_streams * (chunk // 160 + 1) * 80 * 4)
    p0 = bank.input_ptr(0)
    slot = (bank.input_ptr(1) - p0) // 4 if n_streams > 1 else 0
    k = 0
    for _ in range(8):
        M.synth_pcm_window(p0, slot, chunk, k * chunk, n_streams); bank.push_device(ids, lens, out.ptr); k += 1
    iters = 50
    dt = 0.0
    for _ in range(iters):
        # the producer runs on the null stream: finished before the push is timed (the legacy null-stream ordering against the
        # context's stream is a host-side wait in the runtime and would be measured instead of the push)
        M.synth_pcm_window(p0, slot, chunk, k * chunk, n_streams); M.device_synchronize()
        t0 = time.perf_counter(); bank.push_device(ids, lens, out.ptr); dt += time.perf_counter() - t0; k += 1
    dt /= iters
    frames = n_streams * (chunk // 160)
    print(f"streams {n_streams:6d} chunk {chunk:5d}: {dt * 1e3:8.3f} ms per push  {frames / dt / 1e6:8.2f} M frames/s  "
          f"({chunk / 16000.0 / dt:8.1f}x realtime per stream)", flush=True)
    out.free(); bank.close()
