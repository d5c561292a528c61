#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host entry points (melspec_compute_batch_host): pageable vs pinned caller memory.
Usage: tools/host_pipe_bench.py [n_clips ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O

sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 64, 256, 1024]
clip_len = 160000
m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
base = np.stack([O.synth_pcm(c, clip_len) for c in range(8)])
for n in sizes:
    x = np.ascontiguousarray(np.tile(base, ((n + 7) // 8, 1))[:n])
    nf = m.num_frames(clip_len)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(clip_len)
    lens = np.full(n, clip_len, np.uint64)
    out = np.empty(n * nf * 80, np.float32)
    hin, hout = M.HostBuffer(x.size), M.HostBuffer(out.size)
    hin.array[:] = x.reshape(-1)
    for name, a, b in (("pageable", x.reshape(-1), out), ("pinned", hin.array, hout.array)):
        m.compute_batch_host(a, offs, lens, b)
        reps = max(3, min(50, int(2e9 / (x.nbytes + out.nbytes))))
        t0 = time.perf_counter()
        for _ in range(reps):
            m.compute_batch_host(a, offs, lens, b)
        dt = (time.perf_counter() - t0) / reps
        print(f"{n:5d} x 10 s  {name:8s}  {dt * 1e3:8.3f} ms  {n * nf / dt / 1e6:8.1f} M frames/s  in {x.nbytes / dt / 1e9:6.1f} GB/s  out {out.nbytes / dt / 1e9:6.1f} GB/s", flush=True)
    hin.free(); hout.free()
m.close()
