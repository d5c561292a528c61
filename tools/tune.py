#!/usr/bin/env python3
"""Interleaved A/B timing of the fused kernel variants (MELSPEC_VARIANT) on the bench workload,
each checked against the oracle on a few clips first.  Usage: tools/tune.py [variants...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O

variants = [a for a in sys.argv[1:] if not a.startswith("-")] or ["0", "1", "2", "3", "4", "5", "6"]
n_mels = int(os.environ.get("TUNE_MELS", "80"))
n_clips, clip_len = int(os.environ.get("TUNE_CLIPS", "1024")), int(os.environ.get("TUNE_CLIP_LEN", "160000"))
rounds = int(os.environ.get("TUNE_ROUNDS", "5"))
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips)
M.device_synchronize()
ctxs = {}
for v in variants:
    # "8p" = variant 8 precise build; "8g2" = at most 2 workgroups per CU in the grid
    v0 = v
    os.environ["MELSPEC_GRID_PER_CU"] = "4"
    if "g" in v0:
        v0, g = v0.split("g"); os.environ["MELSPEC_GRID_PER_CU"] = g
    vp, prec = (v0[:-1], True) if v0.endswith("p") else (v0, False)
    vv, rt = (vp[:-1], "1") if vp.endswith("r") else (vp, "0")
    vv, sg = (vv.split("s") + ["0"])[:2] if "s" in vv else (vv, "0")     # "8s64" = variant 8, stagger 64
    os.environ["MELSPEC_STAGGER"] = sg
    os.environ["MELSPEC_VARIANT"] = vv
    os.environ["MELSPEC_RUNTIME_LENS"] = rt
    ctxs[v] = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
    if prec:
        ctxs[v].set_precise(True)
fpc = ctxs[variants[0]].num_frames(clip_len)
out = M.DeviceBuffer(n_clips * fpc * n_mels * 4)
want = {c: O.compute_mel_spectrogram_cpu(O.synth_pcm(c, clip_len), 400, 160, n_mels) for c in (0, n_clips // 2 + 1, n_clips - 1)}
res = {v: [] for v in variants}
for v in variants:
    M.synth_pcm_device(out.ptr, 1, fpc * n_mels, 12345, n_clips)   # poison the output with noise
    ctxs[v].compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); ctxs[v].synchronize()
    worst = max(float(np.abs(out.download((fpc, n_mels), offset_bytes=c * fpc * n_mels * 4) - w).max()) for c, w in want.items())
    print(f"variant {v}: parity max|diff| = {worst:.3e}", flush=True)
    assert worst <= 1e-4, (v, worst)
for r in range(rounds):
    for v in variants:
        res[v].append(ctxs[v].time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=int(os.environ.get("TUNE_WARMUP","2")), iters=int(os.environ.get("TUNE_ITERS","10"))))
frames = n_clips * fpc
for v in variants:
    ms = np.array(res[v])
    print(f"variant {v:>3s}: median {np.median(ms):.4f} ms  min {ms.min():.4f} ms  -> {frames / np.median(ms) / 1e6:.1f} G frames/s"
          f"  ({frames * (640 + 4 * n_mels) / np.median(ms) / 1e6:.0f} GB/s algorithmic)", flush=True)
