#!/usr/bin/env python3
"""pow2_frame_kernel (round 4): the power-of-two frame sizes off the 400- / 512-point kernels -- Whisper-style log-mel at n_fft 128 / 256 /
1024 / 2048 and Kaldi fbank at 8 / 32 / 44.1 kHz (fft sizes 256 / 1024 / 2048, src/fbank.rs:66-82): ms per launch, frames/s, achieved
algorithmic GB/s (hop * 4 + n_mels * 4 bytes per frame) against 8 TB/s, parity against the oracle.  1024 clips of 10 s of their rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O

def timed(run, sync, iters=10):
    for _ in range(3): run()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters): run()
    sync()
    return (time.perf_counter() - t0) / iters

n_clips = 1024
for n_fft, hop, n_mels, sr in ((128, 64, 40, 8000.0), (256, 64, 40, 8000.0), (256, 128, 80, 16000.0), (1024, 256, 80, 16000.0), (1024, 160, 80, 16000.0), (2048, 512, 128, 44100.0)):
    clip_len = int(10 * sr)
    pcm = M.DeviceBuffer(n_clips * clip_len * 4)
    M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
    m = M.HipMelSpectrogram(n_fft, hop, sr, n_mels)
    nf = m.num_frames(clip_len)
    out = M.DeviceBuffer(n_clips * nf * n_mels * 4)
    dt = timed(lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), m.synchronize)
    got = out.download((nf, n_mels))
    d = float(np.abs(got - O.compute_mel_spectrogram_cpu(O.synth_pcm(0, clip_len), n_fft, hop, n_mels, sr)).max())
    fr = n_clips * nf
    print(f"log-mel n_fft {n_fft:5d} hop {hop:4d} mels {n_mels:3d} @ {sr / 1000:5.1f} kHz: {dt * 1e3:8.3f} ms  {fr / dt / 1e9:6.3f} G frames/s  {fr * (hop + n_mels) * 4 / dt / 1e9:7.1f} GB/s = {fr * (hop + n_mels) * 4 / dt / 8e12 * 100:5.2f} % of 8 TB/s  parity {d:.1e}", flush=True)
    out.free(); m.close(); pcm.free()
for sr, bins in ((8000.0, 40), (32000.0, 80), (44100.0, 80)):
    clip_len = int(10 * sr)
    pcm = M.DeviceBuffer(n_clips * clip_len * 4)
    M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
    cfg = M.FbankConfig(sample_rate=sr, num_mel_bins=bins)
    fb = M.Fbank(cfg)
    nf = fb.num_frames(clip_len)
    out = M.DeviceBuffer(n_clips * nf * bins * 4)
    dt = timed(lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fb.synchronize)
    got = out.download((nf, bins))
    oc = O.fbank_default_config(); oc.sample_rate = sr; oc.num_mel_bins = bins
    d = float(np.abs(got - O.fbank_compute(O.synth_pcm(0, clip_len), oc)).max())
    fr, shift = n_clips * nf, cfg.frame_shift_samples()
    print(f"Kaldi fbank {sr / 1000:5.1f} kHz fft {cfg.fft_size():5d} bins {bins:3d} (+ CMN): {dt * 1e3:8.3f} ms  {fr / dt / 1e9:6.3f} G frames/s  {fr * (shift + bins) * 4 / dt / 1e9:7.1f} GB/s = {fr * (shift + bins) * 4 / dt / 8e12 * 100:5.2f} % of 8 TB/s  parity {d:.1e}", flush=True)
    out.free(); fb.close(); pcm.free()
