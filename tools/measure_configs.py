#!/usr/bin/env python3
"""Times the BASELINE.json configs on one GPU (device-resident, wall clock around N launches + sync)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O

def timed(fn, sync, iters):
    t0 = time.perf_counter()                      # clocks ramp over the first tens of milliseconds of work
    while time.perf_counter() - t0 < 0.3:
        for _ in range(max(2, iters // 10)): fn()
        sync()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    sync()
    return (time.perf_counter() - t0) / iters

res = {}
def run_mel(tag, n_clips, clip_len, n_mels, iters):
    m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
    fpc = m.num_frames(clip_len)
    pcm = M.DeviceBuffer(n_clips * clip_len * 4); out = M.DeviceBuffer(n_clips * fpc * n_mels * 4)
    M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
    # spin up clocks
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr); m.synchronize()
    dt = timed(lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), m.synchronize, iters)
    frames = n_clips * fpc
    worst = 0.0
    for c in (0, n_clips - 1):
        got = out.download((fpc, n_mels), offset_bytes=c * fpc * n_mels * 4)
        worst = max(worst, float(np.abs(got - O.compute_mel_spectrogram_cpu(O.synth_pcm(c, clip_len), 400, 160, n_mels)).max()))
    res[tag] = dict(ms=dt * 1e3, frames=frames, gframes_per_s=frames / dt / 1e9,
                    algo_GBps=frames * (640 + 4 * n_mels) / dt / 1e9, parity=worst)
    print(tag, res[tag], flush=True)
    pcm.free(); out.free(); m.close()

def run_fbank(tag, n_clips, clip_len, iters):
    fb = M.Fbank()
    fpc = fb.num_frames(clip_len)
    pcm = M.DeviceBuffer(n_clips * clip_len * 4); out = M.DeviceBuffer(n_clips * fpc * 80 * 4)
    M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
    dt = timed(lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fb.synchronize, iters)
    frames = n_clips * fpc
    worst = 0.0
    for c in (0, n_clips - 1):
        got = out.download((fpc, 80), offset_bytes=c * fpc * 80 * 4)
        worst = max(worst, float(np.abs(got - O.fbank_compute(O.synth_pcm(c, clip_len))).max()))
    res[tag] = dict(ms=dt * 1e3, frames=frames, gframes_per_s=frames / dt / 1e9, algo_GBps=frames * 960 / dt / 1e9, parity=worst)
    print(tag, res[tag], flush=True)
    pcm.free(); out.free(); fb.close()

def run_nemo(tag, n_clips, clip_len, n_mels, iters, norm=False, f32=False):
    fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=n_mels, preemphasis=0.97, log_zero_guard=2.0 ** -24, normalize_per_feature=norm))
    if f32: fe.set_precision("f32")        # the reference's own arithmetic type for this frontend: gated at 2e-4 on the bench's noise clips below
    cols = fe.padded_frames(clip_len)
    pcm = M.DeviceBuffer(n_clips * clip_len * 4); out = M.DeviceBuffer(n_clips * cols * n_mels * 4)
    M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
    dt = timed(lambda: fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fe.synchronize, iters)
    frames = n_clips * cols
    cfg = O.blm_default_config(n_mels=n_mels, preemphasis=0.97, log_zero_guard=2.0 ** -24, normalize_per_feature=norm)
    worst = 0.0
    for c in (0, n_clips - 1):
        got = out.download((n_mels, cols), offset_bytes=c * cols * n_mels * 4)
        worst = max(worst, float(np.abs(got - O.blm_compute(O.synth_pcm(c, clip_len), cfg, True)[0]).max()))
    res[tag] = dict(ms=dt * 1e3, frames=frames, gframes_per_s=frames / dt / 1e9, algo_GBps=frames * (640 + 4 * n_mels) / dt / 1e9, parity=worst)
    print(tag, res[tag], flush=True)
    pcm.free(); out.free(); fe.close()

which = sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5", "nemo", "host"]
if "cfg2" in which: run_mel("cfg2_w80_1024x10s", 1024, 160000, 80, 200)
if "cfg3" in which: run_fbank("cfg3_fbank_1024x10s", 1024, 160000, 50)
if "cfg4" in which: run_mel("cfg4_w128_8192x30s", 8192, 480000, 128, 10)
if "cfg5" in which: run_mel("cfg5_w80_8192x30s_per_gpu_share", 8192, 480000, 80, 10)
if "nemo" in which: run_nemo("nemo_parakeet_128_1024x10s", 1024, 160000, 128, 50)
if "nemo" in which: run_nemo("nemo_parakeet_128_1024x10s_normalised", 1024, 160000, 128, 50, True)
if "nemo" in which: run_nemo("nemo_parakeet_128_1024x10s_F32", 1024, 160000, 128, 50, False, True)
if "nemo" in which: run_nemo("nemo_parakeet_128_1024x10s_normalised_F32", 1024, 160000, 128, 50, True, True)
if "host" in which:
    m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
    x = np.concatenate([O.synth_pcm(c, 160000) for c in range(64)])
    m.compute_mel_spectrogram(x)
    t0 = time.perf_counter()
    for _ in range(10): y = m.compute_mel_spectrogram(x)
    dt = (time.perf_counter() - t0) / 10
    res["host_api_64x10s_pcie_inclusive"] = dict(ms=dt * 1e3, frames=int(y.shape[0]), gframes_per_s=y.shape[0] / dt / 1e9)
    print("host", res["host_api_64x10s_pcie_inclusive"], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
