#!/usr/bin/env python3
"""The kernels either side of the headline path, event-timed on resident data with their own algorithmic bytes per unit
(DESIGN.md section 7b): STFT export (f32 / f64, half / full), the mel stage on its own, the generic in-LDS FFT kernel at
n_fft 1024 / 2048, the TGA quantiser, the VAD stencil, a streaming push of 4096 streams x 1 hop, the Kaldi / NeMo / Whisper-512 /
F64 / mel-major kernels.  Run under `rocprofv3 --kernel-trace --stats` (and --pmc FETCH_SIZE / WRITE_SIZE) for the per-kernel view;
tools/profile_aux.sh does all three.  Prints one line per case: ms, algorithmic bytes, GB/s, fraction of 8 TB/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from mel_spec_amd._lib import lib
import ctypes as C

PEAK = 8000.0
only = set(sys.argv[1:])
ITERS = int(os.environ.get("AUX_ITERS", "50"))


def timed(fn, sync, iters=ITERS, warm=10):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    sync()
    return (time.perf_counter() - t0) / iters * 1e3


def report(name, ms, frames, bytes_per_frame, note=""):
    gbs = frames * bytes_per_frame / (ms * 1e-3) / 1e9
    print(f"{name:58s} {ms:8.4f} ms  {frames / ms / 1e6:7.3f} G frames/s  {bytes_per_frame:6.0f} B/frame  {gbs:7.0f} GB/s  frac {gbs / PEAK:.3f}  {note}", flush=True)


def want(k):
    return not only or k in only


n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips)
M.device_synchronize()

if want("stft"):
    m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
    nf = m.num_frames(clip_len)
    clips = 256                       # 256 x 998 frames x 6.4 KB (full, f64) = 1.6 GB
    for f64 in (False, True):
        for full in (False, True):
            bins = m.stft_bins(full)
            el = 16 if f64 else 8
            out = M.DeviceBuffer(clips * nf * bins * el)
            ms = timed(lambda: m.stft_uniform_device(pcm.ptr, clip_len, clip_len, clips, out.ptr, f64, full), m.synchronize, iters=20, warm=3)
            report(f"whisper400_stft_kernel<{'double' if f64 else 'float'}> {'full 400' if full else 'half 201'} bins", ms, clips * nf, 640 + bins * el)
            out.free()
    # the mel stage on its own: f64 half spectra in, mel rows out
    bins = m.stft_bins(False)
    spec = M.DeviceBuffer(clips * nf * bins * 16)
    m.stft_uniform_device(pcm.ptr, clip_len, clip_len, clips, spec.ptr, True, False); m.synchronize()
    mel = M.DeviceBuffer(clips * nf * 80 * 4)
    ms = timed(lambda: m.mel_from_stft_device(spec.ptr, clips * nf, mel.ptr, True, False), m.synchronize, iters=20, warm=3)
    report("mel_stage_jobs_kernel<double> (half f64 spectra -> mel rows)", ms, clips * nf, bins * 16 + 320)
    spec.free(); mel.free(); m.close()

if want("generic"):
    for fft, hop, clips in ((256, 64, 1024), (1024, 256, 1024), (2048, 512, 256), (800, 200, 64)):
        g = M.HipMelSpectrogram(fft, hop, 16000.0, 80)
        nf = g.num_frames(clip_len)
        out = M.DeviceBuffer(clips * nf * 80 * 4)
        ms = timed(lambda: g.compute_uniform_device(pcm.ptr, clip_len, clip_len, clips, out.ptr), g.synchronize, iters=10, warm=2)
        name = "pow2_frame_kernel" if fft & (fft - 1) == 0 else "generic_frame_kernel"
        how = "frames owned by lane groups of a wave" if fft & (fft - 1) == 0 else "in-LDS FFT, one frame per workgroup"
        report(f"{name} n_fft {fft} hop {hop}, {clips} clips ({how})", ms, clips * nf, hop * 4 + 320)
        out.free(); g.close()

if want("quant") or want("vad"):
    m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
    W = m.interleaved_width(clip_len, 2)
    img = M.DeviceBuffer(n_clips * 80 * W * 4)
    ms = timed(lambda: m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, img.ptr, False, 2), m.synchronize)
    report("whisper400_six_kernel mel-major [80][W] store (interleave_frames)", ms, n_clips * m.num_frames(clip_len), 960)
    px = n_clips * 80 * W
    if want("quant"):
        q = M.TgaCodec()
        n, stride, last = q.layout(80, W)
        blobs = M.DeviceBuffer(n_clips * stride)
        back = M.DeviceBuffer(n_clips * 80 * W * 4)
        ms = timed(lambda: q.encode_device(img.ptr, 80 * W, 80, W, n_clips, blobs.ptr, stride), q.synchronize)
        report("quant_minmax_kernel + quant_encode_kernel (5 B/pixel)", ms, px, 5, "per PIXEL")
        ms = timed(lambda: q.decode_device(blobs.ptr, stride, 80, W, n_clips, back.ptr, 80 * W), q.synchronize)
        report("quant_decode_kernel (5 B/pixel)", ms, px, 5, "per PIXEL")
        ms2 = timed(lambda: q.encode_pcm_uniform_device(m, pcm.ptr, clip_len, clip_len, n_clips, 2, img.ptr, blobs.ptr, stride), m.synchronize)
        report("PCM -> TGA fused: mel-major kernel leaving unit extremes + keys + quant_encode_kernel", ms2, n_clips * m.num_frames(clip_len), 640 + 80, "PCM in, bytes out (720 B/frame); the image itself is an intermediate")
        blobs.free(); back.free(); q.close()
    if want("vad"):
        n = int(lib().melspec_vad_mask_len(80, W))
        raw = M.DeviceBuffer(n_clips * n); sm = M.DeviceBuffer(n_clips * n); runs = M.DeviceBuffer(n_clips * 4)
        st = M.DetectionSettings(1.0, 10, 10, 0)._c()
        call = lambda: lib().melspec_vad_boundaries_device(C.c_void_p(img.ptr), 80 * W, 80, W, n_clips, C.byref(st), C.c_void_p(raw.ptr), C.c_void_p(sm.ptr), n,
                                                           C.c_void_p(runs.ptr), None)
        assert call() == 0
        ms = timed(call, M.device_synchronize)
        report("vad_raw_kernel + vad_smooth_kernel + vad_run_kernel (4 B/pixel read)", ms, px, 4, "per PIXEL")
        raw.free(); sm.free(); runs.free()
    img.free(); m.close()

if want("stream"):
    m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
    for n_streams, chunk, vad in ((4096, 160, False), (4096, 160, True), (4096, 1600, False), (4096, 1600, True)):
        bank = M.StreamBank(m, n_streams, chunk)
        acts = M.DeviceBuffer(n_streams * (chunk // 160 + 1) * 8)
        if vad:
            bank.enable_vad(M.DetectionSettings())
        push = (lambda: bank.push_device_vad(ids, lens, out.ptr, acts.ptr)) if vad else (lambda: bank.push_device(ids, lens, out.ptr))
        ids = np.arange(n_streams, dtype=np.uint32)
        lens = np.full(n_streams, chunk, np.uint32)
        out = M.DeviceBuffer(n_streams * (chunk // 160 + 1) * 80 * 4)
        p0 = bank.input_ptr(0)
        slot = (bank.input_ptr(1) - p0) // 4
        k = 0
        for _ in range(8):
            M.synth_pcm_window(p0, slot, chunk, k * chunk, n_streams); push(); k += 1
        dt = 0.0
        for _ in range(ITERS):
            M.synth_pcm_window(p0, slot, chunk, k * chunk, n_streams); M.device_synchronize()
            t0 = time.perf_counter(); push(); dt += time.perf_counter() - t0; k += 1
        ms = dt / ITERS * 1e3
        report(f"stream push {n_streams} streams x {chunk // 160} hop(s) (plan + frames{' + detector stage' if vad else ''} + carry, host-inclusive)", ms,
               n_streams * (chunk // 160), 960)
        out.free(); acts.free(); bank.close()
    m.close()

if want("flavours"):
    m = M.HipMelSpectrogram(400, 160, 16000.0, 80)
    nf = m.num_frames(clip_len)
    out = M.DeviceBuffer(n_clips * (nf + 8) * 128 * 4)
    for mode in ("auto", "f64"):
        m.set_precision(mode)
        ms = m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=20, iters=100)
        report(f"whisper400 80 mels, precision {mode} ({m.plain_kernel_name()[:48]})", ms, n_clips * nf, 960)
    m.close()
    w = M.HipMelSpectrogram(512, 160, 16000.0, 80)
    ms = w.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=10, iters=50)
    report("fbank512_wave_kernel Whisper flavour n_fft 512 (f64)", ms, n_clips * w.num_frames(clip_len), 960)
    w.close()
    fb = M.Fbank()
    ms = timed(lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fb.synchronize)
    report("fbank512_clip_kernel Kaldi 80 bins + CMN (config 3)", ms, n_clips * fb.num_frames(clip_len), 960)
    fb.close()
    for norm in (False, True):
        fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97, normalize_per_feature=norm))
        ms = timed(lambda: fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fe.synchronize)
        report(f"NeMo 128 mels{' + normalize_per_feature' if norm else ''}", ms, n_clips * fe.num_frames(clip_len), 640 + 512)
        fe.close()
    out.free()
pcm.free()
