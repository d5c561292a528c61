"""The f64 kernels of the fused paths, a few launches each (for rocprofv3 / tools/lds_pmc.sh): Whisper F64, Kaldi fbank, NeMo, Whisper 512.
tools/f64_quick.py [f64] [fbank] [nemo] [w512]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
which = sys.argv[1:] or ["f64", "fbank", "nemo", "w512"]
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
def run(name, fn, sync, frames):
    for _ in range(5): fn()
    sync(); t0 = time.perf_counter()
    for _ in range(20): fn()
    sync(); dt = (time.perf_counter() - t0) / 20
    print(f"{os.path.basename(os.environ.get('MELSPEC_LIB', 'default'))} {name}: {dt * 1e3:.3f} ms  {frames / dt / 1e9:.3f} G frames/s", flush=True)
if "f64" in which:
    m = M.HipMelSpectrogram(400, 160, 16000.0, 80); m.set_precision("f64")
    nf = m.num_frames(clip_len); out = M.DeviceBuffer(n_clips * (nf + 8) * 80 * 4)
    run("whisper F64", lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), m.synchronize, n_clips * nf)
if "w512" in which:
    m = M.HipMelSpectrogram(512, 160, 16000.0, 80)
    nf = m.num_frames(clip_len); out = M.DeviceBuffer(n_clips * (nf + 8) * 80 * 4)
    run("whisper n_fft 512", lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), m.synchronize, n_clips * nf)
if "fbank" in which:
    fb = M.Fbank()
    nf = fb.num_frames(clip_len); out = M.DeviceBuffer(n_clips * nf * 80 * 4)
    run("kaldi fbank", lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fb.synchronize, n_clips * nf)
if "nemo" in which:
    nm = M.BatchLogMelSpectrogram()
    nf = nm.num_frames(clip_len); out = M.DeviceBuffer(n_clips * (nm.padded_frames(clip_len) + 8) * 128 * 4)
    run("nemo", lambda: nm.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), nm.synchronize, n_clips * nf)
