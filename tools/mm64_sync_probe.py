"""mel-major store of the f64 six-frame kernel under the sub-group barrier modes (lab build: MELSPEC_MM_SYNC), 1024 x 10 s.
usage: MELSPEC_LIB=.../libmelspec_hip_lab.so MELSPEC_MM_SYNC=<mode> python tools/mm64_sync_probe.py [f64|auto]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mel_spec_amd as M
mode = sys.argv[1] if len(sys.argv) > 1 else "f64"
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4); M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
m = M.HipMelSpectrogram(400, 160, 16000.0, 80); m.set_precision(mode)
nf = m.num_frames(clip_len); out = M.DeviceBuffer(n_clips * (nf + 8) * 80 * 4)
f = lambda: m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, False, 2)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(20): f()
    m.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(200): f()
    m.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
print("MM_SYNC", os.environ.get("MELSPEC_MM_SYNC", "default"), mode, "ms %.4f" % best)
