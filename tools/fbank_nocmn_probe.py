"""Kaldi fbank WITHOUT CMN (fbank512_wave_kernel, a run of units per wave), 1024 x 10 s, a few hundred launches: what the producing kernel
costs on its own.  MELSPEC_LIB selects the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mel_spec_amd as M
from oracle import oracle as O
fb = M.Fbank(M.FbankConfig(apply_cmn=False)); n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4); M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
nf = fb.num_frames(clip_len); out = M.DeviceBuffer(n_clips * nf * 80 * 4)
f = lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    for _ in range(20): f()
    fb.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(100): f()
    fb.synchronize(); best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
cfg = O.fbank_default_config(); cfg.apply_cmn = 0
got = out.download((nf, 80))
print(os.path.basename(os.environ.get("MELSPEC_LIB", "default")), "kaldi no-cmn ms %.4f" % best, "parity %.2e" % np.abs(got - O.fbank_compute(O.synth_pcm(0, clip_len), cfg)).max())
