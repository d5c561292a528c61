#!/usr/bin/env python3
"""Times ablated builds (libmelspec_ablN.so) of the wave kernel: no parity check (results are wrong by design)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips*clip_len*4); M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips)
m = M.HipMelSpectrogram(400,160,16000.0,80); fpc = m.num_frames(clip_len)
out = M.DeviceBuffer(n_clips*fpc*80*4)
ts=[m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=30, iters=100) for _ in range(3)]
print(os.environ.get("MELSPEC_LIB","default"), "variant", os.environ.get("MELSPEC_VARIANT","-"), "ms", min(ts))
''' % ROOT
libs = [os.path.join(ROOT, "mel_spec_amd", a) for a in sys.argv[1:]] or \
       [os.path.join(ROOT, "mel_spec_amd", f"libmelspec_abl{a}.so") for a in (1, 2, 3, 4, 12)]
for lib in [None] + libs + [None]:
    env = dict(os.environ)
    if lib: env["MELSPEC_LIB"] = lib
    subprocess.run([sys.executable, "-c", code], env=env)
