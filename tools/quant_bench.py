#!/usr/bin/env python3
"""Times the TGA quantiser (csrc/tga_quant.hpp) on the mel images of the bench workload:
1024 clips x 10 s -> [1024][80][1000] f32 (interleave_frames even width) -> 1024 TGA blobs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M

n_clips, clip_len, n_mels = int(os.environ.get("Q_CLIPS", "1024")), 160000, 80
m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
W = m.interleaved_width(clip_len, 2)
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips)
img = M.DeviceBuffer(n_clips * n_mels * W * 4)
m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, img.ptr, False, 2)
m.synchronize()
q = M.TgaCodec()
n, stride, last = q.layout(n_mels, W)
blobs = M.DeviceBuffer(n_clips * stride)
back = M.DeviceBuffer(n_clips * n_mels * W * 4)
px = n_clips * n_mels * W
for name, fn, bytes_alg in (("encode", lambda: q.encode_device(img.ptr, n_mels * W, n_mels, W, n_clips, blobs.ptr, stride), 5 * px),
                            ("decode", lambda: q.decode_device(blobs.ptr, stride, n_mels, W, n_clips, back.ptr, n_mels * W), 5 * px)):
    for _ in range(20):
        fn()
    q.synchronize()
    t0 = time.perf_counter()
    iters = 200
    for _ in range(iters):
        fn()
    q.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f"{name}: W={W} {ms:.4f} ms per {n_clips} images  {px / ms / 1e6:.1f} G px/s  {bytes_alg / ms / 1e6:.0f} GB/s algorithmic (5 B/px)"
          f"  = {n_clips * (W - 2) / ms / 1e6:.2f} G frames/s", flush=True)
