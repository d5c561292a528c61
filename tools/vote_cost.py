#!/usr/bin/env python3
"""What AUTO's vote costs (round 4): config 2 (1024 x 10 s) and 128 mels, hash noise and speech, event-timed ms per call for
auto (vote + gated f64 launch), auto without the vote (melspec_set_auto_adaptive(0): one launch) and f64.  Run it under
`rocprofv3 --kernel-trace --stats` to split a call into its two kernels and the gap between them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O

n_clips, clip_len = 1024, 160000
jfk = O.load_wav_f32(os.path.join(ROOT, "tests", "golden", "jfk_f32le.wav"))
sets = {"noise": np.stack([O.synth_pcm(c % 8, clip_len) for c in range(64)]),
        "speech": np.stack([np.resize(np.roll(jfk, -1237 * c), clip_len) for c in range(64)])}
iters = int(os.environ.get("VOTE_ITERS", "400"))
for n_mels in (80, 128):
    m = M.HipMelSpectrogram(400, 160, 16000.0, n_mels)
    nf = m.num_frames(clip_len)
    pcm, out = M.DeviceBuffer(n_clips * clip_len * 4), M.DeviceBuffer(n_clips * nf * n_mels * 4)
    for name, x in sets.items():
        for r in range(n_clips // 64):
            pcm.upload(x, offset_bytes=r * x.nbytes)
        t0 = time.perf_counter()                        # spin the clocks up
        while time.perf_counter() - t0 < 0.5:
            for _ in range(20):
                m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
            m.synchronize()
        res = {}
        for rep in range(3):
            for mode in ("auto", "auto-novote", "f64"):
                m.set_precision(mode.split("-")[0])
                m.set_auto_adaptive(mode != "auto-novote")
                if name == "speech" and mode == "auto-novote":
                    continue
                ms = m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=40, iters=iters)
                res[mode] = min(res.get(mode, 1e9), ms)
        m.set_precision("auto"); m.set_auto_adaptive(True)
        print(f"{n_mels:3d} mels  {name:7s} " + "  ".join(f"{k} {v:.4f} ms" for k, v in res.items()), flush=True)
    pcm.free(); out.free(); m.close()
