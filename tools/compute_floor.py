import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
for nm in (80, 128):
    m = M.HipMelSpectrogram(400, 160, 16000.0, nm)
    out = M.DeviceBuffer(n_clips * m.num_frames(clip_len) * nm * 4)
    for stride, label in ((clip_len, "normal"), (0, "every clip reads clip 0 (input from L2)")):
        for _ in range(300): m.compute_uniform_device(pcm.ptr, stride, clip_len, n_clips, out.ptr)
        m.synchronize()
        ms = min(m.time_uniform_device(pcm.ptr, stride, clip_len, n_clips, out.ptr, warmup=50, iters=400) for _ in range(3))
        print(nm, label, round(ms, 4))
    m.close()
