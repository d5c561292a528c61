"""The fused 512-point kernels in a loop, for tools/pmc_kernel.sh: python tools/pmc512_workload.py [f32] [nemo|nemo80|w512|w512_128 ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mel_spec_amd as M
args = sys.argv[1:]
f32 = "f32" in args
cases = [a for a in args if a != "f32"] or ["nemo", "w512"]
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
out = M.DeviceBuffer(n_clips * 1017 * 128 * 4)
for c in cases:
    if c.startswith("nemo"):
        fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=80 if c == "nemo80" else 128, preemphasis=0.0 if c.endswith("nopre") else 0.97))
        if f32: fe.set_precision("f32")
        for _ in range(40): fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
        fe.synchronize()
    else:
        m = M.HipMelSpectrogram(512, 160, 16000.0, 128 if c == "w512_128" else 80)
        if f32: m.set_precision("f32")
        for _ in range(40): m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
        m.synchronize()
