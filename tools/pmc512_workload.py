import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mel_spec_amd as M
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97))
out = M.DeviceBuffer(n_clips * (fe.num_frames(clip_len) + 16) * 128 * 4)
for _ in range(40): fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
fe.synchronize()
m = M.HipMelSpectrogram(512, 160, 16000.0, 80)
for _ in range(40): m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
m.synchronize()
