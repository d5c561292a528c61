import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mel_spec_amd as M
n_clips = 1024
CASES = ((128, 64, 40, 8000.0), (256, 128, 80, 16000.0), (1024, 256, 80, 16000.0), (2048, 512, 128, 44100.0))
for n_fft, hop, n_mels, sr in CASES:
    if len(sys.argv) > 1 and str(n_fft) not in sys.argv[1:]: continue
    clip_len = int(10 * sr)
    pcm = M.DeviceBuffer(n_clips * clip_len * 4)
    M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
    m = M.HipMelSpectrogram(n_fft, hop, sr, n_mels)
    nf = m.num_frames(clip_len)
    out = M.DeviceBuffer(n_clips * nf * n_mels * 4)
    run = lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
    for _ in range(5): run()
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    m.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{os.path.basename(os.environ.get('MELSPEC_LIB','default'))} n_fft {n_fft}: {dt*1e3:.3f} ms  {n_clips*nf/dt/1e9:.3f} G frames/s  {n_clips*nf*(hop+n_mels)*4/dt/8e12*100:.2f} % of 8 TB/s", flush=True)
