import numpy as np, sys
sys.path.insert(0, '/root/repo')
import mel_spec_amd as M
from oracle import oracle as O
rng = np.random.default_rng(1)
n = 48000
t = np.arange(n) / 16000.0
sigs = {}
for f in (40.0, 200.0, 1000.0, 3990.0, 7960.0):
    for floor_db in (-60, -90, -120, -150):
        sigs[f"tone{f:.0f}_floor{floor_db}"] = (0.99 * np.sin(2 * np.pi * f * t) + 10 ** (floor_db / 20) * rng.standard_normal(n)).astype(np.float32)
sigs["two_tones_far"] = (0.9 * np.sin(2 * np.pi * 100 * t) + 1e-5 * np.sin(2 * np.pi * 7900 * t)).astype(np.float32)
sigs["lowpass_noise"] = np.convolve(rng.standard_normal(n), np.ones(64) / 64, "same").astype(np.float32)
sigs["impulses"] = np.zeros(n, np.float32); sigs["impulses"][::997] = 1.0
sigs["dc_plus_hf"] = (0.9 + 1e-4 * np.sin(2 * np.pi * 7000 * t)).astype(np.float32)
worst = {}
for fft, nm in ((400, 80), (400, 128), (400, 16), (512, 80), (512, 128)):
    m = M.HipMelSpectrogram(fft, 160, 16000.0, nm)
    m.set_precision("f64")
    w = 0.0
    for k, x in sigs.items():
        d = float(np.abs(m.compute_mel_spectrogram(x) - O.compute_mel_spectrogram_cpu(x, fft, 160, nm, 16000.0)).max())
        if d > w: w, wk = d, k
    print(fft, nm, "worst |gpu - oracle| in F64 mode:", w, wk)
    m.close()
