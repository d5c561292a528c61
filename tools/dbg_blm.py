import sys, numpy as np
sys.path.insert(0, '/root/repo')
import mel_spec_amd as M
from oracle import oracle as O
jfk = O.load_wav_f32('/root/repo/tests/golden/jfk_f32le.wav')
fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=128, preemphasis=0.97, log_zero_guard=2.0**-24))
cfg = O.blm_default_config(n_mels=128, preemphasis=0.97, log_zero_guard=2.0**-24)
got = fe.compute(jfk); want, _ = O.blm_compute(jfk, cfg, True)
d = np.abs(got - want)
bad = np.argwhere(d > 2e-5)
print('max', d.max(), 'n bad', len(bad))
print('bad frames', sorted(set(bad[:, 1].tolist()))[:40])
print('bad mels', sorted(set(bad[:, 0].tolist()))[:60])
for m, f in bad[:8]:
    print(m, f, got[m, f], want[m, f])
