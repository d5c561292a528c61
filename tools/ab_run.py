#!/usr/bin/env python3
"""Same-box A/B of library variants (boxes differ by several percent, so variants are interleaved and repeated):
tools/ab_run.py [--case cfg2|cfg4|f64|fbank|fbank_split|nemo|w512|mm] name1 name2 ...   (libraries mel_spec_amd/ab/lib_<name>.so).
Each measurement runs in its own process (MELSPEC_LIB), event-timed, after a spin-up; prints per variant min / median over reps."""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import mel_spec_amd as M
from oracle import oracle as O
case = sys.argv[2]
n_clips, clip_len = 1024, 160000
pcm = M.DeviceBuffer(n_clips * clip_len * 4)
M.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips); M.device_synchronize()
def spin(fn, sync):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(20): fn()
        sync()
def wall(fn, sync, iters):
    spin(fn, sync)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(iters): fn()
        sync()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best
if case in ("cfg2", "cfg4", "f64", "f64_128", "mm", "mm64", "mm64_128", "mm_128", "f32") or case.startswith("nm"):
    nm = 128 if case in ("cfg4", "f64_128", "mm64_128", "mm_128") else (int(case[2:]) if case.startswith("nm") else 80)      # nm<k>: k mels (run-time-lens kernels for non-default banks)
    m = M.HipMelSpectrogram(400, 160, 16000.0, nm)
    if case in ("f64", "f64_128", "mm64", "mm64_128"): m.set_precision("f64")
    if case == "f32": m.set_precision("f32")
    if os.environ.get("AB_NOVOTE"): m.set_auto_adaptive(False)      # variant "name+nv": AUTO without the vote (one launch per call)
    nf = m.num_frames(clip_len)
    out = M.DeviceBuffer(n_clips * (nf + 8) * nm * 4)
    if case in ("mm", "mm64", "mm64_128", "mm_128"):
        fn = lambda: m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, False, 2)
        ms = wall(fn, m.synchronize, 200)
        got = None
    else:
        spin(lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), m.synchronize)
        ms = min(m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=50, iters=400) for _ in range(3))
        got = out.download((nf, nm), offset_bytes=5 * nf * nm * 4)
        want = O.compute_mel_spectrogram_cpu(O.synth_pcm(5, clip_len), 400, 160, nm, 16000.0)
        assert np.abs(got - want).max() <= 1e-4, np.abs(got - want).max()
elif case in ("speech", "speech128"):
    # the reference's own fixture tiled to the config-2 batch, default mode: the launch's vote hands it to the gated f64 kernel
    jfk = O.load_wav_f32(os.path.join(sys.argv[1], "tests", "golden", "jfk_f32le.wav"))
    x = np.stack([np.resize(np.roll(jfk, -1237 * c), clip_len) for c in range(64)])
    pcm.upload(np.tile(x, (n_clips // 64, 1)).reshape(-1))
    nm = 128 if case == "speech128" else 80
    m = M.HipMelSpectrogram(400, 160, 16000.0, nm)
    nf = m.num_frames(clip_len)
    out = M.DeviceBuffer(n_clips * (nf + 8) * nm * 4)
    spin(lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), m.synchronize)
    ms = min(m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=50, iters=300) for _ in range(3))
    got = out.download((nf, nm), offset_bytes=5 * nf * nm * 4)
    assert np.abs(got - O.compute_mel_spectrogram_cpu(x[5], 400, 160, nm, 16000.0)).max() <= 1e-4
elif case in ("w512", "w512_f32", "w512_128", "w512_128_f32"):
    nm = 128 if "128" in case else 80
    m = M.HipMelSpectrogram(512, 160, 16000.0, nm)
    if case.endswith("f32"): m.set_precision("f32")
    out = M.DeviceBuffer(n_clips * m.num_frames(clip_len) * nm * 4)
    spin(lambda: m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), m.synchronize)
    ms = min(m.time_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=20, iters=200) for _ in range(3))
elif case == "fbank":
    fb = M.Fbank()
    out = M.DeviceBuffer(n_clips * fb.num_frames(clip_len) * 80 * 4)
    ms = wall(lambda: fb.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fb.synchronize, 100)
    got = out.download((fb.num_frames(clip_len), 80), offset_bytes=0)
    assert np.abs(got - O.fbank_compute(O.synth_pcm(0, clip_len))).max() <= 1e-4
elif case == "fbank_split":          # rows before CMN + the clips' means (melspec_fbank_compute_uniform_device_split)
    fb = M.Fbank()
    out = M.DeviceBuffer(n_clips * fb.num_frames(clip_len) * 80 * 4)
    means = M.DeviceBuffer(n_clips * 80 * 4)
    ms = wall(lambda: fb.compute_uniform_device_split(pcm.ptr, clip_len, clip_len, n_clips, out.ptr, means.ptr), fb.synchronize, 100)
elif case in ("nemo", "nemo_norm", "nemo_f32", "nemo_norm_f32", "nemo80", "nemo80_f32", "nemo_nopre", "nemo_nopre_f32"):
    fe = M.BatchLogMelSpectrogram(M.BatchLogMelConfig(n_mels=80 if "80" in case else 128, preemphasis=0.0 if "nopre" in case else 0.97, normalize_per_feature=("norm" in case)))
    if case.endswith("f32"): fe.set_precision("f32")
    out = M.DeviceBuffer(n_clips * (fe.num_frames(clip_len) + 16) * 128 * 4)
    ms = wall(lambda: fe.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr), fe.synchronize, 100)
print("MS", ms)
'''
args = sys.argv[1:]
case = "cfg2"
if args and args[0] == "--case":
    case, args = args[1], args[2:]
reps = int(os.environ.get("AB_REPS", "3"))
res = {n: [] for n in args}
for r in range(reps):
    for n in args:
        env = dict(os.environ, MELSPEC_LIB=os.path.join(ROOT, "mel_spec_amd", "ab", f"lib_{n.split('+')[0]}.so"))
        if n.endswith("+nv"):
            env["AB_NOVOTE"] = "1"
        for kv in n.split("+")[1:]:                 # name+MELSPEC_X=1+...: lab switches of a -DMELSPEC_LAB build
            if "=" in kv:
                env[kv.split("=")[0]] = kv.split("=", 1)[1]
        p = subprocess.run([sys.executable, "-c", WORKER, ROOT, case], env=env, capture_output=True, text=True, timeout=600)
        ms = [float(l.split()[1]) for l in p.stdout.splitlines() if l.startswith("MS")]
        if not ms:
            print(n, "FAILED", p.stderr[-600:])
            continue
        res[n].append(ms[0])
base = statistics.median(res[args[0]]) if res[args[0]] else 0
for n in args:
    if res[n]:
        med = statistics.median(res[n])
        print(f"{case:6s} {n:28s} min {min(res[n]):.4f}  median {med:.4f}  ({(med / base - 1) * 100:+.2f} % vs {args[0]})  {['%.4f' % v for v in res[n]]}", flush=True)
