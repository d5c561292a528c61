"""whisper400_six64_kernel<15, LensSix128> (round 5): the f64 six-frame kernel on Whisper large-v3's 128-mel bank -- MELSPEC_PRECISION_F64 on
plain batches (uniform and ragged) and AUTO's gated launch on plain batches.  Since round 6 the f32 launch in front of it is
whisper400_six_wide_runs_kernel<15, LensSix128> -- six frames per wave on twelve waves, 168 VGPRs -- so both launches walk ONE six-frame
plan, ragged batches included (round 5's f32 kernel dealt five-frame units and the uniform batch was planned twice).  The f32 layouts run
whisper400_six_wide_kernel (rounds, the barrier over the three waves of a SIMD); F64 layouts stay on the five-frame precise kernel: the
fifteen-slot LAYOUT instantiation of the f64 kernel was built and measured 1.7 % slower (mel-major F64 at 128 mels 0.5604 -> 0.5697 ms)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SR = 16000.0


def test_f64_mode_at_128_mels_runs_the_six_frame_kernel(gpu, oracle, jfk):
    m = gpu.HipMelSpectrogram(400, 160, SR, 128)
    m.set_precision("f64")
    assert "whisper400_six64_kernel<15" in m.plain_kernel_name()
    # uniform: clips whose frame counts are 0, 1 .. 5 past a multiple of six, and one shorter than a unit
    for n_frames in (1, 5, 6, 7, 11, 12, 313, 998):
        clip_len = 400 + (n_frames - 1) * 160 + 7
        n_clips = 37
        clips = np.stack([np.roll(jfk, -977 * c)[:clip_len] for c in range(n_clips)]).astype(np.float32)
        pcm = gpu.DeviceBuffer(clips.nbytes)
        pcm.upload(clips.reshape(-1))
        out = gpu.DeviceBuffer(n_clips * n_frames * 128 * 4)
        m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
        m.synchronize()
        got = out.download((n_clips, n_frames, 128))
        for c in (0, 18, 36):
            assert np.abs(got[c] - oracle.compute_mel_spectrogram_cpu(clips[c], 400, 160, 128, SR)).max() <= 2e-6, (n_frames, c)
        pcm.free(); out.free()
    # ragged
    lens = [0, 399, 400, 1360, 16000, 5003, 48000, 1361]
    xs = [jfk[1000 * i:1000 * i + n].copy() for i, n in enumerate(lens)]
    flat = np.concatenate(xs)
    offs = np.cumsum([0] + lens[:-1]).astype(np.uint64)
    out, total = m.compute_batch_host(flat, offs, np.asarray(lens, np.uint64))
    cur = 0
    for x in xs:
        want = oracle.compute_mel_spectrogram_cpu(x, 400, 160, 128, SR)
        g = out[cur:cur + want.size].reshape(want.shape)
        cur += want.size
        if want.size:
            assert np.abs(g - want).max() <= 2e-6, len(x)
    # a layout at 128 mels stays on the five-frame f64 kernel and agrees
    img = m.compute_batch_interleaved(np.stack([jfk[:16000], jfk[16000:32000]]), False, 0)
    for k in range(2):
        want = oracle.interleave_frames(oracle.compute_mel_spectrogram_cpu(jfk[16000 * k:16000 * (k + 1)], 400, 160, 128, SR), False, 0)
        assert np.abs(img[k] - want).max() <= 2e-6


def test_auto_at_128_mels_hands_speech_to_the_six_frame_kernel_and_noise_to_nobody(gpu, oracle, jfk):
    m = gpu.HipMelSpectrogram(400, 160, SR, 128)
    m.set_precision("auto")                 # (the suite is also run with MELSPEC_PRECISE=1)
    assert "whisper400_six_wide_runs_kernel<15" in m.plain_kernel_name()
    n_clips, clip_len = 300, 48000
    speech = np.stack([np.roll(jfk, -1237 * c)[:clip_len] for c in range(n_clips)]).astype(np.float32)
    noise = np.stack([oracle.synth_pcm(8 * c, clip_len) for c in range(n_clips)])
    nf = m.num_frames(clip_len)
    pcm = gpu.DeviceBuffer(speech.nbytes)
    out = gpu.DeviceBuffer(n_clips * nf * 128 * 4)
    for name, clips, heavy_want in (("speech", speech, True), ("noise", noise, False), ("speech again", speech, True)):
        pcm.upload(clips.reshape(-1))
        m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
        m.synchronize()
        got = out.download((n_clips, nf, 128))
        heavy = bool(m.auto_state()[0])
        assert heavy == heavy_want, (name, m.auto_state())
        for c in (0, 149, 299):
            d = np.abs(got[c] - oracle.compute_mel_spectrogram_cpu(clips[c], 400, 160, 128, SR)).max()
            assert d <= (2e-6 if heavy else 1e-4), (name, c, d)      # heavy: every frame came from the f64 kernel


def test_auto_ragged_at_128_mels_walks_one_six_frame_plan(gpu, oracle, jfk):
    """Ragged plain batches at 128 mels in AUTO (round 6): the f32 launch and the gated f64 launch read the same six-frame plan.  A light
    batch whose guard noted a few clips (lines over a floor 70 .. 85 dB down) and a heavy one (speech), clip ends inside a wave's run,
    clips shorter than a unit and empty clips in both."""
    m = gpu.HipMelSpectrogram(400, 160, SR, 128)
    m.set_precision("auto")
    rng = np.random.default_rng(21)
    lens = [int(v) for v in rng.integers(500, 60000, 300)] + [0, 399, 400, 400 + 5 * 160, 400 + 6 * 160 + 3]

    def line(n, db, f, seed):
        t = np.arange(n) / SR
        return (0.9 * np.sin(2 * np.pi * f * t) + 10 ** (db / 20) * np.random.default_rng(seed).standard_normal(n)).astype(np.float32)

    light = [oracle.synth_pcm(300 + i, n) if i % 29 else line(n, -70.0 - (i % 16), 900.0 + 13 * i, i) for i, n in enumerate(lens)]
    heavy = [np.resize(np.roll(jfk, -911 * i), n).astype(np.float32) for i, n in enumerate(lens)]
    for name, clips, heavy_want in (("light", light, False), ("heavy", heavy, True), ("light again", light, False)):
        got = m.compute_ragged(clips)
        assert bool(m.auto_state()[0]) == heavy_want, (name, m.auto_state())
        if not heavy_want:
            assert m.guard_last_count() > 0, name
        for i in list(range(0, len(clips), 7)) + list(range(len(clips) - 5, len(clips))):
            w = oracle.compute_mel_spectrogram_cpu(clips[i], 400, 160, 128, SR)
            assert got[i].shape == w.shape and (w.size == 0 or np.abs(got[i] - w).max() <= 1e-4), (name, i, lens[i])
    m.close()
