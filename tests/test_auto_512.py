"""MELSPEC_PRECISION_AUTO at n_fft = 512 (round 6, VERDICT r05 'next' 5): the geometry of the reference's only value-level golden
(`/root/reference/src/rb.rs:134-179`, 512 / 160 / 80) on the pair of 512-point kernels -- the f32 kernel with the precision guard and
the vote, the f64 kernel gated on the verdict (w512_auto_kernel) -- with the n_fft = 400 family's contract: within 1e-4 of the f64
evaluation on every input, and the bits of a batch a function of the batch alone."""
import os

import numpy as np
import pytest

from conftest import GOLDEN  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = 1e-4
SR = 16000.0


def _auto_pass():
    return os.environ.get("MELSPEC_PRECISE", "")[:1] == ""


def _tone_over_floor(n, level_db, f, seed):
    t = np.arange(n) / SR
    rng = np.random.default_rng(seed)
    return (0.9 * np.sin(2 * np.pi * f * t) + 10 ** (level_db / 20) * rng.standard_normal(n)).astype(np.float32)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_auto_512_is_a_function_of_the_batch(gpu, oracle, jfk, n_mels):
    """Speech goes to the gated f64 launch on the FIRST batch of a fresh context; hash noise stays on the f32 launch; the same batch gives
    the same bits whatever the context computed before it; melspec_set_auto_adaptive(0) computes every batch in f64 (this family has no
    in-kernel recompute: without the vote there is nothing to gate)."""
    if not _auto_pass():
        pytest.skip("the suite is being run with a fixed precision mode")
    n_clips, clip_len = 256, 160000
    speech = np.stack([np.resize(np.roll(jfk, -1237 * c), clip_len) for c in range(32)])
    noise = np.stack([oracle.synth_pcm(c, clip_len) for c in range(32)])
    m = gpu.HipMelSpectrogram(512, 160, SR, n_mels)
    assert m.uses_fast_path and m.precision == "auto" and "w512_auto_kernel" in m.plain_kernel_name()
    nf = m.num_frames(clip_len)
    pcm_s, pcm_n = gpu.DeviceBuffer(n_clips * clip_len * 4), gpu.DeviceBuffer(n_clips * clip_len * 4)
    out = gpu.DeviceBuffer(n_clips * nf * n_mels * 4)
    for r in range(n_clips // 32):
        pcm_s.upload(speech, offset_bytes=r * speech.nbytes)
        pcm_n.upload(noise, offset_bytes=r * noise.nbytes)

    def run(pcm, ctx=m):
        out.upload(np.zeros(n_clips * nf * n_mels, np.float32))
        ctx.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
        ctx.synchronize()
        return out.download((n_clips, nf, n_mels))

    want = np.stack([oracle.compute_mel_spectrogram_cpu(speech[c], 512, 160, n_mels, SR) for c in (0, 31)])
    wantn = np.stack([oracle.compute_mel_spectrogram_cpu(noise[c], 512, 160, n_mels, SR) for c in (0, 31)])
    assert m.auto_state() == (False, 0.0)
    s_fresh = run(pcm_s)
    heavy, frac = m.auto_state()
    assert heavy and 0.2 < frac < 0.95, (heavy, frac)               # the f64 launch computed it and reports the frames that would have tripped
    assert np.abs(s_fresh[[0, 255]] - want).max() <= 2e-6
    n_after_speech = run(pcm_n)
    heavy, frac = m.auto_state()
    assert not heavy and frac < 0.02, (heavy, frac)
    assert np.abs(n_after_speech[[0, 255]] - wantn).max() <= TOL
    s_after_noise = run(pcm_s)
    n_after_noise = run(pcm_n); n_again = run(pcm_n)
    assert np.array_equal(s_fresh, s_after_noise)
    assert np.array_equal(n_after_speech, n_after_noise) and np.array_equal(n_after_noise, n_again)
    m2 = gpu.HipMelSpectrogram(512, 160, SR, n_mels)
    assert np.array_equal(run(pcm_n, m2), n_after_speech) and np.array_equal(run(pcm_s, m2), s_fresh)
    m2.close()
    # what it is for: noise-like input at the f32 rate, speech at about the f64 rate
    t_noise = min(m.time_uniform_device(pcm_n.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=20, iters=50) for _ in range(2))
    t_speech = min(m.time_uniform_device(pcm_s.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=20, iters=50) for _ in range(2))
    m.set_precision("f64")
    assert m.precision == "f64"
    t_f64 = min(m.time_uniform_device(pcm_s.ptr, clip_len, clip_len, n_clips, out.ptr, warmup=20, iters=50) for _ in range(2))
    f64_bits = run(pcm_s)
    m.set_precision("auto")
    assert np.abs(f64_bits - s_fresh).max() <= 2e-6                 # heavy = the f64 arithmetic (another instantiation: the last bit of a sum may differ)
    assert t_noise <= 0.92 * t_f64 and t_speech <= t_f64 + 0.08, (t_noise, t_speech, t_f64)
    # no vote -> f64 whatever the input
    m.set_auto_adaptive(False)
    assert m.precision == "f64"
    assert np.abs(run(pcm_n)[[0, 255]] - wantn).max() <= 2e-6
    m.set_auto_adaptive(True)
    assert m.precision == "auto"
    pcm_s.free(); pcm_n.free(); out.free(); m.close()


def test_auto_512_recomputes_the_noted_units_of_a_light_batch(gpu, oracle):
    """A batch the vote calls light (hash noise in the sampled units) with clips the f32 FFT cannot vouch for further back -- a line over
    a floor 70 .. 90 dB down, silence with a click: the f32 launch notes their units, the gated launch recomputes exactly those in f64.
    Within 1e-4 everywhere; the bare f32 kernel (MELSPEC_PRECISION_F32) misses the same clips."""
    if not _auto_pass():
        pytest.skip("the suite is being run with a fixed precision mode")
    n_clips, clip_len, n_mels = 768, 32000, 80
    clips = np.stack([oracle.synth_pcm(c, clip_len) for c in range(n_clips)])
    hard = {700: _tone_over_floor(clip_len, -70.0, 3333.3, 1), 701: _tone_over_floor(clip_len, -90.0, 1000.0, 2), 740: _tone_over_floor(clip_len, -80.0, 6100.0, 3)}
    click = np.zeros(clip_len, np.float32); click[12345] = 0.8
    hard[767] = click
    for c, x in hard.items():
        clips[c] = x
    m = gpu.HipMelSpectrogram(512, 160, SR, n_mels)
    pcm = gpu.DeviceBuffer(clips.nbytes); pcm.upload(clips.reshape(-1))
    nf = m.num_frames(clip_len)
    out = gpu.DeviceBuffer(n_clips * nf * n_mels * 4)

    def run():          # the whole batch in one launch pair (the host pipeline would cut it into 16 MiB chunks, each a batch of its own)
        m.compute_uniform_device(pcm.ptr, clip_len, clip_len, n_clips, out.ptr)
        m.synchronize()
        return out.download((n_clips, nf, n_mels))

    got = run()
    heavy, frac = m.auto_state()
    assert not heavy and 0.0 < frac < 0.05, (heavy, frac)          # light: the noted frames are what the hard clips hold
    assert m.guard_last_count() > 0
    picks = sorted(set(list(hard) + [0, 1, 383, 699, 702, 766]))
    worst = {c: float(np.abs(got[c] - oracle.compute_mel_spectrogram_cpu(clips[c], 512, 160, n_mels, SR)).max()) for c in picks}
    assert max(worst.values()) <= TOL, worst
    assert np.array_equal(run(), got)                             # and again: the same bits
    m.set_precision("f32")
    bare = run()
    miss = max(float(np.abs(bare[c] - oracle.compute_mel_spectrogram_cpu(clips[c], 512, 160, n_mels, SR)).max()) for c in (700, 701, 740))
    assert miss > 1.5 * max(worst[c] for c in (700, 701, 740)), (miss, worst)      # the recompute is what holds these clips, not luck
    pcm.free(); out.free(); m.close()


def test_auto_512_ragged_batches_and_the_streaming_golden(gpu, oracle, jfk):
    """Ragged plain batches take the same pair of launches (clip ends inside a wave's run, clips shorter than a unit, empty clips); the
    reference's golden -- speech -- goes through the vote to the f64 launch and stays within 1e-6 of rust_jfk_golden.npy."""
    m = gpu.HipMelSpectrogram(512, 160, SR, 80)
    rng = np.random.default_rng(5)
    lens = [int(v) for v in rng.integers(600, 90000, 395)] + [0, 511, 512, 672, 512 + 4 * 160 + 7]          # ~110 000 frames: past the size under which a call is all f64
    clips = [oracle.synth_pcm(100 + i, n) if i % 13 else _tone_over_floor(n, -75.0, 2500.0, i)[:n] for i, n in enumerate(lens)]
    got = m.compute_ragged(clips)
    if _auto_pass():
        assert not m.auto_state()[0] and m.guard_last_count() > 0          # light, and the lines over their floors were noted
    for x, g in zip(clips, got):
        w = oracle.compute_mel_spectrogram_cpu(x, 512, 160, 80, SR)
        assert g.shape == w.shape and (w.size == 0 or np.abs(g - w).max() <= TOL)
    golden = np.load(os.path.join(GOLDEN, "rust_jfk_golden.npy"))
    g = m.compute_mel_spectrogram(jfk[128:])
    assert g.shape[0] >= golden.shape[1] and np.abs(g[: golden.shape[1]].T - golden).max() <= 1e-6
    m.close()


def test_auto_512_device_planned_ragged_batch(gpu, oracle):
    """melspec_compute_ragged_device_desc at n_fft = 512 in the default mode: the clip table lives in device memory, the plan is built by a
    kernel and the host only knows an upper bound of the units -- the pair of launches reads the unit count from the plan.  Same bits as
    the host-table call; sampled clips against the oracle."""
    m = gpu.HipMelSpectrogram(512, 160, SR, 80)
    rng = np.random.default_rng(11)
    lens = np.array([int(v) for v in rng.integers(600, 90000, 380)] + [0, 511, 700], np.uint64)
    clips = [oracle.synth_pcm(500 + i, int(n)) if i % 17 else _tone_over_floor(int(n), -80.0, 1800.0, i)[: int(n)] for i, n in enumerate(lens)]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    flat = np.concatenate([c for c in clips if len(c)])
    frames = [m.num_frames(int(n)) for n in lens]
    total = int(sum(frames))
    oo = (np.concatenate([[0], np.cumsum(frames)[:-1]]) * 80).astype(np.uint64)
    din = gpu.DeviceBuffer(flat.nbytes); din.upload(flat)
    d_off, d_len, d_oo = gpu.DeviceBuffer(offs.nbytes), gpu.DeviceBuffer(lens.nbytes), gpu.DeviceBuffer(oo.nbytes)
    d_off.upload(offs); d_len.upload(lens); d_oo.upload(oo)
    a, b = gpu.DeviceBuffer(total * 80 * 4), gpu.DeviceBuffer(total * 80 * 4)
    m.compute_ragged_device(din.ptr, offs, lens, a.ptr, oo); m.synchronize()
    m.compute_ragged_device_desc(din.ptr, d_off.ptr, d_len.ptr, len(clips), b.ptr, d_oo.ptr, total + 1234); m.synchronize()
    ga, gb = a.download((total, 80)), b.download((total, 80))
    assert np.array_equal(ga, gb)
    for c in (0, 17, 34, 200, 379, 381, 382):
        w = oracle.compute_mel_spectrogram_cpu(clips[c], 512, 160, 80, SR)
        lo = int(oo[c]) // 80
        assert w.shape[0] == frames[c] and (w.size == 0 or np.abs(gb[lo:lo + frames[c]] - w).max() <= TOL), c
    for x in (din, d_off, d_len, d_oo, a, b):
        x.free()
    m.close()
