"""Non-finite and denormal samples through every kernel family.

What the reference does with them follows from IEEE arithmetic and Rust's `f64::max`, which drops a NaN operand:
  * Whisper path (src/mel.rs:148-168, 645-654): a NaN sample makes every bin of its frames NaN (each output of a DFT depends on each input);
    `energy.max(1e-10)` maps a NaN band to the floor, so such a frame is the all-floor row -1.5 -- silence.  An infinite sample leaves
    every bin +-Inf or NaN (WHICH depends on the order of operations inside the FFT, rustfft's or anybody's), so every band is +Inf or the
    floor; one +Inf band makes the frame maximum +Inf and `max(x, Inf - 8)` the whole row +Inf.  Implementation-independent statement:
    every row of a frame that touches the sample is all -1.5 or all +Inf, every other frame is untouched.
  * Kaldi fbank (src/fbank.rs:205-233): `(*mel_energy).max(floor)` -> ln(f32::EPSILON) on NaN frames; the CMN mean then folds finite values.
  * NeMo (src/mel.rs:356-368, 721-749): ln(NaN + guard) stays NaN; with normalize_per_feature the NaN reaches the row's mean and every
    valid column of every row is NaN.
The oracle follows the same rule (C's fmax drops a NaN operand like f64::max).  Expected values below come from it; where the pattern of
+Inf / floor depends on the FFT's internals the tests assert the implementation-independent statement instead.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
SR = 16000.0
LN_EPS = float(np.log(np.float64(np.finfo(np.float32).eps)))


def _loud(oracle, n, clip=0):
    return oracle.synth_pcm(clip, n).copy()            # amplitude 1 (clip & 7 == 0): hash noise


def _touch(nf, n_fft, hop, p, before=0):
    """frames whose window [f*hop - before, f*hop + n_fft) holds sample p"""
    f = np.arange(nf)
    return (f * hop - before <= p) & (p < f * hop + n_fft)


def _check_whisper(got, want, touched, what, floor_tol=0.0):
    assert got.shape == want.shape, what
    clean = ~touched
    if clean.any():
        assert np.isfinite(got[clean]).all() and np.abs(got[clean] - want[clean]).max() <= TOL, what
    for f in np.nonzero(touched)[0]:
        row = got[f]
        assert (np.abs(row + 1.5) <= floor_tol).all() or (np.isposinf(row)).all(), (what, f, row[:8])


@pytest.mark.parametrize("mode", ["auto", "f64", "f32"])
@pytest.mark.parametrize("n_mels", [80, 128])
def test_whisper400_nan_inf_denormal_every_batch_shape(gpu, oracle, mode, n_mels):
    m = gpu.HipMelSpectrogram(400, 160, SR, n_mels)
    m.set_precision(mode)
    n = 16000
    nf = m.num_frames(n)
    p = 5003
    variants = {}
    for name, v in (("nan", np.nan), ("+inf", np.inf), ("-inf", -np.inf)):
        x = _loud(oracle, n, 8)
        x[p] = v
        variants[name] = x
    rng = np.random.default_rng(3)
    variants["denormal"] = (rng.integers(-8000, 8000, n).astype(np.float32) * np.float32(1e-42)).astype(np.float32)
    variants["all nan"] = np.full(n, np.nan, np.float32)
    touched = _touch(nf, 400, 160, p)
    for name, x in variants.items():
        want = oracle.compute_mel_spectrogram_cpu(x, 400, 160, n_mels, SR)
        got = m.compute_mel_spectrogram(x)
        if name == "nan":
            assert (want[touched] == -1.5).all()                                  # the oracle states the rule
            assert np.abs(got - want).max() <= TOL and (got[touched] == -1.5).all(), (mode, name)
        elif name == "denormal":
            assert np.isfinite(got).all() and np.abs(got - want).max() <= TOL, (mode, name)
        elif name == "all nan":
            assert (got == -1.5).all() and (want == -1.5).all(), (mode, name)
        else:
            _check_whisper(got, want, touched, (mode, name))
    # a batch: clean clips around the poisoned ones -- uniform, ragged, mel-major; the poisoned clips must not leak into their neighbours
    clips = [_loud(oracle, n, 8 + c) for c in range(12)]
    clips[3] = variants["nan"]; clips[7] = variants["+inf"]; clips[9] = variants["all nan"]; clips[10] = variants["denormal"]
    wants = [oracle.compute_mel_spectrogram_cpu(c, 400, 160, n_mels, SR) for c in clips]
    got = m.compute_batch(np.stack(clips))
    for c in range(12):
        _check_whisper(got[c], wants[c], touched if c in (3, 7) else (np.ones(nf, bool) if c == 9 else np.zeros(nf, bool)), (mode, "uniform", c))
    rag = [c[: n - 137 * k] for k, c in enumerate(clips)]
    gr = m.compute_ragged(rag)
    for c in range(12):
        w = oracle.compute_mel_spectrogram_cpu(rag[c], 400, 160, n_mels, SR)
        t = _touch(w.shape[0], 400, 160, p) if c in (3, 7) else (np.ones(w.shape[0], bool) if c == 9 else np.zeros(w.shape[0], bool))
        _check_whisper(gr[c], w, t, (mode, "ragged", c))
    gm = m.compute_batch_interleaved(np.stack(clips), False, 0)                    # [clip][mel][W]
    for c in range(12):
        wm = oracle.interleave_frames(wants[c], False, 0).reshape(n_mels, -1)
        g = gm[c][:, : nf].T
        _check_whisper(g, wants[c], touched if c in (3, 7) else (np.ones(nf, bool) if c == 9 else np.zeros(nf, bool)), (mode, "mel-major", c))
        assert (gm[c][:, nf:] == wm[:, nf:]).all()                                 # the zero padding of an odd frame count
    m.close()


def test_vote_is_not_wedged_by_non_finite_statistics(gpu, oracle, jfk):
    """A batch full of NaN / Inf frames, then noise, then speech, on one context in the default mode: every batch is decided by its own
    vote (NaN frames never trip the guard: their bands all sit on the floor), and the regime after each is the input's."""
    m = gpu.HipMelSpectrogram(400, 160, SR, 80)
    m.set_precision("auto")                   # (the suite is also run with MELSPEC_PRECISE=1, which starts contexts in f64)
    n = 32000
    bad = np.stack([np.full(n, np.nan, np.float32) if c % 2 else np.full(n, np.inf, np.float32) for c in range(64)])
    noise = np.stack([_loud(oracle, n, c) for c in range(64)])
    speech = np.stack([np.resize(np.roll(jfk, -977 * c), n) for c in range(64)])
    for _ in range(2):
        g = m.compute_batch(bad)
        assert ((g == -1.5) | np.isposinf(g)).all()
        g = m.compute_batch(noise)
        assert not m.auto_state()[0]
        assert max(np.abs(g[c] - oracle.compute_mel_spectrogram_cpu(noise[c], 400, 160, 80, SR)).max() for c in (0, 63)) <= TOL
        g = m.compute_batch(speech)
        assert m.auto_state()[0]                                                   # heavy: the gated f64 kernel computed it
        assert max(np.abs(g[c] - oracle.compute_mel_spectrogram_cpu(speech[c], 400, 160, 80, SR)).max() for c in (0, 63)) <= TOL
    # half the clips poisoned, half speech: the vote still sees the speech
    mixed = speech.copy(); mixed[::2] = bad[::2]
    g = m.compute_batch(mixed)
    assert np.abs(g[1] - oracle.compute_mel_spectrogram_cpu(mixed[1], 400, 160, 80, SR)).max() <= TOL
    assert ((g[0] == -1.5) | np.isposinf(g[0])).all()
    m.close()


@pytest.mark.parametrize("n_fft,hop,n_mels", [(512, 160, 80), (256, 64, 40), (1024, 256, 80), (800, 200, 64)])
def test_other_geometries_nan_inf_denormal(gpu, oracle, n_fft, hop, n_mels):
    """Whisper-512 (fused f64 kernel), the power-of-two kernel (256, 1024) and the workgroup-per-frame kernel (800)."""
    m = gpu.HipMelSpectrogram(n_fft, hop, SR, n_mels)
    n = 24000
    nf = m.num_frames(n)
    p = 9001
    touched = _touch(nf, n_fft, hop, p)
    for name, v in (("nan", np.nan), ("+inf", np.inf), ("-inf", -np.inf)):
        x = _loud(oracle, n, 16)
        x[p] = v
        want = oracle.compute_mel_spectrogram_cpu(x, n_fft, hop, n_mels, SR)
        got = m.compute_mel_spectrogram(x)
        if name == "nan":
            assert np.abs(got - want).max() <= TOL and np.abs(got[touched] + 1.5).max() <= 1e-6, (n_fft, name)
        else:
            _check_whisper(got, want, touched, (n_fft, name), 1e-6)
        gb = m.compute_batch(np.stack([_loud(oracle, n, 17), x, _loud(oracle, n, 18)]))
        _check_whisper(gb[1], want, touched, (n_fft, name, "batch"), 1e-6)
        assert np.abs(gb[2] - oracle.compute_mel_spectrogram_cpu(_loud(oracle, n, 18), n_fft, hop, n_mels, SR)).max() <= TOL
    x = (np.random.default_rng(5).integers(-8000, 8000, n).astype(np.float32) * np.float32(1e-42)).astype(np.float32)
    got = m.compute_mel_spectrogram(x)
    assert np.abs(got - oracle.compute_mel_spectrogram_cpu(x, n_fft, hop, n_mels, SR)).max() <= TOL
    m.close()


@pytest.mark.parametrize("cmn", [False, True])
def test_fbank_nan_inf_denormal(gpu, oracle, cmn):
    """A NaN frame is the floor row ln(f32::EPSILON) (src/fbank.rs:207-221) and, with CMN, joins the column means as such
    (src/fbank.rs:224-233).  One clip through the host call, many through the clip kernel (CMN inside) and the ragged path."""
    fb = gpu.Fbank(gpu.FbankConfig(apply_cmn=cmn))
    cfg = oracle.fbank_default_config(); cfg.apply_cmn = int(cmn)
    n = 16000
    nf = fb.num_frames(n)
    p = 6007
    touched = _touch(nf, 400, 160, p, before=1)                 # pre-emphasis reads the sample in front of the frame
    x = _loud(oracle, n, 24); x[p] = np.nan
    want = oracle.fbank_compute(x, cfg)
    raw = oracle.fbank_compute(x, (lambda c: (setattr(c, "apply_cmn", 0), c)[1])(oracle.fbank_default_config()))
    assert np.allclose(raw[touched], LN_EPS, atol=1e-5) and np.isfinite(want).all()      # the rule, stated by the oracle
    got = fb.compute(x)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= TOL
    # many clips (the workgroup-per-clip kernel takes uniform batches that fill the CUs): NaN clips between clean ones
    clips = np.stack([_loud(oracle, n, 24 + c) for c in range(256)])
    clips[5, p] = np.nan; clips[77] = np.nan; clips[200, 17] = np.nan
    gb = fb.compute_batch(clips)
    for c in (0, 5, 6, 77, 200, 255):
        w = oracle.fbank_compute(clips[c], cfg)
        assert np.isfinite(gb[c]).all() and np.abs(gb[c] - w).max() <= TOL, c
    rag = [clips[c][: n - 97 * c] for c in (5, 6, 77, 200)]
    for g, c in zip(fb.compute_ragged(rag), (5, 6, 77, 200)):
        assert np.abs(g - oracle.fbank_compute(clips[c][: n - 97 * c], cfg)).max() <= TOL, c
    # denormal-only clip: every band on the floor
    d = (np.random.default_rng(7).integers(-8000, 8000, n).astype(np.float32) * np.float32(1e-42)).astype(np.float32)
    assert np.abs(fb.compute(d) - oracle.fbank_compute(d, cfg)).max() <= TOL
    if not cmn:
        # an infinite sample: every band of a touched frame is +Inf or the floor (which, depends on the FFT's internals)
        for v in (np.inf, -np.inf):
            y = _loud(oracle, n, 24); y[p] = v
            g = fb.compute(y)
            w = oracle.fbank_compute(y, cfg)
            assert np.abs(g[~touched] - w[~touched]).max() <= TOL
            gt = g[touched]
            assert (np.isposinf(gt) | (np.abs(gt - LN_EPS) <= 1e-5)).all()
    fb.close()


def test_fbank_other_rates_nan(gpu, oracle):
    """the power-of-two kernel's Kaldi flavour (8 kHz: 256-point frames) and the workgroup kernel behind it"""
    fb = gpu.Fbank(gpu.FbankConfig(sample_rate=8000.0, apply_cmn=True))
    cfg = oracle.fbank_default_config(); cfg.sample_rate = 8000.0
    x = _loud(oracle, 12000, 40); x[3001] = np.nan
    want = oracle.fbank_compute(x, cfg)
    assert np.abs(fb.compute(x) - want).max() <= TOL
    fb.use_generic(True)
    assert np.abs(fb.compute(x) - want).max() <= TOL
    fb.close()


@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("preemph", [0.0, 0.97])
def test_nemo_nan_keeps_the_reference_nan_pattern(gpu, oracle, norm, preemph):
    """ln(NaN + guard) is NaN (src/mel.rs:365-368): the columns of the frames that touch the sample; normalize_per_feature then spreads it
    over every valid column of every row (the row mean, src/mel.rs:721-749).  Same NaN pattern as the oracle, same values elsewhere."""
    kw = dict(n_mels=80, preemphasis=preemph, normalize_per_feature=norm)
    fe = gpu.BatchLogMelSpectrogram(gpu.BatchLogMelConfig(**kw))
    cfg = oracle.blm_default_config(**kw)
    n = 16000
    x = _loud(oracle, n, 32); x[7001] = np.nan
    want, valid = oracle.blm_compute(x, cfg, True)
    if norm:
        want = oracle.blm_normalize(want, valid)
    got = fe.compute(x)
    assert got.shape == want.shape
    assert (np.isnan(got) == np.isnan(want)).all()
    ok = ~np.isnan(want)
    if ok.any():
        assert np.abs(got[ok] - want[ok]).max() <= TOL * max(1.0, float(np.abs(want[ok]).max()))
    assert np.isnan(want).any()
    # neighbours in a batch stay clean
    clips = np.stack([_loud(oracle, n, 33), x, _loud(oracle, n, 34)])
    gb = fe.compute_batch(clips)
    assert (np.isnan(gb[1]) == np.isnan(want)).all() and np.isfinite(gb[0]).all() and np.isfinite(gb[2]).all()
    w0, v0 = oracle.blm_compute(clips[0], cfg, True)
    if norm:
        w0 = oracle.blm_normalize(w0, v0)
    assert np.abs(gb[0] - w0).max() <= TOL * max(1.0, float(np.abs(w0).max()))
    fe.close()


def test_quantiser_and_vad_on_images_with_non_finite_pixels(gpu, oracle):
    """quantize (src/quant.rs:140-153): f32::min / f32::max skip NaN, a +Inf maximum makes the scale 0 and every pixel 0 * x -> 0 or NaN -> 0.
    vad_boundaries (src/vad.rs:256-415): the Sobel gradient of a NaN / Inf pixel is NaN, and NaN >= threshold is false.  Bytes and masks
    equal the oracle's."""
    from mel_spec_amd import TgaCodec, DetectionSettings, vad_boundaries
    m = gpu.HipMelSpectrogram(400, 160, SR, 80)
    x = _loud(oracle, 32000, 48)
    x[9000] = np.inf
    mel = m.compute_mel_spectrogram(x)                 # rows of +Inf or -1.5 where the sample is
    assert np.isposinf(mel).any() or (mel == -1.5).all(axis=1).any()
    img = oracle.interleave_frames(mel, False, 0).ravel()
    codec = TgaCodec()
    for image in (img, np.where(np.arange(img.size) % 97 == 0, np.nan, img).astype(np.float32)):
        assert codec.tga_8bit_data(image, 80) == oracle.tga_8bit_data(image, 80)
    codec.close()
    frames = [mel[i: i + 40].T.copy() for i in range(0, mel.shape[0] - 40, 40)]
    frames.append(np.where(np.random.default_rng(1).random((80, 40)) < 0.02, np.nan, frames[0]).astype(np.float32))
    for fr in frames:
        e = vad_boundaries([fr], DetectionSettings(min_energy=0.3, min_y=3, min_x=5, min_mel=2))
        raw, smooth = oracle.vad_boundaries(fr, 0.3, 3, 5, 2)
        assert (np.asarray(e.raw, bool) == raw).all() and (np.asarray(e.smoothed, bool) == smooth).all()
    m.close()


def test_streaming_bank_nan_chunk(gpu, oracle):
    """a NaN inside a pushed chunk poisons exactly the frames whose window holds it (carry included), like Spectrogram::add (src/stft.rs:48-86)"""
    from mel_spec_amd import StreamBank
    m = gpu.HipMelSpectrogram(400, 160, SR, 80)
    bank = StreamBank(m, 2, 4000)
    x = _loud(oracle, 20000, 56); x[4100] = np.nan
    want = oracle.stream_mel(x, 400, 160, 80, SR)
    rows = []
    for i in range(0, 20000, 4000):
        rows += [r for r in bank.push([1], [x[i: i + 4000]])[0]]
    got = np.stack(rows) if rows else np.zeros((0, 80), np.float32)
    k = min(len(got), len(want))
    assert k > 100 and np.abs(got[:k] - want[:k]).max() <= TOL and (want[:k] == -1.5).all(axis=1).any()
    bank.close(); m.close()
