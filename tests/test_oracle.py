"""Pins the CPU oracle (oracle/melspec_oracle.c) to every fixture / known-answer the
reference's own tests hold for the hot path."""
import os

import numpy as np

from conftest import GOLDEN


def test_mel_filterbank_matches_whisper_fixture(oracle):
    # src/mel.rs:838-850: mel(16000, 400, 80, None, None, false, true) vs mel_filters.npz @1e-7
    want = np.load(os.path.join(GOLDEN, "mel_filters.npz"))["mel_80"].astype(np.float64)
    got = oracle.mel_filterbank(16000.0, 400, 80)
    assert got.shape == (80, 201)
    assert np.abs(got - want).max() <= 1e-7


def test_mel_filterbank_matches_nemo_fixture(oracle):
    # src/mel.rs:853-871
    want = np.load(os.path.join(GOLDEN, "nemo_mel_filters.npz"))["banks"][0].astype(np.float64)
    got = oracle.mel_filterbank(16000.0, 512, 80)
    assert got.shape == want.shape == (80, 257)
    assert np.abs(got - want).max() <= 1e-7


def test_librosa_known_answers(oracle):
    # src/mel.rs:787-835
    assert abs(oracle.hz_to_mel(60.0) - 0.9) <= 1e-3
    assert oracle.mel_to_hz(3.0) == 200.0
    got = np.array([oracle.mel_to_hz(m) for m in (1.0, 2.0, 3.0, 4.0, 5.0)])
    assert np.abs(got - np.array([66.667, 133.333, 200.0, 266.667, 333.333])).max() <= 1e-3
    want = np.array([0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855,
                     853.173, 938.49, 1024.856, 1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532,
                     1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799, 3216.731, 3512.582, 3835.643,
                     4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107,
                     8467.272, 9246.028, 10096.408, 11025.])
    assert np.abs(oracle.mel_frequencies(40, 0.0, 11025.0) - want).max() <= 5e-3
    want = np.array([0., 1378.125, 2756.25, 4134.375, 5512.5, 6890.625, 8268.75, 9646.875, 11025.])
    assert np.abs(oracle.fft_frequencies(22050.0, 16) - want).max() <= 1e-3


def test_fft_is_the_forward_dft(oracle):
    rng = np.random.default_rng(1)
    for n in (2, 8, 20, 25, 200, 231, 400, 512, 97):
        z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert np.abs(oracle.fft_forward(z) - np.fft.fft(z)).max() <= 1e-11 * n


def test_streaming_path_reproduces_rust_jfk_golden(oracle, jfk):
    # src/rb.rs:134-179: streaming 512/160/80 on jfk_f32le.wav vs rust_jfk_golden.npy @1e-6
    want = np.load(os.path.join(GOLDEN, "rust_jfk_golden.npy"))
    got = oracle.stream_mel(jfk, 512, 160, 80, 16000.0)
    assert got.T.shape == want.shape == (80, 1097)
    assert np.abs(got.T - want).max() <= 1e-6


def test_batch_path_equals_streaming_with_alignment_offset(oracle, jfk):
    # streaming frame j covers samples[off + 160 j ...), off = ceil(n_fft/hop)*hop - n_fft
    want = np.load(os.path.join(GOLDEN, "rust_jfk_golden.npy"))
    got = oracle.compute_mel_spectrogram_cpu(jfk[128:], 512, 160, 80, 16000.0)
    assert np.abs(got[:1097].T - want).max() <= 1e-6
    s = oracle.stream_mel(jfk, 400, 160, 80, 16000.0)
    b = oracle.compute_mel_spectrogram_cpu(jfk[80:], 400, 160, 80, 16000.0)
    assert np.array_equal(s, b[:s.shape[0]])


def test_frame_count_edges(oracle):
    # src/stft.rs:153-157 and the shape tests of tests/readme_examples.rs:12-18
    assert oracle.num_frames(399, 400, 160) == 0
    assert oracle.num_frames(400, 400, 160) == 1
    assert oracle.num_frames(559, 400, 160) == 1
    assert oracle.num_frames(560, 400, 160) == 2
    assert oracle.num_frames(160000, 400, 160) == 998
    assert oracle.num_frames(480000, 400, 160) == 2998
    assert oracle.compute_mel_spectrogram_cpu(np.zeros(399, np.float32)).shape == (0, 80)
    out = oracle.compute_mel_spectrogram_cpu(np.zeros(16000, np.float32))
    assert out.shape == (98, 80) and np.all(out == np.float32(-1.5))


def test_fbank_shape_and_kaldi_sanity(oracle, jfk):
    # src/fbank.rs:484-490: frame count must equal the kaldi golden's; values are informational there
    k = np.load(os.path.join(GOLDEN, "kaldi_native_fbank_jfk.npz"))["features"]
    fb = oracle.fbank_compute(jfk)
    assert fb.shape == (k.shape[1], 80) == (1098, 80)
    assert np.all(np.isfinite(fb))
    assert float((fb.astype(np.float64) ** 2).mean()) > 0.1          # src/fbank.rs:533-534
    d = np.abs(fb.T - k)
    assert d.max() < 0.02 and d.mean() < 0.004                        # probed in SURVEY.md §0.4
    cfg = oracle.fbank_default_config(); cfg.apply_cmn = 0
    raw = oracle.fbank_compute(jfk, cfg)
    assert abs(float(raw[0, 0]) - np.log(np.float64(np.finfo(np.float32).eps))) < 1e-4   # src/fbank.rs:596
    assert oracle.fbank_compute(np.zeros(399, np.float32)).shape == (0, 80)
    assert oracle.fbank_compute(np.zeros(16000, np.float32)).shape == (98, 80)       # src/fbank.rs:389-402


def test_golden_file_is_current(oracle, jfk, golden, four_tone):
    assert np.array_equal(golden["jfk_w80"], oracle.compute_mel_spectrogram_cpu(jfk, 400, 160, 80))
    assert np.array_equal(golden["tone_w80"], oracle.compute_mel_spectrogram_cpu(four_tone, 400, 160, 80))
    assert np.array_equal(golden["jfk_fbank_cmn"], oracle.fbank_compute(jfk))


def test_synth_pcm_is_deterministic_and_bounded(oracle):
    a, b = oracle.synth_pcm(5, 4096), oracle.synth_pcm(5, 4096)
    assert np.array_equal(a, b) and np.abs(a).max() < 2.0 ** -5 + 1e-9 and a.std() > 0
    assert np.abs(oracle.synth_pcm(8, 4096)).max() <= 1.0


def test_interleave_frames_rules(oracle):
    # src/mel.rs:480-544 and tests/readme_examples.rs:60-71 (one 80-mel frame, min_width 2 -> 80 x 2)
    fr = np.arange(80, dtype=np.float32).reshape(1, 80) / 80
    out = oracle.interleave_frames(fr, False, 2)
    assert out.shape == (80, 2) and np.array_equal(out[:, 0], fr[0]) and not out[:, 1].any()
    f3 = np.arange(3 * 4, dtype=np.float32).reshape(3, 4) + 1
    assert oracle.interleave_frames(f3, False, 0).shape == (4, 3)              # no evening when min_width == 0
    assert oracle.interleave_frames(f3, False, 2).shape == (4, 4)              # odd count -> one zero frame
    assert oracle.interleave_frames(f3, False, 10).shape == (4, 10)
    assert np.array_equal(oracle.interleave_frames(f3, False, 0), f3.T)
    cm = oracle.interleave_frames(f3, True, 6)
    assert cm.shape == (6, 4) and np.array_equal(cm[:3], f3) and not cm[3:].any()
    import pytest
    with pytest.raises(ValueError):
        oracle.interleave_frames(f3, False, 3)


def test_batch_log_mel_shape_pins(oracle, jfk):
    # src/mel.rs:943-961: 128 mels, preemphasis 0.97, guard 2^-24, per-feature normalisation, 1 s of zeros -> (128, 101)
    cfg = oracle.blm_default_config(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24, normalize_per_feature=1)
    out, valid = oracle.blm_compute(np.zeros(16000, np.float32), cfg, f64=False)
    assert out.shape == (128, 101) and valid == 101 and np.all(np.isfinite(out))
    # README.md:146-149: the Parakeet frontend on jfk gives 128 x 1101
    cfg = oracle.blm_default_config(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24)
    a, _ = oracle.blm_compute(jfk, cfg, f64=False)
    b, _ = oracle.blm_compute(jfk, cfg, f64=True)
    assert a.shape == b.shape == (128, 1101)
    d = np.abs(a - b)          # the reference's own f32 noise against the exact definition
    assert d.max() < 2e-3 and d.mean() < 1e-5
    assert oracle.blm_compute(np.zeros(0, np.float32))[0].shape == (80, 0)       # src/mel.rs:326-332
    assert oracle.blm_compute(np.zeros(1000, np.float32), oracle.blm_default_config(pad_to=8))[0].shape == (80, 8)
