"""CPU check of the fused kernels' arithmetic: tests/emu compiles the very phase functions the
HIP kernels run (mel_spec_amd/csrc/whisper_six.hpp, whisper_wave.hpp, ...) for the host and executes them lane by
lane.  This validates the FFT index algebra and the f32 error budget (<= 1e-4 vs the f64
oracle) without a GPU; the GPU tests then only have to confirm the device agrees."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

TOL = 1e-4   # north_star: within 1e-4 (f32) of the CPU reference


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", d, "-s"])
    L = C.CDLL(os.path.join(d, "libmelspec_emu.so"))
    f32p = C.POINTER(C.c_float)
    u8p = C.POINTER(C.c_uint8)
    L.emu_whisper_six_guard.restype = C.c_longlong
    L.emu_whisper_six_guard.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p, u8p]
    L.emu_whisper_wave_guard.restype = C.c_longlong
    L.emu_whisper_wave_guard.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p, u8p]
    L.emu_whisper_auto.restype = C.c_longlong
    L.emu_whisper_auto.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, f32p, C.POINTER(C.c_longlong)]
    L.emu_small_fft.argtypes = [C.c_int, f32p]
    L.emu_whisper_wave.restype = C.c_longlong
    L.emu_whisper_wave.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p]
    L.emu_whisper_precise.restype = C.c_longlong
    L.emu_whisper_precise.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, f32p]
    L.emu_whisper_six.restype = C.c_longlong
    L.emu_whisper_six.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p]
    L.emu_whisper_six64.restype = C.c_longlong
    L.emu_whisper_six64.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_int, f32p]
    L.emu_w512_wave.restype = C.c_longlong
    L.emu_w512_wave.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, f32p]
    L.emu_blm_wave.restype = C.c_longlong
    L.emu_blm_wave.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                               C.c_float, C.c_int, C.c_float, C.c_longlong, f32p]
    L.emu_fbank_wave.restype = C.c_longlong
    L.emu_fbank_wave.argtypes = [f32p, C.c_longlong, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_float, C.c_int, C.c_int, C.c_int, f32p]

    def run(x, hop=160, n_mels=80, sr=16000.0):
        """the f32 kernel the library picks for this bank: six frames per wave up to 80 mels, five above"""
        x = np.ascontiguousarray(x, np.float32)
        nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
        out = np.full((nf, n_mels), np.nan, np.float32)
        fn = L.emu_whisper_six if n_mels <= 80 else L.emu_whisper_wave
        got = fn(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, 0, out.ctypes.data_as(f32p))
        assert got == nf
        return out

    run.lib = L
    return run


@pytest.mark.parametrize("logm", [6, 7, 8, 9, 10])
def test_pow2_stockham_passes_in_place(emu, logm):
    """pow2_wave.hpp (round 4): the complex M-point transform of pow2_frame_kernel -- radix 8 / 16 / 2 / 4 Stockham passes written back IN
    PLACE into the frame's padded LDS region -- run lane by lane on the host, every pass on a snapshot of its input: the transform is
    numpy's to f64 rounding, so no lane of a pass reads a word another lane of the same pass writes."""
    M = 1 << logm
    rng = np.random.default_rng(logm)
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    buf = np.ascontiguousarray(z).view(np.float64).copy()
    out = np.zeros(2 * M)
    dp = C.POINTER(C.c_double)
    emu.lib.emu_pow2_fft.restype = C.c_int
    emu.lib.emu_pow2_fft.argtypes = [C.c_int, dp, dp]
    assert emu.lib.emu_pow2_fft(logm, buf.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0
    want = np.fft.fft(z)
    assert np.abs(out.view(np.complex128) - want).max() <= 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("n", [8, 10, 16, 20])
def test_small_dfts(emu, n):
    rng = np.random.default_rng(n)
    z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    buf = z.view(np.float32).copy()
    emu.lib.emu_small_fft(n, buf.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.abs(buf.view(np.complex64) - np.fft.fft(z.astype(np.complex128))).max() < 3e-6


@pytest.mark.parametrize("n_mels", [80, 128])
def test_jfk(emu, oracle, jfk, n_mels):
    got = emu(jfk, n_mels=n_mels)
    want = oracle.compute_mel_spectrogram_cpu(jfk, 400, 160, n_mels)
    assert got.shape == want.shape == (1098, n_mels)
    assert np.abs(got - want).max() <= TOL


def test_four_tone_and_noise(emu, oracle, four_tone):
    assert np.abs(emu(four_tone) - oracle.compute_mel_spectrogram_cpu(four_tone)).max() <= TOL
    for c in (0, 3, 7):
        x = oracle.synth_pcm(c, 16000)
        assert np.abs(emu(x) - oracle.compute_mel_spectrogram_cpu(x)).max() <= TOL


@pytest.mark.parametrize("hop,n_mels", [(160, 64), (128, 80), (200, 40), (320, 100), (2, 80)])
def test_other_hops_and_mel_counts(emu, oracle, jfk, hop, n_mels):
    x = jfk[20000:20000 + (6000 if hop > 2 else 500)]
    got = emu(x, hop=hop, n_mels=n_mels)
    want = oracle.compute_mel_spectrogram_cpu(x, 400, hop, n_mels)
    assert got.shape == want.shape and np.abs(got - want).max() <= TOL


def test_edges(emu, oracle):
    assert emu(np.zeros(399, np.float32)).shape == (0, 80)
    for n in (400, 559, 560, 400 + 22 * 160, 400 + 23 * 160):     # one frame .. tile boundary
        x = oracle.synth_pcm(1, n)
        got, want = emu(x), oracle.compute_mel_spectrogram_cpu(x)
        assert got.shape == want.shape and np.abs(got - want).max() <= TOL
    assert np.all(emu(np.zeros(4000, np.float32)) == np.float32(-1.5))
    # a loud click inside digital silence: 8 decades of per-frame dynamic range
    x = np.zeros(4000, np.float32); x[1234] = 1.0
    assert np.abs(emu(x) - oracle.compute_mel_spectrogram_cpu(x)).max() <= TOL


# ---- wave-autonomous kernels (whisper_wave.hpp / fbank_wave.hpp) -------------------------------

def _wave(emu, x, mode, hop=160, n_mels=80, sr=16000.0):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    f32p = C.POINTER(C.c_float)
    got = emu.lib.emu_whisper_wave(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, mode, out.ctypes.data_as(f32p))
    return got, out


@pytest.mark.parametrize("mode", [0, 1])   # run-time / compile-time slot lengths
@pytest.mark.parametrize("n_mels", [80, 128])
def test_wave_kernel_modes(emu, oracle, jfk, mode, n_mels):
    x = jfk[8000:60000]
    got, out = _wave(emu, x, mode, n_mels=n_mels)
    want = oracle.compute_mel_spectrogram_cpu(x, 400, 160, n_mels)
    assert got == want.shape[0] and np.abs(out - want).max() <= TOL


@pytest.mark.parametrize("hop,n_mels,sr", [(160, 64, 16000.0), (128, 40, 8000.0), (320, 100, 22050.0), (160, 1, 16000.0),
                                            (160, 131, 16000.0), (2, 80, 16000.0)])
def test_wave_kernel_interval_scheme_other_filterbanks(emu, oracle, jfk, hop, n_mels, sr):
    x = jfk[20000:20000 + (9000 if hop > 2 else 500)]
    got, out = _wave(emu, x, 0, hop=hop, n_mels=n_mels, sr=sr)
    want = oracle.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)
    assert got == want.shape[0] and np.abs(out - want).max() <= TOL


def test_wave_kernel_edges(emu, oracle):
    for n in (400, 559, 560, 400 + 4 * 160, 400 + 5 * 160, 400 + 11 * 160 + 1):
        x = oracle.synth_pcm(1, n)
        for mode in (0, 1):
            got, out = _wave(emu, x, mode)
            want = oracle.compute_mel_spectrogram_cpu(x)
            assert got == want.shape[0] and np.abs(out - want).max() <= TOL
    got, out = _wave(emu, np.zeros(4000, np.float32), 1)
    assert np.all(out == np.float32(-1.5))


def tone_over_noise_floor(n=32000, level_db=-70.0, f=3333.3, seed=0):
    """Worst case for f32 spectra: a near-full-scale tone puts the per-frame clamp (max - 8) right at the
    level of the noise-only bands, whose f32 FFT error is set by the tone's rounding noise."""
    t = np.arange(n) / 16000.0
    rng = np.random.default_rng(seed)
    return (0.9 * np.sin(2 * np.pi * f * t) + 10 ** (level_db / 20) * rng.standard_normal(n)).astype(np.float32)


def _precise(emu, x, hop=160, n_mels=80, sr=16000.0):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    f32p = C.POINTER(C.c_float)
    assert emu.lib.emu_whisper_precise(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, out.ctypes.data_as(f32p)) == nf
    return out


@pytest.mark.parametrize("n_mels", [80, 128])
def test_precise_kernel_is_f64_accurate(emu, oracle, jfk, n_mels):
    want = oracle.compute_mel_spectrogram_cpu(jfk, 400, 160, n_mels)
    assert np.abs(_precise(emu, jfk, n_mels=n_mels) - want).max() <= 2e-6


def _six64(emu, x, hop=160, n_mels=80, sr=16000.0, mode=0):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    f32p = C.POINTER(C.c_float)
    assert emu.lib.emu_whisper_six64(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, mode, out.ctypes.data_as(f32p)) == nf
    return out


@pytest.mark.parametrize("hop,n_mels,n,mode", [(160, 80, None, 1), (160, 80, None, 0), (128, 40, 3000, 0), (320, 64, 5000, 0), (160, 80, 400, 1),
                                               (160, 80, 400 + 5 * 160 + 3, 1), (160, 80, 400 + 6 * 160, 1), (1, 8, 420, 0)])
def test_six_frame_f64_kernel_exchange_in_two_halves(emu, oracle, jfk, hop, n_mels, n, mode):
    """whisper_six64.hpp on the host, step by step in the kernel's order (rows 0..9, first read, rows 10..19 over them, second read):
    f64-accurate on speech and on the tone-over-a-quiet-floor signal the f32 kernels cannot hold, every slot count, partial last units."""
    x = jfk if n is None else jfk[20000:20000 + n]
    want = oracle.compute_mel_spectrogram_cpu(x, 400, hop, n_mels)
    got = _six64(emu, x, hop=hop, n_mels=n_mels, mode=mode)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6
    if n is None:
        t = np.arange(32000) / 16000.0
        y = (0.9 * np.sin(2 * np.pi * 3333.3 * t) + 10 ** (-70 / 20) * np.random.default_rng(0).standard_normal(32000)).astype(np.float32)
        assert np.abs(_six64(emu, y, mode=mode) - oracle.compute_mel_spectrogram_cpu(y, 400, 160, 80)).max() <= 2e-6
        assert (_six64(emu, np.zeros(4000, np.float32), mode=mode) == -1.5).all()


@pytest.mark.parametrize("n_mels,n,mode", [(128, None, 2), (128, 400 + 6 * 160 + 1, 2), (128, None, 3), (100, 9000, 3), (96, 5000, 3), (134, 5000, 3)])
def test_six_frame_f64_kernel_with_fifteen_mel_slots(emu, oracle, jfk, n_mels, n, mode):
    """The same kernel over the fifteen-slot mel section (round 5: Whisper large-v3's 128-mel bank, LensSix128; run-time lengths for the other
    banks of 81..134 mels, which the library leaves on the five-frame kernel): slot tables, start bins, the ghost lane of every slot."""
    x = jfk if n is None else jfk[30000:30000 + n]
    want = oracle.compute_mel_spectrogram_cpu(x, 400, 160, n_mels)
    got = _six64(emu, x, n_mels=n_mels, mode=mode)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6


def _auto(emu, x, hop=160, n_mels=80, sr=16000.0):
    """MELSPEC_PRECISION_AUTO emulated: f32 kernel + the frames its guard queues recomputed by the f64 kernel."""
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    flagged = C.c_longlong(0)
    f32p = C.POINTER(C.c_float)
    assert emu.lib.emu_whisper_auto(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, out.ctypes.data_as(f32p), C.byref(flagged)) == nf
    return out, int(flagged.value)


def test_f32_worst_case_and_what_the_modes_do_with_it(emu, oracle):
    """A near-full-scale tone over a noise floor ~80 dB below it in the mel domain: every f32 FFT (pocketfft's
    too) leaves the tone's rounding noise in the noise-only bins, and the bands just above the per-frame clamp
    miss 1e-4 (7 kHz: up to ~4e-4) -- MELSPEC_PRECISION_F32.  The default, AUTO, queues exactly these frames for the
    f64 kernel and holds the tolerance; F64 does not care.  (The reference's CUDA back-end also runs a Z2Z (f64) FFT,
    src/cuda.rs:204-219.)"""
    for f, lo, hi in ((3333.3, 2e-5, 1.5e-4), (7000.0, 1e-4, 6e-4)):
        x = tone_over_noise_floor(f=f)
        want = oracle.compute_mel_spectrogram_cpu(x)
        d32 = np.abs(_wave(emu, x, 1)[1] - want).max()
        d32six = np.abs(_six(emu, x, 1)[1] - want).max()
        d64 = np.abs(_precise(emu, x) - want).max()
        auto, flagged = _auto(emu, x)
        assert lo < d32 < hi and lo < d32six < hi, (f, d32, d32six)
        assert d64 <= 2e-6
        assert np.abs(auto - want).max() <= TOL and flagged > 0, (f, np.abs(auto - want).max(), flagged)


@pytest.mark.parametrize("n_mels,hop,sr", [(80, 160, 16000.0), (128, 160, 16000.0), (20, 160, 8000.0), (100, 320, 22050.0), (40, 200, 16000.0)])
def test_auto_mode_holds_the_tolerance_on_a_zoo_of_hard_signals(emu, oracle, jfk, n_mels, hop, sr):
    """The guard threshold (kGuardBand in whisper_wave.hpp) was calibrated with tools/flag_calib.py; this is the
    condensed form: tones, a chirp and impulse trains over floors of every level, speech, noise, silence."""
    rng = np.random.default_rng(42)
    n = 24000
    t = np.arange(n) / sr
    sigs = []
    for f in (0.0125, 0.0625, 0.2083, 0.3125, 0.4375, 0.4875):                 # fractions of the sampling rate
        for lv in (-50, -65, -75, -85):
            sigs.append((0.9 * np.sin(2 * np.pi * f * sr * t) + 10 ** (lv / 20) * rng.standard_normal(n)).astype(np.float32))
    fch = 100 + (0.48 * sr - 100) * (t / t[-1])
    sigs.append((0.8 * np.sin(2 * np.pi * np.cumsum(fch) / sr) + 10 ** (-70 / 20) * rng.standard_normal(n)).astype(np.float32))
    x = rng.standard_normal(n).astype(np.float32) * np.float32(1e-4); x[::173] += 0.7; sigs.append(x)
    sigs.append(jfk[30000:54000])
    worst = 0.0
    for x in sigs:
        got, _ = _auto(emu, x, hop, n_mels, sr)
        worst = max(worst, float(np.abs(got - oracle.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)).max()))
    assert worst <= TOL, worst
    # what must NOT be queued: noise of any level, digital silence, a click in silence
    for x in (oracle.synth_pcm(0, n), oracle.synth_pcm(7, n), np.zeros(n, np.float32)):
        got, flagged = _auto(emu, x, hop, n_mels, sr)
        assert flagged == 0 and np.abs(got - oracle.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)).max() <= TOL


@pytest.mark.parametrize("hop,n_mels,n", [(128, 40, 3000), (320, 100, 5000), (160, 80, 400), (160, 80, 400 + 5 * 160 + 3)])
def test_precise_kernel_geometries_and_edges(emu, oracle, jfk, hop, n_mels, n):
    x = jfk[20000:20000 + n]
    want = oracle.compute_mel_spectrogram_cpu(x, 400, hop, n_mels)
    got = _precise(emu, x, hop=hop, n_mels=n_mels)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6


def _fbank(emu, x, f64=1, shift=160, n_mels=80, sr=16000.0, low=20.0, high=8000.0, preemph=0.97, use_log=1, use_power=1,
           floor_v=float(np.finfo(np.float32).eps)):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // shift + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    f32p = C.POINTER(C.c_float)
    got = emu.lib.emu_fbank_wave(x.ctypes.data_as(f32p), len(x), shift, n_mels, sr, low, high, preemph, floor_v, use_log,
                                 use_power, f64, out.ctypes.data_as(f32p))
    return got, out


def test_fbank_fused_f64_matches_oracle(emu, oracle, jfk):
    cfg = oracle.fbank_default_config(); cfg.apply_cmn = 0
    for x in (jfk, oracle.synth_pcm(3, 16000), oracle.synth_pcm(6, 400 + 7 * 160)):
        got, out = _fbank(emu, x)
        want = oracle.fbank_compute(x, cfg)
        assert got == want.shape[0] and np.abs(out - want).max() <= 2e-5


def test_fbank_fused_f32_cannot_hold_the_tolerance_on_speech(emu, oracle, jfk):
    """Documents why the parity build of the fbank kernel is f64 (fbank_wave.hpp header)."""
    cfg = oracle.fbank_default_config(); cfg.apply_cmn = 0
    got, out = _fbank(emu, jfk, f64=0)
    d = np.abs(out - oracle.fbank_compute(jfk, cfg))
    assert d.max() > TOL and d.mean() < 1e-4


def test_fbank_fused_variants(emu, oracle, jfk):
    x = jfk[40000:56000]
    for kw, args in ((dict(num_mel_bins=40), dict(n_mels=40)), (dict(preemphasis=0.0), dict(preemph=0.0)),
                     (dict(use_power=0), dict(use_power=0)), (dict(use_log_fbank=0), dict(use_log=0)),
                     (dict(low_freq=100.0, high_freq=7000.0), dict(low=100.0, high=7000.0)),
                     (dict(frame_shift_ms=6.3125), dict(shift=101))):
        cfg = oracle.fbank_default_config(); cfg.apply_cmn = 0
        for k_, v in kw.items():
            setattr(cfg, k_, type(getattr(cfg, k_))(v))
        got, out = _fbank(emu, x, **args)
        want = oracle.fbank_compute(x, cfg)
        tol = 2e-5 if kw.get("use_log_fbank", 1) else 2e-5 * max(1.0, float(np.abs(want).max()))
        assert got == want.shape[0] and np.abs(out - want).max() <= tol, kw


@pytest.mark.parametrize("kw", [dict(), dict(n_mels=128, preemphasis=0.97, log_zero_guard=2.0 ** -24), dict(center=0, n_mels=64),
                                dict(pad_to=16, preemphasis=0.5), dict(htk=1, norm=0, f_min=50.0, f_max=7000.0)])
def test_nemo_frontend_fused_f64_matches_oracle(emu, oracle, jfk, kw):
    """BatchLogMelSpectrogram (src/mel.rs:299-385) on the fused 512-point kernel, against the f64 evaluation of the
    reference's definition; the literal f32 restatement is reported like the reference reports its NeMo distance."""
    cfg = oracle.blm_default_config(**kw)
    f32p = C.POINTER(C.c_float)
    for x in (jfk[30000:70000], oracle.synth_pcm(2, 16007), oracle.synth_pcm(4, 300)):
        want, valid = oracle.blm_compute(x, cfg, True)
        if want.size == 0:
            continue
        out = np.full_like(want, np.nan)
        got = emu.lib.emu_blm_wave(x.ctypes.data_as(f32p), len(x), cfg.hop_length, cfg.n_mels, cfg.sample_rate, cfg.f_min,
                                   cfg.f_max, cfg.htk, cfg.norm, cfg.preemphasis, cfg.center, cfg.log_zero_guard,
                                   want.shape[1], out.ctypes.data_as(f32p))
        assert got == valid and np.abs(out - want).max() <= 2e-5
        lit, _ = oracle.blm_compute(x, cfg, False)
        assert np.abs(out - lit).max() < 5e-3 and np.abs(out - lit).mean() < 1e-5


# ---- Whisper flavour of the fused 512-point kernel (n_fft = 512, f64 FFT) -----------------------------------

def _w512(emu, x, hop=160, n_mels=80, sr=16000.0):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 512 else (len(x) - 512) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    f32p = C.POINTER(C.c_float)
    assert emu.lib.emu_w512_wave(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, out.ctypes.data_as(f32p)) == nf
    return out


def test_whisper512_flavour_reproduces_the_reference_golden(emu, jfk):
    """rust_jfk_golden.npy (src/rb.rs:134-179, 512/160/80, @1e-6 in the reference) = batch frames of samples[128:]."""
    want = np.load(os.path.join(ROOT, "tests", "golden", "rust_jfk_golden.npy"))
    got = _w512(emu, jfk[128:])
    assert got[:1097].T.shape == want.shape and np.abs(got[:1097].T - want).max() <= 1e-6


@pytest.mark.parametrize("hop,n_mels,sr,n", [(160, 80, 16000.0, 40000), (128, 128, 16000.0, 20000), (200, 40, 8000.0, 9000), (161, 80, 16000.0, 9000),
                                             (160, 80, 16000.0, 512), (160, 80, 16000.0, 512 + 3 * 160), (160, 80, 16000.0, 512 + 4 * 160 + 7)])
def test_whisper512_flavour_matches_oracle(emu, oracle, jfk, hop, n_mels, sr, n):
    x = jfk[20000:20000 + n]
    want = oracle.compute_mel_spectrogram_cpu(x, 512, hop, n_mels, sr)
    got = _w512(emu, x, hop, n_mels, sr)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6
    assert _w512(emu, np.zeros(511, np.float32)).shape == (0, 80)


# ---- six frames per wave (whisper_six.hpp), the default build for <= 80 mels ------------------------------------

def _six(emu, x, mode, hop=160, n_mels=80, sr=16000.0):
    x = np.ascontiguousarray(x, np.float32)
    nf = 0 if len(x) < 400 else (len(x) - 400) // hop + 1
    out = np.full((nf, n_mels), np.nan, np.float32)
    f32p = C.POINTER(C.c_float)
    got = emu.lib.emu_whisper_six(x.ctypes.data_as(f32p), len(x), hop, n_mels, sr, mode, out.ctypes.data_as(f32p))
    return got, out


@pytest.mark.parametrize("mode", [0, 1])      # runtime / compile-time slot lengths
def test_six_frame_kernel_jfk(emu, oracle, jfk, mode):
    got, out = _six(emu, jfk, mode)
    want = oracle.compute_mel_spectrogram_cpu(jfk, 400, 160, 80)
    assert got == 1098 and np.abs(out - want).max() <= TOL


@pytest.mark.parametrize("hop,n_mels,sr", [(160, 64, 16000.0), (128, 40, 8000.0), (320, 80, 22050.0), (160, 1, 16000.0), (200, 79, 16000.0)])
def test_six_frame_kernel_other_filterbanks(emu, oracle, jfk, hop, n_mels, sr):
    x = jfk[20000:27000]
    got, out = _six(emu, x, 0, hop, n_mels, sr)
    want = oracle.compute_mel_spectrogram_cpu(x, 400, hop, n_mels, sr)
    assert got == want.shape[0] and np.abs(out - want).max() <= TOL


def test_six_frame_kernel_edges_and_coverage(emu, oracle, four_tone):
    assert _six(emu, np.zeros(399, np.float32), 1)[0] == 0
    for n in (400, 559, 560, 400 + 5 * 160, 400 + 6 * 160, 400 + 11 * 160 + 3):       # around the 6-frame unit size
        x = oracle.synth_pcm(2, n)
        got, out = _six(emu, x, 1)
        want = oracle.compute_mel_spectrogram_cpu(x)
        assert got == want.shape[0] and np.abs(out - want).max() <= TOL
    z = np.zeros(8000, np.float32); z[4321] = 1.0
    assert np.abs(_six(emu, z, 1)[1] - oracle.compute_mel_spectrogram_cpu(z)).max() <= TOL
    assert np.abs(_six(emu, four_tone, 1)[1] - oracle.compute_mel_spectrogram_cpu(four_tone)).max() <= TOL
    assert _six(emu, np.zeros(4000, np.float32), 0, n_mels=81)[0] == -1           # 82 intervals > 9 slots of 9: not covered
    assert _six(emu, np.zeros(4000, np.float32), 1, n_mels=64)[0] == -2           # compile-time lengths are Whisper-80 only


@pytest.mark.parametrize("n_mels,six", [(80, 1), (80, 0), (128, 0), (40, 1)])
def test_in_kernel_f64_recompute_is_f64_accurate(emu, oracle, jfk, n_mels, six):
    """whisper_fix64.hpp: one frame recomputed by a whole wavefront (200 = 8 x 25 Good-Thomas, 25 = 5 x 5), as the f32 kernels
    run it for the frames their precision guard trips -- within 2e-6 of the oracle on speech, a tone over a floor, noise, silence."""
    emu.lib.emu_fix_frame.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_double, C.c_int, C.POINTER(C.c_float)]
    f32p = C.POINTER(C.c_float)
    frames = [jfk[20000 + 160 * i:20400 + 160 * i] for i in range(0, 60, 7)]
    frames += [tone_over_noise_floor(f=7000.0)[1000:1400], tone_over_noise_floor(f=333.3, level_db=-85.0)[555:955], oracle.synth_pcm(2, 400), np.zeros(400, np.float32)]
    for x in frames:
        x = np.ascontiguousarray(x, np.float32)
        out = np.full(n_mels, np.nan, np.float32)
        assert emu.lib.emu_fix_frame(x.ctypes.data_as(f32p), n_mels, 16000.0, six, out.ctypes.data_as(f32p)) == 0
        want = oracle.compute_mel_spectrogram_cpu(x, 400, 160, n_mels)[0]
        assert np.abs(out - want).max() <= 2e-6


def _clip_kernel_cmn(x_no_cmn: np.ndarray, waves: int = 8, fpw: int = 4) -> np.ndarray:
    """The column means of fbank512_clip_kernel, in its order, in f32: a wave owns a contiguous eighth of the clip's units of four
    frames and keeps one running sum per frame position; positions (0+1)+(2+3); waves ((0+1)+(2+3))+((4+5)+(6+7)); one division."""
    frames, nm = x_no_cmn.shape
    units = (frames + fpw - 1) // fpw
    part = np.zeros((waves, nm), np.float32)
    for w in range(waves):
        acc = np.zeros((fpw, nm), np.float32)
        for u in range(units * w // waves, units * (w + 1) // waves):
            for fl in range(fpw):
                f = u * fpw + fl
                if f < frames:
                    acc[fl] = acc[fl] + x_no_cmn[f]
        part[w] = (acc[0] + acc[1]) + (acc[2] + acc[3])
    s = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]))
    return x_no_cmn - (s / np.float32(frames)).astype(np.float32)


@pytest.mark.parametrize("n", [160000, 11357, 400, 4000])
def test_clip_kernel_column_sums_stay_inside_the_tolerance(oracle, jfk, n):
    """fbank512_clip_kernel does not fold the CMN's column sums in the reference's order (src/fbank.rs:224-233: ndarray's mean() of a
    strided column, an f32 left fold over the frames) but as a fixed tree.  The tree in f32 on the oracle's un-normalised features
    against the oracle's own CMN: far inside 1e-4 (the reference's fold itself carries ~1e-5 of rounding at 1000 frames)."""
    oc = oracle.fbank_default_config(); oc.apply_cmn = 0
    for x in (jfk[:n], oracle.synth_pcm(3, n)):
        raw = oracle.fbank_compute(x, oc)
        want = oracle.fbank_compute(x)
        got = _clip_kernel_cmn(raw)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 5e-5
        assert np.abs(got.mean(axis=0)).max() < 1e-4


def _blm_normalise_like_the_kernel(rows: np.ndarray, valid: int, pp: int = 28) -> np.ndarray:
    """blm_normalize_kernel's arithmetic in f32: the mean as a left fold (the reference's order), the variance as `pp` strided partial
    sums per row with four accumulators each, added in order, one reciprocal per row."""
    out = rows.copy()
    for r in range(rows.shape[0]):
        v = rows[r, :valid].astype(np.float32)
        s = np.float32(0.0)
        for x in v:
            s = np.float32(s + x)
        mean = np.float32(s / np.float32(valid))
        parts = np.zeros(pp, np.float32)
        for pt in range(pp):
            a = [np.float32(0.0)] * 4
            idx = list(range(pt, valid, pp))
            full = (len(idx) // 4) * 4
            for i in range(0, full, 4):
                for k in range(4):
                    c = np.float32(v[idx[i + k]] - mean)
                    a[k] = np.float32(a[k] + np.float32(c * c))
            for i in range(full, len(idx)):
                c = np.float32(v[idx[i]] - mean)
                a[0] = np.float32(a[0] + np.float32(c * c))
            parts[pt] = np.float32(np.float32(a[0] + a[1]) + np.float32(a[2] + a[3]))
        q = [np.float32(0.0)] * 4
        full = (pp // 4) * 4
        for i in range(0, full, 4):
            for k in range(4):
                q[k] = np.float32(q[k] + parts[i + k])
        for i in range(full, pp):
            q[0] = np.float32(q[0] + parts[i])
        qq = np.float32(np.float32(q[0] + q[1]) + np.float32(q[2] + q[3]))
        denom = np.float32(max(valid - 1, 1))
        rsd = np.float32(1.0) / np.float32(np.sqrt(np.float32(qq / denom)) + np.float32(1e-5))
        out[r, :valid] = ((v - mean) * rsd).astype(np.float32)
    return out


def test_nemo_normaliser_order_stays_inside_the_tolerance(oracle, jfk):
    """The per-feature normaliser keeps the reference's left fold for the mean (src/mel.rs:721-749) and takes the variance as a fixed tree
    and the division as a multiplication by the reciprocal: restated in f32 on the oracle's un-normalised features against the oracle's
    literal normaliser."""
    kw = dict(n_mels=80, preemphasis=0.97)
    x = jfk[5000:5000 + 48000]
    raw, valid = oracle.blm_compute(x, oracle.blm_default_config(**kw), False)
    want, _ = oracle.blm_compute(x, oracle.blm_default_config(normalize_per_feature=True, **kw), False)
    got = _blm_normalise_like_the_kernel(np.asarray(raw, np.float32), int(valid))
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5
