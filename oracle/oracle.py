"""ctypes binding of the CPU oracle (oracle/melspec_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The product package (mel_spec_amd) must never
import this module -- tests/test_boundary.py greps for that.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmelspec_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "melspec_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class FbankConfig(C.Structure):
    """Mirror of oracle_fbank_config (FbankConfig, src/fbank.rs:25-64)."""
    _fields_ = [
        ("sample_rate", C.c_double), ("num_mel_bins", C.c_int),
        ("frame_length_ms", C.c_double), ("frame_shift_ms", C.c_double),
        ("energy_floor", C.c_double), ("use_log_fbank", C.c_int), ("use_power", C.c_int),
        ("preemphasis", C.c_double), ("apply_cmn", C.c_int),
        ("low_freq", C.c_double), ("high_freq", C.c_double),
    ]


class BlmConfig(C.Structure):
    """Mirror of oracle_blm_config (BatchLogMelConfig, src/mel.rs:171-208)."""
    _fields_ = [
        ("sample_rate", C.c_int), ("n_fft", C.c_int), ("win_length", C.c_int), ("hop_length", C.c_int), ("n_mels", C.c_int),
        ("f_min", C.c_double), ("f_max", C.c_double), ("htk", C.c_int), ("norm", C.c_int), ("preemphasis", C.c_float),
        ("center", C.c_int), ("log_zero_guard", C.c_float), ("pad_to", C.c_int), ("normalize_per_feature", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.oracle_num_frames.restype = C.c_int64
        L.oracle_num_frames.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.oracle_hz_to_mel.restype = C.c_double
        L.oracle_hz_to_mel.argtypes = [C.c_double, C.c_int]
        L.oracle_mel_to_hz.restype = C.c_double
        L.oracle_mel_to_hz.argtypes = [C.c_double, C.c_int]
        L.oracle_mel_frequencies.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, f64p]
        L.oracle_fft_frequencies.argtypes = [C.c_double, C.c_int, f64p]
        L.oracle_hann_window.argtypes = [C.c_int, f64p]
        L.oracle_fft_forward.argtypes = [C.c_int, f64p]
        L.oracle_mel_filterbank.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, f64p]
        L.oracle_sparse_stats.restype = C.c_int
        L.oracle_sparse_stats.argtypes = [f64p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_compute_mel_spectrogram_cpu.restype = C.c_int64
        L.oracle_compute_mel_spectrogram_cpu.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, f32p]
        L.oracle_compute_mel_batch.restype = C.c_int64
        L.oracle_compute_mel_batch.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, f32p, C.c_int]
        L.oracle_stream_mel_ex.restype = C.c_int64
        L.oracle_stream_mel_ex.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, f32p, C.c_int64, C.c_int]
        L.oracle_interleave_frames.restype = C.c_int64
        L.oracle_interleave_frames.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, C.c_int64, f32p]
        u8p = C.POINTER(C.c_uint8)
        L.oracle_quantize.restype = None
        L.oracle_quantize.argtypes = [f32p, C.c_int64, u8p, f32p]
        L.oracle_dequantize.restype = None
        L.oracle_dequantize.argtypes = [u8p, C.c_int64, f32p, f32p]
        L.oracle_tga_8bit_data.restype = C.c_int64
        L.oracle_tga_8bit_data.argtypes = [f32p, C.c_int64, C.c_int, u8p]
        L.oracle_vad_boundaries.restype = C.c_int64
        L.oracle_vad_boundaries.argtypes = [f32p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, u8p, u8p]
        L.oracle_vad_longest_run.restype = C.c_int64
        L.oracle_vad_longest_run.argtypes = [u8p, C.c_int64]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_stream_mel.restype = C.c_int64
        L.oracle_stream_mel.argtypes = [f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, f32p, C.c_int64]
        L.oracle_fbank_default_config.argtypes = [C.POINTER(FbankConfig)]
        L.oracle_fbank_frame_length.argtypes = [C.POINTER(FbankConfig)]
        L.oracle_fbank_frame_shift.argtypes = [C.POINTER(FbankConfig)]
        L.oracle_fbank_fft_size.argtypes = [C.POINTER(FbankConfig)]
        L.oracle_kaldi_mel_filterbank.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, f64p]
        L.oracle_fbank_compute.restype = C.c_int64
        L.oracle_fbank_compute.argtypes = [C.POINTER(FbankConfig), f32p, C.c_int64, f32p]
        L.oracle_fbank_batch.restype = C.c_int64
        L.oracle_fbank_batch.argtypes = [C.POINTER(FbankConfig), f32p, C.c_int64, C.c_int64, C.c_int, f32p, C.c_int]
        L.oracle_synth_pcm.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, f32p]
        L.oracle_blm_default_config.argtypes = [C.POINTER(BlmConfig)]
        for name in ("oracle_blm_compute_f32", "oracle_blm_compute_f64"):
            fn = getattr(L, name)
            fn.restype = C.c_int64
            fn.argtypes = [C.POINTER(BlmConfig), f32p, C.c_int64, f32p, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def num_frames(n: int, n_fft: int, hop: int) -> int:
    return int(lib().oracle_num_frames(n, n_fft, hop))


def hann_window(n: int) -> np.ndarray:
    w = np.empty(n, np.float64)
    lib().oracle_hann_window(n, _p(w, C.c_double))
    return w


def fft_forward(z: np.ndarray) -> np.ndarray:
    z = np.ascontiguousarray(z, dtype=np.complex128).copy()
    lib().oracle_fft_forward(z.shape[0], _p(z.view(np.float64), C.c_double))
    return z


def compute_all_cpu(samples, fft_size=400, hop_size=160) -> np.ndarray:
    """Spectrogram::compute_all_cpu (src/stft.rs:89-115): frame_windows (periodic Hann, no padding, src/stft.rs:147-169),
    then the forward complex FFT of every frame -> [frames][fft_size] complex128 (the reference's Vec<Vec<Complex<f64>>>)."""
    x = _f32(samples).astype(np.float64)
    nf = num_frames(x.shape[0], fft_size, hop_size)
    w = hann_window(fft_size)
    out = np.empty((nf, fft_size), np.complex128)
    for f in range(nf):
        out[f] = fft_forward(x[f * hop_size:f * hop_size + fft_size] * w)
    return out


def mel_filterbank(sr: float, n_fft: int, n_mels: int, f_min=None, f_max=None, htk=False, norm=True) -> np.ndarray:
    """mel() of src/mel.rs:547-589 -> dense f64 [n_mels, n_fft//2+1]."""
    out = np.empty((n_mels, n_fft // 2 + 1), np.float64)
    lib().oracle_mel_filterbank(sr, n_fft, n_mels, -1.0 if f_min is None else f_min,
                                -1.0 if f_max is None else f_max, int(htk), int(norm), _p(out, C.c_double))
    return out


def mel_frequencies(n: int, fmin: float, fmax: float, htk=False) -> np.ndarray:
    out = np.empty(n, np.float64)
    lib().oracle_mel_frequencies(n, fmin, fmax, int(htk), _p(out, C.c_double))
    return out


def fft_frequencies(sr: float, n_fft: int) -> np.ndarray:
    out = np.empty(n_fft // 2 + 1, np.float64)
    lib().oracle_fft_frequencies(sr, n_fft, _p(out, C.c_double))
    return out


def hz_to_mel(f, htk=False):
    return float(lib().oracle_hz_to_mel(f, int(htk)))


def mel_to_hz(m, htk=False):
    return float(lib().oracle_mel_to_hz(m, int(htk)))


def compute_mel_spectrogram_cpu(samples, fft_size=400, hop_size=160, n_mels=80, sampling_rate=16000.0) -> np.ndarray:
    """Spectrogram::compute_mel_spectrogram_cpu (src/stft.rs:119-138) -> f32 [frames, n_mels]."""
    x = _f32(samples)
    nf = num_frames(x.shape[0], fft_size, hop_size)
    out = np.empty((nf, n_mels), np.float32)
    if nf:
        got = lib().oracle_compute_mel_spectrogram_cpu(_p(x, C.c_float), x.shape[0], fft_size, hop_size, n_mels,
                                                       sampling_rate, _p(out, C.c_float))
        assert got == nf
    return out


def compute_mel_batch(clips, fft_size=400, hop_size=160, n_mels=80, sampling_rate=16000.0, n_threads=0) -> np.ndarray:
    """[n_clips, clip_len] f32 -> [n_clips, frames, n_mels] f32, clips across OpenMP threads."""
    x = _f32(clips)
    assert x.ndim == 2
    nf = num_frames(x.shape[1], fft_size, hop_size)
    out = np.empty((x.shape[0], nf, n_mels), np.float32)
    if nf and x.shape[0]:
        lib().oracle_compute_mel_batch(_p(x, C.c_float), x.shape[1], x.shape[1], x.shape[0], fft_size, hop_size,
                                       n_mels, sampling_rate, _p(out, C.c_float), n_threads)
    return out


def interleave_frames(frames, major_column_order=False, min_width=0) -> np.ndarray:
    """interleave_frames (src/mel.rs:480-544) on [n_frames, n_mels] -> [n_mels, W] (or [W, n_mels])."""
    x = _f32(frames)
    nf, nm = x.shape
    W = lib().oracle_interleave_frames(_p(x, C.c_float), nf, nm, int(major_column_order), min_width, None)
    if W < 0:
        raise ValueError("frames is empty or min_width is odd")
    out = np.empty((W, nm) if major_column_order else (nm, W), np.float32)
    lib().oracle_interleave_frames(_p(x, C.c_float), nf, nm, int(major_column_order), min_width, _p(out, C.c_float))
    return out


# ---- src/quant.rs ---------------------------------------------------------------------------------

def quantize(frame):
    """quantize (src/quant.rs:140-153) -> (u8 array, (min, max))."""
    x = _f32(frame).ravel()
    out = np.empty(x.shape[0], np.uint8)
    rng = np.empty(2, np.float32)
    lib().oracle_quantize(_p(x, C.c_float), x.shape[0], _p(out, C.c_uint8), _p(rng, C.c_float))
    return out, (rng[0], rng[1])


def dequantize(data, rng) -> np.ndarray:
    """dequantize (src/quant.rs:156-165)."""
    d = np.ascontiguousarray(data, np.uint8).ravel()
    r = np.asarray(rng, np.float32).copy()
    out = np.empty(d.shape[0], np.float32)
    lib().oracle_dequantize(_p(d, C.c_uint8), d.shape[0], _p(r, C.c_float), _p(out, C.c_float))
    return out


def tga_8bit_data(data, n_mels: int) -> bytes:
    """tga_8bit_data (src/quant.rs:38-64) on a major-row-order interleaved image."""
    x = _f32(data).ravel()
    out = np.empty(26 + x.shape[0], np.uint8)
    lib().oracle_tga_8bit_data(_p(x, C.c_float), x.shape[0], n_mels, _p(out, C.c_uint8))
    return out.tobytes()


def chunk_frames_into_strides(frames, n_mels: int, stride_size: int):
    """chunk_frames_into_strides (src/quant.rs:100-136): [n_mels][width] cut into stride x stride tiles, row-major each."""
    x = _f32(frames).ravel()
    width = x.shape[0] // n_mels
    if stride_size == width:
        return [x]
    img = x.reshape(n_mels, width)
    return [np.ascontiguousarray(img[y:y + stride_size, c:c + stride_size]).ravel()
            for y in range(0, n_mels, stride_size) for c in range(0, width, stride_size)]


def tga_8bit(data, n_mels: int):
    """tga_8bit (src/quant.rs:29-36): one TGA per <= 65535-column chunk."""
    return [tga_8bit_data(c, n_mels) for c in chunk_frames_into_strides(data, n_mels, 65535)]


def parse_tga_8bit(blob: bytes) -> np.ndarray:
    """parse_tga_8bit (src/quant.rs:66-88): skip 18 bytes, read {min,max}, dequantise the rest."""
    if len(blob) < 26:
        raise ValueError("failed to fill whole buffer")
    rng = np.frombuffer(blob[18:26], "<f4")
    return dequantize(np.frombuffer(blob[26:], np.uint8), rng)


# ---- src/vad.rs ------------------------------------------------------------------------------------

def vad_boundaries(image, min_energy=0.98, min_y=11, min_x=5, min_mel=2):
    """vad_boundaries (src/vad.rs:256-340) on a [n_mels, width] image -> (raw mask, smoothed mask), bool arrays of width-2."""
    x = _f32(image)
    h, w = x.shape
    raw = np.zeros(max(0, w - 2), np.uint8)
    sm = np.zeros(max(0, w - 2), np.uint8)
    n = lib().oracle_vad_boundaries(_p(x, C.c_float), h, w, min_mel, min_y, float(min_energy), _p(raw, C.c_uint8), _p(sm, C.c_uint8))
    return raw[:n].astype(bool), sm[:n].astype(bool)


def vad_on(smoothed, n: int) -> bool:
    """vad_on (src/vad.rs:229-254): the run counter is only tested from the second intersected column on, so
    n <= 1 means "at least two intersected columns" and n >= 2 "a run of n consecutive ones"."""
    m = np.ascontiguousarray(smoothed, np.uint8)
    if n <= 1:
        return int(m.sum()) >= 2
    return int(lib().oracle_vad_longest_run(_p(m, C.c_uint8), m.shape[0])) >= n


def voice_activity_stream(frames, min_energy=0.98, min_y=11, min_x=5, min_mel=2):
    """VoiceActivityDetector::add_activity (src/vad.rs:162-205) fed the rows of `frames` ([F, n_mels]) one by one: per frame None
    (fewer than min_x frames so far, :173-175) or the tuple (active, frame_index, leading_active_columns, active_columns,
    window_columns) computed from vad_boundaries over the window of the last min_x frames (:178-188).  The buffer trimming of
    :168-172 never changes that window."""
    x = _f32(frames)
    out = []
    for f in range(x.shape[0]):
        if f + 1 < min_x:
            out.append(None)
            continue
        window = np.ascontiguousarray(x[f + 1 - min_x:f + 1].T) if min_x > 0 else np.zeros((x.shape[1], 0), np.float32)
        sm = vad_boundaries(window, min_energy, min_y, min_x, min_mel)[1]         # empty when height < 3 or width < 3 (:265-267)
        inter = np.nonzero(sm)[0]
        lead = 0
        for c in inter:                                                          # leading_active_columns, :212-222
            if c == lead:
                lead += 1
            elif c > lead:
                break
        out.append((bool(inter.size and inter[0] == 0), f, int(lead), int(inter.size), int(sm.size)))
    return out


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def stream_mel(samples, fft_size=512, hop_size=160, n_mels=80, sampling_rate=16000.0, flush_tail=False) -> np.ndarray:
    """Streaming Spectrogram::add + MelSpectrogram::add (src/stft.rs:48-86, src/rb.rs:86-121); flush_tail pushes
    the last (< hop) samples through add() too, zero-padded as src/stft.rs:55-60 does."""
    x = _f32(samples)
    cap = x.shape[0] // hop_size + 2
    out = np.empty((cap, n_mels), np.float32)
    n = lib().oracle_stream_mel_ex(_p(x, C.c_float), x.shape[0], fft_size, hop_size, n_mels, sampling_rate,
                                   _p(out, C.c_float), cap, int(flush_tail))
    return out[:n].copy()


def fbank_default_config() -> FbankConfig:
    c = FbankConfig()
    lib().oracle_fbank_default_config(C.byref(c))
    return c


def kaldi_mel_filterbank(sample_rate=16000.0, fft_size=512, num_mel_bins=80, low_freq=20.0, high_freq=8000.0):
    out = np.empty((num_mel_bins, fft_size // 2 + 1), np.float64)
    lib().oracle_kaldi_mel_filterbank(sample_rate, fft_size, num_mel_bins, low_freq, high_freq, _p(out, C.c_double))
    return out


def fbank_compute(samples, cfg: FbankConfig | None = None) -> np.ndarray:
    """Fbank::compute (src/fbank.rs:141-236) -> f32 [frames, num_mel_bins]."""
    cfg = cfg or fbank_default_config()
    x = _f32(samples)
    fl = lib().oracle_fbank_frame_length(C.byref(cfg))
    fs = lib().oracle_fbank_frame_shift(C.byref(cfg))
    nf = 0 if x.shape[0] < fl else 1 + (x.shape[0] - fl) // fs
    out = np.zeros((nf, cfg.num_mel_bins), np.float32)
    if nf:
        got = lib().oracle_fbank_compute(C.byref(cfg), _p(x, C.c_float), x.shape[0], _p(out, C.c_float))
        assert got == nf
    return out


def fbank_batch(clips, cfg: FbankConfig | None = None, n_threads=0) -> np.ndarray:
    cfg = cfg or fbank_default_config()
    x = _f32(clips)
    fl = lib().oracle_fbank_frame_length(C.byref(cfg))
    fs = lib().oracle_fbank_frame_shift(C.byref(cfg))
    nf = 0 if x.shape[1] < fl else 1 + (x.shape[1] - fl) // fs
    out = np.zeros((x.shape[0], nf, cfg.num_mel_bins), np.float32)
    if nf and x.shape[0]:
        lib().oracle_fbank_batch(C.byref(cfg), _p(x, C.c_float), x.shape[1], x.shape[1], x.shape[0],
                                 _p(out, C.c_float), n_threads)
    return out


def blm_default_config(**kw) -> BlmConfig:
    c = BlmConfig()
    lib().oracle_blm_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, type(getattr(c, k))(v))
    return c


def blm_compute(samples, cfg: BlmConfig | None = None, f64: bool = True):
    """BatchLogMelSpectrogram::compute (src/mel.rs:299-385) -> (features [n_mels, cols] f32, valid_frames).
    f64=False is the literal f32 restatement, f64=True the same definition evaluated in f64."""
    cfg = cfg or blm_default_config()
    x = _f32(samples).reshape(-1)
    fn = lib().oracle_blm_compute_f64 if f64 else lib().oracle_blm_compute_f32
    valid = C.c_int64(0)
    cols = fn(C.byref(cfg), _p(x, C.c_float), x.shape[0], None, C.byref(valid))
    out = np.zeros((cfg.n_mels, cols), np.float32)
    if x.shape[0]:
        fn(C.byref(cfg), _p(x, C.c_float), x.shape[0], _p(out, C.c_float), C.byref(valid))
    return out, int(valid.value)


def project_power(filters, power) -> np.ndarray:
    """SparseMelFilterbank::from_dense(filters).project_power_f64 / _f32 (src/mel.rs:48-71, 106-146) for [frames, fft_bins] power in
    f64 or f32: per mel row the non-zero weights in ascending bin order, `energy += weight * power[bin]` (f32: `weight as f32`),
    a separate multiply and add -- numpy's elementwise ops are exactly that."""
    p2 = np.asarray(power)
    T = p2.dtype.type
    assert T in (np.float32, np.float64)
    fb = np.asarray(filters, dtype=np.float64)
    p2 = p2.reshape(-1, fb.shape[1])
    out = np.zeros((p2.shape[0], fb.shape[0]), T)
    for m in range(fb.shape[0]):
        e = np.zeros(p2.shape[0], T)
        for k in np.nonzero(fb[m])[0]:
            e = e + T(fb[m, k]) * p2[:, k]
        out[:, m] = e
    return out


def log_mel_spectrogram(stft, filters) -> np.ndarray:
    """log_mel_spectrogram(stft, mel_filters) (src/mel.rs:436-441) = project_stft_log10 (:148-168) for [frames, n_fft] complex128
    frames: norm_sqr of the bins below n_fft / 2, the sparse rows in ascending bin order, log10(max(E, 1e-10)); f64, not normalised."""
    z = np.asarray(stft, dtype=np.complex128)
    z = z.reshape(-1, z.shape[-1])
    fb = np.asarray(filters, dtype=np.float64)
    half = z.shape[1] // 2
    pw = z.real * z.real + z.imag * z.imag
    out = np.zeros((z.shape[0], fb.shape[0]))
    for m in range(fb.shape[0]):
        e = np.zeros(z.shape[0])
        for k in np.nonzero(fb[m])[0]:
            e = e + fb[m, k] * (pw[:, k] if k < half else 0.0)
        out[:, m] = np.log10(np.maximum(e, 1e-10))
    return out


def norm_mel(mel_spec) -> np.ndarray:
    """norm_mel (f64, src/mel.rs:448-454) / norm_mel_vec (f32, :457-469): one maximum over everything given, max(x, mmax - 8), (+ 4) / 4"""
    a = np.asarray(mel_spec)
    T = a.dtype.type
    mmax = T(np.nanmax(a) if a.size and not np.all(np.isnan(a)) else -np.inf) - T(8)
    c = np.where(a > mmax, a, mmax).astype(T)          # x.max(mmax): NaN x -> mmax
    return ((c + T(4)) / T(4)).astype(T)


def compute_mel_spectrogram_with_filters(samples, fft_size, hop_size, filters) -> np.ndarray:
    """compute_all_cpu + MelSpectrogram::add with `filters` in place of new()'s bank (src/stft.rs:119-138, src/mel.rs:13-32): log_mel of every
    frame, norm_mel per frame (norm_mel_slice_f64, :645-654), f32 out."""
    lm = log_mel_spectrogram(compute_all_cpu(samples, fft_size, hop_size), filters)
    mmax = lm.max(axis=1, keepdims=True) - 8.0
    return ((np.maximum(lm, mmax) + 4.0) / 4.0).astype(np.float32)


def blm_normalize(features, valid: int) -> np.ndarray:
    """normalize_per_feature (src/mel.rs:721-749): the reference's f32 left folds on a [n_mels, cols] f32 image (a copy is returned)."""
    a = np.array(features, dtype=np.float32, order="C", copy=True)
    fn = lib().oracle_blm_normalize
    fn.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int64, C.c_int64]
    fn.restype = None
    if a.size:
        fn(_p(a, C.c_float), a.shape[0], int(valid), a.shape[1])
    return a


SYNTH_SEED = 0x4D454C53


def synth_pcm(clip: int, n: int, seed: int = SYNTH_SEED) -> np.ndarray:
    """Hash-noise PCM of SURVEY.md §8(d) (bit-identical twin of the device generator)."""
    out = np.empty(n, np.float32)
    lib().oracle_synth_pcm(seed, clip, n, _p(out, C.c_float))
    return out


def load_wav_f32(path: str) -> np.ndarray:
    """Find the 'data' chunk like src/fbank.rs:324-352 and return the f32le payload."""
    b = open(path, "rb").read()
    pos = 12
    while pos + 8 <= len(b):
        cid = b[pos:pos + 4]
        size = int.from_bytes(b[pos + 4:pos + 8], "little")
        if cid == b"data":
            start = pos + 8
            n = (len(b) - start) // 4
            return np.frombuffer(b, dtype="<f4", count=n, offset=start).astype(np.float32)
        pos += 8 + size + (size & 1)
    raise ValueError("no data chunk")
