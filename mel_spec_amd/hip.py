"""Host-side mirror of the reference's GPU plugin interface, over the C ABI.

Names, argument order and error behaviour follow the reference (wavey-ai/mel-spec v0.4.0):

  HipMelSpectrogram(fft_size, hop_size, sampling_rate, n_mels)   CudaMelSpectrogram::new       src/cuda.rs:39-82
  .compute_mel_spectrogram(samples) -> [frames, n_mels] f32      ::compute_mel_spectrogram     src/cuda.rs:88-101
  Fbank(FbankConfig()).compute(samples) -> [frames, n_mels] f32  Fbank::{new,compute}          src/fbank.rs:94,141
  mel(sr, n_fft, n_mels, f_min, f_max, htk, norm)                mel()                         src/mel.rs:547-589

Errors: HipUnavailable == CudaError::Unavailable (construction problems; tests skip on it,
src/cuda.rs:512-518), HipRuntimeError == CudaError::Runtime (per-call problems).
All compute goes through libmelspec_hip.so; there is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import BlmConfigC, FbankConfigC, lib


class HipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class HipUnavailable(HipError):
    """CudaError::Unavailable (src/cuda.rs:10-25)."""


class HipRuntimeError(HipError):
    """CudaError::Runtime (src/cuda.rs:10-25)."""


def _check(rc: int, construct: bool = False) -> None:
    if rc == 0:
        return
    msg = _lib.last_error()
    if construct or rc == _lib.ERR_UNAVAILABLE:
        raise HipUnavailable(rc, msg)
    raise HipRuntimeError(rc, msg)


def device_count() -> int:
    """Usable gfx950 devices (0 if none / runtime unavailable)."""
    n = lib().melspec_device_count()
    return n if n > 0 else 0


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class DeviceBuffer:
    """A raw HBM allocation made through the C ABI (melspec_malloc)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(lib().melspec_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, a: np.ndarray, offset_bytes: int = 0) -> None:
        a = np.ascontiguousarray(a)
        assert offset_bytes + a.nbytes <= self.nbytes
        _check(lib().melspec_memcpy_h2d(C.c_void_p(self.ptr + offset_bytes), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, shape, dtype=np.float32, offset_bytes: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        _check(lib().melspec_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset_bytes), out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr:
            lib().melspec_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuffer:
    """Pinned host memory (melspec_host_alloc) as a numpy f32 array: the host pipeline DMA's it in place."""

    def __init__(self, n_floats: int):
        p = C.c_void_p()
        _check(lib().melspec_host_alloc(C.byref(p), max(int(n_floats), 1) * 4))
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(max(int(n_floats), 1),))[:int(n_floats)]

    def free(self) -> None:
        if self._p is not None and self._p.value:
            self.array = None
            lib().melspec_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synth_pcm_device(buf_ptr: int, clip_stride: int, clip_len: int, first_clip: int, n_clips: int,
                     seed: int = 0x4D454C53, stream: int = 0) -> None:
    _check(lib().melspec_synth_pcm_device(C.c_void_p(buf_ptr), clip_stride, clip_len, first_clip, n_clips, seed,
                                          C.c_void_p(stream)))


def synth_pcm_window(buf_ptr: int, clip_stride: int, n_samples: int, first_sample: int, n_clips: int, first_clip: int = 0,
                     seed: int = 0x4D454C53, stream: int = 0) -> None:
    """Samples [first_sample, first_sample + n_samples) of the synthetic clips, clip c at buf + c*clip_stride floats."""
    _check(lib().melspec_synth_pcm_window_device(C.c_void_p(buf_ptr), clip_stride, first_sample, n_samples, first_clip, n_clips, seed,
                                                 C.c_void_p(stream)))


def device_synchronize() -> None:
    _check(lib().melspec_device_synchronize())


class HipMelSpectrogram:
    """MI355X twin of CudaMelSpectrogram (src/cuda.rs:27-140)."""

    def __init__(self, fft_size: int, hop_size: int, sampling_rate: float, n_mels: int, device: int = -1, filterbank=None):
        """filterbank: None = MelSpectrogram::new's mel(sr, fft, n_mels, None, None, false, true); a dict(f_min=, f_max=, htk=, norm=)
        = SparseMelFilterbank::from_mel (src/mel.rs:73-87); a [n_mels, fft_size // 2 + 1] array = from_dense (src/mel.rs:48-71)."""
        self._h = None
        h = C.c_void_p()
        if filterbank is None:
            rc = lib().melspec_create(C.byref(h), device, int(fft_size), int(hop_size), float(sampling_rate), int(n_mels))
        elif isinstance(filterbank, dict):
            fmin, fmax = filterbank.get("f_min"), filterbank.get("f_max")
            rc = lib().melspec_create_with_filterbank(C.byref(h), device, int(fft_size), int(hop_size), float(sampling_rate), int(n_mels),
                                                      -1.0 if fmin is None else float(fmin), -1.0 if fmax is None else float(fmax),
                                                      int(filterbank.get("htk", False)), int(filterbank.get("norm", True)))
        else:
            fb = np.ascontiguousarray(filterbank, dtype=np.float64)
            assert fb.ndim == 2 and fb.shape[0] == n_mels, fb.shape
            rc = lib().melspec_create_with_dense_filterbank(C.byref(h), device, int(fft_size), int(hop_size), float(sampling_rate), int(n_mels),
                                                            fb.ctypes.data_as(C.POINTER(C.c_double)), fb.shape[1])
        _check(rc, construct=True)
        self._h = h
        self.fft_size, self.hop_size, self.n_mels = int(fft_size), int(hop_size), int(n_mels)
        self.sampling_rate = float(sampling_rate)
        # MELSPEC_PRECISE=1 / =f32: initial precision mode of every context made through this mirror -- how the GPU test suite is run
        # a second time in f64 mode.  A switch of the test mirror: libmelspec_hip.so itself reads no environment variable.
        env = os.environ.get("MELSPEC_PRECISE", "")[:1]
        if env in ("1", "f"):
            self.set_precision("f64" if env == "1" else "f32")

    # -- reference surface --------------------------------------------------------------
    def compute_mel_spectrogram(self, samples) -> np.ndarray:
        """&[f32] -> [frames][n_mels] f32 (Vec<Vec<f32>> in the reference)."""
        x = _f32(samples).reshape(-1)
        nf = self.num_frames(x.shape[0])
        out = np.empty((nf, self.n_mels), np.float32)
        got = C.c_size_t(0)
        _check(lib().melspec_compute_host(self._h, _fp(x), x.shape[0], _fp(out), out.size, C.byref(got)))
        assert got.value == nf
        return out

    # -- additive surface ---------------------------------------------------------------
    def num_frames(self, n_samples: int) -> int:
        return int(lib().melspec_num_frames(self._h, n_samples))

    @property
    def uses_fast_path(self) -> bool:
        return bool(lib().melspec_uses_fast_path(self._h))

    PRECISION = {"auto": 0, "f64": 1, "f32": 2}

    def set_precision(self, mode: str) -> None:
        """melspec_set_precision: "auto" (default: f32 FFT, the frames its error bound does not cover recomputed in f64),
        "f64" (every frame), "f32" (no guard)."""
        _check(lib().melspec_set_precision(self._h, self.PRECISION[mode]))

    @property
    def precision(self) -> str:
        v = int(lib().melspec_precision(self._h))
        return next(k for k, c in self.PRECISION.items() if c == v)

    def set_precise(self, on: bool = True) -> None:
        """melspec_set_precise: on -> "f64", off -> "auto"."""
        _check(lib().melspec_set_precise(self._h, int(on)))

    @property
    def precise(self) -> bool:
        return bool(lib().melspec_is_precise(self._h))

    def stft_bins(self, full: bool = False) -> int:
        return int(lib().melspec_stft_bins(self._h, int(full)))

    def compute_all(self, samples, dtype=np.complex128, full: bool = True) -> np.ndarray:
        """Spectrogram::compute_all_cpu on the GPU (melspec_stft_host): [frames][bins] complex64 / complex128; full = the
        reference's n_fft-bin layout, else the n_fft/2 + 1 non-redundant bins."""
        x = _f32(samples).reshape(-1)
        dt = np.dtype(dtype)
        assert dt in (np.dtype(np.complex64), np.dtype(np.complex128))
        nf, bins = self.num_frames(x.shape[0]), self.stft_bins(full)
        out = np.zeros((nf, bins), dt)
        got = C.c_size_t(0)
        _check(lib().melspec_stft_host(self._h, _fp(x), x.shape[0], out.ctypes.data_as(C.c_void_p), out.size, int(dt == np.dtype(np.complex128)),
                                       int(full), C.byref(got)))
        assert got.value == nf
        return out

    def mel_from_stft(self, spec) -> np.ndarray:
        """MelSpectrogram::add(&fft) for every frame of `spec` ([frames][n_fft] or [frames][n_fft/2 + 1] complex64 / complex128, e.g. what
        compute_all returns): project_stft_log10 + norm_mel in f64 on the GPU (melspec_mel_from_stft_host) -> [frames][n_mels] f32."""
        a = np.ascontiguousarray(spec)
        if a.dtype not in (np.dtype(np.complex64), np.dtype(np.complex128)):
            a = a.astype(np.complex128)
        assert a.ndim == 2 and a.shape[1] in (self.stft_bins(True), self.stft_bins(False)), a.shape
        full = a.shape[1] == self.stft_bins(True)
        out = np.empty((a.shape[0], self.n_mels), np.float32)
        _check(lib().melspec_mel_from_stft_host(self._h, a.ctypes.data_as(C.c_void_p), int(a.dtype == np.dtype(np.complex128)), int(full), a.shape[0],
                                                _fp(out.reshape(-1)), out.size))
        return out

    def mel_from_stft_device(self, d_spec: int, n_frames: int, d_out: int, f64: bool = False, full: bool = False, stream: int = 0) -> None:
        _check(lib().melspec_mel_from_stft_device(self._h, C.c_void_p(d_spec), int(f64), int(full), n_frames, C.c_void_p(d_out), C.c_void_p(stream)))

    def stft_uniform_device(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_out: int, f64: bool = False, full: bool = False,
                            stream: int = 0) -> None:
        _check(lib().melspec_stft_uniform_device(self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips, C.c_void_p(d_out), int(f64), int(full),
                                                 C.c_void_p(stream)))

    def stft_ragged_device(self, d_pcm: int, offsets, lengths, d_out: int, out_offsets=None, f64: bool = False, full: bool = False, stream: int = 0) -> None:
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        _check(lib().melspec_stft_ragged_device(self._h, C.c_void_p(d_pcm), off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p), off.shape[0],
                                                C.c_void_p(d_out), None if oo is None else oo.ctypes.data_as(u64p), int(f64), int(full), C.c_void_p(stream)))

    def release_scratch(self) -> None:
        _check(lib().melspec_release_scratch(self._h))

    def plain_kernel_name(self) -> str:
        return (lib().melspec_plain_kernel_name(self._h) or b"").decode()

    def guard_count(self) -> int:
        """frames that tripped the guard of the "auto" mode since the context was created (synchronises the device)"""
        n = C.c_uint64(0)
        _check(lib().melspec_guard_count(self._h, C.byref(n)))
        return int(n.value)

    def set_auto_adaptive(self, on: bool = True) -> None:
        """melspec_set_auto_adaptive: "auto" lets a vote inside each batch's launch send the whole batch to the f64 kernel when more than 1/8 of the
        sampled frames trip the guard (default on); off: the f32 kernel + per-frame recompute whatever the input"""
        _check(lib().melspec_set_auto_adaptive(self._h, int(on)))

    def auto_state(self):
        """melspec_auto_state -> (the last finished "auto" batch ran on the f64 kernel, fraction of its frames that tripped the guard)"""
        h, f = C.c_int(0), C.c_double(0.0)
        _check(lib().melspec_auto_state(self._h, C.byref(h), C.byref(f)))
        return bool(h.value), float(f.value)

    def guard_last_count(self) -> int:
        """the same since the previous call of this method"""
        total = self.guard_count()
        last = total - getattr(self, "_guard_seen", 0)
        self._guard_seen = total
        return last

    def compute_uniform_device(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_out: int,
                               stream: int = 0) -> None:
        """Device pointers in, device pointers out, asynchronous on `stream`."""
        _check(lib().melspec_compute_uniform_device(self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips,
                                                    C.c_void_p(d_out), C.c_void_p(stream)))

    def interleaved_width(self, n_samples: int, min_width: int = 0) -> int:
        return int(lib().melspec_interleaved_width(self._h, n_samples, min_width))

    def compute_uniform_device_interleaved(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_out: int,
                                           major_column_order: bool = False, min_width: int = 0, stream: int = 0) -> None:
        """interleave_frames (src/mel.rs:480-544) fused into the store; see melspec_hip.h."""
        _check(lib().melspec_compute_uniform_device_interleaved(
            self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips, C.c_void_p(d_out), int(major_column_order),
            min_width, C.c_void_p(stream)))

    def compute_batch_interleaved(self, clips, major_column_order: bool = False, min_width: int = 0) -> np.ndarray:
        """[n_clips, clip_len] -> [n_clips, n_mels, W] (or [n_clips, W, n_mels]) in one launch."""
        x = _f32(clips)
        n_clips, clip_len = x.shape
        W = self.interleaved_width(clip_len, min_width)
        shape = (n_clips, W, self.n_mels) if major_column_order else (n_clips, self.n_mels, W)
        din, dout = DeviceBuffer(x.nbytes), DeviceBuffer(max(16, n_clips * W * self.n_mels * 4))
        try:
            din.upload(x)
            self.compute_uniform_device_interleaved(din.ptr, clip_len, clip_len, n_clips, dout.ptr, major_column_order, min_width)
            self.synchronize()
            return dout.download(shape)
        finally:
            din.free(); dout.free()

    def compute_ragged_device(self, d_pcm: int, offsets, lengths, d_out: int, out_offsets=None, stream: int = 0) -> None:
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        _check(lib().melspec_compute_ragged_device(
            self._h, C.c_void_p(d_pcm), off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p), off.shape[0],
            C.c_void_p(d_out), None if oo is None else oo.ctypes.data_as(u64p), C.c_void_p(stream)))

    def compute_ragged_device_desc(self, d_pcm: int, d_offsets: int, d_lengths: int, n_clips: int, d_out: int, d_out_offsets: int,
                                   max_total_frames: int, stream: int = 0) -> None:
        """melspec_compute_ragged_device_desc: the clip table (u64 offsets / lengths / output offsets) lives in device memory"""
        _check(lib().melspec_compute_ragged_device_desc(self._h, C.c_void_p(d_pcm), C.c_void_p(d_offsets), C.c_void_p(d_lengths), n_clips,
                                                        C.c_void_p(d_out), C.c_void_p(d_out_offsets) if d_out_offsets else None,
                                                        max_total_frames, C.c_void_p(stream)))

    def synchronize(self, stream: int = 0) -> None:
        _check(lib().melspec_synchronize(self._h, C.c_void_p(stream)))

    def time_uniform_device(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_out: int,
                            warmup: int = 3, iters: int = 20) -> float:
        """Average milliseconds per launch, HIP events on the context's stream."""
        ms = C.c_float(0.0)
        _check(lib().melspec_time_uniform_device(self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips,
                                                 C.c_void_p(d_out), warmup, iters, C.byref(ms)))
        return float(ms.value)

    def time_first_kernel(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_out: int, warmup: int = 10, iters: int = 100) -> float:
        """melspec_time_first_kernel: average launch duration (ms) of the first kernel of a call -- the f32 kernel, without "auto"'s gated
        f64 launch behind it -- from a HIP event pair around it in every call"""
        ms = C.c_float(0.0)
        _check(lib().melspec_time_first_kernel(self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips, C.c_void_p(d_out), warmup, iters, C.byref(ms)))
        return float(ms.value)

    def compute_batch_host(self, flat: np.ndarray, offsets, lengths, out: np.ndarray | None = None, out_offsets=None):
        """melspec_compute_batch_host: clip i = flat[offsets[i] : offsets[i] + lengths[i]] -> its frames at out[out_offsets[i]:]
        (floats; None = packed).  Returns (out, total_frames).  flat / out may be pinned (HostBuffer.array)."""
        x = flat if isinstance(flat, np.ndarray) and flat.dtype == np.float32 and flat.flags.c_contiguous else _f32(flat).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        frames = np.array([self.num_frames(int(n)) for n in ln], dtype=np.uint64)
        if out is None:
            out = np.empty(int(frames.sum()) * self.n_mels, np.float32)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        total = C.c_uint64(0)
        _check(lib().melspec_compute_batch_host(self._h, _fp(x.reshape(-1)), off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p), off.shape[0],
                                                _fp(out.reshape(-1)), None if oo is None else oo.ctypes.data_as(u64p), out.size, C.byref(total)))
        return out, int(total.value)

    def compute_batch(self, clips) -> np.ndarray:
        """[n_clips, clip_len] host f32 -> [n_clips, frames, n_mels] host f32 through the chunked host pipeline."""
        x = _f32(clips)
        assert x.ndim == 2
        n_clips, clip_len = x.shape
        nf = self.num_frames(clip_len)
        out = np.empty((n_clips, nf, self.n_mels), np.float32)
        if out.size == 0:
            return out
        offs = np.arange(n_clips, dtype=np.uint64) * np.uint64(clip_len)
        self.compute_batch_host(x.reshape(-1), offs, np.full(n_clips, clip_len, np.uint64), out.reshape(-1))
        return out

    def compute_ragged(self, clips) -> list:
        """List of 1-D host arrays of any lengths -> list of [frames_i, n_mels] arrays, one call of the host pipeline."""
        arrs = [_f32(c).reshape(-1) for c in clips]
        lens = np.array([a.shape[0] for a in arrs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(arrs) else np.zeros(0, np.uint64)
        frames = [self.num_frames(int(n)) for n in lens]
        flat = np.concatenate(arrs) if arrs and int(lens.sum()) else np.zeros(1, np.float32)
        out, _ = self.compute_batch_host(flat, offs, lens)
        res, cur = [], 0
        for f in frames:
            res.append(out[cur:cur + f * self.n_mels].reshape(f, self.n_mels))
            cur += f * self.n_mels
        return res

    def close(self) -> None:
        if self._h is not None:
            lib().melspec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class FbankConfig:
    """FbankConfig::default (src/fbank.rs:46-64)."""
    sample_rate: float = 16000.0
    num_mel_bins: int = 80
    frame_length_ms: float = 25.0
    frame_shift_ms: float = 10.0
    energy_floor: float = 0.0
    use_log_fbank: bool = True
    use_power: bool = True
    preemphasis: float = 0.97
    apply_cmn: bool = True
    low_freq: float = 20.0
    high_freq: float = 0.0

    def to_c(self) -> FbankConfigC:
        return FbankConfigC(self.sample_rate, self.num_mel_bins, self.frame_length_ms, self.frame_shift_ms,
                            self.energy_floor, int(self.use_log_fbank), int(self.use_power), self.preemphasis,
                            int(self.apply_cmn), self.low_freq, self.high_freq)

    def frame_length_samples(self) -> int:   # src/fbank.rs:68-70 (f64::round: half away from zero)
        return int(np.floor((self.frame_length_ms / 1000.0) * self.sample_rate + 0.5))

    def frame_shift_samples(self) -> int:    # src/fbank.rs:73-75
        return int(np.floor((self.frame_shift_ms / 1000.0) * self.sample_rate + 0.5))

    def fft_size(self) -> int:               # src/fbank.rs:78-81
        n, p = self.frame_length_samples(), 1
        while p < n:
            p <<= 1
        return p


class Fbank:
    """MI355X twin of Fbank (src/fbank.rs:85-247)."""

    def __init__(self, config: FbankConfig | None = None, device: int = -1):
        self._h = None
        self.config = config or FbankConfig()
        h = C.c_void_p()
        cc = self.config.to_c()
        _check(lib().melspec_fbank_create(C.byref(h), device, C.byref(cc)), construct=True)
        self._h = h
        self.num_mel_bins = self.config.num_mel_bins

    def num_frames(self, n_samples: int) -> int:
        return int(lib().melspec_fbank_num_frames(self._h, n_samples))

    def dense_filterbank(self) -> np.ndarray:
        """dense_filterbank (src/fbank.rs:244-246): the Kaldi weights [num_mel_bins, fft_size/2 + 1] the projection is built from
        (high_freq == 0 means Nyquist, :108-112)."""
        c = self.config
        return kaldi_mel_filterbank(c.sample_rate, c.fft_size(), c.num_mel_bins, c.low_freq, c.high_freq if c.high_freq != 0.0 else c.sample_rate / 2.0)

    @property
    def uses_fast_path(self) -> bool:
        return bool(lib().melspec_fbank_uses_fast_path(self._h))

    def compute_ragged_device(self, d_pcm: int, offsets, lengths, d_out: int, out_offsets=None, stream: int = 0) -> None:
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        _check(lib().melspec_fbank_compute_ragged_device(self._h, C.c_void_p(d_pcm), off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p), off.shape[0],
                                                         C.c_void_p(d_out), None if oo is None else oo.ctypes.data_as(u64p), C.c_void_p(stream)))

    def compute_ragged_device_desc(self, d_pcm: int, d_offsets: int, d_lengths: int, n_clips: int, d_out: int, d_out_offsets: int,
                                   max_total_frames: int, stream: int = 0) -> None:
        _check(lib().melspec_fbank_compute_ragged_device_desc(self._h, C.c_void_p(d_pcm), C.c_void_p(d_offsets), C.c_void_p(d_lengths), n_clips,
                                                              C.c_void_p(d_out), C.c_void_p(d_out_offsets) if d_out_offsets else None,
                                                              max_total_frames, C.c_void_p(stream)))

    def compute_ragged(self, clips) -> list:
        """list of 1-D host arrays of any lengths -> list of [frames_i, num_mel_bins] arrays (Fbank::compute per clip), one launch"""
        arrs = [_f32(c).reshape(-1) for c in clips]
        lens = np.array([a.shape[0] for a in arrs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(arrs) else np.zeros(0, np.uint64)
        frames = [self.num_frames(int(n)) for n in lens]
        total = sum(frames) * self.num_mel_bins
        flat = np.concatenate(arrs) if arrs and int(lens.sum()) else np.zeros(1, np.float32)
        din, dout = DeviceBuffer(max(flat.nbytes, 16)), DeviceBuffer(max(total * 4, 16))
        try:
            din.upload(flat)
            self.compute_ragged_device(din.ptr, offs, lens, dout.ptr)
            self.synchronize()
            out = dout.download((max(total, 1),))[:total]
        finally:
            din.free(); dout.free()
        res, cur = [], 0
        for f in frames:
            res.append(out[cur:cur + f * self.num_mel_bins].reshape(f, self.num_mel_bins))
            cur += f * self.num_mel_bins
        return res

    def compute_batch_host(self, flat: np.ndarray, offsets, lengths, out: np.ndarray | None = None, out_offsets=None):
        """melspec_fbank_compute_batch_host: clip i = flat[offsets[i] : + lengths[i]] -> its [frames_i, num_mel_bins] rows at
        out[out_offsets[i]:] (floats; None = packed), whole clips through the chunked host pipeline.  Returns (out, total_frames)."""
        x = flat if isinstance(flat, np.ndarray) and flat.dtype == np.float32 and flat.flags.c_contiguous else _f32(flat).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        if out is None:
            out = np.empty(max(sum(self.num_frames(int(n)) for n in ln) * self.num_mel_bins, 1), np.float32)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        total = C.c_uint64(0)
        _check(lib().melspec_fbank_compute_batch_host(self._h, _fp(x.reshape(-1)), off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p), off.shape[0],
                                                      _fp(out.reshape(-1)), None if oo is None else oo.ctypes.data_as(u64p), out.size, C.byref(total)))
        return out, int(total.value)

    def compute_many(self, clips) -> list:
        """list of 1-D host arrays -> list of [frames_i, num_mel_bins] arrays (Fbank::compute per clip) through the host pipeline"""
        arrs = [_f32(c).reshape(-1) for c in clips]
        lens = np.array([a.shape[0] for a in arrs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(arrs) else np.zeros(0, np.uint64)
        flat = np.concatenate(arrs) if arrs and int(lens.sum()) else np.zeros(1, np.float32)
        out, _ = self.compute_batch_host(flat, offs, lens)
        res, cur = [], 0
        for n in lens:
            f = self.num_frames(int(n))
            res.append(out[cur:cur + f * self.num_mel_bins].reshape(f, self.num_mel_bins))
            cur += f * self.num_mel_bins
        return res

    def use_generic(self, on: bool = True) -> None:
        """run on the generic f64 direct-DFT kernel (the on-device cross-check of the fused kernel)"""
        _check(lib().melspec_fbank_use_generic(self._h, int(on)))      # 2: the workgroup-per-frame kernel whatever the geometry

    def compute(self, samples) -> np.ndarray:
        """&[f32] -> Array2<f32> (num_frames, num_mel_bins)."""
        x = _f32(samples).reshape(-1)
        nf = self.num_frames(x.shape[0])
        out = np.zeros((nf, self.num_mel_bins), np.float32)
        got = C.c_size_t(0)
        _check(lib().melspec_fbank_compute_host(self._h, _fp(x), x.shape[0], _fp(out), out.size, C.byref(got)))
        assert got.value == nf
        return out

    def compute_uniform_device(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_out: int,
                               stream: int = 0) -> None:
        _check(lib().melspec_fbank_compute_uniform_device(self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips,
                                                          C.c_void_p(d_out), C.c_void_p(stream)))

    def compute_uniform_device_split(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_rows: int, d_means: int,
                                     stream: int = 0) -> None:
        """melspec_fbank_compute_uniform_device_split (additive): the rows before CMN + the [clip][num_mel_bins] means CMN subtracts"""
        _check(lib().melspec_fbank_compute_uniform_device_split(self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips,
                                                                C.c_void_p(d_rows), C.c_void_p(d_means), C.c_void_p(stream)))

    def synchronize(self, stream: int = 0) -> None:
        _check(lib().melspec_fbank_synchronize(self._h, C.c_void_p(stream)))

    def release_scratch(self) -> None:
        """give the object's grow-only scratch back (pipeline / staging buffers, ragged plans); the next call re-allocates"""
        _check(lib().melspec_fbank_release_scratch(self._h))

    def compute_batch(self, clips) -> np.ndarray:
        x = _f32(clips)
        n_clips, clip_len = x.shape
        nf = self.num_frames(clip_len)
        out = np.zeros((n_clips, nf, self.num_mel_bins), np.float32)
        if out.size == 0:
            return out
        din, dout = DeviceBuffer(x.nbytes), DeviceBuffer(out.nbytes)
        try:
            din.upload(x)
            self.compute_uniform_device(din.ptr, clip_len, clip_len, n_clips, dout.ptr)
            self.synchronize()
            out = dout.download(out.shape)
        finally:
            din.free(); dout.free()
        return out

    def close(self) -> None:
        if self._h is not None:
            lib().melspec_fbank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchLogMelError(HipError):
    """BatchLogMelError::InvalidConfig (src/mel.rs:210-225)."""


@dataclass
class BatchLogMelConfig:
    """BatchLogMelConfig::default (src/mel.rs:189-208)."""
    sample_rate: int = 16000
    n_fft: int = 512
    win_length: int = 400
    hop_length: int = 160
    n_mels: int = 80
    f_min: float = 0.0
    f_max: float | None = None
    htk: bool = False
    norm: bool = True
    preemphasis: float = 0.0
    center: bool = True
    log_zero_guard: float = float(np.finfo(np.float32).eps)
    pad_to: int = 0
    normalize_per_feature: bool = False

    def to_c(self) -> BlmConfigC:
        return BlmConfigC(self.sample_rate, self.n_fft, self.win_length, self.hop_length, self.n_mels, self.f_min,
                          -1.0 if self.f_max is None else self.f_max, int(self.htk), int(self.norm), self.preemphasis,
                          int(self.center), self.log_zero_guard, self.pad_to, int(self.normalize_per_feature))


class BatchLogMelSpectrogram:
    """MI355X twin of BatchLogMelSpectrogram (src/mel.rs:239-396): NeMo/Parakeet-style frontend."""

    def __init__(self, config: BatchLogMelConfig | None = None, device: int = -1):
        self._h = None
        self.config = config or BatchLogMelConfig()
        h = C.c_void_p()
        cc = self.config.to_c()
        rc = lib().melspec_blm_create(C.byref(h), device, C.byref(cc))
        if rc == _lib.ERR_INVALID_ARG:
            raise BatchLogMelError(rc, _lib.last_error())
        _check(rc, construct=True)
        self._h = h

    def set_precision(self, mode: str) -> None:
        """melspec_blm_set_precision: "f32" = the reference's own arithmetic type for this frontend (src/mel.rs:251-252,356-357) on the
        f32 kernel; "auto" (default) / "f64" = f64 up to |X|^2, within 1e-4 of the f64 evaluation of the definition on every input."""
        _check(lib().melspec_blm_set_precision(self._h, HipMelSpectrogram.PRECISION[mode]))

    @property
    def precision(self) -> str:
        """what the next call computes in: 'f32' or 'f64'"""
        return "f32" if int(lib().melspec_blm_precision(self._h)) == 2 else "f64"

    def num_frames(self, n_samples: int) -> int:
        return int(lib().melspec_blm_num_frames(self._h, n_samples))

    def padded_frames(self, n_samples: int) -> int:
        return int(lib().melspec_blm_padded_frames(self._h, n_samples))

    def filters(self, device: int = -1) -> "SparseMelFilterbank":
        """filters (src/mel.rs:286-288): the sparse bank BatchLogMelSpectrogram::new builds (:254-263)."""
        c = self.config
        return SparseMelFilterbank.from_mel(float(c.sample_rate), c.n_fft, c.n_mels, c.f_min, c.sample_rate / 2.0 if c.f_max is None else c.f_max,
                                            c.htk, c.norm, device)

    def compute_flat(self, samples):
        """compute_flat (src/mel.rs:304-307) -> BatchLogMelOutput as (data, rows, cols): the feature-major values, flat."""
        a = self.compute(samples)
        return a.reshape(-1), a.shape[0], a.shape[1]

    def compute(self, samples) -> np.ndarray:
        """&[f32] -> Array2<f32> (n_mels, cols), feature-major (src/mel.rs:299-302)."""
        x = _f32(samples).reshape(-1)
        cols = self.padded_frames(x.shape[0])
        out = np.zeros((self.config.n_mels, cols), np.float32)
        r, c = C.c_size_t(0), C.c_size_t(0)
        _check(lib().melspec_blm_compute_host(self._h, _fp(x), x.shape[0], _fp(out), out.size, C.byref(r), C.byref(c)))
        assert r.value == self.config.n_mels and c.value == cols
        return out

    def compute_uniform_device(self, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, d_out: int, stream: int = 0) -> None:
        _check(lib().melspec_blm_compute_uniform_device(self._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips,
                                                        C.c_void_p(d_out), C.c_void_p(stream)))

    def synchronize(self, stream: int = 0) -> None:
        _check(lib().melspec_blm_synchronize(self._h, C.c_void_p(stream)))

    def release_scratch(self) -> None:
        _check(lib().melspec_blm_release_scratch(self._h))

    def compute_batch_host(self, flat: np.ndarray, offsets, lengths, out: np.ndarray | None = None, out_offsets=None):
        """melspec_blm_compute_batch_host: clip i -> [n_mels, cols_i] floats at out[out_offsets[i]:] (None = packed), whole clips
        through the chunked host pipeline.  Returns (out, total_columns)."""
        x = flat if isinstance(flat, np.ndarray) and flat.dtype == np.float32 and flat.flags.c_contiguous else _f32(flat).reshape(-1)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        if out is None:
            out = np.empty(max(sum(self.padded_frames(int(n)) for n in ln) * self.config.n_mels, 1), np.float32)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, dtype=np.uint64)
        u64p = C.POINTER(C.c_uint64)
        total = C.c_uint64(0)
        _check(lib().melspec_blm_compute_batch_host(self._h, _fp(x.reshape(-1)), off.ctypes.data_as(u64p), ln.ctypes.data_as(u64p), off.shape[0],
                                                    _fp(out.reshape(-1)), None if oo is None else oo.ctypes.data_as(u64p), out.size, C.byref(total)))
        return out, int(total.value)

    def compute_many(self, clips) -> list:
        """list of 1-D host arrays -> list of [n_mels, cols_i] arrays (BatchLogMelSpectrogram::compute per clip) through the host pipeline"""
        arrs = [_f32(c).reshape(-1) for c in clips]
        lens = np.array([a.shape[0] for a in arrs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(arrs) else np.zeros(0, np.uint64)
        flat = np.concatenate(arrs) if arrs and int(lens.sum()) else np.zeros(1, np.float32)
        out, _ = self.compute_batch_host(flat, offs, lens)
        nm = self.config.n_mels
        res, cur = [], 0
        for n in lens:
            c = self.padded_frames(int(n))
            res.append(out[cur:cur + c * nm].reshape(nm, c))
            cur += c * nm
        return res

    def compute_ragged(self, clips) -> list:
        """list of 1-D host arrays of any lengths -> list of [n_mels, cols_i] arrays (BatchLogMelSpectrogram::compute per clip), one launch"""
        arrs = [_f32(c).reshape(-1) for c in clips]
        lens = np.array([a.shape[0] for a in arrs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64) if len(arrs) else np.zeros(0, np.uint64)
        cols = [self.padded_frames(int(n)) for n in lens]
        nm = self.config.n_mels
        total = sum(cols) * nm
        flat = np.concatenate(arrs) if arrs and int(lens.sum()) else np.zeros(1, np.float32)
        din, dout = DeviceBuffer(max(flat.nbytes, 16)), DeviceBuffer(max(total * 4, 16))
        u64p = C.POINTER(C.c_uint64)
        try:
            din.upload(flat)
            _check(lib().melspec_blm_compute_ragged_device(self._h, C.c_void_p(din.ptr), offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), len(arrs),
                                                           C.c_void_p(dout.ptr), None, None))
            self.synchronize()
            out = dout.download((max(total, 1),))[:total]
        finally:
            din.free(); dout.free()
        res, cur = [], 0
        for c in cols:
            res.append(out[cur:cur + c * nm].reshape(nm, c))
            cur += c * nm
        return res

    def compute_batch(self, clips) -> np.ndarray:
        x = _f32(clips)
        n_clips, clip_len = x.shape
        cols = self.padded_frames(clip_len)
        shape = (n_clips, self.config.n_mels, cols)
        if cols == 0:
            return np.zeros(shape, np.float32)
        din, dout = DeviceBuffer(x.nbytes), DeviceBuffer(int(np.prod(shape)) * 4)
        try:
            din.upload(x)
            self.compute_uniform_device(din.ptr, clip_len, clip_len, n_clips, dout.ptr)
            self.synchronize()
            return dout.download(shape)
        finally:
            din.free(); dout.free()

    def close(self) -> None:
        if self._h is not None:
            lib().melspec_blm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseMelFilterbank:
    """SparseMelFilterbank (src/mel.rs:40-168) + log_mel_spectrogram / norm_mel / norm_mel_vec (:436-469) on the GPU (melspec_bank_*)."""

    def __init__(self, h):
        self._h = h

    @classmethod
    def from_dense(cls, filters, device: int = -1):
        fb = np.ascontiguousarray(filters, dtype=np.float64)
        assert fb.ndim == 2
        h = C.c_void_p()
        _check(lib().melspec_bank_from_dense(C.byref(h), device, fb.ctypes.data_as(C.POINTER(C.c_double)), fb.shape[0], fb.shape[1]), construct=True)
        return cls(h)

    @classmethod
    def from_mel(cls, sample_rate: float, n_fft: int, n_mels: int, f_min=None, f_max=None, htk: bool = False, norm: bool = True, device: int = -1):
        h = C.c_void_p()
        _check(lib().melspec_bank_from_mel(C.byref(h), device, float(sample_rate), int(n_fft), int(n_mels), -1.0 if f_min is None else float(f_min),
                                           -1.0 if f_max is None else float(f_max), int(htk), int(norm)), construct=True)
        return cls(h)

    n_mels = property(lambda self: int(lib().melspec_bank_n_mels(self._h)))
    fft_bins = property(lambda self: int(lib().melspec_bank_fft_bins(self._h)))
    non_zero_weights = property(lambda self: int(lib().melspec_bank_non_zero_weights(self._h)))

    def dense_weights(self) -> int:
        return self.n_mels * self.fft_bins

    def weights_for_mel(self, mel_idx: int):
        """weights_for_mel (src/mel.rs:102-104): [(bin, weight), ...] of one row, ascending bins"""
        n = int(lib().melspec_bank_weights_for_mel(self._h, int(mel_idx), None, None, 0))
        if n < 0:
            raise IndexError(mel_idx)
        bins, w = np.zeros(n, np.int32), np.zeros(n, np.float64)
        lib().melspec_bank_weights_for_mel(self._h, int(mel_idx), bins.ctypes.data_as(C.POINTER(C.c_int)), w.ctypes.data_as(C.POINTER(C.c_double)), n)
        return list(zip(bins.tolist(), w.tolist()))

    @staticmethod
    def _dt(a):
        if a.dtype not in (np.dtype(np.float32), np.dtype(np.float64), np.dtype(np.complex64), np.dtype(np.complex128)):
            a = a.astype(np.float64)
        return np.ascontiguousarray(a), int(a.dtype in (np.dtype(np.float64), np.dtype(np.complex128)))

    def project_power(self, power) -> np.ndarray:
        """project_power_f64 / project_power_f32 by the dtype of `power` ([fft_bins] or [frames, fft_bins])"""
        a, f64 = self._dt(np.asarray(power))
        one = a.ndim == 1
        a2 = a.reshape(-1, self.fft_bins)
        out = np.empty((a2.shape[0], self.n_mels), a.dtype)
        _check(lib().melspec_bank_project_power_host(self._h, a2.ctypes.data_as(C.c_void_p), f64, a2.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out[0] if one else out

    def log_mel_spectrogram(self, stft) -> np.ndarray:
        """log_mel_spectrogram(stft, mel_filters) for [n_fft] or [frames, n_fft] complex frames -> [frames, n_mels] f64 (the reference
        returns the [n_mels, 1] column of one frame)"""
        a, f64 = self._dt(np.asarray(stft))
        assert a.dtype in (np.dtype(np.complex64), np.dtype(np.complex128))
        one = a.ndim == 1
        a2 = a.reshape(-1, a.shape[-1])
        out = np.empty((a2.shape[0], self.n_mels), np.float64)
        _check(lib().melspec_bank_log_mel_host(self._h, a2.ctypes.data_as(C.c_void_p), f64, a2.shape[1], a2.shape[0], out.ctypes.data_as(C.POINTER(C.c_double))))
        return out[0] if one else out

    def norm_mel(self, mel_spec) -> np.ndarray:
        """norm_mel (f64 input) / norm_mel_vec (f32 input): one maximum over everything given"""
        a, f64 = self._dt(np.asarray(mel_spec))
        out = np.empty_like(a)
        _check(lib().melspec_bank_norm_mel_host(self._h, a.ctypes.data_as(C.c_void_p), f64, a.size, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self) -> None:
        if self._h is not None:
            lib().melspec_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# -- host-side table builders (CPU only; no GPU needed) ------------------------------------

def mel(sr: float, n_fft: int, n_mels: int, f_min=None, f_max=None, htk: bool = False, norm: bool = True) -> np.ndarray:
    """mel() of src/mel.rs:547-589 -> dense f64 [n_mels, n_fft//2+1]."""
    out = np.empty((n_mels, n_fft // 2 + 1), np.float64)
    _check(lib().melspec_mel_filterbank(sr, n_fft, n_mels, -1.0 if f_min is None else f_min,
                                        -1.0 if f_max is None else f_max, int(htk), int(norm),
                                        out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def hz_to_mel(frequency: float, htk: bool = False) -> float:
    """hz_to_mel (src/mel.rs:591-607)."""
    return float(lib().melspec_hz_to_mel(float(frequency), int(htk)))


def mel_to_hz(mel_value: float, htk: bool = False) -> float:
    """mel_to_hz (src/mel.rs:609-625)."""
    return float(lib().melspec_mel_to_hz(float(mel_value), int(htk)))


def mels_to_hz(mels, htk: bool = False) -> np.ndarray:
    """mels_to_hz (src/mel.rs:627-629)."""
    return np.array([mel_to_hz(m, htk) for m in np.asarray(mels, np.float64).ravel()], np.float64)


def mel_frequencies(n_mels: int, fmin: float, fmax: float, htk: bool = False) -> np.ndarray:
    """mel_frequencies (src/mel.rs:631-637)."""
    out = np.empty(int(n_mels), np.float64)
    _check(lib().melspec_mel_frequencies(int(n_mels), float(fmin), float(fmax), int(htk), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def fft_frequencies(sr: float, n_fft: int) -> np.ndarray:
    """fft_frequencies (src/mel.rs:639-643)."""
    out = np.empty(int(n_fft) // 2 + 1, np.float64)
    _check(lib().melspec_fft_frequencies(float(sr), int(n_fft), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def hann_window(n: int) -> np.ndarray:
    out = np.empty(n, np.float64)
    _check(lib().melspec_hann_window(n, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def kaldi_mel_filterbank(sample_rate=16000.0, fft_size=512, num_mel_bins=80, low_freq=20.0, high_freq=8000.0) -> np.ndarray:
    out = np.empty((num_mel_bins, fft_size // 2 + 1), np.float64)
    _check(lib().melspec_kaldi_mel_filterbank(sample_rate, fft_size, num_mel_bins, low_freq, high_freq,
                                              out.ctypes.data_as(C.POINTER(C.c_double))))
    return out
