"""Streaming side of the reference over the C ABI (melspec_stream_*): a bank of live streams whose
overlap-save state lives in HBM, and a single-stream `RingBuffer` with the reference's method names
(src/rb.rs:18-121: add_frame / add / maybe_mel)."""
import ctypes as C
from collections import deque
from typing import List, Optional, Sequence

import numpy as np

from ._lib import lib
from .hip import HipMelSpectrogram, _check, _f32, _fp
from .vad import DetectionSettings, VadFrameTiming, VoiceActivity

# melspec_vad_activity (include/melspec_hip.h): one 8-byte record per emitted frame
ACTIVITY_DTYPE = np.dtype([("valid", "u1"), ("active", "u1"), ("leading_active_columns", "<u2"), ("active_columns", "<u2"),
                           ("window_columns", "<u2")])

_u32p = C.POINTER(C.c_uint32)


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


class StreamBank:
    """n_streams independent Spectrogram::add states (src/stft.rs:48-86) on one HipMelSpectrogram's geometry."""

    def __init__(self, mel: HipMelSpectrogram, n_streams: int, max_chunk: int):
        self._mel = mel                      # keeps the ctx alive
        self.n_streams, self.max_chunk, self.n_mels = int(n_streams), int(max_chunk), mel.n_mels
        h = C.c_void_p()
        _check(lib().melspec_stream_create(C.byref(h), mel._h, self.n_streams, self.max_chunk), construct=True)
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().melspec_stream_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self, ids: Optional[Sequence[int]] = None) -> None:
        if ids is None:
            _check(lib().melspec_stream_reset(self._h, None, 0))
        else:
            a = _u32(ids)
            _check(lib().melspec_stream_reset(self._h, a.ctypes.data_as(_u32p), a.shape[0]))

    def frames_after(self, stream: int, n_new: int) -> int:
        return int(lib().melspec_stream_frames_after(self._h, stream, n_new))

    def _collect(self, out: np.ndarray, frames: np.ndarray) -> List[np.ndarray]:
        res, cur = [], 0
        for f in frames:
            res.append(out[cur:cur + int(f)])
            cur += int(f)
        return res

    def push(self, ids: Sequence[int], chunks: Sequence) -> List[np.ndarray]:
        """Append chunks[i] (any length <= max_chunk) to stream ids[i]; returns the frames each stream emitted, [k_i, n_mels]."""
        a = _u32(ids)
        xs = [_f32(c).ravel() for c in chunks]
        assert len(xs) == a.shape[0]
        lens = _u32([x.shape[0] for x in xs])
        flat = np.concatenate(xs) if xs else np.zeros(0, np.float32)
        cap = sum(self.frames_after(int(s), int(n)) for s, n in zip(a, lens))
        out = np.empty((cap, self.n_mels), np.float32)
        frames = np.zeros(a.shape[0], np.uint32)
        _check(lib().melspec_stream_push_host(self._h, a.ctypes.data_as(_u32p), _fp(flat), lens.ctypes.data_as(_u32p), a.shape[0],
                                              _fp(out), out.size, frames.ctypes.data_as(_u32p)))
        assert int(frames.sum()) == cap
        return self._collect(out, frames)

    def push_stft(self, ids: Sequence[int], chunks: Sequence, dtype=np.complex128, full: bool = True) -> List[np.ndarray]:
        """push() emitting what Spectrogram::add returns (src/stft.rs:48-86): [k_i, bins] complex spectra per stream."""
        a = _u32(ids)
        xs = [_f32(c).ravel() for c in chunks]
        lens = _u32([x.shape[0] for x in xs])
        flat = np.concatenate(xs) if xs else np.zeros(0, np.float32)
        dt = np.dtype(dtype)
        bins = self._mel.stft_bins(full)
        cap = sum(self.frames_after(int(s), int(n)) for s, n in zip(a, lens))
        out = np.empty((cap, bins), dt)
        frames = np.zeros(a.shape[0], np.uint32)
        _check(lib().melspec_stream_push_host_stft(self._h, a.ctypes.data_as(_u32p), _fp(flat), lens.ctypes.data_as(_u32p), a.shape[0],
                                                   out.ctypes.data_as(C.c_void_p), out.size, frames.ctypes.data_as(_u32p),
                                                   int(dt == np.dtype(np.complex128)), int(full)))
        assert int(frames.sum()) == cap
        return self._collect(out, frames)

    def flush(self, ids: Sequence[int]) -> List[np.ndarray]:
        """Spectrogram::add with the pending (< hop) samples: zero-padded, at most one more frame per stream."""
        a = _u32(ids)
        out = np.empty((a.shape[0], self.n_mels), np.float32)
        frames = np.zeros(a.shape[0], np.uint32)
        _check(lib().melspec_stream_flush_host(self._h, a.ctypes.data_as(_u32p), a.shape[0], _fp(out), out.size, frames.ctypes.data_as(_u32p)))
        return self._collect(out, frames)

    # ---- the detector stage: VoiceActivityDetector::add_activity per stream, inside the push (src/vad.rs:155-205) ----
    def enable_vad(self, settings: Optional[DetectionSettings], timing: Optional[VadFrameTiming] = None) -> None:
        """Every push / flush from now on also feeds each stream's detector on the device; None turns the stage off.
        timing (VoiceActivityDetector::new_with_timing, src/vad.rs:149-153): records then carry their timestamps."""
        self._vad_timing = timing
        if settings is None:
            _check(lib().melspec_stream_enable_vad(self._h, None))
        else:
            c = settings._c()
            _check(lib().melspec_stream_enable_vad(self._h, C.byref(c)))

    def vad_frames(self, stream: int) -> int:
        """frames the stream has fed to its detector: VoiceActivityDetector::frame_index of the next one"""
        return int(lib().melspec_stream_vad_frames(self._h, stream))

    def _activities(self, ids: np.ndarray, first: Sequence[int], acts: np.ndarray, frames: np.ndarray) -> List[List[Optional[VoiceActivity]]]:
        res, cur = [], 0
        for i in range(ids.shape[0]):
            rows = []
            for k in range(int(frames[i])):
                a = acts[cur + k]
                if not a["valid"]:
                    rows.append(None)                   # add_activity returned None: fewer than min_x frames so far
                    continue
                w, n = int(a["window_columns"]), int(a["active_columns"])
                t = getattr(self, "_vad_timing", None)
                rows.append(VoiceActivity(bool(a["active"]), first[i] + k, int(a["leading_active_columns"]), n, w, 0.0 if w == 0 else n / w,
                                          None if t is None else t.timestamps_for_frame(first[i] + k)))
            res.append(rows)
            cur += int(frames[i])
        return res

    def push_vad(self, ids: Sequence[int], chunks: Sequence):
        """push() that also returns, per stream, what add_activity gives for each emitted frame (None or VoiceActivity)."""
        a = _u32(ids)
        xs = [_f32(c).ravel() for c in chunks]
        assert len(xs) == a.shape[0]
        lens = _u32([x.shape[0] for x in xs])
        flat = np.concatenate(xs) if xs else np.zeros(0, np.float32)
        cap = sum(self.frames_after(int(s), int(n)) for s, n in zip(a, lens))
        first = [self.vad_frames(int(s)) for s in a]
        out = np.empty((cap, self.n_mels), np.float32)
        acts = np.zeros(cap, ACTIVITY_DTYPE)
        frames = np.zeros(a.shape[0], np.uint32)
        _check(lib().melspec_stream_push_host_vad(self._h, a.ctypes.data_as(_u32p), _fp(flat), lens.ctypes.data_as(_u32p), a.shape[0],
                                                  _fp(out), out.size, frames.ctypes.data_as(_u32p), acts.ctypes.data_as(C.c_void_p), cap))
        assert int(frames.sum()) == cap
        return self._collect(out, frames), self._activities(a, first, acts, frames)

    def flush_vad(self, ids: Sequence[int]):
        a = _u32(ids)
        first = [self.vad_frames(int(s)) for s in a]
        out = np.empty((a.shape[0], self.n_mels), np.float32)
        acts = np.zeros(a.shape[0], ACTIVITY_DTYPE)
        frames = np.zeros(a.shape[0], np.uint32)
        _check(lib().melspec_stream_flush_host_vad(self._h, a.ctypes.data_as(_u32p), a.shape[0], _fp(out), out.size, frames.ctypes.data_as(_u32p),
                                                   acts.ctypes.data_as(C.c_void_p), acts.shape[0]))
        return self._collect(out, frames), self._activities(a, first, acts, frames)

    # ---- device producers -----------------------------------------------------------------
    def input_ptr(self, stream: int) -> int:
        return int(lib().melspec_stream_input_ptr(self._h, stream) or 0)

    def push_device(self, ids: Sequence[int], lens: Sequence[int], d_out: int, out_offsets=None, stream: int = 0) -> np.ndarray:
        a, ln = _u32(ids), _u32(lens)
        frames = np.zeros(a.shape[0], np.uint32)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, np.uint64)
        _check(lib().melspec_stream_push_device(self._h, a.ctypes.data_as(_u32p), ln.ctypes.data_as(_u32p), a.shape[0], C.c_void_p(d_out),
                                                None if oo is None else oo.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                frames.ctypes.data_as(_u32p), C.c_void_p(stream)))
        return frames

    def push_device_vad(self, ids: Sequence[int], lens: Sequence[int], d_out: int, d_acts: int, out_offsets=None, stream: int = 0) -> np.ndarray:
        """push_device whose activity records (ACTIVITY_DTYPE, packed in entry order) go to device memory at d_acts."""
        a, ln = _u32(ids), _u32(lens)
        frames = np.zeros(a.shape[0], np.uint32)
        oo = None if out_offsets is None else np.ascontiguousarray(out_offsets, np.uint64)
        _check(lib().melspec_stream_push_device_vad(self._h, a.ctypes.data_as(_u32p), ln.ctypes.data_as(_u32p), a.shape[0], C.c_void_p(d_out),
                                                    None if oo is None else oo.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                    frames.ctypes.data_as(_u32p), C.c_void_p(d_acts), C.c_void_p(stream)))
        return frames


class RingBuffer:
    """One stream with the reference's interface (src/rb.rs:18-121): add_frame / add feed samples, maybe_mel
    returns one (n_mels, 1) column per completed hop or None.  Frames are computed on the device in batches
    whenever maybe_mel runs dry, so calling it in the reference's `while let Some(mel) = rb.maybe_mel()` loop
    costs one launch per add_frame, not one per hop."""

    def __init__(self, mel: HipMelSpectrogram, capacity: int = 16384):
        self._bank = StreamBank(mel, 1, max(int(capacity), mel.hop_size))
        self._buf = deque()
        self._n = 0
        self._capacity = max(int(capacity), mel.hop_size)
        self._ready = deque()

    def add_frame(self, samples) -> None:
        x = _f32(samples).ravel()
        # src/rb.rs:60-68: when full, the oldest samples are dropped
        over = self._n + x.shape[0] - self._capacity
        while over > 0 and self._buf:
            head = self._buf[0]
            if head.shape[0] <= over:
                self._buf.popleft(); self._n -= head.shape[0]; over -= head.shape[0]
            else:
                self._buf[0] = head[over:]; self._n -= over; over = 0
        if x.shape[0] > self._capacity:
            x = x[-self._capacity:]
        self._buf.append(x)
        self._n += x.shape[0]

    def add(self, sample: float) -> None:
        self.add_frame(np.array([sample], np.float32))

    def maybe_mel(self) -> Optional[np.ndarray]:
        if not self._ready and self._n:
            x = np.concatenate(list(self._buf))
            self._buf.clear(); self._n = 0
            for f in self._bank.push([0], [x])[0]:
                self._ready.append(f)
        if not self._ready:
            return None
        return self._ready.popleft().astype(np.float64).reshape(-1, 1)

    def close(self) -> None:
        self._bank.close()


class Spectrogram:
    """Spectrogram::{new, add} (src/stft.rs:25-86) with the overlap-save state on the device: add(frames) takes at most hop_size
    samples and returns the [fft_size] complex128 spectrum of the window that ends with them once fft_size samples have been seen,
    else None.  A block shorter than hop_size is zero-padded like the reference's (:57-60); the reference then advances its sample
    count by the real samples only, this mirror by a whole hop -- which matters only for when the very first frame of a stream
    appears.  compute_all_cpu is HipMelSpectrogram.compute_all."""

    def __init__(self, fft_size: int, hop_size: int, device: int = -1):
        self.fft_size, self.hop_size = int(fft_size), int(hop_size)
        self._mel = HipMelSpectrogram(self.fft_size, self.hop_size, 16000.0, 1, device=device)     # geometry only: no mel stage is run
        self._bank = StreamBank(self._mel, 1, self.hop_size)

    def add(self, frames) -> Optional[np.ndarray]:
        x = _f32(frames).ravel()
        assert x.shape[0] <= self.hop_size, "frames must be <= hop_size"
        if x.shape[0] < self.hop_size:
            x = np.concatenate([x, np.zeros(self.hop_size - x.shape[0], np.float32)])
        out = self._bank.push_stft([0], [x], np.complex128, True)[0]
        return out[0] if out.shape[0] else None

    def close(self) -> None:
        self._bank.close(); self._mel.close()


class MelSpectrogram:
    """MelSpectrogram::{new, add} (src/mel.rs:13-32): add(fft) projects one spectrum ([fft_size] complex, what Spectrogram::add
    returns) onto the Slaney bank, log10, per-frame normalisation -> (n_mels, 1) float64 like the reference's Array2<f64>."""

    def __init__(self, fft_size: int, sampling_rate: float, n_mels: int, device: int = -1):
        from .hip import SparseMelFilterbank
        self._bank = SparseMelFilterbank.from_mel(float(sampling_rate), int(fft_size), int(n_mels), device=device)

    def add(self, fft) -> np.ndarray:
        # full f64 like the reference (project_stft_log10 src/mel.rs:148-168, then norm_mel over the frame :645-654): the device's
        # log_mel_spectrogram + norm_mel in f64 (round 4; the mel_from_stft route returned f32 values widened)
        a = np.ascontiguousarray(fft, np.complex128).reshape(1, -1)
        return self._bank.norm_mel(self._bank.log_mel_spectrogram(a)[0]).reshape(-1, 1)

    def close(self) -> None:
        self._bank.close()

