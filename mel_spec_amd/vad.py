"""Host-side mirror of the reference's src/vad.rs over the C ABI (melspec_vad_*): DetectionSettings,
vad_boundaries -> EdgeInfo, vad_on, VoiceActivityDetector.  The column classification (Sobel stencil, count,
majority vote) runs on the device; list building and the detector's window bookkeeping are host logic."""
import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from ._lib import VadSettingsC, lib
from .hip import _check, _f32, _fp


@dataclass
class DetectionSettings:
    """DetectionSettings (src/vad.rs:5-22)."""
    min_energy: float = 0.98
    min_y: int = 11
    min_x: int = 5
    min_mel: int = 2

    def _c(self) -> VadSettingsC:
        return VadSettingsC(float(self.min_energy), int(self.min_y), int(self.min_x), int(self.min_mel))


class EdgeInfo:
    """EdgeInfo (src/vad.rs:488-522); gradient_positions is empty in the reference's current version too."""

    def __init__(self, smoothed: np.ndarray, raw: Optional[np.ndarray] = None, longest_run: int = 0):
        self.smoothed, self.raw, self.longest_run = smoothed, raw, int(longest_run)

    def intersected(self) -> List[int]:
        return np.nonzero(self.smoothed)[0].tolist()

    def non_intersected(self) -> List[int]:
        return np.nonzero(~self.smoothed)[0].tolist()

    def gradient_positions(self) -> set:
        return set()


def vad_boundaries(frames: Sequence, settings: DetectionSettings, device: int = -1) -> EdgeInfo:
    """vad_boundaries(&[Array2<f64>], &settings) (src/vad.rs:256-340): frames are (n_mels, w_i) arrays, concatenated in time."""
    if len(frames) == 0:
        return EdgeInfo(np.zeros(0, bool))
    img = _f32(np.concatenate([np.asarray(f, np.float32).reshape(np.asarray(f).shape[0], -1) for f in frames], axis=1))
    h, w = img.shape
    n = int(lib().melspec_vad_mask_len(h, w))
    raw = np.zeros(n, np.uint8)
    sm = np.zeros(n, np.uint8)
    run = C.c_uint32(0)
    s = settings._c()
    _check(lib().melspec_vad_boundaries_host(device, _fp(img), h, w, C.byref(s), raw.ctypes.data_as(C.c_void_p),
                                             sm.ctypes.data_as(C.c_void_p), C.byref(run)))
    return EdgeInfo(sm.astype(bool), raw.astype(bool), run.value)


def vad_on(edge_info: EdgeInfo, n: int) -> bool:
    """vad_on (src/vad.rs:229-254): the counter is only tested from the second intersected column on."""
    if n <= 1:
        return int(edge_info.smoothed.sum()) >= 2
    return edge_info.longest_run >= n


def leading_active_columns(intersected: Sequence[int]) -> int:
    expected = 0
    for c in intersected:                       # src/vad.rs:216-227
        if c == expected:
            expected += 1
        elif c > expected:
            break
    return expected


@dataclass
class VoiceActivityTimestamps:
    """VoiceActivityTimestamps (src/vad.rs:119-124)."""
    start_ms: int
    center_ms: int
    end_ms: int


def _sample_to_ms(sample: int, sampling_rate: float) -> int:
    return int(math.floor(sample / sampling_rate * 1000.0 + 0.5))        # f64::round of a non-negative value (src/vad.rs:208-210)


@dataclass
class VadFrameTiming:
    """VadFrameTiming (src/vad.rs:90-117): where frame i of the STFT sits in the stream."""
    fft_size: int
    hop_size: int
    sampling_rate: float

    def timestamps_for_frame(self, frame_index: int) -> VoiceActivityTimestamps:
        start = frame_index * self.hop_size
        return VoiceActivityTimestamps(_sample_to_ms(start, self.sampling_rate), _sample_to_ms(start + self.fft_size // 2, self.sampling_rate),
                                       _sample_to_ms(start + self.fft_size, self.sampling_rate))


@dataclass
class VoiceActivity:
    """VoiceActivity (src/vad.rs:126-135)."""
    active: bool
    frame_index: int
    leading_active_columns: int
    active_columns: int
    window_columns: int
    confidence: float
    timestamps: Optional[VoiceActivityTimestamps] = None


def n_frames_for_duration(hop_size: int, sampling_rate: float, duration_ms: int) -> int:
    """FFT frames needed for duration_ms (src/vad.rs:579-583; the reference computes this one in f32)."""
    frame_duration = np.float32(hop_size) / np.float32(sampling_rate) * np.float32(1000.0)
    return int(np.ceil(np.float32(duration_ms) / frame_duration))


def duration_ms_for_n_frames(hop_size: int, sampling_rate: float, total_frames: int) -> int:
    """milliseconds total_frames FFT frames represent (src/vad.rs:586-589)."""
    return int(total_frames * (hop_size / sampling_rate * 1000.0))


def format_milliseconds(milliseconds: int) -> str:
    """HH:MM:SS.mmm (src/vad.rs:592-601)."""
    total_seconds, ms = divmod(int(milliseconds), 1000)
    total_minutes, seconds = divmod(total_seconds, 60)
    hours, minutes = divmod(total_minutes, 60)
    return f"{hours:02}:{minutes:02}:{seconds:02}.{ms:03}"


class VoiceActivityDetector:
    """VoiceActivityDetector::{new, add, add_activity} (src/vad.rs:137-208): the last min_x single-column frames
    form the window that vad_boundaries classifies."""

    def __init__(self, settings: DetectionSettings, device: int = -1, timing: Optional[VadFrameTiming] = None):
        self.settings, self.device, self.timing = settings, device, timing
        self.mel_buffer: List[np.ndarray] = []
        self.frame_index = 0

    @classmethod
    def new_with_timing(cls, settings: DetectionSettings, timing: VadFrameTiming, device: int = -1) -> "VoiceActivityDetector":
        return cls(settings, device, timing)                  # src/vad.rs:149-153

    def add(self, frame) -> Optional[bool]:
        a = self.add_activity(frame)
        return None if a is None else a.active

    def add_activity(self, frame) -> Optional[VoiceActivity]:
        idx = self.frame_index
        self.frame_index += 1
        min_x = self.settings.min_x
        self.mel_buffer.append(np.asarray(frame, np.float32).reshape(np.asarray(frame).shape[0], -1))
        if len(self.mel_buffer) > max(min_x, 128):
            self.mel_buffer = self.mel_buffer[len(self.mel_buffer) - min_x:]
        if len(self.mel_buffer) < min_x:
            return None
        e = vad_boundaries(self.mel_buffer[len(self.mel_buffer) - min_x:], self.settings, self.device)
        inter = e.intersected()
        window = len(inter) + len(e.non_intersected())
        return VoiceActivity(bool(inter) and inter[0] == 0, idx, leading_active_columns(inter), len(inter), window,
                             0.0 if window == 0 else len(inter) / window,
                             None if self.timing is None else self.timing.timestamps_for_frame(idx))
