"""Host-side mirror of the reference's src/quant.rs over the C ABI (melspec_tga_*, melspec_quantize_*):
same function names, argument meaning and return shapes; all arithmetic runs in the HIP kernels of
csrc/tga_quant.hpp.  File I/O (save/load) is the only thing done here."""
import ctypes as C
from typing import List, Tuple

import numpy as np

from ._lib import lib
from .hip import HipRuntimeError, _check, _f32, _fp

U16_MAX = 65535


class QuantizationRange(tuple):
    """QuantizationRange { min, max } (src/quant.rs:5-9)."""

    def __new__(cls, mn, mx):
        return super().__new__(cls, (np.float32(mn), np.float32(mx)))

    min = property(lambda self: self[0])
    max = property(lambda self: self[1])


class TgaCodec:
    """Owns a melspec_tga handle (device stream + scratch)."""

    def __init__(self, device: int = -1):
        h = C.c_void_p()
        _check(lib().melspec_tga_create(C.byref(h), device), construct=True)
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().melspec_tga_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- layout ------------------------------------------------------------------------------
    @staticmethod
    def layout(n_mels: int, width: int) -> Tuple[int, int, int]:
        """(n_chunks, chunk_stride, last_chunk_bytes) of tga_8bit's output for one [n_mels][width] image."""
        n, st, last = C.c_uint32(), C.c_size_t(), C.c_size_t()
        _check(lib().melspec_tga_layout(n_mels, width, C.byref(n), C.byref(st), C.byref(last)))
        return n.value, st.value, last.value

    # ---- host API (src/quant.rs names) -----------------------------------------------------------
    def quantize(self, frame) -> Tuple[np.ndarray, QuantizationRange]:
        x = _f32(frame).ravel()
        out = np.empty(x.shape[0], np.uint8)
        rng = np.empty(2, np.float32)
        _check(lib().melspec_quantize_host(self._h, _fp(x), x.shape[0], out.ctypes.data_as(C.c_void_p), _fp(rng)))
        return out, QuantizationRange(rng[0], rng[1])

    def dequantize(self, data, rng) -> np.ndarray:
        d = np.ascontiguousarray(data, np.uint8).ravel()
        r = np.asarray(tuple(rng), np.float32).copy()
        out = np.empty(d.shape[0], np.float32)
        _check(lib().melspec_dequantize_host(self._h, d.ctypes.data_as(C.c_void_p), d.shape[0], _fp(r), _fp(out)))
        return out

    def tga_8bit(self, data, n_mels: int) -> List[bytes]:
        """tga_8bit (src/quant.rs:29-36): one TGA per <= 65535-column chunk of the [n_mels][width] image."""
        x = _f32(data).ravel()
        width = x.shape[0] // n_mels
        n, st, last = self.layout(n_mels, width)
        if n == 0:
            return []
        out = np.empty(st * (n - 1) + last, np.uint8)
        got = C.c_uint32()
        _check(lib().melspec_tga_encode_host(self._h, _fp(x), x.shape[0], n_mels, out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(got)))
        assert got.value == n
        full = 26 + n_mels * U16_MAX
        return [out[c * st:c * st + (full if c + 1 < n else last)].tobytes() for c in range(n)]

    def tga_8bit_data(self, data, n_mels: int) -> bytes:
        """tga_8bit_data (src/quant.rs:38-64); the image must fit one chunk."""
        blobs = self.tga_8bit(data, n_mels)
        if len(blobs) != 1:
            raise HipRuntimeError(-1, "tga_8bit_data: width must be in 1..65535, use tga_8bit")
        return blobs[0]

    def parse_tga_8bit(self, blob: bytes) -> np.ndarray:
        b = np.frombuffer(blob, np.uint8)
        out = np.empty(max(0, b.shape[0] - 26), np.float32)
        n = C.c_size_t()
        _check(lib().melspec_tga_decode_host(self._h, b.ctypes.data_as(C.c_void_p), b.shape[0], _fp(out), out.shape[0], C.byref(n)))
        return out[:n.value]

    def save_tga_8bit(self, data, n_mels: int, path: str) -> None:
        """save_tga_8bit (src/quant.rs:15-27); the reference asserts width < u16::MAX."""
        x = _f32(data).ravel()
        assert x.shape[0] // n_mels < U16_MAX, "width greater than TARGA max, use tga_8bit"
        with open(path, "wb") as f:
            f.write(self.tga_8bit_data(x, n_mels))

    def load_tga_8bit(self, path: str) -> np.ndarray:
        with open(path, "rb") as f:
            return self.parse_tga_8bit(f.read())

    # ---- device API --------------------------------------------------------------------------------
    def encode_device(self, d_images: int, image_stride: int, n_mels: int, width: int, n_images: int, d_blobs: int,
                      blob_stride: int, stream: int = 0) -> None:
        _check(lib().melspec_tga_encode_device(self._h, C.c_void_p(d_images), image_stride, n_mels, width, n_images,
                                               C.c_void_p(d_blobs), blob_stride, C.c_void_p(stream)))

    def encode_pcm_uniform_device(self, mel, d_pcm: int, clip_stride: int, clip_len: int, n_clips: int, min_width: int, d_images: int,
                                  d_blobs: int, blob_stride: int, stream: int = 0) -> None:
        """melspec_tga_encode_pcm_uniform_device: PCM -> mel-major images (left in d_images) -> one TGA blob per clip, the image's
        {min, max} folded by the mel kernel while it stores (`mel`: a HipMelSpectrogram on the same device)"""
        _check(lib().melspec_tga_encode_pcm_uniform_device(self._h, mel._h, C.c_void_p(d_pcm), clip_stride, clip_len, n_clips, min_width,
                                                           C.c_void_p(d_images), C.c_void_p(d_blobs), blob_stride, C.c_void_p(stream)))

    def decode_device(self, d_blobs: int, blob_stride: int, n_mels: int, width: int, n_images: int, d_images: int,
                      image_stride: int, stream: int = 0) -> None:
        _check(lib().melspec_tga_decode_device(self._h, C.c_void_p(d_blobs), blob_stride, n_mels, width, n_images,
                                               C.c_void_p(d_images), image_stride, C.c_void_p(stream)))

    def synchronize(self) -> None:
        _check(lib().melspec_tga_synchronize(self._h))


def to_array2(frames, n_mels: int) -> np.ndarray:
    """to_array2 (src/quant.rs:168-174): back to (n_mels, width) f64."""
    x = np.asarray(frames, np.float32)
    return x.reshape(n_mels, x.shape[0] // n_mels).astype(np.float64)


def chunk_frames_into_strides(frames, n_mels: int, stride_size: int) -> List[np.ndarray]:
    """chunk_frames_into_strides (src/quant.rs:100-136): the [n_mels][width] image cut into stride_size x stride_size tiles in
    row-major tile order, each tile flattened row-major; the whole image when stride_size == width.  Host utility (tga_8bit's
    chunking itself happens in the encoder's layout on the device)."""
    x = _f32(frames).ravel()
    width = x.shape[0] // n_mels
    if stride_size == width:
        return [x]
    img = x.reshape(n_mels, width)
    return [np.ascontiguousarray(img[y:y + stride_size, c:c + stride_size]).ravel()
            for y in range(0, n_mels, stride_size) for c in range(0, width, stride_size)]

