"""src/quant.rs: 8-bit quantisation + TGA container.

CPU part: the oracle against the reference's own artifact testdata/quantized_mel_golden.tga (byte for byte),
and the kernels' per-thread functions (csrc/tga_quant.hpp, run on the host by tests/emu) against the oracle.
GPU part: the HIP kernels through the C ABI against both.  Everything here is integer/byte work: bit-exact."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _golden_image(oracle, jfk):
    """The image behind quantized_mel_golden.tga (used by src/vad.rs:684-744): the streaming Whisper
    400/160/80 mel of jfk_f32le.wav -- batch frames of samples[80:], SURVEY 3.5 -- mel-major, after two
    all-floor columns, 80 x 1100."""
    m = oracle.compute_mel_spectrogram_cpu(jfk[80:], 400, 160, 80)
    return np.concatenate([np.full((80, 2), -1.5, np.float32), m.T], axis=1)


@pytest.fixture(scope="module")
def gold_blob():
    with open(os.path.join(GOLDEN, "quantized_mel_golden.tga"), "rb") as f:
        return f.read()


@pytest.fixture(scope="module")
def qemu():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", d, "-s"])
    L = C.CDLL(os.path.join(d, "libmelspec_emu.so"))
    vp = C.c_void_p
    L.emu_tga_encode.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, vp, C.c_uint64, C.c_int, vp]
    L.emu_tga_decode.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, vp, C.c_uint64, C.c_int, vp]

    class E:
        @staticmethod
        def tga(img, rows):
            x = np.ascontiguousarray(img, np.float32).ravel()
            width = x.shape[0] // rows
            chunks = (width + 65534) // 65535
            cw = min(width, 65535)
            stride = (26 + rows * cw + 3) & ~3
            out = np.zeros(stride * chunks + 8, np.uint8)
            assert L.emu_tga_encode(x.ctypes.data, rows, width, 1, x.shape[0], out.ctypes.data, stride * chunks, 1, None) == chunks
            return [out[c * stride:c * stride + 26 + rows * min(cw, width - c * cw)].tobytes() for c in range(chunks)]

        @staticmethod
        def quantize(x):
            x = np.ascontiguousarray(x, np.float32).ravel()
            out = np.zeros((x.shape[0] + 3) & ~3, np.uint8)
            rng = np.zeros(2, np.float32)
            L.emu_tga_encode(x.ctypes.data, 1, x.shape[0], 1, x.shape[0], out.ctypes.data, out.shape[0], 0, rng.ctypes.data)
            return out[:x.shape[0]], rng

        @staticmethod
        def parse(blob, rows):
            b = np.frombuffer(blob + b"\0\0\0\0", np.uint8).copy()
            n = len(blob) - 26
            out = np.full(n, np.nan, np.float32)
            L.emu_tga_decode(b.ctypes.data, (len(blob) + 3) & ~3, rows, n // rows, 1, out.ctypes.data, n, 1, None)
            return out

    return E


# ---- oracle pinned by the reference's artifact ---------------------------------------------------

def test_oracle_reproduces_the_reference_tga_byte_for_byte(oracle, jfk, gold_blob):
    img = _golden_image(oracle, jfk)
    blob = oracle.tga_8bit_data(img, 80)
    assert len(blob) == len(gold_blob) == 26 + 80 * 1100
    assert blob[:26] == gold_blob[:26]          # header incl. {min,max} = {-1.5, 1.5359322} bit for bit
    assert blob == gold_blob


def test_oracle_parse_and_requantise(oracle, gold_blob):
    vals = oracle.parse_tga_8bit(gold_blob)
    mn, mx = np.frombuffer(gold_blob[18:26], "<f4")
    px = np.frombuffer(gold_blob[26:], np.uint8)
    assert vals.shape == (88000,) and vals.min() == mn and abs(vals.max() - mx) < 1e-6
    q, _ = oracle.quantize(vals)
    assert np.array_equal(q, px)                # dequantize -> quantize is the identity on the bytes
    with pytest.raises(ValueError):
        oracle.parse_tga_8bit(gold_blob[:25])


def test_oracle_quantize_corner_cases(oracle):
    q, r = oracle.quantize(np.zeros(5, np.float32))          # max == min: scale = inf, 0*inf = NaN -> 0
    assert q.tolist() == [0] * 5 and r == (0.0, 0.0)
    q, r = oracle.quantize(np.array([np.nan, 1, 2, 1.5], np.float32))   # NaN skipped by the folds, pixel 0
    assert q.tolist() == [0, 0, 255, 128] and r == (1.0, 2.0)
    q, r = oracle.quantize(np.array([0.0, 0.5, 1.0], np.float32) * 255)   # x.5 rounds away from zero
    assert q.tolist() == [0, 128, 255]
    assert [len(b) for b in oracle.tga_8bit(np.zeros(3 * 70000, np.float32), 3)] == [26 + 3 * 65535, 26 + 3 * 4465]
    assert oracle.tga_8bit(np.zeros(0, np.float32), 80) == []


# ---- the kernels' arithmetic, on the host --------------------------------------------------------------

def test_emulated_kernels_match_golden_and_oracle(qemu, oracle, jfk, gold_blob):
    img = _golden_image(oracle, jfk)
    assert qemu.tga(img, 80) == [gold_blob]
    assert np.array_equal(qemu.parse(gold_blob, 80), oracle.parse_tga_8bit(gold_blob))


@pytest.mark.parametrize("rows,width", [(80, 1), (80, 3), (80, 998), (128, 3000), (1, 7), (3, 65535), (3, 65536), (2, 140001)])
def test_emulated_kernels_shapes(qemu, oracle, rows, width):
    rng = np.random.default_rng(rows * 1000003 + width)
    img = (rng.standard_normal((rows, width)) * 3).astype(np.float32)
    want = oracle.tga_8bit(img, rows)
    got = qemu.tga(img, rows)
    assert got == want
    for b in want[:2]:
        assert np.array_equal(qemu.parse(b, rows), oracle.parse_tga_8bit(b))


def test_emulated_quantize_corner_cases(qemu, oracle):
    for x in (np.zeros(5, np.float32), np.array([np.nan, 1, 2, 1.5], np.float32), np.array([np.nan] * 3, np.float32),
              np.array([0.0, 0.5, 1.0], np.float32) * 255, np.array([-np.inf, 0, 1], np.float32), np.array([3.0], np.float32),
              np.random.default_rng(1).standard_normal(1001).astype(np.float32)):
        q, r = qemu.quantize(x)
        qo, ro = oracle.quantize(x)
        assert np.array_equal(q, qo) and np.array_equal(np.asarray(ro, np.float32).view(np.uint32), r.view(np.uint32)), x[:4]


# ---- the HIP kernels ------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def codec(gpu):
    c = gpu.TgaCodec()
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_tga(codec, oracle, jfk, gold_blob):
    img = _golden_image(oracle, jfk)
    assert codec.tga_8bit(img, 80) == [gold_blob]
    assert codec.tga_8bit_data(img, 80) == gold_blob
    assert np.array_equal(codec.parse_tga_8bit(gold_blob), oracle.parse_tga_8bit(gold_blob))


@pytest.mark.gpu
def test_gpu_mel_to_tga_end_to_end(gpu, codec, oracle, jfk, gold_blob):
    """PCM -> fused mel kernel (mel-major store) -> quantiser, all on the device, against the reference's file.
    The f32 FFT moves a value across a rounding boundary of the 8-bit grid now and then (by one step);
    the precise build leaves a handful at most."""
    gold_px = np.frombuffer(gold_blob[26:], np.uint8).reshape(80, 1100)[:, 2:]
    for precise, max_flips in ((False, 400), (True, 8)):
        m = gpu.HipMelSpectrogram(400, 160, 16000.0, 80)
        m.set_precise(precise)
        mel = m.compute_mel_spectrogram(jfk[80:])                 # (1098, 80)
        img = np.ascontiguousarray(mel.T)
        img[img < -1.5] = -1.5
        # same {min,max} as the golden as long as the extreme values agree to the bit; quantise with the header's range
        blob = codec.tga_8bit_data(np.concatenate([np.full((80, 2), -1.5, np.float32), img], axis=1), 80)
        px = np.frombuffer(blob[26:], np.uint8).reshape(80, 1100)[:, 2:]
        d = np.abs(px.astype(int) - gold_px.astype(int))
        assert d.max() <= 1 and (d != 0).sum() <= max_flips, (precise, d.max(), (d != 0).sum())
        assert np.frombuffer(blob[18:22], "<f4")[0] == -1.5
        assert abs(np.frombuffer(blob[22:26], "<f4")[0] - np.frombuffer(gold_blob[22:26], "<f4")[0]) <= (1e-4 if not precise else 4e-7)
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,width", [(80, 1), (80, 3), (80, 998), (128, 3000), (1, 7), (3, 65535), (3, 65536), (2, 140001)])
def test_gpu_shapes_match_oracle(codec, oracle, rows, width):
    rng = np.random.default_rng(rows * 1000003 + width)
    img = (rng.standard_normal((rows, width)) * 3).astype(np.float32)
    want = oracle.tga_8bit(img, rows)
    assert codec.tga_8bit(img, rows) == want
    for b in want:
        assert np.array_equal(codec.parse_tga_8bit(b), oracle.parse_tga_8bit(b))


@pytest.mark.gpu
def test_gpu_quantize_corner_cases_and_errors(gpu, codec, oracle):
    for x in (np.zeros(5, np.float32), np.array([np.nan, 1, 2, 1.5], np.float32), np.array([np.nan] * 3, np.float32),
              np.array([0.0, 0.5, 1.0], np.float32) * 255, np.array([-np.inf, 0, 1], np.float32), np.array([3.0], np.float32),
              np.random.default_rng(1).standard_normal(1001).astype(np.float32)):
        q, r = codec.quantize(x)
        qo, ro = oracle.quantize(x)
        assert np.array_equal(q, qo), x[:4]
        assert np.array_equal(np.asarray(ro, np.float32).view(np.uint32), np.asarray(tuple(r), np.float32).view(np.uint32))
        assert np.array_equal(codec.dequantize(q, r).view(np.uint32), oracle.dequantize(qo, ro).view(np.uint32))
    assert codec.tga_8bit(np.zeros(0, np.float32), 80) == []
    with pytest.raises(gpu.HipRuntimeError):
        codec.parse_tga_8bit(b"\0" * 25)
    with pytest.raises(gpu.HipRuntimeError):
        codec.tga_8bit(np.zeros(81, np.float32), 80)


@pytest.mark.gpu
def test_gpu_batched_device_path_config2_shape(gpu, codec, oracle):
    """[clips][80][W] images straight from the mel kernel's mel-major store -> one TGA per clip, then back."""
    n_clips, clip_len = 48, 160000
    m = gpu.HipMelSpectrogram(400, 160, 16000.0, 80)
    W = m.interleaved_width(clip_len, 3000)
    assert W == 3000
    pcm = gpu.DeviceBuffer(n_clips * clip_len * 4)
    gpu.synth_pcm_device(pcm.ptr, clip_len, clip_len, 0, n_clips)
    img = gpu.DeviceBuffer(n_clips * 80 * W * 4)
    m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, img.ptr, False, 3000)
    m.synchronize()
    n, stride, last = codec.layout(80, W)
    assert (n, last) == (1, 26 + 80 * W) and stride % 4 == 0
    blobs = gpu.DeviceBuffer(n_clips * stride)
    codec.encode_device(img.ptr, 80 * W, 80, W, n_clips, blobs.ptr, stride)
    back = gpu.DeviceBuffer(n_clips * 80 * W * 4)
    codec.decode_device(blobs.ptr, stride, 80, W, n_clips, back.ptr, 80 * W)
    codec.synchronize()
    host_img = img.download((n_clips, 80 * W))
    host_blob = blobs.download((n_clips, stride), np.uint8)
    host_back = back.download((n_clips, 80 * W))
    for c in (0, 1, 23, 47):
        want = oracle.tga_8bit_data(host_img[c], 80)
        assert host_blob[c, :last].tobytes() == want
        assert np.array_equal(host_back[c].view(np.uint32), oracle.parse_tga_8bit(want).view(np.uint32))
    step = (host_img.max(1) - host_img.min(1)) / 255
    assert (np.abs(host_back - host_img).max(1) <= step * 0.5001).all()      # quantisation error bound, every clip
    for b in (pcm, img, blobs, back):
        b.free()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_mels,mode,min_width,kind", [(80, "auto", 2, "noise"), (80, "auto", 3000, "noise"), (80, "auto", 2, "speech"), (80, "f64", 0, "speech"),
                                                        (80, "f32", 1200, "speech"), (128, "auto", 2, "speech"), (128, "auto", 0, "noise"), (64, "auto", 2, "tone")])
def test_gpu_pcm_to_tga_with_the_minmax_folded_into_the_mel_store(gpu, codec, oracle, jfk, n_mels, mode, min_width, kind):
    """melspec_tga_encode_pcm_uniform_device: the mel kernel folds every image's {min, max} while it stores the image (also through
    the f64 recompute of AUTO's tripped frames, whose f32 values must not leave a trace in the range, and through zero-padded
    columns), the quantiser reads the image once.  Byte for byte what the two-pass form and the oracle make of the same images."""
    n_clips, clip_len = 40, 48000
    if kind == "noise":
        x = np.stack([oracle.synth_pcm(c, clip_len) for c in range(n_clips)])
    elif kind == "speech":
        x = np.stack([np.resize(np.roll(jfk, -2111 * c), clip_len) for c in range(n_clips)])
    else:
        t = np.arange(clip_len) / 16000.0
        rng = np.random.default_rng(4)
        x = np.stack([(0.9 * np.sin(2 * np.pi * (300 + 170 * c) * t) + 10 ** (-70 / 20) * rng.standard_normal(clip_len)) for c in range(n_clips)])
    x = x.astype(np.float32)
    m = gpu.HipMelSpectrogram(400, 160, 16000.0, n_mels)
    m.set_precision(mode)
    m.set_auto_adaptive(False)                 # one regime for both forms (bits are compared)
    W = m.interleaved_width(clip_len, min_width)
    pcm = gpu.DeviceBuffer(x.nbytes); pcm.upload(x)
    n, stride, last = codec.layout(n_mels, W)
    assert n == 1
    img = gpu.DeviceBuffer(n_clips * n_mels * W * 4); img2 = gpu.DeviceBuffer(n_clips * n_mels * W * 4)
    blobs = gpu.DeviceBuffer(n_clips * stride); blobs2 = gpu.DeviceBuffer(n_clips * stride)
    for rep in range(2):                       # twice: the keys are reset per call
        codec.encode_pcm_uniform_device(m, pcm.ptr, clip_len, clip_len, n_clips, min_width, img.ptr, blobs.ptr, stride)
        m.synchronize()
    if mode == "auto" and kind != "noise":
        assert m.guard_last_count() > 0          # the recompute tail took part
    m.compute_uniform_device_interleaved(pcm.ptr, clip_len, clip_len, n_clips, img2.ptr, False, min_width)
    m.synchronize()
    codec.encode_device(img2.ptr, n_mels * W, n_mels, W, n_clips, blobs2.ptr, stride)
    codec.synchronize()
    a, b = blobs.download((n_clips, stride), np.uint8), blobs2.download((n_clips, stride), np.uint8)
    imgs = img.download((n_clips, n_mels * W))
    assert np.array_equal(imgs, img2.download((n_clips, n_mels * W)))
    assert np.array_equal(a[:, :last], b[:, :last])
    for c in (0, 7, n_clips - 1):
        assert a[c, :last].tobytes() == oracle.tga_8bit_data(imgs[c], n_mels)
        want = oracle.interleave_frames(oracle.compute_mel_spectrogram_cpu(x[c], 400, 160, n_mels, 16000.0), False, min_width)
        assert np.abs(imgs[c].reshape(n_mels, W) - want).max() <= (1e-4 if not (mode == "f32" and kind == "tone") else 1e-3)
    for buf in (pcm, img, img2, blobs, blobs2):
        buf.free()
    m.close()


def test_chunk_frames_into_strides_mirror(oracle):
    """chunk_frames_into_strides (src/quant.rs:100-136), host utility of the mirror, against the oracle's restatement."""
    import mel_spec_amd as M
    x = np.arange(80 * 25, dtype=np.float32)
    for st in (25, 10, 80, 7, 1000):
        a, b = M.chunk_frames_into_strides(x, 80, st), oracle.chunk_frames_into_strides(x, 80, st)
        assert len(a) == len(b) and all(np.array_equal(p, q) for p, q in zip(a, b)), st
