"""The drop-in boundary: the C-ABI library loads, exports every symbol the header declares,
validates arguments like the reference, and the product never touches the oracle."""
import ctypes as C
import os
import re

import pytest

import mel_spec_amd as M
from mel_spec_amd import _lib
from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "melspec_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(melspec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/melspec_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES out of sync with the header"
    assert L.melspec_abi_version() == 1


def test_argument_validation_matches_reference():
    # src/cuda.rs:45-49: zero sizes are rejected at construction -> Unavailable
    for args in ((0, 160, 16000.0, 80), (400, 0, 16000.0, 80), (400, 160, 16000.0, 0), (400, 160, 0.0, 80)):
        with pytest.raises(M.HipUnavailable):
            M.HipMelSpectrogram(*args)
    h = C.c_void_p()
    assert _lib.lib().melspec_create(C.byref(h), -1, 0, 160, 16000.0, 80) == _lib.ERR_INVALID_ARG
    assert "non-zero" in _lib.last_error()
    assert _lib.lib().melspec_create(None, -1, 400, 160, 16000.0, 80) == _lib.ERR_INVALID_ARG
    with pytest.raises(M.HipUnavailable):
        M.Fbank(M.FbankConfig(num_mel_bins=0))


def test_no_device_is_reported_as_unavailable():
    if M.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(M.HipUnavailable) as e:
        M.HipMelSpectrogram(400, 160, 16000.0, 80)
    assert e.value.code == _lib.ERR_UNAVAILABLE       # lets callers self-skip like src/cuda.rs:512-518
    with pytest.raises(M.HipUnavailable):
        M.Fbank()


def test_fbank_config_defaults_match_reference():
    c = M.FbankConfig()   # src/fbank.rs:354-362
    assert (c.sample_rate, c.num_mel_bins, c.frame_length_samples(), c.frame_shift_samples(), c.fft_size()) == \
        (16000.0, 80, 400, 160, 512)
    cc = _lib.FbankConfigC()
    _lib.lib().melspec_fbank_default_config(C.byref(cc))
    d = c.to_c()
    for name, _ in _lib.FbankConfigC._fields_:
        assert getattr(cc, name) == getattr(d, name), name


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "mel_spec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".rs")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                for pat in (r"import\s+oracle", r"from\s+oracle", r"oracle[/\\.]", r"melspec_oracle", r"oracle_[a-z]"):
                    assert not re.search(pat, src), f"{f} references the oracle ({pat})"
    assert "oracle" not in open(os.path.join(ROOT, "include", "melspec_hip.h")).read()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libmelspec_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_null_handles_are_rejected_without_a_device():
    """Every per-object entry point answers a NULL handle with MELSPEC_ERR_INVALID_ARG and a message (no HIP call is made first, so
    this holds on a box without a GPU): never a crash across the ABI, like the integer status convention of src/cuda.rs:164."""
    L = _lib.lib()
    u64 = (C.c_uint64 * 1)(0)
    f32 = (C.c_float * 4)()
    tot = C.c_uint64(0)
    calls = [
        lambda: L.melspec_compute_batch_host(None, f32, u64, u64, 1, f32, None, 4, C.byref(tot)),
        lambda: L.melspec_fbank_compute_batch_host(None, f32, u64, u64, 1, f32, None, 4, C.byref(tot)),
        lambda: L.melspec_blm_compute_batch_host(None, f32, u64, u64, 1, f32, None, 4, C.byref(tot)),
        lambda: L.melspec_mel_from_stft_host(None, f32, 0, 0, 1, f32, 4),
        lambda: L.melspec_mel_from_stft_device(None, None, 0, 0, 1, None, None),
        lambda: L.melspec_fbank_compute_ragged_device(None, None, u64, u64, 1, None, None, None),
        lambda: L.melspec_blm_compute_ragged_device(None, None, u64, u64, 1, None, None, None),
        lambda: L.melspec_fbank_release_scratch(None),
        lambda: L.melspec_blm_release_scratch(None),
        lambda: L.melspec_release_scratch(None),
        lambda: L.melspec_set_precision(None, 0),
    ]
    for call in calls:
        assert call() == _lib.ERR_INVALID_ARG
        assert "NULL" in _lib.last_error()


def test_library_is_built_from_these_sources(tmp_path):
    """melspec_source_hash(): the hash of csrc/ + include/ baked into the library at build time equals the hash of this checkout (conftest
    rebuilds when it does not), and a library without / with another hash is reported as stale -- a prebuilt library that travelled with
    the snapshot cannot silently be an old one."""
    from mel_spec_amd import build as B, _lib
    assert _lib.lib().melspec_source_hash().decode() == B.source_hash() == B.built_hash()
    assert not B.needs_build()
    stale = tmp_path / "libstale.so"
    blob = open(B.LIB_PATH, "rb").read()
    stale.write_bytes(blob.replace(B.HASH_MARKER + B.source_hash().encode(), B.HASH_MARKER + b"0" * 32))
    assert B.built_hash(str(stale)) == "0" * 32 != B.source_hash()
    assert B.built_hash(str(tmp_path / "missing.so")) is None


def test_mel_scale_helpers_reproduce_the_references_known_answers():
    """hz_to_mel / mel_to_hz / mels_to_hz / mel_frequencies / fft_frequencies through the C ABI (host only, no device needed) on the
    values the reference's own tests pin (src/mel.rs:786-834, librosa's doc examples), and against the oracle's restatement."""
    import numpy as np
    import mel_spec_amd as M
    from oracle import oracle as O
    assert abs(M.hz_to_mel(60.0) - 0.9) <= 1e-3 and M.mel_to_hz(3.0) == 200.0
    assert np.abs(M.mels_to_hz([1.0, 2.0, 3.0, 4.0, 5.0]) - [66.667, 133.333, 200.0, 266.667, 333.333]).max() <= 1e-3
    want = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49, 1024.856, 1119.114,
            1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393, 2470.47, 2697.686, 2945.799, 3216.731,
            3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621, 5955.205, 6502.92, 7101.009, 7754.107, 8467.272, 9246.028,
            10096.408, 11025.]
    assert np.abs(M.mel_frequencies(40, 0.0, 11025.0) - want).max() <= 5e-3
    assert np.abs(M.fft_frequencies(22050.0, 16) - [0., 1378.125, 2756.25, 4134.375, 5512.5, 6890.625, 8268.75, 9646.875, 11025.]).max() <= 1e-3
    for htk in (False, True):
        for f in (0.0, 60.0, 999.9, 1000.0, 4000.0, 8000.0):
            assert abs(M.hz_to_mel(f, htk) - O.hz_to_mel(f, htk)) <= 1e-12 * max(1.0, abs(O.hz_to_mel(f, htk)))
            assert abs(M.mel_to_hz(M.hz_to_mel(f, htk), htk) - f) <= 1e-9 * max(1.0, f)
        assert np.abs(M.mel_frequencies(82, 0.0, 8000.0, htk) - O.mel_frequencies(82, 0.0, 8000.0, htk)).max() <= 1e-9
    assert np.array_equal(M.fft_frequencies(16000.0, 400), O.fft_frequencies(16000.0, 400))
