"""ctypes loader of libmelspec_hip.so.  There is no CPU fallback: if the HIP library is
missing or fails to load, importing the product API raises."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

OK = 0
ERR_INVALID_ARG = -1
ERR_UNAVAILABLE = -2
ERR_CAPACITY = -3
ERR_UNSUPPORTED = -4
ERR_INTERNAL = -5


class VadSettingsC(C.Structure):
    _fields_ = [("min_energy", C.c_double), ("min_y", C.c_int), ("min_x", C.c_int), ("min_mel", C.c_int)]


class FbankConfigC(C.Structure):
    """melspec_fbank_config (include/melspec_hip.h) == FbankConfig (src/fbank.rs:25-64)."""
    _fields_ = [
        ("sample_rate", C.c_double), ("num_mel_bins", C.c_int32),
        ("frame_length_ms", C.c_double), ("frame_shift_ms", C.c_double),
        ("energy_floor", C.c_double), ("use_log_fbank", C.c_int32), ("use_power", C.c_int32),
        ("preemphasis", C.c_double), ("apply_cmn", C.c_int32),
        ("low_freq", C.c_double), ("high_freq", C.c_double),
    ]


class BlmConfigC(C.Structure):
    """melspec_blm_config (include/melspec_hip.h) == BatchLogMelConfig (src/mel.rs:171-208)."""
    _fields_ = [
        ("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32), ("hop_length", C.c_int32),
        ("n_mels", C.c_int32), ("f_min", C.c_double), ("f_max", C.c_double), ("htk", C.c_int32), ("norm", C.c_int32),
        ("preemphasis", C.c_float), ("center", C.c_int32), ("log_zero_guard", C.c_float), ("pad_to", C.c_int32),
        ("normalize_per_feature", C.c_int32),
    ]


# name -> (restype, argtypes); must list every symbol include/melspec_hip.h declares
_vp, _f32p, _f64p, _u64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
SIGNATURES = {
    "melspec_abi_version": (C.c_int, []),
    "melspec_source_hash": (C.c_char_p, []),
    "melspec_device_count": (C.c_int, []),
    "melspec_last_error": (C.c_char_p, []),
    "melspec_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
    "melspec_create_with_filterbank": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]),
    "melspec_create_with_dense_filterbank": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _f64p, C.c_int]),
    "melspec_destroy": (None, [_vp]),
    "melspec_bank_from_dense": (C.c_int, [C.POINTER(_vp), C.c_int, _f64p, C.c_int, C.c_int]),
    "melspec_bank_from_mel": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]),
    "melspec_bank_destroy": (None, [_vp]),
    "melspec_bank_n_mels": (C.c_int, [_vp]),
    "melspec_bank_fft_bins": (C.c_int, [_vp]),
    "melspec_bank_non_zero_weights": (C.c_int, [_vp]),
    "melspec_bank_weights_for_mel": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]),
    "melspec_bank_project_power_device": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64, _vp, _vp]),
    "melspec_bank_project_power_host": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t, _vp]),
    "melspec_bank_log_mel_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_uint64, _vp, _vp]),
    "melspec_bank_log_mel_host": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _f64p]),
    "melspec_bank_norm_mel_device": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64, _vp, _vp]),
    "melspec_bank_norm_mel_host": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t, _vp]),
    "melspec_num_frames": (C.c_size_t, [_vp, C.c_size_t]),
    "melspec_max_frames_per_batch": (C.c_size_t, [_vp]),
    "melspec_fft_size": (C.c_int, [_vp]),
    "melspec_hop_size": (C.c_int, [_vp]),
    "melspec_n_mels": (C.c_int, [_vp]),
    "melspec_uses_fast_path": (C.c_int, [_vp]),
    "melspec_set_precision": (C.c_int, [_vp, C.c_int]),
    "melspec_precision": (C.c_int, [_vp]),
    "melspec_guard_count": (C.c_int, [_vp, _u64p]),
    "melspec_set_auto_adaptive": (C.c_int, [_vp, C.c_int]),
    "melspec_auto_state": (C.c_int, [_vp, C.POINTER(C.c_int), _f64p]),
    "melspec_plain_kernel_name": (C.c_char_p, [_vp]),
    "melspec_set_precise": (C.c_int, [_vp, C.c_int]),
    "melspec_is_precise": (C.c_int, [_vp]),
    "melspec_compute_host": (C.c_int, [_vp, _f32p, C.c_size_t, _f32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "melspec_compute_batch_host": (C.c_int, [_vp, _f32p, _u64p, _u64p, C.c_uint32, _f32p, _u64p, C.c_size_t, _u64p]),
    "melspec_mel_from_stft_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_uint64, _vp, _vp]),
    "melspec_mel_from_stft_host": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_size_t, _f32p, C.c_size_t]),
    "melspec_release_scratch": (C.c_int, [_vp]),
    "melspec_fbank_release_scratch": (C.c_int, [_vp]),
    "melspec_blm_release_scratch": (C.c_int, [_vp]),
    "melspec_blm_set_precision": (C.c_int, [_vp, C.c_int]),
    "melspec_blm_precision": (C.c_int, [_vp]),
    "melspec_stft_bins": (C.c_size_t, [_vp, C.c_int]),
    "melspec_stft_uniform_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, C.c_int, C.c_int, _vp]),
    "melspec_stft_ragged_device": (C.c_int, [_vp, _vp, _u64p, _u64p, C.c_uint32, _vp, _u64p, C.c_int, C.c_int, _vp]),
    "melspec_stft_host": (C.c_int, [_vp, _f32p, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "melspec_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "melspec_host_free": (C.c_int, [_vp]),
    "melspec_shard_by_samples": (C.c_int, [_u64p, C.c_uint32, C.c_int, _u32p]),
    "melspec_sharded_create": (C.c_int, [C.POINTER(_vp), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
    "melspec_sharded_destroy": (None, [_vp]),
    "melspec_sharded_n_shards": (C.c_int, [_vp]),
    "melspec_sharded_ctx": (_vp, [_vp, C.c_int]),
    "melspec_sharded_compute_batch_host": (C.c_int, [_vp, _f32p, _u64p, _u64p, C.c_uint32, _f32p, _u64p, C.c_size_t, _u64p]),
    "melspec_sharded_compute_uniform_device": (C.c_int, [_vp, C.POINTER(_vp), C.c_uint64, C.c_uint64, _u32p, C.POINTER(_vp)]),
    "melspec_sharded_compute_ragged_device": (C.c_int, [_vp, C.POINTER(_vp), _u64p, _u64p, _u32p, C.POINTER(_vp), _u64p]),
    "melspec_sharded_synchronize": (C.c_int, [_vp]),
    "melspec_gather_peer": (C.c_int, [C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(_vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int]),
    "melspec_compute_uniform_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, _vp]),
    "melspec_interleaved_width": (C.c_size_t, [_vp, C.c_size_t, C.c_size_t]),
    "melspec_compute_uniform_device_interleaved": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, C.c_int,
                                                             C.c_uint64, _vp]),
    "melspec_compute_ragged_device": (C.c_int, [_vp, _vp, _u64p, _u64p, C.c_uint32, _vp, _u64p, _vp]),
    "melspec_compute_ragged_device_desc": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint64, _vp]),
    "melspec_time_uniform_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, C.c_int, C.c_int, _f32p]),
    "melspec_time_first_kernel": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, C.c_int, C.c_int, _f32p]),
    "melspec_synchronize": (C.c_int, [_vp, _vp]),
    "melspec_mel_filterbank": (C.c_int, [C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _f64p]),
    "melspec_hann_window": (C.c_int, [C.c_int, _f64p]),
    "melspec_hz_to_mel": (C.c_double, [C.c_double, C.c_int]),
    "melspec_mel_to_hz": (C.c_double, [C.c_double, C.c_int]),
    "melspec_mel_frequencies": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]),
    "melspec_fft_frequencies": (C.c_int, [C.c_double, C.c_int, C.POINTER(C.c_double)]),
    "melspec_kaldi_mel_filterbank": (C.c_int, [C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, _f64p]),
    "melspec_fbank_default_config": (None, [C.POINTER(FbankConfigC)]),
    "melspec_fbank_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(FbankConfigC)]),
    "melspec_fbank_destroy": (None, [_vp]),
    "melspec_fbank_num_frames": (C.c_size_t, [_vp, C.c_size_t]),
    "melspec_fbank_num_mel_bins": (C.c_int, [_vp]),
    "melspec_fbank_uses_fast_path": (C.c_int, [_vp]),
    "melspec_fbank_use_generic": (C.c_int, [_vp, C.c_int]),
    "melspec_fbank_compute_host": (C.c_int, [_vp, _f32p, C.c_size_t, _f32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "melspec_fbank_compute_uniform_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, _vp]),
    "melspec_fbank_compute_uniform_device_split": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, _vp, _vp]),
    "melspec_fbank_compute_ragged_device": (C.c_int, [_vp, _vp, _u64p, _u64p, C.c_uint32, _vp, _u64p, _vp]),
    "melspec_fbank_compute_batch_host": (C.c_int, [_vp, _f32p, _u64p, _u64p, C.c_uint32, _f32p, _u64p, C.c_size_t, _u64p]),
    "melspec_fbank_compute_ragged_device_desc": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint64, _vp]),
    "melspec_fbank_synchronize": (C.c_int, [_vp, _vp]),
    "melspec_blm_default_config": (None, [C.POINTER(BlmConfigC)]),
    "melspec_blm_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(BlmConfigC)]),
    "melspec_blm_destroy": (None, [_vp]),
    "melspec_blm_num_frames": (C.c_size_t, [_vp, C.c_size_t]),
    "melspec_blm_padded_frames": (C.c_size_t, [_vp, C.c_size_t]),
    "melspec_blm_compute_host": (C.c_int, [_vp, _f32p, C.c_size_t, _f32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "melspec_blm_compute_uniform_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, _vp]),
    "melspec_blm_compute_ragged_device": (C.c_int, [_vp, _vp, _u64p, _u64p, C.c_uint32, _vp, _u64p, _vp]),
    "melspec_blm_compute_batch_host": (C.c_int, [_vp, _f32p, _u64p, _u64p, C.c_uint32, _f32p, _u64p, C.c_size_t, _u64p]),
    "melspec_blm_synchronize": (C.c_int, [_vp, _vp]),
    "melspec_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "melspec_free": (C.c_int, [_vp]),
    "melspec_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "melspec_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "melspec_device_synchronize": (C.c_int, []),
    "melspec_synth_pcm_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _vp]),
    "melspec_synth_pcm_window_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, _vp]),
    "melspec_stream_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_uint32, C.c_uint32]),
    "melspec_stream_destroy": (None, [_vp]),
    "melspec_stream_reset": (C.c_int, [_vp, _u32p, C.c_uint32]),
    "melspec_stream_frames_after": (C.c_size_t, [_vp, C.c_uint32, C.c_uint32]),
    "melspec_stream_push_host": (C.c_int, [_vp, _u32p, _f32p, _u32p, C.c_uint32, _f32p, C.c_size_t, _u32p]),
    "melspec_stream_push_host_stft": (C.c_int, [_vp, _u32p, _f32p, _u32p, C.c_uint32, _vp, C.c_size_t, _u32p, C.c_int, C.c_int]),
    "melspec_stream_flush_host": (C.c_int, [_vp, _u32p, C.c_uint32, _f32p, C.c_size_t, _u32p]),
    "melspec_stream_input_ptr": (_vp, [_vp, C.c_uint32]),
    "melspec_stream_push_device": (C.c_int, [_vp, _u32p, _u32p, C.c_uint32, _vp, _u64p, _u32p, _vp]),
    "melspec_stream_enable_vad": (C.c_int, [_vp, C.POINTER(VadSettingsC)]),
    "melspec_stream_vad_frames": (C.c_uint64, [_vp, C.c_uint32]),
    "melspec_stream_push_host_vad": (C.c_int, [_vp, _u32p, _f32p, _u32p, C.c_uint32, _f32p, C.c_size_t, _u32p, _vp, C.c_size_t]),
    "melspec_stream_flush_host_vad": (C.c_int, [_vp, _u32p, C.c_uint32, _f32p, C.c_size_t, _u32p, _vp, C.c_size_t]),
    "melspec_stream_push_device_vad": (C.c_int, [_vp, _u32p, _u32p, C.c_uint32, _vp, _u64p, _u32p, _vp, _vp]),
    "melspec_vad_default_settings": (None, [_vp]),
    "melspec_vad_mask_len": (C.c_size_t, [C.c_int, C.c_size_t]),
    "melspec_vad_boundaries_device": (C.c_int, [_vp, C.c_size_t, C.c_int, C.c_size_t, C.c_uint32, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "melspec_vad_boundaries_host": (C.c_int, [C.c_int, _f32p, C.c_int, C.c_size_t, _vp, _vp, _vp, _u32p]),
    "melspec_tga_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "melspec_tga_destroy": (None, [_vp]),
    "melspec_tga_layout": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "melspec_tga_encode_device": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_size_t, C.c_uint32, _vp, C.c_size_t, _vp]),
    "melspec_tga_encode_pcm_uniform_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, _vp, _vp, C.c_size_t, _vp]),
    "melspec_tga_decode_device": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_size_t, C.c_uint32, _vp, C.c_size_t, _vp]),
    "melspec_tga_encode_host": (C.c_int, [_vp, _f32p, C.c_size_t, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_uint32)]),
    "melspec_tga_decode_host": (C.c_int, [_vp, _vp, C.c_size_t, _f32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "melspec_quantize_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "melspec_dequantize_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "melspec_quantize_host": (C.c_int, [_vp, _f32p, C.c_size_t, _vp, _f32p]),
    "melspec_dequantize_host": (C.c_int, [_vp, _vp, C.c_size_t, _f32p, _f32p]),
    "melspec_tga_synchronize": (C.c_int, [_vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libmelspec_hip.so (built in-tree by mel_spec_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        path = os.environ.get("MELSPEC_LIB") or LIB_PATH   # tuning builds of the same ABI
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension has not been built "
                "(run `python -m mel_spec_amd.build`). mel_spec_amd has no CPU fallback.")
        L = C.CDLL(path)
        older = bool(os.environ.get("MELSPEC_LIB")) and bool(os.environ.get("MELSPEC_LIB_OLDER"))   # tools/ab_run.py against a build of an earlier round
        for name, (res, args) in SIGNATURES.items():
            if older and not hasattr(L, name):
                continue
            fn = getattr(L, name)   # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    msg = lib().melspec_last_error()
    return msg.decode("utf-8", "replace") if msg else ""
