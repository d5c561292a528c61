"""mel_spec_amd -- MI355X (gfx950) log-mel spectrogram frontend behind the API of
wavey-ai/mel-spec's GPU plugin slot.  All compute runs in hand-written HIP kernels
(csrc/) reached through the C ABI of libmelspec_hip.so (include/melspec_hip.h)."""
from .hip import (BatchLogMelConfig, BatchLogMelError, BatchLogMelSpectrogram, DeviceBuffer, HostBuffer, Fbank, FbankConfig, HipError, HipMelSpectrogram, HipRuntimeError, HipUnavailable, SparseMelFilterbank,
                  device_count, device_synchronize, fft_frequencies, hann_window, hz_to_mel, kaldi_mel_filterbank, mel, mel_frequencies, mel_to_hz,
                  mels_to_hz, synth_pcm_device, synth_pcm_window)
from .parallel import ShardedMelSpectrogram, gather_peer, shard_by_samples, shard_range
from .quant import QuantizationRange, TgaCodec, chunk_frames_into_strides, to_array2
from .stream import MelSpectrogram, RingBuffer, Spectrogram, StreamBank
from .vad import (DetectionSettings, EdgeInfo, VadFrameTiming, VoiceActivity, VoiceActivityDetector, VoiceActivityTimestamps, duration_ms_for_n_frames,
                  format_milliseconds, n_frames_for_duration, vad_boundaries, vad_on)

__all__ = ["BatchLogMelConfig", "BatchLogMelError", "BatchLogMelSpectrogram", "DeviceBuffer", "HostBuffer", "Fbank", "FbankConfig", "HipError", "HipMelSpectrogram", "HipRuntimeError", "SparseMelFilterbank",
           "HipUnavailable", "device_count", "device_synchronize", "hann_window", "kaldi_mel_filterbank", "mel", "hz_to_mel", "mel_to_hz", "mels_to_hz", "mel_frequencies", "fft_frequencies",
           "synth_pcm_device", "synth_pcm_window", "shard_range", "shard_by_samples", "ShardedMelSpectrogram", "gather_peer", "QuantizationRange", "TgaCodec", "to_array2", "chunk_frames_into_strides", "RingBuffer", "StreamBank", "Spectrogram", "MelSpectrogram", "DetectionSettings", "EdgeInfo", "VoiceActivity", "VoiceActivityDetector", "VadFrameTiming", "VoiceActivityTimestamps", "n_frames_for_duration", "duration_ms_for_n_frames", "format_milliseconds",
           "vad_boundaries", "vad_on"]
