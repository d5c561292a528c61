// melspec_kernels.hpp -- every gfx950 kernel of libmelspec_hip.so in one include, for tools that want them all (the library itself
// compiles one family per translation unit: whisper400.hip, fbank512.hip, pow2.hip, aux.hip -- see mel_spec_amd/build.py).
//
//   kernels_common.hpp       batch description, unit -> clip mapping, statistics sink + vote, wave helpers, RoundSync
//   whisper400_kernels.hpp   fused n_fft = 400 log-mel: whisper400_{wave,six}[_runs]_kernel (f32 + guard), whisper400_precise_kernel,
//                            whisper400_six64[_layout]_kernel (f64), whisper400_stft_kernel
//   fbank512_kernels.hpp     fused 512-point family: fbank512_wave_kernel (Kaldi / NeMo / Whisper-512, f64 and f32),
//                            fbank512_clip_kernel, cmn_kernel, blm_normalize[_ragged]_kernel
//   generic_kernels.hpp      generic_frame_kernel, generic_stft_kernel, pow2_frame_kernel, mel_stage[_jobs]_kernel (f64)
//   aux_kernels.hpp          stream_scatter / stream_carry, plan_ragged_device_kernel, synth_pcm_kernel
#pragma once
#include "kernels_common.hpp"
#include "whisper400_kernels.hpp"
#include "fbank512_kernels.hpp"
#include "generic_kernels.hpp"
#include "aux_kernels.hpp"
