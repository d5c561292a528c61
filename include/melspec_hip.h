/*
 * melspec_hip.h -- C ABI of libmelspec_hip.so, the MI355X (gfx950) log-mel frontend.
 *
 * This is the drop-in boundary for the reference's GPU plugin slot.  Each entry point
 * names the reference interface it replaces (paths relative to wavey-ai/mel-spec v0.4.0).
 * The reference binds its CUDA plugin per *stage* (cufftExecZ2Z + launch_mel_kernel,
 * src/cuda.rs:185-219, src/cuda_kernels.cu:49-66); this library is bound per *batch*:
 * one call = framing + Hann + FFT + power + mel + log10 + per-frame normalisation.
 *
 * Conventions (mirroring src/cuda.rs:10-25,164 and src/cuda_kernels.cu:56-58):
 *   - every function returns int; 0 == success (CUDA_SUCCESS analogue);
 *     > 0  == a hipError_t raised by the runtime (maps to CudaError::Runtime);
 *     < 0  == a library code below (INVALID_ARG / UNAVAILABLE map to
 *             CudaError::Unavailable when raised by *_create, Runtime otherwise);
 *   - nothing throws or aborts across the ABI;
 *   - plain pointers and sizes only; the caller owns every buffer it passes in;
 *     the context owns its device tables, scratch, pinned staging and stream;
 *   - a context is single-threaded (one per host thread per device), like
 *     `&mut self` + raw device pointers in CudaMelSpectrogram (src/cuda.rs:27-36);
 *   - *_host calls are synchronous; *_device calls are stream-ordered and asynchronous.
 */
#ifndef MELSPEC_HIP_H
#define MELSPEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MELSPEC_OK                 0
#define MELSPEC_ERR_INVALID_ARG   (-1)  /* zero/oversized sizes, NULL pointers (src/cuda.rs:45-49) */
#define MELSPEC_ERR_UNAVAILABLE   (-2)  /* no HIP device / not gfx950 / runtime missing (tests self-skip, src/cuda.rs:512-518) */
#define MELSPEC_ERR_CAPACITY      (-3)  /* caller's output buffer is too small */
#define MELSPEC_ERR_UNSUPPORTED   (-4)  /* geometry outside what the kernels cover */
#define MELSPEC_ERR_INTERNAL      (-5)

typedef struct melspec_ctx melspec_ctx;      /* Whisper-style log-mel (HipMelSpectrogram) */
typedef struct melspec_sharded melspec_sharded;   /* one melspec_ctx per GPU of the node, clips split per device */
typedef struct melspec_fbank melspec_fbank;  /* Kaldi-style fbank (Fbank)                  */

/* ---- library / device ------------------------------------------------------------- */

/* ABI version of this header (bumped on incompatible change). */
int melspec_abi_version(void);
/* Hash of the sources this library was built from (mel_spec_amd/build.py: the .hip / .hpp files of csrc/ and the headers of include/), "unknown" for a
 * build made by hand.  The test harness rebuilds when it differs from the hash of the checkout, so a stale prebuilt library cannot pass. */
const char *melspec_source_hash(void);
/* Number of usable gfx950 devices, or MELSPEC_ERR_UNAVAILABLE. */
int melspec_device_count(void);
/* Message for the last failure on this thread (valid until the next failing call).
 * Counterpart of the String inside CudaError::Runtime (src/cuda.rs:10-25). */
const char *melspec_last_error(void);

/* ---- Whisper log-mel: replaces CudaMelSpectrogram (src/cuda.rs:27-140) ------------- */

/* CudaMelSpectrogram::new(fft_size, hop_size, sampling_rate, n_mels) (src/cuda.rs:39-82).
 * Builds the periodic Hann window (src/stft.rs:141-145), the Slaney filterbank
 * mel(sr, fft, n_mels, None, None, false, true) (src/mel.rs:19-24,547-589) and FFT
 * twiddles in f64 on the host, uploads them as f32 tables.  device < 0 -> current device. */
int melspec_create(melspec_ctx **out, int device, int fft_size, int hop_size,
                   double sampling_rate, int n_mels);
/* MelSpectrogram with a caller-chosen filterbank instead of new()'s mel(sr, fft, n_mels, None, None, false, true) (src/mel.rs:19-24):
 * SparseMelFilterbank::from_mel(sr, n_fft, n_mels, f_min, f_max, htk, norm) (src/mel.rs:73-87; f_min < 0 == None, f_max <= 0 == None)
 * or from_dense(filters) (src/mel.rs:48-71; row-major [n_mels][fft_size / 2 + 1] f64) -- what log_mel_spectrogram(stft, mel_filters)
 * (src/mel.rs:436-441) takes.  Everything else as melspec_create; a two-filters-per-bin bank (every triangular one) of <= 131 rows
 * runs on the fused n_fft = 400 / 512 kernels, any other matrix on the generic kernel. */
int melspec_create_with_filterbank(melspec_ctx **out, int device, int fft_size, int hop_size, double sampling_rate, int n_mels,
                                   double f_min, double f_max, int htk, int norm);
int melspec_create_with_dense_filterbank(melspec_ctx **out, int device, int fft_size, int hop_size, double sampling_rate, int n_mels,
                                         const double *filters, int fft_bins);
void melspec_destroy(melspec_ctx *ctx);   /* Drop (src/cuda.rs:142-148,366-375) */

/* frame_windows' count: len < fft ? 0 : (len - fft)/hop + 1 (src/stft.rs:153-157). */
size_t melspec_num_frames(const melspec_ctx *ctx, size_t n_samples);
/* CudaMelSpectrogram::max_frames_per_batch (src/cuda.rs:84-86, 150-155: the reference chunks at 8192 frames / 64 MiB): frames
 * per chunk of the host pipeline here (16 MiB of PCM: 26 212 at 400/160).  The device entry points take any batch in one launch. */
size_t melspec_max_frames_per_batch(const melspec_ctx *ctx);
int melspec_fft_size(const melspec_ctx *ctx);
int melspec_hop_size(const melspec_ctx *ctx);
int melspec_n_mels(const melspec_ctx *ctx);
/* 1 if this geometry runs on the fused FFT kernel, 0 if on the generic DFT kernel. */
int melspec_uses_fast_path(const melspec_ctx *ctx);

/* Arithmetic of the fused n_fft = 400 kernels (the reference computes in f64, src/stft.rs:98-111, and its CUDA plugin runs
 * an f64 FFT too, cufftExecZ2Z src/cuda.rs:204-219).  Every mode but F32 is within 1e-4 of the f64 reference on every input.
 *   AUTO (default)  f32 FFT; in the same pass every frame is checked against an empirical error bound -- a mel band within two
 *                   decades of the per-frame clamp (max - 8, src/mel.rs:645-654) is where the f32 FFT's rounding noise can
 *                   exceed 1e-4 (calibrated with tools/flag_calib.py / flag_calib2.py, soaked by tools/fuzz_gpu.py) -- and a
 *                   frame that fails it is recomputed in f64 by the wavefront that owns it (same launch).  Noise-like input
 *                   never takes that branch (the bench workload runs at the f32 rate).  Speech and tonal material trip the
 *                   guard on 40-100 % of their frames, which is slower than computing everything in f64 -- so the launch takes a
 *                   vote first: the first work unit of every wavefront is the sample, and when more than 1/8 of the sampled
 *                   frames trip the guard the f32 kernel stands down and the f64 kernel queued behind it (a second launch that
 *                   returns at once otherwise) computes the whole batch at the F64 rate.  Plain [clip][frame][mel] batches,
 *                   uniform and ragged, and the padded / mel-major layouts (their sample is the head of the batch); only
 *                   melspec_tga_encode_pcm_uniform_device keeps the f32 kernel + recompute whatever the input.  The vote happens inside
 *                   the batch's own launch and nobody waits for it: the result of a call is a function of its input alone (same
 *                   batch -> same bits, whatever the context computed before; round 3 chose from the previous batch's
 *                   statistics).  A clip's bits can differ between two different batches (f32 regime in one, f64 in the other,
 *                   both within 1e-4); F32 and F64 do not have that, nor AUTO after melspec_set_auto_adaptive(ctx, 0).
 *   F64             window, FFT and |X|^2 in f64 for every frame: ~4e-7 from the reference, about 60 % of the f32 rate.
 *   F32             the f32 kernel alone: ~3e-5 on speech and noise, up to ~5e-4 on a line over a floor 70..90 dB down.
 * Geometries on the generic kernels always compute in f64.  The fused n_fft = 512 kernels (Whisper flavour; the 80- and 128-mel banks at
 * 16 kHz have an f32 instantiation) follow the same three modes since round 6:
 *   AUTO  plain [clip][frame][mel] batches, uniform and ragged, of at least two work units per f32 wave of the grid (6144 units = 24 576
 *         frames on an MI355X) run the f32 kernel with the same guard and the same vote; the f64 kernel queued behind it computes the
 *         whole batch on "heavy" and the units the f32 launch noted on "light" (this family has no in-kernel recompute).  Smaller
 *         batches, the padded / mel-major layouts, other banks, and AUTO after melspec_set_auto_adaptive(ctx, 0): the f64 kernel.
 *   F64   the f64 kernel.
 *   F32   the f32 kernel alone, no guard: 2e-5 on speech, ~1e-6 on noise, up to 4e-4 on a line over a floor 70..90 dB down
 *         (profiles/r06_guard512.txt; round 5's f32 instantiation split straight to powers and was 7e-2 off on speech: it now splits to
 *         amplitudes like the n_fft = 400 kernels).
 * melspec_precision() reports the mode in effect (AUTO only where sizeable plain batches do vote, else F64 / F32). */
#define MELSPEC_PRECISION_AUTO 0
#define MELSPEC_PRECISION_F64  1
#define MELSPEC_PRECISION_F32  2
int melspec_set_precision(melspec_ctx *ctx, int mode);
int melspec_precision(const melspec_ctx *ctx);
/* melspec_set_precision(ctx, on ? F64 : AUTO); is_precise: 1 when every frame is computed in f64. */
int melspec_set_precise(melspec_ctx *ctx, int on);
int melspec_is_precise(const melspec_ctx *ctx);
/* Name of the kernel(s) a plain [clip][frame][mel] batch of this context runs on (for profiles and bench lines). */
const char *melspec_plain_kernel_name(const melspec_ctx *ctx);
/* Frames that tripped AUTO's guard since the context was created (f32 kernel: recomputed in f64; f64 kernel: counted only).
 * Synchronises the device. */
int melspec_guard_count(melspec_ctx *ctx, uint64_t *frames);
/* AUTO's vote (default on; 0: the f32 kernel + per-frame recompute whatever the input).  melspec_auto_state reports, without
 * synchronising: *heavy = 1 when the last FINISHED AUTO batch ran on the f64 kernel, *fraction = the fraction of its frames that tripped
 * the guard (batches of >= 256 frames).  Reporting only: nothing is decided from it. */
int melspec_set_auto_adaptive(melspec_ctx *ctx, int on);
int melspec_auto_state(melspec_ctx *ctx, int *heavy, double *fraction);

/* compute_mel_spectrogram(&mut self, samples: &[f32]) -> Vec<Vec<f32>> (src/cuda.rs:88-101)
 * == Spectrogram::compute_mel_spectrogram_cpu (src/stft.rs:119-138) on the GPU.
 * Host PCM in, host [frames][n_mels] f32 out (row-major, src/stft.rs:133-134).
 * Empty/short input -> *n_frames = 0, MELSPEC_OK (src/cuda.rs:91-93). Synchronous. */
int melspec_compute_host(melspec_ctx *ctx, const float *samples, size_t n_samples,
                         float *out, size_t out_capacity_floats, size_t *n_frames);

/* Additive surface: many host clips in one call.  Clip i = samples + offsets[i] (lengths[i] samples); its frames go to
 * out + out_offsets[i] floats (NULL: packed back to back in clip order); *total_frames = frames of all clips.
 * The call is cut into ~16 MiB chunks whose upload, kernels and download overlap on three streams (the reference plugin
 * chunks at 8192 frames with a synchronise per chunk, src/cuda.rs:150-155,343-351); memory from melspec_host_alloc (or
 * any pinned memory) is DMA'd in place, pageable memory is staged through pinned buffers by helper threads.
 * melspec_compute_host takes the same pipeline for clips longer than ~8 MB.  Synchronous. */
int melspec_compute_batch_host(melspec_ctx *ctx, const float *samples, const uint64_t *offsets, const uint64_t *lengths,
                               uint32_t n_clips, float *out, const uint64_t *out_offsets, size_t out_capacity_floats,
                               uint64_t *total_frames);
/* ---- STFT export: Spectrogram::compute_all_cpu (src/stft.rs:89-115) ---------------------------------------------------
 * The complex spectrum of every frame instead of its mel row (callers such as examples/vad_ten_eval/src/main.rs:232-256 feed
 * it to dense mel code of their own).  Always computed in f64 like the reference (frame_windows + forward FFT); stored as
 * interleaved (re, im) pairs of float (MELSPEC_STFT_F32) or double (MELSPEC_STFT_F64), `bins` per frame:
 * full == 0: n_fft/2 + 1 (the input is real); full != 0: n_fft, the reference's Vec<Complex<f64>> layout, upper half =
 * conjugate mirror.  d_out = [clip][frame][bins]; out offsets / capacities count complex elements. */
#define MELSPEC_STFT_F32 0
#define MELSPEC_STFT_F64 1
size_t melspec_stft_bins(const melspec_ctx *ctx, int full);
int melspec_stft_uniform_device(melspec_ctx *ctx, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len, uint32_t n_clips,
                                void *d_out, int dtype, int full, void *stream);
int melspec_stft_ragged_device(melspec_ctx *ctx, const float *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths,
                               uint32_t n_clips, void *d_out, const uint64_t *h_out_offsets, int dtype, int full, void *stream);
/* compute_all_cpu(samples) on the GPU: host PCM in, host [frames][bins] complex out.  Synchronous. */
int melspec_stft_host(melspec_ctx *ctx, const float *samples, size_t n_samples, void *out, size_t out_capacity_complex,
                      int dtype, int full, size_t *n_frames);

/* ---- the mel stage on its own: MelSpectrogram::add(&fft) (src/mel.rs:13-32) ---------------------------------------------
 * For callers that hold complex STFT frames -- melspec_stft_*'s, or their own (the reference's split API: Spectrogram::add then
 * MelSpectrogram::add): per frame project_stft_log10 (src/mel.rs:148-168: |X[bin]|^2 through the sparse Slaney bank, bins >= n_fft/2
 * contribute nothing, log10(max(E, 1e-10))) and norm_mel_slice_f64 (src/mel.rs:645-654), in f64 like the reference; [frame][n_mels]
 * f32 out.  spec = [frame][bins] interleaved (re, im), dtype / full as in the STFT export (full: n_fft complex per frame, else
 * n_fft/2 + 1). */
int melspec_mel_from_stft_device(melspec_ctx *ctx, const void *d_spec, int dtype, int full, uint64_t n_frames, float *d_out, void *stream);
int melspec_mel_from_stft_host(melspec_ctx *ctx, const void *spec, int dtype, int full, size_t n_frames, float *out, size_t out_capacity_floats);

/* The context's scratch only grows (pipeline buffers sized by the largest chunk, the precision guard's queue of 4 B per frame of
 * the largest batch, ragged plans): this waits for the context's queued work and gives all of it back.  The next call re-allocates. */
int melspec_release_scratch(melspec_ctx *ctx);
/* Pinned host memory (cudaMallocHost of src/cuda.rs:185-199): buffers from here skip the staging copy of the host calls. */
int melspec_host_alloc(void **p, size_t bytes);
int melspec_host_free(void *p);

/* ---- the GPUs of one node (additive: the reference binds one device, src/cuda.rs:246-247) ----------------------------
 * Every frame (Whisper) is independent, so clips are split into contiguous blocks per device, balanced by samples, and no
 * collective touches the data.  bounds[k] .. bounds[k+1] = the clips of shard k, k < n_shards. */
int melspec_shard_by_samples(const uint64_t *lengths, uint32_t n_clips, int n_shards, uint32_t *bounds /* [n_shards + 1] */);
/* One context + stream per listed device (devices == NULL: every gfx950 device; a device may be listed more than once). */
int melspec_sharded_create(melspec_sharded **out, const int *devices, int n_devices, int fft_size, int hop_size,
                           double sampling_rate, int n_mels);
void melspec_sharded_destroy(melspec_sharded *s);
int melspec_sharded_n_shards(const melspec_sharded *s);
melspec_ctx *melspec_sharded_ctx(melspec_sharded *s, int shard);   /* for device-resident work on one shard; owned by s */
/* melspec_compute_batch_host over all devices: one host thread per device drives that device's pipeline on its block of
 * clips; every shard writes its own part of `out`.  Synchronous. */
int melspec_sharded_compute_batch_host(melspec_sharded *s, const float *samples, const uint64_t *offsets, const uint64_t *lengths,
                                       uint32_t n_clips, float *out, const uint64_t *out_offsets, size_t out_capacity_floats,
                                       uint64_t *total_frames);
/* Device-resident shards: shard k's clips are on device k (d_pcm[k]), its frames stay there (d_out[k]); n_clips[k] clips per shard.
 * uniform: clip c of shard k = d_pcm[k] + c * clip_stride.  ragged: the clip tables of the shards one after the other (n_clips[0]
 * entries, then n_clips[1], ...), offsets relative to the shard's own d_pcm[k] / d_out[k]; h_out_offsets NULL = packed per shard.
 * Stream-ordered on every shard's context stream; melspec_sharded_synchronize waits for all shards.  No data crosses a device
 * boundary (per-clip split, no collective). */
int melspec_sharded_compute_uniform_device(melspec_sharded *s, const float *const *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                           const uint32_t *n_clips, float *const *d_out);
int melspec_sharded_compute_ragged_device(melspec_sharded *s, const float *const *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths,
                                          const uint32_t *n_clips, float *const *d_out, const uint64_t *h_out_offsets);
int melspec_sharded_synchronize(melspec_sharded *s);
/* Optional consolidation of device-resident results on one device: bytes[i] bytes at srcs[i] (device src_devices[i]) ->
 * dst + dst_offsets[i] bytes on dst_device, each piece on a stream of its source device (hipMemcpyPeerAsync: the pieces
 * cross their own xGMI links concurrently).  Synchronous; time it separately from the frames/s figure. */
int melspec_gather_peer(int dst_device, void *dst, const int *src_devices, const void *const *srcs, const size_t *bytes,
                        const size_t *dst_offsets, int n);

/* Additive surface (the reference has no multi-clip call): many equal-length clips in
 * one launch, PCM and output resident in HBM.  Clip c = d_pcm[c*clip_stride .. +clip_len).
 * d_out = [clip][frame][mel] f32, frames = melspec_num_frames(clip_len).
 * stream: hipStream_t (NULL = the context's own stream).  Asynchronous. */
int melspec_compute_uniform_device(melspec_ctx *ctx, const float *d_pcm, uint64_t clip_stride,
                                   uint64_t clip_len, uint32_t n_clips, float *d_out, void *stream);

/* interleave_frames(frames, major_column_order, min_width) (src/mel.rs:480-544) fused into the
 * store: per clip the output is n_mels x W floats, W = melspec_interleaved_width(): the frame count,
 * +1 zero column if it is odd and min_width > 0 (whisper.cpp needs an even width), then zero columns up
 * to min_width (must be even).  major_column_order == 0 -> [mel][W] rows (what whisper.cpp's set_mel
 * takes); != 0 -> [W][mel] rows.  Clips with no frame are an error, like the reference's assert. */
size_t melspec_interleaved_width(const melspec_ctx *ctx, size_t n_samples, size_t min_width);
int melspec_compute_uniform_device_interleaved(melspec_ctx *ctx, const float *d_pcm, uint64_t clip_stride,
                                               uint64_t clip_len, uint32_t n_clips, float *d_out,
                                               int major_column_order, uint64_t min_width, void *stream);

/* Ragged batch: clip c = d_pcm[h_offsets[c] .. + h_lengths[c]) (sample units, host arrays).
 * Output of clip c starts at d_out + h_out_offsets[c] (float units); pass NULL to pack the
 * clips back to back in order.  Clips shorter than fft_size produce zero frames. */
int melspec_compute_ragged_device(melspec_ctx *ctx, const float *d_pcm, const uint64_t *h_offsets,
                                  const uint64_t *h_lengths, uint32_t n_clips, float *d_out,
                                  const uint64_t *h_out_offsets, void *stream);

/* The same batch with its clip table in DEVICE memory (a segmenter or VAD on the GPU produces it; nothing is copied back):
 * d_offsets / d_lengths in samples, d_out_offsets in floats (NULL: the clips' frames packed back to back in clip order).  The plan
 * the host would build is built by a kernel on `stream`.  max_total_frames: an upper bound of the frames of all clips together (the
 * capacity of d_out in frames) -- it sizes the launch and the scratch; the true count stays on the device.  Asynchronous. */
int melspec_compute_ragged_device_desc(melspec_ctx *ctx, const float *d_pcm, const uint64_t *d_offsets, const uint64_t *d_lengths,
                                       uint32_t n_clips, float *d_out, const uint64_t *d_out_offsets, uint64_t max_total_frames,
                                       void *stream);

/* Benchmark helper (the reference's #[ignore] Instant-timed benches, src/cuda.rs:547-613): runs
 * `warmup` untimed and `iters` timed melspec_compute_uniform_device calls on the context's own
 * stream between two HIP events and returns the average milliseconds per call. */
int melspec_time_uniform_device(melspec_ctx *ctx, const float *d_pcm, uint64_t clip_stride,
                                uint64_t clip_len, uint32_t n_clips, float *d_out,
                                int warmup, int iters, float *avg_ms);

/* The same calls with a HIP event pair around the FIRST kernel of every call -- in AUTO the f32 kernel, without the gated f64 launch
 * behind it -- and the average of those pairs: the launch duration of the dominant kernel, as a profiler's kernel trace reports it
 * (bench.py's roofline.achieved).  Fused f32 n_fft = 400 contexts only. */
int melspec_time_first_kernel(melspec_ctx *ctx, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len, uint32_t n_clips, float *d_out,
                              int warmup, int iters, float *avg_first_kernel_ms);

/* Wait for everything this context has queued on `stream` (cudaStreamSynchronize, src/cuda.rs:129). */
int melspec_synchronize(melspec_ctx *ctx, void *stream);

/* ---- the stand-alone mel helpers: SparseMelFilterbank, log_mel_spectrogram, norm_mel (src/mel.rs:40-168, 436-469) ----------
 * For callers that keep the reference's split API (examples/vad_ten_eval/src/main.rs:232-256: compute_all_cpu -> their own bank).
 * dtype: MELSPEC_STFT_F32 / MELSPEC_STFT_F64 (element type of the arrays).  The sums are the reference's left folds over its sparse
 * rows with a separate multiply and add: project_power is bit-exact against the f32 / f64 reference arithmetic. */
typedef struct melspec_bank melspec_bank;
/* SparseMelFilterbank::from_dense (src/mel.rs:48-71) / from_mel (:73-87) */
int melspec_bank_from_dense(melspec_bank **out, int device, const double *filters, int n_mels, int fft_bins);
int melspec_bank_from_mel(melspec_bank **out, int device, double sample_rate, int n_fft, int n_mels, double f_min, double f_max,
                          int htk, int norm);
void melspec_bank_destroy(melspec_bank *bank);
int melspec_bank_n_mels(const melspec_bank *bank);              /* n_mels()            src/mel.rs:89-91  */
int melspec_bank_fft_bins(const melspec_bank *bank);            /* fft_bins()          src/mel.rs:93-95  */
int melspec_bank_non_zero_weights(const melspec_bank *bank);    /* non_zero_weights()  src/mel.rs:97-99  */
/* weights_for_mel(mel_idx) (src/mel.rs:102-104): the row's SparseMelWeight { bin, weight } entries in ascending bin order, at most
 * `capacity` of them copied; returns the row's entry count (-1: bad index).  bins / weights may be NULL. */
int melspec_bank_weights_for_mel(const melspec_bank *bank, int mel_idx, int *bins, double *weights, int capacity);
/* project_power_f64 / project_power_f32 (src/mel.rs:106-146) for n_frames rows: power [n_frames][fft_bins] -> [n_frames][n_mels] */
int melspec_bank_project_power_device(melspec_bank *bank, const void *d_power, int dtype, uint64_t n_frames, void *d_out, void *stream);
int melspec_bank_project_power_host(melspec_bank *bank, const void *power, int dtype, size_t n_frames, void *out);
/* log_mel_spectrogram(stft, mel_filters) (src/mel.rs:436-441 -> project_stft_log10 :148-168) for n_frames frames: complex frames
 * [n_frames][n_fft] (interleaved re, im; what compute_all_cpu / melspec_stft_* with full = 1 return) -> log10(max(E, 1e-10)),
 * [n_frames][n_mels] f64, NOT normalised (norm_mel below). */
int melspec_bank_log_mel_device(melspec_bank *bank, const void *d_stft, int dtype, int n_fft, uint64_t n_frames, double *d_out, void *stream);
int melspec_bank_log_mel_host(melspec_bank *bank, const void *stft, int dtype, int n_fft, size_t n_frames, double *out);
/* norm_mel (f64, src/mel.rs:448-454) / norm_mel_vec (f32, :457-469): ONE maximum over the n_values given (a frame, a window of
 * frames, a whole clip: "flexibility in the sample size that's normalised over"), then (max(x, mmax - 8) + 4) / 4. */
int melspec_bank_norm_mel_device(melspec_bank *bank, const void *d_in, int dtype, uint64_t n_values, void *d_out, void *stream);
int melspec_bank_norm_mel_host(melspec_bank *bank, const void *in, int dtype, size_t n_values, void *out);

/* ---- host-side table builders (pure CPU, usable without a GPU) --------------------- */

/* mel(sr, n_fft, n_mels, f_min, f_max, htk, norm) (src/mel.rs:547-589): dense row-major
 * [n_mels][n_fft/2+1] f64.  f_min < 0 == None (0 Hz); f_max <= 0 == None (sr/2).
 * Pinned by testdata/mel_filters.npz @1e-7 (src/mel.rs:838-850). */
int melspec_mel_filterbank(double sr, int n_fft, int n_mels, double f_min, double f_max,
                           int htk, int norm, double *out);
/* hann_window (src/stft.rs:141-145), periodic, f64. */
int melspec_hann_window(int n, double *out);
/* hz_to_mel / mel_to_hz / mel_frequencies / fft_frequencies (src/mel.rs:591-643): the scale the filterbank above is built on
 * (Slaney, or HTK when htk != 0); mel_frequencies fills n_mels values, fft_frequencies n_fft/2 + 1.  Host only. */
double melspec_hz_to_mel(double frequency, int htk);
double melspec_mel_to_hz(double mel, int htk);
int melspec_mel_frequencies(int n_mels, double fmin, double fmax, int htk, double *out);
int melspec_fft_frequencies(double sr, int n_fft, double *out);
/* kaldi_mel_filterbank (src/fbank.rs:253-301): dense [num_mel_bins][fft_size/2+1] f64. */
int melspec_kaldi_mel_filterbank(double sample_rate, int fft_size, int num_mel_bins,
                                 double low_freq, double high_freq, double *out);

/* ---- Kaldi fbank: replaces Fbank (src/fbank.rs:85-247) ------------------------------ */

/* FbankConfig (src/fbank.rs:25-64); `dither` and `use_energy` exist in the reference
 * struct but are never read by Fbank::compute, so they are not carried. */
typedef struct melspec_fbank_config {
    double sample_rate;       /* 16000.0 */
    int32_t num_mel_bins;     /* 80 */
    double frame_length_ms;   /* 25.0 */
    double frame_shift_ms;    /* 10.0 */
    double energy_floor;      /* 0.0 -> FLT_EPSILON (src/fbank.rs:210-214) */
    int32_t use_log_fbank;    /* 1 */
    int32_t use_power;        /* 1 */
    double preemphasis;       /* 0.97 */
    int32_t apply_cmn;        /* 1 */
    double low_freq;          /* 20.0 */
    double high_freq;         /* 0.0 == Nyquist */
} melspec_fbank_config;

void melspec_fbank_default_config(melspec_fbank_config *cfg);          /* FbankConfig::default (src/fbank.rs:46-64) */
int melspec_fbank_create(melspec_fbank **out, int device, const melspec_fbank_config *cfg); /* Fbank::new (src/fbank.rs:94-132) */
void melspec_fbank_destroy(melspec_fbank *fb);
size_t melspec_fbank_num_frames(const melspec_fbank *fb, size_t n_samples); /* src/fbank.rs:147-151 */
int melspec_fbank_num_mel_bins(const melspec_fbank *fb);
/* 1 if this configuration runs on the fused 512-point kernel, 0 if on the generic f64 kernel. */
int melspec_fbank_uses_fast_path(const melspec_fbank *fb);
/* on != 0: run this object on the any-geometry f64 path (an independent device path, kept as the on-device cross-check of the fused
 * kernel): 1 = whatever that path picks for the geometry (power-of-two frame sizes: pow2_frame_kernel), 2 = the workgroup-per-frame
 * kernel whatever the geometry (~60x slower than the fused kernel). */
int melspec_fbank_use_generic(melspec_fbank *fb, int on);
/* Fbank::compute(&self, samples) -> Array2<f32> (frames, num_mel_bins) (src/fbank.rs:141-236). */
int melspec_fbank_compute_host(melspec_fbank *fb, const float *samples, size_t n_samples,
                               float *out, size_t out_capacity_floats, size_t *n_frames);
/* Many equal-length clips, device resident; CMN (src/fbank.rs:224-233) is per clip. */
int melspec_fbank_compute_uniform_device(melspec_fbank *fb, const float *d_pcm, uint64_t clip_stride,
                                         uint64_t clip_len, uint32_t n_clips, float *d_out, void *stream);
/* Additive (round 6; the reference has one output contract, Fbank::compute, src/fbank.rs:141-236): the same batch as TWO outputs --
 * d_rows [clip][frame][num_mel_bins] as they are BEFORE the CMN of src/fbank.rs:224-233, and d_means [clip][num_mel_bins], the column
 * means that CMN subtracts (the fixed summation tree of the fused path: d_rows[c][f][m] - d_means[c][m] in f32 is, bit for bit, what
 * melspec_fbank_compute_uniform_device stores).  For a consumer that folds the subtraction into its own first read of the rows: the CMN's
 * second pass over them is a third of the fused kernel's memory traffic (profiles/r06_fbank_split.txt).  Needs FbankConfig::apply_cmn
 * (without it there are no means: MELSPEC_ERR_INVALID_ARG). */
int melspec_fbank_compute_uniform_device_split(melspec_fbank *fb, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                               uint32_t n_clips, float *d_rows, float *d_means, void *stream);
/* Fbank::compute for clips of any length in one launch (src/fbank.rs:141 is per clip): host or device clip tables, as
 * melspec_compute_ragged_device / _desc; offsets of the output in floats; CMN per clip. */
int melspec_fbank_compute_ragged_device(melspec_fbank *fb, const float *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths,
                                        uint32_t n_clips, float *d_out, const uint64_t *h_out_offsets, void *stream);
int melspec_fbank_compute_ragged_device_desc(melspec_fbank *fb, const float *d_pcm, const uint64_t *d_offsets, const uint64_t *d_lengths,
                                             uint32_t n_clips, float *d_out, const uint64_t *d_out_offsets, uint64_t max_total_frames,
                                             void *stream);
/* Many host clips in one call (the reference calls Fbank::compute once per clip, src/fbank.rs:141): clip i = samples[offsets[i] ..
 * + lengths[i]) -> its [frames_i][num_mel_bins] rows at out + out_offsets[i] floats (NULL: packed in clip order).  Whole clips in
 * chunks through the pinned, double-buffered host pipeline (as melspec_compute_batch_host); pinned samples / out are used in place. */
int melspec_fbank_compute_batch_host(melspec_fbank *fb, const float *samples, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_clips,
                                     float *out, const uint64_t *out_offsets, size_t out_capacity_floats, uint64_t *total_frames);
/* As melspec_release_scratch: waits for the object's own stream and gives its grow-only scratch back (pipeline and staging buffers,
 * ragged plans); work queued on caller streams must have been synchronised by the caller. */
int melspec_fbank_release_scratch(melspec_fbank *fb);
int melspec_fbank_synchronize(melspec_fbank *fb, void *stream);

/* ---- NeMo/Parakeet log-mel frontend: replaces BatchLogMelSpectrogram (src/mel.rs:171-418) ------- */

/* BatchLogMelConfig (src/mel.rs:171-208). */
typedef struct melspec_blm_config {
    int32_t sample_rate;            /* 16000 */
    int32_t n_fft;                  /* 512 */
    int32_t win_length;             /* 400 */
    int32_t hop_length;             /* 160 */
    int32_t n_mels;                 /* 80 */
    double f_min;                   /* 0.0 */
    double f_max;                   /* <= 0 == None (sample_rate / 2) */
    int32_t htk;                    /* 0 */
    int32_t norm;                   /* 1 */
    float preemphasis;              /* 0.0 (NeMo: 0.97) */
    int32_t center;                 /* 1 */
    float log_zero_guard;           /* f32::EPSILON (NeMo: 2^-24) */
    int32_t pad_to;                 /* 0 */
    int32_t normalize_per_feature;  /* 0 */
} melspec_blm_config;

typedef struct melspec_blm melspec_blm;
void melspec_blm_default_config(melspec_blm_config *cfg);                 /* BatchLogMelConfig::default (src/mel.rs:189-208) */
/* BatchLogMelSpectrogram::new (src/mel.rs:248-280); validate_batch_config's messages (src/mel.rs:656-683) come
 * back through melspec_last_error with MELSPEC_ERR_INVALID_ARG (== BatchLogMelError::InvalidConfig).
 * n_fft = 512 / win_length = 400 (the NeMo/Parakeet geometry) runs on the fused 512-point kernel, every other validated
 * config (n_fft <= 4096) on the generic f64 kernel (two orders of magnitude slower, same results). */
int melspec_blm_create(melspec_blm **out, int device, const melspec_blm_config *cfg);
void melspec_blm_destroy(melspec_blm *b);
size_t melspec_blm_num_frames(const melspec_blm *b, size_t n_samples);    /* valid frames (src/mel.rs:387-395) */
size_t melspec_blm_padded_frames(const melspec_blm *b, size_t n_samples); /* cols = pad_len(valid, pad_to) (src/mel.rs:751-756) */
/* compute_flat (src/mel.rs:304-385): feature-major [n_mels][cols] f32; *rows = n_mels, *cols = padded frames.
 * Empty input -> cols = 0 (src/mel.rs:326-332). */
int melspec_blm_compute_host(melspec_blm *b, const float *samples, size_t n_samples, float *out,
                             size_t out_capacity_floats, size_t *rows, size_t *cols);
/* Many equal-length clips resident in HBM; d_out = [clip][n_mels][cols]. */
int melspec_blm_compute_uniform_device(melspec_blm *b, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                       uint32_t n_clips, float *d_out, void *stream);
/* Clips of any length in one launch: clip c = d_pcm[h_offsets[c] .. + h_lengths[c]) -> [n_mels][cols_c] at d_out + h_out_offsets[c]
 * floats (NULL: packed in clip order), cols_c = melspec_blm_padded_frames(b, h_lengths[c]).  Fused kernel (n_fft 512 / win_length 400). */
int melspec_blm_compute_ragged_device(melspec_blm *b, const float *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths,
                                      uint32_t n_clips, float *d_out, const uint64_t *h_out_offsets, void *stream);
/* The same from host memory, many clips per call (BatchLogMelSpectrogram::compute is per clip, src/mel.rs:299): clip i ->
 * [n_mels][cols_i] at out + out_offsets[i] floats (NULL: packed); *total_columns = sum of cols_i.  Host pipeline as above. */
int melspec_blm_compute_batch_host(melspec_blm *b, const float *samples, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_clips,
                                   float *out, const uint64_t *out_offsets, size_t out_capacity_floats, uint64_t *total_columns);
int melspec_blm_release_scratch(melspec_blm *b);   /* as melspec_fbank_release_scratch */
/* Arithmetic of the fused kernel (n_fft 512 / win_length 400, 80 or 128 mels).  MELSPEC_PRECISION_F32: f32 window, FFT, power and
 * projection -- the reference's own arithmetic type for this frontend (src/mel.rs:251-252,356-357), as far from the f64 evaluation of
 * its definition as upstream's f32 code is (2.4e-4 on jfk_f32le.wav) at ~0.8 x the time.  AUTO (default) / F64: f64 up to |X|^2, within 1e-4
 * of that evaluation on every input.  melspec_blm_precision: what the next call will use (F32 or F64). */
int melspec_blm_set_precision(melspec_blm *b, int mode);
int melspec_blm_precision(const melspec_blm *b);
int melspec_blm_synchronize(melspec_blm *b, void *stream);

/* ---- streaming: Spectrogram::add + RingBuffer::maybe_mel (src/stft.rs:48-86, src/rb.rs:60-121) ---- */
/* A bank of n_streams independent live streams over one melspec_ctx (its geometry, tables and
 * kernels).  Per stream the device keeps what the reference keeps on the host: hop_buf's history (the
 * last n_fft - hop samples) and the samples accumulated towards the next hop (RingBuffer's
 * accumulated_samples); the host side of the handle keeps Spectrogram::idx.  A push appends a chunk of
 * any length <= max_chunk to each named stream and emits, per stream, exactly the frames the reference's
 * add_frame()/maybe_mel() loop would: one per completed hop once idx >= n_fft, so the first frame of a
 * stream starts at sample ceil(n_fft/hop)*hop - n_fft (80 for 400/160, 128 for 512/160) -- the alignment
 * testdata/rust_jfk_golden.npy pins (src/rb.rs:134-179).  Frames are computed in place on carry ++ chunk
 * by the batch kernels (no staging copy of the history), then the tail becomes the new carry.
 * The ctx must outlive the bank; one bank per host thread, like the ctx. */
typedef struct melspec_stream melspec_stream;

int melspec_stream_create(melspec_stream **out, melspec_ctx *ctx, uint32_t n_streams, uint32_t max_chunk);
void melspec_stream_destroy(melspec_stream *st);
/* Spectrogram::new state (zero history, idx = 0) for the listed streams; ids == NULL -> all. */
int melspec_stream_reset(melspec_stream *st, const uint32_t *ids, uint32_t n);
/* frames a push of n_new samples to stream id would emit now */
size_t melspec_stream_frames_after(const melspec_stream *st, uint32_t id, uint32_t n_new);

/* Host chunks: samples holds the n chunks back to back (lens[i] samples for stream ids[i], each id at most
 * once per call); out receives the emitted frames back to back in entry order ([frame][mel] each),
 * frames_out[i] how many entry i emitted.  Synchronous. */
int melspec_stream_push_host(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                             float *out, size_t out_capacity_floats, uint32_t *frames_out);
/* The same push emitting what Spectrogram::add itself returns (src/stft.rs:48-86): the spectrum of every completed frame,
 * [frame][bins] complex per entry (melspec_stft_bins; dtype MELSPEC_STFT_F32 / _F64), instead of its mel row. */
int melspec_stream_push_host_stft(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                  void *out, size_t out_capacity_complex, uint32_t *frames_out, int dtype, int full);
/* Spectrogram::add with fewer than hop samples (src/stft.rs:57-60): the pending samples of each listed
 * stream are zero-padded to a hop, idx advances by the real samples only, at most one frame each. */
int melspec_stream_flush_host(melspec_stream *st, const uint32_t *ids, uint32_t n, float *out, size_t out_capacity_floats,
                              uint32_t *frames_out);
/* Device producers write the next chunk of stream id at melspec_stream_input_ptr(st, id) (fixed per
 * stream, 16-byte aligned, room for max_chunk samples) and then push lengths only.  Output goes to
 * d_out + out_offsets[i] floats (NULL -> back to back).  Returns after the launches have completed. */
float *melspec_stream_input_ptr(melspec_stream *st, uint32_t id);
int melspec_stream_push_device(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                               const uint64_t *out_offsets, uint32_t *frames_out, void *stream);

/* ---- 8-bit quantisation + TGA container: replaces src/quant.rs ------------------------ */
/* The reference's wire/disk format right after the mel path: a mel-major interleaved image
 * ([n_mels][width] f32, what melspec_compute_uniform_device_interleaved(.., major_column_order = 0, ..)
 * writes) is quantised to u8 with its own {min,max} and wrapped in an 18-byte TARGA header plus an 8-byte
 * ID field holding {min,max} as little-endian f32 (tga_8bit_data, src/quant.rs:38-64).  Bytes are
 * identical to the CPU's: the arithmetic is the reference's f32 sequence (quantize, src/quant.rs:140-153:
 * f32::min/max folds that skip NaN, scale = 255/(max-min), round half away from zero, clamp). */
typedef struct melspec_tga melspec_tga;

int melspec_tga_create(melspec_tga **out, int device);
void melspec_tga_destroy(melspec_tga *q);

/* tga_8bit (src/quant.rs:29-36) cuts an image into chunks of <= 65535 columns
 * (chunk_frames_into_strides, :100-136), one TGA each.  Layout used by this library for the
 * Vec<Vec<u8>> it returns: chunk c of an image starts at byte c * chunk_stride (a multiple of 4);
 * every chunk but the last holds 26 + n_mels*65535 bytes, the last one last_chunk_bytes.
 * width == 0 -> 0 chunks. */
int melspec_tga_layout(int n_mels, size_t width, uint32_t *n_chunks, size_t *chunk_stride, size_t *last_chunk_bytes);

/* n_images images, image i at d_images + i*image_stride floats -> blobs at d_blobs + i*blob_stride bytes
 * (blob_stride a multiple of 4, >= n_chunks*chunk_stride; d_blobs 4-byte aligned).  Asynchronous on
 * `stream` (NULL -> the handle's stream).  save_tga_8bit (src/quant.rs:15-27) is this plus a file write. */
int melspec_tga_encode_device(melspec_tga *q, const float *d_images, size_t image_stride, int n_mels, size_t width,
                              uint32_t n_images, uint8_t *d_blobs, size_t blob_stride, void *stream);
/* parse_tga_8bit / load_tga_8bit (src/quant.rs:66-98) for the same layout: reads {min,max} from bytes
 * 18..25 of every chunk and dequantises (dequantize, :156-165: u8 * ((max-min)/255) + min, two roundings). */
/* PCM -> TGA blobs, device resident: tga_8bit_data(interleave_frames(mel(clip), false, min_width)) for n_clips uniform clips
 * (src/quant.rs:38-64, src/mel.rs:480-544), the images left in d_images ([n_clips][n_mels][W], W = melspec_interleaved_width) and one
 * blob per clip at d_blobs + i * blob_stride (melspec_tga_layout).  On the fused n_fft = 400 kernels the mel kernel folds each image's
 * {min, max} while it stores it, so the quantiser reads the image once: same bytes as melspec_compute_uniform_device_interleaved followed
 * by melspec_tga_encode_device.  `q` and `ctx` must be on one device; asynchronous on `stream` (NULL: the context's). */
int melspec_tga_encode_pcm_uniform_device(melspec_tga *q, melspec_ctx *ctx, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                          uint32_t n_clips, uint64_t min_width, float *d_images, uint8_t *d_blobs, size_t blob_stride,
                                          void *stream);
int melspec_tga_decode_device(melspec_tga *q, const uint8_t *d_blobs, size_t blob_stride, int n_mels, size_t width,
                              uint32_t n_images, float *d_images, size_t image_stride, void *stream);
/* tga_8bit(data, n_mels) on host memory; n_chunks blobs laid out as melspec_tga_layout says. */
int melspec_tga_encode_host(melspec_tga *q, const float *data, size_t len, int n_mels, uint8_t *out, size_t out_capacity,
                            uint32_t *n_chunks);
/* parse_tga_8bit(blob): skips the 18 header bytes without looking at them, like the reference;
 * n_bytes < 26 -> MELSPEC_ERR_INVALID_ARG ("failed to fill whole buffer"). */
int melspec_tga_decode_host(melspec_tga *q, const uint8_t *blob, size_t n_bytes, float *out, size_t out_capacity, size_t *n_values);

/* quantize / dequantize without the container (src/quant.rs:140-165; src/wasm.rs:113 uses it per frame).
 * range = {min, max}.  Device variants: d_out needs n rounded up to a multiple of 4 bytes. */
int melspec_quantize_device(melspec_tga *q, const float *d_frame, size_t n, uint8_t *d_out, float *d_range, void *stream);
int melspec_dequantize_device(melspec_tga *q, const uint8_t *d_data, size_t n, const float *d_range, float *d_out, void *stream);
int melspec_quantize_host(melspec_tga *q, const float *frame, size_t n, uint8_t *out, float *range);
int melspec_dequantize_host(melspec_tga *q, const uint8_t *data, size_t n, const float *range, float *out);
int melspec_tga_synchronize(melspec_tga *q);

/* ---- VAD column classification: vad_boundaries (src/vad.rs:256-340) -------------------- */
/* The reference's consumer of the mel image ([n_mels][width], what interleave_frames(.., false, ..) or
 * to_array2 (src/quant.rs:168-174) produce): a 3x3 Sobel stencil, per column the count of rows
 * y in [min(min_mel, n_mels-2), n_mels-2) whose squared gradient reaches min_energy^2, raw[x] = count >= min_y
 * (classify_columns_in_frame, :373-415), then a +-4 moving-window majority vote (smooth_mask, :343-360).
 * EdgeInfo::intersected() are the set entries of `smoothed`, non_intersected() the clear ones.  Same f64
 * arithmetic on the same f32 pixels as the reference: the masks are identical, not close. */
typedef struct melspec_vad_settings {      /* DetectionSettings, src/vad.rs:5-22 */
    double min_energy;                     /* 0.98 */
    int min_y;                             /* 11 */
    int min_x;                             /* 5: window of VoiceActivityDetector::add, not used by vad_boundaries */
    int min_mel;                           /* 2 */
} melspec_vad_settings;
void melspec_vad_default_settings(melspec_vad_settings *s);
/* width - 2, or 0 when n_mels < 3 or width < 3 (the reference then returns an empty EdgeInfo) */
size_t melspec_vad_mask_len(int n_mels, size_t width);
/* n_images images at d_images + i*image_stride floats -> byte masks at d_raw / d_smoothed + i*mask_stride
 * (melspec_vad_mask_len entries each) and, if d_longest_run != NULL, the longest run of consecutive intersected
 * columns per image (vad_on(edge_info, n), src/vad.rs:229-254, is `longest_run >= n` for n >= 2).
 * Asynchronous on `stream` of the current device. */
int melspec_vad_boundaries_device(const float *d_images, size_t image_stride, int n_mels, size_t width, uint32_t n_images,
                                  const melspec_vad_settings *settings, uint8_t *d_raw, uint8_t *d_smoothed, size_t mask_stride,
                                  uint32_t *d_longest_run, void *stream);
/* one image in host memory (device < 0: the current one); raw_out may be NULL */
int melspec_vad_boundaries_host(int device, const float *image, int n_mels, size_t width, const melspec_vad_settings *settings,
                                uint8_t *raw_out, uint8_t *smoothed_out, uint32_t *longest_run);

/* ---- the detector inside the streaming bank: VoiceActivityDetector::{new, add, add_activity}, src/vad.rs:137-208, per stream ----
 * The reference keeps the last min_x mel frames of a stream and runs vad_boundaries on that [n_mels][min_x] window for every frame
 * it is given.  With the stage on, every push / flush of the bank also feeds its streams' detectors on the device, from the mel rows
 * the push has just written (no copy back, no launch per frame): each frame costs one 3-column Sobel walk, the stream's state is its
 * last two rows and 64 bits of column history in HBM.  One record per emitted frame, packed in entry order like the rows:
 *   valid = 0 where add_activity returns None (fewer than min_x frames so far); otherwise active, leading_active_columns,
 *   active_columns and window_columns are VoiceActivity's fields (confidence = active_columns / window_columns or 0,
 *   frame_index = melspec_stream_vad_frames before the push + the frame's position in it).
 * Decisions are made on the f32 mel rows the path stores, widened to f64 like the reference's arithmetic.  settings == NULL turns the
 * stage off; turning it on (or melspec_stream_reset) starts every (listed) detector afresh.  min_x <= 66. */
typedef struct melspec_vad_activity {
    uint8_t valid, active;
    uint16_t leading_active_columns, active_columns, window_columns;
} melspec_vad_activity;
int melspec_stream_enable_vad(melspec_stream *st, const melspec_vad_settings *settings);
/* frames stream id has fed to its detector (the frame_index of the next one) */
uint64_t melspec_stream_vad_frames(const melspec_stream *st, uint32_t id);
/* melspec_stream_push_host / _flush_host / _push_device that also return the records (acts: host memory for the host calls, device
 * memory for the device call; capacity in records >= the frames the call emits) */
int melspec_stream_push_host_vad(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                 float *out, size_t out_capacity_floats, uint32_t *frames_out, melspec_vad_activity *acts,
                                 size_t acts_capacity);
int melspec_stream_flush_host_vad(melspec_stream *st, const uint32_t *ids, uint32_t n, float *out, size_t out_capacity_floats,
                                  uint32_t *frames_out, melspec_vad_activity *acts, size_t acts_capacity);
int melspec_stream_push_device_vad(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                                   const uint64_t *out_offsets, uint32_t *frames_out, melspec_vad_activity *d_acts, void *stream);

/* ---- device memory helpers for hosts with no HIP binding of their own --------------- */
/* (what the cudaMalloc/cudaMemcpyAsync externs of src/cuda.rs:185-199 give the Rust side) */
int melspec_malloc(void **dptr, size_t bytes);
int melspec_free(void *dptr);
int melspec_memcpy_h2d(void *dst_device, const void *src_host, size_t bytes);
int melspec_memcpy_d2h(void *dst_host, const void *src_device, size_t bytes);
int melspec_device_synchronize(void);

/* Synthetic PCM of SURVEY.md §8(d), generated on the device for benches/tests:
 * d_out[c*clip_stride + i] = hashnoise(seed, first_clip + c, i), c < n_clips, i < clip_len. */
int melspec_synth_pcm_device(float *d_out, uint64_t clip_stride, uint64_t clip_len,
                             uint64_t first_clip, uint32_t n_clips, uint32_t seed, void *stream);
/* samples [first_sample, first_sample + n_samples) of the same clips (a live producer for the streaming bank) */
int melspec_synth_pcm_window_device(float *d_out, uint64_t clip_stride, uint64_t first_sample, uint64_t n_samples,
                                    uint64_t first_clip, uint32_t n_clips, uint32_t seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MELSPEC_HIP_H */
