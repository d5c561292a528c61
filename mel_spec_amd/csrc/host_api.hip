// host_api.hip -- C ABI of libmelspec_hip.so (include/melspec_hip.h): the log-mel context and the host objects behind it.  Host logic is
// C++ because the reference's host side is compiled code (Rust, src/cuda.rs); this image has no Rust toolchain, so the Rust shim that binds
// this ABI is shipped as source in mel_spec_amd/rust/.  melspec_ctx mirrors CudaMelSpectrogram (src/cuda.rs:27-148): it owns the device
// tables, a stream and grow-only device scratch; melspec_compute_host is compute_mel_spectrogram.
#include "host_common.hpp"
#ifndef MS_WIDE_LAYOUTS_DEFAULT
#define MS_WIDE_LAYOUTS_DEFAULT 1
#endif

namespace melspec {
namespace host {

thread_local std::string g_last_error;

int pick_device(int device, DeviceInfo &info) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(MELSPEC_ERR_UNAVAILABLE, "no HIP device visible");
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= count) return fail(MELSPEC_ERR_INVALID_ARG, "device index out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(MELSPEC_ERR_UNAVAILABLE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MELSPEC_ERR_UNAVAILABLE, "device is not gfx950 (MI355X); this library ships gfx950 code only");
    info.device = device;
    info.cus = prop.multiProcessorCount;
    info.lds_per_block = prop.sharedMemPerBlock;
    return MELSPEC_OK;
}

}  // namespace host
}  // namespace melspec

extern "C" {

int melspec_abi_version(void) { return 1; }

#ifndef MELSPEC_SOURCE_HASH
#define MELSPEC_SOURCE_HASH "unknown"
#endif
static const char kSourceHash[] = "@melspec-source-hash:" MELSPEC_SOURCE_HASH;      // the marker lets build.py read it from the file
const char *melspec_source_hash(void) { return kSourceHash + sizeof("@melspec-source-hash:") - 1; }

int melspec_device_count(void) {
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(MELSPEC_ERR_UNAVAILABLE, "no HIP device visible");
    }
    int usable = 0;
    for (int d = 0; d < count; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
    }
    if (usable == 0) return fail(MELSPEC_ERR_UNAVAILABLE, "no gfx950 device visible");
    return usable;
}

const char *melspec_last_error(void) { return g_last_error.c_str(); }

}  // extern "C"

namespace {
// dense: [n_mels][fft_size / 2 + 1], empty = the default bank of MelSpectrogram::new
int create_ctx(melspec_ctx **out, int device, int fft_size, int hop_size, double sampling_rate, int n_mels, std::vector<double> dense) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    // src/cuda.rs:45-49
    if (fft_size <= 0 || hop_size <= 0 || n_mels <= 0)
        return fail(MELSPEC_ERR_INVALID_ARG, "fft_size, hop_size, and n_mels must be non-zero");
    if (!(sampling_rate > 0.0)) return fail(MELSPEC_ERR_INVALID_ARG, "sampling_rate must be > 0");
    if (fft_size < 2 || fft_size > kMaxGenericFft || n_mels > kMaxGenericMels)
        return fail(MELSPEC_ERR_UNSUPPORTED, "fft_size must be in [2,4096] and n_mels <= 1024");
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    melspec_ctx *c = new (std::nothrow) melspec_ctx();
    if (!c) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    c->dev = info; c->fft_size = fft_size; c->hop_size = hop_size; c->n_mels = n_mels; c->sr = sampling_rate;
    if (dense.empty()) dense = mel_filterbank(sampling_rate, fft_size, n_mels, -1.0, -1.0, false, true);
    c->dense = std::move(dense);
    auto bail = [&](int code) { melspec_destroy(c); return code; };
    if (hipSetDevice(info.device) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipSetDevice failed"));
    if (hipStreamCreate(&c->stream) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipStreamCreate failed"));

    // fused kernels: n_fft == 400, any hop up to 1024 (the 8-byte PCM loads only need 4-byte alignment, as every ragged clip offset
    // already demands), a two-filters-per-bin bank of <= 131 mels
    const bool runtime_lens = lab_int("MELSPEC_RUNTIME_LENS", 0, 0, 1) != 0;
    c->fast = (fft_size == 400) && (hop_size <= 1024) && build_fast_tables(c->dense, n_mels, c->ft, true) &&
              c->ft.interval;
    if (c->fast) {
        c->lens_kind = lens_match<LensI80>(c->ft.slots, n_mels) ? 1 : (lens_match<LensI128>(c->ft.slots, n_mels) ? 2 : 0);
        if (runtime_lens) c->lens_kind = 0;
        c->fast_lds = sizeof(float) * (c->ft.blob.size() + static_cast<size_t>(kWaveWaves) * WaveLayout::slice_floats() + kWaveWaves + 4);   // + RoundSync counters + the vote's words
        PreciseTables pt;
        const bool pt_ok = build_precise_tables(c->ft, pt, true);
        c->precise_lds = pt.blob.size() * 4 + static_cast<size_t>(kPreciseWaves) * PreciseLayout::slice_doubles() * sizeof(double) +
                         kPreciseWaves * sizeof(uint32_t);   // + RoundSync counters
        if (c->fast_lds > kLdsLimit || !pt_ok || c->precise_lds > kLdsLimit) c->fast = false;
        else c->pt = std::move(pt);
    }
    if (!c->fast && fft_size == 512 && lab_int("MELSPEC_W512", 1, 0, 1) != 0 && build_whisper512_tables<double>(c->dense, n_mels, c->ft512)) {
        const size_t slice_bytes = FbankLayout<double>::slice_elems() * sizeof(double) + 512;      // + the frame maxima
        c->waves512 = fused512_waves(c->ft512.blob.size() * 4, slice_bytes);
        c->lds512 = c->ft512.blob.size() * 4 + static_cast<size_t>(c->waves512) * slice_bytes;
        c->fast512 = c->lds512 <= kLdsLimit;
        if (c->fast512 && (rc = upload(c->d_blob512, c->ft512.blob))) return bail(rc);
        if (c->fast512 && w512_f32_bank(c->ft512.slots) && build_whisper512_tables<float>(c->dense, n_mels, c->f512.ft) && (rc = c->f512.finish(512))) return bail(rc);
        // MELSPEC_PRECISION_AUTO on the pair of 512-point kernels (round 6): the statistics words, the vote's tally and verdicts
        if (c->fast512 && c->f512.ok) {
            if ((rc = upload(c->fix.count, std::vector<uint64_t>(8, 0ull)))) return bail(rc);
            if ((rc = upload(c->fix.verdicts, std::vector<uint32_t>(static_cast<size_t>(kVoteSlots) * kVoteSlotStride, 0u)))) return bail(rc);
        }
    }
    if (c->fast && build_six_tables(c->dense, n_mels, c->ft6)) {
        c->lds6 = sizeof(float) * (c->ft6.blob.size() + static_cast<size_t>(kSixWaves) * SixLayout::slice_floats() + kSixWaves + 4);   // + arrival counters + the vote's words
        c->six = c->lds6 <= kLdsLimit;
        c->six_static = runtime_lens ? 0 : lens_match<LensSix80>(c->ft6.slots, n_mels) ? 1 : lens_match<LensSix64>(c->ft6.slots, n_mels) ? 2
                        : lens_match<LensSix40>(c->ft6.slots, n_mels) ? 3 : 0;
        if (c->six && (rc = upload(c->d_blob6, c->ft6.blob))) return bail(rc);
#ifdef MELSPEC_NO_SIX64          // A/B builds (tools/ab_build.sh): the five-frame f64 kernel everywhere
        const bool want64 = false;
#else
        const bool want64 = lab_int("MELSPEC_SIX64", 1, 0, 1) != 0;
#endif
        if (c->six && want64 && build_six64_tables(c->ft6, c->t64)) {
            c->lds64x = c->t64.blob.size() * 4 + static_cast<size_t>(kSix64Waves) * Six64Layout::slice_doubles() * sizeof(double) +
                        (kSix64Waves + 2) * sizeof(uint32_t);       // + the layout kernel's RoundSync counters + guard_wave_done's two words
            c->six64 = c->lds64x <= kLdsLimit;
            if (c->six64 && (rc = upload(c->d_blob64x, c->t64.blob))) return bail(rc);
        }
    }
#ifndef MELSPEC_NO_SIX64
    // Whisper large-v3's bank (128 mels) is past the nine slots of the f32 six-frame kernel, but the f64 one runs its mel phase when its
    // f64 arrays are dead and has the registers for fifteen: MELSPEC_PRECISION_F64 and AUTO's gated launch on plain batches (round 5)
    if (c->fast && !c->six && !runtime_lens && lab_int("MELSPEC_SIX64", 1, 0, 1) != 0) {
        FastTables wide;
        if (build_six_tables(c->dense, n_mels, wide, kSixWideSlots) && lens_match<LensSix128>(wide.slots, n_mels) && build_six64_tables(wide, c->t64)) {
            c->lds64x = c->t64.blob.size() * 4 + static_cast<size_t>(kSix64Waves) * Six64Layout::slice_doubles() * sizeof(double) + (kSix64Waves + 2) * sizeof(uint32_t);
            c->six64 = c->six64_wide = c->lds64x <= kLdsLimit;
            if (c->six64) {
                c->ft6.slots = wide.slots;          // the launch's copy of the slot table (run-time-lens code paths; unused by LensSix128)
                if ((rc = upload(c->d_blob64x, c->t64.blob))) return bail(rc);
                // ... and the f32 kernel of the same shape (round 6): plain batches of this bank leave the five-frame kernel
                c->lds6w = sizeof(float) * (wide.blob.size() + static_cast<size_t>(kSixWideWaves) * SixLayout::slice_floats() + kSixWideWaves + 4);
                c->six_wide32 = c->lds6w <= kLdsLimit && lab_int("MELSPEC_SIX_WIDE32", 1, 0, 1) != 0;
                c->six_wide32_layouts = c->six_wide32 && lab_int("MELSPEC_SIX_WIDE32_LAYOUTS", MS_WIDE_LAYOUTS_DEFAULT, 0, 1) != 0;
                if (c->six_wide32) {
                    c->ft6w = wide;
                    if ((rc = upload(c->d_blob6w, c->ft6w.blob))) return bail(rc);
                }
            }
        }
    }
#endif
    if (c->fast) {
        if ((rc = upload(c->d_blob, c->ft.blob))) return bail(rc);
        if ((rc = upload(c->d_blob64, c->pt.blob))) return bail(rc);
        {
            PreciseTables ps;
            if (!build_precise_tables(c->ft, ps, false)) return bail(fail(MELSPEC_ERR_INTERNAL, "spectrum tables"));
            ps.blob.resize(static_cast<size_t>(PreciseBlob::kCount) * 2);          // the f64 tables only
            if ((rc = upload(c->d_blob64s, ps.blob))) return bail(rc);
        }
        if ((rc = upload(c->fix.tab, build_fix_tables()))) return bail(rc);
        if ((rc = upload(c->fix.count, std::vector<uint64_t>(8, 0ull)))) return bail(rc);
        if ((rc = upload(c->fix.verdicts, std::vector<uint32_t>(static_cast<size_t>(kVoteSlots) * kVoteSlotStride, 0u)))) return bail(rc);
    }
    if (!c->fast) {          // the generic kernel also serves the layouts the fused 512 build does not store
        const int bins = fft_size / 2 + 1;
        // bins >= n_fft/2 contribute nothing (src/mel.rs:155-163)
        if ((rc = c->gt.build(fft_size, fft_size, fft_size / 2, hann_window(fft_size), c->dense, n_mels, bins))) return bail(rc);
        if (c->gt.lds_bytes > kLdsLimit) return bail(fail(MELSPEC_ERR_UNSUPPORTED, "geometry needs more LDS than one workgroup has"));
        if ((rc = generic_allow_lds())) return bail(rc);
    }
    *out = c;
    return MELSPEC_OK;
}
}  // namespace

extern "C" {

int melspec_create(melspec_ctx **out, int device, int fft_size, int hop_size, double sampling_rate, int n_mels) {
    return create_ctx(out, device, fft_size, hop_size, sampling_rate, n_mels, {});
}

int melspec_create_with_filterbank(melspec_ctx **out, int device, int fft_size, int hop_size, double sampling_rate, int n_mels,
                                   double f_min, double f_max, int htk, int norm) {
    if (out) *out = nullptr;
    if (fft_size < 2 || n_mels <= 0 || !(sampling_rate > 0.0) || fft_size > kMaxGenericFft || n_mels > kMaxGenericMels)
        return create_ctx(out, device, fft_size, hop_size, sampling_rate, n_mels, {});       // the common argument checks and messages
    return create_ctx(out, device, fft_size, hop_size, sampling_rate, n_mels,
                      mel_filterbank(sampling_rate, fft_size, n_mels, f_min, f_max, htk != 0, norm != 0));
}

int melspec_create_with_dense_filterbank(melspec_ctx **out, int device, int fft_size, int hop_size, double sampling_rate, int n_mels,
                                         const double *filters, int fft_bins) {
    if (out) *out = nullptr;
    if (!filters) return fail(MELSPEC_ERR_INVALID_ARG, "filters is NULL");
    if (fft_size < 2 || n_mels <= 0 || fft_size > kMaxGenericFft || n_mels > kMaxGenericMels)
        return create_ctx(out, device, fft_size, hop_size, sampling_rate, n_mels, {});
    if (fft_bins != fft_size / 2 + 1) return fail(MELSPEC_ERR_INVALID_ARG, "filters must have fft_size / 2 + 1 columns");
    std::vector<double> dense(filters, filters + static_cast<size_t>(n_mels) * fft_bins);
    for (double w : dense)
        if (!std::isfinite(w)) return fail(MELSPEC_ERR_INVALID_ARG, "filters must be finite");
    return create_ctx(out, device, fft_size, hop_size, sampling_rate, n_mels, std::move(dense));
}

void melspec_destroy(melspec_ctx *c) {
    if (!c) return;
    if (c->dev.device >= 0) (void)hipSetDevice(c->dev.device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->d_blob.release(); c->d_blob64.release(); c->d_blob64s.release(); c->d_blob512.release(); c->f512.d_blob.release(); c->d_blob6.release(); c->d_blob6w.release(); c->d_blob64x.release(); c->gt.release(); c->ragged.release();
    c->fix.release();
    c->dplan.release();
    c->pipe.release();
    c->st_start.release(); c->st_len.release(); c->st_off.release(); c->st_w.release(); c->st_jw.release(); c->st_job.release();
    delete c;
}

size_t melspec_num_frames(const melspec_ctx *c, size_t n_samples) {
    if (!c) return 0;
    uint64_t f; ctx_num_frames(c, n_samples, f);
    return static_cast<size_t>(f);
}
size_t melspec_max_frames_per_batch(const melspec_ctx *c) {
    // frames of one chunk of the host pipeline (16 MiB of PCM); the device entry points have no limit
    if (!c || kPipeChunkSamples < static_cast<uint64_t>(c->fft_size)) return 0;
    return static_cast<size_t>((kPipeChunkSamples - c->fft_size) / c->hop_size + 1);
}
int melspec_fft_size(const melspec_ctx *c) { return c ? c->fft_size : 0; }
int melspec_hop_size(const melspec_ctx *c) { return c ? c->hop_size : 0; }
int melspec_n_mels(const melspec_ctx *c) { return c ? c->n_mels : 0; }
int melspec_uses_fast_path(const melspec_ctx *c) { return c && (c->fast || c->fast512) ? 1 : 0; }

int melspec_set_precision(melspec_ctx *c, int mode) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (mode != MELSPEC_PRECISION_AUTO && mode != MELSPEC_PRECISION_F64 && mode != MELSPEC_PRECISION_F32)
        return fail(MELSPEC_ERR_INVALID_ARG, "precision must be MELSPEC_PRECISION_AUTO, _F64 or _F32");
    c->precision = mode;        // geometries on the generic kernels are f64 whatever the mode; the fused n_fft = 512 kernel is f64 unless F32 is asked for
    return MELSPEC_OK;
}
int melspec_precision(const melspec_ctx *c) {
    if (!c) return MELSPEC_PRECISION_AUTO;
    if (c->fast) return c->precision;
    if (c->precision == MELSPEC_PRECISION_F32 && c->fast512 && c->f512.ok) return MELSPEC_PRECISION_F32;
    if (c->precision == MELSPEC_PRECISION_AUTO && c->fix.adaptive && w512_auto_ok(c)) return MELSPEC_PRECISION_AUTO;      // plain batches vote (the layouts: f64)
    return MELSPEC_PRECISION_F64;
}
int melspec_set_precise(melspec_ctx *c, int on) { return melspec_set_precision(c, on ? MELSPEC_PRECISION_F64 : MELSPEC_PRECISION_AUTO); }
int melspec_is_precise(const melspec_ctx *c) { return c && melspec_precision(c) == MELSPEC_PRECISION_F64 ? 1 : 0; }   // the generic kernels are f64 whatever the mode

const char *melspec_plain_kernel_name(const melspec_ctx *c) {
    // the same decisions launch_ctx takes for a plain (uniform or ragged, [frame][mel]) batch
    if (!c) return "";
    if (!c->fast) {
        if (c->fast512 && c->precision == MELSPEC_PRECISION_F32 && c->f512.ok) return "melspec::fbank512_wave_kernel<float, 12, 1, kFlavorWhisper, RUNS> (n_fft = 512, f32, three waves per SIMD)";
        if (c->precision == MELSPEC_PRECISION_AUTO && c->fix.adaptive && w512_auto_ok(c))
            return "melspec::w512_auto_kernel<float, 12> (n_fft = 512, f32, precision guard + vote) + the gated melspec::w512_auto_kernel<double, 8>";
        if (c->fast512) return "melspec::fbank512_wave_kernel<double, 8, 1, kFlavorWhisper, RUNS> (n_fft = 512, f64)";
        switch (pow2_logm(c->gt)) {
            case 6: return "melspec::pow2_frame_kernel<6, kFlavorWhisper> (n_fft = 128, f64, frames owned by lane groups of a wave)";
            case 7: return "melspec::pow2_frame_kernel<7, kFlavorWhisper> (n_fft = 256, f64, frames owned by lane groups of a wave)";
            case 8: return "melspec::pow2_frame_kernel<8, kFlavorWhisper> (n_fft = 512, f64, frames owned by lane groups of a wave)";
            case 9: return "melspec::pow2_frame_kernel<9, kFlavorWhisper> (n_fft = 1024, f64, frames owned by lane groups of a wave)";
            case 10: return "melspec::pow2_frame_kernel<10, kFlavorWhisper> (n_fft = 2048 as two 512-point halves, f64, frames owned by lane groups of a wave)";
            default: break;
        }
        return "melspec::generic_frame_kernel<256> (f64, one frame per workgroup)";
    }
    if (c->precision == MELSPEC_PRECISION_F64 && c->six64_wide)
        return "melspec::whisper400_six64_kernel<15, LensSix128> (f64 FFT, six frames per wave, three waves per SIMD, fifteen mel slots)";
    if (c->precision == MELSPEC_PRECISION_F64 && c->six64)
        return "melspec::whisper400_six64_kernel<9, .> (f64 FFT, six frames per wave, three waves per SIMD)";
    if (c->precision == MELSPEC_PRECISION_F64)
        return c->ft.slots.n_slots <= 8 ? "melspec::whisper400_precise_kernel<8, ., RUNS> (f64 FFT)" : "melspec::whisper400_precise_kernel<12, ., RUNS> (f64 FFT)";
    const bool fix = c->precision == MELSPEC_PRECISION_AUTO;
    if (c->six)
        return c->six_static == 1 ? (fix ? "melspec::whisper400_six_runs_kernel<9, LensSix80> (precision guard on)" : "melspec::whisper400_six_runs_kernel<9, LensSix80>")
             : c->six_static == 2 ? (fix ? "melspec::whisper400_six_wide_runs_kernel<9, LensSix64> (twelve waves; precision guard on)" : "melspec::whisper400_six_wide_runs_kernel<9, LensSix64> (twelve waves)")
             : c->six_static == 3 ? (fix ? "melspec::whisper400_six_wide_runs_kernel<9, LensSix40> (twelve waves; precision guard on)" : "melspec::whisper400_six_wide_runs_kernel<9, LensSix40> (twelve waves)")
                             : (fix ? "melspec::whisper400_six_runs_kernel<9, LensRuntime> (precision guard on)" : "melspec::whisper400_six_runs_kernel<9, LensRuntime>");
    if (c->six_wide32) return fix ? "melspec::whisper400_six_wide_runs_kernel<15, LensSix128> (six frames per wave, twelve waves; precision guard on)"
                                  : "melspec::whisper400_six_wide_runs_kernel<15, LensSix128> (six frames per wave, twelve waves)";
    if (c->ft.slots.n_slots <= 8) return fix ? "melspec::whisper400_wave_runs_kernel<8, .> (precision guard on)" : "melspec::whisper400_wave_runs_kernel<8, .>";
    return fix ? "melspec::whisper400_wave_runs_kernel<12, .> (precision guard on)" : "melspec::whisper400_wave_runs_kernel<12, .>";
}

int melspec_guard_count(melspec_ctx *c, uint64_t *frames) {
    if (!c || !frames) return fail(MELSPEC_ERR_INVALID_ARG, "ctx/frames is NULL");
    *frames = 0;
    if (!c->fix.count.p) return MELSPEC_OK;
    HIP_TRY(hipSetDevice(c->dev.device));
    HIP_TRY(hipDeviceSynchronize());
    uint64_t n = 0;
    HIP_TRY(hipMemcpy(&n, c->fix.count.p, sizeof(n), hipMemcpyDeviceToHost));
    *frames = n;
    return MELSPEC_OK;
}

int melspec_set_auto_adaptive(melspec_ctx *c, int on) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    c->fix.adaptive = on != 0;
    return MELSPEC_OK;
}

int melspec_auto_state(melspec_ctx *c, int *heavy, double *fraction) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    const bool voting = c->precision == MELSPEC_PRECISION_AUTO && (c->fast || w512_auto_ok(c));
    if (voting) auto_poll(c);
    if (heavy) *heavy = (voting && c->fix.heavy) ? 1 : 0;
    if (fraction) *fraction = c->fix.fraction;
    return MELSPEC_OK;
}

int melspec_compute_uniform_device(melspec_ctx *c, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                   uint32_t n_clips, float *d_out, void *stream) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (n_clips == 0) return MELSPEC_OK;
    uint64_t fpc; ctx_num_frames(c, clip_len, fpc);
    if (fpc == 0) return MELSPEC_OK;   // empty output, like src/cuda.rs:91-93
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    if (n_clips > 1 && clip_stride < clip_len && clip_stride != 0)
        return fail(MELSPEC_ERR_INVALID_ARG, "clip_stride smaller than clip_len");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const BatchPlan pl = plan_uniform(d_pcm, d_out, clip_stride, fpc, n_clips, c->n_mels, ctx_frames_per_unit(c));
    return launch_ctx(c, pl.desc, s);
}


size_t melspec_interleaved_width(const melspec_ctx *c, size_t n_samples, size_t min_width) {
    if (!c) return 0;
    uint64_t f; ctx_num_frames(c, n_samples, f);
    return f == 0 ? 0 : static_cast<size_t>(interleaved_width(f, min_width));
}

int melspec_compute_uniform_device_interleaved(melspec_ctx *c, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                               uint32_t n_clips, float *d_out, int major_column_order, uint64_t min_width,
                                               void *stream) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (min_width % 2 != 0) return fail(MELSPEC_ERR_INVALID_ARG, "min_width must be even");   // src/mel.rs:488
    if (n_clips == 0) return MELSPEC_OK;
    uint64_t fpc; ctx_num_frames(c, clip_len, fpc);
    if (fpc == 0) return fail(MELSPEC_ERR_INVALID_ARG, "frames is empty");                      // src/mel.rs:487
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const BatchPlan pl = plan_uniform(d_pcm, d_out, clip_stride, fpc, n_clips, c->n_mels, ctx_frames_per_unit(c, true),
                                      interleaved_width(fpc, min_width), major_column_order == 0);
    return launch_ctx(c, pl.desc, s);
}

int melspec_compute_ragged_device(melspec_ctx *c, const float *d_pcm, const uint64_t *h_offsets,
                                  const uint64_t *h_lengths, uint32_t n_clips, float *d_out,
                                  const uint64_t *h_out_offsets, void *stream) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (n_clips == 0) return MELSPEC_OK;
    if (!h_offsets || !h_lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    std::vector<uint64_t> frames(n_clips);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_clips; ++i) { ctx_num_frames(c, h_lengths[i], frames[i]); total += frames[i]; }
    if (total == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    BatchPlan pl;
    RaggedSlot *slot = nullptr;
    int rc = plan_ragged(c->ragged, s, d_pcm, d_out, h_offsets, frames, h_out_offsets, n_clips, c->n_mels,
                         ctx_frames_per_unit(c), pl, slot);
    if (!rc) rc = launch_ctx(c, pl.desc, s);
    plan_ragged_done(slot, s);
    return rc;
}

int melspec_compute_ragged_device_desc(melspec_ctx *c, const float *d_pcm, const uint64_t *d_offsets, const uint64_t *d_lengths,
                                       uint32_t n_clips, float *d_out, const uint64_t *d_out_offsets, uint64_t max_total_frames, void *stream) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (n_clips == 0 || max_total_frames == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out || !d_offsets || !d_lengths) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    BatchPlan pl;
    int rc = plan_ragged_device(c->dplan, s, d_pcm, d_out, d_offsets, d_lengths, d_out_offsets, n_clips, static_cast<uint64_t>(c->fft_size),
                                static_cast<uint64_t>(c->hop_size), static_cast<uint32_t>(c->n_mels), ctx_frames_per_unit(c), max_total_frames, pl);
    if (rc) return rc;
    return launch_ctx(c, pl.desc, s);
}

int melspec_time_uniform_device(melspec_ctx *c, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                uint32_t n_clips, float *d_out, int warmup, int iters, float *avg_ms) {
    if (!c || !avg_ms || iters < 1) return fail(MELSPEC_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = MELSPEC_OK;
    for (int i = 0; i < warmup && !rc; ++i)
        rc = melspec_compute_uniform_device(c, d_pcm, clip_stride, clip_len, n_clips, d_out, c->stream);
    if (!rc) {
        (void)hipEventRecord(e0, c->stream);
        for (int i = 0; i < iters && !rc; ++i)
            rc = melspec_compute_uniform_device(c, d_pcm, clip_stride, clip_len, n_clips, d_out, c->stream);
        (void)hipEventRecord(e1, c->stream);
        const hipError_t e = hipEventSynchronize(e1);
        if (!rc && e != hipSuccess) rc = fail_hip(e, "hipEventSynchronize");
        float ms = 0.0f;
        if (!rc) { (void)hipEventElapsedTime(&ms, e0, e1); *avg_ms = ms / iters; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int melspec_time_first_kernel(melspec_ctx *c, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len, uint32_t n_clips, float *d_out,
                              int warmup, int iters, float *avg_first_kernel_ms) {
    if (!c || !avg_first_kernel_ms || iters < 1 || iters > 4096) return fail(MELSPEC_ERR_INVALID_ARG, "bad argument");
    if (!c->fast || c->precision == MELSPEC_PRECISION_F64) return fail(MELSPEC_ERR_UNSUPPORTED, "the fused f32 n_fft = 400 kernels only");
    HIP_TRY(hipSetDevice(c->dev.device));
    int rc = MELSPEC_OK;
    for (int i = 0; i < warmup && !rc; ++i) rc = melspec_compute_uniform_device(c, d_pcm, clip_stride, clip_len, n_clips, d_out, c->stream);
    std::vector<hipEvent_t> ev;
    ev.reserve(2 * static_cast<size_t>(iters));
    c->first_kernel_events = &ev;
    for (int i = 0; i < iters && !rc; ++i) rc = melspec_compute_uniform_device(c, d_pcm, clip_stride, clip_len, n_clips, d_out, c->stream);
    c->first_kernel_events = nullptr;
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (!rc && e != hipSuccess) rc = fail_hip(e, "hipStreamSynchronize");
    double sum = 0.0;
    size_t n = 0;
    for (size_t i = 0; i + 1 < ev.size(); i += 2) {
        float ms = 0.0f;
        if (!rc && hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) { sum += ms; ++n; }
    }
    for (hipEvent_t x : ev) (void)hipEventDestroy(x);
    if (!rc && n == 0) rc = fail(MELSPEC_ERR_INTERNAL, "no launch was timed");
    if (!rc) *avg_first_kernel_ms = static_cast<float>(sum / static_cast<double>(n));
    return rc;
}

int melspec_synchronize(melspec_ctx *c, void *stream) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    HIP_TRY(hipStreamSynchronize(stream ? static_cast<hipStream_t>(stream) : c->stream));
    return MELSPEC_OK;
}

namespace {

// clip (src, n) -> its frames at dst, cut into frame-aligned pieces of at most kPipeChunkSamples samples
void push_segments(const melspec_ctx *c, const float *src, uint64_t n, float *dst, uint64_t frames, std::vector<HostSeg> &segs) {
    if (frames == 0) return;
    const uint64_t fft = static_cast<uint64_t>(c->fft_size), hop = static_cast<uint64_t>(c->hop_size);
    if (n <= kPipeChunkSamples) { segs.push_back(HostSeg{src, n, dst, frames}); return; }
    const uint64_t per = (kPipeChunkSamples - fft) / hop + 1;      // frames per piece
    for (uint64_t f0 = 0; f0 < frames; f0 += per) {
        const uint64_t nf = frames - f0 < per ? frames - f0 : per;
        segs.push_back(HostSeg{src + f0 * hop, (nf - 1) * hop + fft, dst + f0 * static_cast<uint64_t>(c->n_mels), nf});
    }
}

int run_host_pipe(melspec_ctx *c, const std::vector<HostSeg> &segs) {
    const char *where = "";
    const int rc = c->pipe.run(segs, c->n_mels, kPipeChunkSamples, c->stream,
                               [c](const float *d_in, const uint64_t *offs, const uint64_t *lens, uint32_t n, float *d_out,
                                   const uint64_t *ooffs, hipStream_t s) {
                                   return melspec_compute_ragged_device(c, d_in, offs, lens, n, d_out, ooffs, s);
                               }, &where);
    if (rc > 0 && where[0] && std::strcmp(where, "kernel launch") != 0) return fail_hip(static_cast<hipError_t>(rc), where);
    return rc;
}
}  // namespace

int melspec_compute_host(melspec_ctx *c, const float *samples, size_t n_samples, float *out,
                         size_t out_capacity_floats, size_t *n_frames) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (n_frames) *n_frames = 0;
    uint64_t frames; ctx_num_frames(c, n_samples, frames);
    if (frames == 0) return MELSPEC_OK;            // Ok(Vec::new()), src/cuda.rs:91-93
    if (!samples || !out) return fail(MELSPEC_ERR_INVALID_ARG, "samples/out is NULL");
    const uint64_t need = frames * static_cast<uint64_t>(c->n_mels);
    if (out_capacity_floats < need) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(c->dev.device));
    int rc;
    // one clip, or frame-aligned pieces of a long one, through the host pipeline (host_pipe.hpp; calls up to 32 MB take its
    // single-chunk path: one copy each way by the runtime, one launch, one synchronise)
    std::vector<HostSeg> segs;
    push_segments(c, samples, n_samples, out, frames, segs);
    if ((rc = run_host_pipe(c, segs))) return rc;
    if (n_frames) *n_frames = static_cast<size_t>(frames);
    return MELSPEC_OK;
}

int melspec_compute_batch_host(melspec_ctx *c, const float *samples, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_clips,
                               float *out, const uint64_t *out_offsets, size_t out_capacity_floats, uint64_t *total_frames) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (total_frames) *total_frames = 0;
    if (n_clips == 0) return MELSPEC_OK;
    if (!offsets || !lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    std::vector<HostSeg> segs;
    segs.reserve(n_clips);
    uint64_t total = 0, cursor = 0;
    for (uint32_t i = 0; i < n_clips; ++i) {
        uint64_t f; ctx_num_frames(c, lengths[i], f);
        const uint64_t oo = out_offsets ? out_offsets[i] : cursor;
        const uint64_t fl = f * static_cast<uint64_t>(c->n_mels);
        if (f && oo + fl > out_capacity_floats) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
        if (f && (!samples || !out)) return fail(MELSPEC_ERR_INVALID_ARG, "samples/out is NULL");
        push_segments(c, samples + offsets[i], lengths[i], out + oo, f, segs);
        cursor += fl; total += f;
    }
    if (total_frames) *total_frames = total;
    if (total == 0) return MELSPEC_OK;
    HIP_TRY(hipSetDevice(c->dev.device));
    return run_host_pipe(c, segs);
}


// ---- STFT export: Spectrogram::compute_all_cpu (src/stft.rs:89-115) / Spectrogram::add (src/stft.rs:48-86) -----------------
size_t melspec_stft_bins(const melspec_ctx *c, int full) {
    return !c ? 0 : static_cast<size_t>(full ? c->fft_size : c->fft_size / 2 + 1);
}


int melspec_stft_uniform_device(melspec_ctx *c, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len, uint32_t n_clips,
                                void *d_out, int dtype, int full, void *stream) {
    int rc = stft_args(c, dtype);
    if (rc) return rc;
    if (n_clips == 0) return MELSPEC_OK;
    uint64_t fpc; ctx_num_frames(c, clip_len, fpc);
    if (fpc == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    const int bins = static_cast<int>(melspec_stft_bins(c, full));
    const int words = bins * 2 * (dtype == MELSPEC_STFT_F64 ? 2 : 1);
    const BatchPlan pl = plan_uniform(d_pcm, static_cast<float *>(d_out), clip_stride, fpc, n_clips, words, c->fast ? kFPW : 1);
    return launch_stft(c, pl.desc, bins, dtype, stream ? static_cast<hipStream_t>(stream) : c->stream);
}

int melspec_stft_ragged_device(melspec_ctx *c, const float *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths, uint32_t n_clips,
                               void *d_out, const uint64_t *h_out_offsets, int dtype, int full, void *stream) {
    int rc = stft_args(c, dtype);
    if (rc) return rc;
    if (n_clips == 0) return MELSPEC_OK;
    if (!h_offsets || !h_lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    std::vector<uint64_t> frames(n_clips), oo;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_clips; ++i) { ctx_num_frames(c, h_lengths[i], frames[i]); total += frames[i]; }
    if (total == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const int bins = static_cast<int>(melspec_stft_bins(c, full));
    const int words = bins * 2 * (dtype == MELSPEC_STFT_F64 ? 2 : 1);
    if (h_out_offsets) {                          // complex elements -> 32-bit words
        oo.resize(n_clips);
        for (uint32_t i = 0; i < n_clips; ++i) oo[i] = h_out_offsets[i] * static_cast<uint64_t>(words / bins);
    }
    BatchPlan pl;
    RaggedSlot *slot = nullptr;
    rc = plan_ragged(c->ragged, s, d_pcm, static_cast<float *>(d_out), h_offsets, frames, h_out_offsets ? oo.data() : nullptr, n_clips, words,
                     c->fast ? kFPW : 1, pl, slot);
    if (!rc) rc = launch_stft(c, pl.desc, bins, dtype, s);
    plan_ragged_done(slot, s);
    return rc;
}

int melspec_stft_host(melspec_ctx *c, const float *samples, size_t n_samples, void *out, size_t out_capacity_complex, int dtype, int full,
                      size_t *n_frames) {
    int rc = stft_args(c, dtype);
    if (rc) return rc;
    if (n_frames) *n_frames = 0;
    uint64_t frames; ctx_num_frames(c, n_samples, frames);
    if (frames == 0) return MELSPEC_OK;
    if (!samples || !out) return fail(MELSPEC_ERR_INVALID_ARG, "samples/out is NULL");
    const uint64_t need = frames * melspec_stft_bins(c, full);
    if (out_capacity_complex < need) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(c->dev.device));
    const size_t esz = dtype == MELSPEC_STFT_F64 ? 16 : 8;
    DevBuf din, dout;
    auto done = [&](int code) { din.release(); dout.release(); return code; };
    if ((rc = din.ensure(n_samples * sizeof(float))) || (rc = dout.ensure(need * esz))) return done(rc);
    if (hipMemcpyAsync(din.p, samples, n_samples * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpyAsync failed"));
    if ((rc = melspec_stft_uniform_device(c, static_cast<const float *>(din.p), n_samples, n_samples, 1, dout.p, dtype, full, c->stream))) return done(rc);
    if (hipMemcpyAsync(out, dout.p, need * esz, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpyAsync failed"));
    if (hipStreamSynchronize(c->stream) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipStreamSynchronize failed"));
    if (n_frames) *n_frames = static_cast<size_t>(frames);
    return done(MELSPEC_OK);
}

int melspec_release_scratch(melspec_ctx *c) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->pipe.release();
    c->ragged.release();
    c->fix.list.release();
    c->fix.used = false;
    return MELSPEC_OK;
}

int melspec_host_alloc(void **p, size_t bytes) {
    if (!p) return fail(MELSPEC_ERR_INVALID_ARG, "p is NULL");
    *p = nullptr;
    HIP_TRY(hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocDefault));
    return MELSPEC_OK;
}
int melspec_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return MELSPEC_OK;
}

// ---- per-clip sharding over the GPUs of one node (independent units, no data-path collective) -------------
// The reference has no multi-device surface (src/cuda.rs binds one device); this is additive.  One context + stream per
// device, one host thread per device while a call runs; contiguous blocks of clips per device, balanced by samples.

int melspec_shard_by_samples(const uint64_t *lengths, uint32_t n_clips, int n_shards, uint32_t *bounds) {
    if (n_shards < 1 || !bounds || (n_clips && !lengths)) return fail(MELSPEC_ERR_INVALID_ARG, "bad shard_by_samples argument");
    long double total = 0;
    for (uint32_t i = 0; i < n_clips; ++i) total += static_cast<long double>(lengths[i]);
    bounds[0] = 0;
    int r = 1;
    long double acc = 0;
    for (uint32_t i = 0; i < n_clips; ++i) {
        acc += static_cast<long double>(lengths[i]);
        while (r < n_shards && acc >= total * r / n_shards) bounds[r++] = i + 1;      // shard r-1 ends behind clip i
    }
    while (r <= n_shards) bounds[r++] = n_clips;
    return MELSPEC_OK;
}

}  // extern "C"

struct melspec_sharded {
    std::vector<melspec_ctx *> ctx;
};

extern "C" {

int melspec_sharded_create(melspec_sharded **out, const int *devices, int n_devices, int fft_size, int hop_size, double sampling_rate,
                           int n_mels) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    std::vector<int> devs;
    if (devices) {
        if (n_devices < 1) return fail(MELSPEC_ERR_INVALID_ARG, "n_devices must be >= 1");
        devs.assign(devices, devices + n_devices);
    } else {
        // the ordinals of the gfx950 devices themselves: on a node with another GPU in front of them they are not 0 .. n-1
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
        for (int d = 0; d < count; ++d) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) devs.push_back(d);
        }
        if (devs.empty()) return fail(MELSPEC_ERR_UNAVAILABLE, "no gfx950 device visible");
    }
    melspec_sharded *s = new (std::nothrow) melspec_sharded();
    if (!s) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    for (int d : devs) {
        melspec_ctx *c = nullptr;
        const int rc = melspec_create(&c, d, fft_size, hop_size, sampling_rate, n_mels);
        if (rc) { melspec_sharded_destroy(s); return rc; }
        s->ctx.push_back(c);
    }
    *out = s;
    return MELSPEC_OK;
}

void melspec_sharded_destroy(melspec_sharded *s) {
    if (!s) return;
    for (melspec_ctx *c : s->ctx) melspec_destroy(c);
    delete s;
}

int melspec_sharded_n_shards(const melspec_sharded *s) { return s ? static_cast<int>(s->ctx.size()) : 0; }
melspec_ctx *melspec_sharded_ctx(melspec_sharded *s, int shard) {
    return (s && shard >= 0 && shard < static_cast<int>(s->ctx.size())) ? s->ctx[shard] : nullptr;
}

int melspec_sharded_compute_batch_host(melspec_sharded *s, const float *samples, const uint64_t *offsets, const uint64_t *lengths,
                                       uint32_t n_clips, float *out, const uint64_t *out_offsets, size_t out_capacity_floats,
                                       uint64_t *total_frames) {
    if (!s || s->ctx.empty()) return fail(MELSPEC_ERR_INVALID_ARG, "sharded object is NULL");
    if (total_frames) *total_frames = 0;
    if (n_clips == 0) return MELSPEC_OK;
    if (!offsets || !lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    const int n = static_cast<int>(s->ctx.size());
    std::vector<uint32_t> bounds(static_cast<size_t>(n) + 1);
    int rc = melspec_shard_by_samples(lengths, n_clips, n, bounds.data());
    if (rc) return rc;
    // output positions are global (packed in clip order unless given), so every shard writes its own part of `out`
    std::vector<uint64_t> oo(n_clips);
    const int nm = s->ctx[0]->n_mels;
    uint64_t cursor = 0;
    for (uint32_t i = 0; i < n_clips; ++i) {
        oo[i] = out_offsets ? out_offsets[i] : cursor;
        cursor += static_cast<uint64_t>(melspec_num_frames(s->ctx[0], lengths[i])) * nm;
    }
    std::vector<int> rcs(n, MELSPEC_OK);
    std::vector<std::string> msgs(n);
    std::vector<uint64_t> frames(n, 0);
    auto work = [&](int k) {
        const uint32_t lo = bounds[k], hi = bounds[k + 1];
        if (hi == lo) return;
        rcs[k] = melspec_compute_batch_host(s->ctx[k], samples, offsets + lo, lengths + lo, hi - lo, out, oo.data() + lo,
                                            out_capacity_floats, &frames[k]);
        if (rcs[k]) msgs[k] = g_last_error;          // thread-local: carried back to the caller's thread below
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < n; ++k) threads.emplace_back(work, k);
    work(0);
    for (auto &t : threads) t.join();
    uint64_t total = 0;
    for (int k = 0; k < n; ++k) {
        if (rcs[k]) { g_last_error = "shard " + std::to_string(k) + ": " + msgs[k]; return rcs[k]; }
        total += frames[k];
    }
    if (total_frames) *total_frames = total;
    return MELSPEC_OK;
}

// Device-resident shards ("one ctx + stream per device" with the data already where it is computed): shard k's
// clips are on device k, its frames stay there.  The launches are stream-ordered on every shard's own context stream and issued
// from the calling thread (a launch costs microseconds; the host pipeline of the host form is what needs a thread per device);
// melspec_sharded_synchronize waits for all of them.  Nothing crosses a device boundary.
int melspec_sharded_compute_uniform_device(melspec_sharded *s, const float *const *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                           const uint32_t *n_clips, float *const *d_out) {
    if (!s || s->ctx.empty()) return fail(MELSPEC_ERR_INVALID_ARG, "sharded object is NULL");
    if (!d_pcm || !n_clips || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "per-shard array is NULL");
    for (size_t k = 0; k < s->ctx.size(); ++k) {
        if (n_clips[k] == 0) continue;
        const int rc = melspec_compute_uniform_device(s->ctx[k], d_pcm[k], clip_stride, clip_len, n_clips[k], d_out[k], nullptr);
        if (rc) { g_last_error = "shard " + std::to_string(k) + ": " + g_last_error; return rc; }
    }
    return MELSPEC_OK;
}

int melspec_sharded_compute_ragged_device(melspec_sharded *s, const float *const *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths,
                                          const uint32_t *n_clips, float *const *d_out, const uint64_t *h_out_offsets) {
    if (!s || s->ctx.empty()) return fail(MELSPEC_ERR_INVALID_ARG, "sharded object is NULL");
    if (!d_pcm || !n_clips || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "per-shard array is NULL");
    uint64_t first = 0;
    for (size_t k = 0; k < s->ctx.size(); ++k) {
        if (n_clips[k] != 0) {
            if (!h_offsets || !h_lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
            const int rc = melspec_compute_ragged_device(s->ctx[k], d_pcm[k], h_offsets + first, h_lengths + first, n_clips[k], d_out[k],
                                                         h_out_offsets ? h_out_offsets + first : nullptr, nullptr);
            if (rc) { g_last_error = "shard " + std::to_string(k) + ": " + g_last_error; return rc; }
        }
        first += n_clips[k];
    }
    return MELSPEC_OK;
}

int melspec_sharded_synchronize(melspec_sharded *s) {
    if (!s) return fail(MELSPEC_ERR_INVALID_ARG, "sharded object is NULL");
    for (melspec_ctx *c : s->ctx) {
        const int rc = melspec_synchronize(c, nullptr);
        if (rc) return rc;
    }
    return MELSPEC_OK;
}

// Consolidation of per-device results on one device (optional, not part of the frames/s figure): piece i =
// bytes[i] bytes at srcs[i] on src_devices[i] -> dst + dst_offsets[i] on dst_device, every piece on a stream of its source
// device so that the pieces travel over their own xGMI links at the same time.  Synchronous.
int melspec_gather_peer(int dst_device, void *dst, const int *src_devices, const void *const *srcs, const size_t *bytes,
                        const size_t *dst_offsets, int n) {
    if (n <= 0) return MELSPEC_OK;
    if (!dst || !src_devices || !srcs || !bytes || !dst_offsets) return fail(MELSPEC_ERR_INVALID_ARG, "NULL argument");
    std::vector<hipStream_t> streams(n, nullptr);
    int rc = MELSPEC_OK;
    // what failed, for whom: on an 8-GPU node "invalid device ordinal" alone does not say which of the seven links it was
    auto piece_failed = [&](hipError_t e, const char *what, int i) {
        const int code = fail_hip(e, what);
        g_last_error = "melspec_gather_peer: piece " + std::to_string(i) + " of " + std::to_string(n) + ", source device " + std::to_string(src_devices[i]) +
                       " -> destination device " + std::to_string(dst_device) + ": " + g_last_error;
        return code;
    };
    for (int i = 0; i < n && !rc; ++i) {
        if (bytes[i] == 0) continue;
        hipError_t e = hipSetDevice(src_devices[i]);
        if (e != hipSuccess) { rc = piece_failed(e, "hipSetDevice", i); break; }
        // Peer access lets the copy run over the xGMI link between the two devices.  Where the runtime says it cannot be had (can == 0:
        // another node, an IOMMU setting -- and, trivially, a device with itself) the copy below still works, staged by the runtime; where
        // it can, a failure to enable it other than "already enabled" is the piece's error.
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, src_devices[i], dst_device) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        if (can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(dst_device, 0);
            if (pe != hipSuccess) (void)hipGetLastError();
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { rc = piece_failed(pe, "hipDeviceEnablePeerAccess", i); break; }
        }
        if ((e = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking)) != hipSuccess) { rc = piece_failed(e, "hipStreamCreateWithFlags", i); break; }
        e = hipMemcpyPeerAsync(static_cast<char *>(dst) + dst_offsets[i], dst_device, srcs[i], src_devices[i], bytes[i], streams[i]);
        if (e != hipSuccess) rc = piece_failed(e, "hipMemcpyPeerAsync", i);
    }
    // the pieces already queued finish (or fail) before the call returns, whatever happened to a later one: no stream outlives it
    for (int i = 0; i < n; ++i) {
        if (!streams[i]) continue;
        (void)hipSetDevice(src_devices[i]);
        const hipError_t e = hipStreamSynchronize(streams[i]);
        if (e != hipSuccess && !rc) rc = piece_failed(e, "hipStreamSynchronize", i);
        (void)hipStreamDestroy(streams[i]);
    }
    return rc;
}

}  // extern "C"

extern "C" {

// ---- host-side table builders ------------------------------------------------------------

int melspec_mel_filterbank(double sr, int n_fft, int n_mels, double f_min, double f_max, int htk, int norm, double *out) {
    if (!out || n_fft < 2 || n_mels < 1 || !(sr > 0.0)) return fail(MELSPEC_ERR_INVALID_ARG, "bad mel_filterbank argument");
    const std::vector<double> w = mel_filterbank(sr, n_fft, n_mels, f_min, f_max, htk != 0, norm != 0);
    std::memcpy(out, w.data(), w.size() * sizeof(double));
    return MELSPEC_OK;
}

double melspec_hz_to_mel(double frequency, int htk) { return hz_to_mel(frequency, htk != 0); }
double melspec_mel_to_hz(double mel_v, int htk) { return mel_to_hz(mel_v, htk != 0); }

int melspec_mel_frequencies(int n_mels, double fmin, double fmax, int htk, double *out) {
    if (!out || n_mels < 1) return fail(MELSPEC_ERR_INVALID_ARG, "bad mel_frequencies argument");
    // Array1::linspace(min_mel, max_mel, n_mels) mapped through mel_to_hz (src/mel.rs:631-637)
    const double lo = hz_to_mel(fmin, htk != 0), hi = hz_to_mel(fmax, htk != 0);
    const double step = n_mels > 1 ? (hi - lo) / (n_mels - 1) : 0.0;
    for (int i = 0; i < n_mels; ++i) out[i] = mel_to_hz(lo + step * i, htk != 0);
    return MELSPEC_OK;
}

int melspec_fft_frequencies(double sr, int n_fft, double *out) {
    if (!out || n_fft < 1) return fail(MELSPEC_ERR_INVALID_ARG, "bad fft_frequencies argument");
    const double step = sr / n_fft;
    for (int i = 0; i <= n_fft / 2; ++i) out[i] = step * i;
    return MELSPEC_OK;
}

int melspec_hann_window(int n, double *out) {
    if (!out || n < 1) return fail(MELSPEC_ERR_INVALID_ARG, "bad hann_window argument");
    const std::vector<double> w = hann_window(n);
    std::memcpy(out, w.data(), w.size() * sizeof(double));
    return MELSPEC_OK;
}

int melspec_kaldi_mel_filterbank(double sample_rate, int fft_size, int num_mel_bins, double low_freq,
                                 double high_freq, double *out) {
    if (!out || fft_size < 2 || num_mel_bins < 1 || !(sample_rate > 0.0))
        return fail(MELSPEC_ERR_INVALID_ARG, "bad kaldi_mel_filterbank argument");
    const std::vector<double> w = kaldi_mel_filterbank(sample_rate, fft_size, num_mel_bins, low_freq, high_freq);
    std::memcpy(out, w.data(), w.size() * sizeof(double));
    return MELSPEC_OK;
}

}  // extern "C"

extern "C" {

// ---- device memory helpers --------------------------------------------------------------

int melspec_malloc(void **dptr, size_t bytes) {
    if (!dptr) return fail(MELSPEC_ERR_INVALID_ARG, "dptr is NULL");
    *dptr = nullptr;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return MELSPEC_OK;
}
int melspec_free(void *dptr) {
    if (dptr) HIP_TRY(hipFree(dptr));
    return MELSPEC_OK;
}
int melspec_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    if (bytes) HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return MELSPEC_OK;
}
int melspec_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    if (bytes) HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MELSPEC_OK;
}
int melspec_device_synchronize(void) {
    HIP_TRY(hipDeviceSynchronize());
    return MELSPEC_OK;
}

}  // extern "C"
