// fbank512.hip -- the fused 512-point family (fbank512_kernels.hpp) and the two objects of the C ABI that live on it: the Kaldi fbank
// context (melspec_fbank_*, src/fbank.rs:94-313) and the NeMo / Parakeet frontend (melspec_blm_*, src/mel.rs:239-396); Whisper at n_fft = 512
// reaches the same kernels through launch_whisper512.
#include "host_common.hpp"
#include "fbank512_kernels.hpp"

namespace {
// Lens: compile-time slot lengths when the context's filterbank is one of the default banks (8-wave shape only; the
// 4-wave fallback for oversized tables keeps the run-time loop)
template <class T, int FLAVOR, int NSLOTS, class Lens = LensRuntime>
int launch_fused512(int waves, const FbankFastParams &fp, size_t lds, int cus, hipStream_t s) {
    // frame-major plain output (Kaldi always, Whisper-512 without a layout): a contiguous run of units per wave, 8-wave shape only
    constexpr bool kCanRun = FLAVOR != kFlavorNemo;
    const bool plain = !fp.b.mel_major && (fp.b.d_unit_prefix != nullptr || fp.b.out_width == fp.b.frames_per_clip);
    if constexpr (sizeof(T) == 4) {
        // MELSPEC_PRECISION_F32: the f32 instantiation, twelve waves = three per SIMD (158-168 VGPRs without spills; at sixteen waves the
        // 128-VGPR budget spills 24-38 registers inside the unit loop and the kernel is slower than the f64 one, profiles/r05_fb512_twelve_waves.txt)
        static_assert(Lens::kStatic, "the f32 instantiation exists for the compile-time banks");
        static std::atomic<uint64_t> attr12{0};
        if (!device_done(attr12)) {
            int rc = allow_big_lds(&fbank512_wave_kernel<T, 12, 1, FLAVOR, NSLOTS, Lens>, "hipFuncSetAttribute(fbank512_wave_kernel<float>, 12 waves)");
            if (!rc && kCanRun) rc = allow_big_lds(&fbank512_wave_kernel<T, 12, 1, FLAVOR, NSLOTS, Lens, kCanRun>, "hipFuncSetAttribute(fbank512_wave_kernel<float>, runs)");
            if (rc) return rc;
            mark_device_done(attr12);
        }
        const unsigned grid12 = grid_for_xcd((fp.b.n_units + 11) / 12, cus, 1);
        if (kCanRun && plain) hipLaunchKernelGGL((fbank512_wave_kernel<T, 12, 1, FLAVOR, NSLOTS, Lens, kCanRun>), dim3(grid12), dim3(768), lds, s, fp);
        else hipLaunchKernelGGL((fbank512_wave_kernel<T, 12, 1, FLAVOR, NSLOTS, Lens>), dim3(grid12), dim3(768), lds, s, fp);
        HIP_TRY(hipGetLastError());
        return MELSPEC_OK;
    } else {                  // (an else: the f32 instantiation must not instantiate the eight- and four-wave kernels it never launches)
        const bool runs = kCanRun && plain && waves == 8;
        static std::atomic<uint64_t> attr_done{0};          // one bit per device: function attributes are per device
        if (!device_done(attr_done)) {
            int rc = allow_big_lds(&fbank512_wave_kernel<T, 8, 1, FLAVOR, NSLOTS, Lens>, "hipFuncSetAttribute(fbank512_wave_kernel, 8 waves)");
            if (!rc) rc = allow_big_lds(&fbank512_wave_kernel<T, 4, 1, FLAVOR, NSLOTS, LensRuntime>, "hipFuncSetAttribute(fbank512_wave_kernel, 4 waves)");
            if (!rc && kCanRun) rc = allow_big_lds(&fbank512_wave_kernel<T, 8, 1, FLAVOR, NSLOTS, Lens, kCanRun>, "hipFuncSetAttribute(fbank512_wave_kernel, runs)");
            if (rc) return rc;
            mark_device_done(attr_done);
        }
        const uint64_t blocks = (fp.b.n_units + waves - 1) / waves;
        static const int per_cu = lab_int("MELSPEC_FB_GRID_PER_CU", 1, 1, 4096);   // one workgroup is resident per CU; measured best
        const unsigned grid = grid_for_xcd(blocks, cus, per_cu);
        if (runs)
            hipLaunchKernelGGL((fbank512_wave_kernel<T, 8, 1, FLAVOR, NSLOTS, Lens, kCanRun>), dim3(grid), dim3(512), lds, s, fp);
        else if (waves == 8)
            hipLaunchKernelGGL((fbank512_wave_kernel<T, 8, 1, FLAVOR, NSLOTS, Lens>), dim3(grid), dim3(512), lds, s, fp);
        else
            hipLaunchKernelGGL((fbank512_wave_kernel<T, 4, 1, FLAVOR, NSLOTS, LensRuntime>), dim3(grid), dim3(256), lds, s, fp);
        HIP_TRY(hipGetLastError());
        return MELSPEC_OK;
    }
}

void f32_params(const Fused512F32 &f, FbankFastParams &fp) {
    fp.d_blob = static_cast<const uint32_t *>(f.d_blob.p);
    fp.blob_words = static_cast<int>(f.ft.blob.size());
    fp.mel_off_words = f.ft.mel_off_words;
    fp.slots = f.ft.slots;
}
int launch_w512_f32(const Fused512F32 &f, FbankFastParams fp, int cus, hipStream_t s) {
    f32_params(f, fp);
    if (fb_lens_match<LensSlaney80W>(f.ft.slots)) return launch_fused512<float, kFlavorWhisper, kFbSlots, LensSlaney80W>(kFused512F32Waves, fp, f.lds, cus, s);
    return launch_fused512<float, kFlavorWhisper, kBlmSlots, LensSlaney128>(kFused512F32Waves, fp, f.lds, cus, s);
}
int launch_nemo_f32(const Fused512F32 &f, FbankFastParams fp, int cus, hipStream_t s) {
    f32_params(f, fp);
    fp.b.sync_rounds = 0;        // StagedRows instead of RoundSync
    if (fb_lens_match<LensSlaney128>(f.ft.slots)) return launch_fused512<float, kFlavorNemo, kBlmSlots, LensSlaney128>(kFused512F32Waves, fp, f.lds, cus, s);
    return launch_fused512<float, kFlavorNemo, kFbSlots, LensSlaney80>(kFused512F32Waves, fp, f.lds, cus, s);
}
}  // namespace

namespace {
// MELSPEC_PRECISION_AUTO at n_fft = 512: the voting f32 launch and, gated on its verdict, the f64 launch (w512_auto_kernel)
template <int NSLOTS, class Lens>
int launch_w512_auto_t(melspec_ctx *c, const FbankFastParams &fp64, const BatchDesc &desc, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&w512_auto_kernel<float, kFused512F32Waves, NSLOTS, Lens>, "hipFuncSetAttribute(w512_auto_kernel<float>)");
        if (!rc) rc = allow_big_lds(&w512_auto_kernel<double, 8, NSLOTS, Lens>, "hipFuncSetAttribute(w512_auto_kernel<double>)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    FixSink sink{};
    int rc = auto_sink(c, desc, stream, true, sink);
    if (rc) return rc;
    sink.tab = nullptr;
    const unsigned grid32 = grid_for_xcd((desc.n_units + kFused512F32Waves - 1) / kFused512F32Waves, c->dev.cus, 1);
    W512AutoParams q{};
    q.f = fp64;
    f32_params(c->f512, q.f);
    q.fix = sink_armed(c, sink, desc, grid32);
    q.fix.vote_groups = std::min<unsigned>(grid32, static_cast<unsigned>(c->dev.cus));
    hipLaunchKernelGGL((w512_auto_kernel<float, kFused512F32Waves, NSLOTS, Lens>), dim3(grid32), dim3(kFused512F32Waves * 64), c->f512.lds, stream, q);
    HIP_TRY(hipGetLastError());
    // the gated launch: this batch's verdict (its number is c->fix.seq) decides between every unit, the noted units and nothing
    const unsigned grid64 = grid_for_xcd((desc.n_units + 7) / 8, c->dev.cus, 1);
    W512AutoParams g{};
    g.f = fp64;
    FixSink stat{};
    stat.count = sink.count; stat.acc = sink.acc; stat.host = sink.host; stat.list = sink.list;
    g.fix = sink_armed(c, stat, desc, grid64);          // (a launch number of its own: the host tells the two reports apart by kStatFromGated)
    g.fix.frames |= kStatFromGated;
    g.gate = sink.decision;
    g.gate_seq = q.fix.seq;
    hipLaunchKernelGGL((w512_auto_kernel<double, 8, NSLOTS, Lens>), dim3(grid64), dim3(512), c->lds512, stream, g);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}
}  // namespace

namespace melspec {
namespace host {
// can a plain batch of this n_fft = 512 context run in MELSPEC_PRECISION_AUTO's f32 / f64 pair?  (the compile-time banks, eight f64 waves)
bool w512_auto_ok(const melspec_ctx *c) {
    return c->fast512 && c->f512.ok && c->waves512 == 8 && c->fix.count.p != nullptr &&
           (fb_lens_match<LensSlaney80W>(c->ft512.slots) || fb_lens_match<LensSlaney128>(c->ft512.slots));
}

// the Whisper flavour of the 512-point kernel: launch_ctx's branch for the n_fft = 512 contexts
int launch_whisper512(melspec_ctx *c, const BatchDesc &desc, hipStream_t stream) {
        FbankFastParams fp{};
        fp.b = desc;
        fp.d_blob = static_cast<const uint32_t *>(c->d_blob512.p);
        fp.blob_words = static_cast<int>(c->ft512.blob.size());
        fp.mel_off_words = c->ft512.mel_off_words;
        fp.shift = c->hop_size;
        fp.n_mels = c->n_mels;
        fp.use_log = 1; fp.use_power = 1;
        fp.slots = c->ft512.slots;
        if (c->precision == MELSPEC_PRECISION_F32 && c->f512.ok) return launch_w512_f32(c->f512, fp, c->dev.cus, stream);
        // AUTO (round 6): plain batches -- uniform and ragged -- vote like the n_fft = 400 contexts do; the layouts stay on the f64 kernel
        const bool plain = !desc.mel_major && (desc.d_unit_prefix != nullptr || desc.out_width == desc.frames_per_clip);
        // ... when there is a batch to speak of: below two units per f32 wave of the grid (6144 units = 24 576 frames on an MI355X) a call is
        // launch-bound, the pair of launches is slower than the f64 kernel alone, and the f64 kernel it is (also what keeps the streaming
        // bank's hop-sized pushes on the reference's golden within 1e-6; a rule on the batch's size: still a function of the batch alone)
        const bool sizeable = desc.n_units >= 2ull * kFused512F32Waves * static_cast<unsigned>(c->dev.cus);
        if (c->precision == MELSPEC_PRECISION_AUTO && c->fix.adaptive && plain && sizeable && w512_auto_ok(c))
            return fb_lens_match<LensSlaney80W>(c->ft512.slots) ? launch_w512_auto_t<kFbSlots, LensSlaney80W>(c, fp, desc, stream)
                                                                : launch_w512_auto_t<kBlmSlots, LensSlaney128>(c, fp, desc, stream);
        if (fb_lens_match<LensSlaney80W>(c->ft512.slots)) return launch_fused512<double, kFlavorWhisper, kFbSlots, LensSlaney80W>(c->waves512, fp, c->lds512, c->dev.cus, stream);
        if (fb_lens_match<LensSlaney128>(c->ft512.slots)) return launch_fused512<double, kFlavorWhisper, kBlmSlots, LensSlaney128>(c->waves512, fp, c->lds512, c->dev.cus, stream);
        return c->ft512.slots.n_slots <= kFbSlots ? launch_fused512<double, kFlavorWhisper, kFbSlots>(c->waves512, fp, c->lds512, c->dev.cus, stream)
                                                  : launch_fused512<double, kFlavorWhisper, kBlmSlots>(c->waves512, fp, c->lds512, c->dev.cus, stream);
}
}  // namespace host
}  // namespace melspec

// ------------------------------------------------------------------------------------
// Kaldi fbank context
// ------------------------------------------------------------------------------------
struct melspec_fbank {
    DeviceInfo dev;
    melspec_fbank_config cfg{};
    int frame_len = 0, frame_shift = 0, fft_size = 0;
    hipStream_t stream = nullptr;
    bool fast = false;          // fused 512-point kernel (default Kaldi geometry) vs generic f64 kernel
    bool use_generic = false;   // melspec_fbank_use_generic: the direct-DFT kernel as the on-device cross-check
    RaggedScratch ragged;
    DevicePlan dplan;
    HostPipe pipe;              // melspec_fbank_compute_batch_host
    FbankFastTables ft;
    DevBuf d_blob;
    size_t fast_lds = 0;
    int waves = 4;
    GenericTables gt;
    DevBuf h2d, d2h;
};


namespace {
uint64_t fbank_frames(const melspec_fbank *fb, uint64_t n) {
    return n < static_cast<uint64_t>(fb->frame_len) ? 0 : 1 + (n - fb->frame_len) / fb->frame_shift;   // src/fbank.rs:147-151
}
}  // namespace

extern "C" {

void melspec_fbank_default_config(melspec_fbank_config *c) {
    if (!c) return;
    c->sample_rate = 16000.0; c->num_mel_bins = 80; c->frame_length_ms = 25.0; c->frame_shift_ms = 10.0;
    c->energy_floor = 0.0; c->use_log_fbank = 1; c->use_power = 1; c->preemphasis = 0.97; c->apply_cmn = 1;
    c->low_freq = 20.0; c->high_freq = 0.0;
}

int melspec_fbank_create(melspec_fbank **out, int device, const melspec_fbank_config *cfg) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!cfg) return fail(MELSPEC_ERR_INVALID_ARG, "cfg is NULL");
    if (!(cfg->sample_rate > 0.0) || cfg->num_mel_bins <= 0 || !(cfg->frame_length_ms > 0.0) || !(cfg->frame_shift_ms > 0.0))
        return fail(MELSPEC_ERR_INVALID_ARG, "sample_rate, num_mel_bins, frame length and shift must be positive");
    // FbankConfig::{frame_length_samples, frame_shift_samples, fft_size}  (src/fbank.rs:66-82)
    const int frame_len = static_cast<int>(std::llround((cfg->frame_length_ms / 1000.0) * cfg->sample_rate));
    const int frame_shift = static_cast<int>(std::llround((cfg->frame_shift_ms / 1000.0) * cfg->sample_rate));
    if (frame_len < 2 || frame_shift < 1) return fail(MELSPEC_ERR_INVALID_ARG, "frame length/shift round to zero samples");
    int fft_size = 1;
    while (fft_size < frame_len) fft_size <<= 1;
    if (fft_size > kMaxGenericFft || cfg->num_mel_bins > kMaxGenericMels)
        return fail(MELSPEC_ERR_UNSUPPORTED, "fft_size must be <= 4096 and num_mel_bins <= 1024");
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    melspec_fbank *fb = new (std::nothrow) melspec_fbank();
    if (!fb) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    fb->dev = info; fb->cfg = *cfg; fb->frame_len = frame_len; fb->frame_shift = frame_shift; fb->fft_size = fft_size;
    auto bail = [&](int code) { melspec_fbank_destroy(fb); return code; };
    if (hipSetDevice(info.device) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipSetDevice failed"));
    if (hipStreamCreate(&fb->stream) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipStreamCreate failed"));
    const double high = cfg->high_freq == 0.0 ? cfg->sample_rate / 2.0 : cfg->high_freq;
    const int bins = fft_size / 2 + 1;
    // the fused kernel computes in f64 up to |X|^2 (an f32 build cannot hold 1e-4 on quiet mel bands, see fbank_wave.hpp)
    fb->fast = frame_len == 400 && fft_size == 512 &&
               build_fbank_fast_tables<double>(cfg->sample_rate, cfg->num_mel_bins, cfg->low_freq, high, cfg->use_power != 0, fb->ft);
    if (fb->fast) {
        const size_t slice_bytes = FbankLayout<double>::slice_elems() * sizeof(double);
        fb->waves = fused512_waves(fb->ft.blob.size() * 4, slice_bytes);
        fb->fast_lds = fb->ft.blob.size() * 4 + static_cast<size_t>(fb->waves) * slice_bytes;
        if (fb->fast_lds > kLdsLimit) fb->fast = false;
    }
    if (fb->fast && (rc = upload(fb->d_blob, fb->ft.blob))) return bail(rc);
    const std::vector<double> dense = kaldi_mel_filterbank(cfg->sample_rate, fft_size, cfg->num_mel_bins, cfg->low_freq, high);
    if ((rc = fb->gt.build(fft_size, frame_len, bins, povey_window(frame_len), dense, cfg->num_mel_bins, bins))) return bail(rc);
    if (fb->gt.lds_bytes > kLdsLimit) return bail(fail(MELSPEC_ERR_UNSUPPORTED, "geometry needs more LDS than one workgroup has"));
    if ((rc = generic_allow_lds())) return bail(rc);
    *out = fb;
    return MELSPEC_OK;
}

void melspec_fbank_destroy(melspec_fbank *fb) {
    if (!fb) return;
    if (fb->dev.device >= 0) (void)hipSetDevice(fb->dev.device);
    if (fb->stream) { (void)hipStreamSynchronize(fb->stream); (void)hipStreamDestroy(fb->stream); }
    fb->gt.release(); fb->d_blob.release(); fb->h2d.release(); fb->d2h.release(); fb->ragged.release(); fb->dplan.release(); fb->pipe.release();
    delete fb;
}

size_t melspec_fbank_num_frames(const melspec_fbank *fb, size_t n_samples) {
    return fb ? static_cast<size_t>(fbank_frames(fb, n_samples)) : 0;
}
int melspec_fbank_num_mel_bins(const melspec_fbank *fb) { return fb ? fb->cfg.num_mel_bins : 0; }
int melspec_fbank_uses_fast_path(const melspec_fbank *fb) { return fb && fb->fast && !fb->use_generic ? 1 : 0; }
int melspec_fbank_use_generic(melspec_fbank *fb, int on) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    fb->use_generic = on != 0;
    fb->gt.force_generic = on == 2;      // 2: the workgroup-per-frame kernel also where pow2_frame_kernel would take the geometry
    return MELSPEC_OK;
}

static int fbank_launch(melspec_fbank *fb, const BatchPlan &pl, uint32_t n_clips, uint64_t fpc, hipStream_t s, float *d_means = nullptr);

int melspec_fbank_compute_uniform_device(melspec_fbank *fb, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                         uint32_t n_clips, float *d_out, void *stream) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    if (n_clips == 0) return MELSPEC_OK;
    const uint64_t fpc = fbank_frames(fb, clip_len);
    if (fpc == 0) return MELSPEC_OK;   // zeros((0, num_mel_bins)), src/fbank.rs:147-149
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(fb->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : fb->stream;
    const int nm = fb->cfg.num_mel_bins;
    const bool fused = fb->fast && !fb->use_generic;
    const BatchPlan pl = plan_uniform(d_pcm, d_out, clip_stride, fpc, n_clips, nm, fused ? kFbFPW : 1);
    return fbank_launch(fb, pl, n_clips, fpc, s);
}

int melspec_fbank_compute_uniform_device_split(melspec_fbank *fb, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                               uint32_t n_clips, float *d_rows, float *d_means, void *stream) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    if (!fb->cfg.apply_cmn) return fail(MELSPEC_ERR_INVALID_ARG, "the split output is the CMN's two halves: FbankConfig::apply_cmn is off");
    if (n_clips == 0) return MELSPEC_OK;
    const uint64_t fpc = fbank_frames(fb, clip_len);
    if (!d_means) return fail(MELSPEC_ERR_INVALID_ARG, "d_means is NULL");
    HIP_TRY(hipSetDevice(fb->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : fb->stream;
    const int nm = fb->cfg.num_mel_bins;
    if (fpc == 0) {                     // zeros((0, num_mel_bins)): no rows; the mean of nothing is reported as 0
        HIP_TRY(hipMemsetAsync(d_means, 0, static_cast<size_t>(n_clips) * nm * sizeof(float), s));
        return MELSPEC_OK;
    }
    if (!d_pcm || !d_rows) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    const bool fused = fb->fast && !fb->use_generic;
    const BatchPlan pl = plan_uniform(d_pcm, d_rows, clip_stride, fpc, n_clips, nm, fused ? kFbFPW : 1);
    return fbank_launch(fb, pl, n_clips, fpc, s, d_means);
}

// kernels of one batch (uniform or ragged plan): fused 512-point kernel or the generic one, then CMN per clip
static int fbank_launch(melspec_fbank *fb, const BatchPlan &pl, uint32_t n_clips, uint64_t fpc /* frames of the longest clip (LDS budget of the CMN) */, hipStream_t s,
                        float *d_means /* not nullptr: the split output -- un-normalised rows + the clips' column means */) {
    const int nm = fb->cfg.num_mel_bins;
    const bool fused = fb->fast && !fb->use_generic;
    const double floor_v = fb->cfg.energy_floor > 0.0 ? fb->cfg.energy_floor : static_cast<double>(FLT_EPSILON);
    int rc = MELSPEC_OK;
    if (fused) {
        FbankFastParams fp{};
        fp.b = pl.desc;
        fp.d_blob = static_cast<const uint32_t *>(fb->d_blob.p);
        fp.blob_words = static_cast<int>(fb->ft.blob.size());
        fp.mel_off_words = fb->ft.mel_off_words;
        fp.shift = fb->frame_shift;
        fp.n_mels = nm;
        fp.preemph = fb->cfg.preemphasis > 0.0 ? fb->cfg.preemphasis : 0.0;   // src/fbank.rs:172
        fp.floor_v = static_cast<float>(floor_v);
        fp.use_log = fb->cfg.use_log_fbank;
        fp.use_power = fb->cfg.use_power;
        fp.slots = fb->ft.slots;
        // many clips of one length + CMN: the workgroup-per-clip kernel with the normalisation inside (fbank512_clip_kernel) when the
        // clips fill the CUs evenly enough to beat the two-kernel path's 1.29 x (lab builds: MELSPEC_FB_CLIP=0 keeps the two kernels)
        static const bool clip_on = lab_int("MELSPEC_FB_CLIP", 1, 0, 1) != 0;
        const uint32_t cus = static_cast<uint32_t>(fb->dev.cus);
        const uint32_t passes = (n_clips + cus - 1) / cus;
        const bool ragged_by_clip = pl.desc.d_order != nullptr;       // melspec_fbank_compute_ragged_device decided (and checked the alignment)
        if (clip_on && (ragged_by_clip ||
            (fb->cfg.apply_cmn && fb->cfg.use_power && fb->waves == 8 && pl.desc.d_unit_prefix == nullptr && nm % 4 == 0 && nm <= 89 &&
             (reinterpret_cast<uintptr_t>(pl.desc.out) & 15) == 0 && pl.desc.out_stride % 4 == 0 &&
             n_clips >= cus && static_cast<uint64_t>(n_clips) * 100 >= static_cast<uint64_t>(passes) * cus * 85))) {
            static std::atomic<uint64_t> attr_done{0};
            if (!device_done(attr_done)) {
                rc = allow_big_lds(&fbank512_clip_kernel<kFbSlots, LensKaldi80>, "hipFuncSetAttribute(fbank512_clip_kernel)");
                if (!rc) rc = allow_big_lds(&fbank512_clip_kernel<kFbSlots, LensRuntime>, "hipFuncSetAttribute(fbank512_clip_kernel)");
                if (!rc) rc = allow_big_lds(&fbank512_clip_kernel<kFbSlots, LensKaldi80, true>, "hipFuncSetAttribute(fbank512_clip_kernel)");
                if (!rc) rc = allow_big_lds(&fbank512_clip_kernel<kFbSlots, LensRuntime, true>, "hipFuncSetAttribute(fbank512_clip_kernel)");
                if (!rc) rc = allow_big_lds(&fbank512_clip_kernel<kFbSlots, LensKaldi40>, "hipFuncSetAttribute(fbank512_clip_kernel)");
                if (!rc) rc = allow_big_lds(&fbank512_clip_kernel<kFbSlots, LensKaldi40, true>, "hipFuncSetAttribute(fbank512_clip_kernel)");
                if (rc) return rc;
                mark_device_done(attr_done);
            }
            FbankClipParams q{};
            q.f = fp;
            q.frames = fpc;
            static const int clip_skip = lab_int("MELSPEC_FB_CLIP_SKIP", 0, 0, 15);
            q.lab_skip = clip_skip;
            q.d_means = d_means;
            const size_t lds = fb->fast_lds + sizeof(ClipCmnShared<8>);
            if (lds <= kLdsLimit) {
                const bool k80 = fb_lens_match<LensKaldi80>(fb->ft.slots), k40 = fb_lens_match<LensKaldi40>(fb->ft.slots);
                if (ragged_by_clip) {
                    if (k80) hipLaunchKernelGGL((fbank512_clip_kernel<kFbSlots, LensKaldi80, true>), dim3(cus), dim3(512), lds, s, q);
                    else if (k40) hipLaunchKernelGGL((fbank512_clip_kernel<kFbSlots, LensKaldi40, true>), dim3(cus), dim3(512), lds, s, q);
                    else hipLaunchKernelGGL((fbank512_clip_kernel<kFbSlots, LensRuntime, true>), dim3(cus), dim3(512), lds, s, q);
                } else if (k80) hipLaunchKernelGGL((fbank512_clip_kernel<kFbSlots, LensKaldi80>), dim3(cus), dim3(512), lds, s, q);
                else if (k40) hipLaunchKernelGGL((fbank512_clip_kernel<kFbSlots, LensKaldi40>), dim3(cus), dim3(512), lds, s, q);
                else hipLaunchKernelGGL((fbank512_clip_kernel<kFbSlots, LensRuntime>), dim3(cus), dim3(512), lds, s, q);
                HIP_TRY(hipGetLastError());
                return MELSPEC_OK;
            }
        }
        if (fb_lens_match<LensKaldi80>(fb->ft.slots))
            rc = launch_fused512<double, kFlavorKaldi, kFbSlots, LensKaldi80>(fb->waves, fp, fb->fast_lds, fb->dev.cus, s);
        else if (fb_lens_match<LensKaldi40>(fb->ft.slots))
            rc = launch_fused512<double, kFlavorKaldi, kFbSlots, LensKaldi40>(fb->waves, fp, fb->fast_lds, fb->dev.cus, s);
        else
            rc = launch_fused512<double, kFlavorKaldi, kFbSlots>(fb->waves, fp, fb->fast_lds, fb->dev.cus, s);
        if (rc) return rc;
        // the CMN pass below walks clips, not units
    } else {
        rc = launch_generic(fb->gt, pl.desc, fb->frame_shift, 1, fb->cfg.use_log_fbank, fb->cfg.use_power,
                            fb->cfg.preemphasis, floor_v, fb->dev.cus, s);
    }
    if (rc) return rc;
    if (fb->cfg.apply_cmn) {
        CmnParams cp{};
        cp.b = pl.desc;
        cp.n_mels = nm;
        // rows staged per chunk (a multiple of 4: the fold works on units of 4 frames): what fits one workgroup's LDS next to the means and
        // the run sums; two workgroups per CU when a whole clip fits half of it
        const size_t head = static_cast<size_t>((nm + 3) & ~3) * 9 * sizeof(float);
        const bool staged = nm <= 512;
        size_t budget = kLdsLimit - head - 256;
        if (fpc * static_cast<uint64_t>(nm) * sizeof(float) + head <= kLdsLimit / 2 - 256) budget = kLdsLimit / 2 - head - 256;
        uint64_t rows = (budget / (static_cast<size_t>(nm) * sizeof(float))) & ~3ull;
        if (rows > ((fpc + 3) & ~3ull)) rows = (fpc + 3) & ~3ull;
        cp.rows_per_chunk = staged ? static_cast<int>(rows) : 0;
        cp.d_means = d_means;
        static std::atomic<uint64_t> cmn_attr{0};
        if (!device_done(cmn_attr)) {
            if ((rc = allow_big_lds(&cmn_kernel<512>, "hipFuncSetAttribute(cmn_kernel)"))) return rc;
            mark_device_done(cmn_attr);
        }
        const size_t lds = staged ? head + static_cast<size_t>(cp.rows_per_chunk) * nm * sizeof(float)
                                  : (static_cast<size_t>((nm + 3) & ~3) + 8 * 512 + 512) * sizeof(float);
        const unsigned grid = grid_for(n_clips, fb->dev.cus, 8);
        hipLaunchKernelGGL(cmn_kernel<512>, dim3(grid), dim3(512), lds, s, cp);
        HIP_TRY(hipGetLastError());
    }
    return MELSPEC_OK;
}

// Fbank::compute per clip of any length (src/fbank.rs:141): clip c = d_pcm[h_offsets[c] .. + h_lengths[c]) -> its frames at
// d_out + h_out_offsets[c] floats (NULL: packed); CMN per clip.
int melspec_fbank_compute_ragged_device(melspec_fbank *fb, const float *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths,
                                        uint32_t n_clips, float *d_out, const uint64_t *h_out_offsets, void *stream) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    if (n_clips == 0) return MELSPEC_OK;
    if (!h_offsets || !h_lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    std::vector<uint64_t> frames(n_clips);
    uint64_t total = 0, longest = 0;
    for (uint32_t i = 0; i < n_clips; ++i) { frames[i] = fbank_frames(fb, h_lengths[i]); total += frames[i]; longest = std::max(longest, frames[i]); }
    if (total == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(fb->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : fb->stream;
    const bool fused = fb->fast && !fb->use_generic;
    BatchPlan pl;
    RaggedSlot *slot = nullptr;
    // whole clips per workgroup (fbank512_clip_kernel) when the batch can keep every CU busy: at least two clips per CU and no clip
    // longer than half a CU's share; outputs at 16-byte offsets (packed outputs of n_mels % 4 == 0 are)
    const int nm = fb->cfg.num_mel_bins;
    bool by_clip = fused && fb->cfg.apply_cmn && fb->cfg.use_power && fb->waves == 8 && nm % 4 == 0 && nm <= 89 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 &&
                   n_clips >= 2u * static_cast<uint32_t>(fb->dev.cus) && longest * 2 * static_cast<uint64_t>(fb->dev.cus) <= total && longest < (1ull << 31);
    if (by_clip && h_out_offsets)
        for (uint32_t i = 0; i < n_clips && by_clip; ++i) by_clip = h_out_offsets[i] % 4 == 0;
    int rc = plan_ragged(fb->ragged, s, d_pcm, d_out, h_offsets, frames, h_out_offsets, n_clips, nm, fused ? kFbFPW : 1, pl, slot, by_clip);
    if (!rc) rc = fbank_launch(fb, pl, n_clips, longest, s);
    plan_ragged_done(slot, s);
    return rc;
}

// The same with the clip table in device memory (see melspec_compute_ragged_device_desc).
int melspec_fbank_compute_ragged_device_desc(melspec_fbank *fb, const float *d_pcm, const uint64_t *d_offsets, const uint64_t *d_lengths,
                                             uint32_t n_clips, float *d_out, const uint64_t *d_out_offsets, uint64_t max_total_frames,
                                             void *stream) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    if (n_clips == 0 || max_total_frames == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out || !d_offsets || !d_lengths) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(fb->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : fb->stream;
    const bool fused = fb->fast && !fb->use_generic;
    BatchPlan pl;
    int rc = plan_ragged_device(fb->dplan, s, d_pcm, d_out, d_offsets, d_lengths, d_out_offsets, n_clips, static_cast<uint64_t>(fb->frame_len),
                                static_cast<uint64_t>(fb->frame_shift), static_cast<uint32_t>(fb->cfg.num_mel_bins), fused ? kFbFPW : 1,
                                max_total_frames, pl);
    if (rc) return rc;
    return fbank_launch(fb, pl, n_clips, max_total_frames, s);
}

int melspec_fbank_release_scratch(melspec_fbank *fb) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    HIP_TRY(hipSetDevice(fb->dev.device));
    HIP_TRY(hipStreamSynchronize(fb->stream));
    fb->pipe.release(); fb->ragged.release(); fb->dplan.release(); fb->h2d.release(); fb->d2h.release();
    return MELSPEC_OK;
}

int melspec_fbank_synchronize(melspec_fbank *fb, void *stream) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    HIP_TRY(hipSetDevice(fb->dev.device));
    HIP_TRY(hipStreamSynchronize(stream ? static_cast<hipStream_t>(stream) : fb->stream));
    return MELSPEC_OK;
}

int melspec_fbank_compute_host(melspec_fbank *fb, const float *samples, size_t n_samples, float *out,
                               size_t out_capacity_floats, size_t *n_frames) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    if (n_frames) *n_frames = 0;
    const uint64_t frames = fbank_frames(fb, n_samples);
    if (frames == 0) return MELSPEC_OK;
    if (!samples || !out) return fail(MELSPEC_ERR_INVALID_ARG, "samples/out is NULL");
    const uint64_t need = frames * static_cast<uint64_t>(fb->cfg.num_mel_bins);
    if (out_capacity_floats < need) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(fb->dev.device));
    int rc;
    if ((rc = fb->h2d.ensure(n_samples * sizeof(float)))) return rc;
    if ((rc = fb->d2h.ensure(need * sizeof(float)))) return rc;
    HIP_TRY(hipMemcpyAsync(fb->h2d.p, samples, n_samples * sizeof(float), hipMemcpyHostToDevice, fb->stream));
    rc = melspec_fbank_compute_uniform_device(fb, static_cast<const float *>(fb->h2d.p), n_samples, n_samples, 1,
                                              static_cast<float *>(fb->d2h.p), fb->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, fb->d2h.p, need * sizeof(float), hipMemcpyDeviceToHost, fb->stream));
    HIP_TRY(hipStreamSynchronize(fb->stream));
    if (n_frames) *n_frames = static_cast<size_t>(frames);
    return MELSPEC_OK;
}

// Fbank::compute on many host clips in one call: whole clips (the CMN is per clip) in chunks of ~16 MiB of PCM through the pinned,
// double-buffered pipeline of host_pipe.hpp, one ragged launch per chunk.
int melspec_fbank_compute_batch_host(melspec_fbank *fb, const float *samples, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_clips,
                                     float *out, const uint64_t *out_offsets, size_t out_capacity_floats, uint64_t *total_frames) {
    if (!fb) return fail(MELSPEC_ERR_INVALID_ARG, "fbank is NULL");
    if (total_frames) *total_frames = 0;
    if (n_clips == 0) return MELSPEC_OK;
    if (!offsets || !lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    const uint64_t nm = static_cast<uint64_t>(fb->cfg.num_mel_bins);
    std::vector<HostSeg> segs;
    segs.reserve(n_clips);
    uint64_t total = 0, cursor = 0;
    for (uint32_t i = 0; i < n_clips; ++i) {
        const uint64_t f = fbank_frames(fb, lengths[i]);
        const uint64_t oo = out_offsets ? out_offsets[i] : cursor;
        if (f && oo + f * nm > out_capacity_floats) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
        if (f && (!samples || !out)) return fail(MELSPEC_ERR_INVALID_ARG, "samples/out is NULL");
        if (f) segs.push_back(HostSeg{samples + offsets[i], lengths[i], out + oo, f});
        cursor += f * nm; total += f;
    }
    if (total_frames) *total_frames = total;
    if (total == 0) return MELSPEC_OK;
    HIP_TRY(hipSetDevice(fb->dev.device));
    const char *where = "";
    const int rc = fb->pipe.run(segs, fb->cfg.num_mel_bins, kPipeChunkSamples, fb->stream,
                                [fb](const float *d_in, const uint64_t *offs, const uint64_t *lens, uint32_t n, float *d_out,
                                     const uint64_t *ooffs, hipStream_t s) {
                                    return melspec_fbank_compute_ragged_device(fb, d_in, offs, lens, n, d_out, ooffs, s);
                                }, &where);
    if (rc > 0 && where[0] && std::strcmp(where, "kernel launch") != 0) return fail_hip(static_cast<hipError_t>(rc), where);
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// NeMo / Parakeet frontend context (BatchLogMelSpectrogram, src/mel.rs:239-396)
// ------------------------------------------------------------------------------------
struct melspec_blm {
    DeviceInfo dev;
    melspec_blm_config cfg{};
    hipStream_t stream = nullptr;
    bool fast = false;          // fused 512-point kernel (n_fft 512 / win_length 400) vs the generic f64 kernel (any validated config)
    GenericTables gt;
    RaggedScratch ragged;
    DevBuf aux;                 // ragged batches: per-clip sample counts and valid frames, the normaliser's group counter; used in
    hipStream_t aux_stream = nullptr;   //   stream order (a call on another stream first waits for the stream that used it last)
    bool aux_used = false;
    FbankFastTables ft;
    DevBuf d_blob;
    size_t fast_lds = 0;
    int waves = 4;
    int precision = MELSPEC_PRECISION_AUTO;     // melspec_blm_set_precision
    Fused512F32 f32;            // MELSPEC_PRECISION_F32: the reference's own arithmetic type for this frontend (src/mel.rs:251-252,356-357)
    DevBuf h2d, d2h;
    HostPipe pipe;              // melspec_blm_compute_batch_host
};

namespace {
uint64_t blm_valid_frames(const melspec_blm *b, uint64_t n) {       // src/mel.rs:326-332,387-395
    if (n == 0) return 0;
    if (b->cfg.center) return n / b->cfg.hop_length + 1;
    if (n < static_cast<uint64_t>(b->cfg.n_fft)) return 0;
    return (n - b->cfg.n_fft) / b->cfg.hop_length + 1;
}
uint64_t blm_padded(const melspec_blm *b, uint64_t frames) {         // pad_len, src/mel.rs:751-756
    const uint64_t p = b->cfg.pad_to;
    return p == 0 ? frames : (frames + p - 1) / p * p;
}
}  // namespace

extern "C" {

void melspec_blm_default_config(melspec_blm_config *c) {
    if (!c) return;
    c->sample_rate = 16000; c->n_fft = 512; c->win_length = 400; c->hop_length = 160; c->n_mels = 80;
    c->f_min = 0.0; c->f_max = -1.0; c->htk = 0; c->norm = 1; c->preemphasis = 0.0f; c->center = 1;
    c->log_zero_guard = FLT_EPSILON; c->pad_to = 0; c->normalize_per_feature = 0;
}

int melspec_blm_create(melspec_blm **out, int device, const melspec_blm_config *cfg) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!cfg) return fail(MELSPEC_ERR_INVALID_ARG, "cfg is NULL");
    // validate_batch_config (src/mel.rs:656-683), same order and messages
    if (cfg->sample_rate <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: sample_rate must be > 0");
    if (cfg->n_fft <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: n_fft must be > 0");
    if (cfg->win_length <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: win_length must be > 0");
    if (cfg->win_length > cfg->n_fft) return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: win_length must be <= n_fft");
    if (cfg->hop_length <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: hop_length must be > 0");
    if (cfg->n_mels <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: n_mels must be > 0");
    if (!std::isfinite(cfg->log_zero_guard) || cfg->log_zero_guard <= 0.0f)
        return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: log_zero_guard must be finite and > 0");
    if (cfg->pad_to < 0) return fail(MELSPEC_ERR_INVALID_ARG, "invalid log-mel config: pad_to must be >= 0");
    if (cfg->n_fft > kMaxGenericFft || cfg->n_mels > kMaxGenericMels)
        return fail(MELSPEC_ERR_UNSUPPORTED, "n_fft must be <= 4096 and n_mels <= 1024");
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    melspec_blm *b = new (std::nothrow) melspec_blm();
    if (!b) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    b->dev = info; b->cfg = *cfg;
    auto bail = [&](int code) { melspec_blm_destroy(b); return code; };
    if (hipSetDevice(info.device) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipSetDevice failed"));
    if (hipStreamCreate(&b->stream) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipStreamCreate failed"));
    const double f_max = cfg->f_max > 0.0 ? cfg->f_max : cfg->sample_rate / 2.0;   // src/mel.rs:254
    // the NeMo / Parakeet geometry (n_fft 512, win_length 400) runs on the fused 512-point kernel; every other validated config
    // (src/mel.rs:248-280 accepts them all) on the generic f64 kernel
    b->fast = cfg->n_fft == 512 && cfg->win_length == 400 &&
              build_blm_fast_tables<double>(cfg->sample_rate, cfg->n_mels, cfg->f_min, f_max, cfg->htk != 0, cfg->norm != 0, b->ft);
    if (b->fast) {
        const size_t slice_bytes = FbankLayout<double>::slice_elems() * sizeof(double);
        b->waves = fused512_waves(b->ft.blob.size() * 4, slice_bytes);
        b->fast_lds = b->ft.blob.size() * 4 + static_cast<size_t>(b->waves) * slice_bytes + 64;      // + RoundSync counters
        if (b->fast_lds > kLdsLimit) b->fast = false;
    }
    if (b->fast) {
        if ((rc = upload(b->d_blob, b->ft.blob))) return bail(rc);
        if (nemo_f32_bank(b->ft.slots) && build_blm_fast_tables<float>(cfg->sample_rate, cfg->n_mels, cfg->f_min, f_max, cfg->htk != 0, cfg->norm != 0, b->f32.ft) &&
            (rc = b->f32.finish(0, StagedRows<kFused512F32Waves>::bytes(cfg->n_mels)))) return bail(rc);
    } else {
        // the reference's f32 tables: symmetric Hann(win_length) centred in the n_fft frame (src/mel.rs:708-719), f32 weights
        const int N = cfg->n_fft, bins = N / 2 + 1;
        std::vector<double> win(static_cast<size_t>(N), 0.0);
        if (cfg->win_length > 1) {
            const int offset = (N - cfg->win_length) / 2;
            const float pi_f32 = 3.14159265358979323846f;
            for (int i = 0; i < cfg->win_length; ++i) {
                const float phase = (2.0f * pi_f32 * static_cast<float>(i)) / (static_cast<float>(cfg->win_length) - 1.0f);
                win[offset + i] = static_cast<double>(0.5f - (0.5f * std::cos(phase)));
            }
        }
        std::vector<double> dense = mel_filterbank(static_cast<double>(cfg->sample_rate), N, cfg->n_mels, cfg->f_min > 0.0 ? cfg->f_min : -1.0, f_max,
                                                   cfg->htk != 0, cfg->norm != 0);
        for (double &w : dense) w = static_cast<double>(static_cast<float>(w));
        if ((rc = b->gt.build(N, N, bins, win, dense, cfg->n_mels, bins))) return bail(rc);
        if (b->gt.lds_bytes > kLdsLimit) return bail(fail(MELSPEC_ERR_UNSUPPORTED, "geometry needs more LDS than one workgroup has"));
        if ((rc = generic_allow_lds())) return bail(rc);
    }
    *out = b;
    return MELSPEC_OK;
}

void melspec_blm_destroy(melspec_blm *b) {
    if (!b) return;
    if (b->dev.device >= 0) (void)hipSetDevice(b->dev.device);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    b->d_blob.release(); b->f32.d_blob.release(); b->h2d.release(); b->d2h.release(); b->gt.release(); b->ragged.release(); b->aux.release(); b->pipe.release();
    delete b;
}

// F32: the reference's own arithmetic type for this frontend (f32 window, FFT, power and projection, src/mel.rs:251-252,356-357) on the
// f32 instantiation of the fused kernel -- as far from the f64 evaluation of the definition as upstream's own f32 code is (2.4e-4 on
// jfk_f32le.wav, 5e-4 on a chirp; tools/f32_512_probe.py).  AUTO / F64 (the default): f64 from the window to |X|^2, within 1e-4 of that
// evaluation on every input.  Contexts without an f32 kernel (other geometries, other banks) compute in f64 whatever the mode.
int melspec_blm_set_precision(melspec_blm *b, int mode) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "blm is NULL");
    if (mode != MELSPEC_PRECISION_AUTO && mode != MELSPEC_PRECISION_F64 && mode != MELSPEC_PRECISION_F32)
        return fail(MELSPEC_ERR_INVALID_ARG, "precision must be MELSPEC_PRECISION_AUTO, _F64 or _F32");
    b->precision = mode;
    return MELSPEC_OK;
}
int melspec_blm_precision(const melspec_blm *b) {       // the arithmetic the next call will use: MELSPEC_PRECISION_F32 or _F64
    return b && b->precision == MELSPEC_PRECISION_F32 && b->fast && b->f32.ok ? MELSPEC_PRECISION_F32 : MELSPEC_PRECISION_F64;
}

size_t melspec_blm_num_frames(const melspec_blm *b, size_t n) { return b ? static_cast<size_t>(blm_valid_frames(b, n)) : 0; }
size_t melspec_blm_padded_frames(const melspec_blm *b, size_t n) { return b ? static_cast<size_t>(blm_padded(b, blm_valid_frames(b, n))) : 0; }

int melspec_blm_compute_uniform_device(melspec_blm *b, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                       uint32_t n_clips, float *d_out, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "blm is NULL");
    if (n_clips == 0) return MELSPEC_OK;
    const uint64_t valid = blm_valid_frames(b, clip_len), cols = blm_padded(b, valid);
    if (cols == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    const int nm = b->cfg.n_mels;
    int rc;
    if (!b->fast) {
        const BatchPlan pl = plan_uniform(d_pcm, d_out, clip_stride, valid, n_clips, nm, 1, cols, true);
        rc = launch_generic(b->gt, pl.desc, b->cfg.hop_length, 2, 1, 1, static_cast<double>(b->cfg.preemphasis), static_cast<double>(b->cfg.log_zero_guard),
                            b->dev.cus, s, static_cast<long long>(clip_len), b->cfg.center ? b->cfg.n_fft / 2 : 0);
        if (rc) return rc;
    } else {
    const BatchPlan pl = plan_uniform(d_pcm, d_out, clip_stride, valid, n_clips, nm, kFbFPW, cols, true);
    FbankFastParams fp{};
    fp.b = pl.desc;
    // feature-major store: waves holding adjacent units are kept in step (RoundSync); measured best for this kernel, see DESIGN 4.2b
    if (fp.b.sync_rounds < 0) fp.b.sync_rounds = kNemoSync;
    fp.d_blob = static_cast<const uint32_t *>(b->d_blob.p);
    fp.blob_words = static_cast<int>(b->ft.blob.size());
    fp.mel_off_words = b->ft.mel_off_words;
    fp.shift = b->cfg.hop_length;
    fp.n_mels = nm;
    fp.preemph = b->cfg.preemphasis;
    fp.floor_v = b->cfg.log_zero_guard;
    fp.use_log = 1; fp.use_power = 1;
    fp.clip_len = static_cast<long long>(clip_len);
    fp.org0 = b->cfg.center ? -200 : 56;      // tap 0 of the window sits at position (512-400)/2 of the frame
    fp.slots = b->ft.slots;
    if (b->precision == MELSPEC_PRECISION_F32 && b->f32.ok) rc = launch_nemo_f32(b->f32, fp, b->dev.cus, s);
    else if (fb_lens_match<LensSlaney128>(b->ft.slots)) rc = launch_fused512<double, kFlavorNemo, kBlmSlots, LensSlaney128>(b->waves, fp, b->fast_lds, b->dev.cus, s);
    else if (fb_lens_match<LensSlaney80>(b->ft.slots)) rc = launch_fused512<double, kFlavorNemo, kFbSlots, LensSlaney80>(b->waves, fp, b->fast_lds, b->dev.cus, s);
    else rc = b->ft.slots.n_slots <= kFbSlots ? launch_fused512<double, kFlavorNemo, kFbSlots>(b->waves, fp, b->fast_lds, b->dev.cus, s)
                                              : launch_fused512<double, kFlavorNemo, kBlmSlots>(b->waves, fp, b->fast_lds, b->dev.cus, s);
    if (rc) return rc;
    }
    if (b->cfg.normalize_per_feature && valid > 0) {
        BlmNormParams np{};
        np.out = d_out; np.clip_stride = cols * static_cast<uint64_t>(nm); np.row_w = cols; np.valid = valid;
        np.n_clips = n_clips; np.n_mels = nm;
        static const int fold_sel = lab_int("MELSPEC_NORM_FOLD", -1, -1, 12);
        np.fold_sel = fold_sel;
        static const int norm_skip = lab_int("MELSPEC_NORM_SKIP", 0, 0, 7);
        np.lab_skip = norm_skip;
        const uint64_t rows = static_cast<uint64_t>(n_clips) * nm;
        // four workgroups of <= 38 KB per CU measured best (1024 x 10 s x 128 mels, ms per call incl. the 0.72 ms mel kernel: 150 KB x 1: 1.72,
        // 76 x 2: 1.45, 50 x 3: 1.34, 38 x 4: 1.28, 25 x 6: 1.68); MELSPEC_NORM_KB / MELSPEC_NORM_PER_CU override
        size_t stride = (static_cast<size_t>(valid) + 3 + 31) & ~static_cast<size_t>(31);  // whole groups of 32 floats (a row starts up to 3 floats into its first granule) ...
        if ((stride / 4) % 2 == 0) stride += 4;                                              // ... and 4 * odd
        static const int norm_kb = lab_int("MELSPEC_NORM_KB", 38, 8, 158);
        static const int norm_per_cu = lab_int("MELSPEC_NORM_PER_CU", 4, 1, 16);
        const size_t budget = static_cast<size_t>(norm_kb) * 1024 - (64 * 2 + kBlmNormThreads) * sizeof(float);
        size_t per = budget / (stride * sizeof(float));
        int per_cu = norm_per_cu;
        if (per < 4) {                            // long rows (> ~25 s): one workgroup per CU with the whole LDS, up to ~6 min per row
            per = (static_cast<size_t>(150) * 1024) / (stride * sizeof(float));
            per_cu = 1;
        }
        if (per > 64) per = 64;
        np.rows_per_group = static_cast<int>(per);
        np.lds_stride = static_cast<int>(stride);
        static std::atomic<uint64_t> attr_done{0};
        if (!device_done(attr_done)) {
            int rc2 = allow_big_lds(&blm_normalize_kernel, "hipFuncSetAttribute(blm_normalize_kernel)");
            if (rc2) return rc2;
            mark_device_done(attr_done);
        }
        if (per == 0) {
            const unsigned g2 = grid_for((rows + kBlmNormThreads - 1) / kBlmNormThreads, b->dev.cus, 4);
            hipLaunchKernelGGL(blm_normalize_kernel, dim3(g2), dim3(kBlmNormThreads), 0, s, np);
        } else {
            const size_t lds = (per * stride + 2 * per + kBlmNormThreads) * sizeof(float);
            const unsigned g2 = grid_for((rows + per - 1) / per, b->dev.cus, per_cu);
#ifdef MELSPEC_LAB
            static const int norm_dbg = lab_int("MELSPEC_NORM_DBG", 0, 0, 1);
            static uint64_t *d_dbg = nullptr;
            static int dbg_calls = 0;
            if (norm_dbg) {
                if (!d_dbg) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_dbg), 64 * 8 * 8));
                HIP_TRY(hipMemsetAsync(d_dbg, 0, 64 * 8 * 8, s));
                np.dbg = d_dbg;
            }
#endif
            hipLaunchKernelGGL(blm_normalize_kernel, dim3(g2), dim3(kBlmNormThreads), lds, s, np);
#ifdef MELSPEC_LAB
            if (norm_dbg && ++dbg_calls == 20) {
                uint64_t h[64 * 8];
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost));
                double sum[8] = {0};
                for (int b = 0; b < 64; ++b) for (int k = 0; k < 8; ++k) sum[k] += static_cast<double>(h[b * 8 + k]);
                std::fprintf(stderr, "norm phases, us per workgroup (mean of 64): load %.1f  mean %.1f  var %.1f  var-sum %.1f  store %.1f\n",
                             sum[1] / 64 / 100, sum[2] / 64 / 100, sum[3] / 64 / 100, sum[4] / 64 / 100, sum[5] / 64 / 100);
            }
#endif
        }
        HIP_TRY(hipGetLastError());
    }
    return MELSPEC_OK;
}

// BatchLogMelSpectrogram::compute per clip of any length (src/mel.rs:299-385) in one launch: clip c = d_pcm[h_offsets[c] .. + h_lengths[c])
// -> [n_mels][cols_c] floats at d_out + h_out_offsets[c] (NULL: packed in clip order), cols_c = melspec_blm_padded_frames(len_c).
// Fused kernel only (n_fft 512 / win_length 400).
int melspec_blm_compute_ragged_device(melspec_blm *b, const float *d_pcm, const uint64_t *h_offsets, const uint64_t *h_lengths,
                                      uint32_t n_clips, float *d_out, const uint64_t *h_out_offsets, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "blm is NULL");
    if (n_clips == 0) return MELSPEC_OK;
    if (!h_offsets || !h_lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    if (!b->fast) return fail(MELSPEC_ERR_UNSUPPORTED, "ragged batches need the fused kernel (n_fft = 512, win_length = 400)");
    std::vector<uint64_t> cols(n_clips), aux(2 * static_cast<size_t>(n_clips) + 1);        // lengths, valid frames, the normaliser's group counter (0)
    uint64_t total = 0, longest = 0;
    for (uint32_t i = 0; i < n_clips; ++i) {
        const uint64_t valid = blm_valid_frames(b, h_lengths[i]);
        cols[i] = blm_padded(b, valid);
        aux[i] = h_lengths[i];
        aux[n_clips + i] = valid;
        total += cols[i];
        longest = std::max(longest, valid);
    }
    if (total == 0) return MELSPEC_OK;
    if (!d_pcm || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    const int nm = b->cfg.n_mels;
    if (b->aux_used && b->aux_stream != s) HIP_TRY(hipStreamSynchronize(b->aux_stream));
    b->aux_used = true; b->aux_stream = s;
    int rc = b->aux.ensure(aux.size() * sizeof(uint64_t));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b->aux.p, aux.data(), aux.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));    // pageable source: staged before the call returns
    BatchPlan pl;
    RaggedSlot *slot = nullptr;
    rc = plan_ragged(b->ragged, s, d_pcm, d_out, h_offsets, cols, h_out_offsets, n_clips, nm, kFbFPW, pl, slot);
    if (!rc) {
        FbankFastParams fp{};
        fp.b = pl.desc;
        fp.b.mel_major = 1;
        fp.b.sync_rounds = kNemoSync;
        fp.d_blob = static_cast<const uint32_t *>(b->d_blob.p);
        fp.blob_words = static_cast<int>(b->ft.blob.size());
        fp.mel_off_words = b->ft.mel_off_words;
        fp.shift = b->cfg.hop_length;
        fp.n_mels = nm;
        fp.preemph = b->cfg.preemphasis;
        fp.floor_v = b->cfg.log_zero_guard;
        fp.use_log = 1; fp.use_power = 1;
        fp.org0 = b->cfg.center ? -200 : 56;
        fp.d_len = static_cast<const uint64_t *>(b->aux.p);
        fp.d_valid = fp.d_len + n_clips;
        fp.slots = b->ft.slots;
        if (b->precision == MELSPEC_PRECISION_F32 && b->f32.ok) rc = launch_nemo_f32(b->f32, fp, b->dev.cus, s);
        else if (fb_lens_match<LensSlaney128>(b->ft.slots)) rc = launch_fused512<double, kFlavorNemo, kBlmSlots, LensSlaney128>(b->waves, fp, b->fast_lds, b->dev.cus, s);
        else if (fb_lens_match<LensSlaney80>(b->ft.slots)) rc = launch_fused512<double, kFlavorNemo, kFbSlots, LensSlaney80>(b->waves, fp, b->fast_lds, b->dev.cus, s);
        else rc = b->ft.slots.n_slots <= kFbSlots ? launch_fused512<double, kFlavorNemo, kFbSlots>(b->waves, fp, b->fast_lds, b->dev.cus, s)
                                                  : launch_fused512<double, kFlavorNemo, kBlmSlots>(b->waves, fp, b->fast_lds, b->dev.cus, s);
        if (!rc && b->cfg.normalize_per_feature && longest > 0) {
            const uint64_t rows = static_cast<uint64_t>(n_clips) * nm;
            // rows staged whole in LDS like the uniform pass (sized for the longest clip), groups of rows from a counter
            size_t stride = (static_cast<size_t>(longest) + 3 + 31) & ~static_cast<size_t>(31);
            if ((stride / 4) % 2 == 0) stride += 4;
            const size_t fixed = (2 * 64 + kBlmNormThreads + 4 * 64 + 4) * sizeof(float);
            size_t per = (static_cast<size_t>(38) * 1024 - fixed) / (stride * sizeof(float));
            int per_cu = 4;
            if (per < 4) { per = (static_cast<size_t>(150) * 1024 - fixed) / (stride * sizeof(float)); per_cu = 1; }
            if (per > 64) per = 64;
            if (per >= 1 && longest < (1ull << 31)) {
                static std::atomic<uint64_t> attr_done{0};
                if (!device_done(attr_done)) {
                    rc = allow_big_lds(&blm_normalize_ragged_kernel, "hipFuncSetAttribute(blm_normalize_ragged_kernel)");
                    if (!rc) mark_device_done(attr_done);
                }
                if (!rc) {
                    BlmNormRaggedParams rp{};
                    rp.out = d_out; rp.d_out_off = pl.desc.d_out_off; rp.d_cols = pl.desc.d_frames; rp.d_valid = fp.d_valid;
                    rp.n_clips = n_clips; rp.n_mels = nm; rp.rows_per_group = static_cast<int>(per); rp.lds_stride = static_cast<int>(stride);
                    rp.longest = static_cast<uint32_t>(longest);
                    rp.ctr = reinterpret_cast<unsigned *>(static_cast<uint64_t *>(b->aux.p) + 2 * static_cast<size_t>(n_clips));
                    const size_t lds = (per * stride + 2 * per + kBlmNormThreads + 4 * per + 4) * sizeof(float);
                    const unsigned g2 = grid_for((rows + per - 1) / per, b->dev.cus, per_cu);
                    hipLaunchKernelGGL(blm_normalize_ragged_kernel, dim3(g2), dim3(kBlmNormThreads), lds, s, rp);
                    if (hipGetLastError() != hipSuccess) rc = fail(MELSPEC_ERR_INTERNAL, "blm_normalize_ragged_kernel launch failed");
                }
            } else {
                // rows too long for LDS: one thread per row from HBM
                BlmNormParams np{};
                np.fold_sel = -1;
                np.out = d_out; np.n_clips = n_clips; np.n_mels = nm; np.rows_per_group = 0;
                np.d_out_off = pl.desc.d_out_off; np.d_cols = pl.desc.d_frames; np.d_valid = fp.d_valid;
                hipLaunchKernelGGL(blm_normalize_kernel, dim3(grid_for((rows + kBlmNormThreads - 1) / kBlmNormThreads, b->dev.cus, 4)), dim3(kBlmNormThreads), 0, s, np);
                if (hipGetLastError() != hipSuccess) rc = fail(MELSPEC_ERR_INTERNAL, "blm_normalize_kernel launch failed");
            }
        }
    }
    plan_ragged_done(slot, s);
    return rc;
}

int melspec_blm_release_scratch(melspec_blm *b) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "blm is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->aux_used && b->aux_stream != b->stream) HIP_TRY(hipStreamSynchronize(b->aux_stream));
    b->pipe.release(); b->ragged.release(); b->aux.release(); b->h2d.release(); b->d2h.release();
    b->aux_used = false;
    return MELSPEC_OK;
}

int melspec_blm_synchronize(melspec_blm *b, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "blm is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    HIP_TRY(hipStreamSynchronize(stream ? static_cast<hipStream_t>(stream) : b->stream));
    return MELSPEC_OK;
}

int melspec_blm_compute_host(melspec_blm *b, const float *samples, size_t n_samples, float *out, size_t out_capacity_floats,
                             size_t *rows, size_t *cols) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "blm is NULL");
    if (rows) *rows = static_cast<size_t>(b->cfg.n_mels);
    if (cols) *cols = 0;
    const uint64_t c = blm_padded(b, blm_valid_frames(b, n_samples));
    if (c == 0) return MELSPEC_OK;
    if (!samples || !out) return fail(MELSPEC_ERR_INVALID_ARG, "samples/out is NULL");
    const uint64_t need = c * static_cast<uint64_t>(b->cfg.n_mels);
    if (out_capacity_floats < need) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(b->dev.device));
    int rc;
    if ((rc = b->h2d.ensure(n_samples * sizeof(float)))) return rc;
    if ((rc = b->d2h.ensure(need * sizeof(float)))) return rc;
    HIP_TRY(hipMemcpyAsync(b->h2d.p, samples, n_samples * sizeof(float), hipMemcpyHostToDevice, b->stream));
    rc = melspec_blm_compute_uniform_device(b, static_cast<const float *>(b->h2d.p), n_samples, n_samples, 1,
                                            static_cast<float *>(b->d2h.p), b->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, b->d2h.p, need * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (cols) *cols = static_cast<size_t>(c);
    return MELSPEC_OK;
}

// BatchLogMelSpectrogram::compute on many host clips in one call: clip i -> [n_mels][cols_i] floats at out + out_offsets[i] (NULL:
// packed), cols_i = melspec_blm_padded_frames(lengths[i]).  Whole clips in chunks through the host pipeline, one ragged launch per
// chunk (configurations on the generic kernel: one clip at a time).
int melspec_blm_compute_batch_host(melspec_blm *b, const float *samples, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_clips,
                                   float *out, const uint64_t *out_offsets, size_t out_capacity_floats, uint64_t *total_columns) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "blm is NULL");
    if (total_columns) *total_columns = 0;
    if (n_clips == 0) return MELSPEC_OK;
    if (!offsets || !lengths) return fail(MELSPEC_ERR_INVALID_ARG, "offset/length array is NULL");
    const uint64_t nm = static_cast<uint64_t>(b->cfg.n_mels);
    std::vector<HostSeg> segs;
    segs.reserve(n_clips);
    uint64_t total = 0, cursor = 0;
    for (uint32_t i = 0; i < n_clips; ++i) {
        const uint64_t c = blm_padded(b, blm_valid_frames(b, lengths[i]));
        const uint64_t oo = out_offsets ? out_offsets[i] : cursor;
        if (c && oo + c * nm > out_capacity_floats) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
        if (c && (!samples || !out)) return fail(MELSPEC_ERR_INVALID_ARG, "samples/out is NULL");
        if (c) segs.push_back(HostSeg{samples + offsets[i], lengths[i], out + oo, c});
        cursor += c * nm; total += c;
    }
    if (total_columns) *total_columns = total;
    if (total == 0) return MELSPEC_OK;
    HIP_TRY(hipSetDevice(b->dev.device));
    if (!b->fast) {
        for (const HostSeg &sg : segs) {
            const int rc = melspec_blm_compute_host(b, sg.src, static_cast<size_t>(sg.n), sg.dst, static_cast<size_t>(sg.frames * nm), nullptr, nullptr);
            if (rc) return rc;
        }
        return MELSPEC_OK;
    }
    const char *where = "";
    const int rc = b->pipe.run(segs, b->cfg.n_mels, kPipeChunkSamples, b->stream,
                               [b](const float *d_in, const uint64_t *offs, const uint64_t *lens, uint32_t n, float *d_out,
                                   const uint64_t *ooffs, hipStream_t s) {
                                   return melspec_blm_compute_ragged_device(b, d_in, offs, lens, n, d_out, ooffs, s);
                               }, &where);
    if (rc > 0 && where[0] && std::strcmp(where, "kernel launch") != 0) return fail_hip(static_cast<hipError_t>(rc), where);
    return rc;
}


}  // extern "C"
