// pow2.hip -- launchers of the f64 kernels every geometry off the fused ones runs on (generic_kernels.hpp): pow2_frame_kernel for the
// power-of-two frame sizes 128..2048, generic_frame_kernel / generic_stft_kernel for the rest.
#include "host_common.hpp"
#include "generic_kernels.hpp"

namespace melspec {
namespace host {

// Waves per workgroup pow2_frame_kernel<LOGM, .> gets for a bank -- one persistent workgroup per CU with as many waves as its LDS holds
// (the tables are paid once), at most two per SIMD (VGPRs); 0: the bank is past the kernel (more mels than its lanes read out, more bins
// or mels than a job record holds, tables that leave no room for a frame) and the geometry stays on generic_frame_kernel.
template <int LOGM>
int pow2_waves(int n_jobs, int n_mels, int n_bins) {
    using S = Pow2Shape<LOGM>;
    if (n_mels > S::kMelsPerLane * S::LF) return 0;
    if (n_jobs < 1 || n_bins > 4088 || n_mels > 256) return 0;
    int waves = S::kMaxWaves;
    while (waves > 1 && sizeof(double) * static_cast<size_t>(pow2_lds<LOGM>(n_jobs, n_mels, waves).total) > kLdsLimit) --waves;
    return sizeof(double) * static_cast<size_t>(pow2_lds<LOGM>(n_jobs, n_mels, waves).total) > kLdsLimit ? 0 : waves;
}

template <int LOGM, int FLAVOR>
int launch_pow2(const GenericParams &gp, int cus, hipStream_t stream) {
    using S = Pow2Shape<LOGM>;
    const int waves = pow2_waves<LOGM>(gp.n_jobs, gp.n_mels, gp.n_bins);
    if (waves == 0) return -1;
    const size_t lds = sizeof(double) * static_cast<size_t>(pow2_lds<LOGM>(gp.n_jobs, gp.n_mels, waves).total);
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&pow2_frame_kernel<LOGM, FLAVOR>, "hipFuncSetAttribute(pow2_frame_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t groups = (gp.b.n_units + static_cast<uint64_t>(waves) * S::FW - 1) / (static_cast<uint64_t>(waves) * S::FW);
    const unsigned grid = grid_for(groups, cus, 1);
    hipLaunchKernelGGL((pow2_frame_kernel<LOGM, FLAVOR>), dim3(grid), dim3(waves * 64), lds, stream, gp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

// log2 of the complex transform pow2_frame_kernel would run this geometry with (6..10), or 0: generic_frame_kernel
int pow2_logm(const GenericTables &gt) {
    static const bool pow2_on = lab_int("MELSPEC_POW2", 1, 0, 1) != 0;
    if (!pow2_on || gt.force_generic) return 0;
    int waves = 0;
    switch (gt.fft_log2) {
        case 7: waves = pow2_waves<6>(gt.n_jobs, gt.n_mels, gt.n_bins); break;
        case 8: waves = pow2_waves<7>(gt.n_jobs, gt.n_mels, gt.n_bins); break;
        case 9: waves = pow2_waves<8>(gt.n_jobs, gt.n_mels, gt.n_bins); break;
        case 10: waves = pow2_waves<9>(gt.n_jobs, gt.n_mels, gt.n_bins); break;
        case 11: waves = pow2_waves<10>(gt.n_jobs, gt.n_mels, gt.n_bins); break;
        default: break;
    }
    return waves ? gt.fft_log2 - 1 : 0;
}

int launch_generic(const GenericTables &gt, const BatchDesc &desc, int hop, int flavour /* 0 Whisper, 1 Kaldi fbank, 2 NeMo */, int use_log, int use_power,
                   double preemph, double floor_v, int cus, hipStream_t stream, long long clip_len, int pad) {
    if (desc.n_units == 0) return MELSPEC_OK;
    GenericParams gp{};
    gp.b = desc;
    gp.n_fft = gt.n_fft; gp.frame_len = gt.frame_len; gp.hop = hop; gp.n_bins = gt.n_bins; gp.n_mels = gt.n_mels;
    gp.fbank = flavour; gp.use_log = use_log; gp.use_power = use_power; gp.preemph = preemph; gp.floor_v = floor_v;
    gp.clip_len = clip_len; gp.pad = pad;
    static const bool generic_fft = lab_int("MELSPEC_GENERIC_FFT", 1, 0, 1) != 0;
    gp.fft_log2 = generic_fft ? gt.fft_log2 : 0;
    if (generic_fft) gp.plan = gt.plan;
    gp.d_win = static_cast<const double *>(gt.win.p);
    gp.d_tw = static_cast<const double *>(gt.tw.p);
    gp.d_mstart = static_cast<const int *>(gt.mstart.p);
    gp.d_mlen = static_cast<const int *>(gt.mlen.p);
    gp.d_moff = static_cast<const int *>(gt.moff.p);
    gp.d_mw = static_cast<const double *>(gt.mw.p);
    gp.mw_count = gt.mw_count;
    gp.d_jw = static_cast<const double *>(gt.jw.p);
    gp.d_job = static_cast<const int *>(gt.job.p);
    gp.n_jobs = gt.n_jobs;
    // power-of-two frame sizes 128 .. 2048: frames owned by lane groups of a wave (pow2_frame_kernel); lab builds: MELSPEC_POW2=0 keeps
    // the workgroup-per-frame kernel, which is also the on-device cross-check of the tests (melspec_*_use_generic)
    static const bool pow2_on = lab_int("MELSPEC_POW2", 1, 0, 1) != 0;
    if (pow2_on && !gt.force_generic && gt.fft_log2 >= 7 && gt.fft_log2 <= 11) {
        int rc = -1;
        switch (gt.fft_log2 * 4 + flavour) {
#define MS_POW2_CASE(LOG2, LOGM) \
            case LOG2 * 4 + 0: rc = launch_pow2<LOGM, 0>(gp, cus, stream); break; \
            case LOG2 * 4 + 1: rc = launch_pow2<LOGM, 1>(gp, cus, stream); break; \
            case LOG2 * 4 + 2: rc = launch_pow2<LOGM, 2>(gp, cus, stream); break;
            MS_POW2_CASE(7, 6) MS_POW2_CASE(8, 7) MS_POW2_CASE(9, 8) MS_POW2_CASE(10, 9) MS_POW2_CASE(11, 10)
#undef MS_POW2_CASE
        }
        if (rc >= 0) return rc;          // -1: the bank is wider than the kernel's lanes cover
    }
    const unsigned grid = grid_for(desc.n_units, cus, 8);
    hipLaunchKernelGGL(generic_frame_kernel<kGenericNT>, dim3(grid), dim3(kGenericNT), gt.lds_bytes, stream, gp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int generic_allow_lds() { return allow_big_lds(&generic_frame_kernel<kGenericNT>, "hipFuncSetAttribute(generic_frame_kernel)"); }

int launch_generic_stft(melspec_ctx *c, const BatchDesc &desc, int bins, int dtype, hipStream_t s) {
    const int words = bins * 2 * (dtype == MELSPEC_STFT_F64 ? 2 : 1);
    GenericStftParams g{};
    g.b = desc;
    g.n_fft = c->fft_size; g.hop = c->hop_size; g.bins = bins; g.words_per_frame = words; g.f64 = dtype == MELSPEC_STFT_F64;
    g.d_win = static_cast<const double *>(c->gt.win.p);
    g.d_tw = static_cast<const double *>(c->gt.tw.p);
    g.fft_log2 = c->gt.fft_log2;
    g.plan = c->gt.plan;
    const size_t lds = sizeof(double) * (c->gt.plan.n_rad ? 6 : 3) * static_cast<size_t>(c->fft_size);
    if (lds > kLdsLimit) return fail(MELSPEC_ERR_UNSUPPORTED, "geometry needs more LDS than one workgroup has");
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&generic_stft_kernel<kGenericNT>, "hipFuncSetAttribute(generic_stft_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    hipLaunchKernelGGL(generic_stft_kernel<kGenericNT>, dim3(grid_for(desc.n_units, c->dev.cus, 8)), dim3(kGenericNT), lds, s, g);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

}  // namespace host
}  // namespace melspec
