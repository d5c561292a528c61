// host_common.hpp -- what the translation units of libmelspec_hip.so share on the host side: error reporting, device buffers,
// batch planning, the context objects the C ABI hands out, and the declarations of the launchers each family's unit defines.
//
//   host_api.hip    the C ABI of melspec_ctx (src/cuda.rs:27-148's CudaMelSpectrogram), the sharded object, table builders, memory helpers
//   whisper400.hip  launch_ctx: every n_fft = 400 kernel (f32 + guard + vote, f64), the STFT export
//   fbank512.hip    the fused 512-point family: Kaldi fbank (melspec_fbank_*), NeMo frontend (melspec_blm_*), Whisper at n_fft = 512
//   pow2.hip        generic_frame_kernel / pow2_frame_kernel / generic_stft_kernel: every other geometry, f64
//   aux.hip         batch planners, streaming bank, TGA quantiser, VAD columns, stand-alone mel helpers, synthetic PCM
//   melspec_runs.hip  the run-per-wave f32 Whisper kernels, compiled with their own scheduling strategy
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/melspec_hip.h"
#include "fast_tables.hpp"
#include "fbank_tables.hpp"
#include "kernels_common.hpp"
#include "whisper_wave.hpp"
#include "whisper_wave_f64.hpp"
#include "whisper_six.hpp"
#include "whisper_six64.hpp"
#include "whisper_fix64.hpp"
#include "fbank_wave.hpp"
#include "pow2_wave.hpp"
#include "host_pipe.hpp"
#include "stream_plan.hpp"
#include "tables.hpp"

namespace melspec {
namespace host {

extern thread_local std::string g_last_error;      // melspec_last_error(): defined in host_api.hip

inline int fail(int code, const char *what) {
    g_last_error = what;
    return code;
}
inline int fail_hip(hipError_t e, const char *where) {
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    const int c = static_cast<int>(e);
    return c > 0 ? c : MELSPEC_ERR_INTERNAL;
}
#define HIP_TRY(expr)                                         \
    do {                                                      \
        const hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);     \
    } while (0)

constexpr int kGenericNT = 256;
constexpr int kMaxGenericFft = 4096;
constexpr int kMaxGenericMels = 1024;
constexpr int kNemoSync = 0;                // RoundSync mode of the f64 NeMo kernel's feature-major store: none.  (Round 2: pairs of adjacent waves, profiles/r02_nemo.txt;
                                            //  re-measured in round 5 after the clip-edge frames lost their chain of round trips: none -2.0 .. -2.3 % at 80 / 128 mels, pairs four apart +3 %, fours +1 %)
constexpr size_t kLdsLimit = 160 * 1024;   // gfx950: one workgroup may use the whole 160 KiB LDS of a CU
constexpr uint64_t kPipeChunkSamples = 4u << 20;      // host pipeline: 16 MiB of PCM per chunk (host_pipe.hpp)

// Tuning switches exist only in -DMELSPEC_LAB builds (mel_spec_amd.build.build(lab=True), used by tools/): the product
// library runs the measured defaults below and reads no environment variable.
#ifdef MELSPEC_LAB
inline int lab_int(const char *name, int dflt, int lo, int hi) {
    const char *e = std::getenv(name);
    if (!e) return dflt;
    const int v = std::atoi(e);
    return v >= lo && v <= hi ? v : dflt;
}
#else
constexpr int lab_int(const char *, int dflt, int, int) { return dflt; }
#endif

template <typename K>
int allow_big_lds(K kernel, const char *name) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsLimit));
    if (e != hipSuccess) return fail_hip(e, name);
    return MELSPEC_OK;
}

inline uint64_t current_device_bit() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return 1ull << (dev & 63);
}
// the masks are shared by every context of the process (one context per thread and device is the threading model)
inline bool device_done(const std::atomic<uint64_t> &mask) { return (mask.load(std::memory_order_acquire) & current_device_bit()) != 0; }
inline void mark_device_done(std::atomic<uint64_t> &mask) { mask.fetch_or(current_device_bit(), std::memory_order_release); }

struct DeviceInfo {
    int device = -1;
    int cus = 0;
    size_t lds_per_block = 0;
};

int pick_device(int device, DeviceInfo &info);      // host_api.hip

// Grow-only device buffer.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return MELSPEC_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        HIP_TRY(hipMalloc(&p, bytes));
        cap = bytes;
        return MELSPEC_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <typename T>
inline int upload(DevBuf &buf, const std::vector<T> &v) {
    const size_t bytes = v.size() * sizeof(T);
    int rc = buf.ensure(bytes ? bytes : 16);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(buf.p, v.data(), bytes, hipMemcpyHostToDevice));
    return MELSPEC_OK;
}

// The banded filterbank as JOBS for the wave kernels' mel phase (pow2_frame_kernel, mel_stage_jobs_kernel); defined in aux.hip
void build_mel_jobs(const BandedFilterbank &fb, int n_mels, int lf, std::vector<double> &jwv, std::vector<int> &jobv);

struct GenericTables {
    DevBuf win, tw, mstart, mlen, moff, mw, jw, job;
    int n_fft = 0, frame_len = 0, n_bins = 0, n_mels = 0, fft_log2 = 0, mw_count = 0, n_jobs = 0;
    bool force_generic = false;     // the workgroup-per-frame kernel whatever the geometry (cross-checks)
    FftPlan plan{};
    size_t lds_bytes = 0;
    int build(int n_fft_, int frame_len_, int n_bins_, const std::vector<double> &window,
              const std::vector<double> &dense, int n_mels_, int dense_bins) {
        n_fft = n_fft_; frame_len = frame_len_; n_bins = n_bins_; n_mels = n_mels_;
        std::vector<double> twv(2 * static_cast<size_t>(n_fft));
        for (int j = 0; j < n_fft; ++j) {
            const double a = 2.0 * kPi * j / n_fft;
            twv[2 * j] = std::cos(a);
            twv[2 * j + 1] = -std::sin(a);
        }
        const BandedFilterbank fb = band_filterbank(dense, n_mels, dense_bins, n_bins);
        int rc;
        {
            std::vector<double> padded(window);          // n_fft entries, zero from frame_len on: pow2_frame_kernel reads them unconditionally
            if (static_cast<int>(padded.size()) < n_fft) padded.resize(n_fft, 0.0);
            if ((rc = upload(win, padded))) return rc;
        }
        if ((rc = upload(tw, twv))) return rc;
        if ((rc = upload(mstart, fb.start))) return rc;
        if ((rc = upload(mlen, fb.len))) return rc;
        if ((rc = upload(moff, fb.offset))) return rc;
        if ((rc = upload(mw, fb.w))) return rc;
        mw_count = static_cast<int>(fb.w.size());
        {
            std::vector<double> jwv;
            std::vector<int> jobv;
            const int half = n_fft / 2;
            build_mel_jobs(fb, n_mels, half >= 512 ? 64 : half / 8, jwv, jobv);
            n_jobs = jobv.size() == 1 && (jobv[0] >> 20) == 0 ? 0 : static_cast<int>(jobv.size());
            if ((rc = upload(jw, jwv))) return rc;
            if ((rc = upload(job, jobv))) return rc;
        }
        // power-of-two transforms run as an in-LDS FFT over n_fft/2 complex points (the frame slot then holds n_fft doubles)
        fft_log2 = 0;
        if (n_fft >= 8 && (n_fft & (n_fft - 1)) == 0 && frame_len <= n_fft) {
            while ((1 << fft_log2) < n_fft) ++fft_log2;
        }
        // other 2-3-5-smooth sizes (320, 480, 800, 1200 ...; 400 when a geometry is off the fused kernels): mixed-radix passes over
        // n_fft complex points, two LDS buffers; anything else (a prime factor > 5, or no room) keeps the direct DFT
        plan = FftPlan{};
        if (!fft_log2 && n_fft >= 6 && frame_len <= n_fft) {
            int rest = n_fft, nr = 0, rad[14];
            for (int f : {4, 2, 3, 5})
                while (rest % f == 0 && nr < 14) { rad[nr++] = f; rest /= f; }
            const size_t need = sizeof(double) * (2 * static_cast<size_t>(n_fft) + 4 * static_cast<size_t>(n_fft) + n_bins + n_mels + kGenericNT);
            if (rest == 1 && need <= kLdsLimit) {
                plan.n_rad = nr;
                for (int i = 0; i < nr; ++i) plan.packed |= static_cast<unsigned long long>(rad[i]) << (4 * i);
            }
        }
        lds_bytes = sizeof(double) * (2 * static_cast<size_t>(n_fft) + (plan.n_rad ? 4 * static_cast<size_t>(n_fft) : static_cast<size_t>(fft_log2 ? n_fft : frame_len)) +
                                      n_bins + n_mels + kGenericNT);
        return MELSPEC_OK;
    }
    void release() { win.release(); tw.release(); mstart.release(); mlen.release(); moff.release(); mw.release(); jw.release(); job.release(); }
};

// Device-side copies of a ragged batch description.
// A ragged plan travels host -> pinned slot -> device slot -> kernel.  Four slots per context, each with an event recorded
// behind the launch that reads it: a call neither waits for the stream (the copy is truly asynchronous from pinned
// memory) nor overwrites a plan an earlier launch -- possibly on another stream -- may still be reading.
struct RaggedSlot {
    DevBuf dev;
    void *host = nullptr;
    size_t host_cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    int ensure_host(size_t bytes) {
        if (bytes <= host_cap) return MELSPEC_OK;
        if (host) { (void)hipHostFree(host); host = nullptr; host_cap = 0; }
        HIP_TRY(hipHostMalloc(&host, bytes, hipHostMallocDefault));
        host_cap = bytes;
        return MELSPEC_OK;
    }
    void release() {
        if (ev) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); ev = nullptr; }
        if (host) { (void)hipHostFree(host); host = nullptr; host_cap = 0; }
        dev.release();
        pending = false;
    }
};
struct RaggedScratch {
    static constexpr unsigned kSlots = 4;
    RaggedSlot slot[kSlots];
    unsigned next = 0;
    void release() { for (auto &s : slot) s.release(); }
};

struct BatchPlan {
    BatchDesc desc{};
    uint64_t total_frames = 0;
};

// Fill a BatchDesc for n_clips equal-length clips.
inline BatchPlan plan_uniform(const float *d_pcm, float *d_out, uint64_t clip_stride, uint64_t frames_per_clip,
                       uint32_t n_clips, int n_mels, int frames_per_unit, uint64_t out_width = 0, bool mel_major = false) {
    BatchPlan pl;
    BatchDesc &b = pl.desc;
    if (out_width < frames_per_clip) out_width = frames_per_clip;
    b.pcm = d_pcm; b.out = d_out;
    b.clip_stride = clip_stride;
    b.out_stride = out_width * static_cast<uint64_t>(n_mels);
    b.frames_per_clip = frames_per_clip;
    b.out_width = out_width;
    b.mel_major = mel_major ? 1 : 0;
    // mel-major stores keep waves that hold adjacent units in step, so that the 24-byte pieces of a 32-byte sector reach L2
    // together (RoundSync in kernels_common.hpp): -1 = the measured best of the kernel that runs, resolved in launch_ctx.
    // Lab builds: MELSPEC_MM_SYNC 0 none, 1 one workgroup barrier per round, 2/4/8 sub-group barrier over consecutive waves,
    // 16 + 2/4/8 over waves WAVES / size apart; MELSPEC_FM_SYNC=1: workgroup barrier for the padded frame-major layout too.
    static const int mm_mode = [] { const int v = lab_int("MELSPEC_MM_SYNC", -1, -1, 31); const int sz = v & 15; return (v <= 1 || ((sz == 2 || sz == 3 || sz == 4 || sz == 6 || sz == 8) && (v >> 4) <= 1)) ? v : 1; }();   // 3 / 6: the twelve-wave kernels only
    static const bool fm_on = lab_int("MELSPEC_FM_SYNC", 0, 0, 1) != 0;
    b.sync_rounds = mel_major ? mm_mode : (fm_on ? 1 : 0);
    b.frames_per_unit = frames_per_unit;
    b.units_per_clip = static_cast<uint32_t>((out_width + frames_per_unit - 1) / frames_per_unit);
    b.n_clips = n_clips;
    b.n_units = static_cast<uint64_t>(b.units_per_clip) * n_clips;
    pl.total_frames = frames_per_clip * n_clips;
    return pl;
}
// Fills the next slot and queues its upload on `stream`.  The caller launches on `stream` and then calls plan_ragged_done.  (aux.hip)
int plan_ragged(RaggedScratch &rs, hipStream_t stream, const float *d_pcm, float *d_out, const uint64_t *h_off,
                const std::vector<uint64_t> &frames, const uint64_t *h_out_off, uint32_t n_clips, int n_mels,
                int frames_per_unit, BatchPlan &pl, RaggedSlot *&used, bool want_order = false);
void plan_ragged_done(RaggedSlot *sl, hipStream_t stream);

// Ragged plan built on the device from descriptors that live there (plan_ragged_device_kernel).  One buffer per object, used in
// stream order (a call on another stream first waits for the stream that used it last).
struct DevicePlan {
    DevBuf buf;
    hipStream_t last = nullptr;
    bool used = false;
    void release() { buf.release(); used = false; last = nullptr; }
};

int plan_ragged_device(DevicePlan &dp, hipStream_t stream, const float *d_pcm, float *d_out, const uint64_t *d_off, const uint64_t *d_len,
                       const uint64_t *d_out_off, uint32_t n_clips, uint64_t frame_len, uint64_t frame_shift, uint32_t words_per_frame,
                       int frames_per_unit, uint64_t max_total_frames, BatchPlan &pl);      // aux.hip

inline unsigned grid_for(uint64_t units, int cus, int per_cu) {
    const uint64_t cap = static_cast<uint64_t>(cus > 0 ? cus : 256) * per_cu;
    const uint64_t g = units < cap ? units : cap;
    return static_cast<unsigned>(g ? g : 1);
}
// same, rounded up to a multiple of the 8 XCDs for the kernels that reorder their workgroups (xcd_logical_block);
// lab builds: MELSPEC_XCD=0 keeps the dispatcher's order (odd grid sizes switch the reordering off in the kernel)
inline unsigned grid_for_xcd(uint64_t units, int cus, int per_cu) {
    static const bool off = lab_int("MELSPEC_XCD", 1, 0, 1) == 0;
    const unsigned g = grid_for(units, cus, per_cu);
    if (off) return (g % 8 == 0 && g > 1) ? g - 1 : g;
    return (g + 7u) & ~7u;
}

// ---- pow2.hip: every geometry off the fused kernels -------------------------------------------------------------------------------
// log2 of the complex transform pow2_frame_kernel would run this geometry with (6..10), or 0: generic_frame_kernel
int pow2_logm(const GenericTables &gt);
int launch_generic(const GenericTables &gt, const BatchDesc &desc, int hop, int flavour /* 0 Whisper, 1 Kaldi fbank, 2 NeMo */, int use_log, int use_power,
                   double preemph, double floor_v, int cus, hipStream_t stream, long long clip_len = 0, int pad = 0);
int generic_allow_lds();          // hipFuncSetAttribute(generic_frame_kernel): once per context that runs on it

// ---- fbank512.hip: the fused 512-point family -------------------------------------------------------------------------------------
// Waves per workgroup of the fused 512-point kernels: 8 (two per SIMD, one workgroup per CU) when the tables
// and eight 18.5 KB slices fit in LDS, else 4.
inline int fused512_waves(size_t blob_bytes, size_t slice_bytes) {
    if (lab_int("MELSPEC_FB_WAVES", 8, 4, 8) == 4) return 4;
    return blob_bytes + 8 * slice_bytes <= kLdsLimit ? 8 : 4;
}
// does the context's bank have exactly the compile-time slot lengths of Lens?  (lab builds: MELSPEC_RUNTIME_LENS=1 forces the run-time loop)
template <class Lens>
bool fb_lens_match(const MelSlots &ms) {
    static const bool off = lab_int("MELSPEC_RUNTIME_LENS", 0, 0, 1) != 0;
    if (off || ms.n_slots != Lens::kSlots) return false;
    for (int i = 0; i < Lens::kSlots; ++i)
        if (ms.len[i] != Lens::len(i) || ms.woff[i] != Lens::woff(i)) return false;
    return true;
}

// The f32 side of a fused 512-point context (MELSPEC_PRECISION_F32; NeMo / Whisper-512 with one of the compile-time banks).
constexpr int kFused512F32Waves = 12;
struct Fused512F32 {
    bool ok = false;
    FbankFastTables ft;
    DevBuf d_blob;
    size_t lds = 0;
    // extra: bytes per wave behind the slice (the Whisper flavour's frame maxima live inside the slice; slack as on the f64 side)
    // tail: bytes behind the slices and the sixteen counter words (the NeMo flavour's staged rows)
    int finish(size_t extra, size_t tail = 0) {
        lds = ft.blob.size() * 4 + static_cast<size_t>(kFused512F32Waves) * (FbankLayout<float>::slice_elems() * sizeof(float) + extra) + 64 + tail;
        ok = lds <= kLdsLimit;
        return ok ? upload(d_blob, ft.blob) : MELSPEC_OK;
    }
};
inline bool w512_f32_bank(const MelSlots &ms) { return fb_lens_match<LensSlaney80W>(ms) || fb_lens_match<LensSlaney128>(ms); }
inline bool nemo_f32_bank(const MelSlots &ms) { return fb_lens_match<LensSlaney128>(ms) || fb_lens_match<LensSlaney80>(ms); }

}  // namespace host
}  // namespace melspec

using namespace melspec;
using namespace melspec::host;

// ------------------------------------------------------------------------------------
// Whisper log-mel context
// ------------------------------------------------------------------------------------
// MELSPEC_PRECISION_AUTO state (FixSink in kernels_common.hpp): the f64 tables of the in-kernel recompute and its counter.
struct FixState {
    DevBuf tab, count, list;              // count: {u64 frames that tripped the guard, u64 accumulator of the launch in flight, u64 tally of the vote}
    DevBuf verdicts;                      // FixSink::decision: kVoteSlots copies of the last vote's verdict
    hipStream_t last_stream = nullptr;    // the note list is used in stream order: a call on another stream first waits for this one
    bool used = false;
    // Statistics of the guarded launches, published by the kernels into host-mapped memory (FixSink::host) and read here without
    // touching the stream (melspec_auto_state, melspec_guard_count's cheap sibling).  They no longer decide anything: since round 4
    // the kernel a batch runs on is decided by a vote inside the batch's own launch (FixSink::vote in kernels_common.hpp), so the
    // result of a call is a function of its input alone -- round 3 chose from the statistics of the last FINISHED batch, which made
    // the bits of a batch depend on what the context had seen before and on how far the host was ahead of the GPU.
    unsigned long long *host = nullptr;   // {seq << 40 | tripped, seq << 40 | frames (bit 39: published by the gated f64 launch)} of the last finished launch
    uint32_t seq = 0, seen_seq = 0;
    bool adaptive = true;                 // the vote is on (melspec_set_auto_adaptive); off: the f32 kernel + recompute tail whatever the input
    bool heavy = false;                   // the last finished AUTO batch ran on the f64 kernel (reporting only)
    double fraction = 0.0;                // of the last finished launch of >= kAutoMinFrames frames
    void release() {
        tab.release(); count.release(); list.release(); verdicts.release(); used = false; last_stream = nullptr;
        if (host) (void)hipHostFree(host);
        host = nullptr;
    }
};
constexpr unsigned long long kAutoMinFrames = 256;
constexpr unsigned long long kStatFromGated = 1ull << 39;

struct melspec_ctx {
    DeviceInfo dev;
    int fft_size = 0, hop_size = 0, n_mels = 0;
    double sr = 0.0;
    std::vector<double> dense;      // the filterbank, [n_mels][fft_size / 2 + 1]: MelSpectrogram::new's mel(sr, fft, n_mels, None, None, false, true)
                                    // (src/mel.rs:19-24) or the caller's (melspec_create_with_filterbank / _with_dense_filterbank)
    hipStream_t stream = nullptr;
    // fused n_fft = 400 build, five frames per wave (whisper400_wave_*): every bank of <= 131 mels; serves 81..131 mels and
    // carries the tables the f64 kernels share
    bool fast = false;
    FastTables ft;
    DevBuf d_blob;
    size_t fast_lds = 0;
    int lens_kind = 0;      // 0 runtime slot lengths, 1 static Whisper-80, 2 static Whisper-128
    // six-frames-per-wave build (whisper400_six_*): <= 80 mels, every batch shape and layout while the context computes in f32
    bool six = false;
    int six_static = 0;     // the compile-time bank that matches the tables: 1 LensSix80, 2 LensSix64, 3 LensSix40 (0: run-time slot lengths)
    FastTables ft6;
    DevBuf d_blob6;
    size_t lds6 = 0;
    // the f64 kernel on the six-frame skeleton (whisper400_six64_kernel): plain batches of the six-frame contexts in MELSPEC_PRECISION_F64,
    // and AUTO's gated second launch
    bool six64 = false;
    bool six64_wide = false;    // ... with fifteen mel slots (Whisper large-v3's 128-mel bank): plain batches only, c->six is false there
    // the f32 six-frame kernel with fifteen mel slots on twelve waves (whisper400_six_wide_runs_kernel, round 6): plain batches of that bank
    bool six_wide32 = false;
    bool six_wide32_layouts = false;   // ... its padded / mel-major layouts too (whisper400_six_wide_kernel)
    FastTables ft6w;
    DevBuf d_blob6w;
    size_t lds6w = 0;
    Six64Tables t64;
    DevBuf d_blob64x;
    size_t lds64x = 0;
    // fused n_fft = 512 build (f64, Whisper flavour of the 512-point kernel): plain and ragged batches
    bool fast512 = false;
    FbankFastTables ft512;
    DevBuf d_blob512;
    size_t lds512 = 0;
    int waves512 = 4;
    Fused512F32 f512;           // MELSPEC_PRECISION_F32 at n_fft = 512 (the 80- and 128-mel banks)
    // f64 FFT build of the n_fft = 400 kernel: the whole batch (MELSPEC_PRECISION_F64) or the queued frames (AUTO)
    int precision = MELSPEC_PRECISION_AUTO;
    PreciseTables pt;
    DevBuf d_blob64, d_blob64s;      // f64 tables: of the mel kernels (power split) / of the spectrum export
    size_t precise_lds = 0;
    FixState fix;
    std::vector<hipEvent_t> *first_kernel_events = nullptr;   // melspec_time_first_kernel: an event pair around the first launch of every call
    // generic path
    GenericTables gt;
    // the mel stage on its own (melspec_mel_from_stft_*): the banded filterbank in f64, built on first use
    DevBuf st_start, st_len, st_off, st_w, st_jw, st_job;
    int st_n_jobs = 0;
    bool stage_built = false;
    // scratch
    RaggedScratch ragged;
    DevicePlan dplan;
    HostPipe pipe;          // chunked H2D / kernels / D2H pipeline of the host entry points (host_pipe.hpp)
};

namespace melspec {
namespace host {

// ---- whisper400.hip ------------------------------------------------------------------------------------------------------------------
void auto_poll(melspec_ctx *c);
// the launch-specific part of a guarded launch's statistics sink (frames, grid, the launch's number)
FixSink sink_armed(melspec_ctx *c, FixSink sink, const BatchDesc &desc, unsigned grid);
// MELSPEC_PRECISION_AUTO: the context's sink for a launch on `stream` (note list sized for the batch, statistics words, host-mapped
// figures; with_vote: the vote's tally and verdict words too)
int auto_sink(melspec_ctx *c, const BatchDesc &desc, hipStream_t stream, bool with_vote, FixSink &sink);
int launch_ctx(melspec_ctx *c, const BatchDesc &desc, hipStream_t stream);
int launch_stft(melspec_ctx *c, const BatchDesc &desc, int bins, int dtype, hipStream_t s);
// ---- pow2.hip / fbank512.hip: the parts of launch_ctx / launch_stft that run on their kernels ---------------------------------------
int launch_generic_stft(melspec_ctx *c, const BatchDesc &desc, int bins, int dtype, hipStream_t s);
int launch_whisper512(melspec_ctx *c, const BatchDesc &desc, hipStream_t stream);
bool w512_auto_ok(const melspec_ctx *c);
bool twelve_waves_for(const melspec_ctx *c, bool layout);      // whisper400.hip: the six-frame family's batches that run on twelve waves per CU

// frames per work unit of the kernel a batch is planned for (called once per batch, before it is planned).  AUTO plans for the f32
// kernel: when the batch's vote says "heavy", the f64 kernel walks the same plan (whisper400_precise_kernel, MODE 2).
// layout: a padded / mel-major batch (the f64 kernel of the layouts is the five-frame one)
// The six-frame f64 kernel serves a padded / mel-major batch only with one of the compile-time banks: its run-time-lens layout instantiation
// keeps 141 SGPRs' worth of slot tables and reloads 13 spilled registers inside the unit loop (tools/hotloop_spills.py); those banks stay
// on whisper400_precise_kernel's layout form.
inline bool six64_layout_ok(const melspec_ctx *c) { return c->six64 && !c->six64_wide && c->six_static != 0; }

inline int ctx_frames_per_unit(melspec_ctx *c, bool layout = false) {
    if (c->fast) {
        if (c->precision == MELSPEC_PRECISION_F64) return (layout ? six64_layout_ok(c) : c->six64) ? kSixFrames : kFPW;
        if (c->six_wide32 && (!layout || c->six_wide32_layouts)) return kSixFrames;          // the 128-mel bank: six frames per wave on twelve waves
        return c->six ? kSixFrames : kFPW;
    }
    return c->fast512 ? kFbFPW : 1;
}

inline int ctx_num_frames(const melspec_ctx *c, uint64_t n, uint64_t &frames) {
    frames = n < static_cast<uint64_t>(c->fft_size) ? 0 : (n - c->fft_size) / c->hop_size + 1;
    return MELSPEC_OK;
}
template <class Lens>
bool lens_match(const MelSlots &ms, int n_mels) {
    if (ms.n_slots != Lens::kSlots || n_mels != Lens::kMels) return false;
    for (int i = 0; i < Lens::kSlots; ++i)
        if (ms.len[i] != Lens::len(i) || ms.woff[i] != Lens::woff(i)) return false;
    return true;
}

inline int stft_args(const melspec_ctx *c, int dtype) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    return MELSPEC_OK;
}

// interleave_frames' width rule (src/mel.rs:497-516)
inline uint64_t interleaved_width(uint64_t frames, uint64_t min_width) {
    uint64_t nf = frames;
    if (min_width > 0 && (nf & 1)) nf += 1;
    return nf > min_width ? nf : min_width;
}

}  // namespace host
}  // namespace melspec
