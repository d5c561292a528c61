// melspec_hip.hip -- C ABI of libmelspec_hip.so (see include/melspec_hip.h) and the host
// objects behind it.  Host logic is C++ because the reference's host side is compiled code
// (Rust, src/cuda.rs); this image has no Rust toolchain, so the Rust shim that binds this
// ABI is shipped as source in INTEGRATION.md / mel_spec_amd/rust/.
//
// HipMelCtx mirrors CudaMelSpectrogram (src/cuda.rs:27-148): it owns the device tables, a
// stream, and grow-only device scratch; `compute_host` is compute_mel_spectrogram.
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
#include <vector>

#include "../../include/melspec_hip.h"
#include "fast_tables.hpp"
#include "fbank_tables.hpp"
#include "melspec_kernels.hpp"
namespace melspec {
// emitted by melspec_runs.hip (compiled with its own scheduling strategy; see there)
extern template __global__ void whisper400_six_runs_kernel<kSixMaxSlots, LensSix80>(const FastParams);
extern template __global__ void whisper400_six_runs_kernel<kSixMaxSlots, LensSix40>(const FastParams);
extern template __global__ void whisper400_wave_runs_kernel<8, LensI80>(const FastParams);
extern template __global__ void whisper400_wave_runs_kernel<12, LensI128>(const FastParams);
}  // namespace melspec
#include "mel_bank.hpp"
#include "host_pipe.hpp"
#include "stream_plan.hpp"
#include "tga_quant.hpp"
#include "vad_columns.hpp"
#include "tables.hpp"

using namespace melspec;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *what) {
    g_last_error = what;
    return code;
}
int fail_hip(hipError_t e, const char *where) {
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
int lab_int(const char *name, int dflt, int lo, int hi) {
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
int upload(DevBuf &buf, const std::vector<T> &v) {
    const size_t bytes = v.size() * sizeof(T);
    int rc = buf.ensure(bytes ? bytes : 16);
    if (rc) return rc;
    if (bytes) HIP_TRY(hipMemcpy(buf.p, v.data(), bytes, hipMemcpyHostToDevice));
    return MELSPEC_OK;
}

// Tables of the generic (f64 DFT) kernel.
// The banded filterbank as JOBS for the wave kernels' mel phase (pow2_frame_kernel, mel_stage_jobs_kernel): eight consecutive weights
// of one mel per job (the last job of a band padded with zeros), dealt over the rounds as described below; lf = lanes that share a
// frame (a round = lf jobs).  jobv[j] = first bin | mel << 12 | count << 20; jwv: weight pairs (2 q, 2 q + 1) of job j at [q][j].
void build_mel_jobs(const BandedFilterbank &fb, int n_mels, int lf, std::vector<double> &jwv, std::vector<int> &jobv) {
    // A job = eight consecutive bins FROM AN EVEN ONE (its eight powers are four aligned 16-byte LDS reads) with the weights of one mel
    // on them, zero where the band is not; record = first bin | mel << 12 | count << 20 (count > 0: a real job).
    struct Job { int bin, mel, lo, hi; };                    // weights of band entries [lo, hi) sit at bins bin + (entry - lo) + lead
    std::vector<Job> jobs;
    std::vector<int> lead;                                   // zero weights in front of a job's first entry
    for (int m = 0; m < n_mels && m < 256; ++m) {
        const int st = fb.start[m], len = fb.len[m];
        if (len <= 0) continue;
        for (int b = st & ~1; b < st + len; b += 8) {
            const int lo = std::max(b, st) - st, hi = std::min(b + 8, st + len) - st;
            jobs.push_back({b, m, lo, hi});
            lead.push_back(std::max(b, st) - b);
        }
    }
#ifndef MS_POW2_JOBORDER
#define MS_POW2_JOBORDER 1
#endif
    // The lanes of a round read their jobs' powers with ds_read_b128, which the LDS serves in groups of sixteen lanes -- {0-3, 12-15,
    // 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- over the sixteen 16-byte slots of a 256-byte row: the jobs that meet in such a
    // group want first bins whose halves differ mod 16 (mod 8 where a frame has eight lanes: the group then holds the same eight jobs
    // of two pairs of frames, whose rows pow2_pw_shift sets eight slots apart).  In band order they do not -- a band's jobs are 8 bins
    // apart, the low bands 2-3 -- and the reads were 3-4-way (SQ_LDS_BANK_CONFLICT: 25 % of the LDS cycles at n_fft 2048).  So: sets of
    // g jobs; a job with fewer than eight entries may start 2, 4 or 6 bins early (more zeros in front); every job goes to the set that
    // holds the fewest jobs of its residue (then of its mel: ds_add_f64 to one address serialises), fullest residue classes first;
    // the sets padded with empty jobs (count 0) and laid onto the lane groups.  The order is a function of the bank: the sums stay
    // deterministic; a band's pieces are added in another order than the reference's left fold (f64: ~1e-16 relative).
    const int g = lf >= 16 ? 16 : 8;
    const size_t nj = jobs.size(), sets = std::max<size_t>(1, (nj + g - 1) / g);
    std::vector<std::vector<size_t>> grp(sets);
    std::vector<int> shift(nj, 0);
    if (MS_POW2_JOBORDER && lf >= 8) {
        std::vector<std::vector<size_t>> cls(g);
        for (size_t j = 0; j < nj; ++j) cls[(jobs[j].bin >> 1) % g].push_back(j);
        std::vector<int> order(g);
        for (int r = 0; r < g; ++r) order[r] = r;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cls[x].size() > cls[y].size(); });
        std::vector<std::vector<int>> used(sets, std::vector<int>(g, 0));
        for (int r : order)
            for (size_t j : cls[r]) {
                size_t best = sets;
                int best_shift = 0;
                long best_key = 0;
                const int room = 8 - (lead[j] + (jobs[j].hi - jobs[j].lo));
                for (int sh = 0; sh <= room && sh <= jobs[j].bin; sh += 2) {
                    const int res = ((jobs[j].bin - sh) >> 1) % g;
                    for (size_t q = 0; q < sets; ++q) {
                        if (grp[q].size() >= static_cast<size_t>(g)) continue;
                        long same_mel = 0;
                        for (size_t o : grp[q]) same_mel += jobs[o].mel == jobs[j].mel;
                        const long key = (static_cast<long>(used[q][res]) << 40) + (same_mel << 24) + (static_cast<long>(sh) << 16) + static_cast<long>(grp[q].size());
                        if (best == sets || key < best_key) { best = q; best_key = key; best_shift = sh; }
                    }
                }
                grp[best].push_back(j);
                shift[j] = best_shift;
                used[best][((jobs[j].bin - best_shift) >> 1) % g] += 1;
            }
    } else {
        for (size_t j = 0; j < nj; ++j) grp[j / g].push_back(j);
    }
    // lane positions: lf >= 32: two sets per 32 lanes, on the two lane groups of ds_read_b128; else a set = the lanes of a frame
    static const int kLanesOfGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const size_t slots = lf >= 32 ? ((sets + 1) / 2) * 32 : sets * g;
    std::vector<long> at(slots, -1);
    for (size_t q = 0; q < sets; ++q) {
        std::sort(grp[q].begin(), grp[q].end());
        for (size_t i = 0; i < grp[q].size(); ++i)
            at[lf >= 32 ? (q / 2) * 32 + kLanesOfGroup[q & 1][i] : q * g + i] = static_cast<long>(grp[q][i]);
    }
    jobv.assign(nj ? slots : 1, 0);
    std::vector<double> w8((nj ? slots : 1) * 8, 0.0);
    for (size_t pos = 0; pos < slots && nj; ++pos) {
        if (at[pos] < 0) continue;
        const size_t j = static_cast<size_t>(at[pos]);
        const Job &jb = jobs[j];
        const int bin = jb.bin - shift[j], first = lead[j] + shift[j];
        jobv[pos] = bin | (jb.mel << 12) | ((jb.hi - jb.lo) << 20);
        for (int e = jb.lo; e < jb.hi; ++e) w8[8 * pos + first + (e - jb.lo)] = fb.w[static_cast<size_t>(fb.offset[jb.mel]) + e];
    }
    // weight pairs (2 q, 2 q + 1) of the job at position j at [q][j]: the lanes of a round read consecutive 16-byte slots
    const size_t np = jobv.size();
    jwv.assign(8 * np, 0.0);
    for (size_t j = 0; j < np; ++j)
        for (int q = 0; q < 8; ++q) jwv[2 * ((q / 2) * np + j) + (q & 1)] = w8[8 * j + q];
}

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
BatchPlan plan_uniform(const float *d_pcm, float *d_out, uint64_t clip_stride, uint64_t frames_per_clip,
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
    // together (RoundSync in melspec_kernels.hpp): -1 = the measured best of the kernel that runs, resolved in launch_ctx.
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

__global__ void plan_upload_kernel(const uint4 *src, uint4 *dst, size_t n16) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) dst[i] = src[i];
}

// Fills the next slot and queues its upload on `stream`.  The caller launches on `stream` and then calls plan_ragged_done.
// want_order: also upload the clips sorted longest first and a zeroed ticket counter (BatchDesc::d_order / d_ticket) for the kernels that
// hand out whole clips.
int plan_ragged(RaggedScratch &rs, hipStream_t stream, const float *d_pcm, float *d_out, const uint64_t *h_off,
                const std::vector<uint64_t> &frames, const uint64_t *h_out_off, uint32_t n_clips, int n_mels,
                int frames_per_unit, BatchPlan &pl, RaggedSlot *&used, bool want_order = false) {
    RaggedSlot &sl = rs.slot[rs.next++ % RaggedScratch::kSlots];
    used = &sl;
    if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.pending) { HIP_TRY(hipEventSynchronize(sl.ev)); sl.pending = false; }
    uint64_t units = 0;
    for (uint32_t c = 0; c < n_clips; ++c) units += (frames[c] + frames_per_unit - 1) / frames_per_unit;
    const uint64_t n_blocks = (units + kUnitBlock - 1) / kUnitBlock;
    const size_t words64 = static_cast<size_t>(n_clips) * 4 + 1;
    const size_t blk_words = static_cast<size_t>(n_blocks ? n_blocks : 1);
    const size_t bytes = words64 * sizeof(uint64_t) + (blk_words + (want_order ? static_cast<size_t>(n_clips) + 1 : 0)) * sizeof(uint32_t);
    int rc = sl.ensure_host((bytes + 15) & ~static_cast<size_t>(15));
    if (rc) return rc;
    if ((rc = sl.dev.ensure((bytes + 15) & ~static_cast<size_t>(15)))) return rc;
    uint64_t *off = static_cast<uint64_t *>(sl.host), *fr = off + n_clips, *oo = fr + n_clips, *pre = oo + n_clips;
    uint32_t *blk = reinterpret_cast<uint32_t *>(off + words64);
    uint64_t cursor = 0, out_cursor = 0, total = 0;
    for (uint32_t c = 0; c < n_clips; ++c) {
        off[c] = h_off[c];
        fr[c] = frames[c];
        oo[c] = h_out_off ? h_out_off[c] : out_cursor;
        out_cursor += frames[c] * static_cast<uint64_t>(n_mels);
        pre[c] = cursor;
        cursor += (frames[c] + frames_per_unit - 1) / frames_per_unit;
        total += frames[c];
    }
    pre[n_clips] = units;
    {
        uint32_t c = 0;
        for (uint64_t k = 0; k < n_blocks; ++k) {
            const uint64_t u = k * kUnitBlock;
            while (pre[c + 1] <= u) ++c;      // u < units = pre[n_clips]
            blk[k] = c;
        }
    }
    if (want_order) {
        uint32_t *ord = blk + blk_words;
        for (uint32_t c = 0; c < n_clips; ++c) ord[c] = c;
        std::stable_sort(ord, ord + n_clips, [&](uint32_t a, uint32_t b2) { return frames[a] > frames[b2]; });
        ord[n_clips] = 0;       // the ticket counter
    }
    // the upload is a kernel on the launch stream that reads the pinned slot over the bus: an SDMA copy sits in another
    // hardware queue and the hand-over between the queues costs more than the copy
    {
        const size_t n16 = (bytes + 15) / 16;
        const unsigned blocks = static_cast<unsigned>((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
        hipLaunchKernelGGL(plan_upload_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const uint4 *>(sl.host),
                           static_cast<uint4 *>(sl.dev.p), n16);
        HIP_TRY(hipGetLastError());
    }
    const uint64_t *d = static_cast<const uint64_t *>(sl.dev.p);
    BatchDesc &b = pl.desc;
    b = BatchDesc{};
    b.pcm = d_pcm; b.out = d_out; b.n_clips = n_clips; b.n_units = units; b.frames_per_unit = frames_per_unit;
    b.d_off = d; b.d_frames = d + n_clips; b.d_out_off = d + 2 * n_clips; b.d_unit_prefix = d + 3 * n_clips;
    b.d_unit_block = reinterpret_cast<const uint32_t *>(d + words64);
    if (want_order) {
        b.d_order = b.d_unit_block + blk_words;
        b.d_ticket = const_cast<uint32_t *>(b.d_order) + n_clips;
    }
    pl.total_frames = total;
    b.stat_frames = total;
    return MELSPEC_OK;
}
// behind the launch (or the failed attempt) that used the slot
void plan_ragged_done(RaggedSlot *sl, hipStream_t stream) {
    if (sl && sl->ev && hipEventRecord(sl->ev, stream) == hipSuccess) sl->pending = true;
}

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
                       int frames_per_unit, uint64_t max_total_frames, BatchPlan &pl) {
    // every clip with frames has at most frames / fpu + 1 units
    const uint64_t max_units = max_total_frames / frames_per_unit + n_clips;
    const uint64_t max_blocks = max_units / kUnitBlock + 2;
    const size_t words64 = static_cast<size_t>(n_clips) * 4 + 1;
    const size_t bytes = words64 * sizeof(uint64_t) + static_cast<size_t>(max_blocks) * sizeof(uint32_t) + 16;
    if (dp.used && dp.last != stream) HIP_TRY(hipStreamSynchronize(dp.last));
    if (bytes > dp.buf.cap && dp.used) HIP_TRY(hipStreamSynchronize(dp.last));
    int rc = dp.buf.ensure(bytes);
    if (rc) return rc;
    dp.used = true; dp.last = stream;
    PlanParams q{};
    q.d_off = d_off; q.d_len = d_len; q.d_out_off = d_out_off; q.n_clips = n_clips;
    q.frame_len = frame_len; q.frame_shift = frame_shift; q.words_per_frame = words_per_frame;
    q.frames_per_unit = static_cast<uint32_t>(frames_per_unit);
    q.plan = static_cast<uint64_t *>(dp.buf.p);
    q.max_blocks = max_blocks;
    hipLaunchKernelGGL(plan_ragged_device_kernel, dim3(1), dim3(1024), 0, stream, q);
    HIP_TRY(hipGetLastError());
    const uint64_t *d = q.plan;
    BatchDesc &b = pl.desc;
    b = BatchDesc{};
    b.pcm = d_pcm; b.out = d_out; b.n_clips = n_clips; b.n_units = max_units; b.frames_per_unit = frames_per_unit;
    b.d_off = d; b.d_frames = d + n_clips; b.d_out_off = d + 2 * n_clips; b.d_unit_prefix = d + 3 * n_clips;
    b.d_unit_block = reinterpret_cast<const uint32_t *>(d + words64);
    b.d_n_units = d + 4 * static_cast<size_t>(n_clips);        // prefix[n_clips]
    pl.total_frames = max_total_frames;
    return MELSPEC_OK;
}

unsigned grid_for(uint64_t units, int cus, int per_cu) {
    const uint64_t cap = static_cast<uint64_t>(cus > 0 ? cus : 256) * per_cu;
    const uint64_t g = units < cap ? units : cap;
    return static_cast<unsigned>(g ? g : 1);
}
// same, rounded up to a multiple of the 8 XCDs for the kernels that reorder their workgroups (xcd_logical_block);
// lab builds: MELSPEC_XCD=0 keeps the dispatcher's order (odd grid sizes switch the reordering off in the kernel)
unsigned grid_for_xcd(uint64_t units, int cus, int per_cu) {
    static const bool off = lab_int("MELSPEC_XCD", 1, 0, 1) == 0;
    const unsigned g = grid_for(units, cus, per_cu);
    if (off) return (g % 8 == 0 && g > 1) ? g - 1 : g;
    return (g + 7u) & ~7u;
}

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
                   double preemph, double floor_v, int cus, hipStream_t stream, long long clip_len = 0, int pad = 0) {
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

}  // namespace

namespace {
// Waves per workgroup of the fused 512-point kernels: 8 (two per SIMD, one workgroup per CU) when the tables
// and eight 18.5 KB slices fit in LDS, else 4.
int fused512_waves(size_t blob_bytes, size_t slice_bytes) {
    if (lab_int("MELSPEC_FB_WAVES", 8, 4, 8) == 4) return 4;
    return blob_bytes + 8 * slice_bytes <= kLdsLimit ? 8 : 4;
}

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
    }
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
    void params(FbankFastParams &fp) const {
        fp.d_blob = static_cast<const uint32_t *>(d_blob.p);
        fp.blob_words = static_cast<int>(ft.blob.size());
        fp.mel_off_words = ft.mel_off_words;
        fp.slots = ft.slots;
    }
};
bool w512_f32_bank(const MelSlots &ms) { return fb_lens_match<LensSlaney80W>(ms) || fb_lens_match<LensSlaney128>(ms); }
bool nemo_f32_bank(const MelSlots &ms) { return fb_lens_match<LensSlaney128>(ms) || fb_lens_match<LensSlaney80>(ms); }
int launch_w512_f32(const Fused512F32 &f, FbankFastParams fp, int cus, hipStream_t s) {
    f.params(fp);
    if (fb_lens_match<LensSlaney80W>(f.ft.slots)) return launch_fused512<float, kFlavorWhisper, kFbSlots, LensSlaney80W>(kFused512F32Waves, fp, f.lds, cus, s);
    return launch_fused512<float, kFlavorWhisper, kBlmSlots, LensSlaney128>(kFused512F32Waves, fp, f.lds, cus, s);
}
int launch_nemo_f32(const Fused512F32 &f, FbankFastParams fp, int cus, hipStream_t s) {
    f.params(fp);
    fp.b.sync_rounds = 0;        // StagedRows instead of RoundSync
    if (fb_lens_match<LensSlaney128>(f.ft.slots)) return launch_fused512<float, kFlavorNemo, kBlmSlots, LensSlaney128>(kFused512F32Waves, fp, f.lds, cus, s);
    return launch_fused512<float, kFlavorNemo, kFbSlots, LensSlaney80>(kFused512F32Waves, fp, f.lds, cus, s);
}
}  // namespace

// ------------------------------------------------------------------------------------
// Whisper log-mel context
// ------------------------------------------------------------------------------------
// MELSPEC_PRECISION_AUTO state (FixSink in melspec_kernels.hpp): the f64 tables of the in-kernel recompute and its counter.
struct FixState {
    DevBuf tab, count, list;              // count: {u64 frames that tripped the guard, u64 accumulator of the launch in flight, u64 tally of the vote}
    DevBuf verdicts;                      // FixSink::decision: kVoteSlots copies of the last vote's verdict
    hipStream_t last_stream = nullptr;    // the note list is used in stream order: a call on another stream first waits for this one
    bool used = false;
    // Statistics of the guarded launches, published by the kernels into host-mapped memory (FixSink::host) and read here without
    // touching the stream (melspec_auto_state, melspec_guard_count's cheap sibling).  They no longer decide anything: since round 4
    // the kernel a batch runs on is decided by a vote inside the batch's own launch (FixSink::vote in melspec_kernels.hpp), so the
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

namespace {

// MELSPEC_PRECISION_AUTO: take in what the finished launches published (reporting only: melspec_auto_state)
void auto_poll(melspec_ctx *c) {
    FixState &fx = c->fix;
    if (!fx.host) return;
    const volatile unsigned long long *h = fx.host;
    const unsigned long long a = h[0], b = h[1];
    const uint32_t seq = static_cast<uint32_t>(a >> kStatShift);
    if (seq != static_cast<uint32_t>(b >> kStatShift) || seq == fx.seen_seq) return;     // a launch is publishing right now, or nothing new
    fx.seen_seq = seq;
    const unsigned long long tripped = a & kStatMask, frames = b & kStatMask & ~kStatFromGated;
    fx.heavy = (b & kStatFromGated) != 0;
    if (frames < kAutoMinFrames) return;
    fx.fraction = static_cast<double>(tripped) / static_cast<double>(frames);
}

// frames per work unit of the kernel a batch is planned for (called once per batch, before it is planned).  AUTO plans for the f32
// kernel: when the batch's vote says "heavy", the f64 kernel walks the same plan (whisper400_precise_kernel, MODE 2).
// layout: a padded / mel-major batch (the f64 kernel of the layouts is the five-frame one)
// The six-frame f64 kernel serves a padded / mel-major batch only with one of the compile-time banks: its run-time-lens layout instantiation
// keeps 141 SGPRs' worth of slot tables and reloads 13 spilled registers inside the unit loop (tools/hotloop_spills.py); those banks stay
// on whisper400_precise_kernel's layout form.
bool six64_layout_ok(const melspec_ctx *c) { return c->six64 && !c->six64_wide && c->six_static != 0; }

int ctx_frames_per_unit(melspec_ctx *c, bool layout = false) {
    if (c->fast) {
        if (c->precision == MELSPEC_PRECISION_F64) return (layout ? six64_layout_ok(c) : c->six64) ? kSixFrames : kFPW;
        return c->six ? kSixFrames : kFPW;
    }
    return c->fast512 ? kFbFPW : 1;
}

int ctx_num_frames(const melspec_ctx *c, uint64_t n, uint64_t &frames) {
    frames = n < static_cast<uint64_t>(c->fft_size) ? 0 : (n - c->fft_size) / c->hop_size + 1;
    return MELSPEC_OK;
}

// the launch-specific part of a guarded launch's statistics sink (the grid is only known where the launch is made)
FixSink sink_armed(melspec_ctx *c, FixSink sink, const BatchDesc &desc, unsigned grid) {
    if (!sink.acc) return sink;
    sink.frames = desc.stat_frames ? desc.stat_frames
                : desc.d_unit_prefix == nullptr ? static_cast<uint64_t>(desc.n_clips) * desc.frames_per_clip
                                                : desc.n_units * static_cast<uint64_t>(desc.frames_per_unit);   // device-planned ragged: upper bound
    sink.n_groups = grid;
    sink.seq = (c->fix.seq = (c->fix.seq + 1) & 0xffffffu) ? c->fix.seq : (c->fix.seq = 1);      // never 0: the host's "nothing seen yet"
    return sink;
}

PreciseParams precise_params(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat) {
    PreciseParams pp{};
    pp.b = desc;
    pp.stat = stat;
    pp.d_blob = static_cast<const uint32_t *>(c->d_blob64.p);
    pp.blob_words = static_cast<int>(c->pt.blob.size());
    pp.mel_off_words = c->pt.mel_off_words;
    pp.hop = c->hop_size;
    pp.n_mels = c->n_mels;
    pp.slots = c->ft.slots;
    return pp;
}

// the f64 kernel on the whole batch: MELSPEC_PRECISION_F64 (the plan is its own, kFPW frames per unit), or -- gate != nullptr -- AUTO's
// second launch, which runs only when the f32 launch in front of it voted "heavy" and walks THAT launch's plan (plain batches)
template <int NSLOTS, class Lens>
int launch_precise_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate, unsigned gate_value) {
    static std::atomic<uint64_t> attr_done{0};          // one bit per device: function attributes are per device
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_precise_kernel<NSLOTS, Lens, 0>, "hipFuncSetAttribute(whisper400_precise_kernel)");
        if (!rc) rc = allow_big_lds(&whisper400_precise_kernel<NSLOTS, Lens, 1>, "hipFuncSetAttribute(whisper400_precise_kernel, runs)");
        if (!rc) rc = allow_big_lds(&whisper400_precise_kernel<NSLOTS, Lens, 2>, "hipFuncSetAttribute(whisper400_precise_kernel, gated)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const bool walk = gate && !(desc.mel_major || desc.out_width != desc.frames_per_clip);      // gated layouts come with a plan of their own
    const uint64_t steps = walk ? (desc.n_units * static_cast<uint64_t>(desc.frames_per_unit) + kFPW - 1) / kFPW : desc.n_units;
    const uint64_t blocks = (steps + kPreciseWaves - 1) / kPreciseWaves;
    static const int per_cu = lab_int("MELSPEC_PRECISE_GRID_PER_CU", 1, 1, 4096);   // one workgroup is resident per CU
    const unsigned grid = grid_for_xcd(blocks, c->dev.cus, per_cu);
    FixSink armed = sink_armed(c, stat, desc, grid);
    if (gate) armed.frames |= kStatFromGated;
    PreciseParams pp = precise_params(c, desc, armed);
    pp.gate = gate; pp.gate_value = gate_value; pp.plan_fpu = desc.frames_per_unit;
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (gate && !layout)
        hipLaunchKernelGGL((whisper400_precise_kernel<NSLOTS, Lens, 2>), dim3(grid), dim3(kPreciseWaves * 64), c->precise_lds, stream, pp);
    else if (layout)
        hipLaunchKernelGGL((whisper400_precise_kernel<NSLOTS, Lens, 0>), dim3(grid), dim3(kPreciseWaves * 64), c->precise_lds, stream, pp);
    else
        hipLaunchKernelGGL((whisper400_precise_kernel<NSLOTS, Lens, 1>), dim3(grid), dim3(kPreciseWaves * 64), c->precise_lds, stream, pp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_precise(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate = nullptr, unsigned gate_value = 0) {
    if (c->ft.slots.n_slots <= 8)
        return c->lens_kind == 1 ? launch_precise_t<8, LensI80>(c, desc, stat, stream, gate, gate_value) : launch_precise_t<8, LensRuntime>(c, desc, stat, stream, gate, gate_value);
    return c->lens_kind == 2 ? launch_precise_t<12, LensI128>(c, desc, stat, stream, gate, gate_value) : launch_precise_t<12, LensRuntime>(c, desc, stat, stream, gate, gate_value);
}

// the f64 six-frame kernel on a plain batch planned in six-frame units: MELSPEC_PRECISION_F64, or -- gate != nullptr -- AUTO's second
// launch over the plan of the f32 launch in front of it
template <class Lens, int NS = kSixMaxSlots>
int launch_six64_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate, unsigned gate_value) {
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_six64_kernel<NS, Lens>, "hipFuncSetAttribute(whisper400_six64_kernel)");
        if constexpr (Lens::kStatic && NS == kSixMaxSlots)          // the layout form exists for the compile-time banks of up to 80 mels only (six64_layout_ok)
            if (!rc) rc = allow_big_lds(&whisper400_six64_layout_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six64_layout_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t blocks = (desc.n_units + kSix64Waves - 1) / kSix64Waves;
    static const int per_cu = lab_int("MELSPEC_SIX64_GRID_PER_CU", 1, 1, 4096);   // one 12-wave workgroup is resident per CU
    const unsigned grid = grid_for_xcd(blocks, c->dev.cus, per_cu);
    FixSink armed = sink_armed(c, stat, desc, grid);
    if (gate) armed.frames |= kStatFromGated;
    Six64Params pp{};
    pp.b = desc;
    pp.stat = armed;
    pp.d_blob = static_cast<const uint32_t *>(c->d_blob64x.p);
    pp.blob_words = static_cast<int>(c->t64.blob.size());
    pp.mel_off_words = c->t64.mel_off_words;
    pp.hop = c->hop_size;
    pp.n_mels = c->n_mels;
    pp.slots = c->ft6.slots;
    pp.gate = gate; pp.gate_value = gate_value;
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (layout) {
        if constexpr (Lens::kStatic && NS == kSixMaxSlots) hipLaunchKernelGGL((whisper400_six64_layout_kernel<kSixMaxSlots, Lens>), dim3(grid), dim3(kSix64Waves * 64), c->lds64x, stream, pp);
        else return fail(MELSPEC_ERR_INTERNAL, "whisper400_six64_layout_kernel has no run-time-lens form");      // launch_ctx never asks (six64_layout_ok)
    } else {
        hipLaunchKernelGGL((whisper400_six64_kernel<NS, Lens>), dim3(grid), dim3(kSix64Waves * 64), c->lds64x, stream, pp);
    }
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_six64(melspec_ctx *c, const BatchDesc &desc, const FixSink &stat, hipStream_t stream, const unsigned *gate = nullptr, unsigned gate_value = 0) {
    if (c->six64_wide) return launch_six64_t<LensSix128, kSixWideSlots>(c, desc, stat, stream, gate, gate_value);
    return c->six_static == 1 ? launch_six64_t<LensSix80>(c, desc, stat, stream, gate, gate_value)
         : c->six_static == 2 ? launch_six64_t<LensSix64>(c, desc, stat, stream, gate, gate_value)
         : c->six_static == 3 ? launch_six64_t<LensSix40>(c, desc, stat, stream, gate, gate_value)
                              : launch_six64_t<LensRuntime>(c, desc, stat, stream, gate, gate_value);
}

FastParams fast_params(const BatchDesc &desc, const FastTables &ft, const DevBuf &blob, melspec_ctx *c, const FixSink &sink) {
    FastParams fp{};
    fp.b = desc;
    fp.d_blob = static_cast<const float *>(blob.p);
    fp.blob_len = static_cast<int>(ft.blob.size());
    fp.hop = c->hop_size;
    fp.n_mels = c->n_mels;
    fp.slice_floats = WaveLayout::slice_floats();
    fp.slots = ft.slots;
    fp.fix = sink;
    return fp;
}

// 5-frame f32 kernels: plain batches (uniform, ragged) on contiguous runs of units per wave, layouts round-robin
template <int NSLOTS, class Lens>
int launch_wave_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &sink, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_wave_kernel<NSLOTS, Lens>, "hipFuncSetAttribute(whisper400_wave_kernel)");
        if (!rc) rc = allow_big_lds(&whisper400_wave_runs_kernel<NSLOTS, Lens>, "hipFuncSetAttribute(whisper400_wave_runs_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t blocks = (desc.n_units + kWaveWaves - 1) / kWaveWaves;
    // two workgroups are resident per CU; 4 per CU measured best (8192 x 15..45 s x 128 mels: 9.17 vs 9.50 ms)
    static const int per_cu = lab_int("MELSPEC_GRID_PER_CU", 4, 1, 64);
    const unsigned grid = grid_for_xcd(blocks, c->dev.cus, per_cu);
    FixSink armed = sink_armed(c, sink, desc, grid);
    armed.vote_groups = std::min<unsigned>(grid, static_cast<unsigned>(c->dev.cus));           // workgroups that are certainly resident when the launch starts
    const FastParams fp = fast_params(desc, c->ft, c->d_blob, c, armed);
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (layout)
        hipLaunchKernelGGL((whisper400_wave_kernel<NSLOTS, Lens>), dim3(grid), dim3(kWaveWaves * 64), c->fast_lds, stream, fp);
    else
        hipLaunchKernelGGL((whisper400_wave_runs_kernel<NSLOTS, Lens>), dim3(grid), dim3(kWaveWaves * 64), c->fast_lds, stream, fp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_wave(melspec_ctx *c, const BatchDesc &desc, const FixSink &sink, hipStream_t stream) {
    if (c->ft.slots.n_slots <= 8)
        return c->lens_kind == 1 ? launch_wave_t<8, LensI80>(c, desc, sink, stream) : launch_wave_t<8, LensRuntime>(c, desc, sink, stream);
    return c->lens_kind == 2 ? launch_wave_t<12, LensI128>(c, desc, sink, stream) : launch_wave_t<12, LensRuntime>(c, desc, sink, stream);
}

template <class Lens>
int launch_six_t(melspec_ctx *c, const BatchDesc &desc, const FixSink &sink, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    if (!device_done(attr_done)) {
        int rc = allow_big_lds(&whisper400_six_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six_kernel)");
        if (!rc) rc = allow_big_lds(&whisper400_six_runs_kernel<kSixMaxSlots, Lens>, "hipFuncSetAttribute(whisper400_six_runs_kernel)");
        if (rc) return rc;
        mark_device_done(attr_done);
    }
    const uint64_t blocks = (desc.n_units + kSixWaves - 1) / kSixWaves;
    static const int per_cu = lab_int("MELSPEC_SIX_GRID_PER_CU", 1, 1, 4096);     // one 16-wave workgroup per CU
    const dim3 grid(grid_for_xcd(blocks, c->dev.cus, per_cu)), block(kSixWaves * 64);
    FixSink armed = sink_armed(c, sink, desc, grid.x);
    armed.vote_groups = std::min<unsigned>(grid.x, static_cast<unsigned>(c->dev.cus));        // the workgroups resident when the launch starts (one per CU)
    const FastParams fp = fast_params(desc, c->ft6, c->d_blob6, c, armed);
    const bool layout = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    // plain batches, uniform and ragged, take the run-per-wave kernel (no division per unit, the clip record in scalar registers, a
    // wave re-reads its own frame-tail halo): cfg2 0.3105 -> 0.3055 ms, 8192 x 30 s 7.55 -> 7.42 ms against the round-robin deal
    if (layout) hipLaunchKernelGGL((whisper400_six_kernel<kSixMaxSlots, Lens>), grid, block, c->lds6, stream, fp);
    else hipLaunchKernelGGL((whisper400_six_runs_kernel<kSixMaxSlots, Lens>), grid, block, c->lds6, stream, fp);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int launch_ctx(melspec_ctx *c, const BatchDesc &desc_in, hipStream_t stream) {
    if (desc_in.n_units == 0) return MELSPEC_OK;
    BatchDesc desc = desc_in;
    const bool layout_batch = desc.mel_major || desc.out_width != desc.frames_per_clip;   // ragged batches: both zero
    if (desc.sync_rounds < 0) {
        // measured (profiles/r01_variants.txt): six-frame kernel, 16 waves: four waves 4 apart; precise kernel, 8 waves:
        // consecutive pairs; 5-frame kernel, two 8-wave workgroups per CU: pairs 4 apart
        if (c->fast && desc.frames_per_unit == kSixFrames) desc.sync_rounds = 20;
        else if (c->fast && c->precision == MELSPEC_PRECISION_F64) desc.sync_rounds = 2;
        else if (c->fast) desc.sync_rounds = 18;
        else desc.sync_rounds = 1;
    }
    if (!c->fast && c->fast512 && desc.frames_per_unit == kFbFPW) {
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
        if (fb_lens_match<LensSlaney80W>(c->ft512.slots)) return launch_fused512<double, kFlavorWhisper, kFbSlots, LensSlaney80W>(c->waves512, fp, c->lds512, c->dev.cus, stream);
        if (fb_lens_match<LensSlaney128>(c->ft512.slots)) return launch_fused512<double, kFlavorWhisper, kBlmSlots, LensSlaney128>(c->waves512, fp, c->lds512, c->dev.cus, stream);
        return c->ft512.slots.n_slots <= kFbSlots ? launch_fused512<double, kFlavorWhisper, kFbSlots>(c->waves512, fp, c->lds512, c->dev.cus, stream)
                                                  : launch_fused512<double, kFlavorWhisper, kBlmSlots>(c->waves512, fp, c->lds512, c->dev.cus, stream);
    }
    if (!c->fast) return launch_generic(c->gt, desc, c->hop_size, 0, 1, 1, 0.0, 0.0, c->dev.cus, stream);
    if (c->precision == MELSPEC_PRECISION_F64) {
        if ((layout_batch ? six64_layout_ok(c) : c->six64) && desc.frames_per_unit == kSixFrames && desc.d_unit_prefix == nullptr) {
            // mel-major stores of the twelve-wave kernel, measured (tools/mm64_sync_probe.py, 1024 x 10 s): consecutive pairs 0.491 ms, none 0.493,
            // pairs four apart 0.496, fours 0.512, fours one from each SIMD (the f32 kernel's best) 0.520, workgroup barrier 0.533
            if (layout_batch && desc_in.sync_rounds < 0) desc.sync_rounds = 2;
            return launch_six64(c, desc, FixSink{}, stream);
        }
        if (c->six64 && !layout_batch && desc.frames_per_unit == kSixFrames) return launch_six64(c, desc, FixSink{}, stream);
        return launch_precise(c, desc, FixSink{}, stream);
    }
    FixSink sink{};
    bool vote = false;
    if (c->precision == MELSPEC_PRECISION_AUTO) {
        FixState &fx = c->fix;
        if (fx.used && fx.last_stream != stream) HIP_TRY(hipStreamSynchronize(fx.last_stream));
        const size_t need = (static_cast<size_t>(desc.n_units) + 65536) * sizeof(uint64_t);      // one note per unit + a round of slack
        if (need > fx.list.cap) {
            if (fx.used) HIP_TRY(hipStreamSynchronize(fx.last_stream));       // a launch in flight may still write the old list
            int rc = fx.list.ensure(need);
            if (rc) return rc;
        }
        sink.tab = static_cast<const double *>(fx.tab.p);
        sink.list = static_cast<uint64_t *>(fx.list.p);
        if (!fx.host) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&fx.host), 64, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(fx.host, 0, 64);
        }
        fx.used = true; fx.last_stream = stream;
        sink.count = static_cast<unsigned long long *>(fx.count.p);
        sink.acc = sink.count + 1;
        sink.host = fx.host;
        // The vote (FixSink::vote): plain batches and the padded / mel-major layouts (whose sample is the head of the batch: they deal
        // their units round-robin).  Not where the mel kernel also leaves the image extremes for the TGA quantiser (d_unit_ext: the two
        // kernels' units differ): PCM -> TGA keeps the f32 kernel + recompute tail whatever the input.
        vote = fx.adaptive && (!layout_batch || desc.d_unit_ext == nullptr);
        if (vote) {
            sink.vote = sink.count + 2;
            sink.decision = static_cast<unsigned *>(fx.verdicts.p);
        }
    }
    int rc;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (c->first_kernel_events) {
        HIP_TRY(hipEventCreate(&pe0)); HIP_TRY(hipEventCreate(&pe1));
        c->first_kernel_events->push_back(pe0); c->first_kernel_events->push_back(pe1);
        HIP_TRY(hipEventRecord(pe0, stream));
    }
    if (c->six && desc.frames_per_unit == kSixFrames)
        rc = c->six_static == 1 ? launch_six_t<LensSix80>(c, desc, sink, stream) : c->six_static == 2 ? launch_six_t<LensSix64>(c, desc, sink, stream)
           : c->six_static == 3 ? launch_six_t<LensSix40>(c, desc, sink, stream) : launch_six_t<LensRuntime>(c, desc, sink, stream);
    else
        rc = launch_wave(c, desc, sink, stream);
    if (pe1) HIP_TRY(hipEventRecord(pe1, stream));
    if (rc || !vote) return rc;
    // AUTO's second launch: returns at its first instruction unless the launch above voted "heavy" (its number is c->fix.seq)
    const unsigned gate_value = (c->fix.seq & 0xffffffu) << 2 | kVoteDecided | kVoteHeavy;
    FixSink stat{};
    stat.count = sink.count; stat.acc = sink.acc; stat.host = sink.host;
    if (six64_layout_ok(c) && layout_batch && desc.frames_per_unit == kSixFrames && desc.d_unit_prefix == nullptr) {
        if (desc_in.sync_rounds < 0) desc.sync_rounds = 2;
        return launch_six64(c, desc, stat, stream, sink.decision, gate_value);          // the layouts on the six-frame f64 kernel: the f32 launch's own plan
    }
    if (layout_batch && desc.frames_per_unit != kFPW) {
        // the layouts' f64 kernel deals units of its own size: the same (uniform) batch planned for five frames per unit
        BatchPlan p5 = plan_uniform(desc.pcm, desc.out, desc.clip_stride, desc.frames_per_clip, desc.n_clips, c->n_mels, kFPW, desc.out_width, desc.mel_major != 0);
        if (p5.desc.sync_rounds < 0) p5.desc.sync_rounds = 2;          // the precise kernel's measured grouping (consecutive pairs)
        return launch_precise(c, p5.desc, stat, stream, sink.decision, gate_value);
    }
    if (layout_batch && desc_in.sync_rounds < 0) desc.sync_rounds = 2;
    if (c->six64 && !layout_batch && desc.frames_per_unit == kSixFrames) return launch_six64(c, desc, stat, stream, sink.decision, gate_value);
    if (c->six64_wide && !layout_batch && desc.d_unit_prefix == nullptr) {
        // 128 mels: the f32 launch walked five-frame units, the gated kernel deals six -- the same uniform batch planned again (arithmetic only;
        // a ragged batch's plan lives in device arrays made for five-frame units: those stay on the precise kernel)
        const BatchPlan p6 = plan_uniform(desc.pcm, desc.out, desc.clip_stride, desc.frames_per_clip, desc.n_clips, c->n_mels, kSixFrames);
        return launch_six64(c, p6.desc, stat, stream, sink.decision, gate_value);
    }
    return launch_precise(c, desc, stat, stream, sink.decision, gate_value);
}

template <class Lens>
bool lens_match(const MelSlots &ms, int n_mels) {
    if (ms.n_slots != Lens::kSlots || n_mels != Lens::kMels) return false;
    for (int i = 0; i < Lens::kSlots; ++i)
        if (ms.len[i] != Lens::len(i) || ms.woff[i] != Lens::woff(i)) return false;
    return true;
}

}  // namespace

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
        if ((rc = allow_big_lds(&generic_frame_kernel<kGenericNT>, "hipFuncSetAttribute(generic_frame_kernel)"))) return bail(rc);
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
    c->d_blob.release(); c->d_blob64.release(); c->d_blob64s.release(); c->d_blob512.release(); c->f512.d_blob.release(); c->d_blob6.release(); c->d_blob64x.release(); c->gt.release(); c->ragged.release();
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
    return c->precision == MELSPEC_PRECISION_F32 && c->fast512 && c->f512.ok ? MELSPEC_PRECISION_F32 : MELSPEC_PRECISION_F64;
}
int melspec_set_precise(melspec_ctx *c, int on) { return melspec_set_precision(c, on ? MELSPEC_PRECISION_F64 : MELSPEC_PRECISION_AUTO); }
int melspec_is_precise(const melspec_ctx *c) { return c && melspec_precision(c) == MELSPEC_PRECISION_F64 ? 1 : 0; }   // the generic kernels are f64 whatever the mode

const char *melspec_plain_kernel_name(const melspec_ctx *c) {
    // the same decisions launch_ctx takes for a plain (uniform or ragged, [frame][mel]) batch
    if (!c) return "";
    if (!c->fast) {
        if (c->fast512 && c->precision == MELSPEC_PRECISION_F32 && c->f512.ok) return "melspec::fbank512_wave_kernel<float, 12, 1, kFlavorWhisper, RUNS> (n_fft = 512, f32, three waves per SIMD)";
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
             : c->six_static == 2 ? (fix ? "melspec::whisper400_six_runs_kernel<9, LensSix64> (precision guard on)" : "melspec::whisper400_six_runs_kernel<9, LensSix64>")
             : c->six_static == 3 ? (fix ? "melspec::whisper400_six_runs_kernel<9, LensSix40> (precision guard on)" : "melspec::whisper400_six_runs_kernel<9, LensSix40>")
                             : (fix ? "melspec::whisper400_six_runs_kernel<9, LensRuntime> (precision guard on)" : "melspec::whisper400_six_runs_kernel<9, LensRuntime>");
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
    if (c->fast && c->precision == MELSPEC_PRECISION_AUTO) auto_poll(c);
    if (heavy) *heavy = (c->fast && c->precision == MELSPEC_PRECISION_AUTO && c->fix.heavy) ? 1 : 0;
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

// interleave_frames' width rule (src/mel.rs:497-516)
static uint64_t interleaved_width(uint64_t frames, uint64_t min_width) {
    uint64_t nf = frames;
    if (min_width > 0 && (nf & 1)) nf += 1;
    return nf > min_width ? nf : min_width;
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

// ---- the mel stage on its own: MelSpectrogram::add(&fft) (src/mel.rs:13-32) over complex STFT frames --------------------------
namespace {
int stage_tables(melspec_ctx *c) {
    if (c->stage_built) return MELSPEC_OK;
    const int bins = c->fft_size / 2 + 1;
    const BandedFilterbank fb = band_filterbank(c->dense, c->n_mels, bins, c->fft_size / 2);      // bins >= n_fft/2 contribute nothing (src/mel.rs:155-163)
    int rc;
    if ((rc = upload(c->st_start, fb.start))) return rc;
    if ((rc = upload(c->st_len, fb.len))) return rc;
    if ((rc = upload(c->st_off, fb.offset))) return rc;
    if ((rc = upload(c->st_w, fb.w))) return rc;
    {
        std::vector<double> jwv;
        std::vector<int> jobv;
        build_mel_jobs(fb, c->n_mels, 64, jwv, jobv);
        c->st_n_jobs = jobv.size() == 1 && (jobv[0] >> 20) == 0 ? 0 : static_cast<int>(jobv.size());
        if ((rc = upload(c->st_jw, jwv))) return rc;
        if ((rc = upload(c->st_job, jobv))) return rc;
    }
    c->stage_built = true;
    return MELSPEC_OK;
}
}  // namespace

int melspec_mel_from_stft_device(melspec_ctx *c, const void *d_spec, int dtype, int full, uint64_t n_frames, float *d_out, void *stream) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!d_spec || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    int rc = stage_tables(c);
    if (rc) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    MelStageParams p{};
    p.spec = d_spec; p.out = d_out; p.n_frames = n_frames;
    p.stride = static_cast<uint32_t>(full ? c->fft_size : c->fft_size / 2 + 1);
    p.bin_limit = c->fft_size / 2; p.n_mels = c->n_mels;
    p.d_mstart = static_cast<const int *>(c->st_start.p); p.d_mlen = static_cast<const int *>(c->st_len.p);
    p.d_moff = static_cast<const int *>(c->st_off.p); p.d_mw = static_cast<const double *>(c->st_w.p);
    p.d_jw = static_cast<const double *>(c->st_jw.p); p.d_job = static_cast<const int *>(c->st_job.p); p.n_jobs = c->st_n_jobs;
    {
        // the wave-per-frame form with the bank as jobs in LDS (banks of up to 256 mels over up to 4088 bins that fit)
        const size_t lds = sizeof(double) * static_cast<size_t>(mel_stage_lds(p.n_jobs, p.bin_limit, p.n_mels, kMelStageWaves).total);
        if (p.n_jobs > 0 && p.n_mels <= 256 && p.bin_limit <= 4088 && lds <= 64 * 1024) {
            const unsigned grid = grid_for((n_frames + kMelStageWaves - 1) / kMelStageWaves, c->dev.cus, static_cast<int>(std::max<size_t>(1, std::min<size_t>(4, kLdsLimit / lds))));
            if (dtype == MELSPEC_STFT_F64) hipLaunchKernelGGL((mel_stage_jobs_kernel<double>), dim3(grid), dim3(kMelStageWaves * 64), lds, s, p);
            else hipLaunchKernelGGL((mel_stage_jobs_kernel<float>), dim3(grid), dim3(kMelStageWaves * 64), lds, s, p);
            HIP_TRY(hipGetLastError());
            return MELSPEC_OK;
        }
    }
    constexpr int kWaves = 4;
    const size_t lds = static_cast<size_t>(kWaves) * (p.bin_limit + p.n_mels) * sizeof(double);
    if (lds > 64 * 1024) return fail(MELSPEC_ERR_UNSUPPORTED, "geometry needs more LDS than the mel stage kernel has");
    const unsigned grid = grid_for((n_frames + kWaves - 1) / kWaves, c->dev.cus, 16);
    if (dtype == MELSPEC_STFT_F64) hipLaunchKernelGGL((mel_stage_kernel<double, kWaves>), dim3(grid), dim3(kWaves * 64), lds, s, p);
    else hipLaunchKernelGGL((mel_stage_kernel<float, kWaves>), dim3(grid), dim3(kWaves * 64), lds, s, p);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_mel_from_stft_host(melspec_ctx *c, const void *spec, int dtype, int full, size_t n_frames, float *out, size_t out_capacity_floats) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!spec || !out) return fail(MELSPEC_ERR_INVALID_ARG, "spec/out is NULL");
    const size_t need = n_frames * static_cast<size_t>(c->n_mels);
    if (out_capacity_floats < need) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(c->dev.device));
    const size_t in_bytes = n_frames * static_cast<size_t>(full ? c->fft_size : c->fft_size / 2 + 1) * 2 * (dtype == MELSPEC_STFT_F64 ? sizeof(double) : sizeof(float));
    void *d_in = nullptr, *d_o = nullptr;
    HIP_TRY(hipMalloc(&d_in, in_bytes));
    hipError_t e = hipMalloc(&d_o, need * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(d_in); return fail_hip(e, "hipMalloc"); }
    int rc = MELSPEC_OK;
    if ((e = hipMemcpyAsync(d_in, spec, in_bytes, hipMemcpyHostToDevice, c->stream)) != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync");
    if (!rc) rc = melspec_mel_from_stft_device(c, d_in, dtype, full, n_frames, static_cast<float *>(d_o), c->stream);
    if (!rc && (e = hipMemcpyAsync(out, d_o, need * sizeof(float), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess && !rc) rc = fail_hip(e, "hipStreamSynchronize");
    (void)hipFree(d_in); (void)hipFree(d_o);
    return rc;
}

// ---- STFT export: Spectrogram::compute_all_cpu (src/stft.rs:89-115) / Spectrogram::add (src/stft.rs:48-86) -----------------
size_t melspec_stft_bins(const melspec_ctx *c, int full) {
    return !c ? 0 : static_cast<size_t>(full ? c->fft_size : c->fft_size / 2 + 1);
}

namespace {
int launch_stft(melspec_ctx *c, const BatchDesc &desc, int bins, int dtype, hipStream_t s) {
    if (desc.n_units == 0) return MELSPEC_OK;
    const int words = bins * 2 * (dtype == MELSPEC_STFT_F64 ? 2 : 1);
    if (c->fast) {
        static std::atomic<uint64_t> attr_done{0};
        if (!device_done(attr_done)) {
            int rc = allow_big_lds(&whisper400_stft_kernel<float>, "hipFuncSetAttribute(whisper400_stft_kernel<float>)");
            if (!rc) rc = allow_big_lds(&whisper400_stft_kernel<double>, "hipFuncSetAttribute(whisper400_stft_kernel<double>)");
            if (rc) return rc;
            mark_device_done(attr_done);
        }
        StftParams p{};
        p.b = desc;
        p.d_blob = static_cast<const uint32_t *>(c->d_blob64s.p);
        p.blob_words = PreciseBlob::kCount * 2;                    // the f64 tables only, not the mel section behind them
        p.hop = c->hop_size; p.bins = bins; p.words_per_frame = words;
        const size_t lds = static_cast<size_t>(p.blob_words) * 4 + static_cast<size_t>(kPreciseWaves) * PreciseLayout::slice_doubles() * sizeof(double);
        const uint64_t blocks = (desc.n_units + kPreciseWaves - 1) / kPreciseWaves;
        const unsigned grid = grid_for_xcd(blocks, c->dev.cus, 1);
        if (dtype == MELSPEC_STFT_F64) hipLaunchKernelGGL(whisper400_stft_kernel<double>, dim3(grid), dim3(kPreciseWaves * 64), lds, s, p);
        else hipLaunchKernelGGL(whisper400_stft_kernel<float>, dim3(grid), dim3(kPreciseWaves * 64), lds, s, p);
        HIP_TRY(hipGetLastError());
        return MELSPEC_OK;
    }
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
int stft_args(const melspec_ctx *c, int dtype) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    return MELSPEC_OK;
}
}  // namespace

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
    for (int i = 0; i < n && !rc; ++i) {
        if (bytes[i] == 0) continue;
        hipError_t e = hipSetDevice(src_devices[i]);
        if (e == hipSuccess && src_devices[i] != dst_device) {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, src_devices[i], dst_device);
            if (can) { const hipError_t pe = hipDeviceEnablePeerAccess(dst_device, 0); if (pe != hipSuccess) (void)hipGetLastError(); }   // already enabled is fine
        }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(static_cast<char *>(dst) + dst_offsets[i], dst_device, srcs[i], src_devices[i], bytes[i], streams[i]);
        if (e != hipSuccess) rc = fail_hip(e, "melspec_gather_peer");
    }
    for (int i = 0; i < n; ++i) {
        if (!streams[i]) continue;
        (void)hipSetDevice(src_devices[i]);
        const hipError_t e = hipStreamSynchronize(streams[i]);
        if (e != hipSuccess && !rc) rc = fail_hip(e, "melspec_gather_peer: hipStreamSynchronize");
        (void)hipStreamDestroy(streams[i]);
    }
    return rc;
}

}  // extern "C"

// ---- stand-alone mel helpers (mel_bank.hpp): SparseMelFilterbank, project_power, log_mel_spectrogram, norm_mel -------------------
struct melspec_bank {
    DeviceInfo dev;
    hipStream_t stream = nullptr;
    int n_mels = 0, fft_bins = 0, nnz = 0;
    DevBuf row_ptr, bin, w, wf, key, tmp_in, tmp_out;
    hipStream_t key_stream = nullptr;              // norm_mel's scratch word is used in stream order: a call on another stream first waits for this one
    bool key_used = false;
    std::vector<int> h_row_ptr, h_bin;             // the sparse rows on the host (weights_for_mel)
    std::vector<double> h_w;
    BankDesc desc() const {
        return BankDesc{static_cast<const int *>(row_ptr.p), static_cast<const int *>(bin.p), static_cast<const double *>(w.p),
                        static_cast<const float *>(wf.p), n_mels, fft_bins};
    }
};

namespace {
int bank_create(melspec_bank **out, int device, const std::vector<double> &dense, int n_mels, int fft_bins) {
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    melspec_bank *b = new (std::nothrow) melspec_bank();
    if (!b) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    b->dev = info; b->n_mels = n_mels; b->fft_bins = fft_bins;
    auto bail = [&](int code) { melspec_bank_destroy(b); return code; };
    if (hipSetDevice(info.device) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipSetDevice failed"));
    if (hipStreamCreate(&b->stream) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipStreamCreate failed"));
    // from_dense (src/mel.rs:48-71): per row the non-zero entries in ascending bin order
    std::vector<int> row_ptr(static_cast<size_t>(n_mels) + 1, 0), bins;
    std::vector<double> w;
    std::vector<float> wf;
    for (int m = 0; m < n_mels; ++m) {
        for (int k = 0; k < fft_bins; ++k) {
            const double v = dense[static_cast<size_t>(m) * fft_bins + k];
            if (v != 0.0) { bins.push_back(k); w.push_back(v); wf.push_back(static_cast<float>(v)); }
        }
        row_ptr[m + 1] = static_cast<int>(bins.size());
    }
    b->nnz = static_cast<int>(bins.size());
    b->h_row_ptr = row_ptr; b->h_bin = bins; b->h_w = w;
    if ((rc = upload(b->row_ptr, row_ptr)) || (rc = upload(b->bin, bins)) || (rc = upload(b->w, w)) || (rc = upload(b->wf, wf))) return bail(rc);
    if ((rc = b->key.ensure(16))) return bail(rc);
    *out = b;
    return MELSPEC_OK;
}

template <class T>
int norm_launch(const T *d_in, uint64_t n, T *d_out, unsigned long long *key, int cus, hipStream_t s) {
    const unsigned grid = static_cast<unsigned>(std::min<uint64_t>((n + 255) / 256, static_cast<uint64_t>(cus) * 16));
    hipLaunchKernelGGL(norm_init_kernel, dim3(1), dim3(1), 0, s, key);
    hipLaunchKernelGGL(norm_max_kernel<T>, dim3(grid), dim3(256), 0, s, d_in, n, key);
    hipLaunchKernelGGL(norm_map_kernel<T>, dim3(grid), dim3(256), 0, s, d_in, n, key, d_out);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}
}  // namespace

extern "C" {

int melspec_bank_from_dense(melspec_bank **out, int device, const double *filters, int n_mels, int fft_bins) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!filters || n_mels <= 0 || fft_bins <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "filters is NULL or a dimension is not positive");
    return bank_create(out, device, std::vector<double>(filters, filters + static_cast<size_t>(n_mels) * fft_bins), n_mels, fft_bins);
}

int melspec_bank_from_mel(melspec_bank **out, int device, double sample_rate, int n_fft, int n_mels, double f_min, double f_max, int htk, int norm) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!(sample_rate > 0.0) || n_fft < 2 || n_mels <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "sample_rate, n_fft and n_mels must be positive");
    return bank_create(out, device, mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max, htk != 0, norm != 0), n_mels, n_fft / 2 + 1);
}

void melspec_bank_destroy(melspec_bank *b) {
    if (!b) return;
    if (b->dev.device >= 0) (void)hipSetDevice(b->dev.device);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    b->row_ptr.release(); b->bin.release(); b->w.release(); b->wf.release(); b->key.release(); b->tmp_in.release(); b->tmp_out.release();
    delete b;
}

int melspec_bank_n_mels(const melspec_bank *b) { return b ? b->n_mels : 0; }
int melspec_bank_fft_bins(const melspec_bank *b) { return b ? b->fft_bins : 0; }
int melspec_bank_non_zero_weights(const melspec_bank *b) { return b ? b->nnz : 0; }

int melspec_bank_weights_for_mel(const melspec_bank *b, int mel_idx, int *bins, double *weights, int capacity) {
    if (!b || mel_idx < 0 || mel_idx >= b->n_mels) return -1;
    const int lo = b->h_row_ptr[mel_idx], n = b->h_row_ptr[mel_idx + 1] - lo;
    for (int i = 0; i < n && i < capacity; ++i) {
        if (bins) bins[i] = b->h_bin[lo + i];
        if (weights) weights[i] = b->h_w[lo + i];
    }
    return n;
}

int melspec_bank_project_power_device(melspec_bank *b, const void *d_power, int dtype, uint64_t n_frames, void *d_out, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!d_power || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    const uint64_t items = n_frames * static_cast<uint64_t>(b->n_mels);
    if ((items + 255) / 256 > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "too many frames for one call");
    const dim3 grid(static_cast<unsigned>((items + 255) / 256));
    if (dtype == MELSPEC_STFT_F64)
        hipLaunchKernelGGL(bank_project_power_kernel<double>, grid, dim3(256), 0, s, b->desc(), static_cast<const double *>(d_power), static_cast<double *>(d_out), n_frames);
    else
        hipLaunchKernelGGL(bank_project_power_kernel<float>, grid, dim3(256), 0, s, b->desc(), static_cast<const float *>(d_power), static_cast<float *>(d_out), n_frames);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_bank_log_mel_device(melspec_bank *b, const void *d_stft, int dtype, int n_fft, uint64_t n_frames, double *d_out, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_fft < 2 || n_fft < b->fft_bins) return fail(MELSPEC_ERR_INVALID_ARG, "n_fft must be at least the bank's fft_bins");
    if (n_frames == 0) return MELSPEC_OK;
    if (!d_stft || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    const uint64_t items = n_frames * static_cast<uint64_t>(b->n_mels);
    if ((items + 255) / 256 > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "too many frames for one call");
    const dim3 grid(static_cast<unsigned>((items + 255) / 256));
    if (dtype == MELSPEC_STFT_F64)
        hipLaunchKernelGGL(bank_log_mel_kernel<double>, grid, dim3(256), 0, s, b->desc(), static_cast<const double *>(d_stft), n_fft, d_out, n_frames);
    else
        hipLaunchKernelGGL(bank_log_mel_kernel<float>, grid, dim3(256), 0, s, b->desc(), static_cast<const float *>(d_stft), n_fft, d_out, n_frames);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_bank_norm_mel_device(melspec_bank *b, const void *d_in, int dtype, uint64_t n_values, void *d_out, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_values == 0) return MELSPEC_OK;
    if (!d_in || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    unsigned long long *key = static_cast<unsigned long long *>(b->key.p);
    // the three launches of a call (init, max, map) share ONE scratch word per bank: two calls on different streams would race on it
    //  -- like FixState::last_stream, a change of stream waits for the previous one
    if (b->key_used && b->key_stream != s) HIP_TRY(hipStreamSynchronize(b->key_stream));
    b->key_used = true; b->key_stream = s;
    return dtype == MELSPEC_STFT_F64 ? norm_launch<double>(static_cast<const double *>(d_in), n_values, static_cast<double *>(d_out), key, b->dev.cus, s)
                                     : norm_launch<float>(static_cast<const float *>(d_in), n_values, static_cast<float *>(d_out), key, b->dev.cus, s);
}

// host forms: staged through the bank's own buffers, synchronous
static int bank_host_call(melspec_bank *b, const void *in, size_t in_bytes, void *out, size_t out_bytes, int (*run)(melspec_bank *, const void *, void *, void *), void *ctx) {
    HIP_TRY(hipSetDevice(b->dev.device));
    int rc;
    if ((rc = b->tmp_in.ensure(in_bytes + 16)) || (rc = b->tmp_out.ensure(out_bytes + 16))) return rc;
    HIP_TRY(hipMemcpyAsync(b->tmp_in.p, in, in_bytes, hipMemcpyHostToDevice, b->stream));
    if ((rc = run(b, b->tmp_in.p, b->tmp_out.p, ctx))) { (void)hipStreamSynchronize(b->stream); return rc; }
    HIP_TRY(hipMemcpyAsync(out, b->tmp_out.p, out_bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return MELSPEC_OK;
}

int melspec_bank_project_power_host(melspec_bank *b, const void *power, int dtype, size_t n_frames, void *out) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!power || !out) return fail(MELSPEC_ERR_INVALID_ARG, "pointer is NULL");
    const size_t el = dtype == MELSPEC_STFT_F64 ? 8 : 4;
    struct A { int dtype; uint64_t n; } a{dtype, n_frames};
    return bank_host_call(b, power, n_frames * b->fft_bins * el, out, n_frames * b->n_mels * el,
                          [](melspec_bank *bb, const void *i, void *o, void *c) { auto *x = static_cast<A *>(c); return melspec_bank_project_power_device(bb, i, x->dtype, x->n, o, bb->stream); }, &a);
}

int melspec_bank_log_mel_host(melspec_bank *b, const void *stft, int dtype, int n_fft, size_t n_frames, double *out) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_fft < 2) return fail(MELSPEC_ERR_INVALID_ARG, "n_fft must be >= 2");
    if (n_frames == 0) return MELSPEC_OK;
    if (!stft || !out) return fail(MELSPEC_ERR_INVALID_ARG, "pointer is NULL");
    const size_t el = dtype == MELSPEC_STFT_F64 ? 8 : 4;
    struct A { int dtype, n_fft; uint64_t n; } a{dtype, n_fft, n_frames};
    return bank_host_call(b, stft, n_frames * n_fft * 2 * el, out, n_frames * b->n_mels * 8,
                          [](melspec_bank *bb, const void *i, void *o, void *c) { auto *x = static_cast<A *>(c); return melspec_bank_log_mel_device(bb, i, x->dtype, x->n_fft, x->n, static_cast<double *>(o), bb->stream); }, &a);
}

int melspec_bank_norm_mel_host(melspec_bank *b, const void *in, int dtype, size_t n_values, void *out) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_values == 0) return MELSPEC_OK;
    if (!in || !out) return fail(MELSPEC_ERR_INVALID_ARG, "pointer is NULL");
    const size_t el = dtype == MELSPEC_STFT_F64 ? 8 : 4;
    struct A { int dtype; uint64_t n; } a{dtype, n_values};
    return bank_host_call(b, in, n_values * el, out, n_values * el,
                          [](melspec_bank *bb, const void *i, void *o, void *c) { auto *x = static_cast<A *>(c); return melspec_bank_norm_mel_device(bb, i, x->dtype, x->n, o, bb->stream); }, &a);
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
    if ((rc = allow_big_lds(&generic_frame_kernel<kGenericNT>, "hipFuncSetAttribute(generic_frame_kernel)"))) return bail(rc);
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

static int fbank_launch(melspec_fbank *fb, const BatchPlan &pl, uint32_t n_clips, uint64_t fpc, hipStream_t s);

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

// kernels of one batch (uniform or ragged plan): fused 512-point kernel or the generic one, then CMN per clip
static int fbank_launch(melspec_fbank *fb, const BatchPlan &pl, uint32_t n_clips, uint64_t fpc /* frames of the longest clip (LDS budget of the CMN) */, hipStream_t s) {
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
            (fb->cfg.apply_cmn && fb->waves == 8 && pl.desc.d_unit_prefix == nullptr && nm % 4 == 0 && nm <= 89 &&
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
    bool by_clip = fused && fb->cfg.apply_cmn && fb->waves == 8 && nm % 4 == 0 && nm <= 89 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 &&
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
// Streaming: a bank of live streams with device-side overlap-save state
// (Spectrogram::add src/stft.rs:48-86 driven by RingBuffer::maybe_mel src/rb.rs:86-121)
// ------------------------------------------------------------------------------------
struct melspec_stream {
    melspec_ctx *ctx = nullptr;          // geometry, tables, kernels; not owned
    StreamGeom geom{};
    StreamBook book;                     // pending / idx per stream (host side of the state)
    DevBuf state, staging, out;
    RaggedScratch ring;                  // per-push entry tables
    // the detector stage (melspec_stream_enable_vad): VoiceActivityDetector state per stream, in HBM
    bool vad_on = false;
    melspec_vad_settings vad{};
    DevBuf vad_state, vad_prev, vad_acts;
    std::vector<uint64_t> vad_count;     // host copy of StreamVadState::count (VoiceActivityDetector::frame_index)
    // Steady state of a live bank: the same streams pushing the same number of samples from the same pending count, every stream past
    // its first window.  Such a push has the entry table and the ragged plan of the previous one -- both are still on the device --
    // so neither is built or uploaded again (4096 streams x 1 hop: 0.089 -> 0.05 ms per push).
    struct PushCache {
        bool valid = false;
        uint32_t n = 0;
        int fpu = 0;
        const void *d_out = nullptr;
        std::vector<uint32_t> ids, lens, pend;
        std::vector<uint64_t> out_off;       // the caller's row offsets (empty: packed)
        StreamPlan pl;
        const StreamEntry *d_e = nullptr;    // the entries in the ring slot of the push that filled the cache
        BatchPlan plan;
    } cache;
    RaggedScratch plan_ring;             // the ragged plans of the pushes (the context's own ring serves its other callers)
};

namespace {
int stream_plan(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, bool flush, StreamPlan &pl) {
    const char *err = nullptr;
    const int rc = stream_plan_push(st->geom, st->book, ids, lens, n, flush, pl, &err);
    if (rc == 1) return fail(MELSPEC_ERR_INVALID_ARG, err);
    if (rc == 2) return fail(MELSPEC_ERR_CAPACITY, err);
    return MELSPEC_OK;
}
void stream_commit(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, bool flush) {
    stream_commit_push(st->geom, st->book, ids, lens, n, flush);
}

// what a push emits per frame: the mel row (Spectrogram::add + MelSpectrogram::add) or the spectrum (Spectrogram::add alone)
struct StreamEmit {
    bool stft = false;
    int dtype = MELSPEC_STFT_F32, full = 0;
};

// scatter (optional) -> frames -> carry update, all on one stream
// does this push repeat the cached one?  (ids / lens / pending before the push; every stream was past its first window when the cache
// was filled and idx only grows, resets invalidate)
bool stream_cache_hit(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, const void *d_out, const uint64_t *h_out_off) {
    const melspec_stream::PushCache &k = st->cache;
    if (!k.valid || k.n != n || k.d_out != d_out || k.out_off.empty() != (h_out_off == nullptr)) return false;
    if (k.fpu != ctx_frames_per_unit(st->ctx)) return false;            // AUTO changed its regime: another unit size
    for (uint32_t i = 0; i < n; ++i)
        if (ids[i] != k.ids[i] || lens[i] != k.lens[i] || st->book.pending[ids[i]] != k.pend[i]) return false;
    return h_out_off == nullptr || std::memcmp(h_out_off, k.out_off.data(), static_cast<size_t>(n) * sizeof(uint64_t)) == 0;
}

// reuse: `pl` is st->cache.pl and the device still holds its entries and plan (stream_cache_hit); ids / lens: the push's arguments,
// for filling the cache (NULL: do not, e.g. a flush)
int stream_run(melspec_stream *st, const StreamPlan &pl, uint32_t n, const float *d_src, void *d_out, const uint64_t *h_out_off,
               hipStream_t s, const StreamEmit &emit = StreamEmit(), melspec_vad_activity *d_acts = nullptr, bool reuse = false,
               const uint32_t *ids = nullptr, const uint32_t *lens = nullptr) {
    melspec_ctx *c = st->ctx;
    HIP_TRY(hipSetDevice(c->dev.device));
    if (pl.total_frames && !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "d_out is NULL");      // before anything is queued
    if (st->vad_on && emit.stft) return fail(MELSPEC_ERR_UNSUPPORTED, "the detector stage is on: it needs the mel rows of every push");
    if (st->vad_on && pl.total_frames && !d_acts) {                                             // records nobody asked for: internal buffer
        const int rc0 = st->vad_acts.ensure(pl.total_frames * sizeof(melspec_vad_activity));
        if (rc0) return rc0;
        d_acts = static_cast<melspec_vad_activity *>(st->vad_acts.p);
    }
    int rc = MELSPEC_OK;
    // the entries travel like a ragged plan: pinned slot, copy kernel on the launch stream (no SDMA queue hand-over)
    struct SlotGuard { RaggedSlot *sl; hipStream_t s; ~SlotGuard() { plan_ragged_done(sl, s); } } slot_guard{nullptr, s};
    const StreamEntry *d_e = st->cache.d_e;
    if (!reuse) {
        st->cache.valid = false;                     // whatever happens below, the slot the cache points into may be the next one taken
        RaggedSlot &sl = st->ring.slot[st->ring.next++ % RaggedScratch::kSlots];
        if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        if (sl.pending) { HIP_TRY(hipEventSynchronize(sl.ev)); sl.pending = false; }
        const size_t ebytes = (static_cast<size_t>(n) * sizeof(StreamEntry) + 15) & ~static_cast<size_t>(15);
        rc = sl.ensure_host(ebytes);
        if (rc) return rc;
        if ((rc = sl.dev.ensure(ebytes))) return rc;
        std::memcpy(sl.host, pl.entries.data(), static_cast<size_t>(n) * sizeof(StreamEntry));
        if (h_out_off)                                   // caller-placed rows: the detector stage reads them where they are
            for (uint32_t i = 0; i < n; ++i) static_cast<StreamEntry *>(sl.host)[i].out_off = h_out_off[i];
        // from here on the slot is in use by queued work: every exit records its event (the next user of the slot waits for it)
        slot_guard.sl = &sl;
        const size_t n16 = ebytes / 16;
        const unsigned blocks = static_cast<unsigned>((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
        hipLaunchKernelGGL(plan_upload_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, static_cast<const uint4 *>(sl.host),
                           static_cast<uint4 *>(sl.dev.p), n16);
        HIP_TRY(hipGetLastError());
        d_e = static_cast<const StreamEntry *>(sl.dev.p);
    }
    float *state = static_cast<float *>(st->state.p);
    bool any_fill = d_src != nullptr;
    for (uint32_t i = 0; i < n && !any_fill; ++i) any_fill = pl.entries[i].zero_fill != 0;
    if (any_fill) {
        hipLaunchKernelGGL(stream_scatter_kernel, dim3(n), dim3(256), 0, s, state, st->geom.stride, st->geom.in_off, d_e, d_src);
        HIP_TRY(hipGetLastError());
    }
    if (pl.total_frames && emit.stft) {
        std::vector<uint64_t> oo(n);            // complex elements, entries back to back
        uint64_t cur = 0;
        const uint64_t bins = melspec_stft_bins(c, emit.full);
        for (uint32_t i = 0; i < n; ++i) { oo[i] = cur; cur += pl.frames[i] * bins; }
        rc = melspec_stft_ragged_device(c, state, pl.off.data(), pl.len.data(), n, d_out, h_out_off ? h_out_off : oo.data(), emit.dtype, emit.full, s);
        if (rc) return rc;
    } else if (pl.total_frames) {
        if (reuse) {
            rc = launch_ctx(c, st->cache.plan.desc, s);
        } else {
            // melspec_compute_ragged_device with the plan kept: frames per entry are the plan's, the ring is the bank's own
            std::vector<uint64_t> fr(n);
            for (uint32_t i = 0; i < n; ++i) fr[i] = pl.frames[i];
            RaggedSlot *pslot = nullptr;
            const int fpu = ctx_frames_per_unit(c);
            rc = plan_ragged(st->plan_ring, s, state, static_cast<float *>(d_out), pl.off.data(), fr, h_out_off ? h_out_off : pl.out_off.data(), n,
                             c->n_mels, fpu, st->cache.plan, pslot);
            if (!rc) rc = launch_ctx(c, st->cache.plan.desc, s);
            plan_ragged_done(pslot, s);
            // a push that can come again: every stream past its first window (no skipped hops), not a flush
            bool steady = !rc && ids != nullptr && lens != nullptr;
            for (uint32_t i = 0; i < n && steady; ++i) steady = st->book.idx[ids[i]] >= st->geom.n_fft;
            if (steady) {
                melspec_stream::PushCache &k = st->cache;
                k.n = n; k.fpu = fpu; k.d_out = d_out; k.d_e = d_e;
                k.ids.assign(ids, ids + n); k.lens.assign(lens, lens + n);
                k.pend.resize(n);
                for (uint32_t i = 0; i < n; ++i) k.pend[i] = st->book.pending[ids[i]];
                if (h_out_off) k.out_off.assign(h_out_off, h_out_off + n); else k.out_off.clear();
                k.pl = pl;
                k.valid = true;
            }
        }
        if (rc) return rc;
        if (st->vad_on) {
            StreamVadParams vp{};
            vp.entries = d_e; vp.rows = static_cast<const float *>(d_out);
            vp.state = static_cast<StreamVadState *>(st->vad_state.p); vp.prev = static_cast<float *>(st->vad_prev.p);
            vp.acts = reinterpret_cast<VadActivity *>(d_acts);
            vp.n_mels = st->geom.n_mels; vp.min_mel = st->vad.min_mel; vp.min_y = st->vad.min_y; vp.min_x = st->vad.min_x;
            vp.thr = st->vad.min_energy * st->vad.min_energy;
            uint32_t most = 0;
            for (uint32_t i = 0; i < n; ++i) most = std::max(most, pl.frames[i]);
            hipLaunchKernelGGL(stream_vad_kernel, dim3(n), dim3(most <= 1 ? 64 : most <= 2 ? 128 : 256), 0, s, vp);
            HIP_TRY(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(stream_carry_kernel, dim3(n), dim3(256), 0, s, state, st->geom.stride, st->geom.in_off, d_e);
    HIP_TRY(hipGetLastError());
    // the contract of the push calls: the launches have completed on return (a device producer may refill its slot at once)
    HIP_TRY(hipStreamSynchronize(s));
    return MELSPEC_OK;
}
}  // namespace

extern "C" {

int melspec_stream_create(melspec_stream **out, melspec_ctx *ctx, uint32_t n_streams, uint32_t max_chunk) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!ctx) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (n_streams == 0 || max_chunk == 0) return fail(MELSPEC_ERR_INVALID_ARG, "n_streams and max_chunk must be > 0");
    if (ctx->hop_size > ctx->fft_size) return fail(MELSPEC_ERR_UNSUPPORTED, "streaming needs hop_size <= fft_size");
    melspec_stream *st = new (std::nothrow) melspec_stream();
    if (!st) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    st->ctx = ctx;
    st->geom = stream_geometry(static_cast<uint32_t>(ctx->fft_size), static_cast<uint32_t>(ctx->hop_size), static_cast<uint32_t>(ctx->n_mels),
                               n_streams, max_chunk);
    st->book.reset(n_streams);
    if (hipSetDevice(ctx->dev.device) != hipSuccess) { delete st; return fail(MELSPEC_ERR_UNAVAILABLE, "hipSetDevice failed"); }
    const size_t bytes = static_cast<size_t>(n_streams) * st->geom.stride * sizeof(float) + 64;
    int rc = st->state.ensure(bytes);
    if (rc) { delete st; return rc; }
    if (hipMemset(st->state.p, 0, bytes) != hipSuccess) { st->state.release(); delete st; return fail(MELSPEC_ERR_INTERNAL, "hipMemset failed"); }
    *out = st;
    return MELSPEC_OK;
}

void melspec_stream_destroy(melspec_stream *st) {
    if (!st) return;
    if (st->ctx) { (void)hipSetDevice(st->ctx->dev.device); (void)hipStreamSynchronize(st->ctx->stream); }
    st->state.release(); st->ring.release(); st->staging.release(); st->out.release();
    st->vad_state.release(); st->vad_prev.release(); st->vad_acts.release(); st->plan_ring.release();
    delete st;
}

int melspec_stream_reset(melspec_stream *st, const uint32_t *ids, uint32_t n) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    st->cache.valid = false;
    HIP_TRY(hipSetDevice(st->ctx->dev.device));
    if (!ids) {
        HIP_TRY(hipMemsetAsync(st->state.p, 0, static_cast<size_t>(st->geom.n_streams) * st->geom.stride * sizeof(float), st->ctx->stream));
        st->book.reset(st->geom.n_streams);
        if (st->vad_on) {
            HIP_TRY(hipMemsetAsync(st->vad_state.p, 0, static_cast<size_t>(st->geom.n_streams) * sizeof(StreamVadState), st->ctx->stream));
            std::fill(st->vad_count.begin(), st->vad_count.end(), 0ull);
        }
    } else {
        for (uint32_t i = 0; i < n; ++i) {
            if (ids[i] >= st->geom.n_streams) return fail(MELSPEC_ERR_INVALID_ARG, "stream id out of range");
            HIP_TRY(hipMemsetAsync(static_cast<float *>(st->state.p) + ids[i] * st->geom.stride, 0, st->geom.in_off * sizeof(float), st->ctx->stream));
            st->book.pending[ids[i]] = 0; st->book.idx[ids[i]] = 0;
            if (st->vad_on) {
                HIP_TRY(hipMemsetAsync(static_cast<StreamVadState *>(st->vad_state.p) + ids[i], 0, sizeof(StreamVadState), st->ctx->stream));
                st->vad_count[ids[i]] = 0;
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(st->ctx->stream));
    return MELSPEC_OK;
}

size_t melspec_stream_frames_after(const melspec_stream *st, uint32_t id, uint32_t n_new) {
    if (!st || id >= st->geom.n_streams) return 0;
    return stream_frames_after(st->geom, st->book, id, n_new);
}

float *melspec_stream_input_ptr(melspec_stream *st, uint32_t id) {
    if (!st || id >= st->geom.n_streams) return nullptr;
    return static_cast<float *>(st->state.p) + id * st->geom.stride + st->geom.in_off;
}

static_assert(sizeof(melspec_vad_activity) == 8 && sizeof(VadActivity) == 8, "the activity record is 8 bytes on both sides of the ABI");
static void stream_vad_commit(melspec_stream *st, const uint32_t *ids, const StreamPlan &pl, uint32_t n) {
    if (!st->vad_on) return;
    for (uint32_t i = 0; i < n; ++i) st->vad_count[ids[i]] += pl.frames[i];
}

static int stream_push_device_impl(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                                   const uint64_t *h_out_offsets, uint32_t *h_frames, melspec_vad_activity *d_acts, bool want_acts,
                                   void *stream) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    if (want_acts && !st->vad_on) return fail(MELSPEC_ERR_INVALID_ARG, "the detector stage is off (melspec_stream_enable_vad)");
    if (n == 0) return MELSPEC_OK;
    if (!ids || !lens) return fail(MELSPEC_ERR_INVALID_ARG, "ids/lens is NULL");
    StreamPlan fresh;
    const bool reuse = stream_cache_hit(st, ids, lens, n, d_out, h_out_offsets);
    int rc = reuse ? MELSPEC_OK : stream_plan(st, ids, lens, n, false, fresh);
    if (rc) return rc;
    const StreamPlan &pl = reuse ? st->cache.pl : fresh;
    if (want_acts && pl.total_frames && !d_acts) return fail(MELSPEC_ERR_INVALID_ARG, "d_acts is NULL");
    rc = stream_run(st, pl, n, nullptr, d_out, h_out_offsets, stream ? static_cast<hipStream_t>(stream) : st->ctx->stream, StreamEmit(), d_acts,
                    reuse, ids, lens);
    if (rc) return rc;
    stream_commit(st, ids, lens, n, false);
    stream_vad_commit(st, ids, pl, n);
    if (h_frames) std::memcpy(h_frames, pl.frames.data(), static_cast<size_t>(n) * sizeof(uint32_t));
    return MELSPEC_OK;
}

int melspec_stream_push_device(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                               const uint64_t *h_out_offsets, uint32_t *h_frames, void *stream) {
    return stream_push_device_impl(st, ids, lens, n, d_out, h_out_offsets, h_frames, nullptr, false, stream);
}

int melspec_stream_push_device_vad(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                                   const uint64_t *h_out_offsets, uint32_t *h_frames, melspec_vad_activity *d_acts, void *stream) {
    return stream_push_device_impl(st, ids, lens, n, d_out, h_out_offsets, h_frames, d_acts, true, stream);
}

int melspec_stream_enable_vad(melspec_stream *st, const melspec_vad_settings *settings) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    HIP_TRY(hipSetDevice(st->ctx->dev.device));
    HIP_TRY(hipStreamSynchronize(st->ctx->stream));
    if (!settings) { st->vad_on = false; return MELSPEC_OK; }
    if (settings->min_x > kStreamVadMaxX) return fail(MELSPEC_ERR_UNSUPPORTED, "min_x above 66: the column history of a stream is 64 bits");
    if (settings->min_x < 0 || settings->min_y < 0 || settings->min_mel < 0) return fail(MELSPEC_ERR_INVALID_ARG, "negative detection setting");
    const size_t ns = st->geom.n_streams;
    int rc = st->vad_state.ensure(ns * sizeof(StreamVadState));
    if (rc) return rc;
    if ((rc = st->vad_prev.ensure(ns * 2 * st->geom.n_mels * sizeof(float) + 16))) return rc;
    HIP_TRY(hipMemset(st->vad_state.p, 0, ns * sizeof(StreamVadState)));
    HIP_TRY(hipMemset(st->vad_prev.p, 0, ns * 2 * st->geom.n_mels * sizeof(float)));
    st->vad_count.assign(ns, 0ull);
    st->vad = *settings;
    st->vad_on = true;
    return MELSPEC_OK;
}

uint64_t melspec_stream_vad_frames(const melspec_stream *st, uint32_t id) {
    if (!st || !st->vad_on || id >= st->geom.n_streams) return 0;
    return st->vad_count[id];
}

static int stream_push_host_impl(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                 bool flush, void *out, size_t out_capacity, uint32_t *h_frames, const StreamEmit &emit = StreamEmit(),
                                 melspec_vad_activity *acts = nullptr, size_t acts_capacity = 0, bool want_acts = false) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    if (want_acts && !st->vad_on) return fail(MELSPEC_ERR_INVALID_ARG, "the detector stage is off (melspec_stream_enable_vad)");
    if (n == 0) return MELSPEC_OK;
    if (!ids || (!flush && !lens)) return fail(MELSPEC_ERR_INVALID_ARG, "ids/lens is NULL");
    StreamPlan fresh;
    const bool reuse = !flush && !emit.stft && stream_cache_hit(st, ids, lens, n, st->out.p, nullptr);
    int rc = reuse ? MELSPEC_OK : stream_plan(st, ids, lens, n, flush, fresh);
    if (rc) return rc;
    const StreamPlan &pl = reuse ? st->cache.pl : fresh;
    // elements the caller receives: floats (mel rows) or complex values (spectra)
    const uint64_t need = pl.total_frames * (emit.stft ? melspec_stft_bins(st->ctx, emit.full) : static_cast<uint64_t>(st->ctx->n_mels));
    const size_t esz = emit.stft ? (emit.dtype == MELSPEC_STFT_F64 ? 16 : 8) : sizeof(float);
    if (need > out_capacity) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    if (need && !out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    if (want_acts && pl.total_frames > acts_capacity) return fail(MELSPEC_ERR_CAPACITY, "activity buffer too small");
    if (want_acts && pl.total_frames && !acts) return fail(MELSPEC_ERR_INVALID_ARG, "acts is NULL");
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) total += pl.entries[i].len;
    if (total && !samples) return fail(MELSPEC_ERR_INVALID_ARG, "samples is NULL");
    hipStream_t s = st->ctx->stream;
    HIP_TRY(hipSetDevice(st->ctx->dev.device));
    if ((rc = st->staging.ensure(total * sizeof(float) + 16))) return rc;
    if ((rc = st->out.ensure(need * esz + 16))) return rc;
    if (total) HIP_TRY(hipMemcpyAsync(st->staging.p, samples, total * sizeof(float), hipMemcpyHostToDevice, s));
    if (st->vad_on && pl.total_frames && (rc = st->vad_acts.ensure(pl.total_frames * sizeof(melspec_vad_activity)))) return rc;
    // (st->out may have been re-allocated by the ensure above: the cache is keyed on its address, a stale one simply misses next time)
    rc = stream_run(st, pl, n, total ? static_cast<const float *>(st->staging.p) : nullptr, st->out.p, nullptr, s, emit,
                    static_cast<melspec_vad_activity *>(st->vad_acts.p), reuse && st->cache.d_out == st->out.p, flush || emit.stft ? nullptr : ids,
                    flush || emit.stft ? nullptr : lens);
    if (rc) return rc;
    if (need) {
        HIP_TRY(hipMemcpyAsync(out, st->out.p, need * esz, hipMemcpyDeviceToHost, s));
        if (want_acts) HIP_TRY(hipMemcpyAsync(acts, st->vad_acts.p, pl.total_frames * sizeof(melspec_vad_activity), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    stream_commit(st, ids, lens, n, flush);
    stream_vad_commit(st, ids, pl, n);
    if (h_frames) std::memcpy(h_frames, pl.frames.data(), static_cast<size_t>(n) * sizeof(uint32_t));
    return MELSPEC_OK;
}

int melspec_stream_push_host(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                             float *out, size_t out_capacity_floats, uint32_t *h_frames) {
    return stream_push_host_impl(st, ids, samples, lens, n, false, out, out_capacity_floats, h_frames);
}

int melspec_stream_push_host_stft(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                  void *out, size_t out_capacity_complex, uint32_t *h_frames, int dtype, int full) {
    if (st) { const int rc = stft_args(st->ctx, dtype); if (rc) return rc; }
    StreamEmit e; e.stft = true; e.dtype = dtype; e.full = full;
    return stream_push_host_impl(st, ids, samples, lens, n, false, out, out_capacity_complex, h_frames, e);
}

int melspec_stream_flush_host(melspec_stream *st, const uint32_t *ids, uint32_t n, float *out, size_t out_capacity_floats,
                              uint32_t *h_frames) {
    return stream_push_host_impl(st, ids, nullptr, nullptr, n, true, out, out_capacity_floats, h_frames);
}

int melspec_stream_push_host_vad(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                 float *out, size_t out_capacity_floats, uint32_t *h_frames, melspec_vad_activity *acts, size_t acts_capacity) {
    return stream_push_host_impl(st, ids, samples, lens, n, false, out, out_capacity_floats, h_frames, StreamEmit(), acts, acts_capacity, true);
}

int melspec_stream_flush_host_vad(melspec_stream *st, const uint32_t *ids, uint32_t n, float *out, size_t out_capacity_floats,
                                  uint32_t *h_frames, melspec_vad_activity *acts, size_t acts_capacity) {
    return stream_push_host_impl(st, ids, nullptr, nullptr, n, true, out, out_capacity_floats, h_frames, StreamEmit(), acts, acts_capacity, true);
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// 8-bit quantisation + TGA container (src/quant.rs)
// ------------------------------------------------------------------------------------
struct melspec_tga {
    DeviceInfo dev;
    hipStream_t stream = nullptr;
    DevBuf keys, ranges, h2d, d2h, unit_ext;       // unit_ext: the mel kernel's per-unit extremes (melspec_tga_encode_pcm_uniform_device)
    // the min/max keys are one scratch buffer per handle, used in stream order: a call on another stream first waits for
    // the stream that used it last
    hipStream_t keys_stream = nullptr;
    bool keys_used = false;
};

namespace {
size_t round_up4(size_t v) { return (v + 3) & ~static_cast<size_t>(3); }

struct TgaLayout { uint32_t chunks; uint32_t chunk_w; size_t chunk_stride; size_t last_bytes; };
TgaLayout tga_layout(uint32_t rows, uint64_t width) {
    TgaLayout l{};
    if (width == 0) return l;
    l.chunks = static_cast<uint32_t>((width + kTgaMaxWidth - 1) / kTgaMaxWidth);
    l.chunk_w = static_cast<uint32_t>(width < kTgaMaxWidth ? width : kTgaMaxWidth);
    l.chunk_stride = round_up4(kTgaHeader + static_cast<size_t>(rows) * l.chunk_w);
    l.last_bytes = kTgaHeader + static_cast<size_t>(rows) * (width - static_cast<uint64_t>(l.chunks - 1) * l.chunk_w);
    return l;
}

// fills the descriptor and the launch shape shared by encode and decode
int quant_plan(melspec_tga *q, QuantDesc &d, const void *img, size_t image_stride, uint32_t rows, uint64_t width, uint32_t n_images,
               const void *blob, size_t blob_stride, bool header, uint32_t &items, uint32_t &bpi_px, uint32_t &bpi_dw) {
    if (width > 0xffffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "image wider than 2^32-1 columns");
    d = QuantDesc{};
    d.rows = rows; d.width = static_cast<uint32_t>(width); d.n_images = n_images;
    d.img_stride = image_stride; d.blob_stride = blob_stride;
    d.header = header ? kTgaHeader : 0;
    if (header) {
        const TgaLayout l = tga_layout(rows, width);
        d.chunks = l.chunks; d.chunk_w = l.chunk_w; d.chunk_stride = l.chunk_stride;
        if (blob_stride % 4 || blob_stride < l.chunk_stride * l.chunks)
            return fail(MELSPEC_ERR_INVALID_ARG, "blob_stride must be a multiple of 4 and >= n_chunks * chunk_stride (melspec_tga_layout)");
    } else {
        d.chunks = 1; d.chunk_w = d.width; d.chunk_stride = 0;
        if (blob_stride % 4) return fail(MELSPEC_ERR_INVALID_ARG, "blob_stride must be a multiple of 4");
    }
    if (reinterpret_cast<uintptr_t>(blob) % 4) return fail(MELSPEC_ERR_INVALID_ARG, "blob pointer must be 4-byte aligned");
    if (reinterpret_cast<uintptr_t>(img) % 4) return fail(MELSPEC_ERR_INVALID_ARG, "image pointer must be 4-byte aligned");
    d.vec = d.chunks == 1 && reinterpret_cast<uintptr_t>(img) % 16 == 0 && (n_images == 1 || image_stride % 4 == 0);
    const uint64_t items64 = static_cast<uint64_t>(n_images) * d.chunks;
    const uint64_t npx = static_cast<uint64_t>(rows) * d.chunk_w;
    const uint64_t bpx = (npx + kQuantPxPerBlock - 1) / kQuantPxPerBlock;
    const uint64_t bdw = ((d.header + npx + 3) / 4 + kQuantDwPerBlock - 1) / kQuantDwPerBlock;
    if (items64 * bpx > 0x7fffffffull || items64 * bdw > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "batch too large for one launch");
    items = static_cast<uint32_t>(items64); bpi_px = static_cast<uint32_t>(bpx); bpi_dw = static_cast<uint32_t>(bdw);
    int rc = q->keys.ensure(items64 * 2 * sizeof(uint32_t) + 16);
    if (rc) return rc;
    d.keys = static_cast<uint32_t *>(q->keys.p);
    return MELSPEC_OK;
}

int quant_encode(melspec_tga *q, const float *d_img, size_t image_stride, uint32_t rows, uint64_t width, uint32_t n_images,
                 uint8_t *d_blob, size_t blob_stride, bool header, float *d_ranges, hipStream_t stream) {
    if (n_images == 0 || width == 0 || rows == 0) return MELSPEC_OK;
    if (!d_img || !d_blob) return fail(MELSPEC_ERR_INVALID_ARG, "image/blob pointer is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    QuantDesc d;
    uint32_t items, bpx, bdw;
    int rc = quant_plan(q, d, d_img, image_stride, rows, width, n_images, d_blob, blob_stride, header, items, bpx, bdw);
    if (rc) return rc;
    d.img = d_img; d.blob = d_blob; d.ranges = d_ranges;
    if (q->keys_used && q->keys_stream != stream) HIP_TRY(hipStreamSynchronize(q->keys_stream));
    q->keys_used = true; q->keys_stream = stream;
    hipLaunchKernelGGL(quant_init_keys_kernel, dim3((items + 255) / 256), dim3(256), 0, stream, d.keys, items);
    hipLaunchKernelGGL(quant_minmax_kernel, dim3(items * bpx), dim3(kQuantThreads), 0, stream, d, bpx);
    hipLaunchKernelGGL(quant_encode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, stream, d, bdw);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int quant_decode(melspec_tga *q, const uint8_t *d_blob, size_t blob_stride, uint32_t rows, uint64_t width, uint32_t n_images,
                 float *d_img, size_t image_stride, bool header, const float *d_ranges, hipStream_t stream) {
    if (n_images == 0 || width == 0 || rows == 0) return MELSPEC_OK;
    if (!d_img || !d_blob) return fail(MELSPEC_ERR_INVALID_ARG, "image/blob pointer is NULL");
    if (!header && !d_ranges) return fail(MELSPEC_ERR_INVALID_ARG, "range pointer is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    QuantDesc d;
    uint32_t items, bpx, bdw;
    int rc = quant_plan(q, d, d_img, image_stride, rows, width, n_images, d_blob, blob_stride, header, items, bpx, bdw);
    if (rc) return rc;
    d.img_out = d_img; d.blob = const_cast<uint8_t *>(d_blob); d.ranges = const_cast<float *>(d_ranges);
    hipLaunchKernelGGL(quant_decode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, stream, d, bdw);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}
}  // namespace

extern "C" {

int melspec_tga_create(melspec_tga **out, int device) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    melspec_tga *q = new (std::nothrow) melspec_tga();
    if (!q) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    q->dev = info;
    if (hipSetDevice(info.device) != hipSuccess || hipStreamCreate(&q->stream) != hipSuccess) {
        delete q;
        return fail(MELSPEC_ERR_UNAVAILABLE, "hipStreamCreate failed");
    }
    *out = q;
    return MELSPEC_OK;
}

void melspec_tga_destroy(melspec_tga *q) {
    if (!q) return;
    if (q->dev.device >= 0) (void)hipSetDevice(q->dev.device);
    if (q->stream) { (void)hipStreamSynchronize(q->stream); (void)hipStreamDestroy(q->stream); }
    q->keys.release(); q->ranges.release(); q->h2d.release(); q->d2h.release(); q->unit_ext.release();
    delete q;
}

int melspec_tga_layout(int n_mels, size_t width, uint32_t *n_chunks, size_t *chunk_stride, size_t *last_chunk_bytes) {
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    const TgaLayout l = tga_layout(static_cast<uint32_t>(n_mels), width);
    if (n_chunks) *n_chunks = l.chunks;
    if (chunk_stride) *chunk_stride = l.chunk_stride;
    if (last_chunk_bytes) *last_chunk_bytes = l.last_bytes;
    return MELSPEC_OK;
}

int melspec_tga_encode_device(melspec_tga *q, const float *d_images, size_t image_stride, int n_mels, size_t width,
                              uint32_t n_images, uint8_t *d_blobs, size_t blob_stride, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    if (image_stride < static_cast<size_t>(n_mels) * width) return fail(MELSPEC_ERR_INVALID_ARG, "image_stride < n_mels * width");
    return quant_encode(q, d_images, image_stride, static_cast<uint32_t>(n_mels), width, n_images, d_blobs, blob_stride, true, nullptr,
                        stream ? static_cast<hipStream_t>(stream) : q->stream);
}

// PCM -> TGA bytes with the image read once: while it stores the image (mel-major) the mel kernel leaves the extremes of every work
// unit behind (BatchDesc::d_unit_ext: one wave-wide reduction and one 8-byte store per unit), a one-wave-per-image kernel folds them
// into the quantiser's keys, so only the encoding pass reads the image again -- 5 B/pixel moved for 5 B/pixel algorithmic, where minmax + encode moved 9 ("4x smaller D2H").  The bytes are
// those of melspec_compute_uniform_device_interleaved(.., major_column_order = 0, min_width) followed by melspec_tga_encode_device.
int melspec_tga_encode_pcm_uniform_device(melspec_tga *q, melspec_ctx *c, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                          uint32_t n_clips, uint64_t min_width, float *d_images, uint8_t *d_blobs, size_t blob_stride,
                                          void *stream) {
    if (!q || !c) return fail(MELSPEC_ERR_INVALID_ARG, "tga / ctx is NULL");
    if (q->dev.device != c->dev.device) return fail(MELSPEC_ERR_INVALID_ARG, "the codec and the context are on different devices");
    if (min_width % 2 != 0) return fail(MELSPEC_ERR_INVALID_ARG, "min_width must be even");   // src/mel.rs:488
    if (n_clips == 0) return MELSPEC_OK;
    uint64_t fpc; ctx_num_frames(c, clip_len, fpc);
    if (fpc == 0) return fail(MELSPEC_ERR_INVALID_ARG, "frames is empty");                      // src/mel.rs:487
    if (!d_pcm || !d_images || !d_blobs) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const uint64_t width = interleaved_width(fpc, min_width);
    const size_t image_stride = static_cast<size_t>(c->n_mels) * width;
    if (!c->fast || width > kTgaMaxWidth) {
        // geometries on the generic / 512-point kernels, and images wider than one TGA chunk: the two-pass form
        int rc = melspec_compute_uniform_device_interleaved(c, d_pcm, clip_stride, clip_len, n_clips, d_images, 0, min_width, s);
        if (rc) return rc;
        return melspec_tga_encode_device(q, d_images, image_stride, c->n_mels, width, n_clips, d_blobs, blob_stride, s);
    }
    QuantDesc d;
    uint32_t items, bpx, bdw;
    int rc = quant_plan(q, d, d_images, image_stride, static_cast<uint32_t>(c->n_mels), width, n_clips, d_blobs, blob_stride, true, items, bpx, bdw);
    if (rc) return rc;
    d.img = d_images; d.blob = d_blobs; d.ranges = nullptr;
    if (q->keys_used && q->keys_stream != s) HIP_TRY(hipStreamSynchronize(q->keys_stream));
    q->keys_used = true; q->keys_stream = s;
    BatchPlan pl = plan_uniform(d_pcm, d_images, clip_stride, fpc, n_clips, c->n_mels, ctx_frames_per_unit(c, true), width, true);
    if ((rc = q->unit_ext.ensure(static_cast<size_t>(pl.desc.n_units) * 2 * sizeof(int) + 16))) return rc;
    pl.desc.d_unit_ext = static_cast<int *>(q->unit_ext.p);
    if ((rc = launch_ctx(c, pl.desc, s))) return rc;
    // one chunk per image: item == clip; its units' records -> its keys
    hipLaunchKernelGGL(quant_keys_from_units_kernel, dim3(n_clips), dim3(64), 0, s, pl.desc.d_unit_ext, pl.desc.units_per_clip, n_clips, d.keys);
    hipLaunchKernelGGL(quant_encode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, s, d, bdw);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_tga_decode_device(melspec_tga *q, const uint8_t *d_blobs, size_t blob_stride, int n_mels, size_t width,
                              uint32_t n_images, float *d_images, size_t image_stride, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    if (image_stride < static_cast<size_t>(n_mels) * width) return fail(MELSPEC_ERR_INVALID_ARG, "image_stride < n_mels * width");
    return quant_decode(q, d_blobs, blob_stride, static_cast<uint32_t>(n_mels), width, n_images, d_images, image_stride, true, nullptr,
                        stream ? static_cast<hipStream_t>(stream) : q->stream);
}

int melspec_quantize_device(melspec_tga *q, const float *d_frame, size_t n, uint8_t *d_out, float *d_range, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (!d_range) return fail(MELSPEC_ERR_INVALID_ARG, "range pointer is NULL");
    return quant_encode(q, d_frame, n, 1, n, 1, d_out, round_up4(n), false, d_range, stream ? static_cast<hipStream_t>(stream) : q->stream);
}

int melspec_dequantize_device(melspec_tga *q, const uint8_t *d_data, size_t n, const float *d_range, float *d_out, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    return quant_decode(q, d_data, round_up4(n), 1, n, 1, d_out, n, false, d_range, stream ? static_cast<hipStream_t>(stream) : q->stream);
}

int melspec_tga_synchronize(melspec_tga *q) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    HIP_TRY(hipStreamSynchronize(q->stream));
    return MELSPEC_OK;
}

int melspec_quantize_host(melspec_tga *q, const float *frame, size_t n, uint8_t *out, float *range) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (!range) return fail(MELSPEC_ERR_INVALID_ARG, "range pointer is NULL");
    if (n == 0) { range[0] = INFINITY; range[1] = -INFINITY; return MELSPEC_OK; }       // the folds' start values
    if (!frame || !out) return fail(MELSPEC_ERR_INVALID_ARG, "frame/out is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(n * sizeof(float)))) return rc;
    if ((rc = q->d2h.ensure(round_up4(n) + 16))) return rc;
    if ((rc = q->ranges.ensure(16))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, frame, n * sizeof(float), hipMemcpyHostToDevice, q->stream));
    if ((rc = melspec_quantize_device(q, static_cast<const float *>(q->h2d.p), n, static_cast<uint8_t *>(q->d2h.p),
                                      static_cast<float *>(q->ranges.p), q->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, n, hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipMemcpyAsync(range, q->ranges.p, 2 * sizeof(float), hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    return MELSPEC_OK;
}

int melspec_dequantize_host(melspec_tga *q, const uint8_t *data, size_t n, const float *range, float *out) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n == 0) return MELSPEC_OK;
    if (!data || !range || !out) return fail(MELSPEC_ERR_INVALID_ARG, "data/range/out is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(round_up4(n) + 16))) return rc;
    if ((rc = q->d2h.ensure(n * sizeof(float)))) return rc;
    if ((rc = q->ranges.ensure(16))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, data, n, hipMemcpyHostToDevice, q->stream));
    HIP_TRY(hipMemcpyAsync(q->ranges.p, range, 2 * sizeof(float), hipMemcpyHostToDevice, q->stream));
    if ((rc = melspec_dequantize_device(q, static_cast<const uint8_t *>(q->h2d.p), n, static_cast<const float *>(q->ranges.p),
                                        static_cast<float *>(q->d2h.p), q->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, n * sizeof(float), hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    return MELSPEC_OK;
}

int melspec_tga_encode_host(melspec_tga *q, const float *data, size_t len, int n_mels, uint8_t *out, size_t out_capacity,
                            uint32_t *n_chunks) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_chunks) *n_chunks = 0;
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    if (len % static_cast<size_t>(n_mels)) return fail(MELSPEC_ERR_INVALID_ARG, "data length is not a multiple of n_mels");
    const size_t width = len / n_mels;
    if (width == 0) return MELSPEC_OK;
    if (!data || !out) return fail(MELSPEC_ERR_INVALID_ARG, "data/out is NULL");
    const TgaLayout l = tga_layout(static_cast<uint32_t>(n_mels), width);
    const size_t region = l.chunk_stride * l.chunks;
    if (out_capacity < region - l.chunk_stride + l.last_bytes) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(len * sizeof(float)))) return rc;
    if ((rc = q->d2h.ensure(region))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, data, len * sizeof(float), hipMemcpyHostToDevice, q->stream));
    if ((rc = melspec_tga_encode_device(q, static_cast<const float *>(q->h2d.p), len, n_mels, width, 1,
                                        static_cast<uint8_t *>(q->d2h.p), region, q->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, region - l.chunk_stride + l.last_bytes, hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    if (n_chunks) *n_chunks = l.chunks;
    return MELSPEC_OK;
}

int melspec_tga_decode_host(melspec_tga *q, const uint8_t *blob, size_t n_bytes, float *out, size_t out_capacity, size_t *n_values) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_values) *n_values = 0;
    if (!blob || n_bytes < kTgaHeader) return fail(MELSPEC_ERR_INVALID_ARG, "failed to fill whole buffer");   // read_exact, src/quant.rs:74-75
    const size_t npx = n_bytes - kTgaHeader;
    if (npx == 0) return MELSPEC_OK;
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    if (out_capacity < npx) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    if (npx > 0xffffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "more than 2^32-1 pixels");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(round_up4(n_bytes) + 16))) return rc;
    if ((rc = q->d2h.ensure(npx * sizeof(float)))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, blob, n_bytes, hipMemcpyHostToDevice, q->stream));
    // the header's width/height are ignored by the reference too: everything after byte 26 is one row of pixels
    QuantDesc d;
    uint32_t items, bpx, bdw;
    d = QuantDesc{};
    d.rows = 1; d.width = static_cast<uint32_t>(npx); d.n_images = 1; d.chunks = 1; d.chunk_w = d.width;
    d.header = kTgaHeader; d.vec = 1; d.img_out = static_cast<float *>(q->d2h.p); d.blob = static_cast<uint8_t *>(q->h2d.p);
    items = 1; bpx = 0; (void)bpx;
    bdw = static_cast<uint32_t>(((kTgaHeader + npx + 3) / 4 + kQuantDwPerBlock - 1) / kQuantDwPerBlock);
    hipLaunchKernelGGL(quant_decode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, q->stream, d, bdw);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, npx * sizeof(float), hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    if (n_values) *n_values = npx;
    return MELSPEC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// VAD column classification (vad_boundaries, src/vad.rs:256-340)
// ------------------------------------------------------------------------------------
extern "C" {

void melspec_vad_default_settings(melspec_vad_settings *s) {          // DetectionSettings::default, src/vad.rs:13-22
    if (!s) return;
    s->min_energy = 0.98; s->min_y = 11; s->min_x = 5; s->min_mel = 2;
}

size_t melspec_vad_mask_len(int n_mels, size_t width) { return (n_mels < 3 || width < 3) ? 0 : width - 2; }

int melspec_vad_boundaries_device(const float *d_images, size_t image_stride, int n_mels, size_t width, uint32_t n_images,
                                  const melspec_vad_settings *settings, uint8_t *d_raw, uint8_t *d_smoothed, size_t mask_stride,
                                  uint32_t *d_longest_run, void *stream) {
    if (!settings) return fail(MELSPEC_ERR_INVALID_ARG, "settings is NULL");
    if (n_mels < 0 || settings->min_y < 0 || settings->min_mel < 0) return fail(MELSPEC_ERR_INVALID_ARG, "negative size");
    const size_t n = melspec_vad_mask_len(n_mels, width);
    if (n_images == 0) return MELSPEC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {                                        // EdgeInfo::new(empty, empty), src/vad.rs:270-272
        if (d_longest_run) HIP_TRY(hipMemsetAsync(d_longest_run, 0, sizeof(uint32_t) * n_images, s));
        return MELSPEC_OK;
    }
    if (!d_images || !d_smoothed) return fail(MELSPEC_ERR_INVALID_ARG, "image/mask pointer is NULL");
    if (!d_raw) return fail(MELSPEC_ERR_INVALID_ARG, "d_raw is NULL (the vote reads the raw mask)");
    if (mask_stride < n) return fail(MELSPEC_ERR_INVALID_ARG, "mask_stride < width - 2");
    if (image_stride < static_cast<size_t>(n_mels) * width) return fail(MELSPEC_ERR_INVALID_ARG, "image_stride < n_mels * width");
    if (width > 0xffffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "image wider than 2^32-1 columns");
    VadDesc d{};
    d.img = d_images; d.raw = d_raw; d.smoothed = d_smoothed; d.longest = d_longest_run;
    d.img_stride = image_stride; d.mask_stride = mask_stride;
    d.height = static_cast<uint32_t>(n_mels); d.width = static_cast<uint32_t>(width); d.n_images = n_images;
    d.min_mel = settings->min_mel; d.min_y = settings->min_y; d.thr = settings->min_energy * settings->min_energy;
    const uint64_t bpi = (n + 255) / 256;
    if (bpi * n_images > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "batch too large for one launch");
    const unsigned grid = static_cast<unsigned>(bpi * n_images);
    hipLaunchKernelGGL(vad_raw_kernel, dim3(grid), dim3(256), 0, s, d, static_cast<uint32_t>(bpi), d_raw);
    hipLaunchKernelGGL(vad_smooth_kernel, dim3(grid), dim3(256), 0, s, d, static_cast<uint32_t>(bpi), d_raw);
    if (d_longest_run) hipLaunchKernelGGL(vad_run_kernel, dim3(n_images), dim3(64), 0, s, d);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_vad_boundaries_host(int device, const float *image, int n_mels, size_t width, const melspec_vad_settings *settings,
                                uint8_t *raw_out, uint8_t *smoothed_out, uint32_t *longest_run) {
    if (!settings) return fail(MELSPEC_ERR_INVALID_ARG, "settings is NULL");
    if (longest_run) *longest_run = 0;
    const size_t n = melspec_vad_mask_len(n_mels, width);
    if (n == 0) return MELSPEC_OK;
    if (!image || !smoothed_out) return fail(MELSPEC_ERR_INVALID_ARG, "image/out is NULL");
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(info.device));
    DevBuf img, masks, run;
    const size_t px = static_cast<size_t>(n_mels) * width, ms = (n + 15) & ~static_cast<size_t>(15);
    auto done = [&](int code) { img.release(); masks.release(); run.release(); return code; };
    if ((rc = img.ensure(px * sizeof(float))) || (rc = masks.ensure(2 * ms)) || (rc = run.ensure(16))) return done(rc);
    if (hipMemcpy(img.p, image, px * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    uint8_t *m = static_cast<uint8_t *>(masks.p);
    rc = melspec_vad_boundaries_device(static_cast<const float *>(img.p), px, n_mels, width, 1, settings, m, m + ms, ms,
                                       static_cast<uint32_t *>(run.p), nullptr);
    if (rc) return done(rc);
    if (hipDeviceSynchronize() != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "vad kernels failed"));
    if (raw_out && hipMemcpy(raw_out, m, n, hipMemcpyDeviceToHost) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    if (hipMemcpy(smoothed_out, m + ms, n, hipMemcpyDeviceToHost) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    if (longest_run && hipMemcpy(longest_run, run.p, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    return done(MELSPEC_OK);
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
        if ((rc = allow_big_lds(&generic_frame_kernel<kGenericNT>, "hipFuncSetAttribute(generic_frame_kernel)"))) return bail(rc);
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

static int synth_launch(float *d_out, uint64_t clip_stride, uint64_t first_sample, uint64_t n_samples, uint64_t first_clip,
                        uint32_t n_clips, uint32_t seed, void *stream) {
    if (n_clips == 0 || n_samples == 0) return MELSPEC_OK;
    if (!d_out) return fail(MELSPEC_ERR_INVALID_ARG, "d_out is NULL");
    const uint64_t total = static_cast<uint64_t>(n_clips) * n_samples;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_pcm_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_out, clip_stride, n_samples, first_clip, n_clips, seed, first_sample);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_synth_pcm_device(float *d_out, uint64_t clip_stride, uint64_t clip_len, uint64_t first_clip,
                             uint32_t n_clips, uint32_t seed, void *stream) {
    return synth_launch(d_out, clip_stride, 0, clip_len, first_clip, n_clips, seed, stream);
}

int melspec_synth_pcm_window_device(float *d_out, uint64_t clip_stride, uint64_t first_sample, uint64_t n_samples,
                                    uint64_t first_clip, uint32_t n_clips, uint32_t seed, void *stream) {
    return synth_launch(d_out, clip_stride, first_sample, n_samples, first_clip, n_clips, seed, stream);
}

}  // extern "C"
