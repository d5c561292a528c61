// melspec_kernels.hpp -- gfx950 kernels of libmelspec_hip.so (included once, by melspec_hip.hip).
//
//   whisper400_kernel   fused n_fft=400 log-mel (phases in whisper_fast.hpp), f32
//   generic_frame_kernel  any-geometry log-mel / Kaldi fbank, one frame per workgroup, f64 DFT
//   cmn_kernel          per-clip cepstral mean normalisation (src/fbank.rs:224-233)
//   synth_pcm_kernel    hash-noise PCM generator for benches/tests
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "whisper_fast.hpp"
#include "whisper_wave.hpp"
#include "fbank_wave.hpp"
#include "whisper_wave_f64.hpp"
#include "whisper_six.hpp"
#include "whisper_six64.hpp"
#include "whisper_fix64.hpp"
#include "stream_plan.hpp"
#include "pow2_wave.hpp"

// Issue priority of the wave (s_setprio 0..3).  The persistent kernels raise it as a unit progresses (loads + first FFT
// stage 0, second stage 1, mel / log / store 2): the waves sharing a SIMD then stop advancing in lock-step through the
// VMEM-, VALU- and LDS-heavy phases.  Measured (profiles/r01_variants.txt): six-frame Whisper kernel -2.3 .. -3.5 %, fused
// 512-point kernels -8 % (Kaldi) / -11 % (Whisper-512) / 0 (NeMo), precise kernel -3.8 %; the 5-frame kernel with two
// 8-wave workgroups per CU loses 1-4 % under every table tried in its round-robin form and stays at the default priority
// there; its run-per-wave form (whisper400_wave_runs_kernel) gains 5 % (cfg4 9.02 -> 8.53 ms).  Round 3, after the LDS / VALU trims of
// the six-frame kernel (same box, config 2): 0/1/2 0.2867 ms; 1/2/3 0.2863; 0/1/3 0.2875; 0/2/3 0.2887; 0/1/1 0.2917; 0/0/1 0.2934;
// 2/1/0 0.2898; none 0.3146 (+9.8 %).  A one-instruction touch of the next unit's 4 KB of PCM (64-byte pieces, a unit ahead): +7 %.
#ifndef MELSPEC_NO_PRIO
#define MS_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define MS_PRIO(n)
#endif
namespace melspec {

// How work units (tiles of frames) map onto clips.  Uniform batches are pure arithmetic;
// ragged batches look the clip up in a per-16-units table and a prefix table of units per clip.
struct BatchDesc {
    const float *pcm;
    float *out;
    uint64_t clip_stride;      // uniform: samples between clip starts
    uint64_t out_stride;       // uniform: floats between clip outputs
    uint64_t frames_per_clip;  // uniform
    uint32_t units_per_clip;   // uniform
    uint32_t n_clips;
    uint64_t n_units;
    uint64_t out_width;        // uniform: output columns per clip (>= frames_per_clip; the excess is zero-filled)
    int mel_major;             // uniform: 0 = [frame][mel] rows, 1 = [mel][out_width] rows (interleave_frames, src/mel.rs:480-544)
    int frames_per_unit;       // frames a work unit covers (set by the planner; picks the kernel on contexts that have two)
    int sync_rounds;           // LAYOUT kernels: re-align the waves of a workgroup once per round (set for mel-major stores)
    const uint64_t *d_off;       // ragged (device): first sample of clip c
    const uint64_t *d_frames;    // ragged: frames in clip c
    const uint64_t *d_out_off;   // ragged: first output float of clip c
    const uint64_t *d_unit_prefix;  // ragged: first unit of clip c, [n_clips+1]
    const uint32_t *d_unit_block;   // ragged: clip that holds unit k * kUnitBlock
    const uint64_t *d_n_units;      // ragged batches planned on the device (plan_ragged_device_kernel): the unit count lives here and
                                    // n_units above is the host's upper bound (grid and scratch sizes)
    const uint32_t *d_order;        // ragged, host-planned, optional: the clips longest first (whole-clip kernels take them in this order)
    uint32_t *d_ticket;             //   and the counter they take them from (zero when the launch starts)
    uint64_t stat_frames;           // frames of the batch as the planner counted them (host-planned ragged batches: the true total; 0: not set) -- the
                                    //   denominator of the guard statistics (n_units * frames_per_unit over-counts short clips up to 6 x)
    int *d_unit_ext;                // uniform mel-major layouts, optional: [n_units][2] = {smallest, largest} biased value (phase 4) every work unit
                                    //   stored -- the TGA quantiser's first pass (tga_quant.hpp) then reads 8 bytes per unit instead of the image
};

__device__ __forceinline__ uint64_t batch_n_units(const BatchDesc &b) { return b.d_n_units ? *b.d_n_units : b.n_units; }

constexpr uint32_t kUnitBlock = 16;   // granularity of BatchDesc::d_unit_block

struct UnitLoc {
    const float *pcm;   // first sample of the clip
    float *out;         // first output float of the clip
    uint64_t frames;    // frames in the clip
    uint64_t unit;      // unit index inside the clip
    uint32_t clip;      // the clip
};

__device__ __forceinline__ UnitLoc locate_unit(const BatchDesc &b, uint64_t unit) {
    UnitLoc r;
    if (b.d_unit_prefix == nullptr) {
        const uint64_t clip = unit / b.units_per_clip;
        r.unit = unit - clip * b.units_per_clip;
        r.pcm = b.pcm + clip * b.clip_stride;
        r.out = b.out + clip * b.out_stride;
        r.frames = b.frames_per_clip;
        r.clip = static_cast<uint32_t>(clip);
    } else {
        // d_unit_block[k] = clip that holds unit k * kUnitBlock.  The records of that clip and of the next one are fetched
        // together (second round trip); only clips shorter than a block of units need the walk (third and later trips).
        uint32_t lo = b.d_unit_block[unit / kUnitBlock];
        uint64_t p0 = b.d_unit_prefix[lo], p1 = b.d_unit_prefix[lo + 1];
        uint64_t off0 = b.d_off[lo], off1 = b.d_off[lo + 1];                 // [lo + 1] of the last clip: the next array of the
        uint64_t oo0 = b.d_out_off[lo], oo1 = b.d_out_off[lo + 1];           // same plan buffer, fetched and not used
        uint64_t fr0 = b.d_frames[lo], fr1 = b.d_frames[lo + 1];
        if (p1 <= unit) {                                                    // prefix[n_clips] = n_units > unit
            ++lo;
            p0 = p1; off0 = off1; oo0 = oo1; fr0 = fr1;
            if (b.d_unit_prefix[lo + 1] <= unit) {
                do { ++lo; } while (b.d_unit_prefix[lo + 1] <= unit);
                p0 = b.d_unit_prefix[lo]; off0 = b.d_off[lo]; oo0 = b.d_out_off[lo]; fr0 = b.d_frames[lo];
            }
        }
        r.unit = unit - p0;
        r.pcm = b.pcm + off0;
        r.out = b.out + oo0;
        r.frames = fr0;
        r.clip = lo;
    }
    return r;
}

// XCD-aware workgroup order.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs,
// each with its own L2.  Neighbouring workgroups share data -- the 240-sample frame-tail halo on the read side
// and, for mel-major stores, the cache lines at the ends of their 20-byte row pieces -- so consecutive
// *logical* workgroups are placed on the same XCD: logical = (id % 8) * (grid / 8) + id / 8 (grids are launched
// as multiples of 8).  Measured on the mel-major store: HBM writes 690 MB -> see profiles/r01_variants.txt
// (two L2s each holding half a dirty line write it back twice).
constexpr unsigned kXcds = 8;
__device__ __forceinline__ unsigned xcd_logical_block() {
    const unsigned per = gridDim.x / kXcds;
    return (gridDim.x % kXcds) ? blockIdx.x : (blockIdx.x % kXcds) * per + blockIdx.x / kXcds;
}

// ------------------------------------------------------------------------------------
// Fused Whisper kernels, n_fft = 400 (phases in whisper_wave.hpp / whisper_six.hpp).  A work unit is a run of
// consecutive frames of one clip (5 or 6) and belongs to one wavefront; the waves of a workgroup share only the
// table blob.  LDS: [table blob][WAVES x private slice].  No workgroup barrier in the unit loop.
// ------------------------------------------------------------------------------------
// MELSPEC_PRECISION_AUTO: the f32 kernels recompute the frames their precision guard does not trust (wave_phase4) in f64 on the
// spot (whisper_fix64.hpp).  tab == nullptr: guard off (MELSPEC_PRECISION_F32).  count: frames recomputed since the context was
// created (statistics; one atomic per recomputed frame).
struct FixSink {
    const double *tab;      // FixTables in global memory (nullptr: no guard, or -- the f64 kernel -- statistics only)
    uint64_t *list;         // one entry per unit of the launch (+ a round of slack): every wave notes the units it has to revisit in
                            // the part of it that its own units index, so no two waves share an entry
    // Statistics of the launch (guard_wave_done).  acc: one device word per context, zero between launches, to which every workgroup
    // adds {its frames that tripped the guard, 1 << 40} with ONE relaxed atomic; the workgroup that completes the count adds the
    // launch's total to count[0] (frames tripped since the context was created), zeroes acc and writes the launch's figures into
    // host-mapped memory, which the host polls before its next call -- no copy, no event, nothing on the stream, and no fence: a
    // release at agent scope writes back the XCD's whole L2 (measured: +19 % on the 1024-workgroup kernel).
    unsigned long long *acc;     // nullptr: no statistics
    unsigned long long *count;
    unsigned long long *host;    // host-mapped {seq << 40 | tripped, seq << 40 | frames}; the pair is valid when both carry the same seq
    unsigned n_groups;           // workgroups of this launch
    unsigned seq;                // number of this launch (24 bits)
    unsigned long long frames;   // frames of this launch (ragged batches: the host's upper bound)
    // The vote of MELSPEC_PRECISION_AUTO (round 4): which kernel computes THIS batch is decided from the batch itself, inside the
    // launch.  The first work unit of every wave of the first `vote_groups` workgroups (all of them resident when the launch starts)
    // is the sample; each of those workgroups adds {frames that tripped the guard, frames, 1} to `vote` with one relaxed atomic, the
    // workgroup that completes the tally writes `decision` = seq << 2 | 2 | heavy (heavy: more than 1/8 of the sampled frames
    // tripped) and zeroes the tally.  Nobody waits: a wave looks at `decision` after each unit until it carries this launch's number;
    // on "heavy" it stops -- the f64 kernel queued behind this launch (gated on the same word) computes the whole batch, otherwise
    // that launch returns at once and this one finishes with its recompute tail.  The outcome is a function of the batch alone.
    unsigned long long *vote;    // nullptr: no vote (the f32 kernel + recompute tail whatever the input)
    unsigned *decision;          // kVoteSlots copies of the verdict, kVoteSlotStride words apart: workgroup g reads copy g % kVoteSlots (every
                                 // poller of the launch reading ONE word made that word's memory channel the bottleneck: the agent-scope
                                 // loads are served by memory, not by an L2, and queued for tens of microseconds)
    unsigned vote_groups;
};
constexpr unsigned kVoteSlots = 256, kVoteSlotStride = 64;      // 256 B apart

__device__ __forceinline__ uint64_t scalar64(uint64_t v) {
    // the builtin returns a signed int: without the casts a low word with bit 31 set sign-extends over the high word
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}

// locate_unit's results are wave-uniform when `unit` is, but its 64-bit division runs on the vector unit and leaves them in VGPRs; this
// moves them into SGPRs.  Only where the registers are missing (the twelve-wave f32 NeMo kernel: spills gone, -1.1 %): on the other
// round-robin kernels the readfirstlanes put the division's latency in front of everything that follows -- mel-major +1.7 %, mel-major
// F64 +1.5 %, f64 NeMo +5.2 / +6.5 % (128 / 80 mels), same box (profiles/r05_f32_512.txt)
__device__ __forceinline__ UnitLoc scalar_loc(UnitLoc r) {
    r.unit = scalar64(r.unit); r.frames = scalar64(r.frames); r.clip = __builtin_amdgcn_readfirstlane(r.clip);
    r.pcm = reinterpret_cast<const float *>(scalar64(reinterpret_cast<uint64_t>(r.pcm)));
    r.out = reinterpret_cast<float *>(scalar64(reinterpret_cast<uint64_t>(r.out)));
    return r;
}
constexpr int kStatShift = 40;
constexpr unsigned long long kStatMask = (1ull << kStatShift) - 1;

__device__ __forceinline__ unsigned vote_poll(const FixSink &fx);
constexpr unsigned kVoteDecided = 2u, kVoteHeavy = 1u;
// A wave of a guarded launch is through (every wave calls this, also one without units).  wg: two zeroed LDS words of the
// workgroup {frames that tripped the guard, waves through}.
__device__ __forceinline__ void guard_wave_done(const FixSink &fx, unsigned *wg, int waves, int lane, unsigned flagged) {
    if (fx.acc == nullptr || lane != 0) return;
    if (flagged) __hip_atomic_fetch_add(wg, flagged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // LDS operations of a lane execute in order
    const unsigned through = __hip_atomic_fetch_add(wg + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (through + 1 != static_cast<unsigned>(waves)) return;
    const unsigned long long total = __hip_atomic_load(wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long old = __hip_atomic_fetch_add(fx.acc, total | (1ull << kStatShift), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> kStatShift) + 1 != fx.n_groups) return;
    const unsigned long long tripped = ((old & kStatMask) + total) & kStatMask;
    __hip_atomic_store(fx.acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // every other workgroup of the launch has been here
    if (fx.vote) {
        // A voting launch that stood down: the f64 launch behind it reports the batch.  Every sampling workgroup has cast its vote before
        // it arrived here, so this launch's verdict has been stored -- by relaxed stores that nothing orders against this relaxed load
        // (ADVICE r04): a stale word would make both launches count the batch.  One wave per launch retries until the word carries this
        // launch's number (bounded; a fence here or on the stores would write back an XCD's L2).
        unsigned v = vote_poll(fx);
        for (unsigned spin = 0; v == 0 && spin < 4096; ++spin) { __builtin_amdgcn_s_sleep(1); v = vote_poll(fx); }
        if (v & kVoteHeavy) return;
    }
    if (tripped) __hip_atomic_fetch_add(fx.count, tripped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long tag = static_cast<unsigned long long>(fx.seq & 0xffffffu) << kStatShift;
    __hip_atomic_store(fx.host, tag | tripped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fx.host + 1, tag | (fx.frames & kStatMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the vote (FixSink::vote) ----------------------------------------------------------------------------------------------------
constexpr int kVoteFramesShift = 24, kVoteGroupsShift = 48;        // tally word: tripped | frames << 24 | groups << 48
// heavy: the sample says the f64 kernel is the cheaper way to the tolerance (crossover of f32 + tail against it: 14 % of the frames at
// 80 mels, 10 % at 128)
__device__ __forceinline__ bool vote_is_heavy(unsigned long long tripped, unsigned long long frames) { return tripped * 8 > frames; }

// 0: not known yet; kVoteDecided (| kVoteHeavy): this launch's verdict.  Wave-uniform.
__device__ __forceinline__ unsigned vote_poll(const FixSink &fx) {
    const unsigned d = __builtin_amdgcn_readfirstlane(__hip_atomic_load(fx.decision + (blockIdx.x % kVoteSlots) * kVoteSlotStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return (d >> 2) == (fx.seq & 0xffffffu) ? (d & 3u) : 0u;
}
// A wave of a sampling workgroup reports its first unit (also a wave without units: 0, 0); every lane of the wave calls this.  wg: three
// zeroed LDS words of the workgroup {tripped, frames, waves that have reported}.
__device__ __forceinline__ void vote_cast(const FixSink &fx, unsigned *wg, int waves, int lane, unsigned tripped, unsigned frames) {
#ifdef MELSPEC_VOTE_NOCAST
    return;
#endif
    unsigned long long sum = 0;
    bool last = false;
    if (lane == 0) {
        if (tripped) __hip_atomic_fetch_add(wg, tripped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // LDS operations of a lane execute in order
        if (frames) __hip_atomic_fetch_add(wg + 1, frames, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_fetch_add(wg + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1 == static_cast<unsigned>(waves)) {
            const unsigned long long t = __hip_atomic_load(wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long f = __hip_atomic_load(wg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long mine = t | (f << kVoteFramesShift) | (1ull << kVoteGroupsShift);
            sum = __hip_atomic_fetch_add(fx.vote, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
            last = (sum >> kVoteGroupsShift) == fx.vote_groups;
            if (last) __hip_atomic_store(fx.vote, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // every sampling workgroup has been here
        }
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    sum = scalar64(sum);
    const bool heavy = vote_is_heavy(sum & ((1ull << kVoteFramesShift) - 1), (sum >> kVoteFramesShift) & ((1ull << kVoteFramesShift) - 1));
    const unsigned verdict = (fx.seq & 0xffffffu) << 2 | kVoteDecided | (heavy ? kVoteHeavy : 0u);
#pragma unroll
    for (unsigned k = 0; k < kVoteSlots; k += 64)                      // the wave that completes the tally publishes every copy
        __hip_atomic_store(fx.decision + (k + lane) * kVoteSlotStride, verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// What a wave does after a unit while the verdict is unknown.  wg[3]: the workgroup's copy of the verdict (0 until one of its waves
// has seen it): one ds_read per unit.  The global word is read by one wave in four per unit (the waves of a workgroup start together and
// stay roughly in step, so that is one wave per SIMD and unit): waiting for that load also waits for the stores the wave has in flight
// -- gfx950 counts both in vmcnt -- and with every wave polling after every unit all four waves of a SIMD stalled together (+12 us on
// the 290 us launch of config 2; this form: see profiles/r04_vote.txt).
__device__ __forceinline__ unsigned vote_check(const FixSink &fx, unsigned *wg, unsigned units_done, int wave) {
#ifdef MELSPEC_VOTE_NOPOLL
    return 0;
#endif
    unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(wg + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (v == 0 && ((static_cast<unsigned>(wave) ^ units_done) & 3u) == 0) {
        v = vote_poll(fx);
        if (v) __hip_atomic_store(wg + 3, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return v;
}

struct FastParams {
    BatchDesc b;
    const float *d_blob;
    int blob_len;      // floats, multiple of 4
    int hop;
    int n_mels;
    int slice_floats;  // floats per wave (5-frame kernels)
    MelSlots slots;
    FixSink fix;
};

// One-lane-down shift across the whole wave (lane l receives lane l+1's value).
__device__ __forceinline__ float wave_shift_down1(float v) {
    // bound_ctrl on (lane 63, which has no source lane, reads 0): no `old` operand, so no v_mov in front of every shift
    const int x = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}


// ---- extremes of the stored images, per work unit (BatchDesc::d_unit_ext) -------------------------------------------------------
// Wave-wide minimum / maximum of non-negative ints (the biased values of phase 4): an inclusive scan inside each row of 16 lanes
// (row_shr 1, 2, 4, 8), then row_bcast:15 / :31 carry the row results up; lane 63 holds the result.
template <bool MAX>
__device__ __forceinline__ int wave_reduce_int(int v) {
    constexpr int ident = MAX ? 0 : 0x7fffffff;
    auto op = [](int a, int b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
// kmin / kmax: this lane's extremes of the biased values it stored for the unit (0x7fffffff / 0 when it stored nothing, or when its
// frame is going to be recomputed and reports then); every lane of the wave calls these.  Two atomics per unit on the image's keys
// instead of the 8-byte record cost the mel-major kernel 19 %; accumulating in registers over a contiguous range of rounds per
// workgroup (so that a wave stays inside a clip) took the atomics away and cost 30 %: the interleaved rounds are what keeps the
// 24-byte pieces of the stores of all CUs inside one compact region.
__device__ __forceinline__ void unit_ext_store(int *ext, int lane, int kmin, int kmax) {
    const int lo = wave_reduce_int<false>(kmin), hi = wave_reduce_int<true>(kmax);
    if (lane == 0) *reinterpret_cast<int2 *>(ext) = make_int2(lo, hi);
}
// the recompute tail: the frames it recomputed join the record its own wave wrote in the unit loop
__device__ __forceinline__ void unit_ext_merge(int *ext, int lane, int kmin, int kmax) {
    const int lo = wave_reduce_int<false>(kmin), hi = wave_reduce_int<true>(kmax);
    if (lane == 0) {
        const int2 old = *reinterpret_cast<const int2 *>(ext);
        *reinterpret_cast<int2 *>(ext) = make_int2(lo < old.x ? lo : old.x, hi > old.y ? hi : old.y);
    }
}

constexpr int kSixFixOff = 1408;      // float offset of the f64 scratch (400 doubles) inside a six-frame slice: behind power rows and maxima
constexpr int kWaveFixOff = 1104;     // the same inside a five-frame slice
static_assert(kSixFixOff >= SixLayout::kPmaxOff + kSixFrames * SixLayout::kPmaxStride && kSixFixOff + 2 * FixTables::kScratchDoubles <= SixLayout::slice_floats(), "six-frame slice");
static_assert(kWaveFixOff >= WaveLayout::kPmaxOff + kFPW * WaveLayout::kPmaxStride && kWaveFixOff + 2 * FixTables::kScratchDoubles <= WaveLayout::slice_floats(), "five-frame slice");

// f64 power row of frame `f` of the unit (whisper_fix64.hpp): every lane of the wave takes part
// `next`: the frame recomputed after this one (nullptr: none); its samples are loaded while steps 2-4 run
__device__ __forceinline__ void fix_power_row(int lane, FixSamples &smp, const float *next, const double *tab, const FixTw &tw, float *slice, int scratch_off, float *prow) {
    double *z = reinterpret_cast<double *>(slice + scratch_off);
    fix_step1(lane, smp, tab, z);
    if (next) fix_load_samples(lane, next, smp);
    __builtin_amdgcn_wave_barrier();
    fix_step2(lane, tw, z);
    __builtin_amdgcn_wave_barrier();
    fix_step3(lane, z);
    __builtin_amdgcn_wave_barrier();
    fix_step4(lane, tw, z, prow);
    __builtin_amdgcn_wave_barrier();
}

// ballot of wave_phase4's result -> one bit per frame of the unit (LANES lanes per frame)
template <int LANES, int FRAMES>
__device__ __forceinline__ unsigned frame_mask(uint64_t any) {
    unsigned m = 0;
#pragma unroll
    for (int f = 0; f < FRAMES; ++f) m |= ((any >> (LANES * f)) & ((1ull << LANES) - 1)) ? (1u << f) : 0u;
    return m;
}

// The frames `mask` of a unit: their f64 power rows one after the other, then the kernel's own phases 3-4 once for all of them.
// The f32 kernels do not call this inside their unit loop -- with the f64 code in the loop body the register allocator gives the
// hot path 5 % (a call) to 40 % (inlined) away -- but note the unit (FixSink::list) and come back to it when their run is done.
// Six frames x ten lanes.
template <int NSLOTS, class Lens, bool LAYOUT>
__device__ __forceinline__ unsigned six_fix_unit(unsigned mask, int lane, int hop, int n_mels, const MelSlots &ms, const float *blob, float *slice,
                                          const FixSink &fix, const FixTw &tw, const float *src, float *out_tile, long long row_w,
                                          int *ext = nullptr /* LAYOUT: the unit's record in BatchDesc::d_unit_ext, or nullptr */) {
    const int fl = lane / kSixLanes, j = lane - fl * kSixLanes;
    const bool in = lane < kSixFrames * kSixLanes;
    const int *starts = reinterpret_cast<const int *>(blob + SixBlob::kMelStart) + j;
    {
        FixSamples smp;
        fix_load_samples(lane, src + (__builtin_ctz(mask)) * hop, smp);
        for (unsigned rest = mask; rest;) {                                                       // wave-uniform
            const int f = __builtin_ctz(rest);
            rest &= rest - 1;
            fix_power_row(lane, smp, rest ? src + __builtin_ctz(rest) * hop : nullptr, fix.tab, tw, slice, kSixFixOff, slice + f * SixLayout::kPStride);
        }
    }
    // one pass of phases 3-4 over all the recomputed frames of the unit (each has its own power row)
    {
        const bool act = in && ((mask >> fl) & 1u);
        int st[NSLOTS];
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * kSixLanes];       // lanes 60..63 read valid entries too
        float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS], vals[NSLOTS];
        six_phase3_sums<NSLOTS, Lens>(fl, j, act, ms, blob, slice, st, rise, fprev);
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
        six_phase3_finish<NSLOTS>(fl, j, act, n_mels, rise, fnext, slice, vals);
        __builtin_amdgcn_wave_barrier();
        int kmin = 0x7fffffff, kmax = 0;
        six_phase4<NSLOTS, LAYOUT, false, LAYOUT>(fl, j, act, act, n_mels, slice, vals, out_tile, row_w, &kmin, &kmax);
        __builtin_amdgcn_wave_barrier();
        if (LAYOUT && ext) unit_ext_merge(ext, lane, kmin, kmax);
    }
    return static_cast<unsigned>(__builtin_popcount(mask));      // frames recomputed (the caller adds them up: one atomic per wave, not per
                                                                  // frame -- a million atomics on one address took 10 ms)
}

// The same for the five-frame kernels (12 lanes per frame in phases 3-4).
template <int NSLOTS, class Lens, bool LAYOUT>
__device__ __forceinline__ unsigned wave_fix_unit(unsigned mask, int lane, int hop, int n_mels, const MelSlots &ms, const float *blob, float *slice,
                                           const FixSink &fix, const FixTw &tw, const float *src, float *out_tile, long long row_w,
                                           int *ext = nullptr) {
    const int fl3 = lane / 12, j3 = lane - fl3 * 12;
    const bool in3 = lane < kFPW * 12;
    const int *starts = reinterpret_cast<const int *>(blob + FastBlob::kMelStart) + j3;
    {
        FixSamples smp;
        fix_load_samples(lane, src + (__builtin_ctz(mask)) * hop, smp);
        for (unsigned rest = mask; rest;) {                                                       // wave-uniform
            const int f = __builtin_ctz(rest);
            rest &= rest - 1;
            fix_power_row(lane, smp, rest ? src + __builtin_ctz(rest) * hop : nullptr, fix.tab, tw, slice, kWaveFixOff, slice + f * WaveLayout::kPStride);
        }
    }
    {
        const bool act3 = in3 && ((mask >> fl3) & 1u);
        int st[NSLOTS];
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * 12];       // lanes 60..63 (j3 = 0..3 of a sixth frame) read valid entries too
        float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS], vals[NSLOTS];
        wave_phase3i_sums<NSLOTS, Lens>(fl3, j3, act3, ms, blob, slice, st, rise, fprev);
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
        wave_phase3i_finish<NSLOTS>(fl3, j3, act3, n_mels, rise, fnext, slice, vals);
        __builtin_amdgcn_wave_barrier();
        int kmin = 0x7fffffff, kmax = 0;
        wave_phase4<NSLOTS, LAYOUT, false, LAYOUT>(fl3, j3, act3, act3, n_mels, slice, vals, out_tile, row_w, &kmin, &kmax);
        __builtin_amdgcn_wave_barrier();
        if (LAYOUT && ext) unit_ext_merge(ext, lane, kmin, kmax);
    }
    return static_cast<unsigned>(__builtin_popcount(mask));      // frames recomputed (the caller adds them up: one atomic per wave, not per
                                                                  // frame -- a million atomics on one address took 10 ms)
}

// Sub-group barrier of the mel-major stores (BatchDesc::sync_rounds = gsize + 16 * across, gsize in {2, 4, 8}): only the
// gsize waves that hold adjacent units of a round wait for each other (LDS arrival counters, the waiting wave at priority 0
// polling with s_sleep), so the pieces of a 32-byte sector reach L2 together while the other waves of the workgroup keep
// their phases apart.  "across": the group is made of waves WAVES / gsize apart and the units of a round are dealt so that
// it still holds adjacent ones.  sync_rounds == 1 is the plain workgroup barrier, 0 none.
template <int WAVES>
struct RoundSync {
    int gsize, g, slot;
    unsigned round = 0;
    unsigned *arrive;
    __device__ __forceinline__ RoundSync(int mode, int wave, unsigned *counters) : arrive(counters) {
        gsize = mode & 15;
        if (gsize > WAVES) gsize = 1;                 // a group cannot be larger than the workgroup: plain barrier
        const int across = mode >> 4;
        const int ngroups = gsize > 1 ? WAVES / gsize : 1;
        g = gsize > 1 ? (across ? wave % ngroups : wave / gsize) : 0;
        slot = (gsize > 1 && across) ? g * gsize + wave / ngroups : wave;
    }
    // before the stores of a round
    template <int RESTORE_PRIO>
    __device__ __forceinline__ void before_stores(int lane) {
        if (gsize <= 1) return;
        ++round;
        __builtin_amdgcn_s_setprio(0);
        if (lane == 0) {
            __hip_atomic_fetch_add(arrive + g, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned target = round * (unsigned)gsize;
            while (__hip_atomic_load(arrive + g, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(RESTORE_PRIO);
    }
    // at the end of a round
    __device__ __forceinline__ void after_round() const {
        if (gsize == 1) __syncthreads();
    }
};

constexpr int kWaveWaves = 8;     // waves per workgroup of the 5-frame kernels (two workgroups per CU)

// ---- 5 frames per wave (81..131 mels, and every bank the six-frame tables do not cover) ----------------------------
// Padded and/or mel-major output (interleave_frames, BatchDesc::out_width / mel_major): the units are dealt round-robin
// and walked in workgroup-uniform rounds (a wave without a unit idles through the round) so that the mel-major store can
// re-align the waves that hold adjacent units once per round.
template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kWaveWaves * 64, 4) void whisper400_wave_kernel(const FastParams p) {
    constexpr int WAVES = kWaveWaves;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *blob = lds;
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_len; i += WAVES * 64) blob[i] = p.d_blob[i];
    unsigned *arrive = reinterpret_cast<unsigned *>(blob + p.blob_len + WAVES * p.slice_floats);   // RoundSync counters, then the vote's four words
    if (tid < WAVES + 4) arrive[tid] = 0;
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float *slice = blob + p.blob_len + wave * p.slice_floats;
    const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
    const bool in = lane < kFPW * kMelJobs;
    int uoff, voff;
    WaveLayout::row_offsets(j, uoff, voff);
    const int n_mels = Lens::kStatic ? Lens::kMels : p.n_mels;     // compile-time for the two Whisper banks
    const int fl3 = lane / 12, j3 = lane - fl3 * 12;               // 12 lanes per frame in phases 3-4
    const bool in3 = lane < kFPW * 12;
    const int *starts = reinterpret_cast<const int *>(blob + FastBlob::kMelStart) + j3;
    const bool guard = p.fix.tab != nullptr;

    RoundSync<WAVES> rs(p.b.sync_rounds, wave, arrive);
    // this wave's notes: one slot per round, rounds * (its rank among all waves) onwards
    const uint64_t rounds = (p.b.n_units + (uint64_t)gridDim.x * WAVES - 1) / ((uint64_t)gridDim.x * WAVES);
    uint64_t *notes = guard ? p.fix.list + ((uint64_t)xcd_logical_block() * WAVES + rs.slot) * rounds : nullptr;
    unsigned noted = 0;
    int nv = 0;
    auto round = [&](uint64_t first) __attribute__((always_inline)) -> uint64_t {
        const uint64_t unit = first + rs.slot;
        const bool have = unit < p.b.n_units;
        const UnitLoc loc = locate_unit(p.b, have ? unit : first);
        const uint64_t f0 = loc.unit * kFPW;
        const uint64_t left = (have && f0 < loc.frames) ? loc.frames - f0 : 0;
        nv = left < (uint64_t)kFPW ? (int)left : kFPW;
        // columns this unit stores: the clip's frames plus, for padded layouts, zero columns up to out_width
        const uint64_t width = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
        const uint64_t wleft = have ? width - f0 : 0;
        const int ns = wleft < (uint64_t)kFPW ? (int)wleft : kFPW;
        const float *src = loc.pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv;
        const bool act3 = in3 && fl3 < nv;
        wave_phase1(fl, j, act && j < kFftJobs, p.hop, blob, src, slice);
        __builtin_amdgcn_wave_barrier();
        wave_phase2(fl, j, act, blob, slice, uoff, voff);
        __builtin_amdgcn_wave_barrier();
        float vals[NSLOTS];
        {
            // per-lane start bins: re-read every unit (NSLOTS LDS words) rather than held in registers across the loop
            int st[NSLOTS];
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * 12];       // lanes 60..63 (j3 = 0..3 of a sixth frame) read valid entries too
            float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
            wave_phase3i_sums<NSLOTS, Lens>(fl3, j3, act3, p.slots, blob, slice, st, rise, fprev);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
            wave_phase3i_finish<NSLOTS>(fl3, j3, act3, n_mels, rise, fnext, slice, vals);
        }
        __builtin_amdgcn_wave_barrier();
        rs.template before_stores<0>(lane);
        float *out_tile = p.b.mel_major ? loc.out + f0 : loc.out + f0 * (uint64_t)n_mels;
        const long long row_w = p.b.mel_major ? (long long)width : 0;
        int kmin = 0x7fffffff, kmax = 0;
        const bool flag = wave_phase4<NSLOTS, true, true, true>(fl3, j3, in3 && fl3 < ns, act3, n_mels, slice, vals, out_tile, row_w, &kmin, &kmax);
        __builtin_amdgcn_wave_barrier();
        unsigned redo = 0;                  // frames of this unit that the tail recomputes
        uint64_t any = 0;
        if (guard) {
            any = __builtin_amdgcn_ballot_w64(flag);
            if (any != 0) {
                redo = frame_mask<12, kFPW>(any);
                if (lane == 0) notes[noted] = (unit << 8) | redo;
                ++noted;
            }
        }
        if (p.b.d_unit_ext && have) {       // wave-uniform; a frame that is recomputed reports its extremes then
            if ((redo >> fl3) & 1u) { kmin = 0x7fffffff; kmax = 0; }
            unit_ext_store(p.b.d_unit_ext + 2 * unit, lane, kmin, kmax);
        }
        // mel-major: the 8 waves hold 8 adjacent 20-byte pieces of every row; kept in step, the pieces of a cache line
        // reach L2 within microseconds of each other and leave it as one full line
        rs.after_round();
        return any;
    };
    uint64_t first = (uint64_t)xcd_logical_block() * WAVES;
    const uint64_t step = (uint64_t)gridDim.x * WAVES;
    if (guard && p.fix.vote != nullptr) {                              // AUTO's vote, as in whisper400_six_kernel
        unsigned *votew = arrive + WAVES;
        bool sample = blockIdx.x < p.fix.vote_groups;
        unsigned verdict = 0, polled = 0;
        if (sample && first >= p.b.n_units) {
            vote_cast(p.fix, votew, WAVES, lane, 0, 0);
            sample = false;
        }
        for (; first < p.b.n_units && verdict == 0; first += step) {
            const uint64_t any = round(first);
            if (sample) {
                vote_cast(p.fix, votew, WAVES, lane, static_cast<unsigned>(__builtin_popcount(frame_mask<12, kFPW>(any))), static_cast<unsigned>(nv));
                sample = false;
            }
            (void)vote_check(p.fix, votew, ++polled, wave);
            __syncthreads();
            verdict = __builtin_amdgcn_readfirstlane(__hip_atomic_load(votew + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            __syncthreads();
        }
        if (verdict & kVoteHeavy) {
            guard_wave_done(p.fix, arrive + WAVES - 2, WAVES, lane, 0);
            return;
        }
    }
    for (; first < p.b.n_units; first += step) round(first);
    // the units whose frames tripped the precision guard, again, in f64 (no barrier of the rounds involved any more)
    unsigned redone = 0;
    FixTw tw;
    // The tail derives its lane constants (frame slot, start bins, row offsets) afresh from an opaque copy of `lane`: as the
    // SAME values as the hot loop's they stayed live across the tail's register-hungry f64 code, and the allocator spilled them for
    // the loop as well (the mel-major kernel reloaded one from scratch eight times per unit: 0.37 -> 0.45 ms).
    int tlane = lane;
    float *tslice = slice;
    asm volatile("" : "+v"(tlane));
    if (noted) fix_load_tw(tlane, p.fix.tab, tw);
    for (unsigned k = 0; k < noted; ++k) {
        uint64_t e = 0;
        if (lane == 0) e = notes[k];
        e = scalar64(e);
        const uint64_t unit = e >> 8;
        const UnitLoc loc = locate_unit(p.b, unit);
        const uint64_t f0 = loc.unit * kFPW;
        const uint64_t width = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
        float *out_tile = p.b.mel_major ? loc.out + f0 : loc.out + f0 * (uint64_t)n_mels;
        redone += wave_fix_unit<NSLOTS, Lens, true>(static_cast<unsigned>(e & 0xff), tlane, p.hop, n_mels, p.slots, blob, tslice, p.fix, tw,
                                          loc.pcm + f0 * (uint64_t)p.hop, out_tile, p.b.mel_major ? (long long)width : 0,
                                          p.b.d_unit_ext ? p.b.d_unit_ext + 2 * unit : nullptr);
    }
    guard_wave_done(p.fix, arrive + WAVES - 2, WAVES, lane, redone);
}

// ------------------------------------------------------------------------------------
// Six frames per wavefront (phases in whisper_six.hpp), one 16-wave workgroup per CU.
// ------------------------------------------------------------------------------------
// Padded and/or mel-major output: workgroup-uniform rounds like whisper400_wave_kernel.
template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kSixWaves * 64, 4) void whisper400_six_kernel(const FastParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *blob = lds;
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_len; i += kSixWaves * 64) blob[i] = p.d_blob[i];
    // arrival counters of the sub-group barrier (mel-major stores), behind the last slice
    unsigned *arrive = reinterpret_cast<unsigned *>(blob + p.blob_len + kSixWaves * SixLayout::slice_floats());
    if (tid < kSixWaves + 4) arrive[tid] = 0;                          // + the vote's four words
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float *slice = blob + p.blob_len + wave * SixLayout::slice_floats();
    const int fl = lane / kSixLanes, j = lane - fl * kSixLanes;
    const bool in = lane < kSixFrames * kSixLanes;
    int uoff, voff;
    SixLayout::row_offsets(j, uoff, voff);
    const int n_mels = Lens::kStatic ? Lens::kMels : p.n_mels;
    const int *starts = reinterpret_cast<const int *>(blob + SixBlob::kMelStart) + j;
    const bool guard = p.fix.tab != nullptr;
    RoundSync<kSixWaves> rs(p.b.sync_rounds, wave, arrive);
    const uint64_t rounds = (p.b.n_units + (uint64_t)gridDim.x * kSixWaves - 1) / ((uint64_t)gridDim.x * kSixWaves);
    uint64_t *notes = guard ? p.fix.list + ((uint64_t)xcd_logical_block() * kSixWaves + rs.slot) * rounds : nullptr;
    unsigned noted = 0;
    int nv = 0;
    // one round of the workgroup: this wave's unit through phases 1-4; returns the lanes whose guard tripped
    auto round = [&](uint64_t first) __attribute__((always_inline)) -> uint64_t {
        const uint64_t unit = first + rs.slot;
        const bool have = unit < p.b.n_units;
        const UnitLoc loc = locate_unit(p.b, have ? unit : first);
        const uint64_t f0 = loc.unit * kSixFrames;
        const uint64_t left = (have && f0 < loc.frames) ? loc.frames - f0 : 0;
        nv = left < (uint64_t)kSixFrames ? (int)left : kSixFrames;
        // columns this unit stores: the clip's frames plus, for padded layouts, zero columns up to out_width
        const uint64_t width = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
        const uint64_t wleft = have ? width - f0 : 0;
        const int ns = wleft < (uint64_t)kSixFrames ? (int)wleft : kSixFrames;
        const float *src = loc.pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv;
        MS_PRIO(0);
        six_phase1(fl, j, act, p.hop, blob, src, slice);
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(1);
        six_phase2(fl, j, act, blob, slice, uoff, voff);
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(2);
        float vals[NSLOTS];
        {
            // per-lane start bins: re-read every unit (9 LDS words) rather than held in registers across the loop
            int st[NSLOTS];
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * kSixLanes];       // lanes 60..63 (j = 0..3 of a seventh frame) read valid entries too
            float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
            six_phase3_sums<NSLOTS, Lens>(fl, j, act, p.slots, blob, slice, st, rise, fprev);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
            six_phase3_finish<NSLOTS>(fl, j, act, n_mels, rise, fnext, slice, vals);
        }
        __builtin_amdgcn_wave_barrier();
        rs.template before_stores<3>(lane);
        float *out_tile = p.b.mel_major ? loc.out + f0 : loc.out + f0 * (uint64_t)n_mels;
        const long long row_w = p.b.mel_major ? (long long)width : 0;
        int kmin = 0x7fffffff, kmax = 0;
        const bool flag = six_phase4<NSLOTS, true, true, true>(fl, j, in && fl < ns, act, n_mels, slice, vals, out_tile, row_w, &kmin, &kmax);
        __builtin_amdgcn_wave_barrier();
        unsigned redo = 0;                  // frames of this unit that the tail recomputes
        uint64_t any = 0;
        if (guard) {
            any = __builtin_amdgcn_ballot_w64(flag);
            if (any != 0) {
                redo = frame_mask<kSixLanes, kSixFrames>(any);
                if (lane == 0) notes[noted] = (unit << 8) | redo;
                ++noted;
            }
        }
        if (p.b.d_unit_ext && have) {       // wave-uniform; a frame that is recomputed reports its extremes then
            if ((redo >> fl) & 1u) { kmin = 0x7fffffff; kmax = 0; }
            unit_ext_store(p.b.d_unit_ext + 2 * unit, lane, kmin, kmax);
        }
        rs.after_round();
        return any;
    };
    uint64_t first = (uint64_t)xcd_logical_block() * kSixWaves;
    const uint64_t step = (uint64_t)gridDim.x * kSixWaves;
    // AUTO's vote (FixSink::vote), layouts: the sample is the first round of the first vote_groups workgroups, and the workgroup
    // leaves TOGETHER -- its waves wait for each other in RoundSync, so the verdict is read behind a workgroup barrier.  In a loop of
    // its own, like the run-per-wave kernels' (the same code inside the round loop proper cost that loop 19 %).
    if (guard && p.fix.vote != nullptr) {
        unsigned *votew = arrive + kSixWaves;                          // vote_cast's three words, the workgroup's copy of the verdict
        bool sample = blockIdx.x < p.fix.vote_groups;
        unsigned verdict = 0, polled = 0;
        if (sample && first >= p.b.n_units) {                          // a workgroup of the grid's round-up to the 8 XCDs: it still has to be counted
            vote_cast(p.fix, votew, kSixWaves, lane, 0, 0);
            sample = false;
        }
        for (; first < p.b.n_units && verdict == 0; first += step) {
            const uint64_t any = round(first);
            if (sample) {
                vote_cast(p.fix, votew, kSixWaves, lane, static_cast<unsigned>(__builtin_popcount(frame_mask<kSixLanes, kSixFrames>(any))), static_cast<unsigned>(nv));
                sample = false;
            }
            (void)vote_check(p.fix, votew, ++polled, wave);
            __syncthreads();
            verdict = __builtin_amdgcn_readfirstlane(__hip_atomic_load(votew + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            __syncthreads();                                           // nobody publishes a verdict between two waves' reads of it
        }
        if (verdict & kVoteHeavy) {                                    // the f64 kernel behind this launch computes the whole batch
            guard_wave_done(p.fix, arrive + kSixWaves - 2, kSixWaves, lane, 0);
            return;
        }
    }
    for (; first < p.b.n_units; first += step) round(first);
    unsigned redone = 0;
    FixTw tw;
    // The tail derives its lane constants (frame slot, start bins, row offsets) afresh from an opaque copy of `lane`: as the
    // SAME values as the hot loop's they stayed live across the tail's register-hungry f64 code, and the allocator spilled them for
    // the loop as well (the mel-major kernel reloaded one from scratch eight times per unit: 0.37 -> 0.45 ms).
    int tlane = lane;
    float *tslice = slice;
    asm volatile("" : "+v"(tlane));
    if (noted) fix_load_tw(tlane, p.fix.tab, tw);
    for (unsigned k = 0; k < noted; ++k) {
        uint64_t e = 0;
        if (lane == 0) e = notes[k];
        e = scalar64(e);
        const uint64_t unit = e >> 8;
        const UnitLoc loc = locate_unit(p.b, unit);
        const uint64_t f0 = loc.unit * kSixFrames;
        const uint64_t width = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
        float *out_tile = p.b.mel_major ? loc.out + f0 : loc.out + f0 * (uint64_t)n_mels;
        redone += six_fix_unit<NSLOTS, Lens, true>(static_cast<unsigned>(e & 0xff), tlane, p.hop, n_mels, p.slots, blob, tslice, p.fix, tw,
                                         loc.pcm + f0 * (uint64_t)p.hop, out_tile, p.b.mel_major ? (long long)width : 0,
                                         p.b.d_unit_ext ? p.b.d_unit_ext + 2 * unit : nullptr);
    }
    guard_wave_done(p.fix, arrive + kSixWaves - 2, kSixWaves, lane, redone);
}

// Plain [frame][mel] output, uniform and ragged batches, on the six-frame build -- the default kernel of the bench workload.
// A wave takes a contiguous run of units: it locates its first unit once and from then on only steps to the next clip when
// the run crosses a clip end; the clip record lives in scalar registers.  Against a round-robin deal (which the padded /
// mel-major layouts keep, their stores want adjacent units in adjacent waves -- a run-per-wave build of the mel-major store was
// measured: 0.407 ms against 0.350 ms for the rounds with the sub-group barrier): ragged batches lose the two dependent
// look-ups in front of every unit's PCM loads (-6 %), uniform ones the 64-bit division per unit and a wave re-reads its own
// frame-tail halo (cfg2 -1.6 %, 8192 x 30 s -1.7 %).  The unit body is the same.

// A wave's contiguous run of units of a ragged batch and the clip it is in (everything wave-uniform, in scalar registers).
struct ClipRun {
    uint64_t unit, end, c_start, c_end, c_frames;
    const float *c_pcm;
    float *c_out;
    uint32_t clip;
    __device__ __forceinline__ void load_clip(const BatchDesc &b) {
        if (b.d_unit_prefix == nullptr) {                 // uniform batch: arithmetic
            c_start = (uint64_t)clip * b.units_per_clip;
            c_end = c_start + b.units_per_clip;
            c_frames = b.frames_per_clip;
            c_pcm = b.pcm + (uint64_t)clip * b.clip_stride;
            c_out = b.out + (uint64_t)clip * b.out_stride;
            return;
        }
        c_start = scalar64(b.d_unit_prefix[clip]);
        c_frames = scalar64(b.d_frames[clip]);
        c_pcm = b.pcm + scalar64(b.d_off[clip]);
        c_out = b.out + scalar64(b.d_out_off[clip]);
    }
    // false: this wave has no units
    __device__ __forceinline__ bool init(const BatchDesc &b, uint64_t wave_id, uint64_t waves) {
        const uint64_t nu = b.d_n_units ? ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(*b.d_n_units >> 32)) << 32 |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)*b.d_n_units)) : b.n_units;
        const uint64_t run = (nu + waves - 1) / waves;
        unit = wave_id * run;
        end = unit + run < nu ? unit + run : nu;
        if (unit >= end) return false;
        if (b.d_unit_prefix == nullptr) {
            clip = static_cast<uint32_t>(unit / b.units_per_clip);
            load_clip(b);
            return true;
        }
        clip = __builtin_amdgcn_readfirstlane(b.d_unit_block[unit / kUnitBlock]);
        c_end = scalar64(b.d_unit_prefix[clip + 1]);
        while (c_end <= unit) { ++clip; c_end = scalar64(b.d_unit_prefix[clip + 1]); }     // prefix[n_clips] = n_units > unit
        load_clip(b);
        return true;
    }
    __device__ __forceinline__ UnitLoc loc() const {
        UnitLoc r;
        r.unit = unit - c_start; r.pcm = c_pcm; r.out = c_out; r.frames = c_frames; r.clip = clip;
        return r;
    }
    // before each unit: the run may have entered the next clip that has frames
    __device__ __forceinline__ void enter(const BatchDesc &b) {
        if (unit >= c_end) {
            if (b.d_unit_prefix == nullptr) ++clip;
            else do { ++clip; c_end = scalar64(b.d_unit_prefix[clip + 1]); } while (c_end <= unit);
            load_clip(b);
        }
    }
};

template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kSixWaves * 64, 4) void whisper400_six_runs_kernel(const FastParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *blob = lds;
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_len; i += kSixWaves * 64) blob[i] = p.d_blob[i];
    unsigned *wg_done = reinterpret_cast<unsigned *>(blob + p.blob_len + kSixWaves * SixLayout::slice_floats());   // guard_wave_done's two words,
    if (tid < 6) wg_done[tid] = 0;                                                                                 // vote_cast's three, vote_check's one
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float *slice = blob + p.blob_len + wave * SixLayout::slice_floats();
    const int fl = lane / kSixLanes, j = lane - fl * kSixLanes;
    const bool in = lane < kSixFrames * kSixLanes;
    int uoff, voff;
    SixLayout::row_offsets(j, uoff, voff);
    const int n_mels = Lens::kStatic ? Lens::kMels : p.n_mels;
    const int *starts = reinterpret_cast<const int *>(blob + SixBlob::kMelStart) + j;
    const bool guard = p.fix.tab != nullptr;

    ClipRun cr;
    if (!cr.init(p.b, (uint64_t)xcd_logical_block() * kSixWaves + wave, (uint64_t)gridDim.x * kSixWaves)) {
        if (guard && p.fix.vote != nullptr && blockIdx.x < p.fix.vote_groups) vote_cast(p.fix, wg_done + 2, kSixWaves, lane, 0, 0);
        guard_wave_done(p.fix, wg_done, kSixWaves, lane, 0);
        return;
    }
    uint64_t *notes = guard ? p.fix.list + cr.unit : nullptr;        // this wave's notes: the entries its own run indexes
    unsigned noted = 0;
    int nv = 0;
    // one work unit: phases 1-4 and the note for the tail; returns the lanes whose guard tripped.  `pre` (a std::true_type in the
    // vote's loop): between the phases the wave looks at the verdict and leaves the unit on "heavy" -- the batch's f64 launch is
    // waiting for this one to drain, a unit is 7 us long (stand-down 30 -> ~20 us); the unit loop proper is instantiated without it
    unsigned verdict = 0, polled = 0;
    bool may_leave = true;
    auto unit = [&](auto pre) __attribute__((always_inline)) -> uint64_t {
        constexpr bool kPre = decltype(pre)::value;
        cr.enter(p.b);
        const uint64_t f0 = (cr.unit - cr.c_start) * kSixFrames;
        const uint64_t left = cr.c_frames - f0;
        nv = left < (uint64_t)kSixFrames ? (int)left : kSixFrames;
        const float *src = cr.c_pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv;
        MS_PRIO(0);
        six_phase1(fl, j, act, p.hop, blob, src, slice);
        __builtin_amdgcn_wave_barrier();
        if (kPre && may_leave) {
            verdict = vote_check(p.fix, wg_done + 2, ++polled, wave);
            if (verdict & kVoteHeavy) return 0;
        }
        MS_PRIO(1);
        six_phase2(fl, j, act, blob, slice, uoff, voff);
        __builtin_amdgcn_wave_barrier();
        if (kPre && may_leave) {
            verdict = vote_check(p.fix, wg_done + 2, ++polled, wave);
            if (verdict & kVoteHeavy) return 0;
        }
        MS_PRIO(2);
        float vals[NSLOTS];
        {
            int st[NSLOTS];
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * kSixLanes];       // lanes 60..63 (j = 0..3 of a seventh frame) read valid entries too
            float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
            six_phase3_sums<NSLOTS, Lens>(fl, j, act, p.slots, blob, slice, st, rise, fprev);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
            six_phase3_finish<NSLOTS>(fl, j, act, n_mels, rise, fnext, slice, vals);
        }
        __builtin_amdgcn_wave_barrier();
        float *out_tile = cr.c_out + f0 * (uint64_t)n_mels;
        const bool flag = six_phase4<NSLOTS, false, true>(fl, j, act, act, n_mels, slice, vals, out_tile, 0);
        __builtin_amdgcn_wave_barrier();
        uint64_t any = 0;
        if (guard) {
            any = __builtin_amdgcn_ballot_w64(flag);
            if (any != 0) {
                if (lane == 0) notes[noted] = (cr.unit << 8) | frame_mask<kSixLanes, kSixFrames>(any);
                ++noted;
            }
        }
        return any;
    };
    // AUTO's vote (FixSink::vote).  The first units of a voting launch run in a loop of their own until the verdict is known: the
    // same code in the unit loop proper -- a handful of scalar branches that are never taken after the second unit -- cost that loop
    // 19 % (same-box A/B, round 4: the register allocator and the scheduler see one more loop-carried state and two more exits).
    if (guard && p.fix.vote != nullptr) {
        bool sample = blockIdx.x < p.fix.vote_groups;                  // the first unit of every wave of the first vote_groups workgroups is the sample
        may_leave = !sample;                                           // a sampling wave finishes its first unit: the tally waits for it
        for (; cr.unit < cr.end && verdict == 0; ++cr.unit) {
            const uint64_t any = unit(std::true_type{});
            if (verdict & kVoteHeavy) break;
            if (sample) {
                vote_cast(p.fix, wg_done + 2, kSixWaves, lane, static_cast<unsigned>(__builtin_popcount(frame_mask<kSixLanes, kSixFrames>(any))), static_cast<unsigned>(nv));
                sample = false;
                may_leave = true;
            }
            verdict = vote_check(p.fix, wg_done + 2, ++polled, wave);
        }
        if (verdict == 0) verdict = vote_poll(p.fix);                  // a run shorter than the vote
        if (verdict & kVoteHeavy) {                                    // the f64 kernel behind this launch computes the whole batch
            guard_wave_done(p.fix, wg_done, kSixWaves, lane, 0);
            return;
        }
    }
    for (; cr.unit < cr.end; ++cr.unit) unit(std::false_type{});
    // the units whose frames tripped the precision guard, again, in f64
    unsigned redone = 0;
    FixTw tw;
    // The tail derives its lane constants (frame slot, start bins, row offsets) afresh from an opaque copy of `lane`: as the
    // SAME values as the hot loop's they stayed live across the tail's register-hungry f64 code, and the allocator spilled them for
    // the loop as well (the mel-major kernel reloaded one from scratch eight times per unit: 0.37 -> 0.45 ms).
    int tlane = lane;
    float *tslice = slice;
    asm volatile("" : "+v"(tlane));
    if (noted) fix_load_tw(tlane, p.fix.tab, tw);
    for (unsigned k = 0; k < noted; ++k) {
        uint64_t e = 0;
        if (lane == 0) e = notes[k];
        e = scalar64(e);
        const UnitLoc loc = locate_unit(p.b, e >> 8);
        const uint64_t f0 = loc.unit * kSixFrames;
        redone += six_fix_unit<NSLOTS, Lens, false>(static_cast<unsigned>(e & 0xff), tlane, p.hop, n_mels, p.slots, blob, tslice, p.fix, tw,
                                          loc.pcm + f0 * (uint64_t)p.hop, loc.out + f0 * (uint64_t)n_mels, 0);
    }
    guard_wave_done(p.fix, wg_done, kSixWaves, lane, redone);
}

// The same for the 5-frame kernel (81..131 mels): interval mel scheme, direct PCM reads, 8-wave workgroups.
template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kWaveWaves * 64, 4) void whisper400_wave_runs_kernel(const FastParams p) {
    constexpr int WAVES = kWaveWaves;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *blob = lds;
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_len; i += WAVES * 64) blob[i] = p.d_blob[i];
    unsigned *wg_done = reinterpret_cast<unsigned *>(blob + p.blob_len + WAVES * p.slice_floats);   // guard_wave_done's two words, vote_cast's three
    if (tid < 6) wg_done[tid] = 0;
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float *slice = blob + p.blob_len + wave * p.slice_floats;
    const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
    const bool in = lane < kFPW * kMelJobs;
    int uoff, voff;
    WaveLayout::row_offsets(j, uoff, voff);
    const int n_mels = Lens::kStatic ? Lens::kMels : p.n_mels;
    const int fl3 = lane / 12, j3 = lane - fl3 * 12;
    const bool in3 = lane < kFPW * 12;
    const int *starts = reinterpret_cast<const int *>(blob + FastBlob::kMelStart) + j3;
    const bool guard = p.fix.tab != nullptr;
    ClipRun cr;
    if (!cr.init(p.b, (uint64_t)xcd_logical_block() * WAVES + wave, (uint64_t)gridDim.x * WAVES)) {
        if (guard && p.fix.vote != nullptr && blockIdx.x < p.fix.vote_groups) vote_cast(p.fix, wg_done + 2, WAVES, lane, 0, 0);
        guard_wave_done(p.fix, wg_done, WAVES, lane, 0);
        return;
    }
    uint64_t *notes = guard ? p.fix.list + cr.unit : nullptr;
    unsigned noted = 0;
    int nv = 0;
    unsigned verdict = 0, polled = 0;
    bool may_leave = true;
    auto unit = [&](auto pre) __attribute__((always_inline)) -> uint64_t {        // pre: see whisper400_six_runs_kernel
        constexpr bool kPre = decltype(pre)::value;
        cr.enter(p.b);
        const uint64_t f0 = (cr.unit - cr.c_start) * kFPW;
        const uint64_t left = cr.c_frames - f0;
        nv = left < (uint64_t)kFPW ? (int)left : kFPW;
        const float *src = cr.c_pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv, act3 = in3 && fl3 < nv;
        MS_PRIO(0);
        wave_phase1(fl, j, act && j < kFftJobs, p.hop, blob, src, slice);
        __builtin_amdgcn_wave_barrier();
        if (kPre && may_leave) {
            verdict = vote_check(p.fix, wg_done + 2, ++polled, wave);
            if (verdict & kVoteHeavy) return 0;
        }
        MS_PRIO(1);
        wave_phase2(fl, j, act, blob, slice, uoff, voff);
        __builtin_amdgcn_wave_barrier();
        if (kPre && may_leave) {
            verdict = vote_check(p.fix, wg_done + 2, ++polled, wave);
            if (verdict & kVoteHeavy) return 0;
        }
        MS_PRIO(2);
        float vals[NSLOTS];
        {
            int st[NSLOTS];
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * 12];       // lanes 60..63 (j3 = 0..3 of a sixth frame) read valid entries too
            float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
            wave_phase3i_sums<NSLOTS, Lens>(fl3, j3, act3, p.slots, blob, slice, st, rise, fprev);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
            wave_phase3i_finish<NSLOTS>(fl3, j3, act3, n_mels, rise, fnext, slice, vals);
        }
        __builtin_amdgcn_wave_barrier();
        float *out_tile = cr.c_out + f0 * (uint64_t)n_mels;
        const bool flag = wave_phase4<NSLOTS, false, true>(fl3, j3, act3, act3, n_mels, slice, vals, out_tile, 0);
        __builtin_amdgcn_wave_barrier();
        uint64_t any = 0;
        if (guard) {
            any = __builtin_amdgcn_ballot_w64(flag);
            if (any != 0) {
                if (lane == 0) notes[noted] = (cr.unit << 8) | frame_mask<12, kFPW>(any);
                ++noted;
            }
        }
        return any;
    };
    if (guard && p.fix.vote != nullptr) {                              // AUTO's vote, as in whisper400_six_runs_kernel
        bool sample = blockIdx.x < p.fix.vote_groups;
        may_leave = !sample;
        for (; cr.unit < cr.end && verdict == 0; ++cr.unit) {
            const uint64_t any = unit(std::true_type{});
            if (verdict & kVoteHeavy) break;
            if (sample) {
                vote_cast(p.fix, wg_done + 2, WAVES, lane, static_cast<unsigned>(__builtin_popcount(frame_mask<12, kFPW>(any))), static_cast<unsigned>(nv));
                sample = false;
                may_leave = true;
            }
            verdict = vote_check(p.fix, wg_done + 2, ++polled, wave);
        }
        if (verdict == 0) verdict = vote_poll(p.fix);
        if (verdict & kVoteHeavy) {
            guard_wave_done(p.fix, wg_done, WAVES, lane, 0);
            return;
        }
    }
    for (; cr.unit < cr.end; ++cr.unit) unit(std::false_type{});
    unsigned redone = 0;
    FixTw tw;
    // The tail derives its lane constants (frame slot, start bins, row offsets) afresh from an opaque copy of `lane`: as the
    // SAME values as the hot loop's they stayed live across the tail's register-hungry f64 code, and the allocator spilled them for
    // the loop as well (the mel-major kernel reloaded one from scratch eight times per unit: 0.37 -> 0.45 ms).
    int tlane = lane;
    float *tslice = slice;
    asm volatile("" : "+v"(tlane));
    if (noted) fix_load_tw(tlane, p.fix.tab, tw);
    for (unsigned k = 0; k < noted; ++k) {
        uint64_t e = 0;
        if (lane == 0) e = notes[k];
        e = scalar64(e);
        const UnitLoc loc = locate_unit(p.b, e >> 8);
        const uint64_t f0 = loc.unit * kFPW;
        redone += wave_fix_unit<NSLOTS, Lens, false>(static_cast<unsigned>(e & 0xff), tlane, p.hop, n_mels, p.slots, blob, tslice, p.fix, tw,
                                           loc.pcm + f0 * (uint64_t)p.hop, loc.out + f0 * (uint64_t)n_mels, 0);
    }
    guard_wave_done(p.fix, wg_done, WAVES, lane, redone);
}

// ------------------------------------------------------------------------------------
// "Precise" fused Whisper kernels: f64 FFT (whisper_wave_f64.hpp), f32 interval mel + normalisation.
// LDS words: [f64 tables][f32 mel section][WAVES x slice of 2320 doubles].
// ------------------------------------------------------------------------------------
struct PreciseParams {
    BatchDesc b;
    const uint32_t *d_blob;
    int blob_words;       // multiple of 4
    int mel_off_words;    // where the f32 mel section (FastBlob::kMelStart.. of the f32 blob) starts
    int hop;
    int n_mels;
    MelSlots slots;       // woff[] as in the f32 blob (float offsets from FastBlob's base)
    FixSink stat;         // MELSPEC_PRECISION_AUTO running this kernel on a whole batch (most of whose frames trip the guard): the frames
                          // that would have tripped it are counted and published like the f32 kernels do (tab and list unused)
    // MODE 2 (AUTO, queued behind the voting f32 launch): runs only when *gate == gate_value -- the f32 launch's "heavy" verdict --
    // and walks the plan of THAT launch, whose units are plan_fpu frames long (6 on the six-frame contexts), in steps of kFPW frames
    const unsigned *gate;
    unsigned gate_value;
    int plan_fpu;
};

constexpr int kPreciseWaves = 8;    // one workgroup per CU

// MODE 1 (runs): plain [frame][mel] output, a contiguous run of units per wave (ClipRun); MODE 0: the padded / mel-major layouts in
// workgroup-uniform rounds (see whisper400_wave_kernel); MODE 2: plain output over the plan of the f32 launch in front of it, gated
// on that launch's vote (PreciseParams::gate) -- a wave takes a contiguous run of THAT plan's units and walks the frames they cover
// five at a time, clip by clip (one partial step per clip segment of a run: 1.6 % at config 2, nothing on long runs).
template <int NSLOTS, class Lens, int MODE>
__global__ __launch_bounds__(kPreciseWaves * 64) void whisper400_precise_kernel(const PreciseParams p) {
    constexpr int WAVES = kPreciseWaves;
    constexpr bool RUNS = MODE != 0, WALK = MODE == 2;
    constexpr bool LAYOUT = !RUNS;
    if (p.gate != nullptr && *p.gate != p.gate_value) return;        // AUTO's second launch and the batch was light: the f32 launch has finished it
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_words; i += WAVES * 64) ldsw[i] = p.d_blob[i];
    unsigned *arrive = ldsw + p.blob_words + WAVES * PreciseLayout::slice_doubles() * 2;   // RoundSync counters
    if (tid < WAVES) arrive[tid] = 0;
    __syncthreads();
    const double *tb = reinterpret_cast<const double *>(ldsw);
    // the shared phase-3 code addresses the mel tables as offsets from the base of the f32 blob
    const float *fblob = reinterpret_cast<const float *>(ldsw + p.mel_off_words) - FastBlob::kMelStart;

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double *rows = reinterpret_cast<double *>(ldsw + p.blob_words) + wave * PreciseLayout::slice_doubles();
    float *slice = reinterpret_cast<float *>(rows);
    const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
    const bool in = lane < kFPW * kMelJobs;
    const int n_mels = Lens::kStatic ? Lens::kMels : p.n_mels;
    const int fl3 = lane / 12, j3 = lane - fl3 * 12;
    const bool in3 = lane < kFPW * 12;
    int st[NSLOTS];
    {
        const int *starts = reinterpret_cast<const int *>(fblob + FastBlob::kMelStart);
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) st[i] = in3 ? starts[i * 12 + j3] : 0;
    }
    RoundSync<WAVES> rs(LAYOUT ? p.b.sync_rounds : 0, wave, arrive);
    ClipRun cr;
    const bool stats = p.stat.acc != nullptr;
    unsigned flagged = 0;
    if (RUNS && !cr.init(p.b, (uint64_t)xcd_logical_block() * WAVES + wave, (uint64_t)gridDim.x * WAVES)) {
        guard_wave_done(p.stat, arrive + WAVES - 2, WAVES, lane, 0);
        return;
    }
    uint64_t wf0 = 0, wfb = 0;                          // WALK: next frame / end of the clip segment the wave is in
    for (uint64_t first = (uint64_t)xcd_logical_block() * WAVES;; first += (uint64_t)gridDim.x * WAVES) {
        if (WALK) {
            if (wf0 >= wfb) {                           // next clip segment of the run
                if (cr.unit >= cr.end) break;
                cr.enter(p.b);
                const uint64_t seg_end = cr.c_end < cr.end ? cr.c_end : cr.end;
                wf0 = (cr.unit - cr.c_start) * (uint64_t)p.plan_fpu;
                wfb = (seg_end - cr.c_start) * (uint64_t)p.plan_fpu;
                if (wfb > cr.c_frames) wfb = cr.c_frames;
                cr.unit = seg_end;
                if (wf0 >= wfb) continue;
            }
        } else if (RUNS) {
            if (cr.unit >= cr.end) break;
            cr.enter(p.b);
        } else if (first >= p.b.n_units) {
            break;
        }
        const uint64_t unit = first + rs.slot;
        const bool have = RUNS || unit < p.b.n_units;
        const UnitLoc loc = RUNS ? cr.loc() : locate_unit(p.b, have ? unit : first);
        const uint64_t f0 = WALK ? wf0 : loc.unit * kFPW;
        const uint64_t left = WALK ? wfb - wf0 : ((RUNS || (have && f0 < loc.frames)) ? loc.frames - f0 : 0);
        const int nv = left < (uint64_t)kFPW ? (int)left : kFPW;
        // columns this unit stores: the clip's frames plus, for padded layouts, zero columns up to out_width
        const uint64_t width = (LAYOUT && p.b.d_unit_prefix == nullptr) ? p.b.out_width : loc.frames;
        const uint64_t wleft = have ? width - f0 : 0;
        const int ns = LAYOUT ? (wleft < (uint64_t)kFPW ? (int)wleft : kFPW) : nv;
        const float *src = loc.pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv, act3 = in3 && fl3 < nv;
        MS_PRIO(0);
        precise_phase1(fl, j, act && j < kFftJobs, tb, src + fl * p.hop, rows);
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(1);
        precise_phase2(fl, j, act, tb, rows);
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(2);
        float vals[NSLOTS], rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
        wave_phase3i_sums<NSLOTS, Lens>(fl3, j3, act3, p.slots, fblob, slice, st, rise, fprev);
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
        wave_phase3i_finish<NSLOTS>(fl3, j3, act3, n_mels, rise, fnext, slice, vals);
        __builtin_amdgcn_wave_barrier();
        if (LAYOUT) rs.template before_stores<2>(lane);
        bool flag;
        int kmin = 0x7fffffff, kmax = 0;
        if (LAYOUT && p.b.mel_major)
            flag = wave_phase4<NSLOTS, true, true, true>(fl3, j3, in3 && fl3 < ns, act3, n_mels, slice, vals, loc.out + f0, (long long)width, &kmin, &kmax);
        else
            flag = wave_phase4<NSLOTS, LAYOUT, true>(fl3, j3, in3 && fl3 < ns, act3, n_mels, slice, vals, loc.out + f0 * (uint64_t)n_mels, 0);
        __builtin_amdgcn_wave_barrier();
        if (LAYOUT && p.b.mel_major && p.b.d_unit_ext && have) unit_ext_store(p.b.d_unit_ext + 2 * unit, lane, kmin, kmax);
        if (stats) flagged += static_cast<unsigned>(__builtin_popcount(frame_mask<12, kFPW>(__builtin_amdgcn_ballot_w64(flag))));
        if (LAYOUT) rs.after_round();
        if (WALK) wf0 += kFPW;
        else if (RUNS) ++cr.unit;
    }
    guard_wave_done(p.stat, arrive + WAVES - 2, WAVES, lane, flagged);
}

// ------------------------------------------------------------------------------------
// The f64 kernel on the six-frame skeleton (whisper_six64.hpp): plain [frame][mel] batches of <= 80 mels, uniform and ragged, a
// contiguous run of 6-frame units per wave (ClipRun) -- MELSPEC_PRECISION_F64, and AUTO's gated second launch, which walks the very plan
// of the f32 launch in front of it (same unit size).  Twelve waves per workgroup, one workgroup per CU, three waves per SIMD.
// LDS words: [f64 tables][f32 mel section of the six-frame blob][WAVES x slice of 1368 doubles][2 words of guard_wave_done].
// ------------------------------------------------------------------------------------
struct Six64Params {
    BatchDesc b;
    const uint32_t *d_blob;
    int blob_words;       // multiple of 4
    int mel_off_words;    // where the f32 mel section (SixBlob::kMelStart.. of the six-frame blob) starts
    int hop;
    int n_mels;
    MelSlots slots;       // woff[] as in the six-frame blob (float offsets from SixBlob's base)
    FixSink stat;         // statistics only (tab, list unused): the frames that would have tripped the f32 kernels' guard
    const unsigned *gate; // AUTO's second launch: runs only when *gate == gate_value (the f32 launch's "heavy" verdict)
    unsigned gate_value;
};

template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kSix64Waves * 64, 3) void whisper400_six64_kernel(const Six64Params p) {
    constexpr int WAVES = kSix64Waves;
    if (p.gate != nullptr && *p.gate != p.gate_value) return;        // the batch was light: the f32 launch has finished it
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_words; i += WAVES * 64) ldsw[i] = p.d_blob[i];
    unsigned *wg_done = ldsw + p.blob_words + WAVES * Six64Layout::slice_doubles() * 2;
    if (tid < 2) wg_done[tid] = 0;
    __syncthreads();
    const double *tb = reinterpret_cast<const double *>(ldsw);
    // the shared phase-3 code addresses the mel tables as offsets from the base of the six-frame f32 blob
    const float *fblob = reinterpret_cast<const float *>(ldsw + p.mel_off_words) - SixBlob::kMelStart;

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double *rows = reinterpret_cast<double *>(ldsw + p.blob_words) + wave * Six64Layout::slice_doubles();
    float *slice = reinterpret_cast<float *>(rows);
    const int fl = lane / kSixLanes, j = lane - fl * kSixLanes;
    const bool in = lane < kSixFrames * kSixLanes;
    const int rofs = Six64Layout::row_offset(j);
    const int n_mels = Lens::kStatic ? Lens::kMels : p.n_mels;
    const int *starts = reinterpret_cast<const int *>(fblob + SixBlob::kMelStart) + j;
    const bool stats = p.stat.acc != nullptr;
    unsigned flagged = 0;

    ClipRun cr;
    if (!cr.init(p.b, (uint64_t)xcd_logical_block() * WAVES + wave, (uint64_t)gridDim.x * WAVES)) {
        guard_wave_done(p.stat, wg_done, WAVES, lane, 0);
        return;
    }
    for (; cr.unit < cr.end; ++cr.unit) {
        cr.enter(p.b);
        const uint64_t f0 = (cr.unit - cr.c_start) * kSixFrames;
        const uint64_t left = cr.c_frames - f0;
        const int nv = left < (uint64_t)kSixFrames ? (int)left : kSixFrames;
        const float *src = cr.c_pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv;
        MS_PRIO(0);
        // The tables never change, and with __restrict__ the compiler knows it: left alone it hoists the unit loop's ~50 sixteen-byte table
        // reads out of the loop (200 VGPRs of "loop invariants"), spills them in front of the loop and reloads them from scratch inside it.
        // An offset it cannot see through makes the reads belong to the iteration.
        int opaque0 = 0;
        asm volatile("" : "+s"(opaque0));
        const double *tbi = tb + opaque0;
        six64_phases12(fl, j, act, rofs, p.hop, tbi, src, rows, slice);
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(2);
        float vals[NSLOTS];
        {
            int st[NSLOTS];
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * kSixLanes];       // lanes 60..63 read valid entries too
            float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
            six_phase3_sums<NSLOTS, Lens>(fl, j, act, p.slots, fblob, slice, st, rise, fprev);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
            six_phase3_finish<NSLOTS>(fl, j, act, n_mels, rise, fnext, slice, vals);
        }
        __builtin_amdgcn_wave_barrier();
        float *out_tile = cr.c_out + f0 * (uint64_t)n_mels;
        const bool flag = six_phase4<NSLOTS, false, true>(fl, j, act, act, n_mels, slice, vals, out_tile, 0);
        __builtin_amdgcn_wave_barrier();
        if (stats) flagged += static_cast<unsigned>(__builtin_popcount(frame_mask<kSixLanes, kSixFrames>(__builtin_amdgcn_ballot_w64(flag))));
    }
    guard_wave_done(p.stat, wg_done, WAVES, lane, flagged);
}

// The same kernel for the padded / mel-major layouts (interleave_frames, src/mel.rs:480-544; BatchDesc::out_width / mel_major): the units
// are dealt round-robin and walked in workgroup-uniform rounds, the waves that hold adjacent units kept in step before their stores
// (RoundSync, as in whisper400_six_kernel) -- MELSPEC_PRECISION_F64 on a layout, and AUTO's gated launch behind a voting layout launch
// (same six-frame plan).  Uniform batches only (the layouts are).
template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kSix64Waves * 64, 3) void whisper400_six64_layout_kernel(const Six64Params p) {
    constexpr int WAVES = kSix64Waves;
    if (p.gate != nullptr && *p.gate != p.gate_value) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_words; i += WAVES * 64) ldsw[i] = p.d_blob[i];
    unsigned *arrive = ldsw + p.blob_words + WAVES * Six64Layout::slice_doubles() * 2;      // RoundSync counters, then guard_wave_done's two words
    if (tid < WAVES + 2) arrive[tid] = 0;
    __syncthreads();
    const double *tb = reinterpret_cast<const double *>(ldsw);
    const float *fblob = reinterpret_cast<const float *>(ldsw + p.mel_off_words) - SixBlob::kMelStart;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double *rows = reinterpret_cast<double *>(ldsw + p.blob_words) + wave * Six64Layout::slice_doubles();
    float *slice = reinterpret_cast<float *>(rows);
    const int fl = lane / kSixLanes, j = lane - fl * kSixLanes;
    const bool in = lane < kSixFrames * kSixLanes;
    const int rofs = Six64Layout::row_offset(j);
    const int n_mels = Lens::kStatic ? Lens::kMels : p.n_mels;
    const int *starts = reinterpret_cast<const int *>(fblob + SixBlob::kMelStart) + j;
    const bool stats = p.stat.acc != nullptr;
    unsigned flagged = 0;
    RoundSync<WAVES> rs(p.b.sync_rounds, wave, arrive);
    const uint64_t step = (uint64_t)gridDim.x * WAVES;
    for (uint64_t first = (uint64_t)xcd_logical_block() * WAVES; first < p.b.n_units; first += step) {
        const uint64_t unit = first + rs.slot;
        const bool have = unit < p.b.n_units;
        const UnitLoc loc = locate_unit(p.b, have ? unit : first);
        const uint64_t f0 = loc.unit * kSixFrames;
        const uint64_t left = (have && f0 < loc.frames) ? loc.frames - f0 : 0;
        const int nv = left < (uint64_t)kSixFrames ? (int)left : kSixFrames;
        const uint64_t width = p.b.out_width;                     // columns per clip: its frames plus the zero columns of a padded layout
        const uint64_t wleft = have ? width - f0 : 0;
        const int ns = wleft < (uint64_t)kSixFrames ? (int)wleft : kSixFrames;
        const float *src = loc.pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv;
        MS_PRIO(0);
        int opaque0 = 0;
        asm volatile("" : "+s"(opaque0));
        six64_phases12(fl, j, act, rofs, p.hop, tb + opaque0, src, rows, slice);
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(2);
        float vals[NSLOTS];
        {
            int st[NSLOTS];
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * kSixLanes];
            float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
            six_phase3_sums<NSLOTS, Lens>(fl, j, act, p.slots, fblob, slice, st, rise, fprev);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
            six_phase3_finish<NSLOTS>(fl, j, act, n_mels, rise, fnext, slice, vals);
        }
        __builtin_amdgcn_wave_barrier();
        rs.template before_stores<2>(lane);
        float *out_tile = p.b.mel_major ? loc.out + f0 : loc.out + f0 * (uint64_t)n_mels;
        int kmin = 0x7fffffff, kmax = 0;
        const bool flag = six_phase4<NSLOTS, true, true, true>(fl, j, in && fl < ns, act, n_mels, slice, vals, out_tile, p.b.mel_major ? (long long)width : 0, &kmin, &kmax);
        __builtin_amdgcn_wave_barrier();
        if (p.b.d_unit_ext && have) unit_ext_store(p.b.d_unit_ext + 2 * unit, lane, kmin, kmax);
        if (stats) flagged += static_cast<unsigned>(__builtin_popcount(frame_mask<kSixLanes, kSixFrames>(__builtin_amdgcn_ballot_w64(flag))));
        rs.after_round();
    }
    guard_wave_done(p.stat, arrive + WAVES, WAVES, lane, flagged);
}

// STFT export: Spectrogram::compute_all_cpu (src/stft.rs:89-115) -- the complex spectrum itself, f64 phases 1-2 of the
// precise kernel, a contiguous run of units per wave.  Output [clip][frame][bins] complex<T>; BatchDesc's "floats" are
// 32-bit words of that layout (floats per frame = bins * 2 * sizeof(T) / 4).
struct StftParams {
    BatchDesc b;
    const uint32_t *d_blob;   // the f64 table part of the precise blob
    int blob_words;
    int hop;
    int bins;                 // 201 (half spectrum) or 400 (the reference's full layout)
    int words_per_frame;
};

template <class T>
__global__ __launch_bounds__(kPreciseWaves * 64) void whisper400_stft_kernel(const StftParams p) {
    constexpr int WAVES = kPreciseWaves;
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_words; i += WAVES * 64) ldsw[i] = p.d_blob[i];
    __syncthreads();
    const double *tb = reinterpret_cast<const double *>(ldsw);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double *rows = reinterpret_cast<double *>(ldsw + p.blob_words) + wave * PreciseLayout::slice_doubles();
    const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
    const bool in = lane < kFPW * kMelJobs;
    ClipRun cr;
    if (!cr.init(p.b, (uint64_t)xcd_logical_block() * WAVES + wave, (uint64_t)gridDim.x * WAVES)) return;
    for (; cr.unit < cr.end; ++cr.unit) {
        cr.enter(p.b);
        const uint64_t f0 = (cr.unit - cr.c_start) * kFPW;
        const uint64_t left = cr.c_frames - f0;
        const int nv = left < (uint64_t)kFPW ? (int)left : kFPW;
        const float *src = cr.c_pcm + f0 * (uint64_t)p.hop;
        const bool act = in && fl < nv;
        precise_phase1(fl, j, act && j < kFftJobs, tb, src + fl * p.hop, rows);
        __builtin_amdgcn_wave_barrier();
        T *out = reinterpret_cast<T *>(cr.c_out + (f0 + (uint64_t)fl) * (uint64_t)p.words_per_frame);
        precise_phase2_spectrum<T>(fl, j, act, tb, rows, out, p.bins);
        __builtin_amdgcn_wave_barrier();
    }
}

// The same for any n_fft: one frame per workgroup, direct f64 DFT of bins 0..n_fft/2 from an LDS twiddle table (the
// arithmetic of generic_frame_kernel below), the upper half mirrored for the full layout.
// Mixed-radix FFT of n complex points in LDS for the generic kernels (n_fft = 2^a 3^b 5^c that is not a power of two: 320, 400, 480,
// 800, 1200 ...): Stockham auto-sort passes, one per factor (4, 2, 3 or 5), ping-pong between x and y (2n doubles each), the twiddle
// table tw[2j] = cos, tw[2j+1] = -sin of 2 pi j / n.  Pass for radix P, current length len, stride st:
//   y[q + st (P j + r)] = W_len^{j r} * sum_k x[q + st (j + m k)] W_P^{k r},   m = len / P, q < st, j < m.
// Every thread of the workgroup calls it; returns the buffer that holds the result in natural order.
struct FftPlan {
    int n_rad;
    unsigned long long packed;      // four bits per pass, first pass lowest (an array in the kernel arguments, indexed by the pass, would be
                                    // copied to scratch)
};
template <int NT, int P>
__device__ __forceinline__ void lds_fft_pass(int n, int len, int st, const double *tw, const double *x, double *y, int tid) {
    const int m = len / P, wstep = n / len, pstep = n / P;
    for (int b = tid; b < m * st; b += NT) {
        const int j = b / st, q = b - j * st;
        double ar[P], ai[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int at = q + st * (j + m * k);
            ar[k] = x[2 * at]; ai[k] = x[2 * at + 1];
        }
#pragma unroll
        for (int r = 0; r < P; ++r) {
            double vr = ar[0], vi = ai[0];
#pragma unroll
            for (int k = 1; k < P; ++k) {
                const int t = ((k * r) % P) * pstep;           // W_P^{k r}
                const double c = tw[2 * t], sn = tw[2 * t + 1];
                vr += ar[k] * c - ai[k] * sn;
                vi += ar[k] * sn + ai[k] * c;
            }
            const int t = static_cast<int>((static_cast<long long>(j) * r * wstep) % n);      // W_len^{j r}
            const double c = tw[2 * t], sn = tw[2 * t + 1];
            const int to = q + st * (P * j + r);
            y[2 * to] = vr * c - vi * sn;
            y[2 * to + 1] = vr * sn + vi * c;
        }
    }
}
template <int NT>
__device__ __forceinline__ double *lds_fft_mixed(const FftPlan &plan, int n, const double *tw, double *x, double *y, int tid) {
    int len = n, st = 1;
    for (int pass = 0; pass < plan.n_rad; ++pass) {
        const int P = static_cast<int>((plan.packed >> (4 * pass)) & 15ull);
        __syncthreads();
        switch (P) {
            case 2: lds_fft_pass<NT, 2>(n, len, st, tw, x, y, tid); break;
            case 3: lds_fft_pass<NT, 3>(n, len, st, tw, x, y, tid); break;
            case 4: lds_fft_pass<NT, 4>(n, len, st, tw, x, y, tid); break;
            default: lds_fft_pass<NT, 5>(n, len, st, tw, x, y, tid); break;
        }
        len /= P; st *= P;
        double *tmp = x; x = y; y = tmp;
    }
    __syncthreads();
    return x;
}

struct GenericStftParams {
    BatchDesc b;             // units == frames
    int n_fft, hop, bins, words_per_frame, f64;
    int fft_log2;            // log2(n_fft) for a power-of-two n_fft >= 8 (in-LDS FFT as in generic_frame_kernel), else 0
    FftPlan plan;            // n_rad > 0: mixed-radix FFT (lds_fft_mixed) for 2-3-5-smooth n_fft; both zero: direct DFT
    const double *d_win;     // [n_fft]
    const double *d_tw;      // [n_fft] interleaved (cos, -sin) of 2*pi*j/n_fft
};

template <int NT>
__global__ __launch_bounds__(NT) void generic_stft_kernel(const GenericStftParams p) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    double *tw = ldsd;                       // 2*n_fft
    double *xw = tw + 2 * p.n_fft;           // n_fft
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * p.n_fft; i += NT) tw[i] = p.d_tw[i];
    const uint64_t n_units = batch_n_units(p.b);
    for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const UnitLoc loc = locate_unit(p.b, unit);
        const float *x = loc.pcm + loc.unit * (uint64_t)p.hop;
        __syncthreads();
        const int mbits = p.fft_log2 - 1;
        const bool mixed = p.plan.n_rad > 0;
        for (int i = tid; i < p.n_fft; i += NT) {
            // FFT form: sample i of the real frame is component (i & 1) of complex point i >> 1, stored bit-reversed; mixed-radix
            // form: complex point i with a zero imaginary part
            const int at = p.fft_log2 ? static_cast<int>(2 * (mbits > 0 ? (__brev(static_cast<unsigned>(i >> 1)) >> (32 - mbits)) : 0u)) + (i & 1) : (mixed ? 2 * i : i);
            xw[at] = (double)x[i] * p.d_win[i];                                        // src/stft.rs:160-165
            if (mixed) xw[at + 1] = 0.0;
        }
        const int M = p.n_fft >> 1;
        const double *res = xw;
        if (mixed) res = lds_fft_mixed<NT>(p.plan, p.n_fft, tw, xw, xw + 2 * p.n_fft, tid);
        if (p.fft_log2) {
            for (int len = 2; len <= M; len <<= 1) {
                __syncthreads();
                const int half = len >> 1, tstep = p.n_fft / len;
                for (int b = tid; b < (M >> 1); b += NT) {
                    const int g = b / half, j = b - g * half;
                    const int i0 = g * len + j, i1 = i0 + half;
                    const double c = tw[2 * (j * tstep)], sn = tw[2 * (j * tstep) + 1];
                    const double ur = xw[2 * i0], ui = xw[2 * i0 + 1];
                    const double xr = xw[2 * i1], xi = xw[2 * i1 + 1];
                    const double vr = xr * c - xi * sn, vi = xr * sn + xi * c;
                    xw[2 * i0] = ur + vr; xw[2 * i0 + 1] = ui + vi;
                    xw[2 * i1] = ur - vr; xw[2 * i1 + 1] = ui - vi;
                }
            }
        }
        __syncthreads();
        float *o = loc.out + loc.unit * (uint64_t)p.words_per_frame;
        for (int k = tid; k <= p.n_fft / 2; k += NT) {
            double re = 0.0, im = 0.0;
            if (mixed) {
                re = res[2 * k]; im = res[2 * k + 1];
            } else if (p.fft_log2) {
                const int ka = k == M ? 0 : k, kb = (M - k) & (M - 1);
                const double ar = xw[2 * ka], ai = xw[2 * ka + 1];
                const double br = xw[2 * kb], bi = -xw[2 * kb + 1];
                const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
                const double orr = 0.5 * (ai - bi), oi = -0.5 * (ar - br);
                const double c = tw[2 * k], sn = tw[2 * k + 1];
                re = er + (orr * c - oi * sn);
                im = ei + (orr * sn + oi * c);
            } else {
                int idx = 0;
                for (int n = 0; n < p.n_fft; ++n) {
                    re += xw[n] * tw[2 * idx];
                    im += xw[n] * tw[2 * idx + 1];
                    idx += k;
                    if (idx >= p.n_fft) idx -= p.n_fft;
                }
            }
            const int mk = p.n_fft - k;
            const bool mirror = p.bins == p.n_fft && k > 0 && mk > k;
            if (p.f64) {
                double *od = reinterpret_cast<double *>(o);
                od[2 * k] = re; od[2 * k + 1] = im;
                if (mirror) { od[2 * mk] = re; od[2 * mk + 1] = -im; }
            } else {
                o[2 * k] = (float)re; o[2 * k + 1] = (float)im;
                if (mirror) { o[2 * mk] = (float)re; o[2 * mk + 1] = (float)-im; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Fused Kaldi-fbank kernel (phases in fbank_wave.hpp): 4 frames per wavefront (16 lanes each), no workgroup barrier
// in the loop.  Writes un-normalised features; CMN is cmn_kernel.
// ------------------------------------------------------------------------------------
struct FbankFastParams {
    BatchDesc b;            // units of kFbFPW frames
    const uint32_t *d_blob;
    int blob_words;         // 32-bit words, multiple of 4
    int mel_off_words;      // mel section offset inside the blob
    int shift;              // frame shift (hop) in samples
    int n_mels;
    double preemph;
    float floor_v;          // Kaldi: energy floor; NeMo: log_zero_guard
    int use_log, use_power;
    long long clip_len;     // NeMo (uniform batches): samples per clip, for the centre padding
    int org0;               // NeMo: clip index of tap 0 of frame 0 (-200 centred, +56 not centred)
    const uint64_t *d_len;  // NeMo, ragged batches: samples of clip c (BatchDesc::d_frames then holds the PADDED column count of the
    const uint64_t *d_valid;//   clip -- what the units cover and the row width -- and d_valid its valid frames)
    MelSlots slots;
};

constexpr int kFlavorKaldi = 0, kFlavorNemo = 1, kFlavorWhisper = 2;

// The feature-major store of the f32 NeMo kernel, staged through LDS (round 5).  A wave's unit is four adjacent columns of every mel
// row: stored directly that is 16 bytes per row and wave (32 with pairs of waves kept in step, RoundSync) -- 135 write requests per
// unit, 1.5 x write amplification, and a fifth of the kernel's time.  Here the WAVES units of a round (adjacent units: WAVES x 4
// adjacent columns) are put into an LDS image [mel][WAVES x 4] and stored as runs of WAVES x 16 bytes per mel row by all threads, a
// 16-byte piece each.  Two images: a wave drains round r - 1 (after its own phases of round r, when every wave has long staged r - 1:
// the wait below has a round of slack, so the waves keep drifting up to one round apart) and then stages round r over the image of
// round r - 2, which every wave drained before it staged r - 1.  One LDS counter, no workgroup barrier.
// Rows are kCols + 4 floats apart (13 sixteen-byte pieces at twelve waves): the sixteen lanes of a frame (mels j, j + 15, ...) write
// sixteen different 4-bank groups, and a lane's NSLOTS stores are one base address + compile-time offsets (an XOR swizzle of unpadded
// rows costs a VGPR per slot, which the twelve-wave kernel does not have).
template <int WAVES>
struct StagedRows {
    static constexpr int kCols = WAVES * kFbFPW;
    static constexpr int kPitch = kCols + 4;
    struct alignas(16) UnitInfo {
        float *col;          // &out[mel 0][first column of the unit]
        long long row_w;     // floats between mel rows
        int ns;              // columns of the unit that exist in the output (0: no unit this round)
        int pad;
    };
    MS_HD static constexpr size_t image_floats(int n_mels) { return static_cast<size_t>(n_mels) * kPitch; }
    MS_HD static constexpr size_t bytes(int n_mels) { return 2 * (image_floats(n_mels) * sizeof(float) + WAVES * sizeof(UnitInfo)); }

    float *image;            // [2][n_mels][kPitch]
    UnitInfo *info;          // [2][WAVES]
    unsigned *count;         // units staged by the workgroup so far (every wave stages every round, with or without a unit)
    int n_mels;
    unsigned round = 0;

    __device__ __forceinline__ StagedRows(void *base, unsigned *counter, int mels) : count(counter), n_mels(mels) {
        image = static_cast<float *>(base);
        info = reinterpret_cast<UnitInfo *>(image + 2 * image_floats(n_mels));
    }
    __device__ __forceinline__ void wait_staged(unsigned rounds, int lane) const {
        if (lane == 0)
            while (__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < rounds * WAVES) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_wave_barrier();
    }
    // all threads: the image of round `r` to global memory.  tid: the thread's index, opaque to the optimiser (the task -> row / piece
    // arithmetic is wanted here, once per round, not hoisted out of the unit loop into registers that spill)
    __device__ __forceinline__ void drain(unsigned r, int tid) const {
        typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
        const float *img = image + (r & 1u) * image_floats(n_mels);
        const UnitInfo *ui = info + (r & 1u) * WAVES;
        for (int task = tid; task < n_mels * WAVES; task += WAVES * 64) {
            const int m = task / WAVES, c = task - m * WAVES;
            const UnitInfo u = ui[c];
            const f4 v = ld4(img + m * kPitch + (c << 2));
            float *dst = u.col + static_cast<long long>(m) * u.row_w;
            if (u.ns == kFbFPW) {
                *reinterpret_cast<v4u *>(dst) = v4u{v.x, v.y, v.z, v.w};
            } else {
                if (u.ns > 0) dst[0] = v.x;
                if (u.ns > 1) dst[1] = v.y;
                if (u.ns > 2) dst[2] = v.z;
            }
        }
    }
    // a wave's unit of this round (every lane calls; vals: this lane's mel j + 15 i of frame fl, zero for a column past the valid frames)
    template <int NSLOTS>
    __device__ __forceinline__ void put(int wave, int lane, const float (&vals)[NSLOTS], float *col, long long row_w, int ns) {
        const int l = fresh_lane_value(lane), fl = l / kFbLanes, j = l - fl * kFbLanes;      // derived here, not held across the unit loop
        float *mine = image + (round & 1u) * image_floats(n_mels) + j * kPitch + (wave << 2) + fl;
        if (j < kFbOwn) {
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i)
                if (j + kFbOwn * i < n_mels) mine[i * kFbOwn * kPitch] = vals[i];
        }
        if (lane == 0) info[(round & 1u) * WAVES + wave] = UnitInfo{col, row_w, ns, 0};
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        ++round;
    }
};

// FLAVOR = Kaldi: Fbank::compute (src/fbank.rs:141-236), frame-major output, CMN by cmn_kernel.
// FLAVOR = Whisper: compute_mel_spectrogram_cpu at n_fft = 512 (src/stft.rs:119-138): 512-sample frames, Hann,
//                 log10 / per-frame clamp / (x+4)/4, frame-major output (plain and ragged batches).
// FLAVOR = NeMo:  BatchLogMelSpectrogram::compute (src/mel.rs:321-385), feature-major output of
//                 b.out_width columns per mel row (columns past the valid frames are zero).
// RUNS (frame-major plain output: Kaldi always, Whisper-512 without a layout): a contiguous run of units per wave (ClipRun).
template <class T, int WAVES, int MINW, int FLAVOR = kFlavorKaldi, int NSLOTS = kFbSlots, class Lens = LensRuntime, bool RUNS = false>
__global__ __launch_bounds__(WAVES * 64, MINW) void fbank512_wave_kernel(const FbankFastParams p) {
    using L = FbankLayout<T>;
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_words; i += WAVES * 64) ldsw[i] = p.d_blob[i];
    // NeMo: the feature-major store gives every wave 16 bytes of each mel row per unit; the units are walked in workgroup-uniform
    // rounds and the waves that hold adjacent units are kept in step before their stores (RoundSync, as in the mel-major Whisper kernels)
    constexpr bool ROUNDS = FLAVOR == kFlavorNemo;
    // the f32 NeMo kernel stages its feature-major rows in LDS (StagedRows) instead of keeping pairs of waves in step
    constexpr bool STAGE = FLAVOR == kFlavorNemo && sizeof(T) == 4;
    unsigned *arrive = ldsw + p.blob_words + WAVES * L::slice_elems() * (sizeof(T) / 4);     // 16 words: RoundSync counters; [15]: StagedRows
    if (ROUNDS && tid < 16) arrive[tid] = 0;
    __syncthreads();
    const T *tblob = reinterpret_cast<const T *>(ldsw);
    const float *mel = reinterpret_cast<const float *>(ldsw + p.mel_off_words);

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    T *slice = reinterpret_cast<T *>(ldsw + p.blob_words) + wave * L::slice_elems();
    const int fl = lane / kFbLanes, j = lane - fl * kFbLanes;
    const bool in = lane < kFbFPW * kFbLanes;
    // first bin of this lane's interval per slot: held across the unit loop by the compile-time banks; the run-time-lens variants
    // re-read the ten words in front of phase 3 instead (they sit at the 256-VGPR limit: holding them spilled inside the loop)
    int st[NSLOTS];
    const int *starts = reinterpret_cast<const int *>(mel + FbankBlob::kMelStart);
    // (the twelve-wave f32 NeMo kernel has no registers to hold them either)
    constexpr bool HOLD_STARTS = Lens::kStatic && !(FLAVOR == kFlavorNemo && sizeof(T) == 4);
    if (HOLD_STARTS) {
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) st[i] = in ? starts[i * kFbLanes + j] : 0;
    }
    const bool use_power = p.use_power != 0, use_log = p.use_log != 0;
    const T preemph = static_cast<T>(p.preemph);

    static_assert(!(RUNS && FLAVOR == kFlavorNemo), "the feature-major store wants adjacent units in adjacent waves");
    ClipRun cr;
    if (RUNS && !cr.init(p.b, (uint64_t)xcd_logical_block() * WAVES + wave, (uint64_t)gridDim.x * WAVES)) return;
    RoundSync<WAVES> rs((ROUNDS && !STAGE) ? p.b.sync_rounds : 0, wave, arrive);
    StagedRows<STAGE ? WAVES : 4> staged(arrive + 16, arrive + 15, p.n_mels);
    // batches planned on the device (plan_ragged_device_kernel) keep the real unit count in d_n_units; n_units is the host's bound
    const uint64_t n_units = RUNS ? 0 : scalar64(batch_n_units(p.b));
    // (STAGE with a contiguous range of units per workgroup instead of rounds dealt over the grid -- consecutive rounds extending the same
    // mel rows, no division per unit -- was measured: +1.4 %, profiles/r05_f32_512.txt)
    for (uint64_t first = (uint64_t)xcd_logical_block() * WAVES + (ROUNDS ? 0 : wave);; first += (uint64_t)gridDim.x * WAVES) {
        const uint64_t unit = ROUNDS ? first + rs.slot : first;
        if (RUNS) {
            if (cr.unit >= cr.end) break;
            cr.enter(p.b);
        } else if (first >= n_units) {
            break;
        }
        const bool have = !ROUNDS || unit < n_units;       // a wave without a unit idles through the round
        UnitLoc loc = RUNS ? cr.loc() : locate_unit(p.b, have ? unit : first);
        if (STAGE) loc = scalar_loc(loc);          // this kernel has no VGPRs for them
        const uint64_t f0 = loc.unit * kFbFPW;
        // valid frames of the clip (NeMo ragged: loc.frames is the padded width there)
        const uint64_t vframes = (FLAVOR == kFlavorNemo && p.d_valid) ? p.d_valid[loc.clip] : loc.frames;
        const uint64_t left = (have && f0 < vframes) ? vframes - f0 : 0;
        const int nv = left < (uint64_t)kFbFPW ? (int)left : kFbFPW;
        const bool act = in && fl < nv;
        MS_PRIO(0);
        if (FLAVOR == kFlavorKaldi) {
            const float *frame = loc.pcm + (f0 + (uint64_t)(act ? fl : 0)) * (uint64_t)p.shift;
            // the frame mean (src/fbank.rs:165-166: the frame's sixteen lanes, a fixed tree over DPP), DC removal, pre-emphasis and the Povey window
            // from ONE set of loads (fb_kaldi_input)
            if (act) {
                cpx<T> x[16];
                fb_kaldi_input<T>(frame, j, preemph, f0 + fl == 0 && j == 0, tblob, x);
                fb_column_finish<T>(x, j, tblob, slice + fl * L::kXStride);
            }
        } else if (FLAVOR == kFlavorWhisper) {
            w512_phase1<T>(fl, j, act, loc.pcm + (f0 + (uint64_t)(act ? fl : 0)) * (uint64_t)p.shift, tblob, slice);
        } else {
            const long long clip_len = p.d_len ? (long long)p.d_len[loc.clip] : p.clip_len;
            const long long org = (long long)(f0 + (uint64_t)fl) * p.shift + p.org0;
            const bool inside = org >= 1 && org + 400 <= clip_len;
            const bool all_inside = __builtin_amdgcn_ballot_w64(act && !inside) == 0;
            nemo_phase1<T>(fl, j, act, all_inside, loc.pcm, org, clip_len, static_cast<float>(p.preemph), tblob, slice);
        }
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(1);
        {
            cpx<T> own[16], part[8];
            fb_phase2_dft<T, STAGE>(fl, j, act, slice, own);
#pragma unroll
            for (int i = 0; i < 8; ++i) part[i] = {partner16(own[8 + i].re), partner16(own[8 + i].im)};
            if (FLAVOR == kFlavorWhisper) fb_phase2_split<T, true, true>(fl, j, act, tblob, own, part, slice);
            else if (use_power) fb_phase2_split<T, true>(fl, j, act, tblob, own, part, slice);
            else fb_phase2_split<T, false>(fl, j, act, tblob, own, part, slice);
        }
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(2);
        float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS];
        if (!HOLD_STARTS) {
            const int *mine = starts + (STAGE ? fresh_lane_value(j) : j);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) st[i] = in ? mine[i * kFbLanes] : 0;
        }
        fb_phase3_sums<T, NSLOTS, Lens>(fl, j, act, p.slots, mel, slice, st, rise, fprev);
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
        if (FLAVOR == kFlavorKaldi) {
            fb_phase3_store<NSLOTS>(fl, j, act, p.n_mels, p.floor_v, use_log, rise, fnext, loc.out + f0 * (uint64_t)p.n_mels);
        } else if (FLAVOR == kFlavorWhisper) {
            float vals[NSLOTS];
            float *slice_f = reinterpret_cast<float *>(slice);
            w512_phase3_log<NSLOTS>(fl, j, act, p.n_mels, rise, fnext, slice_f, vals);
            __builtin_amdgcn_wave_barrier();
            // columns this unit stores: the clip's frames plus, for padded layouts, zero columns up to out_width
            const uint64_t width = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
            const uint64_t wleft = width - f0;
            const int ns = wleft < (uint64_t)kFbFPW ? (int)wleft : kFbFPW;
            if (p.b.mel_major)
                w512_phase4<NSLOTS>(fl, j, in && fl < ns, act, p.n_mels, slice_f, vals, loc.out + f0, (long long)width);
            else
                w512_phase4<NSLOTS>(fl, j, in && fl < ns, act, p.n_mels, slice_f, vals, loc.out + f0 * (uint64_t)p.n_mels, 0);
        } else {
            const uint64_t row_w = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
            const uint64_t wleft = have ? row_w - f0 : 0;
            const int ns = wleft < (uint64_t)kFbFPW ? (int)wleft : kFbFPW;
            if (STAGE) {
                float vals[NSLOTS];
#pragma unroll
                for (int i = 0; i < NSLOTS; ++i) vals[i] = act ? fast_ln((rise[i] + fnext[i]) + p.floor_v) : 0.0f;     // nemo_phase3_store's value
                if (staged.round > 0) {
                    int dtid = tid;
                    asm volatile("" : "+v"(dtid));          // see StagedRows::drain
                    staged.wait_staged(staged.round, lane);
                    staged.drain(staged.round - 1, dtid);
                }
                staged.template put<NSLOTS>(wave, lane, vals, loc.out + f0, (long long)row_w, ns);
            } else {
                rs.template before_stores<2>(lane);
                nemo_phase3_store<NSLOTS>(fl, j, in && fl < ns, act, p.n_mels, p.floor_v, rise, fnext, loc.out + f0, (long long)row_w);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (ROUNDS) rs.after_round();
        if (RUNS) ++cr.unit;
    }
    if (STAGE && staged.round > 0) {
        staged.wait_staged(staged.round, lane);
        staged.drain(staged.round - 1, tid);
    }
}

// Kaldi fbank with the CMN inside (Fbank::compute incl. src/fbank.rs:224-233), for uniform batches of many clips: a workgroup
// owns whole clips, each of its eight waves a contiguous eighth of the clip's units.  Nothing in it waits on a workgroup barrier:
//   * a wave adds the values it stores to per-lane column sums (one f32 add per stored value), folds the four frame positions
//     at the end of its run and leaves its 80 partial sums in LDS; the wave that arrives last adds the eight partials in a
//     fixed order, divides by the frame count and publishes the clip's means;
//   * the subtraction of clip c is done one clip later: every wave, when it has finished its run of clip c+1, subtracts the
//     means from an eighth of clip c's rows (16 sixteen-byte loads in flight per lane) -- by then the means have long been
//     published, so the wait in front of it never spins in practice, and the rows (319 KB at 10 s; 82 MB over the 256
//     workgroups) come back from the Infinity Cache rather than from HBM.
// The column sums are therefore NOT the reference's order (ndarray's mean() of a strided column is an f32 left fold over the
// frames); they are a fixed tree of 31-term folds, deterministic from run to run, and more accurate than the fold: config 3
// sits 1.5e-5 from the oracle (which folds like the reference) against the 1e-4 bar, the reference's own rounding error in that
// mean being ~1e-5.  cmn_kernel (the reference's order, 1.9e-6) stays the path for everything this kernel does not take:
// ragged batches, n_mels not a multiple of 4, fewer clips than fill the CUs evenly.
// History (profiles/r02_fbank.txt): in-order sums under a ticket / through an LDS ring were 1.06-1.49 ms against 0.92 ms
// for the two kernels; what makes the fusion pay is giving up the order and the barrier.
struct FbankClipParams {
    FbankFastParams f;
    uint64_t frames;        // per clip (uniform batches)
    int lab_skip;           // lab builds, timing ablations (wrong results): 1 = no subtraction, 2 = its loads only, 4 = its stores only
};

template <int WAVES>
struct ClipCmnShared {
    float part[2][WAVES][96];
    float mean[2][96];
    unsigned arrived[2], ready[2];
    unsigned published, claimed, ids[8];     // ragged batches: the workgroup's clips, in the order it took them from the ticket counter
};

// The workgroup's n-th clip of a ragged batch (0xffffffff: the batch is used up).  Whichever wave asks first takes a ticket from the
// device counter and publishes the clip in LDS; the others read it there.  No wave is ever more than two clips ahead of another (the
// subtraction of clip c waits for every wave's run of clip c), so a ring of eight cannot wrap.
template <int WAVES>
MS_DEV uint32_t clip_queue_get(ClipCmnShared<WAVES> *sh, unsigned n, int lane, const BatchDesc &b) {
    unsigned id = 0xffffffffu;
    if (lane == 0) {
        // bounded: a slot that is never published would be a bug; the parity tests catch a wrong result, nothing recovers a hung GPU
        for (unsigned spin = 0; spin < (1u << 22); ++spin) {
            if (__hip_atomic_load(&sh->published, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > n) { id = sh->ids[n & 7u]; break; }
            unsigned expect = n;
            if (__hip_atomic_compare_exchange_strong(&sh->claimed, &expect, n + 1, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                const unsigned t = atomicAdd(b.d_ticket, 1u);
                id = t < b.n_clips ? b.d_order[t] : 0xffffffffu;
                sh->ids[n & 7u] = id;
                __hip_atomic_store(&sh->published, n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(id)));
}

// The subtraction of a finished clip, one wave's share: groups of R = 64 / (n_mels / 4) rows (one 16-byte piece per lane), group
// g belongs to wave g % WAVES, the wave's groups are numbered by `slot` (g = wave + WAVES * slot).
template <int WAVES>
struct ClipCmnSub {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 *o4 = nullptr;          // the finished clip's rows
    f4 m4;                     // this lane's four column means
    uint32_t frames = 0, q4 = 0, R = 1, r = 0, c4 = 0, slots = 0, next = 0;
    bool lane_on = false, have_mean = false;
    int lab = 0;               // lab builds: 2 = loads only, 4 = stores only

    MS_DEV void begin(float *out, uint64_t frames_, int nm, int wave, int lane) {
        o4 = reinterpret_cast<f4 *>(out);
        frames = static_cast<uint32_t>(frames_);
        q4 = static_cast<uint32_t>(nm) >> 2;
        R = 64u / q4;
        r = static_cast<uint32_t>(lane) / q4;
        c4 = static_cast<uint32_t>(lane) - r * q4;
        lane_on = r < R;
        const uint32_t groups = (frames + R - 1) / R;
        slots = groups > static_cast<uint32_t>(wave) ? (groups - wave + WAVES - 1) / WAVES : 0;
        next = 0;
        have_mean = false;
    }
    // wave-uniform; never waits
    MS_DEV bool poll(ClipCmnShared<WAVES> *sh, int par, unsigned expect, int lane) {
        if (have_mean) return true;
        bool ok = false;
        if (lane == 0) ok = __hip_atomic_load(&sh->ready[par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= expect;
        if (__builtin_amdgcn_ballot_w64(ok) == 0) return false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        m4 = *reinterpret_cast<const f4 *>(&sh->mean[par][4 * c4]);
        have_mean = true;
        return true;
    }
    MS_DEV void wait(ClipCmnShared<WAVES> *sh, int par, unsigned expect, int lane) {
        // bounded (a mean that is never published would be a bug; a wrong result is caught by the parity tests, a hung GPU is not recoverable)
        for (unsigned spin = 0; spin < (1u << 22) && !poll(sh, par, expect, lane); ++spin) __builtin_amdgcn_s_sleep(2);
    }
    // the load of one slot (unconditional: rows past the clip re-read its last row); returns the piece's index
    MS_DEV uint32_t load(int wave, uint32_t slot, f4 &v, bool &ok) const {
        const uint32_t row = (static_cast<uint32_t>(wave) + WAVES * slot) * R + r;
        ok = lane_on && slot < slots && row < frames;
        const uint32_t idx = (row < frames ? row : frames - 1) * q4 + c4;
        if (lab & 4) v = m4; else v = o4[idx];
        return idx;
    }
    MS_DEV void store(uint32_t idx, const f4 &v, bool ok) const {
        if (lab & 2) { asm volatile("" :: "v"(v)); return; }
        if (ok) o4[idx] = v - m4;
    }
    // everything that is left, 8 loads in flight, the next batch's loads issued before this batch's stores
    MS_DEV void finish(int wave) {
        constexpr int K = 8;
        if (next >= slots) return;
        f4 v[K], w[K];
        uint32_t iv[K], iw[K];
        bool kv[K], kw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) iv[k] = load(wave, next + k, v[k], kv[k]);
        next += K;
        while (next < slots) {                       // wave-uniform
#pragma unroll
            for (int k = 0; k < K; ++k) iw[k] = load(wave, next + k, w[k], kw[k]);
            next += K;
#pragma unroll
            for (int k = 0; k < K; ++k) store(iv[k], v[k], kv[k]);
#pragma unroll
            for (int k = 0; k < K; ++k) { v[k] = w[k]; iv[k] = iw[k]; kv[k] = kw[k]; }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) store(iv[k], v[k], kv[k]);
    }
};

template <int NSLOTS, class Lens, bool RAGGED = false>
__global__ __launch_bounds__(8 * 64, 1) void fbank512_clip_kernel(const FbankClipParams q) {
    using T = double;
    using L = FbankLayout<T>;
    constexpr int WAVES = 8, NT = WAVES * 64;
    const FbankFastParams &p = q.f;
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_words; i += NT) ldsw[i] = p.d_blob[i];
    auto *sh = reinterpret_cast<ClipCmnShared<WAVES> *>(ldsw + p.blob_words + WAVES * L::slice_elems() * 2);
    if (tid < 2) { sh->arrived[tid] = 0; sh->ready[tid] = 0; }
    if (tid == 2) { sh->published = 0; sh->claimed = 0; }
    __syncthreads();
    const T *tblob = reinterpret_cast<const T *>(ldsw);
    const float *mel = reinterpret_cast<const float *>(ldsw + p.mel_off_words);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    T *slice = reinterpret_cast<T *>(ldsw + p.blob_words) + wave * L::slice_elems();
    const int fl = lane / kFbLanes, j = lane - fl * kFbLanes;
    int st[NSLOTS];
    {
        const int *starts = reinterpret_cast<const int *>(mel + FbankBlob::kMelStart);
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) st[i] = starts[i * kFbLanes + j];
    }
    const bool use_power = p.use_power != 0, use_log = p.use_log != 0;
    const T preemph = static_cast<T>(p.preemph);
    const int nm = p.n_mels;
    unsigned gen = 0;                      // clips this workgroup has finished
    ClipCmnSub<WAVES> sub;                 // the previous clip's subtraction
    sub.lab = q.lab_skip & 6;
    // uniform batches: clips blockIdx.x, + gridDim.x, ... of one length; ragged: the next clip of the batch's longest-first order
    for (unsigned seq = 0;; ++seq) {
        uint32_t clip;
        uint64_t frames;
        const float *pcm;
        float *out;
        if (RAGGED) {
            clip = clip_queue_get<WAVES>(sh, seq, lane, p.b);
            if (clip == 0xffffffffu) break;
            frames = scalar64(p.b.d_frames[clip]);
            if (frames == 0) continue;         // zeros((0, num_mel_bins)), src/fbank.rs:147-149: nothing to write
            pcm = p.b.pcm + scalar64(p.b.d_off[clip]);
            out = p.b.out + scalar64(p.b.d_out_off[clip]);
        } else {
            clip = blockIdx.x + seq * gridDim.x;
            if (clip >= p.b.n_clips) break;
            frames = q.frames;
            pcm = p.b.pcm + (uint64_t)clip * p.b.clip_stride;
            out = p.b.out + (uint64_t)clip * p.b.out_stride;
        }
        const uint32_t units = static_cast<uint32_t>((frames + kFbFPW - 1) / kFbFPW);
        const uint32_t u0 = static_cast<uint32_t>((uint64_t)units * wave / WAVES), u1 = static_cast<uint32_t>((uint64_t)units * (wave + 1) / WAVES);
        const int par = gen & 1;
        const unsigned prev_turn = (gen + 1) / 2;      // == (gen - 1) / 2 + 1 for gen > 0
        float acc[NSLOTS];
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) acc[i] = 0.0f;
        for (uint32_t u = u0; u < u1; ++u) {
            const uint64_t f0 = (uint64_t)u * kFbFPW;
            const uint64_t left = frames - f0;
            const int nv = left < (uint64_t)kFbFPW ? (int)left : kFbFPW;
            const bool act = fl < nv;
            MS_PRIO(0);
            const float *frame = pcm + (f0 + (uint64_t)(act ? fl : 0)) * (uint64_t)p.shift;
            // the frame mean (src/fbank.rs:165-166: the frame's sixteen lanes, a fixed tree over DPP), DC removal, pre-emphasis and the Povey window
            // from ONE set of loads (fb_kaldi_input)
            if (act) {
                cpx<T> x[16];
                fb_kaldi_input<T>(frame, j, preemph, f0 + fl == 0 && j == 0, tblob, x);
                fb_column_finish<T>(x, j, tblob, slice + fl * L::kXStride);
            }
            __builtin_amdgcn_wave_barrier();
            MS_PRIO(1);
            {
                cpx<T> own[16], part[8];
                fb_phase2_dft<T>(fl, j, act, slice, own);
#pragma unroll
                for (int i = 0; i < 8; ++i) part[i] = {partner16(own[8 + i].re), partner16(own[8 + i].im)};
                if (use_power) fb_phase2_split<T, true>(fl, j, act, tblob, own, part, slice);
            else fb_phase2_split<T, false>(fl, j, act, tblob, own, part, slice);
            }
            __builtin_amdgcn_wave_barrier();
            MS_PRIO(2);
            float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS], vals[NSLOTS];
            fb_phase3_sums<T, NSLOTS, Lens>(fl, j, act, p.slots, mel, slice, st, rise, fprev);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) { fnext[i] = wave_shift_down1(fprev[i]); vals[i] = 0.0f; }
            fb_phase3_store<NSLOTS>(fl, j, act, nm, p.floor_v, use_log, rise, fnext, out + f0 * (uint64_t)nm, vals);
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i) acc[i] += vals[i];
            __builtin_amdgcn_wave_barrier();
        }
        MS_PRIO(0);
        // this wave's share of the previous clip's subtraction (its means were published a whole run ago: the wait does not spin).
        // Spreading it over the units of the run -- two pieces loaded after phase 1, stored at the end of the unit -- was measured
        // and is slower (+0.08 ms against +0.07 ms, profiles/r02_fbank.txt): the cost is the extra traffic, not this wave's stall
        if (gen > 0 && !(q.lab_skip & 1)) {
            sub.wait(sh, par ^ 1, prev_turn, lane);
            sub.finish(wave);
        }
        // the wave's column sums: frame positions (0+1)+(2+3), then lanes of position 0 write them
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) {
            acc[i] += __shfl_xor(acc[i], 16);
            acc[i] += __shfl_xor(acc[i], 32);
        }
        if (fl == 0 && j < kFbOwn) {
#pragma unroll
            for (int i = 0; i < NSLOTS; ++i)
                if (j + kFbOwn * i < nm) sh->part[par][wave][j + kFbOwn * i] = acc[i];
        }
        __builtin_amdgcn_wave_barrier();
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&sh->arrived[par], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);    // releases the rows this wave stored, too
        const unsigned turn = gen / 2 + 1;      // how many clips of this parity, this one included
        if (__builtin_amdgcn_ballot_w64(lane == 0 && old == turn * WAVES - 1) != 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float fr = static_cast<float>(frames);
            for (int m = lane; m < nm; m += 64) {
                const float (*pp)[96] = sh->part[par];
                const float s = ((pp[0][m] + pp[1][m]) + (pp[2][m] + pp[3][m])) + ((pp[4][m] + pp[5][m]) + (pp[6][m] + pp[7][m]));
                sh->mean[par][m] = f32_div_rn(s, fr);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(&sh->ready[par], turn, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        sub.begin(out, frames, nm, wave, lane);        // this clip is the next one to subtract
        ++gen;
    }
    if (gen > 0 && !(q.lab_skip & 1)) {
        sub.wait(sh, (gen - 1) & 1, (gen - 1) / 2 + 1, lane);
        sub.finish(wave);
    }
}

// Per-feature normalisation of the NeMo frontend (normalize_per_feature, src/mel.rs:721-749): for every (clip, mel) row the
// mean over the valid frames, the unbiased variance, (v - mean) / (sqrt(var) + 1e-5) -- in the reference's f32 and in the
// reference's order: `iter().sum::<f32>()` is a left fold, and its rounding error in the mean (~1e-4 for 1000 values near
// -10) divided by a small standard deviation is visible in the output (2e-3; a silent clip comes out as a constant
// 0.16 instead of 0).  A tree sum is more accurate and therefore different, so the sums run sequentially: a workgroup
// stages `rows_per_group` whole rows in LDS with coalesced loads, one lane per row folds its row left to right (twice),
// then all threads normalise and store.  rows_per_group == 0 (a row does not fit in LDS): one thread per row from HBM.
struct BlmNormParams {
    float *out;
    uint64_t clip_stride;   // floats between clips = n_mels * row_w
    uint64_t row_w;         // columns per row (padded frames)
    uint64_t valid;         // valid frames
    uint32_t n_clips;
    int n_mels;
    int rows_per_group;     // rows staged per workgroup round (<= 64), 0: rows too long for LDS
    int lds_stride;         // floats between staged rows: 4 * odd (16-byte aligned rows whose per-lane walks spread over the banks)
    int fold_sel;           // the wave that folds = (blockIdx.x >> fold_sel) & 3; < 0: wave 0
    int lab_skip;           // lab builds, timing ablations (wrong results): 1 no folds, 2 no stores, 4 no loads
    uint64_t *dbg;          // lab builds: [64][8] phase times, see MS_NORM_STAMP
    // ragged batches (rows_per_group == 0 form only): per clip the first output float, the row width and the valid frames
    const uint64_t *d_out_off, *d_cols, *d_valid;
};

constexpr int kBlmNormThreads = 256;

// lab builds: thread 0 of the first 64 workgroups adds up the time (100 MHz ticks) between the barriers of a round (MELSPEC_NORM_DBG)
#if defined(MELSPEC_LAB) && !defined(MELSPEC_NORM_NO_STAMPS)
#define MS_NORM_STAMP(k) do { if (p.dbg && tid == 0 && blockIdx.x < 64) { const uint64_t now = wall_clock64(); if ((k) > 0) p.dbg[blockIdx.x * 8 + (k)] += now - stamp; stamp = now; } } while (0)
#else
#define MS_NORM_STAMP(k) do { } while (0)
#endif
// Both normalisers run four 256-thread workgroups per CU (LDS-bound: four waves per SIMD), and the compiler is told so: without the
// attribute its scheduler minimises registers for an occupancy the kernels never have and SERIALISES the nine staging loads of a thread --
// one register quad, load / s_waitcnt vmcnt(0) / LDS write nine times over (uniform kernel 305-320 us instead of 253 for 1024 x 128 rows
// of 1001 frames; ragged, 5..15 s: 0.44 -> 0.34 ms).  Round 5 first met this as "the lab build is 20 % faster": any one of the lab
// build's disabled time stamps happened to flip the heuristic, while scheduling barriers between the loads and the writes keep the array
// of loaded values in scratch memory (350-375 us).  profiles/r05_norm_sched.txt has the whole trail.
#define MS_NORM_OCCUPANCY __attribute__((amdgpu_waves_per_eu(1, 4)))

__device__ __forceinline__ float *blm_row(const BlmNormParams &p, uint64_t row) {
    const uint64_t clip = row / p.n_mels, m = row - clip * p.n_mels;
    return p.out + clip * p.clip_stride + m * p.row_w;
}

// The mean of one row as the reference computes it: `iter().sum::<f32>() / n`, an f32 LEFT FOLD (src/mel.rs:721-749).  Its rounding
// error (~1e-4 for 1000 values near -10) divided by a small standard deviation is visible in the output, so the order is kept: a
// chain of `valid` dependent adds by one lane, and nothing else on its critical path -- the row is read 32 floats at a time (eight
// 16-byte reads) into two register sets filled in turn (a copy "cur = nxt" per group is one v_mov per element: as many
// instructions as the adds).  A lone wave issues one VALU instruction per ~5.6 cycles and a dependent add takes 10.5
// (tools/dep_add.hip): ~4.4 us per 1001-frame row.
// row: 16-byte aligned; the row's values are row[head .. head + valid), head < 4 (the piece of the 16-byte granule in front of the
// row belongs to its neighbour); readable up to the next multiple of 32 floats past head + valid (the excess is never added).
__device__ __forceinline__ float blm_row_mean_lds(const float *row, uint32_t head, uint32_t valid) {
    constexpr int kQ = 8;                      // float4s per group
    const uint32_t lo = head, hi = head + valid;
    const uint32_t groups = (hi + 4 * kQ - 1) / (4 * kQ);
    auto fetch = [&](uint32_t g, f4 (&v)[kQ]) {
#pragma unroll
        for (int i = 0; i < kQ; ++i) v[i] = *reinterpret_cast<const f4 *>(row + (g * kQ + i) * 4);
    };
    float s = 0.0f;
    auto consume = [&](const f4 (&c)[kQ], uint32_t g) {
        const uint32_t k0 = g * 4 * kQ;
        if (k0 >= lo && k0 + 4 * kQ <= hi) {
#pragma unroll
            for (int i = 0; i < kQ; ++i) { s += c[i].x; s += c[i].y; s += c[i].z; s += c[i].w; }
        } else {
#pragma unroll
            for (int i = 0; i < kQ; ++i) {
                const uint32_t k = k0 + 4 * i;
                if (k + 0 >= lo && k + 0 < hi) s += c[i].x;
                if (k + 1 >= lo && k + 1 < hi) s += c[i].y;
                if (k + 2 >= lo && k + 2 < hi) s += c[i].z;
                if (k + 3 >= lo && k + 3 < hi) s += c[i].w;
            }
        }
    };
    f4 a[kQ], b[kQ];
    fetch(0, a);
    uint32_t g = 0;
    for (; g + 1 < groups; g += 2) {
        fetch(g + 1, b);
        consume(a, g);
        fetch(g + 2 < groups ? g + 2 : g + 1, a);
        consume(b, g + 1);
    }
    if (g < groups) consume(a, g);
    return f32_div_rn(s, static_cast<float>(valid));
}

// the same from HBM, one value at a time (rows too long for LDS)
__device__ __forceinline__ void blm_row_stats_slow(const float *r, uint64_t valid, float &mean, float &sd) {
    float s = 0.0f;
    for (uint64_t k = 0; k < valid; ++k) s += r[k];
    mean = f32_div_rn(s, static_cast<float>(valid));
    float q = 0.0f;
    for (uint64_t k = 0; k < valid; ++k) {
        const float c = r[k] - mean;
        q += f32_mul_rn(c, c);
    }
    float denom = static_cast<float>(valid) - 1.0f;
    denom = denom < 1.0f ? 1.0f : denom;
    sd = __builtin_sqrtf(f32_div_rn(q, denom)) + 1e-5f;
}

#ifndef MELSPEC_TEMPLATE_KERNELS_ONLY
__global__ __launch_bounds__(kBlmNormThreads) MS_NORM_OCCUPANCY void blm_normalize_kernel(const BlmNormParams p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const uint64_t rows = (uint64_t)p.n_clips * p.n_mels;
    const int tid = threadIdx.x;
    if (p.rows_per_group == 0) {
        for (uint64_t row = (uint64_t)blockIdx.x * kBlmNormThreads + tid; row < rows; row += (uint64_t)gridDim.x * kBlmNormThreads) {
            float *r;
            uint64_t valid = p.valid;
            if (p.d_out_off) {
                const uint64_t clip = row / p.n_mels, m = row - clip * p.n_mels;
                r = p.out + p.d_out_off[clip] + m * p.d_cols[clip];
                valid = p.d_valid[clip];
                if (valid == 0) continue;
            } else {
                r = blm_row(p, row);
            }
            float mean, sd;
            blm_row_stats_slow(r, valid, mean, sd);
            for (uint64_t k = 0; k < valid; ++k) r[k] = f32_div_rn(r[k] - mean, sd);
        }
        return;
    }
    const int R = p.rows_per_group, S = p.lds_stride;
    float *stat = tile + (size_t)R * S;      // [R][2]
    const int fold_wave = p.fold_sel < 0 ? 0 : static_cast<int>((blockIdx.x >> p.fold_sel) & 3u);
    // Rows of one clip are contiguous and so are the clips (clip_stride == n_mels * row_w): row r starts at out + r * row_w, at
    // any 4-byte alignment (1001 columns for a 10 s clip without pad_to).  Global memory is accessed in whole 16-byte granules
    // all the same: a row whose first float sits `a` floats into its granule is staged from the granule's start, at the same
    // offset `a` in its 16-byte aligned LDS row; the granules a row shares with its neighbours are loaded by both and stored
    // float by float.  kRowsAtOnce rows in flight per thread (a load inside a per-row `if` would be one memory round trip per
    // row; rows past the group re-read its last row, granules past the row its last granule).
    constexpr int kRowsAtOnce = 9;
    const uint32_t valid = static_cast<uint32_t>(p.valid);
    const uint32_t out_f = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p.out) >> 2) & 3u;
    const uint32_t nq_max = (valid + 6) / 4;            // granules of a row at the worst alignment
    // A workgroup owns a contiguous range of rows and walks it in rounds of R.  (Starting the workgroups out of step -- a short
    // first round, a sleep per workgroup -- was measured: no effect; once its phases are cheap the pass is bandwidth-bound.)
    const uint64_t per_wg = (rows + gridDim.x - 1) / gridDim.x;
    const uint64_t row_begin = (uint64_t)blockIdx.x * per_wg;
    const uint64_t row_end = row_begin + per_wg < rows ? row_begin + per_wg : rows;
    float *part = stat + 2 * R;              // [R][PP] partial sums of squares
    uint64_t stamp = 0;
    (void)stamp;
    const int PP = kBlmNormThreads / R;      // threads per row in the variance pass
    for (uint64_t row0 = row_begin; row0 < row_end;) {
        MS_NORM_STAMP(0);
        const int nr = row_end - row0 < (uint64_t)R ? (int)(row_end - row0) : R;
        const uint64_t e00 = row0 * p.row_w;
        for (int rr0 = 0; rr0 < ((p.lab_skip & 4) ? 0 : nr); rr0 += kRowsAtOnce) {
            for (uint32_t q = tid; q < nq_max; q += kBlmNormThreads) {
                f4 v[kRowsAtOnce];
                uint32_t to[kRowsAtOnce];
                uint64_t e0 = e00 + (uint64_t)rr0 * p.row_w;
                uint32_t t = static_cast<uint32_t>(rr0) * S;
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) {
                    const uint32_t a = (out_f + static_cast<uint32_t>(e0)) & 3u;
                    const uint32_t nq = (a + valid + 3) >> 2;
                    const uint32_t qq = q < nq ? q : nq - 1;
                    v[i] = *reinterpret_cast<const f4 *>(p.out + e0 - a + 4 * qq);
                    to[i] = t + 4 * qq;
                    if (rr0 + i + 1 < nr) { e0 += p.row_w; t += S; }
                }
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) *reinterpret_cast<f4 *>(tile + to[i]) = v[i];
            }
        }
        __syncthreads();
        MS_NORM_STAMP(1);
        // the means: a few lanes of ONE wave (fold_sel: which one; measured without effect)
        const int ft = tid - 64 * fold_wave;
        if (ft >= 0 && ft < nr) {
            const uint32_t a = (out_f + static_cast<uint32_t>(e00 + (uint64_t)ft * p.row_w)) & 3u;
            MS_PRIO(3);                          // a chain of dependent adds: every issue slot it is ready for
            stat[2 * ft] = (p.lab_skip & 1) ? 0.0f : blm_row_mean_lds(tile + (size_t)ft * S, a, valid);
            MS_PRIO(0);
        }
        __syncthreads();
        MS_NORM_STAMP(2);
        // the unbiased variance: sum of (v - mean)^2 as a fixed tree over all threads, PP strided partial sums per row added in
        // order.  The reference folds this sum left to right as well; unlike the mean, the order is immaterial here -- either
        // sum is within ~1e-6 (relative) of the exact one, 5e-7 of the standard deviation, and the output moves by |out| * 5e-7.
        {
            const int r = tid / PP, pt = tid - r * PP;
            if (r < nr) {
                const uint32_t a = (out_f + static_cast<uint32_t>(e00 + (uint64_t)r * p.row_w)) & 3u;
                const float *row = tile + (size_t)r * S + a;
                const float mean = stat[2 * r];
                // four sums in turn: the strided loop has a run-time step, and with one accumulator every LDS read waited for
                // the add before it (2.1 us per round, measured with MS_NORM_STAMP; 36 values per thread at 1001 frames)
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                uint32_t k = pt;
                for (; k + 3 * PP < valid; k += 4 * PP) {
                    const float c0 = row[k] - mean, c1 = row[k + PP] - mean, c2 = row[k + 2 * PP] - mean, c3 = row[k + 3 * PP] - mean;
                    a0 += c0 * c0; a1 += c1 * c1; a2 += c2 * c2; a3 += c3 * c3;
                }
                for (; k < valid; k += PP) {
                    const float c = row[k] - mean;
                    a0 += c * c;
                }
                part[r * PP + pt] = (a0 + a1) + (a2 + a3);
            }
        }
        __syncthreads();
        MS_NORM_STAMP(3);
        if (tid < nr) {
            const float *pp = part + tid * PP;
            float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
            int i = 0;
            for (; i + 3 < PP; i += 4) { q0 += pp[i]; q1 += pp[i + 1]; q2 += pp[i + 2]; q3 += pp[i + 3]; }
            for (; i < PP; ++i) q0 += pp[i];
            const float q = (q0 + q1) + (q2 + q3);
            float denom = static_cast<float>(valid) - 1.0f;
            denom = denom < 1.0f ? 1.0f : denom;
            // the row's values are multiplied by 1 / (std + 1e-5) below: within one ulp of the reference's division, 9 divisions
            // per round instead of 36 per thread (the divisions were 4.7 us of a 16 us round)
            const float sd = __builtin_sqrtf(f32_div_rn(q, denom)) + 1e-5f;
            stat[2 * tid + 1] = (p.lab_skip & 1) ? 1.0f : f32_div_rn(1.0f, sd);
        }
        __syncthreads();
        MS_NORM_STAMP(4);
        const uint32_t row_w = static_cast<uint32_t>(p.row_w);
        for (int rr0 = 0; rr0 < ((p.lab_skip & 2) ? 0 : nr); rr0 += kRowsAtOnce) {
            for (uint32_t q = tid; q < nq_max; q += kBlmNormThreads) {
                f4 v[kRowsAtOnce];
                float mean[kRowsAtOnce], rsd[kRowsAtOnce];
                uint32_t t = static_cast<uint32_t>(rr0) * S + 4 * q;
                const float *st = stat + 2 * rr0;
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) {           // every LDS read first (rows past the group: its last row again)
                    v[i] = *reinterpret_cast<const f4 *>(tile + t);
                    mean[i] = st[0]; rsd[i] = st[1];
                    if (rr0 + i + 1 < nr) { t += S; st += 2; }
                }
                uint64_t e0 = e00 + (uint64_t)rr0 * p.row_w;
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) {
                    const uint32_t a = (out_f + static_cast<uint32_t>(e0)) & 3u;
                    const int c0 = static_cast<int>(4 * q) - static_cast<int>(a);       // column of the granule's first float
                    float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                 // columns past the valid frames keep their zeros
                        const float nv = (o[e] - mean[i]) * rsd[i];
                        o[e] = (c0 + e >= 0 && static_cast<uint32_t>(c0 + e) < valid) ? nv : 0.0f;
                    }
                    float *g = p.out + e0 + c0;
                    const bool mine = rr0 + i < nr && 4 * q < a + valid;                 // granules that hold valid frames of a row of the group
                    if (mine) {
                        if (c0 >= 0 && static_cast<uint32_t>(c0 + 3) < row_w) {
                            f4 w = {o[0], o[1], o[2], o[3]};
                            *reinterpret_cast<f4 *>(g) = w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c0 + e >= 0 && static_cast<uint32_t>(c0 + e) < row_w) g[e] = o[e];
                        }
                    }
                    if (rr0 + i + 1 < nr) e0 += p.row_w;
                }
            }
        }
        __syncthreads();
        MS_NORM_STAMP(5);
        row0 += nr;
    }
}
#endif  // MELSPEC_TEMPLATE_KERNELS_ONLY

// The same pass for ragged batches (clips of different lengths in one launch): rows are described per clip (first output float, row
// width, valid frames), a group of R rows is taken from a device counter (rows of long and short clips cost differently, so a static
// split would leave workgroups idle), its rows' descriptions are put in LDS once per round, and every row is staged at ITS alignment.
// Rows without valid frames are left alone.  LDS rows are sized for the longest clip of the batch.
struct BlmNormRaggedParams {
    float *out;
    const uint64_t *d_out_off, *d_cols, *d_valid;   // per clip
    uint32_t n_clips;
    int n_mels;
    int rows_per_group, lds_stride;
    uint32_t longest;       // valid frames of the longest clip
    unsigned *ctr;          // zero at launch
};

#ifndef MELSPEC_TEMPLATE_KERNELS_ONLY
__global__ __launch_bounds__(kBlmNormThreads) MS_NORM_OCCUPANCY void blm_normalize_ragged_kernel(const BlmNormRaggedParams p) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const uint64_t rows = (uint64_t)p.n_clips * p.n_mels;
    const int tid = threadIdx.x;
    const int R = p.rows_per_group, S = p.lds_stride;
    float *stat = tile + (size_t)R * S;      // [R][2]
    float *part = stat + 2 * R;              // [R][PP]
    uint32_t *info = reinterpret_cast<uint32_t *>(part + kBlmNormThreads);     // [R][4]: first float (lo, hi), valid frames, row width
    uint32_t *next = info + 4 * R;
    const int PP = kBlmNormThreads / R;
    constexpr int kRowsAtOnce = 9;
    const uint32_t out_f = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p.out) >> 2) & 3u;
    const f4 *out_base = reinterpret_cast<const f4 *>(p.out - out_f);        // the 16-byte granule `out` starts in
    for (;;) {
        if (tid == 0) next[0] = atomicAdd(p.ctr, 1u);
        __syncthreads();
        const uint64_t row0 = (uint64_t)next[0] * R;
        if (row0 >= rows) break;
        const int nr = rows - row0 < (uint64_t)R ? (int)(rows - row0) : R;
        if (tid < nr) {
            const uint64_t row = row0 + tid, clip = row / p.n_mels, m = row - clip * p.n_mels;
            const uint64_t cols = p.d_cols[clip], e0 = p.d_out_off[clip] + m * cols;
            info[4 * tid] = static_cast<uint32_t>(e0);
            info[4 * tid + 1] = static_cast<uint32_t>(e0 >> 32);
            info[4 * tid + 2] = static_cast<uint32_t>(p.d_valid[clip]);
            info[4 * tid + 3] = static_cast<uint32_t>(cols);
        }
        __syncthreads();
        // granules of the longest row OF THIS GROUP (round 5: both copy loops ran to the longest row of the batch -- clips of 5..15 s
        // made a third of their iterations re-read and re-write a short row's last granule)
        uint32_t gmax = 0;
        for (int rr = 0; rr < nr; ++rr) gmax = info[4 * rr + 2] > gmax ? info[4 * rr + 2] : gmax;
        const uint32_t nq_grp = gmax ? (gmax + 6) / 4 : 0;
        for (int rr0 = 0; rr0 < nr; rr0 += kRowsAtOnce) {
            for (uint32_t q = tid; q < nq_grp; q += kBlmNormThreads) {
                f4 v[kRowsAtOnce];
                uint32_t to[kRowsAtOnce];
                uint64_t from[kRowsAtOnce];          // float index of the granule (from the 16-byte aligned base of `out`)
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) {
                    const int rr = rr0 + i < nr ? rr0 + i : nr - 1;
                    const uint64_t e0 = ((uint64_t)info[4 * rr + 1] << 32) | info[4 * rr];
                    const uint32_t valid = info[4 * rr + 2];
                    const uint32_t a = (out_f + static_cast<uint32_t>(e0)) & 3u;
                    const uint32_t nq = (a + valid + 3) >> 2;
                    const uint32_t qq = q < nq ? q : (nq ? nq - 1 : 0);
                    from[i] = valid ? out_f + e0 - a + 4 * qq : 0;       // a row without frames may own no memory at all: the first granule instead
                    to[i] = static_cast<uint32_t>(rr) * S + 4 * qq;
                }
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) v[i] = out_base[from[i] >> 2];
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) *reinterpret_cast<f4 *>(tile + to[i]) = v[i];
            }
        }
        __syncthreads();
        if (tid < nr) {
            const uint32_t valid = info[4 * tid + 2];
            const uint32_t a = (out_f + info[4 * tid]) & 3u;
            MS_PRIO(3);
            stat[2 * tid] = valid ? blm_row_mean_lds(tile + (size_t)tid * S, a, valid) : 0.0f;
            MS_PRIO(0);
        }
        __syncthreads();
        {
            const int r = tid / PP, pt = tid - r * PP;
            if (r < nr) {
                const uint32_t valid = info[4 * r + 2];
                const uint32_t a = (out_f + info[4 * r]) & 3u;
                const float *row = tile + (size_t)r * S + a;
                const float mean = stat[2 * r];
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                uint32_t k = pt;
                for (; k + 3 * PP < valid; k += 4 * PP) {
                    const float c0 = row[k] - mean, c1 = row[k + PP] - mean, c2 = row[k + 2 * PP] - mean, c3 = row[k + 3 * PP] - mean;
                    a0 += c0 * c0; a1 += c1 * c1; a2 += c2 * c2; a3 += c3 * c3;
                }
                for (; k < valid; k += PP) {
                    const float c = row[k] - mean;
                    a0 += c * c;
                }
                part[r * PP + pt] = (a0 + a1) + (a2 + a3);
            }
        }
        __syncthreads();
        if (tid < nr) {
            const float *pp = part + tid * PP;
            float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
            int i = 0;
            for (; i + 3 < PP; i += 4) { q0 += pp[i]; q1 += pp[i + 1]; q2 += pp[i + 2]; q3 += pp[i + 3]; }
            for (; i < PP; ++i) q0 += pp[i];
            float denom = static_cast<float>(info[4 * tid + 2]) - 1.0f;
            denom = denom < 1.0f ? 1.0f : denom;
            const float sd = __builtin_sqrtf(f32_div_rn((q0 + q1) + (q2 + q3), denom)) + 1e-5f;
            stat[2 * tid + 1] = f32_div_rn(1.0f, sd);
        }
        __syncthreads();
        for (int rr0 = 0; rr0 < nr; rr0 += kRowsAtOnce) {
            for (uint32_t q = tid; q < nq_grp; q += kBlmNormThreads) {
#pragma unroll
                for (int i = 0; i < kRowsAtOnce; ++i) {
                    const int rr = rr0 + i < nr ? rr0 + i : nr - 1;
                    const uint64_t e0 = ((uint64_t)info[4 * rr + 1] << 32) | info[4 * rr];
                    const uint32_t valid = info[4 * rr + 2], row_w = info[4 * rr + 3];
                    const uint32_t a = (out_f + static_cast<uint32_t>(e0)) & 3u;
                    const bool mine = rr0 + i < nr && valid != 0 && 4 * q < a + valid;
                    const f4 v = *reinterpret_cast<const f4 *>(tile + static_cast<uint32_t>(rr) * S + 4 * q);
                    const float mean = stat[2 * rr], rsd = stat[2 * rr + 1];
                    const int c0 = static_cast<int>(4 * q) - static_cast<int>(a);
                    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float nv = (o[e] - mean) * rsd;
                        o[e] = (c0 + e >= 0 && static_cast<uint32_t>(c0 + e) < valid) ? nv : 0.0f;
                    }
                    float *g = p.out + e0 + c0;
                    if (mine) {
                        if (c0 >= 0 && static_cast<uint32_t>(c0 + 3) < row_w) {
                            f4 w = {o[0], o[1], o[2], o[3]};
                            *reinterpret_cast<f4 *>(g) = w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c0 + e >= 0 && static_cast<uint32_t>(c0 + e) < row_w) g[e] = o[e];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}
#endif  // MELSPEC_TEMPLATE_KERNELS_ONLY

// ------------------------------------------------------------------------------------
// The mel stage on its own: MelSpectrogram::add(&fft) (src/mel.rs:13-32) = SparseMelFilterbank::project_stft_log10
// (src/mel.rs:148-168) + norm_mel_slice_f64 (src/mel.rs:645-654) for callers that hold complex STFT frames (their own, or
// melspec_stft_*'s): E[m] = sum over the row's contiguous bins of w * |X[bin]|^2 in ascending bin order (bins >= n_fft/2
// contribute nothing), log10(max(E, 1e-10)), max - 8 clamp, (x + 4) / 4 -- the reference's f64 arithmetic step by step.
// One frame per 64-thread wave of a workgroup; spectra as interleaved (re, im) of float or double, `stride` complex per frame.
// ------------------------------------------------------------------------------------
struct MelStageParams {
    const void *spec;
    float *out;
    uint64_t n_frames;
    uint32_t stride;        // complex elements per frame (n_fft/2 + 1 or n_fft)
    int bin_limit;          // n_fft / 2
    int n_mels;
    const int *d_mstart, *d_mlen, *d_moff;
    const double *d_mw;
    const double *d_jw;     // the same bank as jobs of eight weights (build_mel_jobs, melspec_hip.hip), for mel_stage_jobs_kernel
    const int *d_job;
    int n_jobs;
};

// mel_stage_jobs_kernel: a frame per wave, the bank as jobs in LDS -- the mel phase of pow2_frame_kernel (section 4.3b of DESIGN.md)
// on spectra that come from memory.  The first form (mel_stage_kernel below, kept for banks the tables of this one do not take) read
// every weight from global memory inside a loop whose trip count is the band's width, a lane per mel, and took an f64 log10: 26 %
// of its roofline.  Here: the frame's bins in up to kMelStageBinLoads coalesced loads per lane, the NEXT frame's issued before this
// frame is worked on; jobs of eight weights, two rounds in flight, one ds_add_f64 per job; v_log_f32 like every fused kernel.
constexpr int kMelStageWaves = 8;
constexpr int kMelStageBinLoads = 4;       // 64 lanes x 4: frames of up to 256 bins take the unrolled path
struct MelStageLds { int jw, job, frames, frame_stride, acc, total; };
MS_HD MelStageLds mel_stage_lds(int n_jobs, int bin_limit, int n_mels, int waves) {
    MelStageLds o;
    o.jw = 0;
    o.job = 8 * n_jobs;
    o.frames = (o.job + (n_jobs + 1) / 2 + 31) & ~31;
    o.acc = (bin_limit + 8 + 1) & ~1;                  // a frame: the power row (+ 8 a job may read past it), the band sums
    o.frame_stride = (o.acc + n_mels + 31) & ~31;
    o.total = o.frames + waves * o.frame_stride;
    return o;
}

template <class T>
__global__ __launch_bounds__(kMelStageWaves * 64) void mel_stage_jobs_kernel(const MelStageParams p) {
    extern __shared__ __attribute__((aligned(16))) double stage_lds[];
    struct alignas(2 * sizeof(T)) T2 { T re, im; };
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const MelStageLds at = mel_stage_lds(p.n_jobs, p.bin_limit, p.n_mels, kMelStageWaves);
    double *ljw = stage_lds + at.jw;
    int *ljob = reinterpret_cast<int *>(stage_lds + at.job);
    for (int i = tid; i < 8 * p.n_jobs; i += kMelStageWaves * 64) ljw[i] = p.d_jw[i];
    for (int i = tid; i < p.n_jobs; i += kMelStageWaves * 64) ljob[i] = p.d_job[i];
    double *pw = stage_lds + at.frames + wave * at.frame_stride, *acc = pw + at.acc;
    if (lane < 8) pw[p.bin_limit + lane] = 0.0;
    __syncthreads();
    const int n_jobs = p.n_jobs, bins = p.bin_limit;
    const bool small = bins <= 64 * kMelStageBinLoads;
    const uint64_t step = (uint64_t)gridDim.x * kMelStageWaves;
    uint64_t f = (uint64_t)blockIdx.x * kMelStageWaves + wave;
    if (f >= p.n_frames) return;
    auto fetch = [&](uint64_t frame, T2 (&v)[kMelStageBinLoads]) {
        const T2 *x = reinterpret_cast<const T2 *>(static_cast<const T *>(p.spec) + 2 * frame * p.stride);
#pragma unroll
        for (int i = 0; i < kMelStageBinLoads; ++i) { const int k = lane + 64 * i; v[i] = x[k < bins ? k : bins - 1]; }
    };
    T2 cur[kMelStageBinLoads];
    fetch(f, cur);
    for (;;) {
        const uint64_t nf = f + step;
        const bool more = nf < p.n_frames;                      // wave-uniform
        T2 nxt[kMelStageBinLoads];
        if (more) fetch(nf, nxt);
#pragma unroll
        for (int i = 0; i < kMelStageBinLoads; ++i) {
            const int k = lane + 64 * i;
            const double re = static_cast<double>(cur[i].re), im = static_cast<double>(cur[i].im);
            if (k < bins) pw[k] = re * re + im * im;                                   // norm_sqr, src/mel.rs:159
        }
        if (!small) {
            const T2 *x = reinterpret_cast<const T2 *>(static_cast<const T *>(p.spec) + 2 * f * p.stride);
            for (int k = lane + 64 * kMelStageBinLoads; k < bins; k += 64) {
                const double re = static_cast<double>(x[k].re), im = static_cast<double>(x[k].im);
                pw[k] = re * re + im * im;
            }
        }
        for (int m = lane; m < p.n_mels; m += 64) acc[m] = 0.0;
        for (int jb0 = lane; jb0 < n_jobs + lane; jb0 += 2 * 64) {            // wave-uniform trip count; ascending bins inside a job, src/mel.rs:155-163
            int info[2];
            d2 w[2][4];
            double pv[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t) info[t] = jb0 + 64 * t < n_jobs ? ljob[jb0 + 64 * t] : 0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int jb = jb0 + 64 * t;
                const double *wp = ljw + 2 * (jb < n_jobs ? jb : 0), *pp = pw + (info[t] & 0xfff);
#pragma unroll
                for (int q = 0; q < 4; ++q) w[t][q] = *reinterpret_cast<const d2 *>(wp + q * 2 * n_jobs);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const d2 two = *reinterpret_cast<const d2 *>(pp + 2 * q);
                    pv[t][2 * q] = two.x; pv[t][2 * q + 1] = two.y;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                double e = w[t][0].x * pv[t][0];
                e += w[t][0].y * pv[t][1]; e += w[t][1].x * pv[t][2]; e += w[t][1].y * pv[t][3];
                e += w[t][2].x * pv[t][4]; e += w[t][2].y * pv[t][5]; e += w[t][3].x * pv[t][6]; e += w[t][3].y * pv[t][7];
                if ((info[t] >> 20) > 0) unsafeAtomicAdd(acc + ((info[t] >> 12) & 0xff), e);
            }
        }
        // log10 through v_log_f32 (as the fused kernels), the frame's maximum, clamp, (x + 4) / 4 (src/mel.rs:166, 645-654)
        constexpr int kPer = 4;                                  // 256 mels
        float mv[kPer];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int m = lane + 64 * i;
            mv[i] = 0.0f;
            if (m < p.n_mels) {
                const double e = acc[m];
                mv[i] = fast_log2((float)(e > 1e-10 ? e : 1e-10)) * 0.30102999566398120f;
                mx = mx > mv[i] ? mx : mv[i];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(mx, o, 64); mx = mx > t ? mx : t; }
        const float lo = mx - 8.0f;
        float *o = p.out + f * (uint64_t)p.n_mels;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int m = lane + 64 * i;
            if (m < p.n_mels) o[m] = ((mv[i] > lo ? mv[i] : lo) + 4.0f) * 0.25f;
        }
        if (!more) break;
        f = nf;
#pragma unroll
        for (int i = 0; i < kMelStageBinLoads; ++i) cur[i] = nxt[i];
    }
}

template <class T, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mel_stage_kernel(const MelStageParams p) {
    extern __shared__ __attribute__((aligned(16))) double stage_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *pw = stage_lds + (size_t)wave * (p.bin_limit + p.n_mels);      // [bin_limit] powers, [n_mels] log values
    double *lv = pw + p.bin_limit;
    for (uint64_t f = (uint64_t)blockIdx.x * WAVES + wave; f < p.n_frames; f += (uint64_t)gridDim.x * WAVES) {
        const T *x = static_cast<const T *>(p.spec) + 2 * f * p.stride;
        for (int k = lane; k < p.bin_limit; k += 64) {
            const double re = static_cast<double>(x[2 * k]), im = static_cast<double>(x[2 * k + 1]);
            pw[k] = re * re + im * im;                                   // norm_sqr, src/mel.rs:159
        }
        __builtin_amdgcn_wave_barrier();
        double mx = -1.0e300;
        for (int m = lane; m < p.n_mels; m += 64) {
            const int st = p.d_mstart[m], len = p.d_mlen[m];
            const double *w = p.d_mw + p.d_moff[m];
            double e = 0.0;
            for (int i = 0; i < len; ++i) e += w[i] * pw[st + i];        // ascending bins, src/mel.rs:155-163
            const double v = log10(e > 1e-10 ? e : 1e-10);               // src/mel.rs:166
            lv[m] = v;
            mx = v > mx ? v : mx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(mx, o);
            mx = other > mx ? other : mx;
        }
        __builtin_amdgcn_wave_barrier();
        const double lo = mx - 8.0;                                      // src/mel.rs:645-654
        float *o = p.out + f * (uint64_t)p.n_mels;
        for (int m = lane; m < p.n_mels; m += 64) {
            const double v = lv[m];
            o[m] = static_cast<float>(((v > lo ? v : lo) + 4.0) / 4.0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------
// Generic kernel: any n_fft / hop / n_mels, Whisper or Kaldi-fbank flavour, one frame per
// workgroup iteration, direct DFT in f64 from an LDS twiddle table.  It follows the
// reference's f64 arithmetic step by step (src/stft.rs:119-138, src/fbank.rs:160-222) and
// exists for coverage and as the on-device cross-check of the fused f32 kernels; it is not
// a throughput path.
// ------------------------------------------------------------------------------------
struct GenericParams {
    BatchDesc b;           // units == frames
    int n_fft;             // DFT length
    int frame_len;         // non-zero samples per frame (== n_fft for Whisper)
    int hop;
    int n_bins;            // bins whose power is needed: n_fft/2 (Whisper) or n_fft/2+1 (fbank)
    int n_mels;
    int fbank;             // 0: Whisper log10 + per-frame norm; 1: Kaldi fbank; 2: NeMo BatchLogMelSpectrogram (src/mel.rs:321-385)
    int use_log, use_power;
    double preemph, floor_v;   // NeMo: preemph = the f32 coefficient, floor_v = log_zero_guard
    long long clip_len;    // NeMo (uniform batches): samples per clip
    int pad;               // NeMo: n_fft / 2 when centred (zero padding either side, src/mel.rs:685-694), else 0
    int fft_log2;          // log2(n_fft) when n_fft is a power of two >= 8: the transform is an in-LDS radix-2 FFT; 0: not
    FftPlan plan;          // n_rad > 0: mixed-radix in-LDS FFT (2-3-5-smooth n_fft that is not a power of two); both zero: direct DFT
    const double *d_win;   // [frame_len]
    const double *d_tw;    // [n_fft] interleaved (cos, -sin) of 2*pi*j/n_fft
    const int *d_mstart;   // [n_mels]
    const int *d_mlen;     // [n_mels]
    const int *d_moff;     // [n_mels] offset into d_mw
    const double *d_mw;    // concatenated spans
    int mw_count;          // doubles in d_mw
    // pow2_frame_kernel's view of the same bank: n_jobs jobs of eight consecutive weights of one mel (the last job of a band padded with
    // zeros), d_jw[2 * ((q / 2) * n_jobs + job) + (q & 1)] weight q of a job, d_job[job] = first bin | mel << 12 | count << 20 (count = 1..8 real entries)
    const double *d_jw;
    const int *d_job;
    int n_jobs;
};

template <int NT>
__device__ __forceinline__ double block_reduce(double v, double *red, bool is_max) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const double o = red[tid + s];
            red[tid] = is_max ? (red[tid] > o ? red[tid] : o) : (red[tid] + o);
        }
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

template <int NT>
__global__ __launch_bounds__(NT) void generic_frame_kernel(const GenericParams p) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    double *tw = ldsd;                       // 2*n_fft
    double *xw = tw + 2 * p.n_fft;           // frame_len (direct DFT) or n_fft (FFT: n_fft/2 complex points, bit-reversed)
    const bool mixed = p.plan.n_rad > 0;                              // mixed-radix FFT: xw = two buffers of n_fft complex points
    double *pw = xw + (mixed ? 4 * p.n_fft : (p.fft_log2 ? p.n_fft : p.frame_len));           // n_bins
    // FFT form (power-of-two n_fft): the real frame as n_fft/2 complex points z[n] = x[2n] + i x[2n+1], stored at the bit-reversed
    // index for the in-place decimation-in-time passes below; sample i goes to slot(i)
    const int mbits = p.fft_log2 - 1;
    auto slot = [&](int i) -> int {
        if (mixed) return 2 * i;
        if (!p.fft_log2) return i;
        const unsigned r = mbits > 0 ? (__brev(static_cast<unsigned>(i >> 1)) >> (32 - mbits)) : 0u;
        return static_cast<int>(2 * r) + (i & 1);
    };
    double *mv = pw + p.n_bins;              // n_mels
    double *red = mv + p.n_mels;             // NT
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * p.n_fft; i += NT) tw[i] = p.d_tw[i];

    const uint64_t n_units = batch_n_units(p.b);
    for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const UnitLoc loc = locate_unit(p.b, unit);
        if (loc.unit >= loc.frames) {       // zero column of a padded layout (uniform batches only)
            float *z = p.b.mel_major ? loc.out + loc.unit : loc.out + loc.unit * (uint64_t)p.n_mels;
            const uint64_t zstep = p.b.mel_major ? p.b.out_width : 1;
            for (int m = tid; m < p.n_mels; m += NT) z[m * zstep] = 0.0f;
            continue;
        }
        const uint64_t start = loc.unit * (uint64_t)p.hop;
        const float *x = loc.pcm + start;
        __syncthreads();
        if (!p.fbank) {
            // frame_windows: x[start+i] as f64 * window[i]   (src/stft.rs:160-165)
            for (int i = tid; i < p.frame_len; i += NT) xw[slot(i)] = (double)x[i] * p.d_win[i];
        } else if (p.fbank == 2) {
            // whole-clip pre-emphasis in f32 with the reference's two roundings (src/mel.rs:696-706), zero centre padding, window
            const float coeff = (float)p.preemph;
            for (int i = tid; i < p.frame_len; i += NT) {
                const long long sidx = (long long)start + i - p.pad;
                float v = 0.0f;
                if (sidx >= 0 && sidx < p.clip_len) {
                    v = loc.pcm[sidx];
                    if (coeff != 0.0f && sidx > 0) v = v - f32_mul_rn(coeff, loc.pcm[sidx - 1]);
                }
                xw[slot(i)] = (double)v * p.d_win[i];
            }
        } else {
            // DC removal, pre-emphasis, Povey window   (src/fbank.rs:164-190)
            double part = 0.0;
            for (int i = tid; i < p.frame_len; i += NT) part += (double)x[i];
            const double mean = block_reduce<NT>(part, red, false) / (double)p.frame_len;
            for (int i = tid; i < p.frame_len; i += NT) {
                double v = (double)x[i] - mean;
                if (p.preemph > 0.0) {
                    if (i > 0) v -= p.preemph * ((double)x[i - 1] - mean);
                    else if (start > 0) v -= p.preemph * ((double)*(x - 1) - mean);
                }
                xw[slot(i)] = v * p.d_win[i];
            }
        }
        if (mixed) {
            // imaginary parts and the zero padding, then one Stockham pass per factor of n_fft (lds_fft_mixed)
            for (int i = tid; i < p.n_fft; i += NT) {
                xw[2 * i + 1] = 0.0;
                if (i >= p.frame_len) xw[2 * i] = 0.0;
            }
            const double *res = lds_fft_mixed<NT>(p.plan, p.n_fft, tw, xw, xw + 2 * p.n_fft, tid);
            for (int k = tid; k < p.n_bins; k += NT) {
                const double re = res[2 * k], im = res[2 * k + 1];
                const double ns = re * re + im * im;
                pw[k] = (p.fbank && !p.use_power) ? sqrt(ns) : ns;
            }
        } else if (p.fft_log2) {
            // zero padding up to n_fft (frame_len < n_fft: Kaldi's 400 of 512), then log2(n_fft/2) radix-2 passes over the n_fft/2
            // complex points and the real-FFT split X[k] = E[k] + W_N^k O[k] -- O(N log N) instead of the O(N^2) direct form below
            for (int i = p.frame_len + tid; i < p.n_fft; i += NT) xw[slot(i)] = 0.0;
            const int M = p.n_fft >> 1;
            for (int len = 2; len <= M; len <<= 1) {
                __syncthreads();
                const int half = len >> 1, tstep = p.n_fft / len;
                for (int b = tid; b < (M >> 1); b += NT) {
                    const int g = b / half, j = b - g * half;
                    const int i0 = g * len + j, i1 = i0 + half;
                    const double c = tw[2 * (j * tstep)], sn = tw[2 * (j * tstep) + 1];     // W_len^j = W_N^{j N / len}
                    const double ur = xw[2 * i0], ui = xw[2 * i0 + 1];
                    const double xr = xw[2 * i1], xi = xw[2 * i1 + 1];
                    const double vr = xr * c - xi * sn, vi = xr * sn + xi * c;
                    xw[2 * i0] = ur + vr; xw[2 * i0 + 1] = ui + vi;
                    xw[2 * i1] = ur - vr; xw[2 * i1 + 1] = ui - vi;
                }
            }
            __syncthreads();
            for (int k = tid; k < p.n_bins; k += NT) {
                const int ka = k == M ? 0 : k, kb = (M - k) & (M - 1);        // Z[M] = Z[0]; partner Z[M - k]
                const double ar = xw[2 * ka], ai = xw[2 * ka + 1];
                const double br = xw[2 * kb], bi = -xw[2 * kb + 1];           // conj
                const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);     // E = (A + B) / 2
                const double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);     // O = -i (A - B) / 2 = (di, -dr)
                const double c = tw[2 * k], sn = tw[2 * k + 1];
                const double orr = di, oi = -dr;
                const double re = er + (orr * c - oi * sn), im = ei + (orr * sn + oi * c);
                const double ns = re * re + im * im;
                pw[k] = (p.fbank && !p.use_power) ? sqrt(ns) : ns;
            }
        } else {
            __syncthreads();
            for (int k = tid; k < p.n_bins; k += NT) {
                double re = 0.0, im = 0.0;
                int idx = 0;
                for (int n = 0; n < p.frame_len; ++n) {
                    const double c = tw[2 * idx], s = tw[2 * idx + 1];
                    re += xw[n] * c;
                    im += xw[n] * s;
                    idx += k;
                    if (idx >= p.n_fft) idx -= p.n_fft;
                }
                const double ns = re * re + im * im;
                pw[k] = (p.fbank && !p.use_power) ? sqrt(ns) : ns;
            }
        }
        __syncthreads();
        double mx = -1.0e300;
        for (int m = tid; m < p.n_mels; m += NT) {
            const int st = p.d_mstart[m], len = p.d_mlen[m];
            const double *w = p.d_mw + p.d_moff[m];
            double e = 0.0;
            for (int r = 0; r < len; ++r) e += w[r] * pw[st + r];
            double v;
            if (!p.fbank) {
                v = log10(e > 1e-10 ? e : 1e-10);          // src/mel.rs:166
            } else if (p.fbank == 2) {
                v = log(e + p.floor_v);                    // src/mel.rs:365-368
            } else {
                v = e > p.floor_v ? e : p.floor_v;           // src/fbank.rs:210-218
                if (p.use_log) v = log(v);
            }
            mv[m] = v;
            mx = mx > v ? mx : v;
        }
        const uint64_t width = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
        float *o = p.b.mel_major ? loc.out + loc.unit : loc.out + loc.unit * (uint64_t)p.n_mels;
        const uint64_t ostep = p.b.mel_major ? width : 1;
        if (!p.fbank) {
            const double lo = block_reduce<NT>(mx, red, true) - 8.0;   // src/mel.rs:645-654
            for (int m = tid; m < p.n_mels; m += NT) {
                const double v = mv[m] > lo ? mv[m] : lo;
                o[m * ostep] = (float)((v + 4.0) / 4.0);
            }
        } else if (p.fbank == 2) {
            for (int m = tid; m < p.n_mels; m += NT) o[m * ostep] = (float)mv[m];      // feature-major rows (src/mel.rs:366)
        } else {
            for (int m = tid; m < p.n_mels; m += NT) o[m] = (float)mv[m];
        }
    }
}

// ------------------------------------------------------------------------------------
// pow2_frame_kernel: the power-of-two frame sizes on wave-owned frames (pow2_wave.hpp).  Same parameters, flavours, tables and
// results contract as generic_frame_kernel; units == frames.  LDS (doubles): [tw: W_N^q, q < M][WAVES x FW x frame region].
// ------------------------------------------------------------------------------------

// The samples a lane needs for one frame, as they come from memory: pairs (x[2n], x[2n + 1]) of its P complex points and, for the
// flavours with pre-emphasis, the sample in front of each pair.  Loaded one frame AHEAD of their use (the next frame's loads are in
// flight while this frame's transform runs): at two or three waves per SIMD nothing else hides a 1-2 us HBM round trip.
template <int P, int FLAVOR> struct Pow2Raw {
    f2 pair[P];
    float before[FLAVOR == 0 ? 1 : P];
};

template <int LOGM, int FLAVOR>
__global__ __launch_bounds__(Pow2Shape<LOGM>::kMaxWaves * 64) void pow2_frame_kernel(const GenericParams p) {
    using S = Pow2Shape<LOGM>;
    constexpr int M = S::M, LF = S::LF, FW = S::FW, P = S::P;
    constexpr bool kAhead = (P == 8 && !(FLAVOR == 1 && LF < 64 && !MS_POW2_AHEAD_KS)) || (P == 16 && MS_POW2_AHEAD16 && !(S::kHalves && !MS_POW2_AHEADH));   // the next frame's samples are loaded while this one is transformed
    constexpr bool kWinLds = M <= MS_POW2_WINLDS;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x, n_threads = blockDim.x, n_waves = n_threads >> 6;     // the host picks the waves per workgroup (LDS)
    const Pow2Lds at = pow2_lds<LOGM>(p.n_jobs, p.n_mels, n_waves);
    double *tw = ldsd + at.tw;                           // W_N^q, q <= M / 2 (the split's twiddles)
    double *lwin = ldsd + at.win;                        // 2 * M: the window, zero from frame_len on (M <= 256)
    double *t2 = ldsd + at.t2, *t3 = ldsd + at.t3;       // the twiddles of passes 2 and 3, [r - 1][k]
    double *ljw = ldsd + at.jw;                          // the banded filterbank as jobs of eight weights (GenericParams::d_jw), then the
    int *ljob = reinterpret_cast<int *>(ldsd + at.job);  //   job records {first bin | mel << 12 | count << 20}
    for (int i = tid; i < M + 2; i += n_threads) tw[i] = p.d_tw[i];
    if (kWinLds) for (int i = tid; i < 2 * M; i += n_threads) lwin[i] = p.d_win[i];
    for (int i = tid; i < S::kT2; i += n_threads) stc(t2 + 2 * i, pow2_table_entry(p.d_tw, M, 8, S::R1, i));
    constexpr bool kHalves = S::kHalves;
    for (int i = tid; i < S::kT3; i += n_threads) stc(t3 + 2 * i, pow2_table_entry(p.d_tw, M, kHalves ? 8 : (S::R3 > 1 ? S::R3 : 2), kHalves ? 64 : S::R1 * 8, i));
    double *tc = ldsd + at.tc;                           // kHalves: W_M^k = W_N^{2k}, k < M / 2
    for (int i = tid; i < S::kTc; i += n_threads) stc(tc + 2 * i, pow2_root(p.d_tw, 2 * i, M));
    for (int i = tid; i < 8 * p.n_jobs; i += n_threads) ljw[i] = p.d_jw[i];
    for (int i = tid; i < p.n_jobs; i += n_threads) ljob[i] = p.d_job[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int fs = lane / LF, l = lane - fs * LF;        // frame slot of the wave, lane of the frame
    double *z = ldsd + at.frames + (wave * FW + fs) * at.frame_stride;
    double *pw = z + at.pw + pow2_pw_shift<LOGM>(fs);    // [M + 1]
    double *acc = z + at.acc;                            // [n_mels] band energies of the frame
    constexpr bool kPwAlias = M >= MS_POW2_PWALIAS;
    if (!kPwAlias && l < 8) pw[M + 1 + l] = 0.0;         // what the last job of the top band reads past the row
    const int last = p.frame_len - 1;

    // the twiddles of pass 2 depend on the lane only: W_{8 R1}^{k r}, k = l mod R1 (kept in registers; pass 3's come from the LDS table)
    constexpr bool kTw2Reg = (P == 8 && MS_POW2_TW2REG) || kHalves;
    cpx<double> tw2[kTw2Reg ? 7 : 1];
    // complex point r of the lane: l + r LF, or (kHalves) point l + 64 (r / 2) of the even (r even) / odd half: 2 (l + 64 (r / 2)) + (r & 1)
    auto pt = [&](int r) { return kHalves ? 2 * (l + 64 * (r >> 1)) + (r & 1) : l + r * LF; };
    if (kHalves) {
        const int k2 = l & 7;
#pragma unroll
        for (int r = 1; r < 8; ++r) tw2[r - 1] = pow2_root(p.d_tw, r * k2 * (2 * M / 64), M);
    } else if (kTw2Reg) {
        const int k2 = l & (S::R1 - 1);
#pragma unroll
        for (int r = 1; r < 8; ++r) tw2[r - 1] = pow2_root(p.d_tw, r * k2 * (2 * M / (S::R1 * 8)), M);
    }

    struct Frame {                                       // where a frame is, per lane group
        const float *pcm, *x;
        float *o;
        uint64_t ostep, start;
        // tag = the frame's unit inside its clip (uniform batches; < 2^30) | have << 30 | real << 31, and the clip.  The flags were two
        // `bool` members: with byte-sized members the tail of the struct is not split into registers -- `Frame nxt = cur` and `cur = nxt`
        // went through scratch memory, a load / s_waitcnt vmcnt(0) / store pair at both ends of every iteration of the frame loop, each
        // wait also draining the loads issued ahead for the next frame (found by tools/hotloop_spills.py, round 5).  As two 32-bit words
        // they cost the 1024- and 2048-point instances (at 256 VGPRs) more than the scratch copy did (+1..3 %; n_fft 256 -5 %); packed:
        // n_fft 128 -3 %, 256 -4 %, Kaldi 32 kHz -2 %, 1024 / 2048 unchanged (same box, profiles/r05_pow2.txt).
        uint32_t tag, clip;
        __device__ __forceinline__ bool have() const { return (tag >> 30) & 1u; }
        __device__ __forceinline__ bool real() const { return (tag >> 31) != 0; }
        __device__ __forceinline__ uint32_t in_clip() const { return tag & 0x3fffffffu; }
    };
    const uint64_t n_units = batch_n_units(p.b);
    const uint64_t stride = (uint64_t)gridDim.x * n_waves * FW;
    const bool uniform = p.b.d_unit_prefix == nullptr;
    const bool walk = uniform && p.b.units_per_clip < (1u << 29);     // Frame::tag holds the unit inside its clip in 30 bits
    auto frame_at = [&](const UnitLoc &loc, bool have) {
        Frame f;
        const uint64_t width = uniform ? p.b.out_width : loc.frames;
        f.o = p.b.mel_major ? loc.out + loc.unit : loc.out + loc.unit * (uint64_t)p.n_mels;
        f.ostep = p.b.mel_major ? width : 1;
        const bool real = have && loc.unit < loc.frames;     // otherwise: a zero column of a padded layout (uniform batches), or nothing
        f.start = loc.unit * (uint64_t)p.hop;
        f.pcm = loc.pcm;
        f.x = loc.pcm + f.start;
        f.tag = ((uint32_t)loc.unit & 0x3fffffffu) | (have ? 1u << 30 : 0u) | (real ? 1u << 31 : 0u);
        f.clip = loc.clip;
        return f;
    };
    auto place = [&](uint64_t base) {
        const uint64_t unit = base + fs;
        const bool have = unit < n_units;
        return frame_at(locate_unit(p.b, have ? unit : base), have);
    };
    // The frame `stride` units further on.  locate_unit divides a 64-bit unit index by the units of a clip -- ~100 VALU instructions per
    // lane, a sixth of this kernel's at n_fft 256 when it was done per frame; a uniform batch is walked instead: the step in whole
    // clips and the rest are the same for every lane and every iteration.
    const uint64_t step_clips = uniform ? stride / p.b.units_per_clip : 0;
    const uint32_t step_rest = uniform ? (uint32_t)(stride - step_clips * p.b.units_per_clip) : 0;
    auto advance = [&](const Frame &f, uint64_t nbase) {
        if (!walk || !f.have()) return place(nbase);
        UnitLoc loc;
        uint32_t u = f.in_clip() + step_rest;             // (< 2 units_per_clip <= 2^32: the host plans uniform batches with 32-bit unit counts per clip)
        uint64_t c = (uint64_t)f.clip + step_clips;
        if (u >= p.b.units_per_clip) { u -= p.b.units_per_clip; ++c; }
        loc.unit = u;
        loc.clip = (uint32_t)c;
        loc.pcm = p.b.pcm + c * p.b.clip_stride;
        loc.out = p.b.out + c * p.b.out_stride;
        loc.frames = p.b.frames_per_clip;
        return frame_at(loc, nbase + fs < n_units);
    };
    // Every load is unconditional (clamped index, the value selected afterwards): a load behind its own branch is a serialised memory
    // round trip, and the first form of this kernel -- one predicate per sample -- spent 80 % of its time in them.
    // part: 2 = every point; 0 / 1 (kHalves): the even / the odd points only (r = 2 r' + part)
    auto fetch = [&](const Frame &f, Pow2Raw<P, FLAVOR> &raw, int part = 2) {
        if (!f.real()) return;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            if (part != 2 && (r & 1) != part) continue;
            const int i0 = 2 * pt(r);
            if (FLAVOR == 2) {                           // sample s of the frame = clip[start + s - pad], zero outside the clip
                const long long s0 = (long long)f.start + i0 - p.pad, hi = p.clip_len - 1;
                const long long c0 = s0 < 0 ? 0 : (s0 > hi ? hi : s0), c1 = s0 + 1 < 0 ? 0 : (s0 + 1 > hi ? hi : s0 + 1);
                raw.pair[r] = f2{f.pcm[c0], f.pcm[c1]};
                raw.before[r] = f.pcm[c0 > 0 ? c0 - 1 : 0];
            } else {
                const int pi = i0 + 1 <= last ? i0 : (last >= 1 ? last - 1 : 0);     // never past the frame's last sample
                raw.pair[r] = load2_unaligned(f.x + pi);
                if (FLAVOR == 1) raw.before[r] = (pi > 0 || f.start > 0) ? f.x[pi - 1] : f.x[0];
            }
        }
    };

    // the jobs of a lane are the same for every frame too: the records of its first kJ stay in registers
    const int n_jobs = p.n_jobs;
    constexpr int kJ = (FLAVOR != 0 && (P == 8 || S::kHalves)) ? MS_POW2_JOBS_F : (LF < 64 ? MS_POW2_JOBS_SMALL : (S::kHalves ? MS_POW2_JOBS_H : MS_POW2_JOBS_BIG));     // rounds of jobs in flight together (the Kaldi / NeMo framings hold more registers: spills)
    int info0[kJ];
#pragma unroll
    for (int t = 0; t < kJ; ++t) info0[t] = l + t * LF < n_jobs ? ljob[l + t * LF] : 0;

    uint64_t base = ((uint64_t)blockIdx.x * n_waves + wave) * FW;
    if (base >= n_units) return;
    Frame cur = place(base);
    constexpr bool kFetchPerHalf = S::kHalves && !kAhead;      // the samples of a half are loaded when the half is framed (registers)
    Pow2Raw<P, FLAVOR> raw;
    if (!kFetchPerHalf) fetch(cur, raw);
    for (;;) {
        const uint64_t nbase = base + stride;
        const bool more = nbase < n_units;               // wave-uniform
        Frame nxt = cur;
        Pow2Raw<P, FLAVOR> nraw;
        if (kAhead && more) {
            nxt = advance(cur, nbase);
            fetch(nxt, nraw);
        }
        if (cur.have() && !cur.real()) {
            for (int m = l; m < p.n_mels; m += LF) cur.o[m * cur.ostep] = 0.0f;
        }
        if (cur.real()) {
            // ---- framing: DC removal / pre-emphasis / window per flavour -> the lane's P complex points z[l + r LF] -----------------
            // point(r): the windowed complex point r of the lane.  FLAVOR 1 needs the frame's mean first.
            double mean = 0.0;
            if (FLAVOR == 1) {                           // src/fbank.rs:164-170
                if (kFetchPerHalf) fetch(cur, raw);
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const int i0 = 2 * pt(r);
                    const double xa = (double)(i0 + 1 <= last ? raw.pair[r].x : raw.pair[r].y);     // i0 == last: the clamped pair holds x[last] second
                    part += i0 <= last ? xa : 0.0;
                    part += i0 + 1 <= last ? (double)raw.pair[r].y : 0.0;
                }
                // (the frame's lanes are all inside this branch or all outside it: a frame owns a whole lane group)
#pragma unroll
                for (int d = 1; d < LF; d <<= 1) part += __shfl_xor(part, d, 64);
                mean = part / (double)p.frame_len;
            }
            auto point = [&](int r, d2 w) {
                double va, vb;
                if (FLAVOR == 0) {                       // frame_windows: x[start + i] as f64 (src/stft.rs:160-165)
                    va = (double)raw.pair[r].x; vb = (double)raw.pair[r].y;
                } else if (FLAVOR == 2) {                // src/mel.rs:696-706: whole-clip pre-emphasis in f32, two roundings; zero centre padding
                    const float coeff = (float)p.preemph;
                    const long long s0 = (long long)cur.start + 2 * pt(r) - p.pad;
                    const float a = raw.pair[r].x, b0 = raw.pair[r].y;
                    const float pa = a - f32_mul_rn(coeff, raw.before[r]), pb = b0 - f32_mul_rn(coeff, a);
                    const float fa = (coeff != 0.0f && s0 > 0) ? pa : a, fb = (coeff != 0.0f && s0 + 1 > 0) ? pb : b0;
                    va = (s0 >= 0 && s0 < p.clip_len) ? (double)fa : 0.0;
                    vb = (s0 + 1 >= 0 && s0 + 1 < p.clip_len) ? (double)fb : 0.0;
                } else {                                 // src/fbank.rs:171-190: DC removal, pre-emphasis
                    const int i0 = 2 * pt(r);
                    const double xa = (double)(i0 + 1 <= last ? raw.pair[r].x : raw.pair[r].y), xb = (double)raw.pair[r].y;
                    va = xa - mean; vb = xb - mean;
                    if (p.preemph > 0.0) {
                        vb -= p.preemph * (xa - mean);
                        // the sample in front of xa: the clamped pair of an odd frame's last sample (i0 == last) holds it first (ADVICE r04:
                        // raw.before is then x[last - 2]; the Povey window's last tap is 0, so no test could see it)
                        const double xp = (double)(i0 + 1 <= last ? raw.before[r] : raw.pair[r].x);
                        if (i0 > 0 || cur.start > 0) va -= p.preemph * (xp - mean);
                    }
                }
                return cpx<double>{va * w.x, vb * w.y};
            };
            auto window_of = [&](int r) { return *reinterpret_cast<const d2 *>((kWinLds ? lwin : p.d_win) + 2 * pt(r)); };
            // ---- the complex M-point transform: Stockham passes, in place in the frame's LDS region --------------------------------
            if (kHalves) {
                // the even and the odd points as two 512-point transforms, E at z, O behind it
                // (eight points at a time, window values and all: sixteen at once are the registers of the one-transform form)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    d2 wv[8];
                    cpx<double> half[8];
                    if (kFetchPerHalf) fetch(cur, raw, h);      // (Kaldi: again, after the pass for the mean -- holding all sixteen pairs spills more than the reload costs)
#pragma unroll
                    for (int r = 0; r < 8; ++r) wv[r] = window_of(2 * r + h);
#pragma unroll
                    for (int r = 0; r < 8; ++r) half[r] = point(2 * r + h, wv[r]);
                    double *zh = z + h * M;
                    pow2_pass<9, 8, true>(l, 1, nullptr, zh, half, nullptr);
                    pow2_pass<9, 8, false>(l, 8, t3, zh, nullptr, tw2);          // (t3: any table; the twiddles are tw2)
                    pow2_pass<9, 8, false>(l, 64, t3, zh, nullptr, nullptr);
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            } else {
                cpx<double> reg[P];
                d2 wv[P];
#pragma unroll
                for (int r = 0; r < P; ++r) wv[r] = window_of(r);
#pragma unroll
                for (int r = 0; r < P; ++r) reg[r] = point(r, wv[r]);
                pow2_pass<LOGM, S::R1, true>(l, 1, nullptr, z, reg, nullptr);
                pow2_pass<LOGM, 8, false>(l, S::R1, t2, z, nullptr, kTw2Reg ? tw2 : nullptr);
                if (S::R3 > 1) pow2_pass<LOGM, (S::R3 > 1 ? S::R3 : 2), false>(l, S::R1 * 8, t3, z, nullptr, nullptr);
            }
            // ---- the real-FFT split X[k] = E[k] + W_N^k O[k], E = (Z[k] + conj Z[M-k]) / 2, O = -i (Z[k] - conj Z[M-k]) / 2, and the
            // power row.  X[M-k] comes from the same two points (E -> conj E, O -> conj O, W_N^{M-k} = -conj W_N^k):
            //   X[k] = (er + t1) + i (ei + t2),   X[M-k] = (er - t1) - i (ei - t2),   t1 = di c + dr s,  t2 = di s - dr c
            // so a lane takes the pairs k = l + r LF < M / 2 (k = 0 gives bins 0 and M); bin M / 2 is its own partner.
            auto power2 = [&](int k, double &lo, double &hi) {
                const cpx<double> a = ldc(z + 2 * pow2_slot<LOGM>(k)), b0 = ldc(z + 2 * pow2_slot<LOGM>((M - k) & (M - 1)));
                const cpx<double> w = ldc(tw + 2 * k);
                const double er = 0.5 * (a.re + b0.re), ei = 0.5 * (a.im - b0.im);
                const double dr = 0.5 * (a.re - b0.re), di = 0.5 * (a.im + b0.im);
                const double t1 = di * w.re + dr * w.im, t2v = di * w.im - dr * w.re;
                const double ar = er + t1, ai = ei + t2v, br = er - t1, bi = ei - t2v;
                lo = ar * ar + ai * ai;
                hi = br * br + bi * bi;
            };
            // kHalves: Z[k] = E[k] + W_M^k O[k] and Z[M - k] = Z[(M/2 - k) + M/2] = E[M/2 - k] + conj(W_M^k) O[M/2 - k] (W_M^{M/2 - k} =
            // -conj W_M^k) are formed on the way in: the radix-2 step costs no pass of its own.  Bin M / 2: Z[M/2] = E[0] - O[0].
            auto pair_h = [&](cpx<double> a, cpx<double> b0, int k, double &lo, double &hi) {
                const cpx<double> w = ldc(tw + 2 * k);
                const double er = 0.5 * (a.re + b0.re), ei = 0.5 * (a.im - b0.im);
                const double dr = 0.5 * (a.re - b0.re), di = 0.5 * (a.im + b0.im);
                const double t1 = di * w.re + dr * w.im, t2v = di * w.im - dr * w.re;
                const double ar = er + t1, ai = ei + t2v, br = er - t1, bi = ei - t2v;
                lo = ar * ar + ai * ai;
                hi = br * br + bi * bi;
            };
            auto power2h = [&](int k, double &lo, double &hi) {
                const int km = (M / 2 - k) & (M / 2 - 1);
                const cpx<double> ek = ldc(z + 2 * pow2_slot<9>(k)), ok = ldc(z + M + 2 * pow2_slot<9>(k));
                const cpx<double> em = ldc(z + 2 * pow2_slot<9>(km)), om = ldc(z + M + 2 * pow2_slot<9>(km));
                const cpx<double> wc = ldc(tc + 2 * k);
                const cpx<double> wo = cmul(wc, ok), wm = cmul(cpx<double>{wc.re, -wc.im}, om);
                pair_h(cpx<double>{ek.re + wo.re, ek.im + wo.im}, cpx<double>{em.re + wm.re, em.im + wm.im}, k, lo, hi);
            };
            double plo[P / 2], phi[P / 2];
            double pmid, pmid2;
            __builtin_amdgcn_wave_barrier();                // the split's reads stay behind the last pass's writes ...
            if (kHalves) {
#pragma unroll
                for (int r = 0; r < P / 2; ++r) {
                    power2h(l + r * LF, plo[r], phi[r]);
#if defined(__HIP_DEVICE_COMPILE__)
                    if (r % MS_POW2_HSPLIT == MS_POW2_HSPLIT - 1) __builtin_amdgcn_sched_barrier(0);      // (a pair is five 16-byte loads: all eight at once are 160 registers)
#endif
                }
                const cpx<double> e0 = ldc(z), o0 = ldc(z + M);              // slot(0) = 0
                pair_h(cpx<double>{e0.re - o0.re, e0.im - o0.im}, cpx<double>{e0.re - o0.re, e0.im - o0.im}, M / 2, pmid, pmid2);
            } else {
#pragma unroll
                for (int r = 0; r < P / 2; ++r) power2(l + r * LF, plo[r], phi[r]);
                power2(M / 2, pmid, pmid2);              // every lane, one address: a broadcast
            }
            // magnitudes instead of powers (FbankConfig::use_power off): ONE wave-uniform branch around all the square roots -- as a select
            // inside the pair the compiler evaluated the 2 (P / 2 + 1) IEEE f64 roots of every frame unconditionally (~300 instructions, a
            // third of the Kaldi flavour's arithmetic; the same trap as in fb_phase2_split)
            if (FLAVOR == 1 && !p.use_power) {
#pragma unroll
                for (int r = 0; r < P / 2; ++r) { plo[r] = sqrt(plo[r]); phi[r] = sqrt(phi[r]); }
                pmid = sqrt(pmid);
            }
            __builtin_amdgcn_wave_barrier();                // ... and in front of the power row's writes, which alias the points at M >= 1024
#pragma unroll
            for (int r = 0; r < P / 2; ++r) {
                pw[l + r * LF] = plo[r];
                pw[M - (l + r * LF)] = phi[r];
            }
            if (l == 0) pw[M / 2] = pmid;
            if (kPwAlias && l < 8) pw[M + 1 + l] = 0.0;      // (the row is where the points were)
        }
        // ---- banded mel sums, log, per-flavour epilogue ---------------------------------------------------------------------------
        // The bank as JOBS of eight consecutive weights of one mel (the last job of a band padded): every lane takes a job per round,
        // folds its up-to-eight products left to right and adds the partial sum to the mel's word in LDS (ds_add_f64; the LDS executes
        // a wave's operations in program order and an instruction's lanes in lane order, so the result is the same on every run).  No
        // trip count depends on a band's width, every load is unconditional, all lanes are busy: a mel per lane and step (the form
        // before) left most lanes idle while the lanes of the wide high bands walked 25 bins, with one LDS round trip per tail bin --
        // it was 37-43 % of the kernel.  A band's energy is the reference's left fold (src/mel.rs:155-163) cut into <= 4 pieces.
        // (A first balanced form -- the same number of consecutive ENTRIES per lane, an atomic at every mel boundary inside a lane's
        // range -- had 20 divergent branch sites per frame and was slower than the mel-per-lane form: 3.5 against 2.05 ms.)
        constexpr int kMaxPer = Pow2Shape<LOGM>::kMelsPerLane;            // a lane reads out the mels m = l + LF i (the host checks n_mels <= kMelsPerLane * LF)
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) if (l + LF * i < p.n_mels) acc[l + LF * i] = 0.0;
        // kJ = three rounds at a time: their 36 loads are in flight together (the transform's registers are free here), one LDS round trip
        // instead of three -- at two waves per SIMD the kernel is a chain of such round trips, not of arithmetic
        auto job_triple = [&](const int (&info)[kJ], int jb0) {
            d2 w[kJ][4];
            double pv[kJ][8];
            const int jstep = 2 * n_jobs;
#pragma unroll
            for (int t = 0; t < kJ; ++t) {
                const int jb = jb0 + t * LF;
                const double *wp = ljw + 2 * (jb < n_jobs ? jb : 0), *pp = pw + (info[t] & 0xfff);      // weights 2 q, 2 q + 1 of job j at [q][j]: consecutive lanes, consecutive slots
#pragma unroll
                for (int q = 0; q < 4; ++q) w[t][q] = *reinterpret_cast<const d2 *>(wp + q * jstep);
#pragma unroll
                for (int q = 0; q < 4; ++q) {                      // a job starts at an even bin: four aligned ds_read_b128
                    const d2 two = *reinterpret_cast<const d2 *>(pp + 2 * q);
                    pv[t][2 * q] = two.x; pv[t][2 * q + 1] = two.y;
                }
            }
#pragma unroll
            for (int t = 0; t < kJ; ++t) {
                // (a job's weights past its count are +0 and what it reads past the row is +0: e + 0 * p = e, no selects)
                double e = w[t][0].x * pv[t][0];
                e += w[t][0].y * pv[t][1]; e += w[t][1].x * pv[t][2]; e += w[t][1].y * pv[t][3];
                e += w[t][2].x * pv[t][4]; e += w[t][2].y * pv[t][5]; e += w[t][3].x * pv[t][6]; e += w[t][3].y * pv[t][7];
                if (cur.real() && (info[t] >> 20) > 0) unsafeAtomicAdd(acc + ((info[t] >> 12) & 0xff), e);       // (count 0: the host's padding, or past the last job)
            }
        };
        job_triple(info0, l);
        for (int jb0 = l + kJ * LF; jb0 < n_jobs + l; jb0 += kJ * LF) {            // wave-uniform trip count
            int info[kJ];
#pragma unroll
            for (int t = 0; t < kJ; ++t) info[t] = jb0 + t * LF < n_jobs ? ljob[jb0 + t * LF] : 0;
            job_triple(info, jb0);
        }
        // log2 through v_log_f32 (1 ulp: <= 1.2e-6 of a log10 / ln value, 3e-7 after Whisper's / 4), like the fused kernels
        float mv[kMaxPer];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int m = l + LF * i;
            mv[i] = 0.0f;
            if (cur.real() && m < p.n_mels) {
                const double e = acc[m];
                float vv;
                if (FLAVOR == 0) {
                    vv = fast_log2((float)(e > 1e-10 ? e : 1e-10)) * 0.30102999566398120f;          // src/mel.rs:166
                } else if (FLAVOR == 2) {
                    vv = fast_log2((float)(e + p.floor_v)) * 0.69314718055994531f;                  // src/mel.rs:365-368
                } else {
                    const float t = (float)(e > p.floor_v ? e : p.floor_v);                           // src/fbank.rs:210-218
                    vv = p.use_log ? fast_log2(t) * 0.69314718055994531f : t;
                }
                mv[i] = vv;
                mx = mx > vv ? mx : vv;
            }
        }
        if (FLAVOR == 0) {                                     // src/mel.rs:645-654: clamp at the frame's maximum - 8, (x + 4) / 4
#pragma unroll
            for (int d = 1; d < LF; d <<= 1) { const float t = __shfl_xor(mx, d, 64); mx = mx > t ? mx : t; }
            const float lo = mx - 8.0f;
#pragma unroll
            for (int i = 0; i < kMaxPer; ++i) {
                const int m = l + LF * i;
                if (cur.real() && m < p.n_mels) cur.o[m * cur.ostep] = ((mv[i] > lo ? mv[i] : lo) + 4.0f) * 0.25f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kMaxPer; ++i) {
                const int m = l + LF * i;
                if (cur.real() && m < p.n_mels) cur.o[m * cur.ostep] = mv[i];
            }
        }
        if (!more) break;
        base = nbase;
        if (kAhead) {
            cur = nxt;
            raw = nraw;
        } else {
            cur = advance(cur, base);
            if (!kFetchPerHalf) fetch(cur, raw);
        }
    }
}

// CMN (src/fbank.rs:224-233): per clip and mel column subtract the f32 mean over the clip's frames.  The reference's
// `column(m).mean()` (ndarray on a strided view) is a left fold in f32 followed by one division; its rounding error is ~1e-5 of a
// feature value at 1000 frames.  The column sum here is the FIXED TREE of fbank512_clip_kernel (which cannot afford a serial fold
// inside the producing kernel) -- so that a clip's output bits do not depend on which of the two kernels its batch was given to,
// i.e. on the batch it is in (round 2: the two orders differed by up to 1.6e-5):
//   units of 4 frames; eight contiguous runs of units, run w = [units*w/8, units*(w+1)/8);
//   S[w][p] = left fold, from +0, of the values of frame position p = frame & 3 over the run's units (frames past the clip's end add +0);
//   run sum = (S[w][0] + S[w][1]) + (S[w][2] + S[w][3]);   sum = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));   mean = sum / frames.
// It depends on the clip's frame count only, sits within ~2e-5 of the left fold (tests gate both at 1e-4 against the oracle) and gives
// the fold four independent chains instead of one.
// One workgroup per clip: all 512 threads stage the clip's rows in LDS, a chunk of up to rows_per_chunk (a multiple of 4) at a time
// (coalesced 16-byte loads, every load of a chunk in flight together), lanes m < n_mels fold the chunk from LDS, and when the whole
// clip has been folded every thread subtracts -- the last chunk straight from its LDS copy, the earlier ones re-read (L2 / Infinity
// Cache).  rows_per_chunk == 0 (no staging): the columns are folded from global memory, for banks wider than the staging allows.
struct CmnTree {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    uint32_t w = 0;
    uint64_t units, bound;          // bound: first unit of run w + 1
    float *part;                    // this column's eight run sums, stride `pstride`
    int pstride;
    __device__ __forceinline__ CmnTree(uint64_t frames, float *part_, int pstride_) : units((frames + 3) / 4), part(part_), pstride(pstride_) { bound = units / 8; }
    __device__ __forceinline__ void close() {
        part[w * pstride] = (s0 + s1) + (s2 + s3);
        s0 = s1 = s2 = s3 = 0.0f;
        ++w;
        bound = units * (w + 1) / 8;
    }
    // the four frames of unit u (values past the clip's end: +0)
    __device__ __forceinline__ void unit(uint64_t u, float v0, float v1, float v2, float v3) {
        while (u >= bound) close();                 // runs may be empty (fewer than eight units)
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    __device__ __forceinline__ float finish() {
        while (w < 8) close();
        const float *q = part;
        const int t = pstride;
        return ((q[0] + q[t]) + (q[2 * t] + q[3 * t])) + ((q[4 * t] + q[5 * t]) + (q[6 * t] + q[7 * t]));
    }
};

struct CmnParams {
    BatchDesc b;   // only the clip geometry is used
    int n_mels;
    int rows_per_chunk;
};

template <int NT>
__global__ __launch_bounds__(NT) void cmn_kernel(const CmnParams p) {
    extern __shared__ __attribute__((aligned(16))) float cmn_lds[];
    const int nm = p.n_mels;
    const int tid = threadIdx.x;
    const int R = p.rows_per_chunk;
    const int nmp = (nm + 3) & ~3;
    float *mean_s = cmn_lds;                 // [nmp]
    float *part_s = cmn_lds + nmp;           // the eight run sums of every column: [8][nmp] (staged form) / [8][NT]
    float *rows = part_s + 8 * (R > 0 ? nmp : NT);
    for (uint32_t clip = blockIdx.x; clip < p.b.n_clips; clip += gridDim.x) {
        float *o;
        uint64_t frames;
        if (p.b.d_unit_prefix == nullptr) {
            o = p.b.out + (uint64_t)clip * p.b.out_stride;
            frames = p.b.frames_per_clip;
        } else {
            o = p.b.out + p.b.d_out_off[clip];
            frames = p.b.d_frames[clip];
        }
        if (frames == 0) continue;
        if (R > 0) {
            CmnTree tree(frames, part_s + tid, nmp);
            uint64_t f0 = 0;
            const bool vec = ((reinterpret_cast<uintptr_t>(o) & 15) == 0) && (nm % 4 == 0);
            for (;; f0 += R) {
                const int nr = frames - f0 < (uint64_t)R ? (int)(frames - f0) : R;
                const float *src = o + f0 * nm;
                const int total = nr * nm;
                __syncthreads();                                   // the previous chunk has been folded
                if (vec) {
                    // eight 16-byte loads per thread in flight (a plain copy loop leaves one: ~40 memory round trips per chunk)
                    constexpr int kU = 8;
                    const int nq = total / 4;
                    for (int q0 = tid; q0 < nq; q0 += NT * kU) {
                        f4 v[kU];
#pragma unroll
                        for (int k = 0; k < kU; ++k) {
                            const int q = q0 + k * NT;
                            v[k] = *reinterpret_cast<const f4 *>(src + 4 * (q < nq ? q : q0));
                        }
#pragma unroll
                        for (int k = 0; k < kU; ++k) {
                            const int q = q0 + k * NT;
                            if (q < nq) *reinterpret_cast<f4 *>(rows + 4 * q) = v[k];
                        }
                    }
                } else {
                    for (int i = tid; i < total; i += NT) rows[i] = src[i];
                }
                __syncthreads();
                if (tid < nm) {
                    // chunks start at multiples of 4 frames (rows_per_chunk is one): whole units, then the clip's last, partial unit
                    const float *col = rows + tid;
                    const uint64_t ub = f0 / 4;
                    int r = 0;
                    for (; r + 16 <= nr; r += 16) {
                        float v[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = col[(r + i) * nm];
#pragma unroll
                        for (int i = 0; i < 4; ++i) tree.unit(ub + (r >> 2) + i, v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
                    }
                    for (; r + 4 <= nr; r += 4) tree.unit(ub + (r >> 2), col[r * nm], col[(r + 1) * nm], col[(r + 2) * nm], col[(r + 3) * nm]);
                    if (r < nr)
                        tree.unit(ub + (r >> 2), col[r * nm], r + 1 < nr ? col[(r + 1) * nm] : 0.0f, r + 2 < nr ? col[(r + 2) * nm] : 0.0f, 0.0f);
                }
                if (f0 + nr >= frames) break;
            }
            if (tid < nm) mean_s[tid] = f32_div_rn(tree.finish(), (float)frames);
            __syncthreads();
            // the last chunk from LDS, the earlier ones from memory
            const int nr = (int)(frames - f0);
            const int G = NT / nm;
            const int g = tid / nm, m = tid - g * nm;
            if (g < G) {
                const float mean = mean_s[m];
                for (int r = g; r < nr; r += G) o[(f0 + r) * nm + m] = rows[r * nm + m] - mean;
                // earlier chunks: 8 rows per thread in flight
                uint64_t f = g;
                for (; f + 7 * (uint64_t)G < f0; f += 8 * (uint64_t)G) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = o[(f + k * (uint64_t)G) * nm + m];
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[(f + k * (uint64_t)G) * nm + m] = v[k] - mean;
                }
                for (; f < f0; f += G) o[f * nm + m] -= mean;
            }
            __syncthreads();                                       // mean_s / rows are reused by the next clip
            continue;
        }
        for (int m0 = 0; m0 < nm; m0 += NT) {                 // column chunks when n_mels > NT
            const int cols = nm - m0 < NT ? nm - m0 : NT;
            const int G = NT / cols;                           // frame groups per column
            const int g = tid / cols, m = m0 + tid - g * cols;
            if (tid < cols) {
                constexpr int kB = 16;
                const float *col = o + m0 + tid;
                CmnTree tree(frames, part_s + tid, NT);
                uint64_t f = 0;
                for (; f + kB <= frames; f += kB) {
                    float v[kB];
#pragma unroll
                    for (int i = 0; i < kB; ++i) v[i] = col[(f + i) * nm];
#pragma unroll
                    for (int i = 0; i < kB / 4; ++i) tree.unit(f / 4 + i, v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
                }
                for (; f + 4 <= frames; f += 4) tree.unit(f / 4, col[f * nm], col[(f + 1) * nm], col[(f + 2) * nm], col[(f + 3) * nm]);
                if (f < frames)
                    tree.unit(f / 4, col[f * nm], f + 1 < frames ? col[(f + 1) * nm] : 0.0f, f + 2 < frames ? col[(f + 2) * nm] : 0.0f, 0.0f);
                rows[tid] = f32_div_rn(tree.finish(), (float)frames);
            }
            __syncthreads();
            if (g < G) {
                const float mean = rows[tid - g * cols];
                for (uint64_t f = g; f < frames; f += G) o[f * nm + m] -= mean;
            }
            __syncthreads();
        }
    }
}

// ---- streaming state (Spectrogram::add, src/stft.rs:48-86; RingBuffer::maybe_mel, src/rb.rs:86-121) ----
// Every live stream owns a slot of `stride` floats: [carry, right-aligned so that it ends at `in_off`]
// [the next chunk, always at in_off].  The carry is the reference's hop_buf history (n_fft - hop samples)
// plus the samples RingBuffer has accumulated towards the next hop (< hop).  Frames are computed in place
// by the batch kernels on carry ++ chunk; afterwards the tail of that span becomes the new carry.
// copies host-pushed chunks (one flat staging buffer) into the slots; optionally zero-pads (flush)
#ifndef MELSPEC_TEMPLATE_KERNELS_ONLY
__global__ __launch_bounds__(256) void stream_scatter_kernel(float *state, uint64_t stride, uint32_t in_off, const StreamEntry *entries,
                                                             const float *src) {
    const StreamEntry e = entries[blockIdx.x];
    float *dst = state + e.stream * stride + in_off;
    if (src)
        for (uint32_t i = threadIdx.x; i < e.len; i += 256) dst[i] = src[e.src_off + i];
    for (uint32_t i = threadIdx.x; i < e.zero_fill; i += 256) dst[e.len + i] = 0.0f;
}
#endif  // MELSPEC_TEMPLATE_KERNELS_ONLY

// new carry = the last `keep` samples before in_off + len (+ zero_fill), moved so that they end at in_off:
// a shift to lower addresses by the chunk length.  Ascending 256-sample pieces, each read completely
// before it is written, never touch the source of a later piece.
#ifndef MELSPEC_TEMPLATE_KERNELS_ONLY
__global__ __launch_bounds__(256) void stream_carry_kernel(float *state, uint64_t stride, uint32_t in_off, const StreamEntry *entries) {
    const StreamEntry e = entries[blockIdx.x];
    const uint32_t n = e.len + e.zero_fill;
    if (n == 0) return;
    float *slot = state + e.stream * stride;
    const float *src = slot + in_off + n - e.keep;
    float *dst = slot + in_off - e.keep;
    for (uint32_t base = 0; base < e.keep; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const float v = i < e.keep ? src[i] : 0.0f;
        __syncthreads();
        if (i < e.keep) dst[i] = v;
        __syncthreads();
    }
}
#endif  // MELSPEC_TEMPLATE_KERNELS_ONLY

// Ragged batch whose descriptors live in device memory (melspec_*_ragged_device_desc): the plan the host builds for
// melspec_compute_ragged_device (plan_ragged in melspec_hip.hip), built by one workgroup instead -- per-clip frame counts,
// the prefix of units per clip, packed output offsets when none are given, the clip of every 16th unit -- so that a caller
// whose clip table is produced on the GPU (a VAD, a segmenter) never copies it back.  Layout of `plan` as plan_ragged's:
// [off n][frames n][out_off n][prefix n+1] u64, then the block table (u32).
struct PlanParams {
    const uint64_t *d_off, *d_len, *d_out_off;     // d_out_off may be null: outputs packed in clip order
    uint32_t n_clips;
    uint64_t frame_len, frame_shift;               // frames(n) = n < frame_len ? 0 : (n - frame_len) / frame_shift + 1
    uint32_t words_per_frame;                      // output words (floats) per frame
    uint32_t frames_per_unit;
    uint64_t *plan;
    uint64_t max_blocks;                           // capacity of the block table
};

#ifndef MELSPEC_TEMPLATE_KERNELS_ONLY
__global__ __launch_bounds__(1024) void plan_ragged_device_kernel(const PlanParams q) {
    __shared__ uint64_t part_units[1024], part_out[1024];
    const uint32_t n = q.n_clips, tid = threadIdx.x;
    uint64_t *off = q.plan, *fr = off + n, *oo = fr + n, *pre = oo + n;
    uint32_t *blk = reinterpret_cast<uint32_t *>(pre + n + 1);
    const uint32_t per = (n + 1023) / 1024, c0 = tid * per, c1 = c0 + per < n ? c0 + per : n;
    uint64_t su = 0, so = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const uint64_t len = q.d_len[c];
        const uint64_t f = len < q.frame_len ? 0 : (len - q.frame_len) / q.frame_shift + 1;
        off[c] = q.d_off[c];
        fr[c] = f;
        su += (f + q.frames_per_unit - 1) / q.frames_per_unit;
        so += f * q.words_per_frame;
    }
    part_units[tid] = su; part_out[tid] = so;
    __syncthreads();
    if (tid == 0) {                                  // 1024 partials: a serial scan is 2 us
        uint64_t au = 0, ao = 0;
        for (int i = 0; i < 1024; ++i) {
            const uint64_t u = part_units[i], o = part_out[i];
            part_units[i] = au; part_out[i] = ao;
            au += u; ao += o;
        }
        pre[n] = au;
    }
    __syncthreads();
    su = part_units[tid]; so = part_out[tid];
    for (uint32_t c = c0; c < c1; ++c) {
        const uint64_t f = fr[c];
        const uint64_t u = (f + q.frames_per_unit - 1) / q.frames_per_unit;
        pre[c] = su;
        oo[c] = q.d_out_off ? q.d_out_off[c] : so;
        // the clip of every 16th unit inside [su, su + u)
        for (uint64_t k = (su + kUnitBlock - 1) / kUnitBlock; k * kUnitBlock < su + u && k < q.max_blocks; ++k) blk[k] = c;
        su += u;
        so += f * q.words_per_frame;
    }
}
#endif  // MELSPEC_TEMPLATE_KERNELS_ONLY

// Hash-noise PCM (murmur3 finaliser) of SURVEY.md §8(d); the CPU tests regenerate the same bits.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

#ifndef MELSPEC_TEMPLATE_KERNELS_ONLY
__global__ __launch_bounds__(256) void synth_pcm_kernel(float *out, uint64_t clip_stride, uint64_t clip_len,
                                                        uint64_t first_clip, uint32_t n_clips, uint32_t seed, uint64_t first_sample) {
    const uint64_t total = (uint64_t)n_clips * clip_len;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (uint64_t)gridDim.x * 256) {
        const uint64_t c = g / clip_len, i = g - c * clip_len;
        const uint64_t clip = first_clip + c;
        const uint32_t h = fmix32(seed ^ ((uint32_t)clip * 0x9E3779B1u) ^ ((uint32_t)(first_sample + i) * 0x85EBCA6Bu));
        const float u = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        out[c * clip_stride + i] = u * (1.0f / (float)(1u << (clip & 7u)));
    }
}
#endif  // MELSPEC_TEMPLATE_KERNELS_ONLY

}  // namespace melspec
