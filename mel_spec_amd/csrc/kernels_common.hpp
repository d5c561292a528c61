// kernels_common.hpp -- what every kernel family of libmelspec_hip.so shares: the batch description and the unit -> clip mapping, the
// XCD-aware workgroup order, the statistics sink and the vote of MELSPEC_PRECISION_AUTO, wave-wide helpers, the sub-group barrier of the
// mel-major stores.  The kernels themselves live in whisper400_kernels.hpp, fbank512_kernels.hpp, generic_kernels.hpp and
// aux_kernels.hpp, one translation unit each (mel_spec_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "whisper_fast.hpp"

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
        if (v == 0 || (v & kVoteHeavy)) return;       // heavy: the f64 launch reports; still unknown after the bounded wait: nobody reports this
                                                      // launch (the host keeps the previous figures) rather than both launches counting it
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

// ballot of wave_phase4's result -> one bit per frame of the unit (LANES lanes per frame)
template <int LANES, int FRAMES>
__device__ __forceinline__ unsigned frame_mask(uint64_t any) {
    unsigned m = 0;
#pragma unroll
    for (int f = 0; f < FRAMES; ++f) m |= ((any >> (LANES * f)) & ((1ull << LANES) - 1)) ? (1u << f) : 0u;
    return m;
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
    // the clip that holds `unit` (unit < the batch's units)
    __device__ __forceinline__ void place(const BatchDesc &b) {
        if (b.d_unit_prefix == nullptr) {
            clip = static_cast<uint32_t>(unit / b.units_per_clip);
            load_clip(b);
            return;
        }
        clip = __builtin_amdgcn_readfirstlane(b.d_unit_block[unit / kUnitBlock]);
        c_end = scalar64(b.d_unit_prefix[clip + 1]);
        while (c_end <= unit) { ++clip; c_end = scalar64(b.d_unit_prefix[clip + 1]); }     // prefix[n_clips] = n_units > unit
        load_clip(b);
    }
    // false: this wave has no units
    __device__ __forceinline__ bool init(const BatchDesc &b, uint64_t wave_id, uint64_t waves) {
        const uint64_t nu = b.d_n_units ? ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(*b.d_n_units >> 32)) << 32 |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)*b.d_n_units)) : b.n_units;
        const uint64_t run = (nu + waves - 1) / waves;
        unit = wave_id * run;
        end = unit + run < nu ? unit + run : nu;
        if (unit >= end) return false;
        place(b);
        return true;
    }
    // jump to another unit of the wave's run (the walkers of a note list: w512_auto_kernel's gated launch); u < end
    __device__ __forceinline__ void seek(const BatchDesc &b, uint64_t u) {
        unit = u;
        place(b);
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

constexpr int kWaveWaves = 8;     // waves per workgroup of the 5-frame n_fft = 400 kernels (two workgroups per CU)
constexpr int kPreciseWaves = 8;  // ... of whisper400_precise_kernel / whisper400_stft_kernel (one workgroup per CU)

// radix plan of the mixed-radix in-LDS FFT of the generic kernels (lds_fft_mixed, generic_kernels.hpp); built on the host (GenericTables)
struct FftPlan {
    int n_rad;
    unsigned long long packed;      // four bits per pass, first pass lowest (an array in the kernel arguments, indexed by the pass, would be
                                    // copied to scratch)
};

}  // namespace melspec
