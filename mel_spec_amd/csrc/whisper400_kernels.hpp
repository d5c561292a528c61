// whisper400_kernels.hpp -- the fused n_fft = 400 kernels (phases in whisper_wave.hpp / whisper_six.hpp / whisper_wave_f64.hpp /
// whisper_six64.hpp / whisper_fix64.hpp): five and six frames per wave, f32 with the precision guard, f64, the STFT export.
#pragma once
#include "kernels_common.hpp"
#include "whisper_wave.hpp"
#include "whisper_wave_f64.hpp"
#include "whisper_six.hpp"
#include "whisper_six64.hpp"
#include "whisper_fix64.hpp"

namespace melspec {

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
#define MS_SIX_LAYOUT_WAVES kSixWaves
#include "whisper400_six_body.inc"
#undef MS_SIX_LAYOUT_WAVES
}
// ... with fifteen mel slots on twelve waves (168 VGPRs): the layouts of Whisper large-v3's 128-mel bank (see whisper400_six_wide_runs_kernel)
template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kSixWideWaves * 64, 3) void whisper400_six_wide_kernel(const FastParams p) {
#define MS_SIX_LAYOUT_WAVES kSixWideWaves
#include "whisper400_six_body.inc"
#undef MS_SIX_LAYOUT_WAVES
}

// Plain [frame][mel] output, uniform and ragged batches, on the six-frame build -- the default kernel of the bench workload.
// A wave takes a contiguous run of units: it locates its first unit once and from then on only steps to the next clip when
// the run crosses a clip end; the clip record lives in scalar registers.  Against a round-robin deal (which the padded /
// mel-major layouts keep, their stores want adjacent units in adjacent waves -- a run-per-wave build of the mel-major store was
// measured: 0.407 ms against 0.350 ms for the rounds with the sub-group barrier): ragged batches lose the two dependent
// look-ups in front of every unit's PCM loads (-6 %), uniform ones the 64-bit division per unit and a wave re-reads its own
// frame-tail halo (cfg2 -1.6 %, 8192 x 30 s -1.7 %).  The unit body is the same.


template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kSixWaves * 64, 4) void whisper400_six_runs_kernel(const FastParams p) {
#define MS_SIX_RUNS_WAVES kSixWaves
#include "whisper400_six_runs_body.inc"
#undef MS_SIX_RUNS_WAVES
}

// The same on twelve waves per CU (three per SIMD, up to 168 VGPRs) with FIFTEEN mel slots: Whisper large-v3's 128-mel bank (81..134 mels)
// on the six-frame skeleton.  At sixteen waves the fifteen slots' registers pushed the unit loop into scratch (round 3: 9.5 % slower than
// the five-frame kernel); the f64 twin (whisper400_six64_kernel<15, LensSix128>) has run this shape since round 5.
template <int NSLOTS, class Lens>
__global__ __launch_bounds__(kSixWideWaves * 64, 3) void whisper400_six_wide_runs_kernel(const FastParams p) {
#define MS_SIX_RUNS_WAVES kSixWideWaves
#include "whisper400_six_runs_body.inc"
#undef MS_SIX_RUNS_WAVES
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


}  // namespace melspec
