// fbank512_kernels.hpp -- the fused 512-point kernels (phases in fbank_wave.hpp): Kaldi fbank (wave and workgroup-per-clip forms, CMN),
// the NeMo / Parakeet frontend and its normaliser, Whisper at n_fft = 512; f64 and f32 instantiations.
#pragma once
#include "kernels_common.hpp"
#include "fbank_wave.hpp"

namespace melspec {

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
            if (FLAVOR == kFlavorWhisper) fb_phase2_split<T, true, sizeof(T) == 8>(fl, j, act, tblob, own, part, slice);      // f32: the amplitude form (fbank_tables.hpp)
            // NeMo: power spectra always (src/mel.rs:356-357) -- the magnitude form stays out of its unit loop (742 -> ~400 instructions in phase 2)
            else if (FLAVOR == kFlavorNemo || use_power) fb_phase2_split<T, true>(fl, j, act, tblob, own, part, slice);
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

// ------------------------------------------------------------------------------------
// MELSPEC_PRECISION_AUTO at n_fft = 512 (round 6): Whisper's log-mel on the 512-point kernels with the n_fft = 400 family's contract --
// the f32 kernel where its arithmetic can vouch for 1e-4, f64 where it cannot, decided by the batch itself.  Plain [frame][mel] batches
// (uniform and ragged) of the compile-time banks; two launches of this kernel per call:
//   T = float  (twelve waves): the unit loop of fbank512_wave_kernel<float, 12, .., Whisper, RUNS> with the guard of w512_phase4 on.  The
//              first unit of every wave of the resident workgroups is the sample of the vote (FixSink::vote, kernels_common.hpp); every
//              unit leaves its frame mask in the note list (one 32-bit word per unit of the batch: no atomics).  On "heavy" the waves stop.
//   T = double (eight waves), queued behind it and gated on ITS verdict: "heavy" -> every unit of the batch; "light" -> the units whose
//              word in the list is not zero (a wave reads 64 words of its run at a time and walks the set bits), all four frames of each.
// So a batch costs the f32 rate on noise-like input, the f64 rate on input that trips the guard on more than 1/8 of the sampled frames,
// and the same batch gives the same bits whatever the context computed before it.  Reference arithmetic: src/stft.rs:99-111, src/mel.rs:148-168.
// ------------------------------------------------------------------------------------
struct W512AutoParams {
    FbankFastParams f;
    FixSink fix;              // f32 launch: vote + statistics, list = the note words; gated launch: statistics only (+ list)
    const unsigned *gate;     // gated launch: the voting launch's verdict word
    unsigned gate_seq;        // ... and its number: anything else in the word = not this launch's verdict = nothing to do
};

template <class T, int WAVES, int NSLOTS, class Lens>
__global__ __launch_bounds__(WAVES * 64, 1) void w512_auto_kernel(const W512AutoParams q) {
    using L = FbankLayout<T>;
    constexpr bool F32 = sizeof(T) == 4;
    const FbankFastParams &p = q.f;
    unsigned verdict = 0;
    if (!F32) {          // the gated launch: this batch's verdict, or nothing
        const unsigned g = *q.gate;
        if ((g >> 2) != (q.gate_seq & 0xffffffu) || !(g & kVoteDecided)) return;
        verdict = g & 3u;
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int tid = threadIdx.x;
    for (int i = tid; i < p.blob_words; i += WAVES * 64) ldsw[i] = p.d_blob[i];
    unsigned *wg_done = ldsw + p.blob_words + WAVES * L::slice_elems() * (sizeof(T) / 4);     // guard_wave_done's two words, vote_cast's three, vote_check's one
    if (tid < 6) wg_done[tid] = 0;
    __syncthreads();
    const T *tblob = reinterpret_cast<const T *>(ldsw);
    const float *mel = reinterpret_cast<const float *>(ldsw + p.mel_off_words);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    T *slice = reinterpret_cast<T *>(ldsw + p.blob_words) + wave * L::slice_elems();
    const int fl = lane / kFbLanes, j = lane - fl * kFbLanes;
    const bool in = lane < kFbFPW * kFbLanes;
    int st[NSLOTS];
    const int *starts = reinterpret_cast<const int *>(mel + FbankBlob::kMelStart);
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) st[i] = in ? starts[i * kFbLanes + j] : 0;
    unsigned *list = reinterpret_cast<unsigned *>(q.fix.list);

    ClipRun cr;
    if (!cr.init(p.b, (uint64_t)xcd_logical_block() * WAVES + wave, (uint64_t)gridDim.x * WAVES)) {
        if (F32 && blockIdx.x < q.fix.vote_groups) vote_cast(q.fix, wg_done + 2, WAVES, lane, 0, 0);
        if (F32 || (verdict & kVoteHeavy)) guard_wave_done(q.fix, wg_done, WAVES, lane, 0);          // (who reports: see the end of the kernel)
        return;
    }
    int nv = 0;
    // one work unit (the Whisper flavour of fbank512_wave_kernel's body, plain store); returns the frames whose guard tripped as a mask
    auto unit = [&]() __attribute__((always_inline)) -> unsigned {
        cr.enter(p.b);
        const UnitLoc loc = cr.loc();
        const uint64_t f0 = loc.unit * kFbFPW;
        const uint64_t left = f0 < loc.frames ? loc.frames - f0 : 0;
        nv = left < (uint64_t)kFbFPW ? (int)left : kFbFPW;
        const bool act = in && fl < nv;
        MS_PRIO(0);
        w512_phase1<T, true>(fl, j, act, loc.pcm + (f0 + (uint64_t)(act ? fl : 0)) * (uint64_t)p.shift, tblob, slice);
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(1);
        {
            cpx<T> own[16], part[8];
            fb_phase2_dft<T, false>(fl, j, act, slice, own);
#pragma unroll
            for (int i = 0; i < 8; ++i) part[i] = {partner16(own[8 + i].re), partner16(own[8 + i].im)};
            fb_phase2_split<T, true, sizeof(T) == 8>(fl, j, act, tblob, own, part, slice);
        }
        __builtin_amdgcn_wave_barrier();
        MS_PRIO(2);
        float rise[NSLOTS], fprev[NSLOTS], fnext[NSLOTS], vals[NSLOTS];
        fb_phase3_sums<T, NSLOTS, Lens>(fl, j, act, p.slots, mel, slice, st, rise, fprev);
#pragma unroll
        for (int i = 0; i < NSLOTS; ++i) fnext[i] = wave_shift_down1(fprev[i]);
        float *slice_f = reinterpret_cast<float *>(slice);
        w512_phase3_log<NSLOTS>(fl, j, act, p.n_mels, rise, fnext, slice_f, vals);
        __builtin_amdgcn_wave_barrier();
        const bool flag = w512_phase4<NSLOTS, true>(fl, j, act, act, p.n_mels, slice_f, vals, loc.out + f0 * (uint64_t)p.n_mels, 0);
        __builtin_amdgcn_wave_barrier();
        return frame_mask<kFbLanes, kFbFPW>(__builtin_amdgcn_ballot_w64(flag));
    };
    unsigned flagged = 0;
    if (F32) {
        // the vote: the first units run in a loop of their own until the verdict is known (whisper400_six_runs_kernel says why)
        bool sample = blockIdx.x < q.fix.vote_groups;
        unsigned polled = 0;
        for (; cr.unit < cr.end && verdict == 0; ++cr.unit) {
            const unsigned mask = unit();
            if (lane == 0) list[cr.unit] = mask;
            flagged += static_cast<unsigned>(__builtin_popcount(mask));
            if (sample) {
                vote_cast(q.fix, wg_done + 2, WAVES, lane, static_cast<unsigned>(__builtin_popcount(mask)), static_cast<unsigned>(nv));
                sample = false;
            }
            verdict = vote_check(q.fix, wg_done + 2, ++polled, wave);
        }
        if (verdict == 0) verdict = vote_poll(q.fix);                  // a run shorter than the vote
        if (verdict & kVoteHeavy) {                                    // the gated launch computes the whole batch and reports it
            guard_wave_done(q.fix, wg_done, WAVES, lane, 0);
            return;
        }
        for (; cr.unit < cr.end; ++cr.unit) {
            const unsigned mask = unit();
            if (lane == 0) list[cr.unit] = mask;
            flagged += static_cast<unsigned>(__builtin_popcount(mask));
        }
    } else if (verdict & kVoteHeavy) {
        for (; cr.unit < cr.end; ++cr.unit) flagged += static_cast<unsigned>(__builtin_popcount(unit()));
    } else {
        // light: the units the f32 launch could not vouch for, 64 note words at a time
        flagged = 0;
        for (uint64_t base = cr.unit; base < cr.end; base += 64) {
            const uint64_t u = base + lane;
            const unsigned word = u < cr.end ? list[u] : 0u;
            uint64_t todo = __builtin_amdgcn_ballot_w64(word != 0);
            while (todo) {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                cr.seek(p.b, base + k);
                unit();
            }
        }
    }
    // light batches: the voting launch reports (its flagged frames = the frames recomputed); heavy ones: the gated launch (the frames that
    // would have tripped the guard); a light gated launch reports nothing (FixSink::acc is shared: one report per batch)
    if (F32 || (verdict & kVoteHeavy)) guard_wave_done(q.fix, wg_done, WAVES, lane, flagged);
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
    float *d_means;         // melspec_fbank_compute_uniform_device_split: the clip's column means go here ([clip][n_mels]) and the rows stay
                            // un-normalised -- the second pass over the rows (a third of this kernel's traffic) is left to a consumer that
                            // can fold the subtraction into its own first read; nullptr: the CMN inside (Fbank::compute, src/fbank.rs:224-233)
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
    const bool use_log = p.use_log != 0;
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
                // power spectra only (FbankConfig::use_power, the default): magnitudes run on the two-kernel path -- with both forms of the
                // split behind a run-time branch the unit loop carried 123 f64 + 170 other instructions it never executed
                fb_phase2_split<T, true>(fl, j, act, tblob, own, part, slice);
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
        if (gen > 0 && !(q.lab_skip & 1) && q.d_means == nullptr) {
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
                const float mean_m = f32_div_rn(s, fr);
                sh->mean[par][m] = mean_m;
                if (q.d_means) q.d_means[(uint64_t)clip * nm + m] = mean_m;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(&sh->ready[par], turn, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        sub.begin(out, frames, nm, wave, lane);        // this clip is the next one to subtract
        ++gen;
    }
    if (gen > 0 && !(q.lab_skip & 1) && q.d_means == nullptr) {
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
    float *d_means; // not nullptr: write the clip's column means there ([clip][n_mels]) and leave the rows as they are (the split output)
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
            if (p.d_means) {
                if (tid < nm) p.d_means[(uint64_t)clip * nm + tid] = mean_s[tid];
            } else if (g < G) {
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
            if (p.d_means) {
                if (tid < cols) p.d_means[(uint64_t)clip * nm + m0 + tid] = rows[tid];
            } else if (g < G) {
                const float mean = rows[tid - g * cols];
                for (uint64_t f = g; f < frames; f += G) o[f * nm + m] -= mean;
            }
            __syncthreads();
        }
    }
}


}  // namespace melspec
