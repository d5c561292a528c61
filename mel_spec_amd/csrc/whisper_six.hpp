// whisper_six.hpp -- the fused n_fft = 400 log-mel pipeline with SIX frames per wavefront, ten lanes per frame
// in every phase (lane = 10*frame + j, 60 of 64 lanes busy throughout).
//
// whisper_wave.hpp gives a frame 10 lanes in phase 1 (one DFT-20 column each) but 11 in phase 2 (residue pairs
// j / 20-j, j = 0..10) and 12 in phases 3-4, which caps a wave at 5 frames and leaves 14 lanes idle in phase 1.
// Two of the eleven phase-2 jobs are half jobs: residue 0 pairs with itself (Z[20q] with Z[200-20q]) and so does
// residue 10 (Z[10+20q] with Z[190-20q]); each needs ONE 10-point DFT and five or six Hermitian pairs.  Here lane 0
// of a frame does both (rows 0 and 10: two DFT-10s and eleven pairs, one more than the ten of lanes 1..9), selecting
// its operands with v_cndmask, and the "modulated copy of row 0" that made the pairing uniform is gone.  Phases 3-4
// use the interval scheme over 10-lane groups (9 intervals + ghost per slot; 9 slots for Whisper's 80 mels).
//
// Same arithmetic as whisper_wave.hpp up to the order of a few f32 additions; same tables except tw2 and the mel
// section.  One 16-wave workgroup per CU: 16 slices of 6*404 floats + one table blob = 162.5 KB of the 160 KiB LDS.
//
// Reference steps: frame_windows src/stft.rs:147-169; FFT src/stft.rs:105-111; sparse mel + log10
// src/mel.rs:148-168; per-frame normalisation src/mel.rs:645-654.
#pragma once
#include "whisper_wave.hpp"

namespace melspec {

constexpr int kSixFrames = 6;     // frames per wavefront
constexpr int kSixLanes = 10;     // lanes per frame in every phase
constexpr int kSixOwn = 9;        // intervals a 10-lane group owns per slot (lane 9 is the ghost)
#ifndef MELSPEC_SIX_WAVES
#define MELSPEC_SIX_WAVES 16
#endif
constexpr int kSixWaves = MELSPEC_SIX_WAVES;     // waves per workgroup (one workgroup per CU)
constexpr int kSixMaxSlots = 9;   // ceil(81 / 9): up to 80 mel bins
// the f64 six-frame kernel (whisper_six64.hpp) also serves the banks of 81..134 mels -- Whisper large-v3's 128 -- with fifteen slots per
// lane: its mel phase runs after the f64 arrays are dead, so the registers are there (the f32 six-frame kernel has no such room at 128
// VGPRs: those banks stay on the five-frame kernels there)
constexpr int kSixWideSlots = 15;
constexpr int kSixWideWaves = 12; // waves per workgroup of the f32 six-frame kernel with fifteen slots (whisper400_six_wide_runs_kernel): three per SIMD

struct SixBlob {                  // float offsets inside the table blob
    // the 40 taps of lane t in the order it uses them, w[20*n1 + 2t + {0, 1}] at [t][2*n1 + {0, 1}]: ten 16-byte reads per unit;
    // 44 = 40 + 4 pad: the ten rows start in ten different groups of four banks
    static constexpr int kWinStride = 44;
    static constexpr int kWin = 0;                        // [10][44] Hann
    static constexpr int kTw1 = 10 * kWinStride;
    static constexpr int kTw1Stride = 44;                 // 20 complex + 4 pad
    static constexpr int kTw2Stride = 28;                 // 11 complex + 6 pad: 28 j mod 64 puts the ten rows in ten different groups of four banks
                                                          // (24 made lanes 0 and 8 collide on every 16-byte read: 10 LDS cycles per unit in the bank model)
    static constexpr int kTw2 = kTw1 + 10 * kTw1Stride;   // [10][24]: lane j >= 1: W_400^{j+20s}, s < 10;
                                                          //   lane 0: W_400^{20s} (s <= 5), W_400^{10+20(s-6)} (s = 6..10)
    static constexpr int kMelStart = kTw2 + 10 * kTw2Stride;              // [kSixMaxSlots*10] ints
    static constexpr int kMelW = (kMelStart + kSixMaxSlots * kSixLanes + 3) & ~3;   // (rise, fall) pairs [slot][r][10][2]
};

struct SixLayout {
    static constexpr int kXRow = 20;
    static constexpr int kXStride = 404;                  // == 20 (mod 32): conflict-free b64 row writes for 10-lane frames
    static constexpr int kPStride = 213;                  // power rows (tools/lds_sim: least conflicts among 201..213)
    static constexpr int kPmaxOff = 1280;                 // after the 6 power rows (6*213 = 1278)
    static constexpr int kPmaxStride = 12;                // 10 maxima + 2 pad
    static constexpr int slice_floats() { return kSixFrames * kXStride; }     // 2424
    // position of exchange row k1 inside a frame's block of 20 rows (hill-climbed on the bank model: the two 16-byte
    // row reads of phase 2 collide 6 cycles per pair instead of 15 in natural order; the row writes stay conflict-free)
    MS_HD static constexpr int row_pos(int k1) {
        constexpr int t[20] = {14, 1, 13, 4, 16, 3, 17, 11, 6, 8, 19, 9, 7, 15, 18, 5, 2, 12, 0, 10};
        return t[k1];
    }
    // per-lane constants of phase 2: float offsets of the two rows lane j reads (rows j and 20-j; lane 0: rows 0 and 10)
    MS_HD static void row_offsets(int j, int &uoff, int &voff) {
        uoff = 0; voff = 0;
        for (int k = 0; k < kSixLanes; ++k)
            if (k == j) { uoff = row_pos(k) * kXRow; voff = row_pos(k == 0 ? 10 : 20 - k) * kXRow; }
    }
};

// compile-time slot lengths of Whisper's 80-mel bank over 10-lane groups
template <int MELS, int... L>
struct LensSixStatic {
    static constexpr bool kStatic = true;
    static constexpr int kSlots = sizeof...(L);
    static constexpr int kMels = MELS;
    MS_HD static constexpr int len(int i) {
        constexpr int t[sizeof...(L)] = {L...};
        return t[i];
    }
    MS_HD static constexpr int woff(int i) {
        constexpr int t[sizeof...(L)] = {L...};
        int s = 0;
        for (int k = 0; k < i; ++k) s += t[k];
        return SixBlob::kMelW + 2 * kSixLanes * s;
    }
};
using LensSix80 = LensSixStatic<80, 1, 1, 1, 2, 2, 3, 4, 6, 7>;
// the same over the fifteen-slot mel section (start bins [15][10], then the weights): Whisper large-v3's 128-mel bank
template <int MELS, int... L>
struct LensSixWideStatic {
    static constexpr bool kStatic = true;
    static constexpr int kSlots = sizeof...(L);
    static constexpr int kMels = MELS;
    static constexpr int kMelW = (SixBlob::kMelStart + kSixWideSlots * kSixLanes + 3) & ~3;
    MS_HD static constexpr int len(int i) {
        constexpr int t[sizeof...(L)] = {L...};
        return t[i];
    }
    MS_HD static constexpr int woff(int i) {
        constexpr int t[sizeof...(L)] = {L...};
        int s = 0;
        for (int k = 0; k < i; ++k) s += t[k];
        return kMelW + 2 * kSixLanes * s;
    }
};
using LensSix128 = LensSixWideStatic<128, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5, 5>;
// Whisper-style banks of 64 and 40 mels at 16 kHz (mel(16000, 400, n_mels), src/mel.rs:547-643): with run-time slot lengths the mel
// phase pays one LDS round trip per bin (64 mels: 0.378 ms at 1024 x 10 s against 0.292 for the 80-mel bank, profiles/r03_sched.txt)
using LensSix64 = LensSixStatic<64, 2, 2, 2, 3, 4, 6, 10, 10>;
using LensSix40 = LensSixStatic<40, 2, 3, 5, 11, 14>;

// ---- phase 1: window, DFT-20 over n1 of column t, twiddle W_200^{t*k1}, 20 exchange rows ----------------------
MS_DEV void six_phase1(int fl, int t, bool active, int hop, const float *blob, const float *gsrc /* unit's first sample */,
                       float *slice) {
    if (!active) return;
    const float *s = gsrc + fl * hop + 2 * t;
    cf x[20];
    const float *w = blob + SixBlob::kWin + t * SixBlob::kWinStride;
    // All twenty loads first, and a scheduling barrier behind them.  Left to itself the machine scheduler sometimes sinks them into the
    // butterflies, two at a time behind `s_waitcnt vmcnt(1)` -- twenty serialised HBM round trips per unit.  Which it does depends on
    // code far from here: round 4 added three words to the kernel's parameter block and the mel-major kernel went from 0.339 to
    // 0.436 ms with an unchanged unit loop source (profiles/r04_sched_flip.txt).
    f2 sv[20];
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) sv[n1] = load2_unaligned(s + 20 * n1);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(MS_SCHED_LOADS_FIRST);
#endif
#pragma unroll
    for (int n1 = 0; n1 < 20; n1 += 2) {
        const f4 wv = ld4(w + 2 * n1);
        x[n1] = {sv[n1].x * wv.x, sv[n1].y * wv.y};
        x[n1 + 1] = {sv[n1 + 1].x * wv.z, sv[n1 + 1].y * wv.w};
    }
    fft20(x);
    const float *tw = blob + SixBlob::kTw1 + t * SixBlob::kTw1Stride;
    float *xo = slice + fl * SixLayout::kXStride + 2 * t;
    *reinterpret_cast<f2 *>(xo + SixLayout::row_pos(0) * SixLayout::kXRow) = f2{x[0].re, x[0].im};
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) {
        const f2 wv = *reinterpret_cast<const f2 *>(tw + 2 * k1);
        const cf y = cmul(x[k1], cf{wv.x, wv.y});
        *reinterpret_cast<f2 *>(xo + SixLayout::row_pos(k1) * SixLayout::kXRow) = f2{y.re, y.im};
    }
}

// ---- phase 2: two DFT-10s, Hermitian pairs with W_400, 4*|X|^2 to the power row ------------------------------
// Lanes 1..9: u = DFT(row j), v = DFT(row 20-j); pair s: Z[k] = u[s], Z[200-k] = v[9-s], k = j + 20s.
// Lane 0:     u = DFT(row 0), v = DFT(row 10);   pairs 0..5: u[s] with u[(10-s)%10] (k = 20s);
//             pairs 6..10: v[s-6] with v[15-s] (k = 10 + 20(s-6)).  koff = j (lanes 1..9) or -110 (lane 0).
MS_DEV void six_phase2(int fl, int j, bool active, const float *blob, float *slice, int uoff, int voff) {
    if (!active) return;
    const float *ua = slice + fl * SixLayout::kXStride + uoff;
    const float *va = slice + fl * SixLayout::kXStride + voff;
    cf u[10], v[10];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const f4 a = *reinterpret_cast<const f4 *>(ua + 4 * i);
        const f4 b = *reinterpret_cast<const f4 *>(va + 4 * i);
        u[2 * i] = {a.x, a.y};
        u[2 * i + 1] = {a.z, a.w};
        v[2 * i] = {b.x, b.y};
        v[2 * i + 1] = {b.z, b.w};
    }
    fft10(u);
    fft10(v);
    const bool lane0 = j == 0;
    const int koff = lane0 ? -110 : j;
    const float *tw = blob + SixBlob::kTw2 + j * SixBlob::kTw2Stride;
    float *p = slice + fl * SixLayout::kPStride;
    auto pair = [&](cf zk, cf zm, cf W, float &pk, float &pm) {
        const cf S = {zk.re + zm.re, zk.im - zm.im};
        const cf D = {zk.re - zm.re, zk.im + zm.im};
        const cf wd = cmul(W, D);
        const float ar = S.re + wd.im, ai = S.im - wd.re;
        const float br = S.re - wd.im, bi = S.im + wd.re;
        pk = ar * ar + ai * ai;               // 4*|X[k]|^2 (the mel weights carry the 1/4)
        pm = br * br + bi * bi;               // 4*|X[200-k]|^2
    };
#pragma unroll
    for (int s = 0; s < 10; s += 2) {
        const f4 w2 = *reinterpret_cast<const f4 *>(tw + 2 * s);
        float pk[2], pm[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ss = s + h;
            cf zk, zm;
            if (ss < 6) {
                const cf own = u[(10 - ss) % 10];
                zk = u[ss];
                zm = {lane0 ? own.re : v[9 - ss].re, lane0 ? own.im : v[9 - ss].im};
            } else {
                zk = {lane0 ? v[ss - 6].re : u[ss].re, lane0 ? v[ss - 6].im : u[ss].im};
                zm = {lane0 ? v[15 - ss].re : v[9 - ss].re, lane0 ? v[15 - ss].im : v[9 - ss].im};
            }
            const cf W = h == 0 ? cf{w2.x, w2.y} : cf{w2.z, w2.w};
            pair(zk, zm, W, pk[h], pm[h]);
        }
        // the two stores of a side next to each other: same base, offsets 20 words apart -> one ds_write2_b32 (6 LDS cycles for the
        // two words; four single stores in k, 200-k, k, 200-k order were 4 x 4)
        const int k = (s < 6 ? j : koff) + 20 * s;
        p[k] = pk[0];
        p[k + 20] = pk[1];
        p[200 - k] = pm[0];
        p[180 - k] = pm[1];
    }
    if (lane0) {                              // eleventh pair: Z[90] with Z[110]
        const f2 w2 = *reinterpret_cast<const f2 *>(tw + 20);
        float pk, pm;
        pair(v[4], v[5], cf{w2.x, w2.y}, pk, pm);
        p[90] = pk;
        p[110] = pm;
    }
}

// ---- phase 3: interval sums (each bin read once, feeding a rising and a falling filter), 10-lane groups ----------
template <int NSLOTS, class Lens>
MS_DEV void six_phase3_sums(int fl, int j, bool active, const MelSlots &ms, const float *blob, const float *slice,
                            const int (&st)[NSLOTS], float (&rise)[NSLOTS], float (&fprev)[NSLOTS]) {
    // Every lane computes (a lane without a frame reads frame 0's row: in bounds, and nobody uses its sums): zeroing the 2 x NSLOTS
    // results for the sake of the idle lanes, and starting every sum from a zero register, were 27 v_mov per unit (ISA histogram).
    const float *p = slice + (active ? fl : 0) * SixLayout::kPStride;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        float ar = 0.0f, af = 0.0f;
        if (Lens::kStatic) {
            if (i < Lens::kSlots) {
                const float *pp = p + st[i];
                const float *w = blob + Lens::woff(i < Lens::kSlots ? i : 0) + 2 * j;
#pragma unroll
                for (int r = 0; r < Lens::len(i < Lens::kSlots ? i : 0); ++r) {
                    const f2 wv = ld2_single(w + 2 * kSixLanes * r);
                    const float pv = pp[r];
                    if (r == 0) { ar = wv.x * pv; af = wv.y * pv; }
                    else { ar += wv.x * pv; af += wv.y * pv; }
                    // long slots (banks of fewer mels: 10..14 bins per lane) in pieces of five: with every read of a slot issued up
                    // front the 64-mel bank spilled 121 VGPRs and ran slower than the run-time loop (0.3885 against 0.3741 ms)
#if defined(__HIP_DEVICE_COMPILE__)
                    if (Lens::len(i < Lens::kSlots ? i : 0) > 7 && r % 5 == 4 && r + 1 < Lens::len(i < Lens::kSlots ? i : 0))
                        __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
        } else if (i < ms.n_slots) {
            const float *pp = p + st[i];
            const float *w = blob + ms.woff[i] + 2 * j;
            interval_bins_runtime<2 * kSixLanes>(pp, w, ms.len[i], ar, af);
        }
        rise[i] = ar;
        fprev[i] = af;
    }
}

// mel[m] = rise of interval m (this lane) + fall of interval m+1 (next lane); log10, lane maximum to LDS
// The log-mel values are carried with a bias of +16: log10(max(e, 1e-10)) + 16 lies in [6, ~30], a POSITIVE float, and positive
// floats order like their bit patterns.  Every maximum / minimum of phases 3-4 is then an integer one (v_max_i32 / v_max3_i32 /
// v_min3_i32): a float maximum of a value that comes from memory or from another basic block costs a canonicalising v_max(x, x)
// first (17 per unit in the ISA of the float form), the floor is one v_max against 6.0 instead of compare + select, and the bias
// folds into the two FMAs that were multiplies.  (x + 4) / 4 = biased * 0.25 - 3.
MS_DEV int six_bits(float v) { return __builtin_bit_cast(int, v); }
MS_DEV float six_float(int v) { return __builtin_bit_cast(float, v); }
MS_DEV int six_imax(int a, int b) { return a > b ? a : b; }
MS_DEV int six_imin(int a, int b) { return a < b ? a : b; }
template <int NSLOTS>
MS_DEV void six_phase3_finish(int fl, int j, bool active, int n_mels, const float (&rise)[NSLOTS],
                              const float (&fnext)[NSLOTS] /* fprev of lane+1 */, float *slice, float (&vals)[NSLOTS]) {
    if (!active) return;
    int mx = 0;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const float e = rise[i] + fnext[i];
        // below the floor the logarithm is <= -10 (log2(0) = -inf; of a negative sum, which only a caller's bank can produce, NaN)
        // and the maximum with 6 = -10 + 16 returns exactly 6
        const float v = __builtin_fmaxf(fast_log2(e) * 0.30102999566398120f + 16.0f, 6.0f);
        vals[i] = v;
        if (j < kSixOwn && j + kSixOwn * i < n_mels) mx = six_imax(mx, six_bits(v));
    }
    reinterpret_cast<int *>(slice)[SixLayout::kPmaxOff + fl * SixLayout::kPmaxStride + j] = mx;
}

// ---- phase 4: frame maximum, clamp at max - 8, (x + 4) / 4, store ---------------------------------------------------
// store: this lane's frame column exists in the output; valid: it is a real frame (otherwise a zero column of a padded
// layout).  row_w == 0: [frame][mel] rows; row_w > 0: [mel][row_w] rows (interleave_frames, src/mel.rs:480-544).
// GUARD: returns true on lanes that hold a band within kGuardBand decades of the clamp (see wave_phase4).
struct alignas(16) SixI4 { int x, y, z, w; };
struct alignas(8) SixI2 { int x, y; };
// KEYS (mel-major stores feeding the TGA quantiser, tga_quant.hpp): *kmin / *kmax = the smallest / largest biased value this lane
// stored, a zero column counting as 12 (12 * 0.25 - 3 == 0 exactly); lanes that store nothing leave them untouched.
MS_DEV float six_out(int c) { return six_float(c) * 0.25f - 3.0f; }          // (x + 4) / 4 of the biased value
template <int NSLOTS, bool LAYOUT = false, bool GUARD = false, bool KEYS = false>
MS_DEV bool six_phase4(int fl, int j, bool store, bool valid, int n_mels, const float *slice, const float (&vals)[NSLOTS],
                       float *out_tile, long long row_w, int *kmin = nullptr, int *kmax = nullptr) {
    if (!LAYOUT) { valid = true; row_w = 0; }
    if (!store || j >= kSixOwn) return false;
    int lo = 0;          // bits of (frame maximum - 8), biased
    if (valid) {
        const int *pm = reinterpret_cast<const int *>(slice) + SixLayout::kPmaxOff + fl * SixLayout::kPmaxStride;
        const SixI4 a = *reinterpret_cast<const SixI4 *>(pm), b = *reinterpret_cast<const SixI4 *>(pm + 4);
        const SixI2 c = *reinterpret_cast<const SixI2 *>(pm + 8);
        const int m0 = six_imax(six_imax(a.x, a.y), six_imax(a.z, a.w));
        const int m1 = six_imax(six_imax(b.x, b.y), six_imax(b.z, b.w));
        lo = six_bits(six_float(six_imax(six_imax(m0, m1), six_imax(c.x, c.y))) - 8.0f);      // >= -2: negative only when every band sits on the floor
    }
    float *o = row_w ? out_tile + static_cast<long long>(j) * row_w + fl : out_tile + static_cast<long long>(fl) * n_mels + j;
    const long long step = row_w ? kSixOwn * row_w : kSixOwn;
    int cmin = 0x7f000000, cmax = 0;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const int m = j + kSixOwn * i;
        if (m < n_mels) {
            // vals >= 6 > 0: the integer order is the float order, also against a negative lo; a zero column is the value 12
            const int c = valid ? six_imax(six_bits(vals[i]), lo) : 0x41400000;
            o[i * step] = six_out(c);
            if (GUARD || KEYS) cmin = six_imin(cmin, c);
            if (KEYS) cmax = six_imax(cmax, c);
        }
    }
    if (KEYS) { *kmin = cmin; *kmax = cmax; }
    return GUARD && valid && six_float(cmin) < six_float(lo) + kGuardBand;
}

}  // namespace melspec
