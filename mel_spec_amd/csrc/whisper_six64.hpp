// whisper_six64.hpp -- the f64 build of the n_fft = 400 log-mel pipeline on the SIX-frame skeleton of whisper_six.hpp: six frames per
// wavefront, ten lanes per frame in every phase, window / 400-point real FFT / Hermitian split / |X|^2 in f64 (the reference's
// arithmetic, src/stft.rs:98-111), the f32 interval mel / log10 / clamp phases of whisper_six.hpp behind the power rows.
//
// Why it exists (round 5).  whisper_wave_f64.hpp runs five frames per wave on eleven lanes per frame (50 of 64 lanes busy in phase 1,
// two half jobs in phase 2) with a whole f64 exchange buffer per frame: 18.5 KB of LDS per wave, 8 waves per CU = two per SIMD, and an
// f64 instruction costs the same 4-5 cycles whether 50 or 60 of its lanes work.  Here
//   * a frame has ten lanes throughout (lane 0 does residues 0 and 10, eleven Hermitian pairs, as in whisper_six.hpp): 60 of 64 lanes busy,
//     six frames for about the instruction count five used to cost;
//   * THE EXCHANGE GOES THROUGH LDS IN TWO HALVES.  Lane j of phase 2 reads rows j and 20 - j (lane 0: rows 0 and 10) -- one row out of
//     {0..9} and one out of {10..19}.  The ten lanes of a frame write rows 0..9, every lane reads its first row, the lanes write rows 10..19
//     over the same ten row slots, every lane reads its second row.  The LDS operations of a wave execute in program order, so no barrier is
//     needed; a frame needs 1.8 KB instead of 3.7 KB, a wave 10.9 KB, and twelve waves fit a CU: THREE waves per SIMD (<= 168 VGPRs).
// The result is the same f64 arithmetic as whisper_wave_f64.hpp's (fft20, fft10, the 12-operation power split) in another order of
// lanes; tests/emu runs this source on the host.
//
// Reference steps: frame_windows src/stft.rs:147-169; FFT src/stft.rs:105-111; sparse mel + log10 src/mel.rs:148-168; per-frame
// normalisation src/mel.rs:645-654.
#pragma once
#include "whisper_six.hpp"
#include "whisper_wave_f64.hpp"

// (Scheduling fences between the pieces of the table reads -- five taps / twiddles at a time -- were built while the first form spilled; with
// the phases in one divergent region (six64_phases12) the kernel holds 168 VGPRs without them and is 1 % faster: 0.4349 against 0.4394 ms.)
#if defined(__HIP_DEVICE_COMPILE__) && defined(MELSPEC_SIX64_FENCE)
#define MS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MS_SCHED_FENCE() ((void)0)
#endif

namespace melspec {

#ifndef MELSPEC_SIX64_WAVES
#define MELSPEC_SIX64_WAVES 12
#endif
constexpr int kSix64Waves = MELSPEC_SIX64_WAVES;      // waves per workgroup, one workgroup per CU: three per SIMD

struct Six64Blob {                 // f64 table part, offsets in doubles; the f32 mel section of the six-frame blob follows it
    // every row is read with 16-byte accesses by the ten lanes of a frame (the six frames read the same addresses: broadcast); rows
    // 21 / 11 sixteen-byte slots apart put the ten lanes' reads in ten different slots of the 16 that ds_read_b128 serves per cycle
    static constexpr int kWinStride = 42;                     // [10][42]: the 40 taps of lane t in the order it uses them, w[20 n1 + 2t + {0, 1}]
    static constexpr int kWin = 0;
    static constexpr int kTw1Stride = 42;                     // [10][42]: W_200^{t k1}, k1 < 20
    static constexpr int kTw1 = kWin + 10 * kWinStride;
    static constexpr int kTw2Stride = 22;                     // [10][22]: (2 sin, 4 cos) of W_400^k's angle for the lane's eleven pairs (six_phase2's k's)
    static constexpr int kTw2 = kTw1 + 10 * kTw1Stride;
    static constexpr int kCount = kTw2 + 10 * kTw2Stride;     // 1060 doubles
};

struct Six64Layout {
    static constexpr int kXRow = 22;                          // doubles per exchange row: 10 complex + 1 pad slot
    // doubles per frame: ten rows = 220, padded to == 4 (mod 16): the eight lanes ds_write_b128 serves together then always write 32
    // different banks although they straddle two frames (ten lanes x 16 bytes = 40 banks per frame and row)
    static constexpr int kXStride = 228;
    static constexpr int slice_doubles() { return kSixFrames * kXStride; }       // 1368 doubles = 10 944 bytes per wave
    // row slot (0..9) of the row lane j reads -- in BOTH halves: rows j (first half) and 20 - j / 10 (second half) sit in the same slot.
    // Searched on the ds_read_b128 bank model (sixteen-lane groups {0-3, 12-15, 20-27} ... over 64 banks): 30 conflict cycles per ten
    // reads against 50-60 for the natural order.
    MS_HD static constexpr int slot_of_lane(int j) {
        constexpr int t[10] = {7, 9, 5, 3, 0, 8, 6, 2, 1, 4};
        return t[j];
    }
    MS_HD static constexpr int slot_of_row(int k1) { return slot_of_lane(k1 < 10 ? k1 : (k1 == 10 ? 0 : 20 - k1)); }
    // per-lane constant: double offset of the lane's row slot inside its frame
    MS_HD static int row_offset(int j) {
        int r = 0;
        for (int k = 0; k < kSixLanes; ++k)
            if (k == j) r = slot_of_lane(k) * kXRow;
        return r;
    }
};
static_assert(Six64Layout::slice_doubles() * 2 >= SixLayout::kPmaxOff + kSixFrames * SixLayout::kPmaxStride, "the f32 power rows and maxima alias the head of the slice");

// ---- phase 1: window (f64), DFT-20 over n1 of column t, twiddle W_200^{t k1}: the twenty exchange values of this lane ---------------------
MS_DEV void six64_phase1(int fl, int t, bool active, int hop, const double *MS_RESTRICT tb, const float *gsrc /* unit's first sample */, cd (&x)[20]) {
    if (!active) return;
    const float *s = gsrc + fl * hop + 2 * t;
    f2 sv[20];
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) sv[n1] = load2_unaligned(s + 20 * n1);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(MS_SCHED_LOADS_FIRST);        // six_phase1 (whisper_six.hpp) says why
#endif
    const double *w = tb + Six64Blob::kWin + t * Six64Blob::kWinStride;
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) {
        const cd wv = ldc(w + 2 * n1);
        x[n1] = {static_cast<double>(sv[n1].x) * wv.re, static_cast<double>(sv[n1].y) * wv.im};        // src/stft.rs:163
        if (n1 % 5 == 4) MS_SCHED_FENCE();
    }
    fft20(x);
    MS_SCHED_FENCE();
    const double *tw = tb + Six64Blob::kTw1 + t * Six64Blob::kTw1Stride;
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) {
        x[k1] = cmul(x[k1], ldc(tw + 2 * k1));
        if (k1 % 5 == 4) MS_SCHED_FENCE();
    }
}

// one half of the exchange: rows 10 half .. 10 half + 9 of this lane's column into the frame's ten row slots
MS_DEV void six64_store_half(int fl, int t, bool active, int half, const cd (&x)[20], double *MS_RESTRICT rows) {
    if (!active) return;
    double *xo = rows + fl * Six64Layout::kXStride + 2 * t;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const int k1 = 10 * half + r;
        stc(xo + Six64Layout::slot_of_row(k1) * Six64Layout::kXRow, x[half ? 10 + r : r]);
    }
}

// the lane's row of the half that is in LDS: ten complex values (rofs = Six64Layout::row_offset(j))
MS_DEV void six64_read_row(int fl, bool active, int rofs, const double *MS_RESTRICT rows, cd (&u)[10]) {
    if (!active) return;
    const double *ua = rows + fl * Six64Layout::kXStride + rofs;
#pragma unroll
    for (int i = 0; i < 10; ++i) u[i] = ldc(ua + 2 * i);
}

// ---- phase 2: two DFT-10s, eleven / ten Hermitian pairs straight to the two powers (precise_phase2's 12-operation form), f32 power row ---
// Pairing as in six_phase2: lanes 1..9: Z[k] = u[s], Z[200 - k] = v[9 - s], k = j + 20 s; lane 0: pairs 0..5 u[s] with u[(10 - s) % 10]
// (k = 20 s), pairs 6..10 v[s - 6] with v[15 - s] (k = 10 + 20 (s - 6)).
MS_DEV void six64_phase2(int fl, int j, bool active, const double *MS_RESTRICT tb, cd (&u)[10], cd (&v)[10], float *slice) {
    if (!active) return;
    fft10(u);
    MS_SCHED_FENCE();
    fft10(v);
    MS_SCHED_FENCE();
    const bool lane0 = j == 0;
    const int koff = lane0 ? -110 : j;
    const double *tw = tb + Six64Blob::kTw2 + j * Six64Blob::kTw2Stride;
    float *p = slice + fl * SixLayout::kPStride;
    auto pair = [&](cd zk, cd zm, cd t, float &pk, float &pm) {
        const double a = zk.re * zk.re + zk.im * zk.im, b = zm.re * zm.re + zm.im * zm.im;
        const double c = zk.re * zm.im + zm.re * zk.im;
        const double s2 = a + b, xx = (a - b) * t.re + c * t.im;
        pk = static_cast<float>(2.0 * s2 + xx);              // 4 |X[k]|^2
        pm = static_cast<float>(2.0 * s2 - xx);              // 4 |X[200 - k]|^2
    };
    auto sel = [&](cd a, cd b) { return cd{lane0 ? a.re : b.re, lane0 ? a.im : b.im}; };
#pragma unroll
    for (int s = 0; s < 10; s += 2) {
        float pk[2], pm[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ss = s + h;
            cd zk, zm;
            if (ss < 6) {
                zk = u[ss];
                zm = sel(u[(10 - ss) % 10], v[9 - ss]);
            } else {
                zk = sel(v[ss - 6], u[ss]);
                zm = sel(v[15 - ss], v[9 - ss]);
            }
            pair(zk, zm, ldc(tw + 2 * ss), pk[h], pm[h]);
        }
        MS_SCHED_FENCE();
        const int k = (s < 6 ? j : koff) + 20 * s;
        p[k] = pk[0];
        p[k + 20] = pk[1];
        p[200 - k] = pm[0];
        p[180 - k] = pm[1];
    }
    if (lane0) {                                              // eleventh pair: Z[90] with Z[110]
        float pk, pm;
        pair(v[4], v[5], ldc(tw + 20), pk, pm);
        p[90] = pk;
        p[110] = pm;
    }
}

#if defined(__HIPCC__)
// Phases 1-2 of one unit as the kernel runs them: ONE divergent region.  The value arrays must not be visible outside it: declared in the
// kernel's loop body and handed to the step functions one `if (active)` at a time they are phi nodes of the unit loop -- for the lanes
// without a frame "the value of the previous iteration" -- and 160 VGPRs stay live around the whole loop (the first build spilled 290).
// LDS operations of a wave execute in program order, divergent or not; the wave barriers only pin the compiler's order.
MS_DEV void six64_phases12(int fl, int j, bool active, int rofs, int hop, const double *MS_RESTRICT tb, const float *gsrc, double *rows, float *slice) {
    if (!active) return;
    cd u[10], v[10];
    {
        cd x[20];
        six64_phase1(fl, j, true, hop, tb, gsrc, x);
        six64_store_half(fl, j, true, 0, x, rows);
        __builtin_amdgcn_wave_barrier();
#ifndef MELSPEC_NO_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        six64_read_row(fl, true, rofs, rows, u);
        __builtin_amdgcn_wave_barrier();
        six64_store_half(fl, j, true, 1, x, rows);
    }
    __builtin_amdgcn_wave_barrier();
    six64_read_row(fl, true, rofs, rows, v);
    __builtin_amdgcn_wave_barrier();
    six64_phase2(fl, j, true, tb, u, v, slice);
}
#endif

}  // namespace melspec
