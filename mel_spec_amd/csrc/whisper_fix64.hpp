// whisper_fix64.hpp -- the f64 recompute of ONE frame by a whole wavefront, inside the f32 kernels (MELSPEC_PRECISION_AUTO).
//
// The f32 kernels check every frame against an error bound in phase 4 (wave_phase4: a mel band within kGuardBand decades of the
// per-frame clamp).  A wave notes the units that have such frames and, when its run of units is done, recomputes those frames
// itself: window, 400-point real FFT, Hermitian split and |X|^2 in f64 (the reference's arithmetic, src/stft.rs:98-111), then the
// kernel's own f32 mel / log10 / clamp phases on the new power rows (once per unit).  No second launch (two dependent launches cost the
// 0.3 ms bench step 9.5 us), no cross-wave queue: noise-like input notes nothing (the bench workload pays one ballot per unit and runs
// at the f32 rate), a recomputed frame costs its wave ~4.5 us.  What that is made of was taken apart in round 2 (tools/guard_bench.py,
// 1024 x 10 s, 80 mels, speech = jfk tiled, 66 % of the frames recomputed): phases 3-4 per frame 1.57 ms -> once per unit 1.35 ms;
// the twiddles of steps 2 and 4 (lane constants) loaded once per wave instead of per frame 1.31 ms; the next frame's samples in
// flight during steps 2-4 1.21-1.24 ms; the window taps in registers as well: 47 spilled VGPRs and 1.58 ms, so they stay a table read
// per frame.  A tone over a -70 dB floor (100 % recomputed): 1.87 -> 1.45 ms.  MELSPEC_PRECISION_F64, the dedicated f64 kernel
// (whisper_wave_f64.hpp) on everything, takes 0.50 ms: that is the mode for such input.
//
// Register budget is what shapes it: the f32 kernels live at <= 128 VGPRs (four waves per SIMD), so no lane may hold a 20-point
// f64 DFT (80 VGPRs of data).  The complex-200 transform is spread over the lanes instead, 200 = 8 x 25 by Good-Thomas (no
// twiddles between the two factors) and 25 = 5 x 5 by Cooley-Tukey, four steps through 3.2 KB of the wave's LDS slice:
//   n = (25 n1 + 8 n2) mod 200,  k = (25 k1 + 176 k2) mod 200,  n2 = 5 b + a,  k2 = c + 5 d
//   step 1  lane n2 < 25:        DFT-8 over n1 of z[n] = x[2n] w[2n] + i x[2n+1] w[2n+1]          -> A[k1][n2]
//   step 2  lane (k1, a) < 40:   DFT-5 over b of A[k1][5b + a], times W_25^{a c}                 -> B[k1][c][a]
//   step 3  lane (k1, c) < 40:   DFT-5 over a                                                    -> Z[k]
//   step 4  lane k, k + 64 <= 100: X[k] = (S - i W_400^k D) / 2 from Z[k], Z[200 - k];  4 |X|^2 -> the frame's f32 power row
// The f64 window, W_25 and W_400 tables come from global memory (FixTables, 4.4 KB, L1/L2 resident): LDS is full.
#pragma once
#include <cmath>
#include <vector>

#include "device_fft.hpp"
#include "tables.hpp"

namespace melspec {

struct FixTables {
    static constexpr int kWin = 0;                  // [400] periodic Hann, f64
    static constexpr int kW25 = 400;                // [5 a][5 c] complex W_25^{a c}
    static constexpr int kW400 = kW25 + 50;         // [101] complex W_400^k
    static constexpr int kCount = kW400 + 202;      // 652 doubles
    static constexpr int kScratchDoubles = 400;     // LDS: 200 complex values, reused in place by every step
};

inline std::vector<double> build_fix_tables() {
    std::vector<double> t(FixTables::kCount, 0.0);
    const std::vector<double> win = hann_window(400);
    for (int i = 0; i < 400; ++i) t[FixTables::kWin + i] = win[i];
    for (int a = 0; a < 5; ++a)
        for (int c = 0; c < 5; ++c) {
            const double ang = -2.0 * kPi * ((a * c) % 25) / 25.0;
            t[FixTables::kW25 + 2 * (5 * a + c)] = std::cos(ang);
            t[FixTables::kW25 + 2 * (5 * a + c) + 1] = std::sin(ang);
        }
    for (int k = 0; k <= 100; ++k) {
        const double ang = -2.0 * kPi * k / 400.0;
        t[FixTables::kW400 + 2 * k] = std::cos(ang);
        t[FixTables::kW400 + 2 * k + 1] = std::sin(ang);
    }
    return t;
}

// 8-byte load from a pointer that is only 4-byte aligned (as load2_unaligned in whisper_wave.hpp; repeated to keep this header
// free of the f32 kernel's definitions)
MS_DEV void fix_load2(const float *p, float &a, float &b) {
#if defined(__HIPCC__)
    typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));
    const v2u v = *reinterpret_cast<const v2u *>(p);
    a = v.x; b = v.y;
#else
    a = p[0]; b = p[1];
#endif
}

// step 1: window + DFT-8 over n1 for column n2 = lane
MS_DEV void fix_step1(int lane, const float *MS_RESTRICT frame, const double *MS_RESTRICT tab, double *MS_RESTRICT z) {
    if (lane >= 25) return;
    cd u[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = (25 * n1 + 8 * lane) % 200;
        float a, b;
        fix_load2(frame + 2 * n, a, b);
        u[n1] = {static_cast<double>(a) * tab[FixTables::kWin + 2 * n], static_cast<double>(b) * tab[FixTables::kWin + 2 * n + 1]};   // src/stft.rs:163
    }
    fft8(u);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) stc(z + 2 * (25 * k1 + lane), u[k1]);
}

// step 2: DFT-5 over b, twiddle W_25^{a c}
MS_DEV void fix_step2(int lane, const double *MS_RESTRICT tab, double *z) {
    if (lane >= 40) return;
    const int k1 = lane / 5, a = lane - 5 * k1;
    cd v[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) v[b] = ldc(z + 2 * (25 * k1 + 5 * b + a));
    bf5(v[0], v[1], v[2], v[3], v[4]);
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const cd w = ldc(tab + FixTables::kW25 + 2 * (5 * a + c));
        stc(z + 2 * (25 * k1 + 5 * c + a), c == 0 ? v[0] : cmul(v[c], w));
    }
}

// step 3: DFT-5 over a, scatter to natural order
MS_DEV void fix_step3(int lane, double *z) {
    if (lane >= 40) return;
    const int k1 = lane / 5, c = lane - 5 * k1;
    cd t[5];
#pragma unroll
    for (int a = 0; a < 5; ++a) t[a] = ldc(z + 2 * (25 * k1 + 5 * c + a));
    bf5(t[0], t[1], t[2], t[3], t[4]);
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        const int k = (25 * k1 + 176 * (c + 5 * d)) % 200;
        stc(z + 2 * k, t[d]);
    }
}

// step 4: Hermitian split, 4 |X[k]|^2 as f32 (the interval mel weights carry the 1/4, like wave_phase2) into the frame's power row
MS_DEV void fix_step4(int lane, const double *MS_RESTRICT tab, const double *MS_RESTRICT z, float *MS_RESTRICT prow) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = lane + 64 * r;
        if (k > 100) continue;
        const cd zk = ldc(z + 2 * k), zm = ldc(z + 2 * ((200 - k) % 200));
        const cd S = {zk.re + zm.re, zk.im - zm.im};
        const cd D = {zk.re - zm.re, zk.im + zm.im};
        const cd wd = cmul(ldc(tab + FixTables::kW400 + 2 * k), D);
        const double ar = S.re + wd.im, ai = S.im - wd.re;
        const double br = S.re - wd.im, bi = S.im + wd.re;
        prow[k] = static_cast<float>(ar * ar + ai * ai);
        prow[200 - k] = static_cast<float>(br * br + bi * bi);
    }
}

// step 1 in two halves, so that the samples of the NEXT frame to recompute can be in flight during steps 2-4 of this one (they come
// from HBM or the Infinity Cache: the hot loop read them long ago)
struct FixSamples {
    float a[8], b[8];
};
MS_DEV void fix_load_samples(int lane, const float *MS_RESTRICT frame, FixSamples &s) {
    const int l = lane < 25 ? lane : 0;          // every lane loads (no divergent branch around the loads); lanes >= 25 are not used
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) fix_load2(frame + 2 * ((25 * n1 + 8 * l) % 200), s.a[n1], s.b[n1]);
}
// (The window taps of a lane are the same for every frame too; keeping those 16 doubles in registers next to the twiddles below
// was measured: 47 spilled VGPRs in the tail and 1.58 ms instead of 1.21 ms on speech -- they stay a table read per frame.)
MS_DEV void fix_step1(int lane, const FixSamples &s, const double *MS_RESTRICT tab, double *MS_RESTRICT z) {
    if (lane >= 25) return;
    cd u[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        const int n = (25 * n1 + 8 * lane) % 200;
        u[n1] = {static_cast<double>(s.a[n1]) * tab[FixTables::kWin + 2 * n], static_cast<double>(s.b[n1]) * tab[FixTables::kWin + 2 * n + 1]};
    }
    fft8(u);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) stc(z + 2 * (25 * k1 + lane), u[k1]);
}

// The twiddles of steps 2 and 4 depend on the lane only, not on the frame: a wave with frames to recompute loads them once
// (14 doubles) instead of paying two more dependent trips to the global table per frame.
struct FixTw {
    cd w25[5];      // W_25^{a c}, a = lane % 5
    cd w400[2];     // W_400^k, k = lane, lane + 64
};
MS_DEV void fix_load_tw(int lane, const double *MS_RESTRICT tab, FixTw &tw) {
    const int a = lane % 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) tw.w25[c] = ldc(tab + FixTables::kW25 + 2 * (5 * a + c));
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = lane + 64 * r;
        tw.w400[r] = ldc(tab + FixTables::kW400 + 2 * (k <= 100 ? k : 100));
    }
}
MS_DEV void fix_step2(int lane, const FixTw &tw, double *z) {
    if (lane >= 40) return;
    const int k1 = lane / 5, a = lane - 5 * k1;
    cd v[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) v[b] = ldc(z + 2 * (25 * k1 + 5 * b + a));
    bf5(v[0], v[1], v[2], v[3], v[4]);
#pragma unroll
    for (int c = 0; c < 5; ++c) stc(z + 2 * (25 * k1 + 5 * c + a), c == 0 ? v[0] : cmul(v[c], tw.w25[c]));
}
MS_DEV void fix_step4(int lane, const FixTw &tw, const double *MS_RESTRICT z, float *MS_RESTRICT prow) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = lane + 64 * r;
        if (k > 100) continue;
        const cd zk = ldc(z + 2 * k), zm = ldc(z + 2 * ((200 - k) % 200));
        const cd S = {zk.re + zm.re, zk.im - zm.im};
        const cd D = {zk.re - zm.re, zk.im + zm.im};
        const cd wd = cmul(tw.w400[r], D);
        const double ar = S.re + wd.im, ai = S.im - wd.re;
        const double br = S.re - wd.im, bi = S.im + wd.re;
        prow[k] = static_cast<float>(ar * ar + ai * ai);
        prow[200 - k] = static_cast<float>(br * br + bi * bi);
    }
}

}  // namespace melspec
