// whisper_wave_f64.hpp -- "precise" build of the fused Whisper kernel: phases 1-2 (window, 400-point
// real FFT, Hermitian split, power) in f64, everything after the power rows shared with the f32 kernel
// (whisper_wave.hpp: interval mel sums, log10, per-frame normalisation).
//
// Why it exists: the f32 kernel is within ~3e-5 of the f64 reference on speech and noise, but the
// per-frame clamp at max-8 lets mel bands 80 dB under the frame maximum through, and for a strong
// line over broadband content ~70 dB down the f32 FFT's rounding noise puts the bands next to the clamp
// up to 4.9e-4 off (tools/flag_calib.py).  This build holds ~1e-6 on everything at about 60 % of the
// throughput.  It runs on every frame of a context in MELSPEC_PRECISION_F64 (and its phases 1-2 are the STFT export,
// whisper400_stft_kernel); the default, MELSPEC_PRECISION_AUTO, recomputes just the frames the f32 kernels' guard trips with the
// wave-cooperative form of whisper_fix64.hpp, inside those kernels.
#pragma once
#include "whisper_wave.hpp"

namespace melspec {

// f64 table part, offsets in doubles (same logical tables as FastBlob's first four)
struct PreciseBlob {
    // Row pitches from the bank model (tools/lds_sim.py rules, 16-byte reads in groups of sixteen lanes over 64 banks): with 44 / 20 / a
    // 464-double frame stride 39 % of the LDS cycles of phases 1-2 were conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 38 % measured);
    // 46 / 22 / 462 leave 22 %.
    static constexpr int kWin = 0;                        // [400]
    static constexpr int kTw1Stride = 46;
    static constexpr int kTw1 = 400;                      // [10][46]  W_200^{t*k1}
    static constexpr int kMod = kTw1 + 10 * kTw1Stride;   // [10] complex W_10^{n2}
    static constexpr int kTw2Stride = 22;
    static constexpr int kTw2 = kMod + 20;                // [11][22] per k = j+20q: the mel kernels' blob holds (2 sin, 4 cos) of W_400^k's angle
                                                          // (the power split below), the spectrum export's blob W_400^k itself
    static constexpr int kCount = kTw2 + kMelJobs * kTw2Stride;
};

struct PreciseLayout {
    static constexpr int kXRow = 22;                      // 10 complex + 1 pad: 44-word row pitch, 11 rows hit 11 bank slots
    static constexpr int kXStride = 21 * kXRow;           // 462 doubles per frame (924 words = 28 mod 64: the frames of a lane group spread over the banks)
    static constexpr int slice_doubles() { return kFPW * kXStride; }   // 2310 doubles; power rows / maxima alias its head
};

// frame: first sample of this lane's frame (frame slot fl of the wave)
MS_DEV void precise_phase1(int fl, int t, bool active, const double *MS_RESTRICT tb, const float *MS_RESTRICT frame, double *MS_RESTRICT rows) {
    if (!active) return;
    const float *s = frame + 2 * t;
    cd x[20];
    // all twenty loads first, then a scheduling barrier (six_phase1, whisper_six.hpp, says why)
    f2 sv[20];
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) sv[n1] = load2_unaligned(s + 20 * n1);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(MS_SCHED_LOADS_FIRST);
#endif
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) {
        const cd w = ldc(tb + PreciseBlob::kWin + 20 * n1 + 2 * t);
        x[n1] = {static_cast<double>(sv[n1].x) * w.re, static_cast<double>(sv[n1].y) * w.im};   // src/stft.rs:163
    }
    fft20(x);
    const double *tw = tb + PreciseBlob::kTw1 + t * PreciseBlob::kTw1Stride;
    double *xo = rows + fl * PreciseLayout::kXStride + 2 * t;
    stc(xo, x[0]);
    stc(xo + 20 * PreciseLayout::kXRow, cmul(x[0], ldc(tb + PreciseBlob::kMod + 2 * t)));
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) stc(xo + k1 * PreciseLayout::kXRow, cmul(x[k1], ldc(tw + 2 * k1)));
}

// writes 4*|X|^2 as f32 power rows (stride WaveLayout::kPStride) over the head of the same slice
MS_DEV void precise_phase2(int fl, int j, bool active, const double *MS_RESTRICT tb, double *MS_RESTRICT rows) {
    if (!active) return;
    const int brow = (j == 0) ? 20 : 20 - j;
    const double *ua = rows + fl * PreciseLayout::kXStride + j * PreciseLayout::kXRow;
    const double *va = rows + fl * PreciseLayout::kXStride + brow * PreciseLayout::kXRow;
    cd u[10], v[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        u[i] = ldc(ua + 2 * i);
        v[i] = ldc(va + 2 * i);
    }
    fft10(u);
    fft10(v);
    // Hermitian split straight to the two powers.  With S = zk + conj(zm), D = zk - conj(zm) and w = W_400^k (|w| = 1):
    //   4|X[k]|^2, 4|X[200-k]|^2 = |S|^2 + |D|^2 +- 2 Im(conj(S) D w),   |S|^2 + |D|^2 = 2(|zk|^2 + |zm|^2),
    //   conj(S) D = (|zk|^2 - |zm|^2) + 2i (zk.re zm.im + zm.re zk.im)
    // -- 12 operations per pair instead of 16 (S, D, w D, A, B, |A|^2, |B|^2).  The difference of two large numbers in the weaker of
    // the two bins costs 1.1e-16 * P_strong / P_weak of relative accuracy; Whisper's clamp at max - 8 decades (src/mel.rs:645-654) bounds
    // what matters of that ratio by ~1e8 per band, so the result moves by <= 1e-8 -- the f32 kernels cannot afford this form (6e-8 * 1e8),
    // nor can the fbank flavours of the 512-point kernel (no clamp).
    const double *tw = tb + PreciseBlob::kTw2 + j * PreciseBlob::kTw2Stride;
    float *p = reinterpret_cast<float *>(rows) + fl * WaveLayout::kPStride;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const cd zk = u[q], zm = v[9 - q];
        const double a = zk.re * zk.re + zk.im * zk.im, b = zm.re * zm.re + zm.im * zm.im;
        const double c = zk.re * zm.im + zm.re * zk.im;
        const cd t = ldc(tw + 2 * q);                       // (2 sin, 4 cos)
        const double s2 = a + b, x = (a - b) * t.re + c * t.im;
        p[j + 20 * q] = static_cast<float>(2.0 * s2 + x);
        p[200 - j - 20 * q] = static_cast<float>(2.0 * s2 - x);
    }
}

// STFT export (Spectrogram::compute_all_cpu, src/stft.rs:89-115): phase 2 with the spectrum itself as the result.
// Lane (frame, j) holds 2*X[k] and 2*conj(X[200-k]) for k = j + 20q, q < 10 -- between the eleven lanes every bin 0..200
// (the Nyquist bin is the partner of k = 0).  out: this lane's frame in global memory, `bins` complex values of type T
// (201: the half spectrum; 400: the reference's full layout, the upper half being the conjugate mirror X[400-k] = conj X[k]).
template <class T>
MS_DEV void precise_phase2_spectrum(int fl, int j, bool active, const double *MS_RESTRICT tb, const double *MS_RESTRICT rows,
                                    T *MS_RESTRICT out, int bins) {
    if (!active) return;
    const int brow = (j == 0) ? 20 : 20 - j;
    const double *ua = rows + fl * PreciseLayout::kXStride + j * PreciseLayout::kXRow;
    const double *va = rows + fl * PreciseLayout::kXStride + brow * PreciseLayout::kXRow;
    cd u[10], v[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        u[i] = ldc(ua + 2 * i);
        v[i] = ldc(va + 2 * i);
    }
    fft10(u);
    fft10(v);
    const double *tw = tb + PreciseBlob::kTw2 + j * PreciseBlob::kTw2Stride;
    const bool full = bins > 201;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const cd zk = u[q], zm = v[9 - q];
        const cd S = {zk.re + zm.re, zk.im - zm.im};
        const cd D = {zk.re - zm.re, zk.im + zm.im};
        const cd wd = cmul(ldc(tw + 2 * q), D);
        const double ar = 0.5 * (S.re + wd.im), ai = 0.5 * (S.im - wd.re);      // X[k]
        const double br = 0.5 * (S.re - wd.im), bi = -0.5 * (S.im + wd.re);     // X[200 - k]
        const int k = j + 20 * q, m = 200 - k;
        stc(out + 2 * k, cpx<T>{static_cast<T>(ar), static_cast<T>(ai)});
        stc(out + 2 * m, cpx<T>{static_cast<T>(br), static_cast<T>(bi)});
        if (full) {
            if (k > 0) stc(out + 2 * (400 - k), cpx<T>{static_cast<T>(ar), static_cast<T>(-ai)});
            if (m < 200) stc(out + 2 * (400 - m), cpx<T>{static_cast<T>(br), static_cast<T>(-bi)});
        }
    }
}

}  // namespace melspec
