// pow2_wave.hpp -- fused log-mel / fbank pipeline for the power-of-two frame sizes the 400- and 512-point kernels do not cover:
// n_fft = 128, 256, 1024, 2048 (Kaldi's fft sizes at 8 / 32 / 44.1 kHz, src/fbank.rs:66-82; NeMo's and most Whisper-style choices).
// Round 3 ran them on generic_frame_kernel -- one frame per 256-thread workgroup, a barrier per radix-2 pass: 0.7-1.3 % of the HBM
// roofline.  Here a frame belongs to a group of LF lanes of ONE wavefront from the PCM to the mel row, like the fused kernels:
//
//   the real n_fft-point transform as a complex M = n_fft / 2 point one (z[n] = x[2n] + i x[2n+1]); every lane holds P = M / LF points;
//   Stockham autosort passes of radix 8 (16 for the first pass of M = 1024), the last pass of radix 2 / 4 / 8, each pass: read the inputs
//   from the frame's LDS region into registers, twiddle, butterfly, write back IN PLACE -- the LDS operations of a wave execute in
//   program order and every lane's reads of a pass are issued before any write of that pass, so no barrier and no second buffer;
//   the Hermitian split to |X[k]|^2, the banded mel sums (one mel per lane and step), log / clamp / store.
//
//   M      n_fft   lanes per frame   frames per wave   passes
//   64     128     8                 8                 8 8
//   128    256     16                4                 8 8 2
//   256    512     32                2                 8 8 4        (only where the 512-point kernels do not apply)
//   512    1024    64                1                 8 8 8
//   1024   2048    64                1                 16 8 8
//
// Arithmetic: f64 from the window multiply to the mel sums, as the reference (src/stft.rs:98-111, src/mel.rs:148-168, src/fbank.rs:
// 165-221) and generic_frame_kernel, which stays the independent on-device cross-check (tests).  Element e of a frame's M complex
// points sits at e + (e >> 3): the pass writes of eight lanes land 9 instead of 8 elements apart (144 B: conflict-free 16-byte writes).
#pragma once
#include "device_fft.hpp"

namespace melspec {

template <int LOGM> struct Pow2Shape {
    static constexpr int M = 1 << LOGM;
    static constexpr int LF = M >= 512 ? 64 : M / 8;         // lanes per frame
    static constexpr int FW = 64 / LF;                       // frames per wave
    static constexpr int P = M / LF;                         // complex points per lane (8; 16 for M = 1024)
    static constexpr int R1 = P;                             // radix of the first pass
    static constexpr int R3 = M / (R1 * 8);                  // radix of the last pass (1: two passes only)
    static constexpr int kZ = M + (M >> 3);                  // padded complex points per frame
    static constexpr int frame_doubles() { return 2 * kZ + M + 2; }      // Z, then the power row [M + 1] (+ 1 pad)
    static constexpr int kMelsPerLane = LF >= 16 ? 256 / LF : 16;            // banks of up to 256 mels (128 at M = 64)
    static constexpr int kWaves = M >= 1024 ? 3 : 4;                         // waves per workgroup (M = 1024: 27 KB of LDS per frame)
};

MS_DEV int pow2_pad(int e) { return e + (e >> 3); }

// W_N^q = exp(-2 pi i q / N), N = 2 M, from the table of the half circle tw[q] = W_N^q, q < M
MS_DEV cpx<double> pow2_root(const double *tw, int q, int M) {
    const bool neg = q >= M;
    const cpx<double> w = ldc(tw + 2 * (neg ? q - M : q));
    return neg ? cpx<double>{-w.re, -w.im} : w;
}

template <int R> MS_DEV void pow2_dft(cpx<double> (&v)[R]);
template <> MS_DEV void pow2_dft<2>(cpx<double> (&v)[2]) { bf2(v[0], v[1]); }
template <> MS_DEV void pow2_dft<4>(cpx<double> (&v)[4]) {
    bf4(v[0], v[1], v[2], v[3]);
}
template <> MS_DEV void pow2_dft<8>(cpx<double> (&v)[8]) { fft8(v); }
template <> MS_DEV void pow2_dft<16>(cpx<double> (&v)[16]) { fft16(v); }

// One Stockham pass of radix R over the frame's M points (Ns = product of the radices before it), butterflies j = l + LF * i.
// FIRST: the inputs are already in `reg` (the windowed samples of the lane, reg[r] = z[l + r * M / R]), nothing is read.
// ltw != nullptr: the lane's twiddles of this pass from registers, ltw[i * (R - 1) + r - 1] for butterfly i.
template <int LOGM, int R, bool FIRST>
MS_DEV void pow2_pass(int l, int Ns, const double *tw, double *z, cpx<double> *reg, const cpx<double> *ltw) {
    using S = Pow2Shape<LOGM>;
    constexpr int M = S::M, NB = S::P / R;                   // butterflies per lane
    cpx<double> v[NB][R];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = l + S::LF * i;
#pragma unroll
        for (int r = 0; r < R; ++r) v[i][r] = FIRST ? reg[r] : ldc(z + 2 * pow2_pad(j + r * (M / R)));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = l + S::LF * i;
        const int k = j & (Ns - 1);
        if (!FIRST) {
            const int step = k * (2 * M / (Ns * R));         // W_{Ns R}^{k r} = W_N^{k r N / (Ns R)}
#pragma unroll
            for (int r = 1; r < R; ++r) v[i][r] = cmul(v[i][r], ltw ? ltw[i * (R - 1) + r - 1] : pow2_root(tw, r * step, M));
        }
        pow2_dft<R>(v[i]);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) stc(z + 2 * pow2_pad(j0 + r * Ns), v[i][r]);
    }
}

}  // namespace melspec
