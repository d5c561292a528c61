// pow2_wave.hpp -- fused log-mel / fbank pipeline for the power-of-two frame sizes the 400- and 512-point kernels do not cover:
// n_fft = 128, 256, 1024, 2048 (Kaldi's fft sizes at 8 / 32 / 44.1 kHz, src/fbank.rs:66-82; NeMo's and most Whisper-style choices).
// Round 3 ran them on generic_frame_kernel -- one frame per 256-thread workgroup, a barrier per radix-2 pass: 0.7-1.3 % of the HBM
// roofline.  Here a frame belongs to a group of LF lanes of ONE wavefront from the PCM to the mel row, like the fused kernels:
//
//   the real n_fft-point transform as a complex M = n_fft / 2 point one (z[n] = x[2n] + i x[2n+1]); every lane holds P = M / LF points;
//   Stockham autosort passes of radix 8 (16 for the first pass of M = 1024), the last pass of radix 2 / 4 / 8, each pass: read the inputs
//   from the frame's LDS region into registers, twiddle, butterfly, write back IN PLACE -- the LDS operations of a wave execute in
//   program order and every lane's reads of a pass are issued before any write of that pass, so no barrier and no second buffer;
//   the Hermitian split to |X[k]|^2 (a pair k, M - k per step), the banded mel sums (jobs of eight weights), log / clamp / store.
//
//   M      n_fft   lanes per frame   frames per wave   passes
//   64     128     8                 8                 8 8
//   128    256     16                4                 8 8 2
//   256    512     32                2                 8 8 4        (only where the 512-point kernels do not apply)
//   512    1024    64                1                 8 8 8
//   1024   2048    64                1                 two 512-point transforms (8 8 8 each: the even and the odd points), the radix-2
//                                                      step between them folded into the split (kHalves; as one transform: 16 8 8)
//
// Arithmetic: f64 from the window multiply to the mel sums, as the reference (src/stft.rs:98-111, src/mel.rs:148-168, src/fbank.rs:
// 165-221) and generic_frame_kernel, which stays the independent on-device cross-check (tests).
//
// LDS banks (tools/lds_bank_model.py prices every access of a frame with gfx950's lane groups; profiles/r04_pow2_lds.txt holds the
// counters before and after).  Element e of a frame's M complex points sits in the 16-byte slot e ^ ((e >> log2 R1) & 7): an XOR
// inside every aligned block of eight, by the index of the block.  The first pass writes e = R1 j + r from eight consecutive lanes j
// (ds_write_b128 is served eight lanes at a time over 32 banks: the XOR turns eight times the same slot into eight different ones);
// every later access is to consecutive elements from consecutive lanes, which the XOR only permutes inside a block, and the 16-lane
// groups of ds_read_b128 {0-3, 12-15, 20-27}, ... meet each of the 16 slots of a 256-byte row once.  The first layout, e + (e >> 3),
// made the same writes conflict-free and every read two-way; the twiddles of the later passes came from the half-circle table at a
// stride of 2 r k slots (2- to 16-way); together 44 % of the LDS cycles at n_fft 1024.  Now the twiddles of a pass are a table of
// their own, [r][k] with k along the lanes.  The frames of a wave are a multiple of 256 bytes apart (128 mod 256 for M = 64, whose
// read groups hold lanes of four frames).
#pragma once
#include "device_fft.hpp"
// Build-time knobs of pow2_frame_kernel (the A/B variants of tools/ab_build.sh; the defaults are what profiles/r04_pow2_lds.txt measured
// best; every line of that file's sections 3, 6, 7 and 9 is one of them):
//   MS_POW2_MAXW / _MAXW16 / _MAXWH   waves per workgroup at most: M <= 512 / M = 1024 as one transform / as two halves
//   MS_POW2_WINLDS                    the window in LDS up to this M (beyond: read from L1 / L2)
//   MS_POW2_AHEAD16 / _AHEADH / _AHEAD_KS   the next frame's samples loaded a frame ahead: at 16 points per lane / in the halves form /
//                                     for the Kaldi flavour at n_fft <= 512 (registers decide: off where the prefetch spills)
//   MS_POW2_JOBS_SMALL / _BIG / _H / _F   rounds of mel jobs in flight: frames of 8-32 lanes / of 64 / the halves form / Kaldi and NeMo
//   MS_POW2_PWALIAS                   from this M on the power row takes the place of the frame's points
//   MS_POW2_HALVES, _HSPLIT           M = 1024 as two 512-point transforms; pairs of the split between scheduling barriers
//   MS_POW2_TW2REG                    pass-2 twiddles in registers (P = 8)
#ifndef MS_POW2_MAXW
#define MS_POW2_MAXW 8
#endif
#ifndef MS_POW2_WINLDS
#define MS_POW2_WINLDS 512
#endif
#ifndef MS_POW2_AHEAD16
#define MS_POW2_AHEAD16 1
#endif
#ifndef MS_POW2_JOBS_F
#define MS_POW2_JOBS_F 1
#endif
#ifndef MS_POW2_MAXW16
#define MS_POW2_MAXW16 4       // M = 1024: 16 points per lane want the AGPRs (one wave per SIMD); at 5-6 waves the 256-VGPR cap spills 100-190 dwords: 11-13 ms against 5.0
#endif
#ifndef MS_POW2_PWALIAS
#define MS_POW2_PWALIAS 1024     // from this M on the power row takes the place of the frame's points (the split has read them all)
#endif
#ifndef MS_POW2_JOBS_SMALL
#define MS_POW2_JOBS_SMALL 1
#endif
#ifndef MS_POW2_JOBS_BIG
#define MS_POW2_JOBS_BIG 3
#endif
#ifndef MS_POW2_HALVES
#define MS_POW2_HALVES 1
#endif
#ifndef MS_POW2_MAXWH
#define MS_POW2_MAXWH 6
#endif
#ifndef MS_POW2_AHEADH
#define MS_POW2_AHEADH 0
#endif
#ifndef MS_POW2_HSPLIT
#define MS_POW2_HSPLIT 4
#endif
#ifndef MS_POW2_JOBS_H
#define MS_POW2_JOBS_H 2
#endif
#ifndef MS_POW2_AHEAD_KS
#define MS_POW2_AHEAD_KS 0
#endif
#ifndef MS_POW2_TW2REG
#define MS_POW2_TW2REG 1
#endif

namespace melspec {

template <int LOGM> struct Pow2Shape {
    static constexpr int M = 1 << LOGM;
    static constexpr int LF = M >= 512 ? 64 : M / 8;         // lanes per frame
    static constexpr int FW = 64 / LF;                       // frames per wave
    static constexpr int P = M / LF;                         // complex points per lane (8; 16 for M = 1024)
    static constexpr int R1 = P;                             // radix of the first pass
    static constexpr int R3 = M / (R1 * 8);                  // radix of the last pass (1: two passes only)
    static constexpr int kShift = P == 16 ? 4 : 3;           // log2 R1
    // M = 1024 as TWO 512-point transforms (even and odd points) and a radix-2 step folded into the split: eight points per lane at
    // a time instead of sixteen -- the registers of the 512-point kernel, two waves per SIMD (pow2_frame_kernel, kHalves)
    static constexpr bool kHalves = MS_POW2_HALVES && M >= 1024;
    static constexpr int kTc = kHalves ? M / 2 : 0;          // complex entries of the radix-2 step's table W_M^k, k < M / 2
    static constexpr int kMelsPerLane = LF >= 16 ? 256 / LF : 16;            // banks of up to 256 mels (128 at M = 64)
    static constexpr int kMaxWaves = M >= 1024 ? (kHalves ? MS_POW2_MAXWH : MS_POW2_MAXW16) : MS_POW2_MAXW;      // two per SIMD: the kernels hold 185-240 VGPRs (M = 1024: one, its LDS holds four frames and its 16 points per lane want the AGPRs)
    static constexpr int kT2 = kHalves ? 0 : (P == 16 || !MS_POW2_TW2REG) ? 7 * R1 : 0;         // complex entries of the pass-2 table (P == 8: the lane keeps its seven in registers)
    static constexpr int kT3 = kHalves ? 7 * 64 : R3 > 1 ? (R3 - 1) * R1 * 8 : 0;               // of the pass-3 table
};

// Where everything is in the workgroup's LDS, in doubles; the host sizes the launch with the same function.
struct Pow2Lds {
    int tw, win, t2, t3, tc, jw, job, frames, frame_stride, pw, acc, total;
};
template <int LOGM> MS_HD Pow2Lds pow2_lds(int n_jobs, int n_mels, int waves) {
    using S = Pow2Shape<LOGM>;
    Pow2Lds o;
    o.tw = 0;
    o.win = o.tw + S::M + 32;                              // W_N^q, q <= M / 2: what the split reads
    o.t2 = o.win + (S::M <= MS_POW2_WINLDS ? 2 * S::M : 0);           // M = 1024: its 16 KB would cost one of four resident waves, read from L1 / L2
    o.t3 = o.t2 + 2 * S::kT2;
    o.tc = o.t3 + 2 * S::kT3;
    o.jw = o.tc + 2 * S::kTc;
    o.job = o.jw + 8 * n_jobs;
    o.frames = (o.job + (n_jobs + 1) / 2 + 31) & ~31;
    // inside a frame: Z, the power row [M + 1] (+ 7 a job may read past it, + pad), the band sums
    o.pw = S::M >= MS_POW2_PWALIAS ? 0 : 2 * S::M;
    o.acc = S::M >= MS_POW2_PWALIAS ? 2 * S::M : o.pw + S::M + 10 + (S::LF < 32 ? 32 - S::LF : 0);      // (the frames of a 32-lane group start their rows LF doubles apart: pow2_pw_shift)
    o.frame_stride = ((o.acc + n_mels + 31) & ~31) + (S::M == 64 ? 16 : 0);
    o.total = o.frames + waves * S::FW * o.frame_stride;
    return o;
}

// doubles by which frame slot fs of a wave shifts its power row.  M = 64: a 16-lane group of ds_read_b128 holds the same eight jobs
// of two pairs of frames (lanes 0-3 and 24-27: frames 0 and 3, lanes 12-15 and 20-23: frames 1 and 2), whose rows must sit eight
// 16-byte slots apart; the frames themselves are 128 bytes apart mod 256.
template <int LOGM> MS_DEV int pow2_pw_shift(int fs) {
    constexpr int LF = Pow2Shape<LOGM>::LF;
    if (LF == 8) return ((fs & 3) == 1 || (fs & 3) == 2) ? 16 : 0;
    return 0;
}

template <int LOGM> MS_DEV int pow2_slot(int e) { return e ^ ((e >> Pow2Shape<LOGM>::kShift) & 7); }

// W_N^q = exp(-2 pi i q / N), N = 2 M, from the table of the half circle tw[q] = W_N^q, q < M
MS_DEV cpx<double> pow2_root(const double *tw, int q, int M) {
    const bool neg = q >= M;
    const cpx<double> w = ldc(tw + 2 * (neg ? q - M : q));
    return neg ? cpx<double>{-w.re, -w.im} : w;
}

// entry idx = (r - 1) * Ns + k of the twiddle table of the pass of radix R after Ns points: W_{Ns R}^{k r} = W_N^{k r N / (Ns R)}, N = 2 M
MS_DEV cpx<double> pow2_table_entry(const double *tw, int M, int R, int Ns, int idx) {
    const int r = idx / Ns + 1, k = idx - (r - 1) * Ns;
    return pow2_root(tw, r * k * (2 * M / (Ns * R)), M);
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
// The twiddles W_{Ns R}^{k r}, k = j mod Ns: from registers (ltw[i * (R - 1) + r - 1] for butterfly i) or from the pass's LDS table
// tab[(r - 1) * Ns + k].
template <int LOGM, int R, bool FIRST>
MS_DEV void pow2_pass(int l, int Ns, const double *tab, double *z, cpx<double> *reg, const cpx<double> *ltw) {
    using S = Pow2Shape<LOGM>;
    constexpr int M = S::M, NB = S::P / R;                   // butterflies per lane
    cpx<double> v[NB][R];
    // The passes work in place: every lane's reads of a pass must be issued before any lane's writes of it, and behind the writes of the
    // pass before.  The hardware runs a wave's LDS operations in program order; these barriers (no instruction) make the PROGRAM order the
    // source order -- without them only the compiler's inability to tell the swizzled addresses apart keeps a load from moving over a store.
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_wave_barrier();
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = l + S::LF * i;
#pragma unroll
        for (int r = 0; r < R; ++r) v[i][r] = FIRST ? reg[r] : ldc(z + 2 * pow2_slot<LOGM>(j + r * (M / R)));
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_wave_barrier();
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = l + S::LF * i;
        const int k = j & (Ns - 1);
        if (!FIRST) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[i][r] = cmul(v[i][r], ltw ? ltw[i * (R - 1) + r - 1] : ldc(tab + 2 * ((r - 1) * Ns + k)));
        }
        pow2_dft<R>(v[i]);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) stc(z + 2 * pow2_slot<LOGM>(j0 + r * Ns), v[i][r]);
    }
}

}  // namespace melspec
