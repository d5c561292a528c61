// fbank_wave.hpp -- fused Kaldi-fbank frame pipeline (default geometry: 400-sample frames, 512-point
// FFT), wave-autonomous like whisper_wave.hpp: one wavefront owns kFbFPW = 4 whole frames,
// lane = 16*frame + j (one DPP row per frame).  Reference: Fbank::compute, src/fbank.rs:141-236.
//
// Shape: the exchange rows are f64, 4.6 KB per frame, so frames per wave decide the occupancy.  An earlier
// build ran 7 frames x 9 lanes per wave (each lane two columns in phase 1): 32.5 KB of LDS per wave, 4
// waves per CU, one per SIMD, and the lone wave spent 2/3 of its time waiting on its own LDS/global
// latencies.  4 frames x 16 lanes gives every lane one column in phase 1, 18.6 KB per wave and two waves
// per SIMD; phase 2 has 9 jobs per frame, so 7 of 16 lanes idle there.
//
// Arithmetic type T.  Unlike the Whisper path there is no per-frame clamp here: ln(E) of a mel band
// 90 dB below the frame's strongest band is an output, and rounding the *windowed frame itself* to
// f32 already puts a noise floor ~-149 dB per bin under the frame energy, i.e. ~2e-3 relative error on
// such a band (measured on jfk_f32le.wav: 2.1e-3).  The reference computes in f64 (src/fbank.rs:154-158),
// so the parity build is T = double from the window multiply to |X|^2; T = float is kept for
// throughput experiments only.
//
//   DC removal + pre-emphasis + Povey window (src/fbank.rs:164-190)
//       y[i] = (x[i]-m) - a*(x[i-1]-m) = x[i] - a*x[i-1] - (1-a)*m,   frame[i] = y[i]*w[i]
//       (for the very first sample of a clip the reference applies no pre-emphasis: y[0] = x[0]-m,
//        which is the same formula with x[-1] := m)
//   zero-pad to 512, forward FFT (src/fbank.rs:184-194): real-512 as complex-256 = 16 x 16,
//       z[n] = x[2n] + i*x[2n+1];  n = 16*n1 + n2,  k = k1 + 16*k2
//       phase 1: lane t does the 16-point DFT over n1 for column n2 = t (inputs beyond sample 399 are
//                literal zeros), multiplies by W_256^{n2*k1}, writes row k1
//       phase 2: lane r does the 16-point DFT over n2 of row r (Z[r + 16*k2]), fetches the upper half of
//                its Hermitian partner's DFT (row 16-r, lane (16-r)&15 of the same DPP row) with two DPP
//                moves per dword, and splits 8 pairs -> bins k = r+16s and 256-k, s < 8 (lanes 0 and 8
//                are their own partners; lane 0 pairs Z[16s] with Z[256-16s] and does a 9th split)
//   power, bins 0..=256 (src/fbank.rs:197-203), stored as f32 (sums of non-negative terms from here on)
//   sparse mel, floor, ln (src/fbank.rs:205-221): interval scheme, 16 lanes per frame (15 intervals + ghost)
//   CMN (src/fbank.rs:224-233) is a second kernel (cmn_kernel) because it is a per-clip reduction.
#pragma once
#include "whisper_wave.hpp"

namespace melspec {

constexpr int kFbFPW = 4;        // frames per wavefront
constexpr int kFbLanes = 16;     // lanes per frame: one column each in phase 1; lane 15 is the ghost in phase 3
constexpr int kFbOwn = kFbLanes - 1;   // intervals a 16-lane group owns per slot
constexpr int kFbSlots = 6;      // intervals j + 15*slot, up to 89 mel bins (Kaldi fbank)
constexpr int kBlmSlots = 10;    // up to 149 mel bins (NeMo/Parakeet uses 80 or 128)

// Table blob: a T-typed part (offsets in units of T) followed by the f32/int mel section.
struct FbankBlob {
    static constexpr int kWin = 0;                         // [512] window taps (400 used by the Kaldi / NeMo flavours)
    // row pitches of the two twiddle tables in 16-byte slots: 17 and 9, odd -- the sixteen lanes ds_read_b128 serves together (rows
    // {0-3, 12-15} of one frame, {4-11} of the next) hit sixteen different slots; 18 and 10 (round 3) made rows t and t + 8 collide:
    // 2-way on every twiddle read, 120 of ~1900 LDS cycles per unit (same-box A/B: NeMo -1.7 %, fbank / Whisper-512 +-0.2 %)
#ifndef MS_FB_TW1S
#define MS_FB_TW1S 34
#endif
#ifndef MS_FB_TW2S
#define MS_FB_TW2S 18
#endif
    static constexpr int kTw1Stride = MS_FB_TW1S;          // 16 complex + pad
    static constexpr int kTw1 = 512;                       // [16 n2][36] W_256^{n2*k1}
    static constexpr int kTw2Stride = MS_FB_TW2S;          // 9 complex + pad
    static constexpr int kTw2 = kTw1 + 16 * kTw1Stride;    // [16 r][20] complex W_512^{r+16s}, s = 0..8
    static constexpr int kTCount = kTw2 + 16 * kTw2Stride; // 1344 elements of T
    // mel section, float offsets from its own base
    static constexpr int kMelStart = 0;                                // [kBlmSlots*16] ints
    static constexpr int kMelW = (kBlmSlots * kFbLanes + 3) & ~3;      // pairs [slot][r][16][2]
};

template <class T>
struct FbankLayout {
    // exchange rows, units of T; padded so that the 9 job lanes of a frame reading 9 different rows hit
    // different banks (f64: 16-byte complex reads, rows 68 words apart)
    // f32 (MELSPEC_PRECISION_F32, round 5): unpadded 128-byte rows whose eight 16-byte slots are XOR-swizzled with the row number --
    // the sixteen lanes that read their rows with ds_read_b128 meet eight different slots twice (256 bytes over a 128-byte port: nothing
    // to gain), the sixteen lanes that write a row fill it; a frame is 2 KB, a wave's slice 8 KB (rows 144 bytes apart: 9 KB, and the
    // staged feature-major store of the NeMo flavour would not fit beside twelve slices)
    static constexpr int kXRow = sizeof(T) == 8 ? 34 : 32;
    static constexpr int kXStride = 16 * kXRow;            // lanes of different frames never share an LDS access group
    // offset (units of T) of element n2 of exchange row k1 from the frame's first row
    MS_HD static constexpr int xoff(int k1, int n2) {
        return sizeof(T) == 8 ? k1 * kXRow + 2 * n2 : k1 * kXRow + ((((n2 >> 1) ^ (k1 & 7)) << 2) | ((n2 & 1) << 1));
    }
    static constexpr int kPStride = 259;                   // f32 power rows (bins 0..256), aliased over the rows
    static constexpr int kSumOff = 0;                      // 64 partial sums (units of T): the host form of the frame mean in tests/emu (the kernels: row_sum16)
    static constexpr int slice_elems() { return (kFbFPW * kXStride + 1) & ~1; }   // units of T
};

// frame-sum partial: lane (f,t) adds the samples of its own column, pairs 32*n1 + 2t + {0,1} below 400
// All 13 loads are unconditional (lanes t >= 8 re-read pair 0 for the 13th and drop it), so the compiler issues them
// back to back; a guarded 13th load costs a separate memory round trip per unit.
template <class T>
MS_DEV T fb_partial_sum(const float *frame, int t) {
    f2 a[13];
#pragma unroll
    for (int n1 = 0; n1 < 12; ++n1) a[n1] = load2_unaligned(frame + 32 * n1 + 2 * t);
    a[12] = load2_unaligned(frame + (t < 8 ? 384 + 2 * t : 0));
    T s = 0;
#pragma unroll
    for (int n1 = 0; n1 < 12; ++n1) s += static_cast<T>(a[n1].x) + static_cast<T>(a[n1].y);
    if (t < 8) s += static_cast<T>(a[12].x) + static_cast<T>(a[12].y);
    return s;
}

// DFT over n1 of one column, twiddle by W_256^{n2*k1}, write the 16 exchange rows.
// FRESH (the f32 NeMo kernel, which has no register to spare): the swizzled offsets are recomputed per unit, see fresh_lane_value
template <class T, bool FRESH = false>
MS_DEV void fb_column_finish(cpx<T> (&x)[16], int n2, const T *MS_RESTRICT tblob, T *MS_RESTRICT xf /* the frame's first exchange row */) {
    using L = FbankLayout<T>;
    fft16(x);
    const T *tw = tblob + FbankBlob::kTw1 + n2 * FbankBlob::kTw1Stride;
    const int e = FRESH ? fresh_lane_value(n2) : n2;
    stc(xf + L::xoff(0, e), x[0]);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) stc(xf + L::xoff(k1, e), cmul(x[k1], ldc(tw + 2 * k1)));
}

// One 16-point column n2: samples frame[32*n1 + 2*n2 + {0,1}] below 400 (the rest of the 512-point frame is
// zero padding), pre-emphasis, DC removal, window, DFT over n1, twiddle by W_256^{n2*k1}, exchange rows.
template <class T>
MS_DEV void fb_column(const float *frame, int n2, T preemph, T mean, bool patch_first, const T *tblob, T *xo /* the frame's first exchange row */) {
    cpx<T> x[16];
    const T dc = (T(1) - preemph) * mean;
    // every load first and unconditional: 13 pairs (lanes n2 >= 8 have no 13th pair: they re-read pair 0 and drop it) and
    // the sample in front of each pair (the first sample of a clip has none: it re-reads itself and the value is unused)
    f2 c[13];
    float prev[13];
#pragma unroll
    for (int n1 = 0; n1 < 13; ++n1) {
        const int i = (n1 < 12 || n2 < 8) ? 32 * n1 + 2 * n2 : 0;
        c[n1] = load2_unaligned(frame + i);
        prev[n1] = frame[(n1 == 0 && patch_first) ? 0 : i - 1];
    }
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        x[n1] = {T(0), T(0)};
        if (n1 < 12 || (n1 == 12 && n2 < 8)) {
            const int i = 32 * n1 + 2 * n2;
            // frame_buf[i] = x[i] - mean; frame_buf[i] -= preemph * frame_buf[i-1]  (src/fbank.rs:165-181);
            // the first sample of a clip gets no pre-emphasis
            // = (x[i] - m) - a (x[i-1] - m) = x[i] - a x[i-1] - (1 - a) m: two operations per sample (dc = (1 - a) m)
            const T x0 = static_cast<T>(c[n1].x), x1 = static_cast<T>(c[n1].y);
            const T pe = (x0 - preemph * static_cast<T>(prev[n1])) - dc;
            const T y0 = (n1 == 0 && patch_first) ? x0 - mean : pe;
            const T y1 = (x1 - preemph * x0) - dc;
            const cpx<T> w = ldc(tblob + FbankBlob::kWin + i);
            x[n1] = {y0 * w.re, y1 * w.im};
        }
    }
    fb_column_finish<T>(x, n2, tblob, xo);
}

// (The sample in front of each pair as a row_ror:1 DPP move of the left neighbour's second sample instead of 12 more loads per lane:
// 0.7101 vs 0.7103 ms, neutral -- the loads hit L1.)
// (Round 5, the same for the NeMo flavour's 13 predecessor samples: f32 kernel +0.8 % / +1.3 % (128 / 80 mels), f64 +0.3 / +1.4 % -- slower.
// Without pre-emphasis the f32 kernel runs 6 % faster (0.4255 vs 0.452 ms): that is the 26 separately rounded products, not the loads.)
// (Round 2 measured "the samples loaded once for both the frame sum and the column" as no difference.  Round 5 built it again as
// fb_kaldi_input below -- every load unconditional AND branch-free, so that nothing of the column sits behind the mean's round trip -- and it
// is -10 %: config 3 0.709 -> 0.642 ms, the kernel without CMN 0.616 -> 0.549 ms, same box.  fb_partial_sum / fb_column stay for the host
// emulation, which sums the partials across its emulated lanes itself; the arithmetic is the same operation for operation.)
// phase 1 (after the frame mean is known): lane t does column n2 = t (13 non-zero inputs for t < 8, else 12).
template <class T>
MS_DEV void fb_phase1(int fl, int t, bool active, const float *frame /* this frame's first sample */, bool clip_start,
                      T mean, T preemph, const T *tblob, T *slice) {
    if (!active) return;
    fb_column<T>(frame, t, preemph, mean, clip_start && t == 0, tblob, slice + fl * FbankLayout<T>::kXStride);
}

// ---- Whisper flavour with n_fft = 512 (Spectrogram::compute_mel_spectrogram_cpu, src/stft.rs:119-138, at the
// geometry the reference's RingBuffer golden test and its WGPU tests use: 512/160/80) --------------------------
// Frame = 512 samples, periodic Hann(512), no pre-emphasis / DC removal: every one of the 16 inputs of a column is a
// real sample pair.  Phase 2 is shared; the epilogue is Whisper's log10 / max-8 clamp / (x+4)/4.
// PIN (w512_auto_kernel, round 6): all sixteen loads first, in a loop of their own with a scheduling barrier behind it -- six_phase1
// (whisper_six.hpp) says why: left to itself the machine scheduler put them into the butterflies two at a time behind
// s_waitcnt vmcnt(1) in the 128-mel instance of that kernel's unit loop: sixteen serialised round trips per unit, 0.55 ms instead of 0.37.
template <class T, bool PIN = false>
MS_DEV void w512_phase1(int fl, int t, bool active, const float *frame, const T *tblob, T *slice) {
    if (!active) return;
    cpx<T> x[16];
    if (PIN) {
        f2 sv[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) sv[n1] = load2_unaligned(frame + 32 * n1 + 2 * t);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(MS_SCHED_LOADS_FIRST);
#endif
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const cpx<T> w = ldc(tblob + FbankBlob::kWin + 32 * n1 + 2 * t);
            x[n1] = {static_cast<T>(sv[n1].x) * w.re, static_cast<T>(sv[n1].y) * w.im};
        }
    } else {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int i = 32 * n1 + 2 * t;
            const f2 s = load2_unaligned(frame + i);
            const cpx<T> w = ldc(tblob + FbankBlob::kWin + i);
            x[n1] = {static_cast<T>(s.x) * w.re, static_cast<T>(s.y) * w.im};
        }
    }
    fb_column_finish<T>(x, t, tblob, slice + fl * FbankLayout<T>::kXStride);
}

// ---- NeMo/Parakeet flavour (BatchLogMelSpectrogram, src/mel.rs:299-385) --------------------------
// The 400 window taps sit at positions 56..455 of the 512-point frame; a circular shift does not change
// |X|, so they are processed at positions 0..399 exactly like the Kaldi frame.  Sample `s` of the clip is
// the pre-emphasised waveform (f32, two roundings like `current - (coeff * prev)`, src/mel.rs:696-706),
// zero outside [0, len) (centre padding, src/mel.rs:685-694).
// (nemo_sample(s): 0 for s outside [0, len); clip[s] for s == 0 or coeff == 0; clip[s] - f32(coeff * clip[s - 1]) otherwise.)

// `inside`: every sample this frame touches, and the one before its first, lies inside the clip (all frames
// but the first two and last two of a centred clip), so the guards and the clip-start special case drop out
// and the pair comes from one 8-byte load.
// INSIDE / PRE are wave-uniform and resolved outside the unrolled loop: a data-dependent `coeff == 0 ? a : b` around the
// opaque rounding barrier of f32_mul_rn becomes a branch per sample with an s_waitcnt vmcnt(0) in front of it, i.e. 26
// serialised memory round trips per unit (measured: 1.51 ms per launch against 0.95 ms for this form).
template <class T, bool INSIDE, bool PRE>
MS_DEV void nemo_column(const float *clip, long long org, long long len, int n2, float coeff, const T *tblob, T *xo) {
    cpx<T> x[16];
    f2 c[13];
    float prev[13];
    // the clip in the frame's own coordinates: sample k of the frame is clip[org + k], inside the clip for lo <= k < hi (32-bit: the
    // frame is 400 samples long; a frame that touches its clip at all has lo < 400 and hi > 0)
    const long long lo64 = -org, hi64 = len - org;
    const int lo = lo64 < -1024 ? -1024 : (lo64 > 1024 ? 1024 : static_cast<int>(lo64));
    const int hi = hi64 < -1024 ? -1024 : (hi64 > 1024 ? 1024 : static_cast<int>(hi64));
    if (INSIDE) {
        // interior frame: every load first (13 pairs and the 13 samples in front of them), arithmetic afterwards
#pragma unroll
        for (int n1 = 0; n1 < 13; ++n1) {
            c[n1] = {0.0f, 0.0f};
            prev[n1] = 0.0f;
            if (n1 < 12 || n2 < 8) {
                const float *s = clip + org + 32 * n1 + 2 * n2;
                c[n1] = load2_unaligned(s);
                if (PRE) prev[n1] = s[-1];
            }
        }
    } else {
        // a frame at an end of its clip (two units per clip): the same loads from CLAMPED positions, all of them issued before anything
        // is used, and nemo_sample's cases as selects below.  (Round 5; the form before -- nemo_sample per sample, a branch and a wait
        // around each of its 52 loads -- was a chain of memory round trips, and since round 5's staged store every wave of the workgroup
        // waits at most one round for the wave that walks it.)
#pragma unroll
        for (int n1 = 0; n1 < 13; ++n1) {
            const int i = 32 * n1 + 2 * n2;
            auto at = [&](int k) { return clip[org + (k < lo ? lo : (k >= hi ? hi - 1 : k))]; };
            prev[n1] = at(i - 1);
            c[n1] = {at(i), at(i + 1)};
        }
    }
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        x[n1] = {T(0), T(0)};
        if (n1 < 12 || (n1 == 12 && n2 < 8)) {
            const int i = 32 * n1 + 2 * n2;
            float y0, y1;
            if (INSIDE) {
                y0 = PRE ? c[n1].x - f32_mul_rn(coeff, prev[n1]) : c[n1].x;
                y1 = PRE ? c[n1].y - f32_mul_rn(coeff, c[n1].x) : c[n1].y;
            } else {
                // nemo_sample(s): 0 outside [0, len); clip[0] for s == 0; clip[s] - coeff * clip[s - 1] otherwise (coeff == 0: the product is
                // an exact zero of either sign and cur - (+-0) == cur, except that -0 - (+0) keeps its sign: the same bits as `cur`)
                const float p0 = c[n1].x - f32_mul_rn(coeff, prev[n1]), p1 = c[n1].y - f32_mul_rn(coeff, c[n1].x);
                y0 = (i < lo || i >= hi) ? 0.0f : ((i == lo || coeff == 0.0f) ? c[n1].x : p0);
                y1 = (i + 1 < lo || i + 1 >= hi) ? 0.0f : ((i + 1 == lo || coeff == 0.0f) ? c[n1].y : p1);
            }
            const cpx<T> w = ldc(tblob + FbankBlob::kWin + i);
            x[n1] = {static_cast<T>(y0) * w.re, static_cast<T>(y1) * w.im};
        }
    }
    fb_column_finish<T, sizeof(T) == 4>(x, n2, tblob, xo);
}

// all_inside: wave-uniform -- every active frame of the wave is an interior frame (the guarded form runs only for the
// units at the two ends of a clip)
template <class T>
MS_DEV void nemo_phase1(int fl, int t, bool active, bool all_inside, const float *clip, long long org, long long len, float coeff,
                        const T *tblob, T *slice) {
    T *xo = slice + fl * FbankLayout<T>::kXStride;
    if (all_inside) {
        if (coeff != 0.0f) {
            if (active) nemo_column<T, true, true>(clip, org, len, t, coeff, tblob, xo);
        } else {
            if (active) nemo_column<T, true, false>(clip, org, len, t, coeff, tblob, xo);
        }
    } else {
        if (active) nemo_column<T, false, true>(clip, org, len, t, coeff, tblob, xo);
    }
}

// Value held by lane (16 - r) & 15 of the caller's 16-lane row (r = lane & 15): row_mirror (lane i <- 15-i)
// followed by row_ror:1 (lane i <- i-1).  Frames occupy whole rows, so source and destination lanes always
// share their EXEC state.
#if defined(__HIP_DEVICE_COMPILE__)
// (One ds_bpermute_b32 per dword instead of the two DPP moves was measured too: Whisper-512 0.58 ms against 0.55 ms.)
MS_DEV int dpp_partner16(int x) {
    // bound_ctrl: every lane has a source lane in its row, so no "old" value (and no v_mov to set one up) is needed
    const int m = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, true);
    return __builtin_amdgcn_update_dpp(m, m, 0x121, 0xf, 0xf, true);
}
MS_DEV float partner16(float v) { return __int_as_float(dpp_partner16(__float_as_int(v))); }
MS_DEV double partner16(double v) {
    return __hiloint2double(dpp_partner16(__double2hiint(v)), dpp_partner16(__double2loint(v)));
}
#else
template <class T> MS_DEV T partner16(T v) { return v; }      // host pass of the kernel source only; tests/emu exchanges explicitly
#endif

// Sum of the 16 values of the caller's row in the fixed tree ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)) + (the same over 8..15), delivered to
// every lane of the row: four butterfly levels of DPP moves (swap within pairs, swap pairs within quads, row_half_mirror, row_mirror)
// -- a + b == b + a exactly, so the lane that adds "the other half first" gets the same bits.  Replaces 16 partial sums through LDS
// and 15 f64 additions per lane.
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL> MS_DEV double dpp_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> MS_DEV float dpp_f64(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <class T> MS_DEV T row_sum16(T v) {
    v = v + dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
    v = v + dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
    v = v + dpp_f64<0x141>(v);       // row_half_mirror
    v = v + dpp_f64<0x140>(v);       // row_mirror
    return v;
}
#else
template <class T> MS_DEV T row_sum16(T v) { return v; }      // host pass of the kernel source only; tests/emu sums the partials itself
#endif

// Kaldi input with ONE set of loads: fb_partial_sum + row_sum16 + fb_column's framing -- 13 pairs and the 13 samples in front of them stay
// in f32 registers (39) for both the frame sum and the column, so the column does not start with a second memory round trip behind the
// mean.  Same operations in the same order as the two functions (bit-identical).  Every load unconditional and branch-free: the 13th pair of lanes
// n2 >= 8 re-reads pair 0 and is dropped by selects (a guarded load is a memory round trip of its own).
template <class T>
MS_DEV void fb_kaldi_input(const float *frame, int n2, T preemph, bool patch_first, const T *tblob, cpx<T> (&x)[16]) {
    f2 c[13];
    float prev[13];
#pragma unroll
    for (int n1 = 0; n1 < 13; ++n1) {
        const int i = (n1 < 12 || n2 < 8) ? 32 * n1 + 2 * n2 : 2;       // (not 0: its predecessor would be frame[-1], which for the first frame of a buffer is not mapped)
        c[n1] = load2_unaligned(frame + i);
        prev[n1] = frame[(n1 == 0 && patch_first) ? 0 : i - 1];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(MS_SCHED_LOADS_FIRST);
#endif
    const bool has13 = n2 < 8;
    T s = 0;
#pragma unroll
    for (int n1 = 0; n1 < 12; ++n1) s += static_cast<T>(c[n1].x) + static_cast<T>(c[n1].y);
    {
        const T t13 = static_cast<T>(c[12].x) + static_cast<T>(c[12].y);
        s = has13 ? s + t13 : s;
    }
#ifdef MS_KALDI_MEAN_DIV
    const T mean = row_sum16<T>(s) / T(400);                 // src/fbank.rs:165-166
#else
    // sum * (1 / 400) for src/fbank.rs:165-166's sum / 400: at most one ulp of an f64 away, and a dozen instructions (v_div_scale, v_rcp,
    // the Newton steps, v_div_fmas, v_div_fixup) shorter per unit
    const T mean = row_sum16<T>(s) * (T(1) / T(400));
#endif
    const T dc = (T(1) - preemph) * mean;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        x[n1] = {T(0), T(0)};
        if (n1 < 13) {
            const int i = 32 * n1 + 2 * n2;                  // (n1 == 12, n2 >= 8: past the frame -- the window read stays inside its 512 entries, the result is dropped)
            const T x0 = static_cast<T>(c[n1].x), x1 = static_cast<T>(c[n1].y);
            const T pe = (x0 - preemph * static_cast<T>(prev[n1])) - dc;
            const T y0 = (n1 == 0 && patch_first) ? x0 - mean : pe;
            const T y1 = (x1 - preemph * x0) - dc;
            const cpx<T> w = ldc(tblob + FbankBlob::kWin + i);
            const T a = y0 * w.re, b = y1 * w.im;
            if (n1 < 12) x[n1] = {a, b};
            else x[n1] = {has13 ? a : T(0), has13 ? b : T(0)};
        }
    }
}


// phase 2a: this lane's row of the exchange buffer through a 16-point DFT: own[k2] = Z[r + 16*k2]
template <class T, bool FRESH = false>
MS_DEV void fb_phase2_dft(int fl, int r, bool active, const T *slice, cpx<T> (&own)[16]) {
    if (!active) return;
    const T *row = slice + fl * FbankLayout<T>::kXStride + r * FbankLayout<T>::kXRow;
    if constexpr (sizeof(T) == 4) {
        // two elements per 16-byte slot, logical slot i at physical slot i ^ (r & 7) (FbankLayout::xoff)
        const int sw = FRESH ? fresh_lane_value((r & 7) << 2) : (r & 7) << 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f4 v = *reinterpret_cast<const f4 *>(row + ((i << 2) ^ sw));
            own[2 * i] = {v.x, v.y};
            own[2 * i + 1] = {v.z, v.w};
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) own[i] = ldc(row + 2 * i);
    }
    fft16(own);
}

// phase 2b: Hermitian split with W_512, 4*power (or 2*magnitude) as f32 to LDS (the power rows are written
// over the exchange rows of the same wave).  part[i] = the partner lane's own[8 + i].
//   pair s: Z[k], k = r + 16s, with Z[256-k] = partner[15 - s]; for r == 0 the partner index is 16 - s
//   (mod 16: Z[256 - 16s]), i.e. own[0] for s = 0 and part[8 - s] for s = 1..7; s = 8 is Z[128] with itself.
//   POWER == false (FbankConfig::use_power off: magnitudes) is a compile-time variant: as a run-time select the compiler
//   evaluated the 17 IEEE square roots of every lane unconditionally (~170 instructions per unit, 7 % of the kernel).
//   FAST (the Whisper flavour only; its table holds (2 sin, 4 cos), build_whisper512_tables): the two powers straight from
//   |S|^2 + |D|^2 +- 2 Im(conj(S) D w) -- 12 operations per pair instead of 16, at the price of a difference of large numbers in the
//   weaker bin, which only a clamped output can afford (precise_phase2, whisper_wave_f64.hpp).
template <class T, bool POWER = true, bool FAST = false>
MS_DEV void fb_phase2_split(int fl, int r, bool active, const T *MS_RESTRICT tblob, const cpx<T> (&own)[16],
                            const cpx<T> (&part)[8], T *slice) {
    static_assert(POWER || !FAST, "the direct form computes powers");
    if (!active) return;
    const T *tw = tblob + FbankBlob::kTw2 + r * FbankBlob::kTw2Stride;
    float *p = reinterpret_cast<float *>(slice) + fl * FbankLayout<T>::kPStride;
    const bool lane0 = r == 0;
    auto pair = [&](int s, cpx<T> zk, cpx<T> zm) {
        if (FAST) {
            const T a = zk.re * zk.re + zk.im * zk.im, b = zm.re * zm.re + zm.im * zm.im;
            const T c = zk.re * zm.im + zm.re * zk.im;
            const cpx<T> t = ldc(tw + 2 * s);
            const T s2 = a + b, x = (a - b) * t.re + c * t.im;
            p[r + 16 * s] = static_cast<float>(T(2) * s2 + x);
            p[256 - r - 16 * s] = static_cast<float>(T(2) * s2 - x);
            return;
        }
        const cpx<T> S = {zk.re + zm.re, zk.im - zm.im};
        const cpx<T> D = {zk.re - zm.re, zk.im + zm.im};
        const cpx<T> wd = cmul(ldc(tw + 2 * s), D);
        const T ar = S.re + wd.im, ai = S.im - wd.re;
        const T br = S.re - wd.im, bi = S.im + wd.re;
        float pk = static_cast<float>(ar * ar + ai * ai), pm = static_cast<float>(br * br + bi * bi);   // 4*|X|^2
        if (!POWER) {                                                                             // 2*|X|
            pk = __builtin_sqrtf(pk);
            pm = __builtin_sqrtf(pm);
        }
        p[r + 16 * s] = pk;
        p[256 - r - 16 * s] = pm;
    };
    {
        const cpx<T> zm = {lane0 ? own[0].re : part[7].re, lane0 ? own[0].im : part[7].im};
        pair(0, own[0], zm);
    }
#pragma unroll
    for (int s = 1; s < 8; ++s) {
        const cpx<T> zm = {lane0 ? part[8 - s].re : part[7 - s].re, lane0 ? part[8 - s].im : part[7 - s].im};
        pair(s, own[s], zm);
    }
    // lane 0's ninth value is Z[128], its own partner: W_512^128 = -i makes X[128] = conj(Z[128]), so the split is |Z[128]|^2
    if (lane0) {
        float pk = static_cast<float>(T(4) * (own[8].re * own[8].re + own[8].im * own[8].im));
        if (!POWER) pk = __builtin_sqrtf(pk);
        p[128] = pk;
    }
}

// Compile-time slot lengths of the default filterbanks over 16-lane groups (any other bank: LensRuntime).  With them the
// bin loop unrolls and its LDS reads are issued together; the run-time loop pays one LDS round trip per bin (26-30 per
// unit), which two waves per SIMD cannot hide.
template <int... L>
struct LensFbStatic {
    static constexpr bool kStatic = true;
    static constexpr int kSlots = sizeof...(L);
    MS_HD static constexpr int len(int i) {
        constexpr int t[sizeof...(L)] = {L...};
        return t[i];
    }
    MS_HD static constexpr int woff(int i) {          // float offset from the mel section base
        constexpr int t[sizeof...(L)] = {L...};
        int s = 0;
        for (int k = 0; k < i; ++k) s += t[k];
        return FbankBlob::kMelW + 2 * kFbLanes * s;
    }
};
using LensKaldi80 = LensFbStatic<2, 2, 3, 5, 7, 9>;              // Kaldi mel scale, 80 bins, 20 Hz .. 8 kHz at 16 kHz (FbankConfig::default)
using LensKaldi40 = LensFbStatic<4, 9, 16>;                      // the same scale, 40 bins (the other common Kaldi fbank width)
using LensSlaney80 = LensFbStatic<2, 2, 3, 5, 8, 10>;            // NeMo: Slaney, 80 mels, 0 .. 8 kHz, bins 0..256
using LensSlaney80W = LensFbStatic<2, 2, 3, 5, 8, 9>;            // Whisper at n_fft 512: the same bank over bins < 256
using LensSlaney128 = LensFbStatic<1, 1, 1, 2, 2, 3, 4, 5, 7>;   // 128 mels, both

// phase 3: interval sums over 16-lane groups (lane j<15 owns interval j + 15*slot, j=15 is the ghost)
template <class T, int NSLOTS = kFbSlots, class Lens = LensRuntime>
MS_DEV void fb_phase3_sums(int fl, int j, bool active, const MelSlots &ms, const float *mel /* mel section base */,
                           const T *slice, const int (&st)[NSLOTS], float (&rise)[NSLOTS], float (&fprev)[NSLOTS]) {
    // every lane computes (see six_phase3_sums): a lane without a frame reads frame 0's row
    const float *p = reinterpret_cast<const float *>(slice) + (active ? fl : 0) * FbankLayout<T>::kPStride;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        float ar = 0.0f, af = 0.0f;
        if (Lens::kStatic) {
            if (i < Lens::kSlots) {
                const float *pp = p + st[i];
                const float *w = mel + Lens::woff(i < Lens::kSlots ? i : 0) + 2 * j;
#pragma unroll
                for (int r = 0; r < Lens::len(i < Lens::kSlots ? i : 0); ++r) {
                    const f2 wv = ld2_single(w + 2 * kFbLanes * r);      // one ds_read_b64 (see device_fft.hpp)
                    const float pv = pp[r];
                    if (r == 0) { ar = wv.x * pv; af = wv.y * pv; }
                    else { ar += wv.x * pv; af += wv.y * pv; }
                }
            }
        } else if (i < ms.n_slots) {
            const float *pp = p + st[i];
            const float *w = mel + ms.woff[i] + 2 * j;
            interval_bins_runtime<2 * kFbLanes>(pp, w, ms.len[i], ar, af);
        }
        rise[i] = ar;
        fprev[i] = af;
    }
}

MS_DEV float fast_ln(float x) { return fast_log2(x) * 0.69314718055994531f; }

// floor, ln, store (src/fbank.rs:207-221).  out_tile = &out[first frame of the tile][0].  vals (optional): the stored
// features of this lane, mel j + 15 i of frame fl (for the in-order column sums of the CMN, fbank512_clip_kernel).
template <int NSLOTS = kFbSlots>
MS_DEV void fb_phase3_store(int fl, int j, bool active, int n_mels, float floor_v, bool use_log,
                            const float (&rise)[NSLOTS], const float (&fnext)[NSLOTS], float *out_tile, float *vals = nullptr) {
    if (!active || j >= kFbOwn) return;
    float *o = out_tile + static_cast<long long>(fl) * n_mels + j;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const int m = j + kFbOwn * i;
        if (m < n_mels) {
            float e = rise[i] + fnext[i];
            e = __builtin_fmaxf(e, floor_v);
            const float v = use_log ? fast_ln(e) : e;
            o[kFbOwn * i] = v;
            if (vals) vals[i] = v;
        }
    }
}

// NeMo epilogue: ln(E + guard) (src/mel.rs:365-368), feature-major rows of `row_w` columns
// (src/mel.rs:366); columns past the valid frames are zero (the reference zero-initialises `features`).
template <int NSLOTS>
MS_DEV void nemo_phase3_store(int fl, int j, bool store, bool valid, int n_mels, float guard, const float (&rise)[NSLOTS],
                              const float (&fnext)[NSLOTS], float *out_col /* &out[0][first frame of the tile] */,
                              long long row_w) {
    if (!store || j >= kFbOwn) return;
    float *o = out_col + static_cast<long long>(j) * row_w + fl;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const int m = j + kFbOwn * i;
#ifdef MS_NEMO_NO_STORE      // timing ablation (tools/ab_build.sh): the value is computed and dropped -- what the feature-major stores cost
        if (m < n_mels) { float v = valid ? fast_ln((rise[i] + fnext[i]) + guard) : 0.0f; asm volatile("" :: "v"(v)); (void)o; }
#else
        if (m < n_mels) o[static_cast<long long>(kFbOwn * i) * row_w] = valid ? fast_ln((rise[i] + fnext[i]) + guard) : 0.0f;
#endif
    }
}

// Whisper epilogue for the 512 flavour.  log10(max(E, 1e-10)) (src/mel.rs:148-168); the frame maximum goes through
// 16 LDS words per frame (pmax, behind the power rows), then max(x, mx - 8), (x + 4) / 4 (src/mel.rs:645-654).
constexpr int kW512PmaxOff = kFbFPW * 259 + 4;       // float offset of the maxima inside the slice (16-byte aligned)
// The values are carried with a bias of +16 and compared as integers (whisper_six.hpp, six_phase3_finish).
template <int NSLOTS>
MS_DEV void w512_phase3_log(int fl, int j, bool active, int n_mels, const float (&rise)[NSLOTS], const float (&fnext)[NSLOTS],
                            float *slice_f, float (&vals)[NSLOTS]) {
    if (!active) return;
    int mx = 0;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const float e = rise[i] + fnext[i];
        const float v = __builtin_fmaxf(fast_log2(e) * 0.30102999566398120f + 16.0f, 6.0f);       // log10(max(e, 1e-10)) + 16
        vals[i] = v;
        if (j < kFbOwn && j + kFbOwn * i < n_mels) mx = wave_imax(mx, wave_bits(v));
    }
    reinterpret_cast<int *>(slice_f)[kW512PmaxOff + fl * kFbLanes + j] = mx;
}
// store: this lane's column exists in the output; valid: it is a real frame (otherwise a zero column of a padded
// layout).  row_w == 0: [frame][mel] rows of n_mels; row_w > 0: [mel][row_w] rows (interleave_frames, src/mel.rs:480-544).
// GUARD (MELSPEC_PRECISION_AUTO at n_fft = 512, round 6): true on the lanes that hold a band within kGuardBand decades of the clamp -- the
// bands an f32 FFT cannot vouch for at 1e-4 (wave_phase4 in whisper_wave.hpp has the calibration; the same test on the same biased values)
template <int NSLOTS, bool GUARD = false>
MS_DEV bool w512_phase4(int fl, int j, bool store, bool valid, int n_mels, const float *slice_f, const float (&vals)[NSLOTS],
                        float *out_tile, long long row_w) {
    if (!store || j >= kFbOwn) return false;
    int lo = 0;
    if (valid) {
        const int *pm = reinterpret_cast<const int *>(slice_f) + kW512PmaxOff + fl * kFbLanes;
        int mx = 0;
#pragma unroll
        for (int k = 0; k < kFbLanes; k += 4) {
            const WaveI4 a = *reinterpret_cast<const WaveI4 *>(pm + k);
            mx = wave_imax(mx, wave_imax(wave_imax(a.x, a.y), wave_imax(a.z, a.w)));
        }
        lo = wave_bits(wave_float(mx) - 8.0f);
    }
    float *o = row_w ? out_tile + static_cast<long long>(j) * row_w + fl : out_tile + static_cast<long long>(fl) * n_mels + j;
    const long long step = row_w ? kFbOwn * row_w : kFbOwn;
    int cmin = 0x7f000000;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const int m = j + kFbOwn * i;
        if (m < n_mels) {
            const int c = wave_imax(wave_bits(vals[i]), lo);
            o[i * step] = valid ? wave_float(c) * 0.25f - 3.0f : 0.0f;
            if (GUARD) cmin = wave_imin(cmin, c);
        }
    }
    return GUARD && valid && wave_float(cmin) < wave_float(lo) + kGuardBand;
}

}  // namespace melspec
