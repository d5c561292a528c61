// whisper_fast.hpp -- the fused n_fft=400 log-mel frame pipeline, written as per-thread
// "phase" functions that communicate only through LDS.  The HIP kernel
// (melspec_kernels.hip) calls the phases with __syncthreads() between them; tests/emu runs
// the very same functions on the host, one phase at a time over all thread ids.
//
// Pipeline per frame (reference: Spectrogram::compute_mel_spectrogram_cpu, src/stft.rs:119-138):
//   frame_windows  (src/stft.rs:147-169)  -> phase1: LDS PCM tile * Hann table
//   forward FFT    (src/stft.rs:105-111)  -> phase1+2: real-400 as complex-200 = 20 x 10
//   |X|^2, bins<200 (src/mel.rs:155-163)  -> phase2: Hermitian split, power to LDS
//   sparse mel+log10 (src/mel.rs:148-168) -> phase3: banded dot from LDS, log2*log10(2)
//   per-frame norm (src/mel.rs:645-654)   -> phase3/4: max via LDS, clamp, (x+4)/4
//
// FFT factorisation.  z[n] = x[2n] + i*x[2n+1], n in [0,200).  With n = 10*n1 + n2 and
// k = k1 + 20*k2:  Z[k] = sum_n2 W_10^{n2*k2} * ( W_200^{n2*k1} * sum_n1 W_20^{n1*k1} z[10*n1+n2] ).
//   phase1: thread (frame, t=n2) does the 20-point DFT over n1, multiplies by W_200^{t*k1},
//           writes row k1 of the exchange buffer.
//   phase2: thread (frame, j), j in [0,11), owns residues a=j and b=(20-j)%20 (k = a+20q and
//           its mirror 200-k = b+20(9-q)), does two 10-point DFTs and the real-FFT split
//             X[k] = (S - i*W_400^k*D)/2,  S = Z[k]+conj(Z[200-k]),  D = Z[k]-conj(Z[200-k]).
//           j=0 pairs residue 0 with itself shifted by one (row 20 holds row 0 modulated by
//           W_10^{n2}, so its DFT is Z[20(q+1)]); j=10 pairs residue 10 with itself.
#pragma once
#include "device_fft.hpp"

namespace melspec {

constexpr int kMelJobs = 11;        // threads per frame in phases 2-4
constexpr int kFftJobs = 10;        // threads per frame in phase 1
constexpr int kMaxSlots = 12;       // mel slots per thread of the Whisper kernels (n_mels <= 131)
constexpr int kSlotCap = 17;        // capacity of MelSlots (the NeMo frontend uses up to 17 slots of 8)

// LDS layout of the constant table blob (float offsets).  Built by build_fast_tables().
struct FastBlob {
    static constexpr int kWin = 0;                       // [400] Hann
    static constexpr int kTw1Stride = 44;                // 20 complex + 4 pad (bank spread)
    static constexpr int kTw1 = 400;                     // [10][44] W_200^{t*k1}
    static constexpr int kMod = kTw1 + 10 * kTw1Stride;  // [10] complex W_10^{n2}
    static constexpr int kTw2 = kMod + 20;               // [11][10] complex W_400^{j+20q}
    static constexpr int kMelStart = kTw2 + kMelJobs * 20;   // [kMaxSlots*12] int bit patterns
    // banded scheme: padded weights [slot][r][11]; interval scheme: (rise, fall) pairs [slot][r][12][2]
    static constexpr int kMelW = kMelStart + kMaxSlots * 12;
};

// Per-launch uniform parameters of the mel slots (scalar registers on the device).
struct MelSlots {
    int n_slots;               // banded: ceil(n_mels / 11); interval: ceil((n_mels + 1) / 11)
    int len[kSlotCap];         // padded span length of slot i
    int woff[kSlotCap];        // float offset of slot i's weights inside the blob
};

template <int FPB>
struct FastLayout {
    static constexpr int kP1Threads = FPB * kFftJobs;
    static constexpr int kP2Threads = FPB * kMelJobs;
    static constexpr int kXRow = 20;                 // floats per exchange row (10 complex)
    static constexpr int kXStride = 21 * kXRow + 16; // 436: frame stride == 20 mod 32 (conflict-free b64 writes)
    static constexpr int kPStride = 201;             // power row: bins 0..200, odd stride
    // region A: PCM tile (phase 0/1) aliased with the power rows (phase 2/3)
    static constexpr int region_a(int hop) {
        const int pcm = (FPB - 1) * hop + 400, pw = FPB * kPStride;
        return ((pcm > pw ? pcm : pw) + 3) & ~3;
    }
    static constexpr int region_b() { return FPB * kXStride; }
    static constexpr int region_max() { return FPB * kMelJobs; }
};

// ---- phase 1: window, 20-point DFTs, W_200 twiddle, exchange write -------------------
template <int FPB>
MS_DEV void fast_phase1(int tid, int n_valid, int hop, const float *blob, const float *pcm, float *xchg) {
    using L = FastLayout<FPB>;
    if (tid >= L::kP1Threads) return;
    const int fl = tid / kFftJobs, t = tid - fl * kFftJobs;
    if (fl >= n_valid) return;
    const float *s = pcm + fl * hop + 2 * t;
    const float *w = blob + FastBlob::kWin + 2 * t;
    cf x[20];
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) {
        const f2 sv = *reinterpret_cast<const f2 *>(s + 20 * n1);
        const f2 wv = *reinterpret_cast<const f2 *>(w + 20 * n1);
        x[n1] = {sv.x * wv.x, sv.y * wv.y};
    }
    fft20(x);
    const float *tw = blob + FastBlob::kTw1 + t * FastBlob::kTw1Stride;
    float *xo = xchg + fl * L::kXStride + 2 * t;
    {
        const f2 m = *reinterpret_cast<const f2 *>(blob + FastBlob::kMod + 2 * t);
        const cf y = cmul(x[0], cf{m.x, m.y});
        *reinterpret_cast<f2 *>(xo + 20 * L::kXRow) = f2{y.re, y.im};
        *reinterpret_cast<f2 *>(xo) = f2{x[0].re, x[0].im};
    }
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) {
        const f2 wv = *reinterpret_cast<const f2 *>(tw + 2 * k1);
        const cf y = cmul(x[k1], cf{wv.x, wv.y});
        *reinterpret_cast<f2 *>(xo + k1 * L::kXRow) = f2{y.re, y.im};
    }
}

// ---- phase 2: 10-point DFTs, Hermitian split, power spectrum to LDS ------------------
template <int FPB>
MS_DEV void fast_phase2(int tid, int n_valid, const float *blob, const float *xchg, float *pw) {
    using L = FastLayout<FPB>;
    if (tid >= L::kP2Threads) return;
    const int fl = tid / kMelJobs, j = tid - fl * kMelJobs;
    if (fl >= n_valid) return;
    const int brow = (j == 0) ? 20 : 20 - j;
    const float *ua = xchg + fl * L::kXStride + j * L::kXRow;
    const float *va = xchg + fl * L::kXStride + brow * L::kXRow;
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
    const float *tw = blob + FastBlob::kTw2 + j * 20;
    float *p = pw + fl * L::kPStride;
#pragma unroll
    for (int q = 0; q < 10; q += 2) {
        const f4 w2 = *reinterpret_cast<const f4 *>(tw + 2 * q);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int qq = q + h;
            const cf zk = u[qq], zm = v[9 - qq];
            const cf S = {zk.re + zm.re, zk.im - zm.im};
            const cf D = {zk.re - zm.re, zk.im + zm.im};
            const cf W = h == 0 ? cf{w2.x, w2.y} : cf{w2.z, w2.w};
            const cf wd = cmul(W, D);
            const float ar = S.re + wd.im, ai = S.im - wd.re;   // 2*X[k]
            const float br = S.re - wd.im, bi = S.im + wd.re;   // 2*conj-side X[200-k]
            p[j + 20 * qq] = 0.25f * (ar * ar + ai * ai);
            p[200 - j - 20 * qq] = 0.25f * (br * br + bi * bi);
        }
    }
}

MS_DEV float fast_log2(float x) {
#if defined(__HIPCC__)
    return __builtin_amdgcn_logf(x);   // v_log_f32, 1 ulp; input is >= 1e-10 (normal)
#else
    return __builtin_log2f(x);
#endif
}

// ---- phase 3: banded mel projection, log10, per-thread max ---------------------------
// vals[i] holds log10(max(E,1e-10)) of mel m = j + 11*i.  Returns nothing; writes the
// thread's max over its valid mels to pmax[fl*11 + j].
template <int FPB, int NSLOTS>
MS_DEV void fast_phase3(int tid, int n_valid, int n_mels, const MelSlots &ms, const float *blob,
                        const float *pw, float *pmax, float (&vals)[NSLOTS]) {
    using L = FastLayout<FPB>;
    if (tid >= L::kP2Threads) return;
    const int fl = tid / kMelJobs, j = tid - fl * kMelJobs;
    if (fl >= n_valid) return;
    const float *p = pw + fl * L::kPStride;
    const int *starts = reinterpret_cast<const int *>(blob + FastBlob::kMelStart);
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        float acc = 0.0f;
        if (i < ms.n_slots) {
            const int st = starts[i * kMelJobs + j];
            const float *wrow = blob + ms.woff[i] + j;
            const int len = ms.len[i];
            for (int r = 0; r < len; ++r) acc += wrow[r * kMelJobs] * p[st + r];
        }
        // log10(max(E, 1e-10)); the floored case is exactly -10 like the reference's f64 log10(1e-10)
        const float v = acc > 1e-10f ? fast_log2(acc) * 0.30102999566398120f : -10.0f;
        vals[i] = v;
        if (j + kMelJobs * i < n_mels) mx = __builtin_fmaxf(mx, v);
    }
    pmax[fl * kMelJobs + j] = mx;
}

// ---- phase 4: frame max, clamp, scale, store -----------------------------------------
template <int FPB, int NSLOTS>
MS_DEV void fast_phase4(int tid, int n_valid, int n_mels, const float *pmax, const float (&vals)[NSLOTS],
                        float *out_tile /* &out[first frame of the tile][0] */) {
    using L = FastLayout<FPB>;
    if (tid >= L::kP2Threads) return;
    const int fl = tid / kMelJobs, j = tid - fl * kMelJobs;
    if (fl >= n_valid) return;
    float mx = pmax[fl * kMelJobs];
#pragma unroll
    for (int i = 1; i < kMelJobs; ++i) {
        const float o = pmax[fl * kMelJobs + i];
        mx = mx > o ? mx : o;
    }
    const float lo = mx - 8.0f;
    float *o = out_tile + static_cast<long long>(fl) * n_mels + j;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const int m = j + kMelJobs * i;
        if (m < n_mels) {
            const float v = __builtin_fmaxf(vals[i], lo);
            o[kMelJobs * i] = (v + 4.0f) * 0.25f;
        }
    }
}

}  // namespace melspec
