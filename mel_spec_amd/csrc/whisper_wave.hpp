// whisper_wave.hpp -- wave-autonomous form of the fused n_fft=400 log-mel pipeline.
//
// One 64-lane wavefront owns kFPW = 5 whole frames from PCM to mel rows (derivation of the FFT split in
// whisper_fast.hpp): lane = 11*frame + j in phases 1-2, 12*frame + j in phases 3-4.
// All exchanges go through the wave's private LDS slice, and because LDS operations of one
// wave execute in program order no workgroup barrier is needed anywhere in the tile loop --
// waves of a workgroup only share the read-only table blob.  The slice is reused in place:
//   [FFT exchange rows] -> [power rows | frame maxima]
// which is safe for the same reason (every lane's reads of a stage are issued before any
// lane's writes of the next stage).
//
// Reference steps: frame_windows src/stft.rs:147-169; FFT src/stft.rs:105-111; sparse mel +
// log10 src/mel.rs:148-168; per-frame normalisation src/mel.rs:645-654.
#pragma once
#include "whisper_fast.hpp"

namespace melspec {

constexpr int kFPW = 5;   // frames per wavefront (5 * 11 = 55 of 64 lanes)

struct WaveLayout {
    static constexpr int kXRow = 20;
    static constexpr int kXStride = 436;                 // == 20 (mod 32): conflict-free b64 row writes
    static constexpr int kPStride = 207;                 // power rows: 207 halves the write conflicts of 201 at equal read cost (tools/lds_sim2.py)
    static constexpr int kPmaxStride = 12;               // 11 maxima + 1 pad, 48 B rows (16-byte aligned)
    static constexpr int kPmaxOff = 1036;                // after the 5 power rows (5*207 = 1035)
    // Where exchange row k1 lives inside a frame's block of 21 rows.  Phase 1 writes whole rows (any order is
    // conflict-free); phase 2 reads rows j and 20-j (20 for j = 0) from 55 lanes at once, and with the rows in
    // natural order the 16-byte reads of lanes from different frames collide 11 cycles per pair of reads; this
    // order (hill-climbed on the bank model of tools/lds_sim.py, which reproduces SQ_LDS_BANK_CONFLICT) leaves 4.
    MS_HD static constexpr int row_pos(int k1) {
        constexpr int t[21] = {20, 1, 19, 9, 6, 8, 11, 3, 16, 5, 14, 7, 15, 17, 2, 12, 13, 0, 4, 18, 10};
        return t[k1];
    }
    // per-lane constants of phase 2 (computed once per kernel: j is fixed per lane)
    MS_HD static void row_offsets(int j, int &uoff, int &voff) {
        uoff = 0; voff = 0;
        for (int k = 0; k <= 10; ++k)            // a select chain, not an indexed load
            if (k == j) { uoff = row_pos(k) * kXRow; voff = row_pos(k == 0 ? 20 : 20 - k) * kXRow; }
    }
    static constexpr int slice_floats() { return kFPW * kXStride; }   // 2180
};

// Run-time slot lengths (any bank that is not one of the compile-time ones): the bins of a slot four at a time, so that the
// eight LDS reads of a group are in flight together; one read pair per iteration is one LDS round trip per bin.  The
// summation order is that of the plain loop.
template <int WSTRIDE>
MS_DEV void interval_bins_runtime(const float *pp, const float *w, int len, float &ar, float &af) {
    int r = 0;
    for (; r + 4 <= len; r += 4) {
        f2 wv[4];
        float pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wv[q] = ld2_single(w + WSTRIDE * (r + q));
            pv[q] = pp[r + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ar += wv[q].x * pv[q];
            af += wv[q].y * pv[q];
        }
    }
    for (; r < len; ++r) {
        const f2 wv = ld2_single(w + WSTRIDE * r);
        const float pv = pp[r];
        ar += wv.x * pv;
        af += wv.y * pv;
    }
}

// Slot lengths are compile-time for the two Whisper filterbanks (16 kHz, 80 / 128 mels, below); any
// other (sr, n_mels) uses the runtime lengths in MelSlots.
struct LensRuntime {
    static constexpr bool kStatic = false;
    static constexpr int kSlots = kMaxSlots;
    static constexpr int kMels = 0;
    MS_HD static int len(int) { return 0; }
    MS_HD static int woff(int) { return 0; }
};
// Interval scheme (build_interval_mel): padded interval lengths per slot, weights are float pairs
// over 12 lanes, so a slot of length L occupies 24*L floats.
template <int MELS, int... L>
struct LensIntervalStatic {
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
        return FastBlob::kMelW + 24 * s;
    }
};
using LensI80 = LensIntervalStatic<80, 1, 1, 1, 2, 3, 4, 7, 7>;
using LensI128 = LensIntervalStatic<128, 1, 1, 1, 1, 1, 1, 2, 2, 3, 3, 4, 5>;

// 8-byte load from a pointer that is only 4-byte aligned (clip offsets are arbitrary).
MS_DEV f2 load2_unaligned(const float *p) {
#if defined(__HIPCC__)
    typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));
    const v2u v = *reinterpret_cast<const v2u *>(p);
    return f2{v.x, v.y};
#else
    return f2{p[0], p[1]};
#endif
}

// ---- phase 1 -----------------------------------------------------------------------------
// The frame's samples come straight from global memory (L1/L2 absorb the 2.5x frame overlap).
MS_DEV void wave_phase1(int fl, int t, bool active, int hop, const float *blob, const float *gsrc /* tile's first sample */,
                        float *slice) {
    if (!active) return;
    const float *w = blob + FastBlob::kWin + t * FastBlob::kWinStride;
    const float *s = gsrc + fl * hop + 2 * t;
    cf x[20];
    // all twenty loads first, then a scheduling barrier (six_phase1, whisper_six.hpp, says why)
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
    const float *tw = blob + FastBlob::kTw1 + t * FastBlob::kTw1Stride;
    float *xo = slice + fl * WaveLayout::kXStride + 2 * t;
    {
        const f2 m = *reinterpret_cast<const f2 *>(blob + FastBlob::kMod + 2 * t);
        const cf y = cmul(x[0], cf{m.x, m.y});
        *reinterpret_cast<f2 *>(xo + WaveLayout::row_pos(20) * WaveLayout::kXRow) = f2{y.re, y.im};
        *reinterpret_cast<f2 *>(xo + WaveLayout::row_pos(0) * WaveLayout::kXRow) = f2{x[0].re, x[0].im};
    }
#pragma unroll
    for (int k1 = 1; k1 < 20; ++k1) {
        const f2 wv = *reinterpret_cast<const f2 *>(tw + 2 * k1);
        const cf y = cmul(x[k1], cf{wv.x, wv.y});
        *reinterpret_cast<f2 *>(xo + WaveLayout::row_pos(k1) * WaveLayout::kXRow) = f2{y.re, y.im};
    }
}

// ---- phase 2: reads the exchange rows, writes the power row over the same slice ----------
// Stores 4*|X|^2 (the interval mel weights carry the 1/4).
MS_DEV void wave_phase2(int fl, int j, bool active, const float *blob, float *slice, int uoff, int voff) {
    if (!active) return;
    // uoff / voff: float offsets of rows j and 20-j (20 for j = 0) inside the frame's block, WaveLayout::row_offsets()
    const float *ua = slice + fl * WaveLayout::kXStride + uoff;
    const float *va = slice + fl * WaveLayout::kXStride + voff;
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
    float *p = slice + fl * WaveLayout::kPStride;
#pragma unroll
    for (int q = 0; q < 10; q += 2) {
        const f4 w2 = *reinterpret_cast<const f4 *>(tw + 2 * q);
        float pk[2], pm[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int qq = q + h;
            const cf zk = u[qq], zm = v[9 - qq];
            const cf S = {zk.re + zm.re, zk.im - zm.im};
            const cf D = {zk.re - zm.re, zk.im + zm.im};
            const cf W = h == 0 ? cf{w2.x, w2.y} : cf{w2.z, w2.w};
            const cf wd = cmul(W, D);
            const float ar = S.re + wd.im, ai = S.im - wd.re;
            const float br = S.re - wd.im, bi = S.im + wd.re;
            pk[h] = ar * ar + ai * ai;
            pm[h] = br * br + bi * bi;
        }
        // the two stores of a side next to each other: one ds_write2_b32 each (see six_phase2)
        p[j + 20 * q] = pk[0];
        p[j + 20 * q + 20] = pk[1];
        p[200 - j - 20 * q] = pm[0];
        p[180 - j - 20 * q] = pm[1];
    }
}

// ---- phase 3, interval scheme: lane (frame, j12) with j12 in [0,12) owns interval j12 + 11*slot ----
// sums: rise[i] = sum_{k in I} w_rise*P[k] (mel i's rising part), fprev[i] = sum w_fall*P[k] (the
// falling part of mel i-1).  The kernel then shifts fprev down one lane (DPP) so that lane j12 holds
// the falling part of its own mel, and calls the finish step.
template <int NSLOTS, class Lens>
MS_DEV void wave_phase3i_sums(int fl, int j12, bool active, const MelSlots &ms, const float *blob, const float *slice,
                              const int (&st)[NSLOTS], float (&rise)[NSLOTS], float (&fprev)[NSLOTS]) {
    // every lane computes (see six_phase3_sums): a lane without a frame reads frame 0's row
    const float *p = slice + (active ? fl : 0) * WaveLayout::kPStride;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        float ar = 0.0f, af = 0.0f;
        if (Lens::kStatic) {
            if (i < Lens::kSlots) {
                const float *pp = p + st[i];
                const float *w = blob + Lens::woff(i < Lens::kSlots ? i : 0) + 2 * j12;
#pragma unroll
                for (int r = 0; r < Lens::len(i < Lens::kSlots ? i : 0); ++r) {
                    const f2 wv = ld2_single(w + 24 * r);
                    const float pv = pp[r];
                    if (r == 0) { ar = wv.x * pv; af = wv.y * pv; }
                    else { ar += wv.x * pv; af += wv.y * pv; }
                }
            }
        } else if (i < ms.n_slots) {
            const float *pp = p + st[i];
            const float *w = blob + ms.woff[i] + 2 * j12;
            interval_bins_runtime<24>(pp, w, ms.len[i], ar, af);
        }
        rise[i] = ar;
        fprev[i] = af;
    }
}

// The log-mel values are carried with a bias of +16 (whisper_six.hpp, six_phase3_finish: positive floats order like their bit
// patterns, so every maximum / minimum of phases 3-4 is an integer one and needs no canonicalising v_max(x, x)).
MS_DEV int wave_bits(float v) { return __builtin_bit_cast(int, v); }
MS_DEV float wave_float(int v) { return __builtin_bit_cast(float, v); }
MS_DEV int wave_imax(int a, int b) { return a > b ? a : b; }
MS_DEV int wave_imin(int a, int b) { return a < b ? a : b; }
template <int NSLOTS>
MS_DEV void wave_phase3i_finish(int fl, int j12, bool active, int n_mels, const float (&rise)[NSLOTS],
                                const float (&fnext)[NSLOTS] /* fprev of lane+1 */, float *slice, float (&vals)[NSLOTS]) {
    if (!active) return;
    int mx = 0;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const float e = rise[i] + fnext[i];
        const float v = __builtin_fmaxf(fast_log2(e) * 0.30102999566398120f + 16.0f, 6.0f);       // log10(max(e, 1e-10)) + 16
        vals[i] = v;
        if (j12 < kMelJobs && j12 + kMelJobs * i < n_mels) mx = wave_imax(mx, wave_bits(v));
    }
    reinterpret_cast<int *>(slice)[WaveLayout::kPmaxOff + fl * WaveLayout::kPmaxStride + j12] = mx;
}

// ---- phase 4: frame max, clamp, scale, store ------------------------------------------------
// store: this lane's frame column exists in the output; valid: it is a real frame (otherwise a zero
// column of a padded layout).  row_w == 0: [frame][mel] rows; row_w > 0: [mel][row_w] rows.
//
// Precision guard (GUARD): the per-frame clamp at max - 8 keeps mel bands up to 80 dB under the strongest one, and an
// f32 FFT leaves the strongest line's rounding noise in every bin, so a band within kGuardBand decades of the clamp can be
// off by more than 1e-4 after log10 (measured with tools/flag_calib.py on tones / chirps / speech over noise floors and five
// filterbanks: bands >= 2 decades above the clamp stay under 3.9e-5, bands at the clamp reach 4.9e-4).  The function
// returns true on lanes that hold such a band; the kernel then notes the unit and recomputes the frame in f64 after its run
// (whisper_fix64.hpp).  Bands exactly at the clamp count too: whether a band is clamped is decided by the same noisy
// value.  A silent frame (every band at the 1e-10 floor, nothing within 8 decades below) is never queued.
constexpr float kGuardBand = 2.0f;

struct alignas(16) WaveI4 { int x, y, z, w; };
MS_DEV float wave_out(int c) { return wave_float(c) * 0.25f - 3.0f; }          // (x + 4) / 4 of the biased value
// KEYS: see six_phase4 (whisper_six.hpp)
template <int NSLOTS, bool LAYOUT = true, bool GUARD = false, bool KEYS = false>
MS_DEV bool wave_phase4(int fl, int j, bool store, bool valid, int n_mels, const float *slice, const float (&vals)[NSLOTS],
                        float *out_tile, long long row_w, int *kmin = nullptr, int *kmax = nullptr) {
    if (!LAYOUT) { valid = true; row_w = 0; }     // plain output: every stored column is a real frame
    if (!store || j >= kMelJobs) return false;
    int lo = 0;                                   // bits of (frame maximum - 8), biased by 16
    if (valid) {
        const int *pm = reinterpret_cast<const int *>(slice) + WaveLayout::kPmaxOff + fl * WaveLayout::kPmaxStride;
        const WaveI4 a = *reinterpret_cast<const WaveI4 *>(pm), b = *reinterpret_cast<const WaveI4 *>(pm + 4), c = *reinterpret_cast<const WaveI4 *>(pm + 8);
        const int m0 = wave_imax(wave_imax(a.x, a.y), wave_imax(a.z, a.w));
        const int m1 = wave_imax(wave_imax(b.x, b.y), wave_imax(b.z, b.w));
        const int m2 = wave_imax(wave_imax(c.x, c.y), c.z);           // 11 maxima (the twelfth word is the ghost lane's)
        lo = wave_bits(wave_float(wave_imax(wave_imax(m0, m1), m2)) - 8.0f);
    }
    float *o = row_w ? out_tile + static_cast<long long>(j) * row_w + fl : out_tile + static_cast<long long>(fl) * n_mels + j;
    const long long step = row_w ? kMelJobs * row_w : kMelJobs;
    int cmin = 0x7f000000, cmax = 0;
#pragma unroll
    for (int i = 0; i < NSLOTS; ++i) {
        const int m = j + kMelJobs * i;
        if (m < n_mels) {
            const int c = valid ? wave_imax(wave_bits(vals[i]), lo) : 0x41400000;       // a zero column is the biased value 12
            o[i * step] = wave_out(c);
            if (GUARD || KEYS) cmin = wave_imin(cmin, c);
            if (KEYS) cmax = wave_imax(cmax, c);
        }
    }
    if (KEYS) { *kmin = cmin; *kmax = cmax; }
    return GUARD && valid && wave_float(cmin) < wave_float(lo) + kGuardBand;
}

}  // namespace melspec
