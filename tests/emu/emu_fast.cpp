// emu_fast.cpp -- TEST-ONLY host emulation of the fused kernels.
//
// Runs the exact per-lane phase functions of mel_spec_amd/csrc/whisper_wave.hpp / whisper_six.hpp / fbank_wave.hpp on the
// host (one phase at a time over all thread ids == a barrier between phases), with a float
// array standing in for LDS.  It checks the kernel's index algebra and f32 error budget
// against the f64 oracle without a GPU.  Never linked into the product library.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../mel_spec_amd/csrc/fast_tables.hpp"
#include "../../mel_spec_amd/csrc/whisper_wave.hpp"
#include "../../mel_spec_amd/csrc/whisper_fix64.hpp"
#include "../../mel_spec_amd/csrc/fbank_tables.hpp"
#include "../../mel_spec_amd/csrc/tga_quant.hpp"
#include "../../mel_spec_amd/csrc/vad_columns.hpp"
#include "../../mel_spec_amd/csrc/pow2_wave.hpp"

using namespace melspec;

// Wave-autonomous kernel (whisper_wave.hpp): one 64-lane wave per 5-frame unit, one phase at a
// time over all lanes, the slice poisoned where the kernel promises not to read.
// flags (optional): one byte per frame, 1 where the precision guard of phase 4 fires.
template <int NSLOTS, class Lens>
static long long run_wave(const float *pcm, long long n, int hop, int n_mels, double sr, float *out, uint8_t *flags) {
    FastTables T;
    if (!build_fast_tables(sr, n_mels, T, true)) return -1;
    if (!T.interval) return -3;
    if (n < 400) return 0;
    const long long frames = (n - 400) / hop + 1;
    std::vector<float> slice(WaveLayout::slice_floats());
    std::vector<float> vals(static_cast<size_t>(64) * NSLOTS);
    for (long long f0 = 0; f0 < frames; f0 += kFPW) {
        const int nv = static_cast<int>(std::min<long long>(kFPW, frames - f0));
        std::fill(slice.begin(), slice.end(), 1.0e30f);
        const float *src = pcm + f0 * hop;
        // phase 1 reads every input before writing any exchange row: emulate with a snapshot
        std::vector<float> snap(slice);
        std::vector<float> next(slice);
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
            const bool act = lane < kFPW * kMelJobs && fl < nv;
            std::vector<float> tmp(snap);
            wave_phase1(fl, j, act && j < kFftJobs, hop, T.blob.data(), src, tmp.data());
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next; snap = slice;
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
            const bool act = lane < kFPW * kMelJobs && fl < nv;
            std::vector<float> tmp(snap);
            int uoff, voff;
            WaveLayout::row_offsets(j, uoff, voff);
            wave_phase2(fl, j, act, T.blob.data(), tmp.data(), uoff, voff);
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next; snap = slice;
        {
            std::vector<float> rise(64 * NSLOTS), fprev(64 * NSLOTS + NSLOTS, 0.0f);
            const int *starts = reinterpret_cast<const int *>(T.blob.data() + FastBlob::kMelStart);
            for (int lane = 0; lane < 64; ++lane) {
                const int fl = lane / 12, j = lane - fl * 12;
                const bool act = lane < kFPW * 12 && fl < nv;
                int st[NSLOTS];
                for (int i = 0; i < NSLOTS; ++i) st[i] = lane < kFPW * 12 ? starts[i * 12 + j] : 0;
                wave_phase3i_sums<NSLOTS, Lens>(fl, j, act, T.slots, T.blob.data(), snap.data(), st,
                                                *reinterpret_cast<float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                                *reinterpret_cast<float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane) * NSLOTS]));
            }
            for (int lane = 0; lane < 64; ++lane) {     // wave_shl:1 -> lane l sees lane l+1
                const int fl = lane / 12, j = lane - fl * 12;
                const bool act = lane < kFPW * 12 && fl < nv;
                std::vector<float> tmp(snap);
                wave_phase3i_finish<NSLOTS>(fl, j, act, n_mels,
                                            *reinterpret_cast<const float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                            *reinterpret_cast<const float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane + 1) * NSLOTS]),
                                            tmp.data(), *reinterpret_cast<float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]));
                for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
            }
        }
        slice = next;
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / 12, j = lane - fl * 12;
            const bool act = lane < kFPW * 12 && fl < nv;
            const bool g = wave_phase4<NSLOTS, false, true>(fl, j, act, act, n_mels, slice.data(),
                                *reinterpret_cast<const float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]),
                                out + f0 * n_mels, 0);
            if (g && flags) flags[f0 + fl] = 1;
        }
    }
    return frames;
}

template <class Lens>
static bool lens_ok(const MelSlots &ms) {
    if (ms.n_slots != Lens::kSlots) return false;
    for (int i = 0; i < Lens::kSlots; ++i)
        if (ms.len[i] != Lens::len(i) || ms.woff[i] != Lens::woff(i)) return false;
    return true;
}

// mode: 0 run-time slot lengths, 1 compile-time lengths (only the 16 kHz 80 / 128 mel banks)
extern "C" long long emu_whisper_wave_guard(const float *pcm, long long n, int hop, int n_mels, double sr, int mode, float *out, uint8_t *flags) {
    FastTables T;
    if (!build_fast_tables(sr, n_mels, T, true)) return -1;
    if (!T.interval) return -3;
    if (mode == 1) {
        if (lens_ok<LensI80>(T.slots)) return run_wave<8, LensI80>(pcm, n, hop, n_mels, sr, out, flags);
        if (lens_ok<LensI128>(T.slots)) return run_wave<12, LensI128>(pcm, n, hop, n_mels, sr, out, flags);
        return -2;
    }
    if (T.slots.n_slots <= 8) return run_wave<8, LensRuntime>(pcm, n, hop, n_mels, sr, out, flags);
    return run_wave<12, LensRuntime>(pcm, n, hop, n_mels, sr, out, flags);
}
extern "C" long long emu_whisper_wave(const float *pcm, long long n, int hop, int n_mels, double sr, int mode, float *out) {
    return emu_whisper_wave_guard(pcm, n, hop, n_mels, sr, mode, out, nullptr);
}

// Fused fbank kernel (fbank_wave.hpp), default geometry, no CMN.  out = [frames][n_mels].
template <class T>
static long long run_fbank(const float *pcm, long long n, int shift, int n_mels, double sr, double low, double high,
                           double preemph, float floor_v, int use_log, int use_power, float *out) {
    using L = FbankLayout<T>;
    FbankFastTables F;
    if (!build_fbank_fast_tables<T>(sr, n_mels, low, high, use_power != 0, F)) return -1;
    const T *tblob = reinterpret_cast<const T *>(F.blob.data());
    const float *mel = reinterpret_cast<const float *>(F.blob.data() + F.mel_off_words);
    if (n < 400) return 0;
    const long long frames = (n - 400) / shift + 1;
    std::vector<T> slice(L::slice_elems());
    const int *starts = reinterpret_cast<const int *>(mel + FbankBlob::kMelStart);
    for (long long f0 = 0; f0 < frames; f0 += kFbFPW) {
        const int nv = static_cast<int>(std::min<long long>(kFbFPW, frames - f0));
        std::fill(slice.begin(), slice.end(), T(1.0e30));
        auto lane_info = [&](int lane, int &fl, int &j, bool &act) {
            fl = lane / kFbLanes; j = lane - fl * kFbLanes; act = lane < kFbFPW * kFbLanes && fl < nv;
        };
        std::vector<T> mean(64, T(0));
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            slice[L::kSumOff + lane] = act ? fb_partial_sum<T>(pcm + (f0 + fl) * shift, j) : T(0);
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            if (!act) continue;
            const T *ps = slice.data() + L::kSumOff + fl * kFbLanes;
            const T a = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]));
            const T b = ((ps[8] + ps[9]) + (ps[10] + ps[11])) + ((ps[12] + ps[13]) + (ps[14] + ps[15]));
            mean[lane] = (a + b) / T(400);
        }
        std::vector<T> snap(slice), next(slice);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            std::vector<T> tmp(snap);
            fb_phase1<T>(fl, j, act, pcm + (f0 + (act ? fl : 0)) * shift, f0 + fl == 0, mean[lane], static_cast<T>(preemph), tblob, tmp.data());
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next; snap = slice;
        std::vector<cpx<T>> own(64 * 16);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            fb_phase2_dft<T>(fl, j, act, snap.data(), *reinterpret_cast<cpx<T>(*)[16]>(&own[static_cast<size_t>(lane) * 16]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            std::vector<T> tmp(snap);
            cpx<T> part[8];
            const int src = (lane & ~15) | ((16 - (lane & 15)) & 15);          // what partner16() fetches on the device
            for (int i = 0; i < 8; ++i) part[i] = own[static_cast<size_t>(src) * 16 + 8 + i];
            const auto &mine = *reinterpret_cast<const cpx<T>(*)[16]>(&own[static_cast<size_t>(lane) * 16]);
            if (use_power != 0) fb_phase2_split<T, true>(fl, j, act, tblob, mine, part, tmp.data());
            else fb_phase2_split<T, false>(fl, j, act, tblob, mine, part, tmp.data());
            // power rows are f32 written into the T-typed slice: compare bytes
            const uint32_t *a = reinterpret_cast<const uint32_t *>(tmp.data()), *b0 = reinterpret_cast<const uint32_t *>(snap.data());
            uint32_t *d = reinterpret_cast<uint32_t *>(next.data());
            for (size_t i = 0; i < tmp.size() * sizeof(T) / 4; ++i) if (a[i] != b0[i]) d[i] = a[i];
        }
        slice = next;
        std::vector<float> rise(64 * kFbSlots), fprev(65 * kFbSlots, 0.0f);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            int st[kFbSlots];
            for (int i = 0; i < kFbSlots; ++i) st[i] = lane < kFbFPW * kFbLanes ? starts[i * kFbLanes + j] : 0;
            fb_phase3_sums<T>(fl, j, act, F.slots, mel, slice.data(), st,
                              *reinterpret_cast<float(*)[kFbSlots]>(&rise[static_cast<size_t>(lane) * kFbSlots]),
                              *reinterpret_cast<float(*)[kFbSlots]>(&fprev[static_cast<size_t>(lane) * kFbSlots]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            fb_phase3_store(fl, j, act, n_mels, floor_v, use_log != 0,
                            *reinterpret_cast<const float(*)[kFbSlots]>(&rise[static_cast<size_t>(lane) * kFbSlots]),
                            *reinterpret_cast<const float(*)[kFbSlots]>(&fprev[static_cast<size_t>(lane + 1) * kFbSlots]),
                            out + f0 * n_mels);
        }
    }
    return frames;
}

extern "C" long long emu_fbank_wave(const float *pcm, long long n, int shift, int n_mels, double sr, double low, double high,
                                    double preemph, float floor_v, int use_log, int use_power, int f64, float *out) {
    return f64 ? run_fbank<double>(pcm, n, shift, n_mels, sr, low, high, preemph, floor_v, use_log, use_power, out)
               : run_fbank<float>(pcm, n, shift, n_mels, sr, low, high, preemph, floor_v, use_log, use_power, out);
}

// NeMo flavour of the fused 512 kernel (nemo_phase1 / nemo_phase3_store), f64 arithmetic.  out = [n_mels][cols].
template <class T>
static long long run_blm(const float *pcm, long long n, int hop, int n_mels, int sample_rate, double f_min, double f_max,
                         int htk, int norm, float preemph, int center, float guard, long long cols, float *out) {
    using L = FbankLayout<T>;
    constexpr int NS = kBlmSlots;
    FbankFastTables F;
    if (!build_blm_fast_tables<T>(sample_rate, n_mels, f_min, f_max > 0 ? f_max : sample_rate / 2.0, htk != 0, norm != 0, F)) return -1;
    const T *tblob = reinterpret_cast<const T *>(F.blob.data());
    const float *mel = reinterpret_cast<const float *>(F.blob.data() + F.mel_off_words);
    const long long valid = n == 0 ? 0 : (center ? n / hop + 1 : (n < 512 ? 0 : (n - 512) / hop + 1));
    const int org0 = center ? -200 : 56;
    std::vector<T> slice(L::slice_elems());
    const int *starts = reinterpret_cast<const int *>(mel + FbankBlob::kMelStart);
    for (long long f0 = 0; f0 < cols; f0 += kFbFPW) {
        const int nv = static_cast<int>(std::max<long long>(0, std::min<long long>(kFbFPW, valid - f0)));
        const int ns = static_cast<int>(std::min<long long>(kFbFPW, cols - f0));
        std::fill(slice.begin(), slice.end(), T(1.0e30));
        auto lane_info = [&](int lane, int &fl, int &j, bool &act) {
            fl = lane / kFbLanes; j = lane - fl * kFbLanes; act = lane < kFbFPW * kFbLanes && fl < nv;
        };
        std::vector<T> snap(slice), next(slice);
        bool all_inside = true;     // the kernel's wave-uniform ballot
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            const long long org = (f0 + fl) * hop + org0;
            if (act && !(org >= 1 && org + 400 <= n)) all_inside = false;
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            std::vector<T> tmp(snap);
            nemo_phase1<T>(fl, j, act, all_inside, pcm, (f0 + fl) * hop + org0, n, preemph, tblob, tmp.data());
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next; snap = slice;
        std::vector<cpx<T>> own(64 * 16);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            fb_phase2_dft<T>(fl, j, act, snap.data(), *reinterpret_cast<cpx<T>(*)[16]>(&own[static_cast<size_t>(lane) * 16]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            std::vector<T> tmp(snap);
            cpx<T> part[8];
            const int src = (lane & ~15) | ((16 - (lane & 15)) & 15);          // what partner16() fetches on the device
            for (int i = 0; i < 8; ++i) part[i] = own[static_cast<size_t>(src) * 16 + 8 + i];
            fb_phase2_split<T, true>(fl, j, act, tblob, *reinterpret_cast<const cpx<T>(*)[16]>(&own[static_cast<size_t>(lane) * 16]),
                               part, tmp.data());
            const uint32_t *a = reinterpret_cast<const uint32_t *>(tmp.data()), *b0 = reinterpret_cast<const uint32_t *>(snap.data());
            uint32_t *d = reinterpret_cast<uint32_t *>(next.data());
            for (size_t i = 0; i < tmp.size() * sizeof(T) / 4; ++i) if (a[i] != b0[i]) d[i] = a[i];
        }
        slice = next;
        std::vector<float> rise(64 * NS), fprev(65 * NS, 0.0f);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            int st[NS];
            for (int i = 0; i < NS; ++i) st[i] = lane < kFbFPW * kFbLanes ? starts[i * kFbLanes + j] : 0;
            fb_phase3_sums<T, NS>(fl, j, act, F.slots, mel, slice.data(), st,
                                  *reinterpret_cast<float(*)[NS]>(&rise[static_cast<size_t>(lane) * NS]),
                                  *reinterpret_cast<float(*)[NS]>(&fprev[static_cast<size_t>(lane) * NS]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            nemo_phase3_store<NS>(fl, j, lane < kFbFPW * kFbLanes && fl < ns, act, n_mels, guard,
                                  *reinterpret_cast<const float(*)[NS]>(&rise[static_cast<size_t>(lane) * NS]),
                                  *reinterpret_cast<const float(*)[NS]>(&fprev[static_cast<size_t>(lane + 1) * NS]),
                                  out + f0, cols);
        }
    }
    return valid;
}

extern "C" long long emu_blm_wave(const float *pcm, long long n, int hop, int n_mels, int sample_rate, double f_min, double f_max,
                                  int htk, int norm, float preemph, int center, float guard, long long cols, float *out) {
    return run_blm<double>(pcm, n, hop, n_mels, sample_rate, f_min, f_max, htk, norm, preemph, center, guard, cols, out);
}
// the same kernel source instantiated in f32 (not shipped: tools/nemo_f32_calib.py measures what an f32 NeMo kernel would cost in accuracy)
extern "C" long long emu_blm_wave_f32(const float *pcm, long long n, int hop, int n_mels, int sample_rate, double f_min, double f_max,
                                      int htk, int norm, float preemph, int center, float guard, long long cols, float *out) {
    return run_blm<float>(pcm, n, hop, n_mels, sample_rate, f_min, f_max, htk, norm, preemph, center, guard, cols, out);
}

// Precise kernel (whisper_wave_f64.hpp): f64 phases 1-2, shared f32 interval mel phases.
// VERDICT r03 "next" 1(c), measured rather than argued: phase 2 of the precise kernel in f32 -- the exchange rows (f64 window, f64 first
// DFT-20, f64 twiddle) rounded to f32, the DFT-10s and the Hermitian split in f32 (the 16-operation form of six_phase2).  A dominant line
// then only leaks rounding noise into its own residue class mod 20.  Not a kernel: tools/mixed_f64_f32_calib.py runs the zoo through it.
static void mixed_phase2(int fl, int j, bool active, const double *tb, double *rows) {
    if (!active) return;
    const int brow = (j == 0) ? 20 : 20 - j;
    const double *ua = rows + fl * PreciseLayout::kXStride + j * PreciseLayout::kXRow;
    const double *va = rows + fl * PreciseLayout::kXStride + brow * PreciseLayout::kXRow;
    cpx<float> u[10], v[10];
    for (int i = 0; i < 10; ++i) {
        u[i] = {static_cast<float>(ua[2 * i]), static_cast<float>(ua[2 * i + 1])};
        v[i] = {static_cast<float>(va[2 * i]), static_cast<float>(va[2 * i + 1])};
    }
    fft10(u);
    fft10(v);
    const double *tw = tb + PreciseBlob::kTw2 + j * PreciseBlob::kTw2Stride;       // (2 sin a, 4 cos a), a = -2 pi k / 400
    float *p = reinterpret_cast<float *>(rows) + fl * WaveLayout::kPStride;
    float pk[10], pm[10];
    for (int q = 0; q < 10; ++q) {
        const cpx<float> zk = u[q], zm = v[9 - q];
        const cpx<float> W = {static_cast<float>(tw[2 * q + 1] / 4.0), static_cast<float>(tw[2 * q] / 2.0)};
        const cpx<float> S = {zk.re + zm.re, zk.im - zm.im}, D = {zk.re - zm.re, zk.im + zm.im};
        const cpx<float> wd = cmul(W, D);
        const float ar = S.re + wd.im, ai = S.im - wd.re, br = S.re - wd.im, bi = S.im + wd.re;
        pk[q] = ar * ar + ai * ai;
        pm[q] = br * br + bi * bi;
    }
    for (int q = 0; q < 10; ++q) { p[j + 20 * q] = pk[q]; p[200 - j - 20 * q] = pm[q]; }
}

static long long emu_whisper_precise_impl(const float *pcm, long long n, int hop, int n_mels, double sr, float *out, bool mixed);
extern "C" long long emu_whisper_precise(const float *pcm, long long n, int hop, int n_mels, double sr, float *out) {
    return emu_whisper_precise_impl(pcm, n, hop, n_mels, sr, out, false);
}
extern "C" long long emu_whisper_mixed(const float *pcm, long long n, int hop, int n_mels, double sr, float *out) {
    return emu_whisper_precise_impl(pcm, n, hop, n_mels, sr, out, true);
}
static long long emu_whisper_precise_impl(const float *pcm, long long n, int hop, int n_mels, double sr, float *out, bool mixed) {
    FastTables T;
    if (!build_fast_tables(sr, n_mels, T, true) || !T.interval) return -1;
    PreciseTables P;
    if (!build_precise_tables(T, P, true)) return -2;
    const double *tb = reinterpret_cast<const double *>(P.blob.data());
    const float *fblob = reinterpret_cast<const float *>(P.blob.data() + P.mel_off_words) - FastBlob::kMelStart;
    if (n < 400) return 0;
    const long long frames = (n - 400) / hop + 1;
    constexpr int NS = 12;
    std::vector<double> rows(PreciseLayout::slice_doubles());
    std::vector<float> vals(64 * NS);
    const int *starts = reinterpret_cast<const int *>(fblob + FastBlob::kMelStart);
    for (long long f0 = 0; f0 < frames; f0 += kFPW) {
        const int nv = static_cast<int>(std::min<long long>(kFPW, frames - f0));
        std::fill(rows.begin(), rows.end(), 1.0e30);
        std::vector<double> snap(rows), next(rows);
        auto merge = [&](const std::vector<double> &tmp) {
            const uint32_t *a = reinterpret_cast<const uint32_t *>(tmp.data()), *b0 = reinterpret_cast<const uint32_t *>(snap.data());
            uint32_t *d = reinterpret_cast<uint32_t *>(next.data());
            for (size_t i = 0; i < tmp.size() * 2; ++i) if (a[i] != b0[i]) d[i] = a[i];
        };
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
            const bool act = lane < kFPW * kMelJobs && fl < nv;
            std::vector<double> tmp(snap);
            precise_phase1(fl, j, act && j < kFftJobs, tb, pcm + (f0 + fl) * hop, tmp.data());
            merge(tmp);
        }
        rows = next; snap = rows;
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
            const bool act = lane < kFPW * kMelJobs && fl < nv;
            std::vector<double> tmp(snap);
            if (mixed) mixed_phase2(fl, j, act, tb, tmp.data());
            else precise_phase2(fl, j, act, tb, tmp.data());
            merge(tmp);
        }
        rows = next; snap = rows;
        float *slice = reinterpret_cast<float *>(rows.data());
        std::vector<float> rise(64 * NS), fprev(65 * NS, 0.0f);
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / 12, j = lane - fl * 12;
            const bool act = lane < kFPW * 12 && fl < nv;
            int st[NS];
            for (int i = 0; i < NS; ++i) st[i] = lane < kFPW * 12 ? starts[i * 12 + j] : 0;
            wave_phase3i_sums<NS, LensRuntime>(fl, j, act, T.slots, fblob, slice, st,
                                               *reinterpret_cast<float(*)[NS]>(&rise[static_cast<size_t>(lane) * NS]),
                                               *reinterpret_cast<float(*)[NS]>(&fprev[static_cast<size_t>(lane) * NS]));
        }
        std::vector<float> pm(64, 0.0f);
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / 12, j = lane - fl * 12;
            const bool act = lane < kFPW * 12 && fl < nv;
            std::vector<double> tmp(rows);
            wave_phase3i_finish<NS>(fl, j, act, n_mels, *reinterpret_cast<const float(*)[NS]>(&rise[static_cast<size_t>(lane) * NS]),
                                    *reinterpret_cast<const float(*)[NS]>(&fprev[static_cast<size_t>(lane + 1) * NS]),
                                    reinterpret_cast<float *>(tmp.data()), *reinterpret_cast<float(*)[NS]>(&vals[static_cast<size_t>(lane) * NS]));
            if (act) pm[lane] = reinterpret_cast<float *>(tmp.data())[WaveLayout::kPmaxOff + fl * WaveLayout::kPmaxStride + j];
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / 12, j = lane - fl * 12;
            if (lane < kFPW * 12 && fl < nv) slice[WaveLayout::kPmaxOff + fl * WaveLayout::kPmaxStride + j] = pm[lane];
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / 12, j = lane - fl * 12;
            const bool act = lane < kFPW * 12 && fl < nv;
            wave_phase4<NS, false>(fl, j, act, act, n_mels, slice, *reinterpret_cast<const float(*)[NS]>(&vals[static_cast<size_t>(lane) * NS]),
                                   out + f0 * n_mels, 0);
        }
    }
    return frames;
}

// STFT export on the f64 phases (precise_phase2_spectrum): out = [frames][bins] complex128, bins = 201 or 400
extern "C" long long emu_stft400(const float *pcm, long long n, int hop, int bins, double *out) {
    FastTables T;
    if (!build_fast_tables(16000.0, 80, T, true) || !T.interval) return -1;
    PreciseTables P;
    if (!build_precise_tables(T, P, false)) return -2;
    const double *tb = reinterpret_cast<const double *>(P.blob.data());
    if (n < 400) return 0;
    const long long frames = (n - 400) / hop + 1;
    std::vector<double> rows(PreciseLayout::slice_doubles());
    for (long long f0 = 0; f0 < frames; f0 += kFPW) {
        const int nv = static_cast<int>(std::min<long long>(kFPW, frames - f0));
        std::fill(rows.begin(), rows.end(), 1.0e30);
        std::vector<double> snap(rows), next(rows);
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
            const bool act = lane < kFPW * kMelJobs && fl < nv;
            std::vector<double> tmp(snap);
            precise_phase1(fl, j, act && j < kFftJobs, tb, pcm + (f0 + fl) * hop, tmp.data());
            const uint32_t *a = reinterpret_cast<const uint32_t *>(tmp.data()), *b0 = reinterpret_cast<const uint32_t *>(snap.data());
            uint32_t *d = reinterpret_cast<uint32_t *>(next.data());
            for (size_t i = 0; i < tmp.size() * 2; ++i) if (a[i] != b0[i]) d[i] = a[i];
        }
        rows = next;
        for (int lane = 0; lane < 64; ++lane) {
            const int fl = lane / kMelJobs, j = lane - fl * kMelJobs;
            const bool act = lane < kFPW * kMelJobs && fl < nv;
            precise_phase2_spectrum<double>(fl, j, act, tb, rows.data(), out + (f0 + fl) * 2 * bins, bins);
        }
    }
    return frames;
}

// small DFT checks
extern "C" void emu_small_fft(int n, float *interleaved) {
    if (n == 10) { cf x[10]; std::memcpy(x, interleaved, sizeof x); fft10(x); std::memcpy(interleaved, x, sizeof x); }
    if (n == 20) { cf x[20]; std::memcpy(x, interleaved, sizeof x); fft20(x); std::memcpy(interleaved, x, sizeof x); }
    if (n == 8)  { cf x[8];  std::memcpy(x, interleaved, sizeof x); fft8(x);  std::memcpy(interleaved, x, sizeof x); }
    if (n == 16) { cf x[16]; std::memcpy(x, interleaved, sizeof x); fft16(x); std::memcpy(interleaved, x, sizeof x); }
}

// ---- tga_quant.hpp: the three kernels of the quantiser, thread by thread ---------------------------
static void emu_quant_fill(QuantDesc &d, uint32_t rows, uint64_t width, uint32_t n_images, uint64_t img_stride, uint64_t blob_stride,
                           int header, const void *img) {
    d = QuantDesc{};
    d.rows = rows; d.width = static_cast<uint32_t>(width); d.n_images = n_images; d.img_stride = img_stride; d.blob_stride = blob_stride;
    d.header = header ? kTgaHeader : 0;
    if (header) {
        d.chunks = static_cast<uint32_t>((width + kTgaMaxWidth - 1) / kTgaMaxWidth);
        d.chunk_w = static_cast<uint32_t>(width < kTgaMaxWidth ? width : kTgaMaxWidth);
        d.chunk_stride = (kTgaHeader + static_cast<uint64_t>(rows) * d.chunk_w + 3) & ~3ull;
    } else {
        d.chunks = 1; d.chunk_w = d.width;
    }
    d.vec = d.chunks == 1 && reinterpret_cast<uintptr_t>(img) % 16 == 0 && (n_images == 1 || img_stride % 4 == 0);
}

extern "C" int emu_tga_encode(const float *img, uint32_t rows, uint64_t width, uint32_t n_images, uint64_t img_stride, uint8_t *blob,
                              uint64_t blob_stride, int header, float *ranges) {
    QuantDesc d;
    emu_quant_fill(d, rows, width, n_images, img_stride, blob_stride, header, img);
    const uint32_t items = n_images * d.chunks;
    std::vector<uint32_t> keys(2 * static_cast<size_t>(items));
    for (uint32_t i = 0; i < items; ++i) { keys[2 * i] = kKeyPosInf; keys[2 * i + 1] = kKeyNegInf; }
    for (uint32_t item = 0; item < items; ++item) {
        const uint32_t image = item / d.chunks, c = item - image * d.chunks, cw = chunk_cols(d, c);
        const uint64_t npx = static_cast<uint64_t>(rows) * cw;
        const float *im = img + image * img_stride;
        // min/max pass: per-thread partial folds merged through the ordered keys, in a scrambled order
        for (uint64_t t = 0; t < (npx + 3) / 4; ++t) {
            const uint64_t tt = ((npx + 3) / 4) - 1 - t;             // any order gives the same keys; run it backwards
            float mn = INFINITY, mx = -INFINITY;
            for (uint64_t i = 4 * tt; i < 4 * tt + 4 && i < npx; ++i) {
                const float v = im[image_index(d, c, cw, i)];
                mn = fminf(mn, v); mx = fmaxf(mx, v);
            }
            keys[2 * item] = std::min(keys[2 * item], ordered_key(mn));
            keys[2 * item + 1] = std::max(keys[2 * item + 1], ordered_key(mx));
        }
        const float mn = key_to_float(keys[2 * item]), mx = key_to_float(keys[2 * item + 1]);
        const float scale = f32_div_rn(255.0f, mx - mn);
        if (ranges) { ranges[2 * item] = mn; ranges[2 * item + 1] = mx; }
        uint32_t *out = reinterpret_cast<uint32_t *>(blob + image * blob_stride + c * d.chunk_stride);
        const uint64_t ndw = (d.header + npx + 3) / 4;
        for (uint64_t dw = 0; dw < ndw; ++dw) out[dw] = encode_dword(d, im, c, cw, npx, dw, mn, mx, scale);
    }
    return static_cast<int>(items);
}

extern "C" int emu_tga_decode(const uint8_t *blob, uint64_t blob_stride, uint32_t rows, uint64_t width, uint32_t n_images, float *img,
                              uint64_t img_stride, int header, const float *ranges) {
    QuantDesc d;
    emu_quant_fill(d, rows, width, n_images, img_stride, blob_stride, header, img);
    const uint32_t items = n_images * d.chunks;
    for (uint32_t item = 0; item < items; ++item) {
        const uint32_t image = item / d.chunks, c = item - image * d.chunks, cw = chunk_cols(d, c);
        const uint64_t npx = static_cast<uint64_t>(rows) * cw;
        const uint8_t *b = blob + image * blob_stride + c * d.chunk_stride;
        float mn, mx;
        if (header) { std::memcpy(&mn, b + 18, 4); std::memcpy(&mx, b + 22, 4); }
        else { mn = ranges[2 * item]; mx = ranges[2 * item + 1]; }
        const float scale = f32_div_rn(mx - mn, 255.0f);
        const uint64_t ndw = (d.header + npx + 3) / 4;
        for (uint64_t dw = 0; dw < ndw; ++dw) {
            uint32_t w = 0;
            std::memcpy(&w, b + 4 * dw, 4);
            decode_dword(d, img + image * img_stride, c, cw, npx, dw, w, mn, scale);
        }
    }
    return static_cast<int>(items);
}

// ---- stream_plan.hpp: the streaming bank's bookkeeping with host stand-ins for the three kernels --------
#include "../../mel_spec_amd/csrc/stream_plan.hpp"
extern "C" long long emu_whisper_wave(const float *pcm, long long n, int hop, int n_mels, double sr, int mode, float *out);
struct EmuStream {
    StreamGeom g;
    StreamBook bk;
    std::vector<float> state;
    double sr;
};
extern "C" void *emu_stream_create(int hop, int n_mels, double sr, uint32_t n_streams, uint32_t max_chunk) {
    EmuStream *s = new EmuStream();
    s->g = stream_geometry(400, static_cast<uint32_t>(hop), static_cast<uint32_t>(n_mels), n_streams, max_chunk);
    s->bk.reset(n_streams);
    s->state.assign(static_cast<size_t>(n_streams) * s->g.stride + 16, 0.0f);
    s->sr = sr;
    return s;
}
extern "C" void emu_stream_destroy(void *p) { delete static_cast<EmuStream *>(p); }
extern "C" long long emu_stream_frames_after(void *p, uint32_t id, uint32_t n_new) {
    EmuStream *s = static_cast<EmuStream *>(p);
    return static_cast<long long>(stream_frames_after(s->g, s->bk, id, n_new));
}
// returns total frames (>= 0) or -(error code)
extern "C" long long emu_stream_push(void *p, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n, int flush,
                                     float *out, uint32_t *frames_out) {
    EmuStream *s = static_cast<EmuStream *>(p);
    StreamPlan pl;
    const char *err = nullptr;
    const int rc = stream_plan_push(s->g, s->bk, ids, lens, n, flush != 0, pl, &err);
    if (rc) return -rc;
    for (uint32_t i = 0; i < n; ++i) {                                  // stream_scatter_kernel
        const StreamEntry &e = pl.entries[i];
        float *dst = s->state.data() + e.stream * s->g.stride + s->g.in_off;
        for (uint32_t k = 0; k < e.len; ++k) dst[k] = samples[e.src_off + k];
        for (uint32_t k = 0; k < e.zero_fill; ++k) dst[e.len + k] = 0.0f;
    }
    for (uint32_t i = 0; i < n; ++i) {                                  // the batch kernel on carry ++ chunk
        if (!pl.frames[i]) continue;
        const long long got = emu_whisper_wave(s->state.data() + pl.off[i], static_cast<long long>(pl.len[i]), static_cast<int>(s->g.hop),
                                               static_cast<int>(s->g.n_mels), s->sr, 0, out + pl.out_off[i]);
        if (got != pl.frames[i]) return -100;
    }
    for (uint32_t i = 0; i < n; ++i) {                                  // stream_carry_kernel
        const StreamEntry &e = pl.entries[i];
        const uint32_t nn = e.len + e.zero_fill;
        if (!nn) continue;
        float *slot = s->state.data() + e.stream * s->g.stride;
        std::memmove(slot + s->g.in_off - e.keep, slot + s->g.in_off + nn - e.keep, e.keep * sizeof(float));
    }
    stream_commit_push(s->g, s->bk, ids, lens, n, flush != 0);
    if (frames_out) for (uint32_t i = 0; i < n; ++i) frames_out[i] = pl.frames[i];
    return static_cast<long long>(pl.total_frames);
}

// ---- vad_columns.hpp: the per-thread functions of the two mask kernels ---------------------------------
extern "C" long long emu_vad_boundaries(const float *img, uint32_t height, uint32_t width, int min_mel, int min_y, double min_energy,
                                        uint8_t *raw, uint8_t *smoothed) {
    if (height < 3 || width < 3) return 0;
    const uint32_t n = width - 2;
    for (uint32_t x = 0; x < n; ++x) raw[x] = vad_classify_column(img, height, width, x, min_mel, min_y, min_energy * min_energy);
    for (uint32_t x = 0; x < n; ++x) smoothed[x] = vad_smooth_at(raw, n, x);
    return n;
}

// The detector stage of the streaming bank (stream_vad_kernel), one stream, the workgroup's steps in sequence: `rows` = the F new
// [n_mels] rows of one push; state = {count, hist}; prev = [2][n_mels]; acts = F records of 8 bytes.
extern "C" void emu_stream_vad_push(const float *rows, uint32_t F, uint32_t H, int min_mel, int min_y, int min_x, double min_energy,
                                    uint64_t *state /* count, hist */, float *prev, uint8_t *acts) {
    if (F == 0) return;
    std::vector<uint8_t> rwin(64 + F);
    const uint64_t count = state[0], hist = state[1];
    for (uint32_t i = 0; i < 64; ++i) rwin[63 - i] = static_cast<uint8_t>((hist >> i) & 1u);
    for (uint32_t q = 0; q < F; ++q) {
        bool r = false;
        if (count + q >= 2 && H >= 3) {
            const float *c2 = rows + static_cast<uint64_t>(q) * H;
            const float *c1 = q >= 1 ? c2 - H : prev + H;
            const float *c0 = q >= 2 ? c2 - 2 * static_cast<uint64_t>(H) : (q == 1 ? prev + H : prev);
            r = vad_classify_triple(c0, c1, c2, H, min_mel, min_y, min_energy * min_energy);
        }
        rwin[64 + q] = r;
    }
    for (uint32_t q = 0; q < F; ++q) {
        const VadActivity a = vad_stream_activity(count + q, rwin.data() + 64 + q, H, min_x);
        std::memcpy(acts + 8 * static_cast<size_t>(q), &a, 8);
    }
    uint64_t h2 = 0;
    for (uint32_t i = 0; i < 64; ++i) h2 |= static_cast<uint64_t>(rwin[64 + F - 1 - i] & 1u) << i;
    state[0] = count + F; state[1] = h2;
    for (uint32_t y = 0; y < H; ++y) {
        const float last = rows[static_cast<uint64_t>(F - 1) * H + y];
        const float before = F >= 2 ? rows[static_cast<uint64_t>(F - 2) * H + y] : prev[H + y];
        prev[y] = before; prev[H + y] = last;
    }
}

// ---- Whisper flavour of the fused 512-point kernel (w512_phase1 / fb_phase2_* / fb_phase3_sums / w512_phase3_log / w512_phase4) ----
extern "C" long long emu_w512_wave(const float *pcm, long long n, int hop, int n_mels, double sr, float *out) {
    using T = double;
    using L = FbankLayout<T>;
    constexpr int NS = kBlmSlots;
    FbankFastTables F;
    if (!build_whisper512_tables<T>(sr, n_mels, F)) return -1;
    const T *tblob = reinterpret_cast<const T *>(F.blob.data());
    const float *mel = reinterpret_cast<const float *>(F.blob.data() + F.mel_off_words);
    if (n < 512) return 0;
    const long long frames = (n - 512) / hop + 1;
    std::vector<T> slice(L::slice_elems());
    const int *starts = reinterpret_cast<const int *>(mel + FbankBlob::kMelStart);
    for (long long f0 = 0; f0 < frames; f0 += kFbFPW) {
        const int nv = static_cast<int>(std::min<long long>(kFbFPW, frames - f0));
        std::fill(slice.begin(), slice.end(), T(1.0e30));
        auto lane_info = [&](int lane, int &fl, int &j, bool &act) {
            fl = lane / kFbLanes; j = lane - fl * kFbLanes; act = lane < kFbFPW * kFbLanes && fl < nv;
        };
        std::vector<T> snap(slice), next(slice);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            std::vector<T> tmp(snap);
            w512_phase1<T>(fl, j, act, pcm + (f0 + (act ? fl : 0)) * hop, tblob, tmp.data());
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next; snap = slice;
        std::vector<cpx<T>> own(64 * 16);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            fb_phase2_dft<T>(fl, j, act, snap.data(), *reinterpret_cast<cpx<T>(*)[16]>(&own[static_cast<size_t>(lane) * 16]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            std::vector<T> tmp(snap);
            cpx<T> part[8];
            const int src = (lane & ~15) | ((16 - (lane & 15)) & 15);
            for (int i = 0; i < 8; ++i) part[i] = own[static_cast<size_t>(src) * 16 + 8 + i];
            fb_phase2_split<T, true, true>(fl, j, act, tblob, *reinterpret_cast<const cpx<T>(*)[16]>(&own[static_cast<size_t>(lane) * 16]), part, tmp.data());
            const uint32_t *a = reinterpret_cast<const uint32_t *>(tmp.data()), *b0 = reinterpret_cast<const uint32_t *>(snap.data());
            uint32_t *d = reinterpret_cast<uint32_t *>(next.data());
            for (size_t i = 0; i < tmp.size() * sizeof(T) / 4; ++i) if (a[i] != b0[i]) d[i] = a[i];
        }
        slice = next;
        float *slice_f = reinterpret_cast<float *>(slice.data());
        std::vector<float> rise(64 * NS), fprev(65 * NS, 0.0f), vals(64 * NS);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            int st[NS];
            for (int i = 0; i < NS; ++i) st[i] = starts[i * kFbLanes + j];
            fb_phase3_sums<T, NS>(fl, j, act, F.slots, mel, slice.data(), st,
                                  *reinterpret_cast<float(*)[NS]>(&rise[static_cast<size_t>(lane) * NS]),
                                  *reinterpret_cast<float(*)[NS]>(&fprev[static_cast<size_t>(lane) * NS]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            w512_phase3_log<NS>(fl, j, act, n_mels, *reinterpret_cast<const float(*)[NS]>(&rise[static_cast<size_t>(lane) * NS]),
                                *reinterpret_cast<const float(*)[NS]>(&fprev[static_cast<size_t>(lane + 1) * NS]), slice_f,
                                *reinterpret_cast<float(*)[NS]>(&vals[static_cast<size_t>(lane) * NS]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; lane_info(lane, fl, j, act);
            w512_phase4<NS>(fl, j, act, act, n_mels, slice_f, *reinterpret_cast<const float(*)[NS]>(&vals[static_cast<size_t>(lane) * NS]), out + f0 * n_mels, 0);
        }
    }
    return frames;
}

// ---- whisper_six.hpp: six frames per wave, ten lanes per frame ------------------------------------------------
template <int NSLOTS, class Lens>
static long long run_six(const float *pcm, long long n, int hop, int n_mels, double sr, float *out, uint8_t *flags) {
    FastTables T;
    if (!build_six_tables(sr, n_mels, T)) return -1;
    if (n < 400) return 0;
    const long long frames = (n - 400) / hop + 1;
    std::vector<float> slice(SixLayout::slice_floats());
    std::vector<float> vals(static_cast<size_t>(64) * NSLOTS);
    const int *starts = reinterpret_cast<const int *>(T.blob.data() + SixBlob::kMelStart);
    for (long long f0 = 0; f0 < frames; f0 += kSixFrames) {
        const int nv = static_cast<int>(std::min<long long>(kSixFrames, frames - f0));
        std::fill(slice.begin(), slice.end(), 1.0e30f);
        const float *src = pcm + f0 * hop;
        auto info = [&](int lane, int &fl, int &j, bool &act) { fl = lane / kSixLanes; j = lane - fl * kSixLanes; act = lane < kSixFrames * kSixLanes && fl < nv; };
        std::vector<float> snap(slice), next(slice);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; info(lane, fl, j, act);
            std::vector<float> tmp(snap);
            six_phase1(fl, j, act, hop, T.blob.data(), src, tmp.data());
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next; snap = slice;
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; info(lane, fl, j, act);
            std::vector<float> tmp(snap);
            int uoff, voff;
            SixLayout::row_offsets(j, uoff, voff);
            six_phase2(fl, j, act, T.blob.data(), tmp.data(), uoff, voff);
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next; snap = slice;
        std::vector<float> rise(64 * NSLOTS), fprev(65 * NSLOTS, 0.0f);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; info(lane, fl, j, act);
            int st[NSLOTS];
            for (int i = 0; i < NSLOTS; ++i) st[i] = lane < kSixFrames * kSixLanes ? starts[i * kSixLanes + j] : 0;
            six_phase3_sums<NSLOTS, Lens>(fl, j, act, T.slots, T.blob.data(), snap.data(), st,
                                          *reinterpret_cast<float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                          *reinterpret_cast<float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane) * NSLOTS]));
        }
        for (int lane = 0; lane < 64; ++lane) {     // wave_shl:1 -> lane l sees lane l+1
            int fl, j; bool act; info(lane, fl, j, act);
            std::vector<float> tmp(snap);
            six_phase3_finish<NSLOTS>(fl, j, act, n_mels, *reinterpret_cast<const float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                      *reinterpret_cast<const float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane + 1) * NSLOTS]), tmp.data(),
                                      *reinterpret_cast<float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]));
            for (size_t i = 0; i < tmp.size(); ++i) if (tmp[i] != snap[i]) next[i] = tmp[i];
        }
        slice = next;
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; info(lane, fl, j, act);
            const bool g = six_phase4<NSLOTS, false, true>(fl, j, act, act, n_mels, slice.data(), *reinterpret_cast<const float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]),
                               out + f0 * n_mels, 0);
            if (g && flags) flags[f0 + fl] = 1;
        }
    }
    return frames;
}

template <class Lens>
static bool six_lens_ok(const MelSlots &ms, int n_mels) {
    if (ms.n_slots != Lens::kSlots || n_mels != Lens::kMels) return false;
    for (int i = 0; i < Lens::kSlots; ++i)
        if (ms.len[i] != Lens::len(i) || ms.woff[i] != Lens::woff(i)) return false;
    return true;
}

// mode 0: runtime slot lengths, 1: compile-time lengths (Whisper 80 mels only)
extern "C" long long emu_whisper_six_guard(const float *pcm, long long n, int hop, int n_mels, double sr, int mode, float *out, uint8_t *flags) {
    if (mode == 1) {
        FastTables T;
        if (!build_six_tables(sr, n_mels, T) || !six_lens_ok<LensSix80>(T.slots, n_mels)) return -2;
        return run_six<kSixMaxSlots, LensSix80>(pcm, n, hop, n_mels, sr, out, flags);
    }
    return run_six<kSixMaxSlots, LensRuntime>(pcm, n, hop, n_mels, sr, out, flags);
}
extern "C" long long emu_whisper_six(const float *pcm, long long n, int hop, int n_mels, double sr, int mode, float *out) {
    return emu_whisper_six_guard(pcm, n, hop, n_mels, sr, mode, out, nullptr);
}

// ---- whisper_six64.hpp: the f64 kernel on the six-frame skeleton, exchange through LDS in two halves ---------------------------------
// Step by step as whisper400_six64_kernel runs them, every step over all 64 lanes before the next (what "LDS operations of a wave execute
// in program order" gives the device): phase 1 -> rows 0..9 -> every lane reads its first row -> rows 10..19 over them -> second row ->
// phase 2 -> the f32 phases 3-4 of whisper_six.hpp.
template <int NSLOTS, class Lens>
static long long run_six64(const float *pcm, long long n, int hop, int n_mels, double sr, float *out) {
    FastTables T;
    Six64Tables T64;
    if (!build_six_tables(mel_filterbank(sr, 400, n_mels, -1.0, -1.0, false, true), n_mels, T, NSLOTS) || !build_six64_tables(T, T64)) return -1;
    if (n < 400) return 0;
    const long long frames = (n - 400) / hop + 1;
    const double *tb = reinterpret_cast<const double *>(T64.blob.data());
    const float *fblob = reinterpret_cast<const float *>(T64.blob.data() + T64.mel_off_words) - SixBlob::kMelStart;
    std::vector<double> rows(Six64Layout::slice_doubles());
    float *slice = reinterpret_cast<float *>(rows.data());
    std::vector<float> vals(static_cast<size_t>(64) * NSLOTS);
    const int *starts = reinterpret_cast<const int *>(fblob + SixBlob::kMelStart);
    for (long long f0 = 0; f0 < frames; f0 += kSixFrames) {
        const int nv = static_cast<int>(std::min<long long>(kSixFrames, frames - f0));
        std::fill(rows.begin(), rows.end(), 1.0e300);
        const float *src = pcm + f0 * hop;
        auto info = [&](int lane, int &fl, int &j, bool &act) { fl = lane / kSixLanes; j = lane - fl * kSixLanes; act = lane < kSixFrames * kSixLanes && fl < nv; };
        std::vector<cd> x(64 * 20), u(64 * 10), v(64 * 10);
        auto X = [&](int lane) -> cd(&)[20] { return *reinterpret_cast<cd(*)[20]>(&x[static_cast<size_t>(lane) * 20]); };
        auto U = [&](std::vector<cd> &a, int lane) -> cd(&)[10] { return *reinterpret_cast<cd(*)[10]>(&a[static_cast<size_t>(lane) * 10]); };
        for (int lane = 0; lane < 64; ++lane) { int fl, j; bool act; info(lane, fl, j, act); six64_phase1(fl, j, act, hop, tb, src, X(lane)); }
        for (int lane = 0; lane < 64; ++lane) { int fl, j; bool act; info(lane, fl, j, act); six64_store_half(fl, j, act, 0, X(lane), rows.data()); }
        for (int lane = 0; lane < 64; ++lane) { int fl, j; bool act; info(lane, fl, j, act); six64_read_row(fl, act, Six64Layout::row_offset(j), rows.data(), U(u, lane)); }
        for (int lane = 0; lane < 64; ++lane) { int fl, j; bool act; info(lane, fl, j, act); six64_store_half(fl, j, act, 1, X(lane), rows.data()); }
        for (int lane = 0; lane < 64; ++lane) { int fl, j; bool act; info(lane, fl, j, act); six64_read_row(fl, act, Six64Layout::row_offset(j), rows.data(), U(v, lane)); }
        for (int lane = 0; lane < 64; ++lane) { int fl, j; bool act; info(lane, fl, j, act); six64_phase2(fl, j, act, tb, U(u, lane), U(v, lane), slice); }
        std::vector<float> rise(64 * NSLOTS), fprev(65 * NSLOTS, 0.0f);
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; info(lane, fl, j, act);
            int st[NSLOTS];
            for (int i = 0; i < NSLOTS; ++i) st[i] = lane < kSixFrames * kSixLanes ? starts[i * kSixLanes + j] : 0;
            six_phase3_sums<NSLOTS, Lens>(fl, j, act, T.slots, fblob, slice, st,
                                          *reinterpret_cast<float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                          *reinterpret_cast<float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane) * NSLOTS]));
        }
        for (int lane = 0; lane < 64; ++lane) {     // the maxima go behind the power rows, which every lane has read by now
            int fl, j; bool act; info(lane, fl, j, act);
            six_phase3_finish<NSLOTS>(fl, j, act, n_mels, *reinterpret_cast<const float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                      *reinterpret_cast<const float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane + 1) * NSLOTS]), slice,
                                      *reinterpret_cast<float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]));
        }
        for (int lane = 0; lane < 64; ++lane) {
            int fl, j; bool act; info(lane, fl, j, act);
            six_phase4<NSLOTS, false, false>(fl, j, act, act, n_mels, slice, *reinterpret_cast<const float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]),
                                             out + f0 * n_mels, 0);
        }
    }
    return frames;
}

// mode 0: run-time slot lengths, 1: the compile-time Whisper-80 bank
extern "C" long long emu_whisper_six64(const float *pcm, long long n, int hop, int n_mels, double sr, int mode, float *out) {
    if (mode == 1) {
        FastTables T;
        if (!build_six_tables(sr, n_mels, T) || !six_lens_ok<LensSix80>(T.slots, n_mels)) return -2;
        return run_six64<kSixMaxSlots, LensSix80>(pcm, n, hop, n_mels, sr, out);
    }
    if (mode == 2) {        // fifteen mel slots: Whisper large-v3's 128-mel bank (the f64 kernel only)
        FastTables T;
        if (!build_six_tables(mel_filterbank(sr, 400, n_mels, -1.0, -1.0, false, true), n_mels, T, kSixWideSlots) || !six_lens_ok<LensSix128>(T.slots, n_mels)) return -2;
        return run_six64<kSixWideSlots, LensSix128>(pcm, n, hop, n_mels, sr, out);
    }
    if (mode == 3) return run_six64<kSixWideSlots, LensRuntime>(pcm, n, hop, n_mels, sr, out);      // fifteen slots, run-time lengths (81..134 mels)
    return run_six64<kSixMaxSlots, LensRuntime>(pcm, n, hop, n_mels, sr, out);
}

// The in-kernel f64 recompute of one frame (whisper_fix64.hpp + the f32 phases 3-4 of the kernel that owns the frame), as
// six_fix_unit / wave_fix_unit run it: frame slot f of a unit, the other slots idle.
static void emu_fix_power_row(const float *frame, const std::vector<double> &tab, float *prow) {
    std::vector<double> z(FixTables::kScratchDoubles, 1.0e30), nxt;
    auto step = [&](auto fn) {
        nxt = z;
        for (int lane = 0; lane < 64; ++lane) {
            std::vector<double> tmp(z);
            fn(lane, tmp.data());
            for (size_t i = 0; i < tmp.size(); ++i) if (std::memcmp(&tmp[i], &z[i], sizeof(double)) != 0) nxt[i] = tmp[i];
        }
        z = nxt;
    };
    step([&](int lane, double *zz) { fix_step1(lane, frame, tab.data(), zz); });
    step([&](int lane, double *zz) { fix_step2(lane, tab.data(), zz); });
    step([&](int lane, double *zz) { fix_step3(lane, zz); });
    for (int lane = 0; lane < 64; ++lane) fix_step4(lane, tab.data(), z.data(), prow);
}

template <int NSLOTS>
static int emu_fix_frame_six(const float *frame, int n_mels, double sr, float *out_row) {
    FastTables T;
    if (!build_six_tables(sr, n_mels, T)) return -1;
    const std::vector<double> tab = build_fix_tables();
    const int f = 3;                                             // any slot: exercise a non-zero one
    std::vector<float> slice(SixLayout::slice_floats(), 1.0e30f);
    emu_fix_power_row(frame, tab, slice.data() + f * SixLayout::kPStride);
    const int *starts = reinterpret_cast<const int *>(T.blob.data() + SixBlob::kMelStart);
    std::vector<float> rise(64 * NSLOTS), fprev(65 * NSLOTS, 0.0f), vals(64 * NSLOTS);
    std::vector<float> tile(static_cast<size_t>(kSixFrames) * n_mels, 0.0f);
    auto info = [&](int lane, int &fl, int &j, bool &act) { fl = lane / kSixLanes; j = lane - fl * kSixLanes; act = lane < kSixFrames * kSixLanes && fl == f; };
    for (int lane = 0; lane < 64; ++lane) {
        int fl, j; bool act; info(lane, fl, j, act);
        int st[NSLOTS];
        for (int i = 0; i < NSLOTS; ++i) st[i] = lane < kSixFrames * kSixLanes ? starts[i * kSixLanes + j] : 0;
        six_phase3_sums<NSLOTS, LensRuntime>(fl, j, act, T.slots, T.blob.data(), slice.data(), st,
                                             *reinterpret_cast<float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                             *reinterpret_cast<float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane) * NSLOTS]));
    }
    for (int lane = 0; lane < 64; ++lane) {
        int fl, j; bool act; info(lane, fl, j, act);
        six_phase3_finish<NSLOTS>(fl, j, act, n_mels, *reinterpret_cast<const float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                  *reinterpret_cast<const float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane + 1) * NSLOTS]), slice.data(),
                                  *reinterpret_cast<float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]));
    }
    for (int lane = 0; lane < 64; ++lane) {
        int fl, j; bool act; info(lane, fl, j, act);
        six_phase4<NSLOTS, false, false>(fl, j, act, act, n_mels, slice.data(), *reinterpret_cast<const float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]),
                                         tile.data(), 0);
    }
    std::memcpy(out_row, tile.data() + static_cast<size_t>(f) * n_mels, sizeof(float) * n_mels);
    return 0;
}

template <int NSLOTS>
static int emu_fix_frame_wave(const float *frame, int n_mels, double sr, float *out_row) {
    FastTables T;
    if (!build_fast_tables(sr, n_mels, T, true) || !T.interval) return -1;
    const std::vector<double> tab = build_fix_tables();
    const int f = 2;
    std::vector<float> slice(WaveLayout::slice_floats(), 1.0e30f);
    emu_fix_power_row(frame, tab, slice.data() + f * WaveLayout::kPStride);
    const int *starts = reinterpret_cast<const int *>(T.blob.data() + FastBlob::kMelStart);
    std::vector<float> rise(64 * NSLOTS), fprev(65 * NSLOTS, 0.0f), vals(64 * NSLOTS);
    std::vector<float> tile(static_cast<size_t>(kFPW) * n_mels, 0.0f);
    for (int lane = 0; lane < 64; ++lane) {
        const int fl = lane / 12, j = lane - fl * 12;
        const bool act = lane < kFPW * 12 && fl == f;
        int st[NSLOTS];
        for (int i = 0; i < NSLOTS; ++i) st[i] = lane < kFPW * 12 ? starts[i * 12 + j] : 0;
        wave_phase3i_sums<NSLOTS, LensRuntime>(fl, j, act, T.slots, T.blob.data(), slice.data(), st,
                                               *reinterpret_cast<float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                               *reinterpret_cast<float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane) * NSLOTS]));
    }
    for (int lane = 0; lane < 64; ++lane) {
        const int fl = lane / 12, j = lane - fl * 12;
        const bool act = lane < kFPW * 12 && fl == f;
        wave_phase3i_finish<NSLOTS>(fl, j, act, n_mels, *reinterpret_cast<const float(*)[NSLOTS]>(&rise[static_cast<size_t>(lane) * NSLOTS]),
                                    *reinterpret_cast<const float(*)[NSLOTS]>(&fprev[static_cast<size_t>(lane + 1) * NSLOTS]), slice.data(),
                                    *reinterpret_cast<float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]));
    }
    for (int lane = 0; lane < 64; ++lane) {
        const int fl = lane / 12, j = lane - fl * 12;
        const bool act = lane < kFPW * 12 && fl == f;
        wave_phase4<NSLOTS, false, false>(fl, j, act, act, n_mels, slice.data(), *reinterpret_cast<const float(*)[NSLOTS]>(&vals[static_cast<size_t>(lane) * NSLOTS]),
                                          tile.data(), 0);
    }
    std::memcpy(out_row, tile.data() + static_cast<size_t>(f) * n_mels, sizeof(float) * n_mels);
    return 0;
}

// one 400-sample frame through the in-kernel f64 recompute; six != 0: as the six-frame kernel runs it
extern "C" int emu_fix_frame(const float *frame, int n_mels, double sr, int six, float *out_row) {
    if (six) return emu_fix_frame_six<kSixMaxSlots>(frame, n_mels, sr, out_row);
    FastTables T;
    if (!build_fast_tables(sr, n_mels, T, true) || !T.interval) return -1;
    return T.slots.n_slots <= 8 ? emu_fix_frame_wave<8>(frame, n_mels, sr, out_row) : emu_fix_frame_wave<12>(frame, n_mels, sr, out_row);
}

// MELSPEC_PRECISION_AUTO: the f32 kernel the library would pick (six frames per wave up to 80 mels, else five), then the
// frames its guard trips recomputed by the in-kernel f64 path.  Returns the frame count; *n_flagged the recomputed frames.
extern "C" long long emu_whisper_auto(const float *pcm, long long n, int hop, int n_mels, double sr, float *out, long long *n_flagged) {
    if (n_flagged) *n_flagged = 0;
    if (n < 400) return 0;
    const long long frames = (n - 400) / hop + 1;
    std::vector<uint8_t> flags(static_cast<size_t>(frames), 0);
    FastTables T6;
    const bool six = build_six_tables(sr, n_mels, T6);
    long long got = six ? emu_whisper_six_guard(pcm, n, hop, n_mels, sr, 0, out, flags.data())
                        : emu_whisper_wave_guard(pcm, n, hop, n_mels, sr, 0, out, flags.data());
    if (got != frames) return got;
    long long nf = 0;
    for (long long f = 0; f < frames; ++f) {
        if (!flags[f]) continue;
        ++nf;
        if (emu_fix_frame(pcm + f * hop, n_mels, sr, six ? 1 : 0, out + f * n_mels) != 0) return -4;
    }
    if (n_flagged) *n_flagged = nf;
    return frames;
}


// ---- pow2_wave.hpp: the in-place Stockham passes of pow2_frame_kernel, lane by lane ------------------------------------------------
// The device runs a pass as "every lane reads its inputs, then every lane writes" (one wave, LDS operations in program order); here a
// pass is run per lane on a copy of the pass's input and the words it changed are merged -- a lane that read something another lane
// of the same pass had already overwritten would show up as a wrong transform.
template <int LOGM>
static void emu_pow2_run(const double *in /* M complex */, double *out /* M complex, natural order */) {
    using S = Pow2Shape<LOGM>;
    constexpr int M = S::M;
    std::vector<double> tw(2 * M);
    for (int q = 0; q < M; ++q) { tw[2 * q] = std::cos(2.0 * kPi * q / (2.0 * M)); tw[2 * q + 1] = -std::sin(2.0 * kPi * q / (2.0 * M)); }
    constexpr int kT3 = S::R3 > 1 ? (S::R3 - 1) * S::R1 * 8 : 0;      // (the one-transform form; the kernel runs M = 1024 as two 512-point halves)
    std::vector<double> t2(2 * 7 * S::R1), t3(2 * (kT3 > 0 ? kT3 : 1));
    for (int i = 0; i < 7 * S::R1; ++i) { const cpx<double> w = pow2_table_entry(tw.data(), M, 8, S::R1, i); t2[2 * i] = w.re; t2[2 * i + 1] = w.im; }
    for (int i = 0; i < kT3; ++i) { const cpx<double> w = pow2_table_entry(tw.data(), M, S::R3, S::R1 * 8, i); t3[2 * i] = w.re; t3[2 * i + 1] = w.im; }
    std::vector<double> z(2 * M, 1.0e300);
    auto pass = [&](auto fn) {
        const std::vector<double> snap(z);
        std::vector<double> next(z);
        for (int l = 0; l < S::LF; ++l) {
            std::vector<double> tmp(snap);
            fn(l, tmp.data());
            for (size_t i = 0; i < tmp.size(); ++i) if (std::memcmp(&tmp[i], &snap[i], 8) != 0) next[i] = tmp[i];
        }
        z = next;
    };
    pass([&](int l, double *zz) {
        cpx<double> reg[S::P];
        for (int r = 0; r < S::P; ++r) reg[r] = {in[2 * (l + r * S::LF)], in[2 * (l + r * S::LF) + 1]};
        pow2_pass<LOGM, S::R1, true>(l, 1, nullptr, zz, reg, nullptr);
    });
    pass([&](int l, double *zz) { pow2_pass<LOGM, 8, false>(l, S::R1, t2.data(), zz, nullptr, nullptr); });
    if (S::R3 > 1) pass([&](int l, double *zz) { pow2_pass<LOGM, (S::R3 > 1 ? S::R3 : 2), false>(l, S::R1 * 8, t3.data(), zz, nullptr, nullptr); });
    for (int k = 0; k < M; ++k) { out[2 * k] = z[2 * pow2_slot<LOGM>(k)]; out[2 * k + 1] = z[2 * pow2_slot<LOGM>(k) + 1]; }
}
extern "C" int emu_pow2_fft(int logm, const double *in, double *out) {
    switch (logm) {
        case 6: emu_pow2_run<6>(in, out); return 0;
        case 7: emu_pow2_run<7>(in, out); return 0;
        case 8: emu_pow2_run<8>(in, out); return 0;
        case 9: emu_pow2_run<9>(in, out); return 0;
        case 10: emu_pow2_run<10>(in, out); return 0;
    }
    return -1;
}
