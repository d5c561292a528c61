// fast_tables.hpp -- host builder of the constant blob the n_fft=400 kernel keeps in LDS
// (layout: FastBlob in whisper_fast.hpp).  All values are computed in f64 and rounded once.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "tables.hpp"
#include "whisper_fast.hpp"

namespace melspec {

struct FastTables {
    std::vector<float> blob;
    MelSlots slots{};
    int n_mels = 0;
    int nnz = 0;
    bool interval = false;   // mel section holds the interval scheme (see build_interval_mel)
};

// Interval form of a triangular filterbank.  With edges e_0..e_{M+1}, bin k lies in exactly one
// interval I_i = [e_i, e_{i+1}) and only filters i (rising there) and i-1 (falling there) can be
// non-zero on it, so  mel[m] = sum_{k in I_m} w[m][k] P[k] + sum_{k in I_{m+1}} w[m][k] P[k]:
// every bin is read once and feeds two running sums.  Lane j of a 12-lane frame group owns interval
// i = j + 11*slot (j = 11 duplicates j = 0 of the next slot so that the "fall" sum a mel needs is
// always in the next lane).  Weights are pre-scaled by 1/4 (phase 2 then stores 4*|X|^2).
// Returns false if some non-zero weight violates the two-filters-per-bin structure.
inline bool build_interval_mel(const std::vector<double> &dense, int n_mels, int bins, int bin_limit,
                               std::vector<float> &b, MelSlots &slots, int lanes = 12, double scale = 0.25,
                               int starts_off = FastBlob::kMelStart, int max_slots = kMaxSlots) {
    const int real = lanes - 1;   // intervals per slot (the last lane of a group is the ghost)
    // interval of bin k = the filter whose rising part contains it = last row that is non-zero at k
    // and whose peak is at or after k; derive it from the matrix itself: rows non-zero at k are
    // {i-1, i} (or one of them); i is the larger one unless k sits on filter (i-1)'s falling side only.
    std::vector<int> idx(bin_limit, -1);
    for (int k = 0; k < bin_limit; ++k) {
        int lo = -1, hi = -1;
        for (int m = 0; m < n_mels; ++m)
            if (dense[static_cast<size_t>(m) * bins + k] != 0.0) { if (lo < 0) lo = m; hi = m; }
        if (lo < 0) continue;                 // bin used by no filter
        if (hi - lo > 1) return false;
        if (hi != lo) { idx[k] = hi; continue; }
        // a single filter m: k is on its rising side (interval m) or falling side (interval m+1)
        const double *row = &dense[static_cast<size_t>(lo) * bins];
        int peak = 0;
        for (int q = 1; q < bin_limit; ++q) if (row[q] > row[peak]) peak = q;
        idx[k] = (k <= peak) ? lo : lo + 1;
    }
    // intervals must be contiguous, non-decreasing runs
    int last = -1;
    for (int k = 0; k < bin_limit; ++k) {
        if (idx[k] < 0) continue;
        if (idx[k] < last) return false;
        last = idx[k];
    }
    const int n_int = n_mels + 1;
    std::vector<int> first(n_int, 0), cnt(n_int, 0);
    for (int k = 0; k < bin_limit; ++k) {
        if (idx[k] < 0) continue;
        if (cnt[idx[k]] == 0) first[idx[k]] = k;
        else if (first[idx[k]] + cnt[idx[k]] != k) return false;   // hole inside an interval
        ++cnt[idx[k]];
    }
    const int n_slots = (n_int + real - 1) / real;
    if (n_slots > max_slots || n_slots > kSlotCap) return false;
    slots = MelSlots{};
    slots.n_slots = n_slots;
    for (int s = 0; s < n_slots; ++s) {
        int L = 0;
        for (int j = 0; j < lanes; ++j) {
            const int i = s * real + j;
            if (i < n_int && cnt[i] > L) L = cnt[i];
        }
        slots.len[s] = L;
        slots.woff[s] = static_cast<int>(b.size());
        b.resize(b.size() + static_cast<size_t>(L) * lanes * 2, 0.0f);
        for (int j = 0; j < lanes; ++j) {
            const int i = s * real + j;
            int st = 0;
            if (i < n_int && cnt[i] > 0) {
                st = first[i];
                if (st + L > bin_limit) st = bin_limit - L;
                for (int r = 0; r < L; ++r) {
                    const int k = st + r;
                    if (idx[k] != i) continue;       // padding bins belong to a neighbour interval
                    const double rise = i < n_mels ? dense[static_cast<size_t>(i) * bins + k] : 0.0;
                    const double fall = i >= 1 ? dense[static_cast<size_t>(i - 1) * bins + k] : 0.0;
                    b[slots.woff[s] + (r * lanes + j) * 2] = static_cast<float>(scale * rise);
                    b[slots.woff[s] + (r * lanes + j) * 2 + 1] = static_cast<float>(scale * fall);
                }
            }
            std::memcpy(&b[starts_off + s * lanes + j], &st, sizeof(int));
        }
    }
    return true;
}

// Returns false if the geometry is outside the fused kernel's coverage.
// dense: the filterbank, [n_mels][201] (MelSpectrogram::new's default, or a caller's bank: melspec_create_with_filterbank)
inline bool build_fast_tables(const std::vector<double> &dense, int n_mels, FastTables &out, bool want_interval = false) {
    constexpr int N = 400, M = 200;
    const int n_slots = (n_mels + kMelJobs - 1) / kMelJobs;
    if (n_mels < 1 || n_slots > kMaxSlots) return false;
    std::vector<float> &b = out.blob;
    b.assign(FastBlob::kMelW, 0.0f);
    const std::vector<double> win = hann_window(N);
    for (int t = 0; t < 10; ++t)
        for (int n1 = 0; n1 < 20; ++n1)
            for (int c = 0; c < 2; ++c) b[FastBlob::kWin + t * FastBlob::kWinStride + 2 * n1 + c] = static_cast<float>(win[20 * n1 + 2 * t + c]);
    for (int t = 0; t < 10; ++t)
        for (int k1 = 0; k1 < 20; ++k1) {
            const double a = -2.0 * kPi * ((t * k1) % M) / M;
            b[FastBlob::kTw1 + t * FastBlob::kTw1Stride + 2 * k1] = static_cast<float>(std::cos(a));
            b[FastBlob::kTw1 + t * FastBlob::kTw1Stride + 2 * k1 + 1] = static_cast<float>(std::sin(a));
        }
    for (int n2 = 0; n2 < 10; ++n2) {
        const double a = -2.0 * kPi * n2 / 10.0;
        b[FastBlob::kMod + 2 * n2] = static_cast<float>(std::cos(a));
        b[FastBlob::kMod + 2 * n2 + 1] = static_cast<float>(std::sin(a));
    }
    for (int j = 0; j < kMelJobs; ++j)
        for (int q = 0; q < 10; ++q) {
            const double a = -2.0 * kPi * (j + 20 * q) / N;
            b[FastBlob::kTw2 + j * 20 + 2 * q] = static_cast<float>(std::cos(a));
            b[FastBlob::kTw2 + j * 20 + 2 * q + 1] = static_cast<float>(std::sin(a));
        }
    // MelSpectrogram::new: mel(sr, fft, n_mels, None, None, false, true)  (src/mel.rs:19-24);
    // bins >= n_fft/2 are zeroed by project_stft_log10 (src/mel.rs:155-163).
    const int bins = N / 2 + 1;
    const BandedFilterbank fb = band_filterbank(dense, n_mels, bins, M);
    out.nnz = fb.nnz;
    out.n_mels = n_mels;
    out.interval = false;
    if (want_interval && build_interval_mel(dense, n_mels, bins, M, b, out.slots)) {
        out.interval = true;
        while (b.size() % 4) b.push_back(0.0f);
        return true;
    }
    b.resize(FastBlob::kMelW);
    out.slots = MelSlots{};
    out.slots.n_slots = n_slots;
    for (int i = 0; i < n_slots; ++i) {
        int L = 0;
        for (int j = 0; j < kMelJobs; ++j) {
            const int m = i * kMelJobs + j;
            if (m < n_mels && fb.len[m] > L) L = fb.len[m];
        }
        out.slots.len[i] = L;
        out.slots.woff[i] = static_cast<int>(b.size());
        b.resize(b.size() + static_cast<size_t>(L) * kMelJobs, 0.0f);
        for (int j = 0; j < kMelJobs; ++j) {
            const int m = i * kMelJobs + j;
            int st = 0;
            if (m < n_mels && fb.len[m] > 0) {
                st = fb.start[m];
                if (st + L > M) st = M - L;   // keep every read inside bins [0,200)
                for (int r = 0; r < L; ++r)
                    b[out.slots.woff[i] + r * kMelJobs + j] =
                        static_cast<float>(dense[static_cast<size_t>(m) * bins + st + r]);
            }
            std::memcpy(&b[FastBlob::kMelStart + i * kMelJobs + j], &st, sizeof(int));
        }
    }
    while (b.size() % 4) b.push_back(0.0f);
    return true;
}

inline bool build_fast_tables(double sr, int n_mels, FastTables &out, bool want_interval = false) {
    return build_fast_tables(mel_filterbank(sr, 400, n_mels, -1.0, -1.0, false, true), n_mels, out, want_interval);
}

}  // namespace melspec

#include "whisper_wave_f64.hpp"
#include "whisper_six.hpp"
#include "whisper_six64.hpp"

namespace melspec {

// Tables of the six-frames-per-wave kernel (SixBlob in whisper_six.hpp).  false: filterbank outside its coverage
// (more than 9 slots of 9 intervals, or not a two-filters-per-bin bank).
// max_slots = kSixWideSlots: the fifteen-slot mel section of the f64 six-frame kernel at 81..134 mels (start bins [15][10], then weights)
inline bool build_six_tables(const std::vector<double> &dense, int n_mels, FastTables &out, int max_slots = kSixMaxSlots) {
    constexpr int N = 400, M = 200;
    if (n_mels < 1 || n_mels > kSixOwn * max_slots - 1) return false;
    std::vector<float> &b = out.blob;
    b.assign((SixBlob::kMelStart + max_slots * kSixLanes + 3) & ~3, 0.0f);
    const std::vector<double> win = hann_window(N);
    for (int t = 0; t < 10; ++t)
        for (int n1 = 0; n1 < 20; ++n1)
            for (int c = 0; c < 2; ++c) b[SixBlob::kWin + t * SixBlob::kWinStride + 2 * n1 + c] = static_cast<float>(win[20 * n1 + 2 * t + c]);
    for (int t = 0; t < 10; ++t)
        for (int k1 = 0; k1 < 20; ++k1) {
            const double a = -2.0 * kPi * ((t * k1) % M) / M;
            b[SixBlob::kTw1 + t * SixBlob::kTw1Stride + 2 * k1] = static_cast<float>(std::cos(a));
            b[SixBlob::kTw1 + t * SixBlob::kTw1Stride + 2 * k1 + 1] = static_cast<float>(std::sin(a));
        }
    for (int j = 0; j < kSixLanes; ++j)
        for (int s = 0; s < 11; ++s) {
            int k = j + 20 * s;                                   // lanes 1..9 (slot 10 unused)
            if (j == 0) k = s < 6 ? 20 * s : 10 + 20 * (s - 6);   // lane 0: residue 0, then residue 10
            const double a = -2.0 * kPi * k / N;
            b[SixBlob::kTw2 + j * SixBlob::kTw2Stride + 2 * s] = static_cast<float>(std::cos(a));
            b[SixBlob::kTw2 + j * SixBlob::kTw2Stride + 2 * s + 1] = static_cast<float>(std::sin(a));
        }
    const int bins = N / 2 + 1;
    out.n_mels = n_mels;
    out.nnz = 0;
    out.interval = build_interval_mel(dense, n_mels, bins, M, b, out.slots, kSixLanes, 0.25, SixBlob::kMelStart, max_slots);
    while (b.size() % 4) b.push_back(0.0f);
    return out.interval;
}

inline bool build_six_tables(double sr, int n_mels, FastTables &out) {
    return build_six_tables(mel_filterbank(sr, 400, n_mels, -1.0, -1.0, false, true), n_mels, out);
}

// Blob of the precise kernel: [PreciseBlob tables in f64][the mel section of an interval-scheme f32 blob].
struct PreciseTables {
    std::vector<uint32_t> blob;
    int mel_off_words = 0;
};

// power_split: the blob of the mel kernels (precise_phase2); false: the blob of the spectrum export (precise_phase2_spectrum)
inline bool build_precise_tables(const FastTables &ft, PreciseTables &out, bool power_split) {
    if (!ft.interval) return false;
    constexpr int N = 400, M = 200;
    std::vector<double> t(PreciseBlob::kCount, 0.0);
    const std::vector<double> win = hann_window(N);
    for (int i = 0; i < N; ++i) t[PreciseBlob::kWin + i] = win[i];
    for (int tt = 0; tt < 10; ++tt)
        for (int k1 = 0; k1 < 20; ++k1) {
            const double a = -2.0 * kPi * ((tt * k1) % M) / M;
            t[PreciseBlob::kTw1 + tt * PreciseBlob::kTw1Stride + 2 * k1] = std::cos(a);
            t[PreciseBlob::kTw1 + tt * PreciseBlob::kTw1Stride + 2 * k1 + 1] = std::sin(a);
        }
    for (int n2 = 0; n2 < 10; ++n2) {
        const double a = -2.0 * kPi * n2 / 10.0;
        t[PreciseBlob::kMod + 2 * n2] = std::cos(a);
        t[PreciseBlob::kMod + 2 * n2 + 1] = std::sin(a);
    }
    for (int j = 0; j < kMelJobs; ++j)
        for (int q = 0; q < 10; ++q) {
            const double a = -2.0 * kPi * (j + 20 * q) / N;
            t[PreciseBlob::kTw2 + j * PreciseBlob::kTw2Stride + 2 * q] = power_split ? 2.0 * std::sin(a) : std::cos(a);
            t[PreciseBlob::kTw2 + j * PreciseBlob::kTw2Stride + 2 * q + 1] = power_split ? 4.0 * std::cos(a) : std::sin(a);
        }
    const size_t t_words = t.size() * 2;
    const size_t mel_floats = ft.blob.size() - FastBlob::kMelStart;
    out.mel_off_words = static_cast<int>(t_words);
    out.blob.assign(t_words + mel_floats, 0u);
    std::memcpy(out.blob.data(), t.data(), t.size() * sizeof(double));
    std::memcpy(out.blob.data() + t_words, ft.blob.data() + FastBlob::kMelStart, mel_floats * sizeof(float));
    while (out.blob.size() % 4) out.blob.push_back(0u);
    return true;
}

// Blob of the f64 six-frame kernel (whisper_six64.hpp): [Six64Blob tables in f64][the mel section of the six-frame f32 blob].
struct Six64Tables {
    std::vector<uint32_t> blob;
    int mel_off_words = 0;
};

inline bool build_six64_tables(const FastTables &ft6, Six64Tables &out) {
    if (!ft6.interval) return false;
    constexpr int N = 400, M = 200;
    std::vector<double> t(Six64Blob::kCount, 0.0);
    const std::vector<double> win = hann_window(N);
    for (int tt = 0; tt < 10; ++tt)
        for (int n1 = 0; n1 < 20; ++n1)
            for (int c = 0; c < 2; ++c) t[Six64Blob::kWin + tt * Six64Blob::kWinStride + 2 * n1 + c] = win[20 * n1 + 2 * tt + c];
    for (int tt = 0; tt < 10; ++tt)
        for (int k1 = 0; k1 < 20; ++k1) {
            const double a = -2.0 * kPi * ((tt * k1) % M) / M;
            t[Six64Blob::kTw1 + tt * Six64Blob::kTw1Stride + 2 * k1] = std::cos(a);
            t[Six64Blob::kTw1 + tt * Six64Blob::kTw1Stride + 2 * k1 + 1] = std::sin(a);
        }
    for (int j = 0; j < kSixLanes; ++j)
        for (int s = 0; s < 11; ++s) {
            int k = j + 20 * s;                                   // lanes 1..9 (slot 10 unused)
            if (j == 0) k = s < 6 ? 20 * s : 10 + 20 * (s - 6);   // lane 0: residue 0, then residue 10
            const double a = -2.0 * kPi * k / N;
            t[Six64Blob::kTw2 + j * Six64Blob::kTw2Stride + 2 * s] = 2.0 * std::sin(a);      // the power split of precise_phase2 (whisper_wave_f64.hpp)
            t[Six64Blob::kTw2 + j * Six64Blob::kTw2Stride + 2 * s + 1] = 4.0 * std::cos(a);
        }
    const size_t t_words = t.size() * 2;
    const size_t mel_floats = ft6.blob.size() - SixBlob::kMelStart;
    out.mel_off_words = static_cast<int>(t_words);
    out.blob.assign(t_words + mel_floats, 0u);
    std::memcpy(out.blob.data(), t.data(), t.size() * sizeof(double));
    std::memcpy(out.blob.data() + t_words, ft6.blob.data() + SixBlob::kMelStart, mel_floats * sizeof(float));
    while (out.blob.size() % 4) out.blob.push_back(0u);
    return true;
}

}  // namespace melspec
