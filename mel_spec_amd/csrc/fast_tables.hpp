// fast_tables.hpp -- host builder of the constant blob the n_fft=400 kernel keeps in LDS
// (layout: FastBlob in whisper_fast.hpp).  All values are computed in f64 and rounded once.
#pragma once
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
};

// Returns false if the geometry is outside the fused kernel's coverage.
inline bool build_fast_tables(double sr, int n_mels, FastTables &out) {
    constexpr int N = 400, M = 200;
    const int n_slots = (n_mels + kMelJobs - 1) / kMelJobs;
    if (n_mels < 1 || n_slots > kMaxSlots) return false;
    std::vector<float> &b = out.blob;
    b.assign(FastBlob::kMelW, 0.0f);
    const std::vector<double> win = hann_window(N);
    for (int i = 0; i < N; ++i) b[FastBlob::kWin + i] = static_cast<float>(win[i]);
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
    const std::vector<double> dense = mel_filterbank(sr, N, n_mels, -1.0, -1.0, false, true);
    const BandedFilterbank fb = band_filterbank(dense, n_mels, bins, M);
    out.nnz = fb.nnz;
    out.n_mels = n_mels;
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

}  // namespace melspec
