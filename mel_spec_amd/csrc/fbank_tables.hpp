// fbank_tables.hpp -- host builder of the constant blob of the fused Kaldi-fbank kernel
// (layout: FbankBlob in fbank_wave.hpp); f64 on the host, stored as T (f64 for the parity build).
#pragma once
#include <cstdint>

#include "fast_tables.hpp"
#include "fbank_wave.hpp"

namespace melspec {

struct FbankFastTables {
    std::vector<uint32_t> blob;   // [T-typed tables][mel section: starts (int) + weight pairs (f32)], 16-byte multiple
    int mel_off_words = 0;        // offset of the mel section in 32-bit words
    MelSlots slots{};             // woff[] are float offsets from the mel section base
    bool f64 = true;
};

// Default Kaldi geometry only (400-sample frames, 512-point FFT).  Returns false when the filterbank
// lacks the two-filters-per-bin structure or needs more than kFbSlots slots.
template <class T>
inline bool build_fbank_fast_tables(double sample_rate, int num_mel_bins, double low_freq, double high_freq,
                                    bool use_power, FbankFastTables &out) {
    constexpr int N = 512, FL = 400;
    if (num_mel_bins < 1 || num_mel_bins > 8 * kFbSlots - 1) return false;
    std::vector<T> t(FbankBlob::kTCount, T(0));
    const std::vector<double> win = povey_window(FL);              // src/fbank.rs:98-105
    for (int i = 0; i < FL; ++i) t[FbankBlob::kWin + i] = static_cast<T>(win[i]);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double a = -2.0 * kPi * ((n2 * k1) % 256) / 256.0;
            t[FbankBlob::kTw1 + n2 * FbankBlob::kTw1Stride + 2 * k1] = static_cast<T>(std::cos(a));
            t[FbankBlob::kTw1 + n2 * FbankBlob::kTw1Stride + 2 * k1 + 1] = static_cast<T>(std::sin(a));
        }
    for (int n2 = 0; n2 < 16; ++n2) {
        const double a = -2.0 * kPi * n2 / 16.0;
        t[FbankBlob::kMod + 2 * n2] = static_cast<T>(std::cos(a));
        t[FbankBlob::kMod + 2 * n2 + 1] = static_cast<T>(std::sin(a));
    }
    for (int j = 0; j < 9; ++j)
        for (int q = 0; q < 16; ++q) {
            const double a = -2.0 * kPi * (j + 16 * q) / N;
            t[FbankBlob::kTw2 + j * FbankBlob::kTw2Stride + 2 * q] = static_cast<T>(std::cos(a));
            t[FbankBlob::kTw2 + j * FbankBlob::kTw2Stride + 2 * q + 1] = static_cast<T>(std::sin(a));
        }
    const int bins = N / 2 + 1;
    const std::vector<double> dense = kaldi_mel_filterbank(sample_rate, N, num_mel_bins, low_freq, high_freq);
    // phase 2 stores 4*|X|^2 (or 2*|X|): the weights carry the 1/4 (1/2)
    std::vector<float> mel(FbankBlob::kMelW, 0.0f);
    if (!build_interval_mel(dense, num_mel_bins, bins, bins, mel, out.slots, kFbLanes, use_power ? 0.25 : 0.5,
                            FbankBlob::kMelStart))
        return false;
    if (out.slots.n_slots > kFbSlots) return false;
    while (mel.size() % 4) mel.push_back(0.0f);
    const size_t t_words = t.size() * sizeof(T) / 4;
    out.mel_off_words = static_cast<int>(t_words);
    out.blob.assign(t_words + mel.size(), 0u);
    std::memcpy(out.blob.data(), t.data(), t.size() * sizeof(T));
    std::memcpy(out.blob.data() + t_words, mel.data(), mel.size() * sizeof(float));
    out.f64 = sizeof(T) == 8;
    return true;
}

}  // namespace melspec
