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

// Tables of the fused 512-point kernels: `window` = the 400 taps applied to the frame, `dense` = the
// filterbank [n_mels][257].  Returns false when the filterbank lacks the two-filters-per-bin structure or
// needs more than `max_slots` slots of 15 intervals.
template <class T>
inline bool build_fused512_tables(const std::vector<double> &window, const std::vector<double> &dense, int n_mels,
                                  double scale, int max_slots, FbankFastTables &out, int bin_limit = 257, bool power_split = false) {
    constexpr int N = 512;
    const int FL = static_cast<int>(window.size());       // 400 taps (Kaldi, NeMo) or 512 (Whisper flavour)
    if (n_mels < 1 || n_mels > kFbOwn * max_slots - 1 || (FL != 400 && FL != 512)) return false;
    std::vector<T> t(FbankBlob::kTCount, T(0));
    for (int i = 0; i < FL; ++i) t[FbankBlob::kWin + i] = static_cast<T>(window[i]);
    for (int n2 = 0; n2 < 16; ++n2)
        for (int k1 = 0; k1 < 16; ++k1) {
            const double a = -2.0 * kPi * ((n2 * k1) % 256) / 256.0;
            t[FbankBlob::kTw1 + n2 * FbankBlob::kTw1Stride + 2 * k1] = static_cast<T>(std::cos(a));
            t[FbankBlob::kTw1 + n2 * FbankBlob::kTw1Stride + 2 * k1 + 1] = static_cast<T>(std::sin(a));
        }
    for (int r = 0; r < 16; ++r)
        for (int q = 0; q < 9; ++q) {
            const double a = -2.0 * kPi * (r + 16 * q) / N;
            // power_split (the Whisper flavour, fb_phase2_split<.., FAST>): (2 sin, 4 cos) of the same angle
            t[FbankBlob::kTw2 + r * FbankBlob::kTw2Stride + 2 * q] = static_cast<T>(power_split ? 2.0 * std::sin(a) : std::cos(a));
            t[FbankBlob::kTw2 + r * FbankBlob::kTw2Stride + 2 * q + 1] = static_cast<T>(power_split ? 4.0 * std::cos(a) : std::sin(a));
        }
    const int bins = N / 2 + 1;
    std::vector<float> mel(FbankBlob::kMelW, 0.0f);
    if (!build_interval_mel(dense, n_mels, bins, bin_limit, mel, out.slots, kFbLanes, scale, FbankBlob::kMelStart, max_slots))
        return false;
    while (mel.size() % 4) mel.push_back(0.0f);
    const size_t t_words = t.size() * sizeof(T) / 4;
    out.mel_off_words = static_cast<int>(t_words);
    out.blob.assign(t_words + mel.size(), 0u);
    std::memcpy(out.blob.data(), t.data(), t.size() * sizeof(T));
    std::memcpy(out.blob.data() + t_words, mel.data(), mel.size() * sizeof(float));
    out.f64 = sizeof(T) == 8;
    return true;
}

// Kaldi fbank, default geometry (400-sample frames, 512-point FFT): Povey window (src/fbank.rs:98-105),
// un-normalised Kaldi triangles (src/fbank.rs:253-301).  Phase 2 stores 4*|X|^2 (or 2*|X|), so the
// weights carry the 1/4 (1/2).
template <class T>
inline bool build_fbank_fast_tables(double sample_rate, int num_mel_bins, double low_freq, double high_freq,
                                    bool use_power, FbankFastTables &out) {
    const std::vector<double> dense = kaldi_mel_filterbank(sample_rate, 512, num_mel_bins, low_freq, high_freq);
    return build_fused512_tables<T>(povey_window(400), dense, num_mel_bins, use_power ? 0.25 : 0.5, kFbSlots, out);
}

// NeMo/Parakeet frontend (BatchLogMelSpectrogram::new, src/mel.rs:248-280): symmetric Hann(400) computed in
// f32 exactly like centered_hann_window_f32 (src/mel.rs:708-719), Slaney/HTK mel() with explicit f_min/f_max,
// weights rounded to f32 like project_power_f32 (src/mel.rs:139-145).
template <class T>
inline bool build_blm_fast_tables(int sample_rate, int n_mels, double f_min, double f_max, bool htk, bool norm,
                                  FbankFastTables &out) {
    std::vector<double> win(400);
    const float pi_f32 = 3.14159265358979323846f;
    for (int i = 0; i < 400; ++i) {
        const float phase = (2.0f * pi_f32 * static_cast<float>(i)) / (400.0f - 1.0f);
        win[i] = static_cast<double>(0.5f - (0.5f * std::cos(phase)));
    }
    std::vector<double> dense = mel_filterbank(static_cast<double>(sample_rate), 512, n_mels, f_min > 0.0 ? f_min : -1.0, f_max, htk, norm);
    for (double &w : dense) w = static_cast<double>(static_cast<float>(w));
    return build_fused512_tables<T>(win, dense, n_mels, 0.25, kBlmSlots, out);
}

// Whisper flavour at n_fft = 512: periodic Hann(512) (src/stft.rs:141-145), Slaney mel() over 257 bins of which
// project_stft_log10 uses those below n_fft/2 = 256 (src/mel.rs:155-163).
template <class T>
inline bool build_whisper512_tables(const std::vector<double> &dense /* [n_mels][257] */, int n_mels, FbankFastTables &out) {
    // f64: the table of the direct power form (fb_phase2_split<.., FAST>); f32: plain W_512^k -- the direct form is a difference of large
    // numbers in the weaker bin of a pair, eps x the power of its MIRROR bin: fine at 2^-53, and in f32 it cost the bare kernel three decades
    // (7e-2 on speech; profiles/r06_guard512.txt has the error by depth under the frame maximum before and after)
    return build_fused512_tables<T>(hann_window(512), dense, n_mels, 0.25, kBlmSlots, out, 256, sizeof(T) == 8);
}
template <class T>
inline bool build_whisper512_tables(double sample_rate, int n_mels, FbankFastTables &out) {
    return build_whisper512_tables<T>(mel_filterbank(sample_rate, 512, n_mels, -1.0, -1.0, false, true), n_mels, out);
}

}  // namespace melspec
