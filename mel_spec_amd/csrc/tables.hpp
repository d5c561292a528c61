// tables.hpp -- host-side (f64) builders for every constant table the kernels use.
//
// These mirror the reference's table constructors so that the device tables are the
// f32 roundings of exactly the numbers the CPU path uses:
//   hann_window            src/stft.rs:141-145
//   mel()/hz_to_mel/...    src/mel.rs:547-643
//   SparseMelFilterbank    src/mel.rs:48-71     (contiguous non-zero run per row)
//   povey window           src/fbank.rs:98-105
//   kaldi_mel_filterbank   src/fbank.rs:253-313
// Pure C++17, no HIP: also compiled into the CPU-side tests.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace melspec {

constexpr double kPi = 3.14159265358979323846264338327950288;

inline std::vector<double> hann_window(int n) {
    std::vector<double> w(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) w[i] = 0.5 * (1.0 - std::cos((2.0 * kPi * i) / n));
    return w;
}

inline std::vector<double> povey_window(int n) {
    std::vector<double> w(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        const double a = 2.0 * kPi * i / (n - 1);
        w[i] = std::pow(0.5 - 0.5 * std::cos(a), 0.85);
    }
    return w;
}

inline double hz_to_mel(double f, bool htk) {
    if (htk) return 2595.0 * std::log10(1.0 + f / 700.0);
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}

inline double mel_to_hz(double m, bool htk) {
    if (htk) return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// Dense Slaney/HTK filterbank, row-major [n_mels][n_fft/2+1].
inline std::vector<double> mel_filterbank(double sr, int n_fft, int n_mels, double f_min, double f_max,
                                          bool htk, bool norm) {
    const int bins = n_fft / 2 + 1;
    if (f_min < 0.0) f_min = 0.0;
    if (f_max <= 0.0) f_max = sr / 2.0;
    std::vector<double> edges(static_cast<size_t>(n_mels) + 2);
    const double lo = hz_to_mel(f_min, htk), hi = hz_to_mel(f_max, htk);
    const double step = (hi - lo) / (n_mels + 1);   // linspace over n_mels+2 points
    for (int i = 0; i < n_mels + 2; ++i) edges[i] = mel_to_hz(lo + step * i, htk);
    std::vector<double> w(static_cast<size_t>(n_mels) * bins);
    const double fstep = sr / n_fft;
    for (int m = 0; m < n_mels; ++m) {
        const double up_span = edges[m + 1] - edges[m], down_span = edges[m + 2] - edges[m + 1];
        const double scale = norm ? 2.0 / (edges[m + 2] - edges[m]) : 1.0;
        for (int b = 0; b < bins; ++b) {
            const double f = fstep * b;
            double rise = -(edges[m] - f) / up_span;
            double fall = (edges[m + 2] - f) / down_span;
            rise = std::fmin(std::fmax(rise, 0.0), 1.0);
            fall = std::fmin(std::fmax(fall, 0.0), 1.0);
            w[static_cast<size_t>(m) * bins + b] = std::fmin(rise, fall) * scale;
        }
    }
    return w;
}

inline std::vector<double> kaldi_mel_filterbank(double sr, int fft_size, int n_bins_mel, double low, double high) {
    const int nb = fft_size / 2 + 1;
    auto to_mel = [](double hz) { return 1127.0 * std::log(1.0 + hz / 700.0); };
    auto to_hz = [](double mel) { return 700.0 * (std::exp(mel / 1127.0) - 1.0); };
    const double ml = to_mel(low), mh = to_mel(high);
    std::vector<double> hz(static_cast<size_t>(n_bins_mel) + 2);
    for (int i = 0; i <= n_bins_mel + 1; ++i) hz[i] = to_hz(ml + (mh - ml) * i / (n_bins_mel + 1));
    std::vector<double> w(static_cast<size_t>(n_bins_mel) * nb, 0.0);
    for (int m = 0; m < n_bins_mel; ++m) {
        const double l = hz[m], c = hz[m + 1], r = hz[m + 2];
        if (c <= l || r <= c) continue;
        for (int b = 0; b < nb; ++b) {
            const double f = b * sr / fft_size;
            if (f > l && f <= c) w[static_cast<size_t>(m) * nb + b] = (f - l) / (c - l);
            else if (f > c && f < r) w[static_cast<size_t>(m) * nb + b] = (r - f) / (r - c);
        }
    }
    return w;
}

// Sparse view of a dense filterbank: per row the [first,last] non-zero span.  Triangular
// filters have contiguous support, but a span may still contain exact zeros (e.g. a
// clamped edge); those stay in the span with weight 0, which changes nothing in a sum.
struct BandedFilterbank {
    int n_mels = 0, bins = 0, nnz = 0, max_len = 0;
    std::vector<int> start, len;     // per mel row
    std::vector<double> w;           // concatenated spans
    std::vector<int> offset;         // start of row's span in w
};

inline BandedFilterbank band_filterbank(const std::vector<double> &dense, int n_mels, int bins, int bin_limit) {
    // bin_limit: bins >= bin_limit contribute nothing (the mel path zeroes bin >= n_fft/2,
    // src/mel.rs:155-163); pass `bins` to keep all.
    BandedFilterbank f;
    f.n_mels = n_mels; f.bins = bins;
    f.start.assign(n_mels, 0); f.len.assign(n_mels, 0); f.offset.assign(n_mels, 0);
    for (int m = 0; m < n_mels; ++m) {
        int first = -1, last = -1;
        for (int b = 0; b < bins && b < bin_limit; ++b)
            if (dense[static_cast<size_t>(m) * bins + b] != 0.0) { if (first < 0) first = b; last = b; }
        f.offset[m] = static_cast<int>(f.w.size());
        if (first >= 0) {
            f.start[m] = first; f.len[m] = last - first + 1;
            for (int b = first; b <= last; ++b) {
                const double v = dense[static_cast<size_t>(m) * bins + b];
                f.w.push_back(v); f.nnz += (v != 0.0);
            }
            if (f.len[m] > f.max_len) f.max_len = f.len[m];
        }
    }
    return f;
}

}  // namespace melspec
