// melspec_hip.hpp -- header-only C++ host mirror of the reference's plugin interface, over the C ABI
// (melspec_hip.h).  Same names, argument order and error split as the reference's Rust types:
//
//   melspec::HipMelSpectrogram(fft_size, hop_size, sampling_rate, n_mels)   CudaMelSpectrogram::new   src/cuda.rs:39-82
//   .compute_mel_spectrogram(samples) -> vector<vector<float>>              src/cuda.rs:88-101
//   melspec::Fbank(FbankConfig{}).compute(samples) -> Array2f               Fbank::{new,compute}      src/fbank.rs:94,141
//   melspec::mel(sr, n_fft, n_mels, f_min, f_max, htk, norm)                mel()                     src/mel.rs:547-589
//
// HipUnavailable == CudaError::Unavailable (construction; callers may skip), HipRuntimeError ==
// CudaError::Runtime (per call).  Objects are move-only and single-threaded, like `&mut self`.
#pragma once
#include <cstddef>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "melspec_hip.h"

namespace melspec {

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string &m) : std::runtime_error("[" + std::to_string(c) + "] " + m), code(c) {}
};
struct HipUnavailable : HipError { using HipError::HipError; };
struct HipRuntimeError : HipError { using HipError::HipError; };

namespace detail {
inline void check(int rc, bool constructing) {
    if (rc == MELSPEC_OK) return;
    const char *m = melspec_last_error();
    if (constructing || rc == MELSPEC_ERR_UNAVAILABLE) throw HipUnavailable(rc, m ? m : "");
    throw HipRuntimeError(rc, m ? m : "");
}
}  // namespace detail

class HipMelSpectrogram {
public:
    HipMelSpectrogram(std::size_t fft_size, std::size_t hop_size, double sampling_rate, std::size_t n_mels, int device = -1) {
        detail::check(melspec_create(&ctx_, device, static_cast<int>(fft_size), static_cast<int>(hop_size), sampling_rate,
                                     static_cast<int>(n_mels)), true);
    }
    // MelSpectrogram over SparseMelFilterbank::from_mel(sr, n_fft, n_mels, f_min, f_max, htk, norm) (src/mel.rs:73-87); f_min < 0 / f_max <= 0 == None
    static HipMelSpectrogram with_filterbank(std::size_t fft_size, std::size_t hop_size, double sampling_rate, std::size_t n_mels, double f_min,
                                             double f_max, bool htk, bool norm, int device = -1) {
        HipMelSpectrogram m;
        detail::check(melspec_create_with_filterbank(&m.ctx_, device, static_cast<int>(fft_size), static_cast<int>(hop_size), sampling_rate,
                                                     static_cast<int>(n_mels), f_min, f_max, htk ? 1 : 0, norm ? 1 : 0), true);
        return m;
    }
    // ... over SparseMelFilterbank::from_dense (src/mel.rs:48-71): filters = [n_mels][fft_size / 2 + 1], row-major
    static HipMelSpectrogram with_dense_filterbank(std::size_t fft_size, std::size_t hop_size, double sampling_rate, std::size_t n_mels,
                                                   const std::vector<double> &filters, int device = -1) {
        HipMelSpectrogram m;
        detail::check(melspec_create_with_dense_filterbank(&m.ctx_, device, static_cast<int>(fft_size), static_cast<int>(hop_size), sampling_rate,
                                                           static_cast<int>(n_mels), filters.data(), static_cast<int>(filters.size() / (n_mels ? n_mels : 1))), true);
        return m;
    }
    ~HipMelSpectrogram() { melspec_destroy(ctx_); }
    HipMelSpectrogram(HipMelSpectrogram &&o) noexcept : ctx_(std::exchange(o.ctx_, nullptr)) {}
    HipMelSpectrogram &operator=(HipMelSpectrogram &&o) noexcept {
        if (this != &o) { melspec_destroy(ctx_); ctx_ = std::exchange(o.ctx_, nullptr); }
        return *this;
    }
    HipMelSpectrogram(const HipMelSpectrogram &) = delete;
    HipMelSpectrogram &operator=(const HipMelSpectrogram &) = delete;

    std::size_t n_mels() const { return static_cast<std::size_t>(melspec_n_mels(ctx_)); }
    std::size_t num_frames(std::size_t n_samples) const { return melspec_num_frames(ctx_, n_samples); }

    // frames x n_mels, row-major, one inner vector per frame like the reference's Vec<Vec<f32>>
    std::vector<std::vector<float>> compute_mel_spectrogram(const std::vector<float> &samples) {
        const std::size_t frames = num_frames(samples.size()), nm = n_mels();
        std::vector<float> flat(frames * nm);
        std::size_t got = 0;
        detail::check(melspec_compute_host(ctx_, samples.data(), samples.size(), flat.data(), flat.size(), &got), false);
        std::vector<std::vector<float>> out(got);
        for (std::size_t f = 0; f < got; ++f) out[f].assign(flat.begin() + f * nm, flat.begin() + (f + 1) * nm);
        return out;
    }

    // device-resident batch of equal-length clips (additive surface)
    void compute_uniform_device(const float *d_pcm, std::uint64_t clip_stride, std::uint64_t clip_len, std::uint32_t n_clips,
                                float *d_out, void *stream = nullptr) {
        detail::check(melspec_compute_uniform_device(ctx_, d_pcm, clip_stride, clip_len, n_clips, d_out, stream), false);
    }
    void synchronize(void *stream = nullptr) { detail::check(melspec_synchronize(ctx_, stream), false); }
    // f64 window/FFT/power like the reference's CPU and CUDA paths (melspec_set_precise)
    void set_precise(bool on) { detail::check(melspec_set_precise(ctx_, on ? 1 : 0), false); }
    bool precise() const { return melspec_is_precise(ctx_) != 0; }
    // MELSPEC_PRECISION_AUTO (default) / _F64 / _F32
    void set_precision(int mode) { detail::check(melspec_set_precision(ctx_, mode), false); }
    int precision() const { return melspec_precision(ctx_); }
    // AUTO moves whole batches to the f64 kernel while most frames of the last finished batch needed f64; false pins the f32 kernel
    void set_auto_adaptive(bool on) { detail::check(melspec_set_auto_adaptive(ctx_, on ? 1 : 0), false); }
    bool auto_heavy() { int h = 0; double f = 0.0; detail::check(melspec_auto_state(ctx_, &h, &f), false); return h != 0; }
    // CudaMelSpectrogram::max_frames_per_batch (src/cuda.rs:84-86): frames per chunk of the host pipeline
    std::size_t max_frames_per_batch() const { return melspec_max_frames_per_batch(ctx_); }

    // additive: many host clips in one call through the chunked H2D / kernels / D2H pipeline -> [clip][frame][mel]
    std::vector<std::vector<std::vector<float>>> compute_batch(const std::vector<std::vector<float>> &clips) {
        std::vector<std::uint64_t> offs, lens;
        std::vector<float> flat;
        for (const auto &c : clips) { offs.push_back(flat.size()); lens.push_back(c.size()); flat.insert(flat.end(), c.begin(), c.end()); }
        const std::size_t nm = n_mels();
        std::size_t total = 0;
        for (const auto &c : clips) total += num_frames(c.size());
        std::vector<float> out(total * nm + 1);
        std::uint64_t got = 0;
        if (flat.empty()) flat.push_back(0.0f);
        detail::check(melspec_compute_batch_host(ctx_, flat.data(), offs.data(), lens.data(), static_cast<std::uint32_t>(clips.size()),
                                                 out.data(), nullptr, out.size(), &got), false);
        std::vector<std::vector<std::vector<float>>> res(clips.size());
        std::size_t cur = 0;
        for (std::size_t c = 0; c < clips.size(); ++c) {
            const std::size_t f = num_frames(clips[c].size());
            res[c].resize(f);
            for (std::size_t i = 0; i < f; ++i, cur += nm) res[c][i].assign(out.begin() + cur, out.begin() + cur + nm);
        }
        return res;
    }

    // Spectrogram::compute_all_cpu (src/stft.rs:89-115): [frame][fft_size] interleaved (re, im) doubles
    std::vector<std::vector<double>> compute_all(const std::vector<float> &samples) {
        const std::size_t frames = num_frames(samples.size()), bins = melspec_stft_bins(ctx_, 1);
        std::vector<double> flat(frames * bins * 2 + 2);
        std::size_t got = 0;
        detail::check(melspec_stft_host(ctx_, samples.data(), samples.size(), flat.data(), frames * bins, MELSPEC_STFT_F64, 1, &got), false);
        std::vector<std::vector<double>> out(got);
        for (std::size_t f = 0; f < got; ++f) out[f].assign(flat.begin() + f * bins * 2, flat.begin() + (f + 1) * bins * 2);
        return out;
    }
    // MelSpectrogram::add(&fft) (src/mel.rs:13-32) for every frame of compute_all's output (or any [frame][fft_size] (re, im) doubles)
    std::vector<std::vector<float>> mel_from_stft(const std::vector<std::vector<double>> &frames) {
        const std::size_t nm = static_cast<std::size_t>(melspec_n_mels(ctx_)), bins = melspec_stft_bins(ctx_, 1);
        std::vector<double> flat;
        flat.reserve(frames.size() * bins * 2);
        for (const auto &f : frames) flat.insert(flat.end(), f.begin(), f.end());
        std::vector<float> out(frames.size() * nm + 1);
        detail::check(melspec_mel_from_stft_host(ctx_, flat.data(), MELSPEC_STFT_F64, 1, frames.size(), out.data(), out.size()), false);
        std::vector<std::vector<float>> res(frames.size());
        for (std::size_t f = 0; f < frames.size(); ++f) res[f].assign(out.begin() + f * nm, out.begin() + (f + 1) * nm);
        return res;
    }
    melspec_ctx *raw() { return ctx_; }

private:
    HipMelSpectrogram() = default;          // the factories above
    melspec_ctx *ctx_ = nullptr;
};

struct FbankConfig : melspec_fbank_config {
    FbankConfig() { melspec_fbank_default_config(this); }   // FbankConfig::default, src/fbank.rs:46-64
};

struct Array2f {   // Array2<f32> (rows, cols), row-major
    std::size_t rows = 0, cols = 0;
    std::vector<float> data;
    float operator()(std::size_t r, std::size_t c) const { return data[r * cols + c]; }
};

class Fbank {
public:
    explicit Fbank(const FbankConfig &cfg = FbankConfig(), int device = -1) {
        detail::check(melspec_fbank_create(&fb_, device, &cfg), true);
    }
    ~Fbank() { melspec_fbank_destroy(fb_); }
    Fbank(const Fbank &) = delete;
    Fbank &operator=(const Fbank &) = delete;

    Array2f compute(const std::vector<float> &samples) {
        Array2f a;
        a.cols = static_cast<std::size_t>(melspec_fbank_num_mel_bins(fb_));
        a.rows = melspec_fbank_num_frames(fb_, samples.size());
        a.data.assign(a.rows * a.cols, 0.0f);
        std::size_t got = 0;
        detail::check(melspec_fbank_compute_host(fb_, samples.data(), samples.size(), a.data.data(), a.data.size(), &got), false);
        return a;
    }

    // additive: compute() for many clips in one call (melspec_fbank_compute_batch_host)
    std::vector<Array2f> compute_batch(const std::vector<std::vector<float>> &clips) {
        std::vector<std::uint64_t> offs, lens;
        std::vector<float> flat;
        std::size_t total = 0;
        const std::size_t nm = static_cast<std::size_t>(melspec_fbank_num_mel_bins(fb_));
        for (const auto &c : clips) {
            offs.push_back(flat.size()); lens.push_back(c.size());
            flat.insert(flat.end(), c.begin(), c.end());
            total += melspec_fbank_num_frames(fb_, c.size());
        }
        std::vector<float> out(total * nm + 1);
        std::uint64_t frames = 0;
        detail::check(melspec_fbank_compute_batch_host(fb_, flat.data(), offs.data(), lens.data(), static_cast<std::uint32_t>(clips.size()),
                                                       out.data(), nullptr, out.size(), &frames), false);
        std::vector<Array2f> res(clips.size());
        std::size_t cur = 0;
        for (std::size_t i = 0; i < clips.size(); ++i) {
            res[i].rows = melspec_fbank_num_frames(fb_, clips[i].size());
            res[i].cols = nm;
            res[i].data.assign(out.begin() + cur, out.begin() + cur + res[i].rows * nm);
            cur += res[i].rows * nm;
        }
        return res;
    }

private:
    melspec_fbank *fb_ = nullptr;
};

// src/quant.rs: quantize / dequantize / tga_8bit / parse_tga_8bit, same names and return shapes
struct QuantizationRange { float min, max; };

class TgaCodec {
public:
    explicit TgaCodec(int device = -1) { detail::check(melspec_tga_create(&q_, device), true); }
    ~TgaCodec() { melspec_tga_destroy(q_); }
    TgaCodec(const TgaCodec &) = delete;
    TgaCodec &operator=(const TgaCodec &) = delete;

    std::pair<std::vector<std::uint8_t>, QuantizationRange> quantize(const std::vector<float> &frame) {
        std::vector<std::uint8_t> out(frame.size());
        float r[2] = {0.0f, 0.0f};
        detail::check(melspec_quantize_host(q_, frame.data(), frame.size(), out.data(), r), false);
        return {std::move(out), QuantizationRange{r[0], r[1]}};
    }
    std::vector<float> dequantize(const std::vector<std::uint8_t> &data, const QuantizationRange &range) {
        std::vector<float> out(data.size());
        const float r[2] = {range.min, range.max};
        detail::check(melspec_dequantize_host(q_, data.data(), data.size(), r, out.data()), false);
        return out;
    }
    // one TGA per <= 65535-column chunk of the major-row-order [n_mels][width] image
    std::vector<std::vector<std::uint8_t>> tga_8bit(const std::vector<float> &data, std::size_t n_mels) {
        std::uint32_t n = 0;
        std::size_t stride = 0, last = 0;
        detail::check(melspec_tga_layout(static_cast<int>(n_mels), n_mels ? data.size() / n_mels : 0, &n, &stride, &last), false);
        std::vector<std::vector<std::uint8_t>> res;
        if (n == 0) return res;
        std::vector<std::uint8_t> flat(stride * (n - 1) + last);
        std::uint32_t got = 0;
        detail::check(melspec_tga_encode_host(q_, data.data(), data.size(), static_cast<int>(n_mels), flat.data(), flat.size(), &got), false);
        const std::size_t full = 26 + n_mels * 65535;
        for (std::uint32_t c = 0; c < n; ++c)
            res.emplace_back(flat.begin() + c * stride, flat.begin() + c * stride + (c + 1 < n ? full : last));
        return res;
    }
    std::vector<float> parse_tga_8bit(const std::vector<std::uint8_t> &blob) {
        std::vector<float> out(blob.size() > 26 ? blob.size() - 26 : 0);
        std::size_t n = 0;
        detail::check(melspec_tga_decode_host(q_, blob.data(), blob.size(), out.data(), out.size(), &n), false);
        out.resize(n);
        return out;
    }

private:
    melspec_tga *q_ = nullptr;
};

// One live stream with the reference's RingBuffer interface (src/rb.rs:18-121) over a 1-stream bank:
// add_frame() feeds samples, maybe_mel() hands out one (n_mels) column per completed hop.
class RingBuffer {
public:
    RingBuffer(HipMelSpectrogram &mel, std::size_t capacity = 16384) : n_mels_(mel.n_mels()), cap_(capacity) {
        detail::check(melspec_stream_create(&st_, mel.raw(), 1, static_cast<std::uint32_t>(capacity)), true);
    }
    ~RingBuffer() { melspec_stream_destroy(st_); }
    RingBuffer(const RingBuffer &) = delete;
    RingBuffer &operator=(const RingBuffer &) = delete;

    void add_frame(const std::vector<float> &samples) {
        pending_.insert(pending_.end(), samples.begin(), samples.end());
        if (pending_.size() > cap_) pending_.erase(pending_.begin(), pending_.begin() + (pending_.size() - cap_));   // src/rb.rs:60-68
    }
    void add(float sample) { add_frame(std::vector<float>(1, sample)); }
    std::optional<std::vector<float>> maybe_mel() {
        if (next_ == ready_.size() / (n_mels_ ? n_mels_ : 1) && !pending_.empty()) {
            const std::uint32_t id = 0, len = static_cast<std::uint32_t>(pending_.size());
            std::uint32_t frames = 0;
            ready_.assign(melspec_stream_frames_after(st_, 0, len) * n_mels_, 0.0f);
            next_ = 0;
            detail::check(melspec_stream_push_host(st_, &id, pending_.data(), &len, 1, ready_.data(), ready_.size(), &frames), false);
            ready_.resize(static_cast<std::size_t>(frames) * n_mels_);
            pending_.clear();
        }
        if (next_ * n_mels_ >= ready_.size()) return std::nullopt;
        std::vector<float> col(ready_.begin() + next_ * n_mels_, ready_.begin() + (next_ + 1) * n_mels_);
        ++next_;
        return col;
    }

private:
    melspec_stream *st_ = nullptr;
    std::size_t n_mels_, cap_, next_ = 0;
    std::vector<float> pending_, ready_;
};

// src/vad.rs: DetectionSettings, vad_boundaries -> EdgeInfo, vad_on
struct DetectionSettings : melspec_vad_settings {
    DetectionSettings() { melspec_vad_default_settings(this); }                       // 0.98, 11, 5, 2
    DetectionSettings(double min_energy_, int min_y_, int min_x_, int min_mel_) {
        min_energy = min_energy_; min_y = min_y_; min_x = min_x_; min_mel = min_mel_;
    }
};
struct EdgeInfo {
    std::vector<std::size_t> non_intersected_columns, intersected_columns;
    std::uint32_t longest_run = 0;
    const std::vector<std::size_t> &non_intersected() const { return non_intersected_columns; }
    const std::vector<std::size_t> &intersected() const { return intersected_columns; }
};
// one (n_mels x width) image, row-major (the reference concatenates its Array2 frames along the time axis)
inline EdgeInfo vad_boundaries(const std::vector<float> &image, std::size_t n_mels, const DetectionSettings &settings, int device = -1) {
    EdgeInfo e;
    const std::size_t width = n_mels ? image.size() / n_mels : 0;
    std::vector<std::uint8_t> mask(melspec_vad_mask_len(static_cast<int>(n_mels), width));
    detail::check(melspec_vad_boundaries_host(device, image.data(), static_cast<int>(n_mels), width, &settings, nullptr, mask.data(),
                                              &e.longest_run), false);
    for (std::size_t x = 0; x < mask.size(); ++x) (mask[x] ? e.intersected_columns : e.non_intersected_columns).push_back(x);
    return e;
}
inline bool vad_on(const EdgeInfo &e, std::size_t n) {          // src/vad.rs:229-254
    return n <= 1 ? e.intersected_columns.size() >= 2 : e.longest_run >= n;
}

// VoiceActivity (src/vad.rs:125-135) and the detector behind a live stream: VoiceActivityDetector::add_activity
// (src/vad.rs:155-205) fed by RingBuffer's frames, with both on the device -- one bank stream whose detector stage is on.
struct VoiceActivity {
    bool active;
    std::size_t frame_index, leading_active_columns, active_columns, window_columns;
    double confidence;
};
class StreamDetector {
public:
    StreamDetector(HipMelSpectrogram &mel, const DetectionSettings &settings, std::size_t max_chunk = 16384) : n_mels_(mel.n_mels()) {
        detail::check(melspec_stream_create(&st_, mel.raw(), 1, static_cast<std::uint32_t>(max_chunk)), true);
        const int rc = melspec_stream_enable_vad(st_, &settings);
        if (rc) { melspec_stream_destroy(st_); st_ = nullptr; detail::check(rc, true); }
    }
    ~StreamDetector() { melspec_stream_destroy(st_); }
    StreamDetector(const StreamDetector &) = delete;
    StreamDetector &operator=(const StreamDetector &) = delete;

    // Feeds samples (any length <= max_chunk); one entry per frame they complete: std::nullopt where add_activity returns None.
    // rows (optional) receives the mel rows of those frames, [frame][n_mels].
    std::vector<std::optional<VoiceActivity>> add_frame(const std::vector<float> &samples, std::vector<float> *rows = nullptr) {
        const std::uint32_t id = 0, len = static_cast<std::uint32_t>(samples.size());
        const std::size_t cap = melspec_stream_frames_after(st_, 0, len);
        const std::uint64_t first = melspec_stream_vad_frames(st_, 0);
        std::vector<float> out(cap * n_mels_);
        std::vector<melspec_vad_activity> acts(cap);
        std::uint32_t frames = 0;
        detail::check(melspec_stream_push_host_vad(st_, &id, samples.data(), &len, 1, out.data(), out.size(), &frames, acts.data(), acts.size()), false);
        std::vector<std::optional<VoiceActivity>> res(frames);
        for (std::uint32_t k = 0; k < frames; ++k) {
            const melspec_vad_activity &a = acts[k];
            if (!a.valid) continue;
            res[k] = VoiceActivity{a.active != 0, static_cast<std::size_t>(first + k), a.leading_active_columns, a.active_columns, a.window_columns,
                                   a.window_columns ? static_cast<double>(a.active_columns) / a.window_columns : 0.0};
        }
        if (rows) { out.resize(static_cast<std::size_t>(frames) * n_mels_); rows->swap(out); }
        return res;
    }

private:
    melspec_stream *st_ = nullptr;
    std::size_t n_mels_;
};

// dense [n_mels][n_fft/2+1] row-major; std::nullopt == None
inline std::vector<double> mel(double sr, std::size_t n_fft, std::size_t n_mels, std::optional<double> f_min = std::nullopt,
                               std::optional<double> f_max = std::nullopt, bool htk = false, bool norm = true) {
    std::vector<double> w(n_mels * (n_fft / 2 + 1));
    detail::check(melspec_mel_filterbank(sr, static_cast<int>(n_fft), static_cast<int>(n_mels), f_min.value_or(-1.0),
                                         f_max.value_or(-1.0), htk, norm, w.data()), false);
    return w;
}

inline double hz_to_mel(double frequency, bool htk = false) { return melspec_hz_to_mel(frequency, htk); }     // src/mel.rs:591-607
inline double mel_to_hz(double mel_value, bool htk = false) { return melspec_mel_to_hz(mel_value, htk); }      // src/mel.rs:609-625
inline std::vector<double> mel_frequencies(std::size_t n_mels, double fmin, double fmax, bool htk = false) {   // src/mel.rs:631-637
    std::vector<double> f(n_mels);
    detail::check(melspec_mel_frequencies(static_cast<int>(n_mels), fmin, fmax, htk, f.data()), false);
    return f;
}
inline std::vector<double> fft_frequencies(double sr, std::size_t n_fft) {                                     // src/mel.rs:639-643
    std::vector<double> f(n_fft / 2 + 1);
    detail::check(melspec_fft_frequencies(sr, static_cast<int>(n_fft), f.data()), false);
    return f;
}

}  // namespace melspec
