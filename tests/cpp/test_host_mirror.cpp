// C++ twin of the reference's own GPU-vs-CPU test (src/cuda.rs:489-545) and README shape tests
// (tests/readme_examples.rs:12-29), written against include/melspec_hip.hpp.  Expected values come
// from the CPU oracle library, linked only into this test binary.
//   build: g++ -std=c++17 -Iinclude tests/cpp/test_host_mirror.cpp -Lmel_spec_amd -lmelspec_hip -Loracle -lmelspec_oracle
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "melspec_hip.hpp"

extern "C" long oracle_compute_mel_spectrogram_cpu(const float *, long, int, int, int, double, float *);
extern "C" long oracle_stream_mel(const float *, long, int, int, int, double, float *, long);
extern "C" void oracle_quantize(const float *, long, unsigned char *, float *);
extern "C" long oracle_tga_8bit_data(const float *, long, int, unsigned char *);
extern "C" long oracle_vad_boundaries(const float *, int, long, int, int, double, unsigned char *, unsigned char *);

int main() {
    const double sr = 16000.0;
    std::vector<float> samples(16000);
    for (int i = 0; i < 16000; ++i) {
        const float t = static_cast<float>(i) / 16000.0f, tp = 2.0f * 3.14159265358979323846f;
        samples[i] = 0.6f * std::sin(tp * 220.0f * t) + 0.25f * std::sin(tp * 440.0f * t) + 0.10f * std::sin(tp * 880.0f * t) +
                     0.05f * std::sin(tp * 1760.0f * t);
    }
    try {
        melspec::HipMelSpectrogram bad(0, 160, sr, 80);
        std::puts("FAIL: zero fft_size accepted");
        return 1;
    } catch (const melspec::HipUnavailable &) {
    }
    std::vector<std::vector<float>> gpu;
    try {
        melspec::HipMelSpectrogram hip(400, 160, sr, 80);
        gpu = hip.compute_mel_spectrogram(samples);
        if (!hip.compute_mel_spectrogram(std::vector<float>(399)).empty()) { std::puts("FAIL: short input"); return 1; }
    } catch (const melspec::HipUnavailable &e) {
        std::printf("SKIP: %s\n", e.what());   // like src/cuda.rs:512-518
        return 77;
    }
    std::vector<float> cpu(98 * 80);
    if (oracle_compute_mel_spectrogram_cpu(samples.data(), 16000, 400, 160, 80, sr, cpu.data()) != 98 || gpu.size() != 98) {
        std::puts("FAIL: frame count");
        return 1;
    }
    float max_delta = 0.0f, sum = 0.0f;
    for (size_t f = 0; f < 98; ++f)
        for (size_t m = 0; m < 80; ++m) {
            const float d = std::fabs(gpu[f][m] - cpu[f * 80 + m]);
            max_delta = std::fmax(max_delta, d);
            sum += d;
        }
    std::printf("max delta %.3g mean delta %.3g\n", max_delta, sum / (98 * 80));
    if (!(max_delta <= 1e-4f)) { std::puts("FAIL: tolerance"); return 1; }
    {
        // the reference's split API: compute_all's frames through MelSpectrogram::add == the fused pipeline
        melspec::HipMelSpectrogram hip2(400, 160, sr, 80);
        const auto spec = hip2.compute_all(samples);
        const auto staged = hip2.mel_from_stft(spec);
        float md = 0.0f;
        for (size_t f = 0; f < 98; ++f)
            for (size_t m = 0; m < 80; ++m) md = std::fmax(md, std::fabs(staged[f][m] - cpu[f * 80 + m]));
        std::printf("mel stage on STFT frames: max delta %.3g\n", md);
        if (staged.size() != 98 || !(md <= 2e-6f)) { std::puts("FAIL: mel_from_stft"); return 1; }
    }
    melspec::Fbank fb;
    const melspec::Array2f feats = fb.compute(std::vector<float>(16000, 0.0f));
    if (feats.cols != 80 || feats.rows != 98) { std::puts("FAIL: fbank shape"); return 1; }
    {
        // many clips in one call == one call per clip
        std::vector<std::vector<float>> clips = {std::vector<float>(samples.begin(), samples.begin() + 9000), std::vector<float>(399, 0.5f),
                                                 std::vector<float>(samples.begin() + 100, samples.begin() + 4100)};
        const auto many = fb.compute_batch(clips);
        if (many.size() != 3 || many[1].rows != 0) { std::puts("FAIL: fbank batch shape"); return 1; }
        for (size_t i = 0; i < 3; ++i) {
            const melspec::Array2f one = fb.compute(clips[i]);
            if (one.rows != many[i].rows || one.data != many[i].data) { std::puts("FAIL: fbank batch values"); return 1; }
        }
    }
    const std::vector<double> w = melspec::mel(sr, 400, 80);
    if (w.size() != 80 * 201) { std::puts("FAIL: mel shape"); return 1; }
    // precise build, streaming mirror and the quantiser through the same header
    {
        melspec::HipMelSpectrogram hip(400, 160, sr, 80);
        hip.set_precise(true);
        if (!hip.precise()) { std::puts("FAIL: precise flag"); return 1; }
        const auto pg = hip.compute_mel_spectrogram(samples);
        float md = 0.0f;
        for (size_t f = 0; f < 98; ++f)
            for (size_t m = 0; m < 80; ++m) md = std::fmax(md, std::fabs(pg[f][m] - cpu[f * 80 + m]));
        std::printf("precise max delta %.3g\n", md);
        if (!(md <= 2e-6f)) { std::puts("FAIL: precise tolerance"); return 1; }
        hip.set_precise(false);
        if (hip.precision() != MELSPEC_PRECISION_AUTO || hip.max_frames_per_batch() == 0) { std::puts("FAIL: precision / max_frames_per_batch"); return 1; }
        // a caller's filterbank: from_mel with the defaults == the default context; an HTK bank differs and is finite
        {
            auto same = melspec::HipMelSpectrogram::with_filterbank(400, 160, sr, 80, -1.0, -1.0, false, true);
            same.set_auto_adaptive(false); hip.set_auto_adaptive(false);
            const auto a = same.compute_mel_spectrogram(samples), b = hip.compute_mel_spectrogram(samples);
            for (size_t f = 0; f < 98; ++f)
                for (size_t m = 0; m < 80; ++m)
                    if (a[f][m] != b[f][m]) { std::puts("FAIL: with_filterbank(defaults) != default bank"); return 1; }
            auto htk = melspec::HipMelSpectrogram::with_filterbank(400, 160, sr, 64, 50.0, 7000.0, true, false);
            const auto h = htk.compute_mel_spectrogram(samples);
            if (h.size() != 98 || h[0].size() != 64 || !std::isfinite(h[5][7]) || same.auto_heavy()) { std::puts("FAIL: HTK bank"); return 1; }
            hip.set_auto_adaptive(true);
        }
        // additive batch call and the STFT export through the same header
        const auto batch = hip.compute_batch({samples, std::vector<float>(samples.begin(), samples.begin() + 4000), std::vector<float>(399, 0.0f)});
        if (batch.size() != 3 || batch[0].size() != 98 || batch[1].size() != 23 || !batch[2].empty()) { std::puts("FAIL: batch shape"); return 1; }
        float bd = 0.0f;
        for (size_t f = 0; f < 98; ++f)
            for (size_t m = 0; m < 80; ++m) bd = std::fmax(bd, std::fabs(batch[0][f][m] - cpu[f * 80 + m]));
        if (!(bd <= 1e-4f)) { std::puts("FAIL: batch tolerance"); return 1; }
        const auto spec = hip.compute_all(samples);
        if (spec.size() != 98 || spec[0].size() != 800) { std::puts("FAIL: stft shape"); return 1; }
        for (size_t k = 1; k < 200; ++k)            // real input: X[400 - k] = conj X[k]
            if (std::fabs(spec[5][2 * k] - spec[5][2 * (400 - k)]) > 1e-9 || std::fabs(spec[5][2 * k + 1] + spec[5][2 * (400 - k) + 1]) > 1e-9) { std::puts("FAIL: stft symmetry"); return 1; }

        melspec::RingBuffer rb(hip, 16000);
        std::vector<float> want(100 * 80);
        const long nf = oracle_stream_mel(samples.data(), 16000, 400, 160, 80, sr, want.data(), 100);
        long got = 0;
        float sd = 0.0f;
        for (int p = 0; p < 16000; p += 1000) {
            rb.add_frame(std::vector<float>(samples.begin() + p, samples.begin() + p + 1000));
            while (auto col = rb.maybe_mel()) {
                if (got < nf)
                    for (size_t m = 0; m < 80; ++m) sd = std::fmax(sd, std::fabs((*col)[m] - want[got * 80 + m]));
                ++got;
            }
        }
        std::printf("streaming frames %ld (want %ld) max delta %.3g\n", got, nf, sd);
        if (got != nf || !(sd <= 1e-4f)) { std::puts("FAIL: streaming"); return 1; }

        melspec::TgaCodec codec;
        std::vector<float> img(80 * 98);
        for (size_t f = 0; f < 98; ++f)
            for (size_t m = 0; m < 80; ++m) img[m * 98 + f] = cpu[f * 80 + m];
        const auto blobs = codec.tga_8bit(img, 80);
        std::vector<unsigned char> ref(26 + img.size());
        oracle_tga_8bit_data(img.data(), static_cast<long>(img.size()), 80, ref.data());
        if (blobs.size() != 1 || blobs[0] != ref) { std::puts("FAIL: tga bytes"); return 1; }
        const auto back = codec.parse_tga_8bit(blobs[0]);
        const auto qr = codec.quantize(img);
        if (back.size() != img.size() || !std::equal(qr.first.begin(), qr.first.end(), ref.begin() + 26)) { std::puts("FAIL: quantize"); return 1; }
        // VAD masks of the same image: identical to the oracle's
        std::vector<unsigned char> raw(96), sm(96);
        const melspec::DetectionSettings st(0.5, 3, 5, 0);
        const long nmask = oracle_vad_boundaries(img.data(), 80, 98, st.min_mel, st.min_y, st.min_energy, raw.data(), sm.data());
        const melspec::EdgeInfo e = melspec::vad_boundaries(img, 80, st);
        size_t set = 0;
        for (long x = 0; x < nmask; ++x) set += sm[x];
        if (nmask != 96 || e.intersected().size() != set || e.intersected().size() + e.non_intersected().size() != 96) { std::puts("FAIL: vad mask"); return 1; }
        for (size_t x : e.intersected()) if (!sm[x]) { std::puts("FAIL: vad column"); return 1; }
        // the detector behind a live stream: None for the first min_x - 1 frames, then one record per frame with consecutive indices;
        // a detector with min_y = 0 calls every column intersected (src/vad.rs:381-384)
        melspec::StreamDetector det(hip, melspec::DetectionSettings(1.0, 0, 5, 0), 16000);
        std::vector<float> rows;
        size_t seen = 0;
        for (int p = 0; p < 16000; p += 4000) {
            const auto acts = det.add_frame(std::vector<float>(samples.begin() + p, samples.begin() + p + 4000), &rows);
            if (rows.size() != acts.size() * 80) { std::puts("FAIL: detector rows"); return 1; }
            for (const auto &a : acts) {
                if (seen < 4 ? a.has_value() : !(a && a->active && a->frame_index == seen && a->window_columns == 3 && a->active_columns == 3 &&
                                                 a->leading_active_columns == 3 && a->confidence == 1.0)) { std::puts("FAIL: detector record"); return 1; }
                ++seen;
            }
        }
        if (seen != static_cast<size_t>(nf)) { std::puts("FAIL: detector frames"); return 1; }
    }
    std::puts("OK");
    return 0;
}
