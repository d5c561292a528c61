// C++ twin of the reference's own GPU-vs-CPU test (src/cuda.rs:489-545) and README shape tests
// (tests/readme_examples.rs:12-29), written against include/melspec_hip.hpp.  Expected values come
// from the CPU oracle library, linked only into this test binary.
//   build: g++ -std=c++17 -Iinclude tests/cpp/test_host_mirror.cpp -Lmel_spec_amd -lmelspec_hip -Loracle -lmelspec_oracle
#include <cmath>
#include <cstdio>
#include <vector>

#include "melspec_hip.hpp"

extern "C" long oracle_compute_mel_spectrogram_cpu(const float *, long, int, int, int, double, float *);

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
    melspec::Fbank fb;
    const melspec::Array2f feats = fb.compute(std::vector<float>(16000, 0.0f));
    if (feats.cols != 80 || feats.rows != 98) { std::puts("FAIL: fbank shape"); return 1; }
    const std::vector<double> w = melspec::mel(sr, 400, 80);
    if (w.size() != 80 * 201) { std::puts("FAIL: mel shape"); return 1; }
    std::puts("OK");
    return 0;
}
