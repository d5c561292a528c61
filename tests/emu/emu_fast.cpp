// emu_fast.cpp -- TEST-ONLY host emulation of the fused n_fft=400 kernel.
//
// Runs the exact per-thread phase functions of mel_spec_amd/csrc/whisper_fast.hpp on the
// host (one phase at a time over all thread ids == a barrier between phases), with a float
// array standing in for LDS.  It checks the kernel's index algebra and f32 error budget
// against the f64 oracle without a GPU.  Never linked into the product library.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../mel_spec_amd/csrc/fast_tables.hpp"

using namespace melspec;

template <int FPB, int NSLOTS>
static long long run(const float *pcm, long long n, int hop, int n_mels, double sr, float *out) {
    using L = FastLayout<FPB>;
    FastTables T;
    if (!build_fast_tables(sr, n_mels, T)) return -1;
    if (n < 400) return 0;
    const long long frames = (n - 400) / hop + 1;
    std::vector<float> regA(L::region_a(hop)), regB(L::region_b()), pmax(L::region_max());
    constexpr int NT = 256;
    static_assert(L::kP2Threads <= NT, "block too small");
    std::vector<float> vals(static_cast<size_t>(NT) * NSLOTS);
    for (long long f0 = 0; f0 < frames; f0 += FPB) {
        const int nv = static_cast<int>(std::min<long long>(FPB, frames - f0));
        const int need = (nv - 1) * hop + 400;
        // poison LDS so that any read of an unwritten word shows up
        std::fill(regA.begin(), regA.end(), 1.0e30f);
        std::fill(regB.begin(), regB.end(), 1.0e30f);
        for (int i = 0; i < need; ++i) regA[i] = pcm[f0 * hop + i];
        for (int tid = 0; tid < NT; ++tid) fast_phase1<FPB>(tid, nv, hop, T.blob.data(), regA.data(), regB.data());
        std::fill(regA.begin(), regA.end(), 1.0e30f);   // region A is re-used for the power rows
        for (int tid = 0; tid < NT; ++tid) fast_phase2<FPB>(tid, nv, T.blob.data(), regB.data(), regA.data());
        for (int tid = 0; tid < NT; ++tid)
            fast_phase3<FPB, NSLOTS>(tid, nv, n_mels, T.slots, T.blob.data(), regA.data(), pmax.data(),
                                     *reinterpret_cast<float(*)[NSLOTS]>(&vals[static_cast<size_t>(tid) * NSLOTS]));
        for (int tid = 0; tid < NT; ++tid)
            fast_phase4<FPB, NSLOTS>(tid, nv, n_mels, pmax.data(),
                                     *reinterpret_cast<const float(*)[NSLOTS]>(&vals[static_cast<size_t>(tid) * NSLOTS]),
                                     out + f0 * n_mels);
    }
    return frames;
}

extern "C" long long emu_whisper_fast(const float *pcm, long long n, int hop, int n_mels, double sr, float *out) {
    if (n_mels <= 88) return run<23, 8>(pcm, n, hop, n_mels, sr, out);
    return run<23, 12>(pcm, n, hop, n_mels, sr, out);
}

// power spectrum only (debug): |X[k]|^2, k in [0,200], for the first frame of pcm
extern "C" int emu_fast_power(const float *pcm, double sr, float *pw201) {
    using L = FastLayout<1>;
    FastTables T;
    build_fast_tables(sr, 80, T);
    std::vector<float> regA(L::region_a(160)), regB(L::region_b());
    for (int i = 0; i < 400; ++i) regA[i] = pcm[i];
    for (int tid = 0; tid < 16; ++tid) fast_phase1<1>(tid, 1, 160, T.blob.data(), regA.data(), regB.data());
    for (int tid = 0; tid < 16; ++tid) fast_phase2<1>(tid, 1, T.blob.data(), regB.data(), regA.data());
    std::memcpy(pw201, regA.data(), sizeof(float) * 201);
    return 0;
}

// small DFT checks
extern "C" void emu_small_fft(int n, float *interleaved) {
    if (n == 10) { cf x[10]; std::memcpy(x, interleaved, sizeof x); fft10(x); std::memcpy(interleaved, x, sizeof x); }
    if (n == 20) { cf x[20]; std::memcpy(x, interleaved, sizeof x); fft20(x); std::memcpy(interleaved, x, sizeof x); }
    if (n == 8)  { cf x[8];  std::memcpy(x, interleaved, sizeof x); fft8(x);  std::memcpy(interleaved, x, sizeof x); }
    if (n == 16) { cf x[16]; std::memcpy(x, interleaved, sizeof x); fft16(x); std::memcpy(interleaved, x, sizeof x); }
}
