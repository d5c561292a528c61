// whisper_fast.hpp -- definitions shared by the fused n_fft=400 log-mel kernels (whisper_wave.hpp: five frames per
// wavefront; whisper_six.hpp: six): the layout of the constant table blob in LDS and the per-launch mel slot record.
//
// Pipeline per frame (reference: Spectrogram::compute_mel_spectrogram_cpu, src/stft.rs:119-138):
//   frame_windows  (src/stft.rs:147-169)  -> phase 1: PCM * Hann table
//   forward FFT    (src/stft.rs:105-111)  -> phases 1+2: real-400 as complex-200 = 20 x 10
//   |X|^2, bins<200 (src/mel.rs:155-163)  -> phase 2: Hermitian split, power to LDS
//   sparse mel+log10 (src/mel.rs:148-168) -> phase 3: interval sums from LDS, log2*log10(2)
//   per-frame norm (src/mel.rs:645-654)   -> phases 3/4: max via LDS, clamp, (x+4)/4
//
// FFT factorisation.  z[n] = x[2n] + i*x[2n+1], n in [0,200).  With n = 10*n1 + n2 and
// k = k1 + 20*k2:  Z[k] = sum_n2 W_10^{n2*k2} * ( W_200^{n2*k1} * sum_n1 W_20^{n1*k1} z[10*n1+n2] ).
//   phase 1: lane (frame, t=n2) does the 20-point DFT over n1, multiplies by W_200^{t*k1},
//            writes row k1 of the exchange buffer.
//   phase 2: lane (frame, j) owns residues a=j and b=(20-j)%20 (k = a+20q and its mirror
//            200-k = b+20(9-q)), does two 10-point DFTs and the real-FFT split
//              X[k] = (S - i*W_400^k*D)/2,  S = Z[k]+conj(Z[200-k]),  D = Z[k]-conj(Z[200-k]).
#pragma once
#include "device_fft.hpp"

namespace melspec {

constexpr int kMelJobs = 11;        // threads per frame in phases 2-4
constexpr int kFftJobs = 10;        // threads per frame in phase 1
constexpr int kMaxSlots = 12;       // mel slots per thread of the Whisper kernels (n_mels <= 131)
constexpr int kSlotCap = 17;        // capacity of MelSlots (the NeMo frontend uses up to 17 slots of 8)

// LDS layout of the constant table blob (float offsets).  Built by build_fast_tables().
struct FastBlob {
    // the 40 taps of lane t in the order it uses them, w[20*n1 + 2t + {0, 1}] at [t][2*n1 + {0, 1}]: ten 16-byte reads per unit
    // (taps in natural order came out as ds_read2_b64 pairs, half the LDS rate); 44 = 40 + 4 pad: ten rows in ten groups of four banks
    static constexpr int kWinStride = 44;
    static constexpr int kWin = 0;                       // [10][44] Hann
    static constexpr int kTw1Stride = 44;                // 20 complex + 4 pad (bank spread)
    static constexpr int kTw1 = 10 * kWinStride;         // [10][44] W_200^{t*k1}
    static constexpr int kMod = kTw1 + 10 * kTw1Stride;  // [10] complex W_10^{n2}
    static constexpr int kTw2 = kMod + 20;               // [11][10] complex W_400^{j+20q}
    static constexpr int kMelStart = kTw2 + kMelJobs * 20;   // [kMaxSlots*12] int bit patterns
    // interval scheme: (rise, fall) pairs [slot][r][12][2]
    static constexpr int kMelW = kMelStart + kMaxSlots * 12;
};

// Per-launch uniform parameters of the mel slots (scalar registers on the device).
struct MelSlots {
    int n_slots;               // ceil((n_mels + 1) / 11)
    int len[kSlotCap];         // padded span length of slot i
    int woff[kSlotCap];        // float offset of slot i's weights inside the blob
};

MS_DEV float fast_log2(float x) {
#if defined(__HIPCC__)
    return __builtin_amdgcn_logf(x);   // v_log_f32, 1 ulp; input is >= 1e-10 (normal)
#else
    return __builtin_log2f(x);
#endif
}

}  // namespace melspec
