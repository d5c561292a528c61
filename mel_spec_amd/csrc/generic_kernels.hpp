// generic_kernels.hpp -- the f64 kernels of every other geometry: generic_frame_kernel / generic_stft_kernel (one frame per workgroup),
// pow2_frame_kernel (power-of-two frame sizes on wave-owned frames, pow2_wave.hpp) and the stand-alone mel stage.
#pragma once
#include "kernels_common.hpp"
#include "pow2_wave.hpp"

namespace melspec {

// The same for any n_fft: one frame per workgroup, direct f64 DFT of bins 0..n_fft/2 from an LDS twiddle table (the
// arithmetic of generic_frame_kernel below), the upper half mirrored for the full layout.
// Mixed-radix FFT of n complex points in LDS for the generic kernels (n_fft = 2^a 3^b 5^c that is not a power of two: 320, 400, 480,
// 800, 1200 ...): Stockham auto-sort passes, one per factor (4, 2, 3 or 5), ping-pong between x and y (2n doubles each), the twiddle
// table tw[2j] = cos, tw[2j+1] = -sin of 2 pi j / n.  Pass for radix P, current length len, stride st:
//   y[q + st (P j + r)] = W_len^{j r} * sum_k x[q + st (j + m k)] W_P^{k r},   m = len / P, q < st, j < m.
// Every thread of the workgroup calls it; returns the buffer that holds the result in natural order.
template <int NT, int P>
__device__ __forceinline__ void lds_fft_pass(int n, int len, int st, const double *tw, const double *x, double *y, int tid) {
    const int m = len / P, wstep = n / len, pstep = n / P;
    for (int b = tid; b < m * st; b += NT) {
        const int j = b / st, q = b - j * st;
        double ar[P], ai[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int at = q + st * (j + m * k);
            ar[k] = x[2 * at]; ai[k] = x[2 * at + 1];
        }
#pragma unroll
        for (int r = 0; r < P; ++r) {
            double vr = ar[0], vi = ai[0];
#pragma unroll
            for (int k = 1; k < P; ++k) {
                const int t = ((k * r) % P) * pstep;           // W_P^{k r}
                const double c = tw[2 * t], sn = tw[2 * t + 1];
                vr += ar[k] * c - ai[k] * sn;
                vi += ar[k] * sn + ai[k] * c;
            }
            const int t = static_cast<int>((static_cast<long long>(j) * r * wstep) % n);      // W_len^{j r}
            const double c = tw[2 * t], sn = tw[2 * t + 1];
            const int to = q + st * (P * j + r);
            y[2 * to] = vr * c - vi * sn;
            y[2 * to + 1] = vr * sn + vi * c;
        }
    }
}
template <int NT>
__device__ __forceinline__ double *lds_fft_mixed(const FftPlan &plan, int n, const double *tw, double *x, double *y, int tid) {
    int len = n, st = 1;
    for (int pass = 0; pass < plan.n_rad; ++pass) {
        const int P = static_cast<int>((plan.packed >> (4 * pass)) & 15ull);
        __syncthreads();
        switch (P) {
            case 2: lds_fft_pass<NT, 2>(n, len, st, tw, x, y, tid); break;
            case 3: lds_fft_pass<NT, 3>(n, len, st, tw, x, y, tid); break;
            case 4: lds_fft_pass<NT, 4>(n, len, st, tw, x, y, tid); break;
            default: lds_fft_pass<NT, 5>(n, len, st, tw, x, y, tid); break;
        }
        len /= P; st *= P;
        double *tmp = x; x = y; y = tmp;
    }
    __syncthreads();
    return x;
}

struct GenericStftParams {
    BatchDesc b;             // units == frames
    int n_fft, hop, bins, words_per_frame, f64;
    int fft_log2;            // log2(n_fft) for a power-of-two n_fft >= 8 (in-LDS FFT as in generic_frame_kernel), else 0
    FftPlan plan;            // n_rad > 0: mixed-radix FFT (lds_fft_mixed) for 2-3-5-smooth n_fft; both zero: direct DFT
    const double *d_win;     // [n_fft]
    const double *d_tw;      // [n_fft] interleaved (cos, -sin) of 2*pi*j/n_fft
};

template <int NT>
__global__ __launch_bounds__(NT) void generic_stft_kernel(const GenericStftParams p) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    double *tw = ldsd;                       // 2*n_fft
    double *xw = tw + 2 * p.n_fft;           // n_fft
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * p.n_fft; i += NT) tw[i] = p.d_tw[i];
    const uint64_t n_units = batch_n_units(p.b);
    for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const UnitLoc loc = locate_unit(p.b, unit);
        const float *x = loc.pcm + loc.unit * (uint64_t)p.hop;
        __syncthreads();
        const int mbits = p.fft_log2 - 1;
        const bool mixed = p.plan.n_rad > 0;
        for (int i = tid; i < p.n_fft; i += NT) {
            // FFT form: sample i of the real frame is component (i & 1) of complex point i >> 1, stored bit-reversed; mixed-radix
            // form: complex point i with a zero imaginary part
            const int at = p.fft_log2 ? static_cast<int>(2 * (mbits > 0 ? (__brev(static_cast<unsigned>(i >> 1)) >> (32 - mbits)) : 0u)) + (i & 1) : (mixed ? 2 * i : i);
            xw[at] = (double)x[i] * p.d_win[i];                                        // src/stft.rs:160-165
            if (mixed) xw[at + 1] = 0.0;
        }
        const int M = p.n_fft >> 1;
        const double *res = xw;
        if (mixed) res = lds_fft_mixed<NT>(p.plan, p.n_fft, tw, xw, xw + 2 * p.n_fft, tid);
        if (p.fft_log2) {
            for (int len = 2; len <= M; len <<= 1) {
                __syncthreads();
                const int half = len >> 1, tstep = p.n_fft / len;
                for (int b = tid; b < (M >> 1); b += NT) {
                    const int g = b / half, j = b - g * half;
                    const int i0 = g * len + j, i1 = i0 + half;
                    const double c = tw[2 * (j * tstep)], sn = tw[2 * (j * tstep) + 1];
                    const double ur = xw[2 * i0], ui = xw[2 * i0 + 1];
                    const double xr = xw[2 * i1], xi = xw[2 * i1 + 1];
                    const double vr = xr * c - xi * sn, vi = xr * sn + xi * c;
                    xw[2 * i0] = ur + vr; xw[2 * i0 + 1] = ui + vi;
                    xw[2 * i1] = ur - vr; xw[2 * i1 + 1] = ui - vi;
                }
            }
        }
        __syncthreads();
        float *o = loc.out + loc.unit * (uint64_t)p.words_per_frame;
        for (int k = tid; k <= p.n_fft / 2; k += NT) {
            double re = 0.0, im = 0.0;
            if (mixed) {
                re = res[2 * k]; im = res[2 * k + 1];
            } else if (p.fft_log2) {
                const int ka = k == M ? 0 : k, kb = (M - k) & (M - 1);
                const double ar = xw[2 * ka], ai = xw[2 * ka + 1];
                const double br = xw[2 * kb], bi = -xw[2 * kb + 1];
                const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
                const double orr = 0.5 * (ai - bi), oi = -0.5 * (ar - br);
                const double c = tw[2 * k], sn = tw[2 * k + 1];
                re = er + (orr * c - oi * sn);
                im = ei + (orr * sn + oi * c);
            } else {
                int idx = 0;
                for (int n = 0; n < p.n_fft; ++n) {
                    re += xw[n] * tw[2 * idx];
                    im += xw[n] * tw[2 * idx + 1];
                    idx += k;
                    if (idx >= p.n_fft) idx -= p.n_fft;
                }
            }
            const int mk = p.n_fft - k;
            const bool mirror = p.bins == p.n_fft && k > 0 && mk > k;
            if (p.f64) {
                double *od = reinterpret_cast<double *>(o);
                od[2 * k] = re; od[2 * k + 1] = im;
                if (mirror) { od[2 * mk] = re; od[2 * mk + 1] = -im; }
            } else {
                o[2 * k] = (float)re; o[2 * k + 1] = (float)im;
                if (mirror) { o[2 * mk] = (float)re; o[2 * mk + 1] = (float)-im; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Generic kernel: any n_fft / hop / n_mels, Whisper or Kaldi-fbank flavour, one frame per
// workgroup iteration, direct DFT in f64 from an LDS twiddle table.  It follows the
// reference's f64 arithmetic step by step (src/stft.rs:119-138, src/fbank.rs:160-222) and
// exists for coverage and as the on-device cross-check of the fused f32 kernels; it is not
// a throughput path.
// ------------------------------------------------------------------------------------
struct GenericParams {
    BatchDesc b;           // units == frames
    int n_fft;             // DFT length
    int frame_len;         // non-zero samples per frame (== n_fft for Whisper)
    int hop;
    int n_bins;            // bins whose power is needed: n_fft/2 (Whisper) or n_fft/2+1 (fbank)
    int n_mels;
    int fbank;             // 0: Whisper log10 + per-frame norm; 1: Kaldi fbank; 2: NeMo BatchLogMelSpectrogram (src/mel.rs:321-385)
    int use_log, use_power;
    double preemph, floor_v;   // NeMo: preemph = the f32 coefficient, floor_v = log_zero_guard
    long long clip_len;    // NeMo (uniform batches): samples per clip
    int pad;               // NeMo: n_fft / 2 when centred (zero padding either side, src/mel.rs:685-694), else 0
    int fft_log2;          // log2(n_fft) when n_fft is a power of two >= 8: the transform is an in-LDS radix-2 FFT; 0: not
    FftPlan plan;          // n_rad > 0: mixed-radix in-LDS FFT (2-3-5-smooth n_fft that is not a power of two); both zero: direct DFT
    const double *d_win;   // [frame_len]
    const double *d_tw;    // [n_fft] interleaved (cos, -sin) of 2*pi*j/n_fft
    const int *d_mstart;   // [n_mels]
    const int *d_mlen;     // [n_mels]
    const int *d_moff;     // [n_mels] offset into d_mw
    const double *d_mw;    // concatenated spans
    int mw_count;          // doubles in d_mw
    // pow2_frame_kernel's view of the same bank: n_jobs jobs of eight consecutive weights of one mel (the last job of a band padded with
    // zeros), d_jw[2 * ((q / 2) * n_jobs + job) + (q & 1)] weight q of a job, d_job[job] = first bin | mel << 12 | count << 20 (count = 1..8 real entries)
    const double *d_jw;
    const int *d_job;
    int n_jobs;
};

template <int NT>
__device__ __forceinline__ double block_reduce(double v, double *red, bool is_max) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const double o = red[tid + s];
            red[tid] = is_max ? (red[tid] > o ? red[tid] : o) : (red[tid] + o);
        }
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

template <int NT>
__global__ __launch_bounds__(NT) void generic_frame_kernel(const GenericParams p) {
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    double *tw = ldsd;                       // 2*n_fft
    double *xw = tw + 2 * p.n_fft;           // frame_len (direct DFT) or n_fft (FFT: n_fft/2 complex points, bit-reversed)
    const bool mixed = p.plan.n_rad > 0;                              // mixed-radix FFT: xw = two buffers of n_fft complex points
    double *pw = xw + (mixed ? 4 * p.n_fft : (p.fft_log2 ? p.n_fft : p.frame_len));           // n_bins
    // FFT form (power-of-two n_fft): the real frame as n_fft/2 complex points z[n] = x[2n] + i x[2n+1], stored at the bit-reversed
    // index for the in-place decimation-in-time passes below; sample i goes to slot(i)
    const int mbits = p.fft_log2 - 1;
    auto slot = [&](int i) -> int {
        if (mixed) return 2 * i;
        if (!p.fft_log2) return i;
        const unsigned r = mbits > 0 ? (__brev(static_cast<unsigned>(i >> 1)) >> (32 - mbits)) : 0u;
        return static_cast<int>(2 * r) + (i & 1);
    };
    double *mv = pw + p.n_bins;              // n_mels
    double *red = mv + p.n_mels;             // NT
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * p.n_fft; i += NT) tw[i] = p.d_tw[i];

    const uint64_t n_units = batch_n_units(p.b);
    for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const UnitLoc loc = locate_unit(p.b, unit);
        if (loc.unit >= loc.frames) {       // zero column of a padded layout (uniform batches only)
            float *z = p.b.mel_major ? loc.out + loc.unit : loc.out + loc.unit * (uint64_t)p.n_mels;
            const uint64_t zstep = p.b.mel_major ? p.b.out_width : 1;
            for (int m = tid; m < p.n_mels; m += NT) z[m * zstep] = 0.0f;
            continue;
        }
        const uint64_t start = loc.unit * (uint64_t)p.hop;
        const float *x = loc.pcm + start;
        __syncthreads();
        if (!p.fbank) {
            // frame_windows: x[start+i] as f64 * window[i]   (src/stft.rs:160-165)
            for (int i = tid; i < p.frame_len; i += NT) xw[slot(i)] = (double)x[i] * p.d_win[i];
        } else if (p.fbank == 2) {
            // whole-clip pre-emphasis in f32 with the reference's two roundings (src/mel.rs:696-706), zero centre padding, window
            const float coeff = (float)p.preemph;
            for (int i = tid; i < p.frame_len; i += NT) {
                const long long sidx = (long long)start + i - p.pad;
                float v = 0.0f;
                if (sidx >= 0 && sidx < p.clip_len) {
                    v = loc.pcm[sidx];
                    if (coeff != 0.0f && sidx > 0) v = v - f32_mul_rn(coeff, loc.pcm[sidx - 1]);
                }
                xw[slot(i)] = (double)v * p.d_win[i];
            }
        } else {
            // DC removal, pre-emphasis, Povey window   (src/fbank.rs:164-190)
            double part = 0.0;
            for (int i = tid; i < p.frame_len; i += NT) part += (double)x[i];
            const double mean = block_reduce<NT>(part, red, false) / (double)p.frame_len;
            for (int i = tid; i < p.frame_len; i += NT) {
                double v = (double)x[i] - mean;
                if (p.preemph > 0.0) {
                    if (i > 0) v -= p.preemph * ((double)x[i - 1] - mean);
                    else if (start > 0) v -= p.preemph * ((double)*(x - 1) - mean);
                }
                xw[slot(i)] = v * p.d_win[i];
            }
        }
        if (mixed) {
            // imaginary parts and the zero padding, then one Stockham pass per factor of n_fft (lds_fft_mixed)
            for (int i = tid; i < p.n_fft; i += NT) {
                xw[2 * i + 1] = 0.0;
                if (i >= p.frame_len) xw[2 * i] = 0.0;
            }
            const double *res = lds_fft_mixed<NT>(p.plan, p.n_fft, tw, xw, xw + 2 * p.n_fft, tid);
            for (int k = tid; k < p.n_bins; k += NT) {
                const double re = res[2 * k], im = res[2 * k + 1];
                const double ns = re * re + im * im;
                pw[k] = (p.fbank && !p.use_power) ? sqrt(ns) : ns;
            }
        } else if (p.fft_log2) {
            // zero padding up to n_fft (frame_len < n_fft: Kaldi's 400 of 512), then log2(n_fft/2) radix-2 passes over the n_fft/2
            // complex points and the real-FFT split X[k] = E[k] + W_N^k O[k] -- O(N log N) instead of the O(N^2) direct form below
            for (int i = p.frame_len + tid; i < p.n_fft; i += NT) xw[slot(i)] = 0.0;
            const int M = p.n_fft >> 1;
            for (int len = 2; len <= M; len <<= 1) {
                __syncthreads();
                const int half = len >> 1, tstep = p.n_fft / len;
                for (int b = tid; b < (M >> 1); b += NT) {
                    const int g = b / half, j = b - g * half;
                    const int i0 = g * len + j, i1 = i0 + half;
                    const double c = tw[2 * (j * tstep)], sn = tw[2 * (j * tstep) + 1];     // W_len^j = W_N^{j N / len}
                    const double ur = xw[2 * i0], ui = xw[2 * i0 + 1];
                    const double xr = xw[2 * i1], xi = xw[2 * i1 + 1];
                    const double vr = xr * c - xi * sn, vi = xr * sn + xi * c;
                    xw[2 * i0] = ur + vr; xw[2 * i0 + 1] = ui + vi;
                    xw[2 * i1] = ur - vr; xw[2 * i1 + 1] = ui - vi;
                }
            }
            __syncthreads();
            for (int k = tid; k < p.n_bins; k += NT) {
                const int ka = k == M ? 0 : k, kb = (M - k) & (M - 1);        // Z[M] = Z[0]; partner Z[M - k]
                const double ar = xw[2 * ka], ai = xw[2 * ka + 1];
                const double br = xw[2 * kb], bi = -xw[2 * kb + 1];           // conj
                const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);     // E = (A + B) / 2
                const double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);     // O = -i (A - B) / 2 = (di, -dr)
                const double c = tw[2 * k], sn = tw[2 * k + 1];
                const double orr = di, oi = -dr;
                const double re = er + (orr * c - oi * sn), im = ei + (orr * sn + oi * c);
                const double ns = re * re + im * im;
                pw[k] = (p.fbank && !p.use_power) ? sqrt(ns) : ns;
            }
        } else {
            __syncthreads();
            for (int k = tid; k < p.n_bins; k += NT) {
                double re = 0.0, im = 0.0;
                int idx = 0;
                for (int n = 0; n < p.frame_len; ++n) {
                    const double c = tw[2 * idx], s = tw[2 * idx + 1];
                    re += xw[n] * c;
                    im += xw[n] * s;
                    idx += k;
                    if (idx >= p.n_fft) idx -= p.n_fft;
                }
                const double ns = re * re + im * im;
                pw[k] = (p.fbank && !p.use_power) ? sqrt(ns) : ns;
            }
        }
        __syncthreads();
        double mx = -1.0e300;
        for (int m = tid; m < p.n_mels; m += NT) {
            const int st = p.d_mstart[m], len = p.d_mlen[m];
            const double *w = p.d_mw + p.d_moff[m];
            double e = 0.0;
            for (int r = 0; r < len; ++r) e += w[r] * pw[st + r];
            double v;
            if (!p.fbank) {
                v = log10(e > 1e-10 ? e : 1e-10);          // src/mel.rs:166
            } else if (p.fbank == 2) {
                v = log(e + p.floor_v);                    // src/mel.rs:365-368
            } else {
                v = e > p.floor_v ? e : p.floor_v;           // src/fbank.rs:210-218
                if (p.use_log) v = log(v);
            }
            mv[m] = v;
            mx = mx > v ? mx : v;
        }
        const uint64_t width = p.b.d_unit_prefix == nullptr ? p.b.out_width : loc.frames;
        float *o = p.b.mel_major ? loc.out + loc.unit : loc.out + loc.unit * (uint64_t)p.n_mels;
        const uint64_t ostep = p.b.mel_major ? width : 1;
        if (!p.fbank) {
            const double lo = block_reduce<NT>(mx, red, true) - 8.0;   // src/mel.rs:645-654
            for (int m = tid; m < p.n_mels; m += NT) {
                const double v = mv[m] > lo ? mv[m] : lo;
                o[m * ostep] = (float)((v + 4.0) / 4.0);
            }
        } else if (p.fbank == 2) {
            for (int m = tid; m < p.n_mels; m += NT) o[m * ostep] = (float)mv[m];      // feature-major rows (src/mel.rs:366)
        } else {
            for (int m = tid; m < p.n_mels; m += NT) o[m] = (float)mv[m];
        }
    }
}

// ------------------------------------------------------------------------------------
// pow2_frame_kernel: the power-of-two frame sizes on wave-owned frames (pow2_wave.hpp).  Same parameters, flavours, tables and
// results contract as generic_frame_kernel; units == frames.  LDS (doubles): [tw: W_N^q, q < M][WAVES x FW x frame region].
// ------------------------------------------------------------------------------------

// The samples a lane needs for one frame, as they come from memory: pairs (x[2n], x[2n + 1]) of its P complex points and, for the
// flavours with pre-emphasis, the sample in front of each pair.  Loaded one frame AHEAD of their use (the next frame's loads are in
// flight while this frame's transform runs): at two or three waves per SIMD nothing else hides a 1-2 us HBM round trip.
template <int P, int FLAVOR> struct Pow2Raw {
    f2 pair[P];
    float before[FLAVOR == 0 ? 1 : P];
};

template <int LOGM, int FLAVOR>
__global__ __launch_bounds__(Pow2Shape<LOGM>::kMaxWaves * 64) void pow2_frame_kernel(const GenericParams p) {
    using S = Pow2Shape<LOGM>;
    constexpr int M = S::M, LF = S::LF, FW = S::FW, P = S::P;
    constexpr bool kAhead = (P == 8 && !(FLAVOR == 1 && LF < 64 && !MS_POW2_AHEAD_KS)) || (P == 16 && MS_POW2_AHEAD16 && !(S::kHalves && !MS_POW2_AHEADH));   // the next frame's samples are loaded while this one is transformed
    constexpr bool kWinLds = M <= MS_POW2_WINLDS;
    extern __shared__ __attribute__((aligned(16))) double ldsd[];
    const int tid = threadIdx.x, n_threads = blockDim.x, n_waves = n_threads >> 6;     // the host picks the waves per workgroup (LDS)
    const Pow2Lds at = pow2_lds<LOGM>(p.n_jobs, p.n_mels, n_waves);
    double *tw = ldsd + at.tw;                           // W_N^q, q <= M / 2 (the split's twiddles)
    double *lwin = ldsd + at.win;                        // 2 * M: the window, zero from frame_len on (M <= 256)
    double *t2 = ldsd + at.t2, *t3 = ldsd + at.t3;       // the twiddles of passes 2 and 3, [r - 1][k]
    double *ljw = ldsd + at.jw;                          // the banded filterbank as jobs of eight weights (GenericParams::d_jw), then the
    int *ljob = reinterpret_cast<int *>(ldsd + at.job);  //   job records {first bin | mel << 12 | count << 20}
    for (int i = tid; i < M + 2; i += n_threads) tw[i] = p.d_tw[i];
    if (kWinLds) for (int i = tid; i < 2 * M; i += n_threads) lwin[i] = p.d_win[i];
    for (int i = tid; i < S::kT2; i += n_threads) stc(t2 + 2 * i, pow2_table_entry(p.d_tw, M, 8, S::R1, i));
    constexpr bool kHalves = S::kHalves;
    for (int i = tid; i < S::kT3; i += n_threads) stc(t3 + 2 * i, pow2_table_entry(p.d_tw, M, kHalves ? 8 : (S::R3 > 1 ? S::R3 : 2), kHalves ? 64 : S::R1 * 8, i));
    double *tc = ldsd + at.tc;                           // kHalves: W_M^k = W_N^{2k}, k < M / 2
    for (int i = tid; i < S::kTc; i += n_threads) stc(tc + 2 * i, pow2_root(p.d_tw, 2 * i, M));
    for (int i = tid; i < 8 * p.n_jobs; i += n_threads) ljw[i] = p.d_jw[i];
    for (int i = tid; i < p.n_jobs; i += n_threads) ljob[i] = p.d_job[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int fs = lane / LF, l = lane - fs * LF;        // frame slot of the wave, lane of the frame
    double *z = ldsd + at.frames + (wave * FW + fs) * at.frame_stride;
    double *pw = z + at.pw + pow2_pw_shift<LOGM>(fs);    // [M + 1]
    double *acc = z + at.acc;                            // [n_mels] band energies of the frame
    constexpr bool kPwAlias = M >= MS_POW2_PWALIAS;
    if (!kPwAlias && l < 8) pw[M + 1 + l] = 0.0;         // what the last job of the top band reads past the row
    const int last = p.frame_len - 1;

    // the twiddles of pass 2 depend on the lane only: W_{8 R1}^{k r}, k = l mod R1 (kept in registers; pass 3's come from the LDS table)
    constexpr bool kTw2Reg = (P == 8 && MS_POW2_TW2REG) || kHalves;
    cpx<double> tw2[kTw2Reg ? 7 : 1];
    // complex point r of the lane: l + r LF, or (kHalves) point l + 64 (r / 2) of the even (r even) / odd half: 2 (l + 64 (r / 2)) + (r & 1)
    auto pt = [&](int r) { return kHalves ? 2 * (l + 64 * (r >> 1)) + (r & 1) : l + r * LF; };
    if (kHalves) {
        const int k2 = l & 7;
#pragma unroll
        for (int r = 1; r < 8; ++r) tw2[r - 1] = pow2_root(p.d_tw, r * k2 * (2 * M / 64), M);
    } else if (kTw2Reg) {
        const int k2 = l & (S::R1 - 1);
#pragma unroll
        for (int r = 1; r < 8; ++r) tw2[r - 1] = pow2_root(p.d_tw, r * k2 * (2 * M / (S::R1 * 8)), M);
    }

    struct Frame {                                       // where a frame is, per lane group
        const float *pcm, *x;
        float *o;
        uint64_t ostep, start;
        // tag = the frame's unit inside its clip (uniform batches; < 2^30) | have << 30 | real << 31, and the clip.  The flags were two
        // `bool` members: with byte-sized members the tail of the struct is not split into registers -- `Frame nxt = cur` and `cur = nxt`
        // went through scratch memory, a load / s_waitcnt vmcnt(0) / store pair at both ends of every iteration of the frame loop, each
        // wait also draining the loads issued ahead for the next frame (found by tools/hotloop_spills.py, round 5).  As two 32-bit words
        // they cost the 1024- and 2048-point instances (at 256 VGPRs) more than the scratch copy did (+1..3 %; n_fft 256 -5 %); packed:
        // n_fft 128 -3 %, 256 -4 %, Kaldi 32 kHz -2 %, 1024 / 2048 unchanged (same box, profiles/r05_pow2.txt).
        uint32_t tag, clip;
        __device__ __forceinline__ bool have() const { return (tag >> 30) & 1u; }
        __device__ __forceinline__ bool real() const { return (tag >> 31) != 0; }
        __device__ __forceinline__ uint32_t in_clip() const { return tag & 0x3fffffffu; }
    };
    const uint64_t n_units = batch_n_units(p.b);
    const uint64_t stride = (uint64_t)gridDim.x * n_waves * FW;
    const bool uniform = p.b.d_unit_prefix == nullptr;
    const bool walk = uniform && p.b.units_per_clip < (1u << 29);     // Frame::tag holds the unit inside its clip in 30 bits
    auto frame_at = [&](const UnitLoc &loc, bool have) {
        Frame f;
        const uint64_t width = uniform ? p.b.out_width : loc.frames;
        f.o = p.b.mel_major ? loc.out + loc.unit : loc.out + loc.unit * (uint64_t)p.n_mels;
        f.ostep = p.b.mel_major ? width : 1;
        const bool real = have && loc.unit < loc.frames;     // otherwise: a zero column of a padded layout (uniform batches), or nothing
        f.start = loc.unit * (uint64_t)p.hop;
        f.pcm = loc.pcm;
        f.x = loc.pcm + f.start;
        f.tag = ((uint32_t)loc.unit & 0x3fffffffu) | (have ? 1u << 30 : 0u) | (real ? 1u << 31 : 0u);
        f.clip = loc.clip;
        return f;
    };
    auto place = [&](uint64_t base) {
        const uint64_t unit = base + fs;
        const bool have = unit < n_units;
        return frame_at(locate_unit(p.b, have ? unit : base), have);
    };
    // The frame `stride` units further on.  locate_unit divides a 64-bit unit index by the units of a clip -- ~100 VALU instructions per
    // lane, a sixth of this kernel's at n_fft 256 when it was done per frame; a uniform batch is walked instead: the step in whole
    // clips and the rest are the same for every lane and every iteration.
    const uint64_t step_clips = uniform ? stride / p.b.units_per_clip : 0;
    const uint32_t step_rest = uniform ? (uint32_t)(stride - step_clips * p.b.units_per_clip) : 0;
    auto advance = [&](const Frame &f, uint64_t nbase) {
        if (!walk || !f.have()) return place(nbase);
        UnitLoc loc;
        uint32_t u = f.in_clip() + step_rest;             // (< 2 units_per_clip <= 2^32: the host plans uniform batches with 32-bit unit counts per clip)
        uint64_t c = (uint64_t)f.clip + step_clips;
        if (u >= p.b.units_per_clip) { u -= p.b.units_per_clip; ++c; }
        loc.unit = u;
        loc.clip = (uint32_t)c;
        loc.pcm = p.b.pcm + c * p.b.clip_stride;
        loc.out = p.b.out + c * p.b.out_stride;
        loc.frames = p.b.frames_per_clip;
        return frame_at(loc, nbase + fs < n_units);
    };
    // Every load is unconditional (clamped index, the value selected afterwards): a load behind its own branch is a serialised memory
    // round trip, and the first form of this kernel -- one predicate per sample -- spent 80 % of its time in them.
    // part: 2 = every point; 0 / 1 (kHalves): the even / the odd points only (r = 2 r' + part)
    auto fetch = [&](const Frame &f, Pow2Raw<P, FLAVOR> &raw, int part = 2) {
        if (!f.real()) return;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            if (part != 2 && (r & 1) != part) continue;
            const int i0 = 2 * pt(r);
            if (FLAVOR == 2) {                           // sample s of the frame = clip[start + s - pad], zero outside the clip
                const long long s0 = (long long)f.start + i0 - p.pad, hi = p.clip_len - 1;
                const long long c0 = s0 < 0 ? 0 : (s0 > hi ? hi : s0), c1 = s0 + 1 < 0 ? 0 : (s0 + 1 > hi ? hi : s0 + 1);
                raw.pair[r] = f2{f.pcm[c0], f.pcm[c1]};
                raw.before[r] = f.pcm[c0 > 0 ? c0 - 1 : 0];
            } else {
                const int pi = i0 + 1 <= last ? i0 : (last >= 1 ? last - 1 : 0);     // never past the frame's last sample
                raw.pair[r] = load2_unaligned(f.x + pi);
                if (FLAVOR == 1) raw.before[r] = (pi > 0 || f.start > 0) ? f.x[pi - 1] : f.x[0];
            }
        }
    };

    // the jobs of a lane are the same for every frame too: the records of its first kJ stay in registers
    const int n_jobs = p.n_jobs;
    constexpr int kJ = (FLAVOR != 0 && (P == 8 || S::kHalves)) ? MS_POW2_JOBS_F : (LF < 64 ? MS_POW2_JOBS_SMALL : (S::kHalves ? MS_POW2_JOBS_H : MS_POW2_JOBS_BIG));     // rounds of jobs in flight together (the Kaldi / NeMo framings hold more registers: spills)
    int info0[kJ];
#pragma unroll
    for (int t = 0; t < kJ; ++t) info0[t] = l + t * LF < n_jobs ? ljob[l + t * LF] : 0;

    uint64_t base = ((uint64_t)blockIdx.x * n_waves + wave) * FW;
    if (base >= n_units) return;
    Frame cur = place(base);
    constexpr bool kFetchPerHalf = S::kHalves && !kAhead;      // the samples of a half are loaded when the half is framed (registers)
    Pow2Raw<P, FLAVOR> raw;
    if (!kFetchPerHalf) fetch(cur, raw);
    for (;;) {
        const uint64_t nbase = base + stride;
        const bool more = nbase < n_units;               // wave-uniform
        Frame nxt = cur;
        Pow2Raw<P, FLAVOR> nraw;
        if (kAhead && more) {
            nxt = advance(cur, nbase);
            fetch(nxt, nraw);
        }
        if (cur.have() && !cur.real()) {
            for (int m = l; m < p.n_mels; m += LF) cur.o[m * cur.ostep] = 0.0f;
        }
        if (cur.real()) {
            // ---- framing: DC removal / pre-emphasis / window per flavour -> the lane's P complex points z[l + r LF] -----------------
            // point(r): the windowed complex point r of the lane.  FLAVOR 1 needs the frame's mean first.
            double mean = 0.0;
            if (FLAVOR == 1) {                           // src/fbank.rs:164-170
                if (kFetchPerHalf) fetch(cur, raw);
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const int i0 = 2 * pt(r);
                    const double xa = (double)(i0 + 1 <= last ? raw.pair[r].x : raw.pair[r].y);     // i0 == last: the clamped pair holds x[last] second
                    part += i0 <= last ? xa : 0.0;
                    part += i0 + 1 <= last ? (double)raw.pair[r].y : 0.0;
                }
                // (the frame's lanes are all inside this branch or all outside it: a frame owns a whole lane group)
#pragma unroll
                for (int d = 1; d < LF; d <<= 1) part += __shfl_xor(part, d, 64);
                mean = part / (double)p.frame_len;
            }
            auto point = [&](int r, d2 w) {
                double va, vb;
                if (FLAVOR == 0) {                       // frame_windows: x[start + i] as f64 (src/stft.rs:160-165)
                    va = (double)raw.pair[r].x; vb = (double)raw.pair[r].y;
                } else if (FLAVOR == 2) {                // src/mel.rs:696-706: whole-clip pre-emphasis in f32, two roundings; zero centre padding
                    const float coeff = (float)p.preemph;
                    const long long s0 = (long long)cur.start + 2 * pt(r) - p.pad;
                    const float a = raw.pair[r].x, b0 = raw.pair[r].y;
                    const float pa = a - f32_mul_rn(coeff, raw.before[r]), pb = b0 - f32_mul_rn(coeff, a);
                    const float fa = (coeff != 0.0f && s0 > 0) ? pa : a, fb = (coeff != 0.0f && s0 + 1 > 0) ? pb : b0;
                    va = (s0 >= 0 && s0 < p.clip_len) ? (double)fa : 0.0;
                    vb = (s0 + 1 >= 0 && s0 + 1 < p.clip_len) ? (double)fb : 0.0;
                } else {                                 // src/fbank.rs:171-190: DC removal, pre-emphasis
                    const int i0 = 2 * pt(r);
                    const double xa = (double)(i0 + 1 <= last ? raw.pair[r].x : raw.pair[r].y), xb = (double)raw.pair[r].y;
                    va = xa - mean; vb = xb - mean;
                    if (p.preemph > 0.0) {
                        vb -= p.preemph * (xa - mean);
                        // the sample in front of xa: the clamped pair of an odd frame's last sample (i0 == last) holds it first (ADVICE r04:
                        // raw.before is then x[last - 2]; the Povey window's last tap is 0, so no test could see it)
                        const double xp = (double)(i0 + 1 <= last ? raw.before[r] : raw.pair[r].x);
                        if (i0 > 0 || cur.start > 0) va -= p.preemph * (xp - mean);
                    }
                }
                return cpx<double>{va * w.x, vb * w.y};
            };
            auto window_of = [&](int r) { return *reinterpret_cast<const d2 *>((kWinLds ? lwin : p.d_win) + 2 * pt(r)); };
            // ---- the complex M-point transform: Stockham passes, in place in the frame's LDS region --------------------------------
            if (kHalves) {
                // the even and the odd points as two 512-point transforms, E at z, O behind it
                // (eight points at a time, window values and all: sixteen at once are the registers of the one-transform form)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    d2 wv[8];
                    cpx<double> half[8];
                    if (kFetchPerHalf) fetch(cur, raw, h);      // (Kaldi: again, after the pass for the mean -- holding all sixteen pairs spills more than the reload costs)
#pragma unroll
                    for (int r = 0; r < 8; ++r) wv[r] = window_of(2 * r + h);
#pragma unroll
                    for (int r = 0; r < 8; ++r) half[r] = point(2 * r + h, wv[r]);
                    double *zh = z + h * M;
                    pow2_pass<9, 8, true>(l, 1, nullptr, zh, half, nullptr);
                    pow2_pass<9, 8, false>(l, 8, t3, zh, nullptr, tw2);          // (t3: any table; the twiddles are tw2)
                    pow2_pass<9, 8, false>(l, 64, t3, zh, nullptr, nullptr);
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            } else {
                cpx<double> reg[P];
                d2 wv[P];
#pragma unroll
                for (int r = 0; r < P; ++r) wv[r] = window_of(r);
#pragma unroll
                for (int r = 0; r < P; ++r) reg[r] = point(r, wv[r]);
                pow2_pass<LOGM, S::R1, true>(l, 1, nullptr, z, reg, nullptr);
                pow2_pass<LOGM, 8, false>(l, S::R1, t2, z, nullptr, kTw2Reg ? tw2 : nullptr);
                if (S::R3 > 1) pow2_pass<LOGM, (S::R3 > 1 ? S::R3 : 2), false>(l, S::R1 * 8, t3, z, nullptr, nullptr);
            }
            // ---- the real-FFT split X[k] = E[k] + W_N^k O[k], E = (Z[k] + conj Z[M-k]) / 2, O = -i (Z[k] - conj Z[M-k]) / 2, and the
            // power row.  X[M-k] comes from the same two points (E -> conj E, O -> conj O, W_N^{M-k} = -conj W_N^k):
            //   X[k] = (er + t1) + i (ei + t2),   X[M-k] = (er - t1) - i (ei - t2),   t1 = di c + dr s,  t2 = di s - dr c
            // so a lane takes the pairs k = l + r LF < M / 2 (k = 0 gives bins 0 and M); bin M / 2 is its own partner.
            auto power2 = [&](int k, double &lo, double &hi) {
                const cpx<double> a = ldc(z + 2 * pow2_slot<LOGM>(k)), b0 = ldc(z + 2 * pow2_slot<LOGM>((M - k) & (M - 1)));
                const cpx<double> w = ldc(tw + 2 * k);
                const double er = 0.5 * (a.re + b0.re), ei = 0.5 * (a.im - b0.im);
                const double dr = 0.5 * (a.re - b0.re), di = 0.5 * (a.im + b0.im);
                const double t1 = di * w.re + dr * w.im, t2v = di * w.im - dr * w.re;
                const double ar = er + t1, ai = ei + t2v, br = er - t1, bi = ei - t2v;
                lo = ar * ar + ai * ai;
                hi = br * br + bi * bi;
            };
            // kHalves: Z[k] = E[k] + W_M^k O[k] and Z[M - k] = Z[(M/2 - k) + M/2] = E[M/2 - k] + conj(W_M^k) O[M/2 - k] (W_M^{M/2 - k} =
            // -conj W_M^k) are formed on the way in: the radix-2 step costs no pass of its own.  Bin M / 2: Z[M/2] = E[0] - O[0].
            auto pair_h = [&](cpx<double> a, cpx<double> b0, int k, double &lo, double &hi) {
                const cpx<double> w = ldc(tw + 2 * k);
                const double er = 0.5 * (a.re + b0.re), ei = 0.5 * (a.im - b0.im);
                const double dr = 0.5 * (a.re - b0.re), di = 0.5 * (a.im + b0.im);
                const double t1 = di * w.re + dr * w.im, t2v = di * w.im - dr * w.re;
                const double ar = er + t1, ai = ei + t2v, br = er - t1, bi = ei - t2v;
                lo = ar * ar + ai * ai;
                hi = br * br + bi * bi;
            };
            auto power2h = [&](int k, double &lo, double &hi) {
                const int km = (M / 2 - k) & (M / 2 - 1);
                const cpx<double> ek = ldc(z + 2 * pow2_slot<9>(k)), ok = ldc(z + M + 2 * pow2_slot<9>(k));
                const cpx<double> em = ldc(z + 2 * pow2_slot<9>(km)), om = ldc(z + M + 2 * pow2_slot<9>(km));
                const cpx<double> wc = ldc(tc + 2 * k);
                const cpx<double> wo = cmul(wc, ok), wm = cmul(cpx<double>{wc.re, -wc.im}, om);
                pair_h(cpx<double>{ek.re + wo.re, ek.im + wo.im}, cpx<double>{em.re + wm.re, em.im + wm.im}, k, lo, hi);
            };
            double plo[P / 2], phi[P / 2];
            double pmid, pmid2;
            __builtin_amdgcn_wave_barrier();                // the split's reads stay behind the last pass's writes ...
            if (kHalves) {
#pragma unroll
                for (int r = 0; r < P / 2; ++r) {
                    power2h(l + r * LF, plo[r], phi[r]);
#if defined(__HIP_DEVICE_COMPILE__)
                    if (r % MS_POW2_HSPLIT == MS_POW2_HSPLIT - 1) __builtin_amdgcn_sched_barrier(0);      // (a pair is five 16-byte loads: all eight at once are 160 registers)
#endif
                }
                const cpx<double> e0 = ldc(z), o0 = ldc(z + M);              // slot(0) = 0
                pair_h(cpx<double>{e0.re - o0.re, e0.im - o0.im}, cpx<double>{e0.re - o0.re, e0.im - o0.im}, M / 2, pmid, pmid2);
            } else {
#pragma unroll
                for (int r = 0; r < P / 2; ++r) power2(l + r * LF, plo[r], phi[r]);
                power2(M / 2, pmid, pmid2);              // every lane, one address: a broadcast
            }
            // magnitudes instead of powers (FbankConfig::use_power off): ONE wave-uniform branch around all the square roots -- as a select
            // inside the pair the compiler evaluated the 2 (P / 2 + 1) IEEE f64 roots of every frame unconditionally (~300 instructions, a
            // third of the Kaldi flavour's arithmetic; the same trap as in fb_phase2_split)
            if (FLAVOR == 1 && !p.use_power) {
#pragma unroll
                for (int r = 0; r < P / 2; ++r) { plo[r] = sqrt(plo[r]); phi[r] = sqrt(phi[r]); }
                pmid = sqrt(pmid);
            }
            __builtin_amdgcn_wave_barrier();                // ... and in front of the power row's writes, which alias the points at M >= 1024
#pragma unroll
            for (int r = 0; r < P / 2; ++r) {
                pw[l + r * LF] = plo[r];
                pw[M - (l + r * LF)] = phi[r];
            }
            if (l == 0) pw[M / 2] = pmid;
            if (kPwAlias && l < 8) pw[M + 1 + l] = 0.0;      // (the row is where the points were)
        }
        // ---- banded mel sums, log, per-flavour epilogue ---------------------------------------------------------------------------
        // The bank as JOBS of eight consecutive weights of one mel (the last job of a band padded): every lane takes a job per round,
        // folds its up-to-eight products left to right and adds the partial sum to the mel's word in LDS (ds_add_f64; the LDS executes
        // a wave's operations in program order and an instruction's lanes in lane order, so the result is the same on every run).  No
        // trip count depends on a band's width, every load is unconditional, all lanes are busy: a mel per lane and step (the form
        // before) left most lanes idle while the lanes of the wide high bands walked 25 bins, with one LDS round trip per tail bin --
        // it was 37-43 % of the kernel.  A band's energy is the reference's left fold (src/mel.rs:155-163) cut into <= 4 pieces.
        // (A first balanced form -- the same number of consecutive ENTRIES per lane, an atomic at every mel boundary inside a lane's
        // range -- had 20 divergent branch sites per frame and was slower than the mel-per-lane form: 3.5 against 2.05 ms.)
        constexpr int kMaxPer = Pow2Shape<LOGM>::kMelsPerLane;            // a lane reads out the mels m = l + LF i (the host checks n_mels <= kMelsPerLane * LF)
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) if (l + LF * i < p.n_mels) acc[l + LF * i] = 0.0;
        // kJ = three rounds at a time: their 36 loads are in flight together (the transform's registers are free here), one LDS round trip
        // instead of three -- at two waves per SIMD the kernel is a chain of such round trips, not of arithmetic
        auto job_triple = [&](const int (&info)[kJ], int jb0) {
            d2 w[kJ][4];
            double pv[kJ][8];
            const int jstep = 2 * n_jobs;
#pragma unroll
            for (int t = 0; t < kJ; ++t) {
                const int jb = jb0 + t * LF;
                const double *wp = ljw + 2 * (jb < n_jobs ? jb : 0), *pp = pw + (info[t] & 0xfff);      // weights 2 q, 2 q + 1 of job j at [q][j]: consecutive lanes, consecutive slots
#pragma unroll
                for (int q = 0; q < 4; ++q) w[t][q] = *reinterpret_cast<const d2 *>(wp + q * jstep);
#pragma unroll
                for (int q = 0; q < 4; ++q) {                      // a job starts at an even bin: four aligned ds_read_b128
                    const d2 two = *reinterpret_cast<const d2 *>(pp + 2 * q);
                    pv[t][2 * q] = two.x; pv[t][2 * q + 1] = two.y;
                }
            }
#pragma unroll
            for (int t = 0; t < kJ; ++t) {
                // (a job's weights past its count are +0 and what it reads past the row is +0: e + 0 * p = e, no selects)
                double e = w[t][0].x * pv[t][0];
                e += w[t][0].y * pv[t][1]; e += w[t][1].x * pv[t][2]; e += w[t][1].y * pv[t][3];
                e += w[t][2].x * pv[t][4]; e += w[t][2].y * pv[t][5]; e += w[t][3].x * pv[t][6]; e += w[t][3].y * pv[t][7];
                if (cur.real() && (info[t] >> 20) > 0) unsafeAtomicAdd(acc + ((info[t] >> 12) & 0xff), e);       // (count 0: the host's padding, or past the last job)
            }
        };
        job_triple(info0, l);
        for (int jb0 = l + kJ * LF; jb0 < n_jobs + l; jb0 += kJ * LF) {            // wave-uniform trip count
            int info[kJ];
#pragma unroll
            for (int t = 0; t < kJ; ++t) info[t] = jb0 + t * LF < n_jobs ? ljob[jb0 + t * LF] : 0;
            job_triple(info, jb0);
        }
        // log2 through v_log_f32 (1 ulp: <= 1.2e-6 of a log10 / ln value, 3e-7 after Whisper's / 4), like the fused kernels
        float mv[kMaxPer];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < kMaxPer; ++i) {
            const int m = l + LF * i;
            mv[i] = 0.0f;
            if (cur.real() && m < p.n_mels) {
                const double e = acc[m];
                float vv;
                if (FLAVOR == 0) {
                    vv = fast_log2((float)(e > 1e-10 ? e : 1e-10)) * 0.30102999566398120f;          // src/mel.rs:166
                } else if (FLAVOR == 2) {
                    vv = fast_log2((float)(e + p.floor_v)) * 0.69314718055994531f;                  // src/mel.rs:365-368
                } else {
                    const float t = (float)(e > p.floor_v ? e : p.floor_v);                           // src/fbank.rs:210-218
                    vv = p.use_log ? fast_log2(t) * 0.69314718055994531f : t;
                }
                mv[i] = vv;
                mx = mx > vv ? mx : vv;
            }
        }
        if (FLAVOR == 0) {                                     // src/mel.rs:645-654: clamp at the frame's maximum - 8, (x + 4) / 4
#pragma unroll
            for (int d = 1; d < LF; d <<= 1) { const float t = __shfl_xor(mx, d, 64); mx = mx > t ? mx : t; }
            const float lo = mx - 8.0f;
#pragma unroll
            for (int i = 0; i < kMaxPer; ++i) {
                const int m = l + LF * i;
                if (cur.real() && m < p.n_mels) cur.o[m * cur.ostep] = ((mv[i] > lo ? mv[i] : lo) + 4.0f) * 0.25f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kMaxPer; ++i) {
                const int m = l + LF * i;
                if (cur.real() && m < p.n_mels) cur.o[m * cur.ostep] = mv[i];
            }
        }
        if (!more) break;
        base = nbase;
        if (kAhead) {
            cur = nxt;
            raw = nraw;
        } else {
            cur = advance(cur, base);
            if (!kFetchPerHalf) fetch(cur, raw);
        }
    }
}


}  // namespace melspec
