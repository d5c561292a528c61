// mel_bank.hpp -- the stand-alone mel helpers of src/mel.rs on the device, for callers that keep the reference's split API
// (examples/vad_ten_eval/src/main.rs:232-256 combines compute_all_cpu with its own filterbank and normalisation):
//
//   SparseMelFilterbank::{from_dense, from_mel}            src/mel.rs:48-87     melspec_bank_from_dense / _from_mel
//   SparseMelFilterbank::project_power_f64 / _f32          src/mel.rs:106-146   bank_project_power_kernel<double / float>
//   log_mel_spectrogram(stft, mel_filters)                 src/mel.rs:436-441   bank_log_mel_kernel (project_stft_log10, :148-168)
//   norm_mel / norm_mel_vec                                src/mel.rs:448-469   norm_max_kernel + norm_map_kernel
//
// Rows are the reference's sparse rows (the non-zero entries of the dense matrix, ascending bin) and every sum is the reference's
// left fold with a separate multiply and add (Rust does not contract), so project_power is BIT-EXACT against the oracle in f32 and
// in f64.  One thread per (frame, mel); a frame's power row is read by the threads of its mels, which sit next to each other.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace melspec {

struct BankDesc {
    const int *row_ptr;      // [n_mels + 1]
    const int *bin;          // [nnz]
    const double *w;         // [nnz]
    const float *wf;         // [nnz]: `weight as f32`
    int n_mels, fft_bins;
};

template <class T>
__global__ __launch_bounds__(256) void bank_project_power_kernel(const BankDesc b, const T *power, T *out, uint64_t n_frames) {
#pragma clang fp contract(off)
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t frame = idx / (uint64_t)b.n_mels;
    if (frame >= n_frames) return;
    const int m = (int)(idx - frame * (uint64_t)b.n_mels);
    const T *p = power + frame * (uint64_t)b.fft_bins;
    T e = T(0);
    for (int j = b.row_ptr[m]; j < b.row_ptr[m + 1]; ++j) {
        const T w = sizeof(T) == 8 ? (T)b.w[j] : (T)b.wf[j];
        const T prod = w * p[b.bin[j]];
        e = e + prod;
    }
    out[idx] = e;
}

// stft: [n_frames][n_fft] complex (interleaved re, im) of T; out: [n_frames][n_mels] f64 = log10(max(E, 1e-10)), E in f64
template <class T>
__global__ __launch_bounds__(256) void bank_log_mel_kernel(const BankDesc b, const T *stft, int n_fft, double *out, uint64_t n_frames) {
#pragma clang fp contract(off)
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t frame = idx / (uint64_t)b.n_mels;
    if (frame >= n_frames) return;
    const int m = (int)(idx - frame * (uint64_t)b.n_mels);
    const T *z = stft + frame * (uint64_t)n_fft * 2;
    const int half = n_fft / 2;
    double e = 0.0;
    for (int j = b.row_ptr[m]; j < b.row_ptr[m + 1]; ++j) {
        const int k = b.bin[j];
        double pw = 0.0;
        if (k < half) {                                   // src/mel.rs:155-163
            const double re = (double)z[2 * k], im = (double)z[2 * k + 1];
            const double a = re * re, c = im * im;
            pw = a + c;                                   // Complex::norm_sqr
        }
        const double prod = b.w[j] * pw;
        e = e + prod;
    }
    out[idx] = log10(e > 1e-10 ? e : 1e-10);
}

// order-preserving keys of f32 / f64 bit patterns (non-NaN): a < b <=> key(a) < key(b)
__host__ __device__ inline unsigned long long norm_key(double v) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double norm_unkey(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    __builtin_memcpy(&v, &u, 8);
    return v;
}

// fold(NEG_INFINITY, |acc, x| acc.max(x)) (src/mel.rs:449,460): NaNs are skipped like f64::max / f32::max skip them.  f32 values are
// widened (exact) so that one 64-bit key serves both.  *key starts at norm_key(-inf).
template <class T>
__global__ __launch_bounds__(256) void norm_max_kernel(const T *in, uint64_t n, unsigned long long *key) {
    __shared__ double red[4];
    double mx = -__builtin_inf();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double v = (double)in[i];
        if (v == v && v > mx) mx = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) mx = red[w] > mx ? red[w] : mx;
        atomicMax(key, norm_key(mx));
    }
}

// (x.max(mmax - 8) + 4) / 4 in T (src/mel.rs:450-452, 463-467)
template <class T>
__global__ __launch_bounds__(256) void norm_map_kernel(const T *in, uint64_t n, const unsigned long long *key, T *out) {
#pragma clang fp contract(off)
    const T mmax = (T)norm_unkey(*key) - T(8);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const T x = in[i];
        const T c = (x == x && x > mmax) ? x : mmax;          // x.max(mmax): a NaN x yields mmax
        out[i] = (c + T(4)) / T(4);
    }
}

__global__ void norm_init_kernel(unsigned long long *key) { *key = norm_key(-__builtin_inf()); }

}  // namespace melspec
