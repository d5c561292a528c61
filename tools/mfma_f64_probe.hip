// mfma_f64_probe.hip -- does the f64 matrix pipe of gfx950 buy the 512-point kernels anything?  ("next" 1(a))
//
// Measures, per SIMD, the cost of v_mfma_f64_16x16x4_f64 (1024 f64 FMAs per wave-instruction) alone, of v_fma_f64 alone, of the two
// interleaved inside one wave, and of the two issued by DIFFERENT waves of one SIMD (waves 0-3 of an 8-wave workgroup issue MFMAs,
// waves 4-7 issue VALU FMAs: two waves per SIMD, the occupancy of the f64 kernels).  If the matrix pipe ran beside the VALU the
// mixed runs would take max(a, b), not a + b.  The phase-2 DFT-16 of the 512-point kernels as a dense product is
// 16 x 16 complex = 4 real 16x16x16 products = 16 MFMAs per frame against ~170 VALU f64 instructions per 4-frame unit for the
// FFT form, so the matrix form only pays if an MFMA costs (much) less than 16 VALU instructions AND co-issues.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only; 1: FMA only; 2: both interleaved in every wave; 3: waves 0..3 MFMA, waves 4..7 FMA (8-wave workgroups)
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int iters, int n_mfma, int n_fma) {
    const int wave = threadIdx.x >> 6;
    double a = threadIdx.x * 1e-3, b = 1.0001;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_f = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2) {
            // four independent MFMAs with eight VALU FMAs between each pair
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));
            }
        } else {
            if (do_m) {
                for (int u = 0; u < n_mfma; u += 4) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
                }
            }
            if (do_f) {
                for (int u = 0; u < n_fma; u += 8)
                    asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                 "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));
            }
        }
    }
    const double r = c0.x + c1.y + c2.z + c3.w + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (r == 12345.678) out[0] = r;
}

template <int MODE>
int run(const char *name, double *d, int threads, int n_mfma, int n_fma) {
    const int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<MODE><<<256, threads>>>(d, 10, n_mfma, n_fma);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        k<MODE><<<256, threads>>>(d, iters, n_mfma, n_fma);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double cyc = best * 1e-3 * 2.4e9 / iters;      // SIMD cycles per outer iteration at 2.4 GHz
    printf("%-58s %d thr  %.3f ms  %.0f cycles per iteration per SIMD\n", name, threads, best, cyc);
    return 0;
}

int main() {
    double *d; CHECK(hipMalloc(&d, 1024));
    // one wave per SIMD (256 threads), 16 MFMAs or 256 FMAs per iteration = the same 16384 lane-FMAs
    run<0>("MFMA 16x16x4 f64 x16, 1 wave/SIMD", d, 256, 16, 0);
    run<1>("v_fma_f64 x256, 1 wave/SIMD", d, 256, 0, 256);
    run<0>("MFMA x16, 2 waves/SIMD (x32 per SIMD)", d, 512, 16, 0);
    run<1>("v_fma_f64 x256, 2 waves/SIMD (x512 per SIMD)", d, 512, 0, 256);
    run<2>("same wave: 8 MFMA interleaved with 64 FMA, 1 wave/SIMD", d, 256, 0, 0);
    run<2>("same wave: 8 MFMA interleaved with 64 FMA, 2 waves/SIMD", d, 512, 0, 0);
    run<3>("wave A: MFMA x16, wave B of the same SIMD: v_fma_f64 x256", d, 512, 16, 256);
    run<3>("wave A: MFMA x16, wave B: v_fma_f64 x64", d, 512, 16, 64);
    run<3>("wave A: MFMA x4,  wave B: v_fma_f64 x256", d, 512, 4, 256);
    return 0;
}
