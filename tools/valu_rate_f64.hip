// valu_rate_f64.hip -- issue rate (cycles per wave64 instruction per SIMD) of the f64 VALU ops and the
// f32<->f64 conversions the precise kernel uses, on gfx950.  Same harness as valu_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b0 = 1.0001, b1 = 0.9999;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
            } else if (OP == 1) {
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
            } else if (OP == 2) {
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
            } else if (OP == 3) {   // v_cvt_f64_f32
                asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                             "v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %10\n v_cvt_f64_f32 %7, %11\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
            } else if (OP == 4) {   // v_cvt_f32_f64
                asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                             "v_cvt_f32_f64 %0, %5\n v_cvt_f32_f64 %1, %6\n v_cvt_f32_f64 %2, %7\n v_cvt_f32_f64 %3, %4\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            } else if (OP == 5) {   // 64-bit move pair (v_mov_b64)
                asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n"
                             "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
            }
        }
    }
    double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3;
    if (r == 12345.678) out[0] = r;
}

template <int OP>
int run(const char* name, double* d, int blocks_per_cu, double clk_ghz) {
    const int iters = 10000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<256 * blocks_per_cu, 256>>>(d, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<OP><<<256 * blocks_per_cu, 256>>>(d, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)blocks_per_cu * iters * 64.0;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f cycles/instr/SIMD (at %.2f GHz)\n", name, blocks_per_cu, ms,
           ms * 1e-3 * clk_ghz * 1e9 / instr_per_simd, clk_ghz);
    return 0;
}

int main() {
    double* d; CHECK(hipMalloc(&d, 1024));
    const double clk = 2.4;
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", d, w, clk); run<1>("v_add_f64", d, w, clk); run<2>("v_mul_f64", d, w, clk);
        run<3>("v_cvt_f64_f32", d, w, clk); run<4>("v_cvt_f32_f64", d, w, clk); run<5>("v_mov_b64", d, w, clk);
    }
    return 0;
}
