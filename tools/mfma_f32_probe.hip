// mfma_f32_probe.hip -- does the f32 (and bf16) matrix pipe of gfx950 issue BESIDE the f32 VALU of another wave on the same SIMD?
// ("next" 1(b).)  The f64 twin of this probe (tools/mfma_f64_probe.hip, profiles/r04_mfma_f64_probe.txt) found "sum, not
// maximum" for v_mfma_f64_16x16x4_f64 beside v_fma_f64; the headline kernel is f32 (VALU ~58 % busy, LDS ~60 %), so what matters for it is
// whether v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 (exact f32 products, the only matrix forms that keep the 1e-4 contract without a
// hi/lo split) are issue capacity the SIMD does not have today.
//
// Per SIMD and outer iteration: MFMA alone, v_fma_f32 alone, both interleaved in one wave, both from DIFFERENT waves of one SIMD (the
// low half of the workgroup's waves issues MFMAs, the high half v_fma_f32; wave w sits on SIMD w % 4), at two and at four waves per SIMD
// (the headline kernel's occupancy).  "max" = separate pipes, "sum" = one datapath.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_probe.hip -o /tmp/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

#define FMA8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"

// KIND 0: v_mfma_f32_16x16x4_f32 (1024 lane-FMAs); 1: v_mfma_f32_32x32x2_f32 (2048); 2: v_mfma_f32_16x16x32_bf16 (8192 MACs)
// MODE 0: MFMA only; 1: FMA only; 2: both in every wave (one MFMA, then `per` FMAs, repeated); 3: low half of the waves MFMA, high half FMA
template <int KIND, int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, int n_mfma, int n_fma, int per) {
    const int wave = threadIdx.x >> 6, half = blockDim.x >> 7;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f16v w0 = {0}, w1 = w0;
    bf8v ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b + i); }
    float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < half);
    const bool do_f = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= half);
    auto mfma4 = [&]() {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        } else if (KIND == 1) {
            w0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, w0, 0, 0, 0);
            w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, w1, 0, 0, 0);
            w0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, w0, 0, 0, 0);
            w1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, w1, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c3, 0, 0, 0);
        }
    };
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2) {
            for (int u = 0; u < n_mfma; u += 4) {
                mfma4();
                for (int v = 0; v < per; v += 8)
                    asm volatile(FMA8 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));
            }
        } else {
            if (do_m)
                for (int u = 0; u < n_mfma; u += 4) mfma4();
            if (do_f)
                for (int u = 0; u < n_fma; u += 8)
                    asm volatile(FMA8 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));
        }
    }
    float r = c0.x + c1.y + c2.z + c3.w + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    for (int i = 0; i < 16; ++i) r += w0[i] + w1[i];
    if (r == 12345.678f) out[0] = r;
}

static double g_mhz = 2400.0;

template <int KIND, int MODE>
int run(const char *name, float *d, int threads, int n_mfma, int n_fma, int per = 0) {
    const int iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<KIND, MODE><<<256, threads>>>(d, 10, n_mfma, n_fma, per);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        k<KIND, MODE><<<256, threads>>>(d, iters, n_mfma, n_fma, per);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double cyc = best * 1e-3 * g_mhz * 1e6 / iters;      // SIMD cycles per outer iteration
    printf("%-86s %4d thr  %.3f ms  %6.0f cycles/iter/SIMD\n", name, threads, best, cyc);
    return 0;
}

int main() {
    float *d; CHECK(hipMalloc(&d, 1024));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    g_mhz = p.clockRate / 1000.0;
    printf("# %s, %d CUs, clockRate %.0f MHz (cycles are at that clock)\n", p.name, p.multiProcessorCount, g_mhz);
    printf("# -- v_mfma_f32_16x16x4_f32: 16 MFMAs = 256 v_fma_f32 = 16384 lane-FMAs per wave\n");
    run<0, 0>("A  MFMA 16x16x4 f32 x16, 1 wave/SIMD", d, 256, 16, 0);
    run<0, 1>("B  v_fma_f32 x256, 1 wave/SIMD", d, 256, 0, 256);
    run<0, 0>("A2 MFMA x16 per wave, 2 waves/SIMD", d, 512, 16, 0);
    run<0, 1>("B2 v_fma_f32 x256 per wave, 2 waves/SIMD", d, 512, 0, 256);
    run<0, 3>("C  wave A: MFMA x16, wave B of the same SIMD: v_fma_f32 x256   (A + B = sum, max(A, B) = two pipes)", d, 512, 16, 256);
    run<0, 3>("C' wave A: MFMA x16, wave B: v_fma_f32 x64", d, 512, 16, 64);
    run<0, 3>("C\" wave A: MFMA x4,  wave B: v_fma_f32 x256", d, 512, 4, 256);
    run<0, 0>("A4 MFMA x16 per wave, 4 waves/SIMD", d, 1024, 16, 0);
    run<0, 1>("B4 v_fma_f32 x256 per wave, 4 waves/SIMD", d, 1024, 0, 256);
    run<0, 3>("D  4 waves/SIMD: two issue MFMA x16, two issue v_fma_f32 x256", d, 1024, 16, 256);
    run<0, 2>("E  same wave: (4 MFMA, 64 FMA) x4, 1 wave/SIMD", d, 256, 16, 0, 64);
    run<0, 2>("E4 same wave: (4 MFMA, 64 FMA) x4, 4 waves/SIMD", d, 1024, 16, 0, 64);
    run<0, 2>("F4 same wave: (4 MFMA, 16 FMA) x4, 4 waves/SIMD   (the headline kernel's shape if phase 3 were tiles)", d, 1024, 16, 0, 16);
    printf("# -- v_mfma_f32_32x32x2_f32: 8 MFMAs = 16384 lane-FMAs per wave\n");
    run<1, 0>("A  MFMA 32x32x2 f32 x8, 1 wave/SIMD", d, 256, 8, 0);
    run<1, 0>("A2 MFMA 32x32x2 f32 x8, 2 waves/SIMD", d, 512, 8, 0);
    run<1, 3>("C  wave A: MFMA 32x32x2 x8, wave B: v_fma_f32 x256", d, 512, 8, 256);
    run<1, 3>("D  4 waves/SIMD: two issue MFMA 32x32x2 x8, two issue v_fma_f32 x256", d, 1024, 8, 256);
    printf("# -- v_mfma_f32_16x16x32_bf16 (8192 MACs each): is ANY matrix instruction a second pipe?\n");
    run<2, 0>("A  MFMA 16x16x32 bf16 x16, 1 wave/SIMD", d, 256, 16, 0);
    run<2, 0>("A2 MFMA 16x16x32 bf16 x16, 2 waves/SIMD", d, 512, 16, 0);
    run<2, 3>("C  wave A: MFMA bf16 x16, wave B: v_fma_f32 x256", d, 512, 16, 256);
    run<2, 3>("C' wave A: MFMA bf16 x64, wave B: v_fma_f32 x256", d, 512, 64, 256);
    run<2, 3>("D  4 waves/SIMD: two issue MFMA bf16 x64, two issue v_fma_f32 x256", d, 1024, 64, 256);
    return 0;
}
