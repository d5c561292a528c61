// Latency of a chain of dependent v_add_f32 in one lone wavefront (and of 2 / 4 interleaved chains): what bounds the
// left-fold sums of blm_normalize_kernel / cmn_kernel.  hipcc --offload-arch=gfx950 -O3 tools/dep_add.hip -o /tmp/dep_add
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(const float *in, float *out, long long *cycles, int n) {
    float s[CHAINS];
    for (int c = 0; c < CHAINS; ++c) s[c] = in[threadIdx.x + c];
    const float x = in[64 + threadIdx.x];
    const long long t0 = clock64();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[c]) : "v"(x));
    }
    const long long t1 = clock64();
    float r = 0;
    for (int c = 0; c < CHAINS; ++c) r += s[c];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    float *in, *out; long long *cyc, h;
    hipMalloc(&in, 1024); hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
    hipMemset(in, 0, 1024);
    const int n = 1 << 16;
    for (int lanes : {64, 18}) {
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(lanes), 0, 0, in, out, cyc, n); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(lanes), 0, 0, in, out, cyc, n); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        std::printf("lanes %2d  1 chain : %.2f clock64 ticks per dependent add\n", lanes, (double)h / n);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(lanes), 0, 0, in, out, cyc, n); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        std::printf("lanes %2d  2 chains: %.2f ticks per add (per chain element %.2f)\n", lanes, (double)h / (2.0 * n), (double)h / n);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(lanes), 0, 0, in, out, cyc, n); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        std::printf("lanes %2d  4 chains: %.2f ticks per add (per chain element %.2f)\n", lanes, (double)h / (4.0 * n), (double)h / n);
    }
    return 0;
}
