// calib.hip -- known-byte-count kernels to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950
// for the access widths the mel kernels use (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads).
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib.hip -o tools/calib ; run under rocprofv3 --pmc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read_b32(const float* __restrict__ in, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read_b128(const float4* __restrict__ in, float* out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i]; acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void write_b32(float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)i;
}
__global__ void write_b128(float4* out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1, 2, 3, (float)i);
}
int main() {
    const size_t n = (size_t)1 << 30;   // 1 Gi floats = 4 GiB, far beyond the 256 MiB Infinity Cache
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, 1024);
    hipMemset(a, 0, n * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto f) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s %8.3f ms  %8.1f GB/s  (bytes=%zu)\n", name, ms, n * 4 / ms / 1e6, n * 4);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("read_b32", [&] { read_b32<<<4096, 256>>>(a, b, n); });
        run("read_b128", [&] { read_b128<<<4096, 256>>>((const float4*)a, b, n / 4); });
        run("write_b32", [&] { write_b32<<<4096, 256>>>(a, n); });
        run("write_b128", [&] { write_b128<<<4096, 256>>>((float4*)a, n / 4); });
    }
    hipDeviceSynchronize();
    return 0;
}
