// xcc_map.hip -- which XCD does workgroup id w land on?  (s_getreg_b32 HW_REG_XCC_ID, gfx942/gfx950: id 20, bits 3:0)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    const int n = 64;
    unsigned *d, h[n];
    hipMalloc(&d, n * 4);
    for (int threads : {64, 512}) {
        k<<<n, threads>>>(d);
        hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        printf("threads %d:", threads);
        for (int i = 0; i < n; ++i) printf(" %u", h[i] & 0xf);
        printf("\n");
    }
    return 0;
}
