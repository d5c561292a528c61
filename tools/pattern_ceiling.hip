// pattern_ceiling.hip -- what the memory system gives the ACCESS PATTERN of the headline kernel when no arithmetic stands behind it:
// 256 persistent workgroups of 16 waves, a contiguous run of six-frame units per wave, lane (frame fl, column t) loading the 20
// sample pairs frame + 2t + 20 n1 (8 bytes each; the kernel's 21st-25th loads are halo re-reads that hit L1) and storing nine mel
// values out[(f0 + fl) * 80 + j + 10 i] -- config 2's 1024 x 10 s, 655 MB in + 327 MB out per launch.  Variants: the loads only, the
// stores only, both; and a plain copy (16-byte loads and stores, 2 : 1 read : write) for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pattern_ceiling.hip -o tools/pattern_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kFrames = 998, kClips = 1024, kClipLen = 160000, kHop = 160, kMels = 80;
constexpr int kUnitsPerClip = (kFrames + 5) / 6;

// the same unit walk with the unit's bytes moved as whole 16-byte pieces: the span of its six frames (1360 floats, of which the next
// unit re-reads 400) read once by consecutive lanes, its 480 output floats written by consecutive lanes
template <bool WIDE_LOADS, bool WIDE_STORES>
__global__ __launch_bounds__(1024, 1) void pattern_wide(const float *__restrict__ pcm, float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fl = lane / 10, t = lane - 10 * fl;
    const bool in = lane < 60;
    const uint64_t n_units = (uint64_t)kClips * kUnitsPerClip, waves = (uint64_t)gridDim.x * 16, w = (uint64_t)blockIdx.x * 16 + wave;
    const uint64_t lo = n_units * w / waves, hi = n_units * (w + 1) / waves;
    for (uint64_t u = lo; u < hi; ++u) {
        const uint64_t clip = u / kUnitsPerClip, unit = u - clip * kUnitsPerClip;
        const int f0 = (int)unit * 6, f = f0 + fl;
        const bool act = in && f < kFrames;
        float acc = 0.0f;
        if (WIDE_LOADS) {
            const float4 *s = reinterpret_cast<const float4 *>(pcm + clip * kClipLen + (uint64_t)f0 * kHop);
            const int n4 = (f0 + 6 <= kFrames ? 1360 : (kFrames - f0 - 1) * kHop + 400) / 4;     // new bytes of this unit: 960 floats; span 1360
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lane + 64 * k < 240 && lane + 64 * k < n4 ? s[lane + 64 * k] : make_float4(0, 0, 0, 0);   // 240 x 16 B = the 960 new floats
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        } else if (act) {
            const float *s = pcm + clip * kClipLen + (uint64_t)f * kHop + 2 * t;
            float2 v[20];
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) v[n1] = *reinterpret_cast<const float2 *>(s + 20 * n1);
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) acc += v[n1].x + v[n1].y;
        }
        if (WIDE_STORES) {
            const int nf = f0 + 6 <= kFrames ? 6 : kFrames - f0;
            float4 *o = reinterpret_cast<float4 *>(out + (clip * kFrames + f0) * kMels);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (lane + 64 * k < nf * 20) o[lane + 64 * k] = make_float4(acc, acc + 1, acc + 2, acc + k);
        } else if (act) {
            float *o = out + (clip * kFrames + f) * kMels + t;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[10 * i] = acc + i;
        }
    }
}

template <bool LOADS, bool STORES>
__global__ __launch_bounds__(1024, 1) void pattern(const float *__restrict__ pcm, float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fl = lane / 10, t = lane - 10 * fl;
    const bool in = lane < 60;
    const uint64_t n_units = (uint64_t)kClips * kUnitsPerClip, waves = (uint64_t)gridDim.x * 16, w = (uint64_t)blockIdx.x * 16 + wave;
    const uint64_t lo = n_units * w / waves, hi = n_units * (w + 1) / waves;
    for (uint64_t u = lo; u < hi; ++u) {
        const uint64_t clip = u / kUnitsPerClip, unit = u - clip * kUnitsPerClip;
        const int f = (int)unit * 6 + fl;
        const bool act = in && f < kFrames;
        float acc = 0.0f;
        if (LOADS && act) {
            const float *s = pcm + clip * kClipLen + (uint64_t)f * kHop + 2 * t;
            float2 v[20];
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) v[n1] = *reinterpret_cast<const float2 *>(s + 20 * n1);
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) acc += v[n1].x + v[n1].y;
        }
        if (STORES && act) {
            float *o = out + (clip * kFrames + f) * kMels + t;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[10 * i] = acc + i;
        } else if (acc == 123.456f) {
            out[0] = acc;
        }
    }
}

__global__ void copy21(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n_out4) {     // reads 2 x what it writes
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = in[2 * i], b = in[2 * i + 1];
        out[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

int main() {
    const size_t n_in = (size_t)kClips * kClipLen, n_out = (size_t)kClips * kFrames * kMels;
    float *pcm, *out;
    hipMalloc(&pcm, n_in * 4 + 64); hipMalloc(&out, n_out * 4 + 64);
    hipMemset(pcm, 0, n_in * 4); hipMemset(out, 0, n_out * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, double bytes, auto f) {
        for (int i = 0; i < 200; ++i) f();                         // clocks up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 200;
        printf("%-34s %8.4f ms  %7.0f GB/s of the bytes it moves   (config 2's 982 MB in this time = %5.0f GB/s = %.3f of 8 TB/s)\n", name, ms,
               bytes / ms / 1e6, 982e6 / ms / 1e6, 982e6 / ms / 1e6 / 8000.0);
    };
    const double in_b = (double)n_in * 4, out_b = (double)n_out * 4;
    for (int rep = 0; rep < 2; ++rep) {
        run("pattern: loads only", in_b, [&] { pattern<true, false><<<256, 1024>>>(pcm, out); });
        run("pattern: stores only", out_b, [&] { pattern<false, true><<<256, 1024>>>(pcm, out); });
        run("pattern: loads + stores", in_b + out_b, [&] { pattern<true, true><<<256, 1024>>>(pcm, out); });
        run("wide loads (16 B) + kernel's stores", in_b + out_b, [&] { pattern_wide<true, false><<<256, 1024>>>(pcm, out); });
        run("kernel's loads + wide stores (16 B)", in_b + out_b, [&] { pattern_wide<false, true><<<256, 1024>>>(pcm, out); });
        run("wide loads + wide stores", in_b + out_b, [&] { pattern_wide<true, true><<<256, 1024>>>(pcm, out); });
        run("copy 2:1, 16-byte, 4096 x 256", in_b + out_b, [&] { copy21<<<4096, 256>>>((const float4 *)pcm, (float4 *)out, n_out / 4); });
    }
    return 0;
}
