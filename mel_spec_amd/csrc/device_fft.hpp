// device_fft.hpp -- small in-register forward DFTs (f32) used by the fused frame kernels.
//
// Everything here is straight-line code on named/statically indexed values so that on
// gfx950 it lives entirely in VGPRs.  Sizes 4,5 are butterflies; 10 = 2x5 and 20 = 4x5 use
// the prime-factor (Good-Thomas) index maps, so they need no internal twiddle multiplies;
// 8 and 16 are Cooley-Tukey with literal twiddles.
//
// Forward transform convention (what rustfft's plan_fft_forward computes; call sites
// src/stft.rs:99-110): X[k] = sum_n x[n] * exp(-2*pi*i*n*k/N), unnormalised.
//
// The file is plain C++17: hipcc compiles it for the device, and tests/emu compiles the
// same source for the host to check the index algebra on the CPU.
#pragma once

#if defined(__HIPCC__)
#define MS_DEV __device__ __forceinline__
#define MS_HD __host__ __device__ __forceinline__
#else
#define MS_DEV inline
#define MS_HD inline
#endif

namespace melspec {

struct cf {
    float re, im;
};
struct alignas(8) f2 {
    float x, y;
};
struct alignas(16) f4 {
    float x, y, z, w;
};

MS_DEV cf operator+(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
MS_DEV cf operator-(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
MS_DEV cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
MS_DEV cf mul_neg_i(cf a) { return {a.im, -a.re}; }   // a * (-i)
MS_DEV cf cscale(cf a, float s) { return {a.re * s, a.im * s}; }

MS_DEV void bf2(cf &a, cf &b) {
    const cf t = a - b;
    a = a + b;
    b = t;
}

// X0=a+b+c+d, X1=a-ib-c+id, X2=a-b+c-d, X3=a+ib-c-id
MS_DEV void bf4(cf &x0, cf &x1, cf &x2, cf &x3) {
    const cf s0 = x0 + x2, s1 = x0 - x2, s2 = x1 + x3, s3 = mul_neg_i(x1 - x3);
    x0 = s0 + s2;
    x1 = s1 + s3;
    x2 = s0 - s2;
    x3 = s1 - s3;
}

MS_DEV void bf5(cf &x0, cf &x1, cf &x2, cf &x3, cf &x4) {
    constexpr float c1 = 0.30901699437494742f;    // cos(2pi/5)
    constexpr float c2 = -0.80901699437494742f;   // cos(4pi/5)
    constexpr float s1 = 0.95105651629515357f;    // sin(2pi/5)
    constexpr float s2 = 0.58778525229247313f;    // sin(4pi/5)
    const cf t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    const cf m1 = {x0.re + c1 * t1.re + c2 * t2.re, x0.im + c1 * t1.im + c2 * t2.im};
    const cf m2 = {x0.re + c2 * t1.re + c1 * t2.re, x0.im + c2 * t1.im + c1 * t2.im};
    const cf u1 = {s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im};
    const cf u2 = {s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im};
    x0 = x0 + t1 + t2;
    x1 = {m1.re + u1.im, m1.im - u1.re};   // m1 - i*u1
    x4 = {m1.re - u1.im, m1.im + u1.re};   // m1 + i*u1
    x2 = {m2.re + u2.im, m2.im - u2.re};
    x3 = {m2.re - u2.im, m2.im + u2.re};
}

// 10 = 2 x 5, Good-Thomas: n = (5*n1 + 2*n2) mod 10, k = (5*k1 + 6*k2) mod 10.
MS_DEV void fft10(cf (&x)[10]) {
    cf a[2][5];
#pragma unroll
    for (int n1 = 0; n1 < 2; ++n1)
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) a[n1][n2] = x[(5 * n1 + 2 * n2) % 10];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) bf2(a[0][n2], a[1][n2]);
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1) bf5(a[k1][0], a[k1][1], a[k1][2], a[k1][3], a[k1][4]);
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) x[(5 * k1 + 6 * k2) % 10] = a[k1][k2];
}

// 20 = 4 x 5, Good-Thomas: n = (5*n1 + 4*n2) mod 20, k = (5*k1 + 16*k2) mod 20.
MS_DEV void fft20(cf (&x)[20]) {
    cf a[4][5];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1)
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) a[n1][n2] = x[(5 * n1 + 4 * n2) % 20];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) bf4(a[0][n2], a[1][n2], a[2][n2], a[3][n2]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) bf5(a[k1][0], a[k1][1], a[k1][2], a[k1][3], a[k1][4]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) x[(5 * k1 + 16 * k2) % 20] = a[k1][k2];
}

// 8 = 2 x 4 Cooley-Tukey (n = 4*n1 + n2, k = k1 + 2*k2), twiddles W_8^{n2*k1}.
MS_DEV void fft8(cf (&x)[8]) {
    constexpr float r = 0.70710678118654752f;
    // step 1: 4 radix-2 over n1 (inputs x[n2], x[n2+4])
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) bf2(x[n2], x[n2 + 4]);
    // twiddle the k1=1 half by W_8^{n2}
    x[5] = {r * (x[5].re + x[5].im), r * (x[5].im - x[5].re)};   // * (r - i r)
    x[6] = mul_neg_i(x[6]);
    x[7] = {r * (x[7].im - x[7].re), -r * (x[7].re + x[7].im)};  // * (-r - i r)
    // step 2: radix-4 over n2 for each k1 -> X[k1 + 2*k2]
    bf4(x[0], x[1], x[2], x[3]);
    bf4(x[4], x[5], x[6], x[7]);
    const cf y0 = x[0], y1 = x[4], y2 = x[1], y3 = x[5], y4 = x[2], y5 = x[6], y6 = x[3], y7 = x[7];
    x[0] = y0; x[1] = y1; x[2] = y2; x[3] = y3; x[4] = y4; x[5] = y5; x[6] = y6; x[7] = y7;
}

// 16 = 4 x 4 Cooley-Tukey: n = 4*n1 + n2, k = k1 + 4*k2, twiddles W_16^{n2*k1}.
MS_DEV void fft16(cf (&x)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;   // cos/sin(pi/8)
    constexpr float r = 0.70710678118654752f;
    // step 1: for each n2, radix-4 over n1: inputs x[n2], x[n2+4], x[n2+8], x[n2+12] -> A[k1][n2] in place
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) bf4(x[n2], x[n2 + 4], x[n2 + 8], x[n2 + 12]);
    // twiddles W_16^{n2*k1}; A[k1][n2] sits at x[4*k1 + n2]
    // k1 = 1: W^1, W^2, W^3
    x[5] = cmul(x[5], cf{c1, -s1});
    x[6] = cf{r * (x[6].re + x[6].im), r * (x[6].im - x[6].re)};
    x[7] = cmul(x[7], cf{s1, -c1});
    // k1 = 2: W^2, W^4, W^6
    x[9] = cf{r * (x[9].re + x[9].im), r * (x[9].im - x[9].re)};
    x[10] = mul_neg_i(x[10]);
    x[11] = cf{r * (x[11].im - x[11].re), -r * (x[11].re + x[11].im)};
    // k1 = 3: W^3, W^6, W^9
    x[13] = cmul(x[13], cf{s1, -c1});
    x[14] = cf{r * (x[14].im - x[14].re), -r * (x[14].re + x[14].im)};
    x[15] = cmul(x[15], cf{-c1, s1});
    // step 2: for each k1, radix-4 over n2 -> X[k1 + 4*k2] at x[4*k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) bf4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
    // transpose to natural order
    cf y[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) y[k1 + 4 * k2] = x[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}

}  // namespace melspec
