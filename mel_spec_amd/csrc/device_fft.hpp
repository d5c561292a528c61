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
// __builtin_amdgcn_sched_barrier mask behind a block of global loads: LDS operations and SALU may cross it, VALU and memory
// instructions may not -- the loads stay in front of the arithmetic that uses them, the table reads of that arithmetic may start early
#ifndef MS_SCHED_LOADS_FIRST
#define MS_SCHED_LOADS_FIRST 0x108
#endif
#define MS_HD __host__ __device__ __forceinline__
#else
#define MS_DEV inline
#define MS_HD inline
#endif
// The table blob is read-only while the kernels run and never overlaps a wave's slice; without the qualifier every table
// read stays behind the preceding slice write (one LDS round trip per twiddle: ds_write / ds_read / s_waitcnt lgkmcnt(0)
// triples in the ISA, tools/isa_schedule.py).  Used where two waves per SIMD cannot hide that latency.
#define MS_RESTRICT __restrict__

namespace melspec {

template <class T>
struct cpx {
    T re, im;
};
using cf = cpx<float>;
using cd = cpx<double>;
struct alignas(8) f2 {
    float x, y;
};
struct alignas(16) f4 {
    float x, y, z, w;
};
struct alignas(16) d2 {
    double x, y;
};

// 8 / 16 bytes as ONE load: a struct load is split into scalar loads by the optimiser, and the LDS load/store pass then re-pairs
// them as it likes -- the (rise, fall) weight pairs of the mel phase came out as ds_read2_b32 of two different rows (4 LDS cycles
// for 8 bytes where ds_read_b64 takes 2), the window taps as ds_read2_b64 (8 cycles for 16 bytes where ds_read_b128 takes 4;
// MI355X_MICROARCH.md, LDS table).  A vector-typed load stays whole.
MS_DEV f2 ld2(const float *p) {
#if defined(__HIPCC__)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = *reinterpret_cast<const v2f *>(p);
    return f2{v.x, v.y};
#else
    return f2{p[0], p[1]};
#endif
}
// the same, and not to be paired with a neighbour into ds_read2_b64 (8 LDS cycles for 16 bytes, half the rate of two ds_read_b64)
MS_DEV f2 ld2_single(const float *p) {
#if defined(__HIPCC__)
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef const volatile __attribute__((address_space(3))) v2f *lds_ptr;      // explicit: a volatile access through a generic pointer stays a flat load
    const v2f v = *(lds_ptr)(p);
    return f2{v.x, v.y};
#else
    return f2{p[0], p[1]};
#endif
}
MS_DEV f4 ld4(const float *p) {
#if defined(__HIPCC__)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = *reinterpret_cast<const v4f *>(p);
    return f4{v.x, v.y, v.z, v.w};
#else
    return f4{p[0], p[1], p[2], p[3]};
#endif
}

// one complex value as one 8-byte (f32) / 16-byte (f64) access
template <class T> struct PairOf;
template <> struct PairOf<float> { using type = f2; };
template <> struct PairOf<double> { using type = d2; };
template <class T> MS_DEV cpx<T> ldc(const T *p) {
    const typename PairOf<T>::type v = *reinterpret_cast<const typename PairOf<T>::type *>(p);
    return {v.x, v.y};
}
template <class T> MS_DEV void stc(T *p, cpx<T> v) {
    *reinterpret_cast<typename PairOf<T>::type *>(p) = typename PairOf<T>::type{v.re, v.im};
}

// a*b rounded to f32 on its own: the product must not be contracted into an FMA with the following
// subtraction (HIP's __fmul_rn is a plain multiply and does get contracted), because the reference's
// two roundings are visible in the weakest bins of loud frames (up to 7e-2 of a bin on jfk_f32le.wav).
MS_HD float f32_mul_rn(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float p = a * b;
    asm("" : "+v"(p));      // opaque to the optimiser, but not `volatile`: a volatile asm is a scheduling boundary, and 26 of
    return p;               // them per lane serialised the 13 sample loads of the NeMo phase 1 (1.51 -> 0.9 ms per launch)
#else
    volatile float p = a * b;
    return p;
#endif
}
// `v`, as a value the optimiser has to take from here: what is derived from it is no loop invariant.  Lane constants hoisted out of
// the unit loop are what the register allocator spills first in the twelve-wave kernels, and a reload inside the loop waits (vmcnt)
// for the previous unit's stores; a v_xor per address is cheaper.  A scheduling boundary: once per phase, not per access.
MS_DEV int fresh_lane_value(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}
// a / b correctly rounded in f32 (the reference's plain `/`)
MS_HD float f32_div_rn(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}
template <class T> MS_DEV cpx<T> operator+(cpx<T> a, cpx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <class T> MS_DEV cpx<T> operator-(cpx<T> a, cpx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <class T> MS_DEV cpx<T> cmul(cpx<T> a, cpx<T> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <class T> MS_DEV cpx<T> mul_neg_i(cpx<T> a) { return {a.im, -a.re}; }   // a * (-i)

template <class T>
MS_DEV void bf2(cpx<T> &a, cpx<T> &b) {
    const cpx<T> t = a - b;
    a = a + b;
    b = t;
}

// X0=a+b+c+d, X1=a-ib-c+id, X2=a-b+c-d, X3=a+ib-c-id
template <class T>
MS_DEV void bf4(cpx<T> &x0, cpx<T> &x1, cpx<T> &x2, cpx<T> &x3) {
    const cpx<T> s0 = x0 + x2, s1 = x0 - x2, s2 = x1 + x3, s3 = mul_neg_i(x1 - x3);
    x0 = s0 + s2;
    x1 = s1 + s3;
    x2 = s0 - s2;
    x3 = s1 - s3;
}

template <class T>
MS_DEV void bf5(cpx<T> &x0, cpx<T> &x1, cpx<T> &x2, cpx<T> &x3, cpx<T> &x4) {
    constexpr T c1 = static_cast<T>(0.30901699437494742410);    // cos(2pi/5)
    constexpr T c2 = static_cast<T>(-0.80901699437494742410);   // cos(4pi/5)
    constexpr T s1 = static_cast<T>(0.95105651629515357212);    // sin(2pi/5)
    constexpr T s2 = static_cast<T>(0.58778525229247312917);    // sin(4pi/5)
    const cpx<T> t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    const cpx<T> m1 = {x0.re + c1 * t1.re + c2 * t2.re, x0.im + c1 * t1.im + c2 * t2.im};
    const cpx<T> m2 = {x0.re + c2 * t1.re + c1 * t2.re, x0.im + c2 * t1.im + c1 * t2.im};
    const cpx<T> u1 = {s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im};
    const cpx<T> u2 = {s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im};
    x0 = x0 + t1 + t2;
    x1 = {m1.re + u1.im, m1.im - u1.re};   // m1 - i*u1
    x4 = {m1.re - u1.im, m1.im + u1.re};   // m1 + i*u1
    x2 = {m2.re + u2.im, m2.im - u2.re};
    x3 = {m2.re - u2.im, m2.im + u2.re};
}

// 10 = 2 x 5, Good-Thomas: n = (5*n1 + 2*n2) mod 10, k = (5*k1 + 6*k2) mod 10.
template <class T>
MS_DEV void fft10(cpx<T> (&x)[10]) {
    cpx<T> a[2][5];
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
template <class T>
MS_DEV void fft20(cpx<T> (&x)[20]) {
    cpx<T> a[4][5];
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
template <class T>
MS_DEV void fft8(cpx<T> (&x)[8]) {
    constexpr T r = static_cast<T>(0.70710678118654752440);
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) bf2(x[n2], x[n2 + 4]);
    x[5] = {r * (x[5].re + x[5].im), r * (x[5].im - x[5].re)};   // * (r - i r)
    x[6] = mul_neg_i(x[6]);
    x[7] = {r * (x[7].im - x[7].re), -r * (x[7].re + x[7].im)};  // * (-r - i r)
    bf4(x[0], x[1], x[2], x[3]);
    bf4(x[4], x[5], x[6], x[7]);
    const cpx<T> y0 = x[0], y1 = x[4], y2 = x[1], y3 = x[5], y4 = x[2], y5 = x[6], y6 = x[3], y7 = x[7];
    x[0] = y0; x[1] = y1; x[2] = y2; x[3] = y3; x[4] = y4; x[5] = y5; x[6] = y6; x[7] = y7;
}

// 16 = 4 x 4 Cooley-Tukey: n = 4*n1 + n2, k = k1 + 4*k2, twiddles W_16^{n2*k1}.
template <class T>
MS_DEV void fft16(cpx<T> (&x)[16]) {
    constexpr T c1 = static_cast<T>(0.92387953251128675613), s1 = static_cast<T>(0.38268343236508977173);   // cos/sin(pi/8)
    constexpr T r = static_cast<T>(0.70710678118654752440);
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) bf4(x[n2], x[n2 + 4], x[n2 + 8], x[n2 + 12]);
    // twiddles W_16^{n2*k1}; A[k1][n2] sits at x[4*k1 + n2]
    x[5] = cmul(x[5], cpx<T>{c1, -s1});
    x[6] = cpx<T>{r * (x[6].re + x[6].im), r * (x[6].im - x[6].re)};
    x[7] = cmul(x[7], cpx<T>{s1, -c1});
    x[9] = cpx<T>{r * (x[9].re + x[9].im), r * (x[9].im - x[9].re)};
    x[10] = mul_neg_i(x[10]);
    x[11] = cpx<T>{r * (x[11].im - x[11].re), -r * (x[11].re + x[11].im)};
    x[13] = cmul(x[13], cpx<T>{s1, -c1});
    x[14] = cpx<T>{r * (x[14].im - x[14].re), -r * (x[14].re + x[14].im)};
    x[15] = cmul(x[15], cpx<T>{-c1, s1});
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) bf4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
    cpx<T> y[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) y[k1 + 4 * k2] = x[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}

}  // namespace melspec
