// tga_quant.hpp -- 8-bit quantisation of mel images and the TGA container (src/quant.rs), the
// wire/disk format the reference puts right after the mel path (save_tga_8bit / tga_8bit /
// parse_tga_8bit, src/quant.rs:15-88; quantize / dequantize, :140-165).
//
// An "item" is one chunk of one image: images are [rows][width] f32, row-major (what
// interleave_frames(.., major_column_order = false, ..) makes, rows = n_mels); tga_8bit cuts an image
// into <= 65535-column chunks (chunk_frames_into_strides, src/quant.rs:100-136) and every chunk is
// quantised with its own {min,max} and gets its own 26-byte header.  Two passes over HBM:
//   1. quant_minmax_kernel -- per-item min/max.  min/max are exact and order-independent, so the
//      reduction uses integer atomics on an order-preserving key of the f32 bit pattern and stays
//      bit-reproducible; NaNs are skipped like f32::min / f32::max do.
//   2. quant_encode_kernel -- one aligned output dword (4 pixels or header bytes) per thread step.
// All arithmetic is the reference's f32 sequence (subtract, multiply, round-half-away, clamp), so the
// bytes are identical to the CPU's, not just close.  Algorithmic bytes per pixel: 4 read + 1 written;
// this implementation reads the image twice (9 B/pixel).
//
// The per-thread functions are plain C++ so that tests/emu can run them on the host.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "device_fft.hpp"

namespace melspec {

constexpr uint32_t kTgaHeader = 26;          // 18-byte TARGA header + 8-byte ID field {min,max}
constexpr uint32_t kTgaMaxWidth = 65535;     // u16::MAX, the chunk width of tga_8bit (src/quant.rs:31)

struct QuantDesc {
    const float *img;        // encode: source images / decode: unused
    float *img_out;          // decode: destination images
    uint8_t *blob;           // encode: destination / decode: source
    uint32_t *keys;          // [items][2] ordered keys of {min,max} (encode scratch)
    float *ranges;           // header == 0 only: [items][2] {min,max}, written by encode, read by decode
    uint64_t img_stride;     // floats between images
    uint64_t blob_stride;    // bytes between images
    uint64_t chunk_stride;   // bytes between the chunk blobs of one image
    uint32_t rows, width;    // image shape
    uint32_t chunk_w;        // columns per chunk (== width when there is one chunk)
    uint32_t chunks;         // chunks per image
    uint32_t n_images;
    uint32_t header;         // kTgaHeader, or 0 for the bare quantize()/dequantize()
    uint32_t vec;            // 1: one chunk per image and 16-byte aligned rows of pixels -> vector loads
};

// Order-preserving map f32 -> u32 (for non-NaN values): a < b  <=>  key(a) < key(b).
MS_HD uint32_t ordered_key(float v) {
    uint32_t b;
    std::memcpy(&b, &v, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
MS_HD float key_to_float(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float v;
    std::memcpy(&v, &b, 4);
    return v;
}
constexpr uint32_t kKeyPosInf = 0xff800000u;   // ordered_key(+inf): the fold's start value for min
constexpr uint32_t kKeyNegInf = 0x007fffffu;   // ordered_key(-inf): start value for max

MS_HD uint32_t chunk_cols(const QuantDesc &d, uint32_t c) {
    const uint32_t x0 = c * d.chunk_w;
    return d.width - x0 < d.chunk_w ? d.width - x0 : d.chunk_w;
}
// pixel idx of chunk c (row-major within the chunk) -> float index inside the image
MS_HD uint64_t image_index(const QuantDesc &d, uint32_t c, uint32_t cw, uint64_t idx) {
    if (d.chunks == 1) return idx;
    const uint64_t r = idx / cw;
    return r * d.width + static_cast<uint64_t>(c) * d.chunk_w + (idx - r * cw);
}

// src/quant.rs:147-150
MS_HD uint32_t quantize_px(float v, float mn, float scale) {
    const float dlt = v - mn;
    const float p = f32_mul_rn(dlt, scale);
    const float r = fminf(fmaxf(roundf(p), 0.0f), 255.0f);
    return static_cast<uint32_t>(r);
}
// src/quant.rs:160-162
MS_HD float dequantize_px(uint32_t q, float mn, float scale) {
    const float p = f32_mul_rn(static_cast<float>(q), scale);
    return p + mn;
}

// byte p (< 26) of the header tga_8bit_data builds (src/quant.rs:45-57)
MS_HD uint32_t tga_header_byte(uint32_t p, uint32_t width, uint32_t height, float mn, float mx) {
    uint32_t bits;
    switch (p) {
        case 0: return 8;                       // ID length
        case 2: return 3;                       // uncompressed black-and-white
        case 12: return width & 0xffu;
        case 13: return (width >> 8) & 0xffu;
        case 14: return height & 0xffu;
        case 15: return (height >> 8) & 0xffu;
        case 16: return 8;                      // bits per pixel
        case 18: case 19: case 20: case 21:
            std::memcpy(&bits, &mn, 4);
            return (bits >> (8 * (p - 18))) & 0xffu;
        case 22: case 23: case 24: case 25:
            std::memcpy(&bits, &mx, 4);
            return (bits >> (8 * (p - 22))) & 0xffu;
        default: return 0;
    }
}

// One aligned dword of item (image, chunk): bytes 4*dw .. 4*dw+3 of its blob.
MS_HD uint32_t encode_dword(const QuantDesc &d, const float *img, uint32_t c, uint32_t cw, uint64_t npx, uint64_t dw,
                            float mn, float mx, float scale) {
    const uint64_t p0 = 4 * dw;
    if (p0 >= d.header && p0 - d.header + 4 <= npx && d.vec) {
        const uint64_t i0 = p0 - d.header;          // == 2 mod 4 with the TGA header, == 0 mod 4 without
        const f2 a = *reinterpret_cast<const f2 *>(img + i0), b = *reinterpret_cast<const f2 *>(img + i0 + 2);
        return quantize_px(a.x, mn, scale) | (quantize_px(a.y, mn, scale) << 8) | (quantize_px(b.x, mn, scale) << 16) |
               (quantize_px(b.y, mn, scale) << 24);
    }
    uint32_t out = 0;
    for (uint32_t k = 0; k < 4; ++k) {
        const uint64_t p = p0 + k;
        uint32_t byte = 0;
        if (p < d.header) byte = tga_header_byte(static_cast<uint32_t>(p), cw, d.rows, mn, mx);
        else if (p - d.header < npx) byte = quantize_px(img[image_index(d, c, cw, p - d.header)], mn, scale);
        out |= byte << (8 * k);
    }
    return out;
}

// The 4 pixels of blob dword dw back to f32 (parse_tga_8bit, src/quant.rs:66-88).
MS_HD void decode_dword(const QuantDesc &d, float *img, uint32_t c, uint32_t cw, uint64_t npx, uint64_t dw, uint32_t word,
                        float mn, float scale) {
    const uint64_t p0 = 4 * dw;
    if (p0 >= d.header && p0 - d.header + 4 <= npx && d.vec) {
        const uint64_t i0 = p0 - d.header;
        *reinterpret_cast<f2 *>(img + i0) = f2{dequantize_px(word & 0xffu, mn, scale), dequantize_px((word >> 8) & 0xffu, mn, scale)};
        *reinterpret_cast<f2 *>(img + i0 + 2) =
            f2{dequantize_px((word >> 16) & 0xffu, mn, scale), dequantize_px(word >> 24, mn, scale)};
        return;
    }
    for (uint32_t k = 0; k < 4; ++k) {
        const uint64_t p = p0 + k;
        if (p >= d.header && p - d.header < npx)
            img[image_index(d, c, cw, p - d.header)] = dequantize_px((word >> (8 * k)) & 0xffu, mn, scale);
    }
}

constexpr int kQuantThreads = 256;
constexpr int kQuantMinmaxLoads = 16;                    // min/max pass: float4 loads per thread, all in flight together
constexpr int kQuantPxPerBlock = kQuantThreads * 4 * kQuantMinmaxLoads;
constexpr int kQuantDwPerBlock = kQuantThreads * 4;      // encode/decode: 4 dwords per thread

MS_HD uint64_t quant_item_pixels(const QuantDesc &d, uint32_t c) { return static_cast<uint64_t>(d.rows) * chunk_cols(d, c); }

#if defined(__HIPCC__)

__global__ void quant_init_keys_kernel(uint32_t *keys, uint32_t items) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < items) { keys[2 * i] = kKeyPosInf; keys[2 * i + 1] = kKeyNegInf; }
}

// The first pass when the mel kernel has left {smallest, largest} biased value of every work unit behind (BatchDesc::d_unit_ext,
// melspec_tga_encode_pcm_uniform_device): one wave per image folds its units' records -- 8 bytes per 5 / 6 frames instead of the
// image -- into the keys the encoding pass reads.  stored value = biased * 0.25 - 3 (six_out / wave_out), monotonic in the bias.
__global__ __launch_bounds__(64) void quant_keys_from_units_kernel(const int *unit_ext, uint32_t units_per_image, uint32_t n_images, uint32_t *keys) {
    const uint32_t image = blockIdx.x;
    if (image >= n_images) return;
    const int *e = unit_ext + 2 * static_cast<uint64_t>(image) * units_per_image;
    int lo = 0x7fffffff, hi = 0;
    for (uint32_t u = threadIdx.x; u < units_per_image; u += 64) {
        const int a = e[2 * u], b = e[2 * u + 1];
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if (threadIdx.x == 0) {
        keys[2 * image] = ordered_key(__int_as_float(lo) * 0.25f - 3.0f);
        keys[2 * image + 1] = ordered_key(__int_as_float(hi) * 0.25f - 3.0f);
    }
}

// grid.x = items * blocks_per_item; block b of an item reduces pixels [b * kQuantPxPerBlock, (b + 1) * kQuantPxPerBlock).
// A block inside the item issues its sixteen 16-byte loads per thread unconditionally and together (round 3's form had each of four
// loads behind its own range test -- four serial round trips per thread -- and two atomics per WAVE: 2.8 TB/s; a read-only pass should
// run at the copy rate), folds its waves through LDS and sends one atomic pair.
__global__ __launch_bounds__(kQuantThreads) void quant_minmax_kernel(const QuantDesc d, uint32_t blocks_per_item) {
    const uint32_t item = blockIdx.x / blocks_per_item, blk = blockIdx.x - item * blocks_per_item;
    const uint32_t image = item / d.chunks, c = item - image * d.chunks;
    const uint32_t cw = chunk_cols(d, c);
    const uint64_t npx = static_cast<uint64_t>(d.rows) * cw;
    const float *img = d.img + image * d.img_stride;
    const uint64_t base = static_cast<uint64_t>(blk) * kQuantPxPerBlock;
    float mn = INFINITY, mx = -INFINITY;
    if (d.vec && base + kQuantPxPerBlock <= npx) {
        f4 v[kQuantMinmaxLoads];
#pragma unroll
        for (int k = 0; k < kQuantMinmaxLoads; ++k) v[k] = *reinterpret_cast<const f4 *>(img + base + (static_cast<uint64_t>(k) * kQuantThreads + threadIdx.x) * 4);
#pragma unroll
        for (int k = 0; k < kQuantMinmaxLoads; ++k) {
            mn = fminf(fminf(mn, v[k].x), fminf(v[k].y, fminf(v[k].z, v[k].w)));
            mx = fmaxf(fmaxf(mx, v[k].x), fmaxf(v[k].y, fmaxf(v[k].z, v[k].w)));
        }
    } else {
        for (int k = 0; k < kQuantMinmaxLoads; ++k) {
            const uint64_t i0 = base + (static_cast<uint64_t>(k) * kQuantThreads + threadIdx.x) * 4;
            if (i0 + 4 <= npx && d.vec) {
                const f4 v = *reinterpret_cast<const f4 *>(img + i0);
                mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
                mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
            } else {
                for (uint64_t i = i0; i < i0 + 4 && i < npx; ++i) {
                    const float v = img[image_index(d, c, cw, i)];
                    mn = fminf(mn, v);
                    mx = fmaxf(mx, v);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __shared__ float part[2 * (kQuantThreads / 64)];
    if ((threadIdx.x & 63) == 0) { part[2 * (threadIdx.x >> 6)] = mn; part[2 * (threadIdx.x >> 6) + 1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kQuantThreads / 64; ++w) { mn = fminf(mn, part[2 * w]); mx = fmaxf(mx, part[2 * w + 1]); }
        // a block that saw only NaNs (or nothing) still holds the fold's start values: no-ops for the atomics
        atomicMin(d.keys + 2 * item, ordered_key(mn));
        atomicMax(d.keys + 2 * item + 1, ordered_key(mx));
    }
}

__global__ __launch_bounds__(kQuantThreads) void quant_encode_kernel(const QuantDesc d, uint32_t blocks_per_item) {
    const uint32_t item = blockIdx.x / blocks_per_item, blk = blockIdx.x - item * blocks_per_item;
    const uint32_t image = item / d.chunks, c = item - image * d.chunks;
    const uint32_t cw = chunk_cols(d, c);
    const uint64_t npx = static_cast<uint64_t>(d.rows) * cw;
    const uint64_t ndw = (d.header + npx + 3) / 4;
    const float *img = d.img + image * d.img_stride;
    uint32_t *out = reinterpret_cast<uint32_t *>(d.blob + image * d.blob_stride + c * d.chunk_stride);
    const float mn = key_to_float(d.keys[2 * item]), mx = key_to_float(d.keys[2 * item + 1]);
    const float scale = f32_div_rn(255.0f, mx - mn);                       // src/quant.rs:145
    if (d.ranges && blk == 0 && threadIdx.x == 0) { d.ranges[2 * item] = mn; d.ranges[2 * item + 1] = mx; }
    const uint64_t dw0 = static_cast<uint64_t>(blk) * kQuantDwPerBlock + threadIdx.x;
    const uint64_t dw3 = dw0 + 3 * kQuantThreads;
    if (d.vec && 4 * dw0 >= d.header && 4 * dw3 - d.header + 4 <= npx) {
        // all four dwords of this thread are whole pixel quads: the eight loads first, then the arithmetic (a load inside
        // each `if` is a memory round trip of its own)
        f2 a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *src = img + (4 * (dw0 + static_cast<uint64_t>(k) * kQuantThreads) - d.header);
            a[k] = *reinterpret_cast<const f2 *>(src);
            b[k] = *reinterpret_cast<const f2 *>(src + 2);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            out[dw0 + static_cast<uint64_t>(k) * kQuantThreads] =
                quantize_px(a[k].x, mn, scale) | (quantize_px(a[k].y, mn, scale) << 8) | (quantize_px(b[k].x, mn, scale) << 16) |
                (quantize_px(b[k].y, mn, scale) << 24);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t dw = dw0 + static_cast<uint64_t>(k) * kQuantThreads;
        if (dw < ndw) out[dw] = encode_dword(d, img, c, cw, npx, dw, mn, mx, scale);
    }
}

__global__ __launch_bounds__(kQuantThreads) void quant_decode_kernel(const QuantDesc d, uint32_t blocks_per_item) {
    const uint32_t item = blockIdx.x / blocks_per_item, blk = blockIdx.x - item * blocks_per_item;
    const uint32_t image = item / d.chunks, c = item - image * d.chunks;
    const uint32_t cw = chunk_cols(d, c);
    const uint64_t npx = static_cast<uint64_t>(d.rows) * cw;
    const uint64_t ndw = (d.header + npx + 3) / 4;
    float *img = d.img_out + image * d.img_stride;
    const uint8_t *blob = d.blob + image * d.blob_stride + c * d.chunk_stride;
    float mn, mx;
    if (d.header) {                                                        // bytes 18..25, src/quant.rs:73-80
        uint32_t a = 0, b = 0;
        for (int k = 0; k < 4; ++k) { a |= static_cast<uint32_t>(blob[18 + k]) << (8 * k); b |= static_cast<uint32_t>(blob[22 + k]) << (8 * k); }
        mn = __uint_as_float(a); mx = __uint_as_float(b);
    } else {
        mn = d.ranges[2 * item]; mx = d.ranges[2 * item + 1];
    }
    const float scale = f32_div_rn(mx - mn, 255.0f);                       // src/quant.rs:158
    const uint32_t *words = reinterpret_cast<const uint32_t *>(blob);
    const uint64_t dw0 = static_cast<uint64_t>(blk) * kQuantDwPerBlock + threadIdx.x;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {           // the four loads first
        const uint64_t dw = dw0 + static_cast<uint64_t>(k) * kQuantThreads;
        w[k] = words[dw < ndw ? dw : 0];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t dw = dw0 + static_cast<uint64_t>(k) * kQuantThreads;
        if (dw < ndw) decode_dword(d, img, c, cw, npx, dw, w[k], mn, scale);
    }
}

#endif  // __HIPCC__

}  // namespace melspec
