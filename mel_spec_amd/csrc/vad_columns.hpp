// vad_columns.hpp -- column classification of mel images, the reference's cheap consumer of the mel path
// (vad_boundaries, src/vad.rs:256-340): a 3x3 Sobel stencil over the [n_mels][width] image, a per-column count
// of strong gradients, and a +-4 majority vote along time.  Decisions, not floats, come out, so the arithmetic
// is the reference's f64 sequence on the f32 pixels widened to f64 (to_array2, src/quant.rs:168-174) and the
// masks are bit-identical to the CPU's.
//
// Traffic: the image is read once (each thread walks its column top to bottom keeping the 3x3 neighbourhood
// in registers; the two neighbouring columns are its neighbours' loads, served by L1/L2): 4 B per pixel in,
// 2 bytes per column out.
#pragma once

#include <cstdint>

#include "device_fft.hpp"

namespace melspec {

struct VadDesc {
    const float *img;        // [n_images] images, [height][width] f32 each
    uint8_t *raw;            // [n_images][mask_stride] or nullptr
    uint8_t *smoothed;       // [n_images][mask_stride]
    uint32_t *longest;       // [n_images] longest run of set columns in `smoothed`, or nullptr
    uint64_t img_stride;     // floats between images
    uint64_t mask_stride;    // bytes between masks
    uint32_t height, width, n_images;
    int min_mel, min_y;
    double thr;              // min_energy^2
};

// sobel_gradient_sq, src/vad.rs:470-486
MS_HD double sobel_gradient_sq(double tl, double tc, double tr, double ml, double mr, double bl, double bc, double br) {
    const double gx = (tr + (2.0 * mr) + br) - (tl + (2.0 * ml) + bl);
    const double gy = (bl + (2.0 * bc) + br) - (tl + (2.0 * tc) + tr);
    return (gx * gx) + (gy * gy);
}

// classify_columns_in_frame for column x (src/vad.rs:373-415)
MS_HD bool vad_classify_column(const float *img, uint32_t height, uint32_t width, uint32_t x, int min_mel, int min_y, double thr) {
    if (min_y == 0) return true;
    const uint32_t start_y = static_cast<uint32_t>(min_mel) < height - 2 ? static_cast<uint32_t>(min_mel) : height - 2;
    if (start_y >= height - 2) return false;
    const float *p = img + static_cast<uint64_t>(start_y) * width + x;
    double t0 = p[0], t1 = p[1], t2 = p[2];
    double m0 = p[width], m1 = p[width + 1], m2 = p[width + 2];
    (void)m1;
    int count = 0;
    for (uint32_t y = start_y; y < height - 2; ++y) {
        const float *b = img + static_cast<uint64_t>(y + 2) * width + x;
        const double b0 = b[0], b1 = b[1], b2 = b[2];
        if (sobel_gradient_sq(t0, t1, t2, m0, m2, b0, b1, b2) >= thr && ++count >= min_y) return true;
        t0 = m0; t1 = m1; t2 = m2;
        m0 = b0; m1 = b1; m2 = b2;
    }
    return false;
}

// smooth_mask with window 4 at index i (src/vad.rs:343-360)
MS_HD bool vad_smooth_at(const uint8_t *raw, uint32_t n, uint32_t i) {
    const uint32_t start = i >= 4 ? i - 4 : 0, end = i + 5 < n ? i + 5 : n;
    uint32_t c = 0;
    for (uint32_t k = start; k < end; ++k) c += raw[k];
    return c * 2 >= end - start;
}

#if defined(__HIPCC__)

// one thread per (image, column); grid.x = n_images * blocks_per_image
__global__ __launch_bounds__(256) void vad_raw_kernel(const VadDesc d, uint32_t blocks_per_image, uint8_t *raw) {
    const uint32_t image = blockIdx.x / blocks_per_image, blk = blockIdx.x - image * blocks_per_image;
    const uint32_t x = blk * 256 + threadIdx.x, n = d.width - 2;
    if (x >= n) return;
    raw[image * d.mask_stride + x] = vad_classify_column(d.img + image * d.img_stride, d.height, d.width, x, d.min_mel, d.min_y, d.thr);
}

__global__ __launch_bounds__(256) void vad_smooth_kernel(const VadDesc d, uint32_t blocks_per_image, const uint8_t *raw) {
    const uint32_t image = blockIdx.x / blocks_per_image, blk = blockIdx.x - image * blocks_per_image;
    const uint32_t x = blk * 256 + threadIdx.x, n = d.width - 2;
    if (x >= n) return;
    d.smoothed[image * d.mask_stride + x] = vad_smooth_at(raw + image * d.mask_stride, n, x);
}

// longest run of set columns per image: one wave per image, lanes take contiguous pieces, then a serial stitch by lane 0
__global__ __launch_bounds__(64) void vad_run_kernel(const VadDesc d) {
    const uint32_t image = blockIdx.x, n = d.width - 2, lane = threadIdx.x;
    const uint8_t *m = d.smoothed + image * d.mask_stride;
    const uint32_t per = (n + 63) / 64, lo = lane * per, hi = lo + per < n ? lo + per : n;
    // piece summary: length of the leading run, of the trailing run, the best inside, and whether it is all set
    uint32_t lead = 0, trail = 0, best = 0, cur = 0;
    bool all = true;
    for (uint32_t i = lo; i < hi; ++i) {
        if (m[i]) { ++cur; if (all) lead = cur; if (cur > best) best = cur; } else { cur = 0; all = false; }
    }
    trail = cur;
    __shared__ uint32_t s_lead[64], s_trail[64], s_best[64], s_len[64];
    __shared__ uint8_t s_all[64];
    s_lead[lane] = lead; s_trail[lane] = trail; s_best[lane] = best; s_len[lane] = hi > lo ? hi - lo : 0; s_all[lane] = all;
    __syncthreads();
    if (lane == 0) {
        uint32_t b = 0, carry = 0;
        for (int k = 0; k < 64; ++k) {
            if (s_len[k] == 0) continue;
            if (s_best[k] > b) b = s_best[k];
            if (carry + s_lead[k] > b) b = carry + s_lead[k];
            carry = s_all[k] ? carry + s_len[k] : s_trail[k];
        }
        if (carry > b) b = carry;
        d.longest[image] = b;
    }
}

#endif  // __HIPCC__

}  // namespace melspec
