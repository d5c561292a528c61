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
#include "stream_plan.hpp"

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
    // Eight rows' loads go out together (rows past the image re-read its last row), then the walk with its early exit: one load per
    // step behind the exit test was a memory round trip per row (round 3: 0.121 ms for 1024 x 80 x 1000).
    constexpr uint32_t kRows = 8;
    for (uint32_t y0 = start_y; y0 < height - 2; y0 += kRows) {
        float bv[kRows][3];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t i = 0; i < kRows; ++i) {
            const uint32_t row = y0 + 2 + i < height ? y0 + 2 + i : height - 1;
            const float *b = img + static_cast<uint64_t>(row) * width + x;
            bv[i][0] = b[0]; bv[i][1] = b[1]; bv[i][2] = b[2];
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t i = 0; i < kRows; ++i) {
            if (y0 + i >= height - 2) break;
            const double b0 = bv[i][0], b1 = bv[i][1], b2 = bv[i][2];
            if (sobel_gradient_sq(t0, t1, t2, m0, m2, b0, b1, b2) >= thr && ++count >= min_y) return true;
            t0 = m0; t1 = m1; t2 = m2;
            m0 = b0; m1 = b1; m2 = b2;
        }
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

// ---- the detector inside the streaming bank (VoiceActivityDetector::add_activity, src/vad.rs:155-205) -------------------------------
// The reference classifies, for every frame it is given, the window of the last min_x frames as a [n_mels][min_x] image: raw[x] is
// the Sobel count of window columns x..x+2, x < min_x - 2, smoothed by the +-4 vote, `active` = column 0 survives.  raw[x] of the
// window ending at frame f is a property of the three frames g..g+2, g = f - min_x + 1 + x, alone: R[g].  So a stream needs R of its
// past min_x - 2 triples (bits of `hist`: bit i = R[count - 3 - i]) and its last two mel rows; every new frame costs one Sobel walk.
struct StreamVadState {        // per stream, in HBM next to the bank's sample state
    uint64_t count;            // frames this stream has emitted (VoiceActivityDetector::frame_index)
    uint64_t hist;             // bit i = R[count - 3 - i]
};
struct VadActivity {           // melspec_vad_activity: what add_activity returns for one frame (VoiceActivity, src/vad.rs:126-135)
    uint8_t valid;             // 0: the reference returns None (fewer than min_x frames so far)
    uint8_t active;
    uint16_t leading_active_columns, active_columns, window_columns;
};
constexpr int kStreamVadMaxX = 66;     // min_x - 2 <= 64 history bits

// R for the triple whose columns are the mel rows c0, c1, c2 ([n_mels] each): classify_columns_across_frames, src/vad.rs:417-470
MS_HD bool vad_classify_triple(const float *c0, const float *c1, const float *c2, uint32_t height, int min_mel, int min_y, double thr) {
    if (min_y == 0) return true;
    const uint32_t start_y = static_cast<uint32_t>(min_mel) < height - 2 ? static_cast<uint32_t>(min_mel) : height - 2;
    int count = 0;
    for (uint32_t y = start_y; y < height - 2; ++y) {
        if (sobel_gradient_sq(c0[y], c1[y], c2[y], c0[y + 1], c2[y + 1], c0[y + 2], c1[y + 2], c2[y + 2]) >= thr && ++count >= min_y) return true;
    }
    return false;
}

// The record of the stream's frame f (0-based count of emitted frames).  rw[-j], j = 0..min_x-3, is R of the triple j back from
// the one that ends at frame f (rw[0] = R[f - 2]).
MS_HD VadActivity vad_stream_activity(uint64_t f, const uint8_t *rw, uint32_t height, int min_x) {
    VadActivity a{};
    if (min_x <= 0 || f + 1 < static_cast<uint64_t>(min_x)) return a;            // mel_buffer.len() < min_x: None
    a.valid = 1;
    if (height < 3 || min_x < 3) return a;                                        // vad_boundaries returns an empty EdgeInfo
    const int n = min_x - 2;
    uint64_t raw = 0;                                                             // bit x = raw[x] = R[f - min_x + 1 + x]
    for (int x = 0; x < n; ++x) raw |= static_cast<uint64_t>(rw[-(n - 1 - x)] != 0) << x;
    int act = 0, lead = 0;
    bool leading = true;
    for (int i = 0; i < n; ++i) {                                                 // smooth_mask(.., 4), src/vad.rs:343-360
        const int s = i >= 4 ? i - 4 : 0, e = i + 5 < n ? i + 5 : n;
        int c = 0;
        for (int q = s; q < e; ++q) c += static_cast<int>((raw >> q) & 1u);
        const bool on = c * 2 >= e - s;
        act += on;
        if (leading && on) ++lead; else leading = false;
        if (i == 0) a.active = on;
    }
    a.leading_active_columns = static_cast<uint16_t>(lead);
    a.active_columns = static_cast<uint16_t>(act);
    a.window_columns = static_cast<uint16_t>(n);
    return a;
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

// The detector stage of a push: one workgroup per pushed stream, after the mel kernel has written the stream's new rows.
// rwin = [64 triples of history][R of the new frames, a chunk at a time]; every new frame costs one Sobel walk over three rows.
struct StreamVadParams {
    const StreamEntry *entries;
    const float *rows;          // the push's output buffer, [frame][n_mels] rows at entries[i].out_off
    StreamVadState *state;      // [n_streams]
    float *prev;                // [n_streams][2][n_mels]: the last two rows every stream emitted
    VadActivity *acts;          // [total frames of the push], packed in entry order
    uint32_t n_mels;
    int min_mel, min_y, min_x;
    double thr;
};
constexpr uint32_t kStreamVadChunk = 2048;
// blockDim.x = 64 * waves.  A wave takes one new frame at a time and spreads the Sobel walk over its lanes (lane = mel row): the
// reference's early exit at the min_y-th strong row is an optimisation of `count >= min_y`, which is what the ballots add up.
__global__ __launch_bounds__(256) void stream_vad_kernel(const StreamVadParams p) {
    const StreamEntry e = p.entries[blockIdx.x];
    const uint32_t F = e.frames, tid = threadIdx.x, H = p.n_mels, nt = blockDim.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6, waves = nt >> 6;
    if (F == 0) return;
    __shared__ uint8_t rwin[64 + kStreamVadChunk];
    const StreamVadState st = p.state[e.stream];
    const float *rows = p.rows + e.out_off;
    float *pv = p.prev + static_cast<uint64_t>(e.stream) * 2 * H;
    if (tid < 64) rwin[63 - tid] = static_cast<uint8_t>((st.hist >> tid) & 1u);
    const uint32_t start_y = H >= 3 ? (static_cast<uint32_t>(p.min_mel) < H - 2 ? static_cast<uint32_t>(p.min_mel) : H - 2) : 0;
    uint32_t m = 0;
    for (uint32_t b = 0; b < F; b += kStreamVadChunk) {
        m = F - b < kStreamVadChunk ? F - b : kStreamVadChunk;
        for (uint32_t k = wave; k < m; k += waves) {
            const uint32_t q = b + k;                                      // new frame q closes the triple (q-2, q-1, q)
            bool r = false;
            if (st.count + q >= 2 && H >= 3) {
                const float *c2 = rows + static_cast<uint64_t>(q) * H;
                const float *c1 = q >= 1 ? c2 - H : pv + H;
                const float *c0 = q >= 2 ? c2 - 2 * static_cast<uint64_t>(H) : (q == 1 ? pv + H : pv);
                if (p.min_y == 0) {
                    r = true;
                } else {
                    int count = 0;
                    for (uint32_t y0 = start_y; y0 < H - 2; y0 += 64) {
                        const uint32_t y = y0 + lane;
                        bool strong = false;
                        if (y < H - 2)
                            strong = sobel_gradient_sq(c0[y], c1[y], c2[y], c0[y + 1], c2[y + 1], c0[y + 2], c1[y + 2], c2[y + 2]) >= p.thr;
                        count += __popcll(__ballot(strong));
                    }
                    r = count >= p.min_y;
                }
            }
            if (lane == 0) rwin[64 + k] = r;
        }
        __syncthreads();
        for (uint32_t k = tid; k < m; k += nt) p.acts[e.act_base + b + k] = vad_stream_activity(st.count + b + k, rwin + 64 + k, H, p.min_x);
        __syncthreads();
        if (b + m < F) {                                                   // the chunk's last 64 become the history of the next
            const uint8_t v = tid < 64 ? rwin[m + tid] : 0;
            __syncthreads();
            if (tid < 64) rwin[tid] = v;
            __syncthreads();
        }
    }
    // state for the next push: the newest 64 triples, the last two rows (every read of pv above is behind a barrier)
    if (wave == 0) {
        const uint64_t hist = __ballot((rwin[64 + m - 1 - lane] & 1u) != 0);
        if (lane == 0) p.state[e.stream] = StreamVadState{st.count + F, hist};
    }
    for (uint32_t y = tid; y < H; y += nt) {
        const float last = rows[static_cast<uint64_t>(F - 1) * H + y];
        const float before = F >= 2 ? rows[static_cast<uint64_t>(F - 2) * H + y] : pv[H + y];
        pv[y] = before;
        pv[H + y] = last;
    }
}

#endif  // __HIPCC__

}  // namespace melspec
