// aux_kernels.hpp -- the small kernels around the path: streaming state (scatter / carry), the device-side ragged planner, the
// synthetic PCM generator of SURVEY 8(d).
#pragma once
#include "kernels_common.hpp"
#include "stream_plan.hpp"
#include "pow2_wave.hpp"

namespace melspec {

// ---- streaming state (Spectrogram::add, src/stft.rs:48-86; RingBuffer::maybe_mel, src/rb.rs:86-121) ----
// Every live stream owns a slot of `stride` floats: [carry, right-aligned so that it ends at `in_off`]
// [the next chunk, always at in_off].  The carry is the reference's hop_buf history (n_fft - hop samples)
// plus the samples RingBuffer has accumulated towards the next hop (< hop).  Frames are computed in place
// by the batch kernels on carry ++ chunk; afterwards the tail of that span becomes the new carry.
// copies host-pushed chunks (one flat staging buffer) into the slots; optionally zero-pads (flush)
__global__ __launch_bounds__(256) void stream_scatter_kernel(float *state, uint64_t stride, uint32_t in_off, const StreamEntry *entries,
                                                             const float *src) {
    const StreamEntry e = entries[blockIdx.x];
    float *dst = state + e.stream * stride + in_off;
    if (src)
        for (uint32_t i = threadIdx.x; i < e.len; i += 256) dst[i] = src[e.src_off + i];
    for (uint32_t i = threadIdx.x; i < e.zero_fill; i += 256) dst[e.len + i] = 0.0f;
}

// new carry = the last `keep` samples before in_off + len (+ zero_fill), moved so that they end at in_off:
// a shift to lower addresses by the chunk length.  Ascending 256-sample pieces, each read completely
// before it is written, never touch the source of a later piece.
__global__ __launch_bounds__(256) void stream_carry_kernel(float *state, uint64_t stride, uint32_t in_off, const StreamEntry *entries) {
    const StreamEntry e = entries[blockIdx.x];
    const uint32_t n = e.len + e.zero_fill;
    if (n == 0) return;
    float *slot = state + e.stream * stride;
    const float *src = slot + in_off + n - e.keep;
    float *dst = slot + in_off - e.keep;
    for (uint32_t base = 0; base < e.keep; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const float v = i < e.keep ? src[i] : 0.0f;
        __syncthreads();
        if (i < e.keep) dst[i] = v;
        __syncthreads();
    }
}

// Ragged batch whose descriptors live in device memory (melspec_*_ragged_device_desc): the plan the host builds for
// melspec_compute_ragged_device (plan_ragged in aux.hip), built by one workgroup instead -- per-clip frame counts,
// the prefix of units per clip, packed output offsets when none are given, the clip of every 16th unit -- so that a caller
// whose clip table is produced on the GPU (a VAD, a segmenter) never copies it back.  Layout of `plan` as plan_ragged's:
// [off n][frames n][out_off n][prefix n+1] u64, then the block table (u32).
struct PlanParams {
    const uint64_t *d_off, *d_len, *d_out_off;     // d_out_off may be null: outputs packed in clip order
    uint32_t n_clips;
    uint64_t frame_len, frame_shift;               // frames(n) = n < frame_len ? 0 : (n - frame_len) / frame_shift + 1
    uint32_t words_per_frame;                      // output words (floats) per frame
    uint32_t frames_per_unit;
    uint64_t *plan;
    uint64_t max_blocks;                           // capacity of the block table
};

__global__ __launch_bounds__(1024) void plan_ragged_device_kernel(const PlanParams q) {
    __shared__ uint64_t part_units[1024], part_out[1024];
    const uint32_t n = q.n_clips, tid = threadIdx.x;
    uint64_t *off = q.plan, *fr = off + n, *oo = fr + n, *pre = oo + n;
    uint32_t *blk = reinterpret_cast<uint32_t *>(pre + n + 1);
    const uint32_t per = (n + 1023) / 1024, c0 = tid * per, c1 = c0 + per < n ? c0 + per : n;
    uint64_t su = 0, so = 0;
    for (uint32_t c = c0; c < c1; ++c) {
        const uint64_t len = q.d_len[c];
        const uint64_t f = len < q.frame_len ? 0 : (len - q.frame_len) / q.frame_shift + 1;
        off[c] = q.d_off[c];
        fr[c] = f;
        su += (f + q.frames_per_unit - 1) / q.frames_per_unit;
        so += f * q.words_per_frame;
    }
    part_units[tid] = su; part_out[tid] = so;
    __syncthreads();
    if (tid == 0) {                                  // 1024 partials: a serial scan is 2 us
        uint64_t au = 0, ao = 0;
        for (int i = 0; i < 1024; ++i) {
            const uint64_t u = part_units[i], o = part_out[i];
            part_units[i] = au; part_out[i] = ao;
            au += u; ao += o;
        }
        pre[n] = au;
    }
    __syncthreads();
    su = part_units[tid]; so = part_out[tid];
    for (uint32_t c = c0; c < c1; ++c) {
        const uint64_t f = fr[c];
        const uint64_t u = (f + q.frames_per_unit - 1) / q.frames_per_unit;
        pre[c] = su;
        oo[c] = q.d_out_off ? q.d_out_off[c] : so;
        // the clip of every 16th unit inside [su, su + u)
        for (uint64_t k = (su + kUnitBlock - 1) / kUnitBlock; k * kUnitBlock < su + u && k < q.max_blocks; ++k) blk[k] = c;
        su += u;
        so += f * q.words_per_frame;
    }
}

// Hash-noise PCM (murmur3 finaliser) of SURVEY.md §8(d); the CPU tests regenerate the same bits.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(256) void synth_pcm_kernel(float *out, uint64_t clip_stride, uint64_t clip_len,
                                                        uint64_t first_clip, uint32_t n_clips, uint32_t seed, uint64_t first_sample) {
    const uint64_t total = (uint64_t)n_clips * clip_len;
    for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (uint64_t)gridDim.x * 256) {
        const uint64_t c = g / clip_len, i = g - c * clip_len;
        const uint64_t clip = first_clip + c;
        const uint32_t h = fmix32(seed ^ ((uint32_t)clip * 0x9E3779B1u) ^ ((uint32_t)(first_sample + i) * 0x85EBCA6Bu));
        const float u = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        out[c * clip_stride + i] = u * (1.0f / (float)(1u << (clip & 7u)));
    }
}


// ------------------------------------------------------------------------------------
// The mel stage on its own: MelSpectrogram::add(&fft) (src/mel.rs:13-32) = SparseMelFilterbank::project_stft_log10
// (src/mel.rs:148-168) + norm_mel_slice_f64 (src/mel.rs:645-654) for callers that hold complex STFT frames (their own, or
// melspec_stft_*'s): E[m] = sum over the row's contiguous bins of w * |X[bin]|^2 in ascending bin order (bins >= n_fft/2
// contribute nothing), log10(max(E, 1e-10)), max - 8 clamp, (x + 4) / 4 -- the reference's f64 arithmetic step by step.
// One frame per 64-thread wave of a workgroup; spectra as interleaved (re, im) of float or double, `stride` complex per frame.
// ------------------------------------------------------------------------------------
struct MelStageParams {
    const void *spec;
    float *out;
    uint64_t n_frames;
    uint32_t stride;        // complex elements per frame (n_fft/2 + 1 or n_fft)
    int bin_limit;          // n_fft / 2
    int n_mels;
    const int *d_mstart, *d_mlen, *d_moff;
    const double *d_mw;
    const double *d_jw;     // the same bank as jobs of eight weights (build_mel_jobs, aux.hip), for mel_stage_jobs_kernel
    const int *d_job;
    int n_jobs;
};

// mel_stage_jobs_kernel: a frame per wave, the bank as jobs in LDS -- the mel phase of pow2_frame_kernel (section 4.3b of DESIGN.md)
// on spectra that come from memory.  The first form (mel_stage_kernel below, kept for banks the tables of this one do not take) read
// every weight from global memory inside a loop whose trip count is the band's width, a lane per mel, and took an f64 log10: 26 %
// of its roofline.  Here: the frame's bins in up to kMelStageBinLoads coalesced loads per lane, the NEXT frame's issued before this
// frame is worked on; jobs of eight weights, two rounds in flight, one ds_add_f64 per job; v_log_f32 like every fused kernel.
constexpr int kMelStageWaves = 8;
constexpr int kMelStageBinLoads = 4;       // 64 lanes x 4: frames of up to 256 bins take the unrolled path
struct MelStageLds { int jw, job, frames, frame_stride, acc, total; };
MS_HD MelStageLds mel_stage_lds(int n_jobs, int bin_limit, int n_mels, int waves) {
    MelStageLds o;
    o.jw = 0;
    o.job = 8 * n_jobs;
    o.frames = (o.job + (n_jobs + 1) / 2 + 31) & ~31;
    o.acc = (bin_limit + 8 + 1) & ~1;                  // a frame: the power row (+ 8 a job may read past it), the band sums
    o.frame_stride = (o.acc + n_mels + 31) & ~31;
    o.total = o.frames + waves * o.frame_stride;
    return o;
}

template <class T>
__global__ __launch_bounds__(kMelStageWaves * 64) void mel_stage_jobs_kernel(const MelStageParams p) {
    extern __shared__ __attribute__((aligned(16))) double stage_lds[];
    struct alignas(2 * sizeof(T)) T2 { T re, im; };
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const MelStageLds at = mel_stage_lds(p.n_jobs, p.bin_limit, p.n_mels, kMelStageWaves);
    double *ljw = stage_lds + at.jw;
    int *ljob = reinterpret_cast<int *>(stage_lds + at.job);
    for (int i = tid; i < 8 * p.n_jobs; i += kMelStageWaves * 64) ljw[i] = p.d_jw[i];
    for (int i = tid; i < p.n_jobs; i += kMelStageWaves * 64) ljob[i] = p.d_job[i];
    double *pw = stage_lds + at.frames + wave * at.frame_stride, *acc = pw + at.acc;
    if (lane < 8) pw[p.bin_limit + lane] = 0.0;
    __syncthreads();
    const int n_jobs = p.n_jobs, bins = p.bin_limit;
    const bool small = bins <= 64 * kMelStageBinLoads;
    const uint64_t step = (uint64_t)gridDim.x * kMelStageWaves;
    uint64_t f = (uint64_t)blockIdx.x * kMelStageWaves + wave;
    if (f >= p.n_frames) return;
    auto fetch = [&](uint64_t frame, T2 (&v)[kMelStageBinLoads]) {
        const T2 *x = reinterpret_cast<const T2 *>(static_cast<const T *>(p.spec) + 2 * frame * p.stride);
#pragma unroll
        for (int i = 0; i < kMelStageBinLoads; ++i) { const int k = lane + 64 * i; v[i] = x[k < bins ? k : bins - 1]; }
    };
    T2 cur[kMelStageBinLoads];
    fetch(f, cur);
    for (;;) {
        const uint64_t nf = f + step;
        const bool more = nf < p.n_frames;                      // wave-uniform
        T2 nxt[kMelStageBinLoads];
        if (more) fetch(nf, nxt);
#pragma unroll
        for (int i = 0; i < kMelStageBinLoads; ++i) {
            const int k = lane + 64 * i;
            const double re = static_cast<double>(cur[i].re), im = static_cast<double>(cur[i].im);
            if (k < bins) pw[k] = re * re + im * im;                                   // norm_sqr, src/mel.rs:159
        }
        if (!small) {
            const T2 *x = reinterpret_cast<const T2 *>(static_cast<const T *>(p.spec) + 2 * f * p.stride);
            for (int k = lane + 64 * kMelStageBinLoads; k < bins; k += 64) {
                const double re = static_cast<double>(x[k].re), im = static_cast<double>(x[k].im);
                pw[k] = re * re + im * im;
            }
        }
        for (int m = lane; m < p.n_mels; m += 64) acc[m] = 0.0;
        for (int jb0 = lane; jb0 < n_jobs + lane; jb0 += 2 * 64) {            // wave-uniform trip count; ascending bins inside a job, src/mel.rs:155-163
            int info[2];
            d2 w[2][4];
            double pv[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t) info[t] = jb0 + 64 * t < n_jobs ? ljob[jb0 + 64 * t] : 0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int jb = jb0 + 64 * t;
                const double *wp = ljw + 2 * (jb < n_jobs ? jb : 0), *pp = pw + (info[t] & 0xfff);
#pragma unroll
                for (int q = 0; q < 4; ++q) w[t][q] = *reinterpret_cast<const d2 *>(wp + q * 2 * n_jobs);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const d2 two = *reinterpret_cast<const d2 *>(pp + 2 * q);
                    pv[t][2 * q] = two.x; pv[t][2 * q + 1] = two.y;
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                double e = w[t][0].x * pv[t][0];
                e += w[t][0].y * pv[t][1]; e += w[t][1].x * pv[t][2]; e += w[t][1].y * pv[t][3];
                e += w[t][2].x * pv[t][4]; e += w[t][2].y * pv[t][5]; e += w[t][3].x * pv[t][6]; e += w[t][3].y * pv[t][7];
                if ((info[t] >> 20) > 0) unsafeAtomicAdd(acc + ((info[t] >> 12) & 0xff), e);
            }
        }
        // log10 through v_log_f32 (as the fused kernels), the frame's maximum, clamp, (x + 4) / 4 (src/mel.rs:166, 645-654)
        constexpr int kPer = 4;                                  // 256 mels
        float mv[kPer];
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int m = lane + 64 * i;
            mv[i] = 0.0f;
            if (m < p.n_mels) {
                const double e = acc[m];
                mv[i] = fast_log2((float)(e > 1e-10 ? e : 1e-10)) * 0.30102999566398120f;
                mx = mx > mv[i] ? mx : mv[i];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(mx, o, 64); mx = mx > t ? mx : t; }
        const float lo = mx - 8.0f;
        float *o = p.out + f * (uint64_t)p.n_mels;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int m = lane + 64 * i;
            if (m < p.n_mels) o[m] = ((mv[i] > lo ? mv[i] : lo) + 4.0f) * 0.25f;
        }
        if (!more) break;
        f = nf;
#pragma unroll
        for (int i = 0; i < kMelStageBinLoads; ++i) cur[i] = nxt[i];
    }
}

template <class T, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mel_stage_kernel(const MelStageParams p) {
    extern __shared__ __attribute__((aligned(16))) double stage_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *pw = stage_lds + (size_t)wave * (p.bin_limit + p.n_mels);      // [bin_limit] powers, [n_mels] log values
    double *lv = pw + p.bin_limit;
    for (uint64_t f = (uint64_t)blockIdx.x * WAVES + wave; f < p.n_frames; f += (uint64_t)gridDim.x * WAVES) {
        const T *x = static_cast<const T *>(p.spec) + 2 * f * p.stride;
        for (int k = lane; k < p.bin_limit; k += 64) {
            const double re = static_cast<double>(x[2 * k]), im = static_cast<double>(x[2 * k + 1]);
            pw[k] = re * re + im * im;                                   // norm_sqr, src/mel.rs:159
        }
        __builtin_amdgcn_wave_barrier();
        double mx = -1.0e300;
        for (int m = lane; m < p.n_mels; m += 64) {
            const int st = p.d_mstart[m], len = p.d_mlen[m];
            const double *w = p.d_mw + p.d_moff[m];
            double e = 0.0;
            for (int i = 0; i < len; ++i) e += w[i] * pw[st + i];        // ascending bins, src/mel.rs:155-163
            const double v = log10(e > 1e-10 ? e : 1e-10);               // src/mel.rs:166
            lv[m] = v;
            mx = v > mx ? v : mx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(mx, o);
            mx = other > mx ? other : mx;
        }
        __builtin_amdgcn_wave_barrier();
        const double lo = mx - 8.0;                                      // src/mel.rs:645-654
        float *o = p.out + f * (uint64_t)p.n_mels;
        for (int m = lane; m < p.n_mels; m += 64) {
            const double v = lv[m];
            o[m] = static_cast<float>(((v > lo ? v : lo) + 4.0) / 4.0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}


}  // namespace melspec
