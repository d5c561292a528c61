// aux.hip -- everything around the path: the batch planners, the stand-alone mel stage and SparseMelFilterbank helpers, the streaming bank
// (src/stft.rs:48-86, src/rb.rs:86-121), the TGA quantiser (src/quant.rs), the VAD column stencil (src/vad.rs:373-415), synthetic PCM.
#include "host_common.hpp"
#include "aux_kernels.hpp"
#include "mel_bank.hpp"
#include "tga_quant.hpp"
#include "vad_columns.hpp"

namespace melspec {
namespace host {

// Tables of the generic (f64 DFT) kernel.
// The banded filterbank as JOBS for the wave kernels' mel phase (pow2_frame_kernel, mel_stage_jobs_kernel): eight consecutive weights
// of one mel per job (the last job of a band padded with zeros), dealt over the rounds as described below; lf = lanes that share a
// frame (a round = lf jobs).  jobv[j] = first bin | mel << 12 | count << 20; jwv: weight pairs (2 q, 2 q + 1) of job j at [q][j].
void build_mel_jobs(const BandedFilterbank &fb, int n_mels, int lf, std::vector<double> &jwv, std::vector<int> &jobv) {
    // A job = eight consecutive bins FROM AN EVEN ONE (its eight powers are four aligned 16-byte LDS reads) with the weights of one mel
    // on them, zero where the band is not; record = first bin | mel << 12 | count << 20 (count > 0: a real job).
    struct Job { int bin, mel, lo, hi; };                    // weights of band entries [lo, hi) sit at bins bin + (entry - lo) + lead
    std::vector<Job> jobs;
    std::vector<int> lead;                                   // zero weights in front of a job's first entry
    for (int m = 0; m < n_mels && m < 256; ++m) {
        const int st = fb.start[m], len = fb.len[m];
        if (len <= 0) continue;
        for (int b = st & ~1; b < st + len; b += 8) {
            const int lo = std::max(b, st) - st, hi = std::min(b + 8, st + len) - st;
            jobs.push_back({b, m, lo, hi});
            lead.push_back(std::max(b, st) - b);
        }
    }
#ifndef MS_POW2_JOBORDER
#define MS_POW2_JOBORDER 1
#endif
    // The lanes of a round read their jobs' powers with ds_read_b128, which the LDS serves in groups of sixteen lanes -- {0-3, 12-15,
    // 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- over the sixteen 16-byte slots of a 256-byte row: the jobs that meet in such a
    // group want first bins whose halves differ mod 16 (mod 8 where a frame has eight lanes: the group then holds the same eight jobs
    // of two pairs of frames, whose rows pow2_pw_shift sets eight slots apart).  In band order they do not -- a band's jobs are 8 bins
    // apart, the low bands 2-3 -- and the reads were 3-4-way (SQ_LDS_BANK_CONFLICT: 25 % of the LDS cycles at n_fft 2048).  So: sets of
    // g jobs; a job with fewer than eight entries may start 2, 4 or 6 bins early (more zeros in front); every job goes to the set that
    // holds the fewest jobs of its residue (then of its mel: ds_add_f64 to one address serialises), fullest residue classes first;
    // the sets padded with empty jobs (count 0) and laid onto the lane groups.  The order is a function of the bank: the sums stay
    // deterministic; a band's pieces are added in another order than the reference's left fold (f64: ~1e-16 relative).
    const int g = lf >= 16 ? 16 : 8;
    const size_t nj = jobs.size(), sets = std::max<size_t>(1, (nj + g - 1) / g);
    std::vector<std::vector<size_t>> grp(sets);
    std::vector<int> shift(nj, 0);
    if (MS_POW2_JOBORDER && lf >= 8) {
        std::vector<std::vector<size_t>> cls(g);
        for (size_t j = 0; j < nj; ++j) cls[(jobs[j].bin >> 1) % g].push_back(j);
        std::vector<int> order(g);
        for (int r = 0; r < g; ++r) order[r] = r;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cls[x].size() > cls[y].size(); });
        std::vector<std::vector<int>> used(sets, std::vector<int>(g, 0));
        for (int r : order)
            for (size_t j : cls[r]) {
                size_t best = sets;
                int best_shift = 0;
                long best_key = 0;
                const int room = 8 - (lead[j] + (jobs[j].hi - jobs[j].lo));
                for (int sh = 0; sh <= room && sh <= jobs[j].bin; sh += 2) {
                    const int res = ((jobs[j].bin - sh) >> 1) % g;
                    for (size_t q = 0; q < sets; ++q) {
                        if (grp[q].size() >= static_cast<size_t>(g)) continue;
                        long same_mel = 0;
                        for (size_t o : grp[q]) same_mel += jobs[o].mel == jobs[j].mel;
                        const long key = (static_cast<long>(used[q][res]) << 40) + (same_mel << 24) + (static_cast<long>(sh) << 16) + static_cast<long>(grp[q].size());
                        if (best == sets || key < best_key) { best = q; best_key = key; best_shift = sh; }
                    }
                }
                grp[best].push_back(j);
                shift[j] = best_shift;
                used[best][((jobs[j].bin - best_shift) >> 1) % g] += 1;
            }
    } else {
        for (size_t j = 0; j < nj; ++j) grp[j / g].push_back(j);
    }
    // lane positions: lf >= 32: two sets per 32 lanes, on the two lane groups of ds_read_b128; else a set = the lanes of a frame
    static const int kLanesOfGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const size_t slots = lf >= 32 ? ((sets + 1) / 2) * 32 : sets * g;
    std::vector<long> at(slots, -1);
    for (size_t q = 0; q < sets; ++q) {
        std::sort(grp[q].begin(), grp[q].end());
        for (size_t i = 0; i < grp[q].size(); ++i)
            at[lf >= 32 ? (q / 2) * 32 + kLanesOfGroup[q & 1][i] : q * g + i] = static_cast<long>(grp[q][i]);
    }
    jobv.assign(nj ? slots : 1, 0);
    std::vector<double> w8((nj ? slots : 1) * 8, 0.0);
    for (size_t pos = 0; pos < slots && nj; ++pos) {
        if (at[pos] < 0) continue;
        const size_t j = static_cast<size_t>(at[pos]);
        const Job &jb = jobs[j];
        const int bin = jb.bin - shift[j], first = lead[j] + shift[j];
        jobv[pos] = bin | (jb.mel << 12) | ((jb.hi - jb.lo) << 20);
        for (int e = jb.lo; e < jb.hi; ++e) w8[8 * pos + first + (e - jb.lo)] = fb.w[static_cast<size_t>(fb.offset[jb.mel]) + e];
    }
    // weight pairs (2 q, 2 q + 1) of the job at position j at [q][j]: the lanes of a round read consecutive 16-byte slots
    const size_t np = jobv.size();
    jwv.assign(8 * np, 0.0);
    for (size_t j = 0; j < np; ++j)
        for (int q = 0; q < 8; ++q) jwv[2 * ((q / 2) * np + j) + (q & 1)] = w8[8 * j + q];
}

__global__ void plan_upload_kernel(const uint4 *src, uint4 *dst, size_t n16) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) dst[i] = src[i];
}

// Fills the next slot and queues its upload on `stream`.  The caller launches on `stream` and then calls plan_ragged_done.
// want_order: also upload the clips sorted longest first and a zeroed ticket counter (BatchDesc::d_order / d_ticket) for the kernels that
// hand out whole clips.
int plan_ragged(RaggedScratch &rs, hipStream_t stream, const float *d_pcm, float *d_out, const uint64_t *h_off,
                const std::vector<uint64_t> &frames, const uint64_t *h_out_off, uint32_t n_clips, int n_mels,
                int frames_per_unit, BatchPlan &pl, RaggedSlot *&used, bool want_order) {
    RaggedSlot &sl = rs.slot[rs.next++ % RaggedScratch::kSlots];
    used = &sl;
    if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.pending) { HIP_TRY(hipEventSynchronize(sl.ev)); sl.pending = false; }
    uint64_t units = 0;
    for (uint32_t c = 0; c < n_clips; ++c) units += (frames[c] + frames_per_unit - 1) / frames_per_unit;
    const uint64_t n_blocks = (units + kUnitBlock - 1) / kUnitBlock;
    const size_t words64 = static_cast<size_t>(n_clips) * 4 + 1;
    const size_t blk_words = static_cast<size_t>(n_blocks ? n_blocks : 1);
    const size_t bytes = words64 * sizeof(uint64_t) + (blk_words + (want_order ? static_cast<size_t>(n_clips) + 1 : 0)) * sizeof(uint32_t);
    int rc = sl.ensure_host((bytes + 15) & ~static_cast<size_t>(15));
    if (rc) return rc;
    if ((rc = sl.dev.ensure((bytes + 15) & ~static_cast<size_t>(15)))) return rc;
    uint64_t *off = static_cast<uint64_t *>(sl.host), *fr = off + n_clips, *oo = fr + n_clips, *pre = oo + n_clips;
    uint32_t *blk = reinterpret_cast<uint32_t *>(off + words64);
    uint64_t cursor = 0, out_cursor = 0, total = 0;
    for (uint32_t c = 0; c < n_clips; ++c) {
        off[c] = h_off[c];
        fr[c] = frames[c];
        oo[c] = h_out_off ? h_out_off[c] : out_cursor;
        out_cursor += frames[c] * static_cast<uint64_t>(n_mels);
        pre[c] = cursor;
        cursor += (frames[c] + frames_per_unit - 1) / frames_per_unit;
        total += frames[c];
    }
    pre[n_clips] = units;
    {
        uint32_t c = 0;
        for (uint64_t k = 0; k < n_blocks; ++k) {
            const uint64_t u = k * kUnitBlock;
            while (pre[c + 1] <= u) ++c;      // u < units = pre[n_clips]
            blk[k] = c;
        }
    }
    if (want_order) {
        uint32_t *ord = blk + blk_words;
        for (uint32_t c = 0; c < n_clips; ++c) ord[c] = c;
        std::stable_sort(ord, ord + n_clips, [&](uint32_t a, uint32_t b2) { return frames[a] > frames[b2]; });
        ord[n_clips] = 0;       // the ticket counter
    }
    // the upload is a kernel on the launch stream that reads the pinned slot over the bus: an SDMA copy sits in another
    // hardware queue and the hand-over between the queues costs more than the copy
    {
        const size_t n16 = (bytes + 15) / 16;
        const unsigned blocks = static_cast<unsigned>((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
        hipLaunchKernelGGL(plan_upload_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const uint4 *>(sl.host),
                           static_cast<uint4 *>(sl.dev.p), n16);
        HIP_TRY(hipGetLastError());
    }
    const uint64_t *d = static_cast<const uint64_t *>(sl.dev.p);
    BatchDesc &b = pl.desc;
    b = BatchDesc{};
    b.pcm = d_pcm; b.out = d_out; b.n_clips = n_clips; b.n_units = units; b.frames_per_unit = frames_per_unit;
    b.d_off = d; b.d_frames = d + n_clips; b.d_out_off = d + 2 * n_clips; b.d_unit_prefix = d + 3 * n_clips;
    b.d_unit_block = reinterpret_cast<const uint32_t *>(d + words64);
    if (want_order) {
        b.d_order = b.d_unit_block + blk_words;
        b.d_ticket = const_cast<uint32_t *>(b.d_order) + n_clips;
    }
    pl.total_frames = total;
    b.stat_frames = total;
    return MELSPEC_OK;
}
// behind the launch (or the failed attempt) that used the slot
void plan_ragged_done(RaggedSlot *sl, hipStream_t stream) {
    if (sl && sl->ev && hipEventRecord(sl->ev, stream) == hipSuccess) sl->pending = true;
}

int plan_ragged_device(DevicePlan &dp, hipStream_t stream, const float *d_pcm, float *d_out, const uint64_t *d_off, const uint64_t *d_len,
                       const uint64_t *d_out_off, uint32_t n_clips, uint64_t frame_len, uint64_t frame_shift, uint32_t words_per_frame,
                       int frames_per_unit, uint64_t max_total_frames, BatchPlan &pl) {
    // every clip with frames has at most frames / fpu + 1 units
    const uint64_t max_units = max_total_frames / frames_per_unit + n_clips;
    const uint64_t max_blocks = max_units / kUnitBlock + 2;
    const size_t words64 = static_cast<size_t>(n_clips) * 4 + 1;
    const size_t bytes = words64 * sizeof(uint64_t) + static_cast<size_t>(max_blocks) * sizeof(uint32_t) + 16;
    if (dp.used && dp.last != stream) HIP_TRY(hipStreamSynchronize(dp.last));
    if (bytes > dp.buf.cap && dp.used) HIP_TRY(hipStreamSynchronize(dp.last));
    int rc = dp.buf.ensure(bytes);
    if (rc) return rc;
    dp.used = true; dp.last = stream;
    PlanParams q{};
    q.d_off = d_off; q.d_len = d_len; q.d_out_off = d_out_off; q.n_clips = n_clips;
    q.frame_len = frame_len; q.frame_shift = frame_shift; q.words_per_frame = words_per_frame;
    q.frames_per_unit = static_cast<uint32_t>(frames_per_unit);
    q.plan = static_cast<uint64_t *>(dp.buf.p);
    q.max_blocks = max_blocks;
    hipLaunchKernelGGL(plan_ragged_device_kernel, dim3(1), dim3(1024), 0, stream, q);
    HIP_TRY(hipGetLastError());
    const uint64_t *d = q.plan;
    BatchDesc &b = pl.desc;
    b = BatchDesc{};
    b.pcm = d_pcm; b.out = d_out; b.n_clips = n_clips; b.n_units = max_units; b.frames_per_unit = frames_per_unit;
    b.d_off = d; b.d_frames = d + n_clips; b.d_out_off = d + 2 * n_clips; b.d_unit_prefix = d + 3 * n_clips;
    b.d_unit_block = reinterpret_cast<const uint32_t *>(d + words64);
    b.d_n_units = d + 4 * static_cast<size_t>(n_clips);        // prefix[n_clips]
    pl.total_frames = max_total_frames;
    return MELSPEC_OK;
}

}  // namespace host
}  // namespace melspec

extern "C" {

// ---- the mel stage on its own: MelSpectrogram::add(&fft) (src/mel.rs:13-32) over complex STFT frames --------------------------
namespace {
int stage_tables(melspec_ctx *c) {
    if (c->stage_built) return MELSPEC_OK;
    const int bins = c->fft_size / 2 + 1;
    const BandedFilterbank fb = band_filterbank(c->dense, c->n_mels, bins, c->fft_size / 2);      // bins >= n_fft/2 contribute nothing (src/mel.rs:155-163)
    int rc;
    if ((rc = upload(c->st_start, fb.start))) return rc;
    if ((rc = upload(c->st_len, fb.len))) return rc;
    if ((rc = upload(c->st_off, fb.offset))) return rc;
    if ((rc = upload(c->st_w, fb.w))) return rc;
    {
        std::vector<double> jwv;
        std::vector<int> jobv;
        build_mel_jobs(fb, c->n_mels, 64, jwv, jobv);
        c->st_n_jobs = jobv.size() == 1 && (jobv[0] >> 20) == 0 ? 0 : static_cast<int>(jobv.size());
        if ((rc = upload(c->st_jw, jwv))) return rc;
        if ((rc = upload(c->st_job, jobv))) return rc;
    }
    c->stage_built = true;
    return MELSPEC_OK;
}
}  // namespace

int melspec_mel_from_stft_device(melspec_ctx *c, const void *d_spec, int dtype, int full, uint64_t n_frames, float *d_out, void *stream) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!d_spec || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    int rc = stage_tables(c);
    if (rc) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    MelStageParams p{};
    p.spec = d_spec; p.out = d_out; p.n_frames = n_frames;
    p.stride = static_cast<uint32_t>(full ? c->fft_size : c->fft_size / 2 + 1);
    p.bin_limit = c->fft_size / 2; p.n_mels = c->n_mels;
    p.d_mstart = static_cast<const int *>(c->st_start.p); p.d_mlen = static_cast<const int *>(c->st_len.p);
    p.d_moff = static_cast<const int *>(c->st_off.p); p.d_mw = static_cast<const double *>(c->st_w.p);
    p.d_jw = static_cast<const double *>(c->st_jw.p); p.d_job = static_cast<const int *>(c->st_job.p); p.n_jobs = c->st_n_jobs;
    {
        // the wave-per-frame form with the bank as jobs in LDS (banks of up to 256 mels over up to 4088 bins that fit)
        const size_t lds = sizeof(double) * static_cast<size_t>(mel_stage_lds(p.n_jobs, p.bin_limit, p.n_mels, kMelStageWaves).total);
        if (p.n_jobs > 0 && p.n_mels <= 256 && p.bin_limit <= 4088 && lds <= 64 * 1024) {
            const unsigned grid = grid_for((n_frames + kMelStageWaves - 1) / kMelStageWaves, c->dev.cus, static_cast<int>(std::max<size_t>(1, std::min<size_t>(4, kLdsLimit / lds))));
            if (dtype == MELSPEC_STFT_F64) hipLaunchKernelGGL((mel_stage_jobs_kernel<double>), dim3(grid), dim3(kMelStageWaves * 64), lds, s, p);
            else hipLaunchKernelGGL((mel_stage_jobs_kernel<float>), dim3(grid), dim3(kMelStageWaves * 64), lds, s, p);
            HIP_TRY(hipGetLastError());
            return MELSPEC_OK;
        }
    }
    constexpr int kWaves = 4;
    const size_t lds = static_cast<size_t>(kWaves) * (p.bin_limit + p.n_mels) * sizeof(double);
    if (lds > 64 * 1024) return fail(MELSPEC_ERR_UNSUPPORTED, "geometry needs more LDS than the mel stage kernel has");
    const unsigned grid = grid_for((n_frames + kWaves - 1) / kWaves, c->dev.cus, 16);
    if (dtype == MELSPEC_STFT_F64) hipLaunchKernelGGL((mel_stage_kernel<double, kWaves>), dim3(grid), dim3(kWaves * 64), lds, s, p);
    else hipLaunchKernelGGL((mel_stage_kernel<float, kWaves>), dim3(grid), dim3(kWaves * 64), lds, s, p);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_mel_from_stft_host(melspec_ctx *c, const void *spec, int dtype, int full, size_t n_frames, float *out, size_t out_capacity_floats) {
    if (!c) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!spec || !out) return fail(MELSPEC_ERR_INVALID_ARG, "spec/out is NULL");
    const size_t need = n_frames * static_cast<size_t>(c->n_mels);
    if (out_capacity_floats < need) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(c->dev.device));
    const size_t in_bytes = n_frames * static_cast<size_t>(full ? c->fft_size : c->fft_size / 2 + 1) * 2 * (dtype == MELSPEC_STFT_F64 ? sizeof(double) : sizeof(float));
    void *d_in = nullptr, *d_o = nullptr;
    HIP_TRY(hipMalloc(&d_in, in_bytes));
    hipError_t e = hipMalloc(&d_o, need * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(d_in); return fail_hip(e, "hipMalloc"); }
    int rc = MELSPEC_OK;
    if ((e = hipMemcpyAsync(d_in, spec, in_bytes, hipMemcpyHostToDevice, c->stream)) != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync");
    if (!rc) rc = melspec_mel_from_stft_device(c, d_in, dtype, full, n_frames, static_cast<float *>(d_o), c->stream);
    if (!rc && (e = hipMemcpyAsync(out, d_o, need * sizeof(float), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) rc = fail_hip(e, "hipMemcpyAsync");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess && !rc) rc = fail_hip(e, "hipStreamSynchronize");
    (void)hipFree(d_in); (void)hipFree(d_o);
    return rc;
}

}  // extern "C"

// ---- stand-alone mel helpers (mel_bank.hpp): SparseMelFilterbank, project_power, log_mel_spectrogram, norm_mel -------------------
struct melspec_bank {
    DeviceInfo dev;
    hipStream_t stream = nullptr;
    int n_mels = 0, fft_bins = 0, nnz = 0;
    DevBuf row_ptr, bin, w, wf, key, tmp_in, tmp_out;
    hipStream_t key_stream = nullptr;              // norm_mel's scratch word is used in stream order: a call on another stream first waits for this one
    bool key_used = false;
    std::vector<int> h_row_ptr, h_bin;             // the sparse rows on the host (weights_for_mel)
    std::vector<double> h_w;
    BankDesc desc() const {
        return BankDesc{static_cast<const int *>(row_ptr.p), static_cast<const int *>(bin.p), static_cast<const double *>(w.p),
                        static_cast<const float *>(wf.p), n_mels, fft_bins};
    }
};

namespace {
int bank_create(melspec_bank **out, int device, const std::vector<double> &dense, int n_mels, int fft_bins) {
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    melspec_bank *b = new (std::nothrow) melspec_bank();
    if (!b) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    b->dev = info; b->n_mels = n_mels; b->fft_bins = fft_bins;
    auto bail = [&](int code) { melspec_bank_destroy(b); return code; };
    if (hipSetDevice(info.device) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipSetDevice failed"));
    if (hipStreamCreate(&b->stream) != hipSuccess) return bail(fail(MELSPEC_ERR_UNAVAILABLE, "hipStreamCreate failed"));
    // from_dense (src/mel.rs:48-71): per row the non-zero entries in ascending bin order
    std::vector<int> row_ptr(static_cast<size_t>(n_mels) + 1, 0), bins;
    std::vector<double> w;
    std::vector<float> wf;
    for (int m = 0; m < n_mels; ++m) {
        for (int k = 0; k < fft_bins; ++k) {
            const double v = dense[static_cast<size_t>(m) * fft_bins + k];
            if (v != 0.0) { bins.push_back(k); w.push_back(v); wf.push_back(static_cast<float>(v)); }
        }
        row_ptr[m + 1] = static_cast<int>(bins.size());
    }
    b->nnz = static_cast<int>(bins.size());
    b->h_row_ptr = row_ptr; b->h_bin = bins; b->h_w = w;
    if ((rc = upload(b->row_ptr, row_ptr)) || (rc = upload(b->bin, bins)) || (rc = upload(b->w, w)) || (rc = upload(b->wf, wf))) return bail(rc);
    if ((rc = b->key.ensure(16))) return bail(rc);
    *out = b;
    return MELSPEC_OK;
}

template <class T>
int norm_launch(const T *d_in, uint64_t n, T *d_out, unsigned long long *key, int cus, hipStream_t s) {
    const unsigned grid = static_cast<unsigned>(std::min<uint64_t>((n + 255) / 256, static_cast<uint64_t>(cus) * 16));
    hipLaunchKernelGGL(norm_init_kernel, dim3(1), dim3(1), 0, s, key);
    hipLaunchKernelGGL(norm_max_kernel<T>, dim3(grid), dim3(256), 0, s, d_in, n, key);
    hipLaunchKernelGGL(norm_map_kernel<T>, dim3(grid), dim3(256), 0, s, d_in, n, key, d_out);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}
}  // namespace

extern "C" {

int melspec_bank_from_dense(melspec_bank **out, int device, const double *filters, int n_mels, int fft_bins) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!filters || n_mels <= 0 || fft_bins <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "filters is NULL or a dimension is not positive");
    return bank_create(out, device, std::vector<double>(filters, filters + static_cast<size_t>(n_mels) * fft_bins), n_mels, fft_bins);
}

int melspec_bank_from_mel(melspec_bank **out, int device, double sample_rate, int n_fft, int n_mels, double f_min, double f_max, int htk, int norm) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!(sample_rate > 0.0) || n_fft < 2 || n_mels <= 0) return fail(MELSPEC_ERR_INVALID_ARG, "sample_rate, n_fft and n_mels must be positive");
    return bank_create(out, device, mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max, htk != 0, norm != 0), n_mels, n_fft / 2 + 1);
}

void melspec_bank_destroy(melspec_bank *b) {
    if (!b) return;
    if (b->dev.device >= 0) (void)hipSetDevice(b->dev.device);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    b->row_ptr.release(); b->bin.release(); b->w.release(); b->wf.release(); b->key.release(); b->tmp_in.release(); b->tmp_out.release();
    delete b;
}

int melspec_bank_n_mels(const melspec_bank *b) { return b ? b->n_mels : 0; }
int melspec_bank_fft_bins(const melspec_bank *b) { return b ? b->fft_bins : 0; }
int melspec_bank_non_zero_weights(const melspec_bank *b) { return b ? b->nnz : 0; }

int melspec_bank_weights_for_mel(const melspec_bank *b, int mel_idx, int *bins, double *weights, int capacity) {
    if (!b || mel_idx < 0 || mel_idx >= b->n_mels) return -1;
    const int lo = b->h_row_ptr[mel_idx], n = b->h_row_ptr[mel_idx + 1] - lo;
    for (int i = 0; i < n && i < capacity; ++i) {
        if (bins) bins[i] = b->h_bin[lo + i];
        if (weights) weights[i] = b->h_w[lo + i];
    }
    return n;
}

int melspec_bank_project_power_device(melspec_bank *b, const void *d_power, int dtype, uint64_t n_frames, void *d_out, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!d_power || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    const uint64_t items = n_frames * static_cast<uint64_t>(b->n_mels);
    if ((items + 255) / 256 > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "too many frames for one call");
    const dim3 grid(static_cast<unsigned>((items + 255) / 256));
    if (dtype == MELSPEC_STFT_F64)
        hipLaunchKernelGGL(bank_project_power_kernel<double>, grid, dim3(256), 0, s, b->desc(), static_cast<const double *>(d_power), static_cast<double *>(d_out), n_frames);
    else
        hipLaunchKernelGGL(bank_project_power_kernel<float>, grid, dim3(256), 0, s, b->desc(), static_cast<const float *>(d_power), static_cast<float *>(d_out), n_frames);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_bank_log_mel_device(melspec_bank *b, const void *d_stft, int dtype, int n_fft, uint64_t n_frames, double *d_out, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_fft < 2 || n_fft < b->fft_bins) return fail(MELSPEC_ERR_INVALID_ARG, "n_fft must be at least the bank's fft_bins");
    if (n_frames == 0) return MELSPEC_OK;
    if (!d_stft || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    const uint64_t items = n_frames * static_cast<uint64_t>(b->n_mels);
    if ((items + 255) / 256 > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "too many frames for one call");
    const dim3 grid(static_cast<unsigned>((items + 255) / 256));
    if (dtype == MELSPEC_STFT_F64)
        hipLaunchKernelGGL(bank_log_mel_kernel<double>, grid, dim3(256), 0, s, b->desc(), static_cast<const double *>(d_stft), n_fft, d_out, n_frames);
    else
        hipLaunchKernelGGL(bank_log_mel_kernel<float>, grid, dim3(256), 0, s, b->desc(), static_cast<const float *>(d_stft), n_fft, d_out, n_frames);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_bank_norm_mel_device(melspec_bank *b, const void *d_in, int dtype, uint64_t n_values, void *d_out, void *stream) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_values == 0) return MELSPEC_OK;
    if (!d_in || !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(b->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : b->stream;
    unsigned long long *key = static_cast<unsigned long long *>(b->key.p);
    // the three launches of a call (init, max, map) share ONE scratch word per bank: two calls on different streams would race on it
    //  -- like FixState::last_stream, a change of stream waits for the previous one
    if (b->key_used && b->key_stream != s) HIP_TRY(hipStreamSynchronize(b->key_stream));
    b->key_used = true; b->key_stream = s;
    return dtype == MELSPEC_STFT_F64 ? norm_launch<double>(static_cast<const double *>(d_in), n_values, static_cast<double *>(d_out), key, b->dev.cus, s)
                                     : norm_launch<float>(static_cast<const float *>(d_in), n_values, static_cast<float *>(d_out), key, b->dev.cus, s);
}

// host forms: staged through the bank's own buffers, synchronous
static int bank_host_call(melspec_bank *b, const void *in, size_t in_bytes, void *out, size_t out_bytes, int (*run)(melspec_bank *, const void *, void *, void *), void *ctx) {
    HIP_TRY(hipSetDevice(b->dev.device));
    int rc;
    if ((rc = b->tmp_in.ensure(in_bytes + 16)) || (rc = b->tmp_out.ensure(out_bytes + 16))) return rc;
    HIP_TRY(hipMemcpyAsync(b->tmp_in.p, in, in_bytes, hipMemcpyHostToDevice, b->stream));
    if ((rc = run(b, b->tmp_in.p, b->tmp_out.p, ctx))) { (void)hipStreamSynchronize(b->stream); return rc; }
    HIP_TRY(hipMemcpyAsync(out, b->tmp_out.p, out_bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return MELSPEC_OK;
}

int melspec_bank_project_power_host(melspec_bank *b, const void *power, int dtype, size_t n_frames, void *out) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_frames == 0) return MELSPEC_OK;
    if (!power || !out) return fail(MELSPEC_ERR_INVALID_ARG, "pointer is NULL");
    const size_t el = dtype == MELSPEC_STFT_F64 ? 8 : 4;
    struct A { int dtype; uint64_t n; } a{dtype, n_frames};
    return bank_host_call(b, power, n_frames * b->fft_bins * el, out, n_frames * b->n_mels * el,
                          [](melspec_bank *bb, const void *i, void *o, void *c) { auto *x = static_cast<A *>(c); return melspec_bank_project_power_device(bb, i, x->dtype, x->n, o, bb->stream); }, &a);
}

int melspec_bank_log_mel_host(melspec_bank *b, const void *stft, int dtype, int n_fft, size_t n_frames, double *out) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_fft < 2) return fail(MELSPEC_ERR_INVALID_ARG, "n_fft must be >= 2");
    if (n_frames == 0) return MELSPEC_OK;
    if (!stft || !out) return fail(MELSPEC_ERR_INVALID_ARG, "pointer is NULL");
    const size_t el = dtype == MELSPEC_STFT_F64 ? 8 : 4;
    struct A { int dtype, n_fft; uint64_t n; } a{dtype, n_fft, n_frames};
    return bank_host_call(b, stft, n_frames * n_fft * 2 * el, out, n_frames * b->n_mels * 8,
                          [](melspec_bank *bb, const void *i, void *o, void *c) { auto *x = static_cast<A *>(c); return melspec_bank_log_mel_device(bb, i, x->dtype, x->n_fft, x->n, static_cast<double *>(o), bb->stream); }, &a);
}

int melspec_bank_norm_mel_host(melspec_bank *b, const void *in, int dtype, size_t n_values, void *out) {
    if (!b) return fail(MELSPEC_ERR_INVALID_ARG, "bank is NULL");
    if (dtype != MELSPEC_STFT_F32 && dtype != MELSPEC_STFT_F64) return fail(MELSPEC_ERR_INVALID_ARG, "dtype must be MELSPEC_STFT_F32 or MELSPEC_STFT_F64");
    if (n_values == 0) return MELSPEC_OK;
    if (!in || !out) return fail(MELSPEC_ERR_INVALID_ARG, "pointer is NULL");
    const size_t el = dtype == MELSPEC_STFT_F64 ? 8 : 4;
    struct A { int dtype; uint64_t n; } a{dtype, n_values};
    return bank_host_call(b, in, n_values * el, out, n_values * el,
                          [](melspec_bank *bb, const void *i, void *o, void *c) { auto *x = static_cast<A *>(c); return melspec_bank_norm_mel_device(bb, i, x->dtype, x->n, o, bb->stream); }, &a);
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// Streaming: a bank of live streams with device-side overlap-save state
// (Spectrogram::add src/stft.rs:48-86 driven by RingBuffer::maybe_mel src/rb.rs:86-121)
// ------------------------------------------------------------------------------------
struct melspec_stream {
    melspec_ctx *ctx = nullptr;          // geometry, tables, kernels; not owned
    StreamGeom geom{};
    StreamBook book;                     // pending / idx per stream (host side of the state)
    DevBuf state, staging, out;
    RaggedScratch ring;                  // per-push entry tables
    // the detector stage (melspec_stream_enable_vad): VoiceActivityDetector state per stream, in HBM
    bool vad_on = false;
    melspec_vad_settings vad{};
    DevBuf vad_state, vad_prev, vad_acts;
    std::vector<uint64_t> vad_count;     // host copy of StreamVadState::count (VoiceActivityDetector::frame_index)
    // Steady state of a live bank: the same streams pushing the same number of samples from the same pending count, every stream past
    // its first window.  Such a push has the entry table and the ragged plan of the previous one -- both are still on the device --
    // so neither is built or uploaded again (4096 streams x 1 hop: 0.089 -> 0.05 ms per push).
    struct PushCache {
        bool valid = false;
        uint32_t n = 0;
        int fpu = 0;
        const void *d_out = nullptr;
        std::vector<uint32_t> ids, lens, pend;
        std::vector<uint64_t> out_off;       // the caller's row offsets (empty: packed)
        StreamPlan pl;
        const StreamEntry *d_e = nullptr;    // the entries in the ring slot of the push that filled the cache
        BatchPlan plan;
    } cache;
    RaggedScratch plan_ring;             // the ragged plans of the pushes (the context's own ring serves its other callers)
};

namespace {
int stream_plan(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, bool flush, StreamPlan &pl) {
    const char *err = nullptr;
    const int rc = stream_plan_push(st->geom, st->book, ids, lens, n, flush, pl, &err);
    if (rc == 1) return fail(MELSPEC_ERR_INVALID_ARG, err);
    if (rc == 2) return fail(MELSPEC_ERR_CAPACITY, err);
    return MELSPEC_OK;
}
void stream_commit(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, bool flush) {
    stream_commit_push(st->geom, st->book, ids, lens, n, flush);
}

// what a push emits per frame: the mel row (Spectrogram::add + MelSpectrogram::add) or the spectrum (Spectrogram::add alone)
struct StreamEmit {
    bool stft = false;
    int dtype = MELSPEC_STFT_F32, full = 0;
};

// scatter (optional) -> frames -> carry update, all on one stream
// does this push repeat the cached one?  (ids / lens / pending before the push; every stream was past its first window when the cache
// was filled and idx only grows, resets invalidate)
bool stream_cache_hit(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, const void *d_out, const uint64_t *h_out_off) {
    const melspec_stream::PushCache &k = st->cache;
    if (!k.valid || k.n != n || k.d_out != d_out || k.out_off.empty() != (h_out_off == nullptr)) return false;
    if (k.fpu != ctx_frames_per_unit(st->ctx)) return false;            // AUTO changed its regime: another unit size
    for (uint32_t i = 0; i < n; ++i)
        if (ids[i] != k.ids[i] || lens[i] != k.lens[i] || st->book.pending[ids[i]] != k.pend[i]) return false;
    return h_out_off == nullptr || std::memcmp(h_out_off, k.out_off.data(), static_cast<size_t>(n) * sizeof(uint64_t)) == 0;
}

// reuse: `pl` is st->cache.pl and the device still holds its entries and plan (stream_cache_hit); ids / lens: the push's arguments,
// for filling the cache (NULL: do not, e.g. a flush)
int stream_run(melspec_stream *st, const StreamPlan &pl, uint32_t n, const float *d_src, void *d_out, const uint64_t *h_out_off,
               hipStream_t s, const StreamEmit &emit = StreamEmit(), melspec_vad_activity *d_acts = nullptr, bool reuse = false,
               const uint32_t *ids = nullptr, const uint32_t *lens = nullptr) {
    melspec_ctx *c = st->ctx;
    HIP_TRY(hipSetDevice(c->dev.device));
    if (pl.total_frames && !d_out) return fail(MELSPEC_ERR_INVALID_ARG, "d_out is NULL");      // before anything is queued
    if (st->vad_on && emit.stft) return fail(MELSPEC_ERR_UNSUPPORTED, "the detector stage is on: it needs the mel rows of every push");
    if (st->vad_on && pl.total_frames && !d_acts) {                                             // records nobody asked for: internal buffer
        const int rc0 = st->vad_acts.ensure(pl.total_frames * sizeof(melspec_vad_activity));
        if (rc0) return rc0;
        d_acts = static_cast<melspec_vad_activity *>(st->vad_acts.p);
    }
    int rc = MELSPEC_OK;
    // the entries travel like a ragged plan: pinned slot, copy kernel on the launch stream (no SDMA queue hand-over)
    struct SlotGuard { RaggedSlot *sl; hipStream_t s; ~SlotGuard() { plan_ragged_done(sl, s); } } slot_guard{nullptr, s};
    const StreamEntry *d_e = st->cache.d_e;
    if (!reuse) {
        st->cache.valid = false;                     // whatever happens below, the slot the cache points into may be the next one taken
        RaggedSlot &sl = st->ring.slot[st->ring.next++ % RaggedScratch::kSlots];
        if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        if (sl.pending) { HIP_TRY(hipEventSynchronize(sl.ev)); sl.pending = false; }
        const size_t ebytes = (static_cast<size_t>(n) * sizeof(StreamEntry) + 15) & ~static_cast<size_t>(15);
        rc = sl.ensure_host(ebytes);
        if (rc) return rc;
        if ((rc = sl.dev.ensure(ebytes))) return rc;
        std::memcpy(sl.host, pl.entries.data(), static_cast<size_t>(n) * sizeof(StreamEntry));
        if (h_out_off)                                   // caller-placed rows: the detector stage reads them where they are
            for (uint32_t i = 0; i < n; ++i) static_cast<StreamEntry *>(sl.host)[i].out_off = h_out_off[i];
        // from here on the slot is in use by queued work: every exit records its event (the next user of the slot waits for it)
        slot_guard.sl = &sl;
        const size_t n16 = ebytes / 16;
        const unsigned blocks = static_cast<unsigned>((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
        hipLaunchKernelGGL(plan_upload_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, static_cast<const uint4 *>(sl.host),
                           static_cast<uint4 *>(sl.dev.p), n16);
        HIP_TRY(hipGetLastError());
        d_e = static_cast<const StreamEntry *>(sl.dev.p);
    }
    float *state = static_cast<float *>(st->state.p);
    bool any_fill = d_src != nullptr;
    for (uint32_t i = 0; i < n && !any_fill; ++i) any_fill = pl.entries[i].zero_fill != 0;
    if (any_fill) {
        hipLaunchKernelGGL(stream_scatter_kernel, dim3(n), dim3(256), 0, s, state, st->geom.stride, st->geom.in_off, d_e, d_src);
        HIP_TRY(hipGetLastError());
    }
    if (pl.total_frames && emit.stft) {
        std::vector<uint64_t> oo(n);            // complex elements, entries back to back
        uint64_t cur = 0;
        const uint64_t bins = melspec_stft_bins(c, emit.full);
        for (uint32_t i = 0; i < n; ++i) { oo[i] = cur; cur += pl.frames[i] * bins; }
        rc = melspec_stft_ragged_device(c, state, pl.off.data(), pl.len.data(), n, d_out, h_out_off ? h_out_off : oo.data(), emit.dtype, emit.full, s);
        if (rc) return rc;
    } else if (pl.total_frames) {
        if (reuse) {
            rc = launch_ctx(c, st->cache.plan.desc, s);
        } else {
            // melspec_compute_ragged_device with the plan kept: frames per entry are the plan's, the ring is the bank's own
            std::vector<uint64_t> fr(n);
            for (uint32_t i = 0; i < n; ++i) fr[i] = pl.frames[i];
            RaggedSlot *pslot = nullptr;
            const int fpu = ctx_frames_per_unit(c);
            rc = plan_ragged(st->plan_ring, s, state, static_cast<float *>(d_out), pl.off.data(), fr, h_out_off ? h_out_off : pl.out_off.data(), n,
                             c->n_mels, fpu, st->cache.plan, pslot);
            if (!rc) rc = launch_ctx(c, st->cache.plan.desc, s);
            plan_ragged_done(pslot, s);
            // a push that can come again: every stream past its first window (no skipped hops), not a flush
            bool steady = !rc && ids != nullptr && lens != nullptr;
            for (uint32_t i = 0; i < n && steady; ++i) steady = st->book.idx[ids[i]] >= st->geom.n_fft;
            if (steady) {
                melspec_stream::PushCache &k = st->cache;
                k.n = n; k.fpu = fpu; k.d_out = d_out; k.d_e = d_e;
                k.ids.assign(ids, ids + n); k.lens.assign(lens, lens + n);
                k.pend.resize(n);
                for (uint32_t i = 0; i < n; ++i) k.pend[i] = st->book.pending[ids[i]];
                if (h_out_off) k.out_off.assign(h_out_off, h_out_off + n); else k.out_off.clear();
                k.pl = pl;
                k.valid = true;
            }
        }
        if (rc) return rc;
        if (st->vad_on) {
            StreamVadParams vp{};
            vp.entries = d_e; vp.rows = static_cast<const float *>(d_out);
            vp.state = static_cast<StreamVadState *>(st->vad_state.p); vp.prev = static_cast<float *>(st->vad_prev.p);
            vp.acts = reinterpret_cast<VadActivity *>(d_acts);
            vp.n_mels = st->geom.n_mels; vp.min_mel = st->vad.min_mel; vp.min_y = st->vad.min_y; vp.min_x = st->vad.min_x;
            vp.thr = st->vad.min_energy * st->vad.min_energy;
            uint32_t most = 0;
            for (uint32_t i = 0; i < n; ++i) most = std::max(most, pl.frames[i]);
            hipLaunchKernelGGL(stream_vad_kernel, dim3(n), dim3(most <= 1 ? 64 : most <= 2 ? 128 : 256), 0, s, vp);
            HIP_TRY(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(stream_carry_kernel, dim3(n), dim3(256), 0, s, state, st->geom.stride, st->geom.in_off, d_e);
    HIP_TRY(hipGetLastError());
    // the contract of the push calls: the launches have completed on return (a device producer may refill its slot at once)
    HIP_TRY(hipStreamSynchronize(s));
    return MELSPEC_OK;
}
}  // namespace

extern "C" {

int melspec_stream_create(melspec_stream **out, melspec_ctx *ctx, uint32_t n_streams, uint32_t max_chunk) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!ctx) return fail(MELSPEC_ERR_INVALID_ARG, "ctx is NULL");
    if (n_streams == 0 || max_chunk == 0) return fail(MELSPEC_ERR_INVALID_ARG, "n_streams and max_chunk must be > 0");
    if (ctx->hop_size > ctx->fft_size) return fail(MELSPEC_ERR_UNSUPPORTED, "streaming needs hop_size <= fft_size");
    melspec_stream *st = new (std::nothrow) melspec_stream();
    if (!st) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    st->ctx = ctx;
    st->geom = stream_geometry(static_cast<uint32_t>(ctx->fft_size), static_cast<uint32_t>(ctx->hop_size), static_cast<uint32_t>(ctx->n_mels),
                               n_streams, max_chunk);
    st->book.reset(n_streams);
    if (hipSetDevice(ctx->dev.device) != hipSuccess) { delete st; return fail(MELSPEC_ERR_UNAVAILABLE, "hipSetDevice failed"); }
    const size_t bytes = static_cast<size_t>(n_streams) * st->geom.stride * sizeof(float) + 64;
    int rc = st->state.ensure(bytes);
    if (rc) { delete st; return rc; }
    if (hipMemset(st->state.p, 0, bytes) != hipSuccess) { st->state.release(); delete st; return fail(MELSPEC_ERR_INTERNAL, "hipMemset failed"); }
    *out = st;
    return MELSPEC_OK;
}

void melspec_stream_destroy(melspec_stream *st) {
    if (!st) return;
    if (st->ctx) { (void)hipSetDevice(st->ctx->dev.device); (void)hipStreamSynchronize(st->ctx->stream); }
    st->state.release(); st->ring.release(); st->staging.release(); st->out.release();
    st->vad_state.release(); st->vad_prev.release(); st->vad_acts.release(); st->plan_ring.release();
    delete st;
}

int melspec_stream_reset(melspec_stream *st, const uint32_t *ids, uint32_t n) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    st->cache.valid = false;
    HIP_TRY(hipSetDevice(st->ctx->dev.device));
    if (!ids) {
        HIP_TRY(hipMemsetAsync(st->state.p, 0, static_cast<size_t>(st->geom.n_streams) * st->geom.stride * sizeof(float), st->ctx->stream));
        st->book.reset(st->geom.n_streams);
        if (st->vad_on) {
            HIP_TRY(hipMemsetAsync(st->vad_state.p, 0, static_cast<size_t>(st->geom.n_streams) * sizeof(StreamVadState), st->ctx->stream));
            std::fill(st->vad_count.begin(), st->vad_count.end(), 0ull);
        }
    } else {
        for (uint32_t i = 0; i < n; ++i) {
            if (ids[i] >= st->geom.n_streams) return fail(MELSPEC_ERR_INVALID_ARG, "stream id out of range");
            HIP_TRY(hipMemsetAsync(static_cast<float *>(st->state.p) + ids[i] * st->geom.stride, 0, st->geom.in_off * sizeof(float), st->ctx->stream));
            st->book.pending[ids[i]] = 0; st->book.idx[ids[i]] = 0;
            if (st->vad_on) {
                HIP_TRY(hipMemsetAsync(static_cast<StreamVadState *>(st->vad_state.p) + ids[i], 0, sizeof(StreamVadState), st->ctx->stream));
                st->vad_count[ids[i]] = 0;
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(st->ctx->stream));
    return MELSPEC_OK;
}

size_t melspec_stream_frames_after(const melspec_stream *st, uint32_t id, uint32_t n_new) {
    if (!st || id >= st->geom.n_streams) return 0;
    return stream_frames_after(st->geom, st->book, id, n_new);
}

float *melspec_stream_input_ptr(melspec_stream *st, uint32_t id) {
    if (!st || id >= st->geom.n_streams) return nullptr;
    return static_cast<float *>(st->state.p) + id * st->geom.stride + st->geom.in_off;
}

static_assert(sizeof(melspec_vad_activity) == 8 && sizeof(VadActivity) == 8, "the activity record is 8 bytes on both sides of the ABI");
static void stream_vad_commit(melspec_stream *st, const uint32_t *ids, const StreamPlan &pl, uint32_t n) {
    if (!st->vad_on) return;
    for (uint32_t i = 0; i < n; ++i) st->vad_count[ids[i]] += pl.frames[i];
}

static int stream_push_device_impl(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                                   const uint64_t *h_out_offsets, uint32_t *h_frames, melspec_vad_activity *d_acts, bool want_acts,
                                   void *stream) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    if (want_acts && !st->vad_on) return fail(MELSPEC_ERR_INVALID_ARG, "the detector stage is off (melspec_stream_enable_vad)");
    if (n == 0) return MELSPEC_OK;
    if (!ids || !lens) return fail(MELSPEC_ERR_INVALID_ARG, "ids/lens is NULL");
    StreamPlan fresh;
    const bool reuse = stream_cache_hit(st, ids, lens, n, d_out, h_out_offsets);
    int rc = reuse ? MELSPEC_OK : stream_plan(st, ids, lens, n, false, fresh);
    if (rc) return rc;
    const StreamPlan &pl = reuse ? st->cache.pl : fresh;
    if (want_acts && pl.total_frames && !d_acts) return fail(MELSPEC_ERR_INVALID_ARG, "d_acts is NULL");
    rc = stream_run(st, pl, n, nullptr, d_out, h_out_offsets, stream ? static_cast<hipStream_t>(stream) : st->ctx->stream, StreamEmit(), d_acts,
                    reuse, ids, lens);
    if (rc) return rc;
    stream_commit(st, ids, lens, n, false);
    stream_vad_commit(st, ids, pl, n);
    if (h_frames) std::memcpy(h_frames, pl.frames.data(), static_cast<size_t>(n) * sizeof(uint32_t));
    return MELSPEC_OK;
}

int melspec_stream_push_device(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                               const uint64_t *h_out_offsets, uint32_t *h_frames, void *stream) {
    return stream_push_device_impl(st, ids, lens, n, d_out, h_out_offsets, h_frames, nullptr, false, stream);
}

int melspec_stream_push_device_vad(melspec_stream *st, const uint32_t *ids, const uint32_t *lens, uint32_t n, float *d_out,
                                   const uint64_t *h_out_offsets, uint32_t *h_frames, melspec_vad_activity *d_acts, void *stream) {
    return stream_push_device_impl(st, ids, lens, n, d_out, h_out_offsets, h_frames, d_acts, true, stream);
}

int melspec_stream_enable_vad(melspec_stream *st, const melspec_vad_settings *settings) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    HIP_TRY(hipSetDevice(st->ctx->dev.device));
    HIP_TRY(hipStreamSynchronize(st->ctx->stream));
    if (!settings) { st->vad_on = false; return MELSPEC_OK; }
    if (settings->min_x > kStreamVadMaxX) return fail(MELSPEC_ERR_UNSUPPORTED, "min_x above 66: the column history of a stream is 64 bits");
    if (settings->min_x < 0 || settings->min_y < 0 || settings->min_mel < 0) return fail(MELSPEC_ERR_INVALID_ARG, "negative detection setting");
    const size_t ns = st->geom.n_streams;
    int rc = st->vad_state.ensure(ns * sizeof(StreamVadState));
    if (rc) return rc;
    if ((rc = st->vad_prev.ensure(ns * 2 * st->geom.n_mels * sizeof(float) + 16))) return rc;
    HIP_TRY(hipMemset(st->vad_state.p, 0, ns * sizeof(StreamVadState)));
    HIP_TRY(hipMemset(st->vad_prev.p, 0, ns * 2 * st->geom.n_mels * sizeof(float)));
    st->vad_count.assign(ns, 0ull);
    st->vad = *settings;
    st->vad_on = true;
    return MELSPEC_OK;
}

uint64_t melspec_stream_vad_frames(const melspec_stream *st, uint32_t id) {
    if (!st || !st->vad_on || id >= st->geom.n_streams) return 0;
    return st->vad_count[id];
}

static int stream_push_host_impl(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                 bool flush, void *out, size_t out_capacity, uint32_t *h_frames, const StreamEmit &emit = StreamEmit(),
                                 melspec_vad_activity *acts = nullptr, size_t acts_capacity = 0, bool want_acts = false) {
    if (!st) return fail(MELSPEC_ERR_INVALID_ARG, "stream bank is NULL");
    if (want_acts && !st->vad_on) return fail(MELSPEC_ERR_INVALID_ARG, "the detector stage is off (melspec_stream_enable_vad)");
    if (n == 0) return MELSPEC_OK;
    if (!ids || (!flush && !lens)) return fail(MELSPEC_ERR_INVALID_ARG, "ids/lens is NULL");
    StreamPlan fresh;
    const bool reuse = !flush && !emit.stft && stream_cache_hit(st, ids, lens, n, st->out.p, nullptr);
    int rc = reuse ? MELSPEC_OK : stream_plan(st, ids, lens, n, flush, fresh);
    if (rc) return rc;
    const StreamPlan &pl = reuse ? st->cache.pl : fresh;
    // elements the caller receives: floats (mel rows) or complex values (spectra)
    const uint64_t need = pl.total_frames * (emit.stft ? melspec_stft_bins(st->ctx, emit.full) : static_cast<uint64_t>(st->ctx->n_mels));
    const size_t esz = emit.stft ? (emit.dtype == MELSPEC_STFT_F64 ? 16 : 8) : sizeof(float);
    if (need > out_capacity) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    if (need && !out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    if (want_acts && pl.total_frames > acts_capacity) return fail(MELSPEC_ERR_CAPACITY, "activity buffer too small");
    if (want_acts && pl.total_frames && !acts) return fail(MELSPEC_ERR_INVALID_ARG, "acts is NULL");
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) total += pl.entries[i].len;
    if (total && !samples) return fail(MELSPEC_ERR_INVALID_ARG, "samples is NULL");
    hipStream_t s = st->ctx->stream;
    HIP_TRY(hipSetDevice(st->ctx->dev.device));
    if ((rc = st->staging.ensure(total * sizeof(float) + 16))) return rc;
    if ((rc = st->out.ensure(need * esz + 16))) return rc;
    if (total) HIP_TRY(hipMemcpyAsync(st->staging.p, samples, total * sizeof(float), hipMemcpyHostToDevice, s));
    if (st->vad_on && pl.total_frames && (rc = st->vad_acts.ensure(pl.total_frames * sizeof(melspec_vad_activity)))) return rc;
    // (st->out may have been re-allocated by the ensure above: the cache is keyed on its address, a stale one simply misses next time)
    rc = stream_run(st, pl, n, total ? static_cast<const float *>(st->staging.p) : nullptr, st->out.p, nullptr, s, emit,
                    static_cast<melspec_vad_activity *>(st->vad_acts.p), reuse && st->cache.d_out == st->out.p, flush || emit.stft ? nullptr : ids,
                    flush || emit.stft ? nullptr : lens);
    if (rc) return rc;
    if (need) {
        HIP_TRY(hipMemcpyAsync(out, st->out.p, need * esz, hipMemcpyDeviceToHost, s));
        if (want_acts) HIP_TRY(hipMemcpyAsync(acts, st->vad_acts.p, pl.total_frames * sizeof(melspec_vad_activity), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    stream_commit(st, ids, lens, n, flush);
    stream_vad_commit(st, ids, pl, n);
    if (h_frames) std::memcpy(h_frames, pl.frames.data(), static_cast<size_t>(n) * sizeof(uint32_t));
    return MELSPEC_OK;
}

int melspec_stream_push_host(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                             float *out, size_t out_capacity_floats, uint32_t *h_frames) {
    return stream_push_host_impl(st, ids, samples, lens, n, false, out, out_capacity_floats, h_frames);
}

int melspec_stream_push_host_stft(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                  void *out, size_t out_capacity_complex, uint32_t *h_frames, int dtype, int full) {
    if (st) { const int rc = stft_args(st->ctx, dtype); if (rc) return rc; }
    StreamEmit e; e.stft = true; e.dtype = dtype; e.full = full;
    return stream_push_host_impl(st, ids, samples, lens, n, false, out, out_capacity_complex, h_frames, e);
}

int melspec_stream_flush_host(melspec_stream *st, const uint32_t *ids, uint32_t n, float *out, size_t out_capacity_floats,
                              uint32_t *h_frames) {
    return stream_push_host_impl(st, ids, nullptr, nullptr, n, true, out, out_capacity_floats, h_frames);
}

int melspec_stream_push_host_vad(melspec_stream *st, const uint32_t *ids, const float *samples, const uint32_t *lens, uint32_t n,
                                 float *out, size_t out_capacity_floats, uint32_t *h_frames, melspec_vad_activity *acts, size_t acts_capacity) {
    return stream_push_host_impl(st, ids, samples, lens, n, false, out, out_capacity_floats, h_frames, StreamEmit(), acts, acts_capacity, true);
}

int melspec_stream_flush_host_vad(melspec_stream *st, const uint32_t *ids, uint32_t n, float *out, size_t out_capacity_floats,
                                  uint32_t *h_frames, melspec_vad_activity *acts, size_t acts_capacity) {
    return stream_push_host_impl(st, ids, nullptr, nullptr, n, true, out, out_capacity_floats, h_frames, StreamEmit(), acts, acts_capacity, true);
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// 8-bit quantisation + TGA container (src/quant.rs)
// ------------------------------------------------------------------------------------
struct melspec_tga {
    DeviceInfo dev;
    hipStream_t stream = nullptr;
    DevBuf keys, ranges, h2d, d2h, unit_ext;       // unit_ext: the mel kernel's per-unit extremes (melspec_tga_encode_pcm_uniform_device)
    // the min/max keys are one scratch buffer per handle, used in stream order: a call on another stream first waits for
    // the stream that used it last
    hipStream_t keys_stream = nullptr;
    bool keys_used = false;
};

namespace {
size_t round_up4(size_t v) { return (v + 3) & ~static_cast<size_t>(3); }

struct TgaLayout { uint32_t chunks; uint32_t chunk_w; size_t chunk_stride; size_t last_bytes; };
TgaLayout tga_layout(uint32_t rows, uint64_t width) {
    TgaLayout l{};
    if (width == 0) return l;
    l.chunks = static_cast<uint32_t>((width + kTgaMaxWidth - 1) / kTgaMaxWidth);
    l.chunk_w = static_cast<uint32_t>(width < kTgaMaxWidth ? width : kTgaMaxWidth);
    l.chunk_stride = round_up4(kTgaHeader + static_cast<size_t>(rows) * l.chunk_w);
    l.last_bytes = kTgaHeader + static_cast<size_t>(rows) * (width - static_cast<uint64_t>(l.chunks - 1) * l.chunk_w);
    return l;
}

// fills the descriptor and the launch shape shared by encode and decode
int quant_plan(melspec_tga *q, QuantDesc &d, const void *img, size_t image_stride, uint32_t rows, uint64_t width, uint32_t n_images,
               const void *blob, size_t blob_stride, bool header, uint32_t &items, uint32_t &bpi_px, uint32_t &bpi_dw) {
    if (width > 0xffffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "image wider than 2^32-1 columns");
    d = QuantDesc{};
    d.rows = rows; d.width = static_cast<uint32_t>(width); d.n_images = n_images;
    d.img_stride = image_stride; d.blob_stride = blob_stride;
    d.header = header ? kTgaHeader : 0;
    if (header) {
        const TgaLayout l = tga_layout(rows, width);
        d.chunks = l.chunks; d.chunk_w = l.chunk_w; d.chunk_stride = l.chunk_stride;
        if (blob_stride % 4 || blob_stride < l.chunk_stride * l.chunks)
            return fail(MELSPEC_ERR_INVALID_ARG, "blob_stride must be a multiple of 4 and >= n_chunks * chunk_stride (melspec_tga_layout)");
    } else {
        d.chunks = 1; d.chunk_w = d.width; d.chunk_stride = 0;
        if (blob_stride % 4) return fail(MELSPEC_ERR_INVALID_ARG, "blob_stride must be a multiple of 4");
    }
    if (reinterpret_cast<uintptr_t>(blob) % 4) return fail(MELSPEC_ERR_INVALID_ARG, "blob pointer must be 4-byte aligned");
    if (reinterpret_cast<uintptr_t>(img) % 4) return fail(MELSPEC_ERR_INVALID_ARG, "image pointer must be 4-byte aligned");
    d.vec = d.chunks == 1 && reinterpret_cast<uintptr_t>(img) % 16 == 0 && (n_images == 1 || image_stride % 4 == 0);
    const uint64_t items64 = static_cast<uint64_t>(n_images) * d.chunks;
    const uint64_t npx = static_cast<uint64_t>(rows) * d.chunk_w;
    const uint64_t bpx = (npx + kQuantPxPerBlock - 1) / kQuantPxPerBlock;
    const uint64_t bdw = ((d.header + npx + 3) / 4 + kQuantDwPerBlock - 1) / kQuantDwPerBlock;
    if (items64 * bpx > 0x7fffffffull || items64 * bdw > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "batch too large for one launch");
    items = static_cast<uint32_t>(items64); bpi_px = static_cast<uint32_t>(bpx); bpi_dw = static_cast<uint32_t>(bdw);
    int rc = q->keys.ensure(items64 * 2 * sizeof(uint32_t) + 16);
    if (rc) return rc;
    d.keys = static_cast<uint32_t *>(q->keys.p);
    return MELSPEC_OK;
}

int quant_encode(melspec_tga *q, const float *d_img, size_t image_stride, uint32_t rows, uint64_t width, uint32_t n_images,
                 uint8_t *d_blob, size_t blob_stride, bool header, float *d_ranges, hipStream_t stream) {
    if (n_images == 0 || width == 0 || rows == 0) return MELSPEC_OK;
    if (!d_img || !d_blob) return fail(MELSPEC_ERR_INVALID_ARG, "image/blob pointer is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    QuantDesc d;
    uint32_t items, bpx, bdw;
    int rc = quant_plan(q, d, d_img, image_stride, rows, width, n_images, d_blob, blob_stride, header, items, bpx, bdw);
    if (rc) return rc;
    d.img = d_img; d.blob = d_blob; d.ranges = d_ranges;
    if (q->keys_used && q->keys_stream != stream) HIP_TRY(hipStreamSynchronize(q->keys_stream));
    q->keys_used = true; q->keys_stream = stream;
    hipLaunchKernelGGL(quant_init_keys_kernel, dim3((items + 255) / 256), dim3(256), 0, stream, d.keys, items);
    hipLaunchKernelGGL(quant_minmax_kernel, dim3(items * bpx), dim3(kQuantThreads), 0, stream, d, bpx);
    hipLaunchKernelGGL(quant_encode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, stream, d, bdw);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int quant_decode(melspec_tga *q, const uint8_t *d_blob, size_t blob_stride, uint32_t rows, uint64_t width, uint32_t n_images,
                 float *d_img, size_t image_stride, bool header, const float *d_ranges, hipStream_t stream) {
    if (n_images == 0 || width == 0 || rows == 0) return MELSPEC_OK;
    if (!d_img || !d_blob) return fail(MELSPEC_ERR_INVALID_ARG, "image/blob pointer is NULL");
    if (!header && !d_ranges) return fail(MELSPEC_ERR_INVALID_ARG, "range pointer is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    QuantDesc d;
    uint32_t items, bpx, bdw;
    int rc = quant_plan(q, d, d_img, image_stride, rows, width, n_images, d_blob, blob_stride, header, items, bpx, bdw);
    if (rc) return rc;
    d.img_out = d_img; d.blob = const_cast<uint8_t *>(d_blob); d.ranges = const_cast<float *>(d_ranges);
    hipLaunchKernelGGL(quant_decode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, stream, d, bdw);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}
}  // namespace

extern "C" {

int melspec_tga_create(melspec_tga **out, int device) {
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    melspec_tga *q = new (std::nothrow) melspec_tga();
    if (!q) return fail(MELSPEC_ERR_INTERNAL, "out of host memory");
    q->dev = info;
    if (hipSetDevice(info.device) != hipSuccess || hipStreamCreate(&q->stream) != hipSuccess) {
        delete q;
        return fail(MELSPEC_ERR_UNAVAILABLE, "hipStreamCreate failed");
    }
    *out = q;
    return MELSPEC_OK;
}

void melspec_tga_destroy(melspec_tga *q) {
    if (!q) return;
    if (q->dev.device >= 0) (void)hipSetDevice(q->dev.device);
    if (q->stream) { (void)hipStreamSynchronize(q->stream); (void)hipStreamDestroy(q->stream); }
    q->keys.release(); q->ranges.release(); q->h2d.release(); q->d2h.release(); q->unit_ext.release();
    delete q;
}

int melspec_tga_layout(int n_mels, size_t width, uint32_t *n_chunks, size_t *chunk_stride, size_t *last_chunk_bytes) {
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    const TgaLayout l = tga_layout(static_cast<uint32_t>(n_mels), width);
    if (n_chunks) *n_chunks = l.chunks;
    if (chunk_stride) *chunk_stride = l.chunk_stride;
    if (last_chunk_bytes) *last_chunk_bytes = l.last_bytes;
    return MELSPEC_OK;
}

int melspec_tga_encode_device(melspec_tga *q, const float *d_images, size_t image_stride, int n_mels, size_t width,
                              uint32_t n_images, uint8_t *d_blobs, size_t blob_stride, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    if (image_stride < static_cast<size_t>(n_mels) * width) return fail(MELSPEC_ERR_INVALID_ARG, "image_stride < n_mels * width");
    return quant_encode(q, d_images, image_stride, static_cast<uint32_t>(n_mels), width, n_images, d_blobs, blob_stride, true, nullptr,
                        stream ? static_cast<hipStream_t>(stream) : q->stream);
}

// PCM -> TGA bytes with the image read once: while it stores the image (mel-major) the mel kernel leaves the extremes of every work
// unit behind (BatchDesc::d_unit_ext: one wave-wide reduction and one 8-byte store per unit), a one-wave-per-image kernel folds them
// into the quantiser's keys, so only the encoding pass reads the image again -- 5 B/pixel moved for 5 B/pixel algorithmic, where minmax + encode moved 9 ("4x smaller D2H").  The bytes are
// those of melspec_compute_uniform_device_interleaved(.., major_column_order = 0, min_width) followed by melspec_tga_encode_device.
int melspec_tga_encode_pcm_uniform_device(melspec_tga *q, melspec_ctx *c, const float *d_pcm, uint64_t clip_stride, uint64_t clip_len,
                                          uint32_t n_clips, uint64_t min_width, float *d_images, uint8_t *d_blobs, size_t blob_stride,
                                          void *stream) {
    if (!q || !c) return fail(MELSPEC_ERR_INVALID_ARG, "tga / ctx is NULL");
    if (q->dev.device != c->dev.device) return fail(MELSPEC_ERR_INVALID_ARG, "the codec and the context are on different devices");
    if (min_width % 2 != 0) return fail(MELSPEC_ERR_INVALID_ARG, "min_width must be even");   // src/mel.rs:488
    if (n_clips == 0) return MELSPEC_OK;
    uint64_t fpc; ctx_num_frames(c, clip_len, fpc);
    if (fpc == 0) return fail(MELSPEC_ERR_INVALID_ARG, "frames is empty");                      // src/mel.rs:487
    if (!d_pcm || !d_images || !d_blobs) return fail(MELSPEC_ERR_INVALID_ARG, "device pointer is NULL");
    HIP_TRY(hipSetDevice(c->dev.device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const uint64_t width = interleaved_width(fpc, min_width);
    const size_t image_stride = static_cast<size_t>(c->n_mels) * width;
    if (!c->fast || width > kTgaMaxWidth) {
        // geometries on the generic / 512-point kernels, and images wider than one TGA chunk: the two-pass form
        int rc = melspec_compute_uniform_device_interleaved(c, d_pcm, clip_stride, clip_len, n_clips, d_images, 0, min_width, s);
        if (rc) return rc;
        return melspec_tga_encode_device(q, d_images, image_stride, c->n_mels, width, n_clips, d_blobs, blob_stride, s);
    }
    QuantDesc d;
    uint32_t items, bpx, bdw;
    int rc = quant_plan(q, d, d_images, image_stride, static_cast<uint32_t>(c->n_mels), width, n_clips, d_blobs, blob_stride, true, items, bpx, bdw);
    if (rc) return rc;
    d.img = d_images; d.blob = d_blobs; d.ranges = nullptr;
    if (q->keys_used && q->keys_stream != s) HIP_TRY(hipStreamSynchronize(q->keys_stream));
    q->keys_used = true; q->keys_stream = s;
    BatchPlan pl = plan_uniform(d_pcm, d_images, clip_stride, fpc, n_clips, c->n_mels, ctx_frames_per_unit(c, true), width, true);
    if ((rc = q->unit_ext.ensure(static_cast<size_t>(pl.desc.n_units) * 2 * sizeof(int) + 16))) return rc;
    pl.desc.d_unit_ext = static_cast<int *>(q->unit_ext.p);
    if ((rc = launch_ctx(c, pl.desc, s))) return rc;
    // one chunk per image: item == clip; its units' records -> its keys
    hipLaunchKernelGGL(quant_keys_from_units_kernel, dim3(n_clips), dim3(64), 0, s, pl.desc.d_unit_ext, pl.desc.units_per_clip, n_clips, d.keys);
    hipLaunchKernelGGL(quant_encode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, s, d, bdw);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_tga_decode_device(melspec_tga *q, const uint8_t *d_blobs, size_t blob_stride, int n_mels, size_t width,
                              uint32_t n_images, float *d_images, size_t image_stride, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    if (image_stride < static_cast<size_t>(n_mels) * width) return fail(MELSPEC_ERR_INVALID_ARG, "image_stride < n_mels * width");
    return quant_decode(q, d_blobs, blob_stride, static_cast<uint32_t>(n_mels), width, n_images, d_images, image_stride, true, nullptr,
                        stream ? static_cast<hipStream_t>(stream) : q->stream);
}

int melspec_quantize_device(melspec_tga *q, const float *d_frame, size_t n, uint8_t *d_out, float *d_range, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (!d_range) return fail(MELSPEC_ERR_INVALID_ARG, "range pointer is NULL");
    return quant_encode(q, d_frame, n, 1, n, 1, d_out, round_up4(n), false, d_range, stream ? static_cast<hipStream_t>(stream) : q->stream);
}

int melspec_dequantize_device(melspec_tga *q, const uint8_t *d_data, size_t n, const float *d_range, float *d_out, void *stream) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    return quant_decode(q, d_data, round_up4(n), 1, n, 1, d_out, n, false, d_range, stream ? static_cast<hipStream_t>(stream) : q->stream);
}

int melspec_tga_synchronize(melspec_tga *q) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    HIP_TRY(hipStreamSynchronize(q->stream));
    return MELSPEC_OK;
}

int melspec_quantize_host(melspec_tga *q, const float *frame, size_t n, uint8_t *out, float *range) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (!range) return fail(MELSPEC_ERR_INVALID_ARG, "range pointer is NULL");
    if (n == 0) { range[0] = INFINITY; range[1] = -INFINITY; return MELSPEC_OK; }       // the folds' start values
    if (!frame || !out) return fail(MELSPEC_ERR_INVALID_ARG, "frame/out is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(n * sizeof(float)))) return rc;
    if ((rc = q->d2h.ensure(round_up4(n) + 16))) return rc;
    if ((rc = q->ranges.ensure(16))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, frame, n * sizeof(float), hipMemcpyHostToDevice, q->stream));
    if ((rc = melspec_quantize_device(q, static_cast<const float *>(q->h2d.p), n, static_cast<uint8_t *>(q->d2h.p),
                                      static_cast<float *>(q->ranges.p), q->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, n, hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipMemcpyAsync(range, q->ranges.p, 2 * sizeof(float), hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    return MELSPEC_OK;
}

int melspec_dequantize_host(melspec_tga *q, const uint8_t *data, size_t n, const float *range, float *out) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n == 0) return MELSPEC_OK;
    if (!data || !range || !out) return fail(MELSPEC_ERR_INVALID_ARG, "data/range/out is NULL");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(round_up4(n) + 16))) return rc;
    if ((rc = q->d2h.ensure(n * sizeof(float)))) return rc;
    if ((rc = q->ranges.ensure(16))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, data, n, hipMemcpyHostToDevice, q->stream));
    HIP_TRY(hipMemcpyAsync(q->ranges.p, range, 2 * sizeof(float), hipMemcpyHostToDevice, q->stream));
    if ((rc = melspec_dequantize_device(q, static_cast<const uint8_t *>(q->h2d.p), n, static_cast<const float *>(q->ranges.p),
                                        static_cast<float *>(q->d2h.p), q->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, n * sizeof(float), hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    return MELSPEC_OK;
}

int melspec_tga_encode_host(melspec_tga *q, const float *data, size_t len, int n_mels, uint8_t *out, size_t out_capacity,
                            uint32_t *n_chunks) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_chunks) *n_chunks = 0;
    if (n_mels <= 0 || n_mels > 65535) return fail(MELSPEC_ERR_INVALID_ARG, "n_mels must be in 1..65535");
    if (len % static_cast<size_t>(n_mels)) return fail(MELSPEC_ERR_INVALID_ARG, "data length is not a multiple of n_mels");
    const size_t width = len / n_mels;
    if (width == 0) return MELSPEC_OK;
    if (!data || !out) return fail(MELSPEC_ERR_INVALID_ARG, "data/out is NULL");
    const TgaLayout l = tga_layout(static_cast<uint32_t>(n_mels), width);
    const size_t region = l.chunk_stride * l.chunks;
    if (out_capacity < region - l.chunk_stride + l.last_bytes) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(len * sizeof(float)))) return rc;
    if ((rc = q->d2h.ensure(region))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, data, len * sizeof(float), hipMemcpyHostToDevice, q->stream));
    if ((rc = melspec_tga_encode_device(q, static_cast<const float *>(q->h2d.p), len, n_mels, width, 1,
                                        static_cast<uint8_t *>(q->d2h.p), region, q->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, region - l.chunk_stride + l.last_bytes, hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    if (n_chunks) *n_chunks = l.chunks;
    return MELSPEC_OK;
}

int melspec_tga_decode_host(melspec_tga *q, const uint8_t *blob, size_t n_bytes, float *out, size_t out_capacity, size_t *n_values) {
    if (!q) return fail(MELSPEC_ERR_INVALID_ARG, "tga is NULL");
    if (n_values) *n_values = 0;
    if (!blob || n_bytes < kTgaHeader) return fail(MELSPEC_ERR_INVALID_ARG, "failed to fill whole buffer");   // read_exact, src/quant.rs:74-75
    const size_t npx = n_bytes - kTgaHeader;
    if (npx == 0) return MELSPEC_OK;
    if (!out) return fail(MELSPEC_ERR_INVALID_ARG, "out is NULL");
    if (out_capacity < npx) return fail(MELSPEC_ERR_CAPACITY, "output buffer too small");
    if (npx > 0xffffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "more than 2^32-1 pixels");
    HIP_TRY(hipSetDevice(q->dev.device));
    int rc;
    if ((rc = q->h2d.ensure(round_up4(n_bytes) + 16))) return rc;
    if ((rc = q->d2h.ensure(npx * sizeof(float)))) return rc;
    HIP_TRY(hipMemcpyAsync(q->h2d.p, blob, n_bytes, hipMemcpyHostToDevice, q->stream));
    // the header's width/height are ignored by the reference too: everything after byte 26 is one row of pixels
    QuantDesc d;
    uint32_t items, bpx, bdw;
    d = QuantDesc{};
    d.rows = 1; d.width = static_cast<uint32_t>(npx); d.n_images = 1; d.chunks = 1; d.chunk_w = d.width;
    d.header = kTgaHeader; d.vec = 1; d.img_out = static_cast<float *>(q->d2h.p); d.blob = static_cast<uint8_t *>(q->h2d.p);
    items = 1; bpx = 0; (void)bpx;
    bdw = static_cast<uint32_t>(((kTgaHeader + npx + 3) / 4 + kQuantDwPerBlock - 1) / kQuantDwPerBlock);
    hipLaunchKernelGGL(quant_decode_kernel, dim3(items * bdw), dim3(kQuantThreads), 0, q->stream, d, bdw);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, q->d2h.p, npx * sizeof(float), hipMemcpyDeviceToHost, q->stream));
    HIP_TRY(hipStreamSynchronize(q->stream));
    if (n_values) *n_values = npx;
    return MELSPEC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// VAD column classification (vad_boundaries, src/vad.rs:256-340)
// ------------------------------------------------------------------------------------
extern "C" {

void melspec_vad_default_settings(melspec_vad_settings *s) {          // DetectionSettings::default, src/vad.rs:13-22
    if (!s) return;
    s->min_energy = 0.98; s->min_y = 11; s->min_x = 5; s->min_mel = 2;
}

size_t melspec_vad_mask_len(int n_mels, size_t width) { return (n_mels < 3 || width < 3) ? 0 : width - 2; }

int melspec_vad_boundaries_device(const float *d_images, size_t image_stride, int n_mels, size_t width, uint32_t n_images,
                                  const melspec_vad_settings *settings, uint8_t *d_raw, uint8_t *d_smoothed, size_t mask_stride,
                                  uint32_t *d_longest_run, void *stream) {
    if (!settings) return fail(MELSPEC_ERR_INVALID_ARG, "settings is NULL");
    if (n_mels < 0 || settings->min_y < 0 || settings->min_mel < 0) return fail(MELSPEC_ERR_INVALID_ARG, "negative size");
    const size_t n = melspec_vad_mask_len(n_mels, width);
    if (n_images == 0) return MELSPEC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {                                        // EdgeInfo::new(empty, empty), src/vad.rs:270-272
        if (d_longest_run) HIP_TRY(hipMemsetAsync(d_longest_run, 0, sizeof(uint32_t) * n_images, s));
        return MELSPEC_OK;
    }
    if (!d_images || !d_smoothed) return fail(MELSPEC_ERR_INVALID_ARG, "image/mask pointer is NULL");
    if (!d_raw) return fail(MELSPEC_ERR_INVALID_ARG, "d_raw is NULL (the vote reads the raw mask)");
    if (mask_stride < n) return fail(MELSPEC_ERR_INVALID_ARG, "mask_stride < width - 2");
    if (image_stride < static_cast<size_t>(n_mels) * width) return fail(MELSPEC_ERR_INVALID_ARG, "image_stride < n_mels * width");
    if (width > 0xffffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "image wider than 2^32-1 columns");
    VadDesc d{};
    d.img = d_images; d.raw = d_raw; d.smoothed = d_smoothed; d.longest = d_longest_run;
    d.img_stride = image_stride; d.mask_stride = mask_stride;
    d.height = static_cast<uint32_t>(n_mels); d.width = static_cast<uint32_t>(width); d.n_images = n_images;
    d.min_mel = settings->min_mel; d.min_y = settings->min_y; d.thr = settings->min_energy * settings->min_energy;
    const uint64_t bpi = (n + 255) / 256;
    if (bpi * n_images > 0x7fffffffull) return fail(MELSPEC_ERR_UNSUPPORTED, "batch too large for one launch");
    const unsigned grid = static_cast<unsigned>(bpi * n_images);
    hipLaunchKernelGGL(vad_raw_kernel, dim3(grid), dim3(256), 0, s, d, static_cast<uint32_t>(bpi), d_raw);
    hipLaunchKernelGGL(vad_smooth_kernel, dim3(grid), dim3(256), 0, s, d, static_cast<uint32_t>(bpi), d_raw);
    if (d_longest_run) hipLaunchKernelGGL(vad_run_kernel, dim3(n_images), dim3(64), 0, s, d);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_vad_boundaries_host(int device, const float *image, int n_mels, size_t width, const melspec_vad_settings *settings,
                                uint8_t *raw_out, uint8_t *smoothed_out, uint32_t *longest_run) {
    if (!settings) return fail(MELSPEC_ERR_INVALID_ARG, "settings is NULL");
    if (longest_run) *longest_run = 0;
    const size_t n = melspec_vad_mask_len(n_mels, width);
    if (n == 0) return MELSPEC_OK;
    if (!image || !smoothed_out) return fail(MELSPEC_ERR_INVALID_ARG, "image/out is NULL");
    DeviceInfo info;
    int rc = pick_device(device, info);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(info.device));
    DevBuf img, masks, run;
    const size_t px = static_cast<size_t>(n_mels) * width, ms = (n + 15) & ~static_cast<size_t>(15);
    auto done = [&](int code) { img.release(); masks.release(); run.release(); return code; };
    if ((rc = img.ensure(px * sizeof(float))) || (rc = masks.ensure(2 * ms)) || (rc = run.ensure(16))) return done(rc);
    if (hipMemcpy(img.p, image, px * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    uint8_t *m = static_cast<uint8_t *>(masks.p);
    rc = melspec_vad_boundaries_device(static_cast<const float *>(img.p), px, n_mels, width, 1, settings, m, m + ms, ms,
                                       static_cast<uint32_t *>(run.p), nullptr);
    if (rc) return done(rc);
    if (hipDeviceSynchronize() != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "vad kernels failed"));
    if (raw_out && hipMemcpy(raw_out, m, n, hipMemcpyDeviceToHost) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    if (hipMemcpy(smoothed_out, m + ms, n, hipMemcpyDeviceToHost) != hipSuccess) return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    if (longest_run && hipMemcpy(longest_run, run.p, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(MELSPEC_ERR_INTERNAL, "hipMemcpy failed"));
    return done(MELSPEC_OK);
}

}  // extern "C"

extern "C" {


static int synth_launch(float *d_out, uint64_t clip_stride, uint64_t first_sample, uint64_t n_samples, uint64_t first_clip,
                        uint32_t n_clips, uint32_t seed, void *stream) {
    if (n_clips == 0 || n_samples == 0) return MELSPEC_OK;
    if (!d_out) return fail(MELSPEC_ERR_INVALID_ARG, "d_out is NULL");
    const uint64_t total = static_cast<uint64_t>(n_clips) * n_samples;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_pcm_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_out, clip_stride, n_samples, first_clip, n_clips, seed, first_sample);
    HIP_TRY(hipGetLastError());
    return MELSPEC_OK;
}

int melspec_synth_pcm_device(float *d_out, uint64_t clip_stride, uint64_t clip_len, uint64_t first_clip,
                             uint32_t n_clips, uint32_t seed, void *stream) {
    return synth_launch(d_out, clip_stride, 0, clip_len, first_clip, n_clips, seed, stream);
}

int melspec_synth_pcm_window_device(float *d_out, uint64_t clip_stride, uint64_t first_sample, uint64_t n_samples,
                                    uint64_t first_clip, uint32_t n_clips, uint32_t seed, void *stream) {
    return synth_launch(d_out, clip_stride, first_sample, n_samples, first_clip, n_clips, seed, stream);
}

}  // extern "C"
