/*
 * melspec_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, f64 CPU restatement of the reference's log-mel hot path
 * (wavey-ai/mel-spec v0.4.0).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (mel_spec_amd / libmelspec_hip.so) never links, imports or calls it.
 *
 * Parity pin: the reference is Rust and cannot be built in this image (no
 * cargo/rustc), so this restatement is pinned against the reference's own
 * fixtures (tests/test_oracle.py):
 *   - mel()               vs testdata/mel_filters.npz      @1e-7  (src/mel.rs:838-850)
 *   - mel(n_fft=512)      vs testdata/nemo_mel_filters.npz @1e-7  (src/mel.rs:853-871)
 *   - streaming 512/160/80 on jfk_f32le.wav vs testdata/rust_jfk_golden.npy @1e-6
 *                                                          (src/rb.rs:134-179)
 *   - Whisper 400/160/80 on jfk_f32le.wav[80:] -> tga_8bit_data vs testdata/quantized_mel_golden.tga:
 *     all 88026 bytes equal (tests/test_quant.py; the file is loaded by src/vad.rs:684-744)
 *   - librosa scalar known-answers                         (src/mel.rs:787-835)
 *   - fbank frame count 1098 on JFK; values informational  (src/fbank.rs:484-490)
 * The FFT itself lives in a third-party crate (rustfft ^6.2.0, Cargo.toml:18, no
 * Cargo.lock in tree); it computes the unnormalised forward DFT
 * X[k] = sum_n x[n] exp(-2*pi*i*n*k/N), restated here as a mixed-radix
 * Cooley-Tukey in f64.  Fbank VALUES are "parity unpinned" in the reference
 * (shape-only test); they are pinned to this restatement only.
 *
 * Every function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { double re, im; } cplx;

/* ------------------------------------------------------------------------- */
/* Forward complex DFT of arbitrary length (restates what rustfft's            */
/* plan_fft_forward(n).process_with_scratch computes; call sites               */
/* src/stft.rs:99-110, src/fbank.rs:122-123,193-194).                          */
/* ------------------------------------------------------------------------- */
typedef struct {
    int n;
    cplx *tw;      /* tw[j] = exp(-2*pi*i*j/n), j in [0,n) */
    cplx *scratch; /* n entries */
} fft_plan;

static void fft_plan_init(fft_plan *p, int n) {
    p->n = n;
    p->tw = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    p->scratch = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    for (int j = 0; j < n; ++j) {
        double a = -2.0 * M_PI * (double)j / (double)n;
        p->tw[j].re = cos(a);
        p->tw[j].im = sin(a);
    }
}
static void fft_plan_free(fft_plan *p) { free(p->tw); free(p->scratch); p->tw = p->scratch = NULL; }

static inline cplx cmul(cplx a, cplx b) { cplx r = { a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re }; return r; }
static inline cplx cadd(cplx a, cplx b) { cplx r = { a.re + b.re, a.im + b.im }; return r; }
static inline cplx csub(cplx a, cplx b) { cplx r = { a.re - b.re, a.im - b.im }; return r; }

static int pick_radix(int n) {
    if (n % 4 == 0) return 4;
    if (n % 2 == 0) return 2;
    if (n % 5 == 0) return 5;
    if (n % 3 == 0) return 3;
    for (int p = 7; p * p <= n; p += 2) if (n % p == 0) return p;
    return n;
}

/* Decimation-in-time recursion: out[0..n) = DFT_n(in[0], in[stride], ...). */
static void fft_rec(const fft_plan *pl, int n, int stride, const cplx *in, cplx *out) {
    if (n == 1) { out[0] = in[0]; return; }
    const int N = pl->n;
    const int p = pick_radix(n);
    const int m = n / p;
    for (int q = 0; q < p; ++q) fft_rec(pl, m, stride * p, in + (size_t)q * stride, out + (size_t)q * m);
    const int tstep = N / n;      /* W_n^j = tw[j * tstep] */
    const int pstep = N / p;      /* W_p^j = tw[(j % p) * pstep] */
    if (p == 2) {
        for (int k = 0; k < m; ++k) {
            cplx a = out[k], b = cmul(out[m + k], pl->tw[k * tstep]);
            out[k] = cadd(a, b); out[m + k] = csub(a, b);
        }
    } else if (p == 4) {
        for (int k = 0; k < m; ++k) {
            cplx a = out[k];
            cplx b = cmul(out[m + k],     pl->tw[k * tstep]);
            cplx c = cmul(out[2 * m + k], pl->tw[2 * k * tstep]);
            cplx d = cmul(out[3 * m + k], pl->tw[3 * k * tstep]);
            cplx s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
            cplx js3 = { s3.im, -s3.re };            /* -i * s3 */
            out[k]         = cadd(s0, s2);
            out[m + k]     = cadd(s1, js3);
            out[2 * m + k] = csub(s0, s2);
            out[3 * m + k] = csub(s1, js3);
        }
    } else {
        cplx tmp[64];
        cplx *t = (p <= 64) ? tmp : (cplx *)malloc(sizeof(cplx) * (size_t)p);
        for (int k = 0; k < m; ++k) {
            for (int q = 0; q < p; ++q) t[q] = cmul(out[q * m + k], pl->tw[(q * k) * tstep]);
            for (int r = 0; r < p; ++r) {
                cplx acc = t[0];
                for (int q = 1; q < p; ++q) acc = cadd(acc, cmul(t[q], pl->tw[((q * r) % p) * pstep]));
                out[r * m + k] = acc;
            }
        }
        if (t != tmp) free(t);
    }
}

static void fft_forward_inplace(const fft_plan *pl, cplx *buf) {
    memcpy(pl->scratch, buf, sizeof(cplx) * (size_t)pl->n);
    fft_rec(pl, pl->n, 1, pl->scratch, buf);
}

/* Exposed for tests: forward DFT of n complex points (interleaved re,im). */
void oracle_fft_forward(int n, double *interleaved) {
    fft_plan pl; fft_plan_init(&pl, n);
    fft_forward_inplace(&pl, (cplx *)interleaved);
    fft_plan_free(&pl);
}

/* ------------------------------------------------------------------------- */
/* src/stft.rs:141-145 hann_window (periodic, f64)                             */
/* ------------------------------------------------------------------------- */
void oracle_hann_window(int n, double *w) {
    for (int i = 0; i < n; ++i) w[i] = 0.5 * (1.0 - cos((2.0 * M_PI * (double)i) / (double)n));
}

/* src/stft.rs:153-157 frame count of frame_windows (no padding, no centring) */
int64_t oracle_num_frames(int64_t len, int n_fft, int hop) {
    if (len < n_fft) return 0;
    return (len - n_fft) / hop + 1;
}

/* ------------------------------------------------------------------------- */
/* src/mel.rs:591-643 hz_to_mel / mel_to_hz / mel_frequencies / fft_frequencies */
/* ------------------------------------------------------------------------- */
double oracle_hz_to_mel(double f, int htk) {
    if (htk) return 2595.0 * log10(1.0 + f / 700.0);
    const double f_min = 0.0, f_sp = 200.0 / 3.0, min_log_hz = 1000.0;
    const double min_log_mel = (min_log_hz - f_min) / f_sp;
    const double logstep = log(6.4) / 27.0;
    if (f >= min_log_hz) return min_log_mel + (log(f / min_log_hz) / logstep);
    return (f - f_min) / f_sp;
}
double oracle_mel_to_hz(double mel, int htk) {
    if (htk) return 700.0 * (pow(10.0, mel / 2595.0) - 1.0);
    const double f_min = 0.0, f_sp = 200.0 / 3.0, min_log_hz = 1000.0;
    const double min_log_mel = (min_log_hz - f_min) / f_sp;
    const double logstep = log(6.4) / 27.0;
    if (mel >= min_log_mel) return min_log_hz * exp(logstep * (mel - min_log_mel));
    return f_min + f_sp * mel;
}
/* src/mel.rs:631-637; Array1::linspace(a,b,n)[i] = a + i*(b-a)/(n-1) */
void oracle_mel_frequencies(int n, double fmin, double fmax, int htk, double *out) {
    const double lo = oracle_hz_to_mel(fmin, htk), hi = oracle_hz_to_mel(fmax, htk);
    const double step = (n > 1) ? (hi - lo) / (double)(n - 1) : 0.0;
    for (int i = 0; i < n; ++i) out[i] = oracle_mel_to_hz(lo + step * (double)i, htk);
}
void oracle_fft_frequencies(double sr, int n_fft, double *out) {
    const double step = sr / (double)n_fft;
    for (int i = 0; i <= n_fft / 2; ++i) out[i] = step * (double)i;
}

/* ------------------------------------------------------------------------- */
/* src/mel.rs:547-589 mel(): dense Slaney/HTK filterbank [n_mels, n_fft/2+1]   */
/* f_min < 0 means None (0.0); f_max <= 0 means None (sr/2).                    */
/* ------------------------------------------------------------------------- */
void oracle_mel_filterbank(double sr, int n_fft, int n_mels, double f_min, double f_max,
                           int htk, int norm, double *weights /* n_mels*(n_fft/2+1) */) {
    const int bins = n_fft / 2 + 1;
    if (f_min < 0.0) f_min = 0.0;
    if (f_max <= 0.0) f_max = sr / 2.0;
    double *fftfreqs = (double *)malloc(sizeof(double) * (size_t)bins);
    double *mel_f = (double *)malloc(sizeof(double) * (size_t)(n_mels + 2));
    oracle_fft_frequencies(sr, n_fft, fftfreqs);
    oracle_mel_frequencies(n_mels + 2, f_min, f_max, htk, mel_f);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i];
        const double fd1 = mel_f[i + 2] - mel_f[i + 1];
        for (int b = 0; b < bins; ++b) {
            /* ramps[i][b] = mel_f[i] - fftfreqs[b] */
            double lower = -(mel_f[i] - fftfreqs[b]) / fd0;
            double upper = (mel_f[i + 2] - fftfreqs[b]) / fd1;
            lower = fmin(fmax(lower, 0.0), 1.0);
            upper = fmin(fmax(upper, 0.0), 1.0);
            weights[(size_t)i * bins + b] = fmin(lower, upper);
        }
        if (norm) {
            const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
            for (int b = 0; b < bins; ++b) weights[(size_t)i * bins + b] *= enorm;
        }
    }
    free(fftfreqs); free(mel_f);
}

/* ------------------------------------------------------------------------- */
/* src/mel.rs:34-71 SparseMelFilterbank::from_dense (rows of (bin, weight))     */
/* ------------------------------------------------------------------------- */
typedef struct {
    int n_mels, bins, nnz;
    int *row_ptr;   /* n_mels+1 */
    int *bin;       /* nnz */
    double *w;      /* nnz */
} sparse_fb;

static void sparse_from_dense(sparse_fb *s, const double *dense, int n_mels, int bins) {
    s->n_mels = n_mels; s->bins = bins;
    int nnz = 0;
    for (int i = 0; i < n_mels * bins; ++i) nnz += (dense[i] != 0.0);
    s->nnz = nnz;
    s->row_ptr = (int *)malloc(sizeof(int) * (size_t)(n_mels + 1));
    s->bin = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    s->w = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    int k = 0;
    for (int m = 0; m < n_mels; ++m) {
        s->row_ptr[m] = k;
        for (int b = 0; b < bins; ++b) {
            double v = dense[(size_t)m * bins + b];
            if (v != 0.0) { s->bin[k] = b; s->w[k] = v; ++k; }
        }
    }
    s->row_ptr[n_mels] = k;
}
static void sparse_free(sparse_fb *s) { free(s->row_ptr); free(s->bin); free(s->w); }

/* Returns nnz and (optionally) per-row start/len for the layout tests. */
int oracle_sparse_stats(const double *dense, int n_mels, int bins, int *row_start, int *row_len) {
    sparse_fb s; sparse_from_dense(&s, dense, n_mels, bins);
    for (int m = 0; m < n_mels; ++m) {
        int a = s.row_ptr[m], b = s.row_ptr[m + 1];
        if (row_start) row_start[m] = (b > a) ? s.bin[a] : 0;
        if (row_len) row_len[m] = b - a;
    }
    int nnz = s.nnz; sparse_free(&s); return nnz;
}

/* src/mel.rs:148-168 project_stft_log10 + src/mel.rs:645-654 norm_mel_slice_f64 */
static void mel_add_frame(const sparse_fb *fb, const cplx *stft, int n_fft, double *mel_buf, float *out_row) {
    const int half = n_fft / 2;
    for (int m = 0; m < fb->n_mels; ++m) {
        double energy = 0.0;
        for (int j = fb->row_ptr[m]; j < fb->row_ptr[m + 1]; ++j) {
            const int b = fb->bin[j];
            const double power = (b < half) ? (stft[b].re * stft[b].re + stft[b].im * stft[b].im) : 0.0;
            energy += fb->w[j] * power;
        }
        mel_buf[m] = log10(fmax(energy, 1e-10));
    }
    double mmax = -INFINITY;
    for (int m = 0; m < fb->n_mels; ++m) mmax = fmax(mmax, mel_buf[m]);
    mmax -= 8.0;
    for (int m = 0; m < fb->n_mels; ++m) out_row[m] = (float)((fmax(mel_buf[m], mmax) + 4.0) / 4.0);
}

/* ------------------------------------------------------------------------- */
/* src/stft.rs:119-138 Spectrogram::compute_mel_spectrogram_cpu                 */
/*   -> compute_all_cpu (89-115) -> frame_windows (147-169) -> MelSpectrogram   */
/* out: [frames][n_mels] f32 row-major.  Returns the number of frames.          */
/* ------------------------------------------------------------------------- */
int64_t oracle_compute_mel_spectrogram_cpu(const float *samples, int64_t len, int fft_size, int hop_size,
                                           int n_mels, double sampling_rate, float *out) {
    const int64_t frames = oracle_num_frames(len, fft_size, hop_size);
    if (frames == 0) return 0;
    const int bins = fft_size / 2 + 1;
    double *window = (double *)malloc(sizeof(double) * (size_t)fft_size);
    double *dense = (double *)malloc(sizeof(double) * (size_t)n_mels * bins);
    double *mel_buf = (double *)malloc(sizeof(double) * (size_t)n_mels);
    cplx *buf = (cplx *)malloc(sizeof(cplx) * (size_t)fft_size);
    oracle_hann_window(fft_size, window);
    /* MelSpectrogram::new: mel(sr, fft, n_mels, None, None, false, true)  (src/mel.rs:19-24) */
    oracle_mel_filterbank(sampling_rate, fft_size, n_mels, -1.0, -1.0, 0, 1, dense);
    sparse_fb fb; sparse_from_dense(&fb, dense, n_mels, bins);
    fft_plan pl; fft_plan_init(&pl, fft_size);
    for (int64_t f = 0; f < frames; ++f) {
        const float *x = samples + f * hop_size;
        for (int i = 0; i < fft_size; ++i) { buf[i].re = (double)x[i] * window[i]; buf[i].im = 0.0; }
        fft_forward_inplace(&pl, buf);
        mel_add_frame(&fb, buf, fft_size, mel_buf, out + f * n_mels);
    }
    fft_plan_free(&pl); sparse_free(&fb);
    free(window); free(dense); free(mel_buf); free(buf);
    return frames;
}

/* src/mel.rs:480-544 interleave_frames on single-column frames ([n_frames][n_mels] f32 in).
 * Returns the output width W (columns per mel row); out must hold n_mels * W floats.
 *   min_width > 0 and n_frames odd -> one zero frame appended (whisper.cpp needs an even count);
 *   then zero columns are appended up to min_width.
 *   major_column_order != 0 -> [frame][mel] (waterfall); 0 -> [mel][frame] (what whisper.cpp expects). */
int64_t oracle_interleave_frames(const float *frames, int64_t n_frames, int n_mels, int major_column_order,
                                 int64_t min_width, float *out) {
    if (n_frames <= 0 || (min_width % 2) != 0) return -1;          /* the reference asserts both */
    int64_t nf = n_frames;
    if (min_width > 0 && (nf % 2) != 0) nf += 1;
    const int64_t padding = min_width > nf ? min_width - nf : 0;
    const int64_t W = nf + padding;
    if (!out) return W;
    for (int64_t i = 0; i < W * n_mels; ++i) out[i] = 0.0f;
    for (int64_t f = 0; f < n_frames; ++f)
        for (int m = 0; m < n_mels; ++m) {
            if (major_column_order) out[f * n_mels + m] = frames[f * n_mels + m];
            else out[(int64_t)m * W + f] = frames[f * n_mels + m];
        }
    return W;
}

/* ---- src/quant.rs: 8-bit quantisation and the TGA container ----------------------------------------
 * quantize (src/quant.rs:140-153): min/max by f32::min/f32::max folds from +/-inf (a NaN operand is
 * ignored), scale = 255/(max-min) in f32, pixel = round_half_away((v-min)*scale) clamped to [0,255]
 * (f32::max(NaN,0)=0, so NaN -> 0; max==min gives scale=inf and 0*inf=NaN -> 0). */
void oracle_quantize(const float *frame, int64_t n, uint8_t *out, float *range) {
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = 0; i < n; ++i) { mn = fminf(mn, frame[i]); mx = fmaxf(mx, frame[i]); }
    const float scale = 255.0f / (mx - mn);
    for (int64_t i = 0; i < n; ++i) {
        const float d = frame[i] - mn;
        const float p = d * scale;
        const float r = fminf(fmaxf(roundf(p), 0.0f), 255.0f);
        out[i] = (uint8_t)r;
    }
    range[0] = mn; range[1] = mx;
}

/* dequantize (src/quant.rs:156-165): scale = (max-min)/255 in f32; value = u8 as f32 * scale + min,
 * the product rounded before the sum (Rust does not contract; this file is built with -ffp-contract=off). */
void oracle_dequantize(const uint8_t *data, int64_t n, const float *range, float *out) {
    const float scale = (range[1] - range[0]) / 255.0f;
    for (int64_t i = 0; i < n; ++i) {
        const float p = (float)data[i] * scale;
        out[i] = p + range[0];
    }
}

/* tga_8bit_data (src/quant.rs:38-64): 18-byte TARGA header (ID length 8, type 3 = uncompressed grey,
 * width = len/n_mels and height = n_mels as little-endian u16, 8 bpp), the 8-byte ID field {min,max} as
 * little-endian f32, then the pixels.  out must hold 26 + n bytes; returns that size. */
int64_t oracle_tga_8bit_data(const float *data, int64_t n, int n_mels, uint8_t *out) {
    float range[2];
    oracle_quantize(data, n, out + 26, range);
    const uint16_t width = (uint16_t)(n / n_mels), height = (uint16_t)n_mels;
    memset(out, 0, 18);
    out[0] = 8; out[2] = 3;
    out[12] = (uint8_t)(width & 0xff); out[13] = (uint8_t)(width >> 8);
    out[14] = (uint8_t)(height & 0xff); out[15] = (uint8_t)(height >> 8);
    out[16] = 8;
    memcpy(out + 18, &range[0], 4);       /* x86-64 / gfx950 hosts are little-endian */
    memcpy(out + 22, &range[1], 4);
    return 26 + n;
}

/* ---- src/vad.rs: column classification of a mel image ---------------------------------------------
 * vad_boundaries (src/vad.rs:256-340) on one [height][width] image (the reference concatenates its
 * Array2<f64> frames along the time axis; values are f32 widened to f64, to_array2 src/quant.rs:168-174):
 * raw[x], x < width-2: at least min_y of the rows y in [min(min_mel, height-2), height-2) have a 3x3 Sobel
 * gradient (sobel_gradient_sq, :470-486) with gx^2+gy^2 >= min_energy^2 (classify_columns_in_frame,
 * :373-415); smoothed = moving-window majority vote over [x-4, x+4] clipped to the mask (smooth_mask,
 * :343-360).  Returns the mask length (0 when height < 3 or width < 3). */
int64_t oracle_vad_boundaries(const float *img, int height, int64_t width, int min_mel, int min_y, double min_energy,
                              uint8_t *raw, uint8_t *smoothed) {
    if (height < 3 || width < 3) return 0;
    const int64_t n = width - 2;
    const double thr = min_energy * min_energy;
    const int start_y = min_mel < height - 2 ? min_mel : height - 2;
    for (int64_t x = 0; x < n; ++x) {
        int count = 0;
        uint8_t active = min_y == 0;
        for (int y = start_y; y < height - 2 && !active; ++y) {
            const float *r0 = img + (int64_t)y * width + x, *r1 = r0 + width, *r2 = r1 + width;
            const double tl = r0[0], tc = r0[1], tr = r0[2], ml = r1[0], mr = r1[2], bl = r2[0], bc = r2[1], br = r2[2];
            const double gx = (tr + (2.0 * mr) + br) - (tl + (2.0 * ml) + bl);
            const double gy = (bl + (2.0 * bc) + br) - (tl + (2.0 * tc) + tr);
            if ((gx * gx) + (gy * gy) >= thr && ++count >= min_y) active = 1;
        }
        raw[x] = active;
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t start = i >= 4 ? i - 4 : 0, end = i + 5 < n ? i + 5 : n;
        int64_t c = 0;
        for (int64_t k = start; k < end; ++k) c += raw[k];
        smoothed[i] = (uint8_t)(c * 2 >= end - start);
    }
    return n;
}

/* longest run of consecutive set columns; vad_on(edge_info, n) (src/vad.rs:229-254) == (n <= 1 ? any set : run >= n) */
int64_t oracle_vad_longest_run(const uint8_t *mask, int64_t n) {
    int64_t best = 0, cur = 0;
    for (int64_t i = 0; i < n; ++i) { cur = mask[i] ? cur + 1 : 0; if (cur > best) best = cur; }
    return best;
}

/* Many clips, clips split across OpenMP threads (the all-cores CPU baseline of
 * bench.py).  Clip c is samples[c*clip_stride .. +clip_len); out is
 * [clip][frame][mel].  Same arithmetic as the function above. */
int64_t oracle_compute_mel_batch(const float *samples, int64_t clip_stride, int64_t clip_len, int n_clips,
                                 int fft_size, int hop_size, int n_mels, double sampling_rate,
                                 float *out, int n_threads) {
    const int64_t fpc = oracle_num_frames(clip_len, fft_size, hop_size);
    if (fpc == 0) return 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(static)
#endif
    for (int c = 0; c < n_clips; ++c) {
        oracle_compute_mel_spectrogram_cpu(samples + (int64_t)c * clip_stride, clip_len, fft_size, hop_size,
                                           n_mels, sampling_rate, out + (int64_t)c * fpc * n_mels);
    }
    (void)n_threads;
    return fpc * n_clips;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Streaming path: src/stft.rs:25-86 Spectrogram::{new,add} driven hop-by-hop   */
/* as src/rb.rs:86-121 RingBuffer::maybe_mel does, then MelSpectrogram::add.    */
/* out: [emitted][n_mels]; returns frames emitted.  This is what                */
/* testdata/rust_jfk_golden.npy pins (src/rb.rs:134-179, 512/160/80).           */
/* ------------------------------------------------------------------------- */
/* flush_tail != 0: after the full hops, the remaining (< hop) samples go through one more add(), which
 * zero-pads them and advances idx by their count only (src/stft.rs:55-66). */
int64_t oracle_stream_mel_ex(const float *samples, int64_t len, int fft_size, int hop_size, int n_mels,
                             double sampling_rate, float *out, int64_t out_cap_frames, int flush_tail) {
    const int bins = fft_size / 2 + 1;
    double *window = (double *)malloc(sizeof(double) * (size_t)fft_size);
    double *hop_buf = (double *)calloc((size_t)fft_size, sizeof(double));
    double *dense = (double *)malloc(sizeof(double) * (size_t)n_mels * bins);
    double *mel_buf = (double *)malloc(sizeof(double) * (size_t)n_mels);
    cplx *buf = (cplx *)malloc(sizeof(cplx) * (size_t)fft_size);
    oracle_hann_window(fft_size, window);
    oracle_mel_filterbank(sampling_rate, fft_size, n_mels, -1.0, -1.0, 0, 1, dense);
    sparse_fb fb; sparse_from_dense(&fb, dense, n_mels, bins);
    fft_plan pl; fft_plan_init(&pl, fft_size);
    uint64_t idx = 0;
    int64_t emitted = 0;
    /* RingBuffer::maybe_mel only calls Spectrogram::add with exactly hop_size samples. */
    for (int64_t pos = 0; pos < len; pos += hop_size) {
        const int64_t got = len - pos < hop_size ? len - pos : hop_size;
        if (got < hop_size && !flush_tail) break;
        memmove(hop_buf, hop_buf + hop_size, sizeof(double) * (size_t)(fft_size - hop_size));
        for (int i = 0; i < hop_size; ++i) hop_buf[fft_size - hop_size + i] = i < got ? (double)samples[pos + i] : 0.0;
        idx += (uint64_t)got;
        if (idx >= (uint64_t)fft_size) {
            if (emitted >= out_cap_frames) break;
            for (int j = 0; j < fft_size; ++j) { buf[j].re = hop_buf[j] * window[j]; buf[j].im = 0.0; }
            fft_forward_inplace(&pl, buf);
            mel_add_frame(&fb, buf, fft_size, mel_buf, out + emitted * n_mels);
            ++emitted;
        }
    }
    fft_plan_free(&pl); sparse_free(&fb);
    free(window); free(hop_buf); free(dense); free(mel_buf); free(buf);
    return emitted;
}

int64_t oracle_stream_mel(const float *samples, int64_t len, int fft_size, int hop_size, int n_mels,
                          double sampling_rate, float *out, int64_t out_cap_frames) {
    return oracle_stream_mel_ex(samples, len, fft_size, hop_size, n_mels, sampling_rate, out, out_cap_frames, 0);
}

/* ------------------------------------------------------------------------- */
/* Kaldi fbank: src/fbank.rs                                                   */
/* ------------------------------------------------------------------------- */
typedef struct {
    double sample_rate;      /* FbankConfig (src/fbank.rs:25-64) */
    int num_mel_bins;
    double frame_length_ms;
    double frame_shift_ms;
    double energy_floor;
    int use_log_fbank;
    int use_power;
    double preemphasis;
    int apply_cmn;
    double low_freq;
    double high_freq;
} oracle_fbank_config;

void oracle_fbank_default_config(oracle_fbank_config *c) { /* src/fbank.rs:46-64 */
    c->sample_rate = 16000.0; c->num_mel_bins = 80; c->frame_length_ms = 25.0; c->frame_shift_ms = 10.0;
    c->energy_floor = 0.0; c->use_log_fbank = 1; c->use_power = 1; c->preemphasis = 0.97;
    c->apply_cmn = 1; c->low_freq = 20.0; c->high_freq = 0.0;
}
/* src/fbank.rs:66-82 */
int oracle_fbank_frame_length(const oracle_fbank_config *c) { return (int)round((c->frame_length_ms / 1000.0) * c->sample_rate); }
int oracle_fbank_frame_shift(const oracle_fbank_config *c) { return (int)round((c->frame_shift_ms / 1000.0) * c->sample_rate); }
int oracle_fbank_fft_size(const oracle_fbank_config *c) {
    int n = oracle_fbank_frame_length(c), p = 1;
    while (p < n) p <<= 1;
    return p;
}

/* src/fbank.rs:303-313 */
static double kaldi_hz_to_mel(double hz) { return 1127.0 * log(1.0 + hz / 700.0); }
static double kaldi_mel_to_hz(double mel) { return 700.0 * (exp(mel / 1127.0) - 1.0); }

/* src/fbank.rs:253-301 kaldi_mel_filterbank -> dense [num_mel_bins, fft_size/2+1] */
void oracle_kaldi_mel_filterbank(double sample_rate, int fft_size, int num_mel_bins, double low_freq,
                                 double high_freq, double *filters) {
    const int nb = fft_size / 2 + 1;
    const double mel_low = kaldi_hz_to_mel(low_freq), mel_high = kaldi_hz_to_mel(high_freq);
    double *hz = (double *)malloc(sizeof(double) * (size_t)(num_mel_bins + 2));
    for (int i = 0; i <= num_mel_bins + 1; ++i) {
        double mp = mel_low + (mel_high - mel_low) * (double)i / (double)(num_mel_bins + 1);
        hz[i] = kaldi_mel_to_hz(mp);
    }
    memset(filters, 0, sizeof(double) * (size_t)num_mel_bins * nb);
    for (int m = 0; m < num_mel_bins; ++m) {
        const double left = hz[m], center = hz[m + 1], right = hz[m + 2];
        if (center <= left || right <= center) continue;
        for (int b = 0; b < nb; ++b) {
            const double f = (double)b * sample_rate / (double)fft_size;
            if (f > left && f <= center) filters[(size_t)m * nb + b] = (f - left) / (center - left);
            else if (f > center && f < right) filters[(size_t)m * nb + b] = (right - f) / (right - center);
        }
    }
    free(hz);
}

/* src/fbank.rs:141-236 Fbank::compute -> out [frames][num_mel_bins] f32. Returns frames. */
int64_t oracle_fbank_compute(const oracle_fbank_config *cfg, const float *samples, int64_t len, float *out) {
    const int frame_len = oracle_fbank_frame_length(cfg);
    const int frame_shift = oracle_fbank_frame_shift(cfg);
    const int fft_size = oracle_fbank_fft_size(cfg);
    const int nb = fft_size / 2 + 1;
    const int nm = cfg->num_mel_bins;
    const double preemph = cfg->preemphasis;
    if (len < frame_len) return 0;
    const int64_t num_frames = 1 + (len - frame_len) / frame_shift;

    /* Fbank::new (src/fbank.rs:94-132): povey window, filterbank, plan */
    double *window = (double *)malloc(sizeof(double) * (size_t)frame_len);
    for (int i = 0; i < frame_len; ++i) {
        double a = 2.0 * M_PI * (double)i / (double)(frame_len - 1);
        window[i] = pow(0.5 - 0.5 * cos(a), 0.85);
    }
    const double high = (cfg->high_freq == 0.0) ? cfg->sample_rate / 2.0 : cfg->high_freq;
    double *dense = (double *)malloc(sizeof(double) * (size_t)nm * nb);
    oracle_kaldi_mel_filterbank(cfg->sample_rate, fft_size, nm, cfg->low_freq, high, dense);
    sparse_fb fb; sparse_from_dense(&fb, dense, nm, nb);
    fft_plan pl; fft_plan_init(&pl, fft_size);

    cplx *cbuf = (cplx *)malloc(sizeof(cplx) * (size_t)fft_size);
    double *frame_buf = (double *)malloc(sizeof(double) * (size_t)frame_len);
    double *power = (double *)malloc(sizeof(double) * (size_t)nb);

    for (int64_t f = 0; f < num_frames; ++f) {
        const int64_t start = f * frame_shift;
        const float *x = samples + start;
        double mean = 0.0;
        for (int i = 0; i < frame_len; ++i) mean += (double)x[i];
        mean /= (double)frame_len;
        for (int i = 0; i < frame_len; ++i) frame_buf[i] = (double)x[i] - mean;
        if (preemph > 0.0) {
            for (int i = frame_len - 1; i >= 1; --i) frame_buf[i] -= preemph * frame_buf[i - 1];
            if (start > 0) frame_buf[0] -= preemph * ((double)samples[start - 1] - mean);
        }
        for (int i = 0; i < frame_len; ++i) { cbuf[i].re = frame_buf[i] * window[i]; cbuf[i].im = 0.0; }
        for (int i = frame_len; i < fft_size; ++i) { cbuf[i].re = 0.0; cbuf[i].im = 0.0; }
        fft_forward_inplace(&pl, cbuf);
        for (int b = 0; b < nb; ++b) {
            double ns = cbuf[b].re * cbuf[b].re + cbuf[b].im * cbuf[b].im;
            power[b] = cfg->use_power ? ns : sqrt(ns);
        }
        /* project_power_f64 (src/mel.rs:106-125) */
        for (int m = 0; m < nm; ++m) {
            double e = 0.0;
            for (int j = fb.row_ptr[m]; j < fb.row_ptr[m + 1]; ++j) e += fb.w[j] * power[fb.bin[j]];
            const double floor_v = (cfg->energy_floor > 0.0) ? cfg->energy_floor : (double)FLT_EPSILON;
            e = fmax(e, floor_v);
            if (cfg->use_log_fbank) e = log(e);
            out[f * nm + m] = (float)e;
        }
    }
    /* CMN (src/fbank.rs:224-233): per mel column, f32 mean = sequential f32 sum / n
     * (ndarray's mean() on a strided column view folds left-to-right in f32). */
    if (cfg->apply_cmn && num_frames > 0) {
        for (int m = 0; m < nm; ++m) {
            float sum = 0.0f;
            for (int64_t f = 0; f < num_frames; ++f) sum = sum + out[f * nm + m];
            const float mean = sum / (float)num_frames;
            for (int64_t f = 0; f < num_frames; ++f) out[f * nm + m] -= mean;
        }
    }
    fft_plan_free(&pl); sparse_free(&fb);
    free(window); free(dense); free(cbuf); free(frame_buf); free(power);
    return num_frames;
}

/* Many clips through fbank, clips across OpenMP threads. out [clip][frame][mel]. */
int64_t oracle_fbank_batch(const oracle_fbank_config *cfg, const float *samples, int64_t clip_stride,
                           int64_t clip_len, int n_clips, float *out, int n_threads) {
    const int frame_len = oracle_fbank_frame_length(cfg);
    const int frame_shift = oracle_fbank_frame_shift(cfg);
    if (clip_len < frame_len) return 0;
    const int64_t fpc = 1 + (clip_len - frame_len) / frame_shift;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(static)
#endif
    for (int c = 0; c < n_clips; ++c)
        oracle_fbank_compute(cfg, samples + (int64_t)c * clip_stride, clip_len, out + (int64_t)c * fpc * cfg->num_mel_bins);
    (void)n_threads;
    return fpc * n_clips;
}

/* ------------------------------------------------------------------------- */
/* BatchLogMelSpectrogram (NeMo/Parakeet-style frontend): src/mel.rs:171-418,   */
/* 656-756.  The reference computes this path in f32 (rustfft f32, f32 window,  */
/* f32 projection); `oracle_blm_compute_f32` restates it literally in f32 with  */
/* this file's own FFT, `oracle_blm_compute_f64` evaluates the same definition  */
/* in f64 (only the f32-rounded inputs the reference also has -- window table,  */
/* pre-emphasised waveform, f32 weights -- are kept) and is what the GPU path   */
/* is gated against.  The reference's tests pin only the output shape           */
/* (src/mel.rs:943-961): values are PARITY UNPINNED for this path.              */
/* ------------------------------------------------------------------------- */
typedef struct {
    int sample_rate, n_fft, win_length, hop_length, n_mels;   /* BatchLogMelConfig, src/mel.rs:171-208 */
    double f_min, f_max;                                      /* f_max <= 0 == None (sample_rate/2)     */
    int htk, norm;
    float preemphasis;
    int center;
    float log_zero_guard;
    int pad_to;
    int normalize_per_feature;
} oracle_blm_config;

void oracle_blm_default_config(oracle_blm_config *c) {   /* src/mel.rs:189-208 */
    c->sample_rate = 16000; c->n_fft = 512; c->win_length = 400; c->hop_length = 160; c->n_mels = 80;
    c->f_min = 0.0; c->f_max = -1.0; c->htk = 0; c->norm = 1; c->preemphasis = 0.0f; c->center = 1;
    c->log_zero_guard = FLT_EPSILON; c->pad_to = 0; c->normalize_per_feature = 0;
}
int64_t oracle_blm_num_frames(const oracle_blm_config *c, int64_t len) {   /* src/mel.rs:387-395 */
    if (len == 0) return 0;                                                  /* src/mel.rs:326-332 */
    if (c->center) return len / c->hop_length + 1;
    if (len < c->n_fft) return 0;
    return (len - c->n_fft) / c->hop_length + 1;
}
int64_t oracle_blm_padded_frames(const oracle_blm_config *c, int64_t frames) {   /* pad_len, src/mel.rs:751-756 */
    if (c->pad_to == 0) return frames;
    return ((frames + c->pad_to - 1) / c->pad_to) * c->pad_to;
}

typedef struct { float re, im; } cplxf;
typedef struct { int n; cplxf *tw; cplxf *scratch; } fft_planf;
static void fft_planf_init(fft_planf *p, int n) {
    p->n = n; p->tw = (cplxf *)malloc(sizeof(cplxf) * (size_t)n); p->scratch = (cplxf *)malloc(sizeof(cplxf) * (size_t)n);
    for (int j = 0; j < n; ++j) { double a = -2.0 * M_PI * (double)j / (double)n; p->tw[j].re = (float)cos(a); p->tw[j].im = (float)sin(a); }
}
static void fft_planf_free(fft_planf *p) { free(p->tw); free(p->scratch); }
static inline cplxf cmulf(cplxf a, cplxf b) { cplxf r = { a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re }; return r; }
static inline cplxf caddf(cplxf a, cplxf b) { cplxf r = { a.re + b.re, a.im + b.im }; return r; }
static inline cplxf csubf(cplxf a, cplxf b) { cplxf r = { a.re - b.re, a.im - b.im }; return r; }
static void fft_recf(const fft_planf *pl, int n, int stride, const cplxf *in, cplxf *out) {
    if (n == 1) { out[0] = in[0]; return; }
    const int N = pl->n, p = pick_radix(n), m = n / p;
    for (int q = 0; q < p; ++q) fft_recf(pl, m, stride * p, in + (size_t)q * stride, out + (size_t)q * m);
    const int tstep = N / n, pstep = N / p;
    cplxf t[64];
    for (int k = 0; k < m; ++k) {
        for (int q = 0; q < p; ++q) t[q] = q ? cmulf(out[q * m + k], pl->tw[(q * k) * tstep]) : out[k];
        for (int r = 0; r < p; ++r) {
            cplxf acc = t[0];
            for (int q = 1; q < p; ++q) acc = caddf(acc, cmulf(t[q], pl->tw[((q * r) % p) * pstep]));
            out[r * m + k] = acc;
        }
    }
    (void)csubf;
}

/* shared front part: pre-emphasised waveform (f32, src/mel.rs:696-706), f32 window (708-719), f32 weights */
static void blm_prepare(const oracle_blm_config *c, const float *samples, int64_t len, float *wave, float *window,
                        sparse_fb *fb, double **dense_out) {
    memcpy(wave, samples, sizeof(float) * (size_t)len);
    if (len > 0 && c->preemphasis != 0.0f) {
        float prev = wave[0];
        for (int64_t i = 1; i < len; ++i) { float cur = wave[i]; wave[i] = cur - (c->preemphasis * prev); prev = cur; }
    }
    for (int i = 0; i < c->n_fft; ++i) window[i] = 0.0f;
    if (c->win_length > 1) {
        const int offset = (c->n_fft - c->win_length) / 2;
        const float pi_f32 = 3.14159265358979323846f;
        for (int i = 0; i < c->win_length; ++i) {
            float phase = (2.0f * pi_f32 * (float)i) / ((float)c->win_length - 1.0f);
            window[offset + i] = 0.5f - (0.5f * cosf(phase));
        }
    }
    const int bins = c->n_fft / 2 + 1;
    const double fmax_ = c->f_max > 0.0 ? c->f_max : (double)c->sample_rate / 2.0;
    double *dense = (double *)malloc(sizeof(double) * (size_t)c->n_mels * bins);
    /* SparseMelFilterbank::from_mel(sr, n_fft, n_mels, Some(f_min), Some(f_max), htk, norm), src/mel.rs:254-263 */
    {
        /* mel() with explicit f_min (may be 0.0 exactly, which the None-encoding of oracle_mel_filterbank also means) */
        oracle_mel_filterbank((double)c->sample_rate, c->n_fft, c->n_mels, c->f_min > 0.0 ? c->f_min : -1.0, fmax_, c->htk, c->norm, dense);
    }
    sparse_from_dense(fb, dense, c->n_mels, bins);
    *dense_out = dense;
}

static void blm_normalize(float *features, int n_mels, int64_t valid, int64_t padded) {   /* src/mel.rs:721-749 */
    if (valid == 0) return;
    for (int m = 0; m < n_mels; ++m) {
        float *row = features + (int64_t)m * padded;
        float sum = 0.0f;
        for (int64_t f = 0; f < valid; ++f) sum += row[f];
        const float mean = sum / (float)valid;
        float denom = (float)valid - 1.0f; if (denom < 1.0f) denom = 1.0f;
        float var = 0.0f;
        for (int64_t f = 0; f < valid; ++f) { float d = row[f] - mean; var += d * d; }
        var = var / denom;
        const float sd = sqrtf(var) + 1e-5f;
        for (int64_t f = 0; f < valid; ++f) row[f] = (row[f] - mean) / sd;
    }
}

/* normalize_per_feature (src/mel.rs:721-749) on its own: the reference's f32 left folds applied in place to a [n_mels][padded] image --
 * how the tests check the device normaliser on the device's own un-normalised rows, ill-conditioned ones included. */
void oracle_blm_normalize(float *features, int n_mels, int64_t valid, int64_t padded) { blm_normalize(features, n_mels, valid, padded); }

/* out: [n_mels][padded_frames] f32.  Returns padded_frames (cols); *valid_out = valid frames. */
int64_t oracle_blm_compute_f32(const oracle_blm_config *c, const float *samples, int64_t len, float *out, int64_t *valid_out) {
    const int64_t valid = oracle_blm_num_frames(c, len), padded = oracle_blm_padded_frames(c, valid);
    if (valid_out) *valid_out = valid;
    if (len == 0 || !out) return padded;
    const int N = c->n_fft, bins = N / 2 + 1, pad = c->center ? N / 2 : 0;
    float *wave = (float *)malloc(sizeof(float) * (size_t)len), *window = (float *)malloc(sizeof(float) * (size_t)N);
    float *power = (float *)malloc(sizeof(float) * (size_t)bins);
    cplxf *buf = (cplxf *)malloc(sizeof(cplxf) * (size_t)N);
    sparse_fb fb; double *dense;
    blm_prepare(c, samples, len, wave, window, &fb, &dense);
    fft_planf pl; fft_planf_init(&pl, N);
    for (int64_t i = 0; i < (int64_t)c->n_mels * padded; ++i) out[i] = 0.0f;
    for (int64_t f = 0; f < valid; ++f) {
        const int64_t start = f * c->hop_length;
        for (int i = 0; i < N; ++i) {
            const int64_t s = start + i - pad;                 /* index into the un-padded waveform */
            const float v = (s >= 0 && s < len) ? wave[s] : 0.0f;
            buf[i].re = v * window[i]; buf[i].im = 0.0f;
        }
        memcpy(pl.scratch, buf, sizeof(cplxf) * (size_t)N);
        fft_recf(&pl, N, 1, pl.scratch, buf);
        for (int b = 0; b < bins; ++b) power[b] = buf[b].re * buf[b].re + buf[b].im * buf[b].im;
        for (int m = 0; m < c->n_mels; ++m) {                  /* project_power_f32, src/mel.rs:127-146 */
            float e = 0.0f;
            for (int j = fb.row_ptr[m]; j < fb.row_ptr[m + 1]; ++j) e += (float)fb.w[j] * power[fb.bin[j]];
            out[(int64_t)m * padded + f] = logf(e + c->log_zero_guard);
        }
    }
    if (c->normalize_per_feature) blm_normalize(out, c->n_mels, valid, padded);
    fft_planf_free(&pl); sparse_free(&fb);
    free(dense); free(wave); free(window); free(power); free(buf);
    return padded;
}

/* Same definition with f64 FFT / power / projection / ln; keeps the f32-rounded inputs of the reference. */
int64_t oracle_blm_compute_f64(const oracle_blm_config *c, const float *samples, int64_t len, float *out, int64_t *valid_out) {
    const int64_t valid = oracle_blm_num_frames(c, len), padded = oracle_blm_padded_frames(c, valid);
    if (valid_out) *valid_out = valid;
    if (len == 0 || !out) return padded;
    const int N = c->n_fft, bins = N / 2 + 1, pad = c->center ? N / 2 : 0;
    float *wave = (float *)malloc(sizeof(float) * (size_t)len), *window = (float *)malloc(sizeof(float) * (size_t)N);
    double *power = (double *)malloc(sizeof(double) * (size_t)bins);
    cplx *buf = (cplx *)malloc(sizeof(cplx) * (size_t)N);
    sparse_fb fb; double *dense;
    blm_prepare(c, samples, len, wave, window, &fb, &dense);
    fft_plan pl; fft_plan_init(&pl, N);
    for (int64_t i = 0; i < (int64_t)c->n_mels * padded; ++i) out[i] = 0.0f;
    for (int64_t f = 0; f < valid; ++f) {
        const int64_t start = f * c->hop_length;
        for (int i = 0; i < N; ++i) {
            const int64_t s = start + i - pad;
            const double v = (s >= 0 && s < len) ? (double)wave[s] : 0.0;
            buf[i].re = v * (double)window[i]; buf[i].im = 0.0;
        }
        fft_forward_inplace(&pl, buf);
        for (int b = 0; b < bins; ++b) power[b] = buf[b].re * buf[b].re + buf[b].im * buf[b].im;
        for (int m = 0; m < c->n_mels; ++m) {
            double e = 0.0;
            for (int j = fb.row_ptr[m]; j < fb.row_ptr[m + 1]; ++j) e += (double)(float)fb.w[j] * power[fb.bin[j]];
            out[(int64_t)m * padded + f] = (float)log(e + (double)c->log_zero_guard);
        }
    }
    if (c->normalize_per_feature) blm_normalize(out, c->n_mels, valid, padded);
    fft_plan_free(&pl); sparse_free(&fb);
    free(dense); free(wave); free(window); free(power); free(buf);
    return padded;
}

/* ------------------------------------------------------------------------- */
/* Synthetic PCM generator of SURVEY.md §8(d): bit-identical on CPU and GPU,    */
/* no libm.  x[clip][i] = u * 2^-(clip & 7), u in [-1,1).                       */
/* ------------------------------------------------------------------------- */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h;
}
void oracle_synth_pcm(uint32_t seed, uint64_t clip, uint64_t n, float *out) {
    const float scale = 1.0f / (float)(1u << (clip & 7u));
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t h = fmix32(seed ^ ((uint32_t)clip * 0x9E3779B1u) ^ ((uint32_t)i * 0x85EBCA6Bu));
        float u = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        out[i] = u * scale;
    }
}
