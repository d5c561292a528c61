//! `src/hip.rs` for wavey-ai/mel-spec -- the MI355X (gfx950) backend behind feature `hip`.
//!
//! NOT COMPILED IN THIS REPOSITORY (the build image has no Rust toolchain); it is the binding a
//! maintainer drops into the reference crate.  It is deliberately thin: all work happens behind the
//! C ABI of `libmelspec_hip.so` (include/melspec_hip.h).  Public surface mirrors
//! `CudaMelSpectrogram` (src/cuda.rs:27-140): same constructor arguments, same method, same error split.
use std::ffi::{c_char, c_int, c_void, CStr};
use std::fmt;

/// Same split as `CudaError` (src/cuda.rs:10-25): construction problems are `Unavailable(&'static str)` so that callers and
/// tests can skip (src/cuda.rs:512-518), per-call problems are `Runtime(String)` with the library's message.
#[derive(Debug)]
pub enum HipError {
    Runtime(String),
    Unavailable(&'static str),
}

/// The library reports construction failures as codes (include/melspec_hip.h); `Unavailable` wants a `&'static str`.
fn unavailable(rc: c_int) -> HipError {
    HipError::Unavailable(match rc {
        -1 => "fft_size, hop_size, and n_mels must be non-zero", // src/cuda.rs:45-49
        -2 => "no gfx950 (MI355X) device visible to the HIP runtime",
        -4 => "geometry outside what the HIP kernels cover",
        _ => "HIP backend failed to initialise",
    })
}

impl fmt::Display for HipError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            Self::Runtime(m) => write!(f, "HIP error: {m}"),
            Self::Unavailable(m) => write!(f, "HIP unavailable: {m}"),
        }
    }
}
impl std::error::Error for HipError {}

#[repr(C)]
struct Ctx {
    _private: [u8; 0],
}

#[link(name = "melspec_hip")]
unsafe extern "C" {
    fn melspec_create(out: *mut *mut Ctx, device: c_int, fft: c_int, hop: c_int, sr: f64, n_mels: c_int) -> c_int;
    fn melspec_destroy(ctx: *mut Ctx);
    fn melspec_num_frames(ctx: *const Ctx, n_samples: usize) -> usize;
    fn melspec_compute_host(ctx: *mut Ctx, samples: *const f32, n: usize, out: *mut f32, cap: usize, frames: *mut usize) -> c_int;
    fn melspec_compute_uniform_device(ctx: *mut Ctx, d_pcm: *const f32, stride: u64, len: u64, n_clips: u32,
                                      d_out: *mut f32, stream: *mut c_void) -> c_int;
    fn melspec_synchronize(ctx: *mut Ctx, stream: *mut c_void) -> c_int;
    fn melspec_set_precise(ctx: *mut Ctx, on: c_int) -> c_int;
    fn melspec_set_precision(ctx: *mut Ctx, mode: c_int) -> c_int;
    fn melspec_set_auto_adaptive(ctx: *mut Ctx, on: c_int) -> c_int;
    fn melspec_auto_state(ctx: *mut Ctx, heavy: *mut c_int, fraction: *mut f64) -> c_int;
    fn melspec_create_with_filterbank(out: *mut *mut Ctx, device: c_int, fft: c_int, hop: c_int, sr: f64, n_mels: c_int,
                                      f_min: f64, f_max: f64, htk: c_int, norm: c_int) -> c_int;
    fn melspec_create_with_dense_filterbank(out: *mut *mut Ctx, device: c_int, fft: c_int, hop: c_int, sr: f64, n_mels: c_int,
                                            filters: *const f64, fft_bins: c_int) -> c_int;
    fn melspec_max_frames_per_batch(ctx: *const Ctx) -> usize;
    fn melspec_compute_batch_host(ctx: *mut Ctx, samples: *const f32, offsets: *const u64, lengths: *const u64, n_clips: u32,
                                  out: *mut f32, out_offsets: *const u64, cap: usize, total_frames: *mut u64) -> c_int;
    fn melspec_stft_bins(ctx: *const Ctx, full: c_int) -> usize;
    fn melspec_mel_from_stft_host(ctx: *mut Ctx, spec: *const c_void, dtype: c_int, full: c_int, n_frames: usize, out: *mut f32, cap: usize) -> c_int;
    fn melspec_stft_host(ctx: *mut Ctx, samples: *const f32, n: usize, out: *mut c_void, cap: usize, dtype: c_int, full: c_int,
                         frames: *mut usize) -> c_int;
    // Fbank (src/fbank.rs:85-247)
    fn melspec_fbank_create(out: *mut *mut FbankHandle, device: c_int, cfg: *const FbankConfigC) -> c_int;
    fn melspec_fbank_destroy(fb: *mut FbankHandle);
    fn melspec_fbank_num_frames(fb: *const FbankHandle, n: usize) -> usize;
    fn melspec_fbank_compute_host(fb: *mut FbankHandle, samples: *const f32, n: usize, out: *mut f32, cap: usize, frames: *mut usize) -> c_int;
    fn melspec_fbank_compute_batch_host(fb: *mut FbankHandle, samples: *const f32, offsets: *const u64, lengths: *const u64, n_clips: u32,
                                        out: *mut f32, out_offsets: *const u64, cap: usize, total_frames: *mut u64) -> c_int;
    // BatchLogMelSpectrogram (src/mel.rs:171-418)
    fn melspec_blm_create(out: *mut *mut BlmHandle, device: c_int, cfg: *const BlmConfigC) -> c_int;
    fn melspec_blm_destroy(b: *mut BlmHandle);
    fn melspec_blm_set_precision(b: *mut BlmHandle, mode: c_int) -> c_int;
    fn melspec_blm_precision(b: *const BlmHandle) -> c_int;
    fn melspec_blm_padded_frames(b: *const BlmHandle, n: usize) -> usize;
    fn melspec_blm_num_frames(b: *const BlmHandle, n: usize) -> usize;
    fn melspec_blm_compute_host(b: *mut BlmHandle, samples: *const f32, n: usize, out: *mut f32, cap: usize, rows: *mut usize, cols: *mut usize) -> c_int;
    fn melspec_blm_compute_batch_host(b: *mut BlmHandle, samples: *const f32, offsets: *const u64, lengths: *const u64, n_clips: u32,
                                      out: *mut f32, out_offsets: *const u64, cap: usize, total_columns: *mut u64) -> c_int;
    // vad_boundaries (src/vad.rs:251-340)
    fn melspec_vad_mask_len(n_mels: c_int, width: usize) -> usize;
    fn melspec_vad_boundaries_host(device: c_int, image: *const f32, n_mels: c_int, width: usize, settings: *const VadSettingsC,
                                   raw: *mut u8, smoothed: *mut u8, longest_run: *mut u32) -> c_int;
    fn melspec_last_error() -> *const c_char;
    // src/quant.rs on the device (tga_8bit / parse_tga_8bit / quantize / dequantize)
    fn melspec_tga_create(out: *mut *mut Tga, device: c_int) -> c_int;
    fn melspec_tga_destroy(q: *mut Tga);
    fn melspec_tga_layout(n_mels: c_int, width: usize, n_chunks: *mut u32, chunk_stride: *mut usize, last: *mut usize) -> c_int;
    fn melspec_tga_encode_host(q: *mut Tga, data: *const f32, len: usize, n_mels: c_int, out: *mut u8, cap: usize, n_chunks: *mut u32) -> c_int;
    fn melspec_tga_decode_host(q: *mut Tga, blob: *const u8, n: usize, out: *mut f32, cap: usize, n_values: *mut usize) -> c_int;
    // Spectrogram::add + RingBuffer::maybe_mel with the overlap-save state in HBM
    fn melspec_stream_create(out: *mut *mut Stream, ctx: *mut Ctx, n_streams: u32, max_chunk: u32) -> c_int;
    fn melspec_stream_destroy(st: *mut Stream);
    fn melspec_stream_frames_after(st: *const Stream, id: u32, n_new: u32) -> usize;
    fn melspec_stream_enable_vad(st: *mut Stream, settings: *const VadSettingsC) -> c_int;
    fn melspec_stream_vad_frames(st: *const Stream, id: u32) -> u64;
    fn melspec_stream_push_host_vad(st: *mut Stream, ids: *const u32, samples: *const f32, lens: *const u32, n: u32, out: *mut f32,
                                    out_capacity_floats: usize, frames_out: *mut u32, acts: *mut VadActivityC, acts_capacity: usize) -> c_int;
    fn melspec_stream_push_host(st: *mut Stream, ids: *const u32, samples: *const f32, lens: *const u32, n: u32,
                                out: *mut f32, cap: usize, frames_out: *mut u32) -> c_int;
}

#[repr(C)]
struct Tga {
    _private: [u8; 0],
}
#[repr(C)]
struct FbankHandle {
    _private: [u8; 0],
}
#[repr(C)]
struct BlmHandle {
    _private: [u8; 0],
}
/// melspec_fbank_config (include/melspec_hip.h): FbankConfig (src/fbank.rs:25-44) without `dither` and `use_energy`, which
/// `Fbank::compute` never reads.
#[repr(C)]
struct FbankConfigC {
    sample_rate: f64,
    num_mel_bins: i32,
    frame_length_ms: f64,
    frame_shift_ms: f64,
    energy_floor: f64,
    use_log_fbank: i32,
    use_power: i32,
    preemphasis: f64,
    apply_cmn: i32,
    low_freq: f64,
    high_freq: f64,
}
/// melspec_blm_config == BatchLogMelConfig (src/mel.rs:171-187); f_max <= 0 encodes None.
#[repr(C)]
struct BlmConfigC {
    sample_rate: i32,
    n_fft: i32,
    win_length: i32,
    hop_length: i32,
    n_mels: i32,
    f_min: f64,
    f_max: f64,
    htk: i32,
    norm: i32,
    preemphasis: f32,
    center: i32,
    log_zero_guard: f32,
    pad_to: i32,
    normalize_per_feature: i32,
}
/// melspec_vad_settings == DetectionSettings (src/vad.rs:5-11)
#[repr(C)]
struct VadSettingsC {
    min_energy: f64,
    min_y: c_int,
    min_x: c_int,
    min_mel: c_int,
}
/// melspec_vad_activity: one record per emitted frame
#[repr(C)]
#[derive(Clone, Copy, Default)]
struct VadActivityC {
    valid: u8,
    active: u8,
    leading_active_columns: u16,
    active_columns: u16,
    window_columns: u16,
}
#[repr(C)]
struct Stream {
    _private: [u8; 0],
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(melspec_last_error()).to_string_lossy().into_owned() }
}

pub struct HipMelSpectrogram {
    ctx: *mut Ctx, // raw pointer => !Send + !Sync, like CudaMelSpectrogram's device pointers
    n_mels: usize,
}

impl HipMelSpectrogram {
    pub fn new(fft_size: usize, hop_size: usize, sampling_rate: f64, n_mels: usize) -> Result<Self, HipError> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { melspec_create(&mut ctx, -1, fft_size as c_int, hop_size as c_int, sampling_rate, n_mels as c_int) };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { ctx, n_mels })
    }

    /// `MelSpectrogram` over `SparseMelFilterbank::from_mel(sr, n_fft, n_mels, f_min, f_max, htk, norm)` (src/mel.rs:73-87) instead of
    /// `new()`'s default bank (src/mel.rs:19-24).
    pub fn with_filterbank(fft_size: usize, hop_size: usize, sampling_rate: f64, n_mels: usize, f_min: Option<f64>, f_max: Option<f64>,
                           htk: bool, norm: bool) -> Result<Self, HipError> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe {
            melspec_create_with_filterbank(&mut ctx, -1, fft_size as c_int, hop_size as c_int, sampling_rate, n_mels as c_int,
                                           f_min.unwrap_or(-1.0), f_max.unwrap_or(-1.0), htk as c_int, norm as c_int)
        };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { ctx, n_mels })
    }

    /// The same over `SparseMelFilterbank::from_dense(filters)` (src/mel.rs:48-71): `filters` is `[n_mels][fft_size / 2 + 1]`.
    pub fn with_dense_filterbank(fft_size: usize, hop_size: usize, sampling_rate: f64, filters: &ndarray::Array2<f64>) -> Result<Self, HipError> {
        let f = filters.as_standard_layout();
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe {
            melspec_create_with_dense_filterbank(&mut ctx, -1, fft_size as c_int, hop_size as c_int, sampling_rate, f.nrows() as c_int,
                                                 f.as_ptr(), f.ncols() as c_int)
        };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { ctx, n_mels: f.nrows() })
    }

    /// The default precision mode takes a vote inside every launch (the first work unit of every wave is the sample) and hands batches
    /// in which more than 1/8 of the sampled frames need f64 (speech, tonal material) to the f64 kernel queued behind the launch: a
    /// function of the batch, not of history.  `false` switches the vote off: f32 kernel + per-frame recompute whatever the input.
    pub fn set_auto_adaptive(&mut self, on: bool) -> Result<(), HipError> {
        match unsafe { melspec_set_auto_adaptive(self.ctx, on as c_int) } {
            0 => Ok(()),
            _ => Err(HipError::Runtime(last_error())),
        }
    }

    /// (the last finished batch ran on the f64 kernel, fraction of its frames that needed f64) -- reporting only
    pub fn auto_state(&mut self) -> (bool, f64) {
        let (mut heavy, mut fraction) = (0 as c_int, 0.0f64);
        unsafe { melspec_auto_state(self.ctx, &mut heavy, &mut fraction) };
        (heavy != 0, fraction)
    }

    /// `CudaMelSpectrogram::max_frames_per_batch` (src/cuda.rs:84-86): frames per chunk of the host pipeline.
    pub fn max_frames_per_batch(&self) -> usize {
        unsafe { melspec_max_frames_per_batch(self.ctx) }
    }

    /// 0: f32 FFT + f64 recompute of the frames its error bound does not cover (default, within 1e-4 on every input),
    /// 1: f64 on every frame (mode 0 sends batches of speech / tonal input there by a vote inside their own launch), 2: f32 only.
    pub fn set_precision(&mut self, mode: i32) -> Result<(), HipError> {
        match unsafe { melspec_set_precision(self.ctx, mode as c_int) } {
            0 => Ok(()),
            _ => Err(HipError::Runtime(last_error())),
        }
    }

    /// Additive: many clips in one call through the chunked H2D / kernels / D2H pipeline; `Vec<Vec<Vec<f32>>>` = clip, frame, mel.
    pub fn compute_batch(&mut self, clips: &[&[f32]]) -> Result<Vec<Vec<Vec<f32>>>, HipError> {
        let lens: Vec<u64> = clips.iter().map(|c| c.len() as u64).collect();
        let mut offs = Vec::with_capacity(clips.len());
        let mut flat = Vec::with_capacity(lens.iter().sum::<u64>() as usize);
        for c in clips {
            offs.push(flat.len() as u64);
            flat.extend_from_slice(c);
        }
        let frames: Vec<usize> = clips.iter().map(|c| unsafe { melspec_num_frames(self.ctx, c.len()) }).collect();
        let mut out = vec![0.0f32; frames.iter().sum::<usize>() * self.n_mels];
        let mut total = 0u64;
        let rc = unsafe {
            melspec_compute_batch_host(self.ctx, flat.as_ptr(), offs.as_ptr(), lens.as_ptr(), clips.len() as u32, out.as_mut_ptr(),
                                       std::ptr::null(), out.len(), &mut total)
        };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        let mut cur = 0usize;
        Ok(frames.iter().map(|&f| {
            let rows = out[cur..cur + f * self.n_mels].chunks(self.n_mels).map(|r| r.to_vec()).collect();
            cur += f * self.n_mels;
            rows
        }).collect())
    }

    /// `Spectrogram::compute_all_cpu` (src/stft.rs:89-115): the full `fft_size`-bin complex spectrum of every frame, f64.
    pub fn compute_all(&mut self, samples: &[f32]) -> Result<Vec<Vec<num::Complex<f64>>>, HipError> {
        let frames = unsafe { melspec_num_frames(self.ctx, samples.len()) };
        let bins = unsafe { melspec_stft_bins(self.ctx, 1) };
        let mut flat = vec![num::Complex::<f64>::new(0.0, 0.0); frames * bins];
        let mut got = 0usize;
        let rc = unsafe { melspec_stft_host(self.ctx, samples.as_ptr(), samples.len(), flat.as_mut_ptr() as *mut c_void, flat.len(), 1, 1, &mut got) };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        Ok(flat.chunks(bins).take(got).map(|r| r.to_vec()).collect())
    }

    /// f64 window/FFT/power like `Spectrogram::compute_mel_spectrogram_cpu` and the CUDA backend's Z2Z FFT.
    /// `MelSpectrogram::add(&fft)` (src/mel.rs:13-32) for every frame of `frames` -- the reference's split API: the complex frames of
    /// `compute_all` / `Spectrogram::add` (n_fft bins each) -> `[frame][n_mels]`.
    pub fn mel_from_stft(&mut self, frames: &[Vec<num::Complex<f64>>]) -> Result<Vec<Vec<f32>>, HipError> {
        if frames.is_empty() {
            return Ok(Vec::new());
        }
        let bins = frames[0].len();
        let full = unsafe { melspec_stft_bins(self.ctx, 1) } == bins;
        let mut flat = Vec::with_capacity(frames.len() * bins * 2);
        for f in frames {
            for c in f {
                flat.push(c.re);
                flat.push(c.im);
            }
        }
        let mut out = vec![0.0f32; frames.len() * self.n_mels];
        let rc = unsafe {
            melspec_mel_from_stft_host(self.ctx, flat.as_ptr() as *const c_void, 1, full as c_int, frames.len(), out.as_mut_ptr(), out.len())
        };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        Ok(out.chunks(self.n_mels).map(|r| r.to_vec()).collect())
    }
    pub fn set_precise(&mut self, on: bool) -> Result<(), HipError> {
        match unsafe { melspec_set_precise(self.ctx, on as c_int) } {
            0 => Ok(()),
            _ => Err(HipError::Runtime(last_error())),
        }
    }

    /// Same contract as `CudaMelSpectrogram::compute_mel_spectrogram` / `Spectrogram::compute_mel_spectrogram_cpu`.
    pub fn compute_mel_spectrogram(&mut self, samples: &[f32]) -> Result<Vec<Vec<f32>>, HipError> {
        let frames = unsafe { melspec_num_frames(self.ctx, samples.len()) };
        if frames == 0 {
            return Ok(Vec::new());
        }
        let mut flat = vec![0.0f32; frames * self.n_mels];
        let mut got = 0usize;
        let rc = unsafe { melspec_compute_host(self.ctx, samples.as_ptr(), samples.len(), flat.as_mut_ptr(), flat.len(), &mut got) };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        Ok(flat.chunks(self.n_mels).take(got).map(|row| row.to_vec()).collect())
    }

    /// Additive: many equal-length clips already resident in HBM, one launch, asynchronous on `stream`.
    ///
    /// # Safety
    /// `d_pcm` / `d_out` must be device pointers valid for `n_clips * clip_stride` samples and
    /// `n_clips * frames * n_mels` floats.
    pub unsafe fn compute_uniform_device(&mut self, d_pcm: *const f32, clip_stride: u64, clip_len: u64, n_clips: u32,
                                         d_out: *mut f32, stream: *mut c_void) -> Result<(), HipError> {
        match melspec_compute_uniform_device(self.ctx, d_pcm, clip_stride, clip_len, n_clips, d_out, stream) {
            0 => Ok(()),
            _ => Err(HipError::Runtime(last_error())),
        }
    }

    pub fn synchronize(&mut self) -> Result<(), HipError> {
        match unsafe { melspec_synchronize(self.ctx, std::ptr::null_mut()) } {
            0 => Ok(()),
            _ => Err(HipError::Runtime(last_error())),
        }
    }
}

impl Drop for HipMelSpectrogram {
    fn drop(&mut self) {
        unsafe { melspec_destroy(self.ctx) }
    }
}

#[cfg(test)]
mod tests {
    use super::*;
    use crate::stft::Spectrogram;

    #[test]
    fn hip_matches_cpu_for_whisper_fft_400() {
        let sr = 16_000.0;
        let samples: Vec<f32> = (0..16_000)
            .map(|i| {
                let t = i as f32 / sr as f32;
                let w = 2.0 * std::f32::consts::PI * t;
                0.6 * (220.0 * w).sin() + 0.25 * (440.0 * w).sin() + 0.10 * (880.0 * w).sin() + 0.05 * (1760.0 * w).sin()
            })
            .collect();
        let cpu = Spectrogram::compute_mel_spectrogram_cpu(&samples, 400, 160, 80, sr);
        let mut hip = match HipMelSpectrogram::new(400, 160, sr, 80) {
            Ok(h) => h,
            Err(e) => {
                eprintln!("Skipping hip test: {e}");
                return;
            }
        };
        let gpu = hip.compute_mel_spectrogram(&samples).expect("hip mel spectrogram");
        assert_eq!(cpu.len(), gpu.len());
        let max = cpu.iter().flatten().zip(gpu.iter().flatten()).map(|(a, b)| (a - b).abs()).fold(0.0f32, f32::max);
        assert!(max <= 1e-4, "max delta {max}");
    }
}

/// `quant::tga_8bit` on the device: one TGA per <= 65535-column chunk of a major-row-order image.
pub struct HipTga {
    q: *mut Tga,
}
impl HipTga {
    pub fn new() -> Result<Self, HipError> {
        let mut q = std::ptr::null_mut();
        let rc = unsafe { melspec_tga_create(&mut q, -1) };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { q })
    }
    pub fn tga_8bit(&mut self, data: &[f32], n_mels: usize) -> Result<Vec<Vec<u8>>, HipError> {
        let (mut n, mut stride, mut last) = (0u32, 0usize, 0usize);
        if unsafe { melspec_tga_layout(n_mels as c_int, data.len() / n_mels, &mut n, &mut stride, &mut last) } != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        if n == 0 {
            return Ok(vec![]);
        }
        let mut flat = vec![0u8; stride * (n as usize - 1) + last];
        let mut got = 0u32;
        if unsafe { melspec_tga_encode_host(self.q, data.as_ptr(), data.len(), n_mels as c_int, flat.as_mut_ptr(), flat.len(), &mut got) } != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        let full = 26 + n_mels * 65535;
        Ok((0..n as usize).map(|c| flat[c * stride..c * stride + if c + 1 < n as usize { full } else { last }].to_vec()).collect())
    }
    pub fn parse_tga_8bit(&mut self, blob: &[u8]) -> Result<Vec<f32>, HipError> {
        let mut out = vec![0f32; blob.len().saturating_sub(26)];
        let mut n = 0usize;
        if unsafe { melspec_tga_decode_host(self.q, blob.as_ptr(), blob.len(), out.as_mut_ptr(), out.len(), &mut n) } != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        out.truncate(n);
        Ok(out)
    }
}
impl Drop for HipTga {
    fn drop(&mut self) {
        unsafe { melspec_tga_destroy(self.q) }
    }
}

/// One live stream (`RingBuffer::add_frame` + `maybe_mel` loop, src/rb.rs:60-121) over a 1-stream bank.
pub struct HipStream<'a> {
    st: *mut Stream,
    n_mels: usize,
    _mel: std::marker::PhantomData<&'a mut HipMelSpectrogram>,
}
impl<'a> HipStream<'a> {
    pub fn new(mel: &'a mut HipMelSpectrogram, max_chunk: usize) -> Result<Self, HipError> {
        let mut st = std::ptr::null_mut();
        let rc = unsafe { melspec_stream_create(&mut st, mel.ctx, 1, max_chunk as u32) };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { st, n_mels: mel.n_mels, _mel: std::marker::PhantomData })
    }
    /// Feeds a block of any length and returns the frames it completes, one `Vec<f32>` of n_mels each.
    pub fn add_frame(&mut self, samples: &[f32]) -> Result<Vec<Vec<f32>>, HipError> {
        let (id, len) = (0u32, samples.len() as u32);
        let cap = unsafe { melspec_stream_frames_after(self.st, 0, len) } * self.n_mels;
        let mut flat = vec![0f32; cap];
        let mut frames = 0u32;
        if unsafe { melspec_stream_push_host(self.st, &id, samples.as_ptr(), &len, 1, flat.as_mut_ptr(), cap, &mut frames) } != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        Ok(flat.chunks(self.n_mels).take(frames as usize).map(|c| c.to_vec()).collect())
    }
}
impl HipStream<'_> {
    /// Turns the detector stage on: from now on `add_frame_activity` also returns what
    /// `VoiceActivityDetector::add_activity` (src/vad.rs:155-205) gives for every frame, computed on the device from the rows
    /// the push has just written.
    pub fn enable_vad(&mut self, settings: &crate::vad::DetectionSettings) -> Result<(), HipError> {
        let s = VadSettingsC { min_energy: settings.min_energy, min_y: settings.min_y as c_int, min_x: settings.min_x as c_int, min_mel: settings.min_mel as c_int };
        match unsafe { melspec_stream_enable_vad(self.st, &s) } {
            0 => Ok(()),
            _ => Err(HipError::Runtime(last_error())),
        }
    }
    /// `add_frame` + one `Option<VoiceActivity>` per completed frame (`None` until min_x frames have been seen).
    pub fn add_frame_activity(&mut self, samples: &[f32]) -> Result<(Vec<Vec<f32>>, Vec<Option<crate::vad::VoiceActivity>>), HipError> {
        let (id, len) = (0u32, samples.len() as u32);
        let frames_cap = unsafe { melspec_stream_frames_after(self.st, 0, len) };
        let first = unsafe { melspec_stream_vad_frames(self.st, 0) } as usize;
        let mut flat = vec![0f32; frames_cap * self.n_mels];
        let mut acts = vec![VadActivityC::default(); frames_cap];
        let mut frames = 0u32;
        if unsafe { melspec_stream_push_host_vad(self.st, &id, samples.as_ptr(), &len, 1, flat.as_mut_ptr(), flat.len(), &mut frames,
                                                  acts.as_mut_ptr(), acts.len()) } != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        let rows = flat.chunks(self.n_mels).take(frames as usize).map(|c| c.to_vec()).collect();
        let acts = acts.iter().take(frames as usize).enumerate().map(|(k, a)| {
            if a.valid == 0 {
                return None;
            }
            let (n, w) = (a.active_columns as usize, a.window_columns as usize);
            Some(crate::vad::VoiceActivity {
                active: a.active != 0,
                frame_index: first + k,
                leading_active_columns: a.leading_active_columns as usize,
                active_columns: n,
                window_columns: w,
                confidence: if w == 0 { 0.0 } else { n as f64 / w as f64 },
                timestamps: None,
            })
        }).collect();
        Ok((rows, acts))
    }
}
impl Drop for HipStream<'_> {
    fn drop(&mut self) {
        unsafe { melspec_stream_destroy(self.st) }
    }
}

/// `Fbank` (src/fbank.rs:85-247) on the GPU: same constructor argument, same `compute` signature and result type.
pub struct HipFbank {
    fb: *mut FbankHandle,
    num_mel_bins: usize,
}
impl HipFbank {
    pub fn new(config: crate::fbank::FbankConfig) -> Result<Self, HipError> {
        let c = FbankConfigC {
            sample_rate: config.sample_rate,
            num_mel_bins: config.num_mel_bins as i32,
            frame_length_ms: config.frame_length_ms,
            frame_shift_ms: config.frame_shift_ms,
            energy_floor: config.energy_floor,
            use_log_fbank: config.use_log_fbank as i32,
            use_power: config.use_power as i32,
            preemphasis: config.preemphasis,
            apply_cmn: config.apply_cmn as i32,
            low_freq: config.low_freq,
            high_freq: config.high_freq,
        };
        let mut fb = std::ptr::null_mut();
        let rc = unsafe { melspec_fbank_create(&mut fb, -1, &c) };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { fb, num_mel_bins: config.num_mel_bins })
    }
    /// `Fbank::compute(&self, samples) -> Array2<f32>` (frames, num_mel_bins), src/fbank.rs:141-236
    pub fn compute(&mut self, samples: &[f32]) -> Result<ndarray::Array2<f32>, HipError> {
        let frames = unsafe { melspec_fbank_num_frames(self.fb, samples.len()) };
        let mut flat = vec![0.0f32; frames * self.num_mel_bins];
        let mut got = 0usize;
        let rc = unsafe { melspec_fbank_compute_host(self.fb, samples.as_ptr(), samples.len(), flat.as_mut_ptr(), flat.len(), &mut got) };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        Ok(ndarray::Array2::from_shape_vec((got, self.num_mel_bins), flat).expect("shape"))
    }
    /// Additive: `Fbank::compute` for many clips in one call (one ragged launch per ~16 MiB of PCM through the pinned host pipeline).
    pub fn compute_batch(&mut self, clips: &[&[f32]]) -> Result<Vec<ndarray::Array2<f32>>, HipError> {
        let lengths: Vec<u64> = clips.iter().map(|c| c.len() as u64).collect();
        let mut offsets = Vec::with_capacity(clips.len());
        let mut flat_in = Vec::with_capacity(lengths.iter().sum::<u64>() as usize);
        for c in clips {
            offsets.push(flat_in.len() as u64);
            flat_in.extend_from_slice(c);
        }
        let frames: Vec<usize> = clips.iter().map(|c| unsafe { melspec_fbank_num_frames(self.fb, c.len()) }).collect();
        let mut flat = vec![0.0f32; frames.iter().sum::<usize>() * self.num_mel_bins];
        let mut total = 0u64;
        let rc = unsafe {
            melspec_fbank_compute_batch_host(self.fb, flat_in.as_ptr(), offsets.as_ptr(), lengths.as_ptr(), clips.len() as u32,
                                             flat.as_mut_ptr(), std::ptr::null(), flat.len(), &mut total)
        };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        let mut out = Vec::with_capacity(clips.len());
        let mut cur = 0usize;
        for f in frames {
            let n = f * self.num_mel_bins;
            out.push(ndarray::Array2::from_shape_vec((f, self.num_mel_bins), flat[cur..cur + n].to_vec()).expect("shape"));
            cur += n;
        }
        Ok(out)
    }
}
impl Drop for HipFbank {
    fn drop(&mut self) {
        unsafe { melspec_fbank_destroy(self.fb) }
    }
}

/// `BatchLogMelSpectrogram` (src/mel.rs:239-396) on the GPU.  Invalid configs come back as
/// `BatchLogMelError::InvalidConfig` with the reference's messages (validate_batch_config, src/mel.rs:656-683).
pub struct HipBatchLogMel {
    b: *mut BlmHandle,
    n_mels: usize,
}
impl HipBatchLogMel {
    pub fn new(config: crate::mel::BatchLogMelConfig) -> Result<Self, crate::mel::BatchLogMelError> {
        let c = BlmConfigC {
            sample_rate: config.sample_rate as i32,
            n_fft: config.n_fft as i32,
            win_length: config.win_length as i32,
            hop_length: config.hop_length as i32,
            n_mels: config.n_mels as i32,
            f_min: config.f_min,
            f_max: config.f_max.unwrap_or(-1.0),
            htk: config.htk as i32,
            norm: config.norm as i32,
            preemphasis: config.preemphasis,
            center: config.center as i32,
            log_zero_guard: config.log_zero_guard,
            pad_to: config.pad_to as i32,
            normalize_per_feature: config.normalize_per_feature as i32,
        };
        let mut b = std::ptr::null_mut();
        if unsafe { melspec_blm_create(&mut b, -1, &c) } != 0 {
            return Err(crate::mel::BatchLogMelError::InvalidConfig(last_error()));
        }
        Ok(Self { b, n_mels: config.n_mels })
    }
    /// Additive: `true` = the reference's own f32 arithmetic for this frontend (src/mel.rs:251-252,356-357) on the f32 kernel; `false`
    /// (default) = f64 up to |X|^2.  `Ok(true)`: the next call runs the f32 kernel; `Ok(false)`: the mode was accepted but this context has
    /// no f32 kernel (another bank or geometry): it keeps computing in f64; `Err`: the library refused the call.
    pub fn set_f32(&mut self, on: bool) -> Result<bool, HipError> {
        unsafe {
            check(melspec_blm_set_precision(self.b, if on { 2 } else { 0 }))?;
            Ok(melspec_blm_precision(self.b) == 2)
        }
    }
    /// `compute(&self, samples) -> Array2<f32>` (n_mels, padded frames), src/mel.rs:299-302; the second value is
    /// `BatchLogMelOutput::valid_frames` (src/mel.rs:387-395).
    pub fn compute(&mut self, samples: &[f32]) -> Result<(ndarray::Array2<f32>, usize), HipError> {
        let cols = unsafe { melspec_blm_padded_frames(self.b, samples.len()) };
        let valid = unsafe { melspec_blm_num_frames(self.b, samples.len()) };
        let mut flat = vec![0.0f32; self.n_mels * cols];
        let (mut rows, mut got_cols) = (0usize, 0usize);
        let rc = unsafe { melspec_blm_compute_host(self.b, samples.as_ptr(), samples.len(), flat.as_mut_ptr(), flat.len(), &mut rows, &mut got_cols) };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        Ok((ndarray::Array2::from_shape_vec((self.n_mels, got_cols), flat).expect("shape"), valid))
    }
    /// Additive: `compute` for many clips in one call; every clip comes back as its `(n_mels, cols)` array and valid frame count.
    pub fn compute_batch(&mut self, clips: &[&[f32]]) -> Result<Vec<(ndarray::Array2<f32>, usize)>, HipError> {
        let lengths: Vec<u64> = clips.iter().map(|c| c.len() as u64).collect();
        let mut offsets = Vec::with_capacity(clips.len());
        let mut flat_in = Vec::with_capacity(lengths.iter().sum::<u64>() as usize);
        for c in clips {
            offsets.push(flat_in.len() as u64);
            flat_in.extend_from_slice(c);
        }
        let cols: Vec<usize> = clips.iter().map(|c| unsafe { melspec_blm_padded_frames(self.b, c.len()) }).collect();
        let mut flat = vec![0.0f32; cols.iter().sum::<usize>() * self.n_mels];
        let mut total = 0u64;
        let rc = unsafe {
            melspec_blm_compute_batch_host(self.b, flat_in.as_ptr(), offsets.as_ptr(), lengths.as_ptr(), clips.len() as u32,
                                           flat.as_mut_ptr(), std::ptr::null(), flat.len(), &mut total)
        };
        if rc != 0 {
            return Err(HipError::Runtime(last_error()));
        }
        let mut out = Vec::with_capacity(clips.len());
        let mut cur = 0usize;
        for (c, clip) in cols.into_iter().zip(clips) {
            let n = c * self.n_mels;
            let valid = unsafe { melspec_blm_num_frames(self.b, clip.len()) };
            out.push((ndarray::Array2::from_shape_vec((self.n_mels, c), flat[cur..cur + n].to_vec()).expect("shape"), valid));
            cur += n;
        }
        Ok(out)
    }
}
impl Drop for HipBatchLogMel {
    fn drop(&mut self) {
        unsafe { melspec_blm_destroy(self.b) }
    }
}

/// `vad_boundaries(frames, settings) -> EdgeInfo` (src/vad.rs:251-340) with the Sobel stencil and the majority vote on the GPU.
/// `frames`: the mel images of `interleave_frames(.., false, ..)` / `to_array2`, concatenated along time like the reference does.
pub fn hip_vad_boundaries(frames: &[ndarray::Array2<f64>], settings: &crate::vad::DetectionSettings) -> Result<crate::vad::EdgeInfo, HipError> {
    use std::collections::HashSet;
    let Some(first) = frames.first() else {
        return Ok(crate::vad::EdgeInfo::new(Vec::new(), Vec::new(), HashSet::new()));
    };
    let height = first.nrows();
    let width: usize = frames.iter().map(|f| f.ncols()).sum();
    let mut image = vec![0.0f32; height * width];
    let mut x0 = 0usize;
    for f in frames {
        for y in 0..height {
            for x in 0..f.ncols() {
                image[y * width + x0 + x] = f[[y, x]] as f32;
            }
        }
        x0 += f.ncols();
    }
    let n = unsafe { melspec_vad_mask_len(height as c_int, width) };
    let (mut raw, mut smoothed, mut run) = (vec![0u8; n.max(1)], vec![0u8; n.max(1)], 0u32);
    let s = VadSettingsC { min_energy: settings.min_energy, min_y: settings.min_y as c_int, min_x: settings.min_x as c_int, min_mel: settings.min_mel as c_int };
    let rc = unsafe { melspec_vad_boundaries_host(-1, image.as_ptr(), height as c_int, width, &s, raw.as_mut_ptr(), smoothed.as_mut_ptr(), &mut run) };
    if rc != 0 {
        return Err(HipError::Runtime(last_error()));
    }
    let (mut hit, mut miss) = (Vec::new(), Vec::new());
    for (x, &m) in smoothed.iter().take(n).enumerate() {
        if m != 0 { hit.push(x) } else { miss.push(x) }
    }
    // gradient_positions is only read by as_image (src/vad.rs:523-529), which is outside the accelerated path
    Ok(crate::vad::EdgeInfo::new(miss, hit, HashSet::new()))
}

// ---------------------------------------------------------------------------------------------------------------------------
// The additive API (round 4): device-resident buffers, ragged device batches, the GPUs of one node, the stand-alone filterbank,
// pinned host buffers and PCM -> TGA on the device.  Shaped like src/cuda.rs:27-101: an owning struct per handle, `Drop` frees it,
// every call returns `Result<_, HipError>`.  tests/test_rust_shim.py checks every declaration below against include/melspec_hip.h
// (argument count and the C <-> Rust type of every argument and return value), since this image cannot compile the file.
// ---------------------------------------------------------------------------------------------------------------------------
#[repr(C)]
struct Sharded {
    _private: [u8; 0],
}
#[repr(C)]
struct Bank {
    _private: [u8; 0],
}

unsafe extern "C" {
    fn melspec_device_count() -> c_int;
    fn melspec_malloc(dptr: *mut *mut c_void, bytes: usize) -> c_int;
    fn melspec_free(dptr: *mut c_void) -> c_int;
    fn melspec_memcpy_h2d(dst_device: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    fn melspec_memcpy_d2h(dst_host: *mut c_void, src_device: *const c_void, bytes: usize) -> c_int;
    fn melspec_device_synchronize() -> c_int;
    fn melspec_host_alloc(p: *mut *mut c_void, bytes: usize) -> c_int;
    fn melspec_host_free(p: *mut c_void) -> c_int;
    fn melspec_compute_ragged_device(ctx: *mut Ctx, d_pcm: *const f32, h_offsets: *const u64, h_lengths: *const u64, n_clips: u32,
                                     d_out: *mut f32, h_out_offsets: *const u64, stream: *mut c_void) -> c_int;
    fn melspec_compute_ragged_device_desc(ctx: *mut Ctx, d_pcm: *const f32, d_offsets: *const u64, d_lengths: *const u64, n_clips: u32,
                                          d_out: *mut f32, d_out_offsets: *const u64, max_total_frames: u64, stream: *mut c_void) -> c_int;
    fn melspec_interleaved_width(ctx: *const Ctx, n_samples: usize, min_width: usize) -> usize;
    fn melspec_compute_uniform_device_interleaved(ctx: *mut Ctx, d_pcm: *const f32, clip_stride: u64, clip_len: u64, n_clips: u32,
                                                  d_out: *mut f32, major_column_order: c_int, min_width: u64, stream: *mut c_void) -> c_int;
    // the GPUs of one node (per-clip split, no collective)
    fn melspec_shard_by_samples(lengths: *const u64, n_clips: u32, n_shards: c_int, bounds: *mut u32) -> c_int;
    fn melspec_sharded_create(out: *mut *mut Sharded, devices: *const c_int, n_devices: c_int, fft: c_int, hop: c_int, sr: f64,
                              n_mels: c_int) -> c_int;
    fn melspec_sharded_destroy(s: *mut Sharded);
    fn melspec_sharded_n_shards(s: *const Sharded) -> c_int;
    fn melspec_sharded_ctx(s: *mut Sharded, shard: c_int) -> *mut Ctx;
    fn melspec_sharded_compute_batch_host(s: *mut Sharded, samples: *const f32, offsets: *const u64, lengths: *const u64, n_clips: u32,
                                          out: *mut f32, out_offsets: *const u64, cap: usize, total_frames: *mut u64) -> c_int;
    fn melspec_sharded_compute_uniform_device(s: *mut Sharded, d_pcm: *const *const f32, clip_stride: u64, clip_len: u64,
                                              n_clips: *const u32, d_out: *const *mut f32) -> c_int;
    fn melspec_sharded_compute_ragged_device(s: *mut Sharded, d_pcm: *const *const f32, h_offsets: *const u64, h_lengths: *const u64,
                                             n_clips: *const u32, d_out: *const *mut f32, h_out_offsets: *const u64) -> c_int;
    fn melspec_sharded_synchronize(s: *mut Sharded) -> c_int;
    fn melspec_gather_peer(dst_device: c_int, dst: *mut c_void, src_devices: *const c_int, srcs: *const *const c_void,
                           bytes: *const usize, dst_offsets: *const usize, n: c_int) -> c_int;
    // SparseMelFilterbank / log_mel_spectrogram / norm_mel (src/mel.rs:40-168, 436-469)
    fn melspec_bank_from_dense(out: *mut *mut Bank, device: c_int, filters: *const f64, n_mels: c_int, fft_bins: c_int) -> c_int;
    fn melspec_bank_from_mel(out: *mut *mut Bank, device: c_int, sample_rate: f64, n_fft: c_int, n_mels: c_int, f_min: f64, f_max: f64,
                             htk: c_int, norm: c_int) -> c_int;
    fn melspec_bank_destroy(bank: *mut Bank);
    fn melspec_bank_n_mels(bank: *const Bank) -> c_int;
    fn melspec_bank_fft_bins(bank: *const Bank) -> c_int;
    fn melspec_bank_non_zero_weights(bank: *const Bank) -> c_int;
    fn melspec_bank_weights_for_mel(bank: *const Bank, mel_idx: c_int, bins: *mut c_int, weights: *mut f64, capacity: c_int) -> c_int;
    fn melspec_bank_project_power_host(bank: *mut Bank, power: *const c_void, dtype: c_int, n_frames: usize, out: *mut c_void) -> c_int;
    fn melspec_bank_project_power_device(bank: *mut Bank, d_power: *const c_void, dtype: c_int, n_frames: u64, d_out: *mut c_void,
                                         stream: *mut c_void) -> c_int;
    fn melspec_bank_log_mel_host(bank: *mut Bank, stft: *const c_void, dtype: c_int, n_fft: c_int, n_frames: usize, out: *mut f64) -> c_int;
    fn melspec_bank_log_mel_device(bank: *mut Bank, d_stft: *const c_void, dtype: c_int, n_fft: c_int, n_frames: u64, d_out: *mut f64,
                                   stream: *mut c_void) -> c_int;
    fn melspec_bank_norm_mel_host(bank: *mut Bank, input: *const c_void, dtype: c_int, n_values: usize, out: *mut c_void) -> c_int;
    fn melspec_bank_norm_mel_device(bank: *mut Bank, d_in: *const c_void, dtype: c_int, n_values: u64, d_out: *mut c_void,
                                    stream: *mut c_void) -> c_int;
    // PCM -> TGA blobs without leaving the device
    fn melspec_tga_encode_pcm_uniform_device(q: *mut Tga, ctx: *mut Ctx, d_pcm: *const f32, clip_stride: u64, clip_len: u64, n_clips: u32,
                                             min_width: u64, d_images: *mut f32, d_blobs: *mut u8, blob_stride: usize,
                                             stream: *mut c_void) -> c_int;
    fn melspec_tga_synchronize(q: *mut Tga) -> c_int;
}

const STFT_F32: c_int = 0; // MELSPEC_STFT_F32
const STFT_F64: c_int = 1; // MELSPEC_STFT_F64

fn check(rc: c_int) -> Result<(), HipError> {
    if rc == 0 { Ok(()) } else { Err(HipError::Runtime(last_error())) }
}

/// gfx950 devices the library can use (0 when there is none: callers skip like src/cuda.rs:512-518).
pub fn hip_device_count() -> usize {
    unsafe { melspec_device_count() }.max(0) as usize
}

/// `n` elements of `T` in HBM on the current device (what the cudaMalloc / cudaMemcpy externs of src/cuda.rs:185-199 give the CUDA side).
pub struct HipDeviceBuffer<T: Copy> {
    ptr: *mut c_void,
    len: usize,
    _t: std::marker::PhantomData<T>,
}
impl<T: Copy> HipDeviceBuffer<T> {
    pub fn new(len: usize) -> Result<Self, HipError> {
        let mut ptr = std::ptr::null_mut();
        check(unsafe { melspec_malloc(&mut ptr, len.max(1) * std::mem::size_of::<T>()) })?;
        Ok(Self { ptr, len, _t: std::marker::PhantomData })
    }
    pub fn from_slice(host: &[T]) -> Result<Self, HipError> {
        let b = Self::new(host.len())?;
        check(unsafe { melspec_memcpy_h2d(b.ptr, host.as_ptr() as *const c_void, std::mem::size_of_val(host)) })?;
        Ok(b)
    }
    pub fn to_vec(&self) -> Result<Vec<T>, HipError> {
        let mut v = Vec::<T>::with_capacity(self.len);
        check(unsafe { melspec_memcpy_d2h(v.as_mut_ptr() as *mut c_void, self.ptr, self.len * std::mem::size_of::<T>()) })?;
        unsafe { v.set_len(self.len) };
        Ok(v)
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn as_ptr(&self) -> *const T { self.ptr as *const T }
    pub fn as_mut_ptr(&mut self) -> *mut T { self.ptr as *mut T }
}
impl<T: Copy> Drop for HipDeviceBuffer<T> {
    fn drop(&mut self) {
        unsafe { melspec_free(self.ptr) };
    }
}

/// Pinned host samples (`cudaMallocHost`, src/cuda.rs:185-199): the host calls copy straight out of / into such a buffer.
pub struct HipPinnedBuffer {
    ptr: *mut f32,
    len: usize,
}
impl HipPinnedBuffer {
    pub fn new(len: usize) -> Result<Self, HipError> {
        let mut p = std::ptr::null_mut();
        check(unsafe { melspec_host_alloc(&mut p, len.max(1) * 4) })?;
        unsafe { std::ptr::write_bytes(p as *mut u8, 0, len * 4) };
        Ok(Self { ptr: p as *mut f32, len })
    }
    pub fn as_slice(&self) -> &[f32] { unsafe { std::slice::from_raw_parts(self.ptr, self.len) } }
    pub fn as_mut_slice(&mut self) -> &mut [f32] { unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) } }
}
impl Drop for HipPinnedBuffer {
    fn drop(&mut self) {
        unsafe { melspec_host_free(self.ptr as *mut c_void) };
    }
}

impl HipMelSpectrogram {
    /// Frames of a clip of `n_samples` samples (`(n - fft) / hop + 1`, src/stft.rs:147-169).
    pub fn num_frames(&self, n_samples: usize) -> usize {
        unsafe { melspec_num_frames(self.ctx, n_samples) }
    }

    /// Many clips of any length, PCM and frames resident in HBM: one launch at the kernel rate (3.4 G frames/s on an MI355X against
    /// 50-65 M through the PCIe-bound host calls).  Clip `c` = `pcm[offsets[c] .. + lengths[c]]`; its frames are packed back to back
    /// in clip order.  Returns the frames of every clip; asynchronous until `synchronize`.
    pub fn compute_ragged_device(&mut self, pcm: &HipDeviceBuffer<f32>, offsets: &[u64], lengths: &[u64], out: &mut HipDeviceBuffer<f32>)
                                 -> Result<Vec<usize>, HipError> {
        assert_eq!(offsets.len(), lengths.len());
        let frames: Vec<usize> = lengths.iter().map(|&n| self.num_frames(n as usize)).collect();
        for (o, n) in offsets.iter().zip(lengths) {
            // checked: a wrapping `o + n` would let a crafted pair through and drive device reads out of bounds from safe code
            match o.checked_add(*n) {
                Some(end) if end <= pcm.len() as u64 => {}
                _ => return Err(HipError::Runtime("clip outside the PCM buffer".into())),
            }
        }
        let need = frames.iter().try_fold(0usize, |a, &f| a.checked_add(f)).and_then(|f| f.checked_mul(self.n_mels));
        if need.map_or(true, |n| n > out.len()) {
            return Err(HipError::Runtime("output buffer too small".into()));
        }
        check(unsafe {
            melspec_compute_ragged_device(self.ctx, pcm.as_ptr(), offsets.as_ptr(), lengths.as_ptr(), offsets.len() as u32, out.as_mut_ptr(),
                                          std::ptr::null(), std::ptr::null_mut())
        })?;
        Ok(frames)
    }

    /// The same with the clip table itself in device memory (a segmenter or VAD on the GPU wrote it): nothing is copied back.
    ///
    /// # Safety
    /// `d_offsets` / `d_lengths` must hold `n_clips` entries on the device and describe clips inside `pcm`; `max_total_frames` must
    /// bound the frames of all clips together and fit `out`.
    pub unsafe fn compute_ragged_device_desc(&mut self, pcm: &HipDeviceBuffer<f32>, d_offsets: *const u64, d_lengths: *const u64, n_clips: u32,
                                             out: &mut HipDeviceBuffer<f32>, max_total_frames: u64) -> Result<(), HipError> {
        check(melspec_compute_ragged_device_desc(self.ctx, pcm.as_ptr(), d_offsets, d_lengths, n_clips, out.as_mut_ptr(), std::ptr::null(),
                                                 max_total_frames, std::ptr::null_mut()))
    }

    /// `interleave_frames(frames, major_column_order, min_width)` (src/mel.rs:480-544) fused into the store, uniform clips in HBM:
    /// per clip `n_mels x W` floats, `W = interleaved_width(clip_len, min_width)`.
    pub fn compute_uniform_device_interleaved(&mut self, pcm: &HipDeviceBuffer<f32>, clip_len: usize, n_clips: usize, major_column_order: bool,
                                              min_width: usize, out: &mut HipDeviceBuffer<f32>) -> Result<usize, HipError> {
        let w = unsafe { melspec_interleaved_width(self.ctx, clip_len, min_width) };
        let fits = |a: usize, b: usize, c: usize, cap: usize| a.checked_mul(b).and_then(|x| x.checked_mul(c)).map_or(false, |x| x <= cap);
        if !fits(n_clips, clip_len, 1, pcm.len()) || !fits(n_clips, w, self.n_mels, out.len()) {
            return Err(HipError::Runtime("buffer too small".into()));
        }
        check(unsafe {
            melspec_compute_uniform_device_interleaved(self.ctx, pcm.as_ptr(), clip_len as u64, clip_len as u64, n_clips as u32, out.as_mut_ptr(),
                                                       major_column_order as c_int, min_width as u64, std::ptr::null_mut())
        })?;
        Ok(w)
    }
}

/// Contiguous blocks of clips per shard, balanced by samples: `bounds[k] .. bounds[k + 1]` are the clips of shard `k`.
pub fn hip_shard_by_samples(lengths: &[u64], n_shards: usize) -> Result<Vec<u32>, HipError> {
    let mut bounds = vec![0u32; n_shards + 1];
    check(unsafe { melspec_shard_by_samples(lengths.as_ptr(), lengths.len() as u32, n_shards as c_int, bounds.as_mut_ptr()) })?;
    Ok(bounds)
}

/// One `HipMelSpectrogram` per GPU of the node; clips are split per device, no data crosses a device boundary (the reference binds
/// a single device, src/cuda.rs:246-247).
pub struct HipShardedMelSpectrogram {
    s: *mut Sharded,
    n_mels: usize,
    fft_size: usize,
    hop_size: usize,
}
impl HipShardedMelSpectrogram {
    /// `devices`: `None` = every gfx950 device of the node.
    pub fn new(devices: Option<&[i32]>, fft_size: usize, hop_size: usize, sampling_rate: f64, n_mels: usize) -> Result<Self, HipError> {
        let mut s = std::ptr::null_mut();
        let (p, n) = match devices {
            Some(d) => (d.as_ptr(), d.len() as c_int),
            None => (std::ptr::null(), 0),
        };
        let rc = unsafe { melspec_sharded_create(&mut s, p, n, fft_size as c_int, hop_size as c_int, sampling_rate, n_mels as c_int) };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { s, n_mels, fft_size, hop_size })
    }
    pub fn n_shards(&self) -> usize {
        unsafe { melspec_sharded_n_shards(self.s) }.max(0) as usize
    }
    fn frames(&self, n: usize) -> usize {
        if n < self.fft_size { 0 } else { (n - self.fft_size) / self.hop_size + 1 }
    }
    /// `compute_batch` over all devices: one host thread per device drives that device's pipeline on its block of clips.
    pub fn compute_batch(&mut self, clips: &[&[f32]]) -> Result<Vec<Vec<Vec<f32>>>, HipError> {
        let lens: Vec<u64> = clips.iter().map(|c| c.len() as u64).collect();
        let mut offs = Vec::with_capacity(clips.len());
        let mut flat = Vec::with_capacity(lens.iter().sum::<u64>() as usize);
        for c in clips {
            offs.push(flat.len() as u64);
            flat.extend_from_slice(c);
        }
        let frames: Vec<usize> = clips.iter().map(|c| self.frames(c.len())).collect();
        let mut out = vec![0.0f32; frames.iter().sum::<usize>() * self.n_mels];
        let mut total = 0u64;
        check(unsafe {
            melspec_sharded_compute_batch_host(self.s, flat.as_ptr(), offs.as_ptr(), lens.as_ptr(), clips.len() as u32, out.as_mut_ptr(),
                                               std::ptr::null(), out.len(), &mut total)
        })?;
        let mut cur = 0usize;
        Ok(frames.iter().map(|&f| {
            let rows = out[cur..cur + f * self.n_mels].chunks(self.n_mels).map(|r| r.to_vec()).collect();
            cur += f * self.n_mels;
            rows
        }).collect())
    }
    /// Device-resident shards, equal-length clips: shard `k`'s `n_clips[k]` clips are on device `k` at `d_pcm[k]`, its frames stay
    /// there at `d_out[k]`.  Stream-ordered per shard; `synchronize` waits for all of them.
    ///
    /// # Safety
    /// Every `d_pcm[k]` / `d_out[k]` must be a pointer on shard `k`'s device, valid for that shard's clips / frames.
    pub unsafe fn compute_uniform_device(&mut self, d_pcm: &[*const f32], clip_stride: u64, clip_len: u64, n_clips: &[u32], d_out: &[*mut f32])
                                         -> Result<(), HipError> {
        let n = self.n_shards();
        if d_pcm.len() != n || n_clips.len() != n || d_out.len() != n {
            return Err(HipError::Runtime("one pointer and one clip count per shard".into()));
        }
        check(melspec_sharded_compute_uniform_device(self.s, d_pcm.as_ptr(), clip_stride, clip_len, n_clips.as_ptr(), d_out.as_ptr()))
    }
    /// The ragged form: the clip tables of the shards one after the other (`n_clips[0]` entries, then `n_clips[1]`, ...), offsets
    /// relative to the shard's own `d_pcm[k]`; frames packed per shard.
    ///
    /// # Safety
    /// As `compute_uniform_device`.
    pub unsafe fn compute_ragged_device(&mut self, d_pcm: &[*const f32], offsets: &[u64], lengths: &[u64], n_clips: &[u32], d_out: &[*mut f32])
                                        -> Result<(), HipError> {
        let n = self.n_shards();
        if d_pcm.len() != n || n_clips.len() != n || d_out.len() != n || offsets.len() != lengths.len()
            || n_clips.iter().map(|&c| c as usize).sum::<usize>() != offsets.len() {
            return Err(HipError::Runtime("one pointer and one clip count per shard, one table entry per clip".into()));
        }
        check(melspec_sharded_compute_ragged_device(self.s, d_pcm.as_ptr(), offsets.as_ptr(), lengths.as_ptr(), n_clips.as_ptr(), d_out.as_ptr(),
                                                    std::ptr::null()))
    }
    pub fn synchronize(&mut self) -> Result<(), HipError> {
        check(unsafe { melspec_sharded_synchronize(self.s) })
    }
    /// Precision mode of every shard (`HipMelSpectrogram::set_precision`).
    pub fn set_precision(&mut self, mode: i32) -> Result<(), HipError> {
        for k in 0..self.n_shards() {
            check(unsafe { melspec_set_precision(melspec_sharded_ctx(self.s, k as c_int), mode as c_int) })?;
        }
        Ok(())
    }
}
impl Drop for HipShardedMelSpectrogram {
    fn drop(&mut self) {
        unsafe { melspec_sharded_destroy(self.s) }
    }
}

/// Optional consolidation of device-resident shard results on one device (each piece crosses its own xGMI link).
///
/// # Safety
/// `srcs[i]` must be valid for `bytes[i]` bytes on `src_devices[i]`, `dst` for every `dst_offsets[i] + bytes[i]` on `dst_device`.
pub unsafe fn hip_gather_peer(dst_device: i32, dst: *mut c_void, src_devices: &[i32], srcs: &[*const c_void], bytes: &[usize],
                              dst_offsets: &[usize]) -> Result<(), HipError> {
    check(melspec_gather_peer(dst_device as c_int, dst, src_devices.as_ptr(), srcs.as_ptr(), bytes.as_ptr(), dst_offsets.as_ptr(),
                              srcs.len() as c_int))
}

/// `SparseMelFilterbank` (src/mel.rs:40-168) with the projections on the GPU; the sums are the reference's left folds over its
/// sparse rows, so `project_power_*` is bit-exact against the reference's f32 / f64 arithmetic.
pub struct HipSparseMelFilterbank {
    b: *mut Bank,
}
impl HipSparseMelFilterbank {
    /// `SparseMelFilterbank::from_dense` (src/mel.rs:48-71)
    pub fn from_dense(filters: &ndarray::Array2<f64>) -> Result<Self, HipError> {
        let f = filters.as_standard_layout();
        let mut b = std::ptr::null_mut();
        let rc = unsafe { melspec_bank_from_dense(&mut b, -1, f.as_ptr(), f.nrows() as c_int, f.ncols() as c_int) };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { b })
    }
    /// `SparseMelFilterbank::from_mel` (src/mel.rs:73-87)
    pub fn from_mel(sample_rate: f64, n_fft: usize, n_mels: usize, f_min: Option<f64>, f_max: Option<f64>, htk: bool, norm: bool) -> Result<Self, HipError> {
        let mut b = std::ptr::null_mut();
        let rc = unsafe {
            melspec_bank_from_mel(&mut b, -1, sample_rate, n_fft as c_int, n_mels as c_int, f_min.unwrap_or(-1.0), f_max.unwrap_or(-1.0),
                                  htk as c_int, norm as c_int)
        };
        if rc != 0 {
            return Err(unavailable(rc));
        }
        Ok(Self { b })
    }
    pub fn n_mels(&self) -> usize { unsafe { melspec_bank_n_mels(self.b) as usize } }
    pub fn fft_bins(&self) -> usize { unsafe { melspec_bank_fft_bins(self.b) as usize } }
    pub fn non_zero_weights(&self) -> usize { unsafe { melspec_bank_non_zero_weights(self.b) as usize } }
    /// `weights_for_mel(mel_idx)` (src/mel.rs:102-104): `(bin, weight)` in ascending bin order.
    pub fn weights_for_mel(&self, mel_idx: usize) -> Vec<(usize, f64)> {
        let n = unsafe { melspec_bank_weights_for_mel(self.b, mel_idx as c_int, std::ptr::null_mut(), std::ptr::null_mut(), 0) };
        if n <= 0 {
            return Vec::new();
        }
        let (mut bins, mut w) = (vec![0 as c_int; n as usize], vec![0.0f64; n as usize]);
        unsafe { melspec_bank_weights_for_mel(self.b, mel_idx as c_int, bins.as_mut_ptr(), w.as_mut_ptr(), n) };
        bins.into_iter().map(|b| b as usize).zip(w).collect()
    }
    /// `project_power_f64` (src/mel.rs:106-125) for every row of `power` (`[frames][fft_bins]`) -> `[frames][n_mels]`.
    pub fn project_power_f64(&mut self, power: &ndarray::Array2<f64>) -> Result<ndarray::Array2<f64>, HipError> {
        let p = power.as_standard_layout();
        let mut out = vec![0.0f64; p.nrows() * self.n_mels()];
        check(unsafe { melspec_bank_project_power_host(self.b, p.as_ptr() as *const c_void, STFT_F64, p.nrows(), out.as_mut_ptr() as *mut c_void) })?;
        Ok(ndarray::Array2::from_shape_vec((p.nrows(), self.n_mels()), out).expect("shape"))
    }
    /// `project_power_f32` (src/mel.rs:127-146)
    pub fn project_power_f32(&mut self, power: &ndarray::Array2<f32>) -> Result<ndarray::Array2<f32>, HipError> {
        let p = power.as_standard_layout();
        let mut out = vec![0.0f32; p.nrows() * self.n_mels()];
        check(unsafe { melspec_bank_project_power_host(self.b, p.as_ptr() as *const c_void, STFT_F32, p.nrows(), out.as_mut_ptr() as *mut c_void) })?;
        Ok(ndarray::Array2::from_shape_vec((p.nrows(), self.n_mels()), out).expect("shape"))
    }
    /// `log_mel_spectrogram(stft, mel_filters)` (src/mel.rs:436-441) for the frames of `compute_all`: log10(max(E, 1e-10)), not normalised.
    pub fn log_mel_spectrogram(&mut self, frames: &[Vec<num::Complex<f64>>]) -> Result<ndarray::Array2<f64>, HipError> {
        let n_fft = frames.first().map_or(0, |f| f.len());
        let mut flat = Vec::with_capacity(frames.len() * n_fft * 2);
        for f in frames {
            assert_eq!(f.len(), n_fft);
            for c in f {
                flat.push(c.re);
                flat.push(c.im);
            }
        }
        let mut out = vec![0.0f64; frames.len() * self.n_mels()];
        if !frames.is_empty() {
            check(unsafe { melspec_bank_log_mel_host(self.b, flat.as_ptr() as *const c_void, STFT_F64, n_fft as c_int, frames.len(), out.as_mut_ptr()) })?;
        }
        Ok(ndarray::Array2::from_shape_vec((frames.len(), self.n_mels()), out).expect("shape"))
    }
    /// `norm_mel` (src/mel.rs:448-454): one maximum over everything given, then `(max(x, mmax - 8) + 4) / 4`.
    pub fn norm_mel(&mut self, mel: &ndarray::Array2<f64>) -> Result<ndarray::Array2<f64>, HipError> {
        let m = mel.as_standard_layout();
        let mut out = vec![0.0f64; m.len()];
        check(unsafe { melspec_bank_norm_mel_host(self.b, m.as_ptr() as *const c_void, STFT_F64, m.len(), out.as_mut_ptr() as *mut c_void) })?;
        Ok(ndarray::Array2::from_shape_vec(m.raw_dim(), out).expect("shape"))
    }
    /// `norm_mel_vec` (src/mel.rs:457-469)
    pub fn norm_mel_vec(&mut self, mel: &[f32]) -> Result<Vec<f32>, HipError> {
        let mut out = vec![0.0f32; mel.len()];
        check(unsafe { melspec_bank_norm_mel_host(self.b, mel.as_ptr() as *const c_void, STFT_F32, mel.len(), out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
    /// Device-resident forms (power / spectrum / values already in HBM), asynchronous on the bank's stream.
    ///
    /// # Safety
    /// The pointers must be device pointers of the sizes the host forms document.
    pub unsafe fn project_power_device(&mut self, d_power: *const c_void, f64_data: bool, n_frames: u64, d_out: *mut c_void) -> Result<(), HipError> {
        check(melspec_bank_project_power_device(self.b, d_power, if f64_data { STFT_F64 } else { STFT_F32 }, n_frames, d_out, std::ptr::null_mut()))
    }
    /// # Safety
    /// As `project_power_device`.
    pub unsafe fn log_mel_device(&mut self, d_stft: *const c_void, f64_data: bool, n_fft: usize, n_frames: u64, d_out: *mut f64) -> Result<(), HipError> {
        check(melspec_bank_log_mel_device(self.b, d_stft, if f64_data { STFT_F64 } else { STFT_F32 }, n_fft as c_int, n_frames, d_out, std::ptr::null_mut()))
    }
    /// # Safety
    /// As `project_power_device`.
    pub unsafe fn norm_mel_device(&mut self, d_in: *const c_void, f64_data: bool, n_values: u64, d_out: *mut c_void) -> Result<(), HipError> {
        check(melspec_bank_norm_mel_device(self.b, d_in, if f64_data { STFT_F64 } else { STFT_F32 }, n_values, d_out, std::ptr::null_mut()))
    }
}
impl Drop for HipSparseMelFilterbank {
    fn drop(&mut self) {
        unsafe { melspec_bank_destroy(self.b) }
    }
}

impl HipTga {
    /// `tga_8bit_data(interleave_frames(mel(clip), false, min_width))` (src/quant.rs:38-64) for `n_clips` equal-length clips that are
    /// already in HBM: images `[n_clips][n_mels][W]` and one blob per clip stay on the device; the quantiser reads every image once.
    pub fn encode_pcm_uniform_device(&mut self, mel: &mut HipMelSpectrogram, pcm: &HipDeviceBuffer<f32>, clip_len: usize, n_clips: usize,
                                     min_width: usize, images: &mut HipDeviceBuffer<f32>, blobs: &mut HipDeviceBuffer<u8>) -> Result<(usize, usize), HipError> {
        let w = unsafe { melspec_interleaved_width(mel.ctx, clip_len, min_width) };
        let (mut n, mut stride, mut last) = (0u32, 0usize, 0usize);
        check(unsafe { melspec_tga_layout(mel.n_mels as c_int, w, &mut n, &mut stride, &mut last) })?;
        let blob_stride = stride * n as usize;
        let fits = |a: usize, b: usize, c: usize, cap: usize| a.checked_mul(b).and_then(|x| x.checked_mul(c)).map_or(false, |x| x <= cap);
        if !fits(n_clips, clip_len, 1, pcm.len()) || !fits(n_clips, mel.n_mels, w, images.len()) || !fits(n_clips, blob_stride, 1, blobs.len()) {
            return Err(HipError::Runtime("buffer too small".into()));
        }
        check(unsafe {
            melspec_tga_encode_pcm_uniform_device(self.q, mel.ctx, pcm.as_ptr(), clip_len as u64, clip_len as u64, n_clips as u32, min_width as u64,
                                                  images.as_mut_ptr(), blobs.as_mut_ptr(), blob_stride, std::ptr::null_mut())
        })?;
        check(unsafe { melspec_synchronize(mel.ctx, std::ptr::null_mut()) })?;       // stream NULL = the context's stream
        Ok((w, blob_stride))
    }
}

/// Waits for every queued operation on the current device.
pub fn hip_device_synchronize() -> Result<(), HipError> {
    check(unsafe { melspec_device_synchronize() })
}
