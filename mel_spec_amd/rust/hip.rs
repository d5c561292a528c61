//! `src/hip.rs` for wavey-ai/mel-spec -- the MI355X (gfx950) backend behind feature `hip`.
//!
//! NOT COMPILED IN THIS REPOSITORY (the build image has no Rust toolchain); it is the binding a
//! maintainer drops into the reference crate.  It is deliberately thin: all work happens behind the
//! C ABI of `libmelspec_hip.so` (include/melspec_hip.h).  Public surface mirrors
//! `CudaMelSpectrogram` (src/cuda.rs:27-140): same constructor arguments, same method, same error split.
use std::ffi::{c_char, c_int, c_void, CStr};
use std::fmt;

#[derive(Debug)]
pub enum HipError {
    Runtime(String),
    Unavailable(String),
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
    fn melspec_stream_push_host(st: *mut Stream, ids: *const u32, samples: *const f32, lens: *const u32, n: u32,
                                out: *mut f32, cap: usize, frames_out: *mut u32) -> c_int;
}

#[repr(C)]
struct Tga {
    _private: [u8; 0],
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
            return Err(HipError::Unavailable(last_error()));
        }
        Ok(Self { ctx, n_mels })
    }

    /// f64 window/FFT/power like `Spectrogram::compute_mel_spectrogram_cpu` and the CUDA backend's Z2Z FFT.
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
        if unsafe { melspec_tga_create(&mut q, -1) } != 0 {
            return Err(HipError::Unavailable(last_error()));
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
        if unsafe { melspec_stream_create(&mut st, mel.ctx, 1, max_chunk as u32) } != 0 {
            return Err(HipError::Unavailable(last_error()));
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
impl Drop for HipStream<'_> {
    fn drop(&mut self) {
        unsafe { melspec_stream_destroy(self.st) }
    }
}
