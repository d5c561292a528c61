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
    fn melspec_last_error() -> *const c_char;
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
