//! `HipFftImpl<T>`: a drop-in `kofft::fft::FftImpl<T>` backed by the MI355X library (include/kofft_hip.h).
//!
//! Because `RealFftImpl<T>` is a blanket impl over every `FftImpl<T>` (kofft rfft.rs:837) and `stft()`,
//! `StftStream`, `batch()` are generic over `FftImpl` (stft.rs:76, 160; fft.rs:2156), existing callers
//! only change the type they construct.  The added inherent methods (`fft_batch`, `rfft_batch`,
//! `stft_contiguous`) are the contiguous entry points that escape the `Vec<Vec<_>>` layout.
//!
//! SOURCE ONLY: written against kofft 0.1.5's public API; not compiled in the build image (no rustc).
use core::ffi::{c_char, c_int, c_void};
use core::marker::PhantomData;
use kofft::fft::{Complex, Complex32, Complex64, FftError, FftImpl, FftStrategy};
use kofft::num::Float;

#[repr(C)]
pub struct KofftHipCtx {
    _private: [u8; 0],
}

extern "C" {
    fn kofft_hip_create(device: c_int, out: *mut *mut KofftHipCtx) -> c_int;
    fn kofft_hip_destroy(ctx: *mut KofftHipCtx) -> c_int;
    fn kofft_hip_last_error(ctx: *const KofftHipCtx) -> *const c_char;
    fn kofft_hip_fft_c32(ctx: *mut KofftHipCtx, data: *mut f32, n: usize, batch: usize, inverse: c_int) -> c_int;
    fn kofft_hip_fft_c64(ctx: *mut KofftHipCtx, data: *mut f64, n: usize, batch: usize, inverse: c_int) -> c_int;
    fn kofft_hip_fft_radix4_c32(ctx: *mut KofftHipCtx, data: *mut f32, n: usize, batch: usize) -> c_int;
    fn kofft_hip_fft_radix4_c64(ctx: *mut KofftHipCtx, data: *mut f64, n: usize, batch: usize) -> c_int;
    fn kofft_hip_fft_c32_strided(ctx: *mut KofftHipCtx, data: *mut f32, len: usize, stride: usize, n: usize, inverse: c_int) -> c_int;
    fn kofft_hip_fft_c64_strided(ctx: *mut KofftHipCtx, data: *mut f64, len: usize, stride: usize, n: usize, inverse: c_int) -> c_int;
    fn kofft_hip_rfft_f32(ctx: *mut KofftHipCtx, input: *const f32, out: *mut f32, window: *const f32, n: usize, batch: usize) -> c_int;
    fn kofft_hip_stft_f32(ctx: *mut KofftHipCtx, signal: *const f32, len: usize, window: *const f32, win_len: usize,
                          hop: usize, out: *mut f32, frames: usize) -> c_int;
    fn kofft_hip_stft_parallel_f32(ctx: *mut KofftHipCtx, signal: *const f32, len: usize, window: *const f32, win_len: usize,
                                   hop: usize, out: *mut f32, frames: usize) -> c_int;
    fn kofft_hip_irfft_f32(ctx: *mut KofftHipCtx, input: *const f32, out: *mut f32, n: usize, batch: usize) -> c_int;
    fn kofft_hip_istft_f32(ctx: *mut KofftHipCtx, frames_data: *mut f32, frames: usize, window: *const f32, win_len: usize,
                           hop: usize, output: *mut f32, out_len: usize, scratch: *mut f32, scratch_len: usize) -> c_int;
    fn kofft_hip_istft_parallel_f32(ctx: *mut KofftHipCtx, frames_data: *const f32, frames: usize, window: *const f32,
                                    win_len: usize, hop: usize, output: *mut f32, out_len: usize) -> c_int;
    fn kofft_hip_stft_magnitudes_f32(ctx: *mut KofftHipCtx, samples: *const f32, len: usize, win_len: usize, hop: usize,
                                     mags: *mut f32, frames: usize, max_mag: *mut f32) -> c_int;
    fn kofft_hip_fftnd_c32(ctx: *mut KofftHipCtx, data: *mut f32, depth: usize, rows: usize, cols: usize, inverse: c_int) -> c_int;
    fn kofft_hip_fftnd_c64(ctx: *mut KofftHipCtx, data: *mut f64, depth: usize, rows: usize, cols: usize, inverse: c_int) -> c_int;
    // multi-GPU: one process, one context per device, optional RCCL all-gather (include/kofft_hip.h, "multi-GPU")
    fn kofft_hip_multi_create(ngpu: c_int, devices: *const c_int, out: *mut *mut KofftHipMulti) -> c_int;
    fn kofft_hip_multi_destroy(m: *mut KofftHipMulti) -> c_int;
    fn kofft_hip_multi_last_error(m: *const KofftHipMulti) -> *const c_char;
    fn kofft_hip_multi_stft_f32(m: *mut KofftHipMulti, signal: *const f32, len: usize, window: *const f32, win_len: usize,
                                hop: usize, out: *mut f32, frames: usize, allgather: c_int, d_out_per_gpu: *mut *mut f32) -> c_int;
    fn kofft_hip_stft_f32_multi(ngpu: c_int, signal: *const f32, len: usize, window: *const f32, win_len: usize, hop: usize,
                                out: *mut f32, frames: usize, allgather: c_int) -> c_int;
    fn kofft_hip_multi_fft_c32(m: *mut KofftHipMulti, data: *mut f32, n: usize, batch: usize, inverse: c_int) -> c_int;
    fn kofft_hip_multi_fft_c64(m: *mut KofftHipMulti, data: *mut f64, n: usize, batch: usize, inverse: c_int) -> c_int;
    fn kofft_hip_multi_rfft_f32(m: *mut KofftHipMulti, input: *const f32, out: *mut f32, window: *const f32, n: usize, batch: usize) -> c_int;
    fn kofft_hip_multi_shard(m: *const KofftHipMulti, total: usize, rank: c_int, first: *mut usize, count: *mut usize) -> c_int;
    fn kofft_hip_multi_stft_slice(m: *const KofftHipMulti, len: usize, win_len: usize, hop: usize, frames: usize, rank: c_int,
                                  first_sample: *mut usize, count: *mut usize) -> c_int;
    fn kofft_hip_multi_synchronize(m: *mut KofftHipMulti) -> c_int;
    fn kofft_hip_multi_last_timing_ex(m: *const KofftHipMulti, upload_ms: *mut f32, kernel_ms: *mut f32, gather_ms: *mut f32,
                                      download_ms: *mut f32, wall_ms: *mut f32) -> c_int;
    fn kofft_hip_multi_set_gather(m: *mut KofftHipMulti, mode: c_int) -> c_int;
    fn kofft_hip_multi_gather_mode(m: *const KofftHipMulti, configured: *mut c_int, last: *mut c_int) -> c_int;
    // device-resident twins: one device pointer per device, asynchronous
    fn kofft_hip_multi_fft_c32_dev(m: *mut KofftHipMulti, d_data_per_gpu: *const *mut f32, n: usize, batch: usize, inverse: c_int) -> c_int;
    fn kofft_hip_multi_fft_c64_dev(m: *mut KofftHipMulti, d_data_per_gpu: *const *mut f64, n: usize, batch: usize, inverse: c_int) -> c_int;
    fn kofft_hip_multi_rfft_f32_dev(m: *mut KofftHipMulti, d_in_per_gpu: *const *const f32, d_out_per_gpu: *const *mut f32,
                                    d_window_per_gpu: *const *const f32, n: usize, batch: usize) -> c_int;
    fn kofft_hip_multi_stft_f32_dev(m: *mut KofftHipMulti, d_signal_per_gpu: *const *const f32, len: usize,
                                    d_window_per_gpu: *const *const f32, win_len: usize, hop: usize, frames: usize, allgather: c_int,
                                    d_out_per_gpu: *mut *mut f32) -> c_int;
}

/// Slowest device's time in each phase of the last multi-GPU call (HIP events), milliseconds; absent phases are 0.
#[derive(Debug, Default, Clone, Copy)]
pub struct MultiTiming {
    pub upload_ms: f32,
    pub kernel_ms: f32,
    pub gather_ms: f32,
    pub download_ms: f32,
    pub wall_ms: f32,
}

#[repr(C)]
pub struct KofftHipMulti {
    _private: [u8; 0],
}

/// 0 = Ok, 1..=6 = FftError in declaration order (kofft fft.rs:447-454).  FftError has no variant for a
/// device failure, so a negative status (HIP error, unsupported length) panics with the library's message.
fn status(ctx: *const KofftHipCtx, rc: c_int) -> Result<(), FftError> {
    match rc {
        0 => Ok(()),
        1 => Err(FftError::EmptyInput),
        2 => Err(FftError::NonPowerOfTwoNoStd),
        3 => Err(FftError::MismatchedLengths),
        4 => Err(FftError::InvalidStride),
        5 => Err(FftError::InvalidHopSize),
        6 => Err(FftError::InvalidValue),
        _ => {
            let msg = unsafe { std::ffi::CStr::from_ptr(kofft_hip_last_error(ctx)) }.to_string_lossy().into_owned();
            panic!("kofft-hip device error {rc}: {msg}");
        }
    }
}

/// One device context per instance; `Send` but not `Sync`, like `ScalarFftImpl` (kofft fft.rs:589-605).
pub struct HipFftImpl<T: Float> {
    ctx: *mut KofftHipCtx,
    /// `fft_with_strategy(.., Radix4)` reproduces kofft's `fft_radix4` bytes (fft.rs:1356, 1455-1548; NOT a DFT from n = 16):
    /// strict drop-in, ON by default.  `KOFFT_HIP_RADIX4_COMPAT=0`, or `false` set by the caller, opts out (the true
    /// transform for every strategy).
    pub radix4_compat: bool,
    _t: PhantomData<T>,
}

unsafe impl<T: Float> Send for HipFftImpl<T> {}

impl<T: Float> HipFftImpl<T> {
    pub fn new(device: i32) -> Self {
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { kofft_hip_create(device as c_int, &mut ctx) };
        assert!(rc == 0 && !ctx.is_null(), "kofft_hip_create failed: {rc}");
        let radix4_compat = std::env::var("KOFFT_HIP_RADIX4_COMPAT").map_or(true, |v| v != "0");
        Self { ctx, radix4_compat, _t: PhantomData }
    }
}

impl<T: Float> Default for HipFftImpl<T> {
    fn default() -> Self {
        Self::new(0)
    }
}

impl<T: Float> Drop for HipFftImpl<T> {
    fn drop(&mut self) {
        unsafe { kofft_hip_destroy(self.ctx) };
    }
}

macro_rules! impl_fft {
    ($t:ty, $cplx:ty, $fft:ident, $strided:ident, $radix4:ident) => {
        impl FftImpl<$t> for HipFftImpl<$t> {
            // Complex<T> is #[repr(C)] {re, im} (kofft num.rs:105-110, tests/complex_repr.rs): a slice of n
            // complex values is 2n scalars.
            fn fft(&self, input: &mut [$cplx]) -> Result<(), FftError> {
                status(self.ctx, unsafe { $fft(self.ctx, input.as_mut_ptr() as *mut $t, input.len(), 1, 0) })
            }
            fn ifft(&self, input: &mut [$cplx]) -> Result<(), FftError> {
                status(self.ctx, unsafe { $fft(self.ctx, input.as_mut_ptr() as *mut $t, input.len(), 1, 1) })
            }
            fn fft_strided(&self, input: &mut [$cplx], stride: usize, scratch: &mut [$cplx]) -> Result<(), FftError> {
                status(self.ctx, unsafe {
                    $strided(self.ctx, input.as_mut_ptr() as *mut $t, input.len(), stride, scratch.len(), 0)
                })
            }
            fn ifft_strided(&self, input: &mut [$cplx], stride: usize, scratch: &mut [$cplx]) -> Result<(), FftError> {
                status(self.ctx, unsafe {
                    $strided(self.ctx, input.as_mut_ptr() as *mut $t, input.len(), stride, scratch.len(), 1)
                })
            }
            fn fft_out_of_place_strided(&self, input: &[$cplx], in_stride: usize, output: &mut [$cplx],
                                        out_stride: usize) -> Result<(), FftError> {
                // same checks, same order as kofft fft.rs:1268-1277
                if in_stride == 0 || out_stride == 0 { return Err(FftError::InvalidStride); }
                if input.len() % in_stride != 0 || output.len() % out_stride != 0 { return Err(FftError::InvalidStride); }
                let n = input.len() / in_stride;
                if output.len() / out_stride != n { return Err(FftError::MismatchedLengths); }
                let mut scratch: Vec<$cplx> = (0..n).map(|i| input[i * in_stride]).collect();
                self.fft(&mut scratch)?;
                for i in 0..n { output[i * out_stride] = scratch[i]; }
                Ok(())
            }
            fn ifft_out_of_place_strided(&self, input: &[$cplx], in_stride: usize, output: &mut [$cplx],
                                         out_stride: usize) -> Result<(), FftError> {
                if in_stride == 0 || out_stride == 0 { return Err(FftError::InvalidStride); }
                if input.len() % in_stride != 0 || output.len() % out_stride != 0 { return Err(FftError::InvalidStride); }
                let n = input.len() / in_stride;
                if output.len() / out_stride != n { return Err(FftError::MismatchedLengths); }
                let mut scratch: Vec<$cplx> = (0..n).map(|i| input[i * in_stride]).collect();
                self.ifft(&mut scratch)?;
                for i in 0..n { output[i * out_stride] = scratch[i]; }
                Ok(())
            }
            fn fft_with_strategy(&self, input: &mut [$cplx], strategy: FftStrategy) -> Result<(), FftError> {
                // kofft fft.rs:1337-1363: Radix2 / SplitRadix / Auto run the Stockham path (fft and stockham_fft agree for
                // n >= 2); Radix4 runs the crate's fft_radix4 arm (fft.rs:1356) byte for byte -- not a DFT from n = 16 (its
                // digit-reversal loop is wrong), but a drop-in returns what the crate returns.  `radix4_compat = false`
                // opts out.
                if input.is_empty() { return Err(FftError::EmptyInput); }
                if input.len() == 1 { return Ok(()); }
                if strategy == FftStrategy::Radix4 && self.radix4_compat { return self.fft_radix4(input); }
                self.fft(input)
            }
        }

        impl HipFftImpl<$t> {
            /// `ScalarFftImpl::fft_radix4` (kofft fft.rs:1455-1548), byte for byte.
            pub fn fft_radix4(&self, input: &mut [$cplx]) -> Result<(), FftError> {
                status(self.ctx, unsafe { $radix4(self.ctx, input.as_mut_ptr() as *mut $t, input.len(), 1) })
            }
        }

        impl HipFftImpl<$t> {
            /// `fft::batch` over one contiguous `[batch * n]` buffer (one kernel launch).
            pub fn fft_batch(&self, data: &mut [$cplx], n: usize, inverse: bool) -> Result<(), FftError> {
                if n != 0 && data.len() % n != 0 { return Err(FftError::MismatchedLengths); }
                let batch = if n == 0 { 1 } else { data.len() / n };
                status(self.ctx, unsafe { $fft(self.ctx, data.as_mut_ptr() as *mut $t, n, batch, inverse as c_int) })
            }
        }
    };
}

impl_fft!(f32, Complex32, kofft_hip_fft_c32, kofft_hip_fft_c32_strided, kofft_hip_fft_radix4_c32);
impl_fft!(f64, Complex64, kofft_hip_fft_c64, kofft_hip_fft_c64_strided, kofft_hip_fft_radix4_c64);

macro_rules! impl_ndfft {
    ($t:ty, $cplx:ty, $nd:ident) => {
        impl HipFftImpl<$t> {
            /// `ndfft::fft2d_inplace` (kofft ndfft.rs:74-101): kofft's function takes `&ScalarFftImpl<T>` concretely,
            /// so the device version is an inherent method with the same length checks.
            pub fn fft2d_inplace(&self, data: &mut [$cplx], rows: usize, cols: usize, scratch_col: &mut [$cplx]) -> Result<(), FftError> {
                if rows * cols != data.len() { return Err(FftError::MismatchedLengths); }
                if rows == 0 || cols == 0 { return Ok(()); }
                if scratch_col.len() != rows { return Err(FftError::MismatchedLengths); }
                status(self.ctx, unsafe { $nd(self.ctx, data.as_mut_ptr() as *mut $t, 1, rows, cols, 0) })
            }
            /// `ndfft::fft3d_inplace` (kofft ndfft.rs:114-155).
            pub fn fft3d_inplace(&self, data: &mut [$cplx], depth: usize, rows: usize, cols: usize, tube: &mut [$cplx],
                                 row: &mut [$cplx], col: &mut [$cplx]) -> Result<(), FftError> {
                if depth * rows * cols != data.len() { return Err(FftError::MismatchedLengths); }
                if depth == 0 || rows == 0 || cols == 0 { return Ok(()); }
                if tube.len() != depth || row.len() != rows || col.len() != cols { return Err(FftError::MismatchedLengths); }
                status(self.ctx, unsafe { $nd(self.ctx, data.as_mut_ptr() as *mut $t, depth, rows, cols, 0) })
            }
        }
    };
}
impl_ndfft!(f32, Complex32, kofft_hip_fftnd_c32);
impl_ndfft!(f64, Complex64, kofft_hip_fftnd_c64);

impl HipFftImpl<f32> {
    /// Batched real FFT with an optional fused window: `out` holds `batch * (n/2 + 1)` complex values.
    pub fn rfft_batch(&self, input: &[f32], n: usize, window: Option<&[f32]>, out: &mut [Complex<f32>]) -> Result<(), FftError> {
        if n == 0 { return Err(FftError::EmptyInput); }
        let batch = input.len() / n;
        if input.len() % n != 0 || out.len() != batch * (n / 2 + 1) { return Err(FftError::MismatchedLengths); }
        if let Some(w) = window { if w.len() != n { return Err(FftError::MismatchedLengths); } }
        let wp = window.map_or(core::ptr::null(), |w| w.as_ptr());
        status(self.ctx, unsafe { kofft_hip_rfft_f32(self.ctx, input.as_ptr(), out.as_mut_ptr() as *mut f32, wp, n, batch) })
    }

    /// `stft::stft` into one contiguous `frames * window.len()` buffer (same checks as kofft stft.rs:83-89).
    pub fn stft_contiguous(&self, signal: &[f32], window: &[f32], hop_size: usize, out: &mut [Complex32]) -> Result<(), FftError> {
        let frames = if window.is_empty() { 0 } else { out.len() / window.len() };
        status(self.ctx, unsafe {
            kofft_hip_stft_f32(self.ctx, signal.as_ptr(), signal.len(), window.as_ptr(), window.len(), hop_size,
                               out.as_mut_ptr() as *mut f32, frames)
        })
    }
}

impl HipFftImpl<f32> {
    /// `stft::parallel` (kofft stft.rs:232-263) ignores the `fft` it is handed and builds a `ScalarFftImpl` per
    /// frame; this is the same arithmetic on the device (only `hop_size == 0` is rejected, as there).
    pub fn stft_parallel_contiguous(&self, signal: &[f32], window: &[f32], hop_size: usize, out: &mut [Complex32]) -> Result<(), FftError> {
        let frames = if window.is_empty() { 0 } else { out.len() / window.len() };
        status(self.ctx, unsafe {
            kofft_hip_stft_parallel_f32(self.ctx, signal.as_ptr(), signal.len(), window.as_ptr(), window.len(), hop_size,
                                        out.as_mut_ptr() as *mut f32, frames)
        })
    }

    /// Batched `irfft` (kofft rfft.rs:468-508): `input` holds `batch * (n/2 + 1)` complex values, `out` `batch * n` reals.
    pub fn irfft_batch(&self, input: &[Complex32], n: usize, out: &mut [f32]) -> Result<(), FftError> {
        if n == 0 { return Err(FftError::EmptyInput); }
        let batch = out.len() / n;
        if out.len() % n != 0 || input.len() != batch * (n / 2 + 1) { return Err(FftError::MismatchedLengths); }
        status(self.ctx, unsafe { kofft_hip_irfft_f32(self.ctx, input.as_ptr() as *const f32, out.as_mut_ptr(), n, batch) })
    }

    /// `stft::istft` (kofft stft.rs:117-156) over one contiguous `frames * window.len()` buffer: the frames are
    /// inverse-transformed in place, `output` is accumulated into and normalised, `scratch` receives the window-square sums.
    pub fn istft_contiguous(&self, frames: &mut [Complex32], window: &[f32], hop_size: usize, output: &mut [f32],
                            scratch: &mut [f32]) -> Result<(), FftError> {
        let count = if window.is_empty() { 0 } else { frames.len() / window.len() };
        status(self.ctx, unsafe {
            kofft_hip_istft_f32(self.ctx, frames.as_mut_ptr() as *mut f32, count, window.as_ptr(), window.len(), hop_size,
                                output.as_mut_ptr(), output.len(), scratch.as_mut_ptr(), scratch.len())
        })
    }

    /// `stft::inverse_parallel` (kofft stft.rs:289-343): frames untouched, tiny-norm samples set to zero.
    pub fn inverse_parallel_contiguous(&self, frames: &[Complex32], window: &[f32], hop_size: usize, output: &mut [f32]) -> Result<(), FftError> {
        let count = if window.is_empty() { 0 } else { frames.len() / window.len() };
        status(self.ctx, unsafe {
            kofft_hip_istft_parallel_f32(self.ctx, frames.as_ptr() as *const f32, count, window.as_ptr(), window.len(), hop_size,
                                         output.as_mut_ptr(), output.len())
        })
    }

    /// `visual::spectrogram::stft_magnitudes` (kofft visual/spectrogram.rs:52-76): magnitudes of bins
    /// `0 .. win_len/2` of every Hann-windowed frame (row-major `frames x win_len/2`) and their maximum.
    pub fn stft_magnitudes(&self, samples: &[f32], win_len: usize, hop: usize) -> Result<(Vec<Vec<f32>>, f32), FftError> {
        if hop == 0 { return Err(FftError::InvalidHopSize); }
        let frames = (samples.len() + hop - 1) / hop;
        let half = win_len / 2;
        let mut flat = vec![0.0f32; frames * half];
        let mut max_mag = 0.0f32;
        status(self.ctx, unsafe {
            kofft_hip_stft_magnitudes_f32(self.ctx, samples.as_ptr(), samples.len(), win_len, hop, flat.as_mut_ptr(), frames, &mut max_mag)
        })?;
        Ok((flat.chunks(half.max(1)).take(frames).map(|c| c.to_vec()).collect(), max_mag))
    }
}

/// `stft::parallel` (kofft stft.rs:232-263) across `ngpu` devices of this process: frames are the parallel unit, device
/// `r` computes `[r*ceil(F/G), ...)`; `allgather` adds the RCCL all-gather of BASELINE config #4.  Same checks as
/// `stft::stft` (stft.rs:83-87).  One call: contexts are created and torn down inside (see `HipMulti` to keep them).
pub fn stft_multi(ngpu: usize, signal: &[f32], window: &[f32], hop_size: usize, output: &mut [Vec<Complex32>], allgather: bool)
    -> Result<(), FftError> {
    let (frames, wl) = (output.len(), window.len());
    let mut flat = vec![Complex32::new(0.0, 0.0); frames * wl];
    let rc = unsafe {
        kofft_hip_stft_f32_multi(ngpu as c_int, signal.as_ptr(), signal.len(), window.as_ptr(), wl, hop_size,
                                 flat.as_mut_ptr() as *mut f32, frames, allgather as c_int)
    };
    status(core::ptr::null(), rc)?;
    for (f, frame) in output.iter_mut().enumerate() {
        frame.clear();
        frame.extend_from_slice(&flat[f * wl..(f + 1) * wl]);
    }
    Ok(())
}

/// Handle form of the above: per-device contexts, buffers and RCCL communicators live as long as the value.
pub struct HipMulti {
    h: *mut KofftHipMulti,
}

impl HipMulti {
    pub fn new(ngpu: usize) -> Result<Self, FftError> {
        let mut h = core::ptr::null_mut();
        status(core::ptr::null(), unsafe { kofft_hip_multi_create(ngpu as c_int, core::ptr::null(), &mut h) })?;
        Ok(Self { h })
    }

    pub fn stft_contiguous(&self, signal: &[f32], window: &[f32], hop_size: usize, out: &mut [Complex32], allgather: bool)
        -> Result<(), FftError> {
        let frames = if window.is_empty() { 0 } else { out.len() / window.len() };
        let rc = unsafe {
            kofft_hip_multi_stft_f32(self.h, signal.as_ptr(), signal.len(), window.as_ptr(), window.len(), hop_size,
                                     out.as_mut_ptr() as *mut f32, frames, allgather as c_int, core::ptr::null_mut())
        };
        if rc < 0 {
            let msg = unsafe { std::ffi::CStr::from_ptr(kofft_hip_multi_last_error(self.h)) }.to_string_lossy().into_owned();
            panic!("kofft-hip multi-GPU error {rc}: {msg}");
        }
        status(core::ptr::null(), rc)
    }

    fn check(&self, rc: c_int) -> Result<(), FftError> {
        if rc < 0 {
            let msg = unsafe { std::ffi::CStr::from_ptr(kofft_hip_multi_last_error(self.h)) }.to_string_lossy().into_owned();
            panic!("kofft-hip multi-GPU error {rc}: {msg}");
        }
        status(core::ptr::null(), rc)
    }

    /// `fft::batch` (fft.rs:2156-2175) over `batch` contiguous transforms of length `n`, the batch in G contiguous blocks.
    pub fn fft_batch_c32(&self, data: &mut [Complex32], n: usize, inverse: bool) -> Result<(), FftError> {
        let batch = if n == 0 { 0 } else { data.len() / n };
        self.check(unsafe { kofft_hip_multi_fft_c32(self.h, data.as_mut_ptr() as *mut f32, n, batch, inverse as c_int) })
    }
    pub fn fft_batch_c64(&self, data: &mut [Complex64], n: usize, inverse: bool) -> Result<(), FftError> {
        let batch = if n == 0 { 0 } else { data.len() / n };
        self.check(unsafe { kofft_hip_multi_fft_c64(self.h, data.as_mut_ptr() as *mut f64, n, batch, inverse as c_int) })
    }
    /// `rfft_direct` (rfft.rs:425-465) on every row of `n` reals (optional window product first), rows sharded over the devices.
    pub fn rfft_batch(&self, input: &[f32], out: &mut [Complex32], window: Option<&[f32]>, n: usize) -> Result<(), FftError> {
        let batch = if n == 0 { 0 } else { input.len() / n };
        if out.len() != batch * (n / 2 + 1) || window.map_or(false, |w| w.len() != n) {
            return Err(FftError::MismatchedLengths);
        }
        let w = window.map_or(core::ptr::null(), |w| w.as_ptr());
        self.check(unsafe { kofft_hip_multi_rfft_f32(self.h, input.as_ptr(), out.as_mut_ptr() as *mut f32, w, n, batch) })
    }
    /// (first, count) of the `total` units device `rank` owns.
    pub fn shard(&self, total: usize, rank: usize) -> (usize, usize) {
        let (mut f, mut c) = (0usize, 0usize);
        unsafe { kofft_hip_multi_shard(self.h, total, rank as c_int, &mut f, &mut c) };
        (f, c)
    }
    /// (first_sample, count) of the signal slice device `rank` needs for its frames (block + halo).
    pub fn stft_slice(&self, len: usize, win_len: usize, hop: usize, frames: usize, rank: usize) -> (usize, usize) {
        let (mut f, mut c) = (0usize, 0usize);
        unsafe { kofft_hip_multi_stft_slice(self.h, len, win_len, hop, frames, rank as c_int, &mut f, &mut c) };
        (f, c)
    }
    /// Device-resident forms: `d_*[r]` is a pointer valid on device r; asynchronous, `synchronize()` waits.
    ///
    /// # Safety
    /// The pointers must be device allocations of the sizes include/kofft_hip.h states for each entry.
    pub unsafe fn fft_c32_dev(&self, d_data: &[*mut f32], n: usize, batch: usize, inverse: bool) -> Result<(), FftError> {
        self.check(kofft_hip_multi_fft_c32_dev(self.h, d_data.as_ptr(), n, batch, inverse as c_int))
    }
    /// # Safety
    /// As `fft_c32_dev`.
    pub unsafe fn fft_c64_dev(&self, d_data: &[*mut f64], n: usize, batch: usize, inverse: bool) -> Result<(), FftError> {
        self.check(kofft_hip_multi_fft_c64_dev(self.h, d_data.as_ptr(), n, batch, inverse as c_int))
    }
    /// # Safety
    /// As `fft_c32_dev`.
    pub unsafe fn rfft_dev(&self, d_in: &[*const f32], d_out: &[*mut f32], d_window: Option<&[*const f32]>, n: usize, batch: usize)
        -> Result<(), FftError> {
        let w = d_window.map_or(core::ptr::null(), |w| w.as_ptr());
        self.check(kofft_hip_multi_rfft_f32_dev(self.h, d_in.as_ptr(), d_out.as_ptr(), w, n, batch))
    }
    /// `d_out[r]` null on entry = a buffer owned by the handle is used and written back into the slice.
    ///
    /// # Safety
    /// As `fft_c32_dev`.
    pub unsafe fn stft_dev(&self, d_signal: &[*const f32], len: usize, d_window: &[*const f32], win_len: usize, hop: usize,
                           frames: usize, allgather: bool, d_out: &mut [*mut f32]) -> Result<(), FftError> {
        self.check(kofft_hip_multi_stft_f32_dev(self.h, d_signal.as_ptr(), len, d_window.as_ptr(), win_len, hop, frames,
                                                allgather as c_int, d_out.as_mut_ptr()))
    }
    pub fn synchronize(&self) -> Result<(), FftError> {
        self.check(unsafe { kofft_hip_multi_synchronize(self.h) })
    }
    /// 1 = RCCL (grouped in-place ncclAllGather), 2 = direct (peer copies on a stream per peer)
    pub fn set_gather(&self, mode: i32) -> Result<(), FftError> {
        self.check(unsafe { kofft_hip_multi_set_gather(self.h, mode as c_int) })
    }
    /// the form the last call's exchange ran (0: it had none)
    pub fn last_gather(&self) -> i32 {
        let mut last: c_int = 0;
        unsafe { kofft_hip_multi_gather_mode(self.h, std::ptr::null_mut(), &mut last) };
        last as i32
    }
    pub fn last_timing(&self) -> MultiTiming {
        let mut t = MultiTiming::default();
        unsafe {
            kofft_hip_multi_last_timing_ex(self.h, &mut t.upload_ms, &mut t.kernel_ms, &mut t.gather_ms, &mut t.download_ms, &mut t.wall_ms)
        };
        t
    }
}

impl Drop for HipMulti {
    fn drop(&mut self) {
        unsafe { kofft_hip_multi_destroy(self.h) };
    }
}

#[allow(dead_code)]
fn _assert_traits() {
    fn is_fft_impl<F: FftImpl<f32>>() {}
    is_fft_impl::<HipFftImpl<f32>>();
    let _ = core::mem::size_of::<*mut c_void>();
}
