/*
 * kofft_hip.h -- C ABI of the MI355X (gfx950) implementation of kofft's hot path:
 * batched power-of-two complex FFT, real-FFT post-pass, windowed STFT.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface
 * (okian/kofft v0.1.5, file:line) it stands in for; a Rust shim implementing
 * kofft's `FftImpl<T>` trait (fft.rs:466-587) binds exactly these symbols --
 * see INTEGRATION.md for the shim and for the C++/Python host mirrors.
 *
 * Conventions
 *   - Complex data is interleaved {re, im}: kofft's #[repr(C)] Complex<T>
 *     (num.rs:105-110; tests/complex_repr.rs proves [T;2] compatibility), so a
 *     `&mut [Complex32]` passes as `float*` with 2*len floats.
 *   - Plain pointers and sizes only.  No torch / HIP types in any signature; a
 *     HIP stream crosses as `void*`.
 *   - Return value: 0 = Ok; 1..6 = kofft's FftError variants in declaration order
 *     (fft.rs:447-454); negative = runtime failure that FftError cannot express
 *     (the shim maps those to a panic).
 *   - `*_dev` entry points take DEVICE pointers, enqueue on the context's stream and
 *     return without synchronising.  The un-suffixed twins take HOST pointers, stage
 *     through device memory and return after the result is back in the host buffer.
 *   - A context is cheap, owns its twiddle-table cache (the role of FftPlanner,
 *     fft.rs:332-408) and is NOT thread-safe: one context per thread, exactly like
 *     ScalarFftImpl (Send + !Sync, fft.rs:589-605).
 *   - Arithmetic: every butterfly performs the reference's un-fused operations with
 *     the reference's recurrence-generated twiddle tables (generated on the host with
 *     the reference's recipe and uploaded; no device-side trigonometry).
 *   - Lengths: complex transforms take any n up to 2^26 (powers of two) / 2^25 (others:
 *     the reference's Bluestein arm, fft.rs:1088-1132, built from the same kernels).
 *     rfft / irfft (half length), stft / istft / stft_magnitudes (window length) and every
 *     axis of the 2-D / 3-D transforms take the same range: powers of two up to 2^14 (f32) /
 *     2^13 (f64) run fused in one kernel, anything else is composed from the complex
 *     transform plus pack / post-pass kernels (the reference calls fft.fft on any length too:
 *     rfft.rs:447, stft.rs:102).  Beyond that range: KOFFT_ERR_UNSUPPORTED, never a wrong answer.
 */
#ifndef KOFFT_HIP_H
#define KOFFT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---------------------------------------------------------- */
#define KOFFT_OK 0
/* FftError (fft.rs:447-454), discriminant + 1 */
#define KOFFT_ERR_EMPTY_INPUT 1
#define KOFFT_ERR_NON_POWER_OF_TWO_NO_STD 2
#define KOFFT_ERR_MISMATCHED_LENGTHS 3
#define KOFFT_ERR_INVALID_STRIDE 4
#define KOFFT_ERR_INVALID_HOP_SIZE 5
#define KOFFT_ERR_INVALID_VALUE 6
/* outside FftError */
#define KOFFT_ERR_HIP (-1)         /* a HIP runtime call failed; see kofft_hip_last_error */
#define KOFFT_ERR_UNSUPPORTED (-2) /* length not supported by the device path */
#define KOFFT_ERR_NULL (-3)        /* null context / pointer */
#define KOFFT_ERR_ALLOC (-4)       /* host or device allocation failed */
#define KOFFT_ERR_RCCL (-5)        /* RCCL unavailable or a collective failed; see kofft_hip_multi_last_error */

typedef struct kofft_hip_ctx kofft_hip_ctx;

/* Human-readable name of a status code (static string). */
const char *kofft_hip_strerror(int status);
/* Text of the last HIP failure seen by this context ("" if none). */
const char *kofft_hip_last_error(const kofft_hip_ctx *ctx);
/* Library version string, e.g. "kofft-hip 0.1.0 (gfx950)". */
const char *kofft_hip_version(void);

/* ---- context: stands in for ScalarFftImpl::<T>::default() + its FftPlanner ----
 * (fft.rs:600-613, 332-366).  `device` is the HIP device ordinal. */
int kofft_hip_device_count(int *count);
int kofft_hip_create(int device, kofft_hip_ctx **out);
int kofft_hip_destroy(kofft_hip_ctx *ctx);
/* Use a caller-owned hipStream_t (passed as void*) for all *_dev work; NULL restores
 * the context's own stream. */
int kofft_hip_set_stream(kofft_hip_ctx *ctx, void *hip_stream);
int kofft_hip_synchronize(kofft_hip_ctx *ctx);
/* The context's scratch (host-pointer staging, the large-n intermediate, the Bluestein and composed-length work buffers)
 * only grows with the largest call seen; this synchronises the stream and frees it all (tables stay cached). */
int kofft_hip_release_scratch(kofft_hip_ctx *ctx);
/* (no counterpart in kofft) The large-n path (n beyond one workgroup: two or three factor kernels through an intermediate the
 * context owns) chooses the PLACEMENT of that intermediate: the first call that needs it (full chunks of >= 128 MiB) allocates
 * KOFFT_HIP_BIG_PROBE (default 5, at most 8; 0 / 1 = off) candidates, times its own factor kernels on one chunk of scratch data
 * through each and keeps the fastest (the same kernel reads a 512 MiB hipMalloc at 188 or at 225 us depending on
 * where it lies, for the allocation's lifetime).  This reports the last probe: n candidates, per candidate the first factor's
 * and the whole chunk's time in microseconds (arrays of `cap` floats, may be NULL), and which one was kept (n = 0: no probe ran). */
int kofft_hip_big_probe_info(kofft_hip_ctx *ctx, float *first_us, float *total_us, int cap, int *n, int *pick);

/* ---- tables: the planner recipes, on the host --------------------------------
 * kofft_hip_twiddles_*: FftPlanner::get_twiddles(n) (fft.rs:370-408), n/2 complex.
 * kofft_hip_rfft_table_*: build_twiddle_table(m) (rfft.rs:172-183), m complex.
 * kofft_hip_hann_f32: window::hann(len) (window.rs:24-28). */
int kofft_hip_twiddles_f32(size_t n, float *out);
int kofft_hip_twiddles_f64(size_t n, double *out);
int kofft_hip_rfft_table_f32(size_t m, float *out);
int kofft_hip_rfft_table_f64(size_t m, double *out);
int kofft_hip_hann_f32(size_t len, float *out);

/* ---- complex FFT --------------------------------------------------------------
 * FftImpl::fft / FftImpl::ifft (fft.rs:467-468; ScalarFftImpl fft.rs:1054-1082,
 * 1134-1174) applied in place to `batch` contiguous transforms of length n:
 * fft::batch / batch_inverse (fft.rs:2156-2175) over a contiguous layout.
 * data: batch*n complex.  inverse != 0 selects ifft (conj, fft, conj, *1/n).
 * batch == 0 -> KOFFT_OK (batch() over an empty slice); otherwise n == 0 ->
 * KOFFT_ERR_EMPTY_INPUT; n == 1 is a no-op for both directions. */
int kofft_hip_fft_c32(kofft_hip_ctx *ctx, float *data, size_t n, size_t batch, int inverse);
int kofft_hip_fft_c64(kofft_hip_ctx *ctx, double *data, size_t n, size_t batch, int inverse);
int kofft_hip_fft_c32_dev(kofft_hip_ctx *ctx, float *d_data, size_t n, size_t batch, int inverse);
int kofft_hip_fft_c64_dev(kofft_hip_ctx *ctx, double *d_data, size_t n, size_t batch, int inverse);
/* FftImpl::fft_out_of_place / ifft_out_of_place (fft.rs:469-490), device pointers.
 * d_in and d_out must not partially overlap (d_in == d_out is allowed). */
int kofft_hip_fft_c32_dev_oop(kofft_hip_ctx *ctx, const float *d_in, float *d_out, size_t n,
                              size_t batch, int inverse);
int kofft_hip_fft_c64_dev_oop(kofft_hip_ctx *ctx, const double *d_in, double *d_out, size_t n,
                              size_t batch, int inverse);

/* ScalarFftImpl::fft_radix4 (fft.rs:1455-1548) byte for byte.  kofft's fft_with_strategy(.., FftStrategy::Radix4)
 * (fft.rs:1356) runs it for powers of four, and from n = 16 its output is NOT the DFT (its "bit-reversal for radix-4" loop,
 * fft.rs:1462-1474, flips one bit per base-4 digit instead of reversing the digits).  A drop-in returns the reference's
 * bytes: the host mirrors' fft_with_strategy(.., Radix4) calls these entries BY DEFAULT (round 6); their radix4_compat =
 * false / KOFFT_HIP_RADIX4_COMPAT=0 opts out and gives the true transform for every strategy, like every other entry
 * point here.  The swap loop runs as a gather through its net permutation, butterfly4 in the reference's operation order,
 * the three running-product twiddle sequences of every stage built on the host with Complex::mul (O(n) host work and
 * 12 / 20 bytes of device tables per point, once per (context, n)).  n not a power of four -> fft() (fft.rs:1457-1460;
 * n == 0 -> EMPTY_INPUT); n > 2^26 -> KOFFT_ERR_UNSUPPORTED (the limit of kofft_hip_fft_*).  data: batch * n complex. */
int kofft_hip_fft_radix4_c32(kofft_hip_ctx *ctx, float *data, size_t n, size_t batch);
int kofft_hip_fft_radix4_c64(kofft_hip_ctx *ctx, double *data, size_t n, size_t batch);
int kofft_hip_fft_radix4_c32_dev(kofft_hip_ctx *ctx, const float *d_in, float *d_out, size_t n, size_t batch);
int kofft_hip_fft_radix4_c64_dev(kofft_hip_ctx *ctx, const double *d_in, double *d_out, size_t n, size_t batch);
/* FftPlan::ifft with strategy Radix4 (fft.rs:2040-2055 -> 2037 -> 1356): `c.im = -c.im`, fft_radix4, `c.im = -c.im;
 * c.re * scale; c.im * scale` with scale = 1 / (n as f32 -> T), the conjugations and the scale folded into the first
 * gather and the last stage's store.  Lengths that are not a power of four: ifft()'s arithmetic (fft.rs:1163-1172). */
int kofft_hip_ifft_radix4_c32(kofft_hip_ctx *ctx, float *data, size_t n, size_t batch);
int kofft_hip_ifft_radix4_c64(kofft_hip_ctx *ctx, double *data, size_t n, size_t batch);
int kofft_hip_ifft_radix4_c32_dev(kofft_hip_ctx *ctx, const float *d_in, float *d_out, size_t n, size_t batch);
int kofft_hip_ifft_radix4_c64_dev(kofft_hip_ctx *ctx, const double *d_in, double *d_out, size_t n, size_t batch);

/* FftImpl::fft_strided / ifft_strided (fft.rs:1175-1199, 1236-1260), host pointers:
 * gathers n = scratch_len elements data[i*stride], transforms, scatters back.
 * stride == 0 -> KOFFT_ERR_INVALID_STRIDE; n == 0 -> KOFFT_OK;
 * data_len < (n-1)*stride+1 -> KOFFT_ERR_MISMATCHED_LENGTHS. */
int kofft_hip_fft_c32_strided(kofft_hip_ctx *ctx, float *data, size_t data_len, size_t stride,
                              size_t n, int inverse);
int kofft_hip_fft_c64_strided(kofft_hip_ctx *ctx, double *data, size_t data_len, size_t stride,
                              size_t n, int inverse);

/* ---- real FFT -----------------------------------------------------------------
 * RfftPlanner::rfft_with_scratch -> rfft_direct (rfft.rs:264-282, 425-465) on `batch`
 * contiguous rows of n reals; out: batch * (n/2+1) complex.  `window` (n reals or
 * NULL) is multiplied into each row first -- the framing product of stft.rs:96.
 * n == 0 -> EMPTY_INPUT; odd n -> INVALID_VALUE.  The reference's scratch argument
 * has no counterpart: the post-pass runs out of LDS.
 * irfft: RfftPlanner::irfft_with_scratch -> irfft_direct (rfft.rs:302-320, 468-508);
 * in: batch * (n/2+1) complex, out: batch * n reals. */
int kofft_hip_rfft_f32(kofft_hip_ctx *ctx, const float *in, float *out, const float *window,
                       size_t n, size_t batch);
int kofft_hip_rfft_f32_dev(kofft_hip_ctx *ctx, const float *d_in, float *d_out,
                           const float *d_window, size_t n, size_t batch);
int kofft_hip_irfft_f32(kofft_hip_ctx *ctx, const float *in, float *out, size_t n, size_t batch);
int kofft_hip_irfft_f32_dev(kofft_hip_ctx *ctx, const float *d_in, float *d_out, size_t n,
                            size_t batch);
int kofft_hip_rfft_f64(kofft_hip_ctx *ctx, const double *in, double *out, const double *window,
                       size_t n, size_t batch);
int kofft_hip_rfft_f64_dev(kofft_hip_ctx *ctx, const double *d_in, double *d_out,
                           const double *d_window, size_t n, size_t batch);
int kofft_hip_irfft_f64(kofft_hip_ctx *ctx, const double *in, double *out, size_t n, size_t batch);
int kofft_hip_irfft_f64_dev(kofft_hip_ctx *ctx, const double *d_in, double *d_out, size_t n,
                            size_t batch);

/* ---- STFT ---------------------------------------------------------------------
 * stft::stft (stft.rs:76-105): out = frames * win_len complex, contiguous (the
 * reference's &mut [Vec<Complex32>] flattened).  hop == 0 -> INVALID_HOP_SIZE;
 * frames < ceil(len/hop) -> MISMATCHED_LENGTHS; every provided frame is computed,
 * zero-padded past the end of the signal; win_len == 0 with frames > 0 -> EMPTY_INPUT.
 * The _dev form computes frames [first_frame, first_frame+count) of the same STFT
 * into d_out[0 .. count*win_len) -- the unit one rank owns when frames are sharded
 * across GPUs; it performs stft::parallel's checks only (stft.rs:232-263: hop != 0).
 */
int kofft_hip_stft_f32(kofft_hip_ctx *ctx, const float *signal, size_t len, const float *window,
                       size_t win_len, size_t hop, float *out, size_t frames);
/* stft::parallel (stft.rs:232-263): identical frames, but the only check is hop != 0 --
 * it does not require frames >= ceil(len/hop). */
int kofft_hip_stft_parallel_f32(kofft_hip_ctx *ctx, const float *signal, size_t len,
                                const float *window, size_t win_len, size_t hop, float *out,
                                size_t frames);
/* stft::frame (stft.rs:355-372) and StftStream::next_frame (stft.rs:186-205): the one frame
 * that starts at sample `start` (zero-padded past the end); frame_out: win_len complex. */
int kofft_hip_stft_frame_f32(kofft_hip_ctx *ctx, const float *signal, size_t len,
                             const float *window, size_t win_len, size_t start, float *frame_out);
int kofft_hip_stft_f32_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len,
                           const float *d_window, size_t win_len, size_t hop, float *d_out,
                           size_t first_frame, size_t count);

/* ---- ISTFT (SURVEY 8f "next" row 1) --------------------------------------------
 * stft::istft (stft.rs:117-156): every frame is inverse-transformed IN PLACE (frames_data:
 * frames * win_len complex, contiguous), then overlap-added into `output` (accumulated:
 * the reference does not clear it) with window-square normalisation where the sum exceeds
 * 1e-8; `scratch` receives the window-square sums.  hop == 0 -> INVALID_HOP_SIZE;
 * scratch_len != out_len -> MISMATCHED_LENGTHS.  The per-sample sums run in frame order,
 * exactly as the reference's loop nest does (no atomics). */
int kofft_hip_istft_f32(kofft_hip_ctx *ctx, float *frames_data, size_t frames, const float *window,
                        size_t win_len, size_t hop, float *output, size_t out_len, float *scratch,
                        size_t scratch_len);
int kofft_hip_istft_f32_dev(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window,
                            size_t win_len, size_t hop, float *d_output, size_t out_len,
                            float *d_scratch, size_t scratch_len);
/* stft::inverse_parallel (stft.rs:289-343): the same sums, but the frames are not modified and samples
 * whose window-square sum is <= 1e-8 are set to 0.  Only hop == 0 is rejected. */
int kofft_hip_istft_parallel_f32(kofft_hip_ctx *ctx, const float *frames_data, size_t frames,
                                 const float *window, size_t win_len, size_t hop, float *output,
                                 size_t out_len);
/* stft::inverse_frame (stft.rs:384-399): ifft(frame) in place, then output[start+i] += frame[i].re *
 * window[i] for start+i < out_len; no normalisation. */
int kofft_hip_istft_frame_f32(kofft_hip_ctx *ctx, float *frame, const float *window, size_t win_len,
                              size_t start, float *output, size_t out_len);

/* ---- STFT magnitudes (SURVEY 8f "next" row 2) --------------------------------------
 * visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76): STFT with a Hann window of
 * win_len (window::hann), keeping bins 0 .. win_len/2-1 of every frame as f32 magnitudes
 * sqrt(re*re + im*im), plus the maximum magnitude (0.0 if there are none; NaN never selected).
 * mags: frames * (win_len/2) floats, frames >= ceil(len/hop) (the reference allocates exactly that
 * many).  hop == 0 -> INVALID_HOP_SIZE (the reference would panic dividing by it).  The magnitude
 * is fused into the transform's store: 4x fewer output bytes than stft + a second pass. */
int kofft_hip_stft_magnitudes_f32(kofft_hip_ctx *ctx, const float *samples, size_t len, size_t win_len,
                                  size_t hop, float *mags, size_t frames, float *max_mag);
int kofft_hip_stft_magnitudes_f32_dev(kofft_hip_ctx *ctx, const float *d_samples, size_t len,
                                      size_t win_len, size_t hop, float *d_mags, size_t frames,
                                      float *d_max);

/* ---- 2-D / 3-D FFT (SURVEY 8f "next" row 3) ----------------------------------------
 * ndfft::fft2d_inplace (ndfft.rs:74-101) with depth == 1: FftImpl::fft on every row (length cols), then
 * FftImpl::fft_strided down every column (length rows, stride cols).  ndfft::fft3d_inplace
 * (ndfft.rs:114-155) with depth > 1: z axis (stride rows*cols), y axis (stride cols), x axis (rows).
 * data: depth*rows*cols complex, row-major, in place.  Any zero dimension -> KOFFT_OK (nothing to do).
 * The reference's scratch-length checks belong to the host mirrors (they take the scratch slices).
 * inverse != 0 applies ifft / ifft_strided along the same axes in the same order. */
int kofft_hip_fftnd_c32(kofft_hip_ctx *ctx, float *data, size_t depth, size_t rows, size_t cols, int inverse);
int kofft_hip_fftnd_c64(kofft_hip_ctx *ctx, double *data, size_t depth, size_t rows, size_t cols, int inverse);
int kofft_hip_fftnd_c32_dev(kofft_hip_ctx *ctx, float *d_data, size_t depth, size_t rows, size_t cols, int inverse);
int kofft_hip_fftnd_c64_dev(kofft_hip_ctx *ctx, double *d_data, size_t depth, size_t rows, size_t cols, int inverse);

/* ---- multi-GPU ------------------------------------------------------
 * stft::parallel (stft.rs:232-263) runs rayon over frames; fft::batch (fft.rs:2156-2175) and the row loop around
 * RfftPlanner::rfft_with_scratch (rfft.rs:264-282) run over independent transforms.  The device analogue is ONE process
 * that owns `ngpu` devices (one context + one stream per device): device r works on the contiguous block
 * [r*ceil(U/G), min((r+1)*ceil(U/G), U)) of the U frames / transforms / rows.  STFT: from its own slice of the signal (the
 * slice plus the win_len-hop halo, cut on the host: no halo exchange).  `allgather` != 0 adds BASELINE config #4's exchange:
 * one RCCL ncclAllGather per device (ncclCommInitAll communicators, ncclGroupStart/End, in place) after which every device
 * holds the whole spectrogram in ceil(F/G)-frame slots.  RCCL is bound at run time (dlopen); without it allgather returns
 * KOFFT_ERR_RCCL and everything else still works.
 *
 * HOST-pointer entries return when the result is in the caller's memory.  Each device has its own worker thread that
 * uploads, launches and downloads that device's block (copies from pageable memory block the issuing thread: one thread per
 * device is what lets the G PCIe links run together); only the grouped all-gather is issued by the calling thread.
 * DEVICE-pointer entries (*_dev) take one device pointer per device (arrays of ngpu pointers, entry r valid on device r),
 * enqueue on the per-device streams from the calling thread and return WITHOUT synchronising
 * (kofft_hip_multi_synchronize, or a stream of kofft_hip_multi_context).  The calling thread's current device is left as found.
 *
 * kofft_hip_stft_f32_multi: one call, host pointers in and out; checks = stft::stft's (hop == 0 -> INVALID_HOP_SIZE,
 * frames < ceil(len/hop) -> MISMATCHED_LENGTHS, win_len == 0 -> EMPTY_INPUT), ngpu <= 0 -> INVALID_VALUE.
 * out: frames * win_len complex.  The handle form keeps contexts, buffers, threads and communicators across calls:
 *   kofft_hip_multi_create(ngpu, devices (NULL: 0..ngpu-1), &m)
 *   kofft_hip_multi_stft_f32(m, ..., out (host or NULL), frames, allgather, d_out_per_gpu (NULL or ngpu slots))
 *     d_out_per_gpu[r] receives device r's buffer (owned by m, valid until the next call): its shard, or with allgather
 *     the gathered [G*ceil(F/G), win_len] spectrogram (first `frames` rows are the STFT, the rest zero).
 *   kofft_hip_multi_stft_f32_dev(m, d_signal_per_gpu, len, d_window_per_gpu, win_len, hop, frames, allgather, d_out_per_gpu)
 *     device r holds ITS SLICE of the `len`-sample signal, kofft_hip_multi_stft_slice(m, len, win_len, hop, frames, r, &first,
 *     &count) samples starting at sample `first` (an empty slice may be NULL), and a copy of the window.  d_out_per_gpu is
 *     in/out: a non-NULL entry is the caller's buffer for device r (count_r * win_len complex, or G*ceil(F/G)*win_len with
 *     allgather); a NULL entry is replaced by a buffer owned by m.
 *   kofft_hip_multi_fft_c32 / _c64: fft::batch with the batch in G contiguous blocks, in place, no exchange;
 *   kofft_hip_multi_rfft_f32: rows of n reals -> rows of n/2+1 complex, optional window (BASELINE config #3's shape);
 *   their *_dev twins: d_*_per_gpu[r] points at device r's block of kofft_hip_multi_shard(m, batch, r, ..) rows.
 *   kofft_hip_multi_shard(m, total, rank, &first, &count): the partition above, for callers that place their own data.
 *   kofft_hip_multi_context(m, rank, &ctx, &hip_stream): device r's context and stream (either may be NULL), to order
 *     caller work against the handle's or to call any single-device entry on that device.
 *   kofft_hip_multi_last_timing_ex: the slowest device's time in each phase of the last call, from HIP events on the
 *     per-device streams: upload (host forms), kernel (kernels only -- no copy inside), gather, download (host forms), and
 *     the host forms' wall time from entry to return.  Waits for a *_dev call's events.  Phases a call did not have are 0.
 *   kofft_hip_multi_last_timing (round-2 signature): compute_ms = kernel_ms above, gather_ms. */
typedef struct kofft_hip_multi kofft_hip_multi;
int kofft_hip_multi_create(int ngpu, const int *devices, kofft_hip_multi **out);
int kofft_hip_multi_destroy(kofft_hip_multi *m);
const char *kofft_hip_multi_last_error(const kofft_hip_multi *m);
int kofft_hip_multi_ngpu(const kofft_hip_multi *m);
int kofft_hip_multi_shard(const kofft_hip_multi *m, size_t total, int rank, size_t *first, size_t *count);
int kofft_hip_multi_stft_slice(const kofft_hip_multi *m, size_t len, size_t win_len, size_t hop, size_t frames, int rank,
                               size_t *first_sample, size_t *count);
int kofft_hip_multi_context(const kofft_hip_multi *m, int rank, kofft_hip_ctx **ctx, void **hip_stream);
int kofft_hip_multi_synchronize(kofft_hip_multi *m);
int kofft_hip_multi_last_timing(const kofft_hip_multi *m, float *compute_ms, float *gather_ms);
/* The exchange of the STFT spectra (BASELINE config #4; the device analogue of collecting stft::parallel's frames, stft.rs:232-263)
 * has two forms over the same partition and the same buffers: RCCL (one grouped in-place ncclAllGather per device) and DIRECT
 * (every device pushes its slot to every peer with hipMemcpyPeerAsync on a stream per peer: no RCCL needed, and the direct
 * pattern on point-to-point xGMI whatever RCCL would choose).  Default RCCL, or KOFFT_HIP_MULTI_GATHER=direct at creation;
 * `last` = the form the last call ran (0: it had no exchange). */
#define KOFFT_MULTI_GATHER_RCCL 1
#define KOFFT_MULTI_GATHER_DIRECT 2
int kofft_hip_multi_set_gather(kofft_hip_multi *m, int mode);
int kofft_hip_multi_gather_mode(const kofft_hip_multi *m, int *configured, int *last);
int kofft_hip_multi_last_timing_ex(const kofft_hip_multi *m, float *upload_ms, float *kernel_ms, float *gather_ms,
                                   float *download_ms, float *wall_ms);
int kofft_hip_multi_stft_f32(kofft_hip_multi *m, const float *signal, size_t len, const float *window, size_t win_len,
                             size_t hop, float *out, size_t frames, int allgather, float **d_out_per_gpu);
int kofft_hip_multi_stft_f32_dev(kofft_hip_multi *m, const float *const *d_signal_per_gpu, size_t len,
                                 const float *const *d_window_per_gpu, size_t win_len, size_t hop, size_t frames, int allgather,
                                 float **d_out_per_gpu);
int kofft_hip_stft_f32_multi(int ngpu, const float *signal, size_t len, const float *window, size_t win_len,
                             size_t hop, float *out, size_t frames, int allgather);
int kofft_hip_multi_fft_c32(kofft_hip_multi *m, float *data, size_t n, size_t batch, int inverse);
int kofft_hip_multi_fft_c64(kofft_hip_multi *m, double *data, size_t n, size_t batch, int inverse);
int kofft_hip_multi_rfft_f32(kofft_hip_multi *m, const float *in, float *out, const float *window, size_t n, size_t batch);
int kofft_hip_multi_fft_c32_dev(kofft_hip_multi *m, float *const *d_data_per_gpu, size_t n, size_t batch, int inverse);
int kofft_hip_multi_fft_c64_dev(kofft_hip_multi *m, double *const *d_data_per_gpu, size_t n, size_t batch, int inverse);
int kofft_hip_multi_rfft_f32_dev(kofft_hip_multi *m, const float *const *d_in_per_gpu, float *const *d_out_per_gpu,
                                 const float *const *d_window_per_gpu, size_t n, size_t batch);

#ifdef __cplusplus
}
#endif
#endif /* KOFFT_HIP_H */
