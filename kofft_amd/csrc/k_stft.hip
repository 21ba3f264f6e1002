// k_stft.hip -- STFT framing kernels (stft.rs:76-105), the ISTFT overlap-add (stft.rs:117-156, 289-343, 384-399) and
// stft_magnitudes (visual/spectrogram.rs:52-76) on device pointers.
#include "host_common.hip.h"

namespace kofft {
namespace host {

int stft_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len, const float *d_window, size_t win_len,
             size_t start0, size_t hop, float *d_out, size_t count)
{
    if (count == 0) return KOFFT_OK;
    if (win_len == 0) return KOFFT_ERR_EMPTY_INPUT;  // fft.fft(&mut []) -> fft.rs:1056
    if (!is_pow2(win_len) || win_len > (size_t(1) << max_log2<float>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || (!d_signal && len) || !d_window || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    StftIO io{{}, d_signal, d_window, reinterpret_cast<cpx<float> *>(d_out), len, hop, start0, (int)win_len};
    return dispatch<float, EPI_STORE>(ctx, io, win_len, count);
}

// stft::istft (stft.rs:117-156, mode 1), stft::inverse_parallel (stft.rs:289-343, mode 2), stft::inverse_frame
// (stft.rs:384-399, mode 0): ifft every frame in place, then the ordered overlap-add kernel.
int istft_dev(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window, size_t win_len, size_t hop,
              float *d_output, size_t out_len, float *d_scratch, size_t scratch_len, int mode, size_t start0)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;                               // stft.rs:125 / 299
    if (mode == 1 && scratch_len != out_len) return KOFFT_ERR_MISMATCHED_LENGTHS;  // stft.rs:128
    if (frames > 0 && win_len == 0) return KOFFT_ERR_EMPTY_INPUT;                  // fft.ifft(&mut []) -> fft.rs:1136
    if (frames > 0 && (!is_pow2(win_len) || win_len > (size_t(1) << max_log2<float>()))) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || (frames && (!d_frames || !d_window)) || (out_len && (!d_output || (mode != 0 && !d_scratch)))) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (frames > 0) {
        int rc = fft_dev<float>(ctx, d_frames, d_frames, win_len, frames, 1);
        if (rc) return rc;
    }
    if (out_len > 0) {
        const size_t blocks = (out_len + 255) / 256;
        if (blocks > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
        const cpx<float> *fr = reinterpret_cast<const cpx<float> *>(d_frames);
        if (mode == 0)
            hipLaunchKernelGGL(istft_ola_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, fr, d_window, d_output,
                               d_scratch, frames, win_len, hop, out_len, start0);
        else if (mode == 2)
            hipLaunchKernelGGL(istft_ola_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, fr, d_window, d_output,
                               d_scratch, frames, win_len, hop, out_len, start0);
        else
            hipLaunchKernelGGL(istft_ola_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, fr, d_window, d_output,
                               d_scratch, frames, win_len, hop, out_len, start0);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
    }
    return KOFFT_OK;
}

// visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76): hann(win_len) window, frames x win_len/2 magnitudes
// and their maximum.  d_max receives one float.
int stft_mag_dev(kofft_hip_ctx *ctx, const float *d_samples, size_t len, size_t win_len, size_t hop, float *d_mags,
                 size_t frames, float *d_max)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;  // the reference divides by hop (div_ceil) and would panic
    const size_t required = (len + hop - 1) / hop;
    if (frames < required) return KOFFT_ERR_MISMATCHED_LENGTHS;
    if (frames > 0 && win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (frames > 0 && (!is_pow2(win_len) || win_len > (size_t(1) << max_log2<float>()))) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_max || (frames && (!d_mags || (!d_samples && len)))) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    KOFFT_HIP_TRY(ctx, hipMemsetAsync(d_max, 0, sizeof(float), ctx->stream));  // max_mag starts at 0.0
    if (frames == 0) return KOFFT_OK;
    // hann(win_len), cached per context like a planner table (kind 4)
    const float *d_win = nullptr;
    {
        auto key = std::make_pair(4, win_len);
        auto it = ctx->tables.find(key);
        if (it == ctx->tables.end()) {
            std::vector<float> w(win_len);
            kofft_tables::hann_f32(win_len, w.data());
            void *d = nullptr;
            KOFFT_HIP_TRY(ctx, hipMalloc(&d, win_len * sizeof(float)));
            KOFFT_HIP_TRY(ctx, hipMemcpy(d, w.data(), win_len * sizeof(float), hipMemcpyHostToDevice));
            ctx->tables[key] = d;
            d_win = static_cast<const float *>(d);
        } else {
            d_win = static_cast<const float *>(it->second);
        }
    }
    // one launch: the magnitudes are stored and their maximum reduced by the same kernel (StftMagIO::acc_finish)
    StftMagIO io{{{}, d_samples, d_win, nullptr, len, hop, 0, (int)win_len}, d_mags, reinterpret_cast<unsigned *>(d_max)};
    int rc = dispatch<float, EPI_STORE>(ctx, io, win_len, frames);
    if (rc) return rc;
    return KOFFT_OK;
}


}  // namespace host
}  // namespace kofft
