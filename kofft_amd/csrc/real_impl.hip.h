// real_impl.hip.h -- rfft_direct / irfft_direct (rfft.rs:425-508) on device pointers for one element type.
#pragma once

#include "host_common.hip.h"

namespace kofft {
namespace host {

// ---- composed form for inner lengths the fused kernels do not cover ---------------------------------------------------
// rfft_direct calls fft.fft(&mut output[..m]) for ANY m (rfft.rs:447; Bluestein arm fft.rs:1083-1132, large powers of
// two through the factor path), so the device does too: [window product] -> the m-point complex transform of fft_dev
// -> the post-pass of rfft.rs:450-463 as its own kernel.  Same operations per element as the fused kernels.
template <typename T>
__global__ __launch_bounds__(256) void real_window_kernel(const T *__restrict__ in, const T *__restrict__ window, T *__restrict__ z,
                                                          const size_t n, const size_t total)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) z[i] = in[i] * window[i % n];  // the framing product of stft.rs:96, element by element
}

template <typename T>
__global__ __launch_bounds__(256) void rfft_post_kernel(const cpx<T> *__restrict__ y, const cpx<T> *__restrict__ rtab,
                                                        cpx<T> *__restrict__ out, const size_t m, const size_t total /* batch * (m+1) */)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / (m + 1), k = i % (m + 1);
    const cpx<T> *yr = y + b * m;
    cpx<T> x;
    if (k == 0 || k == m) {  // rfft.rs:450-452
        const cpx<T> y0 = yr[0];
        x = (k == 0) ? mk<T>(y0.re + y0.im, T(0)) : mk<T>(y0.re - y0.im, T(0));
    } else {                 // rfft.rs:454-463
        const T half = T(0.5f);
        const cpx<T> a = yr[k], ymk = yr[m - k], w = rtab[k];
        const cpx<T> bb = mk<T>(ymk.re, -ymk.im);
        const cpx<T> sum = cadd(a, bb), diff = csub(a, bb);
        const cpx<T> t = cmul(w, diff);
        const cpx<T> temp = cadd(sum, mk<T>(t.im, -t.re));
        x = mk<T>(temp.re * half, temp.im * half);
    }
    out[i] = x;
}

template <typename T>
__global__ __launch_bounds__(256) void irfft_pre_kernel(const cpx<T> *__restrict__ in, const cpx<T> *__restrict__ rtab,
                                                        cpx<T> *__restrict__ scratch, const size_t m, const size_t total /* batch * m */)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / m, k = i % m;
    const cpx<T> *row = in + b * (m + 1);
    const T half = T(0.5f);
    cpx<T> s;
    if (k == 0) {  // rfft.rs:491-493
        s = mk<T>((row[0].re + row[m].re) * half, (row[0].re - row[m].re) * half);
    } else {       // rfft.rs:495-503
        const cpx<T> a = row[k], rb = row[m - k], tw = rtab[k];
        const cpx<T> bb = mk<T>(rb.re, -rb.im);
        const cpx<T> sum = cadd(a, bb), diff = csub(a, bb);
        const cpx<T> w = mk<T>(tw.re, -tw.im);
        const cpx<T> t = cmul(w, diff);
        const cpx<T> temp = csub(sum, mk<T>(t.im, -t.re));
        s = mk<T>(temp.re * half, temp.im * half);
    }
    scratch[i] = s;
}

inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

template <typename T>
int rfft_composed_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, const T *d_window, size_t n, size_t batch)
{
    const size_t m = n / 2;
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    size_t chunk = (size_t(512) << 20) / (n * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if ((chunk * (m + 1) + 255) / 256 > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    // Powers of two beyond the single-workgroup sizes (round 3): the window product rides on the factor path's first load
    // (BigColsIO PRE_WINDOW) -- one pass over the input less.  Other lengths (Bluestein's inner transform) keep the kernel.
    const bool fuse_window = d_window && is_pow2(m) && m > (size_t(1) << max_log2<T>()) && ctx->blue_fused;
    // scratch: [windowed input (only with an unfused window)] [Y]
    const size_t zbytes = (d_window && !fuse_window) ? chunk * n * sizeof(T) : 0, ybytes = chunk * m * sizeof(cpx<T>);
    rc = ensure_real_tmp(ctx, zbytes + ybytes);
    if (rc) return rc;
    T *z = static_cast<T *>(ctx->real_tmp);
    T *y = reinterpret_cast<T *>(static_cast<char *>(ctx->real_tmp) + zbytes);
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        const T *src = d_in + b0 * n;
        if (fuse_window) {
            rc = fft_big_windowed_dev<T>(ctx, src, y, d_window, m, nb);
            if (rc) return rc;
        } else if (d_window) {
            hipLaunchKernelGGL(real_window_kernel<T>, dim3(blocks_for(nb * n)), dim3(256), 0, ctx->stream, src, d_window, z, n, nb * n);
            KOFFT_HIP_TRY(ctx, hipGetLastError());
            src = z;
        }
        // z[i] = (x[2i], x[2i+1]) (rfft.rs:444-446) is the row itself read as m complex values
        if (!fuse_window) {
            rc = fft_dev<T>(ctx, src, y, m, nb, 0);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(rfft_post_kernel<T>, dim3(blocks_for(nb * (m + 1))), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const cpx<T> *>(y), rtab, reinterpret_cast<cpx<T> *>(d_out) + b0 * (m + 1), m, nb * (m + 1));
        KOFFT_HIP_TRY(ctx, hipGetLastError());
    }
    return KOFFT_OK;
}

template <typename T>
int irfft_composed_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch)
{
    const size_t m = n / 2;
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    size_t chunk = (size_t(512) << 20) / (n * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if ((chunk * m + 255) / 256 > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    rc = ensure_real_tmp(ctx, chunk * m * sizeof(cpx<T>));
    if (rc) return rc;
    cpx<T> *scratch = static_cast<cpx<T> *>(ctx->real_tmp);
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        hipLaunchKernelGGL(irfft_pre_kernel<T>, dim3(blocks_for(nb * m)), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const cpx<T> *>(d_in) + b0 * (m + 1), rtab, scratch, m, nb * m);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
        // fft.ifft(&mut scratch[..m]) (rfft.rs:504), then output[2i], output[2i+1] = scratch[i].re, .im: the rows of d_out
        rc = fft_dev<T>(ctx, reinterpret_cast<const T *>(scratch), d_out + b0 * n, m, nb, 1);
        if (rc) return rc;
    }
    return KOFFT_OK;
}

template <typename T>
int rfft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, const T *d_window, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;   // rfft.rs:434
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;  // rfft.rs:437
    const size_t m = n / 2;
    if (!complex_len_ok(m)) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!fused_len_ok<T>(m)) return rfft_composed_dev<T>(ctx, d_in, d_out, d_window, n, batch);
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    RfftIO<T> io{{}, d_in, d_window, reinterpret_cast<cpx<T> *>(d_out), rtab, (int)m};
    return dispatch<T, EPI_RFFT>(ctx, io, m, batch);
}

template <typename T>
int irfft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;   // rfft.rs:477
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;  // rfft.rs:480
    const size_t m = n / 2;
    if (!complex_len_ok(m)) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!fused_len_ok<T>(m)) return irfft_composed_dev<T>(ctx, d_in, d_out, n, batch);
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    // threads per transform of the persistent kernels (m/16; m/8 up to m = 512)
    const int tpt = (int)(m <= 64 ? m / 4 : m <= 512 ? m / 8 : m / 16);
    IrfftIO<T> io{{}, reinterpret_cast<const cpx<T> *>(d_in), reinterpret_cast<cpx<T> *>(d_out), rtab, (int)m,
                  (T)1 / (T)(float)m, tpt};
    return dispatch<T, EPI_STORE>(ctx, io, m, batch);
}


}  // namespace host
}  // namespace kofft
