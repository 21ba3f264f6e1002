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

// Post- / pre-pass of the composed route.  Bins k and m - k are computed from the same two inputs (Y[k], Y[m-k]), so one thread does both:
// every input is read once (one thread per OUTPUT read each of them twice, the second time from another XCD's workgroup -- 1.5x the
// pass's bytes: rfft 65536 0.30 -> 0.2x ms for this pass).  grid.x covers j = 0 .. m/2, grid.y walks the rows.  Per element the
// expressions are those of the fused kernels' epilogues (rfft.rs:450-463, 491-503).
template <typename T>
__device__ __forceinline__ cpx<T> rfft_post_one(const cpx<T> a, const cpx<T> ymk, const cpx<T> w)
{
    const T half = T(0.5f);
    const cpx<T> bb = mk<T>(ymk.re, -ymk.im);
    const cpx<T> sum = cadd(a, bb), diff = csub(a, bb);
    const cpx<T> t = cmul(w, diff);
    const cpx<T> temp = cadd(sum, mk<T>(t.im, -t.re));
    return mk<T>(temp.re * half, temp.im * half);
}
template <typename T>
__global__ __launch_bounds__(256) void rfft_post_kernel(const cpx<T> *__restrict__ y, const cpx<T> *__restrict__ rtab,
                                                        cpx<T> *__restrict__ out, const size_t m, const size_t rows)
{
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j > m / 2) return;
    for (size_t b = blockIdx.y; b < rows; b += gridDim.y) {
        const cpx<T> *yr = y + b * m;
        cpx<T> *xr = out + b * (m + 1);
        // (round 6: the outputs are written once and not read back, the intermediate is read once -- streaming stores and loads: same box, three
        // rounds, rfft32 2^17 .. 2^20 +1.9 / +2.4 / +8.4 / +1.8 %, rfft64 +4.8 / +8.7 / +4.7 / +4.0 %)
#define KOFFT_POST_ST(p, v) st_stream((p), (v))
        if (j == 0) {  // rfft.rs:450-452
            const cpx<T> y0 = yr[0];
            KOFFT_POST_ST(&xr[0], mk<T>(y0.re + y0.im, T(0)));
            KOFFT_POST_ST(&xr[m], mk<T>(y0.re - y0.im, T(0)));
        } else {       // rfft.rs:454-463, for k = j and k = m - j (m odd never pairs a bin with itself; m even: j = m/2 does)
            const cpx<T> a = ld_stream(yr + j), c = ld_stream(yr + (m - j));  // (the intermediate: read once)
            KOFFT_POST_ST(&xr[j], rfft_post_one<T>(a, c, rtab[j]));
            if (m - j != j) KOFFT_POST_ST(&xr[m - j], rfft_post_one<T>(c, a, rtab[m - j]));
        }
#undef KOFFT_POST_ST
    }
}

template <typename T>
__global__ __launch_bounds__(256) void irfft_pre_kernel(const cpx<T> *__restrict__ in, const cpx<T> *__restrict__ rtab,
                                                        cpx<T> *__restrict__ scratch, const size_t m, const size_t rows)
{
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j > m / 2) return;
    for (size_t b = blockIdx.y; b < rows; b += gridDim.y) {
        const cpx<T> *row = in + b * (m + 1);
        cpx<T> *sr = scratch + b * m;
        if (j == 0) {  // rfft.rs:491-493
            const T half = T(0.5f);
            const T r0 = row[0].re, rm = row[m].re;
            sr[0] = mk<T>((r0 + rm) * half, (r0 - rm) * half);
        } else {       // rfft.rs:495-503, for k = j and k = m - j
            const cpx<T> a = row[j], c = row[m - j];  // (streaming loads of the caller's bins measured +-1 %: plain)
            sr[j] = irfft_pre_one<T>(a, c, rtab[j]);
            if (m - j != j) sr[m - j] = irfft_pre_one<T>(c, a, rtab[m - j]);
        }
    }
}

// Short rows (m < 512: a 256-thread block per row would idle): one thread per output element over the whole chunk.
template <typename T>
__global__ __launch_bounds__(256) void rfft_post_flat_kernel(const cpx<T> *__restrict__ y, const cpx<T> *__restrict__ rtab,
                                                             cpx<T> *__restrict__ out, const size_t m, const size_t total /* rows * (m+1) */)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / (m + 1), k = i % (m + 1);
    const cpx<T> *yr = y + b * m;
    if (k == 0 || k == m) {  // rfft.rs:450-452
        const cpx<T> y0 = yr[0];
        out[i] = (k == 0) ? mk<T>(y0.re + y0.im, T(0)) : mk<T>(y0.re - y0.im, T(0));
    } else {
        out[i] = rfft_post_one<T>(yr[k], yr[m - k], rtab[k]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void irfft_pre_flat_kernel(const cpx<T> *__restrict__ in, const cpx<T> *__restrict__ rtab,
                                                             cpx<T> *__restrict__ scratch, const size_t m, const size_t total /* rows * m */)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / m, k = i % m;
    const cpx<T> *row = in + b * (m + 1);
    if (k == 0) {  // rfft.rs:491-493
        const T half = T(0.5f);
        scratch[i] = mk<T>((row[0].re + row[m].re) * half, (row[0].re - row[m].re) * half);
    } else {
        scratch[i] = irfft_pre_one<T>(row[k], row[m - k], rtab[k]);
    }
}

inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

template <typename T>
int rfft_composed_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, const T *d_window, size_t n, size_t batch)
{
    const size_t m = n / 2;
    // (Round 5: the last factor on mirrored tile PAIRS -- rows K and 2^LA - K in one workgroup -- with the post-pass in its epilogue, two
    // passes instead of three, was built, bit-exact on every row, and measured SLOWER: 453-678 us for the fused kernel against 185-227 us
    // (last factor on row pairs) + 217-227 us (post-pass kernel) at n = 2^17 .. 2^20 -- the mirror of a line-aligned tile starts 8 bytes into
    // a line, and its partial-line stores cost more than the pass they save; commit 2191bc6 has it, DESIGN 10 item 4 the counters.)
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    size_t chunk = (size_t(512) << 20) / (n * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if ((chunk * (m + 1) + 255) / 256 > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    // Powers of two beyond the single-workgroup sizes (round 3): the window product rides on the factor path's first load
    // (BigColsIO PRE_WINDOW) -- one pass over the input less.  Other lengths (Bluestein's inner transform) keep the kernel.
    // m = 2^15 (f32) / 2^14 (f64): the register-file kernel with the window on its loads (fft_regfile.hip.h: RowWindowIO) -- one pass where
    // the factor path takes two
    // (a chunk with fewer than two transforms per CU -- a short last one -- takes the route below it)
    const bool regfile_window = d_window && m == (size_t(2) << max_log2<T>()) && ctx->use_regfile;
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
        if (m == (size_t(2) << max_log2<T>()) && ctx->use_regfile && ctx->rfft_regfile_epi && nb >= (size_t)ctx->num_cus * 2) {
            // Round 6: m = 2^15 (f32) / 2^14 (f64) in ONE pass -- the register-file kernel with the post-pass as its epilogue (fft_regfile.hip.h:
            // RfftRowIO; Y[k] and Y[m - k] are in the same workgroup), the window on its loads: no intermediate, no rfft_post_kernel launch
            const cpx<T> *tw = nullptr;
            rc = get_table<T>(ctx, Kind<T>::tw, m, &tw);
            if (rc) return rc;
            constexpr int RLA = sizeof(T) == 4 ? 8 : 7, RLB = 7, RQB0 = 4;  // as fft_dev routes the plain transform
            cpx<T> *dst = reinterpret_cast<cpx<T> *>(d_out) + b0 * (m + 1);
            if (d_window) {
                RfftRowIO<T, true> io{{{}, reinterpret_cast<const cpx<T> *>(src), dst, (int)m, (T)1}, reinterpret_cast<const cpx<T> *>(d_window), rtab};
                rc = launch_regfile<T, RLA, RLB, RQB0>(ctx, io, tw, nb);
            } else {
                RfftRowIO<T, false> io{{{}, reinterpret_cast<const cpx<T> *>(src), dst, (int)m, (T)1}, nullptr, rtab};
                rc = launch_regfile<T, RLA, RLB, RQB0>(ctx, io, tw, nb);
            }
            if (rc) return rc;
            continue;
        }
        if (regfile_window && nb >= (size_t)ctx->num_cus * 2) {
            const cpx<T> *tw = nullptr;
            rc = get_table<T>(ctx, Kind<T>::tw, m, &tw);
            if (rc) return rc;
            constexpr int RLA = sizeof(T) == 4 ? 8 : 7, RLB = 7, RQB0 = 4;  // as fft_dev routes the plain transform
            RowWindowIO<T> io{{{}, reinterpret_cast<const cpx<T> *>(src), reinterpret_cast<cpx<T> *>(y), (int)m, (T)1}, reinterpret_cast<const cpx<T> *>(d_window)};
            rc = launch_regfile<T, RLA, RLB, RQB0>(ctx, io, tw, nb);
            if (rc) return rc;
        } else if (fuse_window) {
            rc = fft_big_windowed_dev<T>(ctx, src, y, d_window, m, nb);
            if (rc) return rc;
        } else if (d_window) {
            hipLaunchKernelGGL(real_window_kernel<T>, dim3(blocks_for(nb * n)), dim3(256), 0, ctx->stream, src, d_window, z, n, nb * n);
            KOFFT_HIP_TRY(ctx, hipGetLastError());
            src = z;
        }
        // z[i] = (x[2i], x[2i+1]) (rfft.rs:444-446) is the row itself read as m complex values
        if (!fuse_window && !(regfile_window && nb >= (size_t)ctx->num_cus * 2)) {
            rc = fft_dev<T>(ctx, src, y, m, nb, 0);
            if (rc) return rc;
        }
        if (m >= 512)
            hipLaunchKernelGGL(rfft_post_kernel<T>, dim3(blocks_for(m / 2 + 1), (unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const cpx<T> *>(y), rtab, reinterpret_cast<cpx<T> *>(d_out) + b0 * (m + 1), m, nb);
        else
            hipLaunchKernelGGL(rfft_post_flat_kernel<T>, dim3(blocks_for(nb * (m + 1))), dim3(256), 0, ctx->stream,
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
        if (m == (size_t(2) << max_log2<T>()) && ctx->use_regfile && nb >= (size_t)ctx->num_cus * 2) {
            // m = 2^15 (f32) / 2^14 (f64): the pre-pass on the register-file kernel's loads (fft_regfile.hip.h: IrfftRowIO) -- one pass instead of two
            const cpx<T> *tw = nullptr;
            rc = get_table<T>(ctx, Kind<T>::tw, m, &tw);
            if (rc) return rc;
            constexpr int RLA = sizeof(T) == 4 ? 8 : 7, RLB = 7, RQB0 = 4;  // as fft_dev routes the plain transform
            const T scale = (T)1 / (T)(float)m;                              // fft.rs:1167
            IrfftRowIO<T> io{{{}, reinterpret_cast<const cpx<T> *>(d_in) + b0 * (m + 1), reinterpret_cast<cpx<T> *>(d_out + b0 * n), (int)m, scale}, rtab};
            rc = launch_regfile<T, RLA, RLB, RQB0>(ctx, io, tw, nb);
            if (rc) return rc;
            continue;
        }
        if (m >= 512)
            hipLaunchKernelGGL(irfft_pre_kernel<T>, dim3(blocks_for(m / 2 + 1), (unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const cpx<T> *>(d_in) + b0 * (m + 1), rtab, scratch, m, nb);
        else
            hipLaunchKernelGGL(irfft_pre_flat_kernel<T>, dim3(blocks_for(nb * m)), dim3(256), 0, ctx->stream,
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
