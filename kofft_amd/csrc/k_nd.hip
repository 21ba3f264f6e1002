// k_nd.hip -- ndfft::fft2d_inplace / fft3d_inplace (ndfft.rs:74-155) on device pointers.
#include "host_common.hip.h"

namespace kofft {
namespace host {

// ndfft::fft2d_inplace / fft3d_inplace (ndfft.rs:74-155): one axis at a time, every line of the axis in one launch.
template <typename T>
int fft_axis_dev(kofft_hip_ctx *ctx, T *d_data, size_t len, size_t lines, size_t inner, size_t outer_stride, size_t stride,
                 int inverse)
{
    if (len <= 1 || lines == 0) return KOFFT_OK;  // fft of one element: nothing to do (fft.rs:1059)
    // Long axes: the strided kernel can hold only 2 or 1 adjacent lines of 2^12 / 2^13 points in LDS, i.e. 16- and 8-byte
    // segments (8 lines at 2^10 still beat this route, 4 lines at 2^11 tie with it).  Instead: transpose a panel of lines into contiguous rows, run the batched (persistent) kernels
    // on it, transpose back -- four coalesced passes instead of two scattered ones (4096 x 4096 c32: 0.45 -> 0.2x ms).
    // Every line still goes through the same 1-D transform, so the results are unchanged.
    // ... and the only route for axis lengths the strided kernel does not cover (non-powers of two: Bluestein; beyond 2^14)
    // Power-of-two axes of 2^12 .. 2^14 points (2^13 in f64) over a power-of-two number of adjacent lines: two column-tile passes
    // with the axis's own table instead of transpose -> rows -> transpose (round 4: one pass over the data less; AxisLastIO).
    // c32 axes of 2^11 points too (the strided kernel holds 4 adjacent lines: 32-byte segments) up to 8192 adjacent lines: 2048 x 8192
    // 0.229 -> 0.179 ms, 2048 x 2048 0.059 -> 0.047, 8 x 2048 x 2048 0.497 -> 0.451; over 16384 adjacent lines (2048 x 256 x 64) it loses 10 %,
    // c64 (64-byte segments in the strided kernel) 3 %, and 2^10-point axes lose everywhere (tools/bench_nd.py).
    if (stride == inner && ctx->nd_two_pass && fused_len_ok<T>(len) && (len >= 4096 || (len == 2048 && sizeof(T) == 4 && inner <= 8192)) && is_pow2(inner) &&
        inner >= (size_t)KOFFT_BIG_XPB(T) && lines % inner == 0 && (lines == inner || outer_stride == len * inner) &&
        lines * len * sizeof(cpx<T>) >= (size_t(16) << 20) && len * inner * sizeof(cpx<T>) <= (size_t(1) << 31)) {  // (32-bit buffer descriptors per block)
        return fft_axis2_dev<T>(ctx, d_data, ilog2(len), ilog2(inner), lines / inner, inverse);
    }
    if (stride == inner && (!fused_len_ok<T>(len) || (ctx->nd_transpose && len >= (size_t)ctx->nd_transpose_min && lines * len * sizeof(cpx<T>) >= (size_t(16) << 20)))) {
        const size_t outer = lines / inner;  // dense [len][inner] blocks, outer_stride apart
        const size_t cap = size_t(1) << 30, col_bytes = len * sizeof(cpx<T>);
        size_t P = cap / col_bytes;
        if (P > inner) P = inner;
        if (P >= 32) P &= ~size_t(31);
        if (P == 0) P = 1;
        size_t OG = cap / (P * col_bytes);
        if (OG < 1) OG = 1;
        if (OG > outer) OG = outer;
        if (OG > 65535) OG = 65535;
        const size_t need = OG * P * col_bytes;
        // the panel lives in its own scratch: fft_dev below may use the factor path's intermediate (n > 2^14) or the
        // Bluestein work buffer (other lengths)
        {
            const int prc = ensure_real_tmp(ctx, need);
            if (prc) return prc;
        }
        cpx<T> *panel = static_cast<cpx<T> *>(ctx->real_tmp);
        cpx<T> *data = reinterpret_cast<cpx<T> *>(d_data);
        for (size_t o0 = 0; o0 < outer; o0 += OG) {
            const size_t og = (outer - o0 < OG) ? outer - o0 : OG;
            for (size_t p0 = 0; p0 < inner; p0 += P) {
                const size_t pw = (inner - p0 < P) ? inner - p0 : P;
                cpx<T> *blk = data + o0 * outer_stride + p0;
                // (tile grid flattened onto grid.x: pw * len <= 2^27 elements bounds the product, not either side)
                const unsigned pt = (unsigned)((pw + 31) / 32), lt = (unsigned)((len + 31) / 32);
                dim3 g1(pt * lt, (unsigned)og);
                hipLaunchKernelGGL(transpose_kernel<T>, g1, dim3(256), 0, ctx->stream, blk, panel, len, pw, inner, len, outer_stride,
                                   pw * len, pt);
                KOFFT_HIP_TRY(ctx, hipGetLastError());
                int rc = fft_dev<T>(ctx, reinterpret_cast<T *>(panel), reinterpret_cast<T *>(panel), len, og * pw, inverse);
                if (rc) return rc;
                dim3 g2(lt * pt, (unsigned)og);
                hipLaunchKernelGGL(transpose_kernel<T>, g2, dim3(256), 0, ctx->stream, panel, blk, pw, len, len, inner, pw * len,
                                   outer_stride, lt);
                KOFFT_HIP_TRY(ctx, hipGetLastError());
            }
        }
        return KOFFT_OK;
    }
    const T scale = (T)1 / (T)(float)len;
    if (inverse) {
        StridedIO<T, true> io{reinterpret_cast<cpx<T> *>(d_data), inner, outer_stride, stride, scale};
        return dispatch<T, EPI_STORE>(ctx, io, len, lines);
    }
    StridedIO<T, false> io{reinterpret_cast<cpx<T> *>(d_data), inner, outer_stride, stride, scale};
    return dispatch<T, EPI_STORE>(ctx, io, len, lines);
}

template <typename T>
int fft_nd_dev(kofft_hip_ctx *ctx, T *d_data, size_t depth, size_t rows, size_t cols, int inverse)
{
    if (depth == 0 || rows == 0 || cols == 0) return KOFFT_OK;  // ndfft.rs:84-86, 124-126
    for (size_t n : {depth, rows, cols})
        if (!complex_len_ok(n)) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_data) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if (depth > 1) {  // z axis first (ndfft.rs:131-137): lines (r, c), stride rows*cols
        rc = fft_axis_dev<T>(ctx, d_data, depth, rows * cols, rows * cols, 0, rows * cols, inverse);
        if (rc) return rc;
        // y axis (ndfft.rs:138-144): lines (d, c), stride cols
        rc = fft_axis_dev<T>(ctx, d_data, rows, depth * cols, cols, rows * cols, cols, inverse);
        if (rc) return rc;
        // x axis (ndfft.rs:145-151): contiguous rows
        return fft_dev<T>(ctx, d_data, d_data, cols, depth * rows, inverse);
    }
    // 2-D (ndfft.rs:89-98): rows first, then columns
    if constexpr (sizeof(T) == 4) {
        // c32, rows of 1024 .. 4096 points, 1024 .. 4096 of them: rows + the columns' first two stages fused, then ONE column-tile pass
        if (fft2d_fused_ok(ctx, rows, cols)) return fft2d_fused_c32(ctx, reinterpret_cast<float *>(d_data), rows, cols, inverse);
    }
    rc = fft_dev<T>(ctx, d_data, d_data, cols, rows, inverse);
    if (rc) return rc;
    return fft_axis_dev<T>(ctx, d_data, rows, cols, cols, 0, cols, inverse);
}


template int fft_nd_dev<float>(kofft_hip_ctx *, float *, size_t, size_t, size_t, int);
template int fft_nd_dev<double>(kofft_hip_ctx *, double *, size_t, size_t, size_t, int);

}  // namespace host
}  // namespace kofft
