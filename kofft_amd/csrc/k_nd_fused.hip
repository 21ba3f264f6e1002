// k_nd_fused.hip -- ndfft::fft2d_inplace (ndfft.rs:74-101) for c32 images with 1024 .. 4096-point rows: two passes over the image
// (fft_nd_fused.hip.h).  Its own translation unit so that `make -j` builds it beside the others.
#include "host_common.hip.h"
#include "complex_impl.hip.h"
#include "fft_nd_fused.hip.h"

namespace kofft {
namespace host {

// the column-tile kernels live in k_complex_f32.o (explicit instantiations there): one copy in the library, not one per translation unit
#define KOFFT_CASE(LL)                                                                                                              \
    extern template int launch_tile_persist<float, LL, AxisLastIO<float, false>>(kofft_hip_ctx *, const AxisLastIO<float, false> &, \
                                                                                 const cpx<float> *, size_t);                      \
    extern template int launch_tile_persist<float, LL, AxisLastIO<float, true>>(kofft_hip_ctx *, const AxisLastIO<float, true> &,   \
                                                                                const cpx<float> *, size_t);
KOFFT_CASE(8)
KOFFT_CASE(9)
KOFFT_CASE(10)
KOFFT_CASE(11)
#undef KOFFT_CASE

// fft2d_inplace of c32 images with 1024 / 2048 / 4096-point rows in TWO passes (fft_nd_fused.hip.h): rows + the columns' first two
// stages into the scratch, the columns' remaining stages (frequency prefix K = 2 bits) back into the image.
template <int LC, bool INVERSE>
int fft2d_fused_core(kofft_hip_ctx *ctx, cpx<float> *data, int LT)
{
    const size_t rows = size_t(1) << LT, cols = size_t(1) << LC, elems = rows * cols;
    const cpx<float> *tw_row = nullptr, *tw_col = nullptr;
    int rc = get_table<float>(ctx, Kind<float>::tw, cols, &tw_row);
    if (rc) return rc;
    rc = get_table<float>(ctx, Kind<float>::tw, rows, &tw_col);
    if (rc) return rc;
    rc = ensure_real_tmp(ctx, elems * sizeof(cpx<float>));
    if (rc) return rc;
    cpx<float> *mid = static_cast<cpx<float> *>(ctx->real_tmp);
    {
        constexpr int BLOCK = (1 << LC) / 16;
        constexpr int WG_PER_CU = 512 / BLOCK;  // two wavefronts per SIMD: 256 registers each (244 .. 256 used, nothing spilled)
        constexpr size_t lds = (size_t)persist_slot_elems(LC) * sizeof(cpx<float>);
        static_assert(lds * WG_PER_CU <= 160 * 1024, "LDS budget");
        auto kern = fft2d_rows4_kernel<LC, INVERSE>;
        const size_t groups = rows / 4;
        size_t blocks = (size_t)ctx->num_cus * WG_PER_CU;
        if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
        if (blocks < 1) blocks = 1;
        if (blocks > groups) blocks = groups;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, data, mid, tw_row, tw_col, LT - 2, groups,
                           1.0f / (float)cols);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
    }
    const int S = 2, LS = LT - S;
    AxisLastIO<float, INVERSE> m{mid, data, S, LS, LC, LT - LS, LT - 1 - S, elems, 1.0f / (float)rows};
    // (plain against streaming loads of the intermediate measured equal on 16 .. 128 MiB images, round 5; the factor path's rule)
    m.nt_load = elems * sizeof(cpx<float>) > (size_t(192) << 20);
    const size_t units = size_t(1) << (S + LC);
    switch (LS) {
#define KOFFT_CASE(LL) \
    case LL: return launch_tile_persist<float, LL, AxisLastIO<float, INVERSE>>(ctx, m, tw_col, units);
        KOFFT_CASE(8)  // LS = LT - 2, LT = 10 .. 13 (fft2d_fused_ok)
        KOFFT_CASE(9)
        KOFFT_CASE(10)
        KOFFT_CASE(11)  // 8192 rows (round 6): tiles of 8 columns at 1024 threads, 64-byte runs
#undef KOFFT_CASE
    default: return KOFFT_ERR_UNSUPPORTED;
    }
}

bool fft2d_fused_ok(const kofft_hip_ctx *ctx, size_t rows, size_t cols)
{
    // (fewer than one group of four rows per CU: the three-pass route is faster -- 512 x 4096: 0.034 against 0.026 ms)
    // (8192 rows, round 6: the column pass runs 2^11-point tiles of 8 columns = 64-byte runs -- 8192 x 2048 0.140 -> 0.124 ms, 8192 x 1024 0.076 -> 0.061; a 256 MiB
    // image, 8192 x 4096, no longer fits the Infinity Cache between the passes and LOSES, 0.307 -> 0.369 ms: it keeps rows + two column-tile passes)
    // (KOFFT_HIP_NO_PERSIST=1 forces the generic kernels here too.  The units-per-CU floor of the factor path's tile kernels
    // (big_persist_min_units) is NOT applied: its 4 x cols column units are 16 .. 64 per CU, and the route measured faster down to
    // 1024-point rows on images of at least 32 MiB -- 4096 x 1024: 0.044 -> 0.034 ms -- which is the rule below.)
    return ctx->nd_fused && ctx->use_persist && ctx->big_persist && (cols == 1024 || cols == 2048 || cols == 4096) && is_pow2(rows) && rows >= 1024 && (rows <= 4096 || (rows == 8192 && cols <= 2048)) &&
           rows / 4 >= (size_t)ctx->num_cus && rows * cols >= (size_t(1) << 22);  // (16 MiB images: 1024 x 2048 0.031 against 0.029 ms, 1024 x 1024 a tie)
}

int fft2d_fused_c32(kofft_hip_ctx *ctx, float *d_data, size_t rows, size_t cols, int inverse)
{
    cpx<float> *img = reinterpret_cast<cpx<float> *>(d_data);
    const int LT = ilog2(rows);
    switch (cols) {
    case 1024: return inverse ? fft2d_fused_core<10, true>(ctx, img, LT) : fft2d_fused_core<10, false>(ctx, img, LT);
    case 2048: return inverse ? fft2d_fused_core<11, true>(ctx, img, LT) : fft2d_fused_core<11, false>(ctx, img, LT);
    case 4096: return inverse ? fft2d_fused_core<12, true>(ctx, img, LT) : fft2d_fused_core<12, false>(ctx, img, LT);
    default: return KOFFT_ERR_UNSUPPORTED;
    }
}

}  // namespace host
}  // namespace kofft
