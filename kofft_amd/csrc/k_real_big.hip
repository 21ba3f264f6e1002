// k_real_big.hip -- rfft_direct (rfft.rs:425-465) for inner lengths beyond the single-workgroup kernels in TWO passes over HBM: the
// first factor with the row window on its loads, then the last factor with the post-pass in its epilogue (fft_big_real.hip.h).
// Its own translation unit so that `make -j` builds it beside the others.
#include "host_common.hip.h"
#include "complex_impl.hip.h"
#include "fft_big_real.hip.h"

namespace kofft {
namespace host {

#ifndef KOFFT_RROWS_RL
#define KOFFT_RROWS_RL 3
#endif
#ifndef KOFFT_RROWS_XPB
#define KOFFT_RROWS_XPB(LS) 8
#endif
template <int LS>
int launch_rows_rfft(kofft_hip_ctx *ctx, const cpx<float> *mid, cpx<float> *out, const cpx<float> *tw, const cpx<float> *rtab, int LA,
                     size_t nb, bool nt_load)
{
    // 8 points per thread: three register sets (tile S's result, tile A, the prefetch), the 2 x 8 post-pass table entries and the last
    // pass's twiddles of both tiles all stay in registers (16 points per thread: 270 of them)
    constexpr int RL = KOFFT_RROWS_RL;
    constexpr int TPT = (1 << LS) >> RL;
    constexpr int XPB = KOFFT_RROWS_XPB(LS);
    constexpr int BLOCK = XPB * TPT;
    constexpr size_t lds = TileExchange<float, LS, RL, XPB, false>::bytes + 2 * (size_t)rows_tw_entries<LS, RL>() * XPB * sizeof(cpx<float>) +
                           (size_t)(1 << LS) * sizeof(cpx<float>);
    // wavefronts per SIMD the register budget allows (144 registers at 8 points per thread: three), capped by the LDS
    constexpr int MINW = BLOCK >= 1024 ? 4 : (BLOCK == 512 ? 2 : 3);
    constexpr int WG_BY_REGS = MINW * 256 / BLOCK;
    constexpr int WG_BY_LDS = (int)((160 * 1024) / lds);
    constexpr int WG_PER_CU = WG_BY_REGS < WG_BY_LDS ? (WG_BY_REGS < 1 ? 1 : WG_BY_REGS) : WG_BY_LDS;
    static_assert(WG_PER_CU >= 1 && lds * WG_PER_CU <= 160 * 1024, "LDS budget");
    const size_t rows = size_t(1) << LA;
    if (rows < 2 * XPB || nb > 0xffffffffULL) return KOFFT_ERR_UNSUPPORTED;
    const size_t pairs = rows / (2 * XPB);
    auto kern = fft_rows_rfft_kernel<LS, RL, BLOCK, MINW>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    size_t blocks = (size_t)ctx->num_cus * WG_PER_CU;
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < pairs) blocks = pairs;  // every tile pair needs its workgroup
    blocks -= blocks % pairs;
    size_t groups = blocks / pairs;
    if (groups > nb) groups = nb;
    blocks = groups * pairs;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, mid, out, tw, rtab, LA, (unsigned)nb, (unsigned)pairs,
                       nt_load ? 1 : 0);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// factor sizes of the fused route: the split fft_big_core takes for c32 (the larger factor first), rows of 2^8 .. 2^10 points
bool rfft_big_fused_ok(const kofft_hip_ctx *ctx, size_t m, size_t batch)
{
    if (!ctx->rfft_big_fused || !ctx->big_persist || !is_pow2(m)) return false;
    const int L = ilog2(m);
    if (L < 16 || L > 21) return false;  // (2^15: the register-file kernel; from 2^22: three factors)
    // every workgroup of the last factor needs a transform of its tile pair: smaller batches keep the three-pass route
    const int L3 = L == 21 ? 10 : L / 2, L1 = L - L3;
    const size_t xpb = KOFFT_RROWS_XPB(L3), pairs = (size_t(1) << L1) / (2 * xpb);
    const size_t chunk = std::max<size_t>(1, std::min(batch, ctx->big_chunk_bytes / (m * sizeof(cpx<float>))));
    return chunk * pairs >= (size_t)ctx->num_cus;
}

int rfft_big_fused_f32(kofft_hip_ctx *ctx, const float *d_in, float *d_out, const float *d_window, size_t m, size_t batch)
{
    using T = float;
    const int L = ilog2(m);
    const int L3 = L == 21 ? 10 : L / 2, L1 = L - L3;
    const cpx<T> *tw = nullptr, *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::tw, m, &tw);
    if (rc) return rc;
    rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    const size_t xf_bytes = m * sizeof(cpx<T>);
    size_t chunk = ctx->big_chunk_bytes / xf_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const size_t need = chunk * xf_bytes;
    if (ctx->big_tmp_bytes < need) {
        if (ctx->big_tmp_external) return KOFFT_ERR_ALLOC;
        if (ctx->big_tmp) KOFFT_HIP_TRY(ctx, hipFree(ctx->big_tmp));
        ctx->big_tmp = nullptr;
        ctx->big_tmp_bytes = 0;
        KOFFT_HIP_TRY(ctx, hipMalloc(&ctx->big_tmp, need));
        ctx->big_tmp_bytes = need;
    }
    cpx<T> *mid = static_cast<cpx<T> *>(ctx->big_tmp);
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        // z[i] = (x[2i], x[2i+1]) (rfft.rs:444-446) is the row itself read as m complex values
        const cpx<T> *src = reinterpret_cast<const cpx<T> *>(d_in + b0 * 2 * m);
        cpx<T> *dst = reinterpret_cast<cpx<T> *>(d_out) + b0 * (m + 1);
        if (d_window) {  // the row window on the first factor's loads (stft.rs:96's product, element by element)
            BigColsIO<T, false, PRE_WINDOW> a{src, mid, L - L1, L - L1, m};
            a.pre_tab = reinterpret_cast<const cpx<T> *>(d_window);
            rc = launch_sub<T>(ctx, a, tw, L1, nb << (L - L1), true);
        } else {
            BigColsIO<T, false, PRE_NONE> a{src, mid, L - L1, L - L1, m};
            rc = launch_sub<T>(ctx, a, tw, L1, nb << (L - L1), true);
        }
        if (rc) return rc;
        const bool nt_load = nb * xf_bytes > (size_t(192) << 20);  // (an intermediate that fits the Infinity Cache is read with plain loads)
        switch (L3) {
        case 8: rc = launch_rows_rfft<8>(ctx, mid, dst, tw, rtab, L1, nb, nt_load); break;
        case 9: rc = launch_rows_rfft<9>(ctx, mid, dst, tw, rtab, L1, nb, nt_load); break;
        case 10: rc = launch_rows_rfft<10>(ctx, mid, dst, tw, rtab, L1, nb, nt_load); break;
        default: rc = KOFFT_ERR_UNSUPPORTED; break;
        }
        if (rc) return rc;
    }
    return KOFFT_OK;
}

}  // namespace host
}  // namespace kofft
