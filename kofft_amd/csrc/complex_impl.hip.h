// complex_impl.hip.h -- FftImpl::fft / ifft on device pointers for one element type: the power-of-two kernels, the
// two- / three-factor path for large n and the Bluestein arm.  Included by k_complex_f32.hip / k_complex_f64.hip.
#pragma once

#include "host_common.hip.h"
#include "fft_radix4.hip.h"

namespace kofft {
namespace host {

// ---------------------------------------------------------------------------------
// large n: two factors (fft_big.hip.h)
// ---------------------------------------------------------------------------------

#ifndef KOFFT_ROWS_C32_CAP_KB
#define KOFFT_ROWS_C32_CAP_KB 80
#endif

// The persistent, prefetching form of a factor (fft_tile_persist_kernel): 512 threads per CU (one or several workgroups),
// so that every thread may use 256 registers -- two register sets and a pass's twiddles, no scratch.
template <typename T, int LS, class IO>
int launch_tile_persist(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t units)
{
    constexpr int RL = rl_for(LS);
    constexpr int BLOCK = big_block<T, IO, LS, 128 * 1024>();
    constexpr int TPT = (1 << LS) >> RL;
    constexpr int XPB = BLOCK / TPT;
    constexpr size_t lds = TileExchange<T, LS, RL, XPB, IO::kSplitLds>::bytes +
                           (IO::kTileInvariantTw ? (size_t(1) << LS) / 2 * sizeof(cpx<T>) : 0) +  // + the first factor's table copy
                           (io_tile_group_tw<IO>::value ? (size_t(1) << LS) * sizeof(cpx<T>) : 0);  // or the middle factor's per-run table
    constexpr int WG_PER_CU = BLOCK >= 512 ? 1 : 512 / BLOCK;
    static_assert(lds * WG_PER_CU <= 160 * 1024, "LDS budget");
    if (units % XPB != 0) return KOFFT_ERR_UNSUPPORTED;  // (never: units = transforms << bits, bits >= 7)
    auto kern = fft_tile_persist_kernel<T, LS, RL, BLOCK, IO>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    const size_t ntiles = units / XPB;
    size_t blocks = (size_t)ctx->num_cus * WG_PER_CU;
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < 1) blocks = 1;
    if (blocks > ntiles) blocks = ntiles;
    IO kio = io;
    if constexpr (io_tile_group_tw<IO>::value) {
        // runs of tiles that share a table: as long as possible while every workgroup still gets a run
        size_t tpg = (size_t(1) << kio.JB) / XPB;
        if (tpg < 1 || ((size_t(1) << kio.JB) % XPB) != 0) tpg = 1;
        while (tpg > 1 && ntiles / tpg < blocks) tpg /= 2;
        if (!ctx->big_mid_group) tpg = 1;
        kio.tpg = (int)tpg;
        if (blocks > ntiles / tpg) blocks = ntiles / tpg;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, kio, tw, ntiles);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// The last factor with its table entries resident per row tile (fft_rows_persist_kernel).
template <typename T, int LS, class IO>
int launch_rows_persist(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t nb)
{
    constexpr int RL = rl_for(LS);
    // (c32 rows of 2^10 points: 16-row tiles need 1024 threads at 128 registers and spill 30 of them -- measured slower
    // than 8-row tiles at 512 threads, 2^19: 0.244 vs 0.282 of the roofline)
    constexpr int BLOCK = big_block<T, IO, LS, (sizeof(T) == 4 ? KOFFT_ROWS_C32_CAP_KB : 128) * 1024>();
    constexpr int TPT = (1 << LS) >> RL;
    constexpr int XPB = BLOCK / TPT;
    constexpr size_t lds = TileExchange<T, LS, RL, XPB, IO::kSplitLds>::bytes + (size_t)rows_tw_entries<LS, RL>() * XPB * sizeof(cpx<T>);
    constexpr int WG_PER_CU = BLOCK >= 512 ? 1 : 512 / BLOCK;
    static_assert(lds * WG_PER_CU <= 160 * 1024, "LDS budget");
    if (nb > 0xffffffffULL || ((size_t(1) << io.LA) % XPB) != 0) return KOFFT_ERR_UNSUPPORTED;
    auto kern = fft_rows_persist_kernel<T, LS, RL, BLOCK, IO>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    const size_t KT = (size_t(1) << io.LA) / XPB;  // row tiles per transform
    size_t blocks = (size_t)ctx->num_cus * WG_PER_CU;
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < 1) blocks = 1;
    // one row tile per workgroup: whole groups of workgroups per row tile (no more groups than transforms) when the chip
    // has a CU for every row tile, otherwise one launch per slice of `blocks` row tiles
    for (size_t kt0 = 0; kt0 < KT; kt0 += blocks) {
        const size_t cnt = KT - kt0 < blocks ? KT - kt0 : blocks;
        size_t groups = blocks / cnt;
        if (groups > nb) groups = nb;
        if (groups < 1) groups = 1;
        hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * groups)), dim3(BLOCK), lds, ctx->stream, io, tw, (unsigned)nb, (unsigned)kt0,
                           (unsigned)cnt);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
    }
    return KOFFT_OK;
}

// Few units -- ONE large transform, or a handful: tiles of the full width leave most CUs without a workgroup (2^20 points
// = 1024 columns = 128 tiles of 8), and a single transform is all latency.  Halve the tile width (once or twice) until
// every CU has a workgroup; segments get narrower, but such a transform lives in the L2 / Infinity Cache anyway.
template <class IO>
inline void narrow_tile_adjust(IO &, size_t) {}
template <typename T, bool INVERSE, int POST>
inline void narrow_tile_adjust(BigRowsIO<T, INVERSE, POST> &io, size_t segment_bytes)
{
    if (segment_bytes < 64) io.nt = false;  // streaming stores only for segments of at least half a line (as in fft_big_dev)
    io.nt_load = false;                     // (a single transform's intermediate sits in the caches anyway)
}
template <typename T, bool INVERSE, int PRE>
inline void narrow_tile_adjust(BigColsIO<T, INVERSE, PRE> &io, size_t segment_bytes)
{
    if (segment_bytes < 64) io.nt_in_pieces = false;  // pieces of less than half a line: the neighbours want the rest
}
// (Round 6, tools/kernel_coverage.sh: which narrowed forms a call can reach.  A 2^10-point factor is the FIRST factor of 2^20 / 2^21 points
// only -- at least 1024 columns, i.e. 256 half-width tiles: never narrowed twice.  A 2^7-point LAST factor in f32 exists only as the third
// of three factors (2^21 / 2^22: 2^14 rows and more -- full-width tiles fill the chip); in f64 it is the last factor of 2^14 points too.)
template <class IO> struct sub_is_rows { static constexpr bool value = false; };
template <typename T, bool INVERSE, int POST> struct sub_is_rows<BigRowsIO<T, INVERSE, POST>> { static constexpr bool value = true; };
template <typename T, int LL, class IO>
int launch_sub_one_tile(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t units)
{
    constexpr int B0 = big_block<T, IO, LL>();
    constexpr int TPT = (1 << LL) >> rl_for(LL);
    constexpr bool kNarrow = !(sizeof(T) == 4 && LL == 7 && sub_is_rows<IO>::value);
    const size_t cus = (size_t)ctx->num_cus * (size_t)ctx->big_narrow_per_cu;
    if constexpr (kNarrow && B0 / 2 >= 64 && (B0 / 2) % TPT == 0) {
        if (ctx->big_narrow && units / (B0 / TPT) < cus) {
            IO nio = io;
            if constexpr (LL <= 9 && B0 / 4 >= 64 && (B0 / 4) % TPT == 0) {
                if (units / (B0 / 2 / TPT) < cus) {
                    narrow_tile_adjust(nio, (size_t)(B0 / 4 / TPT) * sizeof(cpx<T>));
                    return launch_wg<T, LL, EPI_STORE, IO, B0 / 4>(ctx, nio, tw, units);
                }
            }
            narrow_tile_adjust(nio, (size_t)(B0 / 2 / TPT) * sizeof(cpx<T>));
            return launch_wg<T, LL, EPI_STORE, IO, B0 / 2>(ctx, nio, tw, units);
        }
    }
    return launch_wg<T, LL, EPI_STORE, IO, B0>(ctx, io, tw, units);
}

// smallest sub-transform a policy is launched with (launch_sub): 2^7, except the plain first-factor policy in f32 -- ndfft's 2048-point axes
// run as 2^6 x 2^5 (fft_axis2_dev; f64 axes take the two-pass route from 4096 points on: 2^7 first)
template <class IO> struct sub_min_ls_of { static constexpr int value = 7; };
template <bool INVERSE> struct sub_min_ls_of<BigColsIO<float, INVERSE, 0>> { static constexpr int value = 6; };
template <class IO> constexpr int sub_min_ls() { return sub_min_ls_of<IO>::value; }

// policies of the LAST factor (BigRowsIO): their persistent form is fft_rows_persist_kernel (launch_rows_persist), taken by fft_big_core
// whenever the conditions of the tile kernel hold -- the generic persistent tile kernel is never their route (round 6: its 40
// BigRowsIO instantiations, reachable only through round 4's compile-time measurement switch, are gone)
template <class IO> struct sub_tile_persist_of { static constexpr bool value = true; };
template <typename T, bool INVERSE, int POST> struct sub_tile_persist_of<BigRowsIO<T, INVERSE, POST>> { static constexpr bool value = false; };

// Launch the generic kernel for a sub-transform of log2 size LS with an arbitrary IO policy.
template <typename T, class IO>
int launch_sub(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, int LS, size_t units, bool persist)
{
    if constexpr (sub_tile_persist_of<IO>::value)
    if (persist && units >= (size_t)ctx->num_cus * ctx->big_persist_min_units) {  // every resident workgroup gets several tiles
        switch (LS) {
#define KOFFT_CASE(LL) \
    case LL: return launch_tile_persist<T, LL, IO>(ctx, io, tw, units);
            KOFFT_CASE(7)
            KOFFT_CASE(8)
            KOFFT_CASE(9)
            KOFFT_CASE(10)
#undef KOFFT_CASE
        case 11:  // first factor of 2^21 = 2^11 x 2^10: c32 8 columns at 1024 threads, c64 4 columns at 512 threads (both measured, DESIGN 5.3;
                  // parity: test_large_n_persistent_factor_kernels c32 (21, 11) and c64 (21, 9))
            if constexpr (IO::kTileInvariantTw) {
                if (ctx->big_first11) return launch_tile_persist<T, 11, IO>(ctx, io, tw, units);
            }
            break;
        default: break;  // 2^11 .. 2^13-point factors: tiles of 4, 2, 1 units -- the one-tile-per-workgroup kernel
        }
    }
    // Sizes a default route can reach (round 6: the instantiations nothing reaches are gone).  The factor path's sub-transforms are 2^7 .. 2^11
    // points: two factors up to 2^22 = 2^11 x 2^11 (fft_big_core forces three factors from 2^23 whatever KOFFT_HIP_BIG_THREE_MIN says), three
    // factors of 7 .. 9 bits beyond.  2^5 / 2^6 only as the first pass of ndfft's two-pass axes, i.e. a first-factor policy without a folded
    // pointwise factor (sub_min_ls).
    switch (LS) {
    // 16 (c32) / 8 (c64) adjacent columns or rows per workgroup = 128-byte segments while the tile fits the LDS budget
    // (sub-transforms up to 2^9 points; 2^10: 8; larger ones fewer still -- big_block).
    case 5:
        if constexpr (sub_min_ls<IO>() <= 5) return launch_wg<T, 5, EPI_STORE, IO, big_block<T, IO, 5>()>(ctx, io, tw, units);
        break;
    case 6:
        if constexpr (sub_min_ls<IO>() <= 6) return launch_wg<T, 6, EPI_STORE, IO, big_block<T, IO, 6>()>(ctx, io, tw, units);
        break;
    case 7: return launch_sub_one_tile<T, 7, IO>(ctx, io, tw, units);
    case 8: return launch_sub_one_tile<T, 8, IO>(ctx, io, tw, units);
    case 9: return launch_sub_one_tile<T, 9, IO>(ctx, io, tw, units);
    case 10: return launch_sub_one_tile<T, 10, IO>(ctx, io, tw, units);
    case 11: return launch_wg<T, 11, EPI_STORE, IO, big_block<T, IO, 11>()>(ctx, io, tw, units);
    default: break;
    }
    return KOFFT_ERR_UNSUPPORTED;
}

// Sub-transforms of the middle factor are at most 2^9 points (three factors cover 2^21 .. 2^26 with 7..9 bits each).
template <typename T>
int launch_mid(kofft_hip_ctx *ctx, const BigMidIO<T> &io, const cpx<T> *tw, int LS, size_t units)
{
    if (ctx->big_persist && units >= (size_t)ctx->num_cus * ctx->big_persist_min_units) {
        switch (LS) {
        case 7: return launch_tile_persist<T, 7, BigMidIO<T>>(ctx, io, tw, units);
        case 8: return launch_tile_persist<T, 8, BigMidIO<T>>(ctx, io, tw, units);
        case 9: return launch_tile_persist<T, 9, BigMidIO<T>>(ctx, io, tw, units);
        default: break;
        }
    }
    switch (LS) {
    case 7: return launch_wg<T, 7, EPI_STORE, BigMidIO<T>, big_block<T, BigMidIO<T>, 7>()>(ctx, io, tw, units);
    case 8: return launch_wg<T, 8, EPI_STORE, BigMidIO<T>, big_block<T, BigMidIO<T>, 8>()>(ctx, io, tw, units);
    case 9: return launch_wg<T, 9, EPI_STORE, BigMidIO<T>, big_block<T, BigMidIO<T>, 9>()>(ctx, io, tw, units);
    default: return KOFFT_ERR_UNSUPPORTED;
    }
}

// ndfft's long strided axes (ndfft.rs:89-98, 131-151) in two column-tile passes: see AxisLastIO (fft_big.hip.h).
// blocks dense [2^LT][2^I] arrays, in place through the context's real_tmp scratch.
template <typename T, bool INVERSE>
int fft_axis2_core(kofft_hip_ctx *ctx, cpx<T> *data, int LT, int I, size_t blocks, int L1)
{
    const size_t len = size_t(1) << LT, block_elems = len << I, block_bytes = block_elems * sizeof(cpx<T>);
    const cpx<T> *tw = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::tw, len, &tw);
    if (rc) return rc;
    size_t chunk = (size_t(512) << 20) / block_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > blocks) chunk = blocks;
    rc = ensure_real_tmp(ctx, chunk * block_bytes);
    if (rc) return rc;
    cpx<T> *mid = static_cast<cpx<T> *>(ctx->real_tmp);
    const int LS = LT - L1;
    const T scale = (T)1 / (T)(float)len;
    for (size_t b0 = 0; b0 < blocks; b0 += chunk) {
        const size_t nb = blocks - b0 < chunk ? blocks - b0 : chunk;
        cpx<T> *blk = data + b0 * block_elems;
        BigColsIO<T, INVERSE> a{blk, mid, LS + I, LS, block_elems};
        rc = launch_sub<T>(ctx, a, tw, L1, nb << (LS + I), ctx->big_persist);
        if (rc) return rc;
        AxisLastIO<T, INVERSE> m{mid, blk, L1, LS, I, LT - LS, LT - 1 - L1, block_elems, scale};
        const size_t units = nb << (L1 + I);
        rc = KOFFT_ERR_UNSUPPORTED;
        // LS = LT - L1 with L1 = 7 from 4096 points on: 5 .. 7 in f32 (axes up to 2^14 points), 5 / 6 in f64 (up to 2^13) -- fft_axis2_dev
        if constexpr (sizeof(T) == 4) {
            if (ctx->big_persist && units >= (size_t)ctx->num_cus * ctx->big_persist_min_units && LS == 7)
                rc = launch_tile_persist<T, 7, AxisLastIO<T, INVERSE>>(ctx, m, tw, units);
        }
        if (rc == KOFFT_ERR_UNSUPPORTED) {
            switch (LS) {
#define KOFFT_CASE(LL) \
    case LL: rc = launch_wg<T, LL, EPI_STORE, AxisLastIO<T, INVERSE>, big_block<T, AxisLastIO<T, INVERSE>, LL>()>(ctx, m, tw, units); break;
                KOFFT_CASE(5)
                KOFFT_CASE(6)
#undef KOFFT_CASE
            case 7:
                if constexpr (sizeof(T) == 4)
                    rc = launch_wg<T, 7, EPI_STORE, AxisLastIO<T, INVERSE>, big_block<T, AxisLastIO<T, INVERSE>, 7>()>(ctx, m, tw, units);
                break;
            default: break;
            }
        }
        if (rc) return rc;
    }
    return KOFFT_OK;
}

template <typename T>
int fft_axis2_dev(kofft_hip_ctx *ctx, T *d_data, int LT, int I, size_t blocks, int inverse)
{
    // 2^12 = 2^7 x 2^5, 2^13 = 2^7 x 2^6, 2^14 = 2^7 x 2^7: the first pass on the persistent prefetching tile kernel (4096 x 4096 c32,
    // same box: transposes 0.224 ms, 2^5 x 2^7 0.169-0.173, 2^6 x 2^6 0.178, 2^7 x 2^5 0.149)
    const int L1 = LT >= 12 ? 7 : LT - 5;  // (2^11 = 2^6 x 2^5, c32 only: k_nd.hip)
    if (L1 < (sizeof(T) == 4 ? 6 : 7) || L1 > 7 || LT - L1 < 5 || LT - L1 > (sizeof(T) == 4 ? 7 : 6)) return KOFFT_ERR_UNSUPPORTED;
    cpx<T> *data = reinterpret_cast<cpx<T> *>(d_data);
    return inverse ? fft_axis2_core<T, true>(ctx, data, LT, I, blocks, L1) : fft_axis2_core<T, false>(ctx, data, LT, I, blocks, L1);
}

// c32 last factor on ROW PAIRS (round 4): BigRowsIO<float, INV, POST_NONE> -> BigRowsIO<f32x2, INV, POST_NONE> in pair units
template <class RowsIO> struct rows_pair_io { static constexpr bool ok = false; };
template <bool INV> struct rows_pair_io<BigRowsIO<float, INV, POST_NONE>> {
    static constexpr bool ok = true;
    using type = BigRowsIO<f32x2, INV, POST_NONE>;
};

template <typename T>
inline int big_rows_per_wg(int LB) { return LB <= 9 ? KOFFT_BIG_XPB(T) : LB == 10 ? 8 : LB == 11 ? 4 : LB == 12 ? 2 : 1; }

// units (columns / rows) per tile of the two persistent factor kernels, as launch_tile_persist / launch_rows_persist compute them
struct NoIO {};
template <typename T>
inline int tile_persist_xpb(int LS)
{
    switch (LS) {
#define KOFFT_CASE(LL) \
    case LL: return big_block<T, NoIO, LL, 128 * 1024>() / ((1 << LL) >> rl_for(LL));
        KOFFT_CASE(7)
        KOFFT_CASE(8)
        KOFFT_CASE(9)
        KOFFT_CASE(10)
        KOFFT_CASE(11)
#undef KOFFT_CASE
    default: return 0;
    }
}
template <typename T>
inline int rows_persist_xpb(int LS)
{
    switch (LS) {
#define KOFFT_CASE(LL) \
    case LL: return big_block<T, NoIO, LL, (sizeof(T) == 4 ? KOFFT_ROWS_C32_CAP_KB : 128) * 1024>() / ((1 << LL) >> rl_for(LL));
        KOFFT_CASE(7)
        KOFFT_CASE(8)
        KOFFT_CASE(9)
        KOFFT_CASE(10)
#undef KOFFT_CASE
    default: return 0;
    }
}

// Placement probe of the factor path's intermediate (round 6; fft_big_core explains why).  K candidate allocations of `need`
// bytes; `run(in, out, cand, ev_mid)` enqueues the caller's own factor kernels for one full chunk from the scratch input through
// the candidate to the scratch output (ev_mid is recorded behind the first factor).  Keeps the candidate with the smallest chunk
// time in ctx->big_tmp, frees the others, leaves the figures in ctx->big_probe_* (kofft_hip_big_probe_info).  Allocation
// failures shrink K; with fewer than two candidates (or no scratch) nothing is measured and the caller allocates as before.
template <class Run>
int big_probe_pick(kofft_hip_ctx *ctx, size_t need, size_t io_bytes, int K, Run run)
{
    if (K > KOFFT_BIG_PROBE_MAX) K = KOFFT_BIG_PROBE_MAX;
    void *io[2] = {nullptr, nullptr};
    void *cand[KOFFT_BIG_PROBE_MAX] = {};
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int got = 0, rc = KOFFT_OK, best = -1;
    auto cleanup = [&](int keep) {
        for (int i = 0; i < got; ++i)
            if (i != keep && cand[i]) (void)hipFree(cand[i]);
        for (void *p : io)
            if (p) (void)hipFree(p);
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    };
    if (hipMalloc(&io[0], io_bytes) != hipSuccess || hipMalloc(&io[1], io_bytes) != hipSuccess) {
        (void)hipGetLastError();
        cleanup(-1);
        return KOFFT_OK;  // no room for the scratch pair: no probe
    }
    for (; got < K; ++got)
        if (hipMalloc(&cand[got], need) != hipSuccess) {
            (void)hipGetLastError();
            cand[got] = nullptr;
            break;
        }
    if (got < 2) {
        if (got == 1) {
            ctx->big_tmp = cand[0];
            ctx->big_tmp_bytes = need;
        }
        cleanup(0);
        return KOFFT_OK;
    }
    hipError_t he = hipMemsetAsync(io[0], 0, io_bytes, ctx->stream);
    for (int i = 0; i < 3 && he == hipSuccess; ++i) he = hipEventCreate(&ev[i]);
    if (he != hipSuccess) {
        ctx->last_error = std::string("placement probe: ") + hipGetErrorString(he);
        cleanup(-1);
        return KOFFT_ERR_HIP;
    }
    float best_us = 0.f;
    for (int i = 0; i < got && rc == KOFFT_OK; ++i) {
        float first_us = 0.f, total_us = 0.f;
        for (int rep = 0; rep < 3 && rc == KOFFT_OK; ++rep) {  // rep 0: warm (tables, code, clocks)
            he = hipEventRecord(ev[0], ctx->stream);
            if (he == hipSuccess) rc = run(io[0], io[1], cand[i], ev[1]);
            if (rc == KOFFT_OK && he == hipSuccess) he = hipEventRecord(ev[2], ctx->stream);
            if (rc == KOFFT_OK && he == hipSuccess) he = hipEventSynchronize(ev[2]);
            float a_ms = 0.f, t_ms = 0.f;
            if (rc == KOFFT_OK && he == hipSuccess) he = hipEventElapsedTime(&a_ms, ev[0], ev[1]);
            if (rc == KOFFT_OK && he == hipSuccess) he = hipEventElapsedTime(&t_ms, ev[0], ev[2]);
            if (he != hipSuccess) {
                ctx->last_error = std::string("placement probe: ") + hipGetErrorString(he);
                rc = KOFFT_ERR_HIP;
            }
            if (rep >= 1 && (rep == 1 || t_ms * 1e3f < total_us)) {
                total_us = t_ms * 1e3f;
                first_us = a_ms * 1e3f;
            }
        }
        ctx->big_probe_first_us[i] = first_us;
        ctx->big_probe_total_us[i] = total_us;
        if (rc == KOFFT_OK && (best < 0 || total_us < best_us)) {
            best = i;
            best_us = total_us;
        }
    }
    if (rc != KOFFT_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        cleanup(-1);
        return rc;
    }
    ctx->big_probe_n = got;
    ctx->big_probe_pick = best;
    ctx->big_tmp = cand[best];
    ctx->big_tmp_bytes = need;
    cleanup(best);
    return KOFFT_OK;
}

// The factor path with the policies of its first and last factor as parameters (round 3): ColsIO / RowsIO are BigColsIO /
// BigRowsIO instances, possibly with a folded pointwise factor (PRE / POST) whose extra fields fix_cols / fix_rows fill in;
// such policies may read input rows of in_row values and write output rows of out_row values (the transform itself is n).
template <typename T, class ColsIO, class RowsIO, class FixCols, class FixRows>
int fft_big_core(kofft_hip_ctx *ctx, const cpx<T> *in_base, size_t in_row, cpx<T> *out_base, size_t out_row, size_t n, size_t batch,
                 FixCols fix_cols, FixRows fix_rows)
{
    const int L = ilog2(n);
    // Two factors while both stay <= 2^10 points (tiles of 8 adjacent columns / rows, 64..128-byte segments); from 2^22
    // on, three factors of 7..9 bits: one more pass over HBM, but every pass keeps full-width tiles (two factors of
    // 11..13 bits shrink the tiles to 4, 2, 1 columns and fall to 0.05..0.16 of the roofline).
    // measured crossover, KOFFT_HIP_BIG_THREE_MIN; from 2^23 always three (two factors would need 2^12-point sub-transforms: tiles of 2 columns)
    const bool three = (L >= ctx->big_three_min && !ctx->big_two_only) || L >= 23;
    // two factors of an odd L: the larger one first when the persistent first-factor kernel covers it (c32, 2^11 points)
    // (only for batches the persistent kernels take: a single 2^21-point transform is faster as 2^10 x 2^11, 31.7 vs 33.6 us)
    const bool first11 = ctx->big_first11 && ctx->big_persist && L == 21 && !three &&
                         (batch << 10) >= (size_t)ctx->num_cus * ctx->big_persist_min_units;
    const bool first_larger = (ctx->big_first_larger >= 0 ? ctx->big_first_larger != 0 : sizeof(T) == 4) && L <= 20 && ctx->big_persist &&
                              (batch << (L - (L + 1) / 2)) >= (size_t)ctx->num_cus * ctx->big_persist_min_units;  // batches only
    // (odd log2 n: the first factor -- the cheaper kernel: one table for all its tiles -- takes the extra bit in f32: 2^15 / 2^17 /
    // 2^19 +1.5..3 % on the same box; c64 shows no difference and keeps the smaller first factor; a single transform
    // -- the one-tile-per-workgroup kernels -- is faster the other way at 2^19, 14.0 vs 15.2 us, and keeps it too)
    // (three factors: a larger LAST factor -- 7/7/9 or 8/7/9 and 7/7/10 instead of 8/8/7 and 8/8/8 -- measured slower, round 4: 2^23 0.184 -> 0.168,
    // 2^24 0.170 -> 0.160 c32, the same for c64)
    const int L1 = three ? (L + 2) / 3 : (first11 ? 11 : (first_larger ? (L + 1) / 2 : L / 2));
    const int L2 = three ? (L - L1 + 1) / 2 : 0;
    const int L3 = L - L1 - L2;
    const cpx<T> *tw = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::tw, n, &tw);
    if (rc) return rc;
    const size_t xf_bytes = n * sizeof(cpx<T>);
    size_t chunk = ctx->big_chunk_bytes / xf_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const size_t need = chunk * xf_bytes * (three ? 2 : 1);
    const T scale = (T)1 / (T)(float)n;
    // one chunk of nb transforms: src -> mid [-> mid2] -> dst
    auto run_chunk = [&](const cpx<T> *src, cpx<T> *dst, cpx<T> *mid, size_t nb, hipEvent_t ev_mid) -> int {
        cpx<T> *mid2 = mid + chunk * n;
        int rc = KOFFT_OK;
        // first factor: stages 0 .. L1-1 down the columns of a 2^L1 x 2^(L-L1) matrix
        ColsIO a{src, mid, L - L1, L - L1, n};
        fix_cols(a);
        const bool first_persist = ctx->big_first_persist >= 0 ? ctx->big_first_persist != 0 : ctx->big_persist;
        const size_t persist_units = (size_t)ctx->num_cus * ctx->big_persist_min_units;
        // Both factors on their persistent kernels (the conditions of launch_sub and of the rows branch below): the intermediate
        // is then block-interleaved -- see BigColsIO::out_lane.
        const bool rows_resident = ctx->big_persist && (nb << (L - L3)) >= persist_units && L3 >= 7 && L3 <= 10;
        // c64 only.  On identical buffers (tools/exp_c64_blocked.py, profiles/r04_c64_blocked_ab.txt): equal where the intermediate's
        // placement is fast, last factor 218-223 -> 203-207 us per chunk where it is slow, never slower.  c32 2^18 .. 2^20
        // lost 4-6 % in process-level A/Bs (0.324 / 0.315 / 0.284 -> 0.305 / 0.295 / 0.269) and keeps the natural layout.
#ifndef KOFFT_BLOCKED_C32
#define KOFFT_BLOCKED_C32 0 /* measurement builds: the block-interleaved intermediate for c32 too */
#endif
        // c32: the last factor on PAIRS of adjacent rows (BigRowsIO<f32x2>: two Complex<f32> values as one 16-byte value through the c64
        // kernel's structure -- 16-row tiles at 512 threads, 128-byte runs on both sides) when its policy has no folded pointwise factor; the
        // first factor then interleaves the rows of a pair (blk_r = 1: 256-byte runs per wavefront store).
        bool rows_pairs = false;
        if constexpr (rows_pair_io<RowsIO>::ok) {
            rows_pairs = ctx->big_row_pairs && !three && first_persist && (nb << (L - L1)) >= persist_units && L1 >= 7 &&
                         (L1 <= 10 || (L1 == 11 && first11)) && rows_resident;
            if (rows_pairs) {
                a.blk_r = 1;
                a.blk_c = 0;
            }
        }
        if ((sizeof(T) == 8 || KOFFT_BLOCKED_C32) && ctx->big_blocked && !three && first_persist && (nb << (L - L1)) >= persist_units && L1 >= 7 && L1 <= 10 && rows_resident) {
            a.blk_c = ilog2((size_t)tile_persist_xpb<T>(L1));
            a.blk_r = ilog2((size_t)rows_persist_xpb<T>(L3));
        }
        rc = launch_sub<T>(ctx, a, tw, L1, nb << (L - L1), first_persist);
        if (rc) return rc;
        if (ev_mid) KOFFT_HIP_TRY(ctx, hipEventRecord(ev_mid, ctx->stream));  // (the probe below: first factor | the rest)
        const cpx<T> *last_in = mid;
        if (three) {
            BigMidIO<T> m{mid, mid2, L1, L2, L3, L - L2, L - 1 - L1, n};
            rc = launch_mid<T>(ctx, m, tw, L2, nb << (L - L2));
            if (rc) return rc;
            last_in = mid2;
        }
        // last factor: the remaining L3 stages along contiguous rows, prefix K of L - L3 bits, output transposed
        const int LP = L - L3;
        RowsIO b{last_in, dst, LP, L3, L - L3, L - 1 - LP, n, scale, big_rows_per_wg<T>(L3) * sizeof(cpx<T>) >= 64};
        fix_rows(b);
        if constexpr (rows_pair_io<RowsIO>::ok) {
            if (rows_pairs) {  // everything in units of one pair (16 bytes): a 2^(LP-1) x 2^L3 matrix in the natural layout
                using PairIO = typename rows_pair_io<RowsIO>::type;
                f32x2 sc;
                sc.x = scale;
                sc.y = scale;
                PairIO bp{reinterpret_cast<const cpx<f32x2> *>(last_in), reinterpret_cast<cpx<f32x2> *>(dst), LP - 1, L3, L - L3, L - 1 - LP, n / 2, sc, true};
                bp.nt_load = nb * xf_bytes > (size_t(192) << 20);
                const cpx<f32x2> *twp = reinterpret_cast<const cpx<f32x2> *>(tw);  // (only ever read through tw_at: the Complex<f32> table)
                rc = KOFFT_ERR_UNSUPPORTED;
                switch (L3) {
                case 7: rc = launch_rows_persist<f32x2, 7>(ctx, bp, twp, nb); break;
                case 8: rc = launch_rows_persist<f32x2, 8>(ctx, bp, twp, nb); break;
                case 9: rc = launch_rows_persist<f32x2, 9>(ctx, bp, twp, nb); break;
                case 10: rc = launch_rows_persist<f32x2, 10>(ctx, bp, twp, nb); break;
                default: break;
                }
                return rc;
            }
        }
        b.blk_r = a.blk_r;
        b.blk_c = a.blk_c;
        // an intermediate small enough to stay in the 256 MiB Infinity Cache is read with plain loads (measured on a
        // copy model, tools/ubench_mall: streaming hints on the caller's buffers only, 3.1 -> 2.5 ms per 2 x 4 GiB)
        // ... and only where a wavefront's load instruction covers at least half a line per row: with 16-row c32 tiles (rows up
        // to 2^9 points) a load is sixteen 32-byte pieces, the rest of each line is wanted by the neighbouring wavefronts a
        // moment later, and a streaming load does not keep it for them (measured, 2 GiB batches: c32 2^15..2^18 0.275-0.295 ->
        // 0.304-0.314 with plain loads, 2^22 0.190 -> 0.198; c32 2^19..2^21 and every c64 size lose 3-10 % without the hint).
#if KOFFT_BLOCKED_C32 == 2 /* ... with streaming loads of its 512-byte runs */
        const size_t load_piece = (size_t)(b.blk_r != 0 ? 64 : 64 / big_rows_per_wg<T>(L3)) * sizeof(cpx<T>);
#else
        const size_t load_piece = (size_t)(64 / big_rows_per_wg<T>(L3)) * sizeof(cpx<T>);
#endif
        b.nt_load = ctx->big_mid_nt >= 0 ? ctx->big_mid_nt != 0 : (nb * xf_bytes > (size_t(192) << 20) && load_piece >= 64);
        rc = KOFFT_ERR_UNSUPPORTED;
        // last factor: rows resident (table entries per row tile in LDS) for batches, else one tile per workgroup (two 512-thread workgroups
        // per CU at 128 registers).  (The generic persistent tile kernel as a third form was round 4's measurement switch: removed in round 6.)
        if (rows_resident) {
            switch (L3) {
            case 7: rc = launch_rows_persist<T, 7>(ctx, b, tw, nb); break;
            case 8: rc = launch_rows_persist<T, 8>(ctx, b, tw, nb); break;
            case 9: rc = launch_rows_persist<T, 9>(ctx, b, tw, nb); break;
            case 10: rc = launch_rows_persist<T, 10>(ctx, b, tw, nb); break;
            default: break;
            }
        }
        if (rc == KOFFT_ERR_UNSUPPORTED) {
            if (b.blk_r != 0) return KOFFT_ERR_UNSUPPORTED;  // (never: the blocked layout is only chosen where the rows kernel runs)
            rc = launch_sub<T>(ctx, b, tw, L3, nb << LP, false);
        }
        return rc;
    };
    if (ctx->big_tmp_bytes < need) {
        if (ctx->big_tmp_external) return KOFFT_ERR_ALLOC;
        if (ctx->big_tmp) KOFFT_HIP_TRY(ctx, hipFree(ctx->big_tmp));
        ctx->big_tmp = nullptr;
        ctx->big_tmp_bytes = 0;
        ctx->big_probe_n = 0;
        // (Round 4, tools/exp_c64_alloc.py: a physically contiguous buffer -- hipExtMallocWithFlags(hipDeviceMallocContiguous) -- made
        // the last factor's reads fast on one box (188-192 us per 512 MiB chunk on 16 of 16 buffers against 217-231 us on most plain
        // ones) but not on the next (4 of 5 slow), slowed the first factor's writes by 10 % on both, and small contiguous buffers
        // returned stale data to the plain loads of the narrow-tile kernels: not used.)
        //
        // Round 6: the PLACEMENT of this buffer decides whether the last factor reads it at 188 or at 225 us per
        // 512 MiB chunk (a property of the allocation, for its lifetime; 2-3 of 8 allocations are slow).  The library owns
        // the buffer, so it chooses: K candidate allocations, this call's own factor kernels timed on one full chunk of scratch data
        // through each (one warm pass, then the better of two), the fastest kept, the rest freed.  Only where it matters and costs
        // little next to the call: full chunks of at least 128 MiB.  Once per (context, buffer size); KOFFT_HIP_BIG_PROBE=0 / 1: off.
        const int K = (chunk * xf_bytes >= (size_t(128) << 20) && batch >= chunk) ? ctx->big_probe : 1;
        if (K >= 2) {
            const int prc = big_probe_pick(ctx, need, chunk * xf_bytes, K, [&](void *in_s, void *out_s, void *cand, hipEvent_t ev_mid) {
                return run_chunk(static_cast<const cpx<T> *>(in_s), static_cast<cpx<T> *>(out_s), static_cast<cpx<T> *>(cand), chunk, ev_mid);
            });
            if (prc) return prc;
        }
        if (!ctx->big_tmp) {
            KOFFT_HIP_TRY(ctx, hipMalloc(&ctx->big_tmp, need));
            ctx->big_tmp_bytes = need;
        }
    }
    cpx<T> *mid0 = static_cast<cpx<T> *>(ctx->big_tmp);
#ifdef KOFFT_EXP_TMP_PRINT
    fprintf(stderr, "kofft big_tmp %p mid %p\n", ctx->big_tmp, (void *)mid0);
#endif
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        rc = run_chunk(in_base + b0 * in_row, out_base + b0 * out_row, mid0, nb, nullptr);
        if (rc) return rc;
    }
    return KOFFT_OK;
}

template <typename T, bool INVERSE>
int fft_big_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch)
{
    return fft_big_core<T, BigColsIO<T, INVERSE>, BigRowsIO<T, INVERSE>>(
        ctx, reinterpret_cast<const cpx<T> *>(d_in), n, reinterpret_cast<cpx<T> *>(d_out), n, n, batch, [](BigColsIO<T, INVERSE> &) {},
        [](BigRowsIO<T, INVERSE> &) {});
}

// The m-point transform of rfft_direct's packed row z[i] = (x[2i], x[2i+1]) (rfft.rs:444-447) with the row window of the
// batched entry folded into the first factor's load (PRE_WINDOW): m a power of two beyond the single-workgroup sizes.
template <typename T>
int fft_big_windowed_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, const T *d_window, size_t m, size_t batch)
{
    using Cols = BigColsIO<T, false, PRE_WINDOW>;
    using Rows = BigRowsIO<T, false>;
    return fft_big_core<T, Cols, Rows>(
        ctx, reinterpret_cast<const cpx<T> *>(d_in), m, reinterpret_cast<cpx<T> *>(d_out), m, m, batch,
        [&](Cols &c) { c.pre_tab = reinterpret_cast<const cpx<T> *>(d_window); }, [](Rows &) {});
}

// ---------------------------------------------------------------------------------
// non-power-of-two lengths: Bluestein (fft.rs:1088-1132)
// ---------------------------------------------------------------------------------
template <typename T>
int get_bluestein(kofft_hip_ctx *ctx, size_t n, size_t m, const cpx<T> **chirp, const cpx<T> **bfft)
{
    const int kc = sizeof(T) == 4 ? 5 : 6, kb = sizeof(T) == 4 ? 7 : 8;
    auto ic = ctx->tables.find(std::make_pair(kc, n));
    auto ib = ctx->tables.find(std::make_pair(kb, n));
    if (ic != ctx->tables.end() && ib != ctx->tables.end()) {
        *chirp = static_cast<const cpx<T> *>(ic->second);
        *bfft = static_cast<const cpx<T> *>(ib->second);
        return KOFFT_OK;
    }
    std::vector<T> hc(2 * n), hb(2 * m);
    if constexpr (sizeof(T) == 4) kofft_tables::bluestein_f32(n, m, (float *)hc.data(), (float *)hb.data());
    else kofft_tables::bluestein_f64(n, m, (double *)hc.data(), (double *)hb.data());
    void *dc = nullptr, *db = nullptr;
    // every failure path below frees both tables (nothing is cached until the last step has succeeded)
    auto fail = [&](const char *what, hipError_t e) {
        if (dc) (void)hipFree(dc);
        if (db) (void)hipFree(db);
        if (e != hipSuccess) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    };
    hipError_t e = hipMalloc(&dc, hc.size() * sizeof(T));
    if (e == hipSuccess) e = hipMalloc(&db, hb.size() * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail("bluestein tables: hipMalloc", e);
        return KOFFT_ERR_HIP;
    }
    e = hipMemcpy(dc, hc.data(), hc.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, hb.data(), hb.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        fail("bluestein tables: upload", e);
        return KOFFT_ERR_HIP;
    }
    // b_fft = fft(b) with the ordinary power-of-two path (fft.rs:425-427)
    int rc = fft_dev<T>(ctx, static_cast<T *>(db), static_cast<T *>(db), m, 1, 0);
    if (rc) {
        fail("", hipSuccess);
        return rc;
    }
    // the table is cached for every later call, on whatever stream that call uses: finish it first
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        fail("bluestein tables: fft(b)", e);
        return KOFFT_ERR_HIP;
    }
    ctx->tables[std::make_pair(kc, n)] = dc;
    ctx->tables[std::make_pair(kb, n)] = db;
    *chirp = static_cast<const cpx<T> *>(dc);
    *bfft = static_cast<const cpx<T> *>(db);
    return KOFFT_OK;
}

// The whole arm in one launch (bluestein_wg_kernel): m = 2^5 .. 2^12 (c32) / 2^10 (c64).
template <typename T, int L, bool INVERSE>
int launch_bluestein_wg(kofft_hip_ctx *ctx, const cpx<T> *in, cpx<T> *out, const cpx<T> *chirp, const cpx<T> *bfft, const cpx<T> *tw,
                        size_t n, size_t batch, T scale_m, T scale_n)
{
    constexpr int RL = rl_for(L);
    constexpr int BLOCK = block_for(L);
    constexpr int TPT = (1 << L) >> RL;
    constexpr int XPB = BLOCK / TPT;
    constexpr size_t lds = lds_wg_bytes<T, false, false, XPB>(1 << L);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = bluestein_wg_kernel<T, L, RL, BLOCK, INVERSE>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    const size_t blocks = (batch + XPB - 1) / XPB;
    if (blocks > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, in, out, chirp, bfft, tw, (int)n, scale_m, scale_n, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// The same arm as a persistent kernel (bluestein_persist_kernel): batches that give every workgroup several transforms.
// workgroups per CU, within what the registers allow (f32, 16 points per thread: 166-208 VGPRs; 8: 94-100; 4: 57-59)
// (8 points per thread from m = 1024 on -- four passes, three or four workgroups per CU -- measured: m = 2048 -3 %, m = 1024 +5 %, m = 4096 +11 % of time)
// f64: 8 points per thread at most (16: the twiddles alone are 176 registers)
template <typename T>
constexpr int blue_persist_rl(int L) { return sizeof(T) == 8 ? (rl_for(L) > 3 ? 3 : rl_for(L)) : (L >= 13 ? 4 : rl_for(L)); }
template <typename T>
constexpr int blue_persist_block(int L) { return ((1 << L) >> blue_persist_rl<T>(L)) > 256 ? ((1 << L) >> blue_persist_rl<T>(L)) : 256; }
template <typename T>
constexpr int blue_persist_wg(int L)
{
    // measured, same box, ms per 65.5 M points: m = 32 / 64 six -> eight 0.41 -> 0.38 / 0.42 -> 0.40; m = 128 four -> five 0.38 -> 0.37; m = 512 four -> five
    // 0.385 -> 0.395; m = 256 / 1024 two -> three 0.39 -> 0.36 / 0.455 -> 0.415; m = 2048 / 4096 need 208 / 194 registers (two)
    if (blue_persist_block<T>(L) > 256) return 1;  // m = 8192 (f32), 4096 (f64): 512 threads, two wavefronts per SIMD
    if (sizeof(T) == 8) return blue_persist_rl<T>(L) >= 3 ? 2 : 4;
    return blue_persist_rl<T>(L) >= 4 ? (L <= 10 ? 3 : 2) : blue_persist_rl<T>(L) == 3 ? (L == 7 ? 5 : 4) : 8;
}
// from one full round of the grid on (measured: n = 1000 x 1100 0.018 -> 0.016 ms, x 2100 0.029 -> 0.023, x 4000 0.042 -> 0.037; n = 250 x 9000
// 0.022 -> 0.018; n = 60 x 30000 0.017 -> 0.014; with a threshold of four rounds these ran the one-workgroup-per-XPB-transforms kernel)
#ifndef KOFFT_BLUE_PERSIST_MIN_ITERS
#define KOFFT_BLUE_PERSIST_MIN_ITERS 1
#endif
template <typename T, int L>
constexpr bool blue_persist_ok()
{
    return L >= 5 && L <= (sizeof(T) == 4 ? 13 : 12);
}
template <typename T, int L, bool INVERSE, class SRC>
int launch_bluestein_persist_src(kofft_hip_ctx *ctx, const SRC &src, cpx<T> *out, const cpx<T> *chirp, const cpx<T> *bfft, const cpx<T> *tw,
                                 size_t n, size_t batch, T scale_m, T scale_n)
{
    constexpr int RL = blue_persist_rl<T>(L);
    constexpr int BLOCK = blue_persist_block<T>(L);
    constexpr int TPT = (1 << L) >> RL;
    constexpr int XPB = BLOCK / TPT;
    constexpr size_t lds = (size_t)XPB * lds_elems(1 << L) * sizeof(cpx<T>);
    static_assert(lds * blue_persist_wg<T>(L) <= 160 * 1024, "LDS budget");
    // (the STFT source's 64-bit sample positions and window values cost registers: one workgroup per CU less where the row form is at its limit)
    constexpr bool kRows = std::is_same<SRC, BlueRowsSrc<T, INVERSE>>::value;
    constexpr int WG = (kRows || blue_persist_wg<T>(L) <= 2 || RL == 2 || L == 9) ? blue_persist_wg<T>(L) : blue_persist_wg<T>(L) - 1;
    auto kern = bluestein_persist_kernel<T, L, RL, BLOCK, WG, INVERSE, SRC>;
    {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    size_t blocks = (size_t)ctx->num_cus * WG;
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < 1) blocks = 1;
    const size_t need = (batch + XPB - 1) / XPB;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, src, out, chirp, bfft, tw, (int)n, scale_m, scale_n, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}
template <typename T, int L, bool INVERSE>
int launch_bluestein_persist(kofft_hip_ctx *ctx, const cpx<T> *in, cpx<T> *out, const cpx<T> *chirp, const cpx<T> *bfft, const cpx<T> *tw,
                             size_t n, size_t batch, T scale_m, T scale_n)
{
    const BlueRowsSrc<T, INVERSE> src{in, (int)n, batch};
    return launch_bluestein_persist_src<T, L, INVERSE>(ctx, src, out, chirp, bfft, tw, n, batch, scale_m, scale_n);
}
template <typename T, int L>
bool blue_persist_pays(const kofft_hip_ctx *ctx, size_t batch)
{
    constexpr int XPB = blue_persist_block<T>(L) / ((1 << L) >> blue_persist_rl<T>(L));
    return ctx->blue_persist && batch >= (size_t)ctx->num_cus * blue_persist_wg<T>(L) * XPB * KOFFT_BLUE_PERSIST_MIN_ITERS;
}

// STFT with a window length that is not a power of two (stft.rs:91-103 calls fft.fft(frame) for any win_len): the frames through the
// persistent Bluestein kernel with the framing product on its loads.  *done = false: the caller takes the composed route.
#ifdef KOFFT_BLUE_STFT_UNIT
int stft_bluestein_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len, const float *d_window, size_t n, size_t start0, size_t hop,
                       float *d_out, size_t count, bool *done)
{
    using T = float;
    *done = false;
    if (!(ctx->blue_fused && ctx->blue_one_kernel && ctx->blue_persist) || is_pow2(n) || n < 3) return KOFFT_OK;
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    const int L = ilog2(m);
    if (L < 5 || L > 13) return KOFFT_OK;
    const cpx<T> *chirp = nullptr, *bfft = nullptr, *tw = nullptr;
    int rc = get_bluestein<T>(ctx, n, m, &chirp, &bfft);
    if (rc) return rc;
    rc = get_table<T>(ctx, Kind<T>::tw, m, &tw);
    if (rc) return rc;
    const T sm = (T)1 / (T)(float)m, sn = (T)1 / (T)(float)n;
    const BlueStftSrc src{d_signal, d_window, len, hop, start0, (int)n};
    cpx<T> *dst = reinterpret_cast<cpx<T> *>(d_out);
    switch (L) {
#define KOFFT_CASE(LL)                                                                                                  \
    case LL:                                                                                                            \
        if (!blue_persist_pays<T, LL>(ctx, count)) return KOFFT_OK;                                                     \
        rc = launch_bluestein_persist_src<T, LL, false>(ctx, src, dst, chirp, bfft, tw, n, count, sm, sn);              \
        break;
        KOFFT_CASE(5)
        KOFFT_CASE(6)
        KOFFT_CASE(7)
        KOFFT_CASE(8)
        KOFFT_CASE(9)
        KOFFT_CASE(10)
        KOFFT_CASE(11)
        KOFFT_CASE(12)
        KOFFT_CASE(13)
#undef KOFFT_CASE
    default: return KOFFT_OK;
    }
    if (rc) return rc;
    *done = true;
    return KOFFT_OK;
}
#endif

// The two fused launches (BlueFirstIO / BlueSecondIO through a scratch) exist at the m no one-launch kernel serves whatever the batch
// (round 6, library diet: the general dispatch instantiated both policies at every size from 1 to 2^14 -- 120 kernels, 100 of them
// reachable only with KOFFT_HIP_BLUESTEIN_ONE=0): m = 8 / 16 (n = 3 .. 8: the straight-line kernels), and where bluestein_wg_kernel
// stops: c32 m = 8192 (batches the persistent form does not take) / 16384, c64 m = 2048 / 4096 (likewise) / 8192.  Any other m:
// KOFFT_ERR_UNSUPPORTED before anything is launched -- the caller takes the separate pointwise kernels.
template <typename T, class IO>
int dispatch_blue(kofft_hip_ctx *ctx, const IO &io, size_t m, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    const int L = ilog2(m);
    if (L == 3) return launch_small<T, 8, EPI_STORE>(ctx, io, batch);
    if (L == 4) return launch_small<T, 16, EPI_STORE>(ctx, io, batch);
    if (L < (sizeof(T) == 4 ? 13 : 11) || L > max_log2<T>()) return KOFFT_ERR_UNSUPPORTED;
    const cpx<T> *tw = nullptr;
    const int rc = get_table<T>(ctx, Kind<T>::tw, m, &tw);
    if (rc) return rc;
    switch (L) {
    case 11:
        if constexpr (sizeof(T) == 8) return launch_wg<T, 11, EPI_STORE>(ctx, io, tw, batch);
        break;
    case 12:
        if constexpr (sizeof(T) == 8) return launch_wg<T, 12, EPI_STORE>(ctx, io, tw, batch);
        break;
    case 13: return launch_wg<T, 13, EPI_STORE>(ctx, io, tw, batch);
    case 14:
        if constexpr (sizeof(T) == 4) return launch_wg<T, 14, EPI_STORE>(ctx, io, tw, batch);
        break;
    default: break;
    }
    return KOFFT_ERR_UNSUPPORTED;
}

template <typename T, bool INVERSE>
int fft_bluestein_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch)
{
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;  // (2n-1).next_power_of_two()
    const cpx<T> *chirp = nullptr, *bfft = nullptr;
    int rc = get_bluestein<T>(ctx, n, m, &chirp, &bfft);
    if (rc) return rc;
    if (ctx->blue_fused && ctx->blue_one_kernel) {
        const int L = ilog2(m);
        // measured (tools/bench_bluestein.py): one workgroup per XPB transforms wins 15..50 % over two launches up to m = 4096 (c32) /
        // 1024 (c64); beyond, that kernel needs more than 256 registers and the two launches are faster (n = 4095: 1.09 vs 1.70 ms).
        // The persistent form (large batches) reaches m = 8192 (c32) / 4096 (c64)
        if (L >= 5 && L <= 13) {
            const cpx<T> *tw = nullptr;
            rc = get_table<T>(ctx, Kind<T>::tw, m, &tw);
            if (rc) return rc;
            const T sm = (T)1 / (T)(float)m, sn = (T)1 / (T)(float)n;
            const cpx<T> *src = reinterpret_cast<const cpx<T> *>(d_in);
            cpx<T> *dst = reinterpret_cast<cpx<T> *>(d_out);
            switch (L) {
#define KOFFT_CASE(LL)                                                                                                  \
    case LL:                                                                                                            \
        if constexpr (blue_persist_ok<T, LL>())                                                                         \
            if (blue_persist_pays<T, LL>(ctx, batch)) return launch_bluestein_persist<T, LL, INVERSE>(ctx, src, dst, chirp, bfft, tw, n, batch, sm, sn); \
        if constexpr (LL <= (sizeof(T) == 4 ? 12 : 10)) return launch_bluestein_wg<T, LL, INVERSE>(ctx, src, dst, chirp, bfft, tw, n, batch, sm, sn); \
        break;
                KOFFT_CASE(5)
                KOFFT_CASE(6)
                KOFFT_CASE(7)
                KOFFT_CASE(8)
                KOFFT_CASE(9)
                KOFFT_CASE(10)
                KOFFT_CASE(11)
                KOFFT_CASE(12)
                KOFFT_CASE(13)
#undef KOFFT_CASE
            default: break;
            }
        }
    }
    const size_t xf_bytes = m * sizeof(cpx<T>);
    size_t chunk = (size_t(512) << 20) / xf_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (ctx->blue_tmp_bytes < chunk * xf_bytes) {
        if (ctx->blue_tmp) KOFFT_HIP_TRY(ctx, hipFree(ctx->blue_tmp));
        ctx->blue_tmp = nullptr;
        ctx->blue_tmp_bytes = 0;
        KOFFT_HIP_TRY(ctx, hipMalloc(&ctx->blue_tmp, chunk * xf_bytes));
        ctx->blue_tmp_bytes = chunk * xf_bytes;
    }
    cpx<T> *a = static_cast<cpx<T> *>(ctx->blue_tmp);
    const T scale_m = (T)1 / (T)(float)m, scale_n = (T)1 / (T)(float)n;
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        const cpx<T> *src = reinterpret_cast<const cpx<T> *>(d_in) + b0 * n;
        cpx<T> *dst = reinterpret_cast<cpx<T> *>(d_out) + b0 * n;
        if (ctx->blue_fused && m <= (size_t(1) << max_log2<T>())) {
            BlueFirstIO<T, INVERSE> io1{{}, src, a, chirp, bfft, (int)n, (int)m};
            rc = dispatch_blue<T>(ctx, io1, m, nb);
            if (rc == KOFFT_OK) {
                BlueSecondIO<T, INVERSE> io2{{}, a, dst, chirp, (int)n, (int)m, scale_m, scale_n};
                rc = dispatch_blue<T>(ctx, io2, m, nb);
                if (rc) return rc;
                continue;
            }
            if (rc != KOFFT_ERR_UNSUPPORTED) return rc;  // (unsupported: no fused pair at this m -- the separate kernels below)
        }
        if (ctx->blue_fused && is_pow2(m) && m > (size_t(1) << max_log2<T>())) {
            // m beyond one workgroup's transform (round 3): the three pointwise steps ride on the factor kernels -- x * chirp and
            // the zero padding on the first factor's load, * fft(b) + conj on the first transform's last store, conj * 1/m * chirp
            // on the second one's (BigColsIO PRE_CHIRP, BigRowsIO POST_BLUE_MID / POST_BLUE_OUT): 4 passes over the padded
            // buffer instead of 7.  Same expressions per element as the three kernels below.
            using Cols1 = BigColsIO<T, INVERSE, PRE_CHIRP>;
            using Rows1 = BigRowsIO<T, false, POST_BLUE_MID>;
            rc = fft_big_core<T, Cols1, Rows1>(
                ctx, src, n, a, m, m, nb,
                [&](Cols1 &c) { c.pre_tab = chirp; c.n_in = (unsigned)n; },
                [&](Rows1 &r) { r.post_tab = bfft; });
            if (rc) return rc;
            using Cols2 = BigColsIO<T, false, PRE_NONE>;
            using Rows2 = BigRowsIO<T, INVERSE, POST_BLUE_OUT>;
            rc = fft_big_core<T, Cols2, Rows2>(
                ctx, a, m, dst, n, m, nb, [](Cols2 &) {},
                [&](Rows2 &r) { r.post_tab = chirp; r.n_out = (unsigned)n; r.scale = scale_m; r.scale_out = scale_n; });
            if (rc) return rc;
            continue;
        }
        const size_t tm = nb * m, tn = nb * n;
        hipLaunchKernelGGL((bluestein_pre_kernel<T, INVERSE>), dim3((unsigned)((tm + 255) / 256)), dim3(256), 0, ctx->stream, src, a,
                           chirp, n, m, tm);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
        rc = fft_dev<T>(ctx, reinterpret_cast<T *>(a), reinterpret_cast<T *>(a), m, nb, 0);
        if (rc) return rc;
        hipLaunchKernelGGL((bluestein_mid_kernel<T>), dim3((unsigned)((tm + 255) / 256)), dim3(256), 0, ctx->stream, a, bfft, m, tm);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
        rc = fft_dev<T>(ctx, reinterpret_cast<T *>(a), reinterpret_cast<T *>(a), m, nb, 0);
        if (rc) return rc;
        hipLaunchKernelGGL((bluestein_post_kernel<T, INVERSE>), dim3((unsigned)((tn + 255) / 256)), dim3(256), 0, ctx->stream, a, dst,
                           chirp, n, m, tn, scale_m, scale_n);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
    }
    return KOFFT_OK;
}

// ---------------------------------------------------------------------------------
// typed entry points behind the C ABI
// ---------------------------------------------------------------------------------
template <typename T>
int fft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch, int inverse)
{
    // argument checks come first and need no device, so the reference's error order is testable anywhere
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;  // fft.rs:1056 / 1136
    if (n > (size_t(1) << (is_pow2(n) ? max_log2_big<T>() : max_log2_big<T>() - 1))) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!is_pow2(n))  // fft.rs:1083-1132
        return inverse ? fft_bluestein_dev<T, true>(ctx, d_in, d_out, n, batch) : fft_bluestein_dev<T, false>(ctx, d_in, d_out, n, batch);
    if (n == (size_t(2) << max_log2<T>()) && ctx->use_regfile && batch >= (size_t)ctx->num_cus * 2) {
        // 256 KiB per transform: one pass over HBM with the transform in the CU's register file (fft_regfile.hip.h)
        const cpx<T> *tw = nullptr;
        const int trc = get_table<T>(ctx, Kind<T>::tw, n, &tw);
        if (trc) return trc;
        const T scale = (T)1 / (T)(float)n;  // fft.rs:1167
        // (c32 as 2^7 x 2^8 -- 128-byte load runs, 64-byte store runs -- measured 0.374-0.378 against 0.396-0.402 for 2^8 x 2^7)
        constexpr int RLA = sizeof(T) == 4 ? 8 : 7, RLB = 7, RQB0 = 4;
        if (inverse) {
            ComplexIO<T, true> io{{}, reinterpret_cast<const cpx<T> *>(d_in), reinterpret_cast<cpx<T> *>(d_out), (int)n, scale};
            return launch_regfile<T, RLA, RLB, RQB0>(ctx, io, tw, batch);
        }
        ComplexIO<T, false> io{{}, reinterpret_cast<const cpx<T> *>(d_in), reinterpret_cast<cpx<T> *>(d_out), (int)n, scale};
        return launch_regfile<T, RLA, RLB, RQB0>(ctx, io, tw, batch);
    }
    if (n > (size_t(1) << max_log2<T>()))
        return inverse ? fft_big_dev<T, true>(ctx, d_in, d_out, n, batch) : fft_big_dev<T, false>(ctx, d_in, d_out, n, batch);
    if (n == 1) {  // fft.rs:1059 / 1139: nothing to do
        if (d_in != d_out)
            KOFFT_HIP_TRY(ctx, hipMemcpyAsync(d_out, d_in, batch * 2 * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
        return KOFFT_OK;
    }
    const T scale = (T)1 / (T)(float)n;  // fft.rs:1167
    if (inverse) {
        ComplexIO<T, true> io{{}, reinterpret_cast<const cpx<T> *>(d_in), reinterpret_cast<cpx<T> *>(d_out), (int)n, scale};
        return dispatch<T, EPI_STORE>(ctx, io, n, batch);
    }
    ComplexIO<T, false> io{{}, reinterpret_cast<const cpx<T> *>(d_in), reinterpret_cast<cpx<T> *>(d_out), (int)n, scale};
    return dispatch<T, EPI_STORE>(ctx, io, n, batch);
}


// ---------------------------------------------------------------------------------
// ScalarFftImpl::fft_radix4 (fft.rs:1455-1548), the reference's bytes (fft_radix4.hip.h): what fft_with_strategy(.., Radix4)
// runs (fft.rs:1356).  inverse = FftPlan::ifft's loop around that arm (fft.rs:2040-2055): conj, fft_radix4, conj * 1/n.
// ---------------------------------------------------------------------------------
template <typename T>
int fft_radix4_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch, int inverse)
{
    if (batch == 0) return KOFFT_OK;
    // fft.rs:1457-1460: anything but a power of four falls back to fft() (n == 0 -> EmptyInput there); around it the plan's
    // conj / conj * scale loop is ifft()'s arithmetic (fft.rs:1163-1172)
    if (!is_pow2(n) || (ilog2(n) & 1)) return fft_dev<T>(ctx, d_in, d_out, n, batch, inverse);
    if (n > (size_t(1) << max_log2_big<T>())) return KOFFT_ERR_UNSUPPORTED;  // the limit of fft_dev itself
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const cpx<T> *in = reinterpret_cast<const cpx<T> *>(d_in);
    cpx<T> *out = reinterpret_cast<cpx<T> *>(d_out);
    if (n == 1) {  // no swap, no stage; the plan's loop: im = -(-im), re * 1, im * 1 -- the value itself
        if (d_in != d_out) KOFFT_HIP_TRY(ctx, hipMemcpyAsync(d_out, d_in, batch * 2 * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
        return KOFFT_OK;
    }
    // tables: (kind 9 / 10, n) = permutation, (11 / 12, n) = stage triples; O(n) host work once per (context, n)
    const int kp = sizeof(T) == 4 ? 9 : 10, kw = sizeof(T) == 4 ? 11 : 12;
    auto ip = ctx->tables.find(std::make_pair(kp, n));
    auto iw = ctx->tables.find(std::make_pair(kw, n));
    if (ip == ctx->tables.end() || iw == ctx->tables.end()) {
        const size_t triples = kofft_tables::radix4_triples(n);
        std::vector<unsigned> hp(n);
        std::vector<T> hw(6 * (triples ? triples : 1));
        if constexpr (sizeof(T) == 4) kofft_tables::radix4_f32(n, hp.data(), (float *)hw.data());
        else kofft_tables::radix4_f64(n, hp.data(), (double *)hw.data());
        void *dp = nullptr, *dw = nullptr;
        hipError_t e = hipMalloc(&dp, hp.size() * sizeof(unsigned));
        if (e == hipSuccess) e = hipMalloc(&dw, hw.size() * sizeof(T));
        if (e == hipSuccess) e = hipMemcpy(dp, hp.data(), hp.size() * sizeof(unsigned), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dw, hw.data(), hw.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (dp) (void)hipFree(dp);
            if (dw) (void)hipFree(dw);
            ctx->last_error = std::string("radix4 tables: ") + hipGetErrorString(e);
            return KOFFT_ERR_HIP;
        }
        ctx->tables[std::make_pair(kp, n)] = dp;
        ctx->tables[std::make_pair(kw, n)] = dw;
        ip = ctx->tables.find(std::make_pair(kp, n));
        iw = ctx->tables.find(std::make_pair(kw, n));
    }
    const unsigned *perm = static_cast<const unsigned *>(ip->second);
    const cpx<T> *w = static_cast<const cpx<T> *>(iw->second);
    const T scale = (T)1 / (T)(float)n;  // fft.rs:2048
    auto grid = [](size_t quads) { return dim3((unsigned)((quads + 255) / 256)); };
    auto first = [&](const cpx<T> *src, cpx<T> *dst, size_t quads) {
        if (inverse) hipLaunchKernelGGL((radix4_first_kernel<T, true>), grid(quads), dim3(256), 0, ctx->stream, src, dst, perm, n, quads, scale);
        else hipLaunchKernelGGL((radix4_first_kernel<T, false>), grid(quads), dim3(256), 0, ctx->stream, src, dst, perm, n, quads, scale);
    };
    if (n == 4) {  // the permutation is the identity and every thread owns its quad: in place is safe
        const size_t quads = batch;
        if (quads > 0x7fffffffULL * 256) return KOFFT_ERR_UNSUPPORTED;
        first(in, out, quads);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
        return KOFFT_OK;
    }
    const size_t xf_bytes = n * sizeof(cpx<T>);
    size_t chunk = (size_t(512) << 20) / xf_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const int prc = ensure_real_tmp(ctx, chunk * xf_bytes);
    if (prc) return prc;
    cpx<T> *tmp = static_cast<cpx<T> *>(ctx->real_tmp);
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (batch - b0 < chunk) ? batch - b0 : chunk, quads = nb * (n / 4);
        first(in + b0 * n, tmp, quads);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
        size_t off = 0;
        for (size_t len = 16; len <= n; len <<= 2) {
            cpx<T> *dst = (len == n) ? out + b0 * n : tmp;  // the last stage lands in the caller's buffer
            if (len == n && inverse)
                hipLaunchKernelGGL((radix4_stage_kernel<T, true>), grid(quads), dim3(256), 0, ctx->stream, tmp, dst, w + 3 * off, len, n, quads, scale);
            else
                hipLaunchKernelGGL((radix4_stage_kernel<T, false>), grid(quads), dim3(256), 0, ctx->stream, tmp, dst, w + 3 * off, len, n, quads, scale);
            KOFFT_HIP_TRY(ctx, hipGetLastError());
            off += len / 4;
        }
    }
    return KOFFT_OK;
}

}  // namespace host
}  // namespace kofft
