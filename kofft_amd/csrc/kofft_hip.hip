// kofft_hip.hip -- context, planner cache, launch dispatch and the extern "C" ABI of
// include/kofft_hip.h.  gfx950 only; compiled with -ffp-contract=off.
#include "../../include/kofft_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "fft_big.hip.h"
#include "fft_persist.hip.h"
#include "fft_wg.hip.h"
#include "tables.h"

using namespace kofft;

// ---------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------
struct kofft_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    bool use_persist = true;  // KOFFT_HIP_NO_PERSIST=1 forces the generic kernels (A/B measurements, tests)
    int persist_grid_pct = 0; // KOFFT_HIP_PERSIST_GRID_PCT: scale the persistent grids (measurements only)
    bool big_two_only = false; // KOFFT_HIP_BIG_TWO_FACTORS=1: never split into three factors (A/B measurements)
    bool blue_fused = true;    // KOFFT_HIP_BLUESTEIN_FUSED=0: pointwise steps as separate kernels at every size
    bool nd_transpose = true;  // KOFFT_HIP_ND_TRANSPOSE=0: long strided axes through the strided kernel
    int nd_transpose_min = 4096;  // KOFFT_HIP_ND_TRANSPOSE_MIN: shortest axis that takes the transpose route (measured: 1024 loses, 2048 ties)
    bool zero_copy = true;     // KOFFT_HIP_ZERO_COPY=0: small host calls through staged copies like large ones
    bool host_pipeline = true; // KOFFT_HIP_HOST_PIPELINE=0: host-pointer batches in one upload / kernel / download
    int host_chunks = 0;       // KOFFT_HIP_HOST_CHUNKS: pieces of a pipelined host batch (default 8)
    bool persist_small = true; // KOFFT_HIP_PERSIST_SMALL=0: n = 128, 256 on the generic kernels (A/B measurements)
    std::string last_error;
    // planner caches: (kind, n) -> device table.  kind 0/1 = FftPlanner twiddles f32/f64,
    // 2/3 = RfftPlanner post-pass table f32/f64.
    std::map<std::pair<int, size_t>, void *> tables;
    // staging for the host-pointer entry points
    void *stage[3] = {nullptr, nullptr, nullptr};
    size_t stage_bytes[3] = {0, 0, 0};
    // intermediate of the two-factor large-n path (fft_big.hip.h): `big_chunk` transforms at a time
    void *big_tmp = nullptr;
    size_t big_tmp_bytes = 0;
    // small host-pointer calls (one frame, one transform): a pinned, device-mapped buffer the kernels read and write
    // directly over PCIe -- one launch and one synchronisation instead of two staged copies around them
    void *pinned = nullptr;      // host address
    void *pinned_dev = nullptr;  // the same memory as the device sees it
    size_t pinned_bytes = 0;
    void *blue_tmp = nullptr;  // zero-padded work buffer of the Bluestein arm
    size_t blue_tmp_bytes = 0;
    size_t big_chunk_bytes = size_t(2048) << 20;  // KOFFT_HIP_BIG_CHUNK_MB; measured: small chunks do not profit from the Infinity Cache, larger launches overlap better
};

namespace {

#define KOFFT_HIP_TRY(ctx, expr)                                                              \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);            \
            return KOFFT_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)

inline bool is_pow2(size_t n) { return n != 0 && (n & (n - 1)) == 0; }
inline int ilog2(size_t n)
{
    int l = 0;
    while ((size_t(1) << l) < n) ++l;
    return l;
}

template <typename T> struct Kind;
template <> struct Kind<float> { static constexpr int tw = 0, rt = 2; };
template <> struct Kind<double> { static constexpr int tw = 1, rt = 3; };

template <typename T>
int get_table(kofft_hip_ctx *ctx, int kind, size_t n, const cpx<T> **out)
{
    auto key = std::make_pair(kind, n);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = static_cast<const cpx<T> *>(it->second);
        return KOFFT_OK;
    }
    const bool is_rfft = kind >= 2;
    const size_t entries = is_rfft ? n : n / 2;
    std::vector<T> host(2 * (entries ? entries : 1));
    if (is_rfft) {
        if constexpr (sizeof(T) == 4) kofft_tables::rfft_table_f32(n, (float *)host.data());
        else kofft_tables::rfft_table_f64(n, (double *)host.data());
    } else {
        if constexpr (sizeof(T) == 4) kofft_tables::twiddles_f32(n, (float *)host.data());
        else kofft_tables::twiddles_f64(n, (double *)host.data());
    }
    void *d = nullptr;
    KOFFT_HIP_TRY(ctx, hipMalloc(&d, host.size() * sizeof(T)));
    // synchronous copy: tables are built once per (context, n), never in a timed region
    hipError_t e = hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        ctx->last_error = std::string("table upload: ") + hipGetErrorString(e);
        return KOFFT_ERR_HIP;
    }
    ctx->tables[key] = d;
    *out = static_cast<const cpx<T> *>(d);
    return KOFFT_OK;
}

constexpr size_t kZeroCopyMax = size_t(512) << 10;  // bytes per direction up to which a host call goes zero-copy
// (measured per-call latency, host memory: n = 64 31 -> 17 us, 1024 33 -> 20, 4096 36 -> 22, 65536 95 -> 79; 1 MiB: no gain)

int ensure_pinned(kofft_hip_ctx *ctx, size_t bytes)
{
    if (ctx->pinned_bytes >= bytes) return KOFFT_OK;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = ctx->pinned_dev = nullptr;
    ctx->pinned_bytes = 0;
    const size_t want = bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes;
    if (hipHostMalloc(&ctx->pinned, want, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        ctx->pinned = nullptr;
        return KOFFT_ERR_ALLOC;
    }
    if (hipHostGetDevicePointer(&ctx->pinned_dev, ctx->pinned, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        return KOFFT_ERR_ALLOC;
    }
    ctx->pinned_bytes = want;
    return KOFFT_OK;
}

int ensure_stage(kofft_hip_ctx *ctx, int which, size_t bytes)
{
    if (ctx->stage_bytes[which] >= bytes) return KOFFT_OK;
    if (ctx->stage[which]) KOFFT_HIP_TRY(ctx, hipFree(ctx->stage[which]));
    ctx->stage[which] = nullptr;
    ctx->stage_bytes[which] = 0;
    KOFFT_HIP_TRY(ctx, hipMalloc(&ctx->stage[which], bytes));
    ctx->stage_bytes[which] = bytes;
    return KOFFT_OK;
}

// ---------------------------------------------------------------------------------
// launch geometry
// ---------------------------------------------------------------------------------
#ifndef KOFFT_RL_BIG
#define KOFFT_RL_BIG 5
#endif
// threads per transform >= 8 (c64) / 16 (c32): every load / store instruction covers whole 128-byte lines per transform
// (A/B on one box: n = 32 c64 0.65 -> 0.79 of the roofline, n = 64 c32 0.64 -> 0.71, n = 128 c32 0.60 -> 0.70.)
constexpr int rl_for(int L) { return (L == 5 || L == 6) ? 2 : (L == 7 || L == 9) ? 3 : (L >= 13 ? KOFFT_RL_BIG : 4); }
constexpr int block_for(int L)
{
    const int tpt = (1 << L) >> rl_for(L);
    return tpt > 256 ? tpt : 256;
}
template <typename T> constexpr int max_log2();
template <> constexpr int max_log2<float>() { return 14; }
template <> constexpr int max_log2<double>() { return 13; }

template <typename T, int L, int EPI, class IO, int BLOCK_OVERRIDE = 0>
int launch_wg(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t batch)
{
    constexpr int RL = rl_for(L);
    constexpr int BLOCK = BLOCK_OVERRIDE ? BLOCK_OVERRIDE : block_for(L);
    constexpr int TPT = (1 << L) >> RL;
    constexpr int XPB = BLOCK / TPT;
    constexpr size_t lds = lds_wg_bytes<T, wg_split_lds<T, L, EPI, IO>(), IO::kSlotMinor, XPB>(1 << L);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fft_wg_kernel<T, L, RL, BLOCK, EPI, IO>;
    if (lds > 64 * 1024) {
        KOFFT_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const size_t blocks = (batch + XPB - 1) / XPB;
    if (blocks > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// Persistent, prefetching kernels (fft_persist.hip.h): the streaming path for large batches.  Every workgroup walks the
// batch with a stride of the grid size and keeps the next transform's loads in flight while it computes.
// PersistCfg<L, IO> is the per-(size, policy) configuration, each value chosen by same-box A/B measurement:
//   BLOCK / RL      threads per workgroup, log2 of the points per thread (threads per transform = n >> RL)
//   MINW            waves per SIMD the kernel must fit (register budget); WG_PER_CU = workgroups launched per CU
//   kInvInLds       window samples / irfft table in one LDS copy per workgroup instead of registers
//   kTwLastInLds    the last pass reads its twiddles from an LDS copy of the table (frees 24..30 VGPRs)
template <int L, class IO> struct PersistCfg;
template <class IO> struct PersistCfgBase {
    static constexpr int NBUF = 1, RL = 4;
    static constexpr bool kInvInLds = IO::kInvInLds, kTwLastInLds = false;
};
template <class IO> struct PersistCfg<13, IO> : PersistCfgBase<IO> { static constexpr int BLOCK = 512, MINW = 2, WG_PER_CU = 1; };
template <class IO> struct PersistCfg<12, IO> : PersistCfgBase<IO> { static constexpr int BLOCK = 256, MINW = 2, WG_PER_CU = 2; };
template <class IO> struct PersistCfg<11, IO> : PersistCfgBase<IO> { static constexpr int BLOCK = 256, MINW = 2, WG_PER_CU = 2; };
template <class IO> struct PersistCfg<10, IO> : PersistCfgBase<IO> {
    static constexpr int BLOCK = 256, MINW = IO::kLeanRegisters ? 3 : 2, WG_PER_CU = MINW;
};
// irfft prefetches two row elements per output: the last pass reads its twiddles from LDS to stay inside 256 VGPRs
template <> struct PersistCfg<10, IrfftIO<float>> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 2;
    static constexpr bool kInvInLds = true, kTwLastInLds = true;
};
template <> struct PersistCfg<11, IrfftIO<float>> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 2;
    static constexpr bool kInvInLds = true, kTwLastInLds = true;
};
// STFT n = 2048 / 4096 is compute-limited (more stages per point): with the window in LDS the kernel fits 3 waves/SIMD
// (158 VGPRs), measured +5.5 % / +3.5 % (the memory-limited complex kernel LOSES 4 % with a third workgroup per CU).
template <class IO> struct PersistCfgStftBig {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = 3, WG_PER_CU = 3;
    static constexpr bool kInvInLds = true, kTwLastInLds = false;
};
template <> struct PersistCfg<12, StftIO> : PersistCfgStftBig<StftIO> {};
template <> struct PersistCfg<11, StftIO> : PersistCfgStftBig<StftIO> {};
template <> struct PersistCfg<12, StftMagIO> : PersistCfgStftBig<StftMagIO> {};
template <> struct PersistCfg<11, StftMagIO> : PersistCfgStftBig<StftMagIO> {};
// rfft 8192 (m = 4096): window pairs in registers so that two workgroups (exchange buffer + post-pass table) fit a CU
template <> struct PersistCfg<12, RfftIO<float>> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 2;
    static constexpr bool kInvInLds = false, kTwLastInLds = false;
};
// n = 64: 4 points per thread, 16 threads per transform, three passes of two stages
template <class IO> struct PersistCfg<6, IO> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 2, MINW = 4, WG_PER_CU = 4;
    static constexpr bool kInvInLds = IO::kInvInLds, kTwLastInLds = false;
};
// n = 256, 128: 8 points per thread, 32 / 16 threads per transform -> 2 / 4 transforms per wavefront
template <class IO> struct PersistCfg<8, IO> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 3, MINW = 4, WG_PER_CU = 4;
    static constexpr bool kInvInLds = IO::kInvInLds, kTwLastInLds = false;
};
template <class IO> struct PersistCfg<7, IO> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 3, MINW = 4, WG_PER_CU = 4;
    static constexpr bool kInvInLds = IO::kInvInLds, kTwLastInLds = false;
};
// n = 512: 8 points per thread so that a transform is still one wavefront (three passes of three stages)
template <class IO> struct PersistCfg<9, IO> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 3, MINW = 4, WG_PER_CU = 4;
    static constexpr bool kInvInLds = IO::kInvInLds, kTwLastInLds = false;
};

// Workgroups per CU actually launched.  The rfft kernels (misaligned 8200-byte output rows, an extra LDS round trip)
// stream better with FEWER concurrent rows once the batch no longer fits the 256 MiB Infinity Cache: measured at 4 GiB
// of input, n = 512 / 1024 / 2048: +6 % / +4 % / +4 % with half the grid (config 3: 3.49 -> 3.37 ms; n = 128 / 256: the
// persistent kernel only beats the generic one, by 10 %, with half the grid); at 512 MiB the
// 2048-point kernel loses 10 % with half the grid, the two smaller ones still gain.  STFT and complex want the full grid.
template <int L, class IO> struct PersistGrid {
    static int wg_per_cu(int base, size_t) { return base; }
};
template <> struct PersistGrid<6, RfftIO<float>> { static int wg_per_cu(int base, size_t) { return base / 2; } };
template <> struct PersistGrid<7, RfftIO<float>> { static int wg_per_cu(int base, size_t) { return base / 2; } };
template <> struct PersistGrid<8, RfftIO<float>> { static int wg_per_cu(int base, size_t) { return base / 2; } };
template <> struct PersistGrid<9, RfftIO<float>> { static int wg_per_cu(int base, size_t) { return base / 2; } };
template <> struct PersistGrid<10, RfftIO<float>> {
    static int wg_per_cu(int base, size_t input_bytes) { return input_bytes > (size_t(1) << 30) ? base / 2 : base; }
};

template <typename T, int L, int EPI, class IO>
int launch_persist(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t batch)
{
    using Cfg = PersistCfg<L, IO>;
    constexpr int RL = Cfg::RL;
    constexpr int XPB = Cfg::BLOCK / ((1 << L) >> RL);
    constexpr size_t lds = (size_t)XPB * Cfg::NBUF * lds_elems(1 << L) * sizeof(cpx<T>) +
                           (Cfg::kInvInLds ? (size_t)(1 << L) * sizeof(typename IO::Inv) : 0) +
                           (EPI == EPI_RFFT ? (size_t)(1 << L) * sizeof(cpx<T>) : 0) +
                           (Cfg::kTwLastInLds ? (size_t)(1 << L) / 2 * sizeof(cpx<T>) : 0);
    static_assert(lds * Cfg::WG_PER_CU <= 160 * 1024, "LDS budget");
    auto kern = fft_persist_kernel<T, L, RL, EPI, IO, Cfg>;
    KOFFT_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    size_t blocks = (size_t)ctx->num_cus * PersistGrid<L, IO>::wg_per_cu(Cfg::WG_PER_CU, batch * sizeof(cpx<T>) << L);
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;  // measurement knob
    if (blocks < 1) blocks = 1;
    const size_t need = (batch + XPB - 1) / XPB;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Cfg::BLOCK), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

template <typename T, int N, int EPI, class IO>
int launch_small(kofft_hip_ctx *ctx, const IO &io, size_t batch, const cpx<T> *tw = nullptr)
{
    const size_t blocks = (batch + kSmallBlock - 1) / kSmallBlock;
    if (blocks > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    constexpr size_t lds = small_lds_bytes<T, N>();
    auto kern = fft_small_kernel<T, N, EPI, IO>;
    if (lds > 64 * 1024) {
        KOFFT_HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kSmallBlock), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// adjacent columns / rows per workgroup: 128-byte segments (16 x c32, 8 x c64) when the LDS budget allows
// (measured, c32: 2^15..2^19 0.19 -> 0.225 of the roofline, 2^22..2^24 0.12 -> 0.15)
#ifndef KOFFT_BIG_XPB
#define KOFFT_BIG_XPB(T) (sizeof(T) == 4 ? 16 : 8)
#endif
template <typename T, class IO, int LS>
constexpr int big_block()
{
    const int tpt = (1 << LS) >> rl_for(LS);
    int xpb = KOFFT_BIG_XPB(T);
    while (xpb > 1 && (xpb * tpt > 1024 || (size_t)xpb * (1 << LS) * 8 > 80 * 1024)) xpb /= 2;
    int block = xpb * tpt;
    if (block < 64) block = 64;
    return block;
}

// Run the n-point transform described by `io` (n a power of two >= 1) over `batch` units.
template <typename T, int EPI, class IO>
int dispatch(kofft_hip_ctx *ctx, const IO &io, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    const int L = ilog2(n);
    if (L > max_log2<T>()) return KOFFT_ERR_UNSUPPORTED;
    switch (L) {
    case 0: return launch_small<T, 1, EPI>(ctx, io, batch);
    case 1: return launch_small<T, 2, EPI>(ctx, io, batch);
    case 2: return launch_small<T, 4, EPI>(ctx, io, batch);
    case 3: return launch_small<T, 8, EPI>(ctx, io, batch);
    case 4: return launch_small<T, 16, EPI>(ctx, io, batch);
    default: break;
    }
    const cpx<T> *tw = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::tw, n, &tw);
    if (rc) return rc;
    // n = 32 in f32: still one thread per transform (64 data registers), IO staged through LDS like the small sizes
    if constexpr (sizeof(T) == 4 && !IO::kSlotMinor) if (L == 5) return launch_small<T, 32, EPI>(ctx, io, batch, tw);
    if constexpr (sizeof(T) == 4 && IO::kPersist) if (ctx->use_persist) {
        // streaming sizes: enough transforms to give every resident workgroup several iterations
        if constexpr (EPI == EPI_STORE && IO::kPersistMaxLog2 >= 13) {
            if (L == 13 && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 13, EPI>(ctx, io, tw, batch);
            if (L == 12 && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 12, EPI>(ctx, io, tw, batch);
        }
        if constexpr (EPI == EPI_RFFT) {
            if (L == 12 && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 12, EPI>(ctx, io, tw, batch);
        }
        if (L == 11 && batch >= (size_t)ctx->num_cus * 16) return launch_persist<T, 11, EPI>(ctx, io, tw, batch);
        if (L == 10 && batch >= (size_t)ctx->num_cus * 32) return launch_persist<T, 10, EPI>(ctx, io, tw, batch);
        if (L == 9 && batch >= (size_t)ctx->num_cus * 64) return launch_persist<T, 9, EPI>(ctx, io, tw, batch);
        if (ctx->persist_small && io.group_rows_ok()) {
            if constexpr (IO::kPersistMinLog2 <= 8)
                if (L == 8 && batch >= (size_t)ctx->num_cus * 128) return launch_persist<T, 8, EPI>(ctx, io, tw, batch);
            if constexpr (IO::kPersistMinLog2 <= 7)
                if (L == 7 && batch >= (size_t)ctx->num_cus * 256) return launch_persist<T, 7, EPI>(ctx, io, tw, batch);
            if constexpr (IO::kPersistMinLog2 <= 6)
                if (L == 6 && batch >= (size_t)ctx->num_cus * 512) return launch_persist<T, 6, EPI>(ctx, io, tw, batch);
        }
    }
    switch (L) {
    // lane-over-lines policies (strided axes): as many adjacent lines per workgroup as big_block allows (128-byte segments)
#define KOFFT_CASE(LL) \
    case LL: return launch_wg<T, LL, EPI, IO, (IO::kSlotMinor ? big_block<T, IO, LL>() : 0)>(ctx, io, tw, batch);
        KOFFT_CASE(5)
        KOFFT_CASE(6)
        KOFFT_CASE(7)
        KOFFT_CASE(8)
        KOFFT_CASE(9)
        KOFFT_CASE(10)
        KOFFT_CASE(11)
        KOFFT_CASE(12)
        KOFFT_CASE(13)
    case 14:
        if constexpr (sizeof(T) == 4) return launch_wg<T, 14, EPI>(ctx, io, tw, batch);
        else return KOFFT_ERR_UNSUPPORTED;
#undef KOFFT_CASE
    default: return KOFFT_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------
// large n: two factors (fft_big.hip.h)
// ---------------------------------------------------------------------------------
template <typename T> constexpr int max_log2_big() { return 26; }

// Launch the generic kernel for a sub-transform of log2 size LS with an arbitrary IO policy.
template <typename T, class IO>
int launch_sub(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, int LS, size_t units)
{
    switch (LS) {
    // 16 (c32) / 8 (c64) adjacent columns or rows per workgroup = 128-byte segments while the tile fits the LDS budget
    // (sub-transforms up to 2^9 points; 2^10: 8; larger ones fewer still -- big_block).
#define KOFFT_CASE(LL) \
    case LL: return launch_wg<T, LL, EPI_STORE, IO, big_block<T, IO, LL>()>(ctx, io, tw, units);
        KOFFT_CASE(7)
        KOFFT_CASE(8)
        KOFFT_CASE(9)
        KOFFT_CASE(10)
        KOFFT_CASE(11)
        KOFFT_CASE(12)
        KOFFT_CASE(13)
#undef KOFFT_CASE
    default: return KOFFT_ERR_UNSUPPORTED;
    }
}

// Sub-transforms of the middle factor are at most 2^9 points (three factors cover 2^21 .. 2^26 with 7..9 bits each).
template <typename T>
int launch_mid(kofft_hip_ctx *ctx, const BigMidIO<T> &io, const cpx<T> *tw, int LS, size_t units)
{
    switch (LS) {
    case 7: return launch_wg<T, 7, EPI_STORE, BigMidIO<T>, big_block<T, BigMidIO<T>, 7>()>(ctx, io, tw, units);
    case 8: return launch_wg<T, 8, EPI_STORE, BigMidIO<T>, big_block<T, BigMidIO<T>, 8>()>(ctx, io, tw, units);
    case 9: return launch_wg<T, 9, EPI_STORE, BigMidIO<T>, big_block<T, BigMidIO<T>, 9>()>(ctx, io, tw, units);
    default: return KOFFT_ERR_UNSUPPORTED;
    }
}

template <typename T>
inline int big_rows_per_wg(int LB) { return LB <= 9 ? KOFFT_BIG_XPB(T) : LB == 10 ? 8 : LB == 11 ? 4 : LB == 12 ? 2 : 1; }

template <typename T, bool INVERSE>
int fft_big_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch)
{
    const int L = ilog2(n);
    // Two factors while both stay <= 2^10 points (tiles of 8 adjacent columns / rows, 64..128-byte segments); from 2^22
    // on, three factors of 7..9 bits: one more pass over HBM, but every pass keeps full-width tiles (two factors of
    // 11..13 bits shrink the tiles to 4, 2, 1 columns and fall to 0.05..0.16 of the roofline).
    const bool three = L >= 22 && !ctx->big_two_only;  // measured crossover (2^21: two factors still ahead)
    const int L1 = three ? (L + 2) / 3 : L / 2;
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
    if (ctx->big_tmp_bytes < need) {
        if (ctx->big_tmp) KOFFT_HIP_TRY(ctx, hipFree(ctx->big_tmp));
        ctx->big_tmp = nullptr;
        ctx->big_tmp_bytes = 0;
        KOFFT_HIP_TRY(ctx, hipMalloc(&ctx->big_tmp, need));
        ctx->big_tmp_bytes = need;
    }
    cpx<T> *mid = static_cast<cpx<T> *>(ctx->big_tmp);
    cpx<T> *mid2 = mid + chunk * n;
    const T scale = (T)1 / (T)(float)n;
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = (batch - b0 < chunk) ? batch - b0 : chunk;
        const cpx<T> *src = reinterpret_cast<const cpx<T> *>(d_in) + b0 * n;
        cpx<T> *dst = reinterpret_cast<cpx<T> *>(d_out) + b0 * n;
        // first factor: stages 0 .. L1-1 down the columns of a 2^L1 x 2^(L-L1) matrix
        BigColsIO<T, INVERSE> a{src, mid, L - L1, L - L1, n};
        rc = launch_sub<T>(ctx, a, tw, L1, nb << (L - L1));
        if (rc) return rc;
        const cpx<T> *last_in = mid;
        if (three) {
            BigMidIO<T> m{mid, mid2, L1, L2, L3, L - L2, L - 1 - L1, n};
            rc = launch_mid<T>(ctx, m, tw, L2, nb << (L - L2));
            if (rc) return rc;
            last_in = mid2;
        }
        // last factor: the remaining L3 stages along contiguous rows, prefix K of L - L3 bits, output transposed
        const int LP = L - L3;
        BigRowsIO<T, INVERSE> b{last_in, dst, LP, L3, L - L3, L - 1 - LP, n, scale, big_rows_per_wg<T>(L3) * sizeof(cpx<T>) >= 64};
        rc = launch_sub<T>(ctx, b, tw, L3, nb << LP);
        if (rc) return rc;
    }
    return KOFFT_OK;
}

// ---------------------------------------------------------------------------------
// non-power-of-two lengths: Bluestein (fft.rs:1088-1132)
// ---------------------------------------------------------------------------------
template <typename T>
int fft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch, int inverse);

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
    KOFFT_HIP_TRY(ctx, hipMalloc(&dc, hc.size() * sizeof(T)));
    KOFFT_HIP_TRY(ctx, hipMalloc(&db, hb.size() * sizeof(T)));
    KOFFT_HIP_TRY(ctx, hipMemcpy(dc, hc.data(), hc.size() * sizeof(T), hipMemcpyHostToDevice));
    KOFFT_HIP_TRY(ctx, hipMemcpy(db, hb.data(), hb.size() * sizeof(T), hipMemcpyHostToDevice));
    // b_fft = fft(b) with the ordinary power-of-two path (fft.rs:425-427)
    int rc = fft_dev<T>(ctx, static_cast<T *>(db), static_cast<T *>(db), m, 1, 0);
    if (rc) return rc;
    ctx->tables[std::make_pair(kc, n)] = dc;
    ctx->tables[std::make_pair(kb, n)] = db;
    *chirp = static_cast<const cpx<T> *>(dc);
    *bfft = static_cast<const cpx<T> *>(db);
    return KOFFT_OK;
}

template <typename T, bool INVERSE>
int fft_bluestein_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch)
{
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;  // (2n-1).next_power_of_two()
    const cpx<T> *chirp = nullptr, *bfft = nullptr;
    int rc = get_bluestein<T>(ctx, n, m, &chirp, &bfft);
    if (rc) return rc;
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
            rc = dispatch<T, EPI_STORE>(ctx, io1, m, nb);
            if (rc) return rc;
            BlueSecondIO<T, INVERSE> io2{{}, a, dst, chirp, (int)n, (int)m, scale_m, scale_n};
            rc = dispatch<T, EPI_STORE>(ctx, io2, m, nb);
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

// Large host batches: the transfers dominate (the kernel is ~100 x shorter than its PCIe time), so the batch goes through
// in chunks with the upload of chunk c+1, the kernel of chunk c and the download of chunk c-1 in flight together.
// Uploads and kernels are issued from the calling thread, downloads from a helper thread (a pageable-memory copy blocks
// its caller), each on its own stream; events order them.  Transforms are independent, so chunking cannot change a
// result.  Measured, 8192 x 4096 c32 from pageable memory: 9.6 -> 7.0 ms (55 -> 77 GB/s over PCIe).
//   up(c, stream)   -> hipError_t : enqueue chunk c's host-to-device copy on `stream`
//   run(c)          -> int        : launch chunk c's kernels on ctx->stream (status code)
//   down(c, stream) -> hipError_t : enqueue chunk c's device-to-host copy on `stream`
template <class Up, class Run, class Down>
int pipeline_chunks(kofft_hip_ctx *ctx, size_t nchunks, Up up, Run run, Down down)
{
    hipStream_t s_in = nullptr, s_out = nullptr;
    KOFFT_HIP_TRY(ctx, hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    KOFFT_HIP_TRY(ctx, hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    std::vector<hipEvent_t> uploaded(nchunks), done(nchunks);
    for (size_t c = 0; c < nchunks; ++c) {
        (void)hipEventCreateWithFlags(&uploaded[c], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&done[c], hipEventDisableTiming);
    }
    std::atomic<size_t> launched{0};
    std::atomic<int> failed{0};
    const int device = ctx->device;
    std::thread downloader;
    try {
        downloader = std::thread([&]() {
        (void)hipSetDevice(device);
        for (size_t c = 0; c < nchunks; ++c) {
            while (launched.load(std::memory_order_acquire) <= c && !failed.load()) std::this_thread::yield();
            if (failed.load()) return;
            if (hipEventSynchronize(done[c]) != hipSuccess || down(c, s_out) != hipSuccess ||
                hipStreamSynchronize(s_out) != hipSuccess) {
                failed = 1;
                return;
            }
        }
        });
    } catch (...) {  // no helper thread available: the caller falls back to the serial path
        for (size_t c = 0; c < nchunks; ++c) {
            (void)hipEventDestroy(uploaded[c]);
            (void)hipEventDestroy(done[c]);
        }
        (void)hipStreamDestroy(s_in);
        (void)hipStreamDestroy(s_out);
        return KOFFT_ERR_ALLOC;
    }
    int rc = KOFFT_OK;
    for (size_t c = 0; c < nchunks && rc == KOFFT_OK && !failed.load(); ++c) {
        if (up(c, s_in) != hipSuccess || hipEventRecord(uploaded[c], s_in) != hipSuccess ||
            hipStreamWaitEvent(ctx->stream, uploaded[c], 0) != hipSuccess) {
            rc = KOFFT_ERR_HIP;
            ctx->last_error = "pipelined upload failed";
            break;
        }
        rc = run(c);
        if (rc == KOFFT_OK && hipEventRecord(done[c], ctx->stream) != hipSuccess) rc = KOFFT_ERR_HIP;
        if (rc == KOFFT_OK) launched.store(c + 1, std::memory_order_release);
    }
    if (rc != KOFFT_OK) failed = 1;
    downloader.join();
    (void)hipStreamSynchronize(ctx->stream);
    for (size_t c = 0; c < nchunks; ++c) {
        (void)hipEventDestroy(uploaded[c]);
        (void)hipEventDestroy(done[c]);
    }
    (void)hipStreamDestroy(s_in);
    (void)hipStreamDestroy(s_out);
    if (rc == KOFFT_OK && failed.load()) {
        rc = KOFFT_ERR_HIP;
        ctx->last_error = "pipelined download failed";
    }
    return rc;
}

// does a host batch of `bytes` (both directions together) in `batch` independent rows go through the pipeline?
inline bool use_host_pipeline(const kofft_hip_ctx *ctx, size_t bytes, size_t batch, size_t row_bytes)
{
    return ctx->host_pipeline && bytes >= (size_t(128) << 20) && batch >= 16 && row_bytes <= (size_t(8) << 20);
}
inline size_t host_chunk_rows(const kofft_hip_ctx *ctx, size_t batch)
{
    const size_t parts = (size_t)(ctx->host_chunks > 0 ? ctx->host_chunks : 8);
    return (batch + parts - 1) / parts;
}

template <typename T>
int fft_host(kofft_hip_ctx *ctx, T *data, size_t n, size_t batch, int inverse)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (n > (size_t(1) << (is_pow2(n) ? max_log2_big<T>() : max_log2_big<T>() - 1))) return KOFFT_ERR_UNSUPPORTED;
    if (n == 1) return KOFFT_OK;
    if (!ctx || !data) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = batch * n * 2 * sizeof(T);
    if (ctx->zero_copy && bytes <= kZeroCopyMax && is_pow2(n) && ensure_pinned(ctx, bytes) == KOFFT_OK) {
        std::memcpy(ctx->pinned, data, bytes);
        int zrc = fft_dev<T>(ctx, static_cast<T *>(ctx->pinned_dev), static_cast<T *>(ctx->pinned_dev), n, batch, inverse);
        if (zrc) return zrc;
        KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(data, ctx->pinned, bytes);
        return KOFFT_OK;
    }
    int rc = ensure_stage(ctx, 0, bytes);
    if (rc) return rc;
    T *d = static_cast<T *>(ctx->stage[0]);
    if (use_host_pipeline(ctx, 2 * bytes, batch, n * 2 * sizeof(T))) {
        const size_t chunk = host_chunk_rows(ctx, batch), row = n * 2;
        auto rows = [&](size_t c) { return (batch - c * chunk < chunk) ? batch - c * chunk : chunk; };
        const int prc = pipeline_chunks(
            ctx, (batch + chunk - 1) / chunk,
            [&](size_t c, hipStream_t st) { return hipMemcpyAsync(d + c * chunk * row, data + c * chunk * row, rows(c) * row * sizeof(T), hipMemcpyHostToDevice, st); },
            [&](size_t c) { return fft_dev<T>(ctx, d + c * chunk * row, d + c * chunk * row, n, rows(c), inverse); },
            [&](size_t c, hipStream_t st) { return hipMemcpyAsync(data + c * chunk * row, d + c * chunk * row, rows(c) * row * sizeof(T), hipMemcpyDeviceToHost, st); });
        if (prc != KOFFT_ERR_ALLOC) return prc;  // (no helper thread: serial path below)
    }
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = fft_dev<T>(ctx, d, d, n, batch, inverse);
    if (rc) return rc;
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KOFFT_OK;
}

// fft_strided / ifft_strided (fft.rs:1175-1199, 1236-1260): gather, transform, scatter.
template <typename T>
int fft_strided_host(kofft_hip_ctx *ctx, T *data, size_t data_len, size_t stride, size_t n, int inverse)
{
    if (stride == 0) return KOFFT_ERR_INVALID_STRIDE;  // fft.rs:1181
    if (n == 0) return KOFFT_OK;                       // fft.rs:1185
    if (data_len < (n - 1) * stride + 1) return KOFFT_ERR_MISMATCHED_LENGTHS;  // fft.rs:1188
    if (!ctx || !data) return KOFFT_ERR_NULL;
    std::vector<T> scratch(2 * n);
    for (size_t i = 0; i < n; ++i) {
        scratch[2 * i] = data[2 * i * stride];
        scratch[2 * i + 1] = data[2 * i * stride + 1];
    }
    int rc = fft_host<T>(ctx, scratch.data(), n, 1, inverse);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
        data[2 * i * stride] = scratch[2 * i];
        data[2 * i * stride + 1] = scratch[2 * i + 1];
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
    if (!is_pow2(m) || m > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    if (!is_pow2(m) || m > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_in || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const cpx<T> *rtab = nullptr;
    int rc = get_table<T>(ctx, Kind<T>::rt, m, &rtab);
    if (rc) return rc;
    // threads per transform of the persistent kernels (m/16; m/8 up to m = 512)
    const int tpt = (int)(m <= 64 ? m / 4 : m <= 512 ? m / 8 : m / 16);
    IrfftIO<T> io{{}, reinterpret_cast<const cpx<T> *>(d_in), reinterpret_cast<cpx<T> *>(d_out), rtab, (int)m,
                  (T)1 / (T)(float)m, tpt};
    return dispatch<T, EPI_STORE>(ctx, io, m, batch);
}

template <typename T>
int rfft_host(kofft_hip_ctx *ctx, const T *in, T *out, const T *window, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;
    const size_t m = n / 2;
    if (!is_pow2(m) || m > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !in || !out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = batch * n * sizeof(T), out_bytes = batch * (m + 1) * 2 * sizeof(T);
    if (ctx->zero_copy && in_bytes + out_bytes + n * sizeof(T) <= kZeroCopyMax &&
        ensure_pinned(ctx, in_bytes + out_bytes + n * sizeof(T) + 512) == KOFFT_OK) {
        // [input | window | output] in the pinned, device-mapped buffer (256-byte aligned pieces)
        const size_t o_win = (in_bytes + 255) & ~size_t(255), o_out = (o_win + n * sizeof(T) + 255) & ~size_t(255);
        char *h = static_cast<char *>(ctx->pinned), *dd = static_cast<char *>(ctx->pinned_dev);
        std::memcpy(h, in, in_bytes);
        if (window) std::memcpy(h + o_win, window, n * sizeof(T));
        int zrc = rfft_dev<T>(ctx, reinterpret_cast<const T *>(dd), reinterpret_cast<T *>(dd + o_out),
                              window ? reinterpret_cast<const T *>(dd + o_win) : nullptr, n, batch);
        if (zrc) return zrc;
        KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(out, h + o_out, out_bytes);
        return KOFFT_OK;
    }
    int rc = ensure_stage(ctx, 0, in_bytes);
    if (rc) return rc;
    rc = ensure_stage(ctx, 1, out_bytes);
    if (rc) return rc;
    T *d_win = nullptr;
    if (window) {
        rc = ensure_stage(ctx, 2, n * sizeof(T));
        if (rc) return rc;
        d_win = static_cast<T *>(ctx->stage[2]);
        KOFFT_HIP_TRY(ctx, hipMemcpyAsync(d_win, window, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    }
    if (use_host_pipeline(ctx, in_bytes + out_bytes, batch, n * sizeof(T))) {
        const size_t chunk = host_chunk_rows(ctx, batch), orow = (m + 1) * 2;
        T *d_in = static_cast<T *>(ctx->stage[0]), *d_out = static_cast<T *>(ctx->stage[1]);
        auto rows = [&](size_t c) { return (batch - c * chunk < chunk) ? batch - c * chunk : chunk; };
        const int prc = pipeline_chunks(
            ctx, (batch + chunk - 1) / chunk,
            [&](size_t c, hipStream_t st) { return hipMemcpyAsync(d_in + c * chunk * n, in + c * chunk * n, rows(c) * n * sizeof(T), hipMemcpyHostToDevice, st); },
            [&](size_t c) { return rfft_dev<T>(ctx, d_in + c * chunk * n, d_out + c * chunk * orow, d_win, n, rows(c)); },
            [&](size_t c, hipStream_t st) { return hipMemcpyAsync(out + c * chunk * orow, d_out + c * chunk * orow, rows(c) * orow * sizeof(T), hipMemcpyDeviceToHost, st); });
        if (prc != KOFFT_ERR_ALLOC) return prc;
    }
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = rfft_dev<T>(ctx, static_cast<const T *>(ctx->stage[0]), static_cast<T *>(ctx->stage[1]), d_win, n, batch);
    if (rc) return rc;
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->stage[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KOFFT_OK;
}

template <typename T>
int irfft_host(kofft_hip_ctx *ctx, const T *in, T *out, size_t n, size_t batch)
{
    if (batch == 0) return KOFFT_OK;
    if (n == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (n % 2 != 0) return KOFFT_ERR_INVALID_VALUE;
    const size_t m = n / 2;
    if (!is_pow2(m) || m > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !in || !out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = batch * (m + 1) * 2 * sizeof(T), out_bytes = batch * n * sizeof(T);
    if (ctx->zero_copy && in_bytes + out_bytes <= kZeroCopyMax && ensure_pinned(ctx, in_bytes + out_bytes + 256) == KOFFT_OK) {
        const size_t o_out = (in_bytes + 255) & ~size_t(255);
        char *h = static_cast<char *>(ctx->pinned), *dd = static_cast<char *>(ctx->pinned_dev);
        std::memcpy(h, in, in_bytes);
        int zrc = irfft_dev<T>(ctx, reinterpret_cast<const T *>(dd), reinterpret_cast<T *>(dd + o_out), n, batch);
        if (zrc) return zrc;
        KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(out, h + o_out, out_bytes);
        return KOFFT_OK;
    }
    int rc = ensure_stage(ctx, 0, in_bytes);
    if (rc) return rc;
    rc = ensure_stage(ctx, 1, out_bytes);
    if (rc) return rc;
    if (use_host_pipeline(ctx, in_bytes + out_bytes, batch, n * sizeof(T))) {
        const size_t chunk = host_chunk_rows(ctx, batch), irow = (m + 1) * 2;
        T *d_in = static_cast<T *>(ctx->stage[0]), *d_out = static_cast<T *>(ctx->stage[1]);
        auto rows = [&](size_t c) { return (batch - c * chunk < chunk) ? batch - c * chunk : chunk; };
        const int prc = pipeline_chunks(
            ctx, (batch + chunk - 1) / chunk,
            [&](size_t c, hipStream_t st) { return hipMemcpyAsync(d_in + c * chunk * irow, in + c * chunk * irow, rows(c) * irow * sizeof(T), hipMemcpyHostToDevice, st); },
            [&](size_t c) { return irfft_dev<T>(ctx, d_in + c * chunk * irow, d_out + c * chunk * n, n, rows(c)); },
            [&](size_t c, hipStream_t st) { return hipMemcpyAsync(out + c * chunk * n, d_out + c * chunk * n, rows(c) * n * sizeof(T), hipMemcpyDeviceToHost, st); });
        if (prc != KOFFT_ERR_ALLOC) return prc;
    }
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = irfft_dev<T>(ctx, static_cast<const T *>(ctx->stage[0]), static_cast<T *>(ctx->stage[1]), n, batch);
    if (rc) return rc;
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->stage[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KOFFT_OK;
}

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

// Host-pointer STFT of frames starting at start0, start0+hop, ...: uploads only the samples
// those frames can see.
int stft_host(kofft_hip_ctx *ctx, const float *signal, size_t len, const float *window, size_t win_len,
              size_t start0, size_t hop, float *out, size_t count)
{
    if (count == 0) return KOFFT_OK;
    if (win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (!is_pow2(win_len) || win_len > (size_t(1) << max_log2<float>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || (!signal && len) || !window || !out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t lo = start0 < len ? start0 : len;
    size_t hi = start0 + (count - 1) * hop + win_len;
    if (hi > len) hi = len;
    if (hi < lo) hi = lo;
    const size_t span = hi - lo;
    const size_t out_bytes = count * win_len * 2 * sizeof(float);
    if (ctx->zero_copy && (span + win_len) * sizeof(float) + out_bytes <= kZeroCopyMax &&
        ensure_pinned(ctx, (span + win_len) * sizeof(float) + out_bytes + 768) == KOFFT_OK) {
        // frame() / StftStream / short signals: [samples | window | spectra] in the pinned, device-mapped buffer
        const size_t o_win = (span * sizeof(float) + 255) & ~size_t(255), o_out = (o_win + win_len * sizeof(float) + 255) & ~size_t(255);
        char *h = static_cast<char *>(ctx->pinned), *dd = static_cast<char *>(ctx->pinned_dev);
        if (span) std::memcpy(h, signal + lo, span * sizeof(float));
        std::memcpy(h + o_win, window, win_len * sizeof(float));
        int zrc = stft_dev(ctx, reinterpret_cast<const float *>(dd), span, reinterpret_cast<const float *>(dd + o_win), win_len,
                           start0 - lo, hop, reinterpret_cast<float *>(dd + o_out), count);
        if (zrc) return zrc;
        KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(out, h + o_out, out_bytes);
        return KOFFT_OK;
    }
    int rc = ensure_stage(ctx, 0, (span ? span : 1) * sizeof(float));
    if (rc) return rc;
    rc = ensure_stage(ctx, 1, out_bytes);
    if (rc) return rc;
    rc = ensure_stage(ctx, 2, win_len * sizeof(float));
    if (rc) return rc;
    if (span)
        KOFFT_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], signal + lo, span * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[2], window, win_len * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    // positions are relative to `lo`; a start past the end of the signal leaves every sample zero
    rc = stft_dev(ctx, static_cast<const float *>(ctx->stage[0]), span, static_cast<const float *>(ctx->stage[2]),
                  win_len, start0 - lo, hop, static_cast<float *>(ctx->stage[1]), count);
    if (rc) return rc;
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->stage[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KOFFT_OK;
}

// stft::istft (stft.rs:117-156, mode 1), stft::inverse_parallel (stft.rs:289-343, mode 2), stft::inverse_frame
// (stft.rs:384-399, mode 0): ifft every frame in place, then the ordered overlap-add kernel.
int istft_dev(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window, size_t win_len, size_t hop,
              float *d_output, size_t out_len, float *d_scratch, size_t scratch_len, int mode = 1, size_t start0 = 0)
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

// host-pointer wrapper shared by istft / inverse_parallel / inverse_frame
int istft_host(kofft_hip_ctx *ctx, float *frames_data, size_t frames, const float *window, size_t win_len, size_t hop,
               float *output, size_t out_len, float *scratch, size_t scratch_len, int mode, size_t start0, bool copy_frames_back)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;
    if (mode == 1 && scratch_len != out_len) return KOFFT_ERR_MISMATCHED_LENGTHS;
    if (frames > 0 && win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (frames > 0 && (!is_pow2(win_len) || win_len > (size_t(1) << max_log2<float>()))) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || (frames && (!frames_data || !window)) || (out_len && (!output || (mode == 1 && !scratch)))) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t fr_bytes = frames * win_len * 2 * sizeof(float);
    const size_t o_bytes = out_len * sizeof(float);
    // one staging allocation: [frames | output | scratch | window]
    const size_t a0 = 0, a1 = (fr_bytes + 255) & ~size_t(255), a2 = a1 + ((o_bytes + 255) & ~size_t(255)),
                 a3 = a2 + ((o_bytes + 255) & ~size_t(255)), total = a3 + win_len * sizeof(float) + 256;
    if (ctx->zero_copy && total <= kZeroCopyMax && ensure_pinned(ctx, total) == KOFFT_OK) {
        // one frame (IstftStream, inverse_frame) or a short batch: the kernels work on the pinned, device-mapped buffer
        char *h = static_cast<char *>(ctx->pinned), *dd = static_cast<char *>(ctx->pinned_dev);
        if (fr_bytes) std::memcpy(h + a0, frames_data, fr_bytes);
        if (o_bytes) std::memcpy(h + a1, output, o_bytes);
        if (win_len) std::memcpy(h + a3, window, win_len * sizeof(float));
        int zrc = istft_dev(ctx, reinterpret_cast<float *>(dd + a0), frames, reinterpret_cast<const float *>(dd + a3), win_len, hop,
                            reinterpret_cast<float *>(dd + a1), out_len, reinterpret_cast<float *>(dd + a2),
                            mode == 1 ? scratch_len : out_len, mode, start0);
        if (zrc) return zrc;
        KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (fr_bytes && copy_frames_back) std::memcpy(frames_data, h + a0, fr_bytes);
        if (o_bytes) {
            std::memcpy(output, h + a1, o_bytes);
            if (mode == 1 && scratch) std::memcpy(scratch, h + a2, o_bytes);
        }
        return KOFFT_OK;
    }
    int rc = ensure_stage(ctx, 0, total);
    if (rc) return rc;
    char *base = static_cast<char *>(ctx->stage[0]);
    if (fr_bytes) KOFFT_HIP_TRY(ctx, hipMemcpyAsync(base + a0, frames_data, fr_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (o_bytes) KOFFT_HIP_TRY(ctx, hipMemcpyAsync(base + a1, output, o_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (win_len) KOFFT_HIP_TRY(ctx, hipMemcpyAsync(base + a3, window, win_len * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = istft_dev(ctx, reinterpret_cast<float *>(base + a0), frames, reinterpret_cast<const float *>(base + a3), win_len, hop,
                   reinterpret_cast<float *>(base + a1), out_len, reinterpret_cast<float *>(base + a2), mode == 1 ? scratch_len : out_len,
                   mode, start0);
    if (rc) return rc;
    if (fr_bytes && copy_frames_back)
        KOFFT_HIP_TRY(ctx, hipMemcpyAsync(frames_data, base + a0, fr_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (o_bytes) {
        KOFFT_HIP_TRY(ctx, hipMemcpyAsync(output, base + a1, o_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (mode == 1 && scratch) KOFFT_HIP_TRY(ctx, hipMemcpyAsync(scratch, base + a2, o_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
    StftMagIO io{{{}, d_samples, d_win, nullptr, len, hop, 0, (int)win_len}, d_mags};
    int rc = dispatch<float, EPI_STORE>(ctx, io, win_len, frames);
    if (rc) return rc;
    const size_t count = frames * (win_len / 2);
    if (count > 0) {
        size_t blocks = (count + 255) / 256;
        if (blocks > (size_t)ctx->num_cus * 8) blocks = (size_t)ctx->num_cus * 8;
        hipLaunchKernelGGL(max_nonneg_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_mags, count,
                           reinterpret_cast<unsigned *>(d_max));
        KOFFT_HIP_TRY(ctx, hipGetLastError());
    }
    return KOFFT_OK;
}

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
    if (ctx->nd_transpose && len >= (size_t)ctx->nd_transpose_min && stride == inner && lines * len * sizeof(cpx<T>) >= (size_t(16) << 20)) {
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
        if (ctx->big_tmp_bytes < need) {
            if (ctx->big_tmp) KOFFT_HIP_TRY(ctx, hipFree(ctx->big_tmp));
            ctx->big_tmp = nullptr;
            ctx->big_tmp_bytes = 0;
            KOFFT_HIP_TRY(ctx, hipMalloc(&ctx->big_tmp, need));
            ctx->big_tmp_bytes = need;
        }
        // NOTE: fft_dev on n <= 16384 never touches big_tmp (only the two-factor path does), so the panel is safe there
        cpx<T> *panel = static_cast<cpx<T> *>(ctx->big_tmp);
        cpx<T> *data = reinterpret_cast<cpx<T> *>(d_data);
        for (size_t o0 = 0; o0 < outer; o0 += OG) {
            const size_t og = (outer - o0 < OG) ? outer - o0 : OG;
            for (size_t p0 = 0; p0 < inner; p0 += P) {
                const size_t pw = (inner - p0 < P) ? inner - p0 : P;
                cpx<T> *blk = data + o0 * outer_stride + p0;
                dim3 g1((unsigned)((pw + 31) / 32), (unsigned)((len + 31) / 32), (unsigned)og);
                hipLaunchKernelGGL(transpose_kernel<T>, g1, dim3(256), 0, ctx->stream, blk, panel, len, pw, inner, len, outer_stride,
                                   pw * len);
                KOFFT_HIP_TRY(ctx, hipGetLastError());
                int rc = fft_dev<T>(ctx, reinterpret_cast<T *>(panel), reinterpret_cast<T *>(panel), len, og * pw, inverse);
                if (rc) return rc;
                dim3 g2((unsigned)((len + 31) / 32), (unsigned)((pw + 31) / 32), (unsigned)og);
                hipLaunchKernelGGL(transpose_kernel<T>, g2, dim3(256), 0, ctx->stream, panel, blk, pw, len, len, inner, pw * len,
                                   outer_stride);
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
        if (!is_pow2(n) || n > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
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
    rc = fft_dev<T>(ctx, d_data, d_data, cols, rows, inverse);
    if (rc) return rc;
    return fft_axis_dev<T>(ctx, d_data, rows, cols, cols, 0, cols, inverse);
}

template <typename T>
int fft_nd_host(kofft_hip_ctx *ctx, T *data, size_t depth, size_t rows, size_t cols, int inverse)
{
    if (depth == 0 || rows == 0 || cols == 0) return KOFFT_OK;
    for (size_t n : {depth, rows, cols})
        if (!is_pow2(n) || n > (size_t(1) << max_log2<T>())) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !data) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = depth * rows * cols * 2 * sizeof(T);
    int rc = ensure_stage(ctx, 0, bytes);
    if (rc) return rc;
    T *d = static_cast<T *>(ctx->stage[0]);
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = fft_nd_dev<T>(ctx, d, depth, rows, cols, inverse);
    if (rc) return rc;
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KOFFT_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------------
extern "C" {

const char *kofft_hip_strerror(int status)
{
    switch (status) {
    case KOFFT_OK: return "Ok";
    case KOFFT_ERR_EMPTY_INPUT: return "FftError::EmptyInput";
    case KOFFT_ERR_NON_POWER_OF_TWO_NO_STD: return "FftError::NonPowerOfTwoNoStd";
    case KOFFT_ERR_MISMATCHED_LENGTHS: return "FftError::MismatchedLengths";
    case KOFFT_ERR_INVALID_STRIDE: return "FftError::InvalidStride";
    case KOFFT_ERR_INVALID_HOP_SIZE: return "FftError::InvalidHopSize";
    case KOFFT_ERR_INVALID_VALUE: return "FftError::InvalidValue";
    case KOFFT_ERR_HIP: return "HIP runtime error (see kofft_hip_last_error)";
    case KOFFT_ERR_UNSUPPORTED: return "length not supported by the device path";
    case KOFFT_ERR_NULL: return "null context or pointer";
    case KOFFT_ERR_ALLOC: return "allocation failed";
    default: return "unknown status";
    }
}

const char *kofft_hip_last_error(const kofft_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

const char *kofft_hip_version(void) { return "kofft-hip 0.1.0 (gfx950)"; }

int kofft_hip_device_count(int *count)
{
    if (!count) return KOFFT_ERR_NULL;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *count = (e == hipSuccess) ? c : 0;
    return e == hipSuccess ? KOFFT_OK : KOFFT_ERR_HIP;
}

int kofft_hip_create(int device, kofft_hip_ctx **out)
{
    if (!out) return KOFFT_ERR_NULL;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return KOFFT_ERR_HIP;
    kofft_hip_ctx *ctx = new (std::nothrow) kofft_hip_ctx();
    if (!ctx) return KOFFT_ERR_ALLOC;
    ctx->device = device;
    if (const char *e = getenv("KOFFT_HIP_NO_PERSIST")) ctx->use_persist = !(e[0] == '1');
    if (const char *e = getenv("KOFFT_HIP_PERSIST_GRID_PCT")) ctx->persist_grid_pct = atoi(e);
    if (const char *e = getenv("KOFFT_HIP_BIG_TWO_FACTORS")) ctx->big_two_only = (e[0] == '1');
    if (const char *e = getenv("KOFFT_HIP_PERSIST_SMALL")) ctx->persist_small = !(e[0] == '0');
    if (const char *e = getenv("KOFFT_HIP_HOST_PIPELINE")) ctx->host_pipeline = !(e[0] == '0');
    if (const char *e = getenv("KOFFT_HIP_ZERO_COPY")) ctx->zero_copy = !(e[0] == '0');
    if (const char *e = getenv("KOFFT_HIP_ND_TRANSPOSE")) ctx->nd_transpose = !(e[0] == '0');
    if (const char *e = getenv("KOFFT_HIP_BLUESTEIN_FUSED")) ctx->blue_fused = !(e[0] == '0');
    if (const char *e = getenv("KOFFT_HIP_ND_TRANSPOSE_MIN")) ctx->nd_transpose_min = atoi(e);
    if (const char *e = getenv("KOFFT_HIP_HOST_CHUNKS")) ctx->host_chunks = atoi(e);
    if (const char *e = getenv("KOFFT_HIP_BIG_CHUNK_MB")) {
        const long mb = atol(e);
        if (mb > 0) ctx->big_chunk_bytes = (size_t)mb << 20;
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->num_cus = prop.multiProcessorCount;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return KOFFT_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return KOFFT_OK;
}

int kofft_hip_destroy(kofft_hip_ctx *ctx)
{
    if (!ctx) return KOFFT_ERR_NULL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->tables) (void)hipFree(kv.second);
    for (int i = 0; i < 3; ++i)
        if (ctx->stage[i]) (void)hipFree(ctx->stage[i]);
    if (ctx->big_tmp) (void)hipFree(ctx->big_tmp);
    if (ctx->blue_tmp) (void)hipFree(ctx->blue_tmp);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return KOFFT_OK;
}

int kofft_hip_set_stream(kofft_hip_ctx *ctx, void *hip_stream)
{
    if (!ctx) return KOFFT_ERR_NULL;
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return KOFFT_OK;
}

int kofft_hip_synchronize(kofft_hip_ctx *ctx)
{
    if (!ctx) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KOFFT_OK;
}

int kofft_hip_twiddles_f32(size_t n, float *out)
{
    if (!out && n >= 2) return KOFFT_ERR_NULL;
    kofft_tables::twiddles_f32(n, out);
    return KOFFT_OK;
}
int kofft_hip_twiddles_f64(size_t n, double *out)
{
    if (!out && n >= 2) return KOFFT_ERR_NULL;
    kofft_tables::twiddles_f64(n, out);
    return KOFFT_OK;
}
int kofft_hip_rfft_table_f32(size_t m, float *out)
{
    if (!out && m) return KOFFT_ERR_NULL;
    kofft_tables::rfft_table_f32(m, out);
    return KOFFT_OK;
}
int kofft_hip_rfft_table_f64(size_t m, double *out)
{
    if (!out && m) return KOFFT_ERR_NULL;
    kofft_tables::rfft_table_f64(m, out);
    return KOFFT_OK;
}
int kofft_hip_hann_f32(size_t len, float *out)
{
    if (!out && len) return KOFFT_ERR_NULL;
    kofft_tables::hann_f32(len, out);
    return KOFFT_OK;
}

int kofft_hip_fft_c32(kofft_hip_ctx *ctx, float *data, size_t n, size_t batch, int inverse)
{
    return fft_host<float>(ctx, data, n, batch, inverse);
}
int kofft_hip_fft_c64(kofft_hip_ctx *ctx, double *data, size_t n, size_t batch, int inverse)
{
    return fft_host<double>(ctx, data, n, batch, inverse);
}
int kofft_hip_fft_c32_dev(kofft_hip_ctx *ctx, float *d_data, size_t n, size_t batch, int inverse)
{
    return fft_dev<float>(ctx, d_data, d_data, n, batch, inverse);
}
int kofft_hip_fft_c64_dev(kofft_hip_ctx *ctx, double *d_data, size_t n, size_t batch, int inverse)
{
    return fft_dev<double>(ctx, d_data, d_data, n, batch, inverse);
}
int kofft_hip_fft_c32_dev_oop(kofft_hip_ctx *ctx, const float *d_in, float *d_out, size_t n, size_t batch,
                              int inverse)
{
    return fft_dev<float>(ctx, d_in, d_out, n, batch, inverse);
}
int kofft_hip_fft_c64_dev_oop(kofft_hip_ctx *ctx, const double *d_in, double *d_out, size_t n, size_t batch,
                              int inverse)
{
    return fft_dev<double>(ctx, d_in, d_out, n, batch, inverse);
}
int kofft_hip_fft_c32_strided(kofft_hip_ctx *ctx, float *data, size_t data_len, size_t stride, size_t n,
                              int inverse)
{
    return fft_strided_host<float>(ctx, data, data_len, stride, n, inverse);
}
int kofft_hip_fft_c64_strided(kofft_hip_ctx *ctx, double *data, size_t data_len, size_t stride, size_t n,
                              int inverse)
{
    return fft_strided_host<double>(ctx, data, data_len, stride, n, inverse);
}

int kofft_hip_rfft_f32(kofft_hip_ctx *ctx, const float *in, float *out, const float *window, size_t n,
                       size_t batch)
{
    return rfft_host<float>(ctx, in, out, window, n, batch);
}
int kofft_hip_rfft_f32_dev(kofft_hip_ctx *ctx, const float *d_in, float *d_out, const float *d_window, size_t n,
                           size_t batch)
{
    return rfft_dev<float>(ctx, d_in, d_out, d_window, n, batch);
}
int kofft_hip_irfft_f32(kofft_hip_ctx *ctx, const float *in, float *out, size_t n, size_t batch)
{
    return irfft_host<float>(ctx, in, out, n, batch);
}
int kofft_hip_irfft_f32_dev(kofft_hip_ctx *ctx, const float *d_in, float *d_out, size_t n, size_t batch)
{
    return irfft_dev<float>(ctx, d_in, d_out, n, batch);
}
int kofft_hip_rfft_f64(kofft_hip_ctx *ctx, const double *in, double *out, const double *window, size_t n,
                       size_t batch)
{
    return rfft_host<double>(ctx, in, out, window, n, batch);
}
int kofft_hip_rfft_f64_dev(kofft_hip_ctx *ctx, const double *d_in, double *d_out, const double *d_window,
                           size_t n, size_t batch)
{
    return rfft_dev<double>(ctx, d_in, d_out, d_window, n, batch);
}
int kofft_hip_irfft_f64(kofft_hip_ctx *ctx, const double *in, double *out, size_t n, size_t batch)
{
    return irfft_host<double>(ctx, in, out, n, batch);
}
int kofft_hip_irfft_f64_dev(kofft_hip_ctx *ctx, const double *d_in, double *d_out, size_t n, size_t batch)
{
    return irfft_dev<double>(ctx, d_in, d_out, n, batch);
}

int kofft_hip_stft_f32(kofft_hip_ctx *ctx, const float *signal, size_t len, const float *window, size_t win_len,
                       size_t hop, float *out, size_t frames)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;               // stft.rs:83
    const size_t required = (len + hop - 1) / hop;                  // stft.rs:86
    if (frames < required) return KOFFT_ERR_MISMATCHED_LENGTHS;    // stft.rs:87
    return stft_host(ctx, signal, len, window, win_len, 0, hop, out, frames);
}

int kofft_hip_stft_parallel_f32(kofft_hip_ctx *ctx, const float *signal, size_t len, const float *window,
                                size_t win_len, size_t hop, float *out, size_t frames)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;  // stft.rs:242 -- the only check parallel() makes
    return stft_host(ctx, signal, len, window, win_len, 0, hop, out, frames);
}

int kofft_hip_stft_frame_f32(kofft_hip_ctx *ctx, const float *signal, size_t len, const float *window,
                             size_t win_len, size_t start, float *frame_out)
{
    return stft_host(ctx, signal, len, window, win_len, start, 1, frame_out, 1);
}

int kofft_hip_stft_f32_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len, const float *d_window,
                           size_t win_len, size_t hop, float *d_out, size_t first_frame, size_t count)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;  // stft.rs:83 / 242
    return stft_dev(ctx, d_signal, len, d_window, win_len, first_frame * hop, hop, d_out, count);
}

int kofft_hip_istft_f32_dev(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window, size_t win_len,
                            size_t hop, float *d_output, size_t out_len, float *d_scratch, size_t scratch_len)
{
    return istft_dev(ctx, d_frames, frames, d_window, win_len, hop, d_output, out_len, d_scratch, scratch_len);
}

int kofft_hip_istft_f32(kofft_hip_ctx *ctx, float *frames_data, size_t frames, const float *window, size_t win_len,
                        size_t hop, float *output, size_t out_len, float *scratch, size_t scratch_len)
{
    return istft_host(ctx, frames_data, frames, window, win_len, hop, output, out_len, scratch, scratch_len, 1, 0, true);
}

int kofft_hip_istft_parallel_f32(kofft_hip_ctx *ctx, const float *frames_data, size_t frames, const float *window,
                                 size_t win_len, size_t hop, float *output, size_t out_len)
{
    // inverse_parallel clones each frame (stft.rs:310): the caller's frames are left untouched
    return istft_host(ctx, const_cast<float *>(frames_data), frames, window, win_len, hop, output, out_len, nullptr, out_len, 2, 0,
                      false);
}

int kofft_hip_istft_frame_f32(kofft_hip_ctx *ctx, float *frame, const float *window, size_t win_len, size_t start,
                              float *output, size_t out_len)
{
    // inverse_frame (stft.rs:384-399): ifft(frame) in place, output[start + i] += frame[i].re * window[i], no normalisation
    return istft_host(ctx, frame, 1, window, win_len, win_len ? win_len : 1, output, out_len, nullptr, out_len, 0, start, true);
}

int kofft_hip_stft_magnitudes_f32_dev(kofft_hip_ctx *ctx, const float *d_samples, size_t len, size_t win_len, size_t hop,
                                      float *d_mags, size_t frames, float *d_max)
{
    return stft_mag_dev(ctx, d_samples, len, win_len, hop, d_mags, frames, d_max);
}

int kofft_hip_stft_magnitudes_f32(kofft_hip_ctx *ctx, const float *samples, size_t len, size_t win_len, size_t hop,
                                  float *mags, size_t frames, float *max_mag)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;
    if (frames < (len + hop - 1) / hop) return KOFFT_ERR_MISMATCHED_LENGTHS;
    if (frames > 0 && win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (frames > 0 && (!is_pow2(win_len) || win_len > (size_t(1) << max_log2<float>()))) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !max_mag || (frames && (!mags || (!samples && len)))) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t m_bytes = frames * (win_len / 2) * sizeof(float);
    int rc = ensure_stage(ctx, 0, (len ? len : 1) * sizeof(float));
    if (rc) return rc;
    rc = ensure_stage(ctx, 1, m_bytes + 256);
    if (rc) return rc;
    rc = ensure_stage(ctx, 2, 256);
    if (rc) return rc;
    if (len) KOFFT_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], samples, len * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = stft_mag_dev(ctx, static_cast<const float *>(ctx->stage[0]), len, win_len, hop, static_cast<float *>(ctx->stage[1]), frames,
                      static_cast<float *>(ctx->stage[2]));
    if (rc) return rc;
    if (m_bytes) KOFFT_HIP_TRY(ctx, hipMemcpyAsync(mags, ctx->stage[1], m_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipMemcpyAsync(max_mag, ctx->stage[2], sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    KOFFT_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KOFFT_OK;
}

int kofft_hip_fftnd_c32(kofft_hip_ctx *ctx, float *data, size_t depth, size_t rows, size_t cols, int inverse)
{
    return fft_nd_host<float>(ctx, data, depth, rows, cols, inverse);
}
int kofft_hip_fftnd_c64(kofft_hip_ctx *ctx, double *data, size_t depth, size_t rows, size_t cols, int inverse)
{
    return fft_nd_host<double>(ctx, data, depth, rows, cols, inverse);
}
int kofft_hip_fftnd_c32_dev(kofft_hip_ctx *ctx, float *d_data, size_t depth, size_t rows, size_t cols, int inverse)
{
    return fft_nd_dev<float>(ctx, d_data, depth, rows, cols, inverse);
}
int kofft_hip_fftnd_c64_dev(kofft_hip_ctx *ctx, double *d_data, size_t depth, size_t rows, size_t cols, int inverse)
{
    return fft_nd_dev<double>(ctx, d_data, depth, rows, cols, inverse);
}

}  // extern "C"
