// host_common.hip.h -- what every translation unit of libkofft_hip.so shares: the context, the planner-table cache,
// launch geometry and the size dispatch.  The kernels are instantiated per family in their own translation units
// (k_complex_*.hip, k_real_*.hip, k_stft.hip, k_nd.hip) so that the library builds in parallel; kofft_hip.hip holds the
// host-pointer wrappers and the extern "C" ABI of include/kofft_hip.h.  gfx950 only; compiled with -ffp-contract=off.
#pragma once

#include "../../include/kofft_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "fft_big.hip.h"
#include "fft_persist.hip.h"
#include "fft_istft.hip.h"
#include "fft_regfile.hip.h"
#include "fft_split_wide.hip.h"
#include "fft_wg.hip.h"
#include "tables.h"

// ---------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------
#define KOFFT_BIG_PROBE_MAX 8  /* candidates of the intermediate's placement probe (big_probe_pick) */
struct kofft_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t order_event = nullptr;  // kofft_hip_set_stream: orders the new stream after the old one
    int num_cus = 256;
    bool use_persist = true;  // KOFFT_HIP_NO_PERSIST=1 forces the generic kernels (A/B measurements, tests)
    bool istft_fused = true;  // KOFFT_HIP_ISTFT_FUSED=0: ISTFT always as inverse transforms + the overlap-add kernel
    bool blue_persist = true; // KOFFT_HIP_BLUESTEIN_PERSIST=0: the one-launch Bluestein arm always as one workgroup per XPB transforms
    int persist_grid_pct = 0; // KOFFT_HIP_PERSIST_GRID_PCT: scale the persistent grids (measurements only)
    bool big_two_only = false; // (a member only -- no environment variable since round 4) never split into three factors (A/B measurements)
    int big_three_min = 22;    // KOFFT_HIP_BIG_THREE_MIN: smallest log2 n split into three factors
    bool small32 = true;       // KOFFT_HIP_SMALL32=0: f32 n = 32 on the thread-group kernel instead of one thread per transform (A/B)
    bool big_first11 = true;   // KOFFT_HIP_BIG_FIRST11=0: 2^21 as 2^10 x 2^11 with the one-tile-per-workgroup kernel for the 2^11 factor (A/B)
    bool big_mid_group = true; // (a member only -- no environment variable since round 4) the middle factor reloads its table entries for every tile (A/B)
    int big_first_larger = -1; // (a member only -- no environment variable since round 4) odd log2 n: which factor takes the extra bit (default: the first in f32)
    bool big_narrow = true;    // KOFFT_HIP_BIG_NARROW=0: full-width tiles even when they leave CUs without a workgroup (A/B)
    int big_narrow_per_cu = 1; // (a member only -- no environment variable since round 4) workgroups per CU the narrowing aims for
    bool big_persist = true;   // KOFFT_HIP_BIG_PERSIST=0: factors on the one-tile-per-workgroup kernel (A/B measurements)
    size_t big_persist_min_units = 32;  // (a member only -- no environment variable since round 4) units (columns / rows) per CU from which the persistent factor kernels run
    int big_first_persist = -1;  // (a member only -- no environment variable since round 4) first factor one tile per workgroup / persistent (default: big_persist)
    int big_mid_nt = -1;       // (compile-time only since round 4; no getenv) force plain / streaming loads of the intermediate (default: by chunk size)
    bool blue_fused = true;    // KOFFT_HIP_BLUESTEIN_FUSED=0: pointwise steps as separate kernels at every size
    bool blue_one_kernel = true;  // KOFFT_HIP_BLUESTEIN_ONE=0: two launches through a scratch even where one workgroup holds m points
    bool nd_transpose = true;  // KOFFT_HIP_ND_TRANSPOSE=0: long strided axes through the strided kernel
    bool nd_fused = true;      // KOFFT_HIP_ND_FUSED=0: 2-D c32 images with 1024 .. 4096-point rows through rows + two column-tile passes instead of the fused two passes (A/B)
    bool nd_two_pass = true;   // KOFFT_HIP_ND_TWO_PASS=0: power-of-two axes of 4096 .. 16384 points through the transposes instead of two column-tile passes (A/B)
    int nd_transpose_min = 4096;  // (a member only -- no environment variable since round 4) shortest axis that takes the transpose route (measured: 1024 loses, 2048 ties)
    bool zero_copy = true;     // KOFFT_HIP_ZERO_COPY=0: small host calls through staged copies like large ones
    bool host_pipeline = true; // KOFFT_HIP_HOST_PIPELINE=0: host-pointer batches in one upload / kernel / download
    int host_chunks = 0;       // (a member only -- no environment variable since round 4) pieces of a pipelined host batch (default 8)
    bool rfft14_wide = true;    // KOFFT_HIP_RFFT14_WIDE=0: rfft / irfft of 32768 reals on the generic kernel
    bool rfft13_persist = true; // KOFFT_HIP_RFFT13_PERSIST=0: rfft / irfft n = 16384 on the generic kernel
    int persist64 = 1;         // KOFFT_HIP_PERSIST64=0: c64 n = 4096 / 8192 on the generic kernel (A/B measurements)
    bool persist_small = true; // KOFFT_HIP_PERSIST_SMALL=0: n = 128, 256 on the generic kernels (A/B measurements)
                               // double-buffered one (measured, same box: c32 0.52-0.53 against 0.61-0.63, STFT 0.41 against 0.43)
    bool rfft_regfile_epi = true;  // KOFFT_HIP_RFFT_REGFILE_EPI=0: rfft of 65536 (f32) / 32768 (f64) reals as register-file transform + post-pass kernel (two passes) instead of one (A/B)
    bool use_regfile = true;   // KOFFT_HIP_REGFILE=0: c32 2^15 / c64 2^14 on the two-factor path instead of the register-file-resident kernel (A/B)
    bool use_split = true;     // KOFFT_HIP_SPLIT=0: n = 8192 on the block-synchronised persistent kernel instead of the wave-split one (A/B)
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
    bool big_row_pairs = true;       // KOFFT_HIP_BIG_ROW_PAIRS=0: c32 last factor one row per thread slot (8-row tiles) instead of row pairs (A/B)
    bool big_blocked = true;         // KOFFT_HIP_BIG_BLOCKED=0: natural layout of the two-factor intermediate (A/B)
    bool big_tmp_external = false;  // KOFFT_EXP_API builds only: the intermediate belongs to the experiment script
    // placement probe of big_tmp (round 6, complex_impl.hip.h: big_probe_pick): candidates timed, figures of the last probe
    int big_probe = 5;               // KOFFT_HIP_BIG_PROBE: candidate allocations per probe (0 / 1: take what hipMalloc hands out)
    int big_probe_n = 0, big_probe_pick = -1;
    float big_probe_first_us[KOFFT_BIG_PROBE_MAX] = {}, big_probe_total_us[KOFFT_BIG_PROBE_MAX] = {};
    // small host-pointer calls (one frame, one transform): a pinned, device-mapped buffer the kernels read and write
    // directly over PCIe -- one launch and one synchronisation instead of two staged copies around them
    void *pinned = nullptr;      // host address
    void *pinned_dev = nullptr;  // the same memory as the device sees it
    size_t pinned_bytes = 0;
    void *blue_tmp = nullptr;  // zero-padded work buffer of the Bluestein arm
    size_t blue_tmp_bytes = 0;
    void *real_tmp = nullptr;  // the inner complex transform of real / STFT lengths the fused kernels do not cover
    size_t real_tmp_bytes = 0;
    size_t big_chunk_bytes = size_t(512) << 20;  // KOFFT_HIP_BIG_CHUNK_MB; measured on config 5 with the persistent factor kernels: 128 MiB 14.7 ms, 256 13.0, 512 12.1, 1024 12.5, 2048 13.1
};

namespace kofft {
namespace host {

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

inline int ensure_pinned(kofft_hip_ctx *ctx, size_t bytes)
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

inline int ensure_stage(kofft_hip_ctx *ctx, int which, size_t bytes)
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

// hipFuncSetAttribute once per (kernel instance, device): `done` is a function-local static of the caller's template
// instance, one bit per device ordinal.
inline int set_dyn_lds_once(kofft_hip_ctx *ctx, std::atomic<unsigned long long> &done, const void *kern, size_t lds)
{
    const unsigned long long bit = 1ull << (ctx->device & 63);
    if (ctx->device < 64 && (done.load(std::memory_order_relaxed) & bit)) return KOFFT_OK;
    KOFFT_HIP_TRY(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (ctx->device < 64) done.fetch_or(bit, std::memory_order_relaxed);
    return KOFFT_OK;
}

// Points per thread of the one-workgroup-per-transform kernel at n = 8192 / 16384: 32 (half the threads, one exchange less)
// or 16.  Measured per kind on one box (tools/sweep.py, KOFFT_RL_BIG=4 against 5): 16 wins for every f64 kernel of 8192
// points (c64 +7 %, rfft64 n = 16384 +18 %, irfft64 +8 %) and for the f32 real-input kernels (rfft32 n = 16384 +22 %,
// 32768 +4 %); 32 wins for c32 16384 (+4 %) and STFT 16384 (+14 %); irfft32 does not care.
template <typename T, int EPI>
constexpr int rl_for_kind(int L)
{
    return (L >= 13 && (sizeof(T) == 8 || EPI == EPI_RFFT)) ? 4 : rl_for(L);
}

#ifndef KOFFT_WG_MIN_BLOCK
#define KOFFT_WG_MIN_BLOCK 256
#endif
#ifndef KOFFT_WG64_SMALL_FROM
#define KOFFT_WG64_SMALL_FROM 9
#endif
template <typename T, int L, int EPI, class IO, int BLOCK_OVERRIDE = 0>
int launch_wg(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t batch)
{
    // (sub-transforms of the large-n path come with their own block size, computed for rl_for)
    constexpr int RL = BLOCK_OVERRIDE ? rl_for(L) : rl_for_kind<T, EPI>(L);
    constexpr int TPT0 = (1 << L) >> RL;
    // f64 from n = 512: one transform per workgroup down to a single wavefront (round 3, same box, 256 -> 64 / 128 threads: rfft64
    // n = 2048 0.62 -> 0.71, c64 2048 0.72 -> 0.75, c64 / rfft64 512 .. 1024 +2..4 %, irfft64 2048 0.63 -> 0.64..0.66; below
    // 512 and for f32 the results are mixed (+-4 %) and the workgroups stay at 256)
    constexpr int MINB = (sizeof(T) == 8 && L >= KOFFT_WG64_SMALL_FROM) ? 64 : KOFFT_WG_MIN_BLOCK;
    constexpr int BLOCK = BLOCK_OVERRIDE ? BLOCK_OVERRIDE : (TPT0 > MINB ? TPT0 : MINB);
    constexpr int TPT = (1 << L) >> RL;
    constexpr int XPB = BLOCK / TPT;
    constexpr size_t lds = lds_wg_bytes<T, wg_split_lds<T, L, EPI, IO>(), IO::kSlotMinor, XPB>(1 << L);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fft_wg_kernel<T, L, RL, BLOCK, EPI, IO>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
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
// c64 n = 8192 (round 3; n = 4096 below): 136 KiB of exchange buffer is one workgroup per CU either way; the generic kernel then runs its load,
// transform and store phases with nothing beside them.  Persistent with the next transform's loads in flight; table entries
// from global memory in every pass (kTwGlobal: no room for them in registers or LDS).
template <bool INV> struct PersistCfg<13, ComplexIO<double, INV>> {
    static constexpr int BLOCK = 512, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 1;
    static constexpr bool kInvInLds = false, kTwLastInLds = false, kTwGlobal = true;
};
template <bool INV> struct PersistCfg<12, ComplexIO<double, INV>> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 2;
    static constexpr bool kInvInLds = false, kTwLastInLds = false, kTwGlobal = true;
};
#ifndef KOFFT_C12_DEPTH
#define KOFFT_C12_DEPTH 1  // (round 5 A/B: two transforms ahead for the n = 4096 kernels -- three register sets, 166 VGPRs)
#endif
template <class IO> struct PersistCfg<12, IO> : PersistCfgBase<IO> { static constexpr int BLOCK = 256, MINW = 2, WG_PER_CU = 2, DEPTH = KOFFT_C12_DEPTH; };
template <class IO> struct PersistCfg<11, IO> : PersistCfgBase<IO> { static constexpr int BLOCK = 256, MINW = 2, WG_PER_CU = 2; };
template <class IO> struct PersistCfg<10, IO> : PersistCfgBase<IO> {
    static constexpr int BLOCK = 256, MINW = IO::kLeanRegisters ? 3 : 2, WG_PER_CU = MINW;
};
// rfft 2048 (m = 1024, one wavefront per transform): with half the grid (PersistGrid below: fewer concurrent rows) a CU
// runs ONE wavefront per SIMD, and one transform of work is too short for the next one's loads to land: two ahead.
#ifndef KOFFT_RFFT10_DEPTH
#define KOFFT_RFFT10_DEPTH 2
#endif
template <> struct PersistCfg<10, RfftIO<float>> : PersistCfgBase<RfftIO<float>> {
    static constexpr int BLOCK = 256, MINW = 2, WG_PER_CU = 2, DEPTH = KOFFT_RFFT10_DEPTH;
};
#ifndef KOFFT_STFT10_DEPTH
#define KOFFT_STFT10_DEPTH 1
#endif
template <> struct PersistCfg<10, StftIO> : PersistCfgBase<StftIO> {
    static constexpr int BLOCK = 256, MINW = 2, WG_PER_CU = 2, DEPTH = KOFFT_STFT10_DEPTH;
};
// stft_magnitudes runs the n-point COMPLEX transform of each real frame (visual/spectrogram.rs:52-76): twice STFT's butterflies for half
// its output -- compute-limited at every window; window in LDS and three workgroups per CU: n = 1024 0.210 -> 0.195 ms (config 4's stream)
template <> struct PersistCfg<10, StftMagIO> {
    // (the last pass's twiddles from an LDS copy -- no spill at three workgroups -- measured slower: 0.205 -> 0.220 ms)
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = 3, WG_PER_CU = 3, DEPTH = KOFFT_STFT10_DEPTH;
    static constexpr bool kInvInLds = true, kTwLastInLds = false;
};
// irfft prefetches two row elements per output: the last pass reads its twiddles from LDS to stay inside 256 VGPRs
#ifndef KOFFT_IRFFT10_WG
#define KOFFT_IRFFT10_WG 2
#endif
#ifndef KOFFT_IRFFT10_DEPTH
#define KOFFT_IRFFT10_DEPTH 1
#endif
#ifndef KOFFT_IRFFT10_TWLDS
#define KOFFT_IRFFT10_TWLDS true
#endif
template <> struct PersistCfg<10, IrfftIO<float>> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = KOFFT_IRFFT10_WG, WG_PER_CU = KOFFT_IRFFT10_WG, DEPTH = KOFFT_IRFFT10_DEPTH;
    static constexpr bool kInvInLds = true, kTwLastInLds = KOFFT_IRFFT10_TWLDS;
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
// (round 4: 8 points per thread -- four passes, 104-107 registers, four wavefronts per SIMD -- measured 8 % slower at both sizes)
template <> struct PersistCfg<12, StftIO> : PersistCfgStftBig<StftIO> {};
template <> struct PersistCfg<11, StftIO> : PersistCfgStftBig<StftIO> {};
// (the magnitude kernels spill 21 / 11 registers at three workgroups per CU -- tools/kernel_regs.py -- and are still 13 % faster than at two:
// n = 2048 / 4096 0.219 / 0.226 against 0.250 / 0.254 ms)
template <> struct PersistCfg<12, StftMagIO> : PersistCfgStftBig<StftMagIO> {};
template <> struct PersistCfg<11, StftMagIO> : PersistCfgStftBig<StftMagIO> {};
// rfft 8192 (m = 4096): window pairs in registers so that two workgroups (exchange buffer + post-pass table) fit a CU
template <> struct PersistCfg<12, RfftIO<float>> {
    static constexpr int BLOCK = 256, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 2;
    static constexpr bool kInvInLds = false, kTwLastInLds = false;
};
#ifndef KOFFT_IRFFT13_TWGLOBAL
#define KOFFT_IRFFT13_TWGLOBAL false
#endif
// rfft 16384 (m = 8192): one 512-thread workgroup per CU (exchange buffer 68 KiB + post-pass table 64 KiB)
template <> struct PersistCfg<13, RfftIO<float>> {
    static constexpr int BLOCK = 512, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 1;
    static constexpr bool kInvInLds = false, kTwLastInLds = false;
};
// irfft 16384 (m = 8192): the same, with the pre-pass table in LDS and every pass's twiddles from global memory
template <> struct PersistCfg<13, IrfftIO<float>> {
    static constexpr int BLOCK = 512, NBUF = 1, RL = 4, MINW = 2, WG_PER_CU = 1;
    static constexpr bool kInvInLds = true, kTwLastInLds = false, kTwGlobal = KOFFT_IRFFT13_TWGLOBAL;
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
#ifndef KOFFT_RFFT6_WG
#define KOFFT_RFFT6_WG 3  // with the group-wide stores of round 3 (fft_persist.hip.h) three workgroups beat two by 6 %
#endif
template <> struct PersistGrid<6, RfftIO<float>> { static int wg_per_cu(int, size_t) { return KOFFT_RFFT6_WG; } };
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
    constexpr size_t lds = (size_t)XPB * Cfg::NBUF * persist_slot_elems(L) * sizeof(cpx<T>) +
                           (Cfg::kInvInLds ? (size_t)(1 << L) * sizeof(typename IO::Inv) : 0) +
                           (EPI == EPI_RFFT ? (size_t)(1 << L) * sizeof(cpx<T>) : 0) +
                           (Cfg::kTwLastInLds ? (size_t)(1 << L) / 2 * sizeof(cpx<T>) : 0);
    static_assert(lds * Cfg::WG_PER_CU <= 160 * 1024, "LDS budget");
    auto kern = fft_persist_kernel<T, L, RL, EPI, IO, Cfg>;
    {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    size_t blocks = (size_t)ctx->num_cus * PersistGrid<L, IO>::wg_per_cu(Cfg::WG_PER_CU, batch * sizeof(cpx<T>) << L);
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;  // measurement knob
    if (blocks < 1) blocks = 1;
    const size_t need = (batch + XPB - 1) / XPB;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Cfg::BLOCK), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// The wave-split kernel (fft_split.hip.h): one workgroup per CU, one s_barrier per transform.
template <typename T, int LA, int LB, class IO>
int launch_split(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t batch)
{
    using Gm = SplitGeom<LA, LB>;
    constexpr size_t lds = 2 * (size_t)Gm::N * sizeof(cpx<T>) + 16;  // (+ the two counters of KOFFT_SPLIT_COUNTERS)
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fft_split_persist_kernel<T, LA, LB, IO>;
    {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    size_t blocks = (size_t)ctx->num_cus;
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < 1) blocks = 1;
    if (blocks > batch) blocks = batch;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::TPT), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}


// n = 2^14 with 32 points per thread (fft_split_wide.hip.h): 512 threads, one workgroup per CU, 128-byte runs both ways
template <typename T, int LA, int LB, int QB0, class IO>
int launch_split_wide(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t batch)
{
    using Gm = SplitWideGeom<LA, LB, QB0>;
    constexpr size_t lds = split_wide_lds_bytes<Gm>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fft_split_wide_persist_kernel<T, LA, LB, QB0, IO>;
    {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    size_t blocks = (size_t)ctx->num_cus * ((160 * 1024) / lds);
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < 1) blocks = 1;
    if (blocks > batch) blocks = batch;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::TPT), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// c32 n = 2^15 / c64 n = 2^14: the whole transform in the CU's register file (fft_regfile.hip.h), one 1024-thread workgroup per CU
template <typename T, int LA, int LB, int QB0, class IO>
int launch_regfile(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t batch)
{
    using Gm = RfGeom<T, LA, LB, QB0>;
    constexpr size_t lds = regfile_lds_bytes<Gm, T>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = fft_regfile_persist_kernel<T, LA, LB, QB0, IO>;
    {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    size_t blocks = (size_t)ctx->num_cus;
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < 1) blocks = 1;
    if (blocks > batch) blocks = batch;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::TPT), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// rfft of 2^15 reals: the same kernel with the post-pass as its epilogue (two instances: with / without a row window)
template <typename T, int LA, int LB, int QB0, class IO>
int launch_split_wide_rfft(kofft_hip_ctx *ctx, const IO &io, const cpx<T> *tw, size_t batch)
{
    using Gm = SplitWideGeom<LA, LB, QB0>;
    constexpr size_t lds = split_wide_lds_bytes<Gm>();
    const bool win = io.window != nullptr;
    auto kern = win ? fft_split_wide_persist_kernel<T, LA, LB, QB0, IO, EPI_RFFT, true> : fft_split_wide_persist_kernel<T, LA, LB, QB0, IO, EPI_RFFT, false>;
    {
        static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};
        const int arc = set_dyn_lds_once(ctx, attr_done[win ? 1 : 0], reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    size_t blocks = (size_t)ctx->num_cus * ((160 * 1024) / lds);
    if (ctx->persist_grid_pct > 0) blocks = blocks * (size_t)ctx->persist_grid_pct / 100;
    if (blocks < 1) blocks = 1;
    if (blocks > batch) blocks = batch;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Gm::TPT), lds, ctx->stream, io, tw, batch);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

template <typename T, int N, int EPI, class IO>
int launch_small(kofft_hip_ctx *ctx, const IO &io, size_t batch, const cpx<T> *tw = nullptr)
{
    constexpr int kSmallBlock = small_block_threads<N, IO>();
    const size_t blocks = (batch + kSmallBlock - 1) / kSmallBlock;
    if (blocks > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    constexpr size_t lds = small_lds_bytes<T, N, IO>();
    auto kern = fft_small_kernel<T, N, EPI, IO>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
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
// LDS_CAP: bytes of exchange buffer a workgroup may use -- 80 KiB where two workgroups share a CU (the one-tile-per-
// workgroup kernels), 128 KiB for the persistent kernels (one workgroup per CU): c32 tiles of 2^10-point sub-transforms
// are then 16 columns = 128-byte segments instead of 8 = 64.
template <typename T, class IO, int LS, size_t LDS_CAP = 80 * 1024>
constexpr int big_block()
{
    const int tpt = (1 << LS) >> rl_for(LS);
    int xpb = KOFFT_BIG_XPB(T);
    // (f64: at most 512 threads -- two register sets of 16 values are 128 VGPRs, a 1024-thread block has 128 in all)
    while (xpb > 1 && (xpb * tpt > (sizeof(T) == 8 ? 512 : 1024) || (size_t)xpb * (1 << LS) * 8 > LDS_CAP)) xpb /= 2;
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
    case 0:
        if constexpr (IO::kLen1) return launch_small<T, 1, EPI>(ctx, io, batch);
        else return KOFFT_ERR_UNSUPPORTED;  // (never: the policy's callers handle n = 1 themselves)
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
    if constexpr (sizeof(T) == 4 && !IO::kSlotMinor) if (L == 5 && ctx->small32) return launch_small<T, 32, EPI>(ctx, io, batch, tw);
    if constexpr (sizeof(T) == 8 && EPI == EPI_STORE && io_split_ok<IO>::value && IO::kPersist) {
        if (L == 13 && ctx->use_persist && ctx->persist64 && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 13, EPI>(ctx, io, tw, batch);
        // n = 4096: two 256-thread workgroups per CU, same box 0.64-0.65 -> 0.67-0.68; n = 2048 measured too: no change (0.70), left
        if (L == 12 && ctx->use_persist && ctx->persist64 && batch >= (size_t)ctx->num_cus * 8) return launch_persist<T, 12, EPI>(ctx, io, tw, batch);
    }
    if constexpr (sizeof(T) == 4 && IO::kPersist) if (ctx->use_persist) {
        // streaming sizes: enough transforms to give every resident workgroup several iterations
        if constexpr (EPI == EPI_STORE && io_split_ok<IO>::value) {
            if (L == 14 && ctx->use_split && batch >= (size_t)ctx->num_cus * 4) return launch_split_wide<T, 7, 7, 2>(ctx, io, tw, batch);
        }
        if constexpr (EPI == EPI_STORE && IO::kPersistMaxLog2 >= 13) {
            if constexpr (io_split_ok<IO>::value) {
                if (L == 13 && ctx->use_split && batch >= (size_t)ctx->num_cus * 4) return launch_split<T, 7, 6>(ctx, io, tw, batch);
            }
            if (L == 13 && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 13, EPI>(ctx, io, tw, batch);
            if (L == 12 && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 12, EPI>(ctx, io, tw, batch);
        }
        if constexpr (EPI == EPI_STORE && io_pairs_in_wave<IO>::value) {
            if (L == 14 && ctx->use_split && ctx->rfft14_wide && batch >= (size_t)ctx->num_cus * 4) return launch_split_wide<T, 7, 7, 2>(ctx, io, tw, batch);
            if (L == 13 && ctx->rfft13_persist && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 13, EPI>(ctx, io, tw, batch);
        }
        if constexpr (EPI == EPI_RFFT) {
            if (L == 14 && ctx->use_split && ctx->rfft14_wide && batch >= (size_t)ctx->num_cus * 4)
                return launch_split_wide_rfft<T, 7, 7, 2>(ctx, io, tw, batch);
            if (L == 13 && ctx->rfft13_persist && batch >= (size_t)ctx->num_cus * 4) return launch_persist<T, 13, EPI>(ctx, io, tw, batch);
        }
        if constexpr (EPI == EPI_RFFT || (EPI == EPI_STORE && IO::kPersistMaxLog2 == 12)) {
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

template <typename T> constexpr int max_log2_big() { return 26; }
// complex lengths fft_dev takes: powers of two up to 2^26, anything else (Bluestein, m = next_pow2(2n-1)) up to 2^25
inline bool complex_len_ok(size_t n) { return n <= (size_t(1) << (is_pow2(n) ? 26 : 25)); }
// inner lengths the fused real / STFT kernels cover (one workgroup per transform); the rest is composed (real_impl.hip.h)
template <typename T> inline bool fused_len_ok(size_t m) { return is_pow2(m) && m <= (size_t(1) << max_log2<T>()); }

inline int ensure_real_tmp(kofft_hip_ctx *ctx, size_t bytes)
{
    if (ctx->real_tmp_bytes >= bytes) return KOFFT_OK;
    if (ctx->real_tmp) KOFFT_HIP_TRY(ctx, hipFree(ctx->real_tmp));
    ctx->real_tmp = nullptr;
    ctx->real_tmp_bytes = 0;
    KOFFT_HIP_TRY(ctx, hipMalloc(&ctx->real_tmp, bytes));
    ctx->real_tmp_bytes = bytes;
    return KOFFT_OK;
}

// ---- typed device-pointer entry points, one translation unit per family ---------------------------------------------
template <typename T>
int fft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch, int inverse);  // k_complex_f32/f64.hip
template <typename T>
int fft_axis2_dev(kofft_hip_ctx *ctx, T *d_data, int LT, int I, size_t blocks, int inverse);  // k_complex_f32/f64.hip: ndfft's long axes in two passes
bool fft2d_fused_ok(const kofft_hip_ctx *ctx, size_t rows, size_t cols);                      // k_nd_fused.hip: shapes the two-pass 2-D route takes
int fft2d_fused_c32(kofft_hip_ctx *ctx, float *d_data, size_t rows, size_t cols, int inverse);  // ... rows + 2 column stages, then one column-tile pass
template <typename T>
int fft_big_windowed_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, const T *d_window, size_t m, size_t batch);  // factor path, window in the first load
template <typename T>
int fft_radix4_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch, int inverse);  // fft.rs:1455-1548, the reference's bytes (inverse: FftPlan::ifft's loop around it)
template <typename T>
int rfft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, const T *d_window, size_t n, size_t batch);  // k_real_f32/f64.hip
template <typename T>
int irfft_dev(kofft_hip_ctx *ctx, const T *d_in, T *d_out, size_t n, size_t batch);
int stft_bluestein_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len, const float *d_window, size_t n, size_t start0, size_t hop,
                       float *d_out, size_t count, bool *done);  // k_complex_f32.hip (complex_impl.hip.h)
int stft_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len, const float *d_window, size_t win_len, size_t start0,
             size_t hop, float *d_out, size_t count);  // k_stft.hip
int istft_dev(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window, size_t win_len, size_t hop,
              float *d_output, size_t out_len, float *d_scratch, size_t scratch_len, int mode = 1, size_t start0 = 0);
int stft_mag_dev(kofft_hip_ctx *ctx, const float *d_samples, size_t len, size_t win_len, size_t hop, float *d_mags,
                 size_t frames, float *d_max);
template <typename T>
int fft_nd_dev(kofft_hip_ctx *ctx, T *d_data, size_t depth, size_t rows, size_t cols, int inverse);  // k_nd.hip

}  // namespace host
}  // namespace kofft
