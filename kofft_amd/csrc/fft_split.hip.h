// fft_split.hip.h -- one workgroup per transform, every WAVEFRONT owning 1024 of its points: the streaming kernel for
// n = 2^13 / 2^14 (f32), where a transform no longer fits one wavefront's registers (fft_persist.hip.h up to 2^12) and the
// block-synchronised exchanges of the generic layout (three exchanges = six s_barriers per 8192-point transform, eight
// waves waiting on each) were what the kernel spent its time on (round 3: 40 % of the wave cycles neither issuing nor
// waiting on a counter, against 31 % at n = 4096).
//
// Same butterflies, same table entries, same order per output value as every other kernel here (fft.rs:836-898); only the
// assignment of butterflies to threads changes.  N = 2^(LA + LB), 16 points per thread:
//   phase A  stages 0 .. LA-1: 2^LB column transforms of 2^LA points (element i = a * 2^LB + j).  Wavefront w owns the
//            CA = 1024 >> LA adjacent columns j = w * CA + x: register passes of 4 and LA - 4 stages with an exchange that
//            only this wavefront's cells take part in -- wave-synchronous, no s_barrier.
//   ONE block-wide exchange: cell (K, j), K the column transform's output frequency.          <- the only s_barrier
//   phase B  stages LA .. L-1: 2^LA row transforms of 2^LB points with frequency prefix K.  Wavefront w owns the
//            RB = 1024 >> LB adjacent rows K = w * RB + y: passes of 4 and LB - 4 stages, again wave-local.
//   output o = q * 2^LA + K: lanes run over y first -- RB adjacent outputs per store (128-byte segments at LB = 6).
// Table indices are the two-factor forms of fft_big.hip.h (TwSubFirst / TwSub): idx_local << LB in phase A,
// (idx_local << LA) + (K << (LB - 1 - s_local)) in phase B.  tools/split_model.py runs this decomposition on the CPU
// against numpy and checks that every stage uses exactly the reference's table entries.
//
// LDS: two buffers of N cells (alternating per transform, so that a wavefront starting the next transform never writes what
// a slower one still reads: one barrier per transform is enough), cell(K, j) = K * 2^LB + (j ^ F(K)) with F linear over
// GF(2), found by search (tools/split_model.py) so that all six access shapes are conflict-free under the per-instruction
// banking (ds_write_b64: 16 lanes on 32 dword banks; ds_read_b64: 32 on 64).  The cell is XOR-linear in the bits of
// (K, j), so every address is (per-thread base) ^ (compile-time constant of the register): one v_xor per access.
// The local exchanges run IN PLACE in the block buffer: in phase A a wavefront only touches logical cells (., j) of its own
// columns, in phase B only (K, .) of its own rows.
#pragma once

#include <type_traits>

#include "fft_persist.hip.h"

// Diagnostic builds only (tools/ubench_split.hip): phase time stamps.  In the product no stamp executes.
#ifndef KOFFT_SPLIT_STAMP
#define KOFFT_SPLIT_STAMP(id)
#endif

namespace kofft {

template <int LA, int LB> struct SplitSwizzle;
// columns of F (bit i of K -> XOR mask on j), tools/split_model.py: residual conflict cycles 0 for every shape
template <> struct SplitSwizzle<6, 6> { static constexpr int F[6] = {3, 9, 24, 30, 1, 4}; };
template <> struct SplitSwizzle<7, 6> { static constexpr int F[7] = {8, 10, 30, 17, 27, 6, 24}; };
template <> struct SplitSwizzle<6, 7> { static constexpr int F[6] = {15, 4, 18, 30, 1, 4}; };
template <> struct SplitSwizzle<7, 7> { static constexpr int F[7] = {15, 4, 18, 30, 8, 4, 26}; };
template <> struct SplitSwizzle<8, 6> { static constexpr int F[8] = {5, 9, 24, 30, 8, 4, 26, 30}; };

template <int LA, int LB>
struct SplitGeom {
    static constexpr int L = LA + LB, N = 1 << L, R = 16, TPT = N / R, W = N / 1024;
    static constexpr int QA1 = LA - 4, QB1 = LB - 4;
    static constexpr int CA = 1024 >> LA, RB = 1024 >> LB;   // columns / rows per wavefront
    static constexpr int TA = 1 << (LA - 4), TB = 1 << (LB - 4);  // threads per column / per row
    static_assert(QA1 >= 1 && QA1 <= 4 && QB1 >= 1 && QB1 <= 4, "each factor is one pass of 4 stages and one of 1..4");
    __host__ __device__ static constexpr int f(int K)
    {
        int r = 0;
        for (int i = 0; i < LA; ++i)
            if ((K >> i) & 1) r ^= SplitSwizzle<LA, LB>::F[i];
        return r;
    }
    // byte offset of logical cell (K, j); XOR-linear: cell(K1 | K2, j1 | j2) = cell(K1, j1) ^ cell(K2, j2) for disjoint bits
    __host__ __device__ static constexpr int cell_bytes(int K, int j) { return ((K << LB) | (j ^ f(K))) * 8; }
    // register parts (compile-time constants once the loops are unrolled)
    __host__ __device__ static constexpr int a0_out_reg(int c) { return cell_bytes(bitrev(c, 4) << (LA - 4), 0); }
    __host__ __device__ static constexpr int a1_in_reg(int u)
    {
        const int g = u >> QA1, c = u & ((1 << QA1) - 1);
        return cell_bytes(((g * TA) << QA1) | c, 0);
    }
    __host__ __device__ static constexpr int a1_out_reg(int u)
    {
        const int g = u >> QA1, c = u & ((1 << QA1) - 1);
        return cell_bytes((bitrev(c, QA1) << 4) | (g * TA), 0);
    }
    __host__ __device__ static constexpr int b0_in_reg(int c) { return cell_bytes(0, c << (LB - 4)); }
    __host__ __device__ static constexpr int b0_out_reg(int c) { return cell_bytes(0, bitrev(c, 4) << (LB - 4)); }
    __host__ __device__ static constexpr int b1_in_reg(int u)
    {
        const int g = u >> QB1, c = u & ((1 << QB1) - 1);
        return cell_bytes(0, ((g * TB) << QB1) | c);
    }
    // element offset of output register u relative to the thread's tauB = (kb << LA) | K
    __host__ __device__ static constexpr int out_reg(int u)
    {
        const int g = u >> QB1, c = u & ((1 << QB1) - 1);
        return (bitrev(c, QB1) << (4 + LA)) | ((g * TB) << LA);
    }
};

// Policies the wave-split kernel is instantiated for: those whose input is one Raw per element through fetch_d / finish
// (complex, STFT, STFT magnitudes).  irfft pairs row elements k and m - k; rfft has an epilogue: both stay where they are.
template <class IO, class = void>
struct io_split_ok { static constexpr bool value = false; };
template <class IO>
struct io_split_ok<IO, decltype((void)IO::kSplitOk)> { static constexpr bool value = IO::kSplitOk; };

// Per-(size, policy) configuration (host_common.hip.h specialises): kInvInRegs -- the policy's thread-invariant operands
// (STFT window samples) live in 16 registers.
template <int LA, int LB, class IO>
struct SplitCfg {
    static constexpr int BLOCK = SplitGeom<LA, LB>::TPT;
};

template <typename T, int LA, int LB, class IO>
struct SplitState {
    using Gm = SplitGeom<LA, LB>;
    cpx<T> twA1[(16 >> Gm::QA1) * ((1 << Gm::QA1) - 1)];
    cpx<T> twB0[15];
    cpx<T> twB1[(16 >> Gm::QB1) * ((1 << Gm::QB1) - 1)];
    typename IO::Inv inv[16];
    int cA, gA1, cB, gB1;  // LDS byte bases: phase A scatters, A1 gather, phase B block gather + B0 scatter, B1 gather
    int tauA, tauB;
};

template <typename T, int Q>
__device__ __forceinline__ void split_compute(cpx<T> *v, const cpx<T> *twr)
{
#pragma unroll
    for (int g = 0; g < (16 >> Q); ++g) reg_pass_r<T, Q>(v + g * (1 << Q), twr + g * ((1 << Q) - 1));
}

// A transform is two phases with the block-wide exchange between them; BUF = which of the two LDS buffers it lives in (a
// compile-time constant: the caller's loop is unrolled twice).
//
// Memory instructions are SPREAD over the phases instead of issued in two bursts (round 3, from s_memtime stamps: with all
// 16 stores at the end and all 16 prefetch loads at the start of the next step, a wavefront spent 26 % of a transform stalled
// ISSUING them -- the eight waves reach that point together, 128 KiB of requests queue behind HBM's drain rate):
//   phase A carries, 4 at a time, the loads of the transform after the one it works on -- into the SAME registers its own
//   inputs came from (dead once `finish` has consumed them; the loads then have a whole step to land);
//   phase B carries, 4 at a time, the stores of the transform BEFORE the one it works on (results wait one step in a second
//   register set).  Both phases then cost about the same, which is what lets the two wavefronts of a SIMD run them in
//   opposite order (below).  Loads are issued before the stores they share a step with: s_waitcnt vmcnt counts in issue
//   order, so waiting for a transform's inputs never waits for stores.
template <typename T, int LA, int LB, class IO>
struct SplitLds {
    typedef T vec2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) vec2 lds_vec2;
    // LDS addresses are formed from the integer offset itself: the dynamic region starts at LDS offset 0 (the kernel has no
    // static __shared__; checked at kernel entry), so no "symbol + offset" add is left per access.
    __device__ __forceinline__ static cpx<T> ld(int byte_off)
    {
        const vec2 f = *(lds_vec2 *)(size_t)(unsigned)byte_off;
        return mk<T>(f.x, f.y);
    }
    __device__ __forceinline__ static void st(int byte_off, const cpx<T> v)
    {
        vec2 f;
        f.x = v.re;
        f.y = v.im;
        *(lds_vec2 *)(size_t)(unsigned)byte_off = f;
    }
};

__device__ __forceinline__ void split_pin()  // keeps a chunk of memory instructions where it is written
{
    __builtin_amdgcn_sched_barrier(0);
}

// Phase A of transform xf (inputs in raw[]): stages 0 .. LA-1, results scattered to cells (K, j) of buffer BUF.
template <typename T, int LA, int LB, int BUF, class IO, class LoadChunk>
__device__ __forceinline__ void split_phase_a(const typename IO::Raw *raw, const SplitState<T, LA, LB, IO> &st, const IO &io,
                                              const cpx<T> *__restrict__ tw, const size_t xf, const LoadChunk &load_chunk)
{
    using Gm = SplitGeom<LA, LB>;
    using Lds = SplitLds<T, LA, LB, IO>;
    constexpr int R = 16;
    constexpr int BOFF = BUF * Gm::N * (int)sizeof(cpx<T>);
    static_assert(sizeof(cpx<T>) == 8, "8-byte cells");
    // The addresses base ^ constant are loop-invariant; left alone the compiler hoists all of them out of the transform
    // loop and keeps them in registers (256 VGPRs and 36 spilled).  Opaque copies per phase keep them one v_xor each.
    int cA = st.cA, gA1 = st.gA1;
    asm volatile("" : "+v"(cA), "+v"(gA1));
    cpx<T> cur[R];
    KOFFT_SPLIT_STAMP(0)
    if (io.inside(xf)) {
#pragma unroll
        for (int u = 0; u < R; ++u) cur[u] = io.finish_in(raw[u], st.inv[u]);
    } else {
#pragma unroll
        for (int u = 0; u < R; ++u) cur[u] = io.finish(xf, u * Gm::TPT + st.tauA, raw[u], st.inv[u]);
    }
    KOFFT_SPLIT_STAMP(1)
    split_pin(); load_chunk(0); split_pin();
    // stages 0 .. 3 (k = 0: table indices are compile-time constants -> scalar loads) ...
    reg_pass<T, LA, 0, 4, true>(cur, 0, tw, TwSubFirst{LB});
    KOFFT_SPLIT_STAMP(2)
    split_pin(); load_chunk(1); split_pin();
    // ... wave-local exchange (this wavefront's columns only) ...
#pragma unroll
    for (int u = 0; u < R; ++u) Lds::st(cA ^ (Gm::a0_out_reg(u) ^ BOFF), cur[u]);
    exchange_sync<true>();
#pragma unroll
    for (int u = 0; u < R; ++u) cur[u] = Lds::ld(gA1 ^ (Gm::a1_in_reg(u) ^ BOFF));
    KOFFT_SPLIT_STAMP(3)
    split_pin(); load_chunk(2); split_pin();
    // ... stages 4 .. LA-1
    split_compute<T, Gm::QA1>(cur, st.twA1);
    KOFFT_SPLIT_STAMP(4)
    split_pin(); load_chunk(3); split_pin();
    exchange_sync<true>();  // the gathers above are done before the cells are overwritten (same wavefront: order only)
#pragma unroll
    for (int u = 0; u < R; ++u) Lds::st(cA ^ (Gm::a1_out_reg(u) ^ BOFF), cur[u]);
    KOFFT_SPLIT_STAMP(5)
}

// Phase B (after the block-wide barrier): stages LA .. L-1 of this thread's row K out of buffer BUF; results left in cur[].
template <typename T, int LA, int LB, int BUF, class IO, class StoreChunk>
__device__ __forceinline__ void split_phase_b(cpx<T> *cur, const SplitState<T, LA, LB, IO> &st, const size_t xf, const StoreChunk &store_chunk)
{
    using Gm = SplitGeom<LA, LB>;
    using Lds = SplitLds<T, LA, LB, IO>;
    constexpr int R = 16;
    constexpr int BOFF = BUF * Gm::N * (int)sizeof(cpx<T>);
    int cB = st.cB, gB1 = st.gB1;
    asm volatile("" : "+v"(cB), "+v"(gB1));
    (void)xf;
    KOFFT_SPLIT_STAMP(6)
#pragma unroll
    for (int u = 0; u < R; ++u) cur[u] = Lds::ld(cB ^ (Gm::b0_in_reg(u) ^ BOFF));
    KOFFT_SPLIT_STAMP(7)
    split_pin(); store_chunk(0); split_pin();
    // stages LA .. LA+3 of row K ...
    reg_pass_r<T, 4>(cur, st.twB0);
    KOFFT_SPLIT_STAMP(8)
    split_pin(); store_chunk(1); split_pin();
    exchange_sync<true>();
#pragma unroll
    for (int u = 0; u < R; ++u) Lds::st(cB ^ (Gm::b0_out_reg(u) ^ BOFF), cur[u]);
    exchange_sync<true>();
#pragma unroll
    for (int u = 0; u < R; ++u) cur[u] = Lds::ld(gB1 ^ (Gm::b1_in_reg(u) ^ BOFF));
    KOFFT_SPLIT_STAMP(9)
    split_pin(); store_chunk(2); split_pin();
    // ... and the rest
    split_compute<T, Gm::QB1>(cur, st.twB1);
    KOFFT_SPLIT_STAMP(10)
    split_pin(); store_chunk(3); split_pin();
    KOFFT_SPLIT_STAMP(11)
}

// The loop, per wavefront:      A(0) | bar | B(0) A(1) | bar | B(1) A(2) | bar | ...        (waves 0 .. W/2-1)
//                               A(0) | bar | A(1) B(0) | bar | A(2) B(1) | bar | ...        (waves W/2 .. W-1)
// Between two barriers a wavefront owes phase B of one transform and phase A of the next, in EITHER order: A(t+1) writes
// the other LDS buffer, whose last readers (B(t-1)) every wave left before the barrier.  The two wavefronts that share a
// SIMD take opposite orders, so that one is in a butterfly pass while the other waits on its LDS exchange or issues memory
// instructions -- in the same order they reach the same kind of instruction together and the SIMD's vector unit idles
// through both exchanges (stamps: 58 % of a transform).
// Measured (tools/ubench_split, one box, 8192 x 8192-point c32): same order 0.599 / 0.608 of the roofline, opposite orders
// 0.591 / 0.592 -- the older wavefront of a SIMD wins every issue conflict, so the younger four crawl through their phase A
// while the older four finish both phases and then idle at the barrier; the SIMD is work-conserving either way and the step
// time does not move.  The same-order loop is what is left (201 VGPRs against 233; the other removed in round 4).
// (Measured on top of this loop and removed in round 4 -- git history before it has the code: the s_barrier replaced by two LDS
// counters, a split barrier: 0.585 / 0.582 against 0.591 / 0.583; memory instructions two or one at a time between the STAGES of
// the passes instead of four at a time between the passes: 0.602-0.634 against 0.607-0.618; no sched_barrier pins: 0.578-0.588.)
template <typename T, int LA, int LB, class IO>
__global__ __launch_bounds__((SplitGeom<LA, LB>::TPT), (SplitGeom<LA, LB>::TPT / 256)) void fft_split_persist_kernel(const IO io, const cpx<T> *__restrict__ tw,
                                                                                        const size_t batch)
{
    using Gm = SplitGeom<LA, LB>;
    constexpr int R = 16;
    using Raw = typename IO::Raw;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem_raw != 0u) __builtin_trap();  // SplitLds's addressing
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ja = lane / Gm::CA, x = lane % Gm::CA;  // phase A: thread ja of column x of this wavefront
    const int jb = lane / Gm::RB, y = lane % Gm::RB;  // phase B: thread jb of row y
    const int col = w * Gm::CA + x, K = w * Gm::RB + y;

    SplitState<T, LA, LB, IO> st;
    st.tauA = (ja << LB) | col;
    st.tauB = (jb << LA) | K;
    st.cA = Gm::cell_bytes(ja, col);
    st.gA1 = Gm::cell_bytes(ja << Gm::QA1, col);
    st.cB = Gm::cell_bytes(K, jb);
    st.gB1 = Gm::cell_bytes(K, jb << Gm::QB1);
    // ---- table entries this thread uses in every transform, fetched once (fft_big.hip.h: TwSubFirst / TwSub)
    {
        constexpr int Q = Gm::QA1;
#pragma unroll
        for (int g = 0; g < (16 >> Q); ++g) {
            const int k = ja + g * Gm::TA;
#pragma unroll
            for (int t = 0; t < Q; ++t)
#pragma unroll
                for (int h = 0; h < (1 << t); ++h)
                    st.twA1[g * ((1 << Q) - 1) + (1 << t) - 1 + h] = tw[((k << (LA - 1 - 4 - t)) + (bitrev(h, t) << (LA - 1 - t))) << LB];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < (1 << t); ++h) st.twB0[(1 << t) - 1 + h] = tw[((bitrev(h, t) << (LB - 1 - t)) << LA) + (K << (LB - 1 - t))];
    {
        constexpr int Q = Gm::QB1;
#pragma unroll
        for (int g = 0; g < (16 >> Q); ++g) {
            const int k = jb + g * Gm::TB;
#pragma unroll
            for (int t = 0; t < Q; ++t)
#pragma unroll
                for (int h = 0; h < (1 << t); ++h)
                    st.twB1[g * ((1 << Q) - 1) + (1 << t) - 1 + h] =
                        tw[(((k << (LB - 1 - 4 - t)) + (bitrev(h, t) << (LB - 1 - t))) << LA) + (K << (LB - 1 - 4 - t))];
        }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) st.inv[u] = io.invariant(u * Gm::TPT + st.tauA);

    const size_t step = gridDim.x;
    size_t base = blockIdx.x;
    if (base >= batch) return;

    Raw raw[R];           // inputs of the next phase A (landed, or still in flight)
    cpx<T> oa[R], ob[R];  // results of even / odd transforms of this workgroup: stored during the NEXT phase B
    const int in_lane_bytes = st.tauA * IO::kRawBytes;
    const int out_lane_bytes = st.tauB * (int)sizeof(cpx<T>);
    typename persist_acc<IO>::type acc{};
    if constexpr (io_has_acc<IO>::value) acc = io.acc_init();
    // 4 loads of a transform, through a descriptor that is EMPTY when there is no such transform (fft_persist.hip.h)
    auto loads = [&](const rsrc_t d, const int chunk) {  // loads 4c .. 4c+3
        const int u0 = 4 * chunk, u1 = u0 + 4;
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (u >= u0 && u < u1) raw[u] = io.fetch_d(d, in_lane_bytes, u * Gm::TPT, 0);
    };
    // 4 stores of a finished transform (an empty descriptor drops them: the first transform has no predecessor)
    auto stores = [&](const cpx<T> *src, const rsrc_t d, const int chunk) {
        const int u0 = 4 * chunk, u1 = u0 + 4;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (u < u0 || u >= u1) continue;
            if constexpr (io_has_acc<IO>::value) io.store_d_acc(d, out_lane_bytes, Gm::out_reg(u), src[u], 0, acc);
            else io.store_d(d, out_lane_bytes, Gm::out_reg(u), src[u], 0);
        }
    };
    {   // prologue: inputs of transform 0, then its phase A (which carries the loads of transform 1)
        const rsrc_t d0 = io.in_desc_n(base, 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) loads(d0, c);
        const rsrc_t d1 = io.in_desc_n(base + step, base + step < batch ? 1 : 0);
        split_phase_a<T, LA, LB, 0>(raw, st, io, tw, base, [&](int c) { loads(d1, c); });
    }
    // One step = the barrier that completes transform t's block exchange, then B(t) and A(t+1) in this wavefront's order.
    // (Two separate loops, not one loop with a branch inside: with both orders in one loop body the register allocator has
    // to reconcile the two paths at every merge -- 256 VGPRs and spills against ~200.)
    {
        bool have_prev = false;
        size_t prev = base;
        //   ONEW: receives the results of t      OPREV: results of t-1, stored during B(t)
#define KOFFT_SPLIT_STEP(OPREV, ONEW, BUF, LEAVE)                                                                    \
    {                                                                                                                \
        const size_t nbase = base + step;                                                                            \
        const bool more = nbase < batch; /* workgroup-uniform */                                                     \
        const rsrc_t n2d = io.in_desc_n(nbase + step, nbase + step < batch ? 1 : 0);                                 \
        const rsrc_t pod = io.out_desc_n(prev, have_prev ? 1 : 0);                                                   \
        __syncthreads(); /* the transform's only s_barrier */                                                        \
        split_phase_b<T, LA, LB, BUF, IO>(ONEW, st, base, [&](int c) { stores(OPREV, pod, c); });                    \
        if (more) split_phase_a<T, LA, LB, 1 - BUF>(raw, st, io, tw, nbase, [&](int c) { loads(n2d, c); });          \
        have_prev = true;                                                                                            \
        prev = base;                                                                                                 \
        if (!more) {                                                                                                 \
            const rsrc_t od = io.out_desc_n(base, 1);                                                                \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) stores(ONEW, od, c);                                       \
            LEAVE;                                                                                                   \
        }                                                                                                            \
        base = nbase;                                                                                                \
    }
        for (;;) {
            KOFFT_SPLIT_STEP(ob, oa, 0, break)
            KOFFT_SPLIT_STEP(oa, ob, 1, break)
        }
#undef KOFFT_SPLIT_STEP
    }
    if constexpr (io_has_acc<IO>::value) io.acc_finish(acc);
}

// (Round 3 had a single-buffer form of this kernel for n = 2^14, fft_split1_persist_kernel: 1024 threads, 16 points per thread, the
// table entries of passes A1 / B0 in LDS.  fft_split_wide.hip.h replaced it at 2^14 (0.43 -> 0.555-0.575) and as two workgroups per CU
// at 2^13 it lost to the kernel above (0.52-0.53 against 0.61-0.63): removed in round 4, git history has it.)

// Q stages on 2^Q values with the entries read from LDS (entry (1 << t) - 1 + h at table + 8 * that).
template <typename T, int Q, class Lds, bool STAGED = false>
__device__ __forceinline__ void reg_pass_lds(cpx<T> *v, const int table_bytes)
{
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const int pos = Q - 1 - t;
        if (STAGED) split_pin();  // one stage's entries at a time (at most 2^(Q-1) live): the scheduler would hoist all 2^Q - 1
#pragma unroll
        for (int h = 0; h < (1 << t); ++h) {
            const cpx<T> w = Lds::ld(table_bytes + ((1 << t) - 1 + h) * (int)sizeof(cpx<T>));
#pragma unroll
            for (int lo = 0; lo < (1 << pos); ++lo) {
                const int c = (h << (pos + 1)) | lo;
                bfly(v[c], v[c | (1 << pos)], w);
            }
        }
    }
}


}  // namespace kofft
