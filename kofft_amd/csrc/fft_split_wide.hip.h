// fft_split_wide.hip.h -- the wave-split kernel with 32 points per thread: a wavefront owns 2048 points (round 3).
//
// fft_split1_persist_kernel<7, 7> (n = 16384, 16 points per thread) hands every wavefront 1024 points = 8 adjacent columns in
// phase A and 8 adjacent rows in phase B: every load and every store instruction touches 64-byte runs, and a copy with that
// access pattern reaches 0.59..0.64 of the roofline where 128-byte runs reach 0.75..0.78 (tools/ubench_seg.hip).  With 2048
// points per wavefront the runs are 16 columns / 16 rows = 128 bytes both ways.  512 threads per transform (two wavefronts per
// SIMD, 256 VGPRs each), one LDS buffer, two s_barriers per transform -- the structure of fft_split1_persist_kernel:
//   phase A  stages 0 .. LA-1: pass A0 = 5 stages on the thread's 32 values (k = 0: compile-time table indices), wave-local
//            exchange, pass A1 = LA - 5 stages on groups of 2^(LA-5) (entries from an LDS table: they depend on the position only);
//   block-wide exchange (cell (K, j));
//   phase B  stages LA .. L-1 of row K: pass B0 = QB0 stages (the entries depend on K only: 2^QB0 - 1 per row, LDS table),
//            wave-local exchange, pass B1 = LB - QB0 stages (5 on all 32 values at QB0 = 2) -- its 31 entries depend on the
//            thread AND the row, one per butterfly in the last stage; they are loaded once and stay in registers across
//            transforms.  (QB0 = 5 would need 31 entries per row in LDS: 31.7 KiB beside the 128 KiB buffer do not fit.)
// Measured on one box each (c32, n = 16384, fraction of the roofline): fft_split1 0.43 -> this kernel 0.50..0.52 with the next
// transform's loads in two halves around pass A0, 0.555..0.575 in four quarters spread over phase A (eight slots of
// four: the same), 0.50 all at once; pass B1's entries re-read per transform -2 %; QB0 = 3 (results
// stored in two halves) the same.  STFT 16384 +1..3 % (compute-limited).  At n = 8192 the same kernel (256 threads, two
// workgroups per CU; <7,6,2>, <7,6,1>, <6,7,2>: 0.57..0.59) loses to the double-buffered fft_split_persist_kernel (0.59..0.62).
// Index maps, table indices and the swizzle: tools/split_model.py (Geom(LA, LB, rlog=5, qa0=5, qb0=QB0)), checked against
// numpy and the reference's per-stage index sets in tests/test_lds_layout.py.
// cell(K, j) = K * 2^LB + (j ^ F(K) ^ G(j >> 5)): G folds the high bits of j into the bank bits -- two lanes of a ds_read_b64
// group that differ only in bit 5 of j (pass B1's gather: 32 consecutive j per thread) share a bank under F alone.
#pragma once

#include "fft_split.hip.h"

namespace kofft {

template <int LA, int LB> struct SplitWideSwizzle;
// tools/split_model.py search(with_g=True): residual conflict cycles 0 for every access shape
template <> struct SplitWideSwizzle<7, 7> {  // QB0 = 2, 3, 4
    static constexpr int F[7] = {2, 9, 24, 5, 1, 4, 26};
    static constexpr int G[2] = {30, 7};
};
template <> struct SplitWideSwizzle<7, 6> {  // QB0 = 1 .. 4
    static constexpr int F[7] = {2, 9, 24, 30, 1, 4, 26};
    static constexpr int G[1] = {30};
};
template <> struct SplitWideSwizzle<6, 7> {  // QB0 = 2, 3, 4
    static constexpr int F[6] = {3, 9, 24, 30, 1, 4};
    static constexpr int G[2] = {26, 30};
};

template <int LA_, int LB_, int QB0_>
struct SplitWideGeom {
    static constexpr int LA = LA_, LB = LB_, L = LA + LB, N = 1 << L, RLOG = 5, R = 32, TPT = N / R, W = N / 2048;
    static constexpr int QA0 = 5, QA1 = LA - 5, QB0 = QB0_, QB1 = LB - QB0_;
    static constexpr int CA = 2048 >> LA, RB = 2048 >> LB;            // columns / rows per wavefront
    static constexpr int TA = (1 << LA) / R, TB = (1 << LB) / R;      // threads per column / per row
    static_assert(QA1 >= 1 && QA1 <= 5 && QB0 >= 1 && QB0 <= 5 && QB1 >= 1 && QB1 <= 5, "pass shapes");
    static_assert(LB >= 5 && LB <= 7, "G covers bits 5 and 6 of j");
    __host__ __device__ static constexpr int f(int K)
    {
        int r = 0;
        for (int i = 0; i < LA; ++i)
            if ((K >> i) & 1) r ^= SplitWideSwizzle<LA, LB>::F[i];
        return r;
    }
    __host__ __device__ static constexpr int gj(int jh)
    {
        int r = 0;
        for (int i = 0; i < LB - 5; ++i)
            if ((jh >> i) & 1) r ^= SplitWideSwizzle<LA, LB>::G[i];
        return r;
    }
    // byte offset of logical cell (K, j); XOR-linear in the bits of (K, j)
    __host__ __device__ static constexpr int cell_bytes(int K, int j) { return ((K << LB) | (j ^ f(K) ^ gj(j >> 5))) * 8; }
    // register parts (compile-time constants once the loops are unrolled); u = (g, c) with c the low Q bits
    __host__ __device__ static constexpr int a0_out_reg(int u) { return cell_bytes(bitrev(u, QA0) << (LA - QA0), 0); }
    __host__ __device__ static constexpr int a1_in_reg(int u)
    {
        const int g = u >> QA1, c = u & ((1 << QA1) - 1);
        return cell_bytes(((g * TA) << QA1) | c, 0);
    }
    __host__ __device__ static constexpr int a1_out_reg(int u)
    {
        const int g = u >> QA1, c = u & ((1 << QA1) - 1);
        return cell_bytes((bitrev(c, QA1) << QA0) | (g * TA), 0);
    }
    __host__ __device__ static constexpr int b0_in_reg(int u)
    {
        const int g = u >> QB0, c = u & ((1 << QB0) - 1);
        return cell_bytes(0, (c << (LB - QB0)) | (g * TB));
    }
    __host__ __device__ static constexpr int b0_out_reg(int u)
    {
        const int g = u >> QB0, c = u & ((1 << QB0) - 1);
        return cell_bytes(0, (bitrev(c, QB0) << (LB - QB0)) | (g * TB));
    }
    __host__ __device__ static constexpr int b1_in_reg(int u)
    {
        const int g = u >> QB1, c = u & ((1 << QB1) - 1);
        return cell_bytes(0, ((g * TB) << QB1) | c);
    }
    // the cell of output register u in the thread's own row: (K, q), q = (bitrev(c) << QB0) | (kb + g * TB) -- register part
    __host__ __device__ static constexpr int y_reg(int u)
    {
        const int g = u >> QB1, c = u & ((1 << QB1) - 1);
        return cell_bytes(0, (bitrev(c, QB1) << QB0) | (g * TB));
    }
    // element offset of output register u relative to the thread's tauB = (kb << LA) | K
    __host__ __device__ static constexpr int out_reg(int u)
    {
        const int g = u >> QB1, c = u & ((1 << QB1) - 1);
        return (bitrev(c, QB1) << (QB0 + LA)) | ((g * TB) << LA);
    }
};

template <class Gm>
constexpr size_t split_wide_lds_bytes()
{
    return (size_t)Gm::N * 8 + (size_t)(1 << Gm::QA0) * ((1 << Gm::QA1) - 1) * 8 + (size_t)(1 << Gm::LA) * ((1 << Gm::QB0) - 1) * 8;
}

// EPI_RFFT (IO = RfftIO<float>): the real-FFT
// post-pass (rfft.rs:450-463) on the transform's results before they leave the CU.  Every thread puts its results Y[o],
// o = q * 2^LA + K, back into its OWN row's cells (K, q) -- wave-local, so only the wavefront's own earlier gathers have to be
// behind it -- and after one more s_barrier thread t computes X[k] for k = 512 s + kk, kk = (t - a) mod 512 (a = the output
// row's offset into its 128-byte line: every wavefront then stores whole lines), from Y[k] = cell(k mod 2^LA, k >> LA),
// Y[m - k] and W[k] (the post-pass table, read from global memory: a third of the transform's input volume, L2-resident).
// Both cells are "per-thread base ^ constant of s" like every other access here: m - k = (2^LA - K) + 2^LA * (2^LB - 1 - q),
// and q = 4 s + q0 complements bit by bit.  (K = 0 -- four threads -- pairs with 2^LB - q instead: address computed per s.)
// WIN (EPI_RFFT with a row window): the thread's 32 window pairs are re-read (L2) at the end of every epilogue, into the registers
// its post-pass table entries have just left, and consumed by the next transform's first lines (resident they are 64 registers
// too many: 118 spilled; pass B1's table entries re-read instead: the same).
// PAIRED input (IO = IrfftIO<float>, irfft of 2^15 reals): value k of the pre-pass (rfft.rs:487-506) needs input[k] and
// input[m - k].  Every element is loaded ONCE (prefetched like any other input), the row goes through the -- at that moment
// free -- exchange buffer in the (K, q) cells of the rfft epilogue, k = 2^LA q + K, and every thread reads its 32 partners back
// (the same "base ^ constant" cells, bit-complemented q); the pre-pass table entries W[k] are requested at the top of the
// transform into the registers the previous results have left.  Two more s_barriers per transform (four in all).
template <typename T, int LA, int LB, int QB0, class IO, int EPI = EPI_STORE, bool WIN = false>
__global__ __launch_bounds__((SplitWideGeom<LA, LB, QB0>::TPT), 2) void fft_split_wide_persist_kernel(const IO io, const cpx<T> *__restrict__ tw, const size_t batch)
{
    using Gm = SplitWideGeom<LA, LB, QB0>;
    using Lds = SplitLds<T, LA, LB, IO>;
    constexpr int R = Gm::R;
    constexpr int EA = (1 << Gm::QA1) - 1, EB0 = (1 << Gm::QB0) - 1, EB1 = (1 << Gm::QB1) - 1;
    constexpr int TABLE_A = Gm::N * (int)sizeof(cpx<T>);                             // [k < 2^QA0][EA]
    constexpr int TABLE_B = TABLE_A + (1 << Gm::QA0) * EA * (int)sizeof(cpx<T>);     // [K < 2^LA][EB0]
    static_assert(sizeof(cpx<T>) == 8, "8-byte cells");
    constexpr bool PAIRED = io_pairs_in_wave<IO>::value;  // IrfftIO<float>
    static_assert(!PAIRED || EPI == EPI_STORE, "paired input with the plain store epilogue");
    using Raw = typename std::conditional<PAIRED, cpx<T>, typename IO::Raw>::type;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem_raw != 0u) __builtin_trap();  // SplitLds's addressing
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ja = lane / Gm::CA, x = lane % Gm::CA;
    const int jb = lane / Gm::RB, y = lane % Gm::RB;
    const int col = w * Gm::CA + x, K = w * Gm::RB + y;

    // ---- tables in LDS, built once (TwSubFirst / TwSub index forms, as in fft_split1_persist_kernel)
    for (int e = tid; e < (1 << Gm::QA0) * EA; e += Gm::TPT) {
        const int k = e / EA, i = e % EA;
        int t = 0;
        while (((2 << t) - 1) <= i) ++t;  // i = (1 << t) - 1 + h
        const int h = i - ((1 << t) - 1);
        int hr = 0;
        for (int b = 0; b < t; ++b) hr |= ((h >> b) & 1) << (t - 1 - b);
        Lds::st(TABLE_A + e * (int)sizeof(cpx<T>), tw[((k << (LA - 1 - Gm::QA0 - t)) + (hr << (LA - 1 - t))) << LB]);
    }
    for (int e = tid; e < (1 << LA) * EB0; e += Gm::TPT) {
        const int row = e / EB0, i = e % EB0;
        int t = 0;
        while (((2 << t) - 1) <= i) ++t;
        const int h = i - ((1 << t) - 1);
        int hr = 0;
        for (int b = 0; b < t; ++b) hr |= ((h >> b) & 1) << (t - 1 - b);
        Lds::st(TABLE_B + e * (int)sizeof(cpx<T>), tw[((hr << (LB - 1 - t)) << LA) + (row << (LB - 1 - t))]);
    }

    const int tauA = (ja << LB) | col;
    const int tauB = (jb << LA) | K;
    const int cA0 = Gm::cell_bytes(ja, col);
    const int gA10 = Gm::cell_bytes(ja << Gm::QA1, col);
    const int cB0 = Gm::cell_bytes(K, jb);
    const int gB10 = Gm::cell_bytes(K, jb << Gm::QB1);
    const int twA1_off = TABLE_A + ja * EA * (int)sizeof(cpx<T>);  // group g: + g * TA * EA entries
    const int twB0_off = TABLE_B + K * EB0 * (int)sizeof(cpx<T>);
    // pass B1's entries: T[(tauB << (LB-1-QB0-t)) + (rev_t(h) << (L-1-t))], t < 5
    const rsrc_t twd = make_rsrc(tw, (unsigned)(Gm::N / 2) * (unsigned)sizeof(cpx<T>));
    constexpr int GB1 = R >> Gm::QB1;  // groups of pass B1
    cpx<T> twb[GB1 * EB1];
    auto load_twb = [&]() {
#pragma unroll
        for (int g = 0; g < GB1; ++g)
#pragma unroll
            for (int t = 0; t < Gm::QB1; ++t)
#pragma unroll
                for (int h = 0; h < (1 << t); ++h)
                    twb[g * EB1 + (1 << t) - 1 + h] = buf_load_cpx<T, AUX_DEFAULT>(
                        twd, (tauB << (LB - 1 - Gm::QB0 - t)) * (int)sizeof(cpx<T>),
                        (((g * Gm::TB) << (Gm::L - 1 - Gm::QB0 - t)) + (bitrev(h, t) << (Gm::L - 1 - t))) * (int)sizeof(cpx<T>));
    };
    constexpr bool TWB_RES = true;  // pass B1's entries stay in registers across transforms (re-read per transform: -2 %; paired input 0.363 against 0.383)
    if (TWB_RES) load_twb();
    typename IO::Inv inv[(EPI == EPI_RFFT || PAIRED) ? 1 : R];
    if constexpr (EPI != EPI_RFFT && !PAIRED) {
#pragma unroll
        for (int u = 0; u < R; ++u) inv[u] = io.invariant(u * Gm::TPT + tauA);
    }
    __syncthreads();  // tables complete

    const size_t step = gridDim.x;
    size_t base = blockIdx.x;
    if (base >= batch) return;

    Raw raw[R];
    cpx<T> winv[WIN ? R : 1];
    auto load_win = [&]() {
        if constexpr (WIN) {
            const rsrc_t wind = make_rsrc(io.window, (unsigned)Gm::N * (unsigned)sizeof(cpx<T>));  // N pairs of reals
#pragma unroll
            for (int u = 0; u < R; ++u) winv[u] = buf_load_cpx<T, AUX_DEFAULT>(wind, tauA * (int)sizeof(cpx<T>), u * Gm::TPT * (int)sizeof(cpx<T>));
        }
    };
    const int in_lane_bytes = tauA * (PAIRED ? (int)sizeof(cpx<T>) : (int)IO::kRawBytes);
    const int out_lane_bytes = tauB * (int)sizeof(cpx<T>);
    typename persist_acc<IO>::type acc{};
    if constexpr (io_has_acc<IO>::value) acc = io.acc_init();
    Raw raw_m{};  // PAIRED: input[m], the partner of k = 0 (the same address in every lane)
    auto fetch1 = [&](const rsrc_t d, const int u) -> Raw {
        if constexpr (PAIRED) return buf_load_cpx<T, AUX_NT>(d, in_lane_bytes, u * Gm::TPT * (int)sizeof(cpx<T>));
        else return io.fetch_d(d, in_lane_bytes, u * Gm::TPT, 0);
    };
    auto loads = [&](const rsrc_t d, const int chunk) {  // 8 of the 32 loads
#pragma unroll
        for (int u = 8 * chunk; u < 8 * chunk + 8; ++u) raw[u] = fetch1(d, u);
        if constexpr (PAIRED)
            if (chunk == 3) raw_m = buf_load_cpx<T, AUX_NT>(d, 0, Gm::N * (int)sizeof(cpx<T>));
    };
    {
        const rsrc_t d0 = io.in_desc_n(base, 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) loads(d0, c);
    }
    load_win();
    for (;;) {
        const size_t nbase = base + step;
        const bool more = nbase < batch;  // workgroup-uniform
        const rsrc_t nd = io.in_desc_n(nbase, more ? 1 : 0);
        const size_t xf = base;
        // (opaque copies: hoisted out of the loop, the 6 x 32 addresses base ^ constant would be kept in registers)
        int cA = cA0, gA1 = gA10, cB = cB0, gB1 = gB10, tA = twA1_off, tB = twB0_off;
        asm volatile("" : "+v"(cA), "+v"(gA1), "+v"(cB), "+v"(gB1), "+v"(tA), "+v"(tB));
        cpx<T> cur[R];
        if constexpr (PAIRED) {
            constexpr int NK = 1 << LA, QT = Gm::TPT >> LA, STEPS = Gm::N / Gm::TPT;
            static_assert(STEPS == R, "one staged element per register");
            const rsrc_t wd = make_rsrc(io.rtab, (unsigned)Gm::N * (unsigned)sizeof(cpx<T>));
            cpx<T> wv[R];
            split_pin();
#pragma unroll
            for (int u = 0; u < R; ++u) wv[u] = buf_load_cpx<T, AUX_DEFAULT>(wd, in_lane_bytes, u * Gm::TPT * (int)sizeof(cpx<T>));
            split_pin();
            // element k = u * TPT + tauA = 2^LA * (QT u + ja) + col: cell (col, QT u + ja); partner m - k: (2^LA - col, complement)
            int stg = Gm::cell_bytes(col, ja);
            int pstg = Gm::cell_bytes((NK - col) & (NK - 1), QT - 1 - ja);
            asm volatile("" : "+v"(stg), "+v"(pstg));
            __syncthreads();  // every wavefront has read the previous transform out of the buffer
#pragma unroll
            for (int u = 0; u < R; ++u) Lds::st(stg ^ Gm::cell_bytes(0, u * QT), raw[u]);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (u % 8 == 0) split_pin();  // eight partners in flight at a time (all 32 hoisted: 64 more registers, 19 spilled)
                int paddr = pstg ^ Gm::cell_bytes(0, (R - 1 - u) * QT);
                if (col == 0) {  // K = 0: m - k = 2^LA * (2^LB - q); q = 0 is k = 0, whose partner is input[m]
                    const int q2 = ((1 << LB) - (u * QT + ja)) & ((1 << LB) - 1);
                    paddr = (q2 ^ Gm::gj(q2 >> 5)) * (int)sizeof(cpx<T>);
                }
                cpx<T> pb = Lds::ld(paddr);
                if (u == 0 && tauA == 0) pb = raw_m;
                cur[u] = io.pre((u == 0 && tauA == 0) ? 0 : 1, raw[u], pb, wv[u]);
            }
        } else if constexpr (EPI == EPI_RFFT && !WIN) {
            // no window: the reference multiplies by exactly 1, which leaves every value as it is (launch_split_wide_rfft checks)
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = raw[u];
        } else if constexpr (EPI == EPI_RFFT) {
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = io.finish_in(raw[u], winv[u]);
        } else if (io.inside(xf)) {
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = io.finish_in(raw[u], inv[u]);
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = io.finish(xf, u * Gm::TPT + tauA, raw[u], inv[u]);
        }
        // the next transform's loads, into the registers `finish` has just consumed: the 32 loads in four quarters spread over phase A
        // (0.555-0.575 of the roofline; in two halves around pass A0 0.50-0.52, all at once 0.50, eight slots of four the same as four
        // of eight -- round 3, one box)
        split_pin(); loads(nd, 0); split_pin();
        reg_pass<T, LA, 0, Gm::QA0, true>(cur, 0, tw, TwSubFirst{LB});
        split_pin(); loads(nd, 1); split_pin();
        __syncthreads();  // every wavefront has read the previous transform out of the buffer
#pragma unroll
        for (int u = 0; u < R; ++u) Lds::st(cA ^ Gm::a0_out_reg(u), cur[u]);
        exchange_sync<true>();
#pragma unroll
        for (int u = 0; u < R; ++u) cur[u] = Lds::ld(gA1 ^ Gm::a1_in_reg(u));
        split_pin(); loads(nd, 2); split_pin();
#pragma unroll
        for (int g = 0; g < (R >> Gm::QA1); ++g)
            reg_pass_lds<T, Gm::QA1, Lds>(cur + g * (1 << Gm::QA1), tA + g * Gm::TA * EA * (int)sizeof(cpx<T>));
        split_pin(); loads(nd, 3); split_pin();
        exchange_sync<true>();
#pragma unroll
        for (int u = 0; u < R; ++u) Lds::st(cA ^ Gm::a1_out_reg(u), cur[u]);
        __syncthreads();  // the block-wide exchange
        if (!TWB_RES) load_twb();
#pragma unroll
        for (int u = 0; u < R; ++u) cur[u] = Lds::ld(cB ^ Gm::b0_in_reg(u));
        {
            cpx<T> tb0[EB0];  // the row's entries of pass B0: the same for every group (k = 0)
#pragma unroll
            for (int i = 0; i < EB0; ++i) tb0[i] = Lds::ld(tB + i * (int)sizeof(cpx<T>));
#pragma unroll
            for (int g = 0; g < (R >> Gm::QB0); ++g) reg_pass_r<T, Gm::QB0>(cur + g * (1 << Gm::QB0), tb0);
        }
        exchange_sync<true>();
#pragma unroll
        for (int u = 0; u < R; ++u) Lds::st(cB ^ Gm::b0_out_reg(u), cur[u]);
        exchange_sync<true>();
#pragma unroll
        for (int u = 0; u < R; ++u) cur[u] = Lds::ld(gB1 ^ Gm::b1_in_reg(u));
        if constexpr (EPI == EPI_RFFT) {
#pragma unroll
            for (int g = 0; g < GB1; ++g) reg_pass_r<T, Gm::QB1>(cur + g * (1 << Gm::QB1), twb + g * EB1);
            exchange_sync<true>();  // this wavefront's last gathers out of its rows are done
#pragma unroll
            for (int u = 0; u < R; ++u) Lds::st(cB ^ Gm::y_reg(u), cur[u]);
            constexpr int NK = 1 << LA, QT = Gm::TPT >> LA;  // threads cover QT = 4 values of q per step
            static_assert(QT >= 1 && (QT & (QT - 1)) == 0 && (Gm::TPT & (Gm::TPT - 1)) == 0, "epilogue geometry");
            const int a = io.row_misalign(xf) & (128 / (int)sizeof(cpx<T>) - 1);
            const int kk = (tid - a) & (Gm::TPT - 1);
            // the post-pass table entries of all this thread's outputs, requested before the barrier (the registers of cur[]
            // are free from here): in chunks between the computations every chunk waited an L2 round trip, 0.35 of the roofline
            constexpr int STEPS = Gm::N / Gm::TPT;
            const rsrc_t wd = make_rsrc(io.rtab, (unsigned)Gm::N * (unsigned)sizeof(cpx<T>));
            const int lane_b = kk * (int)sizeof(cpx<T>);
            cpx<T> wv[STEPS];
            split_pin();
#pragma unroll
            for (int s = 0; s < STEPS; ++s) wv[s] = buf_load_cpx<T, AUX_DEFAULT>(wd, lane_b, s * Gm::TPT * (int)sizeof(cpx<T>));
            split_pin();
            __syncthreads();
            const int Kk = kk & (NK - 1), q0 = kk >> LA;
            int yk = Gm::cell_bytes(Kk, q0);
            int ymk = Gm::cell_bytes((NK - Kk) & (NK - 1), QT - 1 - q0);
            asm volatile("" : "+v"(yk), "+v"(ymk));
            const bool krow0 = Kk == 0;
            const rsrc_t xd = io.out_desc(xf);
            {
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    const cpx<T> ya = Lds::ld(yk ^ Gm::cell_bytes(0, s * QT));
                    int maddr = ymk ^ Gm::cell_bytes(0, (STEPS - 1 - s) * QT);
                    if (krow0) {  // K = 0: m - k = 2^LA * (2^LB - q); q = 0 is k = 0 (value unused)
                        const int q2 = ((1 << LB) - (s * QT + q0)) & ((1 << LB) - 1);
                        maddr = (q2 ^ Gm::gj(q2 >> 5)) * (int)sizeof(cpx<T>);
                    }
                    const cpx<T> yb = Lds::ld(maddr);
                    cpx<T> x = io.post_w(wv[s], ya, yb);
                    if (s == 0 && kk == 0) x = mk<T>(ya.re + ya.im, T(0));  // X[0] (rfft.rs:451)
                    io.store_d(xd, lane_b, s * Gm::TPT, x, 0);
                    if (s == 0 && kk == 0) io.store_d(xd, 0, Gm::N, mk<T>(ya.re - ya.im, T(0)), 0);  // X[m] (rfft.rs:452)
                }
            }
            split_pin();
            load_win();  // (also after the last transform: a conditional reload would keep the old values live across the iteration)
            split_pin();
        } else {
        const rsrc_t od = io.out_desc_n(xf, 1);
#pragma unroll
        for (int g = 0; g < GB1; ++g) {  // results leave as each group completes
            reg_pass_r<T, Gm::QB1>(cur + g * (1 << Gm::QB1), twb + g * EB1);
            split_pin();
#pragma unroll
            for (int u = g << Gm::QB1; u < ((g + 1) << Gm::QB1); ++u) {
                if constexpr (io_has_acc<IO>::value) io.store_d_acc(od, out_lane_bytes, Gm::out_reg(u), cur[u], 0, acc);
                else io.store_d(od, out_lane_bytes, Gm::out_reg(u), cur[u], 0);
            }
            split_pin();
        }
        }
        if (!more) break;
        base = nbase;
    }
    if constexpr (io_has_acc<IO>::value) io.acc_finish(acc);
}

}  // namespace kofft
