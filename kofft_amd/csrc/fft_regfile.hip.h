// fft_regfile.hip.h -- one workgroup per transform of 256 KiB: c32 n = 2^15, c64 n = 2^14 (round 4).
//
// The single-pass kernels stopped where a transform no longer fits the CU's 160 KiB of LDS (c32 2^14, c64 2^13), and the next
// size fell to two passes over HBM (0.31 / 0.35 of the roofline against 0.60 / 0.64).  But the CU's REGISTER FILE holds 512 KiB:
// 1024 threads x 64 data registers are the whole transform, and the LDS only has to carry the exchanges -- in two rounds, the
// real parts and then the imaginary parts, through cells of ONE real (N cells = 128 KiB for both sizes).
//
// Structure = the wave-split kernels (fft_split.hip.h, fft_split_wide.hip.h): N = 2^(LA + LB), R = 32 (f32) / 16 (f64) points
// per thread, 16 wavefronts of 64 R points each:
//   phase A  stages 0 .. LA-1: wavefront w owns CA = 64 R >> LA adjacent columns.  Pass A0 = log2 R stages on the thread's
//            values (k = 0: compile-time table indices, scalar loads), wave-local exchange, pass A1 = LA - log2 R stages
//            (entries from an LDS table: they depend on the position only);
//   block-wide exchange, cell (K, j);
//   phase B  stages LA .. L-1 of row K: wavefront w owns RB = 64 R >> LB adjacent rows.  Pass B0 = QB0 stages (entries depend
//            on K only: LDS table [K][2^QB0 - 1], read as they are used), wave-local exchange, pass B1 = 3 stages on groups of 8
//            values -- their 7 entries per group depend on the thread AND the row and are read from the L2-resident table per
//            transform, a group ahead (f32) / before the block-wide exchange and behind the first group (f64): there is no
//            register left to keep them.
// No register set for a prefetched transform either: the next transform's loads go into a group's registers as soon as that
// group's results have been stored.  What overlaps a workgroup's load and store phases is the OTHER 255 CUs.
// Same butterflies, same table entries T_n[k * n2] (fft.rs:836-898), same order per output value as every other kernel here:
// tools/split_model.py runs the decomposition (Geom(LA, LB, rlog, qa0 = rlog, qb0)) against numpy and the reference's
// per-stage index sets.
// LDS: cell(K, j) = K * 2^LB + (j ^ F(K) ^ G(j >> 5)), one real per cell; F, G from tools/split_model.py (CELL_BYTES = 4:
// ds_write_b32 / ds_read_b32 serve two groups of 32 lanes on 32 banks, a 2-way write conflict is free; 8: the rules of the
// 8-byte cells) -- no conflict cycle in any of the six access shapes.  XOR-linear: every address is base(thread) ^ const(register).
// Barriers per transform: one before the first LDS write (every wavefront has read the previous transform out of its rows)
// and three inside the block-wide exchange (re written | re read | im written | im read).
#pragma once

#include "fft_split.hip.h"

#ifndef KOFFT_RF_SPLIT_BARRIER
#define KOFFT_RF_SPLIT_BARRIER 1
#endif
// Measurement hooks (variant builds only, tools/build_variant.sh; the product defines none of them):
//   KOFFT_RF_NO_MEM     the arithmetic and the exchanges without global traffic inside the loop
//   KOFFT_RF_COPY_ONLY  the loads and stores alone
//   KOFFT_RF_STAMPS     s_memtime at the phase boundaries (tools/rf_stamps.py)
namespace kofft {

#ifdef KOFFT_RF_STAMPS /* diagnostic builds only (tools/rf_stamps.py): s_memtime at the phase boundaries, first 4 workgroups x 8 transforms */
static __device__ unsigned long long g_rf_stamps[4 * 8 * 16 * 16];
#define KOFFT_RF_STAMP(id)                                                                                              \
    if (lane == 0 && blockIdx.x < 4 && iter < 8) g_rf_stamps[((blockIdx.x * 8 + iter) * 16 + w) * 16 + (id)] = __builtin_amdgcn_s_memtime();
#else
#define KOFFT_RF_STAMP(id)
#endif

template <typename T, int LA, int LB, int QB0> struct RfSwizzle;
template <> struct RfSwizzle<float, 8, 7, 4> {
    static constexpr int F[8] = {25, 9, 5, 10, 16, 25, 27, 18};
    static constexpr int G[2] = {1, 3};
};
template <> struct RfSwizzle<float, 7, 8, 5> {
    static constexpr int F[7] = {25, 10, 20, 7, 5, 25, 27};
    static constexpr int G[3] = {18, 1, 3};
};
template <> struct RfSwizzle<double, 7, 7, 4> {  // = SplitSwizzle<7, 7>: the 16-points-per-thread shapes on 8-byte cells
    static constexpr int F[7] = {15, 4, 18, 30, 8, 4, 26};
    static constexpr int G[2] = {0, 0};
};

template <typename T, int LA_, int LB_, int QB0_>
struct RfGeom {
    static constexpr int LA = LA_, LB = LB_, L = LA + LB, N = 1 << L;
    static constexpr int RLOG = sizeof(T) == 4 ? 5 : 4, R = 1 << RLOG, TPT = N / R, PW = 64 * R, W = N / PW;
    static constexpr int QA0 = RLOG, QA1 = LA - RLOG, QB0 = QB0_, QB1 = LB - QB0_;
    static constexpr int CA = PW >> LA, RB = PW >> LB;            // columns / rows per wavefront
    static constexpr int TA = (1 << LA) / R, TB = (1 << LB) / R;  // threads per column / per row
    static constexpr int CELL = (int)sizeof(T);
    static_assert(TPT == 1024 && W == 16, "the whole register file of a CU: 16 wavefronts");
    static_assert(QA1 >= 1 && QA1 <= RLOG && QB0 >= 1 && QB0 <= RLOG && QB1 >= 1 && QB1 <= RLOG, "pass shapes");
    static_assert(LB >= 5 && LB <= 8, "G covers bits 5 .. 7 of j");
    __host__ __device__ static constexpr int f(int K)
    {
        int r = 0;
        for (int i = 0; i < LA; ++i)
            if ((K >> i) & 1) r ^= RfSwizzle<T, LA, LB, QB0>::F[i];
        return r;
    }
    __host__ __device__ static constexpr int gj(int jh)
    {
        int r = 0;
        for (int i = 0; i < LB - 5; ++i)
            if ((jh >> i) & 1) r ^= RfSwizzle<T, LA, LB, QB0>::G[i];
        return r;
    }
    // byte offset of logical cell (K, j); XOR-linear in the bits of (K, j)
    __host__ __device__ static constexpr int cell_bytes(int K, int j) { return ((K << LB) | (j ^ f(K) ^ gj(j >> 5))) * CELL; }
    // register parts (compile-time constants once the loops are unrolled); u = (g, c), c the low Q bits
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
    // element offset of output register u relative to the thread's tauB = (jb << LA) | K
    __host__ __device__ static constexpr int out_reg(int u)
    {
        const int g = u >> QB1, c = u & ((1 << QB1) - 1);
        return (bitrev(c, QB1) << (QB0 + LA)) | ((g * TB) << LA);
    }
    // the register that holds block sb of the output row (block = out_reg / TPT: every thread holds one position of every block)
    __host__ __device__ static constexpr int u_of_block(int sb)
    {
        for (int u = 0; u < R; ++u)
            if ((out_reg(u) / TPT) == sb) return u;
        return 0;
    }
};

template <class Gm, typename T>
constexpr size_t regfile_lds_bytes()
{
    return (size_t)Gm::N * Gm::CELL + (size_t)(1 << Gm::QA0) * ((1 << Gm::QA1) - 1) * sizeof(cpx<T>) +
           (size_t)(1 << Gm::LA) * ((1 << Gm::QB0) - 1) * sizeof(cpx<T>) + 16;  // + the arrival counter
}

// one real per cell, addressed by the integer LDS offset (dynamic LDS starts at 0: checked at kernel entry)
template <typename T>
struct RfCell {
    typedef __attribute__((address_space(3))) T lds_t;
    __device__ __forceinline__ static T ld(int byte_off) { return *(lds_t *)(size_t)(unsigned)byte_off; }
    __device__ __forceinline__ static void st(int byte_off, const T v) { *(lds_t *)(size_t)(unsigned)byte_off = v; }
};

// A policy with a row window (RowWindowIO below): `win` holds the window as N pairs (w[2e], w[2e+1]).
template <class IO, class = void>
struct rf_row_window { static constexpr bool value = false; };
template <class IO>
struct rf_row_window<IO, decltype((void)IO::kRowWindow)> { static constexpr bool value = IO::kRowWindow; };
// rfft_direct's m-point transform of a windowed real row read as m complex values (real_impl.hip.h: rfft_composed_dev at m = 2^15 / 2^14)
template <typename T>
struct RowWindowIO : ComplexIO<T, false> {
    static constexpr bool kRowWindow = true;
    const cpx<T> *__restrict__ win;
};

// irfft_direct's pre-pass on the loads (rfft.rs:491-503): element e of the m-point inverse transform's input is computed from bins e and
// m - e of the (m + 1)-bin row and table entry e.  Bin e is the thread's prefetched element; bin m - e is some other thread's prefetched
// element of the SAME transform (an L2 hit by the time it is wanted) -- no exchange, and the pre-pass kernel's pass over HBM is gone.
template <class IO, class = void>
struct rf_irfft_pre { static constexpr bool value = false; };
template <class IO>
struct rf_irfft_pre<IO, decltype((void)IO::kIrfftPre)> { static constexpr bool value = IO::kIrfftPre; };
template <typename T>
__device__ __forceinline__ cpx<T> csel(const bool c, const cpx<T> a, const cpx<T> b) { return mk<T>(c ? a.re : b.re, c ? a.im : b.im); }
template <typename T>
__device__ __forceinline__ cpx<T> irfft_pre_one(const cpx<T> a, const cpx<T> rb, const cpx<T> tw)
{
    const T half = T(0.5f);
    const cpx<T> bb = mk<T>(rb.re, -rb.im);
    const cpx<T> sum = cadd(a, bb), diff = csub(a, bb);
    const cpx<T> w = mk<T>(tw.re, -tw.im);
    const cpx<T> t = cmul(w, diff);
    const cpx<T> temp = csub(sum, mk<T>(t.im, -t.re));
    return mk<T>(temp.re * half, temp.im * half);
}
template <typename T>
struct IrfftRowIO : ComplexIO<T, true> {  // in: rows of n + 1 bins; out: rows of n complex values = 2n reals; ifft's conj / scale as ComplexIO
    static constexpr bool kIrfftPre = true;
    const cpx<T> *__restrict__ rtab;  // the RfftPlanner table, n entries
    __device__ __forceinline__ rsrc_t in_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(this->in + (cnt > 0 ? xf0 : 0) * (size_t)(this->n + 1), (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)(this->n + 1) * (unsigned)sizeof(cpx<T>));
    }
    // elements between the start of input row xf and the previous 128-byte line boundary (rows are n + 1 bins long: every row starts elsewhere)
    __device__ __forceinline__ int in_row_misalign(size_t xf) const
    {
        return (int)((reinterpret_cast<size_t>(this->in) / sizeof(cpx<T>) + xf * (size_t)(this->n + 1)) & (128 / sizeof(cpx<T>) - 1));
    }
};

// rfft_direct (rfft.rs:425-465) of 2 N reals in ONE pass (round 6): the N-point transform of the packed row z[e] =
// (x[2e], x[2e+1]) (optionally times the row window's pairs, as RowWindowIO) with the post-pass of rfft.rs:450-463 as the kernel's
// epilogue -- out: rows of N + 1 bins.  Y[k] and Y[N - k] are in the same workgroup (the whole transform is), so after pass B1 the results go
// through the exchange region once more, in NATURAL order, the real parts and then the imaginary parts (one real per cell), and every thread
// computes X[k] for k = kk + 1024 s, kk = (thread - a) mod 1024 with a = the output row's offset into its 128-byte line: every wavefront
// stores whole lines although rows are N + 1 values long (the lane rotation of the smaller kernels' EPI_RFFT).
template <class IO, class = void>
struct rf_rfft_epi { static constexpr bool value = false; };
template <class IO>
struct rf_rfft_epi<IO, decltype((void)IO::kRfftEpi)> { static constexpr bool value = IO::kRfftEpi; };
template <typename T, bool WIN>
struct RfftRowIO : ComplexIO<T, false> {  // in: rows of n complex = 2n reals; `out`: rows of n + 1 bins
    static constexpr bool kRowWindow = WIN;
    static constexpr bool kRfftEpi = true;
    const cpx<T> *__restrict__ win;   // the row window as n pairs (WIN), else unused
    const cpx<T> *__restrict__ rtab;  // build_twiddle_table(n), rfft.rs:172-183
    __device__ __forceinline__ rsrc_t out_desc_row(size_t xf) const
    {
        return make_rsrc(this->out + xf * (size_t)(this->n + 1), (unsigned)(this->n + 1) * (unsigned)sizeof(cpx<T>));
    }
    // elements between the start of output row xf and the previous 128-byte line boundary
    __device__ __forceinline__ int row_misalign(size_t xf) const
    {
        return (int)((reinterpret_cast<size_t>(this->out) / sizeof(cpx<T>) + xf * (size_t)(this->n + 1)) & (128 / sizeof(cpx<T>) - 1));
    }
    __device__ __forceinline__ static cpx<T> post_w(cpx<T> w, cpx<T> a, cpx<T> ymk)  // X[k], 1 <= k < n (rfft.rs:454-463)
    {
#ifndef KOFFT_BFLY_NOASM
        if constexpr (sizeof(T) == 4) {
            const v2f wv = {w.re, w.im}, av = {a.re, a.im}, yv = {ymk.re, ymk.im};
            const v2f x = rfft_post_f32_pk(wv, av, yv);
            return mk<T>(x.x, x.y);
        }
#endif
        const T half = T(0.5f);
        const cpx<T> b = mk<T>(ymk.re, -ymk.im);
        const cpx<T> sum = cadd(a, b), diff = csub(a, b);
        const cpx<T> t = cmul(w, diff);
        const cpx<T> temp = cadd(sum, mk<T>(t.im, -t.re));
        return mk<T>(temp.re * half, temp.im * half);
    }
};

template <typename T, int LA, int LB, int QB0, class IO>
__global__ __launch_bounds__(1024, 4) void fft_regfile_persist_kernel(const IO io, const cpx<T> *__restrict__ tw, const size_t batch)
{
    using Gm = RfGeom<T, LA, LB, QB0>;
    using Lds = SplitLds<T, LA, LB, IO>;  // whole complex values: the tables
    using Cell = RfCell<T>;
    constexpr int R = Gm::R, ES = (int)sizeof(cpx<T>);
    constexpr int QA0 = Gm::QA0, QA1 = Gm::QA1, QB1 = Gm::QB1;
    constexpr int EA = (1 << QA1) - 1, EB0 = (1 << QB0) - 1, EB1 = (1 << QB1) - 1;
    constexpr int TABLE_A = Gm::N * Gm::CELL;                 // [k < 2^QA0][EA]
    constexpr int TABLE_B = TABLE_A + (1 << QA0) * EA * ES;   // [K < 2^LA][EB0]
    constexpr int COUNTER = TABLE_B + (1 << LA) * EB0 * ES;   // arrivals at "this wavefront has read the transform out of its rows"
    // (f32 only: same box, 512 MiB / 4 GiB per launch 0.401 / 0.48 -> 0.418 / 0.505; in f64 the extra live values spill 9 registers, +-0)
    constexpr bool SPLITBAR = KOFFT_RF_SPLIT_BARRIER && sizeof(T) == 4;
    constexpr int GB1 = R >> QB1;                             // groups of pass B1
    constexpr int NTW = 1;  // B1 entry sets in registers (2 = a group ahead: 14 more registers, 5 spilled, no faster)
    using Raw = typename IO::Raw;
    static_assert(std::is_same<Raw, cpx<T>>::value, "one complex value per element (ComplexIO)");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem_raw != 0u) __builtin_trap();  // integer LDS addressing
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int ja = lane / Gm::CA, x = lane % Gm::CA;
    const int jb = lane / Gm::RB, y = lane % Gm::RB;
    const int col = w * Gm::CA + x, K = w * Gm::RB + y;

    // ---- tables in LDS, built once (TwSubFirst / TwSub index forms, as in fft_split_wide_persist_kernel)
    for (int e = tid; e < (1 << QA0) * EA; e += Gm::TPT) {
        const int k = e / EA, i = e % EA;
        int t = 0;
        while (((2 << t) - 1) <= i) ++t;  // i = (1 << t) - 1 + h
        const int h = i - ((1 << t) - 1);
        int hr = 0;
        for (int b = 0; b < t; ++b) hr |= ((h >> b) & 1) << (t - 1 - b);
        Lds::st(TABLE_A + e * ES, tw[((k << (LA - 1 - QA0 - t)) + (hr << (LA - 1 - t))) << LB]);
    }
    for (int e = tid; e < (1 << LA) * EB0; e += Gm::TPT) {
        const int row = e / EB0, i = e % EB0;
        int t = 0;
        while (((2 << t) - 1) <= i) ++t;
        const int h = i - ((1 << t) - 1);
        int hr = 0;
        for (int b = 0; b < t; ++b) hr |= ((h >> b) & 1) << (t - 1 - b);
        Lds::st(TABLE_B + e * ES, tw[((hr << (LB - 1 - t)) << LA) + (row << (LB - 1 - t))]);
    }

    // Every per-thread constant of the loop lives in THREE registers (two 16-bit fields each: cell and element indices are below
    // 2^15) and is unpacked where it is used -- ten loop-invariant registers were what the allocator spilled (and every reload from
    // scratch is an s_waitcnt vmcnt(0): it also waits for whatever the wavefront has in flight).
    const unsigned packA = (unsigned)(Gm::cell_bytes(ja, col) / Gm::CELL) | ((unsigned)(Gm::cell_bytes(ja << QA1, col) / Gm::CELL) << 16);
    const unsigned packB = (unsigned)(Gm::cell_bytes(K, jb) / Gm::CELL) | ((unsigned)(Gm::cell_bytes(K, jb << QB1) / Gm::CELL) << 16);
    const unsigned packT = (unsigned)((ja << LB) | col) | ((unsigned)((jb << LA) | K) << 16);  // tauA | tauB << 16
    static_assert(Gm::N <= (1 << 15), "16-bit fields");
    auto fresh = [](unsigned v) {  // an opaque copy: what is unpacked from it cannot be hoisted out of the transform loop
        asm volatile("" : "+v"(v));
        return v;
    };
    auto lo_cell = [&](unsigned p) { return (int)((fresh(p) & 0xffffu) * (unsigned)Gm::CELL); };
    auto hi_cell = [&](unsigned p) { return (int)((fresh(p) >> 16) * (unsigned)Gm::CELL); };
    auto tau_a = [&]() { return (int)(fresh(packT) & 0xffffu); };
    auto tau_b = [&]() { return (int)(fresh(packT) >> 16); };
    const rsrc_t twd = make_rsrc(tw, (unsigned)(Gm::N / 2) * (unsigned)ES);
    // pass B1's entries of group g: T[(tauB << (LB-1-QB0-t)) + ((g TB) << (L-1-QB0-t)) + (rev_t(h) << (L-1-t))], t < QB1
    auto load_twb = [&](const int g, cpx<T> *dst) {
        const int tauB = tau_b();
#pragma unroll
        for (int t = 0; t < QB1; ++t)
#pragma unroll
            for (int h = 0; h < (1 << t); ++h)
                dst[(1 << t) - 1 + h] = buf_load_cpx<T, AUX_DEFAULT>(
                    twd, (tauB << (LB - 1 - QB0 - t)) * ES,
                    (((g * Gm::TB) << (Gm::L - 1 - QB0 - t)) + (bitrev(h, t) << (Gm::L - 1 - t))) * ES);
    };
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    if (tid == 0) *(lds_u32 *)(size_t)(unsigned)COUNTER = 0u;
    __syncthreads();  // tables complete

    const size_t step = gridDim.x;
    size_t base = blockIdx.x;
    if (base >= batch) return;

    // which position of every block a thread LOADS for transform xf: its phase-A position tauA -- except for irfft's rows of N + 1 bins, which
    // start anywhere in a 128-byte line: there thread t loads position (t - a) mod TPT (a = the row's offset into its line), so that every 16
    // lanes cover whole lines (streaming loads of partial lines were fetched from HBM again by the neighbouring instruction: 1.6x the row's
    // bytes, rocprofv3 FETCH_SIZE), and the pre-pass's exchange hands every thread its own position back (below)
    auto load_lane_bytes = [&](const size_t xf_) {
        if constexpr (rf_irfft_pre<IO>::value) return (int)((fresh((unsigned)tid) - (unsigned)io.in_row_misalign(xf_)) & (unsigned)(Gm::TPT - 1)) * (int)IO::kRawBytes;
        else return tau_a() * (int)IO::kRawBytes;
    };
    Raw raw[R];
    {
        const rsrc_t d0 = io.in_desc_n(base, 1);
        const int in_lane_bytes = load_lane_bytes(base);
#pragma unroll
        for (int u = 0; u < R; ++u) raw[u] = io.fetch_d(d0, in_lane_bytes, u * Gm::TPT, 0);
    }
    for (int iter = 0;; ++iter) {  // (2^31 transforms per workgroup would be 512 TiB)
        KOFFT_RF_STAMP(0)
        const size_t nbase = base + step;
        const bool more = nbase < batch;  // workgroup-uniform
        const rsrc_t nd = io.in_desc_n(nbase, more ? 1 : 0);  // empty when there is no next transform: the loads return zeros
        const size_t xf = base;
        cpx<T> cur[R];
        if constexpr (rf_row_window<IO>::value) {
            // rfft_direct's packed row with the batched entry's row window (rfft.rs:444-447 after stft.rs:96): element e = (x[2e] * w[2e],
            // x[2e+1] * w[2e+1]).  The window pairs depend on the thread only; held across transforms they would be R more registers
            // (the kernel has none to spare), so they are read per transform, a few at a time, from the L2-resident table (tau_a() is an
            // opaque copy: the loads cannot be hoisted).
            const rsrc_t wd = make_rsrc(io.win, (unsigned)Gm::N * (unsigned)ES);
            const int wl = tau_a() * ES;
#ifndef KOFFT_RF_WINDOW_CHUNKS
            constexpr bool kWinIntoCur = rf_rfft_epi<IO>::value;  // (RowWindowIO -- the two-pass form behind KOFFT_HIP_RFFT_REGFILE_EPI=0 -- spills 9 registers this way: chunks)
#else
            constexpr bool kWinIntoCur = false;
#endif
            if constexpr (kWinIntoCur) {
            // Round 6: the pairs are loaded INTO cur[] -- the registers this step is about to write anyway -- all R at once, and multiplied in
            // place: one L2 round trip per transform.  (Rounds 4-5 took them through 8 (f32) / 4 (f64) temporaries at a time: R / 8 .. R / 4 round
            // trips, each one exposed -- nothing else is in flight at the top of a transform.)
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = buf_load_cpx<T, AUX_DEFAULT>(wd, wl, u * Gm::TPT * ES);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = mk<T>(raw[u].re * cur[u].re, raw[u].im * cur[u].im);
            } else {
            constexpr int WC = sizeof(T) == 4 ? 8 : 4;  // (f64: eight pairs at a time spill a register)
#pragma unroll
            for (int c = 0; c < R; c += WC) {
                cpx<T> wv[WC];
#pragma unroll
                for (int j = 0; j < WC; ++j) wv[j] = buf_load_cpx<T, AUX_DEFAULT>(wd, wl, (c + j) * Gm::TPT * ES);
#pragma unroll
                for (int j = 0; j < WC; ++j) cur[c + j] = mk<T>(raw[c + j].re * wv[j].re, raw[c + j].im * wv[j].im);
            }
            }
        } else if constexpr (rf_irfft_pre<IO>::value) {
            // Round 6: element e = u TPT + p of the pre-pass needs bin e (this thread's prefetched raw[u]: it holds position p = tauA of every
            // block u) and bin N - e = block STEPS - 1 - u, position TPT - p -- ANOTHER thread's prefetched value.  Rounds 4-5 loaded it again
            // (an L2 hit, but a third of the kernel's load instructions: irfft ran 0.30 where the plain transform runs 0.45).  Now the row goes
            // through the exchange region like the rfft epilogue's (below): whole values in natural order, half the blocks per round (round 1:
            // u < H and u >= 3H; both halves closed under u -> STEPS - 1 - u, the mirror's rank is 2H - 1 - rank), every thread writes its
            // value of each block and reads its partner's.  Position 0 mirrors itself one block further (block STEPS - u: folded into its mirror
            // base), inside the round for every block but the pair (H, 3H) -- both are thread 0's own registers -- and block 0 pairs with bin N.
            constexpr int STEPS = Gm::N / Gm::TPT, H = STEPS / 4, BLK = Gm::TPT * ES;
            static_assert(STEPS == R && 2 * H * BLK == Gm::N * Gm::CELL, "pre-pass geometry: half the row fills the region");
            const rsrc_t rd = io.in_desc_n(xf, 1);  // this transform's row: bins 0 .. N
            const rsrc_t td = make_rsrc(io.rtab, (unsigned)Gm::N * (unsigned)ES);
            const int ta = tau_a();
            const int wcell = load_lane_bytes(xf);  // the position this thread loaded (kRawBytes == ES)
            const int ocell = ta * ES;
            const int mcell = ((Gm::TPT - ta) & (Gm::TPT - 1)) * ES + (ta == 0 ? BLK : 0);
            // thread 0's three odd partners, straight from the row (every lane reads the same elements: one request each)
            const cpx<T> bin_n = buf_load_cpx<T, AUX_DEFAULT>(rd, 0, Gm::N * ES);
            const cpx<T> own_h = buf_load_cpx<T, AUX_DEFAULT>(rd, 0, H * BLK), own_3h = buf_load_cpx<T, AUX_DEFAULT>(rd, 0, 3 * H * BLK);
            constexpr int WC = sizeof(T) == 4 ? 4 : 2;
            if constexpr (SPLITBAR) {  // the region is free: every wavefront has read the previous transform out of its rows (see phase A below)
                const unsigned target = 16u * (unsigned)iter;
                while (__builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)(size_t)(unsigned)COUNTER) < target) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            } else {
                __syncthreads();
            }
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                auto in_round = [&](const int sb) { return ((sb < H) || (sb >= 3 * H)) == (rnd == 0); };
                auto rank = [&](const int sb) { return rnd == 0 ? (sb < H ? sb : sb - 2 * H) : sb - H; };
                auto block_of = [&](const int sp) { return rnd == 0 ? (sp < H ? sp : sp + 2 * H) : sp + H; };
                if (rnd == 1) __syncthreads();  // every wavefront has read round 1's partners
#pragma unroll
                for (int u = 0; u < R; ++u)
                    if (in_round(u)) Lds::st(wcell + rank(u) * BLK, raw[u]);
                // the round's table entries travel across the barrier (in chunks between the computations every chunk waited an L2 round trip)
                cpx<T> tv[2 * H];
                split_pin();
#pragma unroll
                for (int j = 0; j < 2 * H; ++j) tv[j] = buf_load_cpx<T, AUX_DEFAULT>(td, ta * ES, block_of(j) * Gm::TPT * ES);
                split_pin();
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 2 * H; c += WC) {
#pragma unroll
                    for (int j = 0; j < WC; ++j) {
                        const int u = block_of(c + j);
                        cpx<T> mv = Lds::ld(mcell + (2 * H - 1 - (c + j)) * BLK);
                        // (field by field: `ta == 0 ? own_3h : mv` on whole values is a select between two OBJECTS -- the f64 build kept them on the
                        // stack and indexed it per lane, 96 of this kernel's 128 bytes of scratch)
                        if (u == H) mv = csel(ta == 0, own_3h, mv);
                        if (u == 3 * H) mv = csel(ta == 0, own_h, mv);
                        if (u == 0) mv = csel(ta == 0, bin_n, mv);
                        const cpx<T> ov = Lds::ld(ocell + (c + j) * BLK);  // bin e, loaded by whichever thread the row's alignment gave it to
                        cpx<T> sv = irfft_pre_one<T>(ov, mv, tv[c + j]);
                        if (u == 0) {  // e = 0 (rfft.rs:491-493): only the real parts of bins 0 and N
                            const T half = T(0.5f);
                            const cpx<T> s0 = mk<T>((ov.re + mv.re) * half, (ov.re - mv.re) * half);
                            sv = csel(ta == 0, s0, sv);
                        }
                        sv.im = -sv.im;  // ifft: conj on the way in (fft.rs:1163-1165)
                        cur[u] = sv;
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u] = io.finish_in(raw[u], io.invariant(0));
        }
#ifdef KOFFT_RF_COPY_ONLY /* measurement only: loads and stores alone */
        if (true) {
            const rsrc_t od = io.out_desc_n(xf, 1);
            const int out_lane_bytes = tau_b() * ES, in_lane_bytes = load_lane_bytes(nbase);
#pragma unroll
            for (int u = 0; u < R; ++u) io.store_d(od, out_lane_bytes, Gm::out_reg(u), cur[u], 0);
#pragma unroll
            for (int u = 0; u < R; ++u) raw[u] = io.fetch_d(nd, in_lane_bytes, u * Gm::TPT, 0);
            if (!more) break;
            base = nbase;
            continue;
        }
#endif
#ifdef KOFFT_RF_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        KOFFT_RF_STAMP(1)
        // ---- phase A
        reg_pass<T, LA, 0, QA0, true>(cur, 0, tw, TwSubFirst{LB});
        KOFFT_RF_STAMP(2)
        if constexpr (rf_irfft_pre<IO>::value) {
            __syncthreads();  // every wavefront has read round 2's partners out of the region (the split-barrier wait was taken before round 1)
        } else if constexpr (SPLITBAR) {
        // Every wavefront has read the previous transform out of its rows -- a SPLIT barrier: a wavefront arrives (below) as soon as
        // its last gathers are done, long before it has stored its results, loaded the next inputs and run pass A0, and only waits
        // here.  With s_barrier in this place the wavefronts whose memory instructions were queued first sat out those of the last
        // ones (s_memtime stamps, tools/rf_stamps.py: 0.1 .. 21 us at this point), and all sixteen then ran the wave-local exchange
        // and pass A1 together instead of behind one another.
        {
            const unsigned target = 16u * (unsigned)iter;
            while (__builtin_amdgcn_readfirstlane(*(volatile lds_u32 *)(size_t)(unsigned)COUNTER) < target) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        } else {
            __syncthreads();  // every wavefront has read the previous transform out of the buffer
        }
        KOFFT_RF_STAMP(3)
        {
            const int a = lo_cell(packA);
#pragma unroll
            for (int u = 0; u < R; ++u) Cell::st(a ^ Gm::a0_out_reg(u), cur[u].re);
        }
        exchange_sync<true>();
        {
            const int a = hi_cell(packA);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u].re = Cell::ld(a ^ Gm::a1_in_reg(u));
        }
        exchange_sync<true>();
        {
            const int a = lo_cell(packA);
#pragma unroll
            for (int u = 0; u < R; ++u) Cell::st(a ^ Gm::a0_out_reg(u), cur[u].im);
        }
        exchange_sync<true>();
        {
            const int a = hi_cell(packA);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u].im = Cell::ld(a ^ Gm::a1_in_reg(u));
        }
        KOFFT_RF_STAMP(4)
        {
            const int tA = TABLE_A + (tau_a() >> LB) * (EA * ES);  // + ja * EA entries
#pragma unroll
            for (int g = 0; g < (R >> QA1); ++g) reg_pass_lds<T, QA1, Lds>(cur + g * (1 << QA1), tA + g * Gm::TA * EA * ES);
        }
        KOFFT_RF_STAMP(5)
        // ---- the block-wide exchange, real parts then imaginary parts
        exchange_sync<true>();  // this wavefront's gathers above are done before its cells are overwritten
        {
            const int a = lo_cell(packA);
#pragma unroll
            for (int u = 0; u < R; ++u) Cell::st(a ^ Gm::a1_out_reg(u), cur[u].re);
        }
        __syncthreads();
        {
            const int a = lo_cell(packB);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u].re = Cell::ld(a ^ Gm::b0_in_reg(u));
        }
        __syncthreads();
        {
            const int a = lo_cell(packA);
#pragma unroll
            for (int u = 0; u < R; ++u) Cell::st(a ^ Gm::a1_out_reg(u), cur[u].im);
        }
        __syncthreads();
        {
            const int a = lo_cell(packB);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u].im = Cell::ld(a ^ Gm::b0_in_reg(u));
        }
        KOFFT_RF_STAMP(6)
        // ---- phase B: this thread's row K from here on, every cell touched below belongs to this wavefront's rows.
        // Pass B1's first entries travel while pass B0 and the wave-local exchange run.
        cpx<T> twb[NTW][EB1];
        split_pin();
#pragma unroll
        for (int b = 0; b < NTW; ++b) load_twb(b, twb[b]);
        split_pin();
        {
            const int tB = TABLE_B + (tau_b() & ((1 << LA) - 1)) * (EB0 * ES);  // + K * EB0 entries
#pragma unroll
            for (int g = 0; g < (R >> QB0); ++g) reg_pass_lds<T, QB0, Lds>(cur + g * (1 << QB0), tB);
        }
        KOFFT_RF_STAMP(7)
        exchange_sync<true>();
        {
            const int a = lo_cell(packB);
#pragma unroll
            for (int u = 0; u < R; ++u) Cell::st(a ^ Gm::b0_out_reg(u), cur[u].re);
        }
        exchange_sync<true>();
        {
            const int a = hi_cell(packB);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u].re = Cell::ld(a ^ Gm::b1_in_reg(u));
        }
        exchange_sync<true>();
        {
            const int a = lo_cell(packB);
#pragma unroll
            for (int u = 0; u < R; ++u) Cell::st(a ^ Gm::b0_out_reg(u), cur[u].im);
        }
        exchange_sync<true>();
        {
            const int a = hi_cell(packB);
#pragma unroll
            for (int u = 0; u < R; ++u) cur[u].im = Cell::ld(a ^ Gm::b1_in_reg(u));
        }
        if constexpr (rf_rfft_epi<IO>::value) {
            // ---- rfft epilogue.  Pass B1 on every group first (results stay in registers), then two rounds through the exchange region.
#pragma unroll
            for (int g = 0; g < GB1; ++g) {
                reg_pass_r<T, QB1>(cur + g * (1 << QB1), twb[g % NTW]);
                split_pin();
                if (g + NTW < GB1) load_twb(g + NTW, twb[g % NTW]);
                split_pin();
            }
            KOFFT_RF_STAMP(8)
            // Two rounds through the exchange region, WHOLE complex values in natural order: half the row fills it.  The row is STEPS blocks of
            // TPT elements, k = s TPT + p; X[k] needs Y[k] and Y[N - k] = block STEPS - 1 - s, position TPT - p (p != 0).  Round 1 carries the
            // blocks s < H and s >= 3H, round 2 the blocks H <= s < 3H (H = STEPS / 4): both sets are closed under s -> STEPS - 1 - s, and with
            // sp = the block's rank in its round the mirror block is 2H - 1 - sp in both.  Every thread holds ONE position of every block after
            // pass B1 (tauB; block = out_reg(u) / TPT, a constant per register), so it writes half of its registers in each round -- no thread
            // and no wavefront sits a round out, and after round 1's writes half of cur[] is dead everywhere -- and it computes the outputs of
            // position kk = (thread - a) mod TPT (a = the output row's offset into its 128-byte line: every 16 lanes store whole lines although
            // rows are N + 1 values long), half of the blocks per round.  Consecutive lanes touch consecutive cells both ways: no conflict.
            // Position 0 is the exception: its mirror is position 0 of block STEPS - s, i.e. rank 2H - sp instead of 2H - 1 - sp -- one block
            // further (folded into its thread's mirror base) -- which stays inside the round for every block but the pair (H, 3H): those two
            // outputs are computed by the thread that HOLDS both values after pass B1 (tauB = 0: thread 0 has position 0 of every block).
            constexpr int STEPS = Gm::N / Gm::TPT, H = STEPS / 4, BLK = Gm::TPT * ES, TPT_LOG = Gm::L - Gm::RLOG;
            static_assert(STEPS == R && (1 << TPT_LOG) == Gm::TPT && 2 * H * BLK == Gm::N * Gm::CELL, "epilogue geometry: half the row fills the region");
            const int a = io.row_misalign(xf);
            const int kk = (int)((fresh((unsigned)tid) - (unsigned)a) & (unsigned)(Gm::TPT - 1));
            const rsrc_t wd = make_rsrc(io.rtab, (unsigned)Gm::N * (unsigned)ES);
            const rsrc_t xd = io.out_desc_row(xf);
            const int wr = tau_b() * ES;                                      // this thread's position in every block
            const int ry = kk * ES;                                           // Y[kk + s TPT]
            const int rm = ((Gm::TPT - kk) & (Gm::TPT - 1)) * ES + (kk == 0 ? BLK : 0);  // Y[N - k]: position TPT - kk of the mirror block (position 0: one block further)
            const int lane_w = kk * ES;
            const int lane_x = kk == 0 ? 0x40000000 : kk * ES;                // blocks H and 3H of position 0 are thread 0's: beyond the descriptor, dropped
            const int in_lane_bytes = load_lane_bytes(nbase);
            constexpr int CH = sizeof(T) == 4 ? 4 : 2;
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                // block s -> (its round, its rank there)
                auto in_round = [&](const int sb) { return ((sb < H) || (sb >= 3 * H)) == (rnd == 0); };
                auto rank = [&](const int sb) { return rnd == 0 ? (sb < H ? sb : sb - 2 * H) : sb - H; };
                auto block_of = [&](const int sp) { return rnd == 0 ? (sp < H ? sp : sp + 2 * H) : sp + H; };
                __syncthreads();  // the region is free: every wavefront has gathered its last values (round 1) / computed round 1's outputs (round 2)
#pragma unroll
                for (int u = 0; u < R; ++u)
                    if (in_round(Gm::out_reg(u) >> TPT_LOG)) Lds::st(wr + rank(Gm::out_reg(u) >> TPT_LOG) * BLK, cur[u]);
                if (rnd == 0 && tid == 0) {  // X[H TPT] and X[3H TPT] (rfft.rs:454-463)
                    const cpx<T> y1 = cur[Gm::u_of_block(H)], y3 = cur[Gm::u_of_block(3 * H)];
                    const cpx<T> w1 = buf_load_cpx<T, AUX_DEFAULT>(wd, 0, H * BLK), w3 = buf_load_cpx<T, AUX_DEFAULT>(wd, 0, 3 * H * BLK);
                    io.store_d(xd, 0, H * Gm::TPT, IO::post_w(w1, y1, y3), 0);
                    io.store_d(xd, 0, 3 * H * Gm::TPT, IO::post_w(w3, y3, y1), 0);
                }
                // the table entries of this round's outputs travel across the barrier (half of cur[] is dead after round 1's writes, all of it
                // after round 2's); in round 2 the next transform's inputs follow them into cur[]'s registers -- BEHIND the table loads: vmcnt
                // counts in issue order, a table entry requested after them would wait for them
                cpx<T> wv[2 * H];
                split_pin();
#pragma unroll
                for (int j = 0; j < 2 * H; ++j) wv[j] = buf_load_cpx<T, AUX_DEFAULT>(wd, lane_w, block_of(j) * BLK);
                if (rnd == 1) {  // (the first half now, the second when half of this round's table entries have been used: registers)
#pragma unroll
                    for (int u = 0; u < R / 2; ++u) raw[u] = io.fetch_d(nd, in_lane_bytes, u * Gm::TPT, 0);
                }
                split_pin();
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 2 * H; c += CH) {
                    if (rnd == 1 && c == H) {
                        split_pin();
#pragma unroll
                        for (int u = R / 2; u < R; ++u) raw[u] = io.fetch_d(nd, in_lane_bytes, u * Gm::TPT, 0);
                        split_pin();
                    }
                    cpx<T> ya[CH], yb[CH];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        ya[j] = Lds::ld(ry + (c + j) * BLK);
                        yb[j] = Lds::ld(rm + (2 * H - 1 - c - j) * BLK);
                    }
                    if (rnd == 1 && c + CH >= 2 * H) {
                        if constexpr (SPLITBAR) {
                            // arrive: this wavefront has read the transform out of the region (the counter add follows the reads in the LDS
                            // queue only if it is issued after the wait)
                            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) __hip_atomic_fetch_add((lds_u32 *)(size_t)(unsigned)COUNTER, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int sb = block_of(c + j);
                        cpx<T> x = IO::post_w(wv[c + j], ya[j], yb[j]);
                        if (sb == 0 && kk == 0) x = mk<T>(ya[j].re + ya[j].im, T(0));  // X[0] (rfft.rs:451)
                        io.store_d(xd, (sb == H || sb == 3 * H) ? lane_x : lane_w, sb * Gm::TPT, x, 0);
                        if (sb == 0 && kk == 0) io.store_d(xd, 0, Gm::N, mk<T>(ya[j].re - ya[j].im, T(0)), 0);  // X[N] (rfft.rs:452)
                    }
                }
            }
        } else {
        if constexpr (SPLITBAR) {
        // arrive: this wavefront's rows are free (the values just gathered have landed: the counter add follows them in the LDS queue
        // only if it is issued after the wait)
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add((lds_u32 *)(size_t)(unsigned)COUNTER, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        KOFFT_RF_STAMP(8)
        const rsrc_t od = io.out_desc_n(xf, 1);
        const int out_lane_bytes = tau_b() * ES, in_lane_bytes = load_lane_bytes(nbase);
#pragma unroll
        for (int g = 0; g < GB1; ++g) {  // results leave as each group completes; its registers take the next transform's loads
            reg_pass_r<T, QB1>(cur + g * (1 << QB1), twb[g % NTW]);
            split_pin();
            if (g + NTW < GB1) load_twb(g + NTW, twb[g % NTW]);
#pragma unroll
#ifdef KOFFT_RF_NO_MEM /* measurement only: the arithmetic and the exchanges without global traffic inside the loop */
            for (int u = g << QB1; u < ((g + 1) << QB1); ++u) {
                if (!more) io.store_d(od, out_lane_bytes, Gm::out_reg(u), cur[u], 0);
                raw[u] = cur[u];
            }
#else
            for (int u = g << QB1; u < ((g + 1) << QB1); ++u) io.store_d(od, out_lane_bytes, Gm::out_reg(u), cur[u], 0);
#pragma unroll
            for (int u = g << QB1; u < ((g + 1) << QB1); ++u) raw[u] = io.fetch_d(nd, in_lane_bytes, u * Gm::TPT, 0);
#endif
            split_pin();
        }
        }
        KOFFT_RF_STAMP(9)
        if (!more) break;
        base = nbase;
    }
}

}  // namespace kofft
