// fft_persist.hip.h -- persistent, software-pipelined variant of the workgroup Stockham kernel.
//
// Same arithmetic and pass structure as fft_wg_kernel (fft_wg.hip.h), re-organised around what the
// MI355X memory system needs to stream at its copy ceiling:
//   * the grid is sized to the chip (a few workgroups per CU) and every workgroup loops over
//     transforms with a stride of the whole grid;
//   * the NEXT transform's global loads are issued into a second register set before the current
//     transform is computed, so every resident wave always has a full share of a transform of HBM
//     reads in flight while it works on butterflies and LDS exchanges;
//   * everything that depends on the thread but not on the transform is read ONCE per workgroup into
//     registers: the twiddles of passes 1.. (pass 0's indices are compile-time constants -> scalar
//     loads), the window samples of STFT/rfft framing, the rfft post-pass table entries;
//   * n = 4096 (256 threads per transform): two LDS exchange buffers, 2 barriers per transform;
//     n = 1024 (64 threads = one wavefront per transform): the exchange is wave-synchronous, no
//     s_barrier at all -- the four waves of a workgroup free-run.
// Results are bit-identical to fft_wg_kernel: the same butterflies consume the same table entries.
//
// IO policies used here split `load` into a prefetchable part and a per-thread invariant:
//   Raw   fetch(xf, i)            -- the HBM read (kept in flight in the second register set)
//   Inv   invariant(i)            -- transform-independent operand (window sample), read once
//   cpx   finish(xf, i, raw, inv) -- the value the FFT sees
#pragma once

#include "fft_wg.hip.h"

namespace kofft {

// Twiddles of one register pass for one (k) group, in the order reg_pass consumes them:
// entry (1<<t)-1+h  <-  T[(k << (L-1-S0-t)) + (rev_t(h) << (L-1-t))].
template <typename T, int L, int S0, int Q>
__device__ __forceinline__ void load_pass_twiddles(cpx<T> *twr, const int k, const cpx<T> *__restrict__ tw)
{
#pragma unroll
    for (int t = 0; t < Q; ++t) {
#pragma unroll
        for (int h = 0; h < (1 << t); ++h) {
            const int idx = (k << (L - 1 - S0 - t)) + (bitrev(h, t) << (L - 1 - t));
            twr[(1 << t) - 1 + h] = tw[idx];
        }
    }
}

// reg_pass with the twiddles already in registers.
template <typename T, int Q>
__device__ __forceinline__ void reg_pass_r(cpx<T> *v, const cpx<T> *twr)
{
#pragma unroll
    for (int t = 0; t < Q; ++t) {
        const int pos = Q - 1 - t;
#pragma unroll
        for (int h = 0; h < (1 << t); ++h) {
            const cpx<T> w = twr[(1 << t) - 1 + h];
#pragma unroll
            for (int lo = 0; lo < (1 << pos); ++lo) {
                const int c = (h << (pos + 1)) | lo;
                bfly(v[c], v[c | (1 << pos)], w);
            }
        }
    }
}

template <int L, int RL, int P>
using PassGeom = WgGeom<L, RL, P>;

// Cells between the exchange buffers of consecutive transforms.  Where one wavefront carries four transforms (n = 64, 128) a
// ds_read_b64's 32-lane group spans TWO of them: with the buffers 16 cells (mod 32) apart the second transform's 16 cells --
// a run in natural order (the rfft epilogue, irfft's staged row), or the first one's pattern shifted -- fall on the banks the
// first leaves free (KOFFT_PERSIST_SLOT16; simulated tests/test_lds_layout.py-style: n = 128 63 -> 16 conflict cycles per transform).
#ifndef KOFFT_PERSIST_SLOT16
#define KOFFT_PERSIST_SLOT16 1
#endif
__host__ __device__ constexpr int persist_slot_elems(int L)
{
    int e = lds_elems(1 << L);
    if (KOFFT_PERSIST_SLOT16 && L <= 7)
        while ((e & 31) != 16) ++e;
    return e;
}

// The rfft epilogue of a wavefront that carries several transforms stores the group's output as ONE stream (see persist_transform).
// Measured on one box, rfft32, fraction of the roofline: n = 256 0.63 -> 0.69, n = 512 unchanged; n = 128 0.62 -> 0.60 at
// two workgroups per CU but 0.66 at three (PersistGrid<6>), where the row-by-row form drops to 0.61.
#ifndef KOFFT_PERSIST_RFFT_GROUP_MIN_L
#define KOFFT_PERSIST_RFFT_GROUP_MIN_L 6
#endif

template <typename T, int L, int RL, int P>
__device__ __forceinline__ void persist_load_tw(cpx<T> *twr, const int tau, const cpx<T> *__restrict__ tw)
{
    using Gm = PassGeom<L, RL, P>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g) {
        const int m = tau + g * Gm::TPT;
        load_pass_twiddles<T, L, Gm::S0, Gm::Q>(twr + g * ((1 << Gm::Q) - 1), m >> Gm::JB, tw);
    }
}

template <typename T, int L, int RL, int P>
__device__ __forceinline__ void persist_compute(cpx<T> *v, const cpx<T> *twr)
{
    using Gm = PassGeom<L, RL, P>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g) reg_pass_r<T, Gm::Q>(v + g * (1 << Gm::Q), twr + g * ((1 << Gm::Q) - 1));
}

// pass 0: every group has k == 0 (JB == L - Q), so the twiddle indices are constants
template <typename T, int L, int RL>
__device__ __forceinline__ void persist_compute_p0(cpx<T> *v, const cpx<T> *__restrict__ tw)
{
    using Gm = PassGeom<L, RL, 0>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g) reg_pass<T, L, 0, Gm::Q, true>(v + g * (1 << Gm::Q), 0, tw);
}

// LDS / global addressing.  in_index(tau, u) and out_index(tau, u) are concatenations of disjoint bit
// fields, one set of bits coming from the thread and the other from the register number, so
//   index(tau, u) = index(tau, 0) + index(0, u)     and     lds_pad(a + b) = lds_pad(a) + lds_pad(b)
// (no carry can cross bit 4).  Every access is therefore "one per-thread base (a VGPR computed once) +
// a compile-time constant", which the hardware takes as an immediate offset.
template <typename T, int L, int RL, int P>
__device__ __forceinline__ void persist_lds_gather(cpx<T> *v, const cpx<T> *buf, const int gather_base)
{
    using Gm = PassGeom<L, RL, P>;
    const cpx<T> *p = buf + gather_base;  // gather_base = lds_pad(in_index(tau, 0))
#pragma unroll
    for (int u = 0; u < Gm::R; ++u) v[u] = p[lds_pad(Gm::in_index(0, u))];
}

template <typename T, int L, int RL, int P>
__device__ __forceinline__ void persist_lds_scatter(const cpx<T> *v, cpx<T> *buf, const int scatter_base)
{
    using Gm = PassGeom<L, RL, P>;
    cpx<T> *p = buf + scatter_base;  // scatter_base = lds_pad(out_index(tau, 0)) = lds_pad(tau)
#pragma unroll
    for (int u = 0; u < Gm::R; ++u) p[lds_pad(Gm::out_index(0, u))] = v[u];
}

// Exchange synchronisation: a transform owned by one wavefront needs no s_barrier -- a wave's LDS
// instructions execute in order, so only the compiler must be kept from reordering them.
template <bool WAVE>
__device__ __forceinline__ void exchange_sync()
{
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Prefetch distance of a configuration: CFG::DEPTH transforms ahead (1 unless the configuration says otherwise).
template <class CFG, class = void>
struct persist_depth { static constexpr int value = 1; };
template <class CFG>
struct persist_depth<CFG, decltype((void)CFG::DEPTH)> { static constexpr int value = CFG::DEPTH; };

// (Measured in round 3 and removed in round 4: the next transform's loads riding between the current transform's passes, four at a
// time, instead of one burst before it -- n = 4096 with two workgroups per CU 0.673 against 0.700.)
// CFG::kTwGlobal (f64, n = 8192): no pass keeps its table entries in registers (15 x 4 VGPRs per pass) or in LDS (the exchange
// buffer fills it): every pass after the first reads them from the table in global memory, as fft_wg_kernel does (L2 hits).
template <class CFG, class = void>
struct persist_tw_global { static constexpr bool value = false; };
template <class CFG>
struct persist_tw_global<CFG, decltype((void)CFG::kTwGlobal)> { static constexpr bool value = CFG::kTwGlobal; };

// Everything a thread keeps across transforms.
// CFG (kofft_hip.hip: PersistCfg<L, IO>) carries the per-size, per-policy choices: BLOCK, NBUF, MINW and where the
// thread-invariant operands live -- kInvInLds (window samples in one LDS copy per workgroup instead of R registers)
// and kTwLastInLds (the last pass reads its twiddles from an LDS copy of the table instead of holding them in
// registers): both trade LDS reads for the registers that decide between 2 and 3 waves per SIMD.
template <typename T, int L, int RL, int EPI, class IO, class CFG>
struct PersistState {
    static constexpr int R = 1 << RL;
    static constexpr int NP = (L + RL - 1) / RL;
    static constexpr bool TWG = persist_tw_global<CFG>::value;
    static constexpr bool TW1_REG = !TWG && !(NP == 2 && CFG::kTwLastInLds);
    static constexpr bool TW2_REG = !TWG && NP >= 3 && !(CFG::kTwLastInLds && NP == 3);
    static constexpr bool TW3_REG = !TWG && NP >= 4 && !CFG::kTwLastInLds;
    cpx<T> tw1[TW1_REG ? R - 1 : 1], tw2[TW2_REG ? R - 1 : 1], tw3[TW3_REG ? R - 1 : 1];
    typename IO::Inv inv[CFG::kInvInLds ? 1 : R];  // window samples etc. (registers unless staged in LDS)
    const typename IO::Inv *inv_lds;               // [N], natural order (kInvInLds)
    const cpx<T> *tw_lds;                          // the whole table T_N, N/2 entries (kTwLastInLds)
    const cpx<T> *rt_lds;                          // rfft post-pass table W[k], k < N (EPI_RFFT)
    int g1, g2, g3, sc;  // LDS bases: gathers of passes 1, 2, 3; scatter
};

// The prefetched form of a policy's input: IO::Raw, or -- for a policy that pairs row elements k and m - k inside a
// wavefront (io_pairs_in_wave, transforms of at most 64 threads) -- IO::RawPair plus one extra register for element m.
// Transforms of more than one wavefront pair through the exchange buffer instead (pair_lds): the row goes into LDS in natural
// order once and every thread reads its partners back -- one extra exchange, still half the loads and prefetch registers.
template <class IO, int TPT, bool PAIR = io_pairs_in_wave<IO>::value>
struct persist_raw {
    using type = typename IO::Raw;
    static constexpr int extra = 0;
    static constexpr bool pair = false, pair_lds = false;
};
template <class IO, int TPT>
struct persist_raw<IO, TPT, true> {
    using type = typename IO::RawPair;
    static constexpr int extra = 1;
    static constexpr bool pair = TPT <= 64, pair_lds = TPT > 64;
};

// One transform: raw[] holds its (already landed or still in flight) inputs.
struct NoAcc {};
template <class IO, bool HAS = io_has_acc<IO>::value>
struct persist_acc { using type = NoAcc; };
template <class IO>
struct persist_acc<IO, true> { using type = typename IO::Acc; };

// kTwGlobal configurations (c64 n = 4096 / 8192: every pass after the first reads its table entries from global memory) take the NEXT
// transform's loads in NP parts: one at the top of the step, the others each BEHIND a pass's table loads (round 6).  vmcnt counts in order: with the whole prefetch issued at
// the top of the step, waiting for pass 1's entries waited for the prefetch -- the wavefront sat out the prefetch's landing after pass 0 and ran
// passes 1 .. with nothing in flight.  `part(i)` is the kernel's callback that issues part i; the entries of a pass are loaded into registers
// first (persist_load_tw), so that the order "entries, then the part, then the butterflies" is the program's, not the scheduler's.
// (a part at the top of the step too -- pass 0 reads its entries through scalar loads -- measured better at n = 4096, two workgroups per CU, and worse
// at n = 8192, one: same box, three rounds, against round 5's kernels 4096 +5.4 % with, +3.8 % without; 8192 +1.5 % with, +3.6 % without)
#ifndef KOFFT_TWG_TOP_PART
#define KOFFT_TWG_TOP_PART -1 /* -1: by size; 0 / 1: measurement builds */
#endif
__host__ __device__ constexpr bool persist_twg_top_part(int L) { return KOFFT_TWG_TOP_PART < 0 ? L <= 12 : KOFFT_TWG_TOP_PART != 0; }
struct NoPrefetchParts {
    __device__ __forceinline__ void operator()(int) const {}
};
template <typename T, int L, int RL, int EPI, class CFG, class IO, class Parts = NoPrefetchParts>
__device__ __forceinline__ void persist_transform(const typename persist_raw<IO, ((1 << L) >> RL)>::type *raw,
                                                  const PersistState<T, L, RL, EPI, IO, CFG> &st,
                                                  const IO &io, const cpx<T> *__restrict__ tw, cpx<T> *buf0, cpx<T> *buf1,
                                                  const size_t xf0, const int cnt, const int sub, const int tau,
                                                  typename persist_acc<IO>::type &acc, const Parts &part = Parts{})
{
    // The wavefront's group: cnt (0 .. G) valid transforms starting at xf0; this lane belongs to number `sub`.
    constexpr int NBUF = CFG::NBUF;
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    constexpr int NP = (L + RL - 1) / RL;
    constexpr bool WAVE = (TPT <= 64);
    const size_t xf = xf0 + sub;
    const bool active = sub < cnt;
    using LastG = PassGeom<L, RL, NP - 1>;
    using FirstG = PassGeom<L, RL, 0>;

    cpx<T> cur[R];
    constexpr int GRP = TPT >= 64 ? 1 : 64 / TPT;
    if constexpr (persist_raw<IO, TPT>::pair) {
        static_assert(CFG::kInvInLds, "paired input: the table entries come from the LDS copy");
        io.template finish_pairs<R, TPT>(raw, tau, cur, [&](int u) { return (st.inv_lds + tau)[FirstG::in_index(0, u)]; });
    } else if constexpr (persist_raw<IO, TPT>::pair_lds) {
        static_assert(CFG::kInvInLds && NBUF == 1 && !WAVE, "paired input through the exchange buffer");
        // natural order, UNPADDED (N + 1 cells fit the padded buffer): ascending writes and descending reads are runs of
        // consecutive cells.  The buffer is free once the previous transform's last gather is behind a barrier; the barrier
        // in front of this transform's first scatter (NBUF == 1) closes the reads below.
        exchange_sync<WAVE>();
#pragma unroll
        for (int u = 0; u < R; ++u) buf0[tau + u * TPT] = raw[u];
        if (tau == 0) buf0[N] = raw[R];
        exchange_sync<WAVE>();
        io.template finish_pairs_lds<R, TPT>(raw, tau, buf0 + (N - tau), cur, [&](int u) { return (st.inv_lds + tau)[FirstG::in_index(0, u)]; });
    } else if (io.inside(xf0 + (GRP - 1))) {  // wave-uniform: every frame of the group lies inside the signal -> no range select
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if constexpr (CFG::kInvInLds) cur[u] = io.finish_in(raw[u], (st.inv_lds + tau)[FirstG::in_index(0, u)]);
            else cur[u] = io.finish_in(raw[u], st.inv[u]);
        }
    } else if constexpr (io_frame_rem<IO>::value) {
        const int rem = io.frame_rem(xf);  // the frame's existing samples, worked out once
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int i = FirstG::in_index(0, u) + tau;
            if constexpr (CFG::kInvInLds) cur[u] = io.finish_rem(i, raw[u], (st.inv_lds + tau)[FirstG::in_index(0, u)], rem);
            else cur[u] = io.finish_rem(i, raw[u], st.inv[u], rem);
        }
    } else {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int i = FirstG::in_index(0, u) + tau;
            if constexpr (CFG::kInvInLds) cur[u] = io.finish(xf, i, raw[u], (st.inv_lds + tau)[FirstG::in_index(0, u)]);
            else cur[u] = io.finish(xf, i, raw[u], st.inv[u]);
        }
    }

    persist_compute_p0<T, L, RL>(cur, tw);
    if (NBUF == 1) exchange_sync<WAVE>();  // the previous transform's last LDS gathers are done
    persist_lds_scatter<T, L, RL, 0>(cur, buf0, st.sc);
    exchange_sync<WAVE>();
    persist_lds_gather<T, L, RL, 1>(cur, buf0, st.g1);
    using St = PersistState<T, L, RL, EPI, IO, CFG>;
    // (kTwGlobal: the table addresses depend on tau only -- hoisted out of the transform loop they are 3 x 15 64-bit pointers
    // held in registers, 168 VGPRs spilled; an opaque copy of tau per transform keeps them a few VALU operations per pass)
    int tau_tw = tau;
    if constexpr (St::TWG) asm volatile("" : "+v"(tau_tw));
    auto pass_tw_global = [&](auto pass) {  // kTwGlobal: the pass's entries, then a part of the next transform's loads, then the butterflies
        constexpr int P = decltype(pass)::value;
        using Gp = PassGeom<L, RL, P>;
        cpx<T> twr[Gp::G * ((1 << Gp::Q) - 1)];
        persist_load_tw<T, L, RL, P>(twr, tau_tw, tw);
        __builtin_amdgcn_sched_barrier(0);
        part(P - (persist_twg_top_part(L) ? 0 : 1));
        __builtin_amdgcn_sched_barrier(0);
        persist_compute<T, L, RL, P>(cur, twr);
    };
    if constexpr (St::TWG) pass_tw_global(std::integral_constant<int, 1>{});
    else if constexpr (NP == 2 && CFG::kTwLastInLds) wg_compute<T, L, RL, 1>(cur, io, st.tw_lds, xf, tau);
    else persist_compute<T, L, RL, 1>(cur, st.tw1);
    if constexpr (NP >= 3) {
        if (NBUF == 1) exchange_sync<WAVE>();
        persist_lds_scatter<T, L, RL, 1>(cur, buf1, st.sc);
        exchange_sync<WAVE>();
        persist_lds_gather<T, L, RL, 2>(cur, buf1, st.g2);
        if constexpr (St::TWG) pass_tw_global(std::integral_constant<int, 2>{});
        else if constexpr (NP == 3 && CFG::kTwLastInLds) wg_compute<T, L, RL, 2>(cur, io, st.tw_lds, xf, tau);
        else persist_compute<T, L, RL, 2>(cur, st.tw2);
    }
    if constexpr (NP == 4) {  // n = 8192 (NBUF == 1)
        exchange_sync<WAVE>();
        persist_lds_scatter<T, L, RL, 2>(cur, buf0, st.sc);
        exchange_sync<WAVE>();
        persist_lds_gather<T, L, RL, 3>(cur, buf0, st.g3);
        if constexpr (St::TWG) pass_tw_global(std::integral_constant<int, 3>{});
        else if constexpr (CFG::kTwLastInLds) wg_compute<T, L, RL, 3>(cur, io, st.tw_lds, xf, tau);
        else persist_compute<T, L, RL, 3>(cur, st.tw3);
    }

    if constexpr (EPI == EPI_RFFT) {
        // rfft.rs:450-463: Y in natural order through LDS, then X[k] from Y[k], Y[m-k]
        static_assert(L <= 12 || !CFG::kInvInLds, "rfft epilogue: a staged window does not fit next to an 8192-point exchange buffer and the post-pass table");
        cpx<T> *ybuf = (NP == 3) ? buf0 : buf1;  // not the buffer the last gather read from (when NBUF == 2)
        if (NBUF == 1) exchange_sync<WAVE>();
        // Y goes into the buffer in PLAIN natural order, not the padded exchange layout: every access of this epilogue is a
        // run of consecutive elements (the scatter: lanes tau -> Y[c*TPT + tau]; the reads: ascending k and descending
        // m - k), and 32 consecutive 8-byte cells cover the 64 banks exactly once.  Under the i + (i >> 4) padding such a
        // run crosses one or two pad cells and wraps onto its own first banks: 16 % of this kernel's LDS cycles were
        // bank conflicts (profiles/r01_sq_rfft2048.txt).
        {
            cpx<T> *p = ybuf + tau;
#pragma unroll
            for (int u = 0; u < R; ++u) p[LastG::out_index(0, u)] = cur[u];
        }
        exchange_sync<WAVE>();
        if constexpr (GRP > 1 && L >= KOFFT_PERSIST_RFFT_GROUP_MIN_L) {
            // A wavefront holds GRP transforms whose output rows are adjacent in memory: GRP * (N + 1) values back to back.
            // The stores walk that STREAM, not the rows: lane l of store s handles stream element e = 64 s + l - a, where
            // a is the stream's offset into its 128-byte line, so every store instruction covers four whole lines and a row
            // boundary falls INSIDE an instruction.  Row by row (below), every row's first and last line is written as two
            // partial lines by two different instructions.
            constexpr int LINE = 128 / (int)sizeof(cpx<T>);
            constexpr int ROW = N + 1;
            constexpr int SLOT = persist_slot_elems(L);
            static_assert(NBUF == 1, "the group's slots are addressed SLOT apart (gy + r * SLOT): with NBUF buffers per slot they would be NBUF * SLOT apart");
            constexpr int S = (GRP * ROW + LINE - 1 + 63) / 64;
            const int a = io.row_misalign(xf0) & (LINE - 1);  // wave-uniform
            const rsrc_t od = io.out_desc_back_n(xf0, cnt, LINE);
            const cpx<T> *gy = ybuf - sub * SLOT;  // the group's first slot
            const int total = cnt * ROW;
            const int e0 = sub * TPT + tau - a;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int e = e0 + s * 64;
                if (e >= 0 && e < total) {
                    const int r = e / ROW, k = e - r * ROW;
                    const cpx<T> *y = gy + r * SLOT;
                    const int kc = k < N ? k : N - 1;  // k == N: the table and Y[k] reads are clamped, their values unused
                    const cpx<T> yk = y[kc], ymk = y[N - k];  // k == 0 reads cell N: inside the slot, value unused
                    const cpx<T> p = io.post_w(st.rt_lds[kc], yk, ymk);
                    const cpx<T> x = k == 0 ? mk<T>(yk.re + yk.im, T(0)) : (k == N ? mk<T>(ymk.re - ymk.im, T(0)) : p);
                    io.store_d(od, (e + LINE) * (int)sizeof(cpx<T>), 0, x, 0);
                }
            }
        } else if (active) {
            // Output rows are (N+1) complex values back to back, so a row starts `a` elements past a 128-byte line.
            // Lane tau of store g handles k = g*TPT + tau - a: every store instruction then covers whole lines (the
            // row-contiguous form touches 5 lines per 512-byte store, 2 of them partially; measured +8 % on config 3),
            // at the price of one extra, mostly empty, store (g = R) that also carries X[N].
            // (Round 5: these stores keep their branches.  Without them -- every lane storing R + 1 times, the idle ones beyond the
            // descriptor, their LDS reads clamped -- the compiler's waits relax from vmcnt(31 .. 16) to vmcnt(62), and config 3 runs
            // 0.6 % SLOWER, 3.176 against 3.158 ms over three interleaved same-box runs: two more store instructions per transform in
            // every lane cost more than the waits did.  The plain store epilogue below is the other way round.)
            constexpr int LINE = 128 / (int)sizeof(cpx<T>);
            const int a = io.row_misalign(xf) & (LINE - 1);
            const int k0 = tau - a;  // -15 .. TPT-1
            // descriptor based LINE elements before the row, so that every byte offset below is non-negative
            const rsrc_t od = io.out_desc_back_n(xf0, cnt, LINE);
            const int row_off = sub * (int)io.out_row_bytes();
            const cpx<T> y0 = ybuf[0];
            const cpx<T> *yk = ybuf + k0;         // Y[k0 + g*TPT]
            const cpx<T> *ynk = ybuf + (N - k0);  // Y[N - k0 - g*TPT]
            const cpx<T> *wk = st.rt_lds + k0;
            const int lane_bytes = (k0 + LINE) * (int)sizeof(cpx<T>);
            if (k0 >= 0) {  // g = 0: k = k0 (lanes that fall before the row start sit this one out)
                const cpx<T> p = io.post_w(wk[0], yk[0], ynk[0]);  // k0 == 0 reads cell N: in range, value replaced
                io.store_d(od, lane_bytes, 0, k0 == 0 ? mk<T>(y0.re + y0.im, T(0)) : p, row_off);
            }
#pragma unroll
            for (int g = 1; g < R; ++g)
                io.store_d(od, lane_bytes, g * TPT, io.post_w(wk[g * TPT], yk[g * TPT], ynk[-g * TPT]), row_off);
            if (k0 <= 0) {  // g = R: k = N + k0 <= N; k == N is X[N]
                const int kr = (k0 < 0) ? N + k0 : N - 1;  // clamped for the LDS reads of the lane that holds X[N]
                const cpx<T> p = io.post_w(st.rt_lds[kr], ybuf[kr], ybuf[N - kr]);
                io.store_d(od, lane_bytes, R * TPT, k0 == 0 ? mk<T>(y0.re - y0.im, T(0)) : p, row_off);
            }
        }
        if (NBUF == 2) exchange_sync<WAVE>();  // ybuf is the next transform's first exchange buffer
    } else {
        // NO branch around the stores (round 5).  The descriptor covers exactly the cnt valid transforms of the group (EMPTY when there is
        // none), so a lane without a transform -- sub >= cnt: its row offset lies beyond the descriptor -- stores into the bounds check and
        // nothing else.  With `if (active)` here the compiler could not know at the loop head whether the previous transform's R stores had
        // been issued: vmcnt counts loads and stores in issue order, so it assumed they had not and waited for them to COMPLETE before the
        // current transform's inputs could be used -- `s_waitcnt vmcnt(23 .. 16)` in the n = 4096 kernel where vmcnt(47 .. 32) is enough,
        // vmcnt(15 .. 0) in the n = 1024 STFT kernel: every step began by draining its predecessor's stores.
        const rsrc_t od = io.out_desc_n(xf0, cnt);
        const int lane_bytes = tau * (int)sizeof(cpx<T>);
        const int row_off = sub * (int)io.out_row_bytes();
        if constexpr (io_half_spectrum<IO>::value) {
            // which registers hold bins below N/2 is a compile-time fact (round 5): gather them, take their magnitudes together (round 6:
            // StftMagIO::mags_of), store, and fold the transform's maximum into the accumulator -- for lanes that HAVE a transform only
            // (an inactive lane's loads may land inside the group's descriptor and must not reach the maximum)
            static_assert((LastG::out_index(0, 1) % TPT) == 0 && TPT <= N / 2, "register part and thread part of the index: disjoint bit fields");
            constexpr int NK = R / 2;
            cpx<T> kept[NK];
            int ki = 0;
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (LastG::out_index(0, u) < N / 2) kept[ki++] = cur[u];  // (a constant per u)
            float m[NK];
            const typename IO::Acc top = IO::template mags_of<NK>(kept, m);  // the largest sum of squares (StftMagIO: the accumulator's unit)
            ki = 0;
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (LastG::out_index(0, u) < N / 2) io.store_d_mag(od, lane_bytes, LastG::out_index(0, u), m[ki++], row_off);
            if (active) acc = __builtin_fmaxf(acc, top);
        } else {
#ifdef KOFFT_PERSIST_ACTIVE_BRANCH /* measurement only (tools/build_variant.sh): rounds 1-4's branch around the stores, for same-box A/Bs */
        if (active)
#endif
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if constexpr (io_has_acc<IO>::value) io.store_d_acc(od, lane_bytes, LastG::out_index(0, u), cur[u], row_off, acc);
            else io.store_d(od, lane_bytes, LastG::out_index(0, u), cur[u], row_off);
        }
        }
    }
}

// BLOCK threads carry XPB = BLOCK/TPT transforms at a time (TPT = N/16 threads each).
// NBUF = 1: one LDS exchange buffer per transform slot; NBUF = 2 (block-synchronised sizes): exchanges
// alternate between two buffers, which halves the number of barriers.
template <typename T, int L, int RL, int EPI, class IO, class CFG>
__global__ __launch_bounds__(CFG::BLOCK, CFG::MINW) void fft_persist_kernel(const IO io, const cpx<T> *__restrict__ tw,
                                                                            const size_t batch)
{
    constexpr int BLOCK = CFG::BLOCK, NBUF = CFG::NBUF;
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    constexpr int XPB = BLOCK / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    constexpr bool WAVE = (TPT <= 64);          // a transform lives inside one wavefront: wave-synchronous exchanges
    constexpr int G = TPT >= 64 ? 1 : 64 / TPT;  // transforms per wavefront
    static_assert(TPT >= 16 && (TPT >= 64 ? TPT % 64 == 0 : 64 % TPT == 0) && BLOCK % 64 == 0 && BLOCK % TPT == 0, "geometry");
    static_assert(NP >= 2 && NP <= 4 && (NP < 4 || NBUF == 1), "persistent kernel is built for 2 to 4 register passes");
    static_assert(NBUF == 1 || (NBUF == 2 && !WAVE), "NBUF");
    using FirstG = PassGeom<L, RL, 0>;
    using RawSel = persist_raw<IO, TPT>;
    using Raw = typename RawSel::type;
    constexpr int RS = R + RawSel::extra;  // registers of one prefetch set

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int tau = tid % TPT;
    // Transforms of at least a wavefront: the slot is wave-uniform (SGPR).  Smaller ones: a wavefront carries G
    // consecutive transforms (rows adjacent in memory), addressed through ONE descriptor over the group plus a per-lane
    // row offset; `wslot` is the group's first slot (uniform), `sub` this lane's transform inside the group.
    const int sub = (G == 1) ? 0 : (tid & 63) / TPT;
    const int wslot = (G == 1) ? __builtin_amdgcn_readfirstlane(tid / TPT) : __builtin_amdgcn_readfirstlane(tid >> 6) * G;
    const int slot = wslot + sub;
    cpx<T> *buf0 = reinterpret_cast<cpx<T> *>(smem_raw) + (size_t)slot * NBUF * persist_slot_elems(L);
    cpx<T> *buf1 = (NBUF == 2) ? buf0 + persist_slot_elems(L) : buf0;

    // ---- per-thread invariants, fetched once
    PersistState<T, L, RL, EPI, IO, CFG> st;
    using St = PersistState<T, L, RL, EPI, IO, CFG>;
    if constexpr (St::TW1_REG) persist_load_tw<T, L, RL, 1>(st.tw1, tau, tw);
    if constexpr (St::TW2_REG) persist_load_tw<T, L, RL, 2>(st.tw2, tau, tw);
    if constexpr (St::TW3_REG) persist_load_tw<T, L, RL, 3>(st.tw3, tau, tw);
    {
        // transform-independent operands: registers, or (kInvInLds / rfft table) one LDS copy per workgroup
        char *extra = smem_raw + (size_t)XPB * NBUF * persist_slot_elems(L) * sizeof(cpx<T>);
        typename IO::Inv *inv_lds = reinterpret_cast<typename IO::Inv *>(extra);
        extra += CFG::kInvInLds ? N * sizeof(typename IO::Inv) : 0;
        cpx<T> *rt_lds = reinterpret_cast<cpx<T> *>(extra);
        extra += (EPI == EPI_RFFT) ? N * sizeof(cpx<T>) : 0;
        cpx<T> *tw_lds = reinterpret_cast<cpx<T> *>(extra);
        if constexpr (CFG::kInvInLds) {
            for (int i = tid; i < N; i += BLOCK) inv_lds[i] = io.invariant(i);
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) st.inv[u] = io.invariant(FirstG::in_index(0, u) + tau);
        }
        if constexpr (EPI == EPI_RFFT) {
            for (int i = tid; i < N; i += BLOCK) rt_lds[i] = io.rtab[i];
        }
        if constexpr (CFG::kTwLastInLds) {
            for (int i = tid; i < N / 2; i += BLOCK) tw_lds[i] = tw[i];
        }
        st.inv_lds = inv_lds;
        st.rt_lds = rt_lds;
        st.tw_lds = tw_lds;
        if constexpr (CFG::kInvInLds || CFG::kTwLastInLds || EPI == EPI_RFFT) __syncthreads();
    }
    st.g1 = lds_pad(PassGeom<L, RL, 1>::in_index(tau, 0));
    st.g2 = (NP >= 3) ? lds_pad(PassGeom<L, RL, (NP >= 3 ? 2 : 1)>::in_index(tau, 0)) : 0;
    st.g3 = (NP == 4) ? lds_pad(PassGeom<L, RL, NP - 1>::in_index(tau, 0)) : 0;
    st.sc = lds_pad(tau);

    const size_t step = (size_t)gridDim.x * XPB;
    size_t base = (size_t)blockIdx.x * XPB;
    if (base >= batch) return;  // the whole workgroup leaves together

    // Raw register sets swap roles every transform (no register copies): while the transform held in one set is
    // computed, the others receive the loads of the transforms that follow.
    constexpr int DEPTH = persist_depth<CFG>::value;
    static_assert(DEPTH == 1 || DEPTH == 2, "prefetch distance");
    Raw ra[RS], rb[RS];
    const int in_lane_bytes = tau * IO::kRawBytes;
    const int in_row_off = (G == 1) ? 0 : sub * (int)io.in_slot_bytes();
    // valid transforms of this wavefront's group when the workgroup sits at `b`
    auto group_cnt = [&](size_t b) -> int {
        const size_t first = b + wslot;
        return first >= batch ? 0 : (batch - first < (size_t)G ? (int)(batch - first) : G);
    };
    // Issue the loads of the group at `b` (unconditionally, through a descriptor that is EMPTY when there is no such
    // transform -- the bounds check then returns zeros without touching memory; no branch, so the loads carry no
    // register shuffles behind them).
    auto issue = [&](Raw *dst, const size_t b) {
        const rsrc_t d = io.in_desc_n(b + wslot, group_cnt(b));
        if constexpr (RawSel::pair || RawSel::pair_lds) {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = io.fetch_pair_d(d, in_lane_bytes, FirstG::in_index(0, u), in_row_off);
            dst[R] = io.fetch_last_d(d, in_row_off);
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = io.fetch_d(d, in_lane_bytes, FirstG::in_index(0, u), in_row_off);
        }
    };
    issue(ra, base);
    typename persist_acc<IO>::type acc{};
    if constexpr (io_has_acc<IO>::value) acc = io.acc_init();
    auto finish = [&]() {
        if constexpr (io_has_acc<IO>::value) io.acc_finish(acc);
    };

    if constexpr (DEPTH == 1 && St::TWG && !(RawSel::pair || RawSel::pair_lds)) {
        // kTwGlobal: the next transform's loads in NP - 1 parts behind the passes' table loads (persist_transform)
        constexpr bool TOP = persist_twg_top_part(L);
        constexpr int PARTS = TOP ? NP : NP - 1;  // (part 0 at the top of the step,) part p behind pass p's entries
#define KOFFT_PERSIST_STEP_PARTS(CUR, NXT, LEAVE)                                                                    \
    {                                                                                                                \
        const size_t nbase = base + step;                                                                            \
        const bool more = nbase < batch; /* workgroup-uniform */                                                     \
        const rsrc_t nd = io.in_desc_n(nbase + wslot, group_cnt(nbase));                                             \
        auto part = [&](const int i) {                                                                               \
            _Pragma("unroll") for (int u = 0; u < R; ++u)                                                            \
                if (u * PARTS / R == i) NXT[u] = io.fetch_d(nd, in_lane_bytes, FirstG::in_index(0, u), in_row_off);  \
        };                                                                                                           \
        if constexpr (TOP) {                                                                                         \
            part(0);                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        persist_transform<T, L, RL, EPI, CFG>(CUR, st, io, tw, buf0, buf1, base + wslot, group_cnt(base), sub, tau, acc, part); \
        if (!more) LEAVE;                                                                                            \
        base = nbase;                                                                                                \
    }
        KOFFT_PERSIST_STEP_PARTS(ra, rb, { finish(); return; })
        for (;;) {
            KOFFT_PERSIST_STEP_PARTS(rb, ra, break)
            KOFFT_PERSIST_STEP_PARTS(ra, rb, break)
        }
        finish();
#undef KOFFT_PERSIST_STEP_PARTS
    } else if constexpr (DEPTH == 1) {
        // One step: issue the NEXT transform's loads into NXT, then run the transform held in CUR.
#define KOFFT_PERSIST_STEP(CUR, NXT, LEAVE)                                                                          \
    {                                                                                                                \
        const size_t nbase = base + step;                                                                            \
        const bool more = nbase < batch; /* workgroup-uniform */                                                     \
        issue(NXT, nbase);                                                                                           \
        __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ahead of CUR's first use */                          \
        persist_transform<T, L, RL, EPI, CFG>(CUR, st, io, tw, buf0, buf1, base + wslot, group_cnt(base), sub, tau, acc); \
        if (!more) LEAVE;                                                                                            \
        base = nbase;                                                                                                \
    }
        // The first step is peeled off the loop.  s_waitcnt vmcnt counts loads AND stores in issue order, so "CUR has
        // landed" is vmcnt(32 - i): 16 stores of the previous transform and 16 loads of the next one may stay in flight.
        // With the first step inside the loop the compiler must merge the entry state (no stores yet) with the back-edge
        // state and emits vmcnt(15 - i) for both -- which, on the back edge, waits for the previous transform's STORES to
        // complete, once per two transforms.  After the peel both predecessors of the loop header look the same.
        KOFFT_PERSIST_STEP(ra, rb, { finish(); return; })
        for (;;) {
            KOFFT_PERSIST_STEP(rb, ra, break)
            KOFFT_PERSIST_STEP(ra, rb, break)
        }
        finish();
#undef KOFFT_PERSIST_STEP
    } else {
        // (three ahead -- four sets, 226 registers -- measured in round 4 on config 3: 3.305 -> 3.315 ms; the kernel does not wait on latency any more)
        // Two transforms ahead (three sets): for configurations that run ONE wavefront per SIMD, where a transform's
        // own work (~3 us) is not enough time for its successor's loads to land under load.  While transform t is
        // computed, t+1 has been in flight for a whole transform and t+2 is issued.
        Raw rc[RS];
        issue(rb, base + step);
#define KOFFT_PERSIST_STEP2(CUR, FAR, LEAVE)                                                                         \
    {                                                                                                                \
        const size_t nbase = base + step;                                                                            \
        const bool more = nbase < batch; /* workgroup-uniform */                                                     \
        issue(FAR, nbase + step);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        persist_transform<T, L, RL, EPI, CFG>(CUR, st, io, tw, buf0, buf1, base + wslot, group_cnt(base), sub, tau, acc); \
        if (!more) LEAVE;                                                                                            \
        base = nbase;                                                                                                \
    }
        KOFFT_PERSIST_STEP2(ra, rc, { finish(); return; })
        for (;;) {
            KOFFT_PERSIST_STEP2(rb, ra, break)
            KOFFT_PERSIST_STEP2(rc, rb, break)
            KOFFT_PERSIST_STEP2(ra, rc, break)
        }
        finish();
#undef KOFFT_PERSIST_STEP2
    }
}


}  // namespace kofft
