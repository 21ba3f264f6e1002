// fft_big_real.hip.h -- rfft_direct (rfft.rs:425-465) for inner lengths m beyond the single-workgroup kernels WITHOUT a pass of
// its own for the post-pass (round 5): the last factor of the m-point transform and rfft.rs:450-463 in one kernel.
//
// The composed route of rounds 2-4 ran three passes over HBM: first factor (window folded into its loads), last factor, then
// rfft_post_kernel reading Y and writing X.  The post-pass pairs bin k with m - k.  With m = 2^(LA + LB) and the last factor
// working on rows (frequency prefix K, LA bits; fft_big.hip.h), bin k = q * 2^LA + K lives in row K at position q, and
//     m - k = (2^LB - 1 - q) * 2^LA + (2^LA - K)         (K != 0)
// lives in row 2^LA - K at the MIRRORED position.  A tile of XPB adjacent rows S_p = [XPB p + 1, XPB p + XPB] -- shifted by one
// against the tile grid -- has exactly the aligned tile A_p = [2^LA - XPB p - XPB, 2^LA - XPB p - 1] as its mirror (slot s <->
// slot XPB - 1 - s); S_0 .. S_(P-1), P = 2^LA / (2 XPB), cover rows 1 .. 2^(LA-1) and the A tiles rows 2^(LA-1) .. 2^LA - 1.  Row
// H = 2^(LA-1) would be computed twice (it is its own mirror) and row 0 (its own mirror too, at positions q <-> 2^LB - q) not at
// all: in the last pair S's last slot takes row 0 instead of H.
//
// A workgroup owns ONE tile pair and walks the transforms of the launch (as fft_rows_persist_kernel does with one tile: both
// tiles' table entries stay resident -- the staged passes in LDS, the last pass's in registers).  Per transform: tile S through
// the passes (its 2^RL values per thread stay in registers), tile A through the passes, Y_A into the exchange buffer in natural
// order, and every thread then holds a = Y[k] of tile S and reads c = Y[m - k] of tile A from the mirrored cell: it computes and
// stores BOTH X[k] = post(a, c, W[k]) and X[m - k] = post(c, a, W[m - k]) (rfft.rs:454-463; X[0], X[m]: rfft.rs:450-452).  The
// stores have the plain last factor's shape: runs of XPB adjacent rows per position q, ascending for S, descending for A.
// The self-mirrored rows 0 and H of the last pair pair inside themselves (row 0's values through a row buffer in LDS).
//
// Same butterflies, same table entries (T_m through TwSub, W = build_twiddle_table(m)) and the same expressions per output as
// the three-pass route: bit-identical (tests/test_gpu_parity.py::test_rfft_big_fused_*, every row against the oracle).
#pragma once

#include "fft_big.hip.h"

namespace kofft {

template <int L, int RL, int BLOCK, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void fft_rows_rfft_kernel(const cpx<float> *__restrict__ mid, cpx<float> *__restrict__ out,
                                                                                      const cpx<float> *__restrict__ tw,    // T_m
                                                                                      const cpx<float> *__restrict__ rtab,  // W[k], k < m (rfft.rs:172-183)
                                                                                      const int LA, const unsigned nb, const unsigned pairs,
                                                                                      const int nt_load)
{
    using T = float;
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    static_assert(TPT >= 1 && BLOCK % TPT == 0, "bad geometry");
    constexpr int XPB = BLOCK / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP >= 2 && NP <= 4, "pass count");
    using G0 = WgGeom<L, RL, 0>;
    using GL = WgGeom<L, RL, NP - 1>;
    constexpr int QL = GL::Q, GRP = GL::G;
    constexpr int ES = (int)sizeof(cpx<T>);
    constexpr int FULL = R - 1;
    constexpr int TWE = rows_tw_entries<L, RL>() * XPB;  // LDS cells of one tile's staged tables
    using Exch = TileExchange<T, L, RL, XPB, false>;
    static_assert(Exch::bytes >= (size_t)N * XPB * sizeof(cpx<T>), "Y_A in natural order fits the exchange region");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<T> *ynat = reinterpret_cast<cpx<T> *>(smem_raw);  // (the exchange region, between transforms of a tile)
    cpx<T> *tw_lds_s = reinterpret_cast<cpx<T> *>(smem_raw + Exch::bytes);
    cpx<T> *tw_lds_a = tw_lds_s + TWE;
    cpx<T> *row0 = tw_lds_a + TWE;  // [N]: row 0's Y in natural order (last pair only)
    const int tid = threadIdx.x;
    const int tau = tid / XPB;
    const int slot = tid % XPB;

    const unsigned G = gridDim.x, w = blockIdx.x;
    const unsigned groups = G / pairs;
    if (groups == 0 || w >= pairs * groups) return;
    const unsigned p = w % pairs;
    unsigned b = w / pairs;
    if (b >= nb) return;
    const unsigned rows = 1u << LA;
    const bool special = p == pairs - 1;                 // workgroup-uniform
    const bool self_rows = special && slot == XPB - 1;   // this thread's S row is row 0; it also produces row H's outputs
    const unsigned r_s = self_rows ? 0u : XPB * p + 1u + (unsigned)slot;
    const unsigned r_a = rows - XPB * p - XPB + (unsigned)slot;
    const size_t m = (size_t)N << LA;
    const unsigned xf_bytes = (unsigned)(m * sizeof(cpx<T>));

    auto issue_loads = [&](cpx<T> *dst, const unsigned row, const unsigned tb, const bool valid) {
        const rsrc_t d = make_rsrc(mid + (size_t)(valid ? tb : 0) * m, valid ? xf_bytes : 0u);
        const int lane = (int)((row * (unsigned)N + (unsigned)tau) * (unsigned)ES);
        if (nt_load) {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = buf_load_cpx<T, AUX_NT>(d, lane, G0::in_index(0, u) * ES);
        } else {
#pragma unroll
            for (int u = 0; u < R; ++u) dst[u] = buf_load_cpx<T, AUX_DEFAULT>(d, lane, G0::in_index(0, u) * ES);
        }
    };

    // ---- table entries of both tiles, resident for the whole launch (fft_rows_persist_kernel::load_tables, per tile)
    auto stage_group = [&](auto s0, cpx<T> *cells, const int k, const TwSub map) {
        constexpr int S0 = decltype(s0)::value;
#pragma unroll
        for (int t = 0; t < RL; ++t) {
#pragma unroll
            for (int h = 0; h < (1 << t); ++h) {
                const int idx = (k << (L - 1 - S0 - t)) + (bitrev(h, t) << (L - 1 - t));
                cells[((1 << t) - 1 + h) * XPB] = tw[map(idx, S0 + t)];
            }
        }
    };
    cpx<T> twl_s[GRP * ((1 << QL) - 1)], twl_a[GRP * ((1 << QL) - 1)];
    int lds_base[NP > 1 ? NP - 1 : 1];  // (the same cell offsets in both tiles' regions)
    auto load_tables = [&](const unsigned row, cpx<T> *region, cpx<T> *twl) {
        const TwSub map = TwSub{LA, (int)row, L - 1};  // shift = log2 m - L, prefix K = row, kbase = log2 m - 1 - LA
        int off = 0;
#define KOFFT_ROWS_STAGE(P)                                                                                                    \
        if constexpr (P + 1 < NP) {                                                                                            \
            using Gm = WgGeom<L, RL, P>;                                                                                       \
            static_assert(Gm::G == 1 && Gm::Q == RL, "staged passes are full passes");                                         \
            const int k = tau >> Gm::JB;                                                                                       \
            lds_base[P] = (off + k * FULL) * XPB + slot;                                                                       \
            if ((tau & ((1 << Gm::JB) - 1)) == 0) stage_group(std::integral_constant<int, Gm::S0>{}, region + lds_base[P], k, map); \
            off += (1 << Gm::S0) * FULL;                                                                                       \
        }
        KOFFT_ROWS_STAGE(0)
        KOFFT_ROWS_STAGE(1)
        KOFFT_ROWS_STAGE(2)
#undef KOFFT_ROWS_STAGE
#pragma unroll
        for (int g = 0; g < GRP; ++g)
            load_pass_twiddles_map<T, L, GL::S0, QL>(twl + g * ((1 << QL) - 1), (tau + g * TPT) >> GL::JB, tw, map);
    };
    load_tables(r_s, tw_lds_s, twl_s);
    load_tables(r_a, tw_lds_a, twl_a);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the table loads have landed before the loop is entered (see fft_rows_persist_kernel)
    __syncthreads();

    // one tile through the passes, in place in its register set
    auto run_passes = [&](cpx<T> *cur, const cpx<T> *region, const cpx<T> *twl) {
        reg_pass_lds<T, RL, XPB>(cur, region + lds_base[0]);
        if constexpr (NP > 2) {
            Exch::template run<0>(cur, smem_raw, tau, slot);
            reg_pass_lds<T, RL, XPB>(cur, region + lds_base[NP > 2 ? 1 : 0]);
        }
        if constexpr (NP > 3) {
            Exch::template run<1>(cur, smem_raw, tau, slot);
            reg_pass_lds<T, RL, XPB>(cur, region + lds_base[NP > 3 ? 2 : 0]);
        }
        Exch::template run<NP - 2>(cur, smem_raw, tau, slot);
#pragma unroll
        for (int g = 0; g < GRP; ++g) reg_pass_r<T, QL>(cur + g * (1 << QL), twl + g * ((1 << QL) - 1));
    };
    auto post = [&](const cpx<T> wk, const cpx<T> a, const cpx<T> ymk) -> cpx<T> {
        const v2f wv = {wk.re, wk.im}, av = {a.re, a.im}, yv = {ymk.re, ymk.im};
        const v2f x = rfft_post_f32_pk(wv, av, yv);  // rfft.rs:454-463
        return mk<T>(x.x, x.y);
    };

    // W[k] and W[m - k] of this thread's R output pairs: they depend on (q, row) only, not on the transform -- resident in registers
    // for the whole launch (read per transform they are as many bytes as the data, from L2 at best, and every s_waitcnt on them also
    // waits for whatever else the wavefront has in flight: 0.58 ms against 0.40 ms for the two kernels this one replaces).
    cpx<T> wk[R], wmk[R];
    {
        const rsrc_t dw = make_rsrc(rtab, xf_bytes);
        const unsigned H = rows >> 1;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const unsigned q = (unsigned)(GL::out_index(0, u) + tau);
            // (the self-mirrored rows of the last pair: row 0's k = q 2^LA and row H's k = q 2^LA + H)
            const unsigned k = self_rows ? (q << LA) : (q << LA) + r_s;
            const unsigned k2 = self_rows ? (q << LA) + H : (((unsigned)N - 1u - q) << LA) + (rows - r_s);
            wk[u] = buf_load_cpx<T, AUX_DEFAULT>(dw, (int)(k * (unsigned)ES), 0);
            wmk[u] = buf_load_cpx<T, AUX_DEFAULT>(dw, (int)(k2 * (unsigned)ES), 0);
        }
    }
    cpx<T> ra[R], rb[R], ys[R];
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (the table loads have landed: see above)
    issue_loads(ra, r_s, b, true);
    issue_loads(rb, r_a, b, true);
    for (;;) {
        const unsigned nbb = b + groups;
        const bool more = nbb < nb;  // workgroup-uniform
        run_passes(ra, tw_lds_s, twl_s);
#pragma unroll
        for (int u = 0; u < R; ++u) ys[u] = ra[u];
        issue_loads(ra, r_s, nbb, more);  // the next transform's tile S: in flight during tile A and the epilogue
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                  // tile S's last gathers are done before tile A's first scatter
        run_passes(rb, tw_lds_a, twl_a);
        __syncthreads();                  // ... and tile A's, before the exchange region is overwritten with Y_A
        // Y_A in natural order: cell [q][slot]; row 0's Y beside it in the last pair
#pragma unroll
        for (int u = 0; u < R; ++u) ynat[(GL::out_index(0, u) + tau) * XPB + slot] = rb[u];
        if (self_rows) {
#pragma unroll
            for (int u = 0; u < R; ++u) row0[GL::out_index(0, u) + tau] = ys[u];
        }
        __syncthreads();
        const rsrc_t d = make_rsrc(out + (size_t)b * (m + 1), (unsigned)((m + 1) * sizeof(cpx<T>)));
        // Opaque copies of the thread's coordinates, taken INSIDE the loop: every address of the epilogue (2 R table offsets, 2 R store
        // offsets, R LDS cells) depends on the thread only, so the compiler hoists all of them out of the transform loop and keeps them
        // in registers for the whole launch -- 120 registers spilled.  Recomputed per transform they are a few VALU operations each.
        int tau_e = tau, slot_e = slot;
        unsigned rs_e = r_s;
        asm volatile("" : "+v"(tau_e), "+v"(slot_e), "+v"(rs_e));
        if (!self_rows) {
            // k = q 2^LA + r_s (tile S, this thread's registers), m - k = (N - 1 - q) 2^LA + (2^LA - r_s) (tile A, slot XPB - 1 - slot)
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const unsigned q = (unsigned)(GL::out_index(0, u) + tau_e);
                const unsigned k = (q << LA) + rs_e, mk_ = (((unsigned)N - 1u - q) << LA) + (rows - rs_e);
                const cpx<T> a = ys[u], c = ynat[(N - 1 - (int)q) * XPB + (XPB - 1 - slot_e)];
                buf_store_cpx_aux<T, AUX_NT>(post(wk[u], a, c), d, (int)(k * (unsigned)ES), 0);
                buf_store_cpx_aux<T, AUX_NT>(post(wmk[u], c, a), d, (int)(mk_ * (unsigned)ES), 0);
            }
        } else {
            // row 0: k = q 2^LA pairs with (N - q) 2^LA inside the row (q = 0: X[0] and X[m], rfft.rs:450-452);
            // row H: k = q 2^LA + H pairs with (N - 1 - q) 2^LA + H inside tile A's slot 0
            const unsigned H = rows >> 1;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const unsigned q = (unsigned)(GL::out_index(0, u) + tau_e);
                const cpx<T> a0 = ys[u];
                const unsigned k = q << LA;
                const cpx<T> x0 = post(wk[u], a0, row0[(N - (int)q) & (N - 1)]);  // (q = 0: replaced)
                buf_store_cpx_aux<T, AUX_NT>(q == 0 ? mk<T>(a0.re + a0.im, T(0)) : x0, d, (int)(k * (unsigned)ES), 0);
                if (q == 0) buf_store_cpx_aux<T, AUX_NT>(mk<T>(a0.re - a0.im, T(0)), d, (int)((unsigned)m * (unsigned)ES), 0);  // X[m]
                const cpx<T> ah = ynat[(int)q * XPB], ch = ynat[(N - 1 - (int)q) * XPB];
                buf_store_cpx_aux<T, AUX_NT>(post(wmk[u], ah, ch), d, (int)((k + H) * (unsigned)ES), 0);
            }
        }
        if (!more) break;
        issue_loads(rb, r_a, nbb, true);  // the next transform's tile A: in flight during its tile S
        b = nbb;
        __syncthreads();  // the epilogue's LDS reads are done before the next tile's first scatter
    }
}

}  // namespace kofft
