// fft_nd_fused.hip.h -- ndfft::fft2d_inplace (ndfft.rs:74-101) in TWO passes over the image instead of three (round 5).
//
// The reference transforms every row (ndfft.rs:89-91), then every column through a gather / fft / scatter with the column's own
// table T_rows (ndfft.rs:92-98).  Rounds 1-4: rows (one pass), then the column axis as two column-tile passes (stages 0 .. 6 and
// the rest, fft_big.hip.h AxisLastIO) -- three passes, traffic 3.0x the image.
//
// A column's first TWO Stockham stages only combine rows j, j + R/4, j + R/2, j + 3R/4 (R rows; in index bits [b0][b1][j]: stage 0
// maps [b0][b1 j] -> [b0'][b1 j] with w = T_R[0], stage 1 maps [k = b0'][b1][j] -> [b1'][k][j] with w = T_R[k * R/4]; fft.rs:836-898)
// and they do so element by element along the row.  So a workgroup that row-transforms those four rows back to back -- the
// persistent kernel's passes, 16 points per thread, rows of 1024 / 2048 / 4096 points -- holds, per thread, the same 16 columns of all four rows in
// registers and can run both column stages on them before anything is stored: the four rows go out to the intermediate at the
// positions stage 1 writes them ([b1'][k][j]: row j + (k + 2 b1') R/4), and the remaining log2(R) - 2 column stages are ONE
// column-tile pass with frequency prefix K = 2 bits (AxisLastIO with S = 2: tiles of 16 adjacent columns, 128-byte runs, the
// axis's own table T_R through TwSub).  Rows + 2 stages: image -> intermediate; the rest: intermediate -> image.
//
// Same butterflies on the same operands with the same table entries as the reference's row transforms followed by its column
// transforms; only the order of independent butterflies differs: bit-identical (tests/test_gpu_parity.py::test_fft2d_*).
// Inverse: ifft's conj / scale of the row transform (fft.rs:1163-1172) on the row result, then the column transform's conj on
// the way in; the column's conj and 1/R on the way out are AxisLastIO<INVERSE>'s.
#pragma once

#include "fft_persist.hip.h"

namespace kofft {

template <int L, bool INVERSE>
__global__ __launch_bounds__((1 << L) / 16, 2) void fft2d_rows4_kernel(const cpx<float> *__restrict__ in, cpx<float> *__restrict__ mid,
                                                                      const cpx<float> *__restrict__ tw,      // T_cols (the rows)
                                                                      const cpx<float> *__restrict__ tw_col,  // T_R (the columns)
                                                                      const int LQ,                           // log2(R / 4)
                                                                      const size_t groups,                    // images * R / 4
                                                                      const float row_scale)                  // 1 / (cols as f32)
{
    using T = float;
    // one row per workgroup at a time: 256 / 128 / 64 threads for 4096 / 2048 / 1024-point rows, 16 points each, three register
    // passes (4 + 4 + 4 / 3 / 2 stages); a 1024-point row is one wavefront and its exchanges need no s_barrier
    constexpr int RL = 4;
    constexpr int N = 1 << L, R = 1 << RL, TPT = N / R;
    static_assert(L >= 10 && L <= 12, "rows of 1024 .. 4096 points");
    constexpr bool WAVE = TPT <= 64;
    using G0 = WgGeom<L, RL, 0>;
    using GL = WgGeom<L, RL, 2>;
    constexpr int ES = (int)sizeof(cpx<T>);

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem_raw);
    const int tau = threadIdx.x;

    // column stages 0 and 1: T_R[0] and T_R[R / 4] (wave-uniform: scalar loads)
    const cpx<T> wc0 = tw_col[0], wc1 = tw_col[(size_t)1 << LQ];

    cpx<T> tw1[R - 1], tw2[R - 1];
    persist_load_tw<T, L, RL, 1>(tw1, tau, tw);
    persist_load_tw<T, L, RL, 2>(tw2, tau, tw);
    const int sc = lds_pad(tau);
    const int g1 = lds_pad(WgGeom<L, RL, 1>::in_index(tau, 0));
    const int g2 = lds_pad(WgGeom<L, RL, 2>::in_index(tau, 0));
    const int lane_bytes = tau * ES;

    const size_t step = gridDim.x;
    size_t g = blockIdx.x;
    if (g >= groups) return;  // the whole workgroup leaves together
    const size_t jmask = ((size_t)1 << LQ) - 1;
    // row q of group gg: image (gg >> LQ), row (gg & jmask) + q * R/4
    auto row_elems = [&](const size_t gg, const int q) -> size_t { return ((((gg >> LQ) << (LQ + 2)) + (gg & jmask) + ((size_t)q << LQ)) << L); };
    // loads through a descriptor that is EMPTY when there is no such group (zeros, no memory access, no branch: fft_persist.hip.h)
    auto issue = [&](cpx<T> *dst, const size_t gg, const int q) {
        const bool valid = gg < groups;
        const rsrc_t d = make_rsrc(in + (valid ? row_elems(gg, q) : 0), valid ? (unsigned)N * ES : 0u);
#pragma unroll
        for (int u = 0; u < R; ++u) dst[u] = buf_load_cpx<T, AUX_NT>(d, lane_bytes, G0::in_index(0, u) * ES);
    };
    // one row: raw -> v = the row's transform at out_index(tau, u), ready for the column stages
    auto transform = [&](const cpx<T> *raw, cpx<T> *v) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            v[u] = raw[u];
            if (INVERSE) v[u].im = -v[u].im;  // ifft: conj on the way in (fft.rs:1163-1165)
        }
        persist_compute_p0<T, L, RL>(v, tw);
        exchange_sync<WAVE>();  // the previous row's last gathers are done
        persist_lds_scatter<T, L, RL, 0>(v, buf, sc);
        exchange_sync<WAVE>();
        persist_lds_gather<T, L, RL, 1>(v, buf, g1);
        persist_compute<T, L, RL, 1>(v, tw1);
        exchange_sync<WAVE>();
        persist_lds_scatter<T, L, RL, 1>(v, buf, sc);
        exchange_sync<WAVE>();
        persist_lds_gather<T, L, RL, 2>(v, buf, g2);
        persist_compute<T, L, RL, 2>(v, tw2);
        if (INVERSE) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const T im = -v[u].im;  // the row's ifft: conj, then * 1/cols (fft.rs:1168-1172) ...
                v[u] = mk<T>(v[u].re * row_scale, im * row_scale);
                v[u].im = -v[u].im;     // ... and the column's ifft: conj on the way in
            }
        }
    };

    cpx<T> ra[R], rb[R];
    cpx<T> o0[R], o1[R], o2[R], o3[R];
    issue(ra, g, 0);
    for (;;) {
        const size_t ng = g + step;
        issue(rb, g, 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the current row's first use
        transform(ra, o0);
        issue(ra, g, 2);
        __builtin_amdgcn_sched_barrier(0);
        transform(rb, o1);
        issue(rb, g, 3);
        __builtin_amdgcn_sched_barrier(0);
        transform(ra, o2);
        issue(ra, ng, 0);  // the next group's first row (an empty descriptor behind the last group)
        __builtin_amdgcn_sched_barrier(0);
        transform(rb, o3);
        // column stage 0 (k = 0, n2 = R/2): pairs (row b1, row 2 + b1), w = T_R[0]; in place
        // column stage 1 (k = b0', n2 = R/4): e = A[k][0], o = A[k][1], w = T_R[k R/4]; e' -> row k, o' -> row k + 2
#pragma unroll
        for (int u = 0; u < R; ++u) {
            bfly<T, true>(o0[u], o2[u], wc0);
            bfly<T, true>(o1[u], o3[u], wc0);
            bfly<T, true>(o0[u], o1[u], wc0);  // k = 0: rows 0 and 2
            bfly<T, true>(o2[u], o3[u], wc1);  // k = 1: rows 1 and 3
        }
        // (the intermediate is read back by the column pass: default-policy stores, it may stay in the Infinity Cache)
        {
            const rsrc_t d0 = make_rsrc(mid + row_elems(g, 0), (unsigned)N * ES), d1 = make_rsrc(mid + row_elems(g, 1), (unsigned)N * ES);
            const rsrc_t d2 = make_rsrc(mid + row_elems(g, 2), (unsigned)N * ES), d3 = make_rsrc(mid + row_elems(g, 3), (unsigned)N * ES);
#pragma unroll
            for (int u = 0; u < R; ++u) buf_store_cpx_aux<T, AUX_DEFAULT>(o0[u], d0, lane_bytes, GL::out_index(0, u) * ES);
#pragma unroll
            for (int u = 0; u < R; ++u) buf_store_cpx_aux<T, AUX_DEFAULT>(o2[u], d1, lane_bytes, GL::out_index(0, u) * ES);
#pragma unroll
            for (int u = 0; u < R; ++u) buf_store_cpx_aux<T, AUX_DEFAULT>(o1[u], d2, lane_bytes, GL::out_index(0, u) * ES);
#pragma unroll
            for (int u = 0; u < R; ++u) buf_store_cpx_aux<T, AUX_DEFAULT>(o3[u], d3, lane_bytes, GL::out_index(0, u) * ES);
        }
        if (ng >= groups) break;  // workgroup-uniform
        g = ng;
    }
}

}  // namespace kofft
