// fft_persist.hip.h -- persistent, software-pipelined variant of the workgroup Stockham kernel.
//
// Same arithmetic and pass structure as fft_wg_kernel (fft_wg.hip.h), re-organised around what the
// MI355X memory system needs to stream at its copy ceiling:
//   * the grid is sized to the chip (a few workgroups per CU) and every workgroup loops over
//     transforms  xf = blockIdx.x, blockIdx.x + gridDim.x, ...;
//   * the NEXT transform's global loads are issued into a second register set before the current
//     transform is computed, so every resident workgroup always has a full transform (32 KiB at
//     n = 4096) of HBM reads in flight while it works on butterflies and LDS exchanges;
//   * the twiddles of every pass are read from the reference-recipe table ONCE per workgroup into
//     registers (they depend on the thread, not on the transform), which removes the per-transform
//     table traffic (30 KiB per 32 KiB transform at n = 4096) and its L2 latency from the loop.
// Results are bit-identical to fft_wg_kernel: the same butterflies consume the same table entries.
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
struct PassGeom {
    static constexpr int N = 1 << L;
    static constexpr int R = 1 << RL;
    static constexpr int TPT = N / R;
    static constexpr int NP = (L + RL - 1) / RL;
    static constexpr int S0 = P * RL;
    static constexpr int Q = (P == NP - 1) ? (L - RL * (NP - 1)) : RL;
    static constexpr int G = R >> Q;
    static constexpr int JB = L - S0 - Q;
    static constexpr int TWN = G * ((1 << Q) - 1);  // twiddles this thread needs for the pass
};

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
    for (int g = 0; g < Gm::G; ++g) reg_pass<T, L, 0, Gm::Q>(v + g * (1 << Gm::Q), 0, tw);
}

template <typename T, int L, int RL, int P>
__device__ __forceinline__ void persist_lds_gather(cpx<T> *v, const cpx<T> *buf, const int tau)
{
    using Gm = PassGeom<L, RL, P>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g) {
        const int m = tau + g * Gm::TPT;
        const int k = m >> Gm::JB;
        const int j = m & ((1 << Gm::JB) - 1);
#pragma unroll
        for (int c = 0; c < (1 << Gm::Q); ++c) v[g * (1 << Gm::Q) + c] = buf[lds_pad((k << (L - Gm::S0)) | (c << Gm::JB) | j)];
    }
}

template <typename T, int L, int RL, int P>
__device__ __forceinline__ void persist_lds_scatter(const cpx<T> *v, cpx<T> *buf, const int tau)
{
    using Gm = PassGeom<L, RL, P>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g) {
        const int m = tau + g * Gm::TPT;
#pragma unroll
        for (int c = 0; c < (1 << Gm::Q); ++c) buf[lds_pad((bitrev(c, Gm::Q) << (L - Gm::Q)) | m)] = v[g * (1 << Gm::Q) + c];
    }
}

template <typename T, int L, int RL, class IO>
__device__ __forceinline__ void persist_global_gather(cpx<T> *v, const IO &io, const size_t xf, const int tau)
{
    using Gm = PassGeom<L, RL, 0>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g) {
        const int m = tau + g * Gm::TPT;
        const int k = m >> Gm::JB;
        const int j = m & ((1 << Gm::JB) - 1);
#pragma unroll
        for (int c = 0; c < (1 << Gm::Q); ++c) v[g * (1 << Gm::Q) + c] = io.load(xf, (k << L) | (c << Gm::JB) | j);
    }
}

template <typename T, int L, int RL, class IO>
__device__ __forceinline__ void persist_global_scatter(const cpx<T> *v, const IO &io, const size_t xf, const int tau)
{
    using Gm = PassGeom<L, RL, (L + RL - 1) / RL - 1>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g) {
        const int m = tau + g * Gm::TPT;
#pragma unroll
        for (int c = 0; c < (1 << Gm::Q); ++c) io.store(xf, (bitrev(c, Gm::Q) << (L - Gm::Q)) | m, v[g * (1 << Gm::Q) + c]);
    }
}

// One transform per workgroup at a time (TPT == BLOCK), 2 or 3 register passes.
// NBUF = 1: one LDS exchange buffer, 4 barriers per 3-pass transform.
// NBUF = 2: exchanges alternate between two buffers, 2 barriers per 3-pass transform.
template <typename T, int L, int RL, int NBUF, int MINW, class IO>
__global__ __launch_bounds__((1 << L) >> RL, MINW) void fft_persist_kernel(const IO io, const cpx<T> *__restrict__ tw,
                                                                           const size_t batch)
{
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP == 2 || NP == 3, "persistent kernel is built for 2 or 3 register passes");
    static_assert(NBUF == 1 || NBUF == 2, "NBUF");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<T> *buf0 = reinterpret_cast<cpx<T> *>(smem_raw);
    cpx<T> *buf1 = (NBUF == 2) ? buf0 + lds_elems(N) : buf0;

    const int tau = threadIdx.x;

    // per-thread twiddles of passes 1.., fetched once.  Pass 0 has k == 0: its table indices are
    // compile-time constants, so it reads the table through scalar loads (SGPRs), not VGPRs.
    cpx<T> tw1[R - 1], tw2[R - 1];
    persist_load_tw<T, L, RL, 1>(tw1, tau, tw);
    if constexpr (NP == 3) persist_load_tw<T, L, RL, 2>(tw2, tau, tw);

    size_t xf = blockIdx.x;
    if (xf >= batch) return;  // whole workgroup leaves together: no barrier is skipped by part of it

    cpx<T> cur[R], nxt[R];
    persist_global_gather<T, L, RL>(cur, io, xf, tau);

    for (;;) {
        const size_t nxf = xf + gridDim.x;
        const bool more = nxf < batch;  // workgroup-uniform
        if (more) persist_global_gather<T, L, RL>(nxt, io, nxf, tau);  // prefetch: stays in flight below

        persist_compute_p0<T, L, RL>(cur, tw);
        if (NBUF == 1) __syncthreads();  // previous transform's last LDS gathers are done
        persist_lds_scatter<T, L, RL, 0>(cur, buf0, tau);
        __syncthreads();
        persist_lds_gather<T, L, RL, 1>(cur, buf0, tau);
        persist_compute<T, L, RL, 1>(cur, tw1);
        if constexpr (NP == 3) {
            if (NBUF == 1) __syncthreads();
            persist_lds_scatter<T, L, RL, 1>(cur, buf1, tau);
            __syncthreads();
            persist_lds_gather<T, L, RL, 2>(cur, buf1, tau);
            persist_compute<T, L, RL, 2>(cur, tw2);
        }
        persist_global_scatter<T, L, RL>(cur, io, xf, tau);

        if (!more) break;
#pragma unroll
        for (int i = 0; i < R; ++i) cur[i] = nxt[i];
        xf = nxf;
    }
}

}  // namespace kofft
