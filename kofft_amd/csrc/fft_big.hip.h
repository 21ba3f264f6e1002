// fft_big.hip.h -- power-of-two transforms too large for one workgroup's LDS (n > 16384 f32 / 8192 f64),
// e.g. BASELINE config #5: 2^20-point Complex64.  Two factors up to 2^21, three (BigMidIO below) from 2^22.
//
// The reference runs L = log2(n) radix-2 Stockham stages over the whole array (fft.rs:834-898).  In index bits,
// stage s maps [k : s bits][b][j] -> [b'][k][j], so the stages split cleanly into two factors, L = LA + LB:
//
//   factor A = stages 0 .. LA-1.  No frequency prefix yet (k empty); the LB low bits j are untouched.
//       For each j it is a complete 2^LA-point Stockham transform of the column  x_c = in[c * 2^LB + j]
//       whose result q (natural order) lands at  q * 2^LB + j.
//   factor B = stages LA .. L-1.  The prefix K (LA bits) is fixed per row; no j bits remain.
//       For each K it is a 2^LB-point Stockham transform of the contiguous row  x_c = mid[K * 2^LB + c]
//       whose result q lands at  q * 2^LA + K   -- natural order of the full transform.
//
// Every butterfly is the reference's butterfly with the reference's table entry: global stage s, group kk uses
// T_n[kk * 2^(L-1-s)].  For a sub-transform that handles global stages S_off + s_local with local group index
// kk_local, kk = K + 2^S_off * kk_local, hence
//   index = (K << (L-1-S_off-s_local)) + (idx_local << (L - L_sub)),   idx_local = kk_local << (L_sub-1-s_local)
// (TwSub in fft_device.hip.h).  T_n != a subsampling of T_{2^L_sub} bitwise (each table is its own recurrence),
// so both factors index the ONE table T_n.  Only the order of independent butterflies changes: results are
// bit-identical to the reference's 20 sweeps, with 2 passes over HBM instead of 20.
//
// The two factors run as the generic workgroup kernel (fft_wg.hip.h) with these IO policies; the intermediate
// lives in a scratch buffer of `chunk` transforms so that factor B reads it back from the Infinity Cache.
#pragma once

#include "fft_wg.hip.h"

namespace kofft {

// factor A: unit xf = (b, j) with j in [0, 2^LB): column j of transform b.  Adjacent units are adjacent columns,
// so lanes run over units first (kSlotMinor) and a wave touches whole 128-byte lines.
template <typename T, bool INVERSE>
struct BigColsIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;
    static constexpr bool kPairXcd = true;
    static constexpr bool kSplitLds = sizeof(T) == 8;  // c64: 8-byte exchange elements (re / im in two rounds), 8-column tiles fit twice per CU
    static constexpr int kMinWaves = sizeof(T) == 8 ? 4 : 1;  // two 512-thread workgroups per CU need <= 128 VGPRs
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int LB;     // log2 of the column count
    int shift;  // L - LA
    size_t n;   // full transform length
    __device__ __forceinline__ TwSubFirst tw_map(size_t) const { return TwSubFirst{shift}; }
    __device__ __forceinline__ cpx<T> load(size_t xf, int c) const
    {
        const size_t b = xf >> LB, j = xf & ((size_t(1) << LB) - 1);
        cpx<T> v = ld_stream(in + b * n + ((size_t)c << LB) + j);  // read once
        if (INVERSE) v.im = -v.im;  // ifft: conj on the way in (fft.rs:1163-1165)
        return v;
    }
    __device__ __forceinline__ void store(size_t xf, int q, cpx<T> v) const
    {
        const size_t b = xf >> LB, j = xf & ((size_t(1) << LB) - 1);
        out[b * n + ((size_t)q << LB) + j] = v;  // read back by factor B: plain store (non-temporal measured no better)
    }
};

// factor B: unit xf = (b, K) with K in [0, 2^LA): row K of transform b (contiguous), output transposed.
template <typename T, bool INVERSE>
struct BigRowsIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;  // lanes run over adjacent rows K: 64-byte segments for loads and stores
    static constexpr bool kPairXcd = true;
    static constexpr bool kSplitLds = sizeof(T) == 8;
    static constexpr int kMinWaves = sizeof(T) == 8 ? 4 : 1;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int LA, LB;
    int shift;  // L - LB
    int kbase;  // L - 1 - LA
    size_t n;
    T scale;    // 1 / (n as f32 as T), fft.rs:1167
    bool nt;    // non-temporal stores: only when a workgroup's adjacent rows fill at least 64-byte segments
    __device__ __forceinline__ TwSub tw_map(size_t xf) const { return TwSub{shift, (int)(xf & ((size_t(1) << LA) - 1)), kbase}; }
    __device__ __forceinline__ cpx<T> load(size_t xf, int c) const
    {
        const size_t b = xf >> LA, K = xf & ((size_t(1) << LA) - 1);
        return ld_stream(in + b * n + (K << LB) + (size_t)c);  // the intermediate is read exactly once
    }
    __device__ __forceinline__ void store(size_t xf, int q, cpx<T> v) const
    {
        const size_t b = xf >> LA, K = xf & ((size_t(1) << LA) - 1);
        if (INVERSE) {  // conj, then scale (fft.rs:1168-1172)
            const T im = -v.im;
            v = mk<T>(v.re * scale, im * scale);
        }
        if (nt) st_stream(out + b * n + ((size_t)q << LA) + K, v);
        else out[b * n + ((size_t)q << LA) + K] = v;
    }
};

// Middle factor of a three-factor split (n >= 2^21): global stages S .. S+LS-1.  The index bits at that point read
// [K : S bits][c : LS bits][j : JB bits] and the stage group maps them to [q : LS][K : S][j : JB] (same algebra as
// above with both a frequency prefix K and untouched low bits j).  Unit xf = (b, K, j); adjacent units are adjacent j.
template <typename T>
struct BigMidIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;
    static constexpr bool kPairXcd = true;
    static constexpr bool kSplitLds = sizeof(T) == 8;
    static constexpr int kMinWaves = sizeof(T) == 8 ? 4 : 1;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int S, LS, JB;
    int shift;  // L - LS
    int kbase;  // L - 1 - S
    size_t n;
    __device__ __forceinline__ TwSub tw_map(size_t xf) const
    {
        return TwSub{shift, (int)((xf >> JB) & ((size_t(1) << S) - 1)), kbase};
    }
    __device__ __forceinline__ cpx<T> load(size_t xf, int c) const
    {
        const size_t b = xf >> (S + JB), K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        return ld_stream(in + b * n + (K << (LS + JB)) + ((size_t)c << JB) + j);
    }
    __device__ __forceinline__ void store(size_t xf, int q, cpx<T> v) const
    {
        const size_t b = xf >> (S + JB), K = (xf >> JB) & ((size_t(1) << S) - 1), j = xf & ((size_t(1) << JB) - 1);
        out[b * n + ((size_t)q << (S + JB)) + (K << JB) + j] = v;
    }
};

// ndfft (ndfft.rs:74-155): FftImpl::fft_strided over every line of one axis.  Unit xf is one line:
// element i lives at  (xf / inner) * outer_stride + (xf % inner) + i * stride.  Adjacent lines are adjacent in memory,
// so lanes run over lines first (kSlotMinor), as for the column factor above.  Each line is a stand-alone transform
// with its own table T_len (TwPlain), exactly what fft_strided's gather / fft / scatter computes.
template <typename T, bool INVERSE>
struct StridedIO {
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    static constexpr bool kSlotMinor = true;
    static constexpr bool kPairXcd = false;
    static constexpr bool kSplitLds = sizeof(T) == 8;
    static constexpr int kMinWaves = 1;
    cpx<T> *__restrict__ data;  // in place
    size_t inner, outer_stride, stride;
    T scale;  // 1 / (len as f32 as T)
    __device__ __forceinline__ TwPlain tw_map(size_t) const { return {}; }
    __device__ __forceinline__ size_t base(size_t xf) const { return (xf / inner) * outer_stride + (xf % inner); }
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const
    {
        cpx<T> v = data[base(xf) + (size_t)i * stride];
        if (INVERSE) v.im = -v.im;
        return v;
    }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        if (INVERSE) {
            const T im = -v.im;
            v = mk<T>(v.re * scale, im * scale);
        }
        data[base(xf) + (size_t)o * stride] = v;
    }
};

// ---- tiled transpose for the long strided axes of ndfft ------------------------------------------------------------
// dst[b][c][r] = src[b][r][c] for r < rows, c < cols: 32 x 32 tiles through LDS (row padded by one cell), both sides
// coalesced.  Leading dimensions and per-batch strides in elements.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const cpx<T> *__restrict__ src, cpx<T> *__restrict__ dst, const size_t rows,
                                                        const size_t cols, const size_t src_ld, const size_t dst_ld,
                                                        const size_t src_bs, const size_t dst_bs)
{
    __shared__ cpx<T> tile[32][33];
    const size_t b = blockIdx.z;
    const cpx<T> *s = src + b * src_bs;
    cpx<T> *d = dst + b * dst_bs;
    const size_t c0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const size_t r = r0 + ty + k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + k][tx] = ld_stream(s + r * src_ld + c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const size_t c = c0 + ty + k, r = r0 + tx;
        if (r < rows && c < cols) st_stream(d + c * dst_ld + r, tile[tx][ty + k]);
    }
}

// ---- Bluestein arm for non-power-of-two lengths (fft.rs:1088-1132) --------------------------------
// a = x * chirp (zero-padded to m = next_pow2(2n-1)); fft_m; a *= fft(b); conj; fft_m; conj; * 1/m; out = a * chirp.
// The three pointwise steps below use Complex::mul's un-fused form; the two m-point transforms are the ordinary
// power-of-two kernels.  INVERSE folds ifft's conj-in / conj-scale-out (fft.rs:1163-1172) around the same steps.
// Fused forms for m <= one workgroup's transform (n <= 8192 f32 / 4096 f64): the three pointwise steps ride on the two
// m-point transforms' loads and stores -- x is read once (n values), the padded intermediate is written and read once
// (m values), the result is written once (n values): 2n + 2m complex values of traffic instead of 2n + 8m.
// Same operations per element in the same order as the kernels below.
template <typename T, bool INVERSE>
struct BlueFirstIO : PlainTw {  // load: a = x * chirp, zero-padded to m; store: (a * fft(b)), conjugated (ifft's way in)
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ tmp;
    const cpx<T> *__restrict__ chirp;
    const cpx<T> *__restrict__ bfft;
    int n, m;
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const
    {
        const int ic = i < n ? i : n - 1;  // branch-free: clamp the address, select the value
        cpx<T> x = ld_stream(in + xf * (size_t)n + ic);
        if (INVERSE) x.im = -x.im;
        const cpx<T> v = cmul(x, chirp[ic]);
        return i < n ? v : mk<T>(T(0), T(0));
    }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        cpx<T> w = cmul(v, bfft[o]);
        w.im = -w.im;
        tmp[xf * (size_t)m + o] = w;  // read back by the second transform: plain store
    }
};
template <typename T, bool INVERSE>
struct BlueSecondIO : PlainTw {  // store: conj, * 1/m (ifft's way out), * chirp; only the first n values exist
    static constexpr bool kStreams = false;
    static constexpr bool kPersist = false;
    const cpx<T> *__restrict__ tmp;
    cpx<T> *__restrict__ out;
    const cpx<T> *__restrict__ chirp;
    int n, m;
    T scale_m, scale_n;
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const { return ld_stream(tmp + xf * (size_t)m + i); }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        if (o >= n) return;
        v.im = -v.im;
        v = mk<T>(v.re * scale_m, v.im * scale_m);
        cpx<T> r = cmul(v, chirp[o]);
        if (INVERSE) {
            const T im = -r.im;
            r = mk<T>(r.re * scale_n, im * scale_n);
        }
        st_stream(out + xf * (size_t)n + o, r);
    }
};

template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void bluestein_pre_kernel(const cpx<T> *__restrict__ in, cpx<T> *__restrict__ a,
                                                            const cpx<T> *__restrict__ chirp, const size_t n, const size_t m,
                                                            const size_t total /* batch * m */)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t b = idx / m, i = idx % m;
    cpx<T> v = mk<T>(T(0), T(0));
    if (i < n) {
        cpx<T> x = in[b * n + i];
        if (INVERSE) x.im = -x.im;
        v = cmul(x, chirp[i]);
    }
    a[idx] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void bluestein_mid_kernel(cpx<T> *__restrict__ a, const cpx<T> *__restrict__ bfft, const size_t m,
                                                            const size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    cpx<T> v = cmul(a[idx], bfft[idx % m]);
    v.im = -v.im;
    a[idx] = v;
}

template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void bluestein_post_kernel(const cpx<T> *__restrict__ a, cpx<T> *__restrict__ out,
                                                             const cpx<T> *__restrict__ chirp, const size_t n, const size_t m,
                                                             const size_t total /* batch * n */, const T scale_m, const T scale_n)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t b = idx / n, i = idx % n;
    cpx<T> v = a[b * m + i];
    v.im = -v.im;
    v = mk<T>(v.re * scale_m, v.im * scale_m);
    cpx<T> o = cmul(v, chirp[i]);
    if (INVERSE) {
        const T im = -o.im;
        o = mk<T>(o.re * scale_n, im * scale_n);
    }
    out[idx] = o;
}

}  // namespace kofft
