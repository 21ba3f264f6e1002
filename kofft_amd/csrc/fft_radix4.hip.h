// fft_radix4.hip.h -- ScalarFftImpl::fft_radix4 (fft.rs:1455-1548) byte for byte: the arm fft_with_strategy(.., Radix4)
// takes for powers of four (fft.rs:1356) -- and what the host mirrors' fft_with_strategy(.., Radix4) runs BY DEFAULT since
// round 6 (strict drop-in; INTEGRATION.md, "Behavioural differences").  From n = 16 that arm does not compute the DFT -- its
// "bit-reversal for radix-4" loop flips one bit per base-4 digit instead of reversing the digits -- and the reference's bytes
// are what a drop-in returns; `radix4_compat = false` / KOFFT_HIP_RADIX4_COMPAT=0 opts out (the true transform for every
// strategy).  Not a streaming kernel: one launch per radix-4 stage through global memory, the data-independent parts (the
// swap loop's net permutation, the three running-product twiddle sequences of every stage) built on the host with the
// reference's operations (tables.cpp: radix4) and uploaded once per (context, n).
// INV: FftPlan::ifft around this arm (fft.rs:2040-2055 with strategy Radix4): `c.im = -c.im` on the way in, and
// `c.im = -c.im; c.re * scale; c.im * scale` (scale = 1 / (n as f32 -> T)) on the way out of the last stage.
#pragma once

#include "fft_device.hip.h"

namespace kofft {

// butterfly4 (fft.rs:1596-1607): t3 goes through the general complex multiply by (0, -1)
template <typename T>
__device__ __forceinline__ void radix4_bfly(cpx<T> &a, cpx<T> &b, cpx<T> &c, cpx<T> &d)
{
    const cpx<T> t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d);
    const cpx<T> t3 = cmul(csub(b, d), mk<T>(T(0), -T(1)));
    a = cadd(t0, t2);
    b = cadd(t1, t3);
    c = csub(t0, t2);
    d = csub(t1, t3);
}

template <typename T>
__device__ __forceinline__ cpx<T> radix4_plan_out(const cpx<T> v, const T scale)  // fft.rs:2049-2053
{
    return mk<T>(v.re * scale, (-v.im) * scale);
}

// the swap loop (fft.rs:1462-1474) as a gather through its net permutation, fused with the len = 4 stage (fft.rs:1478-1487)
template <typename T, bool INV>
__global__ __launch_bounds__(256) void radix4_first_kernel(const cpx<T> *__restrict__ in, cpx<T> *__restrict__ out,
                                                           const unsigned *__restrict__ perm, const size_t n, const size_t quads,
                                                           const T scale)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= quads) return;
    const size_t per = n / 4, xf = t / per, q = t % per;
    const cpx<T> *src = in + xf * n;
    cpx<T> a = src[perm[4 * q]], b = src[perm[4 * q + 1]], c = src[perm[4 * q + 2]], d = src[perm[4 * q + 3]];
    if constexpr (INV) {  // fft.rs:2044-2046 (before the swap loop: conj commutes with the permutation)
        a.im = -a.im;
        b.im = -b.im;
        c.im = -c.im;
        d.im = -d.im;
    }
    radix4_bfly(a, b, c, d);
    if (INV && n == 4) {  // the only stage
        a = radix4_plan_out(a, scale);
        b = radix4_plan_out(b, scale);
        c = radix4_plan_out(c, scale);
        d = radix4_plan_out(d, scale);
    }
    cpx<T> *dst = out + xf * n + 4 * q;
    dst[0] = a;
    dst[1] = b;
    dst[2] = c;
    dst[3] = d;
}

// one stage len >= 16 (fft.rs:1488-1541): w = this stage's (w1, w2, w3)[j] triples
template <typename T, bool INV_LAST>
__global__ __launch_bounds__(256) void radix4_stage_kernel(const cpx<T> *__restrict__ src, cpx<T> *__restrict__ dst,
                                                           const cpx<T> *__restrict__ w, const size_t len, const size_t n,
                                                           const size_t quads, const T scale)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= quads) return;
    const size_t per = n / 4, xf = t / per, r = t % per, quarter = len / 4;
    const size_t i = (r / quarter) * len, j = r % quarter, base = xf * n + i + j;
    cpx<T> a = src[base];
    cpx<T> b = cmul(src[base + quarter], w[3 * j]);
    cpx<T> c = cmul(src[base + 2 * quarter], w[3 * j + 1]);
    cpx<T> d = cmul(src[base + 3 * quarter], w[3 * j + 2]);
    radix4_bfly(a, b, c, d);
    if constexpr (INV_LAST) {
        a = radix4_plan_out(a, scale);
        b = radix4_plan_out(b, scale);
        c = radix4_plan_out(c, scale);
        d = radix4_plan_out(d, scale);
    }
    dst[base] = a;
    dst[base + quarter] = b;
    dst[base + 2 * quarter] = c;
    dst[base + 3 * quarter] = d;
}

}  // namespace kofft
