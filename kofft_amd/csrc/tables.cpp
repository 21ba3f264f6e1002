// tables.cpp -- host-side planner recipes of the product (NOT the test oracle).
//
// The device never computes trigonometry: kofft's tables are produced by rounding-
// sensitive recurrences, so they are generated here, with the reference's exact
// order of operations, and uploaded.  Compiled with g++ -ffp-contract=off; fmaf / fma and
// sincosf / sincos come from glibc: Rust's mul_add lowers to the former, and its sin_cos()
// -- `(self.sin(), self.cos())` in std -- is merged by LLVM into the latter on
// x86_64-unknown-linux-gnu.  The pair is requested with ONE explicit sincos call so that the
// tables do not depend on this compiler's own merging heuristics (f64 sincos differs from
// sin / cos in the last bit for ~0.14 % of arguments in glibc 2.35; sincosf never does).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "tables.h"

#include <cmath>

namespace {

template <typename T> struct Num;
template <> struct Num<float> {
    static float pi() { return 3.14159265358979323846f; }  // core::f32::consts::PI
    static void sin_cos(float x, float *s, float *c) { ::sincosf(x, s, c); }
    static float fma(float a, float b, float c) { return fmaf(a, b, c); }
};
template <> struct Num<double> {
    static double pi() { return 3.14159265358979323846; }  // core::f64::consts::PI
    static void sin_cos(double x, double *s, double *c) { ::sincos(x, s, c); }
    static double fma(double a, double b, double c) { return ::fma(a, b, c); }
};

// FftPlanner::get_twiddles (fft.rs:391-405)
template <typename T>
void twiddles(size_t n, T *out)
{
    const size_t half = n / 2;
    const T angle = (-(T)2.0f * Num<T>::pi()) / (T)(float)n;
    T s, c;
    Num<T>::sin_cos(angle, &s, &c);
    T w_re = (T)1, w_im = (T)0;
    for (size_t k = 0; k < half; ++k) {
        out[2 * k] = w_re;
        out[2 * k + 1] = w_im;
        const T prev_re = w_re;
        w_re = Num<T>::fma(w_re, c, -(w_im * s));
        w_im = Num<T>::fma(w_im, c, prev_re * s);
    }
}

// build_twiddle_table (rfft.rs:172-183): cur <- cur.mul(w), un-fused complex product
template <typename T>
void rfft_table(size_t m, T *out)
{
    const T angle = -Num<T>::pi() / (T)(float)m;
    T s, c;
    Num<T>::sin_cos(angle, &s, &c);
    T re = (T)1, im = (T)0;
    for (size_t k = 0; k < m; ++k) {
        out[2 * k] = re;
        out[2 * k + 1] = im;
        const T nre = re * c - im * s;
        const T nim = re * s + im * c;
        re = nre;
        im = nim;
    }
}

// FftPlanner::get_bluestein (fft.rs:411-433): chirp[i] = expi(-a_i), b[i] = expi(a_i) mirrored into the tail of a
// zero-padded length-m buffer, a_i = pi * ((i*i) as f32) / (n as f32).  (b's FFT is taken on the device.)
template <typename T>
void bluestein(size_t n, size_t m, T *chirp, T *b)
{
    for (size_t i = 0; i < 2 * m; ++i) b[i] = (T)0;
    for (size_t i = 0; i < n; ++i) {
        const T angle = Num<T>::pi() * (T)(float)(i * i) / (T)(float)n;
        // Complex::expi (num.rs:123-126): re = cos, im = sin of one sin_cos() call
        Num<T>::sin_cos(-angle, &chirp[2 * i + 1], &chirp[2 * i]);
        Num<T>::sin_cos(angle, &b[2 * i + 1], &b[2 * i]);
    }
    for (size_t i = 1; i < n; ++i) {
        b[2 * (m - i)] = b[2 * i];
        b[2 * (m - i) + 1] = b[2 * i + 1];
    }
}

// ScalarFftImpl::fft_radix4 (fft.rs:1455-1548), the parts that do not depend on the data:
//   perm[i]  = index of the INPUT element that sits at position i after the reference's swap loop (fft.rs:1462-1474;
//              it flips one bit per base-4 digit -- not a digit reversal, which is why the arm is not a DFT from n = 16);
//   w        = for every stage len = 16, 64, .. n, the quarter = len/4 triples (w1, w2, w3)[j] the reference builds by
//              starting at (1, 0) and multiplying (Complex::mul, un-fused) by entries 1, 2, 3 of get_twiddles(len) after
//              every j (fft.rs:1490-1530; the sequence restarts for every block i, so one copy per stage serves all).
//              Triples of consecutive stages follow each other (radix4_triples(n) in all).
template <typename T>
void radix4(size_t n, unsigned *perm, T *w)
{
    for (size_t i = 0; i < n; ++i) perm[i] = (unsigned)i;
    size_t j = 0;
    for (size_t i = 1; i < n; ++i) {
        size_t bit = n >> 2;
        while (j & bit) {
            j ^= bit;
            bit >>= 2;
        }
        j ^= bit;
        if (i < j) {
            const unsigned t = perm[i];
            perm[i] = perm[j];
            perm[j] = t;
        }
    }
    T *tw = new T[n >= 16 ? n : 16];
    size_t off = 0;  // triples written so far
    for (size_t len = 16; len <= n; len <<= 2) {
        twiddles<T>(len, tw);
        const T s_re[3] = {tw[2], tw[4], tw[6]}, s_im[3] = {tw[3], tw[5], tw[7]};
        T re[3] = {(T)1, (T)1, (T)1}, im[3] = {(T)0, (T)0, (T)0};
        const size_t quarter = len / 4;
        for (size_t q = 0; q < quarter; ++q) {
            for (int k = 0; k < 3; ++k) {
                w[2 * (3 * (off + q) + k)] = re[k];
                w[2 * (3 * (off + q) + k) + 1] = im[k];
                const T nre = re[k] * s_re[k] - im[k] * s_im[k];  // Complex::mul (num.rs:161-166), un-fused
                const T nim = re[k] * s_im[k] + im[k] * s_re[k];
                re[k] = nre;
                im[k] = nim;
            }
        }
        off += quarter;
    }
    delete[] tw;
}

}  // namespace

namespace kofft_tables {
size_t radix4_triples(size_t n)
{
    size_t t = 0;
    for (size_t len = 16; len <= n; len <<= 2) t += len / 4;
    return t;
}
void radix4_f32(size_t n, unsigned *perm, float *w) { radix4<float>(n, perm, w); }
void radix4_f64(size_t n, unsigned *perm, double *w) { radix4<double>(n, perm, w); }
void bluestein_f32(size_t n, size_t m, float *chirp, float *b) { bluestein<float>(n, m, chirp, b); }
void bluestein_f64(size_t n, size_t m, double *chirp, double *b) { bluestein<double>(n, m, chirp, b); }
void twiddles_f32(size_t n, float *out) { twiddles<float>(n, out); }
void twiddles_f64(size_t n, double *out) { twiddles<double>(n, out); }
void rfft_table_f32(size_t m, float *out) { rfft_table<float>(m, out); }
void rfft_table_f64(size_t m, double *out) { rfft_table<double>(m, out); }
// window::hann (window.rs:24-28), all arithmetic in f32
void hann_f32(size_t len, float *out)
{
    const float pi = 3.14159265358979323846f;
    for (size_t i = 0; i < len; ++i) out[i] = 0.5f - 0.5f * cosf(2.0f * pi * (float)i / (float)len);
}
}  // namespace kofft_tables
