// tables.h -- host-side planner recipes (see tables.cpp).
#pragma once
#include <stddef.h>
namespace kofft_tables {
void twiddles_f32(size_t n, float *out);   // n/2 complex
void twiddles_f64(size_t n, double *out);
void rfft_table_f32(size_t m, float *out); // m complex
void rfft_table_f64(size_t m, double *out);
void hann_f32(size_t len, float *out);
void bluestein_f32(size_t n, size_t m, float *chirp /* n complex */, float *b /* m complex */);
void bluestein_f64(size_t n, size_t m, double *chirp, double *b);
// fft_radix4 (fft.rs:1455-1548), n a power of four: perm = n source indices, w = radix4_triples(n) x (w1, w2, w3) complex
size_t radix4_triples(size_t n);
void radix4_f32(size_t n, unsigned *perm, float *w);
void radix4_f64(size_t n, unsigned *perm, double *w);
}  // namespace kofft_tables
