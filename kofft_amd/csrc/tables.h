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
}  // namespace kofft_tables
