/*
 * kofft_oracle.c -- CPU oracle for the kofft hot path.  TEST INFRASTRUCTURE ONLY;
 * see kofft_oracle.h for who may load it and for the pinning status.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  No -march flags:
 * the reference's default build has no FMA, so products and sums round separately.
 */
#include "kofft_oracle.h"

#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sincosf / sincos */
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- f32 instantiation ---- */
#define REAL float
#define SFX(name) name##_f32
#define RSINCOS sincosf
#define RCOS cosf
#define RFMA fmaf
/* core::f32::consts::PI (num.rs:2, 57-59) */
#define R_PI 3.14159265358979323846f
#include "kofft_oracle_impl.inc"
#undef REAL
#undef SFX
#undef RSINCOS
#undef RCOS
#undef RFMA
#undef R_PI

/* ---- f64 instantiation ---- */
#define REAL double
#define SFX(name) name##_f64
#define RSINCOS sincos
#define RCOS cos
#define RFMA fma
/* core::f64::consts::PI (num.rs:91-93) */
#define R_PI 3.14159265358979323846
#include "kofft_oracle_impl.inc"
#undef REAL
#undef SFX
#undef RSINCOS
#undef RCOS
#undef RFMA
#undef R_PI

/* window.rs:24-28  hann(len): 0.5 - 0.5 * cos(2.0 * PI * i as f32 / len as f32), all f32 */
int ko_hann_f32(size_t len, float *out)
{
    const float pi = 3.14159265358979323846f;
    for (size_t i = 0; i < len; ++i)
        out[i] = 0.5f - 0.5f * cosf(2.0f * pi * (float)i / (float)len);
    return KO_OK;
}

/* stft.rs:91-103  one frame: window-multiply / zero-pad, then the full complex FFT */
static int stft_frame(ko_planner_f32 *p, const float *signal, size_t len, const float *window,
                      size_t win_len, size_t start, float *frame)
{
    for (size_t i = 0; i < win_len; ++i) {
        float x = (start + i < len) ? signal[start + i] * window[i] : 0.0f;
        frame[2 * i] = x;
        frame[2 * i + 1] = 0.0f;
    }
    return ko_fft_p_f32(p, frame, win_len);
}

/* stft.rs:76-105  stft(): hop==0 -> InvalidHopSize; fewer than ceil(len/hop) frames ->
 * MismatchedLengths; EVERY provided frame is computed (not only the required ones).
 * A zero-length window reaches fft.fft(&mut []) -> EmptyInput (fft.rs:1056). */
int ko_stft_f32(const float *signal, size_t len, const float *window, size_t win_len, size_t hop,
                float *out, size_t frames)
{
    if (hop == 0) return KO_ERR_INVALID_HOP_SIZE;
    size_t required = (len + hop - 1) / hop;
    if (frames < required) return KO_ERR_MISMATCHED_LENGTHS;
    return ko_stft_range_f32(signal, len, window, win_len, hop, out, 0, frames);
}

int ko_stft_range_f32(const float *signal, size_t len, const float *window, size_t win_len,
                      size_t hop, float *out, size_t first, size_t count)
{
    if (hop == 0) return KO_ERR_INVALID_HOP_SIZE;
    ko_planner_f32 *p = ko_planner_new_f32();
    if (!p) return KO_ERR_ALLOC;
    int rc = KO_OK;
    for (size_t f = 0; f < count && rc == KO_OK; ++f)
        rc = stft_frame(p, signal, len, window, win_len, (first + f) * hop, out + 2 * win_len * f);
    ko_planner_free_f32(p);
    return rc;
}

/* stft.rs:117-156  istft(): overlap-add with window^2 normalisation (> 1e-8 guard).
 * frames_data (frames*win_len complex) is transformed in place, like the reference's
 * &mut frames.  `output` is accumulated into (+=): the reference does not clear it. */
int ko_istft_f32(float *frames_data, size_t frames, const float *window, size_t win_len,
                 size_t hop, float *output, size_t out_len, float *scratch, size_t scratch_len)
{
    if (hop == 0) return KO_ERR_INVALID_HOP_SIZE;
    if (scratch_len != out_len) return KO_ERR_MISMATCHED_LENGTHS;
    for (size_t i = 0; i < scratch_len; ++i) scratch[i] = 0.0f;
    ko_planner_f32 *p = ko_planner_new_f32();
    if (!p) return KO_ERR_ALLOC;
    int rc = KO_OK;
    for (size_t f = 0; f < frames && rc == KO_OK; ++f) {
        size_t start = f * hop;
        float *frame = frames_data + 2 * win_len * f;
        rc = ko_ifft_p_f32(p, frame, win_len);
        if (rc) break;
        for (size_t i = 0; i < win_len; ++i) {
            if (start + i < out_len) {
                output[start + i] += frame[2 * i] * window[i];
                scratch[start + i] += window[i] * window[i];
            }
        }
    }
    ko_planner_free_f32(p);
    if (rc) return rc;
    for (size_t i = 0; i < out_len; ++i)
        if (scratch[i] > 1e-8f) output[i] /= scratch[i];
    return KO_OK;
}

/* visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76): window = hann(win_len) (window.rs:24-28), frames =
 * ceil(len / hop) STFT frames (compute_stft = stft.rs:76-105), then for the first win_len/2 bins of every frame
 *   mag = (c.re * c.re + c.im * c.im).sqrt()        -- f32, un-fused, correctly rounded sqrt (spectrogram.rs:66)
 *   if mag > max_mag { max_mag = mag }              -- a NaN is never selected (spectrogram.rs:68-70)
 * mags: frames * (win_len/2) floats, row-major.  hop == 0: the reference divides by it (div_ceil) and panics. */
int ko_stft_magnitudes_f32(const float *samples, size_t len, size_t win_len, size_t hop, float *mags, float *max_mag)
{
    if (hop == 0) return KO_ERR_INVALID_HOP_SIZE;
    const size_t frames = (len + hop - 1) / hop;
    const size_t height = win_len / 2;
    *max_mag = 0.0f;
    if (frames == 0) return KO_OK;
    float *window = (float *)malloc((win_len ? win_len : 1) * sizeof(float));
    float *spec = (float *)malloc((frames * win_len * 2 + 1) * sizeof(float));
    if (!window || !spec) {
        free(window);
        free(spec);
        return KO_ERR_ALLOC;
    }
    ko_hann_f32(win_len, window);
    int rc = ko_stft_f32(samples, len, window, win_len, hop, spec, frames);
    if (rc == KO_OK) {
        float mx = 0.0f;
        for (size_t x = 0; x < frames; ++x) {
            for (size_t y = 0; y < height; ++y) {
                const float re = spec[2 * (x * win_len + y)], im = spec[2 * (x * win_len + y) + 1];
                const float rr = re * re, ii = im * im;
                const float mag = sqrtf(rr + ii);
                mags[x * height + y] = mag;
                if (mag > mx) mx = mag;
            }
        }
        *max_mag = mx;
    }
    free(window);
    free(spec);
    return rc;
}

const char *ko_strerror(int code)
{
    switch (code) {
    case KO_OK: return "Ok";
    case KO_ERR_EMPTY_INPUT: return "FftError::EmptyInput";
    case KO_ERR_NON_POWER_OF_TWO_NO_STD: return "FftError::NonPowerOfTwoNoStd";
    case KO_ERR_MISMATCHED_LENGTHS: return "FftError::MismatchedLengths";
    case KO_ERR_INVALID_STRIDE: return "FftError::InvalidStride";
    case KO_ERR_INVALID_HOP_SIZE: return "FftError::InvalidHopSize";
    case KO_ERR_INVALID_VALUE: return "FftError::InvalidValue";
    case KO_ERR_UNSUPPORTED: return "oracle: non-power-of-two length (Bluestein arm not restated)";
    case KO_ERR_ALLOC: return "oracle: allocation failure";
    default: return "unknown";
    }
}
