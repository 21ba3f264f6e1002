// kofft_hip.hpp -- C++17 host mirror of kofft's operator interface for the hot path, on top of the C ABI
// (kofft_hip.h).  The reference is a Rust crate and the build image has no Rust toolchain, so this header plays
// the role the Rust shim (integration/rust/kofft-hip) plays for Rust callers: same names, same argument meaning,
// same error variants and validation order, so that tests read like the reference's own.
//
//   kofft::FftError, FftStrategy                 fft.rs:447-463
//   kofft::Complex<T>, Complex32, Complex64      num.rs:105-110, 232-233   (#[repr(C)] {re, im})
//   kofft::FftImpl<T>  (abstract)                fft.rs:466-587
//   kofft::HipFftImpl<T>  ~ ScalarFftImpl<T>     fft.rs:600-613 + RealFftImpl blanket methods (rfft.rs:775-837)
//   kofft::FftPlanner<T>, RfftPlanner<T>         fft.rs:332-445, rfft.rs:194-338
//   kofft::batch / batch_inverse / multi_channel fft.rs:2156-2191
//   kofft::stft / istft / parallel / frame / StftStream  stft.rs:76-156, 232-263, 355-372, 160-206
//   kofft::hann                                  window.rs:24-28
//
// Result<(), FftError> becomes kofft::Result (is_ok / is_err / unwrap / unwrap_err).  A negative C-ABI status
// (HIP failure, unsupported length) has no FftError variant: it throws kofft::DeviceError, the C++ analogue of
// the Rust shim's panic.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "kofft_hip.h"

namespace kofft {

enum class FftError : int {  // fft.rs:447-454, discriminant + 1 == C ABI status
    EmptyInput = 1,
    NonPowerOfTwoNoStd = 2,
    MismatchedLengths = 3,
    InvalidStride = 4,
    InvalidHopSize = 5,
    InvalidValue = 6,
};

enum class FftStrategy { Radix2, Radix4, SplitRadix, Auto };  // fft.rs:456-463

struct DeviceError : std::runtime_error {
    int status;
    DeviceError(int s, const std::string &detail)
        : std::runtime_error(std::string("kofft_hip status ") + std::to_string(s) + ": " + kofft_hip_strerror(s) +
                             (detail.empty() ? "" : " [" + detail + "]")),
          status(s)
    {
    }
};

class Result {
public:
    Result() : code_(0) {}
    explicit Result(FftError e) : code_(static_cast<int>(e)) {}
    static Result Ok() { return Result(); }
    static Result Err(FftError e) { return Result(e); }
    bool is_ok() const { return code_ == 0; }
    bool is_err() const { return code_ != 0; }
    void unwrap() const
    {
        if (code_ != 0) throw std::logic_error(std::string("called unwrap() on Err(") + kofft_hip_strerror(code_) + ")");
    }
    FftError unwrap_err() const
    {
        if (code_ == 0) throw std::logic_error("called unwrap_err() on Ok");
        return static_cast<FftError>(code_);
    }
    bool operator==(const Result &o) const { return code_ == o.code_; }

private:
    int code_;
};

template <typename T>
struct Complex {  // layout-compatible with T[2] (tests/complex_repr.rs)
    T re, im;
    Complex() : re(0), im(0) {}
    Complex(T r, T i) : re(r), im(i) {}
    static Complex zero() { return Complex(0, 0); }
    Complex add(Complex o) const { return Complex(re + o.re, im + o.im); }
    Complex sub(Complex o) const { return Complex(re - o.re, im - o.im); }
    Complex mul(Complex o) const { return Complex(re * o.re - im * o.im, re * o.im + im * o.re); }  // num.rs:161-166
    bool operator==(const Complex &o) const { return re == o.re && im == o.im; }
};
using Complex32 = Complex<float>;
using Complex64 = Complex<double>;
static_assert(sizeof(Complex32) == 2 * sizeof(float) && sizeof(Complex64) == 2 * sizeof(double), "repr(C) layout");

namespace detail {
template <typename T> struct Abi;
template <> struct Abi<float> {
    static int fft(kofft_hip_ctx *c, float *d, size_t n, size_t b, int inv) { return kofft_hip_fft_c32(c, d, n, b, inv); }
    static int strided(kofft_hip_ctx *c, float *d, size_t len, size_t st, size_t n, int inv) { return kofft_hip_fft_c32_strided(c, d, len, st, n, inv); }
    static int rfft(kofft_hip_ctx *c, const float *i, float *o, const float *w, size_t n, size_t b) { return kofft_hip_rfft_f32(c, i, o, w, n, b); }
    static int irfft(kofft_hip_ctx *c, const float *i, float *o, size_t n, size_t b) { return kofft_hip_irfft_f32(c, i, o, n, b); }
    static int twiddles(size_t n, float *o) { return kofft_hip_twiddles_f32(n, o); }
    static int rfft_table(size_t m, float *o) { return kofft_hip_rfft_table_f32(m, o); }
    static int fftnd(kofft_hip_ctx *c, float *d, size_t dp, size_t r, size_t cl, int inv) { return kofft_hip_fftnd_c32(c, d, dp, r, cl, inv); }
    static int radix4(kofft_hip_ctx *c, float *d, size_t n, size_t b) { return kofft_hip_fft_radix4_c32(c, d, n, b); }
    static int iradix4(kofft_hip_ctx *c, float *d, size_t n, size_t b) { return kofft_hip_ifft_radix4_c32(c, d, n, b); }
};
template <> struct Abi<double> {
    static int fft(kofft_hip_ctx *c, double *d, size_t n, size_t b, int inv) { return kofft_hip_fft_c64(c, d, n, b, inv); }
    static int strided(kofft_hip_ctx *c, double *d, size_t len, size_t st, size_t n, int inv) { return kofft_hip_fft_c64_strided(c, d, len, st, n, inv); }
    static int rfft(kofft_hip_ctx *c, const double *i, double *o, const double *w, size_t n, size_t b) { return kofft_hip_rfft_f64(c, i, o, w, n, b); }
    static int irfft(kofft_hip_ctx *c, const double *i, double *o, size_t n, size_t b) { return kofft_hip_irfft_f64(c, i, o, n, b); }
    static int twiddles(size_t n, double *o) { return kofft_hip_twiddles_f64(n, o); }
    static int rfft_table(size_t m, double *o) { return kofft_hip_rfft_table_f64(m, o); }
    static int fftnd(kofft_hip_ctx *c, double *d, size_t dp, size_t r, size_t cl, int inv) { return kofft_hip_fftnd_c64(c, d, dp, r, cl, inv); }
    static int radix4(kofft_hip_ctx *c, double *d, size_t n, size_t b) { return kofft_hip_fft_radix4_c64(c, d, n, b); }
    static int iradix4(kofft_hip_ctx *c, double *d, size_t n, size_t b) { return kofft_hip_ifft_radix4_c64(c, d, n, b); }
};
}  // namespace detail

// trait FftImpl<T> (fft.rs:466-587): the required methods are pure virtual, the provided ones have the
// reference's default bodies.
template <typename T>
class FftImpl {
public:
    using C = Complex<T>;
    virtual ~FftImpl() = default;
    virtual Result fft(std::vector<C> &input) const = 0;
    virtual Result ifft(std::vector<C> &input) const = 0;
    virtual Result fft_strided(std::vector<C> &input, size_t stride, std::vector<C> &scratch) const = 0;
    virtual Result ifft_strided(std::vector<C> &input, size_t stride, std::vector<C> &scratch) const = 0;
    virtual Result fft_out_of_place_strided(const std::vector<C> &input, size_t in_stride, std::vector<C> &output,
                                            size_t out_stride) const = 0;
    virtual Result ifft_out_of_place_strided(const std::vector<C> &input, size_t in_stride, std::vector<C> &output,
                                             size_t out_stride) const = 0;
    virtual Result fft_with_strategy(std::vector<C> &input, FftStrategy strategy) const = 0;

    Result fft_out_of_place(const std::vector<C> &input, std::vector<C> &output) const  // fft.rs:469-479
    {
        if (input.size() != output.size()) return Result::Err(FftError::MismatchedLengths);
        output = input;
        return fft(output);
    }
    Result ifft_out_of_place(const std::vector<C> &input, std::vector<C> &output) const  // fft.rs:480-490
    {
        if (input.size() != output.size()) return Result::Err(FftError::MismatchedLengths);
        output = input;
        return ifft(output);
    }
    Result fft_split(std::vector<T> &re, std::vector<T> &im) const  // fft.rs:556-570
    {
        if (re.size() != im.size()) return Result::Err(FftError::MismatchedLengths);
        std::vector<C> buf(re.size());
        for (size_t i = 0; i < re.size(); ++i) buf[i] = C(re[i], im[i]);
        Result r = fft(buf);
        if (r.is_err()) return r;
        for (size_t i = 0; i < re.size(); ++i) { re[i] = buf[i].re; im[i] = buf[i].im; }
        return Result::Ok();
    }
    Result ifft_split(std::vector<T> &re, std::vector<T> &im) const  // fft.rs:572-586
    {
        if (re.size() != im.size()) return Result::Err(FftError::MismatchedLengths);
        std::vector<C> buf(re.size());
        for (size_t i = 0; i < re.size(); ++i) buf[i] = C(re[i], im[i]);
        Result r = ifft(buf);
        if (r.is_err()) return r;
        for (size_t i = 0; i < re.size(); ++i) { re[i] = buf[i].re; im[i] = buf[i].im; }
        return Result::Ok();
    }
};

// Drop-in for ScalarFftImpl<T>: one device context per instance, not shared between threads (fft.rs:589-605).
template <typename T>
class HipFftImpl : public FftImpl<T> {
public:
    using C = Complex<T>;
    explicit HipFftImpl(int device = 0)
    {
        int rc = kofft_hip_create(device, &ctx_);
        if (rc != 0) throw DeviceError(rc, "kofft_hip_create");
        const char *e = std::getenv("KOFFT_HIP_RADIX4_COMPAT");
        radix4_compat = !(e && e[0] == '0');
    }
    // fft_with_strategy(.., Radix4) reproduces the reference's fft_radix4 bytes (fft.rs:1356, 1455-1548; NOT a DFT from
    // n = 16): strict drop-in, ON by default since round 6.  KOFFT_HIP_RADIX4_COMPAT=0, or false here, opts out (the true
    // transform for every strategy).
    bool radix4_compat = true;
    static HipFftImpl default_() { return HipFftImpl(0); }
    ~HipFftImpl() override { if (ctx_) kofft_hip_destroy(ctx_); }
    HipFftImpl(const HipFftImpl &) = delete;
    HipFftImpl &operator=(const HipFftImpl &) = delete;
    HipFftImpl(HipFftImpl &&o) noexcept : ctx_(o.ctx_) { o.ctx_ = nullptr; }
    kofft_hip_ctx *raw() const { return ctx_; }

    Result fft(std::vector<C> &input) const override { return st(detail::Abi<T>::fft(ctx_, fp(input), input.size(), 1, 0)); }
    Result ifft(std::vector<C> &input) const override { return st(detail::Abi<T>::fft(ctx_, fp(input), input.size(), 1, 1)); }
    Result stockham_fft(std::vector<C> &input) const { return fft(input); }  // fft.rs:634-640
    Result fft_strided(std::vector<C> &input, size_t stride, std::vector<C> &scratch) const override
    {
        return st(detail::Abi<T>::strided(ctx_, fp(input), input.size(), stride, scratch.size(), 0));
    }
    Result ifft_strided(std::vector<C> &input, size_t stride, std::vector<C> &scratch) const override
    {
        return st(detail::Abi<T>::strided(ctx_, fp(input), input.size(), stride, scratch.size(), 1));
    }
    Result fft_out_of_place_strided(const std::vector<C> &input, size_t in_stride, std::vector<C> &output,
                                    size_t out_stride) const override
    {
        return oop_strided(input, in_stride, output, out_stride, false);
    }
    Result ifft_out_of_place_strided(const std::vector<C> &input, size_t in_stride, std::vector<C> &output,
                                     size_t out_stride) const override
    {
        return oop_strided(input, in_stride, output, out_stride, true);
    }
    // fft.rs:1337-1363.  Radix2 / SplitRadix / Auto run the Stockham path (fft and stockham_fft agree for n >= 2); Radix4
    // runs fft_radix4 (fft.rs:1356) like the reference -- not a DFT from n = 16 (its digit-reversal loop is wrong; DESIGN.md
    // section 1), but a drop-in returns the reference's bytes (kofft_hip_fft_radix4_*).  radix4_compat = false opts out.
    Result fft_with_strategy(std::vector<C> &input, FftStrategy strategy) const override
    {
        if (input.empty()) return Result::Err(FftError::EmptyInput);
        if (input.size() == 1) return Result::Ok();
        if (strategy == FftStrategy::Radix4 && radix4_compat) return fft_radix4(input);
        return fft(input);
    }
    // ScalarFftImpl::fft_radix4 (fft.rs:1455-1548), the reference's bytes
    Result fft_radix4(std::vector<C> &input) const { return st(detail::Abi<T>::radix4(ctx_, fp(input), input.size(), 1)); }
    // FftPlan::ifft's loop around that arm (fft.rs:2040-2055): conj, fft_radix4, conj * 1/(n as f32), on the device
    Result ifft_radix4(std::vector<C> &input) const
    {
        if (input.empty()) return Result::Err(FftError::EmptyInput);
        return st(detail::Abi<T>::iradix4(ctx_, fp(input), input.size(), 1));
    }

    // RealFftImpl<T> blanket methods (rfft.rs:780-833); checks in rfft_direct's order (rfft.rs:433-443)
    Result rfft_with_scratch(std::vector<T> &input, std::vector<C> &output, std::vector<C> &scratch) const
    {
        const size_t n = input.size();
        if (n == 0) return Result::Err(FftError::EmptyInput);
        if (n % 2 != 0) return Result::Err(FftError::InvalidValue);
        if (output.size() != n / 2 + 1 || scratch.size() < n / 2) return Result::Err(FftError::MismatchedLengths);
        return st(detail::Abi<T>::rfft(ctx_, input.data(), fp(output), nullptr, n, 1));
    }
    Result rfft(std::vector<T> &input, std::vector<C> &output) const
    {
        std::vector<C> scratch(input.size() / 2);
        return rfft_with_scratch(input, output, scratch);
    }
    Result irfft_with_scratch(std::vector<C> &input, std::vector<T> &output, std::vector<C> &scratch) const
    {
        const size_t n = output.size();
        if (n == 0) return Result::Err(FftError::EmptyInput);
        if (n % 2 != 0) return Result::Err(FftError::InvalidValue);
        if (input.size() != n / 2 + 1 || scratch.size() < n / 2) return Result::Err(FftError::MismatchedLengths);
        return st(detail::Abi<T>::irfft(ctx_, reinterpret_cast<const T *>(input.data()), output.data(), n, 1));
    }
    Result irfft(std::vector<C> &input, std::vector<T> &output) const
    {
        std::vector<C> scratch(output.size() / 2);
        return irfft_with_scratch(input, output, scratch);
    }

    // added: contiguous batch (fft::batch over one buffer)
    Result fft_batch(std::vector<C> &data, size_t n, bool inverse = false) const
    {
        if (n != 0 && data.size() % n != 0) return Result::Err(FftError::MismatchedLengths);
        return st(detail::Abi<T>::fft(ctx_, fp(data), n, n ? data.size() / n : 1, inverse ? 1 : 0));
    }

    Result st(int rc) const
    {
        if (rc == 0) return Result::Ok();
        if (rc > 0) return Result::Err(static_cast<FftError>(rc));
        throw DeviceError(rc, kofft_hip_last_error(ctx_));
    }

private:
    static T *fp(std::vector<C> &v) { return reinterpret_cast<T *>(v.data()); }
    Result oop_strided(const std::vector<C> &input, size_t in_stride, std::vector<C> &output, size_t out_stride,
                       bool inverse) const  // fft.rs:1261-1336
    {
        if (in_stride == 0 || out_stride == 0) return Result::Err(FftError::InvalidStride);
        if (input.size() % in_stride != 0 || output.size() % out_stride != 0) return Result::Err(FftError::InvalidStride);
        const size_t n = input.size() / in_stride;
        if (output.size() / out_stride != n) return Result::Err(FftError::MismatchedLengths);
        std::vector<C> scratch(n);
        for (size_t i = 0; i < n; ++i) scratch[i] = input[i * in_stride];
        Result r = inverse ? ifft(scratch) : fft(scratch);
        if (r.is_err()) return r;
        for (size_t i = 0; i < n; ++i) output[i * out_stride] = scratch[i];
        return Result::Ok();
    }
    kofft_hip_ctx *ctx_ = nullptr;
};

template <typename T>
class FftPlanner {  // fft.rs:332-445
public:
    std::vector<Complex<T>> get_twiddles(size_t n) const
    {
        std::vector<Complex<T>> t(n / 2);
        detail::Abi<T>::twiddles(n, reinterpret_cast<T *>(t.data()));
        return t;
    }
    FftStrategy plan_strategy(size_t n) const { return (n > 1 && (n & (n - 1)) == 0) ? FftStrategy::SplitRadix : FftStrategy::Auto; }
};

template <typename T>
class RfftPlanner {  // rfft.rs:194-338
public:
    std::vector<Complex<T>> get_twiddles(size_t m) const
    {
        std::vector<Complex<T>> t(m);
        detail::Abi<T>::rfft_table(m, reinterpret_cast<T *>(t.data()));
        return t;
    }
    Result rfft_with_scratch(const HipFftImpl<T> &fft, std::vector<T> &input, std::vector<Complex<T>> &output,
                             std::vector<Complex<T>> &scratch) const
    {
        return fft.rfft_with_scratch(input, output, scratch);
    }
    Result irfft_with_scratch(const HipFftImpl<T> &fft, std::vector<Complex<T>> &input, std::vector<T> &output,
                              std::vector<Complex<T>> &scratch) const
    {
        return fft.irfft_with_scratch(input, output, scratch);
    }
};

// fft::FftPlan (fft.rs:1989-2094): a length and a strategy bound to an implementation.  FftPlan::fft (fft.rs:2012-2038):
// an f32 plan with strategy Radix2 / Radix4 takes the *_with_twiddles shortcut, and every *_with_twiddles is stockham_fft
// (fft.rs:1645-1660) -- the Stockham transform, NOT fft_radix4; every other plan goes through fft_with_strategy, where
// Radix4 means fft_radix4.  FftPlan::ifft (fft.rs:2040-2055) = conj, that same fft, conj * 1/(n as f32).  Not reproduced:
// the reference's f32 Radix2 / Radix4 plan of length 1 panics (unreachable!() in stockham_fft, fft.rs:655-662); identity here.
template <typename T>
class FftPlan {
public:
    size_t n;
    FftStrategy strategy;
    FftPlan(size_t n_, FftStrategy s, const HipFftImpl<T> &f) : n(n_), strategy(s), fft_(f) {}
    Result fft(std::vector<Complex<T>> &input) const
    {
        if (input.size() != n) return Result::Err(FftError::MismatchedLengths);
        if (stockham_shortcut()) return n == 1 ? Result::Ok() : fft_.fft(input);
        return fft_.fft_with_strategy(input, strategy);
    }
    Result ifft(std::vector<Complex<T>> &input) const
    {
        if (input.size() != n) return Result::Err(FftError::MismatchedLengths);
        if (n <= 1) return fft(input);
        if (strategy == FftStrategy::Radix4 && fft_.radix4_compat && !stockham_shortcut()) return fft_.ifft_radix4(input);
        return fft_.ifft(input);  // conj, fft, conj, * 1/(n as f32): ifft's arithmetic
    }
    Result fft_out_of_place(const std::vector<Complex<T>> &input, std::vector<Complex<T>> &output) const
    {
        if (input.size() != n || output.size() != n) return Result::Err(FftError::MismatchedLengths);
        output = input;
        return fft(output);
    }
    Result ifft_out_of_place(const std::vector<Complex<T>> &input, std::vector<Complex<T>> &output) const
    {
        if (input.size() != n || output.size() != n) return Result::Err(FftError::MismatchedLengths);
        output = input;
        return ifft(output);
    }

private:
    bool stockham_shortcut() const  // fft.rs:2016-2035
    {
        return sizeof(T) == 4 && (strategy == FftStrategy::Radix2 || strategy == FftStrategy::Radix4);
    }
    const HipFftImpl<T> &fft_;
};

// rfft::rfft_packed / irfft_packed (rfft.rs:341-420): the same checks and arithmetic as rfft_direct / irfft_direct
template <typename T>
Result rfft_packed(RfftPlanner<T> &, const HipFftImpl<T> &fft, std::vector<T> &input, std::vector<Complex<T>> &output,
                   std::vector<Complex<T>> &scratch)
{
    return fft.rfft_with_scratch(input, output, scratch);
}
template <typename T>
Result irfft_packed(RfftPlanner<T> &, const HipFftImpl<T> &fft, std::vector<Complex<T>> &input, std::vector<T> &output,
                    std::vector<Complex<T>> &scratch)
{
    return fft.irfft_with_scratch(input, output, scratch);
}

// fft::batch / batch_inverse / multi_channel (fft.rs:2156-2191): serial semantics, first error wins
template <typename T>
Result batch(const FftImpl<T> &fft, std::vector<std::vector<Complex<T>>> &batches)
{
    for (auto &b : batches) {
        Result r = fft.fft(b);
        if (r.is_err()) return r;
    }
    return Result::Ok();
}
template <typename T>
Result batch_inverse(const FftImpl<T> &fft, std::vector<std::vector<Complex<T>>> &batches)
{
    for (auto &b : batches) {
        Result r = fft.ifft(b);
        if (r.is_err()) return r;
    }
    return Result::Ok();
}
template <typename T>
Result multi_channel(const FftImpl<T> &fft, std::vector<std::vector<Complex<T>>> &channels) { return batch(fft, channels); }

inline std::vector<float> hann(size_t len)  // window.rs:24-28
{
    std::vector<float> w(len);
    kofft_hip_hann_f32(len, w.data());
    return w;
}

// stft::stft (stft.rs:76-105): output frames are resized to the window length and overwritten
inline Result stft(const std::vector<float> &signal, const std::vector<float> &window, size_t hop_size,
                   std::vector<std::vector<Complex32>> &output, const HipFftImpl<float> &fft, bool check_frames = true)
{
    const size_t frames = output.size(), wl = window.size();
    std::vector<Complex32> flat(frames * wl);
    int rc = check_frames
                 ? kofft_hip_stft_f32(fft.raw(), signal.data(), signal.size(), window.data(), wl, hop_size,
                                      reinterpret_cast<float *>(flat.data()), frames)
                 : kofft_hip_stft_parallel_f32(fft.raw(), signal.data(), signal.size(), window.data(), wl, hop_size,
                                               reinterpret_cast<float *>(flat.data()), frames);
    Result r = fft.st(rc);
    if (r.is_err()) return r;
    for (size_t f = 0; f < frames; ++f) output[f].assign(flat.begin() + f * wl, flat.begin() + (f + 1) * wl);
    return Result::Ok();
}
// stft::istft (stft.rs:117-156): frames are inverse-transformed in place; output is accumulated into and normalised
inline Result istft(std::vector<std::vector<Complex32>> &frames, const std::vector<float> &window, size_t hop_size,
                    std::vector<float> &output, std::vector<float> &scratch, const HipFftImpl<float> &fft)
{
    if (hop_size == 0) return Result::Err(FftError::InvalidHopSize);
    if (scratch.size() != output.size()) return Result::Err(FftError::MismatchedLengths);
    const size_t wl = window.size();
    for (auto &f : frames)
        if (f.size() != wl) return Result::Err(FftError::MismatchedLengths);
    std::vector<Complex32> flat(frames.size() * wl);
    for (size_t f = 0; f < frames.size(); ++f) std::copy(frames[f].begin(), frames[f].end(), flat.begin() + f * wl);
    Result r = fft.st(kofft_hip_istft_f32(fft.raw(), reinterpret_cast<float *>(flat.data()), frames.size(), window.data(), wl,
                                          hop_size, output.data(), output.size(), scratch.data(), scratch.size()));
    if (r.is_err()) return r;
    for (size_t f = 0; f < frames.size(); ++f) frames[f].assign(flat.begin() + f * wl, flat.begin() + (f + 1) * wl);
    return Result::Ok();
}
// stft::inverse_parallel (stft.rs:289-343): frames are left untouched, samples whose window-square sum is <= 1e-8
// become 0; only hop == 0 is rejected (a frame shorter than the window panics in the reference: out_of_range here)
inline Result inverse_parallel(const std::vector<std::vector<Complex32>> &frames, const std::vector<float> &window,
                               size_t hop_size, std::vector<float> &output, const HipFftImpl<float> &fft)
{
    if (hop_size == 0) return Result::Err(FftError::InvalidHopSize);
    const size_t wl = window.size();
    std::vector<Complex32> flat(frames.size() * wl);
    for (size_t f = 0; f < frames.size(); ++f) {
        if (frames[f].size() != wl) throw std::out_of_range("frame length differs from the window (the reference panics)");
        std::copy(frames[f].begin(), frames[f].end(), flat.begin() + f * wl);
    }
    return fft.st(kofft_hip_istft_parallel_f32(fft.raw(), reinterpret_cast<const float *>(flat.data()), frames.size(),
                                               window.data(), wl, hop_size, output.data(), output.size()));
}
// stft::inverse_frame (stft.rs:384-399): ifft(frame) in place, windowed add into output from `start`, no normalisation
inline Result inverse_frame(std::vector<Complex32> &frame_io, const std::vector<float> &window, size_t start,
                            std::vector<float> &output, const HipFftImpl<float> &fft)
{
    if (frame_io.size() != window.size()) throw std::out_of_range("frame length differs from the window (the reference panics)");
    return fft.st(kofft_hip_istft_frame_f32(fft.raw(), reinterpret_cast<float *>(frame_io.data()), window.data(), window.size(),
                                            start, output.data(), output.size()));
}
// visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76): (frames x win_len/2 magnitudes, max magnitude)
inline Result stft_magnitudes(const std::vector<float> &samples, size_t win_len, size_t hop,
                              std::vector<std::vector<float>> &mags, float &max_mag, const HipFftImpl<float> &fft)
{
    if (hop == 0) return Result::Err(FftError::InvalidHopSize);  // the reference divides by hop (panic)
    const size_t frames = (samples.size() + hop - 1) / hop, half = win_len / 2;
    std::vector<float> flat(frames * half);
    Result r = fft.st(kofft_hip_stft_magnitudes_f32(fft.raw(), samples.data(), samples.size(), win_len, hop, flat.data(), frames,
                                                    &max_mag));
    if (r.is_err()) return r;
    mags.assign(frames, std::vector<float>());
    for (size_t f = 0; f < frames; ++f) mags[f].assign(flat.begin() + f * half, flat.begin() + (f + 1) * half);
    return Result::Ok();
}

// ndfft::fft2d_inplace (ndfft.rs:74-101) / fft3d_inplace (ndfft.rs:114-155): same length checks, same order
template <typename T>
Result fft2d_inplace(std::vector<Complex<T>> &data, size_t rows, size_t cols, const HipFftImpl<T> &fft,
                     std::vector<Complex<T>> &scratch_col)
{
    if (rows * cols != data.size()) return Result::Err(FftError::MismatchedLengths);
    if (rows == 0 || cols == 0) return Result::Ok();
    if (scratch_col.size() != rows) return Result::Err(FftError::MismatchedLengths);
    return fft.st(detail::Abi<T>::fftnd(fft.raw(), reinterpret_cast<T *>(data.data()), 1, rows, cols, 0));
}
template <typename T>
Result fft3d_inplace(std::vector<Complex<T>> &data, size_t depth, size_t rows, size_t cols, const HipFftImpl<T> &fft,
                     std::vector<Complex<T>> &tube, std::vector<Complex<T>> &row, std::vector<Complex<T>> &col)
{
    if (depth * rows * cols != data.size()) return Result::Err(FftError::MismatchedLengths);
    if (depth == 0 || rows == 0 || cols == 0) return Result::Ok();
    if (tube.size() != depth || row.size() != rows || col.size() != cols) return Result::Err(FftError::MismatchedLengths);
    return fft.st(detail::Abi<T>::fftnd(fft.raw(), reinterpret_cast<T *>(data.data()), depth, rows, cols, 0));
}

// Multi-GPU STFT: frames sharded over the devices of one process, optional RCCL all-gather of the
// spectra -- the device analogue of stft::parallel's rayon-over-frames (stft.rs:232-263).
class HipMulti {
public:
    explicit HipMulti(int ngpu, const int *devices = nullptr)
    {
        const int rc = kofft_hip_multi_create(ngpu, devices, &h_);
        if (rc > 0) throw std::invalid_argument("kofft_hip_multi_create: invalid device count");
        if (rc != 0) throw DeviceError(rc, "kofft_hip_multi_create");
    }
    ~HipMulti() { if (h_) kofft_hip_multi_destroy(h_); }
    HipMulti(const HipMulti &) = delete;
    HipMulti &operator=(const HipMulti &) = delete;
    int ngpu() const { return kofft_hip_multi_ngpu(h_); }
    // stft::stft's checks and result; frames [r*ceil(F/G), ...) computed by device r
    Result stft(const std::vector<float> &signal, const std::vector<float> &window, size_t hop_size,
                std::vector<std::vector<Complex32>> &output, bool allgather = false)
    {
        const size_t frames = output.size(), wl = window.size();
        std::vector<Complex32> flat(frames * wl);
        const int rc = kofft_hip_multi_stft_f32(h_, signal.data(), signal.size(), window.data(), wl, hop_size,
                                                reinterpret_cast<float *>(flat.data()), frames, allgather ? 1 : 0, nullptr);
        if (rc > 0) return Result::Err(static_cast<FftError>(rc));
        if (rc != 0) throw DeviceError(rc, kofft_hip_multi_last_error(h_));
        for (size_t f = 0; f < frames; ++f) output[f].assign(flat.begin() + f * wl, flat.begin() + (f + 1) * wl);
        return Result::Ok();
    }
    // fft::batch (fft.rs:2156-2175) over a contiguous batch, in G contiguous blocks, no exchange
    Result fft_batch(std::vector<Complex32> &data, size_t n, bool inverse = false)
    {
        return st(kofft_hip_multi_fft_c32(h_, reinterpret_cast<float *>(data.data()), n, n ? data.size() / n : 0, inverse ? 1 : 0));
    }
    Result fft_batch(std::vector<Complex64> &data, size_t n, bool inverse = false)
    {
        return st(kofft_hip_multi_fft_c64(h_, reinterpret_cast<double *>(data.data()), n, n ? data.size() / n : 0, inverse ? 1 : 0));
    }
    // rfft_direct (rfft.rs:425-465) on every row of n reals, optional window product first (BASELINE config #3's shape)
    Result rfft_batch(const std::vector<float> &rows, size_t n, std::vector<Complex32> &out, const std::vector<float> *window = nullptr)
    {
        const size_t batch = n ? rows.size() / n : 0;
        if (window && window->size() != n) return Result::Err(FftError::MismatchedLengths);
        out.resize(batch * (n / 2 + 1));
        return st(kofft_hip_multi_rfft_f32(h_, rows.data(), reinterpret_cast<float *>(out.data()), window ? window->data() : nullptr, n, batch));
    }
    // device-resident twins (one device pointer per device, asynchronous): thin pass-throughs
    Result fft_dev(float *const *d_data, size_t n, size_t batch, bool inverse = false)
    {
        return st(kofft_hip_multi_fft_c32_dev(h_, d_data, n, batch, inverse ? 1 : 0));
    }
    Result fft_dev(double *const *d_data, size_t n, size_t batch, bool inverse = false)
    {
        return st(kofft_hip_multi_fft_c64_dev(h_, d_data, n, batch, inverse ? 1 : 0));
    }
    Result rfft_dev(const float *const *d_in, float *const *d_out, const float *const *d_window, size_t n, size_t batch)
    {
        return st(kofft_hip_multi_rfft_f32_dev(h_, d_in, d_out, d_window, n, batch));
    }
    Result stft_dev(const float *const *d_signal, size_t len, const float *const *d_window, size_t win_len, size_t hop, size_t frames,
                    bool allgather, float **d_out)
    {
        return st(kofft_hip_multi_stft_f32_dev(h_, d_signal, len, d_window, win_len, hop, frames, allgather ? 1 : 0, d_out));
    }
    std::pair<size_t, size_t> shard(size_t total, int rank) const
    {
        size_t f = 0, c = 0;
        kofft_hip_multi_shard(h_, total, rank, &f, &c);
        return {f, c};
    }
    std::pair<size_t, size_t> stft_slice(size_t len, size_t win_len, size_t hop, size_t frames, int rank) const
    {
        size_t f = 0, c = 0;
        kofft_hip_multi_stft_slice(h_, len, win_len, hop, frames, rank, &f, &c);
        return {f, c};
    }
    void synchronize() { (void)st(kofft_hip_multi_synchronize(h_)); }
    struct Timing { float upload_ms = 0, kernel_ms = 0, gather_ms = 0, download_ms = 0, wall_ms = 0; };
    Timing last_timing() const
    {
        Timing t;
        kofft_hip_multi_last_timing_ex(h_, &t.upload_ms, &t.kernel_ms, &t.gather_ms, &t.download_ms, &t.wall_ms);
        return t;
    }
    // the exchange form: KOFFT_MULTI_GATHER_RCCL (grouped ncclAllGather) or KOFFT_MULTI_GATHER_DIRECT (peer copies on per-peer streams)
    Result set_gather(int mode) { return st(kofft_hip_multi_set_gather(h_, mode)); }
    int last_gather() const
    {
        int last = 0;
        kofft_hip_multi_gather_mode(h_, nullptr, &last);
        return last;
    }
    kofft_hip_multi *raw() const { return h_; }

private:
    Result st(int rc) const
    {
        if (rc > 0) return Result::Err(static_cast<FftError>(rc));
        if (rc != 0) throw DeviceError(rc, kofft_hip_multi_last_error(h_));
        return Result::Ok();
    }
    kofft_hip_multi *h_ = nullptr;
};

// stft::parallel (stft.rs:232-263): only hop == 0 is rejected
inline Result parallel(const std::vector<float> &signal, const std::vector<float> &window, size_t hop_size,
                       std::vector<std::vector<Complex32>> &output, const HipFftImpl<float> &fft)
{
    return stft(signal, window, hop_size, output, fft, false);
}
// stft::frame (stft.rs:355-372)
inline Result frame(const std::vector<float> &signal, const std::vector<float> &window, size_t start,
                    std::vector<Complex32> &frame_out, const HipFftImpl<float> &fft)
{
    if (frame_out.size() < window.size()) throw std::out_of_range("frame_out shorter than the window (the reference panics)");
    return fft.st(kofft_hip_stft_frame_f32(fft.raw(), signal.data(), signal.size(), window.data(), window.size(), start,
                                           reinterpret_cast<float *>(frame_out.data())));
}

class StftStream {  // stft.rs:160-206
public:
    static Result create(const std::vector<float> &signal, const std::vector<float> &window, size_t hop_size,
                         const HipFftImpl<float> &fft, StftStream *&out)
    {
        if (hop_size == 0) return Result::Err(FftError::InvalidHopSize);
        out = new StftStream(signal, window, hop_size, fft);
        return Result::Ok();
    }
    // Ok(true) -> *more = true
    Result next_frame(std::vector<Complex32> &out, bool &more)
    {
        more = false;
        if (out.size() != window_.size()) return Result::Err(FftError::MismatchedLengths);
        if (pos_ >= signal_.size()) return Result::Ok();
        Result r = frame(signal_, window_, pos_, out, fft_);
        if (r.is_err()) return r;
        pos_ += hop_;
        more = true;
        return Result::Ok();
    }

private:
    StftStream(const std::vector<float> &s, const std::vector<float> &w, size_t hop, const HipFftImpl<float> &f)
        : signal_(s), window_(w), hop_(hop), pos_(0), fft_(f)
    {
    }
    const std::vector<float> &signal_;
    const std::vector<float> &window_;
    size_t hop_, pos_;
    const HipFftImpl<float> &fft_;
};

// stft::IstftStream (stft.rs:401-518): streaming overlap-add with normalisation.  push_frame hands the frame to the
// device (ifft + windowed add into the ring buffer = inverse_frame's arithmetic); the window-square bookkeeping and the
// "> 1e-8" division of the samples that leave the buffer are host bookkeeping, as in the reference's helper.
class IstftStream {
public:
    static Result create(size_t win_len, size_t hop, std::vector<float> window, const HipFftImpl<float> &fft, IstftStream *&out)
    {
        if (hop == 0) return Result::Err(FftError::InvalidHopSize);
        out = new IstftStream(win_len, hop, std::move(window), fft);
        return Result::Ok();
    }
    // the next `hop` samples of the reconstructed signal
    Result push_frame(const std::vector<Complex32> &frame_in, std::vector<float> &out)
    {
        if (frame_in.size() != win_len_) return Result::Err(FftError::MismatchedLengths);
        if (window_.size() < win_len_) throw std::out_of_range("window shorter than win_len (the reference panics)");
        time_buf_ = frame_in;
        Result r = fft_.st(kofft_hip_istft_frame_f32(fft_.raw(), reinterpret_cast<float *>(time_buf_.data()), window_.data(), win_len_,
                                                     0, buffer_.data() + buf_pos_, win_len_));
        if (r.is_err()) return r;
        for (size_t i = 0; i < win_len_; ++i) norm_buf_[buf_pos_ + i] += window_[i] * window_[i];
        ++frame_count_;
        normalise(out_pos_, out_pos_ + hop_, out);
        out_pos_ += hop_;
        buf_pos_ += hop_;
        if (buf_pos_ + win_len_ > buffer_.size()) {
            buffer_.resize(buf_pos_ + win_len_, 0.0f);
            norm_buf_.resize(buf_pos_ + win_len_, 0.0f);
        }
        for (size_t i = 0; i < hop_; ++i) {
            buffer_[buf_pos_ + win_len_ - hop_ + i] = 0.0f;
            norm_buf_[buf_pos_ + win_len_ - hop_ + i] = 0.0f;
        }
        return Result::Ok();
    }
    // the remaining win_len - hop samples after the last frame (empty before the first frame and on later calls)
    void flush(std::vector<float> &out)
    {
        out.clear();
        if (frame_count_ == 0) return;
        const size_t lo = out_pos_, hi = buf_pos_ + win_len_ - hop_;
        if (lo >= hi) return;
        normalise(lo, hi, out);
        out_pos_ = hi;
    }

private:
    IstftStream(size_t win_len, size_t hop, std::vector<float> window, const HipFftImpl<float> &fft)
        : win_len_(win_len), hop_(hop), window_(std::move(window)), fft_(fft), buffer_(win_len + 2 * hop, 0.0f),
          norm_buf_(win_len + 2 * hop, 0.0f), time_buf_(win_len)
    {
    }
    void normalise(size_t lo, size_t hi, std::vector<float> &out)
    {
        out.resize(hi - lo);
        for (size_t i = lo; i < hi; ++i) {
            if (norm_buf_[i] > 1e-8f) buffer_[i] /= norm_buf_[i];
            norm_buf_[i] = 0.0f;
            out[i - lo] = buffer_[i];
        }
    }
    size_t win_len_, hop_;
    std::vector<float> window_;
    const HipFftImpl<float> &fft_;
    std::vector<float> buffer_, norm_buf_;
    std::vector<Complex32> time_buf_;
    size_t buf_pos_ = 0, out_pos_ = 0, frame_count_ = 0;
};

}  // namespace kofft
