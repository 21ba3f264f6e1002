// fft_wg.hip.h -- the generic "one transform per thread group" Stockham kernel.
//
// A transform of N = 2^L points is owned by TPT = N/R threads (R = 2^RL points per
// thread); a workgroup of BLOCK threads carries BLOCK/TPT transforms.  The L radix-2
// stages run as ceil(L/RL) register passes of up to RL stages each (reg_pass), with the
// data exchanged through LDS between passes.  The first pass reads through the IO
// policy (HBM, with the caller's fused pre-processing) and the last pass writes through
// it, so each element crosses HBM exactly once in each direction.
//
// LDS layout: element i of a transform lives at lds_pad(i) = i + (i >> 4) (one pad
// element per 16), which makes both access shapes of the exchanges conflict-free for
// ds_read_b64 / ds_write_b64 at N = 4096 and N = 1024 ("LDS exchange").
#pragma once

#include "fft_device.hip.h"

namespace kofft {

__host__ __device__ constexpr int lds_pad(int i) { return i + (i >> 4); }
__host__ __device__ constexpr int lds_elems(int n) { return n + (n >> 4) + 1; }

// ---- IO policies -----------------------------------------------------------------
// load(xf, i)  : element i of transform xf as the FFT must see it.
// store(xf, o, v): element o of the finished transform.

struct NoInv {};

// Optional per-thread accumulator a policy may carry across its stores (StftMagIO: the running maximum).  Policies
// without one cost nothing: the kernels test io_has_acc<IO> at compile time.
template <class IO, class = void>
struct io_has_acc { static constexpr bool value = false; };
template <class IO>
struct io_has_acc<IO, decltype((void)IO::kHasAcc)> { static constexpr bool value = IO::kHasAcc; };
// policies whose range test can be hoisted out of the per-sample loop (StftIO and its heirs: frame_rem / finish_rem)
template <class IO, class = void>
struct io_frame_rem { static constexpr bool value = false; };
template <class IO>
struct io_frame_rem<IO, decltype((void)IO::kFrameRem)> { static constexpr bool value = IO::kFrameRem; };
// policies that keep bins 0 .. n/2-1 only and offer mags_of / store_d_mag (StftMagIO)
template <class IO, class = void>
struct io_half_spectrum { static constexpr bool value = false; };
template <class IO>
struct io_half_spectrum<IO, decltype((void)IO::kHalfSpectrum)> { static constexpr bool value = IO::kHalfSpectrum; };

// A policy whose input value k needs row elements k AND m - k (irfft) can get the second one from the lane that loaded it
// instead of loading it again, when a transform's threads share a wavefront: io_pairs_in_wave<IO> (see IrfftIO).
template <class IO, class = void>
struct io_stages_pairs { static constexpr bool value = false; };
template <class IO>
struct io_stages_pairs<IO, decltype((void)IO::kStagePairs)> { static constexpr bool value = IO::kStagePairs; };
template <class IO, class = void>
struct io_pairs_in_wave { static constexpr bool value = false; };
template <class IO>
struct io_pairs_in_wave<IO, decltype((void)IO::kPairInWave)> { static constexpr bool value = IO::kPairInWave; };

// Default for every IO policy of a stand-alone transform: plain table indexing, threads of a transform contiguous.
struct PlainTw {
    static constexpr bool kSlotMinor = false;
    static constexpr bool kPairXcd = false;
    static constexpr bool kSplitLds = false;
    static constexpr int kMinWaves = 1;  // __launch_bounds__ second argument (waves per SIMD the kernel must fit)
    static constexpr int kPersistMaxLog2 = 13;  // largest transform the persistent kernel is built for with this policy
    // smallest one it is USED for (measured against the generic kernel, one box: complex n = 128 / 256 lose 5 %, STFT
    // gains 22 % / 6 % (n = 64: 20 %), irfft 16 % / 29 % (m = 64: 5 %), rfft n = 128 .. 512 gains 4 .. 10 % once its grid
    // is halved -- PersistGrid in kofft_hip.hip)
    static constexpr int kPersistMinLog2 = 9;
    // dispatch() is reached with n = 1 (rfft of two reals, a one-sample STFT window); policies whose callers return before it say false
    static constexpr bool kLen1 = true;
    __host__ __device__ bool group_rows_ok() const { return true; }  // per-lane row offsets fit 32 bits
    __device__ __forceinline__ TwPlain tw_map(size_t) const { return {}; }
};

#ifndef KOFFT_STFT_SMALL_BLOCK32
#define KOFFT_STFT_SMALL_BLOCK32 64
#endif
#ifndef KOFFT_SMALL_BLOCK
#define KOFFT_SMALL_BLOCK 256
#endif
#ifndef KOFFT_C64_SMALL_BLOCK
#define KOFFT_C64_SMALL_BLOCK 64
#endif
#ifndef KOFFT_C_LOAD_AUX
#define KOFFT_C_LOAD_AUX AUX_NT  // cache policy of the complex kernels' descriptor loads (A/B hook: tools/exp_inplace_ab.py)
#endif
// FftImpl::fft (fft.rs:1054) / ifft (fft.rs:1134-1174: conj, fft, conj, *scale).
template <typename T, bool INVERSE>
struct ComplexIO : PlainTw {
    static constexpr int kSmallBlock16 = sizeof(T) == 8 ? KOFFT_C64_SMALL_BLOCK : KOFFT_SMALL_BLOCK;  // fft_small_kernel, n = 16
    static constexpr bool kSplitOk = true;  // fft_split.hip.h
    static constexpr bool kStreams = true;  // descriptor loads in the generic kernels
    static constexpr bool kPersist = true;  // eligible for the persistent prefetching kernel
    static constexpr bool kInvInLds = false;
    static constexpr bool kLeanRegisters = true;
    static constexpr bool kLen1 = false;  // fft_dev copies a one-point transform (fft.rs:1059)
    using Raw = cpx<T>;
    using Inv = NoInv;
    const cpx<T> *__restrict__ in;
    cpx<T> *__restrict__ out;
    int n;
    T scale;  // 1 / (n as f32 as T), fft.rs:1167
    __device__ __forceinline__ Raw fetch(size_t xf, int i) const { return ld_stream(in + xf * (size_t)n + i); }
    // descriptor forms (xf wave-uniform): element iu + lane of transform xf, iu a compile-time constant
    static constexpr int kRawBytes = sizeof(cpx<T>);
    __device__ __forceinline__ rsrc_t in_desc(size_t xf, bool valid) const
    {
        return make_rsrc(in + (valid ? xf : 0) * (size_t)n, valid ? (unsigned)n * sizeof(cpx<T>) : 0u);
    }
    __device__ __forceinline__ rsrc_t out_desc(size_t xf) const { return make_rsrc(out + xf * (size_t)n, (unsigned)n * sizeof(cpx<T>)); }
    // group forms (persistent kernel, transforms smaller than a wavefront): cnt consecutive transforms from xf0;
    // lane offset = row_off (sub-slot * row bytes) + the per-transform offsets of the forms above
    __device__ __forceinline__ rsrc_t in_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(in + (cnt > 0 ? xf0 : 0) * (size_t)n, (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)n * (unsigned)sizeof(cpx<T>));
    }
    __device__ __forceinline__ rsrc_t out_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(out + (cnt > 0 ? xf0 : 0) * (size_t)n, (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)n * (unsigned)sizeof(cpx<T>));
    }
    __device__ __forceinline__ unsigned out_row_bytes() const { return (unsigned)n * sizeof(cpx<T>); }
    // workgroup forms (fft_wg_kernel, several transforms per wave): one descriptor over the xpb consecutive
    // transforms starting at xf0, cut at the end of the batch; lane offset = slot * in_slot_bytes() + element.
    __device__ __forceinline__ bool wg_desc_ok(int) const { return true; }
    __device__ __forceinline__ unsigned in_slot_bytes() const { return (unsigned)n * sizeof(cpx<T>); }
    __device__ __forceinline__ rsrc_t in_desc_wg(size_t xf0, size_t batch, int xpb) const
    {
        const size_t cnt = xf0 < batch ? (batch - xf0 < (size_t)xpb ? batch - xf0 : (size_t)xpb) : 0;
        return make_rsrc(in + (cnt ? xf0 : 0) * (size_t)n, (unsigned)(cnt * n * sizeof(cpx<T>)));
    }
    __device__ __forceinline__ Raw fetch_d(rsrc_t d, int lane_bytes, int iu, int row_off = 0) const
    {
        return buf_load_cpx<T, KOFFT_C_LOAD_AUX>(d, row_off + lane_bytes, iu * (int)sizeof(cpx<T>));
    }
    __device__ __forceinline__ void store_d(rsrc_t d, int lane_bytes, int ou, cpx<T> v, int row_off = 0) const
    {
        if (INVERSE) {
            const T im = -v.im;
            v.re = v.re * scale;
            v.im = im * scale;
        }
        buf_store_cpx<T>(v, d, row_off + lane_bytes, ou * (int)sizeof(cpx<T>));
    }
    __device__ __forceinline__ Inv invariant(int) const { return {}; }
    __device__ __forceinline__ cpx<T> finish(size_t, int, Raw v, Inv) const
    {
        if (INVERSE) v.im = -v.im;
        return v;
    }
    __device__ __forceinline__ bool inside(size_t) const { return true; }
    __device__ __forceinline__ cpx<T> finish_in(Raw v, Inv w) const { return finish(0, 0, v, w); }
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const { return finish(xf, i, fetch(xf, i), {}); }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        if (INVERSE) {
            const T im = -v.im;
            v.re = v.re * scale;
            v.im = im * scale;
        }
        st_stream(out + xf * (size_t)n + o, v);
    }
};

// stft.rs:91-103: frame f starts at start0 + f*hop; x = signal[start+i]*window[i] or 0.
struct StftIO : PlainTw {
    static constexpr int kSmallBlock32 = KOFFT_STFT_SMALL_BLOCK32;  // fft_small_kernel at n = 32 (StftMagIO inherits it)
    static constexpr bool kSplitOk = true;  // fft_split.hip.h
    static constexpr bool kStreams = true;
    static constexpr bool kPersist = true;
    static constexpr int kPersistMinLog2 = 6;
    static constexpr bool kInvInLds = false;  // 16 window samples per thread: registers (LDS staging + 3 waves/SIMD measured slower)
    static constexpr bool kLeanRegisters = false;
    using Raw = float;
    using Inv = float;
    const float *__restrict__ signal;
    const float *__restrict__ window;
    cpx<float> *__restrict__ out;
    size_t len, hop, start0;
    int n;
    __host__ __device__ bool group_rows_ok() const { return hop <= (size_t(1) << 24); }
    __device__ __forceinline__ bool in_range(size_t xf, int i) const { return start0 + xf * hop + (size_t)i < len; }
    __device__ __forceinline__ Raw fetch(size_t xf, int i) const
    {
        return in_range(xf, i) ? signal[start0 + xf * hop + (size_t)i] : 0.0f;
    }
    // descriptor forms: the frame's descriptor covers only the samples that exist, so a frame that runs off the
    // end of the signal reads zeros from the bounds check (stft.rs:95-99) without a per-lane address test.
    static constexpr int kRawBytes = sizeof(float);
    __device__ __forceinline__ rsrc_t in_desc(size_t xf, bool valid) const
    {
#ifdef KOFFT_EXP_STFT_SAMEFRAME /* measurement only (wrong results): every frame reads frame 0's samples -- the bound on what staging the samples could gain */
        xf = 0;
#endif
        const size_t start = start0 + xf * hop;
        const size_t avail = (valid && start < len) ? len - start : 0;
        return make_rsrc(signal + (avail ? start : 0), (unsigned)(avail < (size_t)n ? avail : (size_t)n) * 4u);
    }
    __device__ __forceinline__ rsrc_t out_desc(size_t xf) const { return make_rsrc(out + xf * (size_t)n, (unsigned)n * 8u); }
    // workgroup forms: the descriptor starts at frame xf0's first sample and ends with the signal (or 4 GiB
    // earlier); frames past `batch` read real samples or zeros and are never stored.  32-bit lane offsets
    // bound the hop this form can serve.
    __device__ __forceinline__ bool wg_desc_ok(int xpb) const { return hop <= (size_t(1) << 22) && (size_t)xpb * hop < (size_t(1) << 28); }
    __device__ __forceinline__ unsigned in_slot_bytes() const { return (unsigned)hop * 4u; }
    __device__ __forceinline__ rsrc_t in_desc_wg(size_t xf0, size_t, int) const
    {
#ifdef KOFFT_EXP_STFT_SAMEFRAME
        xf0 = 0;
#endif
        const size_t start = start0 + xf0 * hop;
        size_t avail = start < len ? len - start : 0;
        if (avail > 0x3fffffffULL) avail = 0x3fffffffULL;
        return make_rsrc(signal + (avail ? start : 0), (unsigned)avail * 4u);
    }
    __device__ __forceinline__ Raw fetch_d(rsrc_t d, int lane_bytes, int iu, int row_off = 0) const
    {
        return buf_load_f32<AUX_DEFAULT>(d, row_off + lane_bytes, iu * 4);
    }
    __device__ __forceinline__ void store_d(rsrc_t d, int lane_bytes, int ou, cpx<float> v, int row_off = 0) const
    {
        buf_store_cpx<float>(v, d, row_off + lane_bytes, ou * 8);
    }
    // group forms: cnt consecutive frames from xf0 (frame f of the group starts f*hop samples in); the descriptor ends
    // with the signal, so frames that run off its end read zeros (stft.rs:95-99)
    __device__ __forceinline__ rsrc_t in_desc_n(size_t xf0, int cnt) const
    {
#ifdef KOFFT_EXP_STFT_SAMEFRAME
        xf0 = 0;
#endif
        const size_t start = start0 + xf0 * hop;
        size_t avail = (cnt > 0 && start < len) ? len - start : 0;
        const size_t span = cnt > 0 ? (size_t)(cnt - 1) * hop + (size_t)n : 0;
        if (avail > span) avail = span;
        if (avail > 0x3fffffffULL) avail = 0x3fffffffULL;
        return make_rsrc(signal + (avail ? start : 0), (unsigned)avail * 4u);
    }
    __device__ __forceinline__ rsrc_t out_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(out + (cnt > 0 ? xf0 : 0) * (size_t)n, (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)n * 8u);
    }
    __device__ __forceinline__ unsigned out_row_bytes() const { return (unsigned)n * 8u; }
    __device__ __forceinline__ Inv invariant(int i) const { return window[i]; }
    __device__ __forceinline__ cpx<float> finish(size_t xf, int i, Raw x, Inv w) const
    {
        return mk<float>(in_range(xf, i) ? x * w : 0.0f, 0.0f);  // past the end: exactly +0, whatever the window holds
    }
    // The same test with the frame's sample count worked out ONCE (round 5): rem = samples of frame xf that exist, 0 .. n; sample i is
    // in range iff i < rem -- one 32-bit compare per sample instead of a 64-bit add and compare (and their registers) per sample.
    static constexpr bool kFrameRem = true;
    __device__ __forceinline__ int frame_rem(size_t xf) const
    {
        const size_t start = start0 + xf * hop;
        const size_t avail = start < len ? len - start : 0;
        return (int)(avail < (size_t)n ? avail : (size_t)n);
    }
    __device__ __forceinline__ cpx<float> finish_rem(int i, Raw x, Inv w, int rem) const
    {
        return mk<float>(i < rem ? x * w : 0.0f, 0.0f);  // past the end: exactly +0, whatever the window holds
    }
    // a frame that lies wholly inside the signal needs no per-sample range test (4 VALU instructions per sample)
    __device__ __forceinline__ bool inside(size_t xf) const { return start0 + xf * hop + (size_t)n <= len; }
    __device__ __forceinline__ cpx<float> finish_in(Raw x, Inv w) const { return mk<float>(x * w, 0.0f); }
    __device__ __forceinline__ cpx<float> load(size_t xf, int i) const { return finish(xf, i, fetch(xf, i), invariant(i)); }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<float> v) const
    {
        st_stream(out + xf * (size_t)n + o, v);
    }
};

// visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76), SURVEY 8f row 2: the STFT of stft.rs:91-103 with
// only bins 0 .. n/2-1 kept, as magnitudes sqrt(re*re + im*im) in f32 (un-fused, correctly rounded sqrt).  Fusing the
// magnitude into the store cuts the output from 8 B x n to 4 B x n/2 per frame.  The maximum is a separate reduction.
struct StftMagIO : StftIO {
    float *__restrict__ mags;  // frames x n/2
    unsigned *__restrict__ max_bits;  // the maximum magnitude, as the bit pattern of a non-negative f32 (zeroed by the caller)
    // The maximum rides on the stores (`if mag > max_mag`, spectrogram.rs:68-70: a NaN is never selected).  Round 6: what every thread
    // keeps is the largest SUM OF SQUARES it has rooted -- a correctly rounded square root is monotone, so the largest magnitude IS the root of
    // the largest sum, bit for bit (fmaxf skips NaNs like the reference's comparison; sums are >= +0) -- and the one root is taken when the
    // kernel is done: each wavefront reduces its lanes and issues ONE atomicMax.  Non-negative floats order like their bit patterns, so the
    // unsigned maximum is exact whatever the order of arrival; no second pass over the magnitudes.
    static constexpr bool kHasAcc = true;
    using Acc = float;
    __device__ __forceinline__ Acc acc_init() const { return 0.0f; }
    __device__ __forceinline__ void acc_finish(Acc s) const
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s = __builtin_fmaxf(s, __shfl_xor(s, off));
        const float m = sqrtf(s);
        if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(max_bits, __builtin_bit_cast(unsigned, m));
    }
    // sqrtf is correctly rounded here (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is not
    __device__ __forceinline__ static float sumsq(cpx<float> c) { return c.re * c.re + c.im * c.im; }
    __device__ __forceinline__ void store_acc(size_t xf, int o, cpx<float> v, Acc &acc) const
    {
        if (o < n / 2) {
            const float s = sumsq(v);
            st_stream(mags + xf * (size_t)(n / 2) + o, sqrtf(s));
            acc = __builtin_fmaxf(acc, s);
        }
    }
    __device__ __forceinline__ rsrc_t out_desc(size_t xf) const { return make_rsrc(mags + xf * (size_t)(n / 2), (unsigned)(n / 2) * 4u); }
    __device__ __forceinline__ void store_d_acc(rsrc_t d, int lane_bytes, int ou, cpx<float> v, int row_off, Acc &acc) const
    {
        // lane_bytes = 8 * tau (complex offset); the magnitude row has 4-byte elements
        if (ou + (lane_bytes >> 3) < n / 2) {
            const float s = sumsq(v);
            buf_store_f32(sqrtf(s), d, row_off + (lane_bytes >> 1), ou * 4);
            acc = __builtin_fmaxf(acc, s);
        }
    }
    // (round 5) Which REGISTERS hold bins below n/2 is a compile-time fact in the persistent kernel (fft_persist.hip.h: the register part ou of
    // the last pass's output index and the thread part tau are disjoint bit fields, so ou + tau < n/2 <=> ou < n/2, a constant per register): no
    // range test per store -- 16 runtime branches less per transform -- and the discarded half of the last stage's butterflies is dead code.
    static constexpr bool kHalfSpectrum = true;
    // Round 6: the NK kept magnitudes of one transform together.  sqrtf's expansion costs ~16 VALU instructions per
    // root -- 3 to scale arguments below 2^-96 up by 2^32, v_sqrt_f32, 8 for the +-1 ulp correction (two fma residuals against the
    // neighbours of the estimate), 2 to scale back, 2 to pass +-0 / +inf through -- and the kernel is VALU-bound (0.69 of the issue rate at
    // 0.26 of HBM's).  The guard and the pass-through are decided ONCE per transform and wavefront here: when every sum of squares of the
    // wavefront is a normal number >= 2^-96 or a NaN (min3 / max3 skip NaNs, and a NaN takes the same way through both forms), the roots
    // are v_sqrt_f32 + the two-sided correction alone; otherwise (silence: zeros; subnormal or infinite sums) sqrtf as before.  Any
    // correctly rounded root has the same bits: tools/ubench_sqrt.hip compares the short form with sqrtf on EVERY f32 in [2^-96, inf)
    // (1 879 048 192 values, 0 mismatches; v_sqrt_f32 alone is 1 ulp low on 15 % and 1 ulp high on 0.006 % of them -- both corrections are needed).
    __device__ __forceinline__ static float sqrt_cr_normal(const float x)
    {
        const float r = __builtin_amdgcn_sqrtf(x);
        const float dn = __builtin_bit_cast(float, __builtin_bit_cast(int, r) - 1), up = __builtin_bit_cast(float, __builtin_bit_cast(int, r) + 1);
        float y = r;
        if (__builtin_fmaf(-dn, r, x) <= 0.0f) y = dn;
        if (__builtin_fmaf(-up, r, x) > 0.0f) y = up;
        return y;
    }
    // returns the largest of the NK sums of squares (NaNs skipped): what the accumulator keeps
    template <int NK>
    __device__ __forceinline__ static float mags_of(const cpx<float> (&v)[NK], float (&m)[NK])
    {
        float s[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) s[i] = v[i].re * v[i].re + v[i].im * v[i].im;
        float lo = s[0], hi = s[0];
#pragma unroll
        for (int i = 1; i < NK; ++i) {
            lo = __builtin_fminf(lo, s[i]);
            hi = __builtin_fmaxf(hi, s[i]);
        }
#ifdef KOFFT_MAG_SQRTF /* measurement only (tools/build_variant.sh): round 5's roots, for same-box A/Bs */
        const bool plain = false;
#else
        const bool plain = lo >= 0x1p-96f && hi < __builtin_inff();  // (all NaN: false)
#endif
        if (__builtin_amdgcn_ballot_w64(!plain) == 0) {  // wave-uniform
#pragma unroll
            for (int i = 0; i < NK; ++i) m[i] = sqrt_cr_normal(s[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NK; ++i) m[i] = sqrtf(s[i]);
        }
        return hi;
    }
    __device__ __forceinline__ void store_d_mag(rsrc_t d, int lane_bytes, int ou, float m, int row_off) const
    {
        buf_store_f32(m, d, row_off + (lane_bytes >> 1), ou * 4);
    }
    __device__ __forceinline__ rsrc_t out_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(mags + (cnt > 0 ? xf0 : 0) * (size_t)(n / 2), (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)(n / 2) * 4u);
    }
    __device__ __forceinline__ unsigned out_row_bytes() const { return (unsigned)(n / 2) * 4u; }
};

// rfft.rs:444-446 pack z[i] = (x[2i], x[2i+1]) (with the optional row window of the
// batched entry point); the post-pass of rfft.rs:450-463 runs in the kernel epilogue,
// which writes the m+1 outputs.
template <typename T>
struct RfftIO : PlainTw {
    static constexpr bool kStreams = true;
    static constexpr bool kPersist = true;
    static constexpr int kPersistMinLog2 = 6;
    static constexpr bool kInvInLds = true;  // window pairs + post-pass table: staged in LDS once per workgroup
    static constexpr bool kLeanRegisters = false;
    using Raw = cpx<T>;
    using Inv = cpx<T>;
    const T *__restrict__ in;         // batch rows of 2*m reals
    const T *__restrict__ window;     // 2*m reals or nullptr
    cpx<T> *__restrict__ out;         // batch rows of m+1 complex
    const cpx<T> *__restrict__ rtab;  // build_twiddle_table(m), rfft.rs:172-183
    int m;
    __device__ __forceinline__ Raw fetch(size_t xf, int i) const
    {
        return ld_stream(reinterpret_cast<const cpx<T> *>(in + xf * (size_t)(2 * m)) + i);
    }
    static constexpr int kRawBytes = sizeof(cpx<T>);
    __device__ __forceinline__ rsrc_t in_desc(size_t xf, bool valid) const
    {
        return make_rsrc(in + (valid ? xf : 0) * (size_t)(2 * m), valid ? (unsigned)m * sizeof(cpx<T>) : 0u);
    }
    __device__ __forceinline__ rsrc_t out_desc(size_t xf) const
    {
        return make_rsrc(out + xf * (size_t)(m + 1), (unsigned)(m + 1) * sizeof(cpx<T>));
    }
    // the same row seen from `back` elements earlier (the persistent epilogue never touches those first elements)
    __device__ __forceinline__ rsrc_t out_desc_back(size_t xf, int back) const
    {
        return make_rsrc(out + xf * (size_t)(m + 1) - back, (unsigned)(m + 1 + back) * sizeof(cpx<T>));
    }
    // elements between the start of output row xf and the previous 128-byte line boundary (modulo the line)
    __device__ __forceinline__ int row_misalign(size_t xf) const
    {
        return (int)((reinterpret_cast<size_t>(out) / sizeof(cpx<T>) + xf * (size_t)(m + 1)) & 127);
    }
    __device__ __forceinline__ bool wg_desc_ok(int) const { return true; }
    __device__ __forceinline__ unsigned in_slot_bytes() const { return (unsigned)m * sizeof(cpx<T>); }
    __device__ __forceinline__ rsrc_t in_desc_wg(size_t xf0, size_t batch, int xpb) const
    {
        const size_t cnt = xf0 < batch ? (batch - xf0 < (size_t)xpb ? batch - xf0 : (size_t)xpb) : 0;
        return make_rsrc(in + (cnt ? xf0 : 0) * (size_t)(2 * m), (unsigned)(cnt * m * sizeof(cpx<T>)));
    }
    __device__ __forceinline__ Raw fetch_d(rsrc_t d, int lane_bytes, int iu, int row_off = 0) const
    {
        return buf_load_cpx<T, AUX_NT>(d, row_off + lane_bytes, iu * (int)sizeof(cpx<T>));
    }
    __device__ __forceinline__ void store_d(rsrc_t d, int lane_bytes, int ou, cpx<T> v, int row_off = 0) const
    {
        buf_store_cpx<T>(v, d, row_off + lane_bytes, ou * (int)sizeof(cpx<T>));
    }
    // group forms: cnt consecutive rows from xf0
    __device__ __forceinline__ rsrc_t in_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(in + (cnt > 0 ? xf0 : 0) * (size_t)(2 * m), (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)m * (unsigned)sizeof(cpx<T>));
    }
    __device__ __forceinline__ rsrc_t out_desc_back_n(size_t xf0, int cnt, int back) const
    {
        return make_rsrc(out + xf0 * (size_t)(m + 1) - back, ((unsigned)cnt * (unsigned)(m + 1) + (unsigned)back) * (unsigned)sizeof(cpx<T>));
    }
    __device__ __forceinline__ unsigned out_row_bytes() const { return (unsigned)(m + 1) * sizeof(cpx<T>); }
    // no window: multiply by exactly 1, which leaves every value unchanged bit for bit
    __device__ __forceinline__ Inv invariant(int i) const
    {
        return window ? reinterpret_cast<const cpx<T> *>(window)[i] : mk<T>(T(1), T(1));
    }
    __device__ __forceinline__ cpx<T> finish(size_t, int, Raw v, Inv w) const { return mk<T>(v.re * w.re, v.im * w.im); }
    __device__ __forceinline__ bool inside(size_t) const { return true; }
    __device__ __forceinline__ cpx<T> finish_in(Raw v, Inv w) const { return finish(0, 0, v, w); }
    __device__ __forceinline__ cpx<T> load(size_t xf, int i) const { return finish(xf, i, fetch(xf, i), invariant(i)); }
    // X[k] for 1 <= k < m from Y[k], Y[m-k] and W[k]  (rfft.rs:454-463)
    __device__ __forceinline__ cpx<T> post_w(cpx<T> w, cpx<T> a, cpx<T> ymk) const
    {
#ifndef KOFFT_BFLY_NOASM
        if constexpr (sizeof(T) == 4) {
            const v2f wv = {w.re, w.im}, av = {a.re, a.im}, yv = {ymk.re, ymk.im};
            const v2f x = rfft_post_f32_pk(wv, av, yv);
            return mk<T>(x.x, x.y);
        }
#endif
        const T half = T(0.5f);
        const cpx<T> b = mk<T>(ymk.re, -ymk.im);
        const cpx<T> sum = cadd(a, b), diff = csub(a, b);
        const cpx<T> t = cmul(w, diff);
        const cpx<T> temp = cadd(sum, mk<T>(t.im, -t.re));
        return mk<T>(temp.re * half, temp.im * half);
    }
    __device__ __forceinline__ cpx<T> post(int k, cpx<T> a, cpx<T> ymk) const { return post_w(rtab[k], a, ymk); }
};

// irfft_direct (rfft.rs:487-506): scratch[k] from input[k], input[m-k]; then fft.ifft
// (conj, fft, conj, *1/m); output[2i], output[2i+1] = scratch[i].re, .im.
template <typename T>
struct IrfftIO : PlainTw {
    static constexpr bool kStreams = false;  // generic kernels: per-element loads (two row elements + a table entry each)
    static constexpr bool kPersist = sizeof(T) == 4;  // persistent kernel: both row elements prefetched, table in LDS
    static constexpr int kPersistMaxLog2 = 12;        // up to m = 4096 (paired input: 34 prefetch registers per thread)
    static constexpr int kPersistMinLog2 = 6;
    static constexpr bool kInvInLds = true;
    static constexpr bool kLeanRegisters = false;
    struct Raw { cpx<T> a, rb; };  // input[k], input[m-k]
    // Transforms whose threads share a wavefront (m <= 64 points per thread-group of at most 64 lanes): thread tau holds
    // k = tau + tpt*u, and m - k = (tpt - tau) + tpt*(R-1-u) is register R-1-u of lane tpt - tau -- one ds_bpermute per
    // word instead of a second load (lane 0 pairs with itself: register R-u, and input[m] for u = 0, loaded once per
    // transform as register R of the set).  Halves the load instructions and the prefetch registers.
    static constexpr bool kPairInWave = sizeof(T) == 4;
    static constexpr bool kStagePairs = true;  // generic kernel: the row goes through LDS once instead of being loaded twice
    __device__ __forceinline__ cpx<T> load_a(size_t xf, int k) const { return ld_stream(in + xf * (size_t)(m + 1) + k); }
    __device__ __forceinline__ cpx<T> load_raw(size_t e) const { return ld_stream(in + e); }  // element e of the whole input, rows back to back
    __device__ __forceinline__ cpx<T> pre_staged(int k, cpx<T> a, cpx<T> rb) const { return pre(k, a, rb, rtab[k]); }
    using RawPair = cpx<T>;
    __device__ __forceinline__ RawPair fetch_pair_d(rsrc_t d, int lane_bytes, int iu, int row_off) const
    {
        return buf_load_cpx<T, AUX_NT>(d, row_off + lane_bytes, iu * (int)sizeof(cpx<T>));
    }
    __device__ __forceinline__ RawPair fetch_last_d(rsrc_t d, int row_off) const  // input[m]: the same address for a transform's lanes
    {
        return buf_load_cpx<T, AUX_NT>(d, row_off, m * (int)sizeof(cpx<T>));
    }
    // cur[u] = scratch[k] for this thread's R values; raw[0..R-1] = input[tau + tpt*u], raw[R] = input[m]
    template <int R, int TPT, class InvAt>
    __device__ __forceinline__ void finish_pairs(const RawPair *raw, const int tau, cpx<T> *cur, const InvAt inv_at) const
    {
        static_assert(sizeof(T) == 4, "pairing by ds_bpermute is written for c32");
        const int lane = (int)(threadIdx.x & 63);
        const int partner = ((lane & ~(TPT - 1)) | ((TPT - tau) & (TPT - 1))) << 2;  // byte address of the source lane
        const bool self = tau == 0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const cpx<T> src = raw[R - 1 - u];
            cpx<T> rb;
            rb.re = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner, __builtin_bit_cast(int, src.re)));
            rb.im = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner, __builtin_bit_cast(int, src.im)));
            const cpx<T> own = raw[R - u];  // lane 0: m - k = tpt*(R-u) is its own register R-u (u = 0: input[m])
            if (self) rb = own;
            // k == 0 only for (tau == 0, u == 0): pre()'s select on k is compile-time false for u > 0
            cur[u] = pre(u == 0 ? tau : 1, raw[u], rb, inv_at(u));
        }
    }
    using Inv = cpx<T>;            // W[k] of build_twiddle_table(m)
    const cpx<T> *__restrict__ in;  // batch rows of m+1 complex
    cpx<T> *__restrict__ out;       // batch rows of m complex == 2*m reals
    const cpx<T> *__restrict__ rtab;
    int m;
    T scale;  // 1 / (m as f32 as T)
    int tpt;  // threads per transform of the persistent launch (reversed lane index for input[m-k])
    // ---- descriptor forms (persistent kernel)
    static constexpr int kRawBytes = sizeof(cpx<T>);
    __device__ __forceinline__ rsrc_t in_desc(size_t xf, bool valid) const
    {
        return make_rsrc(in + (valid ? xf : 0) * (size_t)(m + 1), valid ? (unsigned)(m + 1) * sizeof(cpx<T>) : 0u);
    }
    __device__ __forceinline__ rsrc_t out_desc(size_t xf) const { return make_rsrc(out + xf * (size_t)m, (unsigned)m * sizeof(cpx<T>)); }
    __device__ __forceinline__ Raw fetch_d(rsrc_t d, int lane_bytes, int iu, int row_off = 0) const
    {
        // k = iu + tau; m - k = (m - iu - tpt + 1) + (tpt - 1 - tau): a non-negative constant plus the reversed lane
        Raw r;
        r.a = buf_load_cpx<T, AUX_NT>(d, row_off + lane_bytes, iu * (int)sizeof(cpx<T>));
        r.rb = buf_load_cpx<T, AUX_NT>(d, row_off + (tpt - 1) * (int)sizeof(cpx<T>) - lane_bytes, (m - iu - tpt + 1) * (int)sizeof(cpx<T>));
        return r;
    }
    __device__ __forceinline__ void store_d(rsrc_t d, int lane_bytes, int ou, cpx<T> v, int row_off = 0) const
    {
        if (m > 1) {
            const T im = -v.im;
            v = mk<T>(v.re * scale, im * scale);
        }
        buf_store_cpx<T>(v, d, row_off + lane_bytes, ou * (int)sizeof(cpx<T>));
    }
    // group forms: cnt consecutive rows from xf0
    __device__ __forceinline__ rsrc_t in_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(in + (cnt > 0 ? xf0 : 0) * (size_t)(m + 1), (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)(m + 1) * (unsigned)sizeof(cpx<T>));
    }
    __device__ __forceinline__ rsrc_t out_desc_n(size_t xf0, int cnt) const
    {
        return make_rsrc(out + (cnt > 0 ? xf0 : 0) * (size_t)m, (unsigned)(cnt > 0 ? cnt : 0) * (unsigned)m * (unsigned)sizeof(cpx<T>));
    }
    __device__ __forceinline__ unsigned in_slot_bytes() const { return (unsigned)(m + 1) * sizeof(cpx<T>); }
    __device__ __forceinline__ unsigned out_row_bytes() const { return (unsigned)m * sizeof(cpx<T>); }
    __device__ __forceinline__ Inv invariant(int k) const { return rtab[k]; }
    __device__ __forceinline__ cpx<T> finish(size_t, int k, Raw r, Inv tw) const { return pre(k, r.a, r.rb, tw); }
    __device__ __forceinline__ bool inside(size_t) const { return false; }  // finish needs the element index: one form only
    __device__ __forceinline__ cpx<T> finish_in(Raw r, Inv tw) const { return pre(1, r.a, r.rb, tw); }
    // the same from a natural-order LDS copy of the row: mirror points at cell m - tau, so input[m - k] = mirror[-tpt * u]
    template <int R, int TPT, class InvAt>
    __device__ __forceinline__ void finish_pairs_lds(const RawPair *raw, const int tau, const cpx<T> *mirror, cpx<T> *cur, const InvAt inv_at) const
    {
#pragma unroll
        for (int u = 0; u < R; ++u) cur[u] = pre(u == 0 ? tau : 1, raw[u], mirror[-u * TPT], inv_at(u));
    }
    // scratch[k] of irfft_direct (rfft.rs:487-506) from input[k], input[m-k], W[k]; then ifft's conj on the way in
    __device__ __forceinline__ cpx<T> pre(int k, cpx<T> a, cpx<T> rb, cpx<T> tw) const
    {
        const T half = T(0.5f);
        const cpx<T> b = mk<T>(rb.re, -rb.im);
        const cpx<T> sum = cadd(a, b), diff = csub(a, b);
        const cpx<T> w = mk<T>(tw.re, -tw.im);
        const cpx<T> t = cmul(w, diff);
        const cpx<T> temp = csub(sum, mk<T>(t.im, -t.re));
        const cpx<T> general = mk<T>(temp.re * half, temp.im * half);                   // rfft.rs:495-503
        const cpx<T> first = mk<T>((a.re + rb.re) * half, (a.re - rb.re) * half);       // rfft.rs:491-493 (k == 0)
        cpx<T> s = (k == 0) ? first : general;
        if (m > 1) s.im = -s.im;  // ifft: conj on the way in (fft.rs:1163-1165); n == 1 returns early
        return s;
    }
    __device__ __forceinline__ cpx<T> load(size_t xf, int k) const
    {
        // Branch-free: k == 0 reads row[0] and row[m] like every other k reads row[k] and row[m-k]; both forms are
        // evaluated and one is selected (a per-lane branch here would put a wait between consecutive loads).
        const cpx<T> *row = in + xf * (size_t)(m + 1);
        return pre(k, ld_stream(row + k), ld_stream(row + (m - k)), rtab[k]);
    }
    __device__ __forceinline__ void store(size_t xf, int o, cpx<T> v) const
    {
        if (m > 1) {  // ifft: conj, then scale (fft.rs:1168-1172)
            const T im = -v.im;
            v = mk<T>(v.re * scale, im * scale);
        }
        st_stream(out + xf * (size_t)m + o, v);
    }
};

enum : int { EPI_STORE = 0, EPI_RFFT = 1 };

// ---- LDS exchange layouts --------------------------------------------------------------------------------------
// Plain (threads of a transform contiguous in the wave): each transform slot owns lds_elems(n) padded elements.
// Slot-minor (lanes run over XPB adjacent transforms first; 8-byte exchange elements -- c32 values, or the real /
// imaginary halves of c64 values moved in two rounds): the slots are INTERLEAVED at a two-element grain and two
// index bits are XOR-swizzled,
//     cell(idx, slot) = (idx >> 2) * 4*XPB  +  ((idx1 ^ idx3) * 2*XPB)  +  slot * 2  +  (idx0 ^ idx2),
// so that every DS access shape of the kernel is conflict-free at XPB = 8: a ds_write_b64 lane group (16 lanes =
// 2 consecutive threads x 8 slots) covers 16 consecutive cells; a ds_read_b64 lane group (32 lanes = 4 threads x
// 8 slots) covers the 32 cells of one 256-byte row, whether the four threads differ in index bits (0,1) (gather of
// the middle pass) or in bits (2,3) (gather of the last pass).  No padding: XPB * n * 8 bytes exactly.
template <int XPB>
__host__ __device__ constexpr int lds_cell_sm(int idx, int slot)
{
    return ((idx >> 2) * (4 * XPB)) + ((((idx >> 1) ^ (idx >> 3)) & 1) * (2 * XPB)) + slot * 2 + ((idx ^ (idx >> 2)) & 1);
}

// Bytes of LDS per workgroup.  SPLIT: the exchange moves real and imaginary parts in two rounds through a buffer
// of scalars (half the footprint: what lets c64 tiles of 8 columns fit twice per CU).
template <typename T, bool SPLIT, bool SLOT_MINOR, int XPB>
__host__ __device__ constexpr size_t lds_wg_bytes(int n)
{
    const size_t elem = SPLIT ? sizeof(T) : sizeof(cpx<T>);
    return SLOT_MINOR ? (size_t)XPB * n * elem : (size_t)XPB * lds_elems(n) * elem;
}

// Pass geometry: thread tau of a transform holds R = 2^RL registers u = g*2^Q + c.
template <int L, int RL, int P>
struct WgGeom {
    static constexpr int N = 1 << L;
    static constexpr int R = 1 << RL;
    static constexpr int TPT = N / R;
    static constexpr int NP = (L + RL - 1) / RL;
    static constexpr int S0 = P * RL;
    static constexpr int Q = (P == NP - 1) ? (L - RL * (NP - 1)) : RL;
    static constexpr int G = R >> Q;       // independent (k, j) groups held by a thread
    static constexpr int JB = L - S0 - Q;  // bits of j
    __host__ __device__ static constexpr int in_index(int tau, int u)
    {
        const int g = u >> Q, c = u & ((1 << Q) - 1);
        const int m = tau + g * TPT;
        return ((m >> JB) << (L - S0)) | (c << JB) | (m & ((1 << JB) - 1));
    }
    __host__ __device__ static constexpr int out_index(int tau, int u)
    {
        const int g = u >> Q, c = u & ((1 << Q) - 1);
        return (bitrev(c, Q) << (L - Q)) | (tau + g * TPT);
    }
};

// ---- slot-minor cells for tiles of 8 or 16 adjacent units: conflict-free by construction (round 2) -------------------------
// Banking is per instruction (MI355X_MICROARCH.md, LDS): ds_write_b64 serves 4 groups of 16 contiguous lanes on 32 dword
// banks, ds_read_b64 2 groups of 32 lanes on 64.  With lanes = [thread tau][slot], a write group is 2 (XPB = 8) or 1 (16)
// consecutive threads x all slots and a read group 4 or 2 consecutive threads x all slots.  Cells
//     cell(idx, slot) = g(idx) * XPB + slot,      g(idx) = idx with bit 0 ^= parity(idx & M0) and (XPB = 8) bit 1 ^= parity(idx & M1),
// keep a thread's slots contiguous, so a group is conflict-free iff its threads' g(idx) differ in the low one / two bits.
// The threads of a group differ in thread bits 0 (and 1); M0 / M1 collect the index bit each gather shape of this (L, RL)
// puts thread bit 0 / 1 at (scatters keep them at index bits 0 / 1).  Simulated for every pass of L = 5 .. 10 under the
// documented grouping: no conflict; measured on the c32 first factor (16 columns, where the earlier two-element-grain
// formula -- built for 8 -- put slots s and s + 8 of a write group on one bank pair): SQ_LDS_BANK_CONFLICT 1.5e7 -> 0.
template <int L, int RL>
__host__ __device__ constexpr int sm_thread_bit_mask(int which)
{
    constexpr int NP = (L + RL - 1) / RL;
    int m = 0;
    if (NP > 1) m |= WgGeom<L, RL, (NP > 1 ? 1 : 0)>::in_index(1 << which, 0);
    if (NP > 2) m |= WgGeom<L, RL, (NP > 2 ? 2 : 0)>::in_index(1 << which, 0);
    if (NP > 3) m |= WgGeom<L, RL, (NP > 3 ? 3 : 0)>::in_index(1 << which, 0);
    if (NP > 4) m |= WgGeom<L, RL, (NP > 4 ? 4 : 0)>::in_index(1 << which, 0);
    return m | (1 << which);
}

// LDS exchange between pass P and pass P+1: scatter pass P's outputs, gather pass P+1's inputs.
// `base` is the workgroup's exchange region; SM selects the slot-minor interleaved layout.
template <typename T, int L, int RL, int P, bool SPLIT, bool SM, int XPB>
__device__ __forceinline__ void wg_exchange(cpx<T> *v, char *base, const int tau, const int slot)
{
    using Gs = WgGeom<L, RL, P>;
    using Gg = WgGeom<L, RL, P + 1>;
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    if (P > 0) __syncthreads();  // every gather of the previous exchange is done
    if constexpr (SM) {
        // Slot-minor cells as "one of at most four per-thread bases + a compile-time constant".  index(tau, u) is the OR of
        // two disjoint bit fields, T = index(tau, 0) and U = index(0, u), and lds_cell_sm is XOR-linear in the index bits:
        //   cell = [(T >> 2)*4X + 2*slot] + (U >> 2)*4X + 2X*(t13 ^ u13) + (t02 ^ u02),   t13 = T1 ^ T3, t02 = T0 ^ T2, same for U,
        // with u13, u02 constants of the register number.  Left as lds_cell_sm(index(tau, u), slot) the compiler cannot see
        // this and keeps one address register per register u alive across a persistent kernel's tile loop (2 x 16 VGPRs).
        // Tiles of 8 / 16 units: the conflict-free cells g(idx) * XPB + slot above; g is XOR-linear too:
        //   cell = [(T & ~LM) + (low(T) ^ low(U))] * XPB + slot + (U & ~LM) * XPB,   LM = the low one / two index bits g alters.
        constexpr bool UNI = (XPB == 8 || XPB == 16) && (N >> RL) >= 4;
        constexpr int M0 = sm_thread_bit_mask<L, RL>(0), M1 = sm_thread_bit_mask<L, RL>(1), LM = XPB == 8 ? 3 : 1;
        auto low = [&](const int x) -> int {
            return (__builtin_popcount(x & M0) & 1) | (XPB == 8 ? (__builtin_popcount(x & M1) & 1) << 1 : 0);
        };
        auto bases = [&](const int ti, int (&b)[2][2]) {  // ti = index(tau, 0)
            if constexpr (UNI) {
                const int lt = low(ti), core = (ti & ~LM) * XPB + slot;
#pragma unroll
                for (int j = 0; j <= LM; ++j) b[j >> 1][j & 1] = core + ((lt ^ j) * XPB);
            } else {
                const int t13 = ((ti >> 1) ^ (ti >> 3)) & 1, t02 = (ti ^ (ti >> 2)) & 1;
                const int core = (ti >> 2) * (4 * XPB) + slot * 2;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[i][j] = core + ((t13 ^ i) * (2 * XPB)) + (t02 ^ j);
            }
        };
        int bs[2][2], bg[2][2];
        bases(Gs::out_index(tau, 0), bs);
        bases(Gg::in_index(tau, 0), bg);
        auto pick = [&](const int (&b)[2][2], const int U) -> int {  // U = index(0, u): a compile-time constant
            if constexpr (UNI) {
                const int lu = low(U);
                return b[lu >> 1][lu & 1] + (U & ~LM) * XPB;
            } else {
                return b[((U >> 1) ^ (U >> 3)) & 1][(U ^ (U >> 2)) & 1] + (U >> 2) * (4 * XPB);
            }
        };
        auto cs = [&](int u) -> int { return pick(bs, Gs::out_index(0, u)); };
        auto cg = [&](int u) -> int { return pick(bg, Gg::in_index(0, u)); };
        if constexpr (!SPLIT) {
            cpx<T> *buf = reinterpret_cast<cpx<T> *>(base);
#pragma unroll
            for (int u = 0; u < R; ++u) buf[cs(u)] = v[u];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < R; ++u) v[u] = buf[cg(u)];
        } else {
            T *buf = reinterpret_cast<T *>(base);
            T re[R];
#pragma unroll
            for (int u = 0; u < R; ++u) buf[cs(u)] = v[u].re;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < R; ++u) re[u] = buf[cg(u)];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < R; ++u) buf[cs(u)] = v[u].im;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < R; ++u) v[u] = mk<T>(re[u], buf[cg(u)]);
        }
        return;
    }
    auto cell = [&](int idx) -> int { return slot * lds_elems(N) + lds_pad(idx); };
    if constexpr (!SPLIT) {
        cpx<T> *buf = reinterpret_cast<cpx<T> *>(base);
#pragma unroll
        for (int u = 0; u < R; ++u) buf[cell(Gs::out_index(tau, u))] = v[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = buf[cell(Gg::in_index(tau, u))];
    } else {
        T *buf = reinterpret_cast<T *>(base);
        T re[R];
#pragma unroll
        for (int u = 0; u < R; ++u) buf[cell(Gs::out_index(tau, u))] = v[u].re;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) re[u] = buf[cell(Gg::in_index(tau, u))];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) buf[cell(Gs::out_index(tau, u))] = v[u].im;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = mk<T>(re[u], buf[cell(Gg::in_index(tau, u))]);
    }
}

template <typename T, int L, int RL, int P, class IO>
__device__ __forceinline__ void wg_compute(cpx<T> *v, const IO &io, const cpx<T> *__restrict__ tw, const size_t xf, const int tau)
{
    using Gm = WgGeom<L, RL, P>;
#pragma unroll
    for (int g = 0; g < Gm::G; ++g)
        reg_pass<T, L, Gm::S0, Gm::Q, false>(&v[g * (1 << Gm::Q)], (tau + g * Gm::TPT) >> Gm::JB, tw, io.tw_map(xf));
}

// A single transform whose complex exchange buffer would leave room for only one workgroup per CU moves its real
// and imaginary parts in two rounds instead (half the LDS, two workgroups per CU overlap each other's HBM phases).
template <typename T, int L, int EPI, class IO>
constexpr bool wg_split_lds()
{
    return IO::kSplitLds || (EPI == 0 && !IO::kSlotMinor && sizeof(T) == 4 && L == 14);
}

#ifndef KOFFT_RFFT_CHUNK_STORE
#define KOFFT_RFFT_CHUNK_STORE 1
#endif
template <typename T, int L, int RL, int BLOCK, int EPI, class IO>
// (round 6: the factor policies' two-workgroups-per-CU bound -- kMinWaves = 4 in f64, i.e. 128 registers -- now holds at L = 9 only: at L = 10 / 11 the
// c64 kernels spilled 10 .. 38 registers under it (44 .. 156 bytes of scratch per lane); they are the fallback of batches of 2 .. 7 large transforms,
// and one workgroup per CU without scratch replaces two with it)
__global__ __launch_bounds__(BLOCK, (IO::kMinWaves > 1 && BLOCK == 512 && L == 9) ? IO::kMinWaves : (wg_split_lds<T, L, EPI, IO>() && !IO::kSplitLds ? 2 : 1)) void fft_wg_kernel(const IO io, const cpx<T> *__restrict__ tw, const size_t batch)
{
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    static_assert(TPT >= 1 && BLOCK % TPT == 0, "bad geometry");
    constexpr int XPB = BLOCK / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP >= 1 && NP <= 5, "pass count");
    constexpr bool SPLIT = wg_split_lds<T, L, EPI, IO>();
    constexpr bool SM = IO::kSlotMinor;
    static_assert(!((SPLIT || SM) && EPI == EPI_RFFT), "the rfft epilogue reads whole complex values from a plain slot");
    static_assert(!SM || (SPLIT ? sizeof(T) : sizeof(cpx<T>)) == 8, "slot-minor layout is built for 8-byte exchange elements");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const int tid = threadIdx.x;
    // kSlotMinor: consecutive lanes belong to consecutive transforms (used when adjacent transforms are adjacent
    // in memory, e.g. the columns of fft_big's first factor), otherwise to consecutive threads of one transform.
    const int tau = IO::kSlotMinor ? tid / XPB : tid % TPT;
    const int slot = IO::kSlotMinor ? tid % XPB : tid / TPT;
    // kPairXcd: workgroups b and b+8 land on the same XCD at about the same time (round-robin dispatch; a speed
    // heuristic only, results never depend on it).  Handing them ADJACENT tiles lets the two halves of each
    // 128-byte line meet in that XCD's L2 instead of being fetched twice (measured on the column-tile copy: 3.1 -> 4.2 TB/s).
    size_t blk = blockIdx.x;
    if (IO::kPairXcd && blk < (gridDim.x & ~15u)) blk = 16 * (blk / 16) + 2 * (blk % 8) + ((blk / 8) % 2);
    const size_t xf = blk * XPB + slot;
    const bool active = xf < batch;

    cpx<T> v[R];
    {   // pass 0 inputs through the IO policy
        using G0 = WgGeom<L, RL, 0>;
        bool fetched = false;
        if constexpr (IO::kStreams && !IO::kSlotMinor) {
            // Streaming policies: every load of the workgroup goes through ONE buffer descriptor whose bounds
            // check supplies the zeros past the end of the batch / signal, so the R loads issue back to back
            // with no per-element address test or branch (the per-element form serialises on s_waitcnt).
            if (io.wg_desc_ok(XPB)) {
                const rsrc_t d = io.in_desc_wg(blk * XPB, batch, XPB);
                const int lane_bytes = slot * (int)io.in_slot_bytes() + tau * IO::kRawBytes;
                typename IO::Raw raw[R];
#pragma unroll
                for (int u = 0; u < R; ++u) raw[u] = io.fetch_d(d, lane_bytes, G0::in_index(0, u));
                if (io.inside(blk * XPB + (XPB - 1))) {  // workgroup-uniform: every frame of the workgroup is complete
#pragma unroll
                    for (int u = 0; u < R; ++u) v[u] = io.finish_in(raw[u], io.invariant(G0::in_index(tau, u)));
                } else {
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const int i = G0::in_index(tau, u);
                        v[u] = io.finish(xf, i, raw[u], io.invariant(i));
                    }
                }
                fetched = true;
            }
        }
        if constexpr (io_stages_pairs<IO>::value && !SPLIT && !SM) {
            // irfft: value k needs row elements k and m - k.  Load every element ONCE, put the row into the transform's
            // exchange buffer in natural order (unpadded: N + 1 cells fit the padded buffer; pass 0 holds k = tau + TPT*u),
            // read the partners back: one LDS round trip instead of a second, reversed global load per element.
            cpx<T> *nat = reinterpret_cast<cpx<T> *>(smem_raw) + (size_t)slot * lds_elems(N);
            cpx<T> a[R];
            if (active) {
#pragma unroll
                for (int u = 0; u < R; ++u) a[u] = io.load_a(xf, G0::in_index(tau, u));
                if (tau == 0) nat[N] = io.load_a(xf, N);
#pragma unroll
                for (int u = 0; u < R; ++u) nat[G0::in_index(tau, u)] = a[u];
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const int k = G0::in_index(tau, u);
                    v[u] = io.pre_staged(k, a[u], nat[N - k]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < R; ++u) v[u] = mk<T>(T(0), T(0));
            }
            if (NP > 1) __syncthreads();  // the partners are read before the first exchange's scatter reuses the buffer
            fetched = true;
        }
        if (!fetched) {
            // one branch around ALL loads (a per-element test would put a branch and a wait between them)
            if (active) {
#pragma unroll
                for (int u = 0; u < R; ++u) v[u] = io.load(xf, G0::in_index(tau, u));
            } else {
#pragma unroll
                for (int u = 0; u < R; ++u) v[u] = mk<T>(T(0), T(0));
            }
        }
    }
    wg_compute<T, L, RL, 0>(v, io, tw, xf, tau);
    if constexpr (NP > 1) { wg_exchange<T, L, RL, 0, SPLIT, SM, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 1>(v, io, tw, xf, tau); }
    if constexpr (NP > 2) { wg_exchange<T, L, RL, 1, SPLIT, SM, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 2>(v, io, tw, xf, tau); }
    if constexpr (NP > 3) { wg_exchange<T, L, RL, 2, SPLIT, SM, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 3>(v, io, tw, xf, tau); }
    if constexpr (NP > 4) { wg_exchange<T, L, RL, 3, SPLIT, SM, XPB>(v, smem_raw, tau, slot); wg_compute<T, L, RL, 4>(v, io, tw, xf, tau); }

    using GL = WgGeom<L, RL, NP - 1>;
    if constexpr (EPI == EPI_RFFT) {
        // rfft.rs:450-463: Y in natural order through LDS, then X[k] from Y[k], Y[m-k] (m = N)
        cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem_raw) + (size_t)slot * lds_elems(N);
        if (NP > 1) __syncthreads();
#pragma unroll
        for (int u = 0; u < R; ++u) buf[lds_pad(GL::out_index(tau, u))] = v[u];
        __syncthreads();
        // (measured per case, same box: f64 n = 64 0.67 -> 0.80, n = 128 0.74 -> 0.79; f64 n = 256 and f32 n = 128 / 256 no change,
        // f64 n = 512 0.71 -> 0.63: only the first two take this route)
        if constexpr (KOFFT_RFFT_CHUNK_STORE && XPB >= 4 && sizeof(T) == 8 && N <= 64 && !SPLIT) {
            // Short rows (N + 1 <= 257 values, several transforms per workgroup): the workgroup's XPB output rows are ONE
            // contiguous piece of memory.  X goes back into the (now free) LDS rows and the piece is written in memory order
            // by all threads -- whole lines whatever the row length, instead of XPB misaligned row tails per store.
            cpx<T> xo[R];
            const cpx<T> y0 = buf[lds_pad(0)];
#pragma unroll
            for (int g = 0; g < R; ++g) {
                const int k = tau + g * TPT;
                const int kc = k < 1 ? 1 : k;  // LDS addressing only
                const cpx<T> p = io.post_w(io.rtab[k], buf[lds_pad(kc)], buf[lds_pad(N - kc)]);
                xo[g] = (k == 0) ? mk<T>(y0.re + y0.im, T(0)) : p;
            }
            __syncthreads();  // every Y has been read
#pragma unroll
            for (int g = 0; g < R; ++g) buf[tau + g * TPT] = xo[g];
            if (tau == 0) buf[N] = mk<T>(y0.re - y0.im, T(0));
            __syncthreads();
            const size_t row0 = blk * XPB;
            if (row0 < batch) {
                const size_t rows = batch - row0 < (size_t)XPB ? batch - row0 : (size_t)XPB;
                const int total = (int)rows * (N + 1);
                cpx<T> *ochunk = io.out + row0 * (size_t)(N + 1);
                const cpx<T> *all = reinterpret_cast<const cpx<T> *>(smem_raw);
                for (int e = tid; e < total; e += BLOCK) {
                    const int r = e / (N + 1), i = e - r * (N + 1);
                    st_stream(ochunk + e, all[(size_t)r * lds_elems(N) + i]);
                }
            }
        } else if (active) {
            // Rows are N+1 values back to back, so row xf starts `a` elements past a 128-byte line.  Lane tau of
            // store g handles k = g*TPT + tau - a: each store instruction then covers whole lines per row (see the
            // persistent kernel's epilogue; there: +8 %).  g = 0 and g = R are partial, g = R also carries X[N].
            constexpr int LINE = 128 / (int)sizeof(cpx<T>);
            cpx<T> *orow = io.out + xf * (size_t)(N + 1);
            const int k0 = tau - (TPT >= LINE ? (io.row_misalign(xf) & (LINE - 1)) : 0);  // narrower rows: as they come
            // all table entries first, then all results, then all stores: interleaving them makes every store wait
            // for the next table load (the compiler cannot prove the table and the output disjoint)
            // (f64, 16 points per thread: in two halves -- 2 x 17 values of 4 registers each held at once made this
            // epilogue, not the transform, set the kernel's register count: 206 VGPRs = 2 waves per SIMD)
            constexpr int CH = (sizeof(T) == 8 && R >= 16) ? 2 : 1;
            constexpr int PER = (R + 1 + CH - 1) / CH;
            const cpx<T> y0 = buf[lds_pad(0)];
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                cpx<T> w[PER], xo[PER];
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int g = ch * PER + j, k = k0 + g * TPT;
                    if (g <= R) w[j] = io.rtab[k < 0 ? 0 : (k > N - 1 ? N - 1 : k)];
                }
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int g = ch * PER + j, k = k0 + g * TPT;
                    if (g <= R) {
                        const int kc = k < 1 ? 1 : (k > N - 1 ? N - 1 : k);  // LDS addressing only
                        const cpx<T> p = io.post_w(w[j], buf[lds_pad(kc)], buf[lds_pad(N - kc)]);
                        xo[j] = (g == 0 && k == 0) ? mk<T>(y0.re + y0.im, T(0)) : (g == R && k == N) ? mk<T>(y0.re - y0.im, T(0)) : p;
                    }
                }
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const int g = ch * PER + j;
                    if (g == 0) {
                        if (k0 >= 0) st_stream(orow + k0, xo[j]);
                    } else if (g < R) {
                        st_stream(orow + k0 + g * TPT, xo[j]);
                    } else if (g == R) {
                        if (k0 <= 0) st_stream(orow + k0 + N, xo[j]);
                    }
                }
            }
        }
    } else {
        if constexpr (io_has_acc<IO>::value) {
            typename IO::Acc acc = io.acc_init();
            if (active) {
#pragma unroll
                for (int u = 0; u < R; ++u) io.store_acc(xf, GL::out_index(tau, u), v[u], acc);
            }
            io.acc_finish(acc);  // every wavefront of the workgroup, active or not (the reduction uses all lanes)
        } else {
            if (active) {
#pragma unroll
                for (int u = 0; u < R; ++u) io.store(xf, GL::out_index(tau, u), v[u]);
            }
        }
    }
}

// ---- ISTFT overlap-add (stft.rs:135-154) --------------------------------------------------------------------------
// The reference adds frame f's contribution to output[f*hop + i] for f = 0, 1, 2, ... in order.  One thread per output
// sample s replays exactly that order for its own sample: frames f_lo .. f_hi (those with f*hop <= s < f*hop + win_len),
// increasing f, acc += frame[f][s - f*hop].re * window[i]; norm += window[i]*window[i]; then the > 1e-8 normalisation.
// No atomics, so the f32 sums are the reference's sums bit for bit.  `frames` holds the already inverse-transformed
// frames (the reference transforms them in place too).
// MODE 0: stft::inverse_frame (stft.rs:384-399) -- accumulate only, no normalisation (scratch untouched);
// MODE 1: stft::istft (stft.rs:117-156) -- normalise where the window-square sum exceeds 1e-8, else leave the sum;
// MODE 2: stft::inverse_parallel (stft.rs:289-343) -- as MODE 1 but samples with a tiny sum become 0.
// Frame f starts at sample start0 + f*hop.
// The samples a launch covers: `nranges` runs of `range_len` samples, the first at s_first, `range_stride` apart (the whole output: one run
// of out_len samples from 0; istft_fused_kernel leaves the runs at its workgroups' seams and the tail to this kernel).
template <int MODE>
__global__ __launch_bounds__(256) void istft_ola_kernel(const cpx<float> *__restrict__ frames, const float *__restrict__ window,
                                                        float *__restrict__ output, float *__restrict__ scratch,
                                                        const size_t nframes, const size_t win_len, const size_t hop,
                                                        const size_t out_len, const size_t start0, const size_t s_first,
                                                        const size_t range_len, const size_t range_stride, const size_t nranges)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rg = idx / range_len;
    if (rg >= nranges) return;
    const size_t s = s_first + rg * range_stride + (idx - rg * range_len);
    if (s >= out_len) return;
    float acc = output[s];  // accumulated into the caller's buffer (stft.rs:144, 330, 395)
    float norm = 0.0f;      // scratch / norm start from zero (stft.rs:132-134, 325-326)
    if (nframes > 0 && win_len > 0 && s >= start0) {
        const size_t r = s - start0;
        size_t f_hi = r / hop;
        if (f_hi > nframes - 1) f_hi = nframes - 1;
        const size_t f_lo = (r >= win_len) ? (r - win_len) / hop + 1 : 0;
        for (size_t f = f_lo; f <= f_hi; ++f) {
            const size_t i = r - f * hop;
            const float w = window[i];
            acc = acc + frames[f * win_len + i].re * w;
            norm = norm + w * w;
        }
    }
    if (MODE == 0) {
        output[s] = acc;
    } else {
        scratch[s] = norm;
        if (norm > 1e-8f) output[s] = acc / norm;  // stft.rs:150-154 / 335-341
        else output[s] = (MODE == 2) ? 0.0f : acc;
    }
}

// ---- n = 1, 2, 4, 8, 16 (and 32 in f32): one thread per transform ---------
// (fft.rs:1059-1071 dispatch; ifft wraps them with conj / conj*scale via the IO policy)
// A thread's n inputs are n*8 (or 16) consecutive bytes, so lanes reading "their own" element i would touch 64
// different cache lines per instruction.  The workgroup therefore moves its 256 transforms through LDS: global
// reads and writes run over the workgroup's elements in memory order (fully coalesced), and each thread picks its
// row out of LDS (row stride n+1 cells: odd, so the 64 lanes hit distinct banks).
// n = 32: workgroups of ONE wavefront -- nine 17 KiB workgroups per CU, each running its load / transform / store phases on its
// own (the barriers are no-ops), interleave better than two of 66 KiB.  Same box, fraction of the roofline, 256 -> 128 -> 64
// threads: c32 n = 32 0.60..0.71 -> 0.75 -> 0.78..0.80, rfft n = 64 0.59 -> 0.59 -> 0.62..0.63, irfft n = 64 0.60 -> 0.60 -> 0.64,
// STFT n = 32 0.61..0.66 -> 0.61..0.63 -> 0.65..0.69.  (n <= 16 at 64 threads: c64 +2..5 %, c32 n = 8 / 16 -2 %, rfft64 n = 32
// -4 %: mixed, left at 256.)
// Tried and dropped in round 3, same box: PERSISTENT workgroups with the next block's loads prefetched into registers
// (c32 0.69, rfft n = 64 0.60 -> 0.54, STFT n = 32 0.63 -> 0.53: 246 VGPRs, and the phases of two wavefronts per SIMD still do not
// overlap); the 8-threads-per-transform persistent kernel (bit-equal, 0.42: its loads are 64-byte runs).
#ifndef KOFFT_SMALL_BLOCK
#define KOFFT_SMALL_BLOCK 256
#endif
#ifndef KOFFT_SMALL_BLOCK32
#define KOFFT_SMALL_BLOCK32 64
#endif
// (a policy may ask for its own n = 32 workgroup: kSmallBlock32)
template <class IO, class = void>
struct io_small_block32 { static constexpr int value = KOFFT_SMALL_BLOCK32; };
template <class IO>
struct io_small_block32<IO, decltype((void)IO::kSmallBlock32)> { static constexpr int value = IO::kSmallBlock32; };
// (and for n = 16: kSmallBlock16 -- c64 runs 64: 0.72 -> 0.745 on two boxes; n = 2 .. 8 at 64: -4 .. +3 %, left)
template <class IO, class = void>
struct io_small_block16 { static constexpr int value = KOFFT_SMALL_BLOCK; };
template <class IO>
struct io_small_block16<IO, decltype((void)IO::kSmallBlock16)> { static constexpr int value = IO::kSmallBlock16; };
template <int N, class IO>
constexpr int small_block_threads() { return N == 32 ? io_small_block32<IO>::value : N == 16 ? io_small_block16<IO>::value : KOFFT_SMALL_BLOCK; }
template <typename T, int N, class IO>
constexpr size_t small_lds_bytes() { return N == 1 ? 0 : (size_t)small_block_threads<N, IO>() * (N + 1) * sizeof(cpx<T>); }

template <typename T, int N, int EPI, class IO>
__global__ __launch_bounds__((small_block_threads<N, IO>())) void fft_small_kernel(const IO io, const cpx<T> *__restrict__ tw, const size_t batch)
{
    constexpr int B = small_block_threads<N, IO>();
    constexpr int S = N + 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem_raw);
    const int t = threadIdx.x;
    const size_t xf0 = (size_t)blockIdx.x * B;
    const size_t xf = xf0 + t;
    cpx<T> x[N];
    if constexpr (N == 1) {
        if (xf >= batch) return;
        x[0] = io.load(xf, 0);
    } else {
        bool fetched = false;
        if constexpr (IO::kStreams) {
            if (io.wg_desc_ok(B)) {
                const rsrc_t d = io.in_desc_wg(xf0, batch, B);
                typename IO::Raw raw[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const int e = j * B + t, r = e / N, i = e % N;
                    raw[j] = io.fetch_d(d, r * (int)io.in_slot_bytes() + i * IO::kRawBytes, 0);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const int e = j * B + t, r = e / N, i = e % N;
                    buf[r * S + i] = io.finish(xf0 + r, i, raw[j], io.invariant(i));
                }
                fetched = true;
            }
        }
        // (N <= 8 only: at 16 / 32 the N + 1 staging registers next to the N data registers cost more than the second load --
        // measured irfft32 n = 32 0.71 -> 0.66, n = 64 0.62 -> 0.48 with it, n = 8 / 16 0.76 / 0.74 -> 0.80 / 0.79)
        constexpr bool STAGE_ROWS = io_stages_pairs<IO>::value && N <= 8;
        if constexpr (STAGE_ROWS) {
            // irfft: an input row is N + 1 values = exactly one LDS row, and the workgroup's rows are contiguous in memory:
            // copy them in memory order (every element ONCE), then scratch[k] from the row's cells k and N - k.
            const size_t first = xf0 * (size_t)S;
            const size_t avail = (batch - xf0 < (size_t)B ? batch - xf0 : (size_t)B) * (size_t)S;  // elements that exist
            cpx<T> tmp[S];
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const size_t e = (size_t)j * B + t;
                tmp[j] = e < avail ? io.load_raw(first + e) : mk<T>(T(0), T(0));
            }
#pragma unroll
            for (int j = 0; j < S; ++j) buf[(size_t)j * B + t] = tmp[j];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = io.pre_staged(i, buf[t * S + i], buf[t * S + (N - i)]);
            fetched = true;
        }
        if (!fetched) {
            if (xf0 + B <= batch) {  // full workgroup: no per-element test between the loads
                cpx<T> tmp[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const int e = j * B + t;
                    tmp[j] = io.load(xf0 + e / N, e % N);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const int e = j * B + t;
                    buf[(e / N) * S + e % N] = tmp[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const int e = j * B + t, r = e / N, i = e % N;
                    buf[r * S + i] = (xf0 + r < batch) ? io.load(xf0 + r, i) : mk<T>(T(0), T(0));
                }
            }
        }
        if constexpr (!STAGE_ROWS) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; ++i) x[i] = buf[t * S + i];
        }
    }
    if constexpr (N == 2) small_fft2(x);
    if constexpr (N == 4) small_fft4(x);
    if constexpr (N == 8) small_fft8(x);
    if constexpr (N == 16) small_fft16(x);
    // n = 32 is the first size of the table-driven Stockham path (fft.rs:1059-1071): all five stages in registers,
    // k == 0 so every twiddle index is a compile-time constant; output o ends up in register bitrev(o).
    if constexpr (N == 32) reg_pass<T, 5, 0, 5, true>(x, 0, tw, io.tw_map(xf));
    auto Y = [&](int o) -> cpx<T> & { return x[N == 32 ? bitrev(o, 5) : o]; };
    if constexpr (EPI == EPI_RFFT) {
        // rfft.rs:450-463 on registers; a row of LDS (n+1 cells) is exactly one output row
        const cpx<T> x0 = mk<T>(Y(0).re + Y(0).im, T(0)), xn = mk<T>(Y(0).re - Y(0).im, T(0));
        if constexpr (N == 1) {
            cpx<T> *orow = io.out + xf * (size_t)2;
            orow[0] = x0;
            orow[1] = xn;
        } else {
            cpx<T> y[N];
#pragma unroll
            for (int k = 1; k < N; ++k) y[k] = io.post(k, Y(k), Y(N - k));
            buf[t * S] = x0;
            buf[t * S + N] = xn;
#pragma unroll
            for (int k = 1; k < N; ++k) buf[t * S + k] = y[k];
            __syncthreads();
            const size_t cnt = batch - xf0 < (size_t)B ? batch - xf0 : (size_t)B;
            cpx<T> *ochunk = io.out + xf0 * (size_t)S;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const int e = j * B + t;
                if ((size_t)e < cnt * S) st_stream(ochunk + e, buf[e]);
            }
        }
    } else {
        if constexpr (N == 1) {
            if constexpr (!io_has_acc<IO>::value) io.store(xf, 0, x[0]);  // (a 1-point frame has no magnitude bins: n/2 == 0)
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) buf[t * S + i] = Y(i);
            __syncthreads();
            if constexpr (io_has_acc<IO>::value) {
                typename IO::Acc acc = io.acc_init();
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const int e = j * B + t, r = e / N, o = e % N;
                    if (xf0 + r < batch) io.store_acc(xf0 + r, o, buf[r * S + o], acc);
                }
                io.acc_finish(acc);
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const int e = j * B + t, r = e / N, o = e % N;
                    if (xf0 + r < batch) io.store(xf0 + r, o, buf[r * S + o]);
                }
            }
        }
    }
}

}  // namespace kofft
