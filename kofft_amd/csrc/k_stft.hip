// k_stft.hip -- STFT framing kernels (stft.rs:76-105), the ISTFT overlap-add (stft.rs:117-156, 289-343, 384-399) and
// stft_magnitudes (visual/spectrogram.rs:52-76) on device pointers.
#include "host_common.hip.h"

namespace kofft {
namespace host {

// ---- composed form for window lengths the fused kernels do not cover (stft.rs:91-103 calls fft.fft(frame) for ANY
// win_len): the framing product into the output buffer, then fft_dev in place over the frames.
__global__ __launch_bounds__(256) void stft_frame_kernel(const float *__restrict__ signal, const float *__restrict__ window,
                                                         cpx<float> *__restrict__ out, const size_t len, const size_t win_len,
                                                         const size_t hop, const size_t start0, const size_t total /* frames * win_len */)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const size_t f = idx / win_len, i = idx % win_len;
    const size_t pos = start0 + f * hop + i;
    out[idx] = mk<float>(pos < len ? signal[pos] * window[i] : 0.0f, 0.0f);  // stft.rs:95-100
}

// magnitudes of bins 0 .. n/2-1 of composed frames, and their maximum (spectrogram.rs:63-71)
__global__ __launch_bounds__(256) void mag_kernel(const cpx<float> *__restrict__ spec, float *__restrict__ mags, unsigned *__restrict__ max_bits,
                                                  const size_t win_len, const size_t total /* frames * (win_len/2) */)
{
    const size_t half = win_len / 2;
    float m = 0.0f;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const cpx<float> c = spec[(idx / half) * win_len + idx % half];
        const float v = sqrtf(c.re * c.re + c.im * c.im);
        mags[idx] = v;
        if (v > m) m = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        if (o > m) m = o;
    }
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(max_bits, __builtin_bit_cast(unsigned, m));
}

static int stft_composed_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len, const float *d_window, size_t win_len,
                             size_t start0, size_t hop, float *d_out, size_t count)
{
    size_t chunk = (size_t(512) << 20) / (win_len * 8);
    if (chunk < 1) chunk = 1;
    if (chunk > count) chunk = count;
    if ((chunk * win_len + 255) / 256 > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    for (size_t f0 = 0; f0 < count; f0 += chunk) {
        const size_t nf = (count - f0 < chunk) ? count - f0 : chunk;
        cpx<float> *dst = reinterpret_cast<cpx<float> *>(d_out) + f0 * win_len;
        hipLaunchKernelGGL(stft_frame_kernel, dim3((unsigned)((nf * win_len + 255) / 256)), dim3(256), 0, ctx->stream, d_signal, d_window,
                           dst, len, win_len, hop, start0 + f0 * hop, nf * win_len);
        KOFFT_HIP_TRY(ctx, hipGetLastError());
        const int rc = fft_dev<float>(ctx, reinterpret_cast<const float *>(dst), reinterpret_cast<float *>(dst), win_len, nf, 0);
        if (rc) return rc;
    }
    return KOFFT_OK;
}

int stft_dev(kofft_hip_ctx *ctx, const float *d_signal, size_t len, const float *d_window, size_t win_len,
             size_t start0, size_t hop, float *d_out, size_t count)
{
    if (count == 0) return KOFFT_OK;
    if (win_len == 0) return KOFFT_ERR_EMPTY_INPUT;  // fft.fft(&mut []) -> fft.rs:1056
    if (!complex_len_ok(win_len)) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || (!d_signal && len) || !d_window || !d_out) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!fused_len_ok<float>(win_len)) {
        // a window that is not a power of two and enough frames: the framing product on the persistent Bluestein kernel's loads
        bool done = false;
        const int rc = stft_bluestein_dev(ctx, d_signal, len, d_window, win_len, start0, hop, d_out, count, &done);
        if (rc || done) return rc;
        return stft_composed_dev(ctx, d_signal, len, d_window, win_len, start0, hop, d_out, count);
    }
    StftIO io{{}, d_signal, d_window, reinterpret_cast<cpx<float> *>(d_out), len, hop, start0, (int)win_len};
    return dispatch<float, EPI_STORE>(ctx, io, win_len, count);
}

// The ordered overlap-add kernel over `nranges` runs of `range_len` output samples (see istft_ola_kernel).
static int istft_ola_ranges(kofft_hip_ctx *ctx, const float *d_frames, size_t frames, const float *d_window, size_t win_len, size_t hop,
                            float *d_output, size_t out_len, float *d_scratch, int mode, size_t start0, size_t s_first, size_t range_len,
                            size_t range_stride, size_t nranges)
{
    if (out_len == 0 || range_len == 0 || nranges == 0) return KOFFT_OK;
    const size_t blocks = (range_len * nranges + 255) / 256;
    if (blocks > 0x7fffffffULL) return KOFFT_ERR_UNSUPPORTED;
    const cpx<float> *fr = reinterpret_cast<const cpx<float> *>(d_frames);
    if (mode == 0)
        hipLaunchKernelGGL(istft_ola_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, fr, d_window, d_output, d_scratch, frames,
                           win_len, hop, out_len, start0, s_first, range_len, range_stride, nranges);
    else if (mode == 2)
        hipLaunchKernelGGL(istft_ola_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, fr, d_window, d_output, d_scratch, frames,
                           win_len, hop, out_len, start0, s_first, range_len, range_stride, nranges);
    else
        hipLaunchKernelGGL(istft_ola_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, fr, d_window, d_output, d_scratch, frames,
                           win_len, hop, out_len, start0, s_first, range_len, range_stride, nranges);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    return KOFFT_OK;
}

// The fused form (fft_istft.hip.h): win = 2^L, hop = win / C, frames from sample 0, enough frames for every workgroup of the chip-sized
// grid to own a run several steps long.  *done = false: the caller takes the two-kernel route.
template <int L, int CL>
static int launch_istft_fused(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window, float *d_output, size_t out_len,
                              float *d_scratch, int mode, bool *done)
{
    constexpr int RL = rl_for(L);
    constexpr int N = 1 << L, TPT = N >> RL, XPB = 256 / TPT, C = 1 << CL, HOP = N >> CL;
    constexpr int WG = 2;
    constexpr size_t lds = (size_t)XPB * lds_elems(N) * sizeof(cpx<float>) + (size_t)XPB * N * sizeof(float) + (size_t)N * sizeof(float);
    static_assert(lds * WG <= 160 * 1024, "LDS budget");
    *done = false;
    const size_t steps = (frames + XPB - 1) / XPB;
    size_t grid = (size_t)ctx->num_cus * WG;
    if (ctx->persist_grid_pct > 0) grid = grid * (size_t)ctx->persist_grid_pct / 100;
    if (grid < 1) grid = 1;
    const size_t fpw = ((steps + grid - 1) / grid) * XPB;  // frames per workgroup: whole steps
    // (fewer than four steps per run: the two kernels are as fast or faster -- 1024 / 256 x 3000 frames 0.023 against 0.030 ms fused)
    if (fpw < 4 * XPB || fpw < (size_t)(2 * C)) return KOFFT_OK;
    const size_t runs = (frames + fpw - 1) / fpw;
    const cpx<float> *tw = nullptr;
    int rc = get_table<float>(ctx, Kind<float>::tw, N, &tw);
    if (rc) return rc;
    auto kern = istft_fused_kernel<L, RL, CL, WG>;
    {
        static std::atomic<unsigned long long> attr_done{0};
        const int arc = set_dyn_lds_once(ctx, attr_done, reinterpret_cast<const void *>(kern), lds);
        if (arc) return arc;
    }
    const float scale = 1.0f / (float)N;
    hipLaunchKernelGGL(kern, dim3((unsigned)runs), dim3(256), lds, ctx->stream, reinterpret_cast<cpx<float> *>(d_frames), d_window, d_output,
                       d_scratch, tw, frames, out_len, scale, fpw, mode);
    KOFFT_HIP_TRY(ctx, hipGetLastError());
    // seams: the first C - 1 blocks of every run but the first
    if (runs > 1 && C > 1) {
        rc = istft_ola_ranges(ctx, d_frames, frames, d_window, N, HOP, d_output, out_len, d_scratch, mode, 0, fpw * HOP, (size_t)(C - 1) * HOP,
                              fpw * HOP, runs - 1);
        if (rc) return rc;
    }
    // tail: everything after the last run's completed blocks
    const size_t a_last = (runs - 1) * fpw;
    size_t tail_blk = a_last + ((frames - a_last + XPB - 1) / XPB) * XPB;
    if (runs > 1 && tail_blk < a_last + (size_t)(C - 1)) tail_blk = a_last + (size_t)(C - 1);
    const size_t tail = tail_blk * HOP;
    if (tail < out_len) {
        rc = istft_ola_ranges(ctx, d_frames, frames, d_window, N, HOP, d_output, out_len, d_scratch, mode, 0, tail, out_len - tail, 0, 1);
        if (rc) return rc;
    }
    *done = true;
    return KOFFT_OK;
}

static int istft_fused_dev(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window, size_t win_len, size_t hop,
                           float *d_output, size_t out_len, float *d_scratch, int mode, bool *done)
{
    *done = false;
    if (!is_pow2(win_len) || !is_pow2(hop) || hop > win_len) return KOFFT_OK;
    const int L = ilog2(win_len), CL = L - ilog2(hop);
#define KOFFT_CASE(LL, CC) \
    if (L == LL && CL == CC) return launch_istft_fused<LL, CC>(ctx, d_frames, frames, d_window, d_output, out_len, d_scratch, mode, done);
    // hop = win, win / 2, win / 4, win / 8 (L = 8: the 16 frames of a step reach one window further only from win / 2 down)
    KOFFT_CASE(8, 1)
    KOFFT_CASE(8, 2)
    KOFFT_CASE(8, 3)
    KOFFT_CASE(9, 0)
    KOFFT_CASE(9, 1)
    KOFFT_CASE(9, 2)
    KOFFT_CASE(9, 3)
    KOFFT_CASE(10, 0)
    KOFFT_CASE(10, 1)
    KOFFT_CASE(10, 2)
    KOFFT_CASE(10, 3)
    KOFFT_CASE(11, 0)
    KOFFT_CASE(11, 1)
    KOFFT_CASE(11, 2)
    KOFFT_CASE(11, 3)
    KOFFT_CASE(12, 0)
    KOFFT_CASE(12, 1)
    KOFFT_CASE(12, 2)
    KOFFT_CASE(12, 3)
#undef KOFFT_CASE
    return KOFFT_OK;
}

// stft::istft (stft.rs:117-156, mode 1), stft::inverse_parallel (stft.rs:289-343, mode 2), stft::inverse_frame
// (stft.rs:384-399, mode 0): ifft every frame in place, then the ordered overlap-add kernel -- or both in one (istft_fused_dev).
int istft_dev(kofft_hip_ctx *ctx, float *d_frames, size_t frames, const float *d_window, size_t win_len, size_t hop,
              float *d_output, size_t out_len, float *d_scratch, size_t scratch_len, int mode, size_t start0)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;                               // stft.rs:125 / 299
    if (mode == 1 && scratch_len != out_len) return KOFFT_ERR_MISMATCHED_LENGTHS;  // stft.rs:128
    if (frames > 0 && win_len == 0) return KOFFT_ERR_EMPTY_INPUT;                  // fft.ifft(&mut []) -> fft.rs:1136
    if (frames > 0 && !complex_len_ok(win_len)) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || (frames && (!d_frames || !d_window)) || (out_len && (!d_output || (mode != 0 && !d_scratch)))) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (frames > 0 && out_len > 0 && start0 == 0 && ctx->istft_fused && ctx->use_persist) {
        bool done = false;
        const int rc = istft_fused_dev(ctx, d_frames, frames, d_window, win_len, hop, d_output, out_len, d_scratch, mode, &done);
        if (rc || done) return rc;
    }
    if (frames > 0) {
        int rc = fft_dev<float>(ctx, d_frames, d_frames, win_len, frames, 1);
        if (rc) return rc;
    }
    return istft_ola_ranges(ctx, d_frames, frames, d_window, win_len, hop, d_output, out_len, d_scratch, mode, start0, 0, out_len, 0, 1);
}

// visual::spectrogram::stft_magnitudes (visual/spectrogram.rs:52-76): hann(win_len) window, frames x win_len/2 magnitudes
// and their maximum.  d_max receives one float.
int stft_mag_dev(kofft_hip_ctx *ctx, const float *d_samples, size_t len, size_t win_len, size_t hop, float *d_mags,
                 size_t frames, float *d_max)
{
    if (hop == 0) return KOFFT_ERR_INVALID_HOP_SIZE;  // the reference divides by hop (div_ceil) and would panic
    const size_t required = len / hop + (len % hop != 0);
    if (frames < required) return KOFFT_ERR_MISMATCHED_LENGTHS;
    if (frames > 0 && win_len == 0) return KOFFT_ERR_EMPTY_INPUT;
    if (frames > 0 && !complex_len_ok(win_len)) return KOFFT_ERR_UNSUPPORTED;
    if (!ctx || !d_max || (frames && (!d_mags || (!d_samples && len)))) return KOFFT_ERR_NULL;
    KOFFT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    KOFFT_HIP_TRY(ctx, hipMemsetAsync(d_max, 0, sizeof(float), ctx->stream));  // max_mag starts at 0.0
    if (frames == 0) return KOFFT_OK;
    // hann(win_len), cached per context like a planner table (kind 4)
    const float *d_win = nullptr;
    {
        auto key = std::make_pair(4, win_len);
        auto it = ctx->tables.find(key);
        if (it == ctx->tables.end()) {
            std::vector<float> w(win_len);
            kofft_tables::hann_f32(win_len, w.data());
            void *d = nullptr;
            KOFFT_HIP_TRY(ctx, hipMalloc(&d, win_len * sizeof(float)));
            const hipError_t ce = hipMemcpy(d, w.data(), win_len * sizeof(float), hipMemcpyHostToDevice);
            if (ce != hipSuccess) {  // nothing is cached, nothing is left behind
                (void)hipFree(d);
                ctx->last_error = std::string("stft_magnitudes window upload: ") + hipGetErrorString(ce);
                return KOFFT_ERR_HIP;
            }
            ctx->tables[key] = d;
            d_win = static_cast<const float *>(d);
        } else {
            d_win = static_cast<const float *>(it->second);
        }
    }
    if (!fused_len_ok<float>(win_len)) {
        // any other window length: the composed STFT into scratch, then magnitudes + maximum in one pass (in frame chunks)
        size_t chunk = (size_t(512) << 20) / (win_len * 8);
        if (chunk < 1) chunk = 1;
        if (chunk > frames) chunk = frames;
        int rc = ensure_real_tmp(ctx, chunk * win_len * 8);
        if (rc) return rc;
        float *spec = static_cast<float *>(ctx->real_tmp);
        for (size_t f0 = 0; f0 < frames; f0 += chunk) {
            const size_t nf = (frames - f0 < chunk) ? frames - f0 : chunk;
            rc = stft_composed_dev(ctx, d_samples, len, d_win, win_len, f0 * hop, hop, spec, nf);
            if (rc) return rc;
            const size_t total = nf * (win_len / 2);
            if (total) {
                size_t blocks = (total + 255) / 256;
                if (blocks > (size_t)ctx->num_cus * 16) blocks = (size_t)ctx->num_cus * 16;
                hipLaunchKernelGGL(mag_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const cpx<float> *>(spec),
                                   d_mags + f0 * (win_len / 2), reinterpret_cast<unsigned *>(d_max), win_len, total);
                KOFFT_HIP_TRY(ctx, hipGetLastError());
            }
        }
        return KOFFT_OK;
    }
    // one launch: the magnitudes are stored and their maximum reduced by the same kernel (StftMagIO::acc_finish)
    StftMagIO io{{{}, d_samples, d_win, nullptr, len, hop, 0, (int)win_len}, d_mags, reinterpret_cast<unsigned *>(d_max)};
    int rc = dispatch<float, EPI_STORE>(ctx, io, win_len, frames);
    if (rc) return rc;
    return KOFFT_OK;
}


}  // namespace host
}  // namespace kofft
