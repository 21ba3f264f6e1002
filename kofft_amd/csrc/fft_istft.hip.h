// fft_istft.hip.h -- ISTFT (stft.rs:117-156, 289-343) with the overlap-add fused into the frames' inverse transform (round 4).
//
// istft_dev's two kernels move every frame three times: the inverse transform reads the spectrum and writes the time-domain frame in place
// (the reference transforms the caller's frames in place too, stft.rs:137), and the overlap-add kernel reads it again -- 2.9 GB where the
// call's own bytes are 2.07 GB (config 4's spectra).  Here a workgroup owns a RUN of consecutive frames, transforms XPB of them per step
// (the persistent kernel's passes: twiddles of passes 1.. in registers, next step's spectra in flight), writes them back in place and
// hands their real parts to the overlap-add through LDS: output sample s = B * hop + i is the sum over the frames f = B - C + 1 .. B that
// cover it, in increasing f -- the reference's order (stft.rs:139-147: for each frame, for each i: output[f * hop + i] += frame[i].re *
// window[i]; norm likewise) -- kept as a running (acc, norm) pair per sample in REGISTERS while the sample's C = win / hop frames go by:
// thread t holds the samples e = t + 256 m of the step's (XPB + C - 1) hop-blocks; after a step the first XPB blocks are complete
// (normalised and stored, stft.rs:150-154), the rest moves down by XPB blocks -- a static renaming of registers.
//
// Seams.  A block whose covering frames belong to two workgroups (the first C - 1 blocks of every run but the first) would need the
// previous run's last frames FIRST; their spectra are being overwritten in place by that workgroup, so they cannot be recomputed here.
// Those (G - 1) * (C - 1) blocks, the blocks after the last frame's own and everything the frames do not reach are left to the ordered
// overlap-add kernel (istft_ola_kernel, run after this one over just those ranges: it reads the stored time-domain frames).
// Same products, same sums in the same order as istft_ola_kernel: bit-identical output, scratch and frames.
#pragma once

#include "fft_persist.hip.h"

namespace kofft {

template <int L, int RL, int CL /* log2(win / hop) */, int WG_PER_CU>
__global__ __launch_bounds__(256, WG_PER_CU) void istft_fused_kernel(cpx<float> *__restrict__ frames, const float *__restrict__ window,
                                                                      float *__restrict__ output, float *__restrict__ scratch,
                                                                      const cpx<float> *__restrict__ tw, const size_t nframes,
                                                                      const size_t out_len, const float scale, const size_t frames_per_wg,
                                                                      const int mode)
{
    using T = float;
    constexpr int N = 1 << L;
    constexpr int R = 1 << RL;
    constexpr int TPT = N / R;
    static_assert(TPT <= 256 && 256 % TPT == 0, "a transform is at most the workgroup");
    constexpr int XPB = 256 / TPT;
    constexpr int NP = (L + RL - 1) / RL;
    static_assert(NP >= 2 && NP <= 3, "pass count");
    constexpr bool WAVE = TPT <= 64;
    constexpr int C = 1 << CL;
    constexpr int HL = L - CL;
    constexpr int HOP = 1 << HL;
    constexpr int DONE = XPB * HOP;              // samples completed per step
    constexpr int TOUCHED = DONE + N - HOP;      // samples a step's frames reach
    static_assert(DONE % 256 == 0, "the carried accumulators move down by whole registers");
    constexpr int PT = (TOUCHED + 255) / 256;    // samples per thread (the last round may be partial: hop < 256)
    constexpr int PD = DONE / 256;               // ... of which complete after the step
    using G0 = WgGeom<L, RL, 0>;
    using GL = WgGeom<L, RL, NP - 1>;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int tau = tid % TPT, slot = tid / TPT;
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem_raw) + (size_t)slot * lds_elems(N);
    float *fre = reinterpret_cast<float *>(reinterpret_cast<cpx<T> *>(smem_raw) + (size_t)XPB * lds_elems(N));  // [XPB][N] real parts
    float *wl = fre + XPB * N;                                                                                 // [N] window

    const size_t a = (size_t)blockIdx.x * frames_per_wg;  // this run: frames a .. b - 1
    if (a >= nframes) return;
    const size_t b = a + frames_per_wg < nframes ? a + frames_per_wg : nframes;

    for (int i = tid; i < N; i += 256) wl[i] = window[i];
    cpx<T> tw1[R - 1], tw2[NP >= 3 ? R - 1 : 1];
    persist_load_tw<T, L, RL, 1>(tw1, tau, tw);
    if constexpr (NP >= 3) persist_load_tw<T, L, RL, 2>(tw2, tau, tw);
    const int sc = lds_pad(tau);
    const int g1 = lds_pad(WgGeom<L, RL, 1>::in_index(tau, 0));
    const int g2 = NP >= 3 ? lds_pad(WgGeom<L, RL, (NP >= 3 ? 2 : 0)>::in_index(tau, 0)) : 0;

    auto fetch = [&](cpx<T> *raw, size_t f) {
        const size_t fc = f < nframes ? f : nframes - 1;  // past the end: a valid address, the values are never used
        const cpx<T> *row = frames + fc * (size_t)N;
#pragma unroll
        for (int u = 0; u < R; ++u) raw[u] = ld_stream(row + G0::in_index(tau, u));
    };

    float acc[PT], nrm[PT];
    cpx<T> raw[R];
    fetch(raw, a + slot);
    bool first = true;
    for (size_t F = a; F < b; F += XPB) {
        const size_t f = F + slot;
        const bool live = f < b;  // (b <= nframes)
        // the accumulators of the blocks this step reaches for the first time start from the caller's output (stft.rs:144) and 0
        float init[PT];
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const size_t s = F * HOP + (size_t)(tid + 256 * m);
            init[m] = ((first || tid + 256 * m >= N - HOP) && s < out_len) ? output[s] : 0.0f;  // (samples below N - HOP are carried)
        }
        cpx<T> v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            v[u] = raw[u];
            v[u].im = -v[u].im;  // ifft: conj on the way in (fft.rs:1163-1165)
        }
        fetch(raw, f + XPB);
        persist_compute_p0<T, L, RL>(v, tw);
        exchange_sync<WAVE>();
        persist_lds_scatter<T, L, RL, 0>(v, buf, sc);
        exchange_sync<WAVE>();
        persist_lds_gather<T, L, RL, 1>(v, buf, g1);
        persist_compute<T, L, RL, 1>(v, tw1);
        if constexpr (NP >= 3) {
            exchange_sync<WAVE>();
            persist_lds_scatter<T, L, RL, 1>(v, buf, sc);
            exchange_sync<WAVE>();
            persist_lds_gather<T, L, RL, 2>(v, buf, g2);
            persist_compute<T, L, RL, 2>(v, tw2);
        }
        // conj, * 1/n (fft.rs:1168-1172): the frame in place, its real parts to the overlap-add
        {
            cpx<T> *row = frames + (live ? f : a) * (size_t)N;
            float *fr = fre + slot * N;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int o = GL::out_index(tau, u);
                const T im = -v[u].im;
                const cpx<T> w = mk<T>(v[u].re * scale, im * scale);
                if (live) st_stream(row + o, w);
                fr[o] = w.re;
            }
        }
        __syncthreads();
        // overlap-add: sample e = tid + 256 m of the step lies in block e >> HL at offset i0; frame F + k covers it iff 0 <= block - k < C
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const int e = tid + 256 * m;
            if (first || e >= N - HOP) {
                acc[m] = init[m];
                nrm[m] = 0.0f;
            }
            const int blk = e >> HL, i0 = e & (HOP - 1);  // (e >= TOUCHED in a partial last round: blk - k >= C for every k)
#pragma unroll
            for (int k = 0; k < XPB; ++k) {
                const int d = blk - k;
                if (d >= 0 && d < C && F + k < b) {
                    const int i = d * HOP + i0;
                    const float w = wl[i];
                    acc[m] = acc[m] + fre[k * N + i] * w;  // stft.rs:143-145
                    nrm[m] = nrm[m] + w * w;
                }
            }
        }
        // the first XPB blocks are complete; the first C - 1 blocks of a run that is not the first one are seams (see the header)
#pragma unroll
        for (int m = 0; m < PD; ++m) {
            const size_t blk_abs = F + (size_t)((tid + 256 * m) >> HL);
            const size_t s = F * HOP + (size_t)(tid + 256 * m);
            if (s < out_len && (a == 0 || blk_abs >= a + (C - 1))) {
                if (mode == 0) {
                    output[s] = acc[m];
                } else {
                    scratch[s] = nrm[m];
                    if (nrm[m] > 1e-8f) output[s] = acc[m] / nrm[m];  // stft.rs:150-154 / 335-341
                    else output[s] = (mode == 2) ? 0.0f : acc[m];
                }
            }
        }
#pragma unroll
        for (int m = 0; m + PD < PT; ++m) {
            acc[m] = acc[m + PD];
            nrm[m] = nrm[m + PD];
        }
        first = false;
        __syncthreads();  // fre is rewritten by the next step
    }
}

}  // namespace kofft
