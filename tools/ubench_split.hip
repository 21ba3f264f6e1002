// ubench_split.hip -- phase timeline of fft_split_persist_kernel (development tool, not product): s_memtime stamps at the
// phase boundaries of every transform, written by lane 0 of every wavefront of a few workgroups; printed as the mean time
// per phase.  Stamps perturb the kernel (the guide: about +11 % wave cycles), so the un-stamped kernel is timed beside it.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/ubench_split tools/ubench_split.hip kofft_amd/csrc/tables.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifdef STAMPS
#define NSTAMP 12
#define MAXT 40
#ifndef NWAVE
#define NWAVE 8
#endif
__device__ unsigned long long g_stamps[8 * NWAVE * MAXT * NSTAMP];  // [wg < 8][wave][transform][stamp]
__device__ __forceinline__ void split_stamp(int id, size_t xf);
#define KOFFT_SPLIT_STAMP(id) split_stamp(id, xf);
#endif
#include "../kofft_amd/csrc/fft_split.hip.h"
#include "../kofft_amd/csrc/tables.h"
using namespace kofft;
#ifdef STAMPS
__device__ __forceinline__ void split_stamp(int id, size_t xf)
{
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 8) {
        const int t = (int)((xf - blockIdx.x) / gridDim.x);  // this workgroup's t-th transform
        if (t < MAXT) g_stamps[((blockIdx.x * NWAVE + w) * MAXT + t) * NSTAMP + id] = __builtin_amdgcn_s_memtime();
    }
}
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using IO = ComplexIO<float, false>;

int main(int argc, char **argv)
{
#ifndef SPLIT1
#ifndef SPLIT_LA
#define SPLIT_LA 7
#define SPLIT_LB 6
#endif
    constexpr int LA = SPLIT_LA, LB = SPLIT_LB, N = 1 << (LA + LB);
#else
    constexpr int LA = 7, LB = 7, N = 1 << (LA + LB);
#endif
    const size_t batch = argc > 1 ? atol(argv[1]) : 8192;
    const size_t bytes = batch * N * sizeof(cpx<float>);
    std::vector<float> htw(N);
    kofft_tables::twiddles_f32(N, htw.data());
    cpx<float> *dtw, *src, *out;
    CK(hipMalloc(&dtw, N * sizeof(float)));
    CK(hipMemcpy(dtw, htw.data(), N * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&out, bytes));
    {
        std::vector<float> h(1 << 22);
        unsigned long long s = 0x6B6F666674ull;
        for (auto &x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }
        for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy((char *)src + off, h.data(), std::min(h.size() * 4, bytes - off), hipMemcpyHostToDevice));
    }
    IO io{{}, src, out, N, 1.0f / N};
#ifndef SPLIT1
    auto k = fft_split_persist_kernel<float, LA, LB, IO>;
    const size_t lds = 2 * (size_t)N * 8 + 16;
    const int threads = N / 16;
#else
    auto k = fft_split1_persist_kernel<float, LA, LB, IO>;
    const size_t lds = ((size_t)N + 16 * 7 + 128 * 15) * 8;
    const int threads = 1024;
#endif
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(256 * (N <= 4096 ? 2 : 1)), dim3(threads), lds, 0, io, dtw, batch);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(256 * (N <= 4096 ? 2 : 1)), dim3(threads), lds, 0, io, dtw, batch);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 20;
    printf("n=%d (2^%d x 2^%d) batch=%zu: %.4f ms per launch, %.1f GB/s, frac %.3f, %.2f us per transform per CU\n", N, LA, LB, batch, ms, 2.0 * bytes / ms / 1e6,
           2.0 * bytes / ms / 1e6 / 8000.0, ms * 1e3 / ((double)batch / 256));
#ifdef STAMPS
    std::vector<unsigned long long> st(8 * NWAVE * MAXT * NSTAMP);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
    const char *names[NSTAMP] = {"(prev end->)load wait+finish", "A0 compute", "A0 scatter+gather", "A1 compute", "A1 scatter", "barrier", "B0 gather",
                                 "B0 compute", "B0 scatter+B1 gather", "B1 compute", "store issue", "loop/issue next"};
    const int tmax = (int)std::min<size_t>(MAXT, batch / 256) - 1;
    // absolute timeline of one step (workgroup 0, transform t = 10): every wave's stamps relative to wave 0's barrier exit
    {
        const int t = 10, wg = 0;
#ifdef SPLIT1
        const unsigned long long t0 = st[((wg * NWAVE + 0) * MAXT + t) * NSTAMP + 0];
        printf("transform t=%d of workgroup %d, clocks relative to wave 0's start; stamps: 0 top, 1 finish+loads, 2 A0, 3 barrier1, 4 xA, 5 A1, 6 scatter, 7 barrier2, 8 gather, 9 B0, 10 xB, 11 B1+stores, then next top\n", t, wg);
        for (int w = 0; w < NWAVE; ++w) {
            printf("wave %2d:", w);
            for (int i = 0; i < 12; ++i) printf(" %6lld", (long long)(st[((wg * NWAVE + w) * MAXT + t) * NSTAMP + i] - t0));
            printf(" | %6lld\n", (long long)(st[((wg * NWAVE + w) * MAXT + t + 1) * NSTAMP + 0] - t0));
        }
        return 0;
#else
        const unsigned long long t0 = st[((wg * 8 + 0) * MAXT + t) * NSTAMP + 6];
        printf("step t=%d of workgroup %d, clocks relative to wave 0 leaving the barrier; B(t) stamps 6..11 then A(t+1) stamps 0..5\n", t, wg);
        printf("        gB-start B0-start B0-end  xB-end  B1-end  st-end | A:fin-st fin-end A0-end  xA-end  A1-end  sA-end\n");
        for (int w = 0; w < 8; ++w) {
            printf("wave %d:", w);
            for (int i = 6; i < 12; ++i) printf(" %7lld", (long long)(st[((wg * 8 + w) * MAXT + t) * NSTAMP + i] - t0));
            printf(" |");
            for (int i = 0; i < 6; ++i) printf(" %7lld", (long long)(st[((wg * 8 + w) * MAXT + t + 1) * NSTAMP + i] - t0));
            printf("\n");
        }
#endif
    }
    (void)names;
#endif
    return 0;
}
