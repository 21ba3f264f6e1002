// ubench_seg.hip -- what a persistent one-workgroup-per-CU streaming kernel gets out of HBM as a function of the SEGMENT size of
// its loads and stores (development tool).  512 threads, 64 KiB in and 64 KiB out per step (the shape of the 8192-point
// kernel), next step's 16 loads prefetched, 4 memory instructions at a time; a wave instruction of 64 x 8 bytes covers
// 64 / LSEG (loads) or 64 / SSEG (stores) separate runs of LSEG / SSEG lanes, 512 bytes apart.  Optional s_barrier per step.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_seg tools/ubench_seg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../kofft_amd/csrc/fft_device.hip.h"
using namespace kofft;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SEG>
__device__ __forceinline__ int elem_of(int w, int u, int lane)
{
    constexpr int PER = 64 / SEG;  // runs per instruction = column blocks per row
    const int I = w * 16 + u, colblock = I % PER, rowgroup = I / PER;
    return (rowgroup * PER + lane / SEG) * 64 + colblock * SEG + lane % SEG;
}

// NPTS = points per step (8192: 512 threads, one workgroup per CU; 4096: 256 threads, two per CU); CHUNK = memory
// instructions of one kind issued back to back (16 = all loads, then all stores: the burst form of fft_persist_kernel)
template <int LSEG, int SSEG, bool BARRIER, int NPTS, int CHUNK>
__global__ __launch_bounds__(NPTS / 16, 2) void seg_copy(const cpx<float> *in, cpx<float> *out, size_t batch)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    cpx<float> ra[16], rb[16];
    size_t base = blockIdx.x;
    if (base >= batch) return;
    const size_t step = gridDim.x;
    auto issue = [&](cpx<float> *dst, size_t b, int u0) {
        const rsrc_t d = make_rsrc(in + (b < batch ? b : 0) * NPTS, b < batch ? (unsigned)NPTS * 8u : 0u);
#pragma unroll
        for (int u = u0; u < u0 + CHUNK; ++u) dst[u] = buf_load_cpx<float, AUX_NT>(d, elem_of<LSEG>(w, u, lane) * 8, 0);
    };
    auto store = [&](const cpx<float> *src, size_t b, int u0) {
        const rsrc_t d = make_rsrc(out + b * NPTS, (unsigned)NPTS * 8u);
#pragma unroll
        for (int u = u0; u < u0 + CHUNK; ++u) buf_store_cpx<float>(src[u], d, elem_of<SSEG>(w, u, lane) * 8, 0);
    };
#pragma unroll
    for (int c = 0; c < 16; c += CHUNK) issue(ra, base, c);
#define STEP(CUR, NXT)                                                                  \
    {                                                                                   \
        const size_t nb = base + step;                                                  \
        _Pragma("unroll") for (int c = 0; c < 16; c += CHUNK) {                         \
            __builtin_amdgcn_sched_barrier(0);                                          \
            issue(NXT, nb, c);                                                          \
            __builtin_amdgcn_sched_barrier(0);                                          \
            if (BARRIER && c == (16 / CHUNK / 2) * CHUNK) __syncthreads();              \
            store(CUR, base, c);                                                        \
        }                                                                               \
        if (nb >= batch) break;                                                         \
        base = nb;                                                                      \
    }
    for (;;) {
        STEP(ra, rb)
        STEP(rb, ra)
    }
#undef STEP
}

template <int LSEG, int SSEG, bool BARRIER, int NPTS = 8192, int CHUNK = 4>
void run(const cpx<float> *src, cpx<float> *dst, size_t bytes_total)
{
    const size_t batch = bytes_total / (NPTS * 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto k = seg_copy<LSEG, SSEG, BARRIER, NPTS, CHUNK>;
    const unsigned grid = 256 * (8192 / NPTS);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NPTS / 16), 0, 0, src, dst, batch);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NPTS / 16), 0, 0, src, dst, batch);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 20;
    const double bytes = 2.0 * bytes_total;
    printf("%4d points per step, chunks of %2d, loads in runs of %3d B, stores %3d B, barrier %d: %.4f ms  %.0f GB/s  frac %.3f\n", NPTS, CHUNK,
           LSEG * 8, SSEG * 8, (int)BARRIER, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
}

int main()
{
    const size_t bytes = (size_t)8192 * 65536;
    cpx<float> *src, *dst;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 1, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        run<64, 64, true, 8192, 16>(src, dst, bytes);
        run<64, 64, true, 8192, 4>(src, dst, bytes);
        run<64, 64, true, 8192, 1>(src, dst, bytes);
        run<64, 64, false, 8192, 4>(src, dst, bytes);
        run<64, 64, true, 4096, 16>(src, dst, bytes);
        run<64, 64, false, 4096, 16>(src, dst, bytes);
        run<64, 64, true, 4096, 4>(src, dst, bytes);
        run<64, 64, false, 4096, 4>(src, dst, bytes);
        run<64, 64, true, 4096, 1>(src, dst, bytes);
        run<8, 16, true, 8192, 4>(src, dst, bytes);
        run<8, 16, true, 8192, 16>(src, dst, bytes);
        run<4, 16, true, 8192, 4>(src, dst, bytes);
        run<8, 8, true, 8192, 4>(src, dst, bytes);
        run<4, 64, true, 8192, 4>(src, dst, bytes);
    }
    return 0;
}
