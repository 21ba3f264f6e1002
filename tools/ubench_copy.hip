// ubench_copy.hip -- access-shape ceilings for the 4096-pt c32 kernel (not part of the product).
// Each "transform" is a 32 KiB chunk; we time chunk copies with the candidate load/store shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: one WG (256 thr) per chunk, 16 x 8-B loads per thread at t + 256c  (shape of the generic kernel)
__global__ __launch_bounds__(256) void copy8_wg(const float2* __restrict__ in, float2* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * 4096; const int t = threadIdx.x;
    float2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = in[base + t + 256 * c];
#pragma unroll
    for (int c = 0; c < 16; ++c) out[base + t + 256 * c] = v[c];
}
// B: one WG per chunk, 8 x 16-B loads per thread
__global__ __launch_bounds__(256) void copy16_wg(const float4* __restrict__ in, float4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * 2048; const int t = threadIdx.x;
    float4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = in[base + t + 256 * c];
#pragma unroll
    for (int c = 0; c < 8; ++c) out[base + t + 256 * c] = v[c];
}
// C: 16-B loads in the "two 512-B segments per wave instruction" shape of the permlane32 design:
//    lane l<32 reads pair (2u,2u+1) of row c=2m, lane l>=32 the same pair of row c=2m+1.
__global__ __launch_bounds__(256) void copy16_split_wg(const float4* __restrict__ in, float4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * 2048; const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6; const int u = wave * 32 + (lane & 31), h = lane >> 5;
    float4 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = in[base + (size_t)(2 * m + h) * 128 + u];
#pragma unroll
    for (int m = 0; m < 8; ++m) out[base + (size_t)(2 * m + h) * 128 + u] = v[m];
}
// D/E: persistent versions with register prefetch of the next chunk
template <int W>
__global__ __launch_bounds__(256) void copy8_persist(const float2* __restrict__ in, float2* __restrict__ out, int nchunks) {
    const int t = threadIdx.x; float2 cur[16], nxt[16];
    int ch = blockIdx.x;
    if (ch >= nchunks) return;
#pragma unroll
    for (int c = 0; c < 16; ++c) cur[c] = in[(size_t)ch * 4096 + t + 256 * c];
    for (; ch < nchunks; ch += gridDim.x) {
        const int nch = ch + gridDim.x;
        if (nch < nchunks) {
#pragma unroll
            for (int c = 0; c < 16; ++c) nxt[c] = in[(size_t)nch * 4096 + t + 256 * c];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) out[(size_t)ch * 4096 + t + 256 * c] = cur[c];
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = nxt[c];
    }
}
__global__ __launch_bounds__(256) void copy16_persist(const float4* __restrict__ in, float4* __restrict__ out, int nchunks) {
    const int t = threadIdx.x; float4 cur[8], nxt[8];
    int ch = blockIdx.x;
    if (ch >= nchunks) return;
#pragma unroll
    for (int c = 0; c < 8; ++c) cur[c] = in[(size_t)ch * 2048 + t + 256 * c];
    for (; ch < nchunks; ch += gridDim.x) {
        const int nch = ch + gridDim.x;
        if (nch < nchunks) {
#pragma unroll
            for (int c = 0; c < 8; ++c) nxt[c] = in[(size_t)nch * 2048 + t + 256 * c];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) out[(size_t)ch * 2048 + t + 256 * c] = cur[c];
#pragma unroll
        for (int c = 0; c < 8; ++c) cur[c] = nxt[c];
    }
}
// F: plain grid-stride float4 copy (the guide's 6.3 TB/s shape)
__global__ __launch_bounds__(256) void copy16_flat(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

int main() {
    const int nchunks = 65536; const size_t bytes = (size_t)nchunks * 32768;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        std::vector<float> ms;
        for (int r = 0; r < 12; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (r >= 2) ms.push_back(t); }
        std::sort(ms.begin(), ms.end());
        printf("%-28s median %.4f ms  min %.4f ms  -> %.0f GB/s (median)\n", name, ms[ms.size()/2], ms[0], 2.0 * bytes / ms[ms.size()/2] / 1e6);
    };
    for (int rep = 0; rep < 2; ++rep) {
        timeit("copy8_wg", [&]{ hipLaunchKernelGGL(copy8_wg, dim3(nchunks), dim3(256), 0, 0, (const float2*)a, (float2*)b); });
        timeit("copy16_wg", [&]{ hipLaunchKernelGGL(copy16_wg, dim3(nchunks), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
        timeit("copy16_split_wg", [&]{ hipLaunchKernelGGL(copy16_split_wg, dim3(nchunks), dim3(256), 0, 0, (const float4*)a, (float4*)b); });
        for (int k : {2, 3, 4, 6, 8}) {
            char nm[64]; snprintf(nm, 64, "copy8_persist x%d/CU", k);
            timeit(nm, [&]{ hipLaunchKernelGGL(copy8_persist<0>, dim3(256 * k), dim3(256), 0, 0, (const float2*)a, (float2*)b, nchunks); });
            snprintf(nm, 64, "copy16_persist x%d/CU", k);
            timeit(nm, [&]{ hipLaunchKernelGGL(copy16_persist, dim3(256 * k), dim3(256), 0, 0, (const float4*)a, (float4*)b, nchunks); });
        }
        timeit("copy16_flat 2048 blocks", [&]{ hipLaunchKernelGGL(copy16_flat, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16); });
        timeit("copy16_flat 8192 blocks", [&]{ hipLaunchKernelGGL(copy16_flat, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16); });
    }
    return 0;
}
