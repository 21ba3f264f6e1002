// ubench_copy2.hip -- where is the read+write streaming ceiling?  (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <functional>
typedef float f2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copy8_persist(const f2v* __restrict__ in, f2v* __restrict__ out, int nchunks) {
    const int t = threadIdx.x; f2v cur[16], nxt[16];
    int ch = blockIdx.x;
    if (ch >= nchunks) return;
#pragma unroll
    for (int c = 0; c < 16; ++c) cur[c] = NT_LD ? __builtin_nontemporal_load(&in[(size_t)ch * 4096 + t + 256 * c]) : in[(size_t)ch * 4096 + t + 256 * c];
    for (; ch < nchunks; ch += gridDim.x) {
        const int nch = ch + gridDim.x;
        if (nch < nchunks) {
#pragma unroll
            for (int c = 0; c < 16; ++c) nxt[c] = NT_LD ? __builtin_nontemporal_load(&in[(size_t)nch * 4096 + t + 256 * c]) : in[(size_t)nch * 4096 + t + 256 * c];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) { if (NT_ST) __builtin_nontemporal_store(cur[c], &out[(size_t)ch * 4096 + t + 256 * c]); else out[(size_t)ch * 4096 + t + 256 * c] = cur[c]; }
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = nxt[c];
    }
}
__global__ __launch_bounds__(256) void read_only(const f2v* __restrict__ in, f2v* __restrict__ out, int nchunks) {
    const int t = threadIdx.x; f2v acc = {0.f, 0.f};
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        f2v v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = in[(size_t)ch * 4096 + t + 256 * c];
#pragma unroll
        for (int c = 0; c < 16; ++c) { acc.x += v[c].x; acc.y += v[c].y; }
    }
    if (acc.x == 12345.678f) out[t] = acc;
}
__global__ __launch_bounds__(256) void write_only(f2v* __restrict__ out, int nchunks) {
    const int t = threadIdx.x;
    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
#pragma unroll
        for (int c = 0; c < 16; ++c) out[(size_t)ch * 4096 + t + 256 * c] = f2v{(float)ch, (float)c};
    }
}
int main() {
    const int nchunks = 65536; const size_t bytes = (size_t)nchunks * 32768;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; std::function<void()> f; double bytes; std::vector<float> ms; };
    std::vector<V> vs;
    for (int g : {2, 4, 8}) {
        char* n1 = new char[64]; snprintf(n1, 64, "copy plain x%d", g);
        vs.push_back({n1, [=]{ hipLaunchKernelGGL((copy8_persist<false, false>), dim3(256 * g), dim3(256), 0, 0, (const f2v*)a, (f2v*)b, nchunks); }, 2.0 * bytes});
        char* n2 = new char[64]; snprintf(n2, 64, "copy nt-store x%d", g);
        vs.push_back({n2, [=]{ hipLaunchKernelGGL((copy8_persist<false, true>), dim3(256 * g), dim3(256), 0, 0, (const f2v*)a, (f2v*)b, nchunks); }, 2.0 * bytes});
        char* n3 = new char[64]; snprintf(n3, 64, "copy nt-load x%d", g);
        vs.push_back({n3, [=]{ hipLaunchKernelGGL((copy8_persist<true, false>), dim3(256 * g), dim3(256), 0, 0, (const f2v*)a, (f2v*)b, nchunks); }, 2.0 * bytes});
        char* n4 = new char[64]; snprintf(n4, 64, "copy nt-both x%d", g);
        vs.push_back({n4, [=]{ hipLaunchKernelGGL((copy8_persist<true, true>), dim3(256 * g), dim3(256), 0, 0, (const f2v*)a, (f2v*)b, nchunks); }, 2.0 * bytes});
    }
    vs.push_back({"read only x8", [=]{ hipLaunchKernelGGL(read_only, dim3(256 * 8), dim3(256), 0, 0, (const f2v*)a, (f2v*)b, nchunks); }, 1.0 * bytes});
    vs.push_back({"write only x8", [=]{ hipLaunchKernelGGL(write_only, dim3(256 * 8), dim3(256), 0, 0, (f2v*)b, nchunks); }, 1.0 * bytes});
    for (int r = 0; r < 12; ++r) for (auto& v : vs) { hipEventRecord(e0); v.f(); hipEventRecord(e1); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (r >= 3) v.ms.push_back(t); }
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end()); float m = v.ms[v.ms.size() / 2]; printf("%-22s median %.4f ms  min %.4f  -> %.0f GB/s\n", v.name, m, v.ms[0], v.bytes / m / 1e6); }
    return 0;
}
