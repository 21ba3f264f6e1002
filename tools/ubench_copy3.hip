// ubench_copy3.hip -- read+write streaming ceiling at 8 vs 16 bytes per lane, in the persistent / prefetching structure
// of fft_persist_kernel (one 32 KiB "transform" per 256-thread workgroup step).  Development tool (round 2):
// decides whether re-assigning registers so that global accesses are 16 B per lane is worth it.
//   copy8   : 16 x 8-B loads + 16 x 8-B stores per thread and step (what the kernels do today)
//   copy16  : 8 x 16-B loads + 8 x 16-B stores
//   copy16s : as copy16 plus the v_permlane32_swap re-pairing on both sides (what a kernel would need to turn
//             "two adjacent elements of one chunk" into "one element of two chunks" and back)
//   mix     : 16-B loads, 8-B stores and vice versa
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <functional>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDW, int STW, bool SWAP>
__global__ __launch_bounds__(256, 2) void copy_persist(const float *__restrict__ in, float *__restrict__ out, int nchunks)
{
    // chunk = 8192 floats (32 KiB).  LDW / STW = bytes per lane of the loads / stores.
    const int t = threadIdx.x;
    int ch = blockIdx.x;
    if (ch >= nchunks) return;
    f2v cur[16], nxt[16];
    auto load = [&](f2v *r, int c) {
        const float *base = in + (size_t)c * 8192;
        if (LDW == 8) {
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(base) + t + 256 * u);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(base) + t + 256 * u);
                r[2 * u] = f2v{v.x, v.y};
                r[2 * u + 1] = f2v{v.z, v.w};
            }
        }
    };
    auto swap_pairs = [&](f2v *r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r[2 * u].x), __builtin_bit_cast(unsigned, r[2 * u + 1].x), false, false);
            auto b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r[2 * u].y), __builtin_bit_cast(unsigned, r[2 * u + 1].y), false, false);
            r[2 * u] = f2v{__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, b[0])};
            r[2 * u + 1] = f2v{__builtin_bit_cast(float, a[1]), __builtin_bit_cast(float, b[1])};
        }
    };
    load(cur, ch);
    for (;;) {
        const int nch = ch + gridDim.x;
        const bool more = nch < nchunks;
        if (more) load(nxt, nch);
        if (SWAP) {
            swap_pairs(cur);
#pragma unroll
            for (int u = 0; u < 16; ++u) cur[u] = f2v{cur[u].x + 1.0f, cur[u].y};  // something between the two swaps
            swap_pairs(cur);
        }
        float *ob = out + (size_t)ch * 8192;
        if (STW == 8) {
#pragma unroll
            for (int u = 0; u < 16; ++u) __builtin_nontemporal_store(cur[u], reinterpret_cast<f2v *>(ob) + t + 256 * u);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                __builtin_nontemporal_store(f4v{cur[2 * u].x, cur[2 * u].y, cur[2 * u + 1].x, cur[2 * u + 1].y}, reinterpret_cast<f4v *>(ob) + t + 256 * u);
        }
        if (!more) break;
#pragma unroll
        for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
        ch = nch;
    }
}

// config 3's access pattern without any arithmetic: one wavefront per row, 8 KiB in (16 x 512-byte loads), 8200 bytes out
// (17 line-aligned 512-byte stores, the first and last partly masked), next row prefetched.
template <bool ALIGNED>
__global__ __launch_bounds__(256, 2) void rfft_pattern(const f2v *__restrict__ in, f2v *__restrict__ out, int nrows)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    int row = wave;
    if (row >= nrows) return;
    f2v cur[16], nxt[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) cur[u] = __builtin_nontemporal_load(in + (size_t)row * 1024 + lane + 64 * u);
    for (;;) {
        const int nrow = row + nwaves;
        const bool more = nrow < nrows;
        if (more) {
#pragma unroll
            for (int u = 0; u < 16; ++u) nxt[u] = __builtin_nontemporal_load(in + (size_t)nrow * 1024 + lane + 64 * u);
        }
        f2v *orow = out + (size_t)row * 1025;
        const int a = ALIGNED ? (int)(((reinterpret_cast<size_t>(out) >> 3) + (size_t)row * 1025) & 15) : 0;
        const int k0 = lane - a;
#pragma unroll
        for (int g = 0; g <= 16; ++g) {
            const int k = k0 + 64 * g;
            if (k >= 0 && k <= 1024) __builtin_nontemporal_store(cur[g & 15], orow + k);
        }
        if (!more) break;
#pragma unroll
        for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
        row = nrow;
    }
}

// plain grid-sized copy, 16 B per lane, 4 per thread
__global__ __launch_bounds__(256) void copy16_flat(const f4v *__restrict__ in, f4v *__restrict__ out, size_t n4)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f4v a = __builtin_nontemporal_load(in + i), b = __builtin_nontemporal_load(in + i + stride);
        f4v c = __builtin_nontemporal_load(in + i + 2 * stride), d = __builtin_nontemporal_load(in + i + 3 * stride);
        __builtin_nontemporal_store(a, out + i);
        __builtin_nontemporal_store(b, out + i + stride);
        __builtin_nontemporal_store(c, out + i + 2 * stride);
        __builtin_nontemporal_store(d, out + i + 3 * stride);
    }
    for (; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

int main()
{
    const int nchunks = 65536;
    const size_t bytes = (size_t)nchunks * 32768;
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct V { std::string name; std::function<void()> f; std::vector<float> ms; };
    std::vector<V> vs;
    const float *ia = (const float *)a;
    float *ob = (float *)b;
    for (int g : {2, 3, 4}) {
        const dim3 grid(256 * g), blk(256);
        vs.push_back({"copy8   x" + std::to_string(g), [=] { hipLaunchKernelGGL((copy_persist<8, 8, false>), grid, blk, 0, 0, ia, ob, nchunks); }, {}});
        vs.push_back({"copy16  x" + std::to_string(g), [=] { hipLaunchKernelGGL((copy_persist<16, 16, false>), grid, blk, 0, 0, ia, ob, nchunks); }, {}});
        vs.push_back({"copy16s x" + std::to_string(g), [=] { hipLaunchKernelGGL((copy_persist<16, 16, true>), grid, blk, 0, 0, ia, ob, nchunks); }, {}});
        vs.push_back({"ld16st8 x" + std::to_string(g), [=] { hipLaunchKernelGGL((copy_persist<16, 8, false>), grid, blk, 0, 0, ia, ob, nchunks); }, {}});
        vs.push_back({"ld8st16 x" + std::to_string(g), [=] { hipLaunchKernelGGL((copy_persist<8, 16, false>), grid, blk, 0, 0, ia, ob, nchunks); }, {}});
    }
    for (int g : {8, 16, 32})
        vs.push_back({"flat16  x" + std::to_string(g), [=] { hipLaunchKernelGGL(copy16_flat, dim3(256 * g), dim3(256), 0, 0, (const f4v *)a, (f4v *)b, bytes / 16); }, {}});
    {
        const int nrows = (int)(bytes / 8200);  // as many 8200-byte output rows as fit (inputs: 8192 bytes each)
        for (int g : {1, 2, 4}) {
            vs.push_back({"rfftpat  x" + std::to_string(g), [=] { hipLaunchKernelGGL((rfft_pattern<true>), dim3(256 * g), dim3(256), 0, 0, (const f2v *)a, (f2v *)b, nrows); }, {}});
            vs.push_back({"rfftpatU x" + std::to_string(g), [=] { hipLaunchKernelGGL((rfft_pattern<false>), dim3(256 * g), dim3(256), 0, 0, (const f2v *)a, (f2v *)b, nrows); }, {}});
        }
    }
    // clock ramp
    for (int r = 0; r < 200; ++r) vs[0].f();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < 15; ++r)
        for (auto &v : vs) {
            hipEventRecord(e0);
            v.f();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            if (r >= 3) v.ms.push_back(t);
        }
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float m = v.ms[v.ms.size() / 2];
        const double moved = v.name.rfind("rfftpat", 0) == 0 ? (double)(bytes / 8200) * (8192.0 + 8200.0) : 2.0 * bytes;
        printf("%-14s median %.4f ms  min %.4f  -> %.0f GB/s (%.3f of 8 TB/s)\n", v.name.c_str(), m, v.ms[0], moved / m / 1e6, moved / m / 1e6 / 8000.0);
    }
    return 0;
}
