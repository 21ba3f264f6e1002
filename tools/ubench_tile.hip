// ubench_tile.hip -- streaming ceilings for the column-tile access shape of the large-n path (development tool).
// A "tile" = 1024 rows x W columns of 16-byte elements inside a 1024 x 1024 matrix (one 2^20-pt c64 transform);
// lanes run over columns first.  Copy tile -> same position in dst.  W*16 bytes contiguous per row.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// one tile per workgroup; THREADS = 64*W; thread (tau = tid / W, col = tid % W) moves rows tau + 64*c, c < 16
template <int W>
__global__ __launch_bounds__(64 * W) void tile_copy(const d2* __restrict__ in, d2* __restrict__ out) {
    const int tid = threadIdx.x, tau = tid / W, col = tid % W;
    const size_t tiles_per_xf = 1024 / W;
    const size_t xf = blockIdx.x / tiles_per_xf, t = blockIdx.x % tiles_per_xf;
    const size_t base = xf * (1u << 20) + t * W + col;
    d2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = in[base + (size_t)(tau + 64 * c) * 1024];
#pragma unroll
    for (int c = 0; c < 16; ++c) out[base + (size_t)(tau + 64 * c) * 1024] = v[c];
}
// XCD-aware pairing: blocks b and b+8 run on the same XCD (round-robin dispatch) at about the same time; give them the
// two 64-byte halves of the same 128-byte lines so the halves meet in that XCD's L2.
__device__ __forceinline__ size_t pair_remap(size_t b) { return 16 * (b / 16) + 2 * (b % 8) + ((b / 8) % 2); }
__global__ __launch_bounds__(256) void tile_copy_w4_paired(const d2* __restrict__ in, d2* __restrict__ out) {
    constexpr int W = 4;
    const int tid = threadIdx.x, tau = tid / W, col = tid % W;
    const size_t blk = pair_remap(blockIdx.x);
    const size_t xf = blk / 256, t = blk % 256;
    const size_t base = xf * (1u << 20) + t * W + col;
    d2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = in[base + (size_t)(tau + 64 * c) * 1024];
#pragma unroll
    for (int c = 0; c < 16; ++c) out[base + (size_t)(tau + 64 * c) * 1024] = v[c];
}
__global__ __launch_bounds__(256) void rows_to_cols_w4_paired(const d2* __restrict__ in, d2* __restrict__ out) {
    constexpr int W = 4;
    const int tid = threadIdx.x, tau = tid / W, k = tid % W;
    const size_t blk = pair_remap(blockIdx.x);
    const size_t xf = blk / 256, t = blk % 256;
    const size_t K = t * W + k;
    d2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = in[xf * (1u << 20) + K * 1024 + (size_t)(tau + 64 * c)];
#pragma unroll
    for (int c = 0; c < 16; ++c) out[xf * (1u << 20) + (size_t)(tau + 64 * c) * 1024 + K] = v[c];
}
// persistent + register prefetch
template <int W>
__global__ __launch_bounds__(64 * W) void tile_copy_persist(const d2* __restrict__ in, d2* __restrict__ out, size_t ntiles) {
    const int tid = threadIdx.x, tau = tid / W, col = tid % W;
    const size_t tiles_per_xf = 1024 / W;
    auto base_of = [&](size_t tile) { return (tile / tiles_per_xf) * (size_t)(1u << 20) + (tile % tiles_per_xf) * W + col; };
    size_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    d2 cur[16], nxt[16];
    {
        const size_t b = base_of(tile);
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = in[b + (size_t)(tau + 64 * c) * 1024];
    }
    for (;;) {
        const size_t ntile = tile + gridDim.x;
        const bool more = ntile < ntiles;
        if (more) {
            const size_t b = base_of(ntile);
#pragma unroll
            for (int c = 0; c < 16; ++c) nxt[c] = in[b + (size_t)(tau + 64 * c) * 1024];
        }
        const size_t b = base_of(tile);
#pragma unroll
        for (int c = 0; c < 16; ++c) out[b + (size_t)(tau + 64 * c) * 1024] = cur[c];
        if (!more) break;
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = nxt[c];
        tile = ntile;
    }
}
// row-tile -> transposed column-tile (factor B's shape): read W contiguous rows, write W-wide column tile
template <int W>
__global__ __launch_bounds__(64 * W) void tile_rows_to_cols(const d2* __restrict__ in, d2* __restrict__ out) {
    const int tid = threadIdx.x, tau = tid / W, k = tid % W;
    const size_t tiles_per_xf = 1024 / W;
    const size_t xf = blockIdx.x / tiles_per_xf, t = blockIdx.x % tiles_per_xf;
    const size_t K = t * W + k;
    d2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = in[xf * (1u << 20) + K * 1024 + (size_t)(tau + 64 * c)];
#pragma unroll
    for (int c = 0; c < 16; ++c) out[xf * (1u << 20) + (size_t)(tau + 64 * c) * 1024 + K] = v[c];
}
int main() {
    const size_t nxf = 256; const size_t bytes = nxf * (1u << 20) * 16;
    d2 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { std::string name; std::function<void()> f; std::vector<float> ms; };
    std::vector<V> vs;
    vs.push_back({"tile_copy W=4 (256 thr)", [=]{ hipLaunchKernelGGL(tile_copy<4>, dim3(nxf * 256), dim3(256), 0, 0, a, b); }});
    vs.push_back({"tile_copy W=8 (512 thr)", [=]{ hipLaunchKernelGGL(tile_copy<8>, dim3(nxf * 128), dim3(512), 0, 0, a, b); }});
    vs.push_back({"tile_copy W=16 (1024 thr)", [=]{ hipLaunchKernelGGL(tile_copy<16>, dim3(nxf * 64), dim3(1024), 0, 0, a, b); }});
    for (int g : {1, 2, 4}) {
        vs.push_back({"tile_copy_persist W=4 x" + std::to_string(g * 2), [=]{ hipLaunchKernelGGL(tile_copy_persist<4>, dim3(256 * g * 2), dim3(256), 0, 0, a, b, nxf * 256); }});
        vs.push_back({"tile_copy_persist W=8 x" + std::to_string(g), [=]{ hipLaunchKernelGGL(tile_copy_persist<8>, dim3(256 * g), dim3(512), 0, 0, a, b, nxf * 128); }});
    }
    vs.push_back({"tile_copy W=4 XCD-paired", [=]{ hipLaunchKernelGGL(tile_copy_w4_paired, dim3(nxf * 256), dim3(256), 0, 0, a, b); }});
    vs.push_back({"rows_to_cols W=4 XCD-paired", [=]{ hipLaunchKernelGGL(rows_to_cols_w4_paired, dim3(nxf * 256), dim3(256), 0, 0, a, b); }});
    vs.push_back({"rows_to_cols W=4", [=]{ hipLaunchKernelGGL(tile_rows_to_cols<4>, dim3(nxf * 256), dim3(256), 0, 0, a, b); }});
    vs.push_back({"rows_to_cols W=8", [=]{ hipLaunchKernelGGL(tile_rows_to_cols<8>, dim3(nxf * 128), dim3(512), 0, 0, a, b); }});
    for (int r = 0; r < 10; ++r) for (auto& v : vs) { hipEventRecord(e0); v.f(); hipEventRecord(e1); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (r >= 3) v.ms.push_back(t); }
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end()); float m = v.ms[v.ms.size() / 2]; printf("%-32s median %.4f ms -> %.0f GB/s (x4 for 1024 transforms: %.2f ms per pass)\n", v.name.c_str(), m, 2.0 * bytes / m / 1e6, m * 4); }
    return 0;
}
