// ubench_mall.hip -- can the intermediate of a two-pass transform live in the 256 MiB Infinity Cache?  (development tool)
//
// Model of the large-n path: for every chunk c, pass A copies X[c] -> M and pass B copies M -> Y[c], M a FIXED buffer of
// `chunk` bytes reused by every chunk.  If M's lines stay in the Infinity Cache, pass B's reads (and possibly pass A's
// writes) never reach HBM and the pair costs less than two full copies.  Variants: cache hints on the streaming sides
// (X loads, Y stores) and on the M side.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copyk(const f4v *__restrict__ in, f4v *__restrict__ out, size_t n4)
{
    // persistent: 8 x 16 B per thread per step, next step prefetched
    const size_t step = (size_t)gridDim.x * 2048;
    size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    if (base >= n4) return;
    f4v cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cur[u] = NT_LD ? __builtin_nontemporal_load(in + base + 256 * u) : in[base + 256 * u];
    for (;;) {
        const size_t nb = base + step;
        const bool more = nb < n4;
        if (more) {
#pragma unroll
            for (int u = 0; u < 8; ++u) nxt[u] = NT_LD ? __builtin_nontemporal_load(in + nb + 256 * u) : in[nb + 256 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NT_ST) __builtin_nontemporal_store(cur[u], out + base + 256 * u);
            else out[base + 256 * u] = cur[u];
        }
        if (!more) break;
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        base = nb;
    }
}

int main()
{
    const size_t total = size_t(4) << 30;  // bytes in X and in Y
    void *x, *y, *m;
    CK(hipMalloc(&x, total));
    CK(hipMalloc(&y, total));
    CK(hipMalloc(&m, total));
    CK(hipMemset(x, 1, total));
    CK(hipMemset(y, 0, total));
    CK(hipMemset(m, 0, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(512), blk(256);
    auto run = [&](size_t chunk, int mode) {
        // mode 0: X,Y plain, M plain.  1: X,Y nt, M plain.  2: all nt.  3: X,Y plain, M nt
        for (size_t off = 0; off + chunk <= total; off += chunk) {  // whole chunks only: never past the end of X / Y
            const f4v *xi = (const f4v *)((char *)x + off);
            f4v *yo = (f4v *)((char *)y + off);
            f4v *mm = (f4v *)m;
            const size_t n4 = chunk / 16;
            switch (mode) {
            case 0:
                hipLaunchKernelGGL((copyk<false, false>), grid, blk, 0, 0, xi, mm, n4);
                hipLaunchKernelGGL((copyk<false, false>), grid, blk, 0, 0, (const f4v *)mm, yo, n4);
                break;
            case 1:
                hipLaunchKernelGGL((copyk<true, false>), grid, blk, 0, 0, xi, mm, n4);
                hipLaunchKernelGGL((copyk<false, true>), grid, blk, 0, 0, (const f4v *)mm, yo, n4);
                break;
            case 2:
                hipLaunchKernelGGL((copyk<true, true>), grid, blk, 0, 0, xi, mm, n4);
                hipLaunchKernelGGL((copyk<true, true>), grid, blk, 0, 0, (const f4v *)mm, yo, n4);
                break;
            default:
                hipLaunchKernelGGL((copyk<false, true>), grid, blk, 0, 0, xi, mm, n4);
                hipLaunchKernelGGL((copyk<true, false>), grid, blk, 0, 0, (const f4v *)mm, yo, n4);
                break;
            }
        }
    };
    for (int r = 0; r < 30; ++r) run(total, 1);  // clock ramp
    CK(hipDeviceSynchronize());
    const char *names[] = {"XY plain, M plain", "XY nt,    M plain", "all nt           ", "XY plain, M nt   "};
    for (size_t mb : {16, 32, 64, 128, 256, 512, 4096}) {  // divisors of the 4 GiB total
        const size_t chunk = mb << 20;
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<float> ms;
            for (int r = 0; r < 7; ++r) {
                hipEventRecord(e0);
                run(chunk, mode);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float t;
                hipEventElapsedTime(&t, e0, e1);
                if (r >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            const float med = ms[ms.size() / 2];
            // "algorithmic" = read X once + write Y once
            fflush(stdout);
            printf("M = %4zu MiB  %s  %.3f ms  -> %.0f GB/s end to end (X read + Y written), %.0f GB/s counting M too\n", mb, names[mode], med,
                   2.0 * total / med / 1e6, 4.0 * total / med / 1e6);
        }
    }
    return 0;
}
