// ubench_mall2.hip -- does the Infinity Cache serve the large-n path's ACCESS PATTERNS?  (development tool, round 2)
// Per chunk of C transforms (2^20 c64 points each = 16 MiB): pass A copies column tiles X[c] -> M (128-byte segments, 16 KiB
// stride, like the first factor), pass B copies rows of M -> transposed column tiles of Y[c] (like the last factor).
// "reuse": M is ONE buffer of C transforms (stays in the 256 MiB cache if it can); "stream": M advances with c (never reused).
// X loads and Y stores carry the streaming hint, M accesses are plain.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one tile (8 columns x 1024 rows) per 512-thread workgroup, all loads then all stores
__global__ __launch_bounds__(512) void cols_copy(const d2 *__restrict__ in, d2 *__restrict__ out)
{
    const int tid = threadIdx.x, tau = tid / 8, col = tid % 8;
    const size_t xf = blockIdx.x / 128, t = blockIdx.x % 128;
    const size_t base = xf * (size_t(1) << 20) + t * 8 + col;
    d2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = __builtin_nontemporal_load(in + base + (size_t)(tau + 64 * c) * 1024);
#pragma unroll
    for (int c = 0; c < 16; ++c) out[base + (size_t)(tau + 64 * c) * 1024] = v[c];
}
__global__ __launch_bounds__(512) void rows_to_cols(const d2 *__restrict__ in, d2 *__restrict__ out)
{
    const int tid = threadIdx.x, tau = tid / 8, k = tid % 8;
    const size_t xf = blockIdx.x / 128, t = blockIdx.x % 128;
    const size_t K = t * 8 + k;
    d2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = in[xf * (size_t(1) << 20) + K * 1024 + (size_t)(tau + 64 * c)];
#pragma unroll
    for (int c = 0; c < 16; ++c) __builtin_nontemporal_store(v[c], out + xf * (size_t(1) << 20) + (size_t)(tau + 64 * c) * 1024 + K);
}

int main()
{
    const size_t nxf = 256, xfe = size_t(1) << 20;  // 256 transforms = 4 GiB per buffer
    const size_t bytes = nxf * xfe * 16;
    d2 *x, *y, *m;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&y, bytes));
    CK(hipMalloc(&m, bytes));
    CK(hipMemset(x, 1, bytes));
    CK(hipMemset(y, 0, bytes));
    CK(hipMemset(m, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](size_t C, bool reuse) {
        for (size_t c0 = 0; c0 + C <= nxf; c0 += C) {
            d2 *mm = reuse ? m : m + c0 * xfe;
            hipLaunchKernelGGL(cols_copy, dim3((unsigned)(C * 128)), dim3(512), 0, 0, x + c0 * xfe, mm);
            hipLaunchKernelGGL(rows_to_cols, dim3((unsigned)(C * 128)), dim3(512), 0, 0, (const d2 *)mm, y + c0 * xfe);
        }
    };
    for (int r = 0; r < 20; ++r) run(nxf, false);  // clock ramp
    CK(hipDeviceSynchronize());
    for (size_t C : {2, 4, 8, 16, 32, 256}) {
        for (int reuse = 0; reuse < 2; ++reuse) {
            if (C == 256 && reuse) continue;
            std::vector<float> ms;
            for (int r = 0; r < 7; ++r) {
                CK(hipEventRecord(e0));
                run(C, reuse != 0);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (r >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            const float med = ms[ms.size() / 2];
            printf("chunk %3zu transforms (%4zu MiB)  M %s  %.3f ms per 256 transforms  -> x4 = %.2f ms per 1024 (config 5's two passes)\n", C, C * 16,
                   reuse ? "reused  " : "streamed", med, med * 4);
            fflush(stdout);
        }
    }
    return 0;
}
