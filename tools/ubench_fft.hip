// ubench_fft.hip -- variant timing harness for the 4096-pt c32 kernels (development tool, not product).
// Interleaved rounds in one process; every variant's output is compared bit-for-bit with the generic kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../kofft_amd/csrc/fft_persist.hip.h"
#include "../kofft_amd/csrc/tables.h"
using namespace kofft;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    constexpr int L = 12, N = 1 << L;
    const size_t batch = argc > 1 ? atol(argv[1]) : 65536;
    const size_t bytes = batch * N * sizeof(cpx<float>);
    std::vector<float> htw(N); kofft_tables::twiddles_f32(N, htw.data());
    cpx<float>* dtw; CK(hipMalloc(&dtw, N * sizeof(float))); CK(hipMemcpy(dtw, htw.data(), N * sizeof(float), hipMemcpyHostToDevice));
    cpx<float> *src, *ref, *out; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&ref, bytes)); CK(hipMalloc(&out, bytes));
    {   // deterministic pseudo-random input in [-1,1)
        std::vector<float> h((size_t)2 * N * 1024); uint64_t s = 0x6B6F666674ull;
        for (auto& x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }
        for (size_t off = 0; off < batch; off += 1024) CK(hipMemcpy(src + off * N, h.data(), std::min<size_t>(1024, batch - off) * N * sizeof(cpx<float>), hipMemcpyHostToDevice));
    }
    using IO = ComplexIO<float, false>;
    IO io_ref{src, ref, N, 1.0f / N}, io{src, out, N, 1.0f / N};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds1 = lds_elems(N) * sizeof(cpx<float>), lds2 = 2 * lds1;

    auto k_gen = fft_wg_kernel<float, L, 4, 256, EPI_STORE, IO>;
    auto k_p1_a = fft_persist_kernel<float, L, 4, 1, 1, IO>;
    auto k_p1_b = fft_persist_kernel<float, L, 4, 1, 3, IO>;
    auto k_p1_c = fft_persist_kernel<float, L, 4, 1, 4, IO>;
    auto k_p2_a = fft_persist_kernel<float, L, 4, 2, 1, IO>;
    auto k_p2_b = fft_persist_kernel<float, L, 4, 2, 2, IO>;
    CK(hipFuncSetAttribute((const void*)k_p2_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    CK(hipFuncSetAttribute((const void*)k_p2_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));

    hipLaunchKernelGGL(k_gen, dim3(batch), dim3(256), lds1, 0, io_ref, dtw, batch); CK(hipDeviceSynchronize());
    std::vector<char> href(std::min<size_t>(bytes, 64u << 20)), hout(href.size());
    CK(hipMemcpy(href.data(), ref, href.size(), hipMemcpyDeviceToHost));

    struct Var { const char* name; std::function<void()> launch; std::vector<float> ms; bool ok = true; };
    std::vector<Var> vars;
    vars.push_back({"generic wg (1 xf / WG)", [&]{ hipLaunchKernelGGL(k_gen, dim3(batch), dim3(256), lds1, 0, io, dtw, batch); }});
    for (int g : {2, 3, 4}) {
        char* nm = new char[64]; snprintf(nm, 64, "persist nbuf1 minw1 grid %dx256", g);
        vars.push_back({nm, [&, g]{ hipLaunchKernelGGL(k_p1_a, dim3(256 * g), dim3(256), lds1, 0, io, dtw, batch); }});
    }
    vars.push_back({"persist nbuf1 minw3 grid 3x256", [&]{ hipLaunchKernelGGL(k_p1_b, dim3(256 * 3), dim3(256), lds1, 0, io, dtw, batch); }});
    vars.push_back({"persist nbuf1 minw4 grid 4x256", [&]{ hipLaunchKernelGGL(k_p1_c, dim3(256 * 4), dim3(256), lds1, 0, io, dtw, batch); }});
    vars.push_back({"persist nbuf2 minw1 grid 2x256", [&]{ hipLaunchKernelGGL(k_p2_a, dim3(256 * 2), dim3(256), lds2, 0, io, dtw, batch); }});
    vars.push_back({"persist nbuf2 minw2 grid 2x256", [&]{ hipLaunchKernelGGL(k_p2_b, dim3(256 * 2), dim3(256), lds2, 0, io, dtw, batch); }});

    for (auto& v : vars) {  // correctness first
        CK(hipMemset(out, 0xff, bytes));
        v.launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), out, hout.size(), hipMemcpyDeviceToHost));
        v.ok = memcmp(hout.data(), href.data(), href.size()) == 0;
        // also the tail of the batch
        std::vector<char> t1(N * 8), t2(N * 8);
        CK(hipMemcpy(t1.data(), out + (batch - 1) * N, N * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(t2.data(), ref + (batch - 1) * N, N * 8, hipMemcpyDeviceToHost));
        v.ok = v.ok && memcmp(t1.data(), t2.data(), N * 8) == 0;
    }
    for (int round = 0; round < 10; ++round)
        for (auto& v : vars) {
            CK(hipEventRecord(e0)); v.launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); if (round >= 2) v.ms.push_back(t);
        }
    for (auto& v : vars) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        printf("%-34s %s  median %.4f ms  min %.4f ms  %.1f GPoints/s  %.0f GB/s\n", v.name, v.ok ? "BITEXACT" : "MISMATCH", med, v.ms[0],
               batch * N / med / 1e6, 2.0 * bytes / med / 1e6);
    }
    return 0;
}
