// ubench_fft.hip -- variant timing harness for the streaming c32 kernels (development tool, not product).
// Interleaved rounds in one process; every variant's output is compared bit-for-bit with the generic kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
#include "../kofft_amd/csrc/fft_persist.hip.h"
#include "../kofft_amd/csrc/tables.h"
using namespace kofft;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using IO = ComplexIO<float, false>;
struct Var { std::string name; std::function<void()> launch; std::vector<float> ms; bool ok = true; };

// same-bytes streaming reference: persistent copy with prefetch and non-temporal loads (the measured ceiling)
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void copy_ref(const f2v* __restrict__ in, f2v* __restrict__ out, int nchunks) {
    const int t = threadIdx.x; f2v cur[16], nxt[16];
    int ch = blockIdx.x;
    if (ch >= nchunks) return;
#pragma unroll
    for (int c = 0; c < 16; ++c) cur[c] = __builtin_nontemporal_load(&in[(size_t)ch * 4096 + t + 256 * c]);
    for (; ch < nchunks; ch += gridDim.x) {
        const int nch = ch + gridDim.x;
        if (nch < nchunks) {
#pragma unroll
            for (int c = 0; c < 16; ++c) nxt[c] = __builtin_nontemporal_load(&in[(size_t)nch * 4096 + t + 256 * c]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) out[(size_t)ch * 4096 + t + 256 * c] = cur[c];
#pragma unroll
        for (int c = 0; c < 16; ++c) cur[c] = nxt[c];
    }
}

template <int L, int BLOCK, int NBUF, int MINW>
void add_persist(std::vector<Var>& vars, IO io, const cpx<float>* dtw, size_t batch, int wg_per_cu) {
    constexpr int N = 1 << L, XPB = BLOCK / (N >> 4);
    const size_t lds = (size_t)XPB * NBUF * lds_elems(N) * sizeof(cpx<float>);
    auto k = fft_persist_kernel<float, L, 4, BLOCK, NBUF, MINW, EPI_STORE, IO>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char nm[96]; snprintf(nm, 96, "persist L%d blk%d nbuf%d minw%d grid %dx256", L, BLOCK, NBUF, MINW, wg_per_cu);
    const unsigned grid = 256u * wg_per_cu;
    vars.push_back({nm, [=]{ hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), lds, 0, io, dtw, batch); }});
}

template <int L>
int run(size_t batch) {
    constexpr int N = 1 << L;
    const size_t bytes = batch * N * sizeof(cpx<float>);
    std::vector<float> htw(N); kofft_tables::twiddles_f32(N, htw.data());
    cpx<float>* dtw; CK(hipMalloc(&dtw, N * sizeof(float))); CK(hipMemcpy(dtw, htw.data(), N * sizeof(float), hipMemcpyHostToDevice));
    cpx<float> *src, *ref, *out; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&ref, bytes)); CK(hipMalloc(&out, bytes));
    {
        const size_t chunk = (size_t)1 << 22;  // complex values per upload
        std::vector<float> h(2 * chunk); uint64_t s = 0x6B6F666674ull;
        for (auto& x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }
        for (size_t off = 0; off < batch * N; off += chunk) CK(hipMemcpy(src + off, h.data(), std::min(chunk, batch * N - off) * sizeof(cpx<float>), hipMemcpyHostToDevice));
    }
    IO io_ref{{}, src, ref, N, 1.0f / N}, io{{}, src, out, N, 1.0f / N};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int RLg = 4, BLKg = 256, XPBg = BLKg / (N >> RLg);
    const size_t ldsg = (size_t)XPBg * lds_elems(N) * sizeof(cpx<float>);
    auto k_gen = fft_wg_kernel<float, L, RLg, BLKg, EPI_STORE, IO>;
    const unsigned gridg = (unsigned)((batch + XPBg - 1) / XPBg);
    hipLaunchKernelGGL(k_gen, dim3(gridg), dim3(BLKg), ldsg, 0, io_ref, dtw, batch); CK(hipDeviceSynchronize());
    std::vector<char> href(std::min<size_t>(bytes, 64u << 20)), hout(href.size());
    CK(hipMemcpy(href.data(), ref, href.size(), hipMemcpyDeviceToHost));
    std::vector<Var> vars;
    {
        const int nchunks = (int)(bytes / 32768);
        vars.push_back({"copy reference (nt loads, prefetch) x2", [=]{ hipLaunchKernelGGL(copy_ref, dim3(512), dim3(256), 0, 0, (const f2v*)src, (f2v*)out, nchunks); }});
    }
    vars.push_back({"generic wg", [=]{ hipLaunchKernelGGL(k_gen, dim3(gridg), dim3(BLKg), ldsg, 0, io, dtw, batch); }});
    if constexpr (L == 12) {
        add_persist<12, 256, 2, 2>(vars, io, dtw, batch, 2);
        add_persist<12, 256, 1, 2>(vars, io, dtw, batch, 2);
        add_persist<12, 256, 1, 3>(vars, io, dtw, batch, 3);
    } else {
        add_persist<10, 256, 1, 2>(vars, io, dtw, batch, 2);
        add_persist<10, 256, 1, 3>(vars, io, dtw, batch, 3);
        add_persist<10, 256, 1, 4>(vars, io, dtw, batch, 4);
        add_persist<10, 128, 1, 2>(vars, io, dtw, batch, 4);
        add_persist<10, 64, 1, 2>(vars, io, dtw, batch, 8);
        add_persist<10, 512, 1, 2>(vars, io, dtw, batch, 1);
    }
    for (auto& v : vars) {
        CK(hipMemset(out, 0xff, bytes));
        v.launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), out, hout.size(), hipMemcpyDeviceToHost));
        v.ok = memcmp(hout.data(), href.data(), href.size()) == 0 || v.name.rfind("copy", 0) == 0;
        std::vector<char> t1(N * 8 * 4), t2(N * 8 * 4);
        CK(hipMemcpy(t1.data(), out + (batch - 4) * N, t1.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(t2.data(), ref + (batch - 4) * N, t2.size(), hipMemcpyDeviceToHost));
        v.ok = (v.ok && memcmp(t1.data(), t2.data(), t1.size()) == 0) || v.name.rfind("copy", 0) == 0;
    }
    for (int round = 0; round < 12; ++round)
        for (auto& v : vars) {
            CK(hipEventRecord(e0)); v.launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); if (round >= 3) v.ms.push_back(t);
        }
    for (auto& v : vars) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        printf("%-44s %s  median %.4f ms  min %.4f ms  %.1f GPoints/s  %.0f GB/s\n", v.name.c_str(), v.ok ? "BITEXACT" : "MISMATCH", med, v.ms[0],
               batch * N / med / 1e6, 2.0 * bytes / med / 1e6);
    }
    CK(hipFree(src)); CK(hipFree(ref)); CK(hipFree(out)); CK(hipFree(dtw));
    return 0;
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 12;
    if (L == 12) return run<12>(argc > 2 ? atol(argv[2]) : 65536);
    return run<10>(argc > 2 ? atol(argv[2]) : 262144);
}
