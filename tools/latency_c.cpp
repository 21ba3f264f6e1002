// latency_c.cpp -- per-call latency of the host-pointer C ABI without any Python in the way (development tool).
// build: g++ -O2 -std=c++17 -Iinclude tools/latency_c.cpp -o tools/latency_c -Lkofft_amd/lib -lkofft_hip -Wl,-rpath,$PWD/kofft_amd/lib
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kofft_hip.h"

int main(int argc, char **argv)
{
    kofft_hip_ctx *ctx = nullptr;
    if (kofft_hip_create(0, &ctx) != 0) { printf("create failed\n"); return 1; }
    for (size_t n : {64, 1024, 4096, 16384, 65536}) {
        std::vector<float> x0(2 * n), x(2 * n);
        for (size_t i = 0; i < 2 * n; ++i) x0[i] = (float)rand() / RAND_MAX - 0.5f;
        for (int i = 0; i < 50; ++i) { x = x0; kofft_hip_fft_c32(ctx, x.data(), n, 1, 0); }
        const int reps = 2000;
        double busy = 0;
        for (int i = 0; i < reps; ++i) {
            x = x0;
            auto t0 = std::chrono::steady_clock::now();
            int rc = kofft_hip_fft_c32(ctx, x.data(), n, 1, 0);
            busy += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rc) { printf("rc %d\n", rc); return 1; }
        }
        printf("n=%6zu: %.2f us per kofft_hip_fft_c32 call (host memory, C caller)\n", n, busy / reps);
    }
    kofft_hip_destroy(ctx);
    return 0;
}
