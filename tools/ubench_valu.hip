// ubench_valu.hip -- issue cost of the packed-f32 instructions the butterflies are made of (development tool).
// One workgroup of 64 * WAVES threads per CU; every wave runs REPS x 64 independent instructions of one kind; clocks by s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void k(float *out, unsigned long long *clk, int reps)
{
    v2f a[8], b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)threadIdx.x + i, (float)i};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(a[i]) : "v"(a[i]), "v"(b));
                if (KIND == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(a[i].x), "v"(b.x));
                if (KIND == 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i].x) : "v"(a[i].x), "v"(c.x));
                if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(c));
                if (KIND == 6) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i].x) : "v"(a[i].x), "v"(b.x));
                if (KIND == 7) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a[i]) : "v"(a[i]), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
int run(const char *name, int waves)
{
    float *out;
    unsigned long long *clk, h[256 * 16];
    CK(hipMalloc(&out, 256 * 64 * waves * 4));
    CK(hipMalloc(&clk, 256 * waves * 8));
    const int reps = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * waves), 0, 0, out, clk, reps);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64 * waves), 0, 0, out, clk, reps);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, clk, 256 * waves * 8, hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < 256 * waves; ++i) m += (double)h[i];
    m /= 256 * waves;
    const double per = m / (reps * 64.0);
    printf("%-34s %2d waves/CU (%d per SIMD): %.2f clocks per instruction per wave, %.2f per SIMD-instruction\n", name, waves, waves / 4, per,
           per / (waves / 4.0));
    CK(hipFree(out));
    CK(hipFree(clk));
    return 0;
}

int main()
{
    for (int waves : {4, 8}) {
        run<0>("v_pk_mul_f32", waves);
        run<2>("v_pk_mul_f32 op_sel", waves);
        run<1>("v_pk_add_f32", waves);
        run<7>("v_pk_add_f32 neg", waves);
        run<5>("v_pk_fma_f32", waves);
        run<3>("v_mul_f32", waves);
        run<4>("v_add_f32", waves);
        run<6>("v_xor_b32", waves);
    }
    return 0;
}
