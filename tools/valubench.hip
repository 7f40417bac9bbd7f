// valubench.hip -- VALU issue rates on gfx950 for the instruction kinds the FFT butterflies use.
// Each kernel runs N dependent-free instructions per lane (8 independent accumulators) in a loop;
// reports cycles per wave-instruction per SIMD at 1..8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/valubench tools/valubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float *out, int iters, unsigned long long *cyc)
{
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = (v2f){a[i], a[i] + 1.f}; }
    const float c = 1.0001f, d = 0.9999f;
    const v2f pc = {c, d};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
                if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
                if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 7) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 256 * 2048 * 4)); CK(hipMalloc(&cyc, 8));
    const char *names[] = {"v_add_f32", "v_fma_f32", "v_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                           "v_mov_b32", "v_sub_f32"};
    const int iters = 2000;
    for (int kind = 0; kind < 8; kind++) {
        printf("%-14s", names[kind]);
        for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: block = 256 * wps threads (4 SIMDs), 1 block per CU
            const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
            const int blocks = 256 * (256 * wps / threads);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            // cycles per instruction seen by one wave; per-SIMD throughput = that / waves per SIMD
            const double per_wave = (double)c / (iters * 32.0);
            const double winst = (double)blocks * (threads / 64) * iters * 32.0;   // wave-instructions in the launch
            const double per_simd_ns = ms * 1e6 / (winst / 1024.0);               // ns of wall time per wave-inst per SIMD
            printf("  %dw: %.2f cyc/wave, wall %.3f ns/inst/SIMD", wps, per_wave, per_simd_ns);
        }
        printf("\n");
    }
    return 0;
}
