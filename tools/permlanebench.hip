// permlanebench.hip -- v_permlane16_swap_b32 / v_permlane32_swap_b32 (new on gfx950): what they do to the lanes, and what they
// cost beside v_mov_b32 and ds_write_b64 + ds_read_b64 (the LDS exchange they could replace between two wave-local FFT
// stages: a 2 x 2 block transpose over (register pair, 16-lane row bit) per instruction).
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/permlanebench tools/permlanebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__global__ void semantics(unsigned *out)
{
    const unsigned a = threadIdx.x, b = threadIdx.x + 1000;
    const v2u r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const v2u s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r.x; out[64 + threadIdx.x] = r.y; out[128 + threadIdx.x] = s.x; out[192 + threadIdx.x] = s.y;
}

template <int KIND>
__global__ void k(float *out, int iters)
{
    extern __shared__ char smem[];
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
    const unsigned addr = threadIdx.x * 8;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                if (KIND == 0) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
                if (KIND == 1) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
                if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[i + 1]));
                if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[i + 1]));
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

int main()
{
    unsigned *d; CK(hipMalloc(&d, 256 * 4));
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    const char *nm[4] = {"permlane16_swap: new vdst (a = lane, b = 1000 + lane)", "permlane16_swap: new src0", "permlane32_swap: new vdst", "permlane32_swap: new src0"};
    for (int q = 0; q < 4; q++) {
        printf("%s\n  ", nm[q]);
        for (int l = 0; l < 64; l += 4) printf("%u ", h[q * 64 + l]);
        printf("\n");
    }
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4));
    const char *names[] = {"v_permlane16_swap_b32", "v_permlane32_swap_b32", "v_mov_b32", "v_add_f32"};
    const int iters = 2000;
    for (int kind = 0; kind < 4; kind++) {
        printf("%-22s", names[kind]);
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, iters); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double winst = 256.0 * (threads / 64) * iters * 32.0;
            printf("  %dw/SIMD: %.3f ns per instruction and SIMD", wps, ms * 1e6 / (winst / 1024.0));
        }
        printf("\n");
    }
    return 0;
}
