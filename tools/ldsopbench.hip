// ldsopbench.hip -- LDS cycles of the access forms the kernels use, per 16 bytes a lane moves: hipcc's load / store
// optimiser merges neighbouring ds_read_b64 / ds_write_b64 into ds_read2_b64 / ds_write2_b64 (one address register, two
// offsets); MI355X_MICROARCH.md's LDS table prices ds_read2_b64 at 8 LDS cycles per wave-instruction (two accesses served
// in 16-lane groups) against 2 x 2 for two ds_read_b64 (32-lane groups).  Measured here on conflict-free images
// (lane l at byte 8 l, second access 1152 bytes on: a row of the tile image), every CU streaming, 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/ldsopbench tools/ldsopbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
    extern __shared__ char smem[];
    const unsigned addr = (unsigned)((threadIdx.x >> 6) * 4608 + (threadIdx.x & 63) * 8);   // a wave's 4.5 KiB
    v2f a = {1.f, 2.f}, b = {3.f, 4.f};
    v4f q = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) {   // two ds_read_b64
                asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:1152" : "=v"(a), "=v"(b) : "v"(addr) : "memory");
            } else if (KIND == 1) {   // one ds_read2_b64 (offsets in units of 8 bytes)
                asm volatile("ds_read2_b64 %0, %1 offset1:144" : "=v"(q) : "v"(addr) : "memory");
            } else if (KIND == 2) {   // two ds_write_b64
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:1152" :: "v"(addr), "v"(a), "v"(b) : "memory");
            } else if (KIND == 3) {   // one ds_write2_b64
                asm volatile("ds_write2_b64 %0, %1, %2 offset1:144" :: "v"(addr), "v"(a), "v"(b) : "memory");
            } else if (KIND == 4) {   // one ds_read_b128 (16 bytes per lane, contiguous)
                asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(addr * 2) : "memory");
            } else if (KIND == 5) {   // two ds_read_b32 vs
                float x, y;
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:1152" : "=v"(x), "=v"(y) : "v"(addr >> 1) : "memory");
                a.x += x; a.y += y;
            } else if (KIND == 6) {   // one ds_read2_b32
                v2f x;
                asm volatile("ds_read2_b32 %0, %1 offset1:72" : "=v"(x) : "v"(addr >> 1) : "memory");
                a += x;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + b.x + b.y + q.x + q.y + q.z + q.w;
}

int main()
{
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4));
    const char *names[] = {"2 x ds_read_b64", "ds_read2_b64", "2 x ds_write_b64", "ds_write2_b64", "ds_read_b128", "2 x ds_read_b32", "ds_read2_b32"};
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 7; kind++) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            switch (kind) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 16 * 4608 * 2, 0, out, iters); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 16 * 4608 * 2, 0, out, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 16 * 4608 * 2, 0, out, iters); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 16 * 4608 * 2, 0, out, iters); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 16 * 4608 * 2, 0, out, iters); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 16 * 4608 * 2, 0, out, iters); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(1024), 16 * 4608 * 2, 0, out, iters); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        // per CU: 16 waves x iters x 8 pairs
        const double pairs = 16.0 * iters * 8;
        printf("%-18s %7.2f ns per pair and CU  (= %5.1f cycles at 2.3 GHz; %s per lane and pair)\n", names[kind], ms * 1e6 / pairs, ms * 1e6 / pairs * 2.3,
               kind >= 5 ? "8 bytes" : "16 bytes");
    }
    return 0;
}
