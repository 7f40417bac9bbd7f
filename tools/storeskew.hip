// storeskew.hip -- why are the tile members r = 3, 11, 19, 27 of every team slower in the phases that store?
// 256 workgroups (32 per XCD, rank read at run time); workgroup of rank r stores (and/or loads) the
// column-tile pattern of the fused launch -- 512 rows x 128 bytes at a row stride of 4 KiB, column
// offset 128 r -- into its XCD's 2 MiB buffer, `reps` times, and reports its own time per repetition.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/storeskew tools/storeskew.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ unsigned g_rank[8];

template <int MODE>   // 0: 8-byte stores (16 lanes per line), 1: 16-byte stores (8 lanes per line), 2: nt 16-byte loads of the input pattern
__global__ __launch_bounds__(512) void k(float *pool, const float *in, float *res, int reps, int shift)
{
    __shared__ unsigned s_x, s_r;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7;
        s_x = x;
        s_r = atomicAdd(&g_rank[x], 1u) & 31;
    }
    __syncthreads();
    const int x = s_x, r = s_r;
    char *mid = reinterpret_cast<char *>(pool) + (size_t)x * (2 << 20) + shift;
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(mid, 0, 2 << 20, 0x00020000);
    const char *src = reinterpret_cast<const char *>(in) + (size_t)(x * 32 + r) * (4 << 20);   // a private 4 MiB "channel"
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src), 0, 4 << 20, 0x00020000);
    const int w = tid >> 6, l = tid & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int it = 0; it < reps; it++) {
        if (MODE == 0) {
            for (int k = 0; k < 8; k++) {   // wave w: gates w*64 + k*8 + (l>>4)*2.. : 4 lines per instruction
                const int gate = w * 64 + k * 8 + (l >> 4) * 2 + (it & 1);
                v2u d = {(unsigned)it, (unsigned)gate};
                __builtin_amdgcn_raw_buffer_store_b64(d, rd, gate * 4096 + r * 128 + (l & 15) * 8, 0, 0);
            }
        } else if (MODE == 1) {
            for (int k = 0; k < 8; k++) {
                const int gate = w * 64 + k * 8 + (l >> 3);
                v4u d = {(unsigned)it, (unsigned)gate, 0u, 1u};
                __builtin_amdgcn_raw_buffer_store_b128(d, rd, gate * 4096 + r * 128 + (l & 7) * 16, 0, 0);
            }
        } else {
            for (int k = 0; k < 16; k++) {
                const int row = w * 8 + (l >> 3) + 64 * k;
                v4u d = __builtin_amdgcn_raw_buffer_load_b128(rs, row * 4096 + r * 128 + (l & 7) * 16, 0, 2);
                acc += __builtin_bit_cast(float, d.x);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) res[x * 32 + r] = (float)(t1 - t0) / 100.f / reps + (acc == 123.f ? 1.f : 0.f);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    float *pool, *in, *res;
    CK(hipMalloc(&pool, (16 << 20) + 8192));
    CK(hipMalloc(&in, (size_t)256 * (4 << 20)));
    CK(hipMalloc(&res, 256 * 4));
    CK(hipMemset(in, 1, (size_t)256 * (4 << 20)));
    std::vector<float> h(256);
    const char *names[3] = {"8-byte stores", "16-byte stores", "nt loads of the input pattern"};
    for (int mode = 0; mode < 3; mode++)
        for (int shift = 0; shift <= 4096; shift += 4096) {
            unsigned zero[8] = {0};
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rank), zero, sizeof(zero)));
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, pool, in, res, reps, shift); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, pool, in, res, reps, shift); break;
            default: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, pool, in, res, reps, shift); break;
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), res, 256 * 4, hipMemcpyDeviceToHost));
            printf("%s, buffer shift %d: us per repetition by rank (mean over the 8 XCDs)\n  ", names[mode], shift);
            for (int r = 0; r < 32; r++) {
                float s = 0;
                for (int x = 0; x < 8; x++) s += h[x * 32 + r];
                printf("%.2f ", s / 8);
            }
            printf("\n");
        }
    return 0;
}
