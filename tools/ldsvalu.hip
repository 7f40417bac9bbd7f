// ldsvalu.hip -- do vector arithmetic and LDS traffic overlap on a gfx950 CU, or do they add up?
// One workgroup of 1024 threads per CU (4 waves per SIMD, as the fused launch has).  Per loop iteration a wave issues
// NV independent v_fma_f32 and NL ds accesses of 8 bytes per lane (conflict-free, alternating write / read).
//   mode 0: every wave does both (interleaved in one instruction stream)
//   mode 1: waves 0-7 only the arithmetic, waves 8-15 only the LDS accesses (two + two per SIMD), each TWICE the count
// Reported: time with arithmetic only, LDS only, both; "sum" = serialised, "max" = perfectly overlapped.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/ldsvalu tools/ldsvalu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NV, int NL, int WIDTH>
__global__ __launch_bounds__(1024) void k(float *out, int iters, int mode, int do_v, int do_l)
{
    extern __shared__ char smem[];
    const int w = threadIdx.x >> 6;
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    const float c = 1.0001f, d = 0.9999f;
    // every wave owns 64 * WIDTH bytes * 2; lane l touches bytes [l * WIDTH, +WIDTH): conflict-free for 8 and 16
    unsigned addr = (unsigned)(w * 64 * WIDTH * 2 + (threadIdx.x & 63) * WIDTH);
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    v4f x = {a[0], a[1], a[2], a[3]};
    v2f y = {a[0], a[1]};
    const bool v_on = do_v && (mode == 0 || w < 8), l_on = do_l && (mode == 0 || w >= 8);
    const int rep = mode == 0 ? 1 : 2;
    for (int it = 0; it < iters * rep; it++) {
        if (v_on) {
#pragma unroll
            for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(c), "v"(d));
        }
        if (l_on) {
#pragma unroll
            for (int i = 0; i < NL; i += 2) {
                if (WIDTH == 8) {
                    asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(y) : "memory");
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(y) : "v"(addr), "n"(64 * WIDTH) : "memory");
                } else {
                    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(x) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(64 * WIDTH) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float s = x.x + x.y + x.z + x.w + y.x + y.y;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NL, int WIDTH>
void run(float *out, const char *what)
{
    const int iters = 4000;
    for (int mode = 0; mode < 2; mode++) {
        float t[3];
        for (int cfg = 0; cfg < 3; cfg++) {
            const int dv = cfg != 1, dl = cfg != 0;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((k<NV, NL, WIDTH>), dim3(256), dim3(1024), 16 * 64 * WIDTH * 2, 0, out, iters, mode, dv, dl);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[cfg], e0, e1));
            }
        }
        // per CU and iteration: 16 waves * NV arithmetic instructions over 4 SIMDs, 16 * NL accesses of 64 * WIDTH bytes
        const double per_it = 1e6 / iters;
        printf("%-34s mode %d: arithmetic %.1f ns/it (%.2f ns per instr and SIMD), LDS %.1f ns/it (%.1f B/ns per CU), both %.1f  [sum %.1f, max %.1f]\n",
               what, mode, t[0] * per_it, t[0] * per_it / (4.0 * NV), t[1] * per_it, 16.0 * NL * 64 * WIDTH / (t[1] * per_it),
               t[2] * per_it, (t[0] + t[1]) * per_it, (t[0] > t[1] ? t[0] : t[1]) * per_it);
    }
}

int main()
{
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4));
    run<64, 16, 8>(out, "64 fma + 16 ds_b64 per wave");
    run<64, 24, 8>(out, "64 fma + 24 ds_b64 per wave");
    run<64, 32, 8>(out, "64 fma + 32 ds_b64 per wave");
    run<32, 32, 8>(out, "32 fma + 32 ds_b64 per wave");
    run<64, 8, 16>(out, "64 fma + 8 ds_b128 per wave");
    run<64, 16, 16>(out, "64 fma + 16 ds_b128 per wave");
    return 0;
}
