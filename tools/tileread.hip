// tileread.hip -- what does the INPUT access pattern of the fused launch cost by itself?
// 256 workgroups x 512 threads read S matrices [1024][512] complex fp32 (4 MiB each), nothing else:
//   pattern 0: one contiguous 128 KiB chunk per workgroup and step (the best the fabric gives)
//   pattern 1: the fused launch's tiles: 1024 rows x 128 B (16 columns), row stride 4 KiB; the 32 workgroups of an XCD
//              (blockIdx mod 8) take the 32 column tiles of ONE matrix, the 8 XCDs work on 8 different matrices
//   pattern 2: tiles of 512 rows x 256 B (32 columns): two row-halves x 16 column tiles per matrix
//   pattern 3: as 1, requested in quarters (rows r = j mod 4 of the lane's sixteen) with a pause between quarters
// aux: 0 plain loads, 1 nt.  Prints us per matrix and GB/s.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/tileread tools/tileread.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int PATTERN, int AUX>
__global__ __launch_bounds__(512) void k_read(const char *in, float *sink, int matrices)
{
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3;   // 32 members per XCD
    v4f acc = {0, 0, 0, 0};
    for (int mtx = xcd; mtx < matrices; mtx += 8) {
        const char *base = in + (size_t)mtx * (4u << 20);
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 4 << 20, 0x00020000);
        v4u v[16];
        if (PATTERN == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, member * 131072 + r * 8192 + tid * 16, 0, AUX);
        } else if (PATTERN == 1 || PATTERN == 3) {
            const int p0 = w * 8 + (l >> 3), cp = l & 7;
            const int voff = p0 * 4096 + member * 128 + cp * 16;
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int r = j; r < 16; r += 4) v[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 64 * r * 4096, AUX);
                if (PATTERN == 3) __builtin_amdgcn_s_sleep(40);
            }
        } else {
            const int half = member >> 4, ct = member & 15;
            const int p0 = w * 4 + (l >> 4), cp = l & 15;     // 32 positions x 16 rows = 512 rows of this half
            const int voff = (half * 512 + p0) * 4096 + ct * 256 + cp * 16;
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 32 * r * 4096, AUX);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) acc += __builtin_bit_cast(v4f, v[r]);
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[tid] = acc.x;
}

int main(int argc, char **argv)
{
    const int matrices = argc > 1 ? atoi(argv[1]) : 720;
    char *in; float *sink;
    CK(hipMalloc(&in, (size_t)matrices * (4u << 20)));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(in, 1, (size_t)matrices * (4u << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(P, A) do { \
        for (int it = 0; it < 3; it++) { \
            CK(hipEventRecord(e0, 0)); \
            hipLaunchKernelGGL((k_read<P, A>), dim3(256), dim3(512), 0, 0, in, sink, matrices); \
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); } \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("pattern %d aux %d: %.3f us per matrix, %.0f GB/s\n", P, A, ms * 1000 / matrices, matrices * 4.194304e-3 / (ms * 1e-3)); } while (0)
    RUN(0, 0); RUN(0, 2); RUN(1, 0); RUN(1, 2); RUN(2, 0); RUN(2, 2); RUN(3, 2);
    return 0;
}
