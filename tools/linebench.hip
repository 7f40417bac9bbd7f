// linebench.hip -- which 128-byte line of a row is slow to load, and where does the slowness live?
// The fused launches read column tiles: all rows of a matrix at ONE byte offset of the row (2048 rows x 64 bytes at a 1 KiB
// row stride for 2048 x 128; 1024 rows x 128 bytes at 4 KiB for 1024 x 512).  In the 2048 x 128 launch the members that request
// the tiles at byte offset 384..511 of every row are 2.7 us late in every task (profiles/r04/fused_b_slow_line.log).
// Here: 256 workgroups of 512 threads, workgroup (xcd = b & 7, member = b >> 3) reads the launch's tile pattern for tile
// `member` of a matrix `reps` times -- a new matrix every repetition (HBM), or the same one (served by the caches) -- and
// reports its own time; printed per tile index (mean over the 8 XCDs).
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/linebench tools/linebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int SHAPE, int MAP = 0>   // MAP (shape 0): which 16 rows one load instruction covers -- 0: consecutive (the launch), 1: 8 rows apart, 2: four groups of four, 32 apart
// SHAPE 0: 2048 rows x 1 KiB, 8-column tiles (64 B), channel = member >> 4; 1: 1024 rows x 4 KiB, 16-column tiles (128 B);
                       // 2: as 0, but a tile is 32 bytes of line 2 (T >> 2) and 32 bytes of the line behind it (four members share a line pair)
__global__ __launch_bounds__(512) void k(const char *in, float *res, float *sink, int reps, int same, int n_mtx, int shift)
{
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int xcd = blockIdx.x & 7, member = blockIdx.x >> 3;
    const size_t mtx_bytes = SHAPE != 1 ? (size_t)2 * 2048 * 1024 : (size_t)1024 * 4096;   // shape 0: a sector = two channels
    v4f acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; r++) {
        const char *base = in + shift + (size_t)((xcd + 8 * (same ? 0 : r)) % n_mtx) * mtx_bytes;
        v4u v[16];
        if (SHAPE == 0) {
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base + (size_t)(member >> 4) * 2048 * 1024), 0, 2048 * 1024, 0x00020000);
            const int j = l >> 2, cp = l & 3;
            const int p0 = MAP == 0 ? w * 16 + j : MAP == 1 ? j * 8 + w : w * 4 + (j & 3) + 32 * (j >> 2);
            const int voff = p0 * 1024 + (member & 15) * 64 + cp * 16;
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 128 * q * 1024, 2);
        } else if (SHAPE == 2) {
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base + (size_t)(member >> 4) * 2048 * 1024), 0, 2048 * 1024, 0x00020000);
            const int p0 = w * 16 + (l >> 2), cp = l & 3, T = member & 15;
            const int voff = p0 * 1024 + (2 * (T >> 2) + (cp >> 1)) * 128 + (T & 3) * 32 + (cp & 1) * 16;
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 128 * q * 1024, 2);
        } else {
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, 4 << 20, 0x00020000);
            const int p0 = w * 8 + (l >> 3), cp = l & 7;
            const int voff = p0 * 4096 + member * 128 + cp * 16;
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 64 * q * 4096, 2);
        }
#pragma unroll
        for (int q = 0; q < 16; q++) acc += __builtin_bit_cast(v4f, v[q]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) res[blockIdx.x] = (float)(t1 - t0) / 100.0f / reps;
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[tid] = acc.x;
}

int main()
{
    const int n_mtx = 128, reps = 400;
    char *in; float *res, *sink;
    CK(hipMalloc(&in, (size_t)n_mtx * (4u << 20) + 4096));
    CK(hipMemset(in, 1, (size_t)n_mtx * (4u << 20) + 4096));
    CK(hipMalloc(&res, 256 * 4)); CK(hipMalloc(&sink, 4096));
    std::vector<float> h(256);
    struct Case { int shape, map, shift; const char *what; } cases[] = {
        {0, 0, 0, "2048 rows x 64 B at 1 KiB (2048 x 128)"},
        {1, 0, 0, "1024 rows x 128 B at 4 KiB (1024 x 512)"},
        {2, 0, 0, "2048 rows x (32 + 32) B of two neighbouring lines at 1 KiB"},
        {0, 0, 128, "2048 x 128, the whole input 128 bytes further on"},
        {0, 1, 0, "2048 x 128, a load instruction's 16 rows 8 rows apart"},
        {0, 2, 0, "2048 x 128, a load instruction's 16 rows in four groups 32 rows apart"},
    };
    for (const Case &c : cases)
        for (int same = 0; same < 2; same++) {
            const int shape = c.shape;
            for (int rep = 0; rep < 2; rep++) {
                if (shape == 0 && c.map == 0) hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(512), 0, 0, in, res, sink, reps, same, n_mtx, c.shift);
                else if (shape == 0 && c.map == 1) hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(512), 0, 0, in, res, sink, reps, same, n_mtx, c.shift);
                else if (shape == 0) hipLaunchKernelGGL((k<0, 2>), dim3(256), dim3(512), 0, 0, in, res, sink, reps, same, n_mtx, c.shift);
                else if (shape == 2) hipLaunchKernelGGL((k<2, 0>), dim3(256), dim3(512), 0, 0, in, res, sink, reps, same, n_mtx, c.shift);
                else hipLaunchKernelGGL((k<1, 0>), dim3(256), dim3(512), 0, 0, in, res, sink, reps, same, n_mtx, c.shift);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), res, 256 * 4, hipMemcpyDeviceToHost));
            printf("%s, %s: us per tile by tile index (mean over the workgroups with that tile)\n", c.what,
                   same ? "the same matrix every time (cache-served)" : "a new matrix every time (HBM)");
            const int tiles = shape != 1 ? 16 : 32;
            for (int t = 0; t < tiles; t++) {
                double s = 0; int c = 0;
                for (int b = 0; b < 256; b++) {
                    const int member = b >> 3, tile = shape != 1 ? (member & 15) : member;
                    if (tile == t) { s += h[b]; c++; }
                }
                printf(" %5.2f", s / c);
                if (t % 16 == 15) printf("\n");
            }
        }
    return 0;
}
