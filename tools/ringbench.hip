// ringbench.hip -- does an intermediate that is written and read back within a short ring stay
// on the die (Infinity Cache), or does it cost HBM time like a full-size workspace?
//
// One persistent launch walks `sectors` sectors.  Per sector (64 tile-units, spread over the grid):
//   * read one 1024-row x 128-byte column tile of the input (range-pass pattern, 128 KiB)
//   * mode >= 1: write 64 KiB (512 rows x 128 B, row stride 4 KiB) of ring slot  s      % R
//   * mode >= 2: read  64 KiB (16 rows x 4 KiB)                     of ring slot (s - L) % R
// No synchronisation: this measures transport only.  R = sectors reproduces the two-kernel
// path's traffic (everything through HBM); small R keeps the ring inside the 256 MiB cache.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/ringbench tools/ringbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ v4f ld(rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void st(rsrc_t r, int voff, v4f f)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, f), r, voff, 0, AUX);
}

// AUX bits: 0 plain, 2 nt, 16 sc1
template <int MODE, int AIN, int AST, int ALD>
__global__ __launch_bounds__(512) void k_ring(const float *in, float *ring, float *sink, int sectors, int R, int L)
{
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int n = 512;
    const int total = sectors * 64;
    v4f acc = {0, 0, 0, 0};
    const int p0 = w * 8 + (l >> 3), colb = (l & 7) * 16;
#pragma unroll 1
    for (int u = blockIdx.x; u < total; u += gridDim.x) {
        const int s = u >> 6, t = u & 63;                 // sector, tile-unit (2 channels x 32 tiles)
        const int ch = t >> 5, tile = t & 31;
        const float *src = in + ((size_t)s * 2 + ch) * 1024 * n * 2;
        const rsrc_t rs = make_rsrc(src, 1024u * n * 8u);
        const int voff = p0 * n * 8 + tile * 128 + colb;
        v4f v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = ld<AIN>(rs, voff, 64 * r * n * 8);
        if (MODE >= 1) {
            float *dst = ring + ((size_t)(s % R) * 2 + ch) * 512 * n * 2;
            const rsrc_t rd = make_rsrc(dst, 512u * n * 8u);
#pragma unroll
            for (int r = 0; r < 8; r++) st<AST>(rd, voff + 64 * r * n * 8, v[r] + v[r + 8]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) acc += v[r];
        }
        if (MODE >= 2) {
            const int sl = (s + R - (L % R)) % R;
            const float *row = ring + ((size_t)sl * 2 + ch) * 512 * n * 2 + (size_t)(tile * 16 + w * 2) * n * 2;
            const rsrc_t rr = make_rsrc(row, 2u * n * 8u);
#pragma unroll
            for (int r = 0; r < 8; r++) acc += ld<ALD>(rr, l * 16 + r * 1024, 0);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[tid] = acc.x;
}

template <int MODE, int AIN, int AST, int ALD>
static float run(const float *in, float *ring, float *sink, int sectors, int R, int L, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 4; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_ring<MODE, AIN, AST, ALD>), dim3(grid), dim3(512), 0, 0, in, ring, sink, sectors, R, L);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best * 1e3f / sectors;
}

int main(int argc, char **argv)
{
    const int sectors = argc > 1 ? atoi(argv[1]) : 360;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    const size_t in_bytes = (size_t)sectors * 8 << 20, ring_bytes = (size_t)sectors * 4 << 20;
    float *in, *ring, *sink;
    CK(hipMalloc(&in, in_bytes));
    CK(hipMalloc(&ring, ring_bytes));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(in, 1, in_bytes));
    CK(hipMemset(ring, 1, ring_bytes));
    printf("sectors %d grid %d (us/sector; 8 MiB in, 4 MiB ring write, 4 MiB ring read per sector)\n", sectors, grid);
    printf("input only           plain %.3f   nt %.3f\n",
           run<0, 0, 0, 0>(in, ring, sink, sectors, 1, 1, grid), run<0, 2, 0, 0>(in, ring, sink, sectors, 1, 1, grid));
    const int Rs[] = {1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 128, 360};
    for (int R : Rs) {
        if (R > sectors) continue;
        const int L = R >= 8 ? R / 2 : (R > 1 ? 1 : 0);
        printf("R %3d L %2d | w: in nt, st plain %.3f  st nt %.3f | w+r: in nt/st plain/ld plain %.3f  nt/plain/nt %.3f  nt/nt/nt %.3f  "
               "plain/plain/plain %.3f  nt/sc1/sc1 %.3f\n", R, L,
               run<1, 2, 0, 0>(in, ring, sink, sectors, R, L, grid), run<1, 2, 2, 0>(in, ring, sink, sectors, R, L, grid),
               run<2, 2, 0, 0>(in, ring, sink, sectors, R, L, grid), run<2, 2, 0, 2>(in, ring, sink, sectors, R, L, grid),
               run<2, 2, 2, 2>(in, ring, sink, sectors, R, L, grid), run<2, 0, 0, 0>(in, ring, sink, sectors, R, L, grid),
               run<2, 2, 16, 16>(in, ring, sink, sectors, R, L, grid));
        fflush(stdout);
    }
    return 0;
}
