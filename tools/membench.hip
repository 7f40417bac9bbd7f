// membench.hip -- memory-system ceilings for the access patterns of the two passes.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/membench tools/membench.hip ; run on the GPU box.
//   pattern 0: linear float4 copy                      (read B, write B)
//   pattern 1: range-pass pattern: each 512-thread block reads a 1024-row x 128-byte column
//              tile (row stride 4 KiB) and writes the first 512 rows of it (read B, write B/2)
//   pattern 2: same read, no write
//   pattern 3: doppler-pass pattern: one wave reads two 4 KiB rows with 8-byte lane accesses
//   pattern 4: pattern 1 with non-temporal loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_linear(const float4 *in, float4 *out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

template <int MODE>
__global__ __launch_bounds__(512) void k_tile(const float2 *in, float2 *out, int n)
{
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int tiles = n / 16;
    int b = blockIdx.x;
    const int tile = b % tiles; b /= tiles;
    const float2 *src = in + (size_t)b * 1024 * n;
    float2 *dst = out + (size_t)b * 512 * n;
    const int p0 = w * 8 + (l >> 3), col0 = tile * 16 + (l & 7) * 2;
    float4 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float4 *p = reinterpret_cast<const float4 *>(&src[(size_t)(p0 + 64 * r) * n + col0]);
        if (MODE == 4) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
            v[r] = make_float4(t.x, t.y, t.z, t.w);
        } else {
            v[r] = *p;
        }
    }
    if (MODE == 2) {
        float s = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) s += v[r].x + v[r].y + v[r].z + v[r].w;
        if (s == 123.456f) dst[0] = make_float2(s, s);
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++)
            *reinterpret_cast<float4 *>(&dst[(size_t)(p0 + 64 * r) * n + col0]) =
                make_float4(v[r].x + v[r + 8].x, v[r].y + v[r + 8].y, v[r].z + v[r + 8].z, v[r].w + v[r + 8].w);
    }
}

__global__ __launch_bounds__(256) void k_rows(const float2 *in, float *out, int rows_per_sector)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t gate = (size_t)blockIdx.x * 4 + w;
    float s = 0;
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        const float2 *row = in + ((size_t)blockIdx.y * 2 + ch) * rows_per_sector * 512 + gate * 512;
#pragma unroll
        for (int r = 0; r < 8; r++) { float2 x = row[l + 64 * r]; s += x.x + x.y; }
    }
    if (s == 123.456f) out[gate] = s;
}

int main(int argc, char **argv)
{
    const int sectors = argc > 1 ? atoi(argv[1]) : 120;
    const int n = 512;
    const size_t in_elems = (size_t)sectors * 2 * 1024 * n, out_elems = in_elems / 2;
    float2 *in, *out;
    CK(hipMalloc(&in, in_elems * 8));
    CK(hipMalloc(&out, in_elems * 8));
    CK(hipMemset(in, 1, in_elems * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pat = 0; pat < 5; pat++) {
        float best = 1e9;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(e0));
            switch (pat) {
            case 0: hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, (const float4 *)in, (float4 *)out, in_elems / 2); break;
            case 1: hipLaunchKernelGGL(k_tile<1>, dim3(sectors * 2 * (n / 16)), dim3(512), 0, 0, in, out, n); break;
            case 2: hipLaunchKernelGGL(k_tile<2>, dim3(sectors * 2 * (n / 16)), dim3(512), 0, 0, in, out, n); break;
            case 3: hipLaunchKernelGGL(k_rows, dim3(128, sectors), dim3(256), 0, 0, in, (float *)out, 512); break;
            case 4: hipLaunchKernelGGL(k_tile<4>, dim3(sectors * 2 * (n / 16)), dim3(512), 0, 0, in, out, n); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        double rd = in_elems * 8.0, wr = 0;
        if (pat == 0) wr = rd;
        if (pat == 1 || pat == 4) wr = out_elems * 8.0;
        if (pat == 3) rd = (double)sectors * 2 * 512 * 512 * 8;
        printf("pattern %d: %.3f ms  read %.1f MB write %.1f MB -> %.2f TB/s  (%.3f us/sector)\n", pat, best,
               rd / 1e6, wr / 1e6, (rd + wr) / best / 1e9, best * 1e3 / sectors);
    }
    return 0;
}
