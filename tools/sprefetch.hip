// sprefetch.hip -- can a CU pull lines from HBM into its XCD's L2 through the SCALAR path (s_load_dword: scalar cache -> L2,
// not the vector memory pipeline), and how fast?  Round 5: the fused launch runs 27 % faster when its input requests hit the
// L2 (profiles/r05/ab_l2_hit_input.log): the HBM latency of those requests, not their number, is what costs; a prefetcher
// would have to bring the lines in WITHOUT occupying the CU's vector-memory queue for an HBM latency per line.
//   mode 0: scalar touches only : every wave touches its share of the buffer, one s_load_dword per 128-byte line, BATCH loads
//           in flight per wave (lgkmcnt counts at most 15) -> lines/us per CU, GB/s of lines brought in
//   mode 1: vector reads only   : the same lines read with buffer_load_dwordx4 (16 B per lane, 8 lines per instruction)
//   mode 2: scalar touches of chunk k+1 by waves 8..15 of a workgroup while waves 0..7 read chunk k with vector loads: does
//           the vector read then run at L2-hit speed?
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/sprefetch tools/sprefetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int BATCH>
__device__ __forceinline__ void touch_lines(const char *p /* wave-uniform */, int lines)
{
    for (int i = 0; i < lines; i += BATCH) {
        const char *q = p + (size_t)i * 128;
#pragma unroll
        for (int b = 0; b < BATCH; b++) {
            unsigned t;
            asm volatile("s_load_dword %0, %1, %2" : "=s"(t) : "s"(q), "n"(b * 128) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int BATCH>
__global__ __launch_bounds__(1024) void k_scalar(const char *buf, size_t bytes_per_wave, int waves_per_wg)
{
    const int w = threadIdx.x >> 6;
    if (w >= waves_per_wg) return;
    const size_t wave = (size_t)blockIdx.x * waves_per_wg + w;
    // 64-bit uniform pointer: rebuilt from two readfirstlanes
    const unsigned long long a = (unsigned long long)(buf + wave * bytes_per_wave);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const char *p = (const char *)(((unsigned long long)hi << 32) | lo);
    touch_lines<BATCH>(p, (int)(bytes_per_wave / 128));
}

__global__ __launch_bounds__(1024) void k_vector(const char *buf, size_t bytes_per_wave, int waves_per_wg, float *sink)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (w >= waves_per_wg) return;
    const size_t wave = (size_t)blockIdx.x * waves_per_wg + w;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf + wave * bytes_per_wave), 0, (unsigned)bytes_per_wave, 0x00020000);
    v4u acc = {0, 0, 0, 0};
    for (unsigned off = l * 16; off < bytes_per_wave; off += 8 * 1024) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc += __builtin_amdgcn_raw_buffer_load_b128(rs, off + k * 1024, 0, 2);   // nt
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345u) sink[threadIdx.x] = 1.f;
}

// mode 2: workgroup of 16 waves; waves 0..7 read chunk c (vector, nt), waves 8..15 touch chunk c + AHEAD (scalar); chunks of
// `chunk` bytes per workgroup and step; a workgroup barrier per step keeps the two kinds in step
template <int BATCH>
__global__ __launch_bounds__(1024) void k_both(const char *buf, int chunk, int steps, int ahead, int do_touch, float *sink)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const char *mine = buf + (size_t)blockIdx.x * chunk * (steps + ahead + 1);
    v4u acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; s++) {
        if (w < 8) {
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(mine + (size_t)s * chunk), 0, (unsigned)chunk, 0x00020000);
            for (unsigned off = w * 1024 + l * 16; off < (unsigned)chunk; off += 8 * 1024) acc += __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);
        } else if (do_touch) {
            const unsigned long long a = (unsigned long long)(mine + (size_t)(s + ahead) * chunk + (size_t)(w - 8) * (chunk / 8));
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            touch_lines<BATCH>((const char *)(((unsigned long long)hi << 32) | lo), chunk / 8 / 128);
        }
        __syncthreads();
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345u) sink[threadIdx.x] = 1.f;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int waves = argc > 2 ? atoi(argv[2]) : 8;          // waves per workgroup that work (modes 0, 1)
    const int wgs = argc > 3 ? atoi(argv[3]) : 256;
    const size_t per_wave = argc > 4 ? (size_t)atol(argv[4]) : (1u << 20);
    const int ahead = argc > 5 ? atoi(argv[5]) : 1;
    const int do_touch = argc > 6 ? atoi(argv[6]) : 1;
    char *buf;
    float *sink;
    size_t total = mode == 2 ? (size_t)wgs * (128 * 1024) * (64 + ahead + 1) : (size_t)wgs * waves * per_wave;
    CK(hipMalloc(&buf, total + 4096));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(buf, 1, total));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(k_scalar<15>, dim3(wgs), dim3(1024), 0, 0, buf, per_wave, waves);
        else if (mode == 1) hipLaunchKernelGGL(k_vector, dim3(wgs), dim3(1024), 0, 0, buf, per_wave, waves, sink);
        else hipLaunchKernelGGL(k_both<15>, dim3(wgs), dim3(1024), 0, 0, buf, 128 * 1024, 64, ahead, do_touch, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = mode == 2 ? (double)wgs * 128 * 1024 * 64 : (double)total;
        printf("mode %d waves %d wgs %d ahead %d touch %d: %.1f us, %.1f GB/s, %.1f lines/us per workgroup\n", mode, waves, wgs, ahead, do_touch, ms * 1000,
               bytes / ms / 1e6, bytes / 128 / (ms * 1000) / wgs);
    }
    return 0;
}
