// pipeorder.hip -- does an L2-HIT load of one wave wait behind the HBM-latency loads that OTHER waves of its CU issued before it?
// (Round 5: the fused launch runs 27 % faster when its input requests hit the L2 -- profiles/r05/ab_l2_hit_input.log.  If a CU's
// vector-memory returns are in order ACROSS waves, what counts is how many INSTRUCTIONS with an HBM latency are in the queue,
// not how many lines: a prefetch that touches 64 lines per instruction -- one dword per lane -- would then cut the time the
// hand-over's L2 hits stand in the queue by the factor between 64 and the 8 lines a 16-byte-per-lane request covers.)
// One workgroup of 16 waves per CU (256 workgroups): waves 0..7 stream DISTINCT lines from HBM at about the fused launch's rate
// (~ 115 lines/us per CU), waves 8..15 time single L2-hit loads (sc1: past the L1) of a small resident region.
//   stream 0: none    1: 16 bytes per lane (8 lines per instruction), two instructions in flight per wave
//   stream 2: one dword per lane, every lane its own line (64 lines per instruction), one in flight per wave + a pause
// Prints the probes' latency (mean, median, p90) in us.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/pipeorder tools/pipeorder.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int PROBES = 256;
// probe: 0 an 8-byte load into registers (sc1)   1 a dword per lane loaded straight into LDS (buffer_load ... lds)   2 an 8-byte store
// split: 0 waves 0..7 stream, 8..15 probe (two of each on every SIMD)   1 the waves of SIMDs 0, 1 stream, those of SIMDs 2, 3 probe
//        2 even workgroups only stream, odd workgroups only probe (different CUs, the same L2s)
__global__ __launch_bounds__(1024) void k(const char *big, size_t per_cu, char *hot, int stream, int pause, unsigned *lat /* [grid][8][PROBES] */,
                                          unsigned long long *lines_done, float *sink, int iters, int probe, int split)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    __shared__ unsigned landing[8][64];
    __shared__ int stop, n_role[2], role_of[16];
    if (threadIdx.x == 0) { stop = 0; n_role[0] = n_role[1] = 0; }
    __syncthreads();
    // split 1: the role follows the SIMD the wave actually sits on (HW_ID bits 5:4): SIMDs 0, 1 stream, SIMDs 2, 3 probe
    if (l == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int role = split == 2 ? (wave >= 8) : split ? (((hw >> 4) & 3) >= 2) : (wave >= 8);
        role_of[wave] = role | (atomicAdd(&n_role[role], 1) << 1);
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) sink[100] = (float)n_role[0];
    if ((role_of[wave] >> 1) >= 8) return;   // (more than eight of a kind: the surplus waves idle)
    // split 2: the CUs take turns -- even workgroups only stream (their probers leave), odd workgroups only probe: the probes
    // then share the L2 and the fabric with the streams, but not a CU
    const int cu_kind = (blockIdx.x >> 3) & 1;    // workgroup i goes to XCD i % 8: neighbours in i >> 3 share an XCD
    if (split == 2 && (role_of[wave] & 1) != cu_kind) return;
    if (split == 2 && cu_kind == 0) iters = 800;    // a streaming CU runs ~ 0.8 ms (an iteration is one HBM round trip), no probers to wait for
    const int w = (role_of[wave] & 1) * 8 + (role_of[wave] >> 1);
    const int probers = n_role[1] < 8 ? n_role[1] : 8;
    if (w >= 8) {
        // prober: PROBES timed loads of 8 bytes per lane (4 lines) from this CU's 16 KiB of the hot region
        const rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(hot + (size_t)blockIdx.x * 16384, 0, 16384, 0x00020000);
        v2u acc = {0, 0};
        for (int i = 0; i < PROBES; i++) {
            const unsigned off = ((i * 8 + (w - 8)) * 512 + l * 8) & 16383;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            v2u v = {0, 0};
            if (probe == 0) v = __builtin_amdgcn_raw_buffer_load_b64(rh, off, 0, 16);   // sc1
            else if (probe == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (__attribute__((address_space(3))) void *)&landing[w - 8][0], 4, (off & 16383) / 2, 0, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b64(v2u{(unsigned)i, (unsigned)l}, rh, off, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            acc += v;
            if (l == 0) lat[((size_t)blockIdx.x * 8 + (w - 8)) * PROBES + i] = (unsigned)(t1 - t0);
            for (int p = 0, n = 3 + (i * 7 + w * 3) % 5; p < n; p++) __builtin_amdgcn_s_sleep(8);
        }
        if (acc.x + acc.y == 0x12345u) sink[l] = 1.f;
        if (l == 0) atomicAdd(&stop, 1);      // the streamers run until all eight probers are done
        return;
    }
    // streamer
    const char *mine = big + (size_t)blockIdx.x * per_cu + (size_t)w * (per_cu / 8);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(mine), 0, (unsigned)(per_cu / 8), 0x00020000);
    v4u acc = {0, 0, 0, 0};
    unsigned done = 0;
    unsigned off = 0;
    const unsigned span = (unsigned)(per_cu / 8);
    for (int it = 0; it < iters && (split == 2 || *(volatile int *)&stop < probers); it++) {
        if (stream == 1) {
            acc += __builtin_amdgcn_raw_buffer_load_b128(rs, (off + l * 16) % span, 0, 2);
            acc += __builtin_amdgcn_raw_buffer_load_b128(rs, (off + 1024 + l * 16) % span, 0, 2);
            off += 2048;
            done += 16;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (stream == 2) {
            acc.x += __builtin_amdgcn_raw_buffer_load_b32(rs, (off + l * 128) % span, 0, 2);
            off += 8192;
            done += 64;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int p = 0; p < pause; p++) __builtin_amdgcn_s_sleep(127);
        } else {
            __builtin_amdgcn_s_sleep(100);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345u) sink[l] = 1.f;
    if (l == 0) atomicAdd(lines_done, (unsigned long long)done);
}

int main(int argc, char **argv)
{
    const int stream = argc > 1 ? atoi(argv[1]) : 0;
    const int pause = argc > 2 ? atoi(argv[2]) : 40;
    const int probe = argc > 3 ? atoi(argv[3]) : 0;
    const int split = argc > 4 ? atoi(argv[4]) : 0;
    const size_t per_cu = 64u << 20;
    char *big, *hot;
    unsigned *lat;
    unsigned long long *lines;
    float *sink;
    CK(hipMalloc(&big, 256 * per_cu));
    CK(hipMalloc(&hot, 256 * 16384));
    CK(hipMalloc(&lat, 256 * 8 * PROBES * 4));
    CK(hipMalloc(&lines, 8));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(big, 1, 256 * per_cu));
    CK(hipMemset(hot, 1, 256 * 16384));
    CK(hipMemset(lines, 0, 8));
    CK(hipMemset(lat, 0, 256 * 8 * PROBES * 4));
    CK(hipMemset(sink, 0, 4096));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, big, per_cu, hot, stream, pause, lat, lines, sink, 1 << 20, probe, split);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(256 * 8 * PROBES);
    unsigned long long hl = 0;
    CK(hipMemcpy(h.data(), lat, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hl, lines, 8, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (size_t i = 0; i < h.size(); i++) if (i % PROBES >= 16 && h[i] != 0) v.push_back(h[i] / 100.0);   // (0: a workgroup that did not probe)   // s_memrealtime: 100 MHz
    std::sort(v.begin(), v.end());
    double mean = 0;
    for (double x : v) mean += x;
    mean /= v.size();
    float bad = 0;
    CK(hipMemcpy(&bad, sink + 100, 4, hipMemcpyDeviceToHost));
    if (split) printf("(workgroup 0: %g streaming waves on SIMDs 0, 1) ", bad);
    printf("probe %d split %d ", probe, split);
    printf("stream %d pause %d: kernel %.1f us, streamed %.1f lines/us per CU (%.0f GB/s); L2-hit probe latency mean %.2f us, median %.2f, p90 %.2f, p99 %.2f\n",
           stream, pause, ms * 1000, hl / (ms * 1000) / 256, hl * 128.0 / ms / 1e6, mean, v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100]);
    return 0;
}
