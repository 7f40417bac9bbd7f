// l2wb.hip -- does an XCD's L2 keep dirty lines that are overwritten again and again, or does every
// generation of stores go out to the fabric?  (Decides the traffic floor of the fused launch, whose
// 2 MiB per-XCD buffer is rewritten once per channel-task.)
//
// 256 workgroups x 512 threads; workgroup b owns a private chunk of `chunk` bytes and rewrites it
// `reps` times with plain 16-byte stores (optionally reading it back with sc1 loads in between, and
// optionally streaming `stream` bytes of fresh input per repetition with non-temporal loads).
// Run under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE and compare reps = 1 with reps = 50:
//   write-back cache that keeps its dirty lines: WRITE_SIZE ~ chunk * 256, independent of reps
//   every generation written out               : WRITE_SIZE ~ chunk * 256 * reps
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/l2wb tools/l2wb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ unsigned g_rank[8];
template <int MODE>   // 0: stores only; 1: + sc1 read-back; 2: + nt input stream; 3: both; +4: the chunks of one XCD are contiguous
__global__ __launch_bounds__(512) void k_rewrite(float *buf, const float *in, float *sink, int chunk, int reps, int stream, int aux_sel, int st_sel)
{
    const int tid = threadIdx.x;
    __shared__ unsigned s_slot;
    if (tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7;
        s_slot = (MODE & 4) ? x * 32 + (atomicAdd(&g_rank[x], 1u) & 31) : blockIdx.x;
    }
    __syncthreads();
    char *mine = reinterpret_cast<char *>(buf) + (size_t)s_slot * chunk;
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(mine, 0, chunk, 0x00020000);
    v4f acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; r++) {
        if (st_sel >= 4) {
            // the 2048 x 128 launch's slot stores (round 5): a 128-byte line of the rewritten buffer gets its two 64-byte HALVES
            // from two different workgroups, s_slot and s_slot ^ 1 (4: same XCD when MODE & 4), at whatever times they get there;
            // 5: from the same workgroup, all first halves and then all second halves
            const rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(buf) + (size_t)(s_slot & ~1u) * chunk, 0, 2 * chunk, 0x00020000);
            for (int pass = 0; pass < (st_sel == 5 ? 2 : 1); pass++) {
                const int half = st_sel == 5 ? pass : (int)(s_slot & 1);
                const int first = st_sel == 5 ? (int)(s_slot & 1) * (chunk / 128) : 0, lines = st_sel == 5 ? chunk / 128 : 2 * chunk / 128;
                for (int line = tid >> 2; line < lines; line += 128) {
                    v4f v = {(float)r, (float)line, 1.f, 2.f};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rp, (first + line) * 128 + half * 64 + (tid & 3) * 16, 0, 0);
                }
            }
        } else
        for (int off = tid * 16; off < chunk; off += 512 * 16) {
            v4f v = {(float)r, (float)off, 1.f, 2.f};
            // cache policy of the rewritten buffer's stores: 0 plain, 1 nt, 2 sc0 (3: b64 pairs; 4, 5: half lines, above)
            if (st_sel == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rb, off, 0, 0);
            else if (st_sel == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rb, off, 0, 2);
            else if (st_sel == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rb, off, 0, 1);
            else {   // 3: the same bytes as two 8-byte stores per lane, 16 lanes per 128-byte line (the fused launch's tile stores)
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const int l = tid & 63, base = off - l * 16;          // this wave-instruction's 1 KiB
                v2u d = {(unsigned)r, (unsigned)off};
                __builtin_amdgcn_raw_buffer_store_b64(d, rb, base + l * 8, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(d, rb, base + 512 + l * 8, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE & 1)
            for (int off = tid * 16; off < chunk; off += 512 * 16)
                acc += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 16));
        if ((MODE & 2) && aux_sel == 6) {
            // the 2048 x 128 launch's input pattern (round 5): the two workgroups s_slot, s_slot ^ 1 (same XCD when MODE & 4) read
            // the two 64-byte HALVES of the same 128-byte lines at about the same time, non-temporally: 4 lanes x 16 bytes per row
            // of 128 bytes, 16 rows per wave-instruction; every line is asked for twice
            const char *src = reinterpret_cast<const char *>(in) + ((size_t)r * gridDim.x + (s_slot & ~1u)) * stream;
            const rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src), 0, 2 * stream, 0x00020000);
            const int half = s_slot & 1;
            for (int row = tid >> 2; row < 2 * stream / 128; row += 128) {
                const v4u t = __builtin_amdgcn_raw_buffer_load_b128(ri, row * 128 + half * 64 + (tid & 3) * 16, 0, 2);
                acc += __builtin_bit_cast(v4f, t);
            }
        } else if (MODE & 2) {
            const char *src = reinterpret_cast<const char *>(in) + ((size_t)r * gridDim.x + blockIdx.x) * stream;
            const rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src), 0, stream, 0x00020000);
            // cache policy of the stream: 0 plain, 1 nt, 2 sc1 nt, 3 sc0 sc1 nt, 4 sc1, 5 sc0 sc1 (6: nt HALF lines by pairs of workgroups, above)
            for (int off = tid * 16; off < stream; off += 512 * 16) {
                v4u t;
                switch (aux_sel) {
                case 0: t = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 0); break;
                case 1: t = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 2); break;
                case 2: t = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 18); break;
                case 3: t = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 19); break;
                case 4: t = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 16); break;
                default: t = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 17); break;
                }
                acc += __builtin_bit_cast(v4f, t);
            }
        }
        __syncthreads();
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[tid] = acc.x;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    const int chunk = argc > 3 ? atoi(argv[3]) : 65536;      // 64 KiB x 32 CUs = 2 MiB per XCD
    const int stream = argc > 4 ? atoi(argv[4]) : 131072;    // 128 KiB per workgroup and repetition
    const int aux_sel = argc > 5 ? atoi(argv[5]) : 1;
    const int st_sel = argc > 6 ? atoi(argv[6]) : 0;
    const int in_mem = argc > 7 ? atoi(argv[7]) : 0;         // memory of the streamed input: 0 hipMalloc, 1 uncached, 2 fine-grained
    float *buf, *in, *sink;
    CK(hipMalloc(&buf, (size_t)256 * chunk));
    if (in_mem == 0) CK(hipMalloc(&in, (size_t)256 * stream * reps + 4096));
    else CK(hipExtMallocWithFlags((void **)&in, (size_t)256 * stream * reps + 4096, in_mem == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(in, 1, (size_t)256 * stream * reps));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_rewrite<0>, dim3(256), dim3(512), 0, 0, buf, in, sink, chunk, reps, stream, aux_sel, st_sel); break;
    case 1: hipLaunchKernelGGL(k_rewrite<1>, dim3(256), dim3(512), 0, 0, buf, in, sink, chunk, reps, stream, aux_sel, st_sel); break;
    case 2: hipLaunchKernelGGL(k_rewrite<2>, dim3(256), dim3(512), 0, 0, buf, in, sink, chunk, reps, stream, aux_sel, st_sel); break;
    case 3: hipLaunchKernelGGL(k_rewrite<3>, dim3(256), dim3(512), 0, 0, buf, in, sink, chunk, reps, stream, aux_sel, st_sel); break;
    case 4: hipLaunchKernelGGL(k_rewrite<4>, dim3(256), dim3(512), 0, 0, buf, in, sink, chunk, reps, stream, aux_sel, st_sel); break;
    case 5: hipLaunchKernelGGL(k_rewrite<5>, dim3(256), dim3(512), 0, 0, buf, in, sink, chunk, reps, stream, aux_sel, st_sel); break;
    default: hipLaunchKernelGGL(k_rewrite<7>, dim3(256), dim3(512), 0, 0, buf, in, sink, chunk, reps, stream, aux_sel, st_sel); break;
    }
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("in_mem %d: %.1f us per repetition\n", in_mem, ms * 1000 / reps);
    printf("mode %d reps %d chunk %d stream %d: stored per generation %.1f MiB, streamed per repetition %.1f MiB\n", mode, reps,
           chunk, stream, 256.0 * chunk / 1048576, (mode & 2) ? 256.0 * stream / 1048576 : 0.0);
    return 0;
}
