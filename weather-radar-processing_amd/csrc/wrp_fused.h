// wrp_fused.h -- the whole per-sector chain (a2 .. a9) in ONE persistent launch.  The half-height
// intermediate of a sector-channel (2 MiB) does not go through HBM: it is handed from the range FFT to
// the Doppler rows through the L2 of the XCD whose 32 CUs work on that sector-channel.
//
// Reference being replaced: rpv2.cu:409-570 (about 14 passes over a 2-4 MiB array per channel).
//
// Why this shape (all measured on MI355X):
//   * the two-kernel path moves 16.8 MB per sector (8 in, 4 out, 4 back in) and runs at the
//     transport floor of that traffic (3.4 us; tools/ringbench.hip R = 360);
//   * keeping the intermediate in a short ring that stays inside the 256 MiB Infinity Cache still
//     costs 2.6 us per sector: what counts is bytes between an XCD and the fabric, not HBM hits;
//   * input alone streams at 1.4 us per sector, so only an exchange INSIDE an XCD can get there.
//
// Launch: 2 x CUs workgroups of 512 threads; every CU hosts two (78 KiB of LDS each, <= 128 VGPRs).
// At start a workgroup reads the XCD and the CU it runs on (HW_REG_XCC_ID, HW_REG_HW_ID: placement
// is READ, never assumed): of the two workgroups of a CU the OLDER one (lower blockIdx) becomes a
// TILE workgroup, the other a ROW workgroup, so that every CU always has the HBM-bound / barrier-bound
// range stages of one beside the VALU-bound row transforms of the other, and the SIMDs' oldest-first
// issue favours the tile waves (the critical path).  The workgroups of one XCD form a team of
// 32 tile members + 32 row members; a team meets once (its census), teams never talk to each
// other.  Team e owns sectors e, e + teams, ...; a sector is two channel-TASKS q = 0, 1, 2, ...
//   tile member r : range tile (r + q) mod 32 (16 columns) of task q  -> the team's ONE 1 MiB slot
//   row member r  : gates 16 r .. 16 r + 15 of every task (wave w: gate 16 r + w of half 0 and 16 r + 8 + w of half 1), a4 .. a9
// Hand-offs.  A task goes through the slot in two HALVES g = 0, 1 -- the gates with (gate mod 16) in
// [8 g, 8 g + 8), which is exactly what k1-group g of a tile produces (see below); gate -> slot row
// (gate >> 4) * 8 + (gate & 7).  Each half has its own pair of flag arrays:
//   stored[g] : a tile member publishes "half g of q tasks stored" once its stores have drained; a row
//               member waits until all 32 tile members have published the task it wants
//   loaded[g] : a row member publishes once its 8 rows of the half are in registers (one per wave); a tile member
//               waits for all 32 before it puts the next half into the slot
// A tile member stores half 0 in the middle of a tile and half 1 at its end; while the rows of one
// half are loaded and transformed the tile members compute the other half.  (Two slots, one per half,
// give each hand-over a whole task of slack -- and were no faster: 2 MiB rewritten per XCD do not stay
// in the 4 MiB L2 beside the streaming input, see the look in the tile loop.)  Store drains
// are waited for where they cost nothing: half 0 behind the next tile's requests (counted s_waitcnt:
// the loads are younger than the stores), half 1 behind the next tile's stage 1.
// Every POLLER has a 128-byte line of its own per array; byte w of it belongs to writer w.  A writer
// publishes with ONE wave instruction (lane i stores its sequence number, mod 256, into byte `rank` of
// line i: plain stores, the lines stay in the L2), a poller reads its line with one scalar load behind an
// s_dcache_inv (served by the L2; see l2_flag32 / l2_peek_flags for what the earlier forms cost).  All of
// this stays inside one XCD, whose L2 is the point of coherence for its own CUs: tiles are stored with plain
// stores (the lines stay in that L2) and drained with s_waitcnt vmcnt before the flag goes out; rows are
// read with loads that miss the reader's L1 (sc1).  The input is read non-temporally so that it does not
// push the slot out of the L2.  Every spin is bounded; a timeout or a team that is not 32 + 32 is reported
// in FusedCtl::status and the engine falls back to the two-kernel path.
//
// Range FFT: exactly the arithmetic of range_pass_1024<16> (1024 = 16 x 8 x 8, same butterflies,
// same twiddles, same order -> bit-identical results; the library is built with -ffp-contract=off
// and every fused multiply-add is spelled out).  What differs is the LDS image: the sixteen 64-point
// sub-transforms that follow the radix-16 stage are independent, and sub-transform k1 lives
// entirely in wave k1 mod 8 for stages 2 and 3.  So the tile goes through LDS in two GROUPS of
// eight k1 (72 KiB instead of 144 KiB -- which is what lets two workgroups share a CU with
// 16-column tiles, i.e. whole 128-byte lines); group 1 waits in 32 registers meanwhile, and only
// the hand-over from stage 1 needs workgroup barriers (4 per tile; the counts use LDS arrival counters).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wrp_kernels.h"

namespace wrp {

constexpr int FUSED_THREADS = 512;
constexpr int FUSED_MEMBERS = 32;                  // tile members = row members per team = CUs per XCD
constexpr int FUSED_MAX_TEAMS = 8;                 // XCDs of the device this is written for
constexpr int FUSED_STAMP_TASKS = 16;
// The measured choices below (alternatives, numbers and logs: DESIGN.md 4.1; the rejected forms are in the history of this
// file, `git log -- weather-radar-processing_amd/csrc/wrp_fused.h`, and in profiles/r03/ab_*.log):
//   * every row wave polls the L2 itself with s_sleep 1 between polls, and serves both halves, one gate of each;
//   * the next tile is requested ONE load at a time, sixteen pieces over twelve points of the task, non-temporally;
//   * a row wave runs from its notice of a half to its `loaded` flag at s_setprio 3.
constexpr unsigned long long FUSED_JOIN_TICKS = 400000ull;   // 4 ms of s_memrealtime (100 MHz): deadline of the team meeting
constexpr int FUSED_INPUT_AUX = AUX_NT;   // cache policy of the planar input loads: the stream must not push the slot out of the L2
constexpr int FUSED_ROW_PRIO = 3;         // s_setprio level of a row wave between its notice of a half and its `loaded` flag
constexpr int FUSED_POLL_SLEEP = 1;       // s_sleep units (64 cycles) between two polls of a row wave
constexpr int FUSED_STAMPS = 9;   // 0..7 phase stamps per task, 8: identity (task 0)
constexpr int FUSED_SLOT_ROWS = RP_M / 4;                          // 256: the gates of ONE half
constexpr size_t FUSED_TEAM_ELEMS = (size_t)FUSED_SLOT_ROWS * DP_N;   // float2 units: ONE slot[256][512] = 1 MiB per team

// One 128-byte line per POLLER; byte w of it is writer w's sequence number (tasks done, mod 256), written with a plain
// byte store.  A poller wants all 32 bytes EQUAL to its target, and equality is the right test because a writer is
// never AHEAD of a poller's target: a tile member stores half g of task q+1 only after every row member has published
// "half g of task q loaded" (one slot: the looks), and a row member loads half g of task q+1 only after every tile
// member has published it.  So at the moment of a look a writer's byte is the target or the target minus one --
// also across the wrap of the byte.  32 bytes are one s_load_dwordx8.
struct FusedFlags { unsigned char b[FUSED_MEMBERS]; unsigned char pad[128 - FUSED_MEMBERS]; };
struct FusedCtl {               // zeroed by the host once; every launch leaves it zeroed again (the teams' last workgroups)
    unsigned unused0;
    unsigned status;            // 0 ok; 1: a bounded spin gave up; 2: a team is not 32 + 32 workgroups.  Sticky: the host zeroes the block after a failure
    unsigned pad0[30];
    unsigned census[2][8];      // workgroups per kind (0 tile, 1 row) and XCC
    unsigned done[8];           // workgroups of the XCC's team that have left the task loop
    unsigned pad1[8];
    unsigned cu_arrivals[8][256];            // workgroups seen per physical CU (key = HW_ID bits 15:8: se, sh, cu)
    unsigned cu_block[8][256][2];            // blockIdx + 1 of the first and the second workgroup to arrive there
    FusedFlags stored[2][8][FUSED_MEMBERS];  // [half][xcc][line of row member r]: byte t = tasks whose half tile member t has stored
    FusedFlags loaded[2][8][FUSED_MEMBERS];  // [half][xcc][line of tile member t]: byte r = tasks whose half row member r has in registers
};
static_assert(sizeof(FusedCtl) % 16 == 0, "memset block is a multiple of 16 bytes");

// LDS of a tile workgroup: image of ONE k1-group [8 k1][64 positions][16 columns] complex, one
// position (128 B) of padding after every 8 -> all three stages are bank-conflict free (same map
// as RangeTile<16>); the 1024-entry twiddle table lives in the 64 pads; window behind the image.
struct FusedTile {
    static constexpr int ROW_BYTES = 128, BLK_BYTES = 9 * ROW_BYTES, BLOCKS = 64;
    static constexpr int IMG_BYTES = BLOCKS * BLK_BYTES;         // 73728
    static constexpr int OFF_WR = IMG_BYTES;                     // float wr_c[1024]
    static constexpr int OFF_CTL = OFF_WR + RP_M * 4;            // 77824: control words, both kinds
    static constexpr int OFF_TW2 = OFF_CTL + 64;                 // float2 [8][8]: W_64^{p1 k2}, the stage-2 twiddles
    static constexpr int OFF_STAMPS = OFF_TW2 + 512;             // diagnostics build: [16 tasks][9] 64-bit stamps
    static constexpr int LDS_BYTES = OFF_STAMPS + FUSED_STAMP_TASKS * FUSED_STAMPS * 8;   // 79552 -> exactly two workgroups per CU
    // row workgroup: 8 wave buffers, then the Doppler twiddles
    static constexpr int OFF_TWN = 8 * DP_ELEMS * 8;             // 36864
    static_assert(OFF_TWN + DP_TW_ELEMS * 8 <= OFF_CTL, "row workgroup layout fits");
    static_assert(2 * LDS_BYTES <= 160 * 1024 && 3 * LDS_BYTES > 160 * 1024, "exactly two workgroups per CU");
    static __device__ __forceinline__ int addr(int pos, int cp) { return (pos >> 3) * BLK_BYTES + (pos & 7) * ROW_BYTES + cp * 16; }
    // The sixteen columns of a position are one 128-byte row; in the rows of ODD positions the two columns of every pair
    // are swapped (column c sits at (c ^ (pos & 1)) * 8).  Stage 1 writes ONE column of a pair at a time, 8 bytes per lane at
    // a 16-byte stride, and a ds_write_b64 serves sixteen lanes at once -- two rows: unswapped, both rows' 8-byte pieces
    // fall on the same sixteen banks (two passes per write, 55 % of the launch's bank-conflict cycles); swapped, the odd
    // row takes the other sixteen (SQ_LDS_BANK_CONFLICT 21.5 M -> 9.7 M per launch, 9.3 % -> 4.4 % of the LDS's active
    // cycles; the launch is no faster for it: stage 1 is bound by its arithmetic).  Every other access covers whole rows and
    // only sees its columns permuted.  The wire-format launch writes whole rows in stage 1 and keeps the plain order.
    template <bool SWZ = true> static __device__ __forceinline__ int swz(int pos) { return SWZ ? (pos & 1) : 0; }
    // Stage-1 twiddles W_1024^{p0 k1} ARRANGED in the 64 pads so that a lane reads its fifteen at ONE base + immediate
    // offsets, and the eight positions p0 = 8 w .. 8 w + 7 of a wave are 64 contiguous bytes (no bank conflict): pad
    // 8 (p0 >> 3) + (k1 >> 1), byte 64 (k1 & 1) + 8 (p0 & 7).
    static __device__ __forceinline__ int tw1_addr(int p0, int k1)
    {
        return ((p0 >> 3) * 8 + (k1 >> 1)) * BLK_BYTES + 8 * ROW_BYTES + (k1 & 1) * 64 + (p0 & 7) * 8;
    }
    static __device__ __forceinline__ int tw2_addr(int p1, int k2) { return OFF_TW2 + (p1 * 8 + k2) * 8; }
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}
__device__ __forceinline__ unsigned hw_cu_key()   // se, sh, cu of the CU this wave runs on
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return (v >> 8) & 0xff;
}

// Notifications stay inside ONE XCD and inside its L2 (tools/hopbench.hip, profiles/r02/hopbench.log):
//   * a writer publishes its sequence number with ONE wave instruction, lane i storing the byte into line i (the line of
//     poller i): PLAIN stores, the lines stay in the L2.  A line that takes an atomic is written back and dropped -- the
//     first form counted with L2 atomics on replicated counters, and every poll behind an update went out to memory;
//   * a poller looks with s_dcache_inv + an ORDINARY scalar load: served by the L2.  A scalar load with glc is served by
//     MEMORY, every time (hopbench: 650 bytes of FETCH_SIZE per notification and poller; the launch fetched 1.0 MB per
//     sector with its input loads switched off) -- 0.73 us per notification on an idle chip against 0.43 us this way,
//     and under load the difference is the fabric's queueing.  Scalar, because scalar memory has its own path and
//     counter (lgkmcnt): a vector poll returns in order behind the polling wave's own tile requests, and the row
//     workgroup's vector polls queue in the CU's memory pipeline with the tile workgroup's requests (measured slower).
// The lane offset is recomputed at every call: kept across the task loop it is spilled, and the reload of a spilled
// address waits (vmcnt, in order) for every request in flight -- the whole next tile.
__device__ __forceinline__ void l2_flag32(FusedFlags *lines /* wave-uniform */, int l, int rank, unsigned seq)
{
    asm volatile("" : "+v"(l));
    if (l < FUSED_MEMBERS) __hip_atomic_store(&lines[l].b[rank], (unsigned char)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// 0 when all 32 bytes of the line equal the low byte of `seq`
typedef unsigned v8u __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned l2_peek_flags(const FusedFlags *line /* wave-uniform */, unsigned seq)
{
    v8u a;
    asm volatile("s_dcache_inv\n\ts_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(line) : "memory");
    const unsigned pat = (seq & 0xffu) * 0x01010101u;
    return ((a[0] ^ pat) | (a[1] ^ pat)) | ((a[2] ^ pat) | (a[3] ^ pat)) | ((a[4] ^ pat) | (a[5] ^ pat)) | ((a[6] ^ pat) | (a[7] ^ pat));
}

// control words in LDS: address space 3 spelled out, because hipcc does not infer it for volatile
// accesses and would emit flat instructions with sc0 sc1 for them
typedef __attribute__((address_space(3))) volatile int lds_word;

// a row wave: wait until every tile member has published `seq`; false = gave up (status set)
// A poll is one L2 round trip (>= 0.3 us under load) plus the sleep: FUSED_SPIN_BUDGET polls are >= 20 ms, three orders
// of magnitude above the longest legitimate wait inside the task loop (one task, ~10 us), and the launch reports
// "workgroups not co-resident" in milliseconds instead of the second the first budget (2^22 polls) took.
constexpr unsigned FUSED_SPIN_BUDGET = 1u << 16;
__device__ __forceinline__ bool spin_flags(const FusedFlags *line, unsigned seq, unsigned *status)
{
#pragma unroll 1
    for (unsigned spins = 0; spins < FUSED_SPIN_BUDGET; spins++) {
        if (l2_peek_flags(line, seq) == 0) return true;
        if ((spins & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        __builtin_amdgcn_s_sleep(FUSED_POLL_SLEEP);
    }
    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// the same wait for a wave in the middle of a tile: no early return (an exit edge there costs the
// tile loop 14 spilled registers); a wave that gave up remembers it (`failed`, wave-uniform), stops
// waiting and runs on -- the launch is reported as failed through `status` and its output discarded
// skip (wave-uniform): a wave that has nothing to wait for passes straight through, without a branch
__device__ __forceinline__ void spin_flags_sticky(const FusedFlags *line, unsigned seq, int &failed, bool skip = false)
{
    // the whole bounded loop is ONE asm statement: hipcc sees no control flow, so the registers
    // that are live across it (a tile's worth) are not split around a loop and spilled.  The eight
    // dwords of the line land in s[92:99], above what the kernel otherwise uses.
    unsigned budget = __builtin_amdgcn_readfirstlane(failed ? 1u : FUSED_SPIN_BUDGET);
    const unsigned have = __builtin_amdgcn_readfirstlane(skip ? 0xffffffffu : 0u);   // 0: look, else: pass
    const unsigned p32 = __builtin_amdgcn_readfirstlane((seq & 0xffu) * 0x01010101u);
    const unsigned long long pat = ((unsigned long long)p32 << 32) | p32;
    asm volatile("s_cmp_lg_u32 %[have], 0\n\t"
                 "s_cbranch_scc1 wrp_done%=\n"
                 "wrp_spin%=:\n\t"
                 "s_dcache_inv\n\t"
                 "s_load_dwordx8 s[92:99], %[p], 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_xor_b64 s[92:93], s[92:93], %[pat]\n\t"
                 "s_xor_b64 s[94:95], s[94:95], %[pat]\n\t"
                 "s_xor_b64 s[96:97], s[96:97], %[pat]\n\t"
                 "s_xor_b64 s[98:99], s[98:99], %[pat]\n\t"
                 "s_or_b64 s[92:93], s[92:93], s[94:95]\n\t"
                 "s_or_b64 s[96:97], s[96:97], s[98:99]\n\t"
                 "s_or_b64 s[92:93], s[92:93], s[96:97]\n\t"
                 "s_cmp_eq_u64 s[92:93], 0\n\t"
                 "s_cbranch_scc1 wrp_done%=\n\t"
                 "s_sub_u32 %[budget], %[budget], 1\n\t"
                 "s_cmp_eq_u32 %[budget], 0\n\t"
                 "s_cbranch_scc1 wrp_done%=\n\t"
                 "s_sleep 1\n\t"
                 "s_branch wrp_spin%=\n"
                 "wrp_done%=:"
                 : [budget] "+s"(budget) : [p] "s"(line), [pat] "s"(pat), [have] "s"(have)
                 : "memory", "scc", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
    failed |= budget == 0;   // straight-line code here; the caller reports `failed` once, after its loop
}

// ---- the slow lines of the input, touched ahead by the row waves ---------------------------------------------------
// Of the eight 128-byte lines of every KiB of device memory the one at byte 384 is slow to read -- 9.3 us against 7.05 for
// a column tile of the 2048 x 128 launch that falls on it, 4.0 against 3.25 when the caches serve it; it moves with the
// ABSOLUTE address, not with the buffer (tools/linebench.hip, profiles/r04/linebench.log) -- and a team moves at the pace
// of the member whose tile lies on it (profiles/r04/fused_b_slow_line.log).  The row waves stand at their polls for half
// of every task: each of them touches its share of those lines of the input the tile members will ask for NEXT task (one
// dword per line, default cache policy, so that the line waits in the XCD's L2; an eighth of the input, 512 KiB per task
// and team), and the members on the slow tile then find L2 hits: 2048 x 128 launch 1.52 -> 1.30 us/sector (-14 %,
// profiles/r05/ab_b_slow_line_touch.log; wrp_fused_b.h).  In the 1024 x 512 launch a slow tile is a sixteenth of a member's
// requests of a task, not all of them, and the touches gain nothing (-0.5 +- 0.3 %; wire format +3.7 %:
// profiles/r05/ab_a_slow_line_touch.log): not used there.  The value is returned so that the caller can keep its register
// reserved until a later s_waitcnt vmcnt(0) of its own has passed (nothing reads it).
constexpr unsigned FUSED_SLOW_LINE = 384u;      // byte offset of the slow line in every KiB of the address space
__device__ __forceinline__ float fused_touch_slow_lines(const void *range /* wave-uniform */, unsigned range_bytes, int wave_in_team,
                                                        int l, bool valid)
{
    const unsigned first = (FUSED_SLOW_LINE - (unsigned)(unsigned long long)range) & 1023u;   // first byte of the range on a slow line
    asm volatile("" : "+v"(l));
    // lane k < 16 of wave w: the line 256 k + w of the range (a 4 MiB range has 4096 of them); everything else falls
    // outside the descriptor and is dropped by the hardware
    const unsigned line = 256u * (unsigned)(l & 15) + (unsigned)wave_in_team;
    const unsigned voff = (l & 63) < 16 ? first + 1024u * line : 0x7ffffff0u;
    const rsrc_t rs = make_rsrc(range, valid ? range_bytes : 0u);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
}

// ---- tile member: device functions -----------------------------------------------------------
// A QUARTER of this lane's 16 row loads: rows p0 + 64 r with r = QUARTER mod 4 (r = QUARTER, + 4, + 8,
// + 12: exactly the inputs of ONE first-level butterfly of the radix-16 stage, so stage 1 can start
// on the quarters that have arrived); quarter 3 also fetches the lane's two Doppler-window values.
// valid = false -> zero-record descriptor, the hardware drops the loads (see range_load).
// The tile is requested in quarters because a CU takes a bounded number of outstanding requests: a
// whole tile (128 KiB) asked for at once stalls the requesting waves in the ISSUE for most of the
// time the tile needs to arrive, and every L2 hit of the CU's row workgroup queues behind it.
template <int QUARTER>
__device__ __forceinline__ void fused_tile_load(const float2 *src /* wave-uniform */, int col_base, const float *wd,
                                                float4 (&v)[16], float2 &wdv, bool valid)
{
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));   // recompute the lane offsets per call instead of keeping (spilling) them
    const int p0 = w * 8 + (l >> 3), cp = l & 7;
    const rsrc_t rs = make_rsrc(src, valid ? (unsigned)RP_M * DP_N * 8u : 0u);
    const int voff = (p0 * DP_N + col_base + cp * 2) * 8;
#pragma unroll
    for (int r = QUARTER; r < 16; r += 4) v[r] = buf_load_f4<FUSED_INPUT_AUX>(rs, voff, 64 * r * DP_N * 8);
    if (QUARTER == 3) wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)DP_N * 4u), (col_base + cp * 2) * 4, 0);
}

// ONE row load (the next tile is asked for in sixteen pieces over twelve request points of the task).  The lane offset of the tile is computed
// ONCE per task (fused_tile_voff) and kept in a register: recomputed at each of the sixteen points it was 6 % of the
// launch's vector instructions.
__device__ __forceinline__ int fused_tile_voff(int col_base)
{
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    return ((w * 8 + (l >> 3)) * DP_N + col_base + (l & 7) * 2) * 8;
}
template <int R>
__device__ __forceinline__ void fused_tile_load1(const float2 *src /* wave-uniform */, int voff, const float *wd,
                                                 float4 (&v)[16], float2 &wdv, bool valid)
{
    const rsrc_t rs = make_rsrc(src, valid ? (unsigned)RP_M * DP_N * 8u : 0u);
    v[R] = buf_load_f4<FUSED_INPUT_AUX>(rs, voff, 64 * R * DP_N * 8);
    if (R == 15) wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)DP_N * 4u), (voff & (DP_N * 8 - 1)) >> 1, 0);   // (col_base + 2 cp) * 4
}
// stage 1 (a2 + first radix of a3): window, radix 16 over rows p0 + 64 r, twiddle W_1024^{p0 k1};
// k1 < 8 goes to the LDS image (group 0), k1 >= 8 stays in ga / gc (group 1).  The lane's two columns
// are transformed ONE AFTER THE OTHER (and written as 8-byte halves of their 16-byte slots): both at
// once need more registers than the 128 that four waves per SIMD leave (165 in range_pass_1024).
// The lane's fifteen twiddles are the same for both columns: they are read from LDS ONCE per task, in ONE batch in front
// of the first butterfly (fused_stage1_tables), and a column's sixteen window values in one batch too -- read at their
// points of use, as the first form did, every one of the 46 reads is followed by its own s_waitcnt lgkmcnt(0) a few
// instructions later, and a tile wave (two per SIMD) then stands through 46 LDS round trips per task in this stage alone.
struct FusedStage1Tables { cf tw[16]; };
__device__ __forceinline__ void fused_stage1_tables(const unsigned char *smem, FusedStage1Tables &t)
{
    typedef FusedTile T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // per-lane LDS addresses are recomputed per tile, not hoisted + spilled
    tid &= FUSED_THREADS - 1;       // (the compiler knows the range again: address arithmetic folds)
    const int w = wave_id(), l = tid & 63;
    const int p0 = w * 8 + (l >> 3);
    const unsigned char *tw1 = smem + T::tw1_addr(p0, 0);   // k1 further on: a compile-time offset
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++) t.tw[k1] = *reinterpret_cast<const float2 *>(tw1 + T::tw1_addr(0, k1) - T::tw1_addr(0, 0));
}
// one wire sample of one channel, I and Q as big-endian int16 in a dword (sector.cpp:52-62) -> the two floats the CPU
// decode + scatter puts into the planar block (rpv2.cu:372-383); exact, so everything behind it is bit-identical.
// v[r] of the wire-format launch holds the RAW dwords (hh, vv) of the lane's two samples of row r: x, y first sample, z, w second.
__device__ __forceinline__ cf wire_sample(float bits)
{
    const unsigned t = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, bits), __builtin_bit_cast(unsigned, bits), 0x02030001u);   // swap the bytes of both halves
    return make_float2((float)(short)(t & 0xffffu), (float)(short)(t >> 16));
}
// the same for a sample whose HH and VV dwords lie in two registers: `sel` (wave-uniform) picks the channel in the byte
// permute that swaps the bytes anyway -- WIRE_SEL_HH: bytes of hh (S0 of v_perm_b32), WIRE_SEL_VV: bytes of vv (S1)
constexpr unsigned WIRE_SEL_HH = 0x06070405u, WIRE_SEL_VV = 0x02030001u;
__device__ __forceinline__ cf wire_sample2(float hh_bits, float vv_bits, unsigned sel)
{
    const unsigned t = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hh_bits), __builtin_bit_cast(unsigned, vv_bits), sel);
    return make_float2((float)(short)(t & 0xffffu), (float)(short)(t >> 16));
}
template <int COLUMN>   // 0: the lane's first column (v[r].xy), 1: its second (v[r].zw)
__device__ __forceinline__ void fused_stage1(unsigned char *smem, const float4 (&v)[16], float2 wdv, const FusedStage1Tables &t, cf (&g)[8])
{
    typedef FusedTile T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = wave_id(), l = tid & 63, cp = l & 7;
    const int p0 = w * 8 + (l >> 3);
    const int slot = T::addr(p0, cp) + 8 * (COLUMN ^ T::swz(p0));   // position k1*64 + p0 is 8 k1 blocks further on (same parity)
    const float *s_wr = reinterpret_cast<const float *>(smem + T::OFF_WR);
    float wr[16];    // one batch of eight reads (the window of the column's sixteen rows), one wait
#pragma unroll
    for (int r = 0; r < 16; r++) wr[r] = s_wr[p0 + 64 * r];
    cf a[16];
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = COLUMN ? make_float2(v[r].z, v[r].w) : make_float2(v[r].x, v[r].y);
    fft16_scaled<-1>(a, wr, COLUMN ? wdv.y : wdv.x);        // the window rides on the first butterflies (fft_radix.h)
    *reinterpret_cast<float2 *>(smem + slot) = a[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) *reinterpret_cast<float2 *>(smem + slot + k1 * 8 * T::BLK_BYTES) = cmul(a[k1], t.tw[k1]);
#pragma unroll
    for (int k1 = 8; k1 < 16; k1++) g[k1 - 8] = cmul(a[k1], t.tw[k1]);
}

// group 1 from its registers into the image (after group 0 has left it)
__device__ __forceinline__ void fused_group1_to_lds(unsigned char *smem, const cf (&ga)[8], const cf (&gc)[8])
{
    typedef FusedTile T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = wave_id(), l = tid & 63, cp = l & 7;
    const int p0 = w * 8 + (l >> 3);
    const int first = T::addr(p0, cp) + 8 * T::swz(p0), second = T::addr(p0, cp) + 8 * (1 ^ T::swz(p0));   // two conflict-free b64 = one b128 in LDS cycles
#pragma unroll
    for (int j = 0; j < 8; j++) {
        *reinterpret_cast<float2 *>(smem + first + j * 8 * T::BLK_BYTES) = ga[j];
        *reinterpret_cast<float2 *>(smem + second + j * 8 * T::BLK_BYTES) = gc[j];
    }
}

// stages 2 and 3 of the sub-transform this WAVE owns in the current group (image rows w*64 ..) and
// the stores of its gates: wave-private, no workgroup barrier in between.  ONE column per lane here
// (col = l & 15, two items per lane and stage, one after the other), which keeps these stages at
// ~40 registers -- the next tile's 64 registers are in flight meanwhile -- and every 8-byte LDS
// access conflict-free: a wave-instruction covers 4 positions x 128 contiguous bytes.  Same
// arithmetic per element as range_stage12/3.  Item `it` of stage 3 yields the gates
// k1 + 16 k2 + 128 k3 (k1 = w + 8 group, k2 = (l >> 4) + 4 it, k3 < 4) of column col; the chain never
// reads the other half of the gates (rpv2.cu:502).  Stores are plain: the lines stay in the XCD's
// L2, where the row members find them; 16 lanes x 8 bytes = one whole 128-byte line per gate.
// stages 2 and 3 are written per ITEM (two per lane and stage) so that the launch can place a request for the next tile
// between any two of them
template <int IT, bool SWZ = true>
__device__ __forceinline__ void fused_stage2_item(unsigned char *smem)
{
    typedef FusedTile T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = wave_id(), l = tid & 63, col = l & 15;
    // stage 2: radix 8 over positions p1 + 8 r (all of p1's parity), twiddle W_64^{p1 k2}, in place
    const int p1 = (l >> 4) + 4 * IT;
    unsigned char *base = smem + w * 8 * T::BLK_BYTES + (col ^ T::swz<SWZ>(p1)) * 8;   // position w*64 of this lane's column
    cf a[8], t[8];   // the seven twiddles in one batch with the data, not one LDS round trip each
#pragma unroll
    for (int r = 0; r < 8; r++) a[r] = *reinterpret_cast<const float2 *>(base + r * T::BLK_BYTES + p1 * T::ROW_BYTES);
#pragma unroll
    for (int k2 = 1; k2 < 8; k2++) t[k2] = *reinterpret_cast<const float2 *>(smem + T::tw2_addr(p1, k2));
    fft8<-1>(a);
    *reinterpret_cast<float2 *>(base + p1 * T::ROW_BYTES) = a[0];
#pragma unroll
    for (int k2 = 1; k2 < 8; k2++) *reinterpret_cast<float2 *>(base + k2 * T::BLK_BYTES + p1 * T::ROW_BYTES) = cmul(a[k2], t[k2]);
    // (both items' reads in one batch, the second item's twiddles read under the first item's butterfly -- two LDS round
    // trips in the open instead of four, 128 registers -- measured the same in the launch: profiles/r03/ab_stage23_interleaved.log)
    if (IT == 1) wave_lds_fence();
}
__device__ __forceinline__ void fused_stage2(unsigned char *smem)
{
    fused_stage2_item<0>(smem);
    fused_stage2_item<1>(smem);
}
template <int IT, bool SWZ = true>
__device__ __forceinline__ void fused_stage3_item(unsigned char *smem, cf (&o)[2][4])
{
    typedef FusedTile T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = wave_id(), l = tid & 63, col = l & 15;
    unsigned char *base = smem + w * 8 * T::BLK_BYTES + col * 8, *base_odd = smem + w * 8 * T::BLK_BYTES + (col ^ T::swz<SWZ>(1)) * 8;
    const int k2 = (l >> 4) + 4 * IT;     // stage 3: radix 8 over the 8 contiguous positions k2*8 + r
    cf a[8];
#pragma unroll
    for (int r = 0; r < 8; r++) a[r] = *reinterpret_cast<const float2 *>((r & 1 ? base_odd : base) + k2 * T::BLK_BYTES + r * T::ROW_BYTES);
    fft8<-1>(a);
#pragma unroll
    for (int k3 = 0; k3 < 4; k3++) o[IT][k3] = a[k3];
}
__device__ __forceinline__ void fused_stage3(unsigned char *smem, cf (&o)[2][4])
{
    fused_stage3_item<0>(smem, o);
    fused_stage3_item<1>(smem, o);
}
// tee (diagnostics instantiation only, nullptr otherwise): the task's [512 gates][512] block of a copy of the whole
// intermediate in global memory -- what tests compare with the two-kernel path's WRP_STAGE_MID, both halves
__device__ __forceinline__ void fused_store(float2 *mid /* wave-uniform */, int col_base, int group, const cf (&o)[2][4], float2 *tee = nullptr)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = wave_id(), l = tid & 63, col = l & 15;
    const rsrc_t rd = make_rsrc(mid, (unsigned)FUSED_SLOT_ROWS * DP_N * 8u);
    // slot row of gate k1 + 16 k2 + 128 k3 (k1 = w + 8 group, k2 = (l >> 4) + 4 it): (gate >> 4) * 8 + (gate & 7)
    const int voff = ((w + 8 * (l >> 4)) * DP_N + col_base + col) * 8;
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) {   // row offset in the VGPR, soffset 0: see buf_store_f4
            v2f t;
            t.x = o[it][k3].x; t.y = o[it][k3].y;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, t), rd, voff + (32 * it + 64 * k3) * DP_N * 8, 0, 0);
        }
    if (tee) {
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int k3 = 0; k3 < 4; k3++) tee[(size_t)(w + 8 * group + 16 * ((l >> 4) + 4 * it) + 128 * k3) * DP_N + col_base + col] = o[it][k3];
    }
}

// Who am I: XCD, kind (the older / the younger of the two workgroups on this CU), rank inside the team.  Everything here
// stays inside the XCD (L2 atomics of workgroup scope, polls served by that L2): a team needs no other team.  The first
// version met grid-wide (agent-scope arrivals with release, 512 pollers on one word): 31-35 us of every launch.
// What makes the local form sufficient: a CU holds at most two of these workgroups (LDS), so an XCD holds at most 64;
// a grid of 64 T workgroups on T XCDs that are all resident therefore puts exactly two on every CU -- one older (tile),
// one younger (row) -- and 32 + 32 on every XCD.  A team waits (bounded) until ITS census says so; XCC ids must be
// 0 .. T-1 (team x owns sectors x, x + T, ...).  Anything else -- another kernel on the GPU, a partitioned device, the
// undersized test launch -- ends in status 2 and the engine's two-kernel path.
struct FusedSeat { int ok, xcc, kind, rank, teams, trank; };
__device__ __forceinline__ FusedSeat fused_join(FusedCtl *ctl, lds_word *s_ctl)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned x = xcc_id(), key = hw_cu_key();
        const unsigned a = __hip_atomic_fetch_add(&ctl->cu_arrivals[x][key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // The two workgroups of a CU meet: the OLDER one (lower blockIdx: dispatched first) is the tile workgroup.
        // A SIMD issues from its oldest ready wave first, and the tile waves are the critical path of a task; where
        // the younger workgroup had the tile role (it wins the race to the counter on ~3 % of the CUs), that member
        // ran 10 % slower than the others for the whole launch (scalar-probe build, members running free).
        // Both waits of the meeting have a deadline in REAL time (s_memrealtime, 100 MHz): FUSED_JOIN_TICKS = 4 ms; the
        // workgroups of a launch that has the GPU to itself arrive within microseconds of each other.
        const unsigned long long t_join = __builtin_amdgcn_s_memrealtime();
        unsigned other = 0;
        if (a < 2) {
            __hip_atomic_store(&ctl->cu_block[x][key][a], blockIdx.x + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll 1
            while (!other && __builtin_amdgcn_s_memrealtime() - t_join < FUSED_JOIN_TICKS) {
                other = __hip_atomic_load(&ctl->cu_block[x][key][a ^ 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: from the L2
                if (!other) __builtin_amdgcn_s_sleep(4);
            }
        }
        const unsigned kind = other ? (blockIdx.x + 1u < other ? 0u : 1u) : (a & 1u);
        const unsigned rank = __hip_atomic_fetch_add(&ctl->census[kind][x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned teams = gridDim.x / (2u * FUSED_MEMBERS);
        int good = other != 0 && x < teams && gridDim.x == teams * 2u * FUSED_MEMBERS;
        if (good) {   // the team meets: both kinds complete
            good = 0;
#pragma unroll 1
            while (__builtin_amdgcn_s_memrealtime() - t_join < FUSED_JOIN_TICKS) {
                const unsigned ct = __hip_atomic_load(&ctl->census[0][x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned cr = __hip_atomic_load(&ctl->census[1][x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ct + cr >= 2u * FUSED_MEMBERS) { good = ct == (unsigned)FUSED_MEMBERS && cr == (unsigned)FUSED_MEMBERS; break; }
                if (__hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (!good) __hip_atomic_store(&ctl->status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[0] = good;
        s_ctl[1] = (int)x;
        s_ctl[2] = (int)kind;
        s_ctl[3] = (int)rank;
        s_ctl[4] = (int)teams;
        s_ctl[5] = (int)x;
    }
    __syncthreads();
    FusedSeat r;   // wave-uniform by construction; readfirstlane tells the compiler so (scalar address arithmetic)
    r.ok = __builtin_amdgcn_readfirstlane(s_ctl[0]);
    r.xcc = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    r.kind = __builtin_amdgcn_readfirstlane(s_ctl[2]);
    r.rank = __builtin_amdgcn_readfirstlane(s_ctl[3]);
    r.teams = __builtin_amdgcn_readfirstlane(s_ctl[4]);
    r.trank = __builtin_amdgcn_readfirstlane(s_ctl[5]);
    return r;
}

// The end of a workgroup's work.  Failure: tell the host (a word in pinned host memory; nothing is copied back after a
// launch).  Success: the last of the team's 64 workgroups to get here zeroes the team's part of the control block, so
// the next launch needs no memset in front of it.
__device__ __forceinline__ void fused_leave(FusedCtl *ctl, unsigned *host_status, int xcc, lds_word *s_ctl)
{
    const int tid = threadIdx.x;
    // every flag byte this wave has published is in the L2 before the workgroup counts itself as done: the team's last
    // workgroup zeroes the flag lines, and a byte store still in flight (another L2 channel than the counter) would
    // land behind the zeroing and deadlock the next launch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned st = __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st) __hip_atomic_store(host_status, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_ctl[6] = !st && __hip_atomic_fetch_add(&ctl->done[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 2u * FUSED_MEMBERS - 1u;
    }
    __syncthreads();
    if (!s_ctl[6]) return;
    uint4 *z;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    for (int g = 0; g < 2; g++) {
        z = reinterpret_cast<uint4 *>(ctl->stored[g][xcc]);
        for (int e = tid; e < (int)(FUSED_MEMBERS * sizeof(FusedFlags) / 16); e += FUSED_THREADS) z[e] = zero;
        z = reinterpret_cast<uint4 *>(ctl->loaded[g][xcc]);
        for (int e = tid; e < (int)(FUSED_MEMBERS * sizeof(FusedFlags) / 16); e += FUSED_THREADS) z[e] = zero;
    }
    z = reinterpret_cast<uint4 *>(ctl->cu_arrivals[xcc]);
    for (int e = tid; e < 256 / 4; e += FUSED_THREADS) z[e] = zero;
    z = reinterpret_cast<uint4 *>(ctl->cu_block[xcc]);
    for (int e = tid; e < 512 / 4; e += FUSED_THREADS) z[e] = zero;
    if (tid == 0) { ctl->census[0][xcc] = 0; ctl->census[1][xcc] = 0; ctl->done[xcc] = 0; }
}

// ---- wire-format input: the tile workgroups read the sector as it arrives -------------------------------
// 12 bytes per sample: hhI hhQ vvI vvQ vhI vhQ, big-endian int16 (sector.cpp:52-62), [1024 rows][512 samples].
// Lane mapping of the wire-format tile: ONE column (sample) per lane and 32 rows -- lane (c = l & 15, pq = 4 w + (l >> 4))
// holds the (hh, vv) dwords of rows pq + 32 r, r < 32: an 8-byte load per row at a 12-byte lane stride, so every 128-byte
// line of a row's 192-byte segment is touched by exactly ONE instruction (two samples per lane, the planar mapping, touch
// every line twice: 3.30 us/sector non-temporal, 2.96 with the second touch served by the L1 -- and then the input evicts
// the hand-over slot: WRITE_SIZE 2.3 MB per sector; profiles/r03/ab_wire_input_policy.log).  The even rows r = 2 r' are
// the sixteen inputs of the radix-16 butterfly of position p0 = pq, the odd ones those of p0 = pq + 32: the lane's two
// ITEMS (the planar mapping's are two columns at one p0).  ONE set of 64 registers feeds BOTH channel-tasks of the sector,
// VH is never fetched into a register, byte swap + conversion cost three instructions per sample and channel in stage 1.
// HBM: 6 MiB per sector instead of 8, and no decode pass (6 MiB read + 8 MiB written + 8 MiB read again).
// ONE piece of the wire-format tile: the rows pq + 64 RR and pq + 32 + 64 RR (sixteen pieces, requested one at a time:
// eight in each of the sector's two tasks)
template <int RR, int WB>
__device__ __forceinline__ void fused_raw_tile_load1(const unsigned *sector_raw /* wave-uniform */, int col_base, const float *wd,
                                                     float2 (&v)[32], float &wdv, bool valid)
{
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int pq = w * 4 + (l >> 4), c = l & 15;
    const rsrc_t rs = make_rsrc(sector_raw, valid ? (unsigned)RP_M * DP_N * WB : 0u);
    const int voff = (pq * DP_N + col_base + c) * WB;
    v[2 * RR] = buf_load_f2<AUX_NT>(rs, voff, 64 * RR * DP_N * WB);
    v[2 * RR + 1] = buf_load_f2<AUX_NT>(rs, voff, (64 * RR + 32) * DP_N * WB);
    if (RR == 15) {
        const float4 f = buf_load_f4<0>(make_rsrc(wd, (unsigned)DP_N * 4u), ((col_base + c) & ~3) * 4, 0);
        wdv = (c & 3) == 0 ? f.x : (c & 3) == 1 ? f.y : (c & 3) == 2 ? f.z : f.w;
    }
}
template <int QUARTER, int WB>
__device__ __forceinline__ void fused_raw_tile_load(const unsigned *sector_raw /* wave-uniform */, int col_base, const float *wd,
                                                    float2 (&v)[32], float &wdv, bool valid)
{
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int pq = w * 4 + (l >> 4), c = l & 15;
    const rsrc_t rs = make_rsrc(sector_raw, valid ? (unsigned)RP_M * DP_N * WB : 0u);
    const int voff = (pq * DP_N + col_base + c) * WB;
#pragma unroll
    for (int rr = QUARTER; rr < 16; rr += 4) {   // rows pq + 64 rr (item 0) and pq + 32 + 64 rr (item 1)
        v[2 * rr] = buf_load_f2<AUX_NT>(rs, voff, 64 * rr * DP_N * WB);
        v[2 * rr + 1] = buf_load_f2<AUX_NT>(rs, voff, (64 * rr + 32) * DP_N * WB);
    }
    if (QUARTER == 3) {   // the lane's Doppler-window value out of an aligned 16-byte load (element-wise use of a narrower
                          // buffer load's result is miscompiled by this hipcc: see buf_load_f4)
        const float4 f = buf_load_f4<0>(make_rsrc(wd, (unsigned)DP_N * 4u), ((col_base + c) & ~3) * 4, 0);
        wdv = (c & 3) == 0 ? f.x : (c & 3) == 1 ? f.y : (c & 3) == 2 ? f.z : f.w;
    }
}

// stage 1 of ITEM (0: position pq, 1: pq + 32) of the lane's column; s[r] = the channel's dword of row pq + 32 r.  The
// twiddles are read behind the butterfly in two batches of eight (the registers hold the other channel's dwords or the
// next sector's first pieces: no room for fifteen)
template <int ITEM>
__device__ __forceinline__ void fused_stage1_raw(unsigned char *smem, const float (&s)[32], float wdv, cf (&g)[8])
{
    typedef FusedTile T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = wave_id(), l = tid & 63, c = l & 15;
    const int p0 = w * 4 + (l >> 4) + 32 * ITEM;
    const int slot = T::addr(p0, c >> 1) + 8 * (c & 1);
    const float *s_wr = reinterpret_cast<const float *>(smem + T::OFF_WR);
    const unsigned char *tw1 = smem + T::tw1_addr(p0, 0);
    cf a[16];
    float wgt[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        wgt[r] = s_wr[p0 + 64 * r];
        a[r] = wire_sample(s[2 * r + ITEM]);
    }
    fft16_scaled<-1>(a, wgt, wdv);
    *reinterpret_cast<float2 *>(smem + slot) = a[0];
    cf t[8];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) t[k1] = *reinterpret_cast<const float2 *>(tw1 + T::tw1_addr(0, k1) - T::tw1_addr(0, 0));
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) *reinterpret_cast<float2 *>(smem + slot + k1 * 8 * T::BLK_BYTES) = cmul(a[k1], t[k1]);
#pragma unroll
    for (int k1 = 8; k1 < 16; k1++) t[k1 - 8] = *reinterpret_cast<const float2 *>(tw1 + T::tw1_addr(0, k1) - T::tw1_addr(0, 0));
#pragma unroll
    for (int k1 = 8; k1 < 16; k1++) g[k1 - 8] = cmul(a[k1], t[k1 - 8]);
}
__device__ __forceinline__ void fused_raw_group1_to_lds(unsigned char *smem, const cf (&g0)[8], const cf (&g1)[8])
{
    typedef FusedTile T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = wave_id(), l = tid & 63, c = l & 15;
    const int pq = w * 4 + (l >> 4);
    const int slot = T::addr(pq, c >> 1) + 8 * (c & 1);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        *reinterpret_cast<float2 *>(smem + slot + j * 8 * T::BLK_BYTES) = g0[j];
        *reinterpret_cast<float2 *>(smem + slot + j * 8 * T::BLK_BYTES + 4 * T::BLK_BYTES) = g1[j];   // position + 32 = four blocks on
    }
}

// The tile member of the wire-format launch: the loop of fused_chain_1024x512's tile member with the two channel-tasks of
// a sector unrolled.  Same hand-over protocol, same sequence numbers.  The next sector's sixteen pieces are requested EIGHT
// PER TASK: task HH's stage 1 leaves the HH half of every landing pair dead, but an 8-byte load needs a whole pair -- so
// behind that stage the VV dwords are MOVED out of the pairs (32 v_mov per lane and sector), the pairs of the pieces
// 0 .. 7 are free for the rest of task HH, and task VV, whose stage 1 reads the moved dwords, requests the pieces 8 .. 15.
// Round 3 requested all sixteen in task VV (twice the planar launch's request rate while it lasted, none in task HH):
// 2.49 -> 2.17 us/sector, -12.7 % (profiles/r04/ab_wire_requests_in_both_tasks.log), no spilled register (4 before).
// The moves are written as instructions: left to the register allocator (`vv[r] = v[r].y`, it inserts the same 32
// copies) the launch is 1.3 % slower.  Eight pieces on the points 0 1 2 3 | 5 | 6 7 9 of a task's eleven; seven other
// placements +0.0 ... +5.5 % (ab_wire_request_schedules.log).
template <int WB>   // bytes per wire sample: 12, or 8 (VH dropped by the feeder: the 16 lanes of a row segment then cover ONE 128-byte line)
__device__ __forceinline__ void fused_raw_tile_member(unsigned char *smem, const unsigned *raw, float2 *mid, FusedCtl *ctl,
                                                      const RangeConsts &rc, int xcc, int rank, int teams, int trank, int tasks,
                                                      int channels, float2 *tee /* diagnostics: see fused_store; [S][channels][512][512] */)
{
    typedef FusedTile T;
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    // the tile of a sector (both its tasks): rotated from sector to sector as in the planar launch
    auto tile_col = [&](int sec) { return ((rank + sec) & (FUSED_MEMBERS - 1)) * 16; };
    auto sector_src = [&](int sec) { return raw + (size_t)(trank + sec * teams) * RP_M * DP_N * (WB / 4); };
    const int sectors = tasks >> 1;
    float2 v[32];     // the landing pairs: (hh, vv) dwords of the rows pq + 32 r
    float vv[32];     // the VV dwords of the sector in work, moved out of the pairs behind task HH's stage 1
    float wdv;
    fused_raw_tile_load<0, WB>(sector_src(0), tile_col(0), rc.wd, v, wdv, sectors > 0);
    fused_raw_tile_load<1, WB>(sector_src(0), tile_col(0), rc.wd, v, wdv, sectors > 0);
    fused_raw_tile_load<2, WB>(sector_src(0), tile_col(0), rc.wd, v, wdv, sectors > 0);
    fused_raw_tile_load<3, WB>(sector_src(0), tile_col(0), rc.wd, v, wdv, sectors > 0);
    for (int e = tid; e < RP_M; e += FUSED_THREADS) {
        const int p0 = e >> 4, k1 = e & 15;
        *reinterpret_cast<float2 *>(smem + T::tw1_addr(p0, k1)) = rc.tw[(p0 * k1) & (RP_M - 1)];
        reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
    }
    if (tid < 64) *reinterpret_cast<float2 *>(smem + T::tw2_addr(tid >> 3, tid & 7)) = rc.tw[(16 * (tid >> 3) * (tid & 7)) & (RP_M - 1)];
    __syncthreads();
    int *s_arrived = reinterpret_cast<int *>(smem + T::OFF_CTL + 56);
    if (tid < 2) s_arrived[tid] = 0;
    __syncthreads();
    const FusedFlags *my_loaded0 = &ctl->loaded[0][xcc][rank], *my_loaded1 = &ctl->loaded[1][xcc][rank];
    int failed = 0;
    // half 1 of the previous task: drained, counted by the last wave (between the two items of stage 1, as in the planar launch)
    auto drained = [&](int q) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (l == 0) last = atomicAdd(s_arrived, 1) == 8 * q + 7;
        if (q > 0 && __builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->stored[1][xcc], l, rank, (unsigned)q);
    };
    // one channel-task; it requests eight of the next sector's sixteen pieces (valid = false behind the last sector)
    auto task = [&](auto chc, int q, int col, const unsigned *next, int next_col, bool more) {
        constexpr int CH = decltype(chc)::value;
        float2 *tee_task = tee ? tee + ((size_t)(trank + (q >> 1) * teams) * channels + CH) * (RP_M / 2) * DP_N : nullptr;
        cf ga[8], gc[8];
        if (CH == 0) {
            float hh[32];
#pragma unroll
            for (int r = 0; r < 32; r++) hh[r] = v[r].x;
            fused_stage1_raw<0>(smem, hh, wdv, ga);
            drained(q);
            fused_stage1_raw<1>(smem, hh, wdv, gc);
#pragma unroll
            for (int r = 0; r < 32; r++) asm volatile("v_mov_b32 %0, %1" : "=v"(vv[r]) : "v"(v[r].y));
        } else {
            fused_stage1_raw<0>(smem, vv, wdv, ga);
            drained(q);
            fused_stage1_raw<1>(smem, vv, wdv, gc);
        }
        __syncthreads();                    // A1
        cf o[2][4];
#define WRP_LR(K) fused_raw_tile_load1<8 * CH + (K), WB>(next, next_col, rc.wd, v, wdv, more)
        WRP_LR(0);
        fused_stage2_item<0, false>(smem);
        WRP_LR(4);
        fused_stage2_item<1, false>(smem);
        WRP_LR(2);
        fused_stage3_item<0, false>(smem, o);
        WRP_LR(6);
        fused_stage3_item<1, false>(smem, o);
        spin_flags_sticky(my_loaded1, (unsigned)q, failed, w != 0);
        __syncthreads();                    // A2
        fused_store(mid, col, 0, o, tee_task);
        __builtin_amdgcn_sched_barrier(0);
        WRP_LR(1);
        fused_raw_group1_to_lds(smem, ga, gc);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // all but the 2 requests just issued: the stores are in the L2
        int last = 0;
        if (l == 0) last = atomicAdd(s_arrived + 1, 1) == 8 * q + 7;
        if (__builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->stored[0][xcc], l, rank, (unsigned)(q + 1));
        __syncthreads();                    // A3
        WRP_LR(5);
        fused_stage2_item<0, false>(smem);
        WRP_LR(3);
        fused_stage2_item<1, false>(smem);
        fused_stage3_item<0, false>(smem, o);
        WRP_LR(7);
        fused_stage3_item<1, false>(smem, o);
#undef WRP_LR
        spin_flags_sticky(my_loaded0, (unsigned)(q + 1), failed, w != 0);
        __syncthreads();                    // A4
        fused_store(mid, col, 1, o, tee_task);
    };
#pragma unroll 1
    for (int sec = 0; sec < sectors; sec++) {
        const bool more = sec + 1 < sectors;
        const unsigned *next = sector_src(more ? sec + 1 : 0);
        task(std::integral_constant<int, 0>{}, 2 * sec, tile_col(sec), next, tile_col(sec + 1), more);
        task(std::integral_constant<int, 1>{}, 2 * sec + 1, tile_col(sec), next, tile_col(sec + 1), more);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tasks > 0 && w == 0) l2_flag32(ctl->stored[1][xcc], l, rank, (unsigned)tasks);
    if (failed && l == 0) __hip_atomic_store(&ctl->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// target("no-load-store-opt"): hipcc's load / store optimiser merges neighbouring LDS accesses into ds_read2_b64 /
// ds_write2_b64 (one address register, two offsets).  On gfx950 a ds_read2_b64 is served in 16-lane groups and takes 7.9 LDS
// cycles where two ds_read_b64 (32-lane groups) take 4.6 (tools/ldsopbench.hip, profiles/r04/ldsopbench.log): the tile and row
// stages read 51 + 10 such pairs per wave and task.  Without the pass: bit-identical, 123 instead of 128 VGPRs, -1.6 %
// (wire format -3.5 %, 4 instead of 6 spilled registers; profiles/r04/ab_no_ds_read2.log).  The 2048 x 128 launch keeps the
// pass (+2.3 % without it).
#if defined(__HIP_DEVICE_COMPILE__)       // (a code-generation attribute of the gfx950 pass; the host pass does not know it)
#define WRP_NO_DS_MERGE __attribute__((target("no-load-store-opt")))
#else
#define WRP_NO_DS_MERGE
#endif
template <int TAPS, bool STAMPS, int RAW = 0 /* wire-format input: bytes per sample (12 or 8), 0 = the planar block */, bool TEE = false>
__global__ __launch_bounds__(FUSED_THREADS, 4) __attribute__((amdgpu_waves_per_eu(4, 4))) WRP_NO_DS_MERGE void fused_chain_1024x512(
    const float2 *__restrict__ iq,   // [S][C][1024][512]; RAW: the wire format, [S][1024 x 512][RAW bytes]
    float *__restrict__ out,         // [S][512][2]
    float2 *pool,                    // [8][FUSED_TEAM_ELEMS] per team: ONE slot[256][512] through which both halves go
    FusedCtl *ctl, RangeConsts rc, const float2 *__restrict__ tw_n, int n_sectors, int channels, MaTaps taps,
    float k_rr, float k_cal, unsigned *host_status /* pinned host word of this launch */,
    unsigned long long *stamps /* diagnostics: [grid][FUSED_STAMP_TASKS][8] or nullptr */,
    unsigned *frames /* optional (N2): [S][2][1 + 512] words, the products framed for the wire */, const unsigned *frame_hdrs /* [S] header words */,
    float2 *tee /* diagnostics (TEE instantiations): [S][C][512][512], a copy of everything that goes through the slots */)
{
    typedef FusedTile T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_word *s_ctl = (lds_word *)(smem + T::OFF_CTL);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    const int n = DP_N, gates = RP_M / 2;

    const FusedSeat seat = fused_join(ctl, s_ctl);
    const int xcc = seat.xcc, kind = seat.kind, rank = seat.rank, teams = seat.teams, trank = seat.trank;
    if (!seat.ok || rank >= FUSED_MEMBERS) {   // no team, or a surplus workgroup (then the census has failed too)
        fused_leave(ctl, host_status, xcc, s_ctl);
        return;
    }
    const int tasks = 2 * ((n_sectors - trank + teams - 1) / teams);   // channel-tasks of this team
    float2 *mid = pool + (size_t)xcc * FUSED_TEAM_ELEMS;
    // diagnostics build only: stamps go to LDS (no vector-memory traffic in the timed phases) and
    // are copied out at the end by wave 0
    unsigned long long *s_stamps = reinterpret_cast<unsigned long long *>(smem + T::OFF_STAMPS);
    auto stamp = [&](int q, int k) {   // wave 0 -> slots k; row kind: also wave 4 (half 1) -> slots 4 + k
        if (STAMPS && (w == 0 || (kind == 1 && w == 4)) && q < FUSED_STAMP_TASKS) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (l == 0) s_stamps[q * FUSED_STAMPS + k + w] = t;
        }
    };
    if (STAMPS) {
        for (int e = tid; e < FUSED_STAMP_TASKS * FUSED_STAMPS; e += FUSED_THREADS) s_stamps[e] = 0;
        __syncthreads();
        if (tid == 0) s_stamps[FUSED_STAMPS - 1] = ((unsigned long long)kind << 32) | ((unsigned long long)xcc << 16) | (unsigned)rank;
    }
    // diagnostics build: the shader clock this workgroup saw over its task loop = d(s_memtime) / d(s_memrealtime) x 100 MHz
    // (slot 8 of tasks 1 and 2): the launch is clock-limited by the power its DATA draws
    const unsigned long long clk_t0 = STAMPS ? __builtin_amdgcn_s_memtime() : 0, clk_r0 = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0;
    auto flush_stamps = [&]() {
        if (STAMPS && stamps) {
            if (tid == 0) {
                s_stamps[1 * FUSED_STAMPS + FUSED_STAMPS - 1] = __builtin_amdgcn_s_memtime() - clk_t0;
                s_stamps[2 * FUSED_STAMPS + FUSED_STAMPS - 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
            }
            __syncthreads();
            for (int e = tid; e < FUSED_STAMP_TASKS * FUSED_STAMPS; e += FUSED_THREADS)
                stamps[(size_t)blockIdx.x * FUSED_STAMP_TASKS * FUSED_STAMPS + e] = s_stamps[e];
        }
    };

    if (RAW != 0 && kind == 0) {
        fused_raw_tile_member<RAW ? RAW : 12>(smem, reinterpret_cast<const unsigned *>(iq), mid, ctl, rc, xcc, rank, teams, trank, tasks, channels, TEE ? tee : nullptr);
        fused_leave(ctl, host_status, xcc, s_ctl);
    } else if (kind == 0) {
        // =============================== tile member ===============================
        // Tile of task q: (rank + q) mod 32.  The column tiles 3, 11, 19, 27 -- byte offset 384 mod 1024 of
        // every 4 KiB row -- load 18 % slower than the others on this chip (tools/storeskew.hip; an
        // address-interleave artefact of the 4 KiB row stride), and a team moves at the pace of its slowest
        // member: rotating the tiles makes that a transient of each member, which the slack of the
        // hand-overs absorbs, instead of four members that are always late.
        auto tile_col = [&](int q) { return ((rank + q) & (FUSED_MEMBERS - 1)) * 16; };
        auto tile_src = [&](int q) { return iq + ((size_t)(trank + (q >> 1) * teams) * channels + (q & 1)) * RP_M * (size_t)n; };
        float4 v[16];
        float2 wdv;
        fused_tile_load<0>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);   // HBM requests first ...
        fused_tile_load<1>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);
        fused_tile_load<2>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);
        fused_tile_load<3>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);
        for (int e = tid; e < RP_M; e += FUSED_THREADS) {                 // ... tables while they fly
            const int p0 = e >> 4, k1 = e & 15;
            *reinterpret_cast<float2 *>(smem + T::tw1_addr(p0, k1)) = rc.tw[(p0 * k1) & (RP_M - 1)];
            reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
        }
        if (tid < 64) *reinterpret_cast<float2 *>(smem + T::tw2_addr(tid >> 3, tid & 7)) = rc.tw[(16 * (tid >> 3) * (tid & 7)) & (RP_M - 1)];
        __syncthreads();
        // The tile waves are the critical path of a task and the row waves of this CU have slack: when
        // both want the SIMD, the tile wave goes first.
        // (s_setprio for the tile waves was measured: see DESIGN.md)
        int *s_arrived = reinterpret_cast<int *>(smem + T::OFF_CTL + 56);   // two arrival counters (the row kind uses +48, +52)
        if (tid < 2) s_arrived[tid] = 0;
        __syncthreads();
        const FusedFlags *my_loaded0 = &ctl->loaded[0][xcc][rank], *my_loaded1 = &ctl->loaded[1][xcc][rank];
        int failed = 0;
#pragma unroll 1
        for (int q = 0; q < tasks; q++) {
            cf ga[8], gc[8];
            stamp(q, 0);
            FusedStage1Tables s1t;
            fused_stage1_tables(smem, s1t);
            fused_stage1<0>(smem, v, wdv, s1t, ga);
            // Half 1 of the previous tile was stored half a stage ago: its drain costs nothing here.  It is
            // counted NOW, while this CU has no request in flight (the row members load that half at once,
            // and L2 hits of a CU queue -- 3.5 us measured -- behind a request burst that went out before
            // them), by the last wave to get here: an arrival count in LDS instead of a barrier.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int last = 0;
            if (l == 0) last = atomicAdd(s_arrived, 1) == 8 * q + 7;
            if (q > 0 && __builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->stored[1][xcc], l, rank, (unsigned)q);
            fused_stage1<1>(smem, v, wdv, s1t, gc);
            __syncthreads();                    // A1: group 0 is in the image
            stamp(q, 1);
            // v is free: the next tile is requested a quarter at a time over the rest of this one
            const float2 *next = tile_src(q + 1 < tasks ? q + 1 : 0);
            cf o[2][4];
            const int voff_next = fused_tile_voff(tile_col(q + 1));
#define WRP_L1(R) fused_tile_load1<R>(next, voff_next, rc.wd, v, wdv, q + 1 < tasks)
            WRP_L1(0); WRP_L1(8);
            fused_stage2_item<0>(smem);
            WRP_L1(4);
            fused_stage2_item<1>(smem);
            WRP_L1(12);
            fused_stage3_item<0>(smem, o);
            WRP_L1(1);
            fused_stage3_item<1>(smem, o);
            WRP_L1(9);
            // The team's ONE slot (1 MiB: the 256 gates of a half x 512 pulses) takes half 0 and half 1 of every task in
            // turn; it still holds half 1 of task q-1 until every row member has those rows in registers.  One slot
            // instead of one per half: a rewritten buffer of 2 MiB per XCD does not stay in the 4 MiB L2 beside the
            // streaming input (85 % of it was written back every task, 27 % fetched again, and the latest of those
            // row loads is what every tile member waited for); 1 MiB does -- write-backs 3.5 -> 0.5 MB per sector at
            // the same speed.  The look comes as late as it can: one wave, in front of the barrier behind which the
            // stores go out.
            spin_flags_sticky(my_loaded1, (unsigned)q, failed, w != 0);
            __syncthreads();                    // A2: group 0 has left the image; the slot is free for half 0
            float2 *tee_task = TEE && tee ? tee + ((size_t)(trank + (q >> 1) * teams) * channels + (q & 1)) * (RP_M / 2) * n : nullptr;
            fused_store(mid, tile_col(q), 0, o, tee_task);
            // BEHIND the stores, so that a counted wait can tell them apart: the scheduling barrier keeps the four loads
            // below the eight stores whatever alias analysis says about `iq` (restrict) and the descriptor
            __builtin_amdgcn_sched_barrier(0);
            WRP_L1(5); WRP_L1(13); WRP_L1(2); WRP_L1(10);
            fused_group1_to_lds(smem, ga, gc);
            stamp(q, 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the 4 requests just issued: the stores are in the L2
            // ... and counted by the last wave to get here, without waiting for the barrier
            last = 0;
            if (l == 0) last = atomicAdd(s_arrived + 1, 1) == 8 * q + 7;
            if (__builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->stored[0][xcc], l, rank, (unsigned)(q + 1));
            stamp(q, 6);
            __syncthreads();                    // A3: group 1 is in the image
            stamp(q, 3);
            WRP_L1(6);
            fused_stage2_item<0>(smem);
            WRP_L1(14);
            fused_stage2_item<1>(smem);
            WRP_L1(3);
            fused_stage3_item<0>(smem, o);
            WRP_L1(11);
            fused_stage3_item<1>(smem, o);
            WRP_L1(7); WRP_L1(15);
            spin_flags_sticky(my_loaded0, (unsigned)(q + 1), failed, w != 0);
            stamp(q, 7);
            __syncthreads();                    // A4: image free for the next stage 1; the slot is free for half 1 (the rows have half 0 of THIS task)
            fused_store(mid, tile_col(q), 1, o, tee_task);
            stamp(q, 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tasks > 0 && w == 0) l2_flag32(ctl->stored[1][xcc], l, rank, (unsigned)tasks);
        if (failed && l == 0) __hip_atomic_store(&ctl->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flush_stamps();
        fused_leave(ctl, host_status, xcc, s_ctl);
    } else {
        // =============================== row member ===============================
        // Eight independent waves, no workgroup barrier in the loop.  A wave polls the member's flag line itself (scalar
        // loads); the last of a half's waves to have its row(s) in registers (counted in LDS) counts the member as loaded.
        float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;
        float2 *s_twn = reinterpret_cast<float2 *>(smem + T::OFF_TWN);
        doppler_twiddles_to_lds(s_twn, tw_n, tid, FUSED_THREADS);
        if (tid < 4) s_ctl[12 + tid] = 0;   // 12, 13: arrival counts of the halves
        __syncthreads();
        DopplerTwiddles row_tw;     // the lane's fourteen twiddles stay in registers for all of the wave's rows (wrp_kernels.h)
        doppler_row_twiddles(s_twn, l, row_tw);
        const DumpPtrs nodump{};
        // Every wave serves BOTH halves, one row of each: a half is then loaded by eight waves with 8 requests each instead
        // of four waves with 16 -- the row loads sit in the hand-over chain the tile members wait for at their looks.  A wave
        // has finished its row of one half well before the other half is published (a row is ~2 us of the ~10 us task).
        const int r0 = rank * 8 + w;                            // slot row of this wave's gate in either half
        float s_hh[2] = {0.f, 0.f};                             // HH row sums of its two gates, waiting for the VV task
#pragma unroll 1
        for (int q = 0; q < tasks; q++) {
            bool there = true;
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int gate = rank * 16 + 8 * g + w;
                if (g == 0) stamp(q, 0);
                there = there && spin_flags(&ctl->stored[g][xcc][rank], (unsigned)(q + 1), &ctl->status);
                if (!there) break;                              // status is set: the launch is void
                if (g == 0) stamp(q, 1);
                // From the notice to the publication of `loaded` the wave runs at RAISED PRIORITY: these two dozen
                // instructions are in the hand-over chain every tile member of the team waits for, and a row wave is the
                // YOUNGER wave of its SIMD -- the arbiter serves the older tile waves first and leaves it the gaps.
                // 2.32 -> 2.26 us/sector.  (Raised while polling as well: -0.5 % only, the polls then take slots from the
                // tile waves; tile waves raised: no change; the row transform itself raised: +6 %.
                // profiles/r03/ab_wave_priority.log: prio2 / prio3 / prio1 / prio7.)
                __builtin_amdgcn_s_setprio(FUSED_ROW_PRIO);
                cf x[8];
                // (slot rows in the order the lanes read them -- pulses l + 128 k and l + 64 + 128 k side by side, four
                // 16-byte loads per lane instead of eight 8-byte ones -- cost the tile members scattered 8-byte stores and
                // were 0.8 % slower: profiles/r03/ab_paired_slot_rows.log)
                doppler_load_row<AUX_SC1>(mid + (size_t)r0 * n, l, x);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // row in registers: the slot may be overwritten
                int last = 0;
                if (l == 0) last = atomicAdd(reinterpret_cast<int *>(smem + T::OFF_CTL + 48 + 4 * g), 1) == 8 * q + 7;
                if (__builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->loaded[g][xcc], l, rank, (unsigned)(q + 1));
                __builtin_amdgcn_s_setprio(0);
                if (g == 0) stamp(q, 2);
                const float s = doppler_row<false, TAPS, true>(x, wbuf, s_twn, taps, l, gate, false, nodump, row_tw);
                if (g == 0) stamp(q, 3);
                if ((q & 1) == 0) s_hh[g] = s;
                else if (l == 0) {
                    const int sec = trank + (q >> 1) * teams;
                    unsigned *fr = frames ? frames + (size_t)sec * 2 * (1 + gates) : nullptr;
                    reflectivity_store(&out[((size_t)sec * gates + gate) * 2], gate, s_hh[g], s, k_rr, k_cal, fr, gates, fr ? frame_hdrs[sec] : 0u);
                }
            }
            if (!there) break;
        }
        flush_stamps();
        fused_leave(ctl, host_status, xcc, s_ctl);
    }
}

} // namespace wrp
