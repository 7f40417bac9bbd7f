// wrp_engine.hip -- host side of libwrp.so: the C ABI of include/wrp.h on top of the gfx950 kernels of
// wrp_fused.h (one persistent launch per batch, the default), wrp_kernels.h / wrp_shape_b.h (range pass +
// Doppler pass) and wrp_generic.h (any other power-of-two shape).  Mirrors rpv2.cu's generate_constants / prepare_arys /
// initialize_streams / copy_matrix_to_device / perform_stage_1..3 / copy_result_to_host
// (rpv2.cu:283-618) without its per-launch cudaDeviceSynchronize (rpv2.cu:422-569).
#include <hip/hip_runtime.h>

#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "../../include/wrp.h"
#include "wrp_kernels.h"
#include "wrp_generic.h"
#include "wrp_fused.h"
#include "wrp_shape_b.h"
#include "wrp_fused_b.h"

#define WRP_VERSION_STRING "wrp-amd 0.1 (gfx950)"

namespace {

struct Slot {
    hipStream_t stream = nullptr;
    float2 *h_iq = nullptr;   // pinned [C][m][n]
    float2 *d_iq = nullptr;   // device [C][m][n]
    float2 *d_mid = nullptr;  // device [2][m/2][n]
    float *d_out = nullptr;   // device [m/2][2]
    unsigned *d_frames = nullptr;   // device [2][1 + m/2]: the two products framed for the wire (N2)
    unsigned char *h_raw = nullptr;   // pinned [m*n][wire_bytes] wire bytes (allocated on first use)
    unsigned char *d_raw = nullptr;
    hipEvent_t done = nullptr;
    bool busy = false;
    bool loaded = false;      // d_iq holds an uploaded sector
};

constexpr int WRP_RING = 64;            // fused batches that may be outstanding (status words, events)
constexpr int WRP_FUSED_COOLDOWN = 16;  // batches on the two-kernel path after a fused launch that gave up
constexpr int WRP_DECODE_MIN = 8;       // smallest decode workspace (sectors): it grows on demand up to max_batch sectors
constexpr int WRP_WAIT_SPIN_US = 500;   // wrp_wait polls the slot's event this long before it blocks
constexpr int WRP_GATED_ROWS = 2;       // workgroup rows (grid.y) of a GATED decode / Doppler launch: it walks its sectors with them
constexpr int WRP_GATED_DECODE = 128;   // sectors a gated repeat decodes at a time: it never grows the decode workspace beyond this

struct FusedLane {
    wrp::FusedCtl *d_ctl = nullptr;
    float2 *d_pool = nullptr;           // per XCD team: ONE hand-over slot of 1 MiB
    hipEvent_t done = nullptr;          // the lane's last launch: the next one on this lane waits for it (workspace)
    bool used = false;
    bool ctl_dirty = true;              // memset before the next launch (first launch, after a failure)
};
// the optional wire-ready output of a batch: frames [S][2][1 + m/2] words, hdrs [S] header words (both device)
struct Frames { unsigned *frames = nullptr; const unsigned *hdrs = nullptr; };
// raw: `in` is the wire format.  gated: the two-kernel repeat is already queued on the batch's stream behind the launch,
// gated on the launch's status word in device memory -- the host only takes note of a failure
struct FusedBatch { const float2 *in; int n; float *out; int slot; bool raw; bool gated; Frames fr; };

} // namespace

struct wrp_engine {
    wrp_config cfg;
    int device = 0;
    std::string hip_err;
    // constants (device)
    float *d_wr = nullptr;    // [m]  range window * c
    float *d_wd = nullptr;    // [n]
    float2 *d_tw_m = nullptr; // [m]  exp(-2 pi i k / m)
    float2 *d_tw_n = nullptr; // [n]  exp(+2 pi i k / n)
    float2 *d_tw_n_arr = nullptr;   // n = 512: the same values in the per-lane arrangement of doppler_twiddles_to_lds
    wrp::MaTaps taps;
    int taps_pad = 7;
    bool tuned = true;        // m = 1024, n = 512: tuned kernels; otherwise wrp_generic.h
    int wire_bytes = 12;      // bytes per sample of the raw entries: 12 (sector.cpp:52-62), or 8 with WRP_FLAG_WIRE_8 (VH dropped by the feeder)
    bool tuned_b = false;     // m = 2048, n = 128 (BASELINE configs[4]): wrp_shape_b.h; its stage dumps come from wrp_generic.h
    bool persist = false;     // range pass as a fixed grid walking the tiles with prefetch
    int range_tcols = 16;     // column tile of the range pass (tuning: cfg.flags & 0xff)
    // fused persistent launch (wrp_fused.h): batches of >= WRP_FUSED_MIN_SECTORS sectors, ONE in flight per handle (control
    // block and hand-over slots are the handle's; a launch needs every CU, so two could only collide).  Measured: a second
    // workspace and stream, with the next launch held at a gate kernel until the running one is resident, bought 1 % (the
    // dispatcher deals workgroups to the XCDs in order, so the next launch moves in only when the slowest team has left)
    // and depended on how streams map to hardware queues -- not kept.
    bool fused = false;             // the shape and the flags allow the fused launch
    bool fused_armed = false;       // ... and it is in use (false for WRP_FUSED_COOLDOWN batches after one that gave up)
    int fused_cooldown = 0;
    int fused_fallbacks = 0;        // batches that were repeated on the two-kernel path
    int fused_launches = 0;         // fused launches issued (introspection: wrp_fused_launches)
    int fused_cascaded = 0;         // ... of which: launches queued behind a failed one that failed on its sticky status
    int n_cus = 0;
    FusedLane lane;
    hipEvent_t ev_ring[WRP_RING] = {};   // completion of fused batch `slot`
    unsigned *h_status = nullptr;   // pinned + mapped: word `slot` is written by the fused launch itself, only when it failed
    unsigned *d_status = nullptr;   // the same words as the device sees them
    int ring_next = 0;
    std::deque<FusedBatch> outstanding;   // fused batches whose status word has not been looked at yet, oldest first
    // two-kernel workspace; one such batch in flight per handle: the next one's stream waits for ev_batch
    hipStream_t stream = nullptr;
    hipEvent_t ev_batch = nullptr;
    bool batch_pending = false;
    float2 *d_mid = nullptr;  // [max_batch][2][m/2][n]
    int max_batch = 0;
    float2 *d_decode = nullptr;   // [decode_cap][C][m][n]: wire-format batches in front of kernels that read the planar block (allocated on first use)
    int decode_cap = 0;
    // slots + host result table [elev][sector][gate][2]
    std::vector<Slot> slots;
    float *h_result = nullptr; // pinned
    unsigned *h_frames = nullptr;   // pinned [elev][sector][2][1 + m/2]: header word + big-endian floats, as the GPU wrote them
    // The slot path's Doppler pass writes its 12 KiB of products STRAIGHT into the two pinned tables (their device views):
    // no D2H copies.  Two API calls fewer per sector on the feeder thread, and nothing of a slot's chain queues on the copy
    // engine behind the other slots' H2D transfers any more (a 4 KiB D2H behind three 75 us transfers made the C++ feeder
    // 12 us per sector slower than the link: profiles/r05/feeder_breakdown.log)
    float *d_result_tab = nullptr;
    unsigned *d_frames_tab = nullptr;
    // dump scratch
    void *d_dump = nullptr;
    size_t dump_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

#define HIP_TRY(h, expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (h)->hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            return (e_ == hipErrorOutOfMemory) ? WRP_ERR_NOMEM : WRP_ERR_HIP;              \
        }                                                                                  \
    } while (0)

bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
int ilog2(int x) { int b = 0; while ((1 << b) < x) b++; return b; }
// m = 1024, n = 512 (the 00iq.altb shape) and m = 2048, n = 128 (BASELINE configs[4]) run tuned kernels (wrp_fused.h,
// wrp_kernels.h / wrp_fused_b.h, wrp_shape_b.h); every other power-of-two shape up to 2048 x 1024 runs wrp_generic.h.
bool shape_supported(int m, int n) { return is_pow2(m) && is_pow2(n) && m >= 64 && m <= 2048 && n >= 32 && n <= 1024; }
bool shape_tuned(int m, int n) { return m == 1024 && n == 512; }

// rpv2.cu:222-250 generate_hamming_coefficients, kept separable: W[i][j] = wr_c[i] * wd[j]
void make_window(int m, int n, std::vector<float> &wr_c, std::vector<float> &wd)
{
    double p_range = 0, p_doppler = 0;
    for (int i = 0; i < m; i++) p_range += std::pow(0.53836 - 0.46164 * std::cos(2 * M_PI * i / (m - 1)), 2.0);
    p_range /= m;
    for (int j = 0; j < n; j++) p_doppler += std::pow(0.53836 - 0.46164 * std::cos(2 * M_PI * j / (n - 1)), 2.0);
    p_doppler /= n;
    const double k_wind = -1 / (16383.5 * m * n * std::sqrt(50.0));
    const double c = k_wind / std::sqrt(p_range * p_doppler);
    wr_c.resize(m);
    wd.resize(n);
    for (int i = 0; i < m; i++) wr_c[i] = (float)((0.53836 - 0.46164 * std::cos(2 * M_PI * i / (m - 1))) * c);
    for (int j = 0; j < n; j++) wd[j] = (float)(0.53836 - 0.46164 * std::cos(2 * M_PI * j / (n - 1)));
}

// rpv2.cu:252-262 generate_ma_coefficients (taps only; the FFT of the taps is not needed
// because the convolution is evaluated directly)
void make_taps(int count, wrp::MaTaps &t)
{
    double g[9], sum = 0;
    for (int i = 0; i < count; i++) {
        g[i] = std::exp(-(std::pow(i - ((count - 1) / 2), 2.0)) / 2);
        sum += g[i];
    }
    for (int i = 0; i < 9; i++) t.g[i] = i < count ? (float)(g[i] / sum) : 0.f;
    double s = 0;       // what the rows multiply their sum of |.|^2 by: the sum of the taps AS ROUNDED (doppler_row: a7 + a8)
    for (int i = 0; i < count; i++) s += (double)t.g[i];
    t.sum = (float)s;
}

template <int TCOLS, bool DUMP>
void launch_range_t(wrp_engine *h, const float2 *d_iq, int n_sectors, float2 *d_mid, hipStream_t st,
                    const wrp::DumpPtrs &d, const unsigned *gate)
{
    typedef wrp::RangeTile<TCOLS> T;
    const wrp_config &c = h->cfg;
    const dim3 grid(n_sectors * 2 * (c.n / TCOLS)), block(T::THREADS);
    const wrp::RangeConsts rc{h->d_wr, h->d_wd, h->d_tw_m};
    hipLaunchKernelGGL((wrp::range_pass_1024<TCOLS, DUMP>), grid, block, T::LDS_BYTES, st, d_iq, d_mid, rc, c.n,
                       c.channels, d, gate);
}

// gate (device pointer or nullptr): the launch only runs when *gate != 0 (kernels: gate_closed)
void launch_range(wrp_engine *h, const float2 *d_iq, int n_sectors, float2 *d_mid, hipStream_t st,
                  const wrp::DumpPtrs *dump, const unsigned *gate = nullptr)
{
    wrp::DumpPtrs none{};
    none.channel = -1;
    // shape B: the tuned kernel, except for the two dumps that need all m rows (it only ever forms the gates < m/2)
    if (h->tuned_b && !(dump && (dump->hamm || dump->fft1))) {
        const wrp::RangeConsts rc{h->d_wr, h->d_wd, h->d_tw_m};
        const int total = n_sectors * 2 * (wrp::RB_N / 16), grid = std::min(total, h->n_cus);
        hipLaunchKernelGGL(wrp::range_pass_2048, dim3(grid), dim3(wrp::RangeTileB::THREADS), wrp::RangeTileB::LDS_BYTES, st, d_iq,
                           d_mid, rc, h->cfg.channels, total, gate);
        return;
    }
    if (!h->tuned) {
        const wrp_config &c = h->cfg;
        const wrp::RangeConsts rc{h->d_wr, h->d_wd, h->d_tw_m};
        const dim3 grid(n_sectors * 2 * (c.n / wrp::GEN_TC)), block(wrp::GEN_THREADS);
        const size_t lds = (size_t)c.m * wrp::GEN_TC * sizeof(float2);
        if (dump)
            hipLaunchKernelGGL(wrp::generic_range_pass<true>, grid, block, lds, st, d_iq, d_mid, rc, c.m, ilog2(c.m),
                               c.n, c.channels, *dump);
        else
            hipLaunchKernelGGL(wrp::generic_range_pass<false>, grid, block, lds, st, d_iq, d_mid, rc, c.m, ilog2(c.m),
                               c.n, c.channels, none);
        return;
    }
    if (h->persist && !dump) {
        // fixed grid, tiles walked with prefetch; a multiple of 16 blocks keeps the XCD pairing
        const wrp_config &c = h->cfg;
        const wrp::RangeConsts rc{h->d_wr, h->d_wd, h->d_tw_m};
        if (h->range_tcols == 8) {
            typedef wrp::RangeTile<8> T;
            const int total = n_sectors * 2 * (c.n / 8), grid = std::min(total, (2 * h->n_cus) & ~15);
            hipLaunchKernelGGL(wrp::range_pass_1024_persistent<8>, dim3(grid), dim3(T::THREADS), T::LDS_BYTES, st, d_iq,
                               d_mid, rc, c.n, c.channels, total, gate);
        } else {
            typedef wrp::RangeTile<16> T;
            const int total = n_sectors * 2 * (c.n / 16), grid = std::min(total, h->n_cus & ~15);
            hipLaunchKernelGGL(wrp::range_pass_1024_persistent<16>, dim3(grid), dim3(T::THREADS), T::LDS_BYTES, st, d_iq,
                               d_mid, rc, c.n, c.channels, total, gate);
        }
        return;
    }
    if (h->range_tcols == 8) {
        if (dump) launch_range_t<8, true>(h, d_iq, n_sectors, d_mid, st, *dump, gate);
        else launch_range_t<8, false>(h, d_iq, n_sectors, d_mid, st, none, gate);
    } else {
        if (dump) launch_range_t<16, true>(h, d_iq, n_sectors, d_mid, st, *dump, gate);
        else launch_range_t<16, false>(h, d_iq, n_sectors, d_mid, st, none, gate);
    }
}

template <bool DUMP, int TAPS>
void launch_doppler_t(wrp_engine *h, const float2 *d_mid, int n_sectors, float *d_out, hipStream_t st,
                      const wrp::DumpPtrs &d, const unsigned *gate)
{
    const wrp_config &c = h->cfg;
    // a gated launch almost never has anything to do: a few rows of workgroups that would walk the sectors
    const dim3 grid(c.m / 2 / wrp::DP_WAVES, gate ? std::min(n_sectors, WRP_GATED_ROWS) : n_sectors), block(wrp::DP_WAVES * 64);
    hipLaunchKernelGGL((wrp::doppler_pass_512<DUMP, TAPS>), grid, block, 0, st, d_mid, d_out, h->d_tw_n_arr,
                       c.m / 2, n_sectors, h->taps, c.k_range_resolution, c.k_calibration, d, gate);
}

// frames + frame_hdr: the slot path (sector 0 of the launch); fr: the batch entries (every sector, header table)
void launch_doppler(wrp_engine *h, const float2 *d_mid, int n_sectors, float *d_out, hipStream_t st,
                    const wrp::DumpPtrs *dump, unsigned *frames = nullptr, unsigned frame_hdr = 0, Frames fr = Frames{},
                    const unsigned *gate = nullptr)
{
    wrp::DumpPtrs none{};
    none.channel = -1;
    none.frames = fr.frames ? fr.frames : frames;        // (the dump launches never frame: wrp_dump_stage)
    none.frame_hdr = frame_hdr;
    none.frame_hdrs = fr.frames ? fr.hdrs : nullptr;
    if (h->tuned_b && !(dump && (dump->hamm || dump->fft1))) {
        const wrp_config &c = h->cfg;
        const dim3 grid(c.m / 2 / (wrp::DB_WAVES * 2), gate ? std::min(n_sectors, WRP_GATED_ROWS) : n_sectors), block(wrp::DB_WAVES * 64);
#define WRP_DOPPLER_B(TAPS, DUMP)                                                                                    \
    hipLaunchKernelGGL((wrp::doppler_pass_128<TAPS, DUMP>), grid, block, 0, st, d_mid, d_out, h->d_tw_n, c.m / 2, n_sectors, h->taps, \
                       c.k_range_resolution, c.k_calibration, dump ? *dump : none, gate)
        if (h->taps_pad == 7) { if (dump) WRP_DOPPLER_B(7, true); else WRP_DOPPLER_B(7, false); }
        else { if (dump) WRP_DOPPLER_B(9, true); else WRP_DOPPLER_B(9, false); }
#undef WRP_DOPPLER_B
        return;
    }
    if (!h->tuned) {
        const wrp_config &c = h->cfg;
        const dim3 grid(c.m / 2, n_sectors), block(64);
        const size_t lds = (size_t)c.n * 12;
        const wrp::DumpPtrs &d = dump ? *dump : none;
#define WRP_GEN_DOPPLER(DUMP, TAPS)                                                                           \
    hipLaunchKernelGGL((wrp::generic_doppler_pass<DUMP, TAPS>), grid, block, lds, st, d_mid, d_out, h->d_tw_n, \
                       c.m / 2, c.n, ilog2(c.n), h->taps, c.k_range_resolution, c.k_calibration, d)
        if (h->taps_pad == 7) { if (dump) WRP_GEN_DOPPLER(true, 7); else WRP_GEN_DOPPLER(false, 7); }
        else { if (dump) WRP_GEN_DOPPLER(true, 9); else WRP_GEN_DOPPLER(false, 9); }
#undef WRP_GEN_DOPPLER
        return;
    }
    if (h->taps_pad == 7) {
        if (dump) launch_doppler_t<true, 7>(h, d_mid, n_sectors, d_out, st, *dump, gate);
        else launch_doppler_t<false, 7>(h, d_mid, n_sectors, d_out, st, none, gate);
    } else {
        if (dump) launch_doppler_t<true, 9>(h, d_mid, n_sectors, d_out, st, *dump, gate);
        else launch_doppler_t<false, 9>(h, d_mid, n_sectors, d_out, st, none, gate);
    }
}

size_t sector_elems(const wrp_config &c) { return (size_t)c.channels * c.m * c.n; }
size_t mid_elems(const wrp_config &c) { return (size_t)2 * (c.m / 2) * c.n; }
size_t frame_words(const wrp_config &c) { return (size_t)2 * (1 + c.m / 2); }   // the two framed products of one sector
Frames frames_at(Frames fr, const wrp_config &c, int sector)
{
    if (fr.frames) { fr.frames += (size_t)sector * frame_words(c); fr.hdrs += sector; }
    return fr;
}

// one persistent launch for the whole batch: XCD teams keep the intermediate in their L2 (wrp_fused.h)
// d_stamps / d_tee: the diagnostics instantiations (phase stamps; a copy of the whole intermediate in global memory)
int launch_fused(wrp_engine *h, FusedLane &lane, const float2 *d_iq, int n_sectors, float *d_out, hipStream_t st, int slot,
                 unsigned long long *d_stamps = nullptr, bool raw = false, Frames fr = Frames{}, float2 *d_tee = nullptr)
{
    const wrp_config &c = h->cfg;
    // a successful launch leaves the control block zeroed (fused_leave): no memset node in front of the next one
    if (lane.ctl_dirty) {
        HIP_TRY(h, hipMemsetAsync(lane.d_ctl, 0, sizeof(wrp::FusedCtl), st));
        lane.ctl_dirty = false;
    }
    const wrp::RangeConsts rc{h->d_wr, h->d_wd, h->d_tw_m};
    // two workgroups per CU; the test flag launches one per CU, so that no team gets its row members
    const int grid = (c.flags & WRP_FLAG_DEBUG_FUSED_UNDERSIZED) ? h->n_cus : h->n_cus * 2;
    const int form = d_tee ? 2 : d_stamps ? 1 : 0;      // which instantiation
    const int wb = raw ? h->wire_bytes : 0;             // 0: the planar block; 12 / 8: bytes per wire sample
    const bool t7 = h->taps_pad == 7;
    if (h->tuned_b) {
#define WRP_FUSED_B(TAPS, STAMPS, TEE, RAW)                                                                           \
    hipLaunchKernelGGL((wrp::fused_chain_2048x128<TAPS, STAMPS, TEE, RAW>), dim3(grid), dim3(wrp::FUSED_THREADS),     \
                       wrp::FusedTileB::LDS_BYTES, st, d_iq, d_out, lane.d_pool, lane.d_ctl, rc, h->d_tw_n, n_sectors, \
                       c.channels, h->taps, c.k_range_resolution, c.k_calibration, h->d_status + slot, d_stamps,      \
                       fr.frames, fr.hdrs, d_tee)
#define WRP_FUSED_B_FORMS(RAW)                                                                                        \
    do {                                                                                                              \
        if (form == 2) { if (t7) WRP_FUSED_B(7, false, true, RAW); else WRP_FUSED_B(9, false, true, RAW); }           \
        else { if (t7) WRP_FUSED_B(7, false, false, RAW); else WRP_FUSED_B(9, false, false, RAW); }                   \
    } while (0)
        if (wb == 12) WRP_FUSED_B_FORMS(12);      // the wire format straight into the tile workgroups
        else if (wb == 8) WRP_FUSED_B_FORMS(8);
        else if (form == 1) { if (t7) WRP_FUSED_B(7, true, false, 0); else WRP_FUSED_B(9, true, false, 0); }
        else WRP_FUSED_B_FORMS(0);
#undef WRP_FUSED_B_FORMS
#undef WRP_FUSED_B
        HIP_TRY(h, hipGetLastError());
        return WRP_OK;
    }
#define WRP_FUSED(TAPS, STAMPS, RAW, TEE)                                                                             \
    hipLaunchKernelGGL((wrp::fused_chain_1024x512<TAPS, STAMPS, RAW, TEE>), dim3(grid), dim3(wrp::FUSED_THREADS),     \
                       wrp::FusedTile::LDS_BYTES, st, d_iq, d_out, lane.d_pool, lane.d_ctl, rc, h->d_tw_n_arr, n_sectors, \
                       c.channels, h->taps, c.k_range_resolution, c.k_calibration, h->d_status + slot, d_stamps,      \
                       fr.frames, fr.hdrs, d_tee)
#define WRP_FUSED_FORMS(RAW)                                                                                          \
    do {                                                                                                              \
        if (form == 2) { if (t7) WRP_FUSED(7, false, RAW, true); else WRP_FUSED(9, false, RAW, true); }               \
        else { if (t7) WRP_FUSED(7, false, RAW, false); else WRP_FUSED(9, false, RAW, false); }                       \
    } while (0)
    if (wb == 12) WRP_FUSED_FORMS(12);            // the wire format straight into the tile workgroups
    else if (wb == 8) WRP_FUSED_FORMS(8);
    else if (form == 1) { if (t7) WRP_FUSED(7, true, 0, false); else WRP_FUSED(9, true, 0, false); }
    else WRP_FUSED_FORMS(0);
#undef WRP_FUSED_FORMS
#undef WRP_FUSED
    HIP_TRY(h, hipGetLastError());
    return WRP_OK;
}

// range pass + Doppler pass.  frames / frame_hdr: the slot path's framed products (sector 0 of the launch); fr: a batch's;
// gate: see launch_range
int launch_chain(wrp_engine *h, const float2 *d_iq, int n_sectors, float2 *d_mid, float *d_out, hipStream_t st,
                 const wrp::DumpPtrs *dump, unsigned *frames = nullptr, unsigned frame_hdr = 0, Frames fr = Frames{},
                 const unsigned *gate = nullptr)
{
    launch_range(h, d_iq, n_sectors, d_mid, st, dump, gate);
    launch_doppler(h, d_mid, n_sectors, d_out, st, dump, frames, frame_hdr, fr, gate);
    HIP_TRY(h, hipGetLastError());
    return WRP_OK;
}

// The workspaces every batch path shares -- d_mid (two kernels) and d_decode (wire-format batches in front of kernels that
// read the planar block) -- are handed from batch to batch through ONE event: a batch that uses either first makes its
// stream wait for the previous user, and leaves the event behind it.
int workspace_acquire(wrp_engine *h, hipStream_t st)
{
    if (h->batch_pending) HIP_TRY(h, hipStreamWaitEvent(st, h->ev_batch, 0));
    return WRP_OK;
}
int workspace_release(wrp_engine *h, hipStream_t st)
{
    HIP_TRY(h, hipEventRecord(h->ev_batch, st));
    h->batch_pending = true;
    return WRP_OK;
}

// the two-kernel path over a whole batch, in chunks of max_batch sectors, on stream st
int launch_two_kernel_batch(wrp_engine *h, const float2 *in, int n_sectors, float *d_out, hipStream_t st, Frames fr = Frames{},
                            const unsigned *gate = nullptr)
{
    const wrp_config &c = h->cfg;
    int rc = workspace_acquire(h, st);
    for (int s0 = 0; rc == WRP_OK && s0 < n_sectors; s0 += h->max_batch) {
        const int cnt = std::min(h->max_batch, n_sectors - s0);
        rc = launch_chain(h, in + (size_t)s0 * sector_elems(c), cnt, h->d_mid, d_out + (size_t)s0 * (c.m / 2) * 2, st, nullptr,
                          nullptr, 0, frames_at(fr, c, s0), gate);
    }
    return rc != WRP_OK ? rc : workspace_release(h, st);
}

// the decode workspace holds at least `sectors` sectors (at most `limit`, itself at most max_batch, are ever asked for);
// growing it waits for the device
int ensure_decode(wrp_engine *h, int sectors, int limit = 1 << 30)
{
    limit = std::min(limit, h->max_batch);
    sectors = std::min(limit, std::max(sectors, std::min(limit, WRP_DECODE_MIN)));
    if (h->decode_cap >= sectors) return WRP_OK;
    if (h->d_decode) {
        HIP_TRY(h, hipDeviceSynchronize());
        (void)hipFree(h->d_decode);
        h->d_decode = nullptr;
        h->decode_cap = 0;
    }
    HIP_TRY(h, hipMalloc(&h->d_decode, sizeof(float2) * sector_elems(h->cfg) * (size_t)sectors));
    h->decode_cap = sectors;
    return WRP_OK;
}
// cnt sectors of wire bytes -> planar blocks at `dst`
void launch_decode(wrp_engine *h, const unsigned char *raw, float2 *dst, int cnt, hipStream_t st, const unsigned *gate = nullptr)
{
    const int count = h->cfg.m * h->cfg.n;
    const dim3 grid((count + 255) / 256, gate ? std::min(cnt, WRP_GATED_ROWS) : cnt), block(256);
    if (h->wire_bytes == 8)
        hipLaunchKernelGGL(wrp::decode_wire<8>, grid, block, 0, st, (const unsigned *)raw, dst, count, h->cfg.channels, cnt, gate);
    else
        hipLaunchKernelGGL(wrp::decode_wire<12>, grid, block, 0, st, (const unsigned *)raw, dst, count, h->cfg.channels, cnt, gate);
}

// a wire-format batch on the two-kernel path: decode_wire + the two kernels, as many sectors at a time as both workspaces hold
int launch_two_kernel_raw_batch(wrp_engine *h, const unsigned char *raw, int n_sectors, float *d_out, hipStream_t st,
                                Frames fr = Frames{}, const unsigned *gate = nullptr)
{
    const wrp_config &c = h->cfg;
    const size_t count = (size_t)c.m * c.n;
    // a gated repeat works through whatever the workspace holds (submit_fused_piece has made sure of WRP_GATED_DECODE sectors
    // BEFORE its launch): nothing here can fail or wait between a fused launch and its repeat
    int rc = gate ? (h->decode_cap > 0 ? WRP_OK : WRP_ERR_STATE) : ensure_decode(h, n_sectors);
    if (rc == WRP_OK) rc = workspace_acquire(h, st);
    for (int s0 = 0; rc == WRP_OK && s0 < n_sectors; s0 += h->decode_cap) {
        const int cnt = std::min(h->decode_cap, n_sectors - s0);    // decode_cap <= max_batch: d_mid holds them too
        launch_decode(h, raw + (size_t)s0 * count * h->wire_bytes, h->d_decode, cnt, st, gate);
        rc = launch_chain(h, h->d_decode, cnt, h->d_mid, d_out + (size_t)s0 * (c.m / 2) * 2, st, nullptr, nullptr, 0,
                          frames_at(fr, c, s0), gate);
    }
    return rc != WRP_OK ? rc : workspace_release(h, st);
}

// A fused launch that gave up (bounded wait, or the CUs did not host 32 tile + 32 row workgroups per XCD: another kernel
// on the GPU) has said so in its own pinned status word.  Its output is void.  The handle leaves the fused launch alone
// for WRP_FUSED_COOLDOWN batches and then tries again.  The FIRST failure of a run leaves its note; launches that were
// queued behind it fail on its sticky status and are only counted.
void note_fused_failure(wrp_engine *h, unsigned status)
{
    const bool cascade = !h->fused_armed && h->fused_cooldown > 0;
    h->fused_armed = false;
    h->fused_cooldown = WRP_FUSED_COOLDOWN;
    h->fused_fallbacks++;
    h->lane.ctl_dirty = true;
    if (cascade) {
        h->fused_cascaded++;
        return;
    }
    h->hip_err = (status & 2)
        ? "fused launch: an XCD did not host 32 tile + 32 row workgroups; batch repeated on the two-kernel path"
        : "fused launch: a bounded wait gave up (workgroups not co-resident?); batch repeated on the two-kernel path";
}
// ... and the batch is computed again by the two kernels: here and synchronously (batches on the engine's own stream), or
// by the gated launches that are already queued behind it on the caller's stream (submit_fused)
int redo_batch(wrp_engine *h, const FusedBatch &b, unsigned status)
{
    note_fused_failure(h, status);
    if (b.gated) return WRP_OK;
    const std::string note = h->hip_err;
    int rc = b.raw ? launch_two_kernel_raw_batch(h, (const unsigned char *)b.in, b.n, b.out, h->stream, b.fr)
                   : launch_two_kernel_batch(h, b.in, b.n, b.out, h->stream, b.fr);
    if (rc != WRP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->batch_pending = false;
    h->hip_err = note;
    return WRP_OK;
}

// Look at the status words of the fused batches that have completed (block = false) or of all of them, waiting for each
// (block = true); repeat the ones that gave up.  Called from every entry that touches the batch path.
int reap_fused(wrp_engine *h, bool block, size_t leave = 0)
{
    while (h->outstanding.size() > leave) {
        const FusedBatch b = h->outstanding.front();
        if (block) {
            HIP_TRY(h, hipEventSynchronize(h->ev_ring[b.slot]));
        } else {
            const hipError_t q = hipEventQuery(h->ev_ring[b.slot]);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
            HIP_TRY(h, q);
        }
        h->outstanding.pop_front();
        const unsigned st = h->h_status[b.slot];
        h->h_status[b.slot] = 0;
        if (st) {
            const int rc = redo_batch(h, b, st);
            if (rc != WRP_OK) return rc;
        }
    }
    return WRP_OK;
}

// fused launch of (a piece of) one batch; stream = the caller's or nullptr (the engine's).
// On a CALLER's stream the two-kernel repeat is queued right behind the launch, gated on the launch's status word in
// device memory (its workgroups return at once when the launch has succeeded): d_out is right when the stream says so,
// whether or not the host ever looks (stream order alone; include/wrp.h).  On the engine's own stream nothing but
// wrp_check can wait for the batch, and wrp_check repeats a failed launch itself: no gated launches there.
int two_kernel_batch(wrp_engine *h, const void *in, bool raw, int n_sectors, float *d_out, hipStream_t st, Frames fr);

int submit_fused_piece(wrp_engine *h, const float2 *in, int n_sectors, float *d_out, hipStream_t stream, bool raw, Frames fr)
{
    hipStream_t st = stream ? stream : h->stream;
    if (h->outstanding.size() >= (size_t)WRP_RING) {
        const int rc = reap_fused(h, true, WRP_RING - 1);
        if (rc != WRP_OK) return rc;
        if (!h->fused_armed) return two_kernel_batch(h, in, raw, n_sectors, d_out, st, fr);   // one of them had given up
    }
    FusedLane &lane = h->lane;
    // Everything that can fail or wait on the host comes BEFORE the launch: the decode workspace of the gated
    // repeat.  Behind the launch there are only stream operations, and the launch is on the books before the first of them.
    if (stream && raw) {
        const int rc = ensure_decode(h, std::min(n_sectors, WRP_GATED_DECODE), WRP_GATED_DECODE);
        if (rc != WRP_OK) return rc;
    }
    if (lane.used) HIP_TRY(h, hipStreamWaitEvent(st, lane.done, 0));   // control block and slots are free again (free on one stream)
    const int slot = h->ring_next;
    int rc = launch_fused(h, lane, in, n_sectors, d_out, st, slot, nullptr, raw, fr);
    if (rc != WRP_OK) return rc;                                       // nothing was queued
    h->ring_next = (h->ring_next + 1) % WRP_RING;
    h->fused_launches++;
    // on the books: whatever fails below, the launch is waited for (reap_fused, wrp_destroy), its status word is read and
    // zeroed, and the next launch on this lane waits for it
    const hipError_t e_ring = hipEventRecord(h->ev_ring[slot], st), e_lane = hipEventRecord(lane.done, st);
    lane.used = true;
    h->outstanding.push_back(FusedBatch{in, n_sectors, d_out, slot, raw, stream != nullptr, fr});
    if (e_ring != hipSuccess || e_lane != hipSuccess) {
        (void)hipStreamSynchronize(st);                                // no event to wait on: wait here
        HIP_TRY(h, e_ring != hipSuccess ? e_ring : e_lane);
    }
    if (stream) {
        const unsigned *gate = &lane.d_ctl->status;
        rc = raw ? launch_two_kernel_raw_batch(h, (const unsigned char *)in, n_sectors, d_out, st, fr, gate)
                 : launch_two_kernel_batch(h, in, n_sectors, d_out, st, fr, gate);
        // the gated launches read the control block: the lane is free behind THEM
        const hipError_t e = hipEventRecord(lane.done, st);
        if (rc != WRP_OK) return rc;
        HIP_TRY(h, e);
    }
    return WRP_OK;
}

// one batch through the fused launch (planar or wire format: both tuned shapes decode in their tile workgroups)
int submit_fused(wrp_engine *h, const float2 *in, int n_sectors, float *d_out, hipStream_t stream, bool raw = false, Frames fr = Frames{})
{
    return submit_fused_piece(h, in, n_sectors, d_out, stream, raw, fr);
}

// [sector BE16][elevation BE16] as one little-endian word of device / host memory
unsigned frame_header_word(int sector, int elevation)
{
    return ((unsigned)(sector >> 8) & 0xffu) | (((unsigned)sector & 0xffu) << 8) | (((unsigned)(elevation >> 8) & 0xffu) << 16) |
           (((unsigned)elevation & 0xffu) << 24);
}
unsigned *frames_of(wrp_engine *h, int sector, int elevation)
{
    return h->h_frames + ((size_t)elevation * h->cfg.n_sectors + sector) * 2 * (1 + h->cfg.m / 2);
}

int destroy_impl(wrp_engine *h)
{
    if (!h) return WRP_OK;
    (void)hipSetDevice(h->device);
    for (const auto &b : h->outstanding) (void)hipEventSynchronize(h->ev_ring[b.slot]);   // also those on caller streams
    h->outstanding.clear();
    for (auto &s : h->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.h_iq) (void)hipHostFree(s.h_iq);
        if (s.h_raw) (void)hipHostFree(s.h_raw);
        if (s.d_raw) (void)hipFree(s.d_raw);
        if (s.d_iq) (void)hipFree(s.d_iq);
        if (s.d_mid) (void)hipFree(s.d_mid);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_frames) (void)hipFree(s.d_frames);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_batch) (void)hipEventDestroy(h->ev_batch);
    if (h->d_mid) (void)hipFree(h->d_mid);
    if (h->d_decode) (void)hipFree(h->d_decode);
    if (h->lane.done) (void)hipEventDestroy(h->lane.done);
    if (h->lane.d_ctl) (void)hipFree(h->lane.d_ctl);
    if (h->lane.d_pool) (void)hipFree(h->lane.d_pool);
    for (auto &e : h->ev_ring) if (e) (void)hipEventDestroy(e);
    if (h->h_status) (void)hipHostFree(h->h_status);
    if (h->d_dump) (void)hipFree(h->d_dump);
    if (h->h_result) (void)hipHostFree(h->h_result);
    if (h->h_frames) (void)hipHostFree(h->h_frames);
    if (h->d_wr) (void)hipFree(h->d_wr);
    if (h->d_wd) (void)hipFree(h->d_wd);
    if (h->d_tw_m) (void)hipFree(h->d_tw_m);
    if (h->d_tw_n) (void)hipFree(h->d_tw_n);
    if (h->d_tw_n_arr) (void)hipFree(h->d_tw_n_arr);
    delete h;
    return WRP_OK;
}

int create_impl(wrp_engine *h)
{
    const wrp_config &c = h->cfg;
    HIP_TRY(h, hipSetDevice(h->device));
    // up to 144 KiB of dynamic LDS for the range pass
    h->range_tcols = (c.flags & 0xff) == 16 ? 16 : 8;   // default chosen below
    h->tuned = shape_tuned(c.m, c.n);
    h->wire_bytes = (c.flags & WRP_FLAG_WIRE_8) ? 8 : 12;
    h->tuned_b = c.m == wrp::RB_M && c.n == wrp::RB_N && (c.flags & WRP_FLAG_GENERIC_KERNELS) == 0;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::range_pass_2048),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::RangeTileB::LDS_BYTES));
    // the fused launch is the default for the tuned shape; WRP_FLAG_TWO_KERNELS keeps the pair of kernels
    h->fused = (h->tuned || h->tuned_b) && (c.flags & WRP_FLAG_TWO_KERNELS) == 0;
#define WRP_FUSED_B_ATTR(TAPS, STAMPS, TEE, RAW)                                                                    \
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::fused_chain_2048x128<TAPS, STAMPS, TEE, RAW>), \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::FusedTileB::LDS_BYTES))
    WRP_FUSED_B_ATTR(7, false, false, 0);  WRP_FUSED_B_ATTR(9, false, false, 0);
    WRP_FUSED_B_ATTR(7, false, false, 12); WRP_FUSED_B_ATTR(9, false, false, 12);     // wire-format input
    WRP_FUSED_B_ATTR(7, false, false, 8);  WRP_FUSED_B_ATTR(9, false, false, 8);      // ... without VH (WRP_FLAG_WIRE_8)
    WRP_FUSED_B_ATTR(7, true, false, 0);   WRP_FUSED_B_ATTR(9, true, false, 0);       // diagnostics: phase stamps
    WRP_FUSED_B_ATTR(7, false, true, 0);   WRP_FUSED_B_ATTR(9, false, true, 0);       // diagnostics: the intermediate copied out
    WRP_FUSED_B_ATTR(7, false, true, 12);  WRP_FUSED_B_ATTR(9, false, true, 12);
    WRP_FUSED_B_ATTR(7, false, true, 8);   WRP_FUSED_B_ATTR(9, false, true, 8);
#undef WRP_FUSED_B_ATTR
    h->fused_armed = h->fused;
    h->persist = h->tuned && (c.flags & WRP_FLAG_ONE_TILE_PER_BLOCK) == 0;
    if ((c.flags & 0xff) == 0) h->range_tcols = h->persist ? 16 : 8;   // best measured tile for each form
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::range_pass_1024_persistent<8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::RangeTile<8>::LDS_BYTES));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::range_pass_1024_persistent<16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::RangeTile<16>::LDS_BYTES));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::generic_range_pass<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * wrp::GEN_TC * 8));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::generic_range_pass<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * wrp::GEN_TC * 8));
    {
        hipDeviceProp_t prop;
        HIP_TRY(h, hipGetDeviceProperties(&prop, h->device));
        h->n_cus = prop.multiProcessorCount;
    }
#define WRP_FUSED_ATTR(TAPS, STAMPS, RAW, TEE)                                                                      \
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::fused_chain_1024x512<TAPS, STAMPS, RAW, TEE>), \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::FusedTile::LDS_BYTES))
    WRP_FUSED_ATTR(7, false, 0, false);  WRP_FUSED_ATTR(9, false, 0, false);      // the launch
    WRP_FUSED_ATTR(7, false, 12, false); WRP_FUSED_ATTR(9, false, 12, false);     // wire-format input
    WRP_FUSED_ATTR(7, false, 8, false);  WRP_FUSED_ATTR(9, false, 8, false);      // ... without VH (WRP_FLAG_WIRE_8)
    WRP_FUSED_ATTR(7, true, 0, false);   WRP_FUSED_ATTR(9, true, 0, false);       // diagnostics: phase stamps
    WRP_FUSED_ATTR(7, false, 0, true);   WRP_FUSED_ATTR(9, false, 0, true);       // diagnostics: the intermediate copied out
    WRP_FUSED_ATTR(7, false, 12, true);  WRP_FUSED_ATTR(9, false, 12, true);
    WRP_FUSED_ATTR(7, false, 8, true);   WRP_FUSED_ATTR(9, false, 8, true);
#undef WRP_FUSED_ATTR
    HIP_TRY(h, hipEventCreateWithFlags(&h->lane.done, hipEventDisableTiming));
    HIP_TRY(h, hipMalloc(&h->lane.d_ctl, sizeof(wrp::FusedCtl)));
    HIP_TRY(h, hipMalloc(&h->lane.d_pool, sizeof(float2) * wrp::FUSED_TEAM_ELEMS * wrp::FUSED_MAX_TEAMS));
    for (auto &e : h->ev_ring) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(h, hipHostMalloc(&h->h_status, sizeof(unsigned) * WRP_RING, hipHostMallocMapped));
    std::memset(h->h_status, 0, sizeof(unsigned) * WRP_RING);
    HIP_TRY(h, hipHostGetDevicePointer((void **)&h->d_status, h->h_status, 0));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::range_pass_1024<16, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::RangeTile<16>::LDS_BYTES));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::range_pass_1024<16, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::RangeTile<16>::LDS_BYTES));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::range_pass_1024<8, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::RangeTile<8>::LDS_BYTES));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&wrp::range_pass_1024<8, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, wrp::RangeTile<8>::LDS_BYTES));

    std::vector<float> wr, wd;
    make_window(c.m, c.n, wr, wd);
    // The Hamming window is symmetric, w(i) = w(m - 1 - i); the fused launch of shape B keeps only its first half in LDS.
    // As computed (double, rounded to float) the two halves may differ in the last bit of a few entries: the second half
    // is MADE the mirror of the first, for every kernel of this shape, so that they all see the same values.
    if (h->tuned_b)
        for (int i = c.m / 2; i < c.m; i++) wr[i] = wr[c.m - 1 - i];
    std::vector<float2> twm(c.m), twn(c.n);
    for (int k = 0; k < c.m; k++) twm[k] = make_float2((float)std::cos(2 * M_PI * k / c.m), (float)-std::sin(2 * M_PI * k / c.m));
    for (int k = 0; k < c.n; k++) twn[k] = make_float2((float)std::cos(2 * M_PI * k / c.n), (float)std::sin(2 * M_PI * k / c.n));
    make_taps(c.ma_count, h->taps);
    h->taps_pad = c.ma_count <= 7 ? 7 : 9;

    HIP_TRY(h, hipMalloc(&h->d_wr, sizeof(float) * c.m));
    HIP_TRY(h, hipMalloc(&h->d_wd, sizeof(float) * c.n));
    HIP_TRY(h, hipMalloc(&h->d_tw_m, sizeof(float2) * c.m));
    HIP_TRY(h, hipMalloc(&h->d_tw_n, sizeof(float2) * c.n));
    HIP_TRY(h, hipMemcpy(h->d_wr, wr.data(), sizeof(float) * c.m, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_wd, wd.data(), sizeof(float) * c.n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_tw_m, twm.data(), sizeof(float2) * c.m, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_tw_n, twn.data(), sizeof(float2) * c.n, hipMemcpyHostToDevice));
    if (h->tuned) {
        std::vector<float2> arr(wrp::DP_TW_ELEMS);
        for (int e = 0; e < wrp::DP_TW_ELEMS; e++) arr[e] = twn[wrp::doppler_twiddle_index(e)];
        HIP_TRY(h, hipMalloc(&h->d_tw_n_arr, sizeof(float2) * arr.size()));
        HIP_TRY(h, hipMemcpy(h->d_tw_n_arr, arr.data(), sizeof(float2) * arr.size(), hipMemcpyHostToDevice));
    }

    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreate(&h->ev0));
    HIP_TRY(h, hipEventCreate(&h->ev1));
    // 360 sectors (one elevation sweep) per launch pair: large grids amortise launch tails
    // (measured 4.4 -> 3.8 -> 3.6 us/sector at 24 / 120 / 360); the 1.4 GiB workspace is 0.5 % of HBM
    h->max_batch = c.max_batch > 0 ? c.max_batch : 360;
    HIP_TRY(h, hipMalloc(&h->d_mid, sizeof(float2) * mid_elems(c) * h->max_batch));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_batch, hipEventDisableTiming));

    const size_t table = (size_t)c.n_elevations * c.n_sectors * (c.m / 2) * 2;
    HIP_TRY(h, hipHostMalloc(&h->h_result, sizeof(float) * table, hipHostMallocMapped));
    std::memset(h->h_result, 0, sizeof(float) * table);
    HIP_TRY(h, hipHostGetDevicePointer((void **)&h->d_result_tab, h->h_result, 0));
    const size_t ftable = (size_t)c.n_elevations * c.n_sectors * 2 * (1 + c.m / 2);
    HIP_TRY(h, hipHostMalloc(&h->h_frames, sizeof(unsigned) * ftable, hipHostMallocMapped));
    std::memset(h->h_frames, 0, sizeof(unsigned) * ftable);
    HIP_TRY(h, hipHostGetDevicePointer((void **)&h->d_frames_tab, h->h_frames, 0));

    h->slots.resize(c.n_slots);
    for (auto &s : h->slots) {
        HIP_TRY(h, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(h, hipHostMalloc(&s.h_iq, sizeof(float2) * sector_elems(c), hipHostMallocDefault));
        HIP_TRY(h, hipMalloc(&s.d_iq, sizeof(float2) * sector_elems(c)));
        HIP_TRY(h, hipMalloc(&s.d_mid, sizeof(float2) * mid_elems(c)));
        HIP_TRY(h, hipMalloc(&s.d_out, sizeof(float) * (c.m / 2) * 2));
        HIP_TRY(h, hipMalloc(&s.d_frames, sizeof(unsigned) * 2 * (1 + c.m / 2)));
    }
    return WRP_OK;
}

} // namespace

extern "C" {

void wrp_default_config(wrp_config *cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->m = 1024;           // rpv2.cu:40 n_sweeps
    cfg->n = 512;            // rpv2.cu:41 n_samples
    cfg->channels = 2;
    cfg->n_slots = 2;        // rpv2.cu:728
    cfg->n_sectors = 143;    // rpv2.cu:39
    cfg->n_elevations = 9;   // rpv2.cu:42
    cfg->ma_count = 7;       // rpv2.cu:45
    cfg->k_range_resolution = 30.f;
    cfg->k_calibration = 1941.05f;
    cfg->max_batch = 0;
    cfg->flags = 0;
}

int wrp_create(const wrp_config *cfg, int device, wrp_handle *out)
{
    if (!cfg || !out) return WRP_ERR_INVALID;
    *out = nullptr;
    if (cfg->m <= 0 || cfg->n <= 0 || cfg->n_slots < 1 || cfg->n_slots > 64 || cfg->n_sectors < 1 ||
        cfg->n_elevations < 1 || cfg->ma_count < 1 || cfg->ma_count > 9 || cfg->max_batch < 0 ||
        (cfg->flags & ~(0xff | WRP_FLAG_TWO_KERNELS | WRP_FLAG_ONE_TILE_PER_BLOCK | WRP_FLAG_DEBUG_FUSED_UNDERSIZED | WRP_FLAG_GENERIC_KERNELS | WRP_FLAG_WIRE_8)) != 0 ||
        ((cfg->flags & 0xff) != 0 && (cfg->flags & 0xff) != 8 && (cfg->flags & 0xff) != 16) || (cfg->channels != 2 && cfg->channels != 3) || device < 0)
        return WRP_ERR_INVALID;
    if (!shape_supported(cfg->m, cfg->n)) return WRP_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return WRP_ERR_HIP;
    wrp_engine *h = new (std::nothrow) wrp_engine();
    if (!h) return WRP_ERR_NOMEM;
    h->cfg = *cfg;
    h->device = device;
    int rc = create_impl(h);
    if (rc != WRP_OK) {
        destroy_impl(h);
        return rc;
    }
    *out = h;
    return WRP_OK;
}

void wrp_destroy(wrp_handle h) { destroy_impl(h); }

const char *wrp_strerror(int status)
{
    switch (status) {
    case WRP_OK: return "ok";
    case WRP_ERR_INVALID: return "invalid argument";
    case WRP_ERR_HIP: return "HIP runtime error";
    case WRP_ERR_NOMEM: return "out of memory";
    case WRP_ERR_UNSUPPORTED: return "unsupported sector shape";
    case WRP_ERR_STATE: return "call order violated";
    default: return "unknown wrp status";
    }
}

const char *wrp_last_hip_error(wrp_handle h) { return h ? h->hip_err.c_str() : ""; }

int wrp_pinned_slot(wrp_handle h, int slot, void **host_ptr, size_t *bytes)
{
    if (!h || slot < 0 || slot >= (int)h->slots.size() || !host_ptr) return WRP_ERR_INVALID;
    *host_ptr = h->slots[slot].h_iq;
    if (bytes) *bytes = sizeof(float2) * sector_elems(h->cfg);
    return WRP_OK;
}

int wrp_submit(wrp_handle h, int slot, int sector, int elevation)
{
    if (!h || slot < 0 || slot >= (int)h->slots.size() || sector < 0 || sector >= h->cfg.n_sectors ||
        elevation < 0 || elevation >= h->cfg.n_elevations)
        return WRP_ERR_INVALID;
    Slot &s = h->slots[slot];
    if (s.busy) return WRP_ERR_STATE;
    const wrp_config &c = h->cfg;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(s.d_iq, s.h_iq, sizeof(float2) * sector_elems(c), hipMemcpyHostToDevice, s.stream));
    // the products go straight into the host tables at (elevation, sector): see d_result_tab
    const size_t at = (size_t)elevation * c.n_sectors + sector;
    int rc = launch_chain(h, s.d_iq, 1, s.d_mid, h->d_result_tab + at * (c.m / 2) * 2, s.stream, nullptr,
                          h->d_frames_tab + at * 2 * (1 + c.m / 2), frame_header_word(sector, elevation));
    if (rc != WRP_OK) return rc;
    HIP_TRY(h, hipEventRecord(s.done, s.stream));
    s.busy = true;
    s.loaded = true;
    return WRP_OK;
}

int wrp_pinned_raw_slot(wrp_handle h, int slot, void **host_ptr, size_t *bytes)
{
    if (!h || slot < 0 || slot >= (int)h->slots.size() || !host_ptr) return WRP_ERR_INVALID;
    Slot &s = h->slots[slot];
    const size_t nbytes = (size_t)h->cfg.m * h->cfg.n * h->wire_bytes;
    if (!s.h_raw) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipHostMalloc(&s.h_raw, nbytes, hipHostMallocDefault));
        HIP_TRY(h, hipMalloc(&s.d_raw, nbytes));
    }
    *host_ptr = s.h_raw;
    if (bytes) *bytes = nbytes;
    return WRP_OK;
}

int wrp_submit_raw(wrp_handle h, int slot, int sector, int elevation)
{
    if (!h || slot < 0 || slot >= (int)h->slots.size() || sector < 0 || sector >= h->cfg.n_sectors ||
        elevation < 0 || elevation >= h->cfg.n_elevations)
        return WRP_ERR_INVALID;
    Slot &s = h->slots[slot];
    if (s.busy || !s.h_raw) return WRP_ERR_STATE;
    const wrp_config &c = h->cfg;
    const int count = c.m * c.n;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(s.d_raw, s.h_raw, (size_t)count * h->wire_bytes, hipMemcpyHostToDevice, s.stream));
    launch_decode(h, s.d_raw, s.d_iq, 1, s.stream);
    // the products go straight into the host tables at (elevation, sector): see d_result_tab
    const size_t at = (size_t)elevation * c.n_sectors + sector;
    int rc = launch_chain(h, s.d_iq, 1, s.d_mid, h->d_result_tab + at * (c.m / 2) * 2, s.stream, nullptr,
                          h->d_frames_tab + at * 2 * (1 + c.m / 2), frame_header_word(sector, elevation));
    if (rc != WRP_OK) return rc;
    HIP_TRY(h, hipEventRecord(s.done, s.stream));
    s.busy = true;
    s.loaded = true;
    return WRP_OK;
}

int wrp_wait(wrp_handle h, int slot)
{
    if (!h || slot < 0 || slot >= (int)h->slots.size()) return WRP_ERR_INVALID;
    Slot &s = h->slots[slot];
    if (!s.busy) return WRP_ERR_STATE;
    // A slot's chain takes ~0.1 - 0.2 ms: the caller (one feeder thread per GPU, rpv2.cu:665-683) first LOOKS for that long
    // -- hipEventSynchronize puts the thread to sleep, and the wake-up of a sleeping thread costs the cascade more than the
    // sector's transfer takes (profiles/r05/feeder_breakdown.log) -- and only then blocks.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(s.done);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(h, q);
        (void)hipGetLastError();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(WRP_WAIT_SPIN_US)) {
            HIP_TRY(h, hipEventSynchronize(s.done));
            break;
        }
    }
    s.busy = false;
    return WRP_OK;
}

int wrp_result(wrp_handle h, int sector, int elevation, const float **zdb_zdr)
{
    if (!h || !zdb_zdr || sector < 0 || sector >= h->cfg.n_sectors || elevation < 0 ||
        elevation >= h->cfg.n_elevations)
        return WRP_ERR_INVALID;
    *zdb_zdr = h->h_result + ((size_t)elevation * h->cfg.n_sectors + sector) * (h->cfg.m / 2) * 2;
    return WRP_OK;
}

int wrp_result_frame(wrp_handle h, int sector, int elevation, int which, int with_elevation, const unsigned char **frame, size_t *bytes)
{
    if (!h || !frame || which < 0 || which > 1 || sector < 0 || sector >= h->cfg.n_sectors || elevation < 0 ||
        elevation >= h->cfg.n_elevations)
        return WRP_ERR_INVALID;
    unsigned char *f = reinterpret_cast<unsigned char *>(frames_of(h, sector, elevation) + (size_t)which * (1 + h->cfg.m / 2));
    // the GPU wrote [sector BE16][elevation BE16][BE floats]; the 2-byte header of the UDP products (read_single.cc:510-517)
    // is the sector in front of the floats: two bytes rewritten in place, nothing copied
    f[0] = (unsigned char)((sector >> 8) & 0xff);
    f[1] = (unsigned char)(sector & 0xff);
    f[2] = (unsigned char)(((with_elevation ? elevation : sector) >> 8) & 0xff);
    f[3] = (unsigned char)((with_elevation ? elevation : sector) & 0xff);
    *frame = with_elevation ? f : f + 2;
    if (bytes) *bytes = sizeof(float) * (size_t)(h->cfg.m / 2) + (with_elevation ? 4 : 2);
    return WRP_OK;
}

// a batch on the two kernels; while the fused launch is disarmed (one gave up) every such batch of fused size counts down
// to its next chance
namespace {
int two_kernel_batch(wrp_engine *h, const void *in, bool raw, int n_sectors, float *d_out, hipStream_t st, Frames fr)
{
    const int rc = raw ? launch_two_kernel_raw_batch(h, (const unsigned char *)in, n_sectors, d_out, st, fr)
                       : launch_two_kernel_batch(h, (const float2 *)in, n_sectors, d_out, st, fr);
    if (rc == WRP_OK && h->fused && !h->fused_armed && n_sectors >= WRP_FUSED_MIN_SECTORS && --h->fused_cooldown <= 0)
        h->fused_armed = true;     // the fused launch gets another chance
    return rc;
}
} // namespace

// one batch, planar or wire format, with or without framed products: the fused launch where the shape, the size and the
// handle's state allow it, the two kernels otherwise
static int process_batch(wrp_handle h, const void *d_in, bool raw, int n_sectors, float *d_out, Frames fr, void *stream)
{
    if (!h || !d_in || !d_out || n_sectors < 0 || (fr.frames && !fr.hdrs)) return WRP_ERR_INVALID;
    if (n_sectors == 0) return WRP_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // every entry of the batch path looks at the fused launches that have completed since the last look
    int rc = reap_fused(h, false);
    if (rc != WRP_OK) return rc;
    if (h->fused_armed && n_sectors >= WRP_FUSED_MIN_SECTORS) return submit_fused(h, (const float2 *)d_in, n_sectors, d_out, (hipStream_t)stream, raw, fr);
    return two_kernel_batch(h, d_in, raw, n_sectors, d_out, stream ? (hipStream_t)stream : h->stream, fr);
}

int wrp_process_batch_device(wrp_handle h, const void *d_iq, int n_sectors, float *d_out, void *stream)
{
    return process_batch(h, d_iq, false, n_sectors, d_out, Frames{}, stream);
}

int wrp_process_batch_raw_device(wrp_handle h, const void *d_raw, int n_sectors, float *d_out, void *stream)
{
    return process_batch(h, d_raw, true, n_sectors, d_out, Frames{}, stream);
}

int wrp_process_batch_framed_device(wrp_handle h, const void *d_iq, int n_sectors, float *d_out, void *d_frames,
                                    const uint32_t *d_headers, void *stream)
{
    if (!d_frames || !d_headers) return WRP_ERR_INVALID;
    return process_batch(h, d_iq, false, n_sectors, d_out, Frames{(unsigned *)d_frames, (const unsigned *)d_headers}, stream);
}

int wrp_process_batch_raw_framed_device(wrp_handle h, const void *d_raw, int n_sectors, float *d_out, void *d_frames,
                                        const uint32_t *d_headers, void *stream)
{
    if (!d_frames || !d_headers) return WRP_ERR_INVALID;
    return process_batch(h, d_raw, true, n_sectors, d_out, Frames{(unsigned *)d_frames, (const unsigned *)d_headers}, stream);
}

uint32_t wrp_frame_header(int sector, int elevation) { return frame_header_word(sector, elevation); }

int wrp_check(wrp_handle h)
{
    if (!h) return WRP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const int rc = reap_fused(h, true);
    if (rc != WRP_OK) return rc;
    if (h->batch_pending) {
        HIP_TRY(h, hipEventSynchronize(h->ev_batch));
        h->batch_pending = false;
    }
    return WRP_OK;
}

int wrp_fused_fallbacks(wrp_handle h) { return h ? h->fused_fallbacks : 0; }
int wrp_fused_launches(wrp_handle h) { return h ? h->fused_launches : 0; }

int wrp_process_device(wrp_handle h, const void *d_iq, float *d_out, void *stream)
{
    return wrp_process_batch_device(h, d_iq, 1, d_out, stream);
}

int wrp_process_host(wrp_handle h, const void *iq_host, int n_sectors, float *out_host)
{
    if (!h || !iq_host || !out_host || n_sectors < 0) return WRP_ERR_INVALID;
    if (n_sectors == 0) return WRP_OK;
    const wrp_config &c = h->cfg;
    HIP_TRY(h, hipSetDevice(h->device));
    float2 *d_in = nullptr;
    float *d_out = nullptr;
    const size_t in_bytes = sizeof(float2) * sector_elems(c) * n_sectors;
    const size_t out_bytes = sizeof(float) * (c.m / 2) * 2 * (size_t)n_sectors;
    HIP_TRY(h, hipMalloc(&d_in, in_bytes));
    hipError_t e = hipMalloc(&d_out, out_bytes);
    if (e != hipSuccess) { (void)hipFree(d_in); h->hip_err = "hipMalloc(out)"; return WRP_ERR_NOMEM; }
    int rc = WRP_OK;
    e = hipMemcpyAsync(d_in, iq_host, in_bytes, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) rc = wrp_process_batch_device(h, d_in, n_sectors, d_out, nullptr);   // the engine's own stream
    if (e == hipSuccess && rc == WRP_OK) rc = wrp_check(h);     // waits; a fused launch that gave up is repeated here
    if (e == hipSuccess && rc == WRP_OK) e = hipMemcpyAsync(out_host, d_out, out_bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) { h->hip_err = hipGetErrorString(e); return WRP_ERR_HIP; }
    return rc;
}

int wrp_dump_stage(wrp_handle h, int slot, int stage, int channel, void *host_out)
{
    if (!h || slot < 0 || slot >= (int)h->slots.size() || !host_out || channel < 0 || channel > 1)
        return WRP_ERR_INVALID;   // VH (channel 2) is never processed
    Slot &s = h->slots[slot];
    if (s.busy || !s.loaded) return WRP_ERR_STATE;
    const wrp_config &c = h->cfg;
    size_t bytes = 0;
    switch (stage) {
    case WRP_STAGE_01HAMM: case WRP_STAGE_02FFT1: bytes = sizeof(float2) * (size_t)c.m * c.n; break;
    case WRP_STAGE_03FFT2_NOSHIFT: case WRP_STAGE_03FFT2: bytes = sizeof(float2) * (size_t)(c.m / 2) * c.n; break;
    case WRP_STAGE_04ABS: case WRP_STAGE_08POW: bytes = sizeof(float) * (size_t)(c.m / 2) * c.n; break;
    case WRP_STAGE_ROWSUM: bytes = sizeof(float) * (size_t)(c.m / 2); break;
    case WRP_STAGE_MID: bytes = sizeof(float2) * (size_t)(c.m / 2) * c.n; break;
    default: return WRP_ERR_INVALID;
    }
    if (stage == WRP_STAGE_MID) {   // what the production range pass hands to the Doppler pass: no dump instantiation involved
        HIP_TRY(h, hipSetDevice(h->device));
        int rc = launch_chain(h, s.d_iq, 1, s.d_mid, s.d_out, s.stream, nullptr);
        if (rc != WRP_OK) return rc;
        HIP_TRY(h, hipMemcpyAsync(host_out, s.d_mid + (size_t)channel * (c.m / 2) * c.n, bytes, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(h, hipStreamSynchronize(s.stream));
        return WRP_OK;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->dump_bytes < bytes) {
        if (h->d_dump) (void)hipFree(h->d_dump);
        h->d_dump = nullptr; h->dump_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_dump, bytes));
        h->dump_bytes = bytes;
    }
    wrp::DumpPtrs d{};
    d.channel = channel;
    switch (stage) {
    case WRP_STAGE_01HAMM: d.hamm = (float2 *)h->d_dump; break;
    case WRP_STAGE_02FFT1: d.fft1 = (float2 *)h->d_dump; break;
    case WRP_STAGE_03FFT2_NOSHIFT: d.noshift = (float2 *)h->d_dump; break;
    case WRP_STAGE_03FFT2: d.fft2 = (float2 *)h->d_dump; break;
    case WRP_STAGE_04ABS: d.abs2 = (float *)h->d_dump; break;
    case WRP_STAGE_08POW: d.pow = (float *)h->d_dump; break;
    case WRP_STAGE_ROWSUM: d.rowsum = (float *)h->d_dump; break;
    }
    int rc = launch_chain(h, s.d_iq, 1, s.d_mid, s.d_out, s.stream, &d);
    if (rc != WRP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(host_out, h->d_dump, bytes, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipStreamSynchronize(s.stream));
    return WRP_OK;
}

int wrp_time_batch_device(wrp_handle h, const void *d_iq, int n_sectors, float *d_out, int iters,
                          float *ms_total, float *ms_range, float *ms_doppler)
{
    if (!h || !d_iq || !d_out || n_sectors <= 0 || iters <= 0 || !ms_total) return WRP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = wrp_check(h);                       // idle handle
    if (rc != WRP_OK) return rc;
    const int fallbacks = h->fused_fallbacks;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    for (int it = 0; it < iters; it++) {
        rc = wrp_process_batch_device(h, d_iq, n_sectors, d_out, nullptr);
        if (rc != WRP_OK) return rc;
    }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    HIP_TRY(h, hipEventElapsedTime(ms_total, h->ev0, h->ev1));
    rc = wrp_check(h);
    if (rc != WRP_OK) return rc;
    if (h->fused_fallbacks != fallbacks) return WRP_ERR_HIP;   // a timing that includes a repeated batch is not one (hip_err says why)
    if ((ms_range || ms_doppler) && h->fused_armed && n_sectors >= WRP_FUSED_MIN_SECTORS) {
        if (ms_range) *ms_range = 0.f;      // one launch: there is no split to report
        if (ms_doppler) *ms_doppler = 0.f;
    } else if (ms_range || ms_doppler) {
        // second run: one event pair per launch (adds event overhead, so it is kept out of ms_total)
        const wrp_config &c = h->cfg;
        float tr = 0.f, td = 0.f;
        hipEvent_t e0, e1, e2;
        HIP_TRY(h, hipEventCreate(&e0));
        HIP_TRY(h, hipEventCreate(&e1));
        HIP_TRY(h, hipEventCreate(&e2));
        for (int it = 0; it < iters; it++) {
            for (int s0 = 0; s0 < n_sectors; s0 += h->max_batch) {
                const int cnt = std::min(h->max_batch, n_sectors - s0);
                const float2 *in = (const float2 *)d_iq + (size_t)s0 * sector_elems(c);
                float *out = d_out + (size_t)s0 * (c.m / 2) * 2;
                (void)hipEventRecord(e0, h->stream);
                launch_range(h, in, cnt, h->d_mid, h->stream, nullptr);
                (void)hipEventRecord(e1, h->stream);
                launch_doppler(h, h->d_mid, cnt, out, h->stream, nullptr);
                (void)hipEventRecord(e2, h->stream);
                (void)hipEventSynchronize(e2);
                float a = 0.f, b = 0.f;
                (void)hipEventElapsedTime(&a, e0, e1);
                (void)hipEventElapsedTime(&b, e1, e2);
                tr += a;
                td += b;
            }
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
        HIP_TRY(h, hipGetLastError());
        if (ms_range) *ms_range = tr;
        if (ms_doppler) *ms_doppler = td;
    }
    return WRP_OK;
}

// one fused launch on the engine's stream, waited for; its status word is looked at here (no repeat)
static int run_fused_sync_nocheck(wrp_engine *h, const float2 *d_iq, int n_sectors, float *d_out, unsigned long long *d_stamps,
                                  bool raw = false, float2 *d_tee = nullptr);
static int run_fused_sync(wrp_engine *h, const float2 *d_iq, int n_sectors, float *d_out, unsigned long long *d_stamps)
{
    const int rc = wrp_check(h);
    return rc != WRP_OK ? rc : run_fused_sync_nocheck(h, d_iq, n_sectors, d_out, d_stamps);
}
static int run_fused_sync_nocheck(wrp_engine *h, const float2 *d_iq, int n_sectors, float *d_out, unsigned long long *d_stamps,
                                  bool raw, float2 *d_tee)
{
    int rc = WRP_OK;
    const int slot = h->ring_next;
    h->ring_next = (h->ring_next + 1) % WRP_RING;
    rc = launch_fused(h, h->lane, d_iq, n_sectors, d_out, h->stream, slot, d_stamps, raw, Frames{}, d_tee);
    if (rc != WRP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned st = 0;
    for (int k = 0; k < WRP_RING; k++) { st |= h->h_status[k]; h->h_status[k] = 0; }   // (nothing else is outstanding: wrp_check came first)
    if (st) {
        h->lane.ctl_dirty = true;
        h->hip_err = "fused launch gave up (diagnostic entry: not repeated)";
        return WRP_ERR_HIP;
    }
    return WRP_OK;
}

int wrp_debug_fused_stamps(wrp_handle h, const void *d_iq, int n_sectors, float *d_out,
                           unsigned long long *host_stamps, size_t host_count)
{
    if (!h || !d_iq || !d_out || !host_stamps || n_sectors <= 0) return WRP_ERR_INVALID;
    if (!h->tuned && !h->tuned_b) return WRP_ERR_UNSUPPORTED;
    const size_t count = (size_t)h->n_cus * 2 * wrp::FUSED_STAMP_TASKS * wrp::FUSED_STAMPS;
    if (host_count < count) return WRP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned long long *d = nullptr;
    HIP_TRY(h, hipMalloc(&d, count * 8));
    hipError_t e = hipMemsetAsync(d, 0, count * 8, h->stream);
    int rc = e == hipSuccess ? wrp_check(h) : WRP_ERR_HIP;
    // the stamped launch runs BEHIND thirty ordinary ones, back to back on the same stream: the clocks it records are those
    // of the working point, not of a GPU that has just idled through the allocation above
    for (int k = 0; k < 30 && rc == WRP_OK; k++) {
        const int slot = h->ring_next;
        h->ring_next = (h->ring_next + 1) % WRP_RING;
        rc = launch_fused(h, h->lane, (const float2 *)d_iq, n_sectors, d_out, h->stream, slot);
    }
    if (rc == WRP_OK) rc = run_fused_sync_nocheck(h, (const float2 *)d_iq, n_sectors, d_out, d);
    if (e == hipSuccess) e = hipMemcpy(host_stamps, d, count * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { h->hip_err = hipGetErrorString(e); return WRP_ERR_HIP; }
    return rc;
}

int wrp_debug_fused_mid(wrp_handle h, const void *d_iq, int n_sectors, float *d_out, void *host_mid, size_t host_bytes)
{
    if (!h || !d_iq || !d_out || !host_mid || n_sectors < WRP_FUSED_MIN_SECTORS) return WRP_ERR_INVALID;
    if (!h->tuned && !h->tuned_b) return WRP_ERR_UNSUPPORTED;
    const size_t bytes = sizeof(float2) * wrp::FUSED_TEAM_ELEMS * 8;
    if (host_bytes < bytes) return WRP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const int rc = run_fused_sync(h, (const float2 *)d_iq, n_sectors, d_out, nullptr);
    if (rc != WRP_OK) return rc;
    HIP_TRY(h, hipMemcpy(host_mid, h->lane.d_pool, bytes, hipMemcpyDeviceToHost));
    return WRP_OK;
}

int wrp_debug_fused_tee(wrp_handle h, const void *d_in, int raw, int n_sectors, float *d_out, void *d_tee, size_t tee_bytes)
{
    if (!h || !d_in || !d_out || !d_tee || n_sectors < WRP_FUSED_MIN_SECTORS) return WRP_ERR_INVALID;
    if (!h->tuned && !h->tuned_b) return WRP_ERR_UNSUPPORTED;
    const wrp_config &c = h->cfg;
    if (tee_bytes < sizeof(float2) * (size_t)n_sectors * c.channels * (c.m / 2) * c.n) return WRP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const int rc = wrp_check(h);
    return rc != WRP_OK ? rc : run_fused_sync_nocheck(h, (const float2 *)d_in, n_sectors, d_out, nullptr, raw != 0, (float2 *)d_tee);
}

int wrp_get_config(wrp_handle h, wrp_config *cfg)
{
    if (!h || !cfg) return WRP_ERR_INVALID;
    *cfg = h->cfg;
    cfg->max_batch = h->max_batch;
    return WRP_OK;
}

size_t wrp_sector_bytes(wrp_handle h) { return h ? sizeof(float2) * sector_elems(h->cfg) : 0; }
size_t wrp_result_bytes(wrp_handle h) { return h ? sizeof(float) * (size_t)(h->cfg.m / 2) * 2 : 0; }
size_t wrp_algorithmic_bytes(wrp_handle h)
{
    return h ? (size_t)2 * h->cfg.m * h->cfg.n * 8 + (size_t)(h->cfg.m / 2) * 2 * 4 : 0;
}
const char *wrp_version(void) { return WRP_VERSION_STRING; }

int wrp_device_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return -1;
    for (char *c = bus; *c; c++) *c = (char)tolower(*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

} // extern "C"
