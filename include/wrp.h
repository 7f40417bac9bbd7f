/*
 * wrp.h -- C ABI of the MI355X per-sector weather-radar DSP engine (libwrp.so).
 *
 * The reference (rsatrioadi/weather-radar-processing) has no FFI: every GPU
 * variant is one main() with its kernels inline.  The seam this ABI replaces is
 * the stage-function set of rpv2.cu (same names as RadarProcessor's private
 * methods, radar_processor.h:63-84):
 *
 *   generate_constants + prepare_arys + initialize_streams   rpv2.cu:283-341  -> wrp_create
 *   p_iq pinned staging, Dimension4(n, m, 3, streams)        rpv2.cu:289-293  -> wrp_pinned_slot
 *   copy_matrix_to_device                                    rpv2.cu:399-407  \
 *   perform_stage_1 / _2 / _3                                rpv2.cu:409-570   > wrp_submit
 *   copy_result_to_host                                      rpv2.cu:581-618  /
 *   result[sitdim(2, m/2, sectors, elevations)]              rpv2.cu:620-630  -> wrp_wait + wrp_result
 *   commented stage dumps                                    rpv2.cu:582-603  -> wrp_dump_stage
 *   destroy_streams + destroy_arrays                         rpv2.cu:685-722  -> wrp_destroy
 *
 * Plain C types only; no HIP or torch types cross the boundary (a stream is
 * passed as void* = hipStream_t, NULL = the engine's own stream).
 *
 * Ownership: the handle owns all device and pinned memory.  The caller writes
 * only into a pinned slot, and only between wrp_wait(slot) (or creation) and
 * the next wrp_submit(slot).
 * Errors: the reference exits the process (gpuErrchk, rpv2.cu:21-27); this ABI
 * returns 0 or a negative wrp_status and never throws or exits.
 * Threading: one handle per GPU; a handle is not thread-safe; distinct handles
 * may be driven from distinct threads (one feeder thread per GPU).
 */
#ifndef WRP_H
#define WRP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wrp_engine *wrp_handle;

typedef enum {
    WRP_OK = 0,
    WRP_ERR_INVALID = -1,     /* bad argument */
    WRP_ERR_HIP = -2,         /* a HIP runtime call failed (see wrp_last_hip_error) */
    WRP_ERR_NOMEM = -3,
    WRP_ERR_UNSUPPORTED = -4, /* shape has no kernel instantiation */
    WRP_ERR_STATE = -5        /* call order violated (e.g. wait on an idle slot) */
} wrp_status;

/* Mirrors the compile-time constants of rpv2.cu:38-45 as run-time values. */
typedef struct {
    int m;            /* range cells per sector  = range-FFT length   (n_sweeps, 1024) */
    int n;            /* pulses per range cell   = Doppler-FFT length (n_samples, 512) */
    int channels;     /* planes per IQ block: 2 = HH,VV; 3 = HH,VV,VH (Dimension4 copies).
                         VH is carried but never read -- no output depends on it
                         (rpv2.cu:199-213 ignores offset_vh). */
    int n_slots;      /* stream slots = Dimension4 depth (rpv2.cu:728-734 num_streams) */
    int n_sectors;    /* sectors per elevation in the host result table (143) */
    int n_elevations; /* elevations in the host result table (9) */
    int ma_count;     /* moving-average taps (7), 1..9 */
    float k_range_resolution; /* 30    (rpv2.cu:43) */
    float k_calibration;      /* 1941.05 (rpv2.cu:44) */
    int max_batch;    /* sectors processed per internal chunk of wrp_process_batch_device;
                         sizes the device workspace (0 = default) */
    int flags;        /* 0 = the best measured configuration.  Bits 0-7: tuning, column tile of the
                         two-kernel range pass (8 or 16); WRP_FLAG_* below; other bits must be 0 */
} wrp_config;

/* The m = 1024, n = 512 shape and the m = 2048, n = 128 shape run batches of >= WRP_FUSED_MIN_SECTORS sectors as ONE persistent
 * launch whose XCD teams hand the intermediate from the range FFT to the Doppler rows through
 * their L2 (csrc/wrp_fused.h); smaller batches, the slot cascade (wrp_submit) and the shapes without
 * a fused kernel run a range-pass kernel and a Doppler-pass kernel.  Both forms perform the same
 * arithmetic and give bit-identical results.
 * WRP_FLAG_TWO_KERNELS: never use the fused launch (A/B measurements, parity tests). */
#define WRP_FLAG_TWO_KERNELS 0x800
#define WRP_FUSED_MIN_SECTORS 8
/* By default the two-kernel range pass is a fixed grid that walks the tiles and requests the
 * next tile while the current one is being transformed.  This flag selects the
 * one-tile-per-workgroup form instead (same arithmetic, bit-identical results; A/B only). */
#define WRP_FLAG_ONE_TILE_PER_BLOCK 0x400

/* m = 2048, n = 128 (BASELINE configs[4]) has tuned kernels of its own (csrc/wrp_shape_b.h); this flag runs the
 * shape-generic radix-2 kernels instead (every other shape always does): parity tests and A/B */
#define WRP_FLAG_GENERIC_KERNELS 0x8000
/* Wire format of this handle's raw entries (wrp_pinned_raw_slot, wrp_submit_raw, wrp_process_batch_raw[_framed]_device,
 * wrp_debug_fused_tee(raw)): 8 bytes per sample -- hhI hhQ vvI vvQ, big-endian int16 -- instead of 12: the sample of
 * sector.cpp:52-62 without its VH pair (bytes 8..11), which no output reads (rpv2.cu:199-213).  The feeder drops those bytes
 * in the copy it makes anyway (socket / file -> pinned slot: host/wire.h, wire_drop_vh): a third fewer PCIe bytes per sector
 * (4 MiB instead of 6 at m = 1024, n = 512), and the slot path is PCIe-bound.  Results are bit-identical to the 12-byte
 * entries on the same samples.  With channels = 3 the VH plane of the decoded block is left as it is (never read). */
#define WRP_FLAG_WIRE_8 0x10000
/* test hook: the fused launch is issued with half its workgroups, so that it must report (wrp_check)
 * that its teams are incomplete and the handle must fall back to the two kernels */
#define WRP_FLAG_DEBUG_FUSED_UNDERSIZED 0x4000

/* Stage ids for wrp_dump_stage; names follow the reference's fixture files. */
typedef enum {
    WRP_STAGE_01HAMM = 1,         /* m x n complex  after the Hamming window          (a2) */
    WRP_STAGE_02FFT1 = 2,         /* m x n complex  after the range FFT               (a3) */
    WRP_STAGE_03FFT2_NOSHIFT = 3, /* m/2 x n complex Doppler FFT before conj/shift/clip    */
    WRP_STAGE_03FFT2 = 4,         /* m/2 x n complex after conj + shift + clip        (a5) */
    WRP_STAGE_04ABS = 5,          /* m/2 x n real   |.|^2                             (a6) */
    WRP_STAGE_08POW = 6,          /* m/2 x n real   MA-smoothed power                 (a7) */
    WRP_STAGE_ROWSUM = 7,         /* m/2 real       S[i] (a8) AS THE CHAIN FORMS IT: (sum of the taps) x (sum over j of 04ABS[i][j]) --
                                     the row sum of the CIRCULAR moving average is its DC bin (read.cc:290-301, 336-339) -- and NOT the
                                     sum of the dumped 08POW row, from which it differs by rounding only (tested: <= 2e-6 relative) */
    WRP_STAGE_MID = 8             /* m/2 x n complex: the half-height intermediate exactly as the production range pass of
                                     the two-kernel path hands it to the Doppler pass (= rows < m/2 of 02FFT1; no dump
                                     instantiation involved) */
} wrp_stage;

/* Fill *cfg with the reference's constants (m=1024, n=512, channels=2, 2 slots,
 * 143 sectors, 9 elevations, 7 taps, 30, 1941.05). */
void wrp_default_config(wrp_config *cfg);

/* generate_constants + prepare_arys + initialize_streams (rpv2.cu:283-341):
 * window / twiddle / MA tables, per-slot device buffers, pinned staging, streams. */
int wrp_create(const wrp_config *cfg, int device, wrp_handle *out);

/* destroy_streams + destroy_arrays (rpv2.cu:685-722). NULL is a no-op. */
void wrp_destroy(wrp_handle h);

const char *wrp_strerror(int status);
/* Text of the last failing HIP call on this handle ("" if none). */
const char *wrp_last_hip_error(wrp_handle h);

/* p_iq of rpv2.cu:291: pinned, caller-fillable staging buffer of one slot, laid out
 * as Dimension4(n, m, channels, slots) at depth = slot, i.e.
 * [channel][row i < m][col j < n] interleaved complex fp32.  *bytes = channels*m*n*8. */
int wrp_pinned_slot(wrp_handle h, int slot, void **host_ptr, size_t *bytes);

/* copy_matrix_to_device + perform_stage_1..3 + copy_result_to_host (rpv2.cu:399-618),
 * asynchronous on the slot's stream: H2D of the pinned slot, the fused chain, D2H of
 * [m/2][2] floats into the host result table at (elevation, sector). */
int wrp_submit(wrp_handle h, int slot, int sector, int elevation);

/* Wire-format ingest.  The reference decodes a sector on the CPU
 * (Sector::fromByteArray, sector.cpp:52-62) and scatters int16 -> float2 into p_iq
 * (rpv2.cu:369-383, its "restructuring" milliseconds).  Here the caller copies the datagrams
 * as received -- m*n samples of 12 bytes: hhI hhQ vvI vvQ vhI vhQ, big-endian int16 -- into the
 * slot's pinned raw buffer (*bytes = m*n*12) and wrp_submit_raw uploads them (half the PCIe
 * bytes of the 3-plane fp32 block) and decodes on the GPU, bit-identically, before the chain.
 * A handle created with WRP_FLAG_WIRE_8 takes 8-byte samples here (*bytes = m*n*8). */
int wrp_pinned_raw_slot(wrp_handle h, int slot, void **host_ptr, size_t *bytes);
int wrp_submit_raw(wrp_handle h, int slot, int sector, int elevation);

/* Block until the slot's last wrp_submit / wrp_submit_raw has completed.  The call first polls the slot's event for up to
 * 0.5 ms (a slot's chain takes 0.1 - 0.2 ms; the wake-up of a thread that went to sleep in hipEventSynchronize costs the
 * cascade more than a sector's transfer takes) and only then sleeps: one feeder thread per GPU, as rpv2.cu:665-683.
 * The products of the slot path reach the host tables (wrp_result, wrp_result_frame) without a copy: the Doppler pass writes
 * them into the pinned tables itself. */
int wrp_wait(wrp_handle h, int slot);

/* Pointer into the host result table result[sitdim(2, m/2, n_sectors, n_elevations)]
 * (rpv2.cu:736, :626-629): (*zdb_zdr)[gate*2 + 0] = Zdb, [gate*2 + 1] = Zdr. */
int wrp_result(wrp_handle h, int sector, int elevation, const float **zdb_zdr);

/* Egress framing on the GPU.  The reference converts every product to big-endian floats on the CPU
 * and prepends the header (rpv2.cu:631-661: [sector BE16][elevation BE16][m/2 BE floats], topics "B" = Zdb and "C" = Zdr;
 * read_single.cc:510-520: [sector BE16][m/2 BE floats] on UDP 19002 / 19003).  Here the Doppler pass writes both products
 * planar and big-endian behind their header and the slot's D2H copy delivers them wire-ready: *frame points INTO the
 * pinned frame table (which = 0: Zdb, 1: Zdr; with_elevation selects the 4-byte or the 2-byte header; *bytes = 4*(m/2)
 * + 4 or + 2) -- send it as it is.  Valid after wrp_wait of the slot that processed (sector, elevation), until that
 * pair is submitted again OR the next wrp_result_frame call for the same (sector, elevation, which): the two header forms
 * share their bytes (the call rewrites bytes 0..3 of the frame in place), so a frame is fetched, sent, and only then
 * fetched in its other form; one thread per handle, as everywhere in this ABI.  The 2-byte form starts 2 bytes into a
 * word: its floats are not 4-byte aligned (it is a byte string for a socket). */
int wrp_result_frame(wrp_handle h, int sector, int elevation, int which, int with_elevation, const unsigned char **frame,
                     size_t *bytes);

/* Kernel-only entries (device-resident in and out; used for roofline timing and by
 * callers that already hold the IQ block on the GPU).
 * d_iq : [n_sectors][channels][m][n] complex fp32;  d_out : [n_sectors][m/2][2] fp32.
 * stream: hipStream_t as void*, NULL = the engine's compute stream.  Asynchronous. */
int wrp_process_device(wrp_handle h, const void *d_iq, float *d_out, void *stream);
int wrp_process_batch_device(wrp_handle h, const void *d_iq, int n_sectors, float *d_out, void *stream);
/* The same for a batch in the WIRE format: d_raw = [n_sectors][m*n samples][12 bytes], hhI hhQ vvI vvQ vhI
 * vhQ as big-endian int16 (sector.cpp:52-62) -- what wrp_pinned_raw_slot takes, for a whole sweep that already lies in
 * device memory.  m = 1024, n = 512 and m = 2048, n = 128, >= WRP_FUSED_MIN_SECTORS sectors: the tile workgroups of the
 * persistent launch read the samples themselves (byte swap + conversion in registers, in place of the fp32 loads): 6 MiB
 * (3 MiB) of HBM reads per sector instead of 8 (4) and no decode pass.  Otherwise (small batches, other shapes,
 * WRP_FLAG_TWO_KERNELS, the repeat of a launch that gave up) the batch is decoded on the GPU, up to max_batch sectors at a
 * time, in front of the two kernels.  Results are bit-identical to Sector::fromByteArray + the scatter of
 * rpv2.cu:372-383 + wrp_process_batch_device.  WRP_FLAG_WIRE_8: d_raw = [n_sectors][m*n samples][8 bytes] (4 / 2 MiB per
 * sector); at m = 2048, n = 128 a tile member's bytes are then the planar form's (64 of every 1 KiB row). */
int wrp_process_batch_raw_device(wrp_handle h, const void *d_raw, int n_sectors, float *d_out, void *stream);

/* Egress framing for batches (rpv2.cu:631-661, read_single.cc:510-520).  The batch entries above with one
 * more output: d_frames = [n_sectors][2][1 + m/2] 32-bit words in device memory -- per sector the Zdb frame, then the Zdr
 * frame, each a header word followed by m/2 BIG-ENDIAN floats, written by the same lanes that write d_out (all launch
 * forms).  d_headers = [n_sectors] header words in device memory, copied as they are in front of both frames of their
 * sector: wrp_frame_header(sector, elevation) gives the word whose bytes are [sector BE16][elevation BE16] (rpv2.cu's
 * topics "B" / "C"); for the 2-byte header of read_single.cc pass wrp_frame_header(x, sector) and send from byte 2.
 * Frame (s, which) = the 4 * (1 + m/2) bytes at d_frames + (2 s + which) * (1 + m/2) words.  Same completion rules as
 * d_out (below). */
uint32_t wrp_frame_header(int sector, int elevation);
int wrp_process_batch_framed_device(wrp_handle h, const void *d_iq, int n_sectors, float *d_out, void *d_frames,
                                    const uint32_t *d_headers, void *stream);
int wrp_process_batch_raw_framed_device(wrp_handle h, const void *d_raw, int n_sectors, float *d_out, void *d_frames,
                                        const uint32_t *d_headers, void *stream);

/* Completion and ordering.  The batch is ordered on the stream it is given (NULL: the engine's own stream);
 * ONE fused launch is in flight per handle: a batch on another stream first waits, on the device, for the
 * previous one.  wrp_check waits for every batch submitted so far.
 * The fused launch needs all its workgroups resident at once and says so, within milliseconds, when they
 * are not (e.g. another kernel occupies CUs); such a batch is REPEATED on the two-kernel path:
 *   - a batch given a CALLER's stream: the repeat is queued on that stream right behind the launch, gated on the
 *     launch's status word in device memory (when the launch has succeeded its kernels are a few workgroup rows that return
 *     at once: 2 % of a 360-sector batch, bench.py `caller_stream`).  STREAM ORDER
 *     IS ENOUGH: when the stream has passed the batch, d_out (and d_frames) are right, and d_iq / d_raw may be
 *     reused -- like any other asynchronous work on a stream; wrp_check is not needed for correctness.
 *   - a batch on the engine's own stream (stream = NULL): only this library can wait for that stream.  wrp_check (or the
 *     next batch call, once the launch has completed) repeats a launch that gave up, re-reading d_iq / d_raw: outputs
 *     are valid when wrp_check has returned WRP_OK, and the INPUT MUST STAY VALID AND UNMODIFIED UNTIL THEN.
 * After a launch that gave up the handle stays on the two kernels for 16 batches and then tries the fused launch again;
 * wrp_last_hip_error holds the note of the first failure of such a run, wrp_fused_fallbacks counts the repeated batches
 * (launches that were already queued behind the failed one fail on its status and are repeated, and counted, too).
 * wrp_process_host and wrp_time_batch_device check by themselves. */
int wrp_check(wrp_handle h);
int wrp_fused_fallbacks(wrp_handle h);
/* Batches (or pieces of batches) that were issued as a fused launch so far -- which path a batch took (tests, harnesses). */
int wrp_fused_launches(wrp_handle h);

/* Synchronous convenience: host buffers in the same layouts (pageable or pinned). */
int wrp_process_host(wrp_handle h, const void *iq_host, int n_sectors, float *out_host);

/* Debug/parity: re-run the chain on the slot's device IQ block (as last uploaded by
 * wrp_submit) with stage dumping enabled and copy stage `stage` of `channel`
 * (0 = HH, 1 = VV; VH is carried but never processed: WRP_ERR_INVALID) to host_out (sizes per
 * wrp_stage).  Dumps come from the two-kernel form, whose results the fused launch reproduces
 * bit for bit (tested).  m = 2048, n = 128: 03FFT2_NOSHIFT .. ROWSUM and MID come from the tuned kernels of
 * that shape, 01HAMM and 02FFT1 (all m rows, which the tuned range pass never forms) from the shape-generic
 * kernels.  Synchronous. */
int wrp_dump_stage(wrp_handle h, int slot, int stage, int channel, void *host_out);

/* Measurement: run `iters` back-to-back wrp_process_batch_device calls on the engine's
 * own stream bracketed by HIP events ON THAT STREAM; *ms_total = elapsed time of all
 * iterations; ms_range / ms_doppler (optional) = summed time of the range-pass and
 * Doppler-pass kernels measured by per-launch event pairs in a second, separate run. */
int wrp_time_batch_device(wrp_handle h, const void *d_iq, int n_sectors, float *d_out,
                          int iters, float *ms_total, float *ms_range, float *ms_doppler);

/* Diagnostics: one fused launch (a separate instantiation) with in-kernel phase stamps (100 MHz
 * ticks) copied to host_stamps[2 x CUs workgroups][16 tasks][9].  Slot 8 of task 0 identifies the
 * workgroup: kind << 32 | xcc << 16 | rank; the meaning of slots 0-7 per kind is listed in
 * tools/fused_stamps.py.  Synchronous; timing of this call is not representative. */
int wrp_debug_fused_stamps(wrp_handle h, const void *d_iq, int n_sectors, float *d_out,
                           unsigned long long *host_stamps, size_t host_count);

/* Parity of the intermediate: one fused launch over n_sectors >= WRP_FUSED_MIN_SECTORS sectors, then the
 * teams' L2-resident hand-over slots copied to host_mid (8 slots of 1 MiB).  Slot x holds what went through it
 * last: half 1 -- the range-FFT gates g < m/2 with (g mod 16) >= 8, gate g in row (g >> 4) * 8 + (g & 7) -- of
 * the LAST task of the team on XCD x (with n_sectors = 8 on an 8-XCD device: sector x):
 *   m = 1024, n = 512: [256 rows][512] of the VV channel;  m = 2048, n = 128: [2 channels][512 rows][128].
 * Must equal those rows of wrp_dump_stage(WRP_STAGE_MID) bit for bit. */
int wrp_debug_fused_mid(wrp_handle h, const void *d_iq, int n_sectors, float *d_out, void *host_mid, size_t host_bytes);

/* Parity of the WHOLE intermediate: one fused launch (its diagnostics instantiation) whose tile workgroups also write
 * everything they put through the hand-over slots -- both halves of every task -- to d_tee, device memory,
 * [n_sectors][channels][m/2 gates][n] complex.  raw != 0: d_in is the handle's wire format (12 or 8 bytes per sample) and the
 * launch the wire-format one.  Every [m/2][n] block must equal wrp_dump_stage(WRP_STAGE_MID) of that sector
 * and channel bit for bit: the stage dumps come from the two-kernel path, this ties the launch the bench times to them
 * (rpv2.cu:409-502: the reference's intermediate after its range FFT).  Synchronous. */
int wrp_debug_fused_tee(wrp_handle h, const void *d_in, int raw, int n_sectors, float *d_out, void *d_tee, size_t tee_bytes);

/* Introspection for harnesses. */
int wrp_get_config(wrp_handle h, wrp_config *cfg);
size_t wrp_sector_bytes(wrp_handle h);   /* channels*m*n*8 */
size_t wrp_result_bytes(wrp_handle h);   /* (m/2)*2*4      */
/* Algorithmic HBM bytes per sector of the fused chain:
 * 2*m*n*8 (HH and VV read once) + (m/2)*2*4 (result written once). */
size_t wrp_algorithmic_bytes(wrp_handle h);
const char *wrp_version(void);
/* NUMA node of the host the GPU hangs on (sysfs numa_node of its PCI device), or -1 when unknown:
 * the feeder thread of a GPU fills its pinned slots fastest from that node (host/rpv2.cpp --bind-numa). */
int wrp_device_numa_node(int device);

#ifdef __cplusplus
}
#endif
#endif /* WRP_H */
