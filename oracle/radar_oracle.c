/*
 * radar_oracle.c -- CPU oracle for the per-sector weather-radar DSP chain.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C restatement
 * of the reference's CPU algorithm (read.cc = fp64, read_single.cc = fp32) plus
 * the three host codecs on either side of the hot path (sector.cpp, floats.c,
 * dimension.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker / the reported CPU
 * baseline.  The product (libwrp.so) never links or calls anything in oracle/.
 *
 * Pinning (how we know this restatement IS the reference's algorithm):
 *   * out/cpu.bin (127 identical records of 512 fp32 Zdb values written by the
 *     reference's fp32 CPU program on the synthetic sector iq_hh[i][j] = (i, j),
 *     gpu_1fp.cu:295-300) pins the WHOLE HH chain a0..a9 end to end;
 *   * in/04abs.altb -> in/08pow.altb pins the MA convolution (a7) on real data;
 *   * in/08pow.altb -> in/09zdb.altb pins row-sum + reflectivity (a8, a9);
 *   * out/99result.{cpu,gpu}.out == (09zdb, 10zdr) column-wise;
 *   * out/sum.out pins the tree reduction toy;  byte vectors in SURVEY.md
 *     §2 pin the wire codecs; oracle/_ref (the reference's own sector.cpp /
 *     floats.c / dimension.cpp compiled where they lie) pins them bit-exactly.
 * The reference's CPU programs themselves are NOT buildable in this image
 * (read.cc needs <fftw3.h> and libfftw3, read_single.cc needs libfftw3f; neither
 * exists here and writing stand-ins is not allowed), see DESIGN.md.
 * Zdr is pinned only through 10zdr's consistency with 09zdb (the VV
 * intermediates of the real sector were lost, .MISSING_LARGE_BLOBS) -- the VV
 * channel runs the same code as the pinned HH channel.
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define REAL double
#define SUF(x) x##_f64
#include "radar_oracle_impl.inc"
#undef REAL
#undef SUF

#define REAL float
#define SUF(x) x##_f32
#include "radar_oracle_impl.inc"
#undef REAL
#undef SUF

/* ---- host codecs --------------------------------------------------------- */

/* sector.cpp:52-62 Sector::fromByteArray -- 12 bytes per sample:
 * hhI hhQ vvI vvQ vhI vhQ, each a big-endian int16; outputs are interleaved
 * (I,Q) short arrays of 2*sweeps*samples entries per channel. */
void wro_sector_from_bytes(const unsigned char *buff, int sweeps, int samples,
                           short *hh, short *vv, short *vh)
{
    size_t idx = 0;
    for (size_t i = 0; i < (size_t)sweeps * samples; i++) {
        short *dst[3] = { hh, vv, vh };
        for (int c = 0; c < 3; c++)
            for (int q = 0; q < 2; q++) {
                unsigned hi = buff[idx++], lo = buff[idx++];
                dst[c][i * 2 + q] = (short)(((hi << 8) & 0xff00) + (lo & 0xff));
            }
    }
}

/* rpv2.cu:369-383 read_matrix scatter: shorts -> planar complex fp32 in
 * Dimension4(width=n, height=m, copies=3, depth) layout at `depth` = slot. */
void wro_sector_to_planar(const short *hh, const short *vv, const short *vh,
                          int m, int n, int copies, int slot, float *p_iq)
{
    size_t idx = 0;
    const short *src[3] = { hh, vv, vh };
    for (int j = 0; j < m; j++)
        for (int i = 0; i < n; i++) {
            size_t a = idx++, b = idx++;
            for (int c = 0; c < copies && c < 3; c++) {
                size_t o = (size_t)j * n + i + (size_t)c * n * m + (size_t)slot * n * m * copies;
                p_iq[2 * o] = (float)src[c][a];
                p_iq[2 * o + 1] = (float)src[c][b];
            }
        }
}

/* floats.c:3-13 ftob / :15-30 btof / :32-42 aftoab, abtoaf -- float <-> 4
 * big-endian bytes (restated without the reference's long* type punning) */
void wro_ftob(float f, unsigned char *b)
{
    uint32_t u; memcpy(&u, &f, 4);
    b[0] = (unsigned char)(u >> 24); b[1] = (unsigned char)(u >> 16);
    b[2] = (unsigned char)(u >> 8);  b[3] = (unsigned char)u;
}
float wro_btof(const unsigned char *b)
{
    uint32_t u = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    float f; memcpy(&f, &u, 4); return f;
}
void wro_aftoab(const float *af, size_t n, unsigned char *ab)
{
    for (size_t i = 0; i < n; i++) wro_ftob(af[i], &ab[i * 4]);
}
void wro_abtoaf(const unsigned char *ab, size_t n, float *af)
{
    for (size_t i = 0; i < n; i++) af[i] = wro_btof(&ab[i * 4]);
}

/* dimension.cpp:9-11 and :19-21 */
int wro_dim3_at_depth(int w, int h, int x, int y, int depth)
{
    return y * w + x + depth * w * h;
}
int wro_dim4_copy_at_depth(int w, int h, int copies, int x, int y, int copy, int depth)
{
    return y * w + x + copy * w * h + depth * w * h * copies;
}

/* rpv2.cu:620-644 send_results framing: [sector BE16][elev BE16][gates BE floats].
 * `which` 0 = zdb, 1 = zdr; zdb_zdr is [gates][2]. Returns bytes written. */
size_t wro_frame_result(const float *zdb_zdr, int gates, int sector, int elevation,
                        int which, int with_elevation, unsigned char *out)
{
    size_t o = 0;
    out[o++] = (unsigned char)((sector >> 8) & 0xff);
    out[o++] = (unsigned char)(sector & 0xff);
    if (with_elevation) { /* read_single.cc:510-520 omits these two bytes */
        out[o++] = (unsigned char)((elevation >> 8) & 0xff);
        out[o++] = (unsigned char)(elevation & 0xff);
    }
    for (int i = 0; i < gates; i++) { wro_ftob(zdb_zdr[2 * i + which], &out[o]); o += 4; }
    return o;
}

/* error.cpp:15-30 -- relative L2 error over the first n floats, skipping
 * pairs where either value is non-finite */
float wro_rel_l2(const float *cpu, const float *gpu, int n)
{
    float sigdelt = 0.f, sig = 0.f;
    for (int i = 0; i < n; i++) {
        float ue = cpu[i], uc = gpu[i];
        if (isfinite(ue) && isfinite(uc)) { sigdelt += (ue - uc) * (ue - uc); sig += ue * ue; }
    }
    return sqrtf(sigdelt / sig);
}

/* examples/sum.cu:33-71 tree reduction toy (pinned by out/sum.out): per row,
 * repeatedly fold the upper half onto the lower half IN PLACE; like the dump in
 * out/sum.out the whole folded row is returned (element 0 is the row sum) */
void wro_tree_sum_rows(int rows, int n, const float *in_interleaved, float *out_interleaved)
{
    for (int i = 0; i < rows; i++) {
        float *tmp = &out_interleaved[(size_t)i * n * 2];
        memcpy(tmp, &in_interleaved[(size_t)i * n * 2], sizeof(float) * 2 * n);
        for (int s = n / 2; s > 0; s >>= 1)
            for (int j = 0; j < s; j++) {
                tmp[2 * j] += tmp[2 * (j + s)];
                tmp[2 * j + 1] += tmp[2 * (j + s) + 1];
            }
    }
}
