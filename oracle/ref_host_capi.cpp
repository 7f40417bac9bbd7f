// ref_host_capi.cpp -- C-linkage adapter over the REFERENCE's own host codecs.
//
// TEST INFRASTRUCTURE.  Compiled together with /root/reference/{sector.cpp,
// floats.c,dimension.cpp} (where they lie, never copied) into
// oracle/_ref/libref_host.so by oracle/Makefile, so that ctypes can call the
// real Sector::fromByteArray / aftoab / Dimension4::copy_at_depth and the tests
// can check our restatement and our product against them bit for bit.
// Nothing here re-implements reference behaviour: every function forwards.
#include <cstddef>
#include <cstring>

#include "dimension.h"
#include "floats.h"
#include "sector.h"

extern "C" {

// sector.h:13-17
void ref_sector_from_bytes(char *buff, int sweeps, int samples, short *hh, short *vv, short *vh)
{
    Sector s(sweeps, samples);
    s.fromByteArray(buff);
    const size_t cnt = sizeof(short) * 2 * (size_t)sweeps * samples;
    memcpy(hh, s.hh, cnt);
    memcpy(vv, s.vv, cnt);
    memcpy(vh, s.vh, cnt);
}

// floats.h:6-9
void ref_ftob(float f, unsigned char *b) { ftob(f, b); }
float ref_btof(unsigned char *b) { return btof(b); }
void ref_aftoab(float *af, size_t n, unsigned char *ab) { aftoab(af, n, ab); }
void ref_abtoaf(unsigned char *ab, size_t n, float *af) { abtoaf(ab, n, af); }

// dimension.h:4-16
int ref_dim3_at_depth(int w, int h, int d, int x, int y, int depth)
{
    Dimension3 dim(w, h, d);
    return dim.at_depth(x, y, depth);
}
int ref_dim4_copy_at_depth(int w, int h, int c, int d, int x, int y, int copy, int depth)
{
    Dimension4 dim(w, h, c, d);
    return dim.copy_at_depth(x, y, copy, depth);
}
void ref_dim4_sizes(int w, int h, int c, int d, int *m_size, int *total_size)
{
    Dimension4 dim(w, h, c, d);
    *m_size = dim.m_size;
    *total_size = dim.total_size;
}

} // extern "C"
