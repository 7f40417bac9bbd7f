// host_capi.cpp -- C-linkage view of the host types for tests and non-C++ callers.
#include <string.h>

#include "dimension.h"
#include "floats.h"
#include "framing.h"
#include "sector.h"

extern "C" {

void wrph_sector_from_bytes(char *buff, int sweeps, int samples, short *hh, short *vv, short *vh)
{
    Sector s(sweeps, samples);
    s.fromByteArray(buff);
    const size_t cnt = sizeof(short) * 2 * (size_t)sweeps * samples;
    memcpy(hh, s.hh, cnt);
    memcpy(vv, s.vv, cnt);
    memcpy(vh, s.vh, cnt);
}
void wrph_aftoab(float *af, size_t n, unsigned char *ab) { aftoab(af, n, ab); }
void wrph_abtoaf(unsigned char *ab, size_t n, float *af) { abtoaf(ab, n, af); }
int wrph_dim3_at_depth(int w, int h, int d, int x, int y, int depth) { return Dimension3(w, h, d).at_depth(x, y, depth); }
int wrph_dim4_copy_at_depth(int w, int h, int c, int d, int x, int y, int copy, int depth)
{
    return Dimension4(w, h, c, d).copy_at_depth(x, y, copy, depth);
}
size_t wrph_frame_result(const float *zdb_zdr, int gates, int sector, int elevation, int which, int with_elevation,
                         unsigned char *out)
{
    return frame_result(zdb_zdr, gates, sector, elevation, which, with_elevation != 0, out);
}

} // extern "C"
