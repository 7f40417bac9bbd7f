#include "framing.h"

#include <vector>

#include "floats.h"

size_t frame_result(const float *zdb_zdr, int gates, int sector, int elevation, int which, bool with_elevation,
                    unsigned char *out)
{
    size_t o = 0;
    out[o++] = (unsigned char)((sector >> 8) & 0xff);
    out[o++] = (unsigned char)(sector & 0xff);
    if (with_elevation) {
        out[o++] = (unsigned char)((elevation >> 8) & 0xff);
        out[o++] = (unsigned char)(elevation & 0xff);
    }
    std::vector<float> col(gates);
    for (int i = 0; i < gates; i++) col[i] = zdb_zdr[2 * i + which];   // rpv2.cu:626-629
    aftoab(col.data(), (size_t)gates, out + o);                         // rpv2.cu:643-644
    return o + 4 * (size_t)gates;
}
