// floats.h -- IEEE-754 single precision <-> four big-endian bytes.
//
// The radar's consumers expect every product value as 4 bytes, most significant first
// (read_single.cc:516-517, rpv2.cu:643-644).  The four free functions below keep the names and
// signatures a caller of the reference's floats.h:6-9 uses (C++ linkage, as the reference
// compiles floats.c with g++, Makefile:2); wrp_floats:: holds the bit-level helpers they share.
#ifndef WRP_HOST_FLOATS_H
#define WRP_HOST_FLOATS_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace wrp_floats {
inline uint32_t bits_of(float f)
{
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
}
inline float float_of(uint32_t u)
{
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
}
inline void store_be32(uint32_t u, unsigned char *p)
{
    for (int k = 0; k < 4; k++) p[k] = (unsigned char)(u >> (24 - 8 * k));
}
inline uint32_t load_be32(const unsigned char *p)
{
    uint32_t u = 0;
    for (int k = 0; k < 4; k++) u = (u << 8) | p[k];
    return u;
}
}   // namespace wrp_floats

void ftob(float f, unsigned char *buffer);                        // one float  -> buffer[0..3]
float btof(unsigned char *buffer);                                // buffer[0..3] -> one float
void aftoab(float *af, size_t numfloats, unsigned char *ab);      // array forms
void abtoaf(unsigned char *ab, size_t numfloats, float *af);

#endif   // WRP_HOST_FLOATS_H
