// floats.cpp -- see floats.h.  Bit-exact with the reference's floats.c on every pattern,
// including infinities, NaN payloads and signed zero (tests/test_host_cpu.py), without its
// long* type punning.
#include "floats.h"

using namespace wrp_floats;

void ftob(float f, unsigned char *buffer) { store_be32(bits_of(f), buffer); }

float btof(unsigned char *buffer) { return float_of(load_be32(buffer)); }

void aftoab(float *af, size_t numfloats, unsigned char *ab)
{
    for (const float *p = af, *end = af + numfloats; p != end; ++p, ab += 4) store_be32(bits_of(*p), ab);
}

void abtoaf(unsigned char *ab, size_t numfloats, float *af)
{
    for (float *p = af, *end = af + numfloats; p != end; ++p, ab += 4) *p = float_of(load_be32(ab));
}
