// floats.cpp -- big-endian float serialisation (reference semantics: floats.c:3-42) without
// the reference's long* type punning.
#include "floats.h"

#include <stdint.h>
#include <string.h>

void ftob(float f, unsigned char *b)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    b[0] = (unsigned char)(u >> 24);
    b[1] = (unsigned char)(u >> 16);
    b[2] = (unsigned char)(u >> 8);
    b[3] = (unsigned char)u;
}

float btof(unsigned char *b)
{
    const uint32_t u = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void aftoab(float *af, size_t numfloats, unsigned char *ab)
{
    for (size_t i = 0; i < numfloats; i++) ftob(af[i], ab + 4 * i);
}

void abtoaf(unsigned char *ab, size_t numfloats, float *af)
{
    for (size_t i = 0; i < numfloats; i++) af[i] = btof(ab + 4 * i);
}
