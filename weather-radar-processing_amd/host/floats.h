// floats.h -- float <-> 4 big-endian bytes, API-compatible with the reference (floats.h:6-9).
// C++ linkage, as the reference builds floats.c with g++ (Makefile:2).
#ifndef WRP_HOST_FLOATS_H
#define WRP_HOST_FLOATS_H
#include <stddef.h>

void ftob(float f, unsigned char *buffer);
float btof(unsigned char *buffer);
void aftoab(float *af, size_t numfloats, unsigned char *ab);
void abtoaf(unsigned char *ab, size_t numfloats, float *af);
#endif
