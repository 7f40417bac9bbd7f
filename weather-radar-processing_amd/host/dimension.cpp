// dimension.cpp -- row-major index maps (reference semantics: dimension.cpp:9-21).
#include "dimension.h"

Dimension3::Dimension3(int w, int h, int d) : width(w), height(h), depth(d), m_size(w * h), total_size(w * h * d) {}

int Dimension3::at_depth(int x, int y, int d) { return x + width * (y + height * d); }

Dimension4::Dimension4(int w, int h, int c, int d)
    : width(w), height(h), copies(c), depth(d), m_size(w * h), total_size(w * h * c * d) {}

int Dimension4::copy_at_depth(int x, int y, int copy, int d) { return x + width * (y + height * (copy + copies * d)); }
