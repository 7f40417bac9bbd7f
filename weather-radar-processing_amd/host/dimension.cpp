// dimension.cpp -- the index maps are header-only (dimension.h); this translation unit keeps
// the dimension.o that `make all` lists (reference Makefile:2) and checks the two layouts the
// engine relies on at compile time.
#include "dimension.h"

namespace {
constexpr int offset4(int w, int h, int c, int x, int y, int plane, int slot)
{
    return x + w * (y + h * (plane + c * slot));
}
// the production IQ block idim(512, 1024, 3, streams) and result table sitdim(2, 512, 143, 9)
static_assert(offset4(512, 1024, 3, 5, 7, 1, 1) == 5 + 7 * 512 + 1 * 512 * 1024 + 1 * 512 * 1024 * 3, "idim");
static_assert(offset4(2, 512, 143, 1, 10, 3, 2) == 1 + 10 * 2 + 3 * 1024 + 2 * 1024 * 143, "sitdim");
}   // namespace

// run-time self check used by tests of other languages' bindings
extern "C" int wrph_dimension_selfcheck()
{
    Dimension4 idim(512, 1024, 3, 2);
    Dimension3 d3(5, 4, 3);
    return idim.copy_at_depth(5, 7, 1, 1) == offset4(512, 1024, 3, 5, 7, 1, 1) && d3.at_depth(2, 3, 1) == 2 + 3 * 5 + 20 &&
           idim.m_size == 512 * 1024 && idim.total_size == 512 * 1024 * 3 * 2;
}
