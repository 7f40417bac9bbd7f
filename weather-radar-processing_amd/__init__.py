"""weather-radar-processing_amd -- MI355X-native per-sector weather-radar DSP engine.

Python is only a harness here: the product is ``lib/libwrp.so`` (hand-written gfx950 HIP
kernels behind the C ABI of ``include/wrp.h``) plus the C++ host types in ``host/``.  This
module binds the C ABI with ctypes for tests, ``bench.py`` and ``__graft_entry__``.

The directory name is not a valid Python identifier; import it as ``wrp_amd`` (the loader
module at the repository root).  There is deliberately NO CPU fallback: every compute entry
needs libwrp.so and a GPU and raises :class:`WrpError` otherwise.
"""
from .binding import (  # noqa: F401
    Engine,
    FLAG_DEBUG_FUSED_UNDERSIZED,
    FLAG_GENERIC_KERNELS,
    FLAG_ONE_TILE_PER_BLOCK,
    FLAG_TWO_KERNELS,
    FLAG_WIRE_8,
    FUSED_MIN_SECTORS,
    WrpConfig,
    WrpError,
    STAGE_IDS,
    STAGE_SHAPES,
    exported_symbols,
    header_symbols,
    lib_path,
    load_library,
    source_fingerprint,
)
