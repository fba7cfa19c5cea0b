"""MI355X-native image-based-lighting precompute (the `bin/cube` hot path of
TheDevelo/vulkan-renderer): HIP kernels + C ABI in ``csrc/``, the C++ host tool in
``host/``, and this thin ctypes layer used by the tests and the benchmark."""
from .capi import (  # noqa: F401
    IblError,
    LIB_PATH,
    Ibl,
    build,
    exported_symbols,
    load_library,
)
from . import synth  # noqa: F401
