"""volren_amd -- MI355X-native (gfx950, HIP) drop-in for the offline path-tracing path of nihofm/volren.

The compute lives in libvolren_amd.so (hand-written HIP kernels + C++ host classes, C ABI in include/volren_amd.h);
this package is the thin Python binding used by the tests, bench.py and multi-GPU sharding.  No CPU fallback.
"""
from ._lib import LIB_PATH, SYMBOLS, VolrenError, load  # noqa: F401
from .renderer import Renderer, ShardedRenderer, math_probe  # noqa: F401
