"""vectordb_amd — MI355X-native batched two-phase ANN query path of CuddleDB (LLay/vectordb).

Only what the path needs: the HIP library (csrc/ -> libcuddle_hip.so, C ABI in
include/cuddle_hip.h), its ctypes binding, the flat index form + builder, and the host-side
mirror of the reference's `ClusteredIndex` search API.
"""
from ._ffi import CuddleError, METRIC_L2, METRIC_COSINE, METRIC_DOT, CODE_BINARY, CODE_RABITQ  # noqa: F401
from .flat_index import FlatIndex, pack_binary_codes  # noqa: F401
from .index import ClusteredIndex, ClusteredIndexWithRaBitQ, DeviceIndex, SearchStats  # noqa: F401

from .pyvectordb import PyVectorDB  # noqa: E402

__all__ = ["ClusteredIndex", "ClusteredIndexWithRaBitQ", "DeviceIndex", "FlatIndex", "SearchStats", "CuddleError", "PyVectorDB"]
