"""fabstir-vectordb_amd — MI355X-native engine for fabstir-vectordb's hybrid HNSW/IVF
distance-computation hot path.

Layout: csrc/ (HIP kernels + the C ABI of include/fvdb.h), lib/ (built libfvdb_hip.so),
engine.py (ctypes view of the C ABI), host/ (C++ mirror of the reference's IVFIndex /
HNSWIndex / HybridIndex for this path) and index.py (its Python surface).

The directory name carries a hyphen, so import it through `fvdb_import.py` at the repo root
(`import fvdb_import; pkg = fvdb_import.load()`), which registers it as `fabstir_vectordb_amd`.
"""
from . import _capi  # noqa: F401
from . import engine  # noqa: F401
from .engine import *  # noqa: F401,F403
from .engine import Context, DeviceIVF, RowStore  # noqa: F401
from .index import IVFIndex, HNSWIndex, HybridIndex, SearchResults, load_host  # noqa: F401,E402
from . import sharded  # noqa: F401,E402
from . import metadata_filter  # noqa: F401,E402
from . import session  # noqa: F401,E402
from .session import VectorDbSession, VectorId, blake3, rest_search, rest_insert_vector, rest_batch_insert  # noqa: F401,E402
from . import chunked  # noqa: F401,E402
