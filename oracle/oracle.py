"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference algorithm).

TEST INFRASTRUCTURE ONLY — see oracle/oracle.cpp.  Class and method names follow the
reference (IVFIndex, HNSWIndex, HybridIndex; src/ivf/core.rs, src/hnsw/core.rs,
src/hybrid/core.rs) so that tests read like the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

__all__ = [
    "build", "lib", "OracleError", "NotTrained", "DuplicateVector", "DimensionMismatch",
    "InsufficientTrainingData", "VectorNotFound", "NotInitialized", "InvalidConfig",
    "euclidean_distance_scalar", "dot_product_scalar", "cosine_similarity_scalar", "l2_batch",
    "top_k_indices", "top_k_indices_heap", "streaming_top_k", "merge_search_results", "rng_levels",
    "IVFIndex", "HNSWIndex", "HybridIndex",
]


class OracleError(Exception):
    pass


class NotTrained(OracleError):
    pass


class DuplicateVector(OracleError):
    pass


class DimensionMismatch(OracleError):
    pass


class InsufficientTrainingData(OracleError):
    pass


class InconsistentDimensions(OracleError):
    pass


class InvalidConfig(OracleError):
    pass


class VectorNotFound(OracleError):
    pass


class NotInitialized(OracleError):
    pass


_ERR = {1: NotTrained, 2: DuplicateVector, 3: DimensionMismatch, 4: InsufficientTrainingData,
        5: InconsistentDimensions, 6: InvalidConfig, 7: VectorNotFound, 8: NotInitialized}


def _check(rc):
    if rc != 0:
        raise _ERR.get(rc, OracleError)(f"oracle rc={rc}")


def build(force=False):
    """Compile oracle/liboracle.so with gcc (no-op when up to date)."""
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _LIB_PATH


_lib = None
_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)


def _p(a, t):
    return a.ctypes.data_as(t)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, u64, u32, i64, f32, dbl, ci = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int64, C.c_float, C.c_double, C.c_int
    sig = {
        "orc_l2": (f32, [_f32p, _f32p, u64]),
        "orc_dot": (f32, [_f32p, _f32p, u64]),
        "orc_cosine": (f32, [_f32p, _f32p, u64]),
        "orc_l2_batch": (None, [_f32p, _f32p, u64, u64, _f32p]),
        "orc_top_k_indices": (None, [_f32p, u64, u64, _u64p, _u64p]),
        "orc_top_k_indices_heap": (None, [_f32p, u64, u64, _u64p, _u64p]),
        "orc_merge_search_results": (None, [_u64p, _f32p, u64, u64, _u64p, _f32p, _u64p]),
        "orc_streaming_top_k": (None, [_u64p, _f32p, u64, u64, _u64p, _f32p, _u64p]),
        "orc_rng_levels": (None, [u64, u64, _i64p]),
        "orc_ivf_new": (vp, [u64, u64, u64, u64]),
        "orc_ivf_free": (None, [vp]),
        "orc_ivf_train": (ci, [vp, _f32p, u64, u64, _u32p, C.POINTER(ci), _f32p, _f32p]),
        "orc_ivf_set_trained": (ci, [vp, _f32p, u64]),
        "orc_ivf_get_centroids": (ci, [vp, _f32p]),
        "orc_ivf_insert": (ci, [vp, u64, _f32p, u64]),
        "orc_ivf_insert_batch": (ci, [vp, _u64p, _f32p, u64, u64]),
        "orc_ivf_insert_assigned_batch": (ci, [vp, _u64p, _f32p, u64, u64, _u32p]),
        "orc_ivf_find_cluster": (ci, [vp, _f32p, u64, _u64p]),
        "orc_ivf_assign_batch": (ci, [vp, _f32p, u64, u64, _u32p]),
        "orc_ivf_cluster_size": (u64, [vp, u64]),
        "orc_ivf_total_vectors": (u64, [vp]),
        "orc_ivf_list_ids": (None, [vp, u64, _u64p]),
        "orc_ivf_search": (ci, [vp, _f32p, u64, u64, u64, _u64p, _f32p, _u32p]),
        "orc_ivf_batch_search": (ci, [vp, _f32p, u64, u64, u64, u64, _u64p, _f32p, _u32p, u32]),
        "orc_ivf_mark_deleted": (ci, [vp, u64]),
        "orc_ivf_vacuum": (u64, [vp]),
        "orc_hnsw_new": (vp, [u64, u64, u64, u64]),
        "orc_hnsw_free": (None, [vp]),
        "orc_hnsw_insert": (ci, [vp, u64, _f32p, u64, i64]),
        "orc_hnsw_insert_batch": (ci, [vp, _u64p, _f32p, u64, u64, _i64p]),
        "orc_hnsw_search": (ci, [vp, _f32p, u64, u64, u64, _u64p, _f32p, _u32p]),
        "orc_hnsw_batch_search": (ci, [vp, _f32p, u64, u64, u64, u64, _u64p, _f32p, _u32p]),
        "orc_hnsw_node_count": (u64, [vp]),
        "orc_hnsw_entry_point": (ci, [vp, _u64p]),
        "orc_hnsw_level": (i64, [vp, u64]),
        "orc_hnsw_neighbors": (i64, [vp, u64, u64, _u64p, u64]),
        "orc_hnsw_mark_deleted": (ci, [vp, u64]),
        "orc_hnsw_vacuum": (u64, [vp]),
        "orc_hnsw_dist_evals": (u64, [vp]),
        "orc_hnsw_restore": (ci, [vp, _u64p, _f32p, u64, u64, _u32p, _u64p, _u64p, u64]),
        "orc_hybrid_new": (vp, [dbl, u64, ci, u64, u64, u64, u64, u64, u64, u64, u64, u64]),
        "orc_hybrid_free": (None, [vp]),
        "orc_hybrid_initialize": (ci, [vp, _f32p, u64, u64]),
        "orc_hybrid_set_ivf_centroids": (ci, [vp, _f32p, u64]),
        "orc_hybrid_get_ivf_centroids": (ci, [vp, _f32p]),
        "orc_hybrid_insert": (ci, [vp, u64, _f32p, u64, dbl, dbl, i64]),
        "orc_hybrid_search": (ci, [vp, _f32p, u64, u64, u64, u64, ci, ci, u64, u64, dbl, _u64p, _f32p, _u32p]),
        "orc_hybrid_batch_search": (ci, [vp, _f32p, u64, u64, u64, u64, u64, dbl, _u64p, _f32p, _u32p, u32]),
        "orc_hybrid_delete": (ci, [vp, u64, dbl]),
        "orc_hybrid_migrate": (u64, [vp, dbl, dbl]),
        "orc_hybrid_recent_count": (u64, [vp]),
        "orc_hybrid_historical_count": (u64, [vp]),
        "orc_hybrid_is_ivf_trained": (ci, [vp]),
        "orc_hybrid_hnsw": (vp, [vp]),
        "orc_hybrid_ivf": (vp, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- free functions: src/core/vector_ops.rs -------------------------------------------
def euclidean_distance_scalar(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_l2(_p(a, _f32p), _p(b, _f32p), min(a.size, b.size)))


def dot_product_scalar(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot(_p(a, _f32p), _p(b, _f32p), min(a.size, b.size)))


def cosine_similarity_scalar(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_cosine(_p(a, _f32p), _p(b, _f32p), min(a.size, b.size)))


def l2_batch(q, rows):
    q, rows = _f32(q), _f32(rows)
    n, d = rows.shape
    out = np.empty(n, np.float32)
    lib().orc_l2_batch(_p(q, _f32p), _p(rows, _f32p), n, d, _p(out, _f32p))
    return out


def _topk(fn, scores, k):
    s = _f32(scores)
    out = np.empty(max(min(k, s.size), 1), np.uint64)
    n = C.c_uint64(0)
    fn(_p(s, _f32p), s.size, k, _p(out, _u64p), C.byref(n))
    return [int(x) for x in out[: n.value]]


def top_k_indices(scores, k):
    return _topk(lib().orc_top_k_indices, scores, k)


def top_k_indices_heap(scores, k):
    return _topk(lib().orc_top_k_indices_heap, scores, k)


def streaming_top_k(ids, scores, k):
    """StreamingTopK::add for every (id, score) in order, then get_results(): [(id, score)] by descending score."""
    ids = np.ascontiguousarray(ids, np.uint64)
    sc = _f32(scores)
    oi = np.empty(max(min(k, ids.size), 1), np.uint64)
    od = np.empty(max(min(k, ids.size), 1), np.float32)
    n = C.c_uint64(0)
    lib().orc_streaming_top_k(_p(ids, _u64p), _p(sc, _f32p), ids.size, k, _p(oi, _u64p), _p(od, _f32p), C.byref(n))
    return [(int(oi[i]), float(od[i])) for i in range(n.value)]


def merge_search_results(result_sets, k):
    """result_sets: list of lists of (id, distance)."""
    ids = np.array([r[0] for rs in result_sets for r in rs], np.uint64)
    ds = np.array([r[1] for rs in result_sets for r in rs], np.float32)
    oi = np.empty(max(ids.size, 1), np.uint64)
    od = np.empty(max(ids.size, 1), np.float32)
    n = C.c_uint64(0)
    lib().orc_merge_search_results(_p(ids, _u64p), _p(ds, _f32p), ids.size, k, _p(oi, _u64p), _p(od, _f32p), C.byref(n))
    return [(int(oi[i]), float(od[i])) for i in range(n.value)]


def rng_levels(seed, n):
    out = np.empty(n, np.int64)
    lib().orc_rng_levels(seed, n, _p(out, _i64p))
    return out


class _Results:
    """ids/distances/count triple for one query (SearchResult list in the reference)."""

    def __init__(self, ids, dist):
        self.ids = ids
        self.distances = dist

    def __len__(self):
        return len(self.ids)


def _search1(fn, handle, q, k, *mid):
    q = _f32(q)
    ids = np.empty(max(k, 1), np.uint64)
    ds = np.empty(max(k, 1), np.float32)
    cnt = C.c_uint32(0)
    _check(fn(handle, _p(q, _f32p), q.size, k, *mid, _p(ids, _u64p), _p(ds, _f32p), C.byref(cnt)))
    return _Results(ids[: cnt.value].copy(), ds[: cnt.value].copy())


class IVFIndex:
    """src/ivf/core.rs IVFIndex (IVFConfig defaults :50-60)."""

    def __init__(self, n_clusters=256, n_probe=16, train_size=10000, max_iterations=25, seed=0, _handle=None):
        self.n_clusters, self.n_probe, self.max_iterations = n_clusters, n_probe, max_iterations
        self._own = _handle is None
        if _handle is None:
            _handle = lib().orc_ivf_new(n_clusters, n_probe, max_iterations, seed)
            if not _handle:
                raise InvalidConfig("Invalid IVFConfig")
        self._h = _handle
        self.dimension = None

    def __del__(self):
        if getattr(self, "_own", False) and self._h:
            lib().orc_ivf_free(self._h)
            self._h = None

    def train(self, data):
        rows = [np.asarray(r, np.float32) for r in data]
        if len(rows) == 0 or len(rows) < self.n_clusters:
            raise InsufficientTrainingData(f"got {len(rows)}, need {self.n_clusters}")
        d = rows[0].size
        if any(r.size != d for r in rows):
            raise InconsistentDimensions()
        x = _f32(np.stack(rows))
        it, conv, e0, e1 = C.c_uint32(0), C.c_int(0), C.c_float(0), C.c_float(0)
        _check(lib().orc_ivf_train(self._h, _p(x, _f32p), x.shape[0], d, C.byref(it), C.byref(conv), C.byref(e0), C.byref(e1)))
        self.dimension = d
        return dict(iterations=it.value, converged=bool(conv.value), initial_error=e0.value, final_error=e1.value)

    def set_trained(self, centroids):
        c = _f32(centroids)
        assert c.shape[0] == self.n_clusters
        self.dimension = c.shape[1]
        _check(lib().orc_ivf_set_trained(self._h, _p(c, _f32p), c.shape[1]))

    def get_centroids(self):
        out = np.empty((self.n_clusters, self.dimension), np.float32)
        _check(lib().orc_ivf_get_centroids(self._h, _p(out, _f32p)))
        return out

    def insert(self, id, vector):
        v = _f32(vector)
        _check(lib().orc_ivf_insert(self._h, int(id), _p(v, _f32p), v.size))

    def batch_insert(self, ids, vectors):
        ids = np.ascontiguousarray(ids, np.uint64)
        v = _f32(vectors)
        _check(lib().orc_ivf_insert_batch(self._h, _p(ids, _u64p), _p(v, _f32p), v.shape[0], v.shape[1]))

    def batch_insert_assigned(self, ids, vectors, clusters):
        """Setup helper: insert with clusters already known (see oracle.cpp insert_assigned)."""
        ids = np.ascontiguousarray(ids, np.uint64)
        v = _f32(vectors)
        cl = np.ascontiguousarray(clusters, np.uint32)
        _check(lib().orc_ivf_insert_assigned_batch(self._h, _p(ids, _u64p), _p(v, _f32p), v.shape[0], v.shape[1], _p(cl, _u32p)))

    def find_cluster(self, vector):
        v = _f32(vector)
        out = C.c_uint64(0)
        _check(lib().orc_ivf_find_cluster(self._h, _p(v, _f32p), v.size, C.byref(out)))
        return out.value

    def assign(self, vectors):
        v = _f32(vectors)
        out = np.empty(v.shape[0], np.uint32)
        _check(lib().orc_ivf_assign_batch(self._h, _p(v, _f32p), v.shape[0], v.shape[1], _p(out, _u32p)))
        return out

    def get_cluster_size(self, c):
        return int(lib().orc_ivf_cluster_size(self._h, c))

    def list_ids(self, c):
        out = np.empty(self.get_cluster_size(c), np.uint64)
        if out.size:
            lib().orc_ivf_list_ids(self._h, c, _p(out, _u64p))
        return out

    def total_vectors(self):
        return int(lib().orc_ivf_total_vectors(self._h))

    def search(self, query, k, n_probe=None):
        return _search1(lib().orc_ivf_search, self._h, query, k, self.n_probe if n_probe is None else n_probe)

    search_with_config = search

    def batch_search(self, queries, k, n_probe=None, threads=1):
        q = _f32(queries)
        nq, d = q.shape
        ids = np.zeros((nq, max(k, 1)), np.uint64)
        ds = np.full((nq, max(k, 1)), np.inf, np.float32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().orc_ivf_batch_search(self._h, _p(q, _f32p), nq, d, k, self.n_probe if n_probe is None else n_probe,
                                          _p(ids, _u64p), _p(ds, _f32p), _p(cnt, _u32p), threads))
        return ids, ds, cnt

    def mark_deleted(self, id):
        _check(lib().orc_ivf_mark_deleted(self._h, int(id)))

    def vacuum(self):
        return int(lib().orc_ivf_vacuum(self._h))


class HNSWIndex:
    """src/hnsw/core.rs HNSWIndex (HNSWConfig defaults :37-46)."""

    def __init__(self, max_connections=16, max_connections_layer_0=32, ef_construction=200, seed=0, _handle=None):
        self._own = _handle is None
        if _handle is None:
            _handle = lib().orc_hnsw_new(max_connections, max_connections_layer_0, ef_construction, seed)
        self._h = _handle

    def __del__(self):
        if getattr(self, "_own", False) and self._h:
            lib().orc_hnsw_free(self._h)
            self._h = None

    def insert(self, id, vector, level=-1):
        v = _f32(vector)
        _check(lib().orc_hnsw_insert(self._h, int(id), _p(v, _f32p), v.size, level))

    def batch_insert(self, ids, vectors, levels=None):
        ids = np.ascontiguousarray(ids, np.uint64)
        v = _f32(vectors)
        lv = None if levels is None else np.ascontiguousarray(levels, np.int64)
        _check(lib().orc_hnsw_insert_batch(self._h, _p(ids, _u64p), _p(v, _f32p), v.shape[0], v.shape[1],
                                           None if lv is None else _p(lv, _i64p)))

    def search(self, query, k, ef):
        return _search1(lib().orc_hnsw_search, self._h, query, k, ef)

    def batch_search(self, queries, k, ef):
        q = _f32(queries)
        nq, d = q.shape
        ids = np.zeros((nq, max(k, 1)), np.uint64)
        ds = np.full((nq, max(k, 1)), np.inf, np.float32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().orc_hnsw_batch_search(self._h, _p(q, _f32p), nq, d, k, ef, _p(ids, _u64p), _p(ds, _f32p), _p(cnt, _u32p)))
        return ids, ds, cnt

    def node_count(self):
        return int(lib().orc_hnsw_node_count(self._h))

    def entry_point(self):
        out = C.c_uint64(0)
        rc = lib().orc_hnsw_entry_point(self._h, C.byref(out))
        return None if rc else out.value

    def level(self, id):
        return int(lib().orc_hnsw_level(self._h, int(id)))

    def neighbors(self, id, layer):
        buf = np.empty(4096, np.uint64)
        n = lib().orc_hnsw_neighbors(self._h, int(id), layer, _p(buf, _u64p), buf.size)
        if n < 0:
            raise VectorNotFound(id)
        return [int(x) for x in buf[:n]]

    def mark_deleted(self, id):
        _check(lib().orc_hnsw_mark_deleted(self._h, int(id)))

    def vacuum(self):
        return int(lib().orc_hnsw_vacuum(self._h))

    def dist_evals(self):
        return int(lib().orc_hnsw_dist_evals(self._h))

    def restore(self, ids, vectors, levels, nbr_offsets, nbrs, entry):
        ids = np.ascontiguousarray(ids, np.uint64)
        v = _f32(vectors)
        lv = np.ascontiguousarray(levels, np.uint32)
        off = np.ascontiguousarray(nbr_offsets, np.uint64)
        nb = np.ascontiguousarray(nbrs, np.uint64)
        _check(lib().orc_hnsw_restore(self._h, _p(ids, _u64p), _p(v, _f32p), v.shape[0], v.shape[1], _p(lv, _u32p),
                                      _p(off, _u64p), _p(nb, _u64p), int(entry)))


class HybridIndex:
    """src/hybrid/core.rs HybridIndex.  `now`/`timestamp` are seconds (the reference reads Utc::now())."""

    WEEK = 7 * 24 * 3600.0

    def __init__(self, recent_threshold=WEEK, migration_batch_size=100, auto_migrate=True, min_ivf_training_size=10,
                 max_connections=16, max_connections_layer_0=32, ef_construction=200, hnsw_seed=0,
                 n_clusters=3, n_probe=2, max_iterations=25, ivf_seed=0):
        # HybridConfig::default (src/hybrid/core.rs:69-85): IVF 3 clusters / n_probe 2
        self.n_clusters, self.n_probe = n_clusters, n_probe
        self._h = lib().orc_hybrid_new(recent_threshold, migration_batch_size, int(auto_migrate), min_ivf_training_size,
                                       max_connections, max_connections_layer_0, ef_construction, hnsw_seed,
                                       n_clusters, n_probe, max_iterations, ivf_seed)
        if not self._h:
            raise InvalidConfig()
        self.dimension = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_hybrid_free(self._h)
            self._h = None

    def initialize(self, training_data):
        x = _f32(training_data)
        if x.ndim != 2:
            x = x.reshape(len(training_data), -1)
        self.dimension = x.shape[1] if x.size else None
        _check(lib().orc_hybrid_initialize(self._h, _p(x, _f32p), x.shape[0], x.shape[1] if x.ndim == 2 else 0))

    def set_ivf_centroids(self, centroids):
        c = _f32(centroids)
        self.dimension = c.shape[1]
        _check(lib().orc_hybrid_set_ivf_centroids(self._h, _p(c, _f32p), c.shape[1]))

    def get_ivf_centroids(self):
        out = np.empty((self.n_clusters, self.dimension), np.float32)
        _check(lib().orc_hybrid_get_ivf_centroids(self._h, _p(out, _f32p)))
        return out

    def insert_with_timestamp(self, id, vector, timestamp, now, level=-1):
        v = _f32(vector)
        _check(lib().orc_hybrid_insert(self._h, int(id), _p(v, _f32p), v.size, float(timestamp), float(now), level))

    def insert(self, id, vector, now=0.0, level=-1):
        self.insert_with_timestamp(id, vector, now, now, level)

    def search(self, query, k, now=0.0, hnsw_ef=50, ivf_n_probe=10, search_recent=True, search_historical=True,
               recent_k=0, historical_k=0):
        # HybridSearchConfig::default (src/hybrid/core.rs:184-197): ef 50, n_probe 10
        return _search1(lib().orc_hybrid_search, self._h, query, k, hnsw_ef, ivf_n_probe, int(search_recent),
                        int(search_historical), recent_k, historical_k, float(now))

    search_with_config = search

    def batch_search(self, queries, k, now=0.0, hnsw_ef=50, ivf_n_probe=10, threads=1):
        q = _f32(queries)
        nq, d = q.shape
        ids = np.full((nq, max(k, 1)), 2**64 - 1, np.uint64)
        ds = np.full((nq, max(k, 1)), np.inf, np.float32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().orc_hybrid_batch_search(self._h, _p(q, _f32p), nq, d, k, hnsw_ef, ivf_n_probe, float(now),
                                             _p(ids, _u64p), _p(ds, _f32p), _p(cnt, _u32p), threads))
        return ids, ds, cnt

    def delete(self, id, now=0.0):
        _check(lib().orc_hybrid_delete(self._h, int(id), float(now)))

    def migrate_with_threshold(self, threshold, now):
        return int(lib().orc_hybrid_migrate(self._h, float(threshold), float(now)))

    def recent_count(self):
        return int(lib().orc_hybrid_recent_count(self._h))

    def historical_count(self):
        return int(lib().orc_hybrid_historical_count(self._h))

    def is_ivf_trained(self):
        return bool(lib().orc_hybrid_is_ivf_trained(self._h))

    def hnsw(self):
        return HNSWIndex(_handle=lib().orc_hybrid_hnsw(self._h))

    def ivf(self):
        ix = IVFIndex(n_clusters=self.n_clusters, n_probe=self.n_probe, _handle=lib().orc_hybrid_ivf(self._h))
        ix.dimension = self.dimension
        return ix
