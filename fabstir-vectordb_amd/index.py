"""Python surface of the host mirror (lib/libfvdb_host.so): IVFIndex, HNSWIndex, HybridIndex.

Names, argument meaning and error behaviour follow the reference (src/ivf/core.rs,
src/hnsw/core.rs, src/hybrid/core.rs); vector ids are u64 row ids (the 32-byte BLAKE3
VectorId of src/core/types.rs:9-22 is a host-side mapping outside the hot path).  All
search entry points take a BATCH of queries: the reference's per-query loop
(src/ivf/operations.rs:132-145) is what the GPU batch replaces.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import f32p, u32p, u64p
from .engine import (STATUS_TO_EXC, DimensionMismatch, FvdbError, InconsistentDimensions,
                     InsufficientTrainingData, InvalidConfig, _f32, _ptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_capi.LIB_DIR, "libfvdb_host.so")
_host = None
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
vp, u64, u32, i32, i64, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int64, C.c_double

HOST_SIGNATURES = {
    "fvh_ivf_new": (vp, [vp, u32, u32, u32, u32, u64]),
    "fvh_ivf_free": (None, [vp]),
    "fvh_ivf_train": (i32, [vp, f32p, u64, u32, C.POINTER(_capi.TrainResult)]),
    "fvh_ivf_set_trained": (i32, [vp, f32p, u32]),
    "fvh_ivf_get_centroids": (i32, [vp, f32p]),
    "fvh_ivf_is_trained": (i32, [vp]),
    "fvh_ivf_dimension": (u32, [vp]),
    "fvh_ivf_total_vectors": (u64, [vp]),
    "fvh_ivf_active_count": (u64, [vp]),
    "fvh_ivf_cluster_size": (u64, [vp, u32]),
    "fvh_ivf_export_list": (i32, [vp, u32, f32p, u64p, C.POINTER(C.c_uint8)]),
    "fvh_ivf_insert": (i32, [vp, u64, f32p, u32]),
    "fvh_ivf_batch_insert": (i32, [vp, u64p, f32p, u64, u32, u64p, C.POINTER(i32)]),
    "fvh_ivf_find_cluster": (i32, [vp, f32p, u32, u32p]),
    "fvh_ivf_search": (i32, [vp, f32p, u32, u32, u32, u32, u64p, f32p, u32p]),
    "fvh_ivf_mark_deleted": (i32, [vp, u64]),
    "fvh_ivf_is_deleted": (i32, [vp, u64]),
    "fvh_ivf_device": (vp, [vp]),
    "fvh_hnsw_new": (vp, [vp, u32, u32, u32, u64]),
    "fvh_hnsw_free": (None, [vp]),
    "fvh_hnsw_insert": (i32, [vp, u64, f32p, u32, i64]),
    "fvh_hnsw_batch_insert": (i32, [vp, u64p, f32p, u64, u32, i64p, u64p, C.POINTER(i32)]),
    "fvh_hnsw_search": (i32, [vp, f32p, u32, u32, u32, u32, u64p, f32p, u32p]),
    "fvh_hnsw_node_count": (u64, [vp]),
    "fvh_hnsw_active_count": (u64, [vp]),
    "fvh_hnsw_entry_point": (i32, [vp, u64p]),
    "fvh_hnsw_level": (i64, [vp, u64]),
    "fvh_hnsw_neighbors": (i64, [vp, u64, u32, u64p, u64]),
    "fvh_hnsw_mark_deleted": (i32, [vp, u64]),
    "fvh_hnsw_is_deleted": (i32, [vp, u64]),
    "fvh_hnsw_bulk_build": (i32, [vp, u64p, f32p, u64, u32, i64p]),
    "fvh_hnsw_restore": (i32, [vp, u64p, f32p, u64, u32, u32p, u64p, u64p, u64]),
    "fvh_hnsw_graph_slots": (u64, [vp]),
    "fvh_hnsw_graph_edges": (u64, [vp]),
    "fvh_hnsw_export_graph": (None, [vp, u64p, u32p, u64p, u64p]),
    "fvh_hnsw_get_vector": (i32, [vp, u64, f32p]),
    "fvh_hnsw_dist_evals": (u64, [vp]),
    "fvh_hnsw_hops": (u64, [vp]),
    "fvh_hnsw_set_threads": (None, [vp, i32]),
    "fvh_hnsw_dimension": (u32, [vp]),
    "fvh_hnsw_set_device_traversal": (None, [vp, i32]),
    "fvh_hnsw_set_device_insert": (None, [vp, i32, i32]),
    "fvh_hnsw_device_insert": (i32, [vp]),
    "fvh_hnsw_insert_stats": (None, [vp, vp, u64p, u64p]),
    "fvh_hnsw_device_traversal": (i32, [vp]),
    "fvh_hnsw_device_fallbacks": (u64, [vp]),
    "fvh_hnsw_graph_kernel_times": (i32, [vp, f32p, u32p, u64p, u64p]),
    "fvh_hnsw_tie_restarts": (i32, [vp, u64p, u64p]),
    "fvh_hybrid_new": (vp, [vp, vp, dbl, u64, i32, u64, u32, u32, u32, u64, u32, u32, u32, u32, u64]),
    "fvh_hybrid_free": (None, [vp]),
    "fvh_hybrid_initialize": (i32, [vp, f32p, u64, u32]),
    "fvh_hybrid_set_ivf_centroids": (i32, [vp, f32p, u32]),
    "fvh_hybrid_insert": (i32, [vp, u64, f32p, u32, dbl, dbl, i64]),
    "fvh_hybrid_bulk_insert": (i32, [vp, u64p, f32p, u64, u32, f64p, dbl]),
    "fvh_hybrid_bulk_insert_sharded": (i32, [vp, u64p, f32p, u64, u32, f64p, dbl, u32, u32, u32p]),
    "fvh_hybrid_search": (i32, [vp, f32p, u32, u32, u64, u64, u64, i32, i32, u64, u64, dbl, u64p, f32p, u32p]),
    "fvh_hybrid_search_with_filter": (i32, [vp, f32p, u32, u32, u64, vp, vp, dbl, u64p, f32p, u32p]),
    "fvh_hybrid_search_dev": (i32, [vp, vp, u32, u32, u64, u64, u64, i32, i32, u64, u64, dbl, u64p, f32p, u32p]),
    "fvh_hybrid_search_dev_begin": (i32, [vp, u32, vp, u32, u32, u64, u64, u64, i32, i32, u64, u64, dbl]),
    "fvh_hybrid_search_dev_end": (i32, [vp, u32, u64p, f32p, u32p]),
    "fvh_hybrid_attach_comm": (i32, [vp, vp]),
    "fvh_hybrid_search_sharded_begin": (i32, [vp, u32, vp, u32, u32, u64, u64, u64, i32, i32, u64, u64, i32, dbl]),
    "fvh_hybrid_search_sharded_end": (i32, [vp, u32, u64p, f32p, u32p]),
    "fvh_hybrid_sharded_rows": (u32, [vp, u32, i32]),
    "fvh_plan_list_owners": (None, [u64p, u32, u32, u32p]),
    "fvh_merge_parts": (None, [u32, u32, u32, u32, u64p, f32p, u32p, u64p, f32p, u32p, u64p, f32p, u32p]),
    "fvh_hnsw_search_dev": (i32, [vp, vp, u32, u32, u32, u32, u64p, f32p, u32p]),
    "fvh_hnsw_search_dev_begin": (i32, [vp, u32, vp, u32, u32, u32, u32]),
    "fvh_hnsw_search_dev_end": (i32, [vp, u32, vp, u32, u32, u32, u32, u64p, f32p, u32p]),
    "fvh_hybrid_delete": (i32, [vp, u64, dbl]),
    "fvh_hybrid_migrate": (u64, [vp, dbl, dbl]),
    "fvh_hybrid_vacuum": (i32, [vp, u64p, u64p]),
    "fvh_ivf_vacuum": (i32, [vp, u64p]),
    "fvh_hnsw_vacuum": (u64, [vp]),
    "fvh_hybrid_from_parts": (i32, [vp, u64p, f64p, u64, u64, u64, i32]),
    "fvh_hybrid_timestamp_count": (u64, [vp]),
    "fvh_hybrid_export_timestamps": (None, [vp, u64p, f64p]),
    "fvh_hybrid_recent_count": (u64, [vp]),
    "fvh_hybrid_historical_count": (u64, [vp]),
    "fvh_hybrid_is_initialized": (i32, [vp]),
    "fvh_hybrid_is_ivf_trained": (i32, [vp]),
    "fvh_hybrid_hnsw": (vp, [vp]),
    "fvh_hybrid_set_sequential_graph": (None, [vp, i32]),
    "fvh_hybrid_sequential_graph": (i32, [vp]),
    "fvh_hybrid_recent_build_seconds": (dbl, [vp]),
    "fvh_hybrid_set_blocking_writers": (None, [vp, i32]),
    "fvh_hybrid_blocking_writers": (i32, [vp]),
    "fvh_hybrid_ivf": (vp, [vp]),
}


def load_host():
    global _host
    if _host is not None:
        return _host
    _capi.load()  # the engine first: the host mirror links against it
    if not os.path.exists(HOST_LIB_PATH):
        raise ImportError(f"{HOST_LIB_PATH} not found: build it with `make -C fabstir-vectordb_amd`")
    lib = C.CDLL(HOST_LIB_PATH)
    for name, (res, args) in HOST_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _host = lib
    return lib


class SearchResults:
    """B x k ids / distances with per-query hit counts (Vec<Vec<SearchResult>> in the reference)."""

    def __init__(self, ids, distances, counts):
        self.ids, self.distances, self.counts = ids, distances, counts

    def __len__(self):
        return self.ids.shape[0]

    def __getitem__(self, b):
        n = int(self.counts[b])
        return self.ids[b, :n], self.distances[b, :n]

    def scores(self):
        """Node/REST surface score = 1/(1+distance) in f32 (bindings/node/src/session.rs:291, src/api/rest.rs:653)."""
        return (np.float32(1.0) / (np.float32(1.0) + self.distances)).astype(np.float32)


def _rows(x, dim=None):
    x = _f32(x)
    if x.ndim == 1:
        x = x.reshape(1, -1)
    if dim is not None and x.shape[1] != dim:
        raise DimensionMismatch(f"Dimension mismatch: expected {dim}, got {x.shape[1]}")
    return x


class _Base:
    def _check(self, rc):
        if rc:
            msg = self.ctx.lib.fvdb_last_error(self.ctx.h)
            exc = STATUS_TO_EXC.get(rc, FvdbError)((msg.decode() if msg else "") or f"status {rc}")
            exc.status = rc
            raise exc

    def _search(self, fn, q, k, *mid):
        q = _rows(q)
        B = q.shape[0]
        ids = np.empty((B, max(k, 1)), np.uint64)
        ds = np.empty((B, max(k, 1)), np.float32)
        cnt = np.zeros(B, np.uint32)
        self._check(fn(self.h, _ptr(q, f32p), B, q.shape[1], k, *mid, _ptr(ids, u64p), _ptr(ds, f32p), _ptr(cnt, u32p)))
        return SearchResults(ids, ds, cnt)


class IVFIndex(_Base):
    """src/ivf/core.rs IVFIndex (IVFConfig::default :50-60)."""

    def __init__(self, ctx, n_clusters=256, n_probe=16, train_size=10000, max_iterations=25, seed=0, _handle=None):
        self.ctx, self.lib = ctx, load_host()
        self.n_clusters, self.n_probe = n_clusters, n_probe
        self._own = _handle is None
        if _handle is None:
            _handle = self.lib.fvh_ivf_new(ctx.h, n_clusters, n_probe, train_size, max_iterations, seed)
            if not _handle:
                raise InvalidConfig("Invalid IVFConfig")
        self.h = _handle

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.lib.fvh_ivf_free(self.h)
            self.h = None

    def is_trained(self):
        return bool(self.lib.fvh_ivf_is_trained(self.h))

    def dimension(self):
        return int(self.lib.fvh_ivf_dimension(self.h)) if self.is_trained() else None

    def total_vectors(self):
        return int(self.lib.fvh_ivf_total_vectors(self.h))

    def active_count(self):
        return int(self.lib.fvh_ivf_active_count(self.h))

    def train(self, training_data):
        rows = [np.asarray(r, np.float32) for r in training_data]
        if len(rows) == 0 or len(rows) < self.n_clusters:  # src/ivf/core.rs:242-254
            raise InsufficientTrainingData(f"Insufficient training data: got {len(rows)}, need at least {self.n_clusters}")
        d = rows[0].size
        if any(r.size != d for r in rows):  # :257-265
            raise InconsistentDimensions("Inconsistent dimensions in training data")
        x = _f32(np.stack(rows))
        res = _capi.TrainResult()
        self._check(self.lib.fvh_ivf_train(self.h, _ptr(x, f32p), x.shape[0], d, C.byref(res)))
        return dict(iterations=res.iterations, converged=bool(res.converged), initial_error=res.initial_error,
                    final_error=res.final_error)

    def set_trained(self, centroids):
        c = _rows(centroids)
        if c.shape[0] != self.n_clusters:
            raise InvalidConfig(f"expected {self.n_clusters} centroids")
        self._check(self.lib.fvh_ivf_set_trained(self.h, _ptr(c, f32p), c.shape[1]))

    def get_centroids(self):
        out = np.empty((self.n_clusters, self.dimension()), np.float32)
        self._check(self.lib.fvh_ivf_get_centroids(self.h, _ptr(out, f32p)))
        return out

    def insert(self, id, vector):
        v = _f32(vector).reshape(-1)
        self._check(self.lib.fvh_ivf_insert(self.h, int(id), _ptr(v, f32p), v.size))

    def batch_insert(self, ids, vectors):
        """Returns (successful, failed) like BatchInsertResult (src/ivf/operations.rs:107-130)."""
        v = _rows(vectors)
        ids = np.ascontiguousarray(ids, np.uint64)
        ok, err = C.c_uint64(0), C.c_int(0)
        self._check(self.lib.fvh_ivf_batch_insert(self.h, _ptr(ids, u64p), _ptr(v, f32p), v.shape[0], v.shape[1],
                                                  C.byref(ok), C.byref(err)))
        return ok.value, v.shape[0] - ok.value

    def find_cluster(self, vector):
        v = _f32(vector).reshape(-1)
        out = C.c_uint32(0)
        self._check(self.lib.fvh_ivf_find_cluster(self.h, _ptr(v, f32p), v.size, C.byref(out)))
        return out.value

    def get_cluster_size(self, c):
        return int(self.lib.fvh_ivf_cluster_size(self.h, c))

    def vacuum(self):
        out = C.c_uint64(0)
        self._check(self.lib.fvh_ivf_vacuum(self.h, C.byref(out)))
        return out.value

    def export_list(self, c):
        """Rows (f32), ids and live flags of inverted list `c`, in list-position order (save path)."""
        n = self.get_cluster_size(c)
        rows, ids, live = np.empty((n, max(self.dimension(), 1)), np.float32), np.empty(n, np.uint64), np.empty(n, np.uint8)
        self._check(self.lib.fvh_ivf_export_list(self.h, int(c), _ptr(rows, f32p), _ptr(ids, u64p),
                                                 live.ctypes.data_as(C.POINTER(C.c_uint8))))
        return rows, ids, live.astype(bool)

    def search(self, queries, k, n_probe=None):
        return self._search(self.lib.fvh_ivf_search, queries, k, self.n_probe if n_probe is None else n_probe)

    search_with_config = search

    batch_search = search

    def mark_deleted(self, id):
        self._check(self.lib.fvh_ivf_mark_deleted(self.h, int(id)))

    def is_deleted(self, id):
        return bool(self.lib.fvh_ivf_is_deleted(self.h, int(id)))

    # --- views of the device index behind this mirror (accounting, bulk helpers) ---
    def _dev(self):
        return C.c_void_p(self.lib.fvh_ivf_device(self.h))

    def assign(self, vectors):
        """Batched find_nearest_centroid on the GPU (src/ivf/core.rs:373-386)."""
        v = _rows(vectors)
        out = np.empty(v.shape[0], np.uint32)
        self._check(self.ctx.lib.fvdb_ivf_assign(self._dev(), _ptr(v, f32p), v.shape[0], _ptr(out, u32p)))
        return out

    def list_sizes(self):
        out = np.empty(self.n_clusters, np.uint64)
        self._check(self.ctx.lib.fvdb_ivf_list_sizes(self._dev(), _ptr(out, u64p)))
        return out

    def stage_times(self):
        ms = np.zeros(8, np.float32)
        n = self.ctx.lib.fvdb_ivf_stage_times(self._dev(), _ptr(ms, f32p))
        return int(n), dict(zip(("coarse_scan", "coarse_merge", "plan", "fine_scan", "fine_merge", "mfma_filter_kernel"), ms.tolist()))

    def last_stats(self):
        st = _capi.SearchStats()
        self._check(self.ctx.lib.fvdb_ivf_last_stats(self._dev(), C.byref(st)))
        return dict(rows_scanned=st.rows_scanned, work_items=st.work_items, list_rows_touched=st.list_rows_touched)


class HNSWIndex(_Base):
    """src/hnsw/core.rs HNSWIndex (HNSWConfig::default :37-46)."""

    def __init__(self, ctx, max_connections=16, max_connections_layer_0=32, ef_construction=200, seed=0, _handle=None):
        self.ctx, self.lib = ctx, load_host()
        self._own = _handle is None
        if _handle is None:
            _handle = self.lib.fvh_hnsw_new(ctx.h, max_connections, max_connections_layer_0, ef_construction, seed)
        self.h = _handle

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.lib.fvh_hnsw_free(self.h)
            self.h = None

    def insert(self, id, vector, level=-1):
        v = _f32(vector).reshape(-1)
        self._check(self.lib.fvh_hnsw_insert(self.h, int(id), _ptr(v, f32p), v.size, level))

    def batch_insert(self, ids, vectors, levels=None):
        v = _rows(vectors)
        ids = np.ascontiguousarray(ids, np.uint64)
        lv = None if levels is None else np.ascontiguousarray(levels, np.int64)
        ok, err = C.c_uint64(0), C.c_int(0)
        self._check(self.lib.fvh_hnsw_batch_insert(self.h, _ptr(ids, u64p), _ptr(v, f32p), v.shape[0], v.shape[1],
                                                   None if lv is None else _ptr(lv, i64p), C.byref(ok), C.byref(err)))
        return ok.value, v.shape[0] - ok.value

    def bulk_build(self, ids, vectors, levels=None):
        v = _rows(vectors)
        ids = np.ascontiguousarray(ids, np.uint64)
        lv = None if levels is None else np.ascontiguousarray(levels, np.int64)
        self._check(self.lib.fvh_hnsw_bulk_build(self.h, _ptr(ids, u64p), _ptr(v, f32p), v.shape[0], v.shape[1],
                                                 None if lv is None else _ptr(lv, i64p)))

    def search(self, queries, k, ef):
        return self._search(self.lib.fvh_hnsw_search, queries, k, ef)

    def search_dev(self, q_dev, B, dim, k, ef):
        ids = np.empty((B, max(k, 1)), np.uint64)
        ds = np.empty((B, max(k, 1)), np.float32)
        cnt = np.zeros(B, np.uint32)
        self._check(self.lib.fvh_hnsw_search_dev(self.h, q_dev, B, dim, k, ef, _ptr(ids, u64p), _ptr(ds, f32p),
                                                 _ptr(cnt, u32p)))
        return SearchResults(ids, ds, cnt)

    def search_dev_begin(self, slot, q_dev, B, dim, k, ef):
        """Enqueue the device traversal of a batch in `slot` (0..7); the query buffer must stay valid until
        search_dev_end(slot)."""
        rc = self.lib.fvh_hnsw_search_dev_begin(self.h, slot, q_dev, B, dim, k, ef)
        if rc < 0:
            self._check(-rc)
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[slot] = (q_dev, B, dim, k, ef, rc == 1)

    def search_dev_end(self, slot):
        q_dev, B, dim, k, ef, started = self._inflight.pop(slot)
        if not started:  # not a device-path search (host-walk mode, empty index ...): run it now
            return self.search_dev(q_dev, B, dim, k, ef)
        ids = np.empty((B, max(k, 1)), np.uint64)
        ds = np.empty((B, max(k, 1)), np.float32)
        cnt = np.zeros(B, np.uint32)
        self._check(self.lib.fvh_hnsw_search_dev_end(self.h, slot, q_dev, B, dim, k, ef, _ptr(ids, u64p), _ptr(ds, f32p),
                                                     _ptr(cnt, u32p)))
        return SearchResults(ids, ds, cnt)

    def node_count(self):
        return int(self.lib.fvh_hnsw_node_count(self.h))

    def active_count(self):
        return int(self.lib.fvh_hnsw_active_count(self.h))

    def entry_point(self):
        out = C.c_uint64(0)
        return None if self.lib.fvh_hnsw_entry_point(self.h, C.byref(out)) else out.value

    def level(self, id):
        return int(self.lib.fvh_hnsw_level(self.h, int(id)))

    def neighbors(self, id, layer):
        buf = np.empty(1024, np.uint64)
        n = self.lib.fvh_hnsw_neighbors(self.h, int(id), layer, _ptr(buf, u64p), buf.size)
        if n < 0:
            raise STATUS_TO_EXC[7](f"Vector not found: {id}")
        return [int(x) for x in buf[:n]]

    def mark_deleted(self, id):
        self._check(self.lib.fvh_hnsw_mark_deleted(self.h, int(id)))

    def is_deleted(self, id):
        return bool(self.lib.fvh_hnsw_is_deleted(self.h, int(id)))

    def vacuum(self):
        return int(self.lib.fvh_hnsw_vacuum(self.h))

    def get_vector_by_id(self, id):
        out = np.empty(int(self.lib.fvh_hnsw_dimension(self.h)), np.float32)
        self._check(self.lib.fvh_hnsw_get_vector(self.h, int(id), _ptr(out, f32p)))
        return out

    def export_graph(self):
        n = int(self.lib.fvh_hnsw_node_count(self.h))
        slots, edges = int(self.lib.fvh_hnsw_graph_slots(self.h)), int(self.lib.fvh_hnsw_graph_edges(self.h))
        ids, lv = np.empty(n, np.uint64), np.empty(n, np.uint32)
        off, nb = np.empty(slots + 1, np.uint64), np.empty(max(edges, 1), np.uint64)
        self.lib.fvh_hnsw_export_graph(self.h, _ptr(ids, u64p), _ptr(lv, u32p), _ptr(off, u64p), _ptr(nb, u64p))
        return ids, lv, off, nb[:edges]

    def restore(self, ids, vectors, levels, nbr_offsets, nbrs, entry):
        v = _rows(vectors)
        ids = np.ascontiguousarray(ids, np.uint64)
        lv = np.ascontiguousarray(levels, np.uint32)
        off = np.ascontiguousarray(nbr_offsets, np.uint64)
        nb = np.ascontiguousarray(nbrs if len(nbrs) else [0], np.uint64)
        self._check(self.lib.fvh_hnsw_restore(self.h, _ptr(ids, u64p), _ptr(v, f32p), v.shape[0], v.shape[1],
                                              _ptr(lv, u32p), _ptr(off, u64p), _ptr(nb, u64p), int(entry)))

    def dist_evals(self):
        return int(self.lib.fvh_hnsw_dist_evals(self.h))

    def hops(self):
        return int(self.lib.fvh_hnsw_hops(self.h))

    def set_threads(self, t):
        self.lib.fvh_hnsw_set_threads(self.h, int(t))

    def set_device_traversal(self, on):
        """True (default): the layered walk runs on the GPU (one launch per batch); False: on the host
        with one scoring launch per hop (the north_star's split).  Results are identical."""
        self.lib.fvh_hnsw_set_device_traversal(self.h, int(bool(on)))

    def device_traversal(self):
        return bool(self.lib.fvh_hnsw_device_traversal(self.h))

    def set_device_insert(self, on, mode=0):
        """True (default): an insert's searches, links and prunes run on the GPU against the adjacency in HBM
        (fvdb_graph_insert_linked); False: the host algorithm with every distance batch scored on the GPU.
        mode 0 = choose, 1 = one insert at a time, 2 = speculate batches.  The graph is identical."""
        self.lib.fvh_hnsw_set_device_insert(self.h, int(bool(on)), int(mode))

    def device_insert(self):
        return bool(self.lib.fvh_hnsw_device_insert(self.h))

    def insert_stats(self):
        """Sums since construction: device-insert counters, inserts that took the host algorithm, and the
        host -> device bytes of graph structure moved (levels, patched rows, whole-graph installs)."""
        st = (C.c_uint32 * 10)()
        host, up = C.c_uint64(0), C.c_uint64(0)
        self.lib.fvh_hnsw_insert_stats(self.h, C.cast(st, C.c_void_p), C.byref(host), C.byref(up))
        names = ("n_done", "needs_host", "speculated_ok", "searched_in_commit", "commit_stops", "rounds", "expanded",
                 "rows_scored", "tie_restarts", "launches")
        out = {k: int(v) for k, v in zip(names, st)}
        out["host_path_inserts"] = host.value
        out["graph_upload_bytes"] = up.value
        return out

    def device_fallbacks(self):
        return int(self.lib.fvh_hnsw_device_fallbacks(self.h))

    def graph_kernel_times(self):
        """(summed ms, launches timed, rows scored, hops) of the device traversal kernel since the last call
        (timings need ctx profiling on; the counters are always kept)."""
        ms, n, rows, hops = C.c_float(0), C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        self.lib.fvh_hnsw_graph_kernel_times(self.h, C.byref(ms), C.byref(n), C.byref(rows), C.byref(hops))
        return float(ms.value), int(n.value), int(rows.value), int(hops.value)


    def tie_restarts(self):
        """(queries served by the device traversal, queries it searched again with the restated heaps) since creation."""
        q, a = C.c_uint64(0), C.c_uint64(0)
        self.lib.fvh_hnsw_tie_restarts(self.h, C.byref(q), C.byref(a))
        return int(q.value), int(a.value)


class HybridIndex(_Base):
    """src/hybrid/core.rs HybridIndex.  Timestamps / `now` are seconds supplied by the caller."""

    WEEK = 7 * 24 * 3600.0

    def __init__(self, ctx, recent_threshold=WEEK, migration_batch_size=100, auto_migrate=True,
                 min_ivf_training_size=10, max_connections=16, max_connections_layer_0=32, ef_construction=200,
                 hnsw_seed=0, n_clusters=3, n_probe=2, train_size=9, max_iterations=25, ivf_seed=0, ctx_hnsw=None):
        # defaults = HybridConfig::default (src/hybrid/core.rs:69-85)
        self.ctx, self.lib = ctx, load_host()
        self.ctx_hnsw = ctx_hnsw or ctx
        self.n_clusters, self.n_probe = n_clusters, n_probe
        self.config = dict(recent_threshold=recent_threshold, migration_batch_size=migration_batch_size,
                           auto_migrate=auto_migrate, min_ivf_training_size=min_ivf_training_size,
                           max_connections=max_connections, max_connections_layer_0=max_connections_layer_0,
                           ef_construction=ef_construction, hnsw_seed=hnsw_seed, n_clusters=n_clusters, n_probe=n_probe,
                           train_size=train_size, max_iterations=max_iterations, ivf_seed=ivf_seed)
        self.h = self.lib.fvh_hybrid_new(ctx.h, self.ctx_hnsw.h, recent_threshold, migration_batch_size,
                                         int(auto_migrate), min_ivf_training_size, max_connections,
                                         max_connections_layer_0, ef_construction, hnsw_seed, n_clusters, n_probe,
                                         train_size, max_iterations, ivf_seed)
        if not self.h:
            raise InvalidConfig("Invalid HybridConfig")

    def __del__(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.lib.fvh_hybrid_free(self.h)
            self.h = None

    def is_initialized(self):
        return bool(self.lib.fvh_hybrid_is_initialized(self.h))

    def is_ivf_trained(self):
        return bool(self.lib.fvh_hybrid_is_ivf_trained(self.h))

    def initialize(self, training_data):
        x = _f32(training_data)
        if x.ndim != 2:
            x = x.reshape(len(training_data), -1)
        self._check(self.lib.fvh_hybrid_initialize(self.h, _ptr(x, f32p), x.shape[0], x.shape[1]))

    def set_ivf_centroids(self, centroids):
        c = _rows(centroids)
        self._check(self.lib.fvh_hybrid_set_ivf_centroids(self.h, _ptr(c, f32p), c.shape[1]))

    def insert_with_timestamp(self, id, vector, timestamp, now, level=-1):
        v = _f32(vector).reshape(-1)
        self._check(self.lib.fvh_hybrid_insert(self.h, int(id), _ptr(v, f32p), v.size, float(timestamp), float(now), level))

    def insert(self, id, vector, now=0.0, level=-1):
        self.insert_with_timestamp(id, vector, now, now, level)

    def set_sequential_graph(self, on):
        """How bulk_insert builds the recent part's graph: True (default) = the reference's sequential inserts
        (device-resident), False = HNSWIndex.bulk_build (exact nearest-M per layer: an extension, another graph)."""
        self.lib.fvh_hybrid_set_sequential_graph(self.h, int(bool(on)))

    def sequential_graph(self):
        return bool(self.lib.fvh_hybrid_sequential_graph(self.h))

    def set_blocking_writers(self, on):
        """Mutations while batches begun with search_dev_begin are uncollected: False (default) = FvdbError(INVALID) at once,
        True = wait until they are collected (the reference's write guard; for searches and writes on different threads)."""
        self.lib.fvh_hybrid_set_blocking_writers(self.h, int(bool(on)))

    def recent_build_seconds(self):
        return float(self.lib.fvh_hybrid_recent_build_seconds(self.h))

    def bulk_insert(self, ids, vectors, timestamps, now):
        v = _rows(vectors)
        ids = np.ascontiguousarray(ids, np.uint64)
        ts = np.ascontiguousarray(timestamps, np.float64)
        self._check(self.lib.fvh_hybrid_bulk_insert(self.h, _ptr(ids, u64p), _ptr(v, f32p), v.shape[0], v.shape[1],
                                                    _ptr(ts, f64p), float(now)))

    def bulk_insert_sharded(self, ids, vectors, timestamps, now, rank, world):
        """Multi-GPU placement: HNSW replicated, IVF lists owned by `rank` only.  Returns owner[nlist]."""
        v = _rows(vectors)
        ids = np.ascontiguousarray(ids, np.uint64)
        ts = np.ascontiguousarray(timestamps, np.float64)
        owner = np.zeros(self.n_clusters, np.uint32)
        self._check(self.lib.fvh_hybrid_bulk_insert_sharded(self.h, _ptr(ids, u64p), _ptr(v, f32p), v.shape[0],
                                                            v.shape[1], _ptr(ts, f64p), float(now), rank, world,
                                                            _ptr(owner, u32p)))
        return owner

    def search(self, queries, k, now=0.0, hnsw_ef=50, ivf_n_probe=10, search_recent=True, search_historical=True,
               recent_k=0, historical_k=0):
        # HybridSearchConfig::default (src/hybrid/core.rs:184-197)
        return self._search(self.lib.fvh_hybrid_search, queries, k, hnsw_ef, ivf_n_probe, int(search_recent),
                            int(search_historical), recent_k, historical_k, float(now))

    search_with_config = search

    _FILTER_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_void_p)

    def search_with_filter(self, queries, k, matches=None, now=0.0):
        """HybridIndex::search_with_filter (src/hybrid/core.rs:513-549) in the host mirror: 3 k candidates with the
        default search config, the first k whose id `matches(id)` accepts (None = no filter = plain search).  The
        oversampling, the walk over the candidates and the truncation run below Python; `matches` stands for the
        application's metadata_map lookup + MetadataFilter::matches."""
        q = _rows(queries)
        B = q.shape[0]
        ids = np.empty((B, max(k, 1)), np.uint64)
        ds = np.empty((B, max(k, 1)), np.float32)
        cnt = np.zeros(B, np.uint32)
        failure = []

        def trampoline(rid, _user):
            try:
                return 1 if matches(int(rid)) else 0
            except Exception as e:  # noqa: BLE001 — must not unwind through the C frames
                failure.append(e)
                return 0

        cb = self._FILTER_FN(trampoline) if matches is not None else C.cast(None, self._FILTER_FN)
        self._check(self.lib.fvh_hybrid_search_with_filter(self.h, _ptr(q, f32p), B, q.shape[1], k, C.cast(cb, C.c_void_p),
                                                           None, float(now), _ptr(ids, u64p), _ptr(ds, f32p), _ptr(cnt, u32p)))
        if failure:
            raise failure[0]
        return SearchResults(ids, ds, cnt)

    def search_dev(self, q_dev, B, k, now=0.0, hnsw_ef=50, ivf_n_probe=10, search_recent=True, search_historical=True,
                   recent_k=0, historical_k=0, dim=None):
        """Same search with the B x d query batch already resident in HBM (device pointer)."""
        ids = np.empty((B, max(k, 1)), np.uint64)
        ds = np.empty((B, max(k, 1)), np.float32)
        cnt = np.zeros(B, np.uint32)
        self._check(self.lib.fvh_hybrid_search_dev(self.h, q_dev, B, dim, k, hnsw_ef, ivf_n_probe, int(search_recent),
                                                   int(search_historical), recent_k, historical_k, float(now),
                                                   _ptr(ids, u64p), _ptr(ds, f32p), _ptr(cnt, u32p)))
        return SearchResults(ids, ds, cnt)

    SLOTS = 16

    def search_dev_begin(self, slot, q_dev, B, k, now=0.0, hnsw_ef=50, ivf_n_probe=10, search_recent=True,
                         search_historical=True, recent_k=0, historical_k=0, dim=None):
        """Enqueue a batch search in `slot` (0..SLOTS-1) and return at once; several slots may be in flight together
        (the graph walk of one batch uses one wavefront per SIMD, so a second batch's walk runs beside it).  The
        query buffer must stay valid until search_dev_end(slot).  Inserts, deletes and migrations are refused (status
        INVALID) while any slot is in flight, and so is a begin whose `now` makes an auto-migration due."""
        self._check(self.lib.fvh_hybrid_search_dev_begin(self.h, slot, q_dev, B, dim, k, hnsw_ef, ivf_n_probe,
                                                         int(search_recent), int(search_historical), recent_k,
                                                         historical_k, float(now)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[slot] = (B, k)

    def search_dev_end(self, slot):
        """Wait for the batch of `slot` and return its SearchResults."""
        B, k = self._inflight.pop(slot)
        ids = np.empty((B, max(k, 1)), np.uint64)
        ds = np.empty((B, max(k, 1)), np.float32)
        cnt = np.zeros(B, np.uint32)
        self._check(self.lib.fvh_hybrid_search_dev_end(self.h, slot, _ptr(ids, u64p), _ptr(ds, f32p), _ptr(cnt, u32p)))
        return SearchResults(ids, ds, cnt)

    # ---- multi-GPU (sharded.py is the user-facing wrapper) ----
    def attach_comm(self, comm_handle):
        self._check(self.lib.fvh_hybrid_attach_comm(self.h, comm_handle))

    def sharded_rows(self, B, mode):
        return int(self.lib.fvh_hybrid_sharded_rows(self.h, B, mode))

    def search_sharded_begin(self, slot, q_dev, B, k, mode, hnsw_ef=50, ivf_n_probe=10, search_recent=True,
                             search_historical=True, recent_k=0, historical_k=0, dim=None, now=0.0):
        self._check(self.lib.fvh_hybrid_search_sharded_begin(self.h, slot, q_dev, B, dim, k, hnsw_ef, ivf_n_probe,
                                                             int(search_recent), int(search_historical), recent_k,
                                                             historical_k, mode, float(now)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[slot] = (self.sharded_rows(B, mode), k)

    def search_sharded_end(self, slot):
        R, k = self._inflight.pop(slot)
        ids = np.empty((max(R, 1), max(k, 1)), np.uint64)
        ds = np.empty((max(R, 1), max(k, 1)), np.float32)
        cnt = np.zeros(max(R, 1), np.uint32)
        self._check(self.lib.fvh_hybrid_search_sharded_end(self.h, slot, _ptr(ids, u64p), _ptr(ds, f32p), _ptr(cnt, u32p)))
        return SearchResults(ids[:R], ds[:R], cnt[:R])

    def delete(self, id, now=0.0):
        self._check(self.lib.fvh_hybrid_delete(self.h, int(id), float(now)))

    def vacuum(self):
        """Physically drop soft-deleted vectors from both indexes (src/hybrid/core.rs:989-1012)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.fvh_hybrid_vacuum(self.h, C.byref(a), C.byref(b)))
        return {"hnsw_removed": a.value, "ivf_removed": b.value, "total_removed": a.value + b.value}

    def from_parts(self, ids, timestamps, recent_count, historical_count, ivf_trained):
        """HybridIndex::from_parts (src/hybrid/core.rs:857-877): adopt hnsw() / ivf() as restored by the caller."""
        ids = np.ascontiguousarray(ids, np.uint64)
        ts = np.ascontiguousarray(timestamps, np.float64)
        self._check(self.lib.fvh_hybrid_from_parts(self.h, _ptr(ids, u64p), _ptr(ts, f64p), ids.size, int(recent_count),
                                                   int(historical_count), int(bool(ivf_trained))))

    def export_timestamps(self):
        n = int(self.lib.fvh_hybrid_timestamp_count(self.h))
        ids, ts = np.empty(n, np.uint64), np.empty(n, np.float64)
        self.lib.fvh_hybrid_export_timestamps(self.h, _ptr(ids, u64p), _ptr(ts, f64p))
        return ids, ts

    def migrate_with_threshold(self, threshold, now):
        return int(self.lib.fvh_hybrid_migrate(self.h, float(threshold), float(now)))

    def recent_count(self):
        return int(self.lib.fvh_hybrid_recent_count(self.h))

    def historical_count(self):
        return int(self.lib.fvh_hybrid_historical_count(self.h))

    def hnsw(self):
        return HNSWIndex(self.ctx_hnsw, _handle=self.lib.fvh_hybrid_hnsw(self.h))

    def ivf_device_stage_times(self):
        return self.ivf().stage_times()

    def ivf_device_last_stats(self):
        return self.ivf().last_stats()

    def ivf(self):
        return IVFIndex(self.ctx, n_clusters=self.n_clusters, n_probe=self.n_probe, _handle=self.lib.fvh_hybrid_ivf(self.h))
