"""Thin Python view of the C ABI (include/fvdb.h): handles, numpy marshalling, errors.

Error classes follow the reference's enums (IVFError / HNSWError / HybridError,
src/ivf/core.rs:14-39, src/hnsw/core.rs, src/hybrid/core.rs).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import f32p, u32p, u64p


class FvdbError(Exception):
    status = None


class NotTrained(FvdbError):
    pass


class DuplicateVector(FvdbError):
    pass


class DimensionMismatch(FvdbError):
    pass


class InsufficientTrainingData(FvdbError):
    pass


class InconsistentDimensions(FvdbError):
    pass


class InvalidConfig(FvdbError):
    pass


class VectorNotFound(FvdbError):
    pass


class NotInitialized(FvdbError):
    pass


class NonFiniteInput(FvdbError):
    pass


class HipError(FvdbError):
    pass


class OutOfMemory(FvdbError):
    pass


class Unsupported(FvdbError):
    pass


STATUS_TO_EXC = {1: NotTrained, 2: DuplicateVector, 3: DimensionMismatch, 4: InsufficientTrainingData,
                 5: InconsistentDimensions, 6: InvalidConfig, 7: VectorNotFound, 8: NotInitialized,
                 9: NonFiniteInput, 10: HipError, 11: OutOfMemory, 12: Unsupported}


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Context:
    """One GPU + one HIP stream (fvdb_ctx)."""

    def __init__(self, device=0):
        self.lib = _capi.load()
        h = C.c_void_p()
        rc = self.lib.fvdb_ctx_create(device, C.byref(h))
        if rc:
            raise STATUS_TO_EXC.get(rc, FvdbError)(f"fvdb_ctx_create(device={device}) failed with status {rc}")
        self.h = h
        self.device = device

    def check(self, rc):
        if rc:
            msg = self.lib.fvdb_last_error(self.h)
            exc = STATUS_TO_EXC.get(rc, FvdbError)(msg.decode() if msg else f"status {rc}")
            exc.status = rc
            raise exc

    def close(self):
        if getattr(self, "h", None):
            self.lib.fvdb_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        """device, compute units, and the hardware-queue count batches in flight can use (fvdb_ctx_info): hw_queues_source
        1 = set by the host, 2 = set by the library, 3 = could not be applied (HIP was initialised first)."""
        v = (C.c_int * 4)()
        self.check(self.lib.fvdb_ctx_info(self.h, C.cast(v, C.c_void_p)))
        return dict(device=v[0], compute_units=v[1], hw_queues=v[2], hw_queues_source=v[3])

    def device_synchronize(self):
        """Wait for every stream of this context's device (the engine keeps one stream per batch in flight)."""
        self.check(self.lib.fvdb_device_synchronize(self.h))

    def synchronize(self):
        self.check(self.lib.fvdb_ctx_synchronize(self.h))

    def set_profiling(self, on):
        self.check(self.lib.fvdb_ctx_set_profiling(self.h, int(on)))

    # --- raw device buffers (bench keeps its inputs resident in HBM) ---
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.fvdb_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, p):
        self.check(self.lib.fvdb_dev_free(self.h, p))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        self.check(self.lib.fvdb_dev_upload(self.h, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self.check(self.lib.fvdb_dev_download(self.h, out.ctypes.data_as(C.c_void_p), p, out.nbytes))
        return out

    def timer_start(self):
        self.check(self.lib.fvdb_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float(0)
        self.check(self.lib.fvdb_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value


def _pairwise(ctx, fn, q, x):
    q, x = _f32(q), _f32(x)
    if q.ndim == 1:
        q = q.reshape(1, -1)
    if x.ndim == 1:
        x = x.reshape(1, -1)
    if q.shape[1] != x.shape[1]:
        raise DimensionMismatch(f"Dimension mismatch: {q.shape[1]} != {x.shape[1]}")
    out = np.empty((q.shape[0], x.shape[0]), np.float32)
    ctx.check(fn(ctx.h, _ptr(q, f32p), q.shape[0], _ptr(x, f32p), x.shape[0], q.shape[1], _ptr(out, f32p)))
    return out


def dot_products(ctx, q, x):
    """dot_product_scalar for every (query, row) pair (src/core/vector_ops.rs:35-37)."""
    return _pairwise(ctx, ctx.lib.fvdb_dot_products, q, x)


def batch_cosine_similarity(ctx, q, x):
    """batch_cosine_similarity / cosine_similarity_scalar (src/core/vector_ops.rs:8-10,39-49)."""
    return _pairwise(ctx, ctx.lib.fvdb_cosine_similarities, q, x)


def _score_rows(scores):
    s = _f32(scores)
    return s.reshape(1, -1) if s.ndim == 1 else s


def _top_k(ctx, fn, scores, k):
    s = _score_rows(scores)
    B, n = s.shape
    out = np.full((B, max(k, 1)), 2**64 - 1, np.uint64)
    cnt = np.zeros(B, np.uint32)
    ctx.check(fn(ctx.h, _ptr(s, f32p), B, n, k, _ptr(out, u64p), _ptr(cnt, u32p)))
    rows = [out[b, : cnt[b]].astype(np.int64).tolist() for b in range(B)]
    return rows[0] if np.ndim(scores) == 1 else rows


def top_k_indices(ctx, scores, k):
    """top_k_indices (src/core/vector_ops.rs:12-22): indices of the k largest scores, ties in index order.  `scores`
    is one row or a batch of rows (B x n); a batch returns one list per row."""
    return _top_k(ctx, ctx.lib.fvdb_top_k_indices, scores, k)


def top_k_indices_heap(ctx, scores, k):
    """top_k_indices_heap (src/core/vector_ops.rs:180-201), the size-k BinaryHeap variant with its exact tie behaviour."""
    return _top_k(ctx, ctx.lib.fvdb_top_k_indices_heap, scores, k)


def _pairs(ctx, fn, ids, vals, k):
    ids = np.ascontiguousarray(ids, np.uint64)
    v = _f32(vals)
    single = v.ndim == 1
    if single:
        ids, v = ids.reshape(1, -1), v.reshape(1, -1)
    B, n = v.shape
    oi = np.full((B, max(k, 1)), 2**64 - 1, np.uint64)
    ov = np.zeros((B, max(k, 1)), np.float32)
    cnt = np.zeros(B, np.uint32)
    ctx.check(fn(ctx.h, _ptr(ids, u64p), _ptr(v, f32p), B, n, k, _ptr(oi, u64p), _ptr(ov, f32p), _ptr(cnt, u32p)))
    rows = [[(int(oi[b, i]), float(ov[b, i])) for i in range(cnt[b])] for b in range(B)]
    return rows[0] if single else rows


def streaming_top_k(ctx, ids, scores, k):
    """StreamingTopK (src/core/vector_ops.rs:204-263): add every (id, score) in order, then get_results() ->
    [(id, score)] by descending score."""
    return _pairs(ctx, ctx.lib.fvdb_streaming_top_k, ids, scores, k)


def merge_search_results(ctx, result_sets, k):
    """merge_search_results (src/core/vector_ops.rs:24-32): concatenate the result sets [(id, distance), ...], keep each
    id's smallest distance (SearchResult::deduplicate, src/core/types.rs:206-223), ascending, first k."""
    ids = np.array([r[0] for rs in result_sets for r in rs], np.uint64)
    ds = np.array([r[1] for rs in result_sets for r in rs], np.float32)
    if ids.size == 0:
        return []
    return _pairs(ctx, ctx.lib.fvdb_merge_search_results, ids, ds, k)


class DeviceIVF:
    """IVF-flat index resident in HBM (fvdb_ivf): centroids + paged inverted lists."""

    def __init__(self, ctx, d, nlist, dtype="f32"):
        self.ctx, self.lib = ctx, ctx.lib
        self.d, self.nlist, self.dtype = int(d), int(nlist), dtype
        h = C.c_void_p()
        ctx.check(self.lib.fvdb_ivf_create_ex(ctx.h, self.d, self.nlist, {"f32": 0, "f16": 1}[dtype], C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.lib.fvdb_ivf_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rows(self, x):
        x = _f32(x)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        if x.shape[1] != self.d:
            raise DimensionMismatch(f"Dimension mismatch: expected {self.d}, got {x.shape[1]}")
        return x

    def set_centroids(self, centroids):
        c = self._rows(centroids)
        if c.shape[0] != self.nlist:
            raise InvalidConfig(f"expected {self.nlist} centroids, got {c.shape[0]}")
        self.ctx.check(self.lib.fvdb_ivf_set_centroids(self.h, _ptr(c, f32p)))

    def get_centroids(self):
        out = np.empty((self.nlist, self.d), np.float32)
        self.ctx.check(self.lib.fvdb_ivf_get_centroids(self.h, _ptr(out, f32p)))
        return out

    def train(self, x, max_iterations=25, seed=0):
        x = self._rows(x)
        res = _capi.TrainResult()
        self.ctx.check(self.lib.fvdb_ivf_train(self.h, _ptr(x, f32p), x.shape[0], max_iterations, seed, C.byref(res)))
        return dict(iterations=res.iterations, converged=bool(res.converged), initial_error=res.initial_error,
                    final_error=res.final_error)

    def assign(self, x):
        x = self._rows(x)
        out = np.empty(x.shape[0], np.uint32)
        self.ctx.check(self.lib.fvdb_ivf_assign(self.h, _ptr(x, f32p), x.shape[0], _ptr(out, u32p)))
        return out

    def add(self, x, ids):
        x = self._rows(x)
        ids = np.ascontiguousarray(ids, np.uint64)
        cl = np.empty(x.shape[0], np.uint32)
        pos = np.empty(x.shape[0], np.uint32)
        self.ctx.check(self.lib.fvdb_ivf_add(self.h, _ptr(x, f32p), _ptr(ids, u64p), x.shape[0], _ptr(cl, u32p),
                                             _ptr(pos, u32p)))
        return cl, pos

    def add_assigned(self, x, ids, clusters):
        x = self._rows(x)
        ids = np.ascontiguousarray(ids, np.uint64)
        cl = np.ascontiguousarray(clusters, np.uint32)
        pos = np.empty(x.shape[0], np.uint32)
        self.ctx.check(self.lib.fvdb_ivf_add_assigned(self.h, _ptr(x, f32p), _ptr(ids, u64p), x.shape[0],
                                                      _ptr(cl, u32p), _ptr(pos, u32p)))
        return pos

    def set_deleted(self, clusters, pos, deleted=True):
        cl = np.ascontiguousarray(clusters, np.uint32)
        ps = np.ascontiguousarray(pos, np.uint32)
        self.ctx.check(self.lib.fvdb_ivf_set_deleted(self.h, _ptr(cl, u32p), _ptr(ps, u32p), cl.size, int(deleted)))

    def list_sizes(self):
        out = np.empty(self.nlist, np.uint64)
        self.ctx.check(self.lib.fvdb_ivf_list_sizes(self.h, _ptr(out, u64p)))
        return out

    def list_export(self, c):
        """Rows (f32), ids and live flags of list `c` in list-position order, copied back from HBM."""
        import ctypes as C
        n = int(self.list_sizes()[c])
        rows, ids, live = np.empty((n, self.d), np.float32), np.empty(n, np.uint64), np.empty(n, np.uint8)
        self.ctx.check(self.lib.fvdb_ivf_list_export(self.h, int(c), _ptr(rows, f32p), _ptr(ids, u64p),
                                                     live.ctypes.data_as(C.POINTER(C.c_uint8))))
        return rows, ids, live.astype(bool)

    def total_rows(self):
        return int(self.lib.fvdb_ivf_total_rows(self.h))

    def reserve(self, n_rows):
        self.ctx.check(self.lib.fvdb_ivf_reserve(self.h, n_rows))

    def clear(self):
        self.ctx.check(self.lib.fvdb_ivf_clear(self.h))

    def set_global_list_sizes(self, sizes):
        s = np.ascontiguousarray(sizes, np.uint64)
        self.ctx.check(self.lib.fvdb_ivf_set_global_list_sizes(self.h, _ptr(s, u64p)))

    def _out(self, B, k):
        return (np.empty((B, k), np.uint64), np.empty((B, k), np.float32), np.empty(B, np.uint32))

    def search(self, q, k, nprobe):
        q = self._rows(q)
        ids, ds, cnt = self._out(q.shape[0], k)
        self.ctx.check(self.lib.fvdb_ivf_search(self.h, _ptr(q, f32p), q.shape[0], k, nprobe, _ptr(ids, u64p),
                                                _ptr(ds, f32p), _ptr(cnt, u32p)))
        return ids, ds, cnt

    def search_all(self, q, k):
        q = self._rows(q)
        ids, ds, cnt = self._out(q.shape[0], k)
        self.ctx.check(self.lib.fvdb_ivf_search_all(self.h, _ptr(q, f32p), q.shape[0], k, _ptr(ids, u64p),
                                                    _ptr(ds, f32p), _ptr(cnt, u32p)))
        return ids, ds, cnt

    def search_dev(self, q_dev, B, k, nprobe, ids_dev, dist_dev, cnt_dev, keys_dev=None):
        self.ctx.check(self.lib.fvdb_ivf_search_dev(self.h, q_dev, B, k, nprobe, ids_dev, dist_dev, cnt_dev, keys_dev))

    def search_all_dev(self, q_dev, B, k, ids_dev, dist_dev, cnt_dev):
        self.ctx.check(self.lib.fvdb_ivf_search_all_dev(self.h, q_dev, B, k, ids_dev, dist_dev, cnt_dev))

    def coarse(self, q, nprobe):
        q = self._rows(q)
        npb = min(nprobe, self.nlist)
        cl = np.empty((q.shape[0], npb), np.uint32)
        ds = np.empty((q.shape[0], npb), np.float32)
        self.ctx.check(self.lib.fvdb_ivf_coarse(self.h, _ptr(q, f32p), q.shape[0], nprobe, _ptr(cl, u32p), _ptr(ds, f32p)))
        return cl, ds

    def set_coarse_mode(self, mode):
        """0 = matrix cores propose + exact verification (default), 1 = exact scan of every centroid."""
        self.ctx.check(self.lib.fvdb_ivf_set_coarse_mode(self.h, int(mode)))

    def coarse_fallbacks(self):
        v = C.c_uint64(0)
        self.ctx.check(self.lib.fvdb_ivf_coarse_fallbacks(self.h, C.byref(v)))
        return int(v.value)

    def set_scan_mode(self, mode):
        """0 = fp16 MFMA filter + exact verification (default), 1 = exact scan of every probed row."""
        self.ctx.check(self.lib.fvdb_ivf_set_scan_mode(self.h, int(mode)))

    def scan_fallbacks(self):
        v = C.c_uint64(0)
        self.ctx.check(self.lib.fvdb_ivf_scan_fallbacks(self.h, C.byref(v)))
        return int(v.value)

    def scan_fallback_reasons(self):
        out = np.zeros(5, np.uint64)
        self.ctx.check(self.lib.fvdb_ivf_scan_fallback_reasons(self.h, _ptr(out, u64p)))
        return dict(zip(("survivor_overflow", "too_many_candidates", "bound_not_strict", "no_threshold", "refined"), out.tolist()))

    def scan_survivors(self, B):
        out = np.empty(B, np.uint32)
        self.ctx.check(self.lib.fvdb_ivf_scan_survivors(self.h, _ptr(out, u32p), B))
        return out

    def scan_survivor_dump(self, query, max_n=4096):
        """(probe rank, position in list, matrix-core value v) of one query's survivors in the last MFMA batch."""
        rank = np.empty(max_n, np.uint32)
        pos = np.empty(max_n, np.uint32)
        v = np.empty(max_n, np.float32)
        n = C.c_uint32(0)
        self.ctx.check(self.lib.fvdb_ivf_scan_survivor_dump(self.h, query, max_n, _ptr(rank, u32p), _ptr(pos, u32p),
                                                            _ptr(v, f32p), C.byref(n)))
        return rank[:n.value], pos[:n.value], v[:n.value]

    def last_stats(self):
        st = _capi.SearchStats()
        self.ctx.check(self.lib.fvdb_ivf_last_stats(self.h, C.byref(st)))
        return dict(rows_scanned=st.rows_scanned, work_items=st.work_items, list_rows_touched=st.list_rows_touched)

    def stage_times(self):
        ms = np.zeros(8, np.float32)
        n = self.lib.fvdb_ivf_stage_times(self.h, _ptr(ms, f32p))
        return int(n), dict(zip(("coarse_scan", "coarse_merge", "plan", "fine_scan", "fine_merge", "mfma_filter_kernel"), ms.tolist()))


class RowStore:
    """Row-major vector store for gathered candidate scoring (fvdb_store)."""

    def __init__(self, ctx, d, capacity_rows=1024):
        self.ctx, self.lib, self.d = ctx, ctx.lib, int(d)
        h = C.c_void_p()
        ctx.check(self.lib.fvdb_store_create(ctx.h, self.d, capacity_rows, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.lib.fvdb_store_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def append(self, rows):
        r = _f32(rows)
        if r.ndim == 1:
            r = r.reshape(1, -1)
        if r.shape[1] != self.d:
            raise DimensionMismatch(f"Dimension mismatch: expected {self.d}, got {r.shape[1]}")
        first = C.c_uint64(0)
        self.ctx.check(self.lib.fvdb_store_append(self.h, _ptr(r, f32p), r.shape[0], C.byref(first)))
        return first.value

    def rows(self):
        return int(self.lib.fvdb_store_rows(self.h))

    def get(self, row):
        out = np.empty(self.d, np.float32)
        self.ctx.check(self.lib.fvdb_store_get(self.h, row, _ptr(out, f32p)))
        return out

    def score_candidates(self, q, cand):
        q = _f32(q)
        cand = np.ascontiguousarray(cand, np.uint32)
        B, Cn = cand.shape
        out = np.empty((B, Cn), np.float32)
        self.ctx.check(self.lib.fvdb_score_candidates(self.h, _ptr(q, f32p), B, _ptr(cand, u32p), Cn, _ptr(out, f32p)))
        return out


def merge_keys_dev(ctx, keys_dev, ids_dev, G, B, k, out_ids_dev, out_dist_dev, out_cnt_dev):
    ctx.check(ctx.lib.fvdb_merge_keys_dev(ctx.h, keys_dev, ids_dev, G, B, k, out_ids_dev, out_dist_dev, out_cnt_dev))
