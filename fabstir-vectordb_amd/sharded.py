"""Multi-GPU execution of the hybrid search: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

Sharding:
  * IVF lists are owned by ranks (whole lists, largest-first greedy balance); centroids and the query
    batch are replicated; every rank scans the probed lists it owns for ALL queries and emits a B x k
    partial result with the selection keys (distance bits << 32 | global scan position).
  * ONE collective per batch: all-gather of the per-rank (keys, ids) = B*k*16 bytes per rank
    (B=1024, k=10: 160 KB) — latency-bound, so a single fused all-gather, not one per tensor.
  * keys are unique across ranks => the G-way merge by key is exactly the single-GPU result.
  * The HNSW graph is replicated; its queries are split over the ranks (contiguous slices) and the
    slices' results ride in the same all-gather.
The reference has no distributed execution at all; this module is new work on top of the same
HybridIndex surface.
"""
import numpy as np

NO_ID = np.uint64(0xFFFFFFFFFFFFFFFF)
INF_BITS = np.uint32(0x7F800000)


def plan_list_shards(list_sizes, world):
    """owner[list]: largest list first to the least loaded rank (ties -> lower rank/list id).
    Must match fvdbh::plan_list_owners (host/hybrid_index.cpp)."""
    sizes = np.asarray(list_sizes, np.uint64)
    order = np.argsort(-sizes.astype(np.int64), kind="stable")
    load = np.zeros(world, np.uint64)
    owner = np.zeros(sizes.size, np.uint32)
    for L in order:
        r = int(np.argmin(load))  # first minimum = lowest rank
        owner[L] = r
        load[r] += sizes[L]
    return owner


def query_slices(B, world):
    """Contiguous, equal-length (padded) query slices for the replicated HNSW part."""
    per = -(-B // world)
    return [(min(r * per, B), min((r + 1) * per, B)) for r in range(world)], per


def hybrid_merge(h_ids, h_ds, h_cnt, i_ids, i_ds, i_cnt, k):
    """HybridIndex::search_with_config's merge (src/hybrid/core.rs:482-483): HNSW results then IVF
    results, stable sort by distance, truncate(k).  Vectorised over the batch."""
    B = h_ids.shape[0]
    kh, ki = h_ids.shape[1], i_ids.shape[1]
    ids = np.concatenate([h_ids, i_ids], axis=1)
    ds = np.concatenate([h_ds, i_ds], axis=1).astype(np.float32).copy()
    col = np.arange(kh + ki)[None, :]
    valid = np.concatenate([np.arange(kh)[None, :] < h_cnt[:, None], np.arange(ki)[None, :] < i_cnt[:, None]], axis=1)
    ds[~valid] = np.inf
    # stable ascending by distance; invalid (inf) entries sink, but real +inf distances must stay
    # ahead of padding: sort on (invalid, distance) lexicographically
    order = np.lexsort((col.repeat(B, 0), ds, ~valid), axis=1)[:, :k]
    out_ids = np.take_along_axis(ids, order, axis=1)
    out_ds = np.take_along_axis(ds, order, axis=1)
    cnt = np.minimum(h_cnt.astype(np.int64) + i_cnt.astype(np.int64), k).astype(np.uint32)
    pad = np.arange(out_ids.shape[1])[None, :] >= cnt[:, None]
    out_ids[pad] = NO_ID
    out_ds[pad] = np.inf
    return out_ids, out_ds, cnt


def pack_partials(keys, ids, h_ids, h_ds, h_cnt, per, k):
    """One int64 buffer per rank for the single all-gather: [IVF keys | IVF ids | HNSW slice]."""
    B = keys.shape[0]
    hs = np.full((per, 2 * k + 1), -1, np.int64)  # per query: k ids, k distance bits, count
    n = h_ids.shape[0]
    hs[:n, :k] = h_ids.view(np.int64)
    hs[:n, k:2 * k] = h_ds.view(np.uint32).astype(np.int64)
    hs[:n, 2 * k] = h_cnt
    return np.concatenate([keys.view(np.int64).reshape(-1), ids.view(np.int64).reshape(-1), hs.reshape(-1)]), B


def unpack_partials(buf_all, world, B, per, k):
    n_ivf = B * k
    keys = np.empty((world, B, k), np.uint64)
    ids = np.empty((world, B, k), np.uint64)
    h_ids = np.full((world * per, k), NO_ID, np.uint64)
    h_ds = np.full((world * per, k), np.inf, np.float32)
    h_cnt = np.zeros(world * per, np.uint32)
    stride = 2 * n_ivf + per * (2 * k + 1)
    for r in range(world):
        b = buf_all[r * stride:(r + 1) * stride]
        keys[r] = b[:n_ivf].view(np.uint64).reshape(B, k)
        ids[r] = b[n_ivf:2 * n_ivf].view(np.uint64).reshape(B, k)
        hs = b[2 * n_ivf:].reshape(per, 2 * k + 1)
        h_ids[r * per:(r + 1) * per] = hs[:, :k].view(np.uint64)
        h_ds[r * per:(r + 1) * per] = hs[:, k:2 * k].astype(np.uint32).view(np.float32)
        h_cnt[r * per:(r + 1) * per] = np.maximum(hs[:, 2 * k], 0).astype(np.uint32)
    return keys, ids, h_ids[:B], h_ds[:B], h_cnt[:B]


class _Res:
    def __init__(self, ids, distances, counts):
        self.ids, self.distances, self.counts = ids, distances, counts


class ShardedHybrid:
    """HybridIndex across `world` ranks (see module docstring).  Bench/scale surface: bulk placement
    and batched search; per-search auto-migration is not run in this mode (nothing ages during a
    bench; the single-GPU HybridIndex keeps the reference's behaviour)."""

    def __init__(self, fv, hyb, rank, world, dist, torch):
        self.fv, self.hyb, self.rank, self.world, self.dist, self.torch = fv, hyb, rank, world, dist, torch
        self.owner = None
        self._bufs = {}

    def bulk_insert(self, ids, x, ts, now):
        self.owner = self.hyb.bulk_insert_sharded(ids, x, ts, now, self.rank, self.world)
        self.d = x.shape[1]

    def _tensor(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self.torch.empty(shape, dtype=dtype, device="cuda")
            self._bufs[name] = t
        return t

    def search_dev(self, q_dev, q_host, B, k, now, ef, nprobe):
        torch, dist, fv = self.torch, self.dist, self.fv
        import ctypes as C
        ivf, hnsw, ctx = self.hyb.ivf(), self.hyb.hnsw(), self.hyb.ctx
        keys = self._tensor("keys", (B, k), torch.int64)
        ids = self._tensor("ids", (B, k), torch.int64)
        ds = self._tensor("ds", (B, k), torch.float32)
        cnt = self._tensor("cnt", (B,), torch.int32)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        # 1. IVF partial for ALL queries over the lists this rank owns (async on the engine's stream)
        ctx.check(ctx.lib.fvdb_ivf_search_dev(ivf._dev(), q_dev, B, k, nprobe, p(ids), p(ds), p(cnt), p(keys)))
        # 2. HNSW for this rank's slice of the queries (host walk, hops scored on the GPU)
        slices, per = query_slices(B, self.world)
        lo, hi = slices[self.rank]
        if hi > lo:
            q_slice = C.c_void_p(q_dev.value + lo * self.d * 4)
            h = hnsw.search_dev(q_slice, hi - lo, self.d, k, ef)
            h_ids, h_ds, h_cnt = h.ids, h.distances, h.counts
        else:
            h_ids, h_ds, h_cnt = (np.empty((0, k), np.uint64), np.empty((0, k), np.float32), np.empty(0, np.uint32))
        ctx.synchronize()
        # 3. one all-gather carrying IVF (keys, ids) and the HNSW slice
        hs = np.full((per, 2 * k + 1), -1, np.int64)
        n = h_ids.shape[0]
        hs[:n, :k] = h_ids.view(np.int64)
        hs[:n, k:2 * k] = h_ds.view(np.uint32).astype(np.int64)
        hs[:n, 2 * k] = h_cnt
        mine = self._tensor("mine", (2 * B * k + per * (2 * k + 1),), torch.int64)
        mine[:B * k] = keys.reshape(-1)
        mine[B * k:2 * B * k] = ids.reshape(-1)
        mine[2 * B * k:] = torch.from_numpy(hs.reshape(-1)).cuda(non_blocking=False)
        allb = self._tensor("all", (self.world * mine.numel(),), torch.int64)
        dist.all_gather_into_tensor(allb, mine)
        # 4. G-way merge by key on the GPU (fvdb_merge_keys_dev), HNSW slices back on the host
        stride = mine.numel()
        gk = self._tensor("gk", (self.world, B, k), torch.int64)
        gi = self._tensor("gi", (self.world, B, k), torch.int64)
        allv = allb.view(self.world, stride)
        gk.copy_(allv[:, :B * k].reshape(self.world, B, k))
        gi.copy_(allv[:, B * k:2 * B * k].reshape(self.world, B, k))
        torch.cuda.synchronize()
        oi = self._tensor("oi", (B, k), torch.int64)
        od = self._tensor("od", (B, k), torch.float32)
        oc = self._tensor("oc", (B,), torch.int32)
        fv.engine.merge_keys_dev(ctx, p(gk), p(gi), self.world, B, k, p(oi), p(od), p(oc))
        ctx.synchronize()
        i_ids = oi.cpu().numpy().view(np.uint64)
        i_ds = od.cpu().numpy()
        i_cnt = oc.cpu().numpy().view(np.uint32)
        hall = allv[:, 2 * B * k:].reshape(self.world * per, 2 * k + 1).cpu().numpy()
        g_ids = hall[:B, :k].copy().view(np.uint64)
        g_ds = hall[:B, k:2 * k].astype(np.uint32).view(np.float32)
        g_cnt = np.maximum(hall[:B, 2 * k], 0).astype(np.uint32)
        # 5. the reference's hybrid merge
        return _Res(*hybrid_merge(g_ids, g_ds, g_cnt, i_ids, i_ds, i_cnt, k))
