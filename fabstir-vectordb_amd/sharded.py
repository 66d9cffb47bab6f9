"""Multi-GPU execution of the hybrid search: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests and the one-GPU rehearsal).

Sharding, weak scaling — every rank brings its own batch of B queries per step:
  * IVF lists are owned by ranks (whole lists, largest-first greedy balance); centroids are replicated.
  * each rank ranks the centroids for its OWN queries; exchange 1: all-gather of the query batches with their
    probe lists (B*(d+nprobe)*4 bytes per rank) — every rank needs every query, because each scans the probed
    lists IT owns for ALL world*B queries and emits a partial top-k with the selection keys (distance bits << 32 |
    global scan position).
  * The HNSW graph is replicated: each rank searches it for its OWN B queries only (no exchange).
  * exchange 2: all-gather of the partial (keys, ids) = world*B*k*16 bytes per rank (B=1024, k=10, 8 ranks:
    1.3 MB) — one fused collective; each rank then keeps the world partials of its own queries.  (An all-to-all
    would move 1/world of that; at these sizes the collective is latency-bound either way.)
  * keys are unique across ranks => the world-way merge by key is exactly the single-GPU result; then the
    reference's hybrid merge with the rank's HNSW results.
Per rank the IVF work is that of a single GPU holding the whole index for B queries (1/world of the lists,
world times the queries), so queries/s grow with the number of ranks.
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
        self._host_collectives = dist.get_backend() != "nccl"  # gloo: collectives on CPU copies (rehearsal only)
        self._hnsw = hyb.hnsw()  # one wrapper object: it remembers the searches in flight per slot

    def bulk_insert(self, ids, x, ts, now):
        self.owner = self.hyb.bulk_insert_sharded(ids, x, ts, now, self.rank, self.world)
        self.d = x.shape[1]

    def _tensor(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = self.torch.empty(shape, dtype=dtype, device="cuda")
            self._bufs[name] = t
        return t

    def _all_gather(self, out, inp):
        """out[world * n] <- concatenation over ranks of inp[n] (device tensors)."""
        if self._host_collectives:
            o = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, inp.cpu())
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp)

    def upload_queries(self, q):
        """This rank's B x d f32 query batch as a device tensor (kept by the caller across steps)."""
        return self.torch.from_numpy(np.ascontiguousarray(q, np.float32)).cuda()

    SLOTS = 8

    def _slot_ctx(self, slot):
        """Engine context (stream) of the slot's IVF chain: slot 0 = the index's own."""
        if slot == 0:
            return self.hyb.ctx
        c = self._bufs.get(("ctx", slot))
        if c is None:
            c = self.fv.Context(self.hyb.ctx.device)
            self._bufs[("ctx", slot)] = c
        return c

    def search_dev_begin(self, slot, q_local, B, k, now, ef, nprobe):
        """Enqueue this rank's step in `slot`: gather everyone's queries, start the IVF chain over the lists this rank
        owns (slot's stream and scratch) and the graph walk of its own queries.  Every rank must call begin/end in the
        same order (each contains one collective)."""
        torch, W, d = self.torch, self.world, self.d
        import ctypes as C
        ivf, hnsw, ctx = self.hyb.ivf(), self._hnsw, self.hyb.ctx
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        sc = self._slot_ctx(slot)
        on = None if slot == 0 else sc.h
        npb = min(nprobe, self.hyb.n_clusters)
        # coarse stage for this rank's own queries only; the probe lists travel with the queries in ONE collective
        probes = self._tensor(("probes", slot), (B, npb), torch.int32)
        ctx.check(ctx.lib.fvdb_ivf_coarse_dev_slot(ivf._dev(), on, slot, p(q_local), B, nprobe, p(probes)))
        sc.synchronize()
        pack = self._tensor(("pack", slot), (B, d + npb), torch.int32)
        pack[:, :d].copy_(q_local.view(torch.int32))
        pack[:, d:].copy_(probes)
        pack_all = self._tensor(("pack_all", slot), (W * B, d + npb), torch.int32)
        self._all_gather(pack_all.view(-1), pack.view(-1))
        q_all = self._tensor(("q_all", slot), (W * B, d), torch.float32)
        q_all.view(torch.int32).copy_(pack_all[:, :d])
        probes_all = self._tensor(("probes_all", slot), (W * B, npb), torch.int32)
        probes_all.copy_(pack_all[:, d:])
        torch.cuda.current_stream().synchronize()  # the engine runs on its own streams; other slots keep running
        keys = self._tensor(("keys", slot), (W * B, k), torch.int64)
        ids = self._tensor(("ids", slot), (W * B, k), torch.int64)
        ds = self._tensor(("ds", slot), (W * B, k), torch.float32)
        cnt = self._tensor(("cnt", slot), (W * B,), torch.int32)
        # list scan over the lists this rank owns, for every rank's queries, with the probe lists they came with
        ctx.check(ctx.lib.fvdb_ivf_search_probes_dev_slot(ivf._dev(), on, slot, p(q_all), p(probes_all), W * B, k, nprobe,
                                                          p(ids), p(ds), p(cnt), p(keys)))
        hnsw.search_dev_begin(slot, p(q_local), B, d, k, ef)
        self._bufs[("state", slot)] = (q_local, B, k)

    def search_dev_end(self, slot):
        """Collect the step of `slot`: wait for its walk and chain, exchange the partial top-k, merge."""
        torch, fv, W = self.torch, self.fv, self.world
        import ctypes as C
        q_local, B, k = self._bufs.pop(("state", slot))
        ivf, hnsw, ctx = self.hyb.ivf(), self._hnsw, self.hyb.ctx
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        h = hnsw.search_dev_end(slot)
        self._slot_ctx(slot).synchronize()
        if slot == 0:
            ctx.lib.fvdb_ivf_profile_collect(ivf._dev())  # stage timing, when profiling is on
        keys, ids = self._bufs[("keys", slot)], self._bufs[("ids", slot)]
        mine = self._tensor(("mine", slot), (2, W * B, k), torch.int64)
        mine[0].copy_(keys)
        mine[1].copy_(ids)
        allb = self._tensor(("all", slot), (W, 2, W, B, k), torch.int64)
        self._all_gather(allb.view(-1), mine.view(-1))
        gk = self._tensor(("gk", slot), (W, B, k), torch.int64)
        gi = self._tensor(("gi", slot), (W, B, k), torch.int64)
        gk.copy_(allb[:, 0, self.rank])
        gi.copy_(allb[:, 1, self.rank])
        torch.cuda.current_stream().synchronize()
        oi = self._tensor(("oi", slot), (B, k), torch.int64)
        od = self._tensor(("od", slot), (B, k), torch.float32)
        oc = self._tensor(("oc", slot), (B,), torch.int32)
        sc = self._slot_ctx(slot)
        fv.engine.merge_keys_dev(sc, p(gk), p(gi), W, B, k, p(oi), p(od), p(oc))
        sc.synchronize()
        i_ids = oi.cpu().numpy().view(np.uint64)
        i_ds = od.cpu().numpy()
        i_cnt = oc.cpu().numpy().view(np.uint32)
        return _Res(*hybrid_merge(h.ids, h.distances, h.counts, i_ids, i_ds, i_cnt, k))

    def search_dev(self, q_local, B, k, now, ef, nprobe):
        """q_local: this rank's B x d f32 queries (device tensor).  Returns this rank's results."""
        torch, fv, W = self.torch, self.fv, self.world
        import ctypes as C
        import os
        import sys
        import time
        timing = os.environ.get("FVDB_SHARDED_TIMING")  # diagnostic: per-phase host time of every call, rank 0
        tp = [time.perf_counter()]
        mark = (lambda: tp.append(time.perf_counter())) if timing else (lambda: None)

        ivf, hnsw, ctx = self.hyb.ivf(), self.hyb.hnsw(), self.hyb.ctx
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        d = self.d
        # 1. every rank needs every query
        q_all = self._tensor("q_all", (W * B, d), torch.float32)
        self._all_gather(q_all.view(-1), q_local.reshape(-1))
        torch.cuda.synchronize()  # the engine runs on its own streams
        mark()
        # 2. IVF partial for ALL world*B queries over the lists this rank owns (async on the engine's stream)
        keys = self._tensor("keys", (W * B, k), torch.int64)
        ids = self._tensor("ids", (W * B, k), torch.int64)
        ds = self._tensor("ds", (W * B, k), torch.float32)
        cnt = self._tensor("cnt", (W * B,), torch.int32)
        ctx.check(ctx.lib.fvdb_ivf_search_dev(ivf._dev(), p(q_all), W * B, k, nprobe, p(ids), p(ds), p(cnt), p(keys)))
        # 3. HNSW for this rank's own queries, beside the IVF chain on the GPU
        h = hnsw.search_dev(p(q_local), B, d, k, ef)
        ctx.synchronize()
        ctx.lib.fvdb_ivf_profile_collect(ivf._dev())  # stage timing, when profiling is on
        mark()
        # 4. one collective carrying every rank's (keys, ids) for every query; keep the rows of my queries
        mine = self._tensor("mine", (2, W * B, k), torch.int64)
        mine[0].copy_(keys)
        mine[1].copy_(ids)
        allb = self._tensor("all", (W, 2, W, B, k), torch.int64)
        self._all_gather(allb.view(-1), mine.view(-1))
        gk = self._tensor("gk", (W, B, k), torch.int64)
        gi = self._tensor("gi", (W, B, k), torch.int64)
        gk.copy_(allb[:, 0, self.rank])
        gi.copy_(allb[:, 1, self.rank])
        torch.cuda.synchronize()
        mark()
        # 5. world-way merge by key on the GPU (fvdb_merge_keys_dev), then the reference's hybrid merge
        oi = self._tensor("oi", (B, k), torch.int64)
        od = self._tensor("od", (B, k), torch.float32)
        oc = self._tensor("oc", (B,), torch.int32)
        fv.engine.merge_keys_dev(ctx, p(gk), p(gi), W, B, k, p(oi), p(od), p(oc))
        ctx.synchronize()
        mark()
        i_ids = oi.cpu().numpy().view(np.uint64)
        i_ds = od.cpu().numpy()
        i_cnt = oc.cpu().numpy().view(np.uint32)
        mark()
        res = _Res(*hybrid_merge(h.ids, h.distances, h.counts, i_ids, i_ds, i_cnt, k))
        if timing and self.rank == 0:
            mark()
            names = ("gather queries", "ivf+hnsw", "gather partials", "merge kernel", "copies to host", "hybrid merge")
            print(f"[sharded] t_in {tp[0] % 10:.6f} t_out {tp[-1] % 10:.6f} " + ", ".join(f"{n} {1e3 * (b - a):.3f}" for n, a, b in zip(names, tp, tp[1:])) + " ms",
                  file=sys.stderr, flush=True)
        return res
