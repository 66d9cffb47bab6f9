"""Multi-GPU execution of the hybrid search: one process per GPU, inverted lists owned by ranks, RCCL over xGMI.

All of the data path lives below Python, in the C ABI (include/fvdb.h: fvdb_comm_*, fvdb_ivf_search_sharded_begin/_end)
and in the C++ host mirror (HybridIndex::attach_comm / search_sharded_begin / search_sharded_end): centroid ranking,
exchange 1 (all-gather of queries + probe lists), the scan of the lists the rank owns, exchange 2 (all-to-all of the
partial (key, id) lists), the world-way merge by key and the reference's hybrid merge with the replicated graph's
results — on device streams, with no host synchronisation or host copy between the stages.  This module only
  * bootstraps the communicator: rank 0 draws the RCCL unique id (fvdb_comm_unique_id), `torch.distributed` (any
    backend; the CPU `gloo` group is enough) carries its 128 bytes to the other ranks, every rank calls
    fvdb_comm_create (ncclCommInitRank);
  * offers the hosted transport used by the tests and by rehearsals of several ranks on ONE GPU: the same C code
    path, with the two exchanges carried over `torch.distributed` on host buffers (RCCL cannot put two ranks on one
    device);
  * wraps the calls (ShardedHybrid).

Sharding: lists are placed largest-first on the least loaded rank (plan_list_shards = the host
mirror's plan_list_owners), centroids and the HNSW graph are replicated, keys (distance bits << 32 | global scan
position) are unique across ranks, so the merged result is exactly the single-GPU one.  Modes: WEAK — every rank
brings its own batch of B queries per step (global batch world*B); STRONG — the global batch is fixed at B, every
rank holds all of it and produces the results of its slice.  The reference has no distributed execution at all.
"""
import ctypes as C

import numpy as np

from ._capi import f32p, u32p, u64p
from .index import load_host

WEAK, STRONG = 0, 1
NO_ID = np.uint64(0xFFFFFFFFFFFFFFFF)


def _p(a, t):
    return a.ctypes.data_as(t)


def plan_list_shards(list_sizes, world):
    """owner[list]: largest list first to the least loaded rank (ties -> lower rank / list id): the host mirror's
    plan_list_owners (host/hybrid_index.cpp), the placement bulk_insert_sharded uses."""
    sizes = np.ascontiguousarray(list_sizes, np.uint64)
    owner = np.zeros(sizes.size, np.uint32)
    load_host().fvh_plan_list_owners(_p(sizes, u64p), sizes.size, world, _p(owner, u32p))
    return owner


def hybrid_merge(h_ids, h_ds, h_cnt, i_ids, i_ds, i_cnt, k):
    """HybridIndex::search_with_config's merge (src/hybrid/core.rs:476-485) as the host mirror runs it (merge_parts):
    HNSW results then IVF results, stable sort by distance, truncate(k); batched."""
    h_ids, i_ids = np.ascontiguousarray(h_ids, np.uint64), np.ascontiguousarray(i_ids, np.uint64)
    h_ds, i_ds = np.ascontiguousarray(h_ds, np.float32), np.ascontiguousarray(i_ds, np.float32)
    h_cnt, i_cnt = np.ascontiguousarray(h_cnt, np.uint32), np.ascontiguousarray(i_cnt, np.uint32)
    B = h_ids.shape[0]
    ids = np.empty((B, k), np.uint64)
    ds = np.empty((B, k), np.float32)
    cnt = np.zeros(B, np.uint32)
    load_host().fvh_merge_parts(B, k, h_ids.shape[1], i_ids.shape[1], _p(h_ids, u64p), _p(h_ds, f32p), _p(h_cnt, u32p),
                                _p(i_ids, u64p), _p(i_ds, f32p), _p(i_cnt, u32p), _p(ids, u64p), _p(ds, f32p), _p(cnt, u32p))
    return ids, ds, cnt


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


def hosted_exchange(dist, torch, world, rank):
    """The hosted transport's exchange function (fvdb_exchange_fn) over torch.distributed on host buffers.
    op 0: all-gather (send `bytes`, recv world blocks); op 1: all-to-all (block p of send goes to rank p) — done as an
    all-gather of everything followed by picking this rank's column, which every backend supports."""

    def exchange(_user, op, send, recv, nbytes):
        try:
            n_send = nbytes if op == 0 else world * nbytes
            src = torch.frombuffer((C.c_uint8 * n_send).from_address(send), dtype=torch.uint8)
            if op == 0:
                out = torch.empty(world * nbytes, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, src.clone())
            else:
                allb = torch.empty(world * world * nbytes, dtype=torch.uint8)
                dist.all_gather_into_tensor(allb, src.clone())
                out = allb.view(world, world, nbytes)[:, rank, :].contiguous().view(-1)
            C.memmove(recv, out.data_ptr(), world * nbytes)
            return 0
        except Exception:  # noqa: BLE001 — must not unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1

    return _EXCHANGE_FN(exchange)


class Comm:
    """fvdb_comm: `Comm.rccl(ctx, dist, torch)` (RCCL; the id travels over `dist`) or `Comm.hosted(...)`."""

    def __init__(self, ctx, handle, world, rank, keep=None):
        self.ctx, self.h, self.world, self.rank, self._keep = ctx, handle, world, rank, keep

    @classmethod
    def rccl(cls, ctx, dist=None, torch=None, world=1, rank=0):
        lib = ctx.lib
        buf = np.zeros(129, np.uint8)  # 128 bytes of id + 1 status byte (1 = rank 0 could not draw an id)
        if dist is not None and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
        if rank == 0:
            rc = lib.fvdb_comm_unique_id(buf.ctypes.data_as(C.c_void_p))
            if rc:
                buf[128] = 1
        if world > 1:
            # always broadcast, also after a failure on rank 0: every rank must issue the same sequence of `dist` calls
            t = torch.from_numpy(buf)
            if dist.get_backend() == "nccl":
                t = t.cuda()
                dist.broadcast(t, 0)
                buf = t.cpu().numpy()
            else:
                dist.broadcast(t, 0)
        if buf[128]:
            raise RuntimeError("fvdb_comm_unique_id failed on rank 0 (librccl missing?)")
        h = C.c_void_p()
        ctx.check(lib.fvdb_comm_create(ctx.h, buf.ctypes.data_as(C.c_void_p), world, rank, C.byref(h)))
        return cls(ctx, h, world, rank)

    @classmethod
    def hosted(cls, ctx, dist, torch):
        world, rank = dist.get_world_size(), dist.get_rank()
        fn = hosted_exchange(dist, torch, world, rank)
        h = C.c_void_p()
        ctx.check(ctx.lib.fvdb_comm_create_hosted(ctx.h, world, rank, C.cast(fn, C.c_void_p), None, C.byref(h)))
        return cls(ctx, h, world, rank, keep=fn)

    @classmethod
    def loopback(cls, ctx, world, rank=0):
        """Capacity planning on one GPU: rank `rank` of a pretended `world`-rank job.  Development builds only
        (include/fvdb_dev.h; `make -C fabstir-vectordb_amd dev`, FVDB_LIB_DIR=lib_dev) — the product library has no such entry point."""
        fn = getattr(ctx.lib, "fvdb_comm_create_loopback", None)
        if fn is None:
            raise RuntimeError("the loopback communicator exists in the dev build only: make -C fabstir-vectordb_amd dev, FVDB_LIB_DIR=lib_dev")
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        ctx.check(fn(ctx.h, world, rank, C.byref(h)))
        return cls(ctx, h, world, rank)

    def all_gather_dev(self, send_dev, recv_dev, nbytes, on=None):
        self.ctx.check(self.ctx.lib.fvdb_comm_all_gather_dev(self.h, on, send_dev, recv_dev, nbytes))

    def all_to_all_dev(self, send_dev, recv_dev, nbytes, on=None):
        self.ctx.check(self.ctx.lib.fvdb_comm_all_to_all_dev(self.h, on, send_dev, recv_dev, nbytes))

    def close(self):
        if self.h:
            self.ctx.lib.fvdb_comm_destroy(self.h)
            self.h = None


def self_test(comm):
    """One all-gather and one all-to-all of rank-stamped words through `comm`, checked on the host: the bring-up check
    a launcher runs before trusting a fresh communicator with the data path.  Raises on a wrong word."""
    ctx, W, r = comm.ctx, comm.world, comm.rank
    n = 64  # u32 words per block
    send_g = np.full(n, 0xA000 + r, np.uint32)
    send_a = (0xB000 + 256 * r + np.repeat(np.arange(W, dtype=np.uint32), n)).astype(np.uint32)  # block p -> rank p
    d_sg, d_sa = ctx.upload(send_g), ctx.upload(send_a)
    d_rg, d_ra = ctx.alloc(4 * n * W), ctx.alloc(4 * n * W)
    try:
        comm.all_gather_dev(d_sg, d_rg, 4 * n)
        comm.all_to_all_dev(d_sa, d_ra, 4 * n)
        ctx.device_synchronize()
        got_g = ctx.download(d_rg, (W, n), np.uint32)
        got_a = ctx.download(d_ra, (W, n), np.uint32)
    finally:
        for p in (d_sg, d_sa, d_rg, d_ra):
            ctx.free(p)
    want_g = 0xA000 + np.arange(W, dtype=np.uint32)[:, None] + np.zeros((1, n), np.uint32)
    want_a = 0xB000 + 256 * np.arange(W, dtype=np.uint32)[:, None] + r + np.zeros((1, n), np.uint32)
    if not (np.array_equal(got_g, want_g) and np.array_equal(got_a, want_a)):
        raise RuntimeError(f"rank {r}: communicator self-test moved wrong data")


class BringUpFailed(RuntimeError):
    """RCCL could not be brought up on every rank and the caller did not allow the hosted transport."""


def bring_up(ctx, dist, torch, transport="rccl", timeout_s=180.0, log=print, allow_hosted=False):
    """The communicator a multi-rank launcher should use: RCCL, created and self-tested under a watchdog.  When any rank
    fails (or does not finish in `timeout_s`) EVERY rank raises BringUpFailed — a multi-GPU number can then only come
    from RCCL — unless `allow_hosted`, in which case every rank switches to the hosted transport (same C data path, the
    exchanges carried by `dist` on host buffers) and says so loudly.  transport="hosted" asks for that transport
    outright and also needs `allow_hosted`.  Returns (comm, transport actually in use)."""
    if transport == "hosted":
        if not allow_hosted:
            raise BringUpFailed("the hosted transport is a rehearsal aid: pass allow_hosted / --allow-hosted to use it")
        return Comm.hosted(ctx, dist, torch), "hosted"
    import threading
    box = {}

    def work():
        try:
            c = Comm.rccl(ctx, dist, torch)
            self_test(c)
            box["comm"] = c
        except Exception as e:  # noqa: BLE001 — reported below, on every rank
            box["err"] = repr(e)

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    ok = 1 if ("comm" in box and not t.is_alive()) else 0
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return box["comm"], "rccl"
    why = box.get("err", "timed out" if t.is_alive() else "another rank failed")
    if not allow_hosted:
        raise BringUpFailed(f"RCCL bring-up FAILED on at least one rank (this rank: {why})")
    log(f"RCCL bring-up FAILED on at least one rank (this rank: {why}); every rank continues on the HOSTED transport")
    return Comm.hosted(ctx, dist, torch), "hosted (RCCL bring-up failed)"


class ShardedHybrid:
    """HybridIndex across `comm.world` ranks (see the module docstring).  Bench / scale surface: bulk placement and
    batched search with several steps in flight; the per-search auto-migration (src/hybrid/core.rs:437-439) runs on
    every rank when every rank passes the same `now` (the owner of a list appends the aged row, every rank counts it)."""

    SLOTS = 16

    def __init__(self, hyb, comm):
        self.hyb, self.comm = hyb, comm
        self.rank, self.world = comm.rank, comm.world
        self.owner = None
        self._attached = False

    def bulk_insert(self, ids, x, ts, now):
        self.owner = self.hyb.bulk_insert_sharded(ids, x, ts, now, self.rank, self.world)
        self.d = x.shape[1]
        self.hyb.attach_comm(self.comm.h)
        self._attached = True

    def rows(self, B, mode=WEAK):
        """Result rows this rank gets for a step of B queries."""
        return self.hyb.sharded_rows(B, mode)

    def search_dev_begin(self, slot, q_dev, B, k, ef, nprobe, mode=WEAK, now=0.0):
        """Enqueue this rank's step in `slot` (q_dev: device pointer to B x d f32 — the rank's own batch in WEAK mode,
        the global batch in STRONG mode).  Every rank must call begin/end in the same order, with the same `now`."""
        self.hyb.search_sharded_begin(slot, q_dev, B, k, mode, hnsw_ef=ef, ivf_n_probe=nprobe, dim=self.d, now=now)

    def search_dev_end(self, slot):
        return self.hyb.search_sharded_end(slot)

    def search_dev(self, q_dev, B, k, ef, nprobe, mode=WEAK, now=0.0):
        self.search_dev_begin(0, q_dev, B, k, ef, nprobe, mode, now)
        return self.search_dev_end(0)
