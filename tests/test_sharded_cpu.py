"""Multi-rank logic on CPU: shard planning, the hybrid merge, the all-gather packing, and a
world_size-2 gloo run of the sharded IVF scheme (per-rank partial top-k with global keys,
one all-gather, merge by key) checked against the single-index oracle."""
import os
import socket

import numpy as np
import pytest

import fvdb_import
import oracle as orc
from _data import mixture

fv = fvdb_import.load()
sh = fv.sharded


def test_plan_list_shards_balanced_and_deterministic():
    rng = np.random.default_rng(0)
    sizes = rng.integers(0, 5000, 1024)
    for world in (1, 2, 4, 8):
        owner = sh.plan_list_shards(sizes, world)
        assert owner.max() < world and np.array_equal(owner, sh.plan_list_shards(sizes, world))
        loads = np.bincount(owner, weights=sizes, minlength=world)
        assert loads.max() - loads.min() <= sizes.max()  # LPT bound


def test_hybrid_merge_is_the_reference_merge():
    rng = np.random.default_rng(1)
    B, k = 50, 7
    h_ids = rng.integers(0, 1000, (B, k)).astype(np.uint64)
    i_ids = rng.integers(1000, 2000, (B, k)).astype(np.uint64)
    h_ds = np.sort(rng.integers(0, 6, (B, k)).astype(np.float32), axis=1)  # many exact ties
    i_ds = np.sort(rng.integers(0, 6, (B, k)).astype(np.float32), axis=1)
    h_cnt = rng.integers(0, k + 1, B).astype(np.uint32)
    i_cnt = rng.integers(0, k + 1, B).astype(np.uint32)
    ids, ds, cnt = sh.hybrid_merge(h_ids, h_ds, h_cnt, i_ids, i_ds, i_cnt, k)
    for b in range(B):
        allr = [(h_ds[b, i], h_ids[b, i]) for i in range(h_cnt[b])] + [(i_ds[b, i], i_ids[b, i]) for i in range(i_cnt[b])]
        allr.sort(key=lambda t: t[0])  # python sort is stable: HNSW first on ties (src/hybrid/core.rs:482)
        allr = allr[:k]
        assert cnt[b] == len(allr)
        assert [t[1] for t in allr] == ids[b, :cnt[b]].tolist()
        assert [t[0] for t in allr] == ds[b, :cnt[b]].tolist()


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(2)
    B, k, world = 10, 4, 3
    slices, per = sh.query_slices(B, world)
    bufs, truth = [], []
    for r in range(world):
        keys = rng.integers(0, 2**63, (B, k)).astype(np.uint64)
        ids = rng.integers(0, 2**63, (B, k)).astype(np.uint64)
        lo, hi = slices[r]
        hi_, hd_ = rng.integers(0, 99, (hi - lo, k)).astype(np.uint64), rng.random((hi - lo, k)).astype(np.float32)
        hc_ = rng.integers(0, k + 1, hi - lo).astype(np.uint32)
        bufs.append(sh.pack_partials(keys, ids, hi_, hd_, hc_, per, k)[0])
        truth.append((keys, ids, hi_, hd_, hc_))
    keys, ids, h_ids, h_ds, h_cnt = sh.unpack_partials(np.concatenate(bufs), world, B, per, k)
    for r in range(world):
        lo, hi = slices[r]
        assert np.array_equal(keys[r], truth[r][0]) and np.array_equal(ids[r], truth[r][1])
        assert np.array_equal(h_ids[lo:hi], truth[r][2]) and np.array_equal(h_ds[lo:hi], truth[r][3])
        assert np.array_equal(h_cnt[lo:hi], truth[r][4])


# ---- world_size 2 over gloo -----------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, nlist, B, k, nprobe = 3000, 16, 24, 21, 5, 6
    x = mixture(n, d, n_comp=10, seed=70)
    q = mixture(B, d, n_comp=10, seed=71)
    cents = x[:nlist].copy()
    full = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
    full.set_trained(cents)
    ids = np.arange(n, dtype=np.uint64) + 5
    full.batch_insert(ids, x)
    clusters = full.assign(x)
    sizes = np.bincount(clusters, minlength=nlist)
    owner = sh.plan_list_shards(sizes, world)
    # this rank's partial result: rows of probed lists it owns, key = (distance bits << 32 | global seq)
    keys = np.full((B, k), 2**64 - 1, np.uint64)
    pids = np.full((B, k), 2**64 - 1, np.uint64)
    for b in range(B):
        cd = orc.l2_batch(q[b], cents)
        probes = np.argsort(cd, kind="stable")[:nprobe]
        cand = []
        base = 0
        for L in probes:
            members = np.nonzero(clusters == L)[0]  # insertion order = position in list
            if owner[L] == rank and members.size:
                dist_ = orc.l2_batch(q[b], x[members])
                for pos, (m, dv) in enumerate(zip(members, dist_)):
                    cand.append(((int(np.float32(dv).view(np.uint32)) << 32) | (base + pos), int(ids[m])))
            base += -(-int(sizes[L]) // 64) * 64  # seq advances by the padded logical list length
        cand.sort()
        for i, (key, id_) in enumerate(cand[:k]):
            keys[b, i], pids[b, i] = key, id_
    slices, per = sh.query_slices(B, world)
    lo, hi = slices[rank]
    empty = (np.empty((hi - lo, k), np.uint64), np.empty((hi - lo, k), np.float32), np.zeros(hi - lo, np.uint32))
    buf, _ = sh.pack_partials(keys, pids, *empty, per, k)
    mine = torch.from_numpy(buf.copy())
    allb = torch.empty(world * mine.numel(), dtype=torch.int64)
    dist.all_gather_into_tensor(allb, mine)
    gk, gi, _, _, _ = sh.unpack_partials(allb.numpy(), world, B, per, k)
    # merge by key (what fvdb_merge_keys_dev does on the GPU)
    ok = True
    for b in range(B):
        allc = sorted((int(gk[r, b, i]), int(gi[r, b, i])) for r in range(world) for i in range(k)
                      if gk[r, b, i] != np.uint64(2**64 - 1))[:k]
        want = full.search(q[b], k, nprobe)
        got_ids = [c[1] for c in allc]
        got_ds = [np.uint32(c[0] >> 32).view(np.float32) for c in allc]
        ok &= got_ids == want.ids.tolist() and got_ds == want.distances.tolist()
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ivf_scheme_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    orc.build()
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
