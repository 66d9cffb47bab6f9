"""Multi-rank logic that runs without a GPU: the host mirror's list placement and hybrid merge (product C++, called
through libfvdb_host.so), and a world_size-2 gloo run of the sharded scheme in which the product's hosted-transport
exchange function carries both exchange steps between two processes.  The per-rank partial top-k (a GPU kernel in the
product) is supplied by the CPU oracle here; tests/test_00_gpu_sharded_ranks.py runs the same scheme end to end on
the GPU through the C ABI."""
import ctypes as C
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
    # ties: equal lists are dealt round-robin in list order, lowest rank first
    assert sh.plan_list_shards([5, 5, 5, 5], 2).tolist() == [0, 1, 0, 1]
    assert sh.plan_list_shards([1, 9, 1], 3).tolist() == [1, 0, 2]


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
        assert np.all(ids[b, cnt[b]:] == np.uint64(2**64 - 1)) and np.all(np.isinf(ds[b, cnt[b]:]))


# ---- world_size 2 over gloo -----------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exchange(fn, op, send, world):
    """Call the product's fvdb_exchange_fn the way the C library does: raw host pointers."""
    send = np.ascontiguousarray(send)
    nbytes = send.nbytes if op == 0 else send.nbytes // world
    recv = np.empty(world * nbytes, np.uint8)
    rc = fn(None, op, send.ctypes.data_as(C.c_void_p), recv.ctypes.data_as(C.c_void_p), nbytes)
    assert rc == 0
    return recv


def _rank_main(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fn = sh.hosted_exchange(dist, torch, world, rank)  # what Comm.hosted hands to fvdb_comm_create_hosted
    n, d, nlist, B, k, nprobe = 3000, 16, 24, 21, 5, 6
    x = mixture(n, d, n_comp=10, seed=70)
    cents = x[:nlist].copy()
    full = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
    full.set_trained(cents)
    ids = np.arange(n, dtype=np.uint64) + 5
    full.batch_insert(ids, x)
    clusters = full.assign(x)
    sizes = np.bincount(clusters, minlength=nlist)
    owner = sh.plan_list_shards(sizes, world)  # product placement
    # WEAK mode: every rank has its own queries; exchange 1 = all-gather of the query batches
    q_own = mixture(B, d, n_comp=10, seed=71 + rank)
    q_all = _exchange(fn, 0, q_own, world).view(np.float32).reshape(world * B, d)
    ok = all(np.array_equal(q_all[r * B:(r + 1) * B], mixture(B, d, n_comp=10, seed=71 + r)) for r in range(world))
    # this rank's partial result for ALL world*B queries: rows of probed lists it owns, key = (distance bits << 32 |
    # global seq) — the oracle stands in for the GPU scan
    keys = np.full((world * B, k), 2**64 - 1, np.uint64)
    pids = np.full((world * B, k), 2**64 - 1, np.uint64)
    for b in range(world * B):
        cd = orc.l2_batch(q_all[b], cents)
        probes = np.argsort(cd, kind="stable")[:nprobe]
        cand = []
        base = 0
        for L in probes:
            members = np.nonzero(clusters == L)[0]  # insertion order = position in list
            if owner[L] == rank and members.size:
                dist_ = orc.l2_batch(q_all[b], x[members])
                for pos, (m, dv) in enumerate(zip(members, dist_)):
                    cand.append(((int(np.float32(dv).view(np.uint32)) << 32) | (base + pos), int(ids[m])))
            base += -(-int(sizes[L]) // 64) * 64  # seq advances by the padded logical list length
        cand.sort()
        for i, (key, id_) in enumerate(cand[:k]):
            keys[b, i], pids[b, i] = key, id_
    # exchange 2 = all-to-all: block p (the partials of rank p's queries) goes to rank p
    gk = _exchange(fn, 1, keys, world).view(np.uint64).reshape(world, B, k)
    gi = _exchange(fn, 1, pids, world).view(np.uint64).reshape(world, B, k)
    # merge by key (fvdb_merge_keys_dev on the GPU), then the reference's hybrid merge through the host mirror with an
    # empty HNSW part
    for b in range(B):
        allc = sorted((int(gk[r, b, i]), int(gi[r, b, i])) for r in range(world) for i in range(k)
                      if gk[r, b, i] != np.uint64(2**64 - 1))[:k]
        want = full.search(q_own[b], k, nprobe)
        got_ids = np.asarray([c[1] for c in allc], np.uint64)
        got_ds = np.asarray([np.uint32(c[0] >> 32).view(np.float32) for c in allc], np.float32)
        m_ids, m_ds, m_cnt = sh.hybrid_merge(np.zeros((1, k), np.uint64), np.zeros((1, k), np.float32), np.zeros(1, np.uint32),
                                             np.pad(got_ids, (0, k - got_ids.size))[None], np.pad(got_ds, (0, k - got_ds.size))[None],
                                             np.asarray([got_ids.size], np.uint32), k)
        ok &= m_cnt[0] == len(want) and m_ids[0, :m_cnt[0]].tolist() == want.ids.tolist()
        ok &= m_ds[0, :m_cnt[0]].tolist() == want.distances.tolist()
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


def test_bench_launcher_starts_one_rank_per_gpu_as_a_child(monkeypatch):
    # `python bench.py --gpus N` with no WORLD_SIZE: the parent builds the torch.distributed.run command line (one rank
    # per GPU, rendezvous on 127.0.0.1), hands over its own arguments unchanged and returns the child's exit code —
    # without importing the engine (no GPU call in the parent)
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    before = set(sys.modules)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not any(m.startswith("fabstir_vectordb_amd") or m == "fvdb_import" for m in set(sys.modules) - before)
