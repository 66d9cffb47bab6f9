"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle still reproduces them bit for bit (pins the oracle against silent regressions).
GPU: the HIP path, through the C ABI and the host mirror, reproduces them bit for bit.
"""
import os

import numpy as np
import pytest

import oracle as orc
from _data import bits

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DAY = 86400.0


def load(name):
    return np.load(os.path.join(G, name))


def same(got, f):
    gi, gd, gc = got
    assert np.array_equal(gc, f["out_counts"])
    for b in range(gi.shape[0]):
        n = int(gc[b])
        assert np.array_equal(gi[b, :n], f["out_ids"][b, :n]), f"query {b}"
        assert np.array_equal(bits(gd[b, :n]), bits(f["out_dist"][b, :n])), f"query {b}"


# ---- CPU: oracle vs fixtures -----------------------------------------------------------------
@pytest.fixture(scope="module")
def built():
    orc.build()


def test_oracle_reproduces_ivf_fixture(built):
    f = load("ivf_small.npz")
    ix = orc.IVFIndex(n_clusters=f["centroids"].shape[0], n_probe=int(f["nprobe"]))
    ix.set_trained(f["centroids"])
    ix.batch_insert(f["ids"], f["x"])
    for dead in f["deleted"]:
        ix.mark_deleted(int(f["ids"][dead]))
    assert np.array_equal(ix.assign(f["x"]), f["assign"])
    same(ix.batch_search(f["queries"], int(f["k"]), int(f["nprobe"])), f)


def test_oracle_reproduces_hnsw_fixture(built):
    f = load("hnsw_small.npz")
    ix = orc.HNSWIndex(max_connections=8, max_connections_layer_0=16, ef_construction=40, seed=21)
    ix.batch_insert(np.arange(f["x"].shape[0], dtype=np.uint64), f["x"], f["levels"])
    same(ix.batch_search(f["queries"], int(f["k"]), int(f["ef"])), f)


def test_oracle_reproduces_hybrid_fixture(built):
    f = load("hybrid_small.npz")
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=f["centroids"].shape[0], n_probe=3)
    ix = orc.HybridIndex(**kw)
    ix.set_ivf_centroids(f["centroids"])
    now = float(f["now"])
    for i in range(f["x"].shape[0]):
        ix.insert_with_timestamp(i, f["x"][i], now - f["ages"][i], now, int(f["levels"][i]))
    same(ix.batch_search(f["queries"], int(f["k"]), now=now, hnsw_ef=int(f["ef"]), ivf_n_probe=int(f["nprobe"])), f)


# ---- GPU: HIP path vs fixtures ---------------------------------------------------------------
@pytest.fixture(scope="module")
def fv():
    import fvdb_import
    return fvdb_import.load()


@pytest.fixture(scope="module")
def ctx(fv):
    c = fv.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_reproduces_ivf_fixture(fv, ctx):
    f = load("ivf_small.npz")
    gpu = fv.DeviceIVF(ctx, f["x"].shape[1], f["centroids"].shape[0])
    gpu.set_centroids(f["centroids"])
    cl, pos = gpu.add(f["x"], f["ids"])
    assert np.array_equal(cl, f["assign"])
    d = f["deleted"].astype(np.int64)
    gpu.set_deleted(cl[d], pos[d], True)
    same(gpu.search(f["queries"], int(f["k"]), int(f["nprobe"])), f)
    # the same 24 queries repeated to a batch of 48: the matrix-core path (>= 32 queries) gives the same rows
    q2 = np.concatenate([f["queries"], f["queries"]])
    gi, gd, gc = gpu.search(q2, int(f["k"]), int(f["nprobe"]))
    same((gi[:24], gd[:24], gc[:24]), f)
    same((gi[24:], gd[24:], gc[24:]), f)


@pytest.mark.gpu
def test_gpu_reproduces_hnsw_fixture(fv, ctx):
    f = load("hnsw_small.npz")
    ix = fv.HNSWIndex(ctx, 8, 16, 40, seed=21)
    for i in range(f["x"].shape[0]):
        ix.insert(i, f["x"][i], level=int(f["levels"][i]))
    for mode in (True, False):  # device traversal, host walk
        ix.set_device_traversal(mode)
        r = ix.search(f["queries"], int(f["k"]), int(f["ef"]))
        same((r.ids, r.distances, r.counts), f)


@pytest.mark.gpu
def test_gpu_reproduces_hybrid_fixture(fv, ctx):
    f = load("hybrid_small.npz")
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=f["centroids"].shape[0], n_probe=3)
    ix = fv.HybridIndex(ctx, **kw)
    ix.set_ivf_centroids(f["centroids"])
    now = float(f["now"])
    for i in range(f["x"].shape[0]):
        ix.insert_with_timestamp(i, f["x"][i], now - f["ages"][i], now, int(f["levels"][i]))
    r = ix.search(f["queries"], int(f["k"]), now=now, hnsw_ef=int(f["ef"]), ivf_n_probe=int(f["nprobe"]))
    same((r.ids, r.distances, r.counts), f)
