"""Generates the golden fixtures in this directory from the CPU oracle (oracle/oracle.cpp).

The reference is Rust and cannot be built or imported here, and its own tests hold no numeric
vectors for this path, so these fixtures are the ORACLE's outputs on small seeded inputs: they pin the oracle against
silent regressions (tests/test_golden.py, CPU) and give the HIP path a second, frozen target (GPU).  Data only —
inputs and expected outputs.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
from _data import mixture  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DAY = 86400.0


def ivf_case():
    n, d, nlist, k, nprobe = 600, 24, 8, 5, 3
    x = mixture(n, d, n_comp=8, seed=901)
    ids = np.arange(n, dtype=np.uint64) * 5 + 2
    cents = x[:nlist].copy()
    q = mixture(24, d, n_comp=8, seed=902)
    ix = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
    ix.set_trained(cents)
    ix.batch_insert(ids, x)
    for dead in (7, 102, 311):  # soft deletes (src/ivf/operations.rs:569-591)
        ix.mark_deleted(int(ids[dead]))
    oi, od, oc = ix.batch_search(q, k, nprobe)
    np.savez_compressed(os.path.join(HERE, "ivf_small.npz"), x=x, ids=ids, centroids=cents, queries=q, k=k, nprobe=nprobe,
                        deleted=np.array([7, 102, 311]), assign=ix.assign(x), out_ids=oi, out_dist=od, out_counts=oc)


def hnsw_case():
    n, d, k, ef = 300, 16, 5, 20
    x = mixture(n, d, n_comp=6, seed=911)
    levels = orc.rng_levels(21, n)
    q = mixture(20, d, n_comp=6, seed=912)
    ix = orc.HNSWIndex(max_connections=8, max_connections_layer_0=16, ef_construction=40, seed=21)
    ix.batch_insert(np.arange(n, dtype=np.uint64), x, levels)
    oi, od, oc = ix.batch_search(q, k, ef)
    np.savez_compressed(os.path.join(HERE, "hnsw_small.npz"), x=x, levels=np.asarray(levels), queries=q, k=k, ef=ef,
                        out_ids=oi, out_dist=od, out_counts=oc)


def hybrid_case():
    n, d, nlist, k = 400, 20, 6, 8
    x = mixture(n, d, n_comp=6, seed=921)
    cents = x[:nlist].copy()
    now = 1000 * DAY
    rng = np.random.default_rng(3)
    ages = np.where(rng.random(n) < 0.3, 1 * DAY, 30 * DAY)
    levels = orc.rng_levels(22, n)
    q = mixture(16, d, n_comp=6, seed=922)
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=3)
    ix = orc.HybridIndex(**kw)
    ix.set_ivf_centroids(cents)
    for i in range(n):
        ix.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
    oi, od, oc = ix.batch_search(q, k, now=now, hnsw_ef=30, ivf_n_probe=3)
    np.savez_compressed(os.path.join(HERE, "hybrid_small.npz"), x=x, centroids=cents, ages=ages, levels=np.asarray(levels),
                        queries=q, k=k, now=now, ef=30, nprobe=3, out_ids=oi, out_dist=od, out_counts=oc)


if __name__ == "__main__":
    orc.build()
    ivf_case()
    hnsw_case()
    hybrid_case()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
