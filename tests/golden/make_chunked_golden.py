"""Generates tests/golden/chunked_small/: a small hybrid index in the reference's chunked on-disk layout.

The index STATE comes from the CPU oracle (graph links, inverted lists, timestamps, soft deletes), the bytes from the
product's writer (fabstir-vectordb_amd/chunked.py: write_snapshot), and the expected answers from the oracle's own
search on that state — so the GPU test that opens these files checks the whole load path (CBOR decode, graph restore,
GPU re-assignment into lists, from_parts) against the oracle, and the CPU test pins the codec against silent changes
of the bytes.  Data only.  Run from the repo root:  python tests/golden/make_chunked_golden.py
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fvdb_import  # noqa: E402
import oracle as orc  # noqa: E402
from _data import mixture  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DAY = 86400.0
KW = dict(max_connections=6, max_connections_layer_0=12, ef_construction=30, n_clusters=5, n_probe=3)


def main():
    fv = fvdb_import.load()
    ck = fv.chunked
    orc.build()
    n, d, k, now = 240, 12, 6, 1000 * DAY
    x = mixture(n, d, n_comp=5, seed=931)
    x[17] = x[16]  # a duplicate pair inside one list: the tie is broken by list position, which a reload must keep
    q = mixture(12, d, n_comp=5, seed=932)
    id_bytes = np.stack([np.frombuffer(fv.blake3(f"doc-{i}".encode()), np.uint8) for i in range(n)])
    rid = np.asarray([ck.row_id(bytes(b)) for b in id_bytes], np.uint64)
    recent = np.random.default_rng(4).random(n) < 0.3
    recent[[16, 17]] = False
    ts = np.where(recent, now - 1 * DAY, now - 30 * DAY) - np.arange(n) * 0.125  # distinct, exactly representable
    levels = orc.rng_levels(23, n)
    ix = orc.HybridIndex(**KW)
    ix.set_ivf_centroids(x[:KW["n_clusters"]].copy())
    for i in range(n):
        ix.insert_with_timestamp(int(rid[i]), x[i], ts[i], now, int(levels[i]))
    dead_recent, dead_hist = int(np.flatnonzero(recent)[3]), int(np.flatnonzero(~recent)[5])
    ix.delete(int(rid[dead_recent]), now)
    # what a reload yields: the deleted graph node stays deleted, the soft-deleted list row is live again
    # (src/hybrid/persistence.rs:675-682 hashes the saved display strings again) -> expected answers taken here
    oi, od, oc = ix.batch_search(q, k, now=now, hnsw_ef=30, ivf_n_probe=3)
    ix.delete(int(rid[dead_hist]), now)
    h, iv = ix.hnsw(), ix.ivf()
    row_of = {int(r): i for i, r in enumerate(rid)}
    nodes = [int(rid[i]) for i in np.flatnonzero(recent)]
    off, nbrs = [0], []
    for r in nodes:
        for layer in range(h.level(r) + 1):
            nbrs += h.neighbors(r, layer)
            off.append(len(nbrs))
    lists = []
    for c in range(KW["n_clusters"]):
        ids = iv.list_ids(c)
        lists.append((x[[row_of[int(r)] for r in ids]], ids, np.asarray([int(r) != int(rid[dead_hist]) for r in ids], bool)))
    cfg = dict(recent_threshold=7 * DAY, migration_batch_size=100, auto_migrate=True, min_ivf_training_size=10, hnsw_seed=0,
               train_size=9, max_iterations=25, ivf_seed=0, **KW)
    snap = {"config": cfg, "recent_count": ix.recent_count(), "historical_count": ix.historical_count(), "ivf_trained": True,
            "node_ids": np.asarray(nodes, np.uint64), "node_levels": np.asarray([h.level(r) for r in nodes], np.uint32),
            "node_offsets": np.asarray(off, np.uint64), "node_neighbors": np.asarray(nbrs, np.uint64),
            "node_vectors": [x[row_of[r]] for r in nodes], "node_deleted": [r == int(rid[dead_recent]) for r in nodes],
            "entry_point": h.entry_point(), "centroids": x[:KW["n_clusters"]].copy(), "lists": lists,
            "timestamp_ids": rid, "timestamps": ts}
    out = os.path.join(HERE, "chunked_small")
    shutil.rmtree(out, ignore_errors=True)
    table = {int(r): bytes(b) for r, b in zip(rid, id_bytes)}
    ck.write_snapshot(snap, out, "idx", id_table=table, now=now, chunk_size=100)
    np.savez_compressed(os.path.join(out, "expected.npz"), x=x, id_bytes=id_bytes, row_ids=rid, recent=recent, timestamps=ts,
                        levels=np.asarray(levels), queries=q, k=k, now=now, ef=30, nprobe=3, dead_recent=dead_recent,
                        dead_hist=dead_hist, out_ids=oi, out_dist=od, out_counts=oc,
                        node_ids=snap["node_ids"], node_levels=snap["node_levels"], node_offsets=snap["node_offsets"],
                        node_neighbors=snap["node_neighbors"], list_sizes=np.asarray([len(l[1]) for l in lists]))
    print("wrote", sorted(os.path.relpath(os.path.join(dp, f), out) for dp, _, fs in os.walk(out) for f in fs))


if __name__ == "__main__":
    main()
