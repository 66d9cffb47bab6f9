"""Save and re-open an index in the reference's chunked format; time the stages.  python tools/chunked_bench.py [N] [d]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fvdb_import  # noqa: E402
from _data import bits, mixture  # noqa: E402

fv = fvdb_import.load()
ck = fv.chunked
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 384
DAY, now, nlist = 86400.0, 1000 * 86400.0, 256
ctx = fv.Context(0)
x = mixture(N, d, n_comp=256, seed=3)
kw = dict(n_clusters=nlist, n_probe=16, max_connections=16, max_connections_layer_0=32, ef_construction=100)
g = fv.HybridIndex(ctx, **kw)
g.set_ivf_centroids(x[np.random.default_rng(1).choice(N, nlist, replace=False)].copy())
ids = np.arange(N, dtype=np.uint64) * 7919 + 11
ts = np.where(np.arange(N) % 20 == 0, now - DAY, now - 30 * DAY)  # 5 % recent
t0 = time.perf_counter()
g.bulk_insert(ids, x, ts, now)
print(f"built: {g.recent_count()} graph nodes + {g.historical_count()} list rows in {time.perf_counter() - t0:.1f} s", flush=True)
with tempfile.TemporaryDirectory(dir="/tmp") as root:
    t0 = time.perf_counter()
    m = ck.save_index_chunked(g, root, "idx", now=now)
    t_save = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(root) for f in fs)
    print(f"save_index_chunked: {t_save:.1f} s, {len(m['chunks'])} chunks, {size / 1e6:.1f} MB on disk "
          f"({size / N:.0f} B/vector; raw f32 rows {4 * d} B)", flush=True)
    t0 = time.perf_counter()
    raw = [open(os.path.join(root, "idx", "chunks", f"{c['chunk_id']}.cbor"), "rb").read() for c in m["chunks"]]
    t_read = time.perf_counter() - t0
    t0 = time.perf_counter()
    rows = 0
    for data in raw:
        rows += len(ck.read_chunk(data)[3])
    t_dec = time.perf_counter() - t0
    print(f"chunk files read in {t_read:.2f} s; CBOR decode of {rows} rows {t_dec:.1f} s ({sum(map(len, raw)) / t_dec / 1e6:.0f} MB/s)",
          flush=True)
    t0 = time.perf_counter()
    h, table = ck.load_index_chunked(ctx, root, "idx", now=now, **kw)
    t_load = time.perf_counter() - t0
    st = h.ivf().stage_times() if hasattr(h.ivf(), "stage_times") else None
    print(f"load_index_chunked: {t_load:.1f} s total ({h.recent_count()} + {h.historical_count()} vectors)", flush=True)
q = mixture(256, d, n_comp=256, seed=9)
a = g.search(q, 10, now=now, hnsw_ef=50, ivf_n_probe=16)
b = h.search(q, 10, now=now, hnsw_ef=50, ivf_n_probe=16)
same = np.array_equal(a.ids, b.ids) and np.array_equal(bits(a.distances), bits(b.distances))
print(f"256 queries, k=10: loaded index answers identically to the saved one: {same}")
# the reference's reconstruction scores every chunk vector against every centroid once per populated cluster
# (src/hybrid/persistence.rs:614-637): nlist x N x nlist scalar distances
print(f"reference load loop at this size: {nlist} clusters x {rows} vectors x {nlist} centroids = "
      f"{nlist * rows * nlist / 1e9:.1f} G scalar distances of d={d}; here one GPU assignment pass of {rows} x {nlist}")
