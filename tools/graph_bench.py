"""Traversal kernel alone (dev aid): 300K nodes x 384 bulk-built graph, ef 50, k 10; kernel time from HIP events on the
launch stream for batches of 1024 / 2048 / 4096 queries, plus an identity check against the exact-heap kernel.
Variants are chosen by environment (FVDB_GRAPH_NO_FAST=1, FVDB_GRAPH_FAST_R=8|12|16): run one process per variant."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
from bench import Generator

fv = fvdb_import.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 50
d, k = 384, 10
gen = Generator(d=d)
x = np.concatenate([gen.rows(10000, s) for s in range(n // 10000)])
ctx = fv.Context(0)
h = fv.HNSWIndex(ctx, 16, 32, 200, seed=11)
t = time.time()
h.bulk_build(np.arange(n, dtype=np.uint64), x)
print(f"variant NO_FAST={os.environ.get('FVDB_GRAPH_NO_FAST')} R={os.environ.get('FVDB_GRAPH_FAST_R')}: bulk build {time.time() - t:.1f}s", flush=True)
ctx.set_profiling(2)
for B in (1024, 2048, 4096):
    q = gen.rows(B, 10_000_000 + B)
    qd = ctx.upload(q)
    for _ in range(3):
        r = h.search_dev(qd, B, d, k, ef)
    h.graph_kernel_times()
    R = 10
    t = time.perf_counter()
    for _ in range(R):
        r = h.search_dev(qd, B, d, k, ef)
    wall = (time.perf_counter() - t) / R * 1e3
    ms, launches, rows, hops = h.graph_kernel_times()
    print(f"B={B}: kernel {ms / max(launches, 1):.3f} ms  (wall {wall:.3f} ms)  rows/q {rows / launches / B:.0f} hops/q {hops / launches / B:.1f} "
          f"gathered {rows / launches * d * 4 / (ms / launches * 1e-3) / 1e9:.0f} GB/s  fallbacks {h.device_fallbacks()}", flush=True)
    if B == 1024:
        keep = (r.ids.copy(), r.distances.copy(), r.counts.copy())
if os.environ.get("FVDB_GRAPH_CHECK"):
    h.set_device_traversal(False)
    q = gen.rows(1024, 10_000_000 + 1024)
    w = h.search(q, k, ef)
    same = np.array_equal(w.ids, keep[0]) and np.array_equal(w.distances.view(np.uint32), keep[1].view(np.uint32)) and np.array_equal(w.counts, keep[2])
    print("identical to the host walk:", same, flush=True)
