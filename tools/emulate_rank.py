"""Per-rank cost of a multi-GPU step, measured on ONE GPU (capacity planning; fvdb_comm_create_loopback).

Rank 0 of a pretended W-rank job: it owns 1/W of the inverted lists (the product's placement), walks the replicated
graph for its own queries and scans its lists for every query of the global batch; the two exchanges and the
threshold exchange are device copies of its own blocks, so buffer sizes, kernels and stream order are the real ones
and only the fabric is missing.  Results are NOT search results.  Prints ms per step with 8 steps in flight for
W = 1 (the plain single-GPU step), then weak and strong mode at each W given.

Needs the DEVELOPMENT build of the engine (the loopback communicator is not in the product library):
    make -C fabstir-vectordb_amd dev && FVDB_LIB_DIR=lib_dev python tools/emulate_rank.py [W ...]   (default 2 4 8)"""
import os
import sys
import time

os.environ.setdefault("FVDB_LIB_DIR", "lib_dev")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
from bench import DAY, Generator

fv = fvdb_import.load()
sh = fv.sharded
Ws = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
N, d, B, k, nlist, nprobe, ef, depth, steps = 1_000_000, 384, 1024, 10, 1024, 32, 50, 8, 100
gen = Generator(d=d)
x = np.concatenate([gen.rows(10_000, stream=c) for c in range(N // 10_000)])
ids = np.arange(N, dtype=np.uint64)
now = 1000 * DAY
is_recent = np.random.Generator(np.random.Philox(key=99)).random(N) < 0.3
ts = np.where(is_recent, now - 1 * DAY, now - 30 * DAY)
queries = [gen.rows(B, stream=10_000_000 + i) for i in range(8)]
sample = x[np.random.Generator(np.random.Philox(key=5)).choice(N, 100_000, replace=False)]


def build(world):
    ctx_i, ctx_h = fv.Context(0), fv.Context(0)
    hyb = fv.HybridIndex(ctx_i, ctx_hnsw=ctx_h, n_clusters=nlist, n_probe=32, train_size=100_000, max_iterations=25,
                         ivf_seed=7, hnsw_seed=11)
    hyb.initialize(sample)
    comm = sh.Comm.loopback(ctx_i, world, 0)
    S = sh.ShardedHybrid(hyb, comm)
    S.bulk_insert(ids, x, ts, now)
    return ctx_i, hyb, S


def timed(ctx, S, qdev, mode):
    def run(n):
        for i in range(n):
            S.search_dev_begin(i % depth, qdev[i % len(qdev)], B, k, ef, nprobe, mode)
            if i >= depth - 1:
                S.search_dev_end((i - depth + 1) % depth)
        for i in range(max(n - depth + 1, 0), n):
            S.search_dev_end(i % depth)
    run(2 * depth)
    ctx.device_synchronize()
    t = time.perf_counter()
    run(steps)
    ctx.device_synchronize()
    return (time.perf_counter() - t) / steps * 1e3


only_weak = os.environ.get("EMU_ONLY_WEAK") == "1"  # profiling aid: one mode, no W = 1 pass
for W in ([] if only_weak else [1]) + Ws:
    ctx, hyb, S = build(W)
    qdev = [ctx.upload(q) for q in queries]
    owned = int((S.owner == 0).sum())
    w = timed(ctx, S, qdev, sh.WEAK)
    line = f"W={W}: rank 0 owns {owned} of {nlist} lists | weak: {w:.3f} ms per rank-step ({B} own queries, {W * B} scanned)"
    if W > 1 and not only_weak:
        s = timed(ctx, S, qdev, sh.STRONG)
        line += f" | strong: {s:.3f} ms per step of {B} ({-(-B // W)} own)"
    print(line, flush=True)
    del S, hyb, ctx
