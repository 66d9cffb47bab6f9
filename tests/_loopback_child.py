"""Child process of tests/test_00_gpu_sharded_ranks.py: the DEV build's loopback communicator (include/fvdb_dev.h) runs rank 1
of a pretended 4-rank job on the one GPU.  Its results are not search results (the peers' blocks are copies of its own);
what must hold is that the step runs with the real placement and shapes, and that bad shapes are refused."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np

import fvdb_import
from _data import mixture

DAY = 86400.0
fv = fvdb_import.load()
sh = fv.sharded
ctx = fv.Context(0)
assert hasattr(ctx.lib, "fvdb_comm_create_loopback"), "not the dev build"
for bad in ((4, 4), (0, 0)):
    try:
        sh.Comm.loopback(ctx, *bad)
    except fv.FvdbError:
        pass
    else:
        raise AssertionError(f"communicator shape {bad} accepted")
n, d, nlist, k, nprobe, ef, B = 12000, 64, 32, 10, 8, 40, 70
x = mixture(n, d, n_comp=16, sigma=1.0, seed=190)
ids = np.arange(n, dtype=np.uint64)
now = 1000 * DAY
ts = np.where(np.random.default_rng(190).random(n) < 0.25, now - DAY, now - 30 * DAY)
hyb = fv.HybridIndex(ctx, n_clusters=nlist, n_probe=nprobe, hnsw_seed=29)
hyb.set_ivf_centroids(x[:nlist].copy())
comm = sh.Comm.loopback(ctx, 4, 1)
assert (comm.world, comm.rank) == (4, 1)
S = sh.ShardedHybrid(hyb, comm)
S.bulk_insert(ids, x, ts, now)
owned = int((S.owner == 1).sum())
assert 0 < owned < nlist  # this rank holds its share of the lists only
q = ctx.upload(mixture(B, d, n_comp=16, sigma=1.0, seed=191))
r = S.search_dev(q, B, k, ef, nprobe, sh.WEAK)
assert r.ids.shape == (B, k) and r.counts.shape == (B,) and np.all(r.counts <= k)
r = S.search_dev(q, B, k, ef, nprobe, sh.STRONG)
per = -(-B // 4)
assert S.rows(B, sh.STRONG) == per and r.ids.shape[0] == per
comm.close()
print("loopback child ok")
