"""Flat exhaustive scan timing: the scan kernel with full query groups and uniform items."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
fv = fvdb_import.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d, k = 384, 10
rng = np.random.default_rng(0)
x = rng.standard_normal((N, d)).astype(np.float32)
q = rng.standard_normal((B, d)).astype(np.float32)
ctx = fv.Context(0)
ivf = fv.DeviceIVF(ctx, d, 1)
ivf.set_centroids(np.zeros((1, d), np.float32))
ivf.add_assigned(x, np.arange(N, dtype=np.uint64), np.zeros(N, np.uint32))
qd = ctx.upload(q)
ids = ctx.alloc(B * k * 8); ds = ctx.alloc(B * k * 4); cnt = ctx.alloc(B * 4)
for _ in range(2):
    ivf.search_all_dev(qd, B, k, ids, ds, cnt)
ctx.synchronize()
ctx.timer_start()
R = 5
for _ in range(R):
    ivf.search_all_dev(qd, B, k, ids, ds, cnt)
ms = ctx.timer_stop_ms() / R
print(f"flat N={N} B={B}: {ms:.3f} ms  {N*B*d/ms/1e9:.2f} T pair-dims/s  ({3*N*B*d/ms/1e9:.1f} T lane-ops/s)", flush=True)
