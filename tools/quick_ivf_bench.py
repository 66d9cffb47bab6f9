"""Quick IVF timing on the GPU box (development aid, not the contract bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import

fv = fvdb_import.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 8
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
d, k = 384, 10
rng = np.random.default_rng(0)
means = rng.standard_normal((1024, d)).astype(np.float32)
x = (means[rng.integers(0, 1024, N)] + 0.35 * rng.standard_normal((N, d))).astype(np.float32)
q = (means[rng.integers(0, 1024, B)] + 0.35 * rng.standard_normal((B, d))).astype(np.float32)
ctx = fv.Context(0)
ivf = fv.DeviceIVF(ctx, d, nlist)
t0 = time.time()
ivf.set_centroids(x[rng.choice(N, nlist, replace=False)])
ivf.reserve(N)
for s in range(0, N, 100_000):
    ivf.add(x[s:s + 100_000], np.arange(s, min(N, s + 100_000), dtype=np.uint64))
print(f"build {time.time()-t0:.2f}s  lists: max {ivf.list_sizes().max()} mean {ivf.list_sizes().mean():.0f}", flush=True)
qd = ctx.upload(q)
ids = ctx.alloc(B * k * 8); ds = ctx.alloc(B * k * 4); cnt = ctx.alloc(B * 4)
for _ in range(3):
    ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
ctx.synchronize()
ctx.timer_start()
R = 10
for _ in range(R):
    ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
ms = ctx.timer_stop_ms() / R
st = ivf.last_stats()
rows = st["rows_scanned"]
print(f"N={N} nlist={nlist} nprobe={nprobe} B={B}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} QPS  "
      f"rows/q={rows/B:.0f} items={st['work_items']} alg {rows*d*4/ms/1e6:.1f} GB/s  "
      f"pair-dims/s {rows*d/ms/1e9:.2f} T/s", flush=True)
ctx.set_profiling(True)
for _ in range(5):
    ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
n, st = ivf.stage_times()
print({k_: round(v / n, 4) for k_, v in st.items()}, flush=True)
