"""IVF-only timing on bench.py's data (latent mixture, k-means centroids) — development aid.
usage: ivf_bench2.py [N] [nlist] [nprobe] [B]   (env FVDB_SCAN_EXACT / FVDB_COARSE_EXACT select the exact stages)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
import bench

fv = fvdb_import.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 700_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 32
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
d, k = 384, 10
gen = bench.Generator(d=d)
x = gen.rows(N, 0)
q = gen.rows(B, 1)
ctx = fv.Context(0)
ivf = fv.DeviceIVF(ctx, d, nlist)
t0 = time.time()
ivf.train(x[:100_000], seed=7, max_iterations=25)
print(f"train {time.time()-t0:.2f}s", flush=True)
t0 = time.time()
ivf.reserve(N)
for s in range(0, N, 100_000):
    ivf.add(x[s:s + 100_000], np.arange(s, min(N, s + 100_000), dtype=np.uint64))
ls = ivf.list_sizes()
print(f"add {time.time()-t0:.2f}s  lists: min {ls.min()} max {ls.max()} mean {ls.mean():.0f}", flush=True)
qd = ctx.upload(q)
ids = ctx.alloc(B * k * 8); ds = ctx.alloc(B * k * 4); cnt = ctx.alloc(B * 4)
for _ in range(3):
    ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
ctx.synchronize()
ctx.timer_start()
R = 20
for _ in range(R):
    ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
ms = ctx.timer_stop_ms() / R
st = ivf.last_stats()
rows = st["rows_scanned"]
print(f"N={N} nlist={nlist} nprobe={nprobe} B={B}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} QPS  rows/q={rows/B:.0f} "
      f"items={st['work_items']} pair-dims/s {rows*d/ms/1e9:.2f} T/s", flush=True)
ctx.set_profiling(True)
for _ in range(5):
    ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
n, stg = ivf.stage_times()
print({k_: round(v / n, 4) for k_, v in stg.items()}, "scan fallbacks", ivf.scan_fallbacks(), "coarse fallbacks", ivf.coarse_fallbacks(), flush=True)
try:
    sv = ivf.scan_survivors(B)
    print("survivors/query: min %d median %d p90 %d p99 %d max %d  (>512: %d)" % (
        sv.min(), np.median(sv), np.percentile(sv, 90), np.percentile(sv, 99), sv.max(), int((sv > 512).sum())), flush=True)
except Exception as e:
    print("no survivor stats:", e)
