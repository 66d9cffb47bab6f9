"""Config C5 (BASELINE.json configs[4]): 1M x 768 rows stored as fp16, IVF-flat, batch 1024 — the fp16 MFMA distance
path at full size.  Timing of the IVF chain (matrix-core path vs exact path) + recall@10 against the exact flat scan
of the same fp16 rows + a bit-exactness check between the two paths.   usage: c5_bench.py [N] [nprobe]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
import bench

fv = fvdb_import.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nprobe = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d, nlist, B, k = 768, 1024, 1024, 10
gen = bench.Generator(d=d)
x = np.empty((N, d), np.float32)
for c in range(0, N, 10_000):
    x[c:c + 10_000] = gen.rows(min(10_000, N - c), stream=c // 10_000)
q = gen.rows(B, 10_000_000)
ctx = fv.Context(0)
ivf = fv.DeviceIVF(ctx, d, nlist, dtype="f16")
t0 = time.time()
ivf.train(x[:100_000], seed=7, max_iterations=25)
ivf.reserve(N)
for s in range(0, N, 100_000):
    ivf.add(x[s:s + 100_000], np.arange(s, min(N, s + 100_000), dtype=np.uint64))
print(f"build (k-means on 100K + {N} rows as fp16): {time.time()-t0:.1f}s", flush=True)
exact_ids = ivf.search_all(q, k)[0]
qd = ctx.upload(q)
ids = ctx.alloc(B * k * 8); ds = ctx.alloc(B * k * 4); cnt = ctx.alloc(B * 4)
res = {}
for mode in (0, 1):
    ivf.set_scan_mode(mode)
    ivf.set_coarse_mode(mode)
    for _ in range(3):
        ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
    ctx.synchronize()
    ctx.timer_start()
    R = 20
    for _ in range(R):
        ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
    ms = ctx.timer_stop_ms() / R
    got = ctx.download(ids, (B, k), np.uint64)
    gd = ctx.download(ds, (B, k), np.float32)
    res[mode] = (got, gd)
    rec = np.mean([len(set(got[b].tolist()) & set(exact_ids[b].tolist())) / k for b in range(B)])
    st = ivf.last_stats()
    print(f"C5 N={N} d={d} fp16 nlist={nlist} nprobe={nprobe} B={B}: {'matrix-core' if mode == 0 else 'exact'} path "
          f"{ms:.3f} ms/batch  {B/ms*1e3:.0f} QPS  recall@10={rec:.4f}  rows/q={st['rows_scanned']/B:.0f}"
          + (f"  rescans={ivf.scan_fallbacks()}" if mode == 0 else ""), flush=True)
same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
print("matrix-core path == exact path (ids and distance bits):", bool(same))
