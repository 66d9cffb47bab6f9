"""Worst case for the matrix-core filter: iid Gaussian rows (no cluster structure) — how many queries fall back to
the exact rescan, and what the IVF chain costs then.  usage: iid_worst_case.py [N] [nlist] [nprobe]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
fv = fvdb_import.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 32
B, d, k = 1024, 384, 10
rng = np.random.default_rng(0)
x = rng.standard_normal((N, d), dtype=np.float32)
q = rng.standard_normal((B, d), dtype=np.float32)
ctx = fv.Context(0)
cents = x[rng.choice(N, nlist, replace=False)].copy()
for mode in (0, 1):
    ivf = fv.DeviceIVF(ctx, d, nlist)
    ivf.set_centroids(cents)
    ivf.set_scan_mode(mode)
    ivf.reserve(N)
    for s in range(0, N, 100_000):
        ivf.add(x[s:s + 100_000], np.arange(s, min(N, s + 100_000), dtype=np.uint64))
    qd = ctx.upload(q)
    ids = ctx.alloc(B * k * 8); ds = ctx.alloc(B * k * 4); cnt = ctx.alloc(B * 4)
    for _ in range(3):
        ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
    ctx.synchronize()
    f0 = ivf.scan_fallbacks()
    ctx.timer_start()
    R = 40
    for _ in range(R):
        ivf.search_dev(qd, B, k, nprobe, ids, ds, cnt)
    ms = ctx.timer_stop_ms() / R
    fb = (ivf.scan_fallbacks() - f0) / R
    extra = ""
    if mode == 0:
        sv = ivf.scan_survivors(B)
        extra = f"  survivors/query median {int(np.median(sv))} p99 {int(np.percentile(sv, 99))}  fallbacks/batch {fb:.0f}"
    print(f"iid N={N} nlist={nlist} nprobe={nprobe}: scan_mode={'auto' if mode == 0 else 'exact'} {ms:.3f} ms/batch{extra}", flush=True)
    res = ctx.download(ids, (B, k), np.uint64)
    if mode == 0:
        first = res
    else:
        print("identical ids in both modes:", bool(np.array_equal(first, res)))
    ivf.close()
