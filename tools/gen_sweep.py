"""Which synthetic generator lets BOTH parts of the hybrid reach recall@10 >= 0.95?  (dev aid)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
fv = fvdb_import.load()
N, d, B, k, nlist = 200_000, 384, 512, 10, 256
ctx = fv.Context(0)

def gen(L, spread, ncomp, seed, n, ambient=0.02):
    r = np.random.default_rng(1234)
    means = (spread * r.standard_normal((ncomp, L))).astype(np.float32)
    P = np.linalg.qr(r.standard_normal((d, L)))[0].T.astype(np.float32)
    r2 = np.random.default_rng(seed)
    z = means[r2.integers(0, ncomp, n)] + r2.standard_normal((n, L), dtype=np.float32)
    x = z @ P + np.float32(ambient) * r2.standard_normal((n, d), dtype=np.float32)
    return np.ascontiguousarray(x, np.float32)

def recall(found, cnt, exact):
    return np.mean([len(set(found[b, :cnt[b]].tolist()) & set(exact[b].tolist())) / k for b in range(found.shape[0])])

for L, spread, ncomp in [(16, 1.0, 4096), (16, 1.5, 4096), (16, 2.0, 4096), (8, 1.0, 4096), (8, 2.0, 4096), (24, 1.5, 4096), (32, 1.5, 4096)]:
    x = gen(L, spread, ncomp, 1, N); q = gen(L, spread, ncomp, 2, B)
    ids = np.arange(N, dtype=np.uint64)
    t = time.time()
    ivf = fv.DeviceIVF(ctx, d, nlist)
    ivf.train(x[:50000], 15, 7)
    ivf.add(x, ids)
    ex = ivf.search_all(q, k)[0]
    line = f"L={L} spread={spread}: IVF"
    for npb in (4, 8, 16, 32, 64):
        fi, fd, fc = ivf.search(q, k, npb)
        line += f" np{npb}={recall(fi, fc, ex):.3f}"
    ivf.close()
    nh = 60000
    h = fv.HNSWIndex(ctx, 16, 32, 200, seed=11)
    h.bulk_build(ids[:nh], x[:nh])
    flat = fv.DeviceIVF(ctx, d, 1); flat.set_centroids(np.zeros((1, d), np.float32))
    flat.add_assigned(x[:nh], ids[:nh], np.zeros(nh, np.uint32))
    exh = flat.search_all(q, k)[0]; flat.close()
    line += " | HNSW(kNN graph)"
    for ef in (50, 100, 200):
        r = h.search(q, k, ef)
        line += f" ef{ef}={recall(r.ids, r.counts, exh):.3f}"
    print(line, f"({time.time()-t:.1f}s)", flush=True)
    del h
