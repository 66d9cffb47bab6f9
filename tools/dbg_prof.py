import sys, os, faulthandler, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np
import fvdb_import
fv = fvdb_import.load()
rng = np.random.default_rng(0)
N, d, nlist, B, k = 30000, 384, 64, 1024, 10
x = rng.standard_normal((N, d)).astype(np.float32)
q = rng.standard_normal((B, d)).astype(np.float32)
ctxA, ctxB = fv.Context(0), fv.Context(0)
hyb = fv.HybridIndex(ctxA, ctx_hnsw=ctxB, n_clusters=nlist, n_probe=8, max_iterations=5)
hyb.initialize(x[:5000])
now = 1000 * 86400.0
ts = np.where(rng.random(N) < 0.3, now - 86400.0, now - 30 * 86400.0)
hyb.bulk_insert(np.arange(N, dtype=np.uint64), x, ts, now)
qd = ctxA.upload(q)
mode = int(sys.argv[1])
print("built", flush=True)
for i in range(2):
    hyb.search_dev(qd, B, k, now=now, hnsw_ef=50, ivf_n_probe=16, dim=d)
print("plain ok", flush=True)
ctxA.set_profiling(mode)
for i in range(3):
    t = time.time()
    hyb.search_dev(qd, B, k, now=now, hnsw_ef=50, ivf_n_probe=16, dim=d)
    print("prof step", i, time.time() - t, flush=True)
print(hyb.ivf_device_stage_times(), flush=True)
