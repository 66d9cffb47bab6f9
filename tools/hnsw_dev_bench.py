"""Device-resident HNSW traversal timing (dev aid): 300K nodes x 384, B=1024, ef=50."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fvdb_import
from bench import Generator
fv = fvdb_import.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
B, d, k, ef = (int(sys.argv[3]) if len(sys.argv) > 3 else 1024), 384, 10, int(sys.argv[2]) if len(sys.argv) > 2 else 50
gen = Generator(d=d)
x = np.concatenate([gen.rows(10000, s) for s in range(n // 10000)])
q = gen.rows(B, 10_000_000)
ctx = fv.Context(0)
h = fv.HNSWIndex(ctx, 16, 32, 200, seed=11)
t = time.time(); h.bulk_build(np.arange(n, dtype=np.uint64), x); print(f"bulk build {time.time()-t:.1f}s", flush=True)
qd = ctx.upload(q)
for mode in ((True, False) if len(sys.argv) <= 3 else (True,)):
    h.set_device_traversal(mode)
    for _ in range(2): h.search_dev(qd, B, d, k, ef)
    t = time.perf_counter(); R = 5
    for _ in range(R): r = h.search_dev(qd, B, d, k, ef)
    ms = (time.perf_counter() - t) / R * 1e3
    print(f"device_traversal={mode}: {ms:.3f} ms per batch of {B}  ({B/ms*1e3:.0f} QPS)", flush=True)
