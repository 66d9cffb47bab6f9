import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fvdb_import
from _data import mixture
fv = fvdb_import.load()
ctx = fv.Context(0)
d, nlist = 32, 64
x = mixture(nlist * 3, d, seed=701)
ids = np.arange(x.shape[0], dtype=np.uint64)
cents = x[:nlist].copy()
gpu = fv.DeviceIVF(ctx, d, nlist)
gpu.set_centroids(cents)
cl, pos = gpu.add(x, ids)
q = mixture(50, d, seed=702)
for npb in (5, 4, 2, 64):
    gpu.set_scan_mode(0)
    a = gpu.search(q, 10, npb)
    gpu.set_scan_mode(1)
    e = gpu.search(q, 10, npb)
    bad = [i for i in range(q.shape[0]) if not np.array_equal(a[0][i], e[0][i])]
    print("nprobe", npb, "bad queries", bad, "fallbacks", gpu.scan_fallbacks())
    for i in bad[:3]:
        print(" auto ", a[0][i], a[1][i])
        print(" exact", e[0][i], e[1][i])
        pr, _ = gpu.coarse(q[i:i+1], npb)
        print(" probes", pr[0], "list sizes", gpu.list_sizes()[pr[0]], " cl/pos of dup", [(int(cl[j]), int(pos[j])) for j in a[0][i][:10] if j < len(cl)])
