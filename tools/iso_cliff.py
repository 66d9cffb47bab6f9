"""The list-scan filter on SURVEY §8d's isotropic generator (4096 component means ~ N(0, I_384), sigma 0.35), IVF part
only: per nprobe — ms per batch in AUTO / FILTER / EXACT scan modes, queries handed to the exact rescan and why, survivor
counts.  python tools/iso_cliff.py [--n 700000] [--nprobes 8,16,32,64]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fvdb_import  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=700_000)
    ap.add_argument("--nlist", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--nprobes", default="8,16,32,64")
    ap.add_argument("--check", action="store_true", help="FILTER results == EXACT results bit for bit")
    a = ap.parse_args()
    fv = fvdb_import.load()
    ctx = fv.Context(0)
    d, B, k = 384, a.batch, 10
    gen = bench.IsotropicGenerator(d=d)
    x = np.concatenate([gen.rows(10_000, s) for s in range(a.n // 10_000)])
    ids = np.arange(a.n, dtype=np.uint64)
    qs = [gen.rows(B, 10_000_000 + i) for i in range(4)]
    ivf = fv.IVFIndex(ctx, n_clusters=a.nlist, n_probe=32, train_size=100_000, max_iterations=25, seed=7)
    ivf.train(x[np.random.Generator(np.random.Philox(key=5)).choice(a.n, 100_000, replace=False)])
    ivf.batch_insert(ids, x)
    h = ivf._dev()
    qd = [ctx.upload(q) for q in qs]
    oi, od, oc = ctx.alloc(B * k * 8), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    lib = ctx.lib
    import ctypes as C

    def fallbacks():
        v = C.c_uint64(0)
        ctx.check(lib.fvdb_ivf_scan_fallbacks(h, C.byref(v)))
        r = (C.c_uint64 * 5)()
        ctx.check(lib.fvdb_ivf_scan_fallback_reasons(h, r))
        return v.value, list(r)

    for p in [int(v) for v in a.nprobes.split(",")]:
        line = f"nprobe {p:3d}:"
        for mode, name in ((2, "FILTER"), (1, "EXACT")):
            ctx.check(lib.fvdb_ivf_set_scan_mode(h, mode))
            f0, r0 = fallbacks()
            for _ in range(2):
                ctx.check(lib.fvdb_ivf_search_dev(h, qd[0], B, k, p, oi, od, oc, None))
            ctx.synchronize()
            ctx.timer_start()
            R = 8
            for j in range(R):
                ctx.check(lib.fvdb_ivf_search_dev(h, qd[j % 4], B, k, p, oi, od, oc, None))
            ms = ctx.timer_stop_ms() / R
            f1, r1 = fallbacks()
            line += f"  {name} {ms:.3f} ms/batch"
            if mode == 2:
                surv = np.zeros(B, np.uint32)
                ctx.check(lib.fvdb_ivf_scan_survivors(h, surv.ctypes.data_as(C.POINTER(C.c_uint32)), B))
                line += (f" rescans {(f1 - f0) / (R + 2) / B * 100:.1f}% of queries {[int(b_ - a_) for a_, b_ in zip(r0, r1)]}"
                         f" survivors p50 {int(np.median(surv))} p99 {int(np.percentile(surv, 99))} max {int(surv.max())}")
        print(line, flush=True)
        if a.check:
            res = []
            for mode in (2, 1):
                ctx.check(lib.fvdb_ivf_set_scan_mode(h, mode))
                r = ivf.search(qs[1], k, p)
                res.append((r.ids.copy(), r.distances.copy(), r.counts.copy()))
            same = all(np.array_equal(res[0][i].view(np.uint32) if i == 1 else res[0][i],
                                      res[1][i].view(np.uint32) if i == 1 else res[1][i]) for i in range(3))
            print(f"            FILTER == EXACT bit for bit: {same}", flush=True)
            assert same


if __name__ == "__main__":
    main()
