"""Sequential HNSW build (HNSWIndex::insert, src/hnsw/core.rs:226-378): the device-resident insert against the CPU
oracle.  python tools/build_bench.py --n 10000 --d 384 [--mode 0|1|2] [--check] [--gen mixture|refbench|latent]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fvdb_import  # noqa: E402


def gen(name, n, d, seed):
    from _data import mixture
    if name == "mixture":
        return mixture(n, d, n_comp=4096, sigma=0.35, seed=seed)
    if name == "refbench":
        i = (np.arange(n, dtype=np.int64) + seed).astype(np.float32)
        base = np.fmod(i * np.float32(0.001), np.float32(1.0)).astype(np.float32)
        ramp = (np.arange(d, dtype=np.float32) * np.float32(0.0001)).astype(np.float32)
        return (base[:, None] + ramp[None, :]).astype(np.float32)
    if name == "latent":  # bench.py's generator (C3 headline data)
        import bench
        g = bench.Generator(d=d)
        out = np.empty((n, d), np.float32)
        for c in range(0, n, 10000):
            out[c:c + 10000] = g.rows(min(10000, n - c), stream=c // 10000 + (0 if seed == 1234 else 777))
        return out
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, d)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--d", type=int, default=384)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--m0", type=int, default=32)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--gen", default="mixture")
    ap.add_argument("--check", action="store_true", help="build the same graph with the CPU oracle and compare")
    ap.add_argument("--host", action="store_true", help="host algorithm with per-hop GPU scoring instead")
    ap.add_argument("--chunk", type=int, default=0, help="insert in chunks of this many (0 = one batch_insert call)")
    ap.add_argument("--tail", type=int, default=0, help="after the build: this many more inserts, timed on their own")
    ap.add_argument("--tail-mode", type=int, default=1)
    a = ap.parse_args()
    fv = fvdb_import.load()
    import oracle as orc
    orc.build()
    x = gen(a.gen, a.n + a.tail, a.d, 1234)
    xt, x = x[a.n:], x[:a.n]
    ids = np.arange(a.n, dtype=np.uint64)
    levels = orc.rng_levels(42, a.n)
    ctx = fv.Context(0)
    gh = fv.HNSWIndex(ctx, a.m, a.m0, a.efc, seed=42)
    gh.set_device_insert(not a.host, a.mode)
    t0 = time.time()
    if a.chunk:
        for o in range(0, a.n, a.chunk):
            gh.batch_insert(ids[o:o + a.chunk], x[o:o + a.chunk], levels[o:o + a.chunk])
            print(f"  {o + a.chunk} nodes {time.time() - t0:.1f}s", flush=True)
    else:
        gh.batch_insert(ids, x, levels)
    t1 = time.time()
    st = gh.insert_stats()
    print(f"[build] n {a.n} d {a.d} gen {a.gen} mode {a.mode} host {a.host}: {t1 - t0:.2f}s = {a.n / (t1 - t0):.0f} inserts/s "
          f"({(t1 - t0) / a.n * 1e3:.3f} ms each)  stats {st}", flush=True)
    if a.tail:
        gh.set_device_insert(not a.host, a.tail_mode)
        b = gh.insert_stats()
        t0 = time.time()
        gh.batch_insert(np.arange(a.n, a.n + a.tail, dtype=np.uint64), xt, orc.rng_levels(43, a.tail))
        t1 = time.time()
        e = gh.insert_stats()
        print(f"[tail] {a.tail} inserts at {a.n} nodes, mode {a.tail_mode}: {(t1 - t0) / a.tail * 1e3:.3f} ms each; "
              f"{ {k: e[k] - b[k] for k in e} }", flush=True)
    if a.check:
        t0 = time.time()
        oh = orc.HNSWIndex(a.m, a.m0, a.efc, seed=42)
        oh.batch_insert(ids, x, levels)
        t1 = time.time()
        print(f"[oracle] {t1 - t0:.2f}s = {a.n / (t1 - t0):.0f} inserts/s", flush=True)
        assert gh.entry_point() == oh.entry_point(), (gh.entry_point(), oh.entry_point())
        gi, lv, off, nb = gh.export_graph()
        slot, bad = 0, 0
        for r, l in zip(gi.tolist(), lv.tolist()):
            assert l == oh.level(r)
            for layer in range(l + 1):
                if nb[int(off[slot]):int(off[slot + 1])].tolist() != oh.neighbors(r, layer):
                    if bad < 5:
                        print("MISMATCH node", r, "layer", layer, nb[int(off[slot]):int(off[slot + 1])].tolist(), oh.neighbors(r, layer))
                    bad += 1
                slot += 1
        print(f"[check] lists differing from the oracle: {bad}", flush=True)
        assert bad == 0


if __name__ == "__main__":
    main()
