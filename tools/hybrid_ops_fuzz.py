"""Randomised sequences of HybridIndex operations — insert_with_timestamp at random ages (recent -> graph, old -> lists),
time moving on (per-search auto-migration), searches with host-resident and device-resident queries (blocking and
begin/end), deletes, explicit migration — on the GPU index and on the CPU oracle side by side: every search must return
the oracle's ids, distance bits and counts, and the recent / historical counters must agree throughout.
python tools/hybrid_ops_fuzz.py [--cases 20] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fvdb_import  # noqa: E402

DAY = 86400.0


def same_results(got, want, k):
    valid = np.arange(k)[None, :] < np.asarray(want[2])[:, None]
    return (np.array_equal(got.counts, want[2]) and np.array_equal(got.ids[valid], want[0][valid]) and
            np.array_equal(np.ascontiguousarray(got.distances).view(np.uint32)[valid],
                           np.ascontiguousarray(want[1]).view(np.uint32)[valid]))


def one_case(fv, orc, ctx, rng, case, only=-1, log=print):
    d = int(rng.choice([8, 32, 128]))
    nlist = int(rng.choice([4, 8, 24]))
    n_comp = int(rng.choice([2, 8, 64]))
    dup = float(rng.choice([0.0, 0.0, 0.15]))
    steps = int(rng.integers(10, 24))
    seed = int(rng.integers(1, 1 << 30))
    if only != -1 and case != only:
        return 0
    g = np.random.default_rng(seed)
    total = 4000
    means = g.standard_normal((n_comp, d)).astype(np.float32)
    x = means[g.integers(0, n_comp, total)] + np.float32(0.5) * g.standard_normal((total, d)).astype(np.float32)
    nd = int(dup * total)
    if nd:
        x[g.integers(0, total, nd)] = x[g.integers(0, total, nd)]
    x = np.ascontiguousarray(x, np.float32)
    levels = orc.rng_levels(seed, total)
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=int(g.choice([24, 60])), n_clusters=nlist,
              n_probe=int(g.integers(1, nlist + 1)))
    gi, oi = fv.HybridIndex(ctx, **kw), orc.HybridIndex(**kw)
    cents = np.ascontiguousarray(x[g.choice(total, nlist, replace=False)])
    gi.set_ivf_centroids(cents)
    oi.set_ivf_centroids(cents)
    now, at, bad, trail, alive = 1000 * DAY, 0, 0, [], []
    for s in range(steps):
        op = g.choice(["insert", "insert", "insert", "tick", "search", "search", "search_dev", "begin_end", "delete", "migrate"])
        if op == "insert":
            cnt = min(int(g.integers(1, 250)), total - at)
            for i in range(at, at + cnt):
                age = float(g.choice([0.0, 1 * DAY, 6.9 * DAY, 7.1 * DAY, 30 * DAY]))
                gi.insert_with_timestamp(i, x[i], now - age, now, int(levels[i]))
                oi.insert_with_timestamp(i, x[i], now - age, now, int(levels[i]))
            alive += list(range(at, at + cnt))
            at += cnt
        elif op == "tick":
            now += float(g.choice([0.5 * DAY, 3 * DAY, 8 * DAY]))
        elif op in ("search", "search_dev", "begin_end") and at:
            q = np.ascontiguousarray(np.concatenate([x[g.integers(0, at, 12)], x[g.integers(0, total, 12)]]), np.float32)
            k, ef, npb = int(g.choice([3, 10, 20])), int(g.choice([10, 50, 80])), int(g.integers(1, nlist + 1))
            want = oi.batch_search(q, k, now=now, hnsw_ef=ef, ivf_n_probe=npb)
            if op == "search":
                got = gi.search(q, k, now=now, hnsw_ef=ef, ivf_n_probe=npb)
            elif op == "search_dev":
                got = gi.search_dev(ctx.upload(q), q.shape[0], k, now=now, hnsw_ef=ef, ivf_n_probe=npb, dim=d)
            else:
                slot = int(g.integers(0, 8))
                gi.search_dev_begin(slot, ctx.upload(q), q.shape[0], k, now=now, hnsw_ef=ef, ivf_n_probe=npb, dim=d)
                got = gi.search_dev_end(slot)
            ok = same_results(got, want, k)
            bad += 0 if ok else 1
            op += "" if ok else "!"
        elif op == "delete" and alive:
            for i in g.choice(alive, size=min(len(alive), int(g.integers(1, 20))), replace=False).tolist():
                # (a row that went to the graph and has aged past the threshold without being migrated is looked for in
                # the lists and not found, src/hybrid/core.rs delete: both sides must fail alike)
                def outcome(f):
                    try:
                        f()
                        return "ok"
                    except Exception as e:  # noqa: BLE001
                        return type(e).__name__
                a, b = outcome(lambda: gi.delete(i, now)), outcome(lambda: oi.delete(i, now))
                if (a == "ok") != (b == "ok"):
                    bad += 1
                    op += "!"
                if a == "ok":
                    alive.remove(i)
        elif op == "migrate":
            th = float(g.choice([2 * DAY, 7 * DAY]))
            a, b = gi.migrate_with_threshold(th, now), oi.migrate_with_threshold(th, now)
            ok = a == b
            bad += 0 if ok else 1
            op += "" if ok else "!"
        if gi.recent_count() != oi.recent_count() or gi.historical_count() != oi.historical_count():
            bad += 1
            op += "#"
        trail.append(op)
    log(f"case {case:3d}: d {d:3d} lists {nlist:2d} comps {n_comp:2d} dup {dup:.2f} rows {at:4d} recent {gi.recent_count():4d} historical "
        f"{gi.historical_count():4d}  -> {'OK' if bad == 0 else 'MISMATCH in %d operations [%s]' % (bad, ' '.join(trail))}")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1)
    a = ap.parse_args()
    fv = fvdb_import.load()
    import oracle as orc
    orc.build()
    ctx = fv.Context(0)
    rng = np.random.default_rng(a.seed)
    t0, bad = time.time(), 0
    for c in range(a.cases):
        bad += 1 if one_case(fv, orc, ctx, rng, c, a.only) else 0
    print(f"[hybrid ops fuzz] {a.cases} cases, {bad} with mismatches, {time.time() - t0:.0f}s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
