"""Randomised sequences of HNSWIndex operations — batch inserts on the device (speculated or one at a time), inserts
by the host algorithm, single inserts, soft deletes, vacuum, searches with the device traversal and with the host walk
— on the GPU index and on the CPU oracle side by side: every search must return the oracle's ids and distance bits,
and at the end the graphs must be equal list by list.  Exercises the hand-overs between the device-resident graph and
its host cache.  python tools/hnsw_ops_fuzz.py [--cases 20] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fvdb_import  # noqa: E402


def same_results(got, want, k):
    valid = np.arange(k)[None, :] < np.asarray(want[2])[:, None]
    return (np.array_equal(got.counts, want[2]) and np.array_equal(got.ids[valid], want[0][valid]) and
            np.array_equal(np.ascontiguousarray(got.distances).view(np.uint32)[valid],
                           np.ascontiguousarray(want[1]).view(np.uint32)[valid]))


def one_case(fv, orc, ctx, rng, case, only=-1, log=print):
    d = int(rng.choice([8, 24, 100, 384]))
    M = int(rng.choice([4, 8, 16]))
    M0 = int(min(63, 2 * M))
    efc = int(rng.choice([16, 48, 100, 200]))
    n_comp = int(rng.choice([1, 8, 256]))
    dup = float(rng.choice([0.0, 0.0, 0.2]))
    steps = int(rng.integers(8, 20))
    seed = int(rng.integers(1, 1 << 30))
    if only != -1 and case != only:
        return 0
    g = np.random.default_rng(seed)
    means = g.standard_normal((n_comp, d)).astype(np.float32)
    total = 6000
    x = means[g.integers(0, n_comp, total)] + np.float32(0.4) * g.standard_normal((total, d)).astype(np.float32)
    nd = int(dup * total)
    if nd:
        x[g.integers(0, total, nd)] = x[g.integers(0, total, nd)]
    x = np.ascontiguousarray(x, np.float32)
    ids = (np.arange(total, dtype=np.uint64) * 7 + 3)
    levels = orc.rng_levels(seed, total)
    gh, oh = fv.HNSWIndex(ctx, M, M0, efc, seed=seed), orc.HNSWIndex(M, M0, efc, seed=seed)
    at, bad, trail = 0, 0, []
    alive = []
    for s in range(steps):
      try:
        op = g.choice(["batch_dev", "batch_dev", "batch_spec", "batch_host", "single", "delete", "search", "search", "vacuum"])
        if os.environ.get("FUZZ_TRACE"):
            print(f"   step {s}: {op} (nodes so far {at})", flush=True)
        if op.startswith("batch") or op == "single":
            cnt = 1 if op == "single" else int(g.integers(2, 700))
            cnt = min(cnt, total - at)
            if cnt <= 0:
                continue
            if op == "batch_host":
                gh.set_device_insert(False)
            else:
                gh.set_device_insert(True, 2 if op == "batch_spec" else int(g.choice([0, 1])))
            sl = slice(at, at + cnt)

            def attempt(f):
                try:
                    return ("ok", f())
                except Exception as e:  # noqa: BLE001
                    return ("err", type(e).__name__)
            if op == "single":
                ra = attempt(lambda: gh.insert(int(ids[at]), x[at], int(levels[at])))
                rb = attempt(lambda: oh.insert(int(ids[at]), x[at], int(levels[at])))
                g_ok, o_ok, g_none = ra[0] == "ok", rb[0] == "ok", ra[0] != "ok"
            else:
                # (the oracle's batch stops at the first row that fails and raises; the product counts failures and goes on)
                ra = attempt(lambda: gh.batch_insert(ids[sl], x[sl], levels[sl]))
                rb = attempt(lambda: oh.batch_insert(ids[sl], x[sl], levels[sl]))
                g_ok, o_ok = ra[0] == "ok" and ra[1][1] == 0, rb[0] == "ok"
                g_none = ra[0] != "ok" or ra[1][0] == 0
            # after a vacuum that removed the entry point every insert fails on both sides (src/hnsw/core.rs:268-274)
            if g_ok != o_ok or (not g_ok and not g_none):
                bad += 1
                op += "!"
            if g_ok:
                alive += list(range(at, at + cnt))
                at += cnt
            else:
                op += "(failed on both)"
        elif op == "delete" and alive:
            for i in g.choice(alive, size=min(len(alive), int(g.integers(1, 30))), replace=False).tolist():
                gh.mark_deleted(int(ids[i]))
                oh.mark_deleted(int(ids[i]))
                alive.remove(i)
        elif op == "vacuum" and at:
            a, b = gh.vacuum(), oh.vacuum()
            bad += 0 if a == b else 1
        elif op == "search" and at:
            gh.set_device_traversal(bool(g.integers(0, 2)))
            q = np.ascontiguousarray(np.concatenate([x[g.integers(0, at, 16)], x[g.integers(0, total, 16)]]), np.float32)
            k, ef = (int(v) for v in g.choice([[5, 5], [10, 50], [10, 64], [20, 120]]))
            # (after a vacuum that removed the entry point the reference's search fails — "Entry point node not found",
            # src/hnsw/core.rs:422-429 — and so must both sides)
            def outcome(f):
                try:
                    return f()
                except Exception as e:  # noqa: BLE001
                    return type(e).__name__
            a, b = outcome(lambda: gh.search(q, k, ef)), outcome(lambda: oh.batch_search(q, k, ef))
            ok = (isinstance(a, str) and isinstance(b, str)) or (not isinstance(a, str) and not isinstance(b, str) and same_results(a, b, k))
            bad += 0 if ok else 1
            op = f"search(dev={gh.device_traversal()},k={k},ef={ef}){'' if ok else '!'}"
        trail.append(op)
      except Exception as e:  # noqa: BLE001
        trail.append(f"{op}:{type(e).__name__}({e})")
        bad += 1
        break
    gbad = 0
    if at:
        gbad += 0 if gh.entry_point() == oh.entry_point() else 1
        gi, lv, off, nb = gh.export_graph()
        slot = 0
        for r, l in zip(gi.tolist(), lv.tolist()):
            for layer in range(l + 1):
                if nb[int(off[slot]):int(off[slot + 1])].tolist() != oh.neighbors(r, layer):
                    gbad += 1
                slot += 1
    st = gh.insert_stats()
    log(f"case {case:3d}: d {d:3d} M {M:2d} ef {efc:3d} comps {n_comp:3d} dup {dup:.1f} nodes {at:5d}: host-path {st['host_path_inserts']:4d} "
        f"adopted {st['speculated_ok']:5d} dev-fallbacks {gh.device_fallbacks():3d}  -> "
        f"{'OK' if bad + gbad == 0 else 'MISMATCH: %d operations, %d lists [%s]' % (bad, gbad, ' '.join(trail))}")
    return bad + gbad


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
    print(f"[ops fuzz] {a.cases} cases, {bad} with mismatches, {time.time() - t0:.0f}s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
