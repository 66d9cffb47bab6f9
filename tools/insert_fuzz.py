"""Randomised parity sweep of the device-resident insert: many small graphs of varied shape — dimension, degree,
ef_construction, cluster structure, exact duplicates, soft deletes between batches, forced speculation with long batches —
each compared list by list with the CPU oracle.  python tools/insert_fuzz.py [--cases 40] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fvdb_import  # noqa: E402


def one_case(fv, orc, ctx, rng, case, only=-1):
    n = int(rng.integers(600, 5000))
    d = int(rng.choice([3, 8, 16, 24, 48, 100, 128, 384]))
    M = int(rng.choice([4, 6, 8, 12, 16]))
    M0 = int(min(63, M * int(rng.choice([1, 2, 3]))))
    efc = int(rng.choice([8, 20, 40, 64, 100, 200, 256, 300]))
    n_comp = int(rng.choice([1, 2, 8, 64, 4096]))
    sigma = float(rng.choice([0.05, 0.35, 1.0]))
    dup_frac = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
    quant = bool(rng.integers(0, 4) == 0)  # coarse grid: many equal distances between different vectors
    mode = int(rng.choice([0, 2, 2]))
    seed = int(rng.integers(1, 1 << 30))
    if only != -1 and case != only:  # (the generator has been advanced exactly as if the case had run)
        return 0
    g = np.random.default_rng(seed)
    means = g.standard_normal((n_comp, d)).astype(np.float32)
    x = means[g.integers(0, n_comp, n)] + np.float32(sigma) * g.standard_normal((n, d)).astype(np.float32)
    if quant:
        x = np.round(x * 4) / 4
    ndup = int(dup_frac * n)
    if ndup:
        x[g.integers(0, n, ndup)] = x[g.integers(0, n, ndup)]
    x = np.ascontiguousarray(x, np.float32)
    ids = np.arange(n, dtype=np.uint64) + 11
    levels = orc.rng_levels(seed, n)
    gh, oh = fv.HNSWIndex(ctx, M, M0, efc, seed=seed), orc.HNSWIndex(M, M0, efc, seed=seed)
    gh.set_device_insert(True, mode)
    cuts = sorted(set([0, n] + [int(c) for c in g.integers(1, n, int(g.integers(0, 4)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        gh.batch_insert(ids[a:b], x[a:b], levels[a:b])
        oh.batch_insert(ids[a:b], x[a:b], levels[a:b])
        if g.integers(0, 2) and b < n:  # soft deletes between two batches
            for i in g.integers(0, b, max(1, b // 50)).tolist():
                gh.mark_deleted(int(ids[i]))
                oh.mark_deleted(int(ids[i]))
    st = gh.insert_stats()
    bad = 0
    if gh.entry_point() != oh.entry_point():
        bad += 1
    gi, lv, off, nb = gh.export_graph()
    slot = 0
    for r, l in zip(gi.tolist(), lv.tolist()):
        for layer in range(l + 1):
            if nb[int(off[slot]):int(off[slot + 1])].tolist() != oh.neighbors(r, layer):
                bad += 1
            slot += 1
    # searches on the finished graph, device traversal against the oracle's walk: rows of the index itself among the
    # queries (distance 0 and, with duplicates, equal distances), ef below / at / above the sorted-register form
    nq = 96
    q = np.ascontiguousarray(np.concatenate([x[g.integers(0, n, nq // 2)],
                                             means[g.integers(0, n_comp, nq // 2)] + np.float32(sigma) * g.standard_normal((nq // 2, d)).astype(np.float32)]), np.float32)
    if quant:
        q = np.ascontiguousarray(np.round(q * 4) / 4, np.float32)
    sbad = 0
    for k, ef in ((10, 10), (10, 50), (5, 64), (20, 100)):
        got, want = gh.search(q, k, ef), oh.batch_search(q, k, ef)
        valid = np.arange(k)[None, :] < np.asarray(want[2])[:, None]  # entries past a row's count are padding
        same = (np.array_equal(got.counts, want[2]) and np.array_equal(got.ids[valid], want[0][valid]) and
                np.array_equal(np.ascontiguousarray(got.distances).view(np.uint32)[valid], np.ascontiguousarray(want[1]).view(np.uint32)[valid]))
        sbad += 0 if same else 1
        if not same and os.environ.get("FUZZ_VERBOSE"):
            rows = [b for b in range(nq) if got.counts[b] != want[2][b] or not np.array_equal(got.ids[b][valid[b]], want[0][b][valid[b]]) or
                    not np.array_equal(got.distances[b].view(np.uint32)[valid[b]], want[1][b].view(np.uint32)[valid[b]])]
            b = rows[0]
            print(f"   search k {k} ef {ef}: {len(rows)} of {nq} queries differ; query {b} (index row: {b < nq // 2}): counts {got.counts[b]} / {want[2][b]}\n"
                  f"     got  {got.ids[b].tolist()}\n     want {want[0][b].tolist()}\n     got d  {got.distances[b].tolist()}\n     want d {want[1][b].tolist()}")
    bad += sbad
    print(f"case {case:3d}: n {n:5d} d {d:3d} M {M:2d}/{M0:2d} ef {efc:3d} comps {n_comp:4d} sigma {sigma:.2f} dup {dup_frac:.2f} grid {int(quant)} "
          f"mode {mode} batches {len(cuts) - 1}: adopted {st['speculated_ok']:5d} stops {st['commit_stops']:5d} restarts {st['tie_restarts']:4d} "
          f"host {st['host_path_inserts']:3d} dev-fallbacks {gh.device_fallbacks():3d}  -> "
          f"{'OK' if bad == 0 else 'MISMATCH in %d lists / search settings (%d of them searches)' % (bad, sbad)}", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="run this case of the sequence only")
    a = ap.parse_args()
    fv = fvdb_import.load()
    import oracle as orc
    orc.build()
    ctx = fv.Context(0)
    rng = np.random.default_rng(a.seed)
    t0, bad = time.time(), 0
    for c in range(a.cases):
        bad += 1 if one_case(fv, orc, ctx, rng, c, a.only) else 0
    print(f"[fuzz] {a.cases} cases, {bad} with mismatches, {time.time() - t0:.0f}s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
