"""Randomised parity sweep of the IVF search chain (coarse ranking on the matrix cores, threshold, fp16 MFMA filter,
refinement, select, exact rescans) against the CPU oracle: rows, dimension, lists, probes, k, batch size, cluster
structure (well separated .. one blob), exact duplicates, vectors on a coarse grid, soft deletes, queries that are
rows of the index.  Ids and distance bits must be the oracle's.  python tools/ivf_fuzz.py [--cases 40] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fvdb_import  # noqa: E402


def one_case(fv, orc, ctx, rng, case, only=-1, log=print):
    n = int(rng.choice([300, 2000, 9000, 40000]))
    d = int(rng.choice([3, 16, 64, 128, 200, 384, 768]))
    nlist = int(rng.choice([1, 4, 24, 100, 256]))
    n_comp = int(rng.choice([1, 4, 64, 4096]))
    sigma = float(rng.choice([0.05, 0.35, 1.0]))
    dup = float(rng.choice([0.0, 0.0, 0.05, 0.5]))
    quant = bool(rng.integers(0, 4) == 0)
    dele = float(rng.choice([0.0, 0.0, 0.1]))
    seed = int(rng.integers(1, 1 << 30))
    if only != -1 and case != only:
        return 0
    nlist = min(nlist, n)
    g = np.random.default_rng(seed)
    means = g.standard_normal((n_comp, d)).astype(np.float32)
    x = means[g.integers(0, n_comp, n)] + np.float32(sigma) * g.standard_normal((n, d)).astype(np.float32)
    if quant:
        x = np.round(x * 2) / 2
    nd = int(dup * n)
    if nd:
        x[g.integers(0, n, nd)] = x[g.integers(0, n, nd)]
    x = np.ascontiguousarray(x, np.float32)
    ids = np.arange(n, dtype=np.uint64) * 5 + 1
    cents = np.ascontiguousarray(x[g.choice(n, nlist, replace=False)])
    gpu = fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(cents)
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=1)
    cpu.set_trained(cents)
    cl, pos = gpu.add(x, ids)
    cpu.batch_insert(ids, x)
    if dele:
        gone = g.choice(n, int(dele * n), replace=False)
        gpu.set_deleted(np.ascontiguousarray(cl[gone]), np.ascontiguousarray(pos[gone]))
        for i in gone.tolist():
            cpu.mark_deleted(int(ids[i]))
    bad, notes = 0, []
    for _ in range(4):
        B = int(g.choice([1, 7, 64, 257]))
        k = int(g.choice([1, 10, 26, 40]))
        npb = int(g.integers(1, nlist + 1)) if g.integers(0, 2) else min(nlist, int(g.choice([1, 2, 8, 32])))
        q = means[g.integers(0, n_comp, B)] + np.float32(sigma) * g.standard_normal((B, d)).astype(np.float32)
        own = g.integers(0, 2, B).astype(bool)
        q[own] = x[g.integers(0, n, int(own.sum()))]
        if quant:
            q = np.round(q * 2) / 2
        q = np.ascontiguousarray(q, np.float32)
        gi, gd, gc = gpu.search(q, k, npb)
        ci, cd, cc = cpu.batch_search(q, k, npb)
        valid = np.arange(k)[None, :] < cc[:, None]
        ok = (np.array_equal(gc, cc) and np.array_equal(gi[valid], ci[valid]) and
              np.array_equal(np.ascontiguousarray(gd).view(np.uint32)[valid], np.ascontiguousarray(cd).view(np.uint32)[valid]))
        if not ok:
            bad += 1
            notes.append(f"B {B} k {k} nprobe {npb}")
    log(f"case {case:3d}: n {n:5d} d {d:3d} lists {nlist:3d} comps {n_comp:4d} sigma {sigma:.2f} dup {dup:.2f} grid {int(quant)} deleted {dele:.1f}"
        f"  -> {'OK' if bad == 0 else 'MISMATCH: ' + '; '.join(notes)}")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
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
    print(f"[ivf fuzz] {a.cases} cases, {bad} with mismatches, {time.time() - t0:.0f}s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
