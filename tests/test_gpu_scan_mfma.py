"""GPU parity of the matrix-core list scan (csrc/kernels_mfma.h).

fp16 MFMA only FILTERS rows; every row that can reach the top k is scored with the reference's sequential
f32 fold (src/core/vector_ops.rs:51-57) and selected by (distance, scan position) = the reference's stable
sort (src/ivf/core.rs:659-678).  Bar: ids identical and distances bit-identical to the oracle and to the
engine's own exact scan (FVDB_SCAN_EXACT), including inputs built to defeat the filter — exact duplicates
(ties at the bound, more candidates than a wave scores), a huge common offset (error bound cannot close), magnitudes beyond fp16, lists too
short to give a threshold — where the verify stage must notice and rescan exactly.
"""
import numpy as np
import pytest

import fvdb_import
import oracle as orc
from _data import bits, mixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fv():
    return fvdb_import.load()


@pytest.fixture(scope="module")
def ctx(fv):
    orc.build()
    c = fv.Context(0)
    yield c
    c.close()


def build(fv, ctx, x, ids, cents, dtype="f32"):
    nlist, d = cents.shape
    gpu = fv.DeviceIVF(ctx, d, nlist, dtype=dtype) if dtype != "f32" else fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(cents)
    cl, pos = gpu.add(x, ids)
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=min(4, nlist))
    cpu.set_trained(cents)
    if dtype == "f16":  # assignment uses the f32 rows; the stored rows are fp16-rounded
        cpu.batch_insert_assigned(ids, x.astype(np.float16).astype(np.float32), cl)
    else:
        cpu.batch_insert(ids, x)
    return gpu, cpu


def same(a, b):
    ai, ad, ac = a
    bi, bd, bc = b
    assert np.array_equal(ac, bc), "hit counts differ"
    for q in range(ai.shape[0]):
        n = int(bc[q])
        assert np.array_equal(ai[q, :n], bi[q, :n]), f"query {q}: ids differ\n{ai[q,:n]}\n{bi[q,:n]}"
        assert np.array_equal(bits(ad[q, :n]), bits(bd[q, :n])), f"query {q}: distances not bit-identical"


def run_modes(gpu, cpu, q, k, nprobe):
    gpu.set_scan_mode(0)
    f0 = gpu.scan_fallbacks()
    auto = gpu.search(q, k, nprobe)
    fb = gpu.scan_fallbacks() - f0
    gpu.set_scan_mode(1)
    exact = gpu.search(q, k, nprobe)
    assert gpu.scan_fallbacks() - f0 == fb  # the exact scan never touches the counter
    gpu.set_scan_mode(0)
    ref = cpu.batch_search(q, k, nprobe)
    same(exact, ref)
    same(auto, ref)
    return fb


@pytest.mark.parametrize("d,nlist,n,B,k,nprobe", [(128, 64, 40000, 100, 10, 8), (384, 128, 60000, 257, 10, 16),
                                                  (64, 32, 20000, 64, 1, 4), (128, 64, 40000, 96, 26, 12),
                                                  (128, 64, 40000, 96, 58, 12),
                                                  (256, 100, 30000, 33, 5, 100), (16, 16, 5000, 40, 10, 3),
                                                  # every list probed by all 200 queries: query groups of 64, 64, 64 and 8
                                                  (128, 16, 20000, 200, 10, 16), (384, 24, 30000, 113, 7, 24),
                                                  # wide rows: 32-query LDS tile (768), wave form (1024)
                                                  (768, 32, 12000, 70, 10, 8), (1024, 16, 6000, 64, 5, 6)])
def test_mfma_scan_matches_oracle(fv, ctx, d, nlist, n, B, k, nprobe):
    x = mixture(n, d, seed=100 + d)
    ids = np.arange(n, dtype=np.uint64) * 3 + 1
    cents = x[np.random.default_rng(7).choice(n, nlist, replace=False)].copy()
    gpu, cpu = build(fv, ctx, x, ids, cents)
    q = mixture(B, d, seed=200 + d)
    fb = run_modes(gpu, cpu, q, k, nprobe)
    assert fb <= B // 8, f"{fb} of {B} queries fell back to the exact scan on ordinary data"
    # database rows as queries: exact self-match first
    gi, gd, gc = gpu.search(x[:64], 3, min(4, nlist))
    assert np.array_equal(gi[:, 0], ids[:64]) and np.all(gd[:, 0] == 0.0)


def test_fp16_rows(fv, ctx):
    d, nlist, n = 768, 64, 20000
    x = mixture(n, d, seed=301)
    ids = np.arange(n, dtype=np.uint64)
    cents = x[np.random.default_rng(8).choice(n, nlist, replace=False)].copy()
    gpu, cpu = build(fv, ctx, x, ids, cents, dtype="f16")
    q = mixture(80, d, seed=302)
    fb = run_modes(gpu, cpu, q, 10, 8)
    assert fb <= 10


def test_deleted_rows_and_incremental_adds(fv, ctx):
    d, nlist, n = 128, 32, 30000
    x = mixture(n, d, seed=401)
    ids = np.arange(n, dtype=np.uint64)
    cents = x[np.random.default_rng(9).choice(n, nlist, replace=False)].copy()
    gpu = fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(cents)
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=4)
    cpu.set_trained(cents)
    cl, pos = np.empty(n, np.uint32), np.empty(n, np.uint32)
    for s in range(0, n, 7001):  # several adds: the pool grows, norms must follow
        cl[s:s + 7001], pos[s:s + 7001] = gpu.add(x[s:s + 7001], ids[s:s + 7001])
        cpu.batch_insert(ids[s:s + 7001], x[s:s + 7001])
    q = x[:120] + np.float32(0.01)
    first = gpu.search(q, 5, 6)
    dead = np.unique(first[0][:, :2].ravel())
    dead = dead[dead != np.uint64(0xFFFFFFFFFFFFFFFF)]
    di = dead.astype(np.int64)
    gpu.set_deleted(cl[di], pos[di], True)
    for i in dead.tolist():
        cpu.mark_deleted(i)
    run_modes(gpu, cpu, q, 5, 6)
    got = gpu.search(q, 5, 6)[0]
    assert not np.isin(got, dead).any()


def test_duplicates_tie_at_the_bound(fv, ctx):
    # every row appears 40 times: the k-th and (k+6)-th distances are equal; all 40 copies are candidates, get the
    # reference's distance, and the stable-sort order comes from their scan positions
    d, nlist = 64, 8
    base = mixture(500, d, seed=501)
    x = np.ascontiguousarray(np.repeat(base, 40, axis=0))
    ids = np.arange(x.shape[0], dtype=np.uint64)
    gpu, cpu = build(fv, ctx, x, ids, base[:nlist].copy())
    q = base[:48] + np.float32(0.001)
    run_modes(gpu, cpu, q, 10, 3)
    # 100 copies: four sets of candidates for the select stage, still no rescan
    x = np.ascontiguousarray(np.repeat(base[:200], 100, axis=0))
    ids = np.arange(x.shape[0], dtype=np.uint64)
    gpu, cpu = build(fv, ctx, x, ids, base[:nlist].copy())
    assert run_modes(gpu, cpu, q, 10, 3) == 0
    # 300 copies: more candidates than the select stage scores (256) -> every query must be rescanned exactly
    x = np.ascontiguousarray(np.repeat(base[:100], 300, axis=0))
    ids = np.arange(x.shape[0], dtype=np.uint64)
    gpu, cpu = build(fv, ctx, x, ids, base[:nlist].copy())
    fb = run_modes(gpu, cpu, q, 10, 3)
    assert fb == q.shape[0]


def test_large_common_offset_and_fp16_overflow(fv, ctx):
    d, nlist, n = 64, 16, 8000
    rng = np.random.default_rng(601)
    for offset in (3000.0, 1.0e5):  # bound cannot close / |x| beyond the fp16 range
        x = (offset + rng.standard_normal((n, d))).astype(np.float32)
        ids = np.arange(n, dtype=np.uint64)
        cents = x[:nlist].copy()
        gpu, cpu = build(fv, ctx, x, ids, cents)
        q = (offset + rng.standard_normal((40, d))).astype(np.float32)
        run_modes(gpu, cpu, q, 10, 4)


def test_short_lists_have_no_threshold(fv, ctx):
    # 3 rows per list: phase A cannot produce k+6 distances, every row survives, still exact
    d, nlist = 32, 64
    x = mixture(nlist * 3, d, seed=701)
    ids = np.arange(x.shape[0], dtype=np.uint64)
    cents = x[:nlist].copy()
    gpu, cpu = build(fv, ctx, x, ids, cents)
    q = mixture(50, d, seed=702)
    run_modes(gpu, cpu, q, 10, 5)
    run_modes(gpu, cpu, q, 10, 64)


@pytest.mark.parametrize("dtype,scale", [("f32", 1.0), ("f32", 30.0), ("f16", 1.0), ("f32", 1e-3)])
def test_matrix_core_values_stay_inside_the_error_bound(fv, ctx, dtype, scale):
    # every survivor's v + |q|^2 must lie within E_q of the reference's f32 sum (kernels_mfma.h: mfma_error_bound) —
    # the inequality the whole filter rests on, checked here on thousands of (row, query) pairs and several magnitudes
    d, nlist, n, B, nprobe = 128, 16, 12000, 64, 4
    x = (mixture(n, d, seed=901) * np.float32(scale)).astype(np.float32)
    q = (mixture(B, d, seed=902) * np.float32(scale)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    cents = x[np.random.default_rng(11).choice(n, nlist, replace=False)].copy()
    gpu = fv.DeviceIVF(ctx, d, nlist, dtype=dtype) if dtype != "f32" else fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(cents)
    cl, pos = gpu.add(x, ids)
    rows = x.astype(np.float16).astype(np.float32) if dtype == "f16" else x  # what the reference would be given
    gpu.search(q, 10, nprobe)
    probes, _ = gpu.coarse(q, nprobe)
    # rows of each list in position order
    order = np.lexsort((pos, cl))
    start = np.searchsorted(cl[order], np.arange(nlist))
    xmax = np.sqrt(np.max(np.sum(rows.astype(np.float64) ** 2, axis=1)))
    u10, u11, u14, u24 = 2.0 ** -10, 2.0 ** -11, 2.0 ** -14, 2.0 ** -24
    ux = 0.0 if dtype == "f16" else u11  # fp16 mirror of f32 rows is rounded to nearest; fp16 rows are exact
    checked = 0
    for b in range(B):
        rk, ps, v = gpu.scan_survivor_dump(b)
        if rk.size == 0:
            continue
        nq = np.sqrt(np.sum(q[b].astype(np.float64) ** 2))
        E = 1.05 * (2 * (ux + u11 + ux * u11) * xmax * nq + 2 * (2 * d * u24) * xmax * nq
                    + 1.01 * (2 * d + 16) * u24 * (xmax + nq) ** 2 + 2 * u14 * np.sqrt(d) * (xmax + nq)) \
            + 1e-6 * (xmax + nq) ** 2
        r = rows[order[start[probes[b, rk]] + ps]]
        ref = orc.l2_batch(q[b], r).astype(np.float64) ** 2  # reference distances, squared back (adds <= 2^-23 relative)
        qn = np.float32(np.sum(q[b].astype(np.float32) ** 2))
        err = np.abs(v.astype(np.float64) + float(qn) - ref)
        assert np.all(err <= E + 2.0 ** -22 * ref), f"query {b}: max err {err.max():.3e} > bound {E:.3e}"
        checked += rk.size
    assert checked > 500


def test_isotropic_survey_generator_has_no_cliff(fv, ctx):
    # SURVEY §8d's own generator (component means ~ N(0, I_384), rows = mean + 0.35 N(0, I)): in-component distances are
    # concentrated, several components share a list, and for a good share of the queries the list the threshold is sampled
    # from does not hold their component — the first filter pass then lets thousands of rows through.  Those queries get a
    # threshold from their own survivors and a second filter pass (refine_threshold_kernel); nobody is rescanned exactly,
    # and the answers are the exact scan's and the oracle's bit for bit.
    d, n_comp, n, nlist, B, k, nprobe = 384, 1024, 160_000, 256, 256, 10, 32
    rng = np.random.default_rng(77)
    means = rng.standard_normal((n_comp, d)).astype(np.float32)
    x = (means[rng.integers(0, n_comp, n)] + np.float32(0.35) * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    q = (means[rng.integers(0, n_comp, B)] + np.float32(0.35) * rng.standard_normal((B, d)).astype(np.float32)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    ivf = fv.IVFIndex(ctx, n_clusters=nlist, n_probe=nprobe, train_size=40_000, max_iterations=10, seed=3)
    ivf.train(x[:40_000])
    ivf.batch_insert(ids, x)
    h = ivf._dev()
    import ctypes as C
    lib = ctx.lib

    def counters():
        v = C.c_uint64(0)
        ctx.check(lib.fvdb_ivf_scan_fallbacks(h, C.byref(v)))
        r = (C.c_uint64 * 5)()
        ctx.check(lib.fvdb_ivf_scan_fallback_reasons(h, r))
        return v.value, list(r)

    ctx.check(lib.fvdb_ivf_set_scan_mode(h, 2))  # the filter for every batch, second pass always enqueued
    f0, r0 = counters()
    got = ivf.search(q, k, nprobe)
    f1, r1 = counters()
    ctx.check(lib.fvdb_ivf_set_scan_mode(h, 1))
    exact = ivf.search(q, k, nprobe)
    assert np.array_equal(got.ids, exact.ids) and np.array_equal(bits(got.distances), bits(exact.distances))
    assert np.array_equal(got.counts, exact.counts)
    assert f1 - f0 == 0, (f1 - f0, [b - a for a, b in zip(r0, r1)])   # nobody rescanned exactly ...
    assert r1[4] - r0[4] > 0                                           # ... because the loose ones were refined
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
    cpu.set_trained(ivf.get_centroids())
    cpu.batch_insert(ids, x)
    oi, od, oc = cpu.batch_search(q[:48], k, nprobe, threads=4)
    assert np.array_equal(got.ids[:48], oi) and np.array_equal(bits(got.distances[:48]), bits(od))
    # AUTO learns it from the counters (looked at every 2048 queries): once more than one query in a thousand overflows,
    # the second pass is enqueued
    ctx.check(lib.fvdb_ivf_set_scan_mode(h, 0))
    for _ in range(30):
        auto = ivf.search(q, k, nprobe)
        assert np.array_equal(auto.ids, exact.ids) and np.array_equal(bits(auto.distances), bits(exact.distances))
    f2, r2 = counters()
    assert r2[4] > r1[4]
