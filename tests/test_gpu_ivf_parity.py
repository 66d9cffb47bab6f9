"""GPU parity: the HIP IVF path through the C ABI vs the CPU oracle on identical index
structures (same centroids, same insertion order).  Bar: neighbour ids identical, distances
bit-identical (the kernels perform the reference's f32 arithmetic in the reference's order)."""
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


def build_pair(fv, ctx, x, ids, centroids, nprobe_default=4):
    nlist, d = centroids.shape
    gpu = fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(centroids)
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=min(nprobe_default, nlist))
    cpu.set_trained(centroids)
    cl, pos = gpu.add(x, ids)
    cpu.batch_insert(ids, x)
    return gpu, cpu, cl, pos


def assert_same(gpu_res, cpu_res):
    gi, gd, gc = gpu_res
    ci, cd, cc = cpu_res
    assert np.array_equal(gc, cc), "hit counts differ"
    for q in range(gi.shape[0]):
        n = int(cc[q])
        assert np.array_equal(gi[q, :n], ci[q, :n]), f"query {q}: ids differ\n{gi[q,:n]}\n{ci[q,:n]}"
        assert np.array_equal(bits(gd[q, :n]), bits(cd[q, :n])), f"query {q}: distances not bit-identical"
        assert np.all(gi[q, n:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isinf(gd[q, n:]))


def test_reference_toy_index(fv, ctx):
    # the reference's own 2-D fixture: tests/ivf/core.rs:441-474, :345-383
    cents = np.array([[0.1, 0.0], [5.0, 5.0], [-5.0, -5.0]], np.float32)
    x = np.array([[0, 0], [5, 5], [-5, -5], [2.5, 2.5]], np.float32)
    gpu, cpu, cl, _ = build_pair(fv, ctx, x, np.arange(4, dtype=np.uint64), cents)
    assert list(cl) == [cpu.find_cluster(v) for v in x]
    q = np.array([[2.5, 2.5], [0.0, 0.0], [-4.0, -4.0]], np.float32)
    for npb in (1, 2, 3, 10):
        assert_same(gpu.search(q, 4, npb), cpu.batch_search(q, 4, npb))
    ids, ds, cnt = gpu.search(q[:1], 4, 2)
    assert 3 <= cnt[0] <= 4 and 3 in ids[0, : cnt[0]] and ds[0, 0] == 0.0


@pytest.mark.parametrize("d", [384, 10, 37, 3])
def test_random_parity(fv, ctx, d):
    n, nlist, B = 6000, 48, 70
    x = mixture(n, d, seed=11 + d)
    ids = (np.arange(n, dtype=np.uint64) * 7 + 3)
    cents = x[np.random.default_rng(5).choice(n, nlist, replace=False)].copy()
    gpu, cpu, cl, pos = build_pair(fv, ctx, x, ids, cents)
    assert np.array_equal(cl, cpu.assign(x)), "find_nearest_centroid parity"
    assert np.array_equal(gpu.assign(x[:500]), cpu.assign(x[:500]))
    assert np.array_equal(gpu.list_sizes(), np.array([cpu.get_cluster_size(c) for c in range(nlist)], np.uint64))
    q = mixture(B, d, seed=99 + d)
    for k, npb in ((10, 8), (1, 1), (10, nlist), (64, 5), (100, 7), (200, 3)):
        assert_same(gpu.search(q, k, npb), cpu.batch_search(q, k, npb))
    # database rows as queries: exact self-match first (tests/ivf/core.rs:400-413)
    gi, gd, gc = gpu.search(x[:40], 3, 4)
    assert np.array_equal(gi[:, 0], ids[:40]) and np.all(gd[:, 0] == 0.0)


def test_coarse_order_and_distances(fv, ctx):
    d, nlist = 64, 200
    cents = mixture(nlist, d, seed=3)
    gpu = fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(cents)
    q = mixture(33, d, seed=4)
    cl, ds = gpu.coarse(q, 32)
    for i in range(q.shape[0]):
        dist = orc.l2_batch(q[i], cents)
        order = np.argsort(dist, kind="stable")[:32]  # stable: lowest cluster id wins ties
        assert np.array_equal(cl[i], order.astype(np.uint32))
        assert np.array_equal(bits(ds[i]), bits(dist[order]))


def test_ties_keep_scan_order(fv, ctx):
    # duplicates => exact distance ties; order must be (probe rank, position in list)
    d, nlist = 8, 4
    rng = np.random.default_rng(8)
    base = rng.standard_normal((40, d)).astype(np.float32)
    x = np.concatenate([base, base, base[:10]])  # every row 2-3 times
    ids = np.arange(x.shape[0], dtype=np.uint64) + 1000
    cents = base[:nlist].copy()
    gpu, cpu, _, _ = build_pair(fv, ctx, x, ids, cents)
    q = base[:20] + np.float32(0.01)
    for k, npb in ((5, 2), (30, 4), (90, 4)):
        assert_same(gpu.search(q, k, npb), cpu.batch_search(q, k, npb))


def test_deleted_rows_are_skipped(fv, ctx):
    # src/ivf/core.rs:666-669, tests/unit/ivf_deletion_tests.rs:102-128
    n, d, nlist = 3000, 32, 16
    x = mixture(n, d, seed=21)
    ids = np.arange(n, dtype=np.uint64)
    cents = x[:nlist].copy()
    gpu, cpu, cl, pos = build_pair(fv, ctx, x, ids, cents)
    dead = np.random.default_rng(2).choice(n, 400, replace=False)
    gpu.set_deleted(cl[dead], pos[dead], True)
    for i in dead:
        cpu.mark_deleted(int(i))
    q = x[dead[:50]]  # query the deleted rows themselves
    res = gpu.search(q, 10, 6)
    assert_same(res, cpu.batch_search(q, 10, 6))
    assert not np.isin(res[0], dead.astype(np.uint64)).any()
    gpu.set_deleted(cl[dead[:10]], pos[dead[:10]], False)  # undelete restores them
    ids2, ds2, _ = gpu.search(x[dead[:10]], 1, 6)
    assert np.array_equal(ids2[:, 0], dead[:10].astype(np.uint64)) and np.all(ds2[:, 0] == 0.0)


def test_short_results_and_empty_index(fv, ctx):
    # tests/ivf/core.rs:314-321, :385-398
    cents = np.array([[0.1, 0.0], [5.0, 5.0], [-5.0, -5.0]], np.float32)
    gpu = fv.DeviceIVF(ctx, 2, 3)
    gpu.set_centroids(cents)
    ids, ds, cnt = gpu.search([[1.0, 1.0]], 5, 2)
    assert cnt[0] == 0 and np.all(np.isinf(ds))
    gpu.add(np.array([[0, 0], [1, 1], [2, 2]], np.float32), np.arange(3, dtype=np.uint64))
    ids, ds, cnt = gpu.search([[1.5, 1.5]], 10, 3)
    assert cnt[0] == 3 and sorted(ids[0, :3].tolist()) == [0, 1, 2]


def test_errors(fv, ctx):
    gpu = fv.DeviceIVF(ctx, 2, 3)
    with pytest.raises(fv.NotTrained):  # tests/ivf/core.rs:224-234
        gpu.search([[1.0, 2.0]], 1, 1)
    with pytest.raises(fv.NotTrained):
        gpu.add(np.zeros((1, 2), np.float32), np.zeros(1, np.uint64))
    gpu.set_centroids(np.eye(3, 2, dtype=np.float32))
    with pytest.raises(fv.DimensionMismatch):  # tests/ivf/core.rs:294-308
        gpu.search([[1.0, 2.0, 3.0]], 1, 1)
    with pytest.raises(fv.NonFiniteInput):
        gpu.search([[np.nan, 0.0]], 1, 1)
    with pytest.raises(fv.Unsupported):
        gpu.search([[0.0, 0.0]], 257, 1)


def test_search_all_is_exact_knn(fv, ctx):
    n, d, nlist = 5000, 96, 32
    x = mixture(n, d, seed=31)
    ids = np.arange(n, dtype=np.uint64)
    gpu, cpu, _, _ = build_pair(fv, ctx, x, ids, x[:nlist].copy())
    q = mixture(25, d, seed=32)
    gi, gd, gc = gpu.search_all(q, 10)
    for i in range(q.shape[0]):
        dist = orc.l2_batch(q[i], x)
        best = np.sort(dist, kind="stable")[:10]
        assert np.array_equal(bits(gd[i]), bits(best))
        assert np.array_equal(bits(dist[gi[i].astype(np.int64)]), bits(best))


def test_incremental_adds_match_bulk(fv, ctx):
    n, d, nlist = 2000, 20, 8
    x = mixture(n, d, seed=41)
    ids = np.arange(n, dtype=np.uint64)
    cents = x[:nlist].copy()
    gpu, cpu, _, _ = build_pair(fv, ctx, x[:0].reshape(0, d), ids[:0], cents)
    for s in range(0, n, 333):  # grows the pool and the partly filled tail blocks repeatedly
        gpu.add(x[s:s + 333], ids[s:s + 333])
        cpu.batch_insert(ids[s:s + 333], x[s:s + 333])
    q = mixture(30, d, seed=42)
    assert_same(gpu.search(q, 10, 3), cpu.batch_search(q, 10, 3))


def test_dot_and_cosine_utilities(fv, ctx):
    # a3: dot_product_scalar / cosine_similarity_scalar (src/core/vector_ops.rs:35-49), bit-identical folds
    # tests/core/vector_ops.rs:12-26, tests/core/vector_ops_advanced.rs:13-45
    q = np.array([[1.0, 0.0, 0.0]], np.float32)
    x = np.array([[1, 0, 0], [0, 1, 0], [0.707, 0.707, 0], [0, 0, 0]], np.float32)
    sims = fv.engine.batch_cosine_similarity(ctx, q, x)[0]
    assert abs(sims[0] - 1.0) < 1e-6 and abs(sims[1]) < 1e-6 and abs(sims[2] - 0.707) < 0.01 and sims[3] == 0.0
    for size in (16, 64, 128, 256, 512, 1024):
        a = np.sin(np.arange(size, dtype=np.float32)).astype(np.float32)
        b = np.cos(np.arange(size, dtype=np.float32)).astype(np.float32)
        assert fv.engine.dot_products(ctx, a, b)[0, 0] == np.float32(orc.dot_product_scalar(a, b))
    ones = np.ones((1, 256), np.float32)
    assert fv.engine.dot_products(ctx, ones, ones)[0, 0] == 256.0
    assert abs(fv.engine.batch_cosine_similarity(ctx, ones, ones)[0, 0] - 1.0) < 1e-6
    qq, xx = mixture(7, 96, seed=201), mixture(33, 96, seed=202)
    dots, coss = fv.engine.dot_products(ctx, qq, xx), fv.engine.batch_cosine_similarity(ctx, qq, xx)
    for i in range(7):
        for j in range(33):
            assert dots[i, j] == np.float32(orc.dot_product_scalar(qq[i], xx[j]))
            assert coss[i, j] == np.float32(orc.cosine_similarity_scalar(qq[i], xx[j]))
            assert -1.0 - 1e-6 <= coss[i, j] <= 1.0 + 1e-6


def test_split_stages_and_slots_match_the_one_call_search(fv, ctx):
    # fvdb_ivf_coarse_dev_slot + fvdb_ivf_search_probes_dev_slot (the multi-GPU path's two halves), on the index's own
    # context and on a second context with another scratch slot, give exactly fvdb_ivf_search's results
    n, d, nlist, B, k, nprobe = 30000, 64, 32, 96, 10, 6
    x = mixture(n, d, seed=811)
    ids = np.arange(n, dtype=np.uint64) + 11
    cents = x[np.random.default_rng(3).choice(n, nlist, replace=False)].copy()
    gpu, cpu, cl, pos = build_pair(fv, ctx, x, ids, cents)
    q = mixture(B, d, seed=812)
    want = gpu.search(q, k, nprobe)
    assert_same(want, cpu.batch_search(q, k, nprobe))
    lib = ctx.lib
    other = fv.Context(0)
    try:
        for on, slot, c in ((None, 0, ctx), (other.h, 1, other), (other.h, 3, other)):
            qd = c.upload(q)
            pr, oi, od, oc = c.alloc(B * nprobe * 4), c.alloc(B * k * 8), c.alloc(B * k * 4), c.alloc(B * 4)
            ctx.check(lib.fvdb_ivf_coarse_dev_slot(gpu.h, on, slot, qd, B, nprobe, pr))
            ctx.check(lib.fvdb_ivf_search_probes_dev_slot(gpu.h, on, slot, qd, pr, B, k, nprobe, oi, od, oc, None))
            c.synchronize()
            probes = c.download(pr, (B, nprobe), np.uint32)
            cl_want, _ = gpu.coarse(q, nprobe)
            assert np.array_equal(probes, cl_want)
            got = (c.download(oi, (B, k), np.uint64), c.download(od, (B, k), np.float32), c.download(oc, (B,), np.uint32))
            assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0])
            assert np.array_equal(bits(got[1]), bits(want[1]))
            for b_ in (qd, pr, oi, od, oc):
                c.free(b_)
    finally:
        other.close()


def test_randomised_shapes_search_like_the_oracle(fv, ctx):
    # tools/ivf_fuzz.py: rows, dimension, lists, probes, k, batch size, cluster structure, exact duplicates, vectors on a
    # coarse grid, soft deletes, queries that are rows of the index — the whole chain (coarse ranking, threshold, fp16
    # filter, refinement, select, exact rescans) against the oracle, four searches per index
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ivf_fuzz
    rng = np.random.default_rng(1)
    failed = [c for c in range(30) if ivf_fuzz.one_case(fv, orc, ctx, rng, c, c if c % 3 == 0 else -2)]
    assert not failed, failed
