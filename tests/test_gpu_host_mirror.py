"""GPU parity of the host mirror (IVFIndex / HNSWIndex / HybridIndex over the C ABI) against
the CPU oracle: identical graphs, identical neighbour ids, bit-identical distances.
Also restates the reference's behavioural tests on the GPU path."""
import math

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


def assert_same_results(g, cpu_ids, cpu_ds, cpu_cnt):
    assert np.array_equal(g.counts, cpu_cnt)
    for b in range(len(g)):
        n = int(cpu_cnt[b])
        assert np.array_equal(g.ids[b, :n], cpu_ids[b, :n]), f"query {b}"
        assert np.array_equal(bits(g.distances[b, :n]), bits(cpu_ds[b, :n])), f"query {b}"


def same_graph(gh, oh, ids):
    assert gh.entry_point() == oh.entry_point()
    for i in ids:
        lv = oh.level(i)
        assert gh.level(i) == lv
        for l in range(lv + 1):
            assert gh.neighbors(i, l) == oh.neighbors(i, l), f"node {i} layer {l}"


# ---- HNSW -------------------------------------------------------------------------------
def test_hnsw_insert_builds_identical_graph_and_search(fv, ctx):
    n, d = 260, 16
    x = mixture(n, d, n_comp=6, seed=5)
    ids = np.arange(n, dtype=np.uint64) + 10
    levels = orc.rng_levels(42, n)
    gh = fv.HNSWIndex(ctx, 6, 12, 40, seed=42)
    oh = orc.HNSWIndex(6, 12, 40, seed=42)
    gh.batch_insert(ids, x, levels)
    oh.batch_insert(ids, x, levels)
    assert gh.node_count() == n
    same_graph(gh, oh, ids)
    q = mixture(40, d, n_comp=6, seed=6)
    for k, ef in ((10, 50), (1, 1), (5, 200), (30, 30)):
        assert_same_results(gh.search(q, k, ef), *oh.batch_search(q, k, ef))
    assert gh.dist_evals() > 0 and gh.hops() > 0


def test_hnsw_own_level_draws_match_oracle_prng(fv, ctx):
    # both sides draw levels from SplitMix64(seed): same graph without forcing levels
    n, d = 120, 8
    x = mixture(n, d, n_comp=4, seed=15)
    ids = np.arange(n, dtype=np.uint64)
    gh, oh = fv.HNSWIndex(ctx, 4, 8, 30, seed=7), orc.HNSWIndex(4, 8, 30, seed=7)
    gh.batch_insert(ids, x)
    oh.batch_insert(ids, x)
    same_graph(gh, oh, ids)


def test_hnsw_reference_behaviour(fv, ctx):
    # tests/hnsw/core.rs:174-181 empty -> []
    ix = fv.HNSWIndex(ctx)
    assert ix.search([[1.0, 2.0, 3.0]], 5, 200).counts[0] == 0 and ix.entry_point() is None
    # :183-197 single node exact match; :152-166 duplicate; dimension mismatch
    ix.insert(7, [1.0, 2.0, 3.0])
    r = ix.search([[1.0, 2.0, 3.0]], 1, 200)
    assert r.counts[0] == 1 and r.ids[0, 0] == 7 and r.distances[0, 0] < 1e-6
    with pytest.raises(fv.DuplicateVector):
        ix.insert(7, [1.0, 2.0, 3.0])
    with pytest.raises(fv.DimensionMismatch):
        ix.insert(8, [1.0, 2.0])
    with pytest.raises(fv.DimensionMismatch):
        ix.search([[1.0, 2.0]], 1, 10)
    # :228-257 2-D cross
    ix = fv.HNSWIndex(ctx)
    for i, v in enumerate([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1], [0.5, 0.5]]):
        ix.insert(i, v)
    r = ix.search([[0.1, 0.1]], 3, 200)
    assert r.counts[0] == 3 and r.ids[0, 0] == 0
    # :128-150 line: node 2 linked to 1 and 3; :300-316 k > n
    ix = fv.HNSWIndex(ctx)
    for i in range(5):
        ix.insert(i, [float(i)])
    assert {1, 3} <= set(ix.neighbors(2, 0))
    assert ix.search([[1.5]], 10, 200).counts[0] == 5


def test_hnsw_search_accuracy_self_match(fv, ctx):
    # tests/hnsw/core.rs:199-226
    ix = fv.HNSWIndex(ctx, 16, 32, 200, seed=42)
    vecs = np.array([[math.sin(float(i * j)) for j in range(10)] for i in range(100)], np.float32)
    ix.batch_insert(np.arange(100, dtype=np.uint64), vecs)
    r = ix.search(vecs, 1, 200)  # all 100 queries in one lock-step batch
    assert np.all(r.counts == 1) and np.array_equal(r.ids[:, 0], np.arange(100, dtype=np.uint64))
    assert np.all(r.distances[:, 0] < 1e-5)
    for i in range(100):  # degree caps :87-126
        assert len(ix.neighbors(i, 0)) <= 32


def test_hnsw_deleted_nodes(fv, ctx):
    n, d = 150, 12
    x = mixture(n, d, n_comp=3, seed=25)
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(3, n)
    gh, oh = fv.HNSWIndex(ctx, 8, 16, 60, seed=3), orc.HNSWIndex(8, 16, 60, seed=3)
    gh.batch_insert(ids, x, levels)
    oh.batch_insert(ids, x, levels)
    for i in (3, 17, 40, 41, 99):
        gh.mark_deleted(i)
        oh.mark_deleted(i)
    with pytest.raises(fv.VectorNotFound):
        gh.mark_deleted(10_000)
    q = x[[3, 17, 40, 60, 99]]
    g = gh.search(q, 8, 40)
    assert_same_results(g, *oh.batch_search(q, 8, 40))
    assert not np.isin(g.ids, [3, 17, 40, 41, 99]).any()


def test_hnsw_bulk_build_and_restore_parity(fv, ctx):
    n, d = 3000, 24
    x = mixture(n, d, n_comp=8, sigma=1.0, seed=35)  # overlapping components: a connected k-NN graph
    ids = np.arange(n, dtype=np.uint64) * 3 + 1
    gh = fv.HNSWIndex(ctx, 16, 32, 200, seed=11)
    gh.bulk_build(ids, x)
    assert gh.node_count() == n
    gi, lv, off, nb = gh.export_graph()
    assert np.array_equal(gi, ids) and lv.max() >= 2
    # layer-0 neighbours are the exact 32 nearest
    some = [0, 17, 2999]
    for i in some:
        dist = orc.l2_batch(x[i], x)
        dist[i] = np.inf
        want = set(ids[np.argsort(dist, kind="stable")[:32]].tolist())
        assert set(gh.neighbors(int(ids[i]), 0)) == want
    # same graph in the oracle => identical search results
    oh = orc.HNSWIndex(16, 32, 200, seed=11)
    oh.restore(gi, x, lv, off, nb, gh.entry_point())
    q = mixture(64, d, n_comp=8, sigma=1.0, seed=36)
    g = gh.search(q, 10, 50)
    assert_same_results(g, *oh.batch_search(q, 10, 50))
    # and it is a usable index: recall vs exact
    hits = 0
    for b in range(q.shape[0]):
        exact = set(ids[np.argsort(orc.l2_batch(q[b], x), kind="stable")[:10]].tolist())
        hits += len(exact & set(g.ids[b, :10].tolist()))
    assert hits / (10 * q.shape[0]) > 0.8


def test_device_traversal_equals_host_walk_and_oracle(fv, ctx):
    # the layered walk on the GPU (one launch per batch) vs the host walk (one scoring launch per hop)
    n, d = 2500, 40
    x = mixture(n, d, n_comp=6, sigma=1.0, seed=61)
    ids = np.arange(n, dtype=np.uint64) * 5 + 2
    gh = fv.HNSWIndex(ctx, 16, 32, 200, seed=13)
    gh.bulk_build(ids, x)
    gi, lv, off, nb = gh.export_graph()
    oh = orc.HNSWIndex(16, 32, 200, seed=13)
    oh.restore(gi, x, lv, off, nb, gh.entry_point())
    for i in (7, 99, 1500, 2499):
        gh.mark_deleted(int(ids[i]))
        oh.mark_deleted(int(ids[i]))
    q = np.concatenate([mixture(90, d, n_comp=6, sigma=1.0, seed=62), x[[7, 99, 300]]])
    assert gh.device_traversal()
    for k, ef in ((10, 50), (1, 1), (10, 10), (40, 200), (5, 333)):
        dev = gh.search(q, k, ef)
        gh.set_device_traversal(False)
        host = gh.search(q, k, ef)
        gh.set_device_traversal(True)
        assert np.array_equal(dev.counts, host.counts) and np.array_equal(dev.ids, host.ids)
        assert np.array_equal(bits(dev.distances), bits(host.distances))
        assert_same_results(dev, *oh.batch_search(q, k, ef))
    assert gh.device_fallbacks() == 0


def test_device_traversal_duplicate_vectors_tie_order(fv, ctx):
    # exact distance ties exercise the restated BinaryHeap order inside the kernel
    d = 6
    base = mixture(60, d, n_comp=3, sigma=1.0, seed=63)
    x = np.concatenate([base, base, base])
    ids = np.arange(x.shape[0], dtype=np.uint64)
    levels = orc.rng_levels(5, x.shape[0])
    gh, oh = fv.HNSWIndex(ctx, 6, 12, 40, seed=5), orc.HNSWIndex(6, 12, 40, seed=5)
    gh.batch_insert(ids, x, levels)
    oh.batch_insert(ids, x, levels)
    same_graph(gh, oh, ids)
    q = base[:30] + np.float32(0.001)
    for k, ef in ((9, 30), (20, 60)):
        assert_same_results(gh.search(q, k, ef), *oh.batch_search(q, k, ef))


def test_device_traversal_overflow_falls_back_to_host_walk(fv, ctx, monkeypatch):
    n, d = 1200, 12
    x = mixture(n, d, n_comp=4, sigma=1.0, seed=64)
    ids = np.arange(n, dtype=np.uint64)
    gh = fv.HNSWIndex(ctx, 8, 16, 60, seed=9)
    gh.bulk_build(ids, x)
    q = mixture(33, d, n_comp=4, sigma=1.0, seed=65)
    gh.set_device_traversal(False)
    want = gh.search(q, 10, 100)
    gh.set_device_traversal(True)
    monkeypatch.setenv("FVDB_GRAPH_TCAP", "40")  # visited log of 40 nodes: every query outgrows it ...
    got = gh.search(q, 10, 100)
    assert gh.device_fallbacks() == 0             # ... and clears its whole map instead: still finished on the device
    assert np.array_equal(got.ids, want.ids) and np.array_equal(bits(got.distances), bits(want.distances))
    for ef in (100, 40):                          # restated-heap kernel / sorted-register kernel
        got = gh.search(q, 10, ef)
        gh.set_device_traversal(False)
        w2 = gh.search(q, 10, ef)
        gh.set_device_traversal(True)
        assert np.array_equal(got.ids, w2.ids) and np.array_equal(bits(got.distances), bits(w2.distances))
    monkeypatch.delenv("FVDB_GRAPH_TCAP")
    monkeypatch.setenv("FVDB_GRAPH_CAND_CAP", "24")  # a candidate heap of 24 slots: every ef = 100 query overflows it
    got = gh.search(q, 10, 100)
    assert gh.device_fallbacks() == q.shape[0]
    assert np.array_equal(got.ids, want.ids) and np.array_equal(bits(got.distances), bits(want.distances))
    monkeypatch.delenv("FVDB_GRAPH_CAND_CAP")
    again = gh.search(q, 10, 100)  # bitmaps were left clean by the aborted walks
    assert np.array_equal(again.ids, want.ids) and gh.device_fallbacks() == q.shape[0]


# ---- IVFIndex mirror -----------------------------------------------------------------------
TRAIN9 = [[0.0, 0.0], [0.1, 0.1], [0.2, -0.1], [5.0, 5.0], [5.1, 4.9], [4.9, 5.1],
          [-5.0, -5.0], [-4.9, -5.1], [-5.1, -4.9]]


def test_ivf_index_reference_behaviour(fv, ctx):
    ix = fv.IVFIndex(ctx)
    with pytest.raises(fv.NotTrained):  # tests/ivf/core.rs:224-234
        ix.insert(1, [1.0, 2.0])
    with pytest.raises(fv.InvalidConfig):  # IVFConfig::is_valid (:62-70)
        fv.IVFIndex(ctx, n_clusters=2, n_probe=3)
    ix = fv.IVFIndex(ctx, n_clusters=10, n_probe=1, max_iterations=10)
    with pytest.raises(fv.InsufficientTrainingData):  # :157-193
        ix.train([[1.0, 2.0], [3.0, 4.0]])
    ix = fv.IVFIndex(ctx, n_clusters=2, n_probe=1, max_iterations=10)
    with pytest.raises(fv.InconsistentDimensions):  # :195-218
        ix.train([[1.0, 2.0, 3.0], [4.0, 5.0], [6.0, 7.0, 8.0]])
    ix = fv.IVFIndex(ctx, n_clusters=3, n_probe=2, train_size=9, max_iterations=10, seed=42)
    res = ix.train(TRAIN9)
    assert res["final_error"] <= res["initial_error"]  # :125-155
    cents = ix.get_centroids()
    for exp in ([0.1, 0.0], [5.0, 5.0], [-5.0, -5.0]):  # :69-122
        assert min(orc.euclidean_distance_scalar(c, exp) for c in cents) < 1.0
    ix.insert(7, [1.0, 1.0])
    assert ix.total_vectors() == 1 and ix.get_cluster_size(ix.find_cluster([1.0, 1.0])) > 0
    with pytest.raises(fv.DuplicateVector):  # :272-291
        ix.insert(7, [1.0, 1.0])
    with pytest.raises(fv.DimensionMismatch):  # :294-308
        ix.insert(8, [1.0, 2.0, 3.0])
    for i in range(20):  # :415-437
        ang = np.float32(i) * np.float32(math.pi) / np.float32(10.0)
        ix.insert(100 + i, [np.cos(ang) * 5.0, np.sin(ang) * 5.0])
    assert ix.search([[0.0, 0.0]], 10, 1).counts[0] <= ix.search([[0.0, 0.0]], 10, 3).counts[0]
    ix.mark_deleted(7)  # tests/unit/ivf_deletion_tests.rs:102-128
    assert ix.is_deleted(7) and ix.active_count() == ix.total_vectors() - 1
    assert 7 not in ix.search([[1.0, 1.0]], 5, 3).ids[0]
    with pytest.raises(fv.VectorNotFound):
        ix.mark_deleted(424242)
    ok, failed = ix.batch_insert([200, 200, 201], [[0.0, 0.1], [0.0, 0.1], [9.0, 9.0]])
    assert (ok, failed) == (2, 1)


def test_ivf_train_matches_oracle(fv, ctx):
    # same SplitMix64 draws + same sequential sums => same centroids, bit for bit
    n, d, nlist = 1500, 24, 12
    x = mixture(n, d, n_comp=12, seed=45)
    g = fv.IVFIndex(ctx, n_clusters=nlist, n_probe=4, max_iterations=8, seed=9)
    o = orc.IVFIndex(n_clusters=nlist, n_probe=4, max_iterations=8, seed=9)
    rg, ro = g.train(x), o.train(x)
    assert rg["iterations"] == ro["iterations"] and rg["converged"] == ro["converged"]
    assert np.array_equal(bits(g.get_centroids()), bits(o.get_centroids()))
    assert np.float32(rg["initial_error"]) == np.float32(ro["initial_error"])
    assert np.float32(rg["final_error"]) == np.float32(ro["final_error"])
    ids = np.arange(n, dtype=np.uint64)
    g.batch_insert(ids, x)
    o.batch_insert(ids, x)
    q = mixture(30, d, n_comp=12, seed=46)
    assert_same_results(g.search(q, 10, 4), *o.batch_search(q, 10, 4))


# ---- Hybrid -----------------------------------------------------------------------------------
DAY = 86400.0


def create_training_data():
    return [[float(i), float(i) * 0.5] for i in range(10)]


def test_hybrid_reference_behaviour(fv, ctx):
    ix = fv.HybridIndex(ctx)
    assert ix.search([[1.0, 2.0]], 5).counts[0] == 0  # tests/hybrid/core.rs:167-174
    with pytest.raises(fv.NotInitialized):
        ix.insert(1, [0.0, 0.0])
    ix.initialize(create_training_data())
    now = 100 * DAY
    for i in range(3):  # :241-288 mixed
        ix.insert(i, [float(i), float(i)], now=now)
    for i in range(3, 6):
        ix.insert_with_timestamp(i, [float(i), float(i)], now - 30 * DAY, now)
    assert ix.recent_count() == 3 and ix.historical_count() == 3
    r = ix.search([[2.5, 2.5]], 6, now=now)
    assert r.counts[0] == 6
    rec = sum(1 for i in r.ids[0] if i < 3)
    assert rec > 0 and 6 - rec > 0
    assert np.all(np.diff(r.distances[0]) >= 0)
    assert np.all((r.scores() >= 0) & (r.scores() <= 1))  # bindings/node/test/session.test.js:161-163
    with pytest.raises(fv.DuplicateVector):
        ix.insert(1, [9.0, 9.0], now=now)
    # HNSW-only mode (src/hybrid/core.rs:264-269)
    ix = fv.HybridIndex(ctx)
    ix.initialize([[0.0, 0.0], [1.0, 1.0]])
    assert not ix.is_ivf_trained()
    ix.insert_with_timestamp(1, [0.0, 0.0], 0.0, 100 * DAY)
    assert ix.recent_count() == 1
    # top-k count semantics (bindings/node/test/test-topk-bug.js:25-83)
    ix = fv.HybridIndex(ctx)
    ix.initialize([[float(i), 1.0] for i in range(10)])
    for i in range(20):
        ix.insert(i, [float(i), 1.0], now=0.0)
    for k, want in ((3, 3), (10, 10), (100, 20)):
        assert ix.search([[0.0, 1.0]], k, now=0.0, hnsw_ef=max(50, k)).counts[0] == want


def test_hybrid_parity_with_oracle(fv, ctx):
    n, d, nlist = 900, 20, 6
    x = mixture(n, d, n_comp=6, seed=55)
    cents = x[:nlist].copy()
    now = 1000 * DAY
    rng = np.random.default_rng(1)
    ages = np.where(rng.random(n) < 0.3, 1 * DAY, 30 * DAY)  # 30 % recent, 70 % historical
    levels = orc.rng_levels(77, n)
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=3)
    g, o = fv.HybridIndex(ctx, **kw), orc.HybridIndex(**kw)
    g.set_ivf_centroids(cents)
    o.set_ivf_centroids(cents)
    for i in range(n):
        g.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
        o.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
    assert (g.recent_count(), g.historical_count()) == (o.recent_count(), o.historical_count())
    q = mixture(48, d, n_comp=6, seed=56)

    def oracle_batch(k, **kws):
        ids = np.full((q.shape[0], k), 2**64 - 1, np.uint64)
        ds = np.full((q.shape[0], k), np.inf, np.float32)
        cnt = np.zeros(q.shape[0], np.uint32)
        for b in range(q.shape[0]):
            r = o.search(q[b], k, **kws)
            cnt[b] = len(r)
            ids[b, : len(r)] = r.ids
            ds[b, : len(r)] = r.distances
        return ids, ds, cnt

    for kws in (dict(now=now), dict(now=now, hnsw_ef=20, ivf_n_probe=2), dict(now=now, search_recent=False),
                dict(now=now, search_historical=False), dict(now=now, recent_k=3, historical_k=4)):
        assert_same_results(g.search(q, 10, **kws), *oracle_batch(10, **kws))
    # deletes route by age (src/hybrid/core.rs:904-937)
    for i in (0, 1, 2, 3, 4, 5):
        g.delete(i, now)
        o.delete(i, now)
    assert_same_results(g.search(q, 10, now=now), *oracle_batch(10, now=now))
    # 10 days later every recent vector has aged: auto-migration copies them into IVF, HNSW keeps
    # them => duplicates in the merged list (no dedup, src/hybrid/core.rs:482-483)
    later = now + 10 * DAY
    rg, (oi, od, oc) = g.search(q, 10, now=later), oracle_batch(10, now=later)
    assert (g.recent_count(), g.historical_count()) == (o.recent_count(), o.historical_count())
    assert_same_results(rg, oi, od, oc)
    assert any(len(set(rg.ids[b, : rg.counts[b]].tolist())) < rg.counts[b] for b in range(len(rg)))


def test_hybrid_auto_migration_on_device_query_path(fv, ctx):
    # the per-search auto-migration (src/hybrid/core.rs:437-439) must also run when the queries are resident in HBM
    # (search_dev, search_dev_begin/_end): counts and the duplicate-bearing merged lists equal the oracle's
    n, d, nlist = 700, 20, 6
    x = mixture(n, d, n_comp=6, seed=57)
    cents = x[:nlist].copy()
    now = 1000 * DAY
    ages = np.where(np.random.default_rng(3).random(n) < 0.3, 1 * DAY, 30 * DAY)
    levels = orc.rng_levels(80, n)
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=3)
    q = mixture(40, d, n_comp=6, seed=58)
    for mode in ("search_dev", "begin_end"):
        g, o = fv.HybridIndex(ctx, **kw), orc.HybridIndex(**kw)
        g.set_ivf_centroids(cents)
        o.set_ivf_centroids(cents)
        for i in range(n):
            g.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
            o.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
        qd = g.ctx.upload(q)
        later = now + 10 * DAY
        if mode == "search_dev":
            rg = g.search_dev(qd, q.shape[0], 10, now=later, dim=d)
        else:
            g.search_dev_begin(2, qd, q.shape[0], 10, now=later, dim=d)
            rg = g.search_dev_end(2)
        oi, od, oc = o.batch_search(q, 10, now=later)
        assert g.recent_count() == o.recent_count() == 0
        assert g.historical_count() == o.historical_count() == n
        assert_same_results(rg, oi, od, oc)
        assert any(len(set(rg.ids[b, : rg.counts[b]].tolist())) < rg.counts[b] for b in range(len(rg)))


def test_device_traversal_layer0_degree_64(fv, ctx):
    # max_connections_layer_0 = 64: a full neighbour list occupies all 64 lanes of the traversal wavefront
    n, d = 1500, 16
    x = mixture(n, d, n_comp=4, sigma=1.0, seed=66)
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(81, n)
    gh, oh = fv.HNSWIndex(ctx, 32, 64, 100, seed=81), orc.HNSWIndex(32, 64, 100, seed=81)
    gh.batch_insert(ids, x, levels)
    oh.batch_insert(ids, x, levels)
    same_graph(gh, oh, ids)
    assert max(len(gh.neighbors(i, 0)) for i in range(n)) == 64
    q = mixture(50, d, n_comp=4, sigma=1.0, seed=67)
    assert gh.device_traversal()
    for k, ef in ((10, 50), (20, 100)):
        dev = gh.search(q, k, ef)
        gh.set_device_traversal(False)
        host = gh.search(q, k, ef)
        gh.set_device_traversal(True)
        assert np.array_equal(dev.ids, host.ids) and np.array_equal(bits(dev.distances), bits(host.distances))
        assert_same_results(dev, *oh.batch_search(q, k, ef))
    assert gh.device_fallbacks() == 0


def test_hybrid_batches_in_flight_match_one_at_a_time(fv, ctx):
    # search_dev_begin / search_dev_end: several batches in flight (graph walks on their own streams, IVF chains
    # queued on one) give, slot by slot, exactly the results of the one-at-a-time search and of the oracle
    n, d, nlist = 4000, 64, 16
    x = mixture(n, d, n_comp=16, seed=91)
    cents = x[:nlist].copy()
    now = 1000 * DAY
    rng = np.random.default_rng(2)
    ages = np.where(rng.random(n) < 0.3, 1 * DAY, 30 * DAY)
    levels = orc.rng_levels(78, n)
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=4)
    g, o = fv.HybridIndex(ctx, **kw), orc.HybridIndex(**kw)
    g.set_ivf_centroids(cents)
    o.set_ivf_centroids(cents)
    for i in range(n):
        g.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
        o.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
    batches = [mixture(64, d, n_comp=16, seed=200 + j) for j in range(7)]
    qdev = [g.ctx.upload(b) for b in batches]
    want = [g.search_dev(qdev[j], 64, 10, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d) for j in range(len(batches))]
    for depth in (2, 3, 4):
        got = [None] * len(batches)
        for j in range(len(batches)):
            if j >= depth:
                got[j - depth] = g.search_dev_end((j - depth) % depth)
            g.search_dev_begin(j % depth, qdev[j], 64, 10, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d)
        for j in range(max(len(batches) - depth, 0), len(batches)):
            got[j] = g.search_dev_end(j % depth)
        for a, b in zip(got, want):
            assert np.array_equal(a.counts, b.counts) and np.array_equal(a.ids, b.ids)
            assert np.array_equal(bits(a.distances), bits(b.distances))
    # and the oracle, on the first batch
    r = want[0]
    for b in range(64):
        ro = o.search(batches[0][b], 10, now=now, hnsw_ef=30, ivf_n_probe=4)
        assert r.counts[b] == len(ro) and np.array_equal(r.ids[b, : len(ro)], ro.ids)
        assert np.array_equal(bits(r.distances[b, : len(ro)]), bits(np.asarray(ro.distances, np.float32)))


def test_hybrid_refuses_mutation_while_a_batch_is_in_flight(fv, ctx):
    # inserts, deletes and migrations move rows (and may grow the list pool) under a scan that is still running:
    # they are refused until every begun batch has been collected, and work again afterwards
    n, d, nlist = 600, 32, 8
    x = mixture(n, d, n_comp=8, seed=95)
    now = 1000 * DAY
    g = fv.HybridIndex(ctx, max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=4)
    g.set_ivf_centroids(x[:nlist].copy())
    for i in range(n):
        g.insert_with_timestamp(i, x[i], now - (1 if i % 3 else 30) * DAY, now)
    q = g.ctx.upload(mixture(32, d, n_comp=8, seed=96))
    want = g.search_dev(q, 32, 5, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d)
    g.search_dev_begin(0, q, 32, 5, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d)
    with pytest.raises(fv.FvdbError):
        g.insert_with_timestamp(n, x[0] + 1, now, now)
    with pytest.raises(fv.FvdbError):
        g.delete(3, now)
    assert g.migrate_with_threshold(0.5 * DAY, now) == 0
    with pytest.raises(fv.FvdbError):  # a later `now` makes a migration due: slot 1 may not begin under slot 0
        g.search_dev_begin(1, q, 32, 5, now=now + 10 * DAY, hnsw_ef=30, ivf_n_probe=4, dim=d)
    g._inflight.pop(1, None)
    rc, hc = g.recent_count(), g.historical_count()
    got = g.search_dev_end(0)
    assert np.array_equal(got.ids, want.ids) and np.array_equal(bits(got.distances), bits(want.distances))
    assert (g.recent_count(), g.historical_count()) == (rc, hc)
    g.insert_with_timestamp(n, x[0] + 1, now, now)
    g.delete(3, now)
    assert g.migrate_with_threshold(0.5 * DAY, now) > 0


def test_blocking_writers_wait_for_the_batches_in_flight(fv, ctx):
    # set_blocking_writers: an insert issued (on another thread) while a batch is uncollected waits for the collector,
    # like the reference's write guard waits for its readers (src/hybrid/core.rs:457,466), and then goes through
    import threading
    import time
    n, d, nlist = 400, 16, 4
    x = mixture(n, d, n_comp=4, seed=97)
    now = 1000 * DAY
    g = fv.HybridIndex(ctx, max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=4)
    g.set_ivf_centroids(x[:nlist].copy())
    for i in range(n):
        g.insert_with_timestamp(i, x[i], now - (1 if i % 3 else 30) * DAY, now)
    g.set_blocking_writers(True)
    q = g.ctx.upload(mixture(16, d, n_comp=4, seed=98))
    want = g.search_dev(q, 16, 5, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d)
    g.search_dev_begin(0, q, 16, 5, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d)
    done = {}

    def writer():
        g.insert_with_timestamp(n, x[0] + 1, now, now)
        done["t"] = time.time()

    th = threading.Thread(target=writer)
    th.start()
    time.sleep(0.3)
    assert th.is_alive() and "t" not in done      # waiting, not refused
    t_end = time.time()
    got = g.search_dev_end(0)                       # the batch is collected: the writer may go
    th.join(10)
    assert not th.is_alive() and done["t"] >= t_end
    assert np.array_equal(got.ids, want.ids)        # the batch saw the index as it was
    assert g.recent_count() + g.historical_count() == n + 1
    g.set_blocking_writers(False)
    g.search_dev_begin(0, q, 16, 5, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d)
    with pytest.raises(fv.FvdbError):               # the default: refused at once
        g.delete(5, now)
    g.search_dev_end(0)


def test_hybrid_vacuum_matches_oracle(fv, ctx):
    # src/hybrid/core.rs:989-1012 -> src/hnsw/operations.rs:176-200 + src/ivf/operations.rs:625-645: soft-deleted
    # vectors leave the graph (and every neighbour set) and their lists; searches, later inserts and the graph itself
    # stay identical to the oracle's
    n, d, nlist = 700, 20, 6
    x = mixture(n + 60, d, n_comp=6, seed=61)
    cents = x[:nlist].copy()
    now = 1000 * DAY
    ages = np.where(np.random.default_rng(6).random(n + 60) < 0.35, 1 * DAY, 30 * DAY)
    levels = orc.rng_levels(79, n + 60)
    kw = dict(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=3)
    g, o = fv.HybridIndex(ctx, **kw), orc.HybridIndex(**kw)
    g.set_ivf_centroids(cents)
    o.set_ivf_centroids(cents)
    for i in range(n):
        g.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
        o.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
    q = mixture(40, d, n_comp=6, seed=62)

    def oracle_batch(k, **kws):
        ids = np.full((q.shape[0], k), 2**64 - 1, np.uint64)
        ds = np.full((q.shape[0], k), np.inf, np.float32)
        cnt = np.zeros(q.shape[0], np.uint32)
        for b in range(q.shape[0]):
            r = o.search(q[b], k, **kws)
            cnt[b] = len(r)
            ids[b, : len(r)] = r.ids
            ds[b, : len(r)] = r.distances
        return ids, ds, cnt

    entry = g.hnsw().entry_point()
    assert entry == o.hnsw().entry_point()
    recent = [i for i in range(n) if ages[i] < 7 * DAY and i != entry]
    hist = [i for i in range(n) if ages[i] >= 7 * DAY]
    dead = recent[5:45:2] + hist[10:100:3]
    for i in dead:
        g.delete(i, now)
        o.delete(i, now)
    assert_same_results(g.search(q, 10, now=now, hnsw_ef=30, ivf_n_probe=4), *oracle_batch(10, now=now, hnsw_ef=30, ivf_n_probe=4))
    before = (g.hnsw().node_count(), g.ivf().total_vectors())
    stats = g.vacuum()
    assert stats == {"hnsw_removed": 20, "ivf_removed": 30, "total_removed": 50}
    assert (o.hnsw().vacuum(), o.ivf().vacuum()) == (20, 30)
    assert (g.hnsw().node_count(), g.ivf().total_vectors()) == (before[0] - 20, before[1] - 30)
    assert g.hnsw().active_count() == g.hnsw().node_count() and g.ivf().active_count() == g.ivf().total_vectors()
    assert g.vacuum()["total_removed"] == 0
    assert_same_results(g.search(q, 10, now=now, hnsw_ef=30, ivf_n_probe=4), *oracle_batch(10, now=now, hnsw_ef=30, ivf_n_probe=4))
    # the graph: same survivors, same links
    ids, lv, off, nb = g.hnsw().export_graph()
    assert set(ids.tolist()) == set(recent + [entry]) - set(dead) and not np.isin(nb, dead).any()
    slot = 0
    for r, l in zip(ids.tolist(), lv.tolist()):
        for layer in range(l + 1):
            assert nb[int(off[slot]):int(off[slot + 1])].tolist() == o.hnsw().neighbors(r, layer)
            slot += 1
    # lists: survivors in their old order
    for c in range(nlist):
        assert g.ivf().export_list(c)[1].tolist() == o.ivf().list_ids(c).tolist()
    # life goes on: new vectors link into the vacuumed graph / lists identically; a vacuumed id is still on record
    with pytest.raises(fv.DuplicateVector):
        g.insert_with_timestamp(dead[0], x[dead[0]], now, now)
    for i in range(n, n + 60):
        g.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
        o.insert_with_timestamp(i, x[i], now - ages[i], now, int(levels[i]))
    assert_same_results(g.search(q, 10, now=now, hnsw_ef=30, ivf_n_probe=4), *oracle_batch(10, now=now, hnsw_ef=30, ivf_n_probe=4))
    r = g.search_dev(g.ctx.upload(q), q.shape[0], 10, now=now, hnsw_ef=30, ivf_n_probe=4, dim=d)
    assert_same_results(r, *oracle_batch(10, now=now, hnsw_ef=30, ivf_n_probe=4))
    # the reference does not repair a vacuumed entry point: its searches fail from then on (src/hnsw/core.rs:418-429)
    h = fv.HNSWIndex(ctx, max_connections=4, max_connections_layer_0=8, ef_construction=20)
    for i in range(30):
        h.insert(i, x[i], int(levels[i]))
    h.mark_deleted(h.entry_point())
    assert h.vacuum() == 1
    with pytest.raises(fv.FvdbError):
        h.search(q[:2], 3, 10)
    with pytest.raises(fv.FvdbError):
        h.insert(1000, x[40], 0)


def test_randomised_hybrid_operation_sequences_match_the_oracle(fv, ctx):
    # tools/hybrid_ops_fuzz.py: inserts at random ages (graph / lists), time moving on (per-search auto-migration),
    # searches with host- and device-resident queries (blocking and begin/end), deletes (with the reference's
    # looked-for-in-the-wrong-part failures), explicit migration — results, counters and failures equal the oracle's
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hybrid_ops_fuzz
    rng = np.random.default_rng(1)
    failed = [c for c in range(24) if hybrid_ops_fuzz.one_case(fv, orc, ctx, rng, c, c if c % 3 == 2 else -2)]
    assert not failed, failed
