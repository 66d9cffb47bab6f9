"""Pins the CPU oracle against every known-answer / property test the reference holds
for the hot path.  Each test cites the reference test it restates
(paths relative to the reference repository)."""
import math

import numpy as np
import pytest

import oracle as orc


@pytest.fixture(scope="module", autouse=True)
def _build():
    orc.build()


# ---- tests/core/vector_ops_advanced.rs ---------------------------------------------------
def test_euclidean_distance_sqrt128():
    # tests/core/vector_ops_advanced.rs:47-58
    a, b = np.zeros(128, np.float32), np.ones(128, np.float32)
    assert abs(orc.euclidean_distance_scalar(a, b) - math.sqrt(128.0)) < 1e-4


def test_dot_product_accuracy_sin_cos():
    # tests/core/vector_ops_advanced.rs:13-33 (scalar vs 8-lane AVX summation agree to 1e-4)
    for size in (16, 64, 128, 256, 512, 1024):
        a = np.sin(np.arange(size, dtype=np.float32)).astype(np.float32)
        b = np.cos(np.arange(size, dtype=np.float32)).astype(np.float32)
        scalar = orc.dot_product_scalar(a, b)
        lanes = (a * b).reshape(-1, 8).sum(axis=0, dtype=np.float32)  # the AVX variant's order
        assert abs(scalar - float(lanes.sum(dtype=np.float32))) < 1e-4


def test_cosine_ones_256():
    # tests/core/vector_ops_advanced.rs:35-45
    a = np.ones(256, np.float32)
    assert abs(orc.cosine_similarity_scalar(a, a) - 1.0) < 1e-6
    assert orc.dot_product_scalar(a, a) == 256.0  # tests/core/vector_ops.rs:73-83


def test_cosine_zero_vector_is_zero():
    # src/core/vector_ops.rs:44-45
    assert orc.cosine_similarity_scalar(np.zeros(4, np.float32), np.ones(4, np.float32)) == 0.0


def test_batch_cosine_similarity():
    # tests/core/vector_ops.rs:12-26
    q = [1.0, 0.0, 0.0]
    sims = [orc.cosine_similarity_scalar(q, v) for v in ([1, 0, 0], [0, 1, 0], [0.707, 0.707, 0])]
    assert abs(sims[0] - 1.0) < 1e-6 and abs(sims[1]) < 1e-6 and abs(sims[2] - 0.707) < 0.01


def test_euclidean_properties():
    # tests/core/vector_ops.rs:115-136 (proptest: symmetry, non-negativity, self-distance)
    rng = np.random.default_rng(7)
    for _ in range(200):
        n = int(rng.integers(10, 100))
        a = rng.uniform(-100, 100, n).astype(np.float32)
        b = rng.uniform(-100, 100, n).astype(np.float32)
        dab, dba = orc.euclidean_distance_scalar(a, b), orc.euclidean_distance_scalar(b, a)
        assert abs(dab - dba) < 1e-6 and dab >= 0.0
        assert abs(orc.euclidean_distance_scalar(a, a)) < 1e-6


def test_l2_is_sequential_f32_fold():
    # src/core/vector_ops.rs:51-57: left-to-right f32 fold, no FMA, sqrt at the end.
    rng = np.random.default_rng(3)
    a = rng.standard_normal(384).astype(np.float32)
    b = rng.standard_normal(384).astype(np.float32)
    s = np.float32(0.0)
    for x, y in zip(a, b):
        t = np.float32(x - y)
        s = np.float32(s + np.float32(t * t))
    assert orc.euclidean_distance_scalar(a, b) == float(np.sqrt(s))


# ---- top-k helpers ---------------------------------------------------------------------
def test_top_k_selection():
    # tests/core/vector_ops.rs:29-35
    assert orc.top_k_indices([0.1, 0.9, 0.5, 0.7, 0.3, 0.8], 3) == [1, 5, 3]


def test_top_k_heap_implementation():
    # tests/core/vector_ops_advanced.rs:86-100
    scores = [0.9, 0.1, 0.7, 0.3, 0.8, 0.2, 0.6, 0.4, 0.5]
    for k in range(1, len(scores) + 1):
        idx = orc.top_k_indices_heap(scores, k)
        assert len(idx) == k
        assert all(scores[idx[i - 1]] >= scores[idx[i]] for i in range(1, k))
        assert sorted(idx) == sorted(orc.top_k_indices(scores, k))
    assert orc.top_k_indices_heap(scores, 0) == []


def test_streaming_top_k_values():
    # tests/core/vector_ops_advanced.rs:102-124 (StreamingTopK == heap top-k on the scores)
    scores = [0.5, 0.9, 0.3, 0.7, 0.8]
    idx = orc.top_k_indices_heap(scores, 3)
    assert [scores[i] for i in idx] == [0.9, 0.8, 0.7]


def test_streaming_top_k_struct():
    # tests/core/vector_ops_advanced.rs:102-124 through StreamingTopK itself (vec_0..vec_4 -> ids 0..4): top 3 scores
    # 0.9, 0.8, 0.7 in that order; k larger than the stream returns everything, sorted
    got = orc.streaming_top_k([0, 1, 2, 3, 4], [0.5, 0.9, 0.3, 0.7, 0.8], 3)
    assert [g[0] for g in got] == [1, 4, 3]
    assert [g[1] for g in got] == pytest.approx([0.9, 0.8, 0.7])
    assert [g[0] for g in orc.streaming_top_k([7, 8], [0.1, 0.2], 5)] == [8, 7]
    assert orc.streaming_top_k([1, 2], [0.1, 0.2], 0) == []


def test_result_merging_dedup_keeps_min():
    # tests/core/vector_ops.rs:37-71   ids: a=1, b=2, c=3
    merged = orc.merge_search_results([[(1, 0.1), (2, 0.3)], [(2, 0.2), (3, 0.4)]], 3)
    assert [m[0] for m in merged] == [1, 2, 3]
    assert [m[1] for m in merged] == pytest.approx([0.1, 0.2, 0.4])


# ---- IVF: tests/ivf/core.rs ----------------------------------------------------------------
TRAIN9 = [[0.0, 0.0], [0.1, 0.1], [0.2, -0.1], [5.0, 5.0], [5.1, 4.9], [4.9, 5.1],
          [-5.0, -5.0], [-4.9, -5.1], [-5.1, -4.9]]


def create_trained_index():
    # tests/ivf/core.rs:441-474
    ix = orc.IVFIndex(n_clusters=3, n_probe=2, train_size=9, max_iterations=10, seed=42)
    ix.train(TRAIN9)
    return ix


def test_train_basic_clusters():
    # tests/ivf/core.rs:69-122: every centroid within 1.0 of an expected centre
    ix = create_trained_index()
    cents = ix.get_centroids()
    for exp in ([0.1, 0.0], [5.0, 5.0], [-5.0, -5.0]):
        assert min(orc.euclidean_distance_scalar(c, exp) for c in cents) < 1.0


def test_train_error_decreases():
    # tests/ivf/core.rs:125-155
    ix = orc.IVFIndex(n_clusters=3, n_probe=2, max_iterations=10, seed=42)
    res = ix.train(TRAIN9)
    assert res["final_error"] <= res["initial_error"]


def test_train_insufficient_and_mismatched():
    # tests/ivf/core.rs:157-218
    ix = orc.IVFIndex(n_clusters=10, n_probe=1, max_iterations=10)
    with pytest.raises(orc.InsufficientTrainingData):
        ix.train([[1.0, 2.0], [3.0, 4.0]])
    ix = orc.IVFIndex(n_clusters=2, n_probe=1, max_iterations=10)
    with pytest.raises(orc.oracle.InconsistentDimensions):
        ix.train([[1.0, 2.0, 3.0], [4.0, 5.0], [6.0, 7.0, 8.0]])


def test_insert_errors():
    # tests/ivf/core.rs:224-308
    ix = orc.IVFIndex()
    with pytest.raises(orc.NotTrained):
        ix.insert(1, [1.0, 2.0])
    ix = create_trained_index()
    ix.insert(7, [1.0, 1.0])
    assert ix.total_vectors() == 1
    assert ix.get_cluster_size(ix.find_cluster([1.0, 1.0])) > 0
    with pytest.raises(orc.DuplicateVector):
        ix.insert(7, [1.0, 1.0])
    with pytest.raises(orc.DimensionMismatch):
        ix.insert(8, [1.0, 2.0, 3.0])


def test_insert_multiple_distribution():
    # tests/ivf/core.rs:248-270
    ix = create_trained_index()
    for i, v in enumerate([[0, 0], [5, 5], [-5, -5], [2.5, 2.5], [-2.5, -2.5]]):
        ix.insert(i, v)
    sizes = [ix.get_cluster_size(c) for c in range(3)]
    assert sum(sizes) == 5 and all(s > 0 for s in sizes)


def test_search_empty_and_sorted_and_short():
    # tests/ivf/core.rs:314-343, :385-398
    ix = create_trained_index()
    assert len(ix.search([1.0, 1.0], 5)) == 0
    for i in range(5):
        ix.insert(i, [0.1 * i, 0.1 * i])
    r = ix.search([0.25, 0.25], 3)
    assert len(r) == 3 and all(r.distances[i - 1] <= r.distances[i] for i in range(1, 3))
    ix2 = create_trained_index()
    for i in range(3):
        ix2.insert(i, [float(i), float(i)])
    assert len(ix2.search([1.5, 1.5], 10)) == 3


def test_search_multi_probe():
    # tests/ivf/core.rs:345-383: nprobe=2 of 3 clusters, query (2.5,2.5) -> 3..4 hits incl. "d"
    ix = create_trained_index()
    for i, v in enumerate([[0, 0], [5, 5], [-5, -5], [2.5, 2.5]]):
        ix.insert(i, v)
    r = ix.search([2.5, 2.5], 4)
    assert 3 <= len(r) <= 4 and 3 in set(r.ids.tolist())


def test_search_exact_match():
    # tests/ivf/core.rs:400-413
    ix = create_trained_index()
    ix.insert(99, [3.14159, 2.71828])
    r = ix.search([3.14159, 2.71828], 1)
    assert list(r.ids) == [99] and r.distances[0] < 1e-6


def test_custom_n_probe_monotone():
    # tests/ivf/core.rs:415-437
    ix = create_trained_index()
    for i in range(20):
        ang = np.float32(i) * np.float32(math.pi) / np.float32(10.0)
        ix.insert(i, [np.cos(ang) * 5.0, np.sin(ang) * 5.0])
    assert len(ix.search([0.0, 0.0], 10, 1)) <= len(ix.search([0.0, 0.0], 10, 3))


def test_ivf_deleted_never_returned_and_vacuum():
    # tests/unit/ivf_deletion_tests.rs:102-128; src/ivf/operations.rs:569-645
    ix = create_trained_index()
    for i in range(10):
        ix.insert(i, [0.1 * i, 0.0])
    ix.mark_deleted(2)
    ix.mark_deleted(3)
    with pytest.raises(orc.VectorNotFound):
        ix.mark_deleted(1234)
    r = ix.search([0.25, 0.0], 10, 3)
    assert 2 not in r.ids and 3 not in r.ids and len(r) == 8
    assert ix.vacuum() == 2 and ix.total_vectors() == 8


def test_ivf_batch_search_sorted():
    # tests/ivf/operations.rs:92-119
    ix = create_trained_index()
    for i in range(12):
        ix.insert(i, [0.5 * i - 3.0, 0.25 * i])
    ids, ds, cnt = ix.batch_search([[0, 0], [5, 5], [-5, -5]], 3)
    for q in range(3):
        assert cnt[q] >= 1 and all(ds[q, i - 1] <= ds[q, i] for i in range(1, cnt[q]))
        single = ix.search([[0, 0], [5, 5], [-5, -5]][q], 3)
        assert list(single.ids) == list(ids[q, : cnt[q]])


# ---- HNSW: tests/hnsw/core.rs ----------------------------------------------------------
def test_level_assignment_statistics():
    # tests/hnsw/core.rs:42-64 (level-0 share > 60 %, ratios 1.5..2.5) — PRNG is the oracle's own
    lv = orc.rng_levels(12345, 10000)
    counts = np.bincount(lv, minlength=10)
    assert counts[0] > 5500
    for i in range(1, 4):
        if counts[i] > 0:
            assert 1.5 < counts[i - 1] / counts[i] < 3.2


def test_hnsw_first_node_and_duplicate():
    # tests/hnsw/core.rs:71-85, :152-166
    ix = orc.HNSWIndex()
    ix.insert(5, [1.0, 2.0, 3.0])
    assert ix.node_count() == 1 and ix.entry_point() == 5
    with pytest.raises(orc.DuplicateVector):
        ix.insert(5, [1.0, 2.0, 3.0])
    with pytest.raises(orc.DimensionMismatch):
        ix.insert(6, [1.0, 2.0])


def test_hnsw_insert_multiple_degree_caps():
    # tests/hnsw/core.rs:87-126
    ix = orc.HNSWIndex(4, 8, 200, seed=42)
    vs = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [0.5, 0.5, 0], [0.5, 0, 0.5]]
    for i, v in enumerate(vs):
        ix.insert(i, v)
    assert ix.node_count() == 5
    for i in range(5):
        assert 0 < len(ix.neighbors(i, 0)) <= 8


def test_hnsw_line_neighbours():
    # tests/hnsw/core.rs:128-150
    ix = orc.HNSWIndex()
    for i in range(5):
        ix.insert(i, [float(i)])
    nb = set(ix.neighbors(2, 0))
    assert 1 in nb and 3 in nb


def test_hnsw_search_empty_single_and_short():
    # tests/hnsw/core.rs:174-197, :300-316
    ix = orc.HNSWIndex()
    assert len(ix.search([1.0, 2.0, 3.0], 5, 200)) == 0
    ix.insert(1, [1.0, 2.0, 3.0])
    r = ix.search([1.0, 2.0, 3.0], 1, 200)
    assert list(r.ids) == [1] and r.distances[0] < 1e-6
    ix = orc.HNSWIndex()
    for i in range(3):
        ix.insert(i, [float(i)])
    assert len(ix.search([1.5], 10, 200)) == 3


def test_hnsw_search_accuracy_self_match():
    # tests/hnsw/core.rs:199-226: 100 vectors sin(i*j), d=10, every vector finds itself (ef=200)
    ix = orc.HNSWIndex(16, 32, 200, seed=42)
    vecs = np.array([[math.sin(float(i * j)) for j in range(10)] for i in range(100)], np.float32)
    for i in range(100):
        ix.insert(i, vecs[i])
    for i in range(100):
        r = ix.search(vecs[i], 1, 200)
        assert len(r) == 1 and r.distances[0] < 1e-5
        # vec_0 is all-zero only for i=0; rows are distinct so ids must match
        assert r.ids[0] == i


def test_hnsw_k_nearest_cross():
    # tests/hnsw/core.rs:228-257
    ix = orc.HNSWIndex()
    for i, v in enumerate([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1], [0.5, 0.5]]):
        ix.insert(i, v)
    r = ix.search([0.1, 0.1], 3, 200)
    assert len(r) == 3 and r.ids[0] == 0


def test_hnsw_ef_parameter_impact():
    # tests/hnsw/core.rs:259-293
    ix = orc.HNSWIndex(16, 32, 200, seed=42)
    for i in range(500):
        ix.insert(i, [math.sin(float(i * j)) for j in range(20)])
    q = [0.5] * 20
    lo, hi = ix.search(q, 10, 50), ix.search(q, 10, 500)
    assert len(lo) == 10 and len(hi) == 10
    assert hi.distances.mean() <= lo.distances.mean() * 1.1


def test_hnsw_multi_layer_structure():
    # tests/hnsw/core.rs:318-348
    ix = orc.HNSWIndex(4, 8, 200, seed=42)
    multi = False
    for i in range(100):
        ix.insert(i, [float(i)] * 10)
        if ix.level(i) > 0:
            multi = True
            assert len(ix.neighbors(i, 1)) <= len(ix.neighbors(i, 0)) or len(ix.neighbors(i, 1)) <= 4
            assert len(ix.neighbors(i, 1)) <= 4
    assert multi
    for i in range(100):
        assert len(ix.neighbors(i, 0)) <= 8


def test_hnsw_deleted_skipped():
    # src/hnsw/core.rs:510-513, :451-461; src/hnsw/operations.rs:127-137
    ix = orc.HNSWIndex(16, 32, 200, seed=1)
    for i in range(50):
        ix.insert(i, [float(i), 0.0])
    ix.mark_deleted(10)
    r = ix.search([10.0, 0.0], 5, 50)
    assert 10 not in r.ids and len(r) == 5
    with pytest.raises(orc.VectorNotFound):
        ix.mark_deleted(999)


# ---- Hybrid: tests/hybrid/core.rs -------------------------------------------------------
def create_training_data():
    return [[float(i), float(i) * 0.5] for i in range(10)]


DAY = 86400.0


def test_hybrid_search_empty():
    # tests/hybrid/core.rs:167-174
    ix = orc.HybridIndex()
    assert len(ix.search([1.0, 2.0], 5)) == 0


def test_hybrid_recent_only_sorted():
    # tests/hybrid/core.rs:176-203
    ix = orc.HybridIndex()
    ix.initialize(create_training_data())
    now = 100 * DAY
    for i in range(5):
        ix.insert(i, [float(i), 0.0], now=now)
    r = ix.search([2.5, 0.0], 3, now=now)
    assert len(r) == 3 and r.distances[0] <= r.distances[1] <= r.distances[2]
    assert ix.recent_count() == 5 and ix.historical_count() == 0


def test_hybrid_historical_only():
    # tests/hybrid/core.rs:205-239
    ix = orc.HybridIndex()
    ix.initialize(create_training_data())
    now = 100 * DAY
    for i in range(5):
        ix.insert_with_timestamp(100 + i, [0.0, float(i)], now - 30 * DAY, now)
    assert ix.recent_count() == 0 and ix.historical_count() == 5
    r = ix.search([0.0, 2.5], 3, now=now)
    assert len(r) == 3 and all(100 <= i < 105 for i in r.ids)


def test_hybrid_mixed():
    # tests/hybrid/core.rs:241-288
    ix = orc.HybridIndex()
    ix.initialize(create_training_data())
    now = 100 * DAY
    for i in range(3):
        ix.insert(i, [float(i), float(i)], now=now)
    for i in range(3, 6):
        ix.insert_with_timestamp(i, [float(i), float(i)], now - 30 * DAY, now)
    r = ix.search([2.5, 2.5], 6, now=now)
    assert len(r) == 6
    rec = sum(1 for i in r.ids if i < 3)
    assert rec > 0 and 6 - rec > 0


def test_hybrid_hnsw_only_mode_and_uninitialised_insert():
    # src/hybrid/core.rs:262-269 (fewer than min_ivf_training_size), :363-365
    ix = orc.HybridIndex()
    with pytest.raises(orc.NotInitialized):
        ix.insert(1, [0.0, 0.0])
    ix.initialize([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0]])
    assert not ix.is_ivf_trained()
    ix.insert_with_timestamp(1, [0.0, 0.0], 0.0, 100 * DAY)  # old, but IVF untrained => HNSW
    assert ix.recent_count() == 1
    with pytest.raises(orc.DuplicateVector):
        ix.insert(1, [0.0, 0.0])


def test_hybrid_auto_migrate_copies_and_duplicates():
    # src/hybrid/core.rs:437-439, :600-649: migration copies into IVF and never removes from HNSW
    ix = orc.HybridIndex()
    ix.initialize(create_training_data())
    t0 = 100 * DAY
    for i in range(4):
        ix.insert(i, [float(i), 0.0], now=t0)
    r = ix.search([0.0, 0.0], 8, now=t0 + 8 * DAY)  # all four are now older than 7 days
    assert ix.historical_count() == 4
    assert len(r) == 8  # each id twice: once from HNSW, once from IVF (no dedup)
    assert sorted(r.ids.tolist()) == [0, 0, 1, 1, 2, 2, 3, 3]


def test_topk_count_semantics():
    # bindings/node/test/test-topk-bug.js:25-83: k=3/10/100 on 20 vectors -> 3/10/20 hits
    ix = orc.HybridIndex()
    ix.initialize([[float(i), 1.0] for i in range(10)])
    for i in range(20):
        ix.insert(i, [float(i), 1.0], now=0.0)
    for k, want in ((3, 3), (10, 10), (100, 20)):
        assert len(ix.search([0.0, 1.0], k, now=0.0, hnsw_ef=max(50, k))) == want
