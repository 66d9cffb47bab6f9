"""GPU tests of the reference's top-k / merge helpers against the oracle, of concurrent searches on
one index from several host threads (reference: searches hold a read guard, bindings/node/src/session.rs:253), and
of HybridIndex::search_with_filter in the C++ host mirror (src/hybrid/core.rs:513-549)."""
import threading

import numpy as np
import pytest

import fvdb_import
import oracle as orc
from _data import bits, mixture

pytestmark = pytest.mark.gpu
DAY = 86400.0


@pytest.fixture(scope="module")
def fv():
    return fvdb_import.load()


@pytest.fixture(scope="module")
def ctx(fv):
    orc.build()
    c = fv.Context(0)
    yield c
    c.close()


# ---- a4: top_k_indices / top_k_indices_heap / StreamingTopK / merge_search_results ---------------------------------
def test_top_k_known_answers(fv, ctx):
    # tests/core/vector_ops.rs:29-35
    assert fv.top_k_indices(ctx, [0.1, 0.9, 0.5, 0.7, 0.3, 0.8], 3) == [1, 5, 3]
    # tests/core/vector_ops_advanced.rs:86-100
    scores = [0.9, 0.1, 0.7, 0.3, 0.8, 0.2, 0.6, 0.4, 0.5]
    for k in range(1, len(scores) + 1):
        idx = fv.top_k_indices_heap(ctx, scores, k)
        assert len(idx) == k and all(scores[idx[i - 1]] >= scores[idx[i]] for i in range(1, k))
        assert sorted(idx) == sorted(fv.top_k_indices(ctx, scores, k))
    assert fv.top_k_indices_heap(ctx, scores, 0) == [] and fv.top_k_indices(ctx, scores, 0) == []
    # tests/core/vector_ops_advanced.rs:102-124
    got = fv.streaming_top_k(ctx, [0, 1, 2, 3, 4], [0.5, 0.9, 0.3, 0.7, 0.8], 3)
    assert [g[0] for g in got] == [1, 4, 3] and [g[1] for g in got] == pytest.approx([0.9, 0.8, 0.7])
    # tests/core/vector_ops.rs:37-71
    merged = fv.merge_search_results(ctx, [[(1, 0.1), (2, 0.3)], [(2, 0.2), (3, 0.4)]], 3)
    assert [m[0] for m in merged] == [1, 2, 3] and [m[1] for m in merged] == pytest.approx([0.1, 0.2, 0.4])
    with pytest.raises(fv.NonFiniteInput):
        fv.top_k_indices(ctx, [0.1, np.nan], 1)


@pytest.mark.parametrize("n,k", [(1, 1), (5, 10), (64, 64), (65, 7), (1000, 10), (5000, 100), (3000, 256), (300, 200)])
def test_top_k_matches_oracle_with_ties(fv, ctx, n, k):
    rng = np.random.default_rng(n * 31 + k)
    B = 6
    s = rng.standard_normal((B, n)).astype(np.float32)
    s[1] = np.round(s[1] * 2) / 2          # many exact ties: the heap's eviction order matters
    s[2] = 0.25                            # all equal
    s[3, ::3] = -0.0
    s[3, 1::3] = 0.0                       # -0.0 and +0.0 compare equal
    s[4] = np.sort(s[4])                   # ascending: every element enters the heap
    s[5] = np.sort(s[5])[::-1]             # descending: nothing after the first k enters
    got_sort = fv.top_k_indices(ctx, s, k)
    got_heap = fv.top_k_indices_heap(ctx, s, k)
    for b in range(B):
        assert got_sort[b] == orc.top_k_indices(s[b], k), b
        assert got_heap[b] == orc.top_k_indices_heap(s[b], k), b


@pytest.mark.parametrize("n,k", [(7, 3), (200, 16), (2000, 64), (900, 256)])
def test_streaming_top_k_matches_oracle(fv, ctx, n, k):
    rng = np.random.default_rng(n + k)
    B = 4
    ids = rng.permutation(10 * n)[: B * n].reshape(B, n).astype(np.uint64)
    s = rng.standard_normal((B, n)).astype(np.float32)
    s[1] = np.round(s[1] * 2) / 2  # ties: the tuple order (reversed score, then id) decides who leaves the heap
    s[2] = 1.5
    got = fv.streaming_top_k(ctx, ids, s, k)
    for b in range(B):
        want = orc.streaming_top_k(ids[b], s[b], k)
        assert [g[0] for g in got[b]] == [w[0] for w in want], b
        assert np.array_equal(bits(np.asarray([g[1] for g in got[b]], np.float32)),
                              bits(np.asarray([w[1] for w in want], np.float32)))


@pytest.mark.parametrize("sets,per,k", [(2, 10, 10), (8, 10, 10), (8, 32, 100), (3, 200, 256), (16, 4, 5)])
def test_merge_search_results_matches_oracle(fv, ctx, sets, per, k):
    rng = np.random.default_rng(sets * 100 + per)
    for trial in range(4):
        pool_ids = rng.integers(0, per * sets // 2 + 3, size=(sets, per)).astype(np.uint64)  # heavy duplication
        d = np.abs(rng.standard_normal((sets, per))).astype(np.float32)
        if trial == 1:
            d = np.round(d * 4) / 4  # equal distances between different ids and between copies of one id
        if trial == 2:
            d[:] = 0.5
        rs = [[(int(pool_ids[g, i]), float(d[g, i])) for i in range(per)] for g in range(sets)]
        got = fv.merge_search_results(ctx, rs, k)
        want = orc.merge_search_results(rs, k)
        assert [g[0] for g in got] == [w[0] for w in want], trial
        assert [np.float32(g[1]) for g in got] == [np.float32(w[1]) for w in want], trial


# ---- concurrent readers ---------------------------------------------------------------------------------------------
def _build_hybrid(fv, ctx, n=3000, d=48, nlist=12, seed=71):
    x = mixture(n, d, n_comp=12, seed=seed)
    cents = x[:nlist].copy()
    now = 1000 * DAY
    ages = np.where(np.random.default_rng(seed).random(n) < 0.3, 1 * DAY, 30 * DAY)
    g = fv.HybridIndex(ctx, max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=4)
    g.set_ivf_centroids(cents)
    g.bulk_insert(np.arange(n, dtype=np.uint64), x, now - ages, now)
    return g, x, now


def test_concurrent_searches_on_one_hybrid_index(fv, ctx):
    # 4 host threads x 50 searches on ONE index: every result equals the single-threaded one
    g, x, now = _build_hybrid(fv, ctx)
    d = x.shape[1]
    batches = [mixture(24, d, n_comp=12, seed=300 + j) for j in range(10)]
    want = [g.search(b, 10, now=now, hnsw_ef=40, ivf_n_probe=4) for b in batches]
    errors = []

    def worker(t):
        try:
            for it in range(50):
                j = (t * 7 + it) % len(batches)
                got = g.search(batches[j], 10, now=now, hnsw_ef=40, ivf_n_probe=4)
                w = want[j]
                assert np.array_equal(got.counts, w.counts) and np.array_equal(got.ids, w.ids), (t, it)
                assert np.array_equal(bits(got.distances), bits(w.distances)), (t, it)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
    # more threads than slots: callers wait for a free one instead of colliding
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]


def test_concurrent_searches_with_a_writer(fv, ctx):
    # readers keep searching while another thread inserts: inserts wait for the searches in flight (write guard),
    # nothing crashes, and once the writer is done every reader sees the final index
    g, x, now = _build_hybrid(fv, ctx, n=1500, seed=72)
    d = x.shape[1]
    q = mixture(16, d, n_comp=12, seed=400)
    extra = mixture(40, d, n_comp=12, seed=401)
    errors, stop = [], threading.Event()

    def reader():
        try:
            while not stop.is_set():
                r = g.search(q, 5, now=now, hnsw_ef=30, ivf_n_probe=4)
                assert np.all(r.counts == 5) and np.all(np.diff(r.distances, axis=1) >= 0)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    readers = [threading.Thread(target=reader) for _ in range(3)]
    for th in readers:
        th.start()
    for i in range(extra.shape[0]):
        g.insert_with_timestamp(10_000 + i, extra[i], now - (1 if i % 2 else 30) * DAY, now)
    stop.set()
    for th in readers:
        th.join()
    assert not errors, errors[0]
    # the rows that went to the inverted lists (even i: 30 days old) are found exactly when every list is probed
    r = g.search(extra[0:16:2], 1, now=now, hnsw_ef=30, ivf_n_probe=12)
    assert np.array_equal(r.ids[:, 0], np.arange(10_000, 10_016, 2, dtype=np.uint64)) and np.all(r.distances[:, 0] < 1e-6)
    assert g.recent_count() + g.historical_count() == 1500 + extra.shape[0]


def test_concurrent_ivf_searches_through_the_c_abi(fv, ctx):
    # fvdb_ivf_search from several threads on one fvdb_ivf handle: each call leases a scratch set and a stream
    n, d, nlist = 6000, 64, 32
    x = mixture(n, d, n_comp=32, seed=73)
    ivf = fv.DeviceIVF(ctx, d, nlist)
    ivf.set_centroids(x[:nlist].copy())
    ivf.add(x, np.arange(n, dtype=np.uint64))
    batches = [mixture(40, d, n_comp=32, seed=500 + j) for j in range(6)]
    want = [ivf.search(b, 10, 6) for b in batches]
    errors = []

    def worker(t):
        try:
            for it in range(40):
                j = (t + it) % len(batches)
                ids, ds, cnt = ivf.search(batches[j], 10, 6)
                assert np.array_equal(ids, want[j][0]) and np.array_equal(bits(ds), bits(want[j][1])), (t, it)
                assert np.array_equal(cnt, want[j][2])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(10)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]


# ---- search_with_filter below Python --------------------------------------------------------------------------------
def test_search_with_filter_in_the_host_mirror(fv, ctx):
    g, x, now = _build_hybrid(fv, ctx, n=2000, seed=74)
    d = x.shape[1]
    q = mixture(20, d, n_comp=12, seed=600)
    k = 8
    allowed = set(range(0, 2000, 3))  # ids with matching metadata
    got = g.search_with_filter(q, k, lambda rid: rid in allowed, now=now)
    plain = g.search(q, 3 * k, now=now)  # SearchConfig::default with k = 3 k (src/hybrid/core.rs:419-423, 527-531)
    for b in range(q.shape[0]):
        want = [(int(i), dd) for i, dd in zip(plain.ids[b, : plain.counts[b]], plain.distances[b, : plain.counts[b]])
                if int(i) in allowed][:k]
        assert got.counts[b] == len(want)
        assert got.ids[b, : len(want)].tolist() == [w[0] for w in want]
        assert np.array_equal(bits(got.distances[b, : len(want)]), bits(np.asarray([w[1] for w in want], np.float32)))
    # no filter = plain search; a filter nothing passes = empty lists
    none = g.search_with_filter(q, k, None, now=now)
    p2 = g.search(q, k, now=now)
    assert np.array_equal(none.ids, p2.ids) and np.array_equal(bits(none.distances), bits(p2.distances))
    assert np.all(g.search_with_filter(q, k, lambda rid: False, now=now).counts == 0)
    with pytest.raises(ZeroDivisionError):  # an exception inside the predicate surfaces after the call returns
        g.search_with_filter(q[:1], k, lambda rid: 1 // 0, now=now)


def test_ctx_info_reports_the_hardware_queue_setting(fv, ctx):
    # fvdb_ctx_create asks for 16 hardware queues (several batches in flight) unless the host chose a value; the context
    # says which case applies, so a host that initialised HIP first can see that the request came too late
    info = ctx.info()
    assert info["device"] == 0 and info["compute_units"] >= 64
    assert info["hw_queues_source"] in (1, 2, 3)
    if info["hw_queues_source"] == 2:
        assert info["hw_queues"] == 16
    if info["hw_queues_source"] == 3:
        assert info["hw_queues"] == 4
