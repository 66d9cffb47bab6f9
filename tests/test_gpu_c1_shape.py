"""BASELINE config C1 at its own shape: 10K x 384-d f32, HNSW M 16 / M0 32 / ef_construction 200 built by SEQUENTIAL
inserts (the reference's real build, src/hnsw/core.rs:226-378), searched with ef 50, k 10 — on (a) SURVEY §8d's
Gaussian mixture and (b) the reference bench's own generator (benches/chunked_search_bench.rs:30-41: 1000 distinct
vectors, ten exact copies of each — every distance ties).  The graph must be the oracle's node for node, and the
searches bit-identical to the oracle's in both traversal modes (whole walk on the GPU / host walk with per-hop GPU
scoring)."""
import time

import numpy as np
import pytest

import fvdb_import
import oracle as orc
from _data import bits, mixture

pytestmark = pytest.mark.gpu
N, D, NQ = 10_000, 384, 1000


@pytest.fixture(scope="module")
def fv():
    return fvdb_import.load()


@pytest.fixture(scope="module")
def ctx(fv):
    orc.build()
    c = fv.Context(0)
    yield c
    c.close()


def reference_bench_vectors(count, dims, seed):
    # create_test_vectors (benches/chunked_search_bench.rs:30-41), f32 arithmetic as written there
    i = (np.arange(count, dtype=np.int64) + seed).astype(np.float32)
    base = np.fmod(i * np.float32(0.001), np.float32(1.0)).astype(np.float32)
    ramp = (np.arange(dims, dtype=np.float32) * np.float32(0.0001)).astype(np.float32)
    return (base[:, None] + ramp[None, :]).astype(np.float32)


def survey_mixture(n, seed):
    # 4096 component means ~ N(0, I_d), row = mean + 0.35 N(0, I_d)
    return mixture(n, D, n_comp=4096, sigma=0.35, seed=seed)


@pytest.mark.parametrize("generator", ["survey_mixture", "reference_bench"])
def test_c1_sequential_insert_graph_and_searches_match_oracle(fv, ctx, generator):
    if generator == "survey_mixture":
        x, q = survey_mixture(N, 1234), survey_mixture(NQ, 5678)
    else:
        x, q = reference_bench_vectors(N, D, 42), reference_bench_vectors(NQ, D, 7)
    ids = np.arange(N, dtype=np.uint64)
    levels = orc.rng_levels(42, N)
    t0 = time.time()
    oh = orc.HNSWIndex(16, 32, 200, seed=42)
    oh.batch_insert(ids, x, levels)
    t1 = time.time()
    gh = fv.HNSWIndex(ctx, 16, 32, 200, seed=42)
    gh.batch_insert(ids, x, levels)
    t2 = time.time()
    print(f"[c1 {generator}] oracle build {t1 - t0:.1f}s, device build {t2 - t1:.1f}s")
    assert gh.node_count() == N and gh.entry_point() == oh.entry_point()
    gi, lv, off, nb = gh.export_graph()
    slot = 0
    for r, l in zip(gi.tolist(), lv.tolist()):
        assert l == oh.level(r)
        for layer in range(l + 1):
            assert nb[int(off[slot]):int(off[slot + 1])].tolist() == oh.neighbors(r, layer), (r, layer)
            slot += 1
    want = oh.batch_search(q, 10, 50)
    for device in (True, False):
        gh.set_device_traversal(device)
        got = gh.search(q, 10, 50)
        assert np.array_equal(got.counts, want[2]) and np.array_equal(got.ids, want[0]), device
        assert np.array_equal(bits(got.distances), bits(want[1])), device
    gh.set_device_traversal(True)
    # every query finishes on the device on both generators.  The reference bench's data (ten exact copies of every
    # vector: equal distances everywhere) sends every query through the restated-heap search and makes the walks long;
    # a query that outgrows its visited log now clears its whole map at the end of the layer instead of leaving the device
    assert gh.device_fallbacks() == 0
    st = gh.insert_stats()
    assert st["host_path_inserts"] == 0 and st["n_done"] == N
    print(f"[c1 {generator}] device insert: {st}")
    # build time against the CPU oracle's (single thread, the reference's own cost model).  On the mixture the device
    # build takes well under half the oracle's time.  The reference bench's generator walks a line — consecutive inserts are each other's
    # nearest neighbours, so no two inserts of a batch are independent, and every search ties and takes the restated
    # heaps: that data is built one insert at a time at the latency of ONE workgroup, somewhat behind a CPU core.
    # (measured: 2.0 s against 7.3 s on the mixture, 8.5 s against 6.2 s on the reference bench's generator)
    assert t2 - t1 <= (0.5 if generator == "survey_mixture" else 1.6) * (t1 - t0) + 1.0, (t2 - t1, t1 - t0)
    if generator == "survey_mixture":
        # self-match (tests/hnsw/core.rs:199-226) at this shape: the reference's nearest-M neighbour selection (no
        # diversity heuristic) leaves well-separated components poorly connected, so ef = 50 finds ~70 % of the stored
        # vectors themselves on this data — the oracle, i.e. the reference's algorithm, behaves identically
        r = gh.search(x[:200], 1, 50)
        ro = oh.batch_search(x[:200], 1, 50)
        assert np.array_equal(r.ids, ro[0]) and np.mean(r.ids[:, 0] == ids[:200]) > 0.5
