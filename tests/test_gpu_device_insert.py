"""HNSWIndex::insert resident on the GPU (fvdb_graph_insert_linked, kernels_graph_build.h; reference
src/hnsw/core.rs:226-378): the graph must be the CPU oracle's node for node whichever way the inserts run — one at a
time on the device, speculated in batches and committed in order, or by the host algorithm with per-hop GPU scoring —
and the incremental device mirror must move O(M x levels) bytes per insert, never the whole graph."""
import os
import sys

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


def same_graph(gh, oh):
    assert gh.entry_point() == oh.entry_point()
    gi, lv, off, nb = gh.export_graph()
    slot = 0
    for r, l in zip(gi.tolist(), lv.tolist()):
        assert l == oh.level(r)
        for layer in range(l + 1):
            assert nb[int(off[slot]):int(off[slot + 1])].tolist() == oh.neighbors(r, layer), (r, layer)
            slot += 1


def same_results(got, want):
    assert np.array_equal(got.counts, want[2]) and np.array_equal(got.ids, want[0])
    assert np.array_equal(bits(got.distances), bits(want[1]))


CASES = [  # n, d, M, M0, efc, seed
    (400, 16, 6, 12, 40, 1),
    (700, 100, 8, 16, 64, 2),     # d not a multiple of 4 x 32: padded rows, bounds-checked loads
    (500, 384, 16, 32, 200, 3),   # BASELINE shape (C1 / C3 parameters)
    (300, 768, 16, 32, 200, 4),   # C5's dimension
    (350, 40, 4, 8, 300, 5),      # ef_construction above the sorted-register form: restated heaps only
]


@pytest.mark.parametrize("n,d,M,M0,efc,seed", CASES)
@pytest.mark.parametrize("mode", [1, 2])
def test_device_insert_builds_the_oracles_graph(fv, ctx, n, d, M, M0, efc, seed, mode):
    x = mixture(n, d, n_comp=8, seed=seed)
    ids = np.arange(n, dtype=np.uint64) + 7
    levels = orc.rng_levels(seed, n)
    gh, oh = fv.HNSWIndex(ctx, M, M0, efc, seed=seed), orc.HNSWIndex(M, M0, efc, seed=seed)
    gh.set_device_insert(True, mode)
    ok, bad = gh.batch_insert(ids, x, levels)
    assert (ok, bad) == (n, 0)
    oh.batch_insert(ids, x, levels)
    st = gh.insert_stats()
    assert st["host_path_inserts"] == 0 and st["n_done"] == n
    if mode == 2:
        assert st["speculated_ok"] > 0          # some speculated searches were adopted ...
        assert st["commit_stops"] > 0           # ... and some were invalidated by an earlier insert of their batch
    same_graph(gh, oh)
    q = mixture(25, d, n_comp=8, seed=seed + 100)
    for device in (True, False):
        gh.set_device_traversal(device)
        same_results(gh.search(q, 10, 50), oh.batch_search(q, 10, 50))


@pytest.mark.parametrize("n,d,M,M0,efc,n_comp,seed", [
    (6000, 12, 6, 12, 40, 16, 81),      # low dimension: the inserts of a batch land in each other's neighbourhoods all the time
    (4000, 384, 16, 32, 200, 4096, 82),  # near-isotropic rows (SURVEY §8d's mixture): every node is "about as far" as any other
    (5000, 24, 8, 16, 300, 8, 83),      # ef_construction above the register set: restated heaps, order-dependent searches
])
def test_speculated_batches_adopt_only_what_the_reference_would_compute(fv, ctx, n, d, M, M0, efc, n_comp, seed):
    # speculation from the first node on (mode 2), long batches: a speculated search is adopted although rows it expanded
    # were changed by earlier inserts of its batch when every node added to / dropped from those rows is provably
    # irrelevant to it (validate_speculation, kernels_graph_build.h) — the graph must still be the oracle's node for node
    x = mixture(n, d, n_comp=n_comp, seed=seed)
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(seed, n)
    gh, oh = fv.HNSWIndex(ctx, M, M0, efc, seed=seed), orc.HNSWIndex(M, M0, efc, seed=seed)
    gh.set_device_insert(True, 2)
    assert gh.batch_insert(ids, x, levels) == (n, 0)
    oh.batch_insert(ids, x, levels)
    st = gh.insert_stats()
    # (a restated `candidates` heap that outgrows its LDS slots sends that one insert through the host algorithm)
    assert st["host_path_inserts"] <= 4 and st["speculated_ok"] > n // 4
    print(f"[speculation] n {n} d {d} ef {efc}: {st['speculated_ok']} adopted, {st['commit_stops']} stops, "
          f"{st['launches'] // 2} batches")
    same_graph(gh, oh)


def test_device_insert_equals_host_algorithm_and_continues_after_it(fv, ctx):
    # half the nodes by the host algorithm (per-hop GPU scoring), the rest on the device, then one more host insert:
    # each switch hands the graph across (whole install once, then row patches / pulls), the result is the oracle's
    n, d = 600, 32
    x = mixture(n, d, n_comp=5, seed=11)
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(11, n)
    gh, oh = fv.HNSWIndex(ctx, 8, 16, 80, seed=11), orc.HNSWIndex(8, 16, 80, seed=11)
    oh.batch_insert(ids, x, levels)
    gh.set_device_insert(False)
    gh.batch_insert(ids[:250], x[:250], levels[:250])
    gh.set_device_insert(True, 1)
    gh.batch_insert(ids[250:599], x[250:599], levels[250:599])
    gh.set_device_insert(False)
    gh.insert(int(ids[599]), x[599], int(levels[599]))
    st = gh.insert_stats()
    assert st["host_path_inserts"] == 251 and st["n_done"] == 349
    same_graph(gh, oh)


def test_device_insert_own_level_draws_and_failed_inserts(fv, ctx):
    # levels drawn from SplitMix64 in insert order; a duplicate id, a wrong dimension and a NaN fail like the reference's
    # checks do (no level drawn, nothing changed) and the batch goes on (src/hnsw/operations.rs:74-94)
    n, d = 200, 24
    x = mixture(n, d, n_comp=4, seed=21)
    ids = np.arange(n, dtype=np.uint64)
    ids[50] = ids[10]                # duplicate of an earlier id in the same call
    x[120, 3] = np.nan
    gh, oh = fv.HNSWIndex(ctx, 5, 10, 50, seed=9), orc.HNSWIndex(5, 10, 50, seed=9)
    ok, bad = gh.batch_insert(ids, x)
    assert (ok, bad) == (n - 2, 2)
    for i in range(n):
        if i in (50, 120):
            continue
        oh.insert(int(ids[i]), x[i])
    same_graph(gh, oh)
    with pytest.raises(Exception):
        gh.insert(999, np.zeros(d + 1, np.float32))
    with pytest.raises(Exception):
        gh.insert(int(ids[0]), x[0])


def test_device_insert_duplicate_vectors_tie_like_the_reference(fv, ctx):
    # ten exact copies of every vector (the reference bench's generator): every admission ties, the sorted-register
    # search restarts with the restated heaps each time
    n, d = 600, 64
    base = mixture(60, d, n_comp=3, seed=31)
    x = np.tile(base, (10, 1))
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(31, n)
    gh, oh = fv.HNSWIndex(ctx, 6, 12, 48, seed=31), orc.HNSWIndex(6, 12, 48, seed=31)
    gh.set_device_insert(True, 1)
    gh.batch_insert(ids, x, levels)
    oh.batch_insert(ids, x, levels)
    assert gh.insert_stats()["tie_restarts"] > 0
    same_graph(gh, oh)


def test_device_insert_skips_deleted_nodes_and_handles_tall_nodes(fv, ctx):
    n, d = 500, 20
    x = mixture(n, d, n_comp=4, seed=41)
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(41, n).copy()
    levels[300] = 17                 # above the layers one workgroup keeps on chip: that node takes the host algorithm
    gh, oh = fv.HNSWIndex(ctx, 6, 12, 40, seed=41), orc.HNSWIndex(6, 12, 40, seed=41)
    gh.set_device_insert(True, 1)
    gh.batch_insert(ids[:200], x[:200], levels[:200])
    oh.batch_insert(ids[:200], x[:200], levels[:200])
    for i in range(0, 200, 7):       # soft-deleted nodes are visited, never scored, never linked (:511-513)
        gh.mark_deleted(int(ids[i]))
        oh.mark_deleted(int(ids[i]))
    gh.batch_insert(ids[200:], x[200:], levels[200:])
    oh.batch_insert(ids[200:], x[200:], levels[200:])
    assert gh.insert_stats()["host_path_inserts"] == 1
    same_graph(gh, oh)
    q = mixture(20, d, n_comp=4, seed=42)
    same_results(gh.search(q, 10, 50), oh.batch_search(q, 10, 50))


def test_graph_too_large_for_the_chip_takes_the_host_algorithm(fv, ctx, monkeypatch):
    # the device insert keeps the visited bitmap of ALL nodes in LDS; a graph beyond that (about 800K nodes) is linked by
    # the host algorithm, row patches keep the device copy current, searches stay on the device.  The limit is lowered
    # through a test hook to get there with a small graph.
    n, d = 300, 16
    x = mixture(n, d, n_comp=3, seed=71)
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(71, n)
    gh, oh = fv.HNSWIndex(ctx, 6, 12, 40, seed=71), orc.HNSWIndex(6, 12, 40, seed=71)
    gh.batch_insert(ids[:200], x[:200], levels[:200])
    monkeypatch.setenv("FVDB_BUILD_LDS_LIMIT", "4096")
    gh.batch_insert(ids[200:], x[200:], levels[200:])
    monkeypatch.delenv("FVDB_BUILD_LDS_LIMIT")
    oh.batch_insert(ids, x, levels)
    st = gh.insert_stats()
    assert st["host_path_inserts"] == 100 and st["n_done"] == 200
    same_graph(gh, oh)
    q = mixture(20, d, n_comp=3, seed=72)
    same_results(gh.search(q, 5, 30), oh.batch_search(q, 5, 30))


def test_wide_lists_take_the_host_algorithm(fv, ctx):
    n, d = 150, 16
    x = mixture(n, d, n_comp=3, seed=51)
    ids = np.arange(n, dtype=np.uint64)
    levels = orc.rng_levels(51, n)
    gh, oh = fv.HNSWIndex(ctx, 16, 64, 80, seed=51), orc.HNSWIndex(16, 64, 80, seed=51)
    gh.batch_insert(ids, x, levels)  # M0 = 64 > 63: not served by the device insert
    oh.batch_insert(ids, x, levels)
    assert gh.insert_stats()["host_path_inserts"] == n
    same_graph(gh, oh)


def test_interleaved_inserts_and_searches_move_rows_not_the_graph(fv, ctx):
    # the reference's live pattern (bindings/node/src/session.rs:203,340): addVectors interleaved with search.  Per
    # iteration the host -> device graph traffic is the new node's level words (and, on the host path, the rows it
    # changed) — never the adjacency of the whole graph.
    n0, d, it = 20000, 64, 300
    x = mixture(n0 + 2 * it, d, n_comp=64, seed=61)
    ids = np.arange(x.shape[0], dtype=np.uint64)
    levels = orc.rng_levels(61, x.shape[0])
    gh = fv.HNSWIndex(ctx, 8, 16, 60, seed=61)
    gh.set_device_insert(True, 0)
    gh.batch_insert(ids[:n0], x[:n0], levels[:n0])
    oh = orc.HNSWIndex(8, 16, 60, seed=61)
    gi, lv, off, nb = gh.export_graph()
    oh.restore(gi, x[:n0], lv, off, nb, gh.entry_point())
    q = mixture(8 * it, d, n_comp=64, seed=62)
    base = gh.insert_stats()["graph_upload_bytes"]
    whole_graph = n0 * 17 * 4
    for mode, lo in ((True, n0), (False, n0 + it)):  # device inserts, then host-algorithm inserts (row patches)
        gh.set_device_insert(mode, 1)
        for i in range(lo, lo + it):
            gh.insert(int(ids[i]), x[i], int(levels[i]))
            oh.insert(int(ids[i]), x[i], int(levels[i]))
            qq = q[8 * (i - n0):8 * (i - n0) + 8]
            same_results(gh.search(qq, 10, 50), oh.batch_search(qq, 10, 50))
        moved = gh.insert_stats()["graph_upload_bytes"] - base
        base += moved
        per_insert = moved / it
        print(f"[interleaved] {'device' if mode else 'host-algorithm'} inserts: {per_insert:.0f} graph bytes per insert "
              f"(whole layer-0 adjacency: {whole_graph} bytes)")
        assert per_insert <= (8 if mode else 4 * (2 + 64 + 2) * (17 + 4 * 9)) + 64
    assert gh.device_fallbacks() == 0
    same_graph(gh, oh)


def test_randomised_shapes_build_the_oracles_graph(fv, ctx):
    # tools/insert_fuzz.py: dimension, degrees, ef_construction, cluster structure, exact duplicates, vectors on a coarse
    # grid (equal distances between different vectors), soft deletes between batches, forced / chosen speculation — every
    # case list by list against the oracle.  (This sweep is what found the tied-maximum case that must send an EXPANDED
    # member away; longer sweeps: python tools/insert_fuzz.py --cases 60 --seed N.)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import insert_fuzz
    rng = np.random.default_rng(1)
    failed = []
    for c in range(56):  # the generator is advanced through every case; the listed ones are built (55: the tied maximum)
        only = c if c in (3, 11, 17, 23, 31, 38, 42, 45, 50, 55) else -2
        if insert_fuzz.one_case(fv, orc, ctx, rng, c, only):
            failed.append(c)
    assert not failed, failed


def test_randomised_operation_sequences_match_the_oracle(fv, ctx):
    # tools/hnsw_ops_fuzz.py: device batch inserts (one at a time / speculated), host-algorithm inserts, single inserts,
    # soft deletes, vacuum, searches by the device traversal and by the host walk, in random order — the hand-overs
    # between the device-resident graph and its host cache; every search and the final graph equal the oracle's
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hnsw_ops_fuzz
    rng = np.random.default_rng(1)
    failed = [c for c in range(12) if hnsw_ops_fuzz.one_case(fv, orc, ctx, rng, c, c if c in (1, 4, 7, 10, 11) else -2)]
    assert not failed, failed
