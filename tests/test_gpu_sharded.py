"""The multi-GPU code path on one rank: RCCL itself (ncclCommInitRank with one rank; the collectives are then
copies), ShardedHybrid over fvdb_ivf_search_sharded_begin/_end in WEAK and STRONG mode, against the CPU oracle.
(Several ranks: tests/test_00_gpu_sharded_ranks.py.)"""
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


def test_rccl_communicator_of_one_rank(fv, ctx):
    sh = fv.sharded
    comm = sh.Comm.rccl(ctx)  # librccl is dlopen'ed here
    sh.self_test(comm)
    assert (comm.world, comm.rank) == (1, 0)
    assert ctx.lib.fvdb_comm_world(comm.h) == 1 and ctx.lib.fvdb_comm_rank(comm.h) == 0
    a = np.arange(1000, dtype=np.uint32)
    da, db = ctx.upload(a), ctx.alloc(a.nbytes)
    comm.all_gather_dev(da, db, a.nbytes)
    ctx.synchronize()
    assert np.array_equal(ctx.download(db, a.shape, np.uint32), a)
    dc = ctx.alloc(a.nbytes)
    comm.all_to_all_dev(da, dc, a.nbytes)
    ctx.synchronize()
    assert np.array_equal(ctx.download(dc, a.shape, np.uint32), a)
    comm.close()


def test_sharded_hybrid_world_1_over_rccl_matches_oracle(fv, ctx):
    sh = fv.sharded
    n, d, nlist, k, nprobe, ef, B = 8000, 48, 32, 10, 8, 50, 100
    x = mixture(n, d, n_comp=16, sigma=1.0, seed=180)
    ids = np.arange(n, dtype=np.uint64) + 5
    cents = x[:nlist].copy()
    now = 1000 * DAY
    is_recent = np.random.default_rng(180).random(n) < 0.3
    ts = np.where(is_recent, now - 1 * DAY, now - 30 * DAY)
    hyb = fv.HybridIndex(ctx, n_clusters=nlist, n_probe=nprobe, hnsw_seed=23)
    hyb.set_ivf_centroids(cents)
    comm = sh.Comm.rccl(ctx)
    S = sh.ShardedHybrid(hyb, comm)
    S.bulk_insert(ids, x, ts, now)
    assert np.all(S.owner == 0)
    o = orc.HybridIndex(n_clusters=nlist, n_probe=nprobe)
    o.set_ivf_centroids(cents)
    o.ivf().batch_insert(ids[~is_recent], x[~is_recent])
    gi, lv, off, nb_ = hyb.hnsw().export_graph()
    o.hnsw().restore(gi, x[(gi - 5).astype(np.int64)], lv, off, nb_, hyb.hnsw().entry_point())
    qs = [mixture(B, d, n_comp=16, sigma=1.0, seed=181 + j) for j in range(4)]
    qd = [ctx.upload(q) for q in qs]
    want = [o.batch_search(q, k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe) for q in qs]
    for mode in (sh.WEAK, sh.STRONG):
        for j in range(4):
            S.search_dev_begin(j, qd[j], B, k, ef, nprobe, mode)
        for j in range(4):
            r = S.search_dev_end(j)
            oi, od, oc = want[j]
            assert np.array_equal(r.counts, oc) and np.array_equal(r.ids, oi) and np.array_equal(bits(r.distances), bits(od))
    # and it equals the unsharded product path
    plain = fv.HybridIndex(ctx, n_clusters=nlist, n_probe=nprobe, hnsw_seed=23)
    plain.set_ivf_centroids(cents)
    plain.bulk_insert(ids, x, ts, now)
    p = plain.search_dev(qd[0], B, k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe, dim=d)
    r = S.search_dev(qd[0], B, k, ef, nprobe)
    assert np.array_equal(p.ids, r.ids) and np.array_equal(bits(p.distances), bits(r.distances))
    # mutations are refused while a sharded step is in flight, like any other batch
    S.search_dev_begin(0, qd[0], B, k, ef, nprobe)
    with pytest.raises(fv.FvdbError):
        hyb.insert_with_timestamp(10**6, x[0], now, now)
    S.search_dev_end(0)
    # ten days later every recent row is due: the per-search auto-migration (src/hybrid/core.rs:437-439) copies them into
    # the lists of their owner ranks; counts and the duplicate-bearing merged results equal the unsharded index's (whose
    # migration is held against the oracle in test_hybrid_auto_migration_on_device_query_path)
    later = now + 10 * DAY
    for mode in (sh.WEAK, sh.STRONG):
        r = S.search_dev(qd[1], B, k, ef, nprobe, mode, now=later)
        p = plain.search_dev(qd[1], B, k, now=later, hnsw_ef=ef, ivf_n_probe=nprobe, dim=d)
        assert hyb.recent_count() == plain.recent_count() == 0 and hyb.historical_count() == plain.historical_count() == n
        assert np.array_equal(p.counts, r.counts) and np.array_equal(p.ids, r.ids) and np.array_equal(bits(p.distances), bits(r.distances))
        assert any(len(set(r.ids[b, : r.counts[b]].tolist())) < r.counts[b] for b in range(B))  # a row found in both parts
    comm.close()


def test_product_library_has_no_loopback_communicator(fv, ctx):
    # the capacity-planning communicator (results are not search results) lives in the dev build only (include/fvdb_dev.h)
    assert not hasattr(ctx.lib, "fvdb_comm_create_loopback")
    with pytest.raises(RuntimeError):
        fv.sharded.Comm.loopback(ctx, 4, 1)
