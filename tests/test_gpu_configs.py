"""GPU parity at the BASELINE.json configurations' shapes (C2 exactly; C3/C4 through emulated shards
and size-independent properties), all through the C ABI."""
import ctypes as C

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


def test_c2_100k_ivf_flat_nlist1024_nprobe32_batch256(fv, ctx):
    # BASELINE.json configs[1]: 100K x 384 f32, IVF-flat nlist=1024 nprobe=32, batch=256
    n, d, nlist, nprobe, B, k = 100_000, 384, 1024, 32, 256, 10
    x = mixture(n, d, n_comp=256, sigma=0.6, seed=101)
    ids = np.arange(n, dtype=np.uint64)
    cents = x[np.random.default_rng(1).choice(n, nlist, replace=False)].copy()
    gpu = fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(cents)
    cl, pos = gpu.add(x, ids)
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
    cpu.set_trained(cents)
    cpu.batch_insert_assigned(ids, x, cl)
    q = mixture(B, d, n_comp=256, sigma=0.6, seed=102)
    # the oracle's own nearest-centroid scan agrees with the GPU assignment on a sample of rows
    assert np.array_equal(cl[:2000], cpu.assign(x[:2000]))
    gi, gd, gc = gpu.search(q, k, nprobe)
    ci, cd, cc = cpu.batch_search(q, k, nprobe, threads=4)
    assert np.array_equal(gc, cc) and np.array_equal(gi, ci) and np.array_equal(bits(gd), bits(cd))
    st = gpu.last_stats()
    assert st["rows_scanned"] > 0 and st["list_rows_touched"] <= n


def test_c4_lists_sharded_over_8_emulated_gpus(fv, ctx):
    # configs[3] scheme at small scale: lists sharded over G "GPUs" (G indexes on one device), per-shard
    # partial top-k with global keys, merge by key == the single index, bit for bit
    n, d, nlist, nprobe, B, k, G = 40_000, 64, 256, 16, 200, 10, 8
    x = mixture(n, d, n_comp=64, sigma=0.7, seed=111)
    ids = np.arange(n, dtype=np.uint64) + 7
    cents = x[:nlist].copy()
    whole = fv.DeviceIVF(ctx, d, nlist)
    whole.set_centroids(cents)
    cl, _ = whole.add(x, ids)
    sizes = np.bincount(cl, minlength=nlist).astype(np.uint64)
    owner = fv.sharded.plan_list_shards(sizes, G)
    q = mixture(B, d, n_comp=64, sigma=0.7, seed=112)
    qd = ctx.upload(q)
    keys_all = ctx.alloc(G * B * k * 8)
    ids_all = ctx.alloc(G * B * k * 8)
    scratch_d, scratch_c = ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    shards = []
    for g in range(G):
        sh = fv.DeviceIVF(ctx, d, nlist)
        sh.set_centroids(cents)
        mine = owner[cl] == g
        sh.add_assigned(x[mine], ids[mine], cl[mine])
        sh.set_global_list_sizes(sizes)
        shards.append(sh)
        off = g * B * k * 8
        sh.search_dev(qd, B, k, nprobe, C.c_void_p(ids_all.value + off), scratch_d, scratch_c,
                      C.c_void_p(keys_all.value + off))
    oi, od, oc = ctx.alloc(B * k * 8), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    fv.engine.merge_keys_dev(ctx, keys_all, ids_all, G, B, k, oi, od, oc)
    ctx.synchronize()
    m_ids, m_ds, m_cnt = ctx.download(oi, (B, k), np.uint64), ctx.download(od, (B, k), np.float32), ctx.download(oc, B, np.uint32)
    w_ids, w_ds, w_cnt = whole.search(q, k, nprobe)
    assert np.array_equal(m_cnt, w_cnt) and np.array_equal(m_ids, w_ids) and np.array_equal(bits(m_ds), bits(w_ds))
    # and against the oracle itself, not only the engine's own single index
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
    cpu.set_trained(cents)
    cpu.batch_insert(ids, x)
    c_ids, c_ds, c_cnt = cpu.batch_search(q, k, nprobe)
    assert np.array_equal(m_cnt, c_cnt)
    for i in range(B):
        nn = int(c_cnt[i])
        assert np.array_equal(m_ids[i, :nn], c_ids[i, :nn]) and np.array_equal(bits(m_ds[i, :nn]), bits(c_ds[i, :nn]))
    assert max(np.bincount(owner, weights=sizes)) - min(np.bincount(owner, weights=sizes)) <= sizes.max()


def test_c3_scale_properties_hybrid(fv, ctx):
    # size-independent properties of the hybrid path at a 200K scale (full 1M runs in bench.py, which
    # also checks its first 256 queries against the oracle): self-match, ordering, idempotence,
    # device traversal == host walk, more probes never lose hits
    from bench import Generator  # the bench's latent-mixture generator
    n, d, nlist, B, k = 200_000, 384, 256, 512, 10
    gen = Generator(d=d)
    x = np.concatenate([gen.rows(10_000, s) for s in range(n // 10_000)])
    ids = np.arange(n, dtype=np.uint64)
    now, day = 1000 * 86400.0, 86400.0
    recent = np.random.default_rng(3).random(n) < 0.3
    ts = np.where(recent, now - day, now - 30 * day)
    hyb = fv.HybridIndex(ctx, n_clusters=nlist, n_probe=16, train_size=20000, max_iterations=10, ivf_seed=1, hnsw_seed=2)
    hyb.initialize(x[:20000])
    hyb.bulk_insert(ids, x, ts, now)
    assert hyb.recent_count() == int(recent.sum()) and hyb.historical_count() == n - int(recent.sum())
    rows = np.random.default_rng(4).choice(n, B, replace=False)
    r1 = hyb.search(x[rows], k, now=now, hnsw_ef=100, ivf_n_probe=32)
    # self-query (tests/integration/large_dataset_tests.rs:199-222): a historical row's own list is always the
    # nearest centroid's, so IVF must return it at distance 0; the graph part is approximate
    hit = (r1.ids[:, 0] == ids[rows]) & (r1.distances[:, 0] == 0.0)
    assert hit[~recent[rows]].all() and hit[recent[rows]].mean() > 0.9
    assert np.all(np.diff(r1.distances, axis=1) >= 0) and np.all(r1.counts == k)
    r2 = hyb.search(x[rows], k, now=now, hnsw_ef=100, ivf_n_probe=32)
    assert np.array_equal(r1.ids, r2.ids) and np.array_equal(bits(r1.distances), bits(r2.distances))
    hyb.hnsw().set_device_traversal(False)
    r3 = hyb.search(x[rows[:128]], k, now=now, hnsw_ef=100, ivf_n_probe=32)
    hyb.hnsw().set_device_traversal(True)
    assert np.array_equal(r3.ids, r1.ids[:128]) and np.array_equal(bits(r3.distances), bits(r1.distances[:128]))
    q = gen.rows(B, 10_000_000)
    lo = hyb.search(q, k, now=now, hnsw_ef=50, ivf_n_probe=4, search_recent=False)
    hi = hyb.search(q, k, now=now, hnsw_ef=50, ivf_n_probe=64, search_recent=False)
    assert np.all(hi.distances[:, k - 1] <= lo.distances[:, k - 1])  # probing more lists can only improve the k-th


@pytest.mark.parametrize("d", [768, 100])
def test_c5_fp16_rows_ivf_flat(fv, ctx, d):
    # BASELINE.json configs[4] shape (768-d fp16 rows, IVF-flat) at test size.  The reference is f32-only;
    # the contract (SURVEY.md section 7 "hard parts") is agreement with the reference algorithm run on the
    # fp16-rounded rows with f32 accumulation: ids identical, distances bit-identical.
    n, nlist, nprobe, B, k = 20_000, 128, 16, 128, 10
    x = mixture(n, d, n_comp=64, sigma=0.6, seed=131)
    x16 = x.astype(np.float16).astype(np.float32)  # round-to-nearest-even, like the insert kernel
    ids = np.arange(n, dtype=np.uint64)
    cents = x[:nlist].copy()
    gpu = fv.DeviceIVF(ctx, d, nlist, dtype="f16")
    gpu.set_centroids(cents)
    cl, pos = gpu.add(x, ids)
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
    cpu.set_trained(cents)
    cpu.batch_insert_assigned(ids, x16, cl)
    q = mixture(B, d, n_comp=64, sigma=0.6, seed=132)
    for kk, npb in ((k, nprobe), (1, 1), (100, 8)):
        gi, gd, gc = gpu.search(q, kk, npb)
        ci, cd, cc = cpu.batch_search(q, kk, npb, threads=4)
        assert np.array_equal(gc, cc) and np.array_equal(gi, ci) and np.array_equal(bits(gd), bits(cd))
    # against exact f32 ground truth: fp16 rounding barely moves the neighbour sets
    f32 = fv.DeviceIVF(ctx, d, nlist)
    f32.set_centroids(cents)
    f32.add(x, ids)
    ei = f32.search_all(q, k)[0]
    ai = gpu.search_all(q, k)[0]
    recall = np.mean([len(set(ei[b].tolist()) & set(ai[b].tolist())) / k for b in range(B)])
    assert recall > 0.97
    dead = np.arange(0, n, 7)
    gpu.set_deleted(cl[dead], pos[dead], True)
    assert not np.isin(gpu.search(q, k, nprobe)[0], dead.astype(np.uint64)).any()
