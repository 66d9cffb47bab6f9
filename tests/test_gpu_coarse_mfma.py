"""GPU parity of the matrix-core coarse stage (csrc/kernels_coarse.h).

The MFMA contraction only proposes candidate clusters; the ranking is decided with the reference's
sequential f32 arithmetic (src/ivf/core.rs:645-656, src/core/vector_ops.rs:51-57).  Bar: probe lists and
distances bit-identical to (a) the oracle's stable sort over every centroid and (b) the engine's own exact
scan (FVDB_COARSE_EXACT) — including inputs built to defeat the proposal (ties, huge common offsets), where
the kernel must notice and rank exactly.
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


def oracle_coarse(q, cents, nprobe):
    cl = np.empty((q.shape[0], nprobe), np.uint32)
    ds = np.empty((q.shape[0], nprobe), np.float32)
    for i in range(q.shape[0]):
        dist = orc.l2_batch(q[i], cents)
        order = np.argsort(dist, kind="stable")[:nprobe]  # stable: lowest cluster id wins ties
        cl[i] = order
        ds[i] = dist[order]
    return cl, ds


def check(fv, ctx, cents, q, nprobe, expect_fallbacks=None):
    nlist, d = cents.shape
    gpu = fv.DeviceIVF(ctx, d, nlist)
    gpu.set_centroids(cents)
    cl_a, ds_a = gpu.coarse(q, nprobe)
    fb = gpu.coarse_fallbacks()
    gpu.set_coarse_mode(1)
    cl_e, ds_e = gpu.coarse(q, nprobe)
    assert gpu.coarse_fallbacks() == fb  # the exact scan never touches the counter
    cl_o, ds_o = oracle_coarse(q, cents, nprobe)
    assert np.array_equal(cl_a, cl_o) and np.array_equal(bits(ds_a), bits(ds_o))
    assert np.array_equal(cl_e, cl_o) and np.array_equal(bits(ds_e), bits(ds_o))
    if expect_fallbacks == "none":
        assert fb == 0
    elif expect_fallbacks == "some":
        assert fb > 0
    gpu.close() if hasattr(gpu, "close") else None
    return fb


@pytest.mark.parametrize("d,nlist,B,nprobe", [(384, 1024, 257, 32), (384, 1024, 64, 48), (128, 100, 45, 10),
                                              (64, 200, 33, 1), (16, 64, 5, 16), (768, 333, 70, 24)])
def test_mfma_coarse_matches_oracle(fv, ctx, d, nlist, B, nprobe):
    cents = mixture(nlist, d, seed=11)
    q = mixture(B, d, seed=12)
    check(fv, ctx, cents, q, nprobe, expect_fallbacks="none")


def test_shapes_outside_the_mfma_path_use_the_exact_scan(fv, ctx):
    # d % 16 != 0, n_clusters < 64, nprobe > 48: same answers through the exact scan
    for d, nlist, nprobe in [(50, 128, 8), (64, 40, 8), (64, 256, 64)]:
        check(fv, ctx, mixture(nlist, d, seed=21), mixture(17, d, seed=22), nprobe, expect_fallbacks="none")


def test_ties_force_exact_ranking(fv, ctx):
    # 256 centroids that are 8 distinct vectors repeated: every rank is a 32-way tie, the proposal cannot be
    # proven, and the (distance, cluster id) order must still be the stable sort's
    d, nlist = 64, 256
    base = mixture(8, d, seed=31)
    cents = np.ascontiguousarray(base[np.arange(nlist) % 8])
    q = mixture(40, d, seed=32)
    fb = check(fv, ctx, cents, q, 40, expect_fallbacks="some")
    assert fb == q.shape[0]


def test_large_common_offset_defeats_the_error_bound(fv, ctx):
    # |q|, |c| ~ 1e3 * sqrt(d) with unit spread: the expanded form |q|^2 - 2 q.c + |c|^2 cancels catastrophically,
    # the bound cannot close, every query is ranked exactly — and still matches the reference's arithmetic
    d, nlist = 128, 512
    rng = np.random.default_rng(41)
    cents = (1000.0 + rng.standard_normal((nlist, d))).astype(np.float32)
    q = (1000.0 + rng.standard_normal((50, d))).astype(np.float32)
    check(fv, ctx, cents, q, 16, expect_fallbacks="some")


def test_assign_and_search_agree_between_modes(fv, ctx):
    d, nlist, n = 128, 256, 20000
    cents = mixture(nlist, d, seed=51)
    x = mixture(n, d, seed=52)
    ids = np.arange(n, dtype=np.uint64)
    q = mixture(100, d, seed=53)
    out = []
    for mode in (0, 1):
        gpu = fv.DeviceIVF(ctx, d, nlist)
        gpu.set_centroids(cents)
        gpu.set_coarse_mode(mode)
        cl, pos = gpu.add(x, ids)
        out.append((cl, pos, gpu.search(q, 10, 16)))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(bits(a) if a.dtype == np.float32 else a, bits(b) if b.dtype == np.float32 else b)
    cpu = orc.IVFIndex(n_clusters=nlist, n_probe=16)
    cpu.set_trained(cents)
    cpu.batch_insert(ids, x)
    ci, cd, cc = cpu.batch_search(q, 10, 16)
    gi, gd, gc = out[0][2]
    assert np.array_equal(gc, cc)
    for i in range(q.shape[0]):
        assert np.array_equal(gi[i, :gc[i]], ci[i, :cc[i]]) and np.array_equal(bits(gd[i, :gc[i]]), bits(cd[i, :cc[i]]))
