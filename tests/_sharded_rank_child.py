"""One rank of the multi-rank GPU test (tests/test_00_gpu_sharded_ranks.py launches `world` of these through
torch.distributed.run; all ranks share the box's one GPU, the collectives of the C code path travel over gloo through
the hosted transport).  Every rank runs the product — ShardedHybrid over fvdb_ivf_search_sharded_begin/_end — in WEAK
and STRONG mode, several steps in flight, and compares ITS OWN results bit for bit with the CPU oracle's search of the
unsharded index.  Writes `rank<r>.json` into the directory given as argv[1]."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fvdb_import
    import oracle as orc
    from _data import bits, mixture
    fv = fvdb_import.load()
    sh = fv.sharded
    report = {"rank": rank, "world": world, "checks": []}

    DAY = 86400.0
    # B not a multiple of world: ragged last slice; lists of ~700 rows: the filter thresholds come from well-filled lists
    n, d, nlist, k, nprobe, ef, B = 24000, 32, 24, 10, 6, 40, 37
    x = mixture(n, d, n_comp=16, sigma=1.0, seed=170)
    ids = np.arange(n, dtype=np.uint64) * 3 + 11
    cents = x[:nlist].copy()
    now = 1000 * DAY
    is_recent = np.random.default_rng(170).random(n) < 0.3
    ts = np.where(is_recent, now - 1 * DAY, now - 30 * DAY)

    ctx = fv.Context(0)
    hyb = fv.HybridIndex(ctx, max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist,
                         n_probe=nprobe, hnsw_seed=19)
    hyb.set_ivf_centroids(cents)
    comm, used = sh.bring_up(ctx, dist, torch, "hosted", allow_hosted=True)
    assert used == "hosted"
    sh.self_test(comm)  # the launcher's bring-up check: rank-stamped words through both exchange kinds
    S = sh.ShardedHybrid(hyb, comm)
    S.bulk_insert(ids, x, ts, now)
    report["lists_owned"] = int((S.owner == rank).sum())

    # the oracle: the WHOLE index on the CPU (same centroids, same graph)
    o = orc.HybridIndex(max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist, n_probe=nprobe)
    o.set_ivf_centroids(cents)
    hx, hid = x[~is_recent], ids[~is_recent]
    o.ivf().batch_insert(hid, hx)
    gi, lv, off, nb_ = hyb.hnsw().export_graph()
    pos = {int(v): i for i, v in enumerate(ids)}
    o.hnsw().restore(gi, x[[pos[int(g)] for g in gi]], lv, off, nb_, hyb.hnsw().entry_point())

    def same(res, q_rows):
        oi, od, oc = o.batch_search(q_rows, k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe)
        return bool(res.counts.shape[0] == q_rows.shape[0] and np.array_equal(res.counts, oc) and
                    np.array_equal(res.ids, oi) and np.array_equal(bits(res.distances), bits(od)))

    # transports first: all-gather and all-to-all of device buffers through the C ABI
    blk = 5
    send = np.arange(world * blk, dtype=np.uint32) + 1000 * rank
    dsend, drecv = ctx.upload(send), ctx.alloc(world * world * blk * 4)
    comm.all_gather_dev(dsend, drecv, world * blk * 4)
    got = ctx.download(drecv, (world, world * blk), np.uint32)
    report["checks"].append(["all_gather", bool(all(np.array_equal(got[r], np.arange(world * blk, dtype=np.uint32) + 1000 * r)
                                                    for r in range(world)))])
    comm.all_to_all_dev(dsend, drecv, blk * 4)
    got = ctx.download(drecv, (world, blk), np.uint32)
    report["checks"].append(["all_to_all", bool(all(np.array_equal(got[r], np.arange(rank * blk, (rank + 1) * blk, dtype=np.uint32) + 1000 * r)
                                                    for r in range(world)))])

    # WEAK: every rank brings its own batch
    own = [mixture(B, d, n_comp=16, sigma=1.0, seed=900 + 10 * rank + j) for j in range(3)]
    own_dev = [ctx.upload(q) for q in own]
    res = S.search_dev(own_dev[0], B, k, ef, nprobe, sh.WEAK)
    report["checks"].append(["weak", same(res, own[0])])
    # ... several steps in flight
    for j in range(3):
        S.search_dev_begin(j, own_dev[j], B, k, ef, nprobe, sh.WEAK)
    ok = True
    for j in range(3):
        ok &= same(S.search_dev_end(j), own[j])
    report["checks"].append(["weak_in_flight", ok])

    # STRONG: one global batch, identical on every rank; rank r gets slice r
    glob = mixture(B, d, n_comp=16, sigma=1.0, seed=990)
    gdev = ctx.upload(glob)
    per = -(-B // world)
    lo, hi = min(B, rank * per), min(B, (rank + 1) * per)
    res = S.search_dev(gdev, B, k, ef, nprobe, sh.STRONG)
    report["checks"].append(["strong", bool(S.rows(B, sh.STRONG) == hi - lo and same(res, glob[lo:hi]))])
    # a batch smaller than the world: some ranks have an empty slice but still take part in the exchanges
    res = S.search_dev(gdev, 1, k, ef, nprobe, sh.STRONG)
    report["checks"].append(["strong_tiny", bool(res.counts.shape[0] == (1 if rank == 0 else 0) and
                                                 (rank != 0 or same(res, glob[:1])))])
    # ten days later every recent row is due: each rank migrates the same rows at the same step (the owner of a list
    # appends them); results and counts equal the unsharded index's on this GPU, whose migration is oracle-checked
    plain = fv.HybridIndex(ctx, max_connections=8, max_connections_layer_0=16, ef_construction=40, n_clusters=nlist,
                           n_probe=nprobe, hnsw_seed=19)
    plain.set_ivf_centroids(cents)
    plain.bulk_insert(ids, x, ts, now)
    later = now + 10 * DAY

    def same_plain(res, q_rows):
        p = plain.search_dev(ctx.upload(q_rows), q_rows.shape[0], k, now=later, hnsw_ef=ef, ivf_n_probe=nprobe, dim=d)
        return bool(np.array_equal(res.counts, p.counts) and np.array_equal(res.ids, p.ids) and
                    np.array_equal(bits(res.distances), bits(p.distances)))

    res = S.search_dev(own_dev[1], B, k, ef, nprobe, sh.WEAK, now=later)
    ok = same_plain(res, own[1]) and hyb.recent_count() == 0 and hyb.historical_count() == n
    ok = ok and any(len(set(res.ids[b, : res.counts[b]].tolist())) < res.counts[b] for b in range(B))
    report["checks"].append(["weak_after_migration", bool(ok)])
    res = S.search_dev(gdev, B, k, ef, nprobe, sh.STRONG, now=later)
    report["checks"].append(["strong_after_migration", bool(same_plain(res, glob[lo:hi]))])
    # how often the matrix-core filter had to hand a query to the exact rescan on this rank (thresholds shared
    # between the ranks must not make that the normal case)
    fb = C.c_uint64(0)
    ctx.lib.fvdb_ivf_scan_fallbacks(hyb.ivf()._dev(), C.byref(fb))
    report["scan_fallbacks"] = int(fb.value)
    report["queries_scanned"] = int(world * B * 5 + B * 3)  # weak: 5 steps of world*B; strong: B, 1 and B
    report["ok"] = all(c[1] for c in report["checks"])
    json.dump(report, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
