#!/usr/bin/env python
"""bench.py — k-NN queries/sec at recall@10 >= 0.95 on the BASELINE.json headline workload.

Workload (config.workload "c3"): 1M x 384 f32, hybrid HNSW/IVF (30 % recent -> HNSW, 70 %
historical -> IVF-flat nlist 1024, inserted in 10K-vector chunks), batch = 1024 queries, k = 10.
One "step" = one batch of 1024 queries through HybridIndex.search (auto-migration check, IVF
coarse + list scan + top-k on the GPU, HNSW traversal on the host with every hop's candidate
batch scored on the GPU, stable merge).  Query batches are resident in HBM before the timed
region starts; results return to host memory (the reference API returns them to the caller).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`:
every rank brings its own batch of queries per step (weak scaling); IVF lists are sharded over the ranks (the
query batches are all-gathered, every rank scans the probed lists it owns for ALL queries, the per-rank partial
top-k are all-gathered over RCCL and merged by key on the rank that owns the query), the HNSW graph is replicated
and each rank searches it for its own queries.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
"roofline" (dominant kernel = the list-scan kernel, timed with HIP events on its stream) and
"cpu_baseline" (the CPU oracle = the reference algorithm restated, timed on this box's host cores
on a bounded sample of the same queries; it is only the checker/baseline, never the product path).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DAY = 86400.0


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------
# synthetic data: latent Gaussian mixture embedded in d dims (see DESIGN.md "Synthetic data")
# ------------------------------------------------------------------------------------------------
class Generator:
    """rows = (mean[c] + N(0, I_L)) @ P + ambient * N(0, I_d); means ~ spread * N(0, I_L), P: L x d orthonormal.
    Counter-based Philox streams keyed by (seed, chunk) so any rank can regenerate any chunk."""

    def __init__(self, d=384, latent=32, n_comp=4096, spread=1.5, ambient=0.02, seed=1234):
        self.d, self.L, self.n_comp, self.ambient, self.seed = d, latent, n_comp, ambient, seed
        r = np.random.Generator(np.random.Philox(key=seed))
        self.means = (np.float32(spread) * r.standard_normal((n_comp, latent))).astype(np.float32)
        qmat, _ = np.linalg.qr(r.standard_normal((d, latent)))
        self.P = np.ascontiguousarray(qmat.T.astype(np.float32))  # L x d, orthonormal rows

    def rows(self, n, stream):
        r = np.random.Generator(np.random.Philox(key=self.seed + 1000003 * (stream + 1)))
        comp = r.integers(0, self.n_comp, n)
        z = self.means[comp] + r.standard_normal((n, self.L), dtype=np.float32)
        x = z @ self.P
        if self.ambient:
            x += np.float32(self.ambient) * r.standard_normal((n, self.d), dtype=np.float32)
        return np.ascontiguousarray(x, dtype=np.float32)


def usable_cpus():
    """CPUs this process may use: min(affinity, cgroup quota) — the GPU box shows 256 and grants 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        pass
    return max(1, n)


def recall_at_k(found_ids, found_cnt, exact_ids, k):
    hits = 0
    for b in range(found_ids.shape[0]):
        hits += len(set(found_ids[b, : found_cnt[b]].tolist()) & set(exact_ids[b, :k].tolist()))
    return hits / (k * found_ids.shape[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-vectors", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nlist", type=int, default=1024)
    ap.add_argument("--nprobe", type=int, default=0, help="0 = smallest of the sweep reaching the recall target")
    ap.add_argument("--ef", type=int, default=0, help="0 = smallest of the sweep reaching the recall target")
    ap.add_argument("--recent-frac", type=float, default=0.3)
    ap.add_argument("--recall-target", type=float, default=0.95)
    ap.add_argument("--train-sample", type=int, default=100_000)
    ap.add_argument("--query-batches", type=int, default=4)
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hnsw-traversal", choices=["device", "host"], default="device",
                    help="device: the layered walk runs on the GPU (one launch per batch); host: on the host with one "
                         "candidate-scoring launch per hop (the north_star's split).  Identical results.")
    ap.add_argument("--compare-host-walk", type=int, default=5, help="extra steps timed with the host walk (0 = skip)")
    ap.add_argument("--parts", choices=["both", "recent", "historical"], default="both",
                    help="development aid: time only the HNSW or only the IVF part of the hybrid search (recall is then meaningless)")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="batches in flight during the timed region (1 = each step collected before the next is enqueued)")
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--spread", type=float, default=1.5)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    dist = None
    torch = None
    force_sharded = os.environ.get("FVDB_FORCE_SHARDED") == "1"  # exercise the multi-GPU code path on one rank
    if world > 1 or force_sharded:
        import torch  # noqa: F811  (first, so this process uses one HIP runtime)
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("FVDB_DIST_BACKEND", "nccl")  # "gloo": rehearsal of several ranks on ONE GPU
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
        # RCCL prints a version banner on stdout at its first collective; stdout must carry ONE JSON line, so the
        # first collective runs with fd 1 pointed at stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    import fvdb_import
    fv = fvdb_import.load()

    N, d, B, k = args.n, args.dim, args.batch, args.k
    now = 1000 * DAY
    t_setup = time.time()
    gen = Generator(d=d, latent=args.latent, spread=args.spread)
    chunk = 10_000  # the reference's storage chunk (src/hybrid/persistence.rs:189) = our generation/insert unit
    x = np.empty((N, d), np.float32)
    for c in range(0, N, chunk):
        x[c:c + chunk] = gen.rows(min(chunk, N - c), stream=c // chunk)
    ids = np.arange(N, dtype=np.uint64)
    r = np.random.Generator(np.random.Philox(key=99))
    is_recent = r.random(N) < args.recent_frac
    ts = np.where(is_recent, now - 1 * DAY, now - 30 * DAY)
    nb = max(1, args.query_batches)
    # every rank brings its own query batches (weak scaling: world x B queries per step)
    queries = [gen.rows(B, stream=10_000_000 + 1000 * rank + i) for i in range(nb)]
    log(f"data: {N} x {d} generated in {time.time() - t_setup:.1f}s; recent={int(is_recent.sum())}")

    ctx_ivf = fv.Context(local_rank)
    ctx_hnsw = fv.Context(local_rank)
    hyb = fv.HybridIndex(ctx_ivf, ctx_hnsw=ctx_hnsw, n_clusters=args.nlist, n_probe=min(32, args.nlist),
                         train_size=args.train_sample, max_iterations=25, ivf_seed=7, hnsw_seed=11)
    t0 = time.time()
    sample = x[np.random.Generator(np.random.Philox(key=5)).choice(N, min(args.train_sample, N), replace=False)]
    hyb.initialize(sample)
    log(f"IVF k-means ({args.nlist} lists on {sample.shape[0]} rows, GPU): {time.time() - t0:.1f}s")

    # ---- placement: single GPU = everything; multi GPU = lists sharded, HNSW replicated ----
    t0 = time.time()
    sharded = None
    if world == 1 and not force_sharded:
        hyb.bulk_insert(ids, x, ts, now)
    else:
        from fabstir_vectordb_amd import sharded as sh
        sharded = sh.ShardedHybrid(fv, hyb, rank, world, dist, torch)
        sharded.bulk_insert(ids, x, ts, now)
    log(f"index build (HNSW bulk graph {hyb.recent_count()} nodes + IVF {hyb.historical_count()} rows): "
        f"{time.time() - t0:.1f}s")

    # ---- exact ground truth on the GPU (flat scan of all N rows) ----
    t0 = time.time()
    flat = fv.DeviceIVF(ctx_ivf, d, 1)
    flat.set_centroids(np.zeros((1, d), np.float32))
    for c in range(0, N, 200_000):
        flat.add_assigned(x[c:c + 200_000], ids[c:c + 200_000], np.zeros(min(200_000, N - c), np.uint32))
    exact = [flat.search_all(q, k)[0] for q in queries]
    flat.close()
    log(f"exact ground truth: {time.time() - t0:.1f}s")

    qdev = [ctx_ivf.upload(q) for q in queries] if sharded is None else [sharded.upload_queries(q) for q in queries]
    hyb.hnsw().set_device_traversal(args.hnsw_traversal == "device")

    def run(i, nprobe, ef):
        if sharded is not None:
            return sharded.search_dev(qdev[i % nb], B, k, now, ef, nprobe)
        return hyb.search_dev(qdev[i % nb], B, k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe, dim=d)

    def mean_over_ranks(v):
        """Same value on every rank (decisions taken on it keep the ranks' collectives in lockstep)."""
        if dist is None or world == 1:
            return float(v)
        t = torch.tensor([float(v)], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item()) / world

    # ---- operating point: smallest (nprobe, ef) of the sweep with recall@k >= target ----
    sweep = []
    nprobe, ef = args.nprobe, args.ef
    if nprobe == 0 or ef == 0:
        chosen = None
        for e_ in ([ef] if ef else [50, 100, 200]):
            for p_ in ([nprobe] if nprobe else [8, 16, 24, 32, 48, 64, 96, 128]):
                p_ = min(p_, args.nlist)
                res = run(0, p_, e_)
                rec = mean_over_ranks(recall_at_k(res.ids, res.counts, exact[0], k))
                if rec >= args.recall_target:  # confirm on every query batch before accepting
                    rec = mean_over_ranks(np.mean([recall_at_k(res.ids, res.counts, exact[0], k)] +
                                                  [recall_at_k(*(lambda r_: (r_.ids, r_.counts))(run(i, p_, e_)), exact[i], k)
                                                   for i in range(1, nb)]))
                sweep.append({"nprobe": p_, "ef": e_, "recall": round(rec, 4)})
                log(f"sweep nprobe={p_} ef={e_}: recall@{k}={rec:.4f}")
                if rec >= args.recall_target and chosen is None:
                    chosen = (p_, e_)
                    break
            if chosen:
                break
        if chosen is None:
            chosen = (sweep[-1]["nprobe"], sweep[-1]["ef"])
            log("WARNING: recall target not reached in the sweep; using the largest setting")
        nprobe, ef = chosen
    log(f"operating point: nprobe={nprobe} ef={ef}")

    # ---- timed region ----
    depth = max(1, min(args.in_flight, 8))
    kw = dict(now=now, hnsw_ef=ef, ivf_n_probe=nprobe, dim=d, search_recent=args.parts != "historical",
              search_historical=args.parts != "recent")

    if sharded is None:
        begin = lambda slot, i: hyb.search_dev_begin(slot, qdev[i % nb], B, k, **kw)  # noqa: E731
        end = hyb.search_dev_end
    else:  # every rank runs the same sequence of begin/end calls (each holds one collective)
        begin = lambda slot, i: sharded.search_dev_begin(slot, qdev[i % nb], B, k, now, ef, nprobe)  # noqa: E731
        end = sharded.search_dev_end

    def pipelined(nsteps):
        """nsteps searches with up to `depth` batches in flight; returns the last result and the host time spent
        enqueuing / collecting."""
        t_begin = t_end = 0.0
        res = None
        for i in range(nsteps):
            ta = time.perf_counter()
            begin(i % depth, i)  # slot i % depth was collected one iteration ago
            tb = time.perf_counter()
            if i >= depth - 1:
                res = end((i - depth + 1) % depth)
            t_begin += tb - ta
            t_end += time.perf_counter() - tb
        for i in range(max(nsteps - depth + 1, 0), nsteps):
            ta = time.perf_counter()
            res = end(i % depth)
            t_end += time.perf_counter() - ta
        return res, t_begin, t_end

    ctx_ivf.set_profiling(2)      # on before the warm-up: the first profiled search creates events etc.
    ctx_hnsw.set_profiling(2)     # HIP events around the traversal kernel (its own stream)
    # the multi-GPU path has more lazily initialised parts (RCCL channels, torch's staging buffers): extra warm-up
    for i in range(args.warmup if sharded is None else max(args.warmup, 12)):
        run(i, nprobe, ef)
    if depth > 1:
        pipelined(max(args.warmup, 2 * depth))  # every slot has its stream, buffers and traversal state before timing
    log("warmup done")
    hyb.ivf_device_stage_times()  # reset accumulators
    hnsw = hyb.hnsw()
    hnsw.graph_kernel_times()
    if dist is not None:
        # a barrier is followed, a few searches later, by a one-off stall of tens of ms (seen with RCCL on one rank
        # as well): take it outside the timed region — barrier, a few untimed searches, barrier again
        dist.barrier()
        torch.cuda.synchronize()
        for i in range(6):
            run(i, nprobe, ef)
        hyb.ivf_device_stage_times()
        hnsw.graph_kernel_times()
    evals0, hops0 = hnsw.dist_evals(), hnsw.hops()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    ctx_ivf.synchronize()
    ctx_hnsw.synchronize()
    t0 = time.perf_counter()
    last = None
    if depth == 1:
        for i in range(args.steps):
            last = run(i, nprobe, ef)
    else:
        # up to `depth` batches in flight: step i is enqueued (graph walk on its own stream, IVF chain behind the
        # previous batch's on the IVF stream) before step i - depth + 1 is collected and merged on the host.  Every
        # step's results are complete, on the host, inside the timed region.
        last, t_begin, t_end = pipelined(args.steps)
    t_loop = time.perf_counter() - t0
    ctx_ivf.synchronize()
    ctx_hnsw.synchronize()
    if dist is not None:
        torch.cuda.synchronize()
        t_sync = time.perf_counter() - t0
        dist.barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f}s" + (f" (loop {t_loop:.4f}s, synced {t_sync:.4f}s)" if dist is not None else ""))
    if depth > 1:
        log(f"host time per step: enqueue {t_begin / args.steps * 1e3:.3f} ms, collect+merge (incl. waiting) {t_end / args.steps * 1e3:.3f} ms")
    if dist is not None:
        tt = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    graph_ms_sum, graph_launches, graph_rows, graph_hops = hnsw.graph_kernel_times()
    # per-stage timing of the IVF chain: a few more steps, one batch at a time (the events of one chain would be
    # overwritten by the next while several batches are in flight)
    hyb.ivf_device_stage_times()
    for i in range(5):
        run(i, nprobe, ef)
    ctx_ivf.set_profiling(0)
    ctx_hnsw.set_profiling(0)
    hnsw.graph_kernel_times()
    n_prof, stage = hyb.ivf_device_stage_times()
    stats = hyb.ivf_device_last_stats()
    evals, hops = hnsw.dist_evals() - evals0, hnsw.hops() - hops0

    # the same workload with the other traversal mode (reported next to the headline value)
    other = None
    if args.compare_host_walk > 0 and world == 1:
        hnsw.set_device_traversal(args.hnsw_traversal != "device")
        run(0, nprobe, ef)
        ctx_ivf.synchronize()
        t1 = time.perf_counter()
        for i in range(args.compare_host_walk):
            run(i, nprobe, ef)
        ctx_ivf.synchronize()
        dt = (time.perf_counter() - t1) / args.compare_host_walk
        other = {"hnsw_traversal": "host" if args.hnsw_traversal == "device" else "device",
                 "value": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4), "steps": args.compare_host_walk}
        hnsw.set_device_traversal(args.hnsw_traversal == "device")
        log(f"other traversal mode ({other['hnsw_traversal']}): {other['ms_per_step']} ms/step, {other['value']:.0f} QPS")

    recs = []
    for i in range(nb):
        res = run(i, nprobe, ef)
        recs.append(recall_at_k(res.ids, res.counts, exact[i], k))
    recall = mean_over_ranks(np.mean(recs))
    qps = world * B * args.steps / elapsed  # every rank completed B queries per step
    ms_per_step = elapsed / args.steps * 1e3
    log(f"{args.steps} steps: {ms_per_step:.3f} ms/step, {qps:.0f} QPS, recall@{k}={recall:.4f}")

    # ---- roofline of the list scan (the HBM-bound kernel of the path) ----
    # ALGORITHMIC bytes per launch = sum over queries of (rows in its probed lists) x d x 4 (section 8d's
    # nprobe*(N/nlist)*d*s per query, measured rather than averaged).  The kernel that does this work is the
    # matrix-core filter (scan_mfma_kernel) when the matrix-core path runs, else scan_topk_kernel; its duration is
    # measured live with HIP events on its launch stream.
    mfma_path = stage.get("mfma_filter_kernel", 0.0) > 0.0
    scan_ms = (stage["mfma_filter_kernel"] if mfma_path else stage["fine_scan"]) / max(n_prof, 1)
    alg_bytes = stats["rows_scanned"] * d * 4  # rows each query's probed lists hold x row bytes
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    phys_bytes = stats["list_rows_touched"] * d * (2 if mfma_path else 4)  # every probed list read once (fp16 mirror)
    graph_ms = graph_ms_sum / max(graph_launches, 1)
    if args.hnsw_traversal == "device":
        evals, hops = graph_rows, graph_hops
    gather_bytes = (evals / max(args.steps, 1)) * d * 4  # rows scored per step x row bytes
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
        "frac": round(achieved / 8000.0, 4), "traffic": None,
        "kernel": ("fvdb::scan_mfma_kernel<2, 1, 0> (IVF list scan: fp16 MFMA filter over every probed row)" if mfma_path
                   else "fvdb::scan_topk_kernel<16, 1, 1> (IVF list scan, exact)"),
        "kernel_ms": round(scan_ms, 4),
        "algorithmic_bytes_per_launch": int(alg_bytes),
        "rows_scanned_per_query": round(stats["rows_scanned"] / B, 1),
        "physical_lower_bound_bytes_per_launch": int(phys_bytes),
        "physical_lower_bound_GBps": round(phys_bytes / (scan_ms * 1e-3) / 1e9, 1) if scan_ms > 0 else 0.0,
        "physical_lower_bound_frac_of_hbm_peak": round(phys_bytes / (scan_ms * 1e-3) / 1e9 / 8000.0, 4) if scan_ms > 0 else 0.0,
        "pair_dims_per_s": round(stats["rows_scanned"] * d / (scan_ms * 1e-3), 1) if scan_ms > 0 else 0.0,
        "note": "queries probing the same list share one read of it (32 per pass), so algorithmic bytes/s exceed the "
                "HBM peak; the physical lower bound is every probed list streamed once per launch, see DESIGN.md",
        "stage_ms": {k_: round(v / max(n_prof, 1), 4) for k_, v in stage.items()},
        "graph_traversal_kernel": {
            "kernel": "fvdb::hnsw_search_kernel<true> (one wavefront per query, whole traversal on the GPU)",
            "kernel_ms": round(graph_ms, 4), "launches_timed": graph_launches,
            "rows_scored_per_query": round(evals / max(args.steps, 1) / B, 1),
            "gathered_GBps": round(gather_bytes / (graph_ms * 1e-3) / 1e9, 1) if graph_ms > 0 else 0.0,
            "note": "latency-bound pointer chase (pop -> adjacency -> visited -> rows -> heaps per hop); it runs "
                    "concurrently with the IVF chain on its own stream and is the longer of the two"},
        "hnsw": {"hops_per_step": round(hops / args.steps, 1), "dist_evals_per_query": round(evals / args.steps / B, 1)},
    }

    # HBM-side traffic of that kernel from PMC counters (FETCH_SIZE / WRITE_SIZE are collected in their own
    # rocprofv3 passes by tools/pmc_traffic.sh for this same command and committed under profiles/;
    # gfx950: FETCH_SIZE counts wide streaming reads at half their bytes => doubled, see MI355X_MICROARCH.md)
    try:
        pmj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        pm = pmj["list_scan"]
        if N == 1_000_000 and nprobe == pmj.get("nprobe") and world == 1 and mfma_path:
            roofline["traffic"] = int((2 * pm["FETCH_SIZE_KB_avg"] + pm["WRITE_SIZE_KB_avg"]) * 1024)
            roofline["traffic_note"] = "bytes per launch beyond L2 (Infinity Cache + HBM), 2*FETCH_SIZE+WRITE_SIZE, profiles/r01_pmc_traffic.json"
            if scan_ms > 0:
                roofline["traffic_GBps"] = round(roofline["traffic"] / (scan_ms * 1e-3) / 1e9, 1)
                roofline["traffic_frac_of_hbm_peak"] = round(roofline["traffic"] / (scan_ms * 1e-3) / 1e9 / 8000.0, 4)
            roofline["pmc"] = {
                "source": "profiles/r01_pmc_traffic.json (separate rocprofv3 --pmc passes of this command)",
                "mfma_busy_frac_of_busy_simd_cycles": {
                    name: round(pmj[name]["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (4.0 * pmj[name]["SQ_BUSY_CU_CYCLES_avg"]), 4)
                    for name in ("coarse_gemm", "threshold_pass", "list_scan") if name in pmj},
            }
    except Exception:
        pass

    # ---- CPU baseline: the oracle (reference algorithm restated) on the same structures ----
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(fv, hyb, sharded, x, ids, is_recent, ts, now, queries[0], last if nb == 1 else run(0, nprobe, ef),
                           k, nprobe, ef, args)

    if rank == 0:
        out = {
            "metric": "k-NN queries/sec at recall@10>=0.95, 1Mx384 f32",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c3: 1M x 384 f32 hybrid HNSW/IVF (10K-vector chunks), batch 1024, k 10",
                       "n_vectors": N, "dim": d, "batch": B, "global_batch": world * B, "k": k, "recent_frac_hnsw": args.recent_frac,
                       "nlist": args.nlist, "nprobe": nprobe, "hnsw_ef": ef, "hnsw_M": 16, "hnsw_M0": 32, "hnsw_traversal": args.hnsw_traversal, "batches_in_flight": depth,
                       "other_traversal_mode": other, "hnsw_device_fallbacks": hnsw.device_fallbacks(),
                       "recall_at_10": round(recall, 4), "recall_target": args.recall_target, "sweep": sweep,
                       "generator": f"gaussian mixture: 4096 comps, means {args.spread}*N(0,I) in a rank-{args.latent} latent space, "
                                    f"unit within-comp sigma, orthonormal embedding into {d}-d + 0.02 ambient noise",
                       "parallelism": "1 gpu" if world == 1 else f"ivf lists sharded x{world} (each rank scans its lists for all {world}x{B} queries) + all-gather of queries and of partial top-k; hnsw graph replicated, each rank searches its own {B} queries"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(fv, hyb, sharded, x, ids, is_recent, ts, now, q0, gpu_res, k, nprobe, ef, args):
    """Times the CPU oracle on a bounded sample and checks the GPU results against it."""
    import oracle as orc
    orc.build()
    t0 = time.time()
    threads = usable_cpus()
    o = orc.HybridIndex(n_clusters=args.nlist, n_probe=min(32, args.nlist))
    o.set_ivf_centroids(hyb.ivf().get_centroids())
    hist = ~is_recent
    hx, hid = x[hist], ids[hist]
    clusters = hyb.ivf().assign(hx)  # the GPU's find_nearest_centroid (parity-tested against the oracle's)
    o.ivf().batch_insert_assigned(hid, hx, clusters)
    gi, lv, off, nb_ = hyb.hnsw().export_graph()
    oh = o.hnsw()
    oh.restore(gi, x[gi.astype(np.int64)], lv, off, nb_, hyb.hnsw().entry_point())
    setup_s = time.time() - t0
    ns = min(args.cpu_sample, q0.shape[0])
    t0 = time.perf_counter()
    oi1, od1, oc1 = o.batch_search(q0[: max(8, ns // 8)], k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe, threads=1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    oi, od, oc = o.batch_search(q0[:ns], k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe, threads=threads)
    tall = time.perf_counter() - t0
    same = bool(np.array_equal(oc, gpu_res.counts[:ns]) and np.array_equal(oi, gpu_res.ids[:ns]) and
                np.array_equal(od.view(np.uint32), gpu_res.distances[:ns].view(np.uint32)))
    log(f"cpu baseline: setup {setup_s:.1f}s; 1 thread {max(8, ns // 8) / t1:.1f} q/s; {threads} threads {ns / tall:.1f} q/s; "
        f"gpu==oracle on the sample: {same}")
    return {"value": round(ns / tall, 2), "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": f"{ns} queries of the first batch, same index structures and (nprobe, ef); one query per thread",
            "single_thread_value": round(max(8, ns // 8) / t1, 2), "gpu_matches_oracle_on_sample": same}


if __name__ == "__main__":
    main()
