#!/usr/bin/env python
"""bench.py — k-NN queries/sec at recall@10 >= 0.95 on the BASELINE.json headline workload.

Workload (config.workload "c3"): 1M x 384 f32, hybrid HNSW/IVF (30 % recent -> HNSW, 70 % historical -> IVF-flat
nlist 1024, inserted in 10K-vector chunks), batch = 1024 queries, k = 10.  One "step" = one batch of 1024 queries
through HybridIndex's search (IVF coarse + list scan + top-k and the whole HNSW traversal on the GPU, the reference's
stable merge on the host).  Query batches are resident in HBM before the timed region starts; every step's results are
complete, in host memory, inside it (the reference API returns them to the caller).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per
GPU.  IVF lists are sharded over the ranks, the graph is replicated; both exchange steps of a search run over RCCL
inside the engine's C ABI (fvdb_comm_*, fvdb_ivf_search_sharded_begin/_end).  torch.distributed (gloo, CPU) only
carries the 128-byte RCCL id to the ranks and implements the barrier / max-over-ranks of the contract.
--scaling weak (default): every rank brings its own batch per step (global batch N x 1024); strong: the global batch
stays 1024, every rank produces the results of its slice.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
"roofline" — the dominant kernel (the graph traversal) against the HBM roofline, with the list scan as a second entry,
durations measured live with HIP events on the launch streams; and "cpu_baseline" — the CPU oracle (the reference
algorithm restated) timed on this box's host cores on a bounded sample of the same queries (checker/baseline only,
never the product path).  --supplementary adds the two side workloads described in DESIGN.md (isotropic-384 IVF,
sequential-insert hybrid); their outputs are committed under profiles/.
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
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------
# synthetic data
# ------------------------------------------------------------------------------------------------
class Generator:
    """rows = (mean[c] + N(0, I_L)) @ P + ambient * N(0, I_d); means ~ spread * N(0, I_L), P: L x d orthonormal
    (DESIGN.md "Synthetic data").  Counter-based Philox streams keyed by (seed, chunk): any rank regenerates any chunk."""

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


class IsotropicGenerator:
    """SURVEY §8d's generator: 4096 component means ~ N(0, I_d), row = mean + 0.35 N(0, I_d)."""

    def __init__(self, d=384, n_comp=4096, sigma=0.35, seed=1234):
        self.d, self.n_comp, self.sigma, self.seed = d, n_comp, sigma, seed
        r = np.random.Generator(np.random.Philox(key=seed))
        self.means = r.standard_normal((n_comp, d), dtype=np.float32)

    def rows(self, n, stream):
        r = np.random.Generator(np.random.Philox(key=self.seed + 1000003 * (stream + 1)))
        comp = r.integers(0, self.n_comp, n)
        return np.ascontiguousarray(self.means[comp] + np.float32(self.sigma) * r.standard_normal((n, self.d), dtype=np.float32))


def launch_ranks(n):
    """`python bench.py --gpus N` from a plain shell: the same command under torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1.  The child's stdout (one JSON line) is passed through; its exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"launching {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


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
    return hits / (k * max(found_ids.shape[0], 1))


def exact_ground_truth(fv, ctx, x, ids, queries, k):
    """Flat f32 scan on the GPU (exact k-NN)."""
    N, d = x.shape
    flat = fv.DeviceIVF(ctx, d, 1)
    flat.set_centroids(np.zeros((1, d), np.float32))
    for c in range(0, N, 200_000):
        flat.add_assigned(x[c:c + 200_000], ids[c:c + 200_000], np.zeros(min(200_000, N - c), np.uint32))
    out = [flat.search_all(q, k)[0] for q in queries]
    flat.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
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
    ap.add_argument("--query-batches", type=int, default=16, help="distinct query batches cycled through the steps")
    ap.add_argument("--select-batches", type=int, default=4,
                    help="batches the operating point is chosen on; the rest are held out and must confirm it")
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hnsw-traversal", choices=["device", "host"], default="device",
                    help="device: the layered walk runs on the GPU (one launch per batch); host: on the host with one "
                         "candidate-scoring launch per hop (the north_star's split).  Identical results.")
    ap.add_argument("--hnsw-graph", choices=["sequential", "bulk"], default="sequential",
                    help="sequential: the reference's own build — HNSWIndex::insert in id order (src/hnsw/core.rs:226-378), "
                         "device-resident; bulk: exact nearest-M per layer (an extension: another graph)")
    ap.add_argument("--insert-sample", type=int, default=2048,
                    help="inserts timed on GPU and CPU oracle at the full graph size for the insert_path object (0 = skip)")
    ap.add_argument("--compare-host-walk", type=int, default=3, help="extra steps timed with the host walk (0 = skip)")
    ap.add_argument("--parts", choices=["both", "recent", "historical"], default="both",
                    help="development aid: time only the HNSW or only the IVF part of the hybrid search (recall is then meaningless)")
    ap.add_argument("--in-flight", type=int, default=8,
                    help="batches in flight during the timed region (1 = each step collected before the next is enqueued)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="multi-GPU: see the module docstring")
    ap.add_argument("--transport", choices=["rccl", "hosted"], default=os.environ.get("FVDB_TRANSPORT", "rccl"),
                    help="hosted: exchanges carried over torch.distributed on host buffers (several ranks on ONE GPU)")
    ap.add_argument("--allow-hosted", action="store_true",
                    help="multi-GPU: permit the hosted transport (asked for with --transport hosted, or as the fallback when "
                         "RCCL cannot be brought up); without it such a run exits non-zero, so an n_gpus > 1 line is an RCCL line")
    ap.add_argument("--config", choices=["c3", "c2", "c5"], default="c3",
                    help="BASELINE.json config: c3 (default) = 1M x 384 hybrid, the headline; c2 = 100K x 384 f32 IVF-flat, nlist 1024, "
                         "nprobe 32, batch 256; c5 = 1M x 768 rows stored as fp16, IVF-flat, batch 1024")
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--spread", type=float, default=1.5)
    ap.add_argument("--supplementary", action="store_true",
                    help="also run the isotropic-384 IVF workload and the sequential-insert 100K hybrid (minutes)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.config != "c3":
        return run_ivf_config(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called plainly with --gpus N: start one rank per GPU as a FRESH child process (this process has not touched the
        # GPU and never will), relay its one JSON line, exit with its code
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        args.gpus = world
    dist = torch = None
    # stdout carries ONE JSON line and nothing else: libraries (gloo's "Rank 0 is connected to ..." lines, RCCL's version
    # banner) write to file descriptor 1 whenever they like, so fd 1 is pointed at stderr for the whole run and the
    # line goes to a private copy of the original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    force_sharded = os.environ.get("FVDB_FORCE_SHARDED") == "1"  # exercise the multi-GPU code path on one rank
    if world > 1:
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # rendezvous / barrier / max-over-ranks only: the data path's collectives are RCCL calls inside the engine
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import fvdb_import
    fv = fvdb_import.load()
    sh = fv.sharded
    if world > 1 and args.transport == "rccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)  # counting devices does not initialise the GPU
    elif world > 1:
        local_rank = 0  # hosted rehearsal: every rank on the box's one GPU

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
    nsel = max(1, min(args.select_batches, nb))
    strong = world > 1 and args.scaling == "strong"
    mode = sh.STRONG if strong else sh.WEAK
    # weak: every rank brings its own query batches; strong: one stream of global batches, identical on every rank
    qrank = 0 if strong else rank
    queries = [gen.rows(B, stream=10_000_000 + 1000 * qrank + i) for i in range(nb)]
    log(f"data: {N} x {d} generated in {time.time() - t_setup:.1f}s; recent={int(is_recent.sum())}; {nb} query batches")

    ctx_ivf = fv.Context(local_rank)
    ctx_hnsw = fv.Context(local_rank)
    hyb = fv.HybridIndex(ctx_ivf, ctx_hnsw=ctx_hnsw, n_clusters=args.nlist, n_probe=min(32, args.nlist),
                         train_size=args.train_sample, max_iterations=25, ivf_seed=7, hnsw_seed=11)
    hyb.set_sequential_graph(args.hnsw_graph == "sequential")
    t0 = time.time()
    sample = x[np.random.Generator(np.random.Philox(key=5)).choice(N, min(args.train_sample, N), replace=False)]
    hyb.initialize(sample)
    log(f"IVF k-means ({args.nlist} lists on {sample.shape[0]} rows, GPU): {time.time() - t0:.1f}s")

    # ---- placement: single GPU = everything; multi GPU = lists sharded, HNSW replicated ----
    t0 = time.time()
    sharded = comm = None
    transport_used = None
    if world == 1 and not force_sharded:
        hyb.bulk_insert(ids, x, ts, now)
    else:
        if world > 1:
            try:
                comm, transport_used = sh.bring_up(ctx_ivf, dist, torch, args.transport, log=log, allow_hosted=args.allow_hosted)
            except sh.BringUpFailed as e:
                log(f"FATAL: {e}")
                os._exit(3)  # every rank takes this exit (the decision is an all-reduce): no JSON line is printed
        else:
            comm, transport_used = sh.Comm.rccl(ctx_ivf, dist, torch), "rccl"
            sh.self_test(comm)
        sharded = sh.ShardedHybrid(hyb, comm)
        sharded.bulk_insert(ids, x, ts, now)
        if dist is not None:
            dist.barrier()
    graph_build_s = hyb.recent_build_seconds()
    log(f"index build (HNSW {args.hnsw_graph} graph {hyb.recent_count()} nodes in {graph_build_s:.1f}s + IVF "
        f"{hyb.historical_count()} rows): {time.time() - t0:.1f}s; insert stats {hyb.hnsw().insert_stats()}")

    # ---- exact ground truth on the GPU (flat scan of all N rows) ----
    t0 = time.time()
    exact = exact_ground_truth(fv, ctx_ivf, x, ids, queries, k)
    log(f"exact ground truth ({nb} batches): {time.time() - t0:.1f}s")

    qdev = [ctx_ivf.upload(q) for q in queries]
    hyb.hnsw().set_device_traversal(args.hnsw_traversal == "device")
    # rows of the step this rank produces (strong: its slice of the global batch)
    per = -(-B // world)
    lo, hi = (min(B, rank * per), min(B, (rank + 1) * per)) if strong else (0, B)

    def run(i, nprobe, ef):
        if sharded is not None:
            return sharded.search_dev(qdev[i % nb], B, k, ef, nprobe, mode)
        return hyb.search_dev(qdev[i % nb], B, k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe, dim=d)

    def mean_over_ranks(v, weight=1.0):
        """Same value on every rank (decisions taken on it keep the ranks' collectives in lockstep)."""
        if dist is None:
            return float(v)
        t = torch.tensor([float(v) * weight, weight], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0].item() / max(t[1].item(), 1e-30))

    def recall_of(i, nprobe, ef):
        res = run(i, nprobe, ef)
        return mean_over_ranks(recall_at_k(res.ids, res.counts, exact[i % nb][lo:hi], k), weight=float(hi - lo))

    # ---- operating point: the smallest (nprobe, ef) of the sweep whose recall@k reaches the target on the selection
    # batches AND on the held-out batches ----
    sweep = []
    nprobe, ef = args.nprobe, args.ef
    held_out = None
    if nprobe == 0 or ef == 0:
        chosen = None
        for e_ in ([ef] if ef else [50, 100, 200]):
            for p_ in ([nprobe] if nprobe else [8, 16, 24, 32, 48, 64, 96, 128]):
                p_ = min(p_, args.nlist)
                rec = recall_of(0, p_, e_)
                entry = {"nprobe": p_, "ef": e_}
                if rec >= args.recall_target:
                    rec = float(np.mean([rec] + [recall_of(i, p_, e_) for i in range(1, nsel)]))
                    if rec >= args.recall_target and nb > nsel:
                        ho = float(np.mean([recall_of(i, p_, e_) for i in range(nsel, nb)]))
                        entry["recall_held_out"] = round(ho, 4)
                        if ho >= args.recall_target:
                            held_out = ho
                            chosen = (p_, e_)
                    elif rec >= args.recall_target:
                        chosen = (p_, e_)
                entry["recall"] = round(rec, 4)
                sweep.append(entry)
                log(f"sweep nprobe={p_} ef={e_}: recall@{k}={rec:.4f}" + (f" held-out {entry['recall_held_out']:.4f}" if "recall_held_out" in entry else ""))
                if chosen:
                    break
            if chosen:
                break
        if chosen is None:
            chosen = (sweep[-1]["nprobe"], sweep[-1]["ef"])
            log("WARNING: recall target not reached in the sweep; using the largest setting")
        nprobe, ef = chosen
    log(f"operating point: nprobe={nprobe} ef={ef}")

    # ---- timed region ----
    depth = max(1, min(args.in_flight, 16))
    kw = dict(now=now, hnsw_ef=ef, ivf_n_probe=nprobe, dim=d, search_recent=args.parts != "historical",
              search_historical=args.parts != "recent")
    if sharded is None:
        begin = lambda slot, i: hyb.search_dev_begin(slot, qdev[i % nb], B, k, **kw)  # noqa: E731
        end = hyb.search_dev_end
    else:  # every rank runs the same sequence of begin/end calls (each step holds two collectives)
        begin = lambda slot, i: sharded.search_dev_begin(slot, qdev[i % nb], B, k, ef, nprobe, mode)  # noqa: E731
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

    def sync_all():
        ctx_ivf.device_synchronize()  # every stream of the device (one per batch in flight) = torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    ctx_ivf.set_profiling(2)      # on before the warm-up: the first profiled search creates events etc.
    ctx_hnsw.set_profiling(2)     # HIP events around the traversal kernel (its own stream)
    for i in range(args.warmup if sharded is None else max(args.warmup, 8)):
        run(i, nprobe, ef)
    if depth > 1:
        pipelined(max(args.warmup, 2 * depth))  # every slot has its stream, buffers and traversal state before timing
    log("warmup done")
    hyb.ivf_device_stage_times()  # reset accumulators
    hnsw = hyb.hnsw()
    hnsw.graph_kernel_times()
    evals0, hops0 = hnsw.dist_evals(), hnsw.hops()
    sync_all()
    t0 = time.perf_counter()
    last = None
    t_begin = t_end = 0.0
    if depth == 1:
        for i in range(args.steps):
            last = run(i, nprobe, ef)
    else:
        # up to `depth` batches in flight: step i is enqueued (graph walk on its own stream, IVF chain on the slot's)
        # before step i - depth + 1 is collected and merged on the host.  Every step's results are complete, on the
        # host, inside the timed region.
        last, t_begin, t_end = pipelined(args.steps)
    ctx_ivf.device_synchronize()
    t_local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f}s (this rank {t_local:.3f}s)")
    if depth > 1:
        log(f"host time per step: enqueue {t_begin / args.steps * 1e3:.3f} ms, collect+merge (incl. waiting) {t_end / args.steps * 1e3:.3f} ms")
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # host share of a step: collect + translate + merge of batches whose GPU work is already complete
    host_collect_ms = None
    if depth > 1:
        for i in range(depth):
            begin(i, i)
        sync_all()
        th = time.perf_counter()
        for i in range(depth):
            end(i)
        host_collect_ms = (time.perf_counter() - th) / depth * 1e3
        log(f"host collect+merge with the GPU work already done: {host_collect_ms:.3f} ms per step")
    hnsw.graph_kernel_times()  # launches of the timed region overlap each other: per-launch durations are taken below
    # per-kernel / per-stage timing: a few more steps, ONE batch at a time (with several batches in flight the launches
    # of different batches overlap and stretch each other, and one chain's events would be overwritten by the next)
    hyb.ivf_device_stage_times()
    n_solo = 8
    for i in range(n_solo):
        run(i, nprobe, ef)
    graph_ms_sum, graph_launches, graph_rows, graph_hops = hnsw.graph_kernel_times()
    # and the traversal with the card to itself (no list-scan chain beside it): the kernel's own latency-bound time
    graph_alone_ms = None
    if world == 1 and args.hnsw_traversal == "device":
        for i in range(n_solo):
            hnsw.search_dev(qdev[i % nb], B, d, k, ef)
        ms_a, n_a, _, _ = hnsw.graph_kernel_times()
        graph_alone_ms = ms_a / max(n_a, 1)
    ctx_ivf.set_profiling(0)
    ctx_hnsw.set_profiling(0)
    n_prof, stage = hyb.ivf_device_stage_times()
    stats = hyb.ivf_device_last_stats()

    # the same workload with the other traversal mode (reported next to the headline value)
    other = None
    if args.compare_host_walk > 0 and world == 1:
        hnsw.set_device_traversal(args.hnsw_traversal != "device")
        run(0, nprobe, ef)
        ctx_ivf.device_synchronize()
        t1 = time.perf_counter()
        for i in range(args.compare_host_walk):
            run(i, nprobe, ef)
        ctx_ivf.device_synchronize()
        dt = (time.perf_counter() - t1) / args.compare_host_walk
        other = {"hnsw_traversal": "host" if args.hnsw_traversal == "device" else "device",
                 "value": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 4), "steps": args.compare_host_walk}
        hnsw.set_device_traversal(args.hnsw_traversal == "device")
        log(f"other traversal mode ({other['hnsw_traversal']}): {other['ms_per_step']} ms/step, {other['value']:.0f} QPS")

    recall = float(np.mean([recall_of(i, nprobe, ef) for i in range(nb)]))
    global_batch = B if strong else world * B
    qps = global_batch * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    log(f"{args.steps} steps: {ms_per_step:.3f} ms/step, {qps:.0f} QPS, recall@{k}={recall:.4f}")

    roofline = build_roofline(args, d, B, hi - lo, world, stage, n_prof, stats, graph_ms_sum, graph_launches, graph_rows,
                              graph_hops, ms_per_step)
    if graph_alone_ms:
        roofline["kernel_ms_alone"] = round(graph_alone_ms, 4)
        roofline["frac_alone"] = round(roofline["bytes_per_launch"] / (graph_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)

    # ---- CPU baseline: the oracle (reference algorithm restated) on the same structures ----
    cpu = ins = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu, oracle_hyb = cpu_baseline(fv, hyb, x, ids, is_recent, now, queries[0], run(0, nprobe, ef), k, nprobe, ef, args)
        if args.insert_sample > 0:  # last: it grows the graph
            ins = insert_path(hyb, oracle_hyb, gen, N, args.insert_sample, graph_build_s, args)

    supplementary = None
    if args.supplementary and world == 1:
        supplementary = run_supplementary(fv, ctx_ivf, args)

    if rank == 0:
        if world == 1:
            par = "1 gpu"
        elif strong:
            par = (f"ivf lists sharded x{world}: every rank scans its lists for the same {B} queries, all-to-all of the partial "
                   f"top-k, rank r merges and returns slice r; hnsw graph replicated, each rank walks its slice")
        else:
            par = (f"ivf lists sharded x{world}: all-gather of queries + probe lists, every rank scans its lists for all "
                   f"{world}x{B} queries, all-to-all of the partial top-k; hnsw graph replicated, each rank walks its own {B} queries")
        out = {
            "metric": "k-NN queries/sec at recall@10>=0.95, 1Mx384 f32",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c3: 1M x 384 f32 hybrid HNSW/IVF (10K-vector chunks), batch 1024, k 10",
                       "n_vectors": N, "dim": d, "batch": B, "global_batch": global_batch, "k": k, "recent_frac_hnsw": args.recent_frac,
                       "nlist": args.nlist, "nprobe": nprobe, "hnsw_ef": ef, "hnsw_M": 16, "hnsw_M0": 32,
                       "hnsw_graph": ("sequential insert: the reference's HNSWIndex::insert in id order, ef_construction 200, "
                                      "device-resident (fvdb_graph_insert_linked)") if args.hnsw_graph == "sequential" else
                                     "bulk_build: every layer member linked to its exact nearest M (M0) members",
                       "hnsw_graph_build_s": round(graph_build_s, 2),
                       "hnsw_traversal": args.hnsw_traversal, "batches_in_flight": depth, "query_batches": nb,
                       "host_collect_merge_ms_per_step": None if host_collect_ms is None else round(host_collect_ms, 4),
                       "other_traversal_mode": other, "hnsw_device_fallbacks": hnsw.device_fallbacks(),
                       "hnsw_tie_restarts": dict(zip(("queries_served_by_the_traversal_kernel", "searched_again_with_restated_heaps"),
                                                     hnsw.tie_restarts())),
                       "ivf_scan_fallbacks": int(ivf_scan_fallbacks(ctx_ivf, hyb.ivf())),
                       "recall_at_10": round(recall, 4), "recall_target": args.recall_target,
                       "recall_held_out_batches": None if held_out is None else round(held_out, 4), "sweep": sweep,
                       "generator": f"gaussian mixture: 4096 comps, means {args.spread}*N(0,I) in a rank-{args.latent} latent space, "
                                    f"unit within-comp sigma, orthonormal embedding into {d}-d + 0.02 ambient noise",
                       "hw_queues": ctx_ivf.info(),
                       "transport": None if world == 1 and not force_sharded else transport_used,
                       "parallelism": par},
            "roofline": roofline, "cpu_baseline": cpu, "insert_path": ins,
        }
        if supplementary is not None:
            out["supplementary"] = supplementary
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def build_roofline(args, d, B, rows_out, world, stage, n_prof, stats, graph_ms_sum, graph_launches, graph_rows, graph_hops,
                   ms_per_step):
    """frac = max(physical bytes / HBM peak, useful flops / MFMA peak) / kernel time, per launch, for the dominant
    kernel (the graph traversal) and for the list scan.  Durations: HIP events on the launch stream, one batch at a
    time.  SURVEY §8d's per-query algorithmic GB/s is kept as a separately named field (queries probing one list
    share one read of it, so that figure is not a fraction of anything)."""
    n_prof = max(n_prof, 1)
    # ---- graph traversal: every scored row is gathered once per query (no sharing), plus the adjacency rows ----
    gl = max(graph_launches, 1)
    graph_ms = graph_ms_sum / gl
    rows_pl, hops_pl = graph_rows / gl, graph_hops / gl
    g_bytes = rows_pl * d * 4 + hops_pl * 33 * 4
    g_ach = g_bytes / (graph_ms * 1e-3) / 1e9 if graph_ms > 0 else 0.0
    # ---- list scan ----
    mfma_path = stage.get("mfma_filter_kernel", 0.0) > 0.0
    scan_ms = (stage["mfma_filter_kernel"] if mfma_path else stage["fine_scan"]) / n_prof
    phys_bytes = stats["list_rows_touched"] * d * (2 if mfma_path else 4)  # every probed list once (fp16 mirror)
    flops = 2.0 * stats["rows_scanned"] * d                                  # one multiply-add per (row, query, dim)
    t_bytes = phys_bytes / (HBM_PEAK_GBPS * 1e9)
    t_flops = flops / (MFMA_F16_PEAK_TFLOPS * 1e12) if mfma_path else 0.0
    s_frac = max(t_bytes, t_flops) / (scan_ms * 1e-3) if scan_ms > 0 else 0.0
    alg_bytes = stats["rows_scanned"] * d * 4
    roofline = {
        "bound": "hbm", "achieved": round(g_ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(g_ach / HBM_PEAK_GBPS, 4), "traffic": None,
        "kernel": "fvdb::hnsw_search_fast_kernel<3, 16> (HNSW: whole layered traversal, one wavefront per query) — the longest "
                  "kernel of a step; gather-latency bound",
        "kernel_ms": round(graph_ms, 4), "launches_timed": graph_launches,
        "bytes_per_launch": int(g_bytes), "rows_scored_per_query": round(rows_pl / max(rows_out, 1), 1),
        "hops_per_query": round(hops_pl / max(rows_out, 1), 1),
        "note": "bytes = rows scored x d x 4 (each gathered once per query; nothing is shared between queries) + one 132-byte "
                "adjacency row per hop; duration from HIP events on the launch stream, one hybrid batch at a time (the list-scan "
                "chain of the same batch runs beside it on its own stream; kernel_ms_alone / frac_alone: the traversal with the "
                "card to itself)",
        "list_scan": {
            "bound": "hbm" if t_bytes >= t_flops else "mfma",
            "kernel": ("fvdb::scan_mfma_wg_kernel<4> (IVF list scan: fp16 MFMA filter over every probed row, query groups in LDS)" if mfma_path
                       else "fvdb::scan_topk_kernel (IVF list scan, exact)"),
            "kernel_ms": round(scan_ms, 4),
            "physical_lower_bound_bytes": int(phys_bytes), "useful_flops": int(flops),
            "achieved": round(phys_bytes / (scan_ms * 1e-3) / 1e9, 1) if scan_ms > 0 else 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(s_frac, 4),
            "algorithmic_bytes_per_launch_survey_8d": int(alg_bytes),
            "algorithmic_GBps_survey_8d": round(alg_bytes / (scan_ms * 1e-3) / 1e9, 1) if scan_ms > 0 else 0.0,
            "rows_scanned_per_query": round(stats["rows_scanned"] / max(B * (world if args.scaling == "weak" else 1), 1), 1),
            "note": "physical lower bound = every probed list streamed once per launch in fp16; SURVEY 8d's per-query "
                    "algorithmic bytes are shared by the queries probing a list, hence reported, not priced",
        },
        "stage_ms": {k_: round(v / n_prof, 4) for k_, v in stage.items()},
    }
    assert 0.0 <= roofline["frac"] <= 1.0 and 0.0 <= roofline["list_scan"]["frac"] <= 1.0, "roofline fraction out of range"
    # HBM-side traffic from PMC counters (FETCH_SIZE / WRITE_SIZE in their own rocprofv3 passes, tools/pmc_traffic.sh;
    # gfx950: FETCH_SIZE counts wide streaming reads at half their bytes => doubled, see MI355X_MICROARCH.md)
    try:
        pmj = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")))
        if args.n == pmj.get("n_vectors") and world == 1:
            g = pmj.get("graph_traversal")
            if g:
                roofline["traffic"] = int((2 * g["FETCH_SIZE_KB_avg"] + g["WRITE_SIZE_KB_avg"]) * 1024)
                roofline["traffic_note"] = "bytes per launch beyond L2 (Infinity Cache + HBM), 2*FETCH_SIZE+WRITE_SIZE, profiles/r03_pmc_traffic.json"
            ls = pmj.get("list_scan")
            if ls and mfma_path:
                roofline["list_scan"]["traffic"] = int((2 * ls["FETCH_SIZE_KB_avg"] + ls["WRITE_SIZE_KB_avg"]) * 1024)
            tot = (roofline.get("traffic") or 0) + (roofline["list_scan"].get("traffic") or 0)
            assert tot / (ms_per_step * 1e-3) / 1e9 <= HBM_PEAK_GBPS, "PMC traffic per step exceeds the HBM peak"
    except (OSError, KeyError, ValueError):
        pass
    return roofline


def cpu_baseline(fv, hyb, x, ids, is_recent, now, q0, gpu_res, k, nprobe, ef, args):
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
    o.batch_search(q0[: max(8, ns // 8)], k, now=now, hnsw_ef=ef, ivf_n_probe=nprobe, threads=1)
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
            "single_thread_value": round(max(8, ns // 8) / t1, 2), "gpu_matches_oracle_on_sample": same}, o


def run_ivf_config(args):
    """BASELINE.json configs[1] (c2) and configs[4] (c5): IVF-flat only, one GPU, the same JSON contract as the headline.
    A step = one batch through the whole IVF chain (coarse ranking, list scan, selection), queries resident in HBM, up to
    --in-flight batches enqueued on streams of their own; value = queries / wall time of the timed steps."""
    import fvdb_import
    fv = fvdb_import.load()
    cfg = {"c2": dict(N=100_000, d=384, nlist=1024, nprobe=32, B=256, dtype="f32", train=50_000,
                      metric="k-NN queries/sec, 100Kx384 f32 IVF-flat nlist=1024 nprobe=32 batch=256 (BASELINE configs[1])",
                      workload="c2: 100K x 384 f32, IVF-flat nlist 1024 nprobe 32, batch 256, k 10"),
           "c5": dict(N=1_000_000, d=768, nlist=1024, nprobe=32, B=1024, dtype="f16", train=100_000,
                      metric="k-NN queries/sec, 1Mx768 fp16 rows IVF-flat nlist=1024 nprobe=32 batch=1024 (BASELINE configs[4])",
                      workload="c5: 1M x 768 rows stored as fp16, IVF-flat nlist 1024 nprobe 32, batch 1024, k 10")}[args.config]
    N, d, nlist, nprobe, B, k = cfg["N"], cfg["d"], cfg["nlist"], cfg["nprobe"], cfg["B"], args.k
    if args.nprobe:
        nprobe = args.nprobe
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    t0 = time.time()
    gen = Generator(d=d, latent=args.latent, spread=args.spread)
    x = np.empty((N, d), np.float32)
    for c in range(0, N, 10_000):
        x[c:c + 10_000] = gen.rows(min(10_000, N - c), stream=c // 10_000)
    ids = np.arange(N, dtype=np.uint64)
    nb = max(1, min(args.query_batches, 8))
    queries = [gen.rows(B, stream=10_000_000 + i) for i in range(nb)]
    ctx = fv.Context(0)
    ivf = fv.DeviceIVF(ctx, d, nlist, dtype=cfg["dtype"])
    sample = x[np.random.Generator(np.random.Philox(key=5)).choice(N, cfg["train"], replace=False)]
    ivf.train(sample, seed=7, max_iterations=25)  # the reference's k-means (src/ivf/core.rs:240-429), 25 iterations
    ivf.reserve(N)
    clusters = np.empty(N, np.uint32)
    for s_ in range(0, N, 100_000):
        cl, _ = ivf.add(x[s_:s_ + 100_000], ids[s_:s_ + 100_000])
        clusters[s_:s_ + 100_000] = cl
    log(f"{args.config}: {N} x {d} ({cfg['dtype']} rows), k-means + insert {time.time() - t0:.1f}s")
    exact = [ivf.search_all(q, k)[0] for q in queries]
    depth = max(1, min(args.in_flight, 8))
    ctxs = [ctx] + [fv.Context(0) for _ in range(depth - 1)]
    qdev = [ctx.upload(q) for q in queries]
    outs = [(ctx.alloc(B * k * 8), ctx.alloc(B * k * 4), ctx.alloc(B * 4)) for _ in range(depth)]
    lib, h = ctx.lib, ivf.h

    def enqueue(i):
        sl = i % depth
        o = outs[sl]
        ctx.check(lib.fvdb_ivf_search_dev_slot(h, ctxs[sl].h if sl else None, sl, qdev[i % nb], B, k, nprobe, o[0], o[1], o[2], None))

    def result(sl):
        o = outs[sl]
        return (ctx.download(o[0], (B, k), np.uint64), ctx.download(o[1], (B, k), np.float32), ctx.download(o[2], (B,), np.uint32))

    for i in range(max(args.warmup, depth)):
        enqueue(i)
    ctx.device_synchronize()
    rec = []
    for i in range(nb):
        enqueue(i)
        ctxs[i % depth].synchronize()
        gi, _, gc = result(i % depth)
        rec.append(recall_at_k(gi, gc, exact[i], k))
    recall = float(np.mean(rec))
    f0 = ivf.scan_fallbacks()
    ctx.device_synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        enqueue(i)  # stream-ordered per slot: step i waits for step i - depth on its stream, nobody waits on the host
    ctx.device_synchronize()
    elapsed = time.perf_counter() - t1
    qps = B * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    log(f"{args.steps} steps: {ms_per_step:.3f} ms/step, {qps:.0f} QPS, recall@{k}={recall:.4f}, rescans {ivf.scan_fallbacks() - f0}")
    # per-stage / per-kernel durations, one batch at a time (HIP events on the launch stream)
    ctx.set_profiling(1)
    ivf.stage_times()
    n_solo = 8
    for i in range(n_solo):
        ctx.check(lib.fvdb_ivf_search_dev(h, qdev[i % nb], B, k, nprobe, outs[0][0], outs[0][1], outs[0][2], None))
    ctx.synchronize()
    n_prof, stage = ivf.stage_times()
    ctx.set_profiling(0)
    st = ivf.last_stats()
    esz = 2 if (cfg["dtype"] == "f16" or stage.get("mfma_filter_kernel", 0.0) > 0.0) else 4
    mfma_path = stage.get("mfma_filter_kernel", 0.0) > 0.0
    scan_ms = (stage["mfma_filter_kernel"] if mfma_path else stage["fine_scan"]) / max(n_prof, 1)
    phys = st["list_rows_touched"] * d * esz
    flops = 2.0 * st["rows_scanned"] * d
    t_b, t_f = phys / (HBM_PEAK_GBPS * 1e9), (flops / (MFMA_F16_PEAK_TFLOPS * 1e12) if mfma_path else 0.0)
    frac = max(t_b, t_f) / (scan_ms * 1e-3) if scan_ms > 0 else 0.0
    roofline = {"bound": "hbm" if t_b >= t_f else "mfma", "achieved": round(phys / (scan_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(frac, 4), "traffic": None,
                "kernel": "fvdb::scan_mfma_wg_kernel (IVF list scan: fp16 MFMA filter over every probed row)" if mfma_path
                          else "fvdb::scan_topk_kernel (IVF list scan, exact)",
                "kernel_ms": round(scan_ms, 4), "physical_lower_bound_bytes": int(phys), "useful_flops": int(flops),
                "algorithmic_bytes_per_launch_survey_8d": int(st["rows_scanned"] * d * (2 if cfg["dtype"] == "f16" else 4)),
                "rows_scanned_per_query": round(st["rows_scanned"] / B, 1),
                "stage_ms": {k_: round(v / max(n_prof, 1), 4) for k_, v in stage.items()},
                "note": "achieved = every probed list streamed once per launch (fp16 rows / fp16 mirror) / the filter kernel's "
                        "duration (HIP events, one batch at a time); SURVEY 8d's per-query bytes are shared by the queries probing a list"}
    assert 0.0 <= roofline["frac"] <= 1.0
    cpu = None
    if not args.no_cpu_baseline:
        import oracle as orc
        orc.build()
        t2 = time.time()
        o = orc.IVFIndex(n_clusters=nlist, n_probe=nprobe)
        o.set_trained(ivf.get_centroids())
        xo = x.astype(np.float16).astype(np.float32) if cfg["dtype"] == "f16" else x  # the rows the index holds
        o.batch_insert_assigned(ids, xo, clusters)
        ns = min(args.cpu_sample, B)
        threads = usable_cpus()
        ta = time.perf_counter()
        o.batch_search(queries[0][: max(8, ns // 8)], k, nprobe, threads=1)
        t_one = time.perf_counter() - ta
        ta = time.perf_counter()
        oi, od, oc = o.batch_search(queries[0][:ns], k, nprobe, threads=threads)
        t_all = time.perf_counter() - ta
        enqueue(0)
        ctxs[0].synchronize()
        gi, gd, gc = result(0)
        same = bool(np.array_equal(oc, gc[:ns]) and np.array_equal(oi, gi[:ns]) and np.array_equal(od.view(np.uint32), gd[:ns].view(np.uint32)))
        log(f"cpu baseline: setup {time.time() - t2:.1f}s; 1 thread {max(8, ns // 8) / t_one:.1f} q/s; {threads} threads {ns / t_all:.1f} q/s; gpu==oracle: {same}")
        cpu = {"value": round(ns / t_all, 2), "unit": "queries/s", "cores": threads, "kind": "port",
               "sample": f"{ns} queries of the first batch, same centroids, lists and rows"
                         + (" (fp16-rounded, as stored)" if cfg["dtype"] == "f16" else "") + "; one query per thread",
               "single_thread_value": round(max(8, ns // 8) / t_one, 2), "gpu_matches_oracle_on_sample": same}
    out = {"metric": cfg["metric"], "value": round(qps, 1), "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": cfg["workload"], "n_vectors": N, "dim": d, "batch": B, "k": k, "nlist": nlist, "nprobe": nprobe,
                      "row_storage": cfg["dtype"], "recall_at_10": round(recall, 4), "batches_in_flight": depth, "query_batches": nb,
                      "ivf_scan_fallbacks": int(ivf.scan_fallbacks() - f0), "hw_queues": ctx.info(),
                      "arithmetic": "final distances: the reference's f32 fold over the stored rows; fp16 MFMA only discards rows under a proven bound",
                      "generator": f"gaussian mixture: 4096 comps in a rank-{args.latent} latent space embedded into {d}-d + 0.02 ambient noise"},
           "roofline": roofline, "cpu_baseline": cpu}
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def insert_path(hyb, oracle_hyb, gen, N, n_ins, graph_build_s, args):
    """addVectors' path (HNSWIndex::insert, src/hnsw/core.rs:226-378; every addVectors row goes to the HNSW part,
    src/hybrid/core.rs:391-399): `n_ins` fresh rows inserted in order into the benchmark's graph at its full size — on
    the GPU (device-resident insert) and by the CPU oracle holding the same graph — timed, and the resulting lists compared."""
    import oracle as orc
    hn, oh = hyb.hnsw(), oracle_hyb.hnsw()
    n_graph = hn.node_count()
    rows = gen.rows(n_ins, stream=20_000_000)
    new_ids = np.arange(N, N + n_ins, dtype=np.uint64)
    levels = orc.rng_levels(4242, n_ins)
    # a small untimed batch first: it absorbs the one-off capacity doubling of the row store / the host's vector copy that
    # the first insert after a bulk load triggers (~100 ms for 460 MB; amortised over the next 300K inserts in service)
    warm_rows, warm_ids, warm_lv = gen.rows(64, stream=20_500_000), np.arange(N + n_ins, N + n_ins + 64, dtype=np.uint64), orc.rng_levels(4243, 64)
    hn.batch_insert(warm_ids, warm_rows, warm_lv)
    oh.batch_insert(warm_ids, warm_rows, warm_lv)
    n_graph = hn.node_count()
    before = hn.insert_stats()
    t0 = time.perf_counter()
    ok, bad = hn.batch_insert(new_ids, rows, levels)
    t_gpu = time.perf_counter() - t0
    after = hn.insert_stats()
    t0 = time.perf_counter()
    oh.batch_insert(new_ids, rows, levels)
    t_cpu = time.perf_counter() - t0
    same = ok == n_ins and hn.entry_point() == oh.entry_point()
    for i, lv in zip(new_ids.tolist(), levels.tolist()):
        for layer in range(int(lv) + 1):
            nb = hn.neighbors(i, layer)
            same = same and nb == oh.neighbors(i, layer)
            if layer == 0:  # the rows the insert linked back and pruned
                for j in nb[:4]:
                    same = same and hn.neighbors(j, 0) == oh.neighbors(j, 0)
    stats = {k_: after[k_] - before[k_] for k_ in ("speculated_ok", "searched_in_commit", "commit_stops", "tie_restarts",
                                                    "launches", "host_path_inserts")}
    log(f"insert path at {n_graph} nodes: GPU {n_ins / t_gpu:.0f} inserts/s, CPU oracle (1 core) {n_ins / t_cpu:.0f} inserts/s, "
        f"lists identical: {same}; {stats}")
    return {"value": round(n_ins / t_gpu, 1), "unit": "inserts/s", "graph_nodes": n_graph, "hnsw_M": 16, "hnsw_M0": 32,
            "ef_construction": 200,
            "sample": f"{n_ins} fresh rows inserted in order into the benchmark's graph (one batch_insert call; vectors uploaded "
                      f"inside the timed region; preceded by an untimed 64-row batch that absorbs the one-off capacity doubling "
                      f"of the row copies)",
            "cpu_value": round(n_ins / t_cpu, 1), "cpu_cores": 1, "cpu_kind": "port",
            "cpu_note": "the reference's insert is sequential by construction: one core",
            "graph_matches_oracle_on_sample": bool(same), "device_insert": stats,
            "whole_build": {"nodes": n_graph, "seconds": round(graph_build_s, 2),
                            "inserts_per_s": round(n_graph / max(graph_build_s, 1e-9), 1), "graph": args.hnsw_graph}}


def run_supplementary(fv, ctx, args):
    """Two side workloads, so the headline does not rest on the bulk-built graph and the latent-32 generator alone:
    (1) SURVEY §8d's isotropic 384-d mixture, IVF part only (700K rows, nlist 1024): QPS and recall per nprobe;
    (2) a 100K hybrid whose HNSW part (30K nodes) is built by the reference's SEQUENTIAL insert: QPS and recall."""
    out = {}
    d, B, k = args.dim, args.batch, args.k
    # ---- (1) isotropic IVF ----
    gen = IsotropicGenerator(d=d)
    n = 700_000
    x = np.concatenate([gen.rows(10_000, s) for s in range(n // 10_000)])
    ids = np.arange(n, dtype=np.uint64)
    qs = [gen.rows(B, 10_000_000 + i) for i in range(4)]
    ivf = fv.IVFIndex(ctx, n_clusters=args.nlist, n_probe=32, train_size=100_000, max_iterations=25, seed=7)
    ivf.train(x[np.random.Generator(np.random.Philox(key=5)).choice(n, 100_000, replace=False)])
    ivf.batch_insert(ids, x)
    exact = exact_ground_truth(fv, ctx, x, ids, qs, k)
    qd = [ctx.upload(q) for q in qs]
    oi, od, oc = ctx.alloc(B * k * 8), ctx.alloc(B * k * 4), ctx.alloc(B * 4)
    rows = []
    for p in (8, 16, 32, 64):
        rec = float(np.mean([recall_at_k(*(lambda r_: (r_.ids, r_.counts))(ivf.search(qs[i], k, p)), exact[i], k) for i in range(4)]))
        h = ivf._dev()
        for _ in range(3):
            ctx.check(ctx.lib.fvdb_ivf_search_dev(h, qd[0], B, k, p, oi, od, oc, None))
        ctx.synchronize()
        ctx.timer_start()
        R = 20
        for j in range(R):
            ctx.check(ctx.lib.fvdb_ivf_search_dev(h, qd[j % 4], B, k, p, oi, od, oc, None))
        ms = ctx.timer_stop_ms() / R
        rows.append({"nprobe": p, "recall_at_10": round(rec, 4), "ms_per_batch": round(ms, 4), "queries_per_s": round(B / ms * 1e3, 1)})
        log(f"supplementary isotropic IVF nprobe={p}: recall {rec:.4f}, {ms:.3f} ms/batch")
    out["isotropic_384_ivf_700k"] = {"generator": "SURVEY 8d: 4096 comps, means N(0,I_384), sigma 0.35", "nlist": args.nlist,
                                     "batch": B, "one_batch_at_a_time": rows,
                                     "scan_fallbacks": int(ivf_scan_fallbacks(ctx, ivf))}
    del ivf, x
    # ---- (2) sequential-insert hybrid, 100K ----
    gen = Generator(d=d, latent=args.latent, spread=args.spread)
    n = 100_000
    x = np.concatenate([gen.rows(10_000, s) for s in range(n // 10_000)])
    ids = np.arange(n, dtype=np.uint64)
    now = 1000 * DAY
    is_recent = np.random.Generator(np.random.Philox(key=99)).random(n) < args.recent_frac
    hyb = fv.HybridIndex(ctx, n_clusters=256, n_probe=32, train_size=50_000, max_iterations=25, ivf_seed=7, hnsw_seed=11)
    hyb.initialize(x[:50_000])
    t0 = time.time()
    hist = ~is_recent
    hyb.ivf().batch_insert(ids[hist], x[hist])  # the lists; then the graph, node by node, the reference's way
    h = hyb.hnsw()
    h.batch_insert(ids[is_recent], x[is_recent])
    build_s = time.time() - t0
    qs = [gen.rows(B, 10_000_000 + i) for i in range(4)]
    exact = exact_ground_truth(fv, ctx, x, ids, qs, k)
    rows = []
    qd = [ctx.upload(q) for q in qs]
    for p, e in ((16, 50), (32, 50), (32, 100), (64, 200)):
        recs = [recall_at_k(*(lambda r_: (r_.ids, r_.counts))(hyb.search(qs[i], k, now=now, hnsw_ef=e, ivf_n_probe=p)), exact[i], k)
                for i in range(4)]
        for _ in range(2):
            hyb.search_dev(qd[0], B, k, now=now, hnsw_ef=e, ivf_n_probe=p, dim=d)
        t1 = time.perf_counter()
        R = 10
        for j in range(R):
            hyb.search_dev(qd[j % 4], B, k, now=now, hnsw_ef=e, ivf_n_probe=p, dim=d)
        ms = (time.perf_counter() - t1) / R * 1e3
        rows.append({"nprobe": p, "ef": e, "recall_at_10": round(float(np.mean(recs)), 4), "ms_per_batch": round(ms, 4),
                     "queries_per_s": round(B / ms * 1e3, 1)})
        log(f"supplementary sequential-insert hybrid nprobe={p} ef={e}: recall {np.mean(recs):.4f}, {ms:.3f} ms/batch")
    out["hybrid_100k_sequential_insert"] = {"hnsw_nodes": int(is_recent.sum()), "build_s": round(build_s, 1),
                                            "hnsw_build": "HNSWIndex::insert one node at a time (src/hnsw/core.rs:226-378), "
                                                          "every hop's candidates scored on the GPU",
                                            "one_batch_at_a_time": rows}
    return out


def ivf_scan_fallbacks(ctx, ivf):
    import ctypes as C
    v = C.c_uint64(0)
    ctx.lib.fvdb_ivf_scan_fallbacks(ivf._dev(), C.byref(v))
    return v.value


if __name__ == "__main__":
    main()
