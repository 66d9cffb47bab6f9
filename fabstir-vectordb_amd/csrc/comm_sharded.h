// comm_sharded.h — multi-GPU execution of the IVF search inside the C ABI: one process (or host
// thread) per GPU, inverted lists owned by ranks, RCCL over xGMI for the two exchange steps.  Included at the end of
// fvdb_hip.cpp (same translation unit: it drives the single-GPU stages defined there).
//
// The reference has no distributed execution; what is preserved is the result: keys (distance bits << 32 | global scan
// position) are unique across ranks, so the world-way merge by key is exactly the single-index answer.
//
//   weak mode   every rank brings its own B queries per step (global batch world*B)
//                 coarse(own B) -> all-gather {queries, probe lists} -> scan the lists this rank owns for all world*B
//                 queries -> all-to-all of the partial (key, id) lists, B*k*16 bytes per pair -> world-way merge of
//                 the rank's own B queries.
//   strong mode every rank holds the same B queries (global batch fixed at B)
//                 coarse(all B; redundant, 0.15 ms, cheaper than a collective) -> scan own lists -> all-to-all of the
//                 partials, slice p (ceil(B/world) queries) goes to rank p -> merge: rank r ends with the results of
//                 slice r.
// Everything is enqueued on the slot's stream at `begin`; there is no host synchronisation and no host copy between
// the stages.  RCCL is loaded with dlopen at the first communicator, so a single-GPU process never maps it.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool load() {
    if (handle) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) {
      err = std::string("librccl.so not found: ") + dlerror();
      return false;
    }
#define FVDB_RCCL_SYM(field, sym)                                \
  field = (decltype(field))dlsym(handle, sym);                   \
  if (!field) {                                                  \
    err = std::string("RCCL symbol missing: ") + sym;            \
    return false;                                                \
  }
    FVDB_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    FVDB_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    FVDB_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    FVDB_RCCL_SYM(AllGather, "ncclAllGather");
    FVDB_RCCL_SYM(Send, "ncclSend");
    FVDB_RCCL_SYM(Recv, "ncclRecv");
    FVDB_RCCL_SYM(GroupStart, "ncclGroupStart");
    FVDB_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    FVDB_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef FVDB_RCCL_SYM
    return true;
  }
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

struct Xfer {
  const void* send;
  void* recv;
  size_t bytes;  // all-gather: per rank; all-to-all: per (rank, peer) pair
};

}  // namespace

struct fvdb_comm {
  fvdb_ctx* ctx = nullptr;
  int world = 1, rank = 0;
  ncclComm_t nccl = nullptr;        // RCCL transport
  fvdb_exchange_fn fn = nullptr;    // hosted transport (tests / rehearsal): the caller moves host buffers
  void* user = nullptr;
#ifdef FVDB_DEV_TOOLS
  bool loopback = false;            // capacity planning (dev builds only): exchanges are device copies of this rank's own blocks
#else
  static constexpr bool loopback = false;
#endif
  std::mutex mu;                    // one collective at a time per communicator, same order on every rank
  hipEvent_t last = nullptr;        // end of the previous collective of this communicator, on whichever stream it ran
  bool have_last = false;
  HBuf h_send, h_recv;
};

#define RCCLCHK(ctx, call)                                                                 \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      (ctx)->set_err(std::string(#call) + ": " + g_rccl.GetErrorString(r_));               \
      return FVDB_E_RCCL;                                                                  \
    }                                                                                      \
  } while (0)

namespace {

enum { XCHG_ALL_GATHER = 0, XCHG_ALL_TO_ALL = 1 };

// `n` transfers of one kind as ONE step on `on`'s stream.  RCCL: a single group (one fused launch).  Hosted: stream
// sync, device -> host, the caller's exchange function, host -> device.
int comm_exchange(fvdb_comm* c, fvdb_ctx* on, int op, const Xfer* x, int n) {
  std::lock_guard<std::mutex> lk(c->mu);
  const size_t W = (size_t)c->world;
  if (c->nccl) {
    // steps of different slots run on different streams; the collectives of ONE communicator must not overlap on the
    // device either, so each waits (on the device, not on the host) for the previous one to finish
    if (!c->last) HIPCHK(on, hipEventCreateWithFlags(&c->last, hipEventDisableTiming));
    if (c->have_last) HIPCHK(on, hipStreamWaitEvent(on->stream, c->last, 0));
    RCCLCHK(on, g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
      if (op == XCHG_ALL_GATHER) {
        RCCLCHK(on, g_rccl.AllGather(x[i].send, x[i].recv, x[i].bytes, ncclChar, c->nccl, on->stream));
      } else {
        for (size_t p = 0; p < W; ++p) {
          RCCLCHK(on, g_rccl.Send((const char*)x[i].send + p * x[i].bytes, x[i].bytes, ncclChar, (int)p, c->nccl, on->stream));
          RCCLCHK(on, g_rccl.Recv((char*)x[i].recv + p * x[i].bytes, x[i].bytes, ncclChar, (int)p, c->nccl, on->stream));
        }
      }
    }
    RCCLCHK(on, g_rccl.GroupEnd());
    HIPCHK(on, hipEventRecord(c->last, on->stream));
    c->have_last = true;
    return FVDB_OK;
  }
#ifdef FVDB_DEV_TOOLS
  if (c->loopback) {
    for (int i = 0; i < n; ++i)
      for (size_t p = 0; p < W; ++p) {
        const char* src = (const char*)x[i].send + (op == XCHG_ALL_GATHER ? 0 : (size_t)c->rank * x[i].bytes);
        HIPCHK(on, hipMemcpyAsync((char*)x[i].recv + p * x[i].bytes, src, x[i].bytes, hipMemcpyDeviceToDevice, on->stream));
      }
    return FVDB_OK;
  }
#endif
  for (int i = 0; i < n; ++i) {
    const size_t sb = op == XCHG_ALL_GATHER ? x[i].bytes : W * x[i].bytes, rb = W * x[i].bytes;
    HIPCHK(on, c->h_send.ensure(sb));
    HIPCHK(on, c->h_recv.ensure(rb));
    HIPCHK(on, hipMemcpyAsync(c->h_send.p, x[i].send, sb, hipMemcpyDeviceToHost, on->stream));
    HIPCHK(on, hipStreamSynchronize(on->stream));
    if (c->fn(c->user, op, c->h_send.p, c->h_recv.p, x[i].bytes) != 0) FAIL(on, FVDB_E_RCCL, "hosted exchange failed");
    HIPCHK(on, hipMemcpyAsync(x[i].recv, c->h_recv.p, rb, hipMemcpyHostToDevice, on->stream));
    HIPCHK(on, hipStreamSynchronize(on->stream));  // the pinned blocks are reused by the next transfer
  }
  return FVDB_OK;
}

}  // namespace

struct fvdb_sharded {
  fvdb_ivf* ivf = nullptr;
  fvdb_comm* comm = nullptr;
  struct Slot {
    DBuf probes, q_all, probes_all, keys, ids, dist, cnt, gk, gi, u_own, u_all, thr;
    size_t keys_zeroed = 0;
  } slot[fvdb_ivf::kSlots];
};

extern "C" {

int fvdb_comm_unique_id(void* out128) {
  if (!out128) return FVDB_E_INVALID;
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.load()) return FVDB_E_RCCL;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return FVDB_E_RCCL;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(out128, &id, 128);
  return FVDB_OK;
}

int fvdb_comm_create(fvdb_ctx* ctx, const void* id128, int world, int rank, fvdb_comm** out) {
  if (!ctx || !out) return FVDB_E_INVALID;
  *out = nullptr;
  if (!id128 || world < 1 || rank < 0 || rank >= world) FAIL(ctx, FVDB_E_INVALID, "bad communicator shape");
  {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.load()) FAIL(ctx, FVDB_E_RCCL, g_rccl.err);
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  fvdb_comm* c = new (std::nothrow) fvdb_comm();
  if (!c) return FVDB_E_OOM;
  c->ctx = ctx;
  c->world = world;
  c->rank = rank;
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) {
    ctx->set_err(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    delete c;
    return FVDB_E_RCCL;
  }
  *out = c;
  return FVDB_OK;
}

int fvdb_comm_create_hosted(fvdb_ctx* ctx, int world, int rank, fvdb_exchange_fn fn, void* user, fvdb_comm** out) {
  if (!ctx || !out) return FVDB_E_INVALID;
  *out = nullptr;
  if (!fn || world < 1 || rank < 0 || rank >= world) FAIL(ctx, FVDB_E_INVALID, "bad communicator shape");
  fvdb_comm* c = new (std::nothrow) fvdb_comm();
  if (!c) return FVDB_E_OOM;
  c->ctx = ctx;
  c->world = world;
  c->rank = rank;
  c->fn = fn;
  c->user = user;
  *out = c;
  return FVDB_OK;
}

#ifdef FVDB_DEV_TOOLS
// Development builds only (make dev -> lib_dev/, include/fvdb_dev.h): not part of the product library.
int fvdb_comm_create_loopback(fvdb_ctx* ctx, int world, int rank, fvdb_comm** out) {
  if (!ctx || !out) return FVDB_E_INVALID;
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) FAIL(ctx, FVDB_E_INVALID, "bad communicator shape");
  fvdb_comm* c = new (std::nothrow) fvdb_comm();
  if (!c) return FVDB_E_OOM;
  c->ctx = ctx;
  c->world = world;
  c->rank = rank;
  c->loopback = true;
  *out = c;
  return FVDB_OK;
}
#endif

void fvdb_comm_destroy(fvdb_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  if (c->last) (void)hipEventDestroy(c->last);
  c->h_send.release();
  c->h_recv.release();
  delete c;
}
int fvdb_comm_rank(fvdb_comm* c) { return c ? c->rank : -1; }
int fvdb_comm_world(fvdb_comm* c) { return c ? c->world : 0; }

// Stream-ordered collectives on a context's stream, for hosts that exchange their own device buffers (and for the
// tests of the transports): recv holds world blocks of `bytes`.
int fvdb_comm_all_gather_dev(fvdb_comm* c, fvdb_ctx* on, const void* send_dev, void* recv_dev, size_t bytes) {
  if (!c) return FVDB_E_INVALID;
  if (!on) on = c->ctx;
  HIPCHK(on, hipSetDevice(on->device));
  const Xfer x{send_dev, recv_dev, bytes};
  return comm_exchange(c, on, XCHG_ALL_GATHER, &x, 1);
}
int fvdb_comm_all_to_all_dev(fvdb_comm* c, fvdb_ctx* on, const void* send_dev, void* recv_dev, size_t bytes) {
  if (!c) return FVDB_E_INVALID;
  if (!on) on = c->ctx;
  HIPCHK(on, hipSetDevice(on->device));
  const Xfer x{send_dev, recv_dev, bytes};
  return comm_exchange(c, on, XCHG_ALL_TO_ALL, &x, 1);
}

int fvdb_sharded_create(fvdb_ivf* ivf, fvdb_comm* comm, fvdb_sharded** out) {
  if (!ivf || !comm || !out) return FVDB_E_INVALID;
  *out = nullptr;
  if (comm->ctx->device != ivf->ctx->device) FAIL(ivf->ctx, FVDB_E_INVALID, "communicator of another device");
  fvdb_sharded* s = new (std::nothrow) fvdb_sharded();
  if (!s) return FVDB_E_OOM;
  s->ivf = ivf;
  s->comm = comm;
  *out = s;
  return FVDB_OK;
}

void fvdb_sharded_destroy(fvdb_sharded* s) {
  if (!s) return;
  (void)hipSetDevice(s->ivf->ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& sl : s->slot)
    for (DBuf* b : {&sl.probes, &sl.q_all, &sl.probes_all, &sl.keys, &sl.ids, &sl.dist, &sl.cnt, &sl.gk, &sl.gi, &sl.u_own, &sl.u_all,
                    &sl.thr})
      b->release();
  delete s;
}

uint32_t fvdb_sharded_out_rows(fvdb_sharded* s, uint32_t B, int mode) {
  if (!s || B == 0) return 0;
  return mode == FVDB_SHARD_STRONG ? cdiv(B, (uint64_t)s->comm->world) : B;
}

int fvdb_ivf_search_sharded_begin(fvdb_sharded* s, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t k,
                                  uint32_t nprobe, int mode, uint64_t* out_ids_dev, float* out_dist_dev,
                                  uint32_t* out_counts_dev) {
  if (!s) return FVDB_E_INVALID;
  fvdb_ivf* ivf = s->ivf;
  fvdb_comm* c = s->comm;
  fvdb_ctx* ctx = on ? on : ivf->ctx;
  if (slot >= fvdb_ivf::kSlots) FAIL(ctx, FVDB_E_INVALID, "slot out of range");
  if (mode != FVDB_SHARD_WEAK && mode != FVDB_SHARD_STRONG) FAIL(ctx, FVDB_E_INVALID, "unknown sharding mode");
  if (!q_dev || !out_ids_dev || !out_dist_dev || !out_counts_dev) FAIL(ctx, FVDB_E_INVALID, "null buffer");
  if (!ivf->trained) FAIL(ctx, FVDB_E_NOT_TRAINED, "index not trained");
  if (k == 0 || k > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "k must be in 1..FVDB_MAX_K");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t W = (uint32_t)c->world, d = ivf->d;
  const uint32_t np = std::min(nprobe, ivf->nlist);
  if (np == 0 || np > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "nprobe must be in 1..FVDB_MAX_K");
  const bool weak = mode == FVDB_SHARD_WEAK;
  const uint32_t Bo = weak ? B : cdiv(B, W);     // queries whose final result this rank produces
  const uint32_t Bq = weak ? W * B : B;          // queries scanned against the lists this rank owns
  const uint32_t Bs = W * Bo;                    // rows of the partial arrays (>= Bq: strong mode pads the last slice)
  if ((uint64_t)Bs * k >= (1ull << 31)) FAIL(ctx, FVDB_E_UNSUPPORTED, "batch too large for one sharded step");
  fvdb_sharded::Slot& sl = s->slot[slot];
  HIPCHK(ctx, sl.probes.ensure((size_t)B * np * 4));
  HIPCHK(ctx, sl.keys.ensure((size_t)Bs * k * 8));
  HIPCHK(ctx, sl.ids.ensure((size_t)Bs * k * 8));
  HIPCHK(ctx, sl.dist.ensure((size_t)Bs * k * 4));
  HIPCHK(ctx, sl.cnt.ensure((size_t)Bs * 4));
  HIPCHK(ctx, sl.gk.ensure((size_t)Bs * k * 8));
  HIPCHK(ctx, sl.gi.ensure((size_t)Bs * k * 8));
  if (Bs > Bq) {  // padding rows travel through the exchange: "no result" keys, ignored by the merge
    HIPCHK(ctx, hipMemsetAsync(sl.keys.as<uint64_t>() + (size_t)Bq * k, 0xFF, (size_t)(Bs - Bq) * k * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(sl.ids.as<uint64_t>() + (size_t)Bq * k, 0xFF, (size_t)(Bs - Bq) * k * 8, ctx->stream));
  }
  // 1. centroid ranking.  WEAK: the rank's own B queries.  STRONG: the batch is the same on every rank, so each rank
  //    ranks the centroids for ITS slice of it only (the slice whose results it will produce) and the probe lists are
  //    all-gathered (Bo * nprobe * 4 B per rank) — 1/W of the ranking work per rank instead of all of it on every rank
  //    (FVDB_STRONG_SPLIT=0: every rank ranks every query, no exchange; for A/B runs)
  static const bool strong_split = !(getenv("FVDB_STRONG_SPLIT") && atoi(getenv("FVDB_STRONG_SPLIT")) == 0);
  int rc = FVDB_OK;
  const float* q_scan = q_dev;
  const uint32_t* probes_scan = sl.probes.as<uint32_t>();
  if (!weak && W > 1 && strong_split) {
    const uint32_t lo = std::min(B, (uint32_t)c->rank * Bo), mine = std::min(Bo, B - lo);
    HIPCHK(ctx, sl.probes.ensure((size_t)Bo * np * 4));
    HIPCHK(ctx, sl.probes_all.ensure((size_t)W * Bo * np * 4));
    if (mine < Bo)  // the last slices may be short or empty: "no list" entries, skipped by the plan kernels
      HIPCHK(ctx, hipMemsetAsync(sl.probes.as<uint32_t>() + (size_t)mine * np, 0xFF, (size_t)(Bo - mine) * np * 4, ctx->stream));
    if (mine) {
      rc = fvdb_ivf_coarse_dev_slot(ivf, on, slot, q_dev + (size_t)lo * d, mine, nprobe, sl.probes.as<uint32_t>());
      if (rc) return rc;
    }
    const Xfer x[1] = {{sl.probes.p, sl.probes_all.p, (size_t)Bo * np * 4}};
    rc = comm_exchange(c, ctx, XCHG_ALL_GATHER, x, 1);
    if (rc) return rc;
    probes_scan = sl.probes_all.as<uint32_t>();  // row q = slice * Bo + i: the global query index
  } else {
    rc = fvdb_ivf_coarse_dev_slot(ivf, on, slot, q_dev, B, nprobe, sl.probes.as<uint32_t>());
    if (rc) return rc;
  }
  if (weak && W > 1) {  // exchange 1: every rank needs every query, with the probe list it came with
    HIPCHK(ctx, sl.q_all.ensure((size_t)Bq * d * 4));
    HIPCHK(ctx, sl.probes_all.ensure((size_t)Bq * np * 4));
    const Xfer x[2] = {{q_dev, sl.q_all.p, (size_t)B * d * 4}, {sl.probes.p, sl.probes_all.p, (size_t)B * np * 4}};
    rc = comm_exchange(c, ctx, XCHG_ALL_GATHER, x, 2);
    if (rc) return rc;
    q_scan = sl.q_all.as<float>();
    probes_scan = sl.probes_all.as<uint32_t>();
  }
  // 2. the lists this rank owns, for all of them: partial top-k with global selection keys.  The filter's
  //    threshold of a query comes from ONE list — by the logical index's sizes, the same choice on every rank — so only
  //    the rank owning that list computes it; an all-gather of the ranks' arrays (+inf = not mine) and a minimum give
  //    every rank every threshold.  Without this each rank would sample rows for all Bq queries: W times the work.
  if (thr_share_ok(ivf, Bq, k, np)) {
    HIPCHK(ctx, sl.u_own.ensure((size_t)Bq * 4));
    HIPCHK(ctx, sl.thr.ensure((size_t)Bq * 4));
    rc = shared_thresholds_slot(ivf, on, slot, q_scan, probes_scan, Bq, k, np, sl.u_own.as<float>());
    if (rc) return rc;
    const float* u_all = sl.u_own.as<float>();
    if (W > 1) {
      HIPCHK(ctx, sl.u_all.ensure((size_t)W * Bq * 4));
      const Xfer x[1] = {{sl.u_own.p, sl.u_all.p, (size_t)Bq * 4}};
      rc = comm_exchange(c, ctx, XCHG_ALL_GATHER, x, 1);
      if (rc) return rc;
      u_all = sl.u_all.as<float>();
    }
    rc = thr_combine_slot(ivf, on, slot, u_all, W, Bq, sl.thr.as<float>(), c->loopback);
    if (rc) return rc;
    rc = search_probes_thr_slot(ivf, on, slot, q_scan, probes_scan, sl.thr.as<float>(), Bq, k, nprobe,
                                sl.ids.as<uint64_t>(), sl.dist.as<float>(), sl.cnt.as<uint32_t>(), sl.keys.as<uint64_t>());
  } else {
    rc = fvdb_ivf_search_probes_dev_slot(ivf, on, slot, q_scan, probes_scan, Bq, k, nprobe, sl.ids.as<uint64_t>(),
                                         sl.dist.as<float>(), sl.cnt.as<uint32_t>(), sl.keys.as<uint64_t>());
  }
  if (rc) return rc;
  // 3. exchange 2: the partials of rank p's queries go to rank p
  const uint64_t* gk = sl.keys.as<uint64_t>();
  const uint64_t* gi = sl.ids.as<uint64_t>();
  if (W > 1) {
    const Xfer x[2] = {{sl.keys.p, sl.gk.p, (size_t)Bo * k * 8}, {sl.ids.p, sl.gi.p, (size_t)Bo * k * 8}};
    rc = comm_exchange(c, ctx, XCHG_ALL_TO_ALL, x, 2);
    if (rc) return rc;
    gk = sl.gk.as<uint64_t>();
    gi = sl.gi.as<uint64_t>();
  }
  // 4. world-way merge by key: exactly the single-index result for these Bo queries
  return fvdb_merge_keys_dev(ctx, gk, gi, W, Bo, k, out_ids_dev, out_dist_dev, out_counts_dev);
}

int fvdb_ivf_search_sharded_end(fvdb_sharded* s, fvdb_ctx* on, uint32_t slot) {
  if (!s) return FVDB_E_INVALID;
  fvdb_ctx* ctx = on ? on : s->ivf->ctx;
  if (slot >= fvdb_ivf::kSlots) FAIL(ctx, FVDB_E_INVALID, "slot out of range");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

}  // extern "C"
