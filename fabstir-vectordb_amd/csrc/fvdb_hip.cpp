// fvdb_hip.cpp — C ABI (include/fvdb.h) over the HIP kernels.  gfx950 only; compiled with
//   hipcc -x hip -O3 -ffp-contract=off --offload-arch=gfx950
// Host side here is plumbing: memory, launch order, list bookkeeping.  All arithmetic on
// vectors happens in the kernels; there is no CPU fallback anywhere in this library.
#include "fvdb_internal.h"

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "common.h"
#include "kernels_misc.h"
#include "kernels_scan.h"
#include "kernels_coarse.h"
#include "kernels_mfma.h"
#include "kernels_mfma_wg.h"
#include "kernels_util.h"

using namespace fvdb;


// A pool of 64-row blocks in HBM (layout: common.h PoolView).
struct Pool {
  uint32_t d4 = 0;      // 4-dim chunks per row (of the padded dimension)
  uint32_t esize = 4;   // bytes per stored element: 4 = f32, 2 = fp16
  uint32_t cap_blocks = 0, used_blocks = 0;
  void* data = nullptr;
  uint64_t* ids = nullptr;
  uint64_t* valid = nullptr;
  float* norms = nullptr;  // |x|^2 per row (matrix-core filter, kernels_mfma.h)
  bool mirror = false;     // f32 pools: keep an fp16 (round-to-nearest) copy of the rows for the matrix-core filter
  void* half = nullptr;    // [blocks][d8][64] 8-half chunks, same lane = row layout
  // f32 pools with the mirror also keep the rows ROW-MAJOR ([blocks * 64][dpad] f32, bit copies): the select stage
  // scores some twenty single rows per query, and in the blocked layout every 16 bytes of a row sit in a different
  // cache line (12 KB fetched per 1.5 KB row)
  float* rm = nullptr;
  size_t block_bytes() const { return (size_t)d4 * 4 * esize * 64; }
  PoolView view() const { return PoolView{data, ids, valid, d4}; }
  void release() {
    if (norms) (void)hipFree(norms);
    norms = nullptr;
    if (half) (void)hipFree(half);
    half = nullptr;
    if (rm) (void)hipFree(rm);
    rm = nullptr;
    if (data) (void)hipFree(data);
    if (ids) (void)hipFree(ids);
    if (valid) (void)hipFree(valid);
    data = nullptr;
    ids = nullptr;
    valid = nullptr;
    cap_blocks = used_blocks = 0;
  }
  // grow to at least `blocks` capacity, preserving contents (stream-ordered copies)
  int reserve(fvdb_ctx* ctx, uint32_t blocks) {
    if (blocks <= cap_blocks) return FVDB_OK;
    uint32_t ncap = std::max<uint32_t>(blocks, cap_blocks + cap_blocks / 2 + 16);
    void* nd = nullptr;
    uint64_t* ni = nullptr;
    uint64_t* nv = nullptr;
    float* nn = nullptr;
    HIPCHK(ctx, hipMalloc(&nd, (size_t)ncap * block_bytes()));
    // rows never written (the tail of each list's last block) must be finite: they share MFMA instructions with
    // live rows, and 0 x NaN would poison those
    HIPCHK(ctx, hipMemsetAsync(nd, 0, (size_t)ncap * block_bytes(), ctx->stream));
    float* nr = nullptr;
    if (mirror && esize == 4) {
      HIPCHK(ctx, hipMalloc((void**)&nr, (size_t)ncap * block_bytes()));
      HIPCHK(ctx, hipMemsetAsync(nr, 0, (size_t)ncap * block_bytes(), ctx->stream));
      if (used_blocks && rm)
        HIPCHK(ctx, hipMemcpyAsync(nr, rm, (size_t)used_blocks * block_bytes(), hipMemcpyDeviceToDevice, ctx->stream));
    }
    void* nh = nullptr;
    if (mirror) {
      HIPCHK(ctx, hipMalloc(&nh, (size_t)ncap * block_bytes() / 2));
      HIPCHK(ctx, hipMemsetAsync(nh, 0, (size_t)ncap * block_bytes() / 2, ctx->stream));
      if (used_blocks && half)
        HIPCHK(ctx, hipMemcpyAsync(nh, half, (size_t)used_blocks * block_bytes() / 2, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipMalloc(&nn, (size_t)ncap * 64 * sizeof(float)));
    HIPCHK(ctx, hipMemsetAsync(nn, 0, (size_t)ncap * 64 * sizeof(float), ctx->stream));
    HIPCHK(ctx, hipMalloc(&ni, (size_t)ncap * 64 * sizeof(uint64_t)));
    HIPCHK(ctx, hipMalloc(&nv, (size_t)ncap * sizeof(uint64_t)));
    HIPCHK(ctx, hipMemsetAsync(nv, 0, (size_t)ncap * sizeof(uint64_t), ctx->stream));
    if (used_blocks) {
      HIPCHK(ctx, hipMemcpyAsync(nd, data, (size_t)used_blocks * block_bytes(), hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(ni, ids, (size_t)used_blocks * 64 * sizeof(uint64_t), hipMemcpyDeviceToDevice,
                                 ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(nv, valid, (size_t)used_blocks * sizeof(uint64_t), hipMemcpyDeviceToDevice,
                                 ctx->stream));
      HIPCHK(ctx, hipMemcpyAsync(nn, norms, (size_t)used_blocks * 64 * sizeof(float), hipMemcpyDeviceToDevice,
                                 ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (data) (void)hipFree(data);
    if (ids) (void)hipFree(ids);
    if (valid) (void)hipFree(valid);
    if (norms) (void)hipFree(norms);
    if (half) (void)hipFree(half);
    half = nh;
    if (rm) (void)hipFree(rm);
    rm = nr;
    norms = nn;
    data = nd;
    ids = ni;
    valid = nv;
    cap_blocks = ncap;
    return FVDB_OK;
  }
};

// Per-search scratch of an IVF index.  Set 0 lives in the index itself (also used by the mutating entry points);
// sets 1..7 serve the other explicit slots of the *_slot entry points; a further pool of leased sets, each with a
// stream of its own, serves the blocking host-pointer searches, so any number of host threads may search one index
// at once (reference: searches hold a read guard, bindings/node/src/session.rs:253).  A search never modifies the
// index object: the scratch set and the stream it runs on are passed down explicitly (Env).
struct IvfScratch {
  std::mutex enq;  // held while one search's launches are enqueued: searches sharing a set are ordered by the stream
  bool pend_filter = false;
  bool pending_profile = false, pend_coarse = false, pend_fine = false, collecting = false;
  DBuf s_qnorm, s_A;
  DBuf s_qh, s_qn2, s_thr, s_tA, s_pa, s_surv, s_scnt, s_fail, s_mslots, s_sdist, s_probes2;
  DBuf s_q, s_cpart, s_probes, s_cnt, s_fill, s_eoff, s_ioff, s_entries, s_part, s_scalars, s_ceoff, s_cioff;
  DBuf s_in, s_slots, s_ids, s_clusters, s_out_ids, s_out_dist, s_out_cnt, s_cdist;
  hipEvent_t sev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void release_all() {
    DBuf* bufs[] = {&s_qnorm, &s_A, &s_qh, &s_qn2, &s_thr, &s_tA, &s_pa, &s_surv, &s_scnt, &s_fail, &s_mslots, &s_sdist, &s_probes2,
                    &s_q, &s_cpart, &s_probes, &s_cnt, &s_fill, &s_eoff, &s_ioff, &s_entries, &s_part, &s_scalars,
                    &s_ceoff, &s_cioff, &s_in, &s_slots, &s_ids, &s_clusters, &s_out_ids, &s_out_dist, &s_out_cnt, &s_cdist};
    for (DBuf* b : bufs) b->release();
    for (auto& e : sev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
};

struct fvdb_ivf : IvfScratch {
  static constexpr uint32_t kSlots = 16;
  IvfScratch spare[kSlots - 1];  // slots 1..15
  // leased sets for blocking searches called from several host threads: each has its own stream (a private context)
  static constexpr uint32_t kLeases = 8;
  IvfScratch lease_set[kLeases];
  fvdb_ctx* lease_ctx[kLeases] = {};
  uint32_t lease_busy = 0;  // bit i: set i is out (under mu)
  std::mutex mu;            // list table upload, lease bookkeeping, AUTO-mode counters
  std::condition_variable lease_cv;
  std::atomic<IvfScratch*> last_set{nullptr};  // scratch set of the most recent search (diagnostic entry points)
  std::atomic<fvdb_ctx*> last_ctx{nullptr};    // and the context (stream) it ran on
  fvdb_ctx* ctx = nullptr;
  uint32_t d = 0, dpad = 0, d4 = 0, nlist = 0;
  bool trained = false;
  bool f16 = false;  // inverted-list rows stored as fp16 (centroids and queries stay f32)

  // centroid table: row-major copy (host + device) and a blocked pool scanned as "list 0"
  std::vector<float> h_centroids;
  DBuf d_centroids_rm;   // [nlist][d]
  DBuf d_cent_pad;       // [nlist][dpad] zero padded (only when d != dpad)
  DBuf d_cnorm, d_cnmax; // |c|^2 per centroid, max |c|^2 (matrix-core coarse stage)
  DBuf s_fallbacks;
  int coarse_mode = 0;   // 0 = matrix cores + exact verification when applicable, 1 = exact scan only
  int scan_mode = 0;     // same choice for the inverted-list scan
  // AUTO scan mode watches its own hit rate: the rescan counter is copied to pinned host memory behind every
  // matrix-core batch (no sync); when too many queries of the recent batches needed the exact rescan (data the
  // filter cannot separate, e.g. no cluster structure), the next batches go straight to the exact scan
  HBuf h_fb;                 // pinned copy of s_fallbacks[0..7]
  uint64_t mfma_q = 0;       // queries sent down the matrix-core path
  uint64_t fb_seen = 0, q_seen = 0;  // counter / queries at the last decision
  uint32_t exact_batches_left = 0;   // > 0: AUTO is backing off to the exact scan
  uint32_t backoff_len = 0;
  // the second filter pass for queries whose survivors outgrew the buffer (refine_threshold_kernel) costs five small
  // launches per batch: AUTO enqueues them only while such queries have been seen recently (counters [2] and [6])
  uint64_t overflow_seen = 0, overflow_q = 0;
  uint32_t refine_batches_left = 0;
  DBuf d_xmax;           // max |x|^2 over the rows ever added (float bits)
  Pool cpool;
  DBuf c_off, c_blocks, c_glob;  // single-list table for the centroid pool

  // inverted lists: paged blocks
  Pool pool;
  std::vector<std::vector<uint32_t>> list_blocks;  // per list: pool block indices
  std::vector<uint32_t> list_len;                  // rows per list (including soft-deleted)
  uint64_t total_rows = 0;
  uint32_t max_list_blocks = 0;
  bool table_dirty = true;
  DBuf t_off, t_blocks, t_glob, t_len;  // device list table, logical (global) block counts, rows per list
  std::vector<uint32_t> glob_blocks_host;  // empty => local sizes
  bool glob_set = false;

  // per-search scratch
  fvdb_search_stats last_stats{};
  // coarse scan, coarse merge, plan, fine scan, fine merge, [5] the matrix-core filter kernel alone (inside fine scan)
  float stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t stage_calls = 0;
};

namespace {
// what a search runs with: the stream (context) its launches go to and the scratch set they may touch
struct Env {
  fvdb_ctx* ctx;
  IvfScratch* S;
  const float* given_thr = nullptr;  // sharded search: filter thresholds already agreed between the ranks ([B], device)
};
inline IvfScratch& slot_scratch(fvdb_ivf* ivf, uint32_t slot) { return slot == 0 ? *ivf : ivf->spare[slot - 1]; }
}  // namespace


// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
namespace {

struct ScanLaunch {
  PoolView pool;
  const uint32_t* list_off;
  const uint32_t* list_blocks;
  uint32_t nlist;
  const uint32_t* entry_off;
  const uint32_t* item_off;
  const uint2* entries;
  const uint32_t* n_items;
  uint32_t* head;
  const float* queries;
  uint32_t dpad, segb, k, nprobe, maxsegs;
  uint2* part;
  uint32_t max_items = 0;  // host-side upper bound on work items (0 = unknown): sizes the persistent grid
  bool f16 = false;        // rows of `pool` are fp16
};

inline int kr_for(uint32_t k) { return k <= 64 ? 1 : (k <= 128 ? 2 : 4); }
inline uint32_t q_for(uint32_t k) { return k <= 64 ? 16u : (k <= 128 ? 8u : 4u); }

template <int Q, int KR, int ROLE, int ST>
void launch_scan_t(fvdb_ctx* ctx, const ScanLaunch& s) {
  static const uint32_t wgs_per_cu = getenv("FVDB_SCAN_WGS_PER_CU") ? (uint32_t)atoi(getenv("FVDB_SCAN_WGS_PER_CU")) : 8u;
  uint32_t grid = (uint32_t)ctx->num_cus * wgs_per_cu;  // more than fit: surplus workgroups find the queue empty
  if (s.max_items) grid = std::min(grid, std::max<uint32_t>(1u, (s.max_items + 3) / 4));  // 4 waves per workgroup
  hipLaunchKernelGGL((scan_topk_kernel<Q, KR, ROLE, ST>), dim3(grid), dim3(256), 0, ctx->stream, s.pool.data, s.pool.valid,
                     s.pool.d4, s.list_off, s.list_blocks, s.nlist, s.entry_off, s.item_off, (const u32x2*)s.entries,
                     s.n_items, s.head, s.queries, s.dpad, s.segb, s.k, s.nprobe, s.maxsegs, (u32x2*)s.part);
}

template <int ROLE, int ST>
void launch_scan_rs(fvdb_ctx* ctx, const ScanLaunch& s) {
  switch (kr_for(s.k)) {
    case 1: launch_scan_t<16, 1, ROLE, ST>(ctx, s); break;
    case 2: launch_scan_t<8, 2, ROLE, ST>(ctx, s); break;
    default: launch_scan_t<4, 4, ROLE, ST>(ctx, s); break;
  }
}
template <int ROLE>
void launch_scan_r(fvdb_ctx* ctx, const ScanLaunch& s) {
  if (s.f16) launch_scan_rs<ROLE, 1>(ctx, s); else launch_scan_rs<ROLE, 0>(ctx, s);
}
enum { ROLE_COARSE = 0, ROLE_LIST = 1, ROLE_ALL = 2 };
void launch_scan(fvdb_ctx* ctx, const ScanLaunch& s, int role) {
  if (role == ROLE_COARSE) launch_scan_r<ROLE_COARSE>(ctx, s);
  else if (role == ROLE_LIST) launch_scan_r<ROLE_LIST>(ctx, s);
  else launch_scan_r<ROLE_ALL>(ctx, s);
}

void launch_merge(fvdb_ctx* ctx, const MergeArgs& m) {
  const uint32_t grid = cdiv(m.B, 4);
  switch (kr_for(m.k)) {
    case 1: hipLaunchKernelGGL((merge_topk_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, m); break;
    case 2: hipLaunchKernelGGL((merge_topk_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, m); break;
    default: hipLaunchKernelGGL((merge_topk_kernel<4>), dim3(grid), dim3(256), 0, ctx->stream, m); break;
  }
}

// scalars block layout (uint32): [0]=coarse n_items [1]=coarse head [2]=fine n_items [3]=fine head
//                                 [4..9] = stats (3 x u64)
constexpr size_t kScalarsBytes = 64;

int upload_table(fvdb_ivf* ivf) {
  fvdb_ctx* ctx = ivf->ctx;
  std::lock_guard<std::mutex> lk(ivf->mu);  // concurrent searches after a mutation: one of them uploads
  if (!ivf->table_dirty) return FVDB_OK;
  std::vector<uint32_t> off(ivf->nlist + 1, 0), blocks;
  std::vector<uint32_t> glob(ivf->nlist, 0);
  uint32_t mx = 0;
  for (uint32_t L = 0; L < ivf->nlist; ++L) {
    off[L] = (uint32_t)blocks.size();
    blocks.insert(blocks.end(), ivf->list_blocks[L].begin(), ivf->list_blocks[L].end());
    mx = std::max<uint32_t>(mx, (uint32_t)ivf->list_blocks[L].size());
    glob[L] = ivf->glob_set ? ivf->glob_blocks_host[L] : (uint32_t)ivf->list_blocks[L].size();
  }
  off[ivf->nlist] = (uint32_t)blocks.size();
  ivf->max_list_blocks = mx;
  HIPCHK(ctx, ivf->t_off.ensure(off.size() * 4));
  HIPCHK(ctx, ivf->t_blocks.ensure(std::max<size_t>(blocks.size(), 1) * 4));
  HIPCHK(ctx, ivf->t_glob.ensure(glob.size() * 4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->t_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!blocks.empty())
    HIPCHK(ctx, hipMemcpyAsync(ivf->t_blocks.p, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ivf->t_glob.p, glob.data(), glob.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, ivf->t_len.ensure((size_t)ivf->nlist * 4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->t_len.p, ivf->list_len.data(), (size_t)ivf->nlist * 4, hipMemcpyHostToDevice,
                             ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
  ivf->table_dirty = false;
  return FVDB_OK;
}

// queries as [B][dpad] in HBM: q_dev itself when d is a multiple of 4, else a padded copy
int padded_queries(fvdb_ivf* ivf, const Env& E, const float* q_dev, uint32_t B, const float** out) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  if (ivf->d == ivf->dpad) {
    *out = q_dev;
    return FVDB_OK;
  }
  HIPCHK(ctx, S.s_q.ensure((size_t)B * ivf->dpad * 4));
  const uint64_t tot = (uint64_t)B * ivf->dpad;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, ctx->stream, q_dev, ivf->d, ivf->dpad,
                     (uint64_t)B, S.s_q.as<float>());
  *out = S.s_q.as<float>();
  return FVDB_OK;
}

// Coarse stage: rank the centroid table for B queries, keep kc nearest per query.
// Writes u32 cluster ids to out_probes[B][kc] (probe order) and, optionally, their distances.
int run_coarse(fvdb_ivf* ivf, const Env& E, const float* qpad, uint32_t B, uint32_t kc, uint32_t* out_probes,
               float* out_dist) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  static const bool env_exact = getenv("FVDB_COARSE_EXACT") != nullptr;  // tuning aid
  if (ivf->coarse_mode == 0 && !env_exact && ivf->dpad % 16 == 0 && kc <= 48 && ivf->nlist >= 64 && B > 0 &&
      (uint64_t)B * ivf->nlist < (1ull << 31)) {
    // matrix cores propose 64 candidates per query; the reference's arithmetic decides (kernels_coarse.h)
    const uint32_t nlist = ivf->nlist;
    const float* cpad = ivf->d == ivf->dpad ? ivf->d_centroids_rm.as<float>() : ivf->d_cent_pad.as<float>();
    HIPCHK(ctx, S.s_qnorm.ensure((size_t)B * 4));
    HIPCHK(ctx, S.s_A.ensure((size_t)B * nlist * 4));
    if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[0], ctx->stream);
    hipLaunchKernelGGL(row_sqnorm_wave_kernel, dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, qpad, ivf->dpad, ivf->dpad, B,
                       S.s_qnorm.as<float>());
    const uint32_t waves = cdiv(B, 32) * cdiv(nlist, 64);
    hipLaunchKernelGGL(coarse_gemm_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, ctx->stream, qpad, cpad,
                       S.s_qnorm.as<float>(), ivf->d_cnorm.as<float>(), B, nlist, ivf->dpad, S.s_A.as<float>());
    if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[1], ctx->stream);
    hipLaunchKernelGGL(coarse_select_kernel, dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, S.s_A.as<float>(), qpad, cpad,
                       S.s_qnorm.as<float>(), ivf->d_cnmax.as<float>(), B, nlist, ivf->d, ivf->dpad, 64u, kc,
                       out_probes, out_dist, ivf->s_fallbacks.as<uint32_t>());
    if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[2], ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return FVDB_OK;
  }
  const uint32_t cblocks = ivf->cpool.used_blocks;
  const uint32_t segb = 1, Q = q_for(kc);
  const uint32_t maxsegs = cblocks;
  HIPCHK(ctx, S.s_cpart.ensure((size_t)B * maxsegs * kc * 8));
  HIPCHK(ctx, S.s_entries.ensure((size_t)B * std::max<uint32_t>(kc, 1) * 8));
  HIPCHK(ctx, S.s_ceoff.ensure(16));
  HIPCHK(ctx, S.s_cioff.ensure(16));
  HIPCHK(ctx, S.s_scalars.ensure(kScalarsBytes));
  uint32_t* scal = S.s_scalars.as<uint32_t>();
  hipLaunchKernelGGL(plan_all_kernel, dim3(cdiv(std::max<uint32_t>(B, 1), 256)), dim3(256), 0, ctx->stream, B, cblocks,
                     segb, Q, S.s_ceoff.as<uint32_t>(), S.s_cioff.as<uint32_t>(), S.s_entries.as<uint2>(),
                     scal + 0, scal + 1);
  ScanLaunch s{ivf->cpool.view(), ivf->c_off.as<uint32_t>(), ivf->c_blocks.as<uint32_t>(), 1,
               S.s_ceoff.as<uint32_t>(), S.s_cioff.as<uint32_t>(), S.s_entries.as<uint2>(), scal + 0,
               scal + 1, qpad, ivf->dpad, segb, kc, 1, maxsegs, S.s_cpart.as<uint2>(),
               cdiv(cblocks, segb) * cdiv(B, Q)};
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[0], ctx->stream);
  launch_scan(ctx, s, ROLE_COARSE);
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[1], ctx->stream);
  MergeArgs m{};
  m.pool = ivf->cpool.view();
  m.lists = ListTable{ivf->c_off.as<uint32_t>(), ivf->c_blocks.as<uint32_t>(), 1};
  m.probes = nullptr;
  m.glob_blocks = ivf->c_glob.as<uint32_t>();
  m.part = S.s_cpart.as<uint2>();
  m.B = B;
  m.k = kc;
  m.nprobe = 1;
  m.maxsegs = maxsegs;
  m.segb = segb;
  m.out_probes = out_probes;
  m.out_dist = out_dist;
  launch_merge(ctx, m);
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[2], ctx->stream);
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

uint32_t pick_segb(fvdb_ivf* ivf, uint32_t B, uint32_t nprobe) {
  static const int forced = getenv("FVDB_SEGB") ? atoi(getenv("FVDB_SEGB")) : 0;  // tuning aid
  if (forced > 0) return (uint32_t)forced;
  // enough (segment, group) items to fill 256 CUs x 32 waves, without shredding long lists
  const uint64_t pairs = (uint64_t)B * nprobe;
  if (ivf->max_list_blocks >= 4096) return 16;
  if (pairs >= 4096) return 4;
  if (pairs >= 512) return 2;
  return 1;
}

// Fine stage for B queries whose probes[B][np] are already in HBM: every row scored with the reference's fold.
int run_fine_exact(fvdb_ivf* ivf, const Env& E, const float* qpad, uint32_t B, uint32_t k, uint32_t np,
                   const uint32_t* probes, uint64_t* out_ids, float* out_dist, uint32_t* out_counts, uint64_t* out_keys,
                   int role, bool events = true) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  const uint32_t nlist = ivf->nlist;
  const uint32_t segb = pick_segb(ivf, B, np), Q = q_for(k);
  const uint32_t maxsegs = std::max<uint32_t>(1, cdiv(ivf->max_list_blocks, segb));
  const uint64_t part_elems = (uint64_t)B * np * maxsegs * k;
  if (part_elems >= (1ull << 32)) FAIL(ctx, FVDB_E_UNSUPPORTED, "batch too large for one launch (sub-batch it)");
  HIPCHK(ctx, S.s_cnt.ensure((size_t)nlist * 4));
  HIPCHK(ctx, S.s_fill.ensure((size_t)nlist * 4));
  HIPCHK(ctx, S.s_eoff.ensure((size_t)(nlist + 1) * 4));
  HIPCHK(ctx, S.s_ioff.ensure((size_t)(nlist + 1) * 4));
  HIPCHK(ctx, S.s_entries.ensure((size_t)B * np * 8));
  HIPCHK(ctx, S.s_part.ensure((size_t)part_elems * 8));
  HIPCHK(ctx, S.s_scalars.ensure(kScalarsBytes));
  uint32_t* scal = S.s_scalars.as<uint32_t>();
  const uint32_t n = B * np;
  HIPCHK(ctx, hipMemsetAsync(S.s_cnt.p, 0, (size_t)nlist * 4, ctx->stream));
  hipLaunchKernelGGL(plan_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, probes, n,
                     S.s_cnt.as<uint32_t>(), ivf->t_off.as<uint32_t>());
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, S.s_cnt.as<uint32_t>(),
                     ivf->t_off.as<uint32_t>(), ivf->t_len.as<uint32_t>(), nlist, segb, Q, S.s_eoff.as<uint32_t>(),
                     S.s_ioff.as<uint32_t>(), S.s_fill.as<uint32_t>(), scal + 2, scal + 3,
                     (unsigned long long*)(scal + 4));
  hipLaunchKernelGGL(plan_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, probes, n, np,
                     S.s_eoff.as<uint32_t>(), S.s_fill.as<uint32_t>(), S.s_entries.as<uint2>(), ivf->t_off.as<uint32_t>());
  if (ivf->ctx->profiling && events) (void)hipEventRecord(S.sev[3], ctx->stream);
  ScanLaunch s{ivf->pool.view(), ivf->t_off.as<uint32_t>(), ivf->t_blocks.as<uint32_t>(), nlist,
               S.s_eoff.as<uint32_t>(), S.s_ioff.as<uint32_t>(), S.s_entries.as<uint2>(), scal + 2,
               scal + 3, qpad, ivf->dpad, segb, k, np, maxsegs, S.s_part.as<uint2>()};
  s.f16 = ivf->f16;
  launch_scan(ctx, s, role);
  if (ivf->ctx->profiling && events) (void)hipEventRecord(S.sev[4], ctx->stream);
  MergeArgs m{};
  m.pool = ivf->pool.view();
  m.lists = ListTable{ivf->t_off.as<uint32_t>(), ivf->t_blocks.as<uint32_t>(), nlist};
  m.probes = probes;
  m.glob_blocks = ivf->t_glob.as<uint32_t>();
  m.part = S.s_part.as<uint2>();
  m.B = B;
  m.k = k;
  m.nprobe = np;
  m.maxsegs = maxsegs;
  m.segb = segb;
  m.out_ids = out_ids;
  m.out_dist = out_dist;
  m.out_counts = out_counts;
  m.out_keys = out_keys;
  launch_merge(ctx, m);
  if (ivf->ctx->profiling && events) (void)hipEventRecord(S.sev[5], ctx->stream);
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

// Fine stage on the matrix cores (kernels_mfma.h): threshold from the nearest list, fp16 MFMA filter over all
// probed lists, exact verification of the survivors, exact rescan of unproven queries.  Same outputs as
// run_fine_exact.
constexpr uint32_t kMfmaSlack = 6;     // phase A scores k + 6 rows
constexpr uint32_t kMfmaCmax = 4096;   // survivor slots per query
namespace {
int env_u(const char* name, int dflt) { return getenv(name) ? atoi(getenv(name)) : dflt; }

template <int MODE>
void launch_mfma(fvdb_ctx* ctx, const MfmaScanArgs& a, int M, bool f16, uint32_t grid) {
#define FVDB_MFMA_CASE(MM)                                                                                      \
  if (f16) hipLaunchKernelGGL((scan_mfma_kernel<MM, 1, MODE>), dim3(grid), dim3(256), 0, ctx->stream, a);       \
  else hipLaunchKernelGGL((scan_mfma_kernel<MM, 0, MODE>), dim3(grid), dim3(256), 0, ctx->stream, a)
  if (M == 1) { FVDB_MFMA_CASE(1); }
  else if (M == 2) { FVDB_MFMA_CASE(2); }
  else { FVDB_MFMA_CASE(4); }
#undef FVDB_MFMA_CASE
}
}  // namespace

int run_fine_mfma(fvdb_ivf* ivf, const Env& E, const float* qpad, uint32_t B, uint32_t k, uint32_t np,
                  const uint32_t* probes, uint64_t* out_ids, float* out_dist, uint32_t* out_counts, uint64_t* out_keys) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  // tuning aids
  static const int M_env = env_u("FVDB_MFMA_M", 2), segb_env = env_u("FVDB_MFMA_SEGB", 0),
                   segbA_env = env_u("FVDB_MFMA_SEGB_A", 1), capA_env = env_u("FVDB_MFMA_CAP_A", 4), wgs_env = env_u("FVDB_MFMA_WGS_PER_CU", 2);
  static const int wg_m_env = env_u("FVDB_MFMA_WG_M", 4);
  int M = M_env >= 4 ? 4 : (M_env >= 2 ? 2 : 1);
  // the workgroup form of the filter (kernels_mfma_wg.h): fp16 rows, dpad a multiple of 128, 32 or 64 queries per group
  static const int wg_env = env_u("FVDB_MFMA_WG", 1), wg_segb_env = env_u("FVDB_MFMA_WG_SEGB", 16),
                   wg_wgs_env = env_u("FVDB_MFMA_WG_PER_CU", 2);
  int wgM = wg_m_env >= 4 ? 4 : 2;
  if (mfma_wg_lds_bytes(ivf->dpad, 16u * wgM) > 64u * 1024u) wgM = 2;  // wide rows: the 32-query tile still fits
  const bool use_wg = wg_env && M == 2 && (ivf->f16 || ivf->pool.half != nullptr) && ivf->dpad % 128 == 0 &&
                      mfma_wg_lds_bytes(ivf->dpad, 16u * wgM) <= 64u * 1024u;
  if (use_wg) M = wgM;
  const uint32_t Q = 16u * M;
  const uint32_t nlist = ivf->nlist, ka = k + kMfmaSlack, cmax = kMfmaCmax;
  const uint32_t segbA = std::max(1, segbA_env);
  const uint32_t segb = use_wg ? (uint32_t)std::max(4, wg_segb_env)
                               : (segb_env > 0 ? (uint32_t)segb_env : pick_segb(ivf, B, np));
  // partial lists of the exact rescan: same geometry as the exact scan's
  const uint32_t fsegb = pick_segb(ivf, B, np), fmaxsegs = std::max<uint32_t>(1, cdiv(ivf->max_list_blocks, fsegb));
  const uint64_t fpart = (uint64_t)B * np * fmaxsegs * k;
  if (fpart >= (1ull << 32)) FAIL(ctx, FVDB_E_UNSUPPORTED, "batch too large for one launch (sub-batch it)");
  HIPCHK(ctx, S.s_qh.ensure((size_t)(B + 1) * ivf->dpad * 2));
  HIPCHK(ctx, S.s_qn2.ensure((size_t)B * 4));
  HIPCHK(ctx, S.s_thr.ensure((size_t)B * 4));
  HIPCHK(ctx, S.s_tA.ensure((size_t)B * 4));
  HIPCHK(ctx, S.s_pa.ensure((size_t)B * 4));
  HIPCHK(ctx, S.s_mslots.ensure((size_t)B * 64 * 4));
  HIPCHK(ctx, S.s_surv.ensure((size_t)B * cmax * 8));
  HIPCHK(ctx, S.s_sdist.ensure((size_t)B * cmax * 4));
  HIPCHK(ctx, S.s_scnt.ensure((size_t)(B + 2) * 4));  // [B] survivor counts, then nfail and the rescan queue head
  HIPCHK(ctx, S.s_fail.ensure((size_t)B * 4));
  HIPCHK(ctx, S.s_part.ensure((size_t)fpart * 8));
  HIPCHK(ctx, S.s_cnt.ensure((size_t)nlist * 4));
  HIPCHK(ctx, S.s_fill.ensure((size_t)nlist * 4));
  HIPCHK(ctx, S.s_eoff.ensure((size_t)(nlist + 1) * 4));
  HIPCHK(ctx, S.s_ioff.ensure((size_t)(nlist + 1) * 4));
  HIPCHK(ctx, S.s_entries.ensure((size_t)B * np * 8));
  HIPCHK(ctx, S.s_scalars.ensure(kScalarsBytes));
  uint32_t* scal = S.s_scalars.as<uint32_t>();
  const uint32_t grid = (uint32_t)ctx->num_cus * (uint32_t)std::max(1, wgs_env);
  const ListTable lists{ivf->t_off.as<uint32_t>(), ivf->t_blocks.as<uint32_t>(), nlist};
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[3], ctx->stream);

  hipLaunchKernelGGL(prep_queries_kernel, dim3(cdiv(B + 1, 4)), dim3(256), 0, ctx->stream, qpad, B, ivf->dpad,
                     (_Float16*)S.s_qh.p, S.s_qn2.as<float>(), S.s_cnt.as<uint32_t>(), nlist,
                     S.s_scnt.as<uint32_t>(), S.s_mslots.as<uint32_t>());
  auto plan = [&](const uint32_t* pr, uint32_t n, uint32_t npp, uint32_t sb, unsigned long long* stats,
                  uint32_t lsplit = 0xFFFFFFFFu, uint32_t sb_tail = 0) {
    // cnt[] is zero on entry: cleared by prep_queries_kernel for the first plan, by plan_scan_kernel for the second
    hipLaunchKernelGGL(plan_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, pr, n, S.s_cnt.as<uint32_t>(),
                       lists.off);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, S.s_cnt.as<uint32_t>(), lists.off,
                       ivf->t_len.as<uint32_t>(), nlist, sb, Q, S.s_eoff.as<uint32_t>(), S.s_ioff.as<uint32_t>(),
                       S.s_fill.as<uint32_t>(), scal + 2, scal + 3, stats, S.s_cnt.as<uint32_t>(), lsplit, sb_tail);
    hipLaunchKernelGGL(plan_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, pr, n, npp,
                       S.s_eoff.as<uint32_t>(), S.s_fill.as<uint32_t>(), S.s_entries.as<uint2>(), lists.off);
  };
  MfmaScanArgs a{};
  const bool half_rows = ivf->f16 || ivf->pool.half != nullptr;  // the filter reads fp16 rows (stored or mirrored)
  const int x_rounded = ivf->f16 ? 0 : (ivf->pool.half ? 1 : 2);  // vs the rows the reference sees: exact, RNE, RTZ
  a.pool_data = ivf->pool.half ? ivf->pool.half : ivf->pool.data;
  a.pool_valid = ivf->pool.valid;
  a.pool_norms = ivf->pool.norms;
  a.d4 = ivf->d4;
  a.list_off = lists.off;
  a.list_blocks = lists.blocks;
  a.nlist = nlist;
  a.entry_off = S.s_eoff.as<uint32_t>();
  a.item_off = S.s_ioff.as<uint32_t>();
  a.entries = (const u32x2*)S.s_entries.p;
  a.n_items = scal + 2;
  a.head = scal + 3;
  a.qh = (const _Float16*)S.s_qh.p;
  a.zero_row = B;
  a.dpad = ivf->dpad;
  a.thr = S.s_thr.as<float>();
  a.cmax = cmax;
  a.surv = (u32x2*)S.s_surv.p;
  a.sval = S.s_sdist.as<float>();
  a.scnt = S.s_scnt.as<uint32_t>();
  a.slots = S.s_mslots.as<uint32_t>();
  a.capA = (uint32_t)std::max(1, capA_env);
  static const int gii_env = env_u("FVDB_MFMA_GROUPS_IN_ITEM", 0);  // tuning aid / A-B
  a.groups_in_item = gii_env ? 1u : 0u;

  // A. threshold: the smallest v per row slot over the head of a near, well-filled list -> (k+6)-th smallest -> thr.
  //    Direct form: one wave per query, one launch (kernels_mfma.h).  FVDB_MFMA_THRESHOLD_PASS=1 keeps the earlier
  //    form (first_probe + plan + matrix-core MODE 1 pass + threshold_kernel) for A/B runs.
  static const bool thr_pass = getenv("FVDB_MFMA_THRESHOLD_PASS") != nullptr;
  if (E.given_thr) {
    // sharded search: the thresholds were computed once per query by the rank owning the list and combined across the
    // ranks before this call (ivf_shared_thresholds + the exchange in comm_sharded.h)
    a.thr = E.given_thr;
  } else if (!thr_pass) {
    ThresholdArgs t{};
    t.rows = a.pool_data;
    t.pool_valid = a.pool_valid;
    t.pool_norms = a.pool_norms;
    t.d4 = ivf->d4;
    t.lists = lists;
    t.list_len = ivf->t_len.as<uint32_t>();
    t.probes = probes;
    t.qh = (const _Float16*)S.s_qh.p;
    t.queries = qpad;
    t.qn = S.s_qn2.as<float>();
    t.xmax_bits = ivf->d_xmax.as<uint32_t>();
    t.B = B;
    t.np = np;
    t.ka = ka;
    t.dpad = ivf->dpad;
    t.capA = a.capA;
    t.min_rows = 256u;
    t.rows_f16 = x_rounded;
    t.thr = S.s_thr.as<float>();
    if (half_rows) hipLaunchKernelGGL((threshold_direct_kernel<true>), dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, t);
    else hipLaunchKernelGGL((threshold_direct_kernel<false>), dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, t);
  } else {
    hipLaunchKernelGGL(first_probe_kernel, dim3(cdiv(B, 256)), dim3(256), 0, ctx->stream, probes, B, np,
                       ivf->t_len.as<uint32_t>(), 256u, S.s_pa.as<uint32_t>());
    plan(S.s_pa.as<uint32_t>(), B, 1, segbA, nullptr);
    a.segb = segbA;
    launch_mfma<1>(ctx, a, M, half_rows, grid);
    hipLaunchKernelGGL(threshold_kernel, dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, S.s_mslots.as<uint32_t>(),
                       S.s_pa.as<uint32_t>(), S.s_qn2.as<float>(), ivf->d_xmax.as<uint32_t>(), B, ka, ivf->dpad,
                       x_rounded, S.s_thr.as<float>());
  }

  // B. filter over all probed lists
  // workgroup form: the last quarter of the lists in small segments (see MfmaScanArgs::lsplit)
  static const int wg_tail_pct = env_u("FVDB_MFMA_WG_TAIL_PCT", 25), wg_tail_segb = env_u("FVDB_MFMA_WG_SEGB_TAIL", 4);
  a.lsplit = 0xFFFFFFFFu;
  a.segb_tail = segb;
  if (use_wg && wg_tail_pct > 0 && wg_tail_segb > 0 && (uint32_t)wg_tail_segb < segb) {
    a.lsplit = (uint32_t)((uint64_t)nlist * (uint32_t)(100 - std::min(wg_tail_pct, 100)) / 100);
    a.segb_tail = (uint32_t)wg_tail_segb;
  }
  plan(probes, B * np, np, segb, (unsigned long long*)(scal + 4), a.lsplit, a.segb_tail);
  a.segb = segb;
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[6], ctx->stream);
  static const int stamps_env = env_u("FVDB_MFMA_STAMPS", 0);  // dev aid: per-item timeline of the workgroup filter
  static void* stamps_buf = nullptr;
  static int stamps_launch = 0;
  constexpr uint32_t kStampsCap = 32768;
  const bool stamp_now = use_wg && stamps_env && ++stamps_launch >= stamps_env && stamps_launch < stamps_env + 3;
  if (stamp_now) {
    if (!stamps_buf) HIPCHK(ctx, hipMalloc(&stamps_buf, (size_t)kStampsCap * 64));
    HIPCHK(ctx, hipMemsetAsync(stamps_buf, 0, (size_t)kStampsCap * 64, ctx->stream));
    a.stamps = (unsigned long long*)stamps_buf;
    a.stamps_cap = kStampsCap;
  }
  if (use_wg) {
    const dim3 wg_grid((uint32_t)ctx->num_cus * (uint32_t)std::max(1, wg_wgs_env));
    if (M == 4)
      hipLaunchKernelGGL(scan_mfma_wg_kernel<4>, wg_grid, dim3(256), mfma_wg_lds_bytes(ivf->dpad, 64), ctx->stream, a);
    else
      hipLaunchKernelGGL(scan_mfma_wg_kernel<2>, wg_grid, dim3(256), mfma_wg_lds_bytes(ivf->dpad, 32), ctx->stream, a);
  } else
    launch_mfma<0>(ctx, a, M, half_rows, grid);
  if (stamp_now) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h((size_t)kStampsCap * 8);
    HIPCHK(ctx, hipMemcpy(h.data(), stamps_buf, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull, t_max = 0;
    size_t n = 0;
    for (size_t i = 0; i < kStampsCap; ++i)
      if (h[i * 8 + 3]) {
        t_min = std::min(t_min, h[i * 8]);
        t_max = std::max(t_max, h[i * 8 + 3]);
        ++n;
      }
    double s_loc = 0, s_tile = 0, s_comp = 0, s_steps = 0, s_blocks = 0;
    size_t per_xcd[16] = {0};
    std::vector<double> per_step;
    for (size_t i = 0; i < kStampsCap; ++i) {
      const unsigned long long* r = &h[i * 8];
      if (!r[3]) continue;
      const double steps = (double)((r[5] + 3) / 4) * (double)(ivf->dpad / 16);
      s_loc += (double)(r[1] - r[0]);
      s_tile += (double)(r[2] - r[1]);
      s_comp += (double)(r[3] - r[2]);
      s_steps += steps;
      s_blocks += (double)r[5];
      per_xcd[r[7] & 15]++;
      per_step.push_back((double)(r[3] - r[2]) / steps);
    }
    std::sort(per_step.begin(), per_step.end());
    const double tick = 0.01;  // us per tick of the 100 MHz wall clock
    fprintf(stderr,
            "[mfma stamps] launch %d: %zu items, %0.f blocks, span %.1f us | per item (wave 0): locate %.2f us, tile %.2f us, "
            "compute %.2f us | per 16-dim step: mean %.3f us, p10 %.3f, p50 %.3f, p90 %.3f | items per XCD",
            stamps_launch, n, s_blocks, (double)(t_max - t_min) * tick, s_loc / n * tick, s_tile / n * tick, s_comp / n * tick,
            s_comp / s_steps * tick, per_step.empty() ? 0.0 : per_step[per_step.size() / 10] * tick,
            per_step.empty() ? 0.0 : per_step[per_step.size() / 2] * tick,
            per_step.empty() ? 0.0 : per_step[per_step.size() * 9 / 10] * tick);
    for (int x = 0; x < 8; ++x) fprintf(stderr, " %zu", per_xcd[x]);
    // when did the last item START, and how long were the items that finished last
    unsigned long long last_start = 0;
    for (size_t i = 0; i < kStampsCap; ++i)
      if (h[i * 8 + 3]) last_start = std::max(last_start, h[i * 8]);
    fprintf(stderr, " | last item drawn at %.1f us\n", (double)(last_start - t_min) * tick);
    a.stamps = nullptr;
  }
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[7], ctx->stream);
  // B'. queries whose survivors outgrew the buffer get a threshold from those survivors and a second filter pass over
  //     their probes alone (normally nobody: the three plan kernels and the filter find nothing to do)
  static const bool no_refine = getenv("FVDB_MFMA_NO_REFINE") != nullptr;  // tuning aid / A-B
  bool refine = !E.given_thr && !no_refine;
  if (refine && ivf->scan_mode == FVDB_SCAN_AUTO) {
    std::lock_guard<std::mutex> lk(ivf->mu);
    if (ivf->h_fb.p && ivf->mfma_q - ivf->overflow_q >= 2048) {
      // worth five more launches per batch once more than one query in a thousand overflows (a stray one is cheaper to
      // rescan exactly); looked at every 2048 queries, the counters lag by the batches in flight
      const volatile uint32_t* c = (const volatile uint32_t*)ivf->h_fb.p;
      const uint64_t now = (uint64_t)c[2] + c[6], dq = ivf->mfma_q - ivf->overflow_q;
      if ((now - ivf->overflow_seen) * 1000 > dq) ivf->refine_batches_left = 1024;
      ivf->overflow_seen = now;
      ivf->overflow_q = ivf->mfma_q;
    }
    if (ivf->refine_batches_left > 0) ivf->refine_batches_left -= 1;
    else refine = false;
  }
  if (refine) {
    HIPCHK(ctx, S.s_probes2.ensure((size_t)B * np * 4));
    hipLaunchKernelGGL(refine_threshold_kernel, dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, S.s_sdist.as<float>(),
                       S.s_scnt.as<uint32_t>(), probes, S.s_qn2.as<float>(), ivf->d_xmax.as<uint32_t>(), B, np, ka, cmax, ivf->dpad,
                       x_rounded, S.s_thr.as<float>(), S.s_probes2.as<uint32_t>(), ivf->s_fallbacks.as<uint32_t>() + 6);
    plan(S.s_probes2.as<uint32_t>(), B * np, np, segb, nullptr, a.lsplit, a.segb_tail);
    if (use_wg) {
      const dim3 wg_grid((uint32_t)ctx->num_cus * (uint32_t)std::max(1, wg_wgs_env));
      if (M == 4)
        hipLaunchKernelGGL(scan_mfma_wg_kernel<4>, wg_grid, dim3(256), mfma_wg_lds_bytes(ivf->dpad, 64), ctx->stream, a);
      else
        hipLaunchKernelGGL(scan_mfma_wg_kernel<2>, wg_grid, dim3(256), mfma_wg_lds_bytes(ivf->dpad, 32), ctx->stream, a);
    } else {
      launch_mfma<0>(ctx, a, M, half_rows, grid);
    }
  }
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[4], ctx->stream);
  S.pend_filter = true;

  // C. select
  VerifyArgs v{};
  v.pool = ivf->pool.view();
  v.lists = lists;
  v.probes = probes;
  v.glob_blocks = ivf->t_glob.as<uint32_t>();
  v.queries = qpad;
  v.qn = S.s_qn2.as<float>();
  v.xmax_bits = ivf->d_xmax.as<uint32_t>();
  v.thr = E.given_thr ? E.given_thr : S.s_thr.as<float>();
  v.remote_thr = E.given_thr ? 1 : 0;
  v.surv = (const u32x2*)S.s_surv.p;
  v.sval = S.s_sdist.as<float>();
  v.scnt = S.s_scnt.as<uint32_t>();
  v.B = B;
  v.k = k;
  v.ka = ka;
  v.nprobe = np;
  v.d = ivf->dpad;
  v.dpad = ivf->dpad;
  v.cmax = cmax;
  v.rows_f16 = x_rounded;
  v.rows_rm = ivf->pool.rm;
  v.out_ids = out_ids;
  v.out_dist = out_dist;
  v.out_counts = out_counts;
  v.out_keys = out_keys;
  v.fallbacks = ivf->s_fallbacks.as<uint32_t>() + 1;
  v.reasons = ivf->s_fallbacks.as<uint32_t>() + 2;
  v.fail_list = S.s_fail.as<uint32_t>();
  v.nfail = S.s_scnt.as<uint32_t>() + B;
  if (ivf->f16) hipLaunchKernelGGL((select_kernel<1>), dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, v);
  else hipLaunchKernelGGL((select_kernel<0>), dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, v);

  // exact rescan of the queries that were not proven (normally none: both kernels return at once)
  FallbackArgs fa{};
  fa.pool = ivf->pool.view();
  fa.lists = lists;
  fa.probes = probes;
  fa.queries = qpad;
  fa.fail_list = v.fail_list;
  fa.nfail = v.nfail;
  fa.head = S.s_scnt.as<uint32_t>() + B + 1;
  fa.k = k;
  fa.nprobe = np;
  fa.dpad = ivf->dpad;
  fa.segb = fsegb;
  fa.maxsegs = fmaxsegs;
  fa.part = (u32x2*)S.s_part.p;
  if (ivf->f16) hipLaunchKernelGGL((fallback_scan_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, fa);
  else hipLaunchKernelGGL((fallback_scan_kernel<0>), dim3(grid), dim3(256), 0, ctx->stream, fa);
  MergeArgs fm{};
  fm.pool = ivf->pool.view();
  fm.lists = lists;
  fm.probes = probes;
  fm.glob_blocks = ivf->t_glob.as<uint32_t>();
  fm.part = S.s_part.as<uint2>();
  fm.B = B;
  fm.k = k;
  fm.nprobe = np;
  fm.maxsegs = fmaxsegs;
  fm.segb = fsegb;
  fm.out_ids = out_ids;
  fm.out_dist = out_dist;
  fm.out_counts = out_counts;
  fm.out_keys = out_keys;
  fm.qlist = v.fail_list;
  fm.nq = v.nfail;
  launch_merge(ctx, fm);
  if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[5], ctx->stream);
  // the rescan counter, for AUTO's hit-rate watch (run_fine): a 4-byte copy into pinned memory, nobody waits for it
  {
    std::lock_guard<std::mutex> lk(ivf->mu);
    const bool fresh = ivf->h_fb.p == nullptr;
    HIPCHK(ctx, ivf->h_fb.ensure(64));
    if (fresh) std::memset(ivf->h_fb.p, 0, 64);
    ivf->mfma_q += B;
  }
  HIPCHK(ctx, hipMemcpyAsync(ivf->h_fb.p, ivf->s_fallbacks.p, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

int run_fine(fvdb_ivf* ivf, const Env& E, const float* qpad, uint32_t B, uint32_t k, uint32_t np, const uint32_t* probes,
             uint64_t* out_ids, float* out_dist, uint32_t* out_counts, uint64_t* out_keys, int role) {
  IvfScratch& S = *E.S;
  static const bool env_exact = getenv("FVDB_SCAN_EXACT") != nullptr;  // tuning aid
  S.pend_filter = false;
  bool mfma = (ivf->scan_mode == FVDB_SCAN_AUTO || ivf->scan_mode == FVDB_SCAN_FILTER) && !env_exact && role == ROLE_LIST && ivf->dpad % 16 == 0 &&
              k + kMfmaSlack <= 32 && np <= 256 && B >= 32 && B <= 16384 && ivf->pool.norms != nullptr;
  if (mfma && ivf->scan_mode == FVDB_SCAN_AUTO) {
    std::lock_guard<std::mutex> lk(ivf->mu);  // the hit-rate watch is shared by every search on the index
    if (ivf->exact_batches_left > 0) {
      ivf->exact_batches_left -= 1;
      mfma = false;
    } else if (ivf->h_fb.p && ivf->mfma_q - ivf->q_seen >= 2048) {
      // rescans among the matrix-core queries since the last look (the counter lags by the batches in flight)
      const uint64_t fb_now = ((volatile uint32_t*)ivf->h_fb.p)[1];
      const uint64_t dq = ivf->mfma_q - ivf->q_seen, dfb = fb_now - ivf->fb_seen;
      ivf->q_seen = ivf->mfma_q;
      ivf->fb_seen = fb_now;
      if (dfb * 8 > dq) {  // more than 1 in 8: the exact scan is cheaper here; look again after a while, ever more rarely
        ivf->backoff_len = std::min<uint32_t>(ivf->backoff_len ? ivf->backoff_len * 2 : 64, 4096);
        ivf->exact_batches_left = ivf->backoff_len;
        mfma = false;
      } else {
        ivf->backoff_len = 0;
      }
    }
  }
  if (mfma) return run_fine_mfma(ivf, E, qpad, B, k, np, probes, out_ids, out_dist, out_counts, out_keys);
  return run_fine_exact(ivf, E, qpad, B, k, np, probes, out_ids, out_dist, out_counts, out_keys, role);
}

int finish_profile(fvdb_ivf* ivf, const Env& E, bool coarse, bool fine) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  if (!ivf->ctx->profiling) return FVDB_OK;
  if (ivf->ctx->profiling == 2 && !S.collecting) {  // deferred: remember what to fold in later
    S.pend_coarse = coarse;
    S.pend_fine = fine;
    S.pending_profile = true;
    return FVDB_OK;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0;
  if (coarse) {
    (void)hipEventElapsedTime(&ms, S.sev[0], S.sev[1]);
    ivf->stage_ms[0] += ms;
    (void)hipEventElapsedTime(&ms, S.sev[1], S.sev[2]);
    ivf->stage_ms[1] += ms;
  }
  if (fine) {
    if (coarse) {
      (void)hipEventElapsedTime(&ms, S.sev[2], S.sev[3]);
      ivf->stage_ms[2] += ms;
    }
    (void)hipEventElapsedTime(&ms, S.sev[3], S.sev[4]);
    ivf->stage_ms[3] += ms;
    (void)hipEventElapsedTime(&ms, S.sev[4], S.sev[5]);
    ivf->stage_ms[4] += ms;
    if (S.pend_filter) {
      (void)hipEventElapsedTime(&ms, S.sev[6], S.sev[7]);
      ivf->stage_ms[5] += ms;
    }
  }
  ivf->stage_calls += 1;
  return FVDB_OK;
}

// largest sub-batch whose fine-stage partial buffer stays under ~1 GiB
uint32_t sub_batch(fvdb_ivf* ivf, uint32_t B, uint32_t k, uint32_t np) {
  const uint32_t segb = pick_segb(ivf, B, np);
  const uint64_t per_q = (uint64_t)np * std::max<uint32_t>(1, cdiv(ivf->max_list_blocks, segb)) * k * 8;
  uint64_t fit = (1ull << 30) / std::max<uint64_t>(per_q, 1);
  fit = std::max<uint64_t>(fit, 1);
  fit = std::min<uint64_t>(fit, 16384);
  return (uint32_t)std::min<uint64_t>(fit, B);
}

// ---- sharded search: filter thresholds computed once per query across the ranks (comm_sharded.h) ----
// Whether a sharded step of B scanned queries uses the shared thresholds.  Every rank must answer alike (the answer
// decides whether a collective is issued), so only quantities that are the same on every rank enter: shapes, and the
// LOGICAL index's longest list.
bool thr_share_ok(fvdb_ivf* ivf, uint32_t B, uint32_t k, uint32_t np) {
  static const bool env_exact = getenv("FVDB_SCAN_EXACT") != nullptr, env_off = getenv("FVDB_NO_SHARED_THR") != nullptr;
  if (env_exact || env_off || ivf->scan_mode != 0) return false;
  if (!(ivf->dpad % 16 == 0 && k + kMfmaSlack <= 32 && np <= 256 && B >= 32 && B <= 16384)) return false;
  uint32_t gmax = 0;
  if (ivf->glob_set) {
    for (uint32_t b : ivf->glob_blocks_host) gmax = std::max(gmax, b);
  } else {
    gmax = ivf->max_list_blocks;
  }
  // one sub-batch on every rank (sub_batch() with the logical index's longest list: local lists are no longer)
  const uint32_t segb = gmax >= 4096 ? 16 : ((uint64_t)B * np >= 4096 ? 4 : ((uint64_t)B * np >= 512 ? 2 : 1));
  const uint64_t per_q = (uint64_t)np * std::max<uint32_t>(1, cdiv(gmax, segb)) * k * 8;
  return (1ull << 30) / std::max<uint64_t>(per_q, 1) >= B;
}

// U_q (kernels_mfma.h, ThresholdArgs::glob_blocks) for the B queries of a sharded step, +inf where this rank does
// not own the list the threshold comes from.  Leaves the slot's fp16 queries and |q|^2 in place for thr_combine.
int ivf_shared_thresholds(fvdb_ivf* ivf, const Env& E, const float* q_dev, const uint32_t* probes, uint32_t B, uint32_t k,
                          uint32_t np, float* u_out) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = upload_table(ivf);
  if (rc) return rc;
  std::lock_guard<std::mutex> enq(S.enq);
  HIPCHK(ctx, S.s_qn2.ensure((size_t)B * 4));
  if (!ivf->pool.norms || !ivf->d_xmax.p) {  // no rows on this rank yet: it owns nothing
    HIPCHK(ctx, hipMemsetAsync(S.s_qn2.p, 0, (size_t)B * 4, ctx->stream));
    hipLaunchKernelGGL(fill_f32_kernel, dim3(cdiv(B, 256)), dim3(256), 0, ctx->stream, u_out, B, __builtin_huge_valf());
    HIPCHK(ctx, hipGetLastError());
    return FVDB_OK;
  }
  const float* qpad = nullptr;
  rc = padded_queries(ivf, E, q_dev, B, &qpad);
  if (rc) return rc;
  HIPCHK(ctx, S.s_qh.ensure((size_t)(B + 1) * ivf->dpad * 2));
  HIPCHK(ctx, S.s_cnt.ensure((size_t)ivf->nlist * 4));
  HIPCHK(ctx, S.s_scnt.ensure((size_t)(B + 2) * 4));
  HIPCHK(ctx, S.s_mslots.ensure((size_t)B * 64 * 4));
  hipLaunchKernelGGL(prep_queries_kernel, dim3(cdiv(B + 1, 4)), dim3(256), 0, ctx->stream, qpad, B, ivf->dpad,
                     (_Float16*)S.s_qh.p, S.s_qn2.as<float>(), S.s_cnt.as<uint32_t>(), ivf->nlist,
                     S.s_scnt.as<uint32_t>(), S.s_mslots.as<uint32_t>());
  static const int capA_env = env_u("FVDB_MFMA_CAP_A", 4);
  const bool half_rows = ivf->f16 || ivf->pool.half != nullptr;
  ThresholdArgs t{};
  t.rows = ivf->pool.half ? ivf->pool.half : ivf->pool.data;
  t.pool_valid = ivf->pool.valid;
  t.pool_norms = ivf->pool.norms;
  t.d4 = ivf->d4;
  t.lists = ListTable{ivf->t_off.as<uint32_t>(), ivf->t_blocks.as<uint32_t>(), ivf->nlist};
  t.list_len = ivf->t_len.as<uint32_t>();
  t.probes = probes;
  t.qh = (const _Float16*)S.s_qh.p;
  t.queries = qpad;
  t.qn = S.s_qn2.as<float>();
  t.xmax_bits = ivf->d_xmax.as<uint32_t>();
  t.B = B;
  t.np = np;
  t.ka = k + kMfmaSlack;
  t.dpad = ivf->dpad;
  t.capA = (uint32_t)std::max(1, capA_env);
  t.min_rows = 256u;
  t.rows_f16 = ivf->f16 ? 0 : (ivf->pool.half ? 1 : 2);
  t.thr = u_out;
  t.glob_blocks = ivf->t_glob.as<uint32_t>();
  if (half_rows) hipLaunchKernelGGL((threshold_direct_kernel<true>), dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, t);
  else hipLaunchKernelGGL((threshold_direct_kernel<false>), dim3(cdiv(B, 4)), dim3(256), 0, ctx->stream, t);
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

// thr = min over the ranks' U arrays + this rank's own error bound
int ivf_thr_combine(fvdb_ivf* ivf, const Env& E, const float* u_all, uint32_t W, uint32_t B, float* thr_out,
                    bool loopback_fill = false) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  if (!ivf->d_xmax.p) {
    HIPCHK(ctx, ivf->d_xmax.ensure(4));
    HIPCHK(ctx, hipMemsetAsync(ivf->d_xmax.p, 0, 4, ctx->stream));
  }
  hipLaunchKernelGGL(thr_combine_kernel, dim3(cdiv(B, 256)), dim3(256), 0, ctx->stream, u_all, W, B, S.s_qn2.as<float>(),
                     ivf->d_xmax.as<uint32_t>(), (float)ivf->dpad, ivf->f16 ? 0 : (ivf->pool.half ? 1 : 2), thr_out);
#ifdef FVDB_DEV_TOOLS
  if (loopback_fill)
    hipLaunchKernelGGL(thr_loopback_fill_kernel, dim3(1), dim3(1024), 0, ctx->stream, thr_out, S.s_qn2.as<float>(), B);
#else
  (void)loopback_fill;
#endif
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

int check_finite(fvdb_ctx* ctx, const float* x, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i)
    if (!std::isfinite(x[i])) FAIL(ctx, FVDB_E_NONFINITE, "non-finite input value");
  return FVDB_OK;
}

}  // namespace

// GPU_MAX_HW_QUEUES bookkeeping (fvdb_ctx_create / fvdb_ctx_info): 0 unknown, 1 set by the host application, 2 set by this
// library before HIP came up, 3 could not be applied (the HIP runtime was already initialised in this process)
static std::mutex g_hwq_mu;
static int g_hwq_state = 0, g_hwq_value = 0;
static bool hip_runtime_already_up() {  // the ROCm runtime holds /dev/kfd open once it is initialised
  char path[64], tgt[256];
  for (int fd = 0; fd < 1024; ++fd) {
    snprintf(path, sizeof path, "/proc/self/fd/%d", fd);
    const ssize_t n = readlink(path, tgt, sizeof tgt - 1);
    if (n <= 0) continue;
    tgt[n] = 0;
    if (strstr(tgt, "/dev/kfd")) return true;
  }
  return false;
}

// =============================================================================================
// context
// =============================================================================================
extern "C" {

const char* fvdb_version(void) { return "fvdb-hip 0.1 (gfx950)"; }

int fvdb_ctx_create(int device, fvdb_ctx** out) {
  if (!out) return FVDB_E_INVALID;
  *out = nullptr;
  // Batches in flight run on streams of their own (two per batch); the HIP runtime multiplexes streams onto
  // GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue run one after the other.
  // Ask for 16 unless the host application chose a value; this only takes effect if HIP has not been initialised
  // yet in this process (measured: 8 batches in flight, traversal only, 0.42 -> 0.27 ms per 1024-query step).
  // Whether that request can still take effect is recorded for fvdb_ctx_info (and said once on stderr when it cannot).
  {
    std::lock_guard<std::mutex> lk(g_hwq_mu);
    if (g_hwq_state == 0) {
      const char* set = getenv("GPU_MAX_HW_QUEUES");
      if (set) {
        g_hwq_value = atoi(set);
        g_hwq_state = 1;  // the host application's choice
      } else if (hip_runtime_already_up()) {
        g_hwq_value = 4;  // the runtime's default: several batches in flight share hardware queues
        g_hwq_state = 3;
        fprintf(stderr, "[fvdb] warning: HIP was initialised before fvdb_ctx_create, so GPU_MAX_HW_QUEUES=16 cannot be applied; "
                        "with the default of 4 hardware queues batches in flight overlap less (export GPU_MAX_HW_QUEUES=16 before "
                        "the process touches the GPU)\n");
      } else {
        (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
        g_hwq_value = 16;
        g_hwq_state = 2;
      }
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return FVDB_E_HIP;
  fvdb_ctx* ctx = new (std::nothrow) fvdb_ctx();
  if (!ctx) return FVDB_E_OOM;
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return FVDB_E_HIP;
  }
  (void)hipEventCreate(&ctx->ev0);
  (void)hipEventCreate(&ctx->ev1);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  *out = ctx;
  return FVDB_OK;
}

int fvdb_ctx_device(fvdb_ctx* ctx) { return ctx ? ctx->device : -1; }

int fvdb_ctx_info(fvdb_ctx* ctx, fvdb_ctx_info_t* out) {
  if (!ctx || !out) return FVDB_E_INVALID;
  std::lock_guard<std::mutex> lk(g_hwq_mu);
  out->device = ctx->device;
  out->compute_units = ctx->num_cus;
  out->hw_queues = g_hwq_value;
  out->hw_queues_source = g_hwq_state;
  return FVDB_OK;
}

void fvdb_ctx_destroy(fvdb_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->h_stage.release();
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int fvdb_ctx_synchronize(fvdb_ctx* ctx) {
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}
int fvdb_device_synchronize(fvdb_ctx* ctx) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  return FVDB_OK;
}
void* fvdb_ctx_stream(fvdb_ctx* ctx) { return (void*)ctx->stream; }
const char* fvdb_last_error(fvdb_ctx* ctx) {
  if (!ctx) return "null context";
  static thread_local std::string copy;  // another thread's failure must not pull the text from under the reader
  std::lock_guard<std::mutex> lk(ctx->err_mu);
  copy = ctx->err;
  return copy.c_str();
}
int fvdb_ctx_set_profiling(fvdb_ctx* ctx, int on) {
  ctx->profiling = on;
  return FVDB_OK;
}

int fvdb_dev_alloc(fvdb_ctx* ctx, size_t bytes, void** out) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(out, std::max<size_t>(bytes, 16)));
  return FVDB_OK;
}
int fvdb_dev_free(fvdb_ctx* ctx, void* p) {
  HIPCHK(ctx, hipFree(p));
  return FVDB_OK;
}
int fvdb_dev_upload(fvdb_ctx* ctx, void* dst, const void* src, size_t bytes) {
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}
int fvdb_dev_download(fvdb_ctx* ctx, void* dst, const void* src, size_t bytes) {
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}
// ---- pieces for keeping more than one batch in flight (pinned host memory, non-blocking copies, events) ----
int fvdb_host_alloc(fvdb_ctx* ctx, size_t bytes, void** out) {
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipHostMalloc(out, std::max<size_t>(bytes, 16), hipHostMallocDefault));
  return FVDB_OK;
}
void fvdb_host_free(fvdb_ctx* ctx, void* p) {
  (void)ctx;
  if (p) (void)hipHostFree(p);
}
int fvdb_dev_download_async(fvdb_ctx* ctx, void* dst_pinned, const void* src, size_t bytes) {
  HIPCHK(ctx, hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return FVDB_OK;
}
struct fvdb_event {
  hipEvent_t ev = nullptr;
};
int fvdb_event_create(fvdb_ctx* ctx, fvdb_event** out) {
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  fvdb_event* e = new (std::nothrow) fvdb_event();
  if (!e) return FVDB_E_OOM;
  if (hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) != hipSuccess) {
    delete e;
    FAIL(ctx, FVDB_E_HIP, "hipEventCreate failed");
  }
  *out = e;
  return FVDB_OK;
}
void fvdb_event_destroy(fvdb_event* e) {
  if (!e) return;
  if (e->ev) (void)hipEventDestroy(e->ev);
  delete e;
}
int fvdb_event_record(fvdb_ctx* ctx, fvdb_event* e) {
  HIPCHK(ctx, hipEventRecord(e->ev, ctx->stream));
  return FVDB_OK;
}
int fvdb_event_wait(fvdb_ctx* ctx, fvdb_event* e) {
  HIPCHK(ctx, hipEventSynchronize(e->ev));
  return FVDB_OK;
}

int fvdb_timer_start(fvdb_ctx* ctx) {
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return FVDB_OK;
}
int fvdb_timer_stop_ms(fvdb_ctx* ctx, float* out_ms) {
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
  HIPCHK(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
  return FVDB_OK;
}

// =============================================================================================
// similarity utilities (a3)
// =============================================================================================
static int dot_cosine(fvdb_ctx* ctx, const float* q, uint32_t B, const float* x, uint64_t n, uint32_t d, float* out,
                      int cosine) {
  if (!ctx || !q || !x || !out || d == 0) return FVDB_E_INVALID;
  if (B == 0 || n == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, q, (uint64_t)B * d);
  if (rc) return rc;
  rc = check_finite(ctx, x, n * d);
  if (rc) return rc;
  float *dq = nullptr, *dx = nullptr, *dout = nullptr;
  auto done = [&]() {
    if (dq) (void)hipFree(dq);
    if (dx) (void)hipFree(dx);
    if (dout) (void)hipFree(dout);
  };
  if (hipMalloc(&dq, (size_t)B * d * 4) != hipSuccess || hipMalloc(&dx, n * d * 4) != hipSuccess ||
      hipMalloc(&dout, (size_t)B * n * 4) != hipSuccess) {
    done();
    FAIL(ctx, FVDB_E_OOM, "similarity scratch allocation failed");
  }
  (void)hipMemcpyAsync(dq, q, (size_t)B * d * 4, hipMemcpyHostToDevice, ctx->stream);
  (void)hipMemcpyAsync(dx, x, n * d * 4, hipMemcpyHostToDevice, ctx->stream);
  hipLaunchKernelGGL(dot_cosine_kernel, dim3(cdiv((uint64_t)B * n, 256)), dim3(256), 0, ctx->stream, dq, dx, B, n, d,
                     cosine, dout);
  hipError_t e = hipMemcpyAsync(out, dout, (size_t)B * n * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  done();
  if (e != hipSuccess) FAIL(ctx, FVDB_E_HIP, hipGetErrorString(e));
  return FVDB_OK;
}
int fvdb_dot_products(fvdb_ctx* ctx, const float* q, uint32_t B, const float* x, uint64_t n, uint32_t d, float* out) {
  return dot_cosine(ctx, q, B, x, n, d, out, 0);
}
int fvdb_cosine_similarities(fvdb_ctx* ctx, const float* q, uint32_t B, const float* x, uint64_t n, uint32_t d,
                             float* out) {
  return dot_cosine(ctx, q, B, x, n, d, out, 1);
}

// =============================================================================================
// IVF
// =============================================================================================
int fvdb_ivf_create(fvdb_ctx* ctx, uint32_t d, uint32_t nlist, fvdb_ivf** out) {
  return fvdb_ivf_create_ex(ctx, d, nlist, FVDB_F32, out);
}

int fvdb_ivf_create_ex(fvdb_ctx* ctx, uint32_t d, uint32_t nlist, int row_dtype, fvdb_ivf** out) {
  if (!ctx || !out) return FVDB_E_INVALID;
  *out = nullptr;
  if (row_dtype != FVDB_F32 && row_dtype != FVDB_F16) FAIL(ctx, FVDB_E_INVALID, "row_dtype must be FVDB_F32 or FVDB_F16");
  if (d == 0 || nlist == 0) FAIL(ctx, FVDB_E_INVALID, "d and nlist must be > 0");
  if (d > 2048 * 4) FAIL(ctx, FVDB_E_INVALID, "d too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  fvdb_ivf* ivf = new (std::nothrow) fvdb_ivf();
  if (!ivf) return FVDB_E_OOM;
  ivf->ctx = ctx;
  ivf->d = d;
  ivf->f16 = row_dtype == FVDB_F16;
  ivf->dpad = ivf->f16 ? ((d + 15) / 16) * 16 : ((d + 3) / 4) * 4;  // fp16 rows: whole 16-dim steps
  ivf->d4 = ivf->dpad / 4;
  ivf->nlist = nlist;
  ivf->list_blocks.assign(nlist, {});
  ivf->list_len.assign(nlist, 0);
  ivf->pool.d4 = ivf->d4;
  ivf->pool.esize = ivf->f16 ? 2 : 4;
  static const bool no_mirror = getenv("FVDB_NO_FP16_MIRROR") != nullptr;  // tuning aid: filter from the f32 rows
  ivf->pool.mirror = !ivf->f16 && ivf->dpad % 16 == 0 && !no_mirror;
  ivf->cpool.d4 = ivf->d4;
  for (auto& e : ivf->sev) (void)hipEventCreate(&e);
  *out = ivf;
  return FVDB_OK;
}

void fvdb_ivf_destroy(fvdb_ivf* ivf) {
  if (!ivf) return;
  (void)hipSetDevice(ivf->ctx->device);
  (void)hipStreamSynchronize(ivf->ctx->stream);
  ivf->pool.release();
  ivf->cpool.release();
  DBuf* bufs[] = {&ivf->d_xmax, &ivf->d_cent_pad, &ivf->d_cnorm, &ivf->d_cnmax, &ivf->s_fallbacks, &ivf->d_centroids_rm,
                  &ivf->c_off, &ivf->c_blocks, &ivf->c_glob, &ivf->t_off, &ivf->t_blocks, &ivf->t_glob, &ivf->t_len};
  for (DBuf* b : bufs) b->release();
  ivf->release_all();
  for (auto& sp : ivf->spare) sp.release_all();
  for (auto& sp : ivf->lease_set) sp.release_all();
  for (auto& c : ivf->lease_ctx)
    if (c) fvdb_ctx_destroy(c);
  ivf->h_fb.release();
  delete ivf;
}

static int install_centroids(fvdb_ivf* ivf, const float* d_rowmajor /* device [nlist][d] */) {
  fvdb_ctx* ctx = ivf->ctx;
  const uint32_t nlist = ivf->nlist, cblocks = cdiv(nlist, 64);
  int rc = ivf->cpool.reserve(ctx, cblocks);
  if (rc) return rc;
  HIPCHK(ctx, hipMemsetAsync(ivf->cpool.valid, 0, (size_t)ivf->cpool.cap_blocks * 8, ctx->stream));
  ivf->cpool.used_blocks = cblocks;
  std::vector<uint32_t> slots(nlist), off{0, cblocks}, blocks(cblocks), glob{cblocks};
  for (uint32_t i = 0; i < nlist; ++i) slots[i] = i;
  for (uint32_t i = 0; i < cblocks; ++i) blocks[i] = i;
  HIPCHK(ctx, ivf->s_slots.ensure((size_t)nlist * 4));
  HIPCHK(ctx, ivf->c_off.ensure(8));
  HIPCHK(ctx, ivf->c_blocks.ensure((size_t)cblocks * 4));
  HIPCHK(ctx, ivf->c_glob.ensure(4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->s_slots.p, slots.data(), (size_t)nlist * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ivf->c_off.p, off.data(), 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ivf->c_blocks.p, blocks.data(), (size_t)cblocks * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ivf->c_glob.p, glob.data(), 4, hipMemcpyHostToDevice, ctx->stream));
  const uint64_t threads = (uint64_t)nlist * ivf->d4;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, ctx->stream, d_rowmajor, ivf->d,
                     ivf->d4, (uint64_t)nlist, ivf->s_slots.as<uint32_t>(), (const uint64_t*)nullptr,
                     (float4*)ivf->cpool.data, ivf->cpool.ids, (unsigned long long*)ivf->cpool.valid, (void*)nullptr,
                     (float4*)nullptr);
  // matrix-core coarse stage: padded row-major table, |c|^2, max |c|^2
  const float* cpad = d_rowmajor;
  if (ivf->d != ivf->dpad) {
    HIPCHK(ctx, ivf->d_cent_pad.ensure((size_t)nlist * ivf->dpad * 4));
    const uint64_t tot = (uint64_t)nlist * ivf->dpad;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, ctx->stream, d_rowmajor, ivf->d, ivf->dpad,
                       (uint64_t)nlist, ivf->d_cent_pad.as<float>());
    cpad = ivf->d_cent_pad.as<float>();
  }
  HIPCHK(ctx, ivf->d_cnorm.ensure((size_t)nlist * 4));
  HIPCHK(ctx, ivf->d_cnmax.ensure(4));
  HIPCHK(ctx, ivf->s_fallbacks.ensure(32));  // [0] coarse, [1] list scan, [2..5] why a query was not proven (VerifyArgs::reasons)
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(cdiv(nlist, 256)), dim3(256), 0, ctx->stream, cpad, ivf->dpad, ivf->dpad,
                     nlist, ivf->d_cnorm.as<float>());
  hipLaunchKernelGGL(max_f32_kernel, dim3(1), dim3(64), 0, ctx->stream, ivf->d_cnorm.as<float>(), nlist,
                     ivf->d_cnmax.as<float>());
  HIPCHK(ctx, hipMemsetAsync(ivf->s_fallbacks.p, 0, 32, ctx->stream));
  if (ivf->h_fb.p) *(volatile uint32_t*)ivf->h_fb.p = 0;
  ivf->mfma_q = ivf->fb_seen = ivf->q_seen = 0;
  ivf->exact_batches_left = ivf->backoff_len = 0;
  ivf->overflow_seen = ivf->overflow_q = 0;
  ivf->refine_batches_left = 0;
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ivf->trained = true;
  return FVDB_OK;
}

static void reset_lists(fvdb_ivf* ivf) {
  ivf->list_blocks.assign(ivf->nlist, {});
  ivf->list_len.assign(ivf->nlist, 0);
  ivf->total_rows = 0;
  ivf->pool.used_blocks = 0;
  ivf->max_list_blocks = 0;
  ivf->table_dirty = true;
}

int fvdb_ivf_set_centroids(fvdb_ivf* ivf, const float* centroids) {
  fvdb_ctx* ctx = ivf->ctx;
  if (!centroids) FAIL(ctx, FVDB_E_INVALID, "null centroids");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)ivf->nlist * ivf->d;
  int rc = check_finite(ctx, centroids, n);
  if (rc) return rc;
  ivf->h_centroids.assign(centroids, centroids + n);
  HIPCHK(ctx, ivf->d_centroids_rm.ensure(n * 4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->d_centroids_rm.p, centroids, n * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = install_centroids(ivf, ivf->d_centroids_rm.as<float>());
  if (rc) return rc;
  reset_lists(ivf);
  if (ivf->pool.valid && ivf->pool.cap_blocks)
    HIPCHK(ctx, hipMemsetAsync(ivf->pool.valid, 0, (size_t)ivf->pool.cap_blocks * 8, ctx->stream));
  return FVDB_OK;
}

int fvdb_ivf_get_centroids(fvdb_ivf* ivf, float* out) {
  if (!ivf->trained) FAIL(ivf->ctx, FVDB_E_NOT_TRAINED, "index not trained");
  std::memcpy(out, ivf->h_centroids.data(), ivf->h_centroids.size() * 4);
  return FVDB_OK;
}

int fvdb_ivf_clear(fvdb_ivf* ivf) {
  fvdb_ctx* ctx = ivf->ctx;
  reset_lists(ivf);
  if (ivf->pool.valid && ivf->pool.cap_blocks)
    HIPCHK(ctx, hipMemsetAsync(ivf->pool.valid, 0, (size_t)ivf->pool.cap_blocks * 8, ctx->stream));
  return FVDB_OK;
}

int fvdb_ivf_reserve(fvdb_ivf* ivf, uint64_t n_rows) {
  HIPCHK(ivf->ctx, hipSetDevice(ivf->ctx->device));
  // every list may end in a partly filled block
  return ivf->pool.reserve(ivf->ctx, cdiv(n_rows, 64) + ivf->nlist);
}

int fvdb_ivf_list_sizes(fvdb_ivf* ivf, uint64_t* out) {
  for (uint32_t L = 0; L < ivf->nlist; ++L) out[L] = ivf->list_len[L];
  return FVDB_OK;
}
uint64_t fvdb_ivf_total_rows(fvdb_ivf* ivf) { return ivf->total_rows; }

// Copy one inverted list back to the host in list-position order (the save path of the chunked on-disk format,
// src/hybrid/persistence.rs:289-311 walks every inverted list).  fp16 rows are widened exactly.
int fvdb_ivf_list_export(fvdb_ivf* ivf, uint32_t list, float* rows, uint64_t* ids, uint8_t* live) {
  fvdb_ctx* ctx = ivf->ctx;
  if (list >= ivf->nlist) FAIL(ctx, FVDB_E_INVALID, "no such list");
  const uint32_t len = ivf->list_len[list];
  if (len == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const Pool& P = ivf->pool;
  const size_t bb = P.block_bytes();
  std::vector<uint8_t> blk(bb);
  uint64_t bid[64], valid = 0;
  const uint32_t d = ivf->d;
  for (uint32_t b = 0; b * 64 < len; ++b) {
    const uint32_t pb = ivf->list_blocks[list][b];
    HIPCHK(ctx, hipMemcpyAsync(blk.data(), (const char*)P.data + (size_t)pb * bb, bb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(bid, P.ids + (size_t)pb * 64, sizeof bid, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&valid, P.valid + pb, sizeof valid, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t rows_here = std::min<uint32_t>(64, len - b * 64);
    for (uint32_t lane = 0; lane < rows_here; ++lane) {
      const size_t r = (size_t)b * 64 + lane;
      if (rows) {
        float* out = rows + r * d;
        if (P.esize == 4) {  // [d4][64] chunks of 4 floats, lane = row
          const float* src = (const float*)blk.data();
          for (uint32_t j = 0; j < d; ++j) out[j] = src[((size_t)(j >> 2) * 64 + lane) * 4 + (j & 3)];
        } else {  // [d8][64] chunks of 8 halves
          const _Float16* src = (const _Float16*)blk.data();
          for (uint32_t j = 0; j < d; ++j) out[j] = (float)src[((size_t)(j >> 3) * 64 + lane) * 8 + (j & 7)];
        }
      }
      if (ids) ids[r] = bid[lane];
      if (live) live[r] = (uint8_t)((valid >> lane) & 1);
    }
  }
  return FVDB_OK;
}

int fvdb_ivf_set_global_list_sizes(fvdb_ivf* ivf, const uint64_t* sizes) {
  ivf->glob_blocks_host.resize(ivf->nlist);
  for (uint32_t L = 0; L < ivf->nlist; ++L) ivf->glob_blocks_host[L] = cdiv(sizes[L], 64);
  ivf->glob_set = true;
  ivf->table_dirty = true;
  return FVDB_OK;
}

// device-side assign: clusters for n rows already in HBM (row-major [n][d])
static int assign_dev(fvdb_ivf* ivf, const float* x_dev, uint64_t n, uint32_t* out_dev) {
  fvdb_ctx* ctx = ivf->ctx;
  const uint32_t step = 65536;
  for (uint64_t o = 0; o < n; o += step) {
    const uint32_t B = (uint32_t)std::min<uint64_t>(step, n - o);
    const float* qpad = nullptr;
    const Env E{ivf->ctx, ivf};
    int rc = padded_queries(ivf, E, x_dev + o * ivf->d, B, &qpad);
    if (rc) return rc;
    rc = run_coarse(ivf, E, qpad, B, 1, out_dev + o, nullptr);
    if (rc) return rc;
  }
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

int fvdb_ivf_assign(fvdb_ivf* ivf, const float* x, uint64_t n, uint32_t* out_cluster) {
  fvdb_ctx* ctx = ivf->ctx;
  if (!ivf->trained) FAIL(ctx, FVDB_E_NOT_TRAINED, "index not trained");
  if (n == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, x, n * ivf->d);
  if (rc) return rc;
  HIPCHK(ctx, ivf->s_in.ensure(n * ivf->d * 4));
  HIPCHK(ctx, ivf->s_clusters.ensure(n * 4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->s_in.p, x, n * ivf->d * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = assign_dev(ivf, ivf->s_in.as<float>(), n, ivf->s_clusters.as<uint32_t>());
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out_cluster, ivf->s_clusters.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

// rows already staged in s_in (device, row-major); clusters on host
static int append_staged(fvdb_ivf* ivf, const uint64_t* ids, uint64_t n, const uint32_t* cluster, uint32_t* out_pos) {
  fvdb_ctx* ctx = ivf->ctx;
  // count new blocks first so the pool grows once
  std::vector<uint32_t> add_len(ivf->nlist, 0);
  for (uint64_t i = 0; i < n; ++i) {
    if (cluster[i] >= ivf->nlist) FAIL(ctx, FVDB_E_INVALID, "cluster id out of range");
    add_len[cluster[i]]++;
  }
  uint64_t new_blocks = 0;
  for (uint32_t L = 0; L < ivf->nlist; ++L)
    new_blocks += cdiv((uint64_t)ivf->list_len[L] + add_len[L], 64) - ivf->list_blocks[L].size();
  if ((uint64_t)ivf->pool.used_blocks + new_blocks >= (1ull << 26)) FAIL(ctx, FVDB_E_UNSUPPORTED, "pool too large");
  int rc = ivf->pool.reserve(ctx, ivf->pool.used_blocks + (uint32_t)new_blocks);
  if (rc) return rc;
  std::vector<uint32_t> slots(n);
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t L = cluster[i];
    const uint32_t pos = ivf->list_len[L]++;
    if ((pos & 63) == 0) ivf->list_blocks[L].push_back(ivf->pool.used_blocks++);
    slots[i] = ivf->list_blocks[L][pos >> 6] * 64 + (pos & 63);
    if (out_pos) out_pos[i] = pos;
  }
  ivf->total_rows += n;
  ivf->table_dirty = true;
  HIPCHK(ctx, ivf->s_slots.ensure(n * 4));
  HIPCHK(ctx, ivf->s_ids.ensure(n * 8));
  HIPCHK(ctx, hipMemcpyAsync(ivf->s_slots.p, slots.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ids) HIPCHK(ctx, hipMemcpyAsync(ivf->s_ids.p, ids, n * 8, hipMemcpyHostToDevice, ctx->stream));
  if (ivf->f16) {
    const uint32_t d8 = ivf->dpad / 8;
    hipLaunchKernelGGL(scatter_rows_f16_kernel, dim3(cdiv(n * d8, 256)), dim3(256), 0, ctx->stream,
                       ivf->s_in.as<float>(), ivf->d, d8, n, ivf->s_slots.as<uint32_t>(),
                       ids ? ivf->s_ids.as<uint64_t>() : nullptr, ivf->pool.data, ivf->pool.ids,
                       (unsigned long long*)ivf->pool.valid);
  } else {
    const uint64_t threads = n * ivf->d4;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv(threads, 256)), dim3(256), 0, ctx->stream, ivf->s_in.as<float>(),
                       ivf->d, ivf->d4, n, ivf->s_slots.as<uint32_t>(), ids ? ivf->s_ids.as<uint64_t>() : nullptr,
                       (float4*)ivf->pool.data, ivf->pool.ids, (unsigned long long*)ivf->pool.valid, ivf->pool.half,
                       (float4*)ivf->pool.rm);
  }
  if (!ivf->d_xmax.p) {
    HIPCHK(ctx, ivf->d_xmax.ensure(4));
    HIPCHK(ctx, hipMemsetAsync(ivf->d_xmax.p, 0, 4, ctx->stream));
  }
  hipLaunchKernelGGL(pool_row_norms_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, ivf->s_in.as<float>(), ivf->d, n,
                     ivf->f16 ? 1 : 0, ivf->s_slots.as<uint32_t>(), ivf->pool.norms, ivf->d_xmax.as<uint32_t>());
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

int fvdb_ivf_add_assigned(fvdb_ivf* ivf, const float* x, const uint64_t* ids, uint64_t n, const uint32_t* cluster,
                          uint32_t* out_pos) {
  fvdb_ctx* ctx = ivf->ctx;
  if (!ivf->trained) FAIL(ctx, FVDB_E_NOT_TRAINED, "index not trained");
  if (n == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, x, n * ivf->d);
  if (rc) return rc;
  HIPCHK(ctx, ivf->s_in.ensure(n * ivf->d * 4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->s_in.p, x, n * ivf->d * 4, hipMemcpyHostToDevice, ctx->stream));
  return append_staged(ivf, ids, n, cluster, out_pos);
}

int fvdb_ivf_add(fvdb_ivf* ivf, const float* x, const uint64_t* ids, uint64_t n, uint32_t* out_cluster,
                 uint32_t* out_pos) {
  fvdb_ctx* ctx = ivf->ctx;
  if (!ivf->trained) FAIL(ctx, FVDB_E_NOT_TRAINED, "index not trained");
  if (n == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, x, n * ivf->d);
  if (rc) return rc;
  HIPCHK(ctx, ivf->s_in.ensure(n * ivf->d * 4));
  HIPCHK(ctx, ivf->s_clusters.ensure(n * 4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->s_in.p, x, n * ivf->d * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = assign_dev(ivf, ivf->s_in.as<float>(), n, ivf->s_clusters.as<uint32_t>());
  if (rc) return rc;
  std::vector<uint32_t> cl(n);
  HIPCHK(ctx, hipMemcpyAsync(cl.data(), ivf->s_clusters.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (out_cluster) std::memcpy(out_cluster, cl.data(), n * 4);
  return append_staged(ivf, ids, n, cl.data(), out_pos);
}

int fvdb_ivf_set_deleted(fvdb_ivf* ivf, const uint32_t* cluster, const uint32_t* pos, uint64_t n, int deleted) {
  fvdb_ctx* ctx = ivf->ctx;
  if (n == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> slots(n);
  for (uint64_t i = 0; i < n; ++i) {
    if (cluster[i] >= ivf->nlist || pos[i] >= ivf->list_len[cluster[i]]) FAIL(ctx, FVDB_E_NOT_FOUND, "no such row");
    slots[i] = ivf->list_blocks[cluster[i]][pos[i] >> 6] * 64 + (pos[i] & 63);
  }
  HIPCHK(ctx, ivf->s_slots.ensure(n * 4));
  HIPCHK(ctx, hipMemcpyAsync(ivf->s_slots.p, slots.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(set_valid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, ivf->s_slots.as<uint32_t>(), n,
                     deleted, (unsigned long long*)ivf->pool.valid);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

// given_probes (device, [B][min(nprobe, nlist)] cluster ids in probe order): the coarse stage is skipped.
// probes_only (device, same shape): only the coarse stage runs and its result is copied there.
static int search_common(fvdb_ivf* ivf, const Env& E, const float* q_dev, uint32_t B, uint32_t k, uint32_t nprobe, bool all,
                         uint64_t* out_ids, float* out_dist, uint32_t* out_counts, uint64_t* out_keys,
                         const uint32_t* given_probes = nullptr, uint32_t* probes_only = nullptr) {
  fvdb_ctx* ctx = E.ctx;
  IvfScratch& S = *E.S;
  if (!ivf->trained) FAIL(ctx, FVDB_E_NOT_TRAINED, "index not trained");
  if (k == 0 || k > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "k must be in 1..FVDB_MAX_K");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t np = all ? ivf->nlist : std::min(nprobe, ivf->nlist);
  if (np == 0) FAIL(ctx, FVDB_E_INVALID, "nprobe must be > 0");
  if (!all && np > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "nprobe above FVDB_MAX_K");
  int rc = upload_table(ivf);
  if (rc) return rc;
  std::lock_guard<std::mutex> enq(S.enq);  // one search's launches go in as a block
  if (ivf->ctx->profiling)
    for (auto& e : S.sev)
      if (!e) (void)hipEventCreate(&e);
  const uint32_t step = sub_batch(ivf, B, k, np);
  for (uint32_t o = 0; o < B; o += step) {
    const uint32_t b = std::min(step, B - o);
    const float* qpad = nullptr;
    rc = padded_queries(ivf, E, q_dev + (size_t)o * ivf->d, b, &qpad);
    if (rc) return rc;
    HIPCHK(ctx, S.s_probes.ensure((size_t)b * np * 4));
    const uint32_t* probes = S.s_probes.as<uint32_t>();
    if (all) {
      hipLaunchKernelGGL(probes_all_kernel, dim3(cdiv((uint64_t)b * np, 256)), dim3(256), 0, ctx->stream, b, np,
                         S.s_probes.as<uint32_t>());
      if (ivf->ctx->profiling) (void)hipEventRecord(S.sev[2], ctx->stream);
    } else if (given_probes) {
      probes = given_probes + (size_t)o * np;
      if (ivf->ctx->profiling) {  // no coarse stage in this call: zero-length stage intervals
        (void)hipEventRecord(S.sev[0], ctx->stream);
        (void)hipEventRecord(S.sev[1], ctx->stream);
        (void)hipEventRecord(S.sev[2], ctx->stream);
      }
    } else {
      rc = run_coarse(ivf, E, qpad, b, np, probes_only ? probes_only + (size_t)o * np : S.s_probes.as<uint32_t>(), nullptr);
      if (rc) return rc;
    }
    if (probes_only) continue;
    Env Eo = E;
    if (E.given_thr) Eo.given_thr = E.given_thr + o;  // agreed thresholds are indexed like the queries
    rc = run_fine(ivf, Eo, qpad, b, k, np, probes, out_ids ? out_ids + (size_t)o * k : nullptr,
                  out_dist ? out_dist + (size_t)o * k : nullptr, out_counts ? out_counts + o : nullptr,
                  out_keys ? out_keys + (size_t)o * k : nullptr, all ? ROLE_ALL : ROLE_LIST);
    if (rc) return rc;
    rc = finish_profile(ivf, E, !all, true);
    if (rc) return rc;
  }
  ivf->last_set.store(E.S);
  ivf->last_ctx.store(E.ctx);
  return FVDB_OK;
}

int fvdb_ivf_search_dev(fvdb_ivf* ivf, const float* q_dev, uint32_t B, uint32_t k, uint32_t nprobe,
                        uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev, uint64_t* out_keys_dev) {
  return search_common(ivf, Env{ivf->ctx, ivf}, q_dev, B, k, nprobe, false, out_ids_dev, out_dist_dev, out_counts_dev,
                       out_keys_dev);
}

// The explicit-slot entry points: the caller names the scratch set (slot) and the stream (`on`); nothing in the index
// object changes, so calls on different slots may come from different host threads at the same time.
static int slot_env(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, Env* E) {
  if (slot >= fvdb_ivf::kSlots) FAIL(ivf->ctx, FVDB_E_INVALID, "slot out of range");
  if (on && on->device != ivf->ctx->device) FAIL(ivf->ctx, FVDB_E_INVALID, "context of another device");
  E->ctx = on ? on : ivf->ctx;
  E->S = &slot_scratch(ivf, slot);
  return FVDB_OK;
}
static int slot_done(fvdb_ivf* ivf, const Env& E, int rc) {
  if (rc && E.ctx != ivf->ctx) {  // the failure text is read from the index's own context
    std::string m;
    {
      std::lock_guard<std::mutex> lk(E.ctx->err_mu);
      m = E.ctx->err;
    }
    ivf->ctx->set_err(m);
  }
  return rc;
}

int fvdb_ivf_coarse_dev_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t nprobe,
                             uint32_t* out_probes_dev) {
  if (!out_probes_dev) FAIL(ivf->ctx, FVDB_E_INVALID, "null output");
  Env E{};
  int rc = slot_env(ivf, on, slot, &E);
  if (rc) return rc;
  return slot_done(ivf, E, search_common(ivf, E, q_dev, B, 1, nprobe, false, nullptr, nullptr, nullptr, nullptr, nullptr,
                                         out_probes_dev));
}

int fvdb_ivf_search_probes_dev_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev,
                                    const uint32_t* probes_dev, uint32_t B, uint32_t k, uint32_t nprobe,
                                    uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                                    uint64_t* out_keys_dev) {
  if (!probes_dev) FAIL(ivf->ctx, FVDB_E_INVALID, "null probes");
  Env E{};
  int rc = slot_env(ivf, on, slot, &E);
  if (rc) return rc;
  return slot_done(ivf, E, search_common(ivf, E, q_dev, B, k, nprobe, false, out_ids_dev, out_dist_dev, out_counts_dev,
                                         out_keys_dev, probes_dev));
}

// the sharded step's variants (comm_sharded.h): thresholds shared between the ranks
static int shared_thresholds_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev, const uint32_t* probes_dev,
                                  uint32_t B, uint32_t k, uint32_t np, float* u_out) {
  Env E{};
  int rc = slot_env(ivf, on, slot, &E);
  if (rc) return rc;
  return slot_done(ivf, E, ivf_shared_thresholds(ivf, E, q_dev, probes_dev, B, k, np, u_out));
}
static int thr_combine_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* u_all, uint32_t W, uint32_t B,
                            float* thr_out, bool loopback_fill) {
  Env E{};
  int rc = slot_env(ivf, on, slot, &E);
  if (rc) return rc;
  return slot_done(ivf, E, ivf_thr_combine(ivf, E, u_all, W, B, thr_out, loopback_fill));
}
static int search_probes_thr_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev, const uint32_t* probes_dev,
                                  const float* thr_dev, uint32_t B, uint32_t k, uint32_t nprobe, uint64_t* out_ids_dev,
                                  float* out_dist_dev, uint32_t* out_counts_dev, uint64_t* out_keys_dev) {
  Env E{};
  int rc = slot_env(ivf, on, slot, &E);
  if (rc) return rc;
  E.given_thr = thr_dev;
  return slot_done(ivf, E, search_common(ivf, E, q_dev, B, k, nprobe, false, out_ids_dev, out_dist_dev, out_counts_dev,
                                         out_keys_dev, probes_dev));
}

int fvdb_ivf_search_dev_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t k,
                             uint32_t nprobe, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                             uint64_t* out_keys_dev) {
  // every launch goes to `on`'s stream and touches only this slot's scratch, so slots can be in flight together
  Env E{};
  int rc = slot_env(ivf, on, slot, &E);
  if (rc) return rc;
  return slot_done(ivf, E, search_common(ivf, E, q_dev, B, k, nprobe, false, out_ids_dev, out_dist_dev, out_counts_dev,
                                         out_keys_dev));
}

int fvdb_ivf_search_all_dev(fvdb_ivf* ivf, const float* q_dev, uint32_t B, uint32_t k, uint64_t* out_ids_dev,
                            float* out_dist_dev, uint32_t* out_counts_dev) {
  return search_common(ivf, Env{ivf->ctx, ivf}, q_dev, B, k, 0, true, out_ids_dev, out_dist_dev, out_counts_dev, nullptr);
}

// A leased scratch set + stream for one blocking search: any number of host threads may call the host-pointer
// entry points on one index; each call takes a free set (or waits for one) and gives it back when it returns.
// With stage profiling on, the call runs on the index's own stream with set 0 instead (a measuring run is one thread).
extern "C++" {
namespace {
struct Lease {
  fvdb_ivf* ivf;
  int idx = -1;
  Env E{};
  int rc = FVDB_OK;
  explicit Lease(fvdb_ivf* ivf_) : ivf(ivf_) {
    if (ivf->ctx->profiling) {
      E = Env{ivf->ctx, ivf};
      return;
    }
    std::unique_lock<std::mutex> lk(ivf->mu);
    ivf->lease_cv.wait(lk, [&] { return ivf->lease_busy != (1u << fvdb_ivf::kLeases) - 1u; });
    for (uint32_t i = 0; i < fvdb_ivf::kLeases; ++i)
      if (!(ivf->lease_busy & (1u << i))) {
        idx = (int)i;
        break;
      }
    ivf->lease_busy |= 1u << idx;
    if (!ivf->lease_ctx[idx]) {
      rc = fvdb_ctx_create(ivf->ctx->device, &ivf->lease_ctx[idx]);
      if (rc) ivf->ctx->set_err("could not create a stream for a concurrent search");
    }
    E = Env{ivf->lease_ctx[idx], &ivf->lease_set[idx]};
  }
  ~Lease() {
    if (idx < 0) return;
    {
      std::lock_guard<std::mutex> lk(ivf->mu);
      ivf->lease_busy &= ~(1u << idx);
    }
    ivf->lease_cv.notify_one();
  }
};
}  // namespace
}  // extern "C++"

static int search_host(fvdb_ivf* ivf, const float* q, uint32_t B, uint32_t k, uint32_t nprobe, bool all,
                       uint64_t* out_ids, float* out_dist, uint32_t* out_counts) {
  if (!ivf->trained) FAIL(ivf->ctx, FVDB_E_NOT_TRAINED, "index not trained");
  if (B == 0) return FVDB_OK;
  if (k == 0 || k > FVDB_MAX_K) FAIL(ivf->ctx, FVDB_E_UNSUPPORTED, "k must be in 1..FVDB_MAX_K");
  int rc = check_finite(ivf->ctx, q, (uint64_t)B * ivf->d);
  if (rc) return rc;
  Lease L(ivf);
  if (L.rc) return L.rc;
  fvdb_ctx* ctx = L.E.ctx;
  IvfScratch& S = *L.E.S;
  auto run = [&]() -> int {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, S.s_in.ensure((size_t)B * ivf->d * 4));
    HIPCHK(ctx, S.s_out_ids.ensure((size_t)B * k * 8));
    HIPCHK(ctx, S.s_out_dist.ensure((size_t)B * k * 4));
    HIPCHK(ctx, S.s_out_cnt.ensure((size_t)B * 4));
    HIPCHK(ctx, hipMemcpyAsync(S.s_in.p, q, (size_t)B * ivf->d * 4, hipMemcpyHostToDevice, ctx->stream));
    int r = search_common(ivf, L.E, S.s_in.as<float>(), B, k, nprobe, all, S.s_out_ids.as<uint64_t>(),
                          S.s_out_dist.as<float>(), S.s_out_cnt.as<uint32_t>(), nullptr);
    if (r) return r;
    HIPCHK(ctx, hipMemcpyAsync(out_ids, S.s_out_ids.p, (size_t)B * k * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out_dist, S.s_out_dist.p, (size_t)B * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_counts)
      HIPCHK(ctx, hipMemcpyAsync(out_counts, S.s_out_cnt.p, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FVDB_OK;
  };
  return slot_done(ivf, L.E, run());
}

int fvdb_ivf_search(fvdb_ivf* ivf, const float* q, uint32_t B, uint32_t k, uint32_t nprobe, uint64_t* out_ids,
                    float* out_dist, uint32_t* out_counts) {
  return search_host(ivf, q, B, k, nprobe, false, out_ids, out_dist, out_counts);
}
int fvdb_ivf_search_all(fvdb_ivf* ivf, const float* q, uint32_t B, uint32_t k, uint64_t* out_ids, float* out_dist,
                        uint32_t* out_counts) {
  return search_host(ivf, q, B, k, 0, true, out_ids, out_dist, out_counts);
}

int fvdb_ivf_coarse(fvdb_ivf* ivf, const float* q, uint32_t B, uint32_t nprobe, uint32_t* out_clusters,
                    float* out_dist) {
  if (!ivf->trained) FAIL(ivf->ctx, FVDB_E_NOT_TRAINED, "index not trained");
  if (B == 0) return FVDB_OK;
  const uint32_t np = std::min(nprobe, ivf->nlist);
  if (np == 0 || np > FVDB_MAX_K) FAIL(ivf->ctx, FVDB_E_UNSUPPORTED, "nprobe must be in 1..FVDB_MAX_K");
  int rc = check_finite(ivf->ctx, q, (uint64_t)B * ivf->d);
  if (rc) return rc;
  Lease L(ivf);
  if (L.rc) return L.rc;
  fvdb_ctx* ctx = L.E.ctx;
  IvfScratch& S = *L.E.S;
  auto run = [&]() -> int {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> enq(S.enq);
    HIPCHK(ctx, S.s_in.ensure((size_t)B * ivf->d * 4));
    HIPCHK(ctx, S.s_probes.ensure((size_t)B * np * 4));
    HIPCHK(ctx, S.s_cdist.ensure((size_t)B * np * 4));
    HIPCHK(ctx, hipMemcpyAsync(S.s_in.p, q, (size_t)B * ivf->d * 4, hipMemcpyHostToDevice, ctx->stream));
    const float* qpad = nullptr;
    int r = padded_queries(ivf, L.E, S.s_in.as<float>(), B, &qpad);
    if (r) return r;
    if (ivf->ctx->profiling)
      for (auto& e : S.sev)
        if (!e) (void)hipEventCreate(&e);
    r = run_coarse(ivf, L.E, qpad, B, np, S.s_probes.as<uint32_t>(), S.s_cdist.as<float>());
    if (r) return r;
    HIPCHK(ctx, hipMemcpyAsync(out_clusters, S.s_probes.p, (size_t)B * np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_dist)
      HIPCHK(ctx, hipMemcpyAsync(out_dist, S.s_cdist.p, (size_t)B * np * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FVDB_OK;
  };
  return slot_done(ivf, L.E, run());
}

int fvdb_ivf_set_coarse_mode(fvdb_ivf* ivf, int mode) {
  if (!ivf) return FVDB_E_INVALID;
  if (mode != FVDB_COARSE_AUTO && mode != FVDB_COARSE_EXACT) FAIL(ivf->ctx, FVDB_E_INVALID, "unknown coarse mode");
  ivf->coarse_mode = mode;
  return FVDB_OK;
}

int fvdb_ivf_set_scan_mode(fvdb_ivf* ivf, int mode) {
  if (!ivf) return FVDB_E_INVALID;
  if (mode != FVDB_SCAN_AUTO && mode != FVDB_SCAN_EXACT && mode != FVDB_SCAN_FILTER) FAIL(ivf->ctx, FVDB_E_INVALID, "unknown scan mode");
  ivf->scan_mode = mode;
  return FVDB_OK;
}

// the diagnostic entry points below describe the most recent search on the index, whichever scratch set it ran with
// (call them with no search in flight)
static IvfScratch& last_scratch(fvdb_ivf* ivf) {
  IvfScratch* S = ivf->last_set.load();
  return S ? *S : *ivf;
}

int fvdb_ivf_scan_survivors(fvdb_ivf* ivf, uint32_t* out, uint32_t B) {
  fvdb_ctx* ctx = ivf->ctx;
  IvfScratch& S = last_scratch(ivf);
  if (!S.s_scnt.p || S.s_scnt.cap < (size_t)B * 4) FAIL(ctx, FVDB_E_INVALID, "no matrix-core scan of that size has run");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  HIPCHK(ctx, hipMemcpyAsync(out, S.s_scnt.p, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

int fvdb_ivf_scan_survivor_dump(fvdb_ivf* ivf, uint32_t query, uint32_t max_n, uint32_t* rank, uint32_t* pos, float* v,
                                uint32_t* n_out) {
  fvdb_ctx* ctx = ivf->ctx;
  *n_out = 0;
  IvfScratch& S = last_scratch(ivf);
  if (!S.s_scnt.p || !S.s_surv.p || S.s_scnt.cap < (size_t)(query + 1) * 4)
    FAIL(ctx, FVDB_E_INVALID, "no matrix-core scan holding that query has run");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  uint32_t cnt = 0;
  HIPCHK(ctx, hipMemcpyAsync(&cnt, S.s_scnt.as<uint32_t>() + query, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t n = std::min(std::min(cnt, kMfmaCmax), max_n);
  std::vector<uint32_t> sv((size_t)n * 2);
  if (n) {
    HIPCHK(ctx, hipMemcpyAsync(sv.data(), (const char*)S.s_surv.p + (size_t)query * kMfmaCmax * 8, (size_t)n * 8,
                               hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(v, S.s_sdist.as<float>() + (size_t)query * kMfmaCmax, (size_t)n * 4, hipMemcpyDeviceToHost,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  for (uint32_t i = 0; i < n; ++i) {
    rank[i] = sv[2 * i];
    pos[i] = sv[2 * i + 1];
  }
  *n_out = n;
  return FVDB_OK;
}

int fvdb_ivf_scan_fallbacks(fvdb_ivf* ivf, uint64_t* out) {
  fvdb_ctx* ctx = ivf->ctx;
  *out = 0;
  if (!ivf->s_fallbacks.p) return FVDB_OK;
  uint32_t v = 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(&v, ivf->s_fallbacks.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *out = v;
  return FVDB_OK;
}

int fvdb_ivf_scan_fallback_reasons(fvdb_ivf* ivf, uint64_t* out5) {
  fvdb_ctx* ctx = ivf->ctx;
  for (int i = 0; i < 5; ++i) out5[i] = 0;
  if (!ivf->s_fallbacks.p) return FVDB_OK;
  uint32_t v[5] = {0, 0, 0, 0, 0};
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(v, ivf->s_fallbacks.as<uint32_t>() + 2, 20, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 5; ++i) out5[i] = v[i];
  return FVDB_OK;
}

int fvdb_ivf_coarse_fallbacks(fvdb_ivf* ivf, uint64_t* out) {
  fvdb_ctx* ctx = ivf->ctx;
  *out = 0;
  if (!ivf->s_fallbacks.p) return FVDB_OK;
  uint32_t v = 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(&v, ivf->s_fallbacks.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *out = v;
  return FVDB_OK;
}

int fvdb_ivf_last_stats(fvdb_ivf* ivf, fvdb_search_stats* out) {
  fvdb_ctx* ctx = ivf->ctx;
  IvfScratch& S = last_scratch(ivf);
  if (!S.s_scalars.p) {
    std::memset(out, 0, sizeof(*out));
    return FVDB_OK;
  }
  unsigned long long st[3];
  HIPCHK(ctx, hipDeviceSynchronize());
  HIPCHK(ctx, hipMemcpyAsync(st, S.s_scalars.as<uint32_t>() + 4, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  out->rows_scanned = st[0];
  out->work_items = st[1];
  out->list_rows_touched = st[2];
  return FVDB_OK;
}

int fvdb_ivf_profile_collect(fvdb_ivf* ivf) {
  IvfScratch* S = ivf->last_set.load();  // the most recent search, whichever scratch set and stream it used
  fvdb_ctx* c = ivf->last_ctx.load();
  if (!S || !c || !S->pending_profile) return FVDB_OK;
  S->collecting = true;
  int rc = finish_profile(ivf, Env{c, S}, S->pend_coarse, S->pend_fine);
  S->collecting = false;
  S->pending_profile = false;
  return rc;
}

// stage times accumulated while profiling is on: ms[5] = coarse scan, coarse merge, plan, fine scan,
// fine merge; returns the number of searches accumulated and resets.
uint64_t fvdb_ivf_stage_times(fvdb_ivf* ivf, float* ms_out) {
  for (int i = 0; i < 8; ++i) {
    ms_out[i] = ivf->stage_ms[i];
    ivf->stage_ms[i] = 0;
  }
  const uint64_t c = ivf->stage_calls;
  ivf->stage_calls = 0;
  return c;
}

// =============================================================================================
// k-means training on the GPU (src/ivf/core.rs:240-429)
// =============================================================================================
int fvdb_ivf_train(fvdb_ivf* ivf, const float* x, uint64_t n, uint32_t max_iterations, uint64_t seed,
                   fvdb_train_result* out) {
  fvdb_ctx* ctx = ivf->ctx;
  const uint32_t nlist = ivf->nlist, d = ivf->d;
  if (n == 0 || n < nlist) FAIL(ctx, FVDB_E_INSUFFICIENT, "insufficient training data");
  if (max_iterations == 0) FAIL(ctx, FVDB_E_INVALID, "max_iterations must be > 0");
  if (d > 2048) FAIL(ctx, FVDB_E_UNSUPPORTED, "training supports d <= 2048");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, x, n * d);
  if (rc) return rc;
  DBuf dx, dmind, dassign, dnew, ddist, dsc;
  auto cleanup = [&]() {
    dx.release(); dmind.release(); dassign.release(); dnew.release(); ddist.release(); dsc.release();
  };
#define TCHK(call)                                                        \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) {                                               \
      ctx->set_err(std::string(#call) + ": " + hipGetErrorString(e_));    \
      cleanup();                                                          \
      return e_ == hipErrorOutOfMemory ? FVDB_E_OOM : FVDB_E_HIP;         \
    }                                                                     \
  } while (0)
  TCHK(dx.ensure(n * d * 4));
  TCHK(dmind.ensure(n * 4));
  TCHK(dassign.ensure(n * 4));
  TCHK(dnew.ensure(n * 4));
  TCHK(ddist.ensure(n * 4));
  TCHK(dsc.ensure(64));
  TCHK(ivf->d_centroids_rm.ensure((size_t)nlist * d * 4));
  TCHK(hipMemcpyAsync(dx.p, x, n * d * 4, hipMemcpyHostToDevice, ctx->stream));
  float* cent = ivf->d_centroids_rm.as<float>();
  const uint32_t gn = cdiv(n, 256);

  // ---- k-means++ seeding (:336-371); draws from SplitMix64(seed) ----
  struct {
    uint64_t s;
    uint64_t next() {
      uint64_t z = (s += 0x9E3779B97F4A7C15ull);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      return z ^ (z >> 31);
    }
  } rng{seed};
  uint64_t pick = rng.next() % n;
  TCHK(hipMemcpyAsync(cent, dx.as<float>() + pick * d, (size_t)d * 4, hipMemcpyDeviceToDevice, ctx->stream));
  hipLaunchKernelGGL(fill_f32_kernel, dim3(gn), dim3(256), 0, ctx->stream, dmind.as<float>(), n,
                     __builtin_huge_valf());
  uint32_t chosen = 1;
  for (uint32_t i = 1; i < nlist; ++i) {
    hipLaunchKernelGGL(kpp_min_dist_kernel, dim3(gn), dim3(256), 0, ctx->stream, dx.as<float>(), d, n, pick,
                       dmind.as<float>());
    const float u = (float)(rng.next() >> 40) * (1.0f / 16777216.0f);
    hipLaunchKernelGGL(kpp_pick_kernel, dim3(1), dim3(256), 0, ctx->stream, dmind.as<float>(), n, u,
                       (unsigned long long*)dsc.p);
    unsigned long long p = 0;
    TCHK(hipMemcpyAsync(&p, dsc.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    TCHK(hipStreamSynchronize(ctx->stream));
    if (p == ~0ull) continue;  // reference: the loop never fired, no centroid pushed this round
    pick = p;
    TCHK(hipMemcpyAsync(cent + (size_t)chosen * d, dx.as<float>() + pick * d, (size_t)d * 4,
                        hipMemcpyDeviceToDevice, ctx->stream));
    chosen++;
  }
  if (chosen < nlist) {
    cleanup();
    FAIL(ctx, FVDB_E_INVALID, "k-means++ produced fewer centroids than n_clusters (degenerate data)");
  }

  auto error_of = [&](float* out_err) -> int {
    hipLaunchKernelGGL(kmeans_point_dist_kernel, dim3(gn), dim3(256), 0, ctx->stream, dx.as<float>(), d, n,
                       dassign.as<uint32_t>(), cent, ddist.as<float>());
    hipLaunchKernelGGL(seq_sqsum_mean_kernel, dim3(1), dim3(256), 0, ctx->stream, ddist.as<float>(), n,
                       (float*)dsc.p + 4);
    float r[2];
    HIPCHK(ctx, hipMemcpyAsync(r, (float*)dsc.p + 4, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out_err = r[0];
    return FVDB_OK;
  };

  TCHK(hipMemsetAsync(dassign.p, 0, n * 4, ctx->stream));  // assignments start at ClusterId(0) (:280)
  float prev_error = __builtin_huge_valf(), initial_error = 0, final_error = 0;
  rc = error_of(&initial_error);
  if (rc) { cleanup(); return rc; }
  bool converged = false;
  uint32_t iterations = 0;
  for (uint32_t iter = 0; iter < max_iterations; ++iter) {
    iterations = iter + 1;
    rc = install_centroids(ivf, cent);
    if (rc) { cleanup(); return rc; }
    rc = assign_dev(ivf, dx.as<float>(), n, dnew.as<uint32_t>());
    if (rc) { cleanup(); return rc; }
    TCHK(hipMemsetAsync(dsc.p, 0, 4, ctx->stream));
    hipLaunchKernelGGL(count_changed_kernel, dim3(gn), dim3(256), 0, ctx->stream, dnew.as<uint32_t>(),
                       dassign.as<uint32_t>(), n, (uint32_t*)dsc.p);
    uint32_t changed = 0;
    TCHK(hipMemcpyAsync(&changed, dsc.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(nlist), dim3(256), 0, ctx->stream, dx.as<float>(), d, n,
                       dassign.as<uint32_t>(), cent);
    TCHK(hipStreamSynchronize(ctx->stream));
    if (iterations >= max_iterations) break;
    float cur = 0;
    rc = error_of(&cur);
    if (rc) { cleanup(); return rc; }
    const float change = std::fabs(prev_error - cur) / prev_error;
    if (changed == 0 || change < 1e-4f) {
      converged = true;
      if (max_iterations == 10 && n < 20) {  // reference's small-test special case (:313-317)
        prev_error = cur;
        continue;
      }
      break;
    }
    prev_error = cur;
  }
  rc = error_of(&final_error);
  if (rc) { cleanup(); return rc; }
  ivf->h_centroids.resize((size_t)nlist * d);
  TCHK(hipMemcpyAsync(ivf->h_centroids.data(), cent, (size_t)nlist * d * 4, hipMemcpyDeviceToHost, ctx->stream));
  TCHK(hipStreamSynchronize(ctx->stream));
  rc = install_centroids(ivf, cent);
  cleanup();
  if (rc) return rc;
  reset_lists(ivf);
  if (ivf->pool.valid && ivf->pool.cap_blocks)
    HIPCHK(ctx, hipMemsetAsync(ivf->pool.valid, 0, (size_t)ivf->pool.cap_blocks * 8, ctx->stream));
  if (out) {
    out->iterations = iterations;
    out->converged = converged ? 1 : 0;
    out->initial_error = initial_error;
    out->final_error = final_error;
  }
#undef TCHK
  return FVDB_OK;
}

// =============================================================================================
// merge of per-shard partial results
// =============================================================================================
int fvdb_merge_keys_dev(fvdb_ctx* ctx, const uint64_t* keys, const uint64_t* ids, uint32_t G, uint32_t B, uint32_t k,
                        uint64_t* out_ids, float* out_dist, uint32_t* out_counts) {
  if (k == 0 || k > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "k must be in 1..FVDB_MAX_K");
  if (B == 0 || G == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t grid = cdiv(B, 4);
  switch (kr_for(k)) {
    case 1: hipLaunchKernelGGL((merge_keys_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, keys, ids, G, B, k, out_ids, out_dist, out_counts); break;
    case 2: hipLaunchKernelGGL((merge_keys_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, keys, ids, G, B, k, out_ids, out_dist, out_counts); break;
    default: hipLaunchKernelGGL((merge_keys_kernel<4>), dim3(grid), dim3(256), 0, ctx->stream, keys, ids, G, B, k, out_ids, out_dist, out_counts); break;
  }
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

// =============================================================================================
// top-k / merge utilities (a4): src/core/vector_ops.rs:12-32,180-263, src/core/types.rs:206-223
// =============================================================================================
static int check_no_nan(fvdb_ctx* ctx, const float* x, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i)
    if (x[i] != x[i]) FAIL(ctx, FVDB_E_NONFINITE, "NaN score (the reference panics in partial_cmp().unwrap())");
  return FVDB_OK;
}

int fvdb_top_k_indices_dev(fvdb_ctx* ctx, const float* scores_dev, uint32_t B, uint64_t n, uint32_t k, int heap,
                           uint64_t* out_idx_dev, uint32_t* out_counts_dev) {
  if (!ctx || !out_idx_dev || !out_counts_dev) return FVDB_E_INVALID;
  if (k > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "k above FVDB_MAX_K");
  if (n >= 0xFFFFFFFFull) FAIL(ctx, FVDB_E_UNSUPPORTED, "rows longer than 2^32-2 scores");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (k == 0) {  // :181-183 `if k == 0 { return vec![] }`; take(0)
    HIPCHK(ctx, hipMemsetAsync(out_counts_dev, 0, (size_t)B * 4, ctx->stream));
    return FVDB_OK;
  }
  if (heap) {
    const size_t lds = (size_t)(k + 64) * sizeof(UItem) + 16;
    hipLaunchKernelGGL((topk_heap_kernel<false>), dim3(B), dim3(64), lds, ctx->stream, scores_dev, (const uint64_t*)nullptr, B,
                       (uint32_t)n, k, out_idx_dev, (float*)nullptr, out_counts_dev);
  } else {
    const uint32_t grid = cdiv(B, 4);
    switch (kr_for(k)) {
      case 1: hipLaunchKernelGGL((topk_sort_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, scores_dev, B, (uint32_t)n, k, out_idx_dev, out_counts_dev); break;
      case 2: hipLaunchKernelGGL((topk_sort_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, scores_dev, B, (uint32_t)n, k, out_idx_dev, out_counts_dev); break;
      default: hipLaunchKernelGGL((topk_sort_kernel<4>), dim3(grid), dim3(256), 0, ctx->stream, scores_dev, B, (uint32_t)n, k, out_idx_dev, out_counts_dev); break;
    }
  }
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

int fvdb_streaming_top_k_dev(fvdb_ctx* ctx, const uint64_t* ids_dev, const float* scores_dev, uint32_t B, uint64_t n,
                             uint32_t k, uint64_t* out_ids_dev, float* out_scores_dev, uint32_t* out_counts_dev) {
  if (!ctx || !out_ids_dev || !out_counts_dev) return FVDB_E_INVALID;
  if (k > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "k above FVDB_MAX_K");
  if (n >= 0xFFFFFFFFull) FAIL(ctx, FVDB_E_UNSUPPORTED, "rows longer than 2^32-2 scores");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (k == 0) {
    HIPCHK(ctx, hipMemsetAsync(out_counts_dev, 0, (size_t)B * 4, ctx->stream));
    return FVDB_OK;
  }
  const size_t lds = (size_t)(k + 64) * sizeof(UItem) + 16;
  hipLaunchKernelGGL((topk_heap_kernel<true>), dim3(B), dim3(64), lds, ctx->stream, scores_dev, ids_dev, B, (uint32_t)n, k,
                     out_ids_dev, out_scores_dev, out_counts_dev);
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

int fvdb_merge_search_results_dev(fvdb_ctx* ctx, const uint64_t* ids_dev, const float* dist_dev, uint32_t B, uint64_t n,
                                  uint32_t k, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev) {
  if (!ctx || !out_ids_dev || !out_dist_dev || !out_counts_dev) return FVDB_E_INVALID;
  if (k > FVDB_MAX_K) FAIL(ctx, FVDB_E_UNSUPPORTED, "k above FVDB_MAX_K");
  if (n >= (1ull << 20)) FAIL(ctx, FVDB_E_UNSUPPORTED, "more than 2^20 results per query");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (k == 0) {
    HIPCHK(ctx, hipMemsetAsync(out_counts_dev, 0, (size_t)B * 4, ctx->stream));
    return FVDB_OK;
  }
  const uint32_t grid = cdiv(B, 4);
  switch (kr_for(k)) {
    case 1: hipLaunchKernelGGL((merge_dedup_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, ids_dev, dist_dev, B, (uint32_t)n, k, out_ids_dev, out_dist_dev, out_counts_dev); break;
    case 2: hipLaunchKernelGGL((merge_dedup_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, ids_dev, dist_dev, B, (uint32_t)n, k, out_ids_dev, out_dist_dev, out_counts_dev); break;
    default: hipLaunchKernelGGL((merge_dedup_kernel<4>), dim3(grid), dim3(256), 0, ctx->stream, ids_dev, dist_dev, B, (uint32_t)n, k, out_ids_dev, out_dist_dev, out_counts_dev); break;
  }
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

// host-pointer forms: stage, run the device form, copy back
namespace {
struct UtilBufs {
  void *a = nullptr, *b = nullptr, *o1 = nullptr, *o2 = nullptr, *oc = nullptr;
  ~UtilBufs() {
    for (void* p : {a, b, o1, o2, oc})
      if (p) (void)hipFree(p);
  }
};
}  // namespace

static int top_k_host(fvdb_ctx* ctx, const float* scores, uint32_t B, uint64_t n, uint32_t k, int heap, uint64_t* out_idx,
                      uint32_t* out_counts) {
  if (!ctx || !scores || !out_counts || (!out_idx && k)) return FVDB_E_INVALID;
  if (B == 0) return FVDB_OK;
  int rc = check_no_nan(ctx, scores, (uint64_t)B * n);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  UtilBufs u;
  HIPCHK(ctx, hipMalloc(&u.a, std::max<size_t>((size_t)B * n * 4, 16)));
  HIPCHK(ctx, hipMalloc(&u.o1, std::max<size_t>((size_t)B * k * 8, 16)));
  HIPCHK(ctx, hipMalloc(&u.oc, (size_t)B * 4));
  HIPCHK(ctx, hipMemcpyAsync(u.a, scores, (size_t)B * n * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = fvdb_top_k_indices_dev(ctx, (const float*)u.a, B, n, k, heap, (uint64_t*)u.o1, (uint32_t*)u.oc);
  if (rc) return rc;
  if (k) HIPCHK(ctx, hipMemcpyAsync(out_idx, u.o1, (size_t)B * k * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_counts, u.oc, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}
int fvdb_top_k_indices(fvdb_ctx* ctx, const float* scores, uint32_t B, uint64_t n, uint32_t k, uint64_t* out_idx,
                       uint32_t* out_counts) {
  return top_k_host(ctx, scores, B, n, k, 0, out_idx, out_counts);
}
int fvdb_top_k_indices_heap(fvdb_ctx* ctx, const float* scores, uint32_t B, uint64_t n, uint32_t k, uint64_t* out_idx,
                            uint32_t* out_counts) {
  return top_k_host(ctx, scores, B, n, k, 1, out_idx, out_counts);
}

static int pairs_host(fvdb_ctx* ctx, const uint64_t* ids, const float* vals, uint32_t B, uint64_t n, uint32_t k, int merge,
                      uint64_t* out_ids, float* out_vals, uint32_t* out_counts) {
  if (!ctx || !ids || !vals || !out_counts || ((!out_ids || !out_vals) && k)) return FVDB_E_INVALID;
  if (B == 0) return FVDB_OK;
  int rc = check_no_nan(ctx, vals, (uint64_t)B * n);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  UtilBufs u;
  HIPCHK(ctx, hipMalloc(&u.a, std::max<size_t>((size_t)B * n * 8, 16)));
  HIPCHK(ctx, hipMalloc(&u.b, std::max<size_t>((size_t)B * n * 4, 16)));
  HIPCHK(ctx, hipMalloc(&u.o1, std::max<size_t>((size_t)B * k * 8, 16)));
  HIPCHK(ctx, hipMalloc(&u.o2, std::max<size_t>((size_t)B * k * 4, 16)));
  HIPCHK(ctx, hipMalloc(&u.oc, (size_t)B * 4));
  HIPCHK(ctx, hipMemcpyAsync(u.a, ids, (size_t)B * n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(u.b, vals, (size_t)B * n * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = merge ? fvdb_merge_search_results_dev(ctx, (const uint64_t*)u.a, (const float*)u.b, B, n, k, (uint64_t*)u.o1,
                                             (float*)u.o2, (uint32_t*)u.oc)
             : fvdb_streaming_top_k_dev(ctx, (const uint64_t*)u.a, (const float*)u.b, B, n, k, (uint64_t*)u.o1,
                                        (float*)u.o2, (uint32_t*)u.oc);
  if (rc) return rc;
  if (k) {
    HIPCHK(ctx, hipMemcpyAsync(out_ids, u.o1, (size_t)B * k * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out_vals, u.o2, (size_t)B * k * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipMemcpyAsync(out_counts, u.oc, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}
int fvdb_streaming_top_k(fvdb_ctx* ctx, const uint64_t* ids, const float* scores, uint32_t B, uint64_t n, uint32_t k,
                         uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  return pairs_host(ctx, ids, scores, B, n, k, 0, out_ids, out_scores, out_counts);
}
int fvdb_merge_search_results(fvdb_ctx* ctx, const uint64_t* ids, const float* dist, uint32_t B, uint64_t n, uint32_t k,
                              uint64_t* out_ids, float* out_dist, uint32_t* out_counts) {
  return pairs_host(ctx, ids, dist, B, n, k, 1, out_ids, out_dist, out_counts);
}

// =============================================================================================
// row store + candidate scoring
// =============================================================================================
int fvdb_store_create(fvdb_ctx* ctx, uint32_t d, uint64_t capacity_rows, fvdb_store** out) {
  if (!ctx || !out) return FVDB_E_INVALID;
  *out = nullptr;
  if (d == 0) FAIL(ctx, FVDB_E_INVALID, "d must be > 0");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  fvdb_store* s = new (std::nothrow) fvdb_store();
  if (!s) return FVDB_E_OOM;
  s->ctx = ctx;
  s->d = d;
  s->dpad = ((d + 3) / 4) * 4;
  s->cap = std::max<uint64_t>(capacity_rows, 64);
  hipError_t e = hipMalloc(&s->data, s->cap * s->dpad * 4);
  if (e != hipSuccess) {
    delete s;
    FAIL(ctx, FVDB_E_OOM, "store allocation failed");
  }
  *out = s;
  return FVDB_OK;
}

void fvdb_store_destroy(fvdb_store* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->data) (void)hipFree(s->data);
  s->s_q.release();
  s->s_cand.release();
  s->s_out.release();
  s->s_in.release();
  delete s;
}

uint64_t fvdb_store_rows(fvdb_store* s) { return s->rows; }

int fvdb_store_append(fvdb_store* s, const float* rows, uint64_t n, uint64_t* first_row) {
  fvdb_ctx* ctx = s->ctx;
  if (first_row) *first_row = s->rows;
  if (n == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, rows, n * s->d);
  if (rc) return rc;
  if (s->rows + n >= 0xFFFFFFFFull) FAIL(ctx, FVDB_E_UNSUPPORTED, "store limited to 2^32-1 rows");
  if (s->rows + n > s->cap) {
    uint64_t ncap = std::max<uint64_t>(s->rows + n, s->cap + s->cap / 2);
    float* nd = nullptr;
    HIPCHK(ctx, hipMalloc(&nd, ncap * s->dpad * 4));
    if (s->rows)
      HIPCHK(ctx, hipMemcpyAsync(nd, s->data, s->rows * s->dpad * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(s->data);
    s->data = nd;
    s->cap = ncap;
  }
  float* dst = s->data + s->rows * s->dpad;
  if (s->d == s->dpad) {
    HIPCHK(ctx, hipMemcpyAsync(dst, rows, n * s->d * 4, hipMemcpyHostToDevice, ctx->stream));
  } else {
    HIPCHK(ctx, s->s_in.ensure(n * s->d * 4));
    HIPCHK(ctx, hipMemcpyAsync(s->s_in.p, rows, n * s->d * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv(n * s->dpad, 256)), dim3(256), 0, ctx->stream, s->s_in.as<float>(),
                       s->d, s->dpad, n, dst);
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  s->rows += n;
  return FVDB_OK;
}

int fvdb_store_get(fvdb_store* s, uint64_t row, float* out) {
  fvdb_ctx* ctx = s->ctx;
  if (row >= s->rows) FAIL(ctx, FVDB_E_NOT_FOUND, "row out of range");
  HIPCHK(ctx, hipMemcpyAsync(out, s->data + row * s->dpad, (size_t)s->d * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

int fvdb_score_candidates(fvdb_store* s, const float* q, uint32_t B, const uint32_t* cand, uint32_t C, float* out) {
  fvdb_ctx* ctx = s->ctx;
  if (B == 0 || C == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, q, (uint64_t)B * s->d);
  if (rc) return rc;
  for (uint64_t i = 0; i < (uint64_t)B * C; ++i)
    if (cand[i] != FVDB_NO_ROW && cand[i] >= s->rows) FAIL(ctx, FVDB_E_NOT_FOUND, "candidate row out of range");
  HIPCHK(ctx, s->s_in.ensure((size_t)B * s->d * 4));
  HIPCHK(ctx, s->s_q.ensure((size_t)B * s->dpad * 4));
  HIPCHK(ctx, s->s_cand.ensure((size_t)B * C * 4));
  HIPCHK(ctx, s->s_out.ensure((size_t)B * C * 4));
  const float* qd = nullptr;
  if (s->d == s->dpad) {
    HIPCHK(ctx, hipMemcpyAsync(s->s_q.p, q, (size_t)B * s->d * 4, hipMemcpyHostToDevice, ctx->stream));
  } else {
    HIPCHK(ctx, hipMemcpyAsync(s->s_in.p, q, (size_t)B * s->d * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv((uint64_t)B * s->dpad, 256)), dim3(256), 0, ctx->stream,
                       s->s_in.as<float>(), s->d, s->dpad, (uint64_t)B, s->s_q.as<float>());
  }
  qd = s->s_q.as<float>();
  HIPCHK(ctx, hipMemcpyAsync(s->s_cand.p, cand, (size_t)B * C * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(score_candidates_kernel, dim3(cdiv((uint64_t)B * C, 256)), dim3(256), 0, ctx->stream, s->data,
                     s->dpad, qd, s->s_cand.as<uint32_t>(), B, C, C, s->s_out.as<float>());
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, s->s_out.p, (size_t)B * C * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

int fvdb_scorer_create(fvdb_store* s, uint32_t max_B, uint32_t max_C, fvdb_scorer** out) {
  fvdb_ctx* ctx = s->ctx;
  if (!out || max_B == 0 || max_C == 0) FAIL(ctx, FVDB_E_INVALID, "bad scorer shape");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  fvdb_scorer* sc = new (std::nothrow) fvdb_scorer();
  if (!sc) return FVDB_E_OOM;
  sc->store = s;
  sc->max_B = max_B;
  sc->max_C = max_C;
  const size_t nc = (size_t)max_B * max_C;
  if (hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&sc->d_q, (size_t)max_B * s->dpad * 4) != hipSuccess ||
      hipHostMalloc((void**)&sc->h_cand, nc * 4, hipHostMallocMapped) != hipSuccess ||
      hipHostMalloc((void**)&sc->h_dist, nc * 4, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&sc->d_cand, sc->h_cand, 0) != hipSuccess ||
      hipHostGetDevicePointer((void**)&sc->d_dist, sc->h_dist, 0) != hipSuccess) {
    fvdb_scorer_destroy(sc);
    FAIL(ctx, FVDB_E_OOM, "scorer allocation failed");
  }
  std::memset(sc->h_cand, 0xFF, nc * 4);
  *out = sc;
  return FVDB_OK;
}

void fvdb_scorer_destroy(fvdb_scorer* sc) {
  if (!sc) return;
  (void)hipSetDevice(sc->store->ctx->device);
  if (sc->stream) (void)hipStreamSynchronize(sc->stream);
  if (sc->d_q) (void)hipFree(sc->d_q);
  if (sc->h_cand) (void)hipHostFree(sc->h_cand);
  if (sc->h_dist) (void)hipHostFree(sc->h_dist);
  sc->s_rows.release();
  sc->s_in.release();
  if (sc->stream) (void)hipStreamDestroy(sc->stream);
  delete sc;
}

int fvdb_scorer_set_queries(fvdb_scorer* sc, const float* q, uint32_t B) {
  fvdb_store* s = sc->store;
  fvdb_ctx* ctx = s->ctx;
  if (B > sc->max_B) FAIL(ctx, FVDB_E_INVALID, "B above scorer capacity");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = check_finite(ctx, q, (uint64_t)B * s->d);
  if (rc) return rc;
  if (s->d == s->dpad) {
    HIPCHK(ctx, hipMemcpyAsync(sc->d_q, q, (size_t)B * s->d * 4, hipMemcpyHostToDevice, sc->stream));
  } else {
    HIPCHK(ctx, sc->s_in.ensure((size_t)B * s->d * 4));
    HIPCHK(ctx, hipMemcpyAsync(sc->s_in.p, q, (size_t)B * s->d * 4, hipMemcpyHostToDevice, sc->stream));
    hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv((uint64_t)B * s->dpad, 256)), dim3(256), 0, sc->stream,
                       sc->s_in.as<float>(), s->d, s->dpad, (uint64_t)B, sc->d_q);
  }
  HIPCHK(ctx, hipStreamSynchronize(sc->stream));
  return FVDB_OK;
}

int fvdb_scorer_set_query_rows(fvdb_scorer* sc, const uint32_t* rows, uint32_t B) {
  fvdb_store* s = sc->store;
  fvdb_ctx* ctx = s->ctx;
  if (B > sc->max_B) FAIL(ctx, FVDB_E_INVALID, "B above scorer capacity");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  for (uint32_t i = 0; i < B; ++i)
    if (rows[i] >= s->rows) FAIL(ctx, FVDB_E_NOT_FOUND, "query row out of range");
  HIPCHK(ctx, sc->s_rows.ensure((size_t)B * 4));
  HIPCHK(ctx, hipMemcpyAsync(sc->s_rows.p, rows, (size_t)B * 4, hipMemcpyHostToDevice, sc->stream));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv((uint64_t)B * s->dpad, 256)), dim3(256), 0, sc->stream, s->data,
                     sc->s_rows.as<uint32_t>(), s->dpad, B, sc->d_q);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(sc->stream));
  return FVDB_OK;
}

int fvdb_scorer_set_queries_dev(fvdb_scorer* sc, const float* q_dev, uint32_t B) {
  fvdb_store* s = sc->store;
  fvdb_ctx* ctx = s->ctx;
  if (B > sc->max_B) FAIL(ctx, FVDB_E_INVALID, "B above scorer capacity");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (s->d == s->dpad) {
    HIPCHK(ctx, hipMemcpyAsync(sc->d_q, q_dev, (size_t)B * s->d * 4, hipMemcpyDeviceToDevice, sc->stream));
  } else {
    hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv((uint64_t)B * s->dpad, 256)), dim3(256), 0, sc->stream, q_dev, s->d,
                       s->dpad, (uint64_t)B, sc->d_q);
    HIPCHK(ctx, hipGetLastError());
  }
  return FVDB_OK;  // stream-ordered before the next fvdb_scorer_run on this context
}

uint32_t* fvdb_scorer_cand_buffer(fvdb_scorer* sc) { return sc->h_cand; }
const float* fvdb_scorer_dist_buffer(fvdb_scorer* sc) { return sc->h_dist; }

int fvdb_scorer_launch(fvdb_scorer* sc, uint32_t B, uint32_t C) {
  fvdb_store* s = sc->store;
  fvdb_ctx* ctx = s->ctx;
  if (B > sc->max_B || C > sc->max_C) FAIL(ctx, FVDB_E_INVALID, "shape above scorer capacity");
  if (B == 0 || C == 0) return FVDB_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return FVDB_E_HIP;  // current device is per host thread
  hipLaunchKernelGGL(score_candidates_kernel, dim3(cdiv((uint64_t)B * C, 256)), dim3(256), 0, sc->stream, s->data,
                     s->dpad, sc->d_q, sc->d_cand, B, C, sc->max_C, sc->d_dist);
  if (hipGetLastError() != hipSuccess) return FVDB_E_HIP;
  return FVDB_OK;
}

int fvdb_scorer_wait(fvdb_scorer* sc) {
  if (hipStreamSynchronize(sc->stream) != hipSuccess) return FVDB_E_HIP;
  return FVDB_OK;
}

int fvdb_scorer_run(fvdb_scorer* sc, uint32_t B, uint32_t C) {
  int rc = fvdb_scorer_launch(sc, B, C);
  if (rc) return rc;
  return fvdb_scorer_wait(sc);
}


}  // extern "C"

#include "comm_sharded.h"
