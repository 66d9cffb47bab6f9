// hybrid_index.cpp — HybridIndex mirror (src/hybrid/core.rs): age routing, per-search
// auto-migration, HNSW + IVF search and the stable merge.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fvdb_host.hpp"

namespace fvdbh {

HybridIndex::HybridIndex(fvdb_ctx* ctx_ivf, fvdb_ctx* ctx_hnsw, const HybridConfig& cfg) : ctx_ivf_(ctx_ivf), cfg_(cfg) {
  recent_ = new HNSWIndex(ctx_hnsw, cfg.hnsw);
  historical_ = new IVFIndex(ctx_ivf, cfg.ivf);
}

HybridIndex::~HybridIndex() {
  for (Slot& sl : slots_) {
    if (sl.d_hid) fvdb_dev_free(ctx_ivf_, sl.d_hid);  // the other pointers are carved out of these two blocks
    if (sl.h_hid) fvdb_host_free(ctx_ivf_, sl.h_hid);
    if (sl.ivf_done) fvdb_event_destroy(sl.ivf_done);
    if (sl.d_q) fvdb_dev_free(ctx_ivf_, sl.d_q);
    if (sl.ivf_ctx && sl.ivf_ctx != ctx_ivf_) fvdb_ctx_destroy(sl.ivf_ctx);
  }
  if (sharded_) fvdb_sharded_destroy(sharded_);
  delete recent_;
  delete historical_;
}

// src/hybrid/core.rs:262-290
int HybridIndex::initialize(const float* data, uint64_t n, uint32_t dim) {
  if (n < cfg_.min_ivf_training_size) {  // HNSW-only mode
    ivf_trained_ = false;
    initialized_ = true;
    return FVDB_OK;
  }
  int rc = historical_->train(data, n, dim, nullptr);
  if (rc) return rc;
  historical_->clear_lists();  // :278-287
  ivf_trained_ = true;
  initialized_ = true;
  return FVDB_OK;
}

int HybridIndex::set_ivf_centroids(const float* c, uint32_t dim) {
  int rc = historical_->set_trained(c, dim);
  if (rc) return rc;
  ivf_trained_ = true;
  initialized_ = true;
  return FVDB_OK;
}

// The write guard.  Waiting happens WITHOUT holding rw_ (a searcher that is about to collect its batches may need the
// read side first), and a begin takes the read side while it marks its slot active, so no batch can start under a
// mutation.
int HybridIndex::write_lock(std::unique_lock<std::shared_mutex>& w) {
  for (;;) {
    w.lock();  // the blocking searches of other threads hold the read side for their whole duration: waited for here
    if (!busy()) return FVDB_OK;
    w.unlock();  // what is left in flight was begun with search_dev_begin and not collected yet
    if (!writers_wait_) return FVDB_E_INVALID;
    std::unique_lock<std::mutex> lk(slot_mu_);
    slot_cv_.wait(lk, [&] { return !busy_unlocked(); });
  }
}

// src/hybrid/core.rs:357-417
int HybridIndex::insert_with_timestamp(uint64_t id, const float* v, uint32_t dim, double ts, double now,
                                       int64_t level) {
  if (!initialized_) return FVDB_E_NOT_INITIALIZED;
  std::unique_lock<std::shared_mutex> w(rw_, std::defer_lock);  // waits for the blocking searches of other threads (write guard)
  if (int rc = write_lock(w)) return rc;
  if (timestamps_.count(id)) return FVDB_E_DUPLICATE;
  bool to_recent = !ivf_trained_ || age_of(now, ts) < cfg_.recent_threshold_s;
  if (to_recent) {
    int rc = recent_->insert(id, v, dim, level);
    if (rc) return rc;
    recent_count_ += 1;
    pending_migration_.push_back({id, ts});
    pending_min_ts_ = std::min(pending_min_ts_, ts);
  } else {
    int rc = historical_->insert(id, v, dim);
    if (rc) return rc;
    historical_count_ += 1;
  }
  timestamps_[id] = ts;
  ts_order_.push_back(id);
  return FVDB_OK;
}

// the recent part of a bulk load: the reference's sequential inserts (levels drawn in order), or the exact nearest-M graph
int HybridIndex::build_recent(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc, err = 0;
  if (!sequential_graph_) {
    rc = recent_->bulk_build(ids, v, n, dim, nullptr);
  } else {
    uint64_t ok = 0;
    rc = recent_->batch_insert(ids, v, n, dim, nullptr, &ok, &err);
  }
  recent_build_s_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc ? rc : err;
}

// Scale loader: same routing as insert_with_timestamp, but the HNSW part is built in one call (build_recent) and the
// IVF part is assigned/appended in one GPU pass.  Only valid on an index with no vectors yet.
int HybridIndex::bulk_insert(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const double* ts,
                             double now) {
  if (!initialized_) return FVDB_E_NOT_INITIALIZED;
  std::unique_lock<std::shared_mutex> w(rw_);
  if (!ts_order_.empty() || busy()) return FVDB_E_INVALID;
  std::vector<uint64_t> rid, hid;
  std::vector<float> rv, hv;
  for (uint64_t i = 0; i < n; ++i) {
    if (timestamps_.count(ids[i])) return FVDB_E_DUPLICATE;
    timestamps_[ids[i]] = ts[i];
    const bool to_recent = !ivf_trained_ || age_of(now, ts[i]) < cfg_.recent_threshold_s;
    auto& I = to_recent ? rid : hid;
    auto& V = to_recent ? rv : hv;
    I.push_back(ids[i]);
    V.insert(V.end(), v + i * dim, v + (i + 1) * dim);
  }
  ts_order_.assign(ids, ids + n);
  if (!rid.empty()) {
    int rc = build_recent(rid.data(), rv.data(), rid.size(), dim);
    if (rc) return rc;
    recent_count_ = rid.size();
    for (uint64_t i = 0; i < n; ++i)
      if (!ivf_trained_ || age_of(now, ts[i]) < cfg_.recent_threshold_s) {
        pending_migration_.push_back({ids[i], ts[i]});
        pending_min_ts_ = std::min(pending_min_ts_, ts[i]);
      }
  }
  if (!hid.empty()) {
    uint64_t ok = 0;
    int err = 0;
    int rc = historical_->batch_insert(hid.data(), hv.data(), hid.size(), dim, &ok, &err);
    if (rc) return rc;
    if (err) return err;
    historical_count_ = ok;
  }
  return FVDB_OK;
}

// Greedy "largest list to the least loaded rank" placement; ties -> lower rank.
static void plan_list_owners(const std::vector<uint64_t>& sizes, uint32_t world, std::vector<uint32_t>& owner) {
  const size_t nlist = sizes.size();
  std::vector<uint32_t> order(nlist);
  for (size_t i = 0; i < nlist; ++i) order[i] = (uint32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sizes[a] > sizes[b]; });
  std::vector<uint64_t> load(world, 0);
  owner.assign(nlist, 0);
  for (uint32_t L : order) {
    uint32_t best = 0;
    for (uint32_t r = 1; r < world; ++r)
      if (load[r] < load[best]) best = r;
    owner[L] = best;
    load[best] += sizes[L];
  }
}

int HybridIndex::bulk_insert_sharded(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const double* ts,
                                     double now, uint32_t rank, uint32_t world, uint32_t* owner_out) {
  if (!initialized_) return FVDB_E_NOT_INITIALIZED;
  std::unique_lock<std::shared_mutex> w(rw_);
  if (!ts_order_.empty() || world == 0 || rank >= world || busy()) return FVDB_E_INVALID;
  std::vector<uint64_t> rid, hid;
  std::vector<float> rv, hv;
  for (uint64_t i = 0; i < n; ++i) {
    if (timestamps_.count(ids[i])) return FVDB_E_DUPLICATE;
    timestamps_[ids[i]] = ts[i];
    const bool to_recent = !ivf_trained_ || age_of(now, ts[i]) < cfg_.recent_threshold_s;
    auto& I = to_recent ? rid : hid;
    auto& V = to_recent ? rv : hv;
    I.push_back(ids[i]);
    V.insert(V.end(), v + i * dim, v + (i + 1) * dim);
  }
  ts_order_.assign(ids, ids + n);
  if (!rid.empty()) {  // replicated graph: every rank builds the same one
    int rc = build_recent(rid.data(), rv.data(), rid.size(), dim);
    if (rc) return rc;
    recent_count_ = rid.size();
    for (uint64_t i = 0; i < n; ++i)
      if (!ivf_trained_ || age_of(now, ts[i]) < cfg_.recent_threshold_s) {
        pending_migration_.push_back({ids[i], ts[i]});
        pending_min_ts_ = std::min(pending_min_ts_, ts[i]);
      }
  }
  shard_rank_ = rank;
  shard_world_ = world;
  if (ivf_trained_) {
    const uint32_t nlist = historical_->config().n_clusters;
    shard_sizes_.assign(nlist, 0);
    plan_list_owners(shard_sizes_, world, shard_owner_);  // (replaced below when rows go to the lists now)
  }
  if (!hid.empty()) {
    const uint32_t nlist = historical_->config().n_clusters;
    std::vector<uint32_t> cl(hid.size());
    int rc = historical_->assign(hv.data(), hid.size(), dim, cl.data());
    if (rc) return rc;
    std::vector<uint64_t> sizes(nlist, 0);
    for (uint32_t c : cl) sizes[c]++;
    std::vector<uint32_t> owner;
    plan_list_owners(sizes, world, owner);
    shard_owner_ = owner;
    shard_sizes_ = sizes;
    if (owner_out) std::memcpy(owner_out, owner.data(), nlist * sizeof(uint32_t));
    std::vector<uint64_t> kid;
    std::vector<float> kv;
    std::vector<uint32_t> kc;
    for (size_t i = 0; i < hid.size(); ++i)
      if (owner[cl[i]] == rank) {
        kid.push_back(hid[i]);
        kc.push_back(cl[i]);
        kv.insert(kv.end(), hv.begin() + i * dim, hv.begin() + (i + 1) * dim);
      }
    uint64_t ok = 0;
    int err = 0;
    rc = historical_->batch_insert_assigned(kid.data(), kv.data(), kid.size(), dim, kc.data(), &ok, &err);
    if (rc) return rc;
    if (err) return err;
    rc = fvdb_ivf_set_global_list_sizes(historical_->device(), sizes.data());
    if (rc) return rc;
    historical_count_ = hid.size();  // logical count; this rank stores `ok` of them
  }
  return FVDB_OK;
}

void plan_list_owners_host(const uint64_t* sizes, uint32_t nlist, uint32_t world, uint32_t* owner) {
  std::vector<uint64_t> sz(sizes, sizes + nlist);
  std::vector<uint32_t> ow;
  plan_list_owners(sz, world ? world : 1, ow);
  std::memcpy(owner, ow.data(), (size_t)nlist * sizeof(uint32_t));
}

// src/hybrid/core.rs:857-877
int HybridIndex::from_parts(const uint64_t* ids, const double* ts, uint64_t n, uint64_t recent_count,
                            uint64_t historical_count, bool ivf_trained) {
  std::unique_lock<std::shared_mutex> w(rw_);
  if (!ts_order_.empty() || busy()) return FVDB_E_INVALID;
  if (ivf_trained && !historical_->is_trained()) return FVDB_E_NOT_TRAINED;
  for (uint64_t i = 0; i < n; ++i) {
    if (!timestamps_.emplace(ids[i], ts[i]).second) continue;  // a map on disk holds each key once; keep the first
    ts_order_.push_back(ids[i]);
    // ids with a node in the graph are what migrate_with_threshold can still copy into IVF (:626 get_node); one
    // already in a list fails there as a duplicate and leaves the queue, as on every search of the reference
    if (recent_->vector_of(ids[i])) {
      pending_migration_.push_back({ids[i], ts[i]});
      pending_min_ts_ = std::min(pending_min_ts_, ts[i]);
    }
  }
  recent_count_ = recent_count;
  historical_count_ = historical_count;
  ivf_trained_ = ivf_trained;
  initialized_ = true;
  return FVDB_OK;
}

// src/hybrid/core.rs:989-1012
int HybridIndex::vacuum(uint64_t* hnsw_removed, uint64_t* ivf_removed) {
  std::unique_lock<std::shared_mutex> w(rw_, std::defer_lock);
  if (int rc = write_lock(w)) return rc;
  *hnsw_removed = recent_->vacuum();
  return historical_->vacuum(ivf_removed);
}

void HybridIndex::export_timestamps(uint64_t* ids, double* ts) const {
  for (size_t i = 0; i < ts_order_.size(); ++i) {
    ids[i] = ts_order_[i];
    ts[i] = timestamps_.at(ts_order_[i]);
  }
}

// src/hybrid/core.rs:600-649 — copies into IVF, never removes from HNSW (:577-581)
uint64_t HybridIndex::migrate_with_threshold(double threshold_s, double now) {
  std::unique_lock<std::shared_mutex> w(rw_, std::defer_lock);
  if (write_lock(w)) return 0;
  return migrate_locked(threshold_s, now);
}

// the caller holds rw_ exclusively and no batch is in flight
uint64_t HybridIndex::migrate_locked(double threshold_s, double now) {
  // The reference walks the whole timestamps map on every search (:606-617).  Same outcome, O(1) when
  // nothing is due: only ids still living in HNSW alone can migrate, and none is due while the oldest
  // of them is younger than the threshold.
  if (!migration_due(threshold_s, now)) return 0;
  std::vector<Pending> keep;
  std::vector<uint64_t> due;
  double min_ts = 1e300;
  for (const Pending& p : pending_migration_) {
    if (age_of(now, p.ts) >= threshold_s) {
      due.push_back(p.id);
    } else {
      keep.push_back(p);
      min_ts = std::min(min_ts, p.ts);
    }
  }
  if (due.empty()) return 0;
  uint64_t migrated = 0;
  const uint32_t dim = recent_->dimension();
  std::vector<float> xv;
  std::vector<uint64_t> xi;
  for (uint64_t id : due) {
    const float* vec = recent_->vector_of(id);
    if (!vec) continue;  // get_node(id) == None
    xi.push_back(id);
    xv.insert(xv.end(), vec, vec + dim);
  }
  if (!xi.empty() && shard_world_ > 0 && !shard_owner_.empty()) {
    // lists live on their owner ranks (bulk_insert_sharded): every rank runs this with the same due ids in the same
    // order and the same centroids — the owner of a row's list appends it, every rank counts it into the logical list
    // sizes, so list order and the selection keys' positions are what the unsharded index would have.  (An id that
    // is already in a list of ANOTHER rank cannot be seen here; ids that are still pending were never copied.)
    std::vector<uint32_t> cl(xi.size());
    if (historical_->assign(xv.data(), xi.size(), dim, cl.data())) return 0;
    std::vector<uint64_t> kid;
    std::vector<float> kv;
    std::vector<uint32_t> kc;
    for (size_t i = 0; i < xi.size(); ++i) {
      shard_sizes_[cl[i]] += 1;
      if (shard_owner_[cl[i]] == shard_rank_) {
        kid.push_back(xi[i]);
        kc.push_back(cl[i]);
        kv.insert(kv.end(), xv.begin() + i * dim, xv.begin() + (i + 1) * dim);
      }
    }
    uint64_t ok = 0;
    int err = 0;
    if (!kid.empty() && historical_->batch_insert_assigned(kid.data(), kv.data(), kid.size(), dim, kc.data(), &ok, &err)) return 0;
    if (fvdb_ivf_set_global_list_sizes(historical_->device(), shard_sizes_.data())) return 0;
    migrated = xi.size();
  } else if (!xi.empty()) {
    int err = 0;
    int rc = historical_->batch_insert(xi.data(), xv.data(), xi.size(), dim, &migrated, &err);
    if (rc) return 0;  // IVF untrained / dimension mismatch: every insert fails; ids stay pending
  }
  // A copy that failed as a duplicate fails the same way on every later search (the reference
  // retries it each time with no effect), so due ids leave the queue either way.
  pending_migration_.swap(keep);
  pending_min_ts_ = min_ts;
  if (migrated) {
    recent_count_ = recent_count_ >= migrated ? recent_count_ - migrated : 0;
    historical_count_ += migrated;
  }
  return migrated;
}

// src/hybrid/core.rs:425-486 for a batch of queries
int HybridIndex::search(const float* q, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg, double now,
                        uint64_t* ids, float* dist, uint32_t* counts) {
  return search_impl(q, false, B, dim, cfg, now, ids, dist, counts);
}
int HybridIndex::search_dev(const float* q_dev, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg, double now,
                            uint64_t* ids, float* dist, uint32_t* counts) {
  return search_impl(q_dev, true, B, dim, cfg, now, ids, dist, counts);
}

// stable merge of the two parts (:476-485): concatenate recent then historical, stable sort by distance, take k
static void merge_parts(uint32_t B, uint32_t k, uint32_t rk, uint32_t hk, bool have_r, const uint64_t* rid, const float* rd,
                        const uint32_t* rc, bool have_h, const uint64_t* hid, const float* hd, const uint32_t* hc,
                        uint64_t* ids, float* dist, uint32_t* counts) {
  struct R {
    uint64_t id;
    float d;
  };
  std::vector<R> all;
  for (uint32_t b = 0; b < B; ++b) {
    all.clear();
    if (have_r)
      for (uint32_t i = 0; i < rc[b]; ++i) all.push_back({rid[(size_t)b * rk + i], rd[(size_t)b * rk + i]});
    if (have_h)
      for (uint32_t i = 0; i < hc[b]; ++i) all.push_back({hid[(size_t)b * hk + i], hd[(size_t)b * hk + i]});
    std::stable_sort(all.begin(), all.end(), [](const R& a, const R& c) { return a.d < c.d; });  // :482
    if (all.size() > k) all.resize(k);
    for (size_t i = 0; i < all.size(); ++i) {
      ids[(size_t)b * k + i] = all[i].id;
      dist[(size_t)b * k + i] = all[i].d;
    }
    counts[b] = (uint32_t)all.size();
  }
}

void merge_parts_host(uint32_t B, uint32_t k, uint32_t rk, uint32_t hk, const uint64_t* rid, const float* rd,
                      const uint32_t* rc, const uint64_t* hid, const float* hd, const uint32_t* hc, uint64_t* ids,
                      float* dist, uint32_t* counts) {
  for (size_t i = 0; i < (size_t)B * k; ++i) {
    ids[i] = FVDB_NO_ID;
    dist[i] = __builtin_huge_valf();
  }
  merge_parts(B, k, rk, hk, rid != nullptr, rid, rd, rc, hid != nullptr, hid, hd, hc, ids, dist, counts);
}

// Explicit pair for ONE thread that keeps several batches in flight (bench, pipelined servers).
int HybridIndex::search_dev_begin(uint32_t slot, const float* q_dev, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg,
                                  double now) {
  if (slot >= kSlots) return FVDB_E_INVALID;
  Slot& sl = slots_[slot];
  bool others = false;
  {
    std::lock_guard<std::mutex> lk(slot_mu_);
    if (sl.active) return FVDB_E_INVALID;  // the previous batch of this slot was never collected
    others = busy_unlocked();
  }
  // per-search auto-migration (src/hybrid/core.rs:437-439), before this batch counts as in flight.  A due migration
  // moves rows between the two indexes (and may grow the list pool) under the batches still in flight: refuse, the
  // caller collects them and begins again
  if (initialized_ && B != 0 && cfg.k != 0 && cfg_.auto_migrate) {
    std::unique_lock<std::shared_mutex> w(rw_);
    if (migration_due(cfg_.recent_threshold_s, now)) {
      if (others || busy()) return FVDB_E_INVALID;
      migrate_locked(cfg_.recent_threshold_s, now);
    }
  }
  std::shared_lock<std::shared_mutex> r(rw_);  // no mutation is under way while the batch is enqueued
  {
    std::lock_guard<std::mutex> lk(slot_mu_);
    if (sl.active) return FVDB_E_INVALID;
    sl.active = true;
  }
  const int rc = begin_impl(slot, q_dev, B, dim, cfg);
  return rc;
}

int HybridIndex::attach_comm(fvdb_comm* comm) {
  std::unique_lock<std::shared_mutex> w(rw_);
  if (busy()) return FVDB_E_INVALID;
  if (sharded_) fvdb_sharded_destroy(sharded_);
  sharded_ = nullptr;
  comm_ = comm;
  if (!comm) return FVDB_OK;
  if (!ivf_trained_ || !historical_->device()) return FVDB_E_NOT_TRAINED;
  return fvdb_sharded_create(historical_->device(), comm, &sharded_);
}

uint32_t HybridIndex::sharded_rows(uint32_t B, int mode) const {
  if (!comm_ || mode != FVDB_SHARD_STRONG) return B;
  const uint32_t W = (uint32_t)fvdb_comm_world(comm_), r = (uint32_t)fvdb_comm_rank(comm_);
  const uint32_t per = (B + W - 1) / W, lo = std::min(B, r * per), hi = std::min(B, (r + 1) * per);
  return hi - lo;
}

int HybridIndex::search_sharded_begin(uint32_t slot, const float* q_dev, uint32_t B, uint32_t dim,
                                      const HybridSearchConfig& cfg, int mode, double now) {
  if (slot >= kSlots || !sharded_ || (mode != FVDB_SHARD_WEAK && mode != FVDB_SHARD_STRONG)) return FVDB_E_INVALID;
  // per-search auto-migration as in search_dev_begin; `now` is the same on every rank, so every rank migrates the same
  // rows at the same step (a due migration while batches are in flight is refused: collect first)
  if (initialized_ && cfg.k != 0 && cfg_.auto_migrate) {
    bool others;
    {
      std::lock_guard<std::mutex> lk(slot_mu_);
      if (slots_[slot].active) return FVDB_E_INVALID;
      others = busy_unlocked();
    }
    std::unique_lock<std::shared_mutex> w(rw_);
    if (migration_due(cfg_.recent_threshold_s, now)) {
      if (others || busy()) return FVDB_E_INVALID;
      migrate_locked(cfg_.recent_threshold_s, now);
    }
  }
  std::shared_lock<std::shared_mutex> r(rw_);
  {
    std::lock_guard<std::mutex> lk(slot_mu_);
    if (slots_[slot].active) return FVDB_E_INVALID;
    slots_[slot].active = true;
  }
  return begin_impl(slot, q_dev, B, dim, cfg, mode);
}

// enqueue everything for the batch; the slot is already marked active by the caller
int HybridIndex::begin_impl(uint32_t slot, const float* q_dev, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg,
                            int shard_mode) {
  Slot& sl = slots_[slot];
  sl.ivf_in_flight = sl.hnsw_in_flight = false;
  // sharded: `q_own` / `Bown` are the queries whose final results this rank produces (the graph is replicated, so each
  // rank walks it for those only); the IVF part is handed the batch as the mode defines it
  const float* q_own = q_dev;
  uint32_t Bown = B, ivf_rows = B;
  if (shard_mode == FVDB_SHARD_STRONG) {
    const uint32_t W = (uint32_t)fvdb_comm_world(comm_), r = (uint32_t)fvdb_comm_rank(comm_);
    const uint32_t per = (B + W - 1) / W, lo = std::min(B, r * per), hi = std::min(B, (r + 1) * per);
    q_own = q_dev + (size_t)lo * dim;
    Bown = hi - lo;
    ivf_rows = per;
  }
  const uint32_t B_ivf_in = B;
  B = Bown;
  sl.q = q_own;
  sl.B = B;
  sl.dim = dim;
  sl.k = (uint32_t)cfg.k;
  sl.rk = (uint32_t)(cfg.recent_k > 0 ? cfg.recent_k : cfg.k);
  sl.hk = (uint32_t)(cfg.historical_k > 0 ? cfg.historical_k : cfg.k);
  sl.ef = (uint32_t)cfg.hnsw_ef;
  sl.recent = cfg.search_recent;
  if (!initialized_ || sl.k == 0 || (B == 0 && shard_mode < 0)) return FVDB_OK;
  if (cfg.search_recent && B > 0) {
    // the graph walk is latency-bound (one wave per query): enqueue it FIRST so that the list scan launched
    // next fills the rest of every SIMD and the two run concurrently
    int rcb = 0;
    sl.hnsw_in_flight = recent_->search_dev_begin(q_own, B, dim, sl.rk, sl.ef, &rcb, slot);
  }
  if (cfg.search_historical && ivf_trained_) {
    // one device block and one pinned block per slot: [ids R*hk u64 | dist R*hk f32 | counts R u32] -> a single copy
    // (R = rows the IVF part writes: B, or the padded slice length in strong sharded mode)
    const uint64_t need = (uint64_t)ivf_rows * sl.hk;
    const uint64_t bytes = need * 12 + (uint64_t)ivf_rows * 4;
    if (bytes > sl.cap) {
      if (sl.d_hid) fvdb_dev_free(ctx_ivf_, sl.d_hid);
      if (sl.h_hid) fvdb_host_free(ctx_ivf_, sl.h_hid);
      sl.d_hid = sl.h_hid = nullptr;
      sl.cap = 0;
      if (fvdb_dev_alloc(ctx_ivf_, bytes, &sl.d_hid) || fvdb_host_alloc(ctx_ivf_, bytes, &sl.h_hid)) return FVDB_E_OOM;
      sl.cap = bytes;
    }
    sl.d_hd = (char*)sl.d_hid + need * 8;
    sl.d_hc = (char*)sl.d_hid + need * 12;
    sl.h_hd = (char*)sl.h_hid + need * 8;
    sl.h_hc = (char*)sl.h_hid + need * 12;
    // each slot's IVF chain runs on its own stream with its own scratch set, so the chains of consecutive batches
    // overlap (a chain is ~20 dependent launches with gaps between them)
    static const bool one_stream = getenv("FVDB_IVF_ONE_STREAM") != nullptr;  // tuning aid
    if (!sl.ivf_ctx) {
      if (slot == 0 || one_stream) sl.ivf_ctx = ctx_ivf_;
      else if (fvdb_ctx_create(fvdb_ctx_device(ctx_ivf_), &sl.ivf_ctx)) return FVDB_E_HIP;
    }
    fvdb_ctx* on = sl.ivf_ctx == ctx_ivf_ ? nullptr : sl.ivf_ctx;
    if (!sl.ivf_done && fvdb_event_create(sl.ivf_ctx, &sl.ivf_done)) return FVDB_E_HIP;
    if (shard_mode >= 0) {
      // every rank must take part in the step's collectives even when its own slice is empty
      if (dim != historical_->dimension()) return FVDB_E_DIM;
      const int rcs = fvdb_ivf_search_sharded_begin(sharded_, on, on ? slot : 0, q_dev, B_ivf_in, sl.hk,
                                                    (uint32_t)cfg.ivf_n_probe, shard_mode, (uint64_t*)sl.d_hid,
                                                    (float*)sl.d_hd, (uint32_t*)sl.d_hc);
      if (rcs) return rcs;
      sl.ivf_in_flight = true;
    } else {
      sl.ivf_in_flight = historical_->search_dev(q_dev, B, dim, sl.hk, (uint32_t)cfg.ivf_n_probe, (uint64_t*)sl.d_hid,
                                                 (float*)sl.d_hd, (uint32_t*)sl.d_hc, on, on ? slot : 0) == FVDB_OK;
    }
    if (sl.ivf_in_flight) {
      // result copy rides the slot's stream right behind the chain, then the event
      if (fvdb_dev_download_async(sl.ivf_ctx, sl.h_hid, sl.d_hid, (size_t)bytes) || fvdb_event_record(sl.ivf_ctx, sl.ivf_done))
        return FVDB_E_HIP;
    }
  }
  return FVDB_OK;
}

int HybridIndex::search_dev_end(uint32_t slot, uint64_t* ids, float* dist, uint32_t* counts) {
  if (slot >= kSlots) return FVDB_E_INVALID;
  Slot& sl = slots_[slot];
  {
    std::lock_guard<std::mutex> lk(slot_mu_);
    if (!sl.active) return FVDB_E_INVALID;
  }
  struct Release {  // the slot is free again only when its buffers have been read
    HybridIndex* h;
    Slot* sl;
    ~Release() {
      {
        std::lock_guard<std::mutex> lk(h->slot_mu_);
        sl->active = false;
      }
      h->slot_cv_.notify_all();
    }
  } release{this, &sl};
  const uint32_t B = sl.B, k = sl.k;
  for (uint32_t b = 0; b < B; ++b) counts[b] = 0;
  for (size_t i = 0; i < (size_t)B * k; ++i) {
    ids[i] = FVDB_NO_ID;
    dist[i] = __builtin_huge_valf();
  }
  if (!initialized_ || B == 0 || k == 0) return FVDB_OK;
  std::vector<uint64_t> rid;
  std::vector<float> rd;
  std::vector<uint32_t> rc_(B, 0);
  bool have_r = false, have_h = false;
  if (sl.recent) {
    rid.resize((size_t)B * sl.rk);
    rd.resize((size_t)B * sl.rk);
    int rc2 = sl.hnsw_in_flight
                  ? recent_->search_dev_end(sl.q, B, sl.dim, sl.rk, sl.ef, rid.data(), rd.data(), rc_.data(), slot)
                  : recent_->search_dev(sl.q, B, sl.dim, sl.rk, sl.ef, rid.data(), rd.data(), rc_.data());
    have_r = rc2 == FVDB_OK;
  }
  if (sl.ivf_in_flight) {
    have_h = fvdb_event_wait(sl.ivf_ctx, sl.ivf_done) == FVDB_OK;
    bool others = false;
    {
      std::lock_guard<std::mutex> lk(slot_mu_);
      for (const Slot& o : slots_) others = others || (o.active && &o != &sl);
    }
    if (!others) fvdb_ivf_profile_collect(historical_->device());  // stage timing only makes sense one batch at a time
  }
  merge_parts(B, k, sl.rk, sl.hk, have_r, rid.data(), rd.data(), rc_.data(), have_h, (const uint64_t*)sl.h_hid,
              (const float*)sl.h_hd, (const uint32_t*)sl.h_hc, ids, dist, counts);
  return FVDB_OK;
}

// The blocking entry points: any number of host threads.  A call holds the read side of rw_ from its migration
// check to its merge and works in a slot leased for its duration.
int HybridIndex::search_impl(const float* q, bool q_on_device, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg,
                             double now, uint64_t* ids, float* dist, uint32_t* counts) {
  const uint32_t k = (uint32_t)cfg.k;
  for (uint32_t b = 0; b < B; ++b) counts[b] = 0;
  for (size_t i = 0; i < (size_t)B * k; ++i) {
    ids[i] = FVDB_NO_ID;
    dist[i] = __builtin_huge_valf();
  }
  if (!initialized_ || B == 0 || k == 0) return FVDB_OK;
  if (cfg_.auto_migrate) {  // src/hybrid/core.rs:437-439; the reference takes its write locks here too (:621-622)
    bool due;
    {
      std::shared_lock<std::shared_mutex> r(rw_);
      due = migration_due(cfg_.recent_threshold_s, now);
    }
    if (due) {
      std::unique_lock<std::shared_mutex> w(rw_);
      if (migration_due(cfg_.recent_threshold_s, now)) {
        if (busy()) return FVDB_E_INVALID;  // batches begun with search_dev_begin are still uncollected
        migrate_locked(cfg_.recent_threshold_s, now);
      }
    }
  }
  std::shared_lock<std::shared_mutex> r(rw_);
  uint32_t slot = 0;
  {
    std::unique_lock<std::mutex> lk(slot_mu_);
    slot_cv_.wait(lk, [&] {
      for (const Slot& s : slots_)
        if (!s.active) return true;
      return false;
    });
    for (uint32_t i = kSlots; i-- > 0;)  // from the top: the low slots are the ones a pipelining caller names
      if (!slots_[i].active) {
        slot = i;
        break;
      }
    slots_[slot].active = true;
  }
  Slot& sl = slots_[slot];
  auto give_back = [&]() {
    {
      std::lock_guard<std::mutex> lk(slot_mu_);
      sl.active = false;
    }
    slot_cv_.notify_all();
  };
  const float* qd = q;
  if (!q_on_device) {  // stage the batch in HBM once; both parts read it from there
    const uint64_t bytes = (uint64_t)B * dim * 4;
    if (bytes > sl.d_q_cap) {
      if (sl.d_q) fvdb_dev_free(ctx_ivf_, sl.d_q);
      sl.d_q = nullptr;
      sl.d_q_cap = 0;
      if (fvdb_dev_alloc(ctx_ivf_, bytes, &sl.d_q)) {
        give_back();
        return FVDB_E_OOM;
      }
      sl.d_q_cap = bytes;
    }
    if (!sl.ivf_ctx) {
      if (slot == 0) sl.ivf_ctx = ctx_ivf_;
      else if (fvdb_ctx_create(fvdb_ctx_device(ctx_ivf_), &sl.ivf_ctx)) {
        give_back();
        return FVDB_E_HIP;
      }
    }
    for (uint64_t i = 0; i < (uint64_t)B * dim; ++i)
      if (!(q[i] - q[i] == 0.0f)) {  // NaN / Inf: the reference panics in partial_cmp().unwrap()
        give_back();
        return FVDB_E_NONFINITE;
      }
    const int rcu = fvdb_dev_upload(sl.ivf_ctx, sl.d_q, q, bytes);  // waits on the slot's own stream only
    if (rcu) {
      give_back();
      return rcu;
    }
    qd = (const float*)sl.d_q;
  }
  const int rc0 = begin_impl(slot, qd, B, dim, cfg);
  if (rc0) {
    if (sl.hnsw_in_flight || sl.ivf_in_flight) {  // drain what was enqueued before the failure
      std::vector<uint64_t> ti((size_t)B * k);
      std::vector<float> td((size_t)B * k);
      std::vector<uint32_t> tc(B);
      (void)search_dev_end(slot, ti.data(), td.data(), tc.data());
    } else {
      give_back();
    }
    return rc0;
  }
  return search_dev_end(slot, ids, dist, counts);
}

// src/hybrid/core.rs:513-549
int HybridIndex::search_with_filter(const float* q, uint32_t B, uint32_t dim, uint64_t k, FilterFn matches, void* user,
                                    double now, uint64_t* ids, float* dist, uint32_t* counts) {
  HybridSearchConfig cfg;  // SearchConfig::default() with k (:419-423)
  cfg.k = k;
  if (!matches) return search_impl(q, false, B, dim, cfg, now, ids, dist, counts);
  const uint64_t k3 = k * 3;  // k-oversampling, multiplier 3 (:527-529)
  cfg.k = k3;
  std::vector<uint64_t> ci((size_t)B * k3);
  std::vector<float> cd((size_t)B * k3);
  std::vector<uint32_t> cc(B, 0);
  const int rc = search_impl(q, false, B, dim, cfg, now, ci.data(), cd.data(), cc.data());
  if (rc) return rc;
  for (uint32_t b = 0; b < B; ++b) {
    uint32_t w = 0;
    for (uint32_t i = 0; i < cc[b] && w < k; ++i) {  // candidates are sorted by distance already; truncate(k) (:543-546)
      const uint64_t id = ci[(size_t)b * k3 + i];
      if (!matches(id, user)) continue;
      ids[(size_t)b * k + w] = id;
      dist[(size_t)b * k + w] = cd[(size_t)b * k3 + i];
      ++w;
    }
    counts[b] = w;
    for (uint32_t i = w; i < k; ++i) {
      ids[(size_t)b * k + i] = FVDB_NO_ID;
      dist[(size_t)b * k + i] = __builtin_huge_valf();
    }
  }
  return FVDB_OK;
}

// delete: src/hybrid/core.rs:904-937
int HybridIndex::remove(uint64_t id, double now) {
  std::unique_lock<std::shared_mutex> w(rw_, std::defer_lock);
  if (int rc = write_lock(w)) return rc;
  auto it = timestamps_.find(id);
  if (it == timestamps_.end()) return FVDB_E_NOT_FOUND;
  if (age_of(now, it->second) < cfg_.recent_threshold_s) return recent_->mark_deleted(id);
  return historical_->mark_deleted(id);
}

}  // namespace fvdbh
