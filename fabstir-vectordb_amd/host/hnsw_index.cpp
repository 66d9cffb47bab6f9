// hnsw_index.cpp — HNSWIndex mirror (src/hnsw/core.rs, src/hnsw/operations.rs).
// Ids, levels, flags and a copy of the adjacency live here.  Inserts and batch searches normally run whole on the
// device against the adjacency in HBM (fvdb_graph_insert_linked / fvdb_graph_search_dev*); the host algorithm below —
// heaps and visited sets here, every distance batch scored on the GPU through fvdb_scorer_* — is the other mode of
// both (same results), and what takes over for shapes the device kernels do not serve.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fvdb_host.hpp"

namespace fvdbh {

// ---- RustHeap ------------------------------------------------------------------------------
void RustHeap::sift_up(size_t start, size_t pos) {
  Cand elt = data[pos];
  while (pos > start) {
    size_t parent = (pos - 1) / 2;
    if (le(elt, data[parent])) break;
    data[pos] = data[parent];
    pos = parent;
  }
  data[pos] = elt;
}
void RustHeap::push(Cand c) {
  size_t old = data.size();
  data.push_back(c);
  sift_up(0, old);
}
Cand RustHeap::pop() {
  Cand item = data.back();
  data.pop_back();
  if (!data.empty()) {
    std::swap(item, data[0]);
    const size_t end = data.size();
    size_t pos = 0;
    Cand elt = data[0];
    size_t child = 1;
    const size_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim) {
      if (le(data[child], data[child + 1])) child += 1;
      data[pos] = data[child];
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1) {
      data[pos] = data[child];
      pos = child;
    }
    data[pos] = elt;
    sift_up(0, pos);
  }
  return item;
}

// ---- Visited -------------------------------------------------------------------------------
void Visited::reset(size_t expect) {
  size_t want = 64;
  while (want < expect * 2) want <<= 1;
  if (keys.size() < want) {
    keys.assign(want, 0);
    stamp.assign(want, 0);
    epoch = 0;
  }
  epoch += 1;
  if (epoch == 0) {  // wrapped
    std::fill(stamp.begin(), stamp.end(), 0u);
    epoch = 1;
  }
  used = 0;
}
void Visited::grow() {
  std::vector<uint32_t> ok;
  ok.reserve(used);
  for (size_t i = 0; i < keys.size(); ++i)
    if (stamp[i] == epoch) ok.push_back(keys[i]);
  const size_t nsz = keys.size() * 2;
  keys.assign(nsz, 0);
  stamp.assign(nsz, 0);
  epoch = 1;
  used = 0;
  for (uint32_t v : ok) insert(v);
}
bool Visited::insert(uint32_t v) {
  if ((used + 1) * 2 > keys.size()) grow();
  const size_t mask = keys.size() - 1;
  size_t h = (v * 2654435761u) & mask;
  for (;;) {
    if (stamp[h] != epoch) {
      stamp[h] = epoch;
      keys[h] = v;
      used++;
      return true;
    }
    if (keys[h] == v) return false;
    h = (h + 1) & mask;
  }
}

// Worker threads for the lock-step traversal: the CPUs this process may actually use — the
// smaller of its affinity mask and its cgroup CPU quota (a container can see 256 CPUs and own
// 16; oversubscribed OpenMP barriers then collapse under CFS throttling).  FVDB_HOST_THREADS overrides.
static int usable_cpus() {
  if (const char* e = getenv("FVDB_HOST_THREADS")) {
    int v = atoi(e);
    if (v > 0) return v;
  }
  int n = omp_get_num_procs();
  long quota = -1, period = -1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    char q[64];
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
    fclose(f);
  } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    if (fscanf(f1, "%ld", &quota) != 1) quota = -1;
    fclose(f1);
    if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(f2, "%ld", &period) != 1) period = -1;
      fclose(f2);
    }
  }
  if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
  return std::max(1, std::min(n, 64));
}

static inline bool set_insert(std::vector<uint32_t>& s, uint32_t v) {
  if (std::find(s.begin(), s.end(), v) != s.end()) return false;
  s.push_back(v);
  return true;
}

// ---- HNSWIndex -----------------------------------------------------------------------------
HNSWIndex::HNSWIndex(fvdb_ctx* ctx, const HNSWConfig& cfg) : ctx_(ctx), cfg_(cfg), rng_(cfg.seed) {}

HNSWIndex::~HNSWIndex() {
  if (graph_) fvdb_graph_destroy(graph_);
  for (uint32_t i = 0; i < kSlots; ++i) {
    DevSlot& sl = slots_[i];
    fvdb_ctx* c = sl.ctx ? sl.ctx : ctx_;
    if (sl.d_nodes) fvdb_dev_free(c, sl.d_nodes);  // the other pointers are carved out of these two blocks
    if (sl.h_nodes) fvdb_host_free(c, sl.h_nodes);
    if (i > 0 && sl.ctx) fvdb_ctx_destroy(sl.ctx);
  }
  if (d_q_) fvdb_dev_free(ctx_, d_q_);
  for (auto& ln : lanes_)
    if (ln.sc) fvdb_scorer_destroy(ln.sc);
  if (scorer_) fvdb_scorer_destroy(scorer_);
  if (store_) fvdb_store_destroy(store_);
}

bool HNSWIndex::entry_point(uint64_t* id) const {
  if (!has_entry_) return false;
  *id = ids_[entry_];
  return true;
}

// src/hnsw/core.rs:211-224
size_t HNSWIndex::assign_level() {
  size_t level = 0;
  while (rng_.gen_f64() < 0.408) level += 1;
  return level;
}

int HNSWIndex::ensure_store(uint32_t dim) {
  if (store_) return FVDB_OK;
  return fvdb_store_create(ctx_, dim, 1024, &store_);
}

int HNSWIndex::ensure_scorer(uint32_t B, uint32_t C) {
  if (scorer_ && scorer_B_ >= B && scorer_C_ >= C) return FVDB_OK;
  if (scorer_) fvdb_scorer_destroy(scorer_);
  scorer_ = nullptr;
  scorer_B_ = std::max(B, scorer_B_);
  scorer_C_ = std::max(C, scorer_C_);
  return fvdb_scorer_create(store_, scorer_B_, scorer_C_, &scorer_);
}

int HNSWIndex::append_row(const float* v, uint32_t* row) {
  uint64_t first = 0;
  int rc = fvdb_store_append(store_, v, 1, &first);
  if (rc) return rc;
  *row = (uint32_t)first;
  host_vecs_.insert(host_vecs_.end(), v, v + dim_);
  return FVDB_OK;
}

const float* HNSWIndex::vector_of(uint64_t id) const {
  auto it = index_of_.find(id);
  return it == index_of_.end() ? nullptr : &host_vecs_[(size_t)it->second * dim_];
}

int64_t HNSWIndex::level_of(uint64_t id) const {
  auto it = index_of_.find(id);
  return it == index_of_.end() ? -1 : (int64_t)level_[it->second];
}

int64_t HNSWIndex::neighbors(uint64_t id, uint32_t layer, uint64_t* out, uint64_t cap_out) {
  if (ensure_host_graph()) return -1;
  auto it = index_of_.find(id);
  if (it == index_of_.end() || layer > level_[it->second]) return -1;
  const auto& s = nbrs_[it->second][layer];
  for (size_t i = 0; i < s.size() && i < cap_out; ++i) out[i] = ids_[s[i]];
  return (int64_t)s.size();
}

// src/hnsw/operations.rs:127-137
int HNSWIndex::mark_deleted(uint64_t id) {
  auto it = index_of_.find(id);
  if (it == index_of_.end()) return FVDB_E_NOT_FOUND;
  deleted_[it->second] = 1;
  if (graph_ && !host_ahead_) fvdb_graph_set_deleted(graph_, it->second, 1);
  return FVDB_OK;
}
bool HNSWIndex::is_deleted(uint64_t id) const {
  auto it = index_of_.find(id);
  return it != index_of_.end() && deleted_[it->second];
}
uint64_t HNSWIndex::active_count() const {
  uint64_t n = 0;
  for (size_t i = 0; i < ids_.size(); ++i)
    if (registered_[i] && !deleted_[i]) ++n;
  return n;
}

// --------------------------------------------------------------------------------------------
// search_layer (src/hnsw/core.rs:469-554) for B queries in lock-step.  The queries are already
// in the scorer.  Per hop every active query pops candidates until one of them has unvisited,
// live neighbours; those go to the GPU in one launch; the admission rule (:517-531) is then
// applied in neighbour order with the returned distances.
// --------------------------------------------------------------------------------------------
int HNSWIndex::search_layer_batch(uint32_t B, const std::vector<Cand>& entries, const std::vector<uint8_t>& has_entry,
                                  uint32_t ef, uint32_t layer, std::vector<std::vector<Cand>>& results) {
  if (qs_.size() < B) qs_.resize(B);
  results.resize(B);
  const uint32_t maxC = scorer_C_;
  uint32_t* cand = fvdb_scorer_cand_buffer(scorer_);
  const float* dist = fvdb_scorer_dist_buffer(scorer_);
  static const int auto_threads = usable_cpus();
  const int nt = threads_ > 0 ? threads_ : auto_threads;
  const bool par = B >= 32 && nt > 1;
  std::vector<uint32_t> prev_cnt(B, maxC);  // rows start dirty: clear them on first use

#pragma omp parallel for schedule(static) num_threads(nt) if (par)
  for (uint32_t b = 0; b < B; ++b) {
    Query& s = qs_[b];
    s.candidates.clear();
    s.nearest.clear();
    s.pending.clear();
    s.active = false;
    results[b].clear();
    if (!has_entry[b]) continue;
    s.visited.reset((size_t)ef * 8 + 64);
    const Cand e = entries[b];
    s.candidates.push({e.node, e.distance});
    s.nearest.push({e.node, -e.distance});
    s.visited.insert(e.node);
    s.active = true;
  }

  using clk = std::chrono::steady_clock;
  auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (;;) {
    const auto t_a = clk::now();
    uint32_t hopC = 0;
    uint64_t hop_dists = 0;
#pragma omp parallel for schedule(static) num_threads(nt) reduction(max : hopC) reduction(+ : hop_dists) if (par)
    for (uint32_t b = 0; b < B; ++b) {
      Query& s = qs_[b];
      s.pending.clear();
      if (s.active) {
        for (;;) {
          if (s.candidates.empty()) {
            s.active = false;
            break;
          }
          const Cand cur = s.candidates.pop();
          if (cur.distance > -s.nearest.peek().distance) {  // :499-501
            s.active = false;
            break;
          }
          const uint32_t node = cur.node;
          if (registered_[node] && level_[node] >= layer) {
            for (uint32_t nbv : nbrs_[node][layer]) {
              if (!s.visited.insert(nbv)) continue;  // :506-507
              if (!registered_[nbv]) continue;       // nodes.get() == None (:509)
              if (deleted_[nbv]) continue;           // :511-513
              s.pending.push_back(nbv);
            }
          }
          if (!s.pending.empty()) break;
        }
      }
      uint32_t* row = cand + (size_t)b * maxC;
      const uint32_t n = (uint32_t)s.pending.size();
      for (uint32_t i = 0; i < n; ++i) row[i] = s.pending[i];
      for (uint32_t i = n; i < prev_cnt[b]; ++i) row[i] = FVDB_NO_ROW;
      prev_cnt[b] = n;
      hopC = std::max(hopC, n);
      hop_dists += n;
    }
    if (hopC == 0) break;
    static const bool dbg = getenv("FVDB_DEBUG") != nullptr;
    const auto t_b = clk::now();
    int rc = fvdb_scorer_run(scorer_, B, hopC);
    if (rc) return rc;
    const auto t_c = clk::now();
    n_hops_ += 1;
    n_dist_ += hop_dists;
#pragma omp parallel for schedule(static) num_threads(nt) if (par)
    for (uint32_t b = 0; b < B; ++b) {
      Query& s = qs_[b];
      const float* drow = dist + (size_t)b * maxC;
      for (size_t i = 0; i < s.pending.size(); ++i) {
        const float d = drow[i];
        if (d < -s.nearest.peek().distance || s.nearest.len() < ef) {  // :517-519
          s.candidates.push({s.pending[i], d});
          s.nearest.push({s.pending[i], -d});
          if (s.nearest.len() > ef) s.nearest.pop();
        }
      }
    }
    const auto t_d = clk::now();
    t_prepare_us_ += us(t_a, t_b);
    t_gpu_us_ += us(t_b, t_c);
    t_apply_us_ += us(t_c, t_d);
    (void)dbg;
  }

#pragma omp parallel for schedule(static) num_threads(nt) if (par)
  for (uint32_t b = 0; b < B; ++b) {
    if (!has_entry[b]) continue;
    Query& s = qs_[b];
    auto& r = results[b];
    r.reserve(s.nearest.len());
    for (const Cand& c : s.nearest.data) r.push_back({c.node, -c.distance});  // :541-547 (heap order)
    std::stable_sort(r.begin(), r.end(), [](const Cand& a, const Cand& c) { return a.distance < c.distance; });
  }
  return FVDB_OK;
}

// --------------------------------------------------------------------------------------------
// search (src/hnsw/core.rs:398-467), batched
// --------------------------------------------------------------------------------------------
int HNSWIndex::search(const float* q, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids, float* dist,
                      uint32_t* counts) {
  return search_impl(q, false, B, dim, k, ef, ids, dist, counts);
}
int HNSWIndex::search_dev(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids,
                          float* dist, uint32_t* counts) {
  return search_impl(q_dev, true, B, dim, k, ef, ids, dist, counts);
}

// ---- lane primitives: search_layer (:469-554) split into init / prepare-hop / apply-hop / collect ----
int HNSWIndex::lane_ensure(Lane& ln, uint32_t B, uint32_t C) {
  if (ln.sc && ln.cap_B >= B && ln.cap_C >= C) return FVDB_OK;
  if (ln.sc) fvdb_scorer_destroy(ln.sc);
  ln.sc = nullptr;
  ln.cap_B = std::max(B, ln.cap_B);
  ln.cap_C = std::max(C, ln.cap_C);
  return fvdb_scorer_create(store_, ln.cap_B, ln.cap_C, &ln.sc);
}

void HNSWIndex::lane_layer_init(Lane& ln) {
#pragma omp parallel for schedule(static) num_threads(ln.threads) if (ln.threads > 1)
  for (uint32_t b = 0; b < ln.n; ++b) {
    Query& s = ln.qs[b];
    s.candidates.clear();
    s.nearest.clear();
    s.pending.clear();
    s.visited.reset((size_t)ln.ef * 8 + 64);
    const Cand e = ln.cur[b][0];
    s.candidates.push({e.node, e.distance});
    s.nearest.push({e.node, -e.distance});
    s.visited.insert(e.node);
    s.active = true;
  }
}

uint32_t HNSWIndex::lane_hop_prepare(Lane& ln) {
  uint32_t* cand = fvdb_scorer_cand_buffer(ln.sc);
  const uint32_t maxC = ln.cap_C, layer = ln.layer;
  uint32_t hopC = 0;
  uint64_t nd = 0;
#pragma omp parallel for schedule(static) num_threads(ln.threads) reduction(max : hopC) reduction(+ : nd) if (ln.threads > 1)
  for (uint32_t b = 0; b < ln.n; ++b) {
    Query& s = ln.qs[b];
    s.pending.clear();
    if (s.active) {
      for (;;) {
        if (s.candidates.empty()) {
          s.active = false;
          break;
        }
        const Cand cur = s.candidates.pop();
        if (cur.distance > -s.nearest.peek().distance) {  // :499-501
          s.active = false;
          break;
        }
        const uint32_t node = cur.node;
        if (registered_[node] && level_[node] >= layer) {
          for (uint32_t nbv : nbrs_[node][layer]) {
            if (!s.visited.insert(nbv)) continue;  // :506-507
            if (!registered_[nbv]) continue;       // :509
            if (deleted_[nbv]) continue;           // :511-513
            s.pending.push_back(nbv);
          }
        }
        if (!s.pending.empty()) break;
      }
    }
    uint32_t* row = cand + (size_t)b * maxC;
    const uint32_t n = (uint32_t)s.pending.size();
    for (uint32_t i = 0; i < n; ++i) row[i] = s.pending[i];
    for (uint32_t i = n; i < ln.prev_cnt[b]; ++i) row[i] = FVDB_NO_ROW;
    ln.prev_cnt[b] = n;
    hopC = std::max(hopC, n);
    nd += n;
  }
  ln.dists += nd;
  return hopC;
}

void HNSWIndex::lane_hop_apply(Lane& ln) {
  const float* dist = fvdb_scorer_dist_buffer(ln.sc);
  const uint32_t maxC = ln.cap_C, ef = ln.ef;
#pragma omp parallel for schedule(static) num_threads(ln.threads) if (ln.threads > 1)
  for (uint32_t b = 0; b < ln.n; ++b) {
    Query& s = ln.qs[b];
    const float* drow = dist + (size_t)b * maxC;
    for (size_t i = 0; i < s.pending.size(); ++i) {
      const float d = drow[i];
      if (d < -s.nearest.peek().distance || s.nearest.len() < ef) {  // :517-519
        s.candidates.push({s.pending[i], d});
        s.nearest.push({s.pending[i], -d});
        if (s.nearest.len() > ef) s.nearest.pop();
      }
    }
  }
}

void HNSWIndex::lane_layer_collect(Lane& ln) {
#pragma omp parallel for schedule(static) num_threads(ln.threads) if (ln.threads > 1)
  for (uint32_t b = 0; b < ln.n; ++b) {
    Query& s = ln.qs[b];
    if (s.nearest.empty()) continue;  // search_layer returned nothing: keep the previous nearest (:445-447)
    auto& r = ln.cur[b];
    r.clear();
    r.reserve(s.nearest.len());
    for (const Cand& c : s.nearest.data) r.push_back({c.node, -c.distance});  // :541-547
    std::stable_sort(r.begin(), r.end(), [](const Cand& a, const Cand& c) { return a.distance < c.distance; });
  }
}

// One step of a lane's state machine: consume the GPU results that just arrived, do host work until
// the next GPU launch is issued (or the lane's queries are finished).
void HNSWIndex::lane_advance(Lane& ln, const float* q, bool q_on_device, uint32_t ef_final, uint64_t* ids,
                             float* dist, uint32_t* counts) {
  if (ln.done) return;
  if (ln.stage == 0) {  // load this lane's queries, score the entry point (:432-435)
    ln.rc = q_on_device ? fvdb_scorer_set_queries_dev(ln.sc, q + (size_t)ln.lo * dim_, ln.n)
                        : fvdb_scorer_set_queries(ln.sc, q + (size_t)ln.lo * dim_, ln.n);
    if (ln.rc) { ln.done = true; return; }
    uint32_t* cand = fvdb_scorer_cand_buffer(ln.sc);
    for (uint32_t b = 0; b < ln.n; ++b) {
      cand[(size_t)b * ln.cap_C] = entry_;
      ln.prev_cnt[b] = std::max<uint32_t>(ln.prev_cnt[b], 1);
    }
    ln.rc = fvdb_scorer_launch(ln.sc, ln.n, 1);
    if (ln.rc) { ln.done = true; return; }
    ln.dists += ln.n;
    ln.hops += 1;
    ln.stage = 1;
    return;
  }
  if (ln.stage == 1) {
    const float* dbuf = fvdb_scorer_dist_buffer(ln.sc);
    for (uint32_t b = 0; b < ln.n; ++b) ln.cur[b].assign(1, Cand{entry_, dbuf[(size_t)b * ln.cap_C]});
    ln.layer = level_[entry_];
    ln.ef = ln.layer == 0 ? ef_final : 1;
    lane_layer_init(ln);
    ln.stage = 2;
  } else {
    lane_hop_apply(ln);
  }
  for (;;) {
    const uint32_t C = lane_hop_prepare(ln);
    if (C > 0) {
      ln.rc = fvdb_scorer_launch(ln.sc, ln.n, C);
      if (ln.rc) ln.done = true;
      ln.hops += 1;
      return;
    }
    lane_layer_collect(ln);
    if (ln.layer == 0) break;
    ln.layer -= 1;
    ln.ef = ln.layer == 0 ? ef_final : 1;
    lane_layer_init(ln);
  }
  for (uint32_t b = 0; b < ln.n; ++b) {  // :451-466 filter deleted, take k
    uint32_t w = 0;
    const size_t o = (size_t)(ln.lo + b) * ln.k;
    for (const Cand& c : ln.cur[b]) {
      if (!registered_[c.node] || deleted_[c.node]) continue;
      if (w >= ln.k) break;
      ids[o + w] = ids_[c.node];
      dist[o + w] = c.distance;
      ++w;
    }
    counts[ln.lo + b] = w;
  }
  ln.done = true;
}

int HNSWIndex::ensure_graph_handle() {
  if (graph_) return FVDB_OK;
  int rc = fvdb_graph_create(store_, &graph_);
  if (rc) return rc;
  return fvdb_graph_configure(graph_, cfg_.max_connections, cfg_.max_connections_layer_0);
}

// host -> device: install the whole graph, once, when nbrs_ holds changes the device has not seen
int HNSWIndex::sync_graph() {
  std::lock_guard<std::mutex> lk(sync_mu_);  // the first of several concurrent searches after such a change uploads
  int rc = ensure_graph_handle();
  if (rc) return rc;
  if (!host_ahead_) return FVDB_OK;
  const uint32_t n = (uint32_t)ids_.size();
  if (n == 0) {
    host_ahead_ = false;
    return FVDB_OK;
  }
  std::vector<uint32_t> slot_start, adj;
  slot_start.reserve(graph_slots() + 1);
  uint64_t edges = 0;
  for (const auto& nd : nbrs_)
    for (const auto& l : nd) edges += l.size();
  adj.reserve(edges);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t l = 0; l <= level_[i]; ++l) {
      slot_start.push_back((uint32_t)adj.size());
      adj.insert(adj.end(), nbrs_[i][l].begin(), nbrs_[i][l].end());
    }
  slot_start.push_back((uint32_t)adj.size());
  rc = fvdb_graph_upload(graph_, n, level_.data(), deleted_.data(), slot_start.data(), adj.data(), entry_);
  if (rc) return rc;
  host_ahead_ = false;
  return FVDB_OK;
}

// device -> host: pull the lists device inserts have made
int HNSWIndex::ensure_host_graph() {
  std::lock_guard<std::mutex> lk(sync_mu_);
  if (!dev_ahead_) return FVDB_OK;
  const uint32_t n = (uint32_t)ids_.size();
  uint64_t slots = 0;
  for (uint32_t i = 0; i < n; ++i) slots += level_[i] + 1;
  std::vector<uint32_t> slot_start(slots + 1);
  uint64_t edges = 0;
  int rc = fvdb_graph_download(graph_, slot_start.data(), nullptr, 0, &edges);
  if (rc) return rc;
  std::vector<uint32_t> adj(std::max<uint64_t>(edges, 1));
  rc = fvdb_graph_download(graph_, slot_start.data(), adj.data(), adj.size(), &edges);
  if (rc) return rc;
  uint64_t sidx = 0;
  for (uint32_t i = 0; i < n; ++i) {
    nbrs_[i].resize(level_[i] + 1);
    for (uint32_t l = 0; l <= level_[i]; ++l, ++sidx)
      nbrs_[i][l].assign(adj.begin() + slot_start[sidx], adj.begin() + slot_start[sidx + 1]);
  }
  dev_ahead_ = false;
  return FVDB_OK;
}

bool HNSWIndex::device_path_ok(uint32_t ef) const {
  static const bool env_off = getenv("FVDB_HNSW_DEVICE") && atoi(getenv("FVDB_HNSW_DEVICE")) == 0;
  const uint32_t maxdeg = std::max(cfg_.max_connections, cfg_.max_connections_layer_0);
  return device_traversal_ && !env_off && maxdeg <= 64 && ef <= 4096;  // one lane per neighbour
}

// whole batch in one launch, results copied to pinned host memory — all asynchronous on the slot's stream
int HNSWIndex::device_launch(const float* q_dev, uint32_t B, uint32_t k, uint32_t ef, uint32_t slot) {
  if (slot >= kSlots) return FVDB_E_INVALID;
  int rc = sync_graph();
  if (rc) return rc;
  DevSlot& sl = slots_[slot];
  if (!sl.ctx) {
    if (slot == 0) {
      sl.ctx = ctx_;
    } else {
      rc = fvdb_ctx_create(fvdb_ctx_device(ctx_), &sl.ctx);
      if (rc) return rc;
    }
  }
  // one device block and one pinned block per slot: [nodes B*k | dist B*k | counts B | status B] -> a single copy
  const uint64_t need = (uint64_t)B * std::max<uint32_t>(k, 1);
  const uint64_t words = 2 * need + 2 * (uint64_t)B;
  if (words > sl.cap) {
    if (sl.d_nodes) fvdb_dev_free(sl.ctx, sl.d_nodes);
    if (sl.h_nodes) fvdb_host_free(sl.ctx, sl.h_nodes);
    sl.d_nodes = sl.h_nodes = nullptr;
    sl.cap = 0;
    if (fvdb_dev_alloc(sl.ctx, words * 4, &sl.d_nodes) || fvdb_host_alloc(sl.ctx, words * 4, &sl.h_nodes)) return FVDB_E_OOM;
    sl.cap = words;
  }
  auto carve = [&](void* base, void*& dist, void*& cnt, void*& status) {
    dist = (uint32_t*)base + need;
    cnt = (uint32_t*)base + 2 * need;
    status = (uint32_t*)base + 2 * need + B;
  };
  carve(sl.d_nodes, sl.d_dist, sl.d_cnt, sl.d_status);
  carve(sl.h_nodes, sl.h_dist, sl.h_cnt, sl.h_status);
  rc = fvdb_graph_search_dev_slot(graph_, slot == 0 ? nullptr : sl.ctx, slot, q_dev, B, k, ef, (uint32_t*)sl.d_nodes,
                                  (float*)sl.d_dist, (uint32_t*)sl.d_cnt, (uint32_t*)sl.d_status);
  if (!rc) rc = fvdb_dev_download_async(sl.ctx, sl.h_nodes, sl.d_nodes, (size_t)words * 4);
  return rc;
}

// wait + translate; queries the kernel could not finish on chip are listed in `failed`
int HNSWIndex::device_collect(uint32_t B, uint32_t k, uint64_t* ids, float* dist, uint32_t* counts,
                              std::vector<uint32_t>& failed, uint32_t slot) {
  DevSlot& sl = slots_[slot];
  int rc = fvdb_ctx_synchronize(sl.ctx);
  if (rc) return rc;
  const uint32_t* nodes = (const uint32_t*)sl.h_nodes;
  const uint32_t* status = (const uint32_t*)sl.h_status;
  std::memcpy(dist, sl.h_dist, (size_t)B * k * 4);
  std::memcpy(counts, sl.h_cnt, (size_t)B * 4);
  for (uint32_t b = 0; b < B; ++b) {
    if (status[b]) {
      failed.push_back(b);
      counts[b] = 0;
      continue;
    }
    for (uint32_t i = 0; i < k; ++i) {
      const uint32_t nd = nodes[(size_t)b * k + i];
      ids[(size_t)b * k + i] = i < counts[b] ? ids_[nd] : FVDB_NO_ID;
    }
  }
  return FVDB_OK;
}

// rare: on-chip heap / visited log overflow -> host walk for those queries
int HNSWIndex::finish_failed(const float* q, bool q_on_device, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids,
                             float* dist, uint32_t* counts, const std::vector<uint32_t>& failed) {
  if (failed.empty()) return FVDB_OK;
  n_fallback_ += failed.size();
  int rcg = ensure_host_graph();
  if (rcg) return rcg;
  std::lock_guard<std::mutex> lk(walk_mu_);  // the host walk's lanes and scorer are one set per index
  std::vector<float> hq((size_t)failed.size() * dim);
  for (size_t i = 0; i < failed.size(); ++i) {
    if (q_on_device) {
      int rc = fvdb_dev_download(ctx_, &hq[i * dim], q + (size_t)failed[i] * dim, (size_t)dim * 4);
      if (rc) return rc;
    } else {
      std::memcpy(&hq[i * dim], q + (size_t)failed[i] * dim, (size_t)dim * 4);
    }
  }
  std::vector<uint64_t> fi(failed.size() * (size_t)k);
  std::vector<float> fd(failed.size() * (size_t)k);
  std::vector<uint32_t> fc(failed.size());
  int rc = search_host_walk(hq.data(), false, (uint32_t)failed.size(), k, ef, fi.data(), fd.data(), fc.data());
  if (rc) return rc;
  for (size_t i = 0; i < failed.size(); ++i) {
    std::memcpy(ids + (size_t)failed[i] * k, &fi[i * k], (size_t)k * 8);
    std::memcpy(dist + (size_t)failed[i] * k, &fd[i * k], (size_t)k * 4);
    counts[failed[i]] = fc[i];
  }
  return FVDB_OK;
}

bool HNSWIndex::search_dev_begin(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, int* rc,
                                 uint32_t slot) {
  *rc = FVDB_OK;
  if (entry_lost_) return false;  // the caller's search_dev fallback reports the error
  if (!has_entry_ || B == 0 || k == 0 || (has_dim_ && dim != dim_) || !device_path_ok(ef)) return false;
  *rc = device_launch(q_dev, B, k, ef, slot);
  return *rc == FVDB_OK;
}

int HNSWIndex::search_dev_end(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids,
                              float* dist, uint32_t* counts, uint32_t slot) {
  std::vector<uint32_t> failed;
  int rc = device_collect(B, k, ids, dist, counts, failed, slot);
  if (rc) return rc;
  return finish_failed(q_dev, true, dim, k, ef, ids, dist, counts, failed);
}

int HNSWIndex::search_impl(const float* q, bool q_on_device, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef,
                           uint64_t* ids, float* dist, uint32_t* counts) {
  for (uint32_t b = 0; b < B; ++b) counts[b] = 0;
  for (size_t i = 0; i < (size_t)B * k; ++i) {
    ids[i] = FVDB_NO_ID;
    dist[i] = __builtin_huge_valf();
  }
  if (!has_entry_) return FVDB_OK;  // empty index -> empty results (:404-407)
  if (has_dim_ && dim != dim_) return FVDB_E_DIM;
  if (entry_lost_) return FVDB_E_NOT_FOUND;  // "Entry point node not found in index" (:422-429)
  if (B == 0 || k == 0) return FVDB_OK;
  // standalone entry point: slot 0 and one staging buffer => calls from several host threads take turns here
  // (HybridIndex gives each of its concurrent searches a slot of its own through search_dev_begin/_end)
  std::lock_guard<std::mutex> serial(search_mu_);
  if (!device_path_ok(ef)) {
    int rcg = ensure_host_graph();
    if (rcg) return rcg;
    std::lock_guard<std::mutex> lk(walk_mu_);
    return search_host_walk(q, q_on_device, B, k, ef, ids, dist, counts);
  }
  const float* qd = q;
  if (!q_on_device) {  // stage the batch in HBM
    const uint64_t bytes = (uint64_t)B * dim * 4;
    if (bytes > d_q_cap_) {
      if (d_q_) fvdb_dev_free(ctx_, d_q_);
      d_q_ = nullptr;
      d_q_cap_ = 0;
      if (fvdb_dev_alloc(ctx_, bytes, &d_q_)) return FVDB_E_OOM;
      d_q_cap_ = bytes;
    }
    int rc = fvdb_dev_upload(ctx_, d_q_, q, bytes);
    if (rc) return rc;
    qd = (const float*)d_q_;
  }
  int rc = device_launch(qd, B, k, ef, 0);
  if (rc) return rc;
  std::vector<uint32_t> failed;
  rc = device_collect(B, k, ids, dist, counts, failed, 0);
  if (rc) return rc;
  return finish_failed(q, q_on_device, dim, k, ef, ids, dist, counts, failed);
}

int HNSWIndex::search_host_walk(const float* q, bool q_on_device, uint32_t B, uint32_t k, uint32_t ef, uint64_t* ids,
                                float* dist, uint32_t* counts) {
  for (uint32_t b = 0; b < B; ++b) counts[b] = 0;
  for (size_t i = 0; i < (size_t)B * k; ++i) {
    ids[i] = FVDB_NO_ID;
    dist[i] = __builtin_huge_valf();
  }
  static const int auto_threads = usable_cpus();
  const int nt = std::max(1, threads_ > 0 ? threads_ : auto_threads);
  const uint32_t maxdeg = std::max(cfg_.max_connections, cfg_.max_connections_layer_0) + 1;
  // Lanes (one HIP stream each) can be driven round-robin by the calling thread so that one lane's hop
  // is on the GPU while another's host phase runs.  Measured on MI355X (profiles/r01_hnsw_lanes.log):
  // launch + stream sync cost more than the overlap wins — 1 lane 15.5 ms/step, 2 lanes 18.1, 3 lanes
  // 27.2 (1024 queries, 300K nodes) — so the default is ONE lane = one launch per hop for the whole batch.
  static const uint32_t max_lanes = getenv("FVDB_HNSW_LANES") ? std::max(1, atoi(getenv("FVDB_HNSW_LANES"))) : 1;
  const uint32_t step = 16384;
  for (uint32_t o = 0; o < B; o += step) {
    const uint32_t b = std::min(step, B - o);
    uint32_t nl = std::min<uint32_t>(max_lanes, std::max<uint32_t>(1, b / 64));
    const uint32_t per = (b + nl - 1) / nl;
    nl = (b + per - 1) / per;
    if (lanes_.size() < nl) lanes_.resize(nl);
    for (uint32_t l = 0; l < nl; ++l) {
      Lane& ln = lanes_[l];
      int rc = lane_ensure(ln, per, maxdeg);
      if (rc) return rc;
      ln.lo = o + l * per;
      ln.n = std::min(per, o + b - ln.lo);
      ln.k = k;
      ln.stage = 0;
      ln.done = false;
      ln.rc = 0;
      ln.dists = ln.hops = 0;
      ln.threads = ln.n >= 32 ? nt : 1;
      if (ln.qs.size() < ln.n) ln.qs.resize(ln.n);
      if (ln.cur.size() < ln.n) ln.cur.resize(ln.n);
      ln.prev_cnt.assign(ln.cap_B, ln.cap_C);  // rows start dirty: cleared on first use
    }
    uint32_t remaining = 0;
    for (uint32_t l = 0; l < nl; ++l) {
      lane_advance(lanes_[l], q, q_on_device, ef, ids, dist, counts);
      if (!lanes_[l].done) ++remaining;
    }
    while (remaining) {
      for (uint32_t l = 0; l < nl; ++l) {
        Lane& ln = lanes_[l];
        if (ln.done) continue;
        ln.rc = fvdb_scorer_wait(ln.sc);
        if (ln.rc) {
          ln.done = true;
        } else {
          lane_advance(ln, q, q_on_device, ef, ids, dist, counts);
        }
        if (ln.done) --remaining;
      }
    }
    for (uint32_t l = 0; l < nl; ++l) {
      n_dist_ += lanes_[l].dists;
      n_hops_ += lanes_[l].hops;
      if (lanes_[l].rc) return lanes_[l].rc;
    }
  }
  return FVDB_OK;
}

// distances from stored row `base_row` to a list of candidate rows (prune: :588-624)
int HNSWIndex::score_pairs_from_row(uint32_t base_row, const std::vector<uint32_t>& cands, std::vector<float>& out) {
  int rc = fvdb_scorer_set_query_rows(scorer_, &base_row, 1);
  if (rc) return rc;
  uint32_t* cand = fvdb_scorer_cand_buffer(scorer_);
  for (size_t i = 0; i < cands.size(); ++i) cand[i] = cands[i];
  rc = fvdb_scorer_run(scorer_, 1, (uint32_t)cands.size());
  if (rc) return rc;
  n_dist_ += cands.size();
  n_hops_ += 1;
  const float* d = fvdb_scorer_dist_buffer(scorer_);
  out.assign(d, d + cands.size());
  for (size_t i = 0; i < cands.size(); ++i) cand[i] = FVDB_NO_ROW;
  return FVDB_OK;
}

// --------------------------------------------------------------------------------------------
// insert (src/hnsw/core.rs:226-378)
// --------------------------------------------------------------------------------------------
// the reference's checks, in its order (:227-245); non-finite values would panic its partial_cmp().unwrap()
int HNSWIndex::check_insert(uint64_t id, const float* v, uint32_t dim) const {
  if (index_of_.count(id)) return FVDB_E_DUPLICATE;
  if (has_dim_ && dim != dim_) return FVDB_E_DIM;
  if (entry_lost_) return FVDB_E_NOT_FOUND;  // the reference unwraps the missing entry node here (:268-274)
  for (uint32_t j = 0; j < dim; ++j)
    if (!(v[j] - v[j] == 0.0f)) return FVDB_E_NONFINITE;
  return FVDB_OK;
}

bool HNSWIndex::device_insert_ok() const {
  static const bool env_off = getenv("FVDB_HNSW_DEVICE_INSERT") && atoi(getenv("FVDB_HNSW_DEVICE_INSERT")) == 0;
  return device_insert_ && !env_off && cfg_.max_connections <= 63 && cfg_.max_connections_layer_0 <= 63 &&
         cfg_.ef_construction >= 1 && cfg_.ef_construction <= 512 && dim_ <= 1024;
}

int HNSWIndex::insert(uint64_t id, const float* v, uint32_t dim, int64_t forced_level) {
  uint64_t ok = 0;
  int err = 0;
  int rc = batch_insert(&id, v, 1, dim, &forced_level, &ok, &err);
  return rc ? rc : err;
}

void HNSWIndex::finalize_insert(uint32_t row) {  // "nodes.insert(id, node)" (:367-370)
  index_of_[ids_[row]] = row;
  registered_[row] = 1;
  n_registered_ += 1;
}

int HNSWIndex::batch_insert(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const int64_t* levels, uint64_t* n_ok,
                            int* first_error) {
  if (n_ok) *n_ok = 0;
  if (first_error) *first_error = 0;
  uint64_t ok = 0;
  int err = 0;
  auto finish = [&](int rc) {
    if (n_ok) *n_ok = ok;
    if (first_error) *first_error = err;
    return rc;
  };
  for (uint64_t at = 0; at < n;) {
    // the next run of inserts that pass the reference's checks (a failed insert draws no level and changes nothing)
    std::vector<uint64_t> acc;
    std::unordered_set<uint64_t> in_run;
    uint64_t end = at;
    for (; end < n && acc.size() < (1u << 20); ++end) {
      int rc = check_insert(ids[end], v + end * dim, dim);
      if (!rc && in_run.count(ids[end])) rc = FVDB_E_DUPLICATE;
      if (rc) {
        if (!err) err = rc;
        continue;
      }
      if (!has_dim_) {
        dim_ = dim;
        has_dim_ = true;
      }
      in_run.insert(ids[end]);
      acc.push_back(end);
    }
    at = end;
    if (acc.empty()) continue;
    if (!device_insert_ok()) {
      for (uint64_t i : acc) {
        int rc = insert_host(ids[i], v + i * dim, dim, levels ? levels[i] : -1);
        if (rc == 0) ++ok;
        else if (!err) err = rc;
      }
      continue;
    }
    static const bool dbg = getenv("FVDB_BUILD_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_a = now();
    int rc = ensure_store(dim);
    if (rc) return finish(rc);
    rc = sync_graph();  // the device rows must be current before they are edited in place
    if (rc) return finish(rc);
    const auto t_b = now();
    const uint32_t m = (uint32_t)acc.size();
    std::vector<uint32_t> lv(m);
    for (uint32_t j = 0; j < m; ++j) {
      const int64_t f = levels ? levels[acc[j]] : -1;
      lv[j] = f >= 0 ? (uint32_t)f : (uint32_t)assign_level();
    }
    // vectors: one upload
    const float* src = v + acc[0] * dim;
    std::vector<float> packed;
    if (acc.back() - acc[0] + 1 != m) {
      packed.resize((size_t)m * dim);
      for (uint32_t j = 0; j < m; ++j) std::memcpy(&packed[(size_t)j * dim], v + acc[j] * dim, (size_t)dim * 4);
      src = packed.data();
    }
    uint64_t first64 = 0;
    rc = fvdb_store_append(store_, src, m, &first64);
    if (rc) return finish(rc);
    const uint32_t first = (uint32_t)first64;
    host_vecs_.insert(host_vecs_.end(), src, src + (size_t)m * dim);
    for (uint32_t j = 0; j < m; ++j) {
      ids_.push_back(ids[acc[j]]);
      level_.push_back(lv[j]);
      deleted_.push_back(0);
      registered_.push_back(0);  // "not yet in the nodes map" while its links are being made (:370)
      nbrs_.emplace_back(lv[j] + 1);
    }
    const auto t_c = now();
    rc = fvdb_graph_append_nodes(graph_, first, m, lv.data());
    if (rc) return finish(rc);
    const auto t_d = now();
    uint32_t done = 0;
    while (done < m) {
      uint32_t nd = 0;
      fvdb_graph_insert_stats st{};
      rc = fvdb_graph_insert_linked(graph_, first + done, m - done, cfg_.ef_construction, insert_mode_, &nd, &st);
      if (rc == FVDB_E_UNSUPPORTED) {  // e.g. a graph too large for the on-chip visited bitmap: the host algorithm links
        rc = FVDB_OK;                  // this node (rows patched into HBM), and the question is asked again for the next
        nd = 0;
        st.needs_host = 1;
      }
      if (rc) return finish(rc);
      insert_stats_.n_done += st.n_done;
      insert_stats_.speculated_ok += st.speculated_ok;
      insert_stats_.searched_in_commit += st.searched_in_commit;
      insert_stats_.commit_stops += st.commit_stops;
      insert_stats_.rounds += st.rounds;
      insert_stats_.expanded += st.expanded;
      insert_stats_.rows_scored += st.rows_scored;
      insert_stats_.tie_restarts += st.tie_restarts;
      insert_stats_.launches += st.launches;
      n_dist_ += st.rows_scored;  // of the searches the commit workgroup ran (speculated ones are not counted)
      n_hops_ += st.rounds;
      for (uint32_t j = 0; j < nd; ++j) finalize_insert(first + done + j);
      done += nd;
      dev_ahead_ = dev_ahead_ || nd > 0;
      uint32_t e = FVDB_NO_ROW;
      fvdb_graph_entry(graph_, &e, nullptr);
      if (e != FVDB_NO_ROW) {
        entry_ = e;
        has_entry_ = true;
      }
      if (done < m && st.needs_host) {  // level >= 16 or an on-chip heap outgrown: this node takes the host algorithm
        const uint32_t row = first + done;
        rc = ensure_host_graph();
        if (rc) return finish(rc);
        std::vector<std::pair<uint32_t, uint32_t>> touched;
        rc = link_host(row, &touched);
        if (rc) return finish(rc);
        finalize_insert(row);
        rc = push_lists(touched);
        if (rc) return finish(rc);
        rc = fvdb_graph_set_entry(graph_, entry_, row + 1);
        if (rc) return finish(rc);
        n_host_inserts_ += 1;
        done += 1;
      } else if (nd == 0 && done < m) {
        return finish(FVDB_E_HIP);  // no progress and no request for the host path: never expected
      }
    }
    ok += m;
    if (dbg)
      fprintf(stderr, "[batch_insert] %u rows: graph sync %.1f ms, vectors + host bookkeeping %.1f ms, node append %.1f ms, linking %.1f ms\n", m,
              ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d), ms(t_d, now()));
  }
  return finish(FVDB_OK);
}

// rows `touched` (node, layer) -> the device graph
int HNSWIndex::push_lists(const std::vector<std::pair<uint32_t, uint32_t>>& touched) {
  std::vector<uint32_t> nodes, layers, off{0}, flat;
  for (const auto& t : touched) {
    nodes.push_back(t.first);
    layers.push_back(t.second);
    const auto& l = nbrs_[t.first][t.second];
    flat.insert(flat.end(), l.begin(), l.end());
    off.push_back((uint32_t)flat.size());
  }
  if (flat.empty()) flat.push_back(0);
  return fvdb_graph_set_lists(graph_, (uint32_t)nodes.size(), nodes.data(), layers.data(), off.data(), flat.data());
}

// one insert by the host algorithm (the other mode; also dims / degree caps the device insert does not take)
int HNSWIndex::insert_host(uint64_t id, const float* v, uint32_t dim, int64_t forced_level) {
  int rc = ensure_store(dim);
  if (rc) return rc;
  rc = ensure_host_graph();
  if (rc) return rc;
  const uint32_t level = forced_level >= 0 ? (uint32_t)forced_level : (uint32_t)assign_level();
  uint32_t row = 0;
  rc = append_row(v, &row);  // the vector is resident before the graph links to it
  if (rc) return rc;
  ids_.push_back(id);
  level_.push_back(level);
  deleted_.push_back(0);
  registered_.push_back(0);  // "not yet in the nodes map" while its links are being made (:370)
  nbrs_.emplace_back(level + 1);
  std::vector<std::pair<uint32_t, uint32_t>> touched;
  rc = link_host(row, &touched);
  if (rc) return rc;
  finalize_insert(row);
  n_host_inserts_ += 1;
  if (graph_ && !host_ahead_) {  // the device copy agrees with nbrs_ up to this insert: patch the rows it changed
    rc = fvdb_graph_append_nodes(graph_, row, 1, &level);
    if (rc) return rc;
    rc = push_lists(touched);
    if (rc) return rc;
    rc = fvdb_graph_set_entry(graph_, entry_, row + 1);
    if (rc) return rc;
  } else {
    host_ahead_ = true;
  }
  return FVDB_OK;
}

int HNSWIndex::link_host(uint32_t row, std::vector<std::pair<uint32_t, uint32_t>>* touched) {
  const uint32_t level = level_[row];
  bool is_first = false;
  if (!has_entry_) {
    has_entry_ = true;
    entry_ = row;
    is_first = true;
  }
  for (uint32_t lc = 0; lc <= level; ++lc) touched->push_back({row, lc});
  uint32_t entry_level = 0;
  int rc = FVDB_OK;
  if (!is_first) {
    const uint32_t ep = entry_;
    entry_level = level_[ep];
    const uint32_t maxdeg = std::max(cfg_.max_connections, cfg_.max_connections_layer_0) + 1;
    rc = ensure_scorer(1, maxdeg);
    if (rc) return rc;
    std::vector<float> d0;
    rc = score_pairs_from_row(row, {ep}, d0);  // also loads the new vector as the scorer's query
    if (rc) return rc;
    std::vector<Cand> current_nearest{{ep, d0[0]}};
    const uint32_t search_level = std::min(level, entry_level);
    std::vector<uint8_t> has1(1, 1);
    std::vector<std::vector<Cand>> res;
    for (uint32_t lc = search_level + 1; lc-- > 0;) {  // :284-290
      rc = search_layer_batch(1, {current_nearest[0]}, has1, 1, lc, res);
      if (rc) return rc;
      if (!res[0].empty()) current_nearest = res[0];
    }
    for (uint32_t lc = 0; lc <= level; ++lc) {  // :293-362
      const uint32_t m = cap(lc);
      const Cand start = (lc <= search_level && !current_nearest.empty()) ? current_nearest[0] : Cand{ep, d0[0]};
      rc = search_layer_batch(1, {start}, has1, cfg_.ef_construction, lc, res);
      if (rc) return rc;
      std::vector<uint32_t> chosen;  // select_neighbors :556-558
      for (size_t i = 0; i < res[0].size() && i < m; ++i) chosen.push_back(res[0][i].node);
      for (uint32_t nbv : chosen) set_insert(nbrs_[row][lc], nbv);
      std::vector<uint32_t> to_prune;
      for (uint32_t nbv : chosen) {
        if (!registered_[nbv]) continue;
        if (level_[nbv] >= lc) {
          set_insert(nbrs_[nbv][lc], row);
          touched->push_back({nbv, lc});
          if (nbrs_[nbv][lc].size() > m) to_prune.push_back(nbv);
        }
      }
      for (uint32_t nbv : to_prune) {  // prune_neighbors_with_new_node :588-624
        const std::vector<uint32_t> list = nbrs_[nbv][lc];
        std::vector<float> ds;
        rc = score_pairs_from_row(nbv, list, ds);
        if (rc) return rc;
        std::vector<Cand> cs(list.size());
        for (size_t i = 0; i < list.size(); ++i) cs[i] = {list[i], ds[i]};
        std::stable_sort(cs.begin(), cs.end(), [](const Cand& a, const Cand& c) { return a.distance < c.distance; });
        if (cs.size() > m) cs.resize(m);
        nbrs_[nbv][lc].clear();
        for (const Cand& c : cs) set_insert(nbrs_[nbv][lc], c.node);
      }
      if (!to_prune.empty()) {  // put the new vector back as the query for the next layer
        rc = fvdb_scorer_set_query_rows(scorer_, &row, 1);
        if (rc) return rc;
      }
    }
  }
  if (!is_first && level > entry_level) entry_ = row;  // :372-375
  return FVDB_OK;
}

// --------------------------------------------------------------------------------------------
// graph install / export, bulk build
// --------------------------------------------------------------------------------------------
int HNSWIndex::restore(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const uint32_t* levels,
                       const uint64_t* nbr_offsets, const uint64_t* nbrs, uint64_t entry_id) {
  if (!ids_.empty()) return FVDB_E_INVALID;
  if (n == 0) return FVDB_OK;
  dim_ = dim;
  has_dim_ = true;
  int rc = ensure_store(dim);
  if (rc) return rc;
  uint64_t first = 0;
  rc = fvdb_store_append(store_, v, n, &first);
  if (rc) return rc;
  host_vecs_.assign(v, v + n * dim);
  ids_.assign(ids, ids + n);
  level_.assign(levels, levels + n);
  deleted_.assign(n, 0);
  registered_.assign(n, 1);
  n_registered_ = n;
  index_of_.reserve(n * 2);
  for (uint64_t i = 0; i < n; ++i) index_of_[ids[i]] = (uint32_t)i;
  nbrs_.assign(n, {});
  uint64_t slot = 0;
  for (uint64_t i = 0; i < n; ++i) {
    nbrs_[i].resize(levels[i] + 1);
    for (uint32_t l = 0; l <= levels[i]; ++l, ++slot) {
      auto& s = nbrs_[i][l];
      for (uint64_t e = nbr_offsets[slot]; e < nbr_offsets[slot + 1]; ++e) {
        auto it = index_of_.find(nbrs[e]);
        if (it == index_of_.end()) return FVDB_E_NOT_FOUND;
        s.push_back(it->second);
      }
    }
  }
  auto it = index_of_.find(entry_id);
  if (it == index_of_.end()) return FVDB_E_NOT_FOUND;
  entry_ = it->second;
  has_entry_ = true;
  host_ahead_ = true;
  return FVDB_OK;
}

// src/hnsw/operations.rs:176-200: deleted nodes leave the node map and every neighbour set.  Their rows stay in HBM
// unreferenced (the row store is append-only); `registered_ == 0` is this mirror's "nodes.get() == None".  The
// reference does not repair an entry point that was removed: its searches then fail and its insert panics.
uint64_t HNSWIndex::vacuum() {
  if (ensure_host_graph()) return 0;
  std::vector<uint8_t> dead(ids_.size(), 0);
  uint64_t removed = 0;
  for (size_t i = 0; i < ids_.size(); ++i)
    if (registered_[i] && deleted_[i]) {
      dead[i] = 1;
      ++removed;
    }
  if (removed == 0) return 0;
  for (size_t i = 0; i < ids_.size(); ++i) {
    if (dead[i]) {
      registered_[i] = 0;
      auto it = index_of_.find(ids_[i]);
      if (it != index_of_.end() && it->second == i) index_of_.erase(it);
      for (auto& l : nbrs_[i]) l.clear();
    } else if (registered_[i]) {
      for (auto& l : nbrs_[i])
        l.erase(std::remove_if(l.begin(), l.end(), [&](uint32_t x) { return dead[x] != 0; }), l.end());
    }
  }
  n_registered_ -= removed;
  if (has_entry_ && dead[entry_]) entry_lost_ = true;
  host_ahead_ = true;
  return removed;
}

uint64_t HNSWIndex::graph_slots() const {
  uint64_t s = 0;
  for (size_t i = 0; i < ids_.size(); ++i)
    if (registered_[i]) s += level_[i] + 1;
  return s;
}
uint64_t HNSWIndex::graph_edges() {
  if (ensure_host_graph()) return 0;
  uint64_t e = 0;
  for (const auto& n : nbrs_)
    for (const auto& l : n) e += l.size();
  return e;
}
void HNSWIndex::export_graph(uint64_t* ids, uint32_t* levels, uint64_t* nbr_offsets, uint64_t* nbrs) {
  (void)ensure_host_graph();
  uint64_t slot = 0, e = 0, w = 0;
  for (size_t i = 0; i < ids_.size(); ++i) {
    if (!registered_[i]) continue;  // vacuumed
    ids[w] = ids_[i];
    levels[w++] = level_[i];
    for (uint32_t l = 0; l <= level_[i]; ++l, ++slot) {
      nbr_offsets[slot] = e;
      for (uint32_t x : nbrs_[i][l]) nbrs[e++] = ids_[x];
    }
  }
  nbr_offsets[slot] = e;
}

int HNSWIndex::bulk_build(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const int64_t* levels) {
  if (!ids_.empty()) return FVDB_E_INVALID;
  if (n == 0) return FVDB_OK;
  if (n >= 0xFFFFFFFFull) return FVDB_E_UNSUPPORTED;
  {
    std::unordered_set<uint64_t> seen;
    seen.reserve(n * 2);
    for (uint64_t i = 0; i < n; ++i)
      if (!seen.insert(ids[i]).second) return FVDB_E_DUPLICATE;
  }
  dim_ = dim;
  has_dim_ = true;
  int rc = ensure_store(dim);
  if (rc) return rc;
  uint64_t first = 0;
  rc = fvdb_store_append(store_, v, n, &first);
  if (rc) return rc;
  host_vecs_.assign(v, v + n * dim);
  ids_.assign(ids, ids + n);
  level_.resize(n);
  uint32_t maxlevel = 0;
  for (uint64_t i = 0; i < n; ++i) {
    level_[i] = levels ? (uint32_t)levels[i] : (uint32_t)assign_level();
    maxlevel = std::max(maxlevel, level_[i]);
  }
  deleted_.assign(n, 0);
  registered_.assign(n, 1);
  n_registered_ = n;
  index_of_.reserve(n * 2);
  for (uint64_t i = 0; i < n; ++i) index_of_[ids[i]] = (uint32_t)i;
  nbrs_.assign(n, {});
  for (uint64_t i = 0; i < n; ++i) nbrs_[i].resize(level_[i] + 1);
  // entry = earliest node carrying the maximum level (what sequential insertion ends with, :372-375)
  for (uint64_t i = 0; i < n; ++i)
    if (level_[i] == maxlevel) {
      entry_ = (uint32_t)i;
      break;
    }
  has_entry_ = true;

  std::vector<float> zero(dim, 0.0f);
  for (uint32_t l = 0; l <= maxlevel; ++l) {
    std::vector<uint32_t> members;
    for (uint64_t i = 0; i < n; ++i)
      if (level_[i] >= l) members.push_back((uint32_t)i);
    if (members.size() < 2) continue;
    const uint32_t kk = (uint32_t)std::min<uint64_t>(cap(l), members.size() - 1);
    const uint32_t k = kk + 1;  // the member itself comes back at distance 0
    if (k > FVDB_MAX_K) return FVDB_E_UNSUPPORTED;
    fvdb_ivf* flat = nullptr;
    rc = fvdb_ivf_create(ctx_, dim, 1, &flat);
    if (rc) return rc;
    rc = fvdb_ivf_set_centroids(flat, zero.data());
    std::vector<float> mv;
    std::vector<uint64_t> mid(members.size());
    std::vector<uint32_t> mcl(members.size(), 0);
    const float* src = v;
    if (rc == FVDB_OK && members.size() != n) {
      mv.resize(members.size() * (size_t)dim);
      for (size_t j = 0; j < members.size(); ++j)
        std::memcpy(&mv[j * dim], v + (size_t)members[j] * dim, dim * sizeof(float));
      src = mv.data();
    }
    for (size_t j = 0; j < members.size(); ++j) mid[j] = members[j];
    if (rc == FVDB_OK) rc = fvdb_ivf_add_assigned(flat, src, mid.data(), members.size(), mcl.data(), nullptr);
    const uint32_t step = 16384;
    std::vector<uint64_t> oi((size_t)step * k);
    std::vector<float> od((size_t)step * k);
    std::vector<uint32_t> oc(step);
    for (size_t o = 0; rc == FVDB_OK && o < members.size(); o += step) {
      const uint32_t b = (uint32_t)std::min<size_t>(step, members.size() - o);
      rc = fvdb_ivf_search_all(flat, src + o * dim, b, k, oi.data(), od.data(), oc.data());
      if (rc) break;
      for (uint32_t i = 0; i < b; ++i) {
        const uint32_t self = members[o + i];
        auto& s = nbrs_[self][l];
        for (uint32_t e = 0; e < oc[i] && s.size() < kk; ++e) {
          const uint32_t nbv = (uint32_t)oi[(size_t)i * k + e];
          if (nbv != self) s.push_back(nbv);
        }
      }
    }
    fvdb_ivf_destroy(flat);
    if (rc) return rc;
  }
  host_ahead_ = true;
  return FVDB_OK;
}

}  // namespace fvdbh
