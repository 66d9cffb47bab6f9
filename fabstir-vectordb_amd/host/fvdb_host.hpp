// fvdb_host.hpp — host-side mirror of the reference's index classes for the hot path.
//
// The reference's host code is Rust (src/ivf/core.rs, src/hnsw/core.rs, src/hybrid/core.rs);
// no Rust toolchain exists here, so the mirror is C++ above the C ABI (include/fvdb.h), with
// the same names, argument meaning and error behaviour.  It owns the index STRUCTURES (lists
// bookkeeping, HNSW graph, heaps, visited sets, routing, timestamps); every distance comes
// from the GPU through the C ABI.  Nothing in here computes a vector distance on the CPU.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/fvdb.h"

namespace fvdbh {

// Documented PRNG for this build's own draws (HNSW levels, k-means++): SplitMix64.
// The reference uses rand 0.8 StdRng whose stream cannot be reproduced here.
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double gen_f64() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct Cand {
  uint32_t node;   // node index (HNSW) — ids are resolved at the API boundary
  float distance;
};

// std::collections::BinaryHeap restated (sift_up / sift_down_to_bottom), so exact distance
// ties resolve like the reference's heaps.  Ordering: SearchCandidate::cmp, src/hnsw/core.rs:126-137.
struct RustHeap {
  std::vector<Cand> data;
  static bool le(const Cand& a, const Cand& b) { return a.distance >= b.distance; }
  size_t len() const { return data.size(); }
  bool empty() const { return data.empty(); }
  const Cand& peek() const { return data[0]; }
  void clear() { data.clear(); }
  void sift_up(size_t start, size_t pos);
  void push(Cand c);
  Cand pop();
};

// Open-addressing visited set with O(1) reset.
struct Visited {
  std::vector<uint32_t> keys, stamp;
  uint32_t epoch = 0, used = 0;
  void reset(size_t expect);
  bool insert(uint32_t v);  // true if newly inserted
 private:
  void grow();
};

// ------------------------------------------------------------------------------------------
// IVFIndex — src/ivf/core.rs, src/ivf/operations.rs
// ------------------------------------------------------------------------------------------
struct IVFConfig {
  uint32_t n_clusters = 256, n_probe = 16, train_size = 10000, max_iterations = 25;  // :50-60
  uint64_t seed = 0;
  bool is_valid() const { return n_clusters > 0 && n_probe > 0 && n_probe <= n_clusters && train_size > 0 && max_iterations > 0; }
};

class IVFIndex {
 public:
  IVFIndex(fvdb_ctx* ctx, const IVFConfig& cfg);
  ~IVFIndex();
  const IVFConfig& config() const { return cfg_; }
  bool is_trained() const { return trained_; }
  uint32_t dimension() const { return dim_; }
  uint64_t total_vectors() const { return total_; }
  int train(const float* data, uint64_t n, uint32_t dim, fvdb_train_result* out);  // :240
  int set_trained(const float* centroids, uint32_t dim);                          // :509
  int get_centroids(float* out) const;
  int insert(uint64_t id, const float* v, uint32_t dim);                          // :431
  // batch_insert (operations.rs:107-130): sequential semantics, one GPU assignment pass.
  int batch_insert(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, uint64_t* n_ok, int* first_error);
  // rows whose list is already known (load path / shard placement)
  int batch_insert_assigned(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const uint32_t* clusters,
                            uint64_t* n_ok, int* first_error);
  int assign(const float* v, uint64_t n, uint32_t dim, uint32_t* out);
  int find_cluster(const float* v, uint32_t dim, uint32_t* out);                  // :493
  int search(const float* q, uint32_t B, uint32_t dim, uint32_t k, uint32_t n_probe, uint64_t* ids, float* dist,
             uint32_t* counts);                                                   // :626, operations.rs:132
  // same with the queries already resident in HBM (B x d row-major); outputs are device pointers
  // `on` / `slot`: run on another context's stream with that slot's scratch set (several searches in flight)
  int search_dev(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t n_probe, uint64_t* ids_dev,
                 float* dist_dev, uint32_t* counts_dev, fvdb_ctx* on = nullptr, uint32_t slot = 0);
  int mark_deleted(uint64_t id);                                                  // operations.rs:569
  bool is_deleted(uint64_t id) const { return deleted_.count(id) > 0; }
  uint64_t active_count() const { return total_ - deleted_.size(); }
  uint64_t deleted_count() const { return deleted_.size(); }
  int vacuum(uint64_t* removed);                                                  // operations.rs:625
  uint64_t cluster_size(uint32_t c) const;
  // list `c` in list-position order, copied back from HBM (save path, src/hybrid/persistence.rs:289-311)
  int export_list(uint32_t c, float* rows, uint64_t* ids, uint8_t* live) const;
  void clear_lists();                                                             // hybrid initialize :278-287
  fvdb_ivf* device() { return dev_; }

 private:
  struct Loc {
    uint32_t cluster, pos;
  };
  int ensure_device(uint32_t dim);
  int place(const uint64_t* ids, const float* v, uint64_t n, const uint32_t* clusters, uint64_t* n_ok, int* first_error);
  fvdb_ctx* ctx_;
  IVFConfig cfg_;
  fvdb_ivf* dev_ = nullptr;
  uint32_t dim_ = 0;
  bool trained_ = false;
  uint64_t total_ = 0;
  std::unordered_multimap<uint64_t, Loc> where_;  // id -> every list position holding it
  std::unordered_set<uint64_t> deleted_;
};

// ------------------------------------------------------------------------------------------
// HNSWIndex — src/hnsw/core.rs, src/hnsw/operations.rs
// ------------------------------------------------------------------------------------------
struct HNSWConfig {
  uint32_t max_connections = 16, max_connections_layer_0 = 32, ef_construction = 200;  // :37-46
  uint64_t seed = 0;
};

class HNSWIndex {
 public:
  HNSWIndex(fvdb_ctx* ctx, const HNSWConfig& cfg);
  ~HNSWIndex();
  const HNSWConfig& config() const { return cfg_; }
  uint64_t node_count() const { return n_registered_; }
  bool entry_point(uint64_t* id) const;
  uint32_t dimension() const { return dim_; }
  size_t assign_level();                                                           // :211
  int insert(uint64_t id, const float* v, uint32_t dim, int64_t forced_level);     // :226
  // batch_insert (src/hnsw/operations.rs:74-94): the reference's sequential loop, keeps going after a failure.
  // levels[i] < 0 (or levels == nullptr): drawn with assign_level() in order, like the loop would.  The inserts that
  // pass the reference's checks are linked on the device in one call (fvdb_graph_insert_linked), strictly in order.
  int batch_insert(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const int64_t* levels, uint64_t* n_ok,
                   int* first_error);
  // Where an insert's searches, links and prunes run: true (default) = on the device against the adjacency in HBM
  // (kernels_graph_build.h); false = the host algorithm with every distance batch scored on the GPU (fvdb_scorer_*).
  // Same graph either way.  mode: fvdb_graph_insert_linked's (0 choose, 1 one at a time, 2 speculate batches).
  void set_device_insert(bool on, int mode = 0) {
    device_insert_ = on;
    insert_mode_ = mode;
  }
  bool device_insert() const { return device_insert_; }
  const fvdb_graph_insert_stats& insert_stats() const { return insert_stats_; }  // sums since construction
  uint64_t host_path_inserts() const { return n_host_inserts_; }
  uint64_t graph_upload_bytes() const { return graph_ ? fvdb_graph_upload_bytes(graph_) : 0; }
  int search(const float* q, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids, float* dist,
             uint32_t* counts);                                                    // :398 (batched, lock-step hops)
  int search_dev(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids, float* dist,
                 uint32_t* counts);  // queries resident in HBM, results to host
  // Split form for callers that overlap other GPU work with the graph walk: begin enqueues the
  // device-resident traversal (returns false when this search has to use the host walk instead, in which
  // case nothing was enqueued), end waits for it and delivers the results.
  // Device traversal split in two so that several batches can be in flight: begin enqueues the launch and the
  // result copies on the slot's own stream (slot < kSlots), end waits for that slot and finishes on the host.
  static constexpr uint32_t kSlots = 16;
  bool search_dev_begin(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, int* rc, uint32_t slot = 0);
  int search_dev_end(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids, float* dist,
                     uint32_t* counts, uint32_t slot = 0);
  int mark_deleted(uint64_t id);                                                   // operations.rs:127
  bool is_deleted(uint64_t id) const;
  uint64_t active_count() const;
  uint64_t vacuum();                                                               // operations.rs:176
  int64_t level_of(uint64_t id) const;
  int64_t neighbors(uint64_t id, uint32_t layer, uint64_t* out, uint64_t cap);
  const float* vector_of(uint64_t id) const;  // host copy (migration, get_vector_by_id)
  bool contains(uint64_t id) const { return index_of_.count(id) > 0; }
  // Extension (not in the reference): build the graph for n vectors at once.  Levels from the PRNG
  // (or given); per layer every member links to its exact nearest M (M0 on layer 0) members, found
  // with the GPU flat scan — the graph nearest-M selection (:556-558, :588-624) converges to.
  int bulk_build(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const int64_t* levels);
  // Install / export a graph (identical structures for parity runs).
  int restore(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const uint32_t* levels,
              const uint64_t* nbr_offsets, const uint64_t* nbrs, uint64_t entry_id);
  uint64_t graph_slots() const;
  uint64_t graph_edges();
  void export_graph(uint64_t* ids, uint32_t* levels, uint64_t* nbr_offsets, uint64_t* nbrs);
  uint64_t dist_evals() const { return n_dist_; }
  uint64_t hops() const { return n_hops_; }
  void set_threads(int t) { threads_ = t; }
  // Where the layered walk runs for batch searches: false = on the host, every hop's candidates scored
  // by one GPU launch (fvdb_scorer_*); true (default) = entirely on the GPU (fvdb_graph_search_dev), the
  // graph mirrored in HBM.  Same results; inserts always use the host walk.
  void set_device_traversal(bool on) { device_traversal_ = on; }
  bool device_traversal() const { return device_traversal_; }
  uint64_t device_fallbacks() const { return n_fallback_.load(); }
  // profiling on: summed HIP-event duration of the traversal kernel's launches since the last call
  int graph_kernel_times(float* ms_sum, uint32_t* launches, uint64_t* rows_scored, uint64_t* hops) {
    *ms_sum = 0.0f;
    *launches = 0;
    *rows_scored = *hops = 0;
    return graph_ ? fvdb_graph_kernel_times(graph_, ms_sum, launches, rows_scored, hops) : 0;
  }
  // queries served by the traversal kernel / searched a second time with the restated heaps (equal distances)
  int tie_restarts(uint64_t* queries, uint64_t* again) {
    *queries = *again = 0;
    return graph_ ? fvdb_graph_tie_restarts(graph_, queries, again) : 0;
  }

 private:
  struct Query {
    RustHeap candidates, nearest;
    Visited visited;
    std::vector<uint32_t> pending;  // candidates sent to the GPU this hop
    bool active = false;
  };
  uint32_t cap(uint32_t layer) const { return layer == 0 ? cfg_.max_connections_layer_0 : cfg_.max_connections; }
  std::vector<uint32_t>& nb(uint32_t node, uint32_t layer) { return nbrs_[node][layer]; }
  int ensure_store(uint32_t dim);
  int ensure_scorer(uint32_t B, uint32_t C);
  int append_row(const float* v, uint32_t* row);
  // lock-step search_layer (:469-554) for B queries already loaded into the scorer
  int search_layer_batch(uint32_t B, const std::vector<Cand>& entries, const std::vector<uint8_t>& has_entry,
                         uint32_t ef, uint32_t layer, std::vector<std::vector<Cand>>& results);
  int score_pairs_from_row(uint32_t base_row, const std::vector<uint32_t>& cands, std::vector<float>& out);
  int search_impl(const float* q, bool q_on_device, uint32_t B, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids,
                  float* dist, uint32_t* counts);
  // Pipelined batch search: queries are split into a few lanes (one scorer = one HIP stream each) driven
  // round-robin, so that while the GPU scores one lane's hop the host applies and prepares another's.
  // Per query the operation sequence is still search_layer's.
  struct Lane {
    fvdb_scorer* sc = nullptr;
    uint32_t cap_B = 0, cap_C = 0;
    std::vector<Query> qs;
    std::vector<uint32_t> prev_cnt;
    std::vector<std::vector<Cand>> cur;
    uint32_t lo = 0, n = 0, layer = 0, ef = 0, k = 0;
    int stage = 0;  // 0 start, 1 entry scored, 2 hop in flight
    int threads = 1;  // OpenMP threads for this lane's host phases
    bool done = true;
    int rc = 0;
    uint64_t dists = 0, hops = 0;
  };
  int lane_ensure(Lane& ln, uint32_t B, uint32_t C);
  void lane_layer_init(Lane& ln);
  uint32_t lane_hop_prepare(Lane& ln);
  void lane_hop_apply(Lane& ln);
  void lane_layer_collect(Lane& ln);
  void lane_advance(Lane& ln, const float* q, bool q_on_device, uint32_t ef_final, uint64_t* ids, float* dist,
                    uint32_t* counts);
  std::vector<Lane> lanes_;
  int sync_graph();
  bool device_path_ok(uint32_t ef) const;
  int device_launch(const float* q_dev, uint32_t B, uint32_t k, uint32_t ef, uint32_t slot);
  int device_collect(uint32_t B, uint32_t k, uint64_t* ids, float* dist, uint32_t* counts, std::vector<uint32_t>& failed,
                     uint32_t slot);
  int finish_failed(const float* q, bool q_on_device, uint32_t dim, uint32_t k, uint32_t ef, uint64_t* ids, float* dist,
                    uint32_t* counts, const std::vector<uint32_t>& failed);
  int search_host_walk(const float* q, bool q_on_device, uint32_t B, uint32_t k, uint32_t ef, uint64_t* ids,
                       float* dist, uint32_t* counts);
  // The adjacency lists exist twice: nbrs_ (host) and the fixed-stride rows in HBM (graph_).  host_ahead_: nbrs_ holds
  // changes the device has not seen (restore, bulk build, vacuum, inserts made before the device graph existed) — the
  // next device use installs the whole graph once.  dev_ahead_: device inserts have linked nodes whose lists nbrs_
  // does not hold yet — whoever needs nbrs_ (export, neighbours, host walk, vacuum, a host-path insert) pulls them.
  // A host-path insert made while the two agree patches the rows it changed (fvdb_graph_set_lists): no whole-graph
  // upload follows an insert or a delete.
  fvdb_graph* graph_ = nullptr;
  bool host_ahead_ = true, dev_ahead_ = false, device_traversal_ = true, device_insert_ = true;
  int insert_mode_ = 0;
  fvdb_graph_insert_stats insert_stats_{};
  uint64_t n_host_inserts_ = 0;
  int ensure_graph_handle();
  int ensure_host_graph();
  bool device_insert_ok() const;
  int check_insert(uint64_t id, const float* v, uint32_t dim) const;
  int insert_host(uint64_t id, const float* v, uint32_t dim, int64_t forced_level);
  // the host algorithm's links for node `row` (bookkeeping and vector already in place); the (node, layer) lists it
  // changed are appended to `touched`
  int link_host(uint32_t row, std::vector<std::pair<uint32_t, uint32_t>>* touched);
  int push_lists(const std::vector<std::pair<uint32_t, uint32_t>>& touched);
  void finalize_insert(uint32_t row);
  struct DevSlot {  // per in-flight batch: stream (context), device result buffers, pinned host copies
    fvdb_ctx* ctx = nullptr;  // slot 0 borrows ctx_, the others own theirs
    void *d_nodes = nullptr, *d_dist = nullptr, *d_cnt = nullptr, *d_status = nullptr;
    void *h_nodes = nullptr, *h_dist = nullptr, *h_cnt = nullptr, *h_status = nullptr;
    uint64_t cap = 0;
  };
  DevSlot slots_[kSlots];
  void* d_q_ = nullptr;
  uint64_t d_q_cap_ = 0;
  std::atomic<uint64_t> n_fallback_{0};
  // Searches may come from several host threads (HybridIndex leases a slot per call): the graph mirror is synced by
  // one of them, the rare host-walk fallback and the standalone search()/search_dev() entry points are serialised.
  std::mutex sync_mu_, walk_mu_, search_mu_;

  fvdb_ctx* ctx_;
  HNSWConfig cfg_;
  SplitMix64 rng_;
  fvdb_store* store_ = nullptr;
  fvdb_scorer* scorer_ = nullptr;
  uint32_t scorer_B_ = 0, scorer_C_ = 0;
  uint32_t dim_ = 0;
  bool has_dim_ = false, has_entry_ = false, entry_lost_ = false;
  uint32_t entry_ = 0;
  uint64_t n_registered_ = 0;
  std::vector<uint64_t> ids_;
  std::vector<uint32_t> level_;
  std::vector<uint8_t> deleted_, registered_;
  std::vector<std::vector<std::vector<uint32_t>>> nbrs_;  // [node][layer] insertion-ordered sets
  std::unordered_map<uint64_t, uint32_t> index_of_;
  std::vector<float> host_vecs_;
  std::vector<Query> qs_;
  uint64_t n_dist_ = 0, n_hops_ = 0;
  double t_prepare_us_ = 0, t_gpu_us_ = 0, t_apply_us_ = 0;
  int threads_ = 0;
};

// ------------------------------------------------------------------------------------------
// HybridIndex — src/hybrid/core.rs.  `now` / timestamps are seconds passed in by the caller
// (the reference reads Utc::now() at the same points).
// ------------------------------------------------------------------------------------------
struct HybridConfig {
  double recent_threshold_s = 7.0 * 24 * 3600;  // :77
  HNSWConfig hnsw;
  IVFConfig ivf;  // default() below sets 3 clusters / n_probe 2 / train_size 9 like :69-74
  uint64_t migration_batch_size = 100;
  bool auto_migrate = true;
  uint64_t min_ivf_training_size = 10;
  static HybridConfig defaults() {
    HybridConfig c;
    c.ivf.train_size = 9;
    c.ivf.n_clusters = 3;
    c.ivf.n_probe = 2;
    return c;
  }
};

struct HybridSearchConfig {  // :172-197
  bool search_recent = true, search_historical = true;
  uint64_t recent_k = 0, historical_k = 0;
  uint64_t k = 10, hnsw_ef = 50, ivf_n_probe = 10;
};

class HybridIndex {
 public:
  HybridIndex(fvdb_ctx* ctx_ivf, fvdb_ctx* ctx_hnsw, const HybridConfig& cfg);
  ~HybridIndex();
  bool is_initialized() const { return initialized_; }
  bool is_ivf_trained() const { return ivf_trained_; }
  int initialize(const float* data, uint64_t n, uint32_t dim);                                    // :262
  int set_ivf_centroids(const float* c, uint32_t dim);  // install a trained quantizer (parity runs)
  int insert_with_timestamp(uint64_t id, const float* v, uint32_t dim, double ts, double now, int64_t level);  // :357
  int search(const float* q, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg, double now, uint64_t* ids,
             float* dist, uint32_t* counts);                                                       // :425
  // queries resident in HBM: the IVF scan runs asynchronously on its own stream while the host walks
  // the HNSW graph (hop scoring on the second stream); results come back to host memory
  int search_dev(const float* q_dev, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg, double now,
                 uint64_t* ids, float* dist, uint32_t* counts);
  // The same search split in two, so that `kSlots` batches can be in flight at once (the graph walk of one
  // batch occupies one wavefront per SIMD: a second batch's walk runs beside it almost for free).  begin
  // enqueues everything for the batch (traversal kernel on the slot's stream, IVF chain and its result copies on
  // the IVF stream) and returns; end waits for that slot only and merges on the host.  The query buffer must stay
  // valid and unchanged until end.  Mutations between a begin and its end are not allowed.
  static constexpr uint32_t kSlots = HNSWIndex::kSlots;
  int search_dev_begin(uint32_t slot, const float* q_dev, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg,
                       double now);
  int search_dev_end(uint32_t slot, uint64_t* ids, float* dist, uint32_t* counts);
  // Multi-GPU (no counterpart in the reference): this rank's HybridIndex holds the inverted lists it owns
  // (bulk_insert_sharded) and a replica of the graph.  attach_comm binds a communicator (fvdb_comm_create / _hosted);
  // search_sharded_begin enqueues one step in `slot` — the IVF part through fvdb_ivf_search_sharded_begin (both
  // exchanges and the merge by key on the slot's stream), the graph walk of this rank's own queries beside it — and
  // search_sharded_end waits for the slot and applies the reference's hybrid merge.  mode FVDB_SHARD_WEAK: q_dev is
  // this rank's own B queries, B result rows; FVDB_SHARD_STRONG: q_dev is the global batch (same on every rank),
  // result rows = this rank's slice [r*per, min(B,(r+1)*per)) (sharded_rows()).  Every rank makes the same sequence
  // of begin/end calls WITH THE SAME `now`: the per-search auto-migration (src/hybrid/core.rs:437-439) runs on every
  // rank — the same due ids in the same order, assigned by the same centroids; the rank that owns a list appends the
  // row, every rank counts it into the logical list sizes (migrate_locked).
  int attach_comm(fvdb_comm* comm);
  int search_sharded_begin(uint32_t slot, const float* q_dev, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg,
                           int mode, double now = 0.0);
  int search_sharded_end(uint32_t slot, uint64_t* ids, float* dist, uint32_t* counts) {
    return search_dev_end(slot, ids, dist, counts);
  }
  uint32_t sharded_rows(uint32_t B, int mode) const;  // rows search_sharded_end writes on this rank
  // search_with_filter (src/hybrid/core.rs:513-549): ask for 3 k neighbours, keep the first k whose id the host
  // application's metadata filter accepts.  `matches(id, user)` stands for `metadata_map.get(id)` +
  // `MetadataFilter::matches` (an id without metadata does not match); NULL = no filter = plain search.
  typedef int (*FilterFn)(uint64_t id, void* user);
  int search_with_filter(const float* q, uint32_t B, uint32_t dim, uint64_t k, FilterFn matches, void* user, double now,
                         uint64_t* ids, float* dist, uint32_t* counts);
  // Concurrency (reference: tokio RwLock, searches = readers, src/hybrid/core.rs:457,466): search() / search_dev() /
  // search_with_filter() may be called from any number of host threads; each call leases a free slot (stream,
  // traversal state, IVF scratch set, result blocks) and returns it.  Mutations wait for those calls to finish.
  // The explicit search_dev_begin/_end pair is for ONE thread keeping several batches in flight; while such a batch
  // is uncollected, mutations are refused (INVALID) rather than waited for.
  // What a mutation (insert, delete, migrate, vacuum) does while batches begun with search_dev_begin are uncollected:
  // false (default) = returns FVDB_E_INVALID at once; true = waits for them to be collected, like the reference's write
  // guard waits for its readers (src/hybrid/core.rs:457,466) — for hosts whose searches and writes run on different
  // threads (a thread that waits for its OWN uncollected batch would wait for ever).
  void set_blocking_writers(bool on) { writers_wait_ = on; }
  bool blocking_writers() const { return writers_wait_; }
  bool busy() const {  // some batch is in flight: inserts / deletes are refused (or wait) until it is collected
    std::lock_guard<std::mutex> lk(slot_mu_);
    for (const Slot& s : slots_)
      if (s.active) return true;
    return false;
  }
  bool migration_due(double threshold_s, double now) const {
    return !pending_migration_.empty() && age_of(now, pending_min_ts_) >= threshold_s;
  }
  uint64_t migrate_with_threshold(double threshold_s, double now);                                // :600
  int remove(uint64_t id, double now);                                                             // delete :904
  uint64_t recent_count() const { return recent_count_; }
  uint64_t historical_count() const { return historical_count_; }
  // from_parts (src/hybrid/core.rs:857-877): adopt the two indexes as they now are (the caller has restored the
  // graph into recent() and set_trained + inserted the lists of historical()) together with the saved timestamp
  // table and counters; the index becomes initialised.  Only valid while the timestamp table is empty.
  int from_parts(const uint64_t* ids, const double* ts, uint64_t n, uint64_t recent_count, uint64_t historical_count,
                 bool ivf_trained);
  // vacuum (src/hybrid/core.rs:989-1012): both indexes; refused while a batch is in flight
  int vacuum(uint64_t* hnsw_removed, uint64_t* ivf_removed);
  uint64_t timestamp_count() const { return ts_order_.size(); }
  void export_timestamps(uint64_t* ids, double* ts) const;  // insertion order
  HNSWIndex& recent() { return *recent_; }
  IVFIndex& historical() { return *historical_; }
  // How the bulk loaders build the graph of the recent part: true (default) = the reference's own build, sequential
  // inserts in id order (HNSWIndex::batch_insert, on the device); false = HNSWIndex::bulk_build (exact nearest-M
  // per layer — a different graph, an extension).
  void set_sequential_graph(bool on) { sequential_graph_ = on; }
  bool sequential_graph() const { return sequential_graph_; }
  double recent_build_seconds() const { return recent_build_s_; }  // wall time of the last bulk load's graph build
  // bulk loaders for scale runs: route by age like insert_with_timestamp, batched on the GPU
  int bulk_insert(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const double* ts, double now);
  // Multi-GPU placement: the same, but of the historical rows only those whose IVF list is owned by
  // `rank` (owner[list] == rank) are stored on this GPU; the logical list sizes are installed so the
  // tie-break position is global.  With owner == nullptr, lists are dealt largest-first to the least
  // loaded of `world` ranks (deterministic, identical on every rank).  The HNSW part is replicated.
  int bulk_insert_sharded(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, const double* ts, double now,
                          uint32_t rank, uint32_t world, uint32_t* owner_out /* nlist, optional */);

 private:
  static double age_of(double now, double ts) { return now - ts < 0 ? 0.0 : now - ts; }
  int search_impl(const float* q, bool q_on_device, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg,
                  double now, uint64_t* ids, float* dist, uint32_t* counts);
  int begin_impl(uint32_t slot, const float* q_dev, uint32_t B, uint32_t dim, const HybridSearchConfig& cfg,
                 int shard_mode = -1);
  int build_recent(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim);
  bool sequential_graph_ = true;
  std::atomic<bool> writers_wait_{false};
  // exclusive access for a mutation: with no batch in flight, or FVDB_E_INVALID / after waiting (set_blocking_writers)
  int write_lock(std::unique_lock<std::shared_mutex>& w);
  double recent_build_s_ = 0.0;
  fvdb_comm* comm_ = nullptr;
  fvdb_sharded* sharded_ = nullptr;
  // list placement of bulk_insert_sharded (world 0 = not sharded): owner rank and logical size of every list
  uint32_t shard_rank_ = 0, shard_world_ = 0;
  std::vector<uint32_t> shard_owner_;
  std::vector<uint64_t> shard_sizes_;
  uint64_t migrate_locked(double threshold_s, double now);
  bool busy_unlocked() const {
    for (const Slot& s : slots_)
      if (s.active) return true;
    return false;
  }
  mutable std::shared_mutex rw_;   // searches of the blocking entry points: shared; mutations: unique
  mutable std::mutex slot_mu_;     // Slot::active
  std::condition_variable slot_cv_;
  fvdb_ctx* ctx_ivf_;
  struct Slot {  // one batch in flight
    void *d_hid = nullptr, *d_hd = nullptr, *d_hc = nullptr;  // device result buffers of the IVF part
    void *h_hid = nullptr, *h_hd = nullptr, *h_hc = nullptr;  // pinned host copies
    uint64_t cap = 0;
    fvdb_event* ivf_done = nullptr;
    fvdb_ctx* ivf_ctx = nullptr;  // slot 0 borrows ctx_ivf_, the others own a context (stream) each
    bool active = false, ivf_in_flight = false, hnsw_in_flight = false, recent = false;
    void* d_q = nullptr;      // staging for host-resident query batches of the blocking entry points
    uint64_t d_q_cap = 0;
    const float* q = nullptr;
    uint32_t B = 0, dim = 0, k = 0, rk = 0, hk = 0, ef = 0;
  };
  Slot slots_[kSlots];
  HybridConfig cfg_;
  HNSWIndex* recent_;
  IVFIndex* historical_;
  bool initialized_ = false, ivf_trained_ = false;
  std::vector<uint64_t> ts_order_;
  std::unordered_map<uint64_t, double> timestamps_;
  struct Pending {
    uint64_t id;
    double ts;
  };
  std::vector<Pending> pending_migration_;  // (insertion order) ids living in HNSW not yet copied into IVF
  double pending_min_ts_ = 1e300;           // oldest timestamp among them: O(1) "nothing is due" test
  uint64_t recent_count_ = 0, historical_count_ = 0;
};

// Pure host pieces of the multi-GPU path (also called by the CPU tests through the C wrappers):
// greedy "largest list to the least loaded rank" placement, identical on every rank ...
void plan_list_owners_host(const uint64_t* sizes, uint32_t nlist, uint32_t world, uint32_t* owner);
// ... and HybridIndex::search_with_config's merge (src/hybrid/core.rs:476-485): recent results then historical ones,
// stable sort by distance, truncate(k); B queries, rid/rd are B x rk, hid/hd B x hk (either part may be NULL)
void merge_parts_host(uint32_t B, uint32_t k, uint32_t rk, uint32_t hk, const uint64_t* rid, const float* rd,
                      const uint32_t* rc, const uint64_t* hid, const float* hd, const uint32_t* hc, uint64_t* ids,
                      float* dist, uint32_t* counts);

}  // namespace fvdbh
