// capi.cpp — flat C view of the host mirror for ctypes (fvh_* symbols).
#include <cstring>
#include <new>

#include "fvdb_host.hpp"

using namespace fvdbh;

extern "C" {

// ---- IVFIndex ----
void* fvh_ivf_new(fvdb_ctx* ctx, uint32_t n_clusters, uint32_t n_probe, uint32_t train_size, uint32_t max_iterations,
                  uint64_t seed) {
  IVFConfig c;
  c.n_clusters = n_clusters;
  c.n_probe = n_probe;
  c.train_size = train_size;
  c.max_iterations = max_iterations;
  c.seed = seed;
  if (!c.is_valid()) return nullptr;  // IVFIndex::new panics on an invalid config (:171-174)
  return new (std::nothrow) IVFIndex(ctx, c);
}
void fvh_ivf_free(void* p) { delete (IVFIndex*)p; }
int fvh_ivf_train(void* p, const float* x, uint64_t n, uint32_t d, fvdb_train_result* out) {
  return ((IVFIndex*)p)->train(x, n, d, out);
}
int fvh_ivf_set_trained(void* p, const float* c, uint32_t d) { return ((IVFIndex*)p)->set_trained(c, d); }
int fvh_ivf_get_centroids(void* p, float* out) { return ((IVFIndex*)p)->get_centroids(out); }
int fvh_ivf_is_trained(void* p) { return ((IVFIndex*)p)->is_trained(); }
uint32_t fvh_ivf_dimension(void* p) { return ((IVFIndex*)p)->dimension(); }
uint64_t fvh_ivf_total_vectors(void* p) { return ((IVFIndex*)p)->total_vectors(); }
uint64_t fvh_ivf_active_count(void* p) { return ((IVFIndex*)p)->active_count(); }
uint64_t fvh_ivf_cluster_size(void* p, uint32_t c) { return ((IVFIndex*)p)->cluster_size(c); }
int fvh_ivf_export_list(void* p, uint32_t c, float* rows, uint64_t* ids, uint8_t* live) {
  return ((IVFIndex*)p)->export_list(c, rows, ids, live);
}
int fvh_ivf_insert(void* p, uint64_t id, const float* v, uint32_t d) { return ((IVFIndex*)p)->insert(id, v, d); }
int fvh_ivf_batch_insert(void* p, const uint64_t* ids, const float* v, uint64_t n, uint32_t d, uint64_t* n_ok,
                         int* first_error) {
  return ((IVFIndex*)p)->batch_insert(ids, v, n, d, n_ok, first_error);
}
int fvh_ivf_find_cluster(void* p, const float* v, uint32_t d, uint32_t* out) {
  return ((IVFIndex*)p)->find_cluster(v, d, out);
}
int fvh_ivf_search(void* p, const float* q, uint32_t B, uint32_t d, uint32_t k, uint32_t n_probe, uint64_t* ids,
                   float* dist, uint32_t* counts) {
  return ((IVFIndex*)p)->search(q, B, d, k, n_probe, ids, dist, counts);
}
int fvh_ivf_mark_deleted(void* p, uint64_t id) { return ((IVFIndex*)p)->mark_deleted(id); }
int fvh_ivf_is_deleted(void* p, uint64_t id) { return ((IVFIndex*)p)->is_deleted(id); }
void* fvh_ivf_device(void* p) { return ((IVFIndex*)p)->device(); }

// ---- HNSWIndex ----
void* fvh_hnsw_new(fvdb_ctx* ctx, uint32_t M, uint32_t M0, uint32_t efc, uint64_t seed) {
  HNSWConfig c;
  c.max_connections = M;
  c.max_connections_layer_0 = M0;
  c.ef_construction = efc;
  c.seed = seed;
  return new (std::nothrow) HNSWIndex(ctx, c);
}
void fvh_hnsw_free(void* p) { delete (HNSWIndex*)p; }
int fvh_hnsw_insert(void* p, uint64_t id, const float* v, uint32_t d, int64_t level) {
  return ((HNSWIndex*)p)->insert(id, v, d, level);
}
// batch_insert (src/hnsw/operations.rs:74-94): sequential loop, keeps going after a failure
int fvh_hnsw_batch_insert(void* p, const uint64_t* ids, const float* v, uint64_t n, uint32_t d, const int64_t* levels,
                          uint64_t* n_ok, int* first_error) {
  return ((HNSWIndex*)p)->batch_insert(ids, v, n, d, levels, n_ok, first_error);
}
void fvh_hnsw_set_device_insert(void* p, int on, int mode) { ((HNSWIndex*)p)->set_device_insert(on != 0, mode); }
int fvh_hnsw_device_insert(void* p) { return ((HNSWIndex*)p)->device_insert(); }
void fvh_hnsw_insert_stats(void* p, fvdb_graph_insert_stats* out, uint64_t* host_path_inserts, uint64_t* upload_bytes) {
  HNSWIndex* h = (HNSWIndex*)p;
  if (out) *out = h->insert_stats();
  if (host_path_inserts) *host_path_inserts = h->host_path_inserts();
  if (upload_bytes) *upload_bytes = h->graph_upload_bytes();
}
int fvh_hnsw_search(void* p, const float* q, uint32_t B, uint32_t d, uint32_t k, uint32_t ef, uint64_t* ids,
                    float* dist, uint32_t* counts) {
  return ((HNSWIndex*)p)->search(q, B, d, k, ef, ids, dist, counts);
}
uint64_t fvh_hnsw_node_count(void* p) { return ((HNSWIndex*)p)->node_count(); }
uint64_t fvh_hnsw_active_count(void* p) { return ((HNSWIndex*)p)->active_count(); }
int fvh_hnsw_entry_point(void* p, uint64_t* out) { return ((HNSWIndex*)p)->entry_point(out) ? 0 : FVDB_E_NOT_FOUND; }
int64_t fvh_hnsw_level(void* p, uint64_t id) { return ((HNSWIndex*)p)->level_of(id); }
int64_t fvh_hnsw_neighbors(void* p, uint64_t id, uint32_t layer, uint64_t* out, uint64_t cap) {
  return ((HNSWIndex*)p)->neighbors(id, layer, out, cap);
}
int fvh_hnsw_mark_deleted(void* p, uint64_t id) { return ((HNSWIndex*)p)->mark_deleted(id); }
int fvh_hnsw_is_deleted(void* p, uint64_t id) { return ((HNSWIndex*)p)->is_deleted(id); }
int fvh_hnsw_bulk_build(void* p, const uint64_t* ids, const float* v, uint64_t n, uint32_t d, const int64_t* levels) {
  return ((HNSWIndex*)p)->bulk_build(ids, v, n, d, levels);
}
int fvh_hnsw_restore(void* p, const uint64_t* ids, const float* v, uint64_t n, uint32_t d, const uint32_t* levels,
                     const uint64_t* off, const uint64_t* nbrs, uint64_t entry) {
  return ((HNSWIndex*)p)->restore(ids, v, n, d, levels, off, nbrs, entry);
}
uint64_t fvh_hnsw_graph_slots(void* p) { return ((HNSWIndex*)p)->graph_slots(); }
uint64_t fvh_hnsw_graph_edges(void* p) { return ((HNSWIndex*)p)->graph_edges(); }
void fvh_hnsw_export_graph(void* p, uint64_t* ids, uint32_t* levels, uint64_t* off, uint64_t* nbrs) {
  ((HNSWIndex*)p)->export_graph(ids, levels, off, nbrs);
}
int fvh_hnsw_get_vector(void* p, uint64_t id, float* out) {
  HNSWIndex* h = (HNSWIndex*)p;
  const float* v = h->vector_of(id);
  if (!v) return FVDB_E_NOT_FOUND;
  std::memcpy(out, v, h->dimension() * sizeof(float));
  return 0;
}
uint64_t fvh_hnsw_dist_evals(void* p) { return ((HNSWIndex*)p)->dist_evals(); }
uint64_t fvh_hnsw_hops(void* p) { return ((HNSWIndex*)p)->hops(); }
void fvh_hnsw_set_threads(void* p, int t) { ((HNSWIndex*)p)->set_threads(t); }
void fvh_hnsw_set_device_traversal(void* p, int on) { ((HNSWIndex*)p)->set_device_traversal(on != 0); }
int fvh_hnsw_device_traversal(void* p) { return ((HNSWIndex*)p)->device_traversal(); }
uint64_t fvh_hnsw_device_fallbacks(void* p) { return ((HNSWIndex*)p)->device_fallbacks(); }
int fvh_hnsw_tie_restarts(void* p, uint64_t* queries, uint64_t* again) { return ((HNSWIndex*)p)->tie_restarts(queries, again); }
int fvh_hnsw_graph_kernel_times(void* p, float* ms_sum, uint32_t* launches, uint64_t* rows_scored, uint64_t* hops) {
  return ((HNSWIndex*)p)->graph_kernel_times(ms_sum, launches, rows_scored, hops);
}
uint32_t fvh_hnsw_dimension(void* p) { return ((HNSWIndex*)p)->dimension(); }

// ---- HybridIndex ----
void* fvh_hybrid_new(fvdb_ctx* ctx_ivf, fvdb_ctx* ctx_hnsw, double recent_threshold_s, uint64_t migration_batch_size,
                     int auto_migrate, uint64_t min_ivf_training_size, uint32_t M, uint32_t M0, uint32_t efc,
                     uint64_t hseed, uint32_t n_clusters, uint32_t n_probe, uint32_t train_size, uint32_t max_iter,
                     uint64_t iseed) {
  HybridConfig c;
  c.recent_threshold_s = recent_threshold_s;
  c.migration_batch_size = migration_batch_size;
  c.auto_migrate = auto_migrate != 0;
  c.min_ivf_training_size = min_ivf_training_size;
  c.hnsw.max_connections = M;
  c.hnsw.max_connections_layer_0 = M0;
  c.hnsw.ef_construction = efc;
  c.hnsw.seed = hseed;
  c.ivf.n_clusters = n_clusters;
  c.ivf.n_probe = n_probe;
  c.ivf.train_size = train_size;
  c.ivf.max_iterations = max_iter;
  c.ivf.seed = iseed;
  if (!c.ivf.is_valid() || !(recent_threshold_s > 0) || migration_batch_size == 0) return nullptr;
  return new (std::nothrow) HybridIndex(ctx_ivf, ctx_hnsw, c);
}
void fvh_hybrid_free(void* p) { delete (HybridIndex*)p; }
int fvh_hybrid_initialize(void* p, const float* x, uint64_t n, uint32_t d) {
  return ((HybridIndex*)p)->initialize(x, n, d);
}
int fvh_hybrid_set_ivf_centroids(void* p, const float* c, uint32_t d) {
  return ((HybridIndex*)p)->set_ivf_centroids(c, d);
}
int fvh_hybrid_insert(void* p, uint64_t id, const float* v, uint32_t d, double ts, double now, int64_t level) {
  return ((HybridIndex*)p)->insert_with_timestamp(id, v, d, ts, now, level);
}
int fvh_hybrid_bulk_insert(void* p, const uint64_t* ids, const float* v, uint64_t n, uint32_t d, const double* ts,
                           double now) {
  return ((HybridIndex*)p)->bulk_insert(ids, v, n, d, ts, now);
}
int fvh_hybrid_bulk_insert_sharded(void* p, const uint64_t* ids, const float* v, uint64_t n, uint32_t d,
                                   const double* ts, double now, uint32_t rank, uint32_t world, uint32_t* owner_out) {
  return ((HybridIndex*)p)->bulk_insert_sharded(ids, v, n, d, ts, now, rank, world, owner_out);
}
int fvh_hybrid_search(void* p, const float* q, uint32_t B, uint32_t d, uint64_t k, uint64_t ef, uint64_t nprobe,
                      int search_recent, int search_historical, uint64_t recent_k, uint64_t historical_k, double now,
                      uint64_t* ids, float* dist, uint32_t* counts) {
  HybridSearchConfig c;
  c.k = k;
  c.hnsw_ef = ef;
  c.ivf_n_probe = nprobe;
  c.search_recent = search_recent != 0;
  c.search_historical = search_historical != 0;
  c.recent_k = recent_k;
  c.historical_k = historical_k;
  return ((HybridIndex*)p)->search(q, B, d, c, now, ids, dist, counts);
}
// search_with_filter (src/hybrid/core.rs:513-549): `matches(id, user)` is the host application's metadata lookup +
// MetadataFilter::matches (0 = no metadata or no match); NULL = no filter.
int fvh_hybrid_search_with_filter(void* p, const float* q, uint32_t B, uint32_t d, uint64_t k,
                                  int (*matches)(uint64_t, void*), void* user, double now, uint64_t* ids, float* dist,
                                  uint32_t* counts) {
  return ((HybridIndex*)p)->search_with_filter(q, B, d, k, matches, user, now, ids, dist, counts);
}
int fvh_hybrid_search_dev(void* p, const float* q_dev, uint32_t B, uint32_t d, uint64_t k, uint64_t ef, uint64_t nprobe,
                          int search_recent, int search_historical, uint64_t recent_k, uint64_t historical_k,
                          double now, uint64_t* ids, float* dist, uint32_t* counts) {
  HybridSearchConfig c;
  c.k = k;
  c.hnsw_ef = ef;
  c.ivf_n_probe = nprobe;
  c.search_recent = search_recent != 0;
  c.search_historical = search_historical != 0;
  c.recent_k = recent_k;
  c.historical_k = historical_k;
  return ((HybridIndex*)p)->search_dev(q_dev, B, d, c, now, ids, dist, counts);
}
int fvh_hybrid_search_dev_begin(void* p, uint32_t slot, const float* q_dev, uint32_t B, uint32_t d, uint64_t k, uint64_t ef,
                                uint64_t nprobe, int search_recent, int search_historical, uint64_t recent_k,
                                uint64_t historical_k, double now) {
  HybridSearchConfig c;
  c.k = k;
  c.hnsw_ef = ef;
  c.ivf_n_probe = nprobe;
  c.search_recent = search_recent != 0;
  c.search_historical = search_historical != 0;
  c.recent_k = recent_k;
  c.historical_k = historical_k;
  return ((HybridIndex*)p)->search_dev_begin(slot, q_dev, B, d, c, now);
}
// multi-GPU: see HybridIndex::attach_comm / search_sharded_begin in fvdb_host.hpp
int fvh_hybrid_attach_comm(void* p, fvdb_comm* comm) { return ((HybridIndex*)p)->attach_comm(comm); }
int fvh_hybrid_search_sharded_begin(void* p, uint32_t slot, const float* q_dev, uint32_t B, uint32_t d, uint64_t k,
                                    uint64_t ef, uint64_t nprobe, int search_recent, int search_historical,
                                    uint64_t recent_k, uint64_t historical_k, int mode, double now) {
  HybridSearchConfig c;
  c.k = k;
  c.hnsw_ef = ef;
  c.ivf_n_probe = nprobe;
  c.search_recent = search_recent != 0;
  c.search_historical = search_historical != 0;
  c.recent_k = recent_k;
  c.historical_k = historical_k;
  return ((HybridIndex*)p)->search_sharded_begin(slot, q_dev, B, d, c, mode, now);
}
int fvh_hybrid_search_sharded_end(void* p, uint32_t slot, uint64_t* ids, float* dist, uint32_t* counts) {
  return ((HybridIndex*)p)->search_sharded_end(slot, ids, dist, counts);
}
uint32_t fvh_hybrid_sharded_rows(void* p, uint32_t B, int mode) { return ((HybridIndex*)p)->sharded_rows(B, mode); }
// pure host logic of the multi-GPU path, callable without a GPU (CPU tests): list placement and the hybrid merge
void fvh_plan_list_owners(const uint64_t* sizes, uint32_t nlist, uint32_t world, uint32_t* owner) {
  fvdbh::plan_list_owners_host(sizes, nlist, world, owner);
}
void fvh_merge_parts(uint32_t B, uint32_t k, uint32_t rk, uint32_t hk, const uint64_t* rid, const float* rd,
                     const uint32_t* rc, const uint64_t* hid, const float* hd, const uint32_t* hc, uint64_t* ids,
                     float* dist, uint32_t* counts) {
  fvdbh::merge_parts_host(B, k, rk, hk, rid, rd, rc, hid, hd, hc, ids, dist, counts);
}
int fvh_hybrid_search_dev_end(void* p, uint32_t slot, uint64_t* ids, float* dist, uint32_t* counts) {
  return ((HybridIndex*)p)->search_dev_end(slot, ids, dist, counts);
}
// device traversal split in two (several batches in flight): begin returns 1 when the launch is in flight in `slot`,
// 0 when this search cannot run on the device path (call fvh_hnsw_search_dev instead of _end), < 0 on error
int fvh_hnsw_search_dev_begin(void* p, uint32_t slot, const float* q_dev, uint32_t B, uint32_t d, uint32_t k, uint32_t ef) {
  int rc = 0;
  const bool started = ((HNSWIndex*)p)->search_dev_begin(q_dev, B, d, k, ef, &rc, slot);
  if (rc) return rc < 0 ? rc : -rc;
  return started ? 1 : 0;
}
int fvh_hnsw_search_dev_end(void* p, uint32_t slot, const float* q_dev, uint32_t B, uint32_t d, uint32_t k, uint32_t ef,
                            uint64_t* ids, float* dist, uint32_t* counts) {
  return ((HNSWIndex*)p)->search_dev_end(q_dev, B, d, k, ef, ids, dist, counts, slot);
}
int fvh_hnsw_search_dev(void* p, const float* q_dev, uint32_t B, uint32_t d, uint32_t k, uint32_t ef, uint64_t* ids,
                        float* dist, uint32_t* counts) {
  return ((HNSWIndex*)p)->search_dev(q_dev, B, d, k, ef, ids, dist, counts);
}
int fvh_hybrid_delete(void* p, uint64_t id, double now) { return ((HybridIndex*)p)->remove(id, now); }
uint64_t fvh_hybrid_migrate(void* p, double thr, double now) {
  return ((HybridIndex*)p)->migrate_with_threshold(thr, now);
}
int fvh_hybrid_from_parts(void* p, const uint64_t* ids, const double* ts, uint64_t n, uint64_t recent_count,
                          uint64_t historical_count, int ivf_trained) {
  return ((HybridIndex*)p)->from_parts(ids, ts, n, recent_count, historical_count, ivf_trained != 0);
}
int fvh_hybrid_vacuum(void* p, uint64_t* hnsw_removed, uint64_t* ivf_removed) {
  return ((HybridIndex*)p)->vacuum(hnsw_removed, ivf_removed);
}
int fvh_ivf_vacuum(void* p, uint64_t* removed) { return ((IVFIndex*)p)->vacuum(removed); }
uint64_t fvh_hnsw_vacuum(void* p) { return ((HNSWIndex*)p)->vacuum(); }
uint64_t fvh_hybrid_timestamp_count(void* p) { return ((HybridIndex*)p)->timestamp_count(); }
void fvh_hybrid_export_timestamps(void* p, uint64_t* ids, double* ts) { ((HybridIndex*)p)->export_timestamps(ids, ts); }
uint64_t fvh_hybrid_recent_count(void* p) { return ((HybridIndex*)p)->recent_count(); }
uint64_t fvh_hybrid_historical_count(void* p) { return ((HybridIndex*)p)->historical_count(); }
int fvh_hybrid_is_initialized(void* p) { return ((HybridIndex*)p)->is_initialized(); }
int fvh_hybrid_is_ivf_trained(void* p) { return ((HybridIndex*)p)->is_ivf_trained(); }
void* fvh_hybrid_hnsw(void* p) { return &((HybridIndex*)p)->recent(); }
void fvh_hybrid_set_sequential_graph(void* p, int on) { ((HybridIndex*)p)->set_sequential_graph(on != 0); }
int fvh_hybrid_sequential_graph(void* p) { return ((HybridIndex*)p)->sequential_graph(); }
double fvh_hybrid_recent_build_seconds(void* p) { return ((HybridIndex*)p)->recent_build_seconds(); }
void fvh_hybrid_set_blocking_writers(void* p, int on) { ((HybridIndex*)p)->set_blocking_writers(on != 0); }
int fvh_hybrid_blocking_writers(void* p) { return ((HybridIndex*)p)->blocking_writers(); }
void* fvh_hybrid_ivf(void* p) { return &((HybridIndex*)p)->historical(); }

}  // extern "C"
