// ivf_index.cpp — IVFIndex mirror (src/ivf/core.rs, src/ivf/operations.rs) over the C ABI.
#include <algorithm>
#include <cstring>

#include "fvdb_host.hpp"

namespace fvdbh {

IVFIndex::IVFIndex(fvdb_ctx* ctx, const IVFConfig& cfg) : ctx_(ctx), cfg_(cfg) {}

IVFIndex::~IVFIndex() {
  if (dev_) fvdb_ivf_destroy(dev_);
}

int IVFIndex::ensure_device(uint32_t dim) {
  if (dev_ && dim_ == dim) return FVDB_OK;
  if (dev_) {
    fvdb_ivf_destroy(dev_);
    dev_ = nullptr;
  }
  int rc = fvdb_ivf_create(ctx_, dim, cfg_.n_clusters, &dev_);
  if (rc) return rc;
  dim_ = dim;
  return FVDB_OK;
}

// src/ivf/core.rs:240-334
int IVFIndex::train(const float* data, uint64_t n, uint32_t dim, fvdb_train_result* out) {
  if (n == 0 || n < cfg_.n_clusters) return FVDB_E_INSUFFICIENT;
  int rc = ensure_device(dim);
  if (rc) return rc;
  rc = fvdb_ivf_train(dev_, data, n, cfg_.max_iterations, cfg_.seed, out);
  if (rc) return rc;
  trained_ = true;
  where_.clear();
  deleted_.clear();
  total_ = 0;
  return FVDB_OK;
}

// src/ivf/core.rs:509-520
int IVFIndex::set_trained(const float* centroids, uint32_t dim) {
  int rc = ensure_device(dim);
  if (rc) return rc;
  rc = fvdb_ivf_set_centroids(dev_, centroids);
  if (rc) return rc;
  trained_ = true;
  where_.clear();
  deleted_.clear();
  total_ = 0;
  return FVDB_OK;
}

int IVFIndex::get_centroids(float* out) const {
  if (!trained_) return FVDB_E_NOT_TRAINED;
  return fvdb_ivf_get_centroids(dev_, out);
}

void IVFIndex::clear_lists() {
  if (dev_) fvdb_ivf_clear(dev_);
  where_.clear();
  deleted_.clear();
  total_ = 0;
}

// src/ivf/operations.rs:625-645: every row of a soft-deleted id leaves its list, the survivors keep their order.
// The lists live in HBM as packed 64-row blocks: copy the survivors out, empty the lists, append them again.
int IVFIndex::vacuum(uint64_t* removed_out) {
  const uint64_t removed = deleted_.size();
  if (removed_out) *removed_out = removed;
  if (removed == 0) return FVDB_OK;
  if (dev_) {
    std::vector<uint64_t> sizes(cfg_.n_clusters);
    int rc = fvdb_ivf_list_sizes(dev_, sizes.data());
    if (rc) return rc;
    std::vector<float> xv, rows;
    std::vector<uint64_t> xi, ids;
    std::vector<uint32_t> xc;
    for (uint32_t c = 0; c < cfg_.n_clusters; ++c) {
      if (sizes[c] == 0) continue;
      rows.resize(sizes[c] * (size_t)dim_);
      ids.resize(sizes[c]);
      rc = fvdb_ivf_list_export(dev_, c, rows.data(), ids.data(), nullptr);
      if (rc) return rc;
      for (uint64_t i = 0; i < sizes[c]; ++i) {
        if (deleted_.count(ids[i])) continue;
        xi.push_back(ids[i]);
        xc.push_back(c);
        xv.insert(xv.end(), rows.begin() + i * dim_, rows.begin() + (i + 1) * dim_);
      }
    }
    rc = fvdb_ivf_clear(dev_);
    if (rc) return rc;
    where_.clear();
    const uint64_t total_before = total_;
    total_ = 0;
    if (!xi.empty()) {
      uint64_t ok = 0;
      int err = 0;
      rc = place(xi.data(), xv.data(), xi.size(), xc.data(), &ok, &err);
      if (rc) return rc;
    }
    total_ = total_before - removed;  // "total_vectors -= removed_count" (:638), counted in ids like the reference
  } else {
    total_ -= removed;
  }
  deleted_.clear();
  return FVDB_OK;
}

int IVFIndex::export_list(uint32_t c, float* rows, uint64_t* ids, uint8_t* live) const {
  if (c >= cfg_.n_clusters) return FVDB_E_INVALID;
  if (!dev_) return FVDB_OK;  // nothing stored yet
  return fvdb_ivf_list_export(dev_, c, rows, ids, live);
}

uint64_t IVFIndex::cluster_size(uint32_t c) const {
  if (!dev_ || c >= cfg_.n_clusters) return 0;
  std::vector<uint64_t> sizes(cfg_.n_clusters);
  fvdb_ivf_list_sizes(dev_, sizes.data());
  return sizes[c];
}

// src/ivf/core.rs:493-499
int IVFIndex::find_cluster(const float* v, uint32_t dim, uint32_t* out) {
  if (!trained_) return FVDB_E_NOT_TRAINED;
  if (dim != dim_) return FVDB_E_DIM;
  return fvdb_ivf_assign(dev_, v, 1, out);
}

// rows whose cluster is known: per-list duplicate check (InvertedList::insert :128-134), then append
int IVFIndex::place(const uint64_t* ids, const float* v, uint64_t n, const uint32_t* clusters, uint64_t* n_ok,
                    int* first_error) {
  std::vector<uint64_t> keep;
  keep.reserve(n);
  std::unordered_multimap<uint64_t, uint32_t> batch_seen;  // (id -> cluster) accepted earlier in this batch
  for (uint64_t i = 0; i < n; ++i) {
    bool dup = false;
    auto r = where_.equal_range(ids[i]);
    for (auto it = r.first; it != r.second && !dup; ++it) dup = it->second.cluster == clusters[i];
    auto r2 = batch_seen.equal_range(ids[i]);
    for (auto it = r2.first; it != r2.second && !dup; ++it) dup = it->second == clusters[i];
    if (dup) {
      if (first_error && *first_error == 0) *first_error = FVDB_E_DUPLICATE;
      continue;
    }
    batch_seen.emplace(ids[i], clusters[i]);
    keep.push_back(i);
  }
  if (n_ok) *n_ok = keep.size();
  if (keep.empty()) return FVDB_OK;
  std::vector<uint32_t> pos(keep.size());
  int rc;
  if (keep.size() == n) {
    rc = fvdb_ivf_add_assigned(dev_, v, ids, n, clusters, pos.data());
  } else {
    std::vector<float> xv(keep.size() * (size_t)dim_);
    std::vector<uint64_t> xi(keep.size());
    std::vector<uint32_t> xc(keep.size());
    for (size_t j = 0; j < keep.size(); ++j) {
      std::memcpy(&xv[j * dim_], v + keep[j] * dim_, dim_ * sizeof(float));
      xi[j] = ids[keep[j]];
      xc[j] = clusters[keep[j]];
    }
    rc = fvdb_ivf_add_assigned(dev_, xv.data(), xi.data(), keep.size(), xc.data(), pos.data());
  }
  if (rc) return rc;
  for (size_t j = 0; j < keep.size(); ++j) where_.emplace(ids[keep[j]], Loc{clusters[keep[j]], pos[j]});
  total_ += keep.size();
  return FVDB_OK;
}

// src/ivf/core.rs:431-455
int IVFIndex::insert(uint64_t id, const float* v, uint32_t dim) {
  if (!trained_) return FVDB_E_NOT_TRAINED;
  if (dim != dim_) return FVDB_E_DIM;
  uint32_t c = 0;
  int rc = fvdb_ivf_assign(dev_, v, 1, &c);
  if (rc) return rc;
  uint64_t ok = 0;
  int err = 0;
  rc = place(&id, v, 1, &c, &ok, &err);
  if (rc) return rc;
  return ok == 1 ? FVDB_OK : err;
}

// src/ivf/operations.rs:107-130 — per-row outcome like a loop of insert(); one GPU assignment pass
int IVFIndex::batch_insert(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim, uint64_t* n_ok,
                           int* first_error) {
  if (n_ok) *n_ok = 0;
  if (first_error) *first_error = 0;
  if (!trained_) return FVDB_E_NOT_TRAINED;
  if (dim != dim_) return FVDB_E_DIM;
  if (n == 0) return FVDB_OK;
  std::vector<uint32_t> clusters(n);
  int rc = fvdb_ivf_assign(dev_, v, n, clusters.data());
  if (rc) return rc;
  return place(ids, v, n, clusters.data(), n_ok, first_error);
}

int IVFIndex::batch_insert_assigned(const uint64_t* ids, const float* v, uint64_t n, uint32_t dim,
                                    const uint32_t* clusters, uint64_t* n_ok, int* first_error) {
  if (n_ok) *n_ok = 0;
  if (first_error) *first_error = 0;
  if (!trained_) return FVDB_E_NOT_TRAINED;
  if (dim != dim_) return FVDB_E_DIM;
  if (n == 0) return FVDB_OK;
  return place(ids, v, n, clusters, n_ok, first_error);
}

int IVFIndex::assign(const float* v, uint64_t n, uint32_t dim, uint32_t* out) {
  if (!trained_) return FVDB_E_NOT_TRAINED;
  if (dim != dim_) return FVDB_E_DIM;
  return fvdb_ivf_assign(dev_, v, n, out);
}

// src/ivf/core.rs:626-681 for a batch (src/ivf/operations.rs:132-145)
int IVFIndex::search(const float* q, uint32_t B, uint32_t dim, uint32_t k, uint32_t n_probe, uint64_t* ids,
                     float* dist, uint32_t* counts) {
  if (!trained_) return FVDB_E_NOT_TRAINED;
  if (dim != dim_) return FVDB_E_DIM;
  return fvdb_ivf_search(dev_, q, B, k, n_probe, ids, dist, counts);
}

int IVFIndex::search_dev(const float* q_dev, uint32_t B, uint32_t dim, uint32_t k, uint32_t n_probe, uint64_t* ids_dev,
                         float* dist_dev, uint32_t* counts_dev, fvdb_ctx* on, uint32_t slot) {
  if (!trained_) return FVDB_E_NOT_TRAINED;
  if (dim != dim_) return FVDB_E_DIM;
  return fvdb_ivf_search_dev_slot(dev_, on, slot, q_dev, B, k, n_probe, ids_dev, dist_dev, counts_dev, nullptr);
}

// src/ivf/operations.rs:569-591
int IVFIndex::mark_deleted(uint64_t id) {
  auto r = where_.equal_range(id);
  if (r.first == r.second) return FVDB_E_NOT_FOUND;
  deleted_.insert(id);
  std::vector<uint32_t> cl, ps;
  for (auto it = r.first; it != r.second; ++it) {
    cl.push_back(it->second.cluster);
    ps.push_back(it->second.pos);
  }
  return fvdb_ivf_set_deleted(dev_, cl.data(), ps.data(), cl.size(), 1);
}

}  // namespace fvdbh
