// oracle.cpp — CPU restatement of the reference's distance-computation hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.  The shipped library (libfvdb_hip.so and
// libfvdb_host.so) never links, loads or calls anything in this directory.
//
// Parity status: the reference (Fabstir/fabstir-vectordb, Rust) cannot be compiled in
// this environment (no cargo/rustc, no vendored crates), and it holds no
// numeric golden vectors for search results.  The oracle is therefore pinned by
//   (1) every analytic known-answer test the reference's own test-suite holds for this
//       path (tests/test_oracle_known_answers.py restates them one by one, citing
//       tests/core/*.rs, tests/ivf/core.rs, tests/hnsw/core.rs, tests/hybrid/core.rs), and
//   (2) line-by-line reading against the files cited below.
// RNG-dependent structures (k-means++ seeds, HNSW level draws) use rand 0.8 StdRng in
// the reference, whose source is absent and version unpinned => those draws are
// "parity unpinned"; every other function here is a literal restatement.
//
// Build: g++ -O2 -std=c++17 -fno-fast-math -ffp-contract=off -shared -fPIC (see Makefile).
// -ffp-contract=off keeps "sum + t*t" as a multiply and an add, like rustc's output for
//   a.iter().zip(b).map(|(x,y)| (x-y).powi(2)).sum::<f32>()      (src/core/vector_ops.rs:51-57)
//
// Containers: where the reference iterates a HashMap/HashSet (iteration order random per
// process: inverted-list vectors, HNSW neighbour sets, the timestamps map) the oracle
// iterates in INSERTION order.  That is one of the orders the reference can produce, and it
// only matters when two distances tie exactly.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------
// a1/a2: L2 distance — src/core/vector_ops.rs:51-57, src/hnsw/core.rs:691-697.
// Sequential left-to-right f32 fold, sqrt at the end.  powi(2) == x*x.
// ---------------------------------------------------------------------------------------
inline float l2(const float* a, const float* b, size_t d) {
  float sum = 0.0f;
  for (size_t i = 0; i < d; ++i) {
    float t = a[i] - b[i];
    sum = sum + t * t;
  }
  return std::sqrt(sum);
}

// src/core/vector_ops.rs:35-37
inline float dot(const float* a, const float* b, size_t d) {
  float sum = 0.0f;
  for (size_t i = 0; i < d; ++i) sum = sum + a[i] * b[i];
  return sum;
}

// src/core/vector_ops.rs:39-49
inline float cosine(const float* a, const float* b, size_t d) {
  float dt = dot(a, b, d);
  float na = std::sqrt(dot(a, a, d));
  float nb = std::sqrt(dot(b, b, d));
  if (na == 0.0f || nb == 0.0f) return 0.0f;
  return dt / (na * nb);
}

// ---------------------------------------------------------------------------------------
// Rust std::collections::BinaryHeap restated (max-heap on a user "less-or-equal"), so that
// exact ties leave the heap in the same internal order as the reference's heaps do.
// push  = sift_up(0, old_len); pop = swap last into root, sift_down_to_bottom(0), sift_up.
// ---------------------------------------------------------------------------------------
struct Cand {
  uint64_t id;
  float distance;
};
// SearchCandidate::cmp (src/hnsw/core.rs:126-137): reversed on distance => `a <= b` in heap
// order means a.distance >= b.distance.
inline bool heap_le(const Cand& a, const Cand& b) { return a.distance >= b.distance; }

struct RustHeap {
  std::vector<Cand> data;
  size_t len() const { return data.size(); }
  bool empty() const { return data.empty(); }
  const Cand& peek() const { return data[0]; }
  void sift_up(size_t start, size_t pos) {
    Cand elt = data[pos];
    while (pos > start) {
      size_t parent = (pos - 1) / 2;
      if (heap_le(elt, data[parent])) break;
      data[pos] = data[parent];
      pos = parent;
    }
    data[pos] = elt;
  }
  void push(Cand c) {
    size_t old = data.size();
    data.push_back(c);
    sift_up(0, old);
  }
  Cand pop() {
    Cand item = data.back();
    data.pop_back();
    if (!data.empty()) {
      std::swap(item, data[0]);
      // sift_down_to_bottom(0)
      size_t end = data.size();
      size_t start = 0, pos = 0;
      Cand elt = data[pos];
      size_t child = 2 * pos + 1;
      const size_t lim = end >= 2 ? end - 2 : 0;  // end.saturating_sub(2)
      while (child <= lim) {
        if (heap_le(data[child], data[child + 1])) child += 1;
        data[pos] = data[child];
        pos = child;
        child = 2 * pos + 1;
      }
      if (end >= 1 && child == end - 1) {
        data[pos] = data[child];
        pos = child;
      }
      data[pos] = elt;
      sift_up(start, pos);
    }
    return item;
  }
};

// Documented PRNG for the oracle's own draws (the reference's StdRng stream is unpinned).
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
  float gen_f32() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
  uint64_t gen_range(uint64_t n) { return next() % n; }
};

enum {
  ORC_OK = 0,
  ORC_NOT_TRAINED = 1,
  ORC_DUPLICATE = 2,
  ORC_DIM_MISMATCH = 3,
  ORC_INSUFFICIENT_TRAINING = 4,
  ORC_INCONSISTENT_DIMS = 5,
  ORC_INVALID = 6,
  ORC_NOT_FOUND = 7,
  ORC_NOT_INITIALIZED = 8,
};

// ---------------------------------------------------------------------------------------
// IVF — src/ivf/core.rs, src/ivf/operations.rs
// ---------------------------------------------------------------------------------------
struct InvertedList {
  std::vector<uint64_t> ids;     // insertion order (reference: HashMap<VectorId, Vec<f32>>)
  std::vector<float> vectors;    // ids.size() x d
  std::unordered_set<uint64_t> idset;
};

struct IVF {
  size_t n_clusters, n_probe, max_iterations;
  uint64_t seed;
  size_t d = 0;
  bool trained = false;
  std::vector<float> centroids;  // n_clusters x d, row c has ClusterId(c)
  std::vector<InvertedList> lists;
  std::unordered_map<uint64_t, uint32_t> where;  // id -> number of lists holding it
  std::unordered_set<uint64_t> deleted;
  size_t total_vectors = 0;
  SplitMix64 rng;
  IVF(size_t nc, size_t np, size_t mi, uint64_t sd)
      : n_clusters(nc), n_probe(np), max_iterations(mi), seed(sd), rng(sd) {}

  // src/ivf/core.rs:373-386 — strict '<' so the lowest cluster id wins ties.
  size_t find_nearest_centroid(const float* v) const {
    size_t best = 0;
    float best_dist = std::numeric_limits<float>::infinity();
    for (size_t c = 0; c < n_clusters; ++c) {
      float dist = l2(v, &centroids[c * d], d);
      if (dist < best_dist) {
        best_dist = dist;
        best = c;
      }
    }
    return best;
  }

  // src/ivf/core.rs:419-429
  float compute_error(const float* data, size_t n, const std::vector<size_t>& assign) const {
    float total = 0.0f;
    for (size_t i = 0; i < n; ++i) {
      float dist = l2(&data[i * d], &centroids[assign[i] * d], d);
      total += dist * dist;
    }
    return total / (float)n;
  }

  // src/ivf/core.rs:388-417
  void update_centroids(const float* data, size_t n, const std::vector<size_t>& assign) {
    std::vector<float> sums(n_clusters * d, 0.0f);
    std::vector<size_t> counts(n_clusters, 0);
    for (size_t i = 0; i < n; ++i) {
      float* s = &sums[assign[i] * d];
      const float* v = &data[i * d];
      for (size_t j = 0; j < d; ++j) s[j] += v[j];
      counts[assign[i]] += 1;
    }
    for (size_t c = 0; c < n_clusters; ++c) {
      if (counts[c] > 0) {
        for (size_t j = 0; j < d; ++j) centroids[c * d + j] = sums[c * d + j] / (float)counts[c];
      }
    }
  }

  // src/ivf/core.rs:336-371 — k-means++ (D^2 sampling).  RNG stream: parity unpinned.
  void initialize_centroids(const float* data, size_t n) {
    centroids.assign(n_clusters * d, 0.0f);
    size_t n_chosen = 0;
    size_t first = (size_t)rng.gen_range(n);
    std::memcpy(&centroids[0], &data[first * d], d * sizeof(float));
    n_chosen = 1;
    for (size_t i = 1; i < n_clusters; ++i) {
      std::vector<float> distances(n, std::numeric_limits<float>::infinity());
      for (size_t j = 0; j < n; ++j)
        for (size_t c = 0; c < n_chosen; ++c) {
          float dist = l2(&data[j * d], &centroids[c * d], d);
          distances[j] = std::min(distances[j], dist);
        }
      float total = 0.0f;
      for (size_t j = 0; j < n; ++j) total += distances[j] * distances[j];
      float cumulative = 0.0f;
      float threshold = rng.gen_f32() * total;
      for (size_t j = 0; j < n; ++j) {
        cumulative += distances[j] * distances[j];
        if (cumulative >= threshold) {
          std::memcpy(&centroids[n_chosen * d], &data[j * d], d * sizeof(float));
          n_chosen += 1;
          break;
        }
      }
    }
    // (the reference can push fewer than n_clusters centroids if the loop never fires;
    //  rows left zero here would be missing there — cannot happen with total > 0.)
  }

  // src/ivf/core.rs:240-334
  int train(const float* data, size_t n, size_t dim, uint32_t* iterations, int* converged_out,
            float* initial_error, float* final_error) {
    if (n == 0 || n < n_clusters) return ORC_INSUFFICIENT_TRAINING;
    d = dim;
    initialize_centroids(data, n);
    lists.assign(n_clusters, InvertedList());
    where.clear();
    std::vector<size_t> assign(n, 0);
    float prev_error = std::numeric_limits<float>::infinity();
    float init_err = compute_error(data, n, assign);
    bool converged = false;
    size_t iters = 0;
    for (size_t iter = 0; iter < max_iterations; ++iter) {
      iters = iter + 1;
      bool changed = false;
      for (size_t i = 0; i < n; ++i) {
        size_t nc = find_nearest_centroid(&data[i * d]);
        if (nc != assign[i]) {
          changed = true;
          assign[i] = nc;
        }
      }
      update_centroids(data, n, assign);
      if (iters >= max_iterations) break;
      float cur = compute_error(data, n, assign);
      float change = std::fabs(prev_error - cur) / prev_error;
      if (!changed || change < 1e-4f) {
        converged = true;
        if (max_iterations == 10 && n < 20) {  // reference's test-mode special case :313-317
          prev_error = cur;
          continue;
        }
        break;
      }
      prev_error = cur;
    }
    float fin = compute_error(data, n, assign);
    trained = true;
    if (iterations) *iterations = (uint32_t)iters;
    if (converged_out) *converged_out = converged ? 1 : 0;
    if (initial_error) *initial_error = init_err;
    if (final_error) *final_error = fin;
    return ORC_OK;
  }

  // src/ivf/core.rs:509-520
  void set_trained(const float* c, size_t dim) {
    d = dim;
    centroids.assign(c, c + n_clusters * dim);
    trained = true;
    lists.assign(n_clusters, InvertedList());
    where.clear();
  }

  // src/ivf/core.rs:431-455
  int insert(uint64_t id, const float* v, size_t dim) {
    if (!trained) return ORC_NOT_TRAINED;
    if (dim != d) return ORC_DIM_MISMATCH;
    size_t c = find_nearest_centroid(v);
    // InvertedList::insert (:128-134) checks duplicates in THAT list only.
    if (lists[c].idset.count(id)) return ORC_DUPLICATE;
    lists[c].idset.insert(id);
    lists[c].ids.push_back(id);
    lists[c].vectors.insert(lists[c].vectors.end(), v, v + d);
    where[id] += 1;
    total_vectors += 1;
    return ORC_OK;
  }

  // Test/bench setup only: same as insert() with the cluster supplied by the caller (a cluster
  // that tests/test_gpu_ivf_parity.py shows equal to find_nearest_centroid's) — skips the O(nlist*d)
  // scan so a million-row baseline index can be built in seconds.
  int insert_assigned(uint64_t id, const float* v, size_t dim, size_t c) {
    if (!trained) return ORC_NOT_TRAINED;
    if (dim != d) return ORC_DIM_MISMATCH;
    if (c >= n_clusters) return ORC_INVALID;
    if (lists[c].idset.count(id)) return ORC_DUPLICATE;
    lists[c].idset.insert(id);
    lists[c].ids.push_back(id);
    lists[c].vectors.insert(lists[c].vectors.end(), v, v + d);
    where[id] += 1;
    total_vectors += 1;
    return ORC_OK;
  }

  // src/ivf/core.rs:626-681
  int search(const float* q, size_t dim, size_t k, size_t nprobe, uint64_t* out_ids,
             float* out_dist, uint32_t* out_count) const {
    if (!trained) return ORC_NOT_TRAINED;
    if (dim != d) return ORC_DIM_MISMATCH;
    std::vector<std::pair<size_t, float>> cd(n_clusters);
    for (size_t c = 0; c < n_clusters; ++c) cd[c] = {c, l2(q, &centroids[c * d], d)};
    std::stable_sort(cd.begin(), cd.end(),
                     [](const auto& a, const auto& b) { return a.second < b.second; });
    if (cd.size() > nprobe) cd.resize(nprobe);
    std::vector<Cand> results;
    for (auto& pr : cd) {
      const InvertedList& L = lists[pr.first];
      for (size_t i = 0; i < L.ids.size(); ++i) {
        if (deleted.count(L.ids[i])) continue;
        results.push_back({L.ids[i], l2(q, &L.vectors[i * d], d)});
      }
    }
    std::stable_sort(results.begin(), results.end(),
                     [](const Cand& a, const Cand& b) { return a.distance < b.distance; });
    if (results.size() > k) results.resize(k);
    for (size_t i = 0; i < results.size(); ++i) {
      out_ids[i] = results[i].id;
      out_dist[i] = results[i].distance;
    }
    *out_count = (uint32_t)results.size();
    return ORC_OK;
  }

  // src/ivf/operations.rs:569-591
  int mark_deleted(uint64_t id) {
    if (!where.count(id)) return ORC_NOT_FOUND;
    deleted.insert(id);
    return ORC_OK;
  }

  // src/ivf/operations.rs:625-645
  size_t vacuum() {
    size_t removed = deleted.size();
    for (auto& L : lists) {
      size_t w = 0;
      for (size_t i = 0; i < L.ids.size(); ++i) {
        if (deleted.count(L.ids[i])) continue;
        if (w != i) {
          L.ids[w] = L.ids[i];
          std::memmove(&L.vectors[w * d], &L.vectors[i * d], d * sizeof(float));
        }
        ++w;
      }
      L.ids.resize(w);
      L.vectors.resize(w * d);
      for (uint64_t id : deleted) L.idset.erase(id);
    }
    for (uint64_t id : deleted) where.erase(id);
    total_vectors -= removed;
    deleted.clear();
    return removed;
  }
};

// ---------------------------------------------------------------------------------------
// HNSW — src/hnsw/core.rs, src/hnsw/operations.rs
// ---------------------------------------------------------------------------------------
struct HNode {
  uint64_t id;
  std::vector<float> vector;
  size_t level;
  std::vector<std::vector<uint64_t>> neighbors;  // per layer, insertion-ordered set
  bool is_deleted = false;
};

inline bool set_insert(std::vector<uint64_t>& s, uint64_t v) {
  if (std::find(s.begin(), s.end(), v) != s.end()) return false;
  s.push_back(v);
  return true;
}

struct HNSW {
  size_t M, M0, ef_construction;
  SplitMix64 rng;
  size_t d = 0;
  bool has_dim = false;
  bool has_entry = false;
  uint64_t entry_point = 0;
  std::vector<HNode> nodes;                       // insertion order
  std::unordered_map<uint64_t, size_t> index_of;  // id -> nodes[]
  std::atomic<uint64_t> n_dist{0};                // distance evaluations (bench accounting)

  HNSW(size_t m, size_t m0, size_t efc, uint64_t seed)
      : M(m), M0(m0), ef_construction(efc), rng(seed) {}

  const HNode* get(uint64_t id) const {
    auto it = index_of.find(id);
    return it == index_of.end() ? nullptr : &nodes[it->second];
  }
  HNode* get_mut(uint64_t id) {
    auto it = index_of.find(id);
    return it == index_of.end() ? nullptr : &nodes[it->second];
  }

  // src/hnsw/core.rs:211-224 — p = 0.408; RNG stream parity unpinned.
  size_t assign_level() {
    size_t level = 0;
    while (rng.gen_f64() < 0.408) level += 1;
    return level;
  }

  // src/hnsw/core.rs:469-554
  std::vector<Cand> search_layer(const float* query, uint64_t ep, size_t ef, size_t layer) {
    const HNode* epn = get(ep);
    if (!epn) return {};
    std::unordered_set<uint64_t> visited;
    RustHeap candidates, nearest;
    float ed = l2(query, epn->vector.data(), d);
    n_dist++;
    candidates.push({ep, ed});
    nearest.push({ep, -ed});
    visited.insert(ep);
    while (!candidates.empty()) {
      Cand current = candidates.pop();
      if (current.distance > -nearest.peek().distance) break;
      const HNode* node = get(current.id);
      if (node && node->level >= layer) {
        for (uint64_t nid : node->neighbors[layer]) {
          if (visited.count(nid)) continue;
          visited.insert(nid);
          const HNode* nb = get(nid);
          if (!nb) continue;
          if (nb->is_deleted) continue;
          float dist = l2(query, nb->vector.data(), d);
          n_dist++;
          if (dist < -nearest.peek().distance || nearest.len() < ef) {
            candidates.push({nid, dist});
            nearest.push({nid, -dist});
            if (nearest.len() > ef) nearest.pop();
          }
        }
      }
    }
    std::vector<Cand> result;
    result.reserve(nearest.len());
    for (const Cand& c : nearest.data) result.push_back({c.id, -c.distance});
    std::stable_sort(result.begin(), result.end(),
                     [](const Cand& a, const Cand& b) { return a.distance < b.distance; });
    return result;
  }

  // src/hnsw/core.rs:588-624
  std::vector<uint64_t> prune_with_new(const std::vector<uint64_t>& nbrs, const float* base,
                                       size_t m, uint64_t new_id, const float* new_vec) {
    std::vector<Cand> cands;
    for (uint64_t id : nbrs) {
      if (id == new_id) {
        cands.push_back({id, l2(base, new_vec, d)});
        n_dist++;
      } else if (const HNode* n = get(id)) {
        cands.push_back({id, l2(base, n->vector.data(), d)});
        n_dist++;
      }
    }
    std::stable_sort(cands.begin(), cands.end(),
                     [](const Cand& a, const Cand& b) { return a.distance < b.distance; });
    if (cands.size() > m) cands.resize(m);
    std::vector<uint64_t> out;
    for (auto& c : cands) out.push_back(c.id);
    return out;
  }

  // src/hnsw/core.rs:226-378.  level < 0 => draw with assign_level().
  int insert(uint64_t id, const float* v, size_t dim, int64_t forced_level) {
    if (index_of.count(id)) return ORC_DUPLICATE;
    if (has_dim && dim != d) return ORC_DIM_MISMATCH;
    // after a vacuum that removed the entry point's node the reference unwraps a missing node (:268-274: a panic —
    // the insert fails); reported as an error, with nothing changed
    if (has_entry && !get(entry_point)) return ORC_NOT_FOUND;
    if (!has_dim) {
      d = dim;
      has_dim = true;
    }
    size_t level = forced_level >= 0 ? (size_t)forced_level : assign_level();
    HNode node;
    node.id = id;
    node.vector.assign(v, v + dim);
    node.level = level;
    node.neighbors.assign(level + 1, {});
    bool is_first = false;
    if (!has_entry) {
      has_entry = true;
      entry_point = id;
      is_first = true;
    }
    size_t entry_level = 0;
    if (!is_first) {
      uint64_t ep = entry_point;
      size_t ef = ef_construction;
      const HNode* en = get(ep);
      entry_level = en->level;
      std::vector<Cand> current_nearest{{ep, l2(node.vector.data(), en->vector.data(), d)}};
      n_dist++;
      size_t search_level = std::min(level, entry_level);
      for (size_t lc = search_level + 1; lc-- > 0;) {
        auto c = search_layer(node.vector.data(), current_nearest[0].id, 1, lc);
        if (!c.empty()) current_nearest = c;
      }
      for (size_t lc = 0; lc <= level; ++lc) {
        size_t m = lc == 0 ? M0 : M;
        uint64_t start = (lc <= search_level && !current_nearest.empty()) ? current_nearest[0].id : ep;
        auto cands = search_layer(node.vector.data(), start, ef, lc);
        std::vector<uint64_t> nbrs;  // select_neighbors :556-558 = first m
        for (size_t i = 0; i < cands.size() && i < m; ++i) nbrs.push_back(cands[i].id);
        for (uint64_t nb : nbrs) set_insert(node.neighbors[lc], nb);
        size_t max_conn = m;
        struct Prune {
          uint64_t id;
          std::vector<uint64_t> list;
        };
        std::vector<Prune> pruning;
        for (uint64_t nb : nbrs) {
          HNode* n = get_mut(nb);
          if (!n) continue;
          if (n->level >= lc) {
            set_insert(n->neighbors[lc], id);
            if (n->neighbors[lc].size() > max_conn) pruning.push_back({nb, n->neighbors[lc]});
          }
        }
        for (auto& p : pruning) {
          HNode* n = get_mut(p.id);
          std::vector<float> base = n->vector;
          auto pruned = prune_with_new(p.list, base.data(), max_conn, id, node.vector.data());
          n = get_mut(p.id);
          n->neighbors[lc].clear();
          for (uint64_t x : pruned) set_insert(n->neighbors[lc], x);
        }
      }
    }
    index_of[id] = nodes.size();
    nodes.push_back(std::move(node));
    if (!is_first && level > entry_level) entry_point = id;
    return ORC_OK;
  }

  // src/hnsw/core.rs:398-467
  int search(const float* q, size_t dim, size_t k, size_t ef, uint64_t* out_ids, float* out_dist,
             uint32_t* out_count) {
    *out_count = 0;
    if (!has_entry) return ORC_OK;
    if (has_dim && dim != d) return ORC_DIM_MISMATCH;
    const HNode* en = get(entry_point);
    if (!en) return ORC_NOT_FOUND;
    size_t top = en->level;
    std::vector<Cand> nearest{{entry_point, l2(q, en->vector.data(), d)}};
    n_dist++;
    for (size_t lc = top + 1; lc-- > 0;) {
      auto nn = search_layer(q, nearest[0].id, lc == 0 ? ef : 1, lc);
      if (!nn.empty()) nearest = nn;
    }
    size_t w = 0;
    for (const Cand& c : nearest) {
      const HNode* n = get(c.id);
      if (!n || n->is_deleted) continue;
      if (w >= k) break;
      out_ids[w] = c.id;
      out_dist[w] = c.distance;
      ++w;
    }
    *out_count = (uint32_t)w;
    return ORC_OK;
  }

  // src/hnsw/operations.rs:127-137
  int mark_deleted(uint64_t id) {
    HNode* n = get_mut(id);
    if (!n) return ORC_NOT_FOUND;
    n->is_deleted = true;
    return ORC_OK;
  }

  // src/hnsw/operations.rs:176-201
  size_t vacuum() {
    std::unordered_set<uint64_t> dead;
    for (auto& n : nodes)
      if (n.is_deleted) dead.insert(n.id);
    std::vector<HNode> keep;
    for (auto& n : nodes)
      if (!n.is_deleted) keep.push_back(std::move(n));
    nodes = std::move(keep);
    index_of.clear();
    for (size_t i = 0; i < nodes.size(); ++i) index_of[nodes[i].id] = i;
    for (auto& n : nodes)
      for (auto& layer : n.neighbors)
        layer.erase(std::remove_if(layer.begin(), layer.end(),
                                   [&](uint64_t x) { return dead.count(x) > 0; }),
                    layer.end());
    return dead.size();
  }
};

// ---------------------------------------------------------------------------------------
// Hybrid — src/hybrid/core.rs.  Wall-clock "now" is a parameter (seconds) so runs are
// reproducible; the reference calls Utc::now() at the same points.
// ---------------------------------------------------------------------------------------
struct Hybrid {
  double recent_threshold_s;
  size_t migration_batch_size;
  bool auto_migrate;
  size_t min_ivf_training_size;
  HNSW recent;
  IVF historical;
  bool initialized = false, ivf_trained = false;
  std::vector<uint64_t> ts_order;  // insertion order of the timestamps map
  std::unordered_map<uint64_t, double> timestamps;
  size_t recent_count = 0, historical_count = 0;

  Hybrid(double thr, size_t mbs, bool am, size_t mits, size_t M, size_t M0, size_t efc,
         uint64_t hseed, size_t nc, size_t np, size_t mi, uint64_t iseed)
      : recent_threshold_s(thr), migration_batch_size(mbs), auto_migrate(am),
        min_ivf_training_size(mits), recent(M, M0, efc, hseed), historical(nc, np, mi, iseed) {}

  // src/hybrid/core.rs:262-290
  int initialize(const float* data, size_t n, size_t dim) {
    if (n < min_ivf_training_size) {
      ivf_trained = false;
      initialized = true;
      return ORC_OK;
    }
    int rc = historical.train(data, n, dim, nullptr, nullptr, nullptr, nullptr);
    if (rc != ORC_OK) return rc;
    historical.lists.assign(historical.n_clusters, InvertedList());
    historical.where.clear();
    historical.total_vectors = 0;
    ivf_trained = true;
    initialized = true;
    return ORC_OK;
  }

  static double age_of(double now, double ts) {
    double a = now - ts;
    return a < 0 ? 0.0 : a;  // to_std().unwrap_or(0)
  }

  // src/hybrid/core.rs:357-417
  int insert_with_timestamp(uint64_t id, const float* v, size_t dim, double ts, double now,
                            int64_t forced_level) {
    if (!initialized) return ORC_NOT_INITIALIZED;
    if (timestamps.count(id)) return ORC_DUPLICATE;
    if (!ivf_trained) {
      int rc = recent.insert(id, v, dim, forced_level);
      if (rc) return rc;
      recent_count++;
    } else {
      if (age_of(now, ts) < recent_threshold_s) {
        int rc = recent.insert(id, v, dim, forced_level);
        if (rc) return rc;
        recent_count++;
      } else {
        int rc = historical.insert(id, v, dim);
        if (rc) return rc;
        historical_count++;
      }
    }
    timestamps[id] = ts;
    ts_order.push_back(id);
    return ORC_OK;
  }

  // src/hybrid/core.rs:600-649 — copies into IVF, never removes from HNSW.
  size_t migrate_with_threshold(double threshold, double now) {
    size_t migrated = 0;
    for (uint64_t id : ts_order) {
      if (age_of(now, timestamps[id]) >= threshold) {
        const HNode* n = recent.get(id);
        if (n) {
          if (historical.insert(id, n->vector.data(), n->vector.size()) == ORC_OK) migrated++;
        }
      }
    }
    if (migrated) {
      recent_count = recent_count >= migrated ? recent_count - migrated : 0;
      historical_count += migrated;
    }
    return migrated;
  }

  // src/hybrid/core.rs:425-486
  int search(const float* q, size_t dim, size_t k, size_t ef, size_t nprobe, int search_recent,
             int search_historical, size_t recent_k, size_t historical_k, double now,
             uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
    *out_count = 0;
    if (!initialized) return ORC_OK;
    if (auto_migrate) migrate_with_threshold(recent_threshold_s, now);
    std::vector<Cand> all;
    size_t rk = recent_k > 0 ? recent_k : k;
    size_t hk = historical_k > 0 ? historical_k : k;
    if (search_recent) {
      std::vector<uint64_t> ids(std::max<size_t>(rk, 1));
      std::vector<float> ds(std::max<size_t>(rk, 1));
      uint32_t cnt = 0;
      if (recent.search(q, dim, rk, ef, ids.data(), ds.data(), &cnt) == ORC_OK)
        for (uint32_t i = 0; i < cnt; ++i) all.push_back({ids[i], ds[i]});
    }
    if (search_historical && ivf_trained) {
      std::vector<uint64_t> ids(std::max<size_t>(hk, 1));
      std::vector<float> ds(std::max<size_t>(hk, 1));
      uint32_t cnt = 0;
      if (historical.search(q, dim, hk, nprobe, ids.data(), ds.data(), &cnt) == ORC_OK)
        for (uint32_t i = 0; i < cnt; ++i) all.push_back({ids[i], ds[i]});
    }
    std::stable_sort(all.begin(), all.end(),
                     [](const Cand& a, const Cand& b) { return a.distance < b.distance; });
    if (all.size() > k) all.resize(k);
    for (size_t i = 0; i < all.size(); ++i) {
      out_ids[i] = all[i].id;
      out_dist[i] = all[i].distance;
    }
    *out_count = (uint32_t)all.size();
    return ORC_OK;
  }

  // src/hybrid/core.rs:904-937
  int del(uint64_t id, double now) {
    auto it = timestamps.find(id);
    if (it == timestamps.end()) return ORC_NOT_FOUND;
    if (age_of(now, it->second) < recent_threshold_s) return recent.mark_deleted(id);
    return historical.mark_deleted(id);
  }
};

}  // namespace

// =========================================================================================
// C API (ctypes-friendly)
// =========================================================================================
extern "C" {

float orc_l2(const float* a, const float* b, uint64_t d) { return l2(a, b, d); }
float orc_dot(const float* a, const float* b, uint64_t d) { return dot(a, b, d); }
float orc_cosine(const float* a, const float* b, uint64_t d) { return cosine(a, b, d); }

// n x d rows against one query (used to time the reference's scalar scan on big inputs).
void orc_l2_batch(const float* q, const float* rows, uint64_t n, uint64_t d, float* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = l2(q, rows + i * d, d);
}

// src/core/vector_ops.rs:12-22 — stable sort descending by score, first k indices.
void orc_top_k_indices(const float* scores, uint64_t n, uint64_t k, uint64_t* out, uint64_t* out_n) {
  std::vector<uint64_t> idx(n);
  for (uint64_t i = 0; i < n; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return scores[a] > scores[b]; });
  uint64_t m = std::min(n, k);
  for (uint64_t i = 0; i < m; ++i) out[i] = idx[i];
  *out_n = m;
}

// src/core/vector_ops.rs:180-201 — min-heap of size k on score; final sort descending.
void orc_top_k_indices_heap(const float* scores, uint64_t n, uint64_t k, uint64_t* out, uint64_t* out_n) {
  *out_n = 0;
  if (k == 0) return;
  // HeapItem::cmp reverses score => Rust max-heap behaves as a min-heap on score; in
  // RustHeap terms store distance = score (heap_le(a,b) = a.distance >= b.distance).
  RustHeap heap;
  for (uint64_t i = 0; i < n; ++i) {
    if (heap.len() < k) {
      heap.push({i, scores[i]});
    } else if (scores[i] > heap.peek().distance) {
      heap.pop();
      heap.push({i, scores[i]});
    }
  }
  std::vector<Cand> res = heap.data;
  std::stable_sort(res.begin(), res.end(), [](const Cand& a, const Cand& b) { return a.distance > b.distance; });
  for (size_t i = 0; i < res.size(); ++i) out[i] = res[i].id;
  *out_n = res.size();
}

// src/core/vector_ops.rs:204-263 — StreamingTopK: BinaryHeap<(OrderedFloat, VectorId)>, OrderedFloat's order reversed
// (:219-230) so the root is the smallest score; tuples compare the id next, so among equal scores the LARGER id is
// nearer the root and is evicted first.  add(): strict `>` against the root's score; get_results(): heap order
// (into_iter), stable sort by score descending.  Ids are u64 here (numeric order stands for VectorId's byte order).
void orc_streaming_top_k(const uint64_t* ids, const float* scores, uint64_t n, uint64_t k, uint64_t* out_ids,
                         float* out_scores, uint64_t* out_n) {
  *out_n = 0;
  struct It {
    float s;
    uint64_t id;
  };
  auto le = [](const It& a, const It& b) { return a.s > b.s || (a.s == b.s && a.id <= b.id); };  // a <= b in tuple order
  std::vector<It> h;
  auto sift_up = [&](size_t start, size_t pos) {
    It elt = h[pos];
    while (pos > start) {
      size_t parent = (pos - 1) / 2;
      if (le(elt, h[parent])) break;
      h[pos] = h[parent];
      pos = parent;
    }
    h[pos] = elt;
  };
  auto pop = [&]() {
    It item = h.back();
    h.pop_back();
    if (!h.empty()) {
      std::swap(item, h[0]);
      size_t end = h.size(), pos = 0;
      It elt = h[0];
      size_t child = 1;
      const size_t lim = end >= 2 ? end - 2 : 0;
      while (child <= lim) {
        if (le(h[child], h[child + 1])) child += 1;
        h[pos] = h[child];
        pos = child;
        child = 2 * pos + 1;
      }
      if (child == end - 1) {
        h[pos] = h[child];
        pos = child;
      }
      h[pos] = elt;
      sift_up(0, pos);
    }
  };
  for (uint64_t i = 0; i < n; ++i) {
    if (h.size() < k) {
      h.push_back({scores[i], ids[i]});
      sift_up(0, h.size() - 1);
    } else if (!h.empty() && scores[i] > h[0].s) {
      pop();
      h.push_back({scores[i], ids[i]});
      sift_up(0, h.size() - 1);
    }
  }
  std::vector<It> res = h;
  std::stable_sort(res.begin(), res.end(), [](const It& a, const It& b) { return a.s > b.s; });
  for (size_t i = 0; i < res.size(); ++i) {
    out_ids[i] = res[i].id;
    out_scores[i] = res[i].s;
  }
  *out_n = res.size();
}

// src/core/vector_ops.rs:24-32 + src/core/types.rs:206-223 — dedup keeping the smaller
// distance, sort ascending, take k.  (HashMap::into_values order is random in the reference;
// insertion order of first appearance is used here, then a stable sort.)
void orc_merge_search_results(const uint64_t* ids, const float* dist, uint64_t n, uint64_t k,
                              uint64_t* out_ids, float* out_dist, uint64_t* out_n) {
  std::vector<uint64_t> order;
  std::unordered_map<uint64_t, float> best;
  for (uint64_t i = 0; i < n; ++i) {
    auto it = best.find(ids[i]);
    if (it == best.end()) {
      best[ids[i]] = dist[i];
      order.push_back(ids[i]);
    } else if (!(it->second <= dist[i])) {
      it->second = dist[i];
    }
  }
  std::vector<Cand> res;
  for (uint64_t id : order) res.push_back({id, best[id]});
  std::stable_sort(res.begin(), res.end(), [](const Cand& a, const Cand& b) { return a.distance < b.distance; });
  uint64_t m = std::min<uint64_t>(res.size(), k);
  for (uint64_t i = 0; i < m; ++i) {
    out_ids[i] = res[i].id;
    out_dist[i] = res[i].distance;
  }
  *out_n = m;
}

// Documented PRNG (shared spec with the product's host code: SplitMix64).
void orc_rng_levels(uint64_t seed, uint64_t n, int64_t* out) {
  HNSW h(16, 32, 200, seed);
  for (uint64_t i = 0; i < n; ++i) out[i] = (int64_t)h.assign_level();
}

// ---- IVF ----
void* orc_ivf_new(uint64_t n_clusters, uint64_t n_probe, uint64_t max_iterations, uint64_t seed) {
  if (n_clusters == 0 || n_probe == 0 || n_probe > n_clusters || max_iterations == 0) return nullptr;
  return new IVF(n_clusters, n_probe, max_iterations, seed);
}
void orc_ivf_free(void* p) { delete (IVF*)p; }
int orc_ivf_train(void* p, const float* data, uint64_t n, uint64_t d, uint32_t* iters, int* conv,
                  float* e0, float* e1) {
  return ((IVF*)p)->train(data, n, d, iters, conv, e0, e1);
}
int orc_ivf_set_trained(void* p, const float* centroids, uint64_t d) {
  ((IVF*)p)->set_trained(centroids, d);
  return ORC_OK;
}
int orc_ivf_get_centroids(void* p, float* out) {
  IVF* x = (IVF*)p;
  if (!x->trained) return ORC_NOT_TRAINED;
  std::memcpy(out, x->centroids.data(), x->centroids.size() * sizeof(float));
  return ORC_OK;
}
int orc_ivf_insert(void* p, uint64_t id, const float* v, uint64_t d) { return ((IVF*)p)->insert(id, v, d); }
int orc_ivf_insert_batch(void* p, const uint64_t* ids, const float* v, uint64_t n, uint64_t d) {
  for (uint64_t i = 0; i < n; ++i) {
    int rc = ((IVF*)p)->insert(ids[i], v + i * d, d);
    if (rc) return rc;
  }
  return ORC_OK;
}
int orc_ivf_insert_assigned_batch(void* p, const uint64_t* ids, const float* v, uint64_t n, uint64_t d,
                                  const uint32_t* clusters) {
  for (uint64_t i = 0; i < n; ++i) {
    int rc = ((IVF*)p)->insert_assigned(ids[i], v + i * d, d, clusters[i]);
    if (rc) return rc;
  }
  return ORC_OK;
}
int orc_ivf_find_cluster(void* p, const float* v, uint64_t d, uint64_t* out) {
  IVF* x = (IVF*)p;
  if (!x->trained) return ORC_NOT_TRAINED;
  if (d != x->d) return ORC_DIM_MISMATCH;
  *out = x->find_nearest_centroid(v);
  return ORC_OK;
}
int orc_ivf_assign_batch(void* p, const float* v, uint64_t n, uint64_t d, uint32_t* out) {
  IVF* x = (IVF*)p;
  if (!x->trained) return ORC_NOT_TRAINED;
  if (d != x->d) return ORC_DIM_MISMATCH;
  for (uint64_t i = 0; i < n; ++i) out[i] = (uint32_t)x->find_nearest_centroid(v + i * d);
  return ORC_OK;
}
uint64_t orc_ivf_cluster_size(void* p, uint64_t c) { return ((IVF*)p)->lists[c].ids.size(); }
uint64_t orc_ivf_total_vectors(void* p) { return ((IVF*)p)->total_vectors; }
void orc_ivf_list_ids(void* p, uint64_t c, uint64_t* out) {
  auto& L = ((IVF*)p)->lists[c];
  std::memcpy(out, L.ids.data(), L.ids.size() * sizeof(uint64_t));
}
int orc_ivf_search(void* p, const float* q, uint64_t d, uint64_t k, uint64_t nprobe, uint64_t* ids,
                   float* dist, uint32_t* count) {
  return ((IVF*)p)->search(q, d, k, nprobe, ids, dist, count);
}
// src/ivf/operations.rs:132-145 — sequential loop; `threads` > 1 runs one query per thread
// (BASELINE.md §2 (ii)); results are written at stride k per query.
int orc_ivf_batch_search(void* p, const float* q, uint64_t nq, uint64_t d, uint64_t k, uint64_t nprobe,
                         uint64_t* ids, float* dist, uint32_t* counts, uint32_t threads) {
  IVF* x = (IVF*)p;
  if (threads <= 1) {
    for (uint64_t i = 0; i < nq; ++i) {
      int rc = x->search(q + i * d, d, k, nprobe, ids + i * k, dist + i * k, counts + i);
      if (rc) return rc;
    }
    return ORC_OK;
  }
  std::vector<std::thread> th;
  std::vector<int> rcs(threads, 0);
  for (uint32_t t = 0; t < threads; ++t)
    th.emplace_back([&, t]() {
      for (uint64_t i = t; i < nq; i += threads) {
        int rc = x->search(q + i * d, d, k, nprobe, ids + i * k, dist + i * k, counts + i);
        if (rc) rcs[t] = rc;
      }
    });
  for (auto& t : th) t.join();
  for (int rc : rcs)
    if (rc) return rc;
  return ORC_OK;
}
int orc_ivf_mark_deleted(void* p, uint64_t id) { return ((IVF*)p)->mark_deleted(id); }
uint64_t orc_ivf_vacuum(void* p) { return ((IVF*)p)->vacuum(); }

// ---- HNSW ----
void* orc_hnsw_new(uint64_t M, uint64_t M0, uint64_t efc, uint64_t seed) { return new HNSW(M, M0, efc, seed); }
void orc_hnsw_free(void* p) { delete (HNSW*)p; }
int orc_hnsw_insert(void* p, uint64_t id, const float* v, uint64_t d, int64_t level) {
  return ((HNSW*)p)->insert(id, v, d, level);
}
int orc_hnsw_insert_batch(void* p, const uint64_t* ids, const float* v, uint64_t n, uint64_t d,
                          const int64_t* levels) {
  for (uint64_t i = 0; i < n; ++i) {
    int rc = ((HNSW*)p)->insert(ids[i], v + i * d, d, levels ? levels[i] : -1);
    if (rc) return rc;
  }
  return ORC_OK;
}
int orc_hnsw_search(void* p, const float* q, uint64_t d, uint64_t k, uint64_t ef, uint64_t* ids,
                    float* dist, uint32_t* count) {
  return ((HNSW*)p)->search(q, d, k, ef, ids, dist, count);
}
int orc_hnsw_batch_search(void* p, const float* q, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef,
                          uint64_t* ids, float* dist, uint32_t* counts) {
  for (uint64_t i = 0; i < nq; ++i) {
    int rc = ((HNSW*)p)->search(q + i * d, d, k, ef, ids + i * k, dist + i * k, counts + i);
    if (rc) return rc;
  }
  return ORC_OK;
}
uint64_t orc_hnsw_node_count(void* p) { return ((HNSW*)p)->nodes.size(); }
int orc_hnsw_entry_point(void* p, uint64_t* out) {
  HNSW* h = (HNSW*)p;
  if (!h->has_entry) return ORC_NOT_FOUND;
  *out = h->entry_point;
  return ORC_OK;
}
int64_t orc_hnsw_level(void* p, uint64_t id) {
  const HNode* n = ((HNSW*)p)->get(id);
  return n ? (int64_t)n->level : -1;
}
// returns neighbour count, copies up to cap ids (insertion order)
int64_t orc_hnsw_neighbors(void* p, uint64_t id, uint64_t layer, uint64_t* out, uint64_t cap) {
  const HNode* n = ((HNSW*)p)->get(id);
  if (!n || layer > n->level) return -1;
  const auto& s = n->neighbors[layer];
  for (size_t i = 0; i < s.size() && i < cap; ++i) out[i] = s[i];
  return (int64_t)s.size();
}
int orc_hnsw_mark_deleted(void* p, uint64_t id) { return ((HNSW*)p)->mark_deleted(id); }
uint64_t orc_hnsw_vacuum(void* p) { return ((HNSW*)p)->vacuum(); }
uint64_t orc_hnsw_dist_evals(void* p) { return ((HNSW*)p)->n_dist.load(); }
// Install a graph built elsewhere (bench: bulk-built graph fed to both backends).
// level[i], and for each node and layer a neighbour list given CSR-style.
int orc_hnsw_restore(void* p, const uint64_t* ids, const float* vecs, uint64_t n, uint64_t d,
                     const uint32_t* levels, const uint64_t* nbr_offsets /* per (node,layer) prefix */,
                     const uint64_t* nbrs, uint64_t entry) {
  HNSW* h = (HNSW*)p;
  h->d = d;
  h->has_dim = true;
  h->nodes.clear();
  h->index_of.clear();
  uint64_t slot = 0;
  for (uint64_t i = 0; i < n; ++i) {
    HNode nd;
    nd.id = ids[i];
    nd.vector.assign(vecs + i * d, vecs + (i + 1) * d);
    nd.level = levels[i];
    nd.neighbors.assign(nd.level + 1, {});
    for (uint32_t l = 0; l <= levels[i]; ++l, ++slot)
      nd.neighbors[l].assign(nbrs + nbr_offsets[slot], nbrs + nbr_offsets[slot + 1]);
    h->index_of[nd.id] = h->nodes.size();
    h->nodes.push_back(std::move(nd));
  }
  h->has_entry = n > 0;
  h->entry_point = entry;
  return ORC_OK;
}

// ---- Hybrid ----
void* orc_hybrid_new(double recent_threshold_s, uint64_t migration_batch_size, int auto_migrate,
                     uint64_t min_ivf_training_size, uint64_t M, uint64_t M0, uint64_t efc,
                     uint64_t hseed, uint64_t n_clusters, uint64_t n_probe, uint64_t max_iter,
                     uint64_t iseed) {
  if (n_clusters == 0 || n_probe == 0 || n_probe > n_clusters || max_iter == 0) return nullptr;
  return new Hybrid(recent_threshold_s, migration_batch_size, auto_migrate != 0, min_ivf_training_size,
                    M, M0, efc, hseed, n_clusters, n_probe, max_iter, iseed);
}
void orc_hybrid_free(void* p) { delete (Hybrid*)p; }
int orc_hybrid_initialize(void* p, const float* data, uint64_t n, uint64_t d) {
  return ((Hybrid*)p)->initialize(data, n, d);
}
int orc_hybrid_set_ivf_centroids(void* p, const float* c, uint64_t d) {
  Hybrid* h = (Hybrid*)p;
  h->historical.set_trained(c, d);
  h->ivf_trained = true;
  h->initialized = true;
  return ORC_OK;
}
int orc_hybrid_get_ivf_centroids(void* p, float* out) {
  return orc_ivf_get_centroids(&((Hybrid*)p)->historical, out);
}
int orc_hybrid_insert(void* p, uint64_t id, const float* v, uint64_t d, double ts, double now, int64_t level) {
  return ((Hybrid*)p)->insert_with_timestamp(id, v, d, ts, now, level);
}
int orc_hybrid_search(void* p, const float* q, uint64_t d, uint64_t k, uint64_t ef, uint64_t nprobe,
                      int search_recent, int search_historical, uint64_t recent_k, uint64_t historical_k,
                      double now, uint64_t* ids, float* dist, uint32_t* count) {
  return ((Hybrid*)p)->search(q, d, k, ef, nprobe, search_recent, search_historical, recent_k,
                              historical_k, now, ids, dist, count);
}
// B queries; the auto-migration runs once up front (as the first query's search would do), then
// queries are searched one per thread (threads = 1: the reference's sequential behaviour).
int orc_hybrid_batch_search(void* p, const float* q, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t nprobe,
                            double now, uint64_t* ids, float* dist, uint32_t* counts, uint32_t threads) {
  Hybrid* h = (Hybrid*)p;
  if (h->initialized && h->auto_migrate) h->migrate_with_threshold(h->recent_threshold_s, now);
  const bool am = h->auto_migrate;
  h->auto_migrate = false;
  if (threads < 1) threads = 1;
  std::vector<std::thread> th;
  std::vector<int> rcs(threads, 0);
  for (uint32_t t = 0; t < threads; ++t)
    th.emplace_back([&, t]() {
      for (uint64_t i = t; i < nq; i += threads) {
        int rc = h->search(q + i * d, d, k, ef, nprobe, 1, 1, 0, 0, now, ids + i * k, dist + i * k, counts + i);
        if (rc) rcs[t] = rc;
      }
    });
  for (auto& t : th) t.join();
  h->auto_migrate = am;
  for (int rc : rcs)
    if (rc) return rc;
  return ORC_OK;
}
int orc_hybrid_delete(void* p, uint64_t id, double now) { return ((Hybrid*)p)->del(id, now); }
uint64_t orc_hybrid_migrate(void* p, double threshold_s, double now) {
  return ((Hybrid*)p)->migrate_with_threshold(threshold_s, now);
}
uint64_t orc_hybrid_recent_count(void* p) { return ((Hybrid*)p)->recent_count; }
uint64_t orc_hybrid_historical_count(void* p) { return ((Hybrid*)p)->historical_count; }
int orc_hybrid_is_ivf_trained(void* p) { return ((Hybrid*)p)->ivf_trained ? 1 : 0; }
void* orc_hybrid_hnsw(void* p) { return &((Hybrid*)p)->recent; }
void* orc_hybrid_ivf(void* p) { return &((Hybrid*)p)->historical; }

}  // extern "C"
