// fvdb_graph.cpp — the HNSW graph resident in HBM: traversal (fvdb_graph_search_dev*) and construction
// (fvdb_graph_insert_linked) entry points of the C ABI (include/fvdb.h).  gfx950 only.
//
// Adjacency has a fixed stride on every layer — row = [count, neighbours in list order] — so an insert rewrites only
// the rows it touches: the device-side insert (kernels_graph_build.h) edits them in place, a host-side insert patches
// them through fvdb_graph_set_lists.  Nothing re-flattens or re-uploads the whole graph after a mutation.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "fvdb_internal.h"
#include "common.h"
#include "kernels_graph.h"
#include "kernels_graph_fast.h"
#include "kernels_graph_build.h"

using namespace fvdb;

struct fvdb_graph {
  fvdb_store* store = nullptr;
  uint32_t M = 0, M0 = 0;  // degree caps the strides are sized for (fvdb_graph_configure); 0 = follow the uploaded lists
  uint32_t stride0 = 0, strideU = 0;
  uint32_t n = 0, n_cap = 0;        // nodes (= store rows mirrored), capacity
  uint32_t u_rows = 0, u_cap = 0;   // rows of the layers above 0
  uint32_t entry = 0, top_level = 0;
  bool has_entry = false;
  DBuf d_level, d_deleted, d_ubase, d_adj0, d_adjU, d_dist0, d_distU, d_stamp0, d_stampU, d_state;
  DBuf d_spec, d_elog, d_chg, s_patch, s_codes;
  HBuf h_state, h_patch;
  bool dist_valid = false;          // dist0 / distU hold the distance of every stored edge
  uint32_t tag = 0;                 // batch counter for the row stamps (never 0)
  std::vector<uint32_t> h_level, h_ubase;
  uint64_t upload_bytes = 0;        // host -> device bytes of graph STRUCTURE (not vectors) since creation
  fvdb_graph_insert_stats last{};
  DBuf s_q, d_counters;
  uint64_t tot_queries = 0, tot_again = 0;  // traversal counters [3], [2] folded in at every fvdb_graph_kernel_times
  static constexpr uint32_t kSlots = 16;  // batches that may be in flight at once, each on its own stream
  DBuf s_visited[kSlots], s_touched[kSlots], s_spill[kSlots];
  uint32_t vis_B[kSlots] = {}, vis_words = 0, vis_tcap = 0, vis_stride = 0;
  bool uploaded = false;
  std::vector<uint8_t> h_deleted;  // host copy of the flags: searches skip the per-neighbour flag load when none is set
  uint64_t n_deleted = 0;
  // profiling: HIP events around the last launches of the traversal kernel (ring of 64)
  hipEvent_t kev[64][2] = {};
  uint32_t kev_n = 0;   // launches recorded since the last fvdb_graph_kernel_times call
  std::mutex mu;        // launch bookkeeping: searches in different slots may come from different host threads
};

namespace {

__global__ void graph_pad_rows_kernel(const float* __restrict__ src, uint32_t d, uint32_t dpad, uint64_t n, float* __restrict__ dst) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * dpad) return;
  const uint64_t r = i / dpad;
  const uint32_t c = (uint32_t)(i - r * dpad);
  dst[i] = c < d ? src[r * d + c] : 0.0f;
}

// packed rows [code, count, neighbours ...] (stride `ps` words) -> adjacency rows; one wave per row
__global__ __launch_bounds__(256) void graph_patch_kernel(const uint32_t* __restrict__ packed, uint32_t ps, uint32_t n_rows, uint32_t* __restrict__ adj0,
                                                          uint32_t stride0, uint32_t* __restrict__ adjU, uint32_t strideU) {
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (i >= n_rows) return;
  const uint32_t* p = packed + (size_t)i * ps;
  const uint32_t code = p[0], cnt = p[1];
  uint32_t* row = (code >> 31) ? adjU + (size_t)(code & 0x7FFFFFFFu) * strideU : adj0 + (size_t)code * stride0;
  if (lane < cnt) row[1 + lane] = p[2 + lane];
  if (lane == 0) row[0] = cnt;
}

// grow a device array to `new_bytes`, keeping the first `keep_bytes` and zeroing the rest
int grow_keep(fvdb_ctx* ctx, DBuf& b, size_t keep_bytes, size_t new_bytes) {
  if (new_bytes <= b.cap) return FVDB_OK;
  void* np = nullptr;
  HIPCHK(ctx, hipMalloc(&np, new_bytes));
  if (keep_bytes) HIPCHK(ctx, hipMemcpyAsync(np, b.p, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync((char*)np + keep_bytes, 0, new_bytes - keep_bytes, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (b.p) (void)hipFree(b.p);
  b.p = np;
  b.cap = new_bytes;
  return FVDB_OK;
}

int reserve_nodes(fvdb_graph* g, uint32_t n_nodes, uint32_t u_rows) {
  fvdb_ctx* ctx = g->store->ctx;
  if (n_nodes > g->n_cap) {
    const uint32_t nc = std::max<uint32_t>(n_nodes, g->n_cap + g->n_cap / 2 + 1024);
    const size_t keep = g->n;
    int rc;
    if ((rc = grow_keep(ctx, g->d_level, keep * 4, (size_t)nc * 4))) return rc;
    if ((rc = grow_keep(ctx, g->d_deleted, keep * 4, (size_t)nc * 4))) return rc;
    if ((rc = grow_keep(ctx, g->d_ubase, keep * 4, (size_t)nc * 4))) return rc;
    if ((rc = grow_keep(ctx, g->d_stamp0, keep * 4, (size_t)nc * 4))) return rc;
    if ((rc = grow_keep(ctx, g->d_adj0, keep * g->stride0 * 4, (size_t)nc * g->stride0 * 4))) return rc;
    if ((rc = grow_keep(ctx, g->d_dist0, keep * g->stride0 * 4, (size_t)nc * g->stride0 * 4))) return rc;
    g->n_cap = nc;
    for (auto& v : g->vis_B) v = 0;  // visited maps are sized by the node count
  }
  if (u_rows > g->u_cap) {
    const uint32_t uc = std::max<uint32_t>(u_rows, g->u_cap + g->u_cap / 2 + 1024);
    const size_t keep = g->u_rows;
    int rc;
    if ((rc = grow_keep(ctx, g->d_stampU, keep * 4, (size_t)uc * 4))) return rc;
    if ((rc = grow_keep(ctx, g->d_adjU, keep * g->strideU * 4, (size_t)uc * g->strideU * 4))) return rc;
    if ((rc = grow_keep(ctx, g->d_distU, keep * g->strideU * 4, (size_t)uc * g->strideU * 4))) return rc;
    g->u_cap = uc;
  }
  return FVDB_OK;
}

int push_state(fvdb_graph* g, const BuildState& st) {
  fvdb_ctx* ctx = g->store->ctx;
  HIPCHK(ctx, g->d_state.ensure(sizeof(BuildState)));
  HIPCHK(ctx, g->h_state.ensure(sizeof(BuildState)));
  std::memcpy(g->h_state.p, &st, sizeof(BuildState));
  HIPCHK(ctx, hipMemcpyAsync(g->d_state.p, g->h_state.p, sizeof(BuildState), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

int pull_state(fvdb_graph* g, BuildState* st) {
  fvdb_ctx* ctx = g->store->ctx;
  HIPCHK(ctx, g->h_state.ensure(sizeof(BuildState)));
  HIPCHK(ctx, hipMemcpyAsync(g->h_state.p, g->d_state.p, sizeof(BuildState), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(st, g->h_state.p, sizeof(BuildState));
  return FVDB_OK;
}

BuildView build_view(fvdb_graph* g, uint32_t ef, uint32_t cand_cap) {
  fvdb_store* s = g->store;
  BuildView v{};
  v.rows = s->data;
  v.dpad = s->dpad;
  v.level = g->d_level.as<uint32_t>();
  v.deleted = g->d_deleted.as<uint32_t>();
  v.any_deleted = g->n_deleted ? 1u : 0u;
  v.ubase = g->d_ubase.as<uint32_t>();
  v.adj0 = g->d_adj0.as<uint32_t>();
  v.dist0 = g->d_dist0.as<float>();
  v.adjU = g->d_adjU.as<uint32_t>();
  v.distU = g->d_distU.as<float>();
  v.stride0 = g->stride0;
  v.strideU = g->strideU;
  v.stamp0 = g->d_stamp0.as<uint32_t>();
  v.stampU = g->d_stampU.as<uint32_t>();
  v.M = g->M;
  v.M0 = g->M0;
  v.ef = ef;
  v.bitmap_words = (g->n + 31) / 32;
  v.cand_cap = cand_cap;
  v.state = (BuildState*)g->d_state.p;
  v.dbg = nullptr;
  return v;
}

// kernels are instantiated per 128-dim block count; FULL (no bounds checks) for the BASELINE dimensions 384 and 768
#define FVDB_BUILD_DISPATCH(NBV, FULLV, ...) \
  do {                                       \
    constexpr int NB_ = NBV;                 \
    constexpr bool FULL_ = FULLV;            \
    __VA_ARGS__;                             \
  } while (0)
#define FVDB_BUILD_SWITCH(dpad, ...)                                           \
  do {                                                                         \
    const uint32_t nb128_ = ((dpad) + 127) / 128;                              \
    if ((dpad) == 384) FVDB_BUILD_DISPATCH(3, true, __VA_ARGS__);              \
    else if ((dpad) == 768) FVDB_BUILD_DISPATCH(6, true, __VA_ARGS__);         \
    else if (nb128_ == 1) FVDB_BUILD_DISPATCH(1, false, __VA_ARGS__);          \
    else if (nb128_ == 2) FVDB_BUILD_DISPATCH(2, false, __VA_ARGS__);          \
    else if (nb128_ == 3) FVDB_BUILD_DISPATCH(3, false, __VA_ARGS__);          \
    else if (nb128_ == 4) FVDB_BUILD_DISPATCH(4, false, __VA_ARGS__);          \
    else if (nb128_ <= 6) FVDB_BUILD_DISPATCH(6, false, __VA_ARGS__);          \
    else FVDB_BUILD_DISPATCH(8, false, __VA_ARGS__);                           \
  } while (0)

// distances of the stored edges: all rows (codes == nullptr) or the listed ones
int edge_dist(fvdb_graph* g, const uint32_t* codes_dev, const uint32_t* owner_dev, uint32_t n_rows, bool upper_all) {
  fvdb_ctx* ctx = g->store->ctx;
  if (n_rows == 0) return FVDB_OK;
  const BuildView v = build_view(g, 1, 1);
  const size_t lds = 4 * (size_t)kTileRows * kFastStride * 4;
  FVDB_BUILD_SWITCH(g->store->dpad, {
    hipLaunchKernelGGL((graph_edge_dist_kernel<NB_, FULL_>), dim3((n_rows + 3) / 4), dim3(256), lds, ctx->stream, v, codes_dev,
                       owner_dev, n_rows, upper_all ? 1u : 0u);
  });
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

int ensure_edge_dist(fvdb_graph* g) {
  fvdb_ctx* ctx = g->store->ctx;
  if (g->dist_valid) return FVDB_OK;
  int rc = edge_dist(g, nullptr, nullptr, g->n, false);
  if (rc) return rc;
  if (g->u_rows) {  // owner of every upper row
    std::vector<uint32_t> owner(g->u_rows);
    for (uint32_t i = 0; i < g->n; ++i)
      for (uint32_t l = 1; l <= g->h_level[i]; ++l) owner[g->h_ubase[i] + l - 1] = i;
    HIPCHK(ctx, g->s_codes.ensure((size_t)g->u_rows * 4));
    HIPCHK(ctx, hipMemcpyAsync(g->s_codes.p, owner.data(), (size_t)g->u_rows * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    rc = edge_dist(g, nullptr, g->s_codes.as<uint32_t>(), g->u_rows, true);
    if (rc) return rc;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  g->dist_valid = true;
  return FVDB_OK;
}

}  // namespace

extern "C" {

// =============================================================================================
// device-resident graph: structure
// =============================================================================================
int fvdb_graph_create(fvdb_store* s, fvdb_graph** out) {
  if (!s || !out) return FVDB_E_INVALID;
  *out = nullptr;
  fvdb_graph* g = new (std::nothrow) fvdb_graph();
  if (!g) return FVDB_E_OOM;
  g->store = s;
  *out = g;
  return FVDB_OK;
}

void fvdb_graph_destroy(fvdb_graph* g) {
  if (!g) return;
  for (auto& e : g->kev)
    for (auto& x : e)
      if (x) (void)hipEventDestroy(x);
  (void)hipSetDevice(g->store->ctx->device);
  (void)hipStreamSynchronize(g->store->ctx->stream);
  DBuf* bufs[] = {&g->d_level, &g->d_deleted, &g->d_ubase, &g->d_adj0, &g->d_adjU, &g->d_dist0, &g->d_distU, &g->d_stamp0,
                  &g->d_stampU, &g->d_state, &g->d_spec, &g->d_elog, &g->d_chg, &g->s_patch, &g->s_codes, &g->s_q, &g->d_counters};
  for (auto& b : g->s_visited) b.release();
  for (auto& b : g->s_touched) b.release();
  for (auto& b : g->s_spill) b.release();
  for (DBuf* b : bufs) b->release();
  g->h_state.release();
  g->h_patch.release();
  delete g;
}

int fvdb_graph_configure(fvdb_graph* g, uint32_t max_connections, uint32_t max_connections_layer_0) {
  fvdb_ctx* ctx = g->store->ctx;
  if (max_connections == 0 || max_connections_layer_0 == 0) FAIL(ctx, FVDB_E_INVALID, "degree caps must be > 0");
  if (g->n != 0 && (max_connections != g->M || max_connections_layer_0 != g->M0))
    FAIL(ctx, FVDB_E_INVALID, "degree caps are fixed once the graph holds nodes");
  g->M = max_connections;
  g->M0 = max_connections_layer_0;
  return FVDB_OK;
}

int fvdb_graph_upload(fvdb_graph* g, uint32_t n, const uint32_t* levels, const uint8_t* deleted,
                      const uint32_t* slot_start, const uint32_t* adj, uint32_t entry_node) {
  fvdb_ctx* ctx = g->store->ctx;
  if (n == 0 || n > g->store->rows || entry_node >= n) FAIL(ctx, FVDB_E_INVALID, "graph does not match the store");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> slot_of(n), del32(n), ubase(n);
  uint32_t slots = 0, urows = 0;
  for (uint32_t i = 0; i < n; ++i) {
    slot_of[i] = slots;
    ubase[i] = urows;
    slots += levels[i] + 1;
    urows += levels[i];
    del32[i] = deleted ? deleted[i] : 0;
  }
  uint32_t max0 = 0, maxU = 0;
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t l = 0; l <= levels[i]; ++l) {
      const uint32_t c = slot_start[slot_of[i] + l + 1] - slot_start[slot_of[i] + l];
      if (c > 64) FAIL(ctx, FVDB_E_UNSUPPORTED, "neighbour list longer than 64");
      if (l == 0) max0 = std::max(max0, c);
      else maxU = std::max(maxU, c);
    }
  // fixed strides: the configured caps, or the longest list seen if that is longer (an installed graph may exceed them)
  const uint32_t stride0 = std::max(max0, g->M0) + 1, strideU = std::max(maxU, std::max(g->M, 1u)) + 1;
  // start over: the arrays are rebuilt whole (restore / bulk build / vacuum — O(n) operations themselves)
  DBuf* bufs[] = {&g->d_level, &g->d_deleted, &g->d_ubase, &g->d_adj0, &g->d_adjU, &g->d_dist0, &g->d_distU, &g->d_stamp0, &g->d_stampU};
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (DBuf* b : bufs) b->release();
  g->n = g->n_cap = g->u_rows = g->u_cap = 0;
  g->stride0 = stride0;
  g->strideU = strideU;
  int rc = reserve_nodes(g, n + n / 8 + 1024, urows + urows / 8 + 1024);
  if (rc) return rc;
  std::vector<uint32_t> adj0((size_t)n * stride0, 0u), adjU((size_t)std::max(urows, 1u) * strideU, 0u);
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t l = 0; l <= levels[i]; ++l) {
      const uint32_t a0 = slot_start[slot_of[i] + l], c = slot_start[slot_of[i] + l + 1] - a0;
      uint32_t* row = l == 0 ? &adj0[(size_t)i * stride0] : &adjU[(size_t)(ubase[i] + l - 1) * strideU];
      row[0] = c;
      for (uint32_t e = 0; e < c; ++e) {
        if (adj[a0 + e] >= n) FAIL(ctx, FVDB_E_INVALID, "neighbour index out of range");
        row[1 + e] = adj[a0 + e];
      }
    }
  HIPCHK(ctx, hipMemcpyAsync(g->d_adj0.p, adj0.data(), adj0.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (urows) HIPCHK(ctx, hipMemcpyAsync(g->d_adjU.p, adjU.data(), (size_t)urows * strideU * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(g->d_level.p, levels, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(g->d_deleted.p, del32.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(g->d_ubase.p, ubase.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  g->upload_bytes += adj0.size() * 4 + (size_t)urows * strideU * 4 + (size_t)n * 12;
  g->h_deleted.assign(n, 0);
  g->n_deleted = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (del32[i]) {
      g->h_deleted[i] = 1;
      g->n_deleted += 1;
    }
  g->h_level.assign(levels, levels + n);
  g->h_ubase = ubase;
  g->n = n;
  g->u_rows = urows;
  g->entry = entry_node;
  g->top_level = levels[entry_node];
  g->has_entry = true;
  g->uploaded = true;
  g->dist_valid = false;
  BuildState st{};
  st.has_entry = 1;
  st.entry = entry_node;
  st.entry_level = levels[entry_node];
  st.n_linked = n;
  rc = push_state(g, st);
  if (rc) return rc;
  for (auto& v : g->vis_B) v = 0;  // node count may have changed: re-size (and re-zero) the visited bitmaps
  return FVDB_OK;
}

int fvdb_graph_append_nodes(fvdb_graph* g, uint32_t first, uint32_t n_new, const uint32_t* levels) {
  fvdb_ctx* ctx = g->store->ctx;
  if (n_new == 0) return FVDB_OK;
  if (first != g->n || (uint64_t)first + n_new > g->store->rows) FAIL(ctx, FVDB_E_INVALID, "nodes are appended in store-row order");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (g->n == 0 && g->stride0 == 0) {
    g->stride0 = std::max(g->M0, 1u) + 1;
    g->strideU = std::max(g->M, 1u) + 1;
  }
  uint32_t urows = g->u_rows;
  std::vector<uint32_t> ub(n_new);
  for (uint32_t i = 0; i < n_new; ++i) {
    ub[i] = urows;
    urows += levels[i];
  }
  int rc = reserve_nodes(g, first + n_new, urows);
  if (rc) return rc;
  HIPCHK(ctx, g->h_patch.ensure((size_t)n_new * 8));
  uint32_t* hp = (uint32_t*)g->h_patch.p;
  std::memcpy(hp, levels, (size_t)n_new * 4);
  std::memcpy(hp + n_new, ub.data(), (size_t)n_new * 4);
  HIPCHK(ctx, hipMemcpyAsync(g->d_level.as<uint32_t>() + first, hp, (size_t)n_new * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(g->d_ubase.as<uint32_t>() + first, hp + n_new, (size_t)n_new * 4, hipMemcpyHostToDevice, ctx->stream));
  // empty lists, clean flags and stamps for the new rows (a vacuumed-and-reused range never occurs: rows only grow)
  HIPCHK(ctx, hipMemsetAsync(g->d_adj0.as<uint32_t>() + (size_t)first * g->stride0, 0, (size_t)n_new * g->stride0 * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(g->d_deleted.as<uint32_t>() + first, 0, (size_t)n_new * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(g->d_stamp0.as<uint32_t>() + first, 0, (size_t)n_new * 4, ctx->stream));
  if (urows > g->u_rows) {
    HIPCHK(ctx, hipMemsetAsync(g->d_adjU.as<uint32_t>() + (size_t)g->u_rows * g->strideU, 0, (size_t)(urows - g->u_rows) * g->strideU * 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(g->d_stampU.as<uint32_t>() + g->u_rows, 0, (size_t)(urows - g->u_rows) * 4, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  g->upload_bytes += (uint64_t)n_new * 8;
  g->h_level.insert(g->h_level.end(), levels, levels + n_new);
  g->h_ubase.insert(g->h_ubase.end(), ub.begin(), ub.end());
  g->h_deleted.resize(first + n_new, 0);
  g->n = first + n_new;
  g->u_rows = urows;
  if (!g->d_state.p) {
    BuildState st{};
    rc = push_state(g, st);
    if (rc) return rc;
  }
  g->uploaded = true;
  for (auto& v : g->vis_B) v = 0;
  return FVDB_OK;
}

int fvdb_graph_set_lists(fvdb_graph* g, uint32_t n_lists, const uint32_t* nodes, const uint32_t* layers, const uint32_t* offsets,
                         const uint32_t* nbrs) {
  fvdb_ctx* ctx = g->store->ctx;
  if (n_lists == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t ps = 2 + 64;
  HIPCHK(ctx, g->h_patch.ensure((size_t)n_lists * (ps + 2) * 4));
  HIPCHK(ctx, g->s_patch.ensure((size_t)n_lists * (ps + 2) * 4));
  uint32_t* hp = (uint32_t*)g->h_patch.p;
  uint32_t* codes = hp + (size_t)n_lists * ps;
  uint32_t* owner = codes + n_lists;
  for (uint32_t i = 0; i < n_lists; ++i) {
    const uint32_t node = nodes[i], layer = layers[i], c = offsets[i + 1] - offsets[i];
    if (node >= g->n || layer > g->h_level[node]) FAIL(ctx, FVDB_E_NOT_FOUND, "no such (node, layer)");
    if (c + 1 > (layer == 0 ? g->stride0 : g->strideU)) FAIL(ctx, FVDB_E_UNSUPPORTED, "list longer than the row stride");
    const uint32_t code = layer == 0 ? node : (0x80000000u | (g->h_ubase[node] + layer - 1));
    hp[(size_t)i * ps] = code;
    hp[(size_t)i * ps + 1] = c;
    for (uint32_t e = 0; e < c; ++e) {
      if (nbrs[offsets[i] + e] >= g->n) FAIL(ctx, FVDB_E_INVALID, "neighbour index out of range");
      hp[(size_t)i * ps + 2 + e] = nbrs[offsets[i] + e];
    }
    codes[i] = code;
    owner[i] = node;
  }
  const size_t bytes = (size_t)n_lists * (ps + 2) * 4;
  HIPCHK(ctx, hipMemcpyAsync(g->s_patch.p, hp, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(graph_patch_kernel, dim3((n_lists + 3) / 4), dim3(256), 0, ctx->stream, g->s_patch.as<uint32_t>(), ps, n_lists,
                     g->d_adj0.as<uint32_t>(), g->stride0, g->d_adjU.as<uint32_t>(), g->strideU);
  HIPCHK(ctx, hipGetLastError());
  g->upload_bytes += bytes;
  if (g->dist_valid) {  // keep the edge distances of the rewritten rows current
    const uint32_t* dcodes = g->s_patch.as<uint32_t>() + (size_t)n_lists * ps;
    int rc = edge_dist(g, dcodes, dcodes + n_lists, n_lists, false);
    if (rc) return rc;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FVDB_OK;
}

int fvdb_graph_set_entry(fvdb_graph* g, uint32_t entry_node, uint32_t n_linked) {
  fvdb_ctx* ctx = g->store->ctx;
  if (entry_node >= g->n || n_linked > g->n) FAIL(ctx, FVDB_E_INVALID, "no such node");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  g->entry = entry_node;
  g->top_level = g->h_level[entry_node];
  g->has_entry = true;
  BuildState st{};
  st.has_entry = 1;
  st.entry = entry_node;
  st.entry_level = g->top_level;
  st.n_linked = n_linked;
  g->upload_bytes += 16;
  return push_state(g, st);
}

int fvdb_graph_entry(fvdb_graph* g, uint32_t* entry_node, uint32_t* n_nodes) {
  if (entry_node) *entry_node = g->has_entry ? g->entry : FVDB_NO_ROW;
  if (n_nodes) *n_nodes = g->n;
  return FVDB_OK;
}

uint64_t fvdb_graph_upload_bytes(fvdb_graph* g) { return g->upload_bytes; }

int fvdb_graph_download(fvdb_graph* g, uint32_t* slot_start, uint32_t* adj, uint64_t adj_cap, uint64_t* n_edges) {
  fvdb_ctx* ctx = g->store->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> a0((size_t)g->n * g->stride0), aU((size_t)std::max(g->u_rows, 1u) * g->strideU);
  if (g->n) HIPCHK(ctx, hipMemcpyAsync(a0.data(), g->d_adj0.p, a0.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (g->u_rows) HIPCHK(ctx, hipMemcpyAsync(aU.data(), g->d_adjU.p, (size_t)g->u_rows * g->strideU * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t e = 0;
  uint32_t slot = 0;
  for (uint32_t i = 0; i < g->n; ++i)
    for (uint32_t l = 0; l <= g->h_level[i]; ++l, ++slot) {
      const uint32_t* row = l == 0 ? &a0[(size_t)i * g->stride0] : &aU[(size_t)(g->h_ubase[i] + l - 1) * g->strideU];
      if (slot_start) slot_start[slot] = (uint32_t)e;
      if (adj) {
        if (e + row[0] > adj_cap) FAIL(ctx, FVDB_E_INVALID, "adjacency buffer too small");
        std::memcpy(adj + e, row + 1, (size_t)row[0] * 4);
      }
      e += row[0];
    }
  if (slot_start) slot_start[slot] = (uint32_t)e;
  if (n_edges) *n_edges = e;
  return FVDB_OK;
}

int fvdb_graph_set_deleted(fvdb_graph* g, uint32_t node, int deleted) {
  fvdb_ctx* ctx = g->store->ctx;
  if (!g->uploaded || node >= g->n) FAIL(ctx, FVDB_E_NOT_FOUND, "no such node");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t v = deleted ? 1u : 0u;
  if (g->h_deleted[node] != (uint8_t)v) {
    g->n_deleted += v ? 1 : -1;
    g->h_deleted[node] = (uint8_t)v;
  }
  HIPCHK(ctx, hipMemcpyAsync(g->d_deleted.as<uint32_t>() + node, &v, 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  g->upload_bytes += 4;
  return FVDB_OK;
}

// =============================================================================================
// device-resident graph: construction (HNSWIndex::insert, src/hnsw/core.rs:226-378)
// =============================================================================================
int fvdb_graph_insert_linked(fvdb_graph* g, uint32_t first, uint32_t n, uint32_t ef_construction, int mode, uint32_t* n_done,
                             fvdb_graph_insert_stats* stats) {
  fvdb_store* s = g->store;
  fvdb_ctx* ctx = s->ctx;
  if (n_done) *n_done = 0;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (n == 0) return FVDB_OK;
  if (!g->uploaded || (uint64_t)first + n > g->n) FAIL(ctx, FVDB_E_INVALID, "append the nodes first");
  if (g->M == 0 || g->M0 == 0) FAIL(ctx, FVDB_E_INVALID, "fvdb_graph_configure first");
  if (g->M > 63 || g->M0 > 63 || g->stride0 > 64 || g->strideU > 64)
    FAIL(ctx, FVDB_E_UNSUPPORTED, "device insert: neighbour lists of at most 63 entries");
  if (ef_construction == 0 || ef_construction > 512) FAIL(ctx, FVDB_E_UNSUPPORTED, "device insert: ef_construction in 1..512");
  if (s->dpad > 1024) FAIL(ctx, FVDB_E_UNSUPPORTED, "device insert: at most 1024 dimensions");
  if (g->n >= 0x80000000u) FAIL(ctx, FVDB_E_UNSUPPORTED, "device insert: node index needs 31 bits");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // LDS: visited bitmap over all nodes + fixed tables; the restated candidates heap gets what is left (<= 4096 slots)
  const uint32_t words = (g->n + 31) / 32;
  const uint32_t fixed = build_lds_layout(words, ef_construction, 0).total;
  // FVDB_BUILD_LDS_LIMIT: test hook (a small value makes the graph "too large" for the on-chip bitmap)
  const uint32_t lds_max = getenv("FVDB_BUILD_LDS_LIMIT") ? (uint32_t)atoi(getenv("FVDB_BUILD_LDS_LIMIT")) : 160u * 1024u;
  if (fixed + 512 * 8 > lds_max) FAIL(ctx, FVDB_E_UNSUPPORTED, "device insert: graph too large for the on-chip visited bitmap");
  const uint32_t cand_cap = std::min<uint32_t>(4096, (lds_max - fixed) / 8 & ~1u);
  const BuildLds L = build_lds_layout(words, ef_construction, cand_cap);
  int rc = ensure_edge_dist(g);
  if (rc) return rc;
  BuildState st{};
  rc = pull_state(g, &st);
  if (rc) return rc;
  if (st.n_linked != first) FAIL(ctx, FVDB_E_INVALID, "nodes are linked in store-row order");
  st.cursor = 0;
  st.status = 0;
  st.n_valid = st.n_rerun = st.n_stopped = st.rounds = st.consumed = st.scored = st.ties = st.spec_ties = 0;
  std::memset(st.why, 0, sizeof(st.why));
  rc = push_state(g, st);
  if (rc) return rc;
  // speculation pays once an insert touches a small part of the graph (mode 0 = choose; 1 = never; 2 = always)
  static const int env_mode = getenv("FVDB_BUILD_MODE") ? atoi(getenv("FVDB_BUILD_MODE")) : 0;  // tuning aid / A-B
  static const int env_k = getenv("FVDB_BUILD_K") ? atoi(getenv("FVDB_BUILD_K")) : 0;
  static const int env_rerun = getenv("FVDB_BUILD_RERUN") ? atoi(getenv("FVDB_BUILD_RERUN")) : -1;
  if (env_mode) mode = env_mode;
  // FVDB_BUILD_STRICT=1: a speculation is dropped when ANY row it expanded changed (round-3 first form; A/B runs)
  static const uint32_t strict = getenv("FVDB_BUILD_STRICT") ? (uint32_t)atoi(getenv("FVDB_BUILD_STRICT")) : 0u;
  static const int env_kmax = getenv("FVDB_BUILD_KMAX") ? atoi(getenv("FVDB_BUILD_KMAX")) : 0;
  // below this many nodes every insert lands in every other's neighbourhood: one at a time, no speculation
  static const uint32_t seq_below = getenv("FVDB_BUILD_SEQ_BELOW") ? (uint32_t)atoi(getenv("FVDB_BUILD_SEQ_BELOW")) : 256u;
  // (a call never speculates further than it has nodes to link: the logs of a batch are 160 KB per (insert, layer) slot)
  const uint32_t Kmax = std::min<uint32_t>((uint32_t)std::max(1, std::min(env_k > 0 ? env_k : (env_kmax > 0 ? env_kmax : 128), 256)),
                                           std::max<uint32_t>(8, n));
  uint32_t K = env_k > 0 ? Kmax : std::min<uint32_t>(16, Kmax);  // adapts to the run length of adopted speculations
  // 0: the commit workgroup adopts speculated searches up to the first one an earlier insert of the batch invalidated,
  // searches that ONE itself (so every launch pair makes progress) and stops; the rest of the batch is speculated again,
  // in parallel, against the graph as it then stands
  const uint32_t max_rerun = env_rerun >= 0 ? (uint32_t)env_rerun : 0u;
  HIPCHK(ctx, g->d_spec.ensure((size_t)Kmax * kSpecWords * 4));
  HIPCHK(ctx, g->d_elog.ensure((size_t)Kmax * kBuildLayers * kLogWords * 4));
  HIPCHK(ctx, g->d_chg.ensure((size_t)kChgCap * 4 * 4));
  if (n >= 8)  // (a call that cannot speculate skips this)
    HIPCHK(ctx, hipMemsetAsync(g->d_spec.p, 0, (size_t)Kmax * kSpecWords * 4, ctx->stream));  // no stale "usable" flags
  FVDB_BUILD_SWITCH(s->dpad, {
    HIPCHK(ctx, hipFuncSetAttribute((const void*)hnsw_insert_commit_kernel<NB_, FULL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
    HIPCHK(ctx, hipFuncSetAttribute((const void*)hnsw_insert_search_kernel<NB_, FULL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
  });
  BuildView v = build_view(g, ef_construction, cand_cap);
#ifdef FVDB_BUILD_STAMPS
  static unsigned long long* d_dbg = nullptr;
  if (!d_dbg) (void)hipMalloc(&d_dbg, 64 * 8);
  (void)hipMemset(d_dbg, 0, 64 * 8);
  v.dbg = d_dbg;
#endif
  fvdb_graph_insert_stats acc{};
  uint32_t done = 0;
  uint32_t exact_positions = 4;  // speculated searches of a batch that may start again with the restated heaps on a tie
  uint32_t ties_seen = 0;
  static const int env_xf = getenv("FVDB_BUILD_EXACT_FIRST") ? atoi(getenv("FVDB_BUILD_EXACT_FIRST")) : 1;    // A/B
  // where ties are the rule (duplicate vectors) the register-set attempt is wasted work: the next `exact_left` rounds of
  // this loop go straight to the restated heaps, then the question is asked again
  uint32_t exact_left = 0, commit_ties_seen = 0;
  while (done < n) {
    const bool speculate = mode == 2 || (mode == 0 && (uint64_t)first + done >= seq_below && n - done >= 8);
    uint32_t launches = 0;
    v.exact_first = exact_left > 0 ? 1u : 0u;
    if (exact_left) exact_left -= 1;
    if (!speculate) {
      uint32_t chunk = std::min<uint32_t>(n - done, 2048);  // bounds one launch to a fraction of a second
      if (mode == 0 && (uint64_t)first + done < seq_below) chunk = std::min<uint32_t>(chunk, seq_below - (first + done));
      g->tag += 1;
      const uint32_t tag = g->tag;
      FVDB_BUILD_SWITCH(s->dpad, {
        hipLaunchKernelGGL((hnsw_insert_commit_kernel<NB_, FULL_>), dim3(1), dim3(kBuildThreads), L.total, ctx->stream, v, first, n,
                           chunk, tag, 0u, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, 1u);
      });
      launches = 1;
    } else {
      const uint32_t pairs = std::min<uint32_t>(16, (n - done + K - 1) / K);  // the cursor lives on the device: no sync in between
      for (uint32_t p = 0; p < pairs; ++p) {
        g->tag += 1;
        const uint32_t tag = g->tag;
        FVDB_BUILD_SWITCH(s->dpad, {
          hipLaunchKernelGGL((hnsw_insert_search_kernel<NB_, FULL_>), dim3(K, kBuildLayers), dim3(kBuildThreads), L.total, ctx->stream, v,
                             first, n, exact_positions, tag, g->d_spec.as<uint32_t>(), g->d_elog.as<uint32_t>());
          hipLaunchKernelGGL((hnsw_insert_commit_kernel<NB_, FULL_>), dim3(1), dim3(kBuildThreads), L.total, ctx->stream, v, first, n, K,
                             tag, max_rerun, (const uint32_t*)g->d_spec.p, (const uint32_t*)g->d_elog.p, g->d_chg.as<uint32_t>(), strict);
        });
      }
      launches = 2 * pairs;
    }
    HIPCHK(ctx, hipGetLastError());
    const uint32_t before = st.cursor;
    rc = pull_state(g, &st);
    if (rc) return rc;
    acc.launches += launches;
    if (speculate && env_k <= 0 && launches >= 2) {
      // a batch is adopted up to its first conflict: speculating much further than the usual run only adds stragglers
      // (the slowest of the batch's searches sets the launch's duration)
      const uint32_t run = (st.cursor - before) / (launches / 2);
      K = std::min<uint32_t>(Kmax, std::max<uint32_t>(8, 2 * run + 4));
    }
    if (speculate && launches >= 2) {  // where ties are the rule (duplicate vectors) every speculation takes the exact search
      const uint32_t searched = (launches / 2) * K, tied = st.spec_ties - ties_seen;
      if (!v.exact_first) exact_positions = 4 * tied > searched ? Kmax : 4;
      if (env_xf && !v.exact_first && 2 * tied > searched) exact_left = 8;
      ties_seen = st.spec_ties;
    }
    if (!speculate && env_xf && !v.exact_first && 2 * (st.ties - commit_ties_seen) > st.cursor - before) exact_left = 8;
    commit_ties_seen = st.ties;
    if (st.cursor == done && st.status == 0) FAIL(ctx, FVDB_E_HIP, "device insert made no progress");
    done = st.cursor;
    if (st.status) break;
  }
#ifdef FVDB_BUILD_STAMPS
  {
    unsigned long long h[16];
    (void)hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[build stamps] of replay+select (us): replay %.1f  select %.1f  (table = the rest) | admissions %.1f\n", h[6] * 0.01 / std::max(1u, st.n_rerun),
            h[7] * 0.01 / std::max(1u, st.n_rerun), (double)h[8] / std::max(1u, st.n_rerun));
    const double us = 0.01, nn = std::max(1u, st.n_rerun);
    fprintf(stderr, "[build stamps] per searched insert (us): replay+select %.1f  fetch %.1f  score %.1f | greedy %.1f  all searches %.1f  links %.1f"
            " | rounds %.1f expanded %.1f scored %.0f\n", h[0] * us / nn, h[1] * us / nn, h[2] * us / nn, h[3] * us / nn, h[4] * us / nn,
            h[5] * us / std::max(1u, done), st.rounds / nn, st.consumed / nn, st.scored / nn);
  }
#endif
  if (getenv("FVDB_BUILD_DEBUG"))
    fprintf(stderr, "[device insert] %u linked, %u adopted, stops %u: entry %u, gave-up %u, order/strict %u, caps %u, added node nearer than a later pop %u, added node inside the final set %u, dropped node matters %u | "
            "checks %u, touched rows %u\n", done, st.n_valid, st.n_stopped, st.why[1], st.why[2], st.why[3], st.why[4], st.why[5], st.why[6],
            st.why[9], st.why[7], st.why[8]);
  if (getenv("FVDB_BUILD_DEBUG"))
    fprintf(stderr, "[device insert] searches that left the register set: pops tied %u, evictions tied %u, result tied %u, heap overflow %u | speculations given up or restarted %u\n",
            st.why[11], st.why[12], st.why[13], st.why[14], st.spec_ties);
  if (getenv("FVDB_BUILD_DEBUG"))
    fprintf(stderr, "[device insert] second looks: overlapping %u, nodes the popped newcomer would bring in %u, later pop is the maximum %u\n", st.why[15], st.why[16], st.why[17]);
  g->entry = st.entry;
  g->top_level = st.entry_level;
  g->has_entry = st.has_entry != 0;
  acc.n_done = done;
  acc.needs_host = st.status;
  acc.speculated_ok = st.n_valid;
  acc.searched_in_commit = st.n_rerun;
  acc.commit_stops = st.n_stopped;
  acc.rounds = st.rounds;
  acc.expanded = st.consumed;
  acc.rows_scored = st.scored;
  acc.tie_restarts = st.ties;
  g->last = acc;
  if (stats) *stats = acc;
  if (n_done) *n_done = done;
  return FVDB_OK;
}

// =============================================================================================
// device-resident graph traversal
// =============================================================================================
int fvdb_graph_search_dev(fvdb_graph* g, const float* q_dev, uint32_t B, uint32_t k, uint32_t ef,
                          uint32_t* out_nodes_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                          uint32_t* out_status_dev) {
  return fvdb_graph_search_dev_slot(g, nullptr, 0, q_dev, B, k, ef, out_nodes_dev, out_dist_dev, out_counts_dev,
                                    out_status_dev);
}

int fvdb_graph_search_dev_slot(fvdb_graph* g, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t k,
                               uint32_t ef, uint32_t* out_nodes_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                               uint32_t* out_status_dev) {
  fvdb_store* s = g->store;
  fvdb_ctx* ctx = on ? on : s->ctx;
  if (slot >= fvdb_graph::kSlots) FAIL(ctx, FVDB_E_INVALID, "slot out of range");
  if (on && on->device != s->ctx->device) FAIL(ctx, FVDB_E_INVALID, "context of another device");
  if (s->d != s->dpad && slot != 0) FAIL(ctx, FVDB_E_UNSUPPORTED, "padded dimensions use slot 0 only");
  if (!g->uploaded || !g->has_entry) FAIL(ctx, FVDB_E_INVALID, "graph not uploaded");
  if (k == 0 || ef == 0 || ef > 4096) FAIL(ctx, FVDB_E_UNSUPPORTED, "ef must be in 1..4096");
  if (B == 0) return FVDB_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> lk(g->mu);
  const float* qd = q_dev;
  if (s->d != s->dpad) {
    HIPCHK(ctx, g->s_q.ensure((size_t)B * s->dpad * 4));
    hipLaunchKernelGGL(graph_pad_rows_kernel, dim3(cdiv((uint64_t)B * s->dpad, 256)), dim3(256), 0, ctx->stream, q_dev, s->d,
                       s->dpad, (uint64_t)B, g->s_q.as<float>());
    qd = g->s_q.as<float>();
  }
  // visited-log capacity per query: a query that outgrows it clears its whole map at the end of the layer instead of
  // entry by entry (FVDB_GRAPH_TCAP: test hook that forces that)
  const uint32_t words = (g->n + 31) / 32;
  const uint32_t tcap = getenv("FVDB_GRAPH_TCAP") ? std::max(1, atoi(getenv("FVDB_GRAPH_TCAP"))) : 8192;
  // visited set per query: one byte per node while a batch's maps stay under 1 GiB (no atomics, see
  // kernels_graph_fast.h), else one bit per node; the row stride is the same for both views
  static const bool no_bytes = getenv("FVDB_GRAPH_BITMAP") != nullptr;  // tuning aid / A-B (byte map: ~2.5 % faster, 8x the memory)
  const uint32_t vbytes = ((g->n + 63) / 64) * 64;
  const bool bytemap = !no_bytes && (uint64_t)vbytes * std::max<uint32_t>(B, 1024) <= (1ull << 30);
  const uint32_t vstride = bytemap ? vbytes : words * 4;
  if (words != g->vis_words || tcap != g->vis_tcap || vstride != g->vis_stride) {
    for (auto& v : g->vis_B) v = 0;
    g->vis_words = words;
    g->vis_tcap = tcap;
    g->vis_stride = vstride;
  }
  if (B > g->vis_B[slot]) {  // the maps are left all-zero by every search: zero once
    HIPCHK(ctx, g->s_visited[slot].ensure((size_t)B * vstride));
    HIPCHK(ctx, hipMemsetAsync(g->s_visited[slot].p, 0, g->s_visited[slot].cap, ctx->stream));
    HIPCHK(ctx, g->s_touched[slot].ensure((size_t)B * tcap * 4));
    g->vis_B[slot] = B;
  }
  // where the restated candidates heap continues when it outgrows its LDS slots (duplicate-heavy data): no node is
  // admitted twice, so a query never needs more than n slots
  const uint32_t spill_cap = std::min<uint32_t>(((g->n + 63) / 64) * 64, 8192);
  HIPCHK(ctx, g->s_spill[slot].ensure((size_t)B * spill_cap * 8));
  // candidate-heap slots of the exact-heap search: it holds every admitted node not yet expanded; a query that
  // overflows it goes to the host walk (data with many duplicate vectors fills it quickly, so it stays generous:
  // at the default tile size the sorted-register kernel's LDS need is larger anyway)
  const int cand_env = getenv("FVDB_GRAPH_CAND_CAP") ? atoi(getenv("FVDB_GRAPH_CAND_CAP")) : 0;  // test hook: forces the host-walk fallback
  const uint32_t cand_cap = cand_env > 0 ? (uint32_t)cand_env : std::max<uint32_t>(1024, 8 * ef);
  const size_t lds = graph_lds_bytes(s->dpad, ef, cand_cap);
  if (lds > 160 * 1024) FAIL(ctx, FVDB_E_UNSUPPORTED, "dimension / ef too large for the on-chip traversal state");
  static const bool lds_heaps = getenv("FVDB_GRAPH_LDS_HEAPS") != nullptr;  // tuning aid: lane-0 heaps for any ef
  const bool rh = ef <= 63 && !lds_heaps;
  if (lds > 48 * 1024) {
    if (rh) HIPCHK(ctx, hipFuncSetAttribute((const void*)hnsw_search_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else HIPCHK(ctx, hipFuncSetAttribute((const void*)hnsw_search_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  GraphView gv{s->data, g->d_level.as<uint32_t>(), g->d_deleted.as<uint32_t>(), g->d_adj0.as<uint32_t>(), g->d_ubase.as<uint32_t>(),
               g->d_adjU.as<uint32_t>(), g->stride0, g->strideU, g->n, s->dpad, g->entry, g->top_level, g->n_deleted ? 1u : 0u,
               nullptr, nullptr};
  if (!g->d_counters.p) {
    HIPCHK(ctx, g->d_counters.ensure(32));
    HIPCHK(ctx, hipMemsetAsync(g->d_counters.p, 0, 32, ctx->stream));
  }
  gv.counters = (unsigned long long*)g->d_counters.p;
#ifdef FVDB_GRAPH_STAMPS
  static unsigned long long* d_stamps = nullptr;
  constexpr size_t kStampWords = 8 + 3 * 16384 + 4;  // 8 sums, then per query (cycles, hops, start tick) of the last launch
  if (!d_stamps) {
    (void)hipMalloc(&d_stamps, kStampWords * 8);
    (void)hipMemset(d_stamps, 0, kStampWords * 8);
  }
  gv.stamps = d_stamps;
  {
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(kStampWords);
    (void)hipMemcpy(h.data(), d_stamps, kStampWords * 8, hipMemcpyDeviceToHost);
    fprintf(stderr, "[graph stamps, cumulative] s0 %llu s1 %llu s2 %llu s3 %llu | rows %llu rounds %llu hops %llu total %llu | score: issue %llu "
            "first-block wait+products %llu other-block products %llu adds %llu\n",
            h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8 + 3 * 16384], h[8 + 3 * 16384 + 1], h[8 + 3 * 16384 + 2], h[8 + 3 * 16384 + 3]);
    std::vector<unsigned long long> cyc, hp, rt;
    {
      // placement: HW_ID bits [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se; top nibble = XCC
      std::map<unsigned, unsigned> per_cu, per_simd;
      unsigned late = 0;
      unsigned long long first = ~0ull;
      for (uint32_t q = 0; q < 16384; ++q)
        if (h[8 + 3 * q]) first = std::min(first, h[8 + 3 * q + 1]);
      for (uint32_t q = 0; q < 16384; ++q)
        if (h[8 + 3 * q]) {
          const unsigned hw = (unsigned)((h[8 + 3 * q] >> 32) & 0x0FFFFFFFu), xcc = (unsigned)(h[8 + 3 * q] >> 60);
          const unsigned cu = (xcc << 16) | (hw & 0xFF00u);
          per_cu[cu]++;
          per_simd[(cu << 2) | ((hw >> 4) & 3)]++;
          if (h[8 + 3 * q + 1] - first > 10000) late++;
          h[8 + 3 * q] &= 0xFFFFFFFFull;
        }
      unsigned mx_cu = 0, mx_simd = 0;
      for (auto& kv : per_cu) mx_cu = std::max(mx_cu, kv.second);
      for (auto& kv : per_simd) mx_simd = std::max(mx_simd, kv.second);
      fprintf(stderr, "[graph stamps, placement] CUs used %zu (max waves on one CU %u), SIMDs used %zu (max on one %u), waves starting > 100 us late: %u\n",
              per_cu.size(), mx_cu, per_simd.size(), mx_simd, late);
    }
    for (uint32_t q = 0; q < 16384; ++q)
      if (h[8 + 3 * q]) {
        cyc.push_back(h[8 + 3 * q]);
        hp.push_back(h[8 + 3 * q + 1]);
        rt.push_back(h[8 + 3 * q + 2]);
      }
    if (!cyc.empty()) {
      double csum = 0, rsum = 0;
      for (size_t i = 0; i < cyc.size(); ++i) {
        csum += (double)cyc[i];
        rsum += (double)rt[i];
      }
      // hp = start tick (100 MHz), rt = lifetime ticks
      unsigned long long t0 = ~0ull, t1 = 0;
      for (size_t i = 0; i < cyc.size(); ++i) {
        t0 = std::min(t0, hp[i]);
        t1 = std::max(t1, hp[i] + rt[i]);
      }
      std::vector<unsigned long long> st;
      for (size_t i = 0; i < cyc.size(); ++i) st.push_back(hp[i] - t0);
      std::sort(cyc.begin(), cyc.end());
      std::sort(st.begin(), st.end());
      std::sort(rt.begin(), rt.end());
      const size_t n = cyc.size();
      fprintf(stderr, "[graph stamps, last launch] queries %zu  cycles p50 %llu max %llu | wave lifetime us p50 %.1f p99 %.1f max %.1f | clock %.2f GHz | "
              "first start .. last end %.1f us; starts us: p25 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f\n", n, cyc[n / 2], cyc[n - 1],
              rt[n / 2] / 100.0, rt[n * 99 / 100] / 100.0, rt[n - 1] / 100.0, csum / rsum / 10.0, (t1 - t0) / 100.0, st[n / 4] / 100.0,
              st[n / 2] / 100.0, st[n * 3 / 4] / 100.0, st[n * 9 / 10] / 100.0, st[n - 1] / 100.0);
    }
    (void)hipMemset(d_stamps, 0, kStampWords * 8);
  }
#endif
  hipEvent_t* ev = nullptr;
  if (s->ctx->profiling) {  // the store's context carries the switch, whichever stream the launch goes to
    ev = g->kev[g->kev_n & 63];
    if (!ev[0]) {
      (void)hipEventCreate(&ev[0]);
      (void)hipEventCreate(&ev[1]);
    }
    (void)hipEventRecord(ev[0], ctx->stream);
  }
  // ef <= 63: the sorted-register kernel; a query in which two heap members meet with equal distances is re-run by
  // the same wave with the reference's heaps restated (exact on ties)
  static const bool no_fast = getenv("FVDB_GRAPH_NO_FAST") != nullptr;  // tuning aid / A-B
  static const int fast_r = getenv("FVDB_GRAPH_FAST_R") ? atoi(getenv("FVDB_GRAPH_FAST_R")) : 0;
  const uint32_t nb128 = (s->dpad + 127) / 128;
  const bool fast = !no_fast && rh && nb128 <= 8 && g->n < 0x80000000u;
  if (fast) {
    int R = nb128 <= 3 ? 16 : (nb128 == 4 ? 12 : (nb128 <= 6 ? 8 : 6));  // rows per scoring round: registers R*NB*2
    if (nb128 == 3 && (fast_r == 8 || fast_r == 12)) R = fast_r;
    const uint32_t wave_lds = (uint32_t)((std::max(graph_fast_lds_bytes((uint32_t)R), lds) + 15) & ~(size_t)15);
#define FVDB_FAST_LAUNCH_V(NB_, R_, BY_)                                                                                   \
  do {                                                                                                                    \
    if (4 * wave_lds > 48 * 1024)                                                                                         \
      HIPCHK(ctx, hipFuncSetAttribute((const void*)hnsw_search_fast_kernel<NB_, R_, BY_>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * wave_lds)));                  \
    hipLaunchKernelGGL((hnsw_search_fast_kernel<NB_, R_, BY_>), dim3(cdiv(B, 4)), dim3(256), 4 * wave_lds, ctx->stream,   \
                       gv, qd, B, k, ef, cand_cap, wave_lds, g->s_visited[slot].as<uint8_t>(), vstride, words,            \
                       g->s_touched[slot].as<uint32_t>(), tcap, out_nodes_dev, out_dist_dev, out_counts_dev,              \
                       out_status_dev, (HItem*)g->s_spill[slot].p, spill_cap);                                            \
  } while (0)
#define FVDB_FAST_LAUNCH(NB_, R_)                  \
  do {                                             \
    if (bytemap) FVDB_FAST_LAUNCH_V(NB_, R_, true); \
    else FVDB_FAST_LAUNCH_V(NB_, R_, false);       \
  } while (0)
    if (4 * (size_t)wave_lds > 160 * 1024) FAIL(ctx, FVDB_E_UNSUPPORTED, "dimension / ef too large for the on-chip traversal state");
    switch (nb128) {
      case 1: FVDB_FAST_LAUNCH(1, 16); break;
      case 2: FVDB_FAST_LAUNCH(2, 16); break;
      case 3:
        if (R == 8) FVDB_FAST_LAUNCH(3, 8);
        else if (R == 12) FVDB_FAST_LAUNCH(3, 12);
        else FVDB_FAST_LAUNCH(3, 16);
        break;
      case 4: FVDB_FAST_LAUNCH(4, 12); break;
      case 5:
      case 6: FVDB_FAST_LAUNCH(6, 8); break;
      default: FVDB_FAST_LAUNCH(8, 6); break;
    }
#undef FVDB_FAST_LAUNCH_V
#undef FVDB_FAST_LAUNCH
  } else if (rh) {
    hipLaunchKernelGGL(hnsw_search_kernel<true>, dim3(B), dim3(64), lds, ctx->stream, gv, qd, B, k, ef, cand_cap,
                       g->s_visited[slot].as<uint32_t>(), vstride / 4, g->s_touched[slot].as<uint32_t>(), tcap, out_nodes_dev, out_dist_dev,
                       out_counts_dev, out_status_dev, (HItem*)g->s_spill[slot].p, spill_cap);
  } else {
    hipLaunchKernelGGL(hnsw_search_kernel<false>, dim3(B), dim3(64), lds, ctx->stream, gv, qd, B, k, ef, cand_cap,
                       g->s_visited[slot].as<uint32_t>(), vstride / 4, g->s_touched[slot].as<uint32_t>(), tcap, out_nodes_dev, out_dist_dev,
                       out_counts_dev, out_status_dev, (HItem*)nullptr, 0u);
  }
  if (ev) {
    (void)hipEventRecord(ev[1], ctx->stream);
    g->kev_n += 1;
  }
  HIPCHK(ctx, hipGetLastError());
  return FVDB_OK;
}

int fvdb_graph_tie_restarts(fvdb_graph* g, uint64_t* queries, uint64_t* searched_again) {
  fvdb_ctx* ctx = g->store->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());  // searches may run on several streams
  unsigned long long c[4] = {0, 0, 0, 0};
  if (g->d_counters.p) HIPCHK(ctx, hipMemcpy(c, g->d_counters.p, 32, hipMemcpyDeviceToHost));
  if (queries) *queries = g->tot_queries + c[3];
  if (searched_again) *searched_again = g->tot_again + c[2];
  return FVDB_OK;
}

int fvdb_graph_kernel_times(fvdb_graph* g, float* ms_sum, uint32_t* launches, uint64_t* rows_scored, uint64_t* hops) {
  fvdb_ctx* ctx = g->store->ctx;
  *ms_sum = 0.0f;
  *launches = 0;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long c[4] = {0, 0, 0, 0};  // rows scored, hops, queries searched again with the restated heaps, queries
  if (g->d_counters.p) {
    HIPCHK(ctx, hipMemcpy(c, g->d_counters.p, 32, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemset(g->d_counters.p, 0, 32));
  }
  g->tot_queries += c[3];
  g->tot_again += c[2];
  if (getenv("FVDB_GRAPH_DEBUG"))
    fprintf(stderr, "[graph traversal] %llu queries, %llu searched again with the restated heaps (equal distances)\n", c[3], c[2]);
  if (rows_scored) *rows_scored = c[0];
  if (hops) *hops = c[1];
  const uint32_t n = std::min<uint32_t>(g->kev_n, 64);
  for (uint32_t i = 0; i < n; ++i) {
    hipEvent_t* ev = g->kev[(g->kev_n - 1 - i) & 63];
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) {
      *ms_sum += ms;
      *launches += 1;
    }
  }
  g->kev_n = 0;
  return FVDB_OK;
}

}  // extern "C"
