// kernels_graph.h — HNSW search with the traversal resident on the GPU: one wavefront per query
// walks the graph mirrored in HBM (CSR adjacency), so a whole batch search is ONE launch instead of
// one launch + host round trip per hop.
//
// It executes HNSWIndex::search (src/hnsw/core.rs:398-467) and search_layer (:469-554) operation
// for operation: the two heaps are std::collections::BinaryHeap restated (sift_up /
// sift_down_to_bottom; `nearest` in registers with wave-parallel pushes and pops for ef <= 63, `candidates` in LDS
// with all lanes reading the ancestor chain at once — exactly the element moves of the serial routines, so the
// outcome of exact distance ties is the reference's); a popped node's neighbours are filtered
// through the visited set in list order; their distances are the reference's sequential f32 fold
// (one lane per neighbour, each lane streaming its own row from L2 after a whole-hop line prefetch); the admission
// rule (:517-531) is applied in neighbour order.  Results equal the host-side walk's bit for bit
// (tests/test_gpu_host_mirror.py::test_device_traversal_*).
//
// Since round 2 this kernel is the EXACT-TIE path: batch searches run hnsw_search_fast_kernel (kernels_graph_fast.h)
// first, and only queries in which two heap members met with equal distances (or ef > 64, or d > 1024) come here.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

namespace fvdb {

struct GraphView {
  const float* rows;          // [n][dpad] row-major vectors (fvdb_store)
  const uint32_t* level;      // [n]
  const uint32_t* deleted;    // [n] 0/1
  // Adjacency, fixed stride on every layer so that an insert rewrites only the rows it touches (kernels_graph_build.h,
  // fvdb_graph_set_lists): a row is [count, neighbours in list order ...].
  const uint32_t* adj0;       // layer 0: row of node i at adj0[i * stride0]
  const uint32_t* ubase;      // [n] first upper row of the node: layer l >= 1 lives at adjU[(ubase[i] + l - 1) * strideU]
  const uint32_t* adjU;
  uint32_t stride0, strideU;
  uint32_t n, dpad, entry, top_level;
  uint32_t any_deleted;       // 0: no node is flagged, the per-neighbour flag load is skipped
  unsigned long long* counters;  // [0] rows scored, [1] hops — summed over queries (roofline accounting)
  unsigned long long* stamps;  // diagnostic builds only (FVDB_GRAPH_STAMPS): per-phase cycle sums
};

#ifdef FVDB_GRAPH_STAMPS
#define STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define STAMP_ADD(slot, a, b) t_acc[slot] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

struct HItem {
  uint32_t node;
  float d;
};

// SearchCandidate::cmp (src/hnsw/core.rs:126-137): reversed on distance
__device__ __forceinline__ bool h_le(const HItem a, const HItem b) { return a.d >= b.d; }

__device__ __forceinline__ void h_sift_up(HItem* h, uint32_t start, uint32_t pos) {
  const HItem elt = h[pos];
  while (pos > start) {
    const uint32_t parent = (pos - 1) >> 1;
    if (h_le(elt, h[parent])) break;
    h[pos] = h[parent];
    pos = parent;
  }
  h[pos] = elt;
}
__device__ __forceinline__ void h_push(HItem* h, uint32_t& n, const HItem c) {
  h[n] = c;
  h_sift_up(h, 0, n);
  n += 1;
}
__device__ __forceinline__ HItem h_pop(HItem* h, uint32_t& n) {
  n -= 1;
  HItem item = h[n];
  if (n > 0) {
    const HItem root = h[0];
    h[0] = item;
    item = root;
    const uint32_t end = n;
    uint32_t pos = 0;
    const HItem elt = h[0];
    uint32_t child = 1;
    const uint32_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim) {
      if (h_le(h[child], h[child + 1])) child += 1;
      h[pos] = h[child];
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1) {
      h[pos] = h[child];
      pos = child;
    }
    h[pos] = elt;
    h_sift_up(h, 0, pos);
  }
  return item;
}

// ---------------------------------------------------------------------------------------------
// The same BinaryHeap operations, wave-parallel.  `nearest` never holds more than ef + 1 <= 64 items, so it can
// live in registers, lane i = heap slot i; a push or pop is then a handful of cross-lane moves instead of a chain of
// dependent LDS round trips driven by lane 0.  Each function performs exactly the element moves of the serial
// routine it replaces (the path of a sift is a sorted chain, so "shift every smaller ancestor down one level" can
// be decided for all ancestors at once), hence the same heap layout and the same tie behaviour.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rlane_f(float v, uint32_t l) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
}

// place (en, ed) at the hole `pos` and sift it up (h_sift_up with start 0)
__device__ __forceinline__ void rh_sift_up(uint32_t& hn, float& hd, uint32_t pos, uint32_t en, float ed, int lane) {
  const uint32_t x = pos + 1, y = (uint32_t)lane + 1;
  const int bx = 31 - __builtin_clz(x), by = 31 - __builtin_clz(y);
  const bool onpath = by <= bx && (x >> (bx - by)) == y;  // lane is `pos` or one of its ancestors
  const bool self = (uint32_t)lane == pos;
  const bool moves = onpath && !self && !(ed >= hd);  // !h_le(elt, ancestor): the ancestor drops one level
  const int par = lane > 0 ? (lane - 1) >> 1 : 0;
  const int pm = __shfl(moves ? 1 : 0, par);  // every lane takes part: a shuffle only sees active source lanes
  const bool pmoves = lane > 0 && pm != 0;
  const uint32_t upn = __shfl(hn, par);
  const float upd = __shfl(hd, par);
  if (onpath) {
    if (pmoves) {
      hn = upn;
      hd = upd;
    } else if (self || moves) {
      hn = en;
      hd = ed;
    }
  }
}

__device__ __forceinline__ void rh_push(uint32_t& hn, float& hd, uint32_t& n, uint32_t en, float ed, int lane) {
  rh_sift_up(hn, hd, n, en, ed, lane);
  n += 1;
}

// BinaryHeap::pop: the last item replaces the root, sift_down_to_bottom(0), then sift_up; returns the old root
__device__ __forceinline__ void rh_pop(uint32_t& hn, float& hd, uint32_t& n, int lane) {
  n -= 1;
  const uint32_t itn = __builtin_amdgcn_readlane(hn, n);
  const float itd = rlane_f(hd, n);
  if (n == 0) return;
  const uint32_t end = n;
  // larger child of every node (children are compared before anything moves, as in the serial loop)
  const uint32_t l = 2u * lane + 1, r = l + 1;
  const float dl = __shfl(hd, (int)min(l, 63u)), dr = __shfl(hd, (int)min(r, 63u));
  int big = -1;
  if (r < end) big = dl >= dr ? (int)r : (int)l;  // h_le(h[l], h[r]) picks the right child
  else if (l < end) big = (int)l;                  // a lone left child at the very end
  // the path from the root through the larger children
  bool onp = lane == 0;
  const int par = lane > 0 ? (lane - 1) >> 1 : 0;
  const int pbig = __shfl(big, par);
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const bool ponp = __shfl(onp ? 1 : 0, par) != 0;
    onp = onp || (lane > 0 && ponp && pbig == lane);
  }
  const int src = big >= 0 ? big : lane;
  const uint32_t cn = __shfl(hn, src);
  const float cd = __shfl(hd, src);
  const bool bottom_here = onp && big < 0;
  if (onp && big >= 0) {
    hn = cn;
    hd = cd;
  }
  const uint32_t bottom = (uint32_t)__builtin_ctzll(__ballot(bottom_here));
  rh_sift_up(hn, hd, bottom, itn, itd, lane);
}

// Heaps too large for registers (`candidates`) stay in LDS, but all lanes help.  Sift-up: the ancestors of the
// slot are read together, the smaller ones drop one level, the item lands above them.
//
// HeapRef: the heap array.  Slots below `cap` live in LDS; a heap that outgrows them continues in a per-query spill
// area in HBM (data with many equal distances — duplicate vectors — admits thousands of candidates that are never
// expanded; they used to send the query to the host walk).  The upper levels, where every sift passes, stay on chip.
struct HeapRef {
  HItem* lds;
  HItem* spill;   // may be null when cap is never exceeded
  uint32_t cap;   // slots in LDS
  __device__ __forceinline__ HItem get(uint32_t i) const { return i < cap ? lds[i] : spill[i - cap]; }
  __device__ __forceinline__ void set(uint32_t i, const HItem v) const {
    if (i < cap) lds[i] = v;
    else spill[i - cap] = v;
  }
};

__device__ __forceinline__ void heap_sift_up_parallel(const HeapRef h, uint32_t pos, const HItem c, int lane) {
  const uint32_t x = pos + 1;  // 1-based index of the slot
  const int depth = 31 - __builtin_clz(x);
  const bool valid = lane >= 1 && lane <= depth;  // lane j holds ancestor j (1 = parent)
  HItem a = c;
  if (valid) a = h.get((x >> lane) - 1);
  const bool moves = valid && !h_le(c, a);
  const uint64_t mb = __ballot(moves) >> 1;                        // bit j-1: ancestor j moves
  const uint32_t m = (uint32_t)__builtin_ctzll(~mb);               // they form a run from the parent up
  if (valid && (uint32_t)lane <= m) h.set((x >> (lane - 1)) - 1, a);  // ancestor j -> slot of ancestor j-1
  if (lane == 0) h.set((x >> m) - 1, c);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void heap_push_parallel(const HeapRef h, uint32_t& n, const HItem c, int lane) {
  heap_sift_up_parallel(h, n, c, lane);
  n += 1;
}

// BinaryHeap::pop: sift_down_to_bottom walks two levels per round trip (children and grandchildren of the hole are
// fetched together), then the parallel sift-up.  Returns the old root in every lane.
__device__ __forceinline__ HItem heap_pop_parallel(const HeapRef h, uint32_t& n, int lane) {
  n -= 1;
  const HItem item = h.get(n);
  if (n == 0) return item;
  const HItem root = h.get(0);
  const uint32_t end = n;
  uint32_t pos = 0;
  for (;;) {
    const uint32_t c1 = 2 * pos + 1;
    if (c1 >= end) break;  // the hole is a leaf
    const uint32_t idx = lane < 2 ? c1 + lane : 4 * pos + 3 + (lane - 2);  // lanes 2,3 / 4,5: children of c1 / c1+1
    HItem v = item;
    if (lane < 6 && idx < end) v = h.get(idx);
    const bool two1 = c1 + 1 < end;
    const uint32_t l1 = two1 && rlane_f(v.d, 0) >= rlane_f(v.d, 1) ? 1u : 0u;  // h_le(left, right): right child
    const uint32_t b1 = c1 + l1;
    const HItem cv = HItem{(uint32_t)__builtin_amdgcn_readlane(v.node, l1), rlane_f(v.d, l1)};
    if (!two1) {  // a lone left child at the very end: it moves up and the walk ends
      if (lane == 0) h.set(pos, cv);
      pos = b1;
      break;
    }
    const uint32_t g1 = 2 * b1 + 1, la = 2 + 2 * l1;
    if (g1 >= end) {
      if (lane == 0) h.set(pos, cv);
      pos = b1;
      break;
    }
    const bool two2 = g1 + 1 < end;
    const uint32_t l2 = two2 && rlane_f(v.d, la) >= rlane_f(v.d, la + 1) ? la + 1 : la;
    const HItem gv = HItem{(uint32_t)__builtin_amdgcn_readlane(v.node, l2), rlane_f(v.d, l2)};
    if (lane == 0) {
      h.set(pos, cv);
      h.set(b1, gv);
    }
    pos = g1 + (l2 - la);
    if (!two2) break;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  heap_sift_up_parallel(h, pos, item, lane);
  return root;
}

// the same on a heap that lives in LDS whole
__device__ __forceinline__ void lds_push_parallel(HItem* h, uint32_t& n, const HItem c, int lane) {
  heap_push_parallel(HeapRef{h, nullptr, 0xFFFFFFFFu}, n, c, lane);
}
__device__ __forceinline__ HItem lds_pop_parallel(HItem* h, uint32_t& n, int lane) {
  return heap_pop_parallel(HeapRef{h, nullptr, 0xFFFFFFFFu}, n, lane);
}

// Distances of the wave's query to the `np` (<= 64) rows listed in pending[], into pdist[]: one lane per row,
// the reference's left-to-right f32 fold (src/hnsw/core.rs:691-697).
//
// A lone wave per SIMD has nothing to hide HBM latency behind, and a lane's fold consumes its row strictly in
// order, so the rows are first pulled towards the core all at once: every 128-byte line of every row is touched
// by one `global_load_lds_dword` (data discarded into an LDS scratch word per lane, no VGPR held), np*12 loads in
// flight together.  Each lane then streams its own row (row-major, 16 bytes per load, 32 dims per batch, two
// batch ahead) out of L2; the query sits in LDS and is read with broadcast loads.
__device__ __forceinline__ void score_pending(const GraphView& g, const float* q_lds, float* pf_scratch,
                                              const uint32_t* pending, float* pdist, uint32_t np, int lane) {
  const uint32_t dpad = g.dpad;
  const uint32_t lpr = (dpad + 31) >> 5, nlines = np * lpr;  // 32 floats per line
  for (uint32_t l = lane; l < nlines; l += 64) {
    const uint32_t r = l / lpr, ln = l - r * lpr;
    const float* src = g.rows + (size_t)pending[r] * dpad + ln * 32;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)pf_scratch, 4, 0, 0);
  }
  if ((uint32_t)lane < np) {
    const float* row = g.rows + (size_t)pending[lane] * dpad;
    const float4* xp = (const float4*)row;
    const float4* qp = (const float4*)q_lds;  // same address in every lane: LDS broadcast reads
    const uint32_t nb = dpad >> 5;            // whole 32-dim batches
    float acc = 0.0f;
    auto issue = [&](uint32_t bi, float4 (&dst)[8]) {
      const uint32_t bb = bi < nb ? bi : nb - 1;  // past the end: re-request the last batch (keeps counts static)
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = xp[bb * 8 + i];
    };
    auto issue_q = [&](uint32_t bi, float4 (&dst)[8]) {
      const uint32_t bb = bi < nb ? bi : nb - 1;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = qp[bb * 8 + i];
    };
    auto fold = [&](const float4 (&src)[8], const float4 (&qv)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t;
        t = qv[i].x - src[i].x; acc = acc + t * t;
        t = qv[i].y - src[i].y; acc = acc + t * t;
        t = qv[i].z - src[i].z; acc = acc + t * t;
        t = qv[i].w - src[i].w; acc = acc + t * t;
      }
    };
    uint32_t done = 0;
    if (nb >= 2) {
      // rows one batch ahead (64 VGPRs of row data in flight), the query batch read from LDS as the fold starts;
      // two batches per trip so both buffer indices are static.  Kept this lean on purpose: at <= 152 VGPRs two
      // traversal waves AND a list-scan wave fit one SIMD, so the IVF chain of the same step (and the traversal of
      // the next batch in flight) run beside this kernel instead of queueing behind it.
      float4 r0[8], r1[8], qa[8];
      const uint32_t nb2 = nb - nb % 2;
      issue(0, r0);
      for (uint32_t bi = 0; bi < nb2; bi += 2) {
        issue(bi + 1, r1);
        issue_q(bi, qa);
        __builtin_amdgcn_sched_barrier(0);
        fold(r0, qa);
        __builtin_amdgcn_sched_barrier(0);
        issue(bi + 2, r0);
        issue_q(bi + 1, qa);
        __builtin_amdgcn_sched_barrier(0);
        fold(r1, qa);
        __builtin_amdgcn_sched_barrier(0);
      }
      done = nb2;
    }
    for (uint32_t bi = done; bi < nb; ++bi) {
      float4 t8[8], q8[8];
      issue(bi, t8);
      issue_q(bi, q8);
      fold(t8, q8);
    }
    for (uint32_t j = nb << 5; j < dpad; j += 4) {
      const float4 xv = *(const float4*)(row + j);
      const float4 qv = *(const float4*)(q_lds + j);
      float t;
      t = qv.x - xv.x; acc = acc + t * t;
      t = qv.y - xv.y; acc = acc + t * t;
      t = qv.z - xv.z; acc = acc + t * t;
      t = qv.w - xv.w; acc = acc + t * t;
    }
    pdist[lane] = sqrtf(acc);
  }
  __builtin_amdgcn_s_waitcnt(0);  // distances (and the discarded prefetch words) have landed in LDS
  __builtin_amdgcn_wave_barrier();
}

// LDS carve-up per wave (bytes): q [dpad*4] | prefetch scratch [64*4] | pending [64*4] | pdist [64*4] |
// scalars [16*4] | near [(ef+2)*8] | res [(ef+1)*8] | cand [cand_cap*8]
__host__ __device__ inline size_t graph_lds_bytes(uint32_t dpad, uint32_t ef, uint32_t cand_cap) {
  size_t b = (size_t)dpad * 4 + 64 * 4 + 64 * 4 + 64 * 4 + 16 * 4;
  b = (b + 7) & ~(size_t)7;
  b += (size_t)(ef + 2) * 8 + (size_t)(ef + 1) * 8 + (size_t)cand_cap * 8;
  return (b + 15) & ~(size_t)15;
}

// The whole search of query `b` by the calling wavefront, with the reference's heaps restated (exact on distance ties).
// `lds`: graph_lds_bytes(dpad, ef, cand_cap) bytes private to the wave.
// RH: `nearest` in registers + wave-parallel heap pushes (ef <= 63); otherwise both heaps in LDS, driven by lane 0.
template <bool RH>
__device__ __forceinline__ void hnsw_search_exact_body(const GraphView& g, const float* __restrict__ queries, uint32_t b, uint32_t k,
                                                       uint32_t ef_final, uint32_t cand_cap, uint32_t* __restrict__ vis /* this query's bitmap, zero on entry */,
                                                       uint32_t words, uint32_t* __restrict__ tch /* its visited log */, uint32_t tcap,
                                                       uint32_t* __restrict__ out_nodes, float* __restrict__ out_dist,
                                                       uint32_t* __restrict__ out_counts, uint32_t* __restrict__ out_status,
                                                       unsigned char* lds, int lane, HItem* __restrict__ spill = nullptr /* this query's */,
                                                       uint32_t spill_cap = 0) {
  const uint32_t dpad = g.dpad;
  float* q_lds = (float*)lds;
  float* pf_scratch = q_lds + dpad;
  uint32_t* pending = (uint32_t*)(pf_scratch + 64);
  float* pdist = (float*)(pending + 64);
  uint32_t* sc = (uint32_t*)(pdist + 64);  // [0] stop, [1] node, [2] status, [3] nN
  size_t off = ((size_t)((unsigned char*)(sc + 16) - lds) + 7) & ~(size_t)7;
  HItem* near = (HItem*)(lds + off);
  HItem* res = near + (ef_final + 2);
  HItem* cand = res + (ef_final + 1);

  for (uint32_t j = lane; j < dpad; j += 64) q_lds[j] = queries[(size_t)b * dpad + j];
  if (lane == 0) sc[2] = 0;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();

  // nearest = [(entry, dist(q, entry))]  (:432-435)
  if (lane == 0) pending[0] = g.entry;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  score_pending(g, q_lds, pf_scratch, pending, pdist, 1, lane);
  uint32_t n_res = 1;
  if (lane == 0) res[0] = HItem{g.entry, pdist[0]};
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();

#ifdef FVDB_GRAPH_STAMPS
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();
#endif
  uint32_t rows_scored = 1, hops_done = 0;  // the entry point was scored above
  for (uint32_t layer = g.top_level + 1; layer-- > 0;) {
    const uint32_t ef = layer == 0 ? ef_final : 1;
    uint32_t nC = 0, nN = 0, nT = 0;  // RH: wave-uniform; else lane 0's copies are authoritative
    uint32_t nr_node = 0;             // RH: `nearest`, lane i = heap slot i
    float nr_d = 0.0f;
    bool overflow = false;
    // ---- search_layer(query, res[0].node, ef, layer) ----
    const HItem ep = res[0];
    if (RH) {
      heap_push_parallel(HeapRef{cand, spill, cand_cap}, nC, ep, lane);
      rh_push(nr_node, nr_d, nN, ep.node, -ep.d, lane);
      if (lane == 0) {
        atomicOr(&vis[ep.node >> 5], 1u << (ep.node & 31));
        tch[0] = ep.node;
      }
    } else if (lane == 0) {
      h_push(cand, nC, ep);
      h_push(near, nN, HItem{ep.node, -ep.d});
      atomicOr(&vis[ep.node >> 5], 1u << (ep.node & 31));
      tch[0] = ep.node;
    }
    nT = 1;
    for (;;) {
      STAMP(t0s);
      if (RH) {
        uint32_t stop_r = 1, node_r = 0;
        if (nC > 0) {
          const HItem cur = heap_pop_parallel(HeapRef{cand, spill, cand_cap}, nC, lane);
          stop_r = cur.d > -rlane_f(nr_d, 0) ? 1u : 0u;  // :499-501
          node_r = cur.node;
        }
        if (lane == 0) {
          sc[0] = stop_r;
          sc[1] = node_r;
        }
      } else if (lane == 0) {
        uint32_t stop = 0, node = 0;
        if (nC == 0) {
          stop = 1;
        } else {
          const HItem cur = h_pop(cand, nC);
          if (cur.d > -near[0].d) stop = 1;  // :499-501
          node = cur.node;
        }
        sc[0] = stop;
        sc[1] = node;
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      const uint32_t stop = sc[0], node = sc[1];
      if (stop) break;
      STAMP(t1s);
      STAMP_ADD(0, t0s, t1s);
      uint32_t np = 0;
      if (layer == 0 || g.level[node] >= layer) {
        uint32_t cnt, nb = 0;
        if (layer == 0) {  // count and neighbours arrive with one load
          // lane i holds neighbour i (two independent loads, so a list of 64 neighbours fits the 64 lanes)
          const uint32_t* row = g.adj0 + (size_t)node * g.stride0;
          const uint32_t w = row[0];
          nb = (uint32_t)lane + 1 < g.stride0 ? row[lane + 1] : 0u;
          cnt = __builtin_amdgcn_readfirstlane(w);
        } else {
          const uint32_t* row = g.adjU + (size_t)(g.ubase[node] + layer - 1) * g.strideU;
          cnt = __builtin_amdgcn_readfirstlane(row[0]);  // cnt <= 64 (host checks the degree cap)
          if ((uint32_t)lane < cnt) nb = row[lane + 1];
        }
        bool fresh = false, keep = false;
        if ((uint32_t)lane < cnt) {
          const uint32_t bit = 1u << (nb & 31);
          fresh = (atomicOr(&vis[nb >> 5], bit) & bit) == 0;  // visited.insert (:506-507)
          keep = fresh && (g.any_deleted == 0 || g.deleted[nb] == 0);  // :511-513
        }
        const uint64_t fm = __ballot(fresh), km = __ballot(keep);
        const uint64_t lt = (1ull << lane) - 1;
        const uint32_t nf = __popcll(fm);
        // a log that fills up stops growing: the layer's bitmap is then cleared whole instead of entry by entry
        if (fresh && nT + nf <= tcap) tch[nT + __popcll(fm & lt)] = nb;
        nT += nf;
        np = __popcll(km);
        if (keep) pending[__popcll(km & lt)] = nb;  // list order preserved
      }
      if (overflow) break;
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      STAMP(t2s);
      STAMP_ADD(1, t1s, t2s);
      hops_done += 1;
      rows_scored += np;
      if (np) {
        score_pending(g, q_lds, pf_scratch, pending, pdist, np, lane);
        STAMP(t3s);
        STAMP_ADD(2, t2s, t3s);
#ifdef FVDB_GRAPH_STAMPS
        t_acc[4] += np;
        t_acc[5] += 1;
#endif
        // Admission rule (:517-531) in neighbour order.  Once `nearest` is full its worst distance only
        // shrinks while the neighbours are applied, so a candidate that fails against the worst as it
        // stands now can never be admitted later in this hop: all lanes test that at once and lane 0
        // walks only the survivors (in order, re-testing against the current worst).
        if (RH) {
          const float pd = (uint32_t)lane < np ? pdist[lane] : 0.0f;
          const uint32_t pn = (uint32_t)lane < np ? pending[lane] : 0u;
          float worst = -rlane_f(nr_d, 0);
          uint64_t todo = __ballot((uint32_t)lane < np && (nN < ef || pd < worst));
          while (todo) {
            const uint32_t i = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const float d = rlane_f(pd, i);
            if (d < worst || nN < ef) {
              if (nC >= cand_cap + spill_cap) {
                overflow = true;
                break;
              }
              const uint32_t node_i = __builtin_amdgcn_readlane(pn, i);
              heap_push_parallel(HeapRef{cand, spill, cand_cap}, nC, HItem{node_i, d}, lane);
              rh_push(nr_node, nr_d, nN, node_i, -d, lane);
              if (nN > ef) rh_pop(nr_node, nr_d, nN, lane);
              worst = -rlane_f(nr_d, 0);
            }
          }
          if (overflow && lane == 0) sc[2] = 1;
        } else {
          const uint32_t nN0 = __builtin_amdgcn_readfirstlane(nN);
          const float worst0 = -near[0].d;
          const bool maybe = (uint32_t)lane < np && (nN0 < ef || pdist[lane] < worst0);
          uint64_t todo = __ballot(maybe);
          if (lane == 0) {
            float worst = worst0;
            while (todo) {
              const uint32_t i = (uint32_t)__builtin_ctzll(todo);
              todo &= todo - 1;
              const float d = pdist[i];
              if (d < worst || nN < ef) {
                if (nC >= cand_cap) {
                  sc[2] = 1;
                  break;
                }
                h_push(cand, nC, HItem{pending[i], d});
                h_push(near, nN, HItem{pending[i], -d});
                if (nN > ef) (void)h_pop(near, nN);
                worst = -near[0].d;
              }
            }
          }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        STAMP(t4s);
        STAMP_ADD(3, t3s, t4s);
        if (sc[2]) {
          overflow = true;
          break;
        }
      }
#ifdef FVDB_GRAPH_STAMPS
      t_acc[6] += 1;
#endif
    }
    // ---- result of the layer: nearest in heap order, stable-sorted by distance (:541-553) ----
    if (RH) {
      if ((uint32_t)lane < nN) near[lane] = HItem{nr_node, nr_d};
      if (lane == 0) sc[3] = nN;
    } else if (lane == 0) {
      sc[3] = nN;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const uint32_t nn = sc[3];
    if (!overflow) {
      for (uint32_t i0 = 0; i0 < nn; i0 += 64) {
        const uint32_t i = i0 + lane;
        if (i < nn) {
          const HItem me = near[i];
          const float md = -me.d;
          uint32_t rank = 0;
          for (uint32_t j = 0; j < nn; ++j) {
            const float dj = -near[j].d;
            rank += (dj < md || (dj == md && j < i)) ? 1u : 0u;
          }
          res[rank] = HItem{me.node, md};
        }
      }
      n_res = nn;
    }
    // ---- drop this layer's visited set ----
    if (nT <= tcap && !overflow) {
      for (uint32_t i = lane; i < nT; i += 64) vis[tch[i] >> 5] = 0;
    } else {
      for (uint32_t w = lane; w < words; w += 64) vis[w] = 0;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (overflow) {
      if (lane == 0) {
        out_status[b] = 1;
        out_counts[b] = 0;
      }
      return;
    }
  }
#ifdef FVDB_GRAPH_STAMPS
  if (lane == 0 && g.stamps) {
    t_acc[7] = __builtin_amdgcn_s_memtime() - t_begin;
    for (int i = 0; i < 8; ++i) atomicAdd(g.stamps + i, t_acc[i]);
    if (b < 16384) {
      const unsigned long long hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
      g.stamps[8 + 3 * b] = (t_acc[7] & 0xFFFFFFFFull) | (hw << 32) | (xcc << 60);
      g.stamps[8 + 3 * b + 1] = r_begin;
      g.stamps[8 + 3 * b + 2] = __builtin_amdgcn_s_memrealtime() - r_begin;
    }
  }
#endif
  if (lane == 0 && g.counters) {
    atomicAdd(g.counters + 0, (unsigned long long)rows_scored);
    atomicAdd(g.counters + 1, (unsigned long long)hops_done);
  }
  // ---- filter deleted, take k (:451-466) ----
  if (lane == 0) {
    uint32_t w = 0;
    for (uint32_t i = 0; i < n_res && w < k; ++i) {
      const HItem c = res[i];
      if (g.deleted[c.node]) continue;
      out_nodes[(size_t)b * k + w] = c.node;
      out_dist[(size_t)b * k + w] = c.d;
      ++w;
    }
    out_counts[b] = w;
    out_status[b] = 0;
    for (uint32_t i = w; i < k; ++i) {
      out_nodes[(size_t)b * k + i] = 0xFFFFFFFFu;
      out_dist[(size_t)b * k + i] = __uint_as_float(0x7F800000u);
    }
  }
}

template <bool RH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 3))) void hnsw_search_kernel(const GraphView g, const float* __restrict__ queries,
                                                         uint32_t B, uint32_t k, uint32_t ef_final, uint32_t cand_cap,
                                                         uint32_t* __restrict__ visited /* [B][words] zero on entry */,
                                                         uint32_t words, uint32_t* __restrict__ touched /* [B][tcap] */,
                                                         uint32_t tcap, uint32_t* __restrict__ out_nodes,
                                                         float* __restrict__ out_dist, uint32_t* __restrict__ out_counts,
                                                         uint32_t* __restrict__ out_status, HItem* __restrict__ spill /* [B][spill_cap] */,
                                                         uint32_t spill_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t b = blockIdx.x;
  if (b >= B) return;
  hnsw_search_exact_body<RH>(g, queries, b, k, ef_final, cand_cap, visited + (size_t)b * words, words, touched + (size_t)b * tcap, tcap, out_nodes, out_dist, out_counts,
                             out_status, lds, (int)threadIdx.x, spill ? spill + (size_t)b * spill_cap : nullptr, spill ? spill_cap : 0u);
}

}  // namespace fvdb
