// kernels_graph_build.h — HNSWIndex::insert (src/hnsw/core.rs:226-378) resident on the GPU.
//
// One WORKGROUP (8 wavefronts, one per CU) executes one insert: the greedy descent (search_layer with ef = 1,
// :284-290), search_layer(ef_construction) on every layer of the new node (:305), select_neighbors (:556-558),
// the back-links and prune_neighbors_with_new_node (:588-624), against an adjacency that lives in HBM with a
// fixed stride per (node, layer) so that an insert rewrites only the rows it touches.  Inserts are applied
// strictly in order; the graph is the reference's node for node.
//
// Why the results cannot differ from the serial algorithm although the work is reorganised:
//
//  * Distances are pure functions of two rows (the reference's left-to-right f32 fold, computed exactly as in
//    kernels_graph_fast.h: coalesced loads, products transposed through LDS, a per-lane sequential add chain).
//    A search round therefore scores the unvisited neighbours of SEVERAL candidates at once — the kSpec nearest
//    unexpanded ones — on all eight waves, and wave 0 then REPLAYS the reference's loop (pop, visit the neighbours
//    in list order, admission rule :517-531) out of that table, with no memory latency inside the loop.  The replay
//    stops at the first candidate whose list is not in the table (an admission moved a new node to the front) and
//    the next round fetches again.  Scoring a row the serial algorithm would never have scored has no effect.
//
//  * `nearest` is a SET across the registers of wave 0 (4 registers x 64 lanes, ef <= 256) and `candidates` is implicit
//    (the unexpanded members of `nearest`): identical to the reference's two BinaryHeaps as long as the heaps' LAYOUT
//    never decides anything.  It decides only between equal keys at the top of a heap.  Two members sharing the maximum
//    when one has to leave: either may go — the search is the same unless the one that stayed is popped later or ends up
//    among the m entries select_neighbors reads (`amb`).  Two candidates sharing the minimum at a pop: if neither
//    expansion admits anything nearer, they are expanded back to back in either order (`twin_*`).  Equal distances in
//    the first m entries of the result: the stable sort would keep the heap array's order.  In the cases that are not
//    provably immaterial the layer's search starts again with the reference's heaps restated in LDS (lds_push_parallel /
//    lds_pop_parallel, kernels_graph.h), so ties resolve as in the reference.  (On near-isotropic 384-d rows ~12 % of
//    the searches meet equal fp32 distances somewhere inside the heaps; ~1 % have to start again.)
//
//  * prune_neighbors_with_new_node recomputes the distances from the neighbour to its (<= M + 1) list members.
//    Those are the same pure function: the distance of every stored edge is kept beside the adjacency row
//    (dist0 / distU, written when the edge is made; graph_edge_dist_kernel fills them for graphs that were uploaded),
//    and d(base, new) is the new node's own search result — (a-b)^2 == (b-a)^2 bit for bit — so the prune is a stable
//    rank of numbers already on hand and scores nothing.
//
//  * Several inserts are speculated at once (hnsw_insert_search_kernel: one workgroup per (insert, layer), all against
//    the same frozen graph) and committed in order by ONE workgroup (hnsw_insert_commit_kernel).  A speculated search
//    is adopted when it is provably the search the reference would run on the graph as it stands by then
//    (validate_speculation: the rows it expanded are unchanged, or every node an earlier insert of the batch added to
//    or dropped from them leaves its pops, its expansions and its result as they were); otherwise the commit kernel
//    stops and the rest of the batch is speculated again.
#pragma once
#include "kernels_graph_fast.h"

#pragma clang fp contract(off)

namespace fvdb {

constexpr int kBuildWaves = 8;                   // wavefronts per workgroup
constexpr int kBuildThreads = kBuildWaves * 64;
#ifndef FVDB_BUILD_SPEC
#define FVDB_BUILD_SPEC 16
#endif
constexpr int kSpec = FVDB_BUILD_SPEC;           // candidates whose lists are wanted on chip each round
constexpr int kSlots = 32;                       // scored adjacency lists the table holds (>= 2 kSpec)
constexpr int kBuildLayers = 16;                 // layers of one node handled on chip (level <= 15: p = 0.408^16 otherwise)
constexpr int kNearRegs = 4;                     // sorted `nearest`: 4 registers x 64 lanes => ef <= 256
constexpr uint32_t kResWords = 2 + 2 * 64;       // per layer: count, pad, 64 nodes, 64 distances
// one speculated insert in HBM: 64 header words — [1] node, [2] the entry point it started from, [8 + l] the batch tag if
// layer l's search is usable, [24 + l] expanded rows logged by layer l's search, [40 + l] 1 if that search depended on the
// ORDER of a neighbour list (equal distances met), [56 + l] the first log entry that belongs to search_layer(ef) (the ones
// before it are the descent's hops) — then the layers' results
constexpr uint32_t kSpecHdr = 80;
constexpr uint32_t kSpecWords = kSpecHdr + kBuildLayers * kResWords;
constexpr uint32_t kLogCap = 2048;               // expanded rows remembered per speculated (insert, layer): codes, then bounds
constexpr uint32_t kLogWords = 5 * kLogCap;          // row codes | bounds W | popped distances P | first-seen masks (lo, hi)
constexpr uint32_t kChgCap = 8192;               // row changes one commit launch remembers: (row code, node added, node dropped,
                                                 // its position in the row as the batch found it | 64 if unknown)
constexpr uint32_t kChkCap = 512;                // distances one validation may need
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kTileRows = 16;               // rows of a wave's product tile

// device-resident state of a graph under construction (one 64-byte record)
struct BuildState {
  uint32_t has_entry, entry, entry_level, n_linked;  // n_linked: nodes whose insert has completed
  uint32_t cursor;                                   // next node of the current fvdb_graph_insert_linked call
  uint32_t status;                                   // 0 ok; 1: node `cursor` needs the host path (on-chip heap overflow)
  uint32_t n_valid, n_rerun, n_stopped;              // speculation statistics
  uint32_t rounds, consumed, scored, ties;           // search statistics (sums)
  uint32_t spec_ties;                                // speculated searches that met equal distances
  uint32_t why[18];  // commit stops by cause: [1] another entry point, [2] a search of the batch gave up (tie / overflow),
                     // [3] changed row + order-dependent search (or strict), [4] caps, [5] a changed node within the
                     // bound, [6] ... and that bound was +inf (set never filled); [7] checks made, [8] rows touched
};

struct BuildView {
  const float* rows;  // [n][dpad]
  uint32_t dpad;
  const uint32_t* level;
  const uint32_t* deleted;
  uint32_t any_deleted;
  const uint32_t* ubase;
  uint32_t* adj0;
  float* dist0;  // distance of every layer-0 edge, same indexing as adj0 (word 0 of a row unused)
  uint32_t* adjU;
  float* distU;
  uint32_t stride0, strideU;
  uint32_t* stamp0;  // [n]  tag of the batch that last changed the node's layer-0 row
  uint32_t* stampU;  // [upper rows]
  uint32_t M, M0, ef;
  uint32_t bitmap_words;  // visited bitmap (LDS) words: covers every node index of the graph
  uint32_t cand_cap;      // restated `candidates` heap slots in LDS
  uint32_t exact_first;   // 1: skip the register-set attempt (data on which nearly every search has to start again)
  BuildState* state;
  unsigned long long* dbg;  // diagnostic builds (-DFVDB_BUILD_STAMPS): phase times in 10 ns ticks, see fvdb_graph.cpp
};

// LDS carve-up (byte offsets, 16-byte aligned)
struct BuildLds {
  uint32_t bitmap, cand, near, res, nbr, dist, misc, slist, tiles, total;
};
__host__ __device__ inline BuildLds build_lds_layout(uint32_t bitmap_words, uint32_t ef, uint32_t cand_cap) {
  const uint32_t tile_rows = kTileRows;
  BuildLds L;
  uint32_t o = 0;
  auto take = [&](uint32_t bytes) {
    const uint32_t at = o;
    o += (bytes + 15u) & ~15u;
    return at;
  };
  L.bitmap = take(bitmap_words * 4);
  L.cand = take(cand_cap * 8);
  L.near = take((ef + 2) * 8);
  L.res = take(kBuildLayers * kResWords * 4);
  L.nbr = take(kSlots * 64 * 4);
  L.dist = take(kSlots * 64 * 4);
  L.misc = take((64 + 2 * kSpec + kSlots) * 4);  // 64 control words, fetch list (nodes, slots), list counts
  L.slist = take(kSpec * 64 * 2);
  L.tiles = take(kBuildWaves * tile_rows * kFastStride * 4);
  L.total = o;
  return L;
}

#ifdef FVDB_BUILD_STAMPS
#define BSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define BSTAMP_ADD(g, slot, a, b)                                   \
  do {                                                              \
    if (threadIdx.x == 0 && (g).dbg) atomicAdd((g).dbg + (slot), (b) - (a)); \
  } while (0)
#else
#define BSTAMP(var)
#define BSTAMP_ADD(g, slot, a, b)
#endif

// misc words
enum { MS_NSPEC = 0, MS_NSCORE = 1, MS_DONE = 2, MS_TIE = 3, MS_OVER = 4, MS_CUR = 5, MS_CURD = 6, MS_CONFLICT = 7,
       MS_NLOG = 8, MS_NC = 9, MS_NN = 10, MS_ORDER = 11, MS_NT = 12, MS_NCHK = 13, MS_NCHG = 14, MS_CHGOVER = 15, MS_SEG0 = 16, MS_FNODE = 64, MS_FSLOT = 64 + kSpec, MS_CNT = 64 + 2 * kSpec };

// ---------------------------------------------------------------------------------------------
// scoring: RC rows by one wave, one product tile (see score_fixed, kernels_graph_fast.h, for the derivation)
// ---------------------------------------------------------------------------------------------
template <int NB, int RC, bool FULL>
__device__ __forceinline__ float score_tile1(const float* __restrict__ rows, uint32_t dpad, const float2 (&q2)[NB], uint32_t pn,
                                             uint32_t cnt, float* tile, int lane) {
  float2 x[RC][NB];
  const uint32_t last = cnt - 1;
#pragma unroll
  for (int r = 0; r < RC; ++r) {
    const uint32_t rr = (uint32_t)r < last ? (uint32_t)r : last;  // wave-uniform
    const uint32_t node = __builtin_amdgcn_readlane(pn, rr);
    const float* row = rows + (size_t)node * dpad;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      const uint32_t j = (uint32_t)c * 128u + 2u * (uint32_t)lane;
      if (FULL) x[r][c] = *(const float2*)(row + j);
      else x[r][c] = j < dpad ? *(const float2*)(row + j) : make_float2(0.0f, 0.0f);
    }
  }
  const uint32_t lrow = (uint32_t)lane < (uint32_t)RC ? (uint32_t)lane : (uint32_t)(RC - 1);
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const float t0 = q2[c].x - x[r][c].x, t1 = q2[c].y - x[r][c].y;
      *(float2*)(tile + (uint32_t)r * kFastStride + 2u * (uint32_t)lane) = make_float2(t0 * t0, t1 * t1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float4* p = (const float4*)(tile + lrow * kFastStride);
#pragma unroll 16
    for (int i = 0; i < 32; ++i) {
      const float4 v = p[i];
      acc = acc + v.x;
      acc = acc + v.y;
      acc = acc + v.z;
      acc = acc + v.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return sqrtf(acc);
}

// rows one wave scores per pass: 16 x NB float2 registers of row data (8 rows beyond 512 dims)
template <int NB>
struct BuildRows {
  static constexpr uint32_t kMax = NB <= 4 ? 16u : 8u;
};

// `cnt` (1..BuildRows<NB>::kMax) rows, the smallest straight-line form that holds them
template <int NB, bool FULL>
__device__ __forceinline__ float score_upto16(const float* __restrict__ rows, uint32_t dpad, const float2 (&q2)[NB], uint32_t pn,
                                              uint32_t cnt, float* tile, int lane) {
  if (NB <= 4 && cnt > 8) return score_tile1<NB, (NB <= 4 ? 16 : 8), FULL>(rows, dpad, q2, pn, cnt, tile, lane);
  if (cnt > 4) return score_tile1<NB, 8, FULL>(rows, dpad, q2, pn, cnt, tile, lane);
  return score_tile1<NB, 4, FULL>(rows, dpad, q2, pn, cnt, tile, lane);
}

// wave-wide max / min of a float, result in every lane (DPP row shifts + row broadcasts, no LDS)
__device__ __forceinline__ float wave_max_f(float v) {
  const int ninf = (int)0xFF800000u;
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x111, 0xf, 0xf, false)));  // row_shr:1
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x112, 0xf, 0xf, false)));  // row_shr:2
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x114, 0xf, 0xf, false)));  // row_shr:4
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x118, 0xf, 0xf, false)));  // row_shr:8
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x142, 0xa, 0xf, false)));  // row_bcast:15
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), 0x143, 0xc, 0xf, false)));  // row_bcast:31
  return rlane_f(v, 63);
}
__device__ __forceinline__ float wave_min_f(float v) {
  const int pinf = (int)0x7F800000u;
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(pinf, __float_as_int(v), 0x111, 0xf, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(pinf, __float_as_int(v), 0x112, 0xf, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(pinf, __float_as_int(v), 0x114, 0xf, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(pinf, __float_as_int(v), 0x118, 0xf, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(pinf, __float_as_int(v), 0x142, 0xa, 0xf, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(pinf, __float_as_int(v), 0x143, 0xc, 0xf, false)));
  return rlane_f(v, 63);
}

// lane `l` of `old` <- `value` (both wave-uniform): one compare and one select.  (v_writelane_b32 through inline asm cost
// more: the lane select has to travel in M0, the wait states around it are spelled out by hand, and the tied operand made
// the compiler copy the registers of the set around every call — ~100 instructions per admission in the replay loop.)
__device__ __forceinline__ uint32_t writelane_u(uint32_t value, uint32_t l, uint32_t old) {
  const uint32_t me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  return me == l ? value : old;
}

// the same on unsigned values (zero fill: max's identity; min = ~max(~v))
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) { return ~wave_max_u(~v); }

// ---------------------------------------------------------------------------------------------
// per-workgroup context
// ---------------------------------------------------------------------------------------------
struct BuildCtx {
  uint32_t* bitmap;
  HItem* cand;
  HItem* near;
  uint32_t* res;    // [kBuildLayers][kResWords]
  uint32_t* nbr;    // [kSlots][64]  neighbour | deleted << 31
  float* dist;      // [kSlots][64]
  uint32_t* misc;
  uint16_t* slist;  // rows to score this round: list * 64 + position
  float* tile;      // this wave's product tile
  int lane, wave;
};

__device__ __forceinline__ BuildCtx build_ctx(unsigned char* lds, const BuildLds& L) {
  const uint32_t tile_rows = kTileRows;
  BuildCtx c;
  c.lane = threadIdx.x & 63;
  c.wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // in a scalar register: `wave == 0` branches are scalar
  c.bitmap = (uint32_t*)(lds + L.bitmap);
  c.cand = (HItem*)(lds + L.cand);
  c.near = (HItem*)(lds + L.near);
  c.res = (uint32_t*)(lds + L.res);
  c.nbr = (uint32_t*)(lds + L.nbr);
  c.dist = (float*)(lds + L.dist);
  c.misc = (uint32_t*)(lds + L.misc);
  c.slist = (uint16_t*)(lds + L.slist);
  c.tile = (float*)(lds + L.tiles) + (size_t)c.wave * tile_rows * kFastStride;
  return c;
}

__device__ __forceinline__ const uint32_t* adj_row(const BuildView& g, uint32_t node, uint32_t layer) {
  return layer == 0 ? g.adj0 + (size_t)node * g.stride0 : g.adjU + (size_t)(g.ubase[node] + layer - 1) * g.strideU;
}

__device__ __forceinline__ void clear_bitmap(const BuildView& g, BuildCtx& c) {
  uint4* b4 = (uint4*)c.bitmap;
  const uint32_t n4 = (g.bitmap_words + 3) >> 2;  // the carve is padded to 16 bytes
  for (uint32_t i = threadIdx.x; i < n4; i += kBuildThreads) b4[i] = make_uint4(0, 0, 0, 0);
}

// Fetch the adjacency rows of the round's new candidates — misc[MS_FNODE + i] into table slot misc[MS_FSLOT + i],
// i < nf — drop what is visited already, list the rest for scoring.  Wave w takes entries w, w + 8, ...
// A candidate below the layer is not expanded (:503).
__device__ __forceinline__ void fetch_lists(const BuildView& g, BuildCtx& c, uint32_t nf, uint32_t layer) {
  const int lane = c.lane;
  for (uint32_t i = c.wave; i < nf; i += kBuildWaves) {
    const uint32_t node = c.misc[MS_FNODE + i], p = c.misc[MS_FSLOT + i];
    uint32_t cnt = 0, nb = 0;
    if (layer == 0 || g.level[node] >= layer) {
      const uint32_t* row = adj_row(g, node, layer);
      cnt = __builtin_amdgcn_readfirstlane(row[0]);
      if ((uint32_t)lane < cnt) nb = row[1 + lane];
    }
    bool fresh = false;
    if ((uint32_t)lane < cnt) {
      fresh = ((c.bitmap[nb >> 5] >> (nb & 31)) & 1u) == 0;
      if (g.any_deleted && g.deleted[nb]) {
        nb |= 0x80000000u;  // visited like any other neighbour, never scored (:511-513)
        fresh = false;
      }
      c.nbr[p * 64 + lane] = nb;
    }
    const uint64_t fm = __ballot(fresh);
    const uint32_t nfresh = __popcll(fm);
    uint32_t base = 0;
    if (lane == 0) {
      c.misc[MS_CNT + p] = cnt;
      if (nfresh) base = atomicAdd(&c.misc[MS_NSCORE], nfresh);
    }
    base = __builtin_amdgcn_readfirstlane(base);
    if (fresh) c.slist[base + __popcll(fm & ((1ull << lane) - 1))] = (uint16_t)(p * 64 + lane);
  }
}

// Score the listed rows on all waves; distances land in dist[slot][position].
template <int NB, bool FULL>
__device__ __forceinline__ void score_lists(const BuildView& g, BuildCtx& c, const float2 (&q2)[NB], uint32_t U) {
  if (U == 0) return;
  const uint32_t per = (U + kBuildWaves - 1) / kBuildWaves;
  const uint32_t rc = per <= 4 ? 4u : (per <= 8 ? 8u : BuildRows<NB>::kMax);
  for (uint32_t base = (uint32_t)c.wave * rc; base < U; base += kBuildWaves * rc) {
    const uint32_t cnt = min(rc, U - base);
    const uint32_t code = (uint32_t)c.lane < cnt ? (uint32_t)c.slist[base + c.lane] : 0u;
    const uint32_t pn = (uint32_t)c.lane < cnt ? (c.nbr[code] & 0x7FFFFFFFu) : 0u;
    const float d = score_upto16<NB, FULL>(g.rows, g.dpad, q2, pn, cnt, c.tile, c.lane);
    if ((uint32_t)c.lane < cnt) c.dist[code] = d;
  }
}

// ---------------------------------------------------------------------------------------------
// search_layer with ef = 1 (:284-290): with one slot in `nearest` every admission is a new minimum, the next pop
// returns it, and the loop ends at the first hop that admits nobody — a greedy walk.  In a hop the neighbours are
// tested in list order against the running minimum with `<`, so the hop's winner is the FIRST neighbour holding the
// smallest distance below the current one.  Returns (node, distance) in every thread.
// ---------------------------------------------------------------------------------------------
template <int NB, bool FULL>
__device__ __forceinline__ void greedy_layer(const BuildView& g, BuildCtx& c, const float2 (&q2)[NB], uint32_t layer, uint32_t& cur,
                                             float& cur_d, uint32_t* elog) {
  clear_bitmap(g, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicOr(&c.bitmap[cur >> 5], 1u << (cur & 31));
    c.misc[MS_FNODE] = cur;
    c.misc[MS_FSLOT] = 0;
    c.misc[MS_NSCORE] = 0;
  }
  __syncthreads();
  for (;;) {
    fetch_lists(g, c, 1, layer);
    __syncthreads();
    const uint32_t U = c.misc[MS_NSCORE];
    score_lists<NB, FULL>(g, c, q2, U);
    __syncthreads();
    if (c.wave == 0) {
      const int lane = c.lane;
      const uint32_t cnt = c.misc[MS_CNT];
      uint32_t nb = 0;
      bool keep = false;
      float d = __uint_as_float(0x7F800000u);
      if ((uint32_t)lane < cnt) {
        nb = c.nbr[lane];
        const uint32_t id = nb & 0x7FFFFFFFu;
        const uint32_t bit = 1u << (id & 31);
        const bool fresh = (c.bitmap[id >> 5] & bit) == 0;
        if (fresh) atomicOr(&c.bitmap[id >> 5], bit);  // visited.insert (:506-507)
        keep = fresh && (nb >> 31) == 0;
        if (keep) d = c.dist[lane];
      }
      // first lane holding the minimum
      const float m = wave_min_f(d);
      const uint64_t at = __ballot(keep && d == m);
      uint32_t nxt = cur;
      float nd = cur_d;
      uint32_t moved = 0;
      if (at && m < cur_d) {
        const uint32_t l = (uint32_t)__builtin_ctzll(at);
        nxt = __builtin_amdgcn_readlane(nb, l);
        nd = m;
        moved = 1;
      }
      if (lane == 0) {
        if (moved && __popcll(at) > 1) c.misc[MS_ORDER] = 1;  // two neighbours hold the minimum: the first in LIST ORDER wins
        if (elog) {
          const uint32_t nl = c.misc[MS_NLOG];
          if (nl < kLogCap) {
            elog[nl] = layer == 0 ? cur : (0x80000000u | (g.ubase[cur] + layer - 1));
            elog[kLogCap + nl] = __float_as_uint(nd);  // a neighbour farther than this changes nothing in this hop
          }
          c.misc[MS_NLOG] = nl + 1;
        }
        c.misc[MS_CUR] = nxt;
        c.misc[MS_CURD] = __float_as_uint(nd);
        c.misc[MS_DONE] = moved ? 0u : 1u;
        c.misc[MS_FNODE] = nxt;
        c.misc[MS_NSCORE] = 0;
      }
    }
    __syncthreads();
    cur = c.misc[MS_CUR];
    cur_d = __uint_as_float(c.misc[MS_CURD]);
    const uint32_t done = c.misc[MS_DONE];
    __syncthreads();
    if (done) break;
  }
}

// ---------------------------------------------------------------------------------------------
// `nearest` as a SET across kNearRegs registers of wave 0: slot s lives in register s / 64, lane s % 64, slots fill in
// admission order and an admission into the full set overwrites the slot of the maximum.  While no two equal keys meet
// at the top of a heap a priority queue is a function of its contents, so the set with a running maximum (`worst`) and
// a minimum-of-the-unexpanded reduction (the pop of `candidates`) behaves as the reference's two heaps.
// Distances are kept as their bit patterns: for values >= +0 unsigned order is float order.  A free slot holds
// (0xFFFFFFFF, +inf): "expanded" for the minimum search, equal to no finite distance.
// ---------------------------------------------------------------------------------------------
struct NearSet {
  uint32_t n[kNearRegs];  // node | expanded << 31
  uint32_t d[kNearRegs];  // distance bits
};

// write (node, distance bits) into slot `at` (all three wave-uniform)
__device__ __forceinline__ void near_write(NearSet& h, uint32_t at, uint32_t node, uint32_t dbits) {
  const uint32_t me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const bool mine = me == (at & 63);
#pragma unroll
  for (int r = 0; r < kNearRegs; ++r) {  // branch-free: the register is picked by a scalar condition folded into the lane mask
    const bool hit = mine && (at >> 6) == (uint32_t)r;
    h.n[r] = hit ? node : h.n[r];
    h.d[r] = hit ? dbits : h.d[r];
  }
}

// maximum of register r over the slots below ef, the lane that holds it, and how many slots hold it
__device__ __forceinline__ void near_reg_max(const NearSet& h, int r, uint32_t ef, int lane, uint32_t& m, uint32_t& ml, uint32_t& mc) {
  const bool in = (uint32_t)(r * 64 + lane) < ef;
  const uint32_t v = in ? h.d[r] : 0u;
  m = wave_max_u(v);
  const uint64_t at = __ballot(in && v == m);
  ml = at ? (uint32_t)__builtin_ctzll(at) : 0u;
  mc = (uint32_t)__popcll(at);
}

// ---------------------------------------------------------------------------------------------
// search_layer(query, start, ef, layer) (:469-554), whole workgroup.  Result: the first min(64, |nearest|) members in
// ascending order -> res[layer] (count, nodes, distances).
// EXACT = false: `nearest` as a register set; returns false when equal distances met where the heap layout could matter.
// EXACT = true : the reference's heaps restated in LDS; returns false when `candidates` outgrew its LDS slots.
//
// Rounds.  A table of kSlots scored adjacency lists lives in LDS (slot -> node in a register of wave 0).  Wave 0
// replays the reference's loop out of the table until the next candidate's list is missing; it then names the kSpec
// nearest unexpanded candidates, the ones not in the table yet are fetched and scored by all waves, and the replay
// goes on.  A list is fetched once: it stays in the table until its candidate is expanded or falls out of the kSpec
// nearest.
// ---------------------------------------------------------------------------------------------
template <int NB, bool FULL, bool EXACT>
__device__ __forceinline__ bool ef_search_layer(const BuildView& g, BuildCtx& c, const float2 (&q2)[NB], uint32_t layer, uint32_t start,
                                                float start_d, uint32_t* elog, uint32_t* stat) {
  const int lane = c.lane;
  const uint32_t ef = g.ef;
  clear_bitmap(g, c);
  __syncthreads();
  NearSet h;
#pragma unroll
  for (int r = 0; r < kNearRegs; ++r) {
    h.n[r] = 0xFFFFFFFFu;
    h.d[r] = 0x7F800000u;
  }
  uint32_t nN = 0, nC = 0;          // wave 0's copies are the live ones
  uint32_t slot_node = 0xFFFFFFFFu;  // wave 0, lane l < kSlots: the node whose list table slot l holds
  // non-EXACT: the maximum of the set.  While it fills: a running maximum.  Once full: per-register maxima (rm, lane
  // rl), the set's maximum is the largest of them (register wr) — an admission then re-reduces ONE register.
  uint32_t worst = 0, wr = 0;
  uint32_t rm[kNearRegs], rl[kNearRegs], rc[kNearRegs];
#pragma unroll
  for (int r = 0; r < kNearRegs; ++r) rm[r] = rl[r] = rc[r] = 0;
  auto pick_worst = [&]() {
    worst = rm[0];
    wr = 0;
#pragma unroll
    for (int r = 1; r < kNearRegs; ++r)
      if (rm[r] > worst) {
        worst = rm[r];
        wr = (uint32_t)r;
      }
  };
  if (c.wave == 0) {
    if (lane == 0) {
      c.misc[MS_SEG0] = c.misc[MS_NLOG];
      atomicOr(&c.bitmap[start >> 5], 1u << (start & 31));
      c.misc[MS_NSCORE] = 0;
      c.misc[MS_TIE] = 0;
      c.misc[MS_OVER] = 0;
    }
    if (EXACT) {
      lds_push_parallel(c.cand, nC, HItem{start, start_d}, lane);
      lds_push_parallel(c.near, nN, HItem{start, -start_d}, lane);
    } else {
      near_write(h, 0, start, __float_as_uint(start_d));
      nN = 1;
      worst = __float_as_uint(start_d);
      if (ef == 1) {
        rm[0] = worst;
        rl[0] = 0;
        rc[0] = 1;
      }
    }
  }
  uint32_t rounds = 0, consumed = 0, scored = 0;
  // Equal distances inside the heaps (set form only; wave 0).  BinaryHeap's layout decides between equal keys only when
  // two of them are at the top of a heap at once, and even then the choice is often without consequence:
  //  * `amb`: two members shared the maximum when one had to leave.  Either may be the one that left; as long as the one
  //    that stayed is never popped — it leaves too, or sits past the m entries select_neighbors reads — both choices run
  //    the same search.  (Later maxima are smaller, nothing is admitted at the old maximum: one value covers it.)
  //  * `twin_d`, `twin_left`: two candidates shared the minimum at a pop.  If neither expansion admits anything nearer
  //    than the pair, the two are expanded back to back in either order and leave the same set behind.
  uint32_t amb = 0xFFFFFFFFu, twin_d = 0xFFFFFFFFu, twin_left = 0;
  for (;;) {
    BSTAMP(t0);
    if (c.wave == 0) {
      bool tie = false, over = false, finished = false;
      uint32_t tie_why = 0;
      // ---- the reference's loop, out of the table ----
      for (;;) {
        uint32_t node, slot = 0, popd = 0;
        if (EXACT) {
          if (nC == 0) {
            finished = true;
            break;
          }
          const HItem top = c.cand[0];
          if (top.d > -c.near[0].d) {  // :499-501
            finished = true;
            break;
          }
          node = top.node;
          popd = __float_as_uint(top.d);
        } else {
          // candidates.pop(): the nearest member not yet expanded; none left = the reference's exits (:498-501)
          uint32_t key[kNearRegs], m = 0xFFFFFFFFu;
#pragma unroll
          for (int r = 0; r < kNearRegs; ++r) {
            key[r] = (h.n[r] >> 31) ? 0xFFFFFFFFu : h.d[r];
            m = min(m, key[r]);
          }
          const uint32_t mv = wave_min_u(m);
          if (mv == 0xFFFFFFFFu) {
            finished = true;
            break;
          }
          popd = mv;
          node = 0;
          uint32_t holders = 0;
#pragma unroll
          for (int r = kNearRegs - 1; r >= 0; --r) {
            const uint64_t at = __ballot(key[r] == mv);
            holders += (uint32_t)__popcll(at);
            if (at) {
              slot = (uint32_t)r * 64u + (uint32_t)__builtin_ctzll(at);
              node = __builtin_amdgcn_readlane(h.n[r], slot & 63);
            }
          }
          // two candidates share the smallest distance: which one BinaryHeap::pop returns is its layout's business
          if (holders > 2 || mv == amb || (twin_left != 0 && (mv != twin_d || holders != twin_left))) {
            tie = true;
            tie_why = 1;
            break;
          }
          if (holders == 2 && twin_left == 0) {
            twin_d = mv;
            twin_left = 2;
          }
        }
        const uint64_t hit = __ballot(slot_node == node);
        if (hit == 0) break;  // its list is not on chip: fetch
        const uint32_t p = (uint32_t)__builtin_ctzll(hit);
        slot_node = writelane_u(0xFFFFFFFFu, p, slot_node);  // the slot is free again
        if (EXACT) {
          (void)lds_pop_parallel(c.cand, nC, lane);
        } else {
#pragma unroll
          for (int r = 0; r < kNearRegs; ++r) {
            const bool hit = (uint32_t)lane == (slot & 63) && (slot >> 6) == (uint32_t)r;
            h.n[r] = hit ? (node | 0x80000000u) : h.n[r];
          }
        }
        consumed += 1;
        const uint32_t cnt = c.misc[MS_CNT + p];
        uint32_t nb = 0;
        bool keep = false;
        uint32_t db = 0x7F800000u;
        if ((uint32_t)lane < cnt) {
          nb = c.nbr[p * 64 + lane];
          const uint32_t id = nb & 0x7FFFFFFFu;
          const uint32_t bit = 1u << (id & 31);
          const bool fresh = (c.bitmap[id >> 5] & bit) == 0;
          if (fresh) atomicOr(&c.bitmap[id >> 5], bit);  // visited.insert (:506-507)
          keep = fresh && (nb >> 31) == 0;                // deleted: visited, skipped (:511-513)
          if (keep) db = __float_as_uint(c.dist[p * 64 + lane]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint64_t seen = __ballot(keep);
        // admission (:517-531) in list order; `worst` only shrinks while they are applied
        if (EXACT) worst = __float_as_uint(-c.near[0].d);
        uint64_t todo = __ballot(keep && (nN < ef || db < worst));
        while (todo) {
          const uint32_t i = (uint32_t)__builtin_ctzll(todo);
          todo &= todo - 1;
          const uint32_t di = __builtin_amdgcn_readlane(db, i);
          if (!(nN < ef || di < worst)) continue;
          if (!EXACT && twin_left != 0 && di < twin_d) {  // the pair is not expanded back to back: the order matters
            tie = true;
            tie_why = 1;
            break;
          }
          const uint32_t ni = __builtin_amdgcn_readlane(nb, i);
#ifdef FVDB_BUILD_STAMPS
          if (threadIdx.x == 0 && g.dbg) atomicAdd(g.dbg + 8, 1ull);
#endif
          if (EXACT) {
            if (nC >= g.cand_cap) {
              over = true;
              break;
            }
            const float df = __uint_as_float(di);
            lds_push_parallel(c.cand, nC, HItem{ni, df}, lane);
            lds_push_parallel(c.near, nN, HItem{ni, -df}, lane);
            if (nN > ef) (void)lds_pop_parallel(c.near, nN, lane);
            worst = __float_as_uint(-c.near[0].d);
          } else {
            if (nN < ef) {
              near_write(h, nN, ni, di);
              nN += 1;
              worst = max(worst, di);
              if (nN == ef) {  // full from here on: per-register maxima
#pragma unroll
                for (int r = 0; r < kNearRegs; ++r) near_reg_max(h, r, ef, lane, rm[r], rl[r], rc[r]);
                pick_worst();
              }
            } else {  // the maximum leaves (nearest.pop(), :528-530), the newcomer takes its slot
              uint32_t holders = 0;
#pragma unroll
              for (int r = 0; r < kNearRegs; ++r) holders += rm[r] == worst ? rc[r] : 0u;
              uint32_t er = wr, el_ = rl[wr];  // the slot that is given up
              if (holders > 1) {
                // two members share the largest distance: which one leaves is the layout's business.  An EXPANDED one is
                // sent away if there is one: whichever the reference keeps, a survivor that can still be popped is then
                // in this set too, and its pop (if it comes to that) is seen and ends the attempt.
                amb = worst;
#pragma unroll
                for (int r = kNearRegs - 1; r >= 0; --r) {
                  const bool in = (uint32_t)(r * 64 + lane) < ef;
                  const uint64_t ex = __ballot(in && h.d[r] == worst && (h.n[r] >> 31) != 0);
                  if (ex) {
                    er = (uint32_t)r;
                    el_ = (uint32_t)__builtin_ctzll(ex);
                  }
                }
              }
              near_write(h, er * 64u + el_, ni, di);
#pragma unroll
              for (int r = 0; r < kNearRegs; ++r)
                if (er == (uint32_t)r) near_reg_max(h, r, ef, lane, rm[r], rl[r], rc[r]);
              pick_worst();
            }
          }
        }
        if (tie || over) break;
        // the expanded row and the bound its neighbours met: with `nearest` full, a neighbour farther than the maximum
        // left by this expansion was (or would be) turned away — whether the row holds it or not changes nothing
        if (elog && lane == 0) {
          const uint32_t nl = c.misc[MS_NLOG];
          if (nl < kLogCap) {
            elog[nl] = layer == 0 ? node : (0x80000000u | (g.ubase[node] + layer - 1));
            elog[kLogCap + nl] = nN >= ef ? worst : 0x7F800000u;
            elog[2 * kLogCap + nl] = popd | (twin_left != 0 ? 0x80000000u : 0u);  // flag: see validate_speculation
            elog[3 * kLogCap + nl] = (uint32_t)seen;  // list positions whose node was met for the first time here
            elog[4 * kLogCap + nl] = (uint32_t)(seen >> 32);
          }
          c.misc[MS_NLOG] = nl + 1;
        }
        if (twin_left != 0) twin_left -= 1;
      }
      BSTAMP(ta);
      BSTAMP_ADD(g, 6, t0, ta);
      // ---- the kSpec nearest unexpanded candidates; those without a list in the table are fetched ----
      uint32_t nf = 0;
      if (!tie && !over && !finished) {
        uint32_t spec_l = 0xFFFFFFFFu;  // lane i < ns: the i-th named candidate
        uint32_t ns = 0;
        if (EXACT) {
          // the front of the heap array: its root is the next pop for certain, the levels below it are likely followers
          const float w = -c.near[0].d;
          const uint32_t lim = min((uint32_t)kSpec, nC);
          HItem it = HItem{0, 0.0f};
          if ((uint32_t)lane < lim) it = c.cand[lane];
          const bool ok = (uint32_t)lane < lim && !(it.d > w);
          uint64_t om = __ballot(ok);
          while (om) {
            const uint32_t l = (uint32_t)__builtin_ctzll(om);
            om &= om - 1;
            spec_l = writelane_u(__builtin_amdgcn_readlane(it.node, l), ns, spec_l);
            ns += 1;
          }
        } else {
          // all unexpanded members with distance <= t, t = a threshold (bisection on the bit pattern) that names at most
          // kSpec of them — and at least half as many, then it is good enough: which lists are brought on chip ahead of
          // their pop changes nothing but the number of rounds; the minimum is always among them
          uint32_t key[kNearRegs], m = 0xFFFFFFFFu;
#pragma unroll
          for (int r = 0; r < kNearRegs; ++r) {
            key[r] = (h.n[r] >> 31) ? 0xFFFFFFFFu : h.d[r];
            m = min(m, key[r]);
          }
          auto count_le = [&](uint32_t t) {
            uint32_t cn = 0;
#pragma unroll
            for (int r = 0; r < kNearRegs; ++r) cn += __popcll(__ballot(key[r] <= t));
            return cn;
          };
          uint32_t lo = wave_min_u(m), hi = worst;
          uint32_t t = hi;
          if (count_le(hi) > (uint32_t)kSpec) {
            for (int it = 0; it < 14 && hi - lo > 1; ++it) {
              const uint32_t mid = lo + ((hi - lo) >> 1);
              const uint32_t cn = count_le(mid);
              if (cn <= (uint32_t)kSpec) lo = mid;
              else hi = mid;
              if (cn <= (uint32_t)kSpec && 2 * cn >= (uint32_t)kSpec) break;
            }
            t = lo;
          }
#pragma unroll
          for (int r = 0; r < kNearRegs; ++r) {
            uint64_t om = __ballot(key[r] <= t);
            while (om && ns < (uint32_t)kSpec) {
              const uint32_t l = (uint32_t)__builtin_ctzll(om);
              om &= om - 1;
              spec_l = writelane_u(__builtin_amdgcn_readlane(h.n[r], l), ns, spec_l);
              ns += 1;
            }
          }
        }
        BSTAMP(tb);
        BSTAMP_ADD(g, 7, ta, tb);
        // table slots: keep the lists of named candidates, hand the other slots to the named ones without a list
        uint64_t in_s = 0, miss = 0;
        for (uint32_t i = 0; i < ns; ++i) {
          const uint32_t sv = __builtin_amdgcn_readlane(spec_l, i);
          const uint64_t at = __ballot(slot_node == sv);
          if (at) in_s |= at;
          else miss |= 1ull << i;
        }
        uint64_t free_s = ~in_s & ((1ull << kSlots) - 1);
        while (miss) {
          const uint32_t i = (uint32_t)__builtin_ctzll(miss);
          miss &= miss - 1;
          const uint32_t sl = (uint32_t)__builtin_ctzll(free_s);  // kSlots >= 2 kSpec: never exhausted
          free_s &= free_s - 1;
          const uint32_t sv = __builtin_amdgcn_readlane(spec_l, i);
          slot_node = writelane_u(sv, sl, slot_node);
          if (lane == 0) {
            c.misc[MS_FNODE + nf] = sv;
            c.misc[MS_FSLOT + nf] = sl;
          }
          nf += 1;
        }
      }
      if (lane == 0) {
        c.misc[MS_NSPEC] = nf;
        c.misc[MS_NSCORE] = 0;
        if (tie) c.misc[MS_TIE] = tie_why;
        if (over) c.misc[MS_OVER] = 1;
      }
    }
    BSTAMP(t1);
    __syncthreads();
    const uint32_t nf = c.misc[MS_NSPEC];
    if (nf == 0) break;  // search finished, or a tie / overflow was met
    fetch_lists(g, c, nf, layer);
    __syncthreads();
    BSTAMP(t2);
    const uint32_t U = c.misc[MS_NSCORE];
    score_lists<NB, FULL>(g, c, q2, U);
    __syncthreads();
    BSTAMP(t3);
    BSTAMP_ADD(g, 0, t0, t1);
    BSTAMP_ADD(g, 1, t1, t2);
    BSTAMP_ADD(g, 2, t2, t3);
    rounds += 1;
    scored += U;
  }
  const bool failed = (c.misc[MS_TIE] | c.misc[MS_OVER]) != 0;
  if (failed && threadIdx.x == 0) atomicAdd(&g.state->why[c.misc[MS_OVER] ? 14 : 10 + min(c.misc[MS_TIE], 2u)], 1u);
  __syncthreads();
  if (threadIdx.x == 0 && stat) {
    stat[0] += rounds;
    stat[1] += consumed;
    stat[2] += scored;
  }
  if (failed) return false;
  // ---- result: `nearest` in heap order, stable-sorted by distance (:541-553); the first 64 are enough ----
  uint32_t* out = c.res + layer * kResWords;
  if (c.wave == 0) {
    if (!EXACT) {  // the set, any order (no two distances are equal), in the heap array's form (negated distances)
#pragma unroll
      for (int r = 0; r < kNearRegs; ++r)
        if ((uint32_t)(r * 64 + lane) < nN) c.near[r * 64 + lane] = HItem{h.n[r] & 0x7FFFFFFFu, -__uint_as_float(h.d[r])};
    }
    if (lane == 0) {
      c.misc[MS_NN] = nN;
      if (!EXACT && worst == amb && nN <= (layer == 0 ? g.M0 : g.M) + 1) c.misc[MS_TIE] = 3;
    }
  }
  __syncthreads();
  const uint32_t nn = c.misc[MS_NN];
  for (uint32_t i = threadIdx.x; i < nn; i += kBuildThreads) {
    const HItem me = c.near[i];
    const float md = -me.d;
    uint32_t rank = 0;
    bool twin = false;
    for (uint32_t j = 0; j < nn; ++j) {
      const float dj = -c.near[j].d;
      rank += (dj < md || (dj == md && j < i)) ? 1u : 0u;
      twin = twin || (dj == md && j != i);
    }
    if (rank < 64) {
      out[2 + rank] = me.node;
      out[2 + 64 + rank] = __float_as_uint(md);
    }
    // the stable sort leaves equal distances in the heap array's order, which the set does not have
    // (only the first m entries are ever used: select_neighbors)
    if (!EXACT && twin && rank <= (layer == 0 ? g.M0 : g.M)) c.misc[MS_TIE] = 3;
  }
  if (threadIdx.x == 0) out[0] = min(nn, 64u);
  __syncthreads();
  if (!EXACT && c.misc[MS_TIE] != 0) {
    if (threadIdx.x == 0) atomicAdd(&g.state->why[10 + 3], 1u);
    return false;
  }
  return true;
}

// search_layer(ef) with the tie rule: sorted registers first, the restated heaps when two members tie.
// false: the restated `candidates` heap outgrew its LDS slots (host path for this node).
template <int NB, bool FULL>
__device__ __forceinline__ bool ef_search(const BuildView& g, BuildCtx& c, const float2 (&q2)[NB], uint32_t layer, uint32_t start,
                                          float start_d, uint32_t* elog, uint32_t* stat, bool exact_on_tie) {
  const uint32_t log0 = c.misc[MS_NLOG];
  if (g.ef <= (uint32_t)kNearRegs * 64u && !g.exact_first) {
    if (ef_search_layer<NB, FULL, false>(g, c, q2, layer, start, start_d, elog, stat)) return true;
    if (threadIdx.x == 0 && stat) stat[3] += 1;
    if (!exact_on_tie) return false;  // a speculation far down the batch: not worth twice the time of the others
    if (threadIdx.x == 0) c.misc[MS_NLOG] = log0;  // the aborted attempt expanded a prefix of what the exact run expands
    __syncthreads();
  }
  if (threadIdx.x == 0) c.misc[MS_ORDER] = 1;  // heaps with equal keys: the outcome depends on the order of the lists
  return ef_search_layer<NB, FULL, true>(g, c, q2, layer, start, start_d, elog, stat);
}

// ---------------------------------------------------------------------------------------------
// All searches of one insert (:262-305): distance to the entry point, greedy descent from min(level, entry_level) down
// to layer 0, then search_layer(ef) on layers 0..level, each starting at the descent's final node (or at the entry
// point above its level).  Linking on layer l touches layer-l rows only, and the search on layer l + 1 reads layer
// l + 1 rows only, so all of an insert's searches can run before any of its links are made.
// Results -> res[layer]; false = host path needed.
// ---------------------------------------------------------------------------------------------
template <int NB, bool FULL>
__device__ __forceinline__ bool insert_searches(const BuildView& g, BuildCtx& c, uint32_t node, uint32_t level, uint32_t entry,
                                                uint32_t entry_level, uint32_t* elog, uint32_t* stat, bool exact_on_tie = true,
                                                int only_layer = -1 /* >= 0: the descent + that layer's search only */) {
  float2 q2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const uint32_t j = (uint32_t)b * 128u + 2u * (uint32_t)c.lane;
    q2[b] = j < g.dpad ? *(const float2*)(g.rows + (size_t)node * g.dpad + j) : make_float2(0.0f, 0.0f);
  }
  if (threadIdx.x == 0) {
    c.misc[MS_NLOG] = 0;
    c.misc[MS_NSCORE] = 0;
    c.misc[MS_ORDER] = 0;
  }
  // d(q, entry) (:277-281)
  if (c.wave == 0) {
    const float d0 = score_tile1<NB, 4, FULL>(g.rows, g.dpad, q2, entry, 1, c.tile, c.lane);
    if (c.lane == 0) c.misc[MS_CURD] = __float_as_uint(d0);
  }
  __syncthreads();
  const float entry_d = __uint_as_float(c.misc[MS_CURD]);
  __syncthreads();
  uint32_t cur = entry;
  float cur_d = entry_d;
  const uint32_t search_level = min(level, entry_level);
  BSTAMP(tg0);
  // the descent's expansions are logged once per insert: by the whole-insert call, or by the layer-0 call
  if (only_layer < 0 || (uint32_t)only_layer <= search_level)
    for (uint32_t lc = search_level + 1; lc-- > 0;) greedy_layer<NB, FULL>(g, c, q2, lc, cur, cur_d, only_layer <= 0 ? elog : nullptr);
  BSTAMP(tg1);
  BSTAMP_ADD(g, 3, tg0, tg1);
  for (uint32_t lc = 0; lc <= level; ++lc) {
    if (only_layer >= 0 && lc != (uint32_t)only_layer) continue;
    const bool low = lc <= search_level;
    if (!ef_search<NB, FULL>(g, c, q2, lc, low ? cur : entry, low ? cur_d : entry_d, elog, stat, exact_on_tie)) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Links of one insert (:293-362) from res[]: the new node's rows, the back-links, the prunes.  `tag`: this batch.
// ---------------------------------------------------------------------------------------------
// `chg` (commit of a speculated batch): every stamped row change is remembered as (row code, node added, node dropped) so
// that later speculations of the batch can tell whether the change matters to them (validate_speculation).
__device__ __forceinline__ void insert_links(const BuildView& g, BuildCtx& c, uint32_t node, uint32_t level, uint32_t tag, uint32_t* chg) {
  const int lane = c.lane;
  auto record = [&](uint32_t code, uint32_t added, uint32_t dropped, uint32_t pos) {  // one lane
    if (!chg) return;
    const uint32_t at = atomicAdd(&c.misc[MS_NCHG], 1u);
    if (at < kChgCap) {
      chg[4 * at] = code;
      chg[4 * at + 1] = added;
      chg[4 * at + 2] = dropped;
      chg[4 * at + 3] = pos;
    } else {
      c.misc[MS_CHGOVER] = 1;
    }
  };
  for (uint32_t lc = 0; lc <= level; ++lc) {
    const uint32_t* r = c.res + lc * kResWords;
    const uint32_t m = lc == 0 ? g.M0 : g.M;
    const uint32_t take = min(r[0], m);  // select_neighbors: the first m (:556-558)
    const uint32_t stride = lc == 0 ? g.stride0 : g.strideU;
    const size_t mine = lc == 0 ? (size_t)node * g.stride0 : (size_t)(g.ubase[node] + lc - 1) * g.strideU;
    uint32_t* adj = lc == 0 ? g.adj0 : g.adjU;
    float* adjd = lc == 0 ? g.dist0 : g.distU;
    if (c.wave == 0) {
      if ((uint32_t)lane < take) {
        adj[mine + 1 + lane] = r[2 + lane];
        adjd[mine + 1 + lane] = __uint_as_float(r[2 + 64 + lane]);
      }
      if (lane == 0) adj[mine] = take;
    }
    // Back-links: neighbour t belongs to wave t % 8 (distinct neighbours, distinct rows: no order between them).  A wave
    // first REQUESTS everything its (up to 8) neighbours need — level / row index, count, row, edge distances, stamp —
    // and only then works through them: one memory latency per layer instead of three per neighbour.
    constexpr int kPer = (63 + kBuildWaves - 1) / kBuildWaves;
    uint32_t p_nb[kPer], p_row[kPer], p_k[kPer], p_en[kPer], p_st[kPer];
    float p_dn[kPer], p_ed[kPer];
    bool p_on[kPer];
    uint32_t* stamps = lc == 0 ? g.stamp0 : g.stampU;
#pragma unroll
    for (int s = 0; s < kPer; ++s) {
      const uint32_t t = (uint32_t)c.wave + (uint32_t)s * kBuildWaves;
      p_on[s] = t < take;
      p_nb[s] = p_on[s] ? r[2 + t] : 0u;
      p_dn[s] = p_on[s] ? __uint_as_float(r[2 + 64 + t]) : 0.0f;
      p_row[s] = p_nb[s];
    }
    if (lc > 0) {
#pragma unroll
      for (int s = 0; s < kPer; ++s)
        if (p_on[s]) {
          p_on[s] = g.level[p_nb[s]] >= lc;  // :323 (a start node taken from a lower layer)
          p_row[s] = g.ubase[p_nb[s]] + lc - 1;
        }
    }
#pragma unroll
    for (int s = 0; s < kPer; ++s) {
      p_k[s] = p_en[s] = p_st[s] = 0;
      p_ed[s] = 0.0f;
      if (p_on[s]) {
        const size_t at = (size_t)p_row[s] * stride;
        p_k[s] = adj[at];
        if ((uint32_t)lane + 1 < stride) {
          p_en[s] = adj[at + 1 + lane];
          p_ed[s] = adjd[at + 1 + lane];
        }
        p_st[s] = stamps[p_row[s]];
      }
    }
#pragma unroll
    for (int s = 0; s < kPer; ++s) {
      if (!p_on[s]) continue;
      const uint32_t nbv = p_nb[s], row = p_row[s];
      const float dn = p_dn[s];
      const size_t at = (size_t)row * stride;
      const uint32_t code = lc == 0 ? nbv : (0x80000000u | row);
      const uint32_t k = __builtin_amdgcn_readfirstlane(p_k[s]);
      if (k < m && k + 1 < stride) {  // room: append (:324)
        if (lane == 0) {
          adj[at + 1 + k] = node;
          adjd[at + 1 + k] = dn;
          adj[at] = k + 1;
          stamps[row] = tag;
          record(code, node, kNone, 64u);
        }
        continue;
      }
      // prune_neighbors_with_new_node (:588-624): the list + the new node, stable sort by distance, keep m
      const uint32_t tot = k + 1;  // <= 64: the row stride bounds k
      const uint32_t e_n = (uint32_t)lane < k ? p_en[s] : node;
      const float e_d = (uint32_t)lane < k ? p_ed[s] : dn;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < tot; ++j) {
        const float dj = rlane_f(e_d, j);
        rank += (dj < e_d || (dj == e_d && j < (uint32_t)lane)) ? 1u : 0u;
      }
      const bool in = (uint32_t)lane < tot;
      const bool same = __ballot(in && (uint32_t)lane < k && rank != (uint32_t)lane) == 0 && k == m;  // old list kept as it was
      if (!same) {
        if (in && rank < m) {
          adj[at + 1 + rank] = e_n;
          adjd[at + 1 + rank] = e_d;
        }
        const uint32_t added = __builtin_amdgcn_readlane(rank, k) < m ? node : kNone;  // lane k holds the new node
        uint64_t out = __ballot((uint32_t)lane < k && rank >= m);
        // the row as this batch's speculations read it: positions are still theirs if nobody of the batch rewrote it
        const bool untouched = __builtin_amdgcn_readfirstlane(p_st[s]) != tag;
        if (lane == 0) {
          adj[at] = min(tot, m);
          stamps[row] = tag;
        }
        bool first = true;
        do {  // (members only moved: nothing to remember — a search that depends on list order takes the stamp alone)
          uint32_t dropped = kNone, pos = 64;
          if (out) {
            pos = (uint32_t)__builtin_ctzll(out);
            dropped = __builtin_amdgcn_readlane(e_n, pos);
            if (!untouched) pos = 64;
            out &= out - 1;
          }
          if (lane == 0 && (dropped != kNone || (first && added != kNone))) record(code, first ? added : kNone, dropped, pos);
          first = false;
        } while (out);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Is a search speculated against an older graph still the search the reference would run now?
//
// While no two distances inside the heaps are equal, search_layer is a function of the neighbour SETS of the rows it
// expands: with A = everything admitted so far, `nearest` is the ef smallest of A, the next pop is its nearest member
// not yet expanded, the search ends when there is none (`visited` only spares work: what was turned away once is
// farther than every later maximum).  The speculated search logged, per expansion t: the row e_t, the distance P_t of
// the candidate popped, and W_t = the maximum of `nearest` after the expansion (+inf while it is not full).
//
// Let an earlier insert of this batch have added x to (or dropped x from) row e_k.  With x admitted at step k the set is
// the same except that x displaces the current maximum; that lasts until the first t_out >= k with W_t < d(q, x), when
// x itself is the element that leaves.  In between, both runs pop the same candidate as long as that candidate is nearer
// than x (the displaced maximum and x are both farther, so neither is the pop).  Hence, if
//     t_out exists   and   P_t < d(q, x) for every k < t <= t_out,
// every pop, every expansion and the set after t_out are the same with or without x — by induction over all such x
// (several at once displace the top few, which are all farther than the pop; a dropped x that was in the set is the
// mirror image: the logged run is the one "with x", and the condition says it never expanded x and let it go).  A greedy
// hop (ef = 1) is the case t_out = k: the hop ended on distance W_k and x is farther.  Anything else is a conflict:
// x within the final set, x nearer than a later pop, ANY change to an expanded row when the search met equal distances
// (list order then matters: the speculation says so in its header), and any change at all in `strict` mode.
//
// Two second looks (layer 0, x alone inside the set meanwhile), both for an x that IS popped before it can leave:
//  * dropped x that the logged run went on to expand: harmless if the search, no longer meeting x at e_k, meets it in
//    the row of a later expansion that comes before x's own — x then enters there, nothing was popped differently;
//  * added x that the new run pops (a later logged pop is farther): harmless if that expansion admits nobody — every
//    node of x's row is one the search has met by then (rows of the expansions so far, marked in the visited bitmap),
//    is soft-deleted, or is not nearer than the maximum of the moment — and no later pop is the set's maximum (the
//    element x displaces must not be the one the logged run expands) until W drops below d(q, x).
//
// Returns 0 (adopt) or the cause.  Uses the searches' scratch (dist / nbr / slist / cand), idle between two inserts.
// ---------------------------------------------------------------------------------------------
template <int NB, bool FULL>
__device__ __forceinline__ uint32_t validate_speculation(const BuildView& g, BuildCtx& c, uint32_t node, uint32_t level, uint32_t tag,
                                                         const uint32_t* __restrict__ sp, const uint32_t* __restrict__ elog, const uint32_t* chg,
                                                             uint32_t strict /* A/B and bisecting: 1 any change is a conflict, 2 no second
                                                         looks, 4 no first-seen skip, 8 x must be turned away at once */) {
  uint32_t* t_code = (uint32_t*)c.dist;            // touched rows of the log: code, position (layer * kLogCap + entry)
  uint32_t* t_ref = (uint32_t*)c.dist + kChkCap;
  uint32_t* k_ref = (uint32_t*)c.cand;             // checks: node in nbr[i], log position here (cand_cap >= 512 items)
  if (threadIdx.x == 0) {
    c.misc[MS_CONFLICT] = 0;
    c.misc[MS_NT] = 0;
    c.misc[MS_NCHK] = 0;
  }
  __syncthreads();
  const bool coarse = (strict & 1u) || chg == nullptr || c.misc[MS_CHGOVER] != 0;
  for (uint32_t l = 0; l <= level; ++l) {
    const uint32_t nlog = sp[24 + l];
    const bool order = sp[40 + l] != 0;
    const uint32_t* el = elog + (size_t)l * kLogWords;
    for (uint32_t i = threadIdx.x; i < nlog; i += kBuildThreads) {
      const uint32_t code = el[i];
      const uint32_t sv = (code >> 31) ? g.stampU[code & 0x7FFFFFFFu] : g.stamp0[code];
      if (sv != tag) continue;
      if (coarse || order) {
        c.misc[MS_CONFLICT] = 3;
      } else {
        const uint32_t t = atomicAdd(&c.misc[MS_NT], 1u);
        if (t < kChkCap) {
          t_code[t] = code;
          t_ref[t] = l * kLogCap + i;
        } else {
          c.misc[MS_CONFLICT] = 4;
        }
      }
    }
  }
  __syncthreads();
  if (c.misc[MS_CONFLICT]) return c.misc[MS_CONFLICT];
  const uint32_t nt = c.misc[MS_NT];
  if (nt == 0) return 0;
  // the changes those rows saw -> (node, log position) checks
  const uint32_t nrec = min(c.misc[MS_NCHG], kChgCap);
  for (uint32_t r = threadIdx.x; r < nrec; r += kBuildThreads) {
    const uint32_t code = chg[4 * r];
    for (uint32_t t = 0; t < nt; ++t) {
      if (t_code[t] != code) continue;  // (no early exit: the descent and the layer's own search may both have expanded the row)
      for (int e = 1; e <= 2; ++e) {
        const uint32_t x = chg[4 * r + e];
        if (x == kNone) continue;
        if (e == 2) {
          // a dropped node the search had met before it came to this row (or never scored: soft-deleted) was skipped
          // there: the row without it reads the same.  Known when the change was the batch's first to the row.
          const uint32_t pos = chg[4 * r + 3], ref = t_ref[t];
          const uint32_t l = ref / kLogCap, k = ref % kLogCap;
          if (pos < 64 && k >= sp[56 + l] && !(strict & 4u)) {
            const uint32_t* el = elog + (size_t)l * kLogWords;
            const uint32_t w = el[(pos < 32 ? 3 : 4) * kLogCap + k];
            if (((w >> (pos & 31)) & 1u) == 0) continue;
          }
        }
        const uint32_t k = atomicAdd(&c.misc[MS_NCHK], 1u);
        if (k < kChkCap) {
          c.nbr[k] = x;
          c.slist[k] = (uint16_t)k;
          k_ref[k] = t_ref[t] | (e == 2 ? 0x80000000u : 0u);
        } else {
          c.misc[MS_CONFLICT] = 4;
        }
      }
    }
  }
  __syncthreads();
  if (c.misc[MS_CONFLICT]) return c.misc[MS_CONFLICT];
  const uint32_t nchk = c.misc[MS_NCHK];
  if (nchk == 0) return 0;  // the rows were only re-ordered
  float2 q2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const uint32_t j = (uint32_t)b * 128u + 2u * (uint32_t)c.lane;
    q2[b] = j < g.dpad ? *(const float2*)(g.rows + (size_t)node * g.dpad + j) : make_float2(0.0f, 0.0f);
  }
  score_lists<NB, FULL>(g, c, q2, nchk);  // dist[i] overwrites the touched-row list, which is no longer needed
  __syncthreads();
  // ---- stage 1, one thread per check: the rule above.  Outcome in k_res[i]: state << 22 | a << 11 | b with
  //      0 x never entered the set, 1 x was inside over the expansions (k, b], 2 second look needed (below), 3 conflict
  uint32_t* k_res = (uint32_t*)c.cand + kChkCap;
  for (uint32_t i = threadIdx.x; i < nchk; i += kBuildThreads) {
    const float dx = c.dist[i];
    const uint32_t ref = k_ref[i] & 0x7FFFFFFFu;
    const bool dropped = (k_ref[i] >> 31) != 0;
    const uint32_t l = ref / kLogCap, k = ref % kLogCap;
    const uint32_t* el = elog + (size_t)l * kLogWords;
    const uint32_t nlog = sp[24 + l], seg0 = sp[56 + l];
    uint32_t state = 3, a = 0, b = 0, why = 6;  // 6: the log ends with x still inside the set (or the set never filled)
    if (k < seg0) {
      if (__uint_as_float(el[kLogCap + k]) < dx) state = 0;
      else why = 5;
    } else {
      uint32_t t = k;
      for (; t < nlog; ++t) {
        const uint32_t pt = el[2 * kLogCap + t];
        // (a pop that shared its distance with another: the pair's order is immaterial only while neither expansion
        // admits anything nearer than the pair — x, met at such an expansion, has to be farther than that pop as well)
        if ((t > k || (pt >> 31)) && !(__uint_as_float(pt & 0x7FFFFFFFu) < dx)) break;  // a pop not nearer than x
        if (__uint_as_float(el[kLogCap + t]) < dx) {
          state = t == k ? 0u : 1u;
          b = t;
          break;
        }
        if (strict & 8u) break;
      }
      if (state == 3 && t < nlog && t > k && l == 0 && !(strict & 2u)) {
        // x is inside the set when a candidate farther than x is popped: x itself is popped first.
        const uint32_t pt = el[2 * kLogCap + t];
        why = 5;
        if (dropped) {
          // the logged run popped x (= the row logged at t) — harmless if the search, no longer meeting x at k, meets it
          // in another row before that pop (second look: rows k+1 .. t-1)
          if (pt == __float_as_uint(dx) && el[t] == c.nbr[i]) {
            state = 2;
            a = t;
            b = t;
          }
        } else if ((pt >> 31) == 0 && __uint_as_float(pt) > dx) {
          // the run with x pops x before the candidate of t.  If that expansion admits nobody (second look) the run goes on
          // as logged, x inside the set until W drops below it — provided no later pop is the set's (displaced) maximum
          uint32_t u = t;
          bool ok = true;
          for (; u < nlog; ++u) {
            const uint32_t pu = el[2 * kLogCap + u];
            if ((pu >> 31) || !(pu < el[kLogCap + u - 1])) {
              ok = false;
              atomicAdd(&g.state->why[17], 1u);
              break;
            }
            if (__uint_as_float(el[kLogCap + u]) < dx) break;
          }
          if (ok && u < nlog) {
            state = 2;
            a = t;
            b = u;
          } else if (ok) {
            why = 6;
          }
        }
      } else if (state == 3 && t < nlog) {
        why = 5;
      }
    }
    if (state == 3) atomicMax(&c.misc[MS_CONFLICT], dropped ? 9u : why);
    k_res[i] = state << 22 | a << 11 | b;
  }
  __syncthreads();
  if (c.misc[MS_CONFLICT]) return c.misc[MS_CONFLICT];
  // ---- stage 2: the checks that asked for a second look (few).  Each must be alone in the set while it is there: the
  //      arguments above are for one displaced maximum at a time.
  uint32_t n2 = 0;
  for (uint32_t i = 0; i < nchk && n2 <= 4; ++i) n2 += (k_res[i] >> 22) == 2 ? 1u : 0u;
  if (n2 == 0) return 0;
  if (n2 > 4) return 4;
  const uint32_t* el = elog;  // layer 0
  const uint32_t seg0 = sp[56];
  for (uint32_t i = 0; i < nchk; ++i) {
    if ((k_res[i] >> 22) != 2) continue;
    const uint32_t k = (k_ref[i] & 0x7FFFFFFFu) % kLogCap, ta = (k_res[i] >> 11) & 2047u, tb = k_res[i] & 2047u;
    const bool dropped = (k_ref[i] >> 31) != 0;
    const uint32_t x = c.nbr[i];
    __syncthreads();
    if (threadIdx.x == 0) c.misc[MS_DONE] = 0;
    __syncthreads();
    // (a) nobody else inside the set during (k, tb]
    for (uint32_t o = threadIdx.x; o < nchk; o += kBuildThreads) {
      const uint32_t so = k_res[o] >> 22;
      if (o == i || so == 0 || (k_ref[o] & 0x7FFFFFFFu) / kLogCap != 0) continue;
      const uint32_t ko = (k_ref[o] & 0x7FFFFFFFu) % kLogCap, bo = k_res[o] & 2047u;
      if (ko < seg0) continue;
      if (ko < tb && k < bo) {
        c.misc[MS_CONFLICT] = dropped ? 9u : 5u;
        atomicAdd(&g.state->why[15], 1u);
      }
    }
    if (dropped) {
      // (b) x in the row of an expansion after k and before its own: met there instead
      for (uint32_t t = k + 1 + (uint32_t)c.wave; t < ta; t += kBuildWaves) {
        const uint32_t* row = g.adj0 + (size_t)el[t] * g.stride0;
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(row[0]);
        const bool hit = (uint32_t)c.lane < cnt && row[1 + c.lane] == x;
        if (__ballot(hit) && c.lane == 0) c.misc[MS_DONE] = 1;
      }
      __syncthreads();
      if (c.misc[MS_DONE] == 0 && threadIdx.x == 0) c.misc[MS_CONFLICT] = 9;
    } else {
      // (b) everything the search has met before it pops x: the rows of the expansions seg0 .. ta-1 and their nodes
      clear_bitmap(g, c);
      __syncthreads();
      for (uint32_t t = seg0 + (uint32_t)c.wave; t < ta; t += kBuildWaves) {
        const uint32_t e = el[t];
        const uint32_t* row = g.adj0 + (size_t)e * g.stride0;
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(row[0]);
        if ((uint32_t)c.lane < cnt) {
          const uint32_t y = row[1 + c.lane];
          atomicOr(&c.bitmap[y >> 5], 1u << (y & 31));
        }
        if (c.lane == 0) atomicOr(&c.bitmap[e >> 5], 1u << (e & 31));
      }
      __syncthreads();
      // (c) x's own row: whoever is new to the search and alive must be turned away (not nearer than the maximum then)
      if (c.wave == 0) {
        const uint32_t* row = g.adj0 + (size_t)x * g.stride0;
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(row[0]);
        bool fresh = false;
        uint32_t y = 0;
        if ((uint32_t)c.lane < cnt) {
          y = row[1 + c.lane];
          fresh = ((c.bitmap[y >> 5] >> (y & 31)) & 1u) == 0 && !(g.any_deleted && g.deleted[y]);
        }
        const uint64_t fm = __ballot(fresh);
        if (fresh) {
          const uint32_t at = kChkCap + (uint32_t)__popcll(fm & ((1ull << c.lane) - 1));  // above the checks' own entries
          c.nbr[at] = y;
          c.slist[(uint32_t)__popcll(fm & ((1ull << c.lane) - 1))] = (uint16_t)at;
        }
        if (c.lane == 0) c.misc[MS_NSCORE] = (uint32_t)__popcll(fm);
      }
      __syncthreads();
      const uint32_t nf = c.misc[MS_NSCORE];
      // (slist is rewritten: the checks' entries are i -> i, restored below)
      score_lists<NB, FULL>(g, c, q2, nf);
      __syncthreads();
      const float w = __uint_as_float(el[kLogCap + ta - 1]);
      if (threadIdx.x < nf && c.dist[kChkCap + threadIdx.x] < w) {
        c.misc[MS_CONFLICT] = 5;
        atomicAdd(&g.state->why[16], 1u);
      }
      __syncthreads();
      for (uint32_t o = threadIdx.x; o < min(nchk, 64u); o += kBuildThreads) c.slist[o] = (uint16_t)o;
    }
    __syncthreads();
    if (c.misc[MS_CONFLICT]) break;
  }
  __syncthreads();
  const uint32_t why = c.misc[MS_CONFLICT];
  __syncthreads();
  return why;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// Speculation: workgroup (b, l) runs the layer-l search of node first + cursor + b against the graph as it stands (the
// searches of an insert's layers only share the greedy descent, which every one of them repeats: a node with levels
// would otherwise take several times as long as its batch mates, and the slowest workgroup sets the launch's duration).
template <int NB, bool FULL>
__global__ __launch_bounds__(kBuildThreads, 1) void hnsw_insert_search_kernel(const BuildView g, uint32_t first, uint32_t n, uint32_t exact_positions,
                                                                              uint32_t tag, uint32_t* __restrict__ spec /* [grid.x][kSpecWords] */,
                                                                              uint32_t* __restrict__ elogs /* [grid.x][kBuildLayers][kLogWords] */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
  const uint32_t layer = blockIdx.y;
  const BuildState st = *g.state;
  const uint32_t idx = st.cursor + blockIdx.x;
  if (idx >= n || st.status != 0 || !st.has_entry) return;  // the flags in the header keep an older batch's tag: not usable
  const uint32_t node = first + idx;
  const uint32_t level = g.level[node];
  if (level >= (uint32_t)kBuildLayers || layer > level) return;
  const BuildLds L = build_lds_layout(g.bitmap_words, g.ef, g.cand_cap);
  BuildCtx c = build_ctx(lds_b, L);
  uint32_t* out = spec + (size_t)blockIdx.x * kSpecWords;
  uint32_t* elog = elogs + ((size_t)blockIdx.x * kBuildLayers + layer) * kLogWords;
  // a search that meets equal distances starts again with the restated heaps — twice the time: only the first few
  // speculations of a batch, the ones most likely to be adopted, do that; the others are left to the next batch
  // (`exact_positions`: the host raises it to the whole batch on data where ties are the rule, e.g. duplicate vectors)
  uint32_t tstat[4] = {0, 0, 0, 0};
  const bool ok = insert_searches<NB, FULL>(g, c, node, level, st.entry, st.entry_level, elog, tstat, blockIdx.x < exact_positions, (int)layer);
  __syncthreads();
  const uint32_t nlog = c.misc[MS_NLOG];
  if (threadIdx.x == 0) {
    if (tstat[3] || !ok) atomicAdd(&g.state->spec_ties, 1u);
    out[1] = node;
    out[2] = st.entry;
    out[8 + layer] = (ok && nlog <= kLogCap) ? tag : 0u;
    out[24 + layer] = nlog;
    out[40 + layer] = c.misc[MS_ORDER];
    out[56 + layer] = c.misc[MS_SEG0];
  }
  const uint32_t* r = c.res + layer * kResWords;
  for (uint32_t i = threadIdx.x; i < kResWords; i += kBuildThreads) out[kSpecHdr + layer * kResWords + i] = r[i];
}

// Commit: ONE workgroup takes the nodes first + cursor ... in order.  A speculated search is adopted when every row it
// expanded is untouched by this batch and the entry point is the one it started from; otherwise the search runs here
// (at most max_rerun times per launch, then the kernel stops and leaves the rest to the next speculation).
// spec == nullptr: no speculation, every search runs here (small graphs, where every insert touches what the next reads).
template <int NB, bool FULL>
__global__ __launch_bounds__(kBuildThreads, 1) void hnsw_insert_commit_kernel(const BuildView g, uint32_t first, uint32_t n, uint32_t count,
                                                                              uint32_t tag, uint32_t max_rerun,
                                                                              const uint32_t* __restrict__ spec,
                                                                              const uint32_t* __restrict__ elogs, uint32_t* __restrict__ chg,
                                                                              uint32_t strict) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_b[];
  const BuildLds L = build_lds_layout(g.bitmap_words, g.ef, g.cand_cap);
  BuildCtx c = build_ctx(lds_b, L);
  BuildState st = *g.state;
  if (st.status != 0) return;
  if (threadIdx.x == 0) {
    c.misc[MS_NCHG] = 0;
    c.misc[MS_CHGOVER] = 0;
  }
  __syncthreads();
  uint32_t reruns = 0;
  uint32_t stat[4] = {0, 0, 0, 0};
  uint32_t n_valid = 0, n_rerun = 0, stopped = 0, why_stop = 0, n_chk = 0, n_touch = 0;
  uint32_t done = 0;
  for (uint32_t b = 0; b < count; ++b) {
    const uint32_t idx = st.cursor + b;
    if (idx >= n) break;
    const uint32_t node = first + idx;
    const uint32_t level = g.level[node];
    if (level >= (uint32_t)kBuildLayers) {
      st.status = 1;
      break;
    }
    if (!st.has_entry) {  // the first node: entry point, no links (:249-258)
      if (threadIdx.x == 0) {
        g.adj0[(size_t)node * g.stride0] = 0;
        for (uint32_t l = 1; l <= level; ++l) g.adjU[(size_t)(g.ubase[node] + l - 1) * g.strideU] = 0;
      }
      st.has_entry = 1;
      st.entry = node;
      st.entry_level = level;
      st.n_linked += 1;
      done += 1;
      __syncthreads();
      continue;
    }
    bool have = false;
    if (spec) {
      const uint32_t* sp = spec + (size_t)b * kSpecWords;
      bool usable = sp[1] == node && sp[2] == st.entry;
      for (uint32_t l = 0; l <= level; ++l) usable = usable && sp[8 + l] == tag;
      uint32_t why = sp[1] == node && sp[2] == st.entry ? 2u : 1u;
      if (usable) {
        why = validate_speculation<NB, FULL>(g, c, node, level, tag, sp, elogs + (size_t)b * kBuildLayers * kLogWords, chg, strict);
        have = why == 0;
        n_chk += c.misc[MS_NCHK];
        n_touch += c.misc[MS_NT];
        if (have) {
          const uint32_t words = (level + 1) * kResWords;
          for (uint32_t i = threadIdx.x; i < words; i += kBuildThreads) c.res[i] = sp[kSpecHdr + i];
        }
        __syncthreads();
      }
      if (!have) why_stop = why;
    }
    if (have) {
      n_valid += 1;
    } else {
      if (spec && done > 0 && reruns >= max_rerun) {  // the first node of a launch is always taken: progress
        stopped = 1;
        break;
      }
      reruns += 1;
      n_rerun += 1;
      BSTAMP(ts0);
      if (!insert_searches<NB, FULL>(g, c, node, level, st.entry, st.entry_level, nullptr, stat)) {
        st.status = 1;
        break;
      }
      BSTAMP(ts1);
      BSTAMP_ADD(g, 4, ts0, ts1);
    }
    BSTAMP(tl0);
    insert_links(g, c, node, level, tag, spec ? chg : nullptr);
    BSTAMP(tl1);
    BSTAMP_ADD(g, 5, tl0, tl1);
    st.n_linked += 1;
    if (level > st.entry_level) {  // :372-375
      st.entry = node;
      st.entry_level = level;
    }
    done += 1;
    __threadfence_block();
    __syncthreads();  // the rows written above are read by the next insert's searches
  }
  if (threadIdx.x == 0) {
    BuildState* o = g.state;
    o->has_entry = st.has_entry;
    o->entry = st.entry;
    o->entry_level = st.entry_level;
    o->n_linked = st.n_linked;
    o->cursor = st.cursor + done;
    o->status = st.status;
    o->n_valid += n_valid;
    o->n_rerun += n_rerun;
    o->n_stopped += stopped;
    if (stopped) o->why[why_stop < 10 ? why_stop : 0] += 1;
    o->why[7] += n_chk;
    o->why[8] += n_touch;
    o->rounds += stat[0];
    o->consumed += stat[1];
    o->scored += stat[2];
    o->ties += stat[3];
  }
}

// Distance of every stored edge of the listed rows (code: node for layer 0, 0x80000000 | upper row otherwise, with
// `owner[i]` = the node the upper row belongs to), or of ALL layer-0 rows / upper rows when codes == nullptr: one wave
// per row, base = the row's node.  Fills dist0 / distU for graphs whose rows came from the host.
template <int NB, bool FULL>
__global__ __launch_bounds__(256) void graph_edge_dist_kernel(const BuildView g, const uint32_t* __restrict__ codes,
                                                             const uint32_t* __restrict__ owner, uint32_t n_rows, uint32_t upper_all) {
  const uint32_t tile_rows = kTileRows;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_e[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = (float*)lds_e + (size_t)wave * tile_rows * kFastStride;
  const uint32_t i = blockIdx.x * 4 + wave;
  if (i >= n_rows) return;
  uint32_t base_node;
  size_t at;
  bool upper;
  if (codes) {
    const uint32_t code = codes[i];
    upper = (code >> 31) != 0;
    base_node = upper ? owner[i] : code;
    at = upper ? (size_t)(code & 0x7FFFFFFFu) * g.strideU : (size_t)code * g.stride0;
  } else {
    upper = upper_all != 0;
    base_node = upper ? owner[i] : i;
    at = upper ? (size_t)i * g.strideU : (size_t)i * g.stride0;
  }
  const uint32_t* adj = upper ? g.adjU : g.adj0;
  float* adjd = upper ? g.distU : g.dist0;
  const uint32_t k = __builtin_amdgcn_readfirstlane(adj[at]);
  if (k == 0) return;
  float2 q2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const uint32_t j = (uint32_t)b * 128u + 2u * (uint32_t)lane;
    q2[b] = j < g.dpad ? *(const float2*)(g.rows + (size_t)base_node * g.dpad + j) : make_float2(0.0f, 0.0f);
  }
  for (uint32_t o = 0; o < k; o += BuildRows<NB>::kMax) {
    const uint32_t cnt = min(BuildRows<NB>::kMax, k - o);
    const uint32_t pn = (uint32_t)lane < cnt ? adj[at + 1 + o + lane] : 0u;
    const float d = score_upto16<NB, FULL>(g.rows, g.dpad, q2, pn, cnt, tile, lane);
    if ((uint32_t)lane < cnt) adjd[at + 1 + o + lane] = d;
  }
}

}  // namespace fvdb
