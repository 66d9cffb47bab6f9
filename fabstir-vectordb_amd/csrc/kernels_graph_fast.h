// kernels_graph_fast.h — the HNSW traversal kernel the batch search normally runs (ef <= 64): one wavefront per query,
// same results as hnsw_search_kernel (kernels_graph.h) and as the host walk, several times faster per hop.
//
// What changed against the restated-BinaryHeap kernel, and why the results cannot differ:
//
//  * `nearest` is a SORTED register array (lane i = i-th nearest), and `candidates` is not stored at all.
//    In search_layer (src/hnsw/core.rs:469-554) every candidate is pushed to both heaps together (:524-527).  Once
//    `nearest` is full an element leaves it only as its maximum, at a moment when its distance is >= every other
//    member's, and `worst` never grows afterwards; so when `candidates` would pop such an evicted element its
//    distance is > worst and the loop ends (:499-501).  The elements `candidates` can still yield for expansion are
//    therefore exactly the members of `nearest` that have not been expanded yet, in ascending distance: the pop is
//    "lowest lane whose expanded bit is clear", and "no such lane" is both of the reference's exits.  With all
//    distances inside the two heaps distinct, peek/pop of a priority queue are functions of its contents, so a
//    sorted array and std's BinaryHeap behave identically.  When two members hold EQUAL distances (duplicate
//    vectors) the reference's behaviour depends on heap layout: every insertion checks for an equal distance
//    already present, and if one is ever seen the wave abandons the sorted-array search and runs the query again
//    from the start with the restated heaps (hnsw_search_exact_body, kernels_graph.h), so ties still resolve as in
//    the reference.  (A second launch for those queries was tried first: one or two queries per thousand meet an
//    accidental tie on real data, and a launch of their own behind the main one doubled the batch's latency.)
//
//  * Scoring keeps the reference's arithmetic — per row t = q_i - x_i; sum = sum + t*t, i ascending, f32, no FMA —
//    but splits it where it is order-free: the products t*t are computed with the rows loaded COALESCED (a wave
//    instruction reads 512 contiguous bytes of one row; all of the hop's loads are in flight together, in registers),
//    each 128-dim block of products is transposed through a small LDS tile, and lane r then adds row r's products in
//    dimension order.  The sum sees the same addends in the same order, so the bits are the reference's; the VALU
//    work per hop drops from 3 ops per dim per lane-pass to ~1, and the 12 dependent L2 round trips of the old
//    per-lane row stream become one HBM latency.
//
// The visited set stays the per-query bitmap in HBM (atomicOr, cleared from a log at the end of each layer).
#pragma once
#include "kernels_graph.h"

#pragma clang fp contract(off)

namespace fvdb {

#ifndef FVDB_FAST_ADD_UNROLL
#define FVDB_FAST_ADD_UNROLL 16  // LDS reads in flight ahead of the add chain (8: 1 % slower; 32: spills)
#endif
constexpr int kFastAddUnroll = FVDB_FAST_ADD_UNROLL;
#ifndef FVDB_FAST_WAVES16
#define FVDB_FAST_WAVES16 3      // waves per SIMD the 16-row form is compiled for (its LDS allows two workgroups per CU)
#endif
constexpr uint32_t kFastStride = 132;  // floats per staged row block: 128 products + 4 pad (lane r's reads hit 16 distinct bank quads)

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v) {
  // lane i <- lane i-1 (lane 0 <- 0): one VALU op instead of an LDS-crossbar shuffle
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// LDS per wave: pending [64 u32] | scalars [16 u32] | two product tiles [R][kFastStride] floats
#ifndef FVDB_FAST_TILES
#define FVDB_FAST_TILES 2  // product tiles per wave: 2 = the products of block c + 1 are written during the add chain of block c
#endif
__host__ __device__ inline size_t graph_fast_lds_bytes(uint32_t R) {
  return (64 + 16) * 4 + FVDB_FAST_TILES * (size_t)R * kFastStride * 4;
}

// Distances of the wave's query to `cnt` rows (1 <= cnt <= RC; lane r of `pn` holds the r-th row's node): returns, in
// lane r < cnt, sqrt of the reference's sum.  q2[c] = dims (128c + 2*lane, +1) of the query, held in registers for the
// whole kernel.  Straight-line code for exactly RC rows — rows past cnt repeat the last one (an L2 hit) and their sums
// are ignored — so the scheduler can fill the bubbles of the dependent add chain of block c with the products of block
// c + 1, which go to the other tile.  FULL: dpad == NB * 128, no bounds checks.
template <int NB, int RC, bool FULL>
__device__ __forceinline__ float score_fixed(const float* __restrict__ rows, uint32_t dpad, const float2 (&q2)[NB], uint32_t pn,
                                             uint32_t cnt, float* stage, uint32_t tile_floats, int lane
#ifdef FVDB_GRAPH_STAMPS
                                             , unsigned long long* t_acc
#endif
) {
  STAMP(ta);
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
      else x[r][c] = j < dpad ? *(const float2*)(row + j) : make_float2(0.0f, 0.0f);  // dpad % 4 == 0: pairs never straddle it
    }
  }
  STAMP(tb);
  STAMP_ADD(8, ta, tb);
  const uint32_t lrow = (uint32_t)lane < (uint32_t)RC ? (uint32_t)lane : (uint32_t)(RC - 1);  // idle lanes add a valid row too: no branch
  auto products = [&](int c, float* tile) {
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const float t0 = q2[c].x - x[r][c].x, t1 = q2[c].y - x[r][c].y;
      *(float2*)(tile + (uint32_t)r * kFastStride + 2u * (uint32_t)lane) = make_float2(t0 * t0, t1 * t1);
    }
  };
  products(0, stage);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  STAMP(tc1);
  STAMP_ADD(9, tb, tc1);
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    float* cur = FVDB_FAST_TILES == 2 ? stage + (uint32_t)(c & 1) * tile_floats : stage;
    if (FVDB_FAST_TILES == 2 && c + 1 < NB) products(c + 1, stage + (uint32_t)((c + 1) & 1) * tile_floats);
    const float4* p = (const float4*)(cur + lrow * kFastStride);
#pragma unroll kFastAddUnroll
    for (int i = 0; i < 32; ++i) {
      const float4 v = p[i];
      acc = acc + v.x;
      acc = acc + v.y;
      acc = acc + v.z;
      acc = acc + v.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // tile c is rewritten by block c + 2, tile c + 1 is complete
    if (FVDB_FAST_TILES == 1 && c + 1 < NB) {  // one tile: the next block's products only now
      products(c + 1, stage);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  STAMP(tc2);
  STAMP_ADD(11, tc1, tc2);
  return sqrtf(acc);
}

// the smallest straight-line variant that holds the round
template <int NB, int R>
__device__ __forceinline__ float score_round(const GraphView& g, const float2 (&q2)[NB], uint32_t pn, uint32_t cnt,
                                             float* stage, int lane
#ifdef FVDB_GRAPH_STAMPS
                                             , unsigned long long* t_acc
#define FVDB_TACC , t_acc
#else
#define FVDB_TACC
#endif
) {
  const uint32_t dpad = g.dpad, tile = (uint32_t)R * kFastStride;
  if (dpad == (uint32_t)NB * 128u) {
    if (R > 8 && cnt > 8) return score_fixed<NB, R, true>(g.rows, dpad, q2, pn, cnt, stage, tile, lane FVDB_TACC);
    if (R > 4 && cnt > 4) return score_fixed<NB, (R < 8 ? R : 8), true>(g.rows, dpad, q2, pn, cnt, stage, tile, lane FVDB_TACC);
    return score_fixed<NB, (R < 4 ? R : 4), true>(g.rows, dpad, q2, pn, cnt, stage, tile, lane FVDB_TACC);
  }
  if (R > 8 && cnt > 8) return score_fixed<NB, R, false>(g.rows, dpad, q2, pn, cnt, stage, tile, lane FVDB_TACC);
  return score_fixed<NB, (R < 8 ? R : 8), false>(g.rows, dpad, q2, pn, cnt, stage, tile, lane FVDB_TACC);
}
#undef FVDB_TACC

// BYTES: the visited set is one BYTE per node (plain load + plain store: lanes of one instruction always hold different
// nodes, so they never share a byte) instead of one bit per node with atomicOr — scattered integer atomics execute at
// the memory side, one uncached request each, and 32 of them per hop were a chip-wide throughput limit.  The row of a
// query is vstride bytes either way (the exact-heap restart uses its first `words` words as a bitmap).
template <int NB, int R, bool BYTES>
__global__ __launch_bounds__(256, (R <= 8 ? 4 : (R <= 12 ? 3 : FVDB_FAST_WAVES16))) void hnsw_search_fast_kernel(const GraphView g, const float* __restrict__ queries, uint32_t B,
                                                              uint32_t k, uint32_t ef_final, uint32_t cand_cap, uint32_t wave_lds,
                                                              uint8_t* __restrict__ visited /* [B][vstride] zero on entry */,
                                                              uint32_t vstride,
                                                              uint32_t words, uint32_t* __restrict__ touched /* [B][tcap] */,
                                                              uint32_t tcap, uint32_t* __restrict__ out_nodes,
                                                              float* __restrict__ out_dist, uint32_t* __restrict__ out_counts,
                                                              uint32_t* __restrict__ out_status, HItem* __restrict__ spill /* [B][spill_cap] */,
                                                              uint32_t spill_cap) {
  // four independent waves per workgroup (one query each; no workgroup-level synchronisation anywhere): a grid of
  // B/4 workgroups of 4 waves spreads over the CUs one wave per SIMD, where B single-wave workgroups were seen to
  // leave a straggler waiting for a slot behind long-running waves
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_f_all[];
  const int lane = threadIdx.x & 63;
  const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  unsigned char* lds_f = lds_f_all + (threadIdx.x >> 6) * wave_lds;  // max(fast, exact-heap) bytes per wave
  uint32_t* pending = (uint32_t*)lds_f;
  float* stage = (float*)(pending + 64 + 16);  // two tiles
  uint8_t* visb = visited + (size_t)b * vstride;
  uint32_t* vis = (uint32_t*)visb;
  uint32_t* tch = touched + (size_t)b * tcap;
  const uint32_t dpad = g.dpad;
  const uint64_t lt = (1ull << lane) - 1;

  float2 q2[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const uint32_t j = (uint32_t)c * 128u + 2u * (uint32_t)lane;
    q2[c] = j < dpad ? *(const float2*)(queries + (size_t)b * dpad + j) : make_float2(0.0f, 0.0f);
  }

#ifdef FVDB_GRAPH_STAMPS
  unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // nearest = [(entry, dist(q, entry))]  (:432-435)
  const float d_entry = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(score_round<NB, R>(g, q2, g.entry, 1, stage, lane
#ifdef FVDB_GRAPH_STAMPS
                                                                                                            , t_acc
#endif
                                                                                                            ))));
  uint32_t ep_node = g.entry;
  float ep_d = d_entry;

#ifdef FVDB_GRAPH_STAMPS
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();
#endif
  uint32_t rows_scored = 1, hops_done = 0;
  uint32_t hn = 0;   // lane i: node of the i-th nearest | expanded << 31
  float hd = 0.0f;   //         its distance
  uint32_t nN = 0;
  uint32_t status = 0;  // 1: visited log overflow (host walk), 2: equal distances met where the heap layout decides (exact-heap kernel)
  // Equal distances inside the heaps (see kernels_graph_build.h for the argument): BinaryHeap's layout decides between
  // equal keys only when two of them are at the top of a heap at once.  `amb`: two members shared the maximum when one had
  // to leave (an expanded one is sent away; the search is the same unless the survivor is popped or ends up in the
  // result).  `twin_*`: two candidates shared the minimum at a pop (expanded back to back in either order if neither
  // expansion admits anything nearer).  Equal distances in the final list: the stable sort keeps the heap array's order.
  float amb = __uint_as_float(0x7FC00000u), twin_d = 0.0f;  // NaN: none (compares equal to nothing)
  uint32_t twin_left = 0;
  for (uint32_t layer = g.top_level + 1; layer-- > 0 && status == 0;) {
    const uint32_t ef = layer == 0 ? ef_final : 1;
    uint32_t nT = 1;
    // ---- search_layer(query, ep, ef, layer): nearest = candidates = {ep}, visited = {ep} ----
    nN = 1;
    hn = ep_node;
    hd = ep_d;
    amb = __uint_as_float(0x7FC00000u);
    twin_left = 0;
    if (lane == 0) {
      if (BYTES) visb[ep_node] = 1;
      else atomicOr(&vis[ep_node >> 5], 1u << (ep_node & 31));
      tch[0] = ep_node;
    }
    // layer 0: the adjacency row of the LIKELY next candidate (the nearest unexpanded member as things stand before
    // this hop's admissions) is requested while the hop's rows are in flight; a closer newcomer makes it a miss
    uint32_t pf_node = 0xFFFFFFFFu, pf_w = 0, pf_nb = 0;
    for (;;) {
      // candidates.pop(): the nearest member not yet expanded; none left = the reference's exits (:498-501)
      STAMP(t0s);
      const uint64_t open = __ballot((uint32_t)lane < nN && (hn >> 31) == 0);
      if (open == 0) break;
      const uint32_t cl = (uint32_t)__builtin_ctzll(open);
      const uint32_t node = __builtin_amdgcn_readlane(hn, cl);
      {
        // two candidates share the smallest distance: which one BinaryHeap::pop returns is its layout's business
        const float dc = rlane_f(hd, cl);
        const uint32_t holders = (uint32_t)__popcll(open & __ballot((uint32_t)lane < nN && hd == dc));
        if (holders > 2 || dc == amb || (twin_left != 0 && (dc != twin_d || holders != twin_left))) {
          status = 2;
          break;
        }
        if (holders == 2 && twin_left == 0) {
          twin_d = dc;
          twin_left = 2;
        }
      }
      if ((uint32_t)lane == cl) hn |= 0x80000000u;
      uint32_t np = 0;
      if (layer == 0 || g.level[node] >= layer) {
        uint32_t cnt, nb = 0;
        if (layer == 0) {  // lane i holds neighbour i (count and neighbours: two independent loads of one row)
          uint32_t w;
          if (pf_node == node) {
            w = pf_w;
            nb = pf_nb;
          } else {
            const uint32_t* row = g.adj0 + (size_t)node * g.stride0;
            w = row[0];
            nb = (uint32_t)lane + 1 < g.stride0 ? row[lane + 1] : 0u;
          }
          cnt = __builtin_amdgcn_readfirstlane(w);
          const uint64_t rest = open & (open - 1);  // unexpanded members besides this one
          if (rest) {
            pf_node = __builtin_amdgcn_readlane(hn, (uint32_t)__builtin_ctzll(rest));
            const uint32_t* prow = g.adj0 + (size_t)pf_node * g.stride0;
            pf_w = prow[0];
            pf_nb = (uint32_t)lane + 1 < g.stride0 ? prow[lane + 1] : 0u;
          } else {
            pf_node = 0xFFFFFFFFu;
          }
        } else {
          const uint32_t* row = g.adjU + (size_t)(g.ubase[node] + layer - 1) * g.strideU;
          cnt = __builtin_amdgcn_readfirstlane(row[0]);
          if ((uint32_t)lane < cnt) nb = row[lane + 1];
        }
        bool fresh = false, keep = false;
        if ((uint32_t)lane < cnt) {
          if (BYTES) {
            fresh = visb[nb] == 0;                                          // visited.insert (:506-507)
            if (fresh) visb[nb] = 1;
          } else {
            const uint32_t bit = 1u << (nb & 31);
            fresh = (atomicOr(&vis[nb >> 5], bit) & bit) == 0;
          }
          keep = fresh && (g.any_deleted == 0 || g.deleted[nb] == 0);       // :511-513
        }
        const uint64_t fm = __ballot(fresh), km = __ballot(keep);
        const uint32_t nf = __popcll(fm);
        // a log that fills up stops growing: the layer's map is then cleared whole instead of entry by entry
        if (fresh && nT + nf <= tcap) tch[nT + __popcll(fm & lt)] = nb;
        nT += nf;
        np = __popcll(km);
        if (keep) pending[__popcll(km & lt)] = nb;  // list order preserved
      }
      if (status) break;
      hops_done += 1;
      rows_scored += np;
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): `pending` is in LDS; the log store and the prefetch stay in flight
      __builtin_amdgcn_wave_barrier();
      STAMP(t1s);
      STAMP_ADD(0, t0s, t1s);
      for (uint32_t base = 0; base < np && status == 0; base += R) {
        const uint32_t cnt = min((uint32_t)R, np - base);
        STAMP(t2s);
        const uint32_t pn = (uint32_t)lane < cnt ? pending[base + lane] : 0u;
        const float pd = score_round<NB, R>(g, q2, pn, cnt, stage, lane
#ifdef FVDB_GRAPH_STAMPS
                                            , t_acc
#endif
        );
        STAMP(t3s);
        STAMP_ADD(1, t2s, t3s);
#ifdef FVDB_GRAPH_STAMPS
        t_acc[4] += cnt;
        t_acc[5] += 1;
#endif
        // admission (:517-531) in neighbour order; `worst` only shrinks while the wave applies them, so whoever fails
        // against it now fails later too
        float worst = nN ? rlane_f(hd, nN - 1) : 0.0f;
        uint64_t todo = __ballot((uint32_t)lane < cnt && (nN < ef || pd < worst));
        while (todo) {
          const uint32_t i = (uint32_t)__builtin_ctzll(todo);
          todo &= todo - 1;
          const float d = rlane_f(pd, i);
          if (nN < ef || d < worst) {
            const uint32_t nd = __builtin_amdgcn_readlane(pn, i);
            const bool mine = (uint32_t)lane < nN;
#ifdef FVDB_FAST_STRICT_TIES
            if (__ballot(mine && hd == d)) {  // (A/B: the first form — any equal distance inside the heaps ends the attempt)
              status = 2;
              break;
            }
#endif
            if (twin_left != 0 && d < twin_d) {  // the pair is not expanded back to back: the order matters
              status = 2;
              break;
            }
            if (nN == ef) {
              // the maximum leaves (nearest.pop(), :528-530).  Two members sharing it: which one leaves is the layout's
              // business — an EXPANDED one is sent away if there is one, so that a survivor that can still be popped is
              // in this list too and its pop (if it comes to that) is seen
              const uint64_t grp = __ballot(mine && hd == worst);
              if (__popcll(grp) > 1) {
                amb = worst;
                const uint64_t ex = __ballot(mine && hd == worst && (hn >> 31) != 0);
                const uint32_t last = nN - 1;
                if (ex && ((ex >> last) & 1ull) == 0) {  // the last lane holds an unexpanded one: trade places with an expanded one
                  const uint32_t l = (uint32_t)__builtin_ctzll(ex);
                  const uint32_t a = __builtin_amdgcn_readlane(hn, l), z = __builtin_amdgcn_readlane(hn, last);
                  if ((uint32_t)lane == l) hn = z;
                  if ((uint32_t)lane == last) hn = a;
                }
              }
            }
            const uint32_t pos = __popcll(__ballot(mine && hd < d));
            const uint32_t up_n = dpp_wave_shr1(hn);
            const float up_d = __uint_as_float(dpp_wave_shr1(__float_as_uint(hd)));
            if ((uint32_t)lane > pos) {
              hn = up_n;
              hd = up_d;
            } else if ((uint32_t)lane == pos) {
              hn = nd;
              hd = d;
            }
            if (nN < ef) nN += 1;  // else the former maximum fell off the end (nearest.pop(), :528-530)
            worst = rlane_f(hd, nN - 1);
          }
        }
        STAMP(t4s);
        STAMP_ADD(2, t3s, t4s);
      }
#ifdef FVDB_GRAPH_STAMPS
      t_acc[6] += 1;
#endif
      if (status) break;
      if (twin_left != 0) twin_left -= 1;
    }
    // the layer's result: equal distances inside it would come out in the heap array's order, and the survivor of a tied
    // maximum is one of two legal members.  Only the part that is read counts: the first entry on the upper layers, the
    // first k (and the pair at the cut) on layer 0 — all of it when soft-deleted nodes are filtered out afterwards.
    if (status == 0) {
      const uint32_t read = layer != 0 ? 1u : (g.any_deleted ? nN : min(nN, k + 1));
      const bool in = (uint32_t)lane < read;
      const float nxt = __uint_as_float(dpp_wave_shr1(__float_as_uint(hd)));  // lane i <- lane i-1
      if (__ballot(in && lane > 0 && nxt == hd) || __ballot(in && hd == amb)) status = 2;
    }
    // ---- drop this layer's visited set ----
    if (nT <= tcap && status != 1) {
      for (uint32_t i = lane; i < nT; i += 64) {
        if (BYTES) visb[tch[i]] = 0;
        else vis[tch[i] >> 5] = 0;
      }
    } else {
      for (uint32_t w = lane; w < (BYTES ? vstride / 4 : words); w += 64) vis[w] = 0;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // the layer's result is `nearest` in ascending order (:541-553); its first element enters the next layer
    ep_node = __builtin_amdgcn_readlane(hn, 0) & 0x7FFFFFFFu;
    ep_d = rlane_f(hd, 0);
  }
  if (lane == 0 && g.counters) {
    atomicAdd(g.counters + 3, 1ull);
    if (status == 2) atomicAdd(g.counters + 2, 1ull);
  }
  if (status == 2) {
    // equal distances met inside the heaps: this query is searched again, from the start, with the reference's heaps
    // restated (kernels_graph.h) — same wave, same launch; the visited bitmap was left clean above
    hnsw_search_exact_body<true>(g, queries, b, k, ef_final, cand_cap, vis, words, tch, tcap, out_nodes, out_dist,
                                 out_counts, out_status, lds_f, lane, spill ? spill + (size_t)b * spill_cap : nullptr, spill ? spill_cap : 0u);
    return;
  }
  if (status) {
    if (lane == 0) {
      out_status[b] = status;
      out_counts[b] = 0;
    }
    return;
  }
#ifdef FVDB_GRAPH_STAMPS
  if (lane == 0 && g.stamps) {
    t_acc[7] = __builtin_amdgcn_s_memtime() - t_begin;
    for (int i = 0; i < 8; ++i) atomicAdd(g.stamps + i, t_acc[i]);
    for (int i = 8; i < 12; ++i) atomicAdd(g.stamps + 8 + 3 * 16384 + (i - 8), t_acc[i]);
    if (b < 16384) {
      const unsigned long long hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
      g.stamps[8 + 3 * b] = (t_acc[7] & 0xFFFFFFFFull) | (hw << 32) | (xcc << 60);
      g.stamps[8 + 3 * b + 1] = r_begin;  // 100 MHz ticks, one clock for the whole chip
      g.stamps[8 + 3 * b + 2] = __builtin_amdgcn_s_memrealtime() - r_begin;
    }
  }
#endif
  if (lane == 0 && g.counters) {
    atomicAdd(g.counters + 0, (unsigned long long)rows_scored);
    atomicAdd(g.counters + 1, (unsigned long long)hops_done);
  }
  // ---- filter deleted, take k (:451-466) ----
  const uint32_t my = hn & 0x7FFFFFFFu;
  const bool live = (uint32_t)lane < nN && g.deleted[my] == 0;
  const uint64_t lm = __ballot(live);
  const uint32_t rank = __popcll(lm & lt), total = min(__popcll(lm), k);
  if (live && rank < k) {
    out_nodes[(size_t)b * k + rank] = my;
    out_dist[(size_t)b * k + rank] = hd;
  }
  for (uint32_t i = total + lane; i < k; i += 64) {
    out_nodes[(size_t)b * k + i] = 0xFFFFFFFFu;
    out_dist[(size_t)b * k + i] = __uint_as_float(0x7F800000u);
  }
  if (lane == 0) {
    out_counts[b] = total;
    out_status[b] = 0;
  }
}

}  // namespace fvdb
