// kernels_util.h — the reference's top-k / merge helpers as batched kernels, one wavefront per row:
//   top_k_indices            src/core/vector_ops.rs:12-22    stable sort by descending score, first k indices
//   top_k_indices_heap       src/core/vector_ops.rs:180-201  BinaryHeap of size k, strict `>` replacement
//   StreamingTopK            src/core/vector_ops.rs:204-263  the same heap over (score, id) tuples
//   merge_search_results     src/core/vector_ops.rs:24-32 + SearchResult::deduplicate src/core/types.rs:206-223
// No index of the reference calls these; they are part of its public vector_ops surface and their semantics are
// what the engine's own selection and cross-GPU merge follow.  Results are the oracle's bit for bit, ties included.
#pragma once
#include "common.h"
#include "kernels_scan.h"

namespace fvdb {

// total order of finite floats as unsigned integers, ascending; -0.0 and +0.0 compare equal (partial_cmp)
__device__ __forceinline__ uint32_t mono_key(float s) {
  const uint32_t b = __float_as_uint(s == 0.0f ? 0.0f : s);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// top_k_indices: (score descending, index ascending) = what the stable sort at :19 orders by
template <int KR>
__global__ __launch_bounds__(256) void topk_sort_kernel(const float* __restrict__ scores, uint32_t B, uint32_t n, uint32_t k,
                                                        uint64_t* __restrict__ out, uint32_t* __restrict__ out_counts) {
  const int lane = threadIdx.x & 63;
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  WaveTopK<KR> tk;
  tk.init();
  uint32_t th = kInf32, tl = kInf32;
  const float* s = scores + (size_t)row * n;
  for (uint32_t e0 = 0; e0 < n; e0 += 64) {
    const uint32_t e = e0 + lane;
    uint32_t chi = kInf32, clo = 0;
    if (e < n) {
      chi = ~mono_key(s[e]);  // never kInf32 for a non-NaN score
      clo = e;
    }
    offer<KR>(tk, k, chi, clo, th, tl, lane);
  }
  uint32_t count = 0;
#pragma unroll
  for (int rr = 0; rr < KR; ++rr) {
    const uint32_t e = rr * 64 + lane;
    const bool have = e < k && tk.hi[rr] != kInf32;
    count += __popcll(__ballot(have));
    if (e < k) out[(size_t)row * k + e] = have ? (uint64_t)tk.lo[rr] : ~0ull;
  }
  if (lane == 0) out_counts[row] = count;
}

struct UItem {
  float s;
  uint32_t idx;  // position in the input row (output of the index variant)
  uint64_t id;   // StreamingTopK: the caller's id
};

// `a <= b` in the heap's order.  HeapItem::cmp (:169-178) reverses the score; (OrderedFloat, VectorId) tuples compare
// the reversed score first (:219-230), then the id.
template <bool TUPLE>
__device__ __forceinline__ bool u_le(const UItem& a, const UItem& b) {
  if (TUPLE) return a.s > b.s || (a.s == b.s && a.id <= b.id);
  return a.s >= b.s;
}
template <bool TUPLE>
__device__ __forceinline__ void u_sift_up(UItem* h, uint32_t pos) {
  const UItem elt = h[pos];
  while (pos > 0) {
    const uint32_t parent = (pos - 1) >> 1;
    if (u_le<TUPLE>(elt, h[parent])) break;
    h[pos] = h[parent];
    pos = parent;
  }
  h[pos] = elt;
}
// BinaryHeap::pop without returning the root: last item to the root, sift_down_to_bottom(0), sift_up
template <bool TUPLE>
__device__ __forceinline__ void u_pop(UItem* h, uint32_t& n) {
  n -= 1;
  if (n == 0) return;
  const UItem elt = h[n];
  const uint32_t end = n;
  uint32_t pos = 0, child = 1;
  const uint32_t lim = end >= 2 ? end - 2 : 0;
  while (child <= lim) {
    if (u_le<TUPLE>(h[child], h[child + 1])) child += 1;
    h[pos] = h[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (child == end - 1) {
    h[pos] = h[child];
    pos = child;
  }
  h[pos] = elt;
  u_sift_up<TUPLE>(h, pos);
}

// top_k_indices_heap / StreamingTopK: the heap lives in LDS and is driven by lane 0 in input order; the other lanes
// only pre-filter — an element that does not beat the root as it stands can never enter later in its 64-block,
// because the root's score only grows while the heap is full.
template <bool TUPLE>
__global__ __launch_bounds__(64) void topk_heap_kernel(const float* __restrict__ scores, const uint64_t* __restrict__ ids,
                                                       uint32_t B, uint32_t n, uint32_t k, uint64_t* __restrict__ out_ids,
                                                       float* __restrict__ out_scores, uint32_t* __restrict__ out_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_u[];
  UItem* heap = (UItem*)lds_u;         // [k]
  UItem* stage = heap + k;             // [64]
  uint32_t* sc = (uint32_t*)(stage + 64);  // [0] count
  const int lane = threadIdx.x;
  const uint32_t row = blockIdx.x;
  if (row >= B) return;
  const float* s = scores + (size_t)row * n;
  uint32_t cnt = 0;  // lane 0's copy is authoritative; broadcast through sc[0] per block
  if (lane == 0) sc[0] = 0;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (uint32_t e0 = 0; e0 < n; e0 += 64) {
    const uint32_t e = e0 + lane;
    UItem it{0.0f, e, 0};
    if (e < n) {
      it.s = s[e];
      it.id = TUPLE ? ids[(size_t)row * n + e] : (uint64_t)e;
    }
    stage[lane] = it;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    cnt = sc[0];
    const bool full = cnt >= k;
    const float root = full && k > 0 ? heap[0].s : 0.0f;
    uint64_t todo = __ballot(e < n && (!full || it.s > root));
    if (lane == 0) {
      while (todo) {
        const uint32_t i = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const UItem c = stage[i];
        if (cnt < k) {
          heap[cnt] = c;
          u_sift_up<TUPLE>(heap, cnt);
          cnt += 1;
        } else if (c.s > heap[0].s) {  // strict: an equal score never displaces (:190, :245)
          u_pop<TUPLE>(heap, cnt);
          heap[cnt] = c;
          u_sift_up<TUPLE>(heap, cnt);
          cnt += 1;
        }
      }
      sc[0] = cnt;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
  cnt = sc[0];
  // results: the heap's vector order (into_iter), stable-sorted by descending score (:197-199, :253-260)
  for (uint32_t i0 = 0; i0 < cnt; i0 += 64) {
    const uint32_t i = i0 + lane;
    if (i < cnt) {
      const UItem me = heap[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < cnt; ++j) {
        const float sj = heap[j].s;
        rank += (sj > me.s || (sj == me.s && j < i)) ? 1u : 0u;
      }
      out_ids[(size_t)row * k + rank] = TUPLE ? me.id : (uint64_t)me.idx;
      if (out_scores) out_scores[(size_t)row * k + rank] = me.s;
    }
  }
  for (uint32_t i = cnt + lane; i < k; i += 64) {
    out_ids[(size_t)row * k + i] = ~0ull;
    if (out_scores) out_scores[(size_t)row * k + i] = 0.0f;
  }
  if (lane == 0) out_counts[row] = cnt;
}

// merge_search_results: per query the concatenated result sets (id == ~0 is padding).  deduplicate keeps, per id, the
// entry with the smallest distance — the EARLIEST of those on a tie (`existing.distance <= result.distance` keeps the
// existing one) — then sorts ascending by distance; where the reference's tie order comes out of HashMap iteration,
// first appearance of the id in the input decides here (as in the oracle).  First k survive.
template <int KR>
__global__ __launch_bounds__(256) void merge_dedup_kernel(const uint64_t* __restrict__ ids, const float* __restrict__ dist,
                                                          uint32_t B, uint32_t n, uint32_t k, uint64_t* __restrict__ out_ids,
                                                          float* __restrict__ out_dist, uint32_t* __restrict__ out_counts) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= B) return;
  const uint64_t* qi = ids + (size_t)q * n;
  const float* qd = dist + (size_t)q * n;
  WaveTopK<KR> tk;
  tk.init();
  uint32_t th = kInf32, tl = kInf32;
  for (uint32_t e0 = 0; e0 < n; e0 += 64) {
    const uint32_t e = e0 + lane;
    uint32_t chi = kInf32, clo = 0;
    if (e < n && qi[e] != ~0ull) {
      const uint64_t id = qi[e];
      const float d = qd[e];
      uint32_t first = e;
      bool rep = true;  // no other entry of this id is smaller, nor equal and earlier
      for (uint32_t j = 0; j < n; ++j) {
        if (j == e || qi[j] != id) continue;
        if (j < first) first = j;
        const float dj = qd[j];
        if (dj < d || (dj == d && j < e)) rep = false;
      }
      if (rep) {
        chi = mono_key(d);
        clo = first;
      }
    }
    offer<KR>(tk, k, chi, clo, th, tl, lane);
  }
  uint32_t count = 0;
#pragma unroll
  for (int rr = 0; rr < KR; ++rr) {
    const uint32_t e = rr * 64 + lane;
    const bool have = e < k && tk.hi[rr] != kInf32;
    count += __popcll(__ballot(have));
    if (e < k) {
      const size_t o = (size_t)q * k + e;
      uint64_t id = ~0ull;
      float d = __uint_as_float(0x7F800000u);
      if (have) {  // the kept entry of this id: smallest distance, earliest on a tie (its bits, e.g. a -0.0, are returned)
        id = qi[tk.lo[rr]];
        uint32_t best = tk.lo[rr];
        d = qd[best];
        for (uint32_t j = best + 1; j < n; ++j)
          if (qi[j] == id && qd[j] < d) {
            d = qd[j];
            best = j;
          }
      }
      out_ids[o] = id;
      out_dist[o] = d;
    }
  }
  if (lane == 0) out_counts[q] = count;
}

}  // namespace fvdb
