// kernels_scan.h — the hot path: batched exact-L2 list scan with per-wavefront top-k.
//
// Replaces, for a whole query batch at once:
//   * IVFIndex::search_with_config's centroid ranking      (src/ivf/core.rs:645-656)   [coarse]
//   * its inverted-list scan                               (src/ivf/core.rs:659-674)   [fine]
//   * the full sort + truncate(k)                          (src/ivf/core.rs:676-678)
//   * find_nearest_centroid                                (src/ivf/core.rs:373-386)   [k = 1]
// Arithmetic is the reference's, bit for bit: each lane owns ONE database row and folds
// (x_i - q_i)^2 left to right in f32 with separate multiply and add (no FMA: this file is
// compiled with -ffp-contract=off and carries the pragma below), sqrt correctly rounded.
// Selection is by the unique 64-bit key (distance bits << 32 | scan position), so any
// merge order yields the reference's stable-sort result.
//
// Mapping to CDNA4: lane = row (64 rows per block, dimension-chunk-major so a wave's
// global_load_dwordx4 is one contiguous KiB); the Q queries of a work item are wave-uniform
// and come in through the scalar data path (s_load -> SGPR operands of the VALU ops), so a row
// chunk held in 4 VGPRs is reused by Q queries with no LDS traffic at all.  Top-k lives in a
// wave-distributed sorted register list (lane i = i-th best), updated by ballot/readlane/DPP-
// style shuffles only when a candidate beats the current k-th key.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

namespace fvdb {

#if defined(__HIP_DEVICE_COMPILE__)
#define FVDB_CONST_AS __attribute__((address_space(4)))
#else
#define FVDB_CONST_AS
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16_n __attribute__((ext_vector_type(16)));
typedef f32x16_n f32x16;  // loaded with cload16 (query rows are only 16-byte aligned in general)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Uniform (wave-invariant) read-only loads through the constant address space => s_load_*.
// Only used on buffers no kernel in the same launch writes.
template <typename T>
__device__ __forceinline__ T cload(const T* p) {
  return *(const FVDB_CONST_AS T*)(uintptr_t)p;
}

// 16 consecutive query dims (64 B, 16-byte aligned) through the scalar path
__device__ __forceinline__ f32x16_n cload16(const float* p) {
  typedef f32x16_n __attribute__((aligned(16))) f32x16_a;
  return *(const FVDB_CONST_AS f32x16_a*)(uintptr_t)p;
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rlane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }

__device__ __forceinline__ bool key_lt(uint32_t ah, uint32_t al, uint32_t bh, uint32_t bl) {
  return ah < bh || (ah == bh && al < bl);
}

// Wave-distributed sorted list of the 64*KR smallest keys seen: element e = r*64 + lane.
template <int KR>
struct WaveTopK {
  uint32_t hi[KR], lo[KR];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int r = 0; r < KR; ++r) hi[r] = lo[r] = kInf32;
  }

  // key of the k-th best (1-based k), wave-uniform
  __device__ __forceinline__ void kth(uint32_t k, uint32_t& th, uint32_t& tl) const {
    const uint32_t e = k - 1, rk = e >> 6, lk = e & 63;
    th = tl = kInf32;
#pragma unroll
    for (int r = 0; r < KR; ++r)
      if ((uint32_t)r == rk) {
        th = rlane(hi[r], lk);
        tl = rlane(lo[r], lk);
      }
  }

  // insert the wave-uniform key (nh, nl); the largest element falls off the end
  __device__ __forceinline__ void insert(uint32_t nh, uint32_t nl, int lane) {
    uint32_t carry_hi = 0, carry_lo = 0;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const bool gt = key_lt(nh, nl, hi[r], lo[r]);  // element > new key: it moves up one slot
      uint32_t up_hi = __shfl_up(hi[r], 1);
      uint32_t up_lo = __shfl_up(lo[r], 1);
      const uint32_t last_hi = rlane(hi[r], 63), last_lo = rlane(lo[r], 63);
      bool up_gt;
      if (lane == 0) {
        if (r == 0) {
          up_gt = false;
        } else {
          up_hi = carry_hi;
          up_lo = carry_lo;
          up_gt = key_lt(nh, nl, up_hi, up_lo);
        }
      } else {
        up_gt = key_lt(nh, nl, up_hi, up_lo);
      }
      hi[r] = gt ? (up_gt ? up_hi : nh) : hi[r];
      lo[r] = gt ? (up_gt ? up_lo : nl) : lo[r];
      carry_hi = last_hi;
      carry_lo = last_lo;
    }
  }
};

// Bitonic sort of one 64-bit key per lane, ascending by lane.  Used for the first block of a work
// item: sorting 64 candidates (21 compare-exchange steps) replaces up to 64 one-by-one insertions
// into an empty list.
__device__ __forceinline__ void wave_sort64(uint32_t& hi, uint32_t& lo, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t phi = __shfl_xor(hi, j), plo = __shfl_xor(lo, j);
      const bool up = (lane & k) == 0;     // this k-block sorts ascending
      const bool lower = (lane & j) == 0;  // lower lane of the pair
      const bool take = (lower == up) ? key_lt(phi, plo, hi, lo) : key_lt(hi, lo, phi, plo);
      hi = take ? phi : hi;
      lo = take ? plo : lo;
    }
  }
}

// Offer one candidate per lane (chi == kInf32 => none) to the list; th/tl is the running k-th key.
template <int KR>
__device__ __forceinline__ void offer(WaveTopK<KR>& tk, uint32_t k, uint32_t chi, uint32_t clo, uint32_t& th,
                                      uint32_t& tl, int lane) {
  uint64_t m = __ballot(chi != kInf32 && key_lt(chi, clo, th, tl));
  while (m) {
    const uint32_t l = (uint32_t)__builtin_ctzll(m);
    m &= m - 1;
    const uint32_t nh = rlane(chi, l), nl = rlane(clo, l);
    if (key_lt(nh, nl, th, tl)) {  // uniform re-check: earlier inserts may have tightened the bound
      tk.insert(nh, nl, lane);
      tk.kth(k, th, tl);
    }
  }
}

// Row storage: ST = 0 rows are f32 (float4 chunks), ST = 1 rows are IEEE fp16 (8 halfs per 16-byte chunk,
// d padded to a multiple of 16) widened to f32 in registers — the arithmetic after the load is the same
// exact f32 fold either way (config C5: results equal the oracle run on the fp16-rounded rows).
template <int ST>
__device__ __forceinline__ void load_rows16(const void* __restrict__ pool_data, uint32_t blk, uint32_t d4, uint32_t c,
                                            int lane, float (&x)[16]) {
  if (ST == 0) {
    const float4* xp = (const float4*)pool_data + (size_t)blk * d4 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = xp[(size_t)(c + i) * 64];
      x[4 * i + 0] = v.x;
      x[4 * i + 1] = v.y;
      x[4 * i + 2] = v.z;
      x[4 * i + 3] = v.w;
    }
  } else {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8* xp = (const h8*)pool_data + (size_t)blk * (d4 >> 1) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const h8 v = xp[(size_t)((c >> 1) + i) * 64];
#pragma unroll
      for (int t = 0; t < 8; ++t) x[8 * i + t] = (float)v[t];
    }
  }
}

// One work item: rows of blocks [b0, b1) of a list against the ne (<= QQ) queries of a group.
// QQ in {16, 8, 4}: short groups run the narrower instantiation instead of padding to 16.
template <int QQ, int KR, int ST>
__device__ __forceinline__ void scan_item(const void* __restrict__ pool_data, const uint64_t* __restrict__ pool_valid,
                                          const uint32_t d4, const uint32_t* __restrict__ list_blocks,
                                          const u32x2* __restrict__ entries, const float* __restrict__ queries,
                                          const uint32_t dpad, const uint32_t k, const uint32_t nprobe,
                                          const uint32_t maxsegs, u32x2* __restrict__ part, const uint32_t b_begin,
                                          const uint32_t b0, const uint32_t b1, const uint32_t e0, const uint32_t ne,
                                          const uint32_t seg, const int lane) {
  uint32_t qoff[QQ];
#pragma unroll
  for (int j = 0; j < QQ; ++j) {
    const u32x2 e = cload(entries + e0 + ((uint32_t)j < ne ? j : 0));
    qoff[j] = e.x * dpad;
  }

  WaveTopK<KR> tk[QQ];
#pragma unroll
  for (int j = 0; j < QQ; ++j) tk[j].init();

  for (uint32_t b = b0; b < b1; ++b) {
    const uint32_t blk = cload(list_blocks + b_begin + b);
    float acc[QQ];
#pragma unroll
    for (int j = 0; j < QQ; ++j) acc[j] = 0.0f;
    uint32_t c = 0;
    // 16 dims per step: one s_load_dwordx16 per query feeds 48 VALU ops, so the scalar-side address
    // arithmetic is 1/4 of a chunk-at-a-time loop's
    for (; c + 4 <= d4; c += 4) {
      float x[16];
      load_rows16<ST>(pool_data, blk, d4, c, lane, x);
      // query j+1's 16 dims are requested before query j's are consumed (scalar loads return out of
      // order, so the only usable wait is lgkmcnt(0): it lands after a 48-op compute block)
      f32x16 qn = cload16(queries + qoff[0] + 4 * c);
#pragma unroll
      for (int j = 0; j < QQ; ++j) {
        const f32x16 qv = qn;
        if (j + 1 < QQ) qn = cload16(queries + qoff[j + 1] + 4 * c);
        __builtin_amdgcn_sched_barrier(0);  // keep the request ahead of the 48 ops it overlaps with
        float t, a = acc[j];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          t = x[i] - qv[i];
          a = a + t * t;
        }
        acc[j] = a;
      }
    }
    if (ST == 0) {
      const float4* xp = (const float4*)pool_data + (size_t)blk * d4 * 64 + lane;
      for (; c < d4; ++c) {  // d4 % 4 leftover chunks (f32 rows only; fp16 rows are padded to 16 dims)
        const float4 xv = xp[(size_t)c * 64];
#pragma unroll
        for (int j = 0; j < QQ; ++j) {
          const f32x4 qv = cload((const f32x4*)(queries + qoff[j] + 4 * c));
          float t;
          t = xv.x - qv.x; acc[j] = acc[j] + t * t;
          t = xv.y - qv.y; acc[j] = acc[j] + t * t;
          t = xv.z - qv.z; acc[j] = acc[j] + t * t;
          t = xv.w - qv.w; acc[j] = acc[j] + t * t;
        }
      }
    }
    const uint64_t vmask = cload(pool_valid + blk);
    const bool live = (vmask >> lane) & 1ull;
    const uint32_t pos = b * 64 + lane;
#pragma unroll
    for (int j = 0; j < QQ; ++j) {
      const float dist = sqrtf(acc[j]);
      const uint32_t chi = live ? __float_as_uint(dist) : kInf32;
      if (b == b0) {  // empty list: the sorted block IS the list (registers 1.. stay +inf)
        uint32_t shi = chi, slo = live ? pos : kInf32;
        wave_sort64(shi, slo, lane);
        tk[j].hi[0] = shi;
        tk[j].lo[0] = slo;
      } else {
        uint32_t th, tl;
        tk[j].kth(k, th, tl);
        offer<KR>(tk[j], k, chi, pos, th, tl, lane);
      }
    }
  }

#pragma unroll
  for (int j = 0; j < QQ; ++j) {
    if ((uint32_t)j < ne) {
      const u32x2 e = cload(entries + e0 + j);
      const uint32_t slot = ((e.x * nprobe + e.y) * maxsegs + seg) * k;
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const uint32_t el = r * 64 + lane;
        if (el < k) {
          u32x2 v;
          v.x = tk[j].hi[r];
          v.y = tk[j].lo[r];
          part[slot + el] = v;
        }
      }
    }
  }
}

// -----------------------------------------------------------------------------------------
// scan_topk: persistent waves pull (list segment, query group) items from a device work queue.
// -----------------------------------------------------------------------------------------
// ROLE only names the instantiation (0 = coarse ranking over the centroid table, 1 = IVF list scan,
// 2 = exhaustive scan) so that rocprof statistics separate the three uses of the same code.
template <int Q, int KR, int ROLE, int ST>
__global__ __launch_bounds__(256) void scan_topk_kernel(
    const void* __restrict__ pool_data, const uint64_t* __restrict__ pool_valid, const uint32_t d4,
    const uint32_t* __restrict__ list_off, const uint32_t* __restrict__ list_blocks, const uint32_t nlist,
    const uint32_t* __restrict__ entry_off, const uint32_t* __restrict__ item_off,
    const u32x2* __restrict__ entries, const uint32_t* __restrict__ n_items_p, uint32_t* __restrict__ head,
    const float* __restrict__ queries, const uint32_t dpad, const uint32_t segb, const uint32_t k,
    const uint32_t nprobe, const uint32_t maxsegs, u32x2* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const uint32_t n_items = cload(n_items_p);
  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(head, 1u);
    item = rfl(item);
    if (item >= n_items) return;  // every wave reaches this: the queue only grows

    // which list: largest L with item_off[L] <= item (lists with no items are skipped)
    uint32_t lo = 0, hi = nlist;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cload(item_off + mid) <= item) lo = mid; else hi = mid;
    }
    const uint32_t L = lo;
    const uint32_t e_begin = cload(entry_off + L);
    const uint32_t cnt = cload(entry_off + L + 1) - e_begin;
    const uint32_t ngroups = (cnt + Q - 1) / Q;
    const uint32_t local = item - cload(item_off + L);
    const uint32_t seg = local / ngroups, g = local - seg * ngroups;
    const uint32_t b_begin = cload(list_off + L);
    const uint32_t nblk = cload(list_off + L + 1) - b_begin;
    const uint32_t b0 = seg * segb;
    const uint32_t b1 = min(b0 + segb, nblk);
    const uint32_t e0 = e_begin + g * Q;
    const uint32_t ne = min((uint32_t)Q, cnt - g * Q);

    if (ne <= 4)
      scan_item<4, KR, ST>(pool_data, pool_valid, d4, list_blocks, entries, queries, dpad, k, nprobe, maxsegs, part, b_begin,
                       b0, b1, e0, ne, seg, lane);
    else if (ne <= 8)
      scan_item<8, KR, ST>(pool_data, pool_valid, d4, list_blocks, entries, queries, dpad, k, nprobe, maxsegs, part, b_begin,
                       b0, b1, e0, ne, seg, lane);
    else
      scan_item<(Q > 8 ? Q : 8), KR, ST>(pool_data, pool_valid, d4, list_blocks, entries, queries, dpad, k, nprobe, maxsegs,
                                     part, b_begin, b0, b1, e0, ne, seg, lane);
  }
}

// -----------------------------------------------------------------------------------------
// merge: one wave per query folds the per-(rank, segment) partial lists into the final top-k.
// seq = (blocks of earlier-ranked probed lists) * 64 + position  => the reference's scan order.
// -----------------------------------------------------------------------------------------
template <int KR>
__global__ __launch_bounds__(256) void merge_topk_kernel(const MergeArgs m) {
  const int lane = threadIdx.x & 63;
  uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= m.B) return;
  if (m.qlist) {
    if (q >= *m.nq) return;
    q = m.qlist[q];
  }
  WaveTopK<KR> tk;
  tk.init();
  uint32_t th = kInf32, tl = kInf32;
  uint32_t base = 0;
  for (uint32_t r = 0; r < m.nprobe; ++r) {
    const uint32_t L = m.probes ? m.probes[q * m.nprobe + r] : 0u;
    if (L == kInf32) continue;  // "no list" filler
    const uint32_t nblk = m.lists.off[L + 1] - m.lists.off[L];
    const uint32_t nseg = (nblk + m.segb - 1) / m.segb;
    for (uint32_t s = 0; s < nseg; ++s) {
      const uint2* src = m.part + (size_t)((q * m.nprobe + r) * m.maxsegs + s) * m.k;
      for (uint32_t e0 = 0; e0 < m.k; e0 += 64) {
        const uint32_t e = e0 + lane;
        uint32_t chi = kInf32, clo = 0;
        if (e < m.k) {
          const uint2 v = src[e];
          chi = v.x;
          clo = base + v.y;
        }
        offer<KR>(tk, m.k, chi, clo, th, tl, lane);
      }
    }
    base += m.glob_blocks[L] * 64;
  }
  // resolve (seq -> list, position -> caller's row id) and write out
  uint32_t count = 0;
#pragma unroll
  for (int rr = 0; rr < KR; ++rr) {
    const uint32_t e = rr * 64 + lane;
    const uint32_t khi = tk.hi[rr], klo = tk.lo[rr];
    const bool have = e < m.k && khi != kInf32;
    count += __popcll(__ballot(have));
    if (e < m.k) {
      uint64_t id = ~0ull;
      if (have) {
        uint32_t b2 = 0, L = 0;
        for (uint32_t r = 0; r < m.nprobe; ++r) {  // wave-uniform walk, per-lane pick
          const uint32_t Lr = m.probes ? m.probes[q * m.nprobe + r] : 0u;
          if (Lr == kInf32) continue;
          const uint32_t nb = m.glob_blocks[Lr] * 64;
          if (klo >= b2 && klo - b2 < nb) {
            L = Lr;
            break;
          }
          b2 += nb;
        }
        const uint32_t pos = klo - b2;
        const uint32_t blk = m.lists.blocks[m.lists.off[L] + (pos >> 6)];
        id = m.pool.ids[(size_t)blk * 64 + (pos & 63)];
      }
      const size_t o = (size_t)q * m.k + e;
      if (m.out_ids) m.out_ids[o] = id;
      if (m.out_dist) m.out_dist[o] = have ? __uint_as_float(khi) : __uint_as_float(0x7F800000u);
      if (m.out_keys) m.out_keys[o] = ((uint64_t)khi << 32) | klo;
      if (m.out_probes) m.out_probes[o] = have ? (uint32_t)id : kInf32;
    }
  }
  if (lane == 0 && m.out_counts) m.out_counts[q] = count;
}

// G-way merge of per-shard (key, id) partial results; keys are unique across shards.
template <int KR>
__global__ __launch_bounds__(256) void merge_keys_kernel(const uint64_t* __restrict__ keys,
                                                         const uint64_t* __restrict__ ids, uint32_t G, uint32_t B,
                                                         uint32_t k, uint64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_dist,
                                                         uint32_t* __restrict__ out_counts) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= B) return;
  WaveTopK<KR> tk;
  tk.init();
  uint32_t th = kInf32, tl = kInf32;
  for (uint32_t g = 0; g < G; ++g)
    for (uint32_t e0 = 0; e0 < k; e0 += 64) {
      const uint32_t e = e0 + lane;
      uint32_t chi = kInf32, clo = 0;
      if (e < k) {
        const uint64_t key = keys[((size_t)g * B + q) * k + e];
        chi = (uint32_t)(key >> 32);
        clo = (uint32_t)key;
      }
      offer<KR>(tk, k, chi, clo, th, tl, lane);
    }
  uint32_t count = 0;
#pragma unroll
  for (int rr = 0; rr < KR; ++rr) {
    const uint32_t e = rr * 64 + lane;
    const uint32_t khi = tk.hi[rr], klo = tk.lo[rr];
    const bool have = e < k && khi != kInf32;
    count += __popcll(__ballot(have));
    if (e < k) {
      uint64_t id = ~0ull;
      if (have) {  // find the shard entry carrying this key (G*k probes, tiny)
        const uint64_t key = ((uint64_t)khi << 32) | klo;
        for (uint32_t g = 0; g < G && id == ~0ull; ++g)
          for (uint32_t e2 = 0; e2 < k; ++e2) {
            const size_t o = ((size_t)g * B + q) * k + e2;
            if (keys[o] == key) {
              id = ids[o];
              break;
            }
          }
      }
      const size_t o = (size_t)q * k + e;
      out_ids[o] = id;
      out_dist[o] = have ? __uint_as_float(khi) : __uint_as_float(0x7F800000u);
    }
  }
  if (lane == 0 && out_counts) out_counts[q] = count;
}

// -----------------------------------------------------------------------------------------
// plan kernels: probes[B][nprobe] -> per-list entries + work-item prefix sums
// -----------------------------------------------------------------------------------------
// list_off (optional): lists with no blocks on this device are not planned at all — on a rank of a W-rank job that is
// (W-1)/W of the probes of the W*B queries it scans
__global__ void plan_count_kernel(const uint32_t* __restrict__ probes, uint32_t n, uint32_t* __restrict__ cnt,
                                  const uint32_t* __restrict__ list_off = nullptr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t L = probes[i];
    if (L != kInf32 && (!list_off || list_off[L + 1] != list_off[L])) atomicAdd(cnt + L, 1u);
  }
}

// single block of 1024 threads: exclusive scans over lists
__global__ __launch_bounds__(1024) void plan_scan_kernel(const uint32_t* __restrict__ cnt,
                                                         const uint32_t* __restrict__ list_off,
                                                         const uint32_t* __restrict__ list_len, uint32_t nlist,
                                                         uint32_t segb, uint32_t Q, uint32_t* __restrict__ entry_off,
                                                         uint32_t* __restrict__ item_off, uint32_t* __restrict__ fill,
                                                         uint32_t* __restrict__ n_items, uint32_t* __restrict__ head,
                                                         unsigned long long* __restrict__ stats,
                                                         uint32_t* __restrict__ rezero_cnt = nullptr,
                                                         uint32_t lsplit = 0xFFFFFFFFu, uint32_t segb_tail = 0) {
  // lists >= lsplit are cut into segments of segb_tail blocks (the last items of the work queue are small ones)
  __shared__ uint32_t s_e[1024], s_i[1024];
  __shared__ uint32_t carry_e, carry_i;
  __shared__ unsigned long long s_rows, s_touch;
  const uint32_t t = threadIdx.x;
  if (t == 0) {
    carry_e = carry_i = 0;
    s_rows = s_touch = 0;
  }
  __syncthreads();
  for (uint32_t base = 0; base < nlist; base += 1024) {
    const uint32_t L = base + t;
    uint32_t c = 0, it = 0;
    if (L < nlist) {
      c = cnt[L];
      if (rezero_cnt) rezero_cnt[L] = 0;  // consumed: ready for the next plan of the same batch
      const uint32_t nblk = list_off[L + 1] - list_off[L];
      const uint32_t sb = L >= lsplit ? segb_tail : segb;
      const uint32_t nseg = (nblk + sb - 1) / sb;
      it = (c == 0 || nblk == 0) ? 0u : nseg * ((c + Q - 1) / Q);
      fill[L] = 0;
      if (c) {
        atomicAdd(&s_rows, (unsigned long long)c * list_len[L]);
        atomicAdd(&s_touch, (unsigned long long)list_len[L]);
      }
    }
    s_e[t] = c;
    s_i[t] = it;
    __syncthreads();
    for (uint32_t ofs = 1; ofs < 1024; ofs <<= 1) {  // Hillis-Steele inclusive scan
      uint32_t ve = 0, vi = 0;
      if (t >= ofs) {
        ve = s_e[t - ofs];
        vi = s_i[t - ofs];
      }
      __syncthreads();
      s_e[t] += ve;
      s_i[t] += vi;
      __syncthreads();
    }
    if (L < nlist) {
      entry_off[L] = carry_e + s_e[t] - c;
      item_off[L] = carry_i + s_i[t] - it;
    }
    __syncthreads();
    if (t == 1023) {
      carry_e += s_e[1023];
      carry_i += s_i[1023];
    }
    __syncthreads();
  }
  if (t == 0) {
    entry_off[nlist] = carry_e;
    item_off[nlist] = carry_i;
    *n_items = carry_i;
    *head = 0;
    if (stats) {
      stats[0] = s_rows;
      stats[1] = carry_i;
      stats[2] = s_touch;
    }
  }
}

__global__ void plan_fill_kernel(const uint32_t* __restrict__ probes, uint32_t n, uint32_t nprobe,
                                 const uint32_t* __restrict__ entry_off, uint32_t* __restrict__ fill,
                                 uint2* __restrict__ entries, const uint32_t* __restrict__ list_off = nullptr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t L = probes[i];
    if (L != kInf32 && (!list_off || list_off[L + 1] != list_off[L])) {
      const uint32_t p = atomicAdd(fill + L, 1u);
      entries[entry_off[L] + p] = make_uint2(i / nprobe, i % nprobe);
    }
  }
}

// Plan for "every query scans list 0" (coarse ranking over the centroid table, flat scan).
__global__ void plan_all_kernel(uint32_t B, uint32_t nblk, uint32_t segb, uint32_t Q, uint32_t* __restrict__ entry_off,
                                uint32_t* __restrict__ item_off, uint2* __restrict__ entries,
                                uint32_t* __restrict__ n_items, uint32_t* __restrict__ head) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) entries[i] = make_uint2(i, 0u);
  if (i == 0) {
    const uint32_t nseg = (nblk + segb - 1) / segb;
    const uint32_t items = nblk == 0 ? 0u : nseg * ((B + Q - 1) / Q);
    entry_off[0] = 0;
    entry_off[1] = B;
    item_off[0] = 0;
    item_off[1] = items;
    *n_items = items;
    *head = 0;
  }
}

// probes for "probe every list in id order" (exhaustive search / nprobe >= nlist is handled by
// the coarse stage itself; this one serves fvdb_ivf_search_all).
__global__ void probes_all_kernel(uint32_t B, uint32_t nlist, uint32_t* __restrict__ probes) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * nlist) probes[i] = i % nlist;
}

}  // namespace fvdb
