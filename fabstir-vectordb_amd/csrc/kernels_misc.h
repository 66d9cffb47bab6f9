// kernels_misc.h — layout conversion, candidate scoring (HNSW hops) and k-means reductions.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

namespace fvdb {

// Row-major [n][d] (device) -> blocked pool slots.  One thread per (row, 4-dim chunk).
__global__ void scatter_rows_kernel(const float* __restrict__ src, uint32_t d, uint32_t d4, uint64_t n,
                                    const uint32_t* __restrict__ dst_slot, const uint64_t* __restrict__ row_ids,
                                    float4* __restrict__ pool_data, uint64_t* __restrict__ pool_ids,
                                    unsigned long long* __restrict__ pool_valid, void* __restrict__ pool_half,
                                    float4* __restrict__ pool_rm) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t i = t / d4;
  const uint32_t c = (uint32_t)(t % d4);
  if (i >= n) return;
  const uint32_t slot = dst_slot[i];
  const uint32_t blk = slot >> 6, lane = slot & 63;
  const float* r = src + i * d + 4 * c;
  float4 v;
  v.x = (4 * c + 0 < d) ? r[0] : 0.0f;
  v.y = (4 * c + 1 < d) ? r[1] : 0.0f;
  v.z = (4 * c + 2 < d) ? r[2] : 0.0f;
  v.w = (4 * c + 3 < d) ? r[3] : 0.0f;
  pool_data[((size_t)blk * d4 + c) * 64 + lane] = v;
  if (pool_rm) pool_rm[((size_t)blk * 64 + lane) * d4 + c] = v;  // row-major copy (select stage)
  if (pool_half) {  // fp16 mirror (round to nearest even) in the fp16 pool layout: 8 halves per 16-byte chunk
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    *(h4*)((char*)pool_half + (((size_t)blk * (d4 >> 1) + (c >> 1)) * 64 + lane) * 16 + (c & 1) * 8) = hv;
  }
  if (c == 0) {
    pool_ids[(size_t)blk * 64 + lane] = row_ids ? row_ids[i] : i;
    atomicOr(pool_valid + blk, 1ull << lane);
  }
}

// Same for fp16 row storage: one thread per (row, 8-dim chunk), f32 -> fp16 round-to-nearest-even.
__global__ void scatter_rows_f16_kernel(const float* __restrict__ src, uint32_t d, uint32_t d8, uint64_t n,
                                        const uint32_t* __restrict__ dst_slot, const uint64_t* __restrict__ row_ids,
                                        void* __restrict__ pool_data, uint64_t* __restrict__ pool_ids,
                                        unsigned long long* __restrict__ pool_valid) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t i = t / d8;
  const uint32_t c = (uint32_t)(t % d8);
  if (i >= n) return;
  const uint32_t slot = dst_slot[i];
  const uint32_t blk = slot >> 6, lane = slot & 63;
  h8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t j = 8 * c + e;
    v[e] = (_Float16)(j < d ? src[i * d + j] : 0.0f);
  }
  ((h8*)pool_data)[((size_t)blk * d8 + c) * 64 + lane] = v;
  if (c == 0) {
    pool_ids[(size_t)blk * 64 + lane] = row_ids ? row_ids[i] : i;
    atomicOr(pool_valid + blk, 1ull << lane);
  }
}

// Soft delete / undelete: slots = block*64 + lane.
__global__ void set_valid_kernel(const uint32_t* __restrict__ slots, uint64_t n, int deleted,
                                 unsigned long long* __restrict__ pool_valid) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t blk = slots[i] >> 6, lane = slots[i] & 63;
  if (deleted)
    atomicAnd(pool_valid + blk, ~(1ull << lane));
  else
    atomicOr(pool_valid + blk, 1ull << lane);
}

// [n][d] -> [n][dpad] zero padded (queries whose d is not a multiple of 4; store rows).
__global__ void pad_rows_kernel(const float* __restrict__ src, uint32_t d, uint32_t dpad, uint64_t n,
                                float* __restrict__ dst) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * dpad) return;
  const uint64_t i = t / dpad;
  const uint32_t j = (uint32_t)(t % dpad);
  dst[t] = j < d ? src[i * d + j] : 0.0f;
}

// queries[b] = rows[idx[b]] (both [.][dpad])
__global__ void gather_rows_kernel(const float* __restrict__ rows, const uint32_t* __restrict__ idx, uint32_t dpad,
                                   uint32_t B, float* __restrict__ dst) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)B * dpad) return;
  const uint32_t b = (uint32_t)(t / dpad), j = (uint32_t)(t % dpad);
  dst[t] = rows[(size_t)idx[b] * dpad + j];
}

// Candidate scoring for graph hops: lane = one (query, candidate) pair, sequential f32 fold like
// src/hnsw/core.rs:691-697.  cand may live in pinned host memory mapped into the GPU.
__global__ __launch_bounds__(256) void score_candidates_kernel(const float* __restrict__ rows, uint32_t dpad,
                                                               const float* __restrict__ queries,
                                                               const uint32_t* __restrict__ cand, uint32_t B,
                                                               uint32_t C, uint32_t stride, float* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * C) return;
  const uint32_t b = t / C, c = t - b * C;
  const uint32_t row = cand[(size_t)b * stride + c];
  float res = __uint_as_float(0x7F800000u);
  if (row != 0xFFFFFFFFu) {
    const float4* xp = (const float4*)(rows + (size_t)row * dpad);
    const float4* qp = (const float4*)(queries + (size_t)b * dpad);
    float acc = 0.0f;
    for (uint32_t i = 0; i < dpad / 4; ++i) {
      const float4 x = xp[i], q = qp[i];
      float u;
      u = q.x - x.x; acc = acc + u * u;
      u = q.y - x.y; acc = acc + u * u;
      u = q.z - x.z; acc = acc + u * u;
      u = q.w - x.w; acc = acc + u * u;
    }
    res = sqrtf(acc);
  }
  out[(size_t)b * stride + c] = res;
}

// ---- a3 utilities: batched dot product / cosine similarity, B queries x n rows -------------------
// dot_product_scalar (src/core/vector_ops.rs:35-37) is a left-to-right f32 fold of products;
// cosine_similarity_scalar (:39-49) = dot / (sqrt(dot(a,a)) * sqrt(dot(b,b))), 0 when a norm is 0.
// One lane per (query, row) pair folds sequentially, so results equal the reference's bit for bit.
__global__ __launch_bounds__(256) void dot_cosine_kernel(const float* __restrict__ q, const float* __restrict__ x,
                                                         uint32_t B, uint64_t n, uint32_t d, int cosine,
                                                         float* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)B * n) return;
  const uint32_t b = (uint32_t)(t / n);
  const uint64_t r = t - (uint64_t)b * n;
  const float* a = q + (size_t)b * d;
  const float* c = x + r * d;
  float dot = 0.0f, na = 0.0f, nb = 0.0f;
  for (uint32_t j = 0; j < d; ++j) {
    dot = dot + a[j] * c[j];
    if (cosine) {
      na = na + a[j] * a[j];
      nb = nb + c[j] * c[j];
    }
  }
  float res = dot;
  if (cosine) {
    const float sa = sqrtf(na), sb = sqrtf(nb);
    res = (sa == 0.0f || sb == 0.0f) ? 0.0f : dot / (sa * sb);
  }
  out[t] = res;
}

// ---- k-means (src/ivf/core.rs:388-429): sums in data order, like the reference --------------
// One block per cluster; thread t owns dims t, t+256, ...; every thread walks the assignment
// array (wave-uniform branch) and adds member rows in data order.
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float* __restrict__ x, uint32_t d, uint64_t n,
                                                            const uint32_t* __restrict__ assign,
                                                            float* __restrict__ centroids /* [nlist][d] */) {
  const uint32_t c = blockIdx.x;
  constexpr int MAXD = 8;  // d <= 2048
  float sum[MAXD];
#pragma unroll
  for (int s = 0; s < MAXD; ++s) sum[s] = 0.0f;
  uint64_t count = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (assign[i] == c) {
      ++count;
#pragma unroll
      for (int s = 0; s < MAXD; ++s) {
        const uint32_t j = threadIdx.x + s * 256;
        if (j < d) sum[s] += x[i * d + j];
      }
    }
  }
  if (count > 0) {
#pragma unroll
    for (int s = 0; s < MAXD; ++s) {
      const uint32_t j = threadIdx.x + s * 256;
      if (j < d) centroids[(size_t)c * d + j] = sum[s] / (float)count;
    }
  }
}

// dist[i] = L2(x_i, centroid[assign_i]) (sequential fold per lane)
__global__ void kmeans_point_dist_kernel(const float* __restrict__ x, uint32_t d, uint64_t n,
                                         const uint32_t* __restrict__ assign, const float* __restrict__ centroids,
                                         float* __restrict__ dist) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* a = x + i * d;
  const float* b = centroids + (size_t)assign[i] * d;
  float acc = 0.0f;
  for (uint32_t j = 0; j < d; ++j) {
    const float t = a[j] - b[j];
    acc = acc + t * t;
  }
  dist[i] = sqrtf(acc);
}

// Sequential f32 sums in index order (the reference's folds) without paying a global-memory latency per
// element: the block stages 8K-element tiles in LDS with coalesced loads, thread 0 folds each tile.
constexpr int kSeqTile = 8192;

// total = sum_i dist_i*dist_i in index order, out[0] = total / n, out[1] = total   (src/ivf/core.rs:419-429)
// Left-to-right f32 sum of m squares staged in LDS (already squared by all threads in parallel: the products do
// not depend on the running sum), 16 values per trip through four 16-byte LDS reads.  One thread; same additions in
// the same order as a scalar loop.
__device__ __forceinline__ float seq_add16(const float* __restrict__ sq, uint32_t m, float total) {
  uint32_t i = 0;
  for (; i + 16 <= m; i += 16) {
    const float4 a = *(const float4*)(sq + i), b = *(const float4*)(sq + i + 4), c = *(const float4*)(sq + i + 8),
                 d = *(const float4*)(sq + i + 12);
    total += a.x; total += a.y; total += a.z; total += a.w;
    total += b.x; total += b.y; total += b.z; total += b.w;
    total += c.x; total += c.y; total += c.z; total += c.w;
    total += d.x; total += d.y; total += d.z; total += d.w;
  }
  for (; i < m; ++i) total += sq[i];
  return total;
}

__global__ __launch_bounds__(256) void seq_sqsum_mean_kernel(const float* __restrict__ dist, uint64_t n,
                                                             float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[kSeqTile];
  float total = 0.0f;
  for (uint64_t base = 0; base < n; base += kSeqTile) {
    const uint32_t m = (uint32_t)min((uint64_t)kSeqTile, n - base);
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
      const float v = dist[base + i];
      tile[i] = v * v;
    }
    __syncthreads();
    if (threadIdx.x == 0) total = seq_add16(tile, m, total);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = total / (float)n;
    out[1] = total;
  }
}

// k-means++ helpers (src/ivf/core.rs:336-371)
// mind[j] = min(mind[j], L2(x_j, x_pick))
__global__ void kpp_min_dist_kernel(const float* __restrict__ x, uint32_t d, uint64_t n, uint64_t pick,
                                    float* __restrict__ mind) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* a = x + i * d;
  const float* b = x + pick * d;
  float acc = 0.0f;
  for (uint32_t j = 0; j < d; ++j) {
    const float t = a[j] - b[j];
    acc = acc + t * t;
  }
  mind[i] = fminf(mind[i], sqrtf(acc));
}
// sequential: total = sum d^2; threshold = u * total; first j with cumulative >= threshold (:357-367)
__global__ __launch_bounds__(256) void kpp_pick_kernel(const float* __restrict__ mind, uint64_t n, float u,
                                                       unsigned long long* out_pick) {
  __shared__ __attribute__((aligned(16))) float tile[kSeqTile];
  __shared__ float s_threshold;
  __shared__ unsigned long long s_pick;
  float total = 0.0f;
  for (uint64_t base = 0; base < n; base += kSeqTile) {
    const uint32_t m = (uint32_t)min((uint64_t)kSeqTile, n - base);
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
      const float v = mind[base + i];
      tile[i] = v * v;
    }
    __syncthreads();
    if (threadIdx.x == 0) total = seq_add16(tile, m, total);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    s_threshold = u * total;
    s_pick = ~0ull;
  }
  __syncthreads();
  float cumulative = 0.0f;
  for (uint64_t base = 0; base < n; base += kSeqTile) {
    if (s_pick != ~0ull) break;  // uniform: written before the barrier below
    const uint32_t m = (uint32_t)min((uint64_t)kSeqTile, n - base);
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
      const float v = mind[base + i];
      tile[i] = v * v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float threshold = s_threshold;
      // 16 at a time while the threshold is out of reach of this group (partial sums only grow: squares are >= 0),
      // then one by one to find the first index that reaches it
      uint32_t i = 0;
      for (; i + 16 <= m; i += 16) {
        const float after = seq_add16(tile + i, 16, cumulative);
        if (after >= threshold) break;
        cumulative = after;
      }
      for (; i < m; ++i) {
        cumulative += tile[i];
        if (cumulative >= threshold) {
          s_pick = base + i;
          break;
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_pick = s_pick;
}
__global__ void fill_f32_kernel(float* p, uint64_t n, float v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void count_changed_kernel(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, uint64_t n,
                                     uint32_t* __restrict__ changed) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (a[i] != b[i]) {
    b[i] = a[i];
    atomicAdd(changed, 1u);
  }
}

}  // namespace fvdb
