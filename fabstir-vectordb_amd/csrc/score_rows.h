// score_rows.h — the reference's distance (src/core/vector_ops.rs:51-57, src/hnsw/core.rs:691-697) for a handful of
// arbitrary row-major rows against one query, by ONE wavefront, without each lane walking its own row:
//   * a wave instruction reads 512 contiguous bytes of ONE row (lane l = dims 2l, 2l+1 of a 128-dim block); the rows of
//     a block are all requested at once, the next block's while the current one is folded;
//   * t = q_i - x_i; p = t * t is computed in that layout (order-free), the block's products go through a small LDS tile
//     (row stride 132 floats: lane r's ds_read_b128s are bank-conflict free);
//   * lane r then adds row r's products in dimension order — the reference's running sum, addend for addend, f32, no FMA.
// (q - x)^2 and (x - q)^2 are the same bits, so the result equals the per-lane fold's bit for bit.
// Derivation and measurements: kernels_graph_fast.h (score_fixed).  This form takes the dimension at run time.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

namespace fvdb {

constexpr uint32_t kScoreStride = 132;  // floats per staged row: 128 products + 4 pad
constexpr uint32_t kScoreTileFloats = 16 * kScoreStride;  // one wave's tile (<= 16 rows)

// Distances of query `q` (dpad floats, 8-byte aligned, dpad % 4 == 0) to RC rows: row r (< cnt) is
// rows + readlane(pn, base + r) * dpad.  Returns, in lane r < cnt, sqrt of the reference's sum.  `tile`: kScoreTileFloats
// floats of LDS private to the wave.  Rows past cnt repeat the last one (their sums are ignored).
template <int RC>
__device__ __forceinline__ float score_rows_stream(const float* __restrict__ rows, uint32_t dpad, const float* __restrict__ q, uint32_t pn,
                                                   uint32_t base, uint32_t cnt, float* tile, int lane) {
  const uint32_t nb = (dpad + 127) >> 7;
  const float* rp[RC];
  const uint32_t last = cnt - 1;
#pragma unroll
  for (int r = 0; r < RC; ++r) {
    const uint32_t rr = (uint32_t)r < last ? (uint32_t)r : last;  // wave-uniform
    rp[r] = rows + (size_t)__builtin_amdgcn_readlane(pn, base + rr) * dpad;
  }
  const uint32_t j0 = 2u * (uint32_t)lane;
  auto load = [&](uint32_t c, float2 (&x)[RC], float2& qv) {
    const uint32_t j = c * 128u + j0;
    const bool in = j < dpad;  // dpad % 4 == 0: a pair never straddles the end
    qv = in ? *(const float2*)(q + j) : make_float2(0.0f, 0.0f);
#pragma unroll
    for (int r = 0; r < RC; ++r) x[r] = in ? *(const float2*)(rp[r] + j) : make_float2(0.0f, 0.0f);
  };
  const uint32_t lrow = (uint32_t)lane < (uint32_t)RC ? (uint32_t)lane : (uint32_t)(RC - 1);  // idle lanes add a valid row too
  float acc = 0.0f;
  auto fold = [&](const float2 (&x)[RC], const float2 qv) {
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const float t0 = qv.x - x[r].x, t1 = qv.y - x[r].y;
      *(float2*)(tile + (uint32_t)r * kScoreStride + j0) = make_float2(t0 * t0, t1 * t1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float4* p = (const float4*)(tile + lrow * kScoreStride);
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
  };
  float2 xa[RC], xb[RC], qa, qb;
  load(0, xa, qa);
  for (uint32_t c = 0; c < nb; c += 2) {
    if (c + 1 < nb) load(c + 1, xb, qb);
    fold(xa, qa);
    if (c + 1 < nb) {
      if (c + 2 < nb) load(c + 2, xa, qa);
      fold(xb, qb);
    }
  }
  return sqrtf(acc);
}

// up to 16 rows, the smallest straight-line form that holds them
__device__ __forceinline__ float score_rows_upto16(const float* __restrict__ rows, uint32_t dpad, const float* __restrict__ q, uint32_t pn,
                                                   uint32_t base, uint32_t cnt, float* tile, int lane) {
  if (cnt > 8) return score_rows_stream<16>(rows, dpad, q, pn, base, cnt, tile, lane);
  if (cnt > 4) return score_rows_stream<8>(rows, dpad, q, pn, base, cnt, tile, lane);
  return score_rows_stream<4>(rows, dpad, q, pn, base, cnt, tile, lane);
}

// lane i < n (n <= 64): distance to row rows + pn_i * dpad; chunks of 16 one after the other
__device__ __forceinline__ float score_rows_wave(const float* __restrict__ rows, uint32_t dpad, const float* __restrict__ q, uint32_t pn,
                                                 uint32_t n, float* tile, int lane) {
  float out = 0.0f;
  for (uint32_t base = 0; base < n; base += 16) {
    const uint32_t cnt = min(16u, n - base);
    const float d = score_rows_upto16(rows, dpad, q, pn, base, cnt, tile, lane);
    const float mine = __shfl(d, (int)((uint32_t)lane - base) & 63);
    if ((uint32_t)lane >= base && (uint32_t)lane < base + cnt) out = mine;
  }
  return out;
}

}  // namespace fvdb
