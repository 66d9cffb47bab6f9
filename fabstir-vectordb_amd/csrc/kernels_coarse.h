// kernels_coarse.h — coarse quantizer on the matrix cores, exact by verification.
//
// The centroid ranking of IVFIndex::search_with_config (src/ivf/core.rs:645-656) and
// find_nearest_centroid (:373-386) is a dense B x nlist x d contraction: the one place in this path
// where MFMA is the right tool (north_star).  Rounding differs from the reference's scalar fold, so the
// matrix result is used only to PROPOSE candidates; the answer is then computed with the reference's
// arithmetic:
//   1. coarse_gemm_kernel   A[b][c] ~ |q_b|^2 - 2 q_b.c_c + |c_c|^2   (v_mfma_f32_32x32x2_f32)
//   2. coarse_select_kernel one wave per query: the C = nprobe + margin smallest A (wave top-k), the
//      reference's sequential f32 distance for those C centroids (one lane each), exact top-nprobe in
//      (distance, cluster id) order.  A rounding-error bound proves no centroid outside the C
//      candidates can enter (or tie into) the top nprobe; when the bound does not close, the wave
//      scores every centroid exactly instead (same result, more work).
#pragma once
#include "common.h"
#include "kernels_scan.h"
#include "score_rows.h"

#pragma clang fp contract(off)

namespace fvdb {

typedef float f32x16v __attribute__((ext_vector_type(16)));

// norms[i] = sum_j x[i][j]^2 (sequential; any order would do — only used inside the error-bounded proposal)
__global__ void row_sqnorm_kernel(const float* __restrict__ x, uint32_t stride, uint32_t d, uint32_t n,
                                  float* __restrict__ norms) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = x + (size_t)i * stride;
  float s = 0.0f;
  for (uint32_t j = 0; j < d; ++j) s += r[j] * r[j];
  norms[i] = s;
}

// same, one wave per row (the per-batch query norms: 4 us instead of 60)
__global__ __launch_bounds__(256) void row_sqnorm_wave_kernel(const float* __restrict__ x, uint32_t stride, uint32_t d,
                                                              uint32_t n, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* r = x + (size_t)i * stride;
  float s = 0.0f;
  for (uint32_t j = lane; j < d; j += 64) s = __builtin_fmaf(r[j], r[j], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) norms[i] = s;
}

__global__ void max_f32_kernel(const float* __restrict__ v, uint32_t n, float* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float m = 0.0f;
    for (uint32_t i = 0; i < n; ++i) m = fmaxf(m, v[i]);
    *out = m;
  }
}

// One wave = 32 queries x 64 centroids, K swept 16 dims at a time.  Lane l = (i = l & 31, h = l >> 5) feeds
// row i's dims k0 + 8h .. k0 + 8h + 7 (the pairing of k values inside one MFMA is immaterial to the sum).
// q: [B][dpad], c: [nlist][dpad] (zero padded), dpad % 16 == 0.
__global__ __launch_bounds__(256) void coarse_gemm_kernel(const float* __restrict__ q, const float* __restrict__ c,
                                                          const float* __restrict__ qn, const float* __restrict__ cn,
                                                          uint32_t B, uint32_t nlist, uint32_t dpad,
                                                          float* __restrict__ A /* [B][nlist] */) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t tiles_c = (nlist + 63) / 64;
  const uint32_t tb = wave / tiles_c, tc = wave - tb * tiles_c;
  if (tb * 32 >= B) return;
  const uint32_t i = lane & 31, h = lane >> 5;
  const uint32_t qrow = min(tb * 32 + i, B - 1);
  const uint32_t c0row = min(tc * 64 + i, nlist - 1), c1row = min(tc * 64 + 32 + i, nlist - 1);
  const float4* qp = (const float4*)(q + (size_t)qrow * dpad + 8 * h);
  const float4* c0p = (const float4*)(c + (size_t)c0row * dpad + 8 * h);
  const float4* c1p = (const float4*)(c + (size_t)c1row * dpad + 8 * h);
  f32x16v acc0 = {0}, acc1 = {0};
  for (uint32_t k0 = 0; k0 < dpad; k0 += 16) {
    const float4 qa = qp[k0 / 4], qb = qp[k0 / 4 + 1];
    const float4 a0 = c0p[k0 / 4], b0 = c0p[k0 / 4 + 1];
    const float4 a1 = c1p[k0 / 4], b1 = c1p[k0 / 4 + 1];
    const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    const float v0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
    const float v1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[m], v0[m], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[m], v1[m], acc1, 0, 0, 0);
    }
  }
  // C/D layout of 32x32 tiles: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const uint32_t col0 = tc * 64 + i, col1 = col0 + 32;
  const float cn0 = col0 < nlist ? cn[col0] : 0.0f, cn1 = col1 < nlist ? cn[col1] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t row = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < B) {
      const float qq = qn[row];
      if (col0 < nlist) A[(size_t)row * nlist + col0] = fmaxf(qq + cn0 - 2.0f * acc0[r], 0.0f);
      if (col1 < nlist) A[(size_t)row * nlist + col1] = fmaxf(qq + cn1 - 2.0f * acc1[r], 0.0f);
    }
  }
}

// the reference's distance to one centroid (row-major table, stride dpad): src/core/vector_ops.rs:51-57
__device__ __forceinline__ float exact_centroid_dist(const float* __restrict__ q, const float* __restrict__ crow,
                                                     uint32_t dpad) {
  float acc = 0.0f;
  for (uint32_t j = 0; j < dpad; j += 4) {
    const float4 qv = *(const float4*)(q + j), cv = *(const float4*)(crow + j);
    float t;
    t = qv.x - cv.x; acc = acc + t * t;
    t = qv.y - cv.y; acc = acc + t * t;
    t = qv.z - cv.z; acc = acc + t * t;
    t = qv.w - cv.w; acc = acc + t * t;
  }
  return sqrtf(acc);
}

constexpr int kCoarseRegs = 16;  // clusters per lane the register form of the proposal step holds (nlist <= 1024)

// One wave per query.  C = number of candidates proposed by the approximate matrix (<= 64), kc = clusters kept.
__global__ __launch_bounds__(256) void coarse_select_kernel(const float* __restrict__ A, const float* __restrict__ q,
                                                            const float* __restrict__ c /* [nlist][dpad] */,
                                                            const float* __restrict__ qn,
                                                            const float* __restrict__ cn_max, uint32_t B,
                                                            uint32_t nlist, uint32_t d, uint32_t dpad, uint32_t C,
                                                            uint32_t kc, uint32_t* __restrict__ out_probes,
                                                            float* __restrict__ out_dist,
                                                            uint32_t* __restrict__ n_exact_fallbacks) {
  __shared__ __attribute__((aligned(16))) float s_tile[4][kScoreTileFloats];  // product tiles (score_rows.h)
  const int lane = threadIdx.x & 63;
  const uint32_t b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));  // wave-uniform
  if (b >= B) return;
  const float* arow = A + (size_t)b * nlist;
  const float* qrow = q + (size_t)b * dpad;
  // 1. the C smallest approximate distances, (value, cluster id) order
  WaveTopK<1> prop;
  prop.init();
  uint32_t th = kInf32, tl = kInf32;
  if (nlist <= 64u * kCoarseRegs && C <= 64 && C >= 1) {
    // The whole row sits in registers (lane l holds clusters l, l + 64, ...: one coalesced sweep).  The C-th smallest
    // value is found by bisection on the bit pattern (the values are >= +0: unsigned order is float order), the members
    // below it and, in cluster-id order, as many equal to it as still fit are compacted into one key per lane and sorted —
    // the same C keys, in the same order, as inserting the clusters one by one into a sorted list, for a quarter of the
    // instructions.
    __shared__ uint32_t s_sel[4][128];
    uint32_t* sel = s_sel[threadIdx.x >> 6];
    uint32_t v[kCoarseRegs];
    uint32_t lo = kInf32, hi = 0;
#pragma unroll
    for (int r = 0; r < kCoarseRegs; ++r) {
      const uint32_t cc = (uint32_t)r * 64u + (uint32_t)lane;
      v[r] = cc < nlist ? __float_as_uint(arow[cc]) : kInf32;
      lo = min(lo, v[r]);
      if (cc < nlist) hi = max(hi, v[r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, (uint32_t)__shfl_xor(lo, o));
      hi = max(hi, (uint32_t)__shfl_xor(hi, o));
    }
    auto count_le = [&](uint32_t t) {
      uint32_t cn = 0;
#pragma unroll
      for (int r = 0; r < kCoarseRegs; ++r) cn += __popcll(__ballot(v[r] <= t));
      return cn;
    };
    const uint32_t want = min(C, nlist);
    // smallest t with count(v <= t) >= want: invariant count(lo - 1) < want <= count(hi)
    uint32_t a = lo, b = hi;
    while (a < b) {
      const uint32_t mid = a + ((b - a) >> 1);
      if (count_le(mid) >= want) b = mid;
      else a = mid + 1;
    }
    const uint32_t t = a;
    uint32_t n_lt = 0;
#pragma unroll
    for (int r = 0; r < kCoarseRegs; ++r) n_lt += __popcll(__ballot(v[r] < t));
    uint32_t w_lt = 0, w_eq = n_lt;  // write cursors: the values below t first, then the ties in cluster-id order
#pragma unroll
    for (int r = 0; r < kCoarseRegs; ++r) {
      const uint32_t cc = (uint32_t)r * 64u + (uint32_t)lane;
      const uint64_t ml = __ballot(v[r] < t), me = __ballot(v[r] == t && cc < nlist);
      const uint64_t below = (1ull << lane) - 1ull;
      if (v[r] < t) {
        const uint32_t sl = w_lt + __popcll(ml & below);
        sel[2 * sl] = v[r];
        sel[2 * sl + 1] = cc;
      } else if (v[r] == t && cc < nlist) {
        const uint32_t sl = w_eq + __popcll(me & below);
        if (sl < want) {
          sel[2 * sl] = v[r];
          sel[2 * sl + 1] = cc;
        }
      }
      w_lt += __popcll(ml);
      w_eq += __popcll(me);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t shi = kInf32, slo = kInf32;
    if ((uint32_t)lane < want) {
      shi = sel[2 * lane];
      slo = sel[2 * lane + 1];
    }
    wave_sort64(shi, slo, lane);
    prop.hi[0] = shi;
    prop.lo[0] = slo;
    prop.kth(C, th, tl);
  } else
  for (uint32_t c0 = 0; c0 < nlist; c0 += 64) {
    const uint32_t cc = c0 + lane;
    uint32_t chi = kInf32, clo = cc;
    if (cc < nlist) chi = __float_as_uint(arow[cc]);
    if (c0 == 0) {
      uint32_t shi = chi, slo = cc < nlist ? cc : kInf32;
      wave_sort64(shi, slo, lane);
      prop.hi[0] = shi;
      prop.lo[0] = slo;
      prop.kth(C, th, tl);
    } else {
      offer<1>(prop, C, chi, clo, th, tl, lane);
    }
  }
  const bool all_in = C >= nlist;  // every centroid is a candidate: nothing to prove
  // 2. the reference's distance for the candidates
  // (the whole wave scores them: coalesced loads, products through an LDS tile, per-lane sequential sums — score_rows.h)
  uint32_t ehi = kInf32, elo = kInf32;
  const bool cand = (uint32_t)lane < C && prop.hi[0] != kInf32;
  const uint32_t ncand = __popcll(__ballot(cand));  // the proposals are sorted: the valid ones are lanes 0 .. ncand-1
  const float dex = score_rows_wave(c, dpad, qrow, cand ? prop.lo[0] : 0u, ncand, s_tile[threadIdx.x >> 6], lane);
  if (cand) {
    ehi = __float_as_uint(dex);
    elo = prop.lo[0];
  }
  wave_sort64(ehi, elo, lane);  // ascending (distance, cluster id) = the reference's stable sort order
  // 3. can a centroid outside the candidates reach the top kc?  |approx - reference sum| <= 2 eps, where
  //    eps = 1.01 (d+4) 2^-24 (|q| + max|c|)^2 bounds the rounding of either evaluation.
  bool proven = all_in;
  if (!all_in) {
    const float a_c = __uint_as_float(th);  // C-th smallest approximate value: outsiders are >= this
    const float nq = sqrtf(qn[b]), ncm = sqrtf(cn_max[0]);
    const float eps = 1.01f * (float)(d + 4) * 5.9604645e-8f * (nq + ncm) * (nq + ncm);
    const float lower = sqrtf(fmaxf(a_c - 2.0f * eps, 0.0f)) * (1.0f - 4.8e-7f);  // outsiders' distance >= this
    const uint32_t kk = min(kc, C);
    const float d_k = __uint_as_float(rlane(ehi, kk - 1));
    proven = rlane(ehi, kk - 1) != kInf32 && d_k < lower;
  }
  if (!proven) {  // exact over the whole table (same answer; rare)
    WaveTopK<1> ex;
    ex.init();
    uint32_t xh = kInf32, xl = kInf32;
    for (uint32_t c0 = 0; c0 < nlist; c0 += 64) {
      const uint32_t cc = c0 + lane;
      uint32_t chi = kInf32;
      if (cc < nlist) chi = __float_as_uint(exact_centroid_dist(qrow, c + (size_t)cc * dpad, dpad));
      if (c0 == 0) {
        uint32_t shi = chi, slo = cc < nlist ? cc : kInf32;
        wave_sort64(shi, slo, lane);
        ex.hi[0] = shi;
        ex.lo[0] = slo;
        ex.kth(kc, xh, xl);
      } else {
        offer<1>(ex, kc, chi, cc, xh, xl, lane);
      }
    }
    ehi = ex.hi[0];
    elo = ex.lo[0];
    if (lane == 0 && n_exact_fallbacks) atomicAdd(n_exact_fallbacks, 1u);
  }
  if ((uint32_t)lane < kc) {
    const bool have = ehi != kInf32;
    out_probes[(size_t)b * kc + lane] = have ? elo : kInf32;
    if (out_dist) out_dist[(size_t)b * kc + lane] = have ? __uint_as_float(ehi) : __uint_as_float(0x7F800000u);
  }
}

}  // namespace fvdb
