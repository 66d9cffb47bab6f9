// kernels_mfma.h — inverted-list scan on the matrix cores, exact by verification.
//
// The list scan of IVFIndex::search_with_config (src/ivf/core.rs:659-678) is, for the queries that share a
// list, a dense rows x queries x d contraction.  The reference's result is defined by its scalar f32 fold, which
// the matrix cores cannot reproduce bit for bit — so they are used as a FILTER with a proven error bound, and
// every row that could possibly be in the answer is then scored with the reference's arithmetic:
//
//   A. threshold   scan_mfma_kernel MODE 1 evaluates v = |x|^2 - 2 x~.q~ (fp16 MFMA, f32 accumulate) over the
//                  first segment of a probed list with enough rows and keeps the smallest v per row slot;
//                  threshold_kernel takes a_q = the (k+6)-th smallest of those 64 values.  With E_q >=
//                  |v + |q|^2 - (reference's f32 sum)| for EVERY row of the index (input rounding to fp16,
//                  accumulation, norms), k+6 rows have reference sums <= a_q + |q|^2 + E_q.
//   B. filter      scan_mfma_kernel MODE 0 evaluates v for every (row, query) of every probed list; a row
//                  survives iff v <= a_q + 2 E_q.  Every other row has a reference sum > a_q + |q|^2 + E_q.
//   C. select      select_kernel, one wave per query: a'_q = the (k+6)-th smallest v among the survivors
//                  (<= a_q); survivors with v <= a'_q + 2 E_q (some twenty rows) are scored with the reference's
//                  sequential fold and the k smallest (distance, scan position) keys kept.  Every row NOT scored
//                  has a reference sum > S = a'_q + |q|^2 + E_q, so if the k-th kept distance is < sqrt(S)
//                  strictly, no such row can enter or tie: the answer is the reference's.  Otherwise (ties at
//                  the bound, too many survivors, magnitudes outside fp16) the query is rescanned exactly by
//                  fallback_scan_kernel + merge_topk_kernel — same answer, more work, counted in `fallbacks`.
//
// MFMA operand trick: lane = row in the pool layout, so a wave's 16-byte load already IS the A operand of a
// 32x32x16 MFMA for rows 0-31 (k-group 0) and rows 32-63 (k-group 1).  Feeding B = [Q 0; 0 Q] (16 queries)
// makes output column j < 16 the dot products of rows 0-31 and column j >= 16 those of rows 32-63: one
// instruction scores 64 rows x 16 queries x 8 dims with no LDS transposition, at half the MFMA rate — which
// still leaves the kernel bound by the row stream from L2/HBM, not by arithmetic.
#pragma once
#include "score_rows.h"
#include "common.h"
#include "kernels_scan.h"

#pragma clang fp contract(off)

#ifndef FVDB_MFMA_WAVES
#define FVDB_MFMA_WAVES 2  // waves per SIMD the list-scan kernel is compiled for (3 was tried: see DESIGN.md)
#endif

namespace fvdb {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
// two f32 -> packed fp16 pair, round toward zero (one v_cvt_pkrtz_f16_f32)
__device__ __forceinline__ uint32_t pk_rtz(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}
typedef float f32x16m __attribute__((ext_vector_type(16)));

constexpr float kHalfMax = 65504.0f;

// ---------------------------------------------------------------------------------------------
// insert side: |x|^2 per pool row and the running maximum (for the error bound)
// ---------------------------------------------------------------------------------------------
// src rows are the staged row-major f32 inputs; fp16 pools store (and therefore norm) the rounded values.
__global__ void pool_row_norms_kernel(const float* __restrict__ src, uint32_t d, uint64_t n, int f16,
                                      const uint32_t* __restrict__ dst_slot, float* __restrict__ norms,
                                      uint32_t* __restrict__ xmax_bits) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = src + i * d;
  float s = 0.0f;
  for (uint32_t j = 0; j < d; ++j) {
    const float v = f16 ? (float)(_Float16)r[j] : r[j];
    s = __builtin_fmaf(v, v, s);
  }
  norms[dst_slot[i]] = s;
  atomicMax(xmax_bits, __float_as_uint(s));  // s >= 0: float order == unsigned order
}

// ---------------------------------------------------------------------------------------------
// per batch: fp16 copy of the queries (round to nearest even) and |q|^2.  One wave per query.
// ---------------------------------------------------------------------------------------------
// Also clears the per-batch counters of the stages that follow (one launch instead of four memsets):
// cnt[nlist] = 0 (plan), scnt[B + 2] = 0 (survivor counts, nfail, rescan queue head), slots[64 B] = 0xFFFFFFFF.
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, uint32_t B, uint32_t dpad,
                                                           _Float16* __restrict__ qh, float* __restrict__ qn,
                                                           uint32_t* __restrict__ cnt, uint32_t nlist,
                                                           uint32_t* __restrict__ scnt, uint32_t* __restrict__ slots) {
  {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (uint32_t i = gid; i < nlist; i += gsz) cnt[i] = 0;
    for (uint32_t i = gid; i < B + 2; i += gsz) scnt[i] = 0;
    for (uint32_t i = gid; i < 64 * B; i += gsz) slots[i] = 0xFFFFFFFFu;
  }
  const int lane = threadIdx.x & 63;
  const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b > B) return;
  if (b == B) {  // the zero row that inactive MFMA lanes read
    for (uint32_t j = lane; j < dpad; j += 64) qh[(size_t)b * dpad + j] = (_Float16)0.0f;
    return;
  }
  const float* r = q + (size_t)b * dpad;
  float s = 0.0f;
  for (uint32_t j = 2 * lane; j < dpad; j += 128) {
    const float a = r[j], c = r[j + 1];
    s = __builtin_fmaf(a, a, s);
    s = __builtin_fmaf(c, c, s);
    qh[(size_t)b * dpad + j] = (_Float16)a;
    qh[(size_t)b * dpad + j + 1] = (_Float16)c;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) qn[b] = s;
}


// finite floats and -inf keep their order; +inf -> 0xFF800000; NaN -> 0 (sorts first: it gets scored exactly)
__device__ __forceinline__ uint32_t fmap_u32(float v) {
  const uint32_t b = __float_as_uint(v);
  if (v != v) return 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float funmap_u32(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// phase-A list of each query: the first probed list with at least min_rows rows on this device, else the
// longest probed one (kInf32 = no rows at all).  Any list gives a valid bound; a near, full one a tight bound.
__global__ void first_probe_kernel(const uint32_t* __restrict__ probes, uint32_t B, uint32_t np,
                                   const uint32_t* __restrict__ list_len, uint32_t min_rows,
                                   uint32_t* __restrict__ out) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  uint32_t best = kInf32, best_len = 0;
  for (uint32_t r = 0; r < np; ++r) {
    const uint32_t L = probes[(size_t)b * np + r];
    if (L == kInf32) continue;
    const uint32_t len = list_len[L];
    if (len > best_len) {
      best = L;
      best_len = len;
    }
    if (len >= min_rows) break;
  }
  out[b] = best;
}

// ---------------------------------------------------------------------------------------------
// MFMA pass over (list segment, query group) work items.  MODE 0: filter against thr[] -> survivors.
// MODE 1: per (query, segment) the smallest v seen in each of the 64 row slots -> slots[].
// ---------------------------------------------------------------------------------------------
struct MfmaScanArgs {
  const void* pool_data;
  const uint64_t* pool_valid;
  const float* pool_norms;
  uint32_t d4;
  const uint32_t* list_off;
  const uint32_t* list_blocks;
  uint32_t nlist;
  const uint32_t* entry_off;
  const uint32_t* item_off;
  const u32x2* entries;
  const uint32_t* n_items;
  uint32_t* head;
  const _Float16* qh;  // [B + 1][dpad]; row B is all zeros
  uint32_t zero_row;   // = B
  uint32_t dpad, segb;
  // MODE 0
  const float* thr;    // [B]
  uint32_t cmax;
  u32x2* surv;         // [B][cmax]  x = probe rank, y = position in list
  float* sval;         // [B][cmax]  v of the survivor
  uint32_t* scnt;      // [B]
  // MODE 1 (only the first capA blocks of each list are looked at)
  uint32_t* slots;     // [B][64]  smallest v per row slot, order-preserving uint map (0xFFFFFFFF = empty)
  uint32_t capA;
  // 1: the wave that draws group 0 of a list segment scans the segment for ALL its query groups, one after the other
  // (the draws of the other groups are no-ops): the segment comes from HBM once and from this CU's XCD L2 afterwards,
  // instead of once per group through whichever XCDs the groups' waves happen to run on
  uint32_t groups_in_item;
  // dev aid (FVDB_MFMA_STAMPS): per work item 8 x u64 {drawn, located, tile in LDS, done, item, blocks, queries, XCD}
  unsigned long long* stamps;
  uint32_t stamps_cap;
  // workgroup form: lists >= lsplit are cut into segments of segb_tail blocks instead of segb — the queue's last
  // items are small, so the workgroups finish together
  uint32_t lsplit, segb_tail;
};

// 16 dims of row `lane` of block `blk`: the raw 64 (f32) or 32 (fp16) bytes, requested early, and their
// conversion to two MFMA A operands, done late (so the loads of step c+1 fly during the MFMAs of step c)
template <int ST>
struct RowChunk16 {
  float4 v[4];  // ST == 0
  h8v h[2];     // ST == 1
};
template <int ST>
__device__ __forceinline__ void load_a16(const void* __restrict__ pool_data, uint32_t blk, uint32_t d4, uint32_t c4,
                                         int lane, RowChunk16<ST>& r) {
  if (ST == 0) {
    const float4* xp = (const float4*)pool_data + ((size_t)blk * d4 + c4) * 64 + lane;
    r.v[0] = xp[0];
    r.v[1] = xp[64];
    r.v[2] = xp[128];
    r.v[3] = xp[192];
  } else {
    const h8v* xp = (const h8v*)pool_data + ((size_t)blk * (d4 >> 1) + (c4 >> 1)) * 64 + lane;
    r.h[0] = xp[0];
    r.h[1] = xp[64];
  }
}
template <int ST>
__device__ __forceinline__ void operands_a16(const RowChunk16<ST>& r, h8v& a0, h8v& a1) {
  if (ST == 0) {
    a0 = __builtin_bit_cast(h8v, u32x4v{pk_rtz(r.v[0].x, r.v[0].y), pk_rtz(r.v[0].z, r.v[0].w),
                                        pk_rtz(r.v[1].x, r.v[1].y), pk_rtz(r.v[1].z, r.v[1].w)});
    a1 = __builtin_bit_cast(h8v, u32x4v{pk_rtz(r.v[2].x, r.v[2].y), pk_rtz(r.v[2].z, r.v[2].w),
                                        pk_rtz(r.v[3].x, r.v[3].y), pk_rtz(r.v[3].z, r.v[3].w)});
  } else {
    a0 = r.h[0];
    a1 = r.h[1];
  }
}

// End of a block in the filter (MODE 0): v = |x|^2 - 2 acc for the lane's 16 rows x M queries, rows with v <= thr
// survive.  The survivors of a block are appended with ONE atomic per (lane, query) — its count — and plain stores
// behind it: an atomic per survivor made every survivor a round trip to L2 that the whole wave waited for, some
// twenty per block, which cost more than streaming the block.
// D layout: column = lane & 31 (query slot, and which 32-row half), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
struct BlockNorms {
  float4 xn[4];     // |x|^2 of the lane's 16 rows
  uint64_t vmask;   // live rows of the block
};
__device__ __forceinline__ void load_block_norms(const MfmaScanArgs& a, uint32_t blk, uint32_t rowbase, BlockNorms& o) {
  o.vmask = cload(a.pool_valid + blk);
  const float* np = a.pool_norms + (size_t)blk * 64 + rowbase;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) o.xn[r4] = *(const float4*)(np + 8 * r4);
}
template <int M>
__device__ __forceinline__ void emit_survivors(const MfmaScanArgs& a, const f32x16m (&acc)[M], const float (&thr)[M],
                                               const bool (&hasq)[M], const uint32_t (&qidx)[M], const uint32_t (&rnk)[M],
                                               uint32_t b, const BlockNorms& bn, uint32_t rowbase) {
  const uint64_t vmask = bn.vmask;
  float xv[16];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    xv[4 * r4 + 0] = bn.xn[r4].x;
    xv[4 * r4 + 1] = bn.xn[r4].y;
    xv[4 * r4 + 2] = bn.xn[r4].z;
    xv[4 * r4 + 3] = bn.xn[r4].w;
  }
  uint32_t live = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) live |= (uint32_t)((vmask >> (rowbase + 8 * (r >> 2) + (r & 3))) & 1ull) << r;
  uint32_t flags[M], base[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    uint32_t f = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = __builtin_fmaf(-2.0f, acc[m][r], xv[r]);
      f |= (uint32_t)(!(v > thr[m])) << r;  // NaN (non-finite operands) survives
    }
    flags[m] = hasq[m] ? (f & live) : 0u;
  }
#pragma unroll
  for (int m = 0; m < M; ++m) base[m] = flags[m] ? atomicAdd(a.scnt + qidx[m], (uint32_t)__popc(flags[m])) : 0u;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (flags[m] == 0u) continue;
    const size_t row0 = (size_t)qidx[m] * a.cmax;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if ((flags[m] >> r) & 1u) {
        const uint32_t p = base[m] + (uint32_t)__popc(flags[m] & ((1u << r) - 1u));
        if (p < a.cmax) {
          u32x2 sv;
          sv.x = rnk[m];
          sv.y = b * 64 + rowbase + 8 * (r >> 2) + (r & 3);
          a.surv[row0 + p] = sv;
          a.sval[row0 + p] = __builtin_fmaf(-2.0f, acc[m][r], xv[r]);
        }
      }
    }
  }
}

// One work item: blocks [b0, b1) of a list against the ne (<= 16 M) queries of a group.
template <int M, int ST, int MODE>
__device__ __forceinline__ void mfma_item(const MfmaScanArgs& a, const uint32_t b_begin, const uint32_t b0,
                                          const uint32_t b1, const uint32_t e0, const uint32_t ne, const uint32_t seg,
                                          const int lane) {
  const int j = lane & 31, g = lane >> 5, qs = j & 15;
  const bool opnd = g == (j >> 4);  // this lane feeds a non-zero B fragment
  const _Float16* qsrc[M];
  bool qact[M], hasq[M];
  float thr[M];
  uint32_t qidx[M], rnk[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const uint32_t qi = 16 * m + qs;
    const bool has = qi < ne;
    const u32x2 e = a.entries[e0 + (has ? qi : 0)];
    qidx[m] = e.x;
    rnk[m] = e.y;
    hasq[m] = has;
    qact[m] = opnd && has;
    qsrc[m] = a.qh + (size_t)(qact[m] ? e.x : a.zero_row) * a.dpad;  // inactive lanes feed zeros
    thr[m] = (MODE == 0 && has) ? a.thr[e.x] : -__builtin_huge_valf();
  }
  const uint32_t rowbase = ((lane & 16) ? 32u : 0u) + 4u * g;
  float best[MODE == 1 ? M : 1][16];
  if (MODE == 1) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) best[m][r] = __builtin_huge_valf();
  }

  f32x16m acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

  // after the last 16-dim step of block b (list position), blk (pool block): threshold test / slot minima
  auto block_done = [&](uint32_t b, uint32_t blk) {
    if (MODE == 0) {
      BlockNorms bn;
      load_block_norms(a, blk, rowbase, bn);
      emit_survivors<M>(a, acc, thr, hasq, qidx, rnk, b, bn, rowbase);
    } else {
      // D layout: column = lane & 31 (query slot, and which 32-row half), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
      const uint64_t vmask = cload(a.pool_valid + blk);
      const float* np = a.pool_norms + (size_t)blk * 64 + rowbase;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 xn = *(const float4*)(np + 8 * r4);
        const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t row = rowbase + 8 * r4 + t;
          const bool live = (vmask >> row) & 1ull;
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const float v = __builtin_fmaf(-2.0f, acc[m][4 * r4 + t], xv[t]);
            if (live && v < best[m][4 * r4 + t]) best[m][4 * r4 + t] = v;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
  };

  const uint32_t n = a.dpad >> 4;  // 16-dim steps per block
  // ring depth: rows and query fragments are requested D - 1 steps ahead of the MFMAs that consume them.  vmcnt
  // retires loads in issue order, so a step waits for its row chunk from HBM whatever else is in flight: the only
  // way to shorten a step is to have more steps in flight.  The 16-query form has the registers for 8.
  constexpr int D = M == 1 ? 8 : 4;
  if ((n & (D - 1)) == 0) {
    // The item is one stream of steps over (block, chunk), consumed D at a time.  Every load is unconditional
    // (inactive lanes read the zero row, the tail re-requests the last step) so that the wait counts are static and
    // the loads really overlap: HBM latency is ~2 us under load, a step's MFMAs ~50 ns.
    const uint32_t total = (b1 - b0) * n;
    RowChunk16<ST> ring[D];
    h8v qf0[D][M], qf1[D][M];
    uint32_t ib = b0, ic = 0;  // next row step to request
    auto issue_row = [&](RowChunk16<ST>& dst) {
      const bool in = ib < b1;
      const uint32_t blk = cload(a.list_blocks + b_begin + (in ? ib : b1 - 1));
      load_a16<ST>(a.pool_data, blk, a.d4, (in ? ic : n - 1) * 4, lane, dst);
      if (++ic == n) {
        ic = 0;
        ++ib;
      }
    };
    auto issue_q = [&](uint32_t c16, h8v (&d0)[M], h8v (&d1)[M]) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        d0[m] = *(const h8v*)(qsrc[m] + 16 * c16);
        d1[m] = *(const h8v*)(qsrc[m] + 16 * c16 + 8);
      }
    };
#pragma unroll
    for (int u = 0; u < D - 1; ++u) {
      issue_row(ring[u]);
      issue_q((uint32_t)u, qf0[u], qf1[u]);  // n >= D
    }
    uint32_t b = b0, cc = 0;  // step being consumed
    for (uint32_t s = 0; s < total; s += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        issue_row(ring[(u + D - 1) & (D - 1)]);
        const uint32_t cn = cc + u + D - 1;  // n % D == 0 and cc % D == 0: at most one wrap
        issue_q(cn >= n ? cn - n : cn, qf0[(u + D - 1) & (D - 1)], qf1[(u + D - 1) & (D - 1)]);
        h8v a0, a1;
        operands_a16<ST>(ring[u], a0, a1);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf0[u][m], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf1[u][m], acc[m], 0, 0, 0);
        }
      }
      cc += D;
      if (cc == n) {
        cc = 0;
        block_done(b, cload(a.list_blocks + b_begin + b));
        ++b;
      }
    }
  } else {
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t blk = cload(a.list_blocks + b_begin + b);
      for (uint32_t c = 0; c < n; ++c) {
        RowChunk16<ST> rc;
        load_a16<ST>(a.pool_data, blk, a.d4, c * 4, lane, rc);
        h8v a0, a1;
        operands_a16<ST>(rc, a0, a1);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const h8v q0 = *(const h8v*)(qsrc[m] + 16 * c), q1 = *(const h8v*)(qsrc[m] + 16 * c + 8);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, q0, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, q1, acc[m], 0, 0, 0);
        }
      }
      block_done(b, blk);
    }
  }
  if (MODE == 1) {
#pragma unroll
    for (int m = 0; m < M; ++m)
      if (hasq[m]) {
        uint32_t* dst = a.slots + (size_t)qidx[m] * 64 + rowbase;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (best[m][r] < __builtin_huge_valf()) atomicMin(dst + (r & 3) + 8 * (r >> 2), fmap_u32(best[m][r]));
      }
  }
}

template <int M, int ST, int MODE>
__global__ __launch_bounds__(256, FVDB_MFMA_WAVES) void scan_mfma_kernel(const MfmaScanArgs a) {
  const int lane = threadIdx.x & 63;
  constexpr uint32_t Q = 16 * M;
  const uint32_t n_items = cload(a.n_items);
  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(a.head, 1u);
    item = rfl(item);
    if (item >= n_items) return;  // every wave reaches this: the queue only grows
    uint32_t lo = 0, hi = a.nlist;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cload(a.item_off + mid) <= item) lo = mid; else hi = mid;
    }
    const uint32_t L = lo;
    const uint32_t e_begin = cload(a.entry_off + L);
    const uint32_t cnt = cload(a.entry_off + L + 1) - e_begin;
    const uint32_t ngroups = (cnt + Q - 1) / Q;
    const uint32_t local = item - cload(a.item_off + L);
    const uint32_t seg = local / ngroups, g = local - seg * ngroups;
#ifdef FVDB_EXP_HOT  // timing experiment only (wrong results): every segment reads the same 1.5 MB -> all rows L2-hot
    const uint32_t b_begin_true = cload(a.list_off + L);
    const uint32_t nblk = cload(a.list_off + L + 1) - b_begin_true;
    const uint32_t b_begin = 0;
#else
    const uint32_t b_begin = cload(a.list_off + L);
    const uint32_t nblk = cload(a.list_off + L + 1) - b_begin;
#endif
    const uint32_t b0 = seg * a.segb;
    const uint32_t b1 = min(b0 + a.segb, nblk);
    if (MODE == 1 && b0 >= a.capA) continue;
    if (a.groups_in_item && g != 0) continue;
    const uint32_t g_end = a.groups_in_item ? ngroups : g + 1;
    for (uint32_t gg = g; gg < g_end; ++gg) {
      const uint32_t e0 = e_begin + gg * Q;
      const uint32_t ne = min(Q, cnt - gg * Q);
      if (M >= 2 && ne <= 16)
        mfma_item<1, ST, MODE>(a, b_begin, b0, b1, e0, ne, seg, lane);
      else if (M >= 4 && ne <= 32)
        mfma_item<2, ST, MODE>(a, b_begin, b0, b1, e0, ne, seg, lane);
      else
        mfma_item<M, ST, MODE>(a, b_begin, b0, b1, e0, ne, seg, lane);
    }
  }
}

// The reference's distance (src/core/vector_ops.rs:51-57) between the wave-uniform query q and row `rl` of
// pool block `blk` (both may differ per lane).
template <int ST>
__device__ __forceinline__ float exact_row_dist(const void* __restrict__ pool_data, uint32_t d4, uint32_t blk, int rl,
                                                const float* __restrict__ q) {
  float acc = 0.0f;
  uint32_t c = 0;
  for (; c + 4 <= d4; c += 4) {
    float x[16];
    load_rows16<ST>(pool_data, blk, d4, c, rl, x);
    const f32x16 qv = cload16(q + 4 * c);
    float t;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      t = x[i] - qv[i];
      acc = acc + t * t;
    }
  }
  if (ST == 0) {
    const float4* xp = (const float4*)pool_data + (size_t)blk * d4 * 64 + rl;
    for (; c < d4; ++c) {
      const float4 xv = xp[(size_t)c * 64];
      const f32x4 qv = cload((const f32x4*)(q + 4 * c));
      float t;
      t = xv.x - qv.x; acc = acc + t * t;
      t = xv.y - qv.y; acc = acc + t * t;
      t = xv.z - qv.z; acc = acc + t * t;
      t = xv.w - qv.w; acc = acc + t * t;
    }
  }
  return sqrtf(acc);
}

// The same fold over a row stored contiguously (the pool's row-major copy): same values, same order.
__device__ __forceinline__ float exact_row_dist_rm(const float* __restrict__ x, uint32_t d4, const float* __restrict__ q) {
  float acc = 0.0f;
  const float4* xp = (const float4*)x;
  for (uint32_t c = 0; c < d4; ++c) {
    const float4 xv = xp[c];
    const f32x4 qv = cload((const f32x4*)(q + 4 * c));
    float t;
    t = xv.x - qv.x; acc = acc + t * t;
    t = xv.y - qv.y; acc = acc + t * t;
    t = xv.z - qv.z; acc = acc + t * t;
    t = xv.w - qv.w; acc = acc + t * t;
  }
  return sqrtf(acc);
}

// ---------------------------------------------------------------------------------------------
// Error bound and order-preserving float -> uint map shared by the threshold and select kernels.
//   E_q >= | (v + |q|^2) - reference f32 sum | for every row x of the pool:
//     fp16 rounding (2^-11 relative to nearest, 2^-10 toward zero; tiny values <= 2^-14 absolute in case
//     subnormals flush),
//     f32 accumulation inside the MFMA (truncating adds allowed for) and of the norms, the reference's own fold.
//   Returns +inf when magnitudes could leave the fp16 range (the filter then passes every row).
// ---------------------------------------------------------------------------------------------
// x_rounded: how the filter's rows relate to the rows the reference sees: 0 = identical (fp16 storage),
// 1 = rounded to nearest (fp16 mirror of f32 rows), 2 = rounded toward zero (converted on the fly).
// Queries are rounded to nearest (prep_queries_kernel).
__device__ __forceinline__ float mfma_error_bound(float xmax2, float qn2, float d, int x_rounded) {
  const float xm = sqrtf(xmax2) * 1.000001f, nq = sqrtf(qn2) * 1.000001f;
  if (!(xm < kHalfMax && nq < kHalfMax)) return __builtin_huge_valf();
  const float u10 = 9.765625e-4f, u11 = 4.8828125e-4f, u14 = 6.103515625e-5f, u24 = 5.9604645e-8f;
  const float ux = x_rounded == 0 ? 0.0f : (x_rounded == 1 ? u11 : u10);
  const float rnd = ux + u11 + ux * u11;  // relative error of x~.q~ vs x.q
  const float e1 = 2.0f * rnd * xm * nq;
  const float e2 = 2.0f * (2.0f * d * u24) * xm * nq;
  const float e3 = 1.01f * (2.0f * d + 16.0f) * u24 * (xm + nq) * (xm + nq);  // norms + the reference's fold
  const float e4 = 2.0f * u14 * sqrtf(d) * (xm + nq);
  return 1.05f * (e1 + e2 + e3 + e4) + 1e-6f * (xm + nq) * (xm + nq);
}

// ---------------------------------------------------------------------------------------------
// A. threshold: thr_q = a_q + 2 E_q, a_q = ka-th smallest slot minimum.  One wave per query.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void threshold_kernel(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ first,
                                                        const float* __restrict__ qn,
                                                        const uint32_t* __restrict__ xmax_bits, uint32_t B, uint32_t ka,
                                                        uint32_t d, int rows_f16, float* __restrict__ thr_out) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (q >= B) return;
  const float inf = __builtin_huge_valf();
  float out = inf;  // +inf: every row survives (=> the query is rescanned exactly if they are many)
  if (first[q] != kInf32) {
    uint32_t hi = slots[(size_t)q * 64 + lane], lo = (uint32_t)lane;  // kInf32 = empty slot
    wave_sort64(hi, lo, lane);
    const uint32_t kth = rlane(hi, ka - 1);
    const float E = mfma_error_bound(__uint_as_float(*xmax_bits), qn[q], (float)d, rows_f16);
    if (kth != kInf32 && kth != 0u && E < inf) {
      const float a = funmap_u32(kth);
      out = a + 2.0f * E + 1e-6f * fabsf(a);
    }
  }
  if (lane == 0) thr_out[q] = out;
}

// ---------------------------------------------------------------------------------------------
// A (direct form). One wave per query: the first `capA` blocks of a near, well-filled probed list are scored by the
// wave itself — lane = row slot, v = |x|^2 - 2 x~.q~ from the fp16 rows and the fp16 copy of the query with
// v_dot2_f32_f16 (exact products, f32 accumulate: inside the error budget of mfma_error_bound) — the smallest v per
// row slot is kept in a register, the ka-th smallest of the 64 slot minima gives a_q, thr_q = a_q + 2 E_q.
// Replaces first_probe_kernel + a plan + the MODE 1 matrix-core pass + threshold_kernel (six launches, one of them a
// persistent whole-chip kernel of ~0.13 ms that no other batch's scan could share the chip with).
// HALF = true: fp16 rows (the mirror of f32 rows, or fp16 storage); false: f32 rows (no mirror), f32 FMAs.
// ---------------------------------------------------------------------------------------------
struct ThresholdArgs {
  const void* rows;           // pool blocks of the rows the FILTER reads (fp16 mirror / fp16 pool / f32 pool)
  const uint64_t* pool_valid;
  const float* pool_norms;
  uint32_t d4;
  ListTable lists;
  const uint32_t* list_len;   // [nlist] rows per list on this device
  const uint32_t* probes;     // [B][np]
  const _Float16* qh;         // [B + 1][dpad] fp16 queries (prep_queries_kernel)
  const float* queries;       // [B][dpad] f32 queries (HALF = false)
  const float* qn;            // [B] |q|^2
  const uint32_t* xmax_bits;
  uint32_t B, np, ka, dpad, capA, min_rows;
  int rows_f16;               // x_rounded of mfma_error_bound
  float* thr;                 // [B]
  // Sharded search: non-null = [nlist] blocks per list of the LOGICAL index.  The list is then chosen by the global
  // sizes — the same choice on every rank — and only the rank that owns it finds rows there; the others write +inf and
  // the ranks' arrays are combined with a minimum (comm_sharded.h), so each threshold is computed once per step.
  // What is written then is U_q = a_q + E_owner: "k+6 rows of the logical index have reference sums <= U_q + |q|^2";
  // every rank adds the error bound of ITS rows (thr_combine_kernel) — the bound depends on the rank's largest |x|.
  const uint32_t* glob_blocks;
};

template <bool HALF>
__global__ __launch_bounds__(256) void threshold_direct_kernel(const ThresholdArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (q >= a.B) return;
  const float inf = __builtin_huge_valf();
  // the list: the first probed one with at least min_rows rows here, else the longest probed one
  uint32_t best_l = kInf32, best_len = 0;
  for (uint32_t r0 = 0; r0 < a.np && best_len < a.min_rows; r0 += 64) {
    const uint32_t r = r0 + lane;
    uint32_t L = kInf32, len = 0;
    if (r < a.np) {
      L = a.probes[(size_t)q * a.np + r];
      len = L != kInf32 ? (a.glob_blocks ? a.glob_blocks[L] * 64u : a.list_len[L]) : 0u;
    }
    const uint64_t big = __ballot(len >= a.min_rows);
    if (big) {
      const uint32_t l0 = (uint32_t)__builtin_ctzll(big);
      best_l = rlane(L, l0);
      best_len = rlane(len, l0);
      break;
    }
    // longest so far (ties: the earlier rank, as first_probe_kernel's strict '>')
    uint32_t ml = len, mi = (uint32_t)lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t ol = __shfl_xor(ml, o), oi = __shfl_xor(mi, o);
      if (ol > ml || (ol == ml && oi < mi)) {
        ml = ol;
        mi = oi;
      }
    }
    ml = rfl(ml);
    if (ml > best_len) {
      best_len = ml;
      best_l = rlane(L, rfl(mi));
    }
  }
  float out = inf;  // +inf: every row survives
  if (best_l != kInf32 && best_len > 0) {
    const uint32_t b_begin = cload(a.lists.off + best_l);
    const uint32_t nblk = min(cload(a.lists.off + best_l + 1) - b_begin, a.capA);
    float best = inf;
    bool bad = false;
    const uint32_t n8 = a.dpad >> 3;  // 8-dim chunks (dpad % 16 == 0 on this path)
    for (uint32_t b = 0; b < nblk; ++b) {
      const uint32_t blk = cload(a.lists.blocks + b_begin + b);
      float dot = 0.0f;
      if (HALF) {
        const h8v* xp = (const h8v*)a.rows + (size_t)blk * n8 * 64 + lane;
        const _Float16* qrow = a.qh + (size_t)q * a.dpad;
#pragma unroll 4
        for (uint32_t c = 0; c < n8; ++c) {
          const h8v x = xp[(size_t)c * 64];
          const h8v qv = cload((const h8v*)(qrow + 8 * c));  // wave-uniform: scalar loads
          typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int t = 0; t < 4; ++t)
            dot = __builtin_amdgcn_fdot2(h2v{x[2 * t], x[2 * t + 1]}, h2v{qv[2 * t], qv[2 * t + 1]}, dot, false);
        }
      } else {
        const float4* xp = (const float4*)a.rows + (size_t)blk * a.d4 * 64 + lane;
        const float* qrow = a.queries + (size_t)q * a.dpad;
        for (uint32_t c = 0; c < a.d4; ++c) {
          const float4 x = xp[(size_t)c * 64];
          const f32x4 qv = cload((const f32x4*)(qrow + 4 * c));
          dot = __builtin_fmaf(x.x, qv[0], dot);
          dot = __builtin_fmaf(x.y, qv[1], dot);
          dot = __builtin_fmaf(x.z, qv[2], dot);
          dot = __builtin_fmaf(x.w, qv[3], dot);
        }
      }
      const bool live = (cload(a.pool_valid + blk) >> lane) & 1ull;
      const float v = __builtin_fmaf(-2.0f, dot, a.pool_norms[(size_t)blk * 64 + lane]);
      if (live) {
        if (v != v) bad = true;  // non-finite operands: no bound from this list
        else if (v < best) best = v;
      }
    }
    if (!__ballot(bad)) {
      uint32_t hi = best < inf ? fmap_u32(best) : kInf32, lo = (uint32_t)lane;
      wave_sort64(hi, lo, lane);
      const uint32_t kth = rlane(hi, a.ka - 1);
      const float E = mfma_error_bound(__uint_as_float(*a.xmax_bits), a.qn[q], (float)a.dpad, a.rows_f16);
      if (kth != kInf32 && kth != 0u && E < inf) {
        const float av = funmap_u32(kth);
        out = av + (a.glob_blocks ? 1.0f : 2.0f) * E + 1e-6f * fabsf(av);
      }
    }
  }
  if (lane == 0) a.thr[q] = out;
}

// thr[i] = min over ranks of U[w][i] (+inf = "rank w does not own the list") + this rank's error bound for query i:
// a row dropped here has v > U + E_here, hence a reference sum > U + |q|^2, beyond k+6 rows of the logical index.
__global__ void thr_combine_kernel(const float* __restrict__ u_all, uint32_t W, uint32_t n, const float* __restrict__ qn,
                                   const uint32_t* __restrict__ xmax_bits, float d, int rows_f16, float* __restrict__ thr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m = __builtin_huge_valf();
  for (uint32_t w = 0; w < W; ++w) m = fminf(m, u_all[(size_t)w * n + i]);
  thr[i] = m + mfma_error_bound(__uint_as_float(*xmax_bits), qn[i], d, rows_f16);
}
#ifdef FVDB_DEV_TOOLS
// Loopback communicator only (capacity planning, fvdb_comm_create_loopback): the peers that would have supplied the
// thresholds of the queries whose list this rank does not own do not exist, so those entries get a typical one — the
// mean of the known (thr + |q|^2), i.e. a typical squared distance of the (k+6)-th neighbour, minus the query's |q|^2.
__global__ __launch_bounds__(1024) void thr_loopback_fill_kernel(float* __restrict__ thr, const float* __restrict__ qn, uint32_t n) {
  __shared__ float s_sum[1024];
  __shared__ uint32_t s_cnt[1024];
  float sum = 0.0f;
  uint32_t cnt = 0;
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    const float t = thr[i];
    if (t < __builtin_huge_valf()) {
      sum += t + qn[i];
      cnt += 1;
    }
  }
  s_sum[threadIdx.x] = sum;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  for (uint32_t o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
      s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (s_cnt[0] == 0) return;
  const float typ = s_sum[0] / (float)s_cnt[0];
  for (uint32_t i = threadIdx.x; i < n; i += 1024)
    if (!(thr[i] < __builtin_huge_valf())) thr[i] = typ - qn[i];
}
#endif
__global__ void fill_f32_kernel(float* __restrict__ p, uint32_t n, float v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// B'. refine: a query whose survivors outgrew the buffer had a loose threshold (the sampled list did not hold its near
// rows).  The cmax survivors that were stored are actual live rows, so the ka-th smallest v among them is a valid a_q,
// and a far tighter one: the query gets thr = a' + 2 E, an empty survivor buffer and its probes back in probes2[] — the
// filter runs once more over probes2[] (everybody else's row there is all "no list": no work).  One wave per query.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refine_threshold_kernel(const float* __restrict__ sval, uint32_t* __restrict__ scnt,
                                                               const uint32_t* __restrict__ probes, const float* __restrict__ qn,
                                                               const uint32_t* __restrict__ xmax_bits, uint32_t B, uint32_t np,
                                                               uint32_t ka, uint32_t cmax, uint32_t d, int rows_f16,
                                                               float* __restrict__ thr, uint32_t* __restrict__ probes2,
                                                               uint32_t* __restrict__ n_refined) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = rfl(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (q >= B) return;
  const float inf = __builtin_huge_valf();
  bool again = false;
  if (scnt[q] > cmax) {
    WaveTopK<1> ta;
    ta.init();
    uint32_t th = kInf32, tl = kInf32;
    for (uint32_t s0 = 0; s0 < cmax; s0 += 64) {
      const uint32_t s = s0 + lane;
      uint32_t chi = fmap_u32(sval[(size_t)q * cmax + s]), clo = s;
      if (s0 == 0) {
        wave_sort64(chi, clo, lane);
        ta.hi[0] = chi;
        ta.lo[0] = clo;
        ta.kth(ka, th, tl);
      } else {
        offer<1>(ta, ka, chi, clo, th, tl, lane);
      }
    }
    const float E = mfma_error_bound(__uint_as_float(*xmax_bits), qn[q], (float)d, rows_f16);
    if (th != kInf32 && th != 0u && E < inf) {
      const float a = funmap_u32(th);
      const float t2 = a + 2.0f * E + 1e-6f * fabsf(a);
      if (t2 < thr[q]) {
        again = true;
        if (lane == 0) {
          thr[q] = t2;
          scnt[q] = 0;
          if (n_refined) atomicAdd(n_refined, 1u);
        }
      }
    }
  }
  for (uint32_t r = lane; r < np; r += 64) probes2[(size_t)q * np + r] = again ? probes[(size_t)q * np + r] : kInf32;
}

// ---------------------------------------------------------------------------------------------
// C. select
// ---------------------------------------------------------------------------------------------
struct VerifyArgs {
  PoolView pool;
  ListTable lists;
  const uint32_t* probes;       // [B][nprobe]
  const uint32_t* glob_blocks;  // [nlist] blocks per list of the logical index (seq base)
  const float* queries;         // [B][dpad] f32
  const float* qn;              // [B] |q|^2
  const uint32_t* xmax_bits;
  const float* thr;             // [B] filter threshold (+inf = everything passed)
  const u32x2* surv;            // [B][cmax]
  const float* sval;            // [B][cmax]
  const uint32_t* scnt;         // [B]
  uint32_t B, k, ka, nprobe, d, dpad, cmax;
  int rows_f16;
  const float* rows_rm;  // row-major f32 copy of the pool (f32 pools with the fp16 mirror), else null
  // 1: thr[] bounds the (k+6)-th best row of the LOGICAL index and may come from another rank's rows, so fewer than
  // k+6 survivors here is normal: every one of them is scored, and rows filtered out cannot be in the global answer
  int remote_thr;
  uint64_t* out_ids;
  float* out_dist;
  uint32_t* out_counts;
  uint64_t* out_keys;
  uint32_t* fallbacks;   // running counter of queries that were not proven
  uint32_t* reasons;     // [4] of those: survivor buffer overflow, too many candidates for one pass set, the k-th kept
                         // distance not strictly below the bound, no usable threshold
  uint32_t* fail_list;   // [B] queries of this batch to be rescanned exactly (fallback_scan_kernel)
  uint32_t* nfail;       // device scalar, zeroed per batch
};

constexpr uint32_t kSelCand = 256;  // candidates the select stage scores with the reference's fold (four sets of 64)

template <int ST>
__global__ __launch_bounds__(256) void select_kernel(const VerifyArgs m) {
  __shared__ uint32_t s_base[4][256];  // scan-order base (in rows) of each probe rank, per wave
  __shared__ uint32_t s_cand[4][kSelCand];  // survivor indices to score
  __shared__ __attribute__((aligned(16))) float s_tile[4][kScoreTileFloats];  // product tiles (score_rows.h)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t q = rfl(blockIdx.x * 4 + w);
  if (q >= m.B) return;
  const uint32_t np = m.nprobe;  // <= 256 on this path
  const float inf = __builtin_huge_valf();
  // base[r] = 64 * (blocks of the lists ranked before r): exclusive prefix over the probe order
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < np; r0 += 64) {
    const uint32_t r = r0 + lane;
    uint32_t v = 0;
    if (r < np) {
      const uint32_t L = m.probes[(size_t)q * np + r];
      v = L != kInf32 ? m.glob_blocks[L] * 64u : 0u;
    }
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(inc, o);
      if (lane >= o) inc += up;
    }
    if (r < np) s_base[w][r] = carry + inc - v;
    carry += rlane(inc, 63);
  }

  const uint32_t cnt = m.scnt[q];
  const float E = mfma_error_bound(__uint_as_float(*m.xmax_bits), m.qn[q], (float)m.d, m.rows_f16);
  bool proven = cnt <= m.cmax;
  // a' = ka-th smallest v among the survivors (key = mapped v, survivor index)
  uint32_t ath = kInf32;
  if (proven && cnt > 0) {
    WaveTopK<1> ta;
    ta.init();
    uint32_t th = kInf32, tl = kInf32;
    for (uint32_t s0 = 0; s0 < cnt; s0 += 64) {
      const uint32_t s = s0 + lane;
      uint32_t chi = kInf32, clo = s;
      if (s < cnt) chi = fmap_u32(m.sval[(size_t)q * m.cmax + s]);
      if (s0 == 0) {
        uint32_t slo = s < cnt ? s : kInf32;
        wave_sort64(chi, slo, lane);
        ta.hi[0] = chi;
        ta.lo[0] = slo;
        ta.kth(m.ka, th, tl);
      } else {
        offer<1>(ta, m.ka, chi, clo, th, tl, lane);
      }
    }
    ath = th;  // kInf32 when fewer than ka rows survived
  }
  // candidates: v <= a' + 2E (every survivor when fewer than ka survived); all must fit one wave
  float cut = inf;
  if (ath != kInf32 && ath != 0u && E < inf) {
    const float a = funmap_u32(ath);
    cut = a + 2.0f * E + 1e-6f * fabsf(a);
    // never above the filter's own threshold: rows dropped in B only satisfy v > thr, and the bound below is stated
    // for "every unscored row has v > cut" (a' <= a_q makes this a no-op unless the two passes rounded differently)
    cut = fminf(cut, m.thr[q]);
  }
  uint32_t reason = proven ? 0u : 1u;  // 1: the survivor buffer overflowed
  uint32_t ncand = 0;
  if (proven) {
    for (uint32_t s0 = 0; s0 < cnt && ncand <= kSelCand; s0 += 64) {
      const uint32_t s = s0 + lane;
      const bool c = s < cnt && !(m.sval[(size_t)q * m.cmax + s] > cut);
      const uint64_t mask = __ballot(c);
      const uint32_t slot = ncand + __popcll(mask & ((1ull << lane) - 1ull));
      if (c && slot < kSelCand) s_cand[w][slot] = s;
      ncand += __popcll(mask);
    }
    if (ncand > kSelCand) {
      proven = false;
      reason = 2;
    }
  }
  uint32_t khi = kInf32, klo = kInf32;
  if (proven) {
    // the candidates are scored 64 at a time (16 rows per pass of the wave, score_rows.h); the min(k, 64) best keys are
    // kept across the sets in a wave-distributed sorted list
    const float* qrow = m.queries + (size_t)q * m.dpad;
    WaveTopK<1> best;
    best.init();
    uint32_t bh = kInf32, bl = kInf32;
    const uint32_t keep = min(m.k, 64u);
    for (uint32_t c0 = 0; c0 < ncand || c0 == 0; c0 += 64) {
      const uint32_t nset = min(64u, ncand - c0);
      u32x2 sv = {0u, 0u};
      uint32_t blk = 0, chi = kInf32, clo = kInf32;
      if ((uint32_t)lane < nset) {
        sv = m.surv[(size_t)q * m.cmax + s_cand[w][c0 + lane]];
        const uint32_t L = m.probes[(size_t)q * np + sv.x];
        blk = m.lists.blocks[m.lists.off[L] + (sv.y >> 6)];
        clo = s_base[w][sv.x] + sv.y;
      }
      if (nset > 0) {
        if (ST == 0 && m.rows_rm) {
          // rows from the row-major copy, scored by the whole wave: coalesced loads, products through an LDS tile, per-lane
          // sequential sums — one memory latency per 16 rows where each lane walking its own 1.5 KB row took ~25 dependent ones
          const float dsel = score_rows_wave(m.rows_rm, m.dpad, qrow, blk * 64u + (sv.y & 63u), nset, s_tile[w], lane);
          if ((uint32_t)lane < nset) chi = __float_as_uint(dsel);
        } else if ((uint32_t)lane < nset) {
          chi = __float_as_uint(exact_row_dist<ST>(m.pool.data, m.pool.d4, blk, (int)(sv.y & 63), qrow));
        }
      }
      if (c0 == 0) {
        wave_sort64(chi, clo, lane);
        best.hi[0] = chi;
        best.lo[0] = clo;
        best.kth(keep, bh, bl);
      } else {
        offer<1>(best, keep, chi, clo, bh, bl, lane);
      }
      if (ncand == 0) break;
    }
    khi = best.hi[0];
    klo = best.lo[0];
    // Rows not scored: filtered out in B (v > thr >= cut) or pruned here (v > cut).  Their reference sums exceed
    // S = cut - E + |q|^2 (>= a' + |q|^2 + E), so their distances are >= sqrt(S): the k-th kept must be strictly
    // below.  cut == +inf means every live probed row was scored: nothing to prove.
    if (cut < inf) {
      const float S = ((cut - E) + m.qn[q]) * 0.999999f;
      const uint32_t dk = rlane(khi, keep - 1);
      proven = dk != kInf32 && S > 0.0f && __uint_as_float(dk) < sqrtf(S);
      if (!proven) reason = 3;
    } else {
      // a finite threshold from this device's own rows leaves >= ka survivors; fewer means something is off
      proven = m.remote_thr || m.thr[q] == inf;
      if (!proven) reason = 4;
    }
  }
  if (!proven) {  // hand the query to the exact rescan (fallback_scan_kernel + merge over the fail list)
    if (lane == 0) {
      m.fail_list[atomicAdd(m.nfail, 1u)] = q;
      if (m.fallbacks) atomicAdd(m.fallbacks, 1u);
      if (m.reasons && reason) atomicAdd(m.reasons + (reason - 1), 1u);
    }
    return;
  }
  // resolve (seq -> list, position -> caller's row id) and write out
  const bool have = (uint32_t)lane < m.k && khi != kInf32;
  const uint32_t count = __popcll(__ballot(have));
  if ((uint32_t)lane < m.k) {
    uint64_t id = ~0ull;
    if (have) {
      uint32_t r = 0;
      while (r + 1 < np && s_base[w][r + 1] <= klo) ++r;
      // ranks whose list is empty share a base with their successor: the last of them owns the rows
      const uint32_t L = m.probes[(size_t)q * np + r];
      const uint32_t pos = klo - s_base[w][r];
      const uint32_t blk = m.lists.blocks[m.lists.off[L] + (pos >> 6)];
      id = m.pool.ids[(size_t)blk * 64 + (pos & 63)];
    }
    const size_t o = (size_t)q * m.k + lane;
    if (m.out_ids) m.out_ids[o] = id;
    if (m.out_dist) m.out_dist[o] = have ? __uint_as_float(khi) : __uint_as_float(0x7F800000u);
    if (m.out_keys) m.out_keys[o] = ((uint64_t)khi << 32) | klo;
  }
  if (lane == 0 && m.out_counts) m.out_counts[q] = count;
}

// ---------------------------------------------------------------------------------------------
// exact rescan of the queries the verify stage could not prove: persistent waves over
// (failed query, probe rank, list segment) items, one query per item, the reference's fold for every row.
// Partial lists go where the exact scan (kernels_scan.h) puts them, so merge_topk_kernel finishes the job.
// ---------------------------------------------------------------------------------------------
struct FallbackArgs {
  PoolView pool;
  ListTable lists;
  const uint32_t* probes;  // [B][nprobe]
  const float* queries;    // [B][dpad]
  const uint32_t* fail_list;
  const uint32_t* nfail;
  uint32_t* head;          // work-queue head (zeroed per batch)
  uint32_t k, nprobe, dpad, segb, maxsegs;
  u32x2* part;             // [B][nprobe][maxsegs][k]
};

template <int ST>
__global__ __launch_bounds__(256) void fallback_scan_kernel(const FallbackArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t nfail = cload(a.nfail);
  if (nfail == 0) return;  // the usual case: nothing to do, not even the queue atomic
  const uint32_t per_q = a.nprobe * a.maxsegs;
  const uint32_t total = nfail * per_q;
  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(a.head, 1u);
    item = rfl(item);
    if (item >= total) return;  // every wave reaches this
    const uint32_t fi = item / per_q, rem = item - fi * per_q;
    const uint32_t r = rem / a.maxsegs, seg = rem - r * a.maxsegs;
    const uint32_t q = cload(a.fail_list + fi);
    const uint32_t L = a.probes[(size_t)q * a.nprobe + r];
    if (L == kInf32) continue;
    const uint32_t bb = a.lists.off[L], nb = a.lists.off[L + 1] - bb;
    const uint32_t b0 = seg * a.segb;
    if (b0 >= nb) continue;
    const uint32_t b1 = min(b0 + a.segb, nb);
    const float* qrow = a.queries + (size_t)q * a.dpad;
    WaveTopK<1> tk;
    tk.init();
    uint32_t th = kInf32, tl = kInf32;
    for (uint32_t b = b0; b < b1; ++b) {
      const uint32_t blk = a.lists.blocks[bb + b];
      const float dist = exact_row_dist<ST>(a.pool.data, a.pool.d4, blk, lane, qrow);
      const bool live = (a.pool.valid[blk] >> lane) & 1ull;
      uint32_t chi = live ? __float_as_uint(dist) : kInf32;
      uint32_t clo = live ? b * 64 + lane : kInf32;
      if (b == b0) {
        wave_sort64(chi, clo, lane);
        tk.hi[0] = chi;
        tk.lo[0] = clo;
        tk.kth(a.k, th, tl);
      } else {
        offer<1>(tk, a.k, chi, clo, th, tl, lane);
      }
    }
    if ((uint32_t)lane < a.k) {
      u32x2 v;
      v.x = tk.hi[0];
      v.y = tk.lo[0];
      a.part[((size_t)(q * a.nprobe + r) * a.maxsegs + seg) * a.k + lane] = v;
    }
  }
}

}  // namespace fvdb
