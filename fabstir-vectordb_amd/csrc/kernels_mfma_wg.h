// kernels_mfma_wg.h — the list-scan filter (kernels_mfma.h, stage B) with the query group held in LDS.
//
// Same arithmetic as scan_mfma_kernel<2, 1, 0> — the same MFMA instructions on the same operands in the same order, so
// the same v for every (row, query) and the same survivors — organised around what bounded that kernel: its waves
// waited on memory three quarters of the time, every 16-dim step pulled 64 B of query fragments per lane through the
// vector L1 next to 32 B of rows, and the fragments held in flight took twice the registers of the rows.  Here
//   * a work item (list segment, query group) belongs to a WORKGROUP: its four waves copy the group's fp16 queries
//     into LDS once (32 queries: 25 KB; 64: 50 KB), then take the segment's blocks round-robin; with the fragments out
//     of the registers a wave can afford 64 queries per row chunk, so a list probed by 33..64 queries is streamed once
//     instead of twice;
//   * the B operand comes from LDS one step ahead of its MFMAs (ds_read_b128; lanes of the zero half of [Q 0; 0 Q]
//     read a zero row, so the loop has no per-step selects);
//   * the registers that held query fragments now hold rows: 8 steps (16 KB per wave) in flight instead of 4.
// fp16 rows (stored or mirrored) and dpad % 128 == 0 only; every other shape keeps scan_mfma_kernel.
#pragma once
#include "kernels_mfma.h"

#ifndef FVDB_MFMA_WG_WAVES
#define FVDB_MFMA_WG_WAVES 2
#endif

namespace fvdb {

#ifndef FVDB_MFMA_WG_RING
#define FVDB_MFMA_WG_RING 8
#endif
constexpr int kWgRing = FVDB_MFMA_WG_RING;  // row steps in flight per wave
__host__ __device__ inline uint32_t mfma_wg_qstride(uint32_t dpad) { return dpad * 2u + 16u; }  // bytes; +16: bank skew
__host__ __device__ inline uint32_t mfma_wg_lds_bytes(uint32_t dpad, uint32_t Q) { return (Q + 1u) * mfma_wg_qstride(dpad) + 16u; }

// blocks b0 + w, b0 + w + 4, ... < b1 of the list against the ne (<= 16 M) queries in the LDS tile
template <int M>
__device__ __forceinline__ void mfma_item_wg(const MfmaScanArgs& a, const unsigned char* __restrict__ qt,
                                             const uint32_t qstride, const uint32_t b_begin, const uint32_t bw,
                                             const uint32_t b1, const uint32_t e0, const uint32_t ne, const uint32_t zero_row,
                                             const int lane) {
  constexpr int D = M >= 3 ? 4 : kWgRing;  // row steps in flight (48- and 64-query forms: the accumulators take the registers)
  const int j = lane & 31, g = lane >> 5, qs = j & 15;
  const bool opnd = g == (j >> 4);
  bool hasq[M];
  float thr[M];
  uint32_t qidx[M], rnk[M], qoff[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const uint32_t qi = 16 * m + qs;
    const bool has = qi < ne;
    const u32x2 e = a.entries[e0 + (has ? qi : 0)];
    qidx[m] = e.x;
    rnk[m] = e.y;
    hasq[m] = has;
    qoff[m] = ((opnd && has) ? qi : zero_row) * qstride;  // the tile's last row is all zeros
    thr[m] = has ? a.thr[e.x] : -__builtin_huge_valf();
  }
  const uint32_t rowbase = ((lane & 16) ? 32u : 0u) + 4u * g;
  f32x16m acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

  // |x|^2 and the live mask of the block being consumed are requested one ring ahead of its last step: vmcnt retires
  // in order, so asking for them at the end of the block would first drain every row step in flight
  BlockNorms bn;
  auto block_done = [&](uint32_t b) {
    emit_survivors<M>(a, acc, thr, hasq, qidx, rnk, b, bn, rowbase);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
  };

  if (bw >= b1) return;
  const uint32_t n = a.dpad >> 4;                      // 16-dim steps per block; n % D == 0 (checked at launch)
  const uint32_t total = ((b1 - bw + 3) >> 2) * n;     // this wave's steps
  RowChunk16<1> ring[D];
  uint32_t ib = bw, ic = 0;  // next row step to request
  auto issue_row = [&](RowChunk16<1>& dst) {
    const bool in = ib < b1;
    const uint32_t blk = cload(a.list_blocks + b_begin + (in ? ib : bw));
    load_a16<1>(a.pool_data, blk, a.d4, (in ? ic : n - 1) * 4, lane, dst);
    if (++ic == n) {
      ic = 0;
      ib += 4;
    }
  };
  h8v qf0[2][M], qf1[2][M];
  auto read_q = [&](uint32_t c16, h8v (&d0)[M], h8v (&d1)[M]) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      d0[m] = *(const h8v*)(qt + qoff[m] + 32u * c16);
      d1[m] = *(const h8v*)(qt + qoff[m] + 32u * c16 + 16u);
    }
  };
#pragma unroll
  for (int u = 0; u < D - 1; ++u) issue_row(ring[u]);
  read_q(0, qf0[0], qf1[0]);
  uint32_t b = bw, cc = 0;  // step being consumed
  for (uint32_t s = 0; s < total; s += D) {
    if (cc + D == n) load_block_norms(a, cload(a.list_blocks + b_begin + b), rowbase, bn);
#pragma unroll
    for (int u = 0; u < D; ++u) {
      issue_row(ring[(u + D - 1) & (D - 1)]);
      const uint32_t cn = cc + u + 1;  // next step's fragments (wraps to the next block's first chunk)
      read_q(cn >= n ? cn - n : cn, qf0[(u + 1) & 1], qf1[(u + 1) & 1]);
      h8v a0, a1;
      operands_a16<1>(ring[u], a0, a1);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf0[u & 1][m], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf1[u & 1][m], acc[m], 0, 0, 0);
      }
    }
    cc += D;
    if (cc == n) {
      cc = 0;
      block_done(b);
      b += 4;
    }
  }
}

template <int QM>  // 16 QM queries per group (QM = 2 or 4)
__global__ __launch_bounds__(256, FVDB_MFMA_WG_WAVES) void scan_mfma_wg_kernel(const MfmaScanArgs a) {
  constexpr uint32_t Q = 16u * QM;
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t qstride = mfma_wg_qstride(a.dpad);
  uint32_t* s_item = (uint32_t*)(wg_lds + (Q + 1u) * qstride);
  for (uint32_t i = threadIdx.x; i < qstride / 4; i += 256) ((uint32_t*)(wg_lds + Q * qstride))[i] = 0u;
  const uint32_t n_items = cload(a.n_items);
  const uint32_t ppr = a.dpad >> 3;  // 16-byte pieces per query row
  for (;;) {
    __syncthreads();  // the previous item's tile is no longer read (first pass: the zero row is written)
    if (threadIdx.x == 0) *s_item = atomicAdd(a.head, 1u);
    __syncthreads();
    const uint32_t item = rfl(*s_item);
    if (item >= n_items) return;  // every workgroup reaches this: the queue only grows
#ifdef FVDB_MFMA_STAMPS_BUILD
    const unsigned long long t_drawn = wall_clock64();
#endif
    uint32_t lo = 0, hi = a.nlist;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cload(a.item_off + mid) <= item) lo = mid; else hi = mid;
    }
    const uint32_t L = lo;
    const uint32_t e_begin = cload(a.entry_off + L);
    const uint32_t cnt = cload(a.entry_off + L + 1) - e_begin;
    const uint32_t ngroups = (cnt + Q - 1u) / Q;
    const uint32_t local = item - cload(a.item_off + L);
    const uint32_t seg = local / ngroups, g = local - seg * ngroups;
    const uint32_t b_begin = cload(a.list_off + L);
    const uint32_t nblk = cload(a.list_off + L + 1) - b_begin;
    const uint32_t sb = L >= a.lsplit ? a.segb_tail : a.segb;
    const uint32_t b0 = seg * sb;
    const uint32_t b1 = min(b0 + sb, nblk);
    const uint32_t e0 = e_begin + g * Q;
    const uint32_t ne = min(Q, cnt - g * Q);
    // the group's queries -> LDS tile (row qi at qi * qstride), 16 bytes per thread per pass, 4 passes in flight
#ifdef FVDB_MFMA_STAMPS_BUILD
    const unsigned long long t_located = wall_clock64();
#endif
    const uint32_t pieces = ne * ppr;
    for (uint32_t p0 = threadIdx.x; p0 < pieces; p0 += 1024u) {
      h8v v[4];
      uint32_t dst[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t p = p0 + 256u * t;
        const bool in = p < pieces;
        const uint32_t qi = in ? p / ppr : 0u, c = in ? p - qi * ppr : 0u;
        const uint32_t qrow = a.entries[e0 + qi].x;
        v[t] = *(const h8v*)(a.qh + (size_t)qrow * a.dpad + 8u * c);
        dst[t] = in ? qi * qstride + 16u * c : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (dst[t] != 0xFFFFFFFFu) *(h8v*)(wg_lds + dst[t]) = v[t];
    }
    __syncthreads();
#ifdef FVDB_MFMA_STAMPS_BUILD
    const unsigned long long t_tile = wall_clock64();
#endif
    if (ne <= 16)
      mfma_item_wg<1>(a, wg_lds, qstride, b_begin, b0 + (uint32_t)w, b1, e0, ne, Q, lane);
    else if (QM == 2 || ne <= 32)
      mfma_item_wg<2>(a, wg_lds, qstride, b_begin, b0 + (uint32_t)w, b1, e0, ne, Q, lane);
    else if (ne <= 48)
      mfma_item_wg<(QM >= 4 ? 3 : 2)>(a, wg_lds, qstride, b_begin, b0 + (uint32_t)w, b1, e0, ne, Q, lane);
    else
      mfma_item_wg<(QM >= 4 ? 4 : 2)>(a, wg_lds, qstride, b_begin, b0 + (uint32_t)w, b1, e0, ne, Q, lane);
#ifdef FVDB_MFMA_STAMPS_BUILD  // dev aid: -DFVDB_MFMA_STAMPS_BUILD and FVDB_MFMA_STAMPS=<first launch to print>
    if (a.stamps && threadIdx.x == 0 && item < a.stamps_cap) {
      unsigned long long* r = a.stamps + (size_t)item * 8;
      r[0] = t_drawn;
      r[1] = t_located;
      r[2] = t_tile;
      r[3] = wall_clock64();
      r[4] = item;
      r[5] = b1 - b0;
      r[6] = ne;
      r[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID, bits 3:0
    }
#endif
  }
}

}  // namespace fvdb
