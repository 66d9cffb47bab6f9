// common.h — shared device/host structures of the fvdb HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fvdb {

constexpr int kWave = 64;             // CDNA4 wavefront
constexpr uint32_t kInf32 = 0xFFFFFFFFu;  // "no candidate" key half (never a finite distance)

// HBM layout of a vector pool: 64-row blocks, dimension-chunk major inside a block so that
// one wave-instruction (global_load_dwordx4, lane = row) reads 1 KiB contiguous:
//   data[(block * d4 + chunk) * 64 + lane] is a float4 = dims 4*chunk .. 4*chunk+3 of row `lane`.
// d is padded to a multiple of 4 with zeros (adding (0-0)^2 = +0 leaves an f32 sum unchanged).
struct PoolView {
  const void* data;       // float4 chunks (f32 rows) or 8-half chunks (fp16 rows)
  const uint64_t* ids;    // [blocks][64] caller's row ids
  const uint64_t* valid;  // [blocks] bit l set = lane l holds a live (inserted, not deleted) row
  uint32_t d4;            // number of float4 chunks per row
};

// A list is a sequence of pool blocks: blocks[off[L] .. off[L+1]) in scan order.
struct ListTable {
  const uint32_t* off;     // [nlist+1]
  const uint32_t* blocks;  // pool block index
  uint32_t nlist;
};

// Probe plan for one batch: which queries scan which list.
//   entries[entry_off[L] .. entry_off[L+1]) = (query, probe rank) pairs probing list L
//   work item i in [item_off[L], item_off[L+1]) = (segment, query group) of list L
struct Plan {
  const uint32_t* entry_off;  // [nlist+1]
  const uint32_t* item_off;   // [nlist+1]
  const uint2* entries;       // x = query index, y = probe rank
  const uint32_t* n_items;    // device scalar = item_off[nlist]
  uint32_t* head;             // work-queue head (zeroed by the plan kernels)
};

struct ScanArgs {
  PoolView pool;
  ListTable lists;
  Plan plan;
  const float* queries;  // [B][dpad]
  uint32_t dpad;         // 4 * d4
  uint32_t segb;         // blocks per segment
  uint32_t k;            // entries kept per (query, rank, segment)
  uint32_t nprobe;       // ranks per query in `part`
  uint32_t maxsegs;      // segments per (query, rank) slot in `part`
  uint2* part;           // [B][nprobe][maxsegs][k]  x = distance bits, y = position in list
};

struct MergeArgs {
  PoolView pool;
  ListTable lists;
  const uint32_t* probes;      // [B][nprobe] list ids in probe order (nullptr: single list 0)
  const uint32_t* glob_blocks; // [nlist] blocks per list of the LOGICAL index (seq base); may alias local
  const uint2* part;
  uint32_t B, k, nprobe, maxsegs, segb;
  uint64_t* out_ids;      // [B][k] or nullptr
  float* out_dist;        // [B][k] or nullptr
  uint32_t* out_counts;   // [B] or nullptr
  uint64_t* out_keys;     // [B][k] or nullptr
  uint32_t* out_probes;   // [B][k] u32 ids (coarse stage) or nullptr
  const uint32_t* qlist;  // when set: wave i handles query qlist[i], for i < *nq (exact rescan of a few queries)
  const uint32_t* nq;
};

}  // namespace fvdb
