/* fvdb.h — C ABI of the MI355X-native distance/top-k engine for fabstir-vectordb's
 * hybrid HNSW/IVF hot path.
 *
 * The reference (Rust) has NO FFI seam for distance computation: the L2 loops are called
 * directly (src/ivf/core.rs:351,378,424,650,671; src/hnsw/core.rs:279,434,487,515,573,605,611).
 * This header DEFINES the seam exactly at those call sites; each entry point cites the
 * reference code it replaces.  A Rust host binds it with a plain `extern "C"` block
 * (see INTEGRATION.md); this repo's own host code (C++, fabstir-vectordb_amd/host) and the
 * test harness (Python ctypes) use the same symbols.
 *
 * Conventions
 *  - plain pointers + sizes, row-major, no strides; caller owns host buffers for the
 *    duration of a call; the library owns device memory behind opaque handles.
 *  - `*_dev` variants take DEVICE pointers (HBM-resident inputs/outputs) and only enqueue
 *    work on the context's stream (call fvdb_ctx_synchronize or use the stream).
 *  - every call returns an fvdb_status; nothing throws or aborts across the boundary.
 *    Fewer than k hits is NOT an error (reference returns a short list:
 *    tests/ivf/core.rs:385-398, tests/hnsw/core.rs:300-316): see out_counts.
 *  - distances are the reference's arithmetic bit for bit: sqrt(sum_i (a_i-b_i)^2), f32,
 *    summed left to right, no FMA (src/core/vector_ops.rs:51-57).
 *  - ordering: ascending distance; exact ties keep scan order (probe rank, then position
 *    in the list) = what the reference's stable sort gives (src/ivf/core.rs:655,677).
 *  - threading (reference: searches hold a tokio RwLock READ guard, bindings/node/src/session.rs:253,
 *    src/hybrid/core.rs:457,466 — many at once; inserts/deletes hold the write guard):
 *      * the blocking host-pointer searches (fvdb_ivf_search, fvdb_ivf_search_all, fvdb_ivf_coarse) may be
 *        called on ONE index from any number of host threads at once: each call leases a private scratch
 *        set and stream inside the library and returns it when done;
 *      * the `*_slot` entry points may be called concurrently for DIFFERENT slots (each slot = one scratch
 *        set; the caller supplies the stream through `on`); two calls naming the same slot must use the same
 *        stream and are then ordered by it;
 *      * the stream-ordered `*_dev` entry points without a slot use set 0 on the index's own stream;
 *      * a search never modifies the index object.  Mutations (set_centroids, train, add*, set_deleted,
 *        clear, reserve, graph_upload, store_append) need external exclusion against searches and each other,
 *        as the reference's write guard provides.
 */
#ifndef FVDB_H
#define FVDB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fvdb_status {
  FVDB_OK = 0,
  FVDB_E_NOT_TRAINED = 1,   /* IVFError::NotTrained            src/ivf/core.rs:16 */
  FVDB_E_DUPLICATE = 2,     /* IVFError/HNSWError::DuplicateVector (raised by host code) */
  FVDB_E_DIM = 3,           /* DimensionMismatch                src/ivf/core.rs:22 */
  FVDB_E_INSUFFICIENT = 4,  /* InsufficientTrainingData         src/ivf/core.rs:25 */
  FVDB_E_INCONSISTENT = 5,  /* InconsistentDimensions           src/ivf/core.rs:28 */
  FVDB_E_INVALID = 6,       /* InvalidConfig / bad argument     src/ivf/core.rs:31 */
  FVDB_E_NOT_FOUND = 7,     /* VectorNotFound                   src/ivf/core.rs:37 */
  FVDB_E_NOT_INITIALIZED = 8, /* HybridError::NotInitialized    src/hybrid/core.rs */
  FVDB_E_NONFINITE = 9,     /* NaN/Inf input (reference panics: partial_cmp().unwrap()) */
  FVDB_E_HIP = 10,          /* HIP runtime error (message via fvdb_last_error) */
  FVDB_E_OOM = 11,          /* device or host allocation failed */
  FVDB_E_UNSUPPORTED = 12,  /* k or nprobe above the compiled limit (FVDB_MAX_K) */
  FVDB_E_RCCL = 13          /* RCCL missing or a collective failed (message via fvdb_last_error) */
} fvdb_status;

/* Largest k (and nprobe) served by the in-kernel wavefront top-k (64 lanes x 4 registers). */
#define FVDB_MAX_K 256u
/* Row id meaning "no row" in padded outputs / candidate lists. */
#define FVDB_NO_ROW UINT32_MAX
#define FVDB_NO_ID UINT64_MAX

typedef struct fvdb_ctx fvdb_ctx;     /* one GPU + one HIP stream + scratch arenas */
typedef struct fvdb_ivf fvdb_ivf;     /* IVF-flat index resident in HBM (centroids + paged lists) */
typedef struct fvdb_store fvdb_store; /* row-major vector store for gathered candidate scoring */

/* ---- context ------------------------------------------------------------------------ */
int fvdb_ctx_create(int device, fvdb_ctx** out);
void fvdb_ctx_destroy(fvdb_ctx* ctx);
int fvdb_ctx_synchronize(fvdb_ctx* ctx);
int fvdb_device_synchronize(fvdb_ctx* ctx); /* every stream of ctx's device (hipDeviceSynchronize) */
int fvdb_ctx_device(fvdb_ctx* ctx); /* the device ordinal the context was created on */
/* Batches in flight run on streams of their own; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues
 * (default 4), and kernels of streams sharing a queue run one after the other.  fvdb_ctx_create asks for 16 unless the host
 * application set the variable itself — which only works if HIP has not been initialised in the process yet.
 * hw_queues_source: 1 = the host application's setting, 2 = set by this library, 3 = could NOT be applied (HIP was already
 * up; hw_queues then reports the runtime's default and a warning went to stderr: expect ~35 % less overlap). */
typedef struct fvdb_ctx_info_t {
  int device, compute_units, hw_queues, hw_queues_source;
} fvdb_ctx_info_t;
int fvdb_ctx_info(fvdb_ctx* ctx, fvdb_ctx_info_t* out);
void* fvdb_ctx_stream(fvdb_ctx* ctx);          /* hipStream_t, for callers that interleave work */
const char* fvdb_last_error(fvdb_ctx* ctx);    /* message of the last failing call on ctx */
const char* fvdb_version(void);
/* Device memory helpers so a host without a HIP binding can keep inputs resident in HBM. */
int fvdb_dev_alloc(fvdb_ctx* ctx, size_t bytes, void** out);
int fvdb_dev_free(fvdb_ctx* ctx, void* p);
int fvdb_dev_upload(fvdb_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int fvdb_dev_download(fvdb_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Pieces for keeping more than one batch in flight: pinned host memory, a copy that does not wait, and events
 * recorded on a context's stream (fvdb_event_wait blocks the host until the work recorded before it is done). */
int fvdb_host_alloc(fvdb_ctx* ctx, size_t bytes, void** out);
void fvdb_host_free(fvdb_ctx* ctx, void* p);
int fvdb_dev_download_async(fvdb_ctx* ctx, void* dst_pinned, const void* src_dev, size_t bytes);
typedef struct fvdb_event fvdb_event;
int fvdb_event_create(fvdb_ctx* ctx, fvdb_event** out);
void fvdb_event_destroy(fvdb_event* e);
int fvdb_event_record(fvdb_ctx* ctx, fvdb_event* e);
int fvdb_event_wait(fvdb_ctx* ctx, fvdb_event* e);
/* Stream-ordered timing of the region between the two calls (HIP events on ctx's stream). */
int fvdb_timer_start(fvdb_ctx* ctx);
int fvdb_timer_stop_ms(fvdb_ctx* ctx, float* out_ms);
/* Per-stage HIP-event timing of searches on this context.  on = 1: each search synchronises the
 * stream and accumulates its stage times; on = 2: events are only recorded, the caller folds them
 * in with fvdb_ivf_profile_collect() after its own synchronisation point (no extra sync). */
int fvdb_ctx_set_profiling(fvdb_ctx* ctx, int on);

/* ---- similarity utilities ------------------------------------------------------------
 * dot_product_scalar / cosine_similarity_scalar / batch_cosine_similarity
 * (src/core/vector_ops.rs:8-10,35-49; Embedding::cosine_similarity src/core/types.rs:79-103):
 * B queries x n rows -> out[B x n], sequential f32 folds (bit-identical to the reference).
 * No index calls these in the reference; they are part of its public vector_ops surface. */
int fvdb_dot_products(fvdb_ctx* ctx, const float* q, uint32_t B, const float* x, uint64_t n, uint32_t d, float* out);
int fvdb_cosine_similarities(fvdb_ctx* ctx, const float* q, uint32_t B, const float* x, uint64_t n, uint32_t d,
                             float* out);

/* ---- IVF-flat ------------------------------------------------------------------------
 * Replaces IVFIndex's arithmetic: find_nearest_centroid (src/ivf/core.rs:373-386), the
 * coarse ranking + list scan + selection of search_with_config (:626-681) and
 * batch_search (src/ivf/operations.rs:132-145).  Row ids are caller-chosen u64.
 */
int fvdb_ivf_create(fvdb_ctx* ctx, uint32_t d, uint32_t nlist, fvdb_ivf** out);
/* Row storage of the inverted lists.  FVDB_F16 (config C5) keeps rows as IEEE fp16 (rounded to nearest
 * even at insert) and widens them to f32 in registers; centroids, queries, the f32 fold and the ordering
 * are unchanged, so results equal the reference algorithm run on the fp16-rounded rows. */
typedef enum fvdb_dtype { FVDB_F32 = 0, FVDB_F16 = 1 } fvdb_dtype;
int fvdb_ivf_create_ex(fvdb_ctx* ctx, uint32_t d, uint32_t nlist, int row_dtype, fvdb_ivf** out);
void fvdb_ivf_destroy(fvdb_ivf* ivf);
/* set_trained (src/ivf/core.rs:509-520): install nlist x d centroids (host), empty the lists. */
int fvdb_ivf_set_centroids(fvdb_ivf* ivf, const float* centroids);
int fvdb_ivf_get_centroids(fvdb_ivf* ivf, float* out /* nlist x d */);
/* k-means++ + Lloyd on the GPU (src/ivf/core.rs:240-429).  Assignment = the coarse kernel;
 * centroid update and error are summed in data order like the reference.  The seeding
 * draws come from SplitMix64(seed) (the reference's StdRng stream is unpinned). */
typedef struct fvdb_train_result {
  uint32_t iterations;
  uint32_t converged;
  float initial_error;
  float final_error;
} fvdb_train_result;
int fvdb_ivf_train(fvdb_ivf* ivf, const float* x, uint64_t n, uint32_t max_iterations, uint64_t seed,
                   fvdb_train_result* out);
/* Batched find_nearest_centroid: strict '<', lowest cluster id wins ties. */
int fvdb_ivf_assign(fvdb_ivf* ivf, const float* x, uint64_t n, uint32_t* out_cluster);
/* insert (src/ivf/core.rs:431-455) for n rows: assign on the GPU, append to the lists in row
 * order.  out_cluster/out_pos (optional) receive each row's list and position in it. */
int fvdb_ivf_add(fvdb_ivf* ivf, const float* x, const uint64_t* ids, uint64_t n, uint32_t* out_cluster,
                 uint32_t* out_pos);
/* Append rows whose cluster is already known (load path, shard placement). */
int fvdb_ivf_add_assigned(fvdb_ivf* ivf, const float* x, const uint64_t* ids, uint64_t n,
                          const uint32_t* cluster, uint32_t* out_pos);
/* Soft delete (src/ivf/operations.rs:569-591 + the skip at src/ivf/core.rs:666-669). */
int fvdb_ivf_set_deleted(fvdb_ivf* ivf, const uint32_t* cluster, const uint32_t* pos, uint64_t n, int deleted);
int fvdb_ivf_list_sizes(fvdb_ivf* ivf, uint64_t* out /* nlist */);
uint64_t fvdb_ivf_total_rows(fvdb_ivf* ivf);
/* Copy list `list` back to the host in list-position order: rows [len x d] as f32 (fp16 rows widened exactly), ids
 * [len], live [len] (0 = soft-deleted).  Any output may be NULL.  len = fvdb_ivf_list_sizes()[list].  Replaces the
 * walk over `InvertedList.vectors` of the reference's save path (src/hybrid/persistence.rs:289-311). */
int fvdb_ivf_list_export(fvdb_ivf* ivf, uint32_t list, float* rows, uint64_t* ids, uint8_t* live);
int fvdb_ivf_reserve(fvdb_ivf* ivf, uint64_t n_rows);
int fvdb_ivf_clear(fvdb_ivf* ivf);   /* empties the lists, keeps centroids (hybrid initialize :278-287) */
/* Multi-GPU: sizes of ALL lists of the logical index (this rank may own a subset), so the
 * tie-break position `seq` is identical on every rank.  Default = local sizes. */
int fvdb_ivf_set_global_list_sizes(fvdb_ivf* ivf, const uint64_t* sizes /* nlist */);

/* search_with_config for B queries.  out_ids/out_dist are B x k (unused tail = FVDB_NO_ID /
 * +inf), out_counts[B] the number of hits.  nprobe > nlist probes every list. */
int fvdb_ivf_search(fvdb_ivf* ivf, const float* q, uint32_t B, uint32_t k, uint32_t nprobe, uint64_t* out_ids,
                    float* out_dist, uint32_t* out_counts);
/* Same with device pointers.  q is B x d row-major f32 in HBM.  out_keys (optional, B x k u64)
 * receives (distance bits << 32 | seq): the total order used for selection; unique per row, so
 * per-GPU partial results can be merged exactly (fvdb_merge_keys_dev). */
int fvdb_ivf_search_dev(fvdb_ivf* ivf, const float* q_dev, uint32_t B, uint32_t k, uint32_t nprobe,
                        uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                        uint64_t* out_keys_dev);
/* Exhaustive scan of every list (exact k-NN; ground truth for recall, and the a7/a8 kernel
 * with no coarse step). */
int fvdb_ivf_search_all(fvdb_ivf* ivf, const float* q, uint32_t B, uint32_t k, uint64_t* out_ids,
                        float* out_dist, uint32_t* out_counts);
/* Same search on the stream of context `on` (NULL = the index's own) with the slot-th (0..15) set of per-search
 * scratch: searches in different slots and on different contexts may be in flight together.  The index must not be
 * modified while any is.  Stage timing (profiling) is only kept for searches on the index's own context. */
int fvdb_ivf_search_dev_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t k,
                             uint32_t nprobe, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                             uint64_t* out_keys_dev);
/* The two stages separately (multi-GPU: each rank ranks the centroids for its own queries only, the probe lists
 * travel with the queries, and every rank scans its lists for all of them).  probes: B x min(nprobe, n_clusters)
 * cluster ids in probe order, device memory. */
int fvdb_ivf_coarse_dev_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t nprobe,
                             uint32_t* out_probes_dev);
int fvdb_ivf_search_probes_dev_slot(fvdb_ivf* ivf, fvdb_ctx* on, uint32_t slot, const float* q_dev,
                                    const uint32_t* probes_dev, uint32_t B, uint32_t k, uint32_t nprobe,
                                    uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                                    uint64_t* out_keys_dev);
int fvdb_ivf_search_all_dev(fvdb_ivf* ivf, const float* q_dev, uint32_t B, uint32_t k, uint64_t* out_ids_dev,
                            float* out_dist_dev, uint32_t* out_counts_dev);
/* Coarse step alone (src/ivf/core.rs:645-656): the nprobe nearest clusters per query in
 * probe order, and their distances. */
int fvdb_ivf_coarse(fvdb_ivf* ivf, const float* q, uint32_t B, uint32_t nprobe, uint32_t* out_clusters,
                    float* out_dist);

/* Coarse-stage implementation.  FVDB_COARSE_AUTO (default): the query x centroid contraction runs on the
 * matrix cores (MFMA) to propose 64 candidate clusters per query, whose distances are then recomputed with
 * the reference's sequential f32 arithmetic; a rounding-error bound proves the proposal contains the true
 * top nprobe, otherwise that query is ranked exactly over the whole table.  Results are identical to
 * FVDB_COARSE_EXACT (every centroid scored with the reference's arithmetic) by construction.  AUTO applies
 * when padded d % 16 == 0, nprobe <= 48 and n_clusters >= 64; other shapes use the exact scan. */
#define FVDB_COARSE_AUTO 0
#define FVDB_COARSE_EXACT 1
int fvdb_ivf_set_coarse_mode(fvdb_ivf* ivf, int mode);
/* Queries (since the centroids were installed) whose proposal could not be proven and were ranked exactly. */
int fvdb_ivf_coarse_fallbacks(fvdb_ivf* ivf, uint64_t* out);

/* Inverted-list scan implementation.  FVDB_SCAN_AUTO (default): fp16 MFMA evaluates |x|^2 - 2 x.q for every
 * (row, query) of the probed lists and discards the rows that provably cannot reach the top k (threshold from an
 * exact scan of the query's nearest lists plus a rounding-error bound); the few survivors are scored with the
 * reference's sequential f32 fold and selected by (distance, scan position).  A query whose k-th result is not
 * strictly below its threshold is rescanned exactly.  Results are identical to FVDB_SCAN_EXACT (every probed
 * row scored with the reference's arithmetic) by construction.  AUTO also watches its own hit rate: when more than
 * one query in eight of the recent batches needed the exact rescan (data the filter cannot separate), the following
 * 64 batches (doubling up to 4096 while that stays so) use the exact scan.  AUTO applies when padded d % 16 == 0,
 * k <= 26, nprobe <= 256 and the batch has 32..16384 queries; other shapes use the exact scan. */
#define FVDB_SCAN_AUTO 0
#define FVDB_SCAN_EXACT 1
#define FVDB_SCAN_FILTER 2 /* the matrix-core filter for every batch it can serve (AUTO without the hit-rate back-off) */
int fvdb_ivf_set_scan_mode(fvdb_ivf* ivf, int mode);
/* Queries (since the centroids were installed) that were rescanned exactly. */
int fvdb_ivf_scan_fallbacks(fvdb_ivf* ivf, uint64_t* out);
/* of those, by cause: [0] survivor buffer overflow, [1] more candidates than the select stage scores, [2] the k-th kept
 * distance not strictly below the bound of the unscored rows, [3] no usable threshold; and [4] (not a rescan) queries
 * whose survivors outgrew the buffer and were filtered a second time with a threshold taken from those survivors */
int fvdb_ivf_scan_fallback_reasons(fvdb_ivf* ivf, uint64_t* out5);

/* Diagnostic: rows per query that survived the matrix-core filter in the last (sub-)batch of B queries. */
int fvdb_ivf_scan_survivors(fvdb_ivf* ivf, uint32_t* out, uint32_t B);

/* Diagnostic: the survivors of one query of the last matrix-core (sub-)batch — probe rank, position in that list and
 * the matrix-core value v = |x|^2 - 2 x~.q~ — so that a test can hold v + |q|^2 against the reference's sum and the
 * error bound.  At most max_n entries are written. */
int fvdb_ivf_scan_survivor_dump(fvdb_ivf* ivf, uint32_t query, uint32_t max_n, uint32_t* rank, uint32_t* pos, float* v,
                                uint32_t* n_out);

/* Counters of the last search on this index (for roofline accounting). */
typedef struct fvdb_search_stats {
  uint64_t rows_scanned;      /* sum over queries of rows in probed lists (algorithmic) */
  uint64_t work_items;        /* (list segment, query group) items executed */
  uint64_t list_rows_touched; /* rows of the union of probed lists (physical lower bound) */
} fvdb_search_stats;
int fvdb_ivf_last_stats(fvdb_ivf* ivf, fvdb_search_stats* out);
/* With profiling on: ms[8] = coarse scan, coarse merge, plan, fine scan, fine merge, [5] the matrix-core filter
 * kernel alone (part of fine scan; 0 on the exact path), [6..7] reserved (0) — summed over the searches since
 * the last call; returns how many searches (sub-batches) were accumulated. */
uint64_t fvdb_ivf_stage_times(fvdb_ivf* ivf, float* ms_out);
int fvdb_ivf_profile_collect(fvdb_ivf* ivf);  /* profiling mode 2: fold the last search's events in */

/* ---- merge ---------------------------------------------------------------------------
 * G-way merge of per-shard partial top-k by key (a4/a12 semantics: ascending, keep k).
 * keys/ids: G x B x k (device).  Exact because keys are unique. */
int fvdb_merge_keys_dev(fvdb_ctx* ctx, const uint64_t* keys_dev, const uint64_t* ids_dev, uint32_t G, uint32_t B,
                        uint32_t k, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev);

/* ---- top-k / merge utilities -----------------------------------------------------------
 * The reference's public vector_ops helpers, batched: B independent rows per call, k <= FVDB_MAX_K.  Outputs are
 * B x k (unused tail = FVDB_NO_ID / 0), out_counts[B] the number of entries.  A NaN score is FVDB_E_NONFINITE in
 * the host-pointer forms (the reference panics in partial_cmp().unwrap()); the *_dev forms take device pointers,
 * run on ctx's stream and do not inspect their input.
 *  - fvdb_top_k_indices: top_k_indices (src/core/vector_ops.rs:12-22) — indices of the k largest scores, ties in
 *    index order (stable sort).
 *  - fvdb_top_k_indices_heap: top_k_indices_heap (:180-201) — BinaryHeap of size k with strict `>` replacement, then
 *    a stable descending sort of the heap's vector: the tie behaviour of that heap is reproduced exactly.
 *  - fvdb_streaming_top_k: StreamingTopK::add for every (id, score) of a row in order, then get_results()
 *    (:204-263).  Heap order is the (reversed score, id) tuple order; u64 ids stand for VectorIds (a host that
 *    needs the reference's tie order passes the first 8 bytes of the VectorId, big-endian).
 *  - fvdb_merge_search_results: merge_search_results (:24-32) + SearchResult::deduplicate (src/core/types.rs:206-223)
 *    — per row the concatenated result sets (id FVDB_NO_ID = padding); per id the smallest distance is kept (the
 *    earliest entry on a tie), survivors ascending by distance, first k.  Where the reference's order among equal
 *    distances comes out of HashMap iteration, first appearance of the id decides. */
int fvdb_top_k_indices(fvdb_ctx* ctx, const float* scores /* B x n */, uint32_t B, uint64_t n, uint32_t k,
                       uint64_t* out_idx, uint32_t* out_counts);
int fvdb_top_k_indices_heap(fvdb_ctx* ctx, const float* scores, uint32_t B, uint64_t n, uint32_t k, uint64_t* out_idx,
                            uint32_t* out_counts);
int fvdb_top_k_indices_dev(fvdb_ctx* ctx, const float* scores_dev, uint32_t B, uint64_t n, uint32_t k, int heap,
                           uint64_t* out_idx_dev, uint32_t* out_counts_dev);
int fvdb_streaming_top_k(fvdb_ctx* ctx, const uint64_t* ids, const float* scores, uint32_t B, uint64_t n, uint32_t k,
                         uint64_t* out_ids, float* out_scores, uint32_t* out_counts);
int fvdb_streaming_top_k_dev(fvdb_ctx* ctx, const uint64_t* ids_dev, const float* scores_dev, uint32_t B, uint64_t n,
                             uint32_t k, uint64_t* out_ids_dev, float* out_scores_dev, uint32_t* out_counts_dev);
int fvdb_merge_search_results(fvdb_ctx* ctx, const uint64_t* ids, const float* dist, uint32_t B, uint64_t n, uint32_t k,
                              uint64_t* out_ids, float* out_dist, uint32_t* out_counts);
int fvdb_merge_search_results_dev(fvdb_ctx* ctx, const uint64_t* ids_dev, const float* dist_dev, uint32_t B, uint64_t n,
                                  uint32_t k, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_counts_dev);

/* ---- candidate scoring (HNSW) --------------------------------------------------------
 * Replaces euclidean_distance at src/hnsw/core.rs:279,434,487,515 (search_layer) and
 * :573,605,611 (prune): the host walks the graph, each hop's candidate batch is scored here.
 */
int fvdb_store_create(fvdb_ctx* ctx, uint32_t d, uint64_t capacity_rows, fvdb_store** out);
void fvdb_store_destroy(fvdb_store* s);
int fvdb_store_append(fvdb_store* s, const float* rows, uint64_t n, uint64_t* first_row);
uint64_t fvdb_store_rows(fvdb_store* s);
int fvdb_store_get(fvdb_store* s, uint64_t row, float* out /* d */);
/* One-shot: B queries (host), C candidate row indices per query (FVDB_NO_ROW = pad) -> B x C
 * distances (+inf for pads). */
int fvdb_score_candidates(fvdb_store* s, const float* q, uint32_t B, const uint32_t* cand, uint32_t C, float* out);
/* Hop loop: queries stay in HBM, candidates/distances travel through pinned host memory
 * mapped into the GPU (no memcpy calls): one launch + one stream sync per hop. */
typedef struct fvdb_scorer fvdb_scorer;
int fvdb_scorer_create(fvdb_store* s, uint32_t max_B, uint32_t max_C, fvdb_scorer** out);
void fvdb_scorer_destroy(fvdb_scorer* sc);
int fvdb_scorer_set_queries(fvdb_scorer* sc, const float* q, uint32_t B);      /* host rows */
int fvdb_scorer_set_query_rows(fvdb_scorer* sc, const uint32_t* rows, uint32_t B); /* queries = stored rows */
int fvdb_scorer_set_queries_dev(fvdb_scorer* sc, const float* q_dev, uint32_t B); /* B x d rows already in HBM */
uint32_t* fvdb_scorer_cand_buffer(fvdb_scorer* sc);   /* B x max_C, write candidates here */
const float* fvdb_scorer_dist_buffer(fvdb_scorer* sc); /* B x max_C, read distances here */
int fvdb_scorer_run(fvdb_scorer* sc, uint32_t B, uint32_t C); /* scores cand_buffer rows 0..B, columns 0..C; waits */
/* Each scorer owns a HIP stream: several scorers (one per host thread / query group) keep several
 * hops in flight.  launch enqueues the scoring of rows 0..B x columns 0..C, wait blocks until it is done. */
int fvdb_scorer_launch(fvdb_scorer* sc, uint32_t B, uint32_t C);
int fvdb_scorer_wait(fvdb_scorer* sc);

/* ---- device-resident graph traversal --------------------------------------------------------
 * Optional fast path for HNSWIndex::search (src/hnsw/core.rs:398-554): the adjacency lists are
 * mirrored in HBM and one wavefront per query runs the whole layered search in a single launch
 * (same heaps, same visiting order, same distance arithmetic => same results as the host walk that
 * uses fvdb_scorer_*).  Rows are the store's; node index = store row.
 */
typedef struct fvdb_graph fvdb_graph;
int fvdb_graph_create(fvdb_store* s, fvdb_graph** out);
void fvdb_graph_destroy(fvdb_graph* g);
/* Install a whole graph (restore, bulk build, vacuum): levels[n]; deleted[n] (0/1); slot_start[n_slots+1] / adj[]: CSR
 * over (node, layer) slots in node order, layer 0 first (n_slots = sum(level+1)); every list must hold <= 64 neighbours. */
int fvdb_graph_upload(fvdb_graph* g, uint32_t n, const uint32_t* levels, const uint8_t* deleted,
                      const uint32_t* slot_start, const uint32_t* adj, uint32_t entry_node);
int fvdb_graph_set_deleted(fvdb_graph* g, uint32_t node, int deleted);
/* B queries (device, B x d).  out_nodes/out_dist: B x k device buffers, out_counts/out_status: B.
 * status 1 = the query overflowed the on-chip candidate heap and must be searched through the host walk
 * instead (its count is 0). */
int fvdb_graph_search_dev(fvdb_graph* g, const float* q_dev, uint32_t B, uint32_t k, uint32_t ef,
                          uint32_t* out_nodes_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                          uint32_t* out_status_dev);
/* Same on the stream of context `on` (NULL = the store's context), with the slot-th (0..15) set of per-batch
 * traversal state: searches in different slots and on different contexts may be in flight together. */
int fvdb_graph_search_dev_slot(fvdb_graph* g, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t k,
                               uint32_t ef, uint32_t* out_nodes_dev, float* out_dist_dev, uint32_t* out_counts_dev,
                               uint32_t* out_status_dev);
/* ---- device-resident graph construction --------------------------------------------------------
 * HNSWIndex::insert (src/hnsw/core.rs:226-378) with the graph resident in HBM: greedy descent, search_layer
 * (ef_construction), select_neighbors (:556-558), back-links and prune_neighbors_with_new_node (:588-624) run in one
 * workgroup per insert against the adjacency rows in HBM, strictly in insert order — the graph is the one the
 * reference's sequential loop builds.  Adjacency rows have a fixed stride per (node, layer), so every mutation —
 * this one, or a host-side insert patched in with fvdb_graph_set_lists — rewrites only the rows it touches.
 *
 * configure: the degree caps (max_connections, max_connections_layer_0; src/hnsw/core.rs:37-46) the row strides
 *   are sized for; before the first node.
 * append_nodes: store rows [first, first + n) become nodes with empty lists (levels[n]); first must equal the
 *   current node count (node index = store row).
 * insert_linked: links the appended, not yet linked nodes [first, first + n) in order.  mode 0 = choose,
 *   1 = one insert at a time, 2 = speculate a batch of searches against the frozen graph and commit in order
 *   (a speculated search is adopted only if it is provably the search the serial algorithm would run at its turn:
 *   the adjacency rows it expanded are unchanged, or what earlier inserts of the batch added to / dropped from them
 *   cannot alter its pops, its expansions or its result — DESIGN.md section 6a).  The result never depends on the
 *   mode.  Environment knobs for A/B runs only: FVDB_BUILD_MODE, FVDB_BUILD_K / FVDB_BUILD_KMAX (batch size),
 *   FVDB_BUILD_STRICT (1 = adopt only when no expanded row changed), FVDB_BUILD_SEQ_BELOW, FVDB_BUILD_EXACT_FIRST,
 *   FVDB_BUILD_DEBUG (per-call diagnostics on stderr).  *n_done < n with stats->needs_host = 1: node first + *n_done needs the host path (level >= 16 or an
 *   on-chip heap overflow); link it through fvdb_graph_set_lists / fvdb_graph_set_entry and call again.
 * set_lists: overwrite the lists of n_lists (node, layer) rows: offsets[n_lists + 1] into nbrs[].
 * set_entry: entry point + the number of nodes whose links are complete.
 * download: the adjacency in CSR form over (node, layer) slots in node order (slot_start[slots + 1], adj[adj_cap];
 *   either may be NULL), *n_edges = total.  upload_bytes: host -> device bytes of graph STRUCTURE moved so far. */
typedef struct fvdb_graph_insert_stats {
  uint32_t n_done, needs_host;
  uint32_t speculated_ok, searched_in_commit, commit_stops;  /* speculation: adopted / searched by the commit workgroup / early stops */
  uint32_t rounds, expanded, rows_scored, tie_restarts;      /* of the searches run by the commit workgroup */
  uint32_t launches;
} fvdb_graph_insert_stats;
int fvdb_graph_configure(fvdb_graph* g, uint32_t max_connections, uint32_t max_connections_layer_0);
int fvdb_graph_append_nodes(fvdb_graph* g, uint32_t first, uint32_t n, const uint32_t* levels);
int fvdb_graph_insert_linked(fvdb_graph* g, uint32_t first, uint32_t n, uint32_t ef_construction, int mode, uint32_t* n_done,
                             fvdb_graph_insert_stats* stats);
int fvdb_graph_set_lists(fvdb_graph* g, uint32_t n_lists, const uint32_t* nodes, const uint32_t* layers, const uint32_t* offsets,
                         const uint32_t* nbrs);
int fvdb_graph_set_entry(fvdb_graph* g, uint32_t entry_node, uint32_t n_linked);
int fvdb_graph_entry(fvdb_graph* g, uint32_t* entry_node, uint32_t* n_nodes);
int fvdb_graph_download(fvdb_graph* g, uint32_t* slot_start, uint32_t* adj, uint64_t adj_cap, uint64_t* n_edges);
uint64_t fvdb_graph_upload_bytes(fvdb_graph* g);

/* With profiling on (fvdb_ctx_set_profiling): summed duration (HIP events on the launch stream) of the last
 * <= 64 launches of the traversal kernel since the previous call, and how many were summed; and (always) the
 * rows scored and hops taken by all queries since the previous call (either may be NULL).  Synchronises. */
int fvdb_graph_kernel_times(fvdb_graph* g, float* ms_sum, uint32_t* launches, uint64_t* rows_scored, uint64_t* hops);
/* Since the graph was created: queries the traversal kernel served, and how many of them it searched a second time with
 * the reference's BinaryHeaps restated because equal distances met where the heap layout decides (src/hnsw/core.rs:469-554;
 * such a query is the slowest of its launch).  Synchronises. */
int fvdb_graph_tie_restarts(fvdb_graph* g, uint64_t* queries, uint64_t* searched_again);

/* ---- multi-GPU: inverted lists sharded across the GPUs of a node, RCCL over xGMI -------------------------
 * (BASELINE config C4; SURVEY §8e.  The reference has no distributed execution: what is preserved is the result —
 * the merged answer equals the single-index answer bit for bit, because the selection keys are global.)
 * One rank per GPU: a process (torch.distributed-style launch) or a host thread of one process, each with its own
 * fvdb_ctx.  Lists are placed with fvdb_ivf_add_assigned (only the lists the rank owns) + fvdb_ivf_set_global_list_sizes;
 * centroids are replicated.
 *
 * Communicator.  fvdb_comm_unique_id on rank 0 yields 128 bytes (ncclUniqueId) that the host ships to the other
 * ranks out of band (its own rendezvous: TCP store, file, MPI ...); every rank then calls fvdb_comm_create
 * (ncclCommInitRank).  librccl is loaded at that moment (dlopen), never by a single-GPU process.
 * fvdb_comm_create_hosted is the same interface over a caller-supplied exchange of HOST buffers (tests, rehearsals on
 * one GPU, fabrics RCCL does not drive): op 0 = all-gather (send: `bytes`, recv: world blocks of `bytes` in rank
 * order), op 1 = all-to-all (send/recv: world blocks of `bytes`; block p of send goes to rank p); return 0 on success.
 * Collectives of one communicator must be issued in the same order on every rank. */
typedef struct fvdb_comm fvdb_comm;
typedef int (*fvdb_exchange_fn)(void* user, int op, const void* send_host, void* recv_host, size_t bytes);
int fvdb_comm_unique_id(void* out128);
int fvdb_comm_create(fvdb_ctx* ctx, const void* id128, int world, int rank, fvdb_comm** out);
int fvdb_comm_create_hosted(fvdb_ctx* ctx, int world, int rank, fvdb_exchange_fn fn, void* user, fvdb_comm** out);
void fvdb_comm_destroy(fvdb_comm* comm);
int fvdb_comm_rank(fvdb_comm* comm);
int fvdb_comm_world(fvdb_comm* comm);
/* Stream-ordered collectives on `on`'s stream (NULL = the communicator's context): recv_dev holds world blocks. */
int fvdb_comm_all_gather_dev(fvdb_comm* comm, fvdb_ctx* on, const void* send_dev, void* recv_dev, size_t bytes);
int fvdb_comm_all_to_all_dev(fvdb_comm* comm, fvdb_ctx* on, const void* send_dev, void* recv_dev, size_t bytes);

/* Sharded search_with_config.  FVDB_SHARD_WEAK: q_dev = this rank's OWN B queries (global batch world*B); the rank
 * gets the results of its own B queries.  FVDB_SHARD_STRONG: q_dev = the SAME B queries on every rank (global batch
 * B); rank r gets the results of queries [r*per, min(B, (r+1)*per)), per = ceil(B/world) = fvdb_sharded_out_rows().
 * begin enqueues the whole step — centroid ranking, exchange 1 (weak: all-gather of queries + probe lists), scan of
 * the lists this rank owns, exchange 2 (all-to-all of the partial (key, id) lists), world-way merge by key — on
 * `on`'s stream (NULL = the index's own) with the slot-th scratch set; no host synchronisation in between.  Outputs
 * (device, fvdb_sharded_out_rows() x k) are valid after fvdb_ivf_search_sharded_end (or any wait on that stream).
 * Every rank calls begin for the same slots in the same order. */
#define FVDB_SHARD_WEAK 0
#define FVDB_SHARD_STRONG 1
typedef struct fvdb_sharded fvdb_sharded;
int fvdb_sharded_create(fvdb_ivf* ivf, fvdb_comm* comm, fvdb_sharded** out);
void fvdb_sharded_destroy(fvdb_sharded* s);
uint32_t fvdb_sharded_out_rows(fvdb_sharded* s, uint32_t B, int mode);
int fvdb_ivf_search_sharded_begin(fvdb_sharded* s, fvdb_ctx* on, uint32_t slot, const float* q_dev, uint32_t B, uint32_t k,
                                  uint32_t nprobe, int mode, uint64_t* out_ids_dev, float* out_dist_dev,
                                  uint32_t* out_counts_dev);
int fvdb_ivf_search_sharded_end(fvdb_sharded* s, fvdb_ctx* on, uint32_t slot);

#ifdef __cplusplus
}
#endif
#endif /* FVDB_H */
