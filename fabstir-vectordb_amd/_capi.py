"""ctypes declarations for lib/libfvdb_hip.so (C ABI: include/fvdb.h).

The library is the product: if it is missing or fails to load this module raises — there is
no CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never imported here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FVDB_LIB_DIR: another build of BOTH libraries (lib_dev: the dev build with include/fvdb_dev.h's entry points; A/B builds)
LIB_DIR = os.environ.get("FVDB_LIB_DIR") or os.path.join(_HERE, "lib")
if not os.path.isabs(LIB_DIR):
    LIB_DIR = os.path.join(_HERE, LIB_DIR)
LIB_PATH = os.environ.get("FVDB_HIP_LIB") or os.path.join(LIB_DIR, "libfvdb_hip.so")

vp, u64, u32, i32, f32, sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_size_t
f32p, u64p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)


class TrainResult(C.Structure):
    _fields_ = [("iterations", u32), ("converged", u32), ("initial_error", f32), ("final_error", f32)]


class SearchStats(C.Structure):
    _fields_ = [("rows_scanned", u64), ("work_items", u64), ("list_rows_touched", u64)]


# name -> (restype, argtypes).  Every symbol include/fvdb.h declares is listed here and
# tests/test_cabi_symbols.py checks the built library exports each of them.
SIGNATURES = {
    "fvdb_version": (C.c_char_p, []),
    "fvdb_ctx_create": (i32, [i32, C.POINTER(vp)]),
    "fvdb_ctx_destroy": (None, [vp]),
    "fvdb_ctx_synchronize": (i32, [vp]),
    "fvdb_device_synchronize": (i32, [vp]),
    "fvdb_ctx_stream": (vp, [vp]),
    "fvdb_last_error": (C.c_char_p, [vp]),
    "fvdb_dev_alloc": (i32, [vp, sz, C.POINTER(vp)]),
    "fvdb_dev_free": (i32, [vp, vp]),
    "fvdb_dev_upload": (i32, [vp, vp, vp, sz]),
    "fvdb_dev_download": (i32, [vp, vp, vp, sz]),
    "fvdb_timer_start": (i32, [vp]),
    "fvdb_timer_stop_ms": (i32, [vp, f32p]),
    "fvdb_ctx_set_profiling": (i32, [vp, i32]),
    "fvdb_dot_products": (i32, [vp, f32p, u32, f32p, u64, u32, f32p]),
    "fvdb_cosine_similarities": (i32, [vp, f32p, u32, f32p, u64, u32, f32p]),
    "fvdb_ivf_create": (i32, [vp, u32, u32, C.POINTER(vp)]),
    "fvdb_ivf_create_ex": (i32, [vp, u32, u32, i32, C.POINTER(vp)]),
    "fvdb_ivf_destroy": (None, [vp]),
    "fvdb_ivf_set_centroids": (i32, [vp, f32p]),
    "fvdb_ivf_get_centroids": (i32, [vp, f32p]),
    "fvdb_ivf_train": (i32, [vp, f32p, u64, u32, u64, C.POINTER(TrainResult)]),
    "fvdb_ivf_assign": (i32, [vp, f32p, u64, u32p]),
    "fvdb_ivf_add": (i32, [vp, f32p, u64p, u64, u32p, u32p]),
    "fvdb_ivf_add_assigned": (i32, [vp, f32p, u64p, u64, u32p, u32p]),
    "fvdb_ivf_set_deleted": (i32, [vp, u32p, u32p, u64, i32]),
    "fvdb_ivf_list_sizes": (i32, [vp, u64p]),
    "fvdb_ivf_list_export": (i32, [vp, u32, f32p, u64p, C.POINTER(C.c_uint8)]),
    "fvdb_ivf_total_rows": (u64, [vp]),
    "fvdb_ivf_reserve": (i32, [vp, u64]),
    "fvdb_ivf_clear": (i32, [vp]),
    "fvdb_ivf_set_global_list_sizes": (i32, [vp, u64p]),
    "fvdb_ivf_search": (i32, [vp, f32p, u32, u32, u32, u64p, f32p, u32p]),
    "fvdb_ivf_search_dev": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp]),
    "fvdb_ivf_search_all": (i32, [vp, f32p, u32, u32, u64p, f32p, u32p]),
    "fvdb_ivf_search_all_dev": (i32, [vp, vp, u32, u32, vp, vp, vp]),
    "fvdb_ivf_coarse": (i32, [vp, f32p, u32, u32, u32p, f32p]),
    "fvdb_ivf_set_coarse_mode": (i32, [vp, i32]),
    "fvdb_ivf_coarse_fallbacks": (i32, [vp, u64p]),
    "fvdb_ivf_set_scan_mode": (i32, [vp, i32]),
    "fvdb_ivf_scan_fallbacks": (i32, [vp, u64p]),
    "fvdb_ivf_scan_fallback_reasons": (i32, [vp, u64p]),
    "fvdb_ivf_scan_survivors": (i32, [vp, u32p, u32]),
    "fvdb_ivf_scan_survivor_dump": (i32, [vp, u32, u32, u32p, u32p, f32p, u32p]),
    "fvdb_ivf_last_stats": (i32, [vp, C.POINTER(SearchStats)]),
    "fvdb_ivf_stage_times": (u64, [vp, f32p]),
    "fvdb_ivf_profile_collect": (i32, [vp]),
    "fvdb_merge_keys_dev": (i32, [vp, vp, vp, u32, u32, u32, vp, vp, vp]),
    "fvdb_store_create": (i32, [vp, u32, u64, C.POINTER(vp)]),
    "fvdb_store_destroy": (None, [vp]),
    "fvdb_store_append": (i32, [vp, f32p, u64, u64p]),
    "fvdb_store_rows": (u64, [vp]),
    "fvdb_store_get": (i32, [vp, u64, f32p]),
    "fvdb_score_candidates": (i32, [vp, f32p, u32, u32p, u32, f32p]),
    "fvdb_scorer_create": (i32, [vp, u32, u32, C.POINTER(vp)]),
    "fvdb_scorer_destroy": (None, [vp]),
    "fvdb_scorer_set_queries": (i32, [vp, f32p, u32]),
    "fvdb_scorer_set_query_rows": (i32, [vp, u32p, u32]),
    "fvdb_scorer_set_queries_dev": (i32, [vp, vp, u32]),
    "fvdb_scorer_cand_buffer": (u32p, [vp]),
    "fvdb_scorer_dist_buffer": (f32p, [vp]),
    "fvdb_scorer_run": (i32, [vp, u32, u32]),
    "fvdb_graph_create": (i32, [vp, C.POINTER(vp)]),
    "fvdb_graph_destroy": (None, [vp]),
    "fvdb_graph_upload": (i32, [vp, u32, u32p, C.POINTER(C.c_uint8), u32p, u32p, u32]),
    "fvdb_graph_set_deleted": (i32, [vp, u32, i32]),
    "fvdb_graph_configure": (i32, [vp, u32, u32]),
    "fvdb_graph_append_nodes": (i32, [vp, u32, u32, u32p]),
    "fvdb_graph_insert_linked": (i32, [vp, u32, u32, u32, i32, u32p, vp]),
    "fvdb_graph_set_lists": (i32, [vp, u32, u32p, u32p, u32p, u32p]),
    "fvdb_graph_set_entry": (i32, [vp, u32, u32]),
    "fvdb_graph_entry": (i32, [vp, u32p, u32p]),
    "fvdb_graph_download": (i32, [vp, u32p, u32p, u64, u64p]),
    "fvdb_graph_upload_bytes": (u64, [vp]),
    "fvdb_graph_search_dev": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp]),
    "fvdb_graph_search_dev_slot": (i32, [vp, vp, u32, vp, u32, u32, u32, vp, vp, vp, vp]),
    "fvdb_ctx_device": (i32, [vp]),
    "fvdb_ctx_info": (i32, [vp, vp]),
    "fvdb_ivf_search_dev_slot": (i32, [vp, vp, u32, vp, u32, u32, u32, vp, vp, vp, vp]),
    "fvdb_ivf_coarse_dev_slot": (i32, [vp, vp, u32, vp, u32, u32, vp]),
    "fvdb_ivf_search_probes_dev_slot": (i32, [vp, vp, u32, vp, vp, u32, u32, u32, vp, vp, vp, vp]),
    "fvdb_host_alloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
    "fvdb_host_free": (None, [vp, vp]),
    "fvdb_dev_download_async": (i32, [vp, vp, vp, C.c_size_t]),
    "fvdb_event_create": (i32, [vp, C.POINTER(vp)]),
    "fvdb_event_destroy": (None, [vp]),
    "fvdb_event_record": (i32, [vp, vp]),
    "fvdb_event_wait": (i32, [vp, vp]),
    "fvdb_graph_kernel_times": (i32, [vp, f32p, u32p, u64p, u64p]),
    "fvdb_graph_tie_restarts": (i32, [vp, u64p, u64p]),
    "fvdb_scorer_launch": (i32, [vp, u32, u32]),
    "fvdb_scorer_wait": (i32, [vp]),
    "fvdb_top_k_indices": (i32, [vp, f32p, u32, u64, u32, u64p, u32p]),
    "fvdb_top_k_indices_heap": (i32, [vp, f32p, u32, u64, u32, u64p, u32p]),
    "fvdb_top_k_indices_dev": (i32, [vp, vp, u32, u64, u32, i32, vp, vp]),
    "fvdb_streaming_top_k": (i32, [vp, u64p, f32p, u32, u64, u32, u64p, f32p, u32p]),
    "fvdb_streaming_top_k_dev": (i32, [vp, vp, vp, u32, u64, u32, vp, vp, vp]),
    "fvdb_merge_search_results": (i32, [vp, u64p, f32p, u32, u64, u32, u64p, f32p, u32p]),
    "fvdb_merge_search_results_dev": (i32, [vp, vp, vp, u32, u64, u32, vp, vp, vp]),
    "fvdb_comm_unique_id": (i32, [vp]),
    "fvdb_comm_create": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
    "fvdb_comm_create_hosted": (i32, [vp, i32, i32, vp, vp, C.POINTER(vp)]),
    "fvdb_comm_destroy": (None, [vp]),
    "fvdb_comm_rank": (i32, [vp]),
    "fvdb_comm_world": (i32, [vp]),
    "fvdb_comm_all_gather_dev": (i32, [vp, vp, vp, vp, sz]),
    "fvdb_comm_all_to_all_dev": (i32, [vp, vp, vp, vp, sz]),
    "fvdb_sharded_create": (i32, [vp, vp, C.POINTER(vp)]),
    "fvdb_sharded_destroy": (None, [vp]),
    "fvdb_sharded_out_rows": (u32, [vp, u32, i32]),
    "fvdb_ivf_search_sharded_begin": (i32, [vp, vp, u32, vp, u32, u32, u32, i32, vp, vp, vp]),
    "fvdb_ivf_search_sharded_end": (i32, [vp, vp, u32]),
}

_lib = None


def load():
    """dlopen the engine.  Raises (never falls back) when the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C fabstir-vectordb_amd` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # see fvdb_ctx_create: one hardware queue per stream in flight
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the build disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
