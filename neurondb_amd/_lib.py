"""ctypes binding of include/ndbhip.h.  Fails loudly when the HIP library has
not been built — there is no CPU fallback in this package."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = os.environ.get("NDBHIP_LIB") or os.path.join(_HERE, "lib", "libndbhip.so")     # override: A/B of two builds
_HDR = os.path.join(_ROOT, "include", "ndbhip.h")

OK = 0
ERR_INVALID, ERR_NODEVICE, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
STRATEGY_L2, STRATEGY_COSINE, STRATEGY_IP = 1, 2, 3


class NdbHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ndbhip error {code}: {msg}")
        self.code = code


class Cand(C.Structure):
    _fields_ = [("key", C.c_uint32), ("pos", C.c_uint32), ("tid", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("rows_scored", C.c_uint64), ("bytes_scored", C.c_uint64),
                ("scan_launches", C.c_uint64), ("scan_kernel_ms", C.c_double), ("rows_rescored", C.c_uint64),
                ("rows_emitted", C.c_uint64), ("screen16_batches", C.c_uint64), ("screen16_fallbacks", C.c_uint64),
                ("pairs_pruned", C.c_uint64), ("rows_swept", C.c_uint64), ("plane_bytes", C.c_uint64),
                ("cent_screen_batches", C.c_uint64), ("prepares", C.c_uint64), ("prepare_updates", C.c_uint64),
                ("dense_sweeps", C.c_uint64), ("wave_sweeps", C.c_uint64), ("sub_restricted", C.c_uint64)]


class ServiceStats(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("queries", C.c_uint64), ("max_batch", C.c_uint64), ("busy_s", C.c_double)]


def lib_path() -> str:
    return _LIB


def declared_symbols() -> list:
    """Every function include/*.h declares (ndbhip.h: the C ABI; ndb_am.h: the AM callbacks over it; ndb_sql.h:
    the SQL-level batch functions over it; ndb_backend.h: the GPU plugin vtable over it; ndb_service.h: the
    device-owner process and its shared-memory ring)."""
    out = set()
    for name in ("ndbhip.h", "ndb_am.h", "ndb_sql.h", "ndb_backend.h", "ndb_service.h"):
        with open(os.path.join(_ROOT, "include", name)) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out.update(re.findall(r"\b(ndbhip_[a-z0-9_]+|ndb_[a-z][a-z0-9_]+)\s*\(", text))
    return sorted(out)


class NdbItemPointer(C.Structure):
    _fields_ = [("bi_hi", C.c_uint16), ("bi_lo", C.c_uint16), ("posid", C.c_uint16)]


class NdbScanKey(C.Structure):
    _fields_ = [("sk_strategy", C.c_int), ("sk_type", C.c_int), ("sk_argument", C.c_void_p), ("sk_len", C.c_size_t)]


class NdbIndexScan(C.Structure):
    _fields_ = [("indexRelation", C.c_void_p), ("numberOfKeys", C.c_int), ("numberOfOrderBys", C.c_int),
                ("xs_heaptid", NdbItemPointer), ("xs_orderbyval", C.c_float), ("xs_orderbynull", C.c_int),
                ("xs_recheckorderby", C.c_int), ("opaque", C.c_void_p)]


class NdbHipDeviceInfo(C.Structure):
    _fields_ = [("device_id", C.c_int), ("name", C.c_char * 256), ("total_memory_bytes", C.c_size_t),
                ("free_memory_bytes", C.c_size_t), ("compute_major", C.c_int), ("compute_minor", C.c_int),
                ("is_available", C.c_bool)]


_FN = C.CFUNCTYPE


class NdbHipBackend(C.Structure):
    """include/ndb_backend.h: struct ndb_hip_backend (the reference's ndb_gpu_backend members for this path)"""
    _fields_ = [("name", C.c_char_p), ("provider", C.c_char_p), ("kind", C.c_int), ("features", C.c_uint),
                ("priority", C.c_int),
                ("init", _FN(C.c_int)), ("shutdown", _FN(None)), ("is_available", _FN(C.c_int)),
                ("device_count", _FN(C.c_int)), ("device_info", _FN(C.c_int, C.c_int, C.POINTER(NdbHipDeviceInfo))),
                ("set_device", _FN(C.c_int, C.c_int)),
                ("mem_alloc", _FN(C.c_int, C.POINTER(C.c_void_p), C.c_size_t)), ("mem_free", _FN(C.c_int, C.c_void_p)),
                ("memcpy_h2d", _FN(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("memcpy_d2h", _FN(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)),
                ("launch_l2_distance", _FN(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)),
                ("launch_cosine", _FN(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)),
                ("launch_kmeans_assign", _FN(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p)),
                ("launch_kmeans_update", _FN(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p)),
                ("launch_quant_fp16", _FN(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)),
                ("launch_quant_int8", C.c_void_p), ("launch_quant_int4", C.c_void_p),
                ("launch_quant_fp8_e4m3", C.c_void_p), ("launch_quant_fp8_e5m2", C.c_void_p),
                ("launch_quant_binary", C.c_void_p), ("launch_pq_encode", C.c_void_p)]


STREAM_CREATE = _FN(C.c_int, C.POINTER(C.c_void_p))
STREAM_OP = _FN(C.c_int, C.c_void_p)


class NdbKnnRow(C.Structure):
    _fields_ = [("query_no", C.c_int32), ("heaptid", NdbItemPointer), ("id", C.c_int64), ("distance", C.c_float)]


BULKDELETE_CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(NdbItemPointer), C.c_void_p)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise NdbHipError(ERR_NODEVICE,
                          f"{_LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; load it first so that this library binds to the SAME
    # HIP runtime (one runtime per process: streams and events can then be shared with torch).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB)
    vp, i, i64, f = C.c_void_p, C.c_int, C.c_int64, C.c_float
    u64 = C.c_uint64
    sig = {
        "ndbhip_abi_version": (i, []),
        "ndbhip_device_count": (i, []),
        "ndbhip_init": (i, [i]),
        "ndbhip_shutdown": (i, []),
        "ndbhip_last_error": (C.c_char_p, []),
        "ndbhip_set_stream": (i, [vp]),
        "ndbhip_synchronize": (i, []),
        "ndbhip_get_stream": (i, [C.POINTER(vp)]),
        "ndbhip_comm_unique_id": (i, [vp]),
        "ndbhip_comm_init": (i, [vp, i, i]),
        "ndbhip_comm_init_shm": (i, [C.c_char_p, i, i, C.c_size_t]),
        "ndbhip_comm_rank": (i, []),
        "ndbhip_comm_world": (i, []),
        "ndbhip_comm_destroy": (i, []),
        "ndbhip_comm_allgather": (i, [vp, vp, C.c_size_t]),
        "ndbhip_ivf_search_sharded": (i, [vp, vp, i, i, i, i, i64, vp, vp, vp]),
        "ndbhip_ivf_dim": (i, [vp]),
        "ndbhip_stats_get": (i, [C.POINTER(Stats)]),
        "ndbhip_stats_reset": (i, []),
        "ndbhip_profile": (i, [i]),
        "ndbhip_set_scan_mode": (i, [i]),
        "ndbhip_set_option": (i, [C.c_char_p, i]),
        "ndbhip_set_thread_stream": (i, [vp]),
        "ndbhip_mfma_probe": (i, [vp, vp, vp, vp, i, i]),
        "ndbhip_mfma_probe_f32": (i, [vp, vp, vp, vp, i]),
        "ndbhip_debug_phases": (i, [vp]),
        "ndbhip_debug_trace": (i, [vp, i]),
        "ndbhip_debug_h2_phases": (i, [vp]),
        "ndbhip_ivf_search_mapped": (i, [vp, vp, vp, i, i, i, i, i64, vp, vp, vp]),
        "ndbhip_gen_rows_device": (i, [i, C.c_uint64, C.c_uint64, i64, i64, i, i, C.c_float, vp]),
        "ndbhip_gen_rows_host": (i, [i, C.c_uint64, C.c_uint64, i64, i64, i, i, C.c_float, vp]),
        "ndbhip_ivf_create": (i, [i, i, C.POINTER(vp)]),
        "ndbhip_ivf_destroy": (i, [vp]),
        "ndbhip_ivf_set_centroids": (i, [vp, vp, i]),
        "ndbhip_ivf_load": (i, [vp, vp, vp, vp, vp, i64]),
        "ndbhip_ivf_load_device": (i, [vp, vp, vp, vp, vp, i64]),
        "ndbhip_ivf_load_f16": (i, [vp, vp, vp, vp, vp, i64]),
        "ndbhip_ivf_append": (i, [vp, i, vp, vp]),
        "ndbhip_ivf_export": (i, [vp, vp, vp, vp, vp]),
        "ndbhip_ivf_ncentroids": (i, [vp]),
        "ndbhip_ivf_delete": (i, [vp, vp, i64, C.POINTER(i64)]),
        "ndbhip_ivf_to_f16": (i, [vp, i, C.POINTER(vp)]),
        "ndbhip_ivf_share": (i, [vp, C.POINTER(vp)]),
        "ndbhip_ivf_get_nprobe": (i, [vp, C.POINTER(i)]),
        "ndbhip_ivf_set_nprobe": (i, [vp, i]),
        "ndbhip_ivf_shard_slices": (i, [vp, vp, vp, vp, C.POINTER(vp)]),
        "ndbhip_ivf_shard": (i, [vp, vp, C.POINTER(vp)]),
        "ndbhip_ivf_shape": (i, [vp, C.POINTER(i), C.POINTER(i)]),
        "ndbhip_ivf_pages_info": (i, [vp, C.c_uint32, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i64),
                                      C.POINTER(i)]),
        "ndbhip_ivf_pages_unpack": (i, [vp, C.c_uint32, vp, vp, vp, vp]),
        "ndbhip_ivf_load_pages": (i, [C.POINTER(vp), vp, C.c_uint32]),
        "ndbhip_ivf_pages_needed": (i64, [i, i, vp]),
        "ndbhip_ivf_pages_pack": (i, [i, i, i, i, vp, vp, vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]),
        "ndbhip_ivf_write_pages": (i, [vp, i, vp, C.c_uint32, C.POINTER(C.c_uint32)]),
        "ndbhip_ivf_nrows": (i64, [vp]),
        "ndbhip_ivf_max_candidates": (i64, [vp, i]),
        "ndbhip_ivf_search": (i, [vp, vp, i, i, i, i, i64, vp, vp, vp]),
        "ndbhip_ivf_search_device": (i, [vp, vp, i, i, i, i, i64, vp, vp, vp]),
        "ndbhip_ivf_select_clusters": (i, [vp, vp, i, i, vp]),
        "ndbhip_ivf_search_partial_device": (i, [vp, vp, i, i, i, i, i64, vp, vp, vp]),
        "ndbhip_ivf_search_partial_probes_device": (i, [vp, vp, i, i, i, i, i64, vp, vp, vp, vp]),
        "ndbhip_ivf_select_clusters_device": (i, [vp, vp, i, i, vp]),
        "ndbhip_merge_topk_device": (i, [vp, vp, vp, i, i, i, i, vp, vp, vp]),
        "ndbhip_merge_topk_host": (i, [vp, vp, vp, i, i, i, i, vp, vp, vp]),
        "ndbhip_kmeans_device": (i, [vp, i, i, i, i, f, vp, vp, vp, C.POINTER(i), C.POINTER(f)]),
        "ndbhip_ivf_build_sharded": (i, [vp, vp, vp, i64, i, vp, vp]),
        "ndbhip_comm_alltoallv": (i, [vp, vp, vp, vp]),
        "ndbhip_comm_allreduce_min_f32": (i, [vp, C.c_size_t]),
        "ndbhip_ivf_assign_device": (i, [vp, i, i, vp, i64, vp]),
        "ndbhip_ivf_build_device": (i, [vp, vp, vp, i64, i, C.POINTER(i)]),
        "ndbhip_ivf_prepare": (i, [vp, i]),
        "ndbhip_hnsw_create": (i, [i, i, C.POINTER(vp)]),
        "ndbhip_hnsw_destroy": (i, [vp]),
        "ndbhip_hnsw_load": (i, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_uint32, i]),
        "ndbhip_hnsw_build_device": (i, [vp, vp, vp, C.c_uint32, vp, i]),
        "ndbhip_hnsw_build_intended_device": (i, [vp, vp, vp, C.c_uint32, vp, i, i, i]),
        "ndbhip_hnsw_insert_intended_device": (i, [vp, vp, vp, C.c_uint32, vp, i, i, i]),
        "ndbhip_hnsw_insert_intended": (i, [vp, vp, vp, C.c_uint32, vp, i, i, i]),
        "ndbhip_hnsw_search_intended_device": (i, [vp, vp, i, i, i, i, vp, vp, vp, vp, vp]),
        "ndbhip_hnsw_search_intended_w16_device": (i, [vp, vp, i, i, i, i, vp, vp, vp, vp, vp]),
        "ndbhip_hnsw_search_intended": (i, [vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp]),
        "ndbhip_hnsw_share": (i, [vp, C.POINTER(vp)]),
        "ndbhip_hnsw_set_intended_select": (i, [i]),
        "ndbhip_hnsw_pages_info": (i, [vp, C.c_uint32, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i),
                                    C.POINTER(C.c_uint32), C.POINTER(i)]),
        "ndbhip_hnsw_pages_unpack": (i, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp]),
        "ndbhip_hnsw_pages_pack": (i, [i, i, i, i, C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_uint32, i, vp, C.c_uint32]),
        "ndbhip_hnsw_load_pages": (i, [C.POINTER(vp), vp, C.c_uint32]),
        "ndbhip_hnsw_write_pages": (i, [vp, i, i, vp, C.c_uint32, C.POINTER(C.c_uint32)]),
        "ndbhip_hnsw_get_meta": (i, [vp, C.POINTER(i), C.POINTER(i)]),
        "ndbhip_hnsw_set_meta": (i, [vp, i, i]),
        "ndbhip_hnsw_shape": (i, [vp, C.POINTER(i), C.POINTER(i)]),
        "ndbhip_hnsw_export_rows": (i, [vp, vp, vp, vp]),
        "ndbhip_hnsw_set_dead_flags": (i, [vp, vp]),
        "ndbhip_hnsw_insert_device": (i, [vp, vp, vp, C.c_uint32, vp, i]),
        "ndbhip_hnsw_delete": (i, [vp, vp, i64, C.POINTER(i64)]),
        "ndbhip_hnsw_build_stats": (i, [vp, vp]),
        "ndbhip_hnsw_set_search_mode": (i, [i]),
        "ndbhip_hnsw_set_build_mode": (i, [i, i, i]),
        "ndbhip_hnsw_export": (i, [vp, C.POINTER(C.c_uint32), vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(i)]),
        "ndbhip_hnsw_search": (i, [vp, vp, i, i, i, i, vp, vp, vp, vp, vp]),
        "ndbhip_hnsw_search_device": (i, [vp, vp, i, i, i, i, vp, vp, vp, vp, vp]),
        "ndbhip_hnsw_search_layer": (i, [vp, vp, i, i, i, i, vp, vp, vp, vp, vp]),
        "ndbhip_hnsw_search_layer_device": (i, [vp, vp, i, i, i, i, vp, vp, vp, vp, vp]),
        "ndbhip_batch_distance": (i, [vp, vp, vp, i, i, i, i, i]),
        "ndbhip_extract_vector": (i, [i, vp, C.c_size_t, vp, i, C.POINTER(i)]),
        "ndbhip_pair_distance": (i, [vp, vp, vp, i, i, i]),
        "ndbhip_kmeans_assign": (i, [vp, vp, vp, i, i, i]),
        "ndbhip_kmeans_update": (i, [vp, vp, vp, i, i, i]),
        "ndbhip_quant_fp16": (i, [vp, vp, i64]),
        # include/ndb_am.h
        "ndb_service_create": (i, [C.c_char_p, i, i, i, C.POINTER(vp)]),
        "ndb_service_destroy": (i, [vp]),
        "ndb_service_poll": (i, [vp, i, i, i, vp, vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i64)]),
        "ndb_service_query_offset": (i64, [vp, i]),
        "ndb_service_segment": (i, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ndb_service_complete": (i, [vp, i, vp, vp, vp, vp, i, i]),
        "ndb_service_serve_ivf": (i, [vp, vp, i, i, i64, C.POINTER(ServiceStats)]),
        "ndb_service_stop": (i, [vp]),
        "ndb_service_stopped": (i, [vp]),
        "ndb_client_connect": (i, [C.c_char_p, C.POINTER(vp)]),
        "ndb_client_disconnect": (i, [vp]),
        "ndb_client_dim": (i, [vp]),
        "ndb_client_stop_service": (i, [vp]),
        "ndb_client_submit": (i, [vp, vp, i, i, i, i64, C.POINTER(i)]),
        "ndb_client_wait": (i, [vp, i, vp, vp, C.POINTER(i), i]),
        "ndb_client_search": (i, [vp, vp, i, i, i, i64, vp, vp, C.POINTER(i), i]),
        "ndb_client_submit_index": (i, [vp, u64, u64, vp, i, i, i, i64, C.POINTER(i)]),
        "ndb_client_search_index": (i, [vp, u64, u64, vp, i, i, i, i64, vp, vp, C.POINTER(i), i]),
        "ndb_client_index": (i, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "ndb_client_meta_nprobe": (i, [vp]),
        "ndb_service_publish": (i, [vp, u64, u64, i]),
        "ndb_service_reload_wanted": (i, [vp, C.POINTER(u64)]),
        "ndb_service_reclaim": (i, [vp]),
        "ndb_gen_attach": (i, [C.c_char_p, i, C.POINTER(vp)]),
        "ndb_gen_detach": (i, [vp, C.c_char_p]),
        "ndb_gen_get": (u64, [vp, u64]),
        "ndb_gen_bump": (u64, [vp, u64]),
        "ndb_ivfbeginscan_service": (C.POINTER(NdbIndexScan), [u64, u64, i, i]),
        "ndb_am_use_service": (i, [C.c_char_p]),
        "ndb_am_set_guc": (i, [C.c_char_p, i]),
        "ndb_am_get_guc": (i, [C.c_char_p, C.POINTER(i)]),
        "ndb_ivfbeginscan": (C.POINTER(NdbIndexScan), [vp, i, i]),
        "ndb_ivfrescan": (i, [C.POINTER(NdbIndexScan), C.POINTER(NdbScanKey), i, C.POINTER(NdbScanKey), i]),
        "ndb_ivfgettuple": (i, [C.POINTER(NdbIndexScan), i]),
        "ndb_ivfendscan": (None, [C.POINTER(NdbIndexScan)]),
        "ndb_hnswbeginscan": (C.POINTER(NdbIndexScan), [vp, i, i]),
        "ndb_hnswrescan": (i, [C.POINTER(NdbIndexScan), C.POINTER(NdbScanKey), i, C.POINTER(NdbScanKey), i]),
        "ndb_hnswgettuple": (i, [C.POINTER(NdbIndexScan), i]),
        "ndb_hnswendscan": (None, [C.POINTER(NdbIndexScan)]),
        "ndb_ivfinsert": (i, [vp, vp, C.c_size_t, i, C.POINTER(NdbItemPointer)]),
        "ndb_hnswinsert": (i, [vp, vp, C.c_size_t, i, C.POINTER(NdbItemPointer), i]),
        "ndb_hnsw_level_from_uniform": (i, [C.c_double, C.c_float]),
        "ndb_ivfbulkdelete": (i, [vp, BULKDELETE_CALLBACK, vp, C.POINTER(i64)]),
        "ndb_hnswbulkdelete": (i, [vp, BULKDELETE_CALLBACK, vp, C.POINTER(i64)]),
        # include/ndb_sql.h
        "ndb_vector_l2_distance_batch": (i, [vp, vp, i, vp, C.c_size_t, vp]),
        "ndb_vector_cosine_distance_batch": (i, [vp, vp, i, vp, C.c_size_t, vp]),
        "ndb_vector_inner_product_batch": (i, [vp, vp, i, vp, C.c_size_t, vp]),
        "ndb_vector_l2_distance_gpu": (i, [vp, C.c_size_t, vp, C.c_size_t, vp]),
        "ndb_vector_cosine_distance_gpu": (i, [vp, C.c_size_t, vp, C.c_size_t, vp]),
        "ndb_vector_inner_product_gpu": (i, [vp, C.c_size_t, vp, C.c_size_t, vp]),
        "ndb_ivf_knn_search_gpu": (i, [vp, i, vp, vp, i, i, i, vp, C.POINTER(i64)]),
        # include/ndb_backend.h
        "ndb_hip_backend_get": (C.POINTER(NdbHipBackend), []),
        "ndb_hip_backend_streams": (None, [C.POINTER(STREAM_CREATE), C.POINTER(STREAM_OP), C.POINTER(STREAM_OP)]),
        "ndb_hnsw_knn_search_gpu": (i, [vp, i, vp, vp, i, i, i, vp, C.POINTER(i64)]),
        "ndbhip_ivf_build": (i, [vp, vp, vp, i64, i, C.POINTER(i)]),
        "ndbhip_ivf_insert": (i, [vp, vp, vp, C.POINTER(i)]),
        "ndbhip_hnsw_insert": (i, [vp, vp, vp, C.c_uint32, vp, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    return lib().ndbhip_last_error().decode("utf-8", "replace")


def check(rc: int):
    if rc != OK:
        raise NdbHipError(rc, last_error())


_inited = False
_device = -1


def ensure_init(device: int | None = None):
    """Lazy per-process device init (the reference's ndb_gpu_init_if_needed pattern)."""
    global _inited, _device
    if _inited:
        if device is not None and int(device) != _device:
            raise RuntimeError(f"the library is initialised on device {_device}, not {device} (one device per process)")
        return
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
        n = lib().ndbhip_device_count()
        if n > 0:
            device %= n
    check(lib().ndbhip_init(int(device)))
    _inited = True
    _device = int(device)


def stats() -> dict:
    s = Stats()
    check(lib().ndbhip_stats_get(C.byref(s)))
    return {k: getattr(s, k) for k, _ in Stats._fields_}


def use_torch_stream():
    """Order the library's kernels with torch's: both run on torch's current stream afterwards.  The
    device-pointer entry points are asynchronous on the library's stream, so a caller that fills, gathers or
    reads their buffers with torch (bench.py, dist.py, the GPU tests) has to share a stream with them.
    torch's default stream has the handle 0, which ndbhip_set_stream reads as "the library's own stream" — a
    non-blocking stream that does NOT synchronise with the default stream — so in that case torch is first
    switched to a stream of its own (ordered after the work already queued on the default stream)."""
    import torch
    ensure_init(torch.cuda.current_device())
    s = torch.cuda.current_stream()
    if s.cuda_stream == 0:
        new = torch.cuda.Stream()
        new.wait_stream(s)
        torch.cuda.set_stream(new)
        s = new
    check(lib().ndbhip_set_stream(s.cuda_stream))
    return s
