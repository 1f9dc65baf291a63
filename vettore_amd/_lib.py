"""ctypes binding of libvettore_hip.so (include/vettore_flat.h).

The library is the product: if it is missing or cannot be loaded, importing
fails loudly -- there is no Python or CPU fallback for any compute call.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VETTORE_HIP_LIB: another build of the same library (e.g. lib/libvettore_hip_hooks.so, the one
# with the fault-injection hooks two tests need); default: the product library beside this file
LIB_PATH = os.environ.get("VETTORE_HIP_LIB") or os.path.join(_HERE, "lib", "libvettore_hip.so")

VT_OK = 0
ORDER_PAIR, ORDER_AVX, ORDER_SEQ, ORDER_SSE2 = 0, 1, 2, 3
NOMINATE_F32, NOMINATE_BF16 = 1, 2
SHADOW_OFF, SHADOW_AUTO = 0, 1
SHADOW_STATE = {0: "off", 1: "none", 2: "current", 3: "stale", 4: "refused"}
EXCHANGE_HOST, EXCHANGE_RCCL = 0, 1

# every symbol include/vettore_flat.h declares
SYMBOLS = [
    "vt_strerror", "vt_last_error", "vt_abi_version", "vt_device_count", "vt_device_read_peak", "vt_debug_set", "vt_debug_get",
    "vt_hits_len", "vt_hits_id", "vt_hits_raw", "vt_hits_rank_key", "vt_hits_pack", "vt_hits_pack_many", "vt_hit_blocks_merge", "vt_hits_id_bytes", "vt_hits_export",
    "vt_hits_free", "vt_hits_free_many",
    "vt_flat_new", "vt_flat_new_sharded", "vt_flat_shard_count", "vt_flat_shard_device", "vt_flat_shard_len", "vt_flat_shard_memory", "vt_flat_coalesce_stats",
    "vt_flat_route_ids", "vt_flat_set_exchange", "vt_flat_exchange", "vt_flat_exchange_note", "vt_flat_rccl_ranks", "vt_flat_free", "vt_flat_insert", "vt_flat_insert_many", "vt_flat_delete",
    "vt_flat_search", "vt_flat_search_batch", "vt_flat_len", "vt_flat_dimension", "vt_flat_metric",
    "vt_flat_set_reduce_order", "vt_set_default_reduce_order", "vt_flat_set_batch_nominate", "vt_flat_batch_nominate",
    "vt_flat_load_matrix", "vt_flat_load_device_matrix", "vt_flat_quantized_search", "vt_flat_quantized_search_batch", "vt_flat_funnel_search", "vt_flat_funnel_search_batch", "vt_flat_hybrid_search",
    "vt_rank_ids", "vt_flat_set_id_ranks", "vt_flat_stream", "vt_flat_search_begin", "vt_flat_merge_gathered",
    "vt_vector_top_k", "vt_binary_top_k", "vt_normalize_l2", "vt_compress_sign_bits",
    "vt_flat_set_profiling", "vt_flat_get_profile", "vt_flat_get_profile_sized",
    "vt_flat_set_batch_shadow", "vt_flat_batch_shadow", "vt_flat_set_single_nominate", "vt_flat_single_nominate",
]
ABI_VERSION = 4  # VT_ABI_VERSION of the include/vettore_flat.h this file was written against


class Profile(C.Structure):
    _fields_ = [
        ("scan_launches", C.c_uint64), ("scan_ms", C.c_double), ("scan_rows", C.c_uint64),
        ("scan_bytes", C.c_uint64), ("hamming_launches", C.c_uint64), ("hamming_ms", C.c_double),
        ("hamming_bytes", C.c_uint64), ("merge_launches", C.c_uint64), ("merge_ms", C.c_double),
        ("batch_launches", C.c_uint64), ("batch_ms", C.c_double), ("batch_flops", C.c_double),
        ("batch_queries", C.c_uint64), ("batch_fallbacks", C.c_uint64),
        ("prefix_launches", C.c_uint64), ("prefix_ms", C.c_double), ("prefix_bytes", C.c_uint64),
        ("nominate_launches", C.c_uint64), ("nominate_ms", C.c_double), ("nominate_bytes", C.c_uint64),
        ("nominate_flops", C.c_double), ("nominate_queries", C.c_uint64),
        ("nominate_second_passes", C.c_uint64), ("nominate_candidates", C.c_uint64),
        ("hamming_queries", C.c_uint64),
        ("hybrid_device_chains", C.c_uint64),
        ("prefix_queries", C.c_uint64),
        ("nominate_shadow_launches", C.c_uint64), ("shadow_builds", C.c_uint64), ("shadow_build_ms", C.c_double),
        ("shadow_patched_rows", C.c_uint64), ("sweep_queries", C.c_uint64),
    ]


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make` (hipcc --offload-arch=gfx950). "
            "vettore_amd has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    f32p, u64p, szp, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.c_void_p
    L.vt_device_read_peak.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.vt_strerror.restype = C.c_char_p
    L.vt_strerror.argtypes = [C.c_int]
    L.vt_last_error.restype = C.c_char_p
    L.vt_abi_version.restype = C.c_int
    L.vt_device_count.restype = C.c_int
    L.vt_hits_len.restype = C.c_size_t
    L.vt_hits_len.argtypes = [vp]
    L.vt_hits_id.restype = C.POINTER(C.c_char)
    L.vt_hits_id.argtypes = [vp, C.c_size_t, szp]
    L.vt_hits_raw.restype = C.c_float
    L.vt_hits_raw.argtypes = [vp, C.c_size_t]
    L.vt_hits_rank_key.restype = C.c_uint32
    L.vt_hits_rank_key.argtypes = [vp, C.c_size_t]
    L.vt_hits_pack.restype = C.c_size_t
    L.vt_hits_pack.argtypes = [vp, vp, C.c_size_t]
    L.vt_hits_pack_many.restype = None
    L.vt_hits_pack_many.argtypes = [C.POINTER(vp), C.c_size_t, C.c_size_t, vp]
    L.vt_hit_blocks_merge.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]
    L.vt_hits_id_bytes.restype = C.c_size_t
    L.vt_hits_id_bytes.argtypes = [vp]
    L.vt_hits_export.restype = None
    L.vt_hits_export.argtypes = [vp, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.vt_hits_free.restype = None
    L.vt_hits_free.argtypes = [vp]
    L.vt_hits_free_many.restype = None
    L.vt_hits_free_many.argtypes = [C.POINTER(vp), C.c_size_t]
    L.vt_flat_new.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.vt_flat_new_sharded.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_size_t, C.POINTER(vp)]
    L.vt_flat_shard_count.restype = C.c_size_t
    L.vt_flat_shard_count.argtypes = [vp]
    L.vt_flat_shard_device.argtypes = [vp, C.c_size_t]
    L.vt_flat_shard_len.restype = C.c_size_t
    L.vt_flat_shard_len.argtypes = [vp, C.c_size_t]
    L.vt_flat_coalesce_stats.restype = C.c_int
    L.vt_flat_coalesce_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.vt_flat_shard_memory.restype = C.c_int
    L.vt_flat_shard_memory.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.vt_flat_route_ids.argtypes = [vp, C.c_size_t, C.c_char_p, szp, C.POINTER(C.c_uint32)]
    L.vt_flat_set_exchange.argtypes = [vp, C.c_int]
    L.vt_flat_exchange.argtypes = [vp]
    L.vt_flat_exchange_note.argtypes = [vp]
    L.vt_flat_exchange_note.restype = C.c_char_p
    L.vt_flat_rccl_ranks.argtypes = [vp]
    L.vt_flat_free.restype = None
    L.vt_flat_free.argtypes = [vp]
    L.vt_flat_insert.argtypes = [vp, C.c_char_p, C.c_size_t, f32p, C.c_size_t]
    L.vt_flat_insert_many.argtypes = [vp, C.c_size_t, C.c_char_p, szp, f32p, szp]
    L.vt_flat_delete.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.vt_flat_search.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vt_flat_search_batch.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vt_flat_len.restype = C.c_size_t
    L.vt_flat_len.argtypes = [vp]
    L.vt_flat_dimension.restype = C.c_long
    L.vt_flat_dimension.argtypes = [vp]
    L.vt_flat_metric.argtypes = [vp]
    L.vt_flat_set_reduce_order.argtypes = [vp, C.c_int]
    L.vt_set_default_reduce_order.argtypes = [C.c_int]
    L.vt_flat_set_batch_nominate.argtypes = [vp, C.c_int]
    L.vt_flat_batch_nominate.argtypes = [vp]
    L.vt_flat_load_matrix.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_char_p, szp, f32p]
    L.vt_flat_load_device_matrix.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_char_p, szp, vp]
    L.vt_flat_quantized_search.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vt_flat_quantized_search_batch.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vt_flat_funnel_search.argtypes = [vp, f32p, C.c_size_t, szp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vt_flat_funnel_search_batch.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, szp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vt_flat_hybrid_search.argtypes = [vp, f32p, C.c_size_t, C.POINTER(C.c_int), szp, szp, szp, C.c_size_t, C.c_size_t,
                                        C.POINTER(vp)]
    u32p = C.POINTER(C.c_uint32)
    L.vt_rank_ids.argtypes = [C.c_char_p, szp, C.c_size_t, u32p]
    L.vt_flat_set_id_ranks.argtypes = [vp, u32p, C.c_size_t]
    L.vt_flat_stream.restype = vp
    L.vt_flat_stream.argtypes = [vp]
    L.vt_flat_search_begin.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, vp]
    L.vt_flat_merge_gathered.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, u64p, u32p, f32p, u32p, szp]
    L.vt_vector_top_k.argtypes = [C.c_int, C.c_size_t, C.c_char_p, szp, f32p, szp, f32p, C.c_size_t, C.c_int,
                                  C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.vt_binary_top_k.argtypes = [C.c_int, C.c_size_t, C.c_char_p, szp, u64p, szp, u64p, C.c_size_t, C.c_size_t,
                                  C.c_size_t, C.POINTER(vp)]
    L.vt_normalize_l2.argtypes = [C.c_int, C.c_size_t, C.c_size_t, f32p, f32p]
    L.vt_compress_sign_bits.argtypes = [C.c_int, C.c_size_t, C.c_size_t, f32p, u64p]
    L.vt_flat_set_profiling.argtypes = [vp, C.c_int]
    L.vt_flat_get_profile.argtypes = [vp, C.POINTER(Profile), C.c_int]
    L.vt_flat_get_profile_sized.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.vt_flat_set_batch_shadow.argtypes = [vp, C.c_int]
    L.vt_flat_batch_shadow.argtypes = [vp]
    L.vt_flat_set_single_nominate.argtypes = [vp, C.c_int]
    L.vt_flat_single_nominate.argtypes = [vp]
    L.vt_debug_set.argtypes = [C.c_char_p, C.c_long]
    L.vt_debug_get.argtypes = [C.c_char_p, C.POINTER(C.c_long)]
    # a library built from another header would be handed structs of the wrong size
    if L.vt_abi_version() != ABI_VERSION:
        raise ImportError("%s speaks ABI version %d, this binding %d: rebuild with `make`"
                          % (LIB_PATH, L.vt_abi_version(), ABI_VERSION))
    _lib = L
    return L


def error_text(status: int) -> str:
    """The reference's error string for statuses 1..7, else status + detail."""
    L = load()
    msg = L.vt_strerror(status).decode()
    if status >= 16:
        detail = L.vt_last_error().decode()
        if detail:
            msg = "%s: %s" % (msg, detail)
    return msg
