"""ctypes binding of libplhip.so (include/plhip.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is missing or
no MI355X is visible, calls fail loudly (PlHipError / OSError).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

# plhip_dtype_t
I8, I16, I32, I64, U8, U16, U32, U64, F32, F64 = range(10)
BOOL = 10  # Arrow boolean array: the values buffer is a bitmap (group_by_agg / group_by_agg_opts only)
_NP_TO_CODE = {
    np.dtype(np.int8): I8, np.dtype(np.int16): I16, np.dtype(np.int32): I32, np.dtype(np.int64): I64,
    np.dtype(np.uint8): U8, np.dtype(np.uint16): U16, np.dtype(np.uint32): U32, np.dtype(np.uint64): U64,
    np.dtype(np.float32): F32, np.dtype(np.float64): F64,
}
_CODE_TO_NP = {v: k for k, v in _NP_TO_CODE.items()}
DTYPE_NAMES = {I8: "i8", I16: "i16", I32: "i32", I64: "i64", U8: "u8", U16: "u16", U32: "u32", U64: "u64",
               F32: "f32", F64: "f64", BOOL: "bool"}

HOST, DEVICE = 0, 1

# plhip_agg_kind_t
AGG_SUM, AGG_MEAN, AGG_MIN, AGG_MAX, AGG_COUNT, AGG_FIRST, AGG_LAST, AGG_VAR, AGG_STD, AGG_MEDIAN, AGG_N_UNIQUE = range(11)
AGG_NAMES = {"sum": AGG_SUM, "mean": AGG_MEAN, "min": AGG_MIN, "max": AGG_MAX, "count": AGG_COUNT}
AGG_GROUPS_ONLY = {"first": AGG_FIRST, "last": AGG_LAST, "var": AGG_VAR, "std": AGG_STD, "median": AGG_MEDIAN, "n_unique": AGG_N_UNIQUE}  # plhip_agg_groups only

FLAG_MAINTAIN_ORDER = 1
FLAG_OUTPUT_DEVICE = 2
FLAG_KEYS_SORTED = 8
FLAG_PARTIAL = 4
FLAG_KEYS_DENSE = 16
FLAG_EXACT_FOLD = 64
TAKE_NULLABLE_IDX = 256

STRATEGIES = {"auto": 0, "global": 1, "lds": 2, "partitioned": 3, "direct": 4, "tilesort": 5, "sort": 6, "tilesort2": 7, "dense": 8}

STATUS_NAMES = {
    0: "OK", 1: "ComputeError", 2: "ShapeMismatch", 3: "OutOfMemory", 4: "UnsupportedDtype", 5: "HipError",
    6: "InvalidArgument", 7: "NoDevice",
}


class PlHipError(RuntimeError):
    """A non-OK plhip_status_t; `.status` holds the code, the message comes from plhip_last_error()."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class ComputeError(PlHipError):
    pass


class ShapeMismatch(PlHipError):
    pass


_ERR_CLASS = {1: ComputeError, 2: ShapeMismatch}


class ColumnStruct(C.Structure):
    """plhip_column_t"""

    _fields_ = [
        ("values", C.c_void_p),
        ("validity", C.c_void_p),
        ("offset", C.c_int64),
        ("length", C.c_int64),
        ("dtype", C.c_int32),
        ("location", C.c_int32),
    ]


class HashSeeds(C.Structure):
    """plhip_hash_seeds_t"""

    _fields_ = [("null_hash", C.c_uint64), ("true_hash", C.c_uint64), ("false_hash", C.c_uint64)]


class AggStruct(C.Structure):
    """plhip_agg_t"""

    _fields_ = [("column", C.c_int32), ("kind", C.c_int32)]


class FinalStruct(C.Structure):
    """plhip_final_t"""

    _fields_ = [("kind", C.c_int32), ("src_dtype", C.c_int32)]


class OptionsStruct(C.Structure):
    """plhip_options_t"""

    _fields_ = [("struct_size", C.c_uint64), ("n_groups_hint", C.c_int64), ("slice_offset", C.c_int64),
                ("slice_len", C.c_uint64), ("has_slice", C.c_uint32), ("pad", C.c_uint32)]


def np_dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt not in _NP_TO_CODE:
        raise PlHipError(4, f"dtype {dt} is not supported on the group-by path")
    return _NP_TO_CODE[dt]


def code_np_dtype(code: int):
    return np.dtype(np.bool_) if code == BOOL else _CODE_TO_NP[code]


class ChunkedColumnStruct(C.Structure):
    """plhip_chunked_column_t"""
    _fields_ = [("chunks", C.POINTER(ColumnStruct)), ("n_chunks", C.c_int32), ("pad", C.c_int32)]


class Utf8ColumnStruct(C.Structure):
    """plhip_utf8_column_t"""
    _fields_ = [("offsets", C.c_void_p), ("data", C.c_void_p), ("validity", C.c_void_p), ("offset", C.c_int64), ("length", C.c_int64),
                ("location", C.c_int32), ("pad", C.c_int32)]


class ShardedStats(C.Structure):
    """plhip_sharded_stats_t"""
    _fields_ = [("partial_ms", C.c_float), ("pack_ms", C.c_float), ("exchange_ms", C.c_float), ("merge_ms", C.c_float),
                ("finalize_ms", C.c_float), ("partial_kernel_ms", C.c_float), ("partial_dominant_ms", C.c_float),
                ("partial_breakdown_ms", C.c_float * 3), ("shuffle_bytes", C.c_int64), ("partial_groups", C.c_int64),
                ("protocol", C.c_int32), ("host_syncs", C.c_int32), ("partial_strategy", C.c_char * 16),
                ("partial_dominant_kernel", C.c_char * 32)]


ALLTOALL_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64))
COMM_ID_BYTES = 128
FLAG_EXACT_EXCHANGE = 32

LIB_PATH = Path(__file__).resolve().parent / "_lib" / "libplhip.so"
_lib = None


def _preload_hip_runtime() -> None:
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    soname, libamdhip64.so.7, as /opt/rocm's): if libplhip.so pulled in the system copy first, a later
    `import torch` would load a second runtime and find no GPU.  Loading torch's copy first (by path, without
    importing torch) makes the dynamic linker hand that same instance to libplhip.so.  Without torch
    installed, libplhip.so resolves the system ROCm runtime through its RUNPATH."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def lib():
    """Loads libplhip.so once.  Raises OSError with a build hint when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("PLHIP_LIBRARY", LIB_PATH))
    if not path.exists():
        raise OSError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); polars_amd has no CPU fallback"
        )
    _preload_hip_runtime()
    L = C.CDLL(str(path))
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    P = C.POINTER
    L.plhip_abi_version.restype = i32
    L.plhip_init.argtypes = [i32]
    L.plhip_init.restype = i32
    L.plhip_last_error.restype = C.c_char_p
    L.plhip_set_strategy.argtypes = [i32]
    L.plhip_set_strategy.restype = i32
    L.plhip_last_strategy.restype = C.c_char_p
    L.plhip_last_kernel_ms.argtypes = [P(C.c_float), P(C.c_float), P(C.c_char_p)]
    L.plhip_last_kernel_ms.restype = i32
    L.plhip_memcpy.argtypes = [vp, i32, vp, i32, C.c_size_t]
    L.plhip_memcpy.restype = i32
    L.plhip_last_kernel_breakdown.argtypes = [P(C.c_float), P(C.c_float), P(C.c_float)]
    L.plhip_last_kernel_breakdown.restype = i32
    L.plhip_vec_hash.argtypes = [P(ColumnStruct), i32, u64, vp, i32]
    L.plhip_vec_hash.restype = i32
    L.plhip_take.argtypes = [P(ColumnStruct), i32, vp, i64, i32, u32, P(vp)]
    L.plhip_take.restype = i32
    L.plhip_vec_hash_seeded.argtypes = [P(ColumnStruct), i32, P(HashSeeds), vp, i32]
    L.plhip_vec_hash_seeded.restype = i32
    L.plhip_group_by_agg.argtypes = [P(ColumnStruct), i32, P(ColumnStruct), i32, P(AggStruct), i32, u32, P(vp)]
    L.plhip_group_by_agg.restype = i32
    L.plhip_group_by_agg_opts.argtypes = [P(ColumnStruct), i32, P(ColumnStruct), i32, P(AggStruct), i32, u32,
                                          P(OptionsStruct), P(vp)]
    L.plhip_group_by_agg_opts.restype = i32
    L.plhip_group_by_agg_noalloc.argtypes = [P(ColumnStruct), i32, P(ColumnStruct), i32, P(AggStruct), i32, u32, i64,
                                             P(ColumnStruct), i32, P(i64)]
    L.plhip_group_by_agg_noalloc.restype = i32
    L.plhip_comm_unique_id.argtypes = [vp]
    L.plhip_comm_unique_id.restype = i32
    L.plhip_comm_init_rccl.argtypes = [vp, i32, i32, P(vp)]
    L.plhip_comm_init_rccl.restype = i32
    L.plhip_comm_init_host.argtypes = [i32, i32, ALLTOALL_FN, ALLTOALLV_FN, vp, P(vp)]
    L.plhip_comm_init_host.restype = i32
    L.plhip_comm_free.argtypes = [vp]
    L.plhip_comm_free.restype = None
    L.plhip_comm_info.argtypes = [vp, P(i32), P(i32), P(i32)]
    L.plhip_comm_info.restype = i32
    L.plhip_group_by_agg_sharded.argtypes = [vp, P(ColumnStruct), i32, P(ColumnStruct), i32, P(AggStruct), i32, u32,
                                             P(OptionsStruct), P(vp)]
    L.plhip_group_by_agg_sharded.restype = i32
    L.plhip_sharded_last_stats.argtypes = [P(ShardedStats)]
    L.plhip_sharded_last_stats.restype = i32
    L.plhip_group_by_agg_chunked.argtypes = [P(ChunkedColumnStruct), i32, P(ChunkedColumnStruct), i32, P(AggStruct), i32, u32,
                                             P(OptionsStruct), P(vp)]
    L.plhip_group_by_agg_chunked.restype = i32
    L.plhip_utf8_encode.argtypes = [P(Utf8ColumnStruct), P(vp), P(ColumnStruct)]
    L.plhip_utf8_encode.restype = i32
    L.plhip_dictionary_len.argtypes = [vp]
    L.plhip_dictionary_len.restype = C.c_int64
    L.plhip_dictionary_strings.argtypes = [vp, i32, P(Utf8ColumnStruct)]
    L.plhip_dictionary_strings.restype = i32
    L.plhip_dictionary_free.argtypes = [vp]
    L.plhip_dictionary_free.restype = None
    L.plhip_hash_join.argtypes = [P(ColumnStruct), P(ColumnStruct), i32, i32, P(vp)]
    L.plhip_hash_join.restype = i32
    L.plhip_hash_join_multi.argtypes = [P(ColumnStruct), P(ColumnStruct), i32, i32, i32, P(vp)]
    L.plhip_hash_join_multi.restype = i32
    L.plhip_join_num_rows.argtypes = [vp]
    L.plhip_join_num_rows.restype = C.c_int64
    L.plhip_join_buffers.argtypes = [vp, P(vp), P(vp), P(i32)]
    L.plhip_join_buffers.restype = i32
    L.plhip_join_free.argtypes = [vp]
    L.plhip_join_free.restype = None
    L.plhip_set_timing_interval.argtypes = [i32]
    L.plhip_set_timing_interval.restype = i32
    L.plhip_alloc_pinned.argtypes = [C.c_size_t, P(vp)]
    L.plhip_alloc_pinned.restype = i32
    L.plhip_free_pinned.argtypes = [vp]
    L.plhip_free_pinned.restype = i32
    L.plhip_release_workspace.argtypes = []
    L.plhip_release_workspace.restype = i32
    L.plhip_pooled_buffers.argtypes = [i32]
    L.plhip_pooled_buffers.restype = i64
    L.plhip_result_num_rows.argtypes = [vp]
    L.plhip_result_num_rows.restype = i64
    L.plhip_result_num_columns.argtypes = [vp]
    L.plhip_result_num_columns.restype = i32
    L.plhip_result_column.argtypes = [vp, i32, P(ColumnStruct)]
    L.plhip_result_column.restype = i32
    L.plhip_result_free.argtypes = [vp]
    L.plhip_result_free.restype = None
    L.plhip_group_tuples.argtypes = [P(ColumnStruct), i32, i32, i32, P(vp)]
    L.plhip_group_tuples.restype = i32
    L.plhip_groups_num_groups.argtypes = [vp]
    L.plhip_groups_num_groups.restype = i64
    L.plhip_groups_num_rows.argtypes = [vp]
    L.plhip_groups_num_rows.restype = i64
    L.plhip_groups_buffers.argtypes = [vp, P(vp), P(vp), P(vp), P(i32)]
    L.plhip_groups_buffers.restype = i32
    L.plhip_groups_free.argtypes = [vp]
    L.plhip_groups_free.restype = None
    L.plhip_agg_groups.argtypes = [P(ColumnStruct), i32, vp, u32, P(vp)]
    L.plhip_agg_groups.restype = i32
    L.plhip_agg_groups_quantile.argtypes = [P(ColumnStruct), vp, C.c_double, i32, u32, P(vp)]
    L.plhip_agg_groups_quantile.restype = i32
    L.plhip_result_record_bytes.argtypes = [vp]
    L.plhip_result_record_bytes.restype = i64
    L.plhip_result_pack_by_owner.argtypes = [vp, i32, u64, i32, vp, P(i64)]
    L.plhip_result_pack_by_owner.restype = i32
    L.plhip_records_to_frame.argtypes = [vp, i64, P(i32), P(C.c_uint8), i32, P(vp)]
    L.plhip_records_to_frame.restype = i32
    L.plhip_merge_records.argtypes = [vp, i64, P(i32), P(C.c_uint8), i32, i32, P(i32), u32, P(OptionsStruct), P(vp)]
    L.plhip_merge_records.restype = i32
    L.plhip_slot_bytes.argtypes = [i64, i64]
    L.plhip_slot_bytes.restype = i64
    L.plhip_result_pack_slots.argtypes = [vp, i32, u64, i32, i64, vp, P(i64)]
    L.plhip_result_pack_slots.restype = i32
    L.plhip_merge_slots.argtypes = [vp, i32, i64, P(i64), P(i32), P(C.c_uint8), i32, i32, P(i32), u32, P(OptionsStruct), P(vp)]
    L.plhip_merge_slots.restype = i32
    L.plhip_finalize_partial.argtypes = [vp, i32, P(FinalStruct), i32, P(ColumnStruct)]
    L.plhip_finalize_partial.restype = i32
    if L.plhip_abi_version() != 1:
        raise OSError("libplhip.so ABI version mismatch")
    _lib = L
    return L


def check(status: int) -> None:
    if status != 0:
        msg = lib().plhip_last_error().decode("utf-8", "replace")
        raise _ERR_CLASS.get(status, PlHipError)(status, msg)


EXPORTED_SYMBOLS = [
    "plhip_abi_version", "plhip_init", "plhip_last_error", "plhip_set_strategy", "plhip_last_strategy",
    "plhip_last_kernel_ms", "plhip_memcpy", "plhip_last_kernel_breakdown", "plhip_vec_hash", "plhip_vec_hash_seeded", "plhip_take", "plhip_group_by_agg", "plhip_group_by_agg_opts", "plhip_result_num_rows",
    "plhip_result_num_columns", "plhip_result_column", "plhip_result_free", "plhip_group_tuples",
    "plhip_groups_num_groups", "plhip_groups_num_rows", "plhip_groups_buffers", "plhip_groups_free",
    "plhip_agg_groups", "plhip_agg_groups_quantile", "plhip_result_record_bytes", "plhip_result_pack_by_owner", "plhip_records_to_frame", "plhip_merge_records", "plhip_finalize_partial", "plhip_slot_bytes", "plhip_result_pack_slots", "plhip_merge_slots",
    "plhip_group_by_agg_noalloc", "plhip_release_workspace", "plhip_pooled_buffers",
    "plhip_comm_unique_id", "plhip_comm_init_rccl", "plhip_comm_init_host", "plhip_comm_free", "plhip_comm_info", "plhip_group_by_agg_sharded",
    "plhip_sharded_last_stats", "plhip_group_by_agg_chunked",
    "plhip_utf8_encode", "plhip_dictionary_len", "plhip_dictionary_strings", "plhip_dictionary_free",
    "plhip_set_timing_interval", "plhip_alloc_pinned", "plhip_free_pinned", "plhip_hash_join", "plhip_hash_join_multi", "plhip_join_num_rows", "plhip_join_buffers", "plhip_join_free",
]
