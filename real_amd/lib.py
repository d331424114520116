"""ctypes binding of the C ABI in include/real_hip.h (libreal_hip.so).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible,
the calls raise.  PyTorch is only used by callers for device buffers, streams and
torch.distributed -- nothing here depends on it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REAL_HIP_LIB") or os.path.join(_HERE, "libreal_hip.so")     # (REAL_HIP_LIB: an experimental build, bench_support/ab_match.py)

REAL_HIP_OK = 0
REAL_HIP_E_INVALID = -1
REAL_HIP_E_NOMEM = -2
REAL_HIP_E_DEVICE = -3
REAL_HIP_E_OVERFLOW = -4
REAL_HIP_E_STATE = -5
REAL_HIP_E_UNSUPPORTED = -6
REAL_HIP_MAX_PATL = 320
REAL_HIP_MAX_PATL_LONG = 16384

K_MATCH_UNIQUE, K_MATCH_ALL, K_ALL_SORT, K_INDEX, K_MATCH_REPEAT, K_PARSE = range(6)

# every symbol include/real_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "real_hip_scoring_table", "real_hip_create", "real_hip_destroy", "real_hip_strerror",
    "real_hip_last_error", "real_hip_abi_version", "real_hip_set_match_params", "real_hip_wait_event", "real_hip_device_memory", "real_hip_set_text", "real_hip_set_text_symbols",
    "real_hip_set_index_block", "real_hip_build_index_block", "real_hip_index_info", "real_hip_index_build_stats",
    "real_hip_index_table_kind", "real_hip_index_download", "real_hip_index_export", "real_hip_match_unique", "real_hip_match_all", "real_hip_match_unique_submit", "real_hip_wait",
    "real_hip_host_alloc", "real_hip_host_free",
    "real_hip_comm_id", "real_hip_comm_init", "real_hip_comm_destroy", "real_hip_gather_records", "real_hip_gather_hits",
    "real_hip_parse_reads", "real_hip_download", "real_hip_counters_get", "real_hip_kernel_time", "real_hip_timing_enable",
]


class RealHipParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("seedl", C.c_uint32), ("seedkmax", C.c_uint32),
                ("totalkmax", C.c_uint32), ("scores", C.c_uint32), ("prefix_bits", C.c_uint32),
                ("device", C.c_int32), ("table_kind", C.c_uint32), ("filter_mult", C.c_double),
                ("LL", C.c_double * 1024)]


class RealHipBatch(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("on_device", C.c_uint32), ("n_reads", C.c_uint64),
                ("bases", C.c_void_p), ("qual", C.c_void_p), ("offsets", C.c_void_p),
                ("patl", C.c_uint32), ("max_patl", C.c_uint32),
                ("packed", C.c_uint32), ("fresh", C.c_uint32), ("nflags", C.c_void_p)]


class RealHipParsed(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_patl", C.c_uint32), ("n_reads", C.c_uint64), ("n_symbols", C.c_uint64),
                ("bases", C.c_void_p), ("qual", C.c_void_p), ("offsets", C.c_void_p),
                ("id_start", C.c_void_p), ("id_len", C.c_void_p)]


class RealHipBuildStats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("wall_ms", C.c_double), ("kernel_ms", C.c_double),
                ("alloc_ms", C.c_double), ("free_ms", C.c_double), ("alloc_bytes", C.c_uint64), ("alloc_calls", C.c_uint64),
                ("free_calls", C.c_uint64)]


class RealHipCounters(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("lookups", C.c_uint64), ("probes", C.c_uint64),
                ("candidates", C.c_uint64), ("seedpass", C.c_uint64), ("hits", C.c_uint64),
                ("verified", C.c_uint64), ("handed_over", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


HIT_DTYPE = np.dtype([("read", "<u4"), ("pos", "<u4"), ("score", "<f4"), ("frag", "<u2"),
                      ("k", "u1"), ("inverted", "u1")])
assert HIT_DTYPE.itemsize == 16


class RealHipError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__("real_hip status %d: %s" % (status, msg))
        self.status = status


_lib = None


def load():
    """Load libreal_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RealHipError(REAL_HIP_E_DEVICE, "libreal_hip.so is not built: run __graft_entry__.build() "
                                              "(make -C real_amd/csrc); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.real_hip_scoring_table.argtypes = [C.c_double] * 5 + [vp]
    L.real_hip_scoring_table.restype = None
    L.real_hip_create.argtypes = [C.POINTER(vp), C.POINTER(RealHipParams)]
    L.real_hip_destroy.argtypes = [vp]
    L.real_hip_destroy.restype = None
    L.real_hip_strerror.argtypes = [C.c_int]
    L.real_hip_strerror.restype = C.c_char_p
    L.real_hip_last_error.argtypes = [vp]
    L.real_hip_last_error.restype = C.c_char_p
    L.real_hip_set_match_params.argtypes = [vp, u32, u32, u32, C.c_double]
    L.real_hip_wait_event.argtypes = [vp, vp]
    L.real_hip_device_memory.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.real_hip_set_text.argtypes = [vp, u32, vp, vp, u64, vp, u32]
    L.real_hip_set_text_symbols.argtypes = [vp, u32, vp, u64, C.c_int, vp, u32]
    L.real_hip_set_index_block.argtypes = [vp, u64, C.POINTER(vp), C.POINTER(vp)]
    L.real_hip_build_index_block.argtypes = [vp, u64, u64, C.POINTER(u64), C.POINTER(C.c_int)]
    L.real_hip_index_build_stats.argtypes = [vp, C.POINTER(RealHipBuildStats), C.c_int]
    L.real_hip_index_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u32)]
    L.real_hip_index_table_kind.argtypes = [vp, C.POINTER(u32)]
    L.real_hip_parse_reads.argtypes = [vp, vp, u64, C.c_int, C.c_int, C.c_int, C.POINTER(RealHipParsed)]
    L.real_hip_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.real_hip_index_download.argtypes = [vp, C.c_int, vp, vp]
    L.real_hip_index_export.argtypes = [vp, C.c_int, vp, vp]
    L.real_hip_match_unique.argtypes = [vp, C.POINTER(RealHipBatch), vp, vp]
    L.real_hip_match_unique_submit.argtypes = [vp, C.POINTER(RealHipBatch), vp, vp, u32, C.c_int]
    L.real_hip_wait.argtypes = [vp, u32]
    L.real_hip_host_alloc.argtypes = [C.c_size_t]
    L.real_hip_host_alloc.restype = vp
    L.real_hip_host_free.argtypes = [vp]
    L.real_hip_host_free.restype = None
    L.real_hip_comm_id.argtypes = [vp]
    L.real_hip_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.real_hip_comm_destroy.argtypes = [vp]
    L.real_hip_gather_records.argtypes = [vp, C.c_int, vp, vp, u64, vp, vp, u64, C.POINTER(u64)]
    L.real_hip_gather_hits.argtypes = [vp, C.c_int, vp, vp, u64, u64, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.real_hip_match_all.argtypes = [vp, C.POINTER(RealHipBatch), vp, u64, C.POINTER(u64), vp]
    L.real_hip_counters_get.argtypes = [vp, C.POINTER(RealHipCounters), C.c_int]
    L.real_hip_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64), C.c_int]
    L.real_hip_timing_enable.argtypes = [vp, C.c_int]
    _lib = L
    return L


def scoring_table(similarity=0.995, gc=0.41, trans=0.71, err=0.0, gcmut_bias=2.0) -> np.ndarray:
    """Scoring::init defaults: Scoring.cpp:204-208."""
    LL = np.zeros(1024, dtype=np.float64)
    load().real_hip_scoring_table(similarity, gc, trans, err, gcmut_bias, LL.ctypes.data)
    return LL


def _ptr(a) -> Optional[int]:
    """host numpy array, torch tensor (host or device) or raw int address -> address."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(type(a))
