"""ctypes binding of libspmv_hip.so (the C ABI declared in include/spmv_hip.h).

There is no fallback: if the library has not been built, importing a symbol
raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C spmv_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspmv_hip.so")

F64, F32 = 0, 1
CSR_SCALAR, CSR_VECTOR, CSR_MERGE, CSR_BAL, SELL, SELLP, ELL, SELL_DFC = range(8)
ARR_ROW_PTR, ARR_COL, ARR_VAL, ARR_SLICE_PTR, ARR_ROW_ORDER, ARR_TILE_COORD, ARR_VOP_PTR, ARR_VOP_LEN = range(8)

FORMAT_NAMES = {CSR_SCALAR: "csr_scalar", CSR_VECTOR: "csr_vector", CSR_MERGE: "csr_merge",
                CSR_BAL: "csr_bal", SELL: "sell", SELLP: "sellp", ELL: "ell", SELL_DFC: "sell_dfc"}


class Info(C.Structure):
    _fields_ = [("format", C.c_int), ("dtype", C.c_int), ("rows", C.c_int), ("columns", C.c_int),
                ("nnz", C.c_int64), ("stored", C.c_int64), ("slice_height", C.c_int),
                ("sigma", C.c_int64), ("nslices", C.c_int), ("lanes_per_row", C.c_int),
                ("ntiles", C.c_int), ("items_per_tile", C.c_int), ("device_bytes", C.c_int64),
                ("variant", C.c_int), ("hot_columns", C.c_int), ("diag_columns", C.c_int64), ("pattern_blocks", C.c_int), ("hot_variant", C.c_int),
                ("block_map", C.c_int)]


class RowBlockInfo(C.Structure):
    _fields_ = [("format", C.c_int), ("dtype", C.c_int), ("device", C.c_int), ("rows", C.c_int), ("row_begin", C.c_int),
                ("columns_global", C.c_int), ("nnz", C.c_int64), ("interior_nnz", C.c_int64), ("boundary_nnz", C.c_int64),
                ("boundary_rows", C.c_int), ("halo", C.c_int), ("send", C.c_int), ("interior_stored", C.c_int64),
                ("interior_variant", C.c_int), ("boundary_lanes_per_row", C.c_int), ("boundary_variant", C.c_int),
                ("device_bytes", C.c_int64)]


# name -> (restype, argtypes); every symbol include/spmv_hip.h declares
_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
SIGNATURES = {
    "spmv_hip_init": (_i, [_i]),
    "spmv_hip_last_error": (C.c_char_p, []),
    "spmv_hip_mem_info": (_i, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "spmv_hip_load_mtx": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp)]),
    "spmv_hip_free_coo": (None, [_vp]),
    "spmv_hip_load_bin": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp)]),
    "spmv_hip_save_bin": (_i, [C.c_char_p, _i, _i, _i, _vp]),
    "spmv_hip_sort_coo": (None, [_i, _vp]),
    "spmv_hip_create_csr": (_vp, [_i, _i, _i, _i, _vp, _i, _i, _i]),
    "spmv_hip_create_csr_merge": (_vp, [_i, _i, _i, _vp, _i, _i]),
    "spmv_hip_create_csr_bal": (_vp, [_i, _i, _i, _vp, _i, _i]),
    "spmv_hip_create_ell": (_vp, [_i, _i, _i, _vp, _i, _i]),
    "spmv_hip_create_sellp": (_vp, [_i, _i, _i, _vp, _vp, _vp, _i, _i]),
    "spmv_hip_create_sell": (_vp, [_i, _i, _i, _vp, _vp, _vp, _i, _i, _i64]),
    "spmv_hip_create_sell_dfc": (_vp, [_i, _i, _i, _vp, _vp, _vp, _i, _i, _i64]),
    "spmv_hip_create_from_device_csr": (_vp, [_i, _i, _i, _i64, _vp, _vp, _vp, _i, _i64, _i64, _vp]),
    "spmv_hip_create_auto": (_vp, [_i, _i, _i, _vp, _i, _i]),
    "spmv_hip_create_auto_from_device_csr": (_vp, [_i, _i, _i64, _vp, _vp, _vp, _i, _i, _vp]),
    "spmv_hip_mult": (_i, [_vp, _vp, _vp]),
    "spmv_hip_mult_dev": (_i, [_vp, _vp, _vp, _vp]),
    "spmv_hip_mult_mv": (_i, [_vp, _i, _vp, _vp]),
    "spmv_hip_mult_mv_dev": (_i, [_vp, _i, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "spmv_hip_set_row_map": (_i, [_vp, _vp]),
    "spmv_hip_free": (_i, [_vp]),
    "spmv_hip_bench": (_i, [_vp, _i, C.c_double, _vp, C.POINTER(_i), C.POINTER(C.c_double)]),
    "spmv_hip_norm2_dev": (_i, [_i, _vp, _i, _vp, _vp]),
    "spmv_hip_release_stream": (_i, [_vp]),
    "spmv_hip_norm2_dev_noalloc": (_i, [_i64, _vp, _i, _vp, _vp]),
    "spmv_hip_scale_dev": (_i, [_i, _vp, _vp, _i, _vp, _vp]),
    "spmv_hip_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "spmv_hip_analyze": (_i, [_vp, _i]),
    "spmv_hip_get_info": (_i, [_vp, C.POINTER(Info)]),
    "spmv_hip_download": (_i, [_vp, _i, _vp, C.c_size_t]),
    "spmv_hip_partition_rows": (_i, [_i, _vp, _i, _i, _vp]),
    "spmv_hip_partition_rows64": (_i, [_i, _vp, _i, _i, _vp]),
    "spmv_hip_rowblock_create": (_vp, [_i, _i, _i, _i, _i64, _vp, _vp, _vp, _i, _i64, _i64, _vp]),
    "spmv_hip_rowblock_free": (_i, [_vp]),
    "spmv_hip_rowblock_get_info": (_i, [_vp, C.POINTER(RowBlockInfo)]),
    "spmv_hip_rowblock_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "spmv_hip_rowblock_analyze": (_i, [_vp, _i]),
    "spmv_hip_rowblock_halo_columns": (_i, [_vp, _vp]),
    "spmv_hip_rowblock_set_send_list": (_i, [_vp, _i, _vp]),
    "spmv_hip_rowblock_pack": (_i, [_vp, _vp, _vp, _vp]),
    "spmv_hip_rowblock_push": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "spmv_hip_rowblock_mult_interior": (_i, [_vp, _vp, _vp, _vp]),
    "spmv_hip_rowblock_mult_boundary": (_i, [_vp, _vp, _vp, _vp]),
    "spmv_hip_dist_create": (_vp, [_i, _i, _i, _i, _vp, _i, _i64, _i64, _i, _vp]),
    "spmv_hip_dist_create_from_blocks": (_vp, [_i, _vp]),
    "spmv_hip_dist_analyze": (_i, [_vp, _i]),
    "spmv_hip_dist_issue_time": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "spmv_hip_dist_get_peer_access": (_i, [_vp, _vp]),
    "spmv_hip_dist_mult": (_i, [_vp, _vp, _vp]),
    "spmv_hip_dist_bench": (_i, [_vp, _i, C.c_double, _vp, C.POINTER(_i), C.POINTER(C.c_double)]),
    "spmv_hip_dist_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "spmv_hip_dist_get_info": (_i, [_vp, C.POINTER(_i), _vp, _vp]),
    "spmv_hip_dist_free": (_i, [_vp]),
}

_lib = None


def lib():
    """The loaded library.  Raises OSError if libspmv_hip.so is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} not built (make -C spmv_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class SpmvHipError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise SpmvHipError(lib().spmv_hip_last_error().decode())


def check_handle(h):
    if not h:
        raise SpmvHipError(lib().spmv_hip_last_error().decode())
    return h
