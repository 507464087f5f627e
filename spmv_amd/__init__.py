"""spmv_amd -- MI355X-native SpMV engine behind the reference's create_X/mult_X boundary.

The product is libspmv_hip.so (C ABI in include/spmv_hip.h, HIP sources in
spmv_amd/csrc).  This package is the thin Python host side used by the tests and
bench.py: ctypes bindings (_lib), the format/mode mirror of the reference driver
(matrix), synthetic workload generators (generators) and the row-block
multi-GPU layer (distributed).
"""
from ._lib import (F64, F32, CSR_SCALAR, CSR_VECTOR, CSR_MERGE, CSR_BAL, SELL, SELLP, ELL, SELL_DFC,  # noqa: F401
                   ARR_ROW_PTR, ARR_COL, ARR_VAL, ARR_SLICE_PTR, ARR_ROW_ORDER, ARR_TILE_COORD, ARR_VOP_PTR, ARR_VOP_LEN,
                   FORMAT_NAMES, SpmvHipError, lib, check, LIB_PATH)
from .matrix import (Matrix, COO_DTYPE, make_coo, create, MODES, sort_coo, load_mtx, load_bin, save_bin,  # noqa: F401
                     coo_to_csr_host)
