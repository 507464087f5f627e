"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the CPU oracle.

`oracle.lib()` loads oracle/liboracle.so (the from-scratch C restatement,
oracle/spmv_oracle.c).  `oracle.ref()` loads oracle/_ref/libspmv_ref.so (the
reference's own sources compiled by oracle/Makefile) when it has been built.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package, and only as the checker.  Parity status: pinned (see spmv_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libspmv_ref.so")

COO_DTYPE = np.dtype([("i", np.int32), ("j", np.int32), ("a", np.float64)], align=True)
assert COO_DTYPE.itemsize == 16

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_coop = np.ctypeslib.ndpointer(COO_DTYPE, flags="C_CONTIGUOUS")


def build(with_ref: bool = True) -> None:
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    target = "all" if with_ref else "oracle"
    subprocess.run(["make", "-C", _HERE, "-j8", target], check=True,
                   stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        build(with_ref=False)
    L = C.CDLL(ORACLE_SO)
    sig = {
        "orc_sort_coo": (None, [C.c_int64, _coop]),
        "orc_mult_coo": (None, [C.c_int, C.c_int64, _coop, _f64p, _f64p]),
        "orc_load_mtx": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
        "orc_free": (None, [C.c_void_p]),
        "orc_csr_count": (C.c_int64, [C.c_int, C.c_int64, C.c_int, _coop]),
        "orc_csr_fill": (None, [C.c_int, C.c_int64, C.c_int, _coop, _i32p, _i32p, _f64p]),
        "orc_mult_csr_base": (None, [C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p]),
        "orc_mult_mv_csr": (None, [C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
        "orc_mult_csr_base_f32": (None, [C.c_int, _i32p, _i32p, _f32p, _f32p, _f32p]),
        "orc_mult_csr_omp": (None, [C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p]),
        "orc_csr_copy_first_touch": (None, [C.c_int, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, _f64p, _f64p]),
        "orc_merge_path_search": (None, [C.c_int, _i32p, C.c_int, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_mult_csr_merge": (None, [C.c_int, C.c_int, _i32p, _i32p, _f64p, _f64p, _f64p,
                                      C.c_int, _i32p, _f64p]),
        "orc_csr_bal_partition": (None, [C.c_int64, C.c_int, _coop, _i64p, _i64p, _i32p, _i32p, _i32p]),
        "orc_mult_csr_bal": (None, [C.c_int64, C.c_int, _coop, _f64p, _f64p]),
        "orc_sell_row_order": (None, [C.c_int, _i32p, C.c_int64, _i32p, _i32p]),
        "orc_sell_slices": (C.c_int64, [C.c_int, C.c_int, _i32p, _i32p, _i64p]),
        "orc_sell_fill": (None, [C.c_int, C.c_int, _i32p, _i32p, _f64p, _i32p, _i64p, C.c_int64,
                                 _i32p, _f64p]),
        "orc_mult_sell": (None, [C.c_int, C.c_int, _i32p, _i64p, _i32p, _i32p, _f64p, _f64p, _f64p]),
        "orc_sell_dfc": (C.c_int64, [C.c_int, C.c_int, _i32p, _i32p, _i32p, _f64p, _i32p, _i32p, _i64p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "orc_mult_sell_dfc": (None, [C.c_int, C.c_int, _i32p, _i64p, _u8p, _i64p, _i32p, _i32p, _f64p, _f64p, _f64p]),
        "orc_sliced_ell_blocks": (C.c_int64, [C.c_int, C.c_int, _i32p, _i64p]),
        "orc_sliced_ell_fill": (None, [C.c_int, C.c_int, _i32p, _i32p, _f64p, _i64p, _i32p, _f64p]),
        "orc_mult_sliced_ell": (None, [C.c_int, C.c_int, _i64p, _i32p, _f64p, _f64p, _f64p]),
        "orc_vec_norm": (C.c_double, [C.c_int, _f64p]),
        "orc_rel_l2": (C.c_double, [C.c_int, _f64p, _f64p]),
        "orc_driver_init_xy": (None, [C.c_int, _f64p, _f64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


# ---------------------------------------------------------------- helpers ----

def make_coo(i, j, a) -> np.ndarray:
    coo = np.zeros(len(i), dtype=COO_DTYPE)
    coo["i"], coo["j"], coo["a"] = i, j, a
    return coo


def sort_coo(coo: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(coo.copy())
    lib().orc_sort_coo(len(out), out)
    return out


def mult_coo(rows, coo, x):
    y = np.empty(rows, np.float64)
    lib().orc_mult_coo(rows, len(coo), np.ascontiguousarray(coo), x, y)
    return y


def load_mtx(path: str):
    r, c, n = C.c_int(), C.c_int(), C.c_int()
    p = C.c_void_p()
    rc = lib().orc_load_mtx(path.encode(), C.byref(r), C.byref(c), C.byref(n), C.byref(p))
    if rc != 0:
        raise ValueError(f"orc_load_mtx({path}) -> {rc}")
    buf = (C.c_char * (16 * n.value)).from_address(p.value)
    coo = np.frombuffer(buf, dtype=COO_DTYPE, count=n.value).copy()
    lib().orc_free(p)
    return r.value, c.value, coo


def csr_from_coo(rows, coo_sorted, alignment=1):
    L = lib()
    coo_sorted = np.ascontiguousarray(coo_sorted)
    p = L.orc_csr_count(rows, len(coo_sorted), alignment, coo_sorted)
    rp = np.empty(rows + 1, np.int32)
    cj = np.empty(max(p, 1), np.int32)[:p]
    va = np.empty(max(p, 1), np.float64)[:p]
    cj = np.ascontiguousarray(cj); va = np.ascontiguousarray(va)
    L.orc_csr_fill(rows, len(coo_sorted), alignment, coo_sorted, rp, cj, va)
    return rp, cj, va


def mult_csr_base(rp, cj, va, x):
    y = np.empty(len(rp) - 1, np.float64)
    lib().orc_mult_csr_base(len(rp) - 1, rp, cj, va, x, y)
    return y


def mult_csr_base_rows_parallel(rp, cj, va, x):
    """mult_csr_base with the rows dealt over the OpenMP team (the pattern of csr_numa.c:3-23): every row is
    still summed by one thread in ascending order, so the bits are those of mult_csr_base.  For the
    full-size parity tests, where one core would take seconds per product."""
    y = np.empty(len(rp) - 1, np.float64)
    lib().orc_mult_csr_omp(len(rp) - 1, rp, cj, va, x, y)
    return y


def abs_product(rp, cj, va, x):
    """(|A| |x|)_i -- the scale of the component-wise error bound |y - y_ref|_i <= tol * (|A||x|)_i."""
    return mult_csr_base_rows_parallel(rp, cj, np.abs(va), np.abs(x))


def mult_mv_csr(rp, cj, va, X):
    """X[n, columns] -> Y[n, rows]; n products of mult_csr_base."""
    X = np.ascontiguousarray(X, np.float64)
    Y = np.empty((X.shape[0], len(rp) - 1), np.float64)
    lib().orc_mult_mv_csr(len(rp) - 1, rp, cj, va, X.shape[0], X.ctypes.data, X.shape[1], Y.ctypes.data, Y.shape[1])
    return Y


def mult_csr_base_f32(rp, cj, va32, x32):
    y = np.empty(len(rp) - 1, np.float32)
    lib().orc_mult_csr_base_f32(len(rp) - 1, rp, cj, va32, x32, y)
    return y


def merge_path_search(diagonal, rp, nz):
    a, b = C.c_int(), C.c_int()
    n = len(rp) - 1
    lib().orc_merge_path_search(diagonal, np.ascontiguousarray(rp[1:]), n, nz, C.byref(a), C.byref(b))
    return a.value, b.value


def mult_csr_merge(rp, cj, va, x, nthreads):
    n = len(rp) - 1
    y = np.full(n, np.nan)
    rc = np.empty(nthreads, np.int32)
    vc = np.empty(nthreads, np.float64)
    lib().orc_mult_csr_merge(n, len(cj), rp, cj, va, x, y, nthreads, rc, vc)
    return y, rc, vc


def csr_bal_partition(coo_sorted, threads):
    s = np.empty(threads, np.int64); l = np.empty(threads, np.int64)
    fr = np.empty(threads, np.int32); lr = np.empty(threads, np.int32); nr = np.empty(threads, np.int32)
    lib().orc_csr_bal_partition(len(coo_sorted), threads, np.ascontiguousarray(coo_sorted), s, l, fr, lr, nr)
    return s, l, fr, lr, nr


def mult_csr_bal(rows, coo_sorted, x, threads):
    y = np.full(rows, np.nan)
    lib().orc_mult_csr_bal(len(coo_sorted), threads, np.ascontiguousarray(coo_sorted), x, y)
    return y


def sell_from_csr(rp, cj, va, C_=64, sigma=16384):
    """Returns dict(row_order, slice_widths, slice_ptr, col, val, rows_size_sorted)."""
    L = lib()
    n = len(rp) - 1
    rs = np.empty(max(n, 1), np.int32)[:n]; ro = np.empty(max(n, 1), np.int32)[:n]
    rs = np.ascontiguousarray(rs); ro = np.ascontiguousarray(ro)
    L.orc_sell_row_order(n, rp, sigma, rs, ro)
    ns = (n + C_ - 1) // C_
    sw = np.empty(ns, np.int32); sp = np.empty(ns + 1, np.int64)
    padded = L.orc_sell_slices(n, C_, rs, sw, sp)
    sc = np.zeros(padded, np.int32); sv = np.zeros(padded, np.float64)
    L.orc_sell_fill(n, C_, rp, cj, va, ro, sp, padded, sc, sv)
    return dict(row_order=ro, slice_widths=sw, slice_ptr=sp, col=sc, val=sv, rows_size_sorted=rs,
                C=C_, sigma=sigma, nrows=n)


def sell_dfc_from_csr(rp, cj, va, C_=64, sigma=16384):
    """SELL-C-sigma without padding (sellcs_format.c:53-136): dict(row_order, slice_widths, vop_ptr, vop_len,
    slice_ptr, col, val)."""
    L = lib()
    n = len(rp) - 1
    rs = np.ascontiguousarray(np.empty(max(n, 1), np.int32)[:n]); ro = np.ascontiguousarray(np.empty(max(n, 1), np.int32)[:n])
    L.orc_sell_row_order(n, rp, sigma, rs, ro)
    ns = (n + C_ - 1) // C_
    sw = np.zeros(max(ns, 1), np.int32)[:ns]; vp = np.zeros(ns + 1, np.int64)
    sw = np.ascontiguousarray(sw)
    vops = L.orc_sell_dfc(n, C_, rs, rp, cj, va, ro, sw, vp, None, None, None, None)
    vl = np.zeros(max(vops, 1), np.uint8); sp = np.zeros(ns + 1, np.int64)
    nnz = int(rp[-1])
    dc = np.zeros(max(nnz, 1), np.int32); dv = np.zeros(max(nnz, 1), np.float64)
    total = L.orc_sell_dfc(n, C_, rs, rp, cj, va, ro, sw, vp, vl.ctypes.data, sp.ctypes.data, dc.ctypes.data,
                           dv.ctypes.data)
    assert total == nnz
    return dict(row_order=ro, slice_widths=sw, vop_ptr=vp, vop_len=vl[:vops], slice_ptr=sp, col=dc[:nnz], val=dv[:nnz],
                rows_size_sorted=rs, C=C_, sigma=sigma, nrows=n)


def mult_sell_dfc(D, x):
    y = np.zeros(D["nrows"], np.float64)
    vl = D["vop_len"] if len(D["vop_len"]) else np.zeros(1, np.uint8)
    dc = D["col"] if len(D["col"]) else np.zeros(1, np.int32)
    dv = D["val"] if len(D["val"]) else np.zeros(1, np.float64)
    lib().orc_mult_sell_dfc(D["nrows"], D["C"], D["slice_widths"], D["vop_ptr"], np.ascontiguousarray(vl), D["slice_ptr"],
                            D["row_order"], np.ascontiguousarray(dc), np.ascontiguousarray(dv), x, y)
    return y


def mult_sell(S, x):
    y = np.zeros(S["nrows"], np.float64)
    lib().orc_mult_sell(S["nrows"], S["C"], S["slice_widths"], S["slice_ptr"], S["row_order"],
                        S["col"], S["val"], x, y)
    return y


def sliced_ell_from_csr(rp, cj, va, B):
    L = lib()
    n = len(rp) - 1
    nb = (n + B - 1) // B
    bp = np.empty(nb + 1, np.int64)
    tot = L.orc_sliced_ell_blocks(n, B, rp, bp)
    j = np.zeros(tot, np.int32); A = np.zeros(tot, np.float64)
    L.orc_sliced_ell_fill(n, B, rp, cj, va, bp, j, A)
    return dict(block_ptr=bp, j=j, A=A, B=B, nrows=n)


def mult_sliced_ell(E, x):
    y = np.full(E["nrows"], np.nan)
    lib().orc_mult_sliced_ell(E["nrows"], E["B"], E["block_ptr"], E["j"], E["A"], x, y)
    return y


def rel_l2(y, y_ref):
    return lib().orc_rel_l2(len(y), np.ascontiguousarray(y, np.float64), np.ascontiguousarray(y_ref, np.float64))


def driver_init_xy(columns):
    x = np.empty(columns); y = np.empty(columns)
    lib().orc_driver_init_xy(columns, x, y)
    return x, y


# ------------------------------------------------------ compiled reference ----

class _CSR(C.Structure):  # spmv.h:41-47
    _fields_ = [("rows", C.c_int), ("columns", C.c_int), ("nnz", C.c_int), ("memusage", C.c_long),
                ("i", C.POINTER(C.c_int)), ("j", C.POINTER(C.c_int)), ("A", C.POINTER(C.c_double))]


class _CSRMerge(C.Structure):  # spmv.h:87-96
    _fields_ = [("rows", C.c_int), ("columns", C.c_int), ("nnz", C.c_int), ("memusage", C.c_long),
                ("row_ptr", C.POINTER(C.c_int)), ("col_ind", C.POINTER(C.c_int)),
                ("val", C.POINTER(C.c_double)), ("nthreads", C.c_int),
                ("row_carry", C.POINTER(C.c_int)), ("value_carry", C.POINTER(C.c_double))]


class _CSRBalThread(C.Structure):  # spmv.h:101-108
    _fields_ = [("rows", C.c_int), ("first_row", C.c_int), ("last_row", C.c_int),
                ("i", C.POINTER(C.c_int)), ("j", C.POINTER(C.c_int)), ("A", C.POINTER(C.c_double))]


class _SELLP(C.Structure):  # sellp.h:3-12
    _fields_ = [("rows", C.c_int), ("columns", C.c_int), ("nnz", C.c_int), ("num_blocks", C.c_int),
                ("block_ptr", C.POINTER(C.c_int)), ("j", C.POINTER(C.c_int)),
                ("A", C.POINTER(C.c_double)), ("memusage", C.c_size_t)]


class _ELL(C.Structure):  # spmv.h:126-133
    _fields_ = [("rows", C.c_int), ("columns", C.c_int), ("nnz", C.c_int), ("memusage", C.c_long),
                ("num_blocks", C.c_int), ("block_ptr", C.POINTER(C.c_int)),
                ("j", C.POINTER(C.c_int64)), ("A", C.POINTER(C.c_double))]


def ref_available() -> bool:
    return os.path.exists(REF_SO)


def ref():
    """ctypes handle on the compiled reference (raises if it has not been built)."""
    global _ref
    if _ref is not None:
        return _ref
    R = C.CDLL(REF_SO)
    vp, ci = C.c_void_p, C.c_int
    sig = {
        "ref_block": (ci, []),
        "ref_load_mtx": (None, [C.c_char_p, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(vp)]),
        "ref_load_bin": (None, [C.c_char_p, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(vp)]),
        "ref_save_bin": (None, [C.c_char_p, ci, ci, ci, _coop]),
        "ref_free": (None, [vp]),
        "ref_sort_mtx": (None, [ci, _coop]),
        "ref_mult_mtx": (None, [ci, ci, ci, _coop, _f64p, _f64p]),
        "ref_create_csr": (C.POINTER(_CSR), [ci, ci, ci, ci, _coop]),
        "ref_free_csr": (None, [C.POINTER(_CSR)]),
        "ref_mult_csr_base": (None, [C.POINTER(_CSR), _f64p, _f64p]),
        "ref_mult_csr_autovec": (None, [C.POINTER(_CSR), _f64p, _f64p]),
        "ref_mult_csr_vec": (None, [C.POINTER(_CSR), _f64p, _f64p]),
        "ref_create_csr_merge": (C.POINTER(_CSRMerge), [ci, ci, ci, ci, _coop]),
        "ref_mult_csr_merge": (None, [C.POINTER(_CSRMerge), _f64p, _f64p]),
        "ref_search_path_on_diag": (None, [ci, _i32p, ci, ci, C.POINTER(ci), C.POINTER(ci)]),
        "ref_create_csr_bal": (vp, [ci, ci, ci, ci, _coop]),
        "ref_mult_csr_bal": (None, [vp, _f64p, _f64p]),
        "ref_csr_bal_threads": (ci, [vp]),
        "ref_csr_bal_thread": (C.POINTER(_CSRBalThread), [vp, ci]),
        "ref_create_sell": (vp, [C.POINTER(_CSR), ci, C.c_uint64]),
        "ref_mult_sell": (None, [vp, _f64p, _f64p]),
        "ref_sell_nslices": (ci, [vp]),
        "ref_sell_slice_widths": (C.POINTER(ci), [vp]),
        "ref_sell_slice_pointers": (C.POINTER(ci), [vp]),
        "ref_sell_row_order": (C.POINTER(ci), [vp]),
        "ref_create_sell_dfc": (vp, [C.POINTER(_CSR), ci, C.c_uint64]),
        "ref_sell_vop_lengths": (C.POINTER(C.c_uint8), [vp]),
        "ref_sell_vop_pointers": (C.POINTER(C.c_uint64), [vp]),
        "ref_sell_column_indices": (C.POINTER(ci), [vp]),
        "ref_sell_values": (C.POINTER(C.c_double), [vp]),
        "ref_sell_shift": (ci, [vp]),
        "ref_free_sell": (None, [vp]),
        "ref_create_sellp": (C.POINTER(_SELLP), [ci, ci, ci, _i32p, _i32p, _f64p]),
        "ref_mult_sellp": (None, [C.POINTER(_SELLP), _f64p, _f64p]),
        "ref_mult_sellp_autovec": (None, [C.POINTER(_SELLP), _f64p, _f64p]),
        "ref_free_sellp": (None, [C.POINTER(_SELLP)]),
        "ref_create_ell": (C.POINTER(_ELL), [ci, ci, ci, _coop]),
        "ref_mult_ell": (None, [C.POINTER(_ELL), _f64p, _f64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(R, name)
        fn.restype, fn.argtypes = res, args
    _ref = R
    return R


def aligned_empty(n, dtype=np.float64, align=64, fill=None):
    """numpy array whose data pointer is `align`-byte aligned (the reference's
    AVX2 kernels use aligned/streaming loads and stores on x, y: sellp.c:308-313;
    its driver allocates them with aligned_alloc(64), spmv.h:276)."""
    dt = np.dtype(dtype)
    raw = np.empty(n * dt.itemsize + align, np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + n * dt.itemsize].view(dt)
    if fill is not None:
        out[:] = fill
    return out


def as_np(ptr, n, dtype):
    """Copy n elements out of a ctypes pointer."""
    if n == 0:
        return np.empty(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
