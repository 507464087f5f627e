"""Host-side mirror of the reference's format plugins on top of libspmv_hip.

`create(mode, rows, columns, coo)` follows the reference driver's MODE table
(reference src/driver_spmv.c:186-261): the same mode strings pick the same
storage format; `Matrix.mult(x)` is `mult(matrix, x, y)` (driver_spmv.c:276).
Every product runs in the HIP library -- there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

COO_DTYPE = np.dtype([("i", np.int32), ("j", np.int32), ("a", np.float64)], align=True)  # spmv.h:26

_NP = {L.F64: np.float64, L.F32: np.float32}


def make_coo(i, j, a) -> np.ndarray:
    coo = np.zeros(len(i), dtype=COO_DTYPE)
    coo["i"], coo["j"], coo["a"] = i, j, a
    return coo


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Matrix:
    """Owns one spmv_hip_matrix handle."""

    def __init__(self, handle):
        self._h = L.check_handle(handle)
        self._info = None

    # ---- creators (spmv.h:50,98,118,135,233; sellp.h:15) ----
    @classmethod
    def csr(cls, rows, columns, coo, alignment=1, dtype=L.F64, kernel=L.CSR_VECTOR, lanes_per_row=0):
        """create_csr: sorts `coo` in place like the reference (mtx.c:18-34)."""
        coo = _as_coo(coo)
        return cls(L.lib().spmv_hip_create_csr(rows, columns, len(coo), alignment, _ptr(coo), dtype, kernel,
                                               lanes_per_row))

    @classmethod
    def csr_merge(cls, rows, columns, coo, dtype=L.F64, items_per_tile=0):
        coo = _as_coo(coo)
        return cls(L.lib().spmv_hip_create_csr_merge(rows, columns, len(coo), _ptr(coo), dtype, items_per_tile))

    @classmethod
    def csr_bal(cls, rows, columns, coo, dtype=L.F64, nnz_per_tile=0):
        coo = _as_coo(coo)
        return cls(L.lib().spmv_hip_create_csr_bal(rows, columns, len(coo), _ptr(coo), dtype, nnz_per_tile))

    @classmethod
    def ell(cls, rows, columns, coo, dtype=L.F64, slice_height=0):
        coo = _as_coo(coo)
        return cls(L.lib().spmv_hip_create_ell(rows, columns, len(coo), _ptr(coo), dtype, slice_height))

    @classmethod
    def sellp(cls, rows, columns, row_ptr, col, val, dtype=L.F64, slice_height=0):
        rp, cj, va = _as_csr(row_ptr, col, val)
        return cls(L.lib().spmv_hip_create_sellp(rows, columns, len(cj), _ptr(rp), _ptr(cj), _ptr(va), dtype,
                                                 slice_height))

    @classmethod
    def sell(cls, rows, columns, row_ptr, col, val, dtype=L.F64, C_=0, sigma=0):
        rp, cj, va = _as_csr(row_ptr, col, val)
        return cls(L.lib().spmv_hip_create_sell(rows, columns, len(cj), _ptr(rp), _ptr(cj), _ptr(va), dtype, C_,
                                                sigma))

    @classmethod
    def sell_dfc(cls, rows, columns, row_ptr, col, val, dtype=L.F64, C_=0, sigma=0):
        """SELL-C-sigma without padding (csr_to_sellcs_dfc, sellcs_format.c:53-136)."""
        rp, cj, va = _as_csr(row_ptr, col, val)
        return cls(L.lib().spmv_hip_create_sell_dfc(rows, columns, len(cj), _ptr(rp), _ptr(cj), _ptr(va), dtype, C_,
                                                    sigma))

    @classmethod
    def auto(cls, rows, columns, coo, dtype=L.F64, reps=3):
        """Fastest of merge / lane-per-row / vector CSR, SELLP, SELL-64-1024 by trial; sorts `coo` in place."""
        coo = _as_coo(coo)
        return cls(L.lib().spmv_hip_create_auto(rows, columns, len(coo), _ptr(coo), dtype, reps))

    @classmethod
    def auto_from_device_csr(cls, rows, columns, nnz, d_row_ptr, d_col, d_val, dtype=L.F64, reps=3, stream=0,
                             keepalive=None):
        m = cls(L.lib().spmv_hip_create_auto_from_device_csr(rows, columns, nnz, d_row_ptr, d_col, d_val, dtype, reps,
                                                             stream))
        m._keepalive = keepalive
        return m

    @classmethod
    def from_device_csr(cls, fmt, rows, columns, nnz, d_row_ptr, d_col, d_val, dtype=L.F64, p0=0, p1=0, stream=0,
                        keepalive=None):
        """CSR arrays already in HBM (raw device addresses).  `keepalive` is held
        for the life of the matrix (the CSR family adopts the arrays)."""
        m = cls(L.lib().spmv_hip_create_from_device_csr(fmt, rows, columns, nnz, d_row_ptr, d_col, d_val, dtype,
                                                        p0, p1, stream))
        m._keepalive = keepalive
        return m

    # ---- operations ----
    def mult(self, x: np.ndarray) -> np.ndarray:
        """mult_X(matrix, x, y) with host vectors (driver_spmv.c:276)."""
        info = self.info
        x = np.ascontiguousarray(x, np.float64)
        if x.shape != (info.columns,):
            raise ValueError(f"x has shape {x.shape}, matrix has {info.columns} columns")
        y = np.full(info.rows, np.nan)
        L.check(L.lib().spmv_hip_mult(self._h, _ptr(x), _ptr(y)))
        return y

    def set_row_map(self, d_row_map: int, keepalive=None) -> None:
        """Row k of this (CSR scalar/vector) matrix writes y[row_map[k]]; device address of `rows` ints."""
        L.check(L.lib().spmv_hip_set_row_map(self._h, d_row_map))
        self._row_map_keepalive = keepalive

    def mult_dev(self, d_x: int, d_y: int, stream: int = 0) -> None:
        L.check(L.lib().spmv_hip_mult_dev(self._h, d_x, d_y, stream))

    def mult_mv(self, x: np.ndarray) -> np.ndarray:
        """mult_mv_csr(csr, n, x, y) (spmv.h:54): x[n, columns] -> y[n, rows], one product per row of x."""
        info = self.info
        x = np.ascontiguousarray(x, np.float64)
        if x.ndim != 2 or x.shape[1] != info.columns:
            raise ValueError(f"x has shape {x.shape}, expected (n, {info.columns})")
        y = np.full((x.shape[0], info.rows), np.nan)
        L.check(L.lib().spmv_hip_mult_mv(self._h, x.shape[0], _ptr(x), _ptr(y)))
        return y

    def mult_mv_dev(self, n: int, d_x: int, ldx: int, d_y: int, ldy: int, stream: int = 0) -> None:
        L.check(L.lib().spmv_hip_mult_mv_dev(self._h, n, d_x, ldx, d_y, ldy, stream))

    def bench(self, x: np.ndarray, min_iters=10, min_seconds=2.0):
        """The driver's timed power loop (driver_spmv.c:292-308) -> (iters, seconds, GFLOP/s, x_last)."""
        x = np.ascontiguousarray(x, np.float64).copy()
        it, sec = C.c_int(), C.c_double()
        L.check(L.lib().spmv_hip_bench(self._h, min_iters, min_seconds, _ptr(x), C.byref(it), C.byref(sec)))
        gflops = 2.0 * self.info.nnz * it.value / sec.value / 1e9
        return it.value, sec.value, gflops, x

    def set_option(self, name: str, value: int) -> None:
        L.check(L.lib().spmv_hip_set_option(self._h, name.encode(), value))
        self._info = None

    def analyze(self, reps: int = 3) -> int:
        """Pick the fastest kernel variant by trial (sellcs_analyze_matrix, sell.c:30); returns it."""
        L.check(L.lib().spmv_hip_analyze(self._h, reps))
        self._info = None
        return self.info.variant

    @property
    def info(self) -> L.Info:
        if self._info is None:
            info = L.Info()
            L.check(L.lib().spmv_hip_get_info(self._h, C.byref(info)))
            self._info = info
        return self._info

    def download(self, which: int) -> np.ndarray:
        info = self.info
        vt = _NP[info.dtype]
        if which == L.ARR_VOP_LEN:                      # one byte per column; the count is the last vop pointer
            out = np.empty(int(self.download(L.ARR_VOP_PTR)[-1]), np.uint8)
            L.check(L.lib().spmv_hip_download(self._h, which, _ptr(out), out.nbytes))
            return out
        shape, dt = {
            L.ARR_VOP_PTR: (info.nslices + 1, np.int64),
            L.ARR_ROW_PTR: (info.rows + 1, np.int32),
            L.ARR_COL: (info.stored, np.int32),
            L.ARR_VAL: (info.stored, vt),
            L.ARR_SLICE_PTR: (info.nslices + 1, np.int64),
            L.ARR_ROW_ORDER: (info.rows, np.int32),
            L.ARR_TILE_COORD: (2 * (info.ntiles + 1), np.int32),
        }[which]
        out = np.empty(shape, dt)
        L.check(L.lib().spmv_hip_download(self._h, which, _ptr(out), out.nbytes))
        return out

    def free(self) -> None:
        if self._h:
            L.lib().spmv_hip_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _as_coo(coo) -> np.ndarray:
    if not (isinstance(coo, np.ndarray) and coo.dtype == COO_DTYPE and coo.flags.c_contiguous and coo.flags.writeable):
        raise TypeError("coo must be a writable C-contiguous array of COO_DTYPE (it is sorted in place)")
    return coo


def _as_csr(row_ptr, col, val):
    return (np.ascontiguousarray(row_ptr, np.int32), np.ascontiguousarray(col, np.int32),
            np.ascontiguousarray(val, np.float64))


def sort_coo(coo: np.ndarray) -> None:
    """sort_mtx(by_row_mtx) (mtx.c:9-34), in place."""
    coo = _as_coo(coo)
    L.lib().spmv_hip_sort_coo(len(coo), _ptr(coo))


def load_mtx(path: str):
    """load_mtx (mtx.c:36-137) -> (rows, columns, coo)."""
    return _load(L.lib().spmv_hip_load_mtx, path)


def load_bin(path: str):
    """load_bin (mtx.c:139-205) -> (rows, columns, coo)."""
    return _load(L.lib().spmv_hip_load_bin, path)


def save_bin(path: str, rows: int, columns: int, coo: np.ndarray) -> None:
    """save_bin (mtx.c:207-247)."""
    coo = np.ascontiguousarray(coo, dtype=COO_DTYPE)
    L.check(L.lib().spmv_hip_save_bin(path.encode(), rows, columns, len(coo), _ptr(coo)))


def _load(fn, path: str):
    r, c, n, p = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
    L.check(fn(path.encode(), C.byref(r), C.byref(c), C.byref(n), C.byref(p)))
    try:
        buf = (C.c_char * (16 * max(n.value, 1))).from_address(p.value)
        coo = np.frombuffer(buf, dtype=COO_DTYPE, count=n.value).copy()
    finally:
        L.lib().spmv_hip_free_coo(p)
    return r.value, c.value, coo


def coo_to_csr_host(rows, coo_sorted):
    """Plain CSR arrays of a (row, col)-sorted COO; host helper for the
    creators that take CSR arrays (the reference driver gets them from
    create_csr, driver_spmv.c:207,243)."""
    counts = np.bincount(coo_sorted["i"], minlength=rows).astype(np.int64)
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(counts)
    return rp, np.ascontiguousarray(coo_sorted["j"]), np.ascontiguousarray(coo_sorted["a"])


# reference MODE strings (driver_spmv.c:186-261) and the HIP additions
REFERENCE_BLOCK = 4  # _BLOCK of the AVX2 build (spmv.h:20-22)

MODES = ("coo", "csr_base", "csr_autovec", "csr_vec", "csr_highway", "csr_merge", "csr_bal", "sellp_vec", "sellp_autovec",
         "sellp_highway", "ell", "sell")


def create(mode: str, rows: int, columns: int, coo: np.ndarray, dtype=L.F64, sigma=0) -> Matrix:
    """MODE -> creator, as the reference driver's if/else chain does."""
    if mode == "coo":
        # :186-188: create_coo sorts by (row, col), mult_coo (coo.c:3-14) sums each row's entries in that
        # order from 0.0 and stores the sum -- the sums of mult_csr_base bit for bit, so the MODE runs on
        # the lane-per-row CSR kernel
        return Matrix.csr(rows, columns, coo, 1, dtype, L.CSR_SCALAR)
    if mode == "csr_base" or mode == "csr_autovec":          # :192-197
        return Matrix.csr(rows, columns, coo, 1, dtype, L.CSR_SCALAR)
    if mode in ("csr_vec", "csr_highway"):                   # :189-191,198-201 (rows padded to _BLOCK)
        return Matrix.csr(rows, columns, coo, REFERENCE_BLOCK, dtype, L.CSR_VECTOR)
    if mode == "csr_merge":                                  # :230-232
        return Matrix.csr_merge(rows, columns, coo, dtype)
    if mode == "csr_bal":                                    # :233-235
        return Matrix.csr_bal(rows, columns, coo, dtype)
    if mode == "ell":                                        # :236-238
        return Matrix.ell(rows, columns, coo, dtype)
    if mode in ("sellp_vec", "sellp_autovec", "sellp_highway", "sell", "sell_dfc"):
        # :206-221,239-247: create_csr(alignment 1) first, then the sliced format from its arrays
        sort_coo(coo)                                        # create_csr's side effect (csr.c:167)
        rp, cj, va = coo_to_csr_host(rows, coo)
        if mode == "sell":
            return Matrix.sell(rows, columns, rp, cj, va, dtype, 0, sigma)
        if mode == "sell_dfc":                               # no reference MODE: its DFC layout has no CPU kernel
            return Matrix.sell_dfc(rows, columns, rp, cj, va, dtype, 0, sigma)
        return Matrix.sellp(rows, columns, rp, cj, va, dtype)
    if mode == "auto":                                       # no reference MODE: format chosen by trial
        return Matrix.auto(rows, columns, coo, dtype)
    raise ValueError("Error: Unknown execution mode.")       # :258-260
