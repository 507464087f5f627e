#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libspmv_ref.so).

Run in the build container (needs /root/reference to build oracle/_ref):

    make -C oracle all && python tests/golden/make_golden.py

The reference ships no golden vectors or tests, so these fixtures are the pin
for the oracle and the HIP path: inputs (COO, x) and what the reference's own
creators and kernels produced for them.  Only data is stored -- no reference
source.  Build flags of the reference: oracle/Makefile (its Makefile:6,21,25
with SIMD=AVX2, NO_AUTOVEC=ON).

Known reference defects and how the fixtures treat them:
  * mult_sell reads row_order[] past nrows in the last partial slice
    (radix_sort.c:108-113) -> y_sell only stored when rows % 64 == 0.
  * mult_csr_bal is wrong with empty rows (csr_bal.c:29) and create_csr_bal
    reads mtx[0] of an empty chunk -> csr_bal only stored without empty rows
    and with nnz >= threads.
  * mult_sellp (AVX2) stores 4 doubles for the last block -> y has slack.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

import oracle  # noqa: E402
from tests import cases as CS  # noqa: E402


def run_case(name, rows, cols, coo_in):
    R = oracle.ref()
    out = {"rows": np.int64(rows), "cols": np.int64(cols)}
    out["coo_in_i"], out["coo_in_j"], out["coo_in_a"] = coo_in["i"].copy(), coo_in["j"].copy(), coo_in["a"].copy()
    nnz = len(coo_in)
    x = oracle.aligned_empty(cols)
    x[:] = CS.x_for(cols)
    out["x"] = x.copy()
    slack = 64

    # sort_mtx side effect (mtx.c:18-34) and the COO product (mtx.c:270-276)
    coo = np.ascontiguousarray(coo_in.copy())
    R.ref_sort_mtx(nnz, coo)
    out["coo_sorted_i"], out["coo_sorted_j"], out["coo_sorted_a"] = coo["i"].copy(), coo["j"].copy(), coo["a"].copy()
    y = oracle.aligned_empty(rows, fill=0.0)
    R.ref_mult_mtx(rows, cols, nnz, coo, x, y)
    out["y_coo"] = y.copy()

    # create_csr alignment 1 and _BLOCK (csr.c:160-203), three CSR kernels
    for align in (1, R.ref_block()):
        c = R.ref_create_csr(rows, cols, nnz, align, coo)
        rp = oracle.as_np(c.contents.i, rows + 1, np.int32)
        p = int(rp[-1])
        cj = oracle.as_np(c.contents.j, p, np.int32)
        va = oracle.as_np(c.contents.A, p, np.float64)
        out[f"csr{align}_row_ptr"], out[f"csr{align}_col"], out[f"csr{align}_val"] = rp, cj, va
        y = oracle.aligned_empty(rows + slack, fill=np.nan)
        if align == 1:
            R.ref_mult_csr_base(c, x, y); out["y_csr_base"] = y[:rows].copy()
            y[:] = np.nan
            R.ref_mult_csr_autovec(c, x, y); out["y_csr_autovec"] = y[:rows].copy()
        else:
            R.ref_mult_csr_vec(c, x, y); out["y_csr_vec"] = y[:rows].copy()
        if align == 1:
            csr1 = c
        else:
            R.ref_free_csr(c)
    rp1, cj1, va1 = out["csr1_row_ptr"], out["csr1_col"], out["csr1_val"]

    # merge path (csr_merge.c)
    for nt in CS.MERGE_THREADS:
        m = R.ref_create_csr_merge(nt, rows, cols, nnz, coo)
        y = oracle.aligned_empty(rows + slack, fill=np.nan)
        R.ref_mult_csr_merge(m, x, y)
        out[f"y_csr_merge_t{nt}"] = y[:rows].copy()
        out[f"merge_row_carry_t{nt}"] = oracle.as_np(m.contents.row_carry, nt, np.int32)
        out[f"merge_val_carry_t{nt}"] = oracle.as_np(m.contents.value_carry, nt, np.float64)
        # path coordinates of every thread boundary (csr_merge.c:25-30)
        items = rows + nnz
        per = (items + nt - 1) // nt
        coords = []
        row_end = np.ascontiguousarray(rp1[1:]) if rows else np.zeros(1, np.int32)
        for t in range(nt + 1):
            d = min(per * t, items)
            a, b = oracle.C.c_int(), oracle.C.c_int()
            R.ref_search_path_on_diag(d, row_end, rows, nnz, oracle.C.byref(a), oracle.C.byref(b))
            coords.append((a.value, b.value))
        out[f"merge_coords_t{nt}"] = np.array(coords, np.int32)

    # balanced CSR (csr_bal.c)
    if not CS.has_empty_rows(rows, coo):
        for nt in CS.BAL_THREADS:
            if nnz < nt:
                continue
            h = R.ref_create_csr_bal(nt, rows, cols, nnz, coo)
            y = oracle.aligned_empty(rows + slack, fill=np.nan)
            R.ref_mult_csr_bal(h, x, y)
            out[f"y_csr_bal_t{nt}"] = y[:rows].copy()
            part = []
            for t in range(nt):
                th = R.ref_csr_bal_thread(h, t).contents
                part.append((th.rows, th.first_row, th.last_row))
            out[f"bal_part_t{nt}"] = np.array(part, np.int32)

    # SELL-C-sigma (sell.c, sellcs_format.c, radix_sort.c), C = 64
    for sigma in CS.SELL_SIGMAS:
        s = R.ref_create_sell(csr1, 64, sigma)
        ns = R.ref_sell_nslices(s)
        sp = oracle.as_np(R.ref_sell_slice_pointers(s), ns + 1, np.int32)
        out[f"sell_s{sigma}_slice_ptr"] = sp.astype(np.int64)
        out[f"sell_s{sigma}_slice_widths"] = oracle.as_np(R.ref_sell_slice_widths(s), ns, np.int32)
        out[f"sell_s{sigma}_row_order"] = oracle.as_np(R.ref_sell_row_order(s), rows, np.int32)
        padded = int(sp[-1]) if ns else 0
        out[f"sell_s{sigma}_col"] = oracle.as_np(R.ref_sell_column_indices(s), padded, np.int32)
        out[f"sell_s{sigma}_val"] = oracle.as_np(R.ref_sell_values(s), padded, np.float64)
        if rows % 64 == 0:
            y = oracle.aligned_empty(rows + slack, fill=0.0)
            R.ref_mult_sell(s, x, y)
            out[f"y_sell_s{sigma}"] = y[:rows].copy()

    # SELLP, slice height _BLOCK = 4 (sellp.c)
    sp_ = R.ref_create_sellp(rows, cols, nnz, rp1, cj1 if nnz else np.zeros(1, np.int32),
                             va1 if nnz else np.zeros(1))
    nb = sp_.contents.num_blocks
    bp = oracle.as_np(sp_.contents.block_ptr, nb + 1, np.int32)
    out["sellp4_block_ptr"] = bp.astype(np.int64)
    out["sellp4_j"] = oracle.as_np(sp_.contents.j, int(bp[-1]), np.int32)
    out["sellp4_A"] = oracle.as_np(sp_.contents.A, int(bp[-1]), np.float64)
    y = oracle.aligned_empty(rows + slack, fill=np.nan)
    R.ref_mult_sellp(sp_, x, y); out["y_sellp_vec"] = y[:rows].copy()
    y[:] = np.nan
    R.ref_mult_sellp_autovec(sp_, x, y); out["y_sellp_autovec"] = y[:rows].copy()

    # ELL, slice height 4, int64 columns (ell.c)
    e = R.ref_create_ell(rows, cols, nnz, coo)
    nb = e.contents.num_blocks
    bp = oracle.as_np(e.contents.block_ptr, nb + 1, np.int32)
    out["ell4_block_ptr"] = bp.astype(np.int64)
    out["ell4_j"] = oracle.as_np(e.contents.j, int(bp[-1]), np.int64)
    out["ell4_A"] = oracle.as_np(e.contents.A, int(bp[-1]), np.float64)
    y = oracle.aligned_empty(rows + slack, fill=np.nan)
    R.ref_mult_ell(e, x, y); out["y_ell"] = y[:rows].copy()
    return out


def main():
    if not oracle.ref_available():
        sys.exit("oracle/_ref/libspmv_ref.so missing: run `make -C oracle all` where /root/reference exists")
    manifest = {}
    for name, (rows, cols, coo) in CS.all_cases().items():
        out = run_case(name, rows, cols, coo)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        manifest[name] = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
                          "rows": rows, "cols": cols, "nnz": int(len(coo)), "arrays": sorted(out)}
        print(f"{name}: rows {rows} nnz {len(coo)} -> {os.path.getsize(path)} bytes")
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
