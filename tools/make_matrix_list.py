#!/usr/bin/env python3
"""Generate the matrices behind a list file of matrix_lists/ as `.bin` files (the reference's
save_bin layout, reference src/mtx.c:207-247: int rows, int columns, long nnz, then {int i, int j,
double a} entries) so that the C driver / spmv_hip.run work out of the box.

    python tools/make_matrix_list.py synthetic_5              # the default list of spmv_hip.run
    python tools/make_matrix_list.py standins_160 --scale 0.05

The reference's lists name SuiteSparse files that are not in its repository (reference
matrix_lists/matrix_list_160.txt:1-159, .gitignore:44); `standins_160` writes the generated stand-in
of every entry of that list (spmv_amd/standins.py) under the entry's own name, `synthetic_5` five
small matrices of the benchmark families (2-D Laplacian, 27-point stencil, band, R-MAT, skewed rows).
The list files hold paths relative to the repository root; the matrices go to matrices/<list>/
(not tracked).  Host-side tool: no GPU needed.
"""
import argparse
import os
import struct
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from spmv_amd import generators as G  # noqa: E402
from spmv_amd import standins  # noqa: E402


def write_bin(path, rows, cols, coo):
    with open(path, "wb") as f:
        f.write(struct.pack("iiq", rows, cols, len(coo)))
        f.write(coo.tobytes())


SYNTHETIC_5 = (
    ("laplacian2d_300", lambda: G.laplacian2d(300)),
    ("stencil27_40", lambda: G.stencil27(40)),
    ("band_20000_9", lambda: G.band(20000, 9)),
    ("rmat_s14", lambda: G.rmat(14, 16)),
    ("powerlaw_20000", lambda: G.power_law_rows(20000, max_len=3000)),
)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["synthetic_5", "standins_160"])
    ap.add_argument("--scale", type=float, default=0.05, help="standins_160: rows and nnz of every stand-in times this")
    ap.add_argument("--only", default="", help="standins_160: comma list of names")
    ap.add_argument("--device", default="cpu", help="where the generators run (cpu, cuda:0); the files are the same")
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "matrices", args.which)
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "matrix_lists"), exist_ok=True)
    if args.which == "synthetic_5":
        entries = [(name, make) for name, make in SYNTHETIC_5]
    else:
        names = [n for n in standins.names() if not args.only or n in args.only.split(",")]
        entries = [(n, (lambda n=n: standins.standin(n, args.scale, device=args.device))) for n in names]
    lines = []
    for name, make in entries:
        rows, cols, rp, cj, va = make()
        coo = G.csr_to_coo_numpy(rows, rp, cj, va)
        path = os.path.join(out_dir, name + ".bin")
        write_bin(path, rows, cols, coo)
        lines.append(os.path.relpath(path, ROOT))
        print(f"{lines[-1]}: {rows} rows, {len(coo)} nnz")
    list_path = os.path.join(ROOT, "matrix_lists", args.which + ".txt")
    with open(list_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"wrote {os.path.relpath(list_path, ROOT)} ({len(lines)} matrices)")


if __name__ == "__main__":
    main()
