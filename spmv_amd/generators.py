"""Synthetic matrices of the benchmark configurations, built as CSR with torch
ops so the same code runs on the CPU (tests) and in HBM (full sizes).

None of this exists in the reference (its matrices are external MatrixMarket
files, reference matrix_lists/*.txt); its two unused generators create_band /
create_arrow (reference src/mtx.c:278-352) are mirrored by band() / arrow().

Every generator returns (rows, columns, row_ptr[int32], col[int32], val[float64])
with columns ascending inside each row, i.e. the CSR that create_csr
(reference src/csr.c:160-203) builds from the (row, col)-sorted COO.
"""
from __future__ import annotations

import itertools

import numpy as np
import torch


def _finish(rows, cols, counts, col, val):
    row_ptr = torch.zeros(rows + 1, dtype=torch.int64, device=col.device)
    torch.cumsum(counts, 0, out=row_ptr[1:])
    assert int(row_ptr[-1]) < 2**31, "nnz exceeds 32-bit row pointers"
    return rows, cols, row_ptr.to(torch.int32), col.to(torch.int32), val


def laplacian2d(n: int, device="cpu"):
    """5-point Laplacian on an n x n grid: diagonal 4, neighbours -1.
    rows = n^2, nnz = 5n^2 - 4n."""
    return _stencil((n, n), [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)], 4.0, -1.0, 0, n * n, device)


def stencil27(nx: int, ny: int | None = None, nz: int | None = None, device="cpu", row_begin=0, row_end=None,
              chunk_rows=1 << 22):
    """27-point stencil on an nx x ny x nz grid (x fastest): centre 26,
    neighbours -1, truncated at the boundary.  nnz = (3nx-2)(3ny-2)(3nz-2).
    [row_begin, row_end) selects a row block (column indices stay global)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    offs = [(dz, dy, dx) for dz, dy, dx in itertools.product((-1, 0, 1), repeat=3)]
    total = nx * ny * nz
    return _stencil((nz, ny, nx), offs, 26.0, -1.0, row_begin, total if row_end is None else row_end, device,
                    chunk_rows)


def _stencil(dims, offsets, centre, off, row_begin, row_end, device, chunk_rows=1 << 22):
    """Generic box stencil; dims slowest..fastest, offsets in the same order and
    lexicographically ascending (= ascending column)."""
    nd = len(dims)
    total = int(np.prod(dims))
    strides = [int(np.prod(dims[k + 1:])) for k in range(nd)]
    offs = torch.tensor(offsets, dtype=torch.int64, device=device)             # [K, nd]
    delta = (offs * torch.tensor(strides, dtype=torch.int64, device=device)).sum(1)  # [K]
    is_centre = (offs == 0).all(1)
    vals_k = torch.where(is_centre, torch.tensor(centre, dtype=torch.float64, device=device),
                         torch.tensor(off, dtype=torch.float64, device=device))
    counts_l, col_l, val_l = [], [], []
    for r0 in range(row_begin, row_end, chunk_rows):
        r1 = min(r0 + chunk_rows, row_end)
        r = torch.arange(r0, r1, dtype=torch.int64, device=device)
        valid = torch.ones((r1 - r0, len(offsets)), dtype=torch.bool, device=device)
        rem = r
        for k in range(nd):
            coord = rem // strides[k]
            rem = rem - coord * strides[k]
            c = coord[:, None] + offs[None, :, k]
            valid &= (c >= 0) & (c < dims[k])
        cols = r[:, None] + delta[None, :]
        counts_l.append(valid.sum(1))
        col_l.append(cols[valid])
        val_l.append(vals_k.expand(r1 - r0, -1)[valid])
    rows = row_end - row_begin
    return _finish(rows, total, torch.cat(counts_l), torch.cat(col_l), torch.cat(val_l))


def band(n: int, width: int, seed=1, device="cpu"):
    """Band matrix, row i holds columns [i-width+1, i+width) (reference create_band, mtx.c:278-305);
    values uniform(0,1) from a seeded generator instead of rand()."""
    i = torch.arange(n, dtype=torch.int64, device=device)
    lo = (i - width + 1).clamp(min=0)
    hi = (i + width).clamp(max=n)
    counts = hi - lo
    row = torch.repeat_interleave(i, counts)
    start = torch.cumsum(counts, 0) - counts
    col = torch.arange(int(counts.sum()), dtype=torch.int64, device=device) - torch.repeat_interleave(start, counts) \
        + torch.repeat_interleave(lo, counts)
    g = torch.Generator(device="cpu").manual_seed(seed)
    val = torch.rand(col.numel(), dtype=torch.float64, generator=g).to(device)
    del row
    return _finish(n, n, counts, col, val)


def arrow(n: int, width: int, seed=2, device="cpu"):
    """Arrow matrix: `width` dense rows and columns plus a band (reference create_arrow, mtx.c:307-352)."""
    rows_l, cols_l = [], []
    i = np.arange(n)
    for r in range(min(width, n)):
        rows_l.append(np.full(n, r)); cols_l.append(np.arange(n))
    for r in range(width, n):
        head = np.arange(min(width, n))
        lo = max(r - width + 1, width); hi = min(r + width, n)
        tail = np.arange(lo, hi)
        c = np.concatenate([head, tail])
        rows_l.append(np.full(len(c), r)); cols_l.append(c)
    del i
    row = np.concatenate(rows_l); col = np.concatenate(cols_l)
    rng = np.random.default_rng(seed)
    val = rng.random(len(col))
    counts = torch.from_numpy(np.bincount(row, minlength=n)).to(device)
    return _finish(n, n, counts, torch.from_numpy(col).to(device), torch.from_numpy(val).to(device))


def _splitmix64(z: torch.Tensor) -> torch.Tensor:
    """Counter-based 64-bit mixer on int64 tensors (wrapping arithmetic)."""
    def c(v):  # python int -> wrapped int64 constant
        return v - (1 << 64) if v >= (1 << 63) else v
    z = z + c(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * c(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def _lsr(z: torch.Tensor, k: int) -> torch.Tensor:
    """Logical shift right of int64 (torch's >> is arithmetic)."""
    return (z >> k) & ((1 << (64 - k)) - 1)


def _u01(h: torch.Tensor) -> torch.Tensor:
    """Top 53 bits of a 64-bit hash -> double in [0, 1)."""
    return _lsr(h, 11).to(torch.float64) * (1.0 / (1 << 53))


def rmat(scale: int, edge_factor: int = 16, a=0.57, b=0.19, c=0.19, seed=42, device="cpu", chunk_edges=1 << 26):
    """R-MAT (Kronecker) graph as a sparse matrix: 2^scale rows, edge_factor * 2^scale
    generated edges, quadrant probabilities (a, b, c, 1-a-b-c), one counter-based
    random number per (edge, level) so any device reproduces the same edges.
    Duplicate edges are dropped; the value of (i, j) is uniform(0,1) from hash(i, j).
    Empty rows are kept."""
    n = 1 << scale
    nedges = edge_factor * n
    keys_l = []
    ab, abc = a + b, a + b + c
    for e0 in range(0, nedges, chunk_edges):
        e1 = min(e0 + chunk_edges, nedges)
        e = torch.arange(e0, e1, dtype=torch.int64, device=device)
        row = torch.zeros_like(e); col = torch.zeros_like(e)
        for level in range(scale):
            u = _u01(_splitmix64(e * scale + level + (seed << 40)))
            right = ((u >= a) & (u < ab)) | (u >= abc)      # quadrants b, d -> column bit
            down = u >= ab                                   # quadrants c, d -> row bit
            row = (row << 1) | down.to(torch.int64)
            col = (col << 1) | right.to(torch.int64)
        keys_l.append((row << 32) | col)
    keys = torch.unique(torch.cat(keys_l))                   # sorted by (row, col), duplicates dropped
    del keys_l
    row = keys >> 32
    col = keys & 0xFFFFFFFF
    val = _u01(_splitmix64(keys))
    counts = torch.bincount(row, minlength=n)
    return _finish(n, n, counts, col, val)


def power_law_rows(n: int, seed=3, max_len=None, empty_every=7, device="cpu"):
    """Small skewed matrix for tests: row lengths ~ Zipf, every `empty_every`-th
    row empty, one dense row; columns a strided pattern, values hash-uniform."""
    rng = np.random.default_rng(seed)
    max_len = n if max_len is None else min(max_len, n)
    lens = np.minimum(rng.zipf(1.6, n), max_len).astype(np.int64)
    lens[::empty_every] = 0
    lens[n // 2] = max_len
    rows_l, cols_l = [], []
    for r in range(n):
        if lens[r] == 0:
            continue
        c = np.sort(rng.choice(n, size=lens[r], replace=False))
        rows_l.append(np.full(lens[r], r)); cols_l.append(c)
    row = np.concatenate(rows_l); col = np.concatenate(cols_l)
    val = rng.standard_normal(len(col))
    counts = torch.from_numpy(np.bincount(row, minlength=n)).to(device)
    return _finish(n, n, counts, torch.from_numpy(col).to(device), torch.from_numpy(val).to(device))


def csr_to_coo_numpy(rows, row_ptr, col, val):
    """(rows, CSR tensors) -> structured COO array in (row, col) order (reference struct mtx, spmv.h:26)."""
    from .matrix import COO_DTYPE
    rp = row_ptr.cpu().numpy().astype(np.int64)
    coo = np.zeros(int(rp[-1]), dtype=COO_DTYPE)
    coo["i"] = np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp))
    coo["j"] = col.cpu().numpy()
    coo["a"] = val.cpu().numpy()
    return coo


def write_mtx(path: str, rows: int, cols: int, coo: np.ndarray, symmetry="general", field="real") -> None:
    """MatrixMarket coordinate writer (1-based), for driver runs and loader tests."""
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {symmetry}\n% written by spmv_amd.generators\n")
        f.write(f"{rows} {cols} {len(coo)}\n")
        if field == "pattern":
            for e in coo:
                f.write(f"{e['i'] + 1} {e['j'] + 1}\n")
        else:
            for e in coo:
                f.write(f"{e['i'] + 1} {e['j'] + 1} {float(e['a'])!r}\n")
