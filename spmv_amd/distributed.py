"""Row-block SpMV over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no partitioned product; its only precedent is the equal-nnz
contiguous chunking of create_csr_bal (reference src/csr_bal.c:83-85).  Here
rank p owns a contiguous block of rows of a square matrix and the matching block
of x and y.  The block itself -- interior rows in the caller's format, boundary
rows as a row-mapped compact CSR, the list of remote columns -- is built in HBM by
the library (spmv_hip_rowblock_*, include/spmv_hip.h); this module adds what one
process per GPU needs on top: who sends what to whom, and the exchange.  One
product is

    1. pack the x entries other ranks need (spmv_hip_rowblock_pack) and post the
       peer-to-peer exchange -- one RCCL group of sends and receives, direct copies
       over xGMI with every link busy, not a ring -- on a side stream;
    2. meanwhile multiply the INTERIOR rows with the local x block -- the bulk of
       the non-zeros;
    3. when the halo has landed, multiply the BOUNDARY rows against [x_local | halo].

`exchange="allgather"` is the plain alternative: all-gather the whole x, then one
product with global column indices (no overlap; every rank receives all of x).

The local work goes through an `engine`; the default engine is the HIP library.
The CPU tests inject an engine built on the oracle to exercise the bookkeeping and
the exchange with gloo.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L


class HipEngine:
    """Local products on libspmv_hip with tensors resident in HBM.  `exact_order`: the kernel trials may
    only pick variants that leave y's bits as the default kernels give them (set before the first trial,
    so the bits do not depend on what a trial measures)."""

    def __init__(self, dtype, exact_order=False):
        self.dtype = dtype
        self.exact_order = exact_order
        self.tdtype = torch.float64 if dtype == L.F64 else torch.float32

    def _arrays(self, rp, cj, va):
        va = va.to(self.tdtype)
        if cj.numel() == 0:          # the library wants valid pointers
            cj = torch.zeros(1, dtype=torch.int32, device=rp.device)
            va = torch.zeros(1, dtype=self.tdtype, device=rp.device)
        return rp.contiguous(), cj.contiguous(), va.contiguous()

    def matrix(self, fmt, rows, cols, rp, cj, va, p0=0, p1=0):
        from .matrix import Matrix
        nnz = int(rp[-1])
        rp, cj, va = self._arrays(rp, cj, va)
        st = torch.cuda.current_stream().cuda_stream
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), self.dtype,
                                   p0, p1, st, keepalive=(rp, cj, va))
        if self.exact_order:
            m.set_option("exact_order", 1)
        m.analyze(3)
        return _HipLocal(m)

    def rowblock(self, fmt, rows, row_begin, cols_global, rp, cj, va, p0=0, p1=0):
        nnz = int(rp[-1])
        rp, cj, va = self._arrays(rp, cj, va)
        st = torch.cuda.current_stream().cuda_stream
        h = L.check_handle(L.lib().spmv_hip_rowblock_create(fmt, rows, row_begin, cols_global, nnz, rp.data_ptr(),
                                                            cj.data_ptr(), va.data_ptr(), self.dtype, p0, p1, st))
        blk = _HipRowBlock(h)           # the library has copied what it keeps: rp, cj, va may go
        if self.exact_order:
            L.check(L.lib().spmv_hip_rowblock_set_option(h, b"exact_order", 1))
        blk.analyze(3)                  # a new block runs its default kernels until a trial has picked others
        return blk


class _HipLocal:
    def __init__(self, m):
        self.m = m
        self.stored = m.info.stored
        self.variant = m.info.variant

    def mult(self, x, y):
        self.m.mult_dev(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)


class _HipRowBlock:
    """One spmv_hip_rowblock handle."""

    def __init__(self, h):
        self._h = h
        info = L.RowBlockInfo()
        L.check(L.lib().spmv_hip_rowblock_get_info(h, C.byref(info)))
        self.info = info
        self.stored = info.interior_stored + info.boundary_nnz
        self.variant = info.interior_variant
        halo = np.empty(max(info.halo, 1), np.int32)
        L.check(L.lib().spmv_hip_rowblock_halo_columns(h, halo.ctypes.data_as(C.c_void_p)))
        self.halo_columns = torch.from_numpy(halo[:info.halo].astype(np.int64))

    def set_option(self, name, value):
        """Sets the option and repeats the kernel trial under it."""
        L.check(L.lib().spmv_hip_rowblock_set_option(self._h, name.encode(), value))
        self.analyze(3)

    def analyze(self, reps=3):
        L.check(L.lib().spmv_hip_rowblock_analyze(self._h, reps))
        info = L.RowBlockInfo()
        L.check(L.lib().spmv_hip_rowblock_get_info(self._h, C.byref(info)))
        self.info, self.variant = info, info.interior_variant

    def set_send_list(self, index):
        idx = np.ascontiguousarray(index.cpu().numpy(), np.int32)
        L.check(L.lib().spmv_hip_rowblock_set_send_list(self._h, len(idx), idx.ctypes.data_as(C.c_void_p)))

    def pack(self, x, send):
        L.check(L.lib().spmv_hip_rowblock_pack(self._h, x.data_ptr(), send.data_ptr(), torch.cuda.current_stream().cuda_stream))

    def mult_interior(self, x, y):
        L.check(L.lib().spmv_hip_rowblock_mult_interior(self._h, x.data_ptr(), y.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream))

    def mult_boundary(self, x_ext, y):
        L.check(L.lib().spmv_hip_rowblock_mult_boundary(self._h, x_ext.data_ptr(), y.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream))

    def free(self):
        if self._h:
            L.lib().spmv_hip_rowblock_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def row_blocks_by_nnz(row_ptr, world: int, snap: int = 1024):
    """Contiguous row blocks with about nnz/world non-zeros each: block p ends at the first row whose
    row_ptr reaches (p+1)*nnz/world (the equal-share rule of create_csr_bal, reference
    src/csr_bal.c:83-85, on rows instead of entries), moved to the nearest multiple of `snap` rows so
    that SELL sorting windows and slices never straddle two GPUs.  Returns world+1 boundaries.
    (spmv_hip_partition_rows is the library's form of the same rule for host row pointers.)"""
    rows = int(row_ptr.numel()) - 1
    nnz = int(row_ptr[-1])
    targets = torch.tensor([nnz * (p + 1) // world for p in range(world - 1)], dtype=row_ptr.dtype, device=row_ptr.device)
    cuts = torch.searchsorted(row_ptr.contiguous(), targets).cpu().tolist()
    bounds = [0]
    for c in cuts:
        c = (int(c) + snap // 2) // snap * snap if snap > 1 else int(c)
        bounds.append(min(max(c, bounds[-1]), rows))
    bounds.append(rows)
    return bounds


def slice_row_block(row_ptr, col, val, r0: int, r1: int):
    """CSR of rows [r0, r1) of a global CSR (column indices stay global)."""
    a, b = int(row_ptr[r0]), int(row_ptr[r1])
    return (row_ptr[r0:r1 + 1] - row_ptr[r0]).to(torch.int32), col[a:b].clone(), val[a:b].clone()


class RowBlockSpmv:
    """y_local = A[row block] * x for a square matrix partitioned by contiguous row blocks."""

    def __init__(self, fmt, rows, cols_global, rp, cj, va, dtype=L.F64, p0=0, p1=0, row_begin=0, exchange="halo",
                 device=None, engine=None, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows, self.cols_global, self.row_begin = rows, cols_global, row_begin
        self.exchange = exchange
        self.device = device if device is not None else rp.device
        self.on_gpu = torch.device(self.device).type == "cuda"
        self.engine = engine if engine is not None else HipEngine(dtype)
        self.tdtype = torch.float64 if dtype == L.F64 else torch.float32
        dev = self.device

        # row ranges of every rank
        # gloo cannot move device tensors: stage through the host (debug / single-GPU rehearsal only)
        self.stage_host = self.world > 1 and self.on_gpu and dist.get_backend(group) == "gloo"
        self.comm_device = torch.device("cpu") if (self.stage_host or not self.on_gpu) else torch.device(dev)
        mine = torch.tensor([row_begin, rows], dtype=torch.int64, device=self.comm_device)
        allr = self._all_gather(mine)
        begins = torch.stack([t[0] for t in allr]).cpu()
        counts_r = torch.stack([t[1] for t in allr]).cpu()
        if not bool(((begins[1:] == begins[:-1] + counts_r[:-1]).all())) or int(begins[0]) != 0 \
                or int(begins[-1] + counts_r[-1]) != cols_global:
            raise ValueError("row blocks must be contiguous, in rank order, and cover the square matrix")
        self.begins = begins.tolist() + [cols_global]

        self.A_glob = self.block = None
        self.nhalo = 0
        if exchange == "allgather" or self.world == 1:
            if self.world > 1 and len(set(counts_r.tolist())) != 1:
                raise ValueError("allgather exchange needs equal row blocks")
            self.A_glob = self.engine.matrix(fmt, rows, cols_global, rp, cj, va, p0, p1)
            self.x_full = torch.zeros(cols_global, dtype=self.tdtype, device=dev)
            self.x_local = self.x_full[row_begin:row_begin + rows] if self.world == 1 \
                else torch.zeros(rows, dtype=self.tdtype, device=dev)
            self.stored, self.variant = self.A_glob.stored, self.A_glob.variant
            return

        # ---- halo exchange set-up ------------------------------------------------
        # the block (interior matrix, boundary rows, sorted list of remote columns) comes from the engine
        self.block = self.engine.rowblock(fmt, rows, row_begin, cols_global, rp, cj, va, p0, p1)
        halo = self.block.halo_columns.to(self.comm_device)       # sorted global column ids this rank needs
        self.nhalo = int(halo.numel())
        ends = torch.tensor(self.begins[1:], dtype=torch.int64, device=self.comm_device)
        owner = torch.searchsorted(ends, halo, right=True)       # rank that owns each halo column
        need = torch.bincount(owner, minlength=self.world)       # how many I need from each rank
        need_all = self._all_gather(need)
        give = torch.stack(need_all)[:, self.rank].cpu().tolist()  # how many each rank needs from me
        need_l = need.cpu().tolist()
        # tell every owner which of its entries I need (as indices local to the owner)
        need_off = [0]
        for q in range(self.world):
            need_off.append(need_off[-1] + need_l[q])
        sends = {q: (halo[need_off[q]:need_off[q + 1]] - self.begins[q]).contiguous()
                 for q in range(self.world) if q != self.rank and need_l[q]}
        recv_lists = {q: torch.empty(give[q], dtype=torch.int64, device=self.comm_device)
                      for q in range(self.world) if q != self.rank and give[q]}
        self._sendrecv(sends, recv_lists, staged=False)
        if self.on_gpu:
            torch.cuda.synchronize()
        self.peers_send = [q for q in range(self.world) if q != self.rank and give[q]]
        self.peers_recv = [q for q in range(self.world) if q != self.rank and need_l[q]]
        self.send_counts = {q: give[q] for q in self.peers_send}
        self.recv_ranges = {q: (need_off[q], need_off[q + 1]) for q in self.peers_recv}
        send_index = torch.cat([recv_lists[q] for q in self.peers_send]) if self.peers_send else \
            torch.empty(0, dtype=torch.int64)
        self.block.set_send_list(send_index)
        self.nsend = int(send_index.numel())
        self.send_buf = torch.empty(max(self.nsend, 1), dtype=self.tdtype, device=dev)[:self.nsend]

        self.x_ext = torch.zeros(rows + self.nhalo, dtype=self.tdtype, device=dev)
        self.x_local = self.x_ext[:rows]
        self.halo_buf = self.x_ext[rows:]
        self.comm_stream = torch.cuda.Stream(device=dev) if self.on_gpu else None
        self.stored, self.variant = self.block.stored, self.block.variant

    # ------------------------------------------------------------------------------
    def _all_gather(self, t):
        if self.world == 1:
            return [t]
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def _sendrecv(self, sends, recvs, staged=None):
        """Post every send and receive of one exchange together (one NCCL group: direct peer copies).
        `staged`: the tensors live on the GPU while the backend is gloo -- go through the host."""
        staged = self.stage_host if staged is None else staged
        if staged:
            host_s = {q: t.cpu() for q, t in sends.items()}
            host_r = {q: torch.empty(t.shape, dtype=t.dtype) for q, t in recvs.items()}
        else:
            host_s, host_r = sends, recvs
        ops = [dist.P2POp(dist.isend, t, q, group=self.group) for q, t in host_s.items()]
        ops += [dist.P2POp(dist.irecv, t, q, group=self.group) for q, t in host_r.items()]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if staged:
            for q, t in recvs.items():
                t.copy_(host_r[q])

    def _exchange_halo(self):
        if not self.stage_host and self.world > 1:
            # the buffers never move: build the group of sends and receives once, post it every product
            ops = getattr(self, "_halo_ops", None)
            if ops is None:
                ops, off = [], 0
                for q in self.peers_send:
                    n = self.send_counts[q]
                    ops.append(dist.P2POp(dist.isend, self.send_buf[off:off + n], q, group=self.group))
                    off += n
                for q in self.peers_recv:
                    a, b = self.recv_ranges[q]
                    ops.append(dist.P2POp(dist.irecv, self.halo_buf[a:b], q, group=self.group))
                self._halo_ops = ops
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return
        sends, off = {}, 0
        for q in self.peers_send:
            n = self.send_counts[q]
            sends[q] = self.send_buf[off:off + n]
            off += n
        recvs = {q: self.halo_buf[self.recv_ranges[q][0]:self.recv_ranges[q][1]] for q in self.peers_recv}
        self._sendrecv(sends, recvs)

    def mult(self, x, y):
        """y[0:rows] = (A x)[my rows]; x is this rank's block (use `self.x_local` to avoid a copy)."""
        if self.A_glob is not None:
            if self.world == 1:
                if x.data_ptr() != self.x_local.data_ptr():
                    self.x_local.copy_(x)
            elif self.stage_host:
                full = torch.empty(self.cols_global, dtype=self.tdtype)
                dist.all_gather_into_tensor(full, x.contiguous().cpu(), group=self.group)
                self.x_full.copy_(full)
            else:
                dist.all_gather_into_tensor(self.x_full, x.contiguous(), group=self.group)
            self.A_glob.mult(self.x_full, y)
            return y
        if x.data_ptr() != self.x_local.data_ptr():
            self.x_local.copy_(x)
        if self.on_gpu:
            # order of issue: the interior product first, so that the GPU is busy while the host posts the
            # packing and the exchange on the side stream (which only waits for x, not for that product)
            main = torch.cuda.current_stream()
            self.comm_stream.wait_stream(main)
            self.block.mult_interior(self.x_local, y)
            with torch.cuda.stream(self.comm_stream):
                if self.nsend:
                    self.block.pack(self.x_local, self.send_buf)
                self._exchange_halo()
            main.wait_stream(self.comm_stream)
        else:
            if self.nsend:
                self.block.pack(self.x_local, self.send_buf)
            self._exchange_halo()
            self.block.mult_interior(self.x_local, y)
        self.block.mult_boundary(self.x_ext, y)
        return y

    def exchange_only(self):
        """The x exchange of one product without the local products (for timing the communication alone)."""
        if self.A_glob is not None:
            if self.world > 1 and not self.stage_host:
                dist.all_gather_into_tensor(self.x_full, self.x_local.contiguous(), group=self.group)
            return
        if self.nsend:
            self.block.pack(self.x_local, self.send_buf)
        self._exchange_halo()

    def compute_only(self, y):
        """The local products of one product against whatever x / halo values are in place (no exchange)."""
        if self.A_glob is not None:
            self.A_glob.mult(self.x_full, y)
            return y
        self.block.mult_interior(self.x_local, y)
        self.block.mult_boundary(self.x_ext, y)
        return y

    def norm2(self, v):
        """Global ||v||_2 of a row-partitioned vector (the driver's vec_norm, reference driver_spmv.c:21-26)."""
        s = torch.sum(v.to(torch.float64) ** 2).reshape(1)
        if self.world > 1:
            if self.stage_host:
                h = s.cpu()
                dist.all_reduce(h, group=self.group)
                s = h
            else:
                dist.all_reduce(s, group=self.group)
        return float(torch.sqrt(s))
