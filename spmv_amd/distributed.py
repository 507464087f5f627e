"""Row-block SpMV over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no partitioned product; its only precedent is the equal-nnz
contiguous chunking of create_csr_bal (reference src/csr_bal.c:83-85).  Here
rank p owns a contiguous block of rows of a square matrix and the matching block
of x and y.  One product is

    1. pack the x entries other ranks need and post the peer-to-peer exchange
       (direct sends over xGMI, every link busy -- not a ring) on a side stream;
    2. meanwhile multiply the INTERIOR rows (rows whose columns are all local)
       with the local x block -- the bulk of the non-zeros;
    3. when the halo has landed, multiply the BOUNDARY rows (compact CSR with a
       row map) against [x_local | halo].

`exchange="allgather"` is the plain alternative: all-gather the whole x, then one
product with global column indices (no overlap; every rank receives all of x).

The local products go through an `engine`; the default engine is the HIP library.
The CPU tests inject an engine built on the oracle to exercise the partitioning
and the exchange with gloo.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib as L


class HipEngine:
    """Local products on libspmv_hip with tensors resident in HBM."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.tdtype = torch.float64 if dtype == L.F64 else torch.float32

    def matrix(self, fmt, rows, cols, rp, cj, va, p0=0, p1=0, row_map=None):
        from .matrix import Matrix
        nnz = int(rp[-1])
        va = va.to(self.tdtype)
        if cj.numel() == 0:          # the library wants valid pointers
            cj = torch.zeros(1, dtype=torch.int32, device=rp.device)
            va = torch.zeros(1, dtype=self.tdtype, device=rp.device)
        st = torch.cuda.current_stream().cuda_stream
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), self.dtype,
                                   p0, p1, st, keepalive=(rp, cj, va))
        if row_map is not None:
            m.set_row_map(row_map.data_ptr(), keepalive=row_map)
        else:
            m.analyze(3)
        return _HipLocal(m)


class _HipLocal:
    def __init__(self, m):
        self.m = m
        self.stored = m.info.stored
        self.variant = m.info.variant

    def mult(self, x, y):
        self.m.mult_dev(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)


def row_blocks_by_nnz(row_ptr, world: int, snap: int = 1024):
    """Contiguous row blocks with about nnz/world non-zeros each: block p ends at the first row whose
    row_ptr reaches (p+1)*nnz/world (the equal-share rule of create_csr_bal, reference
    src/csr_bal.c:83-85, on rows instead of entries), moved to the nearest multiple of `snap` rows so
    that SELL sorting windows and slices never straddle two GPUs.  Returns world+1 boundaries."""
    rows = int(row_ptr.numel()) - 1
    nnz = int(row_ptr[-1])
    targets = torch.tensor([nnz * (p + 1) // world for p in range(world - 1)], dtype=row_ptr.dtype, device=row_ptr.device)
    cuts = torch.searchsorted(row_ptr.contiguous(), targets).cpu().tolist()
    bounds = [0]
    for c in cuts:
        c = int(round(c / snap)) * snap if snap > 1 else int(c)
        bounds.append(min(max(c, bounds[-1]), rows))
    bounds.append(rows)
    return bounds


def slice_row_block(row_ptr, col, val, r0: int, r1: int):
    """CSR of rows [r0, r1) of a global CSR (column indices stay global)."""
    a, b = int(row_ptr[r0]), int(row_ptr[r1])
    return (row_ptr[r0:r1 + 1] - row_ptr[r0]).to(torch.int32), col[a:b].clone(), val[a:b].clone()


def _csr_select(rows, counts, keep_entry, cj, va):
    """CSR of the entries flagged in keep_entry (row structure given by per-row counts)."""
    row_of = torch.repeat_interleave(torch.arange(rows, device=cj.device), counts)
    new_counts = torch.zeros(rows, dtype=torch.int64, device=cj.device)
    new_counts.index_add_(0, row_of[keep_entry], torch.ones(int(keep_entry.sum()), dtype=torch.int64, device=cj.device))
    rp = torch.zeros(rows + 1, dtype=torch.int64, device=cj.device)
    torch.cumsum(new_counts, 0, out=rp[1:])
    return rp.to(torch.int32), cj[keep_entry], va[keep_entry]


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
        mine = torch.tensor([row_begin, rows], dtype=torch.int64, device=dev)
        allr = self._all_gather(mine)
        begins = torch.stack([t[0] for t in allr]).cpu()
        counts_r = torch.stack([t[1] for t in allr]).cpu()
        if not bool(((begins[1:] == begins[:-1] + counts_r[:-1]).all())) or int(begins[0]) != 0 \
                or int(begins[-1] + counts_r[-1]) != cols_global:
            raise ValueError("row blocks must be contiguous, in rank order, and cover the square matrix")
        self.begins = begins.tolist() + [cols_global]
        my_begin, my_end = row_begin, row_begin + rows

        counts = (rp[1:] - rp[:-1]).to(torch.int64)
        cj64 = cj.to(torch.int64)
        self.A_glob = self.A_int = self.A_bnd = None
        self.nhalo = 0
        if exchange == "allgather" or self.world == 1:
            if self.world > 1 and len(set(counts_r.tolist())) != 1:
                raise ValueError("allgather exchange needs equal row blocks")
            self.A_glob = self.engine.matrix(fmt, rows, cols_global, rp, cj, va, p0, p1)
            self.x_full = torch.zeros(cols_global, dtype=self.tdtype, device=dev)
            self.x_local = self.x_full[my_begin:my_end] if self.world == 1 else torch.zeros(rows, dtype=self.tdtype, device=dev)
            self.stored, self.variant = self.A_glob.stored, self.A_glob.variant
            return

        # ---- halo exchange set-up ------------------------------------------------
        local = (cj64 >= my_begin) & (cj64 < my_end)
        row_of = torch.repeat_interleave(torch.arange(rows, device=dev), counts)
        bnd = torch.zeros(rows, dtype=torch.bool, device=dev)
        bnd[row_of[~local]] = True
        halo = torch.unique(cj64[~local])                       # sorted global column ids this rank needs
        self.nhalo = int(halo.numel())
        ends = torch.tensor(self.begins[1:], dtype=torch.int64, device=dev)
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
        recv_lists = {q: torch.empty(give[q], dtype=torch.int64, device=dev)
                      for q in range(self.world) if q != self.rank and give[q]}
        self._sendrecv(sends, recv_lists)
        if self.on_gpu:
            torch.cuda.synchronize()
        self.peers_send = [q for q in range(self.world) if q != self.rank and give[q]]
        self.peers_recv = [q for q in range(self.world) if q != self.rank and need_l[q]]
        self.send_counts = {q: give[q] for q in self.peers_send}
        self.recv_ranges = {q: (need_off[q], need_off[q + 1]) for q in self.peers_recv}
        self.send_index = torch.cat([recv_lists[q] for q in self.peers_send]) if self.peers_send else \
            torch.empty(0, dtype=torch.int64, device=dev)
        self.send_buf = torch.empty(self.send_index.numel(), dtype=self.tdtype, device=dev)

        # ---- the two local matrices ---------------------------------------------
        self.x_ext = torch.zeros(rows + self.nhalo, dtype=self.tdtype, device=dev)
        self.x_local = self.x_ext[:rows]
        self.halo_buf = self.x_ext[rows:]
        interior_entry = ~bnd[row_of]
        rp_i, cj_i, va_i = _csr_select(rows, counts, interior_entry, cj64, va)
        self.A_int = self.engine.matrix(fmt, rows, rows, rp_i, (cj_i - my_begin).to(torch.int32), va_i, p0, p1)
        self.bnd_rows = torch.nonzero(bnd).flatten().to(torch.int32)
        nb = int(self.bnd_rows.numel())
        if nb:
            cols_ext = torch.where(local, cj64 - my_begin, rows + torch.searchsorted(halo, cj64))
            keep = ~interior_entry
            cnt_b = counts[bnd]
            rp_b = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
            torch.cumsum(cnt_b, 0, out=rp_b[1:])
            self.A_bnd = self.engine.matrix(L.CSR_VECTOR, nb, rows + self.nhalo, rp_b.to(torch.int32),
                                            cols_ext[keep].to(torch.int32), va[keep], 0, 0, row_map=self.bnd_rows)
        self.comm_stream = torch.cuda.Stream(device=dev) if self.on_gpu else None
        self.stored = self.A_int.stored + (self.A_bnd.stored if self.A_bnd else 0)
        self.variant = self.A_int.variant

    # ------------------------------------------------------------------------------
    def _all_gather(self, t):
        if self.world == 1:
            return [t]
        src = t.cpu() if self.stage_host else t
        out = [torch.zeros_like(src) for _ in range(self.world)]
        dist.all_gather(out, src, group=self.group)
        return [o.to(t.device) for o in out]

    def _sendrecv(self, sends, recvs):
        """Post every send and receive of one exchange together (one NCCL group: direct peer copies)."""
        if self.stage_host:
            host_s = {q: t.cpu() for q, t in sends.items()}
            host_r = {q: torch.empty(t.shape, dtype=t.dtype) for q, t in recvs.items()}
        else:
            host_s, host_r = sends, recvs
        ops = [dist.P2POp(dist.isend, t, q, group=self.group) for q, t in host_s.items()]
        ops += [dist.P2POp(dist.irecv, t, q, group=self.group) for q, t in host_r.items()]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.stage_host:
            for q, t in recvs.items():
                t.copy_(host_r[q])

    def _exchange_halo(self):
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
        if self.send_index.numel():
            torch.index_select(self.x_local, 0, self.send_index, out=self.send_buf)
        if self.on_gpu:
            main = torch.cuda.current_stream()
            self.comm_stream.wait_stream(main)
            with torch.cuda.stream(self.comm_stream):
                self._exchange_halo()
            self.A_int.mult(self.x_local, y)           # overlaps the exchange
            main.wait_stream(self.comm_stream)
        else:
            self._exchange_halo()
            self.A_int.mult(self.x_local, y)
        if self.A_bnd is not None:
            self.A_bnd.mult(self.x_ext, y)
        return y

    def exchange_only(self):
        """The x exchange of one product without the local products (for timing the communication alone)."""
        if self.A_glob is not None:
            if self.world > 1 and not self.stage_host:
                dist.all_gather_into_tensor(self.x_full, self.x_local.contiguous(), group=self.group)
            return
        if self.send_index.numel():
            torch.index_select(self.x_local, 0, self.send_index, out=self.send_buf)
        self._exchange_halo()

    def compute_only(self, y):
        """The local products of one product against whatever x / halo values are in place (no exchange)."""
        if self.A_glob is not None:
            self.A_glob.mult(self.x_full, y)
            return y
        self.A_int.mult(self.x_local, y)
        if self.A_bnd is not None:
            self.A_bnd.mult(self.x_ext, y)
        return y

    def norm2(self, v):
        """Global ||v||_2 of a row-partitioned vector (the driver's vec_norm, reference driver_spmv.c:21-26)."""
        s = torch.sum(v.to(torch.float64) ** 2).reshape(1)
        if self.world > 1:
            dist.all_reduce(s, group=self.group)
        return float(torch.sqrt(s))
