// Row-block partitioned product: the matrix cut into contiguous row blocks, one per GPU.
//
// The reference has no partitioned product; its one partitioning precedent is the equal-nnz
// contiguous chunking of create_csr_bal (reference src/csr_bal.c:83-85), which
// spmv_hip_partition_rows applies to whole rows.  A block of a square matrix owns rows
// [row_begin, row_begin + rows) and the same range of x and y.  Its rows are split into
//   interior rows -- every column inside the block's own range: multiplied by the interior
//                    matrix (the caller's format) against the local part of x, which needs no
//                    communication and holds almost all non-zeros;
//   boundary rows -- at least one column owned by another block: kept as a compact CSR with a
//                    row map, multiplied against [x_local | halo] once the halo (the x entries
//                    of the remote columns, sorted by global column) has arrived.  Lane-per-row
//                    CSR when the interior format keeps mult_csr_base's summation order (so the
//                    whole product can be bit-identical to the unpartitioned one), vector CSR
//                    beside the reassociating formats.
// Everything is built in HBM from the block's CSR (global column indices): one classifying pass,
// three scans, one fill pass, a radix sort + unique of the remote columns.
//
// spmv_hip_rowblock_* is the per-GPU piece an exchange layer drives (one process per GPU over
// RCCL: spmv_amd/distributed.py).  spmv_hip_dist_* drives all blocks of one process: the caller's
// one thread hands every product to a helper thread per GPU (hidden inside the handle), each of
// which issues its own block's launches on a stream pair, the halo pushed by peer-to-peer copies
// over xGMI while the interior products run -- what the C driver uses.

#include "internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

using namespace spmv;

struct spmv_hip_rowblock {
    int device = 0, format = 0, dtype = 0;
    int rows = 0, row_begin = 0, columns_global = 0;
    int64_t nnz = 0, interior_nnz = 0, boundary_nnz = 0;
    int boundary_rows = 0, halo = 0;
    spmv_hip_matrix *interior = nullptr;   // rows x rows, the caller's format; boundary rows are empty in it
    spmv_hip_matrix *boundary = nullptr;   // boundary_rows x (rows + halo), lane-per-row or vector CSR with a row map
    int *d_bnd_rows = nullptr;             // [boundary_rows] local row of each boundary row
    int *d_halo_cols = nullptr;            // [halo] global column ids, ascending
    int *d_send_index = nullptr;           // [nsend] local x entries other blocks need, in sending order
    int nsend = 0;
};

namespace spmv {

static inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)std::max<int64_t>((n + per - 1) / per, 1); }

// per row: how many of its entries point outside [begin, end); the row's length goes to the interior
// or to the boundary count accordingly
__global__ __launch_bounds__(256) void rb_classify_kernel(int rows, int begin, int end, const int *__restrict__ row_ptr,
                                                          const int *__restrict__ col, int *__restrict__ remote_count,
                                                          int *__restrict__ interior_count,
                                                          int *__restrict__ boundary_count,
                                                          int *__restrict__ boundary_flag)
{
    const int r = (int)(blockIdx.x * 256 + threadIdx.x);
    if (r > rows) return;
    int remote = 0, len = 0;
    if (r < rows) {
        const int a = row_ptr[r], b = row_ptr[r + 1];
        len = b - a;
        for (int l = a; l < b; l++) {
            const int c = col[l];
            remote += c < begin || c >= end;
        }
    }
    remote_count[r] = remote;                       // entry `rows` = 0 closes the scans
    interior_count[r] = remote ? 0 : len;
    boundary_count[r] = remote ? len : 0;
    boundary_flag[r] = remote > 0;
}

template <typename T>
__global__ __launch_bounds__(256) void rb_fill_kernel(int rows, int begin, int end, const int *__restrict__ row_ptr,
                                                      const int *__restrict__ col, const T *__restrict__ val,
                                                      const int *__restrict__ remote_pos,
                                                      const int *__restrict__ interior_ptr,
                                                      const int *__restrict__ boundary_pos,
                                                      const int *__restrict__ boundary_index,
                                                      int *__restrict__ icol, T *__restrict__ ival,
                                                      int *__restrict__ bnd_rows, int *__restrict__ brow_ptr,
                                                      int *__restrict__ bcol, T *__restrict__ bval,
                                                      int *__restrict__ remote_cols)
{
    const int r = (int)(blockIdx.x * 256 + threadIdx.x);
    if (r > rows) return;
    if (r == rows) {                                // closes the boundary row pointers
        brow_ptr[boundary_index[rows]] = boundary_pos[rows];
        return;
    }
    const int a = row_ptr[r], b = row_ptr[r + 1];
    const bool bnd = remote_pos[r + 1] > remote_pos[r];
    if (!bnd) {
        int w = interior_ptr[r];
        for (int l = a; l < b; l++, w++) { icol[w] = col[l] - begin; ival[w] = val[l]; }
    } else {
        const int k = boundary_index[r];
        bnd_rows[k] = r;
        int w = boundary_pos[r], q = remote_pos[r];
        brow_ptr[k] = w;
        for (int l = a; l < b; l++, w++) {
            const int c = col[l];
            bcol[w] = c;                            // global for now; rb_map_columns_kernel rewrites it
            bval[w] = val[l];
            if (c < begin || c >= end) remote_cols[q++] = c;
        }
    }
}

// boundary entries: a local column becomes its index in x_local, a remote one rows + its rank in the halo
__global__ __launch_bounds__(256) void rb_map_columns_kernel(int64_t n, int rows, int begin, int end, int nhalo,
                                                             const int *__restrict__ halo, int *__restrict__ bcol)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int c = bcol[k];
    if (c >= begin && c < end) { bcol[k] = c - begin; return; }
    int lo = 0, hi = nhalo;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (halo[mid] < c) lo = mid + 1; else hi = mid;
    }
    bcol[k] = rows + lo;
}

template <typename T>
__global__ __launch_bounds__(256) void rb_pack_kernel(int n, const int *__restrict__ index, const T *__restrict__ x,
                                                      T *__restrict__ out)
{
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k < n) out[k] = x[index[k]];
}

// The send list gathered and delivered in one launch: entry k of segment s (the entries seg_begin[s] ..
// seg_begin[s+1]-1 of the list, all for one consumer) is stored to dst[s][k - seg_begin[s]] -- memory of the
// consumer's GPU when peer access is enabled, so the halo crosses xGMI as the kernel's own stores and no copy
// engine has to be programmed per consumer and product.
constexpr int kMaxPushSegments = 16;
struct push_plan {
    int nseg;
    int seg_begin[kMaxPushSegments + 1];
    void *dst[kMaxPushSegments];
};

template <typename T>
__global__ __launch_bounds__(256) void rb_push_kernel(int n, const int *__restrict__ index, const T *__restrict__ x,
                                                      push_plan plan)
{
    const int k = (int)(blockIdx.x * 256 + threadIdx.x);
    if (k >= n) return;
    int s = 0;
    while (s + 1 < plan.nseg && k >= plan.seg_begin[s + 1]) s++;
    ((T *)plan.dst[s])[k - plan.seg_begin[s]] = x[index[k]];
    // Another GPU's memory is cached in this GPU's L2 without coherence; the release at the end of a kernel is
    // only guaranteed to reach this GPU's own scope.  The consumer is ordered behind this kernel by an event,
    // so the lines have to leave for the consumer's memory before the kernel ends: a system-scope release.
    __threadfence_system();
}

// sorted unique values of d_in[0..n) -> *d_out (hipMalloc'ed), *count
static int sort_unique(const int *d_in, int64_t n, int **d_out, int *count, hipStream_t st)
{
    *d_out = nullptr; *count = 0;
    if (n == 0) return 0;
    int *d_sorted = nullptr, *d_unique = nullptr, *d_count = nullptr;
    void *d_temp = nullptr;
    int rc = 1;
    do {
        const size_t sz = (size_t)n;
        SPMV_HIP_TRY(hipMalloc((void **)&d_sorted, sz * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_unique, sz * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_count, sizeof(int)), break);
        size_t t1 = 0, t2 = 0;
        SPMV_HIP_TRY(rocprim::radix_sort_keys(nullptr, t1, d_in, d_sorted, sz, 0, 32, st), break);
        SPMV_HIP_TRY(rocprim::unique(nullptr, t2, d_sorted, d_unique, d_count, sz, rocprim::equal_to<int>(), st), break);
        SPMV_HIP_TRY(hipMalloc(&d_temp, std::max<size_t>(std::max(t1, t2), 16)), break);
        SPMV_HIP_TRY(rocprim::radix_sort_keys(d_temp, t1, d_in, d_sorted, sz, 0, 32, st), break);
        SPMV_HIP_TRY(rocprim::unique(d_temp, t2, d_sorted, d_unique, d_count, sz, rocprim::equal_to<int>(), st), break);
        SPMV_HIP_TRY(hipMemcpyAsync(count, d_count, sizeof(int), hipMemcpyDeviceToHost, st), break);
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        *d_out = d_unique;
        d_unique = nullptr;
        rc = 0;
    } while (0);
    void *scratch[] = {d_sorted, d_unique, d_count, d_temp};
    for (void *p : scratch)
        if (p) (void)hipFree(p);
    return rc;
}

template <typename T>
static int build_rowblock(spmv_hip_rowblock *rb, const int *d_row_ptr, const int *d_col, const T *d_val, int64_t p0,
                          int64_t p1, hipStream_t st)
{
    const int rows = rb->rows, begin = rb->row_begin, end = rb->row_begin + rb->rows;
    const size_t n1 = (size_t)rows + 1;
    int *d_cnt[4] = {nullptr, nullptr, nullptr, nullptr};      // remote, interior, boundary counts, boundary flag
    int *d_pos[4] = {nullptr, nullptr, nullptr, nullptr};      // their exclusive scans
    int *icol = nullptr, *bcol = nullptr, *brp = nullptr, *d_remote = nullptr, *d_irp = nullptr;
    T *ival = nullptr, *bval = nullptr;
    int rc = 1;
    do {
        bool ok = true;
        for (int k = 0; k < 4 && ok; k++)
            ok = hipMalloc((void **)&d_cnt[k], n1 * sizeof(int)) == hipSuccess &&
                 hipMalloc((void **)&d_pos[k], n1 * sizeof(int)) == hipSuccess;
        if (!ok) { set_error("rowblock: hipMalloc of the row counters failed"); break; }
        hipLaunchKernelGGL(rb_classify_kernel, dim3(blocks_for((int64_t)rows + 1)), dim3(256), 0, st, rows, begin, end,
                           d_row_ptr, d_col, d_cnt[0], d_cnt[1], d_cnt[2], d_cnt[3]);
        SPMV_HIP_TRY(hipGetLastError(), break);
        for (int k = 0; k < 4 && ok; k++) ok = device_exclusive_scan(d_cnt[k], d_pos[k], n1, st) == 0;
        if (!ok) break;
        int totals[4];
        for (int k = 0; k < 4 && ok; k++)           // on the caller's stream: a plain hipMemcpy does not wait for a non-blocking one
            ok = hipMemcpyAsync(&totals[k], d_pos[k] + rows, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (!ok) { set_error("rowblock: reading the scan totals failed"); break; }
        const int64_t nremote = totals[0];
        rb->interior_nnz = totals[1];
        rb->boundary_nnz = totals[2];
        rb->boundary_rows = totals[3];
        const size_t ni = (size_t)std::max<int64_t>(rb->interior_nnz, 1), nb = (size_t)std::max<int64_t>(rb->boundary_nnz, 1);
        SPMV_HIP_TRY(hipMalloc((void **)&icol, ni * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&ival, ni * sizeof(T)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&bcol, nb * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&bval, nb * sizeof(T)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&brp, ((size_t)rb->boundary_rows + 1) * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&rb->d_bnd_rows, (size_t)std::max(rb->boundary_rows, 1) * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_remote, (size_t)std::max<int64_t>(nremote, 1) * sizeof(int)), break);
        hipLaunchKernelGGL((rb_fill_kernel<T>), dim3(blocks_for((int64_t)rows + 1)), dim3(256), 0, st, rows, begin, end,
                           d_row_ptr, d_col, d_val, d_pos[0], d_pos[1], d_pos[2], d_pos[3], icol, ival, rb->d_bnd_rows, brp,
                           bcol, bval, d_remote);
        SPMV_HIP_TRY(hipGetLastError(), break);
        if (sort_unique(d_remote, nremote, &rb->d_halo_cols, &rb->halo, st)) break;
        if (rb->boundary_nnz > 0) {
            hipLaunchKernelGGL(rb_map_columns_kernel, dim3(blocks_for(rb->boundary_nnz)), dim3(256), 0, st, rb->boundary_nnz,
                               rows, begin, end, rb->halo, rb->d_halo_cols, bcol);
            SPMV_HIP_TRY(hipGetLastError(), break);
        }
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        // the interior matrix keeps all `rows` rows (boundary rows are empty in it): its row pointers are
        // the scan of the interior counts, which the handle takes over together with icol / ival
        d_irp = d_pos[1];
        d_pos[1] = nullptr;
        rb->interior = adopt_device_csr(rb->format, rows, rows, rb->interior_nnz, d_irp, icol, ival, rb->dtype, p0, p1,
                                        /*own=*/true, st);
        d_irp = nullptr; icol = nullptr; ival = nullptr;       // freed with (or by the failed creation of) the handle
        if (!rb->interior) break;
        if (rb->boundary_rows > 0) {
            const bool exact_family = rb->format == SPMV_HIP_CSR_SCALAR || rb->format == SPMV_HIP_SELL ||
                                      rb->format == SPMV_HIP_SELLP || rb->format == SPMV_HIP_ELL ||
                                      rb->format == SPMV_HIP_SELL_DFC;
            rb->boundary = adopt_device_csr(exact_family ? SPMV_HIP_CSR_SCALAR : SPMV_HIP_CSR_VECTOR, rb->boundary_rows, rows + rb->halo, rb->boundary_nnz, brp,
                                            bcol, bval, rb->dtype, 0, 0, /*own=*/true, st);
            brp = nullptr; bcol = nullptr; bval = nullptr;
            if (!rb->boundary) break;
            if (spmv_hip_set_row_map(rb->boundary, rb->d_bnd_rows)) break;
        }
        rc = 0;
    } while (0);
    void *scratch[] = {d_cnt[0], d_cnt[1], d_cnt[2], d_cnt[3], d_pos[0], d_pos[1], d_pos[2], d_pos[3], icol, ival, bcol, bval,
                       brp, d_remote, d_irp};
    for (void *p : scratch)
        if (p) (void)hipFree(p);
    return rc;
}

// block p ends at the first row whose row_ptr reaches (p+1)*nnz/parts, moved to the nearest multiple of snap
template <typename P>
static int partition_rows_impl(int rows, const P *row_ptr, int parts, int snap, int *bounds)
{
    clear_error();
    if (rows < 0 || parts < 1 || !row_ptr || !bounds) { set_error("partition_rows: bad argument"); return 1; }
    if (snap < 1) snap = 1;
    const int64_t nnz = row_ptr[rows];
    bounds[0] = 0;
    for (int p = 1; p < parts; p++) {
        const int64_t target = (int64_t)((__int128)nnz * p / parts);
        int64_t c = std::lower_bound(row_ptr, row_ptr + rows + 1, target, [](P v, int64_t t) { return (int64_t)v < t; }) - row_ptr;
        c = (c + snap / 2) / snap * snap;
        bounds[p] = (int)std::min<int64_t>(std::max<int64_t>(c, bounds[p - 1]), rows);
    }
    bounds[parts] = rows;
    return 0;
}

} // namespace spmv

extern "C" {

int spmv_hip_partition_rows(int rows, const int *row_ptr, int parts, int snap, int *bounds)
{
    return partition_rows_impl(rows, row_ptr, parts, snap, bounds);
}

int spmv_hip_partition_rows64(int rows, const int64_t *row_ptr, int parts, int snap, int *bounds)
{
    return partition_rows_impl(rows, row_ptr, parts, snap, bounds);
}

spmv_hip_rowblock *spmv_hip_rowblock_create(int format, int rows, int row_begin, int columns_global, int64_t nnz,
                                            const int *d_row_ptr, const int *d_col, const void *d_val, int dtype,
                                            int64_t p0, int64_t p1, void *stream)
{
    clear_error();
    if (rows < 0 || row_begin < 0 || nnz < 0 || (int64_t)row_begin + rows > columns_global) {
        set_error("rowblock: rows [%d, %lld) do not lie inside the %d columns of the square matrix", row_begin,
                  (long long)row_begin + rows, columns_global);
        return nullptr;
    }
    if (dtype != SPMV_HIP_F64 && dtype != SPMV_HIP_F32) { set_error("unknown dtype %d", dtype); return nullptr; }
    if (nnz > 0x7fffffffLL - 4096) { set_error("rowblock: %lld non-zeros exceed the 32-bit positions of one block", (long long)nnz); return nullptr; }
    if (!d_row_ptr || (nnz > 0 && (!d_col || !d_val))) { set_error("NULL device CSR array"); return nullptr; }
    spmv_hip_rowblock *rb = new (std::nothrow) spmv_hip_rowblock();
    if (!rb) { set_error("out of host memory"); return nullptr; }
    if (hipGetDevice(&rb->device) != hipSuccess) { set_error("no HIP device available"); delete rb; return nullptr; }
    rb->format = format; rb->dtype = dtype; rb->rows = rows; rb->row_begin = row_begin; rb->columns_global = columns_global;
    rb->nnz = nnz;
    const int rc = dtype == SPMV_HIP_F32
                       ? build_rowblock<float>(rb, d_row_ptr, d_col, (const float *)d_val, p0, p1, (hipStream_t)stream)
                       : build_rowblock<double>(rb, d_row_ptr, d_col, (const double *)d_val, p0, p1, (hipStream_t)stream);
    if (rc) {
        char keep[512];
        snprintf(keep, sizeof keep, "%s", spmv_hip_last_error());
        spmv_hip_rowblock_free(rb);
        set_error("%s", keep);
        return nullptr;
    }
    return rb;
}

int spmv_hip_rowblock_free(spmv_hip_rowblock *rb)
{
    if (!rb) return 0;
    DeviceGuard on(rb->device);
    if (rb->interior) spmv_hip_free(rb->interior);
    if (rb->boundary) spmv_hip_free(rb->boundary);
    void *ptrs[] = {rb->d_bnd_rows, rb->d_halo_cols, rb->d_send_index};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete rb;
    return 0;
}

int spmv_hip_rowblock_get_info(const spmv_hip_rowblock *rb, spmv_hip_rowblock_info *info)
{
    if (!rb || !info) { set_error("NULL argument"); return 1; }
    memset(info, 0, sizeof *info);
    info->format = rb->format; info->dtype = rb->dtype; info->device = rb->device;
    info->rows = rb->rows; info->row_begin = rb->row_begin; info->columns_global = rb->columns_global;
    info->nnz = rb->nnz; info->interior_nnz = rb->interior_nnz; info->boundary_nnz = rb->boundary_nnz;
    info->boundary_rows = rb->boundary_rows; info->halo = rb->halo; info->send = rb->nsend;
    spmv_hip_info mi;
    if (rb->interior && spmv_hip_get_info(rb->interior, &mi) == 0) {
        info->interior_stored = mi.stored; info->interior_variant = mi.variant; info->device_bytes += mi.device_bytes;
    }
    if (rb->boundary && spmv_hip_get_info(rb->boundary, &mi) == 0) {
        info->boundary_lanes_per_row = mi.lanes_per_row; info->boundary_variant = mi.variant; info->device_bytes += mi.device_bytes;
    }
    info->device_bytes += (int64_t)sizeof(int) * ((int64_t)rb->boundary_rows + rb->halo + rb->nsend);
    return 0;
}

int spmv_hip_rowblock_set_option(spmv_hip_rowblock *rb, const char *name, int64_t value)
{
    clear_error();
    if (!rb || !name) { set_error("NULL argument"); return 1; }
    if (strcmp(name, "exact_order")) { set_error("rowblock: unknown option '%s'", name); return 1; }
    if (spmv_hip_set_option(rb->interior, name, value)) return 1;
    if (rb->boundary && spmv_hip_set_option(rb->boundary, name, value)) return 1;
    return 0;
}

int spmv_hip_rowblock_analyze(spmv_hip_rowblock *rb, int reps)
{
    clear_error();
    if (!rb) { set_error("NULL handle"); return 1; }
    if (spmv_hip_analyze(rb->interior, reps)) return 1;
    if (rb->boundary) {
        // the kernel trial writes y[0..boundary_rows) of a vector of its own: it runs without the row map
        if (spmv_hip_set_row_map(rb->boundary, nullptr)) return 1;
        const int rc = spmv_hip_analyze(rb->boundary, reps);
        if (spmv_hip_set_row_map(rb->boundary, rb->d_bnd_rows) || rc) return 1;
    }
    return 0;
}

int spmv_hip_rowblock_halo_columns(const spmv_hip_rowblock *rb, int *columns)
{
    clear_error();
    if (!rb || (!columns && rb->halo > 0)) { set_error("NULL argument"); return 1; }
    if (rb->halo == 0) return 0;
    DeviceGuard on(rb->device);
    if (!on.ok) return 1;
    SPMV_HIP_TRY(hipMemcpy(columns, rb->d_halo_cols, sizeof(int) * (size_t)rb->halo, hipMemcpyDeviceToHost), return 1);
    return 0;
}

int spmv_hip_rowblock_set_send_list(spmv_hip_rowblock *rb, int count, const int *local_index)
{
    clear_error();
    if (!rb || count < 0 || (count > 0 && !local_index)) { set_error("rowblock: bad send list"); return 1; }
    for (int k = 0; k < count; k++)
        if (local_index[k] < 0 || local_index[k] >= rb->rows) {
            set_error("rowblock: send list entry %d = %d is not a local row of this block (%d rows)", k, local_index[k], rb->rows);
            return 1;
        }
    DeviceGuard on(rb->device);
    if (!on.ok) return 1;
    if (rb->d_send_index) { (void)hipFree(rb->d_send_index); rb->d_send_index = nullptr; rb->nsend = 0; }
    if (count == 0) return 0;
    SPMV_HIP_TRY(hipMalloc((void **)&rb->d_send_index, sizeof(int) * (size_t)count), return 1);
    SPMV_HIP_TRY(hipMemcpy(rb->d_send_index, local_index, sizeof(int) * (size_t)count, hipMemcpyHostToDevice), return 1);
    rb->nsend = count;
    return 0;
}

int spmv_hip_rowblock_pack(spmv_hip_rowblock *rb, const void *d_x, void *d_send, void *stream)
{
    if (!rb) { set_error("NULL handle"); return 1; }
    if (rb->nsend == 0) return 0;
    if (!d_x || !d_send) { set_error("rowblock_pack: NULL vector"); return 1; }
    DeviceGuard on(rb->device);
    if (!on.ok) return 1;
    const dim3 grid(blocks_for(rb->nsend)), block(256);
    if (rb->dtype == SPMV_HIP_F32)
        hipLaunchKernelGGL(rb_pack_kernel<float>, grid, block, 0, (hipStream_t)stream, rb->nsend, rb->d_send_index,
                           (const float *)d_x, (float *)d_send);
    else
        hipLaunchKernelGGL(rb_pack_kernel<double>, grid, block, 0, (hipStream_t)stream, rb->nsend, rb->d_send_index,
                           (const double *)d_x, (double *)d_send);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

int spmv_hip_rowblock_push(spmv_hip_rowblock *rb, const void *d_x, int nseg, const int *seg_begin, void *const *dst,
                           void *stream)
{
    if (!rb) { set_error("NULL handle"); return 1; }
    if (rb->nsend == 0) return 0;
    if (!d_x || !seg_begin || !dst || nseg < 1 || nseg > kMaxPushSegments) {
        set_error("rowblock_push: needs 1..%d segments and their destinations", kMaxPushSegments);
        return 1;
    }
    push_plan plan;
    plan.nseg = nseg;
    for (int k = 0; k <= nseg; k++) plan.seg_begin[k] = seg_begin[k];
    for (int k = 0; k < nseg; k++) {
        if (seg_begin[k + 1] < seg_begin[k] || !dst[k]) { set_error("rowblock_push: bad segment %d", k); return 1; }
        plan.dst[k] = dst[k];
    }
    if (seg_begin[0] != 0 || seg_begin[nseg] != rb->nsend) {
        set_error("rowblock_push: the segments cover %d..%d, the send list has %d entries", seg_begin[0], seg_begin[nseg], rb->nsend);
        return 1;
    }
    DeviceGuard on(rb->device);
    if (!on.ok) return 1;
    const dim3 grid(blocks_for(rb->nsend)), block(256);
    if (rb->dtype == SPMV_HIP_F32)
        hipLaunchKernelGGL(rb_push_kernel<float>, grid, block, 0, (hipStream_t)stream, rb->nsend, rb->d_send_index,
                           (const float *)d_x, plan);
    else
        hipLaunchKernelGGL(rb_push_kernel<double>, grid, block, 0, (hipStream_t)stream, rb->nsend, rb->d_send_index,
                           (const double *)d_x, plan);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

int spmv_hip_rowblock_mult_interior(spmv_hip_rowblock *rb, const void *d_x, void *d_y, void *stream)
{
    if (!rb) { set_error("NULL handle"); return 1; }
    return spmv_hip_mult_dev(rb->interior, d_x, d_y, stream);
}

int spmv_hip_rowblock_mult_boundary(spmv_hip_rowblock *rb, const void *d_x_ext, void *d_y, void *stream)
{
    if (!rb) { set_error("NULL handle"); return 1; }
    if (!rb->boundary) return 0;
    return spmv_hip_mult_dev(rb->boundary, d_x_ext, d_y, stream);
}

} // extern "C"

// ------------------------------------------------------ all blocks of one process --
//
// The caller is one host thread (the reference's driver is single-threaded,
// driver_spmv.c:112-347).  Issuing every block's launches from that thread makes the host the
// bottleneck as the blocks shrink (about ten asynchronous calls per block and product), so the
// handle keeps one helper thread per block, bound to the block's device: a product is ONE command
// that all helpers execute side by side, each issuing only its own block's sequence
//     mark x -> interior product -> pack -> pushes to the consumers -> mark "sent"
//     -> wait for the producers' "sent" marks -> boundary product.
// The one cross-thread dependency inside a product -- a consumer may only wait on a producer's
// "sent" event once that event has been recorded for THIS product -- is a per-block generation
// counter the producer publishes after its hipEventRecord.  SPMV_DIST_THREADS=0 keeps everything
// on the calling thread (same sequence, issued stage by stage over all blocks).

enum { kOpNone = 0, kOpProduct, kOpSync, kOpProductSync, kOpNormScale, kOpAnalyze, kOpExit };

struct spmv_hip_dist {
    int ngpus = 0, dtype = 0, rows = 0;
    int64_t nnz = 0;
    std::vector<int> device, bounds;                 // bounds: ngpus+1 row boundaries
    std::vector<spmv_hip_rowblock *> block;
    std::vector<char *> d_x, d_y, d_send;            // per block: [x_local | halo], y, packed entries for the others
    std::vector<double *> d_norm, d_partial;         // per block: its norm and the partial sums of that reduction
    double *h_norm = nullptr;                        // pinned: the blocks' norms, read by every helper
    std::vector<hipStream_t> compute, comm;
    std::vector<hipEvent_t> x_ready, sent;
    // block q sends block p the entries [send_off[q][p], send_off[q][p+1]) of its send buffer; they land at
    // halo position recv_off[p][q] of block p
    std::vector<std::vector<int>> send_off, recv_off;
    // block q delivers its whole send list with one kernel (stores straight into the consumers' halo buffers)
    // when every consumer is reachable by peer access and there are few enough of them: push_nseg[q] > 0
    std::vector<int> push_nseg;
    std::vector<std::vector<int>> push_begin;
    std::vector<std::vector<void *>> push_dst;
    // peer[q * ngpus + p]: how a push from block q to block p travels -- 2 same device, 1 direct peer-to-peer
    // write (peer access enabled), 0 staged through host memory by the runtime
    std::vector<int> peer;

    // helper threads
    bool threaded = false;
    std::vector<std::thread> helper;
    std::mutex mu;
    std::condition_variable cv_cmd, cv_done;
    std::atomic<uint64_t> seq{0};                    // command generation; `op` / `op_arg` are written before it moves
    int op = kOpNone, op_arg = 0;
    std::atomic<int> done{0}, failed{0}, norm_arrived{0};
    std::unique_ptr<std::atomic<uint64_t>[]> sent_seq;   // per block: generation of its last recorded "sent" event
    std::string error;                               // first failure of a command (under mu)
};

namespace spmv {

static inline void cpu_relax() { __builtin_ia32_pause(); }

// Wait for `ready`: spin for the few microseconds a command or a neighbour normally takes, then sleep.
template <class Pred> static void wait_for(Pred ready, std::mutex &mu, std::condition_variable &cv)
{
    for (int i = 0; i < 30000; i++) {
        if (ready()) return;
        cpu_relax();
    }
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, ready);
}

static void wake(std::mutex &mu, std::condition_variable &cv)
{
    { std::lock_guard<std::mutex> lk(mu); }         // a sleeper is either before its check or already waiting
    cv.notify_all();
}

static void dist_fail(spmv_hip_dist *d, const char *what)
{
    std::lock_guard<std::mutex> lk(d->mu);
    if (!d->failed.load()) d->error = what && what[0] ? what : "dist: a device call failed";
    d->failed.store(1);
}

// ---- the stages of one product, for block p (the block's device is current) ----

static int stage_interior(spmv_hip_dist *d, int p)
{
    // x of this product is in place on the compute stream: mark it, then start the interior rows
    SPMV_HIP_TRY(hipEventRecord(d->x_ready[p], d->compute[p]), return 1);
    return spmv_hip_rowblock_mult_interior(d->block[p], d->d_x[p], d->d_y[p], d->compute[p]);
}

static int stage_push(spmv_hip_dist *d, int q)
{
    const size_t es = dtype_size(d->dtype);
    const int P = d->ngpus;
    SPMV_HIP_TRY(hipStreamWaitEvent(d->comm[q], d->x_ready[q], 0), return 1);
    if (d->push_nseg[q] > 0) {
        if (spmv_hip_rowblock_push(d->block[q], d->d_x[q], d->push_nseg[q], d->push_begin[q].data(), d->push_dst[q].data(),
                                   d->comm[q]))
            return 1;
        SPMV_HIP_TRY(hipEventRecord(d->sent[q], d->comm[q]), return 1);
        return 0;
    }
    if (spmv_hip_rowblock_pack(d->block[q], d->d_x[q], d->d_send[q], d->comm[q])) return 1;
    for (int p = 0; p < P; p++) {
        const int count = d->send_off[q][p + 1] - d->send_off[q][p];
        if (p == q || count == 0) continue;
        char *dst = d->d_x[p] + ((size_t)d->block[p]->rows + (size_t)d->recv_off[p][q]) * es;
        const char *src = d->d_send[q] + (size_t)d->send_off[q][p] * es;
        SPMV_HIP_TRY(hipMemcpyPeerAsync(dst, d->device[p], src, d->device[q], (size_t)count * es, d->comm[q]), return 1);
    }
    SPMV_HIP_TRY(hipEventRecord(d->sent[q], d->comm[q]), return 1);
    return 0;
}

// gen != 0: wait until every producer has recorded its "sent" event of command `gen` (helper threads)
static int stage_boundary(spmv_hip_dist *d, int p, uint64_t gen)
{
    for (int q = 0; q < d->ngpus; q++) {
        if (q == p || d->send_off[q][p + 1] == d->send_off[q][p]) continue;
        if (gen) {
            while (d->sent_seq[q].load(std::memory_order_acquire) < gen) {
                if (d->failed.load(std::memory_order_relaxed)) { set_error("dist: another block's exchange failed"); return 1; }
                cpu_relax();
            }
        }
        SPMV_HIP_TRY(hipStreamWaitEvent(d->compute[p], d->sent[q], 0), return 1);
    }
    return spmv_hip_rowblock_mult_boundary(d->block[p], d->d_x[p], d->d_y[p], d->compute[p]);
}

static int stage_sync(spmv_hip_dist *d, int p)
{
    SPMV_HIP_TRY(hipStreamSynchronize(d->comm[p]), return 1);
    SPMV_HIP_TRY(hipStreamSynchronize(d->compute[p]), return 1);
    return 0;
}

// ||y_p||_2 into the pinned slot p; the caller has the compute stream synchronised when this returns
static int stage_norm(spmv_hip_dist *d, int p)
{
    if (launch_norm2_with(d->d_partial[p], d->block[p]->rows, d->d_y[p], d->dtype, d->d_norm[p], d->compute[p])) return 1;
    SPMV_HIP_TRY(hipMemcpyAsync(&d->h_norm[p], d->d_norm[p], sizeof(double), hipMemcpyDeviceToHost, d->compute[p]), return 1);
    SPMV_HIP_TRY(hipStreamSynchronize(d->compute[p]), return 1);
    return 0;
}

// the norm over all blocks, summed in block order: the same bits in every helper
static double dist_norm(const spmv_hip_dist *d)
{
    double sumsq = 0.0;
    for (int q = 0; q < d->ngpus; q++) sumsq += d->h_norm[q] * d->h_norm[q];
    return std::sqrt(sumsq);
}

static int stage_scale(spmv_hip_dist *d, int p, double norm)
{
    if (launch_scale_value(d->block[p]->rows, d->d_y[p], d->d_x[p], d->dtype, norm, d->compute[p])) return 1;
    SPMV_HIP_TRY(hipStreamSynchronize(d->compute[p]), return 1);
    return 0;
}

// ---- helper threads ----

static int helper_execute(spmv_hip_dist *d, int p, int op, int arg, uint64_t gen)
{
    switch (op) {
    case kOpProduct:
    case kOpProductSync: {
        int rc = stage_interior(d, p) || stage_push(d, p);
        // published even after a failure, so that no consumer spins for it (they see `failed`)
        if (rc) dist_fail(d, spmv_hip_last_error());
        d->sent_seq[p].store(gen, std::memory_order_release);
        if (rc) return 1;
        if (stage_boundary(d, p, gen)) return 1;
        return op == kOpProductSync ? stage_sync(d, p) : 0;
    }
    case kOpSync: return stage_sync(d, p);
    case kOpNormScale: {
        const int rc = stage_norm(d, p);
        if (rc) dist_fail(d, spmv_hip_last_error());
        d->norm_arrived.fetch_add(1, std::memory_order_acq_rel);
        if (rc) return 1;
        while (d->norm_arrived.load(std::memory_order_acquire) < d->ngpus) {
            if (d->failed.load(std::memory_order_relaxed)) { set_error("dist: another block's norm failed"); return 1; }
            cpu_relax();
        }
        if (d->failed.load()) { set_error("dist: another block's norm failed"); return 1; }
        return stage_scale(d, p, dist_norm(d));
    }
    case kOpAnalyze: return spmv_hip_rowblock_analyze(d->block[p], arg);
    default: return 0;
    }
}

static void helper_main(spmv_hip_dist *d, int p)
{
    const bool have_device = hipSetDevice(d->device[p]) == hipSuccess;
    uint64_t seen = 0;
    for (;;) {
        wait_for([&] { return d->seq.load(std::memory_order_acquire) != seen; }, d->mu, d->cv_cmd);
        seen = d->seq.load(std::memory_order_acquire);
        const int op = d->op, arg = d->op_arg;
        if (op != kOpExit) {
            clear_error();
            int rc = 1;
            if (have_device) rc = helper_execute(d, p, op, arg, seen);
            else set_error("dist: hipSetDevice(%d) failed in the helper of block %d", d->device[p], p);
            if (rc) dist_fail(d, spmv_hip_last_error());
        }
        d->done.fetch_add(1, std::memory_order_acq_rel);
        wake(d->mu, d->cv_done);
        if (op == kOpExit) return;
    }
}

// One command to every block; returns when all of them have executed it.
static int dist_run(spmv_hip_dist *d, int op, int arg = 0)
{
    const int P = d->ngpus;
    if (d->threaded) {
        d->done.store(0);
        d->failed.store(0);
        d->norm_arrived.store(0);
        d->op = op;
        d->op_arg = arg;
        d->seq.fetch_add(1, std::memory_order_acq_rel);
        wake(d->mu, d->cv_cmd);
        wait_for([&] { return d->done.load(std::memory_order_acquire) == P; }, d->mu, d->cv_done);
        if (d->failed.load()) {
            std::lock_guard<std::mutex> lk(d->mu);
            set_error("%s", d->error.c_str());
            return 1;
        }
        return 0;
    }
    // the calling thread issues everything, stage by stage over all blocks: first every interior product
    // (no communication, nearly all of the work), so that every GPU is busy within a few launch latencies;
    // then the packing and the pushes on the communication streams; last the boundary rows
    auto each = [&](auto stage) {
        for (int p = 0; p < P; p++) {
            SPMV_HIP_TRY(hipSetDevice(d->device[p]), return 1);
            if (stage(p)) return 1;
        }
        return 0;
    };
    switch (op) {
    case kOpProduct:
    case kOpProductSync:
        if (each([&](int p) { return stage_interior(d, p); }) || each([&](int p) { return stage_push(d, p); }) ||
            each([&](int p) { return stage_boundary(d, p, 0); }))
            return 1;
        return op == kOpProductSync ? each([&](int p) { return stage_sync(d, p); }) : 0;
    case kOpSync: return each([&](int p) { return stage_sync(d, p); });
    case kOpNormScale: {
        if (each([&](int p) { return stage_norm(d, p); })) return 1;
        const double norm = dist_norm(d);
        return each([&](int p) { return stage_scale(d, p, norm); });
    }
    case kOpAnalyze: return each([&](int p) { return spmv_hip_rowblock_analyze(d->block[p], arg); });
    default: return 0;
    }
}

static void dist_stop_helpers(spmv_hip_dist *d)
{
    if (!d->threaded) return;
    d->done.store(0);
    d->op = kOpExit;
    d->seq.fetch_add(1, std::memory_order_acq_rel);
    wake(d->mu, d->cv_cmd);
    for (std::thread &t : d->helper)
        if (t.joinable()) t.join();
    d->helper.clear();
    d->threaded = false;
}

static int dist_upload_x(spmv_hip_dist *d, const double *x)
{
    for (int p = 0; p < d->ngpus; p++) {
        SPMV_HIP_TRY(hipSetDevice(d->device[p]), return 1);
        const int n = d->bounds[p + 1] - d->bounds[p];
        if (n == 0) continue;
        if (d->dtype == SPMV_HIP_F64) {
            SPMV_HIP_TRY(hipMemcpy(d->d_x[p], x + d->bounds[p], sizeof(double) * (size_t)n, hipMemcpyHostToDevice), return 1);
        } else {
            std::vector<float> f(x + d->bounds[p], x + d->bounds[p + 1]);
            SPMV_HIP_TRY(hipMemcpy(d->d_x[p], f.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice), return 1);
        }
    }
    return 0;
}

static int dist_download(spmv_hip_dist *d, const std::vector<char *> &src, double *out)
{
    for (int p = 0; p < d->ngpus; p++) {
        SPMV_HIP_TRY(hipSetDevice(d->device[p]), return 1);
        const int n = d->bounds[p + 1] - d->bounds[p];
        if (n == 0) continue;
        if (d->dtype == SPMV_HIP_F64) {
            SPMV_HIP_TRY(hipMemcpy(out + d->bounds[p], src[p], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost), return 1);
        } else {
            std::vector<float> f((size_t)n);
            SPMV_HIP_TRY(hipMemcpy(f.data(), src[p], sizeof(float) * (size_t)n, hipMemcpyDeviceToHost), return 1);
            for (int i = 0; i < n; i++) out[d->bounds[p] + i] = (double)f[(size_t)i];
        }
    }
    return 0;
}

// Everything of a handle beyond its blocks: vectors, streams, events, peer access, who sends what, helpers.
// d->device, d->bounds and d->block are in place.
static int dist_finish(spmv_hip_dist *d)
{
    const int P = d->ngpus;
    const size_t es = dtype_size(d->dtype);
    d->d_x.assign(P, nullptr); d->d_y.assign(P, nullptr); d->d_send.assign(P, nullptr);
    d->d_norm.assign(P, nullptr); d->d_partial.assign(P, nullptr);
    d->compute.assign(P, nullptr); d->comm.assign(P, nullptr);
    d->x_ready.assign(P, nullptr); d->sent.assign(P, nullptr);
    d->peer.assign((size_t)P * P, 2);
    SPMV_HIP_TRY(hipHostMalloc((void **)&d->h_norm, sizeof(double) * (size_t)P, hipHostMallocDefault), return 1);
    std::vector<std::vector<int>> halo(P);
    const char *require = getenv("SPMV_REQUIRE_P2P");
    const bool require_p2p = require && atoi(require) != 0;
    // peer access for every pair first, before any of the vectors below exists: every allocation the pushes
    // write to is then made with the access already in place
    for (int q = 0; q < P; q++) {
        SPMV_HIP_TRY(hipSetDevice(d->device[q]), return 1);
        // pushes go from q's communication stream into p's memory: q needs access to p
        for (int p = 0; p < P; p++) {
            if (d->device[p] == d->device[q]) continue;
            int can = 0, direct = 0;
            if (hipDeviceCanAccessPeer(&can, d->device[q], d->device[p]) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(d->device[p], 0);
                direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
            d->peer[(size_t)q * P + p] = direct;
            if (!direct) {
                if (require_p2p) {
                    set_error("dist: GPU %d has no peer access to GPU %d (SPMV_REQUIRE_P2P is set)", d->device[q], d->device[p]);
                    return 1;
                }
                fprintf(stderr, "spmv_hip: WARNING: GPU %d has no peer access to GPU %d: the halo of block %d travels to "
                                "block %d through host memory, at PCIe speed (SPMV_REQUIRE_P2P=1 refuses this)\n",
                        d->device[q], d->device[p], q, p);
            }
        }
    }
    for (int q = 0; q < P; q++) {
        SPMV_HIP_TRY(hipSetDevice(d->device[q]), return 1);
        const int n = d->block[q]->rows, nh = d->block[q]->halo;
        halo[q].resize((size_t)nh);
        if (spmv_hip_rowblock_halo_columns(d->block[q], halo[q].data())) return 1;
        SPMV_HIP_TRY(hipMalloc((void **)&d->d_x[q], es * (size_t)std::max(n + nh, 1)), return 1);
        SPMV_HIP_TRY(hipMemset(d->d_x[q], 0, es * (size_t)std::max(n + nh, 1)), return 1);
        SPMV_HIP_TRY(hipMalloc((void **)&d->d_y[q], es * (size_t)std::max(n, 1)), return 1);
        SPMV_HIP_TRY(hipMalloc((void **)&d->d_norm[q], sizeof(double)), return 1);
        SPMV_HIP_TRY(hipMalloc((void **)&d->d_partial[q], sizeof(double) * kNormBlocks), return 1);
        SPMV_HIP_TRY(hipStreamCreateWithFlags(&d->compute[q], hipStreamNonBlocking), return 1);
        SPMV_HIP_TRY(hipStreamCreateWithFlags(&d->comm[q], hipStreamNonBlocking), return 1);
        SPMV_HIP_TRY(hipEventCreateWithFlags(&d->x_ready[q], hipEventDisableTiming), return 1);
        SPMV_HIP_TRY(hipEventCreateWithFlags(&d->sent[q], hipEventDisableTiming), return 1);
    }
    // who sends what: block p's halo is sorted by global column, so what it needs from owner q is one
    // contiguous piece of it; q's send buffer holds the pieces for p = 0, 1, ... in that order
    d->send_off.assign(P, std::vector<int>((size_t)P + 1, 0));
    d->recv_off.assign(P, std::vector<int>((size_t)P + 1, 0));
    std::vector<std::vector<int>> send_list(P);
    for (int p = 0; p < P; p++)
        for (int q = 0; q < P; q++) {
            const auto first = std::lower_bound(halo[p].begin(), halo[p].end(), d->bounds[q]);
            const auto last = std::lower_bound(halo[p].begin(), halo[p].end(), d->bounds[q + 1]);
            d->recv_off[p][q] = (int)(first - halo[p].begin());
            d->recv_off[p][q + 1] = (int)(last - halo[p].begin());
        }
    for (int q = 0; q < P; q++)
        for (int p = 0; p < P; p++) {
            d->send_off[q][p] = (int)send_list[q].size();
            if (p != q)
                for (int k = d->recv_off[p][q]; k < d->recv_off[p][q + 1]; k++) send_list[q].push_back(halo[p][(size_t)k] - d->bounds[q]);
            d->send_off[q][p + 1] = (int)send_list[q].size();
        }
    d->push_nseg.assign(P, 0);
    d->push_begin.assign(P, std::vector<int>());
    d->push_dst.assign(P, std::vector<void *>());
    const char *direct_env = getenv("SPMV_DIST_DIRECT_PUSH");
    const bool direct_push = !direct_env || atoi(direct_env) != 0;
    for (int q = 0; q < P; q++) {
        SPMV_HIP_TRY(hipSetDevice(d->device[q]), return 1);
        if (spmv_hip_rowblock_set_send_list(d->block[q], (int)send_list[q].size(), send_list[q].data())) return 1;
        SPMV_HIP_TRY(hipMalloc((void **)&d->d_send[q], es * std::max<size_t>(send_list[q].size(), 1)), return 1);
        // one store kernel instead of pack + a copy per consumer, where every consumer can be written directly
        bool all_direct = direct_push;
        std::vector<int> begin;
        std::vector<void *> dst;
        for (int p = 0; p < P && all_direct; p++) {
            const int count = d->send_off[q][p + 1] - d->send_off[q][p];
            if (p == q || count == 0) continue;
            all_direct = d->peer[(size_t)q * P + p] != 0;
            begin.push_back(d->send_off[q][p]);
            dst.push_back(d->d_x[p] + ((size_t)d->block[p]->rows + (size_t)d->recv_off[p][q]) * es);
        }
        if (all_direct && !dst.empty() && (int)dst.size() <= kMaxPushSegments) {
            begin.push_back((int)send_list[q].size());
            d->push_nseg[q] = (int)dst.size();
            d->push_begin[q] = begin;
            d->push_dst[q] = dst;
        }
    }
    const char *threads = getenv("SPMV_DIST_THREADS");
    if (!threads || atoi(threads) != 0) {
        d->sent_seq.reset(new std::atomic<uint64_t>[(size_t)P]);
        for (int p = 0; p < P; p++) d->sent_seq[p].store(0);
        d->threaded = true;
        try {
            for (int p = 0; p < P; p++) d->helper.emplace_back(helper_main, d, p);
        } catch (...) {
            dist_stop_helpers(d);
            set_error("dist: could not start the helper threads");
            return 1;
        }
    }
    return 0;
}

} // namespace spmv

extern "C" {

int spmv_hip_dist_free(spmv_hip_dist *d)
{
    if (!d) return 0;
    dist_stop_helpers(d);
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (int p = 0; p < (int)d->device.size(); p++) {
        (void)hipSetDevice(d->device[p]);
        if (p < (int)d->block.size() && d->block[p]) spmv_hip_rowblock_free(d->block[p]);
        if (p < (int)d->d_x.size() && d->d_x[p]) (void)hipFree(d->d_x[p]);
        if (p < (int)d->d_y.size() && d->d_y[p]) (void)hipFree(d->d_y[p]);
        if (p < (int)d->d_send.size() && d->d_send[p]) (void)hipFree(d->d_send[p]);
        if (p < (int)d->d_norm.size() && d->d_norm[p]) (void)hipFree(d->d_norm[p]);
        if (p < (int)d->d_partial.size() && d->d_partial[p]) (void)hipFree(d->d_partial[p]);
        if (p < (int)d->compute.size() && d->compute[p]) (void)hipStreamDestroy(d->compute[p]);
        if (p < (int)d->comm.size() && d->comm[p]) (void)hipStreamDestroy(d->comm[p]);
        if (p < (int)d->x_ready.size() && d->x_ready[p]) (void)hipEventDestroy(d->x_ready[p]);
        if (p < (int)d->sent.size() && d->sent[p]) (void)hipEventDestroy(d->sent[p]);
    }
    if (d->h_norm) (void)hipHostFree(d->h_norm);
    if (have_prev) (void)hipSetDevice(prev);
    delete d;
    return 0;
}

// on failure: frees the handle (and the blocks it holds unless `keep_blocks`), keeps the message
static spmv_hip_dist *dist_abandon(spmv_hip_dist *d, bool keep_blocks)
{
    char keep[512];
    snprintf(keep, sizeof keep, "%s", spmv_hip_last_error());
    if (keep_blocks) std::fill(d->block.begin(), d->block.end(), nullptr);
    spmv_hip_dist_free(d);
    set_error("%s", keep);
    return nullptr;
}

spmv_hip_dist *spmv_hip_dist_create(int format, int rows, int columns, int nnz, struct mtx *coo, int dtype, int64_t p0,
                                    int64_t p1, int ngpus, const int *devices)
{
    clear_error();
    if (rows != columns) { set_error("dist: the partitioned product needs a square matrix (%d x %d)", rows, columns); return nullptr; }
    if (rows < 0 || nnz < 0 || (nnz > 0 && !coo)) { set_error("dist: bad argument"); return nullptr; }
    if (ngpus < 1 || ngpus > 64) { set_error("dist: ngpus %d out of range", ngpus); return nullptr; }
    if (dtype != SPMV_HIP_F64 && dtype != SPMV_HIP_F32) { set_error("unknown dtype %d", dtype); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available"); return nullptr; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    spmv_hip_dist *d = new (std::nothrow) spmv_hip_dist();
    if (!d) { set_error("out of host memory"); return nullptr; }
    d->ngpus = ngpus; d->dtype = dtype; d->rows = rows; d->nnz = nnz;
    const int P = ngpus;
    d->device.resize(P);
    for (int p = 0; p < P; p++) {
        d->device[p] = devices ? devices[p] : p;
        if (d->device[p] < 0 || d->device[p] >= ndev) {
            set_error("dist: device %d of block %d does not exist (%d visible)", d->device[p], p, ndev);
            delete d;
            return nullptr;
        }
    }
    d->block.assign(P, nullptr);
    const size_t es = dtype_size(dtype);
    int rc = 1;
    do {
        // host side, as every creator: (row, col) sort in place, then row pointers
        spmv_hip_sort_coo(nnz, coo);
        std::vector<int> rp((size_t)rows + 1, 0);
        bool ok = true;
        for (int l = 0; l < nnz && ok; l++) {
            ok = coo[l].i >= 0 && coo[l].i < rows && coo[l].j >= 0 && coo[l].j < columns;
            if (ok) rp[(size_t)coo[l].i + 1]++;
        }
        if (!ok) { set_error("dist: coordinate out of range"); break; }
        for (int r = 0; r < rows; r++) rp[(size_t)r + 1] += rp[r];
        d->bounds.resize((size_t)P + 1);
        // boundaries on multiples of 1024 rows: SELL sorting windows and slices never straddle two blocks
        if (spmv_hip_partition_rows(rows, rp.data(), P, 1024, d->bounds.data())) break;
        for (int p = 0; p < P && ok; p++) {
            ok = false;
            SPMV_HIP_TRY(hipSetDevice(d->device[p]), break);
            const int b = d->bounds[p], e = d->bounds[p + 1], n = e - b;
            const int lo = rp[b], cnt = rp[e] - lo;
            std::vector<int> lrp((size_t)n + 1), lcol((size_t)std::max(cnt, 1));
            for (int r = 0; r <= n; r++) lrp[r] = rp[(size_t)b + r] - lo;
            for (int l = 0; l < cnt; l++) lcol[l] = coo[(size_t)lo + l].j;
            int *d_rp = nullptr, *d_col = nullptr;
            char *d_val = nullptr;
            bool up = hipMalloc((void **)&d_rp, sizeof(int) * ((size_t)n + 1)) == hipSuccess &&
                      hipMalloc((void **)&d_col, sizeof(int) * (size_t)std::max(cnt, 1)) == hipSuccess &&
                      hipMalloc((void **)&d_val, es * (size_t)std::max(cnt, 1)) == hipSuccess &&
                      hipMemcpy(d_rp, lrp.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice) == hipSuccess &&
                      (cnt == 0 || hipMemcpy(d_col, lcol.data(), sizeof(int) * (size_t)cnt, hipMemcpyHostToDevice) == hipSuccess);
            if (up && cnt > 0) {
                if (dtype == SPMV_HIP_F64) {
                    std::vector<double> v((size_t)cnt);
                    for (int l = 0; l < cnt; l++) v[l] = coo[(size_t)lo + l].a;
                    up = hipMemcpy(d_val, v.data(), sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice) == hipSuccess;
                } else {
                    std::vector<float> v((size_t)cnt);
                    for (int l = 0; l < cnt; l++) v[l] = (float)coo[(size_t)lo + l].a;
                    up = hipMemcpy(d_val, v.data(), sizeof(float) * (size_t)cnt, hipMemcpyHostToDevice) == hipSuccess;
                }
            }
            if (up) d->block[p] = spmv_hip_rowblock_create(format, n, b, columns, cnt, d_rp, d_col, d_val, dtype, p0, p1, nullptr);
            else set_error("dist: uploading block %d failed", p);
            if (d_rp) (void)hipFree(d_rp);
            if (d_col) (void)hipFree(d_col);
            if (d_val) (void)hipFree(d_val);
            if (!d->block[p]) break;
            ok = true;
        }
        if (!ok) break;
        if (dist_finish(d)) break;
        rc = 0;
    } while (0);
    (void)hipSetDevice(prev);
    return rc ? dist_abandon(d, false) : d;
}

spmv_hip_dist *spmv_hip_dist_create_from_blocks(int ngpus, spmv_hip_rowblock **blocks)
{
    clear_error();
    if (ngpus < 1 || ngpus > 64 || !blocks) { set_error("dist: bad argument"); return nullptr; }
    for (int p = 0; p < ngpus; p++)
        if (!blocks[p]) { set_error("dist: block %d is NULL", p); return nullptr; }
    const int columns = blocks[0]->columns_global;
    int64_t nnz = 0;
    for (int p = 0; p < ngpus; p++) {
        const spmv_hip_rowblock *b = blocks[p];
        const int64_t expect = p ? (int64_t)blocks[p - 1]->row_begin + blocks[p - 1]->rows : 0;
        if (b->row_begin != expect || b->columns_global != columns || b->dtype != blocks[0]->dtype) {
            set_error("dist: block %d (rows from %d, %d columns, dtype %d) does not continue the blocks before it (next row %lld, "
                      "%d columns, dtype %d)", p, b->row_begin, b->columns_global, b->dtype, (long long)expect, columns,
                      blocks[0]->dtype);
            return nullptr;
        }
        nnz += b->nnz;
    }
    if ((int64_t)blocks[ngpus - 1]->row_begin + blocks[ngpus - 1]->rows != columns) {
        set_error("dist: the blocks end at row %lld of %d", (long long)blocks[ngpus - 1]->row_begin + blocks[ngpus - 1]->rows, columns);
        return nullptr;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    spmv_hip_dist *d = new (std::nothrow) spmv_hip_dist();
    if (!d) { set_error("out of host memory"); return nullptr; }
    d->ngpus = ngpus; d->dtype = blocks[0]->dtype; d->rows = columns; d->nnz = nnz;
    d->device.resize(ngpus);
    d->bounds.resize((size_t)ngpus + 1);
    d->block.assign(blocks, blocks + ngpus);
    for (int p = 0; p < ngpus; p++) {
        d->device[p] = blocks[p]->device;
        d->bounds[p] = blocks[p]->row_begin;
    }
    d->bounds[ngpus] = columns;
    const int rc = dist_finish(d);
    (void)hipSetDevice(prev);
    return rc ? dist_abandon(d, true) : d;       // after a failure the blocks stay the caller's
}

int spmv_hip_dist_mult(spmv_hip_dist *d, const double *x, double *y)
{
    clear_error();
    if (!d || !x || !y) { set_error("NULL argument"); return 1; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = dist_upload_x(d, x) || dist_run(d, kOpProductSync) || dist_download(d, d->d_y, y);
    (void)hipSetDevice(prev);
    return rc;
}

int spmv_hip_dist_bench(spmv_hip_dist *d, int min_iters, double min_seconds, double *x, int *iters, double *seconds)
{
    clear_error();
    if (!d || !x || !iters || !seconds) { set_error("NULL argument"); return 1; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = 1, it = 0;
    double total = 0.0;
    do {
        if (dist_upload_x(d, x) || dist_run(d, kOpSync)) break;
        const auto t0 = std::chrono::steady_clock::now();
        bool failed = false;
        for (;;) {
            it++;
            // timed: the exchange and the products of all blocks, from the first launch to the last
            // block's completion (driver_spmv.c:297-300 times the one mult call the same way)
            const auto t1 = std::chrono::steady_clock::now();
            if (dist_run(d, kOpProductSync)) { failed = true; break; }
            const auto t2 = std::chrono::steady_clock::now();
            total += std::chrono::duration<double>(t2 - t1).count();
            // untimed: x = y / ||y||_2 with the norm over all blocks (driver_spmv.c:301-302)
            if (dist_run(d, kOpNormScale)) { failed = true; break; }
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (wall > min_seconds && it >= min_iters) break;
        }
        if (failed) { if (!spmv_hip_last_error()[0]) set_error("dist_bench: a device call failed"); break; }
        if (dist_download(d, d->d_x, x)) break;
        *iters = it;
        *seconds = total;
        rc = 0;
    } while (0);
    (void)hipSetDevice(prev);
    return rc;
}

int spmv_hip_dist_issue_time(spmv_hip_dist *d, int products, double *issue_seconds, double *total_seconds)
{
    clear_error();
    if (!d || products < 1 || !issue_seconds || !total_seconds) { set_error("dist_issue_time: bad argument"); return 1; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    double issue = 0.0, total = 0.0;
    int rc = dist_run(d, kOpSync);
    for (int k = 0; k < products && !rc; k++) {
        const auto t1 = std::chrono::steady_clock::now();
        rc = dist_run(d, kOpProduct);                    // returns once every block's launches are issued
        const auto t2 = std::chrono::steady_clock::now();
        rc = rc || dist_run(d, kOpSync);
        const auto t3 = std::chrono::steady_clock::now();
        issue += std::chrono::duration<double>(t2 - t1).count();
        total += std::chrono::duration<double>(t3 - t1).count();
    }
    (void)hipSetDevice(prev);
    *issue_seconds = issue;
    *total_seconds = total;
    return rc;
}

int spmv_hip_dist_set_option(spmv_hip_dist *d, const char *name, int64_t value)
{
    if (!d) { set_error("NULL handle"); return 1; }
    for (int p = 0; p < d->ngpus; p++)
        if (spmv_hip_rowblock_set_option(d->block[(size_t)p], name, value)) return 1;
    return 0;
}

int spmv_hip_dist_analyze(spmv_hip_dist *d, int reps)
{
    clear_error();
    if (!d) { set_error("NULL handle"); return 1; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    const int rc = dist_run(d, kOpAnalyze, reps);
    (void)hipSetDevice(prev);
    return rc;
}

int spmv_hip_dist_get_info(const spmv_hip_dist *d, int *ngpus, int *bounds, spmv_hip_rowblock_info *blocks)
{
    if (!d) { set_error("NULL handle"); return 1; }
    if (ngpus) *ngpus = d->ngpus;
    if (bounds)
        for (int p = 0; p <= d->ngpus; p++) bounds[p] = d->bounds[(size_t)p];
    if (blocks)
        for (int p = 0; p < d->ngpus; p++)
            if (spmv_hip_rowblock_get_info(d->block[(size_t)p], &blocks[p])) return 1;
    return 0;
}

int spmv_hip_dist_get_peer_access(const spmv_hip_dist *d, int *access)
{
    if (!d || !access) { set_error("NULL argument"); return 1; }
    for (size_t k = 0; k < d->peer.size(); k++) access[k] = d->peer[k];
    return 0;
}

} // extern "C"
