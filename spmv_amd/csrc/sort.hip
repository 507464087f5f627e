// (row, col) sort of an AoS COO array on the device.
//
// The reference sorts every unsorted matrix with libc qsort on the host
// (sort_mtx + by_row_mtx, reference src/mtx.c:9-34) -- and MatrixMarket files
// come column-major, so that is every file: ~10 s per 100 M entries.  Here the
// entries are packed into 64-bit keys (row in the high word, column in the low
// word, sign bits flipped so that unsigned key order equals the signed (i, j)
// order), radix-sorted together with their values by rocPRIM, and unpacked into
// the caller's array.  The sort is stable, so duplicate coordinates keep their
// file order (qsort leaves that unspecified).

#include "internal.h"

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace spmv {

__global__ void coo_pack_kernel(int64_t n, const struct mtx *__restrict__ coo, uint64_t *__restrict__ key,
                                double *__restrict__ val)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const struct mtx e = coo[k];
    key[k] = ((uint64_t)((uint32_t)e.i ^ 0x80000000u) << 32) | ((uint32_t)e.j ^ 0x80000000u);
    val[k] = e.a;
}

__global__ void coo_unpack_kernel(int64_t n, const uint64_t *__restrict__ key, const double *__restrict__ val,
                                  struct mtx *__restrict__ coo)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    struct mtx e;
    e.i = (int)((uint32_t)(key[k] >> 32) ^ 0x80000000u);
    e.j = (int)((uint32_t)key[k] ^ 0x80000000u);
    e.a = val[k];
    coo[k] = e;
}

// Returns 0 when `coo` has been sorted; non-zero (message set) when the device path could not run,
// in which case `coo` is untouched.
int device_sort_coo(int64_t nnz, struct mtx *coo)
{
    if (nnz < 2) return 0;
    struct mtx *d_coo = nullptr;
    uint64_t *d_key[2] = {nullptr, nullptr};
    double *d_val[2] = {nullptr, nullptr};
    void *d_temp = nullptr;
    int rc = 1;
    do {
        const size_t n = (size_t)nnz;
        SPMV_HIP_TRY(hipMalloc((void **)&d_coo, n * sizeof(struct mtx)), break);
        for (int b = 0; b < 2; b++) {
            SPMV_HIP_TRY(hipMalloc((void **)&d_key[b], n * sizeof(uint64_t)), break);
            SPMV_HIP_TRY(hipMalloc((void **)&d_val[b], n * sizeof(double)), break);
        }
        if (!d_val[1]) break;
        SPMV_HIP_TRY(hipMemcpy(d_coo, coo, n * sizeof(struct mtx), hipMemcpyHostToDevice), break);
        const dim3 grid((unsigned)((nnz + 255) / 256)), block(256);
        hipLaunchKernelGGL(coo_pack_kernel, grid, block, 0, nullptr, nnz, d_coo, d_key[0], d_val[0]);
        SPMV_HIP_TRY(hipGetLastError(), break);
        size_t temp_bytes = 0;
        SPMV_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n, 0, 64,
                                               (hipStream_t) nullptr), break);
        SPMV_HIP_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), break);
        SPMV_HIP_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n, 0, 64,
                                               (hipStream_t) nullptr), break);
        hipLaunchKernelGGL(coo_unpack_kernel, grid, block, 0, nullptr, nnz, d_key[1], d_val[1], d_coo);
        SPMV_HIP_TRY(hipGetLastError(), break);
        SPMV_HIP_TRY(hipMemcpy(coo, d_coo, n * sizeof(struct mtx), hipMemcpyDeviceToHost), break);
        rc = 0;
    } while (0);
    if (d_temp) (void)hipFree(d_temp);
    for (int b = 0; b < 2; b++) {
        if (d_key[b]) (void)hipFree(d_key[b]);
        if (d_val[b]) (void)hipFree(d_val[b]);
    }
    if (d_coo) (void)hipFree(d_coo);
    return rc;
}

// ---- COO -> CSR in HBM (create_csr, reference csr.c:160-203, alignment 1) ----

// flags[0]: some entry precedes its predecessor in (row, col) order; [1]: a row index outside [0, rows);
// [2]: a column index outside [0, columns)
__global__ void coo_check_kernel(int64_t n, int rows, int columns, const struct mtx *__restrict__ coo,
                                 int *__restrict__ flags)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const struct mtx e = coo[k];
    if (e.i < 0 || e.i >= rows) flags[1] = 1;
    if (e.j < 0 || e.j >= columns) flags[2] = 1;
    if (k > 0) {
        const struct mtx p = coo[k - 1];
        if (p.i > e.i || (p.i == e.i && p.j > e.j)) flags[0] = 1;
    }
}

// row_ptr[r] = number of entries with row index < r (entries sorted by row)
__global__ void coo_row_ptr_kernel(int rows, int64_t n, const struct mtx *__restrict__ coo, int *__restrict__ row_ptr)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (coo[mid].i < r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = (int)lo;
}

template <typename T>
__global__ void coo_split_kernel(int64_t n, const struct mtx *__restrict__ coo, int *__restrict__ col, T *__restrict__ val)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    col[k] = coo[k].j;
    val[k] = (T)coo[k].a;
}

// Sorts `coo` (host array, in place, as every creator does) and builds its CSR arrays in HBM.
// Returns 0 with three hipMalloc'ed arrays, 1 on an input error (message set), 2 when the device
// path could not run (nothing changed; the caller falls back to the host conversion).
int device_coo_to_csr(int rows, int columns, int64_t nnz, struct mtx *coo, bool values_f64, int **out_rp, int **out_cj,
                      void **out_va)
{
    const size_t n = (size_t)nnz;
    struct mtx *d_coo = nullptr;
    uint64_t *d_key[2] = {nullptr, nullptr};
    double *d_val[2] = {nullptr, nullptr};
    void *d_temp = nullptr;
    int *d_flags = nullptr, *d_rp = nullptr, *d_cj = nullptr;
    void *d_va = nullptr;
    int rc = 2;
    do {
        if (hipMalloc((void **)&d_coo, n * sizeof(struct mtx)) != hipSuccess || hipMalloc((void **)&d_flags, 3 * sizeof(int)) != hipSuccess) break;
        if (hipMemcpy(d_coo, coo, n * sizeof(struct mtx), hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemset(d_flags, 0, 3 * sizeof(int)) != hipSuccess) break;
        const dim3 grid((unsigned)((nnz + 255) / 256)), block(256);
        hipLaunchKernelGGL(coo_check_kernel, grid, block, 0, nullptr, nnz, rows, columns, d_coo, d_flags);
        int flags[3] = {0, 0, 0};
        if (hipMemcpy(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (flags[1]) { set_error("coo row index out of range [0, %d)", rows); rc = 1; break; }
        if (flags[2]) { set_error("column index out of range [0, %d)", columns); rc = 1; break; }
        if (flags[0]) {                                       // not in (row, col) order: radix sort, hand the order back
            bool ok = true;
            for (int b = 0; b < 2 && ok; b++)
                ok = hipMalloc((void **)&d_key[b], n * sizeof(uint64_t)) == hipSuccess &&
                     hipMalloc((void **)&d_val[b], n * sizeof(double)) == hipSuccess;
            if (!ok) break;
            hipLaunchKernelGGL(coo_pack_kernel, grid, block, 0, nullptr, nnz, d_coo, d_key[0], d_val[0]);
            size_t temp_bytes = 0;
            if (rocprim::radix_sort_pairs(nullptr, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n, 0, 64,
                                          (hipStream_t) nullptr) != hipSuccess) break;
            if (hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)) != hipSuccess) break;
            if (rocprim::radix_sort_pairs(d_temp, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n, 0, 64,
                                          (hipStream_t) nullptr) != hipSuccess) break;
            hipLaunchKernelGGL(coo_unpack_kernel, grid, block, 0, nullptr, nnz, d_key[1], d_val[1], d_coo);
            if (hipMemcpy(coo, d_coo, n * sizeof(struct mtx), hipMemcpyDeviceToHost) != hipSuccess) break;
        }
        const size_t vbytes = values_f64 ? sizeof(double) : sizeof(float);
        if (hipMalloc((void **)&d_rp, ((size_t)rows + 1) * sizeof(int)) != hipSuccess ||
            hipMalloc((void **)&d_cj, n * sizeof(int)) != hipSuccess || hipMalloc(&d_va, n * vbytes) != hipSuccess) break;
        hipLaunchKernelGGL(coo_row_ptr_kernel, dim3((unsigned)((rows + 256) / 256)), block, 0, nullptr, rows, nnz, d_coo, d_rp);
        if (values_f64)
            hipLaunchKernelGGL((coo_split_kernel<double>), grid, block, 0, nullptr, nnz, d_coo, d_cj, (double *)d_va);
        else
            hipLaunchKernelGGL((coo_split_kernel<float>), grid, block, 0, nullptr, nnz, d_coo, d_cj, (float *)d_va);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
        *out_rp = d_rp; *out_cj = d_cj; *out_va = d_va;
        d_rp = nullptr; d_cj = nullptr; d_va = nullptr;
        rc = 0;
    } while (0);
    void *scratch[] = {d_coo, d_key[0], d_key[1], d_val[0], d_val[1], d_temp, d_flags, d_rp, d_cj, d_va};
    for (void *p : scratch)
        if (p) (void)hipFree(p);
    return rc;
}

// out[i] = in[0] + ... + in[i-1] on `st` (rocPRIM device scan; its temporary storage is freed before return)
int device_exclusive_scan(const int *d_in, int *d_out, size_t n, hipStream_t st)
{
    if (n == 0) return 0;
    size_t temp_bytes = 0;
    void *d_temp = nullptr;
    SPMV_HIP_TRY(rocprim::exclusive_scan(nullptr, temp_bytes, d_in, d_out, 0, n, rocprim::plus<int>(), st), return 1);
    SPMV_HIP_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), return 1);
    int rc = 0;
    SPMV_HIP_TRY(rocprim::exclusive_scan(d_temp, temp_bytes, d_in, d_out, 0, n, rocprim::plus<int>(), st), rc = 1);
    if (!rc) SPMV_HIP_TRY(hipStreamSynchronize(st), rc = 1);
    (void)hipFree(d_temp);
    return rc;
}

// ---- columns relabeled by how often the matrix references them ----
//
// Every gather of x that misses the L2 costs a 128-byte request to the fabric whatever the size of
// the element, and on matrices with power-law column counts (graphs, circuits) a few hundred thousand
// columns take most of the references while lying scattered over the whole index range.  Numbering
// the columns in descending order of their reference count packs those into the first megabytes of a
// permuted copy of x, which then stay in L2.  The entries keep their places in the rows, so every
// row is summed in the same order as before and the product is bit-identical to the one without
// relabeling.

__global__ void column_count_kernel(int64_t n, const int *__restrict__ col, unsigned *__restrict__ count)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) atomicAdd(&count[col[k]], 1u);
}

__global__ void iota_kernel(int n, int *__restrict__ out)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = k;
}

__global__ void invert_permutation_kernel(int n, const int *__restrict__ old_of_new, int *__restrict__ new_of_old)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) new_of_old[old_of_new[k]] = k;
}

__global__ void relabel_columns_kernel(int64_t n, const int *__restrict__ col, const int *__restrict__ new_of_old,
                                       int *__restrict__ out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = new_of_old[col[k]];
}

// Fills m->d_hot_old (column of x behind each new label), m->d_hot_col (the col array relabeled) and
// m->d_hot_x (room for the permuted x).  0 on success; on failure nothing is left allocated.
int build_hot_columns(spmv_hip_matrix *m, hipStream_t st)
{
    const int cols = m->columns;
    const int64_t n = m->stored;
    unsigned *d_count[2] = {nullptr, nullptr};
    int *d_id = nullptr, *d_new_of_old = nullptr;
    void *d_temp = nullptr;
    int *hot_old = nullptr, *hot_col = nullptr;
    char *hot_x = nullptr;
    int64_t bytes_before = m->device_bytes;
    int rc = 1;
    do {
        if (cols <= 0 || n <= 0) { set_error("hot_columns: empty matrix"); break; }
        const size_t c = (size_t)cols;
        SPMV_HIP_TRY(hipMalloc((void **)&d_count[0], c * sizeof(unsigned)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_count[1], c * sizeof(unsigned)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_id, c * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_new_of_old, c * sizeof(int)), break);
        if (dev_alloc(m, &hot_old, c) || dev_alloc(m, &hot_col, (size_t)n) || dev_alloc(m, &hot_x, c * dtype_size(m->dtype))) break;
        const dim3 block(256), gcol((unsigned)((c + 255) / 256)), gnz((unsigned)((n + 255) / 256));
        SPMV_HIP_TRY(hipMemsetAsync(d_count[0], 0, c * sizeof(unsigned), st), break);
        hipLaunchKernelGGL(column_count_kernel, gnz, block, 0, st, n, m->d_col, d_count[0]);
        hipLaunchKernelGGL(iota_kernel, gcol, block, 0, st, cols, d_id);
        // stable descending sort: columns with equal counts keep their ascending order
        size_t temp_bytes = 0;
        SPMV_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, temp_bytes, d_count[0], d_count[1], d_id, hot_old, c, 0, 32, st), break);
        SPMV_HIP_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), break);
        SPMV_HIP_TRY(rocprim::radix_sort_pairs_desc(d_temp, temp_bytes, d_count[0], d_count[1], d_id, hot_old, c, 0, 32, st), break);
        hipLaunchKernelGGL(invert_permutation_kernel, gcol, block, 0, st, cols, hot_old, d_new_of_old);
        hipLaunchKernelGGL(relabel_columns_kernel, gnz, block, 0, st, n, m->d_col, d_new_of_old, hot_col);
        SPMV_HIP_TRY(hipGetLastError(), break);
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        m->d_hot_old = hot_old; m->d_hot_col = hot_col; m->d_hot_x = hot_x;
        hot_old = nullptr; hot_col = nullptr; hot_x = nullptr;
        rc = 0;
    } while (0);
    void *scratch[] = {d_count[0], d_count[1], d_id, d_new_of_old, d_temp, hot_old, hot_col, hot_x};
    for (void *p : scratch)
        if (p) (void)hipFree(p);
    if (rc) m->device_bytes = bytes_before;
    return rc;
}

// ---- the same, with the columns dealt over the eight XCDs ----
//
// With one ranking for the whole chip every XCD's L2 ends up holding the same hottest columns.  Here
// the ranking is dealt over eight groups -- the first 65536 ranks one at a time, the rest in runs of
// 16 columns (group g owns the labels g*cpx .. g*cpx+cpx-1, cpx = columns/8 rounded up to whole runs),
// so each group owns a contiguous range of the permuted x with an equal share of the references and
// its own hottest columns in front; the matrix is split by group into eight row-aligned pieces that
// are stacked into one CSR of 8*rows "virtual" rows (piece g holds virtual rows g*rows..g*rows+rows-1).
// The balanced kernel walks that CSR in storage order and its workgroup ids are mapped so that each
// XCD takes one contiguous eighth of the tiles -- piece g, give or take a tile -- so the L2 of XCD g
// only ever sees the columns of group g: eight times the cache for the hot columns.  A last pass
// adds the eight partial results of every row in group order.

constexpr int kXcds = 8;

__global__ void xcd_label_kernel(int cols, int cpx, int snake, const int *__restrict__ old_of_rank,
                                 int *__restrict__ label_of_old, int *__restrict__ old_of_label)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cols) return;
    int g, pos;
    if (k < snake) {
        // the hottest ranks one at a time, back and forth over the groups (0..7, 7..0, ...): a run of 16 of
        // them would hand one group several per cent more references than the others, and the groups
        // run side by side, one per XCD
        const int round = k / kXcds, at = k % kXcds;
        g = round & 1 ? kXcds - 1 - at : at;
        pos = round;
    } else {
        // the rest in runs of 16 (one 128-byte line of doubles), so that columns which were neighbours in
        // x and stayed neighbours in the ranking (equal counts) still share a line
        const int rest = k - snake, run = rest / 16;
        g = run % kXcds;
        pos = snake / kXcds + (run / kXcds) * 16 + rest % 16;
    }
    const int label = g * cpx + pos;
    const int old = old_of_rank[k];
    label_of_old[old] = label;
    old_of_label[label] = old;
}

// Rows of up to kXcdLongRow entries are walked by one lane each; a longer row (a power-law matrix has rows of
// several hundred thousand entries, and one lane walking such a row alone takes longer than the rest of
// the build together) is walked by the whole wave, 64 entries per step with coalesced loads.
constexpr int kXcdLongRow = 96;

__device__ __forceinline__ int xcd_group_of(int label, int cpx)
{
    int g = 0;
#pragma unroll
    for (int k = 1; k < kXcds; k++) g += label >= k * cpx;
    return g;
}

// A row beyond kXcdHugeRow entries is cut into segments of kXcdSegment entries, each walked by a wave of its own
// (xcd_count_huge_kernel, xcd_fill_huge_kernel): the first 64 rows of an R-MAT matrix hold millions of entries
// between them, and a single wave walking them one after the other would be the whole build's critical path.
constexpr int kXcdHugeRow = 8192, kXcdSegment = 4096;

// cnt[g*rows + r] = entries of row r whose column belongs to group g; huge rows are only listed here
// (huge_rows[0 .. meta[0]), longest of them in meta[1]) and counted by xcd_count_huge_kernel
__global__ __launch_bounds__(256) void xcd_count_kernel(int rows, int cpx, const int *__restrict__ row_ptr,
                                                        const int *__restrict__ col,
                                                        const int *__restrict__ label_of_old, int *__restrict__ cnt,
                                                        int *__restrict__ huge_rows, int *__restrict__ meta)
{
    __shared__ int c[kXcds][256];              // the eight counters of a lane, one bank per lane
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (int g = 0; g < kXcds; g++) c[g][threadIdx.x] = 0;
    const int rs = r < rows ? row_ptr[r] : 0, re = r < rows ? row_ptr[r + 1] : 0;
    const bool is_huge = re - rs > kXcdHugeRow;
    const bool is_long = re - rs > kXcdLongRow && !is_huge;
    if (is_huge) {
        huge_rows[atomicAdd(&meta[0], 1)] = r;
        atomicMax(&meta[1], re - rs);
    }
    if (!is_long && !is_huge)
        for (int e = rs; e < re; e++) c[xcd_group_of(label_of_old[col[e]], cpx)][threadIdx.x]++;
    unsigned long long longs = __ballot(is_long);
    while (longs) {
        const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(longs));
        longs &= longs - 1;
        const int ls = __builtin_amdgcn_readlane(rs, src), le = __builtin_amdgcn_readlane(re, src);
        int mine[kXcds] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int e = ls + lane; e < le; e += 64) {
            const int g = xcd_group_of(label_of_old[col[e]], cpx);
#pragma unroll
            for (int k = 0; k < kXcds; k++) mine[k] += g == k;
        }
#pragma unroll
        for (int k = 0; k < kXcds; k++) {
            int t = mine[k];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
            if (lane == src) c[k][threadIdx.x] = t;
        }
    }
    if (r < rows && !is_huge)
        for (int g = 0; g < kXcds; g++) cnt[(size_t)g * rows + r] = c[g][threadIdx.x];
}

__global__ void xcd_zero_rows_kernel(int rows, int n, const int *__restrict__ which, int *__restrict__ cnt)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int g = 0; g < kXcds; g++) cnt[(size_t)g * rows + which[k]] = 0;
}

// One wave per (huge row, segment), blockIdx.x = row index * nseg + segment: seg_cnt[blockIdx.x * 8 + g] = entries of
// the segment in group g, also added to the row's total in cnt (zeroed by the caller for these rows).
__global__ __launch_bounds__(64) void xcd_count_huge_kernel(int rows, int cpx, int nseg, const int *__restrict__ row_ptr,
                                                            const int *__restrict__ col,
                                                            const int *__restrict__ label_of_old,
                                                            const int *__restrict__ huge_rows, int *__restrict__ seg_cnt,
                                                            int *__restrict__ cnt)
{
    const int which = (int)(blockIdx.x / (unsigned)nseg), seg = (int)(blockIdx.x % (unsigned)nseg);
    const int r = huge_rows[which], lane = threadIdx.x;
    const int rs = row_ptr[r], re = row_ptr[r + 1];
    const int ls = rs + seg * kXcdSegment, le = min(ls + kXcdSegment, re);
    int mine[kXcds] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = ls + lane; e < le; e += 64) {
        const int g = xcd_group_of(label_of_old[col[e]], cpx);
#pragma unroll
        for (int k = 0; k < kXcds; k++) mine[k] += g == k;
    }
#pragma unroll
    for (int k = 0; k < kXcds; k++) {
        int t = mine[k];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) {
            seg_cnt[((size_t)which * nseg + seg) * kXcds + k] = t;
            if (t) atomicAdd(&cnt[(size_t)k * rows + r], t);
        }
    }
}

// The entries of every row dealt to the row's eight virtual rows, in their order inside the row.
template <typename TIN, typename T>
__global__ __launch_bounds__(256) void xcd_fill_kernel(int rows, int cpx, const int *__restrict__ row_ptr,
                                                       const int *__restrict__ col, const TIN *__restrict__ val,
                                                       const int *__restrict__ label_of_old,
                                                       const int *__restrict__ vrow_ptr, int *__restrict__ vcol,
                                                       T *__restrict__ vval)
{
    __shared__ int cursor[kXcds][256];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int rs = r < rows ? row_ptr[r] : 0, re = r < rows ? row_ptr[r + 1] : 0;
    const bool is_huge = re - rs > kXcdHugeRow;                 // filled by xcd_fill_huge_kernel
    const bool is_long = re - rs > kXcdLongRow && !is_huge;
    if (r < rows && !is_long && !is_huge) {
        for (int g = 0; g < kXcds; g++) cursor[g][threadIdx.x] = vrow_ptr[(size_t)g * rows + r];
        for (int e = rs; e < re; e++) {
            const int label = label_of_old[col[e]];
            const int at = cursor[xcd_group_of(label, cpx)][threadIdx.x]++;
            vcol[at] = label;
            vval[at] = (T)val[e];
        }
    }
    unsigned long long longs = __ballot(is_long);
    const unsigned long long below = (1ull << lane) - 1ull;
    while (longs) {
        const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(longs));
        longs &= longs - 1;
        const int ls = __builtin_amdgcn_readlane(rs, src), le = __builtin_amdgcn_readlane(re, src);
        const int row = __builtin_amdgcn_readlane(r, src);
        int next[kXcds];                            // where each group's next entry of this row goes (wave-uniform)
#pragma unroll
        for (int k = 0; k < kXcds; k++) next[k] = vrow_ptr[(size_t)k * rows + row];
        for (int base = ls; base < le; base += 64) {
            const int e = base + lane;
            const bool live = e < le;
            const int label = live ? label_of_old[col[e]] : 0;
            const int g = xcd_group_of(label, cpx);
            int at = 0;
#pragma unroll
            for (int k = 0; k < kXcds; k++) {       // entry order inside the row is kept: lanes of one group rank by lane
                const unsigned long long mask = __ballot(live && g == k);
                if (g == k) at = next[k] + __popcll(mask & below);
                next[k] += __popcll(mask);
            }
            if (live) {
                vcol[at] = label;
                vval[at] = (T)val[e];
            }
        }
    }
}

// One wave per (huge row, segment): the segment's entries go behind those of the row's earlier segments.
template <typename TIN, typename T>
__global__ __launch_bounds__(64) void xcd_fill_huge_kernel(int rows, int cpx, int nseg, const int *__restrict__ row_ptr,
                                                           const int *__restrict__ col, const TIN *__restrict__ val,
                                                           const int *__restrict__ label_of_old,
                                                           const int *__restrict__ huge_rows,
                                                           const int *__restrict__ seg_cnt,
                                                           const int *__restrict__ vrow_ptr, int *__restrict__ vcol,
                                                           T *__restrict__ vval)
{
    const int which = (int)(blockIdx.x / (unsigned)nseg), seg = (int)(blockIdx.x % (unsigned)nseg);
    const int r = huge_rows[which], lane = threadIdx.x;
    const int rs = row_ptr[r], re = row_ptr[r + 1];
    const int ls = rs + seg * kXcdSegment, le = min(ls + kXcdSegment, re);
    if (ls >= le) return;
    const unsigned long long below = (1ull << lane) - 1ull;
    int next[kXcds];
#pragma unroll
    for (int k = 0; k < kXcds; k++) {
        int before = 0;                             // entries of group k in the row's earlier segments
        for (int sgm = lane; sgm < seg; sgm += 64) before += seg_cnt[((size_t)which * nseg + sgm) * kXcds + k];
        for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
        next[k] = vrow_ptr[(size_t)k * rows + r] + before;
    }
    for (int base = ls; base < le; base += 64) {
        const int e = base + lane;
        const bool live = e < le;
        const int label = live ? label_of_old[col[e]] : 0;
        const int g = xcd_group_of(label, cpx);
        int at = 0;
#pragma unroll
        for (int k = 0; k < kXcds; k++) {
            const unsigned long long mask = __ballot(live && g == k);
            if (g == k) at = next[k] + __popcll(mask & below);
            next[k] += __popcll(mask);
        }
        if (live) {
            vcol[at] = label;
            vval[at] = (T)val[e];
        }
    }
}

// Builds the stacked CSR for `m` (a CSR-based handle).  Outputs are hipMalloc'ed: the virtual row
// pointers [8*rows+1], relabeled columns and values [stored], and the column of x behind every label
// [8*cpx], -1 for the few unused labels.  0 on success.
int build_xcd_columns(const spmv_hip_matrix *m, bool val_is_f64, hipStream_t st, int **out_vrp, int **out_vcol,
                      void **out_vval, int **out_old_of_label, int *out_cpx, int *out_used_cpx)
{
    const int cols = m->columns, rows = m->rows;
    const int64_t n = m->stored;
    if (cols <= 0 || n <= 0 || rows <= 0) { set_error("hot_columns: empty matrix"); return 1; }
    if ((int64_t)rows * kXcds > 0x7fffffffLL - 1) { set_error("hot_columns 2: %d rows x 8 exceed 32-bit row numbers", rows); return 1; }
    const int cpx = (int)(((int64_t)cols + 16 * kXcds - 1) / (16 * kXcds)) * 16;
    const size_t c = (size_t)cols, vrows = (size_t)rows * kXcds, labels = (size_t)cpx * kXcds;
    // ranks dealt one at a time: 10240 per group (what a CU's LDS holds beside 16 waves of tile staging), whole lines
    const int snake = std::min(81920, cols / (16 * kXcds) * (16 * kXcds));
    unsigned *d_count[2] = {nullptr, nullptr};
    int *d_id = nullptr, *d_rank_old = nullptr, *d_label_of_old = nullptr, *d_cnt = nullptr;
    int *d_huge = nullptr, *d_meta = nullptr, *d_seg_cnt = nullptr;
    void *d_temp = nullptr;
    int *vrp = nullptr, *vcol = nullptr, *old_of_label = nullptr;
    void *vval = nullptr;
    int rc = 1;
    do {
        SPMV_HIP_TRY(hipMalloc((void **)&d_count[0], c * sizeof(unsigned)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_count[1], c * sizeof(unsigned)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_id, c * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_rank_old, c * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_label_of_old, c * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&old_of_label, labels * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_cnt, (vrows + 1) * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_huge, ((size_t)(n / kXcdHugeRow) + 1) * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&d_meta, 2 * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&vrp, (vrows + 1) * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&vcol, (size_t)n * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc(&vval, (size_t)n * dtype_size(m->dtype)), break);
        const dim3 block(256), gcol((unsigned)((c + 255) / 256)), gnz((unsigned)((n + 255) / 256)),
            grow((unsigned)(((size_t)rows + 255) / 256));
        SPMV_HIP_TRY(hipMemsetAsync(d_count[0], 0, c * sizeof(unsigned), st), break);
        hipLaunchKernelGGL(column_count_kernel, gnz, block, 0, st, n, m->d_col, d_count[0]);
        hipLaunchKernelGGL(iota_kernel, gcol, block, 0, st, cols, d_id);
        size_t temp_bytes = 0;
        SPMV_HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, temp_bytes, d_count[0], d_count[1], d_id, d_rank_old, c, 0, 32, st), break);
        SPMV_HIP_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), break);
        SPMV_HIP_TRY(rocprim::radix_sort_pairs_desc(d_temp, temp_bytes, d_count[0], d_count[1], d_id, d_rank_old, c, 0, 32, st), break);
        // how many columns the matrix references at all: the sorted counts end in zeros, and the labels of
        // those columns (the tail of every group) never need a value of x
        size_t referenced = 0;
        {
            size_t lo = 0, hi = c;
            bool ok = true;
            while (lo < hi && ok) {
                const size_t mid = (lo + hi) / 2;
                unsigned v = 0;
                ok = hipMemcpyAsync(&v, d_count[1] + mid, sizeof v, hipMemcpyDeviceToHost, st) == hipSuccess &&
                     hipStreamSynchronize(st) == hipSuccess;
                if (v > 0) lo = mid + 1; else hi = mid;
            }
            if (!ok) { set_error("hot_columns 2: reading the sorted counts failed"); break; }
            referenced = lo;
        }
        // positions in use per group: the snake part holds 8 ranks per position, the rest 128 per 16 positions
        if (referenced <= (size_t)snake)
            *out_used_cpx = (int)((referenced + kXcds - 1) / kXcds);
        else
            *out_used_cpx = (int)std::min<size_t>(snake / kXcds + ((referenced - snake + 16 * kXcds - 1) / (16 * kXcds)) * 16, (size_t)cpx);
        SPMV_HIP_TRY(hipMemsetAsync(old_of_label, 0xFF, labels * sizeof(int), st), break);
        hipLaunchKernelGGL(xcd_label_kernel, gcol, block, 0, st, cols, cpx, snake, d_rank_old, d_label_of_old, old_of_label);
        SPMV_HIP_TRY(hipMemsetAsync(d_meta, 0, 2 * sizeof(int), st), break);
        hipLaunchKernelGGL(xcd_count_kernel, grow, block, 0, st, rows, cpx, m->d_row_ptr, m->d_col, d_label_of_old, d_cnt,
                           d_huge, d_meta);
        SPMV_HIP_TRY(hipMemsetAsync(d_cnt + vrows, 0, sizeof(int), st), break);
        SPMV_HIP_TRY(hipGetLastError(), break);
        // the huge rows, a wave per segment
        int meta[2] = {0, 0};
        SPMV_HIP_TRY(hipMemcpyAsync(meta, d_meta, sizeof meta, hipMemcpyDeviceToHost, st), break);
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        const int nhuge = meta[0], nseg = (meta[1] + kXcdSegment - 1) / kXcdSegment;
        if (nhuge > 0) {
            if ((int64_t)nhuge * nseg > 0x7fffffffLL / kXcds) {
                set_error("hot_columns 2: %d rows of more than %d entries, the longest %d", nhuge, kXcdHugeRow, meta[1]);
                break;
            }
            SPMV_HIP_TRY(hipMalloc((void **)&d_seg_cnt, (size_t)nhuge * nseg * kXcds * sizeof(int)), break);
            hipLaunchKernelGGL(xcd_zero_rows_kernel, dim3((unsigned)((nhuge + 255) / 256)), block, 0, st, rows, nhuge, d_huge, d_cnt);
            hipLaunchKernelGGL(xcd_count_huge_kernel, dim3((unsigned)(nseg * nhuge)), dim3(64), 0, st, rows, cpx, nseg,
                               m->d_row_ptr, m->d_col, d_label_of_old, d_huge, d_seg_cnt, d_cnt);
            SPMV_HIP_TRY(hipGetLastError(), break);
        }
        if (device_exclusive_scan(d_cnt, vrp, vrows + 1, st)) break;
        if (m->dtype == SPMV_HIP_F32 && val_is_f64)
            hipLaunchKernelGGL((xcd_fill_kernel<double, float>), grow, block, 0, st, rows, cpx, m->d_row_ptr, m->d_col,
                               (const double *)m->d_val, d_label_of_old, vrp, vcol, (float *)vval);
        else if (m->dtype == SPMV_HIP_F32)
            hipLaunchKernelGGL((xcd_fill_kernel<float, float>), grow, block, 0, st, rows, cpx, m->d_row_ptr, m->d_col,
                               (const float *)m->d_val, d_label_of_old, vrp, vcol, (float *)vval);
        else
            hipLaunchKernelGGL((xcd_fill_kernel<double, double>), grow, block, 0, st, rows, cpx, m->d_row_ptr, m->d_col,
                               (const double *)m->d_val, d_label_of_old, vrp, vcol, (double *)vval);
        if (nhuge > 0) {
            const dim3 gh((unsigned)(nseg * nhuge)), bh(64);
            if (m->dtype == SPMV_HIP_F32 && val_is_f64)
                hipLaunchKernelGGL((xcd_fill_huge_kernel<double, float>), gh, bh, 0, st, rows, cpx, nseg, m->d_row_ptr, m->d_col,
                                   (const double *)m->d_val, d_label_of_old, d_huge, d_seg_cnt, vrp, vcol, (float *)vval);
            else if (m->dtype == SPMV_HIP_F32)
                hipLaunchKernelGGL((xcd_fill_huge_kernel<float, float>), gh, bh, 0, st, rows, cpx, nseg, m->d_row_ptr, m->d_col,
                                   (const float *)m->d_val, d_label_of_old, d_huge, d_seg_cnt, vrp, vcol, (float *)vval);
            else
                hipLaunchKernelGGL((xcd_fill_huge_kernel<double, double>), gh, bh, 0, st, rows, cpx, nseg, m->d_row_ptr, m->d_col,
                                   (const double *)m->d_val, d_label_of_old, d_huge, d_seg_cnt, vrp, vcol, (double *)vval);
        }
        SPMV_HIP_TRY(hipGetLastError(), break);
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        *out_vrp = vrp; *out_vcol = vcol; *out_vval = vval; *out_old_of_label = old_of_label; *out_cpx = cpx;
        vrp = nullptr; vcol = nullptr; vval = nullptr; old_of_label = nullptr;
        rc = 0;
    } while (0);
    void *scratch[] = {d_count[0], d_count[1], d_id, d_rank_old, d_label_of_old, d_cnt, d_huge, d_meta, d_seg_cnt, d_temp, vrp,
                       vcol, vval, old_of_label};
    for (void *p : scratch)
        if (p) (void)hipFree(p);
    return rc;
}

} // namespace spmv

// C-linkage hook for the host-only ingest file (mtxio.cpp is compiled without HIP headers)
extern "C" __attribute__((visibility("hidden"))) int spmv_hip_internal_device_sort_coo(long long nnz, struct mtx *coo)
{
    return spmv::device_sort_coo((int64_t)nnz, coo);
}
