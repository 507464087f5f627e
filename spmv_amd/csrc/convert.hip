// Format converters: build the HBM-resident layouts of every format from a
// CSR that already sits in HBM.  The heavy passes (sigma-window sort, slice
// fill, merge-path partition) run on the device; prefix sums over per-slice /
// per-row metadata run on the host at create time.

#include "internal.h"

#include <climits>

#include <algorithm>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

namespace spmv {

static inline unsigned cdiv64(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// mean row length -> lanes per row for the vector CSR kernel (power of two, 2..64)
int choose_lanes_per_row(int rows, int64_t stored)
{
    double mean = rows > 0 ? (double)stored / rows : 0.0;
    int g = 2;
    while (g < 64 && g < mean) g *= 2;
    return g;
}

// ------------------------------------------------------------- merge path ----

// One thread per tile boundary: the 2-D diagonal search of the reference
// (csr_merge.c:62-85) on the global row-end list.
__global__ void merge_partition_kernel(int ntiles, int items_per_tile, int n, int nz,
                                       const int *__restrict__ row_end, int *__restrict__ tile_coord)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > ntiles) return;
    int64_t d64 = (int64_t)t * items_per_tile;
    int64_t items = (int64_t)n + nz;
    int d = (int)(d64 < items ? d64 : items);
    int lo = d > nz ? d - nz : 0;
    int hi = d < n ? d : n;
    while (lo < hi) {
        int mid = (int)(((int64_t)lo + hi) >> 1);
        if (row_end[mid] <= d - mid - 1) lo = mid + 1; else hi = mid;
    }
    tile_coord[2 * t] = lo;
    tile_coord[2 * t + 1] = d - lo;
}

// tile_pat[t] = L when every row tile t touches (list positions a0..a1) has L entries (1 <= L <= 32) at
// the same distances from the diagonal and the rows are consecutive; 0 otherwise.  The product kernel
// then takes the tile's column indices from the first of those rows.  One wave per tile.
__global__ __launch_bounds__(kWave) void tile_pattern_kernel(int ntiles, int list_len, const int *__restrict__ row_end_list,
                                                             const int *__restrict__ row_index,
                                                             const int *__restrict__ tile_coord,
                                                             const int *__restrict__ col,
                                                             unsigned char *__restrict__ tile_pat, int *__restrict__ npat)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const int a0 = tile_coord[2 * t], b0 = tile_coord[2 * t + 1];
    const int a1 = tile_coord[2 * t + 2], b1 = tile_coord[2 * t + 3];
    const int hi = min(a1, list_len - 1);
    int out = 0;
    if (b1 > b0 && a0 <= hi) {                               // wave-uniform
        const int start0 = a0 ? row_end_list[a0 - 1] : 0;
        const int L = row_end_list[a0] - start0;
        const int row0 = row_index ? row_index[a0] : a0;
        if (L >= 1 && L <= 32) {
            bool ok = true;
            for (int a = a0 + lane; a <= hi && ok; a += kWave) {
                const int s = a ? row_end_list[a - 1] : 0;
                const int row = row_index ? row_index[a] : a;
                ok = row_end_list[a] - s == L && row == row0 + (a - a0);
                for (int k = 0; k < L && ok; k++) ok = col[s + k] - row == col[start0 + k] - row0;
            }
            if (!__ballot(!ok)) out = L;
        }
    }
    if (lane == 0) {
        tile_pat[t] = (unsigned char)out;
        if (out) atomicAdd(npat, 1);
    }
}

static int build_tile_patterns(spmv_hip_matrix *m, int list_len, const int *list, const int *index, hipStream_t st)
{
    if (m->ntiles == 0) return 0;
    int *d_count = nullptr, count = 0;
    if (dev_alloc(m, &m->d_tile_pat, (size_t)m->ntiles)) return 1;
    if (dev_alloc<int>(nullptr, &d_count, 1)) return 1;
    bool ok = hipMemsetAsync(d_count, 0, sizeof(int), st) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(tile_pattern_kernel, dim3((unsigned)m->ntiles), dim3(kWave), 0, st, m->ntiles, list_len, list,
                           index, m->d_tile_coord, m->d_col, m->d_tile_pat, d_count);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    (void)hipFree(d_count);
    if (!ok) { set_error("tiles: scanning for row patterns failed"); return 1; }
    m->pat_tiles = count;
    return 0;
}

int build_merge_tiles(spmv_hip_matrix *m, hipStream_t st)
{
    int64_t items = (int64_t)m->rows + m->stored;
    if (items >= 0x7fffffffLL) { set_error("csr_merge: rows + nnz = %lld does not fit 32-bit path coordinates", (long long)items); return 1; }
    m->ntiles = (int)((items + m->items_per_tile - 1) / m->items_per_tile);
    if (dev_alloc(m, &m->d_tile_coord, 2 * ((size_t)m->ntiles + 1))) return 1;
    if (dev_alloc(m, &m->d_row_carry, (size_t)m->ntiles)) return 1;
    if (dev_alloc(m, (char **)&m->d_val_carry, (size_t)m->ntiles * dtype_size(m->dtype))) return 1;
    hipLaunchKernelGGL(merge_partition_kernel, dim3(cdiv64(m->ntiles + 1, 256)), dim3(256), 0, st, m->ntiles,
                       m->items_per_tile, m->rows, (int)m->stored, m->d_row_ptr + 1, m->d_tile_coord);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    SPMV_HIP_TRY(hipStreamSynchronize(st), return 1);
    return build_tile_patterns(m, m->rows, m->d_row_ptr + 1, nullptr, st);
}

// ------------------------------------------------------------ balanced CSR ----

// Equal-nnz tiles (reference csr_bal.c:83-85) over the list of non-empty rows
// (the reference's per-chunk local CSR holds exactly those: csr_bal.c:45-59).
// tile_coord[2t] = first non-empty row whose end lies beyond the tile start,
// tile_coord[2t+1] = tile start in the non-zero array.
__global__ void row_nonempty_kernel(int rows, const int *__restrict__ row_ptr, int *__restrict__ flag)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) flag[r] = row_ptr[r + 1] > row_ptr[r];
}

// pos = exclusive prefix sum of the non-empty flags: row r is entry pos[r] of the compacted list
__global__ void compact_rows_kernel(int rows, const int *__restrict__ row_ptr, const int *__restrict__ pos,
                                    int *__restrict__ id, int *__restrict__ end)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows || row_ptr[r + 1] <= row_ptr[r]) return;
    id[pos[r]] = r;
    end[pos[r]] = row_ptr[r + 1];
}

// tile t starts at non-zero b = t * per_tile: first list entry whose end lies beyond b (upper bound)
__global__ void bal_partition_kernel(int ntiles, int per_tile, int64_t nnz, int list_len,
                                     const int *__restrict__ end, int *__restrict__ tile_coord)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > ntiles) return;
    const int64_t b64 = (int64_t)t * per_tile;
    const int b = (int)(b64 < nnz ? b64 : nnz);
    int lo = 0, hi = list_len;
    while (lo < hi) {
        int mid = (int)(((int64_t)lo + hi) >> 1);
        if (end[mid] <= b) lo = mid + 1; else hi = mid;
    }
    tile_coord[2 * t] = lo;
    tile_coord[2 * t + 1] = b;
}

int build_bal_tiles(spmv_hip_matrix *m, hipStream_t st)
{
    const int rows = m->rows;
    const int64_t nnz = m->stored;
    const int ch = m->items_per_tile;
    m->ntiles = (int)((nnz + ch - 1) / ch);
    int *d_flag = nullptr, *d_pos = nullptr;
    int rc = 1;
    do {
        m->nz_rows = 0;
        if (rows > 0) {
            if (dev_alloc<int>(nullptr, &d_flag, (size_t)rows) || dev_alloc<int>(nullptr, &d_pos, (size_t)rows)) break;
            hipLaunchKernelGGL(row_nonempty_kernel, dim3(cdiv64(rows, 256)), dim3(256), 0, st, rows, m->d_row_ptr, d_flag);
            SPMV_HIP_TRY(hipGetLastError(), break);
            if (device_exclusive_scan(d_flag, d_pos, (size_t)rows, st)) break;
            int last_pos = 0, last_flag = 0;
            SPMV_HIP_TRY(hipMemcpyAsync(&last_pos, d_pos + rows - 1, sizeof(int), hipMemcpyDeviceToHost, st), break);
            SPMV_HIP_TRY(hipMemcpyAsync(&last_flag, d_flag + rows - 1, sizeof(int), hipMemcpyDeviceToHost, st), break);
            SPMV_HIP_TRY(hipStreamSynchronize(st), break);
            m->nz_rows = last_pos + last_flag;
        }
        m->has_empty_rows = m->nz_rows != rows;
        if (dev_alloc(m, &m->d_nzr_id, (size_t)m->nz_rows)) break;
        if (dev_alloc(m, &m->d_nzr_end, (size_t)m->nz_rows)) break;
        if (dev_alloc(m, &m->d_tile_coord, 2 * ((size_t)m->ntiles + 1))) break;
        if (dev_alloc(m, &m->d_row_carry, (size_t)m->ntiles)) break;
        if (dev_alloc(m, (char **)&m->d_val_carry, (size_t)m->ntiles * dtype_size(m->dtype))) break;
        if (rows > 0) {
            hipLaunchKernelGGL(compact_rows_kernel, dim3(cdiv64(rows, 256)), dim3(256), 0, st, rows, m->d_row_ptr, d_pos,
                               m->d_nzr_id, m->d_nzr_end);
            SPMV_HIP_TRY(hipGetLastError(), break);
        }
        hipLaunchKernelGGL(bal_partition_kernel, dim3(cdiv64(m->ntiles + 1, 256)), dim3(256), 0, st, m->ntiles, ch, nnz,
                           m->nz_rows, m->d_nzr_end, m->d_tile_coord);
        SPMV_HIP_TRY(hipGetLastError(), break);
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        if (build_tile_patterns(m, m->nz_rows, m->d_nzr_end, m->d_nzr_id, st)) break;
        rc = 0;
    } while (0);
    if (d_flag) (void)hipFree(d_flag);
    if (d_pos) (void)hipFree(d_pos);
    return rc;
}

// ---------------------------------------------------------- sliced formats ----

// Stable descending sort of the row lengths inside one sigma window per
// workgroup (reference radix_sort.c:103-122: LSD radix, larger digit first,
// ties keep ascending row index).  A bitonic network in LDS on (length, position)
// keys; windows holding a row of 2^17 entries or more are ranked by counting
// instead: position of row i = #{j : len_j > len_i} + #{j < i : len_j == len_i}.
__global__ __launch_bounds__(1024) void sell_window_sort_kernel(int rows, int window, int padded,
                                                                const int *__restrict__ row_ptr,
                                                                int *__restrict__ row_order,
                                                                int *__restrict__ sorted_len)
{
    extern __shared__ int lens[];                      // `padded` ints: window rounded up to a power of two
    __shared__ int longest;
    const int w0 = blockIdx.x * window;
    const int n = min(window, rows - w0);
    if (threadIdx.x == 0) longest = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        lens[i] = row_ptr[w0 + i + 1] - row_ptr[w0 + i];
        mine = max(mine, lens[i]);
    }
    atomicMax(&longest, mine);
    __syncthreads();
    // Lengths below 2^17 and positions below 2^15 pack into one 32-bit key whose ascending order is
    // "longer first, then lower row": distinct keys, so a bitonic network gives the stable order.
    if (longest < (1 << 17)) {
        unsigned *key = (unsigned *)lens;
        for (int i = threadIdx.x; i < padded; i += blockDim.x)
            key[i] = i < n ? ((unsigned)((1 << 17) - 1 - lens[i]) << 15) | (unsigned)i : 0xFFFFFFFFu;
        __syncthreads();
        for (int k = 2; k <= padded; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < padded; i += blockDim.x) {
                    const int partner = i ^ j;
                    if (partner > i) {
                        const unsigned a = key[i], b = key[partner];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { key[i] = b; key[partner] = a; }
                    }
                }
                __syncthreads();
            }
        for (int r = threadIdx.x; r < n; r += blockDim.x) {
            row_order[w0 + r] = w0 + (int)(key[r] & 0x7FFFu);
            sorted_len[w0 + r] = (1 << 17) - 1 - (int)(key[r] >> 15);
        }
        return;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int li = lens[i];
        int rank = 0;
        for (int j = 0; j < n; j++) {
            int lj = lens[j];
            rank += (lj > li) | ((lj == li) & (j < i));
        }
        row_order[w0 + rank] = w0 + i;
        sorted_len[w0 + rank] = li;
    }
}

// Windows too long for the LDS sort: one stable radix sort over all rows on the key (window, longer
// first); ties keep their ascending row index, as in the reference's LSD radix sort.
__global__ void sell_window_key_kernel(int rows, int64_t window, const int *__restrict__ row_ptr,
                                       unsigned long long *__restrict__ key, int *__restrict__ id)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const unsigned len = (unsigned)(row_ptr[r + 1] - row_ptr[r]);
    key[r] = ((unsigned long long)(r / window) << 32) | (unsigned long long)(0xFFFFFFFFu - len);
    id[r] = r;
}

__global__ void sell_sorted_len_kernel(int rows, const int *__restrict__ row_ptr, const int *__restrict__ row_order,
                                       int *__restrict__ sorted_len)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= rows) return;
    const int r = row_order[k];
    sorted_len[k] = row_ptr[r + 1] - row_ptr[r];
}

// row_order / sorted_len of the sigma-window sort for any window length
static int sell_sort_windows(int rows, int64_t window, const int *d_row_ptr, int *d_row_order, int *d_sorted_len, hipStream_t st)
{
    if (window <= kMaxSigma) {
        int padded = 1;
        while (padded < window) padded <<= 1;
        const size_t lds = (size_t)padded * sizeof(int);
        SPMV_HIP_TRY(hipFuncSetAttribute((const void *)sell_window_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                     return 1);
        hipLaunchKernelGGL(sell_window_sort_kernel, dim3(cdiv64(rows, window)), dim3(1024), lds, st, rows, (int)window, padded,
                           d_row_ptr, d_row_order, d_sorted_len);
        SPMV_HIP_TRY(hipGetLastError(), return 1);
        return 0;
    }
    unsigned long long *key[2] = {nullptr, nullptr};
    int *id = nullptr;
    void *temp = nullptr;
    int rc = 1;
    do {
        const size_t n = (size_t)rows;
        SPMV_HIP_TRY(hipMalloc((void **)&key[0], n * sizeof(unsigned long long)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&key[1], n * sizeof(unsigned long long)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&id, n * sizeof(int)), break);
        hipLaunchKernelGGL(sell_window_key_kernel, dim3(cdiv64(rows, 256)), dim3(256), 0, st, rows, window, d_row_ptr, key[0], id);
        SPMV_HIP_TRY(hipGetLastError(), break);
        size_t bytes = 0;
        SPMV_HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, key[0], key[1], id, d_row_order, n, 0, 64, st), break);
        SPMV_HIP_TRY(hipMalloc(&temp, std::max<size_t>(bytes, 16)), break);
        SPMV_HIP_TRY(rocprim::radix_sort_pairs(temp, bytes, key[0], key[1], id, d_row_order, n, 0, 64, st), break);
        hipLaunchKernelGGL(sell_sorted_len_kernel, dim3(cdiv64(rows, 256)), dim3(256), 0, st, rows, d_row_ptr, d_row_order, d_sorted_len);
        SPMV_HIP_TRY(hipGetLastError(), break);
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        rc = 0;
    } while (0);
    void *scratch[] = {key[0], key[1], id, temp};
    for (void *p : scratch)
        if (p) (void)hipFree(p);
    return rc;
}

// Width of every slice: length of its first row when the rows were sorted
// (reference sellcs_format.c:152-158), longest row otherwise (sellp.c:411-421,
// ell.c:72-82).
__global__ void slice_width_kernel(int rows, int height, int nslices, const int *__restrict__ row_ptr,
                                   const int *__restrict__ sorted_len, int *__restrict__ width)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    int64_t r0 = (int64_t)s * height;
    if (sorted_len) { width[s] = sorted_len[r0]; return; }
    int w = 0;
    for (int k = 0; k < height && r0 + k < rows; k++) w = max(w, row_ptr[r0 + k + 1] - row_ptr[r0 + k]);
    width[s] = w;
}

// Column-major slice fill with (col 0, value 0) padding (reference
// sellcs_format.c:164-187, sellp.c:433-453).  One thread per slice lane.
template <typename Tin, typename Tout>
__global__ __launch_bounds__(256) void slice_fill_kernel(int rows, int height, int64_t lanes,
                                                         const int *__restrict__ row_ptr,
                                                         const int *__restrict__ col, const Tin *__restrict__ val,
                                                         const int *__restrict__ row_order,
                                                         const int64_t *__restrict__ slice_ptr,
                                                         int *__restrict__ scol, Tout *__restrict__ sval)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= lanes) return;
    int s = (int)(r / height), k = (int)(r - (int64_t)s * height);
    int64_t base = slice_ptr[s];
    int width = (int)((slice_ptr[s + 1] - base) / height);
    int beg = 0, len = 0;
    if (r < rows) {
        int src = row_order ? row_order[r] : (int)r;
        beg = row_ptr[src];
        len = row_ptr[src + 1] - beg;
    }
    for (int j = 0; j < width; j++) {
        int64_t at = base + (int64_t)j * height + k;
        bool live = j < len;
        scol[at] = live ? col[beg + j] : 0;
        sval[at] = live ? (Tout)val[beg + j] : (Tout)0;
    }
}

// The same fill for slices of 64 rows, transposed through LDS: one wave per slice takes 16 columns at a
// time; 16-lane groups read 16 consecutive entries of one row each (contiguous 64/128-byte pieces of the
// CSR arrays, every line used by one or two back-to-back steps), the tile is parked in LDS, and lane k
// writes row k's entries, so each store instruction is one full 256/512-byte column of the slice.
constexpr int kFillCols = 16;

template <typename Tin, typename Tout>
__global__ __launch_bounds__(kWave) void slice_fill64_kernel(int rows, int nslices, const int *__restrict__ row_ptr,
                                                             const int *__restrict__ col,
                                                             const Tin *__restrict__ val,
                                                             const int *__restrict__ row_order,
                                                             const int64_t *__restrict__ slice_ptr,
                                                             int *__restrict__ scol, Tout *__restrict__ sval)
{
    constexpr int kStride = kFillCols + 1;                   // odd row stride: conflict-free column reads
    __shared__ int tcol[kWave * kStride];
    __shared__ Tout tval[kWave * kStride];
    const int s = blockIdx.x, lane = threadIdx.x;
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const int r = s * kWave + lane;
    int beg = 0, len = 0;
    if (r < rows) {
        const int src = row_order ? row_order[r] : r;
        beg = row_ptr[src];
        len = row_ptr[src + 1] - beg;
    }
    const int sub = lane / kFillCols, jj = lane % kFillCols;  // 4 rows per load step, 16 columns each
    for (int j0 = 0; j0 < width; j0 += kFillCols) {
        int c[kFillCols];
        Tin v[kFillCols];
#pragma unroll
        for (int i = 0; i < kFillCols; i++) {
            const int row = 4 * i + sub;
            const int b = __shfl(beg, row, kWave), l = __shfl(len, row, kWave);
            const bool live = j0 + jj < l;
            c[i] = live ? col[b + j0 + jj] : 0;
            v[i] = live ? val[b + j0 + jj] : (Tin)0;
        }
#pragma unroll
        for (int i = 0; i < kFillCols; i++) {
            tcol[(4 * i + sub) * kStride + jj] = c[i];
            tval[(4 * i + sub) * kStride + jj] = (Tout)v[i];
        }
        __syncthreads();
        const int ncols = min(kFillCols, width - j0);
        for (int k = 0; k < ncols; k++) {
            const int64_t at = base + (int64_t)(j0 + k) * kWave + lane;
            scol[at] = tcol[lane * kStride + k];
            sval[at] = tval[lane * kStride + k];
        }
        __syncthreads();
    }
}

// ------------------------------------------- SELL-C-sigma without padding ----
// Layout of the reference's csr_to_sellcs_dfc (sellcs_format.c:53-136): rows sorted as for SELL; column
// ("vop") v of a slice stores one entry for every row of the slice longer than v -- a prefix of the slice,
// as its rows are sorted -- and vop_len holds that count minus one (sellcs_utils.c:79-125).

// one thread per slice: its width (longest = first row) and its number of entries
__global__ void dfc_slice_stats_kernel(int rows, int nslices, const int *__restrict__ sorted_len,
                                       int *__restrict__ width, int *__restrict__ entries)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    const int r0 = s * kWave, r1 = min(r0 + kWave, rows);
    int total = 0;
    for (int r = r0; r < r1; r++) total += sorted_len[r];
    width[s] = sorted_len[r0];
    entries[s] = total;
}

// one wave per slice: active lanes - 1 of each of its columns
__global__ __launch_bounds__(kWave) void dfc_vop_len_kernel(int rows, const int *__restrict__ sorted_len,
                                                            const int64_t *__restrict__ vop_ptr,
                                                            uint8_t *__restrict__ vop_len)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int r = s * kWave + lane;
    const int len = r < rows ? sorted_len[r] : 0;
    const int64_t v0 = vop_ptr[s];
    const int width = (int)(vop_ptr[s + 1] - v0);
    for (int v = lane; v - lane < width; v += kWave) {
        int active = 0;
        for (int k = 0; k < kWave; k++) active += __shfl(len, k, kWave) > v;
        if (v < width) vop_len[v0 + v] = (uint8_t)(active - 1);
    }
}

// one wave per slice: 16 columns at a time through LDS (as slice_fill64_kernel), each column written as the
// contiguous run of its active lanes
template <typename Tin, typename Tout>
__global__ __launch_bounds__(kWave) void dfc_fill_kernel(int rows, const int *__restrict__ row_ptr,
                                                         const int *__restrict__ col, const Tin *__restrict__ val,
                                                         const int *__restrict__ row_order,
                                                         const int *__restrict__ sorted_len,
                                                         const int64_t *__restrict__ slice_ptr,
                                                         int *__restrict__ scol, Tout *__restrict__ sval)
{
    constexpr int kStride = kFillCols + 1;
    __shared__ int tcol[kWave * kStride];
    __shared__ Tout tval[kWave * kStride];
    const int s = blockIdx.x, lane = threadIdx.x;
    const int r = s * kWave + lane;
    int beg = 0, len = 0;
    if (r < rows) { beg = row_ptr[row_order[r]]; len = sorted_len[r]; }
    const int width = __shfl(len, 0, kWave);
    int64_t at = slice_ptr[s];
    const int sub = lane / kFillCols, jj = lane % kFillCols;
    for (int j0 = 0; j0 < width; j0 += kFillCols) {
        int c[kFillCols];
        Tin v[kFillCols];
#pragma unroll
        for (int i = 0; i < kFillCols; i++) {
            const int row = 4 * i + sub;
            const int b = __shfl(beg, row, kWave), l = __shfl(len, row, kWave);
            const bool live = j0 + jj < l;
            c[i] = live ? col[b + j0 + jj] : 0;
            v[i] = live ? val[b + j0 + jj] : (Tin)0;
        }
#pragma unroll
        for (int i = 0; i < kFillCols; i++) {
            tcol[(4 * i + sub) * kStride + jj] = c[i];
            tval[(4 * i + sub) * kStride + jj] = (Tout)v[i];
        }
        __syncthreads();
        const int ncols = min(kFillCols, width - j0);
        for (int k = 0; k < ncols; k++) {
            const bool live = j0 + k < len;
            if (live) {
                scol[at + lane] = tcol[lane * kStride + k];
                sval[at + lane] = tval[lane * kStride + k];
            }
            at += __popcll(__ballot(live));
        }
        __syncthreads();
    }
}

// The padding-free layout's counterpart of slice_diag_kernel: diag[first column of slice s + j] = the
// distance col - row shared by the active lanes of column j, or INT_MIN.  One wave per slice.
__global__ __launch_bounds__(kWave) void dfc_diag_kernel(int rows, int nslices, const int *__restrict__ sorted_len,
                                                         const int *__restrict__ row_order,
                                                         const int64_t *__restrict__ slice_ptr,
                                                         const int64_t *__restrict__ vop_ptr,
                                                         const int *__restrict__ scol, int *__restrict__ diag,
                                                         unsigned long long *__restrict__ ndiag)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= nslices) return;
    const int r = s * kWave + lane;
    const int len = r < rows ? sorted_len[r] : 0;
    const int row = r < rows ? row_order[r] : 0;
    const int width = __builtin_amdgcn_readfirstlane(len);      // lane 0 holds the slice's longest row
    const int64_t base = slice_ptr[s], vp = vop_ptr[s];
    int run = 0, found = 0;
    for (int j = 0; j < width; j++) {
        const bool real = j < len;
        const int d = real ? scol[base + run + lane] - row : 0;
        const int d0 = __builtin_amdgcn_readfirstlane(d);
        const int out = __ballot(real && d != d0) ? INT_MIN : d0;
        if (lane == 0) diag[vp + j] = out;
        found += out != INT_MIN;
        run += __popcll(__ballot(real));
    }
    if (lane == 0 && found) atomicAdd(ndiag, (unsigned long long)found);
}

int build_sell_dfc(spmv_hip_matrix *m, const int *d_row_ptr, const int *d_col, const void *d_val,
                   bool val_is_f64, hipStream_t st)
{
    const int rows = m->rows;
    if (m->slice_height != kWave) { set_error("sell_dfc: C %d must be 64 (one wave per slice)", m->slice_height); return 1; }
    m->nslices = (rows + kWave - 1) / kWave;
    const bool f32 = m->dtype == SPMV_HIP_F32;
    if (!val_is_f64 && !f32) { set_error("sell_dfc: float CSR values cannot build a double matrix"); return 1; }
    int *d_width = nullptr, *d_entries = nullptr;
    int rc = 1;
    std::vector<int> width((size_t)m->nslices), entries((size_t)m->nslices);
    std::vector<int64_t> vp((size_t)m->nslices + 1), sp((size_t)m->nslices + 1);
    do {
        if (dev_alloc(m, &m->d_row_order, (size_t)rows) || dev_alloc(m, &m->d_sorted_len, (size_t)rows)) break;
        if (rows > 0) {
            const int64_t window = std::min<int64_t>(m->sigma, rows);
            if (m->sigma % kWave != 0 && m->sigma < rows) {
                set_error("sell_dfc: sigma %lld must be a multiple of C 64", (long long)m->sigma);
                break;
            }
            if (sell_sort_windows(rows, window, d_row_ptr, m->d_row_order, m->d_sorted_len, st)) break;
            if (dev_alloc<int>(nullptr, &d_width, (size_t)m->nslices) || dev_alloc<int>(nullptr, &d_entries, (size_t)m->nslices)) break;
            hipLaunchKernelGGL(dfc_slice_stats_kernel, dim3(cdiv64(m->nslices, 256)), dim3(256), 0, st, rows, m->nslices,
                               m->d_sorted_len, d_width, d_entries);
            SPMV_HIP_TRY(hipGetLastError(), break);
            SPMV_HIP_TRY(hipMemcpyAsync(width.data(), d_width, sizeof(int) * width.size(), hipMemcpyDeviceToHost, st), break);
            SPMV_HIP_TRY(hipMemcpyAsync(entries.data(), d_entries, sizeof(int) * entries.size(), hipMemcpyDeviceToHost, st), break);
        }
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        int64_t vops = 0, total = 0;
        for (int s = 0; s < m->nslices; s++) { vp[s] = vops; sp[s] = total; vops += width[s]; total += entries[s]; }
        vp[m->nslices] = vops; sp[m->nslices] = total;
        m->vops = vops;
        m->stored = total;
        if (dev_alloc(m, &m->d_vop_ptr, vp.size()) || dev_alloc(m, &m->d_slice_ptr, sp.size())) break;
        if (dev_alloc(m, &m->d_vop_len, (size_t)vops)) break;
        if (dev_alloc(m, &m->d_scol, (size_t)total)) break;
        if (dev_alloc(m, (char **)&m->d_sval, (size_t)total * dtype_size(m->dtype))) break;
        SPMV_HIP_TRY(hipMemcpyAsync(m->d_vop_ptr, vp.data(), sizeof(int64_t) * vp.size(), hipMemcpyHostToDevice, st), break);
        SPMV_HIP_TRY(hipMemcpyAsync(m->d_slice_ptr, sp.data(), sizeof(int64_t) * sp.size(), hipMemcpyHostToDevice, st), break);
        if (m->nslices > 0) {
            const dim3 grid((unsigned)m->nslices), block(kWave);
            hipLaunchKernelGGL(dfc_vop_len_kernel, grid, block, 0, st, rows, m->d_sorted_len, m->d_vop_ptr, m->d_vop_len);
            if (val_is_f64 && !f32)
                hipLaunchKernelGGL((dfc_fill_kernel<double, double>), grid, block, 0, st, rows, d_row_ptr, d_col,
                                   (const double *)d_val, m->d_row_order, m->d_sorted_len, m->d_slice_ptr, m->d_scol,
                                   (double *)m->d_sval);
            else if (val_is_f64)
                hipLaunchKernelGGL((dfc_fill_kernel<double, float>), grid, block, 0, st, rows, d_row_ptr, d_col,
                                   (const double *)d_val, m->d_row_order, m->d_sorted_len, m->d_slice_ptr, m->d_scol,
                                   (float *)m->d_sval);
            else
                hipLaunchKernelGGL((dfc_fill_kernel<float, float>), grid, block, 0, st, rows, d_row_ptr, d_col,
                                   (const float *)d_val, m->d_row_order, m->d_sorted_len, m->d_slice_ptr, m->d_scol,
                                   (float *)m->d_sval);
            SPMV_HIP_TRY(hipGetLastError(), break);
        }
        if (vops > 0) {
            // one number per column: the distance of its entries from the diagonal, where they share one
            unsigned long long *d_count = nullptr, count = 0;
            if (dev_alloc(m, &m->d_sdiag, (size_t)vops)) break;
            if (dev_alloc<unsigned long long>(nullptr, &d_count, 1)) break;
            bool ok = hipMemsetAsync(d_count, 0, sizeof count, st) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(dfc_diag_kernel, dim3((unsigned)m->nslices), dim3(kWave), 0, st, rows, m->nslices,
                                   m->d_sorted_len, m->d_row_order, m->d_slice_ptr, m->d_vop_ptr, m->d_scol, m->d_sdiag,
                                   d_count);
                ok = hipGetLastError() == hipSuccess &&
                     hipMemcpyAsync(&count, d_count, sizeof count, hipMemcpyDeviceToHost, st) == hipSuccess &&
                     hipStreamSynchronize(st) == hipSuccess;
            }
            (void)hipFree(d_count);
            if (!ok) { set_error("sell_dfc: scanning the columns for diagonals failed"); break; }
            m->diag_columns = (int64_t)count;
        }
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        rc = 0;
    } while (0);
    if (d_width) (void)hipFree(d_width);
    if (d_entries) (void)hipFree(d_entries);
    return rc;
}

// pattern_len[b] = L when the 64 rows of block b all have L entries (1 <= L <= 32) and entry k of
// every row sits at the same distance from the diagonal; 0 otherwise (and for the last, partial
// block).  One wave per block, lane = row.
__global__ __launch_bounds__(kWave) void csr_block_pattern_kernel(int rows, int columns, const int *__restrict__ row_ptr,
                                                                  const int *__restrict__ col,
                                                                  unsigned char *__restrict__ pattern_len,
                                                                  int *__restrict__ npat)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int row = b * kWave + lane;
    int out = 0;
    if (b * kWave + kWave <= rows) {                      // wave-uniform
        const int beg = row_ptr[row], len = row_ptr[row + 1] - beg;
        const int L = __builtin_amdgcn_readfirstlane(len);
        bool same = len == L;
        if (!__ballot(!same) && L >= 1 && L <= 32) {
            for (int k = 0; k < L; k++) {
                const int d = col[beg + k] - row;
                same = same && d == __builtin_amdgcn_readfirstlane(d);
            }
            if (!__ballot(!same)) out = L;
        }
    }
    if (lane == 0) {
        pattern_len[b] = (unsigned char)out;
        if (out) atomicAdd(npat, 1);
    }
}

int build_block_patterns(spmv_hip_matrix *m, hipStream_t st)
{
    const int nblocks = (m->rows + kWave - 1) / kWave;
    if (nblocks == 0 || m->stored == 0) return 0;
    int *d_count = nullptr, count = 0;
    if (dev_alloc(m, &m->d_block_pat, (size_t)nblocks)) return 1;
    if (dev_alloc<int>(nullptr, &d_count, 1)) return 1;
    bool ok = hipMemsetAsync(d_count, 0, sizeof(int), st) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(csr_block_pattern_kernel, dim3((unsigned)nblocks), dim3(kWave), 0, st, m->rows, m->columns,
                           m->d_row_ptr, m->d_col, m->d_block_pat, d_count);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    (void)hipFree(d_count);
    if (!ok) { set_error("csr: scanning the row blocks for patterns failed"); return 1; }
    m->pat_blocks = count;
    return 0;
}

// Column j of a slice is "diagonal" when all its real entries sit at the same distance from their
// own row: col = row + d (the rows of one grid line of a stencil, a band, a block diagonal).  Such a
// column is described by the single number d, so a kernel that knows it need not read the 64 column
// indices.  diag[first column of slice s + j] = d, or INT_MIN when the column is not diagonal;
// *ndiag counts the diagonal ones.  One wave per slice; padding lanes (j >= length of their row) are
// free.  A column with a real entry whose value is zero gets no descriptor: the product kernels
// recognise padding lanes of described columns by their zero value (they take x[0] there, as the
// reference's (column 0, value 0) padding does).
template <typename T>
__global__ __launch_bounds__(kWave) void slice_diag_kernel(int rows, int nslices, const int *__restrict__ row_ptr,
                                                           const int *__restrict__ row_order,
                                                           const int64_t *__restrict__ slice_ptr,
                                                           const int *__restrict__ scol, const T *__restrict__ sval,
                                                           int *__restrict__ diag,
                                                           unsigned long long *__restrict__ ndiag)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= nslices) return;
    const int r = s * kWave + lane;
    const int row = r < rows ? (row_order ? row_order[r] : r) : 0;
    const int len = r < rows ? row_ptr[row + 1] - row_ptr[row] : 0;
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    int found = 0;
    for (int j = 0; j < width; j++) {
        const bool real = j < len;
        const int d = real ? scol[base + (int64_t)j * kWave + lane] - row : 0;
        const bool zero = real && sval[base + (int64_t)j * kWave + lane] == (T)0;
        const unsigned long long mask = __ballot(real);
        int out = 0;                                           // a column of padding only: any distance will do
        if (mask) {
            const int d0 = __builtin_amdgcn_readlane(d, __builtin_ctzll(mask));
            out = __ballot((real && d != d0) || zero) ? INT_MIN : d0;
        }
        if (lane == 0) diag[(base >> 6) + j] = out;
        found += out != INT_MIN;
    }
    if (lane == 0 && found) atomicAdd(ndiag, (unsigned long long)found);
}

int build_sliced(spmv_hip_matrix *m, const int *d_row_ptr, const int *d_col, const void *d_val,
                 bool val_is_f64, bool sorted, hipStream_t st)
{
    const int rows = m->rows, H = m->slice_height;
    if (H < 1 || H > kWave || (H & (H - 1))) { set_error("slice height %d must be a power of two <= 64", H); return 1; }
    m->nslices = (rows + H - 1) / H;
    int *d_sorted_len = nullptr, *d_width = nullptr;
    int rc = 1;
    std::vector<int> width((size_t)m->nslices);
    std::vector<int64_t> sp((size_t)m->nslices + 1);
    do {
        if (sorted && rows > 0) {
            int64_t window = std::min<int64_t>(m->sigma, rows);
            if (m->sigma % H != 0 && m->sigma < rows) {
                set_error("sell: sigma %lld must be a multiple of C %d", (long long)m->sigma, H);
                break;
            }
            if (dev_alloc(m, &m->d_row_order, (size_t)rows)) break;
            if (dev_alloc<int>(nullptr, &d_sorted_len, (size_t)rows)) break;
            if (sell_sort_windows(rows, window, d_row_ptr, m->d_row_order, d_sorted_len, st)) break;
        }
        if (m->nslices > 0) {
            if (dev_alloc<int>(nullptr, &d_width, (size_t)m->nslices)) break;
            hipLaunchKernelGGL(slice_width_kernel, dim3(cdiv64(m->nslices, 256)), dim3(256), 0, st, rows, H,
                               m->nslices, d_row_ptr, d_sorted_len, d_width);
            SPMV_HIP_TRY(hipGetLastError(), break);
            SPMV_HIP_TRY(hipMemcpyAsync(width.data(), d_width, sizeof(int) * width.size(), hipMemcpyDeviceToHost, st), break);
        }
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        int64_t total = 0;
        for (int s = 0; s < m->nslices; s++) { sp[s] = total; total += (int64_t)width[s] * H; }
        sp[m->nslices] = total;
        if (m->max_padding > 0 && (double)total > m->max_padding * (double)std::max<int64_t>(m->nnz, 1)) {
            set_error("sliced layout would store %lld entries for %lld non-zeros (limit %.1fx)", (long long)total,
                      (long long)m->nnz, m->max_padding);
            break;
        }
        m->stored = total;
        if (dev_alloc(m, &m->d_slice_ptr, sp.size())) break;
        if (dev_alloc(m, &m->d_scol, (size_t)total)) break;
        if (dev_alloc(m, (char **)&m->d_sval, (size_t)total * dtype_size(m->dtype))) break;
        SPMV_HIP_TRY(hipMemcpyAsync(m->d_slice_ptr, sp.data(), sizeof(int64_t) * sp.size(), hipMemcpyHostToDevice, st), break);
        if (total > 0) {
            int64_t lanes = (int64_t)m->nslices * H;
            dim3 grid(cdiv64(lanes, 256)), block(256);
            const bool f32 = m->dtype == SPMV_HIP_F32;
            if (!val_is_f64 && !f32) { set_error("sliced: float CSR values cannot build a double matrix"); break; }
            if (H == kWave) {
                const dim3 g64((unsigned)m->nslices), b64(kWave);
                if (val_is_f64 && !f32)
                    hipLaunchKernelGGL((slice_fill64_kernel<double, double>), g64, b64, 0, st, rows, m->nslices, d_row_ptr,
                                       d_col, (const double *)d_val, m->d_row_order, m->d_slice_ptr, m->d_scol,
                                       (double *)m->d_sval);
                else if (val_is_f64)
                    hipLaunchKernelGGL((slice_fill64_kernel<double, float>), g64, b64, 0, st, rows, m->nslices, d_row_ptr,
                                       d_col, (const double *)d_val, m->d_row_order, m->d_slice_ptr, m->d_scol,
                                       (float *)m->d_sval);
                else
                    hipLaunchKernelGGL((slice_fill64_kernel<float, float>), g64, b64, 0, st, rows, m->nslices, d_row_ptr,
                                       d_col, (const float *)d_val, m->d_row_order, m->d_slice_ptr, m->d_scol,
                                       (float *)m->d_sval);
            } else if (val_is_f64 && !f32)
                hipLaunchKernelGGL((slice_fill_kernel<double, double>), grid, block, 0, st, rows, H, lanes, d_row_ptr, d_col,
                                   (const double *)d_val, m->d_row_order, m->d_slice_ptr, m->d_scol, (double *)m->d_sval);
            else if (val_is_f64 && f32)
                hipLaunchKernelGGL((slice_fill_kernel<double, float>), grid, block, 0, st, rows, H, lanes, d_row_ptr, d_col,
                                   (const double *)d_val, m->d_row_order, m->d_slice_ptr, m->d_scol, (float *)m->d_sval);
            else if (!val_is_f64 && f32)
                hipLaunchKernelGGL((slice_fill_kernel<float, float>), grid, block, 0, st, rows, H, lanes, d_row_ptr, d_col,
                                   (const float *)d_val, m->d_row_order, m->d_slice_ptr, m->d_scol, (float *)m->d_sval);
            else { set_error("sliced: float CSR values cannot build a double matrix"); break; }
            SPMV_HIP_TRY(hipGetLastError(), break);
        }
        if (total > 0 && H == kWave) {
            // one number per slice column: the distance of its entries from the diagonal, where they share one
            unsigned long long *d_count = nullptr, count = 0;
            if (dev_alloc(m, &m->d_sdiag, (size_t)(total / kWave))) break;
            if (dev_alloc<unsigned long long>(nullptr, &d_count, 1)) break;
            bool ok = hipMemsetAsync(d_count, 0, sizeof count, st) == hipSuccess;
            if (ok) {
                if (m->dtype == SPMV_HIP_F32)
                    hipLaunchKernelGGL(slice_diag_kernel<float>, dim3((unsigned)m->nslices), dim3(kWave), 0, st, rows,
                                       m->nslices, d_row_ptr, m->d_row_order, m->d_slice_ptr, m->d_scol,
                                       (const float *)m->d_sval, m->d_sdiag, d_count);
                else
                    hipLaunchKernelGGL(slice_diag_kernel<double>, dim3((unsigned)m->nslices), dim3(kWave), 0, st, rows,
                                       m->nslices, d_row_ptr, m->d_row_order, m->d_slice_ptr, m->d_scol,
                                       (const double *)m->d_sval, m->d_sdiag, d_count);
                ok = hipGetLastError() == hipSuccess &&
                     hipMemcpyAsync(&count, d_count, sizeof count, hipMemcpyDeviceToHost, st) == hipSuccess &&
                     hipStreamSynchronize(st) == hipSuccess;
            }
            (void)hipFree(d_count);
            if (!ok) { set_error("sliced: scanning the slice columns for diagonals failed"); break; }
            m->diag_columns = (int64_t)count;
        }
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        rc = 0;
    } while (0);
    if (d_sorted_len) (void)hipFree(d_sorted_len);
    if (d_width) (void)hipFree(d_width);
    return rc;
}

} // namespace spmv
