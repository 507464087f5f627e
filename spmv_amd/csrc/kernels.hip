// SpMV kernels for gfx950 (MI355X): wave64, 256-thread workgroups, HBM-bound.
//
// Every kernel walks the matrix in storage order so that consecutive logical
// blocks stream consecutive HBM addresses.  Which logical block a workgroup
// takes is the handle's "block_map" (xcd_block, device_utils.h): the hardware
// deals workgroups round-robin over the 8 XCDs, and the map gives each XCD
// either one contiguous eighth of the matrix (its private L2 then keeps that
// eighth's window of x instead of all eight L2s caching the same window) or
// short runs of consecutive blocks in turn (all XCDs then work side by side in
// one region of the matrix and of y, which the memory side likes better on
// matrices whose window of x is small).  The kernel trial picks.
//
// Summation order per kernel (what the parity tolerance has to cover):
//   csr_scalar, sliced (lane per row): ascending column position, one fma per
//     entry -- the order of mult_csr_base (reference csr.c:4-13); bit-identical
//     to it.  The variants that compute column indices instead of reading them
//     (diagonal slice columns, patterned row blocks / tiles) change where an
//     index comes from, never the order or the operands.
//   csr_vector: G interleaved partial sums per row + butterfly.
//   sliced wide variants: 2 (fp64) / 4 (fp32) interleaved partial sums per row.
//   sliced wide-exact variants (fp64): the wide variants' 16-byte loads, then a
//     half-wave swap gives every lane one row's entries, added in ascending
//     column order -- the order of mult_csr_base again.
//   tile (merge-path / balanced): product rounded, then each row's products
//     added in ascending order by one lane (rows longer than kLongRow: 64
//     interleaved partial sums + butterfly); a row cut by tile boundaries gets
//     its pieces added in tile order.

#include "internal.h"
#include "device_utils.h"

#include <algorithm>
#include <climits>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace spmv {

// ------------------------------------------------------------------ helpers --

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    long long bits = __builtin_bit_cast(long long, v);
    int lo = (int)(bits & 0xFFFFFFFFll), hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    bits = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, bits);
}

// Sum over aligned groups of G lanes; every lane of the group ends with the
// total.  DPP inside a 16-lane row (quad_perm xor1 / xor2, row_half_mirror,
// row_mirror), ds_bpermute shuffles across rows.
template <int G, typename T> __device__ __forceinline__ T group_sum(T v)
{
    if (G >= 2) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    if (G >= 4) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    if (G >= 8) v += dpp_mov<0x141>(v);  // row_half_mirror
    if (G >= 16) v += dpp_mov<0x140>(v); // row_mirror
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// ------------------------------------------------------------- CSR, 1 lane --

template <typename T>
__global__ __launch_bounds__(kBlock) void csr_scalar_kernel(int rows, const int *__restrict__ row_ptr,
                                                            const int *__restrict__ col,
                                                            const T *__restrict__ val,
                                                            const int *__restrict__ row_map,
                                                            const T *__restrict__ x, T *__restrict__ y, int map)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int row = (int)(b * kBlock + threadIdx.x);
    if (row >= rows) return;
    int beg = row_ptr[row], end = row_ptr[row + 1];
    T s = 0;
    for (int l = beg; l < end; l++) s = fma_t(x[col[l]], val[l], s);
    y[row_map ? row_map[row] : row] = s;
}

// Lane per row with the matrix stream staged through LDS.  A 64-lane workgroup
// owns 64 consecutive rows, i.e. one contiguous run of the column and value
// arrays; it walks that run in chunks of 64*IPL entries.  Per chunk: coalesced
// (column, value) loads, the x gathers, then (x, value) pairs parked in LDS,
// from where every lane adds the part of its own row that lies in the chunk --
// one fma per entry in ascending position, the accumulator carried from chunk
// to chunk.  Same rounding sequence as csr_scalar_kernel and as mult_csr_base
// (reference csr.c:4-13), but every HBM access is a full-width contiguous one,
// and y leaves as one aligned 64-element store per workgroup.
template <typename T> struct alignas(2 * sizeof(T)) xv_pair { T x, v; };

template <typename T, int IPL, bool NT>
__global__ __launch_bounds__(kWave) void csr_rows_lds_kernel(int rows, const int *__restrict__ row_ptr,
                                                             const int *__restrict__ col,
                                                             const T *__restrict__ val,
                                                             const int *__restrict__ row_map,
                                                             const T *__restrict__ x, T *__restrict__ y, int map)
{
    constexpr int kCap = kWave * IPL;
    __shared__ xv_pair<T> stage[kCap];
    const int blk = (int)xcd_block(blockIdx.x, gridDim.x, map);
    const int lane = threadIdx.x;
    const int row = blk * kWave + lane;
    const int last = min(blk * kWave + kWave, rows);
    const int rs = row_ptr[min(row, rows)];
    const int re = row < rows ? row_ptr[row + 1] : rs;
    const int run_begin = __builtin_amdgcn_readfirstlane(rs);
    const int run_end = row_ptr[last];
    T acc = 0;
    auto load_chunk = [&](int e0, int (&cc)[IPL], T (&vv)[IPL]) {
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const int *p = col + min(max(e0 + lane + i * kWave, run_begin), run_end - 1);
            cc[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const T *p = val + min(max(e0 + lane + i * kWave, run_begin), run_end - 1);
            vv[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    int cc[IPL];
    T vv[IPL];
    // chunks start on 32-entry boundaries of the arrays, so every load instruction covers whole
    // 128-byte lines (lanes before the run's first entry repeat it)
    const int first = run_begin & ~31;
    if (run_begin < run_end) load_chunk(first, cc, vv);
    for (int e0 = first; e0 < run_end; e0 += kCap) {
        const int n = min(kCap, run_end - e0);
        T xx[IPL];
        int cnext[IPL];
        T vnext[IPL];
#pragma unroll
        for (int i = 0; i < IPL; i++) xx[i] = x[cc[i]];
        // the next chunk's stream loads stay in flight during this chunk's row sums
        if (e0 + kCap < run_end) load_chunk(e0 + kCap, cnext, vnext);
#pragma unroll
        for (int i = 0; i < IPL; i++) stage[lane + i * kWave] = xv_pair<T>{xx[i], vv[i]};
#pragma unroll
        for (int i = 0; i < IPL; i++) { cc[i] = cnext[i]; vv[i] = vnext[i]; }
        __syncthreads();
        int k = max(rs, e0) - e0;
        const int kend = min(re, e0 + n) - e0;
        for (; k + 4 <= kend; k += 4) {
            xv_pair<T> p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = stage[k + u];
#pragma unroll
            for (int u = 0; u < 4; u++) acc = fma_t(p[u].x, p[u].v, acc);
        }
        for (; k < kend; k++) acc = fma_t(stage[k].x, stage[k].v, acc);
        __syncthreads();
    }
    if (row < rows) y[row_map ? row_map[row] : row] = acc;
}

// The same kernel for matrices whose rows repeat one pattern.  A block of 64 consecutive rows is
// "patterned" when all its rows have the same length L <= 32 and entry k of every row sits at the
// same distance from the diagonal (grids away from their boundaries, bands, block diagonals); the
// build pass leaves L (0 = not patterned) in one byte per block.  For such a block only the first
// row's L column indices are read: entry e of the block's run belongs to row e / L and column
// row + distance[e % L].  The value stream, the gathers, the LDS staging and the in-order row sums
// are those of csr_rows_lds_kernel, so y stays bit-identical to mult_csr_base (csr.c:4-13); the
// stream shrinks from V + 4 to V bytes per entry.
template <typename T, int IPL, bool NT>
__global__ __launch_bounds__(kWave) void csr_rows_lds_pat_kernel(int rows, const int *__restrict__ row_ptr,
                                                                 const int *__restrict__ col,
                                                                 const T *__restrict__ val,
                                                                 const unsigned char *__restrict__ pattern_len,
                                                                 const int *__restrict__ row_map,
                                                                 const T *__restrict__ x, T *__restrict__ y, int map)
{
    constexpr int kCap = kWave * IPL;
    __shared__ xv_pair<T> stage[kCap];
    __shared__ int dist[32];
    const int blk = (int)xcd_block(blockIdx.x, gridDim.x, map);
    const int lane = threadIdx.x;
    const int row0 = blk * kWave;
    const int row = row0 + lane;
    const int last = min(row0 + kWave, rows);
    const int rs = row_ptr[min(row, rows)];
    const int re = row < rows ? row_ptr[row + 1] : rs;
    const int run_begin = __builtin_amdgcn_readfirstlane(rs);
    const int run_end = row_ptr[last];
    const int L = pattern_len[blk];                       // wave-uniform
    const float inv_len = L ? 1.0f / (float)L : 0.0f;
    if (L) {
        if (lane < L) dist[lane] = col[run_begin + lane] - row0;
        __syncthreads();
    }
    T acc = 0;
    auto load_chunk = [&](int e0, int (&cc)[IPL], T (&vv)[IPL]) {
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const int at = min(max(e0 + lane + i * kWave, run_begin), run_end - 1);
            if (L) {
                const int rel = at - run_begin;
                const int r = (int)(((float)rel + 0.5f) * inv_len);      // rel / L, exact for rel < 2048
                cc[i] = row0 + r + dist[rel - r * L];
            } else {
                cc[i] = NT ? __builtin_nontemporal_load(col + at) : col[at];
            }
        }
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const T *p = val + min(max(e0 + lane + i * kWave, run_begin), run_end - 1);
            vv[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    int cc[IPL];
    T vv[IPL];
    const int first = run_begin & ~31;
    if (run_begin < run_end) load_chunk(first, cc, vv);
    for (int e0 = first; e0 < run_end; e0 += kCap) {
        const int n = min(kCap, run_end - e0);
        T xx[IPL];
        int cnext[IPL];
        T vnext[IPL];
#pragma unroll
        for (int i = 0; i < IPL; i++) xx[i] = x[cc[i]];
        if (e0 + kCap < run_end) load_chunk(e0 + kCap, cnext, vnext);
#pragma unroll
        for (int i = 0; i < IPL; i++) stage[lane + i * kWave] = xv_pair<T>{xx[i], vv[i]};
#pragma unroll
        for (int i = 0; i < IPL; i++) { cc[i] = cnext[i]; vv[i] = vnext[i]; }
        __syncthreads();
        int k = max(rs, e0) - e0;
        const int kend = min(re, e0 + n) - e0;
        for (; k + 4 <= kend; k += 4) {
            xv_pair<T> p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = stage[k + u];
#pragma unroll
            for (int u = 0; u < 4; u++) acc = fma_t(p[u].x, p[u].v, acc);
        }
        for (; k < kend; k++) acc = fma_t(stage[k].x, stage[k].v, acc);
        __syncthreads();
    }
    if (row < rows) y[row_map ? row_map[row] : row] = acc;
}

// --------------------------------------------------------- CSR, G lanes/row --

// Each group of G lanes owns R consecutive rows and walks them side by side, so
// that R (column, value) loads and R gathers per lane are in flight together;
// lane i of a row adds entries i, i+G, i+2G, ... and the G partial sums are
// combined by the DPP butterfly.  The first R lanes of the group store the R
// results, so a wave writes one contiguous run of 64/G*R elements of y.
template <typename T, int G, int R>
__global__ __launch_bounds__(kBlock) void csr_vector_kernel(int rows, const int *__restrict__ row_ptr,
                                                            const int *__restrict__ col,
                                                            const T *__restrict__ val,
                                                            const int *__restrict__ row_map,
                                                            const T *__restrict__ x, T *__restrict__ y, int map)
{
    static_assert(R <= G, "the first R lanes of a group store its R rows");
    constexpr int kRowsPerBlock = kBlock / G * R;
    // the grid is folded into two dimensions (one dimension holds fewer than 2^32 threads)
    const unsigned nblocks = (unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock);
    const unsigned bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (bid >= nblocks) return;
    const unsigned b = xcd_block(bid, nblocks, map);
    const int lane = threadIdx.x % G;
    const int row0 = (int)(b * kRowsPerBlock) + (int)(threadIdx.x / G) * R;
    int ptr[R + 1];
#pragma unroll
    for (int r = 0; r <= R; r++) ptr[r] = row_ptr[min(row0 + r, rows)];
    int longest = 0;
#pragma unroll
    for (int r = 0; r < R; r++) longest = max(longest, ptr[r + 1] - ptr[r]);
    T s[R];
#pragma unroll
    for (int r = 0; r < R; r++) s[r] = 0;
    for (int off = lane; off < longest; off += G) {
        int c[R];
        T v[R], xv[R];
#pragma unroll
        for (int r = 0; r < R; r++) c[r] = ptr[r] + off < ptr[r + 1] ? col[ptr[r] + off] : 0;
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = ptr[r] + off < ptr[r + 1] ? val[ptr[r] + off] : (T)0;
#pragma unroll
        for (int r = 0; r < R; r++) xv[r] = x[c[r]];
#pragma unroll
        for (int r = 0; r < R; r++)
            if (ptr[r] + off < ptr[r + 1]) s[r] = fma_t(xv[r], v[r], s[r]);
    }
    T mine = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const T total = group_sum<G>(s[r]);
        if (lane == r) mine = total;
    }
    const int row = row0 + lane;
    if (lane < R && row < rows) y[row_map ? row_map[row] : row] = mine;
}

// G lanes per row with the stream staged through LDS, for rows much shorter than a wave: the kernel
// above reads such rows in pieces of G entries, so every 128-byte line is fetched by several
// instructions one memory round trip apart and half-consumed lines of all resident waves crowd the
// L2.  Here a 64-lane workgroup owns 64 consecutive rows and walks their contiguous run of the
// arrays in line-aligned chunks of 64*IPL entries (each line loaded once, by one instruction), parks
// the products in LDS, and sums them with G lanes per row -- 64/G rows at a time, G passes over the
// 64 rows -- finishing with the DPP butterfly.  One aligned 64-element store of y per workgroup.
template <typename T, int IPL, int G, bool NT>
__global__ __launch_bounds__(kWave) void csr_vector_lds_kernel(int rows, const int *__restrict__ row_ptr,
                                                               const int *__restrict__ col,
                                                               const T *__restrict__ val,
                                                               const unsigned char *__restrict__ pattern_len,
                                                               const int *__restrict__ row_map,
                                                               const T *__restrict__ x, T *__restrict__ y, int map)
{
    constexpr int kCap = kWave * IPL, kRowsPerPass = kWave / G;
    __shared__ T prod[kCap];
    __shared__ int dist[32];
    const int blk = (int)xcd_block(blockIdx.x, gridDim.x, map);
    const int lane = threadIdx.x, sub = lane % G;
    const int row = blk * kWave + lane;
    const int last = min(blk * kWave + kWave, rows);
    const int rs = row_ptr[min(row, rows)];
    const int re = row < rows ? row_ptr[row + 1] : rs;
    const int run_begin = __builtin_amdgcn_readfirstlane(rs);
    const int run_end = row_ptr[last];
    // a block whose 64 rows repeat one pattern of L distances from the diagonal (pattern_len, null for the
    // variants that read every index) takes its column indices from the first row, as in
    // csr_rows_lds_pat_kernel
    const int L = pattern_len ? pattern_len[blk] : 0;        // wave-uniform
    const float inv_len = L ? 1.0f / (float)L : 0.0f;
    if (L) {
        if (lane < L) dist[lane] = col[run_begin + lane] - blk * kWave;
        __syncthreads();
    }
    int beg[G], end[G];                       // the row this lane helps with in pass p: p * 64/G + lane / G
    T acc[G];
#pragma unroll
    for (int p = 0; p < G; p++) {
        beg[p] = __shfl(rs, p * kRowsPerPass + lane / G, kWave);
        end[p] = __shfl(re, p * kRowsPerPass + lane / G, kWave);
        acc[p] = 0;
    }
    auto load_chunk = [&](int e0, int (&cc)[IPL], T (&vv)[IPL]) {
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const int at = min(max(e0 + lane + i * kWave, run_begin), run_end - 1);
            if (L) {
                const int rel = at - run_begin;
                const int r = (int)(((float)rel + 0.5f) * inv_len);      // rel / L, exact for rel < 2048
                cc[i] = blk * kWave + r + dist[rel - r * L];
            } else {
                cc[i] = NT ? __builtin_nontemporal_load(col + at) : col[at];
            }
        }
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const T *q = val + min(max(e0 + lane + i * kWave, run_begin), run_end - 1);
            vv[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
    };
    int cc[IPL];
    T vv[IPL];
    const int first = run_begin & ~31;
    if (run_begin < run_end) load_chunk(first, cc, vv);
    for (int e0 = first; e0 < run_end; e0 += kCap) {
        const int n = min(kCap, run_end - e0);
        T xx[IPL];
        int cnext[IPL];
        T vnext[IPL];
#pragma unroll
        for (int i = 0; i < IPL; i++) xx[i] = x[cc[i]];
        if (e0 + kCap < run_end) load_chunk(e0 + kCap, cnext, vnext);
#pragma unroll
        for (int i = 0; i < IPL; i++) prod[lane + i * kWave] = xx[i] * vv[i];
#pragma unroll
        for (int i = 0; i < IPL; i++) { cc[i] = cnext[i]; vv[i] = vnext[i]; }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < G; p++) {
            const int kend = min(end[p], e0 + n) - e0;
            for (int k = max(beg[p], e0) - e0 + sub; k < kend; k += G) acc[p] += prod[k];
        }
        __syncthreads();
    }
    // row p * 64/G + g is complete in the lanes of group g after the butterfly: park and store in row order
#pragma unroll
    for (int p = 0; p < G; p++) {
        const T total = group_sum<G>(acc[p]);
        if (sub == 0) prod[p * kRowsPerPass + lane / G] = total;
    }
    __syncthreads();
    if (row < rows) y[row_map ? row_map[row] : row] = prod[lane];
}

// ------------------------------------------- sliced formats, slice height 64 --
// One wave per slice.  Slice storage is column-major (reference
// sellcs_format.c:174-187, sellp.c:440-452): entry j of row k of the slice sits
// at slice_ptr[s] + j*64 + k, so a wave reads whole columns as contiguous runs.
//
// sliced64_kernel ("exact"): lane = row, one fma per column in ascending column
// order -- the summation order of mult_csr_base (reference csr.c:4-13) and of
// sellcs_mv_d_autovector (sellcs_mv_autovector.c:40-52), so y is bit-identical
// to both.  U columns (U*768 bytes per wave, fp64) are in flight per batch.

template <typename T, bool PERM, int U>
__global__ __launch_bounds__(kBlock) void sliced64_kernel(int rows, int nslices,
                                                          const int64_t *__restrict__ slice_ptr,
                                                          const int *__restrict__ row_order,
                                                          const int *__restrict__ scol,
                                                          const T *__restrict__ sval,
                                                          const T *__restrict__ x, T *__restrict__ y, int map)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    int lane = threadIdx.x & (kWave - 1);
    const int r = s * kWave + lane;
    // where the result goes: asked for before the stream, so that the store does not wait for a load of its own
    const int out_row = PERM ? row_order[min(r, rows - 1)] : r;
    int64_t base = slice_ptr[s];
    int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const T *v = sval + base + lane;
    const int *c = scol + base + lane;
    T acc = 0;
    int j = 0;
    for (; j + U <= width; j += U) {
        int cc[U];
        T vv[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = c[(int64_t)(j + u) * kWave];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = v[(int64_t)(j + u) * kWave];
#pragma unroll
        for (int u = 0; u < U; u++) xx[u] = x[cc[u]];
#pragma unroll
        for (int u = 0; u < U; u++) acc = fma_t(xx[u], vv[u], acc);
    }
    for (; j < width; j++) acc = fma_t(x[c[(int64_t)j * kWave]], v[(int64_t)j * kWave], acc);
    if (r < rows) y[out_row] = acc;
}

// sliced64_wide_kernel: the same slices read with 16-byte loads.  Lane l =
// (h = l / HL, q = l % HL) holds the R = 16/sizeof(T) adjacent rows q*R..q*R+R-1
// of column P*jj + h, so one wave instruction moves P = R whole columns (1 KiB
// of values).  A batch of U such steps is loaded, gathered and accumulated at
// once: with U*P >= slice width the whole slice costs one streaming round trip
// and one gather round trip.  Each lane sums the columns == h (mod P) of its
// rows; the P partial sums of a row are then added across the lane groups, so
// the in-row order differs from mult_csr_base (parity by tolerance, not bits).
// Columns past the slice width are read from the last column (no extra HBM
// traffic) and left out of the sums.
template <typename T, bool PERM, int U, bool NT>
__global__ __launch_bounds__(kBlock) void sliced64_wide_kernel(int rows, int nslices,
                                                               const int64_t *__restrict__ slice_ptr,
                                                               const int *__restrict__ row_order,
                                                                    const int *__restrict__ scol,
                                                               const T *__restrict__ sval,
                                                               const T *__restrict__ x, T *__restrict__ y, int map)
{
    constexpr int R = 16 / sizeof(T);        // rows per lane: 2 (fp64), 4 (fp32)
    constexpr int HL = kWave / R;            // lanes per column: 32, 16
    constexpr int P = R;                     // columns per wave instruction: 2, 4
    typedef T TV __attribute__((ext_vector_type(R)));
    typedef int IV __attribute__((ext_vector_type(R)));
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int h = lane / HL, q = lane % HL;
    const int r = s * kWave + lane;
    // where the result goes: asked for before the stream, so that the store does not wait for a load of its own
    const int out_row = PERM ? row_order[min(r, rows - 1)] : r;
    int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const T *v = sval + base + q * R;
    const int *c = scol + base + q * R;
    TV acc = (TV)(0);
    const int steps = (width + P - 1) / P;
    for (int jj = 0; jj < steps; jj += U) {
        IV cc[U];
        TV vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const IV *p = (const IV *)(c + (int64_t)min(P * (jj + u) + h, width - 1) * kWave);
            cc[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const TV *p = (const TV *)(v + (int64_t)min(P * (jj + u) + h, width - 1) * kWave);
            vv[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        TV xx[U];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < R; k++) xx[u][k] = x[cc[u][k]];
        // a step whose P columns all exist needs no mask (wave-uniform test); only the slice's last step
        // can hold columns past the width, and those are left out of the sums, not multiplied by 0: x may
        // be non-finite
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (P * (jj + u) + P <= width) {
#pragma unroll
                for (int k = 0; k < R; k++) acc[k] = fma_t(xx[u][k], vv[u][k], acc[k]);
            } else if (P * (jj + u) < width) {
                const bool live = P * (jj + u) + h < width;
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const T t = fma_t(xx[u][k], vv[u][k], acc[k]);
                    acc[k] = live ? t : acc[k];
                }
            }
        }
    }
    // add the P column groups, then hand row l to lane l for one coalesced store
    T mine = 0;
#pragma unroll
    for (int k = 0; k < R; k++) {
        T t = acc[k];
        if (P >= 4) t += __shfl_xor(t, 16, kWave);
        t += __shfl_xor(t, 32, kWave);
        T from = __shfl(t, lane / R, kWave);
        if ((lane % R) == k) mine = from;
    }
    if (r < rows) y[out_row] = mine;
}

// sliced64_wide_exact_kernel (fp64): the 16-byte loads of sliced64_wide_kernel with the summation order of
// sliced64_kernel.  Lane (h, q) loads the rows 2q, 2q+1 of column 2jj + h; one v_permlane32_swap per 32-bit
// half then trades the upper half-wave's row-2q entry against the lower half-wave's row-(2q+1) entry, after
// which lane q < 32 holds row 2q's entries of columns 2jj and 2jj + 1 and lane 32 + q those of row 2q + 1.
// Every lane then adds its one row's entries in ascending column order, one fma each: the order and operands of
// mult_csr_base (csr.c:4-13), so y is bit-identical to it -- at the memory shape of the wide kernel.
__device__ __forceinline__ void swap_half_waves(double &a, double &b)
{
    // a's lanes 32..63 <-> b's lanes 0..31
    const long long ab = __builtin_bit_cast(long long, a), bb = __builtin_bit_cast(long long, b);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)(ab & 0xFFFFFFFFll), (unsigned)(bb & 0xFFFFFFFFll), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
    a = __builtin_bit_cast(double, ((long long)hi[0] << 32) | (long long)lo[0]);
    b = __builtin_bit_cast(double, ((long long)hi[1] << 32) | (long long)lo[1]);
}

template <bool PERM, int U, bool NT>
__global__ __launch_bounds__(kBlock) void sliced64_wide_exact_kernel(int rows, int nslices,
                                                                     const int64_t *__restrict__ slice_ptr,
                                                                     const int *__restrict__ row_order,
                                                                     const int *__restrict__ scol,
                                                                     const double *__restrict__ sval,
                                                                     const double *__restrict__ x, double *__restrict__ y,
                                                                     int map)
{
    typedef double TV __attribute__((ext_vector_type(2)));
    typedef int IV __attribute__((ext_vector_type(2)));
    constexpr int HL = kWave / 2;
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int h = lane / HL, q = lane % HL;
    // the row this lane sums after the trade: 2q (lower half-wave) or 2q + 1 (upper)
    const int r = s * kWave + 2 * q + h;
    const int out_row = PERM ? row_order[min(r, rows - 1)] : r;
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const double *v = sval + base + q * 2;
    const int *c = scol + base + q * 2;
    double acc = 0;
    const int steps = (width + 1) / 2;
    for (int jj = 0; jj < steps; jj += U) {
        IV cc[U];
        TV vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const IV *p = (const IV *)(c + (int64_t)min(2 * (jj + u) + h, width - 1) * kWave);
            cc[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const TV *p = (const TV *)(v + (int64_t)min(2 * (jj + u) + h, width - 1) * kWave);
            vv[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        TV xx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            xx[u][0] = x[cc[u][0]];
            xx[u][1] = x[cc[u][1]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double x0 = xx[u][0], x1 = xx[u][1], v0 = vv[u][0], v1 = vv[u][1];
            swap_half_waves(x0, x1);
            swap_half_waves(v0, v1);
            // now (x0, v0) is this lane's row at column 2 (jj + u), (x1, v1) at the column after it
            const int col = 2 * (jj + u);
            if (col < width) acc = fma_t(x0, v0, acc);
            if (col + 1 < width) acc = fma_t(x1, v1, acc);
        }
    }
    if (r < rows) y[out_row] = acc;
}

// sliced64_diag_wide_kernel (fp64): sliced64_diag_kernel with the memory shape of sliced64_wide_exact_kernel.  Lane
// (h, q) loads the values of rows 2q, 2q+1 of column 2jj + h with one 16-byte load; a column with a descriptor
// takes its x operands from x[row + d] (its 64 indices stay unread), any other column reads its two indices; the
// half-wave swap then hands each row's entries of columns 2jj, 2jj + 1 to one lane, which adds them in ascending
// column order.  Same operands and order as sliced64_diag_kernel (padding lanes of described columns take x[0]),
// so the same bits.  Where a lane's two rows are neighbours their operands of a described column are neighbours
// in x too and come with one 16-byte gather.
template <bool PERM, int U, bool NT>
__global__ __launch_bounds__(kBlock) void sliced64_diag_wide_kernel(int rows, int cols, int nslices,
                                                                    const int64_t *__restrict__ slice_ptr,
                                                                    const int *__restrict__ row_order,
                                                                    const int *__restrict__ sdiag,
                                                                    const int *__restrict__ scol,
                                                                    const double *__restrict__ sval,
                                                                    const double *__restrict__ x, double *__restrict__ y,
                                                                    int map)
{
    typedef double TV __attribute__((ext_vector_type(2)));
    typedef int IV __attribute__((ext_vector_type(2)));
    constexpr int HL = kWave / 2;
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int h = lane / HL, q = lane % HL;
    // the two rows whose entries this lane loads, and the one it sums after the trade (2q + h)
    const int ra = s * kWave + 2 * q;
    int row_a, row_b;
    if (PERM) {
        if (ra + 1 < rows) {
            const IV rr = *(const IV *)(row_order + ra);
            row_a = rr[0];
            row_b = rr[1];
        } else {
            row_a = row_b = row_order[min(ra, rows - 1)];
        }
    } else {
        row_a = min(ra, rows - 1);
        row_b = min(ra + 1, rows - 1);
    }
    const int r = ra + h;
    const int out_row = h ? row_b : row_a;
    // a slice whose row pairs are all neighbours (always without a row order; with one, wherever the window sort
    // left no hole) takes the two x operands of a diagonal column, x[row + d] and x[row + 1 + d], with one 16-byte load
    const bool all_adjacent = __ballot(row_b != row_a + 1) == 0;                    // wave-uniform
    typedef double TVU __attribute__((ext_vector_type(2), aligned(8)));
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const double *v = sval + base + q * 2;
    const int *c = scol + base + q * 2;
    const int *dg = sdiag + (base >> 6);
    const int last = max(cols - 1, 0);
    const double x0 = width > 0 ? x[0] : 0.0;
    double acc = 0;
    const int steps = (width + 1) / 2;
    for (int jj = 0; jj < steps; jj += U) {
        IV cc[U];
        TV vv[U];
        bool dia0[U], dia1[U], pair[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const TV *p = (const TV *)(v + (int64_t)min(2 * (jj + u) + h, width - 1) * kWave);
            vv[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j0 = min(2 * (jj + u), width - 1), j1 = min(2 * (jj + u) + 1, width - 1);
            const int d0 = __builtin_amdgcn_readfirstlane(dg[j0]), d1 = __builtin_amdgcn_readfirstlane(dg[j1]);
            dia0[u] = d0 != INT_MIN;
            dia1[u] = d1 != INT_MIN;
            pair[u] = false;
            if (dia0[u] && dia1[u]) {
                const int d = h ? d1 : d0;
                const int ca = row_a + d;
                pair[u] = all_adjacent && __ballot(!(ca >= 0 && ca < last)) == 0;   // wave-uniform
                cc[u][0] = min(max(ca, 0), last);
                cc[u][1] = min(max(row_b + d, 0), last);
            } else {
                const IV *p = (const IV *)(c + (int64_t)(h ? j1 : j0) * kWave);
                const IV ld = NT ? __builtin_nontemporal_load(p) : *p;
                const int d = h ? d1 : d0;
                const bool mine = d != INT_MIN;
                cc[u][0] = mine ? min(max(row_a + d, 0), last) : ld[0];
                cc[u][1] = mine ? min(max(row_b + d, 0), last) : ld[1];
            }
        }
        TV xx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (pair[u]) {
                const TVU two = *(const TVU *)(x + cc[u][0]);
                xx[u][0] = two[0];
                xx[u][1] = two[1];
            } else {
                xx[u][0] = x[cc[u][0]];
                xx[u][1] = x[cc[u][1]];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double xa = xx[u][0], xb = xx[u][1], va = vv[u][0], vb = vv[u][1];
            swap_half_waves(xa, xb);
            swap_half_waves(va, vb);
            // (xa, va): this lane's row at column 2 (jj + u); (xb, vb): at the column after it
            const int col = 2 * (jj + u);
            if (col < width) acc = fma_t(dia0[u] && va == 0.0 ? x0 : xa, va, acc);
            if (col + 1 < width) acc = fma_t(dia1[u] && vb == 0.0 ? x0 : xb, vb, acc);
        }
    }
    if (r < rows) y[out_row] = acc;
}

// sliced64_diag_kernel: the exact-order kernel for slices whose columns run along diagonals.  The
// build pass left one number per slice column (sdiag): the distance col - row when all real entries
// of the column share it, INT_MIN otherwise.  For a diagonal column the 64 column indices are not
// read: lane k gathers x[row_k + d] (padding lanes carry value 0 and clamp the address into x).  A
// 27-point stencil, a band or a block-diagonal matrix then streams 8 (fp64) instead of 12 bytes per
// entry.  The descriptor is wave-uniform, so the choice is a scalar branch.  One fma per column in
// ascending column order: bit-identical to sliced64_kernel and mult_csr_base (csr.c:4-13).
// A padding entry is stored as (column 0, value 0) (sellcs_format.c:167,171, sellp.c:440-452), so
// the reference multiplies x[0] by 0 there -- which is NaN for a non-finite x[0].  The build pass
// gives a descriptor only to columns whose real entries all have a non-zero value; in such a column
// a zero value marks a padding lane, and that lane takes x[0] as its operand, so y carries the
// reference's bits for non-finite x too.
template <typename T, bool PERM, int U, bool NT>
__global__ __launch_bounds__(kBlock) void sliced64_diag_kernel(int rows, int cols, int nslices,
                                                               const int64_t *__restrict__ slice_ptr,
                                                               const int *__restrict__ row_order,
                                                               const int *__restrict__ sdiag,
                                                               const int *__restrict__ scol,
                                                               const T *__restrict__ sval,
                                                               const T *__restrict__ x, T *__restrict__ y, int map)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int r = s * kWave + lane;
    const int row = PERM ? row_order[min(r, rows - 1)] : min(r, rows - 1);
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const T *v = sval + base + lane;
    const int *c = scol + base + lane;
    const int *dg = sdiag + (base >> 6);
    const int last = max(cols - 1, 0);
    const T x0 = width > 0 ? x[0] : (T)0;           // a slice has entries only when the matrix has columns
    T acc = 0;
    int j = 0;
    // B columns at once: value loads, descriptors / index loads, gathers, then the fmas in column order
    auto batch = [&](auto bsize) {
        constexpr int B = decltype(bsize)::value;
        int cc[B];
        bool dia[B];
        T vv[B], xx[B];
#pragma unroll
        for (int u = 0; u < B; u++) vv[u] = NT ? __builtin_nontemporal_load(v + (int64_t)(j + u) * kWave) : v[(int64_t)(j + u) * kWave];
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int d = __builtin_amdgcn_readfirstlane(dg[j + u]);
            dia[u] = d != INT_MIN;
            if (dia[u]) cc[u] = min(max(row + d, 0), last);
            else cc[u] = NT ? __builtin_nontemporal_load(c + (int64_t)(j + u) * kWave) : c[(int64_t)(j + u) * kWave];
        }
#pragma unroll
        for (int u = 0; u < B; u++) xx[u] = x[cc[u]];
#pragma unroll
        for (int u = 0; u < B; u++) acc = fma_t(dia[u] && vv[u] == (T)0 ? x0 : xx[u], vv[u], acc);
        j += B;
    };
    while (j + U <= width) batch(std::integral_constant<int, U>());
    if (U > 4)
        while (j + 4 <= width) batch(std::integral_constant<int, 4>());
    while (j < width) batch(std::integral_constant<int, 1>());
    if (r < rows) y[row] = acc;
}

// ----------------------------------------- SELL-C-sigma without padding (DFC) --
// One wave per slice, lane = row (sorted position).  Column v of the slice holds one entry per row
// longer than v; the rows of a slice are sorted by length, so these are lanes 0..active-1 and the
// column is one contiguous run starting where the previous column ended.  The run start is carried
// as a wave-uniform counter (ballot + popcount), so neither vop_len nor vop_ptr is read here.  One
// fma per entry in ascending column order: the summation order of mult_csr_base (csr.c:4-13).
// DIAG: columns whose active lanes all sit at one distance from the diagonal (sdiag, one int per
// column, INT_MIN otherwise; vop_ptr = first column of each slice) are gathered as x[row + d]
// without reading their indices.
template <typename T, int U, bool NT, bool DIAG = false>
__global__ __launch_bounds__(kBlock) void sell_dfc_kernel(int rows, int nslices,
                                                          const int64_t *__restrict__ slice_ptr,
                                                          const int *__restrict__ sorted_len,
                                                          const int *__restrict__ row_order,
                                                          const int *__restrict__ scol,
                                                          const T *__restrict__ sval,
                                                          const T *__restrict__ x, T *__restrict__ y,
                                                          const int64_t *__restrict__ vop_ptr = nullptr,
                                                          const int *__restrict__ sdiag = nullptr, int map = 0)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int r = s * kWave + lane;
    const int len = r < rows ? sorted_len[r] : 0;
    const int width = __builtin_amdgcn_readfirstlane(len);      // lane 0 holds the slice's longest row
    const int *c = scol + slice_ptr[s] + lane;
    const T *v = sval + slice_ptr[s] + lane;
    const int shortest = __builtin_amdgcn_readlane(len, kWave - 1);   // columns below it hold all 64 lanes
    const int row = r < rows ? row_order[r] : 0;                // also where y goes: asked for before the stream
    const int *dg = DIAG ? sdiag + vop_ptr[s] : nullptr;
    int run = 0;                                                // entries of the slice before column j
    T acc = 0;
    int j0 = 0;
    for (; j0 + U <= shortest; j0 += U) {                       // full columns: plain strided slice, no masks
        int cc[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int d = DIAG ? __builtin_amdgcn_readfirstlane(dg[j0 + u]) : INT_MIN;
            if (d != INT_MIN) cc[u] = row + d;
            else cc[u] = NT ? __builtin_nontemporal_load(c + run + u * kWave) : c[run + u * kWave];
        }
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = NT ? __builtin_nontemporal_load(v + run + u * kWave) : v[run + u * kWave];
#pragma unroll
        for (int u = 0; u < U; u++) acc = fma_t(x[cc[u]], vv[u], acc);
        run += U * kWave;
    }
    for (; j0 < width; j0 += U) {
        int off[U], cc[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            off[u] = run;
            run += __popcll(__ballot(j0 + u < len));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int d = DIAG && j0 + u < width ? __builtin_amdgcn_readfirstlane(dg[j0 + u]) : INT_MIN;
            if (d != INT_MIN) cc[u] = j0 + u < len ? row + d : 0;
            else cc[u] = j0 + u < len ? (NT ? __builtin_nontemporal_load(c + off[u]) : c[off[u]]) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            vv[u] = j0 + u < len ? (NT ? __builtin_nontemporal_load(v + off[u]) : v[off[u]]) : (T)0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const T xv = x[cc[u]];
            if (j0 + u < len) acc = fma_t(xv, vv[u], acc);
        }
    }
    if (r < rows) y[row] = acc;
}

// The same product with the partial columns staged through LDS.  The columns below the slice's
// shortest row are read as above.  The rest of the slice is one contiguous run of entries: it is
// loaded in line-aligned chunks of 64*IPL entries by all 64 lanes, gathered, parked in LDS as (x, value)
// pairs, and then walked column by column -- lane k takes entry k of the column while the column has
// one.  Every HBM access is a full-width one whatever the number of active lanes, so slices with a
// long thin tail cost their entries, not their width.  Same summation order.
template <typename T, int IPL, bool NT>
__global__ __launch_bounds__(kWave) void sell_dfc_lds_kernel(int rows, int nslices,
                                                             const int64_t *__restrict__ slice_ptr,
                                                             const int *__restrict__ sorted_len,
                                                             const int *__restrict__ row_order,
                                                             const int *__restrict__ scol,
                                                             const T *__restrict__ sval,
                                                             const T *__restrict__ x, T *__restrict__ y, int map)
{
    constexpr int kCap = kWave * IPL, U = 8;
    __shared__ xv_pair<T> stage[kCap];
    const int s = (int)xcd_block(blockIdx.x, gridDim.x, map);
    const int lane = threadIdx.x;
    const int r = s * kWave + lane;
    const int len = r < rows ? sorted_len[r] : 0;
    const int out_row = r < rows ? row_order[r] : 0;            // asked for before the stream
    const int width = __builtin_amdgcn_readfirstlane(len);
    const int shortest = __builtin_amdgcn_readlane(len, kWave - 1);
    const int64_t base = slice_ptr[s], slice_end = slice_ptr[s + 1];
    const int *c = scol + base + lane;
    const T *v = sval + base + lane;
    int64_t run = 0;
    T acc = 0;
    int j = 0;
    for (; j + U <= shortest; j += U) {
        int cc[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = NT ? __builtin_nontemporal_load(c + run + u * kWave) : c[run + u * kWave];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = NT ? __builtin_nontemporal_load(v + run + u * kWave) : v[run + u * kWave];
#pragma unroll
        for (int u = 0; u < U; u++) acc = fma_t(x[cc[u]], vv[u], acc);
        run += U * kWave;
    }
    while (j < width) {
        const int64_t g0 = base + run;                         // first entry of column j in the arrays
        const int64_t ga = g0 & ~(int64_t)31;                  // chunk start on a 128-byte line of scol
        int cc[IPL];
        T vv[IPL];
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const int64_t at = min(max(ga + lane + i * kWave, g0), slice_end - 1);
            cc[i] = NT ? __builtin_nontemporal_load(scol + at) : scol[at];
            vv[i] = NT ? __builtin_nontemporal_load(sval + at) : sval[at];
        }
#pragma unroll
        for (int i = 0; i < IPL; i++) stage[lane + i * kWave] = xv_pair<T>{x[cc[i]], vv[i]};
        __syncthreads();
        int pos = (int)(g0 - ga);                              // LDS position of column j
        while (j < width) {
            if (j + 4 <= width) {                              // four columns at a time while they fit
                int p[4], q = pos;
#pragma unroll
                for (int u = 0; u < 4; u++) { p[u] = q; q += __popcll(__ballot(j + u < len)); }
                if (q <= kCap) {
                    xv_pair<T> e[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) e[u] = stage[min(p[u] + lane, kCap - 1)];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (j + u < len) acc = fma_t(e[u].x, e[u].v, acc);
                    run += q - pos;
                    pos = q;
                    j += 4;
                    continue;
                }
            }
            const int a = __popcll(__ballot(j < len));
            if (pos + a > kCap) break;                         // column j straddles the chunk: reload from it
            if (j < len) { const xv_pair<T> e = stage[pos + lane]; acc = fma_t(e.x, e.v, acc); }
            pos += a;
            run += a;
            j++;
        }
        __syncthreads();
    }
    if (r < rows) y[out_row] = acc;
}

// ------------------------------------- sliced formats, any slice height <= 64 --
// Lane per row; used for the reference's CPU-shaped slice heights (2, 4, 8).

template <typename T, bool PERM>
__global__ __launch_bounds__(kBlock) void sliced_any_kernel(int rows, int height,
                                                            const int64_t *__restrict__ slice_ptr,
                                                            const int *__restrict__ row_order,
                                                            const int *__restrict__ scol,
                                                            const T *__restrict__ sval,
                                                            const T *__restrict__ x, T *__restrict__ y, int map)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int r = (int)(b * kBlock + threadIdx.x);
    if (r >= rows) return;
    int s = r / height, k = r - s * height;
    const int out_row = PERM ? row_order[r] : r;
    int64_t base = slice_ptr[s];
    int width = (int)((slice_ptr[s + 1] - base) / height);
    T acc = 0;
    for (int j = 0; j < width; j++) {
        int64_t at = base + (int64_t)j * height + k;
        acc = fma_t(x[scol[at]], sval[at], acc);
    }
    y[out_row] = acc;
}

// ------------------------------------------------ merge-path / balanced tiles --
// A tile is the path segment between two precomputed coordinates (a = position
// in the row-end list, b = position in the non-zero list):
// tile_coord[2t..2t+1] -> tile_coord[2t+2..2t+3].
//   merge-path CSR: row-end list = row_ptr+1, tiles are equal shares of the
//     rows+nnz merge path; the coordinates come from the reference's 2-D
//     diagonal search (csr_merge.c:18-30, 62-85) done once at create time,
//     where the reference freezes its thread count too (csr_merge.c:93).
//   balanced CSR:   row-end list = ends of the non-empty rows, tiles are equal
//     shares of the non-zeros (reference csr_bal.c:83-85); row_index maps a
//     list position back to its row.
// Phase 1 stages the tile's row ends (the row pointers) and its products
// val*x[col] in LDS with fully coalesced loads.  Phase 2 sums whole rows out of
// LDS (csr_merge.c:32-41).  The tile's trailing partial row goes to
// (row_carry, val_carry) and is added by tile_fixup_kernel in tile order
// (csr_merge.c:43-58, csr_bal.c:33-37).
//
// One 64-lane workgroup per tile of 64*IPL items, so waves never wait for each
// other.  Phase 2 gives every row that ends in the tile (plus the trailing
// partial row) to one lane, which adds the row's products in ascending order;
// rows longer than kLongRow entries are summed by the whole wave instead.  A
// tile that lies inside a single row reduces its registers directly.
constexpr int kLongRow = 48;

template <typename T, int IPL, int MODE>
__global__ __launch_bounds__(kWave) void wave_tile_kernel(int list_len, const int *__restrict__ row_end_list,
                                                          const int *__restrict__ row_index,
                                                          const int *__restrict__ tile_coord,
                                                          const unsigned char *__restrict__ tile_pat,
                                                          const int *__restrict__ col,
                                                          const T *__restrict__ val,
                                                          const T *__restrict__ x, T *__restrict__ y,
                                                          int *__restrict__ row_carry, T *__restrict__ val_carry, int map)
{
    constexpr int kCap = kWave * IPL;
    __shared__ T prod[kCap];
    __shared__ int rend[kCap];
    __shared__ int dist[32];
    const int t = (int)xcd_block(blockIdx.x, gridDim.x, map);
    const int lane = threadIdx.x;
    const int a0 = tile_coord[2 * t], b0 = tile_coord[2 * t + 1];
    const int a1 = tile_coord[2 * t + 2], b1 = tile_coord[2 * t + 3];
    const int nr = a1 - a0, nz = b1 - b0;

    // phase 1: coalesced (column, value) loads, then the gathers, all in flight together.  The loads
    // start on a 32-entry boundary of the arrays so that each instruction covers whole 128-byte
    // lines (one more, mostly idle, load per lane than the tile has items per lane); lanes outside
    // the tile repeat its first or last entry.
    constexpr int kLoads = MODE ? IPL + 1 : IPL;
    const int first = MODE ? (b0 & ~31) - b0 : 0;   // position of lane 0's first load relative to the tile
    T pp[kLoads];
    // a tile whose rows repeat one pattern of L distances from the diagonal (tile_pat, null for the
    // variants that read every index) takes its column indices from its first row: the entry at
    // position p of the arrays is entry (p - start0) % L of row row0 + (p - start0) / L
    const int L = tile_pat ? tile_pat[t] : 0;                // wave-uniform
    int start0 = 0, row0 = 0;
    if (L) {
        start0 = a0 ? row_end_list[a0 - 1] : 0;
        row0 = row_index ? row_index[a0] : a0;
        if (lane < L) dist[lane] = col[start0 + lane] - row0;
        __syncthreads();
    }
    const float inv_len = L ? 1.0f / (float)L : 0.0f;
    if (nz > 0) {
        int cc[kLoads];
        T vv[kLoads], xx[kLoads];
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const int k = min(max(first + lane + i * kWave, 0), nz - 1);
            if (L) {
                const int rel = b0 + k - start0;
                const int r = (int)(((float)rel + 0.5f) * inv_len);      // rel / L, exact for rel < 4096
                cc[i] = row0 + r + dist[rel - r * L];
            } else {
                cc[i] = MODE == 2 ? __builtin_nontemporal_load(col + b0 + k) : col[b0 + k];
            }
        }
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const T *p = val + b0 + min(max(first + lane + i * kWave, 0), nz - 1);
            vv[i] = MODE == 2 ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < kLoads; i++) xx[i] = x[cc[i]];
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const int k = first + lane + i * kWave;
            pp[i] = k >= 0 && k < nz ? vv[i] * xx[i] : (T)0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kLoads; i++) pp[i] = 0;
    }

    if (nr == 0) {                               // the tile is a piece of one row
        T s = 0;
#pragma unroll
        for (int i = 0; i < kLoads; i++) s += pp[i];
        s = group_sum<kWave>(s);
        if (lane == 0) { row_carry[t] = a1; val_carry[t] = s; }
        return;
    }

#pragma unroll
    for (int i = 0; i < kLoads; i++) {
        const int k = first + lane + i * kWave;
        if (k >= 0 && k < nz) prod[k] = pp[i];
    }
#pragma unroll
    for (int i = 0; i < IPL; i++) {
        const int k = lane + i * kWave;
        if (k < nr) rend[k] = row_end_list[a0 + k] - b0;
    }
    __syncthreads();

    for (int base = 0; base <= nr; base += kWave) {
        const int rr = base + lane;              // row slot: rr < nr ends here, rr == nr is the trailing partial row
        const bool valid = rr <= nr;
        int start = 0, end = 0;
        if (valid) {
            start = rr == 0 ? 0 : rend[rr - 1];
            end = rr < nr ? rend[rr] : nz;
        }
        const int len = end - start;
        T s = 0;
        if (len <= kLongRow) {
            int k = start;
            for (; k + 8 <= end; k += 8) {           // 8 LDS reads in flight, added in ascending order
                T p[8];
#pragma unroll
                for (int u = 0; u < 8; u++) p[u] = prod[k + u];
#pragma unroll
                for (int u = 0; u < 8; u++) s += p[u];
            }
            for (; k < end; k++) s += prod[k];
        }
        unsigned long long longrows = __ballot(len > kLongRow);
        while (longrows) {
            const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(longrows));
            longrows &= longrows - 1;
            const int ls = __builtin_amdgcn_readlane(start, src), le = __builtin_amdgcn_readlane(end, src);
            T part = 0;
            for (int k = ls + lane; k < le; k += kWave) part += prod[k];
            part = group_sum<kWave>(part);
            if (lane == src) s = part;
        }
        if (valid) {
            if (rr < nr) y[row_index ? row_index[a0 + rr] : a0 + rr] = s;
            else { row_carry[t] = a1; val_carry[t] = s; }
        }
    }
}

// Adds the tiles' trailing partial sums to the rows they belong to, in tile
// order (reference csr_merge.c:54-58).  One thread per run of equal row_carry.
template <typename T>
__global__ void tile_fixup_kernel(int ntiles, int list_len, const int *__restrict__ row_index,
                                  const int *__restrict__ row_carry, const T *__restrict__ val_carry,
                                  T *__restrict__ y)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    int a = row_carry[t];
    if (a >= list_len) return;
    if (t > 0 && row_carry[t - 1] == a) return;
    int row = row_index ? row_index[a] : a;
    T acc = y[row];
    for (int u = t; u < ntiles && row_carry[u] == a; u++) acc += val_carry[u];
    y[row] = acc;
}

// ---- balanced tiles with the hottest columns of each column group in LDS ----
//
// For the stacked matrix of the option "hot_columns" = 2 (sort.hip): piece g of the tiles only names
// the labels g*cpx .. g*cpx+cpx-1 of the permuted x, the most referenced first.  A gather that hits
// the L2 still costs a 128-byte line transfer into the L1 for 8 bytes of x, and on a power-law
// matrix those transfers, not HBM, take most of the time of the hits.  Here the workgroups are
// persistent, one per CU: workgroup b serves group b % 8 (workgroups b, b+8, ... share an XCD), loads
// the group's first kHot labels of x into LDS once and then walks its share of the group's tiles, W
// tiles at a time, one per wave.  A column below the mark is read from LDS, any other (the cold
// rest, or another group's in a tile that straddles two pieces) from memory as before.  Everything
// after the gather is wave_tile_kernel's (line-aligned loads, products, row sums in ascending
// order, carries), so y has the bits of that kernel's MODE 2.  The waves never wait for each other
// after the first barrier: each owns its staging arrays, and LDS operations of one wave complete
// in order.
struct tile_groups { int first_tile[9]; };

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int IPL, int W> struct hot_tile_shape {
    static constexpr int kCap = kWave * IPL;
    static constexpr int kStageBytes = W * kCap * (int)(sizeof(T) + sizeof(unsigned short));   // products + row ends (<= kCap: 16 bits)
    static constexpr int kHot = (160 * 1024 - kStageBytes) / (int)sizeof(T);      // all of the CU's LDS
};

template <typename T, int IPL, int W>
__global__ __launch_bounds__(kWave * W) void wave_tile_hot_kernel(const int *__restrict__ row_end_list,
                                                                  const int *__restrict__ row_index,
                                                                  const int *__restrict__ tile_coord,
                                                                  const int *__restrict__ col,
                                                                  const T *__restrict__ val,
                                                                  const T *__restrict__ x, T *__restrict__ y,
                                                                  int *__restrict__ row_carry,
                                                                  T *__restrict__ val_carry, tile_groups groups,
                                                                  int cpx, int nnz)
{
    using shape = hot_tile_shape<T, IPL, W>;
    constexpr int kCap = shape::kCap;
    constexpr int kHot = shape::kHot;
    __shared__ T hot[kHot];
    __shared__ T prod_of_wave[W][kCap];
    __shared__ unsigned short rend_of_wave[W][kCap];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int g = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int hot_base = g * cpx;
    const int nhot = min(kHot, cpx);
    for (int i = threadIdx.x; i < nhot; i += kWave * W) hot[i] = x[hot_base + i];
    __syncthreads();
    T *prod = prod_of_wave[wave];
    unsigned short *rend = rend_of_wave[wave];
    const int t_begin = groups.first_tile[g] + slot * W + wave, t_end = groups.first_tile[g + 1];
    const int stride = slots * W;
    constexpr int kLoads = IPL + 1;
    // balanced tiles are equal shares of the entries: tile t holds entries t*kCap .. min(t*kCap + kCap, nnz) - 1
    auto issue_loads = [&](int t, int (&cc)[kLoads], T (&vv)[kLoads]) {
        const int b0 = (int)min((int64_t)t * kCap, (int64_t)nnz - 1);
        const int nz = min(kCap, nnz - b0);
        const int first = (b0 & ~31) - b0;
#pragma unroll
        for (int i = 0; i < kLoads; i++)
            cc[i] = __builtin_nontemporal_load(col + b0 + min(max(first + lane + i * kWave, 0), nz - 1));
#pragma unroll
        for (int i = 0; i < kLoads; i++)
            vv[i] = __builtin_nontemporal_load(val + b0 + min(max(first + lane + i * kWave, 0), nz - 1));
    };
    // One tile: its gathers, then the stream loads of the wave's next tile (issued whether or not there is
    // one -- the last tile's are repeated -- so that they are in flight while the gathers are waited for),
    // then products, row sums and carries as in wave_tile_kernel.
    auto tile = [&](int t, int (&cc)[kLoads], T (&vv)[kLoads], int (&cn)[kLoads], T (&vn)[kLoads]) {
        const int a0 = tile_coord[2 * t], b0 = tile_coord[2 * t + 1];
        const int a1 = tile_coord[2 * t + 2], b1 = tile_coord[2 * t + 3];
        const int nr = a1 - a0, nz = b1 - b0;
        const int first = (b0 & ~31) - b0;
        T xx[kLoads], pp[kLoads];
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const unsigned rel = (unsigned)(cc[i] - hot_base);
            if (rel < (unsigned)nhot) xx[i] = hot[rel];
            else xx[i] = x[cc[i]];
        }
        issue_loads(min(t + stride, t_end - 1), cn, vn);
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const int k = first + lane + i * kWave;
            pp[i] = k >= 0 && k < nz ? vv[i] * xx[i] : (T)0;
        }
        if (nr == 0) {                               // the tile is a piece of one row
            T s = 0;
#pragma unroll
            for (int i = 0; i < kLoads; i++) s += pp[i];
            s = group_sum<kWave>(s);
            if (lane == 0) { row_carry[t] = a1; val_carry[t] = s; }
            return;
        }
        wave_lds_fence();                            // the previous tile's row sums have read their products
#pragma unroll
        for (int i = 0; i < kLoads; i++) {
            const int k = first + lane + i * kWave;
            if (k >= 0 && k < nz) prod[k] = pp[i];
        }
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const int k = lane + i * kWave;
            if (k < nr) rend[k] = (unsigned short)(row_end_list[a0 + k] - b0);
        }
        wave_lds_fence();
        for (int base = 0; base <= nr; base += kWave) {
            const int rr = base + lane;              // row slot: rr < nr ends here, rr == nr is the trailing partial row
            const bool valid = rr <= nr;
            int start = 0, end = 0;
            if (valid) {
                start = rr == 0 ? 0 : rend[rr - 1];
                end = rr < nr ? rend[rr] : nz;
            }
            const int len = end - start;
            T s = 0;
            if (len <= kLongRow) {
                int k = start;
                for (; k + 8 <= end; k += 8) {
                    T p[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) p[u] = prod[k + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) s += p[u];
                }
                for (; k < end; k++) s += prod[k];
            }
            unsigned long long longrows = __ballot(len > kLongRow);
            while (longrows) {
                const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(longrows));
                longrows &= longrows - 1;
                const int ls = __builtin_amdgcn_readlane(start, src), le = __builtin_amdgcn_readlane(end, src);
                T part = 0;
                for (int k = ls + lane; k < le; k += kWave) part += prod[k];
                part = group_sum<kWave>(part);
                if (lane == src) s = part;
            }
            if (valid) {
                if (rr < nr) y[row_index ? row_index[a0 + rr] : a0 + rr] = s;
                else { row_carry[t] = a1; val_carry[t] = s; }
            }
        }
    };
    if (t_begin < t_end) {
        int ca[kLoads], cb[kLoads];
        T va[kLoads], vb[kLoads];
        issue_loads(t_begin, ca, va);
        for (int t = t_begin;; t += 2 * stride) {
            tile(t, ca, va, cb, vb);
            if (t + stride >= t_end) break;
            tile(t + stride, cb, vb, ca, va);
            if (t + 2 * stride >= t_end) break;
        }
    }
}

// -------------------------------------------------------------- vector ops --

template <typename T>
__global__ __launch_bounds__(kBlock) void sumsq_partial_kernel(int n, const T *__restrict__ x,
                                                               double *__restrict__ partial)
{
    __shared__ double wsum[kBlock / kWave];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double v = (double)x[i];
        s = __builtin_fma(v, v, s);
    }
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(kBlock) void sumsq_final_kernel(int nparts, const double *__restrict__ partial,
                                                             double *__restrict__ out)
{
    __shared__ double wsum[kBlock / kWave];
    double s = 0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) s += partial[i];
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sqrt((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

template <typename T>
__global__ __launch_bounds__(kBlock) void scale_kernel(int n, const T *__restrict__ y, T *__restrict__ x,
                                                       const double *__restrict__ norm)
{
    const double nrm = norm[0];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        x[i] = (T)((double)y[i] / nrm);
}

// the same with the norm handed over by value (the row-block layer sums the blocks' norms on the host)
template <typename T>
__global__ __launch_bounds__(kBlock) void scale_value_kernel(int n, const T *__restrict__ y, T *__restrict__ x, double nrm)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        x[i] = (T)((double)y[i] / nrm);
}

// --------------------------------------------------------------- launchers --

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// A grid of `blocks` workgroups as (x, y) with x <= 2^22, for kernels whose one-dimensional grid would
// pass the limit of 2^32 threads per dimension; the kernel flattens it and drops the surplus.
static dim3 folded_grid(unsigned blocks)
{
    const unsigned gx = std::min(std::max(blocks, 1u), 1u << 22);
    return dim3(gx, (blocks + gx - 1) / gx);
}

template <typename T> static int launch_csr_vector(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    const int *rp = m->d_row_ptr, *cj = m->d_col;
    const T *va = (const T *)m->d_val;
    int rows = m->rows;
    // rows walked side by side by one lane group: variant 1 -> 1, 2 -> 2, 3 -> 4, 4 -> 8 (never more than
    // the group has lanes); variant 0 -> 4 for groups of 16 lanes and more, else 1 (what measurements favour)
    static const int kRowsOfVariant[] = {0, 1, 2, 4, 8};
    // variants 5, 6: the LDS-staged kernel for short rows (2, 4 or 8 lanes per row; 6 = non-temporal loads);
    // 7, 8: the same, patterned row blocks reading one row's column indices
    if (m->variant >= 5 && m->variant <= 8) {
        const dim3 grid(cdiv(rows, kWave)), block(kWave);
        const int g = std::min(m->lanes_per_row, 8);
        const bool nt = m->variant == 6 || m->variant == 8;
        const unsigned char *pat = m->variant >= 7 ? m->d_block_pat : nullptr;
        if (m->variant >= 7 && !pat && m->stored > 0) { set_error("csr_vector: variant %d needs the block patterns", m->variant); return 1; }
#define SPMV_VECLDS(G)                                                                                \
    case G:                                                                                           \
        if (!nt)                                                                                      \
            hipLaunchKernelGGL((csr_vector_lds_kernel<T, 8, G, false>), grid, block, 0, st, rows, rp, cj, va, pat, \
                               m->d_row_map, x, y, m->block_map);                                                   \
        else                                                                                          \
            hipLaunchKernelGGL((csr_vector_lds_kernel<T, 8, G, true>), grid, block, 0, st, rows, rp, cj, va, pat, \
                               m->d_row_map, x, y, m->block_map);                                                   \
        return 0;
        switch (g) {
            SPMV_VECLDS(2)
            SPMV_VECLDS(4)
            SPMV_VECLDS(8)
        default: set_error("csr_vector: lanes_per_row %d is not one of 2,4,8,16,32,64", m->lanes_per_row); return 1;
        }
#undef SPMV_VECLDS
    }
    if (m->variant < 0 || m->variant > 4) { set_error("csr_vector: unknown kernel variant %d", m->variant); return 1; }
    const int side = m->variant == 0 ? (m->lanes_per_row >= 16 ? 4 : 1)
                                     : std::min(kRowsOfVariant[m->variant], m->lanes_per_row);
#define SPMV_VEC_ONE(G, R)                                                                            \
    case G * 16 + R:                                                                                  \
        hipLaunchKernelGGL((csr_vector_kernel<T, G, R>), folded_grid(cdiv(rows, kBlock / G * R)), dim3(kBlock), 0, \
                           st, rows, rp, cj, va, m->d_row_map, x, y, m->block_map);                   \
        break;
#define SPMV_VEC_CASE(G) SPMV_VEC_ONE(G, 1) SPMV_VEC_ONE(G, 2) SPMV_VEC_ONE(G, 4) SPMV_VEC_ONE(G, 8)
    switch (m->lanes_per_row * 16 + side) {
        SPMV_VEC_ONE(2, 1) SPMV_VEC_ONE(2, 2)
        SPMV_VEC_ONE(4, 1) SPMV_VEC_ONE(4, 2) SPMV_VEC_ONE(4, 4)
        SPMV_VEC_CASE(8)
        SPMV_VEC_CASE(16)
        SPMV_VEC_CASE(32)
        SPMV_VEC_CASE(64)
    default:
        set_error("csr_vector: lanes_per_row %d is not one of 2,4,8,16,32,64", m->lanes_per_row);
        return 1;
    }
#undef SPMV_VEC_CASE
#undef SPMV_VEC_ONE
    return 0;
}

// variant 0 = exact-order kernel; 1..4 = 16-byte-load kernel with 2, 4, 7, 14 steps
// per batch (fp64: 4, 8, 14, 28 columns; fp32: 8, 16, 28, 56); +8 = the same with
// non-temporal loads of the streamed arrays.
constexpr int kSlicedVariants[] = {0, 1, 2, 3, 4, 9, 10, 11, 12};

int variant_candidates(const spmv_hip_matrix *m, int *out, int max_out)
{
    int n = 0;
    const bool sliced64 = m->d_slice_ptr && m->slice_height == kWave && m->format != SPMV_HIP_SELL_DFC;
    if (m->format == SPMV_HIP_SELL_DFC) {
        for (int v = 0; v <= 8; v++)                          // see launch_sell_dfc
            if (n < max_out) out[n++] = v;
        if (m->d_sdiag && m->diag_columns * 2 >= m->vops)
            for (int v = 9; v <= 12; v++)
                if (n < max_out) out[n++] = v;
    } else if (sliced64) {
        for (int v : kSlicedVariants)
            if (n < max_out) out[n++] = v;
        if (m->dtype == SPMV_HIP_F64)                         // 16-byte loads in the exact summation order
            for (int v : {24, 25, 26, 27})
                if (n < max_out) out[n++] = v;
        // the kernels that skip the indices of diagonal columns, when at least half the columns are
        if (m->d_sdiag && m->diag_columns * 2 * kWave >= m->stored)
            for (int v : {16, 17, 18, 19, 20, 21, 22, 23})
                if (n < max_out) out[n++] = v;
        if (m->d_sdiag && m->diag_columns * 2 * kWave >= m->stored && m->dtype == SPMV_HIP_F64)
            for (int v : {28, 29, 30, 31})                    // ... with 16-byte value loads
                if (n < max_out) out[n++] = v;
    } else if (m->format == SPMV_HIP_CSR_VECTOR) {
        // vector CSR: candidates are (lanes per row) + 256 * (variant = rows side by side, see launch_csr_vector)
        for (int g = 2; g <= 64; g *= 2)
            for (int v = 1; v <= 3; v++)
                if ((1 << (v - 1)) <= g && n < max_out) out[n++] = g + 256 * v;
        for (int g = 2; g <= 8; g *= 2)                      // LDS-staged kernel for short rows
            for (int v = 5; v <= 6; v++)
                if (n < max_out) out[n++] = g + 256 * v;
        if (m->d_block_pat && !m->d_hot_col && (int64_t)m->pat_blocks * 2 * kWave >= m->rows)
            for (int g = 2; g <= 8; g *= 2)                  // ... reading one row's indices per patterned block
                for (int v = 7; v <= 8; v++)
                    if (n < max_out) out[n++] = g + 256 * v;
    } else if (m->format == SPMV_HIP_CSR_SCALAR) {
        // LDS chunk sizes of the staged lane-per-row kernel (all give the same bits)
        for (int v : {0, 2, 3, 4, 8, 10, 11, 12})
            if (n < max_out) out[n++] = v;
        // the kernels that read one row's indices per patterned block, when at least half the blocks are
        if (m->d_block_pat && !m->d_hot_col && (int64_t)m->pat_blocks * 2 * kWave >= m->rows)
            for (int v : {16, 17, 18, 20, 21, 22})
                if (n < max_out) out[n++] = v;
    } else if (m->format == SPMV_HIP_CSR_MERGE || m->format == SPMV_HIP_CSR_BAL) {
        for (int v : {0, 1})
            if (n < max_out) out[n++] = v;
        if (m->d_tile_pat && !m->d_hot_col && m->pat_tiles * 2 >= m->ntiles && n < max_out) out[n++] = 2;
        // persistent workgroups with each column group's hottest labels in LDS (8 or 16 waves), for the
        // stacked matrix of "hot_columns" = 2
        // (they use all 160 KiB of a CU's LDS: offered only where one workgroup may have that much)
        if (m->group_cpx > 0 && m->items_per_tile == 8 * kWave && m->max_lds >= 160 * 1024)
            for (int v : {3, 4})
                if (n < max_out) out[n++] = v;
    } else if (n < max_out) {
        out[n++] = 0;
    }
    return n;
}

// Whether `candidate` (as listed by variant_candidates) gives y the same bits as the handle's
// current setting -- what spmv_hip_analyze may choose from under the option "exact_order".
//   sliced formats of height 64: 0 and 16..31 keep the order of mult_csr_base; 1..12 reassociate.
//   padding-free SELL, lane-per-row CSR: every variant sums alike.
//   merge-path / balanced tiles: a tile that lies inside one row adds its entries per lane before the
//     butterfly, and which lane holds which entry depends on where the loads start -- at the tile's
//     first entry (variant 0) or on a 128-byte line (1, 2).
//   vector CSR: the bits depend on the lanes per row and on the kernel family (direct loads, one
//     fma per entry: variants 0..4; products staged in LDS, then added: 5..8).
bool candidate_keeps_bits(const spmv_hip_matrix *m, int candidate)
{
    const bool sliced64 = m->d_slice_ptr && m->slice_height == kWave && m->format != SPMV_HIP_SELL_DFC;
    if (sliced64) {
        const bool now_exact = m->variant == 0 || m->variant >= 16;
        const bool cand_exact = candidate == 0 || candidate >= 16;
        return now_exact ? cand_exact : candidate == m->variant;
    }
    if (m->format == SPMV_HIP_CSR_VECTOR) {
        const int g = candidate & 255, v = candidate >> 8;
        return g == m->lanes_per_row && (v >= 5) == (m->variant >= 5);
    }
    if (m->format == SPMV_HIP_CSR_MERGE || m->format == SPMV_HIP_CSR_BAL) return (candidate == 0) == (m->variant == 0);
    return true;
}

template <typename T, bool PERM>
static int launch_sliced64(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    dim3 grid(cdiv(m->nslices, kBlock / kWave)), block(kBlock);
    const T *sv = (const T *)m->d_sval;
#define SPMV_WIDE(U, NT)                                                                                   \
    hipLaunchKernelGGL((sliced64_wide_kernel<T, PERM, U, NT>), grid, block, 0, st, m->rows, m->nslices,    \
                       m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, y, m->block_map)
    switch (m->variant) {
    case 0:
        hipLaunchKernelGGL((sliced64_kernel<T, PERM, 8>), grid, block, 0, st, m->rows, m->nslices, m->d_slice_ptr,
                           m->d_row_order, m->d_scol, sv, x, y, m->block_map);
        break;
    case 1: SPMV_WIDE(2, false); break;
    case 2: SPMV_WIDE(4, false); break;
    case 3: SPMV_WIDE(7, false); break;
    case 4: SPMV_WIDE(14, false); break;
    case 9: SPMV_WIDE(2, true); break;
    case 10: SPMV_WIDE(4, true); break;
    case 11: SPMV_WIDE(7, true); break;
    case 12: SPMV_WIDE(14, true); break;
#define SPMV_WIDE_EXACT(V, U, NT)                                                                          \
    case V:                                                                                                \
        if constexpr (std::is_same<T, double>::value) {                                                    \
            hipLaunchKernelGGL((sliced64_wide_exact_kernel<PERM, U, NT>), grid, block, 0, st, m->rows, m->nslices, \
                               m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, y, m->block_map);         \
        } else {                                                                                           \
            set_error("sliced format: variant %d is an fp64 kernel", V);                                   \
            return 1;                                                                                      \
        }                                                                                                  \
        break;
        // 24..27 (fp64): 16-byte loads with the exact summation order (7, 14 steps per batch; 26, 27 non-temporal)
        SPMV_WIDE_EXACT(24, 7, false)
        SPMV_WIDE_EXACT(25, 14, false)
        SPMV_WIDE_EXACT(26, 7, true)
        SPMV_WIDE_EXACT(27, 14, true)
#undef SPMV_WIDE_EXACT
#define SPMV_DIAG_WIDE(V, U, NT)                                                                           \
    case V:                                                                                                \
        if (!m->d_sdiag && m->stored > 0) { set_error("sliced format: variant %d needs the diagonal descriptors", V); return 1; } \
        if constexpr (std::is_same<T, double>::value) {                                                    \
            hipLaunchKernelGGL((sliced64_diag_wide_kernel<PERM, U, NT>), grid, block, 0, st, m->rows, m->columns, m->nslices, \
                               m->d_slice_ptr, m->d_row_order, m->d_sdiag, m->d_scol, sv, x, y, m->block_map); \
        } else {                                                                                           \
            set_error("sliced format: variant %d is an fp64 kernel", V);                                   \
            return 1;                                                                                      \
        }                                                                                                  \
        break;
        // 28..31 (fp64): the same with the indices of diagonal slice columns not read (7, 4 steps; 30, 31 non-temporal)
        SPMV_DIAG_WIDE(28, 7, false)
        SPMV_DIAG_WIDE(29, 4, false)
        SPMV_DIAG_WIDE(30, 7, true)
        SPMV_DIAG_WIDE(31, 4, true)
#undef SPMV_DIAG_WIDE
#define SPMV_DIAG(V, U, NT)                                                                                \
    case V:                                                                                                \
        if (!m->d_sdiag && m->stored > 0) { set_error("sliced format: variant %d needs the diagonal descriptors", V); return 1; } \
        hipLaunchKernelGGL((sliced64_diag_kernel<T, PERM, U, NT>), grid, block, 0, st, m->rows, m->columns, \
                           m->nslices, m->d_slice_ptr, m->d_row_order, m->d_sdiag, m->d_scol, sv, x, y, m->block_map); \
        break;
        // 16..19: columns along diagonals are not read (8, 9, 27, 16 columns per batch); 20..23: non-temporal loads
        SPMV_DIAG(16, 8, false)
        SPMV_DIAG(17, 9, false)
        SPMV_DIAG(18, 27, false)
        SPMV_DIAG(19, 16, false)
        SPMV_DIAG(20, 8, true)
        SPMV_DIAG(21, 9, true)
        SPMV_DIAG(22, 27, true)
        SPMV_DIAG(23, 16, true)
#undef SPMV_DIAG
    default: set_error("sliced format: unknown kernel variant %d", m->variant); return 1;
    }
#undef SPMV_WIDE
    return 0;
}

template <typename T> static int launch_sell_dfc(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    const dim3 grid(cdiv(m->nslices, kBlock / kWave)), block(kBlock);
    const T *sv = (const T *)m->d_sval;
#define SPMV_DFC(V, U, NT)                                                                               \
    case V:                                                                                              \
        hipLaunchKernelGGL((sell_dfc_kernel<T, U, NT>), grid, block, 0, st, m->rows, m->nslices, m->d_slice_ptr, \
                           m->d_sorted_len, m->d_row_order, m->d_scol, sv, x, y, nullptr, nullptr, m->block_map); \
        break;
#define SPMV_DFC_LDS(V, I, NT)                                                                           \
    case V:                                                                                              \
        hipLaunchKernelGGL((sell_dfc_lds_kernel<T, I, NT>), dim3((unsigned)m->nslices), dim3(kWave), 0, st, m->rows, \
                           m->nslices, m->d_slice_ptr, m->d_sorted_len, m->d_row_order, m->d_scol, sv, x, y, m->block_map); \
        break;
#define SPMV_DFC_DIAG(V, U, NT)                                                                          \
    case V:                                                                                              \
        if (!m->d_sdiag) { set_error("sell_dfc: variant %d needs the diagonal descriptors", V); return 1; } \
        hipLaunchKernelGGL((sell_dfc_kernel<T, U, NT, true>), grid, block, 0, st, m->rows, m->nslices, m->d_slice_ptr, \
                           m->d_sorted_len, m->d_row_order, m->d_scol, sv, x, y, m->d_vop_ptr, m->d_sdiag, m->block_map); \
        break;
    // 0..4: partial columns as masked loads (8, 16, 8 nt, 16 nt, 4 columns per batch);
    // 5..8: partial columns staged through LDS (chunks of 512, 512 nt, 256, 1024 entries);
    // 9..12: as 0..3, the indices of columns along a diagonal not read
    switch (m->stored == 0 && m->variant >= 9 && m->variant <= 12 ? 0 : m->variant) {
        SPMV_DFC_DIAG(9, 8, false)
        SPMV_DFC_DIAG(10, 16, false)
        SPMV_DFC_DIAG(11, 8, true)
        SPMV_DFC_DIAG(12, 16, true)
        SPMV_DFC(0, 8, false)
        SPMV_DFC(1, 16, false)
        SPMV_DFC(2, 8, true)
        SPMV_DFC(3, 16, true)
        SPMV_DFC(4, 4, false)
        SPMV_DFC_LDS(5, 8, false)
        SPMV_DFC_LDS(6, 8, true)
        SPMV_DFC_LDS(7, 4, false)
        SPMV_DFC_LDS(8, 16, false)
    default: set_error("sell_dfc: unknown kernel variant %d", m->variant); return 1;
    }
#undef SPMV_DFC_LDS
#undef SPMV_DFC
#undef SPMV_DFC_DIAG
    return 0;
}

template <typename T> static int launch_sliced(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    const T *sv = (const T *)m->d_sval;
    const bool perm = m->d_row_order != nullptr;
    if (m->slice_height == kWave) {
        return perm ? launch_sliced64<T, true>(m, x, y, st) : launch_sliced64<T, false>(m, x, y, st);
    } else {
        dim3 grid(cdiv(m->rows, kBlock));
        if (perm)
            hipLaunchKernelGGL((sliced_any_kernel<T, true>), grid, dim3(kBlock), 0, st, m->rows, m->slice_height,
                               m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, y, m->block_map);
        else
            hipLaunchKernelGGL((sliced_any_kernel<T, false>), grid, dim3(kBlock), 0, st, m->rows, m->slice_height,
                               m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, y, m->block_map);
    }
    return 0;
}

// scalar CSR variants: 0, 2, 3, 4 = LDS-staged with 512, 1024, 256, 2048 entries per chunk, +8 = the same
// with non-temporal loads of the streamed arrays (for matrices far larger than the caches); 1 = plain
template <typename T> static int launch_csr_scalar(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    const T *va = (const T *)m->d_val;
    const dim3 grid(cdiv(m->rows, kWave)), block(kWave);
#define SPMV_ROWS_ONE(V, I, NT)                                                                          \
    case V:                                                                                              \
        hipLaunchKernelGGL((csr_rows_lds_kernel<T, I, NT>), grid, block, 0, st, m->rows, m->d_row_ptr, m->d_col, \
                           va, m->d_row_map, x, y, m->block_map);                                                      \
        break;
#define SPMV_ROWS_CASE(V, I) SPMV_ROWS_ONE(V, I, false) SPMV_ROWS_ONE(V + 8, I, true)
    // a matrix without entries has no patterns to read: any chunked kernel writes its zeros
    switch (m->variant >= 16 && m->variant <= 22 && !m->d_block_pat && m->stored == 0 ? 0 : m->variant) {
        SPMV_ROWS_CASE(0, 8)
        SPMV_ROWS_CASE(2, 16)
        SPMV_ROWS_CASE(3, 4)
        SPMV_ROWS_CASE(4, 32)
    case 1:
        hipLaunchKernelGGL((csr_scalar_kernel<T>), dim3(cdiv(m->rows, kBlock)), dim3(kBlock), 0, st, m->rows,
                           m->d_row_ptr, m->d_col, va, m->d_row_map, x, y, m->block_map);
        break;
#define SPMV_ROWS_PAT(V, I, NT)                                                                          \
    case V:                                                                                              \
        if (!m->d_block_pat) { set_error("scalar CSR: variant %d needs the block patterns", V); return 1; } \
        hipLaunchKernelGGL((csr_rows_lds_pat_kernel<T, I, NT>), grid, block, 0, st, m->rows, m->d_row_ptr, m->d_col, \
                           va, m->d_block_pat, m->d_row_map, x, y, m->block_map);                                      \
        break;
        // 16..18: patterned row blocks read one row's column indices (512, 1024, 256 entries per chunk); 20..22: non-temporal
        SPMV_ROWS_PAT(16, 8, false)
        SPMV_ROWS_PAT(17, 16, false)
        SPMV_ROWS_PAT(18, 4, false)
        SPMV_ROWS_PAT(20, 8, true)
        SPMV_ROWS_PAT(21, 16, true)
        SPMV_ROWS_PAT(22, 4, true)
#undef SPMV_ROWS_PAT
    default: set_error("scalar CSR: unknown kernel variant %d", m->variant); return 1;
    }
#undef SPMV_ROWS_CASE
#undef SPMV_ROWS_ONE
    return 0;
}

template <typename T> static int launch_tiles(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    const bool bal = m->format == SPMV_HIP_CSR_BAL;
    const int list_len = bal ? m->nz_rows : m->rows;
    const int *list = bal ? m->d_nzr_end : m->d_row_ptr + 1;
    // packed_rows: results go to the position in the non-empty row list instead of the row itself
    const int *index = bal && !m->packed_rows ? m->d_nzr_id : nullptr;
    const T *va = (const T *)m->d_val;
    T *vc = (T *)m->d_val_carry;
    if (bal && m->has_empty_rows && !m->leave_empty_rows)
        SPMV_HIP_TRY(hipMemsetAsync(y, 0, (size_t)m->rows * sizeof(T), st), return 1);
    if (m->ntiles == 0) {
        if (!bal && m->rows > 0) SPMV_HIP_TRY(hipMemsetAsync(y, 0, (size_t)m->rows * sizeof(T), st), return 1);
        return 0;
    }
    // variant 0: loads start at the tile's first entry; 1: line-aligned non-temporal loads (MODE 2); 2: as 1,
    // patterned tiles reading one row's column indices; 3, 4: as 1 with persistent workgroups of 8 / 16 waves
    // that keep the hottest labels of their column group in LDS
    if (m->variant == 3 || m->variant == 4) {
        if (m->group_cpx <= 0 || m->items_per_tile != 8 * kWave) {
            set_error("tile kernel: variant %d needs the column groups of hot_columns 2 and tiles of 512", m->variant);
            return 1;
        }
        if (m->max_lds < 160 * 1024) {
            set_error("tile kernel: variant %d needs 160 KiB of LDS per workgroup, the device allows %d bytes", m->variant, m->max_lds);
            return 1;
        }
        const dim3 grid((unsigned)std::max(m->cus / 8, 1) * 8);        // one workgroup per CU, a multiple of the 8 groups
        tile_groups tg;
        for (int g = 0; g <= 8; g++) tg.first_tile[g] = m->group_first_tile[g];
        if (m->variant == 3)
            hipLaunchKernelGGL((wave_tile_hot_kernel<T, 8, 8>), grid, dim3(kWave * 8), 0, st, list, index, m->d_tile_coord,
                               m->d_col, va, x, y, m->d_row_carry, vc, tg, m->group_cpx, (int)m->stored);
        else
            hipLaunchKernelGGL((wave_tile_hot_kernel<T, 8, 16>), grid, dim3(kWave * 16), 0, st, list, index, m->d_tile_coord,
                               m->d_col, va, x, y, m->d_row_carry, vc, tg, m->group_cpx, (int)m->stored);
        hipLaunchKernelGGL((tile_fixup_kernel<T>), dim3(cdiv(m->ntiles, kBlock)), dim3(kBlock), 0, st, m->ntiles,
                           list_len, index, m->d_row_carry, vc, y);
        return 0;
    }
    if (m->variant < 0 || m->variant > 2) { set_error("tile kernel: unknown kernel variant %d", m->variant); return 1; }
    if (m->variant == 2 && !m->d_tile_pat) { set_error("tile kernel: variant 2 needs the tile patterns"); return 1; }
    const unsigned char *pat = m->variant == 2 ? m->d_tile_pat : nullptr;
    // merge tiles hold items_per_tile path items; balanced tiles hold that many non-zeros
#define SPMV_WTILE_CASE(I)                                                                              \
    case I:                                                                                              \
        if (m->variant == 0)                                                                             \
            hipLaunchKernelGGL((wave_tile_kernel<T, I, 0>), dim3(m->ntiles), dim3(kWave), 0, st, list_len, list, \
                               index, m->d_tile_coord, pat, m->d_col, va, x, y, m->d_row_carry, vc, m->group_cpx > 0 ? 0 : m->block_map);     \
        else                                                                                             \
            hipLaunchKernelGGL((wave_tile_kernel<T, I, 2>), dim3(m->ntiles), dim3(kWave), 0, st, list_len, list, \
                               index, m->d_tile_coord, pat, m->d_col, va, x, y, m->d_row_carry, vc, m->group_cpx > 0 ? 0 : m->block_map);     \
        break;
    switch (m->items_per_tile % kWave ? 0 : m->items_per_tile / kWave) {
        SPMV_WTILE_CASE(4)
        SPMV_WTILE_CASE(8)
        SPMV_WTILE_CASE(16)
        SPMV_WTILE_CASE(32)
    default:
        set_error("tile kernel: items_per_tile %d must be 64 x {4,8,16,32}", m->items_per_tile);
        return 1;
    }
#undef SPMV_WTILE_CASE
    hipLaunchKernelGGL((tile_fixup_kernel<T>), dim3(cdiv(m->ntiles, kBlock)), dim3(kBlock), 0, st, m->ntiles,
                       list_len, index, m->d_row_carry, vc, y);
    return 0;
}

template <typename T> static int launch_mult_t(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    int rc = 0;
    if (m->rows == 0) return 0;
    switch (m->format) {
    case SPMV_HIP_CSR_SCALAR: rc = launch_csr_scalar<T>(m, x, y, st); break;
    case SPMV_HIP_CSR_VECTOR: rc = launch_csr_vector<T>(m, x, y, st); break;
    case SPMV_HIP_CSR_MERGE:
    case SPMV_HIP_CSR_BAL: rc = launch_tiles<T>(m, x, y, st); break;
    case SPMV_HIP_SELL:
    case SPMV_HIP_SELLP:
    case SPMV_HIP_ELL: rc = launch_sliced<T>(m, x, y, st); break;
    case SPMV_HIP_SELL_DFC: rc = launch_sell_dfc<T>(m, x, y, st); break;
    default: set_error("unknown format %d", m->format); return 1;
    }
    if (rc) return rc;
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

// xh[k] = x[old_of_new[k]]: x under the column labels of the "hot_columns" option
template <typename T>
__global__ __launch_bounds__(kBlock) void permute_x_kernel(int n, const int *__restrict__ old_of_new,
                                                           const T *__restrict__ x, T *__restrict__ xh)
{
    int k = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n) return;
    const int old = old_of_new[k];
    xh[k] = old >= 0 ? x[old] : (T)0;             // labels without a column (dealing over the XCDs leaves up to 7)
}

// the same for the labels dealt over column groups: group blockIdx.y, its first `used` labels
template <typename T>
__global__ __launch_bounds__(kBlock) void permute_x_groups_kernel(int used, int cpx, const int *__restrict__ old_of_new,
                                                                  const T *__restrict__ x, T *__restrict__ xh)
{
    const int k = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= used) return;
    const size_t label = (size_t)blockIdx.y * cpx + k;
    const int old = old_of_new[label];
    xh[label] = old >= 0 ? x[old] : (T)0;
}

// presence[r]: bit g set when group g's virtual row of r (g * rows + r) holds entries
__global__ __launch_bounds__(kBlock) void group_presence_kernel(int rows, const int *__restrict__ vrow_ptr,
                                                                unsigned char *__restrict__ presence)
{
    int r = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (r >= rows) return;
    unsigned bits = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        const size_t v = (size_t)g * rows + r;
        bits |= (vrow_ptr[v + 1] > vrow_ptr[v] ? 1u : 0u) << g;
    }
    presence[r] = (unsigned char)bits;
}

int launch_group_presence(int rows, const int *d_vrow_ptr, unsigned char *d_presence, hipStream_t stream)
{
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(group_presence_kernel, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, rows,
                       d_vrow_ptr, d_presence);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

// count[g * nblk + b] = rows of the 64-row block b with entries in column group g: after an exclusive scan,
// the position of the block's first such row among the non-empty virtual rows (which the stacked matrix
// lists group by group, rows ascending)
__global__ __launch_bounds__(kBlock) void group_block_count_kernel(int rows, int nblk, const unsigned char *__restrict__ presence,
                                                                   int *__restrict__ count)
{
    const int r = (int)(blockIdx.x * kBlock + threadIdx.x);
    const unsigned bits = r < rows ? presence[r] : 0u;
    const int blk = r >> 6;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        const unsigned long long mask = __ballot((bits >> g) & 1u);
        if ((threadIdx.x & 63) == 0 && blk < nblk) count[(size_t)g * nblk + blk] = __popcll(mask);
    }
}

int launch_group_block_count(int rows, const unsigned char *d_presence, int *d_count, hipStream_t stream)
{
    if (rows <= 0) return 0;
    const int nblk = (rows + kWave - 1) / kWave;
    hipLaunchKernelGGL(group_block_count_kernel, dim3((unsigned)((rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, rows,
                       nblk, d_presence, d_count);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

// y[r] = the partial results of row r in group order.  The tile kernel left them packed: one slot per
// non-empty virtual row, in the stacked matrix's order, so that they are written and read as dense
// streams.  Row r's slot for group g is base[g][r / 64] + the number of rows before r in its 64-row
// block that have entries in g (a ballot over the wave's presence bits).
// Each wave takes kSumBlocks consecutive 64-row blocks, so that as many independent chains (presence byte ->
// base -> partial results) are in flight per wave: with one block per wave the pass is bound by the three
// dependent round trips, not by its bytes.
constexpr int kSumBlocks = 4;

template <typename T>
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(int rows, int nblk, const unsigned char *__restrict__ presence,
                                                              const int *__restrict__ base, const T *__restrict__ part,
                                                              T *__restrict__ y)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int blk0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * kSumBlocks);
    const unsigned long long before = (1ull << lane) - 1ull;
    unsigned bits[kSumBlocks];
#pragma unroll
    for (int k = 0; k < kSumBlocks; k++) {
        const int r = (blk0 + k) * kWave + lane;
        bits[k] = r < rows ? presence[r] : 0u;
    }
    T p[kSumBlocks][8];
#pragma unroll
    for (int k = 0; k < kSumBlocks; k++) {
        const int blk = min(blk0 + k, nblk - 1);               // blocks past the end have no presence bits set
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const bool has = (bits[k] >> g) & 1u;
            const unsigned long long mask = __ballot(has);
            p[k][g] = 0;
            if (has) p[k][g] = __builtin_nontemporal_load(part + base[(size_t)g * nblk + blk] + __popcll(mask & before));
        }
    }
#pragma unroll
    for (int k = 0; k < kSumBlocks; k++) {
        T s = p[k][0];
#pragma unroll
        for (int g = 1; g < 8; g++) s += p[k][g];
        const int r = (blk0 + k) * kWave + lane;
        if (r < rows) y[r] = s;
    }
}

template <typename T>
static int launch_mult_xcd(const spmv_hip_matrix *m, const T *x, T *y, hipStream_t st)
{
    // only the labels the matrix references: the first xcd_used_cpx of each group
    if (m->xcd_used_cpx > 0)
        hipLaunchKernelGGL((permute_x_groups_kernel<T>), dim3(cdiv(m->xcd_used_cpx, kBlock), 8), dim3(kBlock), 0, st,
                           m->xcd_used_cpx, m->xcd_cpx, m->d_hot_old, x, (T *)m->d_hot_x);
    if (launch_mult(m->xcd_sub, m->d_hot_x, m->d_xcd_y, st)) return 1;
    hipLaunchKernelGGL((sum_partials_kernel<T>), dim3(cdiv(m->rows, kBlock * kSumBlocks)), dim3(kBlock), 0, st, m->rows,
                       (m->rows + kWave - 1) / kWave, m->d_xcd_presence, m->d_xcd_base, (const T *)m->d_xcd_y, y);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

int launch_mult(const spmv_hip_matrix *m, const void *d_x, void *d_y, hipStream_t stream)
{
    if (m->xcd_sub && m->rows > 0) {
        if (m->d_row_map) { set_error("hot_columns 2 does not combine with a row map"); return 1; }
        return m->dtype == SPMV_HIP_F32 ? launch_mult_xcd<float>(m, (const float *)d_x, (float *)d_y, stream)
                                        : launch_mult_xcd<double>(m, (const double *)d_x, (double *)d_y, stream);
    }
    if (m->d_hot_col && m->rows > 0) {
        // permute x, then run the format's kernel on the relabeled column array
        const dim3 grid(cdiv(m->columns, kBlock)), block(kBlock);
        if (m->dtype == SPMV_HIP_F32)
            hipLaunchKernelGGL((permute_x_kernel<float>), grid, block, 0, stream, m->columns, m->d_hot_old,
                               (const float *)d_x, (float *)m->d_hot_x);
        else
            hipLaunchKernelGGL((permute_x_kernel<double>), grid, block, 0, stream, m->columns, m->d_hot_old,
                               (const double *)d_x, (double *)m->d_hot_x);
        spmv_hip_matrix view = *m;
        view.d_col = m->d_hot_col;
        view.d_hot_col = nullptr;
        // the block patterns describe the caller's column numbers, not the relabeled ones
        view.d_block_pat = nullptr;
        if (view.format == SPMV_HIP_CSR_VECTOR && view.variant >= 7) view.variant -= 2;
        view.d_tile_pat = nullptr;
        if ((view.format == SPMV_HIP_CSR_MERGE || view.format == SPMV_HIP_CSR_BAL) && view.variant == 2) view.variant = 1;
        if (view.format == SPMV_HIP_CSR_SCALAR && view.variant >= 16) {
            const int plain[] = {0, 2, 3, 3, 8, 10, 11, 11};              // same chunk size (or the next), every index read
            view.variant = plain[(view.variant - 16) & 7];
        }
        return launch_mult(&view, m->d_hot_x, d_y, stream);
    }
    if (m->dtype == SPMV_HIP_F32) return launch_mult_t<float>(m, (const float *)d_x, (float *)d_y, stream);
    return launch_mult_t<double>(m, (const double *)d_x, (double *)d_y, stream);
}

// Scratch of the two-pass norm: kNormBlocks partial sums.  Launches that may run side by side must
// not share it, so each (device, stream) pair gets its own, allocated on that pair's first call and
// kept until the caller releases the stream (release_norm_scratch) or the process ends (8 KiB each).
static std::mutex norm_scratch_lock;
static std::map<std::pair<int, hipStream_t>, double *> norm_scratch_of;

static double *norm_scratch(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> hold(norm_scratch_lock);
    double *&p = norm_scratch_of[std::make_pair(dev, st)];
    if (!p && hipMalloc((void **)&p, kNormBlocks * sizeof(double)) != hipSuccess) p = nullptr;
    return p;
}

// Frees what the current device holds for `st`; the stream's work is waited for first, since a norm
// still in flight reads the array.  A stream that never ran a norm is not an error.
int release_norm_scratch(hipStream_t st)
{
    int dev = 0;
    SPMV_HIP_TRY(hipGetDevice(&dev), return 1);
    double *p = nullptr;
    {
        std::lock_guard<std::mutex> hold(norm_scratch_lock);
        auto it = norm_scratch_of.find(std::make_pair(dev, st));
        if (it == norm_scratch_of.end()) return 0;
        p = it->second;
        norm_scratch_of.erase(it);
    }
    if (!p) return 0;
    SPMV_HIP_TRY(hipStreamSynchronize(st), return 1);
    SPMV_HIP_TRY(hipFree(p), return 1);
    return 0;
}

int launch_norm2_with(double *partial, int n, const void *d_x, int dtype, double *d_out, hipStream_t st)
{
    // every launch writes partial[0..blocks) before the second kernel reads exactly that range, so
    // whatever an earlier, larger call left behind is never read
    int blocks = (int)std::min<int64_t>(kNormBlocks, cdiv(std::max(n, 1), kBlock));
    if (dtype == SPMV_HIP_F32)
        hipLaunchKernelGGL((sumsq_partial_kernel<float>), dim3(blocks), dim3(kBlock), 0, st, n, (const float *)d_x, partial);
    else
        hipLaunchKernelGGL((sumsq_partial_kernel<double>), dim3(blocks), dim3(kBlock), 0, st, n, (const double *)d_x, partial);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(kBlock), 0, st, blocks, partial, d_out);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

int launch_norm2(int n, const void *d_x, int dtype, double *d_out, hipStream_t st)
{
    double *partial = norm_scratch(st);
    if (!partial) { set_error("norm2: could not allocate scratch"); return 1; }
    return launch_norm2_with(partial, n, d_x, dtype, d_out, st);
}

// ---- norm without allocation ------------------------------------------------------------------
// The same two kernels with their partial sums in an array that is part of the code object, so the
// entry point never allocates.  `scratch` is a host-side stand-in for the array, not a device
// address: each GPU holds its own copy at its own address, looked up once per device (with that
// device current) and kept in scratch_of_device; the kernels take that pointer as an argument.
} // namespace spmv

static __device__ __align__(256) double scratch[spmv::kNormBlocks];

namespace spmv {

static double *scratch_of_device[64] = {nullptr};
static std::mutex scratch_lock;

// Looks up the current device's copy of `scratch` (first call on that device; later calls return at once).
int static_scratch_setup()
{
    int dev = 0;
    SPMV_HIP_TRY(hipGetDevice(&dev), return 1);
    if (dev < 0 || dev >= 64) { set_error("norm2 without allocation: device %d is beyond the 64 the library tracks", dev); return 1; }
    std::lock_guard<std::mutex> hold(scratch_lock);
    if (scratch_of_device[dev]) return 0;
    double *ptr = nullptr;
    SPMV_HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)), return 1);
    if (!ptr) { set_error("norm2 without allocation: the scratch array has no address on device %d", dev); return 1; }
    scratch_of_device[dev] = ptr;
    return 0;
}

int launch_norm2_static(int64_t n, const void *d_x, int dtype, double *d_out, hipStream_t st)
{
    if (n < 0 || n > kNormStaticMax) {
        set_error("norm2 without allocation: n = %lld is outside 0..%lld elements", (long long)n, (long long)kNormStaticMax);
        return 1;
    }
    if (dtype != SPMV_HIP_F64 && dtype != SPMV_HIP_F32) { set_error("unknown dtype %d", dtype); return 1; }
    if (!d_out || (n > 0 && !d_x)) { set_error("norm2 without allocation: NULL vector"); return 1; }
    if (static_scratch_setup()) return 1;
    int dev = 0;
    SPMV_HIP_TRY(hipGetDevice(&dev), return 1);
    return launch_norm2_with(scratch_of_device[dev], (int)n, d_x, dtype, d_out, st);
}

int launch_scale(int n, const void *d_y, void *d_x, int dtype, const double *d_norm, hipStream_t st)
{
    if (n <= 0) return 0;
    int blocks = (int)std::min<int64_t>(2048, cdiv(n, kBlock));
    if (dtype == SPMV_HIP_F32)
        hipLaunchKernelGGL((scale_kernel<float>), dim3(blocks), dim3(kBlock), 0, st, n, (const float *)d_y, (float *)d_x, d_norm);
    else
        hipLaunchKernelGGL((scale_kernel<double>), dim3(blocks), dim3(kBlock), 0, st, n, (const double *)d_y, (double *)d_x, d_norm);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

int launch_scale_value(int n, const void *d_y, void *d_x, int dtype, double norm, hipStream_t st)
{
    if (n <= 0) return 0;
    int blocks = (int)std::min<int64_t>(2048, cdiv(n, kBlock));
    if (dtype == SPMV_HIP_F32)
        hipLaunchKernelGGL((scale_value_kernel<float>), dim3(blocks), dim3(kBlock), 0, st, n, (const float *)d_y, (float *)d_x, norm);
    else
        hipLaunchKernelGGL((scale_value_kernel<double>), dim3(blocks), dim3(kBlock), 0, st, n, (const double *)d_y, (double *)d_x, norm);
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

} // namespace spmv
