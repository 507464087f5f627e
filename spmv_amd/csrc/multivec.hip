// Products with a block of vectors: y_v = A * x_v for v = 0..n-1, the operation the reference
// declares as mult_mv_csr(struct csr *, int n, double *x, double *y) (src/spmv.h:54) and never
// defines.  Vector v is the contiguous array x + v*ldx, its result y + v*ldy.
//
// The matrix is streamed once for NV vectors at a time (NV = 8, 4, 2, then 1 for an odd last
// vector), so its 12 bytes per entry are shared by NV products.  Every
// kernel here is lane-per-row with one fma per entry in ascending position, i.e. every column of
// the result is bit-identical to mult_csr_base (reference csr.c:4-13) of that vector, whatever
// the format of the handle:
//   sliced formats of height 64 (SELL-C-sigma, SELLP, ELL)  -> sliced64_mv_kernel, and with described slice
//                                                              columns and 2, 4, 8 vectors sliced64_mv_pair_kernel
//   sliced formats of other heights                         -> sliced_any_mv_kernel
//   SELL without padding                                    -> sell_dfc_mv_kernel
//   CSR, merge-path CSR, balanced CSR (all hold plain CSR)  -> csr_rows_lds_mv_kernel

#include "internal.h"
#include "device_utils.h"

#include <climits>
#include <type_traits>

namespace spmv {

template <typename T, bool NT> __device__ __forceinline__ T stream_load(const T *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

// One wave per slice, lane = row; U columns of the slice are loaded per batch and used for all NV
// vectors before the next batch.  Where the build pass found a slice column to run along a diagonal
// (sdiag: col - row of all its entries, INT_MIN otherwise; may be null) its indices are not read.
template <typename T, bool PERM, int NV, int U, bool NT>
__global__ __launch_bounds__(kBlock) void sliced64_mv_kernel(int rows, int cols, int nslices,
                                                             const int64_t *__restrict__ slice_ptr,
                                                             const int *__restrict__ row_order,
                                                             const int *__restrict__ sdiag,
                                                             const int *__restrict__ scol,
                                                             const T *__restrict__ sval,
                                                             const T *__restrict__ x, int64_t ldx,
                                                             T *__restrict__ y, int64_t ldy, int map)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const T *v = sval + base + lane;
    const int *c = scol + base + lane;
    const int r = s * kWave + lane;
    const int row = PERM ? row_order[min(r, rows - 1)] : min(r, rows - 1);
    const int *dg = sdiag ? sdiag + (base >> 6) : nullptr;
    const int last = max(cols - 1, 0);
    // the index of entry j of this lane's row: computed for a diagonal column, loaded otherwise.  In a
    // diagonal column a zero value marks a padding lane (the build pass describes no column that
    // holds a real zero); padding is stored as (column 0, value 0) (sellcs_format.c:167,171), so that
    // lane's operand is x_v[0] -- what the reference multiplies by zero
    auto column = [&](int jj, bool &diagonal) {
        const int d = dg ? __builtin_amdgcn_readfirstlane(dg[jj]) : INT_MIN;
        diagonal = d != INT_MIN;
        return diagonal ? min(max(row + d, 0), last) : stream_load<int, NT>(c + (int64_t)jj * kWave);
    };
    T acc[NV], x0[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        acc[k] = 0;
        x0[k] = dg && width > 0 ? x[k * ldx] : (T)0;
    }
    int j = 0;
    for (; j + U <= width; j += U) {
        int cc[U];
        bool dia[U];
        T vv[U], xx[NV][U];
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = stream_load<T, NT>(v + (int64_t)(j + u) * kWave);
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = column(j + u, dia[u]);
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int u = 0; u < U; u++) xx[k][u] = x[k * ldx + cc[u]];
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int u = 0; u < U; u++) acc[k] = fma_t(dia[u] && vv[u] == (T)0 ? x0[k] : xx[k][u], vv[u], acc[k]);
    }
    for (; j < width; j++) {
        bool dia;
        const int cc = column(j, dia);
        const T vv = stream_load<T, NT>(v + (int64_t)j * kWave);
#pragma unroll
        for (int k = 0; k < NV; k++) acc[k] = fma_t(dia && vv == (T)0 ? x0[k] : x[k * ldx + cc], vv, acc[k]);
    }
    if (r < rows) {
#pragma unroll
        for (int k = 0; k < NV; k++) y[k * ldy + row] = acc[k];
    }
}

// Lane l takes `c` of lane l - 1 (UP) or of lane l + 1 (!UP) through a DPP wave shift; the lane that has no
// such neighbour (0 / 63) keeps `edge`.  Every lane of the wave must be active.
template <bool UP, typename T> __device__ __forceinline__ T neighbour_lane(T c, T edge)
{
    constexpr int ctrl = UP ? 0x138 : 0x130;                 // wave_shr:1 / wave_shl:1
    if constexpr (sizeof(T) == 8) {
        typedef int I2 __attribute__((ext_vector_type(2)));
        const I2 cv = __builtin_bit_cast(I2, c), ev = __builtin_bit_cast(I2, edge);
        I2 r;
        r[0] = __builtin_amdgcn_update_dpp(ev[0], cv[0], ctrl, 0xf, 0xf, false);
        r[1] = __builtin_amdgcn_update_dpp(ev[1], cv[1], ctrl, 0xf, 0xf, false);
        return __builtin_bit_cast(T, r);
    } else {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, c),
                                                                 ctrl, 0xf, 0xf, false));
    }
}

// The same product with two rows per lane, for matrices with described slice columns (sdiag not null) and an even
// number of vectors.  With several vectors the lane-per-row kernels wait on many small gathers (8 bytes per lane,
// three columns in flight).  Here lane (h, q) owns the rows at positions 2q, 2q + 1 of the slice for the vectors of
// half h (NV / 2 of them), so that values, indices and results move as 16-byte pairs.  Where the slice's 64 rows are
// neighbours in ascending order and three columns run along the distances d-1, d, d+1 (the x-neighbours of a grid
// stencil), the 6 operands per lane and vector are 66 consecutive elements of x per wave: every lane loads its rows'
// middle operands x[row + d] as one pair, the outer ones are the neighbour lanes' middle operands (DPP wave shifts),
// and the two elements beside the window come for all vectors of the half in one load.  Two such runs are loaded
// together before either is added.  Each row still adds its entries in ascending position with one fma each, on the
// same operands: the bits are those of the lane-per-row kernel.
template <typename T, bool PERM, int NV, bool NT>
__global__ __launch_bounds__(kBlock) void sliced64_mv_pair_kernel(int rows, int cols, int nslices,
                                                                  const int64_t *__restrict__ slice_ptr,
                                                                  const int *__restrict__ row_order,
                                                                  const int *__restrict__ sdiag,
                                                                  const int *__restrict__ scol,
                                                                  const T *__restrict__ sval,
                                                                  const T *__restrict__ x, int64_t ldx,
                                                                  T *__restrict__ y, int64_t ldy, int map)
{
    static_assert(NV % 2 == 0, "the two half-waves share the vectors");
    constexpr int HV = NV / 2, kHalf = kWave / 2;
    typedef T TV2 __attribute__((ext_vector_type(2)));
    typedef T TVU __attribute__((ext_vector_type(2), aligned(sizeof(T))));
    typedef int IV2 __attribute__((ext_vector_type(2)));
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int h = lane >> 5, q = lane & (kHalf - 1);
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) >> 6);
    const TV2 *v = (const TV2 *)(sval + base) + q;           // column jj of the slice: v[jj * 32]
    const IV2 *c = (const IV2 *)(scol + base) + q;
    const int r = s * kWave + 2 * q;
    const int ra = PERM ? row_order[min(r, rows - 1)] : min(r, rows - 1);
    const int rb = PERM ? row_order[min(r + 1, rows - 1)] : min(r + 1, rows - 1);
    const bool pairs = __ballot(r + 1 >= rows || rb != ra + 1) == 0;      // every lane's rows are neighbours
    // the slice's 64 rows in ascending order without a gap (wave-uniform)
    const int row0 = __builtin_amdgcn_readfirstlane(ra);
    const bool in_order = __ballot(r + 1 >= rows || ra != row0 + 2 * q || rb != ra + 1) == 0;
    const int *dg = sdiag + (base >> 6);
    const int last = max(cols - 1, 0);
    const T *xh = x + (int64_t)h * HV * ldx;
    T *yh = y + (int64_t)h * HV * ldy;
    T acca[HV], accb[HV];
#pragma unroll
    for (int i = 0; i < HV; i++) acca[i] = accb[i] = 0;
    // In a described column a zero value marks a padding lane (the build pass describes no column that holds a
    // real zero); padding is stored as (column 0, value 0) (sellcs_format.c:167,171), so its operand is x_v[0] --
    // what the reference multiplies by zero.
    // one column for any two rows
    auto one_column = [&](int jj, int d) {
        const TV2 vv = stream_load<TV2, NT>(v + (int64_t)jj * kHalf);
        const bool dia = d != INT_MIN;
        int ca, cb;
        if (dia) {
            ca = vv[0] == (T)0 ? 0 : min(max(ra + d, 0), last);
            cb = vv[1] == (T)0 ? 0 : min(max(rb + d, 0), last);
        } else {
            const IV2 cc = stream_load<IV2, NT>(c + (int64_t)jj * kHalf);
            ca = cc[0];
            cb = cc[1];
        }
        T xa[HV], xb[HV];
#pragma unroll
        for (int i = 0; i < HV; i++) xa[i] = xh[i * ldx + ca];
#pragma unroll
        for (int i = 0; i < HV; i++) xb[i] = xh[i * ldx + cb];
#pragma unroll
        for (int i = 0; i < HV; i++) {
            acca[i] = fma_t(xa[i], vv[0], acca[i]);
            accb[i] = fma_t(xb[i], vv[1], accb[i]);
        }
    };
    // a run: three columns at the distances d - 1, d, d + 1 whose operands x[row0 + d - 1 .. row0 + 64 + d] exist
    struct run_data {
        TV2 vv[3];
        TVU mid[HV];     // x[ra + d], x[rb + d]
        T beside;        // lane 2i: x[row0 + d - 1], lane 2i + 1: x[row0 + 64 + d], of the half's vector i
    };
    auto is_run = [&](int d0, int d1, int d2) {
        return in_order && d0 != INT_MIN && d1 == d0 + 1 && d2 == d0 + 2 && row0 + d0 >= 0 && row0 + kWave - 1 + d2 <= last;
    };
    auto load_run = [&](int jj, int d, run_data &R) {
#pragma unroll
        for (int u = 0; u < 3; u++) R.vv[u] = stream_load<TV2, NT>(v + (int64_t)(jj + u) * kHalf);
#pragma unroll
        for (int i = 0; i < HV; i++) R.mid[i] = *(const TVU *)(xh + i * ldx + ra + d);
        R.beside = 0;
        if (q < 2 * HV) R.beside = xh[(q >> 1) * ldx + row0 + ((q & 1) ? kWave + d : d - 1)];
    };
    auto add_run = [&](const run_data &R) {
        const bool padded = __ballot(R.vv[0][0] == (T)0 || R.vv[0][1] == (T)0 || R.vv[1][0] == (T)0 ||
                                     R.vv[1][1] == (T)0 || R.vv[2][0] == (T)0 || R.vv[2][1] == (T)0) != 0;
        auto sums = [&](auto check) {
            constexpr bool CHECK = decltype(check)::value;
#pragma unroll
            for (int i = 0; i < HV; i++) {
                T before = neighbour_lane<true>(R.mid[i][1], R.mid[i][1]);      // x[ra + d - 1]
                T after = neighbour_lane<false>(R.mid[i][0], R.mid[i][0]);      // x[rb + d + 1]
                const T outer = __shfl(R.beside, (lane & kHalf) + 2 * i + (q == kHalf - 1 ? 1 : 0));
                if (q == 0) before = outer;
                if (q == kHalf - 1) after = outer;
                const T x0 = CHECK ? xh[i * ldx] : (T)0;
                acca[i] = fma_t(CHECK && R.vv[0][0] == (T)0 ? x0 : before, R.vv[0][0], acca[i]);
                accb[i] = fma_t(CHECK && R.vv[0][1] == (T)0 ? x0 : R.mid[i][0], R.vv[0][1], accb[i]);
                acca[i] = fma_t(CHECK && R.vv[1][0] == (T)0 ? x0 : R.mid[i][0], R.vv[1][0], acca[i]);
                accb[i] = fma_t(CHECK && R.vv[1][1] == (T)0 ? x0 : R.mid[i][1], R.vv[1][1], accb[i]);
                acca[i] = fma_t(CHECK && R.vv[2][0] == (T)0 ? x0 : R.mid[i][1], R.vv[2][0], acca[i]);
                accb[i] = fma_t(CHECK && R.vv[2][1] == (T)0 ? x0 : after, R.vv[2][1], accb[i]);
            }
        };
        if (padded) sums(std::true_type{});
        else sums(std::false_type{});
    };
    auto three_columns = [&](int jj, int d0, int d1, int d2) {
        if (is_run(d0, d1, d2)) {
            run_data R;
            load_run(jj, d1, R);
            add_run(R);
        } else {
            one_column(jj, d0);
            one_column(jj + 1, d1);
            one_column(jj + 2, d2);
        }
    };
    int j = 0;
    for (; j + 6 <= width; j += 6) {
        int d[6];
#pragma unroll
        for (int u = 0; u < 6; u++) d[u] = __builtin_amdgcn_readfirstlane(dg[j + u]);
        if (is_run(d[0], d[1], d[2]) && is_run(d[3], d[4], d[5])) {       // both in flight before either is added
            run_data R0, R1;
            load_run(j, d[1], R0);
            load_run(j + 3, d[4], R1);
            add_run(R0);
            add_run(R1);
        } else {
            three_columns(j, d[0], d[1], d[2]);
            three_columns(j + 3, d[3], d[4], d[5]);
        }
    }
    for (; j + 3 <= width; j += 3)
        three_columns(j, __builtin_amdgcn_readfirstlane(dg[j]), __builtin_amdgcn_readfirstlane(dg[j + 1]),
                      __builtin_amdgcn_readfirstlane(dg[j + 2]));
    for (; j < width; j++) one_column(j, __builtin_amdgcn_readfirstlane(dg[j]));
    if (pairs) {                                  // neighbours in y as well: one 16-byte store per vector
#pragma unroll
        for (int i = 0; i < HV; i++) {
            TVU two;
            two[0] = acca[i];
            two[1] = accb[i];
            *(TVU *)(yh + i * ldy + ra) = two;
        }
    } else {
        if (r < rows) {
#pragma unroll
            for (int i = 0; i < HV; i++) yh[i * ldy + ra] = acca[i];
        }
        if (r + 1 < rows) {
#pragma unroll
            for (int i = 0; i < HV; i++) yh[i * ldy + rb] = accb[i];
        }
    }
}

// Slice heights below 64 (the reference's CPU-shaped blocks): lane per row.
template <typename T, bool PERM, int NV>
__global__ __launch_bounds__(kBlock) void sliced_any_mv_kernel(int rows, int height,
                                                               const int64_t *__restrict__ slice_ptr,
                                                               const int *__restrict__ row_order,
                                                               const int *__restrict__ scol,
                                                               const T *__restrict__ sval,
                                                               const T *__restrict__ x, int64_t ldx,
                                                               T *__restrict__ y, int64_t ldy, int map)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int r = (int)(b * kBlock + threadIdx.x);
    if (r >= rows) return;
    const int s = r / height, k0 = r - s * height;
    const int64_t base = slice_ptr[s];
    const int width = (int)((slice_ptr[s + 1] - base) / height);
    T acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0;
    for (int j = 0; j < width; j++) {
        const int64_t at = base + (int64_t)j * height + k0;
        const int cc = scol[at];
        const T vv = sval[at];
#pragma unroll
        for (int k = 0; k < NV; k++) acc[k] = fma_t(x[k * ldx + cc], vv, acc[k]);
    }
    const int dst = PERM ? row_order[r] : r;
#pragma unroll
    for (int k = 0; k < NV; k++) y[k * ldy + dst] = acc[k];
}

// SELL-C-sigma without padding: column j of a slice holds the rows longer than j, which are lanes
// 0..active-1 because the slice's rows are sorted by length (reference sellcs_format.c:53-136).
// The start of each column is carried as a wave-uniform counter.
template <typename T, int NV, int U, bool NT>
__global__ __launch_bounds__(kBlock) void sell_dfc_mv_kernel(int rows, int nslices,
                                                             const int64_t *__restrict__ slice_ptr,
                                                             const int *__restrict__ sorted_len,
                                                             const int *__restrict__ row_order,
                                                             const int *__restrict__ scol,
                                                             const T *__restrict__ sval,
                                                             const T *__restrict__ x, int64_t ldx,
                                                             T *__restrict__ y, int64_t ldy, int map)
{
    unsigned b = xcd_block(blockIdx.x, gridDim.x, map);
    int s = __builtin_amdgcn_readfirstlane((int)(b * (kBlock / kWave) + threadIdx.x / kWave));
    if (s >= nslices) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int r = s * kWave + lane;
    const int len = r < rows ? sorted_len[r] : 0;
    const int width = __builtin_amdgcn_readfirstlane(len);
    const int *c = scol + slice_ptr[s] + lane;
    const T *v = sval + slice_ptr[s] + lane;
    int run = 0;
    T acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0;
    for (int j0 = 0; j0 < width; j0 += U) {
        int off[U], cc[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            off[u] = run;
            run += __popcll(__ballot(j0 + u < len));
        }
#pragma unroll
        for (int u = 0; u < U; u++) cc[u] = j0 + u < len ? stream_load<int, NT>(c + off[u]) : 0;
#pragma unroll
        for (int u = 0; u < U; u++) vv[u] = j0 + u < len ? stream_load<T, NT>(v + off[u]) : (T)0;
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const T xv = x[k * ldx + cc[u]];
                if (j0 + u < len) acc[k] = fma_t(xv, vv[u], acc[k]);
            }
    }
    if (r < rows) {
        const int dst = row_order[r];
#pragma unroll
        for (int k = 0; k < NV; k++) y[k * ldy + dst] = acc[k];
    }
}

// Plain CSR, lane per row, the matrix stream staged through LDS: a 64-lane workgroup owns 64
// consecutive rows = one contiguous run of the column and value arrays, walks it in line-aligned
// chunks of 64*IPL entries with full-width loads, gathers the NV x values of every entry there and
// parks (value, x_0..x_NV-1) in LDS, from where each lane adds the part of its own row that lies in
// the chunk.
template <typename T, int NV> struct mv_entry { T v; T x[NV]; };

template <typename T, int IPL, int NV, bool NT>
__global__ __launch_bounds__(kWave) void csr_rows_lds_mv_kernel(int rows, const int *__restrict__ row_ptr,
                                                                const int *__restrict__ col,
                                                                const T *__restrict__ val,
                                                                const unsigned char *__restrict__ pattern_len,
                                                                const int *__restrict__ row_map,
                                                                const T *__restrict__ x, int64_t ldx,
                                                                T *__restrict__ y, int64_t ldy, int map)
{
    constexpr int kCap = kWave * IPL;
    __shared__ mv_entry<T, NV> stage[kCap];
    const int blk = (int)xcd_block(blockIdx.x, gridDim.x, map);
    const int lane = threadIdx.x;
    const int row = blk * kWave + lane;
    const int last = min(blk * kWave + kWave, rows);
    const int rs = row_ptr[min(row, rows)];
    const int re = row < rows ? row_ptr[row + 1] : rs;
    const int run_begin = __builtin_amdgcn_readfirstlane(rs);
    const int run_end = row_ptr[last];
    // a block whose 64 rows repeat one pattern of L distances from the diagonal (pattern_len, may be
    // null) reads only its first row's column indices: entry e is row e / L, column row + dist[e % L]
    __shared__ int dist[32];
    const int L = pattern_len ? pattern_len[blk] : 0;        // wave-uniform
    const float inv_len = L ? 1.0f / (float)L : 0.0f;
    if (L) {
        if (lane < L) dist[lane] = col[run_begin + lane] - blk * kWave;
        __syncthreads();
    }
    T acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0;
    for (int e0 = run_begin & ~31; e0 < run_end; e0 += kCap) {
        int cc[IPL];
        T vv[IPL];
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            const int at = min(max(e0 + lane + i * kWave, run_begin), run_end - 1);
            if (L) {
                const int rel = at - run_begin;
                const int r = (int)(((float)rel + 0.5f) * inv_len);      // rel / L, exact for rel < 2048
                cc[i] = blk * kWave + r + dist[rel - r * L];
            } else {
                cc[i] = stream_load<int, NT>(col + at);
            }
        }
#pragma unroll
        for (int i = 0; i < IPL; i++)
            vv[i] = stream_load<T, NT>(val + min(max(e0 + lane + i * kWave, run_begin), run_end - 1));
#pragma unroll
        for (int i = 0; i < IPL; i++) {
            mv_entry<T, NV> e;
            e.v = vv[i];
#pragma unroll
            for (int k = 0; k < NV; k++) e.x[k] = x[k * ldx + cc[i]];
            stage[lane + i * kWave] = e;
        }
        __syncthreads();
        const int kend = min(re, min(e0 + kCap, run_end)) - e0;
        for (int p = max(rs, e0) - e0; p < kend; p++) {
            const mv_entry<T, NV> e = stage[p];
#pragma unroll
            for (int k = 0; k < NV; k++) acc[k] = fma_t(e.x[k], e.v, acc[k]);
        }
        __syncthreads();
    }
    if (row < rows) {
        const int dst = row_map ? row_map[row] : row;
#pragma unroll
        for (int k = 0; k < NV; k++) y[k * ldy + dst] = acc[k];
    }
}

// ------------------------------------------------------------------ launchers --

static inline unsigned blocks_for(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

template <typename T, int NV, bool NT>
static int launch_block(const spmv_hip_matrix *m, const T *x, int64_t ldx, T *y, int64_t ldy, hipStream_t st)
{
    constexpr int U = NV >= 8 ? 2 : (NV == 4 ? 4 : 8);
    switch (m->format) {
    case SPMV_HIP_CSR_SCALAR:
    case SPMV_HIP_CSR_VECTOR:
    case SPMV_HIP_CSR_MERGE:
    case SPMV_HIP_CSR_BAL:
        hipLaunchKernelGGL((csr_rows_lds_mv_kernel<T, 4, NV, NT>), dim3(blocks_for(m->rows, kWave)), dim3(kWave), 0, st,
                           m->rows, m->d_row_ptr, m->d_col, (const T *)m->d_val,
                           m->format == SPMV_HIP_CSR_SCALAR ? m->d_block_pat : nullptr, m->d_row_map, x, ldx, y, ldy, m->block_map);
        break;
    case SPMV_HIP_SELL:
    case SPMV_HIP_SELLP:
    case SPMV_HIP_ELL: {
        const bool perm = m->d_row_order != nullptr;
        const T *sv = (const T *)m->d_sval;
        if (m->slice_height == kWave) {
            const dim3 grid(blocks_for(m->nslices, kBlock / kWave)), block(kBlock);
            // the diagonal descriptors pay when at least half the slice columns have one
            const int *diag = m->d_sdiag && m->diag_columns * 2 * kWave >= m->stored ? m->d_sdiag : nullptr;
            if constexpr (NV % 2 == 0) {
                if (diag) {                       // two rows per lane, runs of neighbouring distances over the lanes
                    if (perm)
                        hipLaunchKernelGGL((sliced64_mv_pair_kernel<T, true, NV, NT>), grid, block, 0, st, m->rows,
                                           m->columns, m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                           ldx, y, ldy, m->block_map);
                    else
                        hipLaunchKernelGGL((sliced64_mv_pair_kernel<T, false, NV, NT>), grid, block, 0, st, m->rows,
                                           m->columns, m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                           ldx, y, ldy, m->block_map);
                    break;
                }
            }
            if (perm)
                hipLaunchKernelGGL((sliced64_mv_kernel<T, true, NV, U, NT>), grid, block, 0, st, m->rows, m->columns,
                                   m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                   ldx, y, ldy, m->block_map);
            else
                hipLaunchKernelGGL((sliced64_mv_kernel<T, false, NV, U, NT>), grid, block, 0, st, m->rows, m->columns,
                                   m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                   ldx, y, ldy, m->block_map);
        } else {
            const dim3 grid(blocks_for(m->rows, kBlock)), block(kBlock);
            if (perm)
                hipLaunchKernelGGL((sliced_any_mv_kernel<T, true, NV>), grid, block, 0, st, m->rows, m->slice_height,
                                   m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, ldx, y, ldy, m->block_map);
            else
                hipLaunchKernelGGL((sliced_any_mv_kernel<T, false, NV>), grid, block, 0, st, m->rows, m->slice_height,
                                   m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, ldx, y, ldy, m->block_map);
        }
        break;
    }
    case SPMV_HIP_SELL_DFC:
        hipLaunchKernelGGL((sell_dfc_mv_kernel<T, NV, (NV >= 8 ? 2 : 4), NT>), dim3(blocks_for(m->nslices, kBlock / kWave)),
                           dim3(kBlock), 0, st, m->rows, m->nslices, m->d_slice_ptr, m->d_sorted_len, m->d_row_order,
                           m->d_scol, (const T *)m->d_sval, x, ldx, y, ldy, m->block_map);
        break;
    default: set_error("unknown format %d", m->format); return 1;
    }
    return 0;
}

template <typename T>
static int launch_mult_mv_t(const spmv_hip_matrix *m, int n, const T *x, int64_t ldx, T *y, int64_t ldy, hipStream_t st)
{
    if (m->rows == 0 || n == 0) return 0;
    // matrices far larger than the caches are streamed with non-temporal loads
    const bool nt = m->stored * (int64_t)(sizeof(T) + 4) > ((int64_t)512 << 20);
    int v = 0;
    while (v < n) {
        const int left = n - v;
        const T *xv = x + (int64_t)v * ldx;
        T *yv = y + (int64_t)v * ldy;
        int rc, took;
        if (left >= 8) {
            took = 8;
            rc = nt ? launch_block<T, 8, true>(m, xv, ldx, yv, ldy, st) : launch_block<T, 8, false>(m, xv, ldx, yv, ldy, st);
        } else if (left >= 4) {
            took = 4;
            rc = nt ? launch_block<T, 4, true>(m, xv, ldx, yv, ldy, st) : launch_block<T, 4, false>(m, xv, ldx, yv, ldy, st);
        } else if (left >= 2) {
            took = 2;
            rc = nt ? launch_block<T, 2, true>(m, xv, ldx, yv, ldy, st) : launch_block<T, 2, false>(m, xv, ldx, yv, ldy, st);
        } else {
            took = 1;
            rc = nt ? launch_block<T, 1, true>(m, xv, ldx, yv, ldy, st) : launch_block<T, 1, false>(m, xv, ldx, yv, ldy, st);
        }
        if (rc) return rc;
        v += took;
    }
    SPMV_HIP_TRY(hipGetLastError(), return 1);
    return 0;
}

int launch_mult_mv(const spmv_hip_matrix *m, int n, const void *d_x, int64_t ldx, void *d_y, int64_t ldy,
                   hipStream_t stream)
{
    if (m->dtype == SPMV_HIP_F32)
        return launch_mult_mv_t<float>(m, n, (const float *)d_x, ldx, (float *)d_y, ldy, stream);
    return launch_mult_mv_t<double>(m, n, (const double *)d_x, ldx, (double *)d_y, ldy, stream);
}

} // namespace spmv
