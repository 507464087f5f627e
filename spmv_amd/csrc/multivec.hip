// Products with a block of vectors: y_v = A * x_v for v = 0..n-1, the operation the reference
// declares as mult_mv_csr(struct csr *, int n, double *x, double *y) (src/spmv.h:54) and never
// defines.  Vector v is the contiguous array x + v*ldx, its result y + v*ldy.
//
// The matrix is streamed once for NV vectors at a time (blocks of 8, then the 1..7 that are left),
// so its 12 bytes per entry are shared by NV products.  Every
// kernel here is lane-per-row with one fma per entry in ascending position, i.e. every column of
// the result is bit-identical to mult_csr_base (reference csr.c:4-13) of that vector, whatever
// the format of the handle:
//   sliced formats of height 64 (SELL-C-sigma, SELLP, ELL)  -> sliced64_mv_kernel; with described slice columns
//                                                              and two or more vectors sliced64_mv_rows_kernel
//   sliced formats of other heights                         -> sliced_any_mv_kernel
//   SELL without padding                                    -> sell_dfc_mv_kernel (full columns as in the rows kernel)
//   CSR, merge-path CSR, balanced CSR (all hold plain CSR)  -> csr_rows_gather_mv_kernel (one vector:
//                                                              csr_rows_lds_mv_kernel)

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

// How the 64 rows of a wave (lane = row) lie to each other, for the walk of runs below: neighbours in ascending order
// everywhere (gap_at = -2), or everywhere except between the lanes gap_at and gap_at + 1 -- what the window sort of
// SELL-C-sigma leaves where it took the two end rows of a grid line out of the interior rows.  In a run of three
// columns at the distances d-1, d, d+1 lane l's outer operands are the middle operands x[row + d] of lanes l-1 and
// l+1; the lanes without such a neighbour (0, 63, and the two beside a gap) find theirs in ONE load that serves all
// NV vectors: lane 2v holds the element before lane 0's window and lane 2v + 1 the one after lane 63's, of vector v;
// with a gap, lane 2 NV + 2v the one after the window of lane gap_at and lane 2 NV + 2v + 1 the one before the
// window of lane gap_at + 1.
struct row_neighbours {
    bool usable;         // the walk of runs applies
    int gap_at;
    int lowest, highest; // smallest and largest row of the wave
    int beside_lanes;    // lanes that take part in the load: 2 NV, with a gap 4 NV
    int beside_vec;      // this lane's vector in it,
    int beside_row;      // the row whose window it sits beside,
    int beside_step;     // and on which side: x[beside_vec * ldx + beside_row + d + beside_step]
    int outer_from;      // lane 2v + outer_from of that load holds the outer operand this lane cannot get from a neighbour
};

template <int NV> __device__ __forceinline__ row_neighbours neighbours_of(int row, int lane, bool all_rows, bool may_gap)
{
    row_neighbours n;
    const int next_row = __shfl_down(row, 1);
    const uint64_t jumps = __ballot(lane < kWave - 1 && next_row != row + 1);
    const bool one_gap = may_gap && all_rows && __popcll(jumps) == 1;
    const int at = one_gap ? __ffsll((unsigned long long)jumps) - 1 : -2;
    const bool gap = one_gap && at >= 1 && at + 1 <= kWave - 2;      // the lanes 0, at, at + 1, 63 are four lanes
    n.usable = all_rows && (jumps == 0 || gap);
    n.gap_at = gap ? at : -2;
    const int row0 = __builtin_amdgcn_readfirstlane(row), row_last = __builtin_amdgcn_readlane(row, kWave - 1);
    const int before_gap = gap ? __builtin_amdgcn_readlane(row, at) : row0;
    const int after_gap = gap ? __builtin_amdgcn_readlane(row, at + 1) : row_last;
    n.lowest = min(row0, after_gap);
    n.highest = max(before_gap, row_last);
    const int second = lane >= 2 * NV;                               // the part of the load that serves the gap
    n.beside_lanes = gap ? 4 * NV : 2 * NV;
    n.beside_vec = (lane - (second ? 2 * NV : 0)) >> 1;
    n.beside_row = (lane & 1) ? (second ? after_gap : row_last) : (second ? before_gap : row0);
    n.beside_step = ((lane & 1) != second) ? 1 : -1;
    n.outer_from = lane == 0 ? 0 : lane == kWave - 1 ? 1 : lane == n.gap_at ? 2 * NV : lane == n.gap_at + 1 ? 2 * NV + 1 : 0;
    return n;
}

// The same product for matrices with described slice columns (sdiag not null) and two or more vectors.  With several
// vectors the product waits on its many small gathers; what a gather costs the CU's texture addresser hardly depends
// on its active lanes or bytes, so the count of vector memory instructions is what to cut.  Where a slice's 64 rows
// are neighbours in ascending order and three columns run along the distances d-1, d, d+1 (the x-neighbours of a
// grid stencil), lane l's operands are x[row + d - 1 .. row + d + 1] = the middle operands of lanes l-1, l, l+1:
// every vector gathers the middle column once with all lanes, the outer two come over the lanes (DPP wave shifts),
// and the element before lane 0's and after lane 63's window come for ALL vectors in one load (lane 2v: before,
// lane 2v + 1: after, of vector v).  A slice whose rows are neighbours except at ONE place (the window sort of
// SELL-C-sigma takes the two end rows of a grid line out of the interior rows) is walked the same way: the two lanes
// beside the gap get their missing operand from two more lanes per vector of that one load.  Two or three such runs
// are in flight together.  Same operands, same order of the sums: the bits of sliced64_mv_kernel.  A zero value in a described column is padding and multiplies x_v[0]
// (sellcs_format.c:167,171), as there.
template <typename T, bool PERM, int NV, bool NT>
__global__ __launch_bounds__(kBlock) void sliced64_mv_rows_kernel(int rows, int cols, int nslices,
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
    const row_neighbours nb = neighbours_of<NV>(row, lane, s * kWave + kWave <= rows, PERM);
    const int *dg = sdiag + (base >> 6);
    const int last = max(cols - 1, 0);
    T acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0;
    auto one_column = [&](int jj, int d) {
        const T vv = stream_load<T, NT>(v + (int64_t)jj * kWave);
        int cc;
        if (d != INT_MIN) cc = vv == (T)0 ? 0 : min(max(row + d, 0), last);
        else cc = stream_load<int, NT>(c + (int64_t)jj * kWave);
        T xx[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) xx[k] = x[k * ldx + cc];
#pragma unroll
        for (int k = 0; k < NV; k++) acc[k] = fma_t(xx[k], vv, acc[k]);
    };
    // a run whose operands all lie inside x (a padding lane of a described column may point anywhere)
    auto is_run = [&](int d0, int d1, int d2) {
        return nb.usable && d0 != INT_MIN && d1 == d0 + 1 && d2 == d0 + 2 && nb.lowest + d0 >= 0 && nb.highest + d2 <= last;
    };
    auto runs = [&](int jj, const int *dmid, auto count) {
        constexpr int R = decltype(count)::value;
        T vv[R][3], mid[R][NV], beside[R];
#pragma unroll
        for (int t = 0; t < R; t++) {
#pragma unroll
            for (int u = 0; u < 3; u++) vv[t][u] = stream_load<T, NT>(v + (int64_t)(jj + 3 * t + u) * kWave);
#pragma unroll
            for (int k = 0; k < NV; k++) mid[t][k] = x[k * ldx + row + dmid[t]];
            beside[t] = 0;
            if (lane < nb.beside_lanes) beside[t] = x[nb.beside_vec * ldx + nb.beside_row + dmid[t] + nb.beside_step];
        }
        bool zero = false;
#pragma unroll
        for (int t = 0; t < R; t++) zero |= vv[t][0] == (T)0 || vv[t][1] == (T)0 || vv[t][2] == (T)0;
        const bool padded = __ballot(zero) != 0;
        auto sums = [&](auto check) {
            constexpr bool CHECK = decltype(check)::value;
#pragma unroll
            for (int t = 0; t < R; t++)
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const T outer = __shfl(beside[t], 2 * k + nb.outer_from);
                    T lo = neighbour_lane<true>(mid[t][k], outer), hi = neighbour_lane<false>(mid[t][k], outer);
                    if (lane == nb.gap_at + 1) lo = outer;
                    if (lane == nb.gap_at) hi = outer;
                    const T x0 = CHECK ? x[k * ldx] : (T)0;
                    acc[k] = fma_t(CHECK && vv[t][0] == (T)0 ? x0 : lo, vv[t][0], acc[k]);
                    acc[k] = fma_t(CHECK && vv[t][1] == (T)0 ? x0 : mid[t][k], vv[t][1], acc[k]);
                    acc[k] = fma_t(CHECK && vv[t][2] == (T)0 ? x0 : hi, vv[t][2], acc[k]);
                }
        };
        if (padded) sums(std::true_type{});
        else sums(std::false_type{});
    };
    constexpr int RB = NV >= 5 ? 2 : 3;
    int j = 0;
    while (j < width) {
        int d[3 * RB];
        int nr = 0;
#pragma unroll
        for (int t = 0; t < RB; t++) {
            if (nr == t && j + 3 * (t + 1) <= width) {
                d[3 * t] = __builtin_amdgcn_readfirstlane(dg[j + 3 * t]);
                d[3 * t + 1] = __builtin_amdgcn_readfirstlane(dg[j + 3 * t + 1]);
                d[3 * t + 2] = __builtin_amdgcn_readfirstlane(dg[j + 3 * t + 2]);
                if (is_run(d[3 * t], d[3 * t + 1], d[3 * t + 2])) nr = t + 1;
            }
        }
        if (nr == RB) {
            int dmid[RB];
#pragma unroll
            for (int t = 0; t < RB; t++) dmid[t] = d[3 * t + 1];
            runs(j, dmid, std::integral_constant<int, RB>{});
            j += 3 * RB;
        } else if (nr >= 1) {
            runs(j, &d[1], std::integral_constant<int, 1>{});
            j += 3;
        } else {
            one_column(j, __builtin_amdgcn_readfirstlane(dg[j]));
            j += 1;
        }
    }
    if (r < rows) {
#pragma unroll
        for (int k = 0; k < NV; k++) y[k * ldy + row] = acc[k];
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
// The start of each column is carried as a wave-uniform counter.  The columns below the slice's shortest row hold
// all 64 lanes: with descriptors (sdiag, one int per column from vop_ptr[s] on; may be null) and two or more
// vectors they are walked like the columns of sliced64_mv_rows_kernel -- described columns compute their indices,
// runs of three neighbouring distances over 64 neighbouring rows gather only the middle column.  There is no
// padding in this layout, so every lane of a described column holds a real entry.
template <typename T, int NV, int U, bool NT>
__global__ __launch_bounds__(kBlock) void sell_dfc_mv_kernel(int rows, int nslices,
                                                             const int64_t *__restrict__ slice_ptr,
                                                             const int *__restrict__ sorted_len,
                                                             const int *__restrict__ row_order,
                                                             const int64_t *__restrict__ vop_ptr,
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
    const int r = s * kWave + lane;
    const int len = r < rows ? sorted_len[r] : 0;
    const int width = __builtin_amdgcn_readfirstlane(len);
    const int *c = scol + slice_ptr[s] + lane;
    const T *v = sval + slice_ptr[s] + lane;
    int run = 0;
    T acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0;
    int first = 0;                                           // columns before it are done
    if (NV >= 2 && sdiag) {
        const int shortest = __builtin_amdgcn_readlane(len, kWave - 1);       // columns below it hold all 64 lanes
        const int row = r < rows ? row_order[r] : 0;
        const row_neighbours nb = neighbours_of<NV>(row, lane, s * kWave + kWave <= rows, true);
        const int *dg = sdiag + vop_ptr[s];
        auto runs = [&](const int *dmid, auto count) {
            constexpr int R = decltype(count)::value;
            T vv[R][3], mid[R][NV], beside[R];
#pragma unroll
            for (int t = 0; t < R; t++) {
#pragma unroll
                for (int u = 0; u < 3; u++) vv[t][u] = stream_load<T, NT>(v + run + (3 * t + u) * kWave);
#pragma unroll
                for (int k = 0; k < NV; k++) mid[t][k] = x[k * ldx + row + dmid[t]];
                beside[t] = 0;
                if (lane < nb.beside_lanes) beside[t] = x[nb.beside_vec * ldx + nb.beside_row + dmid[t] + nb.beside_step];
            }
#pragma unroll
            for (int t = 0; t < R; t++)
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const T outer = __shfl(beside[t], 2 * k + nb.outer_from);
                    T lo = neighbour_lane<true>(mid[t][k], outer), hi = neighbour_lane<false>(mid[t][k], outer);
                    if (lane == nb.gap_at + 1) lo = outer;
                    if (lane == nb.gap_at) hi = outer;
                    acc[k] = fma_t(lo, vv[t][0], acc[k]);
                    acc[k] = fma_t(mid[t][k], vv[t][1], acc[k]);
                    acc[k] = fma_t(hi, vv[t][2], acc[k]);
                }
            run += 3 * R * kWave;
        };
        constexpr int RB = NV >= 5 ? 2 : 3;
        int j = 0;
        while (j < shortest) {
            int d[3 * RB];
            int nr = 0;
#pragma unroll
            for (int t = 0; t < RB; t++) {
                if (nr == t && j + 3 * (t + 1) <= shortest) {
                    d[3 * t] = __builtin_amdgcn_readfirstlane(dg[j + 3 * t]);
                    d[3 * t + 1] = __builtin_amdgcn_readfirstlane(dg[j + 3 * t + 1]);
                    d[3 * t + 2] = __builtin_amdgcn_readfirstlane(dg[j + 3 * t + 2]);
                    if (nb.usable && d[3 * t] != INT_MIN && d[3 * t + 1] == d[3 * t] + 1 && d[3 * t + 2] == d[3 * t] + 2) nr = t + 1;
                }
            }
            if (nr == RB) {
                int dmid[RB];
#pragma unroll
                for (int t = 0; t < RB; t++) dmid[t] = d[3 * t + 1];
                runs(dmid, std::integral_constant<int, RB>{});
                j += 3 * RB;
            } else if (nr >= 1) {
                runs(&d[1], std::integral_constant<int, 1>{});
                j += 3;
            } else {
                const int dj = __builtin_amdgcn_readfirstlane(dg[j]);
                const int cc = dj != INT_MIN ? row + dj : stream_load<int, NT>(c + run);
                const T vv = stream_load<T, NT>(v + run);
                T xx[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) xx[k] = x[k * ldx + cc];
#pragma unroll
                for (int k = 0; k < NV; k++) acc[k] = fma_t(xx[k], vv, acc[k]);
                run += kWave;
                j += 1;
            }
        }
        first = shortest;
    }
    for (int j0 = first; j0 < width; j0 += U) {
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

// The same product with the gathers made by rows.  The chunk parks only (column, value) -- for a patterned block the
// value alone -- and every lane gathers the NV operands of its own row's entries, U entries at a time: one gather
// instruction then touches the few lines of 64 neighbouring rows at one position of their rows (as in the sliced
// kernels) instead of the lines of 64 consecutive entries of two or three rows.  A chunk of 64*IPL entries normally
// holds the whole block, so that all 64 rows are busy.  Entry k sits at k + k/64 in LDS: rows of an even length
// would otherwise meet on a few banks.
template <typename T, int IPL, int NV, int U, bool NT>
__global__ __launch_bounds__(kWave) void csr_rows_gather_mv_kernel(int rows, const int *__restrict__ row_ptr,
                                                                   const int *__restrict__ col,
                                                                   const T *__restrict__ val,
                                                                   const unsigned char *__restrict__ pattern_len,
                                                                   const int *__restrict__ row_map,
                                                                   const T *__restrict__ x, int64_t ldx,
                                                                   T *__restrict__ y, int64_t ldy, int map)
{
    constexpr int kCap = kWave * IPL;
    __shared__ int scol[kCap + kCap / kWave];
    __shared__ T sval[kCap + kCap / kWave];
    __shared__ int dist[32];
    auto skew = [](int k) { return k + (k >> 6); };
    const int blk = (int)xcd_block(blockIdx.x, gridDim.x, map);
    const int lane = threadIdx.x;
    const int row = blk * kWave + lane;
    const int last = min(blk * kWave + kWave, rows);
    const int rs = row_ptr[min(row, rows)];
    const int re = row < rows ? row_ptr[row + 1] : rs;
    const int run_begin = __builtin_amdgcn_readfirstlane(rs);
    const int run_end = row_ptr[last];
    // wave-uniform: every row of a patterned block (all 64 rows inside the matrix) has the columns row + dist[0..L)
    const int L = pattern_len && last == blk * kWave + kWave ? pattern_len[blk] : 0;
    unsigned run_at = 0;                 // bit p: dist[p], dist[p + 1], dist[p + 2] are consecutive numbers
    if (L) {
        if (lane < L) dist[lane] = col[run_begin + lane] - blk * kWave;
        __syncthreads();
        run_at = (unsigned)__ballot(lane + 2 < L && dist[min(lane + 1, 31)] == dist[min(lane, 31)] + 1 &&
                                    dist[min(lane + 2, 31)] == dist[min(lane, 31)] + 2);
    }
    // a patterned block that fits one chunk is walked by positions, all rows side by side
    const bool by_position = L && run_end - (run_begin & ~31) <= kCap;
    T acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0;
    for (int e0 = run_begin & ~31; e0 < run_end; e0 += kCap) {
        const int n = min(kCap, run_end - e0);
        T vv[IPL];
#pragma unroll
        for (int i = 0; i < IPL; i++)
            vv[i] = stream_load<T, NT>(val + min(max(e0 + lane + i * kWave, run_begin), run_end - 1));
        if (!L) {
            int cc[IPL];
#pragma unroll
            for (int i = 0; i < IPL; i++)
                cc[i] = stream_load<int, NT>(col + min(max(e0 + lane + i * kWave, run_begin), run_end - 1));
#pragma unroll
            for (int i = 0; i < IPL; i++) scol[skew(lane + i * kWave)] = cc[i];
        }
#pragma unroll
        for (int i = 0; i < IPL; i++) sval[skew(lane + i * kWave)] = vv[i];
        __syncthreads();
        if (by_position) {
            // Three positions at consecutive distances d-1, d, d+1: the operands of lane l are x[row + d - 1 .. + 1] =
            // the middle operands of lanes l-1, l, l+1.  All lanes gather the middle one; the outer two come over the
            // lanes (DPP wave shifts); the element before row 0's and after row 63's window come for all vectors in
            // one load (lane 2v: before, lane 2v + 1: after, of vector v).  R runs are in flight together.
            const int mine = rs - e0;                            // this row's first entry in the chunk
            const int row0 = blk * kWave;
            auto runs = [&](int p, auto count) {
                constexpr int R = decltype(count)::value;
                T v[R][3], mid[R][NV], beside[R];
#pragma unroll
                for (int t = 0; t < R; t++) {
#pragma unroll
                    for (int u = 0; u < 3; u++) v[t][u] = sval[skew(mine + p + 3 * t + u)];
                    const int d = dist[p + 3 * t + 1];
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) mid[t][kk] = x[kk * ldx + row + d];
                    beside[t] = 0;
                    if (lane < 2 * NV) beside[t] = x[(lane >> 1) * ldx + row0 + ((lane & 1) ? kWave + d : d - 1)];
                }
#pragma unroll
                for (int t = 0; t < R; t++)
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) {
                        const T outer = __shfl(beside[t], 2 * kk + (lane == kWave - 1 ? 1 : 0));
                        const T lo = neighbour_lane<true>(mid[t][kk], outer), hi = neighbour_lane<false>(mid[t][kk], outer);
                        acc[kk] = fma_t(lo, v[t][0], acc[kk]);
                        acc[kk] = fma_t(mid[t][kk], v[t][1], acc[kk]);
                        acc[kk] = fma_t(hi, v[t][2], acc[kk]);
                    }
            };
            constexpr int RB = NV >= 5 ? 2 : 3;
            int p = 0;
            while (p < L) {
                int nr = 0;
                while (nr < RB && p + 3 * (nr + 1) <= L && ((run_at >> (p + 3 * nr)) & 1u)) nr++;
                if (nr == RB) {
                    runs(p, std::integral_constant<int, RB>{});
                    p += 3 * RB;
                } else if (nr >= 1) {
                    runs(p, std::integral_constant<int, 1>{});
                    p += 3;
                } else {
                    const T v = sval[skew(mine + p)];
                    const int c = row + dist[p];
                    T xv[NV];
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) xv[kk] = x[kk * ldx + c];
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) acc[kk] = fma_t(xv[kk], v, acc[kk]);
                    p += 1;
                }
            }
            break;                                               // the block was one chunk
        }
        int k = max(rs, e0) - e0;
        const int kend = min(re, e0 + n) - e0;
        for (; k < kend; k += U) {
            int c[U];
            T v[U], xv[U][NV];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int at = min(k + u, kend - 1);
                v[u] = sval[skew(at)];
                c[u] = L ? row + dist[e0 + at - rs] : scol[skew(at)];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int kk = 0; kk < NV; kk++) xv[u][kk] = x[kk * ldx + c[u]];
#pragma unroll
            for (int u = 0; u < U; u++)
                if (k + u < kend) {
#pragma unroll
                    for (int kk = 0; kk < NV; kk++) acc[kk] = fma_t(xv[u][kk], v[u], acc[kk]);
                }
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
    constexpr int U = NV >= 7 ? 2 : NV >= 5 ? 3 : NV >= 3 ? 4 : 8;
    // A window sort (SELL-C-sigma, with or without padding) gathers a window's short rows in its last slices: with
    // windows of 1024 rows that is every 16th slice = every 4th workgroup, and dealt one by one, in even runs or as
    // contiguous eighths those workgroups all land on the same shader engine of an XCD (8 vectors at C2: 2.2 ms against
    // 1.85 in runs of 3 or 5).  With several vectors such a handle therefore deals runs of an odd length: its own
    // block_map where that is one, else 5.
    const bool windows = NV >= 2 && m->slice_height == kWave && (m->format == SPMV_HIP_SELL_DFC || (m->format == SPMV_HIP_SELL && m->d_row_order));
    const int map = windows && !(m->block_map >= 3 && (m->block_map & 1)) ? 5 : m->block_map;
    switch (m->format) {
    case SPMV_HIP_CSR_SCALAR:
    case SPMV_HIP_CSR_VECTOR:
    case SPMV_HIP_CSR_MERGE:
    case SPMV_HIP_CSR_BAL:
        if constexpr (NV == 1)                 // one vector: gathers in entry order, (x, value) parked
            hipLaunchKernelGGL((csr_rows_lds_mv_kernel<T, 4, NV, NT>), dim3(blocks_for(m->rows, kWave)), dim3(kWave), 0, st,
                               m->rows, m->d_row_ptr, m->d_col, (const T *)m->d_val,
                               m->d_block_pat, m->d_row_map, x, ldx, y, ldy, m->block_map);
        else                                   // gathers by rows; 28 * 64 entries hold a block of 64 rows of up to 28 entries
            hipLaunchKernelGGL((csr_rows_gather_mv_kernel<T, 28, NV, (NV >= 7 ? 3 : NV >= 5 ? 4 : NV >= 3 ? 5 : 9), NT>),
                               dim3(blocks_for(m->rows, kWave)), dim3(kWave), 0, st,
                               m->rows, m->d_row_ptr, m->d_col, (const T *)m->d_val,
                               m->d_block_pat, m->d_row_map, x, ldx, y, ldy, m->block_map);
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
            if constexpr (NV >= 2) {
                if (diag) {                       // runs of neighbouring distances: outer operands over the lanes
                    if (perm)
                        hipLaunchKernelGGL((sliced64_mv_rows_kernel<T, true, NV, NT>), grid, block, 0, st, m->rows,
                                           m->columns, m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                           ldx, y, ldy, map);
                    else
                        hipLaunchKernelGGL((sliced64_mv_rows_kernel<T, false, NV, NT>), grid, block, 0, st, m->rows,
                                           m->columns, m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                           ldx, y, ldy, map);
                    break;
                }
            }
            if (perm)
                hipLaunchKernelGGL((sliced64_mv_kernel<T, true, NV, U, NT>), grid, block, 0, st, m->rows, m->columns,
                                   m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                   ldx, y, ldy, map);
            else
                hipLaunchKernelGGL((sliced64_mv_kernel<T, false, NV, U, NT>), grid, block, 0, st, m->rows, m->columns,
                                   m->nslices, m->d_slice_ptr, m->d_row_order, diag, m->d_scol, sv, x,
                                   ldx, y, ldy, map);
        } else {
            const dim3 grid(blocks_for(m->rows, kBlock)), block(kBlock);
            if (perm)
                hipLaunchKernelGGL((sliced_any_mv_kernel<T, true, NV>), grid, block, 0, st, m->rows, m->slice_height,
                                   m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, ldx, y, ldy, map);
            else
                hipLaunchKernelGGL((sliced_any_mv_kernel<T, false, NV>), grid, block, 0, st, m->rows, m->slice_height,
                                   m->d_slice_ptr, m->d_row_order, m->d_scol, sv, x, ldx, y, ldy, map);
        }
        break;
    }
    case SPMV_HIP_SELL_DFC:
        hipLaunchKernelGGL((sell_dfc_mv_kernel<T, NV, (NV >= 5 ? 2 : 4), NT>), dim3(blocks_for(m->nslices, kBlock / kWave)),
                           dim3(kBlock), 0, st, m->rows, m->nslices, m->d_slice_ptr, m->d_sorted_len, m->d_row_order,
                           m->d_vop_ptr, m->d_vop_ptr && m->diag_columns * 2 >= m->vops ? m->d_sdiag : nullptr,
                           m->d_scol, (const T *)m->d_sval, x, ldx, y, ldy, map);
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
        const int took = n - v < 8 ? n - v : 8;
        const T *xv = x + (int64_t)v * ldx;
        T *yv = y + (int64_t)v * ldy;
        int rc = 1;
        switch (took) {
#define SPMV_MV_BLOCK(N)                                                                                   \
        case N:                                                                                            \
            rc = nt ? launch_block<T, N, true>(m, xv, ldx, yv, ldy, st) : launch_block<T, N, false>(m, xv, ldx, yv, ldy, st); \
            break;
            SPMV_MV_BLOCK(1)
            SPMV_MV_BLOCK(2)
            SPMV_MV_BLOCK(3)
            SPMV_MV_BLOCK(4)
            SPMV_MV_BLOCK(5)
            SPMV_MV_BLOCK(6)
            SPMV_MV_BLOCK(7)
            SPMV_MV_BLOCK(8)
#undef SPMV_MV_BLOCK
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
