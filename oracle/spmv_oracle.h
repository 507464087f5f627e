/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference SpMV algorithms.
 *
 * Written from scratch in plain C from a reading of the reference; each
 * function cites the reference file:line whose behaviour it restates.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  Nothing in the shipped path
 * (spmv_amd/, include/) links, imports or calls it.
 *
 * Parity status: PINNED.  The reference holds no golden vectors of its own, so
 * this restatement is pinned against (a) the compiled reference itself
 * (oracle/_ref/libspmv_ref.so, built by oracle/Makefile from the reference's
 * sources) in tests/test_oracle_vs_ref.py whenever that library is present and
 * (b) the fixtures under tests/golden/ that tests/golden/make_golden.py wrote
 * from that compiled reference.  The float (fp32) variants have no reference
 * counterpart (the reference is double-only): fp32 parity is unpinned.
 *
 * All arrays are caller-allocated flat buffers so that ctypes/numpy can drive
 * every function.  Indices are 32-bit like the reference (spmv.h:41-47) except
 * slice/block pointers, which are 64-bit here (the reference's int32
 * slice_pointers, sellcs-spmv.h:47, overflow past 2^31 padded entries).
 */
#ifndef SPMV_ORACLE_H
#define SPMV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int i; int j; double a; } orc_coo_t;   /* spmv.h:26 */

/* mtx.c:9-34  sort by (row, col) unless already sorted */
void orc_sort_coo(int64_t nnz, orc_coo_t *coo);
/* mtx.c:270-276  COO product, y zeroed first, array order */
void orc_mult_coo(int rows, int64_t nnz, const orc_coo_t *coo, const double *x, double *y);
/* mtx.c:36-137  MatrixMarket coordinate loader; returns 0 or a negative error
 * code where the reference would exit(1).  *coo is malloc'ed. */
int orc_load_mtx(const char *fname, int *rows, int *cols, int *nnz, orc_coo_t **coo);
void orc_free(void *p);

/* csr.c:160-203  padded entry count, then fill (coo must be sorted) */
int64_t orc_csr_count(int rows, int64_t nnz, int alignment, const orc_coo_t *coo);
void orc_csr_fill(int rows, int64_t nnz, int alignment, const orc_coo_t *coo,
                  int *row_ptr, int *col, double *val);
/* csr.c:4-13 */
void orc_mult_csr_base(int rows, const int *row_ptr, const int *col, const double *val,
                       const double *x, double *y);
/* mult_mv_csr(csr, n, x, y) -- declared at spmv.h:54, defined nowhere in the reference, so parity
 * is unpinned by it.  Restated as n independent mult_csr_base products (csr.c:4-13): vector v is
 * x + v*ldx, its result y + v*ldy. */
void orc_mult_mv_csr(int rows, const int *row_ptr, const int *col, const double *val, int n,
                     const double *x, int64_t ldx, double *y, int64_t ldy);
/* fp32 restatement of csr.c:4-13 (float storage, float accumulate; unpinned) */
void orc_mult_csr_base_f32(int rows, const int *row_ptr, const int *col, const float *val,
                           const float *x, float *y);
/* row-parallel CSR over OpenMP threads (pattern of csr_numa.c:3-23); CPU baseline only */
void orc_mult_csr_omp(int rows, const int *row_ptr, const int *col, const double *val,
                      const double *x, double *y);
/* csr_numa.c:25-58: copy of a CSR whose pages are first written by the threads that multiply them
 * (square matrix: x_dst and y_dst, rows elements each, are touched the same way) */
void orc_csr_copy_first_touch(int rows, const int *row_ptr, const int *col, const double *val,
                              int *row_ptr_dst, int *col_dst, double *val_dst, double *x_dst, double *y_dst);

/* csr_merge.c:62-85 */
void orc_merge_path_search(int diagonal, const int *row_end_offsets, int n, int nz,
                           int *a_coord, int *b_coord);
/* csr_merge.c:15-59  (row_carry/value_carry have nthreads entries) */
void orc_mult_csr_merge(int n, int nz, const int *row_ptr, const int *col, const double *val,
                        const double *x, double *y, int nthreads,
                        int *row_carry, double *value_carry);

/* csr_bal.c:66-94,40-64  chunk t = nnz range [start[t], start[t]+len[t]) of the
 * sorted COO; first_row/last_row/nrows_t as create_csr_bal_thread computes them. */
void orc_csr_bal_partition(int64_t nnz, int threads, const orc_coo_t *coo,
                           int64_t *start, int64_t *len, int *first_row, int *last_row,
                           int *nrows_t);
/* csr_bal.c:3-38 restated on the sorted COO + partition (keeps the reference's
 * "rows in a chunk are consecutive" assumption, csr_bal.c:29, i.e. it is only
 * right for matrices without empty rows -- exactly like the reference). */
void orc_mult_csr_bal(int64_t nnz, int threads, const orc_coo_t *coo,
                      const double *x, double *y);

/* sellcs_utils.c:36-48 + radix_sort.c:103-122 (78-101, 36-76):
 * rows_size[] is produced and sorted in place together with row_order[]. */
void orc_sell_row_order(int nrows, const int *row_ptr, int64_t sigma,
                        int *rows_size, int *row_order);
/* sellcs_format.c:144-160  slice widths/pointers; returns padded entry count.
 * slice_ptr has nslices+1 entries. */
int64_t orc_sell_slices(int nrows, int C, const int *rows_size_sorted,
                        int *slice_widths, int64_t *slice_ptr);
/* sellcs_format.c:164-187  zero-filled column-major slices */
void orc_sell_fill(int nrows, int C, const int *row_ptr, const int *col, const double *val,
                   const int *row_order, const int64_t *slice_ptr, int64_t padded,
                   int *sell_col, double *sell_val);
/* SELL-C-sigma without padding ("DFC": every column of a slice stores only its
 * active lanes, a prefix of the slice because its rows are sorted by length).
 * sellcs_format.c:53-136 with sellcs_set_active_lanes, sellcs_utils.c:79-125.
 * Pass 1 (vop_len == NULL): fills slice_widths[nslices], vop_ptr[nslices+1]
 * and returns the number of columns ("vops").  Pass 2: fills vop_len[vops]
 * (active lanes - 1), slice_ptr[nslices+1] (entry offsets), dfc_col / dfc_val
 * (exactly nnz entries) and returns the entry count. */
int64_t orc_sell_dfc(int nrows, int C, const int *rows_size_sorted, const int *row_ptr, const int *col,
                     const double *val, const int *row_order, int *slice_widths, int64_t *vop_ptr,
                     unsigned char *vop_len, int64_t *slice_ptr, int *dfc_col, double *dfc_val);
/* y = A x from the DFC layout: column by column, active lanes only; each row's
 * entries are added in ascending position with one fma (the reference ships no
 * DFC product for CPUs; the order is that of mult_csr_base, csr.c:4-13). */
void orc_mult_sell_dfc(int nrows, int C, const int *slice_widths, const int64_t *vop_ptr,
                       const unsigned char *vop_len, const int64_t *slice_ptr, const int *row_order,
                       const int *dfc_col, const double *dfc_val, const double *x, double *y);
/* sellcs_mv_autovector.c:19-55  accumulates into y (caller zeroes it); lanes
 * past nrows are skipped (the reference reads uninitialised row_order there). */
void orc_mult_sell(int nrows, int C, const int *slice_widths, const int64_t *slice_ptr,
                   const int *row_order, const int *sell_col, const double *sell_val,
                   const double *x, double *y);

/* sellp.c:361-421 / ell.c:54-82  block pointers for slice height B; returns
 * padded entry count; block_ptr has nb+1 entries, nb = ceil(rows/B). */
int64_t orc_sliced_ell_blocks(int rows, int B, const int *row_ptr, int64_t *block_ptr);
/* sellp.c:433-453 / ell.c:89-108  lane-interleaved fill, pad (j=0, A=0) */
void orc_sliced_ell_fill(int rows, int B, const int *row_ptr, const int *col, const double *val,
                         const int64_t *block_ptr, int *j, double *A);
/* sellp.c:321-342 / ell.c:28-45 */
void orc_mult_sliced_ell(int rows, int B, const int64_t *block_ptr, const int *j, const double *A,
                         const double *x, double *y);

/* driver_spmv.c:21-26 */
double orc_vec_norm(int n, const double *x);
/* driver_spmv.c:278-289  relative L2 error */
double orc_rel_l2(int n, const double *y, const double *y_ref);
/* driver_spmv.c:267-273  x/y init with glibc rand() interleaved, x normalised */
void orc_driver_init_xy(int columns, double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
