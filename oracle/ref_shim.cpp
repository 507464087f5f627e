// TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" harness around the *compiled reference* (its sources are
// compiled by oracle/Makefile from where they lie; nothing is copied here).
// The reference is C compiled with g++, so its symbols are C++-mangled; this
// file re-exports the handful needed to (a) validate oracle/spmv_oracle.c and
// (b) generate tests/golden/ vectors.  It contains no algorithm of its own.
//
// Known reference defects this harness steps around (it does not fix them):
//   * mult_sell accumulates into y (sellcs_mv_autovector.c:48) -> callers zero y.
//   * the last partial SELL slice reads row_order[] past nrows
//     (radix_sort.c:108-113 initialises only nrows entries) -> ref_create_sell
//     is only asked for layouts; ref_mult_sell callers use nrows % 64 == 0 or a
//     y buffer with slack and row_order patched by the caller.
//   * mult_sellp (AVX2) stores a full _BLOCK vector past `rows` (sellp.c:313)
//     -> callers pass y with >= 64 doubles of slack, as driver_spmv.c:264 does.

#include "spmv.h"
#include "sellp.h"
#include "sellcs-spmv.h"
#include "sellcs_mv.h"
#include "radix_sort.h"
#include <omp.h>
#include <string.h>

extern uint64_t sigma_window;   // sell.c:4
extern int32_t max_vlen;        // sell.c:5
void searchPathOnDiag(int diagonal, int *list_A, int nor, int nz, int *list_a_coord, int *list_b_coord); // csr_merge.c:62

extern "C" {

int ref_block(void) { return _BLOCK; }

void ref_load_mtx(const char *fname, int *rows, int *cols, int *nnz, struct mtx **coo) { load_mtx(fname, rows, cols, nnz, coo); }
void ref_load_bin(const char *fname, int *rows, int *cols, int *nnz, struct mtx **coo) { load_bin(fname, rows, cols, nnz, coo); }
void ref_save_bin(const char *fname, int rows, int cols, int nnz, struct mtx *coo) { save_bin(fname, rows, cols, nnz, coo); }
void ref_free(void *p) { free(p); }
void ref_sort_mtx(int nnz, struct mtx *coo) { sort_mtx(nnz, coo, by_row_mtx); }
void ref_mult_mtx(int rows, int cols, int nnz, struct mtx *coo, double *x, double *y) { mult_mtx(rows, cols, nnz, coo, x, y); }

// ---- CSR (csr.c) ----
struct csr *ref_create_csr(int rows, int cols, int nnz, int alignment, struct mtx *coo) { return create_csr(rows, cols, nnz, alignment, coo); }
void ref_free_csr(struct csr *c) { free_csr(c); }
void ref_mult_csr_base(struct csr *c, double *x, double *y) { mult_csr_base(c, x, y); }
void ref_mult_csr_autovec(struct csr *c, double *x, double *y) { mult_csr_autovec(c, x, y); }
void ref_mult_csr_vec(struct csr *c, double *x, double *y) { mult_csr(c, x, y); }

// ---- merge-path CSR (csr_merge.c) ----
struct csr_merge *ref_create_csr_merge(int nthreads, int rows, int cols, int nnz, struct mtx *coo)
{
    omp_set_num_threads(nthreads);
    return create_csr_merge(rows, cols, nnz, coo);
}
void ref_mult_csr_merge(struct csr_merge *c, double *x, double *y) { mult_csr_merge(c, x, y); }
void ref_search_path_on_diag(int diagonal, int *list_A, int nor, int nz, int *a, int *b) { searchPathOnDiag(diagonal, list_A, nor, nz, a, b); }

// ---- balanced CSR (csr_bal.c) ----
struct csr_bal *ref_create_csr_bal(int nthreads, int rows, int cols, int nnz, struct mtx *coo)
{
    omp_set_num_threads(nthreads);
    return create_csr_bal(rows, cols, nnz, coo);
}
void ref_mult_csr_bal(struct csr_bal *c, double *x, double *y) { mult_csr_bal(c, x, y); }
int ref_csr_bal_threads(struct csr_bal *c) { return c->threads; }
struct csr_bal_thread *ref_csr_bal_thread(struct csr_bal *c, int t) { return c->p + t; }

// ---- SELL-C-sigma (sell.c, sellcs_format.c, radix_sort.c, sellcs_mv_autovector.c) ----
void *ref_create_sell(struct csr *c, int C, uint64_t sigma)
{
    max_vlen = C;
    sigma_window = sigma;
    return create_sell(c);
}
void ref_mult_sell(void *m, double *x, double *y) { mult_sell(m, x, y); }
// field accessors (layout of sellcs_matrix_t: sellcs-spmv.h:35-55)
int       ref_sell_nslices(void *m)        { return ((sellcs_matrix_t *)m)->nslices; }
int      *ref_sell_slice_widths(void *m)   { return ((sellcs_matrix_t *)m)->slice_widths; }
int      *ref_sell_slice_pointers(void *m) { return ((sellcs_matrix_t *)m)->slice_pointers; }
int      *ref_sell_row_order(void *m)      { return ((sellcs_matrix_t *)m)->row_order; }
int      *ref_sell_column_indices(void *m) { return ((sellcs_matrix_t *)m)->column_indices; }
double   *ref_sell_values(void *m)         { return ((sellcs_matrix_t *)m)->values; }
int       ref_sell_shift(void *m)          { return ((sellcs_matrix_t *)m)->shift; }
void      ref_free_sell(void *m)           { free_sellcs((sellcs_matrix_t *)m); free(m); }
// the same conversion with enable_dfc = 1 (sellcs_format.c:203-225 -> csr_to_sellcs_dfc :53-136); the
// handle is a zero-initialised sellcs_matrix_t that the tests read through the accessors and never free
void *ref_create_sell_dfc(struct csr *c, int C, uint64_t sigma)
{
    sellcs_matrix_t *m = (sellcs_matrix_t *)calloc(1, sizeof(sellcs_matrix_t));
    sellcs_init_params(C, sigma, m);
    sellcs_create_matrix_from_CSR_rd(c->rows, c->columns, c->i, (const index_t *)c->j, c->A, 0, 1, m);
    return m;
}
unsigned char      *ref_sell_vop_lengths(void *m)  { return ((sellcs_matrix_t *)m)->vop_lengths; }
unsigned long long *ref_sell_vop_pointers(void *m) { return (unsigned long long *)((sellcs_matrix_t *)m)->vop_pointers; }

// ---- SELLP (sellp.c) ----
struct sellp *ref_create_sellp(int rows, int cols, int nnz, const int *row_ptr, const int *col, const double *A) { return create_sellp(rows, cols, nnz, row_ptr, col, A); }
void ref_mult_sellp(struct sellp *s, double *x, double *y) { mult_sellp(s, x, y); }
void ref_mult_sellp_autovec(struct sellp *s, double *x, double *y) { mult_sellp_autovec(s, x, y); }
void ref_free_sellp(struct sellp *s) { free_sellp(s); free(s); }

// ---- ELL (ell.c) ----
struct ell *ref_create_ell(int rows, int cols, int nnz, struct mtx *coo) { return create_ell(rows, cols, nnz, coo); }
void ref_mult_ell(struct ell *e, double *x, double *y) { mult_ell(e, x, y); }

} // extern "C"
