/*
 * TEST INFRASTRUCTURE ONLY -- see spmv_oracle.h for scope and parity status.
 *
 * Every multiply uses an explicit fma() so that the rounding sequence is the
 * one the reference gets from g++ -O3 -mfma (its `s += x*a` loops are
 * FMA-contracted) and the one the HIP kernels use.
 */
#define _GNU_SOURCE
#include "spmv_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ COO -- */

static int cmp_row_col(const void *pa, const void *pb)
{
    const orc_coo_t *a = (const orc_coo_t *)pa, *b = (const orc_coo_t *)pb;
    if (a->i != b->i) return a->i < b->i ? -1 : 1;
    if (a->j != b->j) return a->j < b->j ? -1 : 1;
    return 0;
}

/* mtx.c:18-34: scan for the first out-of-order neighbour pair, qsort only then */
void orc_sort_coo(int64_t nnz, orc_coo_t *coo)
{
    for (int64_t k = 1; k < nnz; k++) {
        if (cmp_row_col(&coo[k - 1], &coo[k]) > 0) {
            qsort(coo, (size_t)nnz, sizeof(orc_coo_t), cmp_row_col);
            return;
        }
    }
}

/* mtx.c:270-276 */
void orc_mult_coo(int rows, int64_t nnz, const orc_coo_t *coo, const double *x, double *y)
{
    for (int r = 0; r < rows; r++) y[r] = 0.0;
    for (int64_t k = 0; k < nnz; k++)
        y[coo[k].i] = fma(x[coo[k].j], coo[k].a, y[coo[k].i]);
}

void orc_free(void *p) { free(p); }

/* mtx.c:36-137 */
int orc_load_mtx(const char *fname, int *rows, int *cols, int *nnz, orc_coo_t **out)
{
    FILE *f = fopen(fname, "r");
    if (!f) return -1;                                   /* :39-42 */
    char banner[101], object[101], format[101], type[101], symmetry[101];
    if (fscanf(f, "%100s %100s %100s %100s %100s\n", banner, object, format, type, symmetry) != 5) {
        fclose(f); return -2;                            /* :50-54 */
    }
    if (strcmp(banner, "%%MatrixMarket") != 0) { fclose(f); return -3; }             /* :56-59 */
    if (strcmp(object, "matrix") != 0 || strcmp(format, "coordinate") != 0) {        /* :61-65 */
        fclose(f); return -4;
    }
    char line[1000];
    do {                                                  /* :68-74 */
        if (!fgets(line, sizeof line, f)) { fclose(f); return -5; }
    } while (line[0] == '%');
    long m, n, nz;
    if (sscanf(line, "%ld %ld %ld", &m, &n, &nz) != 3) { fclose(f); return -6; }     /* :77-82 */
    if (m > INT_MAX || n > INT_MAX || nz > INT_MAX) { fclose(f); return -7; }        /* :83-86 */
    if (strcmp(type, "complex") == 0) { fclose(f); return -8; }                      /* :100-103 */

    bool mirrored = strcmp(symmetry, "general") != 0;    /* :88: every non-general kind */
    bool pattern = strcmp(type, "pattern") == 0;
    orc_coo_t *coo = (orc_coo_t *)malloc(sizeof(orc_coo_t) * (size_t)(mirrored ? 2 * nz : nz) + 16);
    if (!coo) { fclose(f); return -9; }

    int k = 0;
    double v = 1.0;
    for (long l = 0; l < nz; l++) {
        int i, j;
        if (pattern) {
            if (fscanf(f, "%d %d\n", &i, &j) != 2) { free(coo); fclose(f); return -10; }
        } else {
            if (fscanf(f, "%d %d %le\n", &i, &j, &v) != 3) { free(coo); fclose(f); return -10; }
            if (v == 0.0) continue;                      /* :118 explicit zeros dropped */
        }
        if (i < 1 || i > m || j < 1 || j > n) { free(coo); fclose(f); return -11; }  /* :120-123 */
        coo[k].i = i - 1; coo[k].j = j - 1; coo[k].a = v;
        if (mirrored && i != j) {                        /* :127-132 no sign change */
            coo[k + 1].i = j - 1; coo[k + 1].j = i - 1; coo[k + 1].a = v;
            k++;
        }
        k++;
    }
    fclose(f);
    *rows = (int)m; *cols = (int)n; *nnz = k; *out = coo;
    return 0;
}

/* ------------------------------------------------------------------ CSR -- */

/* csr.c:169-174 */
int64_t orc_csr_count(int rows, int64_t nnz, int alignment, const orc_coo_t *coo)
{
    int64_t p = 0, l = 0;
    for (int k = 0; k < rows; k++) {
        while (l < nnz && coo[l].i == k) { p++; l++; }
        int64_t e = p % alignment;
        if (e) p += alignment - e;
    }
    return p;
}

/* csr.c:181-199 */
void orc_csr_fill(int rows, int64_t nnz, int alignment, const orc_coo_t *coo,
                  int *row_ptr, int *col, double *val)
{
    int64_t p = 0, l = 0;
    row_ptr[0] = 0;
    for (int k = 0; k < rows; k++) {
        while (l < nnz && coo[l].i == k) {
            col[p] = coo[l].j; val[p] = coo[l].a;
            p++; l++;
        }
        if (alignment != 1) {
            while (p % alignment) { col[p] = 0; val[p] = 0.0; p++; }
        }
        row_ptr[k + 1] = (int)p;
    }
}

/* csr.c:4-13 */
void orc_mult_csr_base(int rows, const int *row_ptr, const int *col, const double *val,
                       const double *x, double *y)
{
    for (int k = 0; k < rows; k++) {
        double s = 0.0;
        for (int l = row_ptr[k]; l < row_ptr[k + 1]; l++) s = fma(x[col[l]], val[l], s);
        y[k] = s;
    }
}

void orc_mult_mv_csr(int rows, const int *row_ptr, const int *col, const double *val, int n,
                     const double *x, int64_t ldx, double *y, int64_t ldy)
{
    for (int v = 0; v < n; v++) orc_mult_csr_base(rows, row_ptr, col, val, x + v * ldx, y + v * ldy);
}

void orc_mult_csr_base_f32(int rows, const int *row_ptr, const int *col, const float *val,
                           const float *x, float *y)
{
    for (int k = 0; k < rows; k++) {
        float s = 0.0f;
        for (int l = row_ptr[k]; l < row_ptr[k + 1]; l++) s = fmaf(x[col[l]], val[l], s);
        y[k] = s;
    }
}

/* csr_numa.c:3-23 pattern: static row blocks over the OpenMP team */
void orc_mult_csr_omp(int rows, const int *row_ptr, const int *col, const double *val,
                      const double *x, double *y)
{
    #pragma omp parallel for schedule(static)
    for (int k = 0; k < rows; k++) {
        double s = 0.0;
        for (int l = row_ptr[k]; l < row_ptr[k + 1]; l++) s = fma(x[col[l]], val[l], s);
        y[k] = s;
    }
}

/* csr_numa.c:25-58 pattern: every thread writes (first touches) the rows it will later multiply, so
 * that their pages land in the memory closest to that thread.  The destination arrays must be fresh
 * allocations whose pages have not been written yet; row_ptr_dst receives all rows+1 pointers. */
void orc_csr_copy_first_touch(int rows, const int *row_ptr, const int *col, const double *val,
                              int *row_ptr_dst, int *col_dst, double *val_dst, double *x_dst, double *y_dst)
{
    #pragma omp parallel for schedule(static)
    for (int k = 0; k < rows; k++) {
        row_ptr_dst[k] = row_ptr[k];
        for (int l = row_ptr[k]; l < row_ptr[k + 1]; l++) { col_dst[l] = col[l]; val_dst[l] = val[l]; }
        x_dst[k] = 0.0;
        y_dst[k] = 0.0;
    }
    row_ptr_dst[rows] = row_ptr[rows];
}

/* ----------------------------------------------------------- merge path -- */

static inline int imin(int a, int b) { return a < b ? a : b; }

/* csr_merge.c:62-85 */
void orc_merge_path_search(int diagonal, const int *row_end_offsets, int n, int nz,
                           int *a_coord, int *b_coord)
{
    int lo = diagonal > nz ? diagonal - nz : 0;
    int hi = imin(diagonal, n);
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (row_end_offsets[mid] <= diagonal - mid - 1) lo = mid + 1;
        else hi = mid;
    }
    *a_coord = imin(lo, n);
    *b_coord = diagonal - lo;
}

/* csr_merge.c:15-59 (the per-thread bodies are independent, so a serial sweep
 * over tid gives the same numbers as the OpenMP team) */
void orc_mult_csr_merge(int n, int nz, const int *row_ptr, const int *col, const double *val,
                        const double *x, double *y, int nthreads,
                        int *row_carry, double *value_carry)
{
    const int *row_end = row_ptr + 1;
    int items = n + nz;
    int per_thread = (items + nthreads - 1) / nthreads;
    for (int tid = 0; tid < nthreads; tid++) {
        int d0 = imin(per_thread * tid, items);
        int d1 = imin(d0 + per_thread, items);
        int a, b, a_end, b_end;
        orc_merge_path_search(d0, row_end, n, nz, &a, &b);
        orc_merge_path_search(d1, row_end, n, nz, &a_end, &b_end);
        double run = 0.0;
        for (; a < a_end; a++) {
            for (; b < row_end[a]; b++) run = fma(val[b], x[col[b]], run);
            y[a] = run;
            run = 0.0;
        }
        for (; b < b_end; b++) run = fma(val[b], x[col[b]], run);
        row_carry[tid] = a_end;
        value_carry[tid] = run;
    }
    for (int tid = 0; tid < nthreads; tid++)
        if (row_carry[tid] < n) y[row_carry[tid]] += value_carry[tid];
}

/* --------------------------------------------------------- balanced CSR -- */

/* csr_bal.c:83-85 (chunk sizes) and :42-50 (first/last row, distinct rows) */
void orc_csr_bal_partition(int64_t nnz, int threads, const orc_coo_t *coo,
                           int64_t *start, int64_t *len, int *first_row, int *last_row,
                           int *nrows_t)
{
    int64_t s = 0;
    for (int t = 0; t < threads; t++) {
        int64_t L = (nnz + t) / threads;
        start[t] = s; len[t] = L;
        const orc_coo_t *c = coo + s;
        first_row[t] = L > 0 ? c[0].i : -1;
        last_row[t] = L > 0 ? c[L - 1].i : -1;
        int r = 0;
        for (int64_t l = 0; l < L; l++)
            if (l == 0 || c[l].i != c[l - 1].i) r++;
        nrows_t[t] = r;
        s += L;
    }
}

/* csr_bal.c:3-38 */
void orc_mult_csr_bal(int64_t nnz, int threads, const orc_coo_t *coo,
                      const double *x, double *y)
{
    int64_t *start = (int64_t *)malloc(sizeof(int64_t) * threads * 2);
    int64_t *len = start + threads;
    int *fr = (int *)malloc(sizeof(int) * threads * 3);
    int *lr = fr + threads, *nr = lr + threads;
    double *buffer = (double *)calloc((size_t)threads, sizeof(double));
    orc_csr_bal_partition(nnz, threads, coo, start, len, fr, lr, nr);
    for (int t = 0; t < threads; t++) {
        const orc_coo_t *c = coo + start[t];
        int64_t l = 0;
        for (int k = 0; k < nr[t]; k++) {
            double s = 0.0;
            int row = c[l].i;
            for (; l < len[t] && c[l].i == row; l++) s = fma(x[c[l].j], c[l].a, s);
            if (k == 0 && t > 0 && fr[t] == lr[t - 1]) buffer[t - 1] = s;   /* :17-19 */
            else y[fr[t] + k] = s;                                          /* :21,29 */
        }
    }
    for (int t = 1; t < threads; t++)                                       /* :33-37 */
        if (fr[t] == lr[t - 1]) y[fr[t]] += buffer[t - 1];
    free(start); free(fr); free(buffer);
}

/* ---------------------------------------------------------- SELL-C-sigma -- */

/* radix_sort.c:36-76: one stable counting pass on a 4-bit digit, larger digits first */
static void radix_pass_desc(int *key, int *pay, int start, int end, unsigned shift,
                            int *tkey, int *tpay)
{
    int len = end - start;
    int64_t bins[16] = {0};
    for (int i = start; i < end; i++) bins[(key[i] >> shift) & 0xF]++;
    int64_t cum = 0;
    for (int b = 0; b < 16; b++) { cum += bins[b]; bins[b] = cum; }
    for (int i = start; i < end; i++) {
        int b = (key[i] >> shift) & 0xF;
        int64_t at = len - bins[b];
        tkey[at] = key[i]; tpay[at] = pay[i];
        bins[b]--;
    }
    memcpy(key + start, tkey, sizeof(int) * (size_t)len);
    memcpy(pay + start, tpay, sizeof(int) * (size_t)len);
}

/* sellcs_utils.c:36-48 (row lengths) + radix_sort.c:103-122 (windows) + :78-101 (digits) */
void orc_sell_row_order(int nrows, const int *row_ptr, int64_t sigma,
                        int *rows_size, int *row_order)
{
    for (int r = 0; r < nrows; r++) { rows_size[r] = row_ptr[r + 1] - row_ptr[r]; row_order[r] = r; }
    int64_t cap = sigma < nrows ? sigma : nrows;
    int *tkey = (int *)malloc(sizeof(int) * (size_t)(cap > 0 ? cap : 1) * 2);
    int *tpay = tkey + (cap > 0 ? cap : 1);
    for (int64_t k = 0; k < nrows; k += sigma) {
        int start = (int)k;
        int end = (k + sigma > nrows) ? nrows : (int)(k + sigma);
        int maxv = rows_size[start];
        for (int i = start + 1; i < end; i++) if (rows_size[i] > maxv) maxv = rows_size[i];
        unsigned shift = 0;
        for (unsigned v = (unsigned)maxv; v; v >>= 4, shift += 4)
            radix_pass_desc(rows_size, row_order, start, end, shift, tkey, tpay);
    }
    free(tkey);
}

/* sellcs_format.c:144-160 */
int64_t orc_sell_slices(int nrows, int C, const int *rows_size_sorted,
                        int *slice_widths, int64_t *slice_ptr)
{
    int nslices = (nrows + C - 1) / C;
    int64_t total = 0;
    for (int s = 0; s < nslices; s++) {
        int w = rows_size_sorted[(int64_t)s * C];       /* first row of the slice */
        slice_ptr[s] = total;
        slice_widths[s] = w;
        total += (int64_t)w * C;
    }
    slice_ptr[nslices] = total;
    return total;
}

/* sellcs_format.c:164-187 */
void orc_sell_fill(int nrows, int C, const int *row_ptr, const int *col, const double *val,
                   const int *row_order, const int64_t *slice_ptr, int64_t padded,
                   int *sell_col, double *sell_val)
{
    memset(sell_col, 0, sizeof(int) * (size_t)padded);
    memset(sell_val, 0, sizeof(double) * (size_t)padded);
    for (int r = 0; r < nrows; r++) {
        int64_t w = slice_ptr[r / C] + (r % C);
        int src = row_order[r];
        for (int i = row_ptr[src]; i < row_ptr[src + 1]; i++) {
            sell_val[w] = val[i]; sell_col[w] = col[i];
            w += C;
        }
    }
}

/* sellcs_format.c:53-136, sellcs_utils.c:79-125 */
int64_t orc_sell_dfc(int nrows, int C, const int *rows_size_sorted, const int *row_ptr, const int *col,
                     const double *val, const int *row_order, int *slice_widths, int64_t *vop_ptr,
                     unsigned char *vop_len, int64_t *slice_ptr, int *dfc_col, double *dfc_val)
{
    int nslices = (nrows + C - 1) / C;
    int64_t vops = 0;
    for (int s = 0; s < nslices; s++) {              /* sellcs_format.c:66-76 */
        slice_widths[s] = rows_size_sorted[s * C];
        vop_ptr[s] = vops;
        vops += rows_size_sorted[s * C];
    }
    vop_ptr[nslices] = vops;
    if (!vop_len) return vops;
    /* sellcs_utils.c:79-125: walk each slice from its shortest row up; every time the row size
     * grows by d, d more columns exist with the lanes of the shorter rows switched off */
    int64_t at = 0;
    for (int i = 0; i < nrows; i += C) {
        int last_row = i + C > nrows ? nrows - 1 : i + C - 1;
        int64_t prev = 0;
        int to_disable = 0;
        int lanes = last_row - i;                     /* active lanes - 1 */
        for (int j = last_row; j >= i; j--) {
            int64_t cur = rows_size_sorted[j];
            int64_t diff = cur - prev;
            if (diff > 0) {
                lanes -= to_disable;
                for (int64_t k = 0; k < diff; k++) vop_len[at++] = (unsigned char)lanes;
                to_disable = 1;
                prev = cur;
            } else {
                to_disable++;
            }
        }
    }
    /* sellcs_format.c:88-98 entry offsets, :109-127 fill */
    int64_t size = 0, v = 0;
    for (int s = 0; s < nslices; s++) {
        slice_ptr[s] = size;
        for (int w = 0; w < slice_widths[s]; w++) size += vop_len[v++] + 1;
    }
    slice_ptr[nslices] = size;
    for (int s = 0; s < nslices; s++) {
        int64_t ins = slice_ptr[s], vid = vop_ptr[s];
        for (int w = 0; w < slice_widths[s]; w++, vid++)
            for (int r = 0; r < vop_len[vid] + 1; r++) {
                int64_t src = row_ptr[row_order[s * C + r]] + w;
                dfc_val[ins] = val[src]; dfc_col[ins] = col[src];
                ins++;
            }
    }
    return size;
}

void orc_mult_sell_dfc(int nrows, int C, const int *slice_widths, const int64_t *vop_ptr,
                       const unsigned char *vop_len, const int64_t *slice_ptr, const int *row_order,
                       const int *dfc_col, const double *dfc_val, const double *x, double *y)
{
    int nslices = (nrows + C - 1) / C;
    for (int r = 0; r < nrows; r++) y[r] = 0.0;
    for (int s = 0; s < nslices; s++) {
        int64_t at = slice_ptr[s];
        for (int w = 0; w < slice_widths[s]; w++) {
            int active = vop_len[vop_ptr[s] + w] + 1;
            for (int k = 0; k < active; k++, at++) {
                int row = row_order[s * C + k];
                y[row] = fma(x[dfc_col[at]], dfc_val[at], y[row]);
            }
        }
    }
}

/* sellcs_mv_autovector.c:19-55 */
void orc_mult_sell(int nrows, int C, const int *slice_widths, const int64_t *slice_ptr,
                   const int *row_order, const int *sell_col, const double *sell_val,
                   const double *x, double *y)
{
    int nslices = (nrows + C - 1) / C;
    for (int s = 0; s < nslices; s++) {
        const double *v = sell_val + slice_ptr[s];
        const int *c = sell_col + slice_ptr[s];
        int row0 = s * C;
        int lanes = nrows - row0 < C ? nrows - row0 : C;
        for (int j = 0; j < slice_widths[s]; j++) {
            for (int k = 0; k < lanes; k++) {
                int dst = row_order[row0 + k];
                y[dst] = fma(x[c[k]], v[k], y[dst]);
            }
            v += C; c += C;
        }
    }
}

/* ------------------------------------------------- sliced ELL (SELLP/ELL) -- */

/* sellp.c:361-421, ell.c:54-82: rows past `rows` in the last block are empty */
int64_t orc_sliced_ell_blocks(int rows, int B, const int *row_ptr, int64_t *block_ptr)
{
    int nb = (rows + B - 1) / B;
    int64_t p = 0;
    block_ptr[0] = 0;
    for (int b = 0; b < nb; b++) {
        int maxlen = 0;
        for (int s = 0; s < B; s++) {
            int r = b * B + s;
            int len = r < rows ? row_ptr[r + 1] - row_ptr[r] : 0;
            if (len > maxlen) maxlen = len;
        }
        p += maxlen;
        block_ptr[b + 1] = p * B;
    }
    return block_ptr[nb];
}

/* sellp.c:433-453, ell.c:89-108 */
void orc_sliced_ell_fill(int rows, int B, const int *row_ptr, const int *col, const double *val,
                         const int64_t *block_ptr, int *j, double *A)
{
    int nb = (rows + B - 1) / B;
    for (int b = 0; b < nb; b++) {
        int64_t width = (block_ptr[b + 1] - block_ptr[b]) / B;
        for (int s = 0; s < B; s++) {
            int r = b * B + s;
            int beg = r < rows ? row_ptr[r] : 0;
            int len = r < rows ? row_ptr[r + 1] - row_ptr[r] : 0;
            for (int64_t k = 0; k < width; k++) {
                int64_t at = block_ptr[b] + k * B + s;
                if (k < len) { j[at] = col[beg + k]; A[at] = val[beg + k]; }
                else { j[at] = 0; A[at] = 0.0; }
            }
        }
    }
}

/* sellp.c:321-342, ell.c:28-45 */
void orc_mult_sliced_ell(int rows, int B, const int64_t *block_ptr, const int *j, const double *A,
                         const double *x, double *y)
{
    int nb = (rows + B - 1) / B;
    double *t = (double *)malloc(sizeof(double) * (size_t)B);
    for (int b = 0; b < nb; b++) {
        for (int s = 0; s < B; s++) t[s] = 0.0;
        for (int64_t l = block_ptr[b]; l < block_ptr[b + 1]; l += B)
            for (int s = 0; s < B; s++) t[s] = fma(x[j[l + s]], A[l + s], t[s]);
        for (int s = 0; s < B && b * B + s < rows; s++) y[b * B + s] = t[s];
    }
    free(t);
}

/* --------------------------------------------------------------- driver -- */

/* driver_spmv.c:21-26 */
double orc_vec_norm(int n, const double *x)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s = fma(x[i], x[i], s);
    return sqrt(s);
}

/* driver_spmv.c:278-289 */
double orc_rel_l2(int n, const double *y, const double *y_ref)
{
    double err = 0.0, nrm = 0.0;
    for (int i = 0; i < n; i++) {
        nrm += y_ref[i] * y_ref[i];
        double d = fabs(y[i] - y_ref[i]);
        err += d * d;
    }
    return nrm != 0.0 ? sqrt(err) / sqrt(nrm) : sqrt(err);
}

/* driver_spmv.c:267-273 */
void orc_driver_init_xy(int columns, double *x, double *y)
{
    for (int i = 0; i < columns; i++) {
        x[i] = ((double)rand()) / RAND_MAX;
        y[i] = ((double)rand()) / RAND_MAX;
    }
    double norm = orc_vec_norm(columns, x);
    for (int i = 0; i < columns; i++) x[i] /= norm;
}
