/*
 * libspmv_hip -- C ABI of the MI355X (gfx950) SpMV engine.
 *
 * Drop-in boundary for the reference's create_X / mult_X / free_X plugin
 * interface (reference src/spmv.h, src/sellp.h) as selected by its driver
 * (reference src/driver_spmv.c:186-261).  Plain pointers and sizes only.
 *
 * Conventions
 *   - Every function returning `int` returns 0 on success, non-zero on error;
 *     creators return NULL on error.  spmv_hip_last_error() gives the message
 *     (the reference prints to stderr and exit()s instead: mtx.c:39-41,
 *     spmv.h:277-281; the C driver in spmv_amd/csrc/driver keeps that
 *     behaviour on top of these status codes).
 *   - `dtype`: SPMV_HIP_F64 computes like the reference (double storage and
 *     accumulation); SPMV_HIP_F32 stores and accumulates in float (the
 *     reference has no float path).
 *   - Overwrite semantics for y in every format (the driver pre-fills y with
 *     garbage, driver_spmv.c:269), never writing past `rows`.
 *   - One host thread drives a handle; calls on one handle are ordered on the
 *     stream they are given.  Products on ONE handle must not overlap (two
 *     streams, or two host threads): merge-path / balanced handles keep their
 *     tiles' carries in the handle, "hot_columns" handles a permuted copy of x
 *     and partial results.  Different handles may run side by side.
 *   - A handle belongs to the device that was current when it was created
 *     (spmv_hip_init); every entry point that takes a handle makes that device
 *     current for the call and restores the caller's device before it returns.
 */
#ifndef SPMV_HIP_H
#define SPMV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference src/spmv.h:26 -- AoS COO entry, 0-based */
#ifndef SPMV_HIP_NO_MTX_STRUCT
struct mtx { int i; int j; double a; };
#endif

typedef struct spmv_hip_matrix spmv_hip_matrix;

enum { SPMV_HIP_F64 = 0, SPMV_HIP_F32 = 1 };

/* storage format + kernel of a handle */
enum {
    SPMV_HIP_CSR_SCALAR = 0,  /* one lane per row                    (mult_csr_base,  csr.c:4-13)   */
    SPMV_HIP_CSR_VECTOR = 1,  /* 2..64 lanes per row, wave reduction (mult_csr,       csr.c:124-150) */
    SPMV_HIP_CSR_MERGE  = 2,  /* merge-path tiles, LDS-staged        (spmvmCRSMerge,  csr_merge.c:15-59) */
    SPMV_HIP_CSR_BAL    = 3,  /* nnz-balanced tiles, LDS-staged      (mult_csr_bal,   csr_bal.c:3-38) */
    SPMV_HIP_SELL       = 4,  /* SELL-C-sigma                        (mult_sell,      sell.c:35-42)  */
    SPMV_HIP_SELLP      = 5,  /* sliced ELL, natural row order       (mult_sellp,     sellp.c:304-342) */
    SPMV_HIP_ELL        = 6,  /* sliced ELL, natural row order       (mult_ell,       ell.c:28-45)   */
    SPMV_HIP_SELL_DFC   = 7   /* SELL-C-sigma without padding: active lanes per column (csr_to_sellcs_dfc, sellcs_format.c:53-136) */
};

/* arrays that spmv_hip_download() can read back (layout parity tests) */
enum {
    SPMV_HIP_ARR_ROW_PTR    = 0,  /* int32  [rows+1]            CSR family                       */
    SPMV_HIP_ARR_COL        = 1,  /* int32  [stored]            CSR col / sliced column_indices  */
    SPMV_HIP_ARR_VAL        = 2,  /* dtype  [stored]            CSR val / sliced values          */
    SPMV_HIP_ARR_SLICE_PTR  = 3,  /* int64  [nslices+1]         sliced formats (slice_pointers / block_ptr) */
    SPMV_HIP_ARR_ROW_ORDER  = 4,  /* int32  [rows]              SELL only (row_order)            */
    SPMV_HIP_ARR_TILE_COORD = 5,  /* int32  [2*(ntiles+1)]      merge: (row,nnz) start of each tile; bal: (compact row, nnz) */
    SPMV_HIP_ARR_VOP_PTR    = 6,  /* int64  [nslices+1]         SELL_DFC: first column of each slice (vop_pointers) */
    SPMV_HIP_ARR_VOP_LEN    = 7   /* uint8  [vop_ptr[nslices]]  SELL_DFC: active lanes - 1 of every column (vop_lengths) */
};

typedef struct spmv_hip_info {
    int     format, dtype;
    int     rows, columns;
    int64_t nnz;          /* true non-zeros (what GFLOP/s counts, driver_spmv.c:307) */
    int64_t stored;       /* entries held in HBM incl. padding                        */
    int     slice_height; /* C (SELL) / block height (SELLP, ELL); 0 otherwise        */
    int64_t sigma;        /* SELL sorting window; 0 otherwise                         */
    int     nslices;      /* sliced formats; 0 otherwise                              */
    int     lanes_per_row;/* CSR_VECTOR group size; 1 for CSR_SCALAR; 0 otherwise     */
    int     ntiles;       /* CSR_MERGE / CSR_BAL workgroup tiles; 0 otherwise         */
    int     items_per_tile;
    int64_t device_bytes; /* HBM held by the handle                                   */
    int     variant;      /* kernel variant in use (spmv_hip_set_option / spmv_hip_analyze) */
    int     hot_columns;  /* option "hot_columns": 0 off, 1 columns relabeled by reference count, 2 dealt over the XCDs */
    int64_t diag_columns; /* sliced formats of height 64: slice columns (64 entries each) whose entries all sit at
                             one distance from the diagonal; the variants >= 16 do not read their indices */
    int     pattern_blocks; /* CSR-based handles: blocks of 64 rows that repeat one pattern of distances from the
                             diagonal; CSR_SCALAR variants >= 16 read one row's indices for each, and the products
                             with a block of vectors walk them by positions */
    int     hot_variant;  /* with hot_columns 2: tile kernel of the copy split by column group (option "hot_variant") */
    int     block_map;    /* option "block_map": how workgroups are dealt over the matrix (0 an eighth per XCD, K runs of K) */
} spmv_hip_info;

/* Selects the device for subsequent creators on this thread (hipSetDevice). */
int spmv_hip_init(int device);
/* Free and total bytes of HBM on the current device (hipMemGetInfo).  The reference's
 * driver sizes a matrix against the host's MemAvailable before loading it
 * (enought_memory, driver_spmv.c:50-110); the C driver here asks the device as well. */
int spmv_hip_mem_info(size_t *free_bytes, size_t *total_bytes);
/* Message of the last failing call on this thread ("" if none). */
const char *spmv_hip_last_error(void);

/* ---- host-side ingest (reference mtx.c) ---------------------------------------
 * load_mtx (mtx.c:36-137): MatrixMarket coordinate real/integer/pattern,
 * general or mirrored symmetry, 1-based -> 0-based, explicit zeros dropped.
 * Returns non-zero (with a message) where the reference prints and exit(1)s.
 * *coo is malloc'ed; release it with spmv_hip_free_coo. */
int spmv_hip_load_mtx(const char *fname, int *rows, int *columns, int *nnz, struct mtx **coo);
void spmv_hip_free_coo(struct mtx *coo);
/* load_bin / save_bin (mtx.c:139-205, 207-247): int rows, int columns, long nnz,
 * then nnz x {int i, int j, double a} (0-based).  Files written by either side
 * are read by the other.  load_bin also range-checks the coordinates. */
int spmv_hip_load_bin(const char *fname, int *rows, int *columns, int *nnz, struct mtx **coo);
int spmv_hip_save_bin(const char *fname, int rows, int columns, int nnz, const struct mtx *coo);
/* sort_mtx(nnz, coo, by_row_mtx) (mtx.c:9-34): (row, col) order, in place,
 * untouched if already sorted. */
void spmv_hip_sort_coo(int nnz, struct mtx *coo);

/* ---- creators from a host COO (reference spmv.h:50,98,118,135) --------------
 * Like the reference creators these sort the caller's coo IN PLACE by
 * (row, col) (mtx.c:18-34) -- the driver's COO check (driver_spmv.c:274)
 * depends on that side effect -- and do not keep a pointer to it. */

/* create_csr (spmv.h:50, csr.c:160-203): `alignment` > 1 pads every row with
 * (j=0, A=0) to a multiple of it.  `kernel` is SPMV_HIP_CSR_SCALAR or
 * SPMV_HIP_CSR_VECTOR; `lanes_per_row` 0 = choose from the mean row length. */
spmv_hip_matrix *spmv_hip_create_csr(int rows, int columns, int nnz, int alignment,
                                     struct mtx *coo, int dtype, int kernel, int lanes_per_row);
/* create_csr_merge (spmv.h:98, csr_merge.c:87-115).  The reference freezes
 * nthreads at create time (csr_merge.c:93); here the tile size plays that
 * role: items_per_tile 0 = default (512; 1024 for fp32 matrices of 2^25 entries
 * and more), else one of 256, 512, 1024, 2048. */
spmv_hip_matrix *spmv_hip_create_csr_merge(int rows, int columns, int nnz, struct mtx *coo,
                                           int dtype, int items_per_tile);
/* create_csr_bal (spmv.h:118, csr_bal.c:66-94): equal-nnz chunks, one per
 * workgroup; nnz_per_tile 0 = default.  Unlike the reference (csr_bal.c:29)
 * matrices with empty rows are handled. */
spmv_hip_matrix *spmv_hip_create_csr_bal(int rows, int columns, int nnz, struct mtx *coo,
                                         int dtype, int nnz_per_tile);
/* create_ell (spmv.h:135, ell.c:48-112): sliced ELL in natural row order.
 * slice_height 0 = 64 (one wave64 per slice); 4 reproduces the reference's
 * SIMD_WIDTH (ell.c:4).  Column indices are 32-bit in HBM (the reference's
 * int64 j, spmv.h:131, is a host-side choice; values are identical). */
spmv_hip_matrix *spmv_hip_create_ell(int rows, int columns, int nnz, struct mtx *coo,
                                     int dtype, int slice_height);

/* ---- creators from host CSR arrays ----------------------------------------- */

/* create_sellp (sellp.h:15, sellp.c:352-458): slice_height 0 = 64; the
 * reference's _BLOCK is 2/4/8 (spmv.h:8-23). */
spmv_hip_matrix *spmv_hip_create_sellp(int rows, int columns, int nnz, const int *row_ptr,
                                       const int *col, const double *A, int dtype, int slice_height);
/* create_sell (spmv.h:233, sell.c:8-33) with the struct csr flattened.
 * C 0 = 64 (sell.c:5); sigma 0 = 16384 (sell.c:4).  sigma must be a multiple
 * of C (otherwise the reference's "first row of the slice is the longest"
 * rule, sellcs_format.c:152-158, does not hold); windows of up to 32768 rows
 * are sorted in LDS, longer ones (up to one window over the whole matrix) by
 * a radix sort on (window, length) -- the same order either way. */
spmv_hip_matrix *spmv_hip_create_sell(int rows, int columns, int nnz, const int *row_ptr,
                                      const int *col, const double *A, int dtype, int C, int64_t sigma);

/* SELL-C-sigma without padding (the reference builds this layout with
 * enable_dfc = 1, sellcs-spmv.h:86-98, sellcs_format.c:53-136, but ships no CPU
 * kernel for it): rows sorted as for SELL; column v of a slice stores only its
 * active lanes, i.e. the rows of the slice longer than v, which form a prefix
 * of the slice; `stored` == nnz.  The product keeps mult_csr_base's summation
 * order.  C must be 64 (one wave per slice); sigma as for SELL. */
spmv_hip_matrix *spmv_hip_create_sell_dfc(int rows, int columns, int nnz, const int *row_ptr,
                                          const int *col, const double *A, int dtype, int C, int64_t sigma);

/* ---- creator from CSR arrays already resident in HBM ------------------------
 * For matrices generated on the device.  d_val is float or double per `dtype`.
 * The CSR arrays are adopted by reference for the CSR family (the caller keeps
 * them alive until spmv_hip_free) and only read for the sliced formats.
 * p0/p1: CSR_VECTOR lanes_per_row / -; CSR_MERGE items_per_tile / -; CSR_BAL
 * nnz_per_tile / -; SELL C / sigma; SELLP, ELL slice_height / -.  0 = default.
 * No reference counterpart (the reference has no device). */
spmv_hip_matrix *spmv_hip_create_from_device_csr(int format, int rows, int columns, int64_t nnz,
                                                 const int *d_row_ptr, const int *d_col, const void *d_val,
                                                 int dtype, int64_t p0, int64_t p1, void *stream);

/* ---- automatic format choice -------------------------------------------------
 * The reference leaves the format to the user (MODE string, spmv.run:20,
 * driver_spmv.c:186-261) and tunes only inside SELL-C-sigma
 * (sellcs_analyze_matrix, sellcs_analyzer.c:46-120).  These creators extend
 * that trial to the formats themselves: merge-path CSR, lane-per-row CSR,
 * vector CSR, SELLP, SELL-64-1024 and SELL-64-16384 (the window of sell.c:5)
 * are each built from the same CSR (the
 * sliced ones only if they store at most 2x the non-zeros), tuned as by
 * spmv_hip_analyze with `reps` timed products per kernel variant, and the
 * fastest handle is returned (spmv_hip_get_info tells which).  create_auto
 * sorts the caller's coo like the other creators; the device variant adopts the
 * caller's CSR arrays when a CSR-family format wins. */
spmv_hip_matrix *spmv_hip_create_auto(int rows, int columns, int nnz, struct mtx *coo, int dtype, int reps);
spmv_hip_matrix *spmv_hip_create_auto_from_device_csr(int rows, int columns, int64_t nnz, const int *d_row_ptr,
                                                      const int *d_col, const void *d_val, int dtype, int reps,
                                                      void *stream);

/* ---- multiply --------------------------------------------------------------- */

/* mult_X(matrix, x, y) (driver_spmv.c:184,276): host vectors, x[columns],
 * y[rows]; copies x up, runs the kernel, copies y back.  Drop-in, PCIe-bound. */
int spmv_hip_mult(spmv_hip_matrix *m, const double *x, double *y);
/* Same product with x and y resident in HBM (dtype elements) on `stream`
 * (a hipStream_t; NULL = default stream).  Asynchronous. */
int spmv_hip_mult_dev(spmv_hip_matrix *m, const void *d_x, void *d_y, void *stream);

/* mult_mv_csr(struct csr *, int n, double *x, double *y) (spmv.h:54; declared
 * there, defined nowhere in the reference): y_v = A * x_v for v = 0..n-1 with
 * the matrix read once per block of up to 8 vectors.  Host form: vector v is
 * x + v*columns, its result y + v*rows.  Works on a handle of any format;
 * each result column is bit-identical to mult_csr_base (csr.c:4-13) of x_v. */
int spmv_hip_mult_mv(spmv_hip_matrix *m, int n, const double *x, double *y);
/* The same with the vector blocks resident in HBM (dtype elements): vector v
 * at d_x + v*ldx (ldx >= columns), result at d_y + v*ldy (ldy >= rows).
 * Asynchronous on `stream`. */
int spmv_hip_mult_mv_dev(spmv_hip_matrix *m, int n, const void *d_x, int64_t ldx, void *d_y, int64_t ldy,
                         void *stream);

/* Row map for a CSR_SCALAR / CSR_VECTOR handle whose rows are a compacted
 * subset of a larger vector: row k writes y[d_row_map[k]] (device array of
 * `rows` ints, kept alive by the caller; NULL restores y[k]).  Used by the
 * row-block multi-GPU layer for the rows that touch halo columns.  No reference
 * counterpart (the reference has no partitioned product). */
int spmv_hip_set_row_map(spmv_hip_matrix *m, const int *d_row_map);

/* ---- row-block partitioned product over several GPUs ----------------------------
 * The reference has no partitioned product; its one precedent is the equal-nnz
 * contiguous chunking of create_csr_bal (csr_bal.c:83-85), applied here to whole
 * rows.  A block of a square matrix owns rows [row_begin, row_begin + rows) and the
 * same range of x and y.  Its rows whose columns all lie in that range ("interior")
 * are multiplied by a matrix in the caller's format against the local part of x;
 * the others ("boundary", a compact CSR with a row map: lane-per-row beside the
 * formats that keep mult_csr_base's summation order, vector CSR otherwise) against
 * [x_local | halo], the halo being the x entries of the block's remote columns in
 * ascending global column order.  Every row is still summed whole by one kernel,
 * so with "exact_order" set on a lane-per-row or sliced format the partitioned
 * product has the bits of mult_csr_base (csr.c:4-13). */
typedef struct spmv_hip_rowblock spmv_hip_rowblock;
typedef struct spmv_hip_dist spmv_hip_dist;

typedef struct spmv_hip_rowblock_info {
    int     format, dtype, device;
    int     rows, row_begin, columns_global;
    int64_t nnz, interior_nnz, boundary_nnz;
    int     boundary_rows;       /* rows with at least one remote column                      */
    int     halo;                /* distinct remote columns                                    */
    int     send;                /* entries of the send list (spmv_hip_rowblock_set_send_list) */
    int64_t interior_stored;     /* entries the interior matrix holds in HBM incl. padding     */
    int     interior_variant, boundary_lanes_per_row, boundary_variant;
    int64_t device_bytes;
} spmv_hip_rowblock_info;

/* bounds[0..parts]: contiguous row blocks with about nnz/parts non-zeros each --
 * block p ends at the first row whose row_ptr reaches (p+1)*nnz/parts (the share
 * rule of csr_bal.c:83-85 on rows) moved to the nearest multiple of `snap` rows
 * (1024 keeps SELL sorting windows and slices inside one block).  Host arrays. */
int spmv_hip_partition_rows(int rows, const int *row_ptr, int parts, int snap, int *bounds);
/* The same rule on 64-bit row pointers, for matrices whose non-zeros outgrow the
 * reference's `int` counts (spmv.h:41-47) and exist only as per-GPU shards. */
int spmv_hip_partition_rows64(int rows, const int64_t *row_ptr, int parts, int snap, int *bounds);

/* Builds a block on the current device from its CSR in HBM (row pointers local to
 * the block, GLOBAL column indices, dtype values); the arrays are only read.
 * format / p0 / p1 as for spmv_hip_create_from_device_csr (the interior matrix). */
spmv_hip_rowblock *spmv_hip_rowblock_create(int format, int rows, int row_begin, int columns_global, int64_t nnz,
                                            const int *d_row_ptr, const int *d_col, const void *d_val, int dtype,
                                            int64_t p0, int64_t p1, void *stream);
int spmv_hip_rowblock_free(spmv_hip_rowblock *rb);
int spmv_hip_rowblock_get_info(const spmv_hip_rowblock *rb, spmv_hip_rowblock_info *info);
/* "exact_order" as for spmv_hip_set_option, applied to the interior and the
 * boundary matrix.  A new block runs each matrix's default kernel (no trial has run
 * yet), so setting the option before the first spmv_hip_rowblock_analyze pins the
 * bits of y whatever the trials measure later. */
int spmv_hip_rowblock_set_option(spmv_hip_rowblock *rb, const char *name, int64_t value);
/* spmv_hip_analyze on the interior and on the boundary matrix (sellcs_analyze_matrix,
 * sell.c:30, for both pieces of the block) */
int spmv_hip_rowblock_analyze(spmv_hip_rowblock *rb, int reps);
/* columns[0..halo): the remote columns, ascending (host array) */
int spmv_hip_rowblock_halo_columns(const spmv_hip_rowblock *rb, int *columns);
/* The local x entries (indices into x_local) other blocks need from this one, in
 * the order they are sent; spmv_hip_rowblock_pack gathers them:
 * d_send[k] = d_x[local_index[k]]. */
int spmv_hip_rowblock_set_send_list(spmv_hip_rowblock *rb, int count, const int *local_index);
int spmv_hip_rowblock_pack(spmv_hip_rowblock *rb, const void *d_x, void *d_send, void *stream);
/* The same gather delivered in one launch: the send list is cut into `nseg` (at most
 * 16) consecutive segments, segment s = entries seg_begin[s] .. seg_begin[s+1]-1
 * (seg_begin[0] = 0, seg_begin[nseg] = the list's length), and entry k of segment s
 * is stored to dst[s][k - seg_begin[s]].  dst[s] may be memory of another GPU that
 * this block's device has peer access to: the halo then crosses xGMI as the
 * kernel's own stores, without a copy per consumer. */
int spmv_hip_rowblock_push(spmv_hip_rowblock *rb, const void *d_x, int nseg, const int *seg_begin, void *const *dst,
                           void *stream);
/* y[0..rows) = interior rows * d_x[0..rows); boundary rows receive 0 here and their
 * value from mult_boundary, which must follow on the same stream once
 * d_x_ext[rows..rows+halo) holds the halo. */
int spmv_hip_rowblock_mult_interior(spmv_hip_rowblock *rb, const void *d_x, void *d_y, void *stream);
int spmv_hip_rowblock_mult_boundary(spmv_hip_rowblock *rb, const void *d_x_ext, void *d_y, void *stream);

/* All blocks of one process behind one handle, called from ONE host thread
 * (reference threading model: the driver is single-threaded,
 * driver_spmv.c:112-347): the matrix is cut with spmv_hip_partition_rows into
 * `ngpus` blocks, block p built on devices[p] (NULL = devices 0..ngpus-1; a device
 * may be named more than once), each with a compute and a communication stream.
 * One product gathers the halo entries on every block and stores them straight into
 * their consumers' vectors over xGMI (one kernel per block on its communication
 * stream: spmv_hip_rowblock_push; pack + one peer-to-peer copy per consumer where a
 * pair of GPUs has no peer access, or with SPMV_DIST_DIRECT_PUSH=0) while the
 * interior products run, then multiplies the boundary rows.  So that the launches
 * of the blocks are issued side by side and not one block after the other, the
 * handle keeps one helper thread per block, hidden from the caller and asleep
 * between calls (SPMV_DIST_THREADS=0 in the environment issues everything from the
 * calling thread instead).  Every block starts on its default kernels:
 * spmv_hip_dist_set_option("exact_order"), then spmv_hip_dist_analyze tune them.
 * A pair of GPUs without peer access makes the runtime stage that pair's pushes
 * through host memory: creation warns on stderr, spmv_hip_dist_get_peer_access
 * reports it, and SPMV_REQUIRE_P2P=1 in the environment makes creation fail
 * instead.  Sorts the caller's coo in place like the other creators. */
spmv_hip_dist *spmv_hip_dist_create(int format, int rows, int columns, int nnz, struct mtx *coo, int dtype, int64_t p0,
                                    int64_t p1, int ngpus, const int *devices);
/* The same handle over blocks the caller has built with spmv_hip_rowblock_create,
 * each on the device it is to run on -- for matrices that never exist as one host
 * COO: with more than 2^31 - 1 non-zeros in all (the reference's `int nnz`,
 * spmv.h:26,41-47, cannot count them; every block still holds fewer), or generated
 * shard by shard in HBM.  blocks[0..ngpus) must tile the square matrix in order
 * (block p begins at the row where block p-1 ends; the last ends at
 * columns_global) and share one dtype.  On success the handle owns the blocks
 * (spmv_hip_dist_free frees them); on failure they stay the caller's. */
spmv_hip_dist *spmv_hip_dist_create_from_blocks(int ngpus, spmv_hip_rowblock **blocks);
/* mult_X with host vectors (driver_spmv.c:276) */
int spmv_hip_dist_mult(spmv_hip_dist *d, const double *x, double *y);
/* the timed power loop (driver_spmv.c:292-308) over all blocks: exchange + products
 * timed on the host clock from the first launch to the last block's completion,
 * x = y / ||y||_2 untimed (every block reduces its part of y on its own GPU; the
 * partial norms meet in pinned host memory and are added in block order) */
int spmv_hip_dist_bench(spmv_hip_dist *d, int min_iters, double min_seconds, double *x, int *iters, double *seconds);
/* Measurement aid for the layer itself: `products` products on whatever x the blocks
 * hold; *issue_seconds = host time until all launches of a product are issued (the
 * part that one thread per GPU shortens), *total_seconds = until the last block has
 * finished, both summed over the products.  No reference counterpart. */
int spmv_hip_dist_issue_time(spmv_hip_dist *d, int products, double *issue_seconds, double *total_seconds);
/* spmv_hip_rowblock_set_option on every block */
int spmv_hip_dist_set_option(spmv_hip_dist *d, const char *name, int64_t value);
/* spmv_hip_rowblock_analyze on every block (the blocks' trials run side by side) */
int spmv_hip_dist_analyze(spmv_hip_dist *d, int reps);
/* access[q * ngpus + p]: how the halo entries block q sends to block p travel --
 * 2 = both blocks sit on one device, 1 = direct peer-to-peer write (peer access from
 * q's device to p's is enabled), 0 = staged through host memory by the runtime. */
int spmv_hip_dist_get_peer_access(const spmv_hip_dist *d, int *access);
/* ngpus, bounds[0..ngpus], blocks[0..ngpus) -- any of them may be NULL */
int spmv_hip_dist_get_info(const spmv_hip_dist *d, int *ngpus, int *bounds, spmv_hip_rowblock_info *blocks);
int spmv_hip_dist_free(spmv_hip_dist *d);

/* free_X (csr.c:205-210, sellp.c:461-465; the reference leaks the others) */
int spmv_hip_free(spmv_hip_matrix *m);

/* ---- measurement (driver_spmv.c:292-308) ------------------------------------
 * Power-iteration loop with x, y resident: each iteration times one multiply
 * with HIP events, then x = y / ||y||_2 untimed, until iterations >= min_iters
 * and wall time > min_seconds.  x (host, columns doubles) is the start vector
 * and receives the last iterate.  rows must equal columns. */
int spmv_hip_bench(spmv_hip_matrix *m, int min_iters, double min_seconds, double *x,
                   int *iters, double *seconds);

/* ---- vector helpers on device (driver_spmv.c:21-26, 301-302) ---------------- */
/* *d_out (double, device) = ||d_x||_2 ; deterministic two-pass reduction on the
 * current device.  The partial sums live in library scratch kept per (device,
 * stream): calls on different streams may overlap; the first call on a
 * (device, stream) pair allocates its 8 KiB, kept until spmv_hip_release_stream
 * or the end of the process. */
int spmv_hip_norm2_dev(int n, const void *d_x, int dtype, double *d_out, void *stream);
/* No counterpart in the reference (it has no streams): gives back what the library
 * holds on the current device for `stream` -- today the partial sums of
 * spmv_hip_norm2_dev.  Call it before hipStreamDestroy in a program that creates
 * and destroys streams; it waits for the stream's work first.  A stream the
 * library never saw is not an error. */
int spmv_hip_release_stream(void *stream);
/* spmv_hip_norm2_dev for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same kernels and the same bits as spmv_hip_norm2_dev, but the 1024
 * partial sums live in an 8 KiB array that is part of the library's code object
 * (one copy per GPU, held until the process ends), so the call allocates nothing.
 * n is limited to 2^31 - 1 elements; a larger n is refused with an error.  The
 * array's address is looked up once per device, by spmv_hip_init(device) or else
 * by the first call on that device (call spmv_hip_init before capturing a graph).
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two streams, two threads) race; calls ordered on one stream do not. */
int spmv_hip_norm2_dev_noalloc(int64_t n, const void *d_x, int dtype, double *d_out, void *stream);
/* d_x[i] = d_y[i] / *d_norm */
int spmv_hip_scale_dev(int n, const void *d_y, void *d_x, int dtype, const double *d_norm, void *stream);

/* ---- tuning (counterpart of sellcs_analyze_matrix, sellcs_analyzer.c:46-120:
 * pick the launch shape by trial).  Options:
 *  "variant" = kernel variant of the handle's format (0 = default);
 *  "hot_columns" = 1: CSR-based handles (scalar, vector, merge, balanced) keep a
 *    second column array whose labels number the columns in descending order
 *    of their reference count, and every product first permutes x to match, so
 *    that the few columns a power-law matrix references most sit in one
 *    cache-resident range of x.  Entry order inside the rows is untouched: y is
 *    bit-identical to the product without the option.  Costs 4 bytes per stored
 *    entry plus (4 + V) bytes per column of HBM and one pass over x per
 *    product; products on one handle must then not overlap (they share the
 *    permuted x).  0 drops the arrays again.  spmv_hip_mult_mv ignores it.
 *    = 2: the ranking is dealt over the 8 XCDs instead, and the handle keeps a
 *    second copy of the matrix split by column group (one group per XCD, so
 *    the eight L2s hold eight different sets of hot columns) plus one partial
 *    result per row and group with entries, which a last pass adds in group
 *    order: y then differs from the plain product by reassociation (tolerance
 *    parity).  Costs (4 + V) bytes per entry plus about 33 bytes per row and
 *    (8 + V) bytes per row and group with entries of HBM; pays once x
 *    is far larger than the eight L2s together (R-MAT scale 26: 20.1 ms
 *    plain, 12.0 ms with 1, 7.4 ms with 2).
 *  "hot_variant" = tile kernel of that second copy (needs "hot_columns" = 2;
 *    spmv_hip_analyze picks it by trial): 0, 1 = the balanced-CSR kernels, one
 *    workgroup per tile; 3, 4 = persistent workgroups of 8 / 16 waves, one per
 *    CU, that keep the first 15360 / 10240 labels (fp64; fp32: 34816 / 28672)
 *    of their column group's share of x in LDS -- a gather served there saves
 *    the 128-byte line an L2 hit moves into the L1.  1, 3 and 4 give the same
 *    bits.
 *  "block_map" = which piece of the matrix a workgroup takes (spmv_hip_analyze
 *    picks it by trial after the kernel variant, from 0, 1, 2, 3, 4, 5, 8; the bits of
 *    y do not depend on it).  The hardware deals consecutive workgroups over the
 *    8 XCDs in turn.  0 (default): every XCD walks one contiguous eighth of the
 *    matrix, so each private L2 sees its own window of x and every line of x
 *    leaves HBM once.  1: consecutive workgroups walk consecutive pieces, so all
 *    XCDs move through the matrix side by side.  K > 1 (up to 2^20): runs of K
 *    consecutive pieces per XCD, dealt in turn; x is then fetched by several L2s
 *    (27-point stencil 252^3: 4.19 GB moved instead of 3.90 with 1), but the
 *    eight XCDs then work side by side in one region of the matrix and of y --
 *    SELL-64-1024 fp64 on that matrix 0.803 ms with 0, 0.794 with 1, 0.761 with
 *    5.  The column-split
 *    copy behind "hot_columns" = 2 always keeps 0 (an XCD has to stay with its
 *    column group).  spmv_hip_mult_mv with two or more vectors on a window-sorted
 *    handle (SELL-C-sigma with or without padding) uses the value only if it is
 *    an odd K >= 3, and 5 otherwise: the short rows of every window of 1024
 *    rows sit in every 4th workgroup, and any other dealing hands those
 *    workgroups to one shader engine of each XCD.
 *  "exact_order" = 1: spmv_hip_analyze may only pick kernel variants that leave
 *    the bits of y as they are under the current setting, so that tuning noise
 *    cannot change results from run to run: for the sliced formats the variants
 *    that keep mult_csr_base's summation order (0, 16..31; a handle currently
 *    on a reassociating variant 1..12 is put back on 0), for vector CSR the
 *    current lanes per row and kernel family, for merge-path and balanced tiles
 *    variant 0 or variants 1 and 2 (they differ where a tile lies inside one
 *    row); lane-per-row CSR and padding-free SELL give the same bits under every
 *    variant.  0 (default) lifts the restriction. */
int spmv_hip_set_option(spmv_hip_matrix *m, const char *name, int64_t value);
/* sellcs_analyze_matrix (sell.c:30, sellcs_analyzer.c:46-120) re-thought for the
 * GPU: times every kernel variant of the handle's format `reps` times on a
 * zero vector and keeps the fastest.  Variants 0 and 16..31 of the sliced
 * formats keep mult_csr_base's summation order (bit-identical y); 1..12 (16-byte
 * loads, two / four partial sums per row) reassociate.  24..27 (fp64) are 16-byte
 * loads in the exact order: a v_permlane32_swap hands each row's entries back to
 * one lane (27-point stencil 252^3, SELLP with every index read: 1.02 -> 0.92 ms,
 * same bits).  All CSR_SCALAR variants (LDS chunk sizes) give the same
 * bits.  For CSR_VECTOR the trial covers lanes per row x rows walked side by
 * side per lane group, and the LDS-staged kernel for short rows (variants 5..8).
 * Where the matrix has regular structure the trial includes the variants that
 * do not read column indices they can compute -- slice columns at one distance
 * from the diagonal (sliced formats 16..23 and, with 16-byte value loads, fp64 28..31; padding-free SELL 9..12), blocks of 64 rows or whole tiles
 * that repeat one row pattern (CSR_SCALAR 16..22, CSR_VECTOR 7..8, merge-path
 * and balanced tiles 2); each gives the bits of its index-reading twin. */
int spmv_hip_analyze(spmv_hip_matrix *m, int reps);

/* ---- introspection for layout-parity tests ---------------------------------- */
int spmv_hip_get_info(const spmv_hip_matrix *m, spmv_hip_info *info);
/* Copies one device array to host; `bytes` must equal the array's size. */
int spmv_hip_download(const spmv_hip_matrix *m, int which, void *host, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HIP_H */
