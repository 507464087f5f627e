// Internal declarations shared by the translation units of libspmv_hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "spmv_hip.h"

namespace spmv {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kBlock = 256;          // threads per workgroup for every SpMV kernel
constexpr int kDefaultSigma = 16384; // sell.c:4
constexpr int kDefaultC = 64;        // sell.c:5 (max_vlen), = one wave64
constexpr int64_t kMaxSigma = 32768; // longest sorting window the LDS sort takes; longer ones go through a radix sort

void set_error(const char *fmt, ...);
void clear_error();

#define SPMV_HIP_TRY(expr, onfail)                                                        \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            spmv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),        \
                            __FILE__, __LINE__);                                          \
            onfail;                                                                       \
        }                                                                                 \
    } while (0)

inline size_t dtype_size(int dtype) { return dtype == SPMV_HIP_F32 ? 4 : 8; }

// Makes `device` current for the scope of an entry point and restores the caller's device after:
// a handle's arrays live on the device it was created on, whatever device the calling thread has
// selected since.
struct DeviceGuard {
    int prev = -1;
    bool switched = false, ok = true;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; set_error("no HIP device available"); return; }
        if (prev != device) {
            ok = hipSetDevice(device) == hipSuccess;
            switched = ok;
            if (!ok) set_error("hipSetDevice(%d) failed", device);
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

} // namespace spmv

// The opaque handle of the C ABI.
struct spmv_hip_matrix {
    int format = 0, dtype = 0, device = 0;
    int cus = 0;                   // compute units of the device (read once, when the handle is made)
    int max_lds = 0;               // bytes of LDS one workgroup may use there
    int rows = 0, columns = 0;
    int64_t nnz = 0;      // true non-zeros
    int64_t stored = 0;   // entries in HBM incl. padding

    // CSR family.  owns_csr == false when the arrays were adopted from the caller.
    bool owns_csr = true;
    int *d_row_ptr = nullptr;
    int *d_col = nullptr;
    void *d_val = nullptr;
    int lanes_per_row = 0;
    const int *d_row_map = nullptr; // CSR_SCALAR / CSR_VECTOR: matrix row k writes y[row_map[k]] (caller's array)
    unsigned char *d_block_pat = nullptr; // CSR-based formats: per block of 64 rows, the common row length when all its
                                          // rows repeat one pattern of distances from the diagonal, else 0
    int pat_blocks = 0;            // how many blocks do
    int variant = 0;               // kernel variant inside the format (spmv_hip_set_option "variant")
    int block_map = 0;             // 0: every XCD walks one contiguous eighth of the matrix; 1: consecutive workgroups
                                   // walk consecutive pieces (spmv_hip_set_option "block_map")
    bool exact_order = false;      // spmv_hip_analyze may only pick variants that leave y's bits as they are
    double max_padding = 0;        // > 0: building a sliced layout fails when stored > max_padding * nnz

    // merge-path / balanced tiles
    int ntiles = 0, items_per_tile = 0;
    int *d_tile_coord = nullptr;   // 2*(ntiles+1) ints
    int *d_row_carry = nullptr;    // ntiles
    void *d_val_carry = nullptr;   // ntiles (dtype)
    unsigned char *d_tile_pat = nullptr; // ntiles: the common row length when all rows of the tile repeat one pattern
    int pat_tiles = 0;             // how many tiles do
    // balanced CSR: compacted non-empty rows
    int nz_rows = 0;               // number of non-empty rows
    int *d_nzr_id = nullptr;       // [nz_rows] original row index
    int *d_nzr_end = nullptr;      // [nz_rows] row end offsets (row_ptr[row+1])
    bool has_empty_rows = false;
    bool packed_rows = false;       // balanced CSR: y[k] belongs to the k-th non-empty row (the caller unpacks)
    bool leave_empty_rows = false;  // balanced CSR: y of empty rows is left alone (the caller skips them)

    // sliced formats (SELL-C-sigma, SELLP, ELL)
    int slice_height = 0, nslices = 0;
    int64_t sigma = 0;
    int64_t *d_slice_ptr = nullptr; // nslices+1
    int *d_row_order = nullptr;     // rows (SELL only)
    int *d_scol = nullptr;          // stored
    void *d_sval = nullptr;         // stored (dtype)
    int *d_sdiag = nullptr;         // stored/64 (height 64 only): per slice column, col - row where all its entries
                                    // share it, INT_MIN otherwise
    int64_t diag_columns = 0;       // how many slice columns have one
    // SELL without padding: length of the row at every sorted position, columns per slice, active lanes
    int *d_sorted_len = nullptr;    // rows
    int64_t *d_vop_ptr = nullptr;   // nslices+1
    uint8_t *d_vop_len = nullptr;   // vops
    int64_t vops = 0;

    // columns relabeled by reference count (option "hot_columns", CSR-based formats): the col array with
    // the new labels, the column of x behind each label, and the permuted copy of x the kernels read
    int *d_hot_col = nullptr;
    int *d_hot_old = nullptr;
    void *d_hot_x = nullptr;
    // "hot_columns" = 2: the columns dealt over the 8 XCDs.  xcd_sub is a balanced-CSR handle over the
    // stacked matrix of 8*rows virtual rows and 8*xcd_cpx labels; d_hot_old / d_hot_x then have one
    // entry per label, d_xcd_y holds the 8 partial results of every row
    spmv_hip_matrix *xcd_sub = nullptr;
    // on the stacked handle itself: labels per column group (0 = not a stacked matrix) and the first
    // tile of each group's piece ([8] = ntiles), for the tile kernels that keep hot labels in LDS
    int group_cpx = 0;
    int group_first_tile[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    void *d_xcd_y = nullptr;
    unsigned char *d_xcd_presence = nullptr;        // per row: bit g set when column group g has entries in the row
    int *d_xcd_base = nullptr;                      // [8 * ceil(rows / 64)] slot of each 64-row block's first partial result per group
    int xcd_cpx = 0;
    int xcd_used_cpx = 0;           // labels per group that the matrix references (the rest of a group never needs x)
    int64_t xcd_bytes = 0;          // what the stacked handle and its vectors added to device_bytes

    // staging vectors for the host-pointer multiply and the bench loop
    void *d_x = nullptr, *d_y = nullptr;
    double *d_norm = nullptr;

    int64_t device_bytes = 0;
};

namespace spmv {

// kernels.hip -- asynchronous launches on `stream`
int launch_mult(const spmv_hip_matrix *m, const void *d_x, void *d_y, hipStream_t stream);
// multivec.hip -- y_v = A x_v for n vectors spaced ldx / ldy elements apart
int launch_mult_mv(const spmv_hip_matrix *m, int n, const void *d_x, int64_t ldx, void *d_y, int64_t ldy,
                   hipStream_t stream);
int variant_candidates(const spmv_hip_matrix *m, int *out, int max_out);
bool candidate_keeps_bits(const spmv_hip_matrix *m, int candidate);
int launch_norm2(int n, const void *d_x, int dtype, double *d_out, hipStream_t stream);
constexpr int kNormBlocks = 1024;                    // partial sums of the two-pass norm
// the norm with the caller's own scratch of kNormBlocks doubles (handles that own streams keep one per stream)
int release_norm_scratch(hipStream_t st);
int launch_norm2_with(double *partial, int n, const void *d_x, int dtype, double *d_out, hipStream_t stream);
constexpr int64_t kNormStaticMax = 0x7fffffffLL;     // elements spmv_hip_norm2_dev_noalloc accepts
int static_scratch_setup();                          // current device's copy of the built-in scratch array
int launch_norm2_static(int64_t n, const void *d_x, int dtype, double *d_out, hipStream_t stream);
int launch_scale(int n, const void *d_y, void *d_x, int dtype, const double *d_norm, hipStream_t stream);
int launch_scale_value(int n, const void *d_y, void *d_x, int dtype, double norm, hipStream_t stream);

// api.hip -- a handle over CSR arrays already in HBM; own = the handle frees them (also when the
// creation fails), otherwise they stay the caller's
spmv_hip_matrix *adopt_device_csr(int format, int rows, int columns, int64_t nnz, int *d_row_ptr, int *d_col, void *d_val,
                                  int dtype, int64_t p0, int64_t p1, bool own, hipStream_t stream);

// convert.hip -- build device layouts from a CSR resident in HBM.
// `val_is_f64`: element type of d_val (host creators upload doubles; the device
// creator hands over dtype elements).
int choose_lanes_per_row(int rows, int64_t stored);
int launch_group_presence(int rows, const int *d_vrow_ptr, unsigned char *d_presence, hipStream_t stream);   // kernels.hip
int launch_group_block_count(int rows, const unsigned char *d_presence, int *d_count, hipStream_t stream);
int build_hot_columns(spmv_hip_matrix *m, hipStream_t stream);                            // sort.hip
int build_xcd_columns(const spmv_hip_matrix *m, bool val_is_f64, hipStream_t stream, int **vrow_ptr, int **vcol,
                      void **vval, int **old_of_label, int *cpx, int *used_cpx);                         // sort.hip
int device_exclusive_scan(const int *d_in, int *d_out, size_t n, hipStream_t stream);   // sort.hip (rocPRIM)
int device_coo_to_csr(int rows, int columns, int64_t nnz, struct mtx *coo, bool values_f64, int **d_row_ptr,
                      int **d_col, void **d_val);                                         // sort.hip
int build_block_patterns(spmv_hip_matrix *m, hipStream_t stream);
int build_merge_tiles(spmv_hip_matrix *m, hipStream_t stream);
int build_bal_tiles(spmv_hip_matrix *m, hipStream_t stream);
int build_sliced(spmv_hip_matrix *m, const int *d_row_ptr, const int *d_col, const void *d_val,
                 bool val_is_f64, bool sorted, hipStream_t stream);
int build_sell_dfc(spmv_hip_matrix *m, const int *d_row_ptr, const int *d_col, const void *d_val,
                   bool val_is_f64, hipStream_t stream);

template <typename T> int dev_alloc(spmv_hip_matrix *m, T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        return 1;
    }
    if (m) m->device_bytes += (int64_t)(count * sizeof(T));
    return 0;
}

} // namespace spmv
