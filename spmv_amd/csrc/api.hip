// C ABI of libspmv_hip: creators, multiply, free, bench loop, introspection.

#include "internal.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

namespace spmv {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}
void clear_error() { g_error[0] = 0; }

static void sort_coo_in_place(int64_t nnz, struct mtx *coo) { spmv_hip_sort_coo((int)nnz, coo); }

// COO (sorted) -> CSR with optional per-row padding to a multiple of
// `alignment` using (j=0, A=0) entries (reference csr.c:160-203).
static int coo_to_csr(int rows, int64_t nnz, int alignment, const struct mtx *coo, std::vector<int> &rp,
                      std::vector<int> &cj, std::vector<double> &va)
{
    if (alignment < 1) { set_error("alignment %d < 1", alignment); return 1; }
    rp.assign((size_t)rows + 1, 0);
    for (int64_t l = 0; l < nnz; l++) {
        if (coo[l].i < 0 || coo[l].i >= rows) { set_error("coo[%lld].i = %d out of range", (long long)l, coo[l].i); return 1; }
        rp[(size_t)coo[l].i + 1]++;
    }
    int64_t p = 0;
    for (int r = 0; r < rows; r++) {
        int64_t len = rp[(size_t)r + 1];
        int64_t padded = (len + alignment - 1) / alignment * alignment;
        // the reference pads the running total, which equals per-row padding
        // because every earlier row already ends on a multiple
        p += padded;
        if (p > 0x7fffffffLL) { set_error("padded nnz exceeds 2^31-1"); return 1; }
        rp[(size_t)r + 1] = (int)p;
    }
    cj.assign((size_t)p, 0);
    va.assign((size_t)p, 0.0);
    int64_t l = 0;
    for (int r = 0; r < rows; r++) {
        int64_t w = rp[r];
        while (l < nnz && coo[l].i == r) { cj[(size_t)w] = coo[l].j; va[(size_t)w] = coo[l].a; w++; l++; }
    }
    return 0;
}

static int check_dims(int rows, int columns, int64_t nnz, int dtype)
{
    if (rows < 0 || columns < 0 || nnz < 0) { set_error("negative dimension"); return 1; }
    if (dtype != SPMV_HIP_F64 && dtype != SPMV_HIP_F32) { set_error("unknown dtype %d", dtype); return 1; }
    // entry positions are 32-bit (as the reference's int nnz and row pointers); the chunked kernels add up to
    // 2048 to a position before clamping it into the row block's run
    if (nnz > 0x7fffffffLL - 4096) { set_error("%lld non-zeros exceed the 32-bit entry positions (limit 2^31 - 4096)", (long long)nnz); return 1; }
    return 0;
}

static int check_columns(int columns, const std::vector<int> &cj)
{
    for (size_t k = 0; k < cj.size(); k++)
        if (cj[k] < 0 || cj[k] >= std::max(columns, 1)) { set_error("column index %d out of range at entry %zu", cj[k], k); return 1; }
    return 0;
}

static spmv_hip_matrix *new_handle(int format, int dtype, int rows, int columns, int64_t nnz)
{
    spmv_hip_matrix *m = new (std::nothrow) spmv_hip_matrix();
    if (!m) { set_error("out of host memory"); return nullptr; }
    m->format = format; m->dtype = dtype; m->rows = rows; m->columns = columns; m->nnz = nnz;
    if (hipGetDevice(&m->device) != hipSuccess) { set_error("no HIP device available"); delete m; return nullptr; }
    if (hipDeviceGetAttribute(&m->cus, hipDeviceAttributeMultiprocessorCount, m->device) != hipSuccess) m->cus = 0;
    if (hipDeviceGetAttribute(&m->max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, m->device) != hipSuccess) m->max_lds = 0;
    (void)hipGetLastError();
    return m;
}

// Upload host CSR (double values) into the handle, converting to the handle's dtype.
static int upload_csr(spmv_hip_matrix *m, const std::vector<int> &rp, const std::vector<int> &cj,
                      const std::vector<double> &va, bool keep_f64_values)
{
    m->stored = (int64_t)cj.size();
    m->owns_csr = true;
    if (dev_alloc(m, &m->d_row_ptr, rp.size())) return 1;
    if (dev_alloc(m, &m->d_col, cj.size())) return 1;
    SPMV_HIP_TRY(hipMemcpy(m->d_row_ptr, rp.data(), sizeof(int) * rp.size(), hipMemcpyHostToDevice), return 1);
    if (!cj.empty()) SPMV_HIP_TRY(hipMemcpy(m->d_col, cj.data(), sizeof(int) * cj.size(), hipMemcpyHostToDevice), return 1);
    if (m->dtype == SPMV_HIP_F64 || keep_f64_values) {
        if (dev_alloc(m, (double **)&m->d_val, va.size())) return 1;
        if (!va.empty()) SPMV_HIP_TRY(hipMemcpy(m->d_val, va.data(), sizeof(double) * va.size(), hipMemcpyHostToDevice), return 1);
    } else {
        std::vector<float> vf(va.begin(), va.end());
        if (dev_alloc(m, (float **)&m->d_val, vf.size())) return 1;
        if (!vf.empty()) SPMV_HIP_TRY(hipMemcpy(m->d_val, vf.data(), sizeof(float) * vf.size(), hipMemcpyHostToDevice), return 1);
    }
    return 0;
}

static void release_csr(spmv_hip_matrix *m)
{
    if (m->owns_csr) {
        if (m->d_row_ptr) (void)hipFree(m->d_row_ptr);
        if (m->d_col) (void)hipFree(m->d_col);
        if (m->d_val) (void)hipFree(m->d_val);
    }
    m->d_row_ptr = nullptr; m->d_col = nullptr; m->d_val = nullptr;
}

static int normalize_tile_param(int64_t p, int dflt, int *out)
{
    int v = p == 0 ? dflt : (int)p;
    const int ok[] = {256, 512, 1024, 2048};     // 64 lanes x {4, 8, 16, 32} items
    for (int k : ok)
        if (v == k) { *out = v; return 0; }
    set_error("tile size %d must be one of 256, 512, 1024, 2048", v);
    return 1;
}

// Finish a handle whose CSR arrays are in HBM (d_row_ptr/d_col/d_val set,
// `csr_val_is_f64` tells the element type of d_val).
static int finish_from_device_csr(spmv_hip_matrix *m, int64_t p0, int64_t p1, bool csr_val_is_f64, hipStream_t st)
{
    // the CSR kernels step through the entry arrays in chunks of up to 2048 with 32-bit positions
    if (m->stored > 0x7fffffffLL - 4096) {
        set_error("%lld stored entries: 32-bit CSR positions need 4096 entries of headroom below 2^31", (long long)m->stored);
        return 1;
    }
    switch (m->format) {
    case SPMV_HIP_CSR_SCALAR: m->lanes_per_row = 1; return build_block_patterns(m, st);
    case SPMV_HIP_CSR_VECTOR: {
        int g = p0 ? (int)p0 : choose_lanes_per_row(m->rows, m->stored);
        if (g < 2 || g > 64 || (g & (g - 1))) { set_error("lanes_per_row %d must be a power of two in 2..64", g); return 1; }
        // left to the library, rows averaging fewer than 32 entries take the LDS-staged kernel (variant 5)
        if (!p0 && g <= 32) { g = std::min(g, 8); m->variant = 5; }
        m->lanes_per_row = g;
        return build_block_patterns(m, st);
    }
    // tiles left to the library: 512 items (one wave, 8 per lane); fp32 entries are a third shorter, and on
    // matrices large enough to still give every CU many tiles 1024 keep as many bytes in flight per wave
    // (stencil 252^3 fp32 +9 %, FEM stand-in +6 %; the 1000^2 Laplacian with its 5 M entries loses 12 %)
    case SPMV_HIP_CSR_MERGE:
        if (normalize_tile_param(p0, m->dtype == SPMV_HIP_F32 && m->stored >= (1 << 25) ? 1024 : 512, &m->items_per_tile)) return 1;
        if (build_merge_tiles(m, st)) return 1;
        return build_block_patterns(m, st);                  // for the products with a block of vectors
    case SPMV_HIP_CSR_BAL:
        if (normalize_tile_param(p0, m->dtype == SPMV_HIP_F32 && m->stored >= (1 << 25) ? 1024 : 512, &m->items_per_tile)) return 1;
        if (build_bal_tiles(m, st)) return 1;
        return build_block_patterns(m, st);
    case SPMV_HIP_SELL:
    case SPMV_HIP_SELLP:
    case SPMV_HIP_ELL: {
        const bool sell = m->format == SPMV_HIP_SELL;
        m->slice_height = p0 ? (int)p0 : kDefaultC;
        m->sigma = sell ? (p1 ? p1 : kDefaultSigma) : 0;
        if (sell && m->sigma < 1) { set_error("sigma must be positive"); return 1; }
        const int *rp = m->d_row_ptr, *cj = m->d_col;
        const void *va = m->d_val;
        int64_t csr_bytes = m->owns_csr ? m->device_bytes : 0;
        int rc = build_sliced(m, rp, cj, va, csr_val_is_f64, sell, st);
        // the sliced layout replaces the CSR arrays
        release_csr(m);
        m->device_bytes -= csr_bytes;
        return rc;
    }
    case SPMV_HIP_SELL_DFC: {
        m->slice_height = p0 ? (int)p0 : kDefaultC;
        m->sigma = p1 ? p1 : kDefaultSigma;
        if (m->sigma < 1) { set_error("sigma must be positive"); return 1; }
        const int *rp = m->d_row_ptr, *cj = m->d_col;
        const void *va = m->d_val;
        int64_t csr_bytes = m->owns_csr ? m->device_bytes : 0;
        int rc = build_sell_dfc(m, rp, cj, va, csr_val_is_f64, st);
        release_csr(m);
        m->device_bytes -= csr_bytes;
        return rc;
    }
    default: set_error("unknown format %d", m->format); return 1;
    }
}

spmv_hip_matrix *adopt_device_csr(int format, int rows, int columns, int64_t nnz, int *d_row_ptr, int *d_col, void *d_val,
                                  int dtype, int64_t p0, int64_t p1, bool own, hipStream_t stream)
{
    spmv_hip_matrix *m = new_handle(format, dtype, rows, columns, nnz);
    if (!m) {
        if (own) { (void)hipFree(d_row_ptr); (void)hipFree(d_col); (void)hipFree(d_val); }
        return nullptr;
    }
    m->owns_csr = own;
    m->stored = nnz;
    m->d_row_ptr = d_row_ptr; m->d_col = d_col; m->d_val = d_val;
    if (own) m->device_bytes = (int64_t)sizeof(int) * ((int64_t)rows + 1) + std::max<int64_t>(nnz, 1) * (int64_t)(sizeof(int) + dtype_size(dtype));
    if (finish_from_device_csr(m, p0, p1, dtype == SPMV_HIP_F64, stream)) {
        spmv_hip_free(m);
        return nullptr;
    }
    return m;
}

static spmv_hip_matrix *create_from_host_csr(int format, int rows, int columns, int64_t nnz, const std::vector<int> &rp,
                                             const std::vector<int> &cj, const std::vector<double> &va, int dtype,
                                             int64_t p0, int64_t p1)
{
    if (check_columns(columns, cj)) return nullptr;
    spmv_hip_matrix *m = new_handle(format, dtype, rows, columns, nnz);
    if (!m) return nullptr;
    const bool sliced = format == SPMV_HIP_SELL || format == SPMV_HIP_SELLP || format == SPMV_HIP_ELL ||
                        format == SPMV_HIP_SELL_DFC;
    if (upload_csr(m, rp, cj, va, /*keep_f64_values=*/sliced) || finish_from_device_csr(m, p0, p1, sliced || dtype == SPMV_HIP_F64, nullptr)) {
        spmv_hip_free(m);
        return nullptr;
    }
    return m;
}

static spmv_hip_matrix *create_from_coo(int format, int rows, int columns, int nnz, int alignment, struct mtx *coo,
                                        int dtype, int64_t p0, int64_t p1)
{
    clear_error();
    if (check_dims(rows, columns, nnz, dtype)) return nullptr;
    if (nnz > 0 && !coo) { set_error("coo is NULL"); return nullptr; }
    // large unpadded inputs: sort and COO -> CSR in HBM (one upload of the COO instead of a host pass
    // over it and an upload of the CSR); SPMV_HIP_DEVICE_CSR_MIN moves the threshold
    const char *env = getenv("SPMV_HIP_DEVICE_CSR_MIN");
    const long device_min = env ? atol(env) : (1L << 22);
    if (alignment == 1 && nnz >= device_min) {
        const bool sliced = format == SPMV_HIP_SELL || format == SPMV_HIP_SELLP || format == SPMV_HIP_ELL ||
                            format == SPMV_HIP_SELL_DFC;
        const bool values_f64 = sliced || dtype == SPMV_HIP_F64;
        int *d_rp = nullptr, *d_cj = nullptr;
        void *d_va = nullptr;
        const int rc = device_coo_to_csr(rows, columns, nnz, coo, values_f64, &d_rp, &d_cj, &d_va);
        if (rc == 1) return nullptr;
        if (rc == 0) {
            clear_error();
            spmv_hip_matrix *m = new_handle(format, dtype, rows, columns, nnz);
            if (!m) { (void)hipFree(d_rp); (void)hipFree(d_cj); (void)hipFree(d_va); return nullptr; }
            m->owns_csr = true;
            m->stored = nnz;
            m->d_row_ptr = d_rp; m->d_col = d_cj; m->d_val = d_va;
            m->device_bytes = (int64_t)sizeof(int) * ((int64_t)rows + 1) + (int64_t)nnz * (int64_t)(sizeof(int) + (values_f64 ? 8 : 4));
            if (finish_from_device_csr(m, p0, p1, values_f64, nullptr)) { spmv_hip_free(m); return nullptr; }
            return m;
        }
        clear_error();                                        // the device path could not run: host conversion
    }
    sort_coo_in_place(nnz, coo);
    std::vector<int> rp, cj;
    std::vector<double> va;
    if (coo_to_csr(rows, nnz, alignment, coo, rp, cj, va)) return nullptr;
    return create_from_host_csr(format, rows, columns, nnz, rp, cj, va, dtype, p0, p1);
}

static spmv_hip_matrix *create_from_csr_arrays(int format, int rows, int columns, int nnz, const int *row_ptr,
                                               const int *col, const double *A, int dtype, int64_t p0, int64_t p1)
{
    clear_error();
    if (check_dims(rows, columns, nnz, dtype)) return nullptr;
    if (!row_ptr || (nnz > 0 && (!col || !A))) { set_error("NULL CSR array"); return nullptr; }
    if (row_ptr[0] != 0 || row_ptr[rows] != nnz) { set_error("row_ptr[0] must be 0 and row_ptr[rows] must equal nnz"); return nullptr; }
    for (int r = 0; r < rows; r++)
        if (row_ptr[r + 1] < row_ptr[r]) { set_error("row_ptr decreases at row %d", r); return nullptr; }
    std::vector<int> rp(row_ptr, row_ptr + rows + 1), cj(col, col + nnz);
    std::vector<double> va(A, A + nnz);
    return create_from_host_csr(format, rows, columns, nnz, rp, cj, va, dtype, p0, p1);
}

static int ensure_vectors(spmv_hip_matrix *m)
{
    const size_t es = dtype_size(m->dtype);
    if (!m->d_x && dev_alloc(m, (char **)&m->d_x, (size_t)std::max(m->columns, 1) * es)) return 1;
    if (!m->d_y && dev_alloc(m, (char **)&m->d_y, (size_t)std::max(m->rows, 1) * es)) return 1;
    if (!m->d_norm && dev_alloc(m, &m->d_norm, 1)) return 1;
    return 0;
}

static int upload_vector(const spmv_hip_matrix *m, void *d, const double *h, int n)
{
    if (n == 0) return 0;
    if (m->dtype == SPMV_HIP_F64) {
        SPMV_HIP_TRY(hipMemcpy(d, h, sizeof(double) * (size_t)n, hipMemcpyHostToDevice), return 1);
    } else {
        std::vector<float> f(h, h + n);
        SPMV_HIP_TRY(hipMemcpy(d, f.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice), return 1);
    }
    return 0;
}

static int download_vector(const spmv_hip_matrix *m, const void *d, double *h, int n)
{
    if (n == 0) return 0;
    if (m->dtype == SPMV_HIP_F64) {
        SPMV_HIP_TRY(hipMemcpy(h, d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost), return 1);
    } else {
        std::vector<float> f((size_t)n);
        SPMV_HIP_TRY(hipMemcpy(f.data(), d, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost), return 1);
        for (int i = 0; i < n; i++) h[i] = (double)f[(size_t)i];
    }
    return 0;
}

} // namespace spmv

using namespace spmv;

extern "C" {

int spmv_hip_init(int device)
{
    clear_error();
    SPMV_HIP_TRY(hipSetDevice(device), return 1);
    // The block -> XCD remap of the kernels (xcd_contiguous_block) and the per-XCD column split of
    // hot_columns 2 assume the 8 XCDs x 32 CUs of one whole MI355X.  In another compute partition the remap
    // is still a bijection, so results do not change; only the L2 locality it was built for is lost.
    int cus = 0;
    SPMV_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device), return 1);
    static bool warned = false;
    if (cus != 256 && !warned) {
        warned = true;
        fprintf(stderr, "spmv_hip: device %d shows %d compute units, not the 256 (8 XCDs) the block placement is "
                        "tuned for; results are unaffected\n", device, cus);
    }
    return static_scratch_setup();        // so that spmv_hip_norm2_dev_noalloc has nothing left to look up
}

const char *spmv_hip_last_error(void) { return g_error; }

int spmv_hip_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    clear_error();
    size_t f = 0, t = 0;
    SPMV_HIP_TRY(hipMemGetInfo(&f, &t), return 1);
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

spmv_hip_matrix *spmv_hip_create_csr(int rows, int columns, int nnz, int alignment, struct mtx *coo, int dtype,
                                     int kernel, int lanes_per_row)
{
    if (kernel != SPMV_HIP_CSR_SCALAR && kernel != SPMV_HIP_CSR_VECTOR) {
        set_error("create_csr: kernel must be SPMV_HIP_CSR_SCALAR or SPMV_HIP_CSR_VECTOR");
        return nullptr;
    }
    return create_from_coo(kernel, rows, columns, nnz, alignment, coo, dtype, lanes_per_row, 0);
}

spmv_hip_matrix *spmv_hip_create_csr_merge(int rows, int columns, int nnz, struct mtx *coo, int dtype, int items_per_tile)
{
    return create_from_coo(SPMV_HIP_CSR_MERGE, rows, columns, nnz, 1, coo, dtype, items_per_tile, 0);
}

spmv_hip_matrix *spmv_hip_create_csr_bal(int rows, int columns, int nnz, struct mtx *coo, int dtype, int nnz_per_tile)
{
    return create_from_coo(SPMV_HIP_CSR_BAL, rows, columns, nnz, 1, coo, dtype, nnz_per_tile, 0);
}

spmv_hip_matrix *spmv_hip_create_ell(int rows, int columns, int nnz, struct mtx *coo, int dtype, int slice_height)
{
    return create_from_coo(SPMV_HIP_ELL, rows, columns, nnz, 1, coo, dtype, slice_height, 0);
}

spmv_hip_matrix *spmv_hip_create_sellp(int rows, int columns, int nnz, const int *row_ptr, const int *col,
                                       const double *A, int dtype, int slice_height)
{
    return create_from_csr_arrays(SPMV_HIP_SELLP, rows, columns, nnz, row_ptr, col, A, dtype, slice_height, 0);
}

spmv_hip_matrix *spmv_hip_create_sell(int rows, int columns, int nnz, const int *row_ptr, const int *col,
                                      const double *A, int dtype, int C, int64_t sigma)
{
    return create_from_csr_arrays(SPMV_HIP_SELL, rows, columns, nnz, row_ptr, col, A, dtype, C, sigma);
}

spmv_hip_matrix *spmv_hip_create_sell_dfc(int rows, int columns, int nnz, const int *row_ptr, const int *col,
                                          const double *A, int dtype, int C, int64_t sigma)
{
    return create_from_csr_arrays(SPMV_HIP_SELL_DFC, rows, columns, nnz, row_ptr, col, A, dtype, C, sigma);
}

spmv_hip_matrix *spmv_hip_create_from_device_csr(int format, int rows, int columns, int64_t nnz, const int *d_row_ptr,
                                                 const int *d_col, const void *d_val, int dtype, int64_t p0, int64_t p1,
                                                 void *stream)
{
    clear_error();
    if (check_dims(rows, columns, nnz, dtype)) return nullptr;
    if (nnz > 0x7fffffffLL) { set_error("nnz %lld exceeds the 32-bit row pointers of CSR", (long long)nnz); return nullptr; }
    if (!d_row_ptr || (nnz > 0 && (!d_col || !d_val))) { set_error("NULL device CSR array"); return nullptr; }
    spmv_hip_matrix *m = new_handle(format, dtype, rows, columns, nnz);
    if (!m) return nullptr;
    m->owns_csr = false;
    m->stored = nnz;
    m->d_row_ptr = const_cast<int *>(d_row_ptr);
    m->d_col = const_cast<int *>(d_col);
    m->d_val = const_cast<void *>(d_val);
    if (finish_from_device_csr(m, p0, p1, dtype == SPMV_HIP_F64, (hipStream_t)stream)) {
        spmv_hip_free(m);
        return nullptr;
    }
    return m;
}

int spmv_hip_mult_dev(spmv_hip_matrix *m, const void *d_x, void *d_y, void *stream)
{
    if (!m) { set_error("NULL handle"); return 1; }
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    return launch_mult(m, d_x, d_y, (hipStream_t)stream);
}

int spmv_hip_mult(spmv_hip_matrix *m, const double *x, double *y)
{
    clear_error();
    if (!m || !x || !y) { set_error("NULL argument"); return 1; }
    if (m->d_row_map) { set_error("row-mapped handles multiply through spmv_hip_mult_dev only"); return 1; }
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    if (ensure_vectors(m)) return 1;
    if (upload_vector(m, m->d_x, x, m->columns)) return 1;
    if (launch_mult(m, m->d_x, m->d_y, nullptr)) return 1;
    SPMV_HIP_TRY(hipDeviceSynchronize(), return 1);
    return download_vector(m, m->d_y, y, m->rows);
}

int spmv_hip_mult_mv_dev(spmv_hip_matrix *m, int n, const void *d_x, int64_t ldx, void *d_y, int64_t ldy, void *stream)
{
    if (!m) { set_error("NULL handle"); return 1; }
    if (n < 0) { set_error("mult_mv: negative vector count %d", n); return 1; }
    if (n > 1 && (ldx < m->columns || ldy < m->rows)) {
        set_error("mult_mv: ldx %lld / ldy %lld shorter than the vectors (%d columns, %d rows)", (long long)ldx,
                  (long long)ldy, m->columns, m->rows);
        return 1;
    }
    if (n > 0 && (!d_x || !d_y) && m->rows > 0) { set_error("mult_mv: NULL vector block"); return 1; }
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    return launch_mult_mv(m, n, d_x, ldx, d_y, ldy, (hipStream_t)stream);
}

int spmv_hip_mult_mv(spmv_hip_matrix *m, int n, const double *x, double *y)
{
    clear_error();
    if (!m || !x || !y) { set_error("NULL argument"); return 1; }
    if (n < 0) { set_error("mult_mv: negative vector count %d", n); return 1; }
    if (m->d_row_map) { set_error("row-mapped handles multiply through spmv_hip_mult_mv_dev only"); return 1; }
    if (n == 0) return 0;
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    const size_t es = dtype_size(m->dtype);
    const size_t nx = (size_t)n * (size_t)m->columns, ny = (size_t)n * (size_t)m->rows;
    char *dx = nullptr, *dy = nullptr;
    int rc = 1;
    do {
        SPMV_HIP_TRY(hipMalloc((void **)&dx, std::max<size_t>(nx, 1) * es), break);
        SPMV_HIP_TRY(hipMalloc((void **)&dy, std::max<size_t>(ny, 1) * es), break);
        bool ok = true;
        for (int v = 0; v < n && ok; v++)
            ok = upload_vector(m, dx + (size_t)v * m->columns * es, x + (size_t)v * m->columns, m->columns) == 0;
        if (!ok) break;
        if (launch_mult_mv(m, n, dx, m->columns, dy, m->rows, nullptr)) break;
        SPMV_HIP_TRY(hipDeviceSynchronize(), break);
        for (int v = 0; v < n && ok; v++)
            ok = download_vector(m, dy + (size_t)v * m->rows * es, y + (size_t)v * m->rows, m->rows) == 0;
        if (!ok) break;
        rc = 0;
    } while (0);
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    return rc;
}

int spmv_hip_free(spmv_hip_matrix *m)
{
    if (!m) return 0;
    DeviceGuard on(m->device);            // a failed switch still frees: hipFree takes pointers of any device
    release_csr(m);
    void *ptrs[] = {m->d_tile_coord, m->d_row_carry, m->d_val_carry, m->d_nzr_id, m->d_nzr_end, m->d_slice_ptr,
                    m->d_row_order, m->d_scol, m->d_sval, m->d_sorted_len, m->d_vop_ptr, m->d_vop_len,
                    m->d_x, m->d_y, m->d_norm, m->d_hot_col, m->d_hot_old, m->d_hot_x, m->d_xcd_y, m->d_xcd_presence, m->d_xcd_base, m->d_sdiag, m->d_block_pat, m->d_tile_pat};
    if (m->xcd_sub) spmv_hip_free(m->xcd_sub);
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
    return 0;
}

int spmv_hip_bench(spmv_hip_matrix *m, int min_iters, double min_seconds, double *x, int *iters, double *seconds)
{
    clear_error();
    if (!m || !x || !iters || !seconds) { set_error("NULL argument"); return 1; }
    if (m->rows != m->columns) { set_error("bench: matrix is not square"); return 1; }
    if (m->d_row_map) { set_error("row-mapped handles multiply through spmv_hip_mult_dev only"); return 1; }
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    if (ensure_vectors(m)) return 1;
    if (upload_vector(m, m->d_x, x, m->columns)) return 1;
    hipEvent_t e0, e1;
    SPMV_HIP_TRY(hipEventCreate(&e0), return 1);
    SPMV_HIP_TRY(hipEventCreate(&e1), return 1);
    int rc = 1, it = 0;
    double total_ms = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        it++;
        SPMV_HIP_TRY(hipEventRecord(e0, nullptr), break);
        if (launch_mult(m, m->d_x, m->d_y, nullptr)) break;
        SPMV_HIP_TRY(hipEventRecord(e1, nullptr), break);
        if (launch_norm2(m->rows, m->d_y, m->dtype, m->d_norm, nullptr)) break;
        if (launch_scale(m->rows, m->d_y, m->d_x, m->dtype, m->d_norm, nullptr)) break;
        SPMV_HIP_TRY(hipEventSynchronize(e1), break);
        float ms = 0;
        SPMV_HIP_TRY(hipEventElapsedTime(&ms, e0, e1), break);
        total_ms += ms;
        double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (wall > min_seconds && it >= min_iters) { rc = 0; break; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    SPMV_HIP_TRY(hipDeviceSynchronize(), return 1);
    *iters = it;
    *seconds = total_ms * 1e-3;
    return download_vector(m, m->d_x, x, m->columns);
}

int spmv_hip_norm2_dev(int n, const void *d_x, int dtype, double *d_out, void *stream)
{
    return launch_norm2(n, d_x, dtype, d_out, (hipStream_t)stream);
}

int spmv_hip_release_stream(void *stream)
{
    return release_norm_scratch((hipStream_t)stream);
}

int spmv_hip_norm2_dev_noalloc(int64_t n, const void *d_x, int dtype, double *d_out, void *stream)
{
    return launch_norm2_static(n, d_x, dtype, d_out, (hipStream_t)stream);
}

int spmv_hip_scale_dev(int n, const void *d_y, void *d_x, int dtype, const double *d_norm, void *stream)
{
    return launch_scale(n, d_y, d_x, dtype, d_norm, (hipStream_t)stream);
}

static void drop_hot_columns(spmv_hip_matrix *m)
{
    if (m->xcd_sub) {
        m->device_bytes -= m->xcd_bytes;
        m->xcd_bytes = 0;
        spmv_hip_free(m->xcd_sub);
        (void)hipFree(m->d_hot_old); (void)hipFree(m->d_hot_x); (void)hipFree(m->d_xcd_y);
        (void)hipFree(m->d_xcd_presence); (void)hipFree(m->d_xcd_base);
        m->xcd_sub = nullptr; m->d_hot_old = nullptr; m->d_hot_x = nullptr; m->d_xcd_y = nullptr; m->xcd_cpx = 0; m->xcd_used_cpx = 0;
        m->d_xcd_presence = nullptr; m->d_xcd_base = nullptr;
        return;
    }
    if (!m->d_hot_col) return;
    m->device_bytes -= (int64_t)sizeof(int) * ((int64_t)m->columns + m->stored) + (int64_t)m->columns * (int64_t)dtype_size(m->dtype);
    (void)hipFree(m->d_hot_col); (void)hipFree(m->d_hot_old); (void)hipFree(m->d_hot_x);
    m->d_hot_col = nullptr; m->d_hot_old = nullptr; m->d_hot_x = nullptr;
}

// "hot_columns" = 2: stack the eight column groups into one balanced-CSR handle of 8*rows virtual rows
static int build_xcd_split(spmv_hip_matrix *m, hipStream_t st)
{
    if (m->d_row_map) { set_error("hot_columns 2 does not combine with a row map"); return 1; }
    int *vrp = nullptr, *vcol = nullptr, *old_of_label = nullptr, cpx = 0, used_cpx = 0;
    void *vval = nullptr;
    if (build_xcd_columns(m, m->dtype == SPMV_HIP_F64, st, &vrp, &vcol, &vval, &old_of_label, &cpx, &used_cpx)) return 1;
    const int64_t es = (int64_t)dtype_size(m->dtype), labels = 8 * (int64_t)cpx;
    spmv_hip_matrix *sub = new_handle(SPMV_HIP_CSR_BAL, m->dtype, 8 * m->rows, (int)labels, m->nnz);
    void *xh = nullptr, *yp = nullptr;
    unsigned char *presence = nullptr;
    int *base_count = nullptr, *base = nullptr;
    int rc = 1;
    do {
        if (!sub) break;
        sub->owns_csr = true;
        sub->stored = m->stored;
        sub->d_row_ptr = vrp; sub->d_col = vcol; sub->d_val = vval;
        sub->device_bytes = (int64_t)sizeof(int) * (8 * (int64_t)m->rows + 1) + m->stored * (4 + es);
        vrp = nullptr; vcol = nullptr; vval = nullptr;                    // the handle frees them from here on
        if (finish_from_device_csr(sub, 512, 0, m->dtype == SPMV_HIP_F64, st)) break;     // the tile size of the LDS-resident kernels
        sub->leave_empty_rows = true;                                     // the final sum skips empty virtual rows
        {
            // where each group's piece starts in the tiles: piece g begins at entry vrow_ptr[g * rows]
            int first_entry[9];
            hipError_t copied = hipSuccess;
            for (int g = 0; g <= 8 && copied == hipSuccess; g++)
                copied = hipMemcpyAsync(&first_entry[g], sub->d_row_ptr + (size_t)g * m->rows, sizeof(int),
                                        hipMemcpyDeviceToHost, st);
            SPMV_HIP_TRY(copied, break);
            SPMV_HIP_TRY(hipStreamSynchronize(st), break);
            if (sub->items_per_tile > 0) {
                for (int g = 0; g < 8; g++) sub->group_first_tile[g] = std::min(first_entry[g] / sub->items_per_tile, sub->ntiles);
                sub->group_first_tile[0] = 0;
                sub->group_first_tile[8] = sub->ntiles;
                sub->group_cpx = cpx;
            }
        }
        SPMV_HIP_TRY(hipMalloc(&xh, (size_t)(labels * es)), break);
        SPMV_HIP_TRY(hipMemsetAsync(xh, 0, (size_t)(labels * es), st), break);     // labels no entry names keep this zero
        // the partial results stay packed, one slot per non-empty virtual row
        sub->packed_rows = true;
        SPMV_HIP_TRY(hipMalloc(&yp, (size_t)std::max<int64_t>(sub->nz_rows, 1) * es), break);
        SPMV_HIP_TRY(hipMalloc((void **)&presence, (size_t)m->rows), break);
        const size_t nbase = 8 * (((size_t)m->rows + kWave - 1) / kWave) + 1;
        SPMV_HIP_TRY(hipMalloc((void **)&base_count, nbase * sizeof(int)), break);
        SPMV_HIP_TRY(hipMalloc((void **)&base, nbase * sizeof(int)), break);
        if (launch_group_presence(m->rows, sub->d_row_ptr, presence, st)) break;
        SPMV_HIP_TRY(hipMemsetAsync(base_count, 0, nbase * sizeof(int), st), break);
        if (launch_group_block_count(m->rows, presence, base_count, st)) break;
        if (device_exclusive_scan(base_count, base, nbase, st)) break;
        SPMV_HIP_TRY(hipStreamSynchronize(st), break);
        m->xcd_sub = sub; m->xcd_cpx = cpx; m->xcd_used_cpx = used_cpx;
        m->d_hot_old = old_of_label; m->d_hot_x = xh; m->d_xcd_y = yp; m->d_xcd_presence = presence; m->d_xcd_base = base;
        m->xcd_bytes = sub->device_bytes + labels * (4 + es) + (int64_t)sub->nz_rows * es + m->rows + (int64_t)(nbase * sizeof(int));
        m->device_bytes += m->xcd_bytes;
        sub = nullptr; old_of_label = nullptr; xh = nullptr; yp = nullptr; presence = nullptr; base = nullptr;
        rc = 0;
    } while (0);
    if (sub) spmv_hip_free(sub);
    void *scratch[] = {vrp, vcol, vval, old_of_label, xh, yp, presence, base_count, base};
    for (void *p : scratch)
        if (p) (void)hipFree(p);
    return rc;
}

int spmv_hip_set_option(spmv_hip_matrix *m, const char *name, int64_t value)
{
    clear_error();
    if (!m || !name) { set_error("NULL argument"); return 1; }
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    if (!strcmp(name, "variant")) { m->variant = (int)value; return 0; }
    if (!strcmp(name, "block_map")) {
        if (value < 0 || value > (1 << 20)) { set_error("block_map takes 0, 1 or a run length up to 2^20"); return 1; }
        m->block_map = (int)value;
        return 0;
    }
    if (!strcmp(name, "hot_variant")) {                     // the tile kernel of the stacked matrix behind hot_columns 2
        if (!m->xcd_sub) { set_error("hot_variant needs hot_columns 2"); return 1; }
        int cand[32];
        const int n = variant_candidates(m->xcd_sub, cand, 32);
        bool known = false;
        for (int i = 0; i < n; i++) known = known || cand[i] == (int)value;
        if (!known) { set_error("hot_variant %lld is not a tile kernel of this handle (0, 1, 3, 4)", (long long)value); return 1; }
        m->xcd_sub->variant = (int)value;
        return 0;
    }
    if (!strcmp(name, "exact_order")) {
        m->exact_order = value != 0;
        // sliced formats: leave the reassociating 16-byte-load variants for the exact-order kernel
        const bool sliced64 = m->d_slice_ptr && m->slice_height == kWave && m->format != SPMV_HIP_SELL_DFC;
        if (m->exact_order && sliced64 && m->variant >= 1 && m->variant <= 12) m->variant = 0;
        return 0;
    }
    if (!strcmp(name, "hot_columns")) {
        if (value == 0) { drop_hot_columns(m); return 0; }
        if (value != 1 && value != 2) { set_error("hot_columns takes 0, 1 or 2"); return 1; }
        if ((value == 1 && m->d_hot_col) || (value == 2 && m->xcd_sub)) return 0;
        drop_hot_columns(m);
        if (m->format != SPMV_HIP_CSR_SCALAR && m->format != SPMV_HIP_CSR_VECTOR && m->format != SPMV_HIP_CSR_MERGE &&
            m->format != SPMV_HIP_CSR_BAL) {
            set_error("hot_columns needs a CSR-based format (scalar, vector, merge, balanced)");
            return 1;
        }
        return value == 1 ? build_hot_columns(m, nullptr) : build_xcd_split(m, nullptr);
    }
    set_error("unknown option '%s'", name);
    return 1;
}

int spmv_hip_set_row_map(spmv_hip_matrix *m, const int *d_row_map)
{
    clear_error();
    if (!m) { set_error("NULL handle"); return 1; }
    if (m->format != SPMV_HIP_CSR_SCALAR && m->format != SPMV_HIP_CSR_VECTOR) {
        set_error("row map is only supported by CSR_SCALAR / CSR_VECTOR handles");
        return 1;
    }
    m->d_row_map = d_row_map;
    return 0;
}

// Values of the option "block_map" a kernel trial compares: an eighth of the matrix per XCD (0), consecutive
// workgroups on consecutive pieces (1), and short runs of workgroups per XCD -- all XCDs then work side by side in
// one region of the matrix and of y (27-point stencil 252^3, fp64: SELL-64-1024 0.803 ms with 0, 0.794 with 1,
// 0.761 with 5; SELLP with every index read 0.946 / 0.921 / 0.909 with 0 / 1 / 4; lane-per-row CSR 1.091 / 1.075 /
// 1.055 with 0 / 1 / 8).
static const int kBlockMapCandidates[] = {0, 1, 2, 3, 4, 5, 8};

// Times every kernel variant of the handle `reps` times and keeps the fastest, then the ways of dealing the
// workgroups over the matrix likewise; *best_ms_out (optional) receives the summed time of the winner's `reps`
// launches (handles with a single variant are timed only when the caller asks for the time).
static int analyze_impl(spmv_hip_matrix *m, int reps, float *best_ms_out)
{
    if (best_ms_out) *best_ms_out = 0;
    if (reps < 1) reps = 3;
    int cand[32];
    int n = variant_candidates(m, cand, 32);
    if (m->rows == 0 || m->d_row_map) return 0;             // row-mapped handles write outside d_y
    if (m->xcd_sub) {
        // the product runs on the stacked handle: tune that one, then time the whole sequence once
        if (analyze_impl(m->xcd_sub, reps, nullptr)) return 1;
        n = 1;
        cand[0] = m->format == SPMV_HIP_CSR_VECTOR ? m->lanes_per_row + 256 * m->variant : m->variant;
    }
    if (n <= 1 && !best_ms_out) return 0;
    if (ensure_vectors(m)) return 1;
    const size_t es = dtype_size(m->dtype);
    SPMV_HIP_TRY(hipMemset(m->d_x, 0, (size_t)std::max(m->columns, 1) * es), return 1);
    hipEvent_t e0, e1;
    SPMV_HIP_TRY(hipEventCreate(&e0), return 1);
    SPMV_HIP_TRY(hipEventCreate(&e1), return 1);
    // a candidate is a kernel variant; for the vector CSR kernel it also carries the lanes per row (low byte)
    const bool vec = m->format == SPMV_HIP_CSR_VECTOR;
    auto apply = [&](int c) {
        if (vec) { m->lanes_per_row = c & 255; m->variant = c >> 8; }
        else m->variant = c;
    };
    const int keep = vec ? m->lanes_per_row + 256 * m->variant : m->variant;
    int best = keep, rc = 0;
    float best_ms = 0;
    if (m->exact_order && !m->xcd_sub) {                    // keep only the candidates that leave y's bits alone
        int kept = 0;
        for (int i = 0; i < n; i++)
            if (candidate_keeps_bits(m, cand[i])) cand[kept++] = cand[i];
        n = kept;
        if (n == 0 || (n == 1 && cand[0] == keep && !best_ms_out)) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return 0;
        }
    }
    // one warm-up launch, then `reps` launches of the handle as it is set, queued one after the other between one pair
    // of events (an event pair around every launch would add the start of a launch on an empty queue to each: 2 us,
    // a sixth of a product of a small matrix, and noise of that size)
    auto time_current = [&](float *total) {
        if (launch_mult(m, m->d_x, m->d_y, nullptr)) return 1;
        if (hipEventRecord(e0, nullptr) != hipSuccess) return 1;
        for (int r = 0; r < reps; r++)
            if (launch_mult(m, m->d_x, m->d_y, nullptr)) return 1;
        if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return 1;
        return hipEventElapsedTime(total, e0, e1) != hipSuccess ? 1 : 0;
    };
    bool first = true;
    auto pick_variant = [&]() {
        for (int i = 0; i < n && !rc; i++) {
            apply(cand[i]);
            float total = 0;
            rc = time_current(&total);
            if (!rc && (first || total < best_ms)) { best_ms = total; best = cand[i]; first = false; }
        }
    };
    pick_variant();
    // then the winner with its workgroups dealt over the matrix in other ways (option "block_map"; the
    // result's bits do not depend on it), and, when another way wins, the variants once more under it.
    // Not for the stacked matrix behind hot_columns 2: there each XCD has to stay with its column group.
    if (!rc && !first && !m->xcd_sub && m->group_cpx == 0) {
        const int keep_map = m->block_map;
        int best_map = keep_map;
        apply(best);
        for (int map : kBlockMapCandidates) {
            if (map == keep_map) continue;
            m->block_map = map;
            float total = 0;
            if ((rc = time_current(&total))) break;
            if (total < best_ms) { best_ms = total; best_map = map; }
        }
        m->block_map = rc ? keep_map : best_map;
        if (!rc && best_map != keep_map && n > 1) pick_variant();
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    apply(rc ? keep : best);
    if (rc && !g_error[0]) set_error("analyze: timing a kernel variant failed");
    if (!rc && best_ms_out) *best_ms_out = best_ms;
    return rc;
}

int spmv_hip_analyze(spmv_hip_matrix *m, int reps)
{
    clear_error();
    if (!m) { set_error("NULL handle"); return 1; }
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    return analyze_impl(m, reps, nullptr);
}

// Candidate formats of the automatic choice, in build order.  The sliced layouts are skipped when they
// would store more than kAutoMaxPadding times the non-zeros.
// SELL-C-sigma is tried with windows of 1024 rows (results stay near their rows) and with the reference's own
// default of 16384 (sell.c:5: less padding where row lengths vary a lot -- FEM stand-in HV15R 0.704 -> 0.664 ms).
struct auto_candidate { int format; int64_t sigma; };
static const auto_candidate kAutoFormats[] = {{SPMV_HIP_CSR_MERGE, 0}, {SPMV_HIP_CSR_SCALAR, 0}, {SPMV_HIP_CSR_VECTOR, 0},
                                              {SPMV_HIP_SELLP, 0},     {SPMV_HIP_SELL, 1024},  {SPMV_HIP_SELL, 16384}};
constexpr double kAutoMaxPadding = 2.0;
constexpr int64_t kHotColumnsMinBytes = (int64_t)4 << 20;   // the L2 of one XCD

static spmv_hip_matrix *view_of_device_csr(int format, int64_t sigma, int rows, int columns, int64_t nnz, const int *d_row_ptr,
                                           const int *d_col, const void *d_val, int dtype, hipStream_t st)
{
    spmv_hip_matrix *m = new_handle(format, dtype, rows, columns, nnz);
    if (!m) return nullptr;
    m->owns_csr = false;
    m->stored = nnz;
    m->d_row_ptr = const_cast<int *>(d_row_ptr);
    m->d_col = const_cast<int *>(d_col);
    m->d_val = const_cast<void *>(d_val);
    m->max_padding = kAutoMaxPadding;
    const int64_t p0 = format == SPMV_HIP_SELL ? kDefaultC : 0, p1 = format == SPMV_HIP_SELL ? sigma : 0;
    if (finish_from_device_csr(m, p0, p1, dtype == SPMV_HIP_F64, st)) {
        spmv_hip_free(m);
        return nullptr;
    }
    return m;
}

// Builds each candidate over the same device CSR, tunes and times it, keeps the fastest.
static spmv_hip_matrix *auto_from_device_csr(int rows, int columns, int64_t nnz, const int *d_row_ptr, const int *d_col,
                                             const void *d_val, int dtype, int reps, hipStream_t st)
{
    spmv_hip_matrix *best = nullptr;
    float best_ms = 0;
    for (const auto_candidate &c : kAutoFormats) {
        if (c.format == SPMV_HIP_SELL && c.sigma > 1024 && rows <= 2048) continue;      // one window either way
        spmv_hip_matrix *m = view_of_device_csr(c.format, c.sigma, rows, columns, nnz, d_row_ptr, d_col, d_val, dtype, st);
        if (!m) continue;                                    // e.g. too much padding: not a candidate
        float ms = 0;
        if (analyze_impl(m, reps, &ms)) { spmv_hip_free(m); continue; }
        if (!best || ms < best_ms) {
            if (best) spmv_hip_free(best);
            best = m; best_ms = ms;
        } else {
            spmv_hip_free(m);
        }
    }
    // a CSR-based winner whose x does not fit one L2: try the columns relabeled by reference count
    // (pays on power-law column counts, costs a pass over x otherwise) and keep it if clearly faster
    const bool csr_based = best && (best->format == SPMV_HIP_CSR_MERGE || best->format == SPMV_HIP_CSR_SCALAR ||
                                    best->format == SPMV_HIP_CSR_VECTOR || best->format == SPMV_HIP_CSR_BAL);
    if (csr_based && (int64_t)columns * (int64_t)dtype_size(dtype) > kHotColumnsMinBytes && nnz > 0) {
        const int keep_variant = best->variant, keep_lanes = best->lanes_per_row;
        float ms = 0;
        if (build_hot_columns(best, st) == 0 && analyze_impl(best, reps, &ms) == 0 && ms < 0.95f * best_ms) {
            best_ms = ms;
            // it pays: then the columns dealt over the XCDs may pay more (second copy of the matrix)
            const int hot_variant = best->variant, hot_lanes = best->lanes_per_row;
            drop_hot_columns(best);
            if (build_xcd_split(best, st) == 0 && analyze_impl(best, reps, &ms) == 0 && ms < 0.95f * best_ms) {
                best_ms = ms;
            } else {
                drop_hot_columns(best);
                best->variant = hot_variant;
                best->lanes_per_row = hot_lanes;
                if (build_hot_columns(best, st)) {            // back to the plain handle
                    best->variant = keep_variant;
                    best->lanes_per_row = keep_lanes;
                }
            }
        } else {
            drop_hot_columns(best);
            best->variant = keep_variant;
            best->lanes_per_row = keep_lanes;
        }
    }
    if (best) {
        clear_error();                                       // skipped candidates left their reason behind
    } else if (!g_error[0]) {
        set_error("create_auto: no format could be built");
    }
    return best;
}

spmv_hip_matrix *spmv_hip_create_auto_from_device_csr(int rows, int columns, int64_t nnz, const int *d_row_ptr,
                                                      const int *d_col, const void *d_val, int dtype, int reps,
                                                      void *stream)
{
    clear_error();
    if (check_dims(rows, columns, nnz, dtype)) return nullptr;
    if (nnz > 0x7fffffffLL) { set_error("nnz %lld exceeds the 32-bit row pointers of CSR", (long long)nnz); return nullptr; }
    if (!d_row_ptr || (nnz > 0 && (!d_col || !d_val))) { set_error("NULL device CSR array"); return nullptr; }
    return auto_from_device_csr(rows, columns, nnz, d_row_ptr, d_col, d_val, dtype, reps, (hipStream_t)stream);
}

spmv_hip_matrix *spmv_hip_create_auto(int rows, int columns, int nnz, struct mtx *coo, int dtype, int reps)
{
    // the CSR arrays live in a merge-path handle; the winner either is that handle, takes the
    // arrays over from it (CSR family), or holds its own sliced copy
    spmv_hip_matrix *base = create_from_coo(SPMV_HIP_CSR_MERGE, rows, columns, nnz, 1, coo, dtype, 0, 0);
    if (!base) return nullptr;
    spmv_hip_matrix *best = auto_from_device_csr(rows, columns, nnz, base->d_row_ptr, base->d_col, base->d_val, dtype,
                                                 reps, nullptr);
    if (!best) { spmv_hip_free(base); return nullptr; }
    if (best->d_row_ptr == base->d_row_ptr) {                // a CSR-family view of base's arrays
        best->owns_csr = true;
        base->owns_csr = false;
        const int64_t csr_bytes = (int64_t)sizeof(int) * (rows + 1) + (int64_t)std::max(nnz, 1) * (sizeof(int) + dtype_size(dtype));
        best->device_bytes += csr_bytes;
    }
    spmv_hip_free(base);
    return best;
}

int spmv_hip_get_info(const spmv_hip_matrix *m, spmv_hip_info *info)
{
    if (!m || !info) { set_error("NULL argument"); return 1; }
    memset(info, 0, sizeof *info);
    info->format = m->format; info->dtype = m->dtype;
    info->rows = m->rows; info->columns = m->columns;
    info->nnz = m->nnz; info->stored = m->stored;
    info->slice_height = m->slice_height; info->sigma = m->sigma; info->nslices = m->nslices;
    info->lanes_per_row = m->lanes_per_row;
    info->ntiles = m->ntiles; info->items_per_tile = m->items_per_tile;
    info->device_bytes = m->device_bytes;
    info->variant = m->variant;
    info->hot_columns = m->xcd_sub ? 2 : (m->d_hot_col != nullptr);
    info->diag_columns = m->diag_columns;
    info->pattern_blocks = m->pat_blocks;
    info->hot_variant = m->xcd_sub ? m->xcd_sub->variant : 0;
    info->block_map = m->block_map;
    return 0;
}

int spmv_hip_download(const spmv_hip_matrix *m, int which, void *host, size_t bytes)
{
    clear_error();
    if (!m || !host) { set_error("NULL argument"); return 1; }
    const void *src = nullptr;
    size_t need = 0;
    const bool sliced = m->d_slice_ptr != nullptr;
    switch (which) {
    case SPMV_HIP_ARR_ROW_PTR: src = m->d_row_ptr; need = sizeof(int) * ((size_t)m->rows + 1); break;
    case SPMV_HIP_ARR_COL: src = sliced ? m->d_scol : m->d_col; need = sizeof(int) * (size_t)m->stored; break;
    case SPMV_HIP_ARR_VAL: src = sliced ? m->d_sval : m->d_val; need = dtype_size(m->dtype) * (size_t)m->stored; break;
    case SPMV_HIP_ARR_SLICE_PTR: src = m->d_slice_ptr; need = sizeof(int64_t) * ((size_t)m->nslices + 1); break;
    case SPMV_HIP_ARR_ROW_ORDER: src = m->d_row_order; need = sizeof(int) * (size_t)m->rows; break;
    case SPMV_HIP_ARR_TILE_COORD: src = m->d_tile_coord; need = sizeof(int) * 2 * ((size_t)m->ntiles + 1); break;
    case SPMV_HIP_ARR_VOP_PTR: src = m->d_vop_ptr; need = sizeof(int64_t) * ((size_t)m->nslices + 1); break;
    case SPMV_HIP_ARR_VOP_LEN: src = m->d_vop_len; need = (size_t)m->vops; break;
    default: set_error("unknown array id %d", which); return 1;
    }
    if (!src) { set_error("array %d is not part of format %d", which, m->format); return 1; }
    if (bytes != need) { set_error("array %d holds %zu bytes, caller passed %zu", which, need, bytes); return 1; }
    if (need == 0) return 0;
    DeviceGuard on(m->device);
    if (!on.ok) return 1;
    SPMV_HIP_TRY(hipMemcpy(host, src, need, hipMemcpyDeviceToHost), return 1);
    return 0;
}

} // extern "C"
