// (row, col) sort of an AoS COO array on the device.
//
// The reference sorts every unsorted matrix with libc qsort on the host
// (sort_mtx + by_row_mtx, reference src/mtx.c:9-34) -- and MatrixMarket files
// come column-major, so that is every file: ~10 s per 100 M entries.  Here the
// entries are packed into 64-bit keys (row in the high word, column in the low
// word, sign bits flipped so that unsigned key order equals the signed (i, j)
// order), radix-sorted together with their values by rocPRIM, and unpacked into
// the caller's array.  The sort is stable, so duplicate coordinates keep their
// file order (qsort leaves that unspecified).

#include "internal.h"

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace spmv {

__global__ void coo_pack_kernel(int64_t n, const struct mtx *__restrict__ coo, uint64_t *__restrict__ key,
                                double *__restrict__ val)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const struct mtx e = coo[k];
    key[k] = ((uint64_t)((uint32_t)e.i ^ 0x80000000u) << 32) | ((uint32_t)e.j ^ 0x80000000u);
    val[k] = e.a;
}

__global__ void coo_unpack_kernel(int64_t n, const uint64_t *__restrict__ key, const double *__restrict__ val,
                                  struct mtx *__restrict__ coo)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    struct mtx e;
    e.i = (int)((uint32_t)(key[k] >> 32) ^ 0x80000000u);
    e.j = (int)((uint32_t)key[k] ^ 0x80000000u);
    e.a = val[k];
    coo[k] = e;
}

// Returns 0 when `coo` has been sorted; non-zero (message set) when the device path could not run,
// in which case `coo` is untouched.
int device_sort_coo(int64_t nnz, struct mtx *coo)
{
    if (nnz < 2) return 0;
    struct mtx *d_coo = nullptr;
    uint64_t *d_key[2] = {nullptr, nullptr};
    double *d_val[2] = {nullptr, nullptr};
    void *d_temp = nullptr;
    int rc = 1;
    do {
        const size_t n = (size_t)nnz;
        SPMV_HIP_TRY(hipMalloc((void **)&d_coo, n * sizeof(struct mtx)), break);
        for (int b = 0; b < 2; b++) {
            SPMV_HIP_TRY(hipMalloc((void **)&d_key[b], n * sizeof(uint64_t)), break);
            SPMV_HIP_TRY(hipMalloc((void **)&d_val[b], n * sizeof(double)), break);
        }
        if (!d_val[1]) break;
        SPMV_HIP_TRY(hipMemcpy(d_coo, coo, n * sizeof(struct mtx), hipMemcpyHostToDevice), break);
        const dim3 grid((unsigned)((nnz + 255) / 256)), block(256);
        hipLaunchKernelGGL(coo_pack_kernel, grid, block, 0, nullptr, nnz, d_coo, d_key[0], d_val[0]);
        SPMV_HIP_TRY(hipGetLastError(), break);
        size_t temp_bytes = 0;
        SPMV_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n, 0, 64,
                                               (hipStream_t) nullptr), break);
        SPMV_HIP_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), break);
        SPMV_HIP_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_key[0], d_key[1], d_val[0], d_val[1], n, 0, 64,
                                               (hipStream_t) nullptr), break);
        hipLaunchKernelGGL(coo_unpack_kernel, grid, block, 0, nullptr, nnz, d_key[1], d_val[1], d_coo);
        SPMV_HIP_TRY(hipGetLastError(), break);
        SPMV_HIP_TRY(hipMemcpy(coo, d_coo, n * sizeof(struct mtx), hipMemcpyDeviceToHost), break);
        rc = 0;
    } while (0);
    if (d_temp) (void)hipFree(d_temp);
    for (int b = 0; b < 2; b++) {
        if (d_key[b]) (void)hipFree(d_key[b]);
        if (d_val[b]) (void)hipFree(d_val[b]);
    }
    if (d_coo) (void)hipFree(d_coo);
    return rc;
}

// out[i] = in[0] + ... + in[i-1] on `st` (rocPRIM device scan; its temporary storage is freed before return)
int device_exclusive_scan(const int *d_in, int *d_out, size_t n, hipStream_t st)
{
    if (n == 0) return 0;
    size_t temp_bytes = 0;
    void *d_temp = nullptr;
    SPMV_HIP_TRY(rocprim::exclusive_scan(nullptr, temp_bytes, d_in, d_out, 0, n, rocprim::plus<int>(), st), return 1);
    SPMV_HIP_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)), return 1);
    int rc = 0;
    SPMV_HIP_TRY(rocprim::exclusive_scan(d_temp, temp_bytes, d_in, d_out, 0, n, rocprim::plus<int>(), st), rc = 1);
    if (!rc) SPMV_HIP_TRY(hipStreamSynchronize(st), rc = 1);
    (void)hipFree(d_temp);
    return rc;
}

} // namespace spmv

// C-linkage hook for the host-only ingest file (mtxio.cpp is compiled without HIP headers)
extern "C" __attribute__((visibility("hidden"))) int spmv_hip_internal_device_sort_coo(long long nnz, struct mtx *coo)
{
    return spmv::device_sort_coo((int64_t)nnz, coo);
}
