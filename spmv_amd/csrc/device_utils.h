// Device helpers shared by the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>

namespace spmv {

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Workgroups b, b+8, b+16, ... share an XCD.  Give XCD-group g the g-th
// contiguous run of logical blocks.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned bid, unsigned nblocks)
{
    unsigned q = nblocks >> 3, r = nblocks & 7u, g = bid & 7u, idx = bid >> 3;
    unsigned base = g < r ? g * (q + 1) : r * (q + 1) + (g - r) * q;
    return base + idx;
}

} // namespace spmv
