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

// `map` (option "block_map"): 0 = one contiguous eighth of the blocks per XCD (above); 1 = logical block =
// hardware block id (consecutive workgroups, which land on different XCDs, walk consecutive pieces of the
// matrix); K > 1 = runs of K consecutive logical blocks dealt over the XCDs in turn (the blocks past the last
// full round of 8 K keep their own number).  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nblocks, int map)
{
    if (map == 0) return xcd_contiguous_block(bid, nblocks);
    const unsigned k = (unsigned)map, round = 8u * k;
    if (k == 1 || bid >= nblocks / round * round) return bid;
    const unsigned g = bid & 7u, idx = bid >> 3;          // XCD, and the block's number among that XCD's blocks
    return (idx / k * 8u + g) * k + idx % k;
}

} // namespace spmv
