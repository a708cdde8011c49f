// Internal, device side: the block-level pieces of the exclusive scans (nb_scan.hip) and of the kernels that fuse a scan with
// the predicate in front of it and the numbering behind it (encoder index sets).
#pragma once
#include <hip/hip_runtime.h>

namespace nbscan {

constexpr int BLOCK = 256;
constexpr int ITEMS = 4;
constexpr int TILE = BLOCK * ITEMS;  // 1024 elements per block
constexpr int FUSED_MAX_BLOCKS = 4096;  // up to here every block sums the block totals in front of it itself (<= 16 KiB from L2)

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}

// inclusive scan of one value per thread across a 256-thread block; returns the exclusive prefix of the thread and the block
// total through `total`
__device__ __forceinline__ int block_excl_scan(int v, int *total) {
    __shared__ int wsum[BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) {
        if (i < w) base += wsum[i];
        tot += wsum[i];
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// sum of block_sums[0 .. b) by the whole block (the exclusive prefix of block b without a pass over the block totals)
__device__ __forceinline__ int blocks_before(const int *__restrict__ block_sums, int b) {
    int s = 0;
    for (int i = threadIdx.x; i < b; i += BLOCK) s += block_sums[i];
    int tot;
    block_excl_scan(s, &tot);
    return tot;
}

}  // namespace nbscan
