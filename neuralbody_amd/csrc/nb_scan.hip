// Device-wide exclusive scan of int32 flags (two small kernels, three beyond 4 M elements), used for deterministic
// stream compaction in the encoder (active-set numbering) and in ray generation
// (mask_at_box compaction, lib/utils/render_utils.py:128-132).
#include "nb_scan.h"

#include "nb_scan_dev.h"

namespace {

using nbscan::block_excl_scan;
constexpr int SCAN_BLOCK = nbscan::BLOCK;
constexpr int SCAN_ITEMS = nbscan::ITEMS;
constexpr int SCAN_TILE = nbscan::TILE;

__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(const int *__restrict__ in, long long n,
                                                                 int *__restrict__ block_sums) {
    const long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += in[base + i];
    int tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of the block sums in place, grand total to *total
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tops_kernel(int *__restrict__ block_sums, int n_blocks,
                                                               int *__restrict__ total) {
    int carry = 0;
    for (int base = 0; base < n_blocks; base += SCAN_BLOCK) {
        const int i = base + threadIdx.x;
        const int v = i < n_blocks ? block_sums[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const int *__restrict__ in, long long n,
                                                                const int *__restrict__ block_sums,
                                                                int *__restrict__ out) {
    const long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_excl_scan(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
}

// scan_apply_kernel for up to FUSED_MAX_BLOCKS blocks: every block sums the totals in front of it itself, the last one writes the
// grand total — no pass over the block totals, one launch less in every index-set chain
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_self_kernel(const int *__restrict__ in, long long n,
                                                                     const int *__restrict__ block_sums, int *__restrict__ out,
                                                                     int *__restrict__ total) {
    const int before = nbscan::blocks_before(block_sums, blockIdx.x);
    const long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    int tot;
    int ex = block_excl_scan(s, &tot) + before;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = before + tot;
}

}  // namespace

long long nb_scan_blocks(long long n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

int nb_exclusive_scan(const int *flags, int *out, int *total, long long n, int *block_sums, hipStream_t st) {
    const int nb = (int)nb_scan_blocks(n);
    if (nb == 0) {
        NB_HIP(hipMemsetAsync(total, 0, sizeof(int), st));
        return NB_OK;
    }
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, st, flags, n, block_sums);
    if (nb <= nbscan::FUSED_MAX_BLOCKS) {
        hipLaunchKernelGGL(scan_apply_self_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, st, flags, n, block_sums, out, total);
        NB_CHECK_LAUNCH("nb_exclusive_scan");
        return NB_OK;
    }
    hipLaunchKernelGGL(scan_tops_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, block_sums, nb, total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, st, flags, n, block_sums, out);
    NB_CHECK_LAUNCH("nb_exclusive_scan");
    return NB_OK;
}

extern "C" int64_t nb_scan_scratch_size(int64_t n) {
    if (n < 0) n = 0;
    // [flags n][positions n][block sums]  (int32), 256-byte aligned sections
    const int64_t a = ((n * 4 + 255) / 256) * 256;
    const int64_t b = ((nb_scan_blocks(n) * 4 + 255) / 256) * 256 + 256;
    return 2 * a + b;
}

void nb_scan_carve(void *scratch, long long n, int **flags, int **pos, int **block_sums) {
    const long long a = ((n * 4 + 255) / 256) * 256;
    char *p = static_cast<char *>(scratch);
    *flags = reinterpret_cast<int *>(p);
    *pos = reinterpret_cast<int *>(p + a);
    *block_sums = reinterpret_cast<int *>(p + 2 * a);
}
