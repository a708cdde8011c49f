// nb_encoder.hip — structured-latent-code encoder for gfx950: a from-scratch replacement of the
// spconv v1.2.1 calls made by SparseConvNet (zju3dv/neuralbody lib/networks/latent_xyzc.py:166-274).
//
// Data structure: a sparse tensor is a compact row matrix [n_rows, C] (fp32) + a dense int32
// INDEX GRID [D,H,W] holding the row id of every voxel (-1 = inactive) + the linear voxel index
// of every row.  With <=7 M voxels at full resolution the grid is 28 MB — trivial next to 288 GB
// of HBM — and turns spconv's hash-table rulebook into one coalescible int32 load per neighbour.
//
// Convolution: each WAVE owns 32 output rows x all Cout channels.  For every kernel offset with at
// least one active neighbour in the tile (wave-uniform skip otherwise) it gathers the 32
// neighbour rows (each half-wave fetches one half of the channels with 16-byte loads) and runs
// v_mfma_f32_32x32x2_f32 with the gathered activations as A and the weight slab W[o] (read in its
// native spconv [kD,kH,kW,Cin,Cout] layout, 128-byte coalesced) as B.  Exact fp32 — BatchNorm with
// batch statistics follows every conv, so the encoder is not a place to drop precision.
// Per-channel sum / sum-of-squares of the conv output are accumulated in fp64 (one atomic per
// channel per wave) for the BatchNorm that follows.
//
// Semantics (spconv source is not available offline -> "parity unpinned" against
// spconv itself, pinned against oracle/spconv_standin.py):
//   SubMConv3d(k=3):           out[p] = sum_k W[k] . in[p + k - 1], active set unchanged
//   SparseConv3d(k=3,s=2,p=1): out[o] = sum_k W[k] . in[2 o - 1 + k], active where any input is
//   BatchNorm1d(eps=1e-3) over ACTIVE rows, then ReLU;  .dense() -> zeros at inactive sites
//   duplicate vertex coordinates: the LAST vertex wins; BN counts unique voxels
#include "nb_scan.h"
#include "nb_scan_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

struct Dims {
    int d, h, w;
};

__host__ __device__ constexpr int tile_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ------------------------------------------------------------------ voxelisation
__global__ void vox_scatter_kernel(const int *__restrict__ coord, int n, Dims g, int *__restrict__ grid) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int d = coord[v * 3], h = coord[v * 3 + 1], w = coord[v * 3 + 2];
    if ((unsigned)d >= (unsigned)g.d || (unsigned)h >= (unsigned)g.h || (unsigned)w >= (unsigned)g.w) return;
    atomicMax(&grid[((long long)d * g.h + h) * g.w + w], v);  // last vertex wins
}

// Winners (the vertex a voxel kept) flagged and counted per 1024-vertex tile, then numbered in vertex order: the exclusive scan
// folded into the kernels on either side of it (each block of the second sums the tile counts in front of it itself).  Two
// kernels because the numbering overwrites the grid cells the flags are read from.
__global__ __launch_bounds__(nbscan::BLOCK) void vox_flag_count_kernel(const int *__restrict__ coord, int n, Dims g,
                                                                       const int *__restrict__ grid, int *__restrict__ flags,
                                                                       int *__restrict__ block_sums) {
    const int v0 = blockIdx.x * nbscan::TILE + threadIdx.x * nbscan::ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i) {
        const int v = v0 + i;
        if (v >= n) break;
        const int d = coord[v * 3], h = coord[v * 3 + 1], w = coord[v * 3 + 2];
        int f = 0;
        if ((unsigned)d < (unsigned)g.d && (unsigned)h < (unsigned)g.h && (unsigned)w < (unsigned)g.w)
            f = grid[((long long)d * g.h + h) * g.w + w] == v;
        flags[v] = f;
        s += f;
    }
    int tot;
    nbscan::block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(nbscan::BLOCK) void vox_number_kernel(const int *__restrict__ coord, int n, Dims g,
                                                                   const int *__restrict__ flags, const int *__restrict__ block_sums,
                                                                   int *__restrict__ grid, int *__restrict__ rows_vert,
                                                                   int *__restrict__ rows_lin, int *__restrict__ n_rows) {
    const int before = nbscan::blocks_before(block_sums, blockIdx.x);
    const int v0 = blockIdx.x * nbscan::TILE + threadIdx.x * nbscan::ITEMS;
    int f[nbscan::ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i) {
        f[i] = v0 + i < n ? flags[v0 + i] : 0;
        s += f[i];
    }
    int tot;
    int r = nbscan::block_excl_scan(s, &tot) + before;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i) {
        if (!f[i]) continue;
        const int v = v0 + i;
        const int d = coord[v * 3], h = coord[v * 3 + 1], w = coord[v * 3 + 2];
        const int lin = (d * g.h + h) * g.w + w;
        rows_vert[r] = v;
        rows_lin[r] = lin;
        grid[lin] = r;
        ++r;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_rows = before + tot;
}

// ------------------------------------------------------------------ strided-conv output index set
__global__ void down_mark_kernel(const int *__restrict__ in_lin, const int *__restrict__ n_in, Dims gi, Dims go,
                                 int *__restrict__ out_grid) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= *n_in) return;
    const int lin = in_lin[r];
    const int x = lin % gi.w, y = (lin / gi.w) % gi.h, z = lin / (gi.w * gi.h);
    // input i feeds outputs o with 2o-1 <= i <= 2o+1:  o = i/2, and (i+1)/2 when i is odd
    const int oz[2] = {z >> 1, (z + 1) >> 1}, oy[2] = {y >> 1, (y + 1) >> 1}, ox[2] = {x >> 1, (x + 1) >> 1};
    for (int a = 0; a < 1 + (z & 1); ++a)
        for (int b = 0; b < 1 + (y & 1); ++b)
            for (int c = 0; c < 1 + (x & 1); ++c)
                if (oz[a] < go.d && oy[b] < go.h && ox[c] < go.w)
                    out_grid[((long long)oz[a] * go.h + oy[b]) * go.w + ox[c]] = 0;  // mark (any value >= 0)
}

// Marked cells counted per 1024-cell tile, then numbered in linear order (row id into the cell, the cell into out_lin, the count
// clamped to the capacity): the scan folded into its neighbours as above; a cell is read and rewritten by one thread only.
__global__ __launch_bounds__(nbscan::BLOCK) void grid_count_kernel(const int *__restrict__ grid, long long n,
                                                                   int *__restrict__ block_sums) {
    const long long i0 = (long long)blockIdx.x * nbscan::TILE + threadIdx.x * nbscan::ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i)
        if (i0 + i < n) s += grid[i0 + i] >= 0;
    int tot;
    nbscan::block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(nbscan::BLOCK) void grid_number_kernel(int *__restrict__ grid, long long n,
                                                                    const int *__restrict__ block_sums, int cap,
                                                                    int *__restrict__ out_lin, int *__restrict__ n_out) {
    const int before = nbscan::blocks_before(block_sums, blockIdx.x);
    const long long i0 = (long long)blockIdx.x * nbscan::TILE + threadIdx.x * nbscan::ITEMS;
    int f[nbscan::ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i) {
        f[i] = i0 + i < n ? grid[i0 + i] >= 0 : 0;
        s += f[i];
    }
    int tot;
    int r = nbscan::block_excl_scan(s, &tot) + before;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i) {
        if (!f[i]) continue;
        grid[i0 + i] = r < cap ? r : -1;
        if (r < cap) out_lin[r] = (int)(i0 + i);
        ++r;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = min(before + tot, cap);
}

// ------------------------------------------------------------------ the index sets of ALL strided levels in three launches
// A cell c of the l-th level below a voxelised base level (k = 3, s = 2, p = 1 each) is active iff an active base voxel p lies within
// [2^l c - (2^l - 1), 2^l c + (2^l - 1)] in every coordinate — the per-level rule of down_mark_kernel composed — so every base voxel
// marks its (at most 8) cells on every level at once, and one count and one numbering launch walk the tiles of all levels' grids.
struct DownAll {
    int n_levels;
    Dims dims[NB_DOWN_LEVELS_MAX];
    int *grid[NB_DOWN_LEVELS_MAX], *out_lin[NB_DOWN_LEVELS_MAX], *n_out[NB_DOWN_LEVELS_MAX];
    int cap[NB_DOWN_LEVELS_MAX];
    long long nvox[NB_DOWN_LEVELS_MAX];
    int tile0[NB_DOWN_LEVELS_MAX + 1];  // first tile (= block) of each level's grid in the count / numbering launches
};

__global__ void down_mark_all_kernel(const int *__restrict__ in_lin, const int *__restrict__ n_in, Dims gi, DownAll a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= *n_in) return;
    const int lin = in_lin[r];
    const int x = lin % gi.w, y = (lin / gi.w) % gi.h, z = lin / (gi.w * gi.h);
    for (int l = 0; l < a.n_levels; ++l) {
        const int sh = l + 1, up = (1 << sh) - 1;  // cells floor(p / 2^sh) and floor((p + 2^sh - 1) / 2^sh)
        const Dims go = a.dims[l];
        const int oz[2] = {z >> sh, (z + up) >> sh}, oy[2] = {y >> sh, (y + up) >> sh}, ox[2] = {x >> sh, (x + up) >> sh};
        for (int i = 0; i < 1 + (oz[1] != oz[0]); ++i)
            for (int j = 0; j < 1 + (oy[1] != oy[0]); ++j)
                for (int k = 0; k < 1 + (ox[1] != ox[0]); ++k)
                    if (oz[i] < go.d && oy[j] < go.h && ox[k] < go.w)
                        a.grid[l][((long long)oz[i] * go.h + oy[j]) * go.w + ox[k]] = 0;  // mark (any value >= 0)
    }
}

__device__ __forceinline__ int down_all_level(const DownAll &a) {  // (block-uniform) the level whose grid this block's tile is in
    int l = 0;
    while (l + 1 < a.n_levels && (int)blockIdx.x >= a.tile0[l + 1]) ++l;
    return l;
}

__global__ __launch_bounds__(nbscan::BLOCK) void grid_count_all_kernel(DownAll a, int *__restrict__ block_sums) {
    const int l = down_all_level(a);
    const int *grid = a.grid[l];
    const long long n = a.nvox[l], i0 = (long long)((int)blockIdx.x - a.tile0[l]) * nbscan::TILE + threadIdx.x * nbscan::ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i)
        if (i0 + i < n) s += grid[i0 + i] >= 0;
    int tot;
    nbscan::block_excl_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(nbscan::BLOCK) void grid_number_all_kernel(DownAll a, const int *__restrict__ block_sums) {
    const int l = down_all_level(a), tile = (int)blockIdx.x - a.tile0[l];
    int *grid = a.grid[l];
    const long long n = a.nvox[l], i0 = (long long)tile * nbscan::TILE + threadIdx.x * nbscan::ITEMS;
    const int cap = a.cap[l];
    const int before = nbscan::blocks_before(block_sums + a.tile0[l], tile);
    int f[nbscan::ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i) {
        f[i] = i0 + i < n ? grid[i0 + i] >= 0 : 0;
        s += f[i];
    }
    int tot;
    int r = nbscan::block_excl_scan(s, &tot) + before;
#pragma unroll
    for (int i = 0; i < nbscan::ITEMS; ++i) {
        if (!f[i]) continue;
        grid[i0 + i] = r < cap ? r : -1;
        if (r < cap) a.out_lin[l][r] = (int)(i0 + i);
        ++r;
    }
    if ((int)blockIdx.x == a.tile0[l + 1] - 1 && threadIdx.x == 0) *a.n_out[l] = min(before + tot, cap);
}

// Row of the input level under kernel offset o of output voxel (z, y, x), or -1.  stride > 0: the forward gather, input voxel
// = stride * out - 1 + k.  stride < 0: the TRANSPOSED gather of a layer of stride -stride (its backward-input product as a convolution
// of its own, offsets already mirrored by nb_enc_conv_pack16 mode 1): "input" voxel = (out - 1 + k) / -stride where that divides.
__device__ __forceinline__ int neighbour_row(const int *__restrict__ in_grid, Dims gi, int z, int y, int x, int o, int stride, bool valid) {
    const int kd = o / 9, kh = (o / 3) % 3, kw = o % 3;
    const int mul = stride > 0 ? stride : 1, low = stride > 0 ? 0 : -stride - 1, sh = stride > 0 ? 0 : (-stride) >> 1;  // -stride in {1, 2}
    int iz = z * mul - 1 + kd, iy = y * mul - 1 + kh, ix = x * mul - 1 + kw;
    if (!valid || ((iz | iy | ix) & low) != 0 || iz < 0 || iy < 0 || ix < 0) return -1;
    iz >>= sh, iy >>= sh, ix >>= sh;
    if (iz >= gi.d || iy >= gi.h || ix >= gi.w) return -1;
    return in_grid[((long long)iz * gi.h + iy) * gi.w + ix];
}

// ------------------------------------------------------------------ a tile's way out
// The tile is stored; its BatchNorm sums (fp64: sum and sum of squares per channel) meet those of the workgroup's other waves in LDS
// and leave as ONE atomic per channel and workgroup.  As one pair of atomics per WAVE the ~900 waves of a 29 k-row level queued on
// the layer's 2 COUT addresses: 14 of the 52 us of a 64 -> 64 launch (profiles/r05_conv_stamps.log).  Every wave of the workgroup
// calls this (a wave without rows brings zeros; `active` false: a wave beyond the NW that hold tiles, for the barrier only); `red`: NW * NT * 64
// doubles of LDS that nobody reads or writes any more.
template <int COUT, int NT, int NW>
__device__ __forceinline__ void store_tile_and_sums(const f32x16 (&acc)[NT], int row0, int n, int ct, float *__restrict__ out_rows,
                                                    double *__restrict__ stats, double *red, int wv, int lane,
                                                    bool active = true) {
    const int i = lane & 31, hi = lane >> 5;
    // D fragment: lane (j = i, hi) holds channel (ct + t) * 32 + j of rows row0 + tile_row(r, hi)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (!active) break;  // (wave-uniform) a wave that only keeps the barrier company
        const int co = (ct + t) * 32 + i;
        const bool cok = (COUT % 32 == 0) || co < COUT;
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = row0 + tile_row(r, hi);
            const float v = acc[t][r];
            if (orow < n && cok) {
                out_rows[(size_t)orow * COUT + co] = v;
                s += (double)v;
                ss += (double)v * (double)v;
            }
        }
        s += __shfl_xor(s, 32);
        ss += __shfl_xor(ss, 32);
        if (hi == 0) {
            red[((wv * NT + t) * 2) * 32 + i] = s;
            red[((wv * NT + t) * 2 + 1) * 32 + i] = ss;
        }
    }
    __syncthreads();
    if (threadIdx.x < NT * 64) {  // thread = (tile, which sum, channel of the tile)
        const int t = threadIdx.x >> 6, which = (threadIdx.x >> 5) & 1, co = (ct + t) * 32 + (threadIdx.x & 31);
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[((w * NT + t) * 2 + which) * 32 + (threadIdx.x & 31)];
        if ((COUT % 32 == 0) || co < COUT) atomicAdd(&stats[which * COUT + co], v);
    }
}

// ------------------------------------------------------------------ sparse 3x3x3 convolution
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv_kernel(const float *__restrict__ in_rows, const int *__restrict__ in_grid,
                                                   Dims gi, const int *__restrict__ out_lin,
                                                   const int *__restrict__ n_out, Dims go, int stride,
                                                   const float *__restrict__ weight, float *__restrict__ out_rows,
                                                   double *__restrict__ stats) {
    // one wave = 32 output rows x 32 output channels (blockIdx.y selects the channel tile): the deep levels have
    // few rows (1.6 k - 13 k), so splitting Cout across waves is what fills the 1024 SIMDs
    constexpr int NT = 1, HALF = CIN / 2;
    const int ct = blockIdx.y;  // channel tile
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = *n_out;
    const int row0 = wave * 32;
    if (blockIdx.x * 128 >= n) return;  // workgroup-uniform: the waves of a live workgroup meet at the barrier of the sums
    const int row = row0 + i;
    const bool valid = row < n;
    const int lin = valid ? out_lin[row] : 0;
    const int x = lin % go.w, y = (lin / go.w) % go.h, z = lin / (go.w * go.h);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // all 27 neighbour indices first (independent loads), then the rows of offset o + 1 in flight while offset o multiplies: as a
    // chain of grid lookup -> row load -> MFMA per offset the 16-channel layers ran at 0.7 us per offset for 0.25 us of MFMAs
    int nbrs[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        nbrs[o] = neighbour_row(in_grid, gi, z, y, x, o, stride, valid);
    }
    f32x4 Ar[2][HALF / 4];
    auto load_rows = [&](int nbr, f32x4 (&dst)[HALF / 4]) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(in_rows + (size_t)(nbr >= 0 ? nbr : 0) * CIN + hi * HALF);
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) dst[q] = p[q];
    };
    load_rows(nbrs[0], Ar[0]);
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        if (o + 1 < 27) load_rows(nbrs[o + 1], Ar[(o + 1) & 1]);
        const int nbr = nbrs[o];
        if (!__any(nbr >= 0)) continue;  // nothing active under this offset for the whole tile
        float A[HALF];
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) {
            const f32x4 v = Ar[o & 1][q];
            A[4 * q] = nbr >= 0 ? v.x : 0.f;
            A[4 * q + 1] = nbr >= 0 ? v.y : 0.f;
            A[4 * q + 2] = nbr >= 0 ? v.z : 0.f;
            A[4 * q + 3] = nbr >= 0 ? v.w : 0.f;
        }
        const float *wo = weight + ((size_t)o * CIN + hi * HALF) * COUT + ct * 32 + i;  // B[k=hi][j=i]
#pragma unroll
        for (int c = 0; c < HALF; ++c) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = (ct + t) * 32 + i;
                const float b = (COUT % 32 == 0 || co < COUT) ? wo[(size_t)c * COUT + t * 32] : 0.f;
                acc[t] = NB_MFMA(A[c], b, acc[t]);
            }
        }
    }
    __shared__ double red[4 * NT * 64];
    store_tile_and_sums<COUT, NT, 4>(acc, row0, n, ct, out_rows, stats, red, threadIdx.x >> 6, lane);
}

// ------------------------------------------------------------------ the same convolution on the 16-bit matrix pipe
// Inference path (no backward record is kept): every operand as fp16 head + fp16 remainder and three products per K chunk
// (A_hi.B_hi + A_hi.B_lo + A_lo.B_hi, fp32 accumulate) on v_mfma_f32_32x32x16_f16 — 8x the K per instruction and half the
// cycles of v_mfma_f32_32x32x2_f32, and 16-byte operand loads instead of one float per lane per MFMA.  fp16, not bf16: the
// operands are BatchNorm outputs and weights, O(1), so the 22 mantissa bits of an fp16 pair (relative error ~2^-21 per
// product) come for free where a bf16 pair has 16; the first version used bf16 and drifted 2.2e-4 from the oracle over the
// 17 layers.  The activated rows arrive ALREADY split from the producing BatchNorm kernel (two fp16 planes in the bytes of
// one fp32 row matrix: [cap, C] heads | [cap, C] remainders), the weights from nb_enc_conv_pack16 in B-fragment order:
// [offset][K chunk][channel tile][head, remainder][lane] x 8 fp16.
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));  // (the name predates the switch to fp16)
typedef _Float16 nb_h16;
// BF: the operands are bf16 head / remainder pairs (the backward-input convolution: gradients span more binades than an
// un-scaled fp16 head holds; a bf16 pair carries 16 mantissa bits, ~2^-16 relative per product) instead of fp16 pairs
typedef __bf16 nb_bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__device__ __forceinline__ f32x16 nb_mfma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nb_bf16x8, a), __builtin_bit_cast(nb_bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
#define NB_MFMA16(a, b, c) nb_mfma16<BF>((a), (b), (c))

// mode 0: the forward weight [27][cin][cout] as fp16 pairs.  mode 1: the weight of the BACKWARD-INPUT convolution of a stride-1
// layer — dIn[q] = sum_o dOut[q + (o' - 1)] . W[26 - o']^T, i.e. the same kernels on the mirrored offsets and the transposed
// slabs — as bf16 pairs: cin / cout are those of the packed convolution (= the layer's cout / cin), w is the layer's weight
__global__ void conv_pack16_kernel(const float *__restrict__ w, int cin, int cout, bf16x8 *__restrict__ out, int mode) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (((o * NC + c) * NTT + t) * 2 + part) * 64 + lane
    const int nc = cin / 16, ntt = cout / 32;
    if (idx >= (long long)27 * nc * ntt * 2 * 64) return;
    const int lane = (int)(idx & 63), part = (int)((idx >> 6) & 1);
    const long long q = idx >> 7;
    const int t = (int)(q % ntt), c = (int)((q / ntt) % nc), o = (int)(q / ((long long)ntt * nc));
    const int j = lane & 31, hi = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * c + 8 * hi + e, n = 32 * t + j;  // input channel, output channel of the packed convolution
        if (mode == 0) {
            const float x = w[((size_t)o * cin + k) * cout + n];
            const nb_h16 h = (nb_h16)x;
            v[e] = part ? (nb_h16)(x - (float)h) : h;
        } else {
            const float x = w[((size_t)(26 - o) * cout + n) * cin + k];
            const __bf16 h = (__bf16)x;
            const __bf16 r = part ? (__bf16)(x - (float)h) : h;
            v[e] = __builtin_bit_cast(nb_h16, r);
        }
    }
    out[idx] = v;
}

// the same for up to NB_PACK_BATCH_MAX (weight, mode) jobs in one launch: blockIdx.y = the job, blockIdx.x over its elements
struct PackBatch {
    const float *w[NB_PACK_BATCH_MAX];
    bf16x8 *out[NB_PACK_BATCH_MAX];
    int cin[NB_PACK_BATCH_MAX], cout[NB_PACK_BATCH_MAX], mode[NB_PACK_BATCH_MAX];
};
__global__ void conv_pack16_batch_kernel(PackBatch b) {
    const int job = blockIdx.y;
    const float *__restrict__ w = b.w[job];
    bf16x8 *__restrict__ out = b.out[job];
    const int cin = b.cin[job], cout = b.cout[job], mode = b.mode[job];
    const int nc = cin / 16, ntt = cout / 32;
    const long long total = (long long)27 * nc * ntt * 2 * 64;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63), part = (int)((idx >> 6) & 1);
        const long long q = idx >> 7;
        const int t = (int)(q % ntt), c = (int)((q / ntt) % nc), o = (int)(q / ((long long)ntt * nc));
        const int j = lane & 31, hi = lane >> 5;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * c + 8 * hi + e, n = 32 * t + j;
            if (mode == 0) {
                const float x = w[((size_t)o * cin + k) * cout + n];
                const nb_h16 h = (nb_h16)x;
                v[e] = part ? (nb_h16)(x - (float)h) : h;
            } else {
                const float x = w[((size_t)(26 - o) * cout + n) * cin + k];
                const __bf16 h = (__bf16)x;
                const __bf16 r = part ? (__bf16)(x - (float)h) : h;
                v[e] = __builtin_bit_cast(nb_h16, r);
            }
        }
        out[idx] = v;
    }
}

// one wave = 32 output rows x NT tiles of 32 output channels (blockIdx.y selects the tile group)
template <int CIN, int COUT, int NT, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv16_kernel(const unsigned short *__restrict__ in_split, long long in_plane,
                                                     const int *__restrict__ in_grid, Dims gi, const int *__restrict__ out_lin,
                                                     const int *__restrict__ n_out, Dims go, int stride,
                                                     const bf16x8 *__restrict__ wp, float *__restrict__ out_rows,
                                                     double *__restrict__ stats) {
    constexpr int NC = CIN / 16, NTT = COUT / 32;
    const int ct = blockIdx.y * NT;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = *n_out;
    const int row0 = wave * 32;
    if (blockIdx.x * 128 >= n) return;  // workgroup-uniform: the waves of a live workgroup meet at the barrier of the sums
    const int row = row0 + i;
    const bool valid = row < n;
    const int lin = valid ? out_lin[row] : 0;
    const int x = lin % go.w, y = (lin / go.w) % go.h, z = lin / (go.w * go.h);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // The wave has ~1.5 companions per SIMD on the deep levels, so nothing hides a dependent load chain (index -> row ->
    // MFMA) but the wave itself: all 27 neighbour indices are fetched first, and the rows of offset o + 1 while offset o is
    // multiplied (the fp32 kernel above pays ~9 k cycles per offset for what is 0.8 k cycles of MFMA work here).
    int nbrs[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        nbrs[o] = neighbour_row(in_grid, gi, z, y, x, o, stride, valid);
    }
    // A fragments: lane (row i, half hi) holds channels 16 c + 8 hi .. + 7 of its neighbour row (zeros when inactive)
    auto load_rows = [&](int nbr, bf16x8 (&ah)[NC], bf16x8 (&al)[NC]) {
        const size_t r = (size_t)(nbr >= 0 ? nbr : 0) * CIN + 8 * hi;
        const bf16x8 *ph = reinterpret_cast<const bf16x8 *>(in_split + r);
        const bf16x8 *pl = reinterpret_cast<const bf16x8 *>(in_split + in_plane + r);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ah[c] = ph[2 * c];
            al[c] = pl[2 * c];
        }
    };
    bf16x8 ah[2][NC], al[2][NC];
    load_rows(nbrs[0], ah[0], al[0]);
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        if (o + 1 < 27) load_rows(nbrs[o + 1], ah[(o + 1) & 1], al[(o + 1) & 1]);
        const int nbr = nbrs[o];
        if (!__any(nbr >= 0)) continue;  // nothing active under this offset for the whole tile
        const bf16x8 *wo = wp + (((size_t)o * NC * NTT + ct) * 2) * 64 + lane;
        bf16x8 zero;
#pragma unroll
        for (int e = 0; e < 8; ++e) zero[e] = (nb_h16)0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bf16x8 a_h = nbr >= 0 ? ah[o & 1][c] : zero, a_l = nbr >= 0 ? al[o & 1][c] : zero;  // row 0 was read for inactive lanes
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 bh = wo[((size_t)c * NTT + t) * 128], bl = wo[((size_t)c * NTT + t) * 128 + 64];
                acc[t] = NB_MFMA16(a_h, bh, acc[t]);
                acc[t] = NB_MFMA16(a_h, bl, acc[t]);
                acc[t] = NB_MFMA16(a_l, bh, acc[t]);
            }
        }
    }
    __shared__ double red[4 * NT * 64];
    store_tile_and_sums<COUT, NT, 4>(acc, row0, n, ct, out_rows, stats, red, threadIdx.x >> 6, lane);
}

typedef const void __attribute__((address_space(1))) *nb_gptr_t;
typedef void __attribute__((address_space(3))) *nb_lptr_t;

// ------------------------------------------------------------------ a wave's 32 gathered rows through LDS
// Loaded straight into the A-fragment registers (lane (i, hi) reads 16 bytes of row i) a gather instruction has every lane on a
// different row and the texture-address path takes it one lane per clock: 16-18 bytes per clock and CU, measured
// (tools/experiments/probe_gather.hip, profiles/r05_probe_gather.log), whatever the rows' placement.  With FOUR ADJACENT LANES on 64
// contiguous bytes it delivers 40-52, so the rows come in by LDS-DMA in that shape — one global_load_lds_dwordx4 = 16 lane quads =
// 1 KiB landing lane-contiguous — and the fragments are read back with ds_read_b128.  A 128-channel row plane is 256 bytes = 4
// pieces of 64; instruction t of a plane carries the 16 pieces of 4 rows.  Which rows, and the 16 bytes of padding between
// instructions, are chosen so that the 16 lanes ds_read_b128 serves in one cycle ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same + 32) fall on 16 different 16-byte slots of the 256-byte bank row.  (64-channel planes were built the same way — 8 rows per
// instruction — and measured no faster than register loads in the pipeline; the kernel below stages 128-channel rows only.)
template <int CIN>
struct RowStage {
    static_assert(CIN == 128, "row stage: 256-byte row planes (128 channels)");
    static constexpr int ROWB = 2 * CIN, NI = ROWB / 32, BLOCK = 1024 + 16, BYTES = 2 * NI * BLOCK;
    int src_base, src_byte, frag_base;
    __device__ __forceinline__ explicit RowStage(int lane) {
        const int q = lane >> 2, j = lane & 3, i = lane & 31, hi = lane >> 5;
        // instruction t carries rows (t / 4) 16 + (t % 4) 4 .. + 3, lane quad = piece * 4 + row % 4
        src_base = q & 3;
        src_byte = (q >> 2) * 64 + j * 16;
        frag_base = ((i >> 4) * 4 + ((i & 15) >> 2)) * BLOCK + (i & 3) * 64 + hi * 16;
    }
    // rows nbr (lane i's neighbour under the offset; < 0: none, row 0 is fetched and the consumer zeroes it) -> the wave's stage, as
    // 2 NI instructions: addresses() once per offset, then piece(k), k = 2 t + plane, wherever the caller wants each issued
    __device__ __forceinline__ void addresses(const unsigned short *in_split, int nbr, const char *(&g)[NI]) const {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int src = src_base + (t >> 2) * 16 + (t & 3) * 4;
            const int row = __shfl(nbr, src);
            g[t] = reinterpret_cast<const char *>(in_split) + (size_t)(row >= 0 ? row : 0) * ROWB + src_byte;
        }
    }
    __device__ __forceinline__ void piece(int k, const char *const (&g)[NI], long long in_plane, char *mine) const {
        const int t = k >> 1, plane = k & 1;
        __builtin_amdgcn_global_load_lds((nb_gptr_t)(g[t] + plane * 2 * in_plane), (nb_lptr_t)(mine + (plane * NI + t) * BLOCK), 16, 0, 0);
    }
    __device__ __forceinline__ void fetch(const unsigned short *in_split, long long in_plane, int nbr, char *mine) const {
        const char *g[NI];
        addresses(in_split, nbr, g);
#pragma unroll
        for (int k = 0; k < 2 * NI; ++k) piece(k, g, in_plane, mine);
    }
    // chunk c of lane (i, hi): bytes [32 c + 16 hi, + 16) of row i = piece c / 2, slot 2 (c % 2) + hi of the quad
    template <typename V, int NC>
    __device__ __forceinline__ void fragments(const char *mine, V (&ah)[NC], V (&al)[NC]) const {
        constexpr int PIECE = 4 * 64;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const char *p = mine + frag_base + (c >> 1) * PIECE + (c & 1) * 32;
            ah[c] = *reinterpret_cast<const V *>(p);
            al[c] = *reinterpret_cast<const V *>(p + NI * BLOCK);
        }
    }
};

// One kernel offset of a wave's tile from an LDS slab ([chunk][tile][head, remainder] 1-KiB fragments): acc[t] += A_hi.B_hi + A_hi.B_lo
// + A_lo.B_hi over the NC K chunks, every accumulator in the order chunk, (hh, hl, lh).  These kernels run one wave per SIMD, so
// nothing fills a wait but the wave itself: chunk c + 1's fragments are read while chunk c multiplies, and the products go round the
// NT accumulators (a dependent MFMA only every NT-th issue slot).  As "read two fragments, wait, three MFMAs into one accumulator"
// the multiply phase of a 64 -> 64 offset stamped 1 900 cycles, like this 1 750 (24 MFMAs = 768 cycles of pipe; the rest is the
// issue of the next offset's loads against a busy texture path: profiles/r05_conv_stamps.log).
struct NoFetch {
    __device__ __forceinline__ void operator()(int) const {}
};
// VPC > 0: the caller's fetches of the NEXT offset (VPC vector-memory instructions per chunk, fetch(k), k = 0 .. NC VPC - 1) are issued
// BETWEEN this offset's MFMAs, one per 3 NT / VPC of them, each fenced by scheduling barriers: issued in a block before the MFMAs
// (where the scheduler puts them) the wave sits out their issue — 60-180 cycles apiece with the texture path busy — before its
// first MFMA, and the offset costs fetch + multiply instead of the larger of the two.
template <int NC, int NT, bool BF, int VPC = 0, typename Fetch = NoFetch>
__device__ __forceinline__ void multiply_offset(f32x16 (&acc)[NT], const bf16x8 (&ah)[NC], const bf16x8 (&al)[NC], bool has,
                                                const bf16x8 *sl, Fetch fetch = Fetch()) {
    static_assert(VPC == 0 || (3 * NT) % VPC == 0, "the chunk's MFMAs split evenly round its fetches");
    constexpr int GROUPS = VPC > 0 ? VPC : 1, PER = 3 * NT / GROUPS;
    bf16x8 b[2][NT][2], zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (nb_h16)0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        b[0][t][0] = sl[(t * 2) * 64];
        b[0][t][1] = sl[(t * 2 + 1) * 64];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                b[(c + 1) & 1][t][0] = sl[(((c + 1) * NT + t) * 2) * 64];
                b[(c + 1) & 1][t][1] = sl[(((c + 1) * NT + t) * 2 + 1) * 64];
            }
        }
        const bf16x8 a_h = has ? ah[c] : zero, a_l = has ? al[c] : zero;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
            for (int q = 0; q < PER; ++q) {  // MFMA m of the chunk: product m / NT (hh, hl, lh) into accumulator m % NT
                const int m = g * PER + q, prod = m / NT, t = m % NT;
                acc[t] = NB_MFMA16(prod < 2 ? a_h : a_l, b[c & 1][t][prod == 1 ? 1 : 0], acc[t]);
            }
            if constexpr (VPC > 0) {
                __builtin_amdgcn_sched_barrier(0);
                fetch(c * VPC + g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (VPC == 0) {  // the order for the scheduler (it otherwise sinks a chunk's reads behind the previous chunk's MFMAs)
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT, 0);
        }
    }
}

// conv16_kernel with the weight slab of the current kernel offset SHARED through LDS by the four waves of a workgroup (128 output
// rows x NT channel tiles): every B fragment is fetched from L2 once per workgroup instead of once per wave, by LDS-DMA (one 1-KiB
// fragment = one global_load_lds_dwordx4, no registers), double buffered, one barrier per offset; at 128 input channels the gathered
// rows come through the waves' LDS stages (RowStage; 130 KB of LDS with two channel tiles per wave).  The 64- and 128-channel
// layers (the per-wave kernel above moves 1.4 GB of operands per 128 -> 128 launch).  Round 4 ran the mid levels as 8 waves with the
// offsets split over two groups, rows straight into registers (its 128 KB of slabs left no room for stages): 69.6 us per
// 128 -> 128 launch against 51.9 for this one (profiles/r05_conv_fetch_ab.log).
template <int CIN, int COUT, int NT, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv16_lds_kernel(const unsigned short *__restrict__ in_split, long long in_plane,
                                                         const int *__restrict__ in_grid, Dims gi, const int *__restrict__ out_lin,
                                                         const int *__restrict__ n_out, Dims go, int stride,
                                                         const bf16x8 *__restrict__ wp, float *__restrict__ out_rows,
                                                         double *__restrict__ stats) {
    constexpr int NC = CIN / 16, NTT = COUT / 32;
    constexpr int NFRAG = NC * NT * 2;        // 1-KiB fragments of one offset's slab: [c][t][head, remainder]
    constexpr int PER_WAVE = NFRAG / 4;       // DMAs per wave per offset
    static_assert(NFRAG % 4 == 0, "slab must split evenly over the four waves");
    // measured in the pipeline (profiles/r05_conv_fetch_ab.log): 128-channel rows 51.9 us staged against 66.7 straight into registers
    // per 128 -> 128 launch; 64-channel rows 39.3 against 38.3 (64 -> 64) and 36.4 against 32.7 (64 -> 128): only the 256-byte planes
    constexpr bool STAGE = CIN == 128 && 2 * NFRAG * 1024 + 4 * RowStage<128>::BYTES <= 160 * 1024;
    typedef RowStage<128> RS;
    constexpr int STAGE_BYTES = STAGE ? RS::BYTES : 0;
    // ONE LDS object: with the row stages as a second __shared__ array hipcc puts an s_waitcnt vmcnt(0) between a DMA into one and
    // the next read of the other (the LDS-DMA bookkeeping of its wait-count pass), which serialises every offset's fetch and multiply
    __shared__ __attribute__((aligned(16))) char lds[2 * NFRAG * 1024 + 4 * STAGE_BYTES];
    char (*const slab)[NFRAG * 1024] = reinterpret_cast<char (*)[NFRAG * 1024]>(lds);
    const int ct = blockIdx.y * NT;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = *n_out;
    const int row0 = (blockIdx.x * 4 + wv) * 32;
    if (blockIdx.x * 128 >= n) return;  // workgroup-uniform: every wave of a live workgroup takes part in the barriers
    const int row = row0 + i;
    const bool valid = row < n;
    const int lin = valid ? out_lin[row] : 0;
    const int x = lin % go.w, y = (lin / go.w) % go.h, z = lin / (go.w * go.h);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto issue_slab_piece = [&](int o, int buf, int q) {  // piece q of this wave's share of offset o's fragments -> slab[buf]
        const int f = wv * PER_WAVE + q;           // local fragment (c, t, part) = ((c * NT) + t) * 2 + part
        const int c = f / (NT * 2), rest = f % (NT * 2);
        const bf16x8 *src = wp + (((size_t)o * NC + c) * NTT + ct) * 2 * 64 + (size_t)rest * 64 + lane;
        __builtin_amdgcn_global_load_lds((nb_gptr_t)src, (nb_lptr_t)(slab[buf] + f * 1024), 16, 0, 0);
    };
    auto issue_slab = [&](int o, int buf) {
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) issue_slab_piece(o, buf, q);
    };
    int nbrs[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        nbrs[o] = neighbour_row(in_grid, gi, z, y, x, o, stride, valid);
    }
    auto load_rows = [&](int nbr, bf16x8 (&ah)[NC], bf16x8 (&al)[NC]) {
        const size_t r = (size_t)(nbr >= 0 ? nbr : 0) * CIN + 8 * hi;
        const bf16x8 *ph = reinterpret_cast<const bf16x8 *>(in_split + r);
        const bf16x8 *pl = reinterpret_cast<const bf16x8 *>(in_split + in_plane + r);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ah[c] = ph[2 * c];
            al[c] = pl[2 * c];
        }
    };
    // STAGE: the wave's 32 neighbour rows reach it through its own LDS stage (RowStage), fetched while the previous offset
    // multiplies; else (no room beside the slabs) straight into a second set of fragment registers
    char *const mine = lds + 2 * NFRAG * 1024 + wv * STAGE_BYTES;
    const RS rs(lane);
    constexpr int NBUF = STAGE ? 1 : 2;
    bf16x8 ah[NBUF][NC], al[NBUF][NC];
    issue_slab(0, 0);
    if constexpr (STAGE) rs.fetch(in_split, in_plane, nbrs[0], mine);
    else load_rows(nbrs[0], ah[0], al[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        const int nbr = nbrs[o];
        const bool live = __any(nbr >= 0);
        const bf16x8 *sl = reinterpret_cast<const bf16x8 *>(slab[o & 1]) + lane;
        if constexpr (STAGE) {
            // one block per offset: the stage's fragments, then the next offset's fetches BETWEEN this offset's MFMAs
            constexpr int VPC = (PER_WAVE + 2 * RS::NI) / NC;
            static_assert(!STAGE || (PER_WAVE + 2 * RS::NI) % NC == 0, "fetches per chunk");
            if (live) {
                rs.fragments(mine, ah[0], al[0]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... before the next offset's rows overwrite the stage
                if (o + 1 < 27) {
                    const char *g[RS::NI];
                    rs.addresses(in_split, nbrs[o + 1], g);
                    // piece k of the next offset: first the wave's share of the slab (the other buffer: its last readers passed the
                    // barrier that ended offset o - 1), then the row planes
                    auto next_piece = [&](int k) {
                        if (k < PER_WAVE) issue_slab_piece(o + 1, (o + 1) & 1, k);
                        else rs.piece(k - PER_WAVE, g, in_plane, mine);
                    };
                    multiply_offset<NC, NT, BF, VPC>(acc, ah[0], al[0], nbr >= 0, sl, next_piece);
                } else {
                    multiply_offset<NC, NT, BF>(acc, ah[0], al[0], nbr >= 0, sl);
                }
            } else if (o + 1 < 27) {
                issue_slab(o + 1, (o + 1) & 1);
                rs.fetch(in_split, in_plane, nbrs[o + 1], mine);
            }
        } else {
            if (o + 1 < 27) {
                issue_slab(o + 1, (o + 1) & 1);
                load_rows(nbrs[o + 1], ah[(o + 1) & 1], al[(o + 1) & 1]);
            }
            if (live) multiply_offset<NC, NT, BF>(acc, ah[o & (NBUF - 1)], al[o & (NBUF - 1)], nbr >= 0, sl);
        }
        if (o + 1 < 27) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs of slab o + 1 (and the prefetched rows) have landed
            __syncthreads();                                  // ... everybody's have, and everybody is done reading slab o
        }
    }
    __syncthreads();  // the slabs are dead: their memory takes the sums
    store_tile_and_sums<COUT, NT, 4>(acc, row0, n, ct, out_rows, stats, reinterpret_cast<double *>(lds), wv, lane);
}

// The same product with the 27 kernel offsets SPLIT OVER THE NW WAVES of a workgroup (one 32-row x 32-channel tile per
// workgroup, wave w takes offsets w, w + NW, ...; partial tiles summed through LDS in wave order: deterministic).  On the deep
// levels a launch has a few hundred to a few thousand rows, i.e. 16-64 workgroups of the kernels above on 256 CUs, and each of
// them walks 27 offsets at one L2 round trip (index -> row -> MFMA, 2.5-4 us) apiece: 67-116 us per 128-channel layer for
// ~20 us of MFMA work per wave (profiles/r03_step_timeline.md).  Here the serial chain is ceil(27 / NW) offsets long and the
// operands of offset k + 1 are re-loaded into the registers offset k has just consumed (rows and weight fragments, chunk by
// chunk), so every load has a full offset of MFMAs to land.
template <int CIN, int COUT, int NW, bool BF = false>
__global__ __launch_bounds__(64 * NW) void conv16_ks_kernel(const unsigned short *__restrict__ in_split, long long in_plane,
                                                           const int *__restrict__ in_grid, Dims gi, const int *__restrict__ out_lin,
                                                           const int *__restrict__ n_out, Dims go, int stride,
                                                           const bf16x8 *__restrict__ wp, float *__restrict__ out_rows,
                                                           double *__restrict__ stats) {
    constexpr int NC = CIN / 16, NTT = COUT / 32;
    constexpr int MAXO = (27 + NW - 1) / NW;
    __shared__ __attribute__((aligned(16))) float red[NW][16][64];
    const int ct = blockIdx.y;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = *n_out;
    const int row0 = blockIdx.x * 32;
    if (row0 >= n) return;  // workgroup-uniform
    const int row = row0 + i;
    const bool valid = row < n;
    const int lin = valid ? out_lin[row] : 0;
    const int x = lin % go.w, y = (lin / go.w) % go.h, z = lin / (go.w * go.h);
    int nbrs[MAXO];
#pragma unroll
    for (int k = 0; k < MAXO; ++k) {
        const int o = wv + NW * k;
        nbrs[k] = neighbour_row(in_grid, gi, z, y, x, o < 27 ? o : 0, stride, o < 27 && valid);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 ah[NC], al[NC], bh[NC], bl[NC];
    auto load_a = [&](int nbr, int c) {
        const size_t r = (size_t)(nbr >= 0 ? nbr : 0) * CIN + 8 * hi + 16 * c;
        ah[c] = *reinterpret_cast<const bf16x8 *>(in_split + r);
        al[c] = *reinterpret_cast<const bf16x8 *>(in_split + in_plane + r);
    };
    auto load_b = [&](int o, int c) {
        const bf16x8 *w = wp + ((((size_t)o * NC + c) * NTT + ct) * 2) * 64 + lane;
        bh[c] = w[0];
        bl[c] = w[64];
    };
    if (wv < 27) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            load_a(nbrs[0], c);
            load_b(wv, c);
        }
    }
    bf16x8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (nb_h16)0.f;
#pragma unroll
    for (int k = 0; k < MAXO; ++k) {
        const int o = wv + NW * k;
        if (o >= 27) break;  // wave-uniform
        const int nbr = nbrs[k];
        const bool more = k + 1 < MAXO && o + NW < 27;
        const bool live = __any(nbr >= 0);  // else: nothing active under this offset for the whole tile
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (live) {
                const bf16x8 a_h = nbr >= 0 ? ah[c] : zero, a_l = nbr >= 0 ? al[c] : zero;  // row 0 was read for inactive lanes
                acc = NB_MFMA16(a_h, bh[c], acc);
                acc = NB_MFMA16(a_h, bl[c], acc);
                acc = NB_MFMA16(a_l, bh[c], acc);
            }
            if (more) {
                load_a(nbrs[k + 1 < MAXO ? k + 1 : k], c);
                load_b(o + NW, c);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
    // wave w finishes accumulator registers [w RPW, (w + 1) RPW) of the tile: D fragment, lane (j = i, hi) holds channel
    // ct * 32 + j of rows row0 + tile_row(r, hi)
    constexpr int RPW = 16 / NW;
    static_assert(16 % NW == 0, "accumulator registers split evenly over the waves");
    const int co = ct * 32 + i;
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wv * RPW + q;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][r][lane];
        const int orow = row0 + tile_row(r, hi);
        if (orow < n) {
            out_rows[(size_t)orow * COUT + co] = v;
            s += (double)v;
            ss += (double)v * (double)v;
        }
    }
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    // the waves' sums meet in LDS: one atomic per channel and workgroup (store_tile_and_sums)
    __shared__ double sred[NW][2][32];
    if (hi == 0) {
        sred[wv][0][i] = s;
        sred[wv][1][i] = ss;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += sred[w][hi][i];
        atomicAdd(&stats[hi * COUT + co], v);
    }
}

// ------------------------------------------------------------------ BatchNorm1d + ReLU (+ .dense())
// One block = 256 threads.  Every block first forms the C per-channel affine pairs (a, b) of y = relu(a x + b) in fp64 — from the
// batch sums (training) or the running statistics — into LDS (block 0 also does the layer's bookkeeping: batch_stats, running
// statistics); then four consecutive channels per thread and trip (one 16-byte load; the fp16 head / remainder planes leave as
// two 8-byte stores).  Round 4 evaluated the fp64 division and square root once per ELEMENT (13 us for a 29 k x 64 layer).
constexpr int BN_MAX_C = 256;
__global__ __launch_bounds__(256) void bn_relu_kernel(float *__restrict__ rows, const int *__restrict__ n_rows, int C,
                                                      const double *__restrict__ stats, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float *__restrict__ rmean,
                                                      float *__restrict__ rvar, int training, float eps, float momentum,
                                                      float *__restrict__ batch_stats, const int *__restrict__ rows_lin,
                                                      float *__restrict__ dense, float *__restrict__ rows_out,
                                                      _Float16 *__restrict__ split_out, long long split_plane) {
    __shared__ float sa[BN_MAX_C], sb[BN_MAX_C];
    const int n = *n_rows;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double mean, var;
        if (training) {
            mean = n > 0 ? stats[c] / n : 0.0;
            var = n > 0 ? fmax(stats[C + c] / n - mean * mean, 0.0) : 0.0;
        } else {
            mean = rmean[c];
            var = rvar[c];
        }
        const double invstd = 1.0 / sqrt(var + (double)eps);
        sa[c] = (float)(invstd * (double)gamma[c]);
        sb[c] = (float)((double)beta[c] - mean * invstd * (double)gamma[c]);
        if (blockIdx.x == 0 && batch_stats) {
            if (training && n > 0) {
                batch_stats[c] = (float)mean;
                batch_stats[C + c] = (float)var;
                if (momentum >= 0.f) {  // nn.BatchNorm1d bookkeeping: unbiased variance into running_var
                    const float unb = (float)(var * ((double)n / (double)(n > 1 ? n - 1 : 1)));
                    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
                    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
                }
            } else {
                batch_stats[c] = 0.f;
                batch_stats[C + c] = 0.f;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && batch_stats) batch_stats[2 * C] = (float)n;
    __syncthreads();
    const long long total4 = (long long)n * C / 4;  // C % 4 == 0
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (long long)gridDim.x * blockDim.x) {
        const long long idx = i4 * 4;
        const int c = (int)(idx % C);
        const f32x4 x = *reinterpret_cast<const f32x4 *>(rows + idx);
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = fmaxf(fmaf(x[k], sa[c + k], sb[c + k]), 0.f);
        if (split_out) {  // for nb_enc_conv16: fp16 head and remainder planes in the bytes of an fp32 row matrix
            h4 h, l;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h[k] = (_Float16)y[k];
                l[k] = (_Float16)(y[k] - (float)h[k]);
            }
            *reinterpret_cast<h4 *>(split_out + idx) = h;
            *reinterpret_cast<h4 *>(split_out + split_plane + idx) = l;
        }
        if (rows_out) *reinterpret_cast<f32x4 *>(rows_out + idx) = y;  // training keeps the raw conv output in `rows` for the backward pass
        else if (!split_out) *reinterpret_cast<f32x4 *>(rows + idx) = y;
        if (dense) *reinterpret_cast<f32x4 *>(dense + (size_t)rows_lin[idx / C] * C + c) = y;
    }
}

__global__ void gather_codes_kernel(const float *__restrict__ codes, const int *__restrict__ rows_vert,
                                    const int *__restrict__ n_rows, int C, float *__restrict__ rows) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)(*n_rows) * C) return;
    const int c = (int)(idx % C);
    const long long r = idx / C;
    rows[idx] = codes[(size_t)rows_vert[r] * C + c];
}

// blocks of bn_relu_kernel for a capacity of `total` elements: four per thread, grid-stride beyond 2048 blocks
inline long long bn_blocks(long long total) {
    const long long b = nb_ceil_div(total > 4 ? total / 4 : 1, 256);
    return b < 1 ? 1 : (b > 2048 ? 2048 : b);
}

template <int CIN, int COUT>
void launch_conv(int n_out_max, hipStream_t st, const float *in_rows, const int *in_grid, Dims gi, const int *out_lin,
                 const int *n_out, Dims go, int stride, const float *weight, float *out_rows, double *stats) {
    hipLaunchKernelGGL((conv_kernel<CIN, COUT>), dim3(nb_ceil_div(n_out_max, 128), (COUT + 31) / 32), dim3(256), 0, st, in_rows, in_grid,
                       gi, out_lin, n_out, go, stride, weight, out_rows, stats);
}

}  // namespace

template <bool BF>
static int conv16_dispatch(const uint16_t *in_split, int32_t in_rows_cap, const int32_t *in_grid, Dims gi, const int32_t *out_lin,
                           const int32_t *n_out, int32_t n_out_max, Dims go, int32_t stride, const uint16_t *wpacked, int32_t cin,
                           int32_t cout, float *out_rows, double *stats, hipStream_t st) {
    const long long plane = (long long)in_rows_cap * cin;
    const int row_groups = (int)nb_ceil_div(n_out_max, 128);
    // small levels (the deepest one of the SMPL grid: ~1.6 k rows): the launch is a handful of workgroups whichever way it is
    // tiled, so the chain of dependent L2 round trips per workgroup is what counts — offsets split over 8 waves (24 us instead
    // of 67-98 us per 128-channel layer); larger levels are L2-bandwidth-bound under that tiling (weights per 32 rows)
    if (n_out_max <= 4096) {
#define NB_CONV16_KS_CASE(CI, CO)                                                                                           \
    if (cin == CI && cout == CO) {                                                                                          \
        hipLaunchKernelGGL((conv16_ks_kernel<CI, CO, 8, BF>), dim3((unsigned)nb_ceil_div(n_out_max, 32), CO / 32), dim3(512), 0, st, \
                           in_split, plane, in_grid, gi, out_lin, n_out, go, stride, reinterpret_cast<const bf16x8 *>(wpacked), \
                           out_rows, stats);                                                                                \
        NB_CHECK_LAUNCH("nb_enc_conv16");                                                                                   \
        return NB_OK;                                                                                                       \
    }
        NB_CONV16_KS_CASE(32, 32)
        NB_CONV16_KS_CASE(32, 64)
        NB_CONV16_KS_CASE(64, 32)
        NB_CONV16_KS_CASE(64, 64)
        NB_CONV16_KS_CASE(64, 128)
        NB_CONV16_KS_CASE(128, 64)
        NB_CONV16_KS_CASE(128, 128)
#undef NB_CONV16_KS_CASE
    }
    // 64- and 128-channel layers: weight slab shared through LDS (measured slower for the 32-channel ones: 35 vs 29 us); all channel
    // tiles per wave when the rows alone give >= 512 workgroups, else two per wave (128-row x 64-channel workgroups)
#define NB_CONV16_LDS_CASE(CI, CO)                                                                                          \
    if (cin == CI && cout == CO) {                                                                                           \
        constexpr int NTT = CO / 32;                                                                                        \
        if (row_groups >= 512 || NTT <= 2)                                                                                  \
            hipLaunchKernelGGL((conv16_lds_kernel<CI, CO, NTT, BF>), dim3(row_groups, 1), dim3(256), 0, st, in_split, plane, in_grid, \
                               gi, out_lin, n_out, go, stride, reinterpret_cast<const bf16x8 *>(wpacked), out_rows, stats); \
        else                                                                                                                \
            hipLaunchKernelGGL((conv16_lds_kernel<CI, CO, 2, BF>), dim3(row_groups, NTT / 2), dim3(256), 0, st, in_split, plane, \
                               in_grid, gi, out_lin, n_out, go, stride, reinterpret_cast<const bf16x8 *>(wpacked), out_rows,     \
                               stats);                                                                                      \
        NB_CHECK_LAUNCH("nb_enc_conv16");                                                                                   \
        return NB_OK;                                                                                                       \
    }
    NB_CONV16_LDS_CASE(64, 32)
    NB_CONV16_LDS_CASE(64, 64)
    NB_CONV16_LDS_CASE(64, 128)
    NB_CONV16_LDS_CASE(128, 64)
    NB_CONV16_LDS_CASE(128, 128)
#undef NB_CONV16_LDS_CASE
#define NB_CONV16_CASE(CI, CO)                                                                                              \
    if (cin == CI && cout == CO) {                                                                                          \
        constexpr int NTT = CO / 32;                                                                                        \
        if (row_groups * 4 >= 2048 || NTT == 1)                                                                             \
            hipLaunchKernelGGL((conv16_kernel<CI, CO, NTT, BF>), dim3(row_groups, 1), dim3(256), 0, st, in_split, plane, in_grid, gi, \
                               out_lin, n_out, go, stride, reinterpret_cast<const bf16x8 *>(wpacked), out_rows, stats);     \
        else                                                                                                                \
            hipLaunchKernelGGL((conv16_kernel<CI, CO, 1, BF>), dim3(row_groups, NTT), dim3(256), 0, st, in_split, plane, in_grid, gi, \
                               out_lin, n_out, go, stride, reinterpret_cast<const bf16x8 *>(wpacked), out_rows, stats);     \
        NB_CHECK_LAUNCH("nb_enc_conv16");                                                                                   \
        return NB_OK;                                                                                                       \
    }
    NB_CONV16_CASE(32, 32)
    NB_CONV16_CASE(32, 64)
#undef NB_CONV16_CASE
    nb_set_error("nb_enc_conv16: unsupported channel pair %d -> %d", cin, cout);
    return NB_EINVAL;
}

extern "C" {

int nb_enc_voxelize(const int32_t *coord, int32_t n_verts, const int32_t dhw[3], int32_t *grid, int32_t *rows_vert,
                    int32_t *rows_lin, int32_t *n_rows, void *scratch, int32_t call_flags, void *stream) {
    NB_REQUIRE(dhw && grid && n_rows, "nb_enc_voxelize: NULL pointer");
    NB_REQUIRE(n_verts == 0 || (coord && rows_vert && rows_lin && scratch), "nb_enc_voxelize: NULL pointer");
    NB_REQUIRE(n_verts >= 0 && dhw[0] > 0 && dhw[1] > 0 && dhw[2] > 0, "nb_enc_voxelize: bad sizes");
    NB_REQUIRE((long long)dhw[0] * dhw[1] * dhw[2] < (1LL << 31), "nb_enc_voxelize: grid too large for int32 indices");
    hipStream_t st = (hipStream_t)stream;
    const Dims g = {dhw[0], dhw[1], dhw[2]};
    const long long nvox = (long long)g.d * g.h * g.w;
    if (!(call_flags & NB_GRID_PREFILLED)) NB_HIP(hipMemsetAsync(grid, 0xFF, nvox * sizeof(int), st));
    if (n_verts == 0) {
        NB_HIP(hipMemsetAsync(n_rows, 0, sizeof(int), st));
        return NB_OK;
    }
    int *flags, *pos, *bs;
    nb_scan_carve(scratch, n_verts, &flags, &pos, &bs);
    (void)pos;
    const dim3 tiles((unsigned)nb_scan_blocks(n_verts)), blk(nbscan::BLOCK);
    hipLaunchKernelGGL(vox_scatter_kernel, dim3(nb_ceil_div(n_verts, 256)), dim3(256), 0, st, coord, n_verts, g, grid);
    hipLaunchKernelGGL(vox_flag_count_kernel, tiles, blk, 0, st, coord, n_verts, g, grid, flags, bs);
    hipLaunchKernelGGL(vox_number_kernel, tiles, blk, 0, st, coord, n_verts, g, flags, bs, grid, rows_vert, rows_lin, n_rows);
    NB_CHECK_LAUNCH("nb_enc_voxelize");
    return NB_OK;
}

int nb_enc_downsample_index(const int32_t *in_lin, const int32_t *n_in, int32_t n_in_max, const int32_t in_dhw[3],
                            const int32_t out_dhw[3], int32_t *out_grid, int32_t *out_lin, int32_t *n_out,
                            int32_t n_out_max, void *scratch, int32_t flags, void *stream) {
    NB_REQUIRE(in_lin && n_in && in_dhw && out_dhw && out_grid && out_lin && n_out && scratch,
               "nb_enc_downsample_index: NULL pointer");
    NB_REQUIRE(n_in_max >= 0 && n_out_max >= 0, "nb_enc_downsample_index: negative capacity");
    const Dims gi = {in_dhw[0], in_dhw[1], in_dhw[2]}, go = {out_dhw[0], out_dhw[1], out_dhw[2]};
    for (int k = 0; k < 3; ++k)
        NB_REQUIRE(out_dhw[k] == (in_dhw[k] + 2 - 3) / 2 + 1 && in_dhw[k] > 0,
                   "nb_enc_downsample_index: out_dhw[%d] = %d is not floor((%d - 1) / 2) + 1", k, out_dhw[k], in_dhw[k]);
    hipStream_t st = (hipStream_t)stream;
    const long long nvox = (long long)go.d * go.h * go.w;
    if (!(flags & NB_GRID_PREFILLED)) NB_HIP(hipMemsetAsync(out_grid, 0xFF, nvox * sizeof(int), st));
    int *fl, *pos, *bs;
    nb_scan_carve(scratch, nvox, &fl, &pos, &bs);
    if (n_in_max > 0)
        hipLaunchKernelGGL(down_mark_kernel, dim3(nb_ceil_div(n_in_max, 256)), dim3(256), 0, st, in_lin, n_in, gi, go,
                           out_grid);
    (void)fl, (void)pos;
    const dim3 tiles((unsigned)nb_scan_blocks(nvox)), blk(nbscan::BLOCK);
    hipLaunchKernelGGL(grid_count_kernel, tiles, blk, 0, st, out_grid, nvox, bs);
    hipLaunchKernelGGL(grid_number_kernel, tiles, blk, 0, st, out_grid, nvox, bs, n_out_max, out_lin, n_out);
    NB_CHECK_LAUNCH("nb_enc_downsample_index");
    return NB_OK;
}

int nb_enc_downsample_index_all(const int32_t *in_lin, const int32_t *n_in, int32_t n_in_max, const int32_t in_dhw[3], int32_t n_levels,
                                int32_t *const out_grid[], int32_t *const out_lin[], int32_t *const n_out[], const int32_t n_out_max[],
                                void *scratch, int32_t flags, void *stream) {
    NB_REQUIRE(in_lin && n_in && in_dhw && out_grid && out_lin && n_out && n_out_max && scratch, "nb_enc_downsample_index_all: NULL pointer");
    NB_REQUIRE(n_levels >= 1 && n_levels <= NB_DOWN_LEVELS_MAX, "nb_enc_downsample_index_all: %d levels (1..%d)", n_levels, NB_DOWN_LEVELS_MAX);
    NB_REQUIRE(n_in_max >= 0 && in_dhw[0] > 0 && in_dhw[1] > 0 && in_dhw[2] > 0, "nb_enc_downsample_index_all: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    DownAll a = {};
    a.n_levels = n_levels;
    Dims d = {in_dhw[0], in_dhw[1], in_dhw[2]};
    const long long nvox1 = (long long)((d.d - 1) / 2 + 1) * ((d.h - 1) / 2 + 1) * ((d.w - 1) / 2 + 1);
    for (int l = 0; l < n_levels; ++l) {
        NB_REQUIRE(out_grid[l] && out_lin[l] && n_out[l] && n_out_max[l] >= 0, "nb_enc_downsample_index_all: level %d: NULL pointer or negative capacity", l);
        d = {(d.d - 1) / 2 + 1, (d.h - 1) / 2 + 1, (d.w - 1) / 2 + 1};
        a.dims[l] = d;
        a.grid[l] = out_grid[l];
        a.out_lin[l] = out_lin[l];
        a.n_out[l] = n_out[l];
        a.cap[l] = n_out_max[l];
        a.nvox[l] = (long long)d.d * d.h * d.w;
        a.tile0[l + 1] = a.tile0[l] + (int)nb_scan_blocks(a.nvox[l]);
        if (!(flags & NB_GRID_PREFILLED)) NB_HIP(hipMemsetAsync(out_grid[l], 0xFF, a.nvox[l] * sizeof(int), st));
    }
    // the tiles' sums of all levels lie in the flags area of a scratch sized for the first level: nb_scan_scratch_size(max(cells of
    // level 1, 64)) — a level of more than one tile has > 1024 cells, so n_levels <= 4 tiles or far fewer tiles than cells
    const long long scratch_cells = nvox1 > 64 ? nvox1 : 64;
    int *fl, *pos, *bs;
    nb_scan_carve(scratch, scratch_cells, &fl, &pos, &bs);
    (void)pos, (void)bs;
    NB_REQUIRE(a.tile0[n_levels] <= scratch_cells, "nb_enc_downsample_index_all: %d tiles for a scratch of %lld ints", a.tile0[n_levels], scratch_cells);
    const Dims gi = {in_dhw[0], in_dhw[1], in_dhw[2]};
    if (n_in_max > 0)
        hipLaunchKernelGGL(down_mark_all_kernel, dim3(nb_ceil_div(n_in_max, 256)), dim3(256), 0, st, in_lin, n_in, gi, a);
    const dim3 tiles((unsigned)a.tile0[n_levels]), blk(nbscan::BLOCK);
    hipLaunchKernelGGL(grid_count_all_kernel, tiles, blk, 0, st, a, fl);
    hipLaunchKernelGGL(grid_number_all_kernel, tiles, blk, 0, st, a, fl);
    NB_CHECK_LAUNCH("nb_enc_downsample_index_all");
    return NB_OK;
}

int nb_enc_conv(const float *in_rows, const int32_t *in_grid, const int32_t in_dhw[3], const int32_t *out_lin,
                const int32_t *n_out, int32_t n_out_max, const int32_t out_dhw[3], int32_t stride, const float *weight,
                int32_t cin, int32_t cout, float *out_rows, double *stats, int32_t flags, void *stream) {
    NB_REQUIRE(in_rows && in_grid && in_dhw && out_lin && n_out && out_dhw && weight && out_rows && stats,
               "nb_enc_conv: NULL pointer");
    NB_REQUIRE(stride == 1 || stride == 2, "nb_enc_conv: stride %d", stride);
    hipStream_t st = (hipStream_t)stream;
    const Dims gi = {in_dhw[0], in_dhw[1], in_dhw[2]}, go = {out_dhw[0], out_dhw[1], out_dhw[2]};
    if (!(flags & NB_CONV_STATS_ZEROED)) NB_HIP(hipMemsetAsync(stats, 0, 2 * (size_t)cout * sizeof(double), st));
    if (n_out_max <= 0) return NB_OK;
#define NB_CONV_CASE(CI, CO)                                                                                     \
    if (cin == CI && cout == CO) {                                                                               \
        launch_conv<CI, CO>(n_out_max, st, in_rows, in_grid, gi, out_lin, n_out, go, stride, weight, out_rows, stats); \
        NB_CHECK_LAUNCH("nb_enc_conv");                                                                          \
        return NB_OK;                                                                                            \
    }
    NB_CONV_CASE(16, 16)
    NB_CONV_CASE(16, 32)
    NB_CONV_CASE(32, 32)
    NB_CONV_CASE(32, 64)
    NB_CONV_CASE(64, 64)
    NB_CONV_CASE(64, 128)
    NB_CONV_CASE(128, 128)
#undef NB_CONV_CASE
    nb_set_error("nb_enc_conv: unsupported channel pair %d -> %d", cin, cout);
    return NB_EINVAL;
}

int nb_enc_bn_relu(float *rows, const int32_t *n_rows, int32_t n_rows_max, int32_t c, const double *stats,
                   const float *gamma, const float *beta, float *running_mean, float *running_var, int training,
                   float eps, float momentum, float *batch_stats, const int32_t *rows_lin, float *dense, float *rows_out,
                   void *stream) {
    NB_REQUIRE(rows && n_rows && gamma && beta, "nb_enc_bn_relu: NULL pointer");
    NB_REQUIRE(training ? stats != nullptr : (running_mean && running_var), "nb_enc_bn_relu: statistics missing");
    NB_REQUIRE(!(training && momentum >= 0.f) || (running_mean && running_var && batch_stats),
               "nb_enc_bn_relu: running statistics / batch_stats required to update them");
    NB_REQUIRE(!dense || rows_lin, "nb_enc_bn_relu: rows_lin required with dense");
    NB_REQUIRE(c > 0 && n_rows_max >= 0, "nb_enc_bn_relu: bad sizes");
    NB_REQUIRE(c % 4 == 0 && c <= BN_MAX_C, "nb_enc_bn_relu: %d channels (a multiple of 4, at most %d)", c, BN_MAX_C);
    const long long total = (long long)n_rows_max * c;
    hipLaunchKernelGGL(bn_relu_kernel, dim3((unsigned)bn_blocks(total)), dim3(256), 0, (hipStream_t)stream, rows, n_rows,
                       c, stats, gamma, beta, running_mean, running_var, training, eps, momentum, batch_stats, rows_lin, dense, rows_out,
                       (_Float16 *)nullptr, 0LL);
    NB_CHECK_LAUNCH("nb_enc_bn_relu");
    return NB_OK;
}

int nb_enc_bn_relu_split(const float *rows, const int32_t *n_rows, int32_t n_rows_max, int32_t c, const double *stats,
                         const float *gamma, const float *beta, float *running_mean, float *running_var, int training,
                         float eps, float momentum, float *batch_stats, const int32_t *rows_lin, float *dense,
                         uint16_t *rows_split, float *rows_out, void *stream) {
    NB_REQUIRE(rows && n_rows && gamma && beta && rows_split, "nb_enc_bn_relu_split: NULL pointer");
    NB_REQUIRE(training ? stats != nullptr : (running_mean && running_var), "nb_enc_bn_relu_split: statistics missing");
    NB_REQUIRE(!(training && momentum >= 0.f) || (running_mean && running_var && batch_stats),
               "nb_enc_bn_relu_split: running statistics / batch_stats required to update them");
    NB_REQUIRE(!dense || rows_lin, "nb_enc_bn_relu_split: rows_lin required with dense");
    NB_REQUIRE(c > 0 && n_rows_max > 0, "nb_enc_bn_relu_split: bad sizes");
    NB_REQUIRE(c % 4 == 0 && c <= BN_MAX_C, "nb_enc_bn_relu_split: %d channels (a multiple of 4, at most %d)", c, BN_MAX_C);
    const long long total = (long long)n_rows_max * c;
    hipLaunchKernelGGL(bn_relu_kernel, dim3((unsigned)bn_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       const_cast<float *>(rows), n_rows, c, stats, gamma, beta, running_mean, running_var, training, eps, momentum,
                       batch_stats, rows_lin, dense, rows_out, reinterpret_cast<_Float16 *>(rows_split), total);
    NB_CHECK_LAUNCH("nb_enc_bn_relu_split");
    return NB_OK;
}

int nb_enc_conv_pack16(const float *weight, int32_t cin, int32_t cout, uint16_t *packed, int32_t mode, void *stream) {
    NB_REQUIRE(weight && packed, "nb_enc_conv_pack16: NULL pointer");
    NB_REQUIRE(mode == 0 || mode == 1, "nb_enc_conv_pack16: mode %d", mode);
    NB_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 32 && cout % 32 == 0, "nb_enc_conv_pack16: channel pair %d -> %d", cin, cout);
    const long long n = 27LL * (cin / 16) * (cout / 32) * 2 * 64;
    hipLaunchKernelGGL(conv_pack16_kernel, dim3(nb_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, weight, cin, cout,
                       reinterpret_cast<bf16x8 *>(packed), mode);
    NB_CHECK_LAUNCH("nb_enc_conv_pack16");
    return NB_OK;
}

int nb_enc_conv_pack16_batch(int32_t n_jobs, const float *const weight[], const int32_t cin[], const int32_t cout[],
                             uint16_t *const packed[], const int32_t mode[], void *stream) {
    NB_REQUIRE(n_jobs >= 0 && n_jobs <= NB_PACK_BATCH_MAX, "nb_enc_conv_pack16_batch: %d jobs (at most %d)", n_jobs, NB_PACK_BATCH_MAX);
    if (n_jobs == 0) return NB_OK;
    NB_REQUIRE(weight && cin && cout && packed && mode, "nb_enc_conv_pack16_batch: NULL pointer");
    PackBatch b = {};
    long long most = 0;
    for (int i = 0; i < n_jobs; ++i) {
        NB_REQUIRE(weight[i] && packed[i], "nb_enc_conv_pack16_batch: job %d: NULL pointer", i);
        NB_REQUIRE(mode[i] == 0 || mode[i] == 1, "nb_enc_conv_pack16_batch: job %d: mode %d", i, mode[i]);
        NB_REQUIRE(cin[i] >= 16 && cin[i] % 16 == 0 && cout[i] >= 32 && cout[i] % 32 == 0, "nb_enc_conv_pack16_batch: job %d: channel pair %d -> %d",
                   i, cin[i], cout[i]);
        b.w[i] = weight[i];
        b.out[i] = reinterpret_cast<bf16x8 *>(packed[i]);
        b.cin[i] = cin[i];
        b.cout[i] = cout[i];
        b.mode[i] = mode[i];
        const long long n = 27LL * (cin[i] / 16) * (cout[i] / 32) * 2 * 64;
        most = n > most ? n : most;
    }
    hipLaunchKernelGGL(conv_pack16_batch_kernel, dim3((unsigned)nb_ceil_div(most, 256), n_jobs), dim3(256), 0, (hipStream_t)stream, b);
    NB_CHECK_LAUNCH("nb_enc_conv_pack16_batch");
    return NB_OK;
}

int nb_enc_conv16(const uint16_t *in_split, int32_t in_rows_cap, const int32_t *in_grid, const int32_t in_dhw[3],
                  const int32_t *out_lin, const int32_t *n_out, int32_t n_out_max, const int32_t out_dhw[3], int32_t stride,
                  const uint16_t *wpacked, int32_t cin, int32_t cout, float *out_rows, double *stats, int32_t flags,
                  void *stream) {
    NB_REQUIRE(in_split && in_grid && in_dhw && out_lin && n_out && out_dhw && wpacked && out_rows && stats,
               "nb_enc_conv16: NULL pointer");
    NB_REQUIRE(stride == 1 || stride == 2 || stride == -2, "nb_enc_conv16: stride %d", stride);
    NB_REQUIRE(in_rows_cap > 0, "nb_enc_conv16: in_rows_cap %d", in_rows_cap);
    hipStream_t st = (hipStream_t)stream;
    const Dims gi = {in_dhw[0], in_dhw[1], in_dhw[2]}, go = {out_dhw[0], out_dhw[1], out_dhw[2]};
    if (!(flags & NB_CONV_STATS_ZEROED)) NB_HIP(hipMemsetAsync(stats, 0, 2 * (size_t)cout * sizeof(double), st));
    if (n_out_max <= 0) return NB_OK;
    return (flags & NB_CONV_BF16) ? conv16_dispatch<true>(in_split, in_rows_cap, in_grid, gi, out_lin, n_out, n_out_max, go, stride, wpacked,
                                                          cin, cout, out_rows, stats, st)
                                  : conv16_dispatch<false>(in_split, in_rows_cap, in_grid, gi, out_lin, n_out, n_out_max, go, stride, wpacked,
                                                           cin, cout, out_rows, stats, st);
}

int nb_enc_gather_codes(const float *codes, const int32_t *rows_vert, const int32_t *n_rows, int32_t n_rows_max,
                        int32_t c, float *rows, void *stream) {
    NB_REQUIRE(codes && rows_vert && n_rows && rows, "nb_enc_gather_codes: NULL pointer");
    if (n_rows_max <= 0) return NB_OK;
    hipLaunchKernelGGL(gather_codes_kernel, dim3(nb_ceil_div((long long)n_rows_max * c, 256)), dim3(256), 0,
                       (hipStream_t)stream, codes, rows_vert, n_rows, c, rows);
    NB_CHECK_LAUNCH("nb_enc_gather_codes");
    return NB_OK;
}

}  // extern "C"
