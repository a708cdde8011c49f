// nb_fold.hip — fc_0 folded into the latent volumes (NB_PREC_F16F6V), encoder side.
//
// Trilinear interpolation (latent_xyzc.py:62-72) and fc_0 (:99) are both linear and nothing sits between them, so
//     fc_0 . interp(V)  =  interp(fc_0 . V).
// nb_fold_build forms U_l[r, :] = fc_0.weight[:, channels of level l] . V_l[voxel of row r] for every ACTIVE voxel of the four
// volumes (22 k / 29 k / 10 k / 1.6 k rows on the 6890-vertex body; ~2 GFLOP per frame) in exact fp32
// (v_mfma_f32_32x32x2_f32 = an fmaf chain) and stores each row as 256 fp16 heads + 256 fp16 remainders (1 KiB): what the march
// (nb_march_fold.hip) fetches per touched voxel and contracts with the trilinear weights on the matrix pipe.
// nb_sparsify derives the active set of a dense volume that came without one.
#include "nb_march_common.h"
#include "nb_scan.h"

typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

namespace {

using namespace nbm;

constexpr int ROWS_WG = 128;  // rows per workgroup of the 32- and 64-channel levels: 4 row tiles per wave share every fc_0 fragment
// the 128-channel levels take 64 rows per workgroup: a workgroup's 4 C MFMAs per wave (64 cycles each on the exact-fp32 pipe) are
// the launch's critical path — 512 of them = 16 us with a SIMD to itself, the whole launch used to take 80
__host__ __device__ constexpr int rows_wg(int level) { return level >= 2 ? 64 : ROWS_WG; }
constexpr int N_OUT = 256;

struct FoldArgs {
    const float *vol[4];
    const int *rows_lin[4];
    const int *n_rows[4];
    int cap[4];
    int row_base[4];
    int tile_base[5];  // first workgroup of each level
    const float *w0;   // [256, 352]
    unsigned short *urows;
    int zero_row;
    int *n_sat;        // optional: how many products left the fp16 range (or are not finite)
};

// one workgroup = 128 rows x 256 outputs of one level; wave w = outputs [64 w, 64 w + 64)
template <int C, int RT>
__device__ __forceinline__ void fold_rows_level(const FoldArgs &a, int L, int tile, char *lds) {
    constexpr int PITCH = C + 1;  // floats: lanes i = 0..31 of an A fragment read rows i at distinct banks
    constexpr int ROWS_WG = 32 * RT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(*a.n_rows[L], a.cap[L]);
    const int row0 = tile * ROWS_WG;
    if (row0 >= n) return;  // workgroup-uniform
    float *at = reinterpret_cast<float *>(lds);
    // ---- stage the gathered rows
    {
        constexpr int Q = C / 4;  // float4 per row
        for (int idx = tid; idx < ROWS_WG * Q; idx += 256) {
            const int r = idx / Q, q = idx % Q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r < n) {
                // the level's rows straight from the encoder (compact [n_rows_max, C], rows_lin NULL) or gathered from the
                // dense volume through rows_lin
                const size_t lin = a.rows_lin[L] ? (size_t)a.rows_lin[L][row0 + r] : (size_t)(row0 + r);
                v = *reinterpret_cast<const f32x4 *>(a.vol[L] + lin * C + q * 4);
            }
            float *d = at + r * PITCH + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    const int i = lane & 31, kh = lane >> 5;
    f32x16 acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][t][r] = 0.f;
    const int cb = lvl_chan_base(L);
    // B[k][n] = fc_0.weight[n][cb + k]: the lane's weight row is contiguous along k — one 16-byte load serves two K = 2 steps
    // (elements kh and 2 + kh; cb is a multiple of 32, the rows 1408 bytes apart: aligned).  As one float per lane and step the
    // loads were 64 different lines per instruction, four bytes used of each
    const float *wb0 = a.w0 + (size_t)(64 * wave + i) * 352 + cb;
    const float *wb1 = wb0 + 32 * 352;
#pragma unroll 2
    for (int k4 = 0; k4 < C / 4; ++k4) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wb0 + 4 * k4), w1 = *reinterpret_cast<const f32x4 *>(wb1 + 4 * k4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = 2 * k4 + h;
            const float b0 = kh ? (h ? w0.w : w0.y) : (h ? w0.z : w0.x), b1 = kh ? (h ? w1.w : w1.y) : (h ? w1.z : w1.x);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float av = at[(32 * rt + i) * PITCH + 2 * kk + kh];
                acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[rt][1], 0, 0, 0);
            }
        }
    }
    // ---- rows out: through an LDS image [32 rows][256 heads | 256 remainders] so that a row leaves as 64 x 16 bytes
    unsigned short *st = reinterpret_cast<unsigned short *>(lds);
    bool sat = false;  // a product beyond +-65504 is clamped (the head would be inf, the remainder inf - inf), a NaN stays a NaN:
                       // either is counted, and Network's 'auto' leaves the fp16 planes for the exact kernel when the count is not 0
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        __syncthreads();  // the A tile / the previous image is no longer read
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tile_row(r, kh), f = 64 * wave + 32 * t + i;
                const float u = acc[rt][t][r];
                sat = sat || !(fabsf(u) <= 65504.f);
                const float v = u > 65504.f ? 65504.f : (u < -65504.f ? -65504.f : u);
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);
                st[row * 512 + f] = __builtin_bit_cast(unsigned short, h);
                st[row * 512 + 256 + f] = __builtin_bit_cast(unsigned short, l);
            }
        __syncthreads();
        for (int idx = tid; idx < 32 * 64; idx += 256) {
            const int row = idx >> 6, piece = idx & 63;
            const int gr = row0 + 32 * rt + row;
            if (gr < n)
                *reinterpret_cast<f32x4 *>(a.urows + ((size_t)(a.row_base[L] + gr) * 512 + piece * 8)) =
                    *reinterpret_cast<const f32x4 *>(st + row * 512 + piece * 8);
        }
    }
    // (rows beyond n are zero-staged: their products are 0)
    if (a.n_sat && __builtin_amdgcn_ballot_w64(sat) != 0ull && lane == 0) atomicAdd(a.n_sat, __builtin_popcountll(__builtin_amdgcn_ballot_w64(sat)));
}

__global__ __launch_bounds__(256, 2) void nb_fold_rows_kernel(FoldArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int b = blockIdx.x;
    if (b == 0 && threadIdx.x < 64)
        *reinterpret_cast<f32x4 *>(a.urows + ((size_t)a.zero_row * 512 + threadIdx.x * 8)) = f32x4{0.f, 0.f, 0.f, 0.f};
    const int L = (b >= a.tile_base[1]) + (b >= a.tile_base[2]) + (b >= a.tile_base[3]);
    const int tile = b - a.tile_base[L];
    if (L == 0) fold_rows_level<32, 4>(a, 0, tile, lds);
    else if (L == 1) fold_rows_level<64, 4>(a, 1, tile, lds);
    else fold_rows_level<128, 2>(a, L, tile, lds);
}

// ---------------------------------------------------------------- active set of a dense volume
__global__ void sparsify_flag_kernel(const float *__restrict__ vol, long long nvox, int c, int *__restrict__ flags) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox) return;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(vol + v * c);
    bool any = false;
    for (int q = 0; q < c / 4; ++q) {
        const f32x4 x = p[q];
        any = any || x.x != 0.f || x.y != 0.f || x.z != 0.f || x.w != 0.f;
    }
    flags[v] = any ? 1 : 0;
}
__global__ void sparsify_assign_kernel(const int *__restrict__ flags, const int *__restrict__ pos, long long nvox, int cap,
                                       int *__restrict__ grid, int *__restrict__ rows_lin, int *__restrict__ n_rows) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) *n_rows = min(*n_rows, cap);  // the scan wrote the grand total here
    if (v >= nvox) return;
    const int p = pos[v];
    const bool on = flags[v] != 0 && p < cap;
    grid[v] = on ? p : -1;
    if (on) rows_lin[p] = (int)v;
}

}  // namespace

extern "C" {

int nb_fold_build(const float *const vol[NB_N_LEVELS], const int32_t *const rows_lin[NB_N_LEVELS],
                  const int32_t *const n_rows[NB_N_LEVELS], const int32_t n_rows_max[NB_N_LEVELS], const float *fc0_w,
                  uint16_t *urows, int32_t *n_saturated, void *stream) {
    NB_REQUIRE(vol && rows_lin && n_rows && n_rows_max && fc0_w && urows, "nb_fold_build: NULL pointer");
    FoldArgs a = {};
    int base = 0, tiles = 0;
    for (int l = 0; l < 4; ++l) {
        NB_REQUIRE(vol[l] && n_rows[l] && n_rows_max[l] >= 1, "nb_fold_build: level %d: NULL pointer or empty capacity", l);
        a.vol[l] = vol[l];
        a.rows_lin[l] = rows_lin[l];
        a.n_rows[l] = n_rows[l];
        a.cap[l] = n_rows_max[l];
        a.row_base[l] = base;
        a.tile_base[l] = tiles;
        base += n_rows_max[l];
        tiles += nb_ceil_div(n_rows_max[l], rows_wg(l));
    }
    a.tile_base[4] = tiles;
    a.w0 = fc0_w;
    a.urows = urows;
    a.zero_row = base;
    a.n_sat = n_saturated;
    NB_REQUIRE(base < (1 << 21), "nb_fold_build: %d rows: the march addresses the 1-KiB rows with 32-bit byte offsets (< 2^21 rows)", base);
    hipStream_t st = (hipStream_t)stream;
    // the largest A tile: 128 rows x (64 + 1) floats or 64 rows x (128 + 1); the 32 KiB output image aliases it
    const size_t lds = (size_t)(ROWS_WG * 65 > 64 * 129 ? ROWS_WG * 65 : 64 * 129) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        NB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nb_fold_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(nb_fold_rows_kernel, dim3(tiles), dim3(256), lds, st, a);
    NB_CHECK_LAUNCH("nb_fold_rows_kernel");
    return NB_OK;
}

int nb_sparsify(const float *vol, const int32_t dhw[3], int32_t c, int32_t *grid, int32_t *rows_lin, int32_t *n_rows,
                int32_t n_rows_max, void *scratch, void *stream) {
    NB_REQUIRE(vol && dhw && grid && rows_lin && n_rows && scratch, "nb_sparsify: NULL pointer");
    NB_REQUIRE(c >= 4 && c % 4 == 0 && n_rows_max >= 1, "nb_sparsify: c = %d, n_rows_max = %d", c, n_rows_max);
    const long long nvox = (long long)dhw[0] * dhw[1] * dhw[2];
    NB_REQUIRE(nvox >= 1 && nvox < (1ll << 31), "nb_sparsify: %lld voxels", nvox);
    hipStream_t st = (hipStream_t)stream;
    int *flags, *pos, *bsum;
    nb_scan_carve(scratch, nvox, &flags, &pos, &bsum);
    const int nb = nb_ceil_div(nvox, 256);
    hipLaunchKernelGGL(sparsify_flag_kernel, dim3(nb), dim3(256), 0, st, vol, nvox, c, flags);
    if (int rc = nb_exclusive_scan(flags, pos, n_rows, nvox, bsum, st)) return rc;
    hipLaunchKernelGGL(sparsify_assign_kernel, dim3(nb), dim3(256), 0, st, flags, pos, nvox, n_rows_max, grid, rows_lin, n_rows);
    NB_CHECK_LAUNCH("nb_sparsify");
    return NB_OK;
}

}  // extern "C"
