// nb_encoder_bwd.hip — backward of the structured-latent-code encoder (training step, §8 row a15): the autograd of
// SparseConvNet (lib/networks/latent_xyzc.py:166-274) as explicit kernels on the index-grid sparse tensors of
// nb_encoder.hip.
//
//   y = relu(gamma * xhat + beta),  xhat = (x - mean) * invstd,  x[r] = sum_o in[nbr(r, o)] @ W[o]
//
//   nb_enc_bn_relu_bwd   g = dy (.) [y > 0];  dgamma = sum g xhat;  dbeta = sum g;
//                        dx = invstd gamma (g - mean(g) - xhat mean(g xhat))          (batch statistics, train())
//   nb_enc_conv_bwd_input   d in[q] = sum_o dx[r(q, o)] @ W[o]^T     (gather form: r(q,o) is the output row whose
//                        receptive field holds q under offset o; stride 1: u = p + 1 - k, stride 2: u = (p + 1 - k) / 2)
//   nb_enc_conv_bwd_weight  dW[o] = sum_r in[nbr(r, o)]^T (x) dx[r]   (MFMA over row chunks, fp32 atomics into dW)
//   nb_enc_scatter_codes_bwd  d c.weight[vertex of row r] = d rows[r]  (Embedding lookup backward, :33-34)
#include "nb_common.h"
#include "nb_trread.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

struct Dims {
    int d, h, w;
};

__host__ __device__ constexpr int tile_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ------------------------------------------------------------------ BatchNorm + ReLU backward
__global__ __launch_bounds__(1024) void bn_bwd_reduce_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                             const float *__restrict__ x, const int *__restrict__ n_rows, int C,
                                                             const float *__restrict__ batch_stats, float eps, double *__restrict__ sums) {
    // block = 1024 threads = 16 row lanes x 64 channels; grid.x over channel groups, grid.y over row slabs.  Few, fat blocks:
    // every block ends in one fp64 atomic per channel and sum, and L2 serialises the atomics of an address — with round 4's
    // 2048 slabs of 256 threads a layer spent its 20 us waiting on 2048 atomics per address (r05_train_kernel_stats.md)
    constexpr int RL = 16;
    __shared__ double pa[RL][64], pb[RL][64];
    const int n = *n_rows;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const long long per = ((long long)n + gridDim.y - 1) / gridDim.y;
    const long long r0 = (long long)blockIdx.y * per, r1 = min((long long)n, r0 + per);
    double a = 0.0, b = 0.0;
    if (c < C) {
        const double mean = batch_stats[c], invstd = 1.0 / sqrt((double)batch_stats[C + c] + (double)eps);
        // four rows per trip, their 12 loads issued together and four independent fp64 chains: the one-row loop ran at the
        // latency of a load + two dependent fp64 adds per row
        double a4[4] = {0.0, 0.0, 0.0, 0.0}, b4[4] = {0.0, 0.0, 0.0, 0.0};
        for (long long r = r0 + w; r < r1; r += 4 * RL) {
            float yv[4], dv[4], xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long rr = r + RL * q;
                const bool ok = rr < r1;
                const long long e = (ok ? rr : r) * C + c;
                yv[q] = y[e];
                dv[q] = ok ? dy[e] : 0.f;
                xv[q] = x[e];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g = yv[q] > 0.f ? dv[q] : 0.f;
                a4[q] += (double)g;
                b4[q] += (double)g * (((double)xv[q] - mean) * invstd);
            }
        }
        a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        b = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    }
    pa[w][threadIdx.x & 63] = a;
    pb[w][threadIdx.x & 63] = b;
    __syncthreads();
    if (w == 0 && c < C) {
        const int t = threadIdx.x;
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int q = 0; q < RL; ++q) {  // fixed order
            sa += pa[q][t];
            sb += pb[q][t];
        }
        atomicAdd(&sums[c], sa);
        atomicAdd(&sums[C + c], sb);
    }
}

__global__ void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ y, const float *__restrict__ x,
                                    const int *__restrict__ n_rows, int C, const float *__restrict__ batch_stats, float eps,
                                    const float *__restrict__ gamma, const double *__restrict__ sums,
                                    float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                    __bf16 *__restrict__ dx_split, long long split_plane) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *n_rows;
    if (idx < C) {
        dgamma[idx] = (float)sums[C + idx];
        dbeta[idx] = (float)sums[idx];
    }
    if (idx >= (long long)n * C) return;
    const int c = (int)(idx % C);
    const double mean = batch_stats[c], invstd = 1.0 / sqrt((double)batch_stats[C + c] + (double)eps);
    const double xhat = ((double)x[idx] - mean) * invstd;
    const double g = y[idx] > 0.f ? (double)dy[idx] : 0.0;
    const float v = (float)(invstd * (double)gamma[c] * (g - sums[c] / n - xhat * sums[C + c] / n));
    dx[idx] = v;
    if (dx_split) {  // bf16 head and remainder planes for nb_enc_conv16(NB_CONV_BF16)
        const __bf16 h = (__bf16)v;
        dx_split[idx] = h;
        dx_split[split_plane + idx] = (__bf16)(v - (float)h);
    }
}

// ------------------------------------------------------------------ conv backward w.r.t. the input rows
// one wave = 32 INPUT rows x 32 input channels (blockIdx.y); K runs over the Cout channels of dx
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv_bwd_in_kernel(const float *__restrict__ dx, const int *__restrict__ out_grid,
                                                          Dims go, const int *__restrict__ in_lin,
                                                          const int *__restrict__ n_in, Dims gi, int stride,
                                                          const float *__restrict__ weight, float *__restrict__ din) {
    constexpr int HALF = COUT / 2;
    const int ct = blockIdx.y;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = *n_in;
    const int row0 = wave * 32;
    if (row0 >= n) return;
    const int row = row0 + i;
    const bool valid = row < n;
    const int lin = valid ? in_lin[row] : 0;
    const int x = lin % gi.w, y = (lin / gi.w) % gi.h, z = lin / (gi.w * gi.h);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // output voxel u with u * stride - 1 + k == p, for every offset first (27 independent lookups), then — on the narrow layers,
    // where a second set of row registers is cheap — the dx rows of offset o + 1 while offset o multiplies: as a chain of lookup ->
    // row -> MFMA per offset the 16 <- 32 layer took 70 us for 6.9 k rows
    int nbrs[27];
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        const int kd = o / 9, kh = (o / 3) % 3, kw = o % 3;
        const int nz = z + 1 - kd, ny = y + 1 - kh, nx = x + 1 - kw;
        int nbr = -1;
        bool ok = valid && nz >= 0 && ny >= 0 && nx >= 0;
        int uz = nz, uy = ny, ux = nx;
        if (stride == 2) {
            ok = ok && !(nz & 1) && !(ny & 1) && !(nx & 1);
            uz = nz >> 1;
            uy = ny >> 1;
            ux = nx >> 1;
        }
        if (ok && uz < go.d && uy < go.h && ux < go.w) nbr = out_grid[((long long)uz * go.h + uy) * go.w + ux];
        nbrs[o] = nbr;
    }
    constexpr bool AHEAD = HALF <= 16;
    f32x4 Ar[AHEAD ? 2 : 1][HALF / 4];
    auto load_rows = [&](int nbr, f32x4 (&dst)[HALF / 4]) {  // (row 0 for a lane without a neighbour: zeroed below)
        const f32x4 *p = reinterpret_cast<const f32x4 *>(dx + (size_t)(nbr >= 0 ? nbr : 0) * COUT + hi * HALF);
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) dst[q] = p[q];
    };
    if (AHEAD) load_rows(nbrs[0], Ar[0]);
#pragma unroll
    for (int o = 0; o < 27; ++o) {
        if (AHEAD && o + 1 < 27) load_rows(nbrs[o + 1], Ar[(o + 1) & 1]);
        const int nbr = nbrs[o];
        if (!__any(nbr >= 0)) continue;
        if (!AHEAD) load_rows(nbr, Ar[0]);
        float A[HALF];
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) {
            const f32x4 v = Ar[AHEAD ? (o & 1) : 0][q];
            A[4 * q] = nbr >= 0 ? v.x : 0.f;
            A[4 * q + 1] = nbr >= 0 ? v.y : 0.f;
            A[4 * q + 2] = nbr >= 0 ? v.z : 0.f;
            A[4 * q + 3] = nbr >= 0 ? v.w : 0.f;
        }
        // B[k = co][j = ci] = W[o][ci][co]  (transposed read of the spconv-layout slab)
        const int ci = ct * 32 + i;
        const bool ciok = (CIN % 32 == 0) || ci < CIN;
        const float *wo = weight + ((size_t)o * CIN + (ciok ? ci : 0)) * COUT + hi * HALF;
#pragma unroll
        for (int c = 0; c < HALF; ++c) {
            const float b = ciok ? wo[c] : 0.f;
            acc = NB_MFMA(A[c], b, acc);
        }
    }
    const int ci = ct * 32 + i;
    if ((CIN % 32 == 0) || ci < CIN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = row0 + tile_row(r, hi);
            if (orow < n) din[(size_t)orow * CIN + ci] = acc[r];
        }
    }
}

// ------------------------------------------------------------------ conv backward w.r.t. the weights
// rule book: nbr[row * 27 + o] = input row read by output row `row` under kernel offset o, or -1.  Built once per layer
// call; the weight-gradient kernel then spends its issue slots on loads and MFMAs instead of redoing the integer
// divisions of the voxel index for every (offset, channel tile) pair (measured: 3.9 -> see DESIGN.md §4.4).
__global__ void conv_rulebook_kernel(const int *__restrict__ in_grid, Dims gi, const int *__restrict__ out_lin,
                                     const int *__restrict__ n_out, Dims go, int stride, int *__restrict__ nbr) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = *n_out;
    if (idx >= (long long)n * 27) return;
    const int row = (int)(idx / 27), o = (int)(idx % 27);
    const int kd = o / 9, kh = (o / 3) % 3, kw = o % 3;
    const int lin = out_lin[row];
    const int x = lin % go.w, y = (lin / go.w) % go.h, z = lin / (go.w * go.h);
    const int iz = z * stride - 1 + kd, iy = y * stride - 1 + kh, ix = x * stride - 1 + kw;
    int v = -1;
    if ((unsigned)iz < (unsigned)gi.d && (unsigned)iy < (unsigned)gi.h && (unsigned)ix < (unsigned)gi.w)
        v = in_grid[((long long)iz * gi.h + iy) * gi.w + ix];
    nbr[idx] = v;
}

// one wave = (offset o, ci tile, co tile, chunk of ROWS_PER_WAVE output rows); D[ci][co] += sum_rows in^T dx
constexpr int ROWS_PER_WAVE = 256;

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv_bwd_w_kernel(const float *__restrict__ in_rows, const int *__restrict__ nbr,
                                                         const int *__restrict__ n_out, const float *__restrict__ dx,
                                                         float *__restrict__ dw) {
    constexpr int TO = (COUT + 31) / 32;
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);  // row chunk
    const int o = blockIdx.y;                               // kernel offset 0..26
    const int ti = blockIdx.z / TO, to = blockIdx.z % TO;
    const int n = *n_out;
    const int row0 = wave * ROWS_PER_WAVE;
    if (row0 >= n) return;
    const int ci = ti * 32 + i, co = to * 32 + i;
    const bool ciok = (CIN % 32 == 0) || ci < CIN, cook = (COUT % 32 == 0) || co < COUT;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bool any = false;
    for (int m0 = 0; m0 < ROWS_PER_WAVE / 2; m0 += 8) {  // 8 K=2 chunks per round: all loads issued before the MFMAs
        float a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = row0 + 2 * (m0 + q) + hi;  // this half-wave's row of the K=2 chunk
            a[q] = 0.f;
            b[q] = 0.f;
            if (row < n) {
                const int nb = nbr[(size_t)row * 27 + o];
                if (nb >= 0) {
                    if (ciok) a[q] = in_rows[(size_t)nb * CIN + ci];
                    if (cook) b[q] = dx[(size_t)row * COUT + co];
                    any = true;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = NB_MFMA(a[q], b[q], acc);  // D[i = ci][j = co] += A[ci][k = row] B[k = row][co]
    }
    if (!__any(any)) return;
    // D fragment: lane (j = co column, hi) holds rows ci_local = tile_row(r, hi)
    if (cook) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cir = ti * 32 + tile_row(r, hi);
            if ((CIN % 32 == 0) || cir < CIN) atomicAdd(&dw[((size_t)o * CIN + cir) * COUT + co], acc[r]);
        }
    }
}

__global__ void scatter_codes_bwd_kernel(const float *__restrict__ drows, const int *__restrict__ rows_vert,
                                         const int *__restrict__ n_rows, int C, float *__restrict__ dcodes) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)(*n_rows) * C) return;
    const int c = (int)(idx % C);
    const long long r = idx / C;
    dcodes[(size_t)rows_vert[r] * C + c] = drows[idx];  // one row per vertex at most: plain store into a zeroed buffer
}

// ------------------------------------------------------------------ conv weight gradient on the 16-bit matrix pipe
// dW[o] = sum_r in[nbr(r, o)]^T (x) dx[r] as v_mfma_f32_32x32x16_bf16 with both operands as bf16 head + remainder (three
// products, fp32 accumulate).  The reduction runs over ROWS, so both MFMA operands are K-major: lane (channel, kg) needs 8
// consecutive rows of one channel, the transpose of how rows are stored.  A workgroup stages 32 gathered input rows and the 32
// matching dx rows row-major in LDS (fp32 -> bf16 pairs on the way; dx arrives as pairs from nb_enc_bn_relu_bwd) and reads the
// fragments with ds_read_b64_tr_b16 (each 16-lane group transposes a [4 rows][16 channels] block: lane l receives channel
// 16 (g & 1) + (l & 15), rows 8 (g >> 1) + 0..3 of the block whose 8-byte segments the lanes address —
// tools/experiments/probe_trread.hip).  Row pitch = 2 C + 32 bytes: the four rows of a block fall into distinct banks.
// One workgroup = (offset o, BW_ROWS consecutive output rows): the accumulators persist over its 32-row chunks and leave as
// ONE set of atomics (the exact-fp32 kernel: one set per 256 rows), wave w owns a block of the [C_in / 32] x [C_out / 32]
// tiles.  The next chunk's global loads are issued before the current chunk's MFMAs (two LDS buffers).
constexpr int BW_ROWS = 1024;
typedef nbtr::bf8 nb_bf8;
__device__ __forceinline__ nb_bf8 tr_frag(unsigned addr_lo4, unsigned addr_hi4) { return nbtr::frag(addr_lo4, addr_hi4); }

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv_bwd_w16_kernel(const float *__restrict__ in_rows, const int *__restrict__ nbr,
                                                           const int *__restrict__ n_out, const unsigned short *__restrict__ dx_split,
                                                           long long dx_plane, float *__restrict__ dw) {
    constexpr int CT = CIN / 32, OT = COUT / 32;               // tiles of dW[o]
    constexpr int WCI = CT >= 2 ? 2 : 1, WCO = OT >= 2 ? 2 : 1;  // waves along each tile axis (4 waves when both >= 2)
    constexpr int TCI = CT / WCI, TCO = OT / WCO;               // tiles per wave
    constexpr int PA = CIN * 2 + 32, PB = COUT * 2 + 32;        // LDS row pitches in bytes
    constexpr int A_BYTES = 32 * PA, B_BYTES = 32 * PB, BUF = 2 * A_BYTES + 2 * B_BYTES;  // head and remainder planes
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int o = blockIdx.x;
    const int n = *n_out;
    const int row_begin = blockIdx.y * BW_ROWS;
    if (row_begin >= n) return;  // workgroup-uniform
    const int row_end = min(n, row_begin + BW_ROWS);
    const int n_chunks = (row_end - row_begin + 31) / 32;
    const bool active = wv < WCI * WCO;
    const int wci = wv / WCO, wco = wv % WCO;

    // staging roles: 8 threads per row; thread (r, p) moves channels [p C / 8, (p + 1) C / 8) of row r
    const int sr = tid >> 3, sp = tid & 7;
    constexpr int A4 = CIN / 32, B8 = COUT / 64 > 0 ? COUT / 64 : 1;  // float4 loads of A, 16-byte loads per plane of B, per thread
    f32x4 ra[A4];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rbh[B8], rbl[B8];
    bool b_lane = (COUT >= 64) || sp < 4;  // C_out = 32: a row's plane is 64 bytes = 4 threads x 16
    auto fetch = [&](int chunk) {
        const int row = row_begin + chunk * 32 + sr;
        int nb = -1;
        if (row < row_end) nb = nbr[(size_t)row * 27 + o];
#pragma unroll
        for (int q = 0; q < A4; ++q)
            ra[q] = nb >= 0 ? *reinterpret_cast<const f32x4 *>(in_rows + (size_t)nb * CIN + sp * (CIN / 8) + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        const bool live = nb >= 0;  // a row without a neighbour under this offset contributes nothing: zero dx too
#pragma unroll
        for (int q = 0; q < B8; ++q) {
            const size_t e = (size_t)row * COUT + (size_t)(sp * B8 + q) * 8;
            rbh[q] = (live && b_lane) ? *reinterpret_cast<const u32x4 *>(dx_split + e) : u32x4{0u, 0u, 0u, 0u};
            rbl[q] = (live && b_lane) ? *reinterpret_cast<const u32x4 *>(dx_split + dx_plane + e) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto stash = [&](int buf) {
        char *base = lds + buf * BUF;
        // A: fp32 -> bf16 head / remainder, 4 values = 8 bytes per plane per float4
#pragma unroll
        for (int q = 0; q < A4; ++q) {
            const float v[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w};
            unsigned short h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const __bf16 hh = (__bf16)v[e];
                h[e] = __builtin_bit_cast(unsigned short, hh);
                l[e] = __builtin_bit_cast(unsigned short, (__bf16)(v[e] - (float)hh));
            }
            const int cb = (sp * (CIN / 8) + 4 * q) * 2;
            *reinterpret_cast<uint2 *>(base + sr * PA + cb) = uint2{h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16)};
            *reinterpret_cast<uint2 *>(base + A_BYTES + sr * PA + cb) = uint2{l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16)};
        }
        if (b_lane) {
#pragma unroll
            for (int q = 0; q < B8; ++q) {
                const int cb = (sp * B8 + q) * 16;
                *reinterpret_cast<u32x4 *>(base + 2 * A_BYTES + sr * PB + cb) = rbh[q];
                *reinterpret_cast<u32x4 *>(base + 2 * A_BYTES + B_BYTES + sr * PB + cb) = rbl[q];
            }
        }
    };

    f32x16 acc[TCI][TCO];
#pragma unroll
    for (int a = 0; a < TCI; ++a)
#pragma unroll
        for (int b = 0; b < TCO; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // fragment addressing: lane l of 16-lane group g addresses segment (row 8 (g >> 1) + (l15 >> 2), channels 16 (g & 1) + 4 (l15 & 3))
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)lds;
    const unsigned a_lane = (8 * (g >> 1) + (l15 >> 2)) * PA + (16 * (g & 1) + 4 * (l15 & 3)) * 2;
    const unsigned b_lane_off = (8 * (g >> 1) + (l15 >> 2)) * PB + (16 * (g & 1) + 4 * (l15 & 3)) * 2;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks) fetch(c + 1);
        if (active) {
            const unsigned base = lds0 + (c & 1) * BUF;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                nb_bf8 ah[TCI], al[TCI], bh[TCO], bl[TCO];
#pragma unroll
                for (int a = 0; a < TCI; ++a) {
                    const unsigned ad = base + a_lane + kc * 16 * PA + ((wci * TCI + a) * 32) * 2;
                    ah[a] = tr_frag(ad, ad + 4 * PA);
                    al[a] = tr_frag(ad + A_BYTES, ad + A_BYTES + 4 * PA);
                }
#pragma unroll
                for (int b = 0; b < TCO; ++b) {
                    const unsigned bd = base + 2 * A_BYTES + b_lane_off + kc * 16 * PB + ((wco * TCO + b) * 32) * 2;
                    bh[b] = tr_frag(bd, bd + 4 * PB);
                    bl[b] = tr_frag(bd + B_BYTES, bd + B_BYTES + 4 * PB);
                }
                // the reads above are inline asm: the compiler does not count them, so the wait is tied to every fragment register
                if constexpr (TCI == 2 && TCO == 2)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(ah[1]), "+v"(al[1]), "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]));
                else if constexpr (TCI == 1 && TCO == 2)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]));
                else if constexpr (TCI == 2 && TCO == 1)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(ah[1]), "+v"(al[1]), "+v"(bh[0]), "+v"(bl[0]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]));
#pragma unroll
                for (int a = 0; a < TCI; ++a)
#pragma unroll
                    for (int b = 0; b < TCO; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    }
            }
        }
        if (c + 1 < n_chunks) stash((c + 1) & 1);  // the other buffer: its readers passed the barrier that ended chunk c - 1
        __syncthreads();
    }
    if (!active) return;
    // D fragment: lane (j = output channel, hi) holds input channels tile_row(r, hi)
    const int j = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int a = 0; a < TCI; ++a)
#pragma unroll
        for (int b = 0; b < TCO; ++b) {
            const int co = (wco * TCO + b) * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (wci * TCI + a) * 32 + tile_row(r, hi);
                atomicAdd(&dw[((size_t)o * CIN + ci) * COUT + co], acc[a][b][r]);
            }
        }
}

}  // namespace

extern "C" {

int nb_enc_bn_relu_bwd(const float *dy, const float *y, const float *x, const int32_t *n_rows, int32_t n_rows_max,
                       int32_t c, const float *batch_stats, float eps, const float *gamma, double *sums, float *dx,
                       float *dgamma, float *dbeta, uint16_t *dx_split, int32_t flags, void *stream) {
    NB_REQUIRE(dy && y && x && n_rows && batch_stats && gamma && sums && dx && dgamma && dbeta,
               "nb_enc_bn_relu_bwd: NULL pointer");
    NB_REQUIRE(c > 0 && n_rows_max >= 0, "nb_enc_bn_relu_bwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & NB_BWD_ZEROED)) NB_HIP(hipMemsetAsync(sums, 0, 2 * (size_t)c * sizeof(double), st));
    // one block per 256 rows of CAPACITY, at most 256 slabs (the live rows are split evenly over whatever the grid holds)
    const int slabs = n_rows_max <= 256 ? 1 : (int)(nb_ceil_div(n_rows_max, 256) < 256 ? nb_ceil_div(n_rows_max, 256) : 256);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb_ceil_div(c, 64), slabs), dim3(1024), 0, st, dy, y, x, n_rows, c,
                       batch_stats, eps, sums);
    const long long total = (long long)n_rows_max * c;
    const long long threads = total > c ? total : c;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nb_ceil_div(threads, 256)), dim3(256), 0, st, dy, y, x, n_rows, c,
                       batch_stats, eps, gamma, sums, dx, dgamma, dbeta, reinterpret_cast<__bf16 *>(dx_split), total);
    NB_CHECK_LAUNCH("nb_enc_bn_relu_bwd");
    return NB_OK;
}

#define NB_FOR_CONV_SHAPES(X) X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 128) X(128, 128)

int nb_enc_conv_bwd_input(const float *dx, const int32_t *out_grid, const int32_t out_dhw[3], const int32_t *in_lin,
                          const int32_t *n_in, int32_t n_in_max, const int32_t in_dhw[3], int32_t stride,
                          const float *weight, int32_t cin, int32_t cout, float *din, void *stream) {
    NB_REQUIRE(dx && out_grid && out_dhw && in_lin && n_in && in_dhw && weight && din, "nb_enc_conv_bwd_input: NULL pointer");
    NB_REQUIRE(stride == 1 || stride == 2, "nb_enc_conv_bwd_input: stride %d", stride);
    if (n_in_max <= 0) return NB_OK;
    const Dims go = {out_dhw[0], out_dhw[1], out_dhw[2]}, gi = {in_dhw[0], in_dhw[1], in_dhw[2]};
    hipStream_t st = (hipStream_t)stream;
#define X(CI, CO)                                                                                                     \
    if (cin == CI && cout == CO) {                                                                                    \
        hipLaunchKernelGGL((conv_bwd_in_kernel<CI, CO>), dim3(nb_ceil_div(n_in_max, 128), (CI + 31) / 32), dim3(256), 0, \
                           st, dx, out_grid, go, in_lin, n_in, gi, stride, weight, din);                              \
        NB_CHECK_LAUNCH("nb_enc_conv_bwd_input");                                                                     \
        return NB_OK;                                                                                                 \
    }
    NB_FOR_CONV_SHAPES(X)
#undef X
    nb_set_error("nb_enc_conv_bwd_input: unsupported channel pair %d -> %d", cin, cout);
    return NB_EINVAL;
}

int nb_enc_conv_bwd_weight(const float *in_rows, const int32_t *in_grid, const int32_t in_dhw[3], const int32_t *out_lin,
                           const int32_t *n_out, int32_t n_out_max, const int32_t out_dhw[3], int32_t stride,
                           const float *dx, const uint16_t *dx_split, int32_t cin, int32_t cout, float *dweight,
                           int32_t *rulebook, int32_t flags, void *stream) {
    NB_REQUIRE(in_rows && in_grid && in_dhw && out_lin && n_out && out_dhw && dx && dweight,
               "nb_enc_conv_bwd_weight: NULL pointer");
    NB_REQUIRE(stride == 1 || stride == 2, "nb_enc_conv_bwd_weight: stride %d", stride);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & NB_BWD_ZEROED)) NB_HIP(hipMemsetAsync(dweight, 0, (size_t)27 * cin * cout * sizeof(float), st));
    if (n_out_max <= 0) return NB_OK;
    NB_REQUIRE(rulebook != nullptr, "nb_enc_conv_bwd_weight: rulebook scratch is NULL");
    const Dims go = {out_dhw[0], out_dhw[1], out_dhw[2]}, gi = {in_dhw[0], in_dhw[1], in_dhw[2]};
    if (!(flags & NB_BWD_RULEBOOK_READY))
        hipLaunchKernelGGL(conv_rulebook_kernel, dim3(nb_ceil_div((long long)n_out_max * 27, 256)), dim3(256), 0, st, in_grid, gi,
                           out_lin, n_out, go, stride, rulebook);
    if (dx_split && cin >= 32) {  // bf16 pairs of dx given: the matrix-pipe kernel (the 16-channel layers stay exact fp32)
        const long long plane = (long long)n_out_max * cout;
#define X16(CI, CO)                                                                                                   \
    if (cin == CI && cout == CO) {                                                                                    \
        hipLaunchKernelGGL((conv_bwd_w16_kernel<CI, CO>), dim3(27, (unsigned)nb_ceil_div(n_out_max, BW_ROWS)), dim3(256), 0, st, \
                           in_rows, rulebook, n_out, dx_split, plane, dweight);                                       \
        NB_CHECK_LAUNCH("nb_enc_conv_bwd_weight");                                                                    \
        return NB_OK;                                                                                                 \
    }
        X16(32, 32) X16(32, 64) X16(64, 64) X16(64, 128) X16(128, 128)
#undef X16
    }
#define X(CI, CO)                                                                                                     \
    if (cin == CI && cout == CO) {                                                                                    \
        hipLaunchKernelGGL((conv_bwd_w_kernel<CI, CO>),                                                               \
                           dim3(nb_ceil_div(n_out_max, 4 * ROWS_PER_WAVE), 27, ((CI + 31) / 32) * ((CO + 31) / 32)),  \
                           dim3(256), 0, st, in_rows, rulebook, n_out, dx, dweight);                                   \
        NB_CHECK_LAUNCH("nb_enc_conv_bwd_weight");                                                                    \
        return NB_OK;                                                                                                 \
    }
    NB_FOR_CONV_SHAPES(X)
#undef X
    nb_set_error("nb_enc_conv_bwd_weight: unsupported channel pair %d -> %d", cin, cout);
    return NB_EINVAL;
}

int nb_enc_scatter_codes_bwd(const float *drows, const int32_t *rows_vert, const int32_t *n_rows, int32_t n_rows_max,
                             int32_t c, float *dcodes, void *stream) {
    NB_REQUIRE(drows && rows_vert && n_rows && dcodes, "nb_enc_scatter_codes_bwd: NULL pointer");
    if (n_rows_max <= 0) return NB_OK;
    hipLaunchKernelGGL(scatter_codes_bwd_kernel, dim3(nb_ceil_div((long long)n_rows_max * c, 256)), dim3(256), 0,
                       (hipStream_t)stream, drows, rows_vert, n_rows, c, dcodes);
    NB_CHECK_LAUNCH("nb_enc_scatter_codes_bwd");
    return NB_OK;
}

}  // extern "C"
