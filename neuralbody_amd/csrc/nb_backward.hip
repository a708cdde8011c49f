// nb_backward.hip — backward pass of the decoder side of the hot path (training step, SURVEY.md §8 row a15:
// lib/train/trainers/if_nerf_clight.py:18-36 calls Renderer.render under autograd; the reference's backward is
// PyTorch autograd through raw2outputs, the Conv1d stack, F.grid_sample and spconv).
//
//   nb_composite_bwd   d(rgb_map, acc_map, depth_map) -> d raw            (raw2outputs, nerf_net_utils.py:19-49)
//   nb_sgemm / nb_gemm_fused   row-major fp32 GEMMs of the MLP backward on v_mfma_f32_32x32x2_f32 (exact fp32):
//                      dX = dY.W with the ReLU mask and the bias-gradient column sums in the epilogue (gemm_rows_kernel),
//                      dW = dY^T.X over the N = rays x samples rows with a split-K kernel (gemm_tn_kernel)
//   nb_relu_bwd / nb_colsum   elementwise mask and bias-gradient reductions (stand-alone forms)
//   nb_trilinear_bwd   dF [N,352] -> gradients of the ACTIVE voxel rows of the four feature volumes
//                      (grid_sample backward restricted to active voxels: inactive sites are constants)
#include <stdlib.h>

#include "nb_march_common.h"
#include "nb_trread.h"

using namespace nbm;

namespace {

// ------------------------------------------------------------------ compositing backward
// One thread per ray, two sweeps.  Forward sweep recomputes alpha_i, T_i and stashes T_i in d_raw[...,3];
// backward sweep carries B_i = sum_{k>i} w_k dw_k:
//   dw_i     = g . c_i (+ d_acc + d_depth z_i - [white_bkgd] sum(g))
//   dc_i     = w_i g                         -> d raw_rgb = dc (.) c (1 - c)
//   dalpha_i = T_i dw_i - B_i / (1 - alpha_i + 1e-10)
//   dsigma_i = dalpha_i dist_i (1 - alpha_i) [sigma_i > 0]
__global__ void composite_bwd_kernel(const float *__restrict__ raw, const float *__restrict__ z,
                                     const float *__restrict__ ray_d, long long n_rays, int S, int white_bkgd,
                                     const float *__restrict__ d_rgb, const float *__restrict__ d_acc,
                                     const float *__restrict__ d_depth, float *__restrict__ d_raw) {
    const long long ray = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= n_rays) return;
    const float dx = ray_d[ray * 3], dy = ray_d[ray * 3 + 1], dz = ray_d[ray * 3 + 2];
    const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    const float g0 = d_rgb[ray * 3], g1 = d_rgb[ray * 3 + 1], g2 = d_rgb[ray * 3 + 2];
    const float ga = d_acc ? d_acc[ray] : 0.f, gd = d_depth ? d_depth[ray] : 0.f;
    const float gw = white_bkgd ? -(g0 + g1 + g2) : 0.f;
    const float *r = raw + ray * S * 4;
    const float *zz = z + ray * S;
    float *o = d_raw + ray * S * 4;
    auto dist_of = [&](int s) { return __fmul_rn((s + 1 < S) ? __fsub_rn(zz[s + 1], zz[s]) : 1e10f, dn); };
    float T = 1.f;
    for (int s = 0; s < S; ++s) {
        const float alpha = 1.f - expf(-fmaxf(r[s * 4 + 3], 0.f) * dist_of(s));
        o[s * 4 + 3] = T;
        T = T * (__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f));
    }
    float B = 0.f;
    for (int s = S - 1; s >= 0; --s) {
        const float sig = r[s * 4 + 3], dist = dist_of(s);
        const float e = expf(-fmaxf(sig, 0.f) * dist);  // 1 - alpha
        const float alpha = 1.f - e;
        const float Ts = o[s * 4 + 3];
        const float w = alpha * Ts;
        const float c0 = 1.f / (1.f + expf(-r[s * 4])), c1 = 1.f / (1.f + expf(-r[s * 4 + 1])),
                    c2 = 1.f / (1.f + expf(-r[s * 4 + 2]));
        const float dw = g0 * c0 + g1 * c1 + g2 * c2 + ga + gd * zz[s] + gw;
        o[s * 4 + 0] = w * g0 * c0 * (1.f - c0);
        o[s * 4 + 1] = w * g1 * c1 * (1.f - c1);
        o[s * 4 + 2] = w * g2 * c2 * (1.f - c2);
        const float dalpha = Ts * dw - B / (__fadd_rn(e, 1e-10f));
        o[s * 4 + 3] = sig > 0.f ? dalpha * dist * e : 0.f;
        B += w * dw;
    }
}

__global__ void relu_bwd_kernel(float *__restrict__ dy, const float *__restrict__ y, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        f32x4 d = *reinterpret_cast<f32x4 *>(dy + i);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(y + i);
        d.x = v.x > 0.f ? d.x : 0.f;
        d.y = v.y > 0.f ? d.y : 0.f;
        d.z = v.z > 0.f ? d.z : 0.f;
        d.w = v.w > 0.f ? d.w : 0.f;
        *reinterpret_cast<f32x4 *>(dy + i) = d;
    } else {
        for (long long k = i; k < n; ++k) dy[k] = y[k] > 0.f ? dy[k] : 0.f;
    }
}

// out[c] (+)= sum_r x[r * ld + c]   — one block per 64 columns x a slab of rows, fp64 partials, float atomics
__global__ void colsum_kernel(const float *__restrict__ x, long long n_rows, int n_cols, int ld, float *__restrict__ out) {
    __shared__ double part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const long long rows_per_block = (n_rows + gridDim.y - 1) / gridDim.y;
    const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
    double s = 0.0;
    if (c < n_cols)
        for (long long r = r0 + w; r < r1; r += 4) s += (double)x[r * ld + c];
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < n_cols) atomicAdd(&out[c], (float)(part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// ------------------------------------------------------------------ trilinear (grid_sample) backward
struct TriArgs {
    SceneDev sc;              // vol[] unused
    const int *grid[4];       // index grid of each dense level (row id or -1)
    float *drows[4];          // gradient of the active rows [n_rows_l, C_l], accumulated with atomics
    const float *wpts;        // [n,3]
    const float *dF;          // [n,352]
    long long n;
    int run;                  // consecutive points [k run, (k+1) run) are the samples of one ray
};

// One thread per (ray, 4-channel group): 88 groups = 8 + 16 + 32 + 32, walking the ray's `run` samples.  Consecutive samples
// of a ray mostly share their base voxel on the coarse levels (sample spacing ~1.5 cm against 4 and 8 cm voxels), so the
// eight corner contributions are summed in registers and leave as atomics only when the base voxel changes: round 2 issued one
// atomic per (sample, channel, corner) — 184 M per training step, with hundreds of them contending for each coarse voxel.
__global__ void trilinear_bwd_kernel(TriArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long ray = t / 88;
    const long long p0 = ray * a.run;
    if (p0 >= a.n) return;
    const int g = (int)(t % 88);
    const int L = g < 8 ? 0 : (g < 24 ? 1 : (g < 56 ? 2 : 3));
    const int q = g - (L == 0 ? 0 : (L == 1 ? 8 : (L == 2 ? 24 : 56)));
    const int C = lvl_c(L);
    const int D = a.sc.dhw[L][0], H = a.sc.dhw[L][1], W = a.sc.dhw[L][2];
    const long long p1 = p0 + a.run < a.n ? p0 + a.run : a.n;
    int bx = 0x7fffffff, by = 0, bz = 0;  // base voxel of the contributions held in acc
    f32x4 acc[8];
    auto flush = [&]() {
        if (bx == 0x7fffffff) return;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int xx = bx + (corner & 1), yy = by + ((corner >> 1) & 1), zz = bz + (corner >> 2);
            if ((unsigned)xx >= (unsigned)W || (unsigned)yy >= (unsigned)H || (unsigned)zz >= (unsigned)D) continue;
            const int row = a.grid[L][((long long)zz * H + yy) * W + xx];
            if (row < 0) continue;  // inactive voxel: a constant zero of .dense(), no parameter behind it
            const f32x4 v = acc[corner];
            if (v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f) continue;
            float *dst = a.drows[L] + (size_t)row * C + q * 4;
            atomicAdd(dst + 0, v.x);
            atomicAdd(dst + 1, v.y);
            atomicAdd(dst + 2, v.z);
            atomicAdd(dst + 3, v.w);
        }
    };
    for (long long pt = p0; pt < p1; ++pt) {
        const f32x4 d = *reinterpret_cast<const f32x4 *>(a.dF + pt * 352 + lvl_chan_base(L) + q * 4);
        if (d.x == 0.f && d.y == 0.f && d.z == 0.f && d.w == 0.f) continue;
        const GridCoord gc = grid_coords(a.sc, a.wpts[pt * 3], a.wpts[pt * 3 + 1], a.wpts[pt * 3 + 2]);
        const float ix = unnorm_clamped(gc.gw, W), iy = unnorm_clamped(gc.gh, H), iz = unnorm_clamped(gc.gd, D);
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
        if (x0 != bx || y0 != by || z0 != bz) {
            flush();
            bx = x0, by = y0, bz = z0;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) acc[corner] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float wx[2] = {(fx + 1.f) - ix, ix - fx}, wy[2] = {(fy + 1.f) - iy, iy - fy}, wz[2] = {(fz + 1.f) - iz, iz - fz};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float w = (wx[corner & 1] * wy[(corner >> 1) & 1]) * wz[corner >> 2];
            acc[corner].x = fmaf(w, d.x, acc[corner].x);
            acc[corner].y = fmaf(w, d.y, acc[corner].y);
            acc[corner].z = fmaf(w, d.z, acc[corner].z);
            acc[corner].w = fmaf(w, d.w, acc[corner].w);
        }
    }
    flush();
}

// ------------------------------------------------------------------ C[m,n] += A^T B over many rows (weight gradients)
// dW = dY^T X has M, N <= 384 but K = rays x samples (65 536 per training step): rocBLAS picks a single-pass macro
// tile for it (1.8 ms per GEMM measured).  Split the row range instead: every wave owns (a 256-row chunk, a 32x32 tile
// of C), accumulates it with v_mfma_f32_32x32x2_f32 (one row pair per MFMA) and adds it to C with fp32 atomics.
constexpr int TN_ROWS = 256;
typedef float f32x16_ __attribute__((ext_vector_type(16)));
__host__ __device__ constexpr int tn_tile_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

#ifndef NB_TN_WAVES
#define NB_TN_WAVES 4
#endif
constexpr int TN_WAVES = NB_TN_WAVES;  // waves (= consecutive row chunks of one tile of C) per workgroup
__global__ __launch_bounds__(64 * TN_WAVES) void gemm_tn_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                      int ldb, long long R, int M, int N, float alpha,
                                                      float *__restrict__ C, int ldc) {
    // the waves of a workgroup take consecutive row chunks of the SAME tile of C: their partial tiles are summed through LDS
    // (wave order: deterministic inside the workgroup) and leave as one atomic per element per workgroup instead of one per
    // wave — the atomics were a third of this kernel's time (23 M per 65 536 x 256 x 352 product)
    __shared__ float red[TN_WAVES][16][64];
    const int lane = threadIdx.x & 63, i = lane & 31, hi = lane >> 5, wv = threadIdx.x >> 6;
    const long long row0 = ((long long)blockIdx.x * TN_WAVES + wv) * TN_ROWS;
    const int am = blockIdx.y * 32 + i, bn = blockIdx.z * 32 + i;
    const bool aok = am < M, bok = bn < N;
    f32x16_ acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (row0 < R) {  // wave-uniform
#pragma unroll 8
        for (int m = 0; m < TN_ROWS / 2; ++m) {
            const long long row = row0 + 2 * m + hi;
            const bool rok = row < R;
            const float a = (rok && aok) ? A[row * lda + am] : 0.f;
            const float b = (rok && bok) ? B[row * ldb + bn] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv][r][lane] = acc[r];
    __syncthreads();
    if (bok) {
#pragma unroll
        for (int q = 0; q < 16 / TN_WAVES; ++q) {
            const int r = (16 / TN_WAVES) * wv + q;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < TN_WAVES; ++w) v += red[w][r][lane];
            const int cm = blockIdx.y * 32 + tn_tile_row(r, hi);
            if (cm < M) atomicAdd(&C[(size_t)cm * ldc + bn], alpha * v);
        }
    }
}

// The same product on the 16-bit matrix pipe: both operands as bf16 head + remainder (three products per K = 16 chunk, fp32
// accumulate, ~2^-16 relative), the recipe of conv_bwd_w16_kernel (nb_encoder_bwd.hip): the reduction runs over ROWS, so both
// MFMA operands are K-major; a workgroup stages 32 rows x 128 columns of A and of B row-major in LDS (fp32 -> bf16 pairs on
// the way) and reads the fragments with ds_read_b64_tr_b16 (nb_trread.h).  One workgroup = a 128 x 128 tile of C over
// TN16_ROWS rows, wave w a 2 x 2 block of its 32 x 32 tiles; the next chunk's loads are issued before the current chunk's
// MFMAs.  Ragged M / N: columns beyond the matrix are staged as zeros and not written back.
constexpr int TN16_ROWS = 1024;
constexpr int TN16_P = 128 * 2 + 32;            // LDS row pitch in bytes
constexpr int TN16_PLANE = 32 * TN16_P;         // one bf16 plane of one operand
constexpr int TN16_BUF = 4 * TN16_PLANE;        // A head | A remainder | B head | B remainder

__global__ __launch_bounds__(256, 2) void gemm_tn16_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                        int ldb, long long R, int M, int N, float alpha,
                                                        float *__restrict__ C, int ldc) {
    __shared__ __attribute__((aligned(16))) char lds[2 * TN16_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const long long row_begin = (long long)blockIdx.z * TN16_ROWS;
    const long long row_end = row_begin + TN16_ROWS < R ? row_begin + TN16_ROWS : R;
    const int n_chunks = (int)((row_end - row_begin + 31) / 32);
    const int wm = wv >> 1, wn = wv & 1;  // the wave's 2 x 2 block of tiles
    // a wave whose block lies outside the matrix has nothing to multiply (it still stages)
    const bool active = m0 + wm * 64 < M && n0 + wn * 64 < N;
    const int sr = tid >> 3, sp = tid & 7;  // staging: 8 threads per row, 16 columns each
    f32x4 ra[4], rb[4];
    auto fetch_one = [&](const float *__restrict__ P, int ld, int c0, int lim, long long row, f32x4 (&r)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + sp * 16 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < row_end) {
                const float *src = P + row * ld + c;
                if (c + 3 < lim) v = *reinterpret_cast<const f32x4 *>(src);
                else {
                    if (c < lim) v.x = src[0];
                    if (c + 1 < lim) v.y = src[1];
                    if (c + 2 < lim) v.z = src[2];
                }
            }
            r[q] = v;
        }
    };
    auto fetch = [&](int chunk) {
        const long long row = row_begin + chunk * 32 + sr;
        fetch_one(A, lda, m0, M, row, ra);
        fetch_one(B, ldb, n0, N, row, rb);
    };
    auto stash_one = [&](char *hp, const f32x4 (&r)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
            unsigned short h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) nbtr::split_bf16(v[e], h[e], l[e]);
            const int cb = (sp * 16 + 4 * q) * 2;
            *reinterpret_cast<uint2 *>(hp + sr * TN16_P + cb) = uint2{h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16)};
            *reinterpret_cast<uint2 *>(hp + TN16_PLANE + sr * TN16_P + cb) = uint2{l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16)};
        }
    };
    auto stash = [&](int buf) {
        stash_one(lds + buf * TN16_BUF, ra);
        stash_one(lds + buf * TN16_BUF + 2 * TN16_PLANE, rb);
    };
    f32x16_ acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const unsigned lds0 = (unsigned)(size_t)lds, loff = nbtr::lane_offset(lane, TN16_P);
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks) fetch(c + 1);
        if (active) {
            const unsigned base = lds0 + (c & 1) * TN16_BUF + loff;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                nbtr::bf8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned ad = base + kc * 16 * TN16_P + ((wm * 2 + t) * 32) * 2;
                    ah[t] = nbtr::frag(ad, ad + 4 * TN16_P);
                    al[t] = nbtr::frag(ad + TN16_PLANE, ad + TN16_PLANE + 4 * TN16_P);
                    const unsigned bd = base + 2 * TN16_PLANE + kc * 16 * TN16_P + ((wn * 2 + t) * 32) * 2;
                    bh[t] = nbtr::frag(bd, bd + 4 * TN16_P);
                    bl[t] = nbtr::frag(bd + TN16_PLANE, bd + TN16_PLANE + 4 * TN16_P);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]), "+v"(ah[1]), "+v"(al[1]), "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]));
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
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
    const int j = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int cn = n0 + (wn * 2 + b) * 32 + j;
            if (cn >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cm = m0 + (wm * 2 + a) * 32 + tn_tile_row(r, hi);
                if (cm < M) atomicAdd(&C[(size_t)cm * ldc + cn], alpha * acc[a][b][r]);
            }
        }
}

__global__ void scale_matrix_kernel(float *__restrict__ C, int m, int n, int ldc, float beta) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)m * n) return;
    float *p = C + (idx / n) * ldc + idx % n;
    *p = beta == 0.f ? 0.f : *p * beta;
}

// ------------------------------------------------------------------ C[R,N] = alpha A[R,K] op(B) + beta C, R large, K and N small
// (dX = dY.W of the MLP backward: R = rays x samples rows, K, N <= 384).  One wave = 32 rows x up to 4 column tiles of 32;
// v_mfma_f32_32x32x2_f32 (exact fp32): A operand = one float per lane (row lane % 32, k = 2 c + lane / 32), B operand = the
// weight element (k, column), C/D in the standard 32x32 layout.  Fused epilogues of the backward chain:
//   mask_y   : C *= (Y[row, col] > 0)   — the ReLU that followed the layer whose input gradient this is
//   colsum   : colsum[col] += sum_rows C (after the mask) — the bias gradient of that layer
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                        int trans_b, long long R, int K, int N, float alpha, float beta,
                                                        float *__restrict__ C, int ldc, const float *__restrict__ mask_y, int ldy,
                                                        float *__restrict__ colsum) {
    // column sums: one fp32 atomic per column per WORKGROUP (LDS reduction over the four waves first) — one per wave made
    // 4096 atomics per address per product and dominated the kernel
    __shared__ float cs_lds[128];
    if (threadIdx.x < 128) cs_lds[threadIdx.x] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    const int col0 = blockIdx.y * 128;
    const long long arow = row0 + i;
    const bool rok = arow < R;
    const float *ap = A + (rok ? arow : 0) * lda;
    f32x16_ acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int nt = row0 < R ? min(4, (N - col0 + 31) / 32) : 0;  // wave-uniform
    for (int k = 0; k < K && nt > 0; k += 2) {
        const int ke = k + kk;
        const bool kok = ke < K;
        const float a = (rok && kok) ? ap[ke] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < nt) {
                const int col = col0 + 32 * t + i;
                float b = 0.f;
                if (kok && col < N) b = trans_b ? B[(size_t)col * ldb + ke] : B[(size_t)ke * ldb + col];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= nt) continue;
        const int col = col0 + 32 * t + i;
        if (col >= N) continue;
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = row0 + tn_tile_row(r, kk);
            if (row >= R) continue;
            float v = alpha * acc[t][r];
            float *cp = C + row * ldc + col;
            if (beta != 0.f) v += beta * *cp;
            if (mask_y && !(mask_y[row * ldy + col] > 0.f)) v = 0.f;
            *cp = v;
            cs += v;
        }
        if (colsum) atomicAdd(&cs_lds[32 * t + i], cs);
    }
    if (colsum) {
        __syncthreads();
        if (threadIdx.x < 128 && col0 + (int)threadIdx.x < N) atomicAdd(&colsum[col0 + threadIdx.x], cs_lds[threadIdx.x]);
    }
}

// fast path of gemm_rows_kernel (K a multiple of 16, 16-byte aligned operands): a 128-row x 128-column block per workgroup.
// The weight panel B[k0 .. k0+15][128 columns] is shared by the four waves through LDS (double buffered, one barrier per
// 16-wide K chunk); every lane reads ITS row's 16 K values of A as one 64-byte cache line straight into registers.  (The
// first version loaded 8 K per step per wave with no sharing: 128 half-used cache-line requests per 16 MFMAs per wave — 5x
// over what the vector L1 can serve; measured 320 us for the 65536 x 256 x 256 product against a 62 us matrix-pipe bound.)
template <bool TRANS_B>
__global__ __launch_bounds__(256, 2) void gemm_rows_fast_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                             long long R, int K, int N, float alpha, float beta, float *__restrict__ C,
                                                             int ldc, const float *__restrict__ mask_y, int ldy,
                                                             float *__restrict__ colsum) {
    constexpr int KC = 16, LDB = 132;  // LDS row stride (floats): 128 + 4 keeps the transposed stores conflict-light
    __shared__ __attribute__((aligned(16))) float bs[2][KC * LDB];
    __shared__ float cs_lds[128];
    if (threadIdx.x < 128) cs_lds[threadIdx.x] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, kk = lane >> 5, wave = tid >> 6;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * 32;
    const int col0 = blockIdx.y * 128;
    const long long arow = min(row0 + i, R - 1);  // rows past the end compute values that are never stored
    const float *ap = A + arow * lda;
    f32x16_ acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // cooperative load of one B chunk: 16 x 128 floats = 512 float4, two per thread
    f32x4 breg[2];
    auto load_b = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = tid + 256 * h;
            if (TRANS_B) {  // B[col][k]: thread -> (col = q / 4, four consecutive k)
                const int col = min(col0 + (q >> 2), N - 1);
                breg[h] = *reinterpret_cast<const f32x4 *>(B + (size_t)col * ldb + k0 + 4 * (q & 3));
            } else {  // B[k][col]: thread -> (k = q / 32, four consecutive columns)
                const int col = col0 + 4 * (q & 31);
                const float *bp = B + (size_t)(k0 + (q >> 5)) * ldb;
                if (col + 3 < N) breg[h] = *reinterpret_cast<const f32x4 *>(bp + col);  // ldb % 4 == 0 and col0 % 4 == 0 on this path
                else breg[h] = f32x4{col < N ? bp[col] : 0.f, col + 1 < N ? bp[col + 1] : 0.f, col + 2 < N ? bp[col + 2] : 0.f, 0.f};
            }
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = tid + 256 * h;
            if (TRANS_B) {
                const int c = q >> 2, k4 = 4 * (q & 3);
                bs[buf][(k4 + 0) * LDB + c] = breg[h].x;
                bs[buf][(k4 + 1) * LDB + c] = breg[h].y;
                bs[buf][(k4 + 2) * LDB + c] = breg[h].z;
                bs[buf][(k4 + 3) * LDB + c] = breg[h].w;
            } else {
                *reinterpret_cast<f32x4 *>(&bs[buf][(q >> 5) * LDB + 4 * (q & 31)]) = breg[h];
            }
        }
    };
    f32x4 a[4];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = *reinterpret_cast<const f32x4 *>(ap + k0 + 4 * c);
    };
    load_b(0);
    load_a(0);
    store_b(0);
    __syncthreads();
    const int nchunk = K / KC;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        f32x4 an[4] = {a[0], a[1], a[2], a[3]};
        if (ch + 1 < nchunk) {
            load_b((ch + 1) * KC);
#pragma unroll
            for (int c = 0; c < 4; ++c) an[c] = *reinterpret_cast<const f32x4 *>(ap + (ch + 1) * KC + 4 * c);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {  // k = 2 c + kk
            const f32x4 q = a[c >> 1];
            const float av = (c & 1) ? (kk ? q.w : q.z) : (kk ? q.y : q.x);
            const float *bp = &bs[buf][(2 * c + kk) * LDB + i];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[32 * t], acc[t], 0, 0, 0);
        }
        if (ch + 1 < nchunk) store_b(buf ^ 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = an[c];
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = col0 + 32 * t + i;
        if (col >= N || row0 >= R) continue;
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = row0 + tn_tile_row(r, kk);
            if (row >= R) continue;
            float v = alpha * acc[t][r];
            float *cp = C + row * ldc + col;
            if (beta != 0.f) v += beta * *cp;
            if (mask_y && !(mask_y[row * ldy + col] > 0.f)) v = 0.f;
            *cp = v;
            cs += v;
        }
        if (colsum) atomicAdd(&cs_lds[32 * t + i], cs);
    }
    if (colsum) {
        __syncthreads();
        if (threadIdx.x < 128 && col0 + (int)threadIdx.x < N) atomicAdd(&colsum[col0 + threadIdx.x], cs_lds[threadIdx.x]);
    }
}

// gemm_rows_fast_kernel (op(B) = B) on the 16-bit matrix pipe, both operands as bf16 head + remainder pairs: the `dX = dY . W`
// products of the MLP backward.  A fragment: a lane's row, 8 consecutive K values = two float4 straight from global memory, split
// in registers.  B fragment: K-major (8 consecutive K of one column) from a [32 K][128 columns] LDS tile of bf16 pairs by
// ds_read_b64_tr_b16 (nb_trread.h), staged once per workgroup and K chunk.  Same tile, same fused epilogue as the fp32 kernel.
__global__ __launch_bounds__(256, 2) void gemm_rows16_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                          long long R, int K, int N, float alpha, float beta, float *__restrict__ C,
                                                          int ldc, const float *__restrict__ mask_y, int ldy,
                                                          float *__restrict__ colsum) {
    constexpr int KC = 32;
    __shared__ __attribute__((aligned(16))) char bs[2][2 * TN16_PLANE];  // [buffer][head plane | remainder plane], 32 rows of pitch TN16_P
    __shared__ float cs_lds[128];
    if (threadIdx.x < 128) cs_lds[threadIdx.x] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, kk = lane >> 5, wave = tid >> 6;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * 32;
    const int col0 = blockIdx.y * 128;
    const long long arow = min(row0 + i, R - 1);  // rows past the end compute values that are never stored
    const float *ap = A + arow * lda;
    f32x16_ acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // cooperative load of one B chunk: 32 x 128 floats: thread -> (k = tid / 8, 16 consecutive columns)
    const int sr = tid >> 3, sp = tid & 7;
    f32x4 breg[4];
    auto load_b = [&](int k0) {
        const float *bp = B + (size_t)(k0 + sr) * ldb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = col0 + sp * 16 + 4 * q;
            if (col + 3 < N) breg[q] = *reinterpret_cast<const f32x4 *>(bp + col);  // ldb % 4 == 0 and col0 % 4 == 0 on this path
            else breg[q] = f32x4{col < N ? bp[col] : 0.f, col + 1 < N ? bp[col + 1] : 0.f, col + 2 < N ? bp[col + 2] : 0.f, 0.f};
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[4] = {breg[q].x, breg[q].y, breg[q].z, breg[q].w};
            unsigned short h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) nbtr::split_bf16(v[e], h[e], l[e]);
            const int cb = (sp * 16 + 4 * q) * 2;
            *reinterpret_cast<uint2 *>(bs[buf] + sr * TN16_P + cb) = uint2{h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16)};
            *reinterpret_cast<uint2 *>(bs[buf] + TN16_PLANE + sr * TN16_P + cb) = uint2{l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16)};
        }
    };
    // this lane's A values of one chunk: K = 16 kc + 8 kk + 0..7, kc = 0, 1
    f32x4 a[4];
    auto load_a = [&](int k0, f32x4 (&d)[4]) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            d[2 * kc] = *reinterpret_cast<const f32x4 *>(ap + k0 + 16 * kc + 8 * kk);
            d[2 * kc + 1] = *reinterpret_cast<const f32x4 *>(ap + k0 + 16 * kc + 8 * kk + 4);
        }
    };
    auto split8 = [&](const f32x4 lo4, const f32x4 hi4, nbtr::bf8 &h, nbtr::bf8 &l) {
        const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
        nbtr::s8 hs, ls;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned short hh, ll;
            nbtr::split_bf16(v[e], hh, ll);
            hs[e] = (short)hh;
            ls[e] = (short)ll;
        }
        h = __builtin_bit_cast(nbtr::bf8, hs);
        l = __builtin_bit_cast(nbtr::bf8, ls);
    };
    const unsigned bs0 = (unsigned)(size_t)&bs[0][0], loff = nbtr::lane_offset(lane, TN16_P);
    load_b(0);
    load_a(0, a);
    store_b(0);
    __syncthreads();
    const int nchunk = K / KC;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        f32x4 an[4] = {a[0], a[1], a[2], a[3]};
        if (ch + 1 < nchunk) {
            load_b((ch + 1) * KC);
            load_a((ch + 1) * KC, an);
        }
        const unsigned base = bs0 + buf * (2 * TN16_PLANE) + loff;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            nbtr::bf8 ah, al, bh[4], bl[4];
            split8(a[2 * kc], a[2 * kc + 1], ah, al);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned bd = base + kc * 16 * TN16_P + (t * 32) * 2;
                bh[t] = nbtr::frag(bd, bd + 4 * TN16_P);
                bl[t] = nbtr::frag(bd + TN16_PLANE, bd + TN16_PLANE + 4 * TN16_P);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]), "+v"(bh[2]), "+v"(bl[2]), "+v"(bh[3]), "+v"(bl[3]));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunk) store_b(buf ^ 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = an[c];
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = col0 + 32 * t + i;
        if (col >= N || row0 >= R) continue;
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = row0 + tn_tile_row(r, kk);
            if (row >= R) continue;
            float v = alpha * acc[t][r];
            float *cp = C + row * ldc + col;
            if (beta != 0.f) v += beta * *cp;
            if (mask_y && !(mask_y[row * ldy + col] > 0.f)) v = 0.f;
            *cp = v;
            cs += v;
        }
        if (colsum) atomicAdd(&cs_lds[32 * t + i], cs);
    }
    if (colsum) {
        __syncthreads();
        if (threadIdx.x < 128 && col0 + (int)threadIdx.x < N) atomicAdd(&colsum[col0 + threadIdx.x], cs_lds[threadIdx.x]);
    }
}

}  // namespace

extern "C" {

int nb_composite_bwd(const float *raw, const float *z_vals, const float *ray_d, int64_t n_rays, int32_t n_samples,
                     int white_bkgd, const float *d_rgb_map, const float *d_acc_map, const float *d_depth_map,
                     float *d_raw, void *stream) {
    NB_REQUIRE(n_rays >= 0 && n_samples >= 1, "nb_composite_bwd: bad sizes");
    if (n_rays == 0) return NB_OK;
    NB_REQUIRE(raw && z_vals && ray_d && d_rgb_map && d_raw, "nb_composite_bwd: NULL pointer");
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(nb_ceil_div(n_rays, 64)), dim3(64), 0, (hipStream_t)stream, raw, z_vals,
                       ray_d, (long long)n_rays, n_samples, white_bkgd, d_rgb_map, d_acc_map, d_depth_map, d_raw);
    NB_CHECK_LAUNCH("composite_bwd_kernel");
    return NB_OK;
}

int nb_sgemm(int trans_a, int trans_b, int32_t m, int32_t n, int32_t k, float alpha, const float *a, int32_t lda,
             const float *b, int32_t ldb, float beta, float *c, int32_t ldc, void *stream) {
    NB_REQUIRE(c && (k == 0 || (a && b)) && m >= 0 && n >= 0 && k >= 0, "nb_sgemm: bad argument");
    if (m == 0 || n == 0) return NB_OK;
    return nb_gemm_fused(trans_a, trans_b, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, nullptr, 0, nullptr, stream);
}

// NB_BWD_SPLIT=0 in the environment (read once) keeps every GEMM on the exact-fp32 kernels: an exact training step for A/B
// runs and debugging; the default sends the large shapes to the 16-bit matrix pipe with bf16 pairs (~2^-16)
static bool bwd_split_enabled() {
    static const bool on = [] {
        const char *e = getenv("NB_BWD_SPLIT");
        return !(e && e[0] == '0');
    }();
    return on;
}

int nb_gemm_fused(int trans_a, int trans_b, int32_t m, int32_t n, int32_t k, float alpha, const float *a, int32_t lda,
                  const float *b, int32_t ldb, float beta, float *c, int32_t ldc, const float *mask_y, int32_t ldy,
                  float *colsum, void *stream) {
    NB_REQUIRE(c && (k == 0 || (a && b)) && m >= 0 && n >= 0 && k >= 0, "nb_gemm_fused: bad argument");
    if (m == 0 || n == 0) return NB_OK;
    hipStream_t st = (hipStream_t)stream;
    if (trans_a) {
        NB_REQUIRE(!trans_b && !mask_y && !colsum, "nb_gemm_fused: op(A) = A^T is the weight-gradient form (no epilogue, B not transposed)");
        if (beta != 1.f)
            hipLaunchKernelGGL(scale_matrix_kernel, dim3(nb_ceil_div((long long)m * n, 256)), dim3(256), 0, st, c, m, n, ldc, beta);
        if (k == 0) {
            NB_CHECK_LAUNCH("scale_matrix_kernel");
            return NB_OK;
        }
        // weight-gradient shapes (M, N >= 32, thousands of rows, 16-byte aligned rows): bf16 pairs on the 16-bit matrix pipe
        if (bwd_split_enabled() && m >= 32 && n >= 32 && k >= 1024 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0) {
            hipLaunchKernelGGL(gemm_tn16_kernel, dim3(nb_ceil_div(m, 128), nb_ceil_div(n, 128), nb_ceil_div(k, TN16_ROWS)), dim3(256), 0,
                               st, a, lda, b, ldb, (long long)k, m, n, alpha, c, ldc);
            NB_CHECK_LAUNCH("gemm_tn16_kernel");
            return NB_OK;
        }
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(nb_ceil_div(k, TN_WAVES * TN_ROWS), nb_ceil_div(m, 32), nb_ceil_div(n, 32)), dim3(64 * TN_WAVES),
                           0, st, a, lda, b, ldb, (long long)k, m, n, alpha, c, ldc);
        NB_CHECK_LAUNCH("gemm_tn_kernel");
        return NB_OK;
    }
    const dim3 grid(nb_ceil_div(m, 128), nb_ceil_div(n, 128)), block(256);
    // k == 0 (an empty product: C = beta C, then the epilogue) goes to the general kernel, whose loop is guarded; the fast
    // kernel pre-loads its first operand panels before looking at k
    const bool aligned = k > 0 && k % 16 == 0 && lda % 4 == 0 && ((uintptr_t)a % 16) == 0 && ldb % 4 == 0 && ((uintptr_t)b % 16) == 0;
    // the MLP backward's dX products (thousands of rows, K a multiple of 32, op(B) = B): bf16 pairs on the 16-bit matrix pipe
    if (bwd_split_enabled() && aligned && !trans_b && k % 32 == 0 && m >= 1024) {
        hipLaunchKernelGGL(gemm_rows16_kernel, grid, block, 0, st, a, lda, b, ldb, (long long)m, k, n, alpha, beta, c, ldc, mask_y, ldy,
                           colsum);
        NB_CHECK_LAUNCH("gemm_rows16_kernel");
        return NB_OK;
    }
    if (aligned && trans_b)
        hipLaunchKernelGGL((gemm_rows_fast_kernel<true>), grid, block, 0, st, a, lda, b, ldb, (long long)m, k, n, alpha, beta, c, ldc, mask_y, ldy, colsum);
    else if (aligned)
        hipLaunchKernelGGL((gemm_rows_fast_kernel<false>), grid, block, 0, st, a, lda, b, ldb, (long long)m, k, n, alpha, beta, c, ldc, mask_y, ldy, colsum);
    else
        hipLaunchKernelGGL(gemm_rows_kernel, grid, block, 0, st, a, lda, b, ldb, trans_b, (long long)m, k, n, alpha, beta, c, ldc,
                           mask_y, ldy, colsum);
    NB_CHECK_LAUNCH("gemm_rows_kernel");
    return NB_OK;
}

int nb_relu_bwd(float *dy, const float *y, int64_t n, void *stream) {
    NB_REQUIRE(n >= 0, "nb_relu_bwd: n < 0");
    if (n == 0) return NB_OK;
    NB_REQUIRE(dy && y, "nb_relu_bwd: NULL pointer");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(nb_ceil_div(nb_ceil_div(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, dy, y,
                       (long long)n);
    NB_CHECK_LAUNCH("relu_bwd_kernel");
    return NB_OK;
}

int nb_colsum(const float *x, int64_t n_rows, int32_t n_cols, int32_t ld, float *out, void *stream) {
    NB_REQUIRE(n_rows >= 0 && n_cols >= 0 && ld >= n_cols, "nb_colsum: bad sizes");
    if (n_rows == 0 || n_cols == 0) return NB_OK;
    NB_REQUIRE(x && out, "nb_colsum: NULL pointer");
    const int slabs = (int)(n_rows < 4096 ? 1 : (n_rows / 1024 < 256 ? n_rows / 1024 : 256));
    hipLaunchKernelGGL(colsum_kernel, dim3(nb_ceil_div(n_cols, 64), slabs), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)n_rows, n_cols, ld, out);
    NB_CHECK_LAUNCH("colsum_kernel");
    return NB_OK;
}

int nb_trilinear_bwd(const nb_scene *scene, const int32_t *const grids[4], float *const drows[4], const float *wpts,
                     const float *d_feat, int64_t n, int32_t run_length, void *stream) {
    NB_REQUIRE(scene && grids && drows && n >= 0 && run_length >= 1, "nb_trilinear_bwd: bad argument");
    if (n == 0) return NB_OK;
    NB_REQUIRE(wpts && d_feat, "nb_trilinear_bwd: NULL pointer");
    TriArgs a = {};
    nb_scene tmp = *scene;
    for (int l = 0; l < 4; ++l) {
        NB_REQUIRE(grids[l] && drows[l], "nb_trilinear_bwd: NULL grid / gradient buffer at level %d", l);
        a.grid[l] = grids[l];
        a.drows[l] = drows[l];
        if (!tmp.vol[l]) tmp.vol[l] = wpts;  // the forward volumes are not read here; any non-NULL pointer passes the check
    }
    if (int rc = fill_scene(&tmp, &a.sc)) return rc;
    a.wpts = wpts;
    a.dF = d_feat;
    a.n = n;
    a.run = run_length;
    hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(nb_ceil_div(nb_ceil_div(n, run_length) * 88, 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    NB_CHECK_LAUNCH("trilinear_bwd_kernel");
    return NB_OK;
}

}  // extern "C"
