// nb_march_msplit.hip — split-bf16 march kernel, "M-split" workgroup organisation (NB_PREC_BF16X3S) for gfx950.
//
// Same arithmetic as nb_march_bf16.hip (v_mfma_f32_32x32x16_bf16, W.X ~= W_hi.X_hi + W_hi.X_lo + W_lo.X_hi, fp32
// accumulate) but a different division of labour, chosen from measurements of that kernel:
//
//   * a workgroup (4 waves) marches 64 rays = two 32-sample N tiles; wave w owns a QUARTER OF EVERY LAYER'S OUTPUT
//     FEATURES (M split) for all 64 samples, so a weight fragment is used for 6 MFMAs instead of 3;
//   * activations live in LDS as ready-made B fragments (bf16 hi + lo, 64 KiB, rewritten in place after every
//     layer: two barriers per layer instead of one per 30 MFMAs); a wave reads 4 B fragments per 12 MFMAs —
//     3.5x fewer LDS reads per MFMA than reading the weights from an LDS ring;
//   * each wave streams ITS OWN quarter of the weights straight from L2 into an 8-fragment register ring
//     (plain global_load_dwordx4, counted vmcnt by the compiler): no LDS-DMA, no page barriers;
//   * <= 256 registers and ~70 KiB LDS: TWO workgroups per CU, so one workgroup's gather / conversion / barrier
//     phases run under the other's MFMAs.
//
// Per-sample work (ray set-up, trilinear gather, positional encoding, compositing) is done by "owner" lanes:
// wave w, lane l owns sample 16 w + (l & 15) and, of that sample, channel quarter / axis `part` = l >> 4.
#include "nb_march_common.h"

using namespace nbm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NB_MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

namespace {

// ---------------------------------------------------------------- weight stream (per wave)
// fragments (1 KiB = 64 lanes x 8 bf16) in consumption order: for every K chunk c, for every M tile m: A_hi, A_lo
#ifndef NB_MS_RING
#define NB_MS_RING 4
#endif
constexpr int S_R = NB_MS_RING;  // register ring depth (fragments): 8 spills more than it prefetches (61 vs 43 spilled registers)
constexpr int NC_P1 = 8, NC_P2 = 8, NC_P3 = 6, NC_H = 16, NC_V2 = 16, NC_V1 = 8;
constexpr int F_P1 = 0;                      // fc_0, K phase 1: pyramid level 3 (columns 224..351)
constexpr int F_P2 = F_P1 + 4 * NC_P1;       // fc_0, K phase 2: level 2 (96..223)
constexpr int F_P3 = F_P2 + 4 * NC_P2;       // fc_0, K phase 3: levels 0 and 1 (0..95)
constexpr int F_L1 = F_P3 + 4 * NC_P3;
constexpr int F_L2 = F_L1 + 4 * NC_H;
constexpr int F_L4 = F_L2 + 4 * NC_H;        // feature_fc . latent_fc[:, :256] merged
constexpr int F_V2 = F_L4 + 4 * NC_H;        // view_fc, K phase over the merged layer's 256 outputs (one M tile per wave)
constexpr int F_V1 = F_V2 + 2 * NC_V2;       // view_fc, K phase over the positional encodings (4 x 32 slots)
constexpr int F_TOTAL = F_V1 + 2 * NC_V1;    // 328
static_assert(F_TOTAL % S_R == 0, "the ring must divide the stream (static slot of every fragment, also across steps)");

// fp32 section of the packed blob (written by nb_pack_kernel, nb_march.hip): offsets in floats
constexpr int F_OFF_B0 = 8 * 44 * 256;
constexpr int F_OFF_B1 = F_OFF_B0 + 256 + 8 * 32 * 256;
constexpr int F_OFF_B2 = F_OFF_B1 + 256 + 8 * 32 * 256;
constexpr int F_OFF_AW = F_OFF_B2 + 256;
constexpr int F_OFF_AB = F_OFF_AW + 256;
constexpr int F_OFF_L4 = F_OFF_AB + 4;
constexpr int F_OFF_LV = F_OFF_L4 + 8 * 32 * 256;
constexpr int F_OFF_BV = F_OFF_LV + 4 * 44 * 256;
constexpr int F_OFF_RW = F_OFF_BV + 128;
constexpr int F_OFF_RB = F_OFF_RW + 384;
constexpr int F_PACK_SIZE = F_OFF_RB + 4;

// ---------------------------------------------------------------- LDS
constexpr int ACT_CHUNK_BYTES = 2 * 2048;            // one K=16 chunk: 2 N tiles x (B_hi 1 KiB + B_lo 1 KiB)
constexpr int ACT_BYTES = 16 * ACT_CHUNK_BYTES;      // 64 KiB
constexpr int TILE_OFF = 8 * ACT_CHUNK_BYTES;        // voxel tiles of the gather live in chunks 8..15 (free in fc_0's K phases)
constexpr int TILE_BYTES = 8192;                     // per wave
constexpr int SCR_A = ACT_BYTES;                     // alpha_fc partial sums [64 samples][4 waves] floats
constexpr int SCR_C = SCR_A + 1024;                  // rgb_fc partial sums [3][64][4] floats
#ifdef NB_ABL_ONEWG
constexpr int LDS_BYTES = SCR_C + 3072 + 16384;
#else
constexpr int LDS_BYTES = SCR_C + 3072;
#endif
#ifndef NB_ABL_ONEWG
static_assert(LDS_BYTES <= 81920, "two workgroups per CU");
#endif

__device__ __forceinline__ void split8s(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}

// B-fragment element (8 consecutive K values of one sample): K index k0 (multiple of 8) of the current phase
__device__ __forceinline__ void write_b8(char *act, int k0, int sample, const float (&v)[8]) {
    bf16x8 h, l;
    split8s(v, h, l);
    char *p = act + (k0 >> 4) * ACT_CHUNK_BYTES + (sample >> 5) * 2048 + ((((k0 >> 3) & 1) << 5) + (sample & 31)) * 16;
    *reinterpret_cast<bf16x8 *>(p) = h;
    *reinterpret_cast<bf16x8 *>(p + 1024) = l;
}

__device__ __forceinline__ f32x16 bias_tile_g(const float *bp, int t, int hi) {
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(bp + (t * 2 + hi) * 16);
    const f32x4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
    return f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
}

struct WRing {
    bf16x8 f[S_R];
};

__device__ __forceinline__ bf16x8 load_frag(const char *wl, int f) {
    return *reinterpret_cast<const bf16x8 *>(wl + (size_t)(f % F_TOTAL) * 1024);
}

// One K phase of one layer for this wave: MT M tiles x 2 N tiles, NC chunks read from LDS chunks 0..NC-1.
template <int F0, int MT, int NC>
__device__ __forceinline__ void layer_s(const char *wl, const char *act, int lane, WRing &ring, f32x16 (&acc)[2][2]) {
#ifdef NB_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    bf16x8 bh[2][2], bl[2][2];
    auto rd_b = [&](int c, int buf) {
        const char *p = act + c * ACT_CHUNK_BYTES + lane * 16;
        bh[buf][0] = *reinterpret_cast<const bf16x8 *>(p);
        bl[buf][0] = *reinterpret_cast<const bf16x8 *>(p + 1024);
        bh[buf][1] = *reinterpret_cast<const bf16x8 *>(p + 2048);
        bl[buf][1] = *reinterpret_cast<const bf16x8 *>(p + 3072);
    };
    rd_b(0, 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cur = c & 1;
        if (c + 1 < NC) rd_b(c + 1, cur ^ 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int f = F0 + 2 * (c * MT + m);
            const bf16x8 ah = ring.f[f % S_R], al = ring.f[(f + 1) % S_R];
            ring.f[f % S_R] = load_frag(wl, f + S_R);
            ring.f[(f + 1) % S_R] = load_frag(wl, f + 1 + S_R);
            acc[m][0] = NB_MFMA16S(ah, bh[cur][0], acc[m][0]);
            acc[m][1] = NB_MFMA16S(ah, bh[cur][1], acc[m][1]);
            acc[m][0] = NB_MFMA16S(ah, bl[cur][0], acc[m][0]);
            acc[m][1] = NB_MFMA16S(ah, bl[cur][1], acc[m][1]);
            acc[m][0] = NB_MFMA16S(al, bh[cur][0], acc[m][0]);
            acc[m][1] = NB_MFMA16S(al, bh[cur][1], acc[m][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef NB_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

template <int MT>
__device__ __forceinline__ void init_bias(const float *bp, int tile0, int hi, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f32x16 b = bias_tile_g(bp, tile0 + m, hi);
        acc[m][0] = b;
        acc[m][1] = b;
    }
}

// (optionally relu'd) accumulators of this wave -> B fragments of the next layer, in place.  Output feature
// 32 * (tile0 + m) + tile_row(r, hi): register group j = r >> 2 holds 4 consecutive features 8 j + 4 hi + (0..3),
// i.e. half of one 16-byte B element; the two half-wave lanes of a sample fill it together (ds_write_b64 each).
template <int MT, bool RELU>
__device__ __forceinline__ void publish_s(char *act, int lane, int tile0, f32x16 (&acc)[2][2]) {
    const int i = lane & 31, hi = lane >> 5;
    __syncthreads();  // every wave is done reading the previous activations
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 hh[4], ll[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[m][n][4 * j + e];
                    if (RELU) {
                        x = relu1(x);
                        acc[m][n][4 * j + e] = x;
                    }
                    hh[e] = (__bf16)x;
                    ll[e] = (__bf16)(x - (float)hh[e]);
                }
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const int chunk = 2 * (tile0 + m) + (j >> 1);
                char *p = act + chunk * ACT_CHUNK_BYTES + n * 2048 + ((j & 1) * 32 + i) * 16 + 8 * hi;
                *reinterpret_cast<u32x2 *>(p) = __builtin_bit_cast(u32x2, hh);
                *reinterpret_cast<u32x2 *>(p + 1024) = __builtin_bit_cast(u32x2, ll);
            }
    __syncthreads();
}

// ---------------------------------------------------------------- gather, owner-lane layout (sample = lane & 15, part = lane >> 4)
__device__ __forceinline__ float red_min16(float v) {
    v = fminf(v, swz_xor<8>(v));
    v = fminf(v, swz_xor<4>(v));
    v = fminf(v, swz_xor<2>(v));
    v = fminf(v, swz_xor<1>(v));
    return v;
}
__device__ __forceinline__ float red_max16(float v) {
    v = fmaxf(v, swz_xor<8>(v));
    v = fmaxf(v, swz_xor<4>(v));
    v = fmaxf(v, swz_xor<2>(v));
    v = fmaxf(v, swz_xor<1>(v));
    return v;
}

// channels [part * C/4, (part+1) * C/4) of level L for this lane's sample; same corner order, weights and zero
// padding as gather_level / gather_level_coop (nb_march_common.h), tile = wave-private LDS
template <int L, typename Sink>
__device__ __forceinline__ void gather_parts(const SceneDev &sc, const GridCoord &g, const WaveBox &wb, int part, int lane,
                                             char *buf, Sink sink) {
    constexpr int C = lvl_c(L), QC = C / 4, PC = C / 4;  // PC 16-byte pieces per voxel
    constexpr int MAX_IT = TILE_BYTES / 1024;
    const int D = sc.dhw[L][0], H = sc.dhw[L][1], W = sc.dhw[L][2];
    const float ix = unnorm_clamped(g.gw, W), iy = unnorm_clamped(g.gh, H), iz = unnorm_clamped(g.gd, D);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx[2] = {(fx + 1.f) - ix, ix - fx};
    const float wy[2] = {(fy + 1.f) - iy, iy - fy};
    const float wz[2] = {(fz + 1.f) - iz, iz - fz};
    const int xlo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.lo.gw, W)), 0), W - 1));
    const int ylo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.lo.gh, H)), 0), H - 1));
    const int zlo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.lo.gd, D)), 0), D - 1));
    const int xhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.hi.gw, W)) + 1, 0), W - 1));
    const int yhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.hi.gh, H)) + 1, 0), H - 1));
    const int zhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.hi.gd, D)) + 1, 0), D - 1));
    const int nx = xhi - xlo + 1, ny = yhi - ylo + 1, nz = zhi - zlo + 1;
    const int pieces = nx * ny * nz * PC;
    float cw[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
        const bool inb = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D;
        cw[corner] = inb ? (wx[dx] * wy[dy]) * wz[dz] : 0.f;
    }
    if (pieces <= TILE_BYTES / 16) {  // wave-uniform: the wave's voxel box fits its tile
        const float rcp_xy = 1.f / (float)(nx * ny), rcp_x = 1.f / (float)nx;
#pragma unroll
        for (int b = 0; b < MAX_IT; b += 4) {  // four coalesced 1-KiB fetches in flight at a time (register budget)
            if (b * 64 < pieces) {
                f32x4 t[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if ((b + it) * 64 < pieces) {
                        const int p = min((b + it) * 64 + lane, pieces - 1);
                        const int v = p / PC, q = p % PC;
                        const int vz = (int)(((float)v + 0.5f) * rcp_xy);
                        const int r = v - vz * nx * ny;
                        const int vy = (int)(((float)r + 0.5f) * rcp_x);
                        const int vx = r - vy * nx;
                        const size_t lin = ((size_t)((zlo + vz) * H + (ylo + vy))) * W + (xlo + vx);
                        t[it] = *reinterpret_cast<const f32x4 *>(sc.vol[L] + lin * C + q * 4);
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if ((b + it) * 64 < pieces) {
                        const int p = (b + it) * 64 + lane;
                        if (p < pieces) *reinterpret_cast<f32x4 *>(buf + p * 16) = t[it];
                    }
                }
            }
        }
        int co[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int xx = x0 + (corner & 1), yy = y0 + ((corner >> 1) & 1), zz = z0 + (corner >> 2);
            const int xc = min(max(xx, xlo), xhi), yc = min(max(yy, ylo), yhi), zc = min(max(zz, zlo), zhi);
            co[corner] = ((((zc - zlo) * ny + (yc - ylo)) * nx + (xc - xlo)) * C + part * QC) * 4;
        }
#pragma unroll
        for (int grp = 0; grp < QC / 8; ++grp) {  // 8 channels at a time: blend, hand over, forget
            float o8[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int corner = 0; corner < 8; ++corner) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(buf + co[corner] + (2 * grp + h) * 16);
                    a.x = fmaf(cw[corner], v.x, a.x);
                    a.y = fmaf(cw[corner], v.y, a.y);
                    a.z = fmaf(cw[corner], v.z, a.z);
                    a.w = fmaf(cw[corner], v.w, a.w);
                }
                o8[4 * h + 0] = a.x;
                o8[4 * h + 1] = a.y;
                o8[4 * h + 2] = a.z;
                o8[4 * h + 3] = a.w;
            }
            sink(grp, o8);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {  // rays far apart (random training rays): read the corners from global memory
        const float *cpb[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int xx = x0 + (corner & 1), yy = y0 + ((corner >> 1) & 1), zz = z0 + (corner >> 2);
            const int xc = min(max(xx, 0), W - 1), yc = min(max(yy, 0), H - 1), zc = min(max(zz, 0), D - 1);
            cpb[corner] = sc.vol[L] + ((size_t)(zc * H + yc) * W + xc) * C + part * QC;
        }
#pragma unroll
        for (int grp = 0; grp < QC / 8; ++grp) {
            float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(cpb[corner] + 8 * grp);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(cpb[corner] + 8 * grp + 4);
                o8[0] = fmaf(cw[corner], v0.x, o8[0]);
                o8[1] = fmaf(cw[corner], v0.y, o8[1]);
                o8[2] = fmaf(cw[corner], v0.z, o8[2]);
                o8[3] = fmaf(cw[corner], v0.w, o8[3]);
                o8[4] = fmaf(cw[corner], v1.x, o8[4]);
                o8[5] = fmaf(cw[corner], v1.y, o8[5]);
                o8[6] = fmaf(cw[corner], v1.z, o8[6]);
                o8[7] = fmaf(cw[corner], v1.w, o8[7]);
            }
            sink(grp, o8);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// level L -> B fragments; kbase = first K index of the level inside the current K phase
template <int L>
__device__ __forceinline__ void gather_publish(const SceneDev &sc, const GridCoord &g, const WaveBox &wb, int part, int lane,
                                               char *tile, char *act, int kbase, int sample) {
    constexpr int QC = lvl_c(L) / 4;
#ifdef NB_ABL_NOGATHER
#pragma unroll
    for (int grp = 0; grp < QC / 8; ++grp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g.gw * (float)(e + 1) + g.gh;
        write_b8(act, kbase + part * QC + 8 * grp, sample, v);
    }
#else
    gather_parts<L>(sc, g, wb, part, lane, tile,
                    [&](int grp, const float (&v)[8]) { write_b8(act, kbase + part * QC + 8 * grp, sample, v); });
#endif
}

// compositing state of the weights output: 16 consecutive depth steps of a ray = 64 bytes, 4 steps per owner lane
struct WeightStore4 {
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    __device__ __forceinline__ void push(const MarchArgs &a, long long ray, int s, int S, int part, bool valid, float w) {
        if ((S & 15) != 0) {
            if (valid && part == 0) a.weights[ray * S + s] = w;
            return;
        }
        const int slot = s & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (slot == part * 4 + i) q[i] = w;
        if (slot == 15 && valid)
            *reinterpret_cast<f32x4 *>(a.weights + ray * S + (s - 15) + part * 4) = f32x4{q[0], q[1], q[2], q[3]};
    }
};

// ---------------------------------------------------------------- the kernel
__global__ __launch_bounds__(256, 2) void nb_march16s_kernel(MarchArgs a, const char *stream) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    char *act = lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int os = lane & 15, part = lane >> 4;  // owner role
    const int sample = 16 * wave + os;            // 0..63 inside the workgroup
    const int grp = xcd_remap(blockIdx.x, a.n_wave_groups);
    long long ray = (long long)grp * 64 + sample;
    const bool valid = ray < a.n_rays;
    if (!valid) ray = a.n_rays - 1;
    if (a.ray_order) ray = a.ray_order[ray];
    const int S = a.n_samples;
    const float ox = a.ray_o[ray * 3 + 0], oy = a.ray_o[ray * 3 + 1], oz = a.ray_o[ray * 3 + 2];
    const float dx = a.ray_d[ray * 3 + 0], dy = a.ray_d[ray * 3 + 1], dz = a.ray_d[ray * 3 + 2];
    const float near = a.near[ray], far = a.far[ray];
    const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    const float vx = dx / dn, vy = dy / dn, vz = dz / dn;
    // positional encoding of the view direction for this lane's axis (constant along the ray): v, sin/cos(v 2^k), k < 4
    float vpe[9];
    {
        const float va = part == 0 ? vx : (part == 1 ? vy : vz);
        const double t = (double)va * NB_INV_2PI;
        vpe[0] = va;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vpe[1 + 2 * k] = sin_rev(t * (double)(1 << k));
            vpe[2 + 2 * k] = sin_rev(t * (double)(1 << k) + 0.25);
        }
    }
    const float *tr = a.t_rand ? a.t_rand + ray * S : nullptr;
    auto z_at = [&](int s) -> float {
        const float zc = z_lin(near, far, a.t_vals[s]);
        if (!tr) return zc;
        const float lower = s == 0 ? zc : 0.5f * __fadd_rn(zc, z_lin(near, far, a.t_vals[s - 1]));
        const float upper = s == S - 1 ? zc : 0.5f * __fadd_rn(z_lin(near, far, a.t_vals[s + 1]), zc);
        return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), tr[s]));
    };

    const char *wl0 = stream + (size_t)wave * F_TOTAL * 1024 + lane * 16;
    WRing ring;
#pragma unroll
    for (int i = 0; i < S_R; ++i) ring.f[i] = load_frag(wl0, i);

    RayAccum ra;
    WeightStore4 wstore;
    float z_cur = z_at(0);
    for (int s = 0; s < S; ++s) {
        // loop-invariant address roots are laundered so that LICM does not hoist (and spill) hundreds of addresses
        int zero = 0, lane_i = lane;
        asm volatile("" : "+s"(zero), "+v"(lane_i));
        const char *wl = wl0 + zero;
        char *actz = act + zero;
        char *tile = actz + TILE_OFF + wave * TILE_BYTES;
        const float *pk = a.pk + zero;
        const float z_next = (s + 1 < S) ? z_at(s + 1) : 0.f;
        const float px = __fadd_rn(ox, __fmul_rn(dx, z_cur));
        const float py = __fadd_rn(oy, __fmul_rn(dy, z_cur));
        const float pz = __fadd_rn(oz, __fmul_rn(dz, z_cur));
        f32x16 acc[2][2];

        // ---- fc_0 in three K phases (the voxel tiles use the upper half of the activation buffer)
        {
            const GridCoord g = grid_coords(a.sc, px, py, pz);
            WaveBox wb;
            wb.lo.gw = red_min16(g.gw);
            wb.lo.gh = red_min16(g.gh);
            wb.lo.gd = red_min16(g.gd);
            wb.hi.gw = red_max16(g.gw);
            wb.hi.gh = red_max16(g.gh);
            wb.hi.gd = red_max16(g.gd);
            gather_publish<3>(a.sc, g, wb, part, lane_i, tile, actz, 0, sample);
            init_bias<2>(pk + F_OFF_B0, 2 * wave, hi, acc);
            __syncthreads();
            layer_s<F_P1, 2, NC_P1>(wl, actz, lane_i, ring, acc);
            __syncthreads();
            gather_publish<2>(a.sc, g, wb, part, lane_i, tile, actz, 0, sample);
            __syncthreads();
            layer_s<F_P2, 2, NC_P2>(wl, actz, lane_i, ring, acc);
            __syncthreads();
            gather_publish<0>(a.sc, g, wb, part, lane_i, tile, actz, 0, sample);
            gather_publish<1>(a.sc, g, wb, part, lane_i, tile, actz, 32, sample);
            __syncthreads();
            layer_s<F_P3, 2, NC_P3>(wl, actz, lane_i, ring, acc);
        }
        publish_s<2, true>(actz, lane_i, 2 * wave, acc);
        // ---- fc_1, fc_2
        init_bias<2>(pk + F_OFF_B1, 2 * wave, hi, acc);
        layer_s<F_L1, 2, NC_H>(wl, actz, lane_i, ring, acc);
        publish_s<2, true>(actz, lane_i, 2 * wave, acc);
        init_bias<2>(pk + F_OFF_B2, 2 * wave, hi, acc);
        layer_s<F_L2, 2, NC_H>(wl, actz, lane_i, ring, acc);
        publish_s<2, true>(actz, lane_i, 2 * wave, acc);  // acc now holds relu(h3)
        // ---- alpha_fc: partial dot product over this wave's 64 features, finished by the owner lanes
        {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float sa = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x4 *aw = reinterpret_cast<const f32x4 *>(pk + F_OFF_AW + hi * 128 + 16 * (2 * wave + m));
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 w4 = aw[q4];
                        sa = fmaf(w4.x, acc[m][n][4 * q4 + 0], sa);
                        sa = fmaf(w4.y, acc[m][n][4 * q4 + 1], sa);
                        sa = fmaf(w4.z, acc[m][n][4 * q4 + 2], sa);
                        sa = fmaf(w4.w, acc[m][n][4 * q4 + 3], sa);
                    }
                }
                sa = add_halves(sa);
                if (hi == 0) reinterpret_cast<float *>(actz + SCR_A)[(n * 32 + (lane_i & 31)) * 4 + wave] = sa;
            }
        }
        // ---- merged feature_fc / latent_fc layer (per-frame latent code folded into the bias)
        init_bias<2>(a.lb + zero, 2 * wave, hi, acc);
        layer_s<F_L4, 2, NC_H>(wl, actz, lane_i, ring, acc);
        publish_s<2, false>(actz, lane_i, 2 * wave, acc);
        // ---- view_fc: one M tile per wave; K phase over the 256 latent-layer outputs, then over the encodings
        init_bias<1>(pk + F_OFF_BV, wave, hi, acc);
        layer_s<F_V2, 1, NC_V2>(wl, actz, lane_i, ring, acc);
        __syncthreads();
        {
            // encodings: lane (sample, axis a = part < 3) writes 32 K slots
            //   [x, (sin, cos)(x 2^k) k<10, v, (sin, cos)(v 2^k) k<4, 0, 0]; part 3 writes zeros
            const float xa = part == 0 ? px : (part == 1 ? py : pz);
            const double t = (double)xa * NB_INV_2PI;
            const float keep = part < 3 ? 1.f : 0.f;
            float e[32];
            e[0] = xa;
#ifdef NB_ABL_NOPE
#pragma unroll
            for (int k = 0; k < 20; ++k) e[1 + k] = xa * (float)k;
#else
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                e[1 + 2 * k] = sin_rev(t * (double)(1 << k));
                e[2 + 2 * k] = sin_rev(t * (double)(1 << k) + 0.25);
            }
#endif
#pragma unroll
            for (int k = 0; k < 9; ++k) e[21 + k] = vpe[k];
            e[30] = 0.f;
            e[31] = 0.f;
#pragma unroll
            for (int grp8 = 0; grp8 < 4; ++grp8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = e[8 * grp8 + q] * keep;
                write_b8(actz, 32 * part + 8 * grp8, sample, v);
            }
        }
        __syncthreads();
        layer_s<F_V1, 1, NC_V1>(wl, actz, lane_i, ring, acc);
        // ---- rgb_fc partial sums over this wave's 32 view features
        {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const f32x4 *rw = reinterpret_cast<const f32x4 *>(pk + F_OFF_RW + (ch * 2 + hi) * 64 + 16 * wave);
                    float sc = 0.f;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 w4 = rw[q4];
                        sc = fmaf(w4.x, relu1(acc[0][n][4 * q4 + 0]), sc);
                        sc = fmaf(w4.y, relu1(acc[0][n][4 * q4 + 1]), sc);
                        sc = fmaf(w4.z, relu1(acc[0][n][4 * q4 + 2]), sc);
                        sc = fmaf(w4.w, relu1(acc[0][n][4 * q4 + 3]), sc);
                    }
                    sc = add_halves(sc);
                    if (hi == 0) reinterpret_cast<float *>(actz + SCR_C)[(ch * 64 + n * 32 + (lane_i & 31)) * 4 + wave] = sc;
                }
        }
        __syncthreads();
        // ---- owner lanes: finish the heads, composite
        float out[4];
        {
            const f32x4 pa = *reinterpret_cast<const f32x4 *>(actz + SCR_A + sample * 16);
            out[3] = ((pa.x + pa.y) + (pa.z + pa.w)) + pk[F_OFF_AB];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const f32x4 pc = *reinterpret_cast<const f32x4 *>(actz + SCR_C + (ch * 64 + sample) * 16);
                out[ch] = ((pc.x + pc.y) + (pc.z + pc.w)) + pk[F_OFF_RB + ch];
            }
        }
        float dist = (s + 1 < S) ? __fsub_rn(z_next, z_cur) : 1e10f;
        dist = __fmul_rn(dist, dn);
        const float w = ra.add(out, z_cur, dist);
        wstore.push(a, ray, s, S, part, valid, w);
        if (valid && part == 0 && a.raw)
            *reinterpret_cast<f32x4 *>(a.raw + (ray * S + s) * 4) = f32x4{out[0], out[1], out[2], out[3]};
        z_cur = z_next;
    }
    if (valid && part == 0) ra.store(a, ray);
}

// ---------------------------------------------------------------- weight stream packing
__device__ __forceinline__ unsigned short bf16_bits_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// view_fc column of encoding slot `slot` (0..31) of axis a (see the kernel): -1 = zero pad
__device__ __forceinline__ int pe_slot_col(int a, int slot) {
    if (a >= 3 || slot >= 30) return -1;
    if (slot == 0) return 256 + 27 + a;
    if (slot <= 20) {
        const int k = (slot - 1) >> 1, is_cos = (slot - 1) & 1;
        return 256 + 27 + 3 + 6 * k + 3 * is_cos + a;
    }
    if (slot == 21) return 256 + a;
    const int k = (slot - 22) >> 1, is_cos = (slot - 22) & 1;
    return 256 + 3 + 6 * k + 3 * is_cos + a;
}

__global__ void nb_pack16s_kernel(nb_mlp_params p, const float *__restrict__ f32_blob, unsigned short *__restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)4 * F_TOTAL * 512) return;
    const int r8 = (int)(e & 7), lane = (int)((e >> 3) & 63), fw = (int)(e >> 9);
    const int w = fw / F_TOTAL, f = fw % F_TOTAL;
    const int i = lane & 31, kg = lane >> 5;
    int f0, mt, layer;
    if (f < F_P2) { f0 = F_P1; mt = 2; layer = 0; }
    else if (f < F_P3) { f0 = F_P2; mt = 2; layer = 1; }
    else if (f < F_L1) { f0 = F_P3; mt = 2; layer = 2; }
    else if (f < F_L2) { f0 = F_L1; mt = 2; layer = 3; }
    else if (f < F_L4) { f0 = F_L2; mt = 2; layer = 4; }
    else if (f < F_V2) { f0 = F_L4; mt = 2; layer = 5; }
    else if (f < F_V1) { f0 = F_V2; mt = 1; layer = 6; }
    else { f0 = F_V1; mt = 1; layer = 7; }
    const int rel = f - f0, is_lo = rel & 1, cm = rel >> 1, m = cm % mt, c = cm / mt;
    const int k = 16 * c + 8 * kg + r8;
    const int row = (mt == 2) ? 64 * w + 32 * m + i : 32 * w + i;
    float v = 0.f;
    if (layer == 0) v = p.fc0_w[row * 352 + 224 + k];
    else if (layer == 1) v = p.fc0_w[row * 352 + 96 + k];
    else if (layer == 2) v = p.fc0_w[row * 352 + k];
    else if (layer == 3) v = p.fc1_w[row * 256 + k];
    else if (layer == 4) v = p.fc2_w[row * 256 + k];
    else if (layer == 5) {
        // merged latent_fc[:, :256] @ feature_fc from the fp32 section (computed in fp64 there), whose layout is
        // ((t*32 + g)*64 + lane')*4 + e with q = 4g+e <-> column col_hidden(q, hi'): invert col_hidden for column k
        const int tt = k >> 5, rr = k & 31, hi2 = (rr >> 2) & 1, r2 = (rr & 3) + 4 * (rr >> 3), q2 = 16 * tt + r2;
        v = f32_blob[F_OFF_L4 + (((row >> 5) * 32 + (q2 >> 2)) * 64 + (hi2 * 32 + (row & 31))) * 4 + (q2 & 3)];
    } else if (layer == 6) v = p.view_w[row * 346 + k];
    else {
        const int col = pe_slot_col(k >> 5, k & 31);
        v = col < 0 ? 0.f : p.view_w[row * 346 + col];
    }
    const unsigned short h = bf16_bits_rne(v);
    const unsigned short l = bf16_bits_rne(v - __uint_as_float((unsigned)h << 16));
    out[((size_t)fw * 64 + lane) * 8 + r8] = is_lo ? l : h;
}

}  // namespace

namespace nbm {

long long msplit_stream_floats() { return (long long)4 * F_TOTAL * 1024 / 4; }

// `packed` = [fp32 section][bf16 ring stream][M-split stream]; `stream_off` = float offset of the last section
int pack_msplit_stream(const nb_mlp_params *p, float *packed, long long stream_off, hipStream_t st) {
    const long long n = (long long)4 * F_TOTAL * 512;
    hipLaunchKernelGGL(nb_pack16s_kernel, dim3(nb_ceil_div(n, 256)), dim3(256), 0, st, *p, packed,
                       reinterpret_cast<unsigned short *>(packed + stream_off));
    NB_CHECK_LAUNCH("nb_pack16s_kernel");
    return NB_OK;
}

int launch_march_msplit(MarchArgs a, long long stream_off, hipStream_t st) {
    a.n_wave_groups = (int)nb_ceil_div(a.n_rays, 64);
    hipLaunchKernelGGL(nb_march16s_kernel, dim3(a.n_wave_groups), dim3(256), 0, st, a,
                       reinterpret_cast<const char *>(a.pk + stream_off));
    NB_CHECK_LAUNCH("nb_march16s_kernel");
    return NB_OK;
}

}  // namespace nbm
