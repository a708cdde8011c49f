// nb_march_common.h — device code shared by the exact-fp32 (nb_march.hip) and the fc_0-folded f16f6 (nb_march_fold.hip)
// decode / march kernels: scene + argument structs, feature-vector layout, the fp32 kernel's trilinear gather,
// positional encoding, sampling helpers, compositing, XCD-aware block remap.
#pragma once
#include "nb_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace nbm {

constexpr int N_PE = 45;

// activation tap of nb_decode_points (floats per point): what the backward pass needs
//   [ F 352 | h1 256 | h2 256 | h3 256 | G 256 (latent_fc output) | V 128 (view_fc, post relu) | PE 90 | pad ]
constexpr int TAP_F = 0, TAP_H1 = 352, TAP_H2 = 608, TAP_H3 = 864, TAP_G = 1120, TAP_V = 1376, TAP_PE = 1504, TAP_WIDTH = 1600;

// relu as ONE v_max_f32: written as fmaxf (or as fmed3(x, 0, inf), which hipcc folds back into a max) the compiler first
// canonicalises the MFMA result with a v_max_f32 x, x, x of its own: two instructions per value in every publish
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// feature row (within a 32-row tile) held by accumulator register r of a lane with half index hi
__host__ __device__ constexpr int tile_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// channel layout of the gathered feature vector (latent_xyzc.py:63-71)
__host__ __device__ constexpr int lvl_c(int l) { return l == 0 ? 32 : (l == 1 ? 64 : 128); }
__host__ __device__ constexpr int lvl_chan_base(int l) { return l == 0 ? 0 : (l == 1 ? 32 : (l == 2 ? 96 : 224)); }
__host__ __device__ constexpr int lvl_reg_base(int l) { return lvl_chan_base(l) / 2; }

// fc_0 input column held in gather register q (0..175) of a lane with half index hi
__host__ __device__ inline int col_feat(int q, int hi) {
    int l = q < 16 ? 0 : (q < 48 ? 1 : (q < 112 ? 2 : 3));
    return lvl_chan_base(l) + hi * (lvl_c(l) / 2) + (q - lvl_reg_base(l));
}
// input column for accumulator element q = 16 * tile + r of the previous layer
__host__ __device__ inline int col_hidden(int q, int hi) { return 32 * (q >> 4) + tile_row(q & 15, hi); }
// view_fc column (346 = 256 + 27 + 63, latent_xyzc.py:113-118) for PE slot c, -1 = zero pad
__host__ __device__ inline int col_pe(int c, int hi) {
    if (c < 12) {  // view direction, frequency k = c/3, axis a = c%3: hi=0 sin, hi=1 cos
        int k = c / 3, a = c % 3;
        return 256 + 3 + 6 * k + 3 * hi + a;
    }
    if (c < 42) {  // world xyz
        int k = (c - 12) / 3, a = (c - 12) % 3;
        return 256 + 27 + 3 + 6 * k + 3 * hi + a;
    }
    if (c < 45) {  // raw inputs: hi=0 viewdir_a, hi=1 xyz_a
        int a = c - 42;
        return hi ? (256 + 27 + a) : (256 + a);
    }
    return -1;
}

// per-frame pose block in DEVICE memory (nb_scene.pose): R[9] row-major | Th[3] | bounds_min[3] (xyz).  The kernels read
// it through the constant address space (wave-uniform scalar loads), so sp_input's R / Th / bounds never visit the host.
typedef const float __attribute__((address_space(4))) *cfloat_ptr;
struct Pose {
    float R[9], Th[3], bmin[3];
};
__device__ __forceinline__ Pose load_pose(const float *pose) {
    cfloat_ptr p = (cfloat_ptr)pose;
    Pose q;
#pragma unroll
    for (int k = 0; k < 9; ++k) q.R[k] = p[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        q.Th[k] = p[9 + k];
        q.bmin[k] = p[12 + k];
    }
    return q;
}

struct SceneDev {
    const float *vol[4];
    int dhw[4][3];
    const float *pose;
    float vs[3];
    float osh[3];
    float fm1[4][3], fp1[4][3];  // (float)(size - 1), (float)size + 1 of every level: scalar operands (a kernel that converts
                                 // them itself keeps the results in VGPRs across its depth loop)
};

// fc_0 folded into the volumes (nb_fold, NB_PREC_F16F6): rows of 256 fp16 heads + 256 fp16 remainders, index grids
struct FoldDev {
    const char *urows;
    const int *grid[4];
    int row_base[4];
    unsigned zero_off;  // byte offset of the all-zero row
};

// sample culling against training-view silhouettes (nb_cull): the reference's fp32 operation order
struct CullDev {
    int n_views, H, W, pre;
    const unsigned char *msk;  // [n_views, H, W]
    const float *cam;          // [n_views, 21]: RT (3x4 row-major) | K (3x3)
    const float *snap;         // R0 (9) | Th0 (3), pre != 0 only
};

__device__ __forceinline__ int cull_pixel(float f, int size) {
    // torch: .round() (half to even) -> .long() -> clamp(0, size-1); non-finite / out-of-int64-range converts to
    // INT64_MIN on the CPU and therefore clamps to 0
    const float r = rintf(f);
    if (!(fabsf(r) < 9.0e18f)) return 0;
    return (int)fminf(fmaxf(r, 0.f), (float)(size - 1));
}

__device__ __forceinline__ bool cull_inside(const CullDev &c, const SceneDev &sc, float px, float py, float pz) {
    float p[3] = {px, py, pz};
    if (c.pre) {  // if_clight_renderer_msk.py:18-32
        const Pose ps = load_pose(sc.pose);
        cfloat_ptr sn = (cfloat_ptr)c.snap;
        const float q[3] = {__fsub_rn(px, ps.Th[0]), __fsub_rn(py, ps.Th[1]), __fsub_rn(pz, ps.Th[2])};
        float can[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            can[j] = __fadd_rn(__fadd_rn(__fmul_rn(q[0], ps.R[j]), __fmul_rn(q[1], ps.R[3 + j])), __fmul_rn(q[2], ps.R[6 + j]));
#pragma unroll
        for (int i = 0; i < 3; ++i)
            p[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(can[0], sn[i * 3]), __fmul_rn(can[1], sn[i * 3 + 1])),
                                       __fmul_rn(can[2], sn[i * 3 + 2])), sn[9 + i]);
    }
    bool inside = true;
    for (int v = 0; v < c.n_views; ++v) {  // if_clight_renderer_mmsk.py:21-38 (loops over batch['Ks'].size(1) views)
        cfloat_ptr RT = (cfloat_ptr)(c.cam + v * 21), K = RT + 12;
        float t[3], q[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            t[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(p[0], RT[i * 4]), __fmul_rn(p[1], RT[i * 4 + 1])),
                                       __fmul_rn(p[2], RT[i * 4 + 2])), RT[i * 4 + 3]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            q[i] = __fadd_rn(__fadd_rn(__fmul_rn(t[0], K[i * 3]), __fmul_rn(t[1], K[i * 3 + 1])),
                             __fmul_rn(t[2], K[i * 3 + 2]));
        const int x = cull_pixel(__fdiv_rn(q[0], q[2]), c.W), y = cull_pixel(__fdiv_rn(q[1], q[2]), c.H);
        inside = inside && c.msk[((size_t)v * c.H + y) * c.W + x] != 0;
    }
    return inside;
}

struct MarchArgs {
    SceneDev sc;
    CullDev cull;
    FoldDev fold;
    const float *pk;  // packed decoder weights (format depends on the kernel family)
    const float *lb;
    // ray mode
    const float *ray_o, *ray_d, *near, *far, *t_vals, *t_rand;
    const int *ray_order;  // optional slot -> ray list (nb_hip.h: >= 0 ray, -(r + 1) padding slot, NB_SLOT_DEAD empty)
    long long n_slots;     // its length (ray_order != NULL), a multiple of 64
    float *rgb_map, *disp_map, *acc_map, *weights, *depth_map, *raw;
    long long n_rays;
    int n_samples;
    int white_bkgd;
    // point mode
    const float *wpts, *viewdir;
    float *raw_out, *dbg;
    long long n_pts;
    int n_wave_groups;
    // last-sample fix-up of NB_PREC_F16F6 (nb_hip.h, nb_march `ill_scratch`): header + records, NULL = none
    float *ill;
    int ill_cap;  // records the scratch holds
};

// ---------------------------------------------------------------- positional encoding
// embedder.py:26-36: [x, sin(x*2^k), cos(x*2^k)]_k ; x*2^k is exact in fp32.
// Range reduction exploits that exactness: t = x / (2 pi) is formed ONCE per coordinate in fp64
// (error 2^-53 relative), t * 2^k is again exact, and the fractional revolution f = u - rint(u) in
// [-0.5, 0.5] goes to the hardware v_sin_f32, which takes its argument in revolutions (measured on
// MI355X: 1.25e-7 max abs error on that interval, better than sinf(2 pi f) in fp32).  cos is
// sin(u + 1/4), added before the reduction, so each lane needs one transcendental per entry instead
// of a ~115-instruction sincosf.  Total error vs torch.sin/cos of the fp32 argument <= ~4e-7 absolute.
#define NB_INV_2PI 0.15915494309189533576888376337251436
__device__ __forceinline__ float sin_rev(double u) {
    const double f = u - rint(u);
    return __builtin_amdgcn_sinf((float)f);
}
__device__ __forceinline__ void pe_view(float (&pe)[N_PE], float vx, float vy, float vz, int hi) {
    const float v[3] = {vx, vy, vz};
    const double q = hi ? 0.25 : 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double t = (double)v[a] * NB_INV_2PI;
#pragma unroll
        for (int k = 0; k < 4; ++k) pe[3 * k + a] = sin_rev(t * (double)(1 << k) + q);  // hi=0 sin, hi=1 cos
    }
}
__device__ __forceinline__ void pe_xyz(float (&pe)[N_PE], float px, float py, float pz, float vx, float vy,
                                       float vz, int hi) {
    const float p[3] = {px, py, pz};
    const float v[3] = {vx, vy, vz};
    const double q = hi ? 0.25 : 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double t = (double)p[a] * NB_INV_2PI;
#pragma unroll
        for (int k = 0; k < 10; ++k) pe[12 + 3 * k + a] = sin_rev(t * (double)(1 << k) + q);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) pe[42 + a] = hi ? p[a] : v[a];
}

// tap: positional encodings in view_fc input order (column 256 + e -> tap[e], e in [0, 90))
__device__ __forceinline__ void dump_pe(const float (&pe)[N_PE], float *dst, int hi) {
#pragma unroll
    for (int c = 0; c < N_PE; ++c) dst[col_pe(c, hi) - 256] = pe[c];
}

// ---------------------------------------------------------------- trilinear gather (K3 + K4)
// latent_xyzc.py:41-60 then grid_sample(align_corners=True, zeros) per level (:62-72).
// Lane (j, hi) accumulates channels [hi*C/2, (hi+1)*C/2) of every level for sample j.
struct GridCoord {
    float gw, gh, gd;  // normalised [-1,1] coordinates in grid_sample's x (W), y (H), z (D) order
};

__device__ __forceinline__ GridCoord grid_coords(const SceneDev &sc, float px, float py, float pz) {
    // (p - Th) @ R
    const Pose ps = load_pose(sc.pose);
    const float qx = px - ps.Th[0], qy = py - ps.Th[1], qz = pz - ps.Th[2];
    const float cx = fmaf(qz, ps.R[6], fmaf(qy, ps.R[3], qx * ps.R[0]));
    const float cy = fmaf(qz, ps.R[7], fmaf(qy, ps.R[4], qx * ps.R[1]));
    const float cz = fmaf(qz, ps.R[8], fmaf(qy, ps.R[5], qx * ps.R[2]));
    // dhw = (xyz[[2,1,0]] - min_dhw) / voxel_size / out_sh * 2 - 1 ; back to xyz order for grid_sample
    GridCoord g;
    g.gd = __fsub_rn(__fmul_rn(__fdiv_rn(__fdiv_rn(cz - ps.bmin[2], sc.vs[0]), sc.osh[0]), 2.f), 1.f);
    g.gh = __fsub_rn(__fmul_rn(__fdiv_rn(__fdiv_rn(cy - ps.bmin[1], sc.vs[1]), sc.osh[1]), 2.f), 1.f);
    g.gw = __fsub_rn(__fmul_rn(__fdiv_rn(__fdiv_rn(cx - ps.bmin[0], sc.vs[2]), sc.osh[2]), 2.f), 1.f);
    return g;
}

// One pyramid level.  The 8 corners x 16 channels of a group are fetched as 32 independent 16-byte
// loads BEFORE any of them is consumed (memory-level parallelism: the gather is L2-latency bound),
// then blended in the reference's corner order tnw, tne, tsw, tse, bnw, bne, bsw, bse.
template <int L>
__device__ __forceinline__ void gather_level(const SceneDev &sc, const GridCoord &g, int hi,
                                             float (&out)[lvl_c(L) / 2]) {
    constexpr int C = lvl_c(L), HALF = C / 2;
    const int D = sc.dhw[L][0], H = sc.dhw[L][1], W = sc.dhw[L][2];
    // grid_sampler_unnormalize, align_corners=True: ((g + 1) / 2) * (size - 1)
    float ix = __fmul_rn(__fdiv_rn(__fadd_rn(g.gw, 1.f), 2.f), (float)(W - 1));
    float iy = __fmul_rn(__fdiv_rn(__fadd_rn(g.gh, 1.f), 2.f), (float)(H - 1));
    float iz = __fmul_rn(__fdiv_rn(__fadd_rn(g.gd, 1.f), 2.f), (float)(D - 1));
    // keep float->int conversion defined for far-away points (all corners are then out of bounds)
    ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);
    iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);
    iz = fminf(fmaxf(iz, -2.f), (float)D + 1.f);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx[2] = {(fx + 1.f) - ix, ix - fx};
    const float wy[2] = {(fy + 1.f) - iy, iy - fy};
    const float wz[2] = {(fz + 1.f) - iz, iz - fz};
    const float *vb = sc.vol[L] + hi * HALF;
    const f32x4 *cp[8];
    float cw[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
        const bool inb = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D;
        cw[corner] = inb ? (wx[dx] * wy[dy]) * wz[dz] : 0.f;
        const int xc = min(max(xx, 0), W - 1), yc = min(max(yy, 0), H - 1), zc = min(max(zz, 0), D - 1);
        cp[corner] = reinterpret_cast<const f32x4 *>(vb + ((size_t)(zc * H + yc) * W + xc) * C);
    }
#pragma unroll
    for (int grp = 0; grp < HALF / 16; ++grp) {
        f32x4 v[8][4];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[corner][q] = cp[corner][grp * 4 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                a.x = fmaf(cw[corner], v[corner][q].x, a.x);
                a.y = fmaf(cw[corner], v[corner][q].y, a.y);
                a.z = fmaf(cw[corner], v[corner][q].z, a.z);
                a.w = fmaf(cw[corner], v[corner][q].w, a.w);
            }
            out[grp * 16 + q * 4 + 0] = a.x;
            out[grp * 16 + q * 4 + 1] = a.y;
            out[grp * 16 + q * 4 + 2] = a.z;
            out[grp * 16 + q * 4 + 3] = a.w;
        }
        // 32 VMEM reads, then the 128 FMAs that consume them; nothing moves across the group boundary
        __builtin_amdgcn_sched_group_barrier(0x020, 32, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 128, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------- cooperative (coalesced) gather
// Neighbouring rays at the same depth step touch the same few voxels (a 512x512 view spaces rays
// ~3.5 mm apart; level-3/4 voxels are 4-8 cm).  The per-lane gather above issues one cache-line
// request per lane per instruction (64 distinct lines -> TA address-rate bound); here the wave first
// fetches the bounding box of voxels it needs ONCE, fully coalesced (consecutive lanes read consecutive
// 16-byte pieces of a voxel's channel vector), into a wave-private LDS tile, and every lane then reads
// its 8 corners from LDS.  Falls back to the per-lane gather when the box does not fit (random rays).
// butterfly exchange inside each 32-lane half with ds_swizzle (bit-mask mode: and 0x1f, xor MASK): unlike
// __shfl_xor it needs no per-lane index register (a loop invariant the allocator would spill and reload)
template <int MASK>
__device__ __forceinline__ float swz_xor(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x1f | (MASK << 10)));
}
// x[lane] + x[lane ^ 32] in every lane (v_permlane32_swap exchanges the two halves)
__device__ __forceinline__ float add_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_min32(float v) {
    v = fminf(v, swz_xor<16>(v));
    v = fminf(v, swz_xor<8>(v));
    v = fminf(v, swz_xor<4>(v));
    v = fminf(v, swz_xor<2>(v));
    v = fminf(v, swz_xor<1>(v));
    return v;  // lanes j and j+32 hold the same sample, so 5 butterfly steps cover the wave
}
__device__ __forceinline__ float wave_max32(float v) {
    v = fmaxf(v, swz_xor<16>(v));
    v = fmaxf(v, swz_xor<8>(v));
    v = fmaxf(v, swz_xor<4>(v));
    v = fmaxf(v, swz_xor<2>(v));
    v = fmaxf(v, swz_xor<1>(v));
    return v;
}

struct WaveBox {
    GridCoord lo, hi;  // component-wise min / max of the normalised coordinates over the wave
};

__device__ __forceinline__ WaveBox wave_box(const GridCoord &g) {
    WaveBox b;
    b.lo.gw = wave_min32(g.gw);
    b.lo.gh = wave_min32(g.gh);
    b.lo.gd = wave_min32(g.gd);
    b.hi.gw = wave_max32(g.gw);
    b.hi.gh = wave_max32(g.gh);
    b.hi.gd = wave_max32(g.gd);
    return b;
}

__device__ __forceinline__ float unnorm_clamped(float gcoord, int size) {
    float i = __fmul_rn(__fdiv_rn(__fadd_rn(gcoord, 1.f), 2.f), (float)(size - 1));
    return fminf(fmaxf(i, -2.f), (float)size + 1.f);
}

template <int L, int BUF_BYTES>
__device__ __forceinline__ void gather_level_coop(const SceneDev &sc, const GridCoord &g, const WaveBox &wb, int hi,
                                                  int lane, char *buf, float (&out)[lvl_c(L) / 2]) {
    constexpr int C = lvl_c(L), HALF = C / 2, PC = C / 4;  // PC 16-byte pieces per voxel
    constexpr int MAX_IT = BUF_BYTES / 1024;
    const int D = sc.dhw[L][0], H = sc.dhw[L][1], W = sc.dhw[L][2];
    // index box of the wave (monotone in the normalised coordinate, same formula as the per-lane indices)
    const int xlo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.lo.gw, W)), 0), W - 1));
    const int ylo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.lo.gh, H)), 0), H - 1));
    const int zlo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.lo.gd, D)), 0), D - 1));
    const int xhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.hi.gw, W)) + 1, 0), W - 1));
    const int yhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.hi.gh, H)) + 1, 0), H - 1));
    const int zhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_clamped(wb.hi.gd, D)) + 1, 0), D - 1));
    const int nx = xhi - xlo + 1, ny = yhi - ylo + 1, nz = zhi - zlo + 1;
    const int pieces = nx * ny * nz * PC;
    if (pieces > BUF_BYTES / 16) {  // wave-uniform
        gather_level<L>(sc, g, hi, out);
        return;
    }
    // ---- fill the tile: piece p = (voxel v, 16-byte quad q) -> buf[p * 16]
    {
        const float rcp_xy = __builtin_amdgcn_rcpf((float)(nx * ny)), rcp_x = __builtin_amdgcn_rcpf((float)nx);  // v + 0.5 absorbs 1 ulp
        f32x4 t[MAX_IT];
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            if (it * 64 < pieces) {  // wave-uniform
                const int p = min(it * 64 + lane, pieces - 1);
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
        for (int it = 0; it < MAX_IT; ++it) {
            if (it * 64 < pieces) {
                const int p = it * 64 + lane;
                if (p < pieces) *reinterpret_cast<f32x4 *>(buf + p * 16) = t[it];
            }
        }
    }
    // ---- per-lane corners from LDS (same index / weight arithmetic as gather_level)
    const float ix = unnorm_clamped(g.gw, W), iy = unnorm_clamped(g.gh, H), iz = unnorm_clamped(g.gd, D);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx[2] = {(fx + 1.f) - ix, ix - fx};
    const float wy[2] = {(fy + 1.f) - iy, iy - fy};
    const float wz[2] = {(fz + 1.f) - iz, iz - fz};
    const f32x4 *cp[8];
    float cw[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
        const bool inb = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D;
        cw[corner] = inb ? (wx[dx] * wy[dy]) * wz[dz] : 0.f;
        const int xc = min(max(xx, xlo), xhi), yc = min(max(yy, ylo), yhi), zc = min(max(zz, zlo), zhi);
        const int lv = ((zc - zlo) * ny + (yc - ylo)) * nx + (xc - xlo);
        cp[corner] = reinterpret_cast<const f32x4 *>(buf + (lv * C + hi * HALF) * 4);
    }
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const f32x4 v = cp[corner][q];
            a.x = fmaf(cw[corner], v.x, a.x);
            a.y = fmaf(cw[corner], v.y, a.y);
            a.z = fmaf(cw[corner], v.z, a.z);
            a.w = fmaf(cw[corner], v.w, a.w);
        }
        out[q * 4 + 0] = a.x;
        out[q * 4 + 1] = a.y;
        out[q * 4 + 2] = a.z;
        out[q * 4 + 3] = a.w;
    }
}

// all four levels through the cooperative gather (tile = wave-private LDS of TILE bytes)
template <int TILE>
__device__ __forceinline__ void gather_features(const SceneDev &sc, float px, float py, float pz, int hi, int lane,
                                                char *tile, float (&F)[176]) {
    const GridCoord g = grid_coords(sc, px, py, pz);
    const WaveBox wb = wave_box(g);
    float f0[16], f1[32], f2[64], f3[64];
    gather_level_coop<0, TILE>(sc, g, wb, hi, lane, tile, f0);
    gather_level_coop<1, TILE>(sc, g, wb, hi, lane, tile, f1);
    gather_level_coop<2, TILE>(sc, g, wb, hi, lane, tile, f2);
    gather_level_coop<3, TILE>(sc, g, wb, hi, lane, tile, f3);
#pragma unroll
    for (int i = 0; i < 16; ++i) F[i] = f0[i];
#pragma unroll
    for (int i = 0; i < 32; ++i) F[16 + i] = f1[i];
#pragma unroll
    for (int i = 0; i < 64; ++i) F[48 + i] = f2[i];
#pragma unroll
    for (int i = 0; i < 64; ++i) F[112 + i] = f3[i];
}

// XCD-aware wave-group remap: consecutive blocks land on different XCDs (block b -> XCD b % 8).  Every XCD takes CHUNKS of
// NB_XCD_CHUNK consecutive ray groups (64 groups = one row of 8 x 8 pixel tiles of a 512-wide image), dealt round robin: the groups
// an XCD's L2 serves are strips of neighbouring tiles, and any run of >= 8 chunks — a rank's share of an image split over several
// GPUs, whose other groups are dead slots — still spreads over all eight XCDs.  (Round 4 gave every XCD one contiguous eighth of
// the list: a 1/8 share of whole tile rows then sat on ONE XCD and marched in 11 ms instead of 1.8, bench extras strong8_*.)
// The groups behind the last full round of 8 chunks are split contiguously.
#ifndef NB_XCD_CHUNK
#define NB_XCD_CHUNK 64
#endif
__device__ __forceinline__ int xcd_remap(int b, int n) {
    constexpr int C = NB_XCD_CHUNK;
    const int full = (n / (8 * C)) * (8 * C);
    if (b < full) {
        const int x = b % 8, i = b / 8;
        return ((i / C) * 8 + x) * C + i % C;
    }
    const int m = n - full, bb = b - full;
    const int q = m / 8, r = m % 8, x = bb % 8, i = bb / 8;
    return full + (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ float z_lin(float near, float far, float t) {
    // near * (1 - t) + far * t   (if_clight_renderer.py:14), no contraction
    return __fadd_rn(__fmul_rn(near, __fsub_rn(1.f, t)), __fmul_rn(far, t));
}


// raw2outputs (nerf_net_utils.py:19-46) for one sample, carried front to back
struct RayAccum {
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, depth = 0.f, accw = 0.f;
    __device__ __forceinline__ float add(const float (&out)[4], float z_cur, float dist_scaled) {
        const float sig = fmaxf(out[3], 0.f);
        const float alpha = 1.f - expf(-sig * dist_scaled);
        const float w = alpha * T;
        T = T * (__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f));
        cr = fmaf(w, 1.f / (1.f + expf(-out[0])), cr);
        cg = fmaf(w, 1.f / (1.f + expf(-out[1])), cg);
        cb = fmaf(w, 1.f / (1.f + expf(-out[2])), cb);
        depth = fmaf(w, z_cur, depth);
        accw += w;
        return w;
    }
    __device__ __forceinline__ void store(const MarchArgs &a, long long ray) {
        if (a.white_bkgd) {
            cr += 1.f - accw;
            cg += 1.f - accw;
            cb += 1.f - accw;
        }
        a.rgb_map[ray * 3 + 0] = cr;
        a.rgb_map[ray * 3 + 1] = cg;
        a.rgb_map[ray * 3 + 2] = cb;
        const float q = depth / accw;  // NaN when acc == 0: torch.max propagates it (nerf_net_utils.py:44-45)
        a.disp_map[ray] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
        a.acc_map[ray] = accw;
        a.depth_map[ray] = depth;
    }
};

// `weights` [n_rays, S] is written one value per ray per depth step; a 4-byte store per lane at a 4*S-byte
// stride turns into one partial HBM write each (measured: WRITE_SIZE 739 MB per 512x512x64 launch for 67 MB of
// payload), and two 16-byte halves per 8 steps still cost 2x the payload (151 MB: HBM writes are 64-byte lines).
// Each lane therefore keeps EIGHT consecutive steps in registers — the hi = 0 half of a sample column steps
// 16m..16m+7, the hi = 1 half steps 16m+8..16m+15 — and stores them as two aligned 16-byte vectors, so a ray's
// two halves fill one 64-byte line with a single store instruction (S % 16 == 0; S % 8 == 0: 32-byte sectors as before).
struct WeightStore {
    float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __device__ __forceinline__ void push(const MarchArgs &a, long long ray, int s, int S, int hi, bool valid, float w) {
        if ((S & 7) != 0) {  // wave-uniform: odd sample counts take the simple path
            if (valid && hi == 0) a.weights[ray * S + s] = w;
            return;
        }
        if ((S & 15) == 0) {
            const int slot = s & 15;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (slot == hi * 8 + i) q[i] = w;
            if (slot == 15 && valid) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(a.weights + ray * S + (s - 15) + hi * 8);
                dst[0] = f32x4{q[0], q[1], q[2], q[3]};
                dst[1] = f32x4{q[4], q[5], q[6], q[7]};
            }
            return;
        }
        const int slot = s & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (slot == hi * 4 + i) q[i] = w;
        if (slot == 7 && valid)
            *reinterpret_cast<f32x4 *>(a.weights + ray * S + (s - 7) + hi * 4) = f32x4{q[0], q[1], q[2], q[3]};
    }
};

inline int fill_scene(const nb_scene *s, SceneDev *d) {
    for (int l = 0; l < 4; ++l) {
        NB_REQUIRE(s->vol[l] != nullptr || s->fold != nullptr, "nb_scene.vol[%d] is NULL", l);
        d->vol[l] = s->vol[l];
        // the kernels index a level with 24-bit multiplies (v_mad_i32_i24) and 32-bit linear voxel numbers
        NB_REQUIRE((long long)s->vol_dhw[l][0] * s->vol_dhw[l][1] * s->vol_dhw[l][2] < (1ll << 24),
                   "nb_scene.vol_dhw[%d]: %d x %d x %d voxels (a level holds fewer than 2^24)", l, s->vol_dhw[l][0], s->vol_dhw[l][1], s->vol_dhw[l][2]);
        for (int k = 0; k < 3; ++k) {
            NB_REQUIRE(s->vol_dhw[l][k] >= 1 && s->vol_dhw[l][k] < (1 << 23), "nb_scene.vol_dhw[%d][%d] = %d", l, k, s->vol_dhw[l][k]);
            d->dhw[l][k] = s->vol_dhw[l][k];
            d->fm1[l][k] = (float)(s->vol_dhw[l][k] - 1);
            d->fp1[l][k] = (float)s->vol_dhw[l][k] + 1.f;
        }
    }
    NB_REQUIRE(s->pose != nullptr, "nb_scene.pose is NULL (device block R[9] | Th[3] | bounds_min[3])");
    d->pose = s->pose;
    for (int k = 0; k < 3; ++k) {
        d->vs[k] = s->voxel_size[k];
        d->osh[k] = (float)s->out_sh[k];
        NB_REQUIRE(s->voxel_size[k] > 0.f && s->out_sh[k] > 0, "nb_scene voxel_size/out_sh must be positive");
    }
    return NB_OK;
}


// fold != NULL is required (and checked) by the kernels that read it only
inline int fill_fold(const nb_scene *s, FoldDev *d) {
    const nb_fold *f = s->fold;
    NB_REQUIRE(f != nullptr, "nb_scene.fold is NULL: NB_PREC_F16F6 marches the fc_0-folded planes of nb_fold_build");
    NB_REQUIRE(f->urows != nullptr && f->zero_row >= 0 && f->zero_row < (1 << 21), "nb_fold: urows NULL or zero_row %d out of range",
               f->zero_row);
    d->urows = reinterpret_cast<const char *>(f->urows);
    for (int l = 0; l < 4; ++l) {
        NB_REQUIRE(f->grid[l] != nullptr && f->row_base[l] >= 0 && f->row_base[l] <= f->zero_row &&
                       (l == 0 || f->row_base[l] >= f->row_base[l - 1]),
                   "nb_fold: level %d: NULL grid or row_base %d outside [previous level, zero_row %d]", l, f->row_base[l], f->zero_row);
        d->grid[l] = f->grid[l];
        d->row_base[l] = f->row_base[l];
    }
    d->zero_off = (unsigned)f->zero_row * 1024u;
    return NB_OK;
}

inline int fill_cull(const nb_cull *c, CullDev *d) {
    d->n_views = 0;
    if (!c) return NB_OK;
    NB_REQUIRE(c->n_views >= 1 && c->n_views <= NB_MAX_CULL_VIEWS && c->H > 0 && c->W > 0, "nb_cull: n_views %d, H %d, W %d",
               c->n_views, c->H, c->W);
    d->n_views = c->n_views;
    d->H = c->H;
    d->W = c->W;
    d->pre = c->pre_affine;
    NB_REQUIRE(c->msk != nullptr && c->cam != nullptr, "nb_cull: msk / cam is NULL");
    NB_REQUIRE(!c->pre_affine || c->snap != nullptr, "nb_cull: pre_affine needs snap (R0 | Th0)");
    d->msk = c->msk;
    d->cam = c->cam;
    d->snap = c->snap;
    return NB_OK;
}

inline void fill_march_args(MarchArgs &a, const float *packed, const float *latent_bias, const float *ray_o,
                            const float *ray_d, const float *near, const float *far, long long n_rays, int n_samples,
                            const float *t_vals, const float *t_rand, const int *ray_order, int white_bkgd,
                            float *rgb_map, float *disp_map,
                            float *acc_map, float *weights, float *depth_map, float *raw) {
    a.pk = packed;
    a.lb = latent_bias;
    a.ray_o = ray_o;
    a.ray_d = ray_d;
    a.near = near;
    a.far = far;
    a.t_vals = t_vals;
    a.t_rand = t_rand;
    a.ray_order = ray_order;
    a.rgb_map = rgb_map;
    a.disp_map = disp_map;
    a.acc_map = acc_map;
    a.weights = weights;
    a.depth_map = depth_map;
    a.raw = raw;
    a.n_rays = n_rays;
    a.n_samples = n_samples;
    a.white_bkgd = white_bkgd;
    a.n_wave_groups = nb_ceil_div(n_rays, 128);
}

// fc_0 folded into the volumes, fp16 + scaled-6-bit for the remaining layers (nb_march_fold.hip); stream_off = float offset of
// its weight stream inside the packed blob, stats: 6 ints behind it (nb_mlp_six_bit_stats_offset)
long long fold_stream_floats();
int pack_fold_stream(const nb_mlp_params *p, float *packed, long long stream_off, hipStream_t st);
int launch_march_fold(MarchArgs a, long long stream_off, hipStream_t st);
int launch_points_fold(MarchArgs a, int density_only, long long stream_off, hipStream_t st);

}  // namespace nbm
