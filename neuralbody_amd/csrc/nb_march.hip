// nb_march.hip — fused Neural Body decode / march kernels for gfx950 (MI355X).
//
// One WAVE owns 32 samples (columns of the MFMA B operand); the two half-waves of a sample
// column (lane j and lane j+32) hold complementary halves of its feature vector.  The whole
// per-sample pipeline
//     world point -> canonical -> grid coords -> 4-level trilinear gather (352 ch)
//     -> fc_0/fc_1/fc_2 -> alpha_fc, (feature_fc . latent_fc merged) -> view_fc -> rgb_fc
// runs with activations RESIDENT IN REGISTERS: layer l is computed transposed,
//     H_l^T[feature, sample] = W_l[feature, k] . H_{l-1}^T[k, sample]
// with v_mfma_f32_32x32x2_f32 (weights = A operand, activations = B operand), so the C/D
// fragment of layer l (lane = sample column, 16 registers = 16 feature rows) is *already* the
// B fragment of layer l+1: no LDS round trip, no conversion.  The k ordering this induces
// (rows {rho(r), rho(r)+4} of each 32-row tile pair up as one K=2 chunk) is absorbed into
// the host-side weight packing (nb_mlp_pack).  Weights stream from L2/L1 as 16-byte,
// fully coalesced fragment loads (1 KiB per wave-instruction).
//
// In ray mode a wave marches 32 neighbouring rays front to back, one depth step per
// iteration, and composites on the fly (transmittance carried in a register), so nothing but
// the per-ray outputs ever reaches HBM.
//
// Reference semantics restated here (paths into zju3dv/neuralbody):
//   lib/networks/renderer/if_clight_renderer.py:11-27,54-92   sampling, viewdir, chunk body
//   lib/networks/latent_xyzc.py:41-72,91-126                  transform, grid coords, MLP
//   lib/networks/embedder.py:10-36                            positional encoding
//   lib/networks/renderer/nerf_net_utils.py:6-51              raw2outputs
//   ATen grid_sampler_3d (trilinear, zeros padding, align_corners=True)

#include "nb_march_common.h"

using namespace nbm;

#define NB_PF 8  // weight fragments in flight per wave
#define NB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

// ---------------------------------------------------------------- packed blob layout (floats)
constexpr int G0 = 44;  // fc_0: 352 inputs = 176 K=2 chunks = 44 groups of 4 chunks
constexpr int GH = 32;  // hidden 256 inputs = 128 chunks = 32 groups
constexpr int GV = 44;  // view_fc: 128 chunks (latent_fc out) + 45 PE chunks + 3 zero chunks
constexpr int OFF_L0 = 0;
constexpr int OFF_B0 = OFF_L0 + 8 * G0 * 256;
constexpr int OFF_L1 = OFF_B0 + 256;
constexpr int OFF_B1 = OFF_L1 + 8 * GH * 256;
constexpr int OFF_L2 = OFF_B1 + 256;
constexpr int OFF_B2 = OFF_L2 + 8 * GH * 256;
constexpr int OFF_AW = OFF_B2 + 256;  // alpha_fc weights [2][128]
constexpr int OFF_AB = OFF_AW + 256;  // alpha_fc bias (4 floats, 1 used)
constexpr int OFF_L4 = OFF_AB + 4;    // merged latent_fc[:, :256] @ feature_fc
constexpr int OFF_LV = OFF_L4 + 8 * GH * 256;
constexpr int OFF_BV = OFF_LV + 4 * GV * 256;
constexpr int OFF_RW = OFF_BV + 128;  // rgb_fc weights [3][2][64]
constexpr int OFF_RB = OFF_RW + 384;  // rgb_fc bias (4 floats, 3 used)
constexpr int PACK_SIZE = OFF_RB + 4;

// ---------------------------------------------------------------- one MLP layer on MFMA
// The A-operand stream of a layer is NT*NG consecutive 1-KiB fragments; it is software-pipelined
// through a ring of PF in-flight 16-byte loads (PF*256 MFMA cycles of lookahead) and the order is
// pinned with sched_barrier so the loads are not sunk back next to their consumers.
template <int NT, int NG, int PF, typename InF>
__device__ __forceinline__ void mlp_layer(const float *__restrict__ wp, const float *__restrict__ bp,
                                          f32x16 (&acc)[NT], InF in, int lane) {
    const int hi = lane >> 5;
    constexpr int TOTAL = NT * NG;
    const f32x4 *a = reinterpret_cast<const f32x4 *>(wp) + lane;
    f32x4 ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = a[i * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 *b4 = reinterpret_cast<const f32x4 *>(bp + (t * 2 + hi) * 16);
        const f32x4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
        f32x16 c = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w,
                    b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int idx = t * NG + g;
            const f32x4 av = ring[idx % PF];
            if (idx + PF < TOTAL) ring[idx % PF] = a[(idx + PF) * 64];
            c = NB_MFMA(av.x, in(4 * g + 0), c);
            c = NB_MFMA(av.y, in(4 * g + 1), c);
            c = NB_MFMA(av.z, in(4 * g + 2), c);
            c = NB_MFMA(av.w, in(4 * g + 3), c);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[t] = c;
    }
}

template <int NT>
__device__ __forceinline__ void relu_tiles(f32x16 (&h)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[t][r] = relu1(h[t][r]);
}

// out[f] for the feature held in (tile t, register r) of this lane
template <int NT>
__device__ __forceinline__ void dump_tiles(const f32x16 (&h)[NT], float *dst, int hi) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[32 * t + tile_row(r, hi)] = h[t][r];
}

// ---------------------------------------------------------------- per-sample decode
// Returns sigma in out[3] and (unless DENSITY_ONLY) rgb logits in out[0..2]; both half-waves
// of a sample column receive the same values.
constexpr int F32_TILE_BYTES = 16384;  // per-wave voxel tile of the cooperative gather (this kernel has no other LDS use)

template <bool DENSITY_ONLY, bool DBG>
__device__ __forceinline__ void decode(const SceneDev &sc, const float *__restrict__ pk, const float *__restrict__ lb,
                                       float px, float py, float pz, const float (&pe)[N_PE], int lane, char *tile,
                                       float (&out)[4], float *dbg) {
    const int hi = lane >> 5;
    f32x16 h[8], acc[8];
    {
        float F[176];
        gather_features<F32_TILE_BYTES>(sc, px, py, pz, hi, lane, tile, F);
        if (DBG && dbg) {
#pragma unroll
            for (int q = 0; q < 176; ++q) dbg[TAP_F + col_feat(q, hi)] = F[q];
        }
        mlp_layer<8, G0, NB_PF>(pk + OFF_L0, pk + OFF_B0, acc, [&](int q) { return F[q]; }, lane);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = acc[t];
    relu_tiles(h);
    if (DBG && dbg) dump_tiles(h, dbg + TAP_H1, hi);
    mlp_layer<8, GH, NB_PF>(pk + OFF_L1, pk + OFF_B1, acc, [&](int q) { return h[q >> 4][q & 15]; }, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = acc[t];
    relu_tiles(h);
    if (DBG && dbg) dump_tiles(h, dbg + TAP_H2, hi);
    mlp_layer<8, GH, NB_PF>(pk + OFF_L2, pk + OFF_B2, acc, [&](int q) { return h[q >> 4][q & 15]; }, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = acc[t];
    relu_tiles(h);
    if (DBG && dbg) dump_tiles(h, dbg + TAP_H3, hi);
    // alpha_fc on the VALU: each half-wave holds 128 of the 256 features of its sample
    {
        const f32x4 *aw = reinterpret_cast<const f32x4 *>(pk + OFF_AW + hi * 128);
        float s = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 32; ++q4) {
            const f32x4 w = aw[q4];
            s = fmaf(w.x, h[q4 >> 2][(q4 & 3) * 4 + 0], s);
            s = fmaf(w.y, h[q4 >> 2][(q4 & 3) * 4 + 1], s);
            s = fmaf(w.z, h[q4 >> 2][(q4 & 3) * 4 + 2], s);
            s = fmaf(w.w, h[q4 >> 2][(q4 & 3) * 4 + 3], s);
        }
        s = add_halves(s);
        out[3] = s + pk[OFF_AB];
    }
    if (DENSITY_ONLY) return;
    // feature_fc and latent_fc[:, :256] merged; per-frame latent folded into the bias `lb`
    mlp_layer<8, GH, NB_PF>(pk + OFF_L4, lb, acc, [&](int q) { return h[q >> 4][q & 15]; }, lane);
    if (DBG && dbg) dump_tiles(acc, dbg + TAP_G, hi);
    // view_fc on [latent_fc out (256) | PE(viewdir) | PE(xyz)]
    f32x16 v[4];
    mlp_layer<4, GV, NB_PF>(
        pk + OFF_LV, pk + OFF_BV, v,
        [&](int q) { return q < 128 ? acc[q >> 4][q & 15] : (q - 128 < N_PE ? pe[q - 128 < N_PE ? q - 128 : 0] : 0.f); },
        lane);
    relu_tiles(v);
    if (DBG && dbg) {
        dump_tiles(v, dbg + TAP_V, hi);
        dump_pe(pe, dbg + TAP_PE, hi);
    }
    // rgb_fc on the VALU
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const f32x4 *rw = reinterpret_cast<const f32x4 *>(pk + OFF_RW + (ch * 2 + hi) * 64);
        float s = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 16; ++q4) {
            const f32x4 w = rw[q4];
            s = fmaf(w.x, v[q4 >> 2][(q4 & 3) * 4 + 0], s);
            s = fmaf(w.y, v[q4 >> 2][(q4 & 3) * 4 + 1], s);
            s = fmaf(w.z, v[q4 >> 2][(q4 & 3) * 4 + 2], s);
            s = fmaf(w.w, v[q4 >> 2][(q4 & 3) * 4 + 3], s);
        }
        s = add_halves(s);
        out[ch] = s + pk[OFF_RB + ch];
    }
}

// ---------------------------------------------------------------- point-mode kernel
template <bool DENSITY_ONLY, bool DBG>
__global__ __launch_bounds__(256) void nb_points_kernel(MarchArgs a) {
    __shared__ __attribute__((aligned(16))) char tiles[4 * F32_TILE_BYTES];
    char *tile = tiles + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * F32_TILE_BYTES;
    const int lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long long idx = wave * 32 + j;
    if (wave * 32 >= a.n_pts) return;  // wave-uniform
    const bool valid = idx < a.n_pts;
    if (!valid) idx = a.n_pts - 1;
    const float px = a.wpts[idx * 3 + 0], py = a.wpts[idx * 3 + 1], pz = a.wpts[idx * 3 + 2];
    float pe[N_PE];
    if (!DENSITY_ONLY) {
        const float vx = a.viewdir[idx * 3 + 0], vy = a.viewdir[idx * 3 + 1], vz = a.viewdir[idx * 3 + 2];
        pe_view(pe, vx, vy, vz, hi);
        pe_xyz(pe, px, py, pz, vx, vy, vz, hi);
    } else {
#pragma unroll
        for (int c = 0; c < N_PE; ++c) pe[c] = 0.f;
    }
    float out[4];
    float *dbg = (DBG && a.dbg && valid) ? a.dbg + idx * TAP_WIDTH : nullptr;
    decode<DENSITY_ONLY, DBG>(a.sc, a.pk, a.lb, px, py, pz, pe, lane, tile, out, dbg);
    if (valid && hi == 0) {
        if (DENSITY_ONLY) {
            a.raw_out[idx] = out[3];
        } else {
            *reinterpret_cast<f32x4 *>(a.raw_out + idx * 4) = f32x4{out[0], out[1], out[2], out[3]};
        }
    }
}

// ---------------------------------------------------------------- ray-mode kernel (march + composite)
__global__ __launch_bounds__(256) void nb_march_kernel(MarchArgs a) {
    __shared__ __attribute__((aligned(16))) char tiles[4 * F32_TILE_BYTES];
    char *tile = tiles + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * F32_TILE_BYTES;
    const int lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
    const int grp = xcd_remap(blockIdx.x, a.n_wave_groups);
    const long long wave = (long long)grp * 4 + (threadIdx.x >> 6);
    if (wave * 32 >= (a.ray_order ? a.n_slots : a.n_rays)) return;  // wave-uniform
    long long ray = wave * 32 + j;
    bool valid = ray < a.n_rays;
    if (!valid) ray = a.n_rays - 1;
    if (a.ray_order) {
        if (a.ray_order[wave * 32] == NB_SLOT_DEAD) return;  // an empty group of slots (wave-uniform)
        const int v = a.ray_order[wave * 32 + j];
        valid = v >= 0;
        ray = valid ? v : -(long long)v - 1;
    }
    const int S = a.n_samples;
    const float ox = a.ray_o[ray * 3 + 0], oy = a.ray_o[ray * 3 + 1], oz = a.ray_o[ray * 3 + 2];
    const float dx = a.ray_d[ray * 3 + 0], dy = a.ray_d[ray * 3 + 1], dz = a.ray_d[ray * 3 + 2];
    const float near = a.near[ray], far = a.far[ray];
    const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    const float vx = dx / dn, vy = dy / dn, vz = dz / dn;  // if_clight_renderer.py:68
    float pe[N_PE];
    pe_view(pe, vx, vy, vz, hi);
    const float *tr = a.t_rand ? a.t_rand + ray * S : nullptr;

    auto z_at = [&](int s) -> float {
        const float zc = z_lin(near, far, a.t_vals[s]);
        if (!tr) return zc;
        // stratified jitter (if_clight_renderer.py:16-23)
        const float lower = s == 0 ? zc : 0.5f * __fadd_rn(zc, z_lin(near, far, a.t_vals[s - 1]));
        const float upper = s == S - 1 ? zc : 0.5f * __fadd_rn(z_lin(near, far, a.t_vals[s + 1]), zc);
        return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), tr[s]));
    };

    RayAccum ra;
    WeightStore wstore;
    float z_cur = z_at(0);
    for (int s = 0; s < S; ++s) {
        const float z_next = (s + 1 < S) ? z_at(s + 1) : 0.f;
        const float px = __fadd_rn(ox, __fmul_rn(dx, z_cur));
        const float py = __fadd_rn(oy, __fmul_rn(dy, z_cur));
        const float pz = __fadd_rn(oz, __fmul_rn(dz, z_cur));
        pe_xyz(pe, px, py, pz, vx, vy, vz, hi);
        float out[4];
        // the weight stream is loop-invariant: hide the base pointers from LICM, which would otherwise
        // hoist all ~5000 fragment loads out of the depth loop and spill them
        // (an opaque zero offset keeps the pointers in the global address space)
        // (same for everything derived from the lane id: recompute inside the body instead of spilling it)
        int zero = 0, lane_i = lane;
        asm volatile("" : "+s"(zero), "+v"(lane_i));
        const bool ins = a.cull.n_views == 0 || cull_inside(a.cull, a.sc, px, py, pz);
        if (__any(ins)) {
            decode<false, false>(a.sc, a.pk + zero, a.lb + zero, px, py, pz, pe, lane_i, tile + zero, out, nullptr);
        }
        if (!ins) out[0] = out[1] = out[2] = out[3] = 0.f;  // culled sample: raw = 0 (if_clight_renderer_mmsk.py:54-59)
        float dist = (s + 1 < S) ? __fsub_rn(z_next, z_cur) : 1e10f;
        dist = __fmul_rn(dist, dn);
        const float w = ra.add(out, z_cur, dist);
        wstore.push(a, ray, s, S, hi, valid, w);
        if (valid && hi == 0 && a.raw)
            *reinterpret_cast<f32x4 *>(a.raw + (ray * S + s) * 4) = f32x4{out[0], out[1], out[2], out[3]};
        z_cur = z_next;
    }
    if (valid && hi == 0) ra.store(a, ray);
}

// ---------------------------------------------------------------- composite-only kernel (raw2outputs)
// One wave per ray, lanes = samples (2 samples per lane when S = 128 ...), wave-level
// exclusive product scan of (1 - alpha + 1e-10).
__global__ __launch_bounds__(256) void nb_composite_kernel(const float *__restrict__ raw, const float *__restrict__ z,
                                                           const float *__restrict__ ray_d, long long n_rays, int S,
                                                           int white_bkgd, float *rgb_map, float *disp_map,
                                                           float *acc_map, float *weights, float *depth_map) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float dx = ray_d[ray * 3], dy = ray_d[ray * 3 + 1], dz = ray_d[ray * 3 + 2];
    const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, depth = 0.f, accw = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool in = s < S;
        float zc = 0.f, alpha = 0.f;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (in) {
            zc = z[ray * S + s];
            r = *reinterpret_cast<const f32x4 *>(raw + (ray * S + s) * 4);
            float dist = (s + 1 < S) ? __fsub_rn(z[ray * S + s + 1], zc) : 1e10f;
            dist = __fmul_rn(dist, dn);
            alpha = 1.f - expf(-fmaxf(r.w, 0.f) * dist);
        }
        const float f = in ? __fadd_rn(__fsub_rn(1.f, alpha), 1e-10f) : 1.f;
        // inclusive product scan across the wave (sequential order is emulated up to fp32 rounding)
        float p = f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float o = __shfl_up(p, off);
            if (lane >= off) p *= o;
        }
        float excl = __shfl_up(p, 1);
        if (lane == 0) excl = 1.f;
        const float w = alpha * (T * excl);
        T = T * __shfl(p, 63);
        if (in) weights[ray * S + s] = w;
        float sr = w / (1.f + expf(-r.x)), sg = w / (1.f + expf(-r.y)), sb = w / (1.f + expf(-r.z));
        float sd = w * zc, sw = w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sr += __shfl_xor(sr, off);
            sg += __shfl_xor(sg, off);
            sb += __shfl_xor(sb, off);
            sd += __shfl_xor(sd, off);
            sw += __shfl_xor(sw, off);
        }
        cr += sr;
        cg += sg;
        cb += sb;
        depth += sd;
        accw += sw;
    }
    if (lane == 0) {
        if (white_bkgd) {
            cr += 1.f - accw;
            cg += 1.f - accw;
            cb += 1.f - accw;
        }
        rgb_map[ray * 3 + 0] = cr;
        rgb_map[ray * 3 + 1] = cg;
        rgb_map[ray * 3 + 2] = cb;
        const float q = depth / accw;
        disp_map[ray] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
        acc_map[ray] = accw;
        depth_map[ray] = depth;
    }
}

// ---------------------------------------------------------------- weight packing
__global__ void nb_pack_kernel(nb_mlp_params p, float *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= PACK_SIZE) return;
    float v = 0.f;
    auto a_pack = [&](int rel, int NG, const float *w, int ld, int kind) -> float {
        // rel = ((t*NG + g)*64 + lane)*4 + i
        const int i = rel & 3, lane = (rel >> 2) & 63, tg = rel >> 8, g = tg % NG, t = tg / NG;
        const int row = 32 * t + (lane & 31), hi = lane >> 5, q = 4 * g + i;
        int col;
        if (kind == 0) col = col_feat(q, hi);
        else if (kind == 1) col = col_hidden(q, hi);
        else col = q < 128 ? col_hidden(q, hi) : col_pe(q - 128, hi);
        return col < 0 ? 0.f : w[(size_t)row * ld + col];
    };
    auto b_pack = [&](int rel, const float *b) -> float {  // [t][hi][16]
        const int r = rel & 15, hi = (rel >> 4) & 1, t = rel >> 5;
        return b[32 * t + tile_row(r, hi)];
    };
    if (e < OFF_B0) v = a_pack(e - OFF_L0, G0, p.fc0_w, 352, 0);
    else if (e < OFF_L1) v = b_pack(e - OFF_B0, p.fc0_b);
    else if (e < OFF_B1) v = a_pack(e - OFF_L1, GH, p.fc1_w, 256, 1);
    else if (e < OFF_L2) v = b_pack(e - OFF_B1, p.fc1_b);
    else if (e < OFF_B2) v = a_pack(e - OFF_L2, GH, p.fc2_w, 256, 1);
    else if (e < OFF_AW) v = b_pack(e - OFF_B2, p.fc2_b);
    else if (e < OFF_AB) {
        const int rel = e - OFF_AW, hi = rel >> 7, q = rel & 127;
        v = p.alpha_w[col_hidden(q, hi)];
    } else if (e < OFF_L4) v = (e == OFF_AB) ? p.alpha_b[0] : 0.f;
    else if (e < OFF_LV) {
        // merged layer: W'[row][col] = sum_m latent_w[row][m] * feature_w[m][col]
        const int rel = e - OFF_L4;
        const int i = rel & 3, lane = (rel >> 2) & 63, tg = rel >> 8, g = tg % GH, t = tg / GH;
        const int row = 32 * t + (lane & 31), hi = lane >> 5, col = col_hidden(4 * g + i, hi);
        double s = 0.0;
        for (int m = 0; m < 256; ++m) s += (double)p.latent_w[row * 384 + m] * (double)p.feature_w[m * 256 + col];
        v = (float)s;
    } else if (e < OFF_BV) v = a_pack(e - OFF_LV, GV, p.view_w, 346, 2);
    else if (e < OFF_RW) v = b_pack(e - OFF_BV, p.view_b);
    else if (e < OFF_RB) {
        const int rel = e - OFF_RW, q = rel & 63, hi = (rel >> 6) & 1, ch = rel >> 7;
        v = p.rgb_w[ch * 128 + col_hidden(q, hi)];
    } else v = (e - OFF_RB < 3) ? p.rgb_b[e - OFF_RB] : 0.f;
    out[e] = v;
}

// out[0..255]: bias of the merged feature_fc / latent_fc layer, MFMA fragment order [tile][hi][16];
// out[256..383]: bias of view_fc with that whole (activation-free) layer pair folded in, i.e. view_b + view_w[:, :256] . (the
// former), same order — for the kernels that run feature_fc, latent_fc and view_fc as ONE linear layer (nb_march_fold.hip)
// One workgroup for the 256 outputs of the merged layer + one per output of the view layer (round 6: a single workgroup walked both
// stages in 34 us — the kernel sits between the encoder and the march of every new frame and in every training step; 129 workgroups:
// 18.6 us, bound by the uncoalesced row reads of the first stage: 9 workgroups or 16-byte loads measured 21-22).  Every workgroup forms
// the merged layer's bias lbn[m] = latent_b[m] + latent_w[m, :256] . feature_b + latent_w[m, 256:] . latent_row in fp64 (thread m:
// 384 FMAs in two chains), block 0 writes it out, block 1 + q folds it through row q of view_w[:, :256].
__global__ __launch_bounds__(256) void nb_latent_bias_kernel(nb_mlp_params p, const float *__restrict__ latent_row, float *__restrict__ out) {
    __shared__ double lbn[256];  // natural order
    __shared__ double part[4];
    const int rel = threadIdx.x;
    {
        const int row = rel;
        double s0 = (double)p.latent_b[row], s1 = 0.0;
        for (int m = 0; m < 256; m += 2) {
            s0 += (double)p.latent_w[row * 384 + m] * (double)p.feature_b[m];
            s1 += (double)p.latent_w[row * 384 + m + 1] * (double)p.feature_b[m + 1];
        }
        for (int m = 0; m < 128; m += 2) {
            s0 += (double)p.latent_w[row * 384 + 256 + m] * (double)latent_row[m];
            s1 += (double)p.latent_w[row * 384 + 256 + m + 1] * (double)latent_row[m + 1];
        }
        lbn[row] = s0 + s1;
    }
    __syncthreads();
    if (blockIdx.x == 0) {  // [t][hi][16] fragment order
        const int r = rel & 15, hi = (rel >> 4) & 1, t = rel >> 5;
        out[rel] = (float)lbn[32 * t + tile_row(r, hi)];
        return;
    }
    const int q = blockIdx.x - 1;  // position in fragment order of the view layer's 128 outputs
    const int r = q & 15, hi = (q >> 4) & 1, t = q >> 5;
    const int row = 32 * t + tile_row(r, hi);
    double v = (double)p.view_w[row * 346 + rel] * lbn[rel];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((rel & 63) == 0) part[rel >> 6] = v;
    __syncthreads();
    if (rel == 0) out[256 + q] = (float)((double)p.view_b[row] + ((part[0] + part[1]) + (part[2] + part[3])));
}

}  // namespace

extern "C" {

static long long fold_stream_off() { return PACK_SIZE; }
int64_t nb_mlp_pack_size(void) { return fold_stream_off() + nbm::fold_stream_floats(); }
int64_t nb_mlp_latent_bias_size(void) { return 384; }
int64_t nb_mlp_six_bit_stats_offset(void) { return nb_mlp_pack_size() - 8; }

static int check_params(const nb_mlp_params *p) {
    NB_REQUIRE(p != nullptr, "nb_mlp_params is NULL");
    const void *const *q = reinterpret_cast<const void *const *>(p);
    for (size_t i = 0; i < sizeof(nb_mlp_params) / sizeof(void *); ++i)
        NB_REQUIRE(q[i] != nullptr, "nb_mlp_params member %d is NULL", (int)i);
    return NB_OK;
}

int nb_mlp_pack(const nb_mlp_params *p, float *packed, void *stream) {
    return nb_mlp_pack_sections(p, packed, NB_PACK_ALL, stream);
}

int nb_mlp_pack_sections(const nb_mlp_params *p, float *packed, int sections, void *stream) {
    if (int rc = check_params(p)) return rc;
    NB_REQUIRE(packed != nullptr, "nb_mlp_pack: packed is NULL");
    NB_REQUIRE((sections & ~NB_PACK_ALL) == 0, "nb_mlp_pack_sections: unknown section bits %d", sections);
    // the fp32 section is always written: the other streams read the merged feature/latent layer from it
    hipLaunchKernelGGL(nb_pack_kernel, dim3(nb_ceil_div(PACK_SIZE, 256)), dim3(256), 0, (hipStream_t)stream, *p, packed);
    NB_CHECK_LAUNCH("nb_pack_kernel");
    if (sections & NB_PACK_F16F6)
        if (int rc = nbm::pack_fold_stream(p, packed, fold_stream_off(), (hipStream_t)stream)) return rc;
    return NB_OK;
}

int nb_mlp_latent_bias(const nb_mlp_params *p, const float *latent_row, float *out, void *stream) {
    if (int rc = check_params(p)) return rc;
    NB_REQUIRE(latent_row && out, "nb_mlp_latent_bias: NULL pointer");
    hipLaunchKernelGGL(nb_latent_bias_kernel, dim3(1 + 128), dim3(256), 0, (hipStream_t)stream, *p, latent_row, out);
    NB_CHECK_LAUNCH("nb_latent_bias_kernel");
    return NB_OK;
}

int nb_decode_points(const nb_scene *scene, const float *packed, const float *latent_bias, const float *wpts,
                     const float *viewdir, int64_t n, int density_only, float *raw_out, float *dbg, int precision,
                     void *stream) {
    NB_REQUIRE(scene && packed, "nb_decode_points: NULL scene / weights");
    NB_REQUIRE(n >= 0, "nb_decode_points: n = %lld", (long long)n);
    if (n == 0) return NB_OK;
    NB_REQUIRE(wpts && raw_out, "nb_decode_points: NULL wpts / raw_out");
    NB_REQUIRE(density_only || (viewdir && latent_bias), "nb_decode_points: viewdir / latent_bias required");
    NB_REQUIRE(precision == NB_PREC_F32 || precision == NB_PREC_F16F6, "nb_decode_points: precision %d", precision);
    NB_REQUIRE(!(dbg && precision == NB_PREC_F16F6), "nb_decode_points: NB_PREC_F16F6 has no activation tap (use NB_PREC_F32)");
    MarchArgs a = {};
    if (int rc = fill_scene(scene, &a.sc)) return rc;
    a.pk = packed;
    a.lb = latent_bias;
    a.wpts = wpts;
    a.viewdir = viewdir;
    a.n_pts = n;
    a.raw_out = raw_out;
    a.dbg = dbg;
    const dim3 grid(nb_ceil_div(n, 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (precision == NB_PREC_F16F6) {
        if (int rc = fill_fold(scene, &a.fold)) return rc;
        return nbm::launch_points_fold(a, density_only, fold_stream_off(), st);
    }
    for (int l = 0; l < 4; ++l) NB_REQUIRE(scene->vol[l] != nullptr, "nb_decode_points: NB_PREC_F32 reads nb_scene.vol[%d]", l);
    if (density_only) hipLaunchKernelGGL((nb_points_kernel<true, false>), grid, block, 0, st, a);
    else if (dbg) hipLaunchKernelGGL((nb_points_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((nb_points_kernel<false, false>), grid, block, 0, st, a);
    NB_CHECK_LAUNCH("nb_points_kernel");
    return NB_OK;
}

int nb_march(const nb_scene *scene, const float *packed, const float *latent_bias, const float *ray_o,
             const float *ray_d, const float *near, const float *far, int64_t n_rays, int32_t n_samples,
             const float *t_vals, const float *t_rand, const int32_t *ray_order, int64_t n_slots, const nb_cull *cull, int white_bkgd,
             float *rgb_map, float *disp_map,
             float *acc_map, float *weights, float *depth_map, float *raw, void *ill_scratch, int64_t ill_scratch_bytes, int precision,
             void *stream) {
    NB_REQUIRE(scene && packed && latent_bias, "nb_march: NULL scene / weights");
    NB_REQUIRE(n_rays >= 0 && n_samples >= 1, "nb_march: n_rays = %lld, n_samples = %d", (long long)n_rays, n_samples);
    if (n_rays == 0) return NB_OK;
    NB_REQUIRE(ray_o && ray_d && near && far && t_vals, "nb_march: NULL ray input");
    NB_REQUIRE(rgb_map && disp_map && acc_map && weights && depth_map, "nb_march: NULL output");
    NB_REQUIRE(!ray_order || (n_slots >= 64 && n_slots % 64 == 0 && n_slots < (1ll << 31)), "nb_march: ray_order with n_slots = %lld (a positive multiple of 64)",
               (long long)n_slots);
    MarchArgs a = {};
    if (int rc = fill_scene(scene, &a.sc)) return rc;
    if (int rc = fill_cull(cull, &a.cull)) return rc;
    fill_march_args(a, packed, latent_bias, ray_o, ray_d, near, far, n_rays, n_samples, t_vals, t_rand, ray_order,
                    white_bkgd,
                    rgb_map, disp_map, acc_map, weights, depth_map, raw);
    a.n_slots = ray_order ? n_slots : 0;
    if (ray_order) a.n_wave_groups = nb_ceil_div(n_slots, 128);
    NB_REQUIRE(precision == NB_PREC_F32 || precision == NB_PREC_F16F6, "nb_march: precision %d", precision);
    if (precision == NB_PREC_F16F6) {
        if (int rc = fill_fold(scene, &a.fold)) return rc;
        if (ill_scratch) {
            const long long cap = (ill_scratch_bytes - NB_ILL_SCRATCH_BYTES(0)) / (4 * NB_ILL_RECORD_FLOATS);
            NB_REQUIRE(((uintptr_t)ill_scratch & 15) == 0 && cap >= 1, "nb_march: ill_scratch must be 16-byte aligned and hold >= 1 record (%lld bytes given)",
                       (long long)ill_scratch_bytes);
            a.ill = static_cast<float *>(ill_scratch);
            a.ill_cap = (int)(cap < (1 << 24) ? cap : (1 << 24));
        }
        return nbm::launch_march_fold(a, fold_stream_off(), (hipStream_t)stream);
    }
    for (int l = 0; l < 4; ++l) NB_REQUIRE(scene->vol[l] != nullptr, "nb_march: NB_PREC_F32 reads nb_scene.vol[%d]", l);
    hipLaunchKernelGGL(nb_march_kernel, dim3(a.n_wave_groups), dim3(256), 0, (hipStream_t)stream, a);
    NB_CHECK_LAUNCH("nb_march_kernel");
    return NB_OK;
}

int nb_composite(const float *raw, const float *z_vals, const float *ray_d, int64_t n_rays, int32_t n_samples,
                 int white_bkgd, float *rgb_map, float *disp_map, float *acc_map, float *weights, float *depth_map,
                 void *stream) {
    NB_REQUIRE(n_rays >= 0 && n_samples >= 1, "nb_composite: bad sizes");
    if (n_rays == 0) return NB_OK;
    NB_REQUIRE(raw && z_vals && ray_d && rgb_map && disp_map && acc_map && weights && depth_map,
               "nb_composite: NULL pointer");
    hipLaunchKernelGGL(nb_composite_kernel, dim3(nb_ceil_div(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, raw,
                       z_vals, ray_d, (long long)n_rays, n_samples, white_bkgd, rgb_map, disp_map, acc_map, weights,
                       depth_map);
    NB_CHECK_LAUNCH("nb_composite_kernel");
    return NB_OK;
}

}  // extern "C"
