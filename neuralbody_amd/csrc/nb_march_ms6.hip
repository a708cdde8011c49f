// nb_march_ms6.hip — the "M-split" organisation of the f16f6 march for gfx950 (NB_PREC_F16F6; round 3) and the kernel behind
// nb_decode_points: the arithmetic of nb_march_f16.hip
//
//     W.X  ~=  W_h.X_h                               v_mfma_f32_32x32x16_f16            (products exact, fp32 accumulate)
//            + fp6(W_h).bf6(X_l) + fp6(W_l).bf6(X_h)  v_mfma_scale_f32_32x32x64_f8f6f4   (K = 64, E8M0 block scales)
//
// (feature_fc, latent_fc and view_fc folded into one layer) with the output features of every layer split over the waves
// of a workgroup, built from the counters of the ring kernel (profiles/r02_march_pmc.md: matrix pipe 0.34 busy, one wave
// per SIMD, every 2-KiB weight record feeding 1-2 MFMAs of a wave at the price of two LDS reads, a counted wait and half an
// LDS-DMA piece):
//
//   * a workgroup (4 waves) marches 64 rays = two 32-sample N tiles; wave w owns a QUARTER OF EVERY LAYER'S OUTPUT
//     FEATURES for all 64 samples, so a weight fragment feeds 2 MFMAs and comes straight from L2 into a register ring
//     (buffer loads, no LDS-DMA, no page barriers);
//   * activations live in LDS as ready-made B operands (fp16 heads by K=16 chunk, the two bf6 forms with their E8M0
//     scale by K=64 block) and are rewritten in place after every layer; the two output tiles of wave w ARE K block w of
//     the next layer, so the run-time block scale (max over 32 values of a lane) stays lane-local exactly as in the ring
//     kernel, and every lane publishes into its own fragment slot;
//   * <= 256 registers and 80 KiB of LDS: TWO workgroups per CU, whose phases overlap (two unsynchronised workgroups take
//     57 k cycles per depth step for both, one alone 54 k).
//
// Measured (profiles/r03_march_kernels.md): 8 % fewer cycles than the ring kernel and 2-4 % MORE time — streaming every
// wave's weight slice from L2 once per 64 samples (17 KiB per sample) costs 15 % of the clock under the board's power cap.
// The ring kernel therefore stays the default march; this one serves nb_decode_points (every point a one-sample ray) and
// precision 'f16f6'.
//
// Per-sample work (ray set-up, trilinear gather, positional encoding, compositing) is done by "owner" lanes: wave w,
// lane l owns sample 16 w + (l & 15) and, of that sample, channel quarter / axis `part` = l >> 4.
//
// K layout.  A layer input is a list of K=64 blocks; block b, half kh is the 32-value group of one lane ("half-block"):
// element e of it sits in fp16 chunk 4 b + e / 8 at B-fragment lane kg = kh, register element e % 8, and at element e
// (heads) / interleaved position (remainders) of the lane's bf6 operands of block b.  Weights are packed to match.
#include "nb_f6_ops.h"

using namespace nbm;

namespace {

// ---------------------------------------------------------------- weight stream (per wave), in 1-KiB pieces
// phase = NB K blocks x MT output tiles of this wave; per block: for each of its 4 chunks, for each tile: A16 (1 piece);
// then for each tile: A6h (W_h in fp6, 2 pieces: multiplies the remainder operand), for each tile: A6l (W_l, 2 pieces)
#ifndef NB_MS6_RING
#define NB_MS6_RING 8
#endif
constexpr int S_R = NB_MS6_RING;  // register ring depth in pieces
__host__ __device__ constexpr int phase_pieces(int nb, int mt) { return nb * mt * 8; }
constexpr int N_PH = 7;
// fc_0 in three K phases of 128 (pyramid level 3 | level 2 | levels 0 and 1 + 8 zero slots per lane), fc_1, fc_2,
// the folded colour head over fc_2's outputs (one tile per wave), view_fc over the positional encodings
constexpr int PH_NB[N_PH] = {2, 2, 2, 4, 4, 4, 2};
constexpr int PH_MT[N_PH] = {2, 2, 2, 2, 2, 1, 1};
__host__ __device__ constexpr int phase_p0(int ph) {
    int p = 0;
    for (int i = 0; i < ph; ++i) p += phase_pieces(PH_NB[i], PH_MT[i]);
    return p;
}
constexpr int P_A = phase_p0(0), P_B = phase_p0(1), P_C = phase_p0(2), P_L1 = phase_p0(3), P_L2 = phase_p0(4), P_VG = phase_p0(5),
              P_VP = phase_p0(6), P_TOTAL = phase_p0(7);
static_assert(P_TOTAL == 272, "pieces per wave per depth step");
static_assert(P_TOTAL % S_R == 0 && S_R % 2 == 0, "static ring slot of every piece, also across steps; six-bit fragments on even pieces");

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

// ---------------------------------------------------------------- LDS
constexpr int CH_BYTES = 2048;                 // one K=16 chunk: 2 N tiles x 1 KiB B fragment (64 lanes x 8 fp16)
constexpr int ACT16_BYTES = 16 * CH_BYTES;     // 32 KiB
constexpr int F6_BYTES = 2048;                 // one bf6 fragment: 64 lanes x (24 B data | scale word | pad), as two 1-KiB halves
constexpr int ACT6_OFF = ACT16_BYTES;          // [block 4][form 2: heads, remainders][N tile 2] fragments
constexpr int ACT6_BYTES = 4 * 2 * 2 * F6_BYTES;  // 32 KiB
constexpr int TILE_BYTES = 8192;               // voxel tile of the gather, per wave: in the halves of both regions that fc_0's
                                               // 128-wide K phases leave free (chunks 8..15, blocks 2..3)
constexpr int TILE_L0 = 3072;                  // levels 0 and 1 share a tile: 24 + 20 voxels (boxes of 4 x 4 pixel blocks: p99 27 / 18)
constexpr int SCR_A = ACT6_OFF + ACT6_BYTES;   // alpha_fc partial sums [64 samples][4 waves] floats
constexpr int SCR_C = SCR_A + 1024;            // rgb_fc partial sums [3][64][4] floats
// small fp32 parameters staged once per workgroup (LDS reads are counted on lgkmcnt: a global load at the head of a layer
// phase would drain the weight ring's vmcnt queue): offsets in floats
constexpr int PRM_OFF = SCR_C + 3072;
constexpr int P_B0 = 0, P_B1 = 256, P_B2 = 512, P_AW = 768, P_RW = 1024, P_AB = 1408, P_RB = 1412, P_LB = 1416, P_SIZE = 1544;
// per-sample ray record (20 floats): ox oy oz near | dx dy dz far | vx vy vz |d| | T r g b | depth acc - - (the compositing
// state).  It lives here and not in registers: with 256 registers per wave hipcc spills ~70 long-lived per-ray values to
// scratch, and every reload is a memory round trip behind an `s_waitcnt vmcnt` (tools/experiments/ms6_phase_times.py)
constexpr int RAY_OFF = PRM_OFF + P_SIZE * 4, RAY_FLOATS = 20;
constexpr int TV_OFF = RAY_OFF + 64 * RAY_FLOATS * 4, TV_MAX = 240;  // t_vals (if_clight_renderer.py:13) of marches with <= TV_MAX samples
static_assert(RAY_OFF % 16 == 0, "16-byte aligned records");
#ifdef MS6_ONEWG
constexpr int LDS_BYTES = TV_OFF + TV_MAX * 4 + 16384;  // experiment: one workgroup per CU
#else
constexpr int LDS_BYTES = TV_OFF + TV_MAX * 4;
static_assert(LDS_BYTES <= 81920, "two workgroups per CU");
#endif
__device__ __forceinline__ int tile_off(int wave) { return wave < 2 ? 8 * CH_BYTES + wave * TILE_BYTES : ACT6_OFF + 2 * 4 * F6_BYTES + (wave - 2) * TILE_BYTES; }

__device__ __forceinline__ f32x16 bias_tile_g(const float *bp, int t, int hi) {
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(bp + (t * 2 + hi) * 16);
    const f32x4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
    return f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
}

// ---------------------------------------------------------------- operands -> LDS
// 32 values of one half-block (block b, half kh) of sample column `slot` (0..31) of N tile n: fp16 heads into the four
// chunks of the block, the two bf6 forms + their E8M0 scales into the block's fragments
struct HalfBlock {
    f16x8 xh[4];
    i32x6 xl, xx;
    int eb;
};
template <bool RELU, class Get>
__device__ __forceinline__ HalfBlock convert_halfblock(Get get) {
    HalfBlock h;
    i32x6 xl[1], xx[1];
    int eb[1];
#ifdef MS6_ABL_NOCONV
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) h.xh[j][q] = (_Float16)0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) h.xh[q][0] = __builtin_bit_cast(f16x2, __float_as_uint(get(8 * q)))[0];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        xl[0][q] = __float_as_int(get(q + 8));
        xx[0][q] = __float_as_int(get(q + 16));
    }
    eb[0] = 127;
#else
    make_operands6<2, RELU>(get, h.xh, xl, xx, eb);
#endif
    h.xl = xl[0];
    h.xx = xx[0];
    h.eb = eb[0];
    return h;
}
__device__ __forceinline__ void store_halfblock(char *act, int b, int kh, int n, int slot, const HalfBlock &h) {
    const int ls = (kh * 32 + slot) * 16;
    char *p16 = act + (4 * b) * CH_BYTES + n * 1024 + ls;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f16x8 *>(p16 + j * CH_BYTES) = h.xh[j];
    char *p6 = act + ACT6_OFF + ((b * 2 + 0) * 2 + n) * F6_BYTES + ls;
    *reinterpret_cast<i32x4 *>(p6) = i32x4{h.xx[0], h.xx[1], h.xx[2], h.xx[3]};
    *reinterpret_cast<i32x4 *>(p6 + 1024) = i32x4{h.xx[4], h.xx[5], h.eb, 0};
    *reinterpret_cast<i32x4 *>(p6 + 2 * F6_BYTES) = i32x4{h.xl[0], h.xl[1], h.xl[2], h.xl[3]};
    *reinterpret_cast<i32x4 *>(p6 + 2 * F6_BYTES + 1024) = i32x4{h.xl[4], h.xl[5], h.eb - 11, 0};
}
template <bool RELU, class Get>
__device__ __forceinline__ void write_halfblock(char *act, int b, int kh, int n, int slot, Get get) {
    store_halfblock(act, b, kh, n, slot, convert_halfblock<RELU>(get));
}
// the same conversion cut into 16 value-pair slices + a finish, for work that rides in the shadow of another phase's MFMAs
struct Conv6 {
    u32x16 hv;
    f32x16 ra, rb;
    float m;
};
template <int I>
__device__ __forceinline__ void conv6_pair(Conv6 &c, float v0, float v1) {  // values 2 I, 2 I + 1 of the half-block
    if constexpr (I == 0) c.m = 0.f;
    const unsigned h = cvt_pk_f16(v0, v1);
    c.hv[I] = h;
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(c.m) : "v"(v0), "v"(v1));
    const float r0 = rem16<0>(v0, h), r1 = rem16<1>(v1, h);
    if constexpr (I < 8) {
        c.ra[2 * I] = r0;
        c.ra[2 * I + 1] = r1;
    } else {
        c.rb[2 * (I - 8)] = r0;
        c.rb[2 * (I - 8) + 1] = r1;
    }
}
__device__ __forceinline__ HalfBlock conv6_finish(const Conv6 &c) {
    HalfBlock h;
    h.eb = block_exponent(c.m);
    cvt_block6(c.hv, c.ra, c.rb, h.eb, h.xx, h.xl);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) h.xh[j] = __builtin_bit_cast(f16x8, u32x4{c.hv[4 * j], c.hv[4 * j + 1], c.hv[4 * j + 2], c.hv[4 * j + 3]});
    return h;
}

// sin and cos of 2 pi x 2^K for the positional encodings (embedder.py:26-36) without fp64: t = x / (2 pi) as an unevaluated
// sum th + tl of two floats (error ~2^-48 |t|), u = th 2^K exact, fract(u) exact, and the hardware sine takes revolutions.
// Against the fp64 reduction of nb_march_common.h: <= 2e-7 absolute (v_fract_f32 clamps just below 1 for tiny negative u).
struct Rev2 {
    float th, tl;
};
__device__ __forceinline__ Rev2 rev2(float x) {
    constexpr float C_HI = 0.15915494f, C_LO = (float)(NB_INV_2PI - (double)0.15915494f);
    Rev2 r;
    r.th = x * C_HI;
    r.tl = fmaf(x, C_HI, -r.th) + x * C_LO;
    return r;
}
template <int K>
__device__ __forceinline__ void sincos_rev2(const Rev2 &t, float &sn, float &cs) {
    constexpr float P2 = (float)(1 << K);
    const float u = t.th * P2;
    sn = __builtin_amdgcn_sinf(fmaf(t.tl, P2, __builtin_amdgcn_fractf(u)));
    cs = __builtin_amdgcn_sinf(fmaf(t.tl, P2, __builtin_amdgcn_fractf(u + 0.25f)));
}

// ---------------------------------------------------------------- one layer phase of this wave
struct WRing {
    i32x8 f[S_R / 2];  // piece p in half (p & 1) of f[(p % S_R) / 2]
};
// the wave's share of the stream through a buffer descriptor (4 SGPRs, wave-uniform): `buffer_load_dwordx4 v, v_off, s[rsrc],
// s_off offen` with the lane's 32-bit offset in a VGPR and the 1-KiB piece stride on the scalar unit — left to the compiler
// as plain pointers every piece costs a 64-bit VALU address (v_add_co / v_addc pairs and their register pairs)
typedef unsigned u32x4v __attribute__((__vector_size__(16)));
struct WSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff;  // lane * 16
};
__device__ __forceinline__ i32x4 load_piece(const WSrc &wl, int p) {
#ifdef MS6_ABL_NOW
    return i32x4{(int)wl.voff, p, 0, 0};
#endif
#ifdef MS6_ABL_NOW2  // no weight traffic at all: finite stand-in values (fp16 6e-5; six-bit scale word 127 = 2^0)
    int c0 = 0x04040404, c2 = 127;
    asm volatile("" : "+v"(c0), "+v"(c2));
    return i32x4{c0, c0, c2, 0};
#endif
#ifdef MS6_ABL_NOW3  // every load hits the same KiB: the L1 delivers, L2 is out of the picture
    p = 0;
#endif
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(wl.rsrc, wl.voff, (p % P_TOTAL) * 1024, 0);
    return i32x4{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}
template <int P>
__device__ __forceinline__ void ring_put(WRing &r, const i32x4 v) {
    i32x8 &d = r.f[(P % S_R) / 2];
    if (P & 1) {
        d[4] = v.x; d[5] = v.y; d[6] = v.z; d[7] = v.w;
    } else {
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
template <int P>
__device__ __forceinline__ i32x4 ring_get(const WRing &r) {
    const i32x8 &d = r.f[(P % S_R) / 2];
    return (P & 1) ? i32x4{d[4], d[5], d[6], d[7]} : i32x4{d[0], d[1], d[2], d[3]};
}

__device__ __forceinline__ f32x16 mfma16(const i32x4 a, const i32x4 b, const f32x16 c) {
#ifdef MS6_ABL_NOM
    return c;
#endif
#ifdef MS6_ABL_NOM2  // operands still loaded and consumed (one VALU instruction)
    f32x16 r = c;
    r[0] = __int_as_float(__float_as_int(r[0]) ^ (a.x & b.x & 1));
    return r;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// A: fp6 e2m3 (cbsz 2), 24 B of data + the lane's E8M0 scale in register 6; B: bf6 e3m2 (blgp 3), same layout
__device__ __forceinline__ f32x16 mfma6(const i32x8 a, const i32x8 b, const f32x16 c) {
#ifdef MS6_ABL_NOM
    return c;
#endif
#if defined(MS6_ABL_NOM2) || defined(MS6_ABL_NOM6)
    f32x16 r = c;
    r[0] = __int_as_float(__float_as_int(r[0]) ^ (a[0] & b[0] & a[6] & b[6] & a[5] & b[5] & 1));
    return r;
#endif
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 3, 0, a[6], 0, b[6]);
}

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// the first S_R pieces of the phase at P0: issued by the caller when the preceding phase did not prefetch them (a register-
// hungry VALU phase in between: a ring that is live across the gather gets spilled, and its reloads wait vmcnt(0))
template <int P0>
__device__ __forceinline__ void ring_prime(const WSrc &wl, WRing &ring) {
    sfor<0, S_R>([&](auto pc) { ring_put<P0 + decltype(pc)::value>(ring, load_piece(wl, P0 + decltype(pc)::value)); });
}

// acc[m][n] += W[tiles of this wave, K range of the phase] . X for the NB blocks at LDS blocks 0..NB-1
// AHEAD: the last S_R pieces' slots are refilled with the first pieces of the FOLLOWING phase (otherwise: ring_prime).
// A phase is a flat list of steps, six per block: the block's four K=16 chunks (MT x 2 fp16 MFMAs each), then its two cross
// terms (k = 0: W_h (fp6) x remainders, LDS form 1; k = 1: W_l x heads, form 0; MT x 2 scaled MFMAs each).  The B operands
// of step t + 1 are read from LDS before the MFMAs of step t are issued.
struct BOps {
    i32x4 m[2][2];  // [buffer][N tile]: main
    i32x8 c[2][2];  // cross: 6 registers of bf6 data, the scale byte in register 6
};
template <int T>
__device__ __forceinline__ void read_b(const char *b16, const char *b6, BOps &x) {
    constexpr int b = T / 6, j = T % 6, buf = T & 1;
#ifdef MS6_ABL_NOB  // B operands read for the first block of a phase only
    if constexpr (T >= 6) return;
#endif
    if constexpr (j < 4) {
        constexpr int c = 4 * b + j;
        x.m[buf][0] = *reinterpret_cast<const i32x4 *>(b16 + c * CH_BYTES);
        x.m[buf][1] = *reinterpret_cast<const i32x4 *>(b16 + c * CH_BYTES + 1024);
    } else {
        constexpr int form = 1 - (j - 4);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const char *q = b6 + ((b * 2 + form) * 2 + n) * F6_BYTES;
            const i32x4 lo = *reinterpret_cast<const i32x4 *>(q), hi4 = *reinterpret_cast<const i32x4 *>(q + 1024);
            x.c[buf][n] = i32x8{lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
        }
    }
}
struct NoFill {
    template <class T>
    __device__ __forceinline__ void operator()(T) const {}
};
// FILL: independent VALU work cut into 6 NB slices; slice t is issued right behind the MFMAs of step t (VALU instructions of
// the SAME wave execute under its MFMAs; another wave's do not: profiles/r03_ms6_coexec.md)
template <int P0, int MT, int NB, bool AHEAD, class Fill = NoFill>
__device__ __forceinline__ void layer_s(const WSrc &wl, const char *act, int lane, WRing &ring, f32x16 (&acc)[2][2], Fill fill = Fill()) {
    constexpr int PPB = 8 * MT;  // pieces per block
    constexpr int PEND = P0 + NB * PPB;
    constexpr int NT = 6 * NB;
    const char *b16 = act + lane * 16;
    const char *b6 = act + ACT6_OFF + lane * 16;
    BOps x;
    read_b<0>(b16, b6, x);
    sfor<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value, b = t / 6, j = t % 6, buf = t & 1;
        if constexpr (t + 1 < NT) read_b<t + 1>(b16, b6, x);
        if constexpr (j < 4) {
            sfor<0, MT>([&](auto mc) {
                constexpr int m = decltype(mc)::value, P = P0 + b * PPB + j * MT + m;
                const i32x4 a = ring_get<P>(ring);
                acc[m][0] = mfma16(a, x.m[buf][0], acc[m][0]);
                acc[m][1] = mfma16(a, x.m[buf][1], acc[m][1]);
                if constexpr (AHEAD || P + S_R < PEND) ring_put<P>(ring, load_piece(wl, P + S_R));
            });
        } else {
            constexpr int k = j - 4;
            // the operands as pinned 8-register tuples: a 6-of-8 use of two separately allocated 16-byte loads costs two
            // copies per operand
            asm volatile("" : "+v"(x.c[buf][0]), "+v"(x.c[buf][1]));
            sfor<0, MT>([&](auto mc) {
                constexpr int m = decltype(mc)::value, P = P0 + b * PPB + 4 * MT + (k * MT + m) * 2;
                static_assert(P % 2 == 0, "a six-bit fragment is one ring entry");
                i32x8 a = ring.f[(P % S_R) / 2];
                asm volatile("" : "+v"(a));
                acc[m][0] = mfma6(a, x.c[buf][0], acc[m][0]);
                acc[m][1] = mfma6(a, x.c[buf][1], acc[m][1]);
                if constexpr (AHEAD || P + S_R < PEND) {
                    ring_put<P>(ring, load_piece(wl, P + S_R));
                    ring_put<P + 1>(ring, load_piece(wl, P + 1 + S_R));
                }
            });
        }
        fill(tc);
        __builtin_amdgcn_sched_barrier(0);
    });
    // pin the end of the accumulator chains HERE: MFMAs are pure, and hipcc otherwise sinks the tail of a phase past the
    // barrier and the following gather / conversion down to the next reader of the tile, keeping the operands they read
    // alive (spilled) all the way
#pragma unroll
    for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(acc[m][0]), "+v"(acc[m][1]));
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

// relu'd accumulators of this wave (two tiles x two N tiles) -> K block `wave` of the next layer, in place: the lane's 16 +
// 16 values of an N tile are half-block (wave, hi) and go into the lane's own fragment slot.  The conversion runs BEFORE the
// barrier that retires the previous activations (it needs this wave's accumulators only), the stores behind it.
__device__ __forceinline__ void publish_s(char *act, int lane, int wave, f32x16 (&acc)[2][2]) {
    const int i = lane & 31, hi = lane >> 5;
    HalfBlock h[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = relu1(acc[m][n][r]);
        h[n] = convert_halfblock<false>([&](int q) { return q < 16 ? acc[0][n][q & 15] : acc[1][n][q & 15]; });
    }
    __syncthreads();  // every wave is done reading the previous activations
#pragma unroll
    for (int n = 0; n < 2; ++n) store_halfblock(act, wave, hi, n, i, h[n]);
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

// a * b + c on the full-rate 24-bit multiplier (v_mul_lo_u32 runs at a quarter of the rate; hipcc turns __mul24 of a sum back
// into it).  b: wave-uniform (SGPR), |a|, |b| < 2^23
__device__ __forceinline__ int mad24s(int a, int b_uniform, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}

// unnorm_clamped (nb_march_common.h) with the level's (float)(size - 1) and (float)size + 1 as scalar kernel arguments
__device__ __forceinline__ float unnorm_s(float gcoord, float fm1, float fp1) {
    const float i = __fmul_rn(__fdiv_rn(__fadd_rn(gcoord, 1.f), 2.f), fm1);
    return fminf(fmaxf(i, -2.f), fp1);
}

// The wave's index box of level L (wave-uniform: SGPRs) for the grid box `wb` of its 16 samples.
struct VoxBox {
    int xlo, ylo, zlo, xhi, yhi, zhi;
};
template <int L>
__device__ __forceinline__ VoxBox vox_box(const SceneDev &sc, const WaveBox &wb) {
    const int D = sc.dhw[L][0], H = sc.dhw[L][1], W = sc.dhw[L][2];
    VoxBox b;
    b.xlo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_s(wb.lo.gw, sc.fm1[L][2], sc.fp1[L][2])), 0), W - 1));
    b.ylo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_s(wb.lo.gh, sc.fm1[L][1], sc.fp1[L][1])), 0), H - 1));
    b.zlo = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_s(wb.lo.gd, sc.fm1[L][0], sc.fp1[L][0])), 0), D - 1));
    b.xhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_s(wb.hi.gw, sc.fm1[L][2], sc.fp1[L][2])) + 1, 0), W - 1));
    b.yhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_s(wb.hi.gh, sc.fm1[L][1], sc.fp1[L][1])) + 1, 0), H - 1));
    b.zhi = __builtin_amdgcn_readfirstlane(min(max((int)floorf(unnorm_s(wb.hi.gd, sc.fm1[L][0], sc.fp1[L][0])) + 1, 0), D - 1));
    return b;
}
__device__ __forceinline__ WaveBox wave_box16(const GridCoord &g) {
    WaveBox wb;
    wb.lo.gw = red_min16(g.gw);
    wb.lo.gh = red_min16(g.gh);
    wb.lo.gd = red_min16(g.gd);
    wb.hi.gw = red_max16(g.gw);
    wb.hi.gh = red_max16(g.gh);
    wb.hi.gd = red_max16(g.gd);
    return wb;
}

// Voxel tile of level L -> wave-private LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B per instruction, no
// registers, no ds_write): piece p = (voxel v of the box, 16-byte quad q of its channel vector) lands at tile + p * 16.
// Issued ONE PHASE AHEAD of the blend that reads it (the fetch latency, 1-2 k cycles from L2, was the largest single item of
// a depth step: profiles/r03_ms6_phases.md), from inline asm so that hipcc keeps counting the weight ring's vmcnt (a DMA it
// knows of makes it wait vmcnt(0) at every ring use); the blend waits vmcnt(0) itself, at a point where the ring is empty.
// `lds_tile` = LDS byte address of the tile (wave-uniform), LIMIT = its size in bytes.
template <int L, int LIMIT>
__device__ __forceinline__ void tile_dma(const SceneDev &sc, const VoxBox &b, int lane, unsigned lds_tile) {
    constexpr int C = lvl_c(L), PC = C / 4;
    constexpr int MAX_IT = LIMIT / 1024;
#ifdef MS6_ABL_NOGATHER
    return;
#endif
    const int H = sc.dhw[L][1], W = sc.dhw[L][2];
    const int nx = b.xhi - b.xlo + 1, ny = b.yhi - b.ylo + 1, nz = b.zhi - b.zlo + 1;
    const int pieces = nx * ny * nz * PC;
    // the previous tile's reads have returned (their values were consumed), but say so
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (pieces > LIMIT / 16) return;  // wave-uniform: the blend reads the corners from global memory
    const float rcp_xy = __builtin_amdgcn_rcpf((float)(nx * ny)), rcp_x = __builtin_amdgcn_rcpf((float)nx);  // v + 0.5 absorbs 1 ulp
    const float *vol = sc.vol[L];
    const int lin0 = (b.zlo * H + b.ylo) * W + b.xlo;  // scalar
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        if (it * 64 < pieces) {  // wave-uniform
            const int p = min(it * 64 + lane, pieces - 1);
            const int v = p / PC, q = p % PC;
            // 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 are full rate, v_mul_lo_u32 is not): every factor is < 2^12
            const int vz = (int)(((float)v + 0.5f) * rcp_xy);
            const int r = mad24s(vz, -(nx * ny), v);
            const int vy = (int)(((float)r + 0.5f) * rcp_x);
            const int vx = mad24s(vy, -nx, r);
            const unsigned lin = (unsigned)mad24s(vz, H * W, mad24s(vy, W, vx + lin0));  // < 2^24 voxels per level
            const unsigned voff = (lin * C + q * 4) * 4u;  // byte offset: the largest volume is 104 MB
            const unsigned dst = lds_tile + it * 1024;
            asm volatile(
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %0, %1"
                :
                : "v"(voff), "s"(vol), "s"(dst)
                : "memory", "m0");
        }
    }
}

// channels [part * C/4, (part+1) * C/4) of level L for this lane's sample from the tile fetched by tile_dma<L, LIMIT> (the
// caller has waited for it); same corner order, weights and zero padding as gather_level (nb_march_common.h)
template <int L, int LIMIT, typename Sink>
__device__ __forceinline__ void gather_parts(const SceneDev &sc, const GridCoord &g, const VoxBox &b, int part, const char *buf, Sink sink) {
    constexpr int C = lvl_c(L), QC = C / 4, PC = C / 4;
#ifdef MS6_ABL_NOGATHER
#pragma unroll
    for (int grp = 0; grp < QC / 8; ++grp) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = g.gw * (float)(e + 1 + grp) + g.gh;
        sink(grp, o8);
    }
    return;
#endif
    const int D = sc.dhw[L][0], H = sc.dhw[L][1], W = sc.dhw[L][2];
    const float ix = unnorm_s(g.gw, sc.fm1[L][2], sc.fp1[L][2]), iy = unnorm_s(g.gh, sc.fm1[L][1], sc.fp1[L][1]), iz = unnorm_s(g.gd, sc.fm1[L][0], sc.fp1[L][0]);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float wx[2] = {(fx + 1.f) - ix, ix - fx};
    const float wy[2] = {(fy + 1.f) - iy, iy - fy};
    const float wz[2] = {(fz + 1.f) - iz, iz - fz};
    const int nx = b.xhi - b.xlo + 1, ny = b.yhi - b.ylo + 1, nz = b.zhi - b.zlo + 1;
    const int pieces = nx * ny * nz * PC;
    float cw[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
        const bool inb = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D;
        cw[corner] = inb ? (wx[dx] * wy[dy]) * wz[dz] : 0.f;
    }
    if (pieces <= LIMIT / 16) {  // wave-uniform: the wave's voxel box is in its tile
        // byte offset of corner (dx, dy, dz) inside the tile = ox[dx] + oy[dy] + oz[dz]: per-axis offsets of the clamped
        // indices (a corner outside the box has weight 0 and may read any row), 24-bit multiplies
        int ox[2], oy[2], oz[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            ox[d] = (min(max(x0 + d, b.xlo), b.xhi) - b.xlo) * (C * 4) + part * (QC * 4);
            oy[d] = mad24s(min(max(y0 + d, b.ylo), b.yhi) - b.ylo, nx * (C * 4), 0);
            oz[d] = mad24s(min(max(z0 + d, b.zlo), b.zhi) - b.zlo, nx * ny * (C * 4), 0);
        }
        int co[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) co[corner] = ox[corner & 1] + oy[(corner >> 1) & 1] + oz[corner >> 2];
        // corner-major: all of the lane's quads of corner c + 1 are read while corner c is blended into QC independent
        // accumulators (quad-major, 8 corner reads then their 32 dependent FMAs, left every LDS round trip exposed:
        // ~50 cycles per ds_read_b128).  Same summation order per channel as the reference's corner order.
        constexpr int NQ = QC / 4, NB = NQ < 4 ? NQ : 4;  // quads per lane, and per pass (register budget: 12 NB registers)
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += NB) {
            f32x4 acc4[NB], r[2][NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                acc4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                r[0][k] = *reinterpret_cast<const f32x4 *>(buf + co[0] + (q0 + k) * 16);
            }
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                if (corner + 1 < 8) {
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        r[(corner + 1) & 1][k] = *reinterpret_cast<const f32x4 *>(buf + co[corner + 1] + (q0 + k) * 16);
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const f32x4 v = r[corner & 1][k];
                    acc4[k].x = fmaf(cw[corner], v.x, acc4[k].x);
                    acc4[k].y = fmaf(cw[corner], v.y, acc4[k].y);
                    acc4[k].z = fmaf(cw[corner], v.z, acc4[k].z);
                    acc4[k].w = fmaf(cw[corner], v.w, acc4[k].w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k + 1 < NB || k < 1; k += 2) {
                if constexpr (NB >= 2) {
                    const float o8[8] = {acc4[k].x, acc4[k].y, acc4[k].z, acc4[k].w, acc4[k + 1].x, acc4[k + 1].y, acc4[k + 1].z, acc4[k + 1].w};
                    sink((q0 + k) / 2, o8);
                }
            }
        }
    } else {  // rays far apart (small images, random rays): read the corners from global memory
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
        }
    }
}
// the tile a blend is about to read has landed (no weight-ring load is in flight at the points where this is called)
__device__ __forceinline__ void tile_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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

// Heads and compositing of ONE finished depth step for this lane's sample, in slices (measured: behind the MFMAs of the next
// step's first phase they cost MORE than exposed — exp / divide chains and LDS round trips do not hide): raw2outputs (nerf_net_utils.py:19-46) exactly as RayAccum::add, state in the LDS ray record.
struct CompState {
    float out[4], dist, w, sig_r, sig_g, sig_b;
    RayAccum ra;
};
template <int T>
__device__ __forceinline__ void composite_slice(CompState &c, const char *actz, const float *pk, int sample, int part, float z_step,
                                                float z_after, bool last, const MarchArgs &a, long long ray, int sidx, int S, bool valid,
                                                WeightStore4 &wstore) {
    const f32x4 *rec = reinterpret_cast<const f32x4 *>(actz + RAY_OFF) + sample * (RAY_FLOATS / 4);
    if constexpr (T == 0) {
        const f32x4 pa = *reinterpret_cast<const f32x4 *>(actz + SCR_A + sample * 16);
        c.out[3] = ((pa.x + pa.y) + (pa.z + pa.w)) + pk[P_AB];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const f32x4 pc = *reinterpret_cast<const f32x4 *>(actz + SCR_C + (ch * 64 + sample) * 16);
            c.out[ch] = ((pc.x + pc.y) + (pc.z + pc.w)) + pk[P_RB + ch];
        }
    } else if constexpr (T == 1) {
        const float d = last ? 1e10f : __fsub_rn(z_after, z_step);
        c.dist = __fmul_rn(d, rec[2].w);
        const f32x4 c0 = rec[3], c1 = rec[4];
        c.ra.T = c0.x; c.ra.cr = c0.y; c.ra.cg = c0.z; c.ra.cb = c0.w; c.ra.depth = c1.x; c.ra.accw = c1.y;
    } else if constexpr (T == 2) {
        const float sig = fmaxf(c.out[3], 0.f);
        const float alpha = 1.f - expf(-sig * c.dist);
        c.w = alpha * c.ra.T;
        c.ra.T = c.ra.T * (__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f));
    } else if constexpr (T == 3) {
        c.sig_r = 1.f / (1.f + expf(-c.out[0]));
    } else if constexpr (T == 4) {
        c.sig_g = 1.f / (1.f + expf(-c.out[1]));
    } else if constexpr (T == 5) {
        c.sig_b = 1.f / (1.f + expf(-c.out[2]));
    } else if constexpr (T == 6) {
        c.ra.cr = fmaf(c.w, c.sig_r, c.ra.cr);
        c.ra.cg = fmaf(c.w, c.sig_g, c.ra.cg);
        c.ra.cb = fmaf(c.w, c.sig_b, c.ra.cb);
        c.ra.depth = fmaf(c.w, z_step, c.ra.depth);
        c.ra.accw += c.w;
        if (part == 0) {
            f32x4 *recw = reinterpret_cast<f32x4 *>(const_cast<char *>(actz) + RAY_OFF) + sample * (RAY_FLOATS / 4);
            recw[3] = f32x4{c.ra.T, c.ra.cr, c.ra.cg, c.ra.cb};
            recw[4] = f32x4{c.ra.depth, c.ra.accw, 0.f, 0.f};
        }
    } else if constexpr (T == 7) {
        wstore.push(a, ray, sidx, S, part, valid, c.w);
#if !defined(MS6_TAP) && !defined(MS6_TIMING)
        if (valid && part == 0 && a.raw)
            *reinterpret_cast<f32x4 *>(a.raw + (ray * S + sidx) * 4) = f32x4{c.out[0], c.out[1], c.out[2], c.out[3]};
#endif
    }
}

// view_fc column of encoding slot `slot` (0..31) of axis a: [x, (sin, cos)(x 2^k) k<10, v, (sin, cos)(v 2^k) k<4, 0, 0]; -1 = zero pad
__host__ __device__ inline int pe_slot_col(int a, int slot) {
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

// ---------------------------------------------------------------- the kernel
// MS6_TIMING (experiment builds): wave 0 of the first 32 workgroups stamps the cycle counter at the phase boundaries of every
// depth step into the `raw` output as [workgroup][step][32] (tools/experiments/ms6_phase_times.py)
#ifdef MS6_TIMING
#define MS6_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (tbuf) {                                                                                          \
            const unsigned long long t__ = __builtin_readcyclecounter();                                     \
            if (lane == 0) tbuf[(i)] = (unsigned)t__;                                                        \
        }                                                                                                    \
    } while (0)
#else
#define MS6_STAMP(i) do { } while (0)
#endif
// stamps 26..31 are sub-stamps of ONE segment: -DMS6_TIMING=1 (or empty) the second gather, -DMS6_TIMING=2 the preparation of the
// next depth step behind the folded view layer
#if defined(MS6_TIMING) && (MS6_TIMING + 0) == 2
#define MS6_SUBA(i) do { } while (0)
#define MS6_SUBB(i) MS6_STAMP(i)
#else
#define MS6_SUBA(i) MS6_STAMP(i)
#define MS6_SUBB(i) do { } while (0)
#endif
// MS6_TAP (debug builds): workgroup 0 dumps, at depth step 0, every layer's accumulators as [layer][feature][sample]
// fp32 into a.raw instead of the raw output (fc_0, fc_1, fc_2 pre-activation: 3 x 256 x 64; folded view layer: 128 x 64) —
// tools/experiments/ms6_tap_check.py compares them with nb_decode_points' fp32 activation tap
// MODE 0: rays (nb_march).  MODE 1 / 2: explicit points (nb_decode_points, raw [n,4] / density [n,1]): a point with its view
// direction is a one-sample "ray" (origin = the point, direction = the view direction taken as given, z = 0) whose decoder
// output is stored instead of composited; MODE 2 stops behind alpha_fc.
template <int MODE>
__global__ __launch_bounds__(256, 2) void nb_march_ms6_kernel(MarchArgs a, const char *stream) {
    constexpr bool POINTS = MODE != 0, DENSITY_ONLY = MODE == 2;
    saturate_fp16_conversions();
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    char *act = lds;
    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5;
    const int os = lane & 15, part = lane >> 4;  // owner role
    const int sample = 16 * wave + os;            // 0..63 inside the workgroup
    const int grp = xcd_remap(blockIdx.x, a.n_wave_groups);
    long long ray = (long long)grp * 64 + sample;
    const long long n_units = POINTS ? a.n_pts : a.n_rays;
    const bool valid = ray < n_units;
    if (!valid) ray = n_units - 1;
    if (!POINTS && a.ray_order) ray = a.ray_order[ray];
    const int S = POINTS ? 1 : a.n_samples;
    float ox, oy, oz, dx, dy, dz, near, far, dn, vx, vy, vz;
    if constexpr (POINTS) {
        ox = a.wpts[ray * 3 + 0], oy = a.wpts[ray * 3 + 1], oz = a.wpts[ray * 3 + 2];
        dx = DENSITY_ONLY ? 0.f : a.viewdir[ray * 3 + 0], dy = DENSITY_ONLY ? 0.f : a.viewdir[ray * 3 + 1];
        dz = DENSITY_ONLY ? 1.f : a.viewdir[ray * 3 + 2];
        near = far = 0.f;
        dn = 1.f;
        vx = dx, vy = dy, vz = dz;  // latent_xyzc.py:113 embeds the direction it is handed
    } else {
        ox = a.ray_o[ray * 3 + 0], oy = a.ray_o[ray * 3 + 1], oz = a.ray_o[ray * 3 + 2];
        dx = a.ray_d[ray * 3 + 0], dy = a.ray_d[ray * 3 + 1], dz = a.ray_d[ray * 3 + 2];
        near = a.near[ray], far = a.far[ray];
        dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        vx = dx / dn, vy = dy / dn, vz = dz / dn;
    }
    if (part == 0) {
        f32x4 *rec = reinterpret_cast<f32x4 *>(lds + RAY_OFF) + sample * (RAY_FLOATS / 4);
        rec[0] = f32x4{ox, oy, oz, near};
        rec[1] = f32x4{dx, dy, dz, far};
        rec[2] = f32x4{vx, vy, vz, dn};
        rec[3] = f32x4{1.f, 0.f, 0.f, 0.f};
        rec[4] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float *tr = (!POINTS && a.t_rand) ? a.t_rand + ray * S : nullptr;
    // a global load here would sit in the depth loop behind an s_waitcnt vmcnt: the table is staged in LDS
    const bool tv_lds = !POINTS && S <= TV_MAX;
    auto tval = [&](int s) -> float { return tv_lds ? reinterpret_cast<const float *>(lds + TV_OFF)[s] : a.t_vals[s]; };
    auto z_at = [&](int s, float near, float far) -> float {
        if constexpr (POINTS) return 0.f;
        const float zc = z_lin(near, far, tval(s));
        if (!tr) return zc;
        const float lower = s == 0 ? zc : 0.5f * __fadd_rn(zc, z_lin(near, far, tval(s - 1)));
        const float upper = s == S - 1 ? zc : 0.5f * __fadd_rn(z_lin(near, far, tval(s + 1)), zc);
        return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), tr[s]));
    };

    {
        float *prm = reinterpret_cast<float *>(lds + PRM_OFF);
        for (int i = tid; i < P_SIZE; i += 256) {
            float v;
            if (i < P_B1) v = a.pk[F_OFF_B0 + i - P_B0];
            else if (i < P_B2) v = a.pk[F_OFF_B1 + i - P_B1];
            else if (i < P_AW) v = a.pk[F_OFF_B2 + i - P_B2];
            else if (i < P_RW) v = a.pk[F_OFF_AW + i - P_AW];
            else if (i < P_AB) v = a.pk[F_OFF_RW + i - P_RW];
            else if (i < P_RB) v = a.pk[F_OFF_AB + i - P_AB];
            else if (i < P_LB) v = a.pk[F_OFF_RB + i - P_RB];
            else v = DENSITY_ONLY ? 0.f : a.lb[256 + i - P_LB];  // bias of the folded view layer (nb_mlp_latent_bias, second block)
            prm[i] = v;
        }
        if (tv_lds)
            for (int i = tid; i < S; i += 256) reinterpret_cast<float *>(lds + TV_OFF)[i] = a.t_vals[i];
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(stream) + (size_t)wave * P_TOTAL * 1024, 0, P_TOTAL * 1024, 0x00020000);
    WRing ring;

    WeightStore4 wstore;
    float z_cur = z_at(0, near, far);
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned lds_tile = (unsigned)(unsigned long long)(lptr_t)(lds + tile_off(wave));  // wave-uniform LDS byte address
    // grid coordinates and box of the step about to be marched (carried from the point where its first tile is requested)
    GridCoord g = grid_coords(a.sc, __fadd_rn(ox, __fmul_rn(dx, z_cur)), __fadd_rn(oy, __fmul_rn(dy, z_cur)), __fadd_rn(oz, __fmul_rn(dz, z_cur)));
    WaveBox wb = wave_box16(g);
    tile_dma<3, TILE_BYTES>(a.sc, vox_box<3>(a.sc, wb), lane, lds_tile);
    for (int s = 0; s < S; ++s) {
        // loop-invariant address roots are laundered so that LICM does not hoist (and spill) hundreds of addresses
        int zero = 0, lane_i = lane;
        asm volatile("" : "+s"(zero), "+v"(lane_i));
        // (these shadow the prologue's: everything derived from the lane id is re-derived from the laundered copy)
        const int hi = lane_i >> 5, os = lane_i & 15, part = lane_i >> 4, sample = 16 * wave + os;
        const WSrc wl = {wrsrc, (unsigned)lane_i * 16u};
        char *actz = act + zero;
        char *tile = actz + tile_off(wave);
        const float *pk = reinterpret_cast<const float *>(actz + PRM_OFF);
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(actz + RAY_OFF) + sample * (RAY_FLOATS / 4);
        f32x16 acc[2][2];
        const int sn = sample >> 5, ss = sample & 31;  // N tile and column of this lane's sample
#ifdef MS6_TIMING
        unsigned *tbuf = (blockIdx.x < 32 && wave == 0 && a.raw) ? reinterpret_cast<unsigned *>(a.raw) + ((size_t)blockIdx.x * S + s) * 32 : nullptr;
#endif
        MS6_STAMP(0);

        // ---- fc_0 in three K phases of 128 = two blocks; part p fills half-block (p >> 1, p & 1).  The voxel tile of each
        // phase was requested (LDS-DMA) a phase earlier: level 3 behind the previous step's view layer, level 2 behind this
        // step's first blend, levels 0 and 1 (half a tile each) behind the second
        {
            float v[32];
            const VoxBox b3 = vox_box<3>(a.sc, wb);
            tile_wait();
            gather_parts<3, TILE_BYTES>(a.sc, g, b3, part, tile, [&](int grp8, const float (&o)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * grp8 + e] = o[e];
            });
            const VoxBox b2 = vox_box<2>(a.sc, wb);
            {
                HalfBlock hb = convert_halfblock<false>([&](int q) { return v[q]; });
                tile_dma<2, TILE_BYTES>(a.sc, b2, lane_i, lds_tile);
                store_halfblock(actz, part >> 1, part & 1, sn, ss, hb);
            }
            ring_prime<P_A>(wl, ring);
            init_bias<2>(pk + P_B0, 2 * wave, hi, acc);
            MS6_STAMP(1);
            __syncthreads();
            MS6_STAMP(2);
            layer_s<P_A, 2, 2, false>(wl, actz, lane_i, ring, acc);
            MS6_STAMP(3);
            __syncthreads();
            MS6_STAMP(4);
            tile_wait();
            MS6_SUBA(26);
            gather_parts<2, TILE_BYTES>(a.sc, g, b2, part, tile, [&](int grp8, const float (&o)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * grp8 + e] = o[e];
            });
#ifdef MS6_TIMING
            asm volatile("" : "+v"(v[0]), "+v"(v[8]), "+v"(v[16]), "+v"(v[24]), "+v"(v[31]));
#endif
            MS6_SUBA(27);
            const VoxBox b0 = vox_box<0>(a.sc, wb), b1 = vox_box<1>(a.sc, wb);
            MS6_SUBA(28);
            {
                HalfBlock hb = convert_halfblock<false>([&](int q) { return v[q]; });
#ifdef MS6_TIMING
                asm volatile("" : "+v"(hb.xl), "+v"(hb.xx));
#endif
                MS6_SUBA(29);
                tile_dma<0, TILE_L0>(a.sc, b0, lane_i, lds_tile);
                tile_dma<1, TILE_BYTES - TILE_L0>(a.sc, b1, lane_i, lds_tile + TILE_L0);
                MS6_SUBA(30);
                store_halfblock(actz, part >> 1, part & 1, sn, ss, hb);
            }
            MS6_SUBA(31);
            ring_prime<P_B>(wl, ring);
            MS6_STAMP(5);
            __syncthreads();
            MS6_STAMP(6);
            layer_s<P_B, 2, 2, false>(wl, actz, lane_i, ring, acc);
            MS6_STAMP(7);
            __syncthreads();
            MS6_STAMP(8);
            tile_wait();
            gather_parts<0, TILE_L0>(a.sc, g, b0, part, tile, [&](int grp8, const float (&o)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * grp8 + e] = o[e];
            });
            gather_parts<1, TILE_BYTES - TILE_L0>(a.sc, g, b1, part, tile + TILE_L0, [&](int grp8, const float (&o)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 + 8 * grp8 + e] = o[e];
            });
#pragma unroll
            for (int e = 24; e < 32; ++e) v[e] = 0.f;
            write_halfblock<false>(actz, part >> 1, part & 1, sn, ss, [&](int q) { return v[q]; });
            ring_prime<P_C>(wl, ring);
            MS6_STAMP(9);
            __syncthreads();
            MS6_STAMP(10);
            layer_s<P_C, 2, 2, true>(wl, actz, lane_i, ring, acc);
            MS6_STAMP(11);
        }
#ifdef MS6_TAP
#define MS6_DUMP(LAYER, MT_)                                                                                              \
    if (blockIdx.x == 0 && s == 0 && a.raw) {                                                                            \
        for (int m = 0; m < (MT_); ++m)                                                                                   \
            for (int n = 0; n < 2; ++n)                                                                                   \
                for (int r = 0; r < 16; ++r)                                                                              \
                    a.raw[((LAYER) * 256 + 32 * ((MT_) * wave + m) + tile_row(r, hi)) * 64 + n * 32 + (lane & 31)] = acc[m][n][r]; \
    }
#else
#define MS6_DUMP(LAYER, MT_)
#endif
        MS6_DUMP(0, 2)
        publish_s(actz, lane_i, wave, acc);
        MS6_STAMP(12);
        // ---- fc_1, fc_2
        init_bias<2>(pk + P_B1, 2 * wave, hi, acc);
        // The positional encodings of this step (lane (sample, axis a = part < 3): x_a, (sin, cos)(x_a 2^k) k < 10, v_a, (sin, cos)
        // (v_a 2^k) k < 4, two zeros; part 3: zeros) and their conversion into operands ride behind fc_1's MFMAs, one slice per
        // MFMA step; the finished half-block waits in registers until the view layer has released the activation buffers.
        Conv6 pec;
        {
            const f32x4 ro = rec[0], rd = rec[1], rv = rec[2];  // ox oy oz near | dx dy dz far | vx vy vz |d|
            const float keep = part < 3 ? 1.f : 0.f;
            const float xa = part == 0 ? __fadd_rn(ro.x, __fmul_rn(rd.x, z_cur))
                                       : (part == 1 ? __fadd_rn(ro.y, __fmul_rn(rd.y, z_cur)) : __fadd_rn(ro.z, __fmul_rn(rd.z, z_cur)));
            const float va = part == 0 ? rv.x : (part == 1 ? rv.y : rv.z);
            Rev2 tx, tv;
            float e[32];
            auto pe_fill = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t == 0) {
                    tx = rev2(xa);
                    tv = rev2(va);
                    e[0] = xa;
                    e[21] = va;
                    e[30] = 0.f;
                    e[31] = 0.f;
                } else if constexpr (t <= 10) {
#ifdef MS6_ABL_NOPE
                    e[2 * t - 1] = xa * (float)t;
                    e[2 * t] = xa + (float)t;
#else
                    sincos_rev2<t - 1>(tx, e[2 * t - 1], e[2 * t]);
#endif
                } else if constexpr (t <= 14) {
                    sincos_rev2<t - 11>(tv, e[2 * t], e[2 * t + 1]);
                } else if constexpr (t >= 16) {
                    constexpr int i = t - 16;
                    conv6_pair<2 * i>(pec, e[4 * i] * keep, e[4 * i + 1] * keep);
                    conv6_pair<2 * i + 1>(pec, e[4 * i + 2] * keep, e[4 * i + 3] * keep);
                }
            };
            if constexpr (DENSITY_ONLY) layer_s<P_L1, 2, 4, true>(wl, actz, lane_i, ring, acc);
            else layer_s<P_L1, 2, 4, true>(wl, actz, lane_i, ring, acc, pe_fill);
        }
        HalfBlock peh;
        if constexpr (!DENSITY_ONLY) peh = conv6_finish(pec);
        MS6_STAMP(13);
        MS6_DUMP(1, 2)
        publish_s(actz, lane_i, wave, acc);
        MS6_STAMP(14);
        init_bias<2>(pk + P_B2, 2 * wave, hi, acc);
        layer_s<P_L2, 2, 4, true>(wl, actz, lane_i, ring, acc);
        MS6_STAMP(15);
        MS6_DUMP(2, 2)
        if constexpr (DENSITY_ONLY) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] = relu1(acc[m][n][r]);
        } else {
            publish_s(actz, lane_i, wave, acc);  // acc now holds relu(h3)
        }
        MS6_STAMP(16);
        // ---- alpha_fc: partial dot product over this wave's 64 features, finished by the owner lanes
        {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float s4[4] = {0.f, 0.f, 0.f, 0.f};  // four independent chains
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x4 *aw = reinterpret_cast<const f32x4 *>(pk + P_AW + hi * 128 + 16 * (2 * wave + m));
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 w4 = aw[q4];
                        s4[0] = fmaf(w4.x, acc[m][n][4 * q4 + 0], s4[0]);
                        s4[1] = fmaf(w4.y, acc[m][n][4 * q4 + 1], s4[1]);
                        s4[2] = fmaf(w4.z, acc[m][n][4 * q4 + 2], s4[2]);
                        s4[3] = fmaf(w4.w, acc[m][n][4 * q4 + 3], s4[3]);
                    }
                }
                float sa = add_halves((s4[0] + s4[1]) + (s4[2] + s4[3]));
                if (hi == 0) reinterpret_cast<float *>(actz + SCR_A)[(n * 32 + (lane_i & 31)) * 4 + wave] = sa;
            }
        }
        float z_next = 0.f;
        if constexpr (!DENSITY_ONLY) {
        // ---- the colour head's linear part as ONE layer (feature_fc . latent_fc . view_fc folded at pack time, bias: second
        // block of nb_mlp_latent_bias): one tile per wave, K phase over fc_2's outputs, then over the encodings
        init_bias<1>(pk + P_LB, wave, hi, acc);
        MS6_STAMP(17);
        layer_s<P_VG, 1, 4, false>(wl, actz, lane_i, ring, acc);
        MS6_STAMP(18);
        __syncthreads();
        MS6_STAMP(19);
        // the upper halves of the activation buffers are free until the next step's fc_0 is published: request the next
        // step's level-3 tile now (in flight under the encodings, the last MFMA phase, the heads and the compositing)
        const f32x4 ro = rec[0], rd = rec[1];  // ox oy oz near | dx dy dz far
        z_next = (s + 1 < S) ? z_at(s + 1, ro.w, rd.w) : 0.f;
#if defined(MS6_TIMING) && (MS6_TIMING + 0) == 2
        asm volatile("" : "+v"(z_next));
#endif
        MS6_SUBB(26);
        if constexpr (!POINTS) {
            const float nx_ = __fadd_rn(ro.x, __fmul_rn(rd.x, z_next)), ny_ = __fadd_rn(ro.y, __fmul_rn(rd.y, z_next)),
                        nz_ = __fadd_rn(ro.z, __fmul_rn(rd.z, z_next));
            g = grid_coords(a.sc, nx_, ny_, nz_);
#if defined(MS6_TIMING) && (MS6_TIMING + 0) == 2
            asm volatile("" : "+v"(g.gw), "+v"(g.gh), "+v"(g.gd));
#endif
            MS6_SUBB(27);
            wb = wave_box16(g);
#if defined(MS6_TIMING) && (MS6_TIMING + 0) == 2
            asm volatile("" : "+v"(wb.lo.gw), "+v"(wb.hi.gd));
#endif
            MS6_SUBB(28);
            const VoxBox nb3 = vox_box<3>(a.sc, wb);
            MS6_SUBB(29);
            tile_dma<3, TILE_BYTES>(a.sc, nb3, lane_i, lds_tile);
        }
        MS6_SUBB(30);
        ring_prime<P_VP>(wl, ring);  // behind the DMA (vmcnt retires in order), ahead of the encodings that cover its latency
        MS6_SUBB(31);
        store_halfblock(actz, part >> 1, part & 1, sn, ss, peh);  // the encodings converted behind fc_1
        MS6_STAMP(20);
        __syncthreads();
        MS6_STAMP(21);
        layer_s<P_VP, 1, 2, false>(wl, actz, lane_i, ring, acc);
        MS6_STAMP(22);
        MS6_DUMP(3, 1)
        // ---- rgb_fc partial sums over this wave's 32 view features
        {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const f32x4 *rw = reinterpret_cast<const f32x4 *>(pk + P_RW + (ch * 2 + hi) * 64 + 16 * wave);
                    float c4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 w4 = rw[q4];
                        c4[0] = fmaf(w4.x, relu1(acc[0][n][4 * q4 + 0]), c4[0]);
                        c4[1] = fmaf(w4.y, relu1(acc[0][n][4 * q4 + 1]), c4[1]);
                        c4[2] = fmaf(w4.z, relu1(acc[0][n][4 * q4 + 2]), c4[2]);
                        c4[3] = fmaf(w4.w, relu1(acc[0][n][4 * q4 + 3]), c4[3]);
                    }
                    float sc = add_halves((c4[0] + c4[1]) + (c4[2] + c4[3]));
                    if (hi == 0) reinterpret_cast<float *>(actz + SCR_C)[(ch * 64 + n * 32 + (lane_i & 31)) * 4 + wave] = sc;
                }
        }
        }
        MS6_STAMP(23);
        __syncthreads();
        MS6_STAMP(24);
        // ---- owner lanes: finish the heads; composite (rays) or hand the decoder output over (points)
        {
            CompState cs;
            if constexpr (POINTS) {
                composite_slice<0>(cs, actz, pk, sample, part, z_cur, z_next, true, a, ray, s, S, valid, wstore);
                if (valid && part == 0) {
                    if constexpr (DENSITY_ONLY) a.raw_out[ray] = cs.out[3];
                    else *reinterpret_cast<f32x4 *>(a.raw_out + ray * 4) = f32x4{cs.out[0], cs.out[1], cs.out[2], cs.out[3]};
                }
            } else {
                sfor<0, 8>([&](auto tc) { composite_slice<decltype(tc)::value>(cs, actz, pk, sample, part, z_cur, z_next, s + 1 >= S, a, ray, s, S, valid, wstore); });
            }
        }
        MS6_STAMP(25);
        z_cur = z_next;
    }
    if (!POINTS && valid && part == 0) {
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(lds + RAY_OFF) + sample * (RAY_FLOATS / 4);
        const f32x4 c0 = rec[3], c1 = rec[4];
        RayAccum ra;
        ra.T = c0.x; ra.cr = c0.y; ra.cg = c0.z; ra.cb = c0.w; ra.depth = c1.x; ra.accw = c1.y;
        ra.store(a, ray);
    }
}

// ---------------------------------------------------------------- weight stream packing
// input column of the layer phase `ph` held by element e of half-block (b, kh); -1 = zero padding
__device__ __forceinline__ int phase_col(int ph, int b, int kh, int e) {
    const int p = 2 * b + kh;  // fc_0 / encodings: the owner lane's part
    switch (ph) {
        case 0: return 224 + 32 * p + e;
        case 1: return 96 + 32 * p + e;
        case 2: return e < 8 ? 8 * p + e : (e < 24 ? 32 + 16 * p + (e - 8) : -1);
        case 3: case 4: case 5: return col_hidden(32 * b + e, kh);
        default: return pe_slot_col(p, e);
    }
}
__device__ __forceinline__ float phase_weight(const nb_mlp_params &p, const float *f32_blob, int ph, int row, int b, int kh, int e) {
    const int col = phase_col(ph, b, kh, e);
    if (col < 0) return 0.f;
    if (ph < 3) return p.fc0_w[row * 352 + col];
    if (ph == 3) return p.fc1_w[row * 256 + col];
    if (ph == 4) return p.fc2_w[row * 256 + col];
    if (ph == 5) {
        // view_w[:, :256] . (latent_w[:, :256] . feature_w): the inner product comes from the fp32 section (formed in fp64
        // there, fragment order: invert col_hidden), the outer one is summed in fp64 here
        const int tt = col >> 5, rr = col & 31, hi2 = (rr >> 2) & 1, r2 = (rr & 3) + 4 * (rr >> 3), q2 = 16 * tt + r2;
        double s = 0.0;
        for (int m = 0; m < 256; ++m)
            s += (double)p.view_w[row * 346 + m] *
                 (double)f32_blob[F_OFF_L4 + (((m >> 5) * 32 + (q2 >> 2)) * 64 + (hi2 * 32 + (m & 31))) * 4 + (q2 & 3)];
        return (float)s;
    }
    return p.view_w[row * 346 + col];
}

// one thread per (wave, piece, lane): the lane's 16 bytes
__global__ void nb_pack_ms6_kernel(nb_mlp_params p, const float *__restrict__ f32_blob, unsigned *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * P_TOTAL * 64) return;
    const int lane = t & 63, piece = (t >> 6) % P_TOTAL, w = (t >> 6) / P_TOTAL;
    const int i = lane & 31, kg = lane >> 5;
    int ph = 0, p0 = 0;
    for (int q = 0; q < N_PH; ++q) {
        const int n = phase_pieces(PH_NB[q], PH_MT[q]);
        if (piece < p0 + n) {
            ph = q;
            break;
        }
        p0 += n;
    }
    const int mt = PH_MT[ph], ppb = 8 * mt;
    const int rel = piece - p0, b = rel / ppb, r = rel % ppb;
    unsigned w32[4] = {0u, 0u, 0u, 0u};
    if (r < 4 * mt) {  // A16 of chunk j, tile m
        const int j = r / mt, m = r % mt;
        const int row = (mt == 2 ? 64 * w + 32 * m : 32 * w) + i;
        for (int q = 0; q < 8; q += 2) {
            const f16x2 hp = {(_Float16)phase_weight(p, f32_blob, ph, row, b, kg, 8 * j + q),
                              (_Float16)phase_weight(p, f32_blob, ph, row, b, kg, 8 * j + q + 1)};
            w32[q / 2] = __builtin_bit_cast(unsigned, hp);
        }
    } else {  // six-bit fragment: k = 0 W_h (multiplies the interleaved remainder operand), k = 1 W_l (natural order)
        const int r6 = r - 4 * mt, k = r6 / (2 * mt), m = (r6 / 2) % mt, half = r6 & 1;
        const int row = (mt == 2 ? 64 * w + 32 * m : 32 * w) + i;
        float wv[32], amax = 0.f;
        for (int e = 0; e < 32; ++e) {
            const int n = k ? e : 16 * (e & 1) + (e >> 1);
            const float wt = phase_weight(p, f32_blob, ph, row, b, kg, n);
            const float h = (float)(_Float16)wt;
            wv[e] = k ? wt - h : h;
            amax = fmaxf(amax, fabsf(wv[e]));
        }
        int ex = 0;
        if (amax > 0.f && amax < 3.0e38f) {
            ex = ilogbf(amax / 7.5f);
            if (ldexpf(7.5f, ex) < amax) ++ex;  // the smallest power of two with max / 2^ex <= 7.5
            ex = min(max(ex, -120), 120);
        }
        unsigned w8[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        for (int e = 0; e < 32; ++e) {
            const unsigned code = fp6_e2m3_bits(ldexpf(wv[e], -ex));
            const int bit = 6 * e;
            w8[bit >> 5] |= code << (bit & 31);
            if ((bit & 31) > 26) w8[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
        }
        w8[6] = (unsigned)(127 + ex);
        for (int q = 0; q < 4; ++q) w32[q] = w8[4 * half + q];
    }
    unsigned *dst = out + ((size_t)(w * P_TOTAL + piece) * 64 + lane) * 4;
    for (int q = 0; q < 4; ++q) dst[q] = w32[q];
}

}  // namespace

namespace nbm {

long long ms6_stream_floats() { return (long long)4 * P_TOTAL * 1024 / 4; }

int pack_ms6_stream(const nb_mlp_params *p, float *packed, long long stream_off, hipStream_t st) {
    const long long n = (long long)4 * P_TOTAL * 64;
    hipLaunchKernelGGL(nb_pack_ms6_kernel, dim3(nb_ceil_div(n, 256)), dim3(256), 0, st, *p, packed,
                       reinterpret_cast<unsigned *>(packed + stream_off));
    NB_CHECK_LAUNCH("nb_pack_ms6_kernel");
    return NB_OK;
}

int launch_march_ms6(MarchArgs a, long long stream_off, hipStream_t st) {
    a.n_wave_groups = (int)nb_ceil_div(a.n_rays, 64);
    hipLaunchKernelGGL(nb_march_ms6_kernel<0>, dim3(a.n_wave_groups), dim3(256), 0, st, a,
                       reinterpret_cast<const char *>(a.pk + stream_off));
    NB_CHECK_LAUNCH("nb_march_ms6_kernel");
    return NB_OK;
}

// nb_decode_points on the same kernel: every point a one-sample ray whose decoder output is stored instead of composited
int launch_points_ms6(MarchArgs a, int density_only, long long stream_off, hipStream_t st) {
    a.n_wave_groups = (int)nb_ceil_div(a.n_pts, 64);
    const char *stream = reinterpret_cast<const char *>(a.pk + stream_off);
    if (density_only) hipLaunchKernelGGL(nb_march_ms6_kernel<2>, dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    else hipLaunchKernelGGL(nb_march_ms6_kernel<1>, dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    NB_CHECK_LAUNCH("nb_march_ms6_kernel (points)");
    return NB_OK;
}

}  // namespace nbm
