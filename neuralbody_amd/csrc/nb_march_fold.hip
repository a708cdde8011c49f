// nb_march_fold.hip — the fused march / point decoder with fc_0 FOLDED INTO THE VOLUME (NB_PREC_F16F6), gfx950.
//
// Trilinear interpolation (latent_xyzc.py:62-72) and fc_0 (:99) are linear with nothing between them:
//     fc_0 . interp(V) = interp(fc_0 . V)  =  sum over the voxels v a sample touches of  wt(v, sample) . U[v, :]
// with U = fc_0 . V stored per ACTIVE voxel by nb_fold_build (nb_fold.hip: 256 fp16 heads + 256 fp16 remainders per row).  For the
// 64 samples a workgroup marches per depth step the touched voxels of the four pyramid levels are a few small boxes (8 x 8 pixel
// tiles: 59 voxels on average, 110 at p99), so the first layer becomes ONE small matrix product on the matrix pipe
//     H1_pre [256 features x 64 samples]  =  U^T [256 x K voxels] . Wt [K x 64]
// with K = the workgroup's voxel list (padded to 16), U_h.Wt_h + U_h.Wt_l + U_l.Wt_h on v_mfma_f32_32x32x16_f16 (fp16 head +
// fp16 remainder of both operands: products exact, fp32 accumulation, the dropped term is 2^-22): fp32-level accuracy.
// Against a per-sample gather + fc_0 (round 3's kernel) this removes the 2816 gather FMAs and 352 operand conversions per sample,
// 96 of 272 weight pieces per wave and depth step, and two thirds of fc_0's MFMAs (12 per 16 voxels and wave instead of 144 per
// step).  Everything behind fc_0 is the "M-split" f16f6 organisation:
//
//   * a workgroup (4 waves) marches 64 rays; wave w owns a QUARTER OF EVERY LAYER'S OUTPUT FEATURES for all 64 samples; a weight
//     fragment feeds 2 MFMAs and comes straight from L2 into a register ring;
//   * activations live in LDS as ready-made B operands (fp16 heads by K=16 chunk, the two bf6 forms with their E8M0 scale by
//     K=64 block), rewritten in place after every layer; <= 256 registers and 80 KiB of LDS: two workgroups per CU.
//
// The folded first layer, per depth step (the preparation of step s + 1 rides inside step s):
//   boxes   owner lane (sample, level = lane >> 4) -> clamped floor / floor + 1 of its level coordinate, 16-lane min / max,
//           one box per (wave, level) in LDS; every lane then merges the four waves' boxes of its level: the workgroup's box,
//           its voxel count n_L, the level's offset k0_L in the K list (readlane), K = sum n_L                    [behind fc_1]
//   table   lane j of level L -> voxel j of the box -> index grid -> byte offset of its U row (inactive / padding: the zero row)
//           into a K-entry LDS table                                                            [loads before, stores behind fc_2]
//   U       every wave fetches ITS 64 features of the K rows by LDS-DMA (global_load_lds_dwordx4, 16-voxel chunks of 4 KiB:
//           heads | remainders) into a wave-private region, in an image whose 16-lane [4 voxel][16 feature] blocks
//           ds_read_b64_tr_b16 transposes into K-major A fragments, conflict-free                  [behind the last MFMA phase]
//   Wt      owner lanes zero their samples' B-fragment slots and scatter the 8 corner weights of their level (head and
//           remainder, ATen's corner order and zero padding: a corner outside the volume writes nothing)             [same place]
//   MFMA    per 16-voxel chunk and wave: 2 x 2 tiles x 3 products.
// A list longer than 128 voxels is marched in passes of 128 over the same boxes (up to 1024 voxels: neighbouring but not dense
// points, wide pixel footprints); beyond that (rays far apart: small images, random rays) in sample groups of 16 or, failing
// that, one sample at a time through the very same steps (K = 32 for a single sample).
//
// Per-sample work (ray set-up, positional encoding, compositing) is done by "owner" lanes: wave w, lane l owns sample
// 16 w + (l & 15) and, of that sample, pyramid level / axis `part` = l >> 4.
#include "nb_f6_ops.h"
#include "nb_march_hooks.h"

using namespace nbm;

namespace {

// ---------------------------------------------------------------- weight stream (per wave), in 1-KiB pieces
// phase = NB K blocks x MT output tiles of this wave; per block: for each of its 4 chunks, for each tile: A16 (1 piece: the fp16
// heads, the main product); then its two CROSS TERMS in execution order, for each tile one fragment: k = 0 W_h (multiplies the
// remainder operand), k = 1 W_l (multiplies the head operand).  A cross fragment is fp6 e2m3 — 24 bytes of data + its E8M0 scale
// in the lane's seventh dword, 2 pieces — or, for the terms in NB_FP4_TERMS (bit k), fp4 e2m1 — 16 bytes of data, 1 piece, its
// scale one byte of the block's SCALE DWORD (byte k MT + m; one 256-byte load per block from behind the pieces).  The cross
// terms are 2^-11 of the main term, so their weight operand needs few bits: fp4 takes the stream from 176 to 132 pieces per wave
// and depth step — the stream (L2 -> L1, 64 B/clk/CU) is what the MFMA phases wait for (profiles/r06_march_premises.log) — for
// 2^-13 instead of 2^-15 relative error per term (profiles/r06_precision_fp4.md).
#ifndef NB_FP4_TERMS
#define NB_FP4_TERMS 3
#endif
constexpr bool FP4_K[2] = {(NB_FP4_TERMS & 1) != 0, (NB_FP4_TERMS & 2) != 0};
constexpr bool ANY_FP4 = FP4_K[0] || FP4_K[1];
// six-bit fragments are two pieces = one (even-aligned) ring entry: when only W_h is four-bit its fragment goes second
constexpr int TERM_ORDER[2] = {(FP4_K[0] && !FP4_K[1]) ? 1 : 0, (FP4_K[0] && !FP4_K[1]) ? 0 : 1};
__host__ __device__ constexpr int term_pieces(int k) { return FP4_K[k] ? 1 : 2; }
// pieces per block (rounded to even: the next block's six-bit fragments stay even-aligned; an odd count leaves one hole)
__host__ __device__ constexpr int block_pieces(int mt) { return (4 * mt + mt * term_pieces(0) + mt * term_pieces(1) + 1) & ~1; }
// first piece (inside its block) of the fragment of the cross term executed `slot`-th (0 / 1), tile m
__host__ __device__ constexpr int cross_piece(int mt, int slot, int m) {
    return 4 * mt + (slot == 0 ? 0 : mt * term_pieces(TERM_ORDER[0])) + m * term_pieces(TERM_ORDER[slot]);
}
constexpr int S_R = 8;  // register ring depth in pieces
__host__ __device__ constexpr int phase_pieces(int nb, int mt) { return nb * block_pieces(mt); }
constexpr int N_PH = 4;
// fc_1, fc_2, the folded colour head over fc_2's outputs (one tile per wave), view_fc over the positional encodings
constexpr int PH_NB[N_PH] = {4, 4, 4, 2};
constexpr int PH_MT[N_PH] = {2, 2, 1, 1};
__host__ __device__ constexpr int phase_p0(int ph) {
    int p = 0;
    for (int i = 0; i < ph; ++i) p += phase_pieces(PH_NB[i], PH_MT[i]);
    return p;
}
__host__ __device__ constexpr int phase_s0(int ph) {  // first scale dword of the phase (one per block)
    int n = 0;
    for (int i = 0; i < ph; ++i) n += PH_NB[i];
    return n;
}
constexpr int P_L1 = phase_p0(0), P_L2 = phase_p0(1), P_VG = phase_p0(2), P_VP = phase_p0(3), P_TOTAL = phase_p0(4);
constexpr int N_SCALE = phase_s0(4);                        // 14 scale dwords (256 bytes each) behind the pieces
constexpr int WAVE_STREAM = P_TOTAL * 1024 + ((N_SCALE * 256 + 1023) & ~1023);  // bytes of one wave's share of the stream
static_assert(P_TOTAL == (NB_FP4_TERMS == 3 ? 132 : (NB_FP4_TERMS == 0 ? 176 : 160)), "pieces per wave per depth step");
static_assert(S_R % 2 == 0 && P_L1 % 2 == 0 && P_L2 % 2 == 0 && P_VG % 2 == 0 && P_VP % 2 == 0, "six-bit fragments on even pieces");
__host__ __device__ constexpr int phase_of_p0(int p0) { return p0 == P_L1 ? 0 : (p0 == P_L2 ? 1 : (p0 == P_VG ? 2 : 3)); }

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
constexpr int ACT_BYTES = ACT6_OFF + ACT6_BYTES;  // the folded first layer uses all of it: Wt chunks, then the waves' U regions
constexpr int WT_CHUNK = 4096;                 // Wt of 16 voxels: [form: heads, remainders][N tile] B fragments
constexpr int U_CHUNK = 4096;                  // one wave's 64 features of 16 voxels: [form][M tile] x 1 KiB transposable image
constexpr int K_CAP = 128;                     // voxels per pass (8 chunks)
constexpr int KM_CAP = 1024;                   // longest voxel list marched in several passes (tier 3)
constexpr int SCR_A = ACT_BYTES;               // alpha_fc partial sums [64 samples][4 waves] floats
constexpr int SCR_C = SCR_A + 1024;            // rgb_fc partial sums [3][64][4] floats
constexpr int WBOX_OFF = SCR_C;                // ... and, between two uses of those, the (wave, level) boxes: 16 x 8 ints
// small fp32 parameters staged once per workgroup (LDS reads are counted on lgkmcnt: a global load at the head of a layer
// phase would drain the weight ring's vmcnt queue): offsets in floats
constexpr int PRM_OFF = SCR_C + 3072;
constexpr int P_B0 = 0, P_B1 = 256, P_B2 = 512, P_AW = 768, P_RW = 1024, P_AB = 1408, P_RB = 1412, P_LB = 1416, P_SIZE = 1544;
// per-sample ray record (20 floats): ox oy oz near | dx dy dz far | vx vy vz |d| | T r g b | depth acc - - (the compositing
// state).  It lives here and not in registers: with 256 registers per wave hipcc spills long-lived per-ray values to scratch
constexpr int RAY_OFF = PRM_OFF + P_SIZE * 4, RAY_FLOATS = 20;
static_assert(RAY_OFF % 16 == 0, "16-byte aligned records");
// the folded first layer's bookkeeping
constexpr int LC_OFF = RAY_OFF + 64 * RAY_FLOATS * 4;  // per level: D H W fmx | fmy fmz fpx fpy | fpz grid_lo grid_hi row_base
constexpr int LVL_OFF = LC_OFF + 4 * 48;               // per level, of the step about to be marched: xlo ylo zlo nx | nxy k0 n -
constexpr int HDR_OFF = LVL_OFF + 4 * 32;              // K | - | - | -
constexpr int TBL_OFF = HDR_OFF + 16;                  // K_CAP byte offsets of U rows
constexpr int DUMMY_OFF = TBL_OFF + K_CAP * 4;         // where the weights of corners outside the volume go
constexpr int LDS_BYTES = DUMMY_OFF + 16;
static_assert(LDS_BYTES <= 81920, "two workgroups per CU");
static_assert(LC_OFF % 16 == 0 && LVL_OFF % 16 == 0 && HDR_OFF % 16 == 0, "vector reads");

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
    make_operands6<2, RELU>(get, h.xh, xl, xx, eb);
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
    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(wl.rsrc, wl.voff, (p % P_TOTAL) * 1024, 0);
    return i32x4{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
}
// the E8M0 scales of one block's four-bit fragments: one dword per lane (byte k MT + m)
__device__ __forceinline__ int load_scales(const WSrc &wl, int sidx) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(wl.rsrc, wl.voff >> 2, P_TOTAL * 1024 + sidx * 256, 0);
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
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// A: fp6 e2m3 (cbsz 2), 24 B of data + the lane's E8M0 scale in register 6; B: bf6 e3m2 (blgp 3), same layout
__device__ __forceinline__ f32x16 mfma6(const i32x8 a, const i32x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 3, 0, a[6], 0, b[6]);
}

// A: fp4 e2m1 (cbsz 4), 16 bytes of data; its E8M0 scale = byte OPSEL of `scales`; B as above
template <int OPSEL>
__device__ __forceinline__ f32x16 mfma4(const i32x4 a, int scales, const i32x8 b, const f32x16 c) {
    const i32x8 a8 = __builtin_shufflevector(a, a, 0, 1, 2, 3, -1, -1, -1, -1);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b, c, 4, 3, OPSEL, scales, 0, b[6]);
}

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// the first S_R pieces of the phase at P0: issued by the caller when the preceding phase did not prefetch them
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
    if constexpr (j < 4) {
        constexpr int c = 4 * b + j;
        x.m[buf][0] = *reinterpret_cast<const i32x4 *>(b16 + c * CH_BYTES);
        x.m[buf][1] = *reinterpret_cast<const i32x4 *>(b16 + c * CH_BYTES + 1024);
    } else {
        constexpr int form = 1 - TERM_ORDER[j - 4];  // W_h multiplies the remainders (form 1), W_l the heads (form 0)
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
    constexpr int PPB = block_pieces(MT);  // pieces per block
    constexpr int PEND = P0 + NB * PPB;
    constexpr int NT = 6 * NB;
    constexpr int S0 = phase_s0(phase_of_p0(P0));
    const char *b16 = act + lane * 16;
    const char *b6 = act + ACT6_OFF + lane * 16;
    BOps x;
    int sc[2] = {0, 0};  // scale dwords of the blocks' four-bit fragments: block b in sc[b & 1], loaded one block ahead
    if constexpr (ANY_FP4) sc[0] = load_scales(wl, S0);
    read_b<0>(b16, b6, x);
    sfor<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value, b = t / 6, j = t % 6, buf = t & 1;
        if constexpr (t + 1 < NT) read_b<t + 1>(b16, b6, x);
        if constexpr (ANY_FP4 && j == 0 && b + 1 < NB) sc[(b + 1) & 1] = load_scales(wl, S0 + b + 1);
        if constexpr (j < 4) {
            sfor<0, MT>([&](auto mc) {
                constexpr int m = decltype(mc)::value, P = P0 + b * PPB + j * MT + m;
                const i32x4 a = ring_get<P>(ring);
                acc[m][0] = mfma16(a, x.m[buf][0], acc[m][0]);
                acc[m][1] = mfma16(a, x.m[buf][1], acc[m][1]);
                if constexpr (AHEAD || P + S_R < PEND) ring_put<P>(ring, load_piece(wl, P + S_R));
            });
        } else {
            constexpr int slot = j - 4, k = TERM_ORDER[slot];
            // the operands as pinned 8-register tuples: a 6-of-8 use of two separately allocated 16-byte loads costs two
            // copies per operand
            asm volatile("" : "+v"(x.c[buf][0]), "+v"(x.c[buf][1]));
            sfor<0, MT>([&](auto mc) {
                constexpr int m = decltype(mc)::value, P = P0 + b * PPB + cross_piece(MT, slot, m);
                if constexpr (FP4_K[k]) {
                    const i32x4 a = ring_get<P>(ring);
                    acc[m][0] = mfma4<k * MT + m>(a, sc[b & 1], x.c[buf][0], acc[m][0]);
                    acc[m][1] = mfma4<k * MT + m>(a, sc[b & 1], x.c[buf][1], acc[m][1]);
                    if constexpr (AHEAD || P + S_R < PEND) ring_put<P>(ring, load_piece(wl, P + S_R));
                } else {
                    static_assert(P % 2 == 0, "a six-bit fragment is one ring entry");
                    i32x8 a = ring.f[(P % S_R) / 2];
                    asm volatile("" : "+v"(a));
                    acc[m][0] = mfma6(a, x.c[buf][0], acc[m][0]);
                    acc[m][1] = mfma6(a, x.c[buf][1], acc[m][1]);
                    if constexpr (AHEAD || P + S_R < PEND) {
                        ring_put<P>(ring, load_piece(wl, P + S_R));
                        ring_put<P + 1>(ring, load_piece(wl, P + 1 + S_R));
                    }
                }
            });
        }
        fill(tc);
        __builtin_amdgcn_sched_barrier(0);
    });
    // pin the end of the accumulator chains HERE: MFMAs are pure, and hipcc otherwise sinks the tail of a phase past the
    // barrier and the following conversion down to the next reader of the tile, keeping the operands they read alive
#pragma unroll
    for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(acc[m][0]), "+v"(acc[m][1]));
}

template <int MT>
__device__ __forceinline__ void init_bias(const float *bp, int tile0, int hi, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        // the second N tile's copy straight from LDS as well: 4 reads instead of 16 v_mov
        acc[m][0] = bias_tile_g(bp, tile0 + m, hi);
        asm volatile("" ::: "memory");
        acc[m][1] = bias_tile_g(bp, tile0 + m, hi);
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

// ---------------------------------------------------------------- the folded first layer
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v *lds_s4p;

// unnorm_clamped (nb_march_common.h) with the level's (float)(size - 1) and (float)size + 1 given
__device__ __forceinline__ float unnorm_s(float gcoord, float fm1, float fp1) {
    const float i = __fmul_rn(__fdiv_rn(__fadd_rn(gcoord, 1.f), 2.f), fm1);
    return fminf(fmaxf(i, -2.f), fp1);
}
// a * b + c on the full-rate 24-bit multiplier; |a|, |b| < 2^23
__device__ __forceinline__ int mad24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// min / max over the 16 lanes of a DPP row (= the 16 samples of one (wave, level)): row rotations, plain VALU instructions —
// the ds_swizzle butterflies they replace are LDS-crossbar round trips behind an lgkmcnt wait each
template <int N>
__device__ __forceinline__ int row_ror(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x120 | N, 0xf, 0xf, false);
}
__device__ __forceinline__ int red_min16i(int v) {
    v = min(v, row_ror<8>(v));
    v = min(v, row_ror<4>(v));
    v = min(v, row_ror<2>(v));
    v = min(v, row_ror<1>(v));
    return v;
}
__device__ __forceinline__ int red_max16i(int v) {
    v = max(v, row_ror<8>(v));
    v = max(v, row_ror<4>(v));
    v = max(v, row_ror<2>(v));
    v = max(v, row_ror<1>(v));
    return v;
}

// the lane's pyramid level (part): sizes, unnormalisation constants, index grid, first U row
struct Lvl {
    int D, H, W;
    float fmx, fmy, fmz, fpx, fpy, fpz;
    const int *grid;
    int rbase;
};
__device__ __forceinline__ Lvl load_lvl(const char *actz, int part) {
    const i32x4 *p = reinterpret_cast<const i32x4 *>(actz + LC_OFF + part * 48);
    const i32x4 a = p[0], b = p[1], c = p[2];
    Lvl l;
    l.D = a.x; l.H = a.y; l.W = a.z;
    l.fmx = __int_as_float(a.w); l.fmy = __int_as_float(b.x); l.fmz = __int_as_float(b.y);
    l.fpx = __int_as_float(b.z); l.fpy = __int_as_float(b.w); l.fpz = __int_as_float(c.x);
    l.grid = reinterpret_cast<const int *>(((unsigned long long)(unsigned)c.z << 32) | (unsigned long long)(unsigned)c.y);
    l.rbase = c.w;
    return l;
}
struct LvlIdx {
    float ix, iy, iz;
    int x0, y0, z0;
};
__device__ __forceinline__ LvlIdx level_index(const Lvl &lv, const GridCoord &g) {
    LvlIdx q;
    q.ix = unnorm_s(g.gw, lv.fmx, lv.fpx);
    q.iy = unnorm_s(g.gh, lv.fmy, lv.fpy);
    q.iz = unnorm_s(g.gd, lv.fmz, lv.fpz);
    q.x0 = (int)floorf(q.ix);
    q.y0 = (int)floorf(q.iy);
    q.z0 = (int)floorf(q.iz);
    return q;
}

// boxes: this lane's sample (if `take`) -> clamped [floor, floor + 1] of its level, merged over the 16 lanes of (wave, level);
// lanes os == 0 of the waves with `write` store the box at slot (wslot, level)
constexpr int BOX_BIG = 1 << 24;
__device__ __forceinline__ void prep_boxes(char *actz, const Lvl &lv, const GridCoord &g, bool take, int wslot, bool write, int os, int part) {
    const LvlIdx q = level_index(lv, g);
    int xlo = take ? min(max(q.x0, 0), lv.W - 1) : BOX_BIG, xhi = take ? min(max(q.x0 + 1, 0), lv.W - 1) : -1;
    int ylo = take ? min(max(q.y0, 0), lv.H - 1) : BOX_BIG, yhi = take ? min(max(q.y0 + 1, 0), lv.H - 1) : -1;
    int zlo = take ? min(max(q.z0, 0), lv.D - 1) : BOX_BIG, zhi = take ? min(max(q.z0 + 1, 0), lv.D - 1) : -1;
    xlo = red_min16i(xlo); ylo = red_min16i(ylo); zlo = red_min16i(zlo);
    xhi = red_max16i(xhi); yhi = red_max16i(yhi); zhi = red_max16i(zhi);
    const int any = __builtin_amdgcn_ballot_w64(take) != 0ull;  // does the wave decode any sample at all
    if (write && os == 0) {
        i32x4 *d = reinterpret_cast<i32x4 *>(actz + WBOX_OFF + (wslot * 4 + part) * 32);
        d[0] = i32x4{xlo, ylo, zlo, xhi};
        d[1] = i32x4{yhi, zhi, any, 0};
    }
}

// the merged box of the lane's level over the first `nw` wave slots, its place in the K list, and (uniform) K
struct Prep {
    int xlo, ylo, zlo, nx, nxy, n, k0;
    int K;     // uniform
    int tier;  // uniform: 0 = one pass over all 64 samples, 3 = the same list in passes of K_CAP voxels, 1 = groups of 16 samples
               // (one wave's), 2 = single samples
    int any;   // uniform: at least one sample takes part
};
__device__ __forceinline__ int box_count(int xlo, int ylo, int zlo, int xhi, int yhi, int zhi) {
    return (xhi < xlo || yhi < ylo || zhi < zlo) ? 0 : (xhi - xlo + 1) * (yhi - ylo + 1) * (zhi - zlo + 1);
}
template <int NW>
__device__ __forceinline__ Prep prep_wg(const char *actz, int part) {
    int xlo = BOX_BIG, ylo = BOX_BIG, zlo = BOX_BIG, xhi = -1, yhi = -1, zhi = -1;
    int n16[NW], any = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const i32x4 *s = reinterpret_cast<const i32x4 *>(actz + WBOX_OFF + (w * 4 + part) * 32);
        const i32x4 a = s[0], b = s[1];
        any |= b.z;
        n16[w] = box_count(a.x, a.y, a.z, a.w, b.x, b.y);
        xlo = min(xlo, a.x); ylo = min(ylo, a.y); zlo = min(zlo, a.z);
        xhi = max(xhi, a.w); yhi = max(yhi, b.x); zhi = max(zhi, b.y);
    }
    Prep p;
    p.xlo = xlo; p.ylo = ylo; p.zlo = zlo;
    p.nx = xhi - xlo + 1;
    p.nxy = p.nx * (yhi - ylo + 1);
    p.n = box_count(xlo, ylo, zlo, xhi, yhi, zhi);
    const int n0 = __builtin_amdgcn_readlane(p.n, 0), n1 = __builtin_amdgcn_readlane(p.n, 16), n2 = __builtin_amdgcn_readlane(p.n, 32),
              n3 = __builtin_amdgcn_readlane(p.n, 48);
    p.K = n0 + n1 + n2 + n3;
    p.k0 = part == 0 ? 0 : (part == 1 ? n0 : (part == 2 ? n0 + n1 : n0 + n1 + n2));
    p.tier = 0;
    p.any = any;
    if constexpr (NW > 1) {
        if (p.K > K_CAP && p.K <= KM_CAP) {  // uniform
            p.tier = 3;
        } else if (p.K > K_CAP) {
            int worst = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                worst = max(worst, __builtin_amdgcn_readlane(n16[w], 0) + __builtin_amdgcn_readlane(n16[w], 16) +
                                       __builtin_amdgcn_readlane(n16[w], 32) + __builtin_amdgcn_readlane(n16[w], 48));
            p.tier = worst <= K_CAP ? 1 : 2;
        }
    }
    return p;
}

// table: lane (wave, os) of level L looks up voxels j = 16 wave + os and j + 64 of its level's box (n_L <= K <= K_CAP = 128)
struct TblLoad {
    int rid[2], idx[2];
};
__device__ __forceinline__ TblLoad tbl_issue(const Prep &p, const Lvl &lv, int wave, int os) {
    TblLoad t;
    const float rcp_xy = __builtin_amdgcn_rcpf((float)max(p.nxy, 1)), rcp_x = __builtin_amdgcn_rcpf((float)max(p.nx, 1));  // v + 0.5 absorbs 1 ulp
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = 64 * q + 16 * wave + os;
        t.rid[q] = -1;
        t.idx[q] = -1;
        if (j < p.n) {
            const int vz = (int)(((float)j + 0.5f) * rcp_xy);
            const int r = mad24(vz, -p.nxy, j);
            const int vy = (int)(((float)r + 0.5f) * rcp_x);
            const int vx = mad24(vy, -p.nx, r);
            const int lin = mad24(mad24(p.zlo + vz, lv.H, p.ylo + vy), lv.W, p.xlo + vx);  // < 2^24 voxels per level
            t.rid[q] = lv.grid[lin];
            t.idx[q] = p.k0 + j;
        }
    }
    return t;
}
__device__ __forceinline__ void tbl_store(char *actz, const TblLoad &t, const Lvl &lv, unsigned zero_off, int K, int tid) {
    unsigned *tbl = reinterpret_cast<unsigned *>(actz + TBL_OFF);
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (t.idx[q] >= 0) tbl[t.idx[q]] = t.rid[q] < 0 ? zero_off : (unsigned)(lv.rbase + t.rid[q]) << 10;
    if (tid < 16 && K + tid < ((K + 15) & ~15)) tbl[K + tid] = zero_off;  // the padding of the last chunk
}
// table of ONE PASS of a long voxel list (tier 3): thread t < 128 looks up K-list position k_lo + t — its level from the
// levels' offsets, then the voxel of that level's box
__device__ __forceinline__ void tbl_pass(char *actz, unsigned zero_off, int k_lo, int K, int tid) {
    if (tid >= K_CAP) return;
    unsigned *tbl = reinterpret_cast<unsigned *>(actz + TBL_OFF);
    const int k = k_lo + tid;
    unsigned e = zero_off;
    if (k < K) {
        const int k1 = *reinterpret_cast<const int *>(actz + LVL_OFF + 1 * 32 + 20), k2 = *reinterpret_cast<const int *>(actz + LVL_OFF + 2 * 32 + 20),
                  k3 = *reinterpret_cast<const int *>(actz + LVL_OFF + 3 * 32 + 20);
        const int L = (k >= k1) + (k >= k2) + (k >= k3);
        const i32x4 *lp = reinterpret_cast<const i32x4 *>(actz + LVL_OFF + L * 32);
        const i32x4 b0 = lp[0], b1 = lp[1];  // xlo ylo zlo nx | nxy k0 n -
        const Lvl lv = load_lvl(actz, L);
        const int j = k - b1.y;
        const float rcp_xy = __builtin_amdgcn_rcpf((float)max(b1.x, 1)), rcp_x = __builtin_amdgcn_rcpf((float)max(b0.w, 1));
        const int vz = (int)(((float)j + 0.5f) * rcp_xy);
        const int r = mad24(vz, -b1.x, j);
        const int vy = (int)(((float)r + 0.5f) * rcp_x);
        const int vx = mad24(vy, -b0.w, r);
        const int lin = mad24(mad24(b0.z + vz, lv.H, b0.y + vy), lv.W, b0.x + vx);
        const int rid = lv.grid[lin];
        if (rid >= 0) e = (unsigned)(lv.rbase + rid) << 10;
    }
    tbl[tid] = e;
}
__device__ __forceinline__ void lvl_store(char *actz, const Prep &p, int tid, int os, int part) {
    if (tid < 64 && os == 0) {
        i32x4 *d = reinterpret_cast<i32x4 *>(actz + LVL_OFF + part * 32);
        d[0] = i32x4{p.xlo, p.ylo, p.zlo, p.nx};
        d[1] = i32x4{p.nxy, p.k0, p.n, 0};
    }
    if (tid == 0) *reinterpret_cast<int *>(actz + HDR_OFF) = p.K;
}

// the K list of the step about to be marched, per wave: chunks, the wave's U region (R chunk slots behind the Wt chunks)
struct UCfg {
    int nch, R;
    unsigned ring;  // LDS byte address of the wave's region
    int ring_off;   // the same as an offset into the workgroup's LDS
};
__device__ __forceinline__ UCfg ucfg_k(int K, unsigned lds_base, int wave);
__device__ __forceinline__ UCfg ucfg(const char *actz, unsigned lds_base, int wave) {
    return ucfg_k(__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(actz + HDR_OFF)), lds_base, wave);
}
__device__ __forceinline__ UCfg ucfg_k(int K, unsigned lds_base, int wave) {
    UCfg u;
    u.nch = (K + 15) >> 4;
    u.R = u.nch <= 3 ? u.nch : (u.nch == 4 ? 3 : 2);  // (64 KiB - 4 KiB nch) / (4 waves x 4 KiB)
    u.ring_off = WT_CHUNK * u.nch + wave * u.R * U_CHUNK;
    u.ring = lds_base + (unsigned)u.ring_off;
    return u;
}
// One chunk (16 voxels) of this wave's 64 features, heads and remainders: 4 LDS-DMA instructions of 64 lanes x 16 bytes.  The
// image per (form, M tile) KiB is [4 voxel groups][2 feature halves][4 voxels][16 features]: lane s fetches 8 features
// (16 bytes) of voxel 4 (s >> 4) + ((s >> 1) & 3), feature half (s >> 3) & 1, octet s & 1 — a [4 voxel][16 feature] block is 128
// contiguous bytes, the two blocks a 32-lane half of ds_read_b64_tr_b16 reads are 256 contiguous bytes (no bank conflict).
// From inline asm so that hipcc keeps counting the weight ring's vmcnt (a DMA it knows of makes it wait vmcnt(0) at every
// ring use); the consumer waits vmcnt(0) itself.
__device__ __forceinline__ void dma_chunk(const MarchArgs &a, const char *actz, int lane, int wave, int c, unsigned dst) {
    const int v = ((lane >> 4) << 2) | ((lane >> 1) & 3);
    const unsigned e = *reinterpret_cast<const unsigned *>(actz + TBL_OFF + (16 * c + v) * 4);
    const unsigned voff0 = e + (unsigned)(64 * wave + 16 * ((lane >> 3) & 1) + 8 * (lane & 1)) * 2u;
    const char *base = a.fold.urows;
#pragma unroll
    for (int form = 0; form < 2; ++form)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned voff = voff0 + form * 512 + m * 64;
            const unsigned d = dst + form * 2048 + m * 1024;
            asm volatile(
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %0, %1"
                :
                : "v"(voff), "s"(base), "s"(d)
                : "memory", "m0");
        }
}
__device__ __forceinline__ void dma_initial(const MarchArgs &a, const char *actz, int lane, int wave, const UCfg &u) {
    for (int c = 0; c < u.R; ++c) dma_chunk(a, actz, lane, wave, c, u.ring + c * U_CHUNK);  // R <= nch
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Wt: zero the B-fragment slots of this wave's 16 samples (64 slots of 16 bytes per chunk: one per lane), then every owner lane
// that `take`s part scatters the 8 corner weights of its level.  Element k of sample column n (N tile nt): chunk k >> 4, fragment
// lane 32 ((k >> 3) & 1) + n, element k & 7; heads at form 0, remainders 2 KiB behind.  LDS executes a wave's instructions in
// order, so the scatter lands on the zeros without a barrier in between.
// (k_lo: first K-list position of the pass being built; corners outside [k_lo, k_lo + 16 nch) write nothing)
__device__ __forceinline__ void wt_build(char *actz, const Lvl &lv, const GridCoord &g, bool take, int wave, int os, int part, int nch,
                                         int k_lo = 0) {
    const int nt = wave >> 1, n = 16 * (wave & 1) + os;
    {
        char *z = actz + (part >> 1) * 2048 + nt * 1024 + ((part & 1) * 32 + n) * 16;
        for (int c = 0; c < nch; ++c) *reinterpret_cast<i32x4 *>(z + c * WT_CHUNK) = i32x4{0, 0, 0, 0};
    }
    if (take) {
        const i32x4 *lp = reinterpret_cast<const i32x4 *>(actz + LVL_OFF + part * 32);
        const i32x4 b0 = lp[0], b1 = lp[1];  // xlo ylo zlo nx | nxy k0 n -
        const LvlIdx q = level_index(lv, g);
        const float fx = (float)q.x0, fy = (float)q.y0, fz = (float)q.z0;
        const float wx[2] = {(fx + 1.f) - q.ix, q.ix - fx};
        const float wy[2] = {(fy + 1.f) - q.iy, q.iy - fy};
        const float wz[2] = {(fz + 1.f) - q.iz, q.iz - fz};
        const int kbase = b1.y - k_lo + mad24(q.z0 - b0.z, b1.x, mad24(q.y0 - b0.y, b0.w, q.x0 - b0.x));
        const int sbase = nt * 1024 + n * 16;
        float cw[8];
        int ad[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
            const int xx = q.x0 + dx, yy = q.y0 + dy, zz = q.z0 + dz;
            const bool inb = (unsigned)xx < (unsigned)lv.W && (unsigned)yy < (unsigned)lv.H && (unsigned)zz < (unsigned)lv.D;
            cw[corner] = (wx[dx] * wy[dy]) * wz[dz];
            const int k = kbase + dx + dy * b0.w + dz * b1.x;
            const int off = ((k << 8) & ~0xfff) | ((k << 6) & 0x200) | ((k << 1) & 0xe);
            ad[corner] = (inb && (unsigned)k < (unsigned)(16 * nch)) ? off + sbase : DUMMY_OFF;
        }
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            const unsigned h = cvt_pk_f16(cw[2 * cp], cw[2 * cp + 1]);
            const unsigned l = cvt_pk_f16(rem16<0>(cw[2 * cp], h), rem16<1>(cw[2 * cp + 1], h));
            *reinterpret_cast<unsigned short *>(actz + ad[2 * cp]) = (unsigned short)(h & 0xffffu);
            *reinterpret_cast<unsigned short *>(actz + ad[2 * cp + 1]) = (unsigned short)(h >> 16);
            *reinterpret_cast<unsigned short *>(actz + ad[2 * cp] + 2048) = (unsigned short)(l & 0xffffu);
            *reinterpret_cast<unsigned short *>(actz + ad[2 * cp + 1] + 2048) = (unsigned short)(l >> 16);
        }
    }
}

// K-major A fragment (32 features x 16 voxels) of one (form, M tile) KiB of the image above: two transposing reads
__device__ __forceinline__ f16x8 tr_frag(const char *p) {
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p));
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4p)(p + 256));
    typedef short s8v __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(f16x8, s8v{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}
// acc += U^T . Wt over the step's K list: per chunk 4 A fragments (2 tiles x heads, remainders) and 4 B fragments feed 12 MFMAs.
// Software-pipelined: the fragments of chunk c + 1 are read from LDS (and, beyond the R resident chunks, its DMA awaited with
// a COUNTED vmcnt: 4 instructions per younger chunk) before the MFMAs of chunk c are issued; the slot of chunk c is refilled
// with chunk c + R as soon as its reads have returned.
struct FoldFrags {
    f16x8 ah[2], al[2];
    i32x4 bh[2], bl[2];
};
__device__ __forceinline__ void fold_wait(int younger) {  // the chunk with `younger` chunks issued behind it has landed
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void fold_read(FoldFrags &f, const char *ub, const char *wtc) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f.ah[m] = tr_frag(ub + m * 1024);
        f.al[m] = tr_frag(ub + 2048 + m * 1024);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        f.bh[n] = *reinterpret_cast<const i32x4 *>(wtc + n * 1024);
        f.bl[n] = *reinterpret_cast<const i32x4 *>(wtc + 2048 + n * 1024);
    }
}
__device__ __forceinline__ void fold_mma(const FoldFrags &f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)  // U_h.Wt_h, U_h.Wt_l, U_l.Wt_h: four independent accumulators per product
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p == 2 ? f.al[m] : f.ah[m], __builtin_bit_cast(f16x8, p == 1 ? f.bl[n] : f.bh[n]),
                                                                   acc[m][n], 0, 0, 0);
}
__device__ __forceinline__ void fold_mfma(const MarchArgs &a, char *actz, int lane, int wave, const UCfg &u, f32x16 (&acc)[2][2],
                                          unsigned *tbuf = nullptr) {
    if (u.nch == 0) return;  // uniform
    FOLD_SUB(17);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int tro = (g4 >> 1) * 512 + (g4 & 1) * 128 + (l15 >> 2) * 32 + (l15 & 3) * 8;
    const char *wt = actz + lane * 16;
    const char *ring = actz + u.ring_off + tro;
    int issued = u.R - 1;  // highest chunk whose DMA has been issued
    FoldFrags f0, f1;
    fold_wait(issued);
    FOLD_SUB(18);
    fold_read(f0, ring, wt);
    NB_HOOK_FOLD_FIRST_READ(f0);
    FOLD_SUB(19);
    auto step = [&](FoldFrags &cur, FoldFrags &nxt, int c, int slot) {
        if (c + u.R < u.nch) {  // uniform: refill this chunk's slot; its reads must have returned
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur.ah[0]), "+v"(cur.ah[1]), "+v"(cur.al[0]), "+v"(cur.al[1])::"memory");
            dma_chunk(a, actz, lane, wave, c + u.R, u.ring + slot * U_CHUNK);
            issued = c + u.R;
        }
        if (c + 1 < u.nch) {
            const int ns = slot + 1 == u.R ? 0 : slot + 1;
            fold_wait(issued - (c + 1));
            fold_read(nxt, ring + ns * U_CHUNK, wt + (c + 1) * WT_CHUNK);
        }
        fold_mma(cur, acc);
    };
    int slot = 0;
    for (int c = 0; c < u.nch; c += 2) {
        step(f0, f1, c, slot);
        slot = slot + 1 == u.R ? 0 : slot + 1;
        if (c + 1 < u.nch) {
            step(f1, f0, c + 1, slot);
            slot = slot + 1 == u.R ? 0 : slot + 1;
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(acc[m][0]), "+v"(acc[m][1]));
    FOLD_SUB(20);
    if (tbuf && lane == 0) tbuf[21] = (unsigned)u.nch;
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

// Heads and compositing of ONE finished depth step for this lane's sample: raw2outputs (nerf_net_utils.py:19-46) exactly as
// RayAccum::add, state in the LDS ray record, the work split over the sample's four owner lanes — every lane forms sigma,
// alpha and the weight; part 0 carries transmittance, depth and opacity, parts 1..3 one colour channel each (its logit, its
// sigmoid, its accumulator); `ins` false (culled sample): raw = 0 (if_clight_renderer_mmsk.py:54-59).
__device__ __forceinline__ float head_sum(const char *actz, int off, float bias) {
    const f32x4 p = *reinterpret_cast<const f32x4 *>(actz + off);
    return ((p.x + p.y) + (p.z + p.w)) + bias;
}
// The LAST sample's interval is 1e10 (nerf_net_utils.py:28): its alpha is a step function of the sign of its density, and this
// arithmetic's density error (~3e-4 absolute on the bench scene) can put a ray on the other side of the step.  A ray whose last
// density is within NB_ILL_SIGMA of zero while it still carries transmittance is therefore LISTED — its state in front of the
// sample, the sample's colour logits — and nb_march_fixup_kernel recomputes that one density at fp32 level and composites the
// sample again (same stream, no host involved).  ~1e-5 of the rays; a culled sample has density 0 by definition and is not listed.
// Called in front of the last step's composite_step (the ray record still holds the state in front of the sample), once per ray.
__device__ __noinline__ void list_ill_ray(const char *actz, const float *pk, float *ill, int cap, int sample, int part, float z_step, int ray) {
    const float *recf = reinterpret_cast<const float *>(actz + RAY_OFF) + sample * RAY_FLOATS;
    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(recf + 12);  // T r g b
    const float sigma_raw = head_sum(actz, SCR_A + sample * 16, pk[P_AB]);
    const bool hit = ray >= 0 && fabsf(sigma_raw) < NB_ILL_SIGMA && c0.x > NB_ILL_T_MIN;
    if (__builtin_amdgcn_ballot_w64(hit) == 0ull) return;
    int slot = -1;
    if (hit && part == 0) slot = atomicAdd(reinterpret_cast<int *>(ill), 1);
    slot = __builtin_amdgcn_ds_bpermute((sample & 15) << 2, slot);  // from the sample's part-0 lane
    if (!hit || slot >= cap) return;
    float *irec = ill + NB_ILL_HEADER_FLOATS + (long long)slot * NB_ILL_RECORD_FLOATS;
    if (part == 0) {
        *reinterpret_cast<f32x4 *>(irec) = f32x4{__int_as_float(ray), c0.x, recf[16], recf[17]};
        *reinterpret_cast<f32x4 *>(irec + 4) = f32x4{z_step, __fmul_rn(1e10f, recf[11]), sigma_raw, 0.f};
    } else {
        const int ch = part - 1;
        irec[8 + ch] = part == 1 ? c0.y : (part == 2 ? c0.z : c0.w);
        irec[12 + ch] = head_sum(actz, SCR_C + (ch * 64 + sample) * 16, pk[P_RB + ch]);
    }
}

__device__ __forceinline__ void composite_step(char *actz, const float *pk, int sample, int part, float z_step, float z_after, bool last,
                                               const MarchArgs &a, long long ray, int sidx, int S, bool valid, WeightStore4 &wstore, bool ins) {
    float *recf = reinterpret_cast<float *>(actz + RAY_OFF) + sample * RAY_FLOATS;
    const f32x4 c0 = *reinterpret_cast<const f32x4 *>(recf + 12);  // T r g b
    const float sigma_raw = ins ? head_sum(actz, SCR_A + sample * 16, pk[P_AB]) : 0.f;
    const float d = last ? 1e10f : __fsub_rn(z_after, z_step);
    const float dist = __fmul_rn(d, recf[11]);
    const float sig = fmaxf(sigma_raw, 0.f);
    const float alpha = 1.f - expf(-sig * dist);
    const float w = alpha * c0.x;
    if (part == 0) {
        const float depth = recf[16], accw = recf[17];
        recf[12] = c0.x * (__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f));
        recf[16] = fmaf(w, z_step, depth);
        recf[17] = accw + w;
    } else {
        const int ch = part - 1;
        const float o = ins ? head_sum(actz, SCR_C + (ch * 64 + sample) * 16, pk[P_RB + ch]) : 0.f;
        const float col = part == 1 ? c0.y : (part == 2 ? c0.z : c0.w);
        recf[12 + part] = fmaf(w, 1.f / (1.f + expf(-o)), col);
    }
    wstore.push(a, ray, sidx, S, part, valid, w);
    if (NB_HOOK_RAW_IS_OUTPUT && a.raw && valid && part == 0) {  // uniform in a.raw
        float o3[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o3[ch] = ins ? head_sum(actz, SCR_C + (ch * 64 + sample) * 16, pk[P_RB + ch]) : 0.f;
        *reinterpret_cast<f32x4 *>(a.raw + (ray * S + sidx) * 4) = f32x4{o3[0], o3[1], o3[2], sigma_raw};
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
// FOLD_STAMP / FOLD_SUB / FOLD_DUMP: empty in the product build (nb_march_hooks.h); experiment builds pre-include
// tools/experiments/fold_instrument.h (cycle stamps at the phase boundaries, per-layer accumulator taps)
// MODE 0: rays (nb_march).  MODE 1 / 2: explicit points (nb_decode_points, raw [n,4] / density [n,1]): a point with its view
// direction is a one-sample "ray" (origin = the point, direction = the view direction taken as given, z = 0) whose decoder
// output is stored instead of composited; MODE 2 stops behind alpha_fc.
// CULL (rays only): nb_cull — a sample that projects outside any silhouette is not decoded: it joins no voxel box, scatters no
// weights, and its raw output is 0 (if_clight_renderer_mmsk.py:54-59); a depth step without a single inside sample skips its
// layers altogether (workgroup-uniform).
template <int MODE, bool CULL>
__global__ __launch_bounds__(256, 2) void nb_march_fold_kernel(MarchArgs a, const char *stream) {
    constexpr bool POINTS = MODE != 0, DENSITY_ONLY = MODE == 2;
    static_assert(!(CULL && POINTS), "sample culling is a property of the ray march");
    saturate_fp16_conversions();
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    char *act = lds;
    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = xcd_remap(blockIdx.x, a.n_wave_groups);
    long long ray = (long long)grp * 64 + 16 * wave + (lane & 15);
    const long long n_units = POINTS ? a.n_pts : a.n_rays;
    bool valid = ray < n_units;
    if (!valid) ray = n_units - 1;
    if (!POINTS && a.ray_order) {
        if (a.ray_order[(long long)grp * 64] == NB_SLOT_DEAD) return;  // an empty group of slots (workgroup-uniform, before any barrier)
        const int v = a.ray_order[(long long)grp * 64 + 16 * wave + (lane & 15)];
        valid = v >= 0;
        ray = valid ? v : -(long long)v - 1;  // a padding slot marches that ray's data and stores nothing
    }
    const int S = POINTS ? 1 : a.n_samples;
    {
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
        if ((lane >> 4) == 0) {
            f32x4 *rec = reinterpret_cast<f32x4 *>(lds + RAY_OFF) + (16 * wave + (lane & 15)) * (RAY_FLOATS / 4);
            rec[0] = f32x4{ox, oy, oz, near};
            rec[1] = f32x4{dx, dy, dz, far};
            rec[2] = f32x4{vx, vy, vz, dn};
            rec[3] = f32x4{1.f, 0.f, 0.f, 0.f};
            rec[4] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float *tr = (!POINTS && a.t_rand) ? a.t_rand + ray * S : nullptr;
    // t_vals (if_clight_renderer.py:13) through the scalar cache: wave-uniform index, no vector-memory counter involved
    auto tval = [&](int s) -> float { return ((cfloat_ptr)a.t_vals)[s]; };
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
        if (tid < 4) {
            int *lc = reinterpret_cast<int *>(lds + LC_OFF + tid * 48);
            const unsigned long long gp = (unsigned long long)a.fold.grid[tid];
            lc[0] = a.sc.dhw[tid][0]; lc[1] = a.sc.dhw[tid][1]; lc[2] = a.sc.dhw[tid][2];
            lc[3] = __float_as_int(a.sc.fm1[tid][2]); lc[4] = __float_as_int(a.sc.fm1[tid][1]); lc[5] = __float_as_int(a.sc.fm1[tid][0]);
            lc[6] = __float_as_int(a.sc.fp1[tid][2]); lc[7] = __float_as_int(a.sc.fp1[tid][1]); lc[8] = __float_as_int(a.sc.fp1[tid][0]);
            lc[9] = (int)(unsigned)(gp & 0xffffffffull); lc[10] = (int)(unsigned)(gp >> 32);
            lc[11] = a.fold.row_base[tid];
        }
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(stream) + (size_t)wave * WAVE_STREAM, 0, WAVE_STREAM, 0x00020000);
    WRing ring;
    WeightStore4 wstore;
    typedef void __attribute__((address_space(3))) *lptr_t;
    const unsigned lds_base = (unsigned)(unsigned long long)(lptr_t)(lds);  // wave-uniform LDS byte address

    float z_cur;
    GridCoord g;      // grid coordinates of the step about to be marched (this lane's sample)
    bool ins = true;  // ... and whether it survives the silhouette test
    int tier, active;  // uniform: how the step's voxel list is marched (Prep::tier); any sample to decode at all
    // The voxel list of step `sn` in one go (step 0, and behind a step whose layers were skipped): sample position, boxes,
    // the workgroup's K list and its table.  Ends with everything written but not yet published (the caller's barrier).
    auto prep_sequential = [&](int sn, float &z_out, int lane_c) {
        const int os = lane_c & 15, part = lane_c >> 4, sample = 16 * wave + os;
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(lds + RAY_OFF) + sample * (RAY_FLOATS / 4);
        const f32x4 ro = rec[0], rd = rec[1];
        z_out = z_at(sn, ro.w, rd.w);
        const float px = __fadd_rn(ro.x, __fmul_rn(rd.x, z_out)), py = __fadd_rn(ro.y, __fmul_rn(rd.y, z_out)),
                    pz = __fadd_rn(ro.z, __fmul_rn(rd.z, z_out));
        g = grid_coords(a.sc, px, py, pz);
        if constexpr (CULL) ins = cull_inside(a.cull, a.sc, px, py, pz);
        const Lvl lv = load_lvl(lds, part);
        prep_boxes(lds, lv, g, ins, wave, true, os, part);
        __syncthreads();
        const Prep pr = prep_wg<4>(lds, part);
        tier = __builtin_amdgcn_readfirstlane(pr.tier);
        active = __builtin_amdgcn_readfirstlane(pr.any);
        if (tier == 0) {
            const TblLoad tl = tbl_issue(pr, lv, wave, os);
            tbl_store(lds, tl, lv, a.fold.zero_off, pr.K, tid);
        }
        lvl_store(lds, pr, tid, os, part);
    };
    // The step's U rows requested and its trilinear weights built (behind the barrier that published table and boxes).
    auto fetch_and_weights = [&](int lane_c) {
        if (tier == 0 && active) {
            const int os = lane_c & 15, part = lane_c >> 4;
            const UCfg u = ucfg(lds, lds_base, wave);
            dma_initial(a, lds, lane_c, wave, u);
            const Lvl lv = load_lvl(lds, part);
            wt_build(lds, lv, g, ins, wave, os, part, u.nch);
        }
    };
    prep_sequential(0, z_cur, lane);
    __syncthreads();
    fetch_and_weights(lane);
    for (int s = 0; s < S; ++s) {
        // loop-invariant address roots are laundered so that LICM does not hoist (and spill) hundreds of addresses
        int zero = 0, lane_i = lane;
        asm volatile("" : "+s"(zero), "+v"(lane_i));
        const int hi = lane_i >> 5, os = lane_i & 15, part = lane_i >> 4, sample = 16 * wave + os;
        const WSrc wl = {wrsrc, (unsigned)lane_i * 16u};
        char *actz = act + zero;
        const float *pk = reinterpret_cast<const float *>(actz + PRM_OFF);
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(actz + RAY_OFF) + sample * (RAY_FLOATS / 4);
        const int sn = sample >> 5, ss = sample & 31;  // N tile and column of this lane's sample
        const bool more = s + 1 < S;                   // uniform
        const bool ins_cur = ins;
        float z_next = 0.f;
        int tier_next = 0, active_next = 1;
        NB_HOOK_STEP_BEGIN;
        FOLD_STAMP(0);
        __syncthreads();  // Wt is visible
        FOLD_STAMP(1);
        if (!CULL || active) {
            f32x16 acc[2][2];
            // ---- fc_0 folded into the volume: H1_pre = b0 + U^T . Wt over the step's voxel list
            init_bias<2>(pk + P_B0, 2 * wave, hi, acc);
            if (tier == 0) {
                const UCfg u = ucfg(actz, lds_base, wave);
                fold_mfma(a, actz, lane_i, wave, u, acc, NB_HOOK_TBUF);
            } else if (tier == 3) {
                // a list of up to KM_CAP voxels (points that are neighbours but not dense, wide pixel footprints): the same boxes,
                // marched in passes of K_CAP voxels, each through table -> U -> Wt -> MFMA
                const Lvl lv = load_lvl(actz, part);
                const int K = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(actz + HDR_OFF));
                for (int k_lo = 0; k_lo < K; k_lo += K_CAP) {
                    tbl_pass(actz, a.fold.zero_off, k_lo, K, tid);
                    __syncthreads();
                    const UCfg u = ucfg_k(min(K - k_lo, K_CAP), lds_base, wave);
                    dma_initial(a, actz, lane_i, wave, u);
                    wt_build(actz, lv, g, ins_cur, wave, os, part, u.nch, k_lo);
                    __syncthreads();
                    fold_mfma(a, actz, lane_i, wave, u, acc);
                    __syncthreads();
                }
            } else {
                // rays far apart: sample groups of 16 (one wave's) or single samples, each through boxes -> table -> U -> Wt -> MFMA
                const Lvl lv = load_lvl(actz, part);
                const int n_groups = tier == 1 ? 4 : 64;
                for (int gi = 0; gi < n_groups; ++gi) {
                    const bool take = ins_cur && (tier == 1 ? wave == gi : sample == gi);
                    prep_boxes(actz, lv, g, take, 0, wave == (tier == 1 ? gi : gi >> 4), os, part);
                    __syncthreads();
                    const Prep pr = prep_wg<1>(actz, part);
                    const TblLoad tl = tbl_issue(pr, lv, wave, os);
                    tbl_store(actz, tl, lv, a.fold.zero_off, pr.K, tid);
                    lvl_store(actz, pr, tid, os, part);
                    __syncthreads();
                    const UCfg u = ucfg(actz, lds_base, wave);
                    dma_initial(a, actz, lane_i, wave, u);
                    wt_build(actz, lv, g, take, wave, os, part, u.nch);
                    __syncthreads();
                    fold_mfma(a, actz, lane_i, wave, u, acc);
                    __syncthreads();
                }
            }
            FOLD_DUMP(0, 2)
            FOLD_STAMP(2);
            ring_prime<P_L1>(wl, ring);
            publish_s(actz, lane_i, wave, acc);
            FOLD_STAMP(3);
            // ---- fc_1, fc_2
            init_bias<2>(pk + P_B1, 2 * wave, hi, acc);
            // The positional encodings of this step (lane (sample, axis a = part < 3): x_a, (sin, cos)(x_a 2^k) k < 10, v_a, (sin,
            // cos)(v_a 2^k) k < 4, two zeros; part 3: unused) and their conversion into operands ride behind fc_1's MFMAs, one slice
            // per MFMA step; the finished half-block waits in registers until the view layer has released the activation buffers.
            Conv6 pec;
            {
                const f32x4 ro = rec[0], rd = rec[1], rv = rec[2];  // ox oy oz near | dx dy dz far | vx vy vz |d|
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
                        sincos_rev2<t - 1>(tx, e[2 * t - 1], e[2 * t]);
                    } else if constexpr (t <= 14) {
                        sincos_rev2<t - 11>(tv, e[2 * t], e[2 * t + 1]);
                    } else if constexpr (t >= 16) {
                        constexpr int i = t - 16;
                        // (part 3 converts the z axis once more: its slots meet zero weights, pe_slot_col)
                        conv6_pair<2 * i>(pec, e[4 * i], e[4 * i + 1]);
                        conv6_pair<2 * i + 1>(pec, e[4 * i + 2], e[4 * i + 3]);
                    }
                };
                if constexpr (DENSITY_ONLY) layer_s<P_L1, 2, 4, true>(wl, actz, lane_i, ring, acc);
                else layer_s<P_L1, 2, 4, true>(wl, actz, lane_i, ring, acc, pe_fill);
            }
            HalfBlock peh;
            if constexpr (!DENSITY_ONLY) peh = conv6_finish(pec);
            FOLD_DUMP(1, 2)
            FOLD_STAMP(4);
            publish_s(actz, lane_i, wave, acc);
            FOLD_STAMP(5);
            init_bias<2>(pk + P_B2, 2 * wave, hi, acc);
            layer_s<P_L2, 2, 4, true>(wl, actz, lane_i, ring, acc);
            FOLD_DUMP(2, 2)
            FOLD_STAMP(6);
            // ---- the next step's sample: depth, grid coordinates, silhouette test, (wave, level) boxes — published by the
            // barriers of the publish below
            if (more) {
                const f32x4 ro = rec[0], rd = rec[1];
                z_next = z_at(s + 1, ro.w, rd.w);
                const float px = __fadd_rn(ro.x, __fmul_rn(rd.x, z_next)), py = __fadd_rn(ro.y, __fmul_rn(rd.y, z_next)),
                            pz = __fadd_rn(ro.z, __fmul_rn(rd.z, z_next));
                g = grid_coords(a.sc, px, py, pz);
                if constexpr (CULL) ins = cull_inside(a.cull, a.sc, px, py, pz);
                const Lvl lv = load_lvl(actz, part);
                prep_boxes(actz, lv, g, ins, wave, true, os, part);
            }
            FOLD_STAMP(7);
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
            FOLD_STAMP(8);
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
            if constexpr (!DENSITY_ONLY) {
                // ---- the next step's voxel list: the workgroup's boxes, K, and the index-grid lookups (in flight under the MFMAs
                // of the colour head's first phase, stored behind them)
                TblLoad tl;
                int k_next = 0;
                if (more) {
                    const Lvl lv = load_lvl(actz, part);
                    const Prep pr = prep_wg<4>(actz, part);
                    tier_next = __builtin_amdgcn_readfirstlane(pr.tier);
                    active_next = __builtin_amdgcn_readfirstlane(pr.any);
                    k_next = __builtin_amdgcn_readfirstlane(pr.K);
                    tl.rid[0] = tl.rid[1] = tl.idx[0] = tl.idx[1] = -1;
                    if (tier_next == 0) tl = tbl_issue(pr, lv, wave, os);
                    lvl_store(actz, pr, tid, os, part);
                }
                FOLD_STAMP(9);
                // ---- the colour head's linear part as ONE layer (feature_fc . latent_fc . view_fc folded at pack time, bias:
                // second block of nb_mlp_latent_bias): one tile per wave, K phase over fc_2's outputs, then over the encodings
                init_bias<1>(pk + P_LB, wave, hi, acc);
                layer_s<P_VG, 1, 4, false>(wl, actz, lane_i, ring, acc);
                FOLD_STAMP(10);
                if (more && tier_next == 0) {
                    const Lvl lv = load_lvl(actz, part);
                    tbl_store(actz, tl, lv, a.fold.zero_off, k_next, tid);
                }
                __syncthreads();
                ring_prime<P_VP>(wl, ring);
                store_halfblock(actz, part >> 1, part & 1, sn, ss, peh);  // the encodings converted behind fc_1
                __syncthreads();
                FOLD_STAMP(11);
                layer_s<P_VP, 1, 2, false>(wl, actz, lane_i, ring, acc);
                FOLD_STAMP(12);
                FOLD_DUMP(3, 1)
                // ---- rgb_fc partial sums over this wave's 32 view features
                {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        float rl[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) rl[r] = relu1(acc[0][n][r]);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            const f32x4 *rw = reinterpret_cast<const f32x4 *>(pk + P_RW + (ch * 2 + hi) * 64 + 16 * wave);
                            float c4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4) {
                                const f32x4 w4 = rw[q4];
                                c4[0] = fmaf(w4.x, rl[4 * q4 + 0], c4[0]);
                                c4[1] = fmaf(w4.y, rl[4 * q4 + 1], c4[1]);
                                c4[2] = fmaf(w4.z, rl[4 * q4 + 2], c4[2]);
                                c4[3] = fmaf(w4.w, rl[4 * q4 + 3], c4[3]);
                            }
                            float sc = add_halves((c4[0] + c4[1]) + (c4[2] + c4[3]));
                            if (hi == 0) reinterpret_cast<float *>(actz + SCR_C)[(ch * 64 + n * 32 + (lane_i & 31)) * 4 + wave] = sc;
                        }
                    }
                }
            }
        } else if (more) {
            prep_sequential(s + 1, z_next, lane_i);
            tier_next = tier;
            active_next = active;
        }
        FOLD_STAMP(13);
        __syncthreads();  // the activation buffers are free; the heads' partial sums, the next step's table and boxes are visible
        FOLD_STAMP(14);
        // ---- the next step's U rows and trilinear weights: in flight / built under this step's heads and compositing
        if (more) {
            tier = tier_next;
            active = active_next;
            fetch_and_weights(lane_i);
        }
        FOLD_STAMP(15);
        // ---- owner lanes: finish the heads; composite (rays) or hand the decoder output over (points)
        {
            if constexpr (POINTS) {
                if (valid && part == 0) {
                    const float sigma = head_sum(actz, SCR_A + sample * 16, pk[P_AB]);
                    if constexpr (DENSITY_ONLY) {
                        a.raw_out[ray] = sigma;
                    } else {
                        float o3[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) o3[ch] = head_sum(actz, SCR_C + (ch * 64 + sample) * 16, pk[P_RB + ch]);
                        *reinterpret_cast<f32x4 *>(a.raw_out + ray * 4) = f32x4{o3[0], o3[1], o3[2], sigma};
                    }
                }
            } else {
                if (s + 1 >= S && a.ill != nullptr)  // uniform
                    list_ill_ray(actz, pk, a.ill, a.ill_cap, sample, part, z_cur, (valid && (!CULL || ins_cur)) ? (int)ray : -1);
                composite_step(actz, pk, sample, part, z_cur, z_next, s + 1 >= S, a, ray, s, S, valid, wstore, !CULL || ins_cur);
            }
        }
        FOLD_STAMP(16);
        z_cur = z_next;
    }
    if (!POINTS && valid && (lane >> 4) == 0) {
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(lds + RAY_OFF) + (16 * wave + (lane & 15)) * (RAY_FLOATS / 4);
        const f32x4 c0 = rec[3], c1 = rec[4];
        RayAccum ra;
        ra.T = c0.x; ra.cr = c0.y; ra.cg = c0.z; ra.cb = c0.w; ra.depth = c1.x; ra.accw = c1.y;
        ra.store(a, ray);
    }
}

// ---------------------------------------------------------------- last-sample fix-up
// The rays composite_step listed (their last density within NB_ILL_SIGMA of zero): one workgroup per record recomputes that ONE
// density at fp32 level — fc_0 through the same folded planes (head + remainder = 22 bits of fc_0 . V, trilinear weights in fp32,
// fp64 accumulation), fc_1 / fc_2 / alpha_fc from the fp32 fragments of the packed blob (fp64 accumulation, activations rounded to
// fp32 between the layers like the reference's) — and composites the last sample again from the listed state, exactly as
// RayAccum::add / store do.  Thread f owns natural feature f.
__device__ __forceinline__ int frag_bias_index(int f) {  // natural feature f inside a [tile][hi][16] block (b_pack, nb_march.hip)
    const int t = f >> 5, i = f & 31;
    return (t * 2 + ((i >> 2) & 1)) * 16 + (i & 3) + 4 * (i >> 3);
}
// row `row` of a 256 x 256 layer stored as fp32 A fragments (a_pack kind 1): the 16 bytes at ((tile * 32 + g) * 64 + hi * 32 + row % 32)
// hold the weights of input columns 8 g + 4 hi .. + 3 (col_hidden(4 g + i, hi) = 8 g + 4 hi + i)
__device__ __forceinline__ double frag_row_dot(const float *wfrag, const float *h, int row) {
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(wfrag) + (size_t)(row >> 5) * 32 * 64 + (row & 31);
    double s = 0.0;
    for (int g = 0; g < 32; ++g)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const f32x4 w = w4[g * 64 + hi * 32];
            const float *hh = h + 8 * g + 4 * hi;
            s = fma((double)w.x, (double)hh[0], s);
            s = fma((double)w.y, (double)hh[1], s);
            s = fma((double)w.z, (double)hh[2], s);
            s = fma((double)w.w, (double)hh[3], s);
        }
    return s;
}
__global__ __launch_bounds__(256) void nb_march_fixup_kernel(MarchArgs a) {
    __shared__ float rec[NB_ILL_RECORD_FLOATS];
    __shared__ float cw[32];
    __shared__ int crow[32];
    __shared__ float h[2][256];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    int *hdr = reinterpret_cast<int *>(a.ill);
    const int n = min(hdr[0], a.ill_cap);
    const int S = a.n_samples;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        __syncthreads();
        if (tid < NB_ILL_RECORD_FLOATS) rec[tid] = a.ill[NB_ILL_HEADER_FLOATS + (long long)it * NB_ILL_RECORD_FLOATS + tid];
        __syncthreads();
        const long long ray = __float_as_int(rec[0]);
        const float z = rec[4];
        if (tid < 32) {  // (level, corner): weight and U row, with the march's own arithmetic (prep_sequential, wt_build)
            const int L = tid >> 3, corner = tid & 7;
            const float px = __fadd_rn(a.ray_o[ray * 3 + 0], __fmul_rn(a.ray_d[ray * 3 + 0], z)),
                        py = __fadd_rn(a.ray_o[ray * 3 + 1], __fmul_rn(a.ray_d[ray * 3 + 1], z)),
                        pz = __fadd_rn(a.ray_o[ray * 3 + 2], __fmul_rn(a.ray_d[ray * 3 + 2], z));
            const GridCoord g = grid_coords(a.sc, px, py, pz);
            Lvl lv;
            lv.D = a.sc.dhw[L][0]; lv.H = a.sc.dhw[L][1]; lv.W = a.sc.dhw[L][2];
            lv.fmx = a.sc.fm1[L][2]; lv.fmy = a.sc.fm1[L][1]; lv.fmz = a.sc.fm1[L][0];
            lv.fpx = a.sc.fp1[L][2]; lv.fpy = a.sc.fp1[L][1]; lv.fpz = a.sc.fp1[L][0];
            const LvlIdx q = level_index(lv, g);
            const float fx = (float)q.x0, fy = (float)q.y0, fz = (float)q.z0;
            const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
            const float wx = dx ? q.ix - fx : (fx + 1.f) - q.ix, wy = dy ? q.iy - fy : (fy + 1.f) - q.iy, wz = dz ? q.iz - fz : (fz + 1.f) - q.iz;
            const int xx = q.x0 + dx, yy = q.y0 + dy, zz = q.z0 + dz;
            const bool inb = (unsigned)xx < (unsigned)lv.W && (unsigned)yy < (unsigned)lv.H && (unsigned)zz < (unsigned)lv.D;
            int row = -1;
            if (inb) {
                const int rid = a.fold.grid[L][((size_t)zz * lv.H + yy) * lv.W + xx];
                if (rid >= 0) row = a.fold.row_base[L] + rid;
            }
            cw[tid] = (wx * wy) * wz;
            crow[tid] = row;
        }
        __syncthreads();
        {
            double s = (double)a.pk[F_OFF_B0 + frag_bias_index(tid)];
            for (int c = 0; c < 32; ++c) {
                if (crow[c] < 0) continue;  // uniform
                const _Float16 *u = reinterpret_cast<const _Float16 *>(a.fold.urows) + (size_t)crow[c] * 512;
                s = fma((double)cw[c], (double)(float)u[tid] + (double)(float)u[256 + tid], s);
            }
            h[0][tid] = fmaxf((float)s, 0.f);
        }
        __syncthreads();
        h[1][tid] = fmaxf((float)((double)a.pk[F_OFF_B1 + frag_bias_index(tid)] + frag_row_dot(a.pk + F_OFF_B0 + 256, h[0], tid)), 0.f);
        __syncthreads();
        const float h3 = fmaxf((float)((double)a.pk[F_OFF_B2 + frag_bias_index(tid)] + frag_row_dot(a.pk + F_OFF_B1 + 256, h[1], tid)), 0.f);
        {
            const int t = tid >> 5, i = tid & 31;  // alpha_fc: [hi][q] with column col_hidden(q, hi)
            double p = (double)a.pk[F_OFF_AW + ((i >> 2) & 1) * 128 + 16 * t + (i & 3) + 4 * (i >> 3)] * (double)h3;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) p += __shfl_xor(p, m);
            if ((tid & 63) == 0) red[tid >> 6] = p;
        }
        __syncthreads();
        if (tid == 0) {
            const float sigma = (float)(((red[0] + red[1]) + (red[2] + red[3])) + (double)a.pk[F_OFF_AB]);
            RayAccum ra;
            ra.T = rec[1]; ra.depth = rec[2]; ra.accw = rec[3];
            ra.cr = rec[8]; ra.cg = rec[9]; ra.cb = rec[10];
            const float out4[4] = {rec[12], rec[13], rec[14], sigma};
            const float w = ra.add(out4, z, rec[5]);
            ra.store(a, ray);
            a.weights[ray * S + (S - 1)] = w;
            if (a.raw) a.raw[(ray * S + (S - 1)) * 4 + 3] = sigma;
            if ((sigma > 0.f) != (rec[6] > 0.f)) atomicAdd(&hdr[1], 1);  // the march had taken the other branch of the step
        }
    }
}

// ---------------------------------------------------------------- weight stream packing
// input column of the layer phase `ph` held by element e of half-block (b, kh); -1 = zero padding
__device__ __forceinline__ int phase_col(int ph, int b, int kh, int e) {
    if (ph < 3) return col_hidden(32 * b + e, kh);
    return pe_slot_col(2 * b + kh, e);  // the encodings: the owner lane's part
}
__device__ __forceinline__ float phase_weight(const nb_mlp_params &p, const float *f32_blob, int ph, int row, int b, int kh, int e) {
    const int col = phase_col(ph, b, kh, e);
    if (col < 0) return 0.f;
    if (ph == 0) return p.fc1_w[row * 256 + col];
    if (ph == 1) return p.fc2_w[row * 256 + col];
    if (ph == 2) {
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

// fp4 e2m1 (sign, 2 exponent bits of bias 1, 1 mantissa bit: 0 0.5 1 1.5 2 3 4 6), round to nearest (ties to the even code)
__device__ __forceinline__ unsigned fp4_e2m1_bits(float v) {
    const unsigned sgn = v < 0.f ? 8u : 0u;
    const float a = fminf(fabsf(v), 6.f);
    constexpr float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    unsigned code = 0;
    float best = a;
    for (unsigned c = 1; c < 8; ++c) {
        const float d = fabsf(a - grid[c]);
        if (d < best || (d == best && (c & 1u) == 0u)) {
            best = d;
            code = c;
        }
    }
    return sgn | code;
}

// one thread per (wave, piece, lane): the lane's 16 bytes (+, for a four-bit fragment, its scale byte in the block's scale dword)
__global__ void nb_pack_fold_kernel(nb_mlp_params p, const float *__restrict__ f32_blob, unsigned *__restrict__ out, int *__restrict__ stats) {
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
    const int mt = PH_MT[ph], ppb = block_pieces(mt);
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
    } else {  // a cross fragment: k = 0 W_h (multiplies the interleaved remainder operand), k = 1 W_l (natural order) — or the block's hole
        int r6 = r - 4 * mt, k = -1, m = 0, half = 0;
        for (int slot = 0; slot < 2; ++slot) {
            const int kk = TERM_ORDER[slot], width = mt * term_pieces(kk);
            if (r6 < width) {
                k = kk;
                m = r6 / term_pieces(kk);
                half = r6 % term_pieces(kk);
                break;
            }
            r6 -= width;
        }
        if (k >= 0) {
            const bool four = FP4_K[k];
            const float top = four ? 6.f : 7.5f;  // largest magnitude of the format
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
                ex = ilogbf(amax / top);
                if (ldexpf(top, ex) < amax) ++ex;  // the smallest power of two with max / 2^ex <= top
                ex = min(max(ex, -120), 120);
            }
            if (four) {
                for (int e = 0; e < 32; ++e) w32[e >> 3] |= fp4_e2m1_bits(ldexpf(wv[e], -ex)) << (4 * (e & 7));
                unsigned char *sc = reinterpret_cast<unsigned char *>(out) + (size_t)w * WAVE_STREAM + (size_t)P_TOTAL * 1024 +
                                    (size_t)(phase_s0(ph) + b) * 256 + lane * 4 + (k * mt + m);
                *sc = (unsigned char)(127 + ex);
            } else {
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
            // statistic behind nb_mlp_six_bit_stats_offset(): how many non-zero head weights sit below 1/8 of their block's
            // maximum, i.e. where e2m3 keeps fewer than 3 bits and e2m1 none (per layer: small, non-zero)
            if (k == 0 && half == 0) {
                int small = 0, nz = 0;
                for (int e = 0; e < 32; ++e) {
                    nz += wv[e] != 0.f;
                    small += wv[e] != 0.f && fabsf(wv[e]) < 0.125f * amax;
                }
                const int layer = ph < 2 ? ph : 2;  // fc_1, fc_2, the folded colour head (both of its K phases)
                atomicAdd(&stats[2 * layer], small);
                atomicAdd(&stats[2 * layer + 1], nz);
            }
        }
    }
    unsigned *dst = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(out) + (size_t)w * WAVE_STREAM) + ((size_t)piece * 64 + lane) * 4;
    for (int q = 0; q < 4; ++q) dst[q] = w32[q];
}

}  // namespace

namespace nbm {

long long fold_stream_floats() { return (long long)4 * WAVE_STREAM / 4 + 8; }  // + the six-bit statistic (6 ints, 2 pad)

int pack_fold_stream(const nb_mlp_params *p, float *packed, long long stream_off, hipStream_t st) {
    const long long n = (long long)4 * P_TOTAL * 64;
    int *stats = reinterpret_cast<int *>(packed + stream_off + (long long)4 * WAVE_STREAM / 4);
    NB_REQUIRE(hipMemsetAsync(stats, 0, 8 * sizeof(int), st) == hipSuccess, "pack_fold_stream: hipMemsetAsync failed");
    hipLaunchKernelGGL(nb_pack_fold_kernel, dim3(nb_ceil_div(n, 256)), dim3(256), 0, st, *p, packed,
                       reinterpret_cast<unsigned *>(packed + stream_off), stats);
    NB_CHECK_LAUNCH("nb_pack_fold_kernel");
    return NB_OK;
}

int launch_march_fold(MarchArgs a, long long stream_off, hipStream_t st) {
    a.n_wave_groups = (int)nb_ceil_div(a.ray_order ? a.n_slots : a.n_rays, 64);
    const char *stream = reinterpret_cast<const char *>(a.pk + stream_off);
    if (a.ill) NB_REQUIRE(hipMemsetAsync(a.ill, 0, NB_ILL_HEADER_FLOATS * sizeof(float), st) == hipSuccess, "nb_march: hipMemsetAsync(ill_scratch) failed");
    if (a.cull.n_views) hipLaunchKernelGGL((nb_march_fold_kernel<0, true>), dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    else hipLaunchKernelGGL((nb_march_fold_kernel<0, false>), dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    NB_CHECK_LAUNCH("nb_march_fold_kernel");
    if (a.ill) {
        hipLaunchKernelGGL(nb_march_fixup_kernel, dim3(NB_ILL_FIXUP_BLOCKS), dim3(256), 0, st, a);
        NB_CHECK_LAUNCH("nb_march_fixup_kernel");
    }
    return NB_OK;
}

// nb_decode_points on the same kernel: every point a one-sample ray whose decoder output is stored instead of composited
int launch_points_fold(MarchArgs a, int density_only, long long stream_off, hipStream_t st) {
    a.n_wave_groups = (int)nb_ceil_div(a.n_pts, 64);
    const char *stream = reinterpret_cast<const char *>(a.pk + stream_off);
    if (density_only) hipLaunchKernelGGL((nb_march_fold_kernel<2, false>), dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    else hipLaunchKernelGGL((nb_march_fold_kernel<1, false>), dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    NB_CHECK_LAUNCH("nb_march_fold_kernel (points)");
    return NB_OK;
}

}  // namespace nbm
