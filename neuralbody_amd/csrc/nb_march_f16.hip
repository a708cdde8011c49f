// nb_march_f16.hip — "f16f8" march kernel for gfx950 (MI355X): fp16 main product + scaled 8-bit cross terms.
//
// Same per-wave organisation as nb_march_bf16.hip (one wave = 32 sample columns, activations resident in registers,
// layers computed transposed so the C/D fragment of layer l is the B operand of layer l+1, weights streamed through an
// LDS ring by LDS-DMA), different arithmetic.  With W = W_h + W_l and X = X_h + X_l (fp16 head + remainder):
//
//     W.X  ~=  W_h.X_h                                   v_mfma_f32_32x32x16_f16            (products exact, fp32 accumulate)
//            + fp8(W_h 2^a).bf8(X_l 2^12) 2^-(a+12)      v_mfma_scale_f32_32x32x64_f8f6f4   (K = 64 per instruction; the
//            + fp8(W_l 2^b).bf8(X_h)      2^-b            E8M0 scale operands undo the 2^k)
//
// (feature_fc, latent_fc and view_fc are folded into one layer at pack time, see the stream geometry below.)
// The cross terms are 2^-11 of the main term, so 3-4 significant bits are enough for them: ~2^-15 relative error per
// term (round 1's three bf16 products: 2^-16) at 1.94 instead of 3 matrix-pipe units — on MI355X the K=64 scaled 8-bit
// MFMA costs 1.88x a K=16 fp16 MFMA (28.7 vs 15.3 ns, profiles/r02_probe_mxrate.log).  Activations use bf8 e5m2 (fp16's exponent range:
// nothing to clamp), weights fp8 e4m3 with one power-of-two scale per layer chosen at pack time from max|W_h|, max|W_l|
// (nb_f16_scales_kernel).  Operand layout of the scaled MFMA (probed, profiles/r02_probe_mx.log): lane l holds
// row/column l%32 and K elements 32*(l/32)..+31 in its 8 registers, little endian; the scale byte of lane l applies to
// that row/column and K half.
//
// Measured RGB error against the reference renderer over all fixtures: profiles/r02_precision_sweep.md.
//
// -DF_SIX=1 (nb_march_f6.hip compiles this file a second time) builds the "f16f6" variant: the cross terms in SIX bits —
// weights fp6 e2m3 with a pack-time E8M0 scale per (row, 32 K), activations bf6 e3m2 with a run-time E8M0 scale per
// (sample, 32 K) taken from the exponent of the block's largest |value|.  With both operands in six bits the K=64 scaled
// MFMA issues at the rate of ONE K=16 fp16 MFMA (8-bit operands: 1.9x; profiles/r02_probe_mxrate.log), and one
// v_cvt_scalef32_pk32_bf6_f16 / v_cvt_scalef32_2xpk16_bf6_f32 converts 32 values (profiles/r02_probe_cvt6.log).
#include <type_traits>
#include <utility>

#include "nb_f6_ops.h"

using namespace nbm;

#ifndef F_SIX
#define F_SIX 0
#endif

constexpr bool SIX = F_SIX != 0;
// one K=64 block of an 8-bit / 6-bit activation operand
using XB = std::conditional_t<SIX, i32x6, i32x8>;

namespace nbm {

// --------------------------------------------------------------- weight stream geometry
// 2-KiB records.  A layer phase (NT output tiles in pairs, NBLK K-blocks of 64, the last block with NCH_LAST 16-wide
// chunks) is streamed pair by pair, block by block:  M(c) = [A16(c,t0) | A16(c,t1)] for every chunk c of the block, then
// X8h(t0), X8h(t1), X8l(t0), X8l(t1) = fp8 fragments (64 lanes x 32 B) of W_h and W_l for that block.
constexpr int FREC_BYTES = 2048;
constexpr int FPAGE_RECS = 12;
constexpr int FPAGE_BYTES = FPAGE_RECS * FREC_BYTES;  // 24 KiB
constexpr int FN_SLOTS = 5;
constexpr int FAHEAD = FN_SLOTS - 1;
constexpr int FDMA_PER_WAVE = FPAGE_BYTES / 1024 / 4;  // 6

__host__ __device__ constexpr int recs_per_pair(int nblk, int nch_last) { return (nblk - 1) * 8 + nch_last + 4; }
__host__ __device__ constexpr int recs_phase(int nt, int nblk, int nch_last) { return (nt / 2) * recs_per_pair(nblk, nch_last); }
// fc_0 is ONE phase over the 176 gathered values of a lane in the order [level 1 (32) | level 2 (64) | level 3 (64) |
// level 0 (16) | 16 zeros] = 6 K-blocks, the last with 2 chunks (all four pyramid levels are gathered before the layer
// starts: their tile fetches are in flight together, and the 184 records run without a gather in between).
// feature_fc, latent_fc and view_fc have NO activation between them (latent_xyzc.py:105-119): they run as ONE linear layer
// view_w[:, :256] . latent_w[:, :256] . feature_w (128 x 256, product formed in fp64 at pack time) over fc_2's output, plus
// view_w[:, 256:] over the encodings; the per-frame latent code and all three biases are in nb_mlp_latent_bias()'s second
// block.  (Rounds 1-2a ran feature_fc . latent_fc as a 256 x 256 layer of its own: 128 MFMAs, 64 records and one operand
// conversion per depth step for nothing.)
constexpr int FR_F0 = 0;
constexpr int FR_L1 = FR_F0 + recs_phase(8, 6, 2);
constexpr int FR_L2 = FR_L1 + recs_phase(8, 4, 4);
constexpr int FR_VG = FR_L2 + recs_phase(8, 4, 4);   // folded colour-head layer over fc_2's 256 outputs
constexpr int FR_VP = FR_VG + recs_phase(4, 4, 4);   // view_fc over the 45 (x2 halves) positional-encoding slots, padded to 64
constexpr int FN_RECS = FR_VP + recs_phase(4, 2, 2);
constexpr int FN_RECS_PAD = (FN_RECS + 5 * FPAGE_RECS - 1) / (5 * FPAGE_RECS) * (5 * FPAGE_RECS);  // zero records up to a multiple of FN_SLOTS pages
static_assert(FN_RECS == 532, "record count");
constexpr int FN_PAGES = FN_RECS_PAD / FPAGE_RECS;
static_assert(FN_PAGES % FN_SLOTS == 0, "page p must always land in slot p % FN_SLOTS, also across the step wrap-around");
constexpr int F_N_SCALES = 16;  // ints behind the stream: E8M0 scale operands (W_h, W_l) of fc_0, fc_1, fc_2, the folded view layer

}  // namespace nbm

namespace {

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

// small parameters staged in LDS behind the ring: offsets in floats (P_SC: the 10 scale operands, ints)
constexpr int P_B0 = 0, P_B1 = 256, P_B2 = 512, P_LB = 768, P_BV = 1024, P_AW = 1152, P_RW = 1408, P_AB = 1792, P_RB = 1796,
              P_SC = 1800, P_SIZE = 1800 + F_N_SCALES;
constexpr int RING_BYTES = FN_SLOTS * FPAGE_BYTES;  // 120 KiB
constexpr int PARAM_BYTES = 8192;
constexpr int TILE_BYTES = 8192;
constexpr int LDS_BYTES = RING_BYTES + PARAM_BYTES + 4 * TILE_BYTES;
static_assert(LDS_BYTES <= 163840, "LDS budget (160 KiB per workgroup)");
static_assert((P_SIZE + 4) * 4 <= PARAM_BYTES, "parameter region overflow (4 ints of cull flags follow the parameters)");

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct FRing {
    const char *stream;  // this wave's share of the stream (uniform)
    char *lds;
    int wave_off;
    int lane;
    char *tile;
    int base[FN_SLOTS];  // lane * 16 + slot * FPAGE_BYTES
};

__device__ __forceinline__ void f_issue_page(const FRing &rg, int page) {
    const int slot = page % FN_SLOTS;
    // uniform 64-bit base (scalar adds) + the lane's 32-bit byte offset: `global_load_lds_dwordx4 v_off, s[base] offset:imm`
    const char *base = rg.stream + (size_t)page * FPAGE_BYTES;
    const unsigned voff = (unsigned)rg.lane * 16u;
    // the instruction offset (13-bit signed: at most 4095) advances BOTH the global and the LDS address: one M0 value
    // per four pieces
    char *dst = rg.lds + slot * FPAGE_BYTES + rg.wave_off;
    static_for<FDMA_PER_WAVE>([&](auto ic) {
        constexpr int i = decltype(ic)::value, grp = i / 4, off = (i % 4) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + grp * 4096 + voff), (lptr_t)(dst + grp * 4096), 16, off, 0);
    });
}

// one 1-KiB piece of a page (the lane's 16 bytes): pieces 4, 5 take a second M0 / base (13-bit instruction offset).
// Issued from inline asm in the `v_off, s[base]` form: left to the compiler the depth loop uses a 64-bit per-lane address
// (one v_lshl_add_u64 + copies per piece, all VALU issue slots, which in this kernel are as expensive as MFMA slots)
template <int I>
__device__ __forceinline__ void f_issue_piece(const FRing &rg, int page) {
    const int slot = page % FN_SLOTS;
    constexpr int grp = I / 4, off = (I % 4) * 1024;
#ifdef F_DMA_BUILTIN
    const char *base = rg.stream + (size_t)page * FPAGE_BYTES;
    const unsigned voff = (unsigned)rg.lane * 16u;
    char *dst = rg.lds + slot * FPAGE_BYTES + rg.wave_off;
    __builtin_amdgcn_global_load_lds((gptr_t)(base + grp * 4096 + voff), (lptr_t)(dst + grp * 4096), 16, off, 0);
#else
    const char *base = rg.stream + (size_t)page * FPAGE_BYTES + grp * 4096;  // wave-uniform
    const unsigned ldsoff = (unsigned)(unsigned long long)(lptr_t)(rg.lds + slot * FPAGE_BYTES + rg.wave_off + grp * 4096);
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:%3"
        :
        : "v"(rg.base[0]), "s"(base), "s"(ldsoff), "n"(off)
        : "memory", "m0");
#endif
}

// before the first record of `page` is read: everything but the DMAs of the FAHEAD-1 younger pages has landed, and every
// wave is done with the slot that is refilled next
__device__ __forceinline__ void f_turn_page(const FRing &rg, int page) {
#ifdef F_ABL_NODMA
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((FAHEAD - 1) * FDMA_PER_WAVE) : "memory");
#endif
#ifndef F_ABL_NOBAR
    asm volatile("s_barrier" ::: "memory");
#endif
#if !defined(F_ABL_NODMA) && defined(F_DMA_BURST)
    f_issue_page(rg, (page + FAHEAD) % FN_PAGES);
#endif
}

struct Rec {
    i32x4 p0, p1;  // the lane's two 16-byte pieces of a 2-KiB record
};

static_assert(FPAGE_BYTES <= 65536, "a page must be addressable with the 16-bit ds_read offset");
// fragment reads are issued from inline asm (not counted by the compiler) and waited for with hand-counted lgkmcnt,
// exactly as in nb_march_bf16.hip
template <int REC>
__device__ __forceinline__ void load_rec(const FRing &rg, Rec &f) {
    if (REC % FPAGE_RECS == 0) f_turn_page(rg, REC / FPAGE_RECS);
    constexpr int slot = (REC / FPAGE_RECS) % FN_SLOTS;
    constexpr int off = (REC % FPAGE_RECS) * FREC_BYTES;
#ifdef F_ABL_NOLDS
    asm volatile("" : "=v"(f.p0), "=v"(f.p1) : "v"(rg.base[slot]));
    return;
#endif
#if !defined(F_DMA_BURST) && !defined(F_ABL_NODMA) && !defined(F_DMA_BUILTIN)
    // the six DMA pieces of the page FAHEAD pages ahead ride on the odd records of this page instead of forming a burst
    // behind the page's barrier (where the matrix pipe has nothing to do); same issue order as the burst, so the counted
    // vmcnt of f_turn_page is unchanged.  One asm block with the record's fragment reads: M0 (the LDS destination) is
    // written first, the two reads give it the wait state the LDS-DMA instruction needs (no s_nop), and the uniform 64-bit
    // source address stays in SGPRs (`v_off, s[base]` form: no per-lane 64-bit VALU add)
    if constexpr ((REC % FPAGE_RECS) % 2 == 1 && (REC % FPAGE_RECS) / 2 < FDMA_PER_WAVE) {
        constexpr int I = (REC % FPAGE_RECS) / 2, grp = I / 4, poff = (I % 4) * 1024;
        constexpr int page = (REC / FPAGE_RECS + FAHEAD) % FN_PAGES, pslot = page % FN_SLOTS;
        const char *base = rg.stream + (size_t)page * FPAGE_BYTES + grp * 4096;  // wave-uniform
        const unsigned ldsoff = (unsigned)(unsigned long long)(lptr_t)(rg.lds + pslot * FPAGE_BYTES + rg.wave_off + grp * 4096);
        asm volatile(
            "s_mov_b32 m0, %6\n\t"
            "ds_read_b128 %0, %2 offset:%3\n\t"
            "ds_read_b128 %1, %2 offset:%4\n\t"
            "global_load_lds_dwordx4 %7, %5 offset:%8"
            : "=&v"(f.p0), "=&v"(f.p1)
            : "v"(rg.base[slot]), "n"(off), "n"(off + 1024), "s"(base), "s"(ldsoff), "v"(rg.base[0]), "n"(poff)
            : "memory", "m0");
        return;
    }
#endif
    asm volatile(
        "ds_read_b128 %0, %2 offset:%3\n\t"
        "ds_read_b128 %1, %2 offset:%4"
        : "=&v"(f.p0), "=&v"(f.p1)
        : "v"(rg.base[slot]), "n"(off), "n"(off + 1024)
        : "memory");
#if !defined(F_DMA_BURST) && !defined(F_ABL_NODMA) && defined(F_DMA_BUILTIN)
    if constexpr ((REC % FPAGE_RECS) % 2 == 1 && (REC % FPAGE_RECS) / 2 < FDMA_PER_WAVE)
        f_issue_piece<(REC % FPAGE_RECS) / 2>(rg, (REC / FPAGE_RECS + FAHEAD) % FN_PAGES);
#endif
}
// LDS operations retire in order: "at most NEWER outstanding" = everything older than the NEWER reads issued last has landed
template <int NEWER>
__device__ __forceinline__ void wait_rec(Rec &f) {
#ifdef F_WAIT_ASM
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.p0), "+v"(f.p1) : "n"(NEWER));
#else
    // the builtin, fenced: as inline asm with the fragment registers tied to it (so that the MFMA cannot move above the wait)
    // hipcc treats the statement as a writer of those registers and puts an `s_nop 0` in front of every MFMA
    static_assert(NEWER < 16, "lgkmcnt field");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F | (NEWER << 8));  // vmcnt 63, expcnt 7, lgkmcnt NEWER
    __builtin_amdgcn_sched_barrier(0);
    (void)f;
#endif
}

__device__ __forceinline__ f32x16 mfma_main(const i32x4 a, const f16x8 b, const f32x16 c) {
#ifdef F_ABL_NOM
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), b, c, 0, 0, 0);
}
// A: fp8 e4m3 (cbsz 0), B: bf8 e5m2 (blgp 1); scale operands: E8M0 in byte 0 of each lane's register
__device__ __forceinline__ f32x16 mfma_cross(const Rec &a, const i32x8 b, const f32x16 c, int scale_a, int scale_b) {
#ifdef F_ABL_NOX
    return c;
#endif
    const i32x8 av = {a.p0.x, a.p0.y, a.p0.z, a.p0.w, a.p1.x, a.p1.y, a.p1.z, a.p1.w};
#ifdef F_ABL_UNSCALED
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b, c, 0, 1, 0, 0, 0, 0);
#endif
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b, c, 0, 1, 0, scale_a, 0, scale_b);
}

// six-bit cross term: A fp6 e2m3 (cbsz 2) in the first 24 bytes of the lane's record share, its E8M0 scale in byte 24;
// B bf6 e3m2 (blgp 3), E8M0 scale = byte BSEL of `sb` (one register carries the scales of four K blocks)
// the record share as ONE 8-register tuple (only the first six are the operand; register 6 is the scale): the two 4-register
// fragment reads coalesce into it in place — a 6-of-8 use of two separate reads costs two copies per record.  Pinned BEFORE the
// counted wait of the record: an (empty) asm directly in front of the MFMA makes hipcc put an `s_nop 0` between them.
__device__ __forceinline__ i32x8 cross6_operand(const Rec &a) {
    i32x8 av = {a.p0.x, a.p0.y, a.p0.z, a.p0.w, a.p1.x, a.p1.y, a.p1.z, a.p1.w};
    asm volatile("" : "+v"(av));
    return av;
}
template <int BSEL>
__device__ __forceinline__ f32x16 mfma_cross6(const i32x8 av, const i32x6 b, const f32x16 c, int sb) {
#ifdef F_ABL_NOX
    return c;
#endif
    const i32x8 bv = {b[0], b[1], b[2], b[3], b[4], b[5], 0, 0};
#ifdef F_NO_OPSEL
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 2, 3, 0, av[6], 0, sb >> (8 * BSEL));
#endif
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 2, 3, 0, av[6], BSEL, sb);
}

__device__ __forceinline__ f32x16 f_bias_tile(const float *bp, int t, int hi) {
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(bp + (t * 2 + hi) * 16);
    const f32x4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
    return f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
}

// ---------------------------------------------------------------- operands
// 16 consecutive values of a lane (one accumulator tile, or 16 gathered features / encodings) in the three forms the
// MFMAs consume: fp16 head (two K=16 chunks), bf8 of the remainder * 2^12 and bf8 of the value (half a K=64 block each)
constexpr int LO_SHIFT = 12;  // |X_l| <= 2^-11 |X|: the scaled remainder stays below 2 |X|, inside bf8's range whenever X_h is finite

// relu as exactly one v_max_f32 (hipcc lowers fmaxf / fmed3 on a value that feeds inline asm to canonicalise + max)
__device__ __forceinline__ float relu_asm(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// acc += w * relu(x) as one asm block (separate statements get an `s_nop 0` between the asm v_max and the compiler's v_fmac)
__device__ __forceinline__ void relu_fma(float &acc, float w, float x) {
    float t;
    asm("v_max_f32 %1, 0, %2\n\tv_fmac_f32 %0, %3, %1" : "+v"(acc), "=&v"(t) : "v"(x), "v"(w));
}
__device__ __forceinline__ void relu_fma3(float &a0, float &a1, float &a2, float w0, float w1, float w2, float x) {
    float t;
    asm("v_max_f32 %3, 0, %4\n\tv_fmac_f32 %0, %5, %3\n\tv_fmac_f32 %1, %6, %3\n\tv_fmac_f32 %2, %7, %3"
        : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(t)
        : "v"(x), "v"(w0), "v"(w1), "v"(w2));
}
template <bool RELU, class Get>
__device__ __forceinline__ void make_ops16(Get get, f16x8 &h0, f16x8 &h1, i32x4 &l, i32x4 &x) {
    int lw[4], xw[4];
    unsigned hw[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {  // one 32-bit word of the 8-bit forms = 4 values
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = get(4 * w + i);
            if (RELU) v[i] = relu1(v[i]);
        }
#ifdef F_ABL_NOCONV
        hw[2 * w] = __float_as_uint(v[0]);
        hw[2 * w + 1] = __float_as_uint(v[1]);
        lw[w] = __float_as_int(v[2]);
        xw[w] = __float_as_int(v[3]);
        continue;
#endif
        const unsigned ha = cvt_pk_f16(v[0], v[1]), hb = cvt_pk_f16(v[2], v[3]);
        hw[2 * w] = ha;
        hw[2 * w + 1] = hb;
        // the scaled conversion DIVIDES by its scale operand (probed, tools/experiments/probe_cvt.hip)
        constexpr float inv = 1.0f / (float)(1 << LO_SHIFT);
        const i16x2 zero = {0, 0};
        i16x2 lp = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(zero, rem16<0>(v[0], ha), rem16<1>(v[1], ha), inv, false);
        lp = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(lp, rem16<0>(v[2], hb), rem16<1>(v[3], hb), inv, true);
        lw[w] = __builtin_bit_cast(int, lp);
        const int xp = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
        xw[w] = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], xp, true);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    h0 = __builtin_bit_cast(f16x8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    h1 = __builtin_bit_cast(f16x8, u32x4{hw[4], hw[5], hw[6], hw[7]});
    l = i32x4{lw[0], lw[1], lw[2], lw[3]};
    x = i32x4{xw[0], xw[1], xw[2], xw[3]};
}

__device__ __forceinline__ i32x8 cat8(const i32x4 a, const i32x4 b) { return i32x8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}; }

// NG groups of 16 values -> 2 NG chunks and ceil(NG / 2) blocks (an odd last group is padded with zeros)
template <int NG, bool RELU, class Get>
__device__ __forceinline__ void make_operands(Get get, f16x8 (&xh)[2 * NG], i32x8 (&xl)[(NG + 1) / 2], i32x8 (&xx)[(NG + 1) / 2]) {
    i32x4 l[NG + 1], x[NG + 1];
#pragma unroll
    for (int g = 0; g < NG; ++g) make_ops16<RELU>([&](int i) { return get(16 * g + i); }, xh[2 * g], xh[2 * g + 1], l[g], x[g]);
    l[NG] = i32x4{0, 0, 0, 0};
    x[NG] = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < (NG + 1) / 2; ++b) {
        xl[b] = cat8(l[2 * b], l[2 * b + 1]);
        xx[b] = cat8(x[2 * b], x[2 * b + 1]);
    }
    // pin the conversions HERE: they are pure, and left alone the compiler sinks them (and the MFMAs that consume them)
    // past the fragment reads of the following layer phase, whose records then pile up in registers (238 spills)
#pragma unroll
    for (int c = 0; c < 2 * NG; ++c) asm volatile("" : "+v"(xh[c]));
#pragma unroll
    for (int b = 0; b < (NG + 1) / 2; ++b) asm volatile("" : "+v"(xl[b]), "+v"(xx[b]));
}

// block exponents four to a register: byte b % 4 of sh[b / 4] for the head block, of sl[b / 4] (= - 11) for the remainder
// block; unused bytes hold 12 so that the bytewise subtraction never borrows
template <int NB>
__device__ __forceinline__ void pack_exps(const int (&eb)[NB], int (&sh)[(NB + 3) / 4], int (&sl)[(NB + 3) / 4]) {
#pragma unroll
    for (int i = 0; i < (NB + 3) / 4; ++i) {
        int w = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) w |= (4 * i + k < NB ? eb[4 * i + k < NB ? 4 * i + k : 0] : 12) << (8 * k);
        sh[i] = w;
        sl[i] = w - 0x0b0b0b0b;
    }
}

// ---------------------------------------------------------------- one layer phase
// A finished accumulator tile pair is converted into the next layer's operands IN THE SHADOW of the following pair's
// MFMAs: 8 slices of 4 values (relu, fp16 head, remainder, two 8-bit packs: ~14 VALU) per tile, one slice behind each of
// the first 16 records of the next pair, fenced in place with sched_barrier (fillers placed like this are ~70 % hidden,
// anything left to the scheduler clusters after the MFMAs: profiles/r02_probe_filler.log).  Only the last pair's
// conversion stays exposed.
struct NextOps8 {  // operands of the next layer being assembled word by word
    unsigned h[64];
    int l[32], x[32];
};
struct NextOps6 {  // ... block by block: heads of pair b = K block b, its two bf6 forms, the exponents (byte b)
    u32x16 hv[4];
    i32x6 l6[4], x6[4];
    int sh, sl;
};
using NextOps = std::conditional_t<SIX, NextOps6, NextOps8>;
// Conversion of values 2 P, 2 P + 1 (P = 0..7) of a finished accumulator tile into their share of tile t's operand words,
// as TWO hand-written half-slices.  Each is one asm block: between separate asm statements and conversion builtins hipcc
// inserts a conservative `s_nop 0` per statement (it cannot see what the asm wrote), ~1000 issue slots per depth step.
//   half A: AGPR -> VGPR, relu, fp16 head pair            (5 instructions, 3 without relu)
//   half B: bf8 of the values, remainders, bf8 of the remainders * 2^12   (4 instructions)
// Hazards inside: the partial-register conversions (op_sel word writes) are not read within the next instruction; the
// accumulators read here were last written at least two MFMAs earlier (see layer_phase).
struct SliceRegs {
    float v0, v1;
    unsigned h;
};
template <bool RELU>
__device__ __forceinline__ void cv_half_a(float a0, float a1, SliceRegs &r) {
#ifdef F_ABL_NOCONV
    asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=&v"(r.v0), "=&v"(r.v1) : "a"(a0), "a"(a1));
    r.h = __float_as_uint(r.v0);
    return;
#endif
    if (RELU)
        asm volatile(
            "v_accvgpr_read_b32 %0, %3\n\t"
            "v_accvgpr_read_b32 %1, %4\n\t"
            "v_max_f32 %0, 0, %0\n\t"
            "v_max_f32 %1, 0, %1\n\t"
            "v_cvt_pk_f16_f32 %2, %0, %1"
            : "=&v"(r.v0), "=&v"(r.v1), "=&v"(r.h)
            : "a"(a0), "a"(a1));
    else
        asm volatile(
            "v_accvgpr_read_b32 %0, %3\n\t"
            "v_accvgpr_read_b32 %1, %4\n\t"
            "v_cvt_pk_f16_f32 %2, %0, %1"
            : "=&v"(r.v0), "=&v"(r.v1), "=&v"(r.h)
            : "a"(a0), "a"(a1));
}
// W = 0: low word of the two 8-bit operand registers (their high word is written later by the W = 1 half-slice)
template <int W>
__device__ __forceinline__ void cv_half_b(SliceRegs &r, int &l, int &x) {
#ifdef F_ABL_NOCONV
    l = __float_as_int(r.v1);
    x = __float_as_int(r.v0);
    return;
#endif
    const float inv = 1.0f / (float)(1 << LO_SHIFT);  // the scaled conversion DIVIDES by its scale operand
    if (W == 0)
        asm volatile(
            "v_cvt_pk_bf8_f32 %1, %2, %3\n\t"
            "v_fma_mix_f32 %2, %4, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %3, %4, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_cvt_scalef32_pk_bf8_f32 %0, %2, %3, %5"
            : "=&v"(l), "=&v"(x), "+v"(r.v0), "+v"(r.v1)
            : "v"(r.h), "s"(inv));
    else
        asm volatile(
            "v_cvt_pk_bf8_f32 %1, %2, %3 op_sel:[0,0,1]\n\t"
            "v_fma_mix_f32 %2, %4, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %3, %4, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_cvt_scalef32_pk_bf8_f32 %0, %2, %3, %5 op_sel:[0,0,0,1]"
            : "+v"(l), "+v"(x), "+v"(r.v0), "+v"(r.v1)
            : "v"(r.h), "s"(inv));
}
// six-bit variant of the second half-slice: fp16 head pair, running block maximum, the two remainders (4 instructions)
__device__ __forceinline__ void cv6_half_b(SliceRegs &r, float &m, float &r0, float &r1) {
#ifdef F_ABL_NOCONV
    r0 = r.v0;
    r1 = r.v1;
    return;
#endif
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_max3_f32 %1, %1, |%4|, |%5|\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(r.h), "+v"(m), "=&v"(r0), "=&v"(r1)
        : "v"(r.v0), "v"(r.v1));
}
// first half-slice without the head conversion (it moves to the second half in the six-bit variant)
template <bool RELU>
__device__ __forceinline__ void cv6_half_a(float a0, float a1, SliceRegs &r) {
    if (RELU)
        asm volatile(
            "v_accvgpr_read_b32 %0, %2\n\t"
            "v_accvgpr_read_b32 %1, %3\n\t"
            "v_max_f32 %0, 0, %0\n\t"
            "v_max_f32 %1, 0, %1"
            : "=&v"(r.v0), "=&v"(r.v1)
            : "a"(a0), "a"(a1));
    else
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=&v"(r.v0), "=&v"(r.v1) : "a"(a0), "a"(a1));
}

template <int NT8>
__device__ __forceinline__ void ops_from(const NextOps6 &o, f16x8 (&xh)[2 * NT8], i32x6 (&xl)[NT8 / 2], i32x6 (&xx)[NT8 / 2], int (&sh)[1],
                                         int (&sl)[1]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int c = 0; c < 2 * NT8; ++c)
        xh[c] = __builtin_bit_cast(f16x8, u32x4{o.hv[c / 4][4 * (c % 4)], o.hv[c / 4][4 * (c % 4) + 1], o.hv[c / 4][4 * (c % 4) + 2],
                                                o.hv[c / 4][4 * (c % 4) + 3]});
#pragma unroll
    for (int b = 0; b < NT8 / 2; ++b) {
        xl[b] = o.l6[b];
        xx[b] = o.x6[b];
    }
    sh[0] = o.sh;
    sl[0] = o.sl;
}
template <int NT8>
__device__ __forceinline__ void ops_from(const NextOps8 &o, f16x8 (&xh)[2 * NT8], i32x8 (&xl)[NT8 / 2], i32x8 (&xx)[NT8 / 2], int (&)[1], int (&)[1]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int c = 0; c < 2 * NT8; ++c) xh[c] = __builtin_bit_cast(f16x8, u32x4{o.h[4 * c], o.h[4 * c + 1], o.h[4 * c + 2], o.h[4 * c + 3]});
#pragma unroll
    for (int b = 0; b < NT8 / 2; ++b) {
        xl[b] = i32x8{o.l[8 * b], o.l[8 * b + 1], o.l[8 * b + 2], o.l[8 * b + 3], o.l[8 * b + 4], o.l[8 * b + 5], o.l[8 * b + 6], o.l[8 * b + 7]};
        xx[b] = i32x8{o.x[8 * b], o.x[8 * b + 1], o.x[8 * b + 2], o.x[8 * b + 3], o.x[8 * b + 4], o.x[8 * b + 5], o.x[8 * b + 6], o.x[8 * b + 7]};
    }
}

// acc[NT] (+)= W[:, K range of this phase] . X, X given as NBLK blocks (4 fp16 chunks + bf8 remainder + bf8 value each; the
// last block may carry only NCH_LAST chunks).  sc_h / sc_l: E8M0 scale operands of W_h / W_l for this layer.
// CV: 0 = leave the result in acc; 1 = relu and convert the finished tiles into the next layer's operands `out` (in flight, see above).
template <int REC0, int NT, int NBLK, int NCH_LAST, bool INIT, int CV = 0>
__device__ __forceinline__ void layer_phase(const FRing &rg, const float *bp, f32x16 (&acc)[NT], const f16x8 *xh, const XB *xl,
                                            const XB *xx, const int *xsh, const int *xsl, int sc_h, int sc_l, NextOps *out = nullptr,
                                            unsigned *trace = nullptr) {
    const int hi = rg.lane >> 5;
    constexpr int RPP = recs_per_pair(NBLK, NCH_LAST);
    constexpr int NREC = (NT / 2) * RPP;
    constexpr int SC_XL = 127 - LO_SHIFT, SC_ONE = 127;
    // In-flight conversion: the 32 values a lane holds of the PREVIOUS pair are converted in 32 half-slices (16 when only the
    // fp16 heads are needed) spread over ALL records of this pair, main and cross: the wave is issue-bound (about one
    // issue slot per 4 cycles: a 32-cycle fp16 MFMA hides ~5 other instructions, a 51-cycle 8-bit one ~10), and whole
    // slices behind the main records only (9 VALU + 2 fragment reads + wait + half a DMA piece per 2 MFMAs) overran that.
    constexpr int NMAIN = (NBLK - 1) * 4 + NCH_LAST;
    constexpr int NH = 32;                       // half-slices per pair
    constexpr int HPR = (NH + RPP - 1) / RPP;    // per record
    SliceRegs sr;
    // fragment reads run TWO records ahead of the MFMAs (one record = 64-100 matrix-pipe cycles, less than the LDS
    // latency under load)
#ifndef F_PF
#define F_PF 3  // records of lookahead (measured: 2 -> 21.4 ms, 3 -> 21.0, 4 -> 21.3, 5 -> 21.8)
#endif
#ifdef F_DUMMY_FILL
    float rg_dummy[8];
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t rg_dummy2[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) rg_dummy[i] = (float)(rg.lane + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) rg_dummy2[i] = f32x2_t{(float)(rg.lane + i), 1.f};
#endif
    Rec buf[F_PF + 1];
    static_for<F_PF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < NREC) load_rec<REC0 + i>(rg, buf[i]);
    });
    f32x16 c0, c1;
#if F_SIX
    f32x16 ra, rb;  // remainders of the pair being converted (even / odd tile)
    float mx = 0.f;  // its running max |value|
#endif
    auto half_slice = [&](auto tpc, auto qc) {  // half-slice q of the pair tpp
        constexpr int tpp = decltype(tpc)::value, q = decltype(qc)::value;
        constexpr int sl = q / 2, half = q % 2;
        constexpr int tt = sl / 8, t = 2 * tpp + tt, P = sl % 8;
#if F_SIX
        if constexpr (half == 0) {
            cv6_half_a<true>(acc[t][2 * P], acc[t][2 * P + 1], sr);
        } else {
            float r0, r1;
            cv6_half_b(sr, mx, r0, r1);
            out->hv[tpp][8 * tt + P] = sr.h;
            if constexpr (tt == 0) {
                ra[2 * P] = r0;
                ra[2 * P + 1] = r1;
            } else {
                rb[2 * P] = r0;
                rb[2 * P + 1] = r1;
            }
            if constexpr (q == NH - 1) {  // the pair = K block tpp of the next layer is complete: its two bf6 forms
                const int te = block_exponent(mx);
                cvt_block6(out->hv[tpp], ra, rb, te, out->x6[tpp], out->l6[tpp]);
                asm volatile("" : "+v"(out->x6[tpp]), "+v"(out->l6[tpp]));  // keep the (pure) conversions here
                out->sh = tpp == 0 ? te : (out->sh | (te << (8 * tpp)));
                if constexpr (tpp == NT / 2 - 1) out->sl = out->sh - 0x0b0b0b0b;  // every byte >= 12: no borrow
                mx = 0.f;
            }
        }
#else
        if constexpr (half == 0) {
            cv_half_a<true>(acc[t][2 * P], acc[t][2 * P + 1], sr);
            out->h[8 * t + P] = sr.h;
        } else {
            cv_half_b<(P & 1)>(sr, out->l[4 * t + P / 2], out->x[4 * t + P / 2]);
        }
#endif
    };
    static_for<NREC>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int tp = k / RPP, j0 = k % RPP;
        // a pair's records: ALL main records first (chunk m = j0), then the cross records block by block.  (Alternating them
        // block by block put an 8-pass fp16 and a 16-pass 8-bit MFMA back to back on the same accumulator eight times per
        // pair, and every such hand-over stalled the chain for hundreds of cycles: profiles/r02_march_f16_phases.md)
        constexpr bool is_main = j0 < NMAIN;
        constexpr int b = is_main ? (j0 / 4 < NBLK ? j0 / 4 : NBLK - 1) : (j0 - NMAIN) / 4;
        constexpr int xw = is_main ? 0 : (j0 - NMAIN) % 4;  // which cross record: X8h(t0), X8h(t1), X8l(t0), X8l(t1)
        if (j0 == 0) {
            if (INIT) {
                c0 = f_bias_tile(bp, 2 * tp, hi);
                c1 = f_bias_tile(bp, 2 * tp + 1, hi);
            } else {
                c0 = acc[2 * tp];
                c1 = acc[2 * tp + 1];
            }
        }
        Rec &cur = buf[k % (F_PF + 1)];
        if constexpr (k + F_PF < NREC) load_rec<REC0 + k + F_PF>(rg, buf[(k + F_PF) % (F_PF + 1)]);
#if F_SIX
        i32x8 cur8;
        if constexpr (!is_main) cur8 = cross6_operand(cur);
#endif
        if constexpr (k + F_PF < NREC) wait_rec<2 * F_PF>(cur);
        else wait_rec<2 * (NREC - 1 - k)>(cur);
        // half-slices [j0 * HPR, (j0 + 1) * HPR) of the previous pair ride on this record.  (Measured for the six-bit variant,
        // whose cross MFMAs are as short as one fp16 MFMA: slices on the main records only, one behind each of their two
        // MFMAs, is 0.2 ms SLOWER than this even spread, A/B on one box.)
        auto run_slice = [&]() {
            if constexpr (CV != 0 && tp > 0 && j0 * HPR < NH) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<HPR>([&](auto ic) {
                    constexpr int q = j0 * HPR + decltype(ic)::value;
                    if constexpr (q < NH) half_slice(std::integral_constant<int, tp - 1>{}, std::integral_constant<int, q>{});
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#ifdef F_DUMMY_FILL  // experiment: F_DUMMY_FILL independent VALU instructions behind every MFMA of fc_0 (kind F_DUMMY_KIND)
        auto dummy = [&rg_dummy, &rg_dummy2]() {
            if constexpr (REC0 == FR_F0) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<F_DUMMY_FILL>([&rg_dummy, &rg_dummy2](auto ic) {
                    constexpr int i = decltype(ic)::value;
#if F_DUMMY_KIND == 0
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(rg_dummy[i % 8]) : "v"(rg_dummy[(i + 3) % 8]));
#else
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(rg_dummy2[i % 4]) : "v"(rg_dummy2[(i + 1) % 4]));
#endif
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#else
        auto dummy = [&]() {};
#endif
        if constexpr (is_main) {
            c0 = mfma_main(cur.p0, xh[j0], c0);
            run_slice();
            dummy();
            c1 = mfma_main(cur.p1, xh[j0], c1);
            dummy();
        } else {
#if F_SIX
#ifndef F_ABL_NOXL
            if constexpr (xw == 0) c0 = mfma_cross6<b % 4>(cur8, xl[b], c0, xsl[b / 4]);
            else if constexpr (xw == 1) c1 = mfma_cross6<b % 4>(cur8, xl[b], c1, xsl[b / 4]);
#endif
#ifndef F_ABL_NOXH
            if constexpr (xw == 2) c0 = mfma_cross6<b % 4>(cur8, xx[b], c0, xsh[b / 4]);
            else if constexpr (xw == 3) c1 = mfma_cross6<b % 4>(cur8, xx[b], c1, xsh[b / 4]);
#endif
#else
            if constexpr (xw == 0) c0 = mfma_cross(cur, xl[b], c0, sc_h, SC_XL);
            else if constexpr (xw == 1) c1 = mfma_cross(cur, xl[b], c1, sc_h, SC_XL);
            else if constexpr (xw == 2) c0 = mfma_cross(cur, xx[b], c0, sc_l, SC_ONE);
            else c1 = mfma_cross(cur, xx[b], c1, sc_l, SC_ONE);
#endif
            run_slice();
            dummy();
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef F_TIMING
        if (trace) {
            const unsigned long long t__ = __builtin_readcyclecounter();
            if (rg.lane == 0) trace[k] = (unsigned)t__;
        }
#endif
        if (j0 == RPP - 1) {
            // pin the end of the pair's accumulator chains here: the MFMAs are pure, and code sinking otherwise moves the
            // tail of every chain into the block that first reads the tile (past the next level's gather), keeping the
            // records they read alive in registers
            asm volatile("" : "+a"(c0), "+a"(c1));
            acc[2 * tp] = c0;
            acc[2 * tp + 1] = c1;
        }
    });
    if constexpr (CV != 0) {  // the last pair: exposed
        // its first tile was last written by the second-last MFMA, 16 passes + the issue of the last one ago: the asm reads
        // below are invisible to hipcc's hazard recognizer (the XDL write -> VALU read distance is 19 wait states)
        asm volatile("s_nop 7" ::: "memory");
        static_for<NH>([&](auto qc) { half_slice(std::integral_constant<int, NT / 2 - 1>{}, qc); });
    }
}

// F_TIMING (experiment builds): wave 0 of workgroup 0 stamps s_memtime at the phase boundaries of every depth step into
// the `raw` output (tools/experiments/phase_times.py turns them into a cycle budget)
#ifdef F_TIMING
#define F_STAMP(i) do { if (tbuf) { const unsigned long long t__ = __builtin_readcyclecounter(); if (rg.lane == 0) tbuf[(i)] = (unsigned)t__; } } while (0)
#else
#define F_STAMP(i) do { } while (0)
#endif
// fc_0 operand slot (0..175) of the q-th gathered value of pyramid level L (see the stream geometry above)
__host__ __device__ constexpr int f0_slot(int L, int q) { return L == 1 ? q : (L == 2 ? 32 + q : (L == 3 ? 96 + q : 160 + q)); }

__device__ __forceinline__ void decode_f16(const SceneDev &sc, const FRing &rg, float px, float py, float pz, float vx, float vy, float vz,
                                           float (&pe)[N_PE], float (&out)[4], unsigned *tbuf, unsigned *tbuf0 = nullptr) {
    const int hi = rg.lane >> 5;
    const float *prm = reinterpret_cast<const float *>(rg.lds + RING_BYTES);
    const int *scl = reinterpret_cast<const int *>(prm + P_SC);
    f32x16 acc[8];
    f16x8 xh[16];
    XB xl[4], xx[4];
    int xsh[1] = {0}, xsl[1] = {0};  // block exponents of the activation operands (six-bit variant)
    NextOps nx;
    {
        // gather: the tile fetches of all four pyramid levels are issued first (32 coalesced 16-byte loads per lane in
        // flight together), then level after level goes registers -> LDS tile -> trilinear blend -> operands
        const GridCoord g = grid_coords(sc, px, py, pz);
        const WaveBox wb = wave_box(g);
        F_STAMP(1);
        f16x8 fh[24];
        XB fl[6], fx[6];
        int fsh[2] = {0, 0}, fsl[2] = {0, 0};
#if F_SIX
        int fe[6];
#endif
        // NG groups of 16 gathered values -> chunks C0.. and K blocks B0.. of fc_0's operand
        auto emit = [&](auto ngc, const float *f, auto c0c, auto b0c) {
            constexpr int NG = decltype(ngc)::value, C0 = decltype(c0c)::value, B0 = decltype(b0c)::value, NB = (NG + 1) / 2;
            f16x8 h[2 * NG];
            XB l[NB], x[NB];
#if F_SIX
            int eb[NB];
            make_operands6<NG, false>([&](int i) { return f[i]; }, h, l, x, eb);
#pragma unroll
            for (int q = 0; q < NB; ++q) fe[B0 + q] = eb[q];
#else
            make_operands<NG, false>([&](int i) { return f[i]; }, h, l, x);
#endif
#pragma unroll
            for (int c = 0; c < 2 * NG; ++c) fh[C0 + c] = h[c];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                fl[B0 + q] = l[q];
                fx[B0 + q] = x[q];
            }
        };
        using std::integral_constant;
#ifdef F_ABL_NOGATHER
        {
            float f[192];
#pragma unroll
            for (int i = 0; i < 192; ++i) f[i] = i < 176 ? g.gw * (float)(i + 1) + g.gh : 0.f;
            emit(integral_constant<int, 12>{}, f, integral_constant<int, 0>{}, integral_constant<int, 0>{});
        }
#else
        CoopFetch<TILE_BYTES> c0f, c1f, c2f, c3f;
        coop_issue<0, TILE_BYTES>(sc, wb, rg.lane, c0f);
        coop_issue<1, TILE_BYTES>(sc, wb, rg.lane, c1f);
        coop_issue<2, TILE_BYTES>(sc, wb, rg.lane, c2f);
        coop_issue<3, TILE_BYTES>(sc, wb, rg.lane, c3f);
        F_STAMP(2);
        {
            // level 0 fills the first half of the last block (slots 160..175), the second half is zero
            float f0[16];
            coop_finish<0, TILE_BYTES>(sc, g, c0f, hi, rg.lane, rg.tile, f0);
            emit(integral_constant<int, 1>{}, f0, integral_constant<int, 20>{}, integral_constant<int, 5>{});
        }
        F_STAMP(3);
        {
            float f1[32];
            coop_finish<1, TILE_BYTES>(sc, g, c1f, hi, rg.lane, rg.tile, f1);
            emit(integral_constant<int, 2>{}, f1, integral_constant<int, 0>{}, integral_constant<int, 0>{});
        }
        F_STAMP(4);
        {
            float f2[64];
            coop_finish<2, TILE_BYTES>(sc, g, c2f, hi, rg.lane, rg.tile, f2);
            emit(integral_constant<int, 4>{}, f2, integral_constant<int, 4>{}, integral_constant<int, 1>{});
        }
        F_STAMP(5);
        {
            float f3[64];
            coop_finish<3, TILE_BYTES>(sc, g, c3f, hi, rg.lane, rg.tile, f3);
            emit(integral_constant<int, 4>{}, f3, integral_constant<int, 12>{}, integral_constant<int, 3>{});
        }
#endif
#if F_SIX
        pack_exps<6>(fe, fsh, fsl);
#endif
        F_STAMP(6);
        // fc_0: 6 K-blocks (22 chunks), finished tiles converted (relu) into fc_1's operands on the fly
        layer_phase<FR_F0, 8, 6, 2, true, 1>(rg, prm + P_B0, acc, fh, fl, fx, fsh, fsl, scl[0], scl[1], &nx);
        F_STAMP(13);
    }
    ops_from<8>(nx, xh, xl, xx, xsh, xsl);
#ifdef F_TIMING  // per-record trace of fc_1
    layer_phase<FR_L1, 8, 4, 4, true, 1>(rg, prm + P_B1, acc, xh, xl, xx, xsh, xsl, scl[2], scl[3], &nx, tbuf ? tbuf0 + 8192 + 128 * ((tbuf - tbuf0) / 32) : nullptr);
#else
    layer_phase<FR_L1, 8, 4, 4, true, 1>(rg, prm + P_B1, acc, xh, xl, xx, xsh, xsl, scl[2], scl[3], &nx);
#endif
    F_STAMP(14);
    ops_from<8>(nx, xh, xl, xx, xsh, xsl);
    // fc_2.  alpha_fc (fp32, VALU) is NOT folded into the conversion slices: its weights come from LDS, and a
    // compiler-visible LDS read inside the record loop makes hipcc wait lgkmcnt(0), which drains the fragment prefetch
    // at every slice (measured with F_TIMING: fc_2 took 22.5k cycles against 13.3k for the identical fc_1).  It runs on
    // the finished accumulators before the tail conversion overwrites nothing it needs (acc stays intact).
    layer_phase<FR_L2, 8, 4, 4, true, 1>(rg, prm + P_B2, acc, xh, xl, xx, xsh, xsl, scl[4], scl[5], &nx);
    float s_alpha = 0.f;
    {
        const f32x4 *aw = reinterpret_cast<const f32x4 *>(prm + P_AW + hi * 128);
#pragma unroll
        for (int q4 = 0; q4 < 32; ++q4) {
            const f32x4 w = aw[q4];
            relu_fma(s_alpha, w.x, acc[q4 >> 2][(q4 & 3) * 4 + 0]);
            relu_fma(s_alpha, w.y, acc[q4 >> 2][(q4 & 3) * 4 + 1]);
            relu_fma(s_alpha, w.z, acc[q4 >> 2][(q4 & 3) * 4 + 2]);
            relu_fma(s_alpha, w.w, acc[q4 >> 2][(q4 & 3) * 4 + 3]);
        }
    }
    out[3] = add_halves(s_alpha) + prm[P_AB];
    F_STAMP(15);
    ops_from<8>(nx, xh, xl, xx, xsh, xsl);
    F_STAMP(16);
    // the colour head's linear part as ONE layer in two K phases: the folded feature_fc / latent_fc / view_fc product over
    // fc_2's 256 outputs (bias: the second block of nb_mlp_latent_bias), then view_fc over the positional encodings; rgb_fc
    // (fp32, VALU) follows on the finished tiles
    f32x16 v[4];
    layer_phase<FR_VG, 4, 4, 4, true>(rg, prm + P_LB, v, xh, xl, xx, xsh, xsl, scl[6], scl[7]);
    F_STAMP(17);
    float s_rgb[3] = {0.f, 0.f, 0.f};
    {
#ifdef F_ABL_NOPE
#pragma unroll
        for (int c = 12; c < N_PE; ++c) pe[c] = px * (float)c;
#else
        pe_xyz(pe, px, py, pz, vx, vy, vz, hi);
#endif
        f16x8 ph[6];
        XB pl[2], pxx[2];
        int psh[1] = {0}, psl[1] = {0};
#if F_SIX
        int pee[2];
        make_operands6<3, false>([&](int i) { return i < N_PE ? pe[i < N_PE ? i : 0] : 0.f; }, ph, pl, pxx, pee);
        pack_exps<2>(pee, psh, psl);
#else
        make_operands<3, false>([&](int i) { return i < N_PE ? pe[i < N_PE ? i : 0] : 0.f; }, ph, pl, pxx);
#endif
        F_STAMP(18);
        layer_phase<FR_VP, 4, 2, 2, false>(rg, prm + P_LB, v, ph, pl, pxx, psh, psl, scl[6], scl[7]);
    }
    // rgb_fc in fp32 on the VALU (outside the record loop for the same reason as alpha_fc)
    {
        const f32x4 *rw0 = reinterpret_cast<const f32x4 *>(prm + P_RW + (0 * 2 + hi) * 64);
        const f32x4 *rw1 = reinterpret_cast<const f32x4 *>(prm + P_RW + (1 * 2 + hi) * 64);
        const f32x4 *rw2 = reinterpret_cast<const f32x4 *>(prm + P_RW + (2 * 2 + hi) * 64);
#pragma unroll
        for (int q4 = 0; q4 < 16; ++q4) {
            const f32x4 w0 = rw0[q4], w1 = rw1[q4], w2 = rw2[q4];
#pragma unroll
            for (int e = 0; e < 4; ++e) relu_fma3(s_rgb[0], s_rgb[1], s_rgb[2], w0[e], w1[e], w2[e], v[q4 >> 2][(q4 & 3) * 4 + e]);
        }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out[ch] = add_halves(s_rgb[ch]) + prm[P_RB + ch];
#if !defined(F_DMA_BURST) && !defined(F_ABL_NODMA)
    // the stream is zero-padded to a whole number of ring rounds: the DMA pieces that would ride on the padding records
    static_for<FN_RECS_PAD - FN_RECS>([&](auto ic) {
        constexpr int REC = FN_RECS + decltype(ic)::value;
        if constexpr ((REC % FPAGE_RECS) % 2 == 1 && (REC % FPAGE_RECS) / 2 < FDMA_PER_WAVE)
            f_issue_piece<(REC % FPAGE_RECS) / 2>(rg, (REC / FPAGE_RECS + FAHEAD) % FN_PAGES);
    });
#endif
    F_STAMP(19);
}

__device__ __forceinline__ FRing f_ring_begin(const float *pk, const float *lb, const char *stream, char *lds) {
    {
        float *prm = reinterpret_cast<float *>(lds + RING_BYTES);
        const int *scales = reinterpret_cast<const int *>(stream + (size_t)FN_RECS_PAD * FREC_BYTES);
        for (int i = threadIdx.x; i < P_SIZE; i += 256) {
            float v;
            if (i < P_B1) v = pk[F_OFF_B0 + i - P_B0];
            else if (i < P_B2) v = pk[F_OFF_B1 + i - P_B1];
            else if (i < P_LB) v = pk[F_OFF_B2 + i - P_B2];
            else if (i < P_BV) v = i - P_LB < 128 ? lb[256 + i - P_LB] : 0.f;  // bias of the folded view layer
            else if (i < P_AW) v = pk[F_OFF_BV + i - P_BV];
            else if (i < P_RW) v = pk[F_OFF_AW + i - P_AW];
            else if (i < P_AB) v = pk[F_OFF_RW + i - P_RW];
            else if (i < P_RB) v = pk[F_OFF_AB + i - P_AB];
            else if (i < P_SC) v = pk[F_OFF_RB + i - P_RB];
            else v = __int_as_float(scales[i - P_SC]);
            prm[i] = v;
        }
        __syncthreads();
    }
    FRing rg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    rg.wave_off = wave * (FDMA_PER_WAVE * 1024);
    rg.stream = stream + rg.wave_off;
    rg.lds = lds;
    rg.lane = threadIdx.x & 63;
    rg.tile = lds + RING_BYTES + PARAM_BYTES + wave * TILE_BYTES;
#pragma unroll
    for (int sl = 0; sl < FN_SLOTS; ++sl) {
        rg.base[sl] = rg.lane * 16 + sl * FPAGE_BYTES;
        asm volatile("" : "+v"(rg.base[sl]));
    }
#pragma unroll
    for (int p = 0; p < FAHEAD; ++p) f_issue_page(rg, p);
    return rg;
}

// ---------------------------------------------------------------- ray mode
#if F_SIX
#define F_KERNEL nb_march_f6_kernel
#define F_KERNEL_NAME "nb_march_f6_kernel"
#else
#define F_KERNEL nb_march_f16_kernel
#define F_KERNEL_NAME "nb_march_f16_kernel"
#endif
// FULL = false: no sample culling and no `raw` output (the plain renderer: what bench.py and run.py's evaluation loop launch).
// The two rarely used paths cost 30 SGPRs that live across the depth loop; without them the kernel spills 43 instead of 72
// scalars and restores 60 instead of 232 per depth step (v_readlane = VALU issue slots, which this kernel is short of).
template <bool FULL>
__global__ __launch_bounds__(256) void F_KERNEL(MarchArgs a, const char *stream) {
    nbm::saturate_fp16_conversions();
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const FRing rg = f_ring_begin(a.pk, a.lb, stream, lds);
    const int lane = rg.lane, j = lane & 31, hi = lane >> 5;
    const int grp = xcd_remap(blockIdx.x, a.n_wave_groups);
    const long long wave = (long long)grp * 4 + rg.wave_off / (FDMA_PER_WAVE * 1024);
    long long ray = wave * 32 + j;
    const bool valid = ray < a.n_rays;
    if (!valid) ray = a.n_rays - 1;
    if (a.ray_order) ray = a.ray_order[ray];
    const int S = a.n_samples;
    const float ox = a.ray_o[ray * 3 + 0], oy = a.ray_o[ray * 3 + 1], oz = a.ray_o[ray * 3 + 2];
    const float dx = a.ray_d[ray * 3 + 0], dy = a.ray_d[ray * 3 + 1], dz = a.ray_d[ray * 3 + 2];
    const float near = a.near[ray], far = a.far[ray];
    const float dn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    const float vx = dx / dn, vy = dy / dn, vz = dz / dn;
    float pe[N_PE];
    pe_view(pe, vx, vy, vz, hi);
    const float *tr = a.t_rand ? a.t_rand + ray * S : nullptr;

    auto z_at = [&](int s) -> float {
        const float zc = z_lin(near, far, a.t_vals[s]);
        if (!tr) return zc;
        const float lower = s == 0 ? zc : 0.5f * __fadd_rn(zc, z_lin(near, far, a.t_vals[s - 1]));
        const float upper = s == S - 1 ? zc : 0.5f * __fadd_rn(z_lin(near, far, a.t_vals[s + 1]), zc);
        return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), tr[s]));
    };

    RayAccum ra;
    WeightStore wstore;
    float z_cur = z_at(0);
    for (int s = 0; s < S; ++s) {
        // see nb_march_bf16.hip: a compiler-visible vmcnt(0) at the top of the step (the pages in flight were requested a
        // whole step ago) so that the gather's ordinary loads get counted waits; loop-invariant roots are laundered
        __builtin_amdgcn_s_waitcnt(0x0F70);
        const float z_next = (s + 1 < S) ? z_at(s + 1) : 0.f;
        const float px = __fadd_rn(ox, __fmul_rn(dx, z_cur));
        const float py = __fadd_rn(oy, __fmul_rn(dy, z_cur));
        const float pz = __fadd_rn(oz, __fmul_rn(dz, z_cur));
        float out[4];
        int zero = 0, lane_i = rg.lane;
        asm volatile("" : "+s"(zero), "+v"(lane_i));
        FRing r2;
        r2.lds = rg.lds;
        r2.lane = lane_i;
        r2.stream = rg.stream + zero;
        r2.wave_off = rg.wave_off + zero;
        r2.tile = rg.tile + zero;
#pragma unroll
        for (int sl = 0; sl < FN_SLOTS; ++sl) {
            r2.base[sl] = lane_i * 16 + sl * FPAGE_BYTES;
            asm volatile("" : "+v"(r2.base[sl]));
        }
        // sample culling (nb_cull): a WORKGROUP decision, the four waves walk the weight ring in lock step
        bool ins = true, run = true;
        if (FULL && a.cull.n_views) {
            ins = cull_inside(a.cull, a.sc, px, py, pz);
            int *flags = reinterpret_cast<int *>(rg.lds + RING_BYTES) + P_SIZE;
            const int any_wave = __any(ins) ? 1 : 0;
            if (lane_i == 0) flags[rg.wave_off / (FDMA_PER_WAVE * 1024)] = any_wave;
            __syncthreads();
            run = __builtin_amdgcn_readfirstlane(flags[0] | flags[1] | flags[2] | flags[3]) != 0;
            __syncthreads();
        }
#ifdef F_TIMING
        unsigned *tbuf = (FULL && blockIdx.x == 0 && rg.wave_off == 0 && a.raw) ? reinterpret_cast<unsigned *>(a.raw) + 32 * s : nullptr;
        if (tbuf && lane_i == 0) tbuf[0] = (unsigned)__builtin_readcyclecounter();
#else
        unsigned *tbuf = nullptr;
#endif
        if (run) decode_f16(a.sc, r2, px, py, pz, vx, vy, vz, pe, out, tbuf, FULL ? reinterpret_cast<unsigned *>(a.raw) : nullptr);
        if (!ins || !run) out[0] = out[1] = out[2] = out[3] = 0.f;
        float dist = (s + 1 < S) ? __fsub_rn(z_next, z_cur) : 1e10f;
        dist = __fmul_rn(dist, dn);
        const float w = ra.add(out, z_cur, dist);
        wstore.push(a, ray, s, S, hi, valid, w);
#ifdef F_TIMING
        if (tbuf && lane_i == 0) tbuf[20] = (unsigned)__builtin_readcyclecounter();
#else
        if (FULL && valid && hi == 0 && a.raw)
            *reinterpret_cast<f32x4 *>(a.raw + (ray * S + s) * 4) = f32x4{out[0], out[1], out[2], out[3]};
#endif
        z_cur = z_next;
    }
    if (valid && hi == 0) ra.store(a, ray);
    // pages prefetched past the end of the work are still in flight: let them land before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
}

// ---------------------------------------------------------------- weight stream packing
// weight of layer phase `ph` at (output row, operand element q of a lane with half index kg); 0 for padding
// phases: 0 fc_0, 1 fc_1, 2 fc_2, 3 the folded colour-head layer over fc_2's outputs, 4 view_fc over the encodings
__device__ __forceinline__ float phase_weight(const nb_mlp_params &p, const float *f32_blob, int ph, int row, int q, int kg) {
    if (ph == 0) {  // fc_0: slots [level 1 | level 2 | level 3 | level 0 | zeros]
        int L, idx;
        if (q < 32) { L = 1; idx = q; }
        else if (q < 96) { L = 2; idx = q - 32; }
        else if (q < 160) { L = 3; idx = q - 96; }
        else if (q < 176) { L = 0; idx = q - 160; }
        else return 0.f;
        return p.fc0_w[row * 352 + col_feat(lvl_reg_base(L) + idx, kg)];
    }
    if (ph == 1) return p.fc1_w[row * 256 + col_hidden(q, kg)];
    if (ph == 2) return p.fc2_w[row * 256 + col_hidden(q, kg)];
    if (ph == 3) {
        // view_w[:, :256] . (latent_w[:, :256] . feature_w): the inner product comes from the fp32 section (formed in fp64
        // there, fragment order: invert col_hidden), the outer one is summed in fp64 here
        const int col = col_hidden(q, kg);
        const int tt = col >> 5, rr = col & 31, hi2 = (rr >> 2) & 1, r2 = (rr & 3) + 4 * (rr >> 3), q2 = 16 * tt + r2;
        double s = 0.0;
        for (int m = 0; m < 256; ++m)
            s += (double)p.view_w[row * 346 + m] *
                 (double)f32_blob[F_OFF_L4 + (((m >> 5) * 32 + (q2 >> 2)) * 64 + (hi2 * 32 + (m & 31))) * 4 + (q2 & 3)];
        return (float)s;
    }
    const int col = col_pe(q, kg);  // q >= 45 -> -1
    return col < 0 ? 0.f : p.view_w[row * 346 + col];
}
__device__ __forceinline__ int phase_layer(int ph) { return ph < 3 ? ph : 3; }
constexpr int N_PHASES = 5, N_LAYERS = 4;

struct PhaseGeom {
    int rec0, nt, nblk, nch_last;
};
__device__ __forceinline__ PhaseGeom phase_geom(int ph) {
    switch (ph) {
        case 0: return {FR_F0, 8, 6, 2};
        case 1: return {FR_L1, 8, 4, 4};
        case 2: return {FR_L2, 8, 4, 4};
        case 3: return {FR_VG, 4, 4, 4};
        default: return {FR_VP, 4, 2, 2};
    }
}

// per layer: max |W_h| and max |W_l| -> largest power-of-two scales that keep the fp8 images inside +-448, stored as the
// E8M0 operands (127 - exponent) the scaled MFMA needs to undo them; out[2 * layer] for W_h, out[2 * layer + 1] for W_l
__global__ void nb_f16_scales_kernel(nb_mlp_params p, const float *__restrict__ f32_blob, int *__restrict__ out) {
    const int layer = blockIdx.x;  // fc_0, fc_1, fc_2, the folded view layer
    __shared__ float mh[256], ml[256];
    float a = 0.f, b = 0.f;
    const int ph0 = layer, ph1 = layer < 3 ? layer + 1 : N_PHASES;  // view layer = phases 3 and 4
    for (int ph = ph0; ph < ph1; ++ph) {
        const PhaseGeom g = phase_geom(ph);
        const int rows = 32 * g.nt, nq = 32 * g.nblk;
        for (int e = threadIdx.x; e < rows * nq * 2; e += blockDim.x) {
            const int kg = e & 1, q = (e >> 1) % nq, row = (e >> 1) / nq;
            const float w = phase_weight(p, f32_blob, ph, row, q, kg);
            const float h = (float)(_Float16)w;
            a = fmaxf(a, fabsf(h));
            b = fmaxf(b, fabsf(w - h));
        }
    }
    mh[threadIdx.x] = a;
    ml[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            mh[threadIdx.x] = fmaxf(mh[threadIdx.x], mh[threadIdx.x + s]);
            ml[threadIdx.x] = fmaxf(ml[threadIdx.x], ml[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) {
        const float m = threadIdx.x == 0 ? mh[0] : ml[0];
        int e = 0;
        if (m > 0.f && m < 3.0e38f) {
            e = ilogbf(448.f / m);          // 2^e <= 448 / m
            e = min(max(e, -100), 100);
            if (ldexpf(m, e) > 448.f) --e;  // guard the rounding of the division
        }
        out[2 * layer + threadIdx.x] = 127 - e;
    }
}

__device__ __forceinline__ unsigned fp8_e4m3_bits(float v) {
    v = fminf(fmaxf(v, -448.f), 448.f);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xffu;
}

// one thread per (record, lane): the lane's 32 bytes of the record
__global__ void nb_pack_f16_kernel(nb_mlp_params p, const float *__restrict__ f32_blob, const int *__restrict__ scales,
                                   int *__restrict__ stats, unsigned *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FN_RECS_PAD * 64) return;
    const int rec = t >> 6, lane = t & 63, i = lane & 31, kg = lane >> 5;
    if (rec >= FN_RECS) {  // padding records: never read, zero for determinism
        unsigned *padp = out + (size_t)rec * (FREC_BYTES / 4);
        for (int k = 0; k < 4; ++k) padp[lane * 4 + k] = padp[256 + lane * 4 + k] = 0u;
        return;
    }
    int ph = N_PHASES - 1;
    for (int q = 0; q + 1 < N_PHASES; ++q)
        if (rec < phase_geom(q + 1).rec0) {
            ph = q;
            break;
        }
    const PhaseGeom g = phase_geom(ph);
    const int rpp = recs_per_pair(g.nblk, g.nch_last);
    const int rel = rec - g.rec0, tp = rel / rpp, j0 = rel % rpp;
    const int nmain = (g.nblk - 1) * 4 + g.nch_last;
    const bool is_main = j0 < nmain;
    const int b = is_main ? j0 / 4 : (j0 - nmain) / 4;  // main record j0 = chunk j0 of the phase; then 4 cross records per block
    unsigned w32[8];
    if (is_main) {  // main record: A16(c, t0) | A16(c, t1)
        const int c = j0;
        for (int half = 0; half < 2; ++half) {
            const int row = 32 * (2 * tp + half) + i;
            for (int r = 0; r < 8; r += 2) {
                const f16x2 hp = {(_Float16)phase_weight(p, f32_blob, ph, row, 8 * c + r, kg),
                                  (_Float16)phase_weight(p, f32_blob, ph, row, 8 * c + r + 1, kg)};
                w32[half * 4 + r / 2] = __builtin_bit_cast(unsigned, hp);
            }
        }
    } else {  // X8h(t0), X8h(t1), X8l(t0), X8l(t1): 32 8-bit (6-bit) values of K-block b
        const int which = (j0 - nmain) % 4, row = 32 * (2 * tp + (which & 1)) + i, lo = which >> 1;
#if F_SIX
        // fp6 e2m3 with the lane's own E8M0 scale (row, K half).  The W_h record multiplies the REMAINDER operand, whose
        // elements are interleaved [v0, v16, v1, v17, ...] (v_cvt_scalef32_2xpk16_bf6_f32); the W_l record the head operand
        float wv[32], amax = 0.f;
        for (int e = 0; e < 32; ++e) {
            const int n = lo ? e : 16 * (e & 1) + (e >> 1);
            const float w = phase_weight(p, f32_blob, ph, row, 32 * b + n, kg);
            const float h = (float)(_Float16)w;
            wv[e] = lo ? w - h : h;
            amax = fmaxf(amax, fabsf(wv[e]));
        }
        int ex = 0;
        if (amax > 0.f && amax < 3.0e38f) {
            ex = ilogbf(amax / 7.5f);
            if (ldexpf(7.5f, ex) < amax) ++ex;  // the smallest power of two with max / 2^ex <= 7.5
            ex = min(max(ex, -120), 120);
        }
        for (int k = 0; k < 8; ++k) w32[k] = 0u;
        for (int e = 0; e < 32; ++e) {
            const unsigned code = fp6_e2m3_bits(ldexpf(wv[e], -ex));
            const int bit = 6 * e;
            w32[bit >> 5] |= code << (bit & 31);
            if ((bit & 31) > 26) w32[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
        }
        w32[6] = (unsigned)(127 + ex);
        // statistic behind nb_mlp_six_bit_stats_offset(): how many non-zero head weights sit below 1/8 of their block's
        // maximum, i.e. in e2m3's subnormal range where they keep fewer than 3 bits (per layer: small, non-zero)
        if (!lo) {
            int small = 0, nz = 0;
            for (int e = 0; e < 32; ++e) {
                nz += wv[e] != 0.f;
                small += wv[e] != 0.f && fabsf(wv[e]) < 0.125f * amax;
            }
            atomicAdd(&stats[2 * phase_layer(ph)], small);
            atomicAdd(&stats[2 * phase_layer(ph) + 1], nz);
        }
#else
        const int e8 = scales[2 * phase_layer(ph) + lo];  // 127 - exponent
        const float mul = ldexpf(1.f, 127 - e8);
        for (int e = 0; e < 32; e += 4) {
            unsigned word = 0;
            for (int k = 0; k < 4; ++k) {
                const float w = phase_weight(p, f32_blob, ph, row, 32 * b + e + k, kg);
                const float h = (float)(_Float16)w;
                word |= fp8_e4m3_bits((lo ? w - h : h) * mul) << (8 * k);
            }
            w32[e / 4] = word;
        }
#endif
    }
    // piece 0 (bytes 0..15 of the lane) at lane * 16, piece 1 at 1024 + lane * 16
    unsigned *recp = out + (size_t)rec * (FREC_BYTES / 4);
    for (int k = 0; k < 4; ++k) {
        recp[lane * 4 + k] = w32[k];
        recp[256 + lane * 4 + k] = w32[4 + k];
    }
}

}  // namespace

namespace nbm {

#if F_SIX
long long f6_stream_floats() { return ((long long)FN_RECS_PAD * FREC_BYTES + F_N_SCALES * 4) / 4; }
#else
long long f16_stream_floats() { return ((long long)FN_RECS_PAD * FREC_BYTES + F_N_SCALES * 4) / 4; }
#endif

// `packed` = [fp32 section][bf16 ring stream][M-split stream][f16f8 stream | scales][f16f6 stream | unused scales]
#if F_SIX
int pack_f6_stream(const nb_mlp_params *p, float *packed, long long stream_off, hipStream_t st) {
#else
int pack_f16_stream(const nb_mlp_params *p, float *packed, long long stream_off, hipStream_t st) {
#endif
    unsigned *stream = reinterpret_cast<unsigned *>(packed + stream_off);
    int *scales = reinterpret_cast<int *>(stream + (size_t)FN_RECS_PAD * FREC_BYTES / 4);
#if F_SIX
    // the 16 words behind the stream carry no layer scales in this variant (every lane of a record has its own) but the
    // small-element statistic the pack kernel accumulates
    NB_REQUIRE(hipMemsetAsync(scales, 0, F_N_SCALES * sizeof(int), st) == hipSuccess, "pack_f6_stream: hipMemsetAsync failed");
#else
    hipLaunchKernelGGL(nb_f16_scales_kernel, dim3(N_LAYERS), dim3(256), 0, st, *p, packed, scales);
    NB_CHECK_LAUNCH("nb_f16_scales_kernel");
#endif
    hipLaunchKernelGGL(nb_pack_f16_kernel, dim3(nb_ceil_div((long long)FN_RECS_PAD * 64, 256)), dim3(256), 0, st, *p, packed, scales, scales, stream);
    NB_CHECK_LAUNCH("nb_pack_f16_kernel");
    return NB_OK;
}

#if F_SIX
int launch_march_f6(const MarchArgs &a, long long stream_off, hipStream_t st) {
#else
int launch_march_f16(const MarchArgs &a, long long stream_off, hipStream_t st) {
#endif
    const char *stream = reinterpret_cast<const char *>(a.pk + stream_off);
    if (a.cull.n_views || a.raw) hipLaunchKernelGGL(F_KERNEL<true>, dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    else hipLaunchKernelGGL(F_KERNEL<false>, dim3(a.n_wave_groups), dim3(256), 0, st, a, stream);
    NB_CHECK_LAUNCH(F_KERNEL_NAME);
    return NB_OK;
}

}  // namespace nbm
