// nb_march_bf16.hip — split-bf16 ("bf16x3") decode / march kernels for gfx950 (MI355X).
//
// Same per-wave organisation as nb_march.hip (one wave = 32 sample columns, activations resident
// in registers, layers computed transposed so the MFMA C/D fragment of layer l is the B operand of
// layer l+1), but the GEMMs run on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA
// rate) with BOTH operands split into bf16 hi + lo parts and three products per K-chunk
//     W.X  ~=  W_hi.X_hi + W_hi.X_lo + W_lo.X_hi          (fp32 accumulate; the dropped W_lo.X_lo
// term is 2^-16 relative), which keeps the renderer inside the 1e-4 RGB parity budget where a
// single bf16 product does not (1.8e-4 single, 9e-7 split).
//
// Weight traffic: a wave consumes 2 KiB of weight fragments (A_hi + A_lo) per 3 MFMAs — 85 B/clk/CU,
// more than the 64 B/clk vector L1 can deliver — so the weight stream goes through LDS: the four
// waves of a workgroup march in lock step through ONE linear stream of 648 fragment records
// (1.33 MB per depth step, identical every step), DMA'd by global_load_lds into a 5-slot ring of
// 20-KiB pages (4 pages in flight ahead of the consumer) and read back with conflict-free
// ds_read_b128.  One s_barrier per page (30 MFMAs) both publishes the landed page and retires the
// slot that is refilled next; DMA completion is tracked with counted s_waitcnt vmcnt, never 0.
#include <type_traits>

#include "nb_march_common.h"

using namespace nbm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NB_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

namespace nbm {

// --------------------------------------------------------------- weight stream geometry
constexpr int REC_BYTES = 2048;  // one K=16 chunk of one 32-row tile: A_hi (1 KiB) + A_lo (1 KiB)
// ring geometry (measured: 10-record pages x 5 slots 30.45 ms vs 20-record pages x 3 slots 30.85 ms per image)
#ifndef NB_PAGE_RECS
#define NB_PAGE_RECS 10
#define NB_N_SLOTS 5
#define NB_N_RECS_PAD 650
#endif
constexpr int PAGE_RECS = NB_PAGE_RECS;
constexpr int PAGE_BYTES = PAGE_RECS * REC_BYTES;  // 20 KiB
constexpr int N_SLOTS = NB_N_SLOTS;
constexpr int AHEAD = N_SLOTS - 1;  // pages in flight ahead of the page being consumed
constexpr int NC0 = 22, NCH = 16, NCV = 22;
constexpr int REC_L0 = 0;
constexpr int REC_L1 = REC_L0 + 8 * NC0;
constexpr int REC_L2 = REC_L1 + 8 * NCH;
constexpr int REC_L4 = REC_L2 + 8 * NCH;
constexpr int REC_LV = REC_L4 + 8 * NCH;
constexpr int N_RECS = REC_LV + 4 * NCV;  // 648 records carry weights
constexpr int N_RECS_PAD = NB_N_RECS_PAD;  // + zero records so that the page count is a multiple of the ring size
constexpr int N_PAGES = N_RECS_PAD / PAGE_RECS;  // 65
static_assert(N_RECS_PAD % PAGE_RECS == 0 && N_RECS_PAD >= N_RECS, "stream must be a whole number of pages");
static_assert(N_PAGES % N_SLOTS == 0, "page p must always land in slot p % N_SLOTS, also across the step wrap-around");
static_assert(PAGE_RECS % 2 == 0, "record pairs must not straddle pages");

constexpr int DMA_PER_WAVE = PAGE_BYTES / 1024 / 4;  // 5 one-KiB pieces per wave per page

}  // namespace nbm

namespace {

// small fp32 parameters are shared with the f32 blob (same accumulator layout): offsets in floats
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

// small fp32 parameters staged in LDS behind the ring (ordinary global loads next to in-flight LDS-DMA
// make hipcc wait vmcnt(0), which would drain the weight pipeline at every tile): offsets in floats
constexpr int P_B0 = 0, P_B1 = 256, P_B2 = 512, P_LB = 768, P_BV = 1024, P_AW = 1152, P_RW = 1408, P_AB = 1792,
              P_RB = 1796, P_SIZE = 1800;
constexpr int RING_BYTES = N_SLOTS * PAGE_BYTES;
constexpr int PARAM_BYTES = 8192;
constexpr int TILE_BYTES = 8192;  // per-wave voxel tile of the cooperative gather
constexpr int LDS_BYTES = RING_BYTES + PARAM_BYTES + 4 * TILE_BYTES;
static_assert(LDS_BYTES <= 163840, "LDS budget (160 KiB per workgroup)");
static_assert((P_SIZE + 4) * 4 <= 8192, "parameter region overflow (4 ints of cull flags follow the parameters)");

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

struct Ring {
    const char *stream;  // global: N_RECS * REC_BYTES bytes, already advanced to this wave's share (uniform)
    char *lds;           // [ring N_SLOTS * PAGE_BYTES][params][4 x voxel tile]
    int wave_off;        // byte offset of this wave's share inside a page (uniform)
    int lane;
    char *tile;          // this wave's voxel tile
    int base[N_SLOTS];   // lane * 16 + slot * PAGE_BYTES: opaque byte offsets of the ring slots
};

// All address arithmetic below is wave-uniform (SGPR) except the single lane * 16 term, so every DMA
// is `global_load_lds_dwordx4 v_lane16, s[base]` — no per-DMA 64-bit VGPR address to keep alive.
__device__ __forceinline__ void issue_piece(const Ring &rg, int page, int i) {
    const int slot = page % N_SLOTS;
    const char *src = rg.stream + (size_t)page * PAGE_BYTES + i * 1024 + rg.lane * 16;
    char *dst = rg.lds + slot * PAGE_BYTES + rg.wave_off + i * 1024;  // wave-uniform; the DMA adds lane * 16
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

__device__ __forceinline__ void issue_page(const Ring &rg, int page) {
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) issue_piece(rg, page, i);
}

// Called before the first record of `page` is read.  My DMAs issued after that page's are those of
// pages page+1 .. page+AHEAD-1: allow exactly that many to stay in flight, then rendezvous so that
// every wave's share has landed (and every wave is done with the slot that gets refilled), then refill
// that slot with page + AHEAD in one burst.  (Measured alternative, -DNB_DMA_SPREAD: one DMA piece per
// record pair instead of a burst per page — 39.4 vs 32.1 ms per 512x512x64 image, i.e. slower.)
__device__ __forceinline__ void turn_page(const Ring &rg, int page) {
#if defined(NB_ABL_NOWAIT) || defined(NB_ABL_NODMA)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * DMA_PER_WAVE) : "memory");
#endif
#ifndef NB_ABL_NOBARRIER
    asm volatile("s_barrier" ::: "memory");
#endif
#if !defined(NB_ABL_NODMA) && !defined(NB_DMA_SPREAD)
    issue_page(rg, (page + AHEAD) % N_PAGES);
#endif
}

// ds_read_b128 carries a 16-bit immediate offset: one opaque base register per ring slot reaches every
// fragment with base + immediate, instead of one VALU address computation per read
static_assert(PAGE_BYTES <= 65536, "a page must be addressable with the 16-bit ds_read offset");
__device__ __forceinline__ bf16x8 lds_frag(const Ring &rg, int rec, int lo) {
    const int page = rec / PAGE_RECS, slot = page % N_SLOTS;
    const int off = (rec % PAGE_RECS) * REC_BYTES + lo * 1024;
    return *reinterpret_cast<const bf16x8 *>(rg.lds + rg.base[slot] + off);
}

#ifdef NB_ABL_NOGATHER
template <int L, int B>
__device__ __forceinline__ void fake_level(const SceneDev &sc, const GridCoord &g, const WaveBox &, int hi, int, char *,
                                           float (&out)[lvl_c(L) / 2]) {
#pragma unroll
    for (int i = 0; i < lvl_c(L) / 2; ++i) out[i] = g.gw * (float)(i + 1) + g.gh;
}
#endif
#ifdef NB_ABL_NOMFMA
#undef NB_MFMA16
#define NB_MFMA16(a, b, c) (c)
#endif

// split 8 fp32 values into bf16 hi (round to nearest even) and bf16 lo = rne(v - hi)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 &hi, bf16x8 &lo) {
#ifdef NB_ABL_NOCONV
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    hi = __builtin_bit_cast(bf16x8, a);
    lo = __builtin_bit_cast(bf16x8, b);
    return;
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        hi[i] = h;
        lo[i] = (__bf16)(v[i] - (float)h);
    }
}

__device__ __forceinline__ f32x16 bias_tile(const float *bp, int t, int hi) {
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(bp + (t * 2 + hi) * 16);
    const f32x4 b0 = b4[0], b1 = b4[1], b2 = b4[2], b3 = b4[3];
    return f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
}

// One layer: NT output tiles (processed in interleaved pairs: two independent accumulator chains),
// NC K=16 chunks.  Stream order within a pair of tiles: (c, t0), (c, t1), (c+1, t0), ...
// The LDS fragment reads run one record pair ahead of the MFMAs; sched_barrier pins that order
// (left alone, the scheduler clusters hundreds of ds_reads and spills).
struct Frag4 {
    bf16x8 ah0, al0, ah1, al1;
};

// hipcc waits lgkmcnt(0) before the first use of ANY LDS read once an LDS-DMA has been issued (measured: 348
// full drains per depth step, i.e. the prefetched fragments of the NEXT record pair were waited for together
// with the current ones and the LDS latency was exposed in every other iteration).  The fragment reads are
// therefore issued from inline asm, which the compiler does not count, and waited for by hand with a COUNTED
// lgkmcnt: LDS operations retire in order, so "at most 4 outstanding" means everything older than the four reads
// just issued for the next pair — in particular the current pair's fragments — has landed.
template <int REC0>
__device__ __forceinline__ void load_pair(const Ring &rg, int k, Frag4 &f) {
    const int rec = REC0 + 2 * k;
    if (rec % PAGE_RECS == 0) turn_page(rg, rec / PAGE_RECS);
#if defined(NB_DMA_SPREAD) && !defined(NB_ABL_NODMA)
    issue_piece(rg, (rec / PAGE_RECS + AHEAD) % N_PAGES, (rec % PAGE_RECS) / 2);  // after the page's barrier
#endif
    const int page = rec / PAGE_RECS, slot = page % N_SLOTS;
    const int addr = rg.base[slot] + (rec % PAGE_RECS) * REC_BYTES;  // lane * 16 + slot base + record offset
#ifdef NB_ABL_NOLDS
    asm volatile("" : "=v"(f.ah0), "=v"(f.al0), "=v"(f.ah1), "=v"(f.al1) : "v"(addr));
    return;
#endif
    // one statement: four reads of two consecutive records (A_hi, A_lo of tile t0, then of tile t1)
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:1024\n\t"
        "ds_read_b128 %2, %4 offset:2048\n\t"
        "ds_read_b128 %3, %4 offset:3072"
        : "=&v"(f.ah0), "=&v"(f.al0), "=&v"(f.ah1), "=&v"(f.al1)
        : "v"(addr)
        : "memory");
}

// wait until the fragments in `f` have landed while the NEWER reads (4 of them, or none) may stay in flight
template <int NEWER>
__device__ __forceinline__ void wait_pair(Frag4 &f) {
#ifdef NB_ABL_NOLGKM
    asm volatile("" : "+v"(f.ah0), "+v"(f.al0), "+v"(f.ah1), "+v"(f.al1));
#else
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.ah0), "+v"(f.al0), "+v"(f.ah1), "+v"(f.al1) : "n"(NEWER));
#endif
}

#ifdef NB_LDS_SPREAD
// Spread variant: the four fragment reads of the NEXT record pair are issued one at a time in the shadow of the
// first four MFMAs of the current pair (consumption order A_hi0, A_hi1, A_lo0, A_lo1) instead of as one burst: the
// four waves of a workgroup run in lock step, so bursts of 16 x 1 KiB back up the LDS queue and the in-order wave
// blocks on the read issue with the matrix pipe idle (measured: removing the reads saves 4.2 of 20.7 ms).
template <int REC0>
__device__ __forceinline__ int pair_addr(const Ring &rg, int k) {
    const int rec = REC0 + 2 * k;
    if (rec % PAGE_RECS == 0) turn_page(rg, rec / PAGE_RECS);
    const int page = rec / PAGE_RECS, slot = page % N_SLOTS;
    return rg.base[slot] + (rec % PAGE_RECS) * REC_BYTES;
}
template <int OFF>
__device__ __forceinline__ void read_frag(bf16x8 &dst, int addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int NEWER>
__device__ __forceinline__ void wait_frag(bf16x8 &f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(NEWER));
}
#endif

// INIT: start the NT accumulator tiles from the bias; otherwise continue accumulating into `acc`
// (a layer whose K range is consumed in several phases: fc_0 level by level, view_fc in two parts).
template <int REC0, int NT, int NC, bool INIT = true>
__device__ __forceinline__ void mlp_layer16(const Ring &rg, const float *bp, f32x16 (&acc)[NT],
                                            const bf16x8 (&xh)[NC], const bf16x8 (&xl)[NC]) {
    const int hi = rg.lane >> 5;
    constexpr int NP = NT / 2 * NC;
    Frag4 buf[2];
#ifdef NB_LDS_SPREAD
    {
        const int a0 = pair_addr<REC0>(rg, 0);
        read_frag<0>(buf[0].ah0, a0);
        read_frag<2048>(buf[0].ah1, a0);
        read_frag<1024>(buf[0].al0, a0);
        read_frag<3072>(buf[0].al1, a0);
    }
#else
    load_pair<REC0>(rg, 0, buf[0]);
#endif
    f32x16 c0, c1;
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
        if (INIT) {
            c0 = bias_tile(bp, 2 * tp, hi);
            c1 = bias_tile(bp, 2 * tp + 1, hi);
        } else {
            c0 = acc[2 * tp];
            c1 = acc[2 * tp + 1];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = tp * NC + c;
            Frag4 &cur = buf[k & 1];
#ifdef NB_LDS_SPREAD
            Frag4 &nxt = buf[(k + 1) & 1];
            constexpr int dummy = 0;
            (void)dummy;
            const bool more = k + 1 < NP;
            int an = 0;
            if (more) an = pair_addr<REC0>(rg, k + 1);
            __builtin_amdgcn_sched_barrier(0);
            // outstanding reads at this point: this pair's four (oldest first: hi0, hi1, lo0, lo1)
            wait_frag<3>(cur.ah0);
            c0 = NB_MFMA16(cur.ah0, xh[c], c0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) read_frag<0>(nxt.ah0, an);
            if (more) wait_frag<3>(cur.ah1); else wait_frag<2>(cur.ah1);
            c1 = NB_MFMA16(cur.ah1, xh[c], c1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) read_frag<2048>(nxt.ah1, an);
            c0 = NB_MFMA16(cur.ah0, xl[c], c0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) read_frag<1024>(nxt.al0, an);
            c1 = NB_MFMA16(cur.ah1, xl[c], c1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) read_frag<3072>(nxt.al1, an);
            if (more) wait_frag<5>(cur.al0); else wait_frag<1>(cur.al0);
            c0 = NB_MFMA16(cur.al0, xh[c], c0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) wait_frag<4>(cur.al1); else wait_frag<0>(cur.al1);
            c1 = NB_MFMA16(cur.al1, xh[c], c1);
            __builtin_amdgcn_sched_barrier(0);
            continue;
#endif
            if (k + 1 < NP) {
                load_pair<REC0>(rg, k + 1, buf[(k + 1) & 1]);
                wait_pair<4>(cur);
            } else {
                wait_pair<0>(cur);
            }
            c0 = NB_MFMA16(cur.ah0, xh[c], c0);
            c1 = NB_MFMA16(cur.ah1, xh[c], c1);
            c0 = NB_MFMA16(cur.ah0, xl[c], c0);
            c1 = NB_MFMA16(cur.ah1, xl[c], c1);
            c0 = NB_MFMA16(cur.al0, xh[c], c0);
            c1 = NB_MFMA16(cur.al1, xh[c], c1);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[2 * tp] = c0;
        acc[2 * tp + 1] = c1;
    }
}

// ---- in-flight operand conversion (NB_CV_INTERLEAVE) -------------------------------------------------------------
// The relu + hi/lo split of a finished tile pair is cut into two-value slices (~9 VALU) and placed by hand, fenced with
// sched_barrier, in the MFMA gaps of the NEXT tile pair of the same layer (measured: fillers fenced in place are about
// two-thirds hidden up to 5 per gap, while anything left to the scheduler clusters and costs its full issue time).
// Only the last tile pair's conversion stays exposed.  Operands are kept as packed words: word (2t + (r>>3))*4 +
// ((r&7)>>1) holds the bf16 pair of accumulator registers (r, r+1) of tile t = elements of K chunk 2t + (r>>3).
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
struct PackedOps {
    unsigned h[64], l[64];
};
__device__ __forceinline__ bf16x8 ops_hi(const PackedOps &p, int c) {
    return __builtin_bit_cast(bf16x8, u32x4s{p.h[4 * c], p.h[4 * c + 1], p.h[4 * c + 2], p.h[4 * c + 3]});
}
__device__ __forceinline__ bf16x8 ops_lo(const PackedOps &p, int c) {
    return __builtin_bit_cast(bf16x8, u32x4s{p.l[4 * c], p.l[4 * c + 1], p.l[4 * c + 2], p.l[4 * c + 3]});
}
template <bool RELU>
__device__ __forceinline__ void cv_slice(const f32x16 &src, int t, int j, PackedOps &o) {  // registers (2j, 2j+1) of tile t
    float x0 = src[2 * j], x1 = src[2 * j + 1];
    if (RELU) {
        x0 = relu1(x0);
        x1 = relu1(x1);
    }
#ifdef NB_ABL_NOCONV
    o.h[(2 * t + (j >> 2)) * 4 + (j & 3)] = __float_as_uint(x0);
    o.l[(2 * t + (j >> 2)) * 4 + (j & 3)] = __float_as_uint(x1);
    return;
#endif
    const bf16x2s hp = {(__bf16)x0, (__bf16)x1};
    const unsigned hw = __builtin_bit_cast(unsigned, hp);
    const float f0 = __uint_as_float(hw << 16), f1 = __uint_as_float(hw & 0xffff0000u);
    const bf16x2s lp = {(__bf16)(x0 - f0), (__bf16)(x1 - f1)};
    o.h[(2 * t + (j >> 2)) * 4 + (j & 3)] = hw;
    o.l[(2 * t + (j >> 2)) * 4 + (j & 3)] = __builtin_bit_cast(unsigned, lp);
}
template <bool RELU>
__device__ __forceinline__ void cv_tiles(const f32x16 (&acc)[8], int t0, int t1, PackedOps &o) {
#pragma unroll
    for (int t = t0; t < t1; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) cv_slice<RELU>(acc[t], t, j, o);
}

// 8-tile layer (or last K phase of fc_0) whose output is converted on the fly into `o` (all tiles but the last pair;
// the caller converts tiles 6 and 7).  XIN = PackedOps or plain fragment arrays via the two accessors.
template <int REC0, int NC, bool INIT, bool RELU, typename XH, typename XL>
__device__ __forceinline__ void mlp_layer16_cv(const Ring &rg, const float *bp, f32x16 (&acc)[8], XH xh, XL xl, PackedOps &o) {
    const int hi = rg.lane >> 5;
    constexpr int NP = 4 * NC;
    constexpr int SL = 16 / NC;  // slices per iteration: one tile pair = 16 two-value slices
    static_assert(NC == 8 || NC == 16, "slice schedule written for 8 or 16 chunks");
    Frag4 buf[2];
    load_pair<REC0>(rg, 0, buf[0]);
    f32x16 c0, c1;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        if (INIT) {
            c0 = bias_tile(bp, 2 * tp, hi);
            c1 = bias_tile(bp, 2 * tp + 1, hi);
        } else {
            c0 = acc[2 * tp];
            c1 = acc[2 * tp + 1];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = tp * NC + c;
            Frag4 &cur = buf[k & 1];
            if (k + 1 < NP) {
                load_pair<REC0>(rg, k + 1, buf[(k + 1) & 1]);
                wait_pair<4>(cur);
            } else {
                wait_pair<0>(cur);
            }
            const bf16x8 bh = xh(c), bl = xl(c);
            c0 = NB_MFMA16(cur.ah0, bh, c0);
            c1 = NB_MFMA16(cur.ah1, bh, c1);
            __builtin_amdgcn_sched_barrier(0);
            if (tp > 0) {
                const int s0 = SL * c;  // slice index 0..15 inside the previous pair: tile 2(tp-1) + (s>>3), pair s&7
                cv_slice<RELU>(acc[2 * (tp - 1) + (s0 >> 3)], 2 * (tp - 1) + (s0 >> 3), s0 & 7, o);
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = NB_MFMA16(cur.ah0, bl, c0);
            c1 = NB_MFMA16(cur.ah1, bl, c1);
            __builtin_amdgcn_sched_barrier(0);
            if (tp > 0 && SL == 2) {
                const int s1 = SL * c + 1;
                cv_slice<RELU>(acc[2 * (tp - 1) + (s1 >> 3)], 2 * (tp - 1) + (s1 >> 3), s1 & 7, o);
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = NB_MFMA16(cur.al0, bh, c0);
            c1 = NB_MFMA16(cur.al1, bh, c1);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[2 * tp] = c0;
        acc[2 * tp + 1] = c1;
    }
}

// relu + split of a finished layer into the next layer's B operands: chunk 2t+h <- registers 8h..8h+7 of tile t
template <bool RELU>
__device__ __forceinline__ void tiles_to_operands(const f32x16 (&acc)[8], bf16x8 (&xh)[16], bf16x8 (&xl)[16]) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = RELU ? relu1(acc[t][8 * h + r]) : acc[t][8 * h + r];
            split8(v, xh[2 * t + h], xl[2 * t + h]);
        }
}

template <int NT>
__device__ __forceinline__ void dump_tiles(const f32x16 (&h)[NT], float *dst, int hi, bool relu) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[32 * t + tile_row(r, hi)] = relu ? fmaxf(h[t][r], 0.f) : h[t][r];
}

template <int NCL>
__device__ __forceinline__ void feats_to_operands(const float (&f)[8 * NCL], bf16x8 (&fh)[NCL], bf16x8 (&fl)[NCL],
                                                  float *dbg, int dbg_base, int hi) {
#pragma unroll
    for (int c = 0; c < NCL; ++c) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = f[8 * c + r];
        split8(v, fh[c], fl[c]);
        if (dbg) {
#pragma unroll
            for (int r = 0; r < 8; ++r) dbg[TAP_F + col_feat(dbg_base + 8 * c + r, hi)] = v[r];
        }
    }
}

template <bool DENSITY_ONLY, bool DBG>
__device__ __forceinline__ void decode16(const SceneDev &sc, const Ring &rg, float px, float py, float pz,
                                         float vx, float vy, float vz, float (&pe)[N_PE], float (&out)[4],
                                         float *dbg) {
    const int hi = rg.lane >> 5;
    const float *prm = reinterpret_cast<const float *>(rg.lds + RING_BYTES);
    f32x16 acc[8];
    bf16x8 xh[16], xl[16];
#ifdef NB_CV_INTERLEAVE
    PackedOps xa, xb;
#endif
    {
        // fc_0 level by level: gather one pyramid level (fp32), split it into bf16 B operands and
        // accumulate its K range into all 8 output tiles before touching the next level, so only one
        // level's features are ever live (the weight stream is ordered to match, see nb_pack16_kernel)
        const GridCoord g = grid_coords(sc, px, py, pz);
        const WaveBox wb = wave_box(g);
#ifdef NB_ABL_NOGATHER
#define gather_level_coop fake_level
#endif
        {
            float f[16];
            bf16x8 fh[2], fl[2];
            gather_level_coop<0, TILE_BYTES>(sc, g, wb, hi, rg.lane, rg.tile, f);
            feats_to_operands<2>(f, fh, fl, DBG ? dbg : nullptr, 0, hi);
            mlp_layer16<REC_L0 + 8 * 0, 8, 2, true>(rg, prm + P_B0, acc, fh, fl);
        }
        {
            float f[32];
            bf16x8 fh[4], fl[4];
            gather_level_coop<1, TILE_BYTES>(sc, g, wb, hi, rg.lane, rg.tile, f);
            feats_to_operands<4>(f, fh, fl, DBG ? dbg : nullptr, 16, hi);
            mlp_layer16<REC_L0 + 8 * 2, 8, 4, false>(rg, prm + P_B0, acc, fh, fl);
        }
        {
            float f[64];
            bf16x8 fh[8], fl[8];
            gather_level_coop<2, TILE_BYTES>(sc, g, wb, hi, rg.lane, rg.tile, f);
            feats_to_operands<8>(f, fh, fl, DBG ? dbg : nullptr, 48, hi);
            mlp_layer16<REC_L0 + 8 * 6, 8, 8, false>(rg, prm + P_B0, acc, fh, fl);
        }
        {
            float f[64];
            bf16x8 fh[8], fl[8];
            gather_level_coop<3, TILE_BYTES>(sc, g, wb, hi, rg.lane, rg.tile, f);
            feats_to_operands<8>(f, fh, fl, DBG ? dbg : nullptr, 112, hi);
#ifdef NB_CV_INTERLEAVE
            mlp_layer16_cv<REC_L0 + 8 * 14, 8, false, true>(
                rg, prm + P_B0, acc, [&](int c) { return fh[c]; }, [&](int c) { return fl[c]; }, xa);
#else
            mlp_layer16<REC_L0 + 8 * 14, 8, 8, false>(rg, prm + P_B0, acc, fh, fl);
#endif
        }
#ifdef NB_ABL_NOGATHER
#undef gather_level_coop
#endif
    }
    if (DBG && dbg) dump_tiles(acc, dbg + TAP_H1, hi, true);
#ifdef NB_CV_INTERLEAVE
    cv_tiles<true>(acc, 6, 8, xa);
    mlp_layer16_cv<REC_L1, NCH, true, true>(
        rg, prm + P_B1, acc, [&](int c) { return ops_hi(xa, c); }, [&](int c) { return ops_lo(xa, c); }, xb);
    if (DBG && dbg) dump_tiles(acc, dbg + TAP_H2, hi, true);
    cv_tiles<true>(acc, 6, 8, xb);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        xh[c] = ops_hi(xb, c);
        xl[c] = ops_lo(xb, c);
    }
    mlp_layer16<REC_L2, 8, NCH>(rg, prm + P_B2, acc, xh, xl);
#else
    tiles_to_operands<true>(acc, xh, xl);
    mlp_layer16<REC_L1, 8, NCH>(rg, prm + P_B1, acc, xh, xl);
    if (DBG && dbg) dump_tiles(acc, dbg + TAP_H2, hi, true);
    tiles_to_operands<true>(acc, xh, xl);
    mlp_layer16<REC_L2, 8, NCH>(rg, prm + P_B2, acc, xh, xl);
#endif
    if (DBG && dbg) dump_tiles(acc, dbg + TAP_H3, hi, true);
    // alpha_fc in fp32 on the VALU from the un-split fc_2 output
    {
        const f32x4 *aw = reinterpret_cast<const f32x4 *>(prm + P_AW + hi * 128);
        float s = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 32; ++q4) {
            const f32x4 w = aw[q4];
            s = fmaf(w.x, relu1(acc[q4 >> 2][(q4 & 3) * 4 + 0]), s);
            s = fmaf(w.y, relu1(acc[q4 >> 2][(q4 & 3) * 4 + 1]), s);
            s = fmaf(w.z, relu1(acc[q4 >> 2][(q4 & 3) * 4 + 2]), s);
            s = fmaf(w.w, relu1(acc[q4 >> 2][(q4 & 3) * 4 + 3]), s);
        }
        s = add_halves(s);
        out[3] = s + prm[P_AB];
    }
    // the ring protocol needs every wave to walk the whole stream, so DENSITY_ONLY still runs the
    // colour head (its result is simply not stored)
#ifdef NB_CV_INTERLEAVE
    cv_tiles<true>(acc, 0, 8, xa);  // fc_2's output also feeds alpha_fc in fp32 above: converted after the layer
    mlp_layer16_cv<REC_L4, NCH, true, false>(
        rg, prm + P_LB, acc, [&](int c) { return ops_hi(xa, c); }, [&](int c) { return ops_lo(xa, c); }, xb);
    if (DBG && dbg) dump_tiles(acc, dbg + TAP_G, hi, false);
    cv_tiles<false>(acc, 6, 8, xb);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        xh[c] = ops_hi(xb, c);
        xl[c] = ops_lo(xb, c);
    }
    f32x16 v[4];
#else
    tiles_to_operands<true>(acc, xh, xl);
    mlp_layer16<REC_L4, 8, NCH>(rg, prm + P_LB, acc, xh, xl);
    if (DBG && dbg) dump_tiles(acc, dbg + TAP_G, hi, false);
    // view_fc in two K phases: the 256 outputs of the merged latent layer, then the positional encodings
    f32x16 v[4];
    tiles_to_operands<false>(acc, xh, xl);
#endif
    mlp_layer16<REC_LV, 4, 16, true>(rg, prm + P_BV, v, xh, xl);
    {
        // the 30 sin/cos of the world point are only needed here: computing them late keeps registers
        // free during the gather and the trunk
#ifdef NB_ABL_NOPE
#pragma unroll
        for (int c = 12; c < N_PE; ++c) pe[c] = px * (float)c;
#else
        if (!DENSITY_ONLY) pe_xyz(pe, px, py, pz, vx, vy, vz, hi);
#endif
        bf16x8 ph[6], pl[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {  // 45 positional-encoding slots, zero padded to 48
            float t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = (8 * c + r < N_PE) ? pe[(8 * c + r < N_PE) ? 8 * c + r : 0] : 0.f;
            split8(t, ph[c], pl[c]);
        }
        mlp_layer16<REC_LV + 4 * 16, 4, 6, false>(rg, prm + P_BV, v, ph, pl);
    }
#if defined(NB_DMA_SPREAD) && !defined(NB_ABL_NODMA)
    // the two padding records form the 325th record pair of the step: issue the DMA piece that rides on it
    issue_piece(rg, (N_RECS / PAGE_RECS + AHEAD) % N_PAGES, (N_RECS % PAGE_RECS) / 2);
#endif
    if (DBG && dbg) {
        dump_tiles(v, dbg + TAP_V, hi, true);
        dump_pe(pe, dbg + TAP_PE, hi);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const f32x4 *rw = reinterpret_cast<const f32x4 *>(prm + P_RW + (ch * 2 + hi) * 64);
        float s = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < 16; ++q4) {
            const f32x4 w = rw[q4];
            s = fmaf(w.x, relu1(v[q4 >> 2][(q4 & 3) * 4 + 0]), s);
            s = fmaf(w.y, relu1(v[q4 >> 2][(q4 & 3) * 4 + 1]), s);
            s = fmaf(w.z, relu1(v[q4 >> 2][(q4 & 3) * 4 + 2]), s);
            s = fmaf(w.w, relu1(v[q4 >> 2][(q4 & 3) * 4 + 3]), s);
        }
        s = add_halves(s);
        out[ch] = s + prm[P_RB + ch];
    }
}

__device__ __forceinline__ Ring ring_begin(const float *pk, const float *lb, char *lds) {
    {
        float *prm = reinterpret_cast<float *>(lds + RING_BYTES);
        for (int i = threadIdx.x; i < P_SIZE; i += 256) {
            float v;
            if (i < P_B1) v = pk[F_OFF_B0 + i - P_B0];
            else if (i < P_B2) v = pk[F_OFF_B1 + i - P_B1];
            else if (i < P_LB) v = pk[F_OFF_B2 + i - P_B2];
            else if (i < P_BV) v = lb[i - P_LB];
            else if (i < P_AW) v = pk[F_OFF_BV + i - P_BV];
            else if (i < P_RW) v = pk[F_OFF_AW + i - P_AW];
            else if (i < P_AB) v = pk[F_OFF_RW + i - P_RW];
            else if (i < P_RB) v = pk[F_OFF_AB + i - P_AB];
            else v = pk[F_OFF_RB + i - P_RB];
            prm[i] = v;
        }
        __syncthreads();
    }
    Ring rg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    rg.wave_off = wave * (DMA_PER_WAVE * 1024);
    rg.stream = reinterpret_cast<const char *>(pk + F_PACK_SIZE) + rg.wave_off;
    rg.lds = lds;
    rg.lane = threadIdx.x & 63;
    rg.tile = lds + RING_BYTES + PARAM_BYTES + wave * TILE_BYTES;
#pragma unroll
    for (int sl = 0; sl < N_SLOTS; ++sl) {
        rg.base[sl] = rg.lane * 16 + sl * PAGE_BYTES;
        asm volatile("" : "+v"(rg.base[sl]));
    }
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) issue_page(rg, p);
    return rg;
}

__device__ __forceinline__ void ring_end() {
    // pages prefetched past the end of the work are still in flight: let them land before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
}

// ---------------------------------------------------------------- point mode
template <bool DENSITY_ONLY, bool DBG>
__global__ __launch_bounds__(256) void nb_points16_kernel(MarchArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const Ring rg = ring_begin(a.pk, a.lb, lds);
    const int lane = rg.lane, j = lane & 31, hi = lane >> 5;
    const long long wave = (long long)blockIdx.x * 4 + rg.wave_off / (DMA_PER_WAVE * 1024);
    long long idx = wave * 32 + j;
    const bool valid = idx < a.n_pts;  // whole waves past the end keep walking the ring (barriers!)
    if (!valid) idx = a.n_pts - 1;
    const float px = a.wpts[idx * 3 + 0], py = a.wpts[idx * 3 + 1], pz = a.wpts[idx * 3 + 2];
    float pe[N_PE];
#pragma unroll
    for (int c = 0; c < N_PE; ++c) pe[c] = 0.f;
    float v3[3] = {0.f, 0.f, 0.f};
    if (!DENSITY_ONLY) {
        const float vx = a.viewdir[idx * 3 + 0], vy = a.viewdir[idx * 3 + 1], vz = a.viewdir[idx * 3 + 2];
        pe_view(pe, vx, vy, vz, hi);
        v3[0] = vx;
        v3[1] = vy;
        v3[2] = vz;
    }
    float out[4];
    float *dbg = (DBG && a.dbg && valid) ? a.dbg + idx * TAP_WIDTH : nullptr;
    decode16<DENSITY_ONLY, DBG>(a.sc, rg, px, py, pz, v3[0], v3[1], v3[2], pe, out, dbg);
    if (valid && hi == 0) {
        if (DENSITY_ONLY) {
            a.raw_out[idx] = out[3];
        } else {
            *reinterpret_cast<f32x4 *>(a.raw_out + idx * 4) = f32x4{out[0], out[1], out[2], out[3]};
        }
    }
    ring_end();
}

// ---------------------------------------------------------------- ray mode
__global__ __launch_bounds__(256) void nb_march16_kernel(MarchArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const Ring rg = ring_begin(a.pk, a.lb, lds);
    const int lane = rg.lane, j = lane & 31, hi = lane >> 5;
    const int grp = xcd_remap(blockIdx.x, a.n_wave_groups);
    const long long wave = (long long)grp * 4 + rg.wave_off / (DMA_PER_WAVE * 1024);
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
        // The pages prefetched for this step's first layers were requested during the previous step and have
        // landed by now; a compiler-visible vmcnt(0) here costs nothing and clears hipcc's "LDS-DMA pending"
        // state, so the gather's ordinary loads below get counted waits instead of a drain per load.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        const float z_next = (s + 1 < S) ? z_at(s + 1) : 0.f;
        const float px = __fadd_rn(ox, __fmul_rn(dx, z_cur));
        const float py = __fadd_rn(oy, __fmul_rn(dy, z_cur));
        const float pz = __fadd_rn(oz, __fmul_rn(dz, z_cur));
        float out[4];
        // Everything derived from the lane id, the ray id and the stream base is loop-invariant; left alone,
        // LICM hoists hundreds of such values (DMA addresses, LDS bases, output pointers) out of the depth
        // loop and the allocator spills them, reloading each behind a vmcnt(0).  Laundering the roots through
        // an empty asm makes the compiler recompute them (a few ALU ops) inside the body instead.
        int zero = 0, lane_i = rg.lane;
        asm volatile("" : "+s"(zero), "+v"(lane_i));
        Ring r2;
        r2.lds = rg.lds;
        r2.lane = lane_i;
        r2.stream = rg.stream + zero;
        r2.wave_off = rg.wave_off + zero;
        r2.tile = rg.tile + zero;
#pragma unroll
        for (int sl = 0; sl < N_SLOTS; ++sl) {
            r2.base[sl] = lane_i * 16 + sl * PAGE_BYTES;
            asm volatile("" : "+v"(r2.base[sl]));  // keep base + immediate addressing (do not fold into per-read adds)
        }
        // Sample culling (nb_cull).  The four waves walk the weight ring in lock step, so skipping a depth step is a
        // WORKGROUP decision: decode only if any sample of the 128 rays survives (two extra barriers per step, paid
        // only when culling is on); culled samples of a decoded step get raw = 0.
        bool ins = true, run = true;
        if (a.cull.n_views) {
            ins = cull_inside(a.cull, a.sc, px, py, pz);
            int *flags = reinterpret_cast<int *>(rg.lds + RING_BYTES) + P_SIZE;
            const int any_wave = __any(ins) ? 1 : 0;  // evaluated by the whole wave, stored by one lane
            if (lane_i == 0) flags[rg.wave_off / (DMA_PER_WAVE * 1024)] = any_wave;
            __syncthreads();
            // readfirstlane: the decision is uniform by construction; say so, so that the ring's barriers and DMA sit
            // under a scalar branch instead of an exec-masked region
            run = __builtin_amdgcn_readfirstlane(flags[0] | flags[1] | flags[2] | flags[3]) != 0;
            __syncthreads();
#ifdef NB_CULL_NOSKIP
            run = true;
#endif
        }
        if (run) decode16<false, false>(a.sc, r2, px, py, pz, vx, vy, vz, pe, out, nullptr);
        if (!ins || !run) out[0] = out[1] = out[2] = out[3] = 0.f;
        float dist = (s + 1 < S) ? __fsub_rn(z_next, z_cur) : 1e10f;
        dist = __fmul_rn(dist, dn);
        const float w = ra.add(out, z_cur, dist);
        wstore.push(a, ray, s, S, hi, valid, w);
        if (valid && hi == 0 && a.raw)
            *reinterpret_cast<f32x4 *>(a.raw + (ray * S + s) * 4) = f32x4{out[0], out[1], out[2], out[3]};
        z_cur = z_next;
    }
    if (valid && hi == 0) ra.store(a, ray);
    ring_end();
}

// ---------------------------------------------------------------- weight stream packing
__device__ __forceinline__ unsigned short bf16_rne_bits(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ void nb_pack16_kernel(nb_mlp_params p, const float *__restrict__ f32_blob, unsigned short *__restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one (rec, lane, r) weight, writes hi and lo
    if (e >= (long long)N_RECS_PAD * 64 * 8) return;
    const int r = (int)(e & 7), lane = (int)((e >> 3) & 63), rec = (int)(e >> 9);
    if (rec >= N_RECS) {  // padding records: never multiplied, zero for determinism
        unsigned short *padp = out + (size_t)rec * (REC_BYTES / 2);
        padp[lane * 8 + r] = 0;
        padp[512 + lane * 8 + r] = 0;
        return;
    }
    const int i = lane & 31, kg = lane >> 5;
    // locate (layer, tile, chunk): within a layer records go pair by pair, (c, t0), (c, t1), ...
    int base, nc, layer;
    if (rec < REC_L1) { layer = 0; base = REC_L0; nc = NC0; }
    else if (rec < REC_L2) { layer = 1; base = REC_L1; nc = NCH; }
    else if (rec < REC_L4) { layer = 2; base = REC_L2; nc = NCH; }
    else if (rec < REC_LV) { layer = 4; base = REC_L4; nc = NCH; }
    else { layer = 5; base = REC_LV; nc = NCV; }
    // records of a layer phase go tile pair by tile pair: (c, t0), (c, t1), (c+1, t0), ...  fc_0 is split
    // into four K phases (one per pyramid level), view_fc into two (latent-layer outputs, encodings)
    int rel = rec - base, c0 = 0, n = nc, nt = (layer == 5) ? 4 : 8;
    if (layer == 0) {
        if (rel < 8 * 2) { c0 = 0; n = 2; }
        else if (rel < 8 * 6) { c0 = 2; n = 4; }
        else if (rel < 8 * 14) { c0 = 6; n = 8; }
        else { c0 = 14; n = 8; }
    } else if (layer == 5) {
        if (rel < 4 * 16) { c0 = 0; n = 16; }
        else { c0 = 16; n = 6; }
    }
    rel -= nt * c0;
    const int tp = rel / (2 * n), c = c0 + ((rel % (2 * n)) >> 1), t = 2 * tp + (rel & 1);
    const int row = 32 * t + i, q = 8 * c + r;
    float w = 0.f;
    if (layer == 0) w = p.fc0_w[row * 352 + col_feat(q, kg)];
    else if (layer == 1) w = p.fc1_w[row * 256 + col_hidden(q, kg)];
    else if (layer == 2) w = p.fc2_w[row * 256 + col_hidden(q, kg)];
    else if (layer == 4) {
        // merged latent_fc[:, :256] @ feature_fc, taken from the f32 blob (already computed in fp64 there):
        // f32 layout ((t*32 + g)*64 + lane')*4 + e with chunk2 = 4g+e <-> column col_hidden(chunk2, hi')
        const int col = col_hidden(q, kg);
        // invert col_hidden: find (q2, hi2) with col_hidden(q2, hi2) == col
        const int tt = col >> 5, rr = col & 31, hi2 = (rr >> 2) & 1, r2 = (rr & 3) + 4 * (rr >> 3);
        const int q2 = 16 * tt + r2;
        w = f32_blob[F_OFF_L4 + ((t * 32 + (q2 >> 2)) * 64 + (hi2 * 32 + i)) * 4 + (q2 & 3)];
    } else {
        const int col = q < 128 ? col_hidden(q, kg) : col_pe(q - 128, kg);
        w = col < 0 ? 0.f : p.view_w[row * 346 + col];
    }
    const unsigned short h = bf16_rne_bits(w);
    const float hf = __uint_as_float((unsigned)h << 16);
    const unsigned short l = bf16_rne_bits(w - hf);
    unsigned short *recp = out + (size_t)rec * (REC_BYTES / 2);
    recp[lane * 8 + r] = h;
    recp[512 + lane * 8 + r] = l;
}

}  // namespace

namespace nbm {

long long bf16_stream_floats() { return (long long)N_RECS_PAD * REC_BYTES / 4; }

int pack_bf16_stream(const nb_mlp_params *p, float *packed, hipStream_t st) {
    const long long n = (long long)N_RECS_PAD * 64 * 8;
    hipLaunchKernelGGL(nb_pack16_kernel, dim3(nb_ceil_div(n, 256)), dim3(256), 0, st, *p, packed,
                       reinterpret_cast<unsigned short *>(packed + F_PACK_SIZE));
    NB_CHECK_LAUNCH("nb_pack16_kernel");
    return NB_OK;
}

int launch_points_bf16(const MarchArgs &a, int density_only, hipStream_t st) {
    const dim3 grid(nb_ceil_div(a.n_pts, 128)), block(256);
    if (density_only) hipLaunchKernelGGL((nb_points16_kernel<true, false>), grid, block, 0, st, a);
    else if (a.dbg) hipLaunchKernelGGL((nb_points16_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((nb_points16_kernel<false, false>), grid, block, 0, st, a);
    NB_CHECK_LAUNCH("nb_points16_kernel");
    return NB_OK;
}

int launch_march_bf16(const MarchArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(nb_march16_kernel, dim3(a.n_wave_groups), dim3(256), 0, st, a);
    NB_CHECK_LAUNCH("nb_march16_kernel");
    return NB_OK;
}

}  // namespace nbm
