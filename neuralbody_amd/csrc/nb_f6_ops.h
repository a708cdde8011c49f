// nb_f6_ops.h — operand conversions of the "f16f6" arithmetic (nb_march_fold.hip):
//     W.X ~= W_h.X_h (fp16 MFMA) + fp6(W_h).bf6(X_l) + fp6(W_l).bf6(X_h) (scaled K=64 MFMA).
#pragma once
#include "nb_march_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));

namespace nbm {

// MODE.FP16_OVFL = 1 for the rest of the wave's life: a conversion to fp16 whose input is finite and beyond +-65504
// SATURATES instead of returning inf (tools/experiments/probe_f16ovfl.hip; v_cvt_pk_f16_f32 has no clamp of its own).  The
// fp16 head of an activation is the one place where these arithmetics have less range than fp32: without this, one
// feature or layer output above 65504 turns into inf, inf - inf in the remainder, and a NaN pixel; with it the product is
// merely inaccurate there.  Costs one scalar instruction per wave.
__device__ __forceinline__ void saturate_fp16_conversions() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

// fp16 head of two values (round to nearest even) and the exact fp32 remainder x - fp16(x) as ONE v_fma_mix_f32 each
// (hipcc's own lowering of `x - (float)(_Float16)x` converts every value twice: 9 instead of 5 instructions per pair)
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int SEL>
__device__ __forceinline__ float rem16(float x, unsigned h) {
    float r;
    if (SEL == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}

// Six-bit variant.  One K block = 32 consecutive values of a lane: fp16 heads (4 chunks), bf6 of the heads, bf6 of the
// remainders, and the block's E8M0 exponent t: the head block is stored / 2^(t - 127) with t = exponent(max |v|) - 3 (the
// largest value lands in [8, 16) of bf6's +-28), the remainder block / 2^(t - 11 - 127) (|remainder| <= 2^-11 |value|).
// Element order inside the bf6 registers: heads natural (v_cvt_scalef32_pk32_bf6_f16), remainders interleaved
// [v0, v16, v1, v17, ...] (v_cvt_scalef32_2xpk16_bf6_f32) — the weight records of the two cross terms are packed to match.
__device__ __forceinline__ int block_exponent(float m) {
    return max(__float_as_int(m) >> 23, 15) - 3;  // m >= 0; t - 11 >= 1 stays a valid E8M0 / float exponent
}
__device__ __forceinline__ void cvt_block6(const u32x16 hv, const f32x16 ra, const f32x16 rb, int t, i32x6 &x, i32x6 &l) {
    const float sf = __int_as_float(t << 23), sfl = __int_as_float((t - 11) << 23);  // the conversions DIVIDE by the scale
    // Inline asm with EARLY-CLOBBER results, not the builtins: these are multi-pass instructions that write their six result
    // registers while still reading the scale operand, and hipcc (ROCm 7.2) is free to allocate the result over the scale
    // register — every element converted after the first pass then sees a clobbered scale and saturates
    // (tools/experiments/probe_cross6.hip: `v_cvt_scalef32_2xpk16_bf6_f32 v[6:11], v[32:47], v[48:63], v7`).
    asm volatile("v_cvt_scalef32_pk32_bf6_f16 %0, %1, %2" : "=&v"(x) : "v"(hv), "v"(sf));
    asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, %3" : "=&v"(l) : "v"(ra), "v"(rb), "v"(sfl));
}
// NG groups of 16 values -> 2 NG chunks, ceil(NG / 2) blocks and their exponents packed four to a register (byte b % 4 of
// sh[b / 4]: head block, of sl[b / 4]: remainder block)
template <int NG, bool RELU, class Get>
__device__ __forceinline__ void make_operands6(Get get, f16x8 (&xh)[2 * NG], i32x6 (&xl)[(NG + 1) / 2], i32x6 (&xx)[(NG + 1) / 2],
                                               int (&eb)[(NG + 1) / 2]) {
    constexpr int NB = (NG + 1) / 2;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        u32x16 hv;
        f32x16 ra, rb;
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // value pair i of the block: group 2 b (i < 8) or 2 b + 1
            const int g = 2 * b + (i >> 3);
            float v0 = 0.f, v1 = 0.f;
            if (g < NG) {
                v0 = get(16 * g + 2 * (i & 7));
                v1 = get(16 * g + 2 * (i & 7) + 1);
                if (RELU) {
                    v0 = relu1(v0);
                    v1 = relu1(v1);
                }
            }
            const unsigned h = cvt_pk_f16(v0, v1);
            hv[i] = h;
            // ONE v_max3_f32 with |.| source modifiers: written as fmaxf(m, fmaxf(fabsf(v0), fabsf(v1))) hipcc canonicalises
            // both inputs first (a v_max_f32 x, |x|, |x| each): three instructions per value pair instead of one
            asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(v0), "v"(v1));
            const float r0 = rem16<0>(v0, h), r1 = rem16<1>(v1, h);
            if (i < 8) {
                ra[2 * i] = r0;
                ra[2 * i + 1] = r1;
            } else {
                rb[2 * (i - 8)] = r0;
                rb[2 * (i - 8) + 1] = r1;
            }
        }
        eb[b] = block_exponent(m);
        cvt_block6(hv, ra, rb, eb[b], xx[b], xl[b]);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * b + c < 2 * NG) xh[4 * b + c] = __builtin_bit_cast(f16x8, u32x4{hv[4 * c], hv[4 * c + 1], hv[4 * c + 2], hv[4 * c + 3]});
    }
#pragma unroll
    for (int c = 0; c < 2 * NG; ++c) asm volatile("" : "+v"(xh[c]));
#pragma unroll
    for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(xl[b]), "+v"(xx[b]));
}
// fp6 e2m3 (sign, 2 exponent bits of bias 1, 3 mantissa bits; subnormal step 1/8, largest 7.5), round to nearest even
__device__ __forceinline__ unsigned fp6_e2m3_bits(float v) {
    const unsigned sgn = v < 0.f ? 32u : 0u;
    const float a = fminf(fabsf(v), 7.5f);
    unsigned code;
    if (a < 1.f) {
        code = (unsigned)rintf(a * 8.f);  // 8 = 1.0, the first normal
    } else {
        int e = a >= 4.f ? 2 : (a >= 2.f ? 1 : 0);
        int m = (int)rintf((ldexpf(a, -e) - 1.f) * 8.f);
        if (m == 8) {
            m = 0;
            ++e;
        }
        code = (unsigned)(((e + 1) << 3) | m);
        if (code > 31u) code = 31u;
    }
    return sgn | code;
}

}  // namespace nbm
