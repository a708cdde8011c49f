// probe_cross6.hip — one K=64 block of the six-bit cross terms end to end, as nb_march_f16.hip (-DF_SIX) computes them:
// weights fp6 e2m3 + per-lane E8M0, activations bf6 e3m2 from v_cvt_scalef32_pk32_bf6_f16 / v_cvt_scalef32_2xpk16_bf6_f32
// with the block exponent, against the exact products W_h.X_l and W_l.X_h.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ unsigned fp6_e2m3_bits(float v) {
    const unsigned sgn = v < 0.f ? 32u : 0u;
    const float a = fminf(fabsf(v), 7.5f);
    unsigned code;
    if (a < 1.f) code = (unsigned)rintf(a * 8.f);
    else {
        int e = a >= 4.f ? 2 : (a >= 2.f ? 1 : 0);
        int m = (int)rintf((ldexpf(a, -e) - 1.f) * 8.f);
        if (m == 8) { m = 0; ++e; }
        code = (unsigned)(((e + 1) << 3) | m);
        if (code > 31u) code = 31u;
    }
    return sgn | code;
}
__device__ unsigned cvt_pk_f16(float a, float b) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// W [32 rows][64 k], X [32 cols][64 k] (k = 32 * half + natural index n); out[0]: D of W_h.X_l, out[1]: D of W_l.X_h  (each [32][32])
__global__ void k(const float *W, const float *X, float *out, int order, int dsb) {
    const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
    for (int lo = 0; lo < 2; ++lo) {
        // ---- A: as nb_pack_f16_kernel
        float wv[32], amax = 0.f;
        for (int e = 0; e < 32; ++e) {
            const int n = lo ? e : 16 * (e & 1) + (e >> 1);
            const float w = W[i * 64 + 32 * kg + n];
            const float h = (float)(_Float16)w;
            wv[e] = lo ? w - h : h;
            amax = fmaxf(amax, fabsf(wv[e]));
        }
        int ex = 0;
        if (amax > 0.f) { ex = ilogbf(amax / 7.5f); if (ldexpf(7.5f, ex) < amax) ++ex; }
        unsigned w32[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int e = 0; e < 32; ++e) {
            const unsigned code = fp6_e2m3_bits(ldexpf(wv[e], -ex));
            const int bit = 6 * e;
            w32[bit >> 5] |= code << (bit & 31);
            if ((bit & 31) > 26) w32[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
        }
        w32[6] = (unsigned)(127 + ex);
        // ---- B: as make_operands6
        u32x16 hv; f32x16 ra, rb; float m = 0.f;
        for (int p = 0; p < 16; ++p) {
            const float v0 = X[i * 64 + 32 * kg + 2 * p], v1 = X[i * 64 + 32 * kg + 2 * p + 1];
            const unsigned h = cvt_pk_f16(v0, v1);
            hv[p] = h;
            m = fmaxf(m, fmaxf(fabsf(v0), fabsf(v1)));
            const float h0 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffff)), h1 = (float)__builtin_bit_cast(_Float16, (unsigned short)(h >> 16));
            if (p < 8) { ra[2 * p] = v0 - h0; ra[2 * p + 1] = v1 - h1; } else { rb[2 * (p - 8)] = v0 - h0; rb[2 * (p - 8) + 1] = v1 - h1; }
        }
        const int t = max(__float_as_int(m) >> 23, 15) - 3;
        const float sf = __int_as_float(t << 23), sfl = __int_as_float((t - 11) << 23);
        const i32x6 x6 = __builtin_bit_cast(i32x6, __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(__builtin_bit_cast(f16x32, hv), sf));
        const i32x6 l6 = __builtin_bit_cast(i32x6, __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(ra, rb, sfl));
        if (order == 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(ra), "v"(rb), "v"(sfl), "v"(l6));  // sources and scale pinned
        if (order == 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(sfl), "v"(l6));                    // only the scale
        if (order == 2) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::"v"(ra), "v"(rb), "v"(l6));            // only the sources
        const i32x6 b6 = lo ? x6 : l6;
        const int sb = lo ? t : t - 11 + dsb;
        i32x8 av = {(int)w32[0], (int)w32[1], (int)w32[2], (int)w32[3], (int)w32[4], (int)w32[5], (int)w32[6], 0};
        asm volatile("" : "+v"(av));
        i32x8 bv = {b6[0], b6[1], b6[2], b6[3], b6[4], b6[5], 0, 0};
        asm volatile("" : "+v"(bv));
        f32x16 c = {};
        asm volatile("" : "+v"(c));
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 2, 3, 0, av[6], 0, sb);
        // D layout of the 32x32 MFMA: lane (col = lane % 32, hi = lane / 32) holds rows 8 * (r / 4) + 4 * hi + r % 4
        if (i == 0) { for (int q = 0; q < 8; ++q) { out[2048 + lo * 64 + kg * 16 + q] = __int_as_float(av[q]); out[2048 + lo * 64 + kg * 16 + 8 + q] = __int_as_float(q < 6 ? bv[q] : (q == 6 ? sb : 0)); } }
        for (int r = 0; r < 16; ++r) out[lo * 1024 + (8 * (r / 4) + 4 * kg + r % 4) * 32 + i] = c[r];
    }
}
int main() {
    static float W[32 * 64], X[32 * 64], D[2048 + 128];
    srand(3);
    for (int i = 0; i < 2048; ++i) { W[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.3f; X[i] = fmaxf(0.f, (rand() / (float)RAND_MAX - 0.3f) * 2.f); }
    float *dW, *dX, *dD;
    CK(hipMalloc(&dW, sizeof W)); CK(hipMalloc(&dX, sizeof X)); CK(hipMalloc(&dD, sizeof D));
    CK(hipMemcpy(dW, W, sizeof W, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X, sizeof X, hipMemcpyHostToDevice));
    for (int order = 0; order < 4; ++order)
    for (int dsb = 0; dsb <= 0; ++dsb) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dW, dX, dD, order, dsb);
    CK(hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost));
    printf("variant %d (0: sources + scale pinned behind the conversion, 1: scale only, 2: sources only, 3: nothing)\n", order);
    for (int lo = 0; lo < 2; ++lo) {
        double num = 0, den = 0, worst = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) {
                    const float w = W[r * 64 + kk], x = X[c * 64 + kk];
                    const float wh = (float)(_Float16)w, xh = (float)(_Float16)x;
                    ref += lo ? (double)(w - wh) * xh : (double)wh * (x - xh);
                }
                const double d = D[lo * 1024 + r * 32 + c] - ref;
                num += d * d; den += ref * ref; worst = fmax(worst, fabs(d));
            }
        printf("%s: rms error / rms value = %.4f (worst |error| %.3e, rms value %.3e); D[0][0] = %.6e\n", lo ? "W_l.X_h" : "W_h.X_l", sqrt(num / den), worst, sqrt(den / 1024), D[lo * 1024]);
    }
    // host decode of what lanes 0 and 32 (row 0 / column 0, both K halves) fed the MFMA
    for (int lo = 0; lo < 2; ++lo) {
        double dot = 0;
        for (int kg = 0; kg < 2; ++kg) {
            unsigned a[8], b[8];
            memcpy(a, &D[2048 + lo * 64 + kg * 16], 32); memcpy(b, &D[2048 + lo * 64 + kg * 16 + 8], 32);
            const double sa = ldexp(1.0, (int)(a[6] & 255) - 127), sb = ldexp(1.0, (int)(b[6] & 255) - 127);
            printf("lo %d half %d: A scale byte %u, B scale byte %u\n", lo, kg, a[6] & 255, b[6] & 255);
            for (int e = 0; e < 32; ++e) {
                const int bit = 6 * e;
                unsigned long long ta = a[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? a[bit / 32 + 1] : 0) << 32);
                unsigned long long tb = b[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? b[bit / 32 + 1] : 0) << 32);
                const unsigned ca = (ta >> (bit % 32)) & 63, cb = (tb >> (bit % 32)) & 63;
                const int ea = (ca >> 3) & 3, ma = ca & 7;
                double va = ea == 0 ? ma / 8.0 : ldexp(1.0 + ma / 8.0, ea - 1);
                if (ca & 32) va = -va;
                const int eb = (cb >> 2) & 7, mb = cb & 3;
                double vb = eb == 0 ? ldexp(mb / 4.0, -2) : ldexp(1.0 + mb / 4.0, eb - 3);
                if (cb & 32) vb = -vb;
                dot += va * sa * vb * sb;
                if (false) {
                    const int n = lo ? e : 16 * (e & 1) + (e >> 1);
                    const float w = W[n], x = X[n];
                    const float wh = (float)(_Float16)w, xh = (float)(_Float16)x;
                    printf("   e %d: A %.6e (want %.6e)  B %.6e (want %.6e)\n", e, va * sa, lo ? w - wh : wh, vb * sb, lo ? xh : x - xh);
                }
            }
        }
        printf("lo %d: host dot of the decoded operands %.6e, MFMA D[0][0] %.6e\n", lo, dot, D[lo * 1024]);
    }
    }
    return 0;
}
