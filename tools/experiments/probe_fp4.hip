// probe_fp4.hip — round 5: can the cross terms' WEIGHT operand be fp4 e2m1 (16 bytes per lane and K=64 fragment instead of the
// 24 + 8 of fp6)?  hipcc -O3 --offload-arch=gfx950 tools/experiments/probe_fp4.hip -o exp_bin/probe_fp4, run on the MI355X.
//   A. layout of an fp4 A operand beside a bf6 B operand in v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz 4, blgp 3): hypothesis
//      row = lane % 32, k = 32 (lane / 32) + e, element e in bits [4 e, 4 e + 4) of the lane's first four registers;
//   B. which byte of the scale register op_sel picks (four fragments' E8M0 scales in one register);
//   C. issue rate of fp4 x bf6 against fp6 x bf6 and the fp16 K=16 MFMA.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

template <int OPSEL>
__global__ void mx46_kernel(const v8i *a, const v8i *b, const int *sa, const int *sb, v16f *d) {
    const int l = threadIdx.x, t = blockIdx.x;
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t * 64 + l], b[t * 64 + l], acc, 4, 3, OPSEL, sa[t * 64 + l], 0, sb[t * 64 + l]);
    d[t * 64 + l] = acc;
}

static int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
static void put4(unsigned *regs, int e, unsigned code) { regs[e >> 3] |= (code & 0xfu) << (4 * (e & 7)); }
static void put6(unsigned *regs, int e, unsigned code) {
    const int bit = 6 * e;
    unsigned long long v = (unsigned long long)(code & 0x3fu) << (bit & 31);
    regs[bit >> 5] |= (unsigned)v;
    if ((bit & 31) > 26) regs[(bit >> 5) + 1] |= (unsigned)(v >> 32);
}
// fp4 e2m1: 0 .5 1 1.5 2 3 4 6 (codes 0..7), sign = bit 3; here: integers in {-4..4} \ {no 5}
static unsigned enc4(int v) {
    static const int code_of[5] = {0, 2, 4, 5, 6};
    return (v < 0 ? 8u : 0u) | (unsigned)code_of[std::abs(v)];
}
// bf6 e3m2 (bias 3): integers -3..3: 1 -> e3 m0, 2 -> e4 m0, 3 -> e4 m2
static unsigned enc6b(int v) {
    static const int code_of[4] = {0, 12, 16, 18};
    return (v < 0 ? 32u : 0u) | (unsigned)code_of[std::abs(v)];
}

template <int KIND>  // 0: fp16 K=16, 1: fp6 x bf6, 2: fp4 x bf6, 3: fp4 x fp4
__global__ void rate_kernel(float *out, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 3 + i;
        b[i] = threadIdx.x * 5 + i;
    }
    v16f acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (KIND == 0) {
                h8 ha = __builtin_bit_cast(h8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), hb = __builtin_bit_cast(h8, __builtin_shufflevector(b, b, 0, 1, 2, 3));
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[q], 0, 0, 0);
            } else if (KIND == 1) {
                acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], 2, 3, 0, 127, 0, 127);
            } else if (KIND == 2) {
                acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], 4, 3, 0, 127, 0, 127);
            } else {
                acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[q], 4, 4, 0, 127, 0, 127);
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void rate(const char *name) {
    float *d;
    CK(hipMalloc(&d, 256 * 4 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 16384;
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1024), dim3(256), 0, 0, d, 64);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1024), dim3(256), 0, 0, d, iters);  // 4 waves per CU: one per SIMD
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-22s %.3f ms = %.1f ns per MFMA (1024 workgroups of 4 waves on 256 CUs: 4 generations)\n", name, ms, ms * 1e6 / (4.0 * iters * 4));
    CK(hipFree(d));
}

int main() {
    const int NT = 6;
    std::vector<unsigned> A(NT * 64 * 8, 0), B(NT * 64 * 8, 0);
    std::vector<int> SA(NT * 64, 127), SB(NT * 64, 127), Am(32 * 64), Bm(64 * 32);
    srand(2);
    for (auto &x : Am) x = rand() % 9 - 4;
    for (auto &x : Bm) x = rand() % 7 - 3;
    for (int t = 0; t < NT; ++t)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 32; ++e) {
                const int av = t == 0 ? Am[(l & 31) * 64 + 32 * (l >> 5) + e] : 1, bv = t == 0 ? Bm[(32 * (l >> 5) + e) * 32 + (l & 31)] : 1;
                put4(&A[(t * 64 + l) * 8], e, enc4(av));
                put6(&B[(t * 64 + l) * 8], e, enc6b(bv));
            }
    // garbage in registers 4..7 of the fp4 operand must not matter
    for (int l = 0; l < 64; ++l)
        for (int r = 4; r < 8; ++r) A[(0 * 64 + l) * 8 + r] = 0xdeadbeefu;
    for (int l = 0; l < 64; ++l) {
        SA[1 * 64 + l] = 127 + (l % 4);                               // per-lane scale (row and K half)
        for (int t = 2; t < 6; ++t) SA[t * 64 + l] = 127 | (128 << 8) | (129 << 16) | (130 << 24);  // op_sel t - 2
    }
    unsigned *dA, *dB;
    int *dSA, *dSB;
    float *dD;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dSA, SA.size() * 4));
    CK(hipMalloc(&dSB, SB.size() * 4));
    CK(hipMalloc(&dD, NT * 64 * 16 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dSA, SA.data(), SA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dSB, SB.data(), SB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mx46_kernel<0>, dim3(3), dim3(64), 0, 0, (const v8i *)dA, (const v8i *)dB, dSA, dSB, (v16f *)dD);
    hipLaunchKernelGGL(mx46_kernel<1>, dim3(1), dim3(64), 0, 0, (const v8i *)dA + 3 * 64, (const v8i *)dB + 3 * 64, dSA + 3 * 64, dSB + 3 * 64, (v16f *)dD + 3 * 64);
    hipLaunchKernelGGL(mx46_kernel<2>, dim3(1), dim3(64), 0, 0, (const v8i *)dA + 4 * 64, (const v8i *)dB + 4 * 64, dSA + 4 * 64, dSB + 4 * 64, (v16f *)dD + 4 * 64);
    hipLaunchKernelGGL(mx46_kernel<3>, dim3(1), dim3(64), 0, 0, (const v8i *)dA + 5 * 64, (const v8i *)dB + 5 * 64, dSA + 5 * 64, dSB + 5 * 64, (v16f *)dD + 5 * 64);
    CK(hipDeviceSynchronize());
    std::vector<float> D(NT * 64 * 16);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    auto at = [&](int t, int i, int j) {
        for (int r = 0; r < 16; ++r)
            for (int h = 0; h < 2; ++h)
                if (crow(r, 32 * h) == i) return D[(t * 64 + 32 * h + j) * 16 + r];
        return NAN;
    };
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int ref = 0;
            for (int k = 0; k < 64; ++k) ref += Am[i * 64 + k] * Bm[k * 32 + j];
            if (at(0, i, j) != (float)ref) ++bad;
        }
    printf("A. fp4 (cbsz 4) x bf6 (blgp 3), element e in bits [4e, 4e+4), row = lane %% 32, k = 32 (lane / 32) + e: %s (%d mismatches of 1024; D[0][0..3] = %g %g %g %g)\n",
           bad ? "FAILS" : "HOLDS", bad, at(0, 0, 0), at(0, 0, 1), at(0, 0, 2), at(0, 0, 3));
    printf("B. scale_a = 127 + lane %% 4 (op_sel 0): D[0..7][0] = ");
    for (int i = 0; i < 8; ++i) printf("%g ", at(1, i, 0));
    printf(" (64 = both K halves at 2^0)\n   scale_a bytes 127|128|129|130, op_sel 0..3: D[0][0] = %g %g %g %g  (64 128 256 512 if op_sel picks the byte)\n", at(2, 0, 0), at(3, 0, 0),
           at(4, 0, 0), at(5, 0, 0));
    rate<0>("C. f16 32x32x16");
    rate<1>("   fp6 x bf6 32x32x64");
    rate<2>("   fp4 x bf6 32x32x64");
    rate<3>("   fp4 x fp4 32x32x64");
    return 0;
}
