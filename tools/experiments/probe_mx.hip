// probe_mx.hip — one-shot hardware probes behind the round-2 march design (run on the MI355X; hipcc --offload-arch=gfx950):
//   A. operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 and fp6 e2m3): row / column / K position of every
//      (lane, element) slot, how the E8M0 scale operands are applied;
//   B. fp16 MFMA: are fp16 subnormal inputs flushed?
//   C. co-issue: a matrix-only wave and a VALU-only wave on the same SIMD (512-thread workgroups) — do they overlap?
//      and two phase-alternating waves per SIMD vs one.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// ------------------------------------------------------------------ A. scaled MFMA, raw operands from memory
template <int FMT>
__global__ void mx_kernel(const v8i *a, const v8i *b, const int *sa, const int *sb, v16f *d, int n) {
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        v16f acc = {};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t * 64 + l], b[t * 64 + l], acc, FMT, FMT, 0, sa[t * 64 + l], 0,
                                                              sb[t * 64 + l]);
        d[t * 64 + l] = acc;
    }
}

static int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// element e (0..31) of a lane's operand: fp8 -> byte e; fp6 -> bits [6e, 6e+6)
static void put_elem(int fmt, unsigned *regs, int e, unsigned code) {
    if (fmt == 0) {
        regs[e >> 2] |= (code & 0xffu) << (8 * (e & 3));
    } else {
        const int bit = 6 * e;
        unsigned long long v = (unsigned long long)(code & 0x3fu) << (bit & 31);
        regs[bit >> 5] |= (unsigned)v;
        if ((bit & 31) > 26) regs[(bit >> 5) + 1] |= (unsigned)(v >> 32);
    }
}
// small integers: fp8 e4m3 (bias 7) / fp6 e2m3 (bias 1)
static unsigned enc_int(int fmt, int v) {
    if (v == 0) return 0;
    const unsigned s = v < 0;
    const int a = std::abs(v);
    if (fmt == 0) {  // e4m3: value = 2^(e-7) * (1 + m/8)
        int e = 0;
        while ((1 << (e + 1)) <= a) ++e;
        const int m = (a * 8 >> e) - 8;
        return (s << 7) | ((unsigned)(e + 7) << 3) | (unsigned)m;
    }
    // e2m3: e=0 subnormal m/8 ; e>=1: 2^(e-1) * (1 + m/8); integers 1..7 exactly: 1 -> e1 m0, 2 -> e2 m0, 3 -> e2 m4, 4 -> e3 m0, 5 e3 m2, 6 e3 m4, 7 e3 m6
    int e = 0;
    while ((1 << (e + 1)) <= a) ++e;
    const int m = (a * 8 >> e) - 8;
    return (s << 5) | ((unsigned)(e + 1) << 3) | (unsigned)m;
}

static void probe_layout(int fmt) {
    printf("---- scaled MFMA 32x32x64, format %s\n", fmt == 0 ? "fp8 e4m3" : "fp6 e2m3");
    // test list: t = 0: random matrices under hypothesis H1 (row = lane%32, k = 32*(lane/32) + e, same for B with col)
    //            t = 1..64: A one-hot at (lane 0 / lane 32, element e) with B's slot (h', e') carrying digit values -> pairing
    const int NT = 1 + 128 + 4;
    std::vector<unsigned> A(NT * 64 * 8, 0), B(NT * 64 * 8, 0);
    std::vector<int> SA(NT * 64, 127), SB(NT * 64, 127);
    std::vector<int> Am(32 * 64), Bm(64 * 32);
    srand(1);
    for (auto &x : Am) x = rand() % 7 - 3;
    for (auto &x : Bm) x = rand() % 7 - 3;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 32; ++e) {
            put_elem(fmt, &A[(0 * 64 + l) * 8], e, enc_int(fmt, Am[(l & 31) * 64 + 32 * (l >> 5) + e]));
            put_elem(fmt, &B[(0 * 64 + l) * 8], e, enc_int(fmt, Bm[(32 * (l >> 5) + e) * 32 + (l & 31)]));
        }
    for (int q = 0; q < 64; ++q)      // A slot (h = q/32, e = q%32) on row 0
        for (int pass = 0; pass < 2; ++pass) {
            const int t = 1 + q * 2 + pass;
            put_elem(fmt, &A[(t * 64 + 32 * (q >> 5)) * 8], q & 31, enc_int(fmt, 1));
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 32; ++e) {
                    const int id = 32 * (l >> 5) + e;  // B slot id
                    put_elem(fmt, &B[(t * 64 + l) * 8], e, enc_int(fmt, pass ? (id >> 3) : (id & 7)));
                }
        }
    // scale probes: all ones; t = 129: scale_a byte0 = 127 + (lane % 4) ; t = 130: scale_b likewise; t = 131: scale_a = 128 only on lane 5;
    // t = 132: scale_a byte1 = 130 (opsel 0 must ignore it)
    for (int t = 129; t < 133; ++t)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 32; ++e) {
                put_elem(fmt, &A[(t * 64 + l) * 8], e, enc_int(fmt, 1));
                put_elem(fmt, &B[(t * 64 + l) * 8], e, enc_int(fmt, 1));
            }
    for (int l = 0; l < 64; ++l) {
        SA[129 * 64 + l] = 127 + (l % 4);
        SB[130 * 64 + l] = 127 + (l % 4);
        SA[131 * 64 + l] = l == 5 ? 128 : 127;
        SA[132 * 64 + l] = 127 | (130 << 8);
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
    if (fmt == 0)
        hipLaunchKernelGGL(mx_kernel<0>, dim3(NT), dim3(64), 0, 0, (const v8i *)dA, (const v8i *)dB, dSA, dSB, (v16f *)dD, NT);
    else
        hipLaunchKernelGGL(mx_kernel<2>, dim3(NT), dim3(64), 0, 0, (const v8i *)dA, (const v8i *)dB, dSA, dSB, (v16f *)dD, NT);
    CK(hipDeviceSynchronize());
    std::vector<float> D(NT * 64 * 16);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    auto at = [&](int t, int i, int j) {  // D[i][j] of test t under the standard C layout
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
    printf("H1 (row = lane%%32, k = 32*(lane/32) + element, little-endian packing): %s (%d mismatches of 1024)\n",
           bad ? "FAILS" : "HOLDS", bad);
    int ident = 0;
    for (int q = 0; q < 64; ++q) {
        const int id = (int)at(1 + 2 * q, 0, 0) + 8 * (int)at(2 + 2 * q, 0, 0);
        if (id == q) ++ident;
        else printf("  A slot (h=%d,e=%d) pairs with B slot (h=%d,e=%d); row-0 check D[0][1]=%g D[1][0]=%g\n", q >> 5, q & 31, id >> 5, id & 31,
                    at(1 + 2 * q, 0, 1), at(1 + 2 * q, 1, 0));
    }
    printf("one-hot pairing: %d of 64 A slots pair with the same-numbered B slot\n", ident);
    printf("scale_a = 127 + lane%%4: D[0..7][0] =");
    for (int i = 0; i < 8; ++i) printf(" %g", at(129, i, 0));
    printf("   (64 = both K halves at 2^0)\n");
    printf("scale_b = 127 + lane%%4: D[0][0..7] =");
    for (int j = 0; j < 8; ++j) printf(" %g", at(130, 0, j));
    printf("\nscale_a = 128 on lane 5 only: D[4..6][0] = %g %g %g, D[5][7] = %g\n", at(131, 4, 0), at(131, 5, 0), at(131, 6, 0), at(131, 5, 7));
    printf("scale_a byte1 = 130, byte0 = 127, opsel 0: D[0][0] = %g\n", at(132, 0, 0));
    hipFree(dA);
    hipFree(dB);
    hipFree(dSA);
    hipFree(dSB);
    hipFree(dD);
}

// ------------------------------------------------------------------ B. fp16 subnormals through the MFMA
__global__ void f16_denorm_kernel(float *out) {
    h8 a = {}, b = {};
    const int l = threadIdx.x;
    if (l < 32) {
        a[0] = (_Float16)5.9604645e-8f * (_Float16)3.0f;  // 3 * 2^-24: subnormal
        b[0] = (_Float16)1024.0f;
    }
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (l == 0) out[0] = acc[0];
    b8 c = {}, d = {};
    if (l < 32) {
        c[0] = (__bf16)1e-39f;  // bf16 subnormal
        d[0] = (__bf16)1e30f;
    }
    acc = v16f{};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, acc, 0, 0, 0);
    if (l == 0) out[1] = acc[0];
}

// ------------------------------------------------------------------ C. co-issue
// role 0 (waves 0..3 of a 512-thread group, or all waves of a 256-thread group): MFMA chains; role 1: VALU work
template <int KIND>
__global__ __launch_bounds__(512) void coissue_kernel(float *out, int n_mfma_iters, int n_valu_iters, int valu_per_iter) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (wave < 4) {
        if (n_mfma_iters <= 0) return;
        v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
        b8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (__bf16)(float)(lane + i);
            b[i] = (__bf16)(float)(lane - i);
        }
        for (int it = 0; it < n_mfma_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
        const float m = 1.0001f, ad = 0.5f;
        float acc = 0.f;
        for (int it = 0; it < n_valu_iters; ++it) {
            if (KIND == 0) {
                for (int v = 0; v < valu_per_iter; v += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], m, ad);
                }
            } else if (KIND == 1) {  // conversion-like: cvt_pk_bf16 + sub + ds traffic
                for (int v = 0; v < valu_per_iter; v += 8) {
                    const float4 q = *reinterpret_cast<const float4 *>(&lds[((it + v) * 64 + lane * 4) & 8188]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const __bf16 h = (__bf16)x[i];
                        x[i + 4] += x[i] - (float)h;
                    }
                    x[0] += q.x;
                    x[1] += q.y;
                    x[2] += q.z;
                    x[3] += q.w;
                }
            }
        }
        for (int i = 0; i < 8; ++i) acc += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = acc;
    }
}

// two phase-alternating waves per SIMD: every wave runs [P MFMAs][Q VALU] per iteration
__global__ __launch_bounds__(512) void phased_kernel(float *out, int iters, int mfma_per_phase, int valu_per_phase, int stagger) {
    extern __shared__ float dyn_lds[];  // sized by the launch to limit the groups per CU
    if (iters < 0) dyn_lds[threadIdx.x] = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v16f c0 = {}, c1 = {};
    b8 a, b;
    float x[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(float)(lane + i);
        b[i] = (__bf16)(float)(lane - i);
        x[i] = (float)(lane + i);
    }
    const float m = 1.0001f, ad = 0.5f;
    if (stagger && wave >= 4) {  // start the second half of the group half a period later
        for (int v = 0; v < valu_per_phase; v += 8)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], m, ad);
    }
    for (int it = 0; it < iters; ++it) {
        for (int u = 0; u < mfma_per_phase; u += 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
        for (int v = 0; v < valu_per_phase; v += 8)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], m, ad);
        a[0] = (__bf16)x[0];
    }
    float acc = c0[0] + c1[1];
    for (int i = 0; i < 8; ++i) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
static float time_ms(F f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    probe_layout(0);
    probe_layout(2);
    float *out;
    CK(hipMalloc(&out, 1 << 24));
    hipLaunchKernelGGL(f16_denorm_kernel, dim3(1), dim3(64), 0, 0, out);
    float h[2];
    CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
    printf("---- subnormal inputs: fp16 MFMA (3*2^-24)*1024 = %g (expect 1.8310547e-4 if kept), bf16 MFMA 1e-39*1e30 = %g\n", h[0], h[1]);

    printf("---- co-issue, 256 workgroups x 512 threads (waves 0-3 matrix, 4-7 VALU), 16 MFMAs per matrix iteration\n");
    const int NM = 4096;  // 65536 MFMAs per matrix wave = 2.1 M cycles
    for (int kind = 0; kind < 2; ++kind)
        for (int vpm : {0, 2, 4, 5, 6, 8}) {  // VALU instructions per MFMA slot in the partner wave
            const int vper = 16 * (vpm ? vpm : 1) * (kind ? 1 : 1);
            auto both = [&] {
                if (kind == 0) hipLaunchKernelGGL(coissue_kernel<0>, dim3(256), dim3(512), 0, 0, out, NM, vpm ? NM : 0, vper);
                else hipLaunchKernelGGL(coissue_kernel<1>, dim3(256), dim3(512), 0, 0, out, NM, vpm ? NM : 0, vper);
            };
            auto valu_only = [&] {
                if (kind == 0) hipLaunchKernelGGL(coissue_kernel<0>, dim3(256), dim3(512), 0, 0, out, 0, vpm ? NM : 0, vper);
                else hipLaunchKernelGGL(coissue_kernel<1>, dim3(256), dim3(512), 0, 0, out, 0, vpm ? NM : 0, vper);
            };
            const float tb = time_ms(both), tv = vpm ? time_ms(valu_only) : 0.f;
            printf("kind %d (%s) %d per MFMA slot: both %.3f ms, partner waves alone %.3f ms\n", kind,
                   kind ? "cvt+sub+ds_read" : "v_fma", vpm, tb, tv);
        }
    printf("---- phase-alternating waves: [64 MFMAs][Q v_fma] x 1024, 256 groups; 1 wave/SIMD (256 thr) vs 2 waves/SIMD (512 thr, same total work per CU = half the groups)\n");
    for (int q : {0, 128, 256, 384, 512}) {
        const float t1 = time_ms([&] { hipLaunchKernelGGL(phased_kernel, dim3(512), dim3(256), 100 * 1024, 0, out, 1024, 64, q, 0); });
        const float t2 = time_ms([&] { hipLaunchKernelGGL(phased_kernel, dim3(256), dim3(512), 0, 0, out, 1024, 64, q, 0); });
        const float t3 = time_ms([&] { hipLaunchKernelGGL(phased_kernel, dim3(256), dim3(512), 0, 0, out, 1024, 64, q, 1); });
        printf("Q = %3d VALU per 64 MFMAs: 512 groups x 256 thr %.3f ms | 256 groups x 512 thr %.3f ms | staggered %.3f ms  (MFMA-only floor = Q 0)\n",
               q, t1, t2, t3);
    }
    return 0;
}
