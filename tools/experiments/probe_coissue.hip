// probe_coissue.hip — does a VALU/LDS "partner" wave hide under a matrix wave on the same SIMD, and what makes it so?
// 512-thread workgroups: waves 0-3 issue MFMAs (one per SIMD), waves 4-7 are the partners.  Variants: s_nop padding
// after every MFMA in the matrix wave (leaves the VALU issue port to the partner while the matrix pipe is busy),
// s_setprio on either role, number of independent accumulator chains, partner instruction kind.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e = (x);                                                                   \
        if (e != hipSuccess) {                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);      \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

template <int PAD>
__device__ __forceinline__ void pad() {
    if (PAD >= 16) asm volatile("s_nop 15");
    if (PAD >= 32) asm volatile("s_nop 15");
    if (PAD % 16 > 0) asm volatile("s_nop %0" ::"n"(PAD % 16 > 0 ? PAD % 16 - 1 : 0));
}

// PAD: wait states inserted after every MFMA; CHAINS: independent accumulators; KIND: partner work
// KIND 0 v_fma; 1 cvt_pk_bf16 + sub (operand split); 2 ds_read_b128 + 4 fma; 3 global_load_dwordx4 (L2) + 4 fma
template <int PAD, int CHAINS, int KIND>
__global__ __launch_bounds__(512) void k(float *out, const float *src, int n_mfma_iters, int n_part_iters, int per_iter, int prio_m,
                                         int prio_p) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = (float)threadIdx.x;
    lds[threadIdx.x + 512] = 1.f;
    __syncthreads();
    if (wave < 4) {
        if (n_mfma_iters <= 0) return;
        if (prio_m == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_m == 2) __builtin_amdgcn_s_setprio(2);
        if (prio_m == 3) __builtin_amdgcn_s_setprio(3);
        v16f c[4] = {};
        b8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (__bf16)(float)(lane + i);
            b[i] = (__bf16)(float)(lane - i);
        }
        for (int it = 0; it < n_mfma_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                c[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u % CHAINS], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                pad<PAD>();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        if (n_part_iters <= 0) return;
        if (prio_p == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_p == 2) __builtin_amdgcn_s_setprio(2);
        if (prio_p == 3) __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
        const float m = 1.0001f, ad = 0.5f;
        for (int it = 0; it < n_part_iters; ++it) {
            for (int v = 0; v < per_iter; v += 8) {
                if (KIND == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], m, ad);
                } else if (KIND == 1) {
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {  // 2 values: cvt_pk, 2 shifts, 2 subs, (cvt_pk) ~ 8 VALU per 4 values
                        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                        const b2 hp = {(__bf16)x[i], (__bf16)x[i + 1]};
                        const unsigned hw = __builtin_bit_cast(unsigned, hp);
                        x[i + 4] += x[i] - __uint_as_float(hw << 16);
                        x[i + 5] += x[i + 1] - __uint_as_float(hw & 0xffff0000u);
                    }
                } else if (KIND == 2) {
                    const float4 q = *reinterpret_cast<const float4 *>(&lds[((it + v) * 64 + lane * 4) & 8188]);
                    x[0] = fmaf(x[0], m, q.x);
                    x[1] = fmaf(x[1], m, q.y);
                    x[2] = fmaf(x[2], m, q.z);
                    x[3] = fmaf(x[3], m, q.w);
                } else {
                    const float4 q = *reinterpret_cast<const float4 *>(&src[(((it * 131 + v) * 64 + lane) * 4) & 0xffffc]);
                    x[0] = fmaf(x[0], m, q.x);
                    x[1] = fmaf(x[1], m, q.y);
                    x[2] = fmaf(x[2], m, q.z);
                    x[3] = fmaf(x[3], m, q.w);
                }
            }
        }
        float acc = 0.f;
        for (int i = 0; i < 8; ++i) acc += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = acc;
    }
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

static float *out, *src;
const int NM = 2048;  // x16 MFMAs per matrix wave

template <int PAD, int CHAINS, int KIND>
static void run(const char *tag, int per_slot, int prio_m, int prio_p) {
    const int per = 16 * per_slot;
    const float tm = time_ms([&] { hipLaunchKernelGGL((k<PAD, CHAINS, KIND>), dim3(256), dim3(512), 0, 0, out, src, NM, 0, per, prio_m, prio_p); });
    const float tp = time_ms([&] { hipLaunchKernelGGL((k<PAD, CHAINS, KIND>), dim3(256), dim3(512), 0, 0, out, src, 0, NM, per, prio_m, prio_p); });
    const float tb = time_ms([&] { hipLaunchKernelGGL((k<PAD, CHAINS, KIND>), dim3(256), dim3(512), 0, 0, out, src, NM, NM, per, prio_m, prio_p); });
    printf("%-34s pad %2d chains %d prio m%d/p%d, %d per MFMA slot: matrix alone %.3f | partner alone %.3f | both %.3f ms  (hidden %.0f %%)\n", tag, PAD,
           CHAINS, prio_m, prio_p, per_slot, tm, tp, tb, 100.f * (tm + tp - tb) / (tp < tm ? tp : tm));
}

int main() {
    CK(hipMalloc(&out, 1 << 22));
    CK(hipMalloc(&src, 1 << 22));
    CK(hipMemset(src, 0, 1 << 22));
    printf("256 groups x 512 threads; matrix waves: %d x 16 MFMA 32x32x16 bf16\n", NM);
    run<0, 4, 0>("v_fma", 4, 0, 0);
    run<8, 4, 0>("v_fma", 4, 0, 0);
    run<16, 4, 0>("v_fma", 4, 0, 0);
    run<20, 4, 0>("v_fma", 4, 0, 0);
    run<24, 4, 0>("v_fma", 4, 0, 0);
    run<28, 4, 0>("v_fma", 4, 0, 0);
    run<24, 4, 0>("v_fma", 5, 0, 0);
    run<24, 4, 0>("v_fma", 6, 0, 0);
    run<20, 4, 0>("v_fma", 6, 0, 0);
    run<0, 2, 0>("v_fma", 4, 0, 0);
    run<0, 1, 0>("v_fma", 4, 0, 0);
    run<0, 4, 0>("v_fma", 4, 0, 1);
    run<0, 4, 0>("v_fma", 4, 0, 3);
    run<0, 4, 0>("v_fma", 4, 1, 0);
    run<24, 4, 0>("v_fma", 4, 0, 1);
    run<24, 4, 0>("v_fma", 4, 1, 0);
    run<0, 4, 1>("operand split (cvt_pk/shift/sub)", 4, 0, 0);
    run<24, 4, 1>("operand split (cvt_pk/shift/sub)", 4, 0, 0);
    run<0, 4, 2>("ds_read_b128 + 4 fma", 4, 0, 0);
    run<24, 4, 2>("ds_read_b128 + 4 fma", 4, 0, 0);
    run<0, 4, 3>("global_load_dwordx4 + 4 fma", 4, 0, 0);
    run<24, 4, 3>("global_load_dwordx4 + 4 fma", 4, 0, 0);
    return 0;
}
