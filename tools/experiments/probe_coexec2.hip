// probe_coexec2.hip — round 3: who can execute VALU instructions while a 32x32x16 MFMA occupies a SIMD's matrix pipe?
// Every instruction is inline asm (no SLP packing, no reordering).  512-thread workgroups = 2 waves per SIMD; one workgroup
// per CU (LDS pad).  Arms:
//   A  matrix waves only: [MFMA + OWN own v_fma_f32] x N                       (OWN = 0, 2, 4, 6, 8)
//   B  partner waves only: PART v_fma_f32 per slot x N
//   C  both at once: the matrix waves' time is what matters (do the partner's instructions fit in its shadow?)
//   D  matrix waves with s_nop padding after the MFMA instead of own fillers, beside the partner
// Output: ns per MFMA slot for each arm (32 cycles at 2.4 GHz = 13.3 ns).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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

template <int N>
__device__ __forceinline__ void fmas(float (&x)[8]) {  // N independent-ish v_fma_f32 (8 chains)
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(1.0001f), "v"(0.5f));
}
template <int N>
__device__ __forceinline__ void nops() {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("s_nop 3");  // 4 wait states each
}

// OWN: own v_fma per MFMA (matrix waves); NOPS: s_nop 3 per MFMA; PART: v_fma per slot (partner waves)
template <int OWN, int NOPS, int PART>
__global__ __launch_bounds__(512) void k(float *out, int n_matrix, int n_partner) {
    __shared__ float pad_lds[36 * 1024];  // > 80 KiB: one workgroup per CU
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) pad_lds[0] = 1.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    if (wave < 4) {
        v16f c[4] = {};
        h8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)(float)(lane & 7);
            b[i] = (_Float16)(float)(i);
        }
        for (int it = 0; it < n_matrix; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u & 3]) : "v"(a), "v"(b));
                fmas<OWN>(x);
                nops<NOPS>();
            }
        }
        float s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = s + pad_lds[0];
    } else {
        for (int it = 0; it < n_partner; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) fmas<PART>(x);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
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

static float *out;
const int NM = 4096;  // x16 slots per wave

template <int OWN, int NOPS, int PART>
static void run() {
    const float tm = time_ms([&] { hipLaunchKernelGGL((k<OWN, NOPS, PART>), dim3(256), dim3(512), 0, 0, out, NM, 0); });
    const float tp = time_ms([&] { hipLaunchKernelGGL((k<OWN, NOPS, PART>), dim3(256), dim3(512), 0, 0, out, 0, NM); });
    const float tb = time_ms([&] { hipLaunchKernelGGL((k<OWN, NOPS, PART>), dim3(256), dim3(512), 0, 0, out, NM, NM); });
    const double slot = 1e6 / (NM * 16.0);
    printf("own %d  nops %d  partner %d per slot:  matrix alone %6.2f ns/slot | partner alone %6.2f | both %6.2f | sum %6.2f  -> %s\n", OWN,
           NOPS, PART, tm * slot, tp * slot, tb * slot, (tm + tp) * slot,
           tb < 0.5 * (tm + tp) + 0.5 * (tm > tp ? tm : tp) ? "mostly OVERLAPPED" : "mostly serial");
}

int main() {
    CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    printf("one 32x32x16 f16 MFMA = 32 cycles = 13.3 ns at 2.4 GHz (16.6 at 1.93)\n");
    printf("-- own fillers, no partner work measured separately (A, B, C with partner 6 per slot)\n");
    run<0, 0, 6>();
    run<2, 0, 6>();
    run<4, 0, 6>();
    run<6, 0, 6>();
    run<8, 0, 6>();
    run<12, 0, 6>();
    printf("-- partner load sweep beside a bare MFMA stream\n");
    run<0, 0, 2>();
    run<0, 0, 4>();
    run<0, 0, 8>();
    run<0, 0, 12>();
    printf("-- s_nop padding in the matrix wave (4 wait states each) beside 6 partner v_fma per slot\n");
    run<0, 1, 6>();
    run<0, 2, 6>();
    run<0, 3, 6>();
    run<0, 4, 6>();
    run<0, 6, 6>();
    printf("-- own fillers AND partner\n");
    run<4, 0, 4>();
    run<6, 0, 6>();
    return 0;
}
