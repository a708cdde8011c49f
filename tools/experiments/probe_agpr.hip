// probe_agpr.hip — does it matter for the MFMA issue rate where the accumulator (C/D) and the B operand live (VGPR vs AGPR)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int ACC_A, int B_A, int X>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    v16f c0 = {}, c1 = {};
    h8 a16, b16;
    v8i a8, b8;
    for (int i = 0; i < 8; ++i) { a16[i] = (_Float16)(float)(lane + i); b16[i] = (_Float16)(float)(lane - i); a8[i] = lane + i; b8[i] = lane - i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!X) {
                if (ACC_A && B_A) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a16), "a"(b16)); asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a16), "a"(b16)); }
                else if (ACC_A) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a16), "v"(b16)); asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a16), "v"(b16)); }
                else if (B_A) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a16), "a"(b16)); asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a16), "a"(b16)); }
                else { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a16), "v"(b16)); asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a16), "v"(b16)); }
            } else {
                if (ACC_A && B_A) { asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 blgp:1" : "+a"(c0) : "v"(a8), "a"(b8)); asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 blgp:1" : "+a"(c1) : "v"(a8), "a"(b8)); }
                else if (ACC_A) { asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 blgp:1" : "+a"(c0) : "v"(a8), "v"(b8)); asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 blgp:1" : "+a"(c1) : "v"(a8), "v"(b8)); }
                else { asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 blgp:1" : "+v"(c0) : "v"(a8), "v"(b8)); asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 blgp:1" : "+v"(c1) : "v"(a8), "v"(b8)); }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
static float *out;
template <int ACC_A, int B_A, int X> static void run() {
    const float t = time_ms([&] { hipLaunchKernelGGL((k<ACC_A, B_A, X>), dim3(256), dim3(256), 0, 0, out, 4096); });
    printf("%s MFMA, accumulator in %s, B operand in %s: %.3f ms = %.2f ns per MFMA\n", X ? "K=64 8-bit" : "K=16 fp16 ", ACC_A ? "AGPR" : "VGPR", B_A ? "AGPR" : "VGPR", t, t * 1e6 / (4096 * 16));
}
int main() {
    CK(hipMalloc(&out, 1 << 22));
    run<0, 0, 0>(); run<1, 0, 0>(); run<0, 1, 0>(); run<1, 1, 0>();
    run<0, 0, 1>(); run<1, 0, 1>(); run<1, 1, 1>();
    return 0;
}
