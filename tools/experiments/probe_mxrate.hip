// probe_mxrate.hip — issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 by operand format (one wave per SIMD, two alternating
// accumulators), against v_mfma_f32_32x32x16_f16.  Formats (cbsz = A, blgp = B): 0 fp8 e4m3, 1 bf8 e5m2, 2 fp6 e2m3, 3 bf6 e3m2, 4 fp4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int FA, int FB, bool SCALED>
__global__ __launch_bounds__(256) void k(float *out, int iters, int sa, int sb) {
    const int lane = threadIdx.x & 63;
    v16f c0 = {}, c1 = {};
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + lane * 0 ; b[i] = 0x38383838; }
    h8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)1.f; bh[i] = (_Float16)0.5f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (FA < 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
            } else if (SCALED) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, FA < 0 ? 0 : FA, FB < 0 ? 0 : FB, 0, sa, 0, sb);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, FA < 0 ? 0 : FA, FB < 0 ? 0 : FB, 0, sa, 0, sb);
            } else {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, FA < 0 ? 0 : FA, FB < 0 ? 0 : FB, 0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, FA < 0 ? 0 : FA, FB < 0 ? 0 : FB, 0, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1];
}
static float *out;
template <int FA, int FB, bool SCALED> static void run(const char *tag) {
    const int iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<FA, FB, SCALED>), dim3(256), dim3(256), 0, 0, out, iters, 127, 127); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<FA, FB, SCALED>), dim3(256), dim3(256), 0, 0, out, iters, 127, 127);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = 16.0 * iters;
    printf("%-34s %.3f ms  = %.1f ns per MFMA (= %.0f cycles at 2.4 GHz)\n", tag, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
    CK(hipMalloc(&out, 1 << 20));
    run<-1, -1, false>("f16 32x32x16");
    run<0, 1, true>("scaled fp8 x bf8");
    run<0, 1, false>("unscaled fp8 x bf8");
    run<2, 3, true>("scaled fp6 x bf6");
    run<2, 3, false>("unscaled fp6 x bf6");
    run<2, 1, true>("scaled fp6 x bf8");
    run<0, 3, true>("scaled fp8 x bf6");
    run<4, 4, true>("scaled fp4 x fp4");
    run<2, 2, true>("scaled fp6 x fp6");
    return 0;
}
