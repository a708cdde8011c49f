// probe_x.hip — issue interval of the K=64 8-bit MFMA in the shapes the march uses: scaled vs unscaled opcode, 2 vs 4 accumulator
// chains, alternating with fp16 MFMAs like the record stream (8 main + 4 cross per block)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
// MODE 0: 12 cross MFMAs per iteration; 1: stream-like: (M M) x4, X X X X ; 2: 16 main only
template <int MODE, int SCALED, int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, int iters, int sa, int sb) {
    const int lane = threadIdx.x & 63;
    v16f c[4] = {};
    h8 a16, b16;
    v8i a8, b8;
    for (int i = 0; i < 8; ++i) { a16[i] = (_Float16)(float)(lane + i); b16[i] = (_Float16)(float)(lane - i); a8[i] = lane * 0x01010101 + i; b8[i] = lane * 0x01010101 - i; }
    auto X = [&](int ch) {
        if (SCALED) c[ch] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[ch], 0, 1, 0, sa, 0, sb);
        else c[ch] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[ch], 0, 1, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto M = [&](int ch) { c[ch] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, c[ch], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 12; ++u) X(u % CHAINS);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { M(0); M(1); }
            X(0); X(1); X(0); X(1);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) { M(0); M(1); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
static float *out;
template <int MODE, int SCALED, int CHAINS> static void run(const char *tag, int per_iter) {
    const int it = 4096;
    const float t = time_ms([&] { hipLaunchKernelGGL((k<MODE, SCALED, CHAINS>), dim3(256), dim3(256), 0, 0, out, it, 127, 115); });
    printf("%-46s scaled %d chains %d: %.3f ms = %.1f ns per iteration of %s\n", tag, SCALED, CHAINS, t, t * 1e6 / it, per_iter == 12 ? "12 X" : (per_iter == 1 ? "8 M + 4 X" : "16 M"));
}
int main() {
    CK(hipMalloc(&out, 1 << 22));
    run<2, 0, 2>("fp16 main only", 16);
    run<0, 1, 4>("cross only", 12);
    run<0, 1, 2>("cross only", 12);
    run<0, 1, 1>("cross only", 12);
    run<0, 0, 4>("cross only", 12);
    run<0, 0, 2>("cross only", 12);
    run<1, 1, 2>("record stream (8 main, 4 cross)", 1);
    run<1, 0, 2>("record stream (8 main, 4 cross)", 1);
    return 0;
}
