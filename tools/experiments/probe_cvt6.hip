// probe_cvt6.hip — semantics and cost of v_cvt_scalef32_pk32_bf6_f32 (32 fp32 -> 32 bf6 e3m2 in 6 registers) on gfx950:
// element order, the role of the scale operand, rounding, saturation; and its issue time against 16 v_cvt_pk_bf8_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_sem(const float *in, float scale, unsigned *out, int kind) {
    v16f a, b;
    v32h hh;
    for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
    for (int i = 0; i < 32; ++i) hh[i] = (_Float16)in[i];
    v6u r;
    if (kind == 0) r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, scale);
    else r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(hh, scale);
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}
template <int KIND>
__global__ __launch_bounds__(256) void k_time(float *out, int iters) {
    v32f v;
    v16f a, b;
    v32h hh;
    for (int i = 0; i < 32; ++i) { v[i] = (float)(threadIdx.x + i); hh[i] = (_Float16)v[i]; }
    for (int i = 0; i < 16; ++i) { a[i] = v[i]; b[i] = v[16 + i]; }
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            const v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, 4.0f);
            acc ^= r[0] ^ r[5];
            a[0] += 1.f;
        } else if (KIND == 2) {
            const v6u r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(hh, 4.0f);
            acc ^= r[0] ^ r[5];
            hh[0] += (_Float16)1.f;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc ^= (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(v[2 * i], v[2 * i + 1], 0, false);
            v[0] += 1.f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}
static float decode_bf6(unsigned c) {  // e3m2, bias 3
    const int s = c >> 5, e = (c >> 2) & 7, m = c & 3;
    const float v = e == 0 ? ldexpf((float)m / 4.f, -2) : ldexpf(1.f + m / 4.f, e - 3);
    return s ? -v : v;
}
int main() {
    float h[32];
    for (int i = 0; i < 32; ++i) h[i] = (i % 2 ? -1.f : 1.f) * (0.05f + 0.9f * i);
    h[30] = 100.f; h[31] = 0.3f;
    float *din; unsigned *dout; float *tout;
    CK(hipMalloc(&din, 128)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&tout, 1 << 22));
    CK(hipMemcpy(din, h, 128, hipMemcpyHostToDevice));
    for (int kind = 0; kind < 2; ++kind)
    for (float scale : {1.0f, 4.0f, 0.25f}) {
        printf("%s\n", kind == 0 ? "v_cvt_scalef32_2xpk16_bf6_f32 (in[0..15], in[16..31])" : "v_cvt_scalef32_pk32_bf6_f16");
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(1), 0, 0, din, scale, dout, kind);
        unsigned r[6]; CK(hipMemcpy(r, dout, 24, hipMemcpyDeviceToHost));
        printf("scale operand %.2f:\n", scale);
        for (int i = 0; i < 32; ++i) {
            const int bit = 6 * i;
            unsigned long long two = r[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? r[bit / 32 + 1] : 0) << 32);
            const unsigned c = (two >> (bit % 32)) & 63;
            printf("  in % 8.3f -> element %2d code %2u = % 7.4f (x scale = % 8.3f, / scale = % 8.3f)\n", h[i], i, c, decode_bf6(c), decode_bf6(c) * scale, decode_bf6(c) / scale);
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 3; ++kind) {
        auto go = [&] { if (kind == 0) hipLaunchKernelGGL(k_time<0>, dim3(1024), dim3(256), 0, 0, tout, 4096); else if (kind == 1) hipLaunchKernelGGL(k_time<1>, dim3(1024), dim3(256), 0, 0, tout, 4096); else hipLaunchKernelGGL(k_time<2>, dim3(1024), dim3(256), 0, 0, tout, 4096); };
        go(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f ms for 4096 iterations x 4 waves/CU-SIMD... (%.1f ns per 32 values per wave-slot)\n", kind == 0 ? "1 x cvt_scalef32_2xpk16_bf6_f32" : kind == 1 ? "16 x cvt_pk_bf8_f32" : "1 x cvt_scalef32_pk32_bf6_f16", ms, ms * 1e6 / 4096 / 4);
    }
    return 0;
}
