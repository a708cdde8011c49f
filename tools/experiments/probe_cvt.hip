// probe_cvt.hip — semantics of the fp8 / bf8 conversion instructions used by the f16f8 decoder (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, float *o, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float a = x[i];
    // plain bf8 / fp8 packs (word 0), decoded back
    const int wb = __builtin_amdgcn_cvt_pk_bf8_f32(a, a, 0, false);
    const int wf = __builtin_amdgcn_cvt_pk_fp8_f32(a, a, 0, false);
    o[i * 6 + 0] = __builtin_amdgcn_cvt_f32_bf8(wb, 0);
    o[i * 6 + 1] = __builtin_amdgcn_cvt_f32_fp8(wf, 0);
    // scaled packs with scale = 4096 and 1/4096
    s2 old = {0, 0};
    const s2 r1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(old, a, a, 4096.0f, false);
    const s2 r2 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(old, a, a, 1.0f / 4096.0f, false);
    o[i * 6 + 2] = __builtin_amdgcn_cvt_f32_bf8(__builtin_bit_cast(int, r1), 0);
    o[i * 6 + 3] = __builtin_amdgcn_cvt_f32_bf8(__builtin_bit_cast(int, r2), 0);
    const s2 r3 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, a, 0.25f, false);
    o[i * 6 + 4] = __builtin_amdgcn_cvt_f32_fp8(__builtin_bit_cast(int, r3), 0);
    const s2 r4 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(old, a, 2 * a, 1.0f, true);  // word 1
    o[i * 6 + 5] = __builtin_amdgcn_cvt_f32_bf8(__builtin_bit_cast(int, r4), 2) + 1000.f * __builtin_amdgcn_cvt_f32_bf8(__builtin_bit_cast(int, r4), 3);
}
int main() {
    const int n = 12;
    float hx[n] = {1.0f, 1.3f, -2.7f, 1e-3f, 3e-5f, 447.f, 449.f, 1000.f, 60000.f, 1e6f, 0.f, 2.44140625e-4f};
    float *dx, *dout, ho[n * 6];
    hipMalloc(&dx, sizeof(hx));
    hipMalloc(&dout, sizeof(ho));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    printf("%12s | %12s %12s | %14s %14s | %12s | %s\n", "x", "bf8(x)", "fp8(x)", "sbf8(x,4096)", "sbf8(x,1/4096)", "sfp8(x,.25)", "word1: bf8(2x)+1000*byte3");
    for (int i = 0; i < n; ++i)
        printf("%12g | %12g %12g | %14g %14g | %12g | %g\n", hx[i], ho[i * 6], ho[i * 6 + 1], ho[i * 6 + 2], ho[i * 6 + 3], ho[i * 6 + 4], ho[i * 6 + 5]);
    return 0;
}
