// probe_mx6sub.hip — does v_mfma_scale_f32_32x32x64_f8f6f4 honour SUBNORMAL fp6 e2m3 / bf6 e3m2 operands?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ i32x8 fill6(unsigned code) {  // 32 copies of a 6-bit code
    unsigned long long lo = 0, w[3];
    unsigned r[6] = {0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 32; ++e) {
        const int bit = 6 * e;
        r[bit >> 5] |= code << (bit & 31);
        if ((bit & 31) > 26) r[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
    }
    (void)lo; (void)w;
    return i32x8{(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0};
}
__global__ void k(float *out, unsigned ca, unsigned cb) {
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(ca), fill6(cb), c, 2, 3, 0, 127, 0, 127);
    if (threadIdx.x == 0) out[0] = c[0];
}
// MODE 0: both e2m3; 1: A e2m3 x B e3m2 with B nonzero only in lanes < 32; 2: the same, lanes >= 32; 3: A e3m2 x B e2m3;
// 4: both e3m2; 5: A fp8 e4m3 (1.0 = 0x38) x B e3m2
__global__ void k2(float *out, int mode) {
    v16f c = {};
    const int lane = threadIdx.x;
    if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(8), fill6(8), c, 2, 2, 0, 127, 0, 127);
    if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(8), fill6(lane < 32 ? 12 : 0), c, 2, 3, 0, 127, 0, 127);
    if (mode == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(8), fill6(lane >= 32 ? 12 : 0), c, 2, 3, 0, 127, 0, 127);
    if (mode == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(12), fill6(8), c, 3, 2, 0, 127, 0, 127);
    if (mode == 4) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(12), fill6(12), c, 3, 3, 0, 127, 0, 127);
    if (mode == 5) {
        const i32x8 a8 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, fill6(12), c, 0, 3, 0, 127, 0, 127);
    }
    if (mode == 6) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(8), fill6(12), c, 2, 3, 0, 127, 0, 127);
    if (mode == 7) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fill6(8), fill6(12), c, 2, 3, 0, 127 + (lane >> 8), 0, 127 + (lane >> 8));
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float *d, h;
    CK(hipMalloc(&d, 4));
    struct { unsigned ca, cb; const char *what; float expect; } t[] = {
        {8, 12, "A 1.0 (normal) x B 1.0 (normal)", 64.f},
        {4, 12, "A 0.5 (e2m3 SUBNORMAL 4/8) x B 1.0", 32.f},
        {1, 12, "A 0.125 (e2m3 subnormal 1/8) x B 1.0", 8.f},
        {8, 2, "A 1.0 x B 0.125 (e3m2 SUBNORMAL 2/4 * 2^-2)", 8.f},
        {8, 1, "A 1.0 x B 0.0625 (e3m2 subnormal 1/4 * 2^-2)", 4.f},
        {31, 31, "A 7.5 x B 28 (largest codes)", 64.f * 7.5f * 28.f},
    };
    for (auto &q : t) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, q.ca, q.cb);
        CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
        printf("%-52s D = %10.4f (expected %10.4f if the codes are honoured)\n", q.what, h, q.expect);
    }
    const char *names[] = {"A e2m3 1.0 x B e2m3 1.0", "A e2m3 1.0 x B e3m2 1.0 in lanes < 32 only", "A e2m3 1.0 x B e3m2 1.0 in lanes >= 32 only",
                           "A e3m2 1.0 x B e2m3 1.0", "A e3m2 1.0 x B e3m2 1.0", "A fp8 1.0 x B e3m2 1.0", "A e2m3 1.0 x B e3m2 1.0 (constants, literal scales)", "the same, scales from registers"};
    for (int m = 0; m < 8; ++m) {
        hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, m);
        CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
        printf("%-52s D = %10.4f\n", names[m], h);
    }
    return 0;
}
