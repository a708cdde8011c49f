// probe_f16ovfl.hip — does MODE.FP16_OVFL (bit 23 of HW_REG_MODE on gfx9-family ISAs) make v_cvt_pk_f16_f32 and
// v_cvt_f16_f32 saturate to +-65504 instead of returning inf on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int set) {
    if (set) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    float a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
    unsigned r, s;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s) : "v"(a));
    out[threadIdx.x * 2] = r;
    out[threadIdx.x * 2 + 1] = s;
}
int main() {
    float h[8] = {1.0f, -2.5f, 65504.f, 70000.f, 1e6f, -1e9f, __builtin_inff(), 65520.f};
    float *d; unsigned *o, ho[8];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int set = 0; set < 2; ++set) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, set);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", set);
        for (int i = 0; i < 4; ++i) printf("  pk(%g, %g) = %04x %04x  cvt = %04x |", h[2 * i], h[2 * i + 1], ho[2 * i] & 0xffff, ho[2 * i] >> 16, ho[2 * i + 1] & 0xffff);
        printf("\n");
    }
    printf("(7bff = 65504, 7c00 = inf, fbff = -65504, fc00 = -inf)\n");
    return 0;
}
