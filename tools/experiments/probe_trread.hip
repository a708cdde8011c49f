// probe_trread.hip — semantics of ds_read_b64_tr_b16 on gfx950, as needed for K-major MFMA operands from a row-major LDS image:
// every lane passes the address of 4 contiguous 16-bit elements (8 B); within each group of 16 lanes the 16 x 4 block is
// transposed: which elements does lane l receive?  The image is img[row][64 cols] with img[r][c] = 100 r + c.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(int *out) {
    __shared__ short img[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (short)(100 * (i / 64) + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, l15 = l & 15, g = l >> 4;
    // my guess for an MFMA 32x32x16 operand: channels 16 (g & 1) + l15, rows 8 (g >> 1) + 0..3
    const int row = 8 * (g >> 1) + (l15 >> 2), col = 16 * (g & 1) + 4 * (l15 & 3);
    const unsigned addr = (unsigned)(size_t)(&img[row * 64 + col]);
    s4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    int *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" r%d c%d |", h[l * 4 + j] / 100, h[l * 4 + j] % 100);
        printf("   want col %d rows %d..%d\n", 16 * ((l >> 4) & 1) + (l & 15), 8 * (l >> 5), 8 * (l >> 5) + 3);
    }
    return 0;
}
