// probe_write.hip — round 6: what does rocprofv3's WRITE_SIZE report for the march's store patterns?  (MI355X_MICROARCH.md: WRITE_SIZE is
// uncalibrated on gfx950 — "calibrate on a known byte count in your own access pattern".)
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/probe_write.hip -o exp_bin/probe_write
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out -o w -- exp_bin/probe_write
// Every kernel writes exactly 64 MiB (A, B, C, D) or 1 MiB (E); the buffer is 256 MiB apart per kernel so that nothing is overwritten.
//   A  stream:    lane i writes 16 bytes at 16 i (whole 1-KiB bursts per wave instruction)
//   B  weights:   the march's pattern — 4 lanes of a ray write one 64-byte chunk (rays 256 bytes apart), the ray's four chunks in
//                 four separate instructions, all four at once
//   C  weights, chunks of a 128-byte line far apart in time: pass 1 writes chunks 0 and 2 of every ray, pass 2 (behind a grid-wide
//                 delay: another launch) chunks 1 and 3 — what the march does (a chunk every 16 depth steps)
//   D  ray-major 64-byte: one lane writes a ray's 64-byte chunk (4 x 16 bytes by 4 instructions)
//   E  scalars:   4 bytes per ray, 8 consecutive rays per 8 consecutive lanes (32-byte runs: the per-ray maps in tile order)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr long RAYS = 262144, S = 64;
__global__ void k_stream(f4 *o) { const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; o[i] = f4{1.f, 2.f, 3.f, (float)i}; }
__global__ void k_weights(float *o, int cmask) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, ray = t >> 2, part = t & 3;
    for (int c = 0; c < 4; ++c)
        if ((cmask >> c) & 1) *reinterpret_cast<f4 *>(o + ray * S + c * 16 + part * 4) = f4{1.f, 2.f, (float)c, (float)t};
}
__global__ void k_raymajor(float *o) {
    const long ray = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int c = 0; c < 4; ++c)
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f4 *>(o + ray * S + c * 16 + q * 4) = f4{1.f, (float)q, (float)c, (float)ray};
}
__global__ void k_scalar(float *o) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;  // tile order: lane l of a wave -> ray 512 (l / 8) + l % 8 of its tile
    const long tile = t >> 6, l = t & 63, ray = (tile >> 6) * 4096 + (l >> 3) * 512 + (tile & 63) * 8 + (l & 7);
    o[ray] = (float)t;
}
int main() {
    char *buf;
    const size_t MB = 1 << 20;
    CK(hipMalloc(&buf, 1400 * MB));
    CK(hipMemset(buf, 0, 1400 * MB));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(64 * MB / 16 / 256), dim3(256), 0, 0, reinterpret_cast<f4 *>(buf));
        hipLaunchKernelGGL(k_weights, dim3(RAYS * 4 / 256), dim3(256), 0, 0, reinterpret_cast<float *>(buf + 256 * MB), 15);
        hipLaunchKernelGGL(k_weights, dim3(RAYS * 4 / 256), dim3(256), 0, 0, reinterpret_cast<float *>(buf + 512 * MB), 5);
        CK(hipDeviceSynchronize());
        CK(hipMemset(buf + 1280 * MB, 1, 100 * MB));  // something else through the caches in between
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_weights, dim3(RAYS * 4 / 256), dim3(256), 0, 0, reinterpret_cast<float *>(buf + 512 * MB), 10);
        hipLaunchKernelGGL(k_raymajor, dim3(RAYS / 256), dim3(256), 0, 0, reinterpret_cast<float *>(buf + 768 * MB));
        hipLaunchKernelGGL(k_scalar, dim3(RAYS / 256), dim3(256), 0, 0, reinterpret_cast<float *>(buf + 1024 * MB));
        CK(hipDeviceSynchronize());
    }
    printf("done: k_stream 64 MiB, k_weights(15) 64 MiB, k_weights(5) + k_weights(10) 32 + 32 MiB, k_raymajor 64 MiB, k_scalar 1 MiB per repetition\n");
    return 0;
}
