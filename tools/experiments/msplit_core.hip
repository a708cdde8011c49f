// Experiment (not part of the product): MLP core of an "M-split" march kernel.
//   workgroup = 4 waves = 64 samples (2 N-tiles of 32); wave w owns a quarter of every layer's OUTPUT features;
//   activations live in LDS as ready-made MFMA B fragments (bf16 hi + lo), rewritten in place after every layer
//   (2 barriers per layer); weights stream straight from L2 into a register prefetch ring (no LDS ring, no DMA);
//   <= 256 VGPRs so that TWO workgroups share a CU and one wave's VALU/LDS phases hide under the other's MFMAs.
// Measures: time for 262144 x 64 samples of the 5-layer trunk (972 MFMAs per wave per 64-sample step).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

#ifndef RING
#define RING 12
#endif
constexpr int R = RING;
constexpr int NCH0 = 22, NCH = 16, NCHV = 22;                       // K chunks of fc_0, hidden layers, view layer
constexpr int F_TOTAL = 2 * (2 * (NCH0 + 3 * NCH) + NCHV);           // fragments per wave per step = 324
static_assert(F_TOTAL % R == 0, "ring must divide the stream");
constexpr int ACT_BYTES = 16 * 2 * 2048;                             // 16 chunks x 2 n-tiles x (hi 1 KiB + lo 1 KiB)

struct Args {
    const char *w;   // [4 waves][F_TOTAL][1 KiB]
    float *out;      // [n_wg * 256]
    int steps;
};

__device__ __forceinline__ void split4(const float *v, u32x2 &h, u32x2 &l) {
    __bf16 hh[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = fmaxf(v[i], 0.f);
        hh[i] = (__bf16)x;
        ll[i] = (__bf16)(x - (float)hh[i]);
    }
    h = __builtin_bit_cast(u32x2, *reinterpret_cast<__bf16(*)[4]>(hh));
    l = __builtin_bit_cast(u32x2, *reinterpret_cast<__bf16(*)[4]>(ll));
}

// one layer for this wave: MT m-tiles x 2 n-tiles, NC chunks; weights = fragments F0.. of the wave's stream
#ifndef SIDE
#define SIDE 0
#endif
template <int F0, int MT, int NC>
__device__ __forceinline__ void layer(const char *wl, char *act, int lane, bf16x8 (&ring)[R], f32x16 (&acc)[2][2],
                                      float (&side)[8], float sx) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.01f * (float)(r + m);
#ifdef PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    bf16x8 bh[2][2], bl[2][2];
    auto rdB = [&](int c, int buf) {
        const char *p = act + ((c % 16) * 2) * 2048 + lane * 16;
        bh[buf][0] = *reinterpret_cast<const bf16x8 *>(p);
        bl[buf][0] = *reinterpret_cast<const bf16x8 *>(p + 1024);
        bh[buf][1] = *reinterpret_cast<const bf16x8 *>(p + 2048);
        bl[buf][1] = *reinterpret_cast<const bf16x8 *>(p + 3072);
    };
    rdB(0, 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cur = c & 1;
        if (c + 1 < NC) rdB(c + 1, cur ^ 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int f = F0 + 2 * (c * MT + m);
            const bf16x8 ah = ring[f % R], al = ring[(f + 1) % R];
#if defined(NOLOAD)
            asm volatile("" : "+v"(ring[f % R]), "+v"(ring[(f + 1) % R]));
#elif defined(WSMALL)
            ring[f % R] = *reinterpret_cast<const bf16x8 *>(wl + (size_t)((f + R) % 8) * 1024);
            ring[(f + 1) % R] = *reinterpret_cast<const bf16x8 *>(wl + (size_t)((f + 1 + R) % 8) * 1024);
#else
            ring[f % R] = *reinterpret_cast<const bf16x8 *>(wl + (size_t)((f + R) % F_TOTAL) * 1024);
            ring[(f + 1) % R] = *reinterpret_cast<const bf16x8 *>(wl + (size_t)((f + 1 + R) % F_TOTAL) * 1024);
#endif
#ifdef HAND
#define FILL(k)                                                                                                    \
    {                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        _Pragma("unroll") for (int q = 0; q < HAND; ++q)                                                           \
            asm volatile("v_fma_f32 %0, %0, %1, %2"                                                               \
                         : "+v"(side[((k) * HAND + q) & 7])                                                        \
                         : "v"(sx), "v"(side[((k) * HAND + q + 3) & 7]));                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
            acc[m][0] = MFMA(ah, bh[cur][0], acc[m][0]);
            FILL(0)
            acc[m][1] = MFMA(ah, bh[cur][1], acc[m][1]);
            FILL(1)
            acc[m][0] = MFMA(ah, bl[cur][0], acc[m][0]);
            FILL(2)
            acc[m][1] = MFMA(ah, bl[cur][1], acc[m][1]);
            FILL(3)
            acc[m][0] = MFMA(al, bh[cur][0], acc[m][0]);
            FILL(4)
            acc[m][1] = MFMA(al, bh[cur][1], acc[m][1]);
            FILL(5)
#else
            acc[m][0] = MFMA(ah, bh[cur][0], acc[m][0]);
            acc[m][1] = MFMA(ah, bh[cur][1], acc[m][1]);
            acc[m][0] = MFMA(ah, bl[cur][0], acc[m][0]);
            acc[m][1] = MFMA(ah, bl[cur][1], acc[m][1]);
            acc[m][0] = MFMA(al, bh[cur][0], acc[m][0]);
            acc[m][1] = MFMA(al, bh[cur][1], acc[m][1]);
#pragma unroll
            for (int q = 0; q < SIDE; ++q) side[q & 7] = fmaf(side[q & 7], sx, side[(q + 3) & 7]);
#ifdef INTERLEAVE
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, SIDE / 6, 0);
            }
#endif
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// relu + split + in-place rewrite of the activation fragments (features [64 w + 32 m, +32) of the next layer's K)
template <int MT>
__device__ __forceinline__ void publish(char *act, int lane, int wave, const f32x16 (&acc)[2][2]) {
    const int i = lane & 31, hi = lane >> 5;
#ifdef NOPUBLISH
    return;
#endif
#ifndef NOBAR
    __syncthreads();  // everyone is done reading the previous activations
#endif
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[m][n][4 * j + e];
                u32x2 h, l;
                split4(v, h, l);
                const int chunk = (MT == 2 ? 4 * wave + 2 * m : 2 * wave) + (j >> 1);
                char *p = act + (chunk * 2 + n) * 2048 + ((j & 1) * 32 + i) * 16 + 8 * hi;
                *reinterpret_cast<u32x2 *>(p) = h;
                *reinterpret_cast<u32x2 *>(p + 1024) = l;
            }
#ifndef NOBAR
    __syncthreads();
#endif
}

#ifndef WGPC
#define WGPC 2
#endif
__global__ __launch_bounds__(256, WGPC) void core_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *act = lds;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < ACT_BYTES / 4; i += 256) reinterpret_cast<unsigned *>(act)[i] = 0x3c003f80u + (i & 7);
    __syncthreads();
    const char *wl = a.w + (size_t)wave * F_TOTAL * 1024 + lane * 16;
    bf16x8 ring[R];
#pragma unroll
    for (int i = 0; i < R; ++i) ring[i] = *reinterpret_cast<const bf16x8 *>(wl + (size_t)i * 1024);
    f32x16 acc[2][2];
    float sum = 0.f;
    float side[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) side[q] = 0.001f * (float)(threadIdx.x + q);
    const float sx = 0.999f + 1e-6f * (float)(threadIdx.x & 3);
#ifdef STAGGER
    if ((blockIdx.x >> 8) & 1)
        for (int i = 0; i < STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    for (int s = 0; s < a.steps; ++s) {
        int zero = 0;
        asm volatile("" : "+s"(zero));
        const char *w2 = wl + zero;
        constexpr int FA = 0, FB = FA + 4 * NCH0, FC = FB + 4 * NCH, FD = FC + 4 * NCH, FE = FD + 4 * NCH;
        layer<FA, 2, NCH0>(w2, act, lane, ring, acc, side, sx);
        publish<2>(act, lane, wave, acc);
        layer<FB, 2, NCH>(w2, act, lane, ring, acc, side, sx);
        publish<2>(act, lane, wave, acc);
        layer<FC, 2, NCH>(w2, act, lane, ring, acc, side, sx);
        publish<2>(act, lane, wave, acc);
        layer<FD, 2, NCH>(w2, act, lane, ring, acc, side, sx);
        publish<2>(act, lane, wave, acc);
        layer<FE, 1, NCHV>(w2, act, lane, ring, acc, side, sx);
        publish<1>(act, lane, wave, acc);
        sum += acc[0][0][0] + acc[0][1][5];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += side[q];
    }
    a.out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char **argv) {
    const int n_wg = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 64, reps = 5;
    const size_t wbytes = (size_t)4 * F_TOTAL * 1024;
    std::vector<unsigned short> hw(wbytes / 2);
    unsigned x = 12345;
    for (auto &v : hw) {
        x = x * 1664525u + 1013904223u;
        v = (unsigned short)(0x3a00 + ((x >> 16) & 0x1ff) + ((x >> 31) << 15));  // small +- bf16 values
    }
    char *dw;
    float *dout;
    hipMalloc(&dw, wbytes);
    hipMalloc(&dout, (size_t)n_wg * 256 * 4);
    hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice);
    Args a{dw, dout, steps};
    hipFuncSetAttribute((const void *)core_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ACT_BYTES + 40960);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int r = 0; r < reps + 1; ++r) {
        if (r == 1) hipEventRecord(e0);
        hipLaunchKernelGGL(core_kernel, dim3(n_wg), dim3(256), ACT_BYTES + (WGPC == 1 ? 40960 : 0), 0, a);
    }
    hipEventRecord(e1);
    hipError_t err = hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double mfma = (double)n_wg * steps * 4 * 972;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, core_kernel, 256, ACT_BYTES);
    printf("ring %d: %s, %d WGs x %d steps: %.3f ms per launch; %.1f%% of the bf16 MFMA peak at 2.4 GHz (%.0f TFLOP/s executed); "
           "occupancy %d WG/CU\n", R, hipGetErrorString(err), n_wg, steps, ms,
           100.0 * mfma * 32 / (ms * 1e-3 * 2.4e9 * 1024), mfma * 32768 / (ms * 1e-3) / 1e12, occ);
    return 0;
}
