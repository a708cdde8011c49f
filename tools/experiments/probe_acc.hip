// probe_acc.hip — do VALU fillers hide behind MFMAs as a function of (a) how many independent accumulator chains the MFMAs
// rotate through, (b) accumulators pinned to AGPRs or VGPRs, (c) fillers written as C++ fmaf or as asm volatile?
// One wave per SIMD (256 threads, 512 groups), 2 x 2048 x 16 fp16 32x32x16 MFMAs per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC, int K, int AGPR, int ASMFILL, int ACCREAD = 0>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    v16f c[4] = {};
    h8 a, b;
    float x[8];
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)(lane + i); b[i] = (_Float16)(float)(lane - i); x[i] = (float)(lane + i); }
    v16f spare;  // a finished tile that no MFMA of the loop touches (ACCREAD: read it AGPR -> VGPR behind every MFMA)
    for (int i = 0; i < 16; ++i) spare[i] = (float)(lane * i);
    asm volatile("" : "+a"(spare));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            c[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u % NACC], 0, 0, 0);
            if (AGPR) asm volatile("" : "+a"(c[u % NACC]));
            __builtin_amdgcn_sched_barrier(0);
            if (ACCREAD == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[u % 8]) : "a"(spare[u % 16]));
            if (ACCREAD == 2) asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=&v"(x[u % 8]), "=&v"(x[(u + 1) % 8]) : "a"(spare[u % 16]), "a"(spare[(u + 5) % 16]));
            if (ACCREAD == 3) asm volatile("s_waitcnt lgkmcnt(6)\n\ts_nop 0" ::: "memory");
            if (ACCREAD == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_fma_mix_f32 %1, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=&v"(x[u % 8]), "+v"(x[(u + 1) % 8]) : "v"(x[(u + 2) % 8]));
#pragma unroll
            for (int f = 0; f < K; ++f) {
                const int i = (u * K + f) % 8;
                if (ASMFILL) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(x[(i + 3) % 8]));
                else x[i] = fmaf(x[i], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float acc = 0.f;
    for (int q = 0; q < NACC; ++q) acc += c[q][q];
    for (int i = 0; i < 8; ++i) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
static float *out;
template <int NACC, int K, int AGPR, int ASMFILL, int ACCREAD = 0> static void run() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&] { hipLaunchKernelGGL((k<NACC, K, AGPR, ASMFILL, ACCREAD>), dim3(512), dim3(256), 0, 0, out, 2048); };
    go(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("chains %d, %d fillers/MFMA, acc in %s, fillers %s, extra %d: %.3f ms\n", NACC, K, AGPR ? "AGPR" : "compiler's choice", ASMFILL ? "asm" : "C++", ACCREAD, ms);
}
int main() {
    CK(hipMalloc(&out, 1 << 22));
    run<4, 0, 0, 0>(); run<4, 5, 0, 0>(); run<4, 8, 0, 0>();
    run<2, 0, 0, 0>(); run<2, 5, 0, 0>(); run<2, 8, 0, 0>();
    run<1, 0, 0, 0>(); run<1, 5, 0, 0>();
    run<2, 0, 1, 0>(); run<2, 5, 1, 0>(); run<2, 8, 1, 0>();
    run<4, 0, 1, 0>(); run<4, 5, 1, 0>();
    run<2, 5, 1, 1>(); run<4, 5, 1, 1>(); run<2, 5, 0, 1>();
    printf("extra behind every MFMA: 1 = one v_accvgpr_read of an idle AGPR tile, 2 = two, 3 = s_waitcnt lgkmcnt + s_nop, 4 = cvt_pk_f16 + fma_mix\n");
    run<2, 0, 1, 1, 1>(); run<2, 0, 1, 1, 2>(); run<2, 4, 1, 1, 1>(); run<2, 4, 1, 1, 2>(); run<2, 4, 1, 1, 3>(); run<2, 0, 1, 1, 3>(); run<2, 3, 1, 1, 4>(); run<2, 0, 1, 1, 4>();
    return 0;
}
