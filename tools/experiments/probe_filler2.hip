// probe_filler2.hip — the real march loop shape: per record [2 ds_read_b128 two records ahead, s_waitcnt lgkmcnt(4), 2 MFMAs,
// K filler instructions], fenced with sched_barrier.  Which filler kinds hide behind the MFMAs, and where should they sit?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int SEL> __device__ __forceinline__ float rem16(float x, unsigned h) {
    float r;
    if (SEL == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
// KIND 0: none; 1: K v_fma; 2: one conversion slice of 2 values from VGPRs (9 instr); 3: the same reading its inputs from an AGPR tile
// (v_accvgpr_read of a finished accumulator); 4: 2 ds_read_b128 + 8 pk_fma (gather blend unit x2)
// POS 0: fillers after both MFMAs; 1: split between the two MFMAs
template <int KIND, int K, int POS, int LDSREAD>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    v16f c0 = {}, c1 = {}, done = {};
    for (int i = 0; i < 16; ++i) done[i] = (float)(lane + i);
    h8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(float)(lane - i);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    unsigned hacc = 0;
    int lacc = 0;
    i32x4 r[3][2];
    const int addr = lane * 16;
    for (int q = 0; q < 3; ++q) r[q][0] = r[q][1] = i32x4{lane, lane, lane, lane};
    auto filler = [&](int u, int part) {
        if (KIND == 1) {
#pragma unroll
            for (int f = 0; f < K / (POS ? 2 : 1); ++f) x[(u + f + 4 * part) % 8] = fmaf(x[(u + f + 4 * part) % 8], 1.0001f, 0.5f);
        } else if (KIND == 2 || KIND == 3) {
            if (POS && part == 1) return;  // one slice per record, placed after the first MFMA when POS = 1
            float v0, v1;
            if (KIND == 3) { v0 = done[(2 * u) % 16]; v1 = done[(2 * u + 1) % 16]; }
            else { v0 = x[u % 8]; v1 = x[(u + 1) % 8]; }
            asm("v_max_f32 %0, 0, %0" : "+v"(v0));
            asm("v_max_f32 %0, 0, %0" : "+v"(v1));
            const unsigned h = cvt_pk_f16(v0, v1);
            hacc ^= h;
            const i16x2 z = {0, 0};
            const i16x2 lp = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(z, rem16<0>(v0, h), rem16<1>(v1, h), 1.f / 4096.f, false);
            lacc ^= __builtin_bit_cast(int, lp) ^ __builtin_amdgcn_cvt_pk_bf8_f32(v0, v1, 0, false);
        } else if (KIND == 4) {
            if (POS && part == 1) return;
            const float4 q0 = *reinterpret_cast<const float4 *>(&lds[(u * 64 + lane * 4) & 8188]);
            const float4 q1 = *reinterpret_cast<const float4 *>(&lds[(u * 64 + lane * 4 + 2048) & 8188]);
            x[0] = fmaf(q0.x, x[7], x[0]); x[1] = fmaf(q0.y, x[7], x[1]); x[2] = fmaf(q0.z, x[7], x[2]); x[3] = fmaf(q0.w, x[7], x[3]);
            x[4] = fmaf(q1.x, x[6], x[4]); x[5] = fmaf(q1.y, x[6], x[5]); x[0] = fmaf(q1.z, x[6], x[0]); x[1] = fmaf(q1.w, x[6], x[1]);
        }
    };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (LDSREAD) {
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(r[(u + 2) % 3][0]), "=&v"(r[(u + 2) % 3][1]) : "v"(addr), "n"(0), "n"(1024) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r[u % 3][0]), "+v"(r[u % 3][1]));
            }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, r[u % 3][0]), b, c0, 0, 0, 0);
            if (POS) {
                __builtin_amdgcn_sched_barrier(0);
                filler(u, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, r[u % 3][1]), b, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            filler(u, POS ? 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float acc = c0[0] + c1[1] + (float)hacc + (float)lacc;
    for (int i = 0; i < 8; ++i) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
static float *out;
template <int KIND, int K, int POS, int LDSREAD> static void run(const char *tag) {
    const float t = time_ms([&] { hipLaunchKernelGGL((k<KIND, K, POS, LDSREAD>), dim3(512), dim3(256), 100 * 1024, 0, out, 2048); });
    printf("%-44s K %2d pos %d frag-reads %d: %.3f ms\n", tag, K, POS, LDSREAD, t);
}
int main() {
    CK(hipMalloc(&out, 1 << 22));
    printf("one wave per SIMD, 2 x 2048 x 24 fp16 MFMAs per SIMD\n");
    run<0, 0, 0, 0>("bare MFMAs");
    run<0, 0, 0, 1>("MFMAs + fragment reads");
    run<1, 8, 0, 1>("v_fma after the pair");
    run<1, 8, 1, 1>("v_fma split");
    run<1, 16, 0, 1>("v_fma after the pair");
    run<1, 16, 1, 1>("v_fma split");
    run<2, 0, 0, 1>("conversion slice (VGPR in), after the pair");
    run<2, 0, 1, 1>("conversion slice (VGPR in), between the MFMAs");
    run<3, 0, 0, 1>("conversion slice (AGPR in), after the pair");
    run<3, 0, 1, 1>("conversion slice (AGPR in), between the MFMAs");
    run<4, 0, 0, 1>("2 ds_read_b128 + 8 fma, after the pair");
    run<4, 0, 1, 1>("2 ds_read_b128 + 8 fma, between the MFMAs");
    run<2, 0, 0, 0>("conversion slice, no fragment reads");
    return 0;
}
