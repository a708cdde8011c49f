// probe_filler.hip — how many non-matrix instructions hide behind each MFMA when they come from the SAME wave
// (fenced in place with sched_barrier), with one or two such waves per SIMD; and cross-wave variants with s_nop padding + priority.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

// KIND 0: v_fma fillers; 1: ds_read_b128 fillers (+ use); 2: cvt_pk_bf16 + shifts + subs (operand split)
// MF 0: bf16 32x32x16; 1: fp8 scaled 32x32x64 (64-cycle)
template <int K, int KIND, int MF>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    v16f c[4] = {};
    b8 a, b;
    v8i a8, b8v;
    float x[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(float)(lane + i);
        b[i] = (__bf16)(float)(lane - i);
        a8[i] = lane * 0x01010101 + i;
        b8v[i] = lane * 0x01010101 - i;
        x[i] = (float)(lane + i);
    }
    const float m = 1.0001f, ad = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MF == 0) c[u % 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u % 4], 0, 0, 0);
            else c[u % 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8v, c[u % 4], 0, 0, 0, 127, 0, 127);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < K; ++f) {
                const int i = (u * K + f) % 8;
                if (KIND == 0) x[i] = fmaf(x[i], m, ad);
                else if (KIND == 1) {
                    const float4 q = *reinterpret_cast<const float4 *>(&lds[((it + u * K + f) * 64 + lane * 4) & 8188]);
                    x[i] += q.x;
                    ++f;
                } else {
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    const b2 hp = {(__bf16)x[i], (__bf16)x[(i + 1) % 8]};
                    x[(i + 2) % 8] = __uint_as_float(__builtin_bit_cast(unsigned, hp) << 16);
                    ++f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float acc = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    for (int i = 0; i < 8; ++i) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
static float time_ms(F f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
static float *out;
template <int K, int KIND, int MF>
static void run(const char *tag) {
    const int it = 2048;
    // one wave per SIMD: 256-thread groups, 100 KiB of LDS each so that only one fits a CU; 512 groups = 2 rounds
    const float t1 = time_ms([&] { hipLaunchKernelGGL((k<K, KIND, MF>), dim3(512), dim3(256), 100 * 1024, 0, out, it); });
    // two waves per SIMD: 512-thread groups, 256 groups = 1 round, same work per SIMD
    const float t2 = time_ms([&] { hipLaunchKernelGGL((k<K, KIND, MF>), dim3(256), dim3(512), 100 * 1024, 0, out, it); });
    printf("%-28s %d fillers per MFMA: 1 wave/SIMD %.3f ms | 2 waves/SIMD %.3f ms\n", tag, K, t1, t2);
}

int main() {
    CK(hipMalloc(&out, 1 << 22));
    printf("each SIMD executes 2 x 2048 x 16 MFMAs in every arm; bf16 32x32x16\n");
    run<0, 0, 0>("v_fma");
    run<2, 0, 0>("v_fma");
    run<4, 0, 0>("v_fma");
    run<5, 0, 0>("v_fma");
    run<6, 0, 0>("v_fma");
    run<8, 0, 0>("v_fma");
    run<12, 0, 0>("v_fma");
    run<2, 1, 0>("ds_read_b128+add");
    run<4, 1, 0>("ds_read_b128+add");
    run<6, 1, 0>("ds_read_b128+add");
    run<4, 2, 0>("cvt_pk_bf16+shift");
    run<6, 2, 0>("cvt_pk_bf16+shift");
    printf("fp8 scaled 32x32x64 (K=64)\n");
    run<0, 0, 1>("v_fma");
    run<4, 0, 1>("v_fma");
    run<8, 0, 1>("v_fma");
    run<12, 0, 1>("v_fma");
    run<16, 0, 1>("v_fma");
    return 0;
}
