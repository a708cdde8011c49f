// probe_icache.hip — how large may a straight-line loop body be before instruction fetch limits the march's record loop?
// Body = N records of [2 ds_read_b128, counted wait, 2 fp16 MFMAs or 1 8-bit MFMA], forced straight-line with a template loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <class F, int... I> __device__ __forceinline__ void sfi(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { sfi(f, std::make_integer_sequence<int, N>{}); }

template <int NREC, int BAR = 0, int DMA = 0>
__global__ __launch_bounds__(256) void k(float *out, int iters, const char *wsrc = nullptr) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.0f;
    __syncthreads();
    v16f c0 = {}, c1 = {};
    h8 b16;
    v8i b8;
    for (int i = 0; i < 8; ++i) { b16[i] = (_Float16)(float)(lane - i); b8[i] = lane * 0x01010101 - i; }
    i32x4 r[3][2];
    for (int q = 0; q < 3; ++q) r[q][0] = r[q][1] = i32x4{lane, lane, lane, lane};
    const int addr = lane * 16;
    for (int it = 0; it < iters; ++it) {
        static_for<NREC>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (BAR && u % 12 == 0) {
                if (DMA) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
            }
            if constexpr (DMA && (u % 2) == 1) {
                typedef const void __attribute__((address_space(1))) *gptr_t;
                typedef void __attribute__((address_space(3))) *lptr_t;
                const char *src = wsrc + (size_t)((it * NREC + u) % 512) * 2048 + (threadIdx.x >> 6) * 6144 + lane * 16;
                char *dst = reinterpret_cast<char *>(lds) + 32768 + ((u / 24) % 2) * 24576 + (threadIdx.x >> 6) * 6144;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, ((u % 12) / 2 % 4) * 1024, 0);
            }
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(r[(u + 2) % 3][0]), "=&v"(r[(u + 2) % 3][1]) : "v"(addr), "n"((u % 12) * 2048), "n"((u % 12) * 2048 + 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r[u % 3][0]), "+v"(r[u % 3][1]));
            if constexpr ((u % 8) < 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, r[u % 3][0]), b16, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, r[u % 3][1]), b16, c1, 0, 0, 0);
            } else {
                const i32x4 p0 = r[u % 3][0], p1 = r[u % 3][1];
                const v8i a = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                if constexpr (u & 1) c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b8, c1, 0, 1, 0, 127, 0, 115);
                else c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b8, c0, 0, 1, 0, 127, 0, 115);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
static float *out, *wsrc;
template <int NREC, int BAR = 0, int DMA = 0> static void run() {
    const int total = 49152, it = total / NREC;
    const float t = time_ms([&] { hipLaunchKernelGGL((k<NREC, BAR, DMA>), dim3(256), dim3(256), 100 * 1024, 0, out, it, (const char *)wsrc); });
    
    
    printf("body %4d records, barrier %d, DMA %d: %.3f ms = %.1f ns per record\n", NREC, BAR, DMA, t, t * 1e6 / (it * NREC));
}
int main() {
    CK(hipMalloc(&out, 1 << 22));
    CK(hipMalloc(&wsrc, 4 << 20));
    CK(hipMemset(wsrc, 0, 4 << 20));
    run<96, 1, 0>(); run<96, 0, 1>(); run<96, 1, 1>(); run<384, 1, 1>();
    run<96>(); run<1536>();
    return 0;
}
