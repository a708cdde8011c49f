// probe_x2.hip — per-record cost in the march's record loop: [2 fragment ds_read_b128 two records ahead, counted wait, MFMA(s)].
// main record = 2 fp16 MFMAs; cross record = 1 K=64 8-bit MFMA.  Is the fragment-read issue hidden behind a single cross MFMA?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
// PATTERN 0: 12 main records; 1: 12 cross records (1 MFMA each); 2: 12 cross records with 2 MFMAs each (two tiles from one record... i.e.
// what a 2-tile 8-bit fragment would allow); 3: kernel-like 4 main + 4 cross
template <int PATTERN, int SCALED, int READS, int UNROLL = 1>
__global__ __launch_bounds__(256) void k(float *out, int iters, int sa, int sb) {
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
    auto X = [&](v16f &c, const i32x4 p0, const i32x4 p1) {
        const v8i a = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        if (SCALED) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b8, c, 0, 1, 0, sa, 0, sb);
        else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b8, c, 0, 1, 0, 0, 0, 0);
    };
    for (int it = 0; it < iters / UNROLL; ++it) {
#pragma unroll
        for (int u = 0; u < 12 * UNROLL; ++u) {
            if (READS) {
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(r[(u + 2) % 3][0]), "=&v"(r[(u + 2) % 3][1]) : "v"(addr), "n"(0), "n"(1024) : "memory");
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r[u % 3][0]), "+v"(r[u % 3][1]));
            }
            const bool cross = PATTERN == 1 || PATTERN == 2 || (PATTERN == 3 && (u % 8) >= 4);
            if (!cross) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, r[u % 3][0]), b16, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, r[u % 3][1]), b16, c1, 0, 0, 0);
            } else if (PATTERN == 2) {
                X(c0, r[u % 3][0], r[u % 3][1]);
                X(c1, r[u % 3][1], r[u % 3][0]);
            } else {
                X((u & 1) ? c1 : c0, r[u % 3][0], r[u % 3][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}
static float *out;
template <int PATTERN, int SCALED, int READS, int UNROLL = 1> static void run(const char *tag) {
    const int it = 4096;
    const float t = time_ms([&] { hipLaunchKernelGGL((k<PATTERN, SCALED, READS, UNROLL>), dim3(256), dim3(256), 64 * 1024, 0, out, it, 127, 115); });
    printf("%-52s scaled %d reads %d unroll %3d: %.3f ms = %.1f ns per record\n", tag, SCALED, READS, UNROLL, t, t * 1e6 / it / 12);
}
int main() {
    CK(hipMalloc(&out, 1 << 22));
    run<0, 0, 0>("main records (2 fp16 MFMAs)");
    run<0, 0, 1>("main records (2 fp16 MFMAs)");
    run<1, 1, 0>("cross records (1 scaled 8-bit MFMA)");
    run<1, 1, 1>("cross records (1 scaled 8-bit MFMA)");
    run<1, 0, 1>("cross records (1 unscaled 8-bit MFMA)");
    run<2, 1, 1>("cross records with 2 MFMAs per record");
    run<3, 1, 1>("4 main + 4 cross");
    run<3, 0, 1>("4 main + 4 cross");
    run<3, 1, 1, 8>("4 main + 4 cross, straight-line body x8");
    run<3, 1, 1, 32>("4 main + 4 cross, straight-line body x32");
    run<3, 1, 1, 64>("4 main + 4 cross, straight-line body x64");
    run<3, 1, 1, 128>("4 main + 4 cross, straight-line body x128");
    return 0;
}
