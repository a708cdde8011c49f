// probe_gather.hip — round 5: what a CU's vector memory path delivers for the access pattern of the sparse convolutions' row
// gathers.  hipcc -O3 --offload-arch=gfx950 tools/experiments/probe_gather.hip -o exp_bin/probe_gather, run on the MI355X.
// One workgroup of 8 waves per CU; every wave fetches the two fp16 planes of 32 rows per "offset" (ROWB bytes per row and plane,
// rows out of an L2-resident matrix), in one of three lane -> address maps:
//   A. the A-fragment map of v_mfma_f32_32x32x16_f16 as the kernels load it now: lane (i = lane % 32, hi = lane / 32), instruction
//      c reads bytes [32 c + 16 hi, + 16) of row i — 64 lanes, 32 rows, 32 bytes used of every 128-byte line per instruction;
//   B. four rows per instruction, lane p reads piece p / 4 of row 4 q + p % 4 (what an LDS transposer laid out [piece][row] would
//      issue);
//   C. ROWB / 16 lanes per row, contiguous (whole rows, fully coalesced);
//   D. four lanes per 64-byte piece, consecutive quads on DIFFERENT rows (quad Q: row Q % RPI, 64-byte piece Q / RPI);
//   E. map C by LDS-DMA (global_load_lds_dwordx4: 1 KiB per instruction lands lane-contiguous in LDS), then the A fragments
//      read back with ds_read_b128 — the whole staged path;
//   F. the march's weight stream: every instruction one contiguous 1-KiB piece (lane l its bytes [16 l, 16 l + 16)) out of a 704-KiB
//      L2-resident matrix, eight pieces in flight per wave — the ceiling of the vector-memory path for fully coalesced wave loads.
// The same bytes in every map.  Rows are consecutive with gaps (a slab of a level in voxel order) or random.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

// idx: [wave][offset][32] row numbers
template <int MAP, int ROWB>
__global__ __launch_bounds__(512) void gather_kernel(const char *__restrict__ rows, long long plane, const int *__restrict__ idx,
                                                     int offsets, int *__restrict__ out) {
    constexpr int NI = ROWB / 32;  // instructions per plane and offset: 32 rows x ROWB bytes / 1 KiB
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 8 + (threadIdx.x >> 6);
    const int *my = idx + (size_t)wave * offsets * 32;
    v4i acc = {0, 0, 0, 0};
    // the row numbers of offset o + 1 are fetched while offset o's rows are (one index -> row chain per wave would measure latency)
    auto row_of = [&](const int *ix, int q) -> int {
        if (MAP == 0) return ix[lane & 31];
        if (MAP == 1) {  // 1 KiB per instruction = RPI rows; lane p: row p % RPI of the instruction's rows, piece p / RPI
            constexpr int RPI = 1024 / ROWB;
            return ix[q * RPI + lane % RPI];
        }
        if (MAP == 3) {
            constexpr int RPI = 1024 / ROWB;
            return ix[q * RPI + (lane >> 2) % RPI];
        }
        constexpr int LPR = ROWB / 16, RPI = 64 / LPR;
        return ix[q * RPI + lane / LPR];
    };
    const int piece = MAP == 0 ? 16 * (lane >> 5)
                               : (MAP == 1 ? 16 * (lane / (1024 / ROWB))
                                           : (MAP == 3 ? 64 * ((lane >> 2) / (1024 / ROWB)) + 16 * (lane & 3) : 16 * (lane % (ROWB / 16))));
    int id[NI], nx[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) id[q] = row_of(my, q);
    for (int o = 0; o < offsets; ++o) {
        const int *ixn = my + (o + 1 < offsets ? o + 1 : o) * 32;
#pragma unroll
        for (int q = 0; q < NI; ++q) nx[q] = row_of(ixn, q);
        v4i v[2][NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const size_t off = (size_t)id[q] * ROWB + piece + (MAP == 0 ? 32 * q : 0);
            v[0][q] = *reinterpret_cast<const v4i *>(rows + off);
            v[1][q] = *reinterpret_cast<const v4i *>(rows + plane + off);
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) acc += v[0][q] ^ v[1][q];
#pragma unroll
        for (int q = 0; q < NI; ++q) id[q] = nx[q];
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) out[threadIdx.x] = 1;
}

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;
template <int ROWB>
__global__ __launch_bounds__(512) void gather_lds_kernel(const char *__restrict__ rows, long long plane, const int *__restrict__ idx,
                                                         int offsets, int *__restrict__ out) {
    constexpr int NI = ROWB / 32, LPR = ROWB / 16, RPI = 64 / LPR;
    constexpr int WAVE_BYTES = 2 * NI * 1040;  // 1 KiB per instruction + 16 bytes of padding
    __shared__ __attribute__((aligned(16))) char stage[8 * WAVE_BYTES];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wave = blockIdx.x * 8 + wv;
    const int *my = idx + (size_t)wave * offsets * 32;
    char *mine = stage + wv * WAVE_BYTES;
    v4i acc = {0, 0, 0, 0};
    int id[NI], nx[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) id[q] = my[q * RPI + lane / LPR];
    for (int o = 0; o < offsets; ++o) {
        const int *ixn = my + (o + 1 < offsets ? o + 1 : o) * 32;
#pragma unroll
        for (int q = 0; q < NI; ++q) nx[q] = ixn[q * RPI + lane / LPR];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const size_t off = (size_t)id[q] * ROWB + 16 * (lane % LPR);
            __builtin_amdgcn_global_load_lds((gptr_t)(rows + off), (lptr_t)(mine + q * 1040), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(rows + plane + off), (lptr_t)(mine + (NI + q) * 1040), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the A fragments: lane (i, hi), chunk c: bytes [32 c + 16 hi, + 16) of row i; row i sits in instruction i / RPI at (i % RPI) * ROWB
        const int i = lane & 31, hi = lane >> 5;
        const char *r0 = mine + (i / RPI) * 1040 + (i % RPI) * ROWB + 16 * hi;
#pragma unroll
        for (int c = 0; c < NI; ++c) {
            acc += *reinterpret_cast<const v4i *>(r0 + 32 * c) ^ *reinterpret_cast<const v4i *>(r0 + NI * 1040 + 32 * c);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NI; ++q) id[q] = nx[q];
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) out[threadIdx.x] = 1;
}

template <int ROWB>
static void run_lds(const char *rows, long long plane, const int *idx, int offsets, int *out, const char *what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather_lds_kernel<ROWB>), dim3(256), dim3(512), 0, 0, rows, plane, idx, offsets, out);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gather_lds_kernel<ROWB>), dim3(256), dim3(512), 0, 0, rows, plane, idx, offsets, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes_cu = 8.0 * offsets * 32 * 2 * ROWB;
    printf("  %-10s map E, %3d-byte rows: %7.1f us, %6.1f bytes per ns and CU (%5.1f per clock at 2.1 GHz), %.2f TB/s over 256 CUs (DMA, wait, fragment reads: serial per wave)\n",
           what, ROWB, ms * 1e3, bytes_cu / (ms * 1e6), bytes_cu / (ms * 1e6) / 2.1, 256 * bytes_cu / (ms * 1e9));
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(const char *__restrict__ w, int pieces, int iters, int *__restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v4i acc = {0, 0, 0, 0};
    int p = (blockIdx.x * 37 + wv * 11) % pieces;
    for (int it = 0; it < iters; ++it) {
        v4i v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r] = *reinterpret_cast<const v4i *>(w + (size_t)p * 1024 + lane * 16);
            p = p + 1 == pieces ? 0 : p + 1;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += v[r];
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) out[threadIdx.x] = 1;
}

template <int WAVES>
static void run_stream(const char *w, int *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int pieces = 704, iters = 400;
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((stream_kernel<WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, w, pieces, iters, out);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((stream_kernel<WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, w, pieces, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes_cu = (double)WAVES * iters * 8 * 1024;
    printf("  map F, %d waves per CU: %7.1f us, %6.1f bytes per ns and CU (%5.1f per clock at 2.1 GHz), %.2f TB/s over 256 CUs\n", WAVES, ms * 1e3,
           bytes_cu / (ms * 1e6), bytes_cu / (ms * 1e6) / 2.1, 256 * bytes_cu / (ms * 1e9));
}

template <int MAP, int ROWB>
static void run(const char *rows, long long plane, const int *idx, int offsets, int *out, const char *what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather_kernel<MAP, ROWB>), dim3(256), dim3(512), 0, 0, rows, plane, idx, offsets, out);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gather_kernel<MAP, ROWB>), dim3(256), dim3(512), 0, 0, rows, plane, idx, offsets, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes_cu = 8.0 * offsets * 32 * 2 * ROWB;  // per CU
    printf("  %-10s map %c, %3d-byte rows: %7.1f us, %6.1f bytes per ns and CU (%5.1f per clock at 2.1 GHz), %.2f TB/s over 256 CUs\n", what,
           "ABCD"[MAP], ROWB, ms * 1e3, bytes_cu / (ms * 1e6), bytes_cu / (ms * 1e6) / 2.1, 256 * bytes_cu / (ms * 1e9));
}

int main() {
    const int n_rows = 12000, offsets = 200, waves = 256 * 8;
    const long long plane = (long long)n_rows * 256;
    char *rows;
    CK(hipMalloc(&rows, 2 * plane));
    CK(hipMemset(rows, 1, 2 * plane));
    int *out, *idx;
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&idx, (size_t)waves * offsets * 32 * 4));
    std::vector<int> h((size_t)waves * offsets * 32);
    for (int kind = 0; kind < 2; ++kind) {
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        for (int w = 0; w < waves; ++w)
            for (int o = 0; o < offsets; ++o) {
                int r = rnd() % (n_rows - 200);
                for (int i = 0; i < 32; ++i) {
                    if (kind == 0) r += 1 + (rnd() % 8 == 0 ? rnd() % 5 : 0);  // consecutive rows, a gap now and then
                    else r = rnd() % n_rows;
                    h[((size_t)w * offsets + o) * 32 + i] = r;
                }
            }
        CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const char *what = kind == 0 ? "slab rows" : "random";
        printf("%s (8 waves per CU, %d offsets of 32 rows x 2 planes per wave):\n", what, offsets);
        run<0, 256>(rows, plane, idx, offsets, out, what);
        run<1, 256>(rows, plane, idx, offsets, out, what);
        run<2, 256>(rows, plane, idx, offsets, out, what);
        run<3, 256>(rows, plane, idx, offsets, out, what);
        run_lds<256>(rows, plane, idx, offsets, out, what);
        run<0, 128>(rows, plane / 2, idx, offsets, out, what);
        run<1, 128>(rows, plane / 2, idx, offsets, out, what);
        run<2, 128>(rows, plane / 2, idx, offsets, out, what);
        run<3, 128>(rows, plane / 2, idx, offsets, out, what);
        run_lds<128>(rows, plane / 2, idx, offsets, out, what);
    }
    printf("contiguous 1-KiB pieces of a 704-KiB matrix (the march's weight stream: 1 408 KiB per CU and depth step in ~40.6 k cycles = 35 bytes per clock):\n");
    run_stream<4>(rows, out);
    run_stream<8>(rows, out);
    run_stream<16>(rows, out);
    return 0;
}
