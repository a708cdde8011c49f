// nb_trread.h — K-major MFMA fragments from a row-major LDS tile by the hardware transpose read of gfx950.
// ds_read_b64_tr_b16: every lane passes the address of 4 contiguous 16-bit elements; each group of 16 lanes transposes the
// [4 rows][16 columns] block its lanes address (lane l15 addresses row l15 >> 2, columns 4 (l15 & 3) .. + 3) and lane l15
// receives column l15, rows 0..3 (tools/experiments/probe_trread.hip).  For a 32x32x16 MFMA operand (lane l: row / column
// l & 31 of the tile, K elements 8 (l >> 5) .. + 7) group g = l >> 4 takes columns 16 (g & 1) .. + 15 and rows 8 (g >> 1) .. + 7
// in two reads.  A row pitch of 2 C + 32 bytes (C a multiple of 32 channels) keeps the four rows of a block in distinct banks.
#pragma once
#include <hip/hip_runtime.h>

namespace nbtr {

typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// byte offset, inside a tile of row pitch `pitch`, of the segment lane `lane` addresses for K group rows 8 (lane >> 5) .. + 3
__device__ __forceinline__ unsigned lane_offset(int lane, int pitch) {
    const int l15 = lane & 15, g = lane >> 4;
    return (unsigned)((8 * (g >> 1) + (l15 >> 2)) * pitch + (16 * (g & 1) + 4 * (l15 & 3)) * 2);
}
// the fragment: rows +0..3 at `addr`, rows +4..7 at `addr + 4 pitch` (LDS byte addresses).  The compiler's own builtin: it
// counts the reads on lgkmcnt and places the wait in front of the first consumer (rounds 1-3 issued them from inline asm,
// where correctness hung on how the register allocator treated the asm outputs: ADVICE r03)
typedef __attribute__((address_space(3))) s4 *lds_s4_ptr;
__device__ __forceinline__ bf8 frag(unsigned addr_lo4, unsigned addr_hi4) {
    const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(unsigned long long)addr_lo4);
    const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(unsigned long long)addr_hi4);
    const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf8, v);
}
// fp32 -> bf16 head and remainder
__device__ __forceinline__ void split_bf16(float v, unsigned short &h, unsigned short &l) {
    const __bf16 hh = (__bf16)v;
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)hh));
}

}  // namespace nbtr
