// fold_instrument.h — the instrumented builds of neuralbody_amd/csrc/nb_march_fold.hip (not part of the product):
//   NB_EXTRA_FLAGS="-DFOLD_TIMING -include tools/experiments/fold_instrument.h" NB_LIB_SUFFIX=_timing python -m neuralbody_amd.build
//   NB_EXTRA_FLAGS="-DFOLD_TAP -include tools/experiments/fold_instrument.h"    NB_LIB_SUFFIX=_tap    python -m neuralbody_amd.build
// FOLD_TIMING (fold_phase_times.py): wave 0 of the first 32 workgroups stamps the cycle counter at the phase boundaries of every
// depth step into the `raw` output as [workgroup][step][32].  FOLD_TAP (fold_check.py tap): workgroup 0 dumps, at depth step 0,
// every layer's accumulators as [layer][feature][sample] fp32 into `raw` (fc_0, fc_1, fc_2 pre-activation: 3 x 256 x 64; folded
// view layer: 128 x 64), to be compared with nb_decode_points' fp32 activation tap.
#pragma once
#define NB_MARCH_HOOKS_DEFINED
#ifdef FOLD_TIMING
#define FOLD_STAMP(i)                                                   \
    do {                                                                \
        if (tbuf) {                                                     \
            const unsigned long long t__ = __builtin_readcyclecounter(); \
            if (lane == 0) tbuf[(i)] = (unsigned)t__;                   \
        }                                                               \
    } while (0)
#define FOLD_SUB(i) FOLD_STAMP(i)
#define NB_HOOK_STEP_BEGIN \
    unsigned *tbuf = (blockIdx.x < 32 && wave == 0 && a.raw) ? reinterpret_cast<unsigned *>(a.raw) + ((size_t)blockIdx.x * S + s) * 32 : nullptr
#define NB_HOOK_TBUF tbuf
#define NB_HOOK_FOLD_FIRST_READ(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((f).ah[0]), "+v"((f).bl[1])::"memory")
#else
#define FOLD_STAMP(i) do { } while (0)
#define FOLD_SUB(i) do { } while (0)
#define NB_HOOK_STEP_BEGIN do { } while (0)
#define NB_HOOK_TBUF nullptr
#define NB_HOOK_FOLD_FIRST_READ(f) do { } while (0)
#endif
#ifdef FOLD_TAP
#define FOLD_DUMP(LAYER, MT_)                                                                                             \
    if (blockIdx.x == 0 && s == 0 && a.raw) {                                                                            \
        for (int m = 0; m < (MT_); ++m)                                                                                   \
            for (int n = 0; n < 2; ++n)                                                                                   \
                for (int r = 0; r < 16; ++r)                                                                              \
                    a.raw[((LAYER) * 256 + 32 * ((MT_) * wave + m) + tile_row(r, hi)) * 64 + n * 32 + (lane & 31)] = acc[m][n][r]; \
    }
#else
#define FOLD_DUMP(LAYER, MT_)
#endif
#define NB_HOOK_RAW_IS_OUTPUT false
