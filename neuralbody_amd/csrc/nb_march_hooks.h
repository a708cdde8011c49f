// nb_march_hooks.h — instrumentation points of nb_march_fold.hip.  The product build sees them empty.  An experiment build
// pre-includes tools/experiments/fold_instrument.h (hipcc -include ...), which defines them first: cycle stamps at the phase
// boundaries of a depth step (tools/experiments/fold_phase_times.py) or per-layer accumulator taps (fold_check.py tap).
#pragma once
#ifndef NB_MARCH_HOOKS_DEFINED
#define FOLD_STAMP(i) do { } while (0)            // phase boundary i of the depth step
#define FOLD_SUB(i) do { } while (0)              // ... inside the folded first layer
#define FOLD_DUMP(LAYER, MT_)                     // the layer's accumulators of workgroup 0, depth step 0
#define NB_HOOK_STEP_BEGIN do { } while (0)       // top of a depth step
#define NB_HOOK_TBUF nullptr                      // where the folded first layer stamps
#define NB_HOOK_FOLD_FIRST_READ(f) do { } while (0)
#define NB_HOOK_RAW_IS_OUTPUT true                // `raw` carries the decoder output (and not an instrument's records)
#endif
