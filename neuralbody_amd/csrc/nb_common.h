// Shared host-side helpers of libnb_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "nb_hip.h"

void nb_set_error(const char *fmt, ...);

#define NB_REQUIRE(cond, ...)      \
    do {                           \
        if (!(cond)) {             \
            nb_set_error(__VA_ARGS__); \
            return NB_EINVAL;      \
        }                          \
    } while (0)

#define NB_CHECK_LAUNCH(what)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            nb_set_error("%s: %s", (what), hipGetErrorString(e__));                  \
            return NB_ELAUNCH;                                                       \
        }                                                                            \
    } while (0)

#define NB_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            nb_set_error("%s: %s", #call, hipGetErrorString(e__));                   \
            return NB_ELAUNCH;                                                       \
        }                                                                            \
    } while (0)

static inline int nb_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
