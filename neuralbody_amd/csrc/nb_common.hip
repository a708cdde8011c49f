// Error reporting and device queries of libnb_hip.so.
#include "nb_common.h"

#include <string.h>

static thread_local char g_err[512] = "";

void nb_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char *nb_last_error(void) { return g_err; }

int nb_abi_version(void) { return NB_ABI_VERSION; }

int nb_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        nb_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return NB_ENODEV;
    }
    return n;
}

}  // extern "C"
