// api.cpp — error reporting and device queries of libmpreid_hip.so
#include "common.h"

#include <cstring>

static thread_local char g_err[512] = "";

void mpreid_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mpreid_version(void) { return 100; } /* 0.1.0 */

extern "C" const char *mpreid_last_error(void) { return g_err; }

extern "C" int mpreid_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int mpreid_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return MPREID_OK;
}
