// C-ABI housekeeping: version, last-error string, device probe.
#include "vft_common.h"
#include <string.h>

static thread_local char g_err[512] = {0};

void vft_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vft_last_error(void) { return g_err; }

extern "C" int vft_version(void) { return 100; }  // 0.1.0

// Fills name (<= cap bytes) with the gcnArchName of `device`; returns the CU count, or -1.
extern "C" int vft_device_info(int device, char* name, int cap) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        vft_set_error("vft_device_info: hipGetDeviceProperties(%d) failed", device);
        return -1;
    }
    if (name && cap > 0) {
        strncpy(name, prop.gcnArchName, (size_t)cap - 1);
        name[cap - 1] = 0;
    }
    return prop.multiProcessorCount;
}
