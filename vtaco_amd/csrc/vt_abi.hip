// Error plumbing and device queries shared by every entry point of libvtaco_hip.so.
#include <stdio.h>
#include <string.h>

#include "vt_common.h"

static thread_local char g_err[512] = "";

int vt_fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int vt_check(hipError_t e, const char *where) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof g_err, "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}

int vt_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

extern "C" {
int vt_abi_version(void) { return VT_ABI_VERSION; }
const char *vt_last_error(void) { return g_err; }
}
