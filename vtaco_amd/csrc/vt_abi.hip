// Error plumbing and device queries shared by every entry point of libvtaco_hip.so.
#include <stdio.h>
#include <string.h>
#include <stdint.h>
#include <mutex>
#include <set>
#include <utility>

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

hipError_t vt_max_dyn_lds(const void *fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, fn})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, fn});
    return e;
}

__global__ void __launch_bounds__(256) vt_fill32_kernel(unsigned *dst, unsigned pattern, size_t n) {
    const size_t n4 = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? n / 4 : 0;
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    const uint4 p4 = make_uint4(pattern, pattern, pattern, pattern);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d4[i] = p4;
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = pattern;
}

int vt_fill32(void *dst, unsigned pattern, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;
    if (!n) return 0;
    size_t g = (n / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(vt_fill32_kernel, dim3((unsigned)g), dim3(256), 0, stream, (unsigned *)dst, pattern, n);
    return vt_check(hipGetLastError(), "vt_fill32");
}

extern "C" {
int vt_abi_version(void) { return VT_ABI_VERSION; }
const char *vt_last_error(void) { return g_err; }
}
