// What does cold code cost?  Every wave of a 256-workgroup launch runs (a) 8192 independent-ish v_add_f32 as straight-line code
// (32 KB, each cache line executed once) and (b) the same 8192 adds as 64 trips over a 128-instruction loop; shader-clock counts per
// wave, median over waves, first launch and a repeat.  hipcc --offload-arch=gfx950 -O3 tools/probe/icache_probe.hip -o icache_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define ADD8 "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\t" \
             "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\t"
#define ADD64 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8
#define ADD128 ADD64 ADD64

template <int LINEAR_KB>
__global__ void __launch_bounds__(512) probe(float *out, unsigned long long *t) {
    float x = threadIdx.x, y = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (LINEAR_KB) {
        // LINEAR_KB * 1024 / 4 adds, straight line
#pragma unroll
        for (int i = 0; i < LINEAR_KB * 2; ++i) asm volatile(ADD128 : "+v"(x) : "v"(y));
    } else {
#pragma unroll 1
        for (int i = 0; i < 64; ++i) asm volatile(ADD128 : "+v"(x) : "v"(y));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, float *out, unsigned long long *t, int adds) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, t);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2048);
        hipMemcpy(h.data(), t, 2048 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-28s launch %d: median %7llu counts per wave (%.2f per add), min %llu, max %llu\n", name, rep, h[1024], (double)h[1024] / adds, h[0], h[2047]);
    }
}

int main() {
    float *out; unsigned long long *t;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 2048 * 8);
    run("loop 64 x 128 adds", probe<0>, out, t, 8192);
    run("straight line 4 KB", probe<4>, out, t, 1024);
    run("straight line 8 KB", probe<8>, out, t, 2048);
    run("straight line 32 KB", probe<32>, out, t, 8192);
    run("loop 64 x 128 adds", probe<0>, out, t, 8192);
    return 0;
}
