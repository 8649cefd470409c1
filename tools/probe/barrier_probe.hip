// Cost of a device-side barrier between the phases of a persistent kernel on MI355X: N barriers back to back, per variant.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_probe tools/probe/barrier_probe.hip && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned nwg, unsigned &gen) {
    if (MODE == 0) __threadfence();                       // every wave: release at agent scope (buffer_wbl2 sc1)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 1) __threadfence();
        const unsigned old = __hip_atomic_fetch_add(bar, 1u, MODE == 2 ? __ATOMIC_RELAXED : __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nwg - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(bar + 1, 1u, MODE == 2 ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(bar + 1, MODE == 2 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
        if (MODE == 1) __threadfence();
    }
    gen += 1;
    __syncthreads();
    if (MODE == 0) __threadfence();
}

template <int MODE>
__global__ void __launch_bounds__(1024) probe(unsigned *bar, int n, float *data, int touch) {
    unsigned gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        if (touch) {   // every workgroup writes 4 KB and reads another workgroup's 4 KB of the previous round
            data[(size_t)(i & 1) * gridDim.x * 1024 + blockIdx.x * 1024 + (threadIdx.x & 1023)] = acc + i;
        }
        grid_barrier<MODE>(bar, gridDim.x, gen);
        if (touch) acc += data[(size_t)(i & 1) * gridDim.x * 1024 + ((blockIdx.x + 37) % gridDim.x) * 1024 + (threadIdx.x & 1023)];
    }
    if (acc == 12345.f) data[0] = acc;
}

template <int MODE>
void run(const char *name, int wgs, int threads, int touch) {
    unsigned *bar; float *data;
    hipMalloc(&bar, 256); hipMemset(bar, 0, 256);
    hipMalloc(&data, (size_t)2 * wgs * 1024 * 4); hipMemset(data, 0, (size_t)2 * wgs * 1024 * 4);
    const int n = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(threads), 0, 0, bar, 100, data, touch);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(threads), 0, 0, bar, n, data, touch);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s %3d workgroups x %4d threads, touch %d: %.2f us per barrier (%s)\n", name, wgs, threads, touch, 1e3 * ms / n, hipGetErrorString(hipGetLastError()));
    hipFree(bar); hipFree(data);
}

int main() {
    for (int touch = 0; touch < 2; ++touch)
        for (int wgs : {64, 128, 256})
            for (int threads : {256, 1024}) {
                run<0>("fence by every wave", wgs, threads, touch);
                run<1>("fence by thread 0", wgs, threads, touch);
                run<2>("no fence (relaxed atomics)", wgs, threads, touch);
            }
    return 0;
}
