// Microbenchmark of the decode kernel's core: layers of 16 dependent f32 MFMAs whose B operands
// are relu() of the previous layer's accumulator, A operands from LDS.  TILES = independent
// point tiles interleaved in one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }
template <int TILES, int MODE>
__global__ void __launch_bounds__(1024) k(float *out, int iters) {
    __shared__ float w[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) w[i] = 1e-3f * (i & 1023) - 0.5f;
    __syncthreads();
    f32x16 x[TILES], y[TILES];
    for (int j = 0; j < TILES; ++j) for (int s = 0; s < 16; ++s) { x[j][s] = threadIdx.x * 0.01f + s + j; y[j][s] = 0.f; }
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const float *wl = w + (it & 15) * 1024;
        if (MODE == 0) {                       // layer-major: tile 0's whole layer, then tile 1's
#pragma unroll
            for (int j = 0; j < TILES; ++j)
#pragma unroll
                for (int s = 0; s < 16; ++s) y[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[s * 64 + lane], relu1(x[j][s]), y[j], 0, 0, 0);
        } else if (MODE == 2) {                // no relu at all: B operand straight from the accumulator
#pragma unroll
            for (int j = 0; j < TILES; ++j)
#pragma unroll
                for (int s = 0; s < 16; ++s) y[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[s * 64 + lane], x[j][s], y[j], 0, 0, 0);
        } else if (MODE == 3) {                // relu hoisted in front of the chain
#pragma unroll
            for (int j = 0; j < TILES; ++j) {
                f32x16 r;
#pragma unroll
                for (int s = 0; s < 16; ++s) r[s] = relu1(x[j][s]);
                asm volatile("" : "+v"(r));
#pragma unroll
                for (int s = 0; s < 16; ++s) y[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[s * 64 + lane], r[s], y[j], 0, 0, 0);
            }
        } else if (MODE == 4) {                // like 0 but without the y = t*1e-3 rescale (pure swap)
#pragma unroll
            for (int j = 0; j < TILES; ++j)
#pragma unroll
                for (int s = 0; s < 16; ++s) y[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[s * 64 + lane], relu1(x[j][s]), y[j], 0, 0, 0);
        } else {                               // step-major: the tiles' MFMAs alternate, sharing the A operand
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float a = wl[s * 64 + lane];
#pragma unroll
                for (int j = 0; j < TILES; ++j) y[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, relu1(x[j][s]), y[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < TILES; ++j) { f32x16 t = x[j]; x[j] = y[j]; y[j] = (MODE == 4) ? t : t * 1e-3f; }
    }
    float r = 0;
    for (int j = 0; j < TILES; ++j) for (int s = 0; s < 16; ++s) r += x[j][s];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int TILES, int MODE>
void run(int threads, const char *name) {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 1500;
    hipFuncSetAttribute((const void *)k<TILES, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<TILES, MODE><<<256, threads>>>(out, 10);
    hipEventRecord(e0);
    k<TILES, MODE><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 16 * TILES * (threads / 64) / 4.0;
    double tf = 256.0 * 4 * mfma_per_simd * 4096 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d tiles/wave %d: %.3f ms, %.1f TF\n", name, threads / 256, TILES, ms, tf);
    hipFree(out);
}
int main() {
    run<1, 0>(512, "relu interleaved"); run<1, 0>(1024, "relu interleaved");
    run<1, 2>(512, "no relu"); run<1, 2>(1024, "no relu");
    run<1, 3>(512, "relu hoisted"); run<1, 3>(1024, "relu hoisted");
    run<1, 4>(512, "no rescale"); run<1, 4>(1024, "no rescale");
    return 0;
}
