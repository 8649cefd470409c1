// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 for dependent chains, by waves/SIMD and ILP.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ILP, bool LDSA>
__global__ void __launch_bounds__(1024) k(float *out, int iters) {
    __shared__ float w[16 * 64];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) w[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[ILP];
    for (int j = 0; j < ILP; ++j) for (int s = 0; s < 16; ++s) acc[j][s] = threadIdx.x * 0.01f + s;
    float a = 0.5f + threadIdx.x * 1e-3f, b = 0.25f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float av = LDSA ? w[s * 64 + lane] : a;
#pragma unroll
            for (int j = 0; j < ILP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[j], 0, 0, 0);
        }
    }
    float r = 0;
    for (int j = 0; j < ILP; ++j) for (int s = 0; s < 16; ++s) r += acc[j][s];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int ILP, bool LDSA>
void run(int threads, const char *name) {
    float *out; hipMalloc(&out, 256 * 1024 * 4 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<ILP, LDSA><<<256, threads>>>(out, 10);
    hipEventRecord(e0);
    k<ILP, LDSA><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 16 * ILP * (threads / 64) / 4.0;
    double tf = 256.0 * 4 * mfma_per_simd * 4096 / (ms * 1e-3) / 1e12;
    printf("%-28s threads %4d (waves/SIMD %d) ILP %d: %.3f ms, %.1f TF, %.1f ns per MFMA per SIMD\n", name, threads, threads / 256, ILP, ms, tf, ms * 1e6 / mfma_per_simd);
    hipFree(out);
}
int main() {
    run<1, false>(256, "dep chain, reg A");
    run<1, false>(512, "dep chain, reg A");
    run<1, false>(1024, "dep chain, reg A");
    run<2, false>(256, "2 chains, reg A");
    run<2, false>(512, "2 chains, reg A");
    run<4, false>(256, "4 chains, reg A");
    run<1, true>(256, "dep chain, LDS A");
    run<1, true>(512, "dep chain, LDS A");
    run<1, true>(1024, "dep chain, LDS A");
    run<2, true>(512, "2 chains, LDS A");
    return 0;
}
