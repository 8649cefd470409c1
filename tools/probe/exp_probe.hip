// Microbenchmark: what does an exponential cost on a gfx950 SIMD, and does v_exp_f32 (transcendental, quarter rate) overlap with
// ordinary / packed f32 VALU instructions of the same or of another wave?  The TransformerFusion passes issue 16 v_exp_f32 + 16
// FMAs per 32 x 32 score tile; the verdict of round 4 asks whether HALF the tile on a packed-FMA polynomial would overlap with the
// other half on v_exp_f32.
//   KIND 0: 16 x v_exp_f32                      KIND 1: 16 x v_fma_f32
//   KIND 2: 16 x v_exp_f32 + 16 x v_fma_f32     KIND 3: degree-7 polynomial for 16 values on v_pk_fma_f32 (8 pairs x 7)
//   KIND 4: 8 x v_exp_f32 + the polynomial for the other 8 values (4 pairs x 7), interleaved
//   KIND 5: 16 x v_exp_f32 + 16 x v_pk_fma_f32
// Cycles per step (s_memtime) for one wave per SIMD and for four (the second column is what a SIMD spends per wave-step).
//   hipcc --offload-arch=gfx950 -O3 -o exp_probe exp_probe.hip && ./exp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters) {
    float v[16];
    f2 p[8], q[8];
    for (int j = 0; j < 16; ++j) v[j] = 0.001f * (threadIdx.x + j);
    for (int j = 0; j < 8; ++j) { p[j] = f2{0.01f * j, 0.02f * j}; q[j] = f2{0.3f + 0.001f * threadIdx.x, 0.1f * j}; }
    const float c1 = 0.999f, c2 = 1e-3f;
    const f2 pc = f2{0.11f, 0.11f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0 || KIND == 2 || KIND == 5) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
                if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7][j >> 3]) : "v"(c1), "v"(c2));
                if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7]) : "v"(q[j & 7]), "v"(pc));
            }
        } else if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c1), "v"(c2));
        } else if (KIND == 3) {
#pragma unroll
            for (int d = 0; d < 7; ++d)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(q[j]), "v"(pc));
        } else if (KIND == 4) {
#pragma unroll
            for (int d = 0; d < 7; ++d) {
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(q[j]), "v"(pc));
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[d]));
            }
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int j = 0; j < 16; ++j) s += v[j];
    for (int j = 0; j < 8; ++j) s += p[j][0] + p[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
static void run(const char *name, float *out, unsigned long long *cyc) {
    const int iters = 4096;
    for (int threads : {256, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        // one step of EVERY resident wave of the SIMD takes c / iters shader cycles: / waves = SIMD cycles per wave-step
        printf("%-58s %d wave(s)/SIMD: %7.1f cycles per step, %6.1f SIMD cycles per wave-step\n", name, threads / 256, (double)c / iters,
               (double)c / iters / (threads / 256));
    }
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, sizeof(*cyc));
    run<0>("16 v_exp_f32", out, cyc);
    run<1>("16 v_fma_f32", out, cyc);
    run<2>("16 v_exp_f32 + 16 v_fma_f32", out, cyc);
    run<3>("poly7 for 16 values (56 v_pk_fma_f32)", out, cyc);
    run<4>("8 v_exp_f32 + poly7 for 8 values (28 v_pk_fma_f32)", out, cyc);
    run<5>("16 v_exp_f32 + 16 v_pk_fma_f32", out, cyc);
    return 0;
}
