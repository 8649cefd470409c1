// Microbenchmark: does VALU work issue under bf16 MFMAs on gfx950?  One kernel runs, per loop step, two
// independent v_mfma_f32_32x32x16_bf16 chains (like the two-brick decode) with V independent VALU ops
// (v_fma / v_cvt_pk_bf16_f32 / v_max mix) placed after every MFMA.  If the pipes overlap, time stays flat
// until V reaches what fits under one MFMA; if they do not, time grows from V = 1.
//   hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip && ./overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// same MFMA work per group in the 16x16x32 shape: two 4-pass MFMAs instead of one 8-pass MFMA, V/2 VALU ops after each
template <int V>
__global__ void __launch_bounds__(512) k16(float *out, int iters) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) for (int s = 0; s < 4; ++s) acc[j][s] = threadIdx.x * 0.01f + s + j;
    bf16x8 wa, xb;
    for (int j = 0; j < 8; ++j) { wa[j] = (__bf16)(0.001f * (threadIdx.x + j)); xb[j] = (__bf16)(0.5f + 0.01f * j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 1.0f + 0.001f * (threadIdx.x + j);
    const float c1 = 0.999f, c2 = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V / 2; ++q) {
                float &r = v[(q + 4 * (m & 1)) & 7];
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
            }
        }
    }
    float r = 0;
    for (int j = 0; j < 4; ++j) for (int s = 0; s < 4; ++s) r += acc[j][s];
    for (int j = 0; j < 8; ++j) r += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V>
void run16(int threads) {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k16<V><<<256, threads>>>(out, 10);
    hipEventRecord(e0);
    k16<V><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 8 * (threads / 256);                 // groups of two 16x16x32 MFMAs + V VALU
    printf("16x16x32   mfma+valu V=%2d waves/SIMD %d: %.3f ms, %.1f ns per group (2 MFMAs) per SIMD\n", V, threads / 256, ms, ms * 1e6 / steps);
    hipFree(out);
}

template <int V, int KIND, bool MFMA>
__global__ void __launch_bounds__(512) k(float *out, int iters) {
    f32x16 accA, accB;
    for (int s = 0; s < 16; ++s) { accA[s] = threadIdx.x * 0.01f + s; accB[s] = threadIdx.x * 0.02f - s; }
    bf16x8 wa, xb;
    for (int j = 0; j < 8; ++j) { wa[j] = (__bf16)(0.001f * (threadIdx.x + j)); xb[j] = (__bf16)(0.5f + 0.01f * j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 1.0f + 0.001f * (threadIdx.x + j);
    const float c1 = 0.999f, c2 = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MFMA) {
                if (m & 1) accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, accB, 0, 0, 0);
                else accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, accA, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < V; ++q) {
                float &r = v[q & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
                else if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(c2));
                else { unsigned t; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(r), "v"(c1)); asm volatile("" :: "v"(t)); }
            }
        }
    }
    float r = 0;
    for (int s = 0; s < 16; ++s) r += accA[s] + accB[s];
    for (int j = 0; j < 8; ++j) r += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int V, int KIND, bool MFMA>
void run(int threads, const char *name) {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V, KIND, MFMA><<<256, threads>>>(out, 10);
    hipEventRecord(e0);
    k<V, KIND, MFMA><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 8 * (threads / 256);                 // (MFMA + V VALU) groups per SIMD
    printf("%-10s %s V=%2d waves/SIMD %d: %.3f ms, %.1f ns per group per SIMD (= %.0f cycles at 2.4 GHz)\n", name,
           MFMA ? "mfma+valu" : "valu only", V, threads / 256, ms, ms * 1e6 / steps, ms * 1e6 / steps * 2.4);
    hipFree(out);
}

int main() {
    run<0, 0, true>(256, "fma"); run<0, 0, true>(512, "fma");
    run<2, 0, true>(256, "fma"); run<4, 0, true>(256, "fma"); run<6, 0, true>(256, "fma"); run<8, 0, true>(256, "fma");
    run<12, 0, true>(256, "fma"); run<16, 0, true>(256, "fma");
    run<4, 0, true>(512, "fma"); run<8, 0, true>(512, "fma"); run<16, 0, true>(512, "fma");
    run<8, 0, false>(256, "fma"); run<16, 0, false>(256, "fma"); run<8, 0, false>(512, "fma");
    run<8, 1, true>(256, "max"); run<8, 1, false>(256, "max");
    run<8, 2, true>(256, "cvt_pk"); run<8, 2, false>(256, "cvt_pk"); run<8, 2, true>(512, "cvt_pk");
    run16<0>(256); run16<0>(512); run16<4>(512); run16<8>(512); run16<12>(512); run16<16>(512); run16<8>(256); run16<16>(256);
    return 0;
}
