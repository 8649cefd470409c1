// Microbenchmark: what two waves on ONE SIMD can issue together on gfx950 (the question behind the decode kernel's
// structure): 512-thread workgroups, one per CU; waves w and w + 4 share a SIMD.  Each wave runs a loop of
// MA x { one v_mfma_f32_32x32x16_f16 (one dependent chain) + VA independent VALU instructions } (role A: waves 0-3) or the
// same with (MB, VB) (role B: waves 4-7); MA / MB = 0 gives a VALU-only stream, VA / VB = 0 an MFMA-only one, MB = VB = 0 an
// absent partner.  Prints shader cycles per loop step for both roles.
//   hipcc --offload-arch=gfx950 -O3 -o pair_probe pair_probe.hip && ./pair_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int M, int V, int KIND>
__device__ __forceinline__ void body(f32x16 &acc, const f16x8 &wa, const f16x8 &xb, float (&v)[8], unsigned (&u)[8], int iters) {
    const float c1 = 0.999f, c2 = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (M) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                float &r = v[q & 7];
                unsigned &w = u[q & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
                else if (KIND == 1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w) : "v"(r), "v"(c1));
                else if (KIND == 2) asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(w));
                else if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(w), "v"(c1));
                else if (KIND == 4) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "+v"(w) : "v"(u[(q + 1) & 7]), "v"(r));
                else if (KIND == 5) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, 4.0" : "+v"(w) : "v"(u[(q + 1) & 7]));
                else if (KIND == 6) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, 4.0" : "+v"(w) : "v"(r), "v"(c1));
                else if (KIND == 7) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(w) : "v"(r), "v"(c1));
            }
            if (!M && !V) asm volatile("s_nop 0");
        }
    }
}

template <int MA, int VA, int MB, int VB, int KIND>
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *cyc, int iters) {
    f32x16 acc;
    for (int s = 0; s < 16; ++s) acc[s] = threadIdx.x * 0.01f + s;
    f16x8 wa, xb;
    for (int j = 0; j < 8; ++j) { wa[j] = (_Float16)(0.001f * (threadIdx.x + j)); xb[j] = (_Float16)(0.5f + 0.01f * j); }
    float v[8];
    unsigned u[8];
    for (int j = 0; j < 8; ++j) { v[j] = 1.0f + 0.001f * (threadIdx.x + j); u[j] = threadIdx.x * 77 + j; }
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);     // 0: waves 0-3, 1: waves 4-7
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) body<MA, VA, KIND>(acc, wa, xb, v, u, iters);
    else if (MB || VB) body<MB, VB, KIND>(acc, wa, xb, v, u, iters);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int s = 0; s < 16; ++s) r += acc[s];
    for (int j = 0; j < 8; ++j) r += v[j] + (float)u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 255) == 0) cyc[blockIdx.x * 2 + role] = t1 - t0;
}

template <int MA, int VA, int MB, int VB, int KIND>
void run(const char *what) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 512 * 8);
    const int iters = 2000;
    k<MA, VA, MB, VB, KIND><<<256, 512>>>(out, cyc, 10);
    k<MA, VA, MB, VB, KIND><<<256, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[512];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) { a += (double)h[2 * i]; b += (double)h[2 * i + 1]; }
    printf("%-46s A(%d mfma + %d valu): %6.1f cycles/step   B(%d mfma + %d valu): %6.1f cycles/step\n", what, MA, VA, a / 256 / iters / 8,
           MB, VB, (MB || VB) ? b / 256 / iters / 8 : 0.0);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, 0, 0, 0, 0>("mfma chain alone");
    run<1, 0, 1, 0, 0>("two mfma chains");
    run<0, 4, 0, 0, 0>("4 v_fma alone");
    run<0, 4, 0, 4, 0>("4 v_fma + 4 v_fma");
    run<1, 0, 0, 4, 0>("mfma chain | 4 v_fma per step");
    run<1, 0, 0, 8, 0>("mfma chain | 8 v_fma per step");
    run<1, 0, 0, 16, 0>("mfma chain | 16 v_fma per step");
    run<1, 4, 0, 0, 0>("mfma + 4 v_fma alone");
    run<1, 6, 0, 0, 0>("mfma + 6 v_fma alone");
    run<1, 8, 0, 0, 0>("mfma + 8 v_fma alone");
    run<1, 4, 1, 4, 0>("mfma + 4 v_fma | same");
    run<1, 6, 1, 6, 0>("mfma + 6 v_fma | same");
    run<1, 8, 1, 8, 0>("mfma + 8 v_fma | same");
    run<1, 12, 1, 12, 0>("mfma + 12 v_fma | same");
    run<1, 4, 0, 0, 1>("mfma + 4 cvt_pkrtz alone");
    run<1, 4, 0, 0, 2>("mfma + 4 pk_max alone");
    run<1, 4, 0, 0, 3>("mfma + 4 fma_mix_f32 alone");
    run<1, 4, 0, 0, 4>("mfma + 4 fma_mixlo_f16 alone");
    run<1, 8, 1, 8, 1>("mfma + 8 cvt_pkrtz | same");
    run<1, 8, 1, 8, 3>("mfma + 8 fma_mix_f32 | same");
    run<1, 8, 1, 8, 4>("mfma + 8 fma_mixlo_f16 | same");
    run<1, 0, 0, 8, 4>("mfma chain | 8 fma_mixlo_f16");
    run<0, 8, 0, 0, 0>("8 v_fma alone");
    run<0, 8, 0, 0, 1>("8 cvt_pkrtz alone");
    run<0, 8, 0, 0, 5>("8 cvt_scalef32_pk_fp8_f16 alone");
    run<0, 8, 0, 0, 6>("8 cvt_scalef32_pk_fp8_f32 alone");
    run<0, 8, 0, 0, 7>("8 cvt_pk_fp8_f32 alone");
    run<0, 8, 0, 8, 5>("8 cvt_scalef32_pk_fp8_f16 | same");
    run<1, 4, 0, 0, 5>("mfma + 4 cvt_scalef32_pk_fp8_f16 alone");
    run<1, 4, 0, 0, 6>("mfma + 4 cvt_scalef32_pk_fp8_f32 alone");
    run<1, 4, 0, 0, 7>("mfma + 4 cvt_pk_fp8_f32 alone");
    run<1, 0, 0, 8, 5>("mfma chain | 8 cvt_scalef32_pk_fp8_f16");
    run<1, 0, 0, 8, 6>("mfma chain | 8 cvt_scalef32_pk_fp8_f32");
    run<1, 0, 0, 8, 7>("mfma chain | 8 cvt_pk_fp8_f32");
    run<1, 0, 0, 8, 1>("mfma chain | 8 cvt_pkrtz");
    run<1, 0, 0, 8, 3>("mfma chain | 8 fma_mix_f32");
    return 0;
}
