// Microbenchmark: issue cost of individual VALU instructions in the shadow of v_mfma_f32_32x32x16_f16 on gfx950.
// One wave per SIMD runs 8 x (MFMA + V copies of one instruction) per loop step, two accumulator chains; the in-kernel
// shader clock (s_memtime) gives cycles per MFMA gap.  32 = the MFMA hides everything.
//   hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define KINDS 12
static const char *names[KINDS] = {"v_fma_f32", "v_max_f32", "v_cvt_pkrtz_f16_f32", "v_pk_max_f16", "v_fma_mixlo_f16", "v_fma_mixhi_f16",
                                   "v_fma_mix_f32", "v_cvt_pk_bf16_f32", "v_and_b32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_f16_f32"};

template <int V, int KIND>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
    f32x16 accA, accB;
    for (int s = 0; s < 16; ++s) { accA[s] = threadIdx.x * 0.01f + s; accB[s] = threadIdx.x * 0.02f - s; }
    f16x8 wa, xb;
    for (int j = 0; j < 8; ++j) { wa[j] = (_Float16)(0.001f * (threadIdx.x + j)); xb[j] = (_Float16)(0.5f + 0.01f * j); }
    float v[8];
    unsigned u[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int j = 0; j < 8; ++j) { v[j] = 1.0f + 0.001f * (threadIdx.x + j); u[j] = threadIdx.x * 77 + j; p[j] = f2{v[j], v[j] + 1.0f}; }
    const float c1 = 0.999f, c2 = 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m & 1) accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, accB, 0, 0, 0);
            else accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xb, accA, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                float &r = v[q & 7];
                unsigned &w = u[q & 7];
                f2 &pp = p[q & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
                else if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(c2));
                else if (KIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w) : "v"(r), "v"(c1));
                else if (KIND == 3) asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(w));
                else if (KIND == 4) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "+v"(w) : "v"(u[(q + 1) & 7]), "v"(r));
                else if (KIND == 5) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(w) : "v"(u[(q + 1) & 7]), "v"(r));
                else if (KIND == 6) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(c1));
                else if (KIND == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(r), "v"(c1));
                else if (KIND == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w) : "v"(u[(q + 1) & 7]));
                else if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pp) : "v"(p[(q + 1) & 7]));
                else if (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pp) : "v"(p[(q + 1) & 7]));
                else if (KIND == 11) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(r), "v"(c1));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int s = 0; s < 16; ++s) r += accA[s] + accB[s];
    for (int j = 0; j < 8; ++j) r += v[j] + (float)u[j] + p[j][0] + p[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int KIND>
double run() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    k<V, KIND><<<256, 256>>>(out, cyc, 10);
    k<V, KIND><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    hipFree(out); hipFree(cyc);
    return s / 256 / iters / 8;
}

template <int KIND>
void row() {
    printf("%-22s cycles per MFMA gap: V=0 %.1f  V=2 %.1f  V=4 %.1f  V=6 %.1f  V=8 %.1f\n", names[KIND],
           run<0, KIND>(), run<2, KIND>(), run<4, KIND>(), run<6, KIND>(), run<8, KIND>());
}

int main() {
    row<0>(); row<1>(); row<2>(); row<3>(); row<4>(); row<5>(); row<6>(); row<7>(); row<8>(); row<9>(); row<10>(); row<11>();
    return 0;
}
