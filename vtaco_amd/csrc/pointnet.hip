// PointNet per-point MLP for gfx950 (inference): the fully connected layers around the local pooling of
// LocalPoolPointnet.forward (reference src/encoder/pointnet.py:154-162) -- fc_pos, the five ResnetBlockFC
// (src/layers.py:8-50) on [net | pooled], fc_c -- as two kernels instead of ~9 host-framework launches per block.
//
// 3000 points x 10 KFLOP per block is 31 MFLOP: far below anything a matrix core or HBM would bound; the
// cost of this stage is the number of launches (each ~5 us on a 1.6 ms scene).  So: plain f32 FMAs, one thread
// per (point, output channel), weights transposed into LDS once per workgroup (consecutive channels on
// consecutive banks; rows padded to an odd length so the transposing writes spread over the banks too), the point's
// input row read as an LDS broadcast.  f32 throughout; every dot product runs as four interleaved partial sums
// (k mod 4), a dependent chain of 64 FMAs otherwise being the whole kernel time.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vt_common.h"
#include "vtaco_hip.h"

namespace {

constexpr int PN_THREADS = 256;

// wt[k * ld + j] = w[j * n + k] (nn.Linear [out][in] -> [in][out|1]); 16-byte loads when the row length allows
__device__ __forceinline__ void transpose_to_lds(float *wt, int ld, const float *w, int rows, int n) {
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < rows * n4; i += PN_THREADS) {
            const int j = i / n4, k = (i - j * n4) * 4;
            const float4 v = reinterpret_cast<const float4 *>(w)[i];
            wt[k * ld + j] = v.x; wt[(k + 1) * ld + j] = v.y; wt[(k + 2) * ld + j] = v.z; wt[(k + 3) * ld + j] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < rows * n; i += PN_THREADS) { const int j = i / n, k = i - j * n; wt[k * ld + j] = w[i]; }
    }
}

// sum_k wt[k * ld + j] * f(r[k]) as four partial sums (k mod 4), combined pairwise; RELU applies max(.,0) to r
template <bool RELU>
__device__ __forceinline__ float dot_cols(const float *wt, int ld, int j, const float *r, int n) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        const float r0 = RELU ? fmaxf(r[k], 0.0f) : r[k], r1 = RELU ? fmaxf(r[k + 1], 0.0f) : r[k + 1];
        const float r2 = RELU ? fmaxf(r[k + 2], 0.0f) : r[k + 2], r3 = RELU ? fmaxf(r[k + 3], 0.0f) : r[k + 3];
        a0 = fmaf(wt[k * ld + j], r0, a0);
        a1 = fmaf(wt[(k + 1) * ld + j], r1, a1);
        a2 = fmaf(wt[(k + 2) * ld + j], r2, a2);
        a3 = fmaf(wt[(k + 3) * ld + j], r3, a3);
    }
    for (; k < n; ++k) a0 = fmaf(wt[k * ld + j], RELU ? fmaxf(r[k], 0.0f) : r[k], a0);
    return (a0 + a1) + (a2 + a3);
}

// out[n][j] = b[j] + sum_k w[j][k] * x[n][k]
__global__ void __launch_bounds__(PN_THREADS)
linear_rows_kernel(const float *x, const float *w, const float *b, float *out, int N, int Cin, int Cout) {
    extern __shared__ float lds[];                       // wt [Cin][Cout|1] | rows [PTS][Cin]
    const int ldo = Cout | 1;
    float *wt = lds, *rows = lds + (size_t)Cin * ldo;
    const int pts = PN_THREADS / Cout;
    transpose_to_lds(wt, ldo, w, Cout, Cin);
    const int lp = threadIdx.x / Cout, j = threadIdx.x - lp * Cout;
    for (int n0 = blockIdx.x * pts; n0 < N; n0 += gridDim.x * pts) {
        __syncthreads();
        for (int i = threadIdx.x; i < pts * Cin; i += PN_THREADS) {
            const int p = i / Cin, n = n0 + p;
            rows[i] = n < N ? x[(size_t)n * Cin + (i - p * Cin)] : 0.0f;
        }
        __syncthreads();
        const int n = n0 + lp;
        if (lp < pts && n < N) {
            out[(size_t)n * Cout + j] = (b ? b[j] : 0.0f) + dot_cols<false>(wt, ldo, j, rows + lp * Cin, Cin);
        }
    }
}

// ResnetBlockFC on the (virtually concatenated) rows [x1 | x2]:
//   h = b0 + W0 relu(x);  out = b1 + W1 relu(h) + (Ws x  or  x when there is no shortcut layer)
__global__ void __launch_bounds__(PN_THREADS)
resblock_fc_kernel(const float *x1, int C1, const float *x2, int C2, int N,
                   const float *w0, const float *b0, const float *w1, const float *b1, const float *ws,
                   int H, int O, float *out) {
    extern __shared__ float lds[];                       // w0t [C][H|1] | w1t [H][O|1] | wst [C][O|1] | rows [PTS][C] | hid [PTS][H]
    const int C = C1 + C2, width = H > O ? H : O, pts = PN_THREADS / width;
    const int ldh = H | 1, ldo = O | 1;
    float *w0t = lds, *w1t = w0t + (size_t)C * ldh, *wst = w1t + (size_t)H * ldo;
    float *rows = wst + (ws ? (size_t)C * ldo : 0), *hid = rows + (size_t)pts * C;
    transpose_to_lds(w0t, ldh, w0, H, C);
    transpose_to_lds(w1t, ldo, w1, O, H);
    if (ws) transpose_to_lds(wst, ldo, ws, O, C);
    const int lp = threadIdx.x / width, j = threadIdx.x - lp * width;
    for (int n0 = blockIdx.x * pts; n0 < N; n0 += gridDim.x * pts) {
        __syncthreads();
        for (int i = threadIdx.x; i < pts * C; i += PN_THREADS) {
            const int p = i / C, k = i - p * C, n = n0 + p;
            float v = 0.0f;
            if (n < N) v = k < C1 ? x1[(size_t)n * C1 + k] : x2[(size_t)n * C2 + (k - C1)];
            rows[i] = v;
        }
        __syncthreads();
        const int n = n0 + lp;
        const bool live = lp < pts && n < N;
        const float *r = rows + lp * C;
        if (live && j < H) {
            hid[lp * H + j] = fmaxf(b0[j] + dot_cols<true>(w0t, ldh, j, r, C), 0.0f);
        }
        __syncthreads();
        if (live && j < O) {
            const float dx = b1[j] + dot_cols<false>(w1t, ldo, j, hid + lp * H, H);
            const float xs = ws ? dot_cols<false>(wst, ldo, j, r, C) : r[j];      // no layer: size_in == size_out, identity
            out[(size_t)n * O + j] = xs + dx;
        }
    }
}

inline unsigned rows_grid(int N, int pts) {
    size_t g = ((size_t)N + pts - 1) / pts;
    const size_t cap = (size_t)vt_num_cus() * 4;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

}  // namespace

extern "C" {

int vt_linear_rows(const float *x, const float *w, const float *b, int64_t N, int Cin, int Cout, float *out, void *stream) {
    if (N == 0) return 0;
    if (!x || !w || !out || N < 0 || N > INT32_MAX || Cin <= 0 || Cout <= 0) return vt_fail(VT_ERR_INVALID, "vt_linear_rows: bad argument");
    if (Cout > PN_THREADS) return vt_fail(VT_ERR_UNSUPPORTED, "vt_linear_rows: more than 256 output channels");
    const int pts = PN_THREADS / Cout;
    const size_t lds = ((size_t)Cin * (Cout | 1) + (size_t)pts * Cin) * sizeof(float);
    if (lds > 64 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_linear_rows: weights do not fit 64 KiB of LDS");
    hipLaunchKernelGGL(linear_rows_kernel, dim3(rows_grid((int)N, pts)), dim3(PN_THREADS), lds, (hipStream_t)stream,
                       x, w, b, out, (int)N, Cin, Cout);
    return vt_check(hipGetLastError(), "vt_linear_rows");
}

int vt_resblock_fc(const float *x1, int C1, const float *x2, int C2, int64_t N,
                   const float *w0, const float *b0, const float *w1, const float *b1, const float *ws,
                   int H, int O, float *out, void *stream) {
    if (N == 0) return 0;
    if (!x1 || C1 <= 0 || (x2 && C2 <= 0) || !w0 || !b0 || !w1 || !b1 || !out || N < 0 || N > INT32_MAX || H <= 0 || O <= 0)
        return vt_fail(VT_ERR_INVALID, "vt_resblock_fc: bad argument");
    if (!x2) C2 = 0;
    const int C = C1 + C2;
    if (!ws && C != O) return vt_fail(VT_ERR_INVALID, "vt_resblock_fc: no shortcut layer needs size_in == size_out");
    if (H > PN_THREADS || O > PN_THREADS) return vt_fail(VT_ERR_UNSUPPORTED, "vt_resblock_fc: more than 256 hidden / output channels");
    const int pts = PN_THREADS / (H > O ? H : O);
    const size_t lds = ((size_t)C * (H | 1) + (size_t)H * (O | 1) + (ws ? (size_t)C * (O | 1) : 0) + (size_t)pts * (C + H)) * sizeof(float);
    if (lds > 64 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_resblock_fc: weights do not fit 64 KiB of LDS");
    hipLaunchKernelGGL(resblock_fc_kernel, dim3(rows_grid((int)N, pts)), dim3(PN_THREADS), lds, (hipStream_t)stream,
                       x1, C1, x2, C2, (int)N, w0, b0, w1, b1, ws, H, O, out);
    return vt_check(hipGetLastError(), "vt_resblock_fc");
}

}  // extern "C"
