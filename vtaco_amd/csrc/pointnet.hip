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

// ---- backward (training): autograd of the layers above (reference: loss.backward() through src/layers.py:8-50 and the
// nn.Linear calls of pointnet.py:154-162) ---------------------------------------------------------------------------
// natural-layout copy w[rows][n] -> LDS [rows][ld]: sum_j w[j][k] v[j] is then dot_cols(w, ld, k, v, rows)
__device__ __forceinline__ void copy_to_lds(float *dst, int ld, const float *w, int rows, int n) {
    for (int i = threadIdx.x; i < rows * n; i += PN_THREADS) { const int j = i / n, k = i - j * n; dst[j * ld + k] = w[i]; }
}

// Data gradient of ResnetBlockFC at the rows [x1 | x2] given d out [N][O]:
//   h = b0 + W0 relu(x) (recomputed), a = relu(h);  d h = (W1^T d out) . [h > 0];  d x = (W0^T d h) . [x > 0] + Ws^T d out (or + d out)
// Writes d x1 / d x2 and leaves a and d h ([N][H] each) for the weight-gradient GEMMs.
__global__ void __launch_bounds__(PN_THREADS)
resblock_fc_bwd_kernel(const float *x1, int C1, const float *x2, int C2, int N,
                       const float *w0, const float *b0, const float *w1, const float *ws, int H, int O,
                       const float *dout, float *dx1, float *dx2, float *act, float *dh) {
    extern __shared__ float lds[];   // w0t [C][H|1] | w0n [H][C|1] | w1n [O][H|1] | wsn [O][C|1] | xr [PTS][C] | dor [PTS][O] | dhr [PTS][H] | hr [PTS][H]
    const int C = C1 + C2;
    int width = C > H ? C : H; if (O > width) width = O;
    const int pts = PN_THREADS / width;
    const int ldh = H | 1, ldc = C | 1;
    float *w0t = lds, *w0n = w0t + (size_t)C * ldh, *w1n = w0n + (size_t)H * ldc, *wsn = w1n + (size_t)O * ldh;
    float *xr = wsn + (ws ? (size_t)O * ldc : 0), *dor = xr + (size_t)pts * C, *dhr = dor + (size_t)pts * O, *hr = dhr + (size_t)pts * H;
    transpose_to_lds(w0t, ldh, w0, H, C);
    copy_to_lds(w0n, ldc, w0, H, C);
    copy_to_lds(w1n, ldh, w1, O, H);
    if (ws) copy_to_lds(wsn, ldc, ws, O, C);
    const int lp = threadIdx.x / width, j = threadIdx.x - lp * width;
    for (int n0 = blockIdx.x * pts; n0 < N; n0 += gridDim.x * pts) {
        __syncthreads();
        for (int i = threadIdx.x; i < pts * C; i += PN_THREADS) {
            const int pq = i / C, k = i - pq * C, n = n0 + pq;
            float v = 0.0f;
            if (n < N) v = k < C1 ? x1[(size_t)n * C1 + k] : x2[(size_t)n * C2 + (k - C1)];
            xr[i] = v;
        }
        for (int i = threadIdx.x; i < pts * O; i += PN_THREADS) {
            const int pq = i / O, n = n0 + pq;
            dor[i] = n < N ? dout[(size_t)n * O + (i - pq * O)] : 0.0f;
        }
        __syncthreads();
        const int n = n0 + lp;
        const bool live = lp < pts && n < N;
        const float *r = xr + lp * C, *go = dor + lp * O;
        if (live && j < H) {
            const float h = b0[j] + dot_cols<true>(w0t, ldh, j, r, C);
            hr[lp * H + j] = h;
            act[(size_t)n * H + j] = fmaxf(h, 0.0f);
            const float da = dot_cols<false>(w1n, ldh, j, go, O);
            const float g = h > 0.0f ? da : 0.0f;
            dhr[lp * H + j] = g;
            dh[(size_t)n * H + j] = g;
        }
        __syncthreads();
        if (live && j < C) {
            const float back = dot_cols<false>(w0n, ldc, j, dhr + lp * H, H);
            const float v = (r[j] > 0.0f ? back : 0.0f) + (ws ? dot_cols<false>(wsn, ldc, j, go, O) : go[j]);
            if (j < C1) dx1[(size_t)n * C1 + j] = v;
            else if (dx2) dx2[(size_t)n * C2 + (j - C1)] = v;
        }
    }
}

// Weight gradient of a linear layer over tall inputs: dW[m][k] = sum_n G[n][m] X[n][k], db[m] = sum_n G[n][m], with
// X = [x1 | x2] (optionally relu'd), as f32 MFMA outer products (two points per v_mfma_f32_32x32x2_f32) over chunks of
// 1024 points; the per-chunk partials are summed in chunk order by rows_wgrad_reduce_kernel (bit-reproducible).
typedef float pf32x16 __attribute__((ext_vector_type(16)));
constexpr int RW_CHUNK = 1024, RW_PART = 1024 + 32;

__global__ void __launch_bounds__(256)
rows_wgrad_kernel(const float *G, int M, const float *x1, int C1, const float *x2, int C2, int relu_x, int N, float *partial) {
    __shared__ float red[4][RW_PART];
    const int K = C1 + C2;
    const int chunk = blockIdx.x, mt = blockIdx.y, kt = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kk = lane >> 5;
    const int m = mt * 32 + col, k = kt * 32 + col;
    pf32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0f;
    float csum = 0.0f;
    const int p0 = chunk * RW_CHUNK + wave * (RW_CHUNK / 4);
    const int p1 = min(p0 + RW_CHUNK / 4, N);
    // each MFMA contracts two points; eight MFMAs' operands are loaded together, branch-free (indices clamped into the
    // arrays, zeros selected afterwards): one load round trip per 16 points instead of one per 2 (45 -> ~10 us per layer)
    const int mc = min(m, M - 1), kc = min(k, K - 1);
    const float *xsrc = kc < C1 ? x1 + kc : x2 + (kc - C1);
    const int xstride = kc < C1 ? C1 : C2;
    const bool mok = m < M, kok = k < K;
    for (int pb = p0; pb < p1; pb += 16) {
        float gv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = min(pb + 2 * u + kk, N - 1);
            gv[u] = G[(size_t)p * M + mc];
            xv[u] = xsrc[(size_t)p * xstride];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = pb + 2 * u + kk < p1;
            const float g1 = (ok && mok) ? gv[u] : 0.0f;
            float x1v = (ok && kok) ? xv[u] : 0.0f;
            if (relu_x) x1v = fmaxf(x1v, 0.0f);
            csum += g1;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, x1v, acc, 0, 0, 0);     // D[m][k] += G[p][m] * X[p][k]
        }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) red[wave][s * 64 + lane] = acc[s];
    csum += __shfl_xor(csum, 32);
    if (lane < 32) red[wave][1024 + lane] = csum;
    __syncthreads();
    float *dst = partial + (((size_t)chunk * gridDim.y + mt) * gridDim.z + kt) * RW_PART;
    for (int e = threadIdx.x; e < RW_PART; e += 256) dst[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

__global__ void __launch_bounds__(256)
rows_wgrad_reduce_kernel(const float *partial, int nchunks, int M, int K, int MT, int KT, float *dW, float *db) {
    const int mt = blockIdx.x, kt = blockIdx.y;
    for (int e = threadIdx.x; e < RW_PART; e += 256) {
        float s = 0.0f;
        int c = 0;
        for (; c + 8 <= nchunks; c += 8) {                       // eight partials in flight, added in chunk order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partial[(((size_t)(c + u) * MT + mt) * KT + kt) * RW_PART + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; c < nchunks; ++c) s += partial[(((size_t)c * MT + mt) * KT + kt) * RW_PART + e];
        if (e < 1024) {
            const int r = e >> 6, l = e & 63;
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), k = kt * 32 + (l & 31);   // accumulator row of register r, lane half
            if (m < M && k < K) dW[(size_t)m * K + k] = s;
        } else if (kt == 0 && db) {
            const int m = mt * 32 + (e - 1024);
            if (m < M) db[m] = s;
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

int vt_resblock_fc_bwd(const float *x1, int C1, const float *x2, int C2, int64_t N,
                       const float *w0, const float *b0, const float *w1, const float *ws, int H, int O,
                       const float *dout, float *dx1, float *dx2, float *act, float *dh, void *stream) {
    if (N == 0) return 0;
    if (!x1 || C1 <= 0 || (x2 && C2 <= 0) || !w0 || !b0 || !w1 || !dout || !dx1 || !act || !dh || N < 0 || N > INT32_MAX || H <= 0 || O <= 0)
        return vt_fail(VT_ERR_INVALID, "vt_resblock_fc_bwd: bad argument");
    if (!x2) C2 = 0;
    const int C = C1 + C2;
    if (!ws && C != O) return vt_fail(VT_ERR_INVALID, "vt_resblock_fc_bwd: no shortcut layer needs size_in == size_out");
    int width = C > H ? C : H; if (O > width) width = O;
    if (width > PN_THREADS) return vt_fail(VT_ERR_UNSUPPORTED, "vt_resblock_fc_bwd: more than 256 channels");
    const int pts = PN_THREADS / width;
    const size_t lds = ((size_t)C * (H | 1) + (size_t)H * (C | 1) + (size_t)O * (H | 1) + (ws ? (size_t)O * (C | 1) : 0) +
                        (size_t)pts * (C + O + 2 * H)) * sizeof(float);
    if (lds > 64 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_resblock_fc_bwd: weights do not fit 64 KiB of LDS");
    hipLaunchKernelGGL(resblock_fc_bwd_kernel, dim3(rows_grid((int)N, pts)), dim3(PN_THREADS), lds, (hipStream_t)stream,
                       x1, C1, x2, C2, (int)N, w0, b0, w1, ws, H, O, dout, dx1, dx2, act, dh);
    return vt_check(hipGetLastError(), "vt_resblock_fc_bwd");
}

size_t vt_rows_wgrad_workspace_bytes(int64_t N, int M, int K) {
    if (N <= 0 || M <= 0 || K <= 0) return 0;
    const size_t nchunks = (size_t)((N + RW_CHUNK - 1) / RW_CHUNK);
    return nchunks * (size_t)((M + 31) / 32) * (size_t)((K + 31) / 32) * RW_PART * sizeof(float);
}

int vt_rows_wgrad(const float *G, int M, const float *x1, int C1, const float *x2, int C2, int relu_x, int64_t N,
                  void *workspace, size_t workspace_bytes, float *dW, float *db, void *stream) {
    if (!G || M <= 0 || !x1 || C1 <= 0 || (x2 && C2 <= 0) || N <= 0 || N > INT32_MAX || !workspace || !dW)
        return vt_fail(VT_ERR_INVALID, "vt_rows_wgrad: bad argument");
    if (!x2) C2 = 0;
    const int K = C1 + C2, MT = (M + 31) / 32, KT = (K + 31) / 32;
    const int nchunks = (int)((N + RW_CHUNK - 1) / RW_CHUNK);
    if (workspace_bytes < vt_rows_wgrad_workspace_bytes(N, M, K)) return vt_fail(VT_ERR_WORKSPACE, "vt_rows_wgrad: workspace too small");
    hipLaunchKernelGGL(rows_wgrad_kernel, dim3(nchunks, MT, KT), dim3(256), 0, (hipStream_t)stream,
                       G, M, x1, C1, x2, C2, relu_x, (int)N, (float *)workspace);
    hipLaunchKernelGGL(rows_wgrad_reduce_kernel, dim3(MT, KT), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, nchunks, M, K, MT, KT, dW, db);
    return vt_check(hipGetLastError(), "vt_rows_wgrad");
}

}  // extern "C"
