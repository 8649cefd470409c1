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
#include <stdlib.h>

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

// the same sum with a compile-time length: fully unrolled, so the 2 N LDS reads are in flight together instead of one dependent
// read per FMA step (the run-time form is bound by LDS latency: ~37 cycles per FMA measured in the one-launch kernel).  Same partial
// sums, same order: bit-identical to dot_cols.
template <bool RELU, int N>
__device__ __forceinline__ float dot_cols_n(const float *wt, int ld, int j, const float *r) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    float rv[N], wv[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { rv[k] = RELU ? fmaxf(r[k], 0.0f) : r[k]; wv[k] = wt[k * ld + j]; }
#pragma unroll
    for (int k = 0; k + 4 <= N; k += 4) {
        a0 = fmaf(wv[k], rv[k], a0);
        a1 = fmaf(wv[k + 1], rv[k + 1], a1);
        a2 = fmaf(wv[k + 2], rv[k + 2], a2);
        a3 = fmaf(wv[k + 3], rv[k + 3], a3);
    }
#pragma unroll
    for (int k = N & ~3; k < N; ++k) a0 = fmaf(wv[k], rv[k], a0);
    return (a0 + a1) + (a2 + a3);
}

// run-time length, compile-time code for the widths the shipped encoders use (32, 64): the per-layer kernels' dot products
template <bool RELU>
__device__ __forceinline__ float dot_cols_fast(const float *wt, int ld, int j, const float *r, int n) {
    if (n == 64) return dot_cols_n<RELU, 64>(wt, ld, j, r);
    if (n == 32) return dot_cols_n<RELU, 32>(wt, ld, j, r);
    return dot_cols<RELU>(wt, ld, j, r, n);
}

// two rows against the same weight column: each weight read feeds two FMAs (per row the sum is dot_cols's, bit for bit)
template <bool RELU, int N>
__device__ __forceinline__ void dot_cols_n2(const float *wt, int ld, int j, const float *r0, const float *r1, float &s0, float &s1) {
    float a[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
    float wv[N], u0[N], u1[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        wv[k] = wt[k * ld + j];
        u0[k] = RELU ? fmaxf(r0[k], 0.0f) : r0[k];
        u1[k] = RELU ? fmaxf(r1[k], 0.0f) : r1[k];
    }
#pragma unroll
    for (int k = 0; k + 4 <= N; k += 4)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = fmaf(wv[k + e], u0[k + e], a[e]); c[e] = fmaf(wv[k + e], u1[k + e], c[e]); }
    static_assert((N & 3) == 0, "multiple of four");
    s0 = (a[0] + a[1]) + (a[2] + a[3]);
    s1 = (c[0] + c[1]) + (c[2] + c[3]);
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
            out[(size_t)n * Cout + j] = (b ? b[j] : 0.0f) + dot_cols_fast<false>(wt, ldo, j, rows + lp * Cin, Cin);
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
            hid[lp * H + j] = fmaxf(b0[j] + dot_cols_fast<true>(w0t, ldh, j, r, C), 0.0f);
        }
        __syncthreads();
        if (live && j < O) {
            const float dx = b1[j] + dot_cols_fast<false>(w1t, ldo, j, hid + lp * H, H);
            const float xs = ws ? dot_cols_fast<false>(wst, ldo, j, r, C) : r[j];      // no layer: size_in == size_out, identity
            out[(size_t)n * O + j] = xs + dx;
        }
    }
}

// ---- the whole per-point MLP in ONE launch (inference, one voxel index) ------------------------------------------------
// fc_pos -> block 0 -> 4 x (local max-pool over the point's cell, concat, block) -> fc_c used to be eleven launches (two linear
// layers, five blocks, four pools: ~75 us of an 0.75 ms encode, each a chain of three dependent global round trips in front of a few
// hundred FMAs).  The pooling couples only the points of one cell, and the voxel index holds the points sorted by cell: a workgroup
// that owns every cell whose FIRST sorted point falls into its window of 8-16 sorted positions holds complete cells -- its points
// are one contiguous range of sorted positions, of any length (a cell of 8192 points is one workgroup's range) -- so all five blocks
// and the four pools run inside the workgroup with no grid-wide step: the per-cell maxima live in LDS (at most PF_WIN cells), the
// points' feature rows pass from block to block through a scratch array the workgroup alone reads and writes.  Arithmetic and
// summation order are those of linear_rows_kernel / resblock_fc_kernel (same dot_cols): the result equals the launch-per-layer path
// bit for bit (the per-cell maxima are LDS float-max atomics: exact in any order).
constexpr int PF_WIN = 16, PF_PTS = 16, PF_H = 32, PF_BLOCKS = 5, PF_CACHE = 64;

struct PnFusedArgs {
    const float *pts;                     // [B,T,3]
    const int *order, *seg_lo, *seg_hi;   // the voxel index (vt_voxel_build)
    const float *pos_w, *pos_b;           // fc_pos [2H][3], [2H]
    const float *w0[PF_BLOCKS], *b0[PF_BLOCKS], *w1[PF_BLOCKS], *b1[PF_BLOCKS], *ws[PF_BLOCKS];   // fc_0 [H][2H], fc_1 [H][H], shortcut [H][2H]
    const float *c_w, *c_b;               // fc_c [c_dim][H], [c_dim]
    float *scratch;                       // [B,T,H] feature rows by sorted position
    float *out;                           // [B,T,c_dim] by point, or null
    // the voxeliser's mean (generate_grid_features) from the same kernel: every owned cell's mean feature into the zero-filled
    // channels-last grid, and the workgroup's (sum, sum of squares) of those means = GroupNorm partial sums of the grid
    const int *idx;                       // [B,T] cell id per point (with grid_cl)
    float *grid_cl;                       // [B,V,c_dim] or null
    float *part;                          // [B][workgroups per scene][c_dim][2] (with grid_cl)
    long long V;
    int T, c_dim;
    int win;                              // window of sorted positions per workgroup (<= PF_WIN): chosen so that one round of workgroups covers the cloud
};

__global__ void __launch_bounds__(PN_THREADS)
pointnet_fused_kernel(PnFusedArgs a) {
    constexpr int H = PF_H, C = 2 * H, LDH = H | 1;
    __shared__ float w0t[C * LDH], w1t[H * LDH], wst[C * LDH];
    __shared__ float post[3 * (C | 1)], cwt[H * 65];
    __shared__ float bias[3][64];                                  // fc_0.bias | fc_1.bias of the current block, fc_pos.bias / fc_c.bias
    __shared__ float rows[PF_PTS][C], hid[PF_PTS][H], outs[PF_PTS][H];
    __shared__ float smax[PF_WIN][H], pcur[PF_WIN][H];
    // the first PF_CACHE points of the workgroup's range (all of them unless a cell is dense) stay in LDS: point index, cell,
    // coordinates and the feature row between blocks -- a pass then touches no global memory (every global access inside the
    // pass loop is a dependent round trip of ~1.5 us: with them the kernel took 60 us)
    __shared__ int meta_t[PF_CACHE], meta_sg[PF_CACHE];
    __shared__ float srow[PF_CACHE][H], spts[PF_CACHE][3];
    __shared__ int head_rank[PF_WIN], range[2];
    __shared__ int cell_n[PF_WIN], cell_id[PF_WIN], seg_pass[PF_PTS];
    __shared__ float couts[PF_PTS][64];
    const int b = blockIdx.y, T = a.T, win = a.win, wstart = blockIdx.x * win;
    const int *order = a.order + (size_t)b * T, *seg_lo = a.seg_lo + (size_t)b * T, *seg_hi = a.seg_hi + (size_t)b * T;
    const float *pts = a.pts + (size_t)b * T * 3;
    float *scratch = a.scratch + (size_t)b * T * H;
    if (threadIdx.x < 64) {
        // the cells this workgroup owns: their first sorted point lies in the window
        const int i = threadIdx.x, jpos = wstart + i;
        int t = 0, lo = -1, hi = 0;
        if (i < win && jpos < T) { t = order[jpos]; lo = seg_lo[t]; hi = seg_hi[t]; }
        const bool head = i < win && jpos < T && lo == jpos;
        const unsigned long long m = __ballot(head);
        const int rk = __popcll(m & ((1ull << i) - 1ull));
        if (i < win) head_rank[i] = rk;
        if (head && a.grid_cl) { cell_n[rk] = hi - lo; cell_id[rk] = a.idx[(size_t)b * T + t]; }
        const int last = m ? 63 - __builtin_clzll(m) : 0;
        const int end = __shfl(hi, last);
        if (i == 0) { range[0] = m ? wstart + __builtin_ctzll(m) : -1; range[1] = end; }
    }
    __syncthreads();
    const int first = range[0], end = range[1];
    if (first < 0) {                                                // the window lies inside a cell owned further left
        if (a.part && (int)threadIdx.x < 2 * a.c_dim) a.part[((size_t)b * gridDim.x + blockIdx.x) * a.c_dim * 2 + threadIdx.x] = 0.0f;
        return;
    }
    // grid mode: thread c < c_dim adds its channel of the pass's points to the running sum of the current cell, in sorted order
    // (the order of the voxeliser's own mean), and closes a cell when the next one starts
    int run_seg = -1;
    float run_acc = 0.0f, st_sum = 0.0f, st_sq = 0.0f;
    auto close_cell = [&](int sgc, float acc) {
        const float mean = acc / (float)cell_n[sgc];
        a.grid_cl[((size_t)b * a.V + (size_t)cell_id[sgc]) * a.c_dim + threadIdx.x] = mean;
        st_sum += mean; st_sq = fmaf(mean, mean, st_sq);
    };
    const int lp = threadIdx.x / H, j = threadIdx.x - lp * H;
    transpose_to_lds(post, C | 1, a.pos_w, C, 3);
    transpose_to_lds(cwt, 65, a.c_w, a.c_dim, H);
    if (threadIdx.x < 64) { bias[2][threadIdx.x] = a.pos_b[threadIdx.x]; }
    for (int i = threadIdx.x; i < PF_WIN * H; i += PN_THREADS) (&smax[0][0])[i] = -INFINITY;
    for (int i = threadIdx.x; i < PF_CACHE && first + i < end; i += PN_THREADS) {
        const int t = order[first + i];
        meta_t[i] = t; meta_sg[i] = seg_lo[t] - wstart;             // (window offset of the cell's head; its rank below)
        spts[i][0] = pts[3 * t]; spts[i][1] = pts[3 * t + 1]; spts[i][2] = pts[3 * t + 2];
    }
    // a block's weights: five 16-byte pieces and one bias value per thread, requested one block AHEAD (under the previous block's
    // passes) and transposed into LDS at the block's start -- three dependent fetch + store rounds per block otherwise
    float4 wr[5];
    float br = 0.0f;
    auto wfetch = [&](int blk) {
        const float4 *p0 = reinterpret_cast<const float4 *>(a.w0[blk]), *p1 = reinterpret_cast<const float4 *>(a.w1[blk]),
                     *ps = reinterpret_cast<const float4 *>(a.ws[blk]);
        wr[0] = p0[threadIdx.x]; wr[1] = p0[threadIdx.x + PN_THREADS]; wr[2] = p1[threadIdx.x];
        wr[3] = ps[threadIdx.x]; wr[4] = ps[threadIdx.x + PN_THREADS];
        if (threadIdx.x < H) br = a.b0[blk][threadIdx.x];
        else if (threadIdx.x < 2 * H) br = a.b1[blk][threadIdx.x - H];
    };
    auto put = [&](float *wt, int n4, int i, const float4 &v) {     // piece i of a [rows][4 n4] matrix -> wt[k][row]
        const int row = i / n4, k = (i - row * n4) * 4;
        wt[k * LDH + row] = v.x; wt[(k + 1) * LDH + row] = v.y; wt[(k + 2) * LDH + row] = v.z; wt[(k + 3) * LDH + row] = v.w;
    };
    wfetch(0);
    for (int blk = 0; blk < PF_BLOCKS; ++blk) {
        __syncthreads();                                            // the previous block is done with the weights and with pcur
        put(w0t, C / 4, threadIdx.x, wr[0]); put(w0t, C / 4, threadIdx.x + PN_THREADS, wr[1]);
        put(w1t, H / 4, threadIdx.x, wr[2]);
        put(wst, C / 4, threadIdx.x, wr[3]); put(wst, C / 4, threadIdx.x + PN_THREADS, wr[4]);
        if (threadIdx.x < H) bias[0][threadIdx.x] = br;
        else if (threadIdx.x < 2 * H) bias[1][threadIdx.x - H] = br;
        if (blk + 1 < PF_BLOCKS) wfetch(blk + 1);
        if (blk > 0)
            for (int i = threadIdx.x; i < PF_WIN * H; i += PN_THREADS) { (&pcur[0][0])[i] = (&smax[0][0])[i]; (&smax[0][0])[i] = -INFINITY; }
        for (int q0 = first; q0 < end; q0 += PF_PTS) {
            __syncthreads();
            // a thread serves output channel j of TWO points of the pass (lp and lp + 8): every weight read from LDS feeds two FMAs
            int tq[2], sgv[2];
            bool livev[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pl = lp + 8 * u, q = q0 + pl, ci = q - first;
                const bool live = q < end, cached = ci < PF_CACHE;
                int t = 0, sg = 0;
                if (live) {
                    if (cached) { t = meta_t[ci]; sg = head_rank[meta_sg[ci]]; }
                    else { t = order[q]; sg = head_rank[seg_lo[t] - wstart]; }
                }
                tq[u] = t; livev[u] = live; sgv[u] = sg;
                if (blk == 0) {
                    // rows = fc_pos(p): two outputs per thread and point
                    float p3[3] = {0.f, 0.f, 0.f};
                    if (live) {
                        if (cached) { p3[0] = spts[ci][0]; p3[1] = spts[ci][1]; p3[2] = spts[ci][2]; }
                        else { p3[0] = pts[3 * t]; p3[1] = pts[3 * t + 1]; p3[2] = pts[3 * t + 2]; }
                    }
                    rows[pl][j] = bias[2][j] + dot_cols_n<false, 3>(post, C | 1, j, p3);
                    rows[pl][j + H] = bias[2][j + H] + dot_cols_n<false, 3>(post, C | 1, j + H, p3);
                } else {
                    rows[pl][j] = live ? (cached ? srow[ci][j] : scratch[(size_t)q * H + j]) : 0.0f;
                    rows[pl][j + H] = live ? pcur[sg][j] : 0.0f;
                }
            }
            __syncthreads();
            {
                float h0, h1;
                dot_cols_n2<true, C>(w0t, LDH, j, rows[lp], rows[lp + 8], h0, h1);
                hid[lp][j] = fmaxf(bias[0][j] + h0, 0.0f);
                hid[lp + 8][j] = fmaxf(bias[0][j] + h1, 0.0f);
            }
            __syncthreads();
            float d0, d1, x0, x1;
            dot_cols_n2<false, H>(w1t, LDH, j, hid[lp], hid[lp + 8], d0, d1);
            dot_cols_n2<false, C>(wst, LDH, j, rows[lp], rows[lp + 8], x0, x1);
            const float o[2] = {x0 + (bias[1][j] + d0), x1 + (bias[1][j] + d1)};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pl = lp + 8 * u, q = q0 + pl, ci = q - first;
                outs[pl][j] = o[u];
                if (livev[u] && blk + 1 < PF_BLOCKS) {
                    if (ci < PF_CACHE) srow[ci][j] = o[u]; else scratch[(size_t)q * H + j] = o[u];
                    // into the cell's maximum: an LDS float max per (point, channel), all in parallel (max is exact in any order; a
                    // single thread walking the pass's points was a chain of ~50 dependent LDS round trips: 2.3 us per pass)
                    __hip_atomic_fetch_max(&smax[sgv[u]][j], o[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __syncthreads();
            if (blk + 1 < PF_BLOCKS) {
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int pl = lp + 8 * u;
                    if (j == 0) seg_pass[pl] = livev[u] ? sgv[u] : -1;
                    if (livev[u])
                        for (int oc = j; oc < a.c_dim; oc += H) {   // fc_c straight from the last block's rows
                            const float cv = a.c_b[oc] + dot_cols_n<false, H>(cwt, 65, oc, outs[pl]);
                            if (a.out) a.out[((size_t)b * T + tq[u]) * a.c_dim + oc] = cv;
                            couts[pl][oc] = cv;
                        }
                }
                if (a.grid_cl) {
                    __syncthreads();
                    if ((int)threadIdx.x < a.c_dim) {
                        int sg16[PF_PTS];
                        float v16[PF_PTS];
#pragma unroll
                        for (int i = 0; i < PF_PTS; ++i) { sg16[i] = seg_pass[i]; v16[i] = couts[i][threadIdx.x]; }
#pragma unroll
                        for (int i = 0; i < PF_PTS; ++i) {
                            if (sg16[i] < 0) continue;
                            if (sg16[i] != run_seg) { if (run_seg >= 0) close_cell(run_seg, run_acc); run_seg = sg16[i]; run_acc = 0.0f; }
                            run_acc += v16[i];
                        }
                    }
                }
            }
        }
    }
    if (a.grid_cl && (int)threadIdx.x < a.c_dim) {
        if (run_seg >= 0) close_cell(run_seg, run_acc);
        float *dst = a.part + (((size_t)b * gridDim.x + blockIdx.x) * a.c_dim + threadIdx.x) * 2;
        dst[0] = st_sum; dst[1] = st_sq;
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
            const float h = b0[j] + dot_cols_fast<true>(w0t, ldh, j, r, C);
            hr[lp * H + j] = h;
            act[(size_t)n * H + j] = fmaxf(h, 0.0f);
            const float da = dot_cols_fast<false>(w1n, ldh, j, go, O);
            const float g = h > 0.0f ? da : 0.0f;
            dhr[lp * H + j] = g;
            dh[(size_t)n * H + j] = g;
        }
        __syncthreads();
        if (live && j < C) {
            const float back = dot_cols_fast<false>(w0n, ldc, j, dhr + lp * H, H);
            const float v = (r[j] > 0.0f ? back : 0.0f) + (ws ? dot_cols_fast<false>(wsn, ldc, j, go, O) : go[j]);
            if (j < C1) dx1[(size_t)n * C1 + j] = v;
            else if (dx2) dx2[(size_t)n * C2 + (j - C1)] = v;
        }
    }
}

// ---- ResnetBlockFC backward on the f32 matrix core (hidden 32: the shipped encoders) -------------------------------------------
// resblock_fc_bwd_kernel spends 58 us per call on 24 000 points: every FMA of its dot products reads LDS.  Here a wave owns 32
// points and every product is D[row][point] += A[row][k] B[k][point] on v_mfma_f32_32x32x2_f32 (exact f32 products, f32
// accumulation; the sums run in another order than the per-thread dot products: f32 rounding level):
//   h   = b0 + W0 relu(x)           32 (C = 64) k-steps     act = relu(h)
//   da  = W1^T dout                 16                      dh  = (h > 0 ? da : 0)          (both in the accumulator layout: no exchange)
//   dx  = (x > 0 ? W0^T dh : 0)     16 per 32 input rows    (dh through LDS: the B operand wants it by k, the accumulator has it by lane)
//       + Ws^T dout (or dout)       16 per 32 input rows    on the same accumulators
// Weights as [row][K | 1] in LDS (odd pitches: conflict-free for both operand orders), a wave's x / dout / dh tiles likewise.
typedef float pn_f32x16 __attribute__((ext_vector_type(16)));
typedef float pn_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int pn_chan(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ void pn_store16(float *row, const pn_f32x16 &v, int h) {
    pn_f32x4 *q = reinterpret_cast<pn_f32x4 *>(row + 4 * h);
    q[0] = pn_f32x4{v[0], v[1], v[2], v[3]}; q[2] = pn_f32x4{v[4], v[5], v[6], v[7]};
    q[4] = pn_f32x4{v[8], v[9], v[10], v[11]}; q[6] = pn_f32x4{v[12], v[13], v[14], v[15]};
}
constexpr int RM_WAVES = 4;
__host__ __device__ inline size_t rm_lds_floats(int C, bool has_ws) {
    return (size_t)32 * (C | 1) + 32 * 33 + (has_ws ? (size_t)32 * (C | 1) : 0) + (size_t)RM_WAVES * (32 * (C | 1) + 2 * 32 * 33);
}

// (C and the shortcut are template parameters: with run-time trip counts every k-step waited for its own two LDS reads -- 39 us per
// call; unrolled, a product's reads are all in flight before its first MFMA)
template <int C, bool HAS_WS>
__global__ void __launch_bounds__(RM_WAVES * 64)
resblock_fc_bwd_mfma_kernel(const float *x1, int C1, const float *x2, int C2, int N, const float *w0, const float *b0, const float *w1,
                            const float *ws_, const float *dout, float *dx1, float *dx2, float *act, float *dh) {
    extern __shared__ float rl[];
    constexpr int ldc = C | 1, nkb = C / 32;
    const float *ws = HAS_WS ? ws_ : nullptr;
    float *w0n = rl, *w1n = w0n + 32 * ldc, *wsn = w1n + 32 * 33;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, kk = lane >> 5;
    float *xr = wsn + (HAS_WS ? 32 * ldc : 0) + wave * (32 * ldc + 2 * 32 * 33), *dor = xr + 32 * ldc, *gr = dor + 32 * 33;
    for (int i = threadIdx.x; i < 32 * C; i += RM_WAVES * 64) { const int r = i / C, k = i - r * C; w0n[r * ldc + k] = w0[i]; if (HAS_WS) wsn[r * ldc + k] = ws[i]; }
    for (int i = threadIdx.x; i < 32 * 32; i += RM_WAVES * 64) w1n[(i >> 5) * 33 + (i & 31)] = w1[i];
    const int ntile = (N + 31) / 32;
    for (int tile = blockIdx.x * RM_WAVES + wave; tile - wave < ntile; tile += gridDim.x * RM_WAVES) {
        const int n0 = tile * 32;
        const bool live = tile < ntile;
        __syncthreads();                                            // the weights; the previous tile's readers
        if (live) {
            for (int i = lane; i < 32 * C; i += 64) {
                const int p = i / C, k = i - p * C, n = min(n0 + p, N - 1);
                xr[p * ldc + k] = k < C1 ? x1[(size_t)n * C1 + k] : x2[(size_t)n * C2 + (k - C1)];
            }
            for (int i = lane; i < 32 * 32; i += 64) { const int p = i >> 5, n = min(n0 + p, N - 1); dor[p * 33 + (i & 31)] = dout[(size_t)n * 32 + (i & 31)]; }
        }
        __syncthreads();
        pn_f32x16 h, da;
        const int n = n0 + m;                                       // this lane's point (accumulator column)
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { h[r] = b0[pn_chan(r, kk)]; da[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < C / 2; ++s)
                h = __builtin_amdgcn_mfma_f32_32x32x2f32(w0n[m * ldc + 2 * s + kk], fmaxf(xr[m * ldc + 2 * s + kk], 0.0f), h, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s)
                da = __builtin_amdgcn_mfma_f32_32x32x2f32(w1n[(2 * s + kk) * 33 + m], dor[m * 33 + 2 * s + kk], da, 0, 0, 0);
            pn_f32x16 a, g;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a[r] = fmaxf(h[r], 0.0f);
                g[r] = h[r] > 0.0f ? da[r] : 0.0f;
                gr[m * 33 + pn_chan(r, kk)] = g[r];
            }
            if (n < N) { pn_store16(act + (size_t)n * 32, a, kk); pn_store16(dh + (size_t)n * 32, g, kk); }
        }
        __syncthreads();                                            // dh by k for the next products
        if (live) {
#pragma unroll
            for (int kb = 0; kb < nkb; ++kb) {
                pn_f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(w0n[(2 * s + kk) * ldc + kb * 32 + m], gr[m * 33 + 2 * s + kk], d, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = kb * 32 + pn_chan(r, kk);
                    d[r] = (xr[m * ldc + k] > 0.0f ? d[r] : 0.0f) + (HAS_WS ? 0.0f : dor[m * 33 + k]);
                }
                if (HAS_WS) {
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        d = __builtin_amdgcn_mfma_f32_32x32x2f32(wsn[(2 * s + kk) * ldc + kb * 32 + m], dor[m * 33 + 2 * s + kk], d, 0, 0, 0);
                }
                if (n < N) {
                    const int c0 = kb * 32;
                    if (c0 < C1) pn_store16(dx1 + (size_t)n * C1 + c0, d, kk);
                    else if (dx2) pn_store16(dx2 + (size_t)n * C2 + (c0 - C1), d, kk);
                }
            }
        }
    }
}


// Weight gradient of a linear layer over tall inputs: dW[m][k] = sum_n G[n][m] X[n][k], db[m] = sum_n G[n][m], with
// X = [x1 | x2] (optionally relu'd), as f32 MFMA outer products (two points per v_mfma_f32_32x32x2_f32) over chunks of
// 1024 points; the per-chunk partials are summed in chunk order by rows_wgrad_reduce_kernel (bit-reproducible).
typedef float pf32x16 __attribute__((ext_vector_type(16)));
constexpr int RW_CHUNK = 1024, RW_PART = 1024 + 32;

__device__ __forceinline__ void rows_wgrad_tile(const float *G, int M, const float *x1, int C1, const float *x2, int C2, int relu_x, int N,
                                                float *dst, int chunk, int mt, int kt, float (*red)[RW_PART]) {
    const int K = C1 + C2;
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
    for (int e = threadIdx.x; e < RW_PART; e += 256) dst[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

__global__ void __launch_bounds__(256)
rows_wgrad_kernel(const float *G, int M, const float *x1, int C1, const float *x2, int C2, int relu_x, int N, float *partial) {
    __shared__ float red[4][RW_PART];
    const int chunk = blockIdx.x, mt = blockIdx.y, kt = blockIdx.z;
    rows_wgrad_tile(G, M, x1, C1, x2, C2, relu_x, N, partial + (((size_t)chunk * gridDim.y + mt) * gridDim.z + kt) * RW_PART, chunk, mt, kt, red);
}

__device__ __forceinline__ void rows_wgrad_reduce_tile(const float *partial, size_t chunk_stride, int nchunks, int M, int K, int mt, int kt,
                                                       float *dW, float *db) {
    for (int e = threadIdx.x; e < RW_PART; e += 256) {
        float s = 0.0f;
        int c = 0;
        for (; c + 8 <= nchunks; c += 8) {                       // eight partials in flight, added in chunk order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(c + u) * chunk_stride + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; c < nchunks; ++c) s += partial[(size_t)c * chunk_stride + e];
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

__global__ void __launch_bounds__(256)
rows_wgrad_reduce_kernel(const float *partial, int nchunks, int M, int K, int MT, int KT, float *dW, float *db) {
    const int mt = blockIdx.x, kt = blockIdx.y;
    rows_wgrad_reduce_tile(partial + ((size_t)mt * KT + kt) * RW_PART, (size_t)MT * KT * RW_PART, nchunks, M, K, mt, kt, dW, db);
}

// Several such products over the same rows in ONE pair of launches (a ResnetBlockFC's three weight gradients -- fc_1, fc_0, the
// shortcut -- were six launches of ~10 us each for a few MFLOP): blockIdx.y walks the 32 x 32 tiles of all jobs; every tile has its own
// partials [chunk][RW_PART] behind `tile_base` floats of the workspace.
constexpr int RW_MAX_JOBS = 3, RW_MAX_TILES = 16;
struct RowsJob { const float *G, *x1, *x2; int M, C1, C2, relu_x; float *dW, *db; };
struct RowsJobs { RowsJob job[RW_MAX_JOBS]; int njobs, ntiles, tile_job[RW_MAX_TILES], tile_mt[RW_MAX_TILES], tile_kt[RW_MAX_TILES]; };

__global__ void __launch_bounds__(256)
rows_wgrad_multi_kernel(RowsJobs js, int N, int nchunks, float *partial) {
    __shared__ float red[4][RW_PART];
    const int chunk = blockIdx.x, t = blockIdx.y;
    const RowsJob &j = js.job[js.tile_job[t]];
    rows_wgrad_tile(j.G, j.M, j.x1, j.C1, j.x2, j.C2, j.relu_x, N, partial + ((size_t)t * nchunks + chunk) * RW_PART, chunk, js.tile_mt[t], js.tile_kt[t], red);
}

__global__ void __launch_bounds__(256)
rows_wgrad_multi_reduce_kernel(RowsJobs js, int nchunks, const float *partial) {
    const int t = blockIdx.x;
    const RowsJob &j = js.job[js.tile_job[t]];
    rows_wgrad_reduce_tile(partial + (size_t)t * nchunks * RW_PART, RW_PART, nchunks, j.M, j.C1 + j.C2, js.tile_mt[t], js.tile_kt[t], j.dW, j.db);
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
    // hidden 32 (the shipped encoders): the products on the f32 matrix core, a wave per 32 points (VTACO_RESBLOCK_MFMA=0: the FMA kernel)
    static const bool mfma_off = getenv("VTACO_RESBLOCK_MFMA") && getenv("VTACO_RESBLOCK_MFMA")[0] == '0';
    const bool aligned = ((reinterpret_cast<uintptr_t>(dx1) | reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(dh) |
                           reinterpret_cast<uintptr_t>(dx2)) & 15) == 0;
    if (!mfma_off && H == 32 && O == 32 && (C == 32 || C == 64) && (C1 & 31) == 0 && (C2 & 31) == 0 && aligned) {
        const size_t lb = rm_lds_floats(C, ws != nullptr) * sizeof(float);
        const int ntile = (int)((N + 31) / 32), wgs = (ntile + RM_WAVES - 1) / RM_WAVES, cap = vt_num_cus();
        const dim3 grid((unsigned)(wgs < cap ? wgs : cap)), block(RM_WAVES * 64);
        auto launch = [&](auto kern) -> int {
            const hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
            if (e != hipSuccess) return vt_check(e, "vt_resblock_fc_bwd: hipFuncSetAttribute");
            hipLaunchKernelGGL(kern, grid, block, lb, (hipStream_t)stream, x1, C1, x2, C2, (int)N, w0, b0, w1, ws, dout, dx1, dx2, act, dh);
            return vt_check(hipGetLastError(), "vt_resblock_fc_bwd");
        };
        if (C == 64) return ws ? launch(&resblock_fc_bwd_mfma_kernel<64, true>) : launch(&resblock_fc_bwd_mfma_kernel<64, false>);
        return ws ? launch(&resblock_fc_bwd_mfma_kernel<32, true>) : launch(&resblock_fc_bwd_mfma_kernel<32, false>);
    }
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

static int pointnet_window(int B, int T) {
    // one workgroup per CU is resident (256 registers x 4 waves): the smallest window that still covers the cloud in one round of
    // workgroups keeps a workgroup at one 16-point pass per block (3000 points: 12 positions -> 250 workgroups, 25 us; 16 -> 40 us; 10 -> 44 us)
    const int64_t pts_total = (int64_t)B * T, cus = vt_num_cus();
    int win = (int)((pts_total + cus - 1) / cus);
    if (win < 8) win = 8;
    if (win > PF_WIN) win = PF_WIN;
    return win;
}

int vt_pointnet_mlp_stat_blocks(int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    const int win = pointnet_window(B, T);
    return (T + win - 1) / win;
}

int vt_pointnet_mlp_fused(const float *pts, int B, int T, const int *order, const int *seg_lo, const int *seg_hi,
                          const float *pos_w, const float *pos_b, const float *const *block_w, int hidden,
                          const float *c_w, const float *c_b, int c_dim, float *scratch, float *out,
                          const int *idx, int R, float *grid_cl, float *grid_part, void *stream) {
    if (!pts || !order || !seg_lo || !seg_hi || !pos_w || !pos_b || !block_w || !c_w || !c_b || !scratch || (!out && !grid_cl) || B <= 0 || T <= 0)
        return vt_fail(VT_ERR_INVALID, "vt_pointnet_mlp_fused: bad argument");
    if (grid_cl && (!idx || !grid_part || R < 1)) return vt_fail(VT_ERR_INVALID, "vt_pointnet_mlp_fused: the grid output needs idx, R and grid_part");
    if (hidden != PF_H || c_dim <= 0 || c_dim > 64)
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_pointnet_mlp_fused: built for hidden_dim 32 and c_dim <= 64 (the shipped encoders); use the per-layer kernels");
    for (int i = 0; i < 5 * PF_BLOCKS; ++i)
        if (i % 5 != 1 && i % 5 != 3 && (reinterpret_cast<uintptr_t>(block_w[i]) & 15))
            return vt_fail(VT_ERR_UNSUPPORTED, "vt_pointnet_mlp_fused: weight matrices must be 16-byte aligned");
    PnFusedArgs a;
    a.pts = pts; a.order = order; a.seg_lo = seg_lo; a.seg_hi = seg_hi; a.pos_w = pos_w; a.pos_b = pos_b;
    for (int i = 0; i < PF_BLOCKS; ++i) {
        a.w0[i] = block_w[5 * i]; a.b0[i] = block_w[5 * i + 1]; a.w1[i] = block_w[5 * i + 2]; a.b1[i] = block_w[5 * i + 3]; a.ws[i] = block_w[5 * i + 4];
        if (!a.w0[i] || !a.b0[i] || !a.w1[i] || !a.b1[i] || !a.ws[i]) return vt_fail(VT_ERR_INVALID, "vt_pointnet_mlp_fused: null block weight");
    }
    a.c_w = c_w; a.c_b = c_b; a.scratch = scratch; a.out = out; a.T = T; a.c_dim = c_dim;
    a.idx = idx; a.grid_cl = grid_cl; a.part = grid_cl ? grid_part : nullptr; a.V = (long long)R * R * R;
    const int win = pointnet_window(B, T);
    a.win = win;
    hipLaunchKernelGGL(pointnet_fused_kernel, dim3((unsigned)((T + win - 1) / win), (unsigned)B), dim3(PN_THREADS), 0, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_pointnet_mlp_fused");
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

size_t vt_resblock_wgrad_workspace_bytes(int64_t N, int C, int H, int O, int has_shortcut) {
    if (N <= 0 || C <= 0 || H <= 0 || O <= 0) return 0;
    const size_t nchunks = (size_t)((N + RW_CHUNK - 1) / RW_CHUNK);
    const size_t tiles = (size_t)((O + 31) / 32) * ((H + 31) / 32) + (size_t)((H + 31) / 32) * ((C + 31) / 32) +
                         (has_shortcut ? (size_t)((O + 31) / 32) * ((C + 31) / 32) : 0);
    return tiles > RW_MAX_TILES ? 0 : tiles * nchunks * RW_PART * sizeof(float);
}

int vt_resblock_wgrad(const float *x1, int C1, const float *x2, int C2, int64_t N, const float *act, const float *dh, const float *dout,
                      int H, int O, void *workspace, size_t workspace_bytes,
                      float *dw0, float *db0, float *dw1, float *db1, float *dws, void *stream) {
    if (!x1 || C1 <= 0 || (x2 && C2 <= 0) || !act || !dh || !dout || N <= 0 || N > INT32_MAX || H <= 0 || O <= 0 || !workspace || !dw0 || !dw1)
        return vt_fail(VT_ERR_INVALID, "vt_resblock_wgrad: bad argument");
    if (!x2) C2 = 0;
    const int C = C1 + C2;
    const size_t need = vt_resblock_wgrad_workspace_bytes(N, C, H, O, dws != nullptr);
    if (!need) return vt_fail(VT_ERR_UNSUPPORTED, "vt_resblock_wgrad: more than 16 tiles of 32 x 32: use vt_rows_wgrad per product");
    if (workspace_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_resblock_wgrad: workspace too small");
    RowsJobs js{};
    js.job[0] = RowsJob{dout, act, nullptr, O, H, 0, 0, dw1, db1};              // fc_1:     dW1 = dout^T relu(h)
    js.job[1] = RowsJob{dh, x1, x2, H, C1, C2, 1, dw0, db0};                    // fc_0:     dW0 = dh^T relu(x)
    js.job[2] = RowsJob{dout, x1, x2, O, C1, C2, 0, dws, nullptr};              // shortcut: dWs = dout^T x
    js.njobs = dws ? 3 : 2;
    int t = 0;
    for (int j = 0; j < js.njobs; ++j) {
        const int MT = (js.job[j].M + 31) / 32, KT = (js.job[j].C1 + js.job[j].C2 + 31) / 32;
        for (int mt = 0; mt < MT; ++mt)
            for (int kt = 0; kt < KT; ++kt) { js.tile_job[t] = j; js.tile_mt[t] = mt; js.tile_kt[t] = kt; ++t; }
    }
    js.ntiles = t;
    const int nchunks = (int)((N + RW_CHUNK - 1) / RW_CHUNK);
    hipLaunchKernelGGL(rows_wgrad_multi_kernel, dim3(nchunks, t), dim3(256), 0, (hipStream_t)stream, js, (int)N, nchunks, (float *)workspace);
    hipLaunchKernelGGL(rows_wgrad_multi_reduce_kernel, dim3(t), dim3(256), 0, (hipStream_t)stream, js, nchunks, (const float *)workspace);
    return vt_check(hipGetLastError(), "vt_resblock_wgrad");
}

}  // extern "C"
