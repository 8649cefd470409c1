// TransformerFusion backward for gfx950 (training): gradients of vt_fusion_fwd_train with respect to c_img, c and every
// parameter of the two attention units (reference src/TransformerFusion.py:65-113 RelationUnit incl. the column
// re-normalisation :104, :13-25 TransNonlinear with its two dropouts, :116-146 / :173-219 encoder / decoder layers with
// InstanceNorm over the N points, :311-333; entered from src/conv_onet/models/decoder.py:258 under loss.backward()).
//
// Like the forward, nothing N x N is ever stored: every pass recomputes its 32 x 32 score tiles on the matrix core.
// With  E = exp(Q K^T),  P = E / l  (row softmax),  s_k = sum_q P_qk,  A = P / (1e-9 + s),  V' = V / (1e-9 + s),  O = A V:
//   dV_k   = sum_q A_qk dO_q                                       (pass "dv": fixed k, streamed q)
//   t'_k   = (V_k . dV_k) / (1e-9 + s_k)                           (the column re-normalisation's own term, per key, no pass)
//   dP_qk  = dO_q . V'_k - t'_k ;   u_q = sum_k P_qk dP_qk ;   dS_qk = P_qk (dP_qk - u_q)
//   dQ_q   = sum_k dS_qk K_k = M1_q - u_q M2_q,  M1 = sum_k P (dP) K,  M2 = sum_k P K   (pass "dq": fixed q, streamed k; u falls out)
//   dK_k   = sum_q dS_qk Q_q                                       (pass "dk": fixed k, streamed q)
// Exact-f32 matrix core throughout (v_mfma_f32_32x32x2_f32): a score tile leaves the MFMA with the fixed index on the lane and
// the streamed index in the 16 registers -- the B operand of the accumulation that follows, as in the forward.
// The per-point parts (InstanceNorm, LayerNorm, the TransNonlinear MLP with its replayed dropout masks, trans_conv, the
// l2-normalised projections) run as accumulator-layout MLPs, one wave per 32 points; weight gradients are reduced over fixed
// chunks of points in chunk order (bit-reproducible, no atomics).
#include "fusion_common.h"

namespace {

constexpr float LOG2E = 1.44269504088896341f;
constexpr int QS = 66;      // LDS row pitch of a streamed 64-d tile (2 mod 64: the 64 lanes of an A-operand read hit 64 banks)
constexpr int PS = 34;      // ... of a streamed 32-d payload tile

// ---- transposed epilogue weights for the data gradient: fragment [s][lane] = W[row chan_of(s, lane>>5)][col lane&31] ----
constexpr int FT_WT = 0, FT_W1A = 1024, FT_W1B = 2048, FT_W2A = 3072, FT_W2B = 4096, FT_BLOB = 5120;
__global__ void fusion_pack_t_kernel(FusionUnitDev u, float *blob) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < FT_BLOB; e += gridDim.x * blockDim.x) {
        const int L = e >> 10, s = (e >> 6) & 15, l = e & 63, i = l & 31, h = l >> 5, k = chan_of(s, h);
        float v;
        if (L == 0) v = u.Wt[k * 32 + i];                 // dd = Wt^T dr0
        else if (L == 1) v = u.l1w[k * 32 + i];           // dr += W1a^T dha   (linear1 rows 0..31)
        else if (L == 2) v = u.l1w[(32 + k) * 32 + i];    // dr += W1b^T dhb   (linear1 rows 32..63)
        else if (L == 3) v = u.l2w[k * 64 + i];           // dha = W2a^T dY    (linear2 cols 0..31)
        else v = u.l2w[k * 64 + 32 + i];                  // dhb = W2b^T dY    (linear2 cols 32..63)
        blob[e] = v;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ---- f32 projections with their norms: Qf = l2norm(Xq WQ^T), Kf = l2norm(Xk WK^T); one wave per point, lane = key column ----
__global__ void __launch_bounds__(256)
fb_proj_kernel(const float *Xq, const float *Xk, FusionUnitDev u, float *Qf, float *Kf, float *nq, float *nk, int P) {
    __shared__ float wq[32][64], wk[32][64];
    __shared__ float xs[4][2][32];
    for (int e = threadIdx.x; e < 2048; e += 256) {
        const int c = e >> 6, d = e & 63;
        wq[c][d] = u.WQ[d * 32 + c];
        wk[c][d] = u.WK[d * 32 + c];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave;
    if (p < P) xs[wave][lane >> 5][lane & 31] = (lane < 32 ? Xq : Xk)[(size_t)p * 32 + (lane & 31)];
    __syncthreads();
    if (p >= P) return;
    float q = 0.0f, k = 0.0f;
#pragma unroll
    for (int c = 0; c < 32; ++c) { q = fmaf(xs[wave][0][c], wq[c][lane], q); k = fmaf(xs[wave][1][c], wk[c][lane], k); }
    const float n_q = fmaxf(sqrtf(wave_sum(q * q)), 1e-12f), n_k = fmaxf(sqrtf(wave_sum(k * k)), 1e-12f);
    Qf[(size_t)p * 64 + lane] = q / n_q;
    Kf[(size_t)p * 64 + lane] = k / n_k;
    if (lane == 0) { nq[p] = n_q; nk[p] = n_k; }
}

// ---- out = relu(InstanceNorm_N(Z)) backward; one block per scene -------------------------------------------------------
__global__ void __launch_bounds__(1024)
fb_inorm_bwd_kernel(const float *Z, const float *dOut, float *dZ, int N, int pitch = 32) {
    __shared__ float red[2][32][33];
    __shared__ float st[4][32];                                     // mean, rstd, mean(g), mean(g xhat)
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t base = (size_t)blockIdx.x * N * pitch + 32 * blockIdx.y;       // (scene, 32-channel slice of a wider model)
    const float *z = Z + base, *dout = dOut + base;
    float *dz = dZ + base;
    auto reduce = [&](float a, float b, int slot_a, int slot_b, float scale) {
        red[0][g][c] = a; red[1][g][c] = b;
        __syncthreads();
        if (threadIdx.x < 64) {
            const int w = threadIdx.x >> 5, cc = threadIdx.x & 31;
            float t = 0.0f;
            for (int i = 0; i < 32; ++i) t += red[w][i][cc];
            st[w ? slot_b : slot_a][cc] = t * scale;
        }
        __syncthreads();
    };
    const float shift = z[c];                                      // mean = z_0 + mean(z - z_0), as the forward
    float s = 0.0f;
    for (int n = g; n < N; n += 32) s += z[(size_t)n * pitch + c] - shift;
    reduce(s, 0.0f, 0, 3, 1.0f / (float)N);
    const float m = shift + st[0][c];
    s = 0.0f;
    for (int n = g; n < N; n += 32) { const float d = z[(size_t)n * pitch + c] - m; s = fmaf(d, d, s); }
    reduce(s, 0.0f, 1, 3, 1.0f / (float)N);
    if (threadIdx.x < 32) st[1][threadIdx.x] = 1.0f / sqrtf(st[1][threadIdx.x] + 1e-5f);
    __syncthreads();
    const float rs = st[1][c];
    float sg = 0.0f, sgx = 0.0f;
    for (int n = g; n < N; n += 32) {
        const float xh = (z[(size_t)n * pitch + c] - m) * rs;
        const float gg = xh > 0.0f ? dout[(size_t)n * pitch + c] : 0.0f;
        sg += gg; sgx = fmaf(gg, xh, sgx);
    }
    reduce(sg, sgx, 2, 3, 1.0f / (float)N);
    const float mg = st[2][c], mgx = st[3][c];
    for (int n = g; n < N; n += 32) {
        const float xh = (z[(size_t)n * pitch + c] - m) * rs;
        const float gg = xh > 0.0f ? dout[(size_t)n * pitch + c] : 0.0f;
        dz[(size_t)n * pitch + c] = rs * (gg - mg - xh * mgx);
    }
}

// ---- per-point backward of  Z = Xq + LN(r + drop2(W2 drop1(relu(W1 r + b1)) + b2)),  r = relu(Wt (Xq - O)) ------------------
// One wave per 32 points, everything in the accumulator layout (point on the lane, channel chan_of(r,h) in register r); the
// forward epilogue is recomputed from Xq and the saved O with the replayed dropout masks.
struct EpiOut {
    float *gXq, *dO, *dOs, *D, *dR0, *R, *dH0, *Hd, *dY, *dy, *dyxh;
};
__global__ void __launch_bounds__(256)
fb_epilogue_bwd_kernel(const float *Xq, const float *O, const float *dZ, const float *linv, const float *blob_f, const float *blob_t,
                       EpiOut w, int P, DropCfg dc) {
    __shared__ __attribute__((aligned(16))) float lf[FU_BLOB];
    __shared__ __attribute__((aligned(16))) float lt[FT_BLOB];
    for (int i = threadIdx.x; i < FU_BLOB; i += 256) lf[i] = blob_f[i];
    for (int i = threadIdx.x; i < FT_BLOB; i += 256) lt[i] = blob_t[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= P) return;
    const int p = min(p0 + j, P - 1);
    const bool live = p0 + j < P;
    const uint32_t pt = (uint32_t)p;
    const f32x16 x = load_acc16(Xq + (size_t)p * 32, h), o = load_acc16(O + (size_t)p * 32, h);
    const f32x16 dz = load_acc16(dZ + (size_t)p * 32, h);
    f32x16 zero;
#pragma unroll
    for (int s = 0; s < 16; ++s) zero[s] = 0.0f;
    // ---- forward, recomputed
    const f32x16 d = x - o;
    const f32x16 r0 = dense32<false>(zero, lf + FU_WT, d, lane);
    const f32x16 r = relu16(r0);
    f32x16 ha0 = load_frag16(lf + FU_BIAS + 0 * 32 + h * 16), hb0 = load_frag16(lf + FU_BIAS + 1 * 32 + h * 16);
    ha0 = dense32<false>(ha0, lf + FU_W1A, r, lane);
    hb0 = dense32<false>(hb0, lf + FU_W1B, r, lane);
    f32x16 ma, mb, m2;                                              // dropout factors (0 or 1/(1-p))
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        ma[s] = drop_mask(dc, 0, pt, (uint32_t)chan_of(s, h));
        mb[s] = drop_mask(dc, 0, pt, (uint32_t)(32 + chan_of(s, h)));
        m2[s] = drop_mask(dc, 1, pt, (uint32_t)chan_of(s, h));
    }
    f32x16 ha, hb;
#pragma unroll
    for (int s = 0; s < 16; ++s) { ha[s] = relu1(ha0[s]) * ma[s]; hb[s] = relu1(hb0[s]) * mb[s]; }
    f32x16 y = load_frag16(lf + FU_BIAS + 2 * 32 + h * 16);
    y = dense32<false>(y, lf + FU_W2A, ha, lane);
    y = dense32<false>(y, lf + FU_W2B, hb, lane);
    f32x16 t2;
#pragma unroll
    for (int s = 0; s < 16; ++s) t2[s] = fmaf(y[s], m2[s], r[s]);
    float m = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) m += t2[s];
    m += __shfl_xor(m, 32);
    m *= (1.0f / 32.0f);
    float var = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) { const float c = t2[s] - m; var = fmaf(c, c, var); }
    var += __shfl_xor(var, 32);
    const float rstd = 1.0f / sqrtf(var * (1.0f / 32.0f) + 1e-5f);
    const f32x16 ga = load_frag16(lf + FU_BIAS + 3 * 32 + h * 16);
    // ---- backward
    f32x16 xh, dxh, dyxh;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        xh[s] = (t2[s] - m) * rstd;
        dxh[s] = dz[s] * ga[s];
        dyxh[s] = dz[s] * xh[s];
        s1 += dxh[s];
        s2 = fmaf(dxh[s], xh[s], s2);
    }
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    s1 *= (1.0f / 32.0f); s2 *= (1.0f / 32.0f);
    f32x16 dt2, dY;
#pragma unroll
    for (int s = 0; s < 16; ++s) { dt2[s] = rstd * (dxh[s] - s1 - xh[s] * s2); dY[s] = dt2[s] * m2[s]; }
    f32x16 dha = dense32<false>(zero, lt + FT_W2A, dY, lane), dhb = dense32<false>(zero, lt + FT_W2B, dY, lane);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        dha[s] = ha0[s] > 0.0f ? dha[s] * ma[s] : 0.0f;
        dhb[s] = hb0[s] > 0.0f ? dhb[s] * mb[s] : 0.0f;
    }
    f32x16 dr = dense32<false>(dt2, lt + FT_W1A, dha, lane);
    dr = dense32<false>(dr, lt + FT_W1B, dhb, lane);
    f32x16 dr0;
#pragma unroll
    for (int s = 0; s < 16; ++s) dr0[s] = r0[s] > 0.0f ? dr[s] : 0.0f;
    const f32x16 dd = dense32<false>(zero, lt + FT_WT, dr0, lane);
    if (!live) return;
    const float li = linv[p];
    f32x16 gx, dO, dOs;
#pragma unroll
    for (int s = 0; s < 16; ++s) { gx[s] = dz[s] + dd[s]; dO[s] = -dd[s]; dOs[s] = -dd[s] * li; }
    const size_t a32 = (size_t)p * 32, a64 = (size_t)p * 64;
    store_acc16(w.gXq + a32, gx, h); store_acc16(w.dO + a32, dO, h); store_acc16(w.dOs + a32, dOs, h);
    store_acc16(w.D + a32, d, h); store_acc16(w.dR0 + a32, dr0, h); store_acc16(w.R + a32, r, h);
    store_acc16(w.dH0 + a64, dha, h); store_acc16(w.dH0 + a64 + 32, dhb, h);
    store_acc16(w.Hd + a64, ha, h); store_acc16(w.Hd + a64 + 32, hb, h);
    store_acc16(w.dY + a32, dY, h); store_acc16(w.dy + a32, dz, h); store_acc16(w.dyxh + a32, dyxh, h);
}

// ---- streamed tiles of the N x N passes ------------------------------------------------------------------------------
// A workgroup (4 waves) owns 128 fixed rows and streams every row of the other operand through LDS in 32-row tiles,
// double-buffered: a 64-d row (Q or K, f32), a 32-d payload row and two per-row scalars.  Rows beyond N arrive as zeros.
constexpr int NT = 256, NFIX = 128;
struct StreamTile {
    float qk[32 * QS];
    float pay[32 * PS];
    float sc0[32], sc1[32];
};
struct StreamRegs {
    f32x4 a0, a1, b;
    float s0, s1;
};
__device__ __forceinline__ void stream_fetch(StreamRegs &r, const float *qk, const float *pay, const float *s0, const float *s1,
                                             float pay_scale_by_s0, int row0, int N, int pitch = 32, float s1_default = 1.0f) {
    const int t = threadIdx.x;
    {   // 32 rows x 16 float4 of the 64-d part: two per thread
        const int i0 = t, i1 = t + 256;
        const int r0 = row0 + (i0 >> 4), r1 = row0 + (i1 >> 4);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        r.a0 = r0 < N ? *reinterpret_cast<const f32x4 *>(qk + (size_t)r0 * 64 + (i0 & 15) * 4) : z;
        r.a1 = r1 < N ? *reinterpret_cast<const f32x4 *>(qk + (size_t)r1 * 64 + (i1 & 15) * 4) : z;
        const int rp = row0 + (t >> 3);
        r.b = rp < N ? *reinterpret_cast<const f32x4 *>(pay + (size_t)rp * pitch + (t & 7) * 4) : z;
        if (pay_scale_by_s0 != 0.0f && rp < N) {                      // V' = V / (1e-9 + s)
            const float inv = 1.0f / (1e-9f + s0[rp]);
            r.b = r.b * inv;
        }
    }
    r.s0 = 0.0f; r.s1 = 0.0f;
    if (t < 32) {
        const int rr = row0 + t;
        if (rr < N) { r.s0 = s0 ? s0[rr] : 1.0f; r.s1 = s1 ? s1[rr] : s1_default; }
    }
}
__device__ __forceinline__ void stream_store(StreamTile &tile, const StreamRegs &r) {
    const int t = threadIdx.x;
    {
        const int i0 = t, i1 = t + 256;
        float *d0 = tile.qk + (i0 >> 4) * QS + (i0 & 15) * 4, *d1 = tile.qk + (i1 >> 4) * QS + (i1 & 15) * 4;
        d0[0] = r.a0.x; d0[1] = r.a0.y; d0[2] = r.a0.z; d0[3] = r.a0.w;
        d1[0] = r.a1.x; d1[1] = r.a1.y; d1[2] = r.a1.z; d1[3] = r.a1.w;
        float *dp = tile.pay + (t >> 3) * PS + (t & 7) * 4;
        dp[0] = r.b.x; dp[1] = r.b.y; dp[2] = r.b.z; dp[3] = r.b.w;
    }
    if (t < 32) { tile.sc0[t] = r.s0; tile.sc1[t] = r.s1; }
}
// score tile: D[i][j] = stream_row_i . fixed_row_j (64-d); `fixed[s]` = fixed row j, element 2s + kk, times log2(e)
__device__ __forceinline__ f32x16 score_f32(const float *tile_qk, const float (&fixed)[32], int j, int kk) {
    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = mfma(tile_qk[j * QS + 2 * s + kk], fixed[s], acc);
    return acc;
}
// D[i][j] = stream_payload_row_i . fixed_payload_row_j (32-d)
__device__ __forceinline__ f32x16 pay_dot(const float *tile_pay, const float (&fixed)[16], int j, int kk) {
    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mfma(tile_pay[j * PS + 2 * s + kk], fixed[s], acc);
    return acc;
}
__device__ __forceinline__ void load_fixed64(float (&f)[32], const float *row, int kk, float scale) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(row + 4 * i);
        f[2 * i] = (kk ? t.y : t.x) * scale;
        f[2 * i + 1] = (kk ? t.w : t.z) * scale;
    }
}
__device__ __forceinline__ void load_fixed32(float (&f)[16], const float *row, int kk, float scale) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(row + 4 * i);
        f[2 * i] = (kk ? t.y : t.x) * scale;
        f[2 * i + 1] = (kk ? t.w : t.z) * scale;
    }
}

// pass "dv": dV_k = sum_q A_qk dO_q and t'_k.  fixed: K rows; streamed: Q rows with payload linv_q dO_q
__global__ void __launch_bounds__(NT)
fb_dv_kernel(const float *Qf, const float *Kf, const float *dOs, const float *V, const float *scol, float *dV, float *tp, int N, int nrb, int B,
             int pitch = 32, int off = 0, int acc_tp = 0) {          // (pitch, off: a 32-channel slice of a wider payload; acc_tp: t' sums over the slices)
    __shared__ __attribute__((aligned(16))) StreamTile tiles[2];
    int rb, b;
    chunk_of_workgroup(nrb, B, rb, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const size_t base = (size_t)b * N;
    Qf += base * 64; Kf += base * 64; dOs += base * pitch + off; V += base * pitch + off; scol += base; dV += base * pitch + off; tp += base;
    const int k0 = rb * NFIX + wave * 32, k = min(k0 + j, N - 1);
    float fixed[32];
    load_fixed64(fixed, Kf + (size_t)k * 64, h, LOG2E);
    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0f;
    const int ntile = (N + 31) / 32;
    StreamRegs sr;
    stream_fetch(sr, Qf, dOs, nullptr, nullptr, 0.0f, 0, N, pitch);
    stream_store(tiles[0], sr);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const StreamTile &cur = tiles[t & 1];
        if (t + 1 < ntile) stream_fetch(sr, Qf, dOs, nullptr, nullptr, 0.0f, (t + 1) * 32, N, pitch);
        f32x16 e = score_f32(cur.qk, fixed, j, h);                 // lane (k,h) reg r: streamed q = chan_of(r,h)
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(e[r]);
        // dV^T[c][k] += sum_q dOs[q][c] E[q][k]: A lane (c,kk) = payload[q = chan_of(s,kk)][c]
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = mfma(cur.pay[chan_of(s, h) * PS + j], e[s], acc);
        if (t + 1 < ntile) stream_store(tiles[(t + 1) & 1], sr);
        __syncthreads();
    }
    const float inv = 1.0f / (1e-9f + scol[k]);
    const f32x16 v = load_acc16(V + (size_t)k * pitch, h);
    float dot = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) { acc[s] *= inv; dot = fmaf(v[s], acc[s], dot); }
    dot += __shfl_xor(dot, 32);
    if (k0 + j < N) {
        store_acc16(dV + (size_t)k * pitch, acc, h);
        if (h == 0) tp[k] = (acc_tp ? tp[k] : 0.0f) + dot * inv;
    }
}

// pass "dq": dQ_q = M1_q - u_q M2_q and u_q.  fixed: Q rows (+ dO_q, 1/l_q); streamed: K rows with payload V'_k, scalar t'_k
__global__ void __launch_bounds__(NT)
fb_dq_kernel(const float *Qf, const float *Kf, const float *dO, const float *V, const float *scol, const float *tp,
             const float *linv, float *dQ, float *u, int N, int nrb, int B,
             int pitch = 32, int off = 0, int first = 1, float *M2 = nullptr) {
    // M2 != null (a wider model, one launch per 32-channel slice of dO / V'): dP_qk = sum over the slices of dO_q . V'_k - t'_k is linear in
    // the payload, so a slice adds ITS share of M1 and u (t' enters with the first slice only); M2 comes out once; fbw_dq_combine finishes
    __shared__ __attribute__((aligned(16))) StreamTile tiles[2];
    int rb, b;
    chunk_of_workgroup(nrb, B, rb, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const size_t base = (size_t)b * N;
    Qf += base * 64; Kf += base * 64; dO += base * pitch + off; V += base * pitch + off; scol += base; tp += base; linv += base; dQ += base * 64; u += base;
    if (M2) M2 += base * 64;
    const int q0 = rb * NFIX + wave * 32, q = min(q0 + j, N - 1);
    float fixed[32], dofix[16];
    load_fixed64(fixed, Qf + (size_t)q * 64, h, LOG2E);
    load_fixed32(dofix, dO + (size_t)q * pitch, h, 1.0f);
    const float li = linv[q];
    f32x16 m1a, m1b, m2a, m2b;
#pragma unroll
    for (int s = 0; s < 16; ++s) { m1a[s] = 0.0f; m1b[s] = 0.0f; m2a[s] = 0.0f; m2b[s] = 0.0f; }
    float usum = 0.0f;
    const int ntile = (N + 31) / 32;
    StreamRegs sr;
    // scalars: sc0 = column sum s_k (scales the payload V -> V'), stored as ... see below; sc1 = t'_k
    stream_fetch(sr, Kf, V, scol, first ? tp : nullptr, 1.0f, 0, N, pitch, 0.0f);
    stream_store(tiles[0], sr);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const StreamTile &cur = tiles[t & 1];
        if (t + 1 < ntile) stream_fetch(sr, Kf, V, scol, first ? tp : nullptr, 1.0f, (t + 1) * 32, N, pitch, 0.0f);
        f32x16 e = score_f32(cur.qk, fixed, j, h);                 // lane (q,h) reg r: streamed k = chan_of(r,h)
        const f32x16 g = pay_dot(cur.pay, dofix, j, h);            // dO_q . V'_k
        const f32x16 tt = load_acc16(cur.sc1, h);
        const int kbase = t * 32;
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pe = (kbase + chan_of(r, h) < N) ? __builtin_amdgcn_exp2f(e[r]) * li : 0.0f;     // P_qk
            e[r] = pe;
            a[r] = pe * (g[r] - tt[r]);
            usum += a[r];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float *krow = cur.qk + chan_of(s, h) * QS;
            const float ka = krow[j], kb = krow[32 + j];
            m1a = mfma(ka, a[s], m1a); m1b = mfma(kb, a[s], m1b);
            m2a = mfma(ka, e[s], m2a); m2b = mfma(kb, e[s], m2b);
        }
        if (t + 1 < ntile) stream_store(tiles[(t + 1) & 1], sr);
        __syncthreads();
    }
    usum += __shfl_xor(usum, 32);
    if (M2) {
        if (q0 + j < N) {
            if (first) {
                store_acc16(M2 + (size_t)q * 64, m2a, h);
                store_acc16(M2 + (size_t)q * 64 + 32, m2b, h);
            } else {
                m1a = m1a + load_acc16(dQ + (size_t)q * 64, h);
                m1b = m1b + load_acc16(dQ + (size_t)q * 64 + 32, h);
                usum += u[q];
            }
            store_acc16(dQ + (size_t)q * 64, m1a, h);
            store_acc16(dQ + (size_t)q * 64 + 32, m1b, h);
            if (h == 0) u[q] = usum;
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) { m1a[s] -= usum * m2a[s]; m1b[s] -= usum * m2b[s]; }
    if (q0 + j < N) {
        store_acc16(dQ + (size_t)q * 64, m1a, h);
        store_acc16(dQ + (size_t)q * 64 + 32, m1b, h);
        if (h == 0) u[q] = usum;
    }
}

// pass "dk": dK_k = sum_q dS_qk Q_q.  fixed: K rows (+ V'_k, t'_k); streamed: Q rows with payload dO_q, scalars 1/l_q and u_q
__global__ void __launch_bounds__(NT)
fb_dk_kernel(const float *Qf, const float *Kf, const float *dO, const float *V, const float *scol, const float *tp,
             const float *linv, const float *u, float *dK, int N, int nrb, int B,
             int pitch = 32, int off = 0, int first = 1) {            // (a slice of a wider payload: t'_k and u_q enter with the first, dK adds up)
    __shared__ __attribute__((aligned(16))) StreamTile tiles[2];
    int rb, b;
    chunk_of_workgroup(nrb, B, rb, b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const size_t base = (size_t)b * N;
    Qf += base * 64; Kf += base * 64; dO += base * pitch + off; V += base * pitch + off; scol += base; tp += base; linv += base; u += base; dK += base * 64;
    const int k0 = rb * NFIX + wave * 32, k = min(k0 + j, N - 1);
    float fixed[32], vfix[16];
    load_fixed64(fixed, Kf + (size_t)k * 64, h, LOG2E);
    load_fixed32(vfix, V + (size_t)k * pitch, h, 1.0f / (1e-9f + scol[k]));
    const float tk = first ? tp[k] : 0.0f;
    f32x16 da, db;
#pragma unroll
    for (int s = 0; s < 16; ++s) { da[s] = 0.0f; db[s] = 0.0f; }
    const int ntile = (N + 31) / 32;
    StreamRegs sr;
    stream_fetch(sr, Qf, dO, linv, first ? u : nullptr, 0.0f, 0, N, pitch, 0.0f);                  // rows beyond N: 1/l = 0 -> no contribution
    stream_store(tiles[0], sr);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const StreamTile &cur = tiles[t & 1];
        if (t + 1 < ntile) stream_fetch(sr, Qf, dO, linv, first ? u : nullptr, 0.0f, (t + 1) * 32, N, pitch, 0.0f);
        const f32x16 e = score_f32(cur.qk, fixed, j, h);           // lane (k,h) reg r: streamed q = chan_of(r,h)
        const f32x16 g = pay_dot(cur.pay, vfix, j, h);             // dO_q . V'_k
        const f32x16 ll = load_acc16(cur.sc0, h), uu = load_acc16(cur.sc1, h);
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = __builtin_amdgcn_exp2f(e[r]) * ll[r] * (g[r] - tk - uu[r]);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float *qrow = cur.qk + chan_of(s, h) * QS;
            da = mfma(qrow[j], a[s], da);
            db = mfma(qrow[32 + j], a[s], db);
        }
        if (t + 1 < ntile) stream_store(tiles[(t + 1) & 1], sr);
        __syncthreads();
    }
    if (k0 + j < N) {
        if (!first) { da = da + load_acc16(dK + (size_t)k * 64, h); db = db + load_acc16(dK + (size_t)k * 64 + 32, h); }
        store_acc16(dK + (size_t)k * 64, da, h);
        store_acc16(dK + (size_t)k * 64 + 32, db, h);
    }
}

// ---- backward of the l2-normalised projections; one wave per point -------------------------------------------------
// dPq = (dQ - Q (Q.dQ)) / |Pq| (written over dQ), likewise dPk; gXq += dPq WQ; gXk = dPk WK + dV WV
__global__ void __launch_bounds__(256)
fb_proj_bwd_kernel(const float *Qf, const float *Kf, const float *nq, const float *nk, float *dQ, float *dK, const float *dV,
                   FusionUnitDev u, float *gXq, float *gXk, int P) {
    __shared__ float wq[64][33], wk[64][33], wv[32][33];
    __shared__ float ds[4][3][64];
    for (int e = threadIdx.x; e < 2048; e += 256) { wq[e >> 5][e & 31] = u.WQ[e]; wk[e >> 5][e & 31] = u.WK[e]; }
    for (int e = threadIdx.x; e < 1024; e += 256) wv[e >> 5][e & 31] = u.WV[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave;
    float dpq = 0.0f, dpk = 0.0f;
    if (p < P) {
        const float q = Qf[(size_t)p * 64 + lane], k = Kf[(size_t)p * 64 + lane];
        const float dq = dQ[(size_t)p * 64 + lane], dk = dK[(size_t)p * 64 + lane];
        const float sq = wave_sum(q * dq), sk = wave_sum(k * dk);
        dpq = (dq - q * sq) / nq[p];
        dpk = (dk - k * sk) / nk[p];
        dQ[(size_t)p * 64 + lane] = dpq;
        dK[(size_t)p * 64 + lane] = dpk;
        ds[wave][0][lane] = dpq; ds[wave][1][lane] = dpk;
        if (lane < 32) ds[wave][2][lane] = dV[(size_t)p * 32 + lane];
    }
    __syncthreads();
    if (p >= P) return;
    const int c = lane & 31;
    float acc = 0.0f;
    if (lane < 32) {
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc = fmaf(ds[wave][0][d], wq[d][c], acc);
        gXq[(size_t)p * 32 + c] += acc;
    } else {
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc = fmaf(ds[wave][1][d], wk[d][c], acc);
#pragma unroll 8
        for (int d = 0; d < 32; ++d) acc = fmaf(ds[wave][2][d], wv[d][c], acc);
        gXk[(size_t)p * 32 + c] = acc;
    }
}

// ---- weight gradients: dW[o][i] = sum_p G[p][o] X[p][i] (K = 0: column sums of G), reduced over 512-point chunks ----------
constexpr int WCHUNK = 512, WJOBS = 10, WMAX = 2048;
struct WJob {
    const float *G, *X;
    float *dst;
    int M, K, accumulate;
};
struct WJobs {
    WJob j[WJOBS];
};
__global__ void __launch_bounds__(256)
fb_wgrad_kernel(WJobs jobs, int P, float *partials, int nchunk) {
    __shared__ float gs[64][64], xs[64][64];
    const WJob jb = jobs.j[blockIdx.y];
    const int M = jb.M, K = jb.K ? jb.K : 1, t = threadIdx.x;
    const int o = t & (M - 1), i0 = t / M, istep = 256 / M, nacc = (M * K + 255) / 256;
    float acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = 0.0f;
    const int pbeg = blockIdx.x * WCHUNK, pend = min(pbeg + WCHUNK, P);
    for (int p0 = pbeg; p0 < pend; p0 += 64) {
        const int np = min(64, pend - p0);
        __syncthreads();
        for (int e = t; e < 64 * M; e += 256) { const int pp = e / M, oo = e - pp * M; gs[pp][oo] = pp < np ? jb.G[(size_t)(p0 + pp) * M + oo] : 0.0f; }
        if (jb.K) for (int e = t; e < 64 * K; e += 256) { const int pp = e / K, ii = e - pp * K; xs[pp][ii] = pp < np ? jb.X[(size_t)(p0 + pp) * K + ii] : 0.0f; }
        __syncthreads();
        if (jb.K) {
            for (int pp = 0; pp < 64; ++pp) {
                const float g = gs[pp][o];
#pragma unroll
                for (int a = 0; a < 8; ++a) if (a < nacc) acc[a] = fmaf(g, xs[pp][i0 + istep * a], acc[a]);
            }
        } else if (t < M) {
            for (int pp = 0; pp < 64; ++pp) acc[0] += gs[pp][o];
        }
    }
    float *dst = partials + ((size_t)blockIdx.y * nchunk + blockIdx.x) * WMAX;
    if (jb.K) {
#pragma unroll
        for (int a = 0; a < 8; ++a) if (a < nacc) dst[o * K + i0 + istep * a] = acc[a];
    } else if (t < M) {
        dst[o] = acc[0];
    }
}
__global__ void __launch_bounds__(256)
fb_wreduce_kernel(WJobs jobs, const float *partials, int nchunk) {
    const WJob jb = jobs.j[blockIdx.x];
    const int n = jb.M * (jb.K ? jb.K : 1);
    for (int e = threadIdx.x; e < n; e += 256) {
        float s = 0.0f;
        for (int c = 0; c < nchunk; ++c) s += partials[((size_t)blockIdx.x * nchunk + c) * WMAX + e];
        jb.dst[e] = jb.accumulate ? jb.dst[e] + s : s;
    }
}

__global__ void fb_add_kernel(const float *a, const float *b, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = b ? a[i] + b[i] : a[i];
}

__global__ void fb_mask_kernel(DropCfg dc, uint32_t which, int width, float *out, size_t n) {   // (dc.shift: 6 at d_model 32, 8 beyond)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = drop_mask(dc, which, (uint32_t)(i / width), (uint32_t)(i % width));
}

// ---- scratch of one backward call (re-used by the three attention calls) ---------------------------------------------------
struct BwdWs {
    float *Qf, *Kf, *nq, *nk, *dZ, *gXq, *gXk, *dO, *dOs, *D, *dR0, *R, *dH0, *Hd, *dY, *dy, *dyxh, *dV, *tp, *u, *dQ, *dK;
    float *dM, *dT, *partials, *blob_f[2], *blob_t[2];
};
size_t bwd_layout(int B, int N, BwdWs *ws, char *base) {
    const size_t P = (size_t)B * N, nchunk = (P + WCHUNK - 1) / WCHUNK;
    size_t off = 0;
    auto take = [&](size_t floats) { float *p = base ? (float *)(base + off) : nullptr; off += (floats * 4 + 255) / 256 * 256; return p; };
    BwdWs w;
    w.Qf = take(P * 64); w.Kf = take(P * 64); w.nq = take(P); w.nk = take(P);
    w.dZ = take(P * 32); w.gXq = take(P * 32); w.gXk = take(P * 32); w.dO = take(P * 32); w.dOs = take(P * 32);
    w.D = take(P * 32); w.dR0 = take(P * 32); w.R = take(P * 32); w.dH0 = take(P * 64); w.Hd = take(P * 64);
    w.dY = take(P * 32); w.dy = take(P * 32); w.dyxh = take(P * 32);
    w.dV = take(P * 32); w.tp = take(P); w.u = take(P); w.dQ = take(P * 64); w.dK = take(P * 64);
    w.dM = take(P * 32); w.dT = take(P * 32);
    w.partials = take((size_t)WJOBS * nchunk * WMAX);
    for (int i = 0; i < 2; ++i) { w.blob_f[i] = take(FU_BLOB); w.blob_t[i] = take(FT_BLOB); }
    if (ws) *ws = w;
    return off;
}

// backward of one attention call: dOut -> (dXq, dXk) and the unit's parameter gradients
void unit_bwd(int call, const float *Xq, const float *Xk, const float *dOut, const FusionUnitDev &u, const vt_fusion_unit_grads &g,
              int accumulate, const float *blob_f, const float *blob_t, const FusionSaved &sv, const BwdWs &w, float *dXq, float *dXk,
              int B, int N, float p_drop, unsigned long long seed, hipStream_t s) {
    const int P = B * N;
    const dim3 pg((P + 3) / 4), eg((P + 127) / 128), tg((unsigned)((N + NFIX - 1) / NFIX) * (unsigned)B);
    const int nrb = (N + NFIX - 1) / NFIX;
    hipLaunchKernelGGL(fb_proj_kernel, pg, dim3(256), 0, s, Xq, Xk, u, w.Qf, w.Kf, w.nq, w.nk, P);
    hipLaunchKernelGGL(fb_inorm_bwd_kernel, dim3(B), dim3(1024), 0, s, (const float *)sv.Z[call], dOut, w.dZ, N);
    const EpiOut eo{w.gXq, w.dO, w.dOs, w.D, w.dR0, w.R, w.dH0, w.Hd, w.dY, w.dy, w.dyxh};
    hipLaunchKernelGGL(fb_epilogue_bwd_kernel, eg, dim3(256), 0, s, Xq, (const float *)sv.O[call], (const float *)w.dZ,
                       (const float *)sv.linv[call], blob_f, blob_t, eo, P, drop_cfg(p_drop, seed, (uint32_t)call));
    hipLaunchKernelGGL(fb_dv_kernel, tg, dim3(NT), 0, s, (const float *)w.Qf, (const float *)w.Kf, (const float *)w.dOs,
                       (const float *)sv.V[call], (const float *)sv.s[call], w.dV, w.tp, N, nrb, B);
    hipLaunchKernelGGL(fb_dq_kernel, tg, dim3(NT), 0, s, (const float *)w.Qf, (const float *)w.Kf, (const float *)w.dO,
                       (const float *)sv.V[call], (const float *)sv.s[call], (const float *)w.tp, (const float *)sv.linv[call], w.dQ, w.u, N, nrb, B);
    hipLaunchKernelGGL(fb_dk_kernel, tg, dim3(NT), 0, s, (const float *)w.Qf, (const float *)w.Kf, (const float *)w.dO,
                       (const float *)sv.V[call], (const float *)sv.s[call], (const float *)w.tp, (const float *)sv.linv[call],
                       (const float *)w.u, w.dK, N, nrb, B);
    hipLaunchKernelGGL(fb_proj_bwd_kernel, pg, dim3(256), 0, s, (const float *)w.Qf, (const float *)w.Kf, (const float *)w.nq,
                       (const float *)w.nk, w.dQ, w.dK, (const float *)w.dV, u, w.gXq, w.gXk, P);
    WJobs jobs;
    jobs.j[0] = WJob{w.dQ, Xq, g.WQ, 64, 32, accumulate};
    jobs.j[1] = WJob{w.dK, Xk, g.WK, 64, 32, accumulate};
    jobs.j[2] = WJob{w.dV, Xk, g.WV, 32, 32, accumulate};
    jobs.j[3] = WJob{w.dR0, w.D, g.trans_conv, 32, 32, accumulate};
    jobs.j[4] = WJob{w.dH0, w.R, g.linear1_w, 64, 32, accumulate};
    jobs.j[5] = WJob{w.dH0, nullptr, g.linear1_b, 64, 0, accumulate};
    jobs.j[6] = WJob{w.dY, w.Hd, g.linear2_w, 32, 64, accumulate};
    jobs.j[7] = WJob{w.dY, nullptr, g.linear2_b, 32, 0, accumulate};
    jobs.j[8] = WJob{w.dyxh, nullptr, g.norm2_w, 32, 0, accumulate};
    jobs.j[9] = WJob{w.dy, nullptr, g.norm2_b, 32, 0, accumulate};
    const int nchunk = (P + WCHUNK - 1) / WCHUNK;
    hipLaunchKernelGGL(fb_wgrad_kernel, dim3(nchunk, WJOBS), dim3(256), 0, s, jobs, P, w.partials, nchunk);
    hipLaunchKernelGGL(fb_wreduce_kernel, dim3(WJOBS), dim3(256), 0, s, jobs, (const float *)w.partials, nchunk);
    const size_t n = (size_t)P * 32;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dXk == nullptr) {                                           // self-attention: Xq and Xk are the same tensor
        hipLaunchKernelGGL(fb_add_kernel, dim3(blocks), dim3(256), 0, s, (const float *)w.gXq, (const float *)w.gXk, dXq, n);
    } else {
        hipLaunchKernelGGL(fb_add_kernel, dim3(blocks), dim3(256), 0, s, (const float *)w.gXq, (const float *)nullptr, dXq, n);
        hipLaunchKernelGGL(fb_add_kernel, dim3(blocks), dim3(256), 0, s, (const float *)w.gXk, (const float *)nullptr, dXk, n);
    }
}

// =====================================================================================================================================
// d_model C in {64, 96, 128} (the reference's AttentionDecoder defaults to c_dim = d_model = 128, decoder.py:176-207): the backward of
// fusion_fwd_train_wide.  The three N x N passes above are LINEAR in their payload (dO, V'), so they run once per 32-channel slice of it
// with the per-key / per-query scalars (t'_k, u_q) entering in the first slice and the slices' results added up (the score tiles are
// recomputed per slice: C / 32 times the 32-wide cost).  The per-point parts -- projections, the epilogue MLP with LayerNorm over C
// channels and the replayed dropout masks, the projections' backward -- are plain VALU kernels, one wave per point with the weights in
// LDS (a step has 16 k points: ~1 GFLOP), and the ten weight gradients are vt_rows_wgrad's tall-skinny products.
// =====================================================================================================================================
extern "C" int vt_rows_wgrad(const float *G, int M, const float *x1, int C1, const float *x2, int C2, int relu_x, int64_t N,
                             void *workspace, size_t workspace_bytes, float *dW, float *db, void *stream);
extern "C" size_t vt_rows_wgrad_workspace_bytes(int64_t N, int M, int K);

// Qf = l2norm(Xq WQ^T), Kf = l2norm(Xk WK^T) in f32 with their norms; one wave per point, lane = key column
template <int C>
__global__ void __launch_bounds__(256)
fbw_proj_kernel(const float *Xq, const float *Xk, FusionUnitDev u, float *Qf, float *Kf, float *nq, float *nk, int P) {
    extern __shared__ float fbw_lds[];
    float *wq = fbw_lds, *wk = wq + C * 64, *xs = wk + C * 64;        // [C][64] x 2, [4 waves][2][C]
    for (int e = threadIdx.x; e < 64 * C; e += 256) {
        const int d = e / C, c = e - d * C;
        wq[c * 64 + d] = u.WQ[e];
        wk[c * 64 + d] = u.WK[e];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    for (int p = blockIdx.x * 4 + wave; p < P; p += gridDim.x * 4) {
        float *xw = xs + wave * 2 * C;
        for (int c = lane; c < C; c += 64) { xw[c] = Xq[(size_t)p * C + c]; xw[C + c] = Xk[(size_t)p * C + c]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float q = 0.0f, k = 0.0f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) { q = fmaf(xw[c], wq[c * 64 + lane], q); k = fmaf(xw[C + c], wk[c * 64 + lane], k); }
        const float n_q = fmaxf(sqrtf(wave_sum(q * q)), 1e-12f), n_k = fmaxf(sqrtf(wave_sum(k * k)), 1e-12f);
        Qf[(size_t)p * 64 + lane] = q / n_q;
        Kf[(size_t)p * 64 + lane] = k / n_k;
        if (lane == 0) { nq[p] = n_q; nk[p] = n_k; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// per-point backward of  Z = Xq + LN(r + drop2(W2 drop1(relu(W1 r + b1)) + b2)),  r = relu(Wt (Xq - O))  at width C: one wave per point,
// channel c on lane c % 64 (two per lane beyond 64), vectors exchanged through a per-wave LDS strip, weights in LDS once per workgroup
// (rows padded to an odd pitch: W v reads a row per lane, W^T v a column per lane, both conflict-free)
template <int C>
__global__ void __launch_bounds__(256)
fbw_epilogue_bwd_kernel(const float *Xq, const float *O, const float *dZ, const float *linv, FusionUnitDev u, EpiOut w, int P, DropCfg dc) {
    constexpr int PT = C + 1, NC = (C + 63) / 64;
    extern __shared__ float fbw_lds[];
    float *wt = fbw_lds, *w1 = wt + C * PT, *w2 = w1 + 64 * PT, *strip = w2 + C * 65;    // Wt [C][C+1], W1 [64][C+1], W2 [C][65]
    for (int e = threadIdx.x; e < C * C; e += 256) wt[(e / C) * PT + e % C] = u.Wt[e];
    for (int e = threadIdx.x; e < 64 * C; e += 256) w1[(e / C) * PT + e % C] = u.l1w[e];
    for (int e = threadIdx.x; e < C * 64; e += 256) w2[(e >> 6) * 65 + (e & 63)] = u.l2w[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *va = strip + wave * (2 * C + 128), *vb = va + C, *vh = vb + C;           // two C-vectors and a 64-vector (+ 64 spare)
    __syncthreads();
    auto sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    for (int p = blockIdx.x * 4 + wave; p < P; p += gridDim.x * 4) {
        const uint32_t pt = (uint32_t)p;
        float x[NC], d[NC], r0[NC], r[NC], t2[NC], dz[NC], m2[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            const bool on = c < C;
            x[k] = on ? Xq[(size_t)p * C + c] : 0.0f;
            d[k] = on ? x[k] - O[(size_t)p * C + c] : 0.0f;
            dz[k] = on ? dZ[(size_t)p * C + c] : 0.0f;
            m2[k] = on ? drop_mask(dc, 1, pt, (uint32_t)c) : 0.0f;
            if (on) va[c] = d[k];
        }
        sync();
        // ---- forward, recomputed: r0 = Wt d, r = relu(r0)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            float acc = 0.0f;
            if (c < C) for (int i = 0; i < C; ++i) acc = fmaf(wt[c * PT + i], va[i], acc);
            r0[k] = acc; r[k] = fmaxf(acc, 0.0f);
            if (c < C) vb[c] = r[k];
        }
        sync();
        // ha0 = W1 r + b1 (lane = hidden row), ha = relu(ha0) m1
        float ha0 = u.l1b[lane];
        for (int i = 0; i < C; ++i) ha0 = fmaf(w1[lane * PT + i], vb[i], ha0);
        const float m1 = drop_mask(dc, 0, pt, (uint32_t)lane);
        const float ha = fmaxf(ha0, 0.0f) * m1;
        vh[lane] = ha;
        sync();
        // y = W2 ha + b2; t2 = y m2 + r; LayerNorm statistics over the C channels
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            float y = c < C ? u.l2b[c] : 0.0f;
            if (c < C) for (int i = 0; i < 64; ++i) y = fmaf(w2[c * 65 + i], vh[i], y);
            t2[k] = c < C ? fmaf(y, m2[k], r[k]) : 0.0f;
            sum += t2[k];
        }
        const float mean = wave_sum(sum) * (1.0f / (float)C);
        float var = 0.0f;
#pragma unroll
        for (int k = 0; k < NC; ++k) { const float cc = (lane + 64 * k < C) ? t2[k] - mean : 0.0f; var = fmaf(cc, cc, var); }
        const float rstd = 1.0f / sqrtf(wave_sum(var) * (1.0f / (float)C) + 1e-5f);
        // ---- backward
        float xh[NC], dxh[NC], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            xh[k] = c < C ? (t2[k] - mean) * rstd : 0.0f;
            dxh[k] = c < C ? dz[k] * u.lnw[c] : 0.0f;
            s1 += dxh[k]; s2 = fmaf(dxh[k], xh[k], s2);
        }
        s1 = wave_sum(s1) * (1.0f / (float)C); s2 = wave_sum(s2) * (1.0f / (float)C);
        float dt2[NC], dY[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            dt2[k] = c < C ? rstd * (dxh[k] - s1 - xh[k] * s2) : 0.0f;
            dY[k] = dt2[k] * m2[k];
            if (c < C) va[c] = dY[k];
        }
        sync();
        // dha = W2^T dY (lane = hidden column), masked by relu'(ha0) and m1
        float dha = 0.0f;
        for (int i = 0; i < C; ++i) dha = fmaf(w2[i * 65 + lane], va[i], dha);
        dha = ha0 > 0.0f ? dha * m1 : 0.0f;
        sync();
        vh[lane] = dha;
        sync();
        // dr = dt2 + W1^T dha; dr0 = relu'(r0) dr
        float dr0[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            float acc = dt2[k];
            if (c < C) for (int i = 0; i < 64; ++i) acc = fmaf(w1[i * PT + c], vh[i], acc);
            dr0[k] = (c < C && r0[k] > 0.0f) ? acc : 0.0f;
            if (c < C) vb[c] = dr0[k];
        }
        sync();
        // dd = Wt^T dr0
        const float li = linv[p];
        const size_t aC = (size_t)p * C, a64 = (size_t)p * 64;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            if (c >= C) continue;
            float dd = 0.0f;
            for (int i = 0; i < C; ++i) dd = fmaf(wt[i * PT + c], vb[i], dd);
            w.gXq[aC + c] = dz[k] + dd; w.dO[aC + c] = -dd; w.dOs[aC + c] = -dd * li;
            w.D[aC + c] = d[k]; w.dR0[aC + c] = dr0[k]; w.R[aC + c] = r[k];
            w.dY[aC + c] = dY[k]; w.dy[aC + c] = dz[k]; w.dyxh[aC + c] = dz[k] * xh[k];
        }
        w.dH0[a64 + lane] = dha; w.Hd[a64 + lane] = ha;
        sync();
    }
}

// dQ = M1 - u M2 (the row term of the softmax backward, after the slices' M1 and u have been added up)
__global__ void fbw_dq_combine_kernel(float *dQ, const float *M2, const float *u, size_t P) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P * 64; i += (size_t)gridDim.x * blockDim.x) dQ[i] -= u[i >> 6] * M2[i];
}

// backward of the l2-normalised projections at width C: dPq = (dQ - Q (Q.dQ)) / |Pq| (written over dQ), likewise dPk;
// gXq += dPq WQ; gXk = dPk WK + dV WV.  One wave per point, the three weight matrices in LDS ([row][C]: a column per lane)
template <int C>
__global__ void __launch_bounds__(256)
fbw_proj_bwd_kernel(const float *Qf, const float *Kf, const float *nq, const float *nk, float *dQ, float *dK, const float *dV,
                    FusionUnitDev u, float *gXq, float *gXk, int P) {
    constexpr int NC = (C + 63) / 64;
    extern __shared__ float fbw_lds[];
    float *wq = fbw_lds, *wk = wq + 64 * C, *wv = wk + 64 * C, *strip = wv + C * C;
    for (int e = threadIdx.x; e < 64 * C; e += 256) { wq[e] = u.WQ[e]; wk[e] = u.WK[e]; }
    for (int e = threadIdx.x; e < C * C; e += 256) wv[e] = u.WV[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *sq = strip + wave * (128 + C), *sk = sq + 64, *sv = sk + 64;
    __syncthreads();
    for (int p = blockIdx.x * 4 + wave; p < P; p += gridDim.x * 4) {
        const float q = Qf[(size_t)p * 64 + lane], k = Kf[(size_t)p * 64 + lane];
        const float dq = dQ[(size_t)p * 64 + lane], dk = dK[(size_t)p * 64 + lane];
        const float dpq = (dq - q * wave_sum(q * dq)) / nq[p], dpk = (dk - k * wave_sum(k * dk)) / nk[p];
        dQ[(size_t)p * 64 + lane] = dpq;
        dK[(size_t)p * 64 + lane] = dpk;
        sq[lane] = dpq; sk[lane] = dpk;
        for (int c = lane; c < C; c += 64) sv[c] = dV[(size_t)p * C + c];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < NC; ++kk) {
            const int c = lane + 64 * kk;
            if (c >= C) continue;
            float a = 0.0f, b = 0.0f;
            for (int dd = 0; dd < 64; ++dd) { a = fmaf(sq[dd], wq[dd * C + c], a); b = fmaf(sk[dd], wk[dd * C + c], b); }
            for (int dd = 0; dd < C; ++dd) b = fmaf(sv[dd], wv[dd * C + c], b);
            gXq[(size_t)p * C + c] += a;
            gXk[(size_t)p * C + c] = b;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// column sums of G [P][M] over fixed 512-row chunks, then over the chunks in order (the LayerNorm's affine gradients)
__global__ void __launch_bounds__(256) fbw_colsum_kernel(const float *G, int M, int P, float *partials) {
    const int pbeg = blockIdx.x * WCHUNK, pend = min(pbeg + WCHUNK, P);
    for (int m = threadIdx.x; m < M; m += 256) {
        float acc = 0.0f;
        for (int p = pbeg; p < pend; ++p) acc += G[(size_t)p * M + m];
        partials[(size_t)blockIdx.x * M + m] = acc;
    }
}
__global__ void __launch_bounds__(256) fbw_colsum_reduce_kernel(const float *partials, int M, int nchunk, float *dst, int accumulate) {
    for (int m = threadIdx.x; m < M; m += 256) {
        float sum = 0.0f;
        for (int c = 0; c < nchunk; ++c) sum += partials[(size_t)c * M + m];
        dst[m] = accumulate ? dst[m] + sum : sum;
    }
}
__global__ void fbw_axpy_kernel(float *dst, const float *src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

struct BwdWsW {
    float *Qf, *Kf, *nq, *nk, *dZ, *gXq, *gXk, *dO, *dOs, *D, *dR0, *R, *dH0, *Hd, *dY, *dy, *dyxh, *dV, *tp, *u, *dQ, *dK, *M2;
    float *dM, *dT, *wtmp, *colp;
    void *rows_ws;
    size_t rows_ws_bytes;
};
size_t bwd_layout_wide(int B, int N, int C, BwdWsW *ws, char *base) {
    const size_t P = (size_t)B * N, nchunk = (P + WCHUNK - 1) / WCHUNK;
    size_t off = 0;
    auto take = [&](size_t floats) { float *p = base ? (float *)(base + off) : nullptr; off += (floats * 4 + 255) / 256 * 256; return p; };
    BwdWsW w;
    w.Qf = take(P * 64); w.Kf = take(P * 64); w.nq = take(P); w.nk = take(P);
    w.dZ = take(P * C); w.gXq = take(P * C); w.gXk = take(P * C); w.dO = take(P * C); w.dOs = take(P * C);
    w.D = take(P * C); w.dR0 = take(P * C); w.R = take(P * C); w.dH0 = take(P * 64); w.Hd = take(P * 64);
    w.dY = take(P * C); w.dy = take(P * C); w.dyxh = take(P * C);
    w.dV = take(P * C); w.tp = take(P); w.u = take(P); w.dQ = take(P * 64); w.dK = take(P * 64); w.M2 = take(P * 64);
    w.dM = take(P * C); w.dT = take(P * C);
    w.wtmp = take((size_t)C * C + 64);                              // a weight gradient of the second use of the shared unit, before it is added
    w.colp = take(nchunk * (size_t)C);
    w.rows_ws_bytes = vt_rows_wgrad_workspace_bytes((int64_t)P, C, C);
    w.rows_ws = take(w.rows_ws_bytes / 4 + 64);
    if (ws) *ws = w;
    return off;
}

template <int C>
int unit_bwd_wide(int call, const float *Xq, const float *Xk, const float *dOut, const FusionUnitDev &u, const vt_fusion_unit_grads &g,
                  int accumulate, const FusionSaved &sv, const BwdWsW &w, float *dXq, float *dXk, int B, int N, float p_drop,
                  unsigned long long seed, hipStream_t s) {
    const int P = B * N, nrb = (N + NFIX - 1) / NFIX, cus = vt_num_cus();
    const dim3 tg((unsigned)nrb * (unsigned)B);
    const int pgrid = (P + 3) / 4 < 2 * cus ? (P + 3) / 4 : 2 * cus;
    const size_t lds_proj = (size_t)(2 * C * 64 + 4 * 2 * C) * 4, lds_epi = (size_t)(C * (C + 1) + 64 * (C + 1) + C * 65 + 4 * (2 * C + 128)) * 4,
                 lds_pb = (size_t)(2 * 64 * C + C * C + 4 * (128 + C)) * 4;
    hipLaunchKernelGGL(fbw_proj_kernel<C>, dim3(pgrid), dim3(256), lds_proj, s, Xq, Xk, u, w.Qf, w.Kf, w.nq, w.nk, P);
    hipLaunchKernelGGL(fb_inorm_bwd_kernel, dim3(B, C / 32), dim3(1024), 0, s, (const float *)sv.Z[call], dOut, w.dZ, N, C);
    const EpiOut eo{w.gXq, w.dO, w.dOs, w.D, w.dR0, w.R, w.dH0, w.Hd, w.dY, w.dy, w.dyxh};
    hipLaunchKernelGGL(fbw_epilogue_bwd_kernel<C>, dim3(pgrid < cus ? pgrid : cus), dim3(256), lds_epi, s, Xq, (const float *)sv.O[call], (const float *)w.dZ,
                       (const float *)sv.linv[call], u, eo, P, drop_cfg(p_drop, seed, (uint32_t)call, 8));
    for (int sl = 0; sl < C / 32; ++sl)
        hipLaunchKernelGGL(fb_dv_kernel, tg, dim3(NT), 0, s, (const float *)w.Qf, (const float *)w.Kf, (const float *)w.dOs,
                           (const float *)sv.V[call], (const float *)sv.s[call], w.dV, w.tp, N, nrb, B, C, 32 * sl, sl > 0);
    for (int sl = 0; sl < C / 32; ++sl)
        hipLaunchKernelGGL(fb_dq_kernel, tg, dim3(NT), 0, s, (const float *)w.Qf, (const float *)w.Kf, (const float *)w.dO,
                           (const float *)sv.V[call], (const float *)sv.s[call], (const float *)w.tp, (const float *)sv.linv[call], w.dQ, w.u, N, nrb, B,
                           C, 32 * sl, sl == 0, w.M2);
    {
        const size_t n = (size_t)P * 64;
        hipLaunchKernelGGL(fbw_dq_combine_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, w.dQ,
                           (const float *)w.M2, (const float *)w.u, (size_t)P);
    }
    for (int sl = 0; sl < C / 32; ++sl)
        hipLaunchKernelGGL(fb_dk_kernel, tg, dim3(NT), 0, s, (const float *)w.Qf, (const float *)w.Kf, (const float *)w.dO,
                           (const float *)sv.V[call], (const float *)sv.s[call], (const float *)w.tp, (const float *)sv.linv[call],
                           (const float *)w.u, w.dK, N, nrb, B, C, 32 * sl, sl == 0);
    hipLaunchKernelGGL(fbw_proj_bwd_kernel<C>, dim3(pgrid < cus ? pgrid : cus), dim3(256), lds_pb, s, (const float *)w.Qf, (const float *)w.Kf,
                       (const float *)w.nq, (const float *)w.nk, w.dQ, w.dK, (const float *)w.dV, u, w.gXq, w.gXk, P);
    // weight gradients: dW [M][K] = sum_p G[p][m] X[p][k] (+ db = column sums of G)
    auto wgrad = [&](const float *G, int M, const float *X, int K, float *dW, float *db) -> int {
        float *dst = accumulate ? w.wtmp : dW;
        float *dbt = db ? (accumulate ? w.wtmp + (size_t)M * K : db) : nullptr;
        const int rc = vt_rows_wgrad(G, M, X, K, nullptr, 0, 0, (int64_t)P, w.rows_ws, w.rows_ws_bytes, dst, dbt, s);
        if (rc) return rc;
        if (accumulate) {
            hipLaunchKernelGGL(fbw_axpy_kernel, dim3(64), dim3(256), 0, s, dW, (const float *)dst, (size_t)M * K);
            if (db) hipLaunchKernelGGL(fbw_axpy_kernel, dim3(1), dim3(256), 0, s, db, (const float *)dbt, (size_t)M);
        }
        return 0;
    };
    int rc;
    if ((rc = wgrad(w.dQ, 64, Xq, C, g.WQ, nullptr))) return rc;
    if ((rc = wgrad(w.dK, 64, Xk, C, g.WK, nullptr))) return rc;
    if ((rc = wgrad(w.dV, C, Xk, C, g.WV, nullptr))) return rc;
    if ((rc = wgrad(w.dR0, C, w.D, C, g.trans_conv, nullptr))) return rc;
    if ((rc = wgrad(w.dH0, 64, w.R, C, g.linear1_w, g.linear1_b))) return rc;
    if ((rc = wgrad(w.dY, C, w.Hd, 64, g.linear2_w, g.linear2_b))) return rc;
    const int nchunk = (P + WCHUNK - 1) / WCHUNK;
    hipLaunchKernelGGL(fbw_colsum_kernel, dim3(nchunk), dim3(256), 0, s, (const float *)w.dyxh, C, P, w.colp);
    hipLaunchKernelGGL(fbw_colsum_reduce_kernel, dim3(1), dim3(256), 0, s, (const float *)w.colp, C, nchunk, g.norm2_w, accumulate);
    hipLaunchKernelGGL(fbw_colsum_kernel, dim3(nchunk), dim3(256), 0, s, (const float *)w.dy, C, P, w.colp);
    hipLaunchKernelGGL(fbw_colsum_reduce_kernel, dim3(1), dim3(256), 0, s, (const float *)w.colp, C, nchunk, g.norm2_b, accumulate);
    const size_t n = (size_t)P * C;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dXk == nullptr) {
        hipLaunchKernelGGL(fb_add_kernel, dim3(blocks), dim3(256), 0, s, (const float *)w.gXq, (const float *)w.gXk, dXq, n);
    } else {
        hipLaunchKernelGGL(fb_add_kernel, dim3(blocks), dim3(256), 0, s, (const float *)w.gXq, (const float *)nullptr, dXq, n);
        hipLaunchKernelGGL(fb_add_kernel, dim3(blocks), dim3(256), 0, s, (const float *)w.gXk, (const float *)nullptr, dXk, n);
    }
    return 0;
}

template <int C>
int fusion_bwd_wide(const float *d_out, const float *c_img, const float *c, int B, int N, const vt_fusion_params *p, float p_drop,
                    unsigned long long seed, const void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                    float *d_c_img, float *d_c, const vt_fusion_grads *grads, hipStream_t s) {
    FusionSaved sv;
    BwdWsW w;
    if (saved_bytes < fusion_saved_layout(B, N, &sv, (char *)saved, C)) return vt_fail(VT_ERR_WORKSPACE, "vt_fusion_bwd: saved-state buffer too small");
    if (workspace_bytes < bwd_layout_wide(B, N, C, &w, (char *)workspace)) return vt_fail(VT_ERR_WORKSPACE, "vt_fusion_bwd: workspace too small");
    hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&fbw_proj_kernel<C>), 160 * 1024);
    if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&fbw_epilogue_bwd_kernel<C>), 160 * 1024);
    if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&fbw_proj_bwd_kernel<C>), 160 * 1024);
    if (e != hipSuccess) return vt_check(e, "vt_fusion_bwd: hipFuncSetAttribute");
    const FusionUnitDev us = unit_of(p->self_attn), ux = unit_of(p->cross_attn);
    int rc;
    if ((rc = unit_bwd_wide<C>(2, sv.T, sv.M, d_out, ux, grads->cross_attn, 0, sv, w, w.dT, w.dM, B, N, p_drop, seed, s))) return rc;
    if ((rc = unit_bwd_wide<C>(1, c_img, c_img, w.dT, us, grads->self_attn, 0, sv, w, d_c_img, nullptr, B, N, p_drop, seed, s))) return rc;
    if ((rc = unit_bwd_wide<C>(0, c, c, w.dM, us, grads->self_attn, 1, sv, w, d_c, nullptr, B, N, p_drop, seed, s))) return rc;
    return vt_check(hipGetLastError(), "vt_fusion_bwd");
}

}  // namespace

extern "C" {

size_t vt_fusion_bwd_workspace_bytes_wide(int B, int N, int d_model) {
    if (B <= 0 || N <= 0 || d_model <= 0 || (d_model & 31) || d_model > 128) return 0;
    return d_model == 32 ? bwd_layout(B, N, nullptr, nullptr) : bwd_layout_wide(B, N, d_model, nullptr, nullptr);
}

size_t vt_fusion_bwd_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return bwd_layout(B, N, nullptr, nullptr);
}

int vt_fusion_bwd(const float *d_out, const float *c_img, const float *c, int B, int N, const vt_fusion_params *p, float p_drop,
                  unsigned long long seed, const void *saved, size_t saved_bytes, void *workspace, size_t workspace_bytes,
                  float *d_c_img, float *d_c, const vt_fusion_grads *grads, void *stream) {
    if (!d_out || !c_img || !c || !p || !saved || !workspace || !d_c_img || !d_c || !grads)
        return vt_fail(VT_ERR_INVALID, "vt_fusion_bwd: null argument");
    if (B <= 0 || N <= 0) return vt_fail(VT_ERR_INVALID, "vt_fusion_bwd: bad size");
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return vt_fail(VT_ERR_INVALID, "vt_fusion_bwd: p_drop must be in [0, 1)");
    const float *const *gp = reinterpret_cast<const float *const *>(grads);
    for (size_t i = 0; i < sizeof(vt_fusion_grads) / sizeof(float *); ++i)
        if (!gp[i]) return vt_fail(VT_ERR_INVALID, "vt_fusion_bwd: null gradient buffer");
    if (p->key_dim == 64 && p->d_model == 64) return fusion_bwd_wide<64>(d_out, c_img, c, B, N, p, p_drop, seed, saved, saved_bytes, workspace, workspace_bytes, d_c_img, d_c, grads, (hipStream_t)stream);
    if (p->key_dim == 64 && p->d_model == 96) return fusion_bwd_wide<96>(d_out, c_img, c, B, N, p, p_drop, seed, saved, saved_bytes, workspace, workspace_bytes, d_c_img, d_c, grads, (hipStream_t)stream);
    if (p->key_dim == 64 && p->d_model == 128) return fusion_bwd_wide<128>(d_out, c_img, c, B, N, p, p_drop, seed, saved, saved_bytes, workspace, workspace_bytes, d_c_img, d_c, grads, (hipStream_t)stream);
    if (p->d_model != 32 || p->key_dim != 64) return vt_fail(VT_ERR_UNSUPPORTED, "vt_fusion_bwd: d_model in {32, 64, 96, 128} with key_feature_dim = 64");
    FusionSaved sv;
    BwdWs w;
    if (saved_bytes < fusion_saved_layout(B, N, &sv, (char *)saved)) return vt_fail(VT_ERR_WORKSPACE, "vt_fusion_bwd: saved-state buffer too small");
    if (workspace_bytes < bwd_layout(B, N, &w, (char *)workspace)) return vt_fail(VT_ERR_WORKSPACE, "vt_fusion_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const FusionUnitDev us = unit_of(p->self_attn), ux = unit_of(p->cross_attn);
    hipLaunchKernelGGL(fusion_pack_kernel, dim3(6), dim3(1024), 0, s, us, w.blob_f[0]);
    hipLaunchKernelGGL(fusion_pack_kernel, dim3(6), dim3(1024), 0, s, ux, w.blob_f[1]);
    hipLaunchKernelGGL(fusion_pack_t_kernel, dim3(5), dim3(1024), 0, s, us, w.blob_t[0]);
    hipLaunchKernelGGL(fusion_pack_t_kernel, dim3(5), dim3(1024), 0, s, ux, w.blob_t[1]);
    // out = cross(tgt, mem); tgt = self(c_img); mem = self(c): the shared self-attention's gradients are the SUM of its two uses
    unit_bwd(2, sv.T, sv.M, d_out, ux, grads->cross_attn, 0, w.blob_f[1], w.blob_t[1], sv, w, w.dT, w.dM, B, N, p_drop, seed, s);
    unit_bwd(1, c_img, c_img, w.dT, us, grads->self_attn, 0, w.blob_f[0], w.blob_t[0], sv, w, d_c_img, nullptr, B, N, p_drop, seed, s);
    unit_bwd(0, c, c, w.dM, us, grads->self_attn, 1, w.blob_f[0], w.blob_t[0], sv, w, d_c, nullptr, B, N, p_drop, seed, s);
    return vt_check(hipGetLastError(), "vt_fusion_bwd");
}

int vt_fusion_dropout_mask(float p_drop, unsigned long long seed, int call, int which, int points, float *mask, void *stream) {
    if (!mask || points <= 0 || call < 0 || call > 2 || which < 0 || which > 1) return vt_fail(VT_ERR_INVALID, "vt_fusion_dropout_mask: bad argument");
    const int width = which == 0 ? 64 : 32;
    const size_t n = (size_t)points * width;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(fb_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, drop_cfg(p_drop, seed, (uint32_t)call),
                       (uint32_t)which, width, mask, n);
    return vt_check(hipGetLastError(), "vt_fusion_dropout_mask");
}

int vt_fusion_dropout_mask_wide(float p_drop, unsigned long long seed, int call, int which, int points, int d_model, float *mask, void *stream) {
    if (!mask || points <= 0 || call < 0 || call > 2 || which < 0 || which > 1 || d_model <= 32 || (d_model & 31) || d_model > 128)
        return vt_fail(VT_ERR_INVALID, "vt_fusion_dropout_mask_wide: bad argument");
    const int width = which == 0 ? 64 : d_model;
    const size_t n = (size_t)points * width;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(fb_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, drop_cfg(p_drop, seed, (uint32_t)call, 8),
                       (uint32_t)which, width, mask, n);
    return vt_check(hipGetLastError(), "vt_fusion_dropout_mask_wide");
}

}  // extern "C"
