// LocalDecoder.forward / forward_img for the shapes the shipped kernels (decode.hip: hidden = c_dim = 32, relu) do not cover:
// hidden_size and c_dim any multiples of 32 up to 256 (the class defaults of the reference are 256 / 128,
// src/conv_onet/models/decoder.py:24-51), n_blocks up to VT_MAX_BLOCKS, and `leaky`: leaky_relu(0.2) in front of the output heads
// (decoder.py:46-49, 157; the ResnetBlockFC activations are ReLU whatever `leaky` says, layers.py:33).  Exact f32:
// v_mfma_f32_32x32x2_f32 with f32 operands, so the result is the f32 network's up to summation order.  Inference
// (vt_decode_fwd_wide) and training: vt_decode_fwd_wide_train saves the layer inputs point-major, vt_decode_bwd_wide walks the
// network back on transposed weight fragments (same streaming scheme) and leaves the per-layer output gradients for the weight
// gradients, which are vt_rows_wgrad's tall-skinny products over those two sets of rows (pointnet.hip).
//
// A workgroup of eight waves owns 32 query points.  The weights do not fit LDS at these widths (256/128/5: 3.3 MB), so they stream
// from L2 in fragment order (one coalesced 16-byte load per lane = four k-steps of one 32-row block) and the ACTIVATIONS live in
// LDS: the sampled features c [c_dim][32 points] and two [hidden][32] buffers the layers ping-pong through; the residual
// stream `net` never leaves the registers of the wave that owns its 32-row block (wave w: rows 32 w .. 32 w + 31; with four waves
// and two blocks each a 256-wide decoder kept one wave per SIMD: 55.7 ms per 128^3 against 36.6 with eight; widths up to 128
// run four waves: eight, half of them idle, took 7.1 ms at 64 / 32 against 5.3).
//   per point: 2 * (p_in + n_blocks * (c_dim + 2 hidden) * hidden) flop; at 256/128/5 the f32 matrix pipe bounds a 128^3 lattice
//   at ~22 ms (157 TFLOP/s), the weight stream at 3.3 MB per 32 points from L2 at about the same.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vt_common.h"
#include "decode_common.h"

namespace {

constexpr int WIDE_PTS = 32, WIDE_PITCH = 33;                           // pitch 33: the channel-major writes of a wave hit 32 banks
constexpr int WIDE_MAX = 256;

struct WideArgs {
    DecodeArgs d;               // grid, pts / lattice, c_img, out, out2, N, total, R, divisor (blob unused)
    const float *blob;
    int H, C, nb, Kp, p_in;     // hidden, c_dim, blocks, fc_p's K padded to a multiple of 8
    int leaky;
    int nearest;                // sample_mode 'nearest': the voxel at the rounded coordinate instead of the trilinear blend
    float *save;                // training forward: the layer inputs, point-major (wide_save_layout), or null
};

// what the training forward keeps for the backward, point-major rows (the X operands of the weight gradients):
//   c [P][C] | per block i: a0_i = relu(net + fc_c_i(c)) [P][H], a1_i = relu(fc_0_i(a0_i)) [P][H] | af = actvn(net_final) [P][H]
struct WideSave { size_t c, blk, af, total; };
__host__ __device__ inline WideSave wide_save_layout(size_t P, int H, int C, int nb) {
    WideSave l;
    l.c = 0; l.blk = P * C; l.af = l.blk + (size_t)nb * 2 * P * H; l.total = l.af + P * H;
    return l;
}
// output-side gradients vt_decode_bwd_wide leaves for the weight gradients, point-major:
//   DN_i [P][H], i = 0 .. nb: gradient of the residual stream in front of block i (DN_nb: behind the last block)
//   DH_i [P][H], i = 0 .. nb - 1: gradient of fc_0_i's output
struct WideGws { size_t dn, dh, total; };
__host__ __device__ inline WideGws wide_gws_layout(size_t P, int H, int nb) {
    WideGws l;
    l.dn = 0; l.dh = (size_t)(nb + 1) * P * H; l.total = l.dh + (size_t)nb * P * H;
    return l;
}
// transposed-weight blob of the backward: per block W1^T [H][H], W0^T [H][H], Wc^T [C][H] as fragments; then fc_p_img's
// c_img columns transposed [C][H] (zeros for a decoder without tactile concat); then fc_out.w [H], fc_out_contact.w [H]
struct WideLayoutT { size_t w_blk, w_1, w_0, w_c, w_pc, heads, total; };
__host__ __device__ inline WideLayoutT wide_layout_t(int H, int C, int nb) {
    WideLayoutT l;
    l.w_1 = (size_t)H * H; l.w_0 = (size_t)H * H; l.w_c = (size_t)C * H;
    l.w_blk = 0;
    l.w_pc = (size_t)nb * (l.w_1 + l.w_0 + l.w_c);
    l.heads = l.w_pc + (size_t)C * H;
    l.total = l.heads + 2 * (size_t)H + 4;
    return l;
}

// blob layout, in floats (host and device agree through these)
struct WideLayout {
    size_t w_p, w_blk, w_c, w_0, w_1;       // fragments: fc_p at w_p; block i at w_blk + i * (w_c + w_0 + w_1): fc_c, fc_0, fc_1
    size_t bias, total;                     // biases: fc_p [H]; per block fc_c, fc_0, fc_1 [H] each; then fc_out.w [H], fc_out.b,
};                                          // fc_out2.w [H], fc_out2.b (zeros without a contact head), padded to 4 floats
__host__ __device__ inline WideLayout wide_layout(int H, int C, int nb, int Kp) {
    WideLayout l;
    l.w_p = 0;
    l.w_c = (size_t)H * C; l.w_0 = (size_t)H * H; l.w_1 = (size_t)H * H;
    l.w_blk = (size_t)H * Kp;
    l.bias = l.w_blk + (size_t)nb * (l.w_c + l.w_0 + l.w_1);
    l.total = l.bias + (size_t)H * (1 + 3 * nb) + 2 * (size_t)H + 4;
    return l;
}

// W [H][K] (row stride `ld`, columns >= `kin` read as zero) -> fragments [H/32][K/8][64 lanes][4]: lane (row r = l & 31, kg = l >> 5),
// element e = W[32 ob + r][8 kq + 2 e + kg] -- the A operands of four consecutive 32x32x2 k-steps
// (rs, ks): strides of the source in rows and columns -- (ld, 1) for W itself, (1, ld) for its transpose
__global__ void wide_pack_kernel(const float *w, int H, int K, int kin, size_t rs, size_t ks, float *dst) {
    const size_t total = (size_t)H * K;
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(f & 3), l = (int)((f >> 2) & 63);
        const size_t q = f >> 8;
        const int kq = (int)(q % (K / 8)), ob = (int)(q / (K / 8));
        const int row = 32 * ob + (l & 31), k = 8 * kq + 2 * e + (l >> 5);
        dst[f] = (w && k < kin) ? w[(size_t)row * rs + (size_t)k * ks] : 0.0f;
    }
}
__global__ void wide_copy_kernel(const float *src, float *dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] : 0.0f;
}

__device__ __forceinline__ float actvn(float x, int leaky) { return x > 0.0f ? x : (leaky ? 0.2f * x : 0.0f); }

// column k3 of point g's tactile feature: the dense tensor [total][C], or the finger's row of the [F][C] table by id (255: none --
// the form the 256^3 configuration needs: one byte per point instead of 4 C)
__device__ __forceinline__ float wide_cimg(const DecodeArgs &d, uint32_t g, int k3, int C) {
    if (d.cimg_ids) {
        const unsigned id = d.cimg_ids[g];
        return id >= d.cimg_nf ? 0.0f : d.cimg_table[(size_t)id * C + k3];
    }
    return d.c_img[(size_t)g * C + k3];
}

// acc += W[32 rows of block ob][K] . X[K][32 points]; wf: the layer's fragments, x: LDS [K][WIDE_PITCH]
__device__ __forceinline__ f32x16 wide_gemm(f32x16 acc, const float *wf, int ob, int K, const float *x, int lane) {
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(wf) + (size_t)ob * (K / 8) * 64 + lane;
    const float *xb = x + (lane >> 5) * WIDE_PITCH + (lane & 31);

    for (int kq = 0; kq < K / 8; ++kq) {
        const f32x4 a = w4[(size_t)kq * 64];
        const float *xr = xb + kq * 8 * WIDE_PITCH;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xr[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xr[2 * WIDE_PITCH], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xr[4 * WIDE_PITCH], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xr[6 * WIDE_PITCH], acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ f32x16 bias16(const float *b, int ob, int kg) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = b[32 * ob + chan_of(i, kg)];
    return r;
}

template <int WIDE_WAVES>                                           // 4 for hidden <= 128, 8 above: one 32-row block per wave at most
__global__ void __launch_bounds__(WIDE_WAVES * 64)
decode_wide_kernel(WideArgs a) {
    constexpr int WIDE_THREADS = WIDE_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float wl[];      // c [C][33] | buffer A [max(H, Kp)][33] | buffer B [H][33] | heads [waves][2 lane halves][2 heads][32]
    const DecodeArgs &d = a.d;
    const int H = a.H, C = a.C, nh = H / 32, Kp = a.Kp;
    const int rowsA = H > Kp ? H : Kp;
    float *cl = wl, *bufA = cl + (size_t)C * WIDE_PITCH, *bufB = bufA + (size_t)rowsA * WIDE_PITCH, *heads = bufB + (size_t)H * WIDE_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kg = lane >> 5;
    const WideLayout lay = wide_layout(H, C, a.nb, Kp);
    const float *bias = a.blob + lay.bias;
    const uint32_t ntiles = (d.total + WIDE_PTS - 1) / WIDE_PTS;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- the tile's inputs: sampled features c (decoder.py:62-68), fc_p's input rows [p | c_img | 0] ----
        {
            constexpr int PG = WIDE_THREADS / 32;
            const int ch = tid & 31, pg = tid >> 5;                 // thread = (channel lane, point group): points pg, pg + PG, ...
#pragma unroll
            for (int i = 0; i < WIDE_PTS / PG; ++i) {
                const int pt = pg + PG * i;
                uint32_t g = tile * WIDE_PTS + pt;
                if (g >= d.total) g = d.total - 1u;
                const uint32_t b = g / d.N;
                float px, py, pz;
                point_of(d, g, g - b * d.N, px, py, pz);
                const Tri t = tri_setup(px, py, pz, d.divisor, d.R);
                const float *gb = d.grid + (size_t)b * d.R * d.R * d.R * C;
                // F.grid_sample(mode='nearest', padding_mode='border', align_corners=True): nearbyint (half to even) of the clipped coordinate
                const size_t near = a.nearest ? (((size_t)__builtin_rintf(grid_coord(pz, d.divisor, d.R)) * d.R + (size_t)__builtin_rintf(grid_coord(py, d.divisor, d.R))) * d.R +
                                                 (size_t)__builtin_rintf(grid_coord(px, d.divisor, d.R))) * C : 0;
                for (int cb = 0; cb < C; cb += 32) {
                    if (d.c_direct) {                               // the conditioning features given directly (the MLP behind a fuser)
                        cl[(cb + ch) * WIDE_PITCH + pt] = d.c_direct[(size_t)g * C + cb + ch];
                        continue;
                    }
                    if (a.nearest) {
                        const float v = gb[near + cb + ch];
                        cl[(cb + ch) * WIDE_PITCH + pt] = v;
                        if (a.save && tile * WIDE_PTS + pt < d.total) a.save[(size_t)g * C + cb + ch] = v;
                        continue;
                    }
                    float acc = 0.0f;
#pragma unroll
                    for (int dz = 0; dz < 2; ++dz) {
                        const int zz = dz ? t.z1 : t.z0;
                        const float wz = dz ? t.wz1 : t.wz0;
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy) {
                            const int yy = dy ? t.y1 : t.y0;
                            const float wy = dy ? t.wy1 : t.wy0;
                            const size_t row = ((size_t)zz * d.R + yy) * d.R;
                            acc = fmaf(gb[(row + t.x0) * C + cb + ch], (t.wx0 * wy) * wz, acc);
                            acc = fmaf(gb[(row + t.x1) * C + cb + ch], (t.wx1 * wy) * wz, acc);
                        }
                    }
                    cl[(cb + ch) * WIDE_PITCH + pt] = acc;
                    if (a.save && tile * WIDE_PTS + pt < d.total) a.save[(size_t)g * C + cb + ch] = acc;
                }
                if (ch < 3) bufA[ch * WIDE_PITCH + pt] = ch == 0 ? px : (ch == 1 ? py : pz);
                for (int k = 3 + ch; k < Kp; k += 32)
                    bufA[k * WIDE_PITCH + pt] = (a.p_in > 3 && k < a.p_in) ? wide_cimg(d, g, k - 3, a.p_in - 3) : 0.0f;
            }
        }
        __syncthreads();
        // ---- fc_p (decoder.py:139 / 81) ----
        f32x16 net;
        const int ob = wave;                                        // nh <= WIDE_WAVES; waves beyond the width only keep the barriers
        // training forward: this lane's point row in the point-major save slots
        const uint32_t gj = tile * WIDE_PTS + (uint32_t)j;
        const bool keep = a.save != nullptr && gj < d.total && ob < nh;
        const WideSave sv = wide_save_layout(d.total, H, C, a.nb);
        if (ob < nh) net = wide_gemm(bias16(bias, ob, kg), a.blob + lay.w_p, ob, Kp, bufA, lane);
        __syncthreads();                                            // bufA is free again
        // ---- n_blocks x (fc_c add, ResnetBlockFC: layers.py:41-50; its activations are ReLU) ----
        for (int blk = 0; blk < a.nb; ++blk) {
            const float *wb = a.blob + lay.w_blk + (size_t)blk * (lay.w_c + lay.w_0 + lay.w_1);
            const float *bb = bias + (size_t)H * (1 + 3 * blk);
            if (ob < nh) {
                const f32x16 bc = bias16(bb, ob, kg);
#pragma unroll
                for (int i = 0; i < 16; ++i) net[i] += bc[i];
                net = wide_gemm(net, wb, ob, C, cl, lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) bufA[(32 * ob + chan_of(i, kg)) * WIDE_PITCH + j] = actvn(net[i], 0);
                if (keep) store_acc16(a.save + sv.blk + ((size_t)(2 * blk) * d.total + gj) * H + 32 * ob, relu16(net), kg);
            }
            __syncthreads();
            if (ob < nh) {
                const f32x16 hid = wide_gemm(bias16(bb + H, ob, kg), wb + lay.w_c, ob, H, bufA, lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) bufB[(32 * ob + chan_of(i, kg)) * WIDE_PITCH + j] = actvn(hid[i], 0);
                if (keep) store_acc16(a.save + sv.blk + ((size_t)(2 * blk + 1) * d.total + gj) * H + 32 * ob, relu16(hid), kg);
            }
            __syncthreads();
            if (ob < nh) {
                const f32x16 b1 = bias16(bb + 2 * H, ob, kg);
#pragma unroll
                for (int i = 0; i < 16; ++i) net[i] += b1[i];
                net = wide_gemm(net, wb + lay.w_c + lay.w_0, ob, H, bufB, lane);
            }
        }
        // ---- fc_out / fc_out_contact on actvn(net) (decoder.py:157-158, 128-131) ----
        const float *ow = bias + (size_t)H * (1 + 3 * a.nb), *ow2 = ow + H + 1;
        float o1 = 0.0f, o2 = 0.0f;
        if (ob < nh) {
            f32x16 af;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * ob + chan_of(i, kg);
                const float v = actvn(net[i], a.leaky);
                af[i] = v;
                o1 = fmaf(ow[row], v, o1);
                o2 = fmaf(ow2[row], v, o2);
            }
            if (keep) store_acc16(a.save + sv.af + (size_t)gj * H + 32 * ob, af, kg);
        }
        heads[((wave * 2 + kg) * 2 + 0) * 32 + j] = o1;
        heads[((wave * 2 + kg) * 2 + 1) * 32 + j] = o2;
        __syncthreads();
        if (tid < 64) {
            const int pt = tid & 31, which = tid >> 5;
            float o = which ? ow2[H] : ow[H];
#pragma unroll
            for (int w = 0; w < 2 * WIDE_WAVES; ++w) o += heads[(w * 2 + which) * 32 + pt];     // waves and lane halves in a fixed order
            const uint32_t g = tile * WIDE_PTS + pt;
            float *dst = which ? d.out2 : d.out;
            if (g < d.total && dst) dst[g] = o;
        }
        __syncthreads();
    }
}


// ---- backward: data gradients ----------------------------------------------------------------------------------------------
// With N_i the residual stream in front of block i, mid_i = N_i + fc_c_i(c), a0 = relu(mid_i), hid = fc_0(a0), a1 = relu(hid),
// N_{i+1} = mid_i + fc_1(a1):      d a1 = W1^T dN_{i+1},  dH_i = d a1 . [a1 > 0],  dN_i = dN_{i+1} + (W0^T dH_i) . [a0 > 0],
// d c += Wc_i^T dN_i,  and at the front  d c_img = Wp[:, 3:]^T dN_0.  The wave that owns rows 32 ob .. 32 ob + 31 keeps its
// slice of dN in registers through the whole walk; the full dN / dH of the 32 points pass through two LDS buffers, the B
// operands of the transposed-weight GEMMs (two barriers per block).  dN_i / dH_i also go to the point-major workspace the
// weight gradients read; d c is scattered to the channels-last grid gradient with f32 atomics (the 8 trilinear corners, or
// the one voxel of 'nearest').
struct WideBwdArgs {
    DecodeArgs d;               // pts, N, total, R, divisor; grid / out unused
    const float *blob_t, *save, *grad_out, *grad_out2;
    float *gws, *grad_grid, *grad_cimg;
    int H, C, nb, leaky, nearest;
    float *grad_c;              // d c per point [total][C] instead of the scatter (the MLP on given features)
};

template <int WIDE_WAVES>
__global__ void __launch_bounds__(WIDE_WAVES * 64)
decode_wide_bwd_kernel(WideBwdArgs a) {
    constexpr int WIDE_THREADS = WIDE_WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float wl[];      // bufG [H][33] | bufH [H][33] | d c [C][33]
    const DecodeArgs &d = a.d;
    const int H = a.H, C = a.C, nh = H / 32, nc = C / 32;
    float *bufG = wl, *bufH = bufG + (size_t)H * WIDE_PITCH, *cl = bufH + (size_t)H * WIDE_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kg = lane >> 5;
    const int ob = wave;
    const WideLayoutT lay = wide_layout_t(H, C, a.nb);
    const WideSave sv = wide_save_layout(d.total, H, C, a.nb);
    const WideGws gw = wide_gws_layout(d.total, H, a.nb);
    const float *ow = a.blob_t + lay.heads, *ow2 = ow + H;
    const uint32_t ntiles = (d.total + WIDE_PTS - 1) / WIDE_PTS;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t gj = tile * WIDE_PTS + (uint32_t)j;
        const bool live = gj < d.total;
        const size_t gr = live ? gj : d.total - 1u;                  // row to read for a lane past the end (its results are dropped)
        auto put = [&](float *buf, const f32x16 &v) {
#pragma unroll
            for (int i = 0; i < 16; ++i) buf[(32 * ob + chan_of(i, kg)) * WIDE_PITCH + j] = v[i];
        };
        // ---- the heads: dN_nb = (g w_out + g2 w_out2) . actvn'(net) ----
        f32x16 dn;
        if (ob < nh) {
            const float go = live ? a.grad_out[gj] : 0.0f, go2 = (live && a.grad_out2) ? a.grad_out2[gj] : 0.0f;
            const f32x16 af = load_acc16(a.save + sv.af + gr * H + 32 * ob, kg);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * ob + chan_of(i, kg);
                const float slope = af[i] > 0.0f ? 1.0f : (a.leaky ? 0.2f : 0.0f);     // leaky_relu's / relu's derivative as ATen takes it at 0
                dn[i] = (go * ow[row] + go2 * ow2[row]) * slope;
            }
            put(bufG, dn);
            if (live) store_acc16(a.gws + gw.dn + ((size_t)a.nb * d.total + gj) * H + 32 * ob, dn, kg);
        }
        f32x16 dc;
#pragma unroll
        for (int i = 0; i < 16; ++i) dc[i] = 0.0f;
        __syncthreads();
        for (int blk = a.nb - 1; blk >= 0; --blk) {
            const float *wb = a.blob_t + lay.w_blk + (size_t)blk * (lay.w_1 + lay.w_0 + lay.w_c);
            f32x16 z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.0f;
            if (ob < nh) {
                f32x16 dh = wide_gemm(z, wb, ob, H, bufG, lane);                       // W1^T dN_{i+1}
                const f32x16 a1 = load_acc16(a.save + sv.blk + ((size_t)(2 * blk + 1) * d.total + gr) * H + 32 * ob, kg);
#pragma unroll
                for (int i = 0; i < 16; ++i) dh[i] = a1[i] > 0.0f ? dh[i] : 0.0f;
                put(bufH, dh);
                if (live) store_acc16(a.gws + gw.dh + ((size_t)blk * d.total + gj) * H + 32 * ob, dh, kg);
            }
            __syncthreads();                                          // bufH complete; every read of bufG (dN_{i+1}) done
            if (ob < nh) {
                const f32x16 da = wide_gemm(z, wb + lay.w_1, ob, H, bufH, lane);       // W0^T dH_i
                const f32x16 a0 = load_acc16(a.save + sv.blk + ((size_t)(2 * blk) * d.total + gr) * H + 32 * ob, kg);
#pragma unroll
                for (int i = 0; i < 16; ++i) dn[i] += a0[i] > 0.0f ? da[i] : 0.0f;
                put(bufG, dn);                                        // dN_i
                if (live) store_acc16(a.gws + gw.dn + ((size_t)blk * d.total + gj) * H + 32 * ob, dn, kg);
            }
            __syncthreads();                                          // bufG = dN_i complete; every read of bufH done
            if (ob < nc && (a.grad_grid || a.grad_c)) dc = wide_gemm(dc, wb + lay.w_1 + lay.w_0, ob, H, bufG, lane);     // Wc_i^T dN_i
        }
        // ---- the front: d c_img = Wp[:, 3:]^T dN_0 (bufG still holds dN_0) ----
        if (ob < nc && a.grad_cimg) {
            f32x16 z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.0f;
            const f32x16 dci = wide_gemm(z, a.blob_t + lay.w_pc, ob, H, bufG, lane);
            if (live) store_acc16(a.grad_cimg + (size_t)gj * C + 32 * ob, dci, kg);
        }
        if (a.grad_c && ob < nc && live) store_acc16(a.grad_c + (size_t)gj * C + 32 * ob, dc, kg);
        // ---- d c -> the grid gradient ----
        if (a.grad_grid) {
            if (ob < nc) {
#pragma unroll
                for (int i = 0; i < 16; ++i) cl[(32 * ob + chan_of(i, kg)) * WIDE_PITCH + j] = dc[i];
            }
            __syncthreads();
            constexpr int PG = WIDE_THREADS / 32;
            const int ch = tid & 31, pg = tid >> 5;
#pragma unroll 1
            for (int i = 0; i < WIDE_PTS / PG; ++i) {
                const int pt = pg + PG * i;
                const uint32_t g = tile * WIDE_PTS + pt;
                if (g >= d.total) continue;
                const uint32_t b = g / d.N;
                float px, py, pz;
                point_of(d, g, g - b * d.N, px, py, pz);
                float *gb = a.grad_grid + (size_t)b * d.R * d.R * d.R * C;
                if (a.nearest) {
                    const size_t near = (((size_t)__builtin_rintf(grid_coord(pz, d.divisor, d.R)) * d.R + (size_t)__builtin_rintf(grid_coord(py, d.divisor, d.R))) * d.R +
                                         (size_t)__builtin_rintf(grid_coord(px, d.divisor, d.R))) * C;
                    for (int cb = 0; cb < C; cb += 32) atomicAdd(gb + near + cb + ch, cl[(cb + ch) * WIDE_PITCH + pt]);
                    continue;
                }
                const Tri t = tri_setup(px, py, pz, d.divisor, d.R);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int zz = (k & 4) ? t.z1 : t.z0, yy = (k & 2) ? t.y1 : t.y0, xx = (k & 1) ? t.x1 : t.x0;
                    const float w = (((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0)) * ((k & 4) ? t.wz1 : t.wz0);
                    if (w == 0.0f) continue;
                    float *dst = gb + (((size_t)zz * d.R + yy) * d.R + xx) * C;
                    for (int cb = 0; cb < C; cb += 32) atomicAdd(dst + cb + ch, w * cl[(cb + ch) * WIDE_PITCH + pt]);
                }
            }
        }
        __syncthreads();                                              // the buffers are free for the next tile
    }
}


// ---- the same forward on the f16 matrix core with split operands ("f16x3": W x = W_lo x_hi + W_hi x_lo + W_hi x_hi on
// v_mfma_f32_32x32x16_f16, f32 accumulation; 21-22 mantissa bits per operand: f32-level logits while the hidden activations stay
// inside the half range, which the kernel watches as the shipped-shape kernels do -- decode_common.h, range_report) -------------
// The exact-f32 kernel above is bound by the slow f32 MFMA (0.61 of its peak at 256 / 128) and, right behind it, by the weight
// stream (3.3 MB per 32 points from L2).  Here a workgroup owns 64 points as TWO 32-point groups that share every weight fragment
// (half the stream per point), the fragments are [32 rows][16 k] half pairs -- two 16-byte loads per lane and k-step, the same
// bytes as f32 -- and the activations live in LDS point-major as hi / lo half planes, so that a lane's B operand (8 consecutive
// channels of its point) is one ds_read_b128.  One activation buffer instead of two (64 points x 256 channels x 4 bytes x 2 would not
// fit beside the sampled features): a layer's output is written behind a barrier that waits for the readers of its input, i.e.
// four barriers per block; the fc_c product of the next block, which reads the sampled features only, runs in front of the first.
constexpr int WH_PTS = 64;
// pairs of 32-point groups per workgroup tile: the waves a narrow layer leaves over take further pairs (LDS: <= 4 pairs)
__host__ __device__ inline int wideh_pairs(int waves, int nh) { const int n = waves / nh; return n < 1 ? 1 : (n > 4 ? 4 : n); }
struct WideHArgs {
    DecodeArgs d;
    const float *blob;
    int H, C, nb, Kp, p_in, leaky, nearest;
    unsigned *status;
    int npair;                  // sets of NG 32-point groups per workgroup tile (the launcher sizes it: spare waves, LDS)
    int heads_on_c;             // the heads' partial sums share the c planes' LDS (the planes fill it)
};
__host__ __device__ inline int wideh_pitch(int ch) { return ch * 2 + 16; }             // bytes per point row of a half plane: 16 lanes x 16 bytes tile the 64 banks

__device__ __forceinline__ void wideh_split2(float a, float b, unsigned &hi, unsigned &lo) {
    const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    hi = __builtin_bit_cast(unsigned, hp);
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a - (float)hp[0], b - (float)hp[1]));
}

// acc_g += W[rows 32 ob ..][K] . X_g[K][32 points], g = 0 .. NG - 1; wf: the layer's fragments [ob][K / 16][hi, lo][64 lanes][8 halves].
// The fragments come from L2 (a round trip is ~10 k-steps of MFMAs): WIDEH_AHEAD k-steps of them are kept in flight in registers.
// (hidden <= 128 runs four-wave workgroups, several per CU: there the plain loop with its 122 registers hides the latency by occupancy)
// NG = 32-point groups that share every fragment: the stream out of the L2s (865 M requests per 128^3 at 256 / 128 with NG = 2:
// profiles/r06_pmc_wide_summary.csv) shrinks by 2 / NG per point.
template <int WIDEH_AHEAD, int NG>
__device__ __forceinline__ void wideh_gemm(f32x16 (&acc)[NG], const float *wf, int ob, int K, const char *xh, const char *xl, int pitch, int lane) {
    const int nks = K / 16;
    const u32x4 *w = reinterpret_cast<const u32x4 *>(wf) + (size_t)ob * nks * 128 + lane;
    const int j = lane & 31, kg = lane >> 5;
    const char *rh = xh + j * pitch + kg * 16, *rl = xl + j * pitch + kg * 16;
    auto products = [&](const f16x8 &wh, const f16x8 &wl, const f16x8 (&bh)[NG], const f16x8 (&bl)[NG]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[g], acc[g], 0, 0, 0);
    };
    auto loadb = [&](f16x8 (&bh)[NG], f16x8 (&bl)[NG], int ks) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            bh[g] = *reinterpret_cast<const f16x8 *>(rh + g * 32 * pitch + ks * 32);
            bl[g] = *reinterpret_cast<const f16x8 *>(rl + g * 32 * pitch + ks * 32);
        }
    };
    if constexpr (WIDEH_AHEAD <= 1) {
#pragma unroll 2
        for (int ks = 0; ks < nks; ++ks) {
            f16x8 bh[NG], bl[NG];
            const f16x8 wh = __builtin_bit_cast(f16x8, w[(size_t)ks * 128]), wl = __builtin_bit_cast(f16x8, w[(size_t)ks * 128 + 64]);
            loadb(bh, bl, ks);
            products(wh, wl, bh, bl);
        }
        return;
    }
    // eight-wave workgroups (one per CU, 256 registers per lane): the weight fragments WIDEH_AHEAD k-steps ahead, the activation operands one
    // k-step ahead (two register sets; with three groups one set -- the SIMD's other wave covers the LDS read), and nothing else in flight:
    // the scheduler is fenced per k-step, or it hoists every LDS read of the unrolled steps in front of the first MFMA and spills
    u32x4 wq[WIDEH_AHEAD][2];
#pragma unroll
    for (int i = 0; i < WIDEH_AHEAD; ++i)
        if (i < nks) { wq[i][0] = w[(size_t)i * 128]; wq[i][1] = w[(size_t)i * 128 + 64]; }
    constexpr int NB = NG >= 3 ? 1 : 2;
    f16x8 bh[NB][NG], bl[NB][NG];
    if (NB == 2) loadb(bh[0], bl[0], 0);
    for (int base = 0; base < nks; base += WIDEH_AHEAD) {
#pragma unroll
        for (int i = 0; i < WIDEH_AHEAD; ++i) {
            const int ks = base + i;
            if (ks >= nks) break;
            const f16x8 wh = __builtin_bit_cast(f16x8, wq[i][0]), wl = __builtin_bit_cast(f16x8, wq[i][1]);
            if (ks + WIDEH_AHEAD < nks) { wq[i][0] = w[(size_t)(ks + WIDEH_AHEAD) * 128]; wq[i][1] = w[(size_t)(ks + WIDEH_AHEAD) * 128 + 64]; }
            if (NB == 2) { if (ks + 1 < nks) loadb(bh[(i + 1) & 1], bl[(i + 1) & 1], ks + 1); }
            else loadb(bh[0], bl[0], ks);
            products(wh, wl, bh[NB == 2 ? (i & 1) : 0], bl[NB == 2 ? (i & 1) : 0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// relu(v) of one group's 16 accumulator values -> the hi / lo planes: rows 32 ob + chan_of(r, kg) of point `pt` (four runs of four channels)
__device__ __forceinline__ void wideh_store(char *ah, char *al, int pitch, int pt, int ob, int kg, const f32x16 &v, unsigned &rmax) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned h0, l0, h1, l1;
        wideh_split2(fmaxf(v[4 * q], 0.0f), fmaxf(v[4 * q + 1], 0.0f), h0, l0);
        wideh_split2(fmaxf(v[4 * q + 2], 0.0f), fmaxf(v[4 * q + 3], 0.0f), h1, l1);
        range_track(rmax, h0); range_track(rmax, h1);
        const int off = pt * pitch + (32 * ob + 8 * q + 4 * kg) * 2;
        *reinterpret_cast<u32x2 *>(ah + off) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(al + off) = u32x2{l0, l1};
    }
}

// NG: 32-point groups per wave (a "set" of GP = 32 NG points shares every fragment); AH: weight k-steps in flight per wave
template <int WIDE_WAVES, int NG, int AH>
__global__ void __launch_bounds__(WIDE_WAVES * 64)
decode_wide_h_kernel(WideHArgs a) {
    constexpr int WIDE_THREADS = WIDE_WAVES * 64, GP = 32 * NG;
    extern __shared__ __attribute__((aligned(16))) char whs[];      // c hi | c lo [64][pitch(C)] ; act hi | act lo [64][pitch(max(H, Kp))] ; heads
    const DecodeArgs &d = a.d;
    const int H = a.H, C = a.C, nh = H / 32, Kp = a.Kp;
    const int pc = wideh_pitch(C), pa = wideh_pitch(H > Kp ? H : Kp);
    // narrow layers leave waves over (hidden 64: two 32-row blocks for four waves): the spare waves take further PAIRS of 32-point
    // groups instead of idling -- wave w owns rows 32 (w % nh) of pair w / nh, the tile grows to `npair` pairs' worth of points, and a
    // weight fragment streamed from L2 serves `npair` times the points (the second wave's load of it hits L1)
    const int npair = a.npair, PTS = GP * npair;
    char *ch_ = whs, *cl_ = ch_ + PTS * pc, *ah = cl_ + PTS * pc, *al = ah + PTS * pa;
    // [waves][2 lane halves][2 heads][GP points of the wave's set]; where the planes leave no room for it, on top of the c planes (last
    // read by the last block's fc_c product, four barriers before the heads are written; rewritten behind the tile's last barrier)
    float *heads = a.heads_on_c ? reinterpret_cast<float *>(ch_) : reinterpret_cast<float *>(al + PTS * pa);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kg = lane >> 5;
    const WideLayout lay = wide_layout(H, C, a.nb, Kp);
    const float *bias = a.blob + lay.bias;
    const uint32_t ntiles = (d.total + PTS - 1) / PTS;
    unsigned rmax = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- the tile's inputs as half pairs: sampled features c (decoder.py:62-68), fc_p's input rows [p | c_img | 0] ----
        {
            constexpr int PG = WIDE_THREADS / 32;
            const int ch = tid & 31, pg = tid >> 5;
            for (int pt = pg; pt < PTS; pt += PG) {
                uint32_t g = tile * PTS + pt;
                if (g >= d.total) g = d.total - 1u;
                const uint32_t b = g / d.N;
                float px, py, pz;
                point_of(d, g, g - b * d.N, px, py, pz);
                const Tri t = tri_setup(px, py, pz, d.divisor, d.R);
                const float *gb = d.grid + (size_t)b * d.R * d.R * d.R * C;
                const size_t near = a.nearest ? (((size_t)__builtin_rintf(grid_coord(pz, d.divisor, d.R)) * d.R + (size_t)__builtin_rintf(grid_coord(py, d.divisor, d.R))) * d.R +
                                                 (size_t)__builtin_rintf(grid_coord(px, d.divisor, d.R))) * C : 0;
                for (int cb = 0; cb < C; cb += 32) {
                    float acc = 0.0f;
                    if (d.c_direct) acc = d.c_direct[(size_t)g * C + cb + ch];      // the conditioning features given directly
                    else if (a.nearest) acc = gb[near + cb + ch];
                    else {
#pragma unroll
                        for (int dz = 0; dz < 2; ++dz) {
                            const int zz = dz ? t.z1 : t.z0;
                            const float wz = dz ? t.wz1 : t.wz0;
#pragma unroll
                            for (int dy = 0; dy < 2; ++dy) {
                                const int yy = dy ? t.y1 : t.y0;
                                const float wy = dy ? t.wy1 : t.wy0;
                                const size_t row = ((size_t)zz * d.R + yy) * d.R;
                                acc = fmaf(gb[(row + t.x0) * C + cb + ch], (t.wx0 * wy) * wz, acc);
                                acc = fmaf(gb[(row + t.x1) * C + cb + ch], (t.wx1 * wy) * wz, acc);
                            }
                        }
                    }
                    const _Float16 hv = (_Float16)acc;
                    *reinterpret_cast<_Float16 *>(ch_ + pt * pc + (cb + ch) * 2) = hv;
                    *reinterpret_cast<_Float16 *>(cl_ + pt * pc + (cb + ch) * 2) = (_Float16)(acc - (float)hv);
                }
                for (int k = ch; k < Kp; k += 32) {
                    float v = 0.0f;
                    if (k < 3) v = k == 0 ? px : (k == 1 ? py : pz);
                    else if (a.p_in > 3 && k < a.p_in) v = wide_cimg(d, g, k - 3, a.p_in - 3);
                    const _Float16 hv = (_Float16)v;
                    *reinterpret_cast<_Float16 *>(ah + pt * pa + k * 2) = hv;
                    *reinterpret_cast<_Float16 *>(al + pt * pa + k * 2) = (_Float16)(v - (float)hv);
                }
            }
        }
        __syncthreads();
        const int ob = wave % nh, pr = wave / nh;                   // nh <= WIDE_WAVES; waves beyond nh * npair only keep the barriers
        const bool on = pr < npair;
        // this wave's set of point groups: rows GP pr .. GP pr + GP - 1 of the planes
        char *const chp = ch_ + pr * GP * pc, *const clp = cl_ + pr * GP * pc, *const ahp = ah + pr * GP * pa, *const alp = al + pr * GP * pa;
        // ---- fc_p (decoder.py:139 / 81) ----
        f32x16 net[NG];
        if (on) {
            net[0] = bias16(bias, ob, kg);
#pragma unroll
            for (int g = 1; g < NG; ++g) net[g] = net[0];
            wideh_gemm<AH, NG>(net, a.blob + lay.w_p, ob, Kp, ahp, alp, pa, lane);
        }
        // ---- n_blocks x (fc_c add, ResnetBlockFC: layers.py:41-50; its activations are ReLU) ----
        for (int blk = 0; blk < a.nb; ++blk) {
            const float *wb = a.blob + lay.w_blk + (size_t)blk * (lay.w_c + lay.w_0 + lay.w_1);
            const float *bb = bias + (size_t)H * (1 + 3 * blk);
            if (on) {
                const f32x16 bc = bias16(bb, ob, kg);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) net[g][i] += bc[i];
                wideh_gemm<AH, NG>(net, wb, ob, C, chp, clp, pc, lane);
            }
            __syncthreads();                                        // the readers of the activation buffer (fc_p / the last fc_1) are done
            if (on) {
#pragma unroll
                for (int g = 0; g < NG; ++g) wideh_store(ahp, alp, pa, j + 32 * g, ob, kg, net[g], rmax);
            }
            __syncthreads();
            f32x16 hid[NG];
            if (on) {
                hid[0] = bias16(bb + H, ob, kg);
#pragma unroll
                for (int g = 1; g < NG; ++g) hid[g] = hid[0];
                wideh_gemm<AH, NG>(hid, wb + lay.w_c, ob, H, ahp, alp, pa, lane);
            }
            __syncthreads();                                        // fc_0's readers are done
            if (on) {
#pragma unroll
                for (int g = 0; g < NG; ++g) wideh_store(ahp, alp, pa, j + 32 * g, ob, kg, hid[g], rmax);
            }
            __syncthreads();
            if (on) {
                const f32x16 b1 = bias16(bb + 2 * H, ob, kg);
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) net[g][i] += b1[i];
                wideh_gemm<AH, NG>(net, wb + lay.w_c + lay.w_0, ob, H, ahp, alp, pa, lane);
            }
        }
        // ---- fc_out / fc_out_contact on actvn(net) (decoder.py:157-158, 128-131): f32 dot products, as in the exact kernel ----
        const float *ow = bias + (size_t)H * (1 + 3 * a.nb), *ow2 = ow + H + 1;
        float o1[NG], o2[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { o1[g] = 0.0f; o2[g] = 0.0f; }
        if (on) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * ob + chan_of(i, kg);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float v = actvn(net[g][i], a.leaky);
                    o1[g] = fmaf(ow[row], v, o1[g]); o2[g] = fmaf(ow2[row], v, o2[g]);
                }
            }
        }
        if (a.heads_on_c) __syncthreads();                          // (the c planes' last readers are four barriers back already; kept explicit)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            heads[((wave * 2 + kg) * 2 + 0) * GP + j + 32 * g] = o1[g];
            heads[((wave * 2 + kg) * 2 + 1) * GP + j + 32 * g] = o2[g];
        }
        __syncthreads();
        for (int e = tid; e < 2 * PTS; e += WIDE_THREADS) {
            const int pt = e % PTS, which = e / PTS, pp = pt / GP, q = pt % GP;
            float o = which ? ow2[H] : ow[H];
            for (int w = 2 * pp * nh; w < 2 * (pp + 1) * nh; ++w)     // the waves of this point's set and their lane halves, in a fixed order
                o += heads[(w * 2 + which) * GP + q];
            const uint32_t g = tile * PTS + pt;
            float *dst = which ? d.out2 : d.out;
            if (g < d.total && dst) dst[g] = o;
        }
        __syncthreads();
    }
    range_report(rmax, a.status);
}

// W [H][K] (row stride ld, columns >= kin read as zero) -> split-f16 fragments [H/32][K/16][hi, lo][64 lanes][8 halves]:
// lane (row r = l & 31, kg = l >> 5), element e = W[32 ob + r][16 ks + 8 kg + e]
__global__ void wideh_pack_kernel(const float *w, int H, int K, int kin, int ld, float *dst) {
    const size_t frags = (size_t)H * K / 8;                          // (hi, lo) fragment pairs of eight values: one per (row block, k-step, lane)
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < frags; f += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(f & 63);
        const size_t r = f >> 6;
        const int ks = (int)(r % (K / 16)), ob = (int)(r / (K / 16));
        const int row = 32 * ob + (l & 31), k0 = 16 * ks + 8 * (l >> 5);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = k0 + e < kin ? w[(size_t)row * ld + k0 + e] : 0.0f;
            const _Float16 h = (_Float16)x;
            hi[e] = h;
            lo[e] = (_Float16)(x - (float)h);
        }
        f16x8 *o = reinterpret_cast<f16x8 *>(dst) + ((size_t)ob * (K / 16) + ks) * 128 + l;
        o[0] = hi; o[64] = lo;
    }
}

#include "decode_wide_pipe.inc"

// the shapes of decode_wide_p_kernel (VTACO_WIDE_PIPE=0: the streaming kernel, A/B)
static bool wide_pipe_shape(int hidden, int c_dim, int n_blocks, int p_in) {
    static const bool off = getenv("VTACO_WIDE_PIPE") && getenv("VTACO_WIDE_PIPE")[0] == '0';
    return !off && hidden == 64 && c_dim == 32 && n_blocks >= 1 && n_blocks <= 5 && p_in == 3;
}

int wide_shape_ok(int hidden, int c_dim, int n_blocks, int p_in) {
    return hidden >= 32 && hidden <= WIDE_MAX && hidden % 32 == 0 && c_dim >= 32 && c_dim <= WIDE_MAX && c_dim % 32 == 0 &&
           n_blocks >= 1 && n_blocks <= VT_MAX_BLOCKS && (p_in == 3 || p_in == 3 + c_dim);
}

}  // namespace

extern "C" {

size_t vt_decoder_wide_blob_bytes(int hidden, int c_dim, int n_blocks, int p_in) {
    if (!wide_shape_ok(hidden, c_dim, n_blocks, p_in)) return 0;
    return wide_layout(hidden, c_dim, n_blocks, (p_in + 7) / 8 * 8).total * sizeof(float);
}

int vt_decoder_pack_wide(const vt_decoder_params *p, float *blob, size_t blob_bytes, void *stream) {
    if (!p || !blob) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide: null argument");
    const int H = p->hidden, C = p->c_dim, nb = p->n_blocks, Kp = (p->p_in + 7) / 8 * 8;
    if (!wide_shape_ok(H, C, nb, p->p_in))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decoder_pack_wide: hidden and c_dim must be multiples of 32 up to 256, p_in 3 or 3 + c_dim");
    const WideLayout lay = wide_layout(H, C, nb, Kp);
    if (blob_bytes < lay.total * sizeof(float)) return vt_fail(VT_ERR_WORKSPACE, "vt_decoder_pack_wide: blob too small");
    if (!p->fc_p_w || !p->fc_p_b || !p->fc_out_w || !p->fc_out_b) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide: null parameter");
    hipStream_t st = (hipStream_t)stream;
    auto pack = [&](const float *w, int K, int kin, int ld, float *dst) {
        const size_t total = (size_t)H * K;
        hipLaunchKernelGGL(wide_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, H, K, kin, (size_t)ld, (size_t)1, dst);
    };
    auto copy = [&](const float *src, float *dst, int n) {
        hipLaunchKernelGGL(wide_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
    };
    pack(p->fc_p_w, Kp, p->p_in, p->p_in, blob + lay.w_p);
    float *bias = blob + lay.bias;
    copy(p->fc_p_b, bias, H);
    for (int i = 0; i < nb; ++i) {
        if (!p->fc_c_w[i] || !p->fc_c_b[i] || !p->fc0_w[i] || !p->fc0_b[i] || !p->fc1_w[i] || !p->fc1_b[i])
            return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide: null block parameter");
        float *wb = blob + lay.w_blk + (size_t)i * (lay.w_c + lay.w_0 + lay.w_1);
        pack(p->fc_c_w[i], C, C, C, wb);
        pack(p->fc0_w[i], H, H, H, wb + lay.w_c);
        pack(p->fc1_w[i], H, H, H, wb + lay.w_c + lay.w_0);
        copy(p->fc_c_b[i], bias + (size_t)H * (1 + 3 * i), H);
        copy(p->fc0_b[i], bias + (size_t)H * (2 + 3 * i), H);
        copy(p->fc1_b[i], bias + (size_t)H * (3 + 3 * i), H);
    }
    float *ow = bias + (size_t)H * (1 + 3 * nb);
    copy(p->fc_out_w, ow, H);
    copy(p->fc_out_b, ow + H, 1);
    copy(p->fc_out2_w, ow + H + 1, H);
    copy(p->fc_out2_b, ow + 2 * H + 1, 1);
    return vt_check(hipGetLastError(), "vt_decoder_pack_wide");
}

}  // extern "C"

static int wide_fwd_impl(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                         int lattice_nx, float lattice_box, int64_t lattice_first,
                         const float *c_img, const float *blob, int hidden, int n_blocks, int flags, double padding,
                         float *out, float *out2, float *save, void *stream,
                         const unsigned char *finger_ids = nullptr, const float *finger_feats = nullptr, const float *c_direct = nullptr, int n_fingers = 0) {
    if ((!grid_cl && !c_direct) || !blob || !out) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide: null argument");
    if (c_img && finger_ids) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide: give c_img or finger ids, not both");
    const int p_in = (c_img || finger_ids) ? 3 + C : 3;
    if (!wide_shape_ok(hidden, C, n_blocks, p_in))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_wide: hidden and c_dim must be multiples of 32 up to 256");
    if (B <= 0 || R < 2 || N <= 0) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide: bad size");
    if ((unsigned long long)B * (unsigned long long)N >= 0x7fffffffull) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_wide: B*N must stay below 2^31");
    if (!pts) {
        if (lattice_nx < 2 || lattice_first < 0 || lattice_first + N > (int64_t)lattice_nx * lattice_nx * lattice_nx)
            return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide: lattice range outside nx^3");
    }
    WideArgs a{};
    a.d.grid = grid_cl; a.d.pts = pts; a.d.c_img = c_img; a.d.out = out; a.d.out2 = out2;
    a.d.cimg_ids = finger_ids; a.d.cimg_table = finger_ids ? finger_feats : nullptr; a.d.cimg_nf = finger_ids ? (uint32_t)n_fingers : 0u; a.d.c_direct = c_direct;
    a.d.N = (uint32_t)N; a.d.total = (uint32_t)((uint64_t)B * (uint64_t)N); a.d.lattice_first = (uint32_t)lattice_first;
    a.d.R = R; a.d.nx = lattice_nx; a.d.box = lattice_box; a.d.divisor = (float)(1.0 + padding + 10e-4);
    a.blob = blob; a.H = hidden; a.C = C; a.nb = n_blocks; a.p_in = p_in; a.Kp = (p_in + 7) / 8 * 8; a.leaky = (flags & VT_WIDE_LEAKY) ? 1 : 0; a.nearest = (flags & VT_WIDE_NEAREST) ? 1 : 0;
    a.save = save;
    const int rowsA = hidden > a.Kp ? hidden : a.Kp;
    const int waves = hidden <= 128 ? 4 : 8;
    const size_t lds = ((size_t)(C + rowsA + hidden) * WIDE_PITCH + (size_t)waves * 2 * 2 * 32) * sizeof(float);
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_wide_kernel<4>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_wide_kernel<8>), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_decode_fwd_wide: hipFuncSetAttribute");
        attr = true;
    }
    const uint32_t ntiles = (a.d.total + WIDE_PTS - 1) / WIDE_PTS;
    const uint32_t cap = (uint32_t)vt_num_cus() * 8u;
    const dim3 grid(ntiles < cap ? ntiles : cap);
    if (waves == 4) hipLaunchKernelGGL(decode_wide_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decode_wide_kernel<8>, grid, dim3(512), lds, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decode_fwd_wide");
}

extern "C" {

int vt_decode_fwd_wide(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                       int lattice_nx, float lattice_box, int64_t lattice_first,
                       const float *c_img, const float *blob, int hidden, int n_blocks, int flags, double padding,
                       float *out, float *out2, void *stream) {
    return wide_fwd_impl(grid_cl, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, c_img, blob, hidden, n_blocks, flags, padding,
                         out, out2, nullptr, stream);
}

// the same forward with the tactile feature by finger id (generation.py:159-255 builds the dense [1, nx^3, C] tensor on the host:
// 8.6 GB at 256^3 / c_dim 128; here one byte per point and the [F][C] table)
int vt_decode_fwd_wide_ids(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                           int lattice_nx, float lattice_box, int64_t lattice_first,
                           const unsigned char *finger_ids, const float *finger_feats, int n_fingers,
                           const float *blob, int hidden, int n_blocks, int flags, double padding,
                           float *out, float *out2, void *stream) {
    if (!finger_ids || !finger_feats || n_fingers <= 0 || n_fingers > 255) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide_ids: bad finger table");
    return wide_fwd_impl(grid_cl, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, blob, hidden, n_blocks, flags, padding,
                         out, out2, nullptr, stream, finger_ids, finger_feats, nullptr, n_fingers);
}

// ---- split-f16 form of the same forward (inference) ----
size_t vt_decoder_wide_blob_f16x3_bytes(int hidden, int c_dim, int n_blocks, int p_in) {
    if (!wide_shape_ok(hidden, c_dim, n_blocks, p_in)) return 0;
    return wide_layout(hidden, c_dim, n_blocks, (p_in + 15) / 16 * 16).total * sizeof(float);
}

int vt_decoder_pack_wide_f16x3(const vt_decoder_params *p, float *blob, size_t blob_bytes, void *stream) {
    if (!p || !blob) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide_f16x3: null argument");
    const int H = p->hidden, C = p->c_dim, nb = p->n_blocks, Kp = (p->p_in + 15) / 16 * 16;
    if (!wide_shape_ok(H, C, nb, p->p_in))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decoder_pack_wide_f16x3: hidden and c_dim must be multiples of 32 up to 256, p_in 3 or 3 + c_dim");
    const WideLayout lay = wide_layout(H, C, nb, Kp);
    if (blob_bytes < lay.total * sizeof(float)) return vt_fail(VT_ERR_WORKSPACE, "vt_decoder_pack_wide_f16x3: blob too small");
    if (!p->fc_p_w || !p->fc_p_b || !p->fc_out_w || !p->fc_out_b) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide_f16x3: null parameter");
    (void)vt_decode_status_dev();                                      // the device's status block exists before a launch can be captured into a graph
    hipStream_t st = (hipStream_t)stream;
    auto pack = [&](const float *w, int K, int kin, int ld, float *dst) {
        const size_t frags = (size_t)H * K / 8;
        hipLaunchKernelGGL(wideh_pack_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, st, w, H, K, kin, ld, dst);
    };
    auto copy = [&](const float *src, float *dst, int n) {
        hipLaunchKernelGGL(wide_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
    };
    pack(p->fc_p_w, Kp, p->p_in, p->p_in, blob + lay.w_p);
    float *bias = blob + lay.bias;
    copy(p->fc_p_b, bias, H);
    for (int i = 0; i < nb; ++i) {
        if (!p->fc_c_w[i] || !p->fc_c_b[i] || !p->fc0_w[i] || !p->fc0_b[i] || !p->fc1_w[i] || !p->fc1_b[i])
            return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide_f16x3: null block parameter");
        float *wb = blob + lay.w_blk + (size_t)i * (lay.w_c + lay.w_0 + lay.w_1);
        pack(p->fc_c_w[i], C, C, C, wb);
        pack(p->fc0_w[i], H, H, H, wb + lay.w_c);
        pack(p->fc1_w[i], H, H, H, wb + lay.w_c + lay.w_0);
        copy(p->fc_c_b[i], bias + (size_t)H * (1 + 3 * i), H);
        copy(p->fc0_b[i], bias + (size_t)H * (2 + 3 * i), H);
        copy(p->fc1_b[i], bias + (size_t)H * (3 + 3 * i), H);
    }
    float *ow = bias + (size_t)H * (1 + 3 * nb);
    copy(p->fc_out_w, ow, H);
    copy(p->fc_out_b, ow + H, 1);
    copy(p->fc_out2_w, ow + H + 1, H);
    copy(p->fc_out2_b, ow + 2 * H + 1, 1);
    return vt_check(hipGetLastError(), "vt_decoder_pack_wide_f16x3");
}

static int wideh_fwd_impl(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                          int lattice_nx, float lattice_box, int64_t lattice_first,
                          const float *c_img, const unsigned char *finger_ids, const float *finger_feats,
                          const float *blob, int hidden, int n_blocks, int flags, double padding,
                          float *out, float *out2, void *stream, const float *c_direct = nullptr, void *ws = nullptr, size_t ws_bytes = 0, int n_fingers = 0) {
    if ((!grid_cl && !c_direct) || !blob || !out) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide_f16x3: null argument");
    if (c_img && finger_ids) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide_f16x3: give c_img or finger ids, not both");
    const int p_in = (c_img || finger_ids) ? 3 + C : 3;
    if (!wide_shape_ok(hidden, C, n_blocks, p_in))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_wide_f16x3: hidden and c_dim must be multiples of 32 up to 256");
    if (B <= 0 || R < 2 || N <= 0) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide_f16x3: bad size");
    if ((unsigned long long)B * (unsigned long long)N >= 0x7fffffffull) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_wide_f16x3: B*N must stay below 2^31");
    if (!pts && (lattice_nx < 2 || lattice_first < 0 || lattice_first + N > (int64_t)lattice_nx * lattice_nx * lattice_nx))
        return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide_f16x3: lattice range outside nx^3");
    WideHArgs a{};
    a.d.grid = grid_cl; a.d.pts = pts; a.d.c_img = c_img; a.d.out = out; a.d.out2 = out2;
    a.d.cimg_ids = finger_ids; a.d.cimg_table = finger_ids ? finger_feats : nullptr; a.d.cimg_nf = finger_ids ? (uint32_t)n_fingers : 0u; a.d.c_direct = c_direct;
    a.d.N = (uint32_t)N; a.d.total = (uint32_t)((uint64_t)B * (uint64_t)N); a.d.lattice_first = (uint32_t)lattice_first;
    a.d.R = R; a.d.nx = lattice_nx; a.d.box = lattice_box; a.d.divisor = (float)(1.0 + padding + 10e-4);
    a.blob = blob; a.H = hidden; a.C = C; a.nb = n_blocks; a.p_in = p_in; a.Kp = (p_in + 15) / 16 * 16;
    a.leaky = (flags & VT_WIDE_LEAKY) ? 1 : 0; a.nearest = (flags & VT_WIDE_NEAREST) ? 1 : 0;
    a.status = vt_decode_status_dev();
    // 64 / 32 / <= 5 without tactile input columns: the weights in registers, the tiles through a pipeline of waves (decode_wide_pipe.inc)
    // on features given per point -- the caller's (c_direct), or the grid's samples left in the caller's workspace by a pre-pass
    // (the pipeline reads the features as 16-byte pieces: a feature tensor or workspace that is not 16-byte aligned keeps the streaming kernel)
    if (wide_pipe_shape(hidden, C, n_blocks, p_in) &&
        (c_direct ? ((size_t)c_direct & 15) == 0 : (ws && ((size_t)ws & 15) == 0 && ws_bytes >= (size_t)a.d.total * C * sizeof(float)))) {
        if (!c_direct) {
            // a lattice slab of whole x-plane pairs: the LDS-staged gather of the shipped-shape kernels (vt_st3_sample_lattice: footprints by
            // LDS-DMA, the same corner and FMA order, 30 us per 2^19 points against 43 for the plain gather); anything else the plain gather
            int covered = 0;
            if (!pts && !a.nearest) {
                const int src = vt_st3_sample_lattice(grid_cl, B, R, C, N, lattice_nx, lattice_box, lattice_first, padding, (float *)ws, stream, &covered);
                if (src) return src;
            }
            if (!covered) {
                const unsigned long long items = (unsigned long long)a.d.total * (C / 4);
                const unsigned long long want = (items + 255) / 256, cap = (unsigned long long)vt_num_cus() * 32ull;
                hipLaunchKernelGGL(wide_sample_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, (hipStream_t)stream, a.d, a.nearest, C, (float *)ws);
            }
            a.d.c_direct = (const float *)ws;
        }
        const size_t lds = wp_lds_bytes(n_blocks);
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_wide_p_kernel), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_decode_fwd_wide_f16x3: hipFuncSetAttribute");
        const uint32_t ntiles = (a.d.total + WP_PTS - 1) / WP_PTS, cap = (uint32_t)vt_num_cus();
#ifdef VT_DIAG_WP
        {   // (variant build: the stage waves' phase counts of this launch, printed per wave index)
            static unsigned long long *dbg = nullptr;
            if (!dbg) hipMalloc(&dbg, (size_t)cap * 16 * 8 * sizeof(unsigned long long));
            hipMemsetAsync(dbg, 0, (size_t)cap * 16 * 8 * sizeof(unsigned long long), (hipStream_t)stream);
            a.d.save = reinterpret_cast<float *>(dbg);
            hipLaunchKernelGGL(decode_wide_p_kernel, dim3(ntiles < cap ? ntiles : cap), dim3((2 * n_blocks + 2) * 64), lds, (hipStream_t)stream, a);
            hipStreamSynchronize((hipStream_t)stream);
            static unsigned long long host[512 * 16 * 8];
            hipMemcpy(host, dbg, (size_t)cap * 16 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            if (getenv("VTACO_WP_PRINT"))
                for (int w = 0; w < 2 * n_blocks; ++w) {
                    double s_[8] = {0};
                    for (uint32_t g = 0; g < cap; ++g) for (int i = 0; i < 8; ++i) s_[i] += (double)host[((size_t)g * 16 + w) * 8 + i] / cap;
                    fprintf(stderr, "wave %2d: wait-res %8.0f  A %8.0f  bar1 %8.0f  B1 %8.0f  bar2 %8.0f  B2 %8.0f  loop %8.0f   (counts per launch)\n", w, s_[1], s_[2], s_[3], s_[4], s_[5], s_[6], s_[0]);
                }
            return vt_check(hipGetLastError(), "vt_decode_fwd_wide_f16x3");
        }
#endif
        hipLaunchKernelGGL(decode_wide_p_kernel, dim3(ntiles < cap ? ntiles : cap), dim3((2 * n_blocks + 2) * 64), lds, (hipStream_t)stream, a);
        return vt_check(hipGetLastError(), "vt_decode_fwd_wide_f16x3");
    }
    const int waves = hidden <= 128 ? 4 : 8;
    const int wid = hidden > a.Kp ? hidden : a.Kp;
    // pairs of point groups per tile: what the spare waves can take, as far as the planes fit the LDS
    int npair = wideh_pairs(waves, hidden / 32);
    // groups per wave: three where the eight-wave workgroup's planes still fit the LDS (256 / 128: 150 KB with the heads' partial sums on
    // top of the c planes) -- a fragment out of the L2s then serves 96 points instead of 64; two elsewhere (VTACO_WIDE_NG3=0: two everywhere)
    auto planes_of = [&](int gp, int np) { return (size_t)2 * gp * np * wideh_pitch(C) + (size_t)2 * gp * np * wideh_pitch(wid); };
    auto heads_of = [&](int gp) { return (size_t)waves * 2 * 2 * gp * sizeof(float); };
    static const bool ng3_off = getenv("VTACO_WIDE_NG3") != nullptr && atoi(getenv("VTACO_WIDE_NG3")) == 0;
    const bool ng3 = waves == 8 && npair == 1 && !ng3_off && planes_of(96, 1) <= 160 * 1024 && heads_of(96) <= (size_t)2 * 96 * wideh_pitch(C);
    const int gp = ng3 ? 96 : WH_PTS;
    while (npair > 1 && planes_of(gp, npair) + heads_of(gp) > 150 * 1024) --npair;
    a.npair = npair;
    a.heads_on_c = ng3 && planes_of(gp, npair) + heads_of(gp) > 160 * 1024;
    const int tile_pts = gp * npair;
    const size_t lds = planes_of(gp, npair) + (a.heads_on_c ? 0 : heads_of(gp));
    if (lds > 160 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_wide_f16x3: the activation planes of this shape do not fit the LDS");
    const uint32_t ntiles = (a.d.total + tile_pts - 1) / tile_pts;
    const uint32_t cap = (uint32_t)vt_num_cus() * 4u;
    const dim3 grid(ntiles < cap ? ntiles : cap);
    auto go = [&](auto kern, int threads) -> int {
        const hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(kern), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_decode_fwd_wide_f16x3: hipFuncSetAttribute");
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, (hipStream_t)stream, a);
        return 0;
    };
    // (fragment k-steps in flight: 2 .. 8 measured within 3 % of each other at 256 / 128; 4 keeps both forms off the stack)
    const int lrc = waves == 4 ? go(&decode_wide_h_kernel<4, 2, 1>, 256) : ng3 ? go(&decode_wide_h_kernel<8, 3, 4>, 512) : go(&decode_wide_h_kernel<8, 2, 4>, 512);
    if (lrc) return lrc;
    return vt_check(hipGetLastError(), "vt_decode_fwd_wide_f16x3");
}

int vt_decode_fwd_wide_f16x3(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                             int lattice_nx, float lattice_box, int64_t lattice_first,
                             const float *c_img, const float *blob, int hidden, int n_blocks, int flags, double padding,
                             float *out, float *out2, void *stream) {
    return wideh_fwd_impl(grid_cl, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, c_img, nullptr, nullptr, blob, hidden, n_blocks,
                          flags, padding, out, out2, stream);
}

size_t vt_decode_wide_f16x3_workspace_bytes(int64_t total_points, int hidden, int c_dim, int n_blocks, int tactile) {
    if (total_points <= 0 || !wide_pipe_shape(hidden, c_dim, n_blocks, tactile ? 3 + c_dim : 3)) return 0;
    return (size_t)total_points * c_dim * sizeof(float);
}

int vt_decode_fwd_wide_f16x3_ws(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                                int lattice_nx, float lattice_box, int64_t lattice_first,
                                const float *c_img, const float *blob, int hidden, int n_blocks, int flags, double padding,
                                float *out, float *out2, void *workspace, size_t workspace_bytes, void *stream) {
    return wideh_fwd_impl(grid_cl, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, c_img, nullptr, nullptr, blob, hidden, n_blocks,
                          flags, padding, out, out2, stream, nullptr, workspace, workspace_bytes);
}

int vt_decode_fwd_wide_f16x3_ids(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                                 int lattice_nx, float lattice_box, int64_t lattice_first,
                                 const unsigned char *finger_ids, const float *finger_feats, int n_fingers,
                                 const float *blob, int hidden, int n_blocks, int flags, double padding,
                                 float *out, float *out2, void *stream) {
    if (!finger_ids || !finger_feats || n_fingers <= 0 || n_fingers > 255) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide_f16x3_ids: bad finger table");
    return wideh_fwd_impl(grid_cl, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, finger_ids, finger_feats, blob, hidden,
                          n_blocks, flags, padding, out, out2, stream, nullptr, nullptr, 0, n_fingers);
}

// the conditioned MLP alone on features given per point (AttentionDecoder.forward_img behind its fuser, decoder.py:259-271): c [B][N][C]
// instead of the grid gather; the blob is the one packed with fc_p (p_in = 3)
int vt_decode_mlp_fwd_wide(const float *c, int B, int C, const float *pts, int64_t N,
                           int lattice_nx, float lattice_box, int64_t lattice_first,
                           const float *blob, int hidden, int n_blocks, int flags, float *out, float *out2, void *stream) {
    if (!c) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_fwd_wide: null features");
    return wide_fwd_impl(nullptr, B, 2, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, blob, hidden, n_blocks, flags, 0.1,
                         out, out2, nullptr, stream, nullptr, nullptr, c);
}

int vt_decode_mlp_fwd_wide_f16x3(const float *c, int B, int C, const float *pts, int64_t N,
                                 int lattice_nx, float lattice_box, int64_t lattice_first,
                                 const float *blob, int hidden, int n_blocks, int flags, float *out, float *out2, void *stream) {
    if (!c) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_fwd_wide_f16x3: null features");
    return wideh_fwd_impl(nullptr, B, 2, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, nullptr, nullptr, blob, hidden, n_blocks,
                          flags, 0.1, out, out2, stream, c);
}

// ---- training (decoder.py:24-51, 135-161 under autograd: training.py:476-489, 879) ----
size_t vt_decode_wide_save_floats(int64_t total_points, int hidden, int c_dim, int n_blocks) {
    if (total_points <= 0 || !wide_shape_ok(hidden, c_dim, n_blocks, 3)) return 0;
    return wide_save_layout((size_t)total_points, hidden, c_dim, n_blocks).total;
}

size_t vt_decode_wide_gws_floats(int64_t total_points, int hidden, int c_dim, int n_blocks) {
    if (total_points <= 0 || !wide_shape_ok(hidden, c_dim, n_blocks, 3)) return 0;
    return wide_gws_layout((size_t)total_points, hidden, n_blocks).total;
}

int vt_decode_fwd_wide_train(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                             const float *c_img, const float *blob, int hidden, int n_blocks, int flags, double padding,
                             float *out, float *out2, float *save, void *stream) {
    if (!pts || !save) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide_train: null argument");
    return wide_fwd_impl(grid_cl, B, R, C, pts, N, 0, 0.0f, 0, c_img, blob, hidden, n_blocks, flags, padding, out, out2, save, stream);
}

size_t vt_decoder_wide_blob_t_bytes(int hidden, int c_dim, int n_blocks) {
    if (!wide_shape_ok(hidden, c_dim, n_blocks, 3)) return 0;
    return wide_layout_t(hidden, c_dim, n_blocks).total * sizeof(float);
}

int vt_decoder_pack_wide_t(const vt_decoder_params *p, float *blob_t, size_t blob_bytes, void *stream) {
    if (!p || !blob_t) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide_t: null argument");
    const int H = p->hidden, C = p->c_dim, nb = p->n_blocks;
    if (!wide_shape_ok(H, C, nb, p->p_in))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decoder_pack_wide_t: hidden and c_dim must be multiples of 32 up to 256, p_in 3 or 3 + c_dim");
    const WideLayoutT lay = wide_layout_t(H, C, nb);
    if (blob_bytes < lay.total * sizeof(float)) return vt_fail(VT_ERR_WORKSPACE, "vt_decoder_pack_wide_t: blob too small");
    if (!p->fc_p_w || !p->fc_out_w) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide_t: null parameter");
    hipStream_t st = (hipStream_t)stream;
    // fragments of M [rows][K] with M[r][k] = src[r * rs + k * ks]
    auto pack = [&](const float *src, int rows, int K, size_t rs, size_t ks, float *dst) {
        const size_t total = (size_t)rows * K;
        hipLaunchKernelGGL(wide_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, rows, K, K, rs, ks, dst);
    };
    auto copy = [&](const float *src, float *dst, int n) {
        hipLaunchKernelGGL(wide_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
    };
    for (int i = 0; i < nb; ++i) {
        if (!p->fc_c_w[i] || !p->fc0_w[i] || !p->fc1_w[i]) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_wide_t: null block parameter");
        float *wb = blob_t + lay.w_blk + (size_t)i * (lay.w_1 + lay.w_0 + lay.w_c);
        pack(p->fc1_w[i], H, H, 1, (size_t)H, wb);                         // W1^T [H][H]: (r, k) = W1[k][r]
        pack(p->fc0_w[i], H, H, 1, (size_t)H, wb + lay.w_1);
        pack(p->fc_c_w[i], C, H, 1, (size_t)C, wb + lay.w_1 + lay.w_0);    // Wc^T [C][H]: (r, k) = Wc[k][r], Wc is [H][C]
    }
    // fc_p_img's c_img columns transposed [C][H]: (r, k) = Wp[k][3 + r]; a decoder packed with fc_p (p_in = 3) has none
    pack(p->p_in > 3 ? p->fc_p_w + 3 : nullptr, C, H, 1, (size_t)p->p_in, blob_t + lay.w_pc);
    copy(p->fc_out_w, blob_t + lay.heads, H);
    copy(p->fc_out2_w, blob_t + lay.heads + H, H);
    return vt_check(hipGetLastError(), "vt_decoder_pack_wide_t");
}

}  // extern "C"

static int wide_bwd_impl(int B, int R, int C, const float *pts, int64_t N, const float *blob_t, int hidden, int n_blocks, int flags,
                         double padding, const float *grad_out, const float *grad_out2, const float *save, float *gws,
                         float *grad_grid_cl, float *grad_c_img, float *grad_c, void *stream) {
    if (!pts || !blob_t || !grad_out || !save || !gws) return vt_fail(VT_ERR_INVALID, "vt_decode_bwd_wide: null argument");
    if (!wide_shape_ok(hidden, C, n_blocks, 3))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_bwd_wide: hidden and c_dim must be multiples of 32 up to 256");
    if (B <= 0 || R < 2 || N <= 0) return vt_fail(VT_ERR_INVALID, "vt_decode_bwd_wide: bad size");
    if ((unsigned long long)B * (unsigned long long)N >= 0x7fffffffull) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_bwd_wide: B*N must stay below 2^31");
    WideBwdArgs a{};
    a.d.pts = pts; a.d.N = (uint32_t)N; a.d.total = (uint32_t)((uint64_t)B * (uint64_t)N); a.d.R = R;
    a.d.divisor = (float)(1.0 + padding + 10e-4);
    a.blob_t = blob_t; a.save = save; a.grad_out = grad_out; a.grad_out2 = grad_out2; a.gws = gws;
    a.grad_grid = grad_grid_cl; a.grad_cimg = grad_c_img; a.grad_c = grad_c;
    a.H = hidden; a.C = C; a.nb = n_blocks; a.leaky = (flags & VT_WIDE_LEAKY) ? 1 : 0; a.nearest = (flags & VT_WIDE_NEAREST) ? 1 : 0;
    const int widest = hidden > C ? hidden : C;
    const int waves = widest <= 128 ? 4 : 8;
    const size_t lds = (size_t)(2 * hidden + C) * WIDE_PITCH * sizeof(float);
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_wide_bwd_kernel<4>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_wide_bwd_kernel<8>), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_decode_bwd_wide: hipFuncSetAttribute");
        attr = true;
    }
    const uint32_t ntiles = (a.d.total + WIDE_PTS - 1) / WIDE_PTS;
    const uint32_t cap = (uint32_t)vt_num_cus() * 8u;
    const dim3 grid(ntiles < cap ? ntiles : cap);
    if (waves == 4) hipLaunchKernelGGL(decode_wide_bwd_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decode_wide_bwd_kernel<8>, grid, dim3(512), lds, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decode_bwd_wide");
}

extern "C" {

int vt_decode_bwd_wide(int B, int R, int C, const float *pts, int64_t N, const float *blob_t, int hidden, int n_blocks, int flags,
                       double padding, const float *grad_out, const float *grad_out2, const float *save, float *gws,
                       float *grad_grid_cl, float *grad_c_img, void *stream) {
    return wide_bwd_impl(B, R, C, pts, N, blob_t, hidden, n_blocks, flags, padding, grad_out, grad_out2, save, gws, grad_grid_cl, grad_c_img,
                         nullptr, stream);
}

// the conditioned MLP on given features under autograd (AttentionDecoder.forward_img behind its fuser at the widths beyond 32 / 32):
// the forward of vt_decode_mlp_fwd_wide that keeps every layer's input (the c slot of ``save`` stays unwritten: the caller holds c),
// and the data pass that returns d c per point [B][N][C] instead of scattering it to a grid
int vt_decode_mlp_fwd_wide_train(const float *c, int B, int C, const float *pts, int64_t N, const float *blob, int hidden, int n_blocks,
                                 int flags, float *out, float *out2, float *save, void *stream) {
    if (!c || !pts || !save) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_fwd_wide_train: null argument");
    return wide_fwd_impl(nullptr, B, 2, C, pts, N, 0, 0.0f, 0, nullptr, blob, hidden, n_blocks, flags, 0.1, out, out2, save, stream,
                         nullptr, nullptr, c);
}

int vt_decode_mlp_bwd_wide(int B, int C, const float *pts, int64_t N, const float *blob_t, int hidden, int n_blocks, int flags,
                           const float *grad_out, const float *grad_out2, const float *save, float *gws, float *grad_c, void *stream) {
    if (!grad_c) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_bwd_wide: null argument");
    return wide_bwd_impl(B, 2, C, pts, N, blob_t, hidden, n_blocks, flags, 0.1, grad_out, grad_out2, save, gws, nullptr, nullptr, grad_c, stream);
}

}  // extern "C"
