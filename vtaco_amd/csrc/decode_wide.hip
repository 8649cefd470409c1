// LocalDecoder.forward / forward_img for the shapes the shipped kernels (decode.hip: hidden = c_dim = 32, relu) do not cover:
// hidden_size and c_dim any multiples of 32 up to 256 (the class defaults of the reference are 256 / 128,
// src/conv_onet/models/decoder.py:24-51), n_blocks up to VT_MAX_BLOCKS, and `leaky`: leaky_relu(0.2) in front of the output heads
// (decoder.py:46-49, 157; the ResnetBlockFC activations are ReLU whatever `leaky` says, layers.py:33).  Inference
// only, exact f32: v_mfma_f32_32x32x2_f32 with f32 operands, so the result is the f32 network's up to summation order.
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
};

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
__global__ void wide_pack_kernel(const float *w, int H, int K, int kin, int ld, float *dst) {
    const size_t total = (size_t)H * K;
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(f & 3), l = (int)((f >> 2) & 63);
        const size_t q = f >> 8;
        const int kq = (int)(q % (K / 8)), ob = (int)(q / (K / 8));
        const int row = 32 * ob + (l & 31), k = 8 * kq + 2 * e + (l >> 5);
        dst[f] = k < kin ? w[(size_t)row * ld + k] : 0.0f;
    }
}
__global__ void wide_copy_kernel(const float *src, float *dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] : 0.0f;
}

__device__ __forceinline__ float actvn(float x, int leaky) { return x > 0.0f ? x : (leaky ? 0.2f * x : 0.0f); }

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
                    if (a.nearest) { cl[(cb + ch) * WIDE_PITCH + pt] = gb[near + cb + ch]; continue; }
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
                }
                if (ch < 3) bufA[ch * WIDE_PITCH + pt] = ch == 0 ? px : (ch == 1 ? py : pz);
                for (int k = 3 + ch; k < Kp; k += 32)
                    bufA[k * WIDE_PITCH + pt] = (d.c_img && k < a.p_in) ? d.c_img[(size_t)g * (a.p_in - 3) + (k - 3)] : 0.0f;
            }
        }
        __syncthreads();
        // ---- fc_p (decoder.py:139 / 81) ----
        f32x16 net;
        const int ob = wave;                                        // nh <= WIDE_WAVES; waves beyond the width only keep the barriers
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
            }
            __syncthreads();
            if (ob < nh) {
                const f32x16 hid = wide_gemm(bias16(bb + H, ob, kg), wb + lay.w_c, ob, H, bufA, lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) bufB[(32 * ob + chan_of(i, kg)) * WIDE_PITCH + j] = actvn(hid[i], 0);
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
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 32 * ob + chan_of(i, kg);
                const float v = actvn(net[i], a.leaky);
                o1 = fmaf(ow[row], v, o1);
                o2 = fmaf(ow2[row], v, o2);
            }
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
        hipLaunchKernelGGL(wide_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, H, K, kin, ld, dst);
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

int vt_decode_fwd_wide(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                       int lattice_nx, float lattice_box, int64_t lattice_first,
                       const float *c_img, const float *blob, int hidden, int n_blocks, int flags, double padding,
                       float *out, float *out2, void *stream) {
    if (!grid_cl || !blob || !out) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_wide: null argument");
    const int p_in = c_img ? 3 + C : 3;
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
    a.d.N = (uint32_t)N; a.d.total = (uint32_t)((uint64_t)B * (uint64_t)N); a.d.lattice_first = (uint32_t)lattice_first;
    a.d.R = R; a.d.nx = lattice_nx; a.d.box = lattice_box; a.d.divisor = (float)(1.0 + padding + 10e-4);
    a.blob = blob; a.H = hidden; a.C = C; a.nb = n_blocks; a.p_in = p_in; a.Kp = (p_in + 7) / 8 * 8; a.leaky = (flags & VT_WIDE_LEAKY) ? 1 : 0; a.nearest = (flags & VT_WIDE_NEAREST) ? 1 : 0;
    const int rowsA = hidden > a.Kp ? hidden : a.Kp;
    const int waves = hidden <= 128 ? 4 : 8;
    const size_t lds = ((size_t)(C + rowsA + hidden) * WIDE_PITCH + (size_t)waves * 2 * 2 * 32) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_wide_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_wide_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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

}  // extern "C"
