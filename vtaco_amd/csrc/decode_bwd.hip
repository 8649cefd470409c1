// Backward of the fused decode (training): what PyTorch autograd computes for
// LocalDecoder.forward / forward_img in the reference (triggered at
// src/conv_onet/training.py:79,89,96).
//
// Three launches, no float atomics on the parameter gradients:
//   decode_bwd_data_kernel   one wave = 32 points.  Walks the MLP backwards with the
//       TRANSPOSED weights on the f32 matrix core (same accumulator-chaining trick as the
//       forward: W^T g lands with the point on the lane), using the relu masks of the
//       activations the training forward saved.  Leaves every layer's output-side gradient
//       in a workspace, writes d c_img, and scatters d c into the channels-last grid
//       gradient with the 8 trilinear weights (f32 atomics: the only cross-point sum here).
//   decode_wgrad_kernel      dW = G^T X per dense layer as MFMA outer products over chunks
//       of 1024 points, read straight from the [point][channel] workspaces (256-B coalesced
//       rows); bias gradients are the column sums of G.  One partial per (layer, chunk).
//   decode_wgrad_reduce_kernel   sums the partials in chunk order (bit-reproducible) into
//       the nn.Linear [out][in] layout.
#include "decode_common.h"

namespace {

// ---- transposed-weight blob (floats) ---------------------------------------------------
// 16 layers x [16 k-steps][64 lanes]:  0..4 fc_c{i}^T (output in gather layout)
//                                      5..9 fc_1{i}^T, 10..14 fc_0{i}^T (accumulator layout)
//                                      15   fc_p_img[:,3:]^T (gather layout)
// then fc_out.weight fragment [2][16] and fc_out_contact.weight fragment [2][16] (accumulator layout; zeros without a contact head)
constexpr int VT_OFFT_OUT = 16 * 1024;
constexpr int VT_OFFT_OUT2 = VT_OFFT_OUT + 32;
constexpr int VT_BLOBT_FLOATS = VT_OFFT_OUT2 + 32;

struct PackArgs {
    vt_decoder_params p;
    float *blob;
};

__global__ void decoder_pack_t_kernel(PackArgs a) {
    const vt_decoder_params &p = a.p;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < VT_BLOBT_FLOATS; e += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (e < VT_OFFT_OUT) {
            const int L = e >> 10, s = (e >> 6) & 15, l = e & 63, i = l & 31, h = l >> 5;
            const int k = chan_of(s, h);                     // output channel the k-step contracts over
            // D row i represents: accumulator layout -> channel i; gather layout -> 16*hh + r
            const int hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
            const int gather_row = 16 * hh + r;
            if (L < 5) v = p.fc_c_w[L][k * 32 + gather_row];
            else if (L < 10) v = p.fc1_w[L - 5][k * 32 + i];
            else if (L < 15) v = p.fc0_w[L - 10][k * 32 + i];
            else v = (p.p_in > 3) ? p.fc_p_w[k * p.p_in + 3 + gather_row] : 0.0f;
        } else if (e < VT_OFFT_OUT2) {
            const int q = e - VT_OFFT_OUT;
            v = p.fc_out_w[chan_of(q & 15, q >> 4)];
        } else {
            const int q = e - VT_OFFT_OUT2;
            v = p.fc_out2_w ? p.fc_out2_w[chan_of(q & 15, q >> 4)] : 0.0f;
        }
        a.blob[e] = v;
    }
}

struct BwdArgs {
    DecodeArgs d;            // grid (unused), pts / lattice, c_img (only its presence), N, total, R, ...
    const float *blobT;
    const float *grad_out;   // [total]
    const float *grad_out2;  // [total] gradient of the contact head's logits (forward_contact), or null
    const float *save;       // [12][total][32]
    float *gws;              // [11][total][32]
    float *grad_grid;        // [B,R,R,R,32] channels-last, accumulated
    float *grad_c_img;       // [total][32] or null
    float *grad_c;           // [total][32] or null: d(sampled features), for callers that transform them
};

template <int THREADS>
__global__ void __launch_bounds__(THREADS, 2)
decode_bwd_data_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.blobT);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = threadIdx.x; i < VT_BLOBT_FLOATS / 4; i += THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, pl = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    constexpr int WPB = THREADS / 64;
    const uint32_t total = a.d.total;
    const uint32_t ntiles = (total + 31u) >> 5;
    const size_t slot = (size_t)total * 32;
    const int R = a.d.R;

    for (uint32_t tile = blockIdx.x * WPB + wave; tile < ntiles; tile += gridDim.x * WPB) {
        unsigned lds_off = 0;
        asm volatile("" : "+v"(lds_off));
        const float *L = lds + lds_off;
        uint32_t g = tile * 32u + pl;
        const bool live = g < total;
        if (!live) g = total - 1u;
        const float go = live ? a.grad_out[g] : 0.0f;
        const float *srow = a.save + (size_t)g * 32;
        float *grow = a.gws + (size_t)g * 32;

        // d net_5 = (go * fc_out.weight + go2 * fc_out_contact.weight) (.) relu'(net_5)
        f32x16 G;
        {
            const f32x16 wo = load_frag16(L + VT_OFFT_OUT + h * 16);
            const f32x16 rn = load_acc16(srow + 11 * slot, h);
            if (a.grad_out2) {
                const float go2 = live ? a.grad_out2[g] : 0.0f;
                const f32x16 wo2 = load_frag16(L + VT_OFFT_OUT2 + h * 16);
#pragma unroll
                for (int s = 0; s < 16; ++s) G[s] = (rn[s] > 0.0f) ? fmaf(go2, wo2[s], go * wo[s]) : 0.0f;
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) G[s] = (rn[s] > 0.0f) ? go * wo[s] : 0.0f;
            }
        }
        f32x16 dc;
#pragma unroll
        for (int s = 0; s < 16; ++s) dc[s] = 0.0f;

#pragma unroll 1
        for (int i = 4; i >= 0; --i) {
            if (live) store_acc16(grow + (6 + i) * slot, G, h);              // d block_i output
            if (i < 4) dc = dense32<false>(dc, L + (i + 1) * 1024, G, lane);  // += fc_c{i+1}^T G
            f32x16 t;
#pragma unroll
            for (int s = 0; s < 16; ++s) t[s] = 0.0f;
            t = dense32<false>(t, L + (5 + i) * 1024, G, lane);               // fc_1^T G
            {
                const f32x16 rh = load_acc16(srow + (6 + i) * slot, h);
#pragma unroll
                for (int s = 0; s < 16; ++s) t[s] = (rh[s] > 0.0f) ? t[s] : 0.0f;
            }
            if (live) store_acc16(grow + (1 + i) * slot, t, h);              // d h_i
            f32x16 u;
#pragma unroll
            for (int s = 0; s < 16; ++s) u[s] = 0.0f;
            u = dense32<false>(u, L + (10 + i) * 1024, t, lane);              // fc_0^T d h
            {
                const f32x16 rx = load_acc16(srow + (1 + i) * slot, h);
#pragma unroll
                for (int s = 0; s < 16; ++s) G[s] = G[s] + ((rx[s] > 0.0f) ? u[s] : 0.0f);
            }
        }
        if (live) store_acc16(grow, G, h);                                   // d x_0
        dc = dense32<false>(dc, L, G, lane);                                  // += fc_c0^T d x_0
        if (a.grad_c_img) {
            f32x16 di;
#pragma unroll
            for (int s = 0; s < 16; ++s) di[s] = 0.0f;
            di = dense32<false>(di, L + 15 * 1024, G, lane);
            if (live) store_gather16(a.grad_c_img + (size_t)g * 32, di, h);
        }

        if (live && a.grad_c) store_gather16(a.grad_c + (size_t)g * 32, dc, h);
        // ---- d grid: scatter d c (channels 16h..16h+15 on this lane) to the 8 corners ----
        if (live && a.grad_grid) {
            const uint32_t b = g / a.d.N, n = g - b * a.d.N;
            float px, py, pz;
            point_of(a.d, g, n, px, py, pz);
            const Tri t = tri_setup(px, py, pz, a.d.divisor, R);
            float *gb = a.grad_grid + (size_t)b * R * R * R * 32 + 16 * h;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int zz = (k & 4) ? t.z1 : t.z0, yy = (k & 2) ? t.y1 : t.y0, xx = (k & 1) ? t.x1 : t.x0;
                const float w = (((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0)) * ((k & 4) ? t.wz1 : t.wz0);
                if (w != 0.0f) {
                    float *dst = gb + (((size_t)zz * R + yy) * R + xx) * 32;
#pragma unroll
                    for (int s = 0; s < 16; ++s) atomicAdd(dst + s, w * dc[s]);
                }
            }
        }
    }
}

// backward of vt_sample_grid alone: scatter d feat [total][C] to the 8 corners of every point (blockIdx.y = the 32-channel slice)
__global__ void __launch_bounds__(256)
sample_grid_bwd_kernel(DecodeArgs d, const float *grad_feat, float *grad_grid, int C) {
    const int lane = threadIdx.x & 63, pl = lane & 31, h = lane >> 5;
    grad_feat += 32 * blockIdx.y; grad_grid += 32 * blockIdx.y;
    const uint32_t ntiles = (d.total + 31u) >> 5;
    const int R = d.R;
    for (uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += gridDim.x * 4) {
        const uint32_t g = tile * 32u + pl;
        if (g >= d.total) continue;
        const uint32_t b = g / d.N, n = g - b * d.N;
        float px, py, pz;
        point_of(d, g, n, px, py, pz);
        const Tri t = tri_setup(px, py, pz, d.divisor, R);
        const f32x16 dc = load_frag16(grad_feat + (size_t)g * C + 16 * h);
        float *gb = grad_grid + (size_t)b * R * R * R * C + 16 * h;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int zz = (k & 4) ? t.z1 : t.z0, yy = (k & 2) ? t.y1 : t.y0, xx = (k & 1) ? t.x1 : t.x0;
            const float w = (((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0)) * ((k & 4) ? t.wz1 : t.wz0);
            if (w != 0.0f) {
                float *dst = gb + (((size_t)zz * R + yy) * R + xx) * C;
#pragma unroll
                for (int s = 0; s < 16; ++s) atomicAdd(dst + s, w * dc[s]);
            }
        }
    }
}


// The same scatter with the points grouped by trilinear cell first (vt_voxel_build at resolution R - 1 bins a point by
// floor(p_nor (R - 1)) = the cell whose 8 corners it touches): one wave per cell sums its points' contributions to the 8 corners
// in registers -- ascending point order -- and issues 8 x 32 atomics per CELL instead of per point.  Training samples are not
// spread evenly: the contact clouds put up to 128 points of a scene within millimetres, i.e. into a handful of cells, and f32
// atomics that collide on one address retire one after the other (decode_bwd_data_kernel's own scatter: 0.92 ms of a step's
// 16 384 points against 0.1 ms for everything else it does).  A point whose own cell differs from its bin's (the two
// normalisations round differently on a cell boundary) falls back to its own 8 x 32 atomics: same sum either way.
__global__ void __launch_bounds__(256)
sample_grid_bwd_sorted_kernel(DecodeArgs d, const float *grad_feat, const int *order, const int *seg_lo, const int *seg_hi, float *grad_grid, int C) {
    const int lane = threadIdx.x & 63, ch = (lane & 31) + 32 * blockIdx.y, half = lane >> 5;         // blockIdx.y = the 32-channel slice
    const int R = d.R;
    for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < d.total; g += gridDim.x * 4) {
        const uint32_t b = g / d.N, t = g - b * d.N;
        const int lo = seg_lo[g], hi = seg_hi[g];
        const int *ord = order + (size_t)b * d.N;
        if ((uint32_t)ord[lo] != t) continue;                       // not the first point of its cell: the head's wave does the cell
        float px, py, pz;
        point_of(d, g, t, px, py, pz);
        const Tri head = tri_setup(px, py, pz, d.divisor, R);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
        float *gb = grad_grid + (size_t)b * R * R * R * C + ch;
        for (int j = lo + half; j < hi; j += 2) {
            const uint32_t n = (uint32_t)ord[j], gn = b * d.N + n;
            point_of(d, gn, n, px, py, pz);
            const Tri tr = tri_setup(px, py, pz, d.divisor, R);
            const float v = grad_feat[(size_t)gn * C + ch];
            const bool same = tr.x0 == head.x0 && tr.y0 == head.y0 && tr.z0 == head.z0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = (((k & 1) ? tr.wx1 : tr.wx0) * ((k & 2) ? tr.wy1 : tr.wy0)) * ((k & 4) ? tr.wz1 : tr.wz0);
                if (same) acc[k] = fmaf(w, v, acc[k]);
                else if (w != 0.0f) {
                    const int zz = (k & 4) ? tr.z1 : tr.z0, yy = (k & 2) ? tr.y1 : tr.y0, xx = (k & 1) ? tr.x1 : tr.x0;
                    atomicAdd(gb + (((size_t)zz * R + yy) * R + xx) * C, w * v);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += __shfl_xor(acc[k], 32);       // even + odd positions of the cell
        if (half == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (acc[k] == 0.0f) continue;
                const int zz = (k & 4) ? head.z1 : head.z0, yy = (k & 2) ? head.y1 : head.y0, xx = (k & 1) ? head.x1 : head.x0;
                atomicAdd(gb + (((size_t)zz * R + yy) * R + xx) * C, acc[k]);
            }
        }
    }
}

// ---- weight gradients ----------------------------------------------------------------------
// jobs: 0..4 fc_c{i} (G: i==0 ? gws0 : gws[6+i-1]; X: save0)    5..9 fc_0{i} (gws[1+i], save[1+i])
//       10..14 fc_1{i} (gws[6+i], save[6+i])    15 fc_p[:, :3] (gws0, pts)    16 fc_p_img[:, 3:] (gws0, c_img)
//       17 fc_out (grad_out as row 0, the contact head's grad_out2 as row 1, save[11])
constexpr int N_JOBS = 18;
constexpr int CHUNK = 1024;
constexpr int PART = 1024 + 32;       // D[16 regs][64 lanes] + column sums of G

struct WgradArgs {
    const float *save, *gws, *grad_out, *grad_out2, *pts, *c_img;
    float *partial;          // [N_JOBS][nchunks][PART]
    DecodeArgs d;            // for lattice-mode points
    uint32_t total, nchunks;
};

__global__ void __launch_bounds__(256)
decode_wgrad_kernel(WgradArgs a) {
    __shared__ float red[4][PART];
    const int job = blockIdx.y;
    const uint32_t chunk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kk = lane >> 5;
    const size_t slot = (size_t)a.total * 32;
    const float *G = nullptr, *X = nullptr;
    if (job < 5) { G = a.gws + (job == 0 ? 0 : (6 + job - 1)) * slot; X = a.save; }
    else if (job < 10) { G = a.gws + (1 + job - 5) * slot; X = a.save + (1 + job - 5) * slot; }
    else if (job < 15) { G = a.gws + (6 + job - 10) * slot; X = a.save + (6 + job - 10) * slot; }
    else if (job == 15) { G = a.gws; }
    else if (job == 16) { G = a.gws; X = a.c_img; }
    else { X = a.save + 11 * slot; }
    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0f;
    float csum = 0.0f;
    const bool skip = (job == 16 && !a.c_img);
    const uint32_t p0 = chunk * CHUNK + wave * (CHUNK / 4);
    const uint32_t p1 = min(p0 + CHUNK / 4, a.total);
    if (!skip) {
        for (uint32_t p = p0 + kk; p < p1 + kk; p += 2) {     // each MFMA contracts two points
            const bool ok = p < p1;
            float gv, xv;
            if (job == 17) gv = !ok ? 0.0f : (col == 0 ? a.grad_out[p] : ((col == 1 && a.grad_out2) ? a.grad_out2[p] : 0.0f));
            else gv = ok ? G[(size_t)p * 32 + col] : 0.0f;
            if (job == 15) {
                xv = 0.0f;
                if (ok && col < 3) {
                    if (a.pts) xv = a.pts[(size_t)p * 3 + col];
                    else {
                        float px, py, pz;
                        point_of(a.d, p, p % a.d.N, px, py, pz);
                        xv = col == 0 ? px : (col == 1 ? py : pz);
                    }
                }
            } else xv = ok ? X[(size_t)p * 32 + col] : 0.0f;
            csum += gv;
            acc = mfma(gv, xv, acc);                            // D[out][in] += G[p][out] * X[p][in]
        }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) red[wave][s * 64 + lane] = acc[s];
    csum += __shfl_xor(csum, 32);
    if (lane < 32) red[wave][1024 + lane] = csum;
    __syncthreads();
    float *dst = a.partial + ((size_t)job * a.nchunks + chunk) * PART;
    for (int e = threadIdx.x; e < PART; e += 256) dst[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

struct ReduceArgs {
    const float *partial;
    float *out;              // flat parameter gradients, see vt_decode_wgrad
    uint32_t nchunks;
    int p_in;
    int contact;             // 1: the flat buffer ends with fc_out_contact.w [32] | fc_out_contact.b [1]
};

// flat layout: fc_p.w [32*p_in] | fc_p.b [32] | fc_c.w [5][1024] | fc_c.b [5][32] | fc_0.w [5][1024] | fc_0.b [5][32]
//              | fc_1.w [5][1024] | fc_1.b [5][32] | fc_out.w [32] | fc_out.b [1]
__global__ void __launch_bounds__(256)
decode_wgrad_reduce_kernel(ReduceArgs a) {
    const int job = blockIdx.x;
    const int off_pw = 0, off_pb = 32 * a.p_in;
    const int off_cw = off_pb + 32, off_cb = off_cw + 5 * 1024;
    const int off_0w = off_cb + 5 * 32, off_0b = off_0w + 5 * 1024;
    const int off_1w = off_0b + 5 * 32, off_1b = off_1w + 5 * 1024;
    const int off_ow = off_1b + 5 * 32, off_ob = off_ow + 32, off_ow2 = off_ob + 1, off_ob2 = off_ow2 + 32;
    for (int e = threadIdx.x; e < PART; e += 256) {
        float s = 0.0f;
        const float *src = a.partial + (size_t)job * a.nchunks * PART + e;
        for (uint32_t c = 0; c < a.nchunks; ++c) s += src[(size_t)c * PART];
        if (e < 1024) {
            const int r = e >> 6, l = e & 63, in = l & 31, out = chan_of(r, l >> 5);
            if (job < 5) a.out[off_cw + job * 1024 + out * 32 + in] = s;
            else if (job < 10) a.out[off_0w + (job - 5) * 1024 + out * 32 + in] = s;
            else if (job < 15) a.out[off_1w + (job - 10) * 1024 + out * 32 + in] = s;
            else if (job == 15) { if (in < 3) a.out[off_pw + out * a.p_in + in] = s; }
            else if (job == 16) { if (a.p_in > 3) a.out[off_pw + out * a.p_in + 3 + in] = s; }
            else { if (out == 0) a.out[off_ow + in] = s; else if (out == 1 && a.contact) a.out[off_ow2 + in] = s; }
        } else {
            const int o = e - 1024;
            if (job == 0) { a.out[off_cb + o] = s; a.out[off_pb + o] = s; }
            else if (job < 5) a.out[off_cb + job * 32 + o] = s;
            else if (job < 10) a.out[off_0b + (job - 5) * 32 + o] = s;
            else if (job < 15) a.out[off_1b + (job - 10) * 32 + o] = s;
            else if (job == 17 && o == 0) a.out[off_ob] = s;
            else if (job == 17 && o == 1 && a.contact) a.out[off_ob2] = s;
        }
    }
}

bool fill_decode_args(DecodeArgs &d, int B, int R, int C, const float *pts, int64_t N, int nx, float box,
                      int64_t first, double padding, const char *who, int &rc, bool any_width = false) {
    rc = 0;
    if (B <= 0 || R < 2 || N <= 0) { rc = vt_fail(VT_ERR_INVALID, who); return false; }
    if (any_width ? (C <= 0 || (C & 31)) : C != 32) { rc = vt_fail(VT_ERR_UNSUPPORTED, any_width ? "c_dim must be a multiple of 32" : "c_dim must be 32"); return false; }
    if ((int64_t)B * N >= (int64_t)1 << 31) { rc = vt_fail(VT_ERR_UNSUPPORTED, "B*N must be < 2^31"); return false; }
    if (!pts && nx < 2) { rc = vt_fail(VT_ERR_INVALID, "lattice mode needs nx >= 2"); return false; }
    d.c_direct = nullptr; d.brick = 0; d.cimg_ids = nullptr; d.cimg_table = nullptr; d.cimg_nf = 0; d.grid = nullptr; d.pts = pts; d.c_img = nullptr; d.blob = nullptr; d.out = nullptr; d.out2 = nullptr; d.save = nullptr;
    d.N = (uint32_t)N; d.total = (uint32_t)((int64_t)B * N); d.lattice_first = (uint32_t)first;
    d.R = R; d.nx = nx; d.box = box; d.divisor = (float)(1.0 + padding + 10e-4);
    return true;
}

}  // namespace

extern "C" {

size_t vt_decoder_blob_t_bytes(int hidden, int c_dim, int n_blocks) {
    if (hidden != 32 || c_dim != 32 || n_blocks != 5) return 0;
    return (size_t)VT_BLOBT_FLOATS * sizeof(float);
}

int vt_decoder_pack_t(const vt_decoder_params *p, float *blob, size_t blob_bytes, void *stream) {
    if (!p || !blob) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack_t: null argument");
    if (p->hidden != 32 || p->c_dim != 32 || p->n_blocks != 5)
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decoder_pack_t: only hidden=32, c_dim=32, n_blocks=5 are built");
    if (blob_bytes < (size_t)VT_BLOBT_FLOATS * sizeof(float)) return vt_fail(VT_ERR_WORKSPACE, "vt_decoder_pack_t: blob too small");
    PackArgs a;
    a.p = *p;
    a.blob = blob;
    hipLaunchKernelGGL(decoder_pack_t_kernel, dim3(17), dim3(1024), 0, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decoder_pack_t");
}

size_t vt_decode_save_bytes(int64_t total_points) { return (size_t)VT_SAVE_SLOTS * (size_t)total_points * 32 * sizeof(float); }
size_t vt_decode_gws_bytes(int64_t total_points) { return (size_t)VT_GWS_SLOTS * (size_t)total_points * 32 * sizeof(float); }

static int decode_bwd_launch(BwdArgs &a, void *stream) {
    constexpr int THREADS = 512;
    const int64_t ntiles = ((int64_t)a.d.total + 31) / 32;
    int64_t blocks = (ntiles + THREADS / 64 - 1) / (THREADS / 64);
    const int64_t cap = 2 * vt_num_cus();
    if (blocks > cap) blocks = cap;
    const size_t lds_bytes = (size_t)VT_BLOBT_FLOATS * sizeof(float);
    bool attr_set = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr_set) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_bwd_data_kernel<THREADS>), (int)lds_bytes);
        if (e != hipSuccess) return vt_check(e, "vt_decode_bwd: hipFuncSetAttribute");
        attr_set = true;
    }
    hipLaunchKernelGGL(decode_bwd_data_kernel<THREADS>, dim3((unsigned)blocks), dim3(THREADS), lds_bytes, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decode_bwd");
}

int vt_decode_bwd(int B, int R, int C, const float *pts, int64_t N,
                  int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                  const float *blob_t, const float *grad_out, const float *save, float *gws,
                  float *grad_grid_cl, float *grad_c_img, void *stream) {
    return vt_decode_bwd_contact(B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, padding, blob_t, grad_out, nullptr, save, gws,
                                 grad_grid_cl, grad_c_img, stream);
}

int vt_decode_bwd_contact(int B, int R, int C, const float *pts, int64_t N,
                          int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                          const float *blob_t, const float *grad_out, const float *grad_out2, const float *save, float *gws,
                          float *grad_grid_cl, float *grad_c_img, void *stream) {
    if (!blob_t || !grad_out || !save || !gws) return vt_fail(VT_ERR_INVALID, "vt_decode_bwd: null argument");
    BwdArgs a;
    int rc;
    if (!fill_decode_args(a.d, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, padding, "vt_decode_bwd: bad size", rc)) return rc;
    a.blobT = blob_t; a.grad_out = grad_out; a.grad_out2 = grad_out2; a.save = save; a.gws = gws; a.grad_grid = grad_grid_cl; a.grad_c_img = grad_c_img;
    a.grad_c = nullptr;
    return decode_bwd_launch(a, stream);
}

int vt_decode_bwd_dc(int B, int R, int C, const float *pts, int64_t N, double padding,
                     const float *blob_t, const float *grad_out, const float *grad_out2, const float *save, float *gws,
                     float *grad_c, float *grad_c_img, void *stream) {
    if (!blob_t || !grad_out || !save || !gws || !grad_c || !pts) return vt_fail(VT_ERR_INVALID, "vt_decode_bwd_dc: null argument");
    BwdArgs a;
    int rc;
    if (!fill_decode_args(a.d, B, R, C, pts, N, 0, 0.0f, 0, padding, "vt_decode_bwd_dc: bad size", rc)) return rc;
    a.blobT = blob_t; a.grad_out = grad_out; a.grad_out2 = grad_out2; a.save = save; a.gws = gws; a.grad_grid = nullptr; a.grad_c_img = grad_c_img;
    a.grad_c = grad_c;
    return decode_bwd_launch(a, stream);
}

int vt_decode_mlp_bwd(int B, int C, const float *pts, int64_t N,
                      int lattice_nx, float lattice_box, int64_t lattice_first,
                      const float *blob_t, const float *grad_out, const float *save, float *gws,
                      float *grad_c, void *stream) {
    if (!blob_t || !grad_out || !save || !gws || !grad_c) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_bwd: null argument");
    BwdArgs a;
    int rc;
    if (!fill_decode_args(a.d, B, 2, C, pts, N, lattice_nx, lattice_box, lattice_first, 0.1, "vt_decode_mlp_bwd: bad size", rc)) return rc;
    a.blobT = blob_t; a.grad_out = grad_out; a.grad_out2 = nullptr; a.save = save; a.gws = gws; a.grad_grid = nullptr; a.grad_c_img = nullptr;
    a.grad_c = grad_c;
    return decode_bwd_launch(a, stream);
}

int vt_sample_grid_bwd(int B, int R, int C, const float *pts, int64_t N,
                       int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                       const float *grad_feat, float *grad_grid_cl, void *stream) {
    if (!grad_feat || !grad_grid_cl) return vt_fail(VT_ERR_INVALID, "vt_sample_grid_bwd: null argument");
    DecodeArgs d;
    int rc;
    if (!fill_decode_args(d, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, padding, "vt_sample_grid_bwd: bad size", rc, true)) return rc;
    int64_t blocks = (((int64_t)d.total + 31) / 32 + 3) / 4;
    const int64_t cap = 8 * vt_num_cus();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sample_grid_bwd_kernel, dim3((unsigned)blocks, (unsigned)(C / 32)), dim3(256), 0, (hipStream_t)stream, d, grad_feat, grad_grid_cl, C);
    return vt_check(hipGetLastError(), "vt_sample_grid_bwd");
}

int vt_sample_grid_bwd_sorted(int B, int R, int C, const float *pts, int64_t N, double padding, const float *grad_feat,
                              const int *order, const int *seg_lo, const int *seg_hi, float *grad_grid_cl, void *stream) {
    if (!grad_feat || !grad_grid_cl || !pts || !order || !seg_lo || !seg_hi) return vt_fail(VT_ERR_INVALID, "vt_sample_grid_bwd_sorted: null argument");
    DecodeArgs d;
    int rc;
    if (!fill_decode_args(d, B, R, C, pts, N, 0, 0.0f, 0, padding, "vt_sample_grid_bwd_sorted: bad size", rc, true)) return rc;
    int64_t blocks = ((int64_t)d.total + 3) / 4;
    const int64_t cap = 32 * vt_num_cus();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sample_grid_bwd_sorted_kernel, dim3((unsigned)blocks, (unsigned)(C / 32)), dim3(256), 0, (hipStream_t)stream, d, grad_feat, order, seg_lo, seg_hi, grad_grid_cl, C);
    return vt_check(hipGetLastError(), "vt_sample_grid_bwd_sorted");
}

size_t vt_decode_wgrad_workspace_bytes(int64_t total_points) {
    const size_t nchunks = (size_t)((total_points + CHUNK - 1) / CHUNK);
    return (size_t)N_JOBS * nchunks * PART * sizeof(float);
}

size_t vt_decode_wgrad_floats(int p_in) { return (size_t)32 * p_in + 32 + 3 * (5 * 1024 + 5 * 32) + 32 + 1; }
size_t vt_decode_wgrad_floats_contact(int p_in) { return vt_decode_wgrad_floats(p_in) + 32 + 1; }

int vt_decode_wgrad(int B, const float *pts, int64_t N, int lattice_nx, float lattice_box, int64_t lattice_first,
                    const float *c_img, const float *grad_out, const float *save, const float *gws,
                    void *workspace, size_t workspace_bytes, float *grads, void *stream) {
    return vt_decode_wgrad_contact(B, pts, N, lattice_nx, lattice_box, lattice_first, c_img, grad_out, nullptr, save, gws,
                                   workspace, workspace_bytes, grads, stream);
}

int vt_decode_wgrad_contact(int B, const float *pts, int64_t N, int lattice_nx, float lattice_box, int64_t lattice_first,
                            const float *c_img, const float *grad_out, const float *grad_out2, const float *save, const float *gws,
                            void *workspace, size_t workspace_bytes, float *grads, void *stream) {
    if (!grad_out || !save || !gws || !workspace || !grads) return vt_fail(VT_ERR_INVALID, "vt_decode_wgrad: null argument");
    WgradArgs a;
    int rc;
    if (!fill_decode_args(a.d, B, 2, 32, pts, N, lattice_nx, lattice_box, lattice_first, 0.1, "vt_decode_wgrad: bad size", rc)) return rc;
    a.total = a.d.total;
    a.nchunks = (a.total + CHUNK - 1) / CHUNK;
    if (workspace_bytes < (size_t)N_JOBS * a.nchunks * PART * sizeof(float)) return vt_fail(VT_ERR_WORKSPACE, "vt_decode_wgrad: workspace too small");
    a.save = save; a.gws = gws; a.grad_out = grad_out; a.grad_out2 = grad_out2; a.pts = pts; a.c_img = c_img; a.partial = (float *)workspace;
    hipLaunchKernelGGL(decode_wgrad_kernel, dim3(a.nchunks, N_JOBS), dim3(256), 0, (hipStream_t)stream, a);
    ReduceArgs r;
    r.partial = a.partial; r.out = grads; r.nchunks = a.nchunks; r.p_in = c_img ? 35 : 3; r.contact = grad_out2 ? 1 : 0;
    hipLaunchKernelGGL(decode_wgrad_reduce_kernel, dim3(N_JOBS), dim3(256), 0, (hipStream_t)stream, r);
    return vt_check(hipGetLastError(), "vt_decode_wgrad");
}

}  // extern "C"
