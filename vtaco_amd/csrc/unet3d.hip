// 3-D U-Net building blocks for gfx950, inference forward (SURVEY.md section 8f "next" row 1:
// the UNet3D that refines the scattered feature grid, reference src/encoder/unet3d.py:449-474,
// is 140 GFLOP/scene -- 2x the decoder at 128^3 -- and PyTorch-ROCm/MIOpen runs its f32 conv3d
// on a naive kernel here: 12.6 ms/scene).
//
// Everything is channels-last [B, D, H, W, C] f32, which is also the layout the decode kernel
// samples, so the encoder's output needs no transpose.
//   gn_partial/gn_finalize   GroupNorm statistics -> per (scene, channel) scale/shift
//   conv3d_gcr_kernel        'gcr' SingleConv = GroupNorm -> Conv3d(3x3x3, pad 1, no bias) -> ReLU
//                            as an implicit GEMM on the f32 matrix core: D[cout][voxel] +=
//                            W[cout][tap,cin] * Xn[tap,cin][voxel]; the normalised input tile
//                            (+1 halo, zero outside the volume, as the reference pads AFTER the
//                            norm) is staged in LDS 32 channels at a time; weights are read from a
//                            fragment-ordered copy (256-B coalesced, L1/L2 resident).  The input may
//                            be the virtual concat [skip | nearest-upsampled low] of a decoder
//                            level: neither the upsample nor the concat is materialised.
//   maxpool / conv1x1        the 2x2x2 max-pool between encoder levels and the final 1x1x1 conv.
#include "decode_common.h"

namespace {

constexpr int CPAD = 33;            // LDS row: 32 channels + 1 pad -> conflict-free ds_read_b32

struct Src {
    const float *skip;   // [B,D,H,W,C1]
    const float *low;    // [B,D/2,H/2,W/2,C2] or null (nearest-upsampled, concatenated after skip)
    int C1, C2;
    int D, H, W;
};

__device__ __forceinline__ float src_at(const Src &s, int b, int z, int y, int x, int c) {
    if (c < s.C1) return s.skip[((((size_t)b * s.D + z) * s.H + y) * s.W + x) * s.C1 + c];
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    return s.low[((((size_t)b * D2 + (z >> 1)) * H2 + (y >> 1)) * W2 + (x >> 1)) * s.C2 + (c - s.C1)];
}

// ---- GroupNorm statistics ---------------------------------------------------------------------
// partial[chunk][b][c] = (sum, sumsq) over the chunk's voxels, in double
__global__ void __launch_bounds__(256) gn_partial_kernel(Src s, int nchunks, double *partial) {
    __shared__ double red[8][32][2];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int C = s.C1 + s.C2;
    const size_t V = (size_t)s.D * s.H * s.W;
    const size_t v0 = V * chunk / nchunks, v1 = V * (chunk + 1) / nchunks;
    const int c = threadIdx.x & 31, vg = threadIdx.x >> 5;
    for (int cb = 0; cb < C; cb += 32) {
        double sum = 0.0, sq = 0.0;
        for (size_t v = v0 + vg; v < v1; v += 8) {
            const int x = (int)(v % s.W), y = (int)((v / s.W) % s.H), z = (int)(v / ((size_t)s.W * s.H));
            const double t = (double)src_at(s, b, z, y, x, cb + c);
            sum += t; sq += t * t;
        }
        red[vg][c][0] = sum; red[vg][c][1] = sq;
        __syncthreads();
        if (threadIdx.x < 32) {
            double a = 0.0, q = 0.0;
            for (int i = 0; i < 8; ++i) { a += red[i][c][0]; q += red[i][c][1]; }
            double *dst = partial + (((size_t)chunk * gridDim.y + b) * C + cb + c) * 2;
            dst[0] = a; dst[1] = q;
        }
        __syncthreads();
    }
}

// scale[b][c] = rstd*gamma, shift[b][c] = beta - mean*rstd*gamma  (biased variance, as torch)
__global__ void gn_finalize_kernel(const double *partial, int nchunks, int B, int C, int groups, double count,
                                   const float *gamma, const float *beta, float eps, float *scale_shift) {
    const int b = blockIdx.x, g = threadIdx.x;
    if (g >= groups) return;
    const int cpg = C / groups;
    double sum = 0.0, sq = 0.0;
    for (int ch = 0; ch < nchunks; ++ch)
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            const double *p = partial + (((size_t)ch * B + b) * C + c) * 2;
            sum += p[0]; sq += p[1];
        }
    const double n = count * cpg, mean = sum / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        const double sc = rstd * (double)gamma[c];
        scale_shift[((size_t)b * C + c) * 2 + 0] = (float)sc;
        scale_shift[((size_t)b * C + c) * 2 + 1] = (float)((double)beta[c] - mean * sc);
    }
}

// ---- weights -> fragment order [cin_blk][tap][co_blk][16 k-steps][64 lanes] -------------------
__global__ void conv3d_pack_kernel(const float *w, int Cout, int Cin, float *packed, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63), s = (int)((e >> 6) & 15);
        size_t r = e >> 10;
        const int nco = Cout / 32;
        const int cob = (int)(r % nco); r /= nco;
        const int tap = (int)(r % 27);
        const int cib = (int)(r / 27);
        const int co = cob * 32 + (l & 31), ci = cib * 32 + 2 * s + (l >> 5);
        packed[e] = w[((size_t)co * Cin + ci) * 27 + tap];          // torch [co][ci][kz][ky][kx]
    }
}

struct ConvArgs {
    Src s;
    const float *scale_shift;   // [B][Cin][2] or null (no norm)
    const float *wp;            // packed weights
    float *out;                 // [B,D,H,W,Cout]
    int Cout, relu;
    int TX, TY, TZ;             // block tile of output voxels (TX*TY*TZ = 32 * waves)
    int tiles_x, tiles_y, tiles_z;
};

// NCO = output-channel blocks (of 32) per workgroup; gridDim.y covers Cout / (32*NCO)
template <int NCO>
__global__ void __launch_bounds__(512, 2)
conv3d_gcr_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [(TZ+2)(TY+2)(TX+2)][CPAD]
    const Src &s = a.s;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, kk = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int tz = t % a.tiles_z;
    const int b = t / a.tiles_z;
    const int x0 = tx * a.TX, y0 = ty * a.TY, z0 = tz * a.TZ;
    const int PX = a.TX + 2, PY = a.TY + 2, PZ = a.TZ + 2;
    const int nvox = PX * PY * PZ;
    const int Cin = s.C1 + s.C2;
    const int co_blk0 = blockIdx.y * NCO, nco_all = a.Cout / 32;
    // this wave's 32 output voxels: rows of TX within the block tile
    const int rows = 32 / a.TX;                                  // y-rows per wave
    const int wy = wave * rows + j / a.TX;                       // row index inside the block tile (y, then z)
    const int lx = j % a.TX, ly = wy % a.TY, lz = wy / a.TY;
    const int center = ((lz + 1) * PY + (ly + 1)) * PX + (lx + 1);

    f32x16 acc[NCO];
#pragma unroll
    for (int n = 0; n < NCO; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    for (int cib = 0; cib < Cin / 32; ++cib) {
        __syncthreads();
        // stage the normalised input tile for channels [32 cib, 32 cib + 32)
        for (int e = threadIdx.x; e < nvox * 32; e += blockDim.x) {
            const int c = e & 31, v = e >> 5;
            const int px = v % PX, py = (v / PX) % PY, pz = v / (PX * PY);
            const int gx = x0 + px - 1, gy = y0 + py - 1, gz = z0 + pz - 1;
            float val = 0.0f;
            if (gx >= 0 && gx < s.W && gy >= 0 && gy < s.H && gz >= 0 && gz < s.D) {
                const int ch = cib * 32 + c;
                val = src_at(s, b, gz, gy, gx, ch);
                if (a.scale_shift) {
                    const float *ss = a.scale_shift + ((size_t)b * Cin + ch) * 2;
                    val = fmaf(val, ss[0], ss[1]);
                }
            }
            tile[v * CPAD + c] = val;
        }
        __syncthreads();
        const float *wc = a.wp + ((size_t)cib * 27 * nco_all) * 1024;
#pragma unroll 1
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
            const float *xin = tile + (center + (dz * PY + dy) * PX + dx) * CPAD + kk;
            const float *wt = wc + ((size_t)tap * nco_all + co_blk0) * 1024 + lane;
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const float bv = xin[2 * st];
#pragma unroll
                for (int n = 0; n < NCO; ++n) acc[n] = mfma(wt[n * 1024 + st * 64], bv, acc[n]);
            }
        }
    }
    // epilogue: ReLU, channels-last store (lane = voxel, 16 registers = channels chan_of(r,h))
    const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
    if (gx < s.W && gy < s.H && gz < s.D) {
        float *orow = a.out + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.Cout;
#pragma unroll
        for (int n = 0; n < NCO; ++n) {
            f32x16 v = acc[n];
            if (a.relu) v = relu16(v);
            store_acc16(orow + (co_blk0 + n) * 32, v, kk);
        }
    }
}

__global__ void __launch_bounds__(256)
maxpool3d_cl_kernel(const float *x, float *out, int D, int H, int W, int C, size_t total) {
    const int D2 = D / 2, H2 = H / 2, W2 = W / 2;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        size_t v = e / C;
        const int ox = (int)(v % W2); v /= W2;
        const int oy = (int)(v % H2); v /= H2;
        const int oz = (int)(v % D2);
        const size_t b = v / D2;
        float m = -INFINITY;
        for (int k = 0; k < 8; ++k) {
            const int z = 2 * oz + (k >> 2), y = 2 * oy + ((k >> 1) & 1), xx = 2 * ox + (k & 1);
            m = fmaxf(m, x[((((size_t)b * D + z) * H + y) * W + xx) * C + c]);
        }
        out[e] = m;
    }
}

// out[v][co] = bias[co] + sum_ci w[co][ci] x[v][ci]   (final 1x1x1 conv)
__global__ void __launch_bounds__(256)
conv1x1_cl_kernel(const float *x, const float *w, const float *bias, float *out, int Cin, int Cout, size_t V) {
    extern __shared__ float ws[];                              // w [Cout][Cin+1] (padded rows) + bias [Cout]
    const int ld = Cin + 1;
    for (int i = threadIdx.x; i < Cout * Cin; i += 256) ws[(i / Cin) * ld + i % Cin] = w[i];
    for (int i = threadIdx.x; i < Cout; i += 256) ws[Cout * ld + i] = bias ? bias[i] : 0.0f;
    __syncthreads();
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < V * Cout; e += (size_t)gridDim.x * 256) {
        const int co = (int)(e % Cout);
        const size_t v = e / Cout;
        const float *xr = x + v * Cin;
        float a = ws[Cout * ld + co];
        for (int ci = 0; ci < Cin; ++ci) a = fmaf(ws[co * ld + ci], xr[ci], a);
        out[e] = a;
    }
}

// scatter-mean straight into a channels-last grid (see voxel.hip for the segment bookkeeping)
__global__ void __launch_bounds__(256)
scatter_mean_cl_kernel(const float *feat, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                       float *grid, int T, int C, size_t V, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const size_t bt = e / C, b = bt / T;
        const int t = (int)(bt - b * T);
        const int lo = seg_lo[bt], hi = seg_hi[bt];
        const int *ord = order + b * T;
        if (ord[lo] != t) continue;
        const float *fb = feat + b * T * C;
        float sum = 0.0f;
        for (int q = lo; q < hi; ++q) sum += fb[(size_t)ord[q] * C + c];
        grid[(b * V + (size_t)idx[bt]) * C + c] = sum / (float)(hi - lo);
    }
}

bool src_ok(const Src &s, int B) {
    if (!s.skip || B <= 0 || s.D <= 0 || s.H <= 0 || s.W <= 0 || s.C1 <= 0 || (s.C1 & 31)) return false;
    if (s.low && (s.C2 <= 0 || (s.C2 & 31) || (s.D & 1) || (s.H & 1) || (s.W & 1))) return false;
    if (!s.low && s.C2 != 0) return false;
    return true;
}

}  // namespace

extern "C" {

size_t vt_conv3d_packed_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || (Cout & 31) || (Cin & 31)) return 0;
    return (size_t)27 * Cout * Cin;
}

int vt_conv3d_pack(const float *w, int Cout, int Cin, float *packed, void *stream) {
    if (!w || !packed) return vt_fail(VT_ERR_INVALID, "vt_conv3d_pack: null argument");
    const size_t total = vt_conv3d_packed_floats(Cout, Cin);
    if (!total) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_pack: channel counts must be multiples of 32");
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv3d_pack_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, packed, total);
    return vt_check(hipGetLastError(), "vt_conv3d_pack");
}

size_t vt_gn_workspace_bytes(int B, int C) { return (size_t)512 * B * C * 2 * sizeof(double); }

int vt_gn_scale_shift(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                      int groups, const float *gamma, const float *beta, double eps,
                      void *workspace, size_t workspace_bytes, float *scale_shift, void *stream) {
    Src s{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(s, B) || !gamma || !beta || !workspace || !scale_shift) return vt_fail(VT_ERR_INVALID, "vt_gn_scale_shift: bad argument");
    const int C = s.C1 + s.C2;
    if (groups <= 0 || groups > 64 || C % groups) return vt_fail(VT_ERR_INVALID, "vt_gn_scale_shift: bad group count");
    const size_t V = (size_t)D * H * W;
    int nchunks = (int)((V + 511) / 512);
    if (nchunks > 512) nchunks = 512;
    if (workspace_bytes < (size_t)nchunks * B * C * 2 * sizeof(double)) return vt_fail(VT_ERR_WORKSPACE, "vt_gn_scale_shift: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunks, B), dim3(256), 0, st, s, nchunks, (double *)workspace);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, st, (const double *)workspace, nchunks, B, C, groups,
                       (double)V, gamma, beta, (float)eps, scale_shift);
    return vt_check(hipGetLastError(), "vt_gn_scale_shift");
}

int vt_conv3d_gcr(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                  const float *scale_shift, const float *packed_w, int Cout, int relu, float *out, void *stream) {
    ConvArgs a;
    a.s = Src{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(a.s, B) || !packed_w || !out) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr: bad argument");
    if (Cout <= 0 || (Cout & 31)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr: Cout must be a multiple of 32");
    a.scale_shift = scale_shift; a.wp = packed_w; a.out = out; a.Cout = Cout; a.relu = relu;
    // block tile: 8 waves x 32 voxels; a wave covers rows of TX <= 32 voxels along x
    a.TX = W >= 32 ? 32 : (W >= 16 ? 16 : (W >= 8 ? 8 : 4));
    const int rows_total = 8 * (32 / a.TX);                       // y-rows (then z) per block
    a.TY = rows_total < H ? rows_total : H;
    while (rows_total % a.TY) --a.TY;                              // TY must divide the row count
    a.TZ = rows_total / a.TY;
    a.tiles_x = (W + a.TX - 1) / a.TX; a.tiles_y = (H + a.TY - 1) / a.TY; a.tiles_z = (D + a.TZ - 1) / a.TZ;
    const size_t lds = (size_t)(a.TX + 2) * (a.TY + 2) * (a.TZ + 2) * CPAD * sizeof(float);
    if (lds > 160 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr: tile does not fit LDS");
    const int nco = Cout / 32;
    const size_t spatial_blocks = (size_t)a.tiles_x * a.tiles_y * a.tiles_z * B;
    int per = nco;                                                 // cout blocks per workgroup: fewer when the volume is small
    while (per > 1 && (per > 4 || spatial_blocks * (nco / per) < 256)) per >>= 1;
    if (nco % per) per = 1;
    const dim3 grid((unsigned)spatial_blocks, (unsigned)(nco / per));
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3d_gcr_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3d_gcr_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3d_gcr_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr: hipFuncSetAttribute");
        attr_set = true;
    }
    if (per == 1) hipLaunchKernelGGL(conv3d_gcr_kernel<1>, grid, dim3(512), lds, st, a);
    else if (per == 2) hipLaunchKernelGGL(conv3d_gcr_kernel<2>, grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL(conv3d_gcr_kernel<4>, grid, dim3(512), lds, st, a);
    return vt_check(hipGetLastError(), "vt_conv3d_gcr");
}

int vt_maxpool3d_cl(const float *x, int B, int D, int H, int W, int C, float *out, void *stream) {
    if (!x || !out || B <= 0 || C <= 0 || D < 2 || H < 2 || W < 2) return vt_fail(VT_ERR_INVALID, "vt_maxpool3d_cl: bad argument");
    const size_t total = (size_t)B * (D / 2) * (H / 2) * (W / 2) * C;
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(maxpool3d_cl_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, out, D, H, W, C, total);
    return vt_check(hipGetLastError(), "vt_maxpool3d_cl");
}

int vt_conv1x1_cl(const float *x, int64_t V, int Cin, const float *w, const float *bias, int Cout, float *out, void *stream) {
    if (!x || !w || !out || V <= 0 || Cin <= 0 || Cout <= 0) return vt_fail(VT_ERR_INVALID, "vt_conv1x1_cl: bad argument");
    const size_t lds = ((size_t)Cout * (Cin + 1) + Cout) * sizeof(float);
    if (lds > 64 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv1x1_cl: weights do not fit 64 KiB of LDS");
    size_t g = ((size_t)V * Cout + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(conv1x1_cl_kernel, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream, x, w, bias, out, Cin, Cout, (size_t)V);
    return vt_check(hipGetLastError(), "vt_conv1x1_cl");
}

int vt_voxel_scatter_mean_cl_fwd(const float *feat, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                                 int B, int T, int C, int R, float *grid_cl, void *stream) {
    if (!feat || !idx || !order || !seg_lo || !seg_hi || !grid_cl) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_cl_fwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_cl_fwd: bad size");
    const size_t V = (size_t)R * R * R, total = (size_t)B * T * C;
    hipError_t e = hipMemsetAsync(grid_cl, 0, (size_t)B * C * V * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return vt_check(e, "vt_voxel_scatter_mean_cl_fwd: memset");
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(scatter_mean_cl_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       feat, idx, order, seg_lo, seg_hi, grid_cl, T, C, V, total);
    return vt_check(hipGetLastError(), "vt_voxel_scatter_mean_cl_fwd");
}

}  // extern "C"
