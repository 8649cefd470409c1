// 3-D U-Net building blocks for gfx950, inference forward (SURVEY.md section 8f "next" row 1:
// the UNet3D that refines the scattered feature grid, reference src/encoder/unet3d.py:449-474,
// is 140 GFLOP/scene -- 2x the decoder at 128^3 -- and PyTorch-ROCm/MIOpen runs its f32 conv3d
// on a naive kernel here: 12.6 ms/scene).
//
// Everything is channels-last [B, D, H, W, C] f32, which is also the layout the decode kernel
// samples, so the encoder's output needs no transpose.
//   channel_stats/gn_finalize GroupNorm statistics from producer-side partial sums -> scale/shift
//   conv3d_gcr_kernel        'gcr' SingleConv = GroupNorm -> Conv3d(3x3x3, pad 1, no bias) -> ReLU
//                            as an implicit GEMM on the f32 matrix core: D[cout][voxel] +=
//                            W[cout][tap,cin] * Xn[tap,cin][voxel]; the normalised input tile
//                            (+1 halo, zero outside the volume, as the reference pads AFTER the
//                            norm) is staged in LDS 32 channels at a time; weights are read from a
//                            fragment-ordered copy (256-B coalesced, L1/L2 resident).  The input may
//                            be the virtual concat [skip | nearest-upsampled low] of a decoder
//                            level: neither the upsample nor the concat is materialised.
//   maxpool / conv1x1        the 2x2x2 max-pool between encoder levels and the final 1x1x1 conv.
#include <string.h>
#include <type_traits>
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

// ---- GroupNorm statistics without a finalising launch ---------------------------------------------------------------
// vt_unet3d_fwd runs 14 normalised convolutions; a finalising launch in front of each (gn_finalize_kernel: ~5 us of
// dependent-launch latency around a microsecond of work) was 70 us of a 0.67 ms encoder.  Here the producer of a tensor
// adds its workgroup's (sum, sumsq) to a few accumulator rows (GnOut) and every workgroup of the consumer reduces those
// rows to the layer's scale / shift table in LDS during its prologue (GnIn) -- no launch, no ticket, no tail.
//   * The accumulators are integers: a workgroup's float sum t enters as trunc(t * 2^40) split into its low 32 bits and
//     the rest, each added to its own 64-bit cell -- exact for |t| >= 2^-17, overflow-free below 2^45 per 1024 arrivals --
//     so the total does not depend on the order of arrival: replays are bit-identical.
//   * One cell group per ATOM of 4 adjacent channels (summed exactly, as integers, before the atomics) and per row; a
//     workgroup adds to row (its index mod rows).  Device-scope atomics on one address retire at ~50 ns each (measured:
//     1024 arrivals on one address stretched a 9 us kernel to 60 us), so rows = arrivals / 8.
//   * (An earlier form had the last-arriving workgroup finalise behind a ticket: the ticket's own serialisation and the
//     three dependent round trips at the tail cost what the launch did.)
//   * A non-finite workgroup sum raises the scene's flag and the consumer's scale / shift come out NaN, as in float.
struct GnOut {
    unsigned long long *acc = nullptr;              // [B][rows][C/4][sum, sumsq][low word, high part]; null: not collected
    unsigned *flag = nullptr;                       // [B] non-finite flag
    int C = 0, rows = 1;                            // rows: a power of two
};
struct GnIn {                                       // the virtual concat [src 0 | upsample(src 1)]; src 1 optional
    const unsigned long long *acc[2] = {nullptr, nullptr};
    const unsigned *flag[2] = {nullptr, nullptr};
    int C[2] = {0, 0}, rows[2] = {1, 1};
    const float *gamma = nullptr, *beta = nullptr;
    int groups = 1;
    float eps = 0.0f;
    double count = 0.0;                             // voxels per scene at the consumer's resolution
};

constexpr int GN_ATOM = 4;
constexpr int GN_MAX_CIN = 512;                                     // channels a consumer's table holds (= HB_MAX_CIN)
constexpr size_t GN_SCRATCH_BYTES = (size_t)(GN_MAX_CIN / GN_ATOM) * 4 * 8 + 64 * 2 * 8;
__host__ __device__ inline size_t gn_acc_words(int B, int rows, int C) { return (size_t)B * rows * (C / GN_ATOM) * 4; }

// wave 0 of a workgroup, all 64 lanes: lane = channel * 2 + {sum, sumsq} of the channels [c0, c0 + 32) of scene b
__device__ __forceinline__ void gn_out_add(const GnOut &o, int b, unsigned wg, int c0, float tsum) {
    const int lane = threadIdx.x & 63;
    // "finite" = inside the accumulators' capacity: |t| < 2^45 keeps 1024 arrivals below 2^63 in the high cell (and the
    // double -> integer conversion below defined); a larger sum (sumsq beyond ~3.5e13 per workgroup), an infinity or a NaN
    // raises the scene's flag instead of wrapping silently -- the consumer's scale / shift then come out NaN
    const bool finite = fabsf(tsum) < 0x1p45f;                      // false for NaN
    const double v = finite ? (double)tsum * 0x1p40 : 0.0;          // exact
    const double h = floor(v * 0x1p-32);
    long long hi = (long long)h;
    unsigned long long lo = (unsigned long long)(v - h * 0x1p32);   // in [0, 2^32): what lies below 2^-40 is dropped
#pragma unroll
    for (int x = 2; x <= 4; x <<= 1) { hi += __shfl_xor(hi, x); lo += __shfl_xor(lo, x); }     // the atom's 4 channels, exactly
    if (__builtin_amdgcn_ballot_w64(!finite) && lane == 0) atomicOr(o.flag + b, 1u);
    if ((lane & 6) == 0) {
        const int A = o.C / GN_ATOM, row = (int)(wg & (unsigned)(o.rows - 1));
        unsigned long long *d = o.acc + ((((size_t)b * o.rows + row) * A + (c0 >> 2) + (lane >> 3)) * 2 + (lane & 1)) * 2;
        atomicAdd(d, lo);
        atomicAdd(d + 1, (unsigned long long)hi);
    }
}

__device__ __forceinline__ double gn_cell_value(long long hi, unsigned long long lo) {
    return ((double)hi * 0x1p32 + (double)lo) * 0x1p-40;
}

// Every thread of the workgroup (nthr of them): ssl[c] = pre * (scale, shift) of the consumer's channels for scene b, in two steps.
// gn_in_request asks for everything that comes from global memory (the rows, the flags, gamma / beta) at once -- ONE round trip (with
// the loads where they are used it was three, and cost a workgroup more than the finalising launch had) -- and the caller may put
// its own first requests behind it; gn_in_finish reduces in LDS and ends with a barrier.  scratch: GN_SCRATCH_BYTES of LDS nobody
// else touches during the call.  LDS_ONLY: the barriers wait for LDS traffic only (a caller with vector-memory requests in flight
// that must stay in flight).  The arithmetic after the sums is gn_finalize_kernel's.
template <int GN_PRE>                                               // row words a thread keeps in flight (the rest, if any, are read late)
struct GnReq { unsigned long long w[GN_PRE]; float gm[2], bt[2]; unsigned bad; };

template <int GN_PRE>
__device__ __forceinline__ GnReq<GN_PRE> gn_in_request(const GnIn &in, int b, int tid, int nthr) {
    GnReq<GN_PRE> r;
    const int A0 = in.C[0] / GN_ATOM, A1 = in.acc[1] ? in.C[1] / GN_ATOM : 0, Ct = in.C[0] + (in.acc[1] ? in.C[1] : 0);
    const int n0 = in.rows[0] * A0 * 4, n = n0 + in.rows[1] * A1 * 4;
    const unsigned long long *base0 = in.acc[0] + (size_t)b * n0, *base1 = A1 ? in.acc[1] + (size_t)b * (n - n0) : nullptr;
#pragma unroll
    for (int k = 0; k < GN_PRE; ++k) {
        const int i = tid + k * nthr;
        r.w[k] = i < n0 ? base0[i] : (i < n ? base1[i - n0] : 0ull);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + k * nthr;
        r.gm[k] = c < Ct ? in.gamma[c] : 0.0f;
        r.bt[k] = c < Ct ? in.beta[c] : 0.0f;
    }
    r.bad = tid < in.groups ? (in.flag[0][b] | (in.flag[1] ? in.flag[1][b] : 0u)) : 0u;
    return r;
}

template <bool LDS_ONLY>
__device__ __forceinline__ void gn_barrier() {
    if (LDS_ONLY) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0) alone
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else __syncthreads();
}

// table slot of channel c: [c][scale, shift], or (QUAD) per channel quad [4 scales][4 shifts] -- the layout the specialised-wave conv
// kernels read as 16-byte vectors
template <bool QUAD>
__device__ __forceinline__ int gn_slot(int c, int shift) { return QUAD ? (c >> 2) * 8 + shift * 4 + (c & 3) : c * 2 + shift; }

// grp (or null): [group][mean, rstd] as floats, somewhere that outlives the scratch
template <bool LDS_ONLY, int GN_PRE, bool QUAD = false>
__device__ __forceinline__ void gn_in_finish(const GnIn &in, const GnReq<GN_PRE> &r, int b, float pre, float *ssl, void *scratch, int tid, int nthr,
                                             float *grp = nullptr) {
    unsigned long long *cell = reinterpret_cast<unsigned long long *>(scratch);     // [atom of the concat][sum, sumsq][low, high]
    const int A0 = in.C[0] / GN_ATOM, A1 = in.acc[1] ? in.C[1] / GN_ATOM : 0, Ct = in.C[0] + (in.acc[1] ? in.C[1] : 0);
    double *stat = reinterpret_cast<double *>(cell + (size_t)(A0 + A1) * 4);       // [group][mean, rstd]
    const int per0 = A0 * 4, n0 = in.rows[0] * per0, per1 = A1 * 4, n = n0 + in.rows[1] * per1;
    for (int i = tid; i < (A0 + A1) * 4; i += nthr) cell[i] = 0ull;
    gn_barrier<LDS_ONLY>();
    auto slot = [&](int i) { return i < n0 ? i % per0 : per0 + (i - n0) % per1; };
#pragma unroll
    for (int k = 0; k < GN_PRE; ++k) {
        const int i = tid + k * nthr;
        if (i < n) atomicAdd(cell + slot(i), r.w[k]);               // ds_add_u64: integers, any order
    }
    if (n > GN_PRE * nthr) {                                        // (more rows than the request covers: not on the shipped shapes)
        const unsigned long long *base0 = in.acc[0] + (size_t)b * n0, *base1 = A1 ? in.acc[1] + (size_t)b * (n - n0) : nullptr;
        for (int i = tid + GN_PRE * nthr; i < n; i += nthr) atomicAdd(cell + slot(i), i < n0 ? base0[i] : base1[i - n0]);
    }
    gn_barrier<LDS_ONLY>();
    const int cpg = Ct / in.groups, apg = cpg / GN_ATOM;
    if (tid < in.groups) {
        long long hi[2][2] = {{0, 0}, {0, 0}};                      // [src][sum, sumsq]
        unsigned long long lo[2][2] = {{0, 0}, {0, 0}};
        for (int k = 0; k < apg; ++k) {
            const int at = tid * apg + k, src = at >= A0;
#pragma unroll
            for (int q = 0; q < 2; ++q) { lo[src][q] += cell[at * 4 + q * 2]; hi[src][q] += (long long)cell[at * 4 + q * 2 + 1]; }
        }
        const double tsum = gn_cell_value(hi[0][0], lo[0][0]) + 8.0 * gn_cell_value(hi[1][0], lo[1][0]);   // a `low` voxel stands for
        const double tsq = gn_cell_value(hi[0][1], lo[0][1]) + 8.0 * gn_cell_value(hi[1][1], lo[1][1]);    // its 8 upsampled copies
        const double cnt = in.count * cpg;
        double mean = tsum / cnt;
        double var = tsq / cnt - mean * mean;                       // biased variance, as torch
        if (var < 0.0) var = 0.0;
        double rstd = 1.0 / sqrt(var + (double)in.eps);
        if (r.bad) { mean = __builtin_nan(""); rstd = __builtin_nan(""); }
        stat[tid * 2] = mean; stat[tid * 2 + 1] = rstd;
        if (grp) { grp[tid * 2] = (float)mean; grp[tid * 2 + 1] = (float)rstd; }
    }
    gn_barrier<LDS_ONLY>();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tid + k * nthr;
        if (c < Ct) {
            const int g = c / cpg;
            const double sc = stat[g * 2 + 1] * (double)r.gm[k];
            ssl[gn_slot<QUAD>(c, 0)] = pre * (float)sc;
            ssl[gn_slot<QUAD>(c, 1)] = pre * (float)((double)r.bt[k] - stat[g * 2] * sc);
        }
    }
    for (int c = tid + 2 * nthr; c < Ct; c += nthr) {
        const int g = c / cpg;
        const double sc = stat[g * 2 + 1] * (double)in.gamma[c];
        ssl[gn_slot<QUAD>(c, 0)] = pre * (float)sc;
        ssl[gn_slot<QUAD>(c, 1)] = pre * (float)((double)in.beta[c] - stat[g * 2] * sc);
    }
    gn_barrier<LDS_ONLY>();
}

__device__ __forceinline__ void gn_in_scale_shift(const GnIn &in, int b, float pre, float *ssl, void *scratch, int tid, int nthr) {
    const GnReq<8> r = gn_in_request<8>(in, b, tid, nthr);
    gn_in_finish<false>(in, r, b, pre, ssl, scratch, tid, nthr);
}

// ---- GroupNorm statistics ---------------------------------------------------------------------
// Every producer of a tensor leaves per-block partial sums part[b][blk][c] = (sum, sumsq) over its
// voxels (the conv epilogue for conv outputs, channel_stats_kernel for pooled tensors and the
// scattered input).  The consumer's GroupNorm reduces them (fixed order: bit-reproducible): the
// statistics of the virtual concat [skip | upsample(low)] are the per-channel sums of skip plus
// 8x those of low (nearest upsampling repeats every value 8 times).
__global__ void __launch_bounds__(256)
channel_stats_kernel(const float *x, int C, size_t V, int nblk, float *part, GnOut stat_out) {
    __shared__ float red[8][32][2];
    const int b = blockIdx.y, blk = blockIdx.x;
    const size_t v0 = V * blk / nblk, v1 = V * (blk + 1) / nblk;
    const int c = threadIdx.x & 31, vg = threadIdx.x >> 5;
    const float *xb = x + (size_t)b * V * C;
    for (int cb = 0; cb < C; cb += 32) {
        float sum = 0.0f, sq = 0.0f;
        for (size_t v = v0 + vg; v < v1; v += 8) {
            const float t = xb[v * C + cb + c];
            sum += t; sq = fmaf(t, t, sq);
        }
        red[vg][c][0] = sum; red[vg][c][1] = sq;
        __syncthreads();
        if (part && threadIdx.x < 32) {
            float a = 0.0f, q = 0.0f;
            for (int i = 0; i < 8; ++i) { a += red[i][c][0]; q += red[i][c][1]; }
            float *dst = part + (((size_t)b * nblk + blk) * C + cb + c) * 2;
            dst[0] = a; dst[1] = q;
        }
        if (stat_out.acc && threadIdx.x < 64) {                    // thread = channel * 2 + {sum, sumsq}: the same eight terms in the same order
            float a = 0.0f;
            for (int i = 0; i < 8; ++i) a += red[i][threadIdx.x >> 1][threadIdx.x & 1];
            gn_out_add(stat_out, b, (unsigned)blk, cb, a);
        }
        __syncthreads();
    }
}

struct StatSrc { const float *part; int nblk, C; };

// one block per (scene, group): scale[b][c] = rstd*gamma, shift[b][c] = beta - mean*rstd*gamma.
// The (channel, partial block) pairs of the group are flattened over the threads (8-byte loads, consecutive
// threads on consecutive channels of one partial block) and reduced with wave shuffles: the kernel is a
// latency chain in front of every convolution, so it is kept short.
__global__ void __launch_bounds__(256)
gn_finalize_kernel(StatSrc s1, StatSrc s2, int groups, double count, const float *gamma, const float *beta,
                   float eps, float *scale_shift, unsigned long long *zero, size_t zero_words) {
    __shared__ double wred[4][2];
    __shared__ double stat[2];
    const int b = blockIdx.x, g = blockIdx.y;
    // vt_unet3d_fwd's first finalisation also clears the accumulator rows (GnOut) of the launches behind it
    for (size_t i = (size_t)(g * gridDim.x + b) * 256 + threadIdx.x; i < zero_words; i += (size_t)gridDim.x * gridDim.y * 256) zero[i] = 0ull;
    const int C = s1.C + s2.C, cpg = C / groups;
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    double sum = 0.0, sq = 0.0;
#pragma unroll
    for (int src = 0; src < 2; ++src) {
        const StatSrc &s = src ? s2 : s1;
        const int coff = src ? s1.C : 0;
        const double mult = src ? 8.0 : 1.0;                     // a `low` voxel stands for its 8 upsampled copies
        if (!s.part) continue;
        const int k0 = (c_lo > coff ? c_lo : coff) - coff, k1 = (c_hi < coff + s.C ? c_hi : coff + s.C) - coff;
        const int nk = k1 - k0;
        if (nk <= 0) continue;
        const int total = nk * s.nblk;
        const float2 *base = reinterpret_cast<const float2 *>(s.part) + (size_t)b * s.nblk * s.C + k0;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int blk = i / nk, k = i - blk * nk;
            const float2 p = base[(size_t)blk * s.C + k];
            sum += mult * (double)p.x; sq += mult * (double)p.y;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); sq += __shfl_xor(sq, o); }
    if ((threadIdx.x & 63) == 0) { wred[threadIdx.x >> 6][0] = sum; wred[threadIdx.x >> 6][1] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tsum = wred[0][0] + wred[1][0] + wred[2][0] + wred[3][0];
        const double tsq = wred[0][1] + wred[1][1] + wred[2][1] + wred[3][1];
        const double n = count * cpg, mean = tsum / n;
        double var = tsq / n - mean * mean;                      // biased variance, as torch
        if (var < 0.0) var = 0.0;
        stat[0] = mean; stat[1] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        const int c = g * cpg + threadIdx.x;
        const double sc = stat[1] * (double)gamma[c];
        scale_shift[((size_t)b * C + c) * 2 + 0] = (float)sc;
        scale_shift[((size_t)b * C + c) * 2 + 1] = (float)((double)beta[c] - stat[0] * sc);
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
    float *part;                // [B][spatial blocks][Cout][2] partial (sum, sumsq) of the output, or null
    int Cout, relu;
    int TX, TY, TZ;             // block tile of output voxels (TX*TY*TZ = 32 * waves)
    int tiles_x, tiles_y, tiles_z;
    float *kws = nullptr;       // K-split launches (gridDim.z slices of the input channels): [ksplit][B,D,H,W,Cout] raw partial sums
    int ksplit = 1;
    GnOut stat_out;             // the output's statistics into accumulator rows (instead of `part`)
    GnIn stat_in;               // the input's statistics from accumulator rows: scale / shift computed in the prologue (instead of `scale_shift`)
};

// stage channels [32 cib, 32 cib+32) of the normalised input tile (origin x0-1,y0-1,z0-1) into LDS;
// `tid`/`nthr` = this thread's rank among the threads that share the tile
__device__ __forceinline__ void stage_tile(float *tile, const ConvArgs &a, int b, int cib, int x0, int y0, int z0,
                                           int PX, int PY, int nvox, int tid, int nthr) {
    const Src &s = a.s;
    const int Cin = s.C1 + s.C2;
    const int sc4 = (tid & 7) * 4, ch = cib * 32 + sc4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.scale_shift) {
        const float *ss = a.scale_shift + ((size_t)b * Cin + ch) * 2;
        sc = f32x4{ss[0], ss[2], ss[4], ss[6]}; sh = f32x4{ss[1], ss[3], ss[5], ss[7]};
    }
    const bool from_low = ch >= s.C1;
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    for (int v = tid >> 3; v < nvox; v += nthr >> 3) {
        const int px = v % PX, r2 = v / PX, py = r2 % PY, pz = r2 / PY;
        const int gx = x0 + px - 1, gy = y0 + py - 1, gz = z0 + pz - 1;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (gx >= 0 && gx < s.W && gy >= 0 && gy < s.H && gz >= 0 && gz < s.D) {
            const float *src = from_low
                ? s.low + ((((size_t)b * D2 + (gz >> 1)) * H2 + (gy >> 1)) * W2 + (gx >> 1)) * s.C2 + (ch - s.C1)
                : s.skip + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * s.C1 + ch;
            val = *reinterpret_cast<const f32x4 *>(src);
            val.x = fmaf(val.x, sc.x, sh.x); val.y = fmaf(val.y, sc.y, sh.y);
            val.z = fmaf(val.z, sc.z, sh.z); val.w = fmaf(val.w, sc.w, sh.w);
        }
        float *d = tile + v * CPAD + sc4;
        d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
}

// 27 taps x 16 k-steps of MFMAs for one staged channel block; weight fragments double-buffered in registers
template <int NCO>
__device__ __forceinline__ void conv_taps(f32x16 (&acc)[NCO], const float *tile, const float *wc, int nco_all,
                                          int center, int PX, int PY, int kk) {
    float wn[NCO][16];
#pragma unroll
    for (int n = 0; n < NCO; ++n)
#pragma unroll
        for (int st = 0; st < 16; ++st) wn[n][st] = wc[n * 1024 + st * 64];
#pragma unroll 1
    for (int tap = 0; tap < 27; ++tap) {
        float wcur[NCO][16];
#pragma unroll
        for (int n = 0; n < NCO; ++n)
#pragma unroll
            for (int st = 0; st < 16; ++st) wcur[n][st] = wn[n][st];
        if (tap < 26) {
            const float *wt = wc + (size_t)(tap + 1) * nco_all * 1024;
#pragma unroll
            for (int n = 0; n < NCO; ++n)
#pragma unroll
                for (int st = 0; st < 16; ++st) wn[n][st] = wt[n * 1024 + st * 64];
        }
        const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
        const float *xin = tile + (center + (dz * PY + dy) * PX + dx) * CPAD + kk;
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const float bv = xin[2 * st];
#pragma unroll
            for (int n = 0; n < NCO; ++n) acc[n] = mfma(wcur[n][st], bv, acc[n]);
        }
    }
}

// Sum of x over each 32-lane half of the wave, valid in the half's LAST lane (31 / 63): four row-shift adds (an inclusive
// scan inside each row of 16 lanes) and one row broadcast, all DPP modifiers of v_add_f32 -- no LDS traffic.  (The five
// __shfl_xor steps this replaces compile to ds_bpermute: 160 of them per output tile and wave were ~15 000 cycles, more
// than the tile's 27 taps -- in-kernel stamps, tools/diag_conv.py.)
__device__ __forceinline__ float half_wave_sum(float x) {
#define VT_DPP_ADD(CTRL, ROWMASK) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xF, true))
    VT_DPP_ADD(0x111, 0xF);     // row_shr:1
    VT_DPP_ADD(0x112, 0xF);     // row_shr:2
    VT_DPP_ADD(0x114, 0xF);     // row_shr:4
    VT_DPP_ADD(0x118, 0xF);     // row_shr:8  -> lane 15 of every row holds the row's sum
    VT_DPP_ADD(0x142, 0xA);     // row_bcast:15 into rows 1 and 3 -> lanes 31 and 63 hold their half's sum
#undef VT_DPP_ADD
    return x;
}

// per-channel (sum, sumsq) over the wave's 32 voxels of one 32-channel block -> sred[chan][2]
__device__ __forceinline__ void wave_stats(const f32x16 &v, bool valid, int j, int kk, float *sred) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float x = valid ? v[r] : 0.0f;
        const float sm = half_wave_sum(x), sq = half_wave_sum(x * x);
        if (j == 31) { sred[chan_of(r, kk) * 2] = sm; sred[chan_of(r, kk) * 2 + 1] = sq; }
    }
}

// WAVES waves per workgroup (each 32 voxels x 32*NCO output channels); gridDim.y covers Cout/(32*NCO)
template <int NCO, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES >= 8 ? 2 : 1)
conv3d_gcr_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [(TZ+2)(TY+2)(TX+2)][CPAD], then stats scratch
    constexpr int THREADS = WAVES * 64;
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
        stage_tile(tile, a, b, cib, x0, y0, z0, PX, PY, nvox, threadIdx.x, THREADS);
        __syncthreads();
        conv_taps<NCO>(acc, tile, a.wp + ((size_t)cib * 27 * nco_all + co_blk0) * 1024 + lane, nco_all, center, PX, PY, kk);
    }
    // epilogue: ReLU, channels-last store (lane = voxel, 16 registers = channels chan_of(r,h)),
    // and the per-block (sum, sumsq) of what was stored, for the next layer's GroupNorm
    const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
    const bool valid = gx < s.W && gy < s.H && gz < s.D;
    float *orow = a.out + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.Cout;
    __syncthreads();                                              // tile[] is free: reuse it for the stats
    float *sred = tile;                                           // [WAVES][NCO*32][2]
#pragma unroll
    for (int n = 0; n < NCO; ++n) {
        f32x16 v = acc[n];
        if (a.relu) v = relu16(v);
        if (valid) store_acc16(orow + (co_blk0 + n) * 32, v, kk);
        if (a.part) wave_stats(v, valid, j, kk, sred + (wave * NCO + n) * 64);
    }
    if (a.part) {
        __syncthreads();
        const int nsp = a.tiles_x * a.tiles_y * a.tiles_z;
        const int spatial = blockIdx.x % nsp;
        for (int e = threadIdx.x; e < NCO * 32 * 2; e += THREADS) {
            float tsum = 0.0f;
            for (int w = 0; w < WAVES; ++w) tsum += sred[w * NCO * 64 + e];
            const int n = e >> 6, c2 = e & 63;                    // c2 = channel*2 + {sum,sq}
            a.part[(((size_t)b * nsp + spatial) * a.Cout + (co_blk0 + n) * 32) * 2 + c2] = tsum;
        }
    }
}

// Small volumes with many input channels (the 8^3 / 16^3 levels): ONE 32-voxel x 32-channel output
// tile per workgroup, its WAVES waves split the input-channel blocks (K) between them, each with a
// private staged tile, and their accumulators are summed through LDS.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 1)
conv3d_gcr_ksplit_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];    // WAVES private tiles
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
    const int ncib = (s.C1 + s.C2) / 32, nco_all = a.Cout / 32, co_blk = blockIdx.y;
    const int wy = j / a.TX, lx = j % a.TX, ly = wy % a.TY, lz = wy / a.TY;
    const int center = ((lz + 1) * PY + (ly + 1)) * PX + (lx + 1);
    float *tile = lds + (size_t)wave * nvox * CPAD;
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
    for (int c0 = 0; c0 < ncib; c0 += WAVES) {
        const int cib = c0 + wave;
        __syncthreads();
        if (cib < ncib) stage_tile(tile, a, b, cib, x0, y0, z0, PX, PY, nvox, lane, 64);
        __syncthreads();
        if (cib < ncib)
            conv_taps<1>(acc, tile, a.wp + ((size_t)cib * 27 * nco_all + co_blk) * 1024 + lane, nco_all, center, PX, PY, kk);
    }
    __syncthreads();
    float *red = lds;                                              // [WAVES-1][16][64]
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (wave != 0) return;
    f32x16 v = acc[0];
    for (int w = 0; w < WAVES - 1; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += red[(w * 16 + r) * 64 + lane];
    if (a.relu) v = relu16(v);
    const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
    const bool valid = gx < s.W && gy < s.H && gz < s.D;
    if (valid) store_acc16(a.out + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.Cout + co_blk * 32, v, kk);
    if (a.part) {
        const int nsp = a.tiles_x * a.tiles_y * a.tiles_z;
        wave_stats(v, valid, j, kk, a.part + (((size_t)b * nsp + blockIdx.x % nsp) * a.Cout + co_blk * 32) * 2);
    }
}

// ---- split-bf16 variant of the 'gcr' conv for the large volumes ---------------------------------
// Same implicit GEMM, on the bf16 matrix core with split operands (decode_common.h): the normalised
// input is split into hi/lo bf16 ONCE while it is staged (16 channels at a time: an 80-byte LDS row
// per voxel = 16 hi + 16 lo + pad), the weights once at pack time, and every 16-channel tap is
// W_lo x_hi + W_hi x_lo + W_hi x_hi = 3 x v_mfma_f32_32x32x16_bf16 per 32 voxels x 32 couts
// (96 matrix cycles against 8 x 64 on the f32 core).  A workgroup (16 waves) owns an 8 x 8 x 8 output
// tile; each wave owns one 4 x 8 voxel patch of one z-plane; the 27 taps' weight fragments of the
// current 16 channels sit in LDS next to the input tile (54 KB).
// The halo rows are pitched 12 in x: with 4 x 8 patches the 16 lanes of every ds_read_b128 group
// then fall on 16 distinct bank groups.  The next 16 channels (input and weights) are fetched into
// registers before the 27 taps run, so the global latency hides behind the matrix work.
constexpr int SB_ROW = 80;
constexpr int SB_PX = 12;
// Tile 8 x 8 x TZ voxels, 2*TZ waves: wave w owns z-plane w/2, x-half w%2 (one 4 x 8 patch).  TZ = 8 (16 waves) for the
// large levels; TZ = 2 (4 waves, 4x the workgroups) for the 16^3 / 8^3 levels, where 8^3 tiles cannot fill the chip.
constexpr int sb_rows(int TZ) { return (TZ + 2) * 10 * SB_PX; }
constexpr int sb_threads(int TZ) { return 128 * TZ; }
constexpr int sb_iters(int TZ) { return ((TZ + 2) * 100 * 4 + sb_threads(TZ) - 1) / sb_threads(TZ); }
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// weights -> [cin/16][tap][cout/32][hi,lo][64 lanes][8 bf16]: lane (co, kg) element e = cin 16q + 8kg + e
// half != 0: the same fragments with IEEE-half hi / lo parts (vt_conv3d_pack_f16x3_thin; conv3d_gcr_s_kernel<.., .., true>)
__global__ void conv3d_pack_s_kernel(const float *w, int Cout, int Cin, float *packed, size_t total, int half) {
    // one thread per 16-byte fragment
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(f & 63), part = (int)((f >> 6) & 1);
        size_t r = f >> 7;
        const int nco = Cout / 32;
        const int cob = (int)(r % nco); r /= nco;
        const int tap = (int)(r % 27);
        const int q = (int)(r / 27);
        const int co = cob * 32 + (l & 31), kg = l >> 5;
        bf16x8 v;
        f16x8 vh;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = w[((size_t)co * Cin + 16 * q + 8 * kg + e) * 27 + tap];
            const __bf16 hb = (__bf16)x;
            v[e] = part ? (__bf16)(x - (float)hb) : hb;
            const _Float16 hh = (_Float16)x;
            vh[e] = part ? (_Float16)(x - (float)hh) : hh;
        }
        if (half) reinterpret_cast<f16x8 *>(packed)[f] = vh;
        else reinterpret_cast<bf16x8 *>(packed)[f] = v;
    }
}

constexpr int SB_WFRAGS = 27 * 128;               // one (16-channel block, cout block)'s weight fragments: 27 taps x (hi, lo) x 64 lanes
constexpr int sb_witers(int TZ) { return (SB_WFRAGS + sb_threads(TZ) - 1) / sb_threads(TZ); }
constexpr size_t sb_lds(int TZ) { return (size_t)sb_rows(TZ) * SB_ROW + (size_t)SB_WFRAGS * 16 + (size_t)GN_MAX_CIN * 2 * sizeof(float); }

// KSPLIT: gridDim.z workgroups share an output tile, each over its contiguous 1/gridDim.z of the 16-channel blocks; the raw
// accumulators go to a.kws[blockIdx.z] and conv_ksum_kernel adds them up in slice order (ReLU and the statistics move there).
// For the thin levels of ONE scene (16^3, 8^3): 16-128 output tiles cannot fill 256 CUs, and a workgroup that walks 24
// channel blocks alone is a 77 us chain (384 -> 128 at 16^3).
// HALF: the operands as IEEE-half hi / lo pairs instead of bf16 pairs (21-22 mantissa bits instead of 16; the inputs are GroupNorm
// outputs and conv weights, inside the half range): the thin levels of a network whose large levels run the split-f16 kernels -- the
// 16^3-class levels on bf16 pairs were most of the encoder's remaining drift
template <int TZ, bool KSPLIT = false, bool HALF = false>
__global__ void __launch_bounds__(sb_threads(TZ))
conv3d_gcr_s_kernel(ConvArgs a) {
    typedef typename std::conditional<HALF, f16x8, bf16x8>::type e8;
    typedef typename std::conditional<HALF, f16x4, bf16x4>::type e4;
    typedef typename std::conditional<HALF, _Float16, __bf16>::type e1;
    constexpr int SB_ROWS = sb_rows(TZ), SB_THREADS = sb_threads(TZ), SB_ITERS = sb_iters(TZ), SB_WITERS = sb_witers(TZ);
    constexpr int NVOX = (TZ + 2) * 100;                            // padded tile voxels
    extern __shared__ __attribute__((aligned(16))) char stile[];    // [SB_ROWS x SB_ROW input][27 x 2 KB weights]; then stats scratch
    const Src &s = a.s;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, kg = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int tz = t % a.tiles_z;
    const int b = t / a.tiles_z;
    const int x0 = tx * 8, y0 = ty * 8, z0 = tz * TZ;
    const int Cin = s.C1 + s.C2, ncq = Cin / 16;
    const int co_blk = blockIdx.y, nco_all = a.Cout / 32;
    const int lx = j & 3, ly = j >> 2;
    const int wz = wave >> 1, wx = (wave & 1) * 4;
    const int center = ((wz + 1) * 10 + (ly + 1)) * SB_PX + (lx + wx + 1);
    e8 *wlds = reinterpret_cast<e8 *>(stile + (size_t)SB_ROWS * SB_ROW);
    float *ssl = reinterpret_cast<float *>(stile + (size_t)SB_ROWS * SB_ROW + (size_t)SB_WFRAGS * 16);   // GnIn: the layer's scale / shift table

    // ---- staging plan (the same for every channel block): thread -> (voxel, 4 of the 16 channels) ----
    const int sc4 = (threadIdx.x & 3) * 4;
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    int vskip[SB_ITERS], vlow[SB_ITERS], lrow[SB_ITERS];
    unsigned inside = 0;
#pragma unroll
    for (int it = 0; it < SB_ITERS; ++it) {
        const int v = (threadIdx.x >> 2) + it * (SB_THREADS / 4);
        const int px = v % 10, r2 = v / 10, py = r2 % 10, pz = r2 / 10;
        const int gx = x0 + px - 1, gy = y0 + py - 1, gz = z0 + pz - 1;
        const bool in = v < NVOX && gx >= 0 && gx < s.W && gy >= 0 && gy < s.H && gz >= 0 && gz < s.D;
        vskip[it] = in ? ((b * s.D + gz) * s.H + gy) * s.W + gx : 0;
        vlow[it] = in ? ((b * D2 + (gz >> 1)) * H2 + (gy >> 1)) * W2 + (gx >> 1) : 0;
        lrow[it] = v < NVOX ? ((pz * 10 + py) * SB_PX + px) * SB_ROW + sc4 * 2 : -1;
        if (in) inside |= 1u << it;
    }
    f32x4 pre[SB_ITERS];
    e8 wpre[SB_WITERS];
    auto fetch = [&](int q) {                      // next 16 input channels and their 27 taps of weights -> registers
        const int ch = q * 16 + sc4;
        const bool from_low = ch >= s.C1;
#pragma unroll
        for (int it = 0; it < SB_ITERS; ++it) {
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (inside >> it & 1u) {
                const float *src = from_low ? s.low + (size_t)vlow[it] * s.C2 + (ch - s.C1) : s.skip + (size_t)vskip[it] * s.C1 + ch;
                val = *reinterpret_cast<const f32x4 *>(src);
            }
            pre[it] = val;
        }
        const e8 *wq = reinterpret_cast<const e8 *>(a.wp) + ((size_t)q * 27 * nco_all + co_blk) * 128;
#pragma unroll
        for (int it = 0; it < SB_WITERS; ++it) {
            const int f = threadIdx.x + it * SB_THREADS;
            if (f < SB_WFRAGS) wpre[it] = wq[(size_t)(f >> 7) * nco_all * 128 + (f & 127)];
        }
    };
    auto commit = [&](int q) {                     // GroupNorm affine (zero padding AFTER the norm), split, LDS
        const int ch = q * 16 + sc4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.stat_in.acc[0]) {
            const float *ss = ssl + ch * 2;
            sc = f32x4{ss[0], ss[2], ss[4], ss[6]}; sh = f32x4{ss[1], ss[3], ss[5], ss[7]};
        } else if (a.scale_shift) {
            const float *ss = a.scale_shift + ((size_t)b * Cin + ch) * 2;
            sc = f32x4{ss[0], ss[2], ss[4], ss[6]}; sh = f32x4{ss[1], ss[3], ss[5], ss[7]};
        }
#pragma unroll
        for (int it = 0; it < SB_ITERS; ++it) {
            if (lrow[it] < 0) continue;
            const bool in = inside >> it & 1u;
            e4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = in ? fmaf(pre[it][e], sc[e], sh[e]) : 0.0f;
                const e1 hb = (e1)x;
                hi[e] = hb;
                lo[e] = (e1)(x - (float)hb);
            }
            *reinterpret_cast<e4 *>(stile + lrow[it]) = hi;
            *reinterpret_cast<e4 *>(stile + lrow[it] + 32) = lo;
        }
#pragma unroll
        for (int it = 0; it < SB_WITERS; ++it) {
            const int f = threadIdx.x + it * SB_THREADS;
            if (f < SB_WFRAGS) wlds[f] = wpre[it];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    const int q_lo = KSPLIT ? (int)blockIdx.z * (ncq / a.ksplit) : 0, q_hi = KSPLIT ? q_lo + ncq / a.ksplit : ncq;
    // the 4-wave form has the registers to keep the first chunk's request in flight across the statistics' round trip
    if (TZ == 2) fetch(q_lo);
    if (a.stat_in.acc[0]) gn_in_scale_shift(a.stat_in, b, 1.0f, ssl, stile, threadIdx.x, SB_THREADS);
    if (TZ != 2) fetch(q_lo);
    for (int q = q_lo; q < q_hi; ++q) {
        __syncthreads();                                           // the previous taps are done with the tile
        commit(q);
        __syncthreads();
        if (q + 1 < q_hi) fetch(q + 1);
        // the four operand fragments of tap t+1 are requested before the three MFMAs of tap t issue: one LDS latency
        // per tap hides under the matrix pipe instead of two being exposed in front of it (two taps ahead spills the
        // 16-wave kernel past its 128 registers and was slower)
        struct TapOps { e8 wh, wl, xh, xl; };
        auto tap_ops = [&](int tap) {
            TapOps o;
            o.wh = wlds[tap * 128 + lane]; o.wl = wlds[tap * 128 + 64 + lane];
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
            const char *xin = stile + (center + (dz * 10 + dy) * SB_PX + dx) * SB_ROW + kg * 16;
            o.xh = *reinterpret_cast<const e8 *>(xin);
            o.xl = *reinterpret_cast<const e8 *>(xin + 32);
            return o;
        };
        TapOps cur = tap_ops(0);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            TapOps nxt = cur;
            if (tap + 1 < 27) nxt = tap_ops(tap + 1);
            __builtin_amdgcn_sched_barrier(0);                     // keep the requests ahead of the MFMAs (the scheduler sinks them otherwise)
            acc = mfma_s(cur.wl, cur.xh, acc);                     // (decode_common.h: the bf16 or the f16 32x32x16 MFMA by operand type)
            acc = mfma_s(cur.wh, cur.xl, acc);
            acc = mfma_s(cur.wh, cur.xh, acc);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    }
    // epilogue: as conv3d_gcr_kernel (lane = voxel, 16 registers = channels chan_of(r,kg))
    __syncthreads();
    float *sred = reinterpret_cast<float *>(stile);               // [2*TZ waves][32][2]
    {
        const int gx = x0 + lx + wx, gy = y0 + ly, gz = z0 + wz;
        const bool valid = gx < s.W && gy < s.H && gz < s.D;
        const size_t off = ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.Cout;
        if (KSPLIT) {
            const size_t slab = (size_t)(gridDim.x / (a.tiles_x * a.tiles_y * a.tiles_z)) * s.D * s.H * s.W * a.Cout;
            if (valid) store_acc16(a.kws + blockIdx.z * slab + off + co_blk * 32, acc, kg);
            return;
        }
        float *orow = a.out + off;
        f32x16 v = acc;
        if (a.relu) v = relu16(v);
        if (valid) store_acc16(orow + co_blk * 32, v, kg);
        if (a.part || a.stat_out.acc) wave_stats(v, valid, j, kg, sred + wave * 64);
    }
    if (a.part || a.stat_out.acc) {
        __syncthreads();
        const int nsp = a.tiles_x * a.tiles_y * a.tiles_z;
        const int spatial = blockIdx.x % nsp;
        if (threadIdx.x < 64) {
            float tsum = 0.0f;
            for (int w = 0; w < 2 * TZ; ++w) tsum += sred[w * 64 + threadIdx.x];
            if (a.part) a.part[(((size_t)b * nsp + spatial) * a.Cout + co_blk * 32) * 2 + threadIdx.x] = tsum;
            if (a.stat_out.acc) gn_out_add(a.stat_out, b, (unsigned)spatial, co_blk * 32, tsum);
        }
    }
}

// out = relu(sum over the K slices, in slice order) and the (sum, sumsq) partials of every 128-voxel block: one workgroup per
// (block, 32 output channels), thread = (voxel lane 0..31, 4 channels), four voxels each -- every slice load of a thread is
// independent of the others (the kernel is one round trip deep)
constexpr int KSUM_VOX = 128;
template <int KS>       // slices known at compile time (0: run-time `ks`): their loads are issued together, not one round trip each
__global__ void __launch_bounds__(256)
conv_ksum_kernel(const float *kws, int ks, size_t slab, int Cout, int relu, float *out, float *part, GnOut stat_out, int blocks_per_scene) {
    __shared__ float red[32][8][8];
    const int vl = threadIdx.x >> 3, q = threadIdx.x & 7;
    f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KSUM_VOX / 32; ++i) {
        const size_t e = ((size_t)blockIdx.x * KSUM_VOX + i * 32 + vl) * Cout + blockIdx.y * 32 + q * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(kws + e);
        if (KS) {
            f32x4 w[KS ? KS : 1];
#pragma unroll
            for (int k = 1; k < KS; ++k) w[k] = *reinterpret_cast<const f32x4 *>(kws + k * slab + e);
#pragma unroll
            for (int k = 1; k < KS; ++k) { v.x += w[k].x; v.y += w[k].y; v.z += w[k].z; v.w += w[k].w; }
        } else {
            for (int k = 1; k < ks; ++k) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(kws + k * slab + e);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
        }
        if (relu) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        *reinterpret_cast<f32x4 *>(out + e) = v;
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
        sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
    }
    if (!part && !stat_out.acc) return;
    float *r = red[vl][q];
    r[0] = sm.x; r[1] = sq.x; r[2] = sm.y; r[3] = sq.y; r[4] = sm.z; r[5] = sq.z; r[6] = sm.w; r[7] = sq.w;
    __syncthreads();
    if (threadIdx.x < 64) {                                        // threadIdx.x = channel * 2 + {sum, sumsq} of the 32-channel block
        float t = 0.0f;
        for (int v = 0; v < 32; ++v) t += red[v][threadIdx.x >> 3][threadIdx.x & 7];
        if (part) part[((size_t)blockIdx.x * Cout + blockIdx.y * 32) * 2 + threadIdx.x] = t;
        if (stat_out.acc) gn_out_add(stat_out, blockIdx.x / blocks_per_scene, blockIdx.x % blocks_per_scene, blockIdx.y * 32, t);
    }
}

// ---- the same convolution for TWO workgroups per CU --------------------------------------------------------
// The kernel above keeps one workgroup per CU (94-151 KB of LDS), so its memory half (global fetch round trips,
// barriers, the output store) and its tap half never overlap: 34 + 28 us on a 32->32 conv at 64^3.  Here the tile is
// 8 x 8 x 4 (8 waves, 57.6 KB) and the 27 taps of weights pass through LDS in three 9-tap slabs (18.4 KB, fetched one
// slab ahead into registers): 76 KB, 128 registers -- two workgroups share a CU and one's taps run under the other's
// fetch / commit / store.  Costs: six barriers per 16-channel block instead of two, halo factor 2.34 instead of 1.95.
constexpr int S4_TZ = 4, S4_THREADS = 512, S4_NVOX = 600, S4_ROWS = (S4_TZ + 2) * 10 * SB_PX;
constexpr int S4_ITERS = (S4_NVOX * 4 + S4_THREADS - 1) / S4_THREADS;         // 5
constexpr int S4_SLAB = 9 * 128;                                               // fragments per slab
constexpr int S4_WITERS = (S4_SLAB + S4_THREADS - 1) / S4_THREADS;             // 3
constexpr size_t S4_LDS = (size_t)S4_ROWS * SB_ROW + (size_t)S4_SLAB * 16 + (size_t)GN_MAX_CIN * 2 * sizeof(float);
static_assert(S4_LDS <= 80 * 1024, "two workgroups must fit the 160 KB of a CU");

template <bool HALF = false>
__global__ void __launch_bounds__(S4_THREADS, 4)        // second argument: waves per SIMD (HIP), i.e. two workgroups per CU -> 128 registers
conv3d_gcr_s4_kernel(ConvArgs a) {
    typedef typename std::conditional<HALF, f16x8, bf16x8>::type e8;
    typedef typename std::conditional<HALF, f16x4, bf16x4>::type e4;
    typedef typename std::conditional<HALF, _Float16, __bf16>::type e1;
    extern __shared__ __attribute__((aligned(16))) char stile[];    // [S4_ROWS x SB_ROW input][9 x 2 KB weights]; then stats scratch
    const Src &s = a.s;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, kg = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int tz = t % a.tiles_z;
    const int b = t / a.tiles_z;
    const int x0 = tx * 8, y0 = ty * 8, z0 = tz * S4_TZ;
    const int Cin = s.C1 + s.C2, ncq = Cin / 16;
    const int co_blk = blockIdx.y, nco_all = a.Cout / 32;
    const int lx = j & 3, ly = j >> 2;
    const int wz = wave >> 1, wx = (wave & 1) * 4;
    const int center = ((wz + 1) * 10 + (ly + 1)) * SB_PX + (lx + wx + 1);
    e8 *wlds = reinterpret_cast<e8 *>(stile + (size_t)S4_ROWS * SB_ROW);
    float *ssl = reinterpret_cast<float *>(stile + (size_t)S4_ROWS * SB_ROW + (size_t)S4_SLAB * 16);     // GnIn: the layer's scale / shift table
    if (a.stat_in.acc[0]) gn_in_scale_shift(a.stat_in, b, 1.0f, ssl, stile, threadIdx.x, S4_THREADS);

    const int sc4 = (threadIdx.x & 3) * 4;
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    int vskip[S4_ITERS], vlow[S4_ITERS], lrow[S4_ITERS];
    unsigned inside = 0;
#pragma unroll
    for (int it = 0; it < S4_ITERS; ++it) {
        const int v = (threadIdx.x >> 2) + it * (S4_THREADS / 4);
        const int px = v % 10, r2 = v / 10, py = r2 % 10, pz = r2 / 10;
        const int gx = x0 + px - 1, gy = y0 + py - 1, gz = z0 + pz - 1;
        const bool in = v < S4_NVOX && gx >= 0 && gx < s.W && gy >= 0 && gy < s.H && gz >= 0 && gz < s.D;
        vskip[it] = in ? ((b * s.D + gz) * s.H + gy) * s.W + gx : 0;
        vlow[it] = in ? ((b * D2 + (gz >> 1)) * H2 + (gy >> 1)) * W2 + (gx >> 1) : 0;
        lrow[it] = v < S4_NVOX ? ((pz * 10 + py) * SB_PX + px) * SB_ROW + sc4 * 2 : -1;
        if (in) inside |= 1u << it;
    }
    f32x4 pre[S4_ITERS];
    e8 wpre[S4_WITERS];
    auto fetch_in = [&](int q) {                   // next 16 input channels -> registers
        const int ch = q * 16 + sc4;
        const bool from_low = ch >= s.C1;
#pragma unroll
        for (int it = 0; it < S4_ITERS; ++it) {
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (inside >> it & 1u) {
                const float *src = from_low ? s.low + (size_t)vlow[it] * s.C2 + (ch - s.C1) : s.skip + (size_t)vskip[it] * s.C1 + ch;
                val = *reinterpret_cast<const f32x4 *>(src);
            }
            pre[it] = val;
        }
    };
    auto fetch_w = [&](int q, int slab) {          // nine taps of weights of channel block q -> registers
        const e8 *wq = reinterpret_cast<const e8 *>(a.wp) + (((size_t)q * 27 + slab * 9) * nco_all + co_blk) * 128;
#pragma unroll
        for (int it = 0; it < S4_WITERS; ++it) {
            const int f = threadIdx.x + it * S4_THREADS;
            if (f < S4_SLAB) wpre[it] = wq[(size_t)(f >> 7) * nco_all * 128 + (f & 127)];
        }
    };
    auto commit_in = [&](int q) {                  // GroupNorm affine (zero padding AFTER the norm), split, LDS
        const int ch = q * 16 + sc4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.stat_in.acc[0]) {
            const float *ss = ssl + ch * 2;
            sc = f32x4{ss[0], ss[2], ss[4], ss[6]}; sh = f32x4{ss[1], ss[3], ss[5], ss[7]};
        } else if (a.scale_shift) {
            const float *ss = a.scale_shift + ((size_t)b * Cin + ch) * 2;
            sc = f32x4{ss[0], ss[2], ss[4], ss[6]}; sh = f32x4{ss[1], ss[3], ss[5], ss[7]};
        }
#pragma unroll
        for (int it = 0; it < S4_ITERS; ++it) {
            if (lrow[it] < 0) continue;
            const bool in = inside >> it & 1u;
            e4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = in ? fmaf(pre[it][e], sc[e], sh[e]) : 0.0f;
                const e1 hb = (e1)x;
                hi[e] = hb;
                lo[e] = (e1)(x - (float)hb);
            }
            *reinterpret_cast<e4 *>(stile + lrow[it]) = hi;
            *reinterpret_cast<e4 *>(stile + lrow[it] + 32) = lo;
        }
    };
    auto commit_w = [&]() {
#pragma unroll
        for (int it = 0; it < S4_WITERS; ++it) {
            const int f = threadIdx.x + it * S4_THREADS;
            if (f < S4_SLAB) wlds[f] = wpre[it];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    struct TapOps { e8 wh, wl, xh, xl; };
    auto run_slab = [&](int slab) {                // nine taps: dz = slab - 1
        auto tap_ops = [&](int tl) {
            TapOps o;
            o.wh = wlds[tl * 128 + lane]; o.wl = wlds[tl * 128 + 64 + lane];
            const int dy = tl / 3 - 1, dx = tl % 3 - 1;
            const char *xin = stile + (center + ((slab - 1) * 10 + dy) * SB_PX + dx) * SB_ROW + kg * 16;
            o.xh = *reinterpret_cast<const e8 *>(xin);
            o.xl = *reinterpret_cast<const e8 *>(xin + 32);
            return o;
        };
        TapOps cur = tap_ops(0);
#pragma unroll
        for (int tl = 0; tl < 9; ++tl) {
            TapOps nxt = cur;
            if (tl + 1 < 9) nxt = tap_ops(tl + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma_s(cur.wl, cur.xh, acc);
            acc = mfma_s(cur.wh, cur.xl, acc);
            acc = mfma_s(cur.wh, cur.xh, acc);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    };

    fetch_in(0);
    fetch_w(0, 0);
    for (int q = 0; q < ncq; ++q) {
        __syncthreads();                                           // the previous taps are done with the tile and the slab
        commit_in(q);
        commit_w();
        __syncthreads();
        fetch_w(q, 1);
        if (q + 1 < ncq) fetch_in(q + 1);
        run_slab(0);
        __syncthreads();
        commit_w();
        __syncthreads();
        fetch_w(q, 2);
        run_slab(1);
        __syncthreads();
        commit_w();
        __syncthreads();
        if (q + 1 < ncq) fetch_w(q + 1, 0);
        run_slab(2);
    }
    // epilogue: as conv3d_gcr_s_kernel
    __syncthreads();
    float *sred = reinterpret_cast<float *>(stile);               // [8 waves][32][2]
    {
        const int gx = x0 + lx + wx, gy = y0 + ly, gz = z0 + wz;
        const bool valid = gx < s.W && gy < s.H && gz < s.D;
        float *orow = a.out + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.Cout;
        f32x16 v = acc;
        if (a.relu) v = relu16(v);
        if (valid) store_acc16(orow + co_blk * 32, v, kg);
        if (a.part || a.stat_out.acc) wave_stats(v, valid, j, kg, sred + wave * 64);
    }
    if (a.part || a.stat_out.acc) {
        __syncthreads();
        const int nsp = a.tiles_x * a.tiles_y * a.tiles_z;
        const int spatial = blockIdx.x % nsp;
        if (threadIdx.x < 64) {
            float tsum = 0.0f;
            for (int w = 0; w < 2 * S4_TZ; ++w) tsum += sred[w * 64 + threadIdx.x];
            if (a.part) a.part[(((size_t)b * nsp + spatial) * a.Cout + co_blk * 32) * 2 + threadIdx.x] = tsum;
            if (a.stat_out.acc) gn_out_add(a.stat_out, b, (unsigned)spatial, co_blk * 32, tsum);
        }
    }
}

// ---- split-f16 'gcr' conv, persistent and double-buffered ("f16x3"; the 64^3 and 32^3 levels) -----------------
// The split-bf16 kernels above alternate "stage 16 channels" and "run 27 taps" behind workgroup barriers with ONE LDS
// image: on a 32->32 conv at 64^3 the staging half (global round trips, barriers, the output store) takes 34 of the
// 62 us and the matrix pipe idles through it.  This kernel removes the alternation:
//   * channel chunks of EIGHT: an MFMA k-step (K = 16) is a PAIR of taps x 8 input channels (lane half 0 reads tap 2s,
//     lane half 1 tap 2s+1; 27 taps = 14 k-steps, the 28th half-step has zero weights), so a chunk's image is half the
//     size -- the hi and lo planes are 16-byte rows (8 halves), conflict-free at the 12-voxel row pitch without padding --
//     and BOTH the image (2 x 38.4 KB at 8^3) and the chunk's weight fragments (2 x 28 KB) are double-buffered in LDS:
//     chunk n+1 is normalised, split and written while chunk n's taps run, chunk n+2 is in flight from global memory
//     into registers; ONE barrier per chunk.
//   * half of the waves of every SIMD write the next chunk before their taps, the other half between their 7th and 8th
//     k-step, so the VALU work of the commit issues under the other waves' MFMAs instead of all 16 waves committing at once.
//   * persistent workgroups: a workgroup walks its tiles of one scene with the same pipeline running across tile
//     boundaries (no per-tile prologue), keeps the GroupNorm partial sums of its outputs per lane and reduces them once.
//   * operands are IEEE halves (hi = half(x), lo = half(x - hi): 21-22 mantissa bits; the inputs are GroupNorm outputs
//     and conv weights, far inside the half range): the error against the exact-f32 kernel is at f32 rounding level.
// Tile 8 x 8 x TZ, 2*TZ waves (TZ = 8: 16 waves, 64^3-class volumes; TZ = 4: 8 waves for the 32^3 class).
constexpr int HB_PX = 12;
constexpr int hb_rows(int TZ) { return (TZ + 2) * 10 * HB_PX; }
constexpr int hb_nvox(int TZ) { return (TZ + 2) * 100; }
constexpr int hb_threads(int TZ) { return 128 * TZ; }
constexpr int hb_iters(int TZ) { return (hb_nvox(TZ) * 2 + hb_threads(TZ) - 1) / hb_threads(TZ); }     // float4 items per thread and chunk
constexpr int HB_KSTEPS = 14;
constexpr int HB_WFRAGS = HB_KSTEPS * 2 * 64;                                                             // 16-byte fragments per chunk
constexpr int hb_witers(int TZ) { return (HB_WFRAGS + hb_threads(TZ) - 1) / hb_threads(TZ); }
constexpr size_t hb_img_bytes(int TZ) { return (size_t)hb_rows(TZ) * 32; }                               // hi plane + lo plane
constexpr int HB_MAX_CIN = 512;                                                                          // scale / shift table in LDS
constexpr size_t hb_lds(int TZ) { return 2 * hb_img_bytes(TZ) + 2 * (size_t)HB_WFRAGS * 16 + (size_t)2 * TZ * 64 * sizeof(float) + HB_MAX_CIN * 2 * sizeof(float); }
static_assert(hb_lds(8) <= 160 * 1024, "split-f16 conv: LDS budget");

// weights -> [cin/8][cout/32][k-step 0..13][hi,lo][64 lanes][8 halves]: lane (co, kg), element e = cin 8q + e of tap 2s + kg
// transposed: w is the [Cin][Cout][27] weight of the conv whose DATA GRADIENT this one is -- (co, ci, tap) reads w[ci][co][26 - tap]
// (channels swapped, taps flipped) without a flipped / transposed copy of the tensor
__global__ void conv3d_pack_h_kernel(const float *w, int Cout, int Cin, float *packed, size_t total, int transposed = 0) {
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(f & 63), part = (int)((f >> 6) & 1);
        size_t r = f >> 7;
        const int ks = (int)(r % HB_KSTEPS); r /= HB_KSTEPS;
        const int nco = Cout / 32;
        const int cob = (int)(r % nco);
        const int q = (int)(r / nco);
        const int co = cob * 32 + (l & 31), kg = l >> 5, tap = 2 * ks + kg;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = tap >= 27 ? 0.0f : transposed ? w[((size_t)(8 * q + e) * Cout + co) * 27 + (26 - tap)] : w[((size_t)co * Cin + 8 * q + e) * 27 + tap];
            const _Float16 hb = (_Float16)x;
            v[e] = part ? (_Float16)(x - (float)hb) : hb;
        }
        reinterpret_cast<f16x8 *>(packed)[f] = v;
    }
}

struct HbArgs {
    ConvArgs c;
    int wgs_per_scene;          // persistent workgroups per scene (= partial-statistics blocks per scene)
    const float *in_absmax;     // or null.  Device scalar max |input|: the input is multiplied by the power of two that brings
                                // it to ~2^10 before the split (and the result divided by it), so that tensors far below the
                                // half range -- output gradients in the data-gradient convolution -- keep all their bits
    // the UNet3D's final 1x1x1 convolution in the epilogue of its last 3x3x3 layer (Cout = 32 -> 32 channels, specialised-wave kernel
    // only): relu(conv) stays in registers, is split into half pairs -- the accumulator layout is the next MFMA's B operand, as
    // between the decoder's layers -- and six MFMAs against fin_w (vt_conv1x1_pack_f16x3) + fin_b give what is stored
    const float *fin_w = nullptr;
    const float *fin_b = nullptr;
    float *fin_y = nullptr;     // or null: relu(conv) itself goes here as well (the training forward keeps it for the backward)
    // or null.  [B][D/8][H/8][W/8] bytes: 1 = no voxel of that 8^3 block's halo (as far as this layer reads) differs from zero before
    // the GroupNorm, i.e. the normalised input there is the per-channel shift.  The specialised-wave kernel then skips the block's taps:
    // its output is sum over the in-volume taps of T[tap][cout] = sum_cin W[cout][cin][tap] shift[cin] (vt_voxel_tile_flags makes the
    // flags; plain layers only: no `low`, no in_absmax)
    const unsigned char *tile_skip = nullptr;
    unsigned skip_mask = 0xffu;         // which bits of a flag byte count (vt_voxel_tile_flags: bit 0 = 10^3 halo empty, bit 1 = 12^3 halo too)
    // The layer BEHIND a flagged first layer (32 -> 32 channels, 8 groups): over a block whose 12^3 halo holds no point the first layer's
    // output is a constant per border class, so this layer's output is one too -- per voxel class c2 (per axis: coordinate 0, 1, inner,
    // R - 2, R - 1: 125 classes)  K2[c2][co] = sum_g rstd_g (A_g - mean_g C_g) + B_g  with the second GroupNorm's group statistics and
    //   A_g = sum over the in-volume taps and the group's channels of W2 gamma2 relu(K1[class of the neighbour]),  C_g = ... of W2 gamma2,
    //   B_g = ... of W2 beta2
    // which depend on the FIRST layer's statistics only.  The first layer's launch writes them (cls_out [B][125][32][24]; cls_w: the second
    // layer's packed weights, cls_gamma / cls_beta its GroupNorm), the second layer's reads them (cls_in) with tile_skip / skip_mask = 2.
    // data-gradient convs (training): the tensor x [B][D][H][W][Cout] whose product with the output replaces the squares in the
    // per-workgroup statistics -- (sum out, sum out * x) per channel are the GroupNorm backward's two sums (gn_bwd_stats_kernel),
    // so that pass over dxn and x is not launched (the XST instantiation of the specialised-wave kernel)
    const float *stat_x = nullptr;
    const float *cls_w = nullptr, *cls_gamma = nullptr, *cls_beta = nullptr;
    float *cls_out = nullptr;
    const float *cls_in = nullptr;
    // decoder-entry layers in per-parity form (conv3d_gcr_up_kernel): the merged class weights of the `low` channels
    // (vt_conv3d_pack_f16x3_up); c.wp then only serves the skip channels' chunks
    const float *wp_up = nullptr;
};
constexpr int HB_SKIP_LIST = 64;                                     // tiles of either kind a workgroup can hold in its lists
constexpr int HB_CLS = 125, HB_CLS_ROW = 24;                        // border classes of the layer behind a flagged one; floats per (class, cout)
constexpr size_t hb_lds_sparse(int TZ) { return hb_lds(TZ) + 2 * 27 * 32 * sizeof(float) + 2 * HB_SKIP_LIST * sizeof(unsigned short) + 64 * sizeof(int); }
static_assert(hb_lds_sparse(8) <= 160 * 1024, "split-f16 conv with skip lists: LDS budget");

template <int TZ>
__global__ void __launch_bounds__(hb_threads(TZ))
conv3d_gcr_h_kernel(HbArgs ha) {
    constexpr int ROWS = hb_rows(TZ), NVOX = hb_nvox(TZ), THREADS = hb_threads(TZ), ITERS = hb_iters(TZ);
    constexpr int IMG = (int)hb_img_bytes(TZ), WBUF = HB_WFRAGS * 16;
    extern __shared__ __attribute__((aligned(16))) char hl[];      // [2 images][2 weight buffers][stats scratch]
    const ConvArgs &a = ha.c;
    const Src &s = a.s;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int b = blockIdx.x / ha.wgs_per_scene, wg = blockIdx.x - b * ha.wgs_per_scene;
    const int nsp = a.tiles_x * a.tiles_y * a.tiles_z;
    const int ntile = (nsp - wg + ha.wgs_per_scene - 1) / ha.wgs_per_scene;      // this workgroup's tiles: wg, wg + wgs, ...
    const int Cin = s.C1 + s.C2, ncq = Cin / 8;
    const int co_blk = blockIdx.y, nco_all = a.Cout / 32;
    const int lx = j & 3, ly = j >> 2;
    const int wz = wave >> 1, wx = (wave & 1) * 4;
    const int center = ((wz + 1) * 10 + (ly + 1)) * HB_PX + (lx + wx + 1);
    char *wbase = hl + 2 * IMG;
    float *sred = reinterpret_cast<float *>(hl + 2 * IMG + 2 * WBUF);
    float *ssl = sred + 2 * TZ * 64;                                // [Cin][2] GroupNorm scale / shift of this scene (read by every commit)
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    float pre_scale = 1.0f, post_scale = 1.0f;
    if (ha.in_absmax) {
        const float m = *ha.in_absmax;
        if (m > 0.0f && m < 3.0e38f) {
            const int e = 10 - ilogbf(m);                               // 2^e * m in [2^10, 2^11)
            pre_scale = ldexpf(1.0f, e < -100 ? -100 : (e > 100 ? 100 : e));
            post_scale = 1.0f / pre_scale;                              // exact: a power of two
        }
    }
    if (a.stat_in.acc[0]) gn_in_scale_shift(a.stat_in, b, pre_scale, ssl, hl, threadIdx.x, THREADS);   // (scratch: the first image buffer, still idle)
    else
        for (int i = threadIdx.x; i < 2 * Cin; i += THREADS)
            ssl[i] = pre_scale * (a.scale_shift ? a.scale_shift[(size_t)b * Cin * 2 + i] : ((i & 1) ? 0.0f : 1.0f));
    __syncthreads();

    // staging plan, the same for every tile: item -> (halo voxel, which four of the chunk's eight channels)
    int pxyz[ITERS], lrow[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int item = threadIdx.x + it * THREADS, v = item >> 1;
        const int px = v % 10, r2 = v / 10, py = r2 % 10, pz = r2 / 10;
        pxyz[it] = px | (py << 8) | (pz << 16);
        lrow[it] = v < NVOX ? ((pz * 10 + py) * HB_PX + px) * 16 + (item & 1) * 8 : -1;
    }
    const int c4 = (threadIdx.x & 1) * 4;                           // THREADS is even: the item's channel half is the thread's
    auto tile_origin = [&](int k, int &x0, int &y0, int &z0) {
        int t = wg + k * ha.wgs_per_scene;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y; t /= a.tiles_y;
        x0 = tx * 8; y0 = ty * 8; z0 = t * TZ;
    };

    f32x4 pre[ITERS];
    unsigned pre_in = 0;                                           // which of pre[] lie inside the volume
    // chunk n's weight fragments: global -> LDS buffer n & 1 by LDS-DMA (fragment order = LDS order: 28 pieces of 1 KiB,
    // no registers, no VALU); counted in vmcnt in issue order with the other vector-memory operations of the wave
    int d_q = 0;
    auto dma_w = [&](int n) {
        const int q = d_q;
        if (++d_q == ncq) d_q = 0;
        const f16x8 *wq = reinterpret_cast<const f16x8 *>(a.wp) + ((size_t)q * nco_all + co_blk) * HB_WFRAGS;
        char *dst = wbase + (n & 1) * WBUF;
        for (int p = wave; p < HB_WFRAGS / 64; p += 2 * TZ)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wq + p * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
    };
    // chunks are fetched, committed and consumed in order n = 0, 1, 2, ...: running (tile, channel chunk) counters instead of
    // divisions per call
    int f_q = 0, f_x0, f_y0, f_z0, f_k = 0;
    tile_origin(0, f_x0, f_y0, f_z0);
    auto fetch = [&](int) {                                        // the next chunk = (tile f_k, channels 8 f_q ..) -> registers
        const int q = f_q, x0 = f_x0, y0 = f_y0, z0 = f_z0;
        if (++f_q == ncq) { f_q = 0; ++f_k; tile_origin(f_k, f_x0, f_y0, f_z0); }
        const int ch = q * 8 + c4;
        const bool from_low = ch >= s.C1;
        pre_in = 0;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int gx = x0 + (pxyz[it] & 255) - 1, gy = y0 + ((pxyz[it] >> 8) & 255) - 1, gz = z0 + (pxyz[it] >> 16) - 1;
            const bool in = lrow[it] >= 0 && gx >= 0 && gx < s.W && gy >= 0 && gy < s.H && gz >= 0 && gz < s.D;
            // every wave issues exactly ITERS loads per chunk (the counted vmcnt wait relies on it): out-of-volume items
            // read a clamped, valid address and are zeroed at the commit
            const int cx = min(max(gx, 0), s.W - 1), cy = min(max(gy, 0), s.H - 1), cz = min(max(gz, 0), s.D - 1);
            const float *src = from_low
                ? s.low + (size_t)(((b * D2 + (cz >> 1)) * H2 + (cy >> 1)) * W2 + (cx >> 1)) * s.C2 + (ch - s.C1)
                : s.skip + (size_t)(((b * s.D + cz) * s.H + cy) * s.W + cx) * s.C1 + ch;
            pre[it] = *reinterpret_cast<const f32x4 *>(src);
            if (in) pre_in |= 1u << it;
        }
    };
    int c_q = 0;
    auto commit = [&](int n) {                                     // registers -> LDS buffers n & 1: GroupNorm affine (zero padding AFTER the norm), split
        const int ch = c_q * 8 + c4;
        if (++c_q == ncq) c_q = 0;
        const float *ss = ssl + ch * 2;
        const f32x4 sc = {ss[0], ss[2], ss[4], ss[6]}, sh = {ss[1], ss[3], ss[5], ss[7]};
        char *img = hl + (n & 1) * IMG;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            if (lrow[it] < 0) continue;
            const bool in = pre_in >> it & 1u;
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = in ? fmaf(pre[it][e], sc[e], sh[e]) : 0.0f;
                const _Float16 hb = (_Float16)x;
                hi[e] = hb;
                lo[e] = (_Float16)(x - (float)hb);
            }
            *reinterpret_cast<f16x4 *>(img + lrow[it]) = hi;
            *reinterpret_cast<f16x4 *>(img + ROWS * 16 + lrow[it]) = lo;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    sred[wave * 64 + lane] = 0.0f;                                 // the wave's running (sum, sumsq) per output channel (only this wave touches it)

    struct Ops { f16x8 wh, wl, xh, xl; };
    auto ops_of = [&](int n, int ks) {
        Ops o;
        const f16x8 *wl = reinterpret_cast<const f16x8 *>(wbase + (n & 1) * WBUF);
        o.wh = wl[ks * 128 + lane]; o.wl = wl[ks * 128 + 64 + lane];
        const int t0 = 2 * ks, t1 = 2 * ks + 1 < 27 ? 2 * ks + 1 : 13;                 // the 28th half-step: any row, its weights are zero
        const int r0 = ((t0 / 9 - 1) * 10 + ((t0 / 3) % 3 - 1)) * HB_PX + (t0 % 3 - 1);
        const int r1 = ((t1 / 9 - 1) * 10 + ((t1 / 3) % 3 - 1)) * HB_PX + (t1 % 3 - 1);
        const char *xin = hl + (n & 1) * IMG + (center + (kg ? r1 : r0)) * 16;
        o.xh = *reinterpret_cast<const f16x8 *>(xin);
        o.xl = *reinterpret_cast<const f16x8 *>(xin + ROWS * 16);
        return o;
    };
    auto taps = [&](int n, int ks0, int ks1) {
        Ops cur = ops_of(n, ks0);
#pragma unroll
        for (int ks = ks0; ks < ks1; ++ks) {
            Ops nxt = cur;
            if (ks + 1 < ks1) nxt = ops_of(n, ks + 1);
            __builtin_amdgcn_sched_barrier(0);                     // keep the requests ahead of the MFMAs
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl, cur.xh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.xl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.xh, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    };

    const int N = ntile * ncq;
    int e_q = 0, e_k = 0;
    const bool late = (wave >> 2) & 1;                              // waves w and w + 4 share a SIMD: one commits first, one mid-taps
    // s_waitcnt vmcnt(ITERS) alone: everything but the wave's ITERS youngest vector-memory operations (the register fetch of
    // chunk n + 2, issued after the DMA of chunk n + 1) has landed
    constexpr int WAIT_DMA = 0x0F70 | ITERS;
    if (N > 0) {
        dma_w(0);
        fetch(0);
        commit(0);
        if (N > 1) fetch(1);
    }
    if (N > 1) __builtin_amdgcn_s_waitcnt(WAIT_DMA); else __builtin_amdgcn_s_waitcnt(0x0F70);   // chunk 0's weights are in LDS; chunk 1's fetch stays in flight
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        if (!late) {
            if (n + 1 < N) { commit(n + 1); dma_w(n + 1); }
            if (n + 2 < N) fetch(n + 2);
            taps(n, 0, HB_KSTEPS);
        } else {
            taps(n, 0, HB_KSTEPS / 2);
            if (n + 1 < N) { commit(n + 1); dma_w(n + 1); }
            if (n + 2 < N) fetch(n + 2);
            taps(n, HB_KSTEPS / 2, HB_KSTEPS);
        }
        if (++e_q == ncq) {                                          // the tile's last chunk: relu, store, statistics, fresh accumulator
            int x0, y0, z0;
            tile_origin(e_k, x0, y0, z0);
            e_q = 0; ++e_k;
            const int gx = x0 + lx + wx, gy = y0 + ly, gz = z0 + wz;
            float *orow = a.out + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.Cout;
            f32x16 v = acc;
            if (ha.in_absmax) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] *= post_scale;
            }
            if (a.relu) v = relu16(v);
            store_acc16(orow + co_blk * 32, v, kg);
            if (a.part || a.stat_out.acc) {
                // per-channel (sum, sumsq) over the wave's 32 voxels, added to the wave's running sums in LDS (tile order: fixed)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sm = half_wave_sum(v[r]), sq = half_wave_sum(v[r] * v[r]);
                    if (j == 31) { float *d = sred + wave * 64 + chan_of(r, kg) * 2; d[0] += sm; d[1] += sq; }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        }
        if (n + 2 < N) __builtin_amdgcn_s_waitcnt(WAIT_DMA); else __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    if (a.part || a.stat_out.acc) {
        if (threadIdx.x < 64) {
            float tsum = 0.0f;
            for (int w = 0; w < 2 * TZ; ++w) tsum += sred[w * 64 + threadIdx.x];
            if (a.part) a.part[(((size_t)b * ha.wgs_per_scene + wg) * a.Cout + co_blk * 32) * 2 + threadIdx.x] = tsum;
            if (a.stat_out.acc) gn_out_add(a.stat_out, b, (unsigned)wg, co_blk * 32, tsum);
        }
    }
}

#ifndef VT_HX_ABL
#define VT_HX_ABL 0     // ablations for timing only (wrong results): 1 = no staging pieces, 2 = no chunk barrier, 4 = no operand reads after a chunk's first
#endif
#ifdef VT_DIAG_HB
// diagnostic build only (tools/build_variant.sh hb "-DVT_DIAG_HB"; tools/diag_conv.py): per-wave shader-clock sums of the kernel's phases
// (slots 0..13), the wave's lifetime in 100 MHz ticks (14) and in shader-clock counts (15)
__device__ unsigned long long vt_diag_hb_buf[8192 * 16];
#define HB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         dg_sum[i] += t_ - dg_last; dg_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define HB_DIAG_BEGIN unsigned long long dg_sum[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dg_last = __builtin_amdgcn_s_memtime(); \
                      const unsigned long long dg_first = dg_last, dg_real = __builtin_amdgcn_s_memrealtime();
#define HB_DIAG_END(WAVES) do { if (lane == 0) { unsigned long long *d_ = vt_diag_hb_buf + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * (WAVES) + wave) * 16; \
                         for (int i_ = 0; i_ < 14; ++i_) d_[i_] = dg_sum[i_]; d_[14] = __builtin_amdgcn_s_memrealtime() - dg_real; \
                         d_[15] = __builtin_amdgcn_s_memtime() - dg_first; } } while (0)
#else
#define HB_STAMP(i) do { } while (0)
#define HB_DIAG_BEGIN
#define HB_DIAG_END(WAVES) do { } while (0)
#endif

// ---- the same kernel with specialised waves ------------------------------------------------------------------------
// In conv3d_gcr_h_kernel every wave alternates between its taps and its share of the next chunk's commit / fetch; the stamps
// show a wave spending 27 k cycles of a 91 k-cycle layer in taps and most of the rest in phases that a wave busy with MFMAs
// issues slowly (commit 13 k, fetch issue 9 k, barrier skew 14 k).  Here the first TZ waves only run taps -- each owns a whole
// z-plane of the tile: two 4 x 8 patches, two accumulators fed by the same weight fragments (6 LDS reads per 6 MFMAs instead of
// 4 per 3) -- and the other TZ waves only load: commit chunk n+1, DMA its weights, fetch chunk n+2.  One barrier per chunk.
// Workgroup barrier that orders LDS traffic only: __syncthreads() is a full workgroup fence, and with LDS-DMA writes in flight the
// compiler implements its release half (also of a fence restricted to the local address space) as s_waitcnt vmcnt(0) -- draining the
// register prefetch of the chunks ahead at every chunk barrier, so that each iteration pays a full memory latency.  Here: a bare
// s_barrier behind lgkmcnt(0); the weights' DMA is waited for by hand (counted vmcnt) before it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);                            // lgkmcnt(0) alone: this wave's LDS writes and reads have completed
    asm volatile("" ::: "memory");                                 // (compiler: no LDS access moves across the barrier)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// FIN: the variant with the final 1x1x1 conv in the epilogue (its own instantiation: the extra registers must not cost the other
// layers their fourth wave per SIMD)
template <int TZ, bool FIN = false, bool XST = false>
__global__ void __launch_bounds__(hb_threads(TZ))
conv3d_gcr_hw_kernel(HbArgs ha) {
    constexpr int ROWS = hb_rows(TZ), NVOX = hb_nvox(TZ), THREADS = hb_threads(TZ), LTHREADS = 64 * TZ;
    constexpr int ITERS = (NVOX + LTHREADS - 1) / LTHREADS;        // loader items (a halo voxel x the chunk's eight channels) per thread
    constexpr int IMG = (int)hb_img_bytes(TZ), WBUF = HB_WFRAGS * 16;
    extern __shared__ __attribute__((aligned(16))) char hl[];      // [2 images][2 weight buffers][stats scratch][scale / shift]
    const ConvArgs &a = ha.c;
    const Src &s = a.s;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int b = blockIdx.x / ha.wgs_per_scene, wg = blockIdx.x - b * ha.wgs_per_scene;
    const int nsp = a.tiles_x * a.tiles_y * a.tiles_z;
    const int ntile = (nsp - wg + ha.wgs_per_scene - 1) / ha.wgs_per_scene;
    const int Cin = s.C1 + s.C2, ncq = Cin / 8;
    const int co_blk = blockIdx.y, nco_all = a.Cout / 32;
    char *wbase = hl + 2 * IMG;
    float *sred = reinterpret_cast<float *>(hl + 2 * IMG + 2 * WBUF);
    float *ssl = sred + 2 * TZ * 64;
    float pre_scale = 1.0f, post_scale = 1.0f;
    if (ha.in_absmax) {
        const float m = *ha.in_absmax;
        if (m > 0.0f && m < 3.0e38f) {
            const int e = 10 - ilogbf(m);
            pre_scale = ldexpf(1.0f, e < -100 ? -100 : (e > 100 ? 100 : e));
            post_scale = 1.0f / pre_scale;
        }
    }
    // GnIn: the statistics are requested here and reduced (gn_in_finish, scratch: the first image buffer) behind the loaders' first
    // requests, so that the two round trips overlap
    HB_DIAG_BEGIN
    const bool stats_in = a.stat_in.acc[0] != nullptr;
    constexpr int GN_PRE = TZ == 8 ? 4 : 8;                         // (1024 threads: 4 words each cover the shipped rows, and 8 spill)
    GnReq<GN_PRE> gn_rq;
    if (stats_in) gn_rq = gn_in_request<GN_PRE>(a.stat_in, b, threadIdx.x, THREADS);
    else
        for (int i = threadIdx.x; i < 2 * Cin; i += THREADS)       // (the per-quad layout: [4 scales][4 shifts])
            ssl[gn_slot<true>(i >> 1, i & 1)] = pre_scale * (a.scale_shift ? a.scale_shift[(size_t)b * Cin * 2 + i] : ((i & 1) ? 0.0f : 1.0f));
    if (wave < TZ) sred[wave * 64 + lane] = 0.0f;
    __syncthreads();
    // ---- tiles to skip (ha.tile_skip): the workgroups of a scene deal the tiles that need their taps among themselves in tile order, and
    // likewise the tiles whose output is a constant per border class; without flags a workgroup walks wg, wg + wgs, ... as before
    float *ttab = ssl + HB_MAX_CIN * 2, *ktab = ttab + 27 * 32;      // [27 taps][32 couts], [27 border classes][32 couts]
    unsigned short *list_ne = reinterpret_cast<unsigned short *>(ktab + 27 * 32), *list_e = list_ne + HB_SKIP_LIST;
    int *lcnt = reinterpret_cast<int *>(list_e + HB_SKIP_LIST);      // [0..31] per-wave counts of a round, [32], [33] running totals
    float *grp = reinterpret_cast<float *>(lcnt + 40);               // [8 groups][mean, rstd] of the input's GroupNorm (cls_in)
    const bool sparse = !FIN && ha.tile_skip != nullptr;
    const bool cls_make = sparse && ha.cls_out != nullptr, cls_use = sparse && ha.cls_in != nullptr;
    int n_ne = ntile, n_e = 0;
    if (sparse) {
        const int t8x = s.W >> 3, t8y = s.H >> 3;
        const unsigned char *flags = ha.tile_skip + (size_t)b * (s.D >> 3) * t8y * t8x;
        if (threadIdx.x < 2) lcnt[32 + threadIdx.x] = 0;
        __syncthreads();
        for (int base = 0; base < nsp; base += THREADS) {
            const int t = base + threadIdx.x;
            bool valid = t < nsp, skip = false;
            if (valid) {
                const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, tz = t / (a.tiles_x * a.tiles_y);
                skip = (flags[(((tz * TZ) >> 3) * t8y + ty) * t8x + tx] & ha.skip_mask) != 0;
            }
            const unsigned long long m_ne = __ballot(valid && !skip), m_e = __ballot(valid && skip);
            if (lane == 0) { lcnt[wave] = __popcll(m_ne); lcnt[16 + wave] = __popcll(m_e); }
            __syncthreads();
            int r_ne = lcnt[32], r_e = lcnt[33];
            for (int w = 0; w < wave; ++w) { r_ne += lcnt[w]; r_e += lcnt[16 + w]; }
            const unsigned long long lt = (1ull << lane) - 1ull;
            r_ne += __popcll(m_ne & lt); r_e += __popcll(m_e & lt);
            if (valid && !skip && r_ne % ha.wgs_per_scene == wg && r_ne / ha.wgs_per_scene < HB_SKIP_LIST) list_ne[r_ne / ha.wgs_per_scene] = (unsigned short)t;
            // (dealt from the last workgroup backwards: the first ones hold the blocks with taps)
            if (valid && skip && r_e % ha.wgs_per_scene == ha.wgs_per_scene - 1 - wg && r_e / ha.wgs_per_scene < HB_SKIP_LIST) list_e[r_e / ha.wgs_per_scene] = (unsigned short)t;
            __syncthreads();
            if (threadIdx.x == 0) {
                int a_ne = 0, a_e = 0;
                for (int w = 0; w < THREADS / 64; ++w) { a_ne += lcnt[w]; a_e += lcnt[16 + w]; }
                lcnt[32] += a_ne; lcnt[33] += a_e;
            }
            __syncthreads();
        }
        const int tot_ne = lcnt[32], tot_e = lcnt[33];
        n_ne = tot_ne > wg ? (tot_ne - wg + ha.wgs_per_scene - 1) / ha.wgs_per_scene : 0;
        const int wge = ha.wgs_per_scene - 1 - wg;
        n_e = tot_e > wge ? (tot_e - wge + ha.wgs_per_scene - 1) / ha.wgs_per_scene : 0;
    }
    // cls_make: which workgroups compute class rows (unit u = index among the F sharing workgroups: cout u % 32 and every P-th class
    // from u / 32; where at least 32 workgroups of the scene have no tile with taps, those share the rows -- they have nothing else to
    // do -- else all do; the launcher asks for >= 32 workgroups per scene).  Only they pay the two barriers of that step.
    int cls_P = 0, cls_wi = 0;
    bool cls_unit = false;
    if (cls_make) {
        const int tot_ne = lcnt[32], spare = ha.wgs_per_scene - tot_ne;
        const int F = spare >= 32 ? spare : ha.wgs_per_scene;
        cls_wi = spare >= 32 ? wg - tot_ne : wg;
        cls_P = F >= 256 ? 8 : F >> 5;
        cls_unit = cls_wi >= 0 && cls_wi < 32 * cls_P;
    }
    auto tile_origin = [&](int k, int &x0, int &y0, int &z0) {
        int t = sparse ? (int)list_ne[k] : wg + k * ha.wgs_per_scene;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y; t /= a.tiles_y;
        x0 = tx * 8; y0 = ty * 8; z0 = t * TZ;
    };
    const int N = n_ne * ncq;

    if (wave >= TZ) {
        // ================================================ loader waves ================================================
        // (beside waves that issue MFMAs back to back every instruction of a loader wave -- vector, scalar or memory -- costs ~25-60
        // shader-clock counts, so the loaders are bounded by instruction COUNT: items of eight channels (two 16-byte requests, one
        // 16-byte row per plane: half the LDS stores), buffer loads whose per-chunk address is `per-tile voffset + scalar soffset`
        // (no address arithmetic per chunk; an offset in front of or behind the tensor returns zeros instead of faulting), the affine
        // table read as 16-byte vectors, LDS-DMA pieces addressed by a scalar base)
        const int lt = threadIdx.x - LTHREADS, lwave = wave - TZ;
        const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
        const unsigned nscene = gridDim.x / ha.wgs_per_scene;
        const __amdgpu_buffer_rsrc_t r_skip = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.skip), 0, nscene * (unsigned)(s.D * s.H * s.W) * (unsigned)s.C1 * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_low = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s.low ? s.low : s.skip), 0,
                                                                                s.low ? nscene * (unsigned)(D2 * H2 * W2) * (unsigned)s.C2 * 4u : 0u, 0x00020000);
        const unsigned lane16 = (unsigned)lane * 16u, lds0 = (unsigned)(size_t)hl;
        int d_q = 0;
        auto dma_w = [&](int n) {
            const int q = d_q;
            if (++d_q == ncq) d_q = 0;
            const char *wq = reinterpret_cast<const char *>(a.wp) + ((size_t)q * nco_all + co_blk) * (HB_WFRAGS * 16);
            const unsigned dst = lds0 + 2 * IMG + (n & 1) * WBUF;
            // a FIXED number of DMA instructions per wave (the last piece is issued twice by some waves: same bytes to the same place);
            // inline assembly: `scalar piece base + the lane's 32-bit offset` (through the builtin the compiler keeps a per-lane 64-bit
            // address of every piece in registers and adds one vector instruction per piece)
            constexpr int NP = HB_WFRAGS / 64, PW = (NP + TZ - 1) / TZ;
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int p = min(lwave + i * TZ, NP - 1);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + p * 1024), "v"(lane16), "s"(wq + p * 1024) : "memory");
            }
        };
        if (N > 0) dma_w(0);                                        // chunk 0's weights travel while the item tables are built
        // per item, once: its halo position, its LDS row, and its byte offsets from the tile origin in the full-resolution source and in
        // the half-resolution `low` source (tile origins are multiples of 8, so the halving distributes)
        int pxyz[ITERS], lrow[ITERS], voff[ITERS], loff[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int v = lt + it * LTHREADS;
            const int px = v % 10, r2 = v / 10, py = r2 % 10, pz = r2 / 10;
            pxyz[it] = px | (py << 8) | (pz << 16);
            lrow[it] = v < NVOX ? ((pz * 10 + py) * HB_PX + px) * 16 : -1;
            voff[it] = (((pz - 1) * s.H + (py - 1)) * s.W + (px - 1)) * s.C1 * 4;
            loff[it] = ((((pz - 1) >> 1) * H2 + ((py - 1) >> 1)) * W2 + ((px - 1) >> 1)) * s.C2 * 4;
        }
        unsigned f_full = 0;                                       // the items this thread has at all (lrow >= 0)
#pragma unroll
        for (int it = 0; it < ITERS; ++it) if (lrow[it] >= 0) f_full |= 1u << it;
        // two register sets for the input prefetch: chunk n lives in set n & 1 and is requested two chunks ahead
        struct PreSet { f32x4 v[2 * ITERS]; unsigned in; };
        PreSet preA, preB;
        int f_q = 0, f_k = 0;
        unsigned vo[ITERS], lo[ITERS];                             // this tile's byte offsets of the items (wrapped when in front of the tensor)
        unsigned f_in = 0;                                         // which items of the current tile lie inside the volume
        auto enter_tile = [&](int k) {                             // once per tile: origin (wave-uniform), offsets and the inside mask
            int x0, y0, z0;
            tile_origin(k, x0, y0, z0);
            const unsigned vbase = (unsigned)((((b * s.D + z0) * s.H + y0) * s.W + x0) * s.C1) * 4u;
            const unsigned lbase = (unsigned)((((b * D2 + (z0 >> 1)) * H2 + (y0 >> 1)) * W2 + (x0 >> 1)) * s.C2) * 4u;
            f_in = 0;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                vo[it] = vbase + (unsigned)voff[it];
                lo[it] = lbase + (unsigned)loff[it];
                const int gx = x0 + (pxyz[it] & 255) - 1, gy = y0 + ((pxyz[it] >> 8) & 255) - 1, gz = z0 + (pxyz[it] >> 16) - 1;
                if (lrow[it] >= 0 && gx >= 0 && gx < s.W && gy >= 0 && gy < s.H && gz >= 0 && gz < s.D) f_in |= 1u << it;
            }
        };
        enter_tile(0);
        HB_STAMP(9);
        auto fetch = [&](PreSet &ps) {
            const int ch = f_q * 8;
            ps.in = f_in;
            // every wave issues exactly 2 ITERS loads per chunk (the counted vmcnt wait relies on it); halo voxels outside the volume
            // but inside the tensor read a neighbouring row and are zeroed at the commit
            if (ch >= s.C1) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    ps.v[2 * it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_low, lo[it], (ch - s.C1) * 4, 0));
                    ps.v[2 * it + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_low, lo[it] + 16, (ch - s.C1) * 4, 0));
                }
            } else {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    ps.v[2 * it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_skip, vo[it], ch * 4, 0));
                    ps.v[2 * it + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_skip, vo[it] + 16, ch * 4, 0));
                }
            }
            if (++f_q == ncq) { f_q = 0; ++f_k; enter_tile(f_k); }
        };
        int c_q = 0;
        auto commit_as = [&](int n, const PreSet &ps, auto padded_tag) {
            constexpr bool PADDED = decltype(padded_tag)::value;
            const f32x4 *tq = reinterpret_cast<const f32x4 *>(ssl) + c_q * 4;        // the chunk's two channel quads: [4 scales][4 shifts] each
            if (++c_q == ncq) c_q = 0;
            const f32x4 sc[2] = {tq[0], tq[2]}, sh[2] = {tq[1], tq[3]};
            char *img = hl + (n & 1) * IMG;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                if (lrow[it] < 0) continue;
                float x[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * h + e] = fmaf(ps.v[2 * it + h][e], sc[h][e], sh[h][e]);
                if constexpr (PADDED) {
                    if (!(ps.in >> it & 1u)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = 0.0f;
                    }
                }
                // hi = half(x) (packed conversion), lo = half(x - hi) with the subtraction reading the half in place (v_fma_mix_f32):
                // 8 instructions per 4 values
                unsigned hw_[4], lw_[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float t0, t1, t2, t3;
                    asm("v_cvt_pk_f16_f32 %0, %8, %9\n\t"
                        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"
                        "v_fma_mix_f32 %4, %0, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
                        "v_fma_mix_f32 %5, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                        "v_fma_mix_f32 %6, %1, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
                        "v_fma_mix_f32 %7, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                        "v_cvt_pk_f16_f32 %2, %4, %5\n\t"
                        "v_cvt_pk_f16_f32 %3, %6, %7"
                        : "=&v"(hw_[2 * h]), "=&v"(hw_[2 * h + 1]), "=&v"(lw_[2 * h]), "=&v"(lw_[2 * h + 1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                        : "v"(x[4 * h]), "v"(x[4 * h + 1]), "v"(x[4 * h + 2]), "v"(x[4 * h + 3]));
                }
                *reinterpret_cast<u32x4 *>(img + lrow[it]) = u32x4{hw_[0], hw_[1], hw_[2], hw_[3]};
                *reinterpret_cast<u32x4 *>(img + ROWS * 16 + lrow[it]) = u32x4{lw_[0], lw_[1], lw_[2], lw_[3]};
            }
            __builtin_amdgcn_sched_barrier(0);                      // (the requests into these registers stay behind the commit)
        };
        // tiles inside the volume (216 of the 512 of a 64^3 level) need no zero padding: one wave-uniform test per chunk saves the selects
        auto commit = [&](int n, const PreSet &ps) {
            if (__builtin_amdgcn_ballot_w64(ps.in != f_full) == 0ull) commit_as(n, ps, std::false_type{});
            else commit_as(n, ps, std::true_type{});
        };
        constexpr int REQ = 2 * ITERS;
        constexpr int WAIT_DMA = 0x0F70 | (REQ > 15 ? 15 : REQ);  // vmcnt(REQ): all but the youngest request (the register fetch) have landed
        // (all workgroups of a launch start together, and what they request first comes in at ~11 bytes per cycle and CU: chunk 0 and its
        // weights travel alone; chunk 1's request follows chunk 0's commit -- VT_CONV_EARLY_REQ: the former order, for A/B)
        if (N > 0) {
            fetch(preA);
#ifdef VT_CONV_EARLY_REQ
            if (N > 1) fetch(preB);
#endif
        }
        HB_STAMP(10);
        if (stats_in) gn_in_finish<true, GN_PRE, true>(a.stat_in, gn_rq, b, pre_scale, ssl, hl, threadIdx.x, THREADS, cls_use ? grp : nullptr);
        HB_STAMP(11);
        if (N > 0) {
            commit(0, preA);
#ifndef VT_CONV_EARLY_REQ
            if (N > 1) fetch(preB);
#endif
            if (N > 2) fetch(preA);
        }
        HB_STAMP(12);
        if (sparse && !cls_use) lds_barrier();                      // (the tap waves: T is complete, the class table follows)
        if (cls_unit) { lds_barrier(); lds_barrier(); }             // (... the class table is complete; the next layer's weights are in LDS)
        constexpr int WAIT_DMA0 = 0x0F70 | (2 * REQ > 15 ? 15 : 2 * REQ);
        if (N > 2) __builtin_amdgcn_s_waitcnt(WAIT_DMA0); else __builtin_amdgcn_s_waitcnt(0x0F70);   // chunk 0's weights have landed
        HB_STAMP(13);
        lds_barrier();
        HB_STAMP(0);
        // iteration n: DMA chunk n+1's weights, commit chunk n+1 (requested two iterations ago), request chunk n+3 into the freed set;
        // the counted wait lets that youngest request fly on and, vmcnt being in order, also covers the DMA and chunk n+2's request
        auto iteration = [&](int n, PreSet &ps) {
            if (n + 1 < N) { dma_w(n + 1); commit(n + 1, ps); HB_STAMP(1); }
            if (n + 3 < N) {
                fetch(ps);
                HB_STAMP(2);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(WAIT_DMA);
            } else __builtin_amdgcn_s_waitcnt(0x0F70);
            HB_STAMP(3);
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
            HB_STAMP(4);
        };
        for (int n = 0; n < N; n += 2) {
            iteration(n, preB);                                    // chunk n + 1 is odd
            if (n + 1 < N) iteration(n + 1, preA);
        }
        // the layer behind a flagged one: the class table of constant tile k (its 27 classes, two voxels deep at a face of the volume)
        // from the class rows and this layer's group statistics, while the tap waves finish the tiles before it; two tables (K's and
        // T's place) alternate, one barrier per tile
        if (cls_use)
            for (int k = 0; k < n_e; ++k) {
                int t = list_e[k];
                const int tx = t % a.tiles_x; t /= a.tiles_x;
                const int ty = t % a.tiles_y; t /= a.tiles_y;
                const int x0 = tx * 8, y0 = ty * 8, z0 = t * TZ;
                const bool lox = x0 == 0, hix = x0 + 8 == s.W, loy = y0 == 0, hiy = y0 + 8 == s.H, loz = z0 == 0, hiz = z0 + TZ == s.D;
                float *tab = (k & 1) ? ttab : ktab;
                for (int id = lt; id < 27 * 32; id += LTHREADS) {
                    const int cls = id >> 5, co = id & 31, lz = cls / 9, ly = (cls / 3) % 3, lxx = cls % 3;
                    const int gz = loz ? lz : hiz ? 4 - lz : 2, gy = loy ? ly : hiy ? 4 - ly : 2, gx = lox ? lxx : hix ? 4 - lxx : 2;
                    const f32x4 *row = reinterpret_cast<const f32x4 *>(ha.cls_in + ((size_t)b * HB_CLS * 32 + ((gz * 5 + gy) * 5 + gx) * 32 + co) * HB_CLS_ROW);
                    const f32x4 A0 = row[0], A1 = row[1], C0 = row[2], C1 = row[3], B0 = row[4], B1 = row[5];
                    float kk = 0.0f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        kk += fmaf(grp[2 * g + 1], fmaf(-grp[2 * g], C0[g], A0[g]), B0[g]);
                        kk += fmaf(grp[2 * g + 9], fmaf(-grp[2 * g + 8], C1[g], A1[g]), B1[g]);
                    }
                    tab[id] = kk;
                }
                lds_barrier();
            }
    } else {
        // ================================================= tap waves ==================================================
        const int lx = j & 3, ly = j >> 2;
        const int center = ((wave + 1) * 10 + (ly + 1)) * HB_PX + (lx + 1);      // patch 0; patch 1 sits 4 voxels along x
        if (stats_in) gn_in_finish<true, GN_PRE, true>(a.stat_in, gn_rq, b, pre_scale, ssl, hl, threadIdx.x, THREADS, cls_use ? grp : nullptr);
        if (sparse && !cls_use && (n_e > 0 || cls_unit)) {
            // T[tap][cout] = sum_cin W[cout][cin][tap] shift[cin] from the packed fragments (hi + lo), while the loaders stage chunk 0
            for (int id = threadIdx.x; id < 27 * 32; id += LTHREADS) {
                const int tap = id >> 5, co = id & 31, fl = (tap >> 1) * 128 + co + 32 * (tap & 1);
                float t = 0.0f;
                for (int q = 0; q < ncq; ++q) {
                    const f16x8 *wq = reinterpret_cast<const f16x8 *>(a.wp) + ((size_t)q * nco_all + co_blk) * HB_WFRAGS;
                    const f16x8 wh = wq[fl], wl = wq[fl + 64];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t = fmaf((float)wh[e] + (float)wl[e], ssl[(2 * q + (e >> 2)) * 8 + 4 + (e & 3)], t);   // (shift of channel 8 q + e)
                }
                ttab[id] = t;
            }
        }
        if (sparse && !cls_use) {
            // K[class][cout] = the sum of T over the taps a voxel of that border class has inside the volume (class = per axis: first
            // voxel, inner, last voxel)
            lds_barrier();
            if (n_e > 0 || cls_unit)
                for (int id = threadIdx.x; id < 27 * 32; id += LTHREADS) {
                    const int cls = id >> 5, co = id & 31, cz = cls / 9, cy = (cls / 3) % 3, cx = cls % 3;
                    float k = 0.0f;
                    for (int tap = 0; tap < 27; ++tap) {
                        const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
                        const bool in = !(cz == 0 && dz < 0) && !(cz == 2 && dz > 0) && !(cy == 0 && dy < 0) && !(cy == 2 && dy > 0) &&
                                        !(cx == 0 && dx < 0) && !(cx == 2 && dx > 0);
                        k += in ? ttab[tap * 32 + co] : 0.0f;
                    }
                    ktab[id] = k;
                }
        }
        if (cls_unit) {
            // the class rows of the layer behind this one (HbArgs::cls_out): the unit's 27 x 32 weights go through LDS once (T's place: T is
            // dead), then the rows
            lds_barrier();
            const int P = cls_P, co = cls_wi & 31, part = cls_wi >> 5;
            const bool unit = true;
            if (unit && threadIdx.x < 27 * 4) {
                const int tap = threadIdx.x >> 2, q = threadIdx.x & 3, fl = (tap >> 1) * 128 + co + 32 * (tap & 1);
                const f16x8 *w2 = reinterpret_cast<const f16x8 *>(ha.cls_w) + (size_t)q * HB_WFRAGS;
                const f16x8 wh = w2[fl], wl = w2[fl + 64];
#pragma unroll
                for (int e = 0; e < 8; ++e) ttab[tap * 32 + 8 * q + e] = (float)wh[e] + (float)wl[e];
            }
            lds_barrier();
            // class of the neighbour at offset d of a voxel of axis class c (0, 1, inner, R - 2, R - 1): first / inner / last, -1: outside
            auto nb = [](int c, int d) { return c == 0 ? (d < 0 ? -1 : d) : c == 1 ? (d < 0 ? 0 : 1) : c == 2 ? 1 : c == 3 ? (d > 0 ? 2 : 1) : (d < 0 ? 1 : d == 0 ? 2 : -1); };
            const int nrows = unit ? (HB_CLS - part + P - 1) / P : 0;
            // four lanes per (row, group): one per z offset of the taps (the fourth idles), summed with two lane exchanges -- a lone
            // wave issues an instruction every ~6 cycles here, so the work is spread over every tap wave
            for (int item = threadIdx.x; item < nrows * 32; item += LTHREADS) {
                const int d = item & 3, g = (item >> 2) & 7, c2 = part + P * (item >> 5), cz = c2 / 25, cy = (c2 / 5) % 5, cx = c2 % 5;
                const f32x4 gm = *reinterpret_cast<const f32x4 *>(ha.cls_gamma + 4 * g), bt = *reinterpret_cast<const f32x4 *>(ha.cls_beta + 4 * g);
                const int nz = d < 3 ? nb(cz, d - 1) : -1;
                float A = 0.0f, C = 0.0f, Bv = 0.0f;
                if (nz >= 0) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int ny = nb(cy, dy - 1);
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int nx = nb(cx, dx - 1);
                            if (ny < 0 || nx < 0) continue;
                            const f32x4 k1 = *reinterpret_cast<const f32x4 *>(ktab + ((nz * 3 + ny) * 3 + nx) * 32 + 4 * g);
                            const f32x4 w = *reinterpret_cast<const f32x4 *>(ttab + (d * 9 + dy * 3 + dx) * 32 + 4 * g);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float wgm = w[e] * gm[e];
                                A = fmaf(wgm, fmaxf(k1[e], 0.0f), A);
                                C += wgm;
                                Bv = fmaf(w[e], bt[e], Bv);
                            }
                        }
                    }
                }
                A += __shfl_xor(A, 1); C += __shfl_xor(C, 1); Bv += __shfl_xor(Bv, 1);
                A += __shfl_xor(A, 2); C += __shfl_xor(C, 2); Bv += __shfl_xor(Bv, 2);
                if (d == 0) {
                    float *o = ha.cls_out + ((size_t)b * HB_CLS * 32 + c2 * 32 + co) * HB_CLS_ROW;
                    o[g] = A; o[8 + g] = C; o[16 + g] = Bv;
                }
            }
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        struct Ops { f16x8 wh, wl, x0h, x0l, x1h, x1l; };
        auto ops_of = [&](int n, int ks) {
            Ops o;
            const f16x8 *wl = reinterpret_cast<const f16x8 *>(wbase + (n & 1) * WBUF);
            o.wh = wl[ks * 128 + lane]; o.wl = wl[ks * 128 + 64 + lane];
            const int t0 = 2 * ks, t1 = 2 * ks + 1 < 27 ? 2 * ks + 1 : 13;
            const int r0 = ((t0 / 9 - 1) * 10 + ((t0 / 3) % 3 - 1)) * HB_PX + (t0 % 3 - 1);
            const int r1 = ((t1 / 9 - 1) * 10 + ((t1 / 3) % 3 - 1)) * HB_PX + (t1 % 3 - 1);
            const char *xin = hl + (n & 1) * IMG + (center + (kg ? r1 : r0)) * 16;
            o.x0h = *reinterpret_cast<const f16x8 *>(xin);
            o.x0l = *reinterpret_cast<const f16x8 *>(xin + ROWS * 16);
            o.x1h = *reinterpret_cast<const f16x8 *>(xin + 64);
            o.x1l = *reinterpret_cast<const f16x8 *>(xin + ROWS * 16 + 64);
            return o;
        };
        int e_q = 0, e_k = 0;
        // relu, store, statistics of one finished tile (its two patches in acc0 / acc1)
        auto finish_tile = [&](int x0, int y0, int z0) {
            f32x16 ssum, ssq;
#pragma unroll
            for (int pch = 0; pch < 2; ++pch) {
                f32x16 v = pch ? acc1 : acc0;
                const int gx = x0 + lx + 4 * pch, gy = y0 + ly, gz = z0 + wave;
                const size_t vox = (((size_t)b * s.D + gz) * s.H + gy) * s.W + gx;
                float *orow = a.out + vox * a.Cout;
                f32x16 xv;
                if constexpr (XST) xv = load_acc16(ha.stat_x + vox * a.Cout + co_blk * 32, kg);     // (requested ahead of the stores)
                if (ha.in_absmax) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] *= post_scale;
                }
                if (a.relu) v = relu16(v);
                if (FIN) {                                      // the final 1x1x1 conv on the registers; no statistics: nothing reads them
                    const SplitP<2> xs = split16<false, 2>(v);
                    f32x16 o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = ha.fin_b ? ha.fin_b[chan_of(r, kg)] : 0.0f;
                    o = dense32s<2>(o, ha.fin_w, xs, lane);
                    store_acc16(orow, o, kg);
                    if (ha.fin_y) store_acc16(ha.fin_y + vox * a.Cout, v, kg);
                    continue;
                }
                store_acc16(orow + co_blk * 32, v, kg);
#pragma unroll
                for (int r = 0; r < 16; ++r) {                  // the two patches' contributions per lane first: ONE lane reduction per tile
                    const float w2 = XST ? xv[r] : v[r];
                    ssum[r] = pch ? ssum[r] + v[r] : v[r];
                    ssq[r] = pch ? fmaf(v[r], w2, ssq[r]) : v[r] * w2;
                }
            }
            HB_STAMP(2);
            if ((a.part || a.stat_out.acc) && !FIN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sm = half_wave_sum(ssum[r]), sq = half_wave_sum(ssq[r]);
                    // (ds_add_f32 without a return value: the read-modify-write chains `d[0] += sm` compiled to were 32 dependent LDS
                    // round trips per tile; the slot belongs to this wave alone, whose LDS operations complete in order)
                    if (j == 31) { float *d = sred + wave * 64 + chan_of(r, kg) * 2; atomicAdd(d, sm); atomicAdd(d + 1, sq); }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
            HB_STAMP(3);
        };
        lds_barrier();                                           // the loaders' prologue
        HB_STAMP(0);
        for (int n = 0; n < N; ++n) {
            Ops cur = ops_of(n, 0);
#pragma unroll
            for (int ks = 0; ks < HB_KSTEPS; ++ks) {
                Ops nxt = cur;
                if (ks + 1 < HB_KSTEPS) nxt = ops_of(n, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl, cur.x0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl, cur.x1h, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x0l, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x1l, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x1h, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
            HB_STAMP(1);
            if (++e_q == ncq) {                                      // the tile's last chunk: relu, store, statistics, fresh accumulators
                int x0, y0, z0;
                tile_origin(e_k, x0, y0, z0);
                e_q = 0; ++e_k;
                finish_tile(x0, y0, z0);
            }
            lds_barrier();
            HB_STAMP(4);
        }
        // the tiles whose normalised input is the shift everywhere: every voxel takes the constant of its border class
        // (cls_use: the layer behind such a layer -- the tile's 27 classes, two voxels deep at a face of the volume, from the class rows
        // and this layer's group statistics)
#pragma unroll 1
        for (int k = 0; k < n_e; ++k) {
            int t = list_e[k];
            const int tx = t % a.tiles_x; t /= a.tiles_x;
            const int ty = t % a.tiles_y; t /= a.tiles_y;
            const int x0 = tx * 8, y0 = ty * 8, z0 = t * TZ;
            const bool lox = x0 == 0, hix = x0 + 8 == s.W, loy = y0 == 0, hiy = y0 + 8 == s.H, loz = z0 == 0, hiz = z0 + TZ == s.D;
            if (cls_use) lds_barrier();                             // (the loaders have built this tile's class table)
            const float *tab = (cls_use && (k & 1)) ? ttab : ktab;
#pragma unroll
            for (int pch = 0; pch < 2; ++pch) {
                const int gx = x0 + lx + 4 * pch, gy = y0 + ly, gz = z0 + wave;
                int cls;
                if (cls_use) {
                    const int cz = loz ? min(gz, 2) : hiz ? min(s.D - 1 - gz, 2) : 2, cy = loy ? min(gy, 2) : hiy ? min(s.H - 1 - gy, 2) : 2;
                    const int cx = lox ? min(gx, 2) : hix ? min(s.W - 1 - gx, 2) : 2;
                    cls = (cz * 3 + cy) * 3 + cx;
                } else cls = ((gz == 0 ? 0 : gz == s.D - 1 ? 2 : 1) * 3 + (gy == 0 ? 0 : gy == s.H - 1 ? 2 : 1)) * 3 + (gx == 0 ? 0 : gx == s.W - 1 ? 2 : 1);
                const f32x4 *kr = reinterpret_cast<const f32x4 *>(tab + cls * 32 + 4 * kg);
                f32x16 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 kv = kr[2 * q];                     // channels chan_of(4 q .. 4 q + 3, kg) = 8 q + 4 kg + (0..3)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q + e] = kv[e];
                }
                if (pch) acc1 = v; else acc0 = v;
            }
            HB_STAMP(5);
            finish_tile(x0, y0, z0);
        }
    }
    if (sparse) lds_barrier();                                      // the statistics of the constant tiles are in before wave 0 reads them
    if ((a.part || a.stat_out.acc) && threadIdx.x < 64) {
        float tsum = 0.0f;
        for (int w = 0; w < TZ; ++w) tsum += sred[w * 64 + threadIdx.x];
        if (a.part) a.part[(((size_t)b * ha.wgs_per_scene + wg) * a.Cout + co_blk * 32) * 2 + threadIdx.x] = tsum;
        if (a.stat_out.acc) gn_out_add(a.stat_out, b, (unsigned)wg, co_blk * 32, tsum);
    }
    HB_DIAG_END(2 * TZ);
}

#include "unet3d_up.inc"

// ---- the same kernel with the support work in the tap waves' own instruction streams -----------------------------------------
// conv3d_gcr_hw_kernel's loader waves are what bounds it: beside waves that issue MFMAs back to back ANOTHER wave's vector
// instructions get about one issue slot per 32-cycle MFMA (a chunk's ~150 loader instructions take as long as the tap waves' 168
// MFMAs and then some: chunk period 7.5 k cycles against 5.4 k of matrix time), while a wave's OWN vector instructions, placed
// behind an MFMA in program order, issue in that MFMA's shadow.  Here there are no loader waves: TZ waves, each runs the taps of
// its z-plane (two patches, as before) and carries an eighth of the next chunk's commit, weight DMA and the request two chunks
// further on as one small piece per MFMA gap (2-4 vector instructions, pinned there by scheduling barriers).  Same LDS images, same
// MFMA order per accumulator: bit-identical output.

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int TZ, bool FIN = false>
__global__ void __launch_bounds__(64 * TZ)
conv3d_gcr_hx_kernel(HbArgs ha) {
    constexpr int ROWS = hb_rows(TZ), NVOX = hb_nvox(TZ), THREADS = 64 * TZ;
    constexpr int ITERS = (2 * NVOX + THREADS - 1) / THREADS;
    constexpr int IMG = (int)hb_img_bytes(TZ), WBUF = HB_WFRAGS * 16;
    constexpr int NP = HB_WFRAGS / 64, PW = (NP + TZ - 1) / TZ;        // weight DMA: 1-KiB pieces per chunk, per wave
    // the gaps of a chunk (14 k-steps x 6 MFMAs): commit item i in gaps [8 i, 8 i + 8), then the DMA pieces, then the requests
    constexpr int G_DMA = ITERS * 8 + 2, G_FETCH = G_DMA + PW + 2;
    static_assert(G_FETCH + ITERS <= HB_KSTEPS * 6, "conv3d_gcr_hx_kernel: more pieces than gaps");
    extern __shared__ __attribute__((aligned(16))) char hl[];      // [2 images][2 weight buffers][stats scratch][scale / shift]
    const ConvArgs &a = ha.c;
    const Src &s = a.s;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int b = blockIdx.x / ha.wgs_per_scene, wg = blockIdx.x - b * ha.wgs_per_scene;
    const int nsp = a.tiles_x * a.tiles_y * a.tiles_z;
    const int ntile = (nsp - wg + ha.wgs_per_scene - 1) / ha.wgs_per_scene;
    const int Cin = s.C1 + s.C2, ncq = Cin / 8;
    const int co_blk = blockIdx.y, nco_all = a.Cout / 32;
    char *wbase = hl + 2 * IMG;
    float *sred = reinterpret_cast<float *>(hl + 2 * IMG + 2 * WBUF);
    float *ssl = sred + 2 * TZ * 64;
    float pre_scale = 1.0f, post_scale = 1.0f;
    if (ha.in_absmax) {
        const float m = *ha.in_absmax;
        if (m > 0.0f && m < 3.0e38f) {
            const int e = 10 - ilogbf(m);
            pre_scale = ldexpf(1.0f, e < -100 ? -100 : (e > 100 ? 100 : e));
            post_scale = 1.0f / pre_scale;
        }
    }
    HB_DIAG_BEGIN
    const bool stats_in = a.stat_in.acc[0] != nullptr;
    constexpr int GN_PRE = 8;
    GnReq<GN_PRE> gn_rq;
    if (stats_in) gn_rq = gn_in_request<GN_PRE>(a.stat_in, b, threadIdx.x, THREADS);
    else
        for (int i = threadIdx.x; i < 2 * Cin; i += THREADS)
            ssl[i] = pre_scale * (a.scale_shift ? a.scale_shift[(size_t)b * Cin * 2 + i] : ((i & 1) ? 0.0f : 1.0f));
    sred[wave * 64 + lane] = 0.0f;
    __syncthreads();
    HB_STAMP(0);
    auto tile_origin = [&](int k, int &x0, int &y0, int &z0) {
        int t = wg + k * ha.wgs_per_scene;
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y; t /= a.tiles_y;
        x0 = tx * 8; y0 = ty * 8; z0 = t * TZ;
    };
    const int N = ntile * ncq;

    // ---- the wave's share of the staging: items lt + i THREADS (halo voxel, channel half), as the loader waves of the hw kernel ----
    const int lt = threadIdx.x;
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    int pxyz[ITERS], lrow[ITERS], voff[ITERS], loff[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int item = lt + it * THREADS, v = item >> 1;
        const int px = v % 10, r2 = v / 10, py = r2 % 10, pz = r2 / 10;
        pxyz[it] = v < NVOX ? (px | (py << 8) | (pz << 16)) : (0xFF | (0xFF << 8));      // no item: a position outside every volume
        // (a thread without an item writes its zeros to a padding row -- px = 10 of the first image row -- that no tap reads)
        lrow[it] = (v < NVOX ? ((pz * 10 + py) * HB_PX + px) * 16 : 10 * 16) + (item & 1) * 8;
        voff[it] = ((pz - 1) * s.H + (py - 1)) * s.W + (px - 1);
        loff[it] = (((pz - 1) >> 1) * H2 + ((py - 1) >> 1)) * W2 + ((px - 1) >> 1);
    }
    const int c4 = (lt & 1) * 4;
    const int vmax = (int)((size_t)gridDim.x / ha.wgs_per_scene * s.D * s.H * s.W) - 1, lmax = s.low ? vmax / 8 : 0;
    struct PreSet { f32x4 v[ITERS]; unsigned in; };
    PreSet preA, preB;
    int d_q = 0;
    int f_q = 0, f_k = 0, f_vbase = 0, f_lbase = 0;
    unsigned f_in = 0;
    auto enter_tile = [&](int k) {
        int x0, y0, z0;
        tile_origin(k, x0, y0, z0);
        f_vbase = ((b * s.D + z0) * s.H + y0) * s.W + x0;
        f_lbase = ((b * D2 + (z0 >> 1)) * H2 + (y0 >> 1)) * W2 + (x0 >> 1);
        f_in = 0;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int gx = x0 + (pxyz[it] & 255) - 1, gy = y0 + ((pxyz[it] >> 8) & 255) - 1, gz = z0 + (pxyz[it] >> 16) - 1;
            f_in |= (unsigned)(((unsigned)gx < (unsigned)s.W) & ((unsigned)gy < (unsigned)s.H) & ((unsigned)gz < (unsigned)s.D)) << it;
        }
    };
    enter_tile(0);
    // one item's request; the last of a chunk advances the (tile, channel chunk) counters
    auto fetch_item = [&](PreSet &ps, auto it_tag) {
        constexpr int it = decltype(it_tag)::value;
        if (f_q * 8 >= s.C1)                                        // wave-uniform: a chunk lies in one source
            ps.v[it] = *reinterpret_cast<const f32x4 *>(s.low + (f_q * 8 - s.C1) + c4 + (size_t)(unsigned)min(max(f_lbase + loff[it], 0), lmax) * (unsigned)s.C2);
        else
            ps.v[it] = *reinterpret_cast<const f32x4 *>(s.skip + f_q * 8 + c4 + (size_t)(unsigned)min(max(f_vbase + voff[it], 0), vmax) * (unsigned)s.C1);
        if constexpr (it == 0) ps.in = f_in;
        if constexpr (it == ITERS - 1) {
            if (++f_q == ncq) { f_q = 0; ++f_k; enter_tile(f_k); }
        }
    };
    auto fetch = [&](PreSet &ps) { static_for<0, ITERS>([&](auto it) { fetch_item(ps, it); }); };
    auto dma_piece = [&](int n, int q, int i) {
        const f16x8 *wq = reinterpret_cast<const f16x8 *>(a.wp) + ((size_t)q * nco_all + co_blk) * HB_WFRAGS;
        char *dst = wbase + (n & 1) * WBUF;
        const int p = min(wave + i * TZ, NP - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wq + p * 64 + lane),
                                         (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
    };
    auto dma_w = [&](int n) {
        const int q = d_q;
        if (++d_q == ncq) d_q = 0;
#pragma unroll
        for (int i = 0; i < PW; ++i) dma_piece(n, q, i);
    };
    int c_q = 0;
    // the commit of one item in eight steps of 2-4 vector instructions (GroupNorm affine with the zero padding after it; hi =
    // half(x), lo = half(x - hi): see the hw kernel)
    struct CommitRegs { float x0, x1, x2, x3, t0, t1, t2, t3; unsigned h01, h23, l01, l23; };
    auto commit_step = [&](CommitRegs &r, const PreSet &ps, const f32x4 &sc, const f32x4 &sh, char *img, auto it_tag, auto st_tag) {
        constexpr int it = decltype(it_tag)::value, st = decltype(st_tag)::value;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (st == 0) {
            const bool in = ps.in >> it & 1u;
            r.x0 = in ? fmaf(ps.v[it][0], sc[0], sh[0]) : 0.0f;
            r.x1 = in ? fmaf(ps.v[it][1], sc[1], sh[1]) : 0.0f;
        } else if constexpr (st == 1) {
            const bool in = ps.in >> it & 1u;
            r.x2 = in ? fmaf(ps.v[it][2], sc[2], sh[2]) : 0.0f;
            r.x3 = in ? fmaf(ps.v[it][3], sc[3], sh[3]) : 0.0f;
        } else if constexpr (st == 2) {
            asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                         : "=&v"(r.h01), "=&v"(r.h23) : "v"(r.x0), "v"(r.x1), "v"(r.x2), "v"(r.x3));
        } else if constexpr (st == 3) {
            asm volatile("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                         : "=&v"(r.t0), "=&v"(r.t1) : "v"(r.h01), "v"(r.x0), "v"(r.x1));
        } else if constexpr (st == 4) {
            asm volatile("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                         : "=&v"(r.t2), "=&v"(r.t3) : "v"(r.h23), "v"(r.x2), "v"(r.x3));
        } else if constexpr (st == 5) {
            asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                         : "=&v"(r.l01), "=&v"(r.l23) : "v"(r.t0), "v"(r.t1), "v"(r.t2), "v"(r.t3));
        } else if constexpr (st == 6) {
            *reinterpret_cast<u32x2 *>(img + lrow[it]) = u32x2{r.h01, r.h23};
        } else {
            *reinterpret_cast<u32x2 *>(img + ROWS * 16 + lrow[it]) = u32x2{r.l01, r.l23};
        }
    };
    auto chunk_scale = [&](f32x4 &sc, f32x4 &sh) {                  // the scale / shift of the next chunk to commit (this thread's four channels)
        const int ch = c_q * 8 + c4;
        if (++c_q == ncq) c_q = 0;
        const f32x4 *ss = reinterpret_cast<const f32x4 *>(ssl + ch * 2);
        const f32x4 s01 = ss[0], s23 = ss[1];
        sc = f32x4{s01[0], s01[2], s23[0], s23[2]};
        sh = f32x4{s01[1], s01[3], s23[1], s23[3]};
    };
    auto commit = [&](int n, const PreSet &ps) {                    // (prologue: chunk 0 in one go)
        f32x4 sc, sh;
        chunk_scale(sc, sh);
        char *img = hl + (n & 1) * IMG;
        static_for<0, ITERS>([&](auto it) {
            CommitRegs r;
            static_for<0, 8>([&](auto st) { commit_step(r, ps, sc, sh, img, it, st); });
        });
    };

    // ---- the wave's taps: z-plane `wave` of the tile, patches at x = 0..3 and 4..7 ----
    const int lx = j & 3, ly = j >> 2;
    const int center = ((wave + 1) * 10 + (ly + 1)) * HB_PX + (lx + 1);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    struct Ops { f16x8 wh, wl, x0h, x0l, x1h, x1l; };
    auto ops_of = [&](int n, int ks) {
        Ops o;
        const f16x8 *wl = reinterpret_cast<const f16x8 *>(wbase + (n & 1) * WBUF);
        o.wh = wl[ks * 128 + lane]; o.wl = wl[ks * 128 + 64 + lane];
        const int t0 = 2 * ks, t1 = 2 * ks + 1 < 27 ? 2 * ks + 1 : 13;
        const int r0 = ((t0 / 9 - 1) * 10 + ((t0 / 3) % 3 - 1)) * HB_PX + (t0 % 3 - 1);
        const int r1 = ((t1 / 9 - 1) * 10 + ((t1 / 3) % 3 - 1)) * HB_PX + (t1 % 3 - 1);
        const char *xin = hl + (n & 1) * IMG + (center + (kg ? r1 : r0)) * 16;
        o.x0h = *reinterpret_cast<const f16x8 *>(xin);
        o.x0l = *reinterpret_cast<const f16x8 *>(xin + ROWS * 16);
        o.x1h = *reinterpret_cast<const f16x8 *>(xin + 64);
        o.x1l = *reinterpret_cast<const f16x8 *>(xin + ROWS * 16 + 64);
        return o;
    };

    constexpr int WAIT_DMA = 0x0F70 | ITERS;                      // vmcnt(ITERS): all but the youngest ITERS operations (the register fetch) have landed
    HB_STAMP(11);
    if (N > 0) {
        dma_w(0);
        HB_STAMP(12);
        fetch(preA);
        HB_STAMP(13);
        if (N > 1) fetch(preB);
    }
    HB_STAMP(1);
    if (stats_in) gn_in_finish<true>(a.stat_in, gn_rq, b, pre_scale, ssl, hl, threadIdx.x, THREADS);
    HB_STAMP(2);
    if (N > 0) {
        commit(0, preA);
        HB_STAMP(3);
        if (N > 2) fetch(preA);
    }
    constexpr int WAIT_DMA0 = 0x0F70 | (2 * ITERS > 15 ? 15 : 2 * ITERS);
    if (N > 2) __builtin_amdgcn_s_waitcnt(WAIT_DMA0); else __builtin_amdgcn_s_waitcnt(0x0F70);
    HB_STAMP(4);
    lds_barrier();
    HB_STAMP(5);

    int e_q = 0, e_k = 0;
    // chunk n: its taps, with chunk n+1's commit (from ps, requested two chunks ago), its weight DMA and chunk n+3's request (into
    // ps again) in the gaps.  ALL: every one of those exists (the steady state: no branches in the gaps)
    auto body = [&](int n, PreSet &ps, auto all_tag) {
        constexpr bool ALL = decltype(all_tag)::value;
        const bool do_c = ALL || n + 1 < N, do_f = ALL || n + 3 < N;
        f32x4 sc = {0.0f, 0.0f, 0.0f, 0.0f}, sh = sc;
        int dq = 0;
        if (do_c) {
            chunk_scale(sc, sh);
            dq = d_q;
            if (++d_q == ncq) d_q = 0;
        }
        char *img = hl + ((n + 1) & 1) * IMG;
        CommitRegs cr;
        auto piece = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            if constexpr ((VT_HX_ABL & 1) != 0) {
            } else if constexpr (g < ITERS * 8) {
                if constexpr ((VT_HX_ABL & 8) == 0)
                if (do_c) commit_step(cr, ps, sc, sh, img, std::integral_constant<int, g / 8>{}, std::integral_constant<int, g % 8>{});
            } else if constexpr (g >= G_DMA && g < G_DMA + PW) {
                if constexpr ((VT_HX_ABL & 16) == 0)
                if (do_c) dma_piece(n + 1, dq, g - G_DMA);
            } else if constexpr (g >= G_FETCH && g < G_FETCH + ITERS) {
                if constexpr ((VT_HX_ABL & 32) == 0)
                if (do_f) fetch_item(ps, std::integral_constant<int, g - G_FETCH>{});
            }
        };
        Ops cur = ops_of(n, 0);
        static_for<0, HB_KSTEPS>([&](auto ks_tag) {
            constexpr int ks = decltype(ks_tag)::value;
            Ops nxt = cur;
            if constexpr (ks + 1 < HB_KSTEPS && (VT_HX_ABL & 4) == 0) nxt = ops_of(n, ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl, cur.x0h, acc0, 0, 0, 0);
            piece(std::integral_constant<int, ks * 6 + 0>{});
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl, cur.x1h, acc1, 0, 0, 0);
            piece(std::integral_constant<int, ks * 6 + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x0l, acc0, 0, 0, 0);
            piece(std::integral_constant<int, ks * 6 + 2>{});
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x1l, acc1, 0, 0, 0);
            piece(std::integral_constant<int, ks * 6 + 3>{});
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x0h, acc0, 0, 0, 0);
            piece(std::integral_constant<int, ks * 6 + 4>{});
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh, cur.x1h, acc1, 0, 0, 0);
            piece(std::integral_constant<int, ks * 6 + 5>{});
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        });
        HB_STAMP(6);
        if (++e_q == ncq) {                                          // the tile's last chunk: relu, store, statistics, fresh accumulators
            int x0, y0, z0;
            tile_origin(e_k, x0, y0, z0);
            e_q = 0; ++e_k;
            f32x16 ssum, ssq;
#pragma unroll
            for (int pch = 0; pch < 2; ++pch) {
                f32x16 v = pch ? acc1 : acc0;
                const int gx = x0 + lx + 4 * pch, gy = y0 + ly, gz = z0 + wave;
                float *orow = a.out + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.Cout;
                if (ha.in_absmax) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] *= post_scale;
                }
                if (a.relu) v = relu16(v);
                if (FIN) {
                    const SplitP<2> xs = split16<false, 2>(v);
                    f32x16 o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = ha.fin_b ? ha.fin_b[chan_of(r, kg)] : 0.0f;
                    o = dense32s<2>(o, ha.fin_w, xs, lane);
                    store_acc16(orow, o, kg);
                    if (ha.fin_y) store_acc16(ha.fin_y + (orow - a.out), v, kg);
                    continue;
                }
                store_acc16(orow + co_blk * 32, v, kg);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ssum[r] = pch ? ssum[r] + v[r] : v[r];
                    ssq[r] = pch ? fmaf(v[r], v[r], ssq[r]) : v[r] * v[r];
                }
            }
            HB_STAMP(7);
            if ((a.part || a.stat_out.acc) && !FIN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sm = half_wave_sum(ssum[r]), sq = half_wave_sum(ssq[r]);
                    // (inline asm: the compiler puts s_waitcnt vmcnt(0) in front of an LDS atomic while LDS-DMA is in flight -- here the
                    // next chunk's weights and two chunks of input requests, a full memory latency per tile)
                    if (j == 31) {
                        const unsigned d = (unsigned)(size_t)(sred + wave * 64 + chan_of(r, kg) * 2) & 0xFFFFFu;
                        asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2 offset:4" :: "v"(d), "v"(sm), "v"(sq) : "memory");
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
            HB_STAMP(8);
        }
        // chunk n+1's weights have landed (the stores of an epilogue count in vmcnt too: they are younger than the request, so the
        // counted wait only waits longer)
        if (do_f) __builtin_amdgcn_s_waitcnt(WAIT_DMA); else __builtin_amdgcn_s_waitcnt(0x0F70);
        HB_STAMP(9);
        if constexpr ((VT_HX_ABL & 2) == 0) lds_barrier();
        HB_STAMP(10);
    };
    int n = 0;
    for (; n + 4 < N; n += 2) {
        body(n, preB, std::true_type{});                           // chunk n + 1 is odd: set B
        body(n + 1, preA, std::true_type{});
    }
    for (; n < N; ++n) {
        if (n & 1) body(n, preA, std::false_type{}); else body(n, preB, std::false_type{});
    }
    if ((a.part || a.stat_out.acc) && threadIdx.x < 64) {
        float tsum = 0.0f;
        for (int w = 0; w < TZ; ++w) tsum += sred[w * 64 + threadIdx.x];
        if (a.part) a.part[(((size_t)b * ha.wgs_per_scene + wg) * a.Cout + co_blk * 32) * 2 + threadIdx.x] = tsum;
        if (a.stat_out.acc) gn_out_add(a.stat_out, b, (unsigned)wg, co_blk * 32, tsum);
    }
    HB_DIAG_END(TZ);
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

// max-pool and the pooled tensor's GroupNorm partial sums in one pass: the block / thread layout and the summation order of
// channel_stats_kernel over the pooled tensor (bit-identical partials), the eight inputs of a pooled value read where that
// kernel read the value (one launch and one 2-8 MB round trip less per encoder level)
__global__ void __launch_bounds__(256)
maxpool3d_cl_stats_kernel(const float *x, float *out, int D, int H, int W, int C, int nblk, float *part, GnOut stat_out) {
    __shared__ float red[8][32][2];
    const int b = blockIdx.y, blk = blockIdx.x;
    const int D2 = D / 2, H2 = H / 2, W2 = W / 2;
    const size_t V = (size_t)D2 * H2 * W2;
    const size_t v0 = V * blk / nblk, v1 = V * (blk + 1) / nblk;
    const int c = threadIdx.x & 31, vg = threadIdx.x >> 5;
    const float *xb = x + (size_t)b * D * H * W * C;
    float *ob = out + (size_t)b * V * C;
    for (int cb = 0; cb < C; cb += 32) {
        float sum = 0.0f, sq = 0.0f;
        for (size_t v = v0 + vg; v < v1; v += 8) {
            const int ox = (int)(v % W2), oy = (int)((v / W2) % H2), oz = (int)(v / ((size_t)W2 * H2));
            const float *p = xb + ((((size_t)2 * oz) * H + 2 * oy) * W + 2 * ox) * C + cb + c;
            const size_t sy = (size_t)W * C, sz = (size_t)H * W * C;
            float m = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[sy], p[sy + C]));
            m = fmaxf(m, fmaxf(fmaxf(p[sz], p[sz + C]), fmaxf(p[sz + sy], p[sz + sy + C])));
            ob[v * C + cb + c] = m;
            sum += m; sq = fmaf(m, m, sq);
        }
        red[vg][c][0] = sum; red[vg][c][1] = sq;
        __syncthreads();
        if (part && threadIdx.x < 32) {
            float a = 0.0f, q = 0.0f;
            for (int i = 0; i < 8; ++i) { a += red[i][c][0]; q += red[i][c][1]; }
            float *dst = part + (((size_t)b * nblk + blk) * C + cb + c) * 2;
            dst[0] = a; dst[1] = q;
        }
        if (stat_out.acc && threadIdx.x < 64) {                    // thread = channel * 2 + {sum, sumsq}: the same eight terms in the same order
            float a = 0.0f;
            for (int i = 0; i < 8; ++i) a += red[i][threadIdx.x >> 1][threadIdx.x & 1];
            gn_out_add(stat_out, b, (unsigned)blk, cb, a);
        }
        __syncthreads();
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

// The same 1x1x1 conv on the f32 matrix core for 32-channel multiples (the UNet3D's final layer): one wave per
// 32-voxel tile, lane (voxel, k-half) reads its half row of x as four 16-byte loads, the weight rows come from L1.
// 64 MFMA cycles per 4 KB of input: the kernel streams at memory speed (the scalar form above issues one load
// instruction per input channel and thread and runs at a quarter of that).
__global__ void __launch_bounds__(256)
conv1x1_mfma_kernel(const float *x, const float *w, const float *bias, float *out, int Cin, int Cout, size_t V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kh = lane >> 5;
    const int ncib = Cin / 32, ncob = Cout / 32;
    const size_t ntile = (V + 31) / 32;
    for (size_t tile = (size_t)blockIdx.x * 4 + wave; tile < ntile; tile += (size_t)gridDim.x * 4) {
        const size_t v = tile * 32 + j;
        const bool valid = v < V;
        for (int cob = 0; cob < ncob; ++cob) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias ? bias[cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.0f;
            for (int cib = 0; cib < ncib; ++cib) {
                f32x16 xv, wv;
#pragma unroll
                for (int r = 0; r < 16; ++r) xv[r] = 0.0f;
                if (valid) xv = load_frag16(x + v * Cin + cib * 32 + 16 * kh);
                wv = load_frag16(w + (size_t)(cob * 32 + j) * Cin + cib * 32 + 16 * kh);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc = mfma(wv[q], xv[q], acc);
            }
            if (valid) store_acc16(out + v * Cout + cob * 32, acc, kh);
        }
    }
}

bool src_ok(const Src &s, int B) {
    if (!s.skip || B <= 0 || s.D <= 0 || s.H <= 0 || s.W <= 0 || s.C1 <= 0 || (s.C1 & 31)) return false;
    if (s.low && (s.C2 <= 0 || (s.C2 & 31) || (s.D & 1) || (s.H & 1) || (s.W & 1))) return false;
    if (!s.low && s.C2 != 0) return false;
    return true;
}

static int conv_tile(int D, int H, int W, int waves, int &TX, int &TY, int &TZ) {
    TX = W >= 32 ? 32 : (W >= 16 ? 16 : (W >= 8 ? 8 : 4));
    const int rows_total = waves * (32 / TX);
    TY = rows_total < H ? rows_total : H;
    while (rows_total % TY) --TY;
    TZ = rows_total / TY;
    return ((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ);
}

// K-split mode: used when even one-wave tiles cannot fill the chip and there are >= 2 input blocks
static bool conv_use_ksplit(int B, int D, int H, int W, int nco, int ncib) {
    int TX, TY, TZ;
    return ncib >= 2 && (size_t)conv_tile(D, H, W, 1, TX, TY, TZ) * B * nco < 1024;
}

static int conv_waves(int B, int D, int H, int W, int nco) {
    // fewest blocks-per-launch that still fills the chip: 8 waves per workgroup when the volume is
    // large, down to 1 for the 8^3 / 16^3 levels
    for (int waves = 8; waves > 1; waves >>= 1) {
        int TX, TY, TZ;
        if ((size_t)conv_tile(D, H, W, waves, TX, TY, TZ) * B * nco >= 512) return waves;
    }
    return 1;
}

// split-bf16 kernel: dimensions in multiples of 8, 32-channel multiples, enough tiles
static bool conv_s_eligible(int B, int D, int H, int W, int Cin, int Cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || ((D | H | W) & 7) || Cin <= 0 || (Cin & 31) || Cout <= 0 || (Cout & 31)) return false;
    if ((size_t)B * D * H * W >= ((size_t)1 << 31)) return false;                    // 32-bit voxel indices
    // even the thin 8 x 8 x 2 tiles must give 64 workgroups (one per four CUs): below that (the 8^3 level of one scene)
    // the exact-f32 K-split kernel is as fast.  VTACO_CONV_THIN=0 keeps the thin tiles off (A/B runs).
    static const bool thin = !(getenv("VTACO_CONV_THIN") && getenv("VTACO_CONV_THIN")[0] == '0');
    if ((size_t)(D / 8) * (H / 8) * (W / 8) * B * (Cout / 32) >= 64) return true;
    return thin && (size_t)(D / 2) * (H / 8) * (W / 8) * B * (Cout / 32) >= 64;
}

// tile depth: 8^3 tiles when they give the chip enough workgroups, 8 x 8 x 2 tiles below that
static int conv_s_tz(int B, int D, int H, int W, int Cout) {
    // 8^3 tiles, one workgroup per CU, where they give every CU two rounds or more (the 64^3 level: 61 vs 60 us for either
    // kernel); the two-workgroups-per-CU kernel on 8x8x4 tiles between that and 64 tiles (the 32^3 level: 192->64 97 -> 74 us);
    // thin 8x8x2 tiles below.  VTACO_CONV_TZ=8 / 4 forces one of the first two (A/B runs).
    static const int forced = getenv("VTACO_CONV_TZ") ? atoi(getenv("VTACO_CONV_TZ")) : 0;
    const size_t tiles8 = (size_t)(D / 8) * (H / 8) * (W / 8) * B * (Cout / 32);
    if (tiles8 >= 64) return (forced == 8 || forced == 4) ? forced : (tiles8 >= 512 ? 8 : 4);
    return 2;
}

// K-split plan of the split-bf16 kernel for the levels whose output tiles cannot fill the chip (16^3 and 8^3 of one scene):
// returns the number of K slices (0: not applicable) and the tile depth.  8^3 tiles when they and up to 8 slices give >= 128
// workgroups (every 16-channel block of weights -- 54 KB -- is then fetched once per 512 voxels instead of once per 128),
// 8 x 8 x 2 tiles otherwise; slices = the power of two that brings the launch nearest to one workgroup per CU.
// VTACO_CONV_KSPLIT=0 turns the path off, VTACO_CONV_KSPLIT=<tz>,<ks> forces a plan where it is valid (A/B runs).
static int conv_sk_plan(int B, int D, int H, int W, int Cin, int Cout, int &tz) {
    tz = 0;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || ((D | H | W) & 7) || Cin <= 0 || (Cin & 31) || Cout <= 0 || (Cout & 31) || Cout > 1024) return 0;
    static const char *knob = getenv("VTACO_CONV_KSPLIT");
    if (knob && knob[0] == '0') return 0;
    const int ncq = Cin / 16;
    const size_t t8 = (size_t)(D / 8) * (H / 8) * (W / 8) * B * (Cout / 32), t2 = 4 * t8;
    if (t2 > 128) return 0;                                        // the plain kernels have the workgroups (one per two CUs or more)
    const bool forced = knob && strchr(knob, ',');
    if (ncq < 8 && !forced) return 0;                              // 64 input channels: four blocks -- the second launch costs what the split saves (measured)
    if (forced) {
        const int ftz = atoi(knob), fks = atoi(strchr(knob, ',') + 1);
        if ((ftz == 8 || ftz == 2) && fks >= 2 && fks <= 16 && ncq % fks == 0) { tz = ftz; return fks; }
    }
    int best = 0;
    for (int ks = 8; ks >= 2; ks >>= 1)
        if (ncq % ks == 0 && t8 * ks >= 128 && t8 * ks <= 256) { best = ks; break; }
    if (best) { tz = 8; return best; }
    for (int ks = 2; ks <= 8; ks <<= 1)
        if (ncq % ks == 0) { best = ks; if (t2 * ks >= 256) break; }
    if (best) tz = 2;
    return best;
}

// split-f16 persistent kernel: 8^3 tiles (16 waves) when they give every CU a workgroup, 8x8x4 tiles (8 waves) when those do;
// 0 = not covered (the 16^3-class levels stay on the thin-tile split-bf16 kernel or the f32 kernels)
static int conv_h_tz(int B, int D, int H, int W, int Cin, int Cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || ((D | H | W) & 7) || Cin <= 0 || (Cin & 31) || Cout <= 0 || (Cout & 31)) return 0;
    if ((size_t)B * D * H * W >= ((size_t)1 << 31) || Cin > HB_MAX_CIN) return 0;          // 32-bit voxel indices; the scale / shift table in LDS
    static const int forced = getenv("VTACO_CONV_HTZ") ? atoi(getenv("VTACO_CONV_HTZ")) : 0;     // A/B knob
    const size_t nco = Cout / 32, t8 = (size_t)(D / 8) * (H / 8) * (W / 8) * B * nco, t4 = 2 * t8, t2 = 4 * t8;
    if (forced == 8 || forced == 4 || forced == 2) return (forced == 8 ? t8 : forced == 4 ? t4 : t2) >= 64 ? forced : 0;
    if (t8 >= 256) return 8;
    if (t4 >= 128) return 4;
    static const bool thin = getenv("VTACO_CONV_HTHIN") != nullptr;      // A/B knob: 8x8x2 tiles (4 waves) for the 16^3-class levels
    if (thin && t2 >= 64) return 2;
    return 0;
}
static int conv_h_wgs_per_scene(int B, int D, int H, int W, int Cout, int tz) {
    const int nsp = (D / tz) * (H / 8) * (W / 8);
    int wgs = vt_num_cus() / (B * (Cout / 32));
    if (wgs < 1) wgs = 1;
    return wgs < nsp ? wgs : nsp;
}

template <int NCO, int WAVES>
static int conv_launch(const ConvArgs &a, dim3 grid, size_t lds, hipStream_t st) {
    bool attr_set = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr_set) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_kernel<NCO, WAVES>), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr: hipFuncSetAttribute");
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_gcr_kernel<NCO, WAVES>), grid, dim3(WAVES * 64), lds, st, a);
    return 0;
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

// partial-statistics buffers: floats needed for a tensor [B, D,H,W, C]
size_t vt_stats_floats(int B, int D, int H, int W, int C) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    // generous upper bound on the number of partial blocks a producer uses (conv tiles or stats chunks)
    const size_t V = (size_t)D * H * W;
    size_t nblk = (V + 31) / 32;
    if (nblk < 1024) nblk = 1024;
    return (size_t)B * nblk * C * 2 + 4;
}

// number of spatial partial blocks vt_conv3d_gcr writes per scene for this output shape
int vt_conv3d_stat_blocks(int B, int D, int H, int W, int Cin, int Cout) {
    int TX, TY, TZ;
    const int nco = Cout / 32;
    if (conv_use_ksplit(B, D, H, W, nco, Cin / 32)) return conv_tile(D, H, W, 1, TX, TY, TZ);
    return conv_tile(D, H, W, conv_waves(B, D, H, W, nco), TX, TY, TZ);
}

static int channel_stats_launch(const float *x, int B, int64_t V, int C, int nblk, float *part, const GnOut &stat_out, void *stream) {
    if (!x || (!part && !stat_out.acc) || B <= 0 || V <= 0 || C <= 0 || (C & 31) || nblk <= 0) return vt_fail(VT_ERR_INVALID, "vt_channel_stats: bad argument");
    hipLaunchKernelGGL(channel_stats_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, x, C, (size_t)V, nblk, part, stat_out);
    return vt_check(hipGetLastError(), "vt_channel_stats");
}

int vt_channel_stats(const float *x, int B, int64_t V, int C, int nblk, float *part, void *stream) {
    if (!part) return vt_fail(VT_ERR_INVALID, "vt_channel_stats: bad argument");
    return channel_stats_launch(x, B, V, C, nblk, part, GnOut{}, stream);
}

static int gn_scale_shift_launch(const float *part1, int nblk1, int C1, const float *part2, int nblk2, int C2,
                                 int B, int64_t voxels, int groups, const float *gamma, const float *beta, double eps,
                                 float *scale_shift, unsigned long long *zero, size_t zero_words, void *stream) {
    if (!part1 || nblk1 <= 0 || C1 <= 0 || !gamma || !beta || !scale_shift || B <= 0 || voxels <= 0)
        return vt_fail(VT_ERR_INVALID, "vt_gn_scale_shift: bad argument");
    if (!part2) { nblk2 = 0; C2 = 0; }
    const int C = C1 + C2;
    if (groups <= 0 || C % groups || C / groups > 256) return vt_fail(VT_ERR_INVALID, "vt_gn_scale_shift: bad group count");
    StatSrc s1{part1, nblk1, C1}, s2{part2, nblk2, C2};
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B, groups), dim3(256), 0, (hipStream_t)stream, s1, s2, groups,
                       (double)voxels, gamma, beta, (float)eps, scale_shift, zero, zero_words);
    return vt_check(hipGetLastError(), "vt_gn_scale_shift");
}

int vt_gn_scale_shift(const float *part1, int nblk1, int C1, const float *part2, int nblk2, int C2,
                      int B, int64_t voxels, int groups, const float *gamma, const float *beta, double eps,
                      float *scale_shift, void *stream) {
    return gn_scale_shift_launch(part1, nblk1, C1, part2, nblk2, C2, B, voxels, groups, gamma, beta, eps, scale_shift, nullptr, 0, stream);
}

int vt_conv3d_gcr(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                  const float *scale_shift, const float *packed_w, int Cout, int relu, float *out,
                  float *out_part, void *stream) {
    ConvArgs a;
    a.s = Src{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(a.s, B) || !packed_w || !out) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr: bad argument");
    if (Cout <= 0 || (Cout & 31)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr: Cout must be a multiple of 32");
    a.scale_shift = scale_shift; a.wp = packed_w; a.out = out; a.part = out_part; a.Cout = Cout; a.relu = relu;
    const int nco = Cout / 32;
    const int ncib = (a.s.C1 + a.s.C2) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (conv_use_ksplit(B, D, H, W, nco, ncib)) {
        const int nsp1 = conv_tile(D, H, W, 1, a.TX, a.TY, a.TZ);
        a.tiles_x = (W + a.TX - 1) / a.TX; a.tiles_y = (H + a.TY - 1) / a.TY; a.tiles_z = (D + a.TZ - 1) / a.TZ;
        const size_t one = (size_t)(a.TX + 2) * (a.TY + 2) * (a.TZ + 2) * CPAD * sizeof(float);
        const int kw = ncib >= 4 ? 4 : 2;
        size_t lds = one * kw;
        if (lds < (size_t)(kw - 1) * 16 * 64 * sizeof(float)) lds = (size_t)(kw - 1) * 16 * 64 * sizeof(float);
        const dim3 grid((unsigned)((size_t)nsp1 * B), (unsigned)nco);
        bool ks_attr = false;        // (vt_max_dyn_lds keeps the per-device record)
        if (!ks_attr) {
            hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_ksplit_kernel<4>), 160 * 1024);
            if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_ksplit_kernel<2>), 160 * 1024);
            if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr: hipFuncSetAttribute");
            ks_attr = true;
        }
        if (kw == 4) hipLaunchKernelGGL(conv3d_gcr_ksplit_kernel<4>, grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(conv3d_gcr_ksplit_kernel<2>, grid, dim3(128), lds, st, a);
        return vt_check(hipGetLastError(), "vt_conv3d_gcr");
    }
    const int waves = conv_waves(B, D, H, W, nco);
    const int nsp = conv_tile(D, H, W, waves, a.TX, a.TY, a.TZ);
    a.tiles_x = (W + a.TX - 1) / a.TX; a.tiles_y = (H + a.TY - 1) / a.TY; a.tiles_z = (D + a.TZ - 1) / a.TZ;
    size_t lds = (size_t)(a.TX + 2) * (a.TY + 2) * (a.TZ + 2) * CPAD * sizeof(float);
    const size_t spatial_blocks = (size_t)nsp * B;
    int per = 1;                                                   // cout blocks per workgroup: share the staged tile when
    if (waves == 8) {                                              // the launch has blocks to spare
        per = nco;
        while (per > 1 && (per > 4 || spatial_blocks * (nco / per) < 512 || nco % per)) per >>= 1;
    }
    const size_t stat_lds = (size_t)waves * per * 64 * sizeof(float);
    if (stat_lds > lds) lds = stat_lds;
    if (lds > 160 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr: tile does not fit LDS");
    const dim3 grid((unsigned)spatial_blocks, (unsigned)(nco / per));
    int rc = 0;
    if (waves == 8) rc = per == 4 ? conv_launch<4, 8>(a, grid, lds, st) : per == 2 ? conv_launch<2, 8>(a, grid, lds, st) : conv_launch<1, 8>(a, grid, lds, st);
    else if (waves == 4) rc = conv_launch<1, 4>(a, grid, lds, st);
    else if (waves == 2) rc = conv_launch<1, 2>(a, grid, lds, st);
    else rc = conv_launch<1, 1>(a, grid, lds, st);
    if (rc) return rc;
    return vt_check(hipGetLastError(), "vt_conv3d_gcr");
}

int vt_conv3d_pack_bf16x3(const float *w, int Cout, int Cin, float *packed, void *stream) {
    if (!w || !packed) return vt_fail(VT_ERR_INVALID, "vt_conv3d_pack_bf16x3: null argument");
    if (!vt_conv3d_packed_floats(Cout, Cin)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_pack_bf16x3: channel counts must be multiples of 32");
    const size_t frags = (size_t)27 * Cout * Cin / 4;              // 16-byte fragments: hi and lo of every 8 cin
    size_t g = (frags + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv3d_pack_s_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, packed, frags, 0);
    return vt_check(hipGetLastError(), "vt_conv3d_pack_bf16x3");
}

int vt_conv3d_stat_blocks_bf16x3(int B, int D, int H, int W, int Cin, int Cout) {
    return conv_s_eligible(B, D, H, W, Cin, Cout) ? (D / conv_s_tz(B, D, H, W, Cout)) * (H / 8) * (W / 8) : 0;
}

// half: the packed weights and the kernel's operands are IEEE-half pairs (vt_conv3d_pack_f16x3_thin) instead of bf16 pairs
static int conv_s_launch(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                         const float *scale_shift, const float *packed_w_bf16x3, int Cout, int relu, float *out,
                         float *out_part, const GnIn &stat_in, const GnOut &stat_out, void *stream, bool half = false) {
    ConvArgs a;
    a.stat_in = stat_in; a.stat_out = stat_out;
    a.s = Src{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(a.s, B) || !packed_w_bf16x3 || !out) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_bf16x3: bad argument");
    if (!conv_s_eligible(B, D, H, W, a.s.C1 + a.s.C2, Cout))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_bf16x3: shape not covered (see vt_conv3d_stat_blocks_bf16x3); use vt_conv3d_gcr");
    a.scale_shift = scale_shift; a.wp = packed_w_bf16x3; a.out = out; a.part = out_part; a.Cout = Cout; a.relu = relu;
    const int tz = conv_s_tz(B, D, H, W, Cout);
    a.TX = a.TY = 8; a.TZ = tz;
    a.tiles_x = W / 8; a.tiles_y = H / 8; a.tiles_z = D / tz;
    const dim3 grid((unsigned)((size_t)a.tiles_x * a.tiles_y * a.tiles_z * B), (unsigned)(Cout / 32));
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<8>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<2>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<8, false, true>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<2, false, true>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s4_kernel<false>), (int)S4_LDS);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s4_kernel<true>), (int)S4_LDS);
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_bf16x3: hipFuncSetAttribute");
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (tz == 4) {
        if (half) hipLaunchKernelGGL(conv3d_gcr_s4_kernel<true>, grid, dim3(S4_THREADS), S4_LDS, st, a);
        else hipLaunchKernelGGL(conv3d_gcr_s4_kernel<false>, grid, dim3(S4_THREADS), S4_LDS, st, a);
    } else if (tz == 8) {
        if (half) hipLaunchKernelGGL((conv3d_gcr_s_kernel<8, false, true>), grid, dim3(sb_threads(8)), sb_lds(8), st, a);
        else hipLaunchKernelGGL(conv3d_gcr_s_kernel<8>, grid, dim3(sb_threads(8)), sb_lds(8), st, a);
    } else {
        if (half) hipLaunchKernelGGL((conv3d_gcr_s_kernel<2, false, true>), grid, dim3(sb_threads(2)), sb_lds(2), st, a);
        else hipLaunchKernelGGL(conv3d_gcr_s_kernel<2>, grid, dim3(sb_threads(2)), sb_lds(2), st, a);
    }
    return vt_check(hipGetLastError(), "vt_conv3d_gcr_bf16x3");
}

int vt_conv3d_gcr_bf16x3(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                         const float *scale_shift, const float *packed_w_bf16x3, int Cout, int relu, float *out,
                         float *out_part, void *stream) {
    return conv_s_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_bf16x3, Cout, relu, out, out_part, GnIn{}, GnOut{}, stream);
}

// ---- the K-split form of vt_conv3d_gcr_bf16x3 for the thin levels (conv_sk_plan) ----
size_t vt_conv3d_ksplit_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout) {
    int tz;
    const int ks = conv_sk_plan(B, D, H, W, Cin, Cout, tz);
    return ks ? (size_t)ks * B * D * H * W * Cout * sizeof(float) : 0;
}

int vt_conv3d_stat_blocks_ksplit(int B, int D, int H, int W, int Cin, int Cout) {
    int tz;
    return conv_sk_plan(B, D, H, W, Cin, Cout, tz) ? (int)((size_t)D * H * W / KSUM_VOX) : 0;
}

static int conv_sk_launch(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                          const float *scale_shift, const float *packed_w_bf16x3, int Cout, int relu, float *out,
                          float *out_part, const GnIn &stat_in, const GnOut &stat_out, void *workspace, size_t workspace_bytes, void *stream,
                          bool half = false) {
    ConvArgs a;
    a.stat_in = stat_in;
    a.s = Src{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(a.s, B) || !packed_w_bf16x3 || !out || !workspace) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_bf16x3_ksplit: bad argument");
    int tz;
    const int ks = conv_sk_plan(B, D, H, W, a.s.C1 + a.s.C2, Cout, tz);
    if (!ks) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_bf16x3_ksplit: shape not covered (see vt_conv3d_ksplit_workspace_bytes)");
    const size_t slab = (size_t)B * D * H * W * Cout;
    if (workspace_bytes < slab * ks * sizeof(float)) return vt_fail(VT_ERR_WORKSPACE, "vt_conv3d_gcr_bf16x3_ksplit: workspace too small");
    a.scale_shift = scale_shift; a.wp = packed_w_bf16x3; a.out = nullptr; a.part = nullptr; a.Cout = Cout; a.relu = 0;
    a.kws = (float *)workspace; a.ksplit = ks;
    a.TX = a.TY = 8; a.TZ = tz;
    a.tiles_x = W / 8; a.tiles_y = H / 8; a.tiles_z = D / tz;
    const dim3 grid((unsigned)((size_t)a.tiles_x * a.tiles_y * a.tiles_z * B), (unsigned)(Cout / 32), (unsigned)ks);
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<8, true>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<2, true>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<8, true, true>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_s_kernel<2, true, true>), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_bf16x3_ksplit: hipFuncSetAttribute");
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (tz == 8 && half) hipLaunchKernelGGL((conv3d_gcr_s_kernel<8, true, true>), grid, dim3(sb_threads(8)), sb_lds(8), st, a);
    else if (tz == 8) hipLaunchKernelGGL((conv3d_gcr_s_kernel<8, true>), grid, dim3(sb_threads(8)), sb_lds(8), st, a);
    else if (half) hipLaunchKernelGGL((conv3d_gcr_s_kernel<2, true, true>), grid, dim3(sb_threads(2)), sb_lds(2), st, a);
    else hipLaunchKernelGGL((conv3d_gcr_s_kernel<2, true>), grid, dim3(sb_threads(2)), sb_lds(2), st, a);
    const dim3 sgrid((unsigned)((size_t)B * D * H * W / KSUM_VOX), (unsigned)(Cout / 32));
#define VT_KSUM(KS) hipLaunchKernelGGL(conv_ksum_kernel<KS>, sgrid, dim3(256), 0, st, (const float *)workspace, ks, slab, Cout, relu, out, out_part, \
                                       stat_out, (int)((size_t)D * H * W / KSUM_VOX))
    if (ks == 8) VT_KSUM(8); else if (ks == 4) VT_KSUM(4); else if (ks == 2) VT_KSUM(2); else VT_KSUM(0);
#undef VT_KSUM
    return vt_check(hipGetLastError(), "vt_conv3d_gcr_bf16x3_ksplit");
}

int vt_conv3d_gcr_bf16x3_ksplit(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                                const float *scale_shift, const float *packed_w_bf16x3, int Cout, int relu, float *out,
                                float *out_part, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_sk_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_bf16x3, Cout, relu, out, out_part, GnIn{}, GnOut{}, workspace, workspace_bytes, stream);
}

// the thin-level kernels (vt_conv3d_gcr_bf16x3 / _ksplit shapes) on IEEE-half pairs: same fragment layout, same sizes, same arguments
int vt_conv3d_pack_f16x3_thin(const float *w, int Cout, int Cin, float *packed, void *stream) {
    if (!w || !packed) return vt_fail(VT_ERR_INVALID, "vt_conv3d_pack_f16x3_thin: null argument");
    if (!vt_conv3d_packed_floats(Cout, Cin)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_pack_f16x3_thin: channel counts must be multiples of 32");
    const size_t frags = (size_t)Cout * Cin * 27 / 4;
    size_t g = (frags + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv3d_pack_s_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, packed, frags, 1);
    return vt_check(hipGetLastError(), "vt_conv3d_pack_f16x3_thin");
}

int vt_conv3d_gcr_f16x3_thin(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                             const float *scale_shift, const float *packed_w_f16x3_thin, int Cout, int relu, float *out,
                             float *out_part, void *stream) {
    return conv_s_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_f16x3_thin, Cout, relu, out, out_part, GnIn{}, GnOut{}, stream, true);
}

int vt_conv3d_gcr_f16x3_thin_ksplit(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                                    const float *scale_shift, const float *packed_w_f16x3_thin, int Cout, int relu, float *out,
                                    float *out_part, void *workspace, size_t workspace_bytes, void *stream) {
    return conv_sk_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_f16x3_thin, Cout, relu, out, out_part, GnIn{}, GnOut{},
                          workspace, workspace_bytes, stream, true);
}

size_t vt_conv3d_packed_floats_f16x3(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || (Cout & 31) || (Cin & 31)) return 0;
    return (size_t)(Cin / 8) * (Cout / 32) * HB_WFRAGS * 4;          // 14 k-steps (27 taps + one zero half-step) of 16-byte fragments
}

int vt_conv3d_pack_f16x3(const float *w, int Cout, int Cin, float *packed, void *stream) {
    if (!w || !packed) return vt_fail(VT_ERR_INVALID, "vt_conv3d_pack_f16x3: null argument");
    const size_t frags = vt_conv3d_packed_floats_f16x3(Cout, Cin) / 4;
    if (!frags) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_pack_f16x3: channel counts must be multiples of 32");
    size_t g = (frags + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv3d_pack_h_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, packed, frags);
    return vt_check(hipGetLastError(), "vt_conv3d_pack_f16x3");
}

int vt_conv3d_pack_f16x3_t(const float *w, int Cout, int Cin, float *packed, void *stream) {
    if (!w || !packed) return vt_fail(VT_ERR_INVALID, "vt_conv3d_pack_f16x3_t: null argument");
    const size_t frags = vt_conv3d_packed_floats_f16x3(Cin, Cout) / 4;          // the data-gradient conv maps Cout -> Cin channels
    if (!frags) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_pack_f16x3_t: channel counts must be multiples of 32");
    size_t g = (frags + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv3d_pack_h_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, packed, frags, 1);
    return vt_check(hipGetLastError(), "vt_conv3d_pack_f16x3_t");
}

int vt_conv3d_stat_blocks_f16x3(int B, int D, int H, int W, int Cin, int Cout) {
    const int tz = conv_h_tz(B, D, H, W, Cin, Cout);
    return tz ? conv_h_wgs_per_scene(B, D, H, W, Cout, tz) : 0;
}

int vt_conv3d_gcr_f16x3(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                        const float *scale_shift, const float *packed_w_f16x3, int Cout, int relu, float *out,
                        float *out_part, void *stream) {
    return vt_conv3d_gcr_f16x3_scaled(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_f16x3, Cout, relu, out, out_part, nullptr, stream);
}

// is the specialised-wave kernel the one vt_conv3d_gcr_f16x3 launches for this shape?  (the final-conv epilogue lives there only)
static bool conv_h_specialised(int tz) {
    static const bool spec = !(getenv("VTACO_CONV_SPEC") && getenv("VTACO_CONV_SPEC")[0] == '0');
    return spec && tz != 0 && tz != 2;
}
// the specialised-wave kernels address their sources through buffer descriptors (32-bit byte offsets): tensors of 4 GiB and more
// stay on the uniform-wave kernel's 64-bit addressing
static bool conv_h_fits_32bit(const Src &s, int B) {
    const size_t lim = (size_t)1 << 32;
    if ((size_t)B * s.D * s.H * s.W * s.C1 * 4 >= lim) return false;
    return !s.low || (size_t)B * (s.D / 2) * (s.H / 2) * (s.W / 2) * s.C2 * 4 < lim;
}

static bool conv_h_inline() {
    static const bool on = getenv("VTACO_CONV_SPEC") && getenv("VTACO_CONV_SPEC")[0] == '2';
    return on;
}

// the link between a flagged first layer and the layer behind it (HbArgs::cls_*): `out` on the first layer's launch (with the second
// layer's packed weights and GroupNorm parameters), `in` on the second's
struct ClsLink { const float *w = nullptr, *gamma = nullptr, *beta = nullptr; float *out = nullptr; const float *in = nullptr; };
static int conv_h_launch(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                         const float *scale_shift, const float *packed_w_f16x3, int Cout, int relu, float *out,
                         float *out_part, const float *in_absmax, const float *fin_w, const float *fin_b, void *stream,
                         const GnIn &stat_in = GnIn{}, const GnOut &stat_out = GnOut{}, const unsigned char *tile_skip = nullptr,
                         const ClsLink &cls = ClsLink{}, const float *stat_x = nullptr, float *fin_y = nullptr);
// would conv_h_launch take skip flags on this plain layer?
static bool conv_h_skip_accepts(int B, int D, int H, int W, int Cin, int Cout);
static int conv_h_wgs_of(int B, int D, int H, int W, int Cin, int Cout);

int vt_conv3d_gcr_f16x3_scaled(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                               const float *scale_shift, const float *packed_w_f16x3, int Cout, int relu, float *out,
                               float *out_part, const float *in_absmax, void *stream) {
    return conv_h_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_f16x3, Cout, relu, out, out_part, in_absmax, nullptr, nullptr, stream);
}

int vt_conv3d_xstats_blocks(int B, int D, int H, int W, int Cin, int Cout) {
    const int tz = conv_h_tz(B, D, H, W, Cin, Cout);
    const Src src{nullptr, nullptr, Cin, 0, D, H, W};
    if (!tz || !conv_h_specialised(tz) || conv_h_inline() || !conv_h_fits_32bit(src, B)) return 0;
    return conv_h_wgs_per_scene(B, D, H, W, Cout, tz);
}

int vt_conv3d_gcr_f16x3_xstats(const float *g, int C, int B, int D, int H, int W, const float *packed_w_f16x3, int Cout,
                               const float *in_absmax, const float *stat_x, float *out, float *out_part, void *stream) {
    if (!stat_x || !out_part) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_f16x3_xstats: null argument");
    return conv_h_launch(g, C, nullptr, 0, B, D, H, W, nullptr, packed_w_f16x3, Cout, 0, out, out_part, in_absmax, nullptr, nullptr, stream,
                         GnIn{}, GnOut{}, nullptr, ClsLink{}, stat_x);
}

int vt_conv3d_gcr_f16x3_skip(const float *x, int C, int B, int D, int H, int W, const float *scale_shift, const float *packed_w_f16x3,
                             int Cout, int relu, const unsigned char *tile_flags, float *out, float *out_part, void *stream) {
    if (!tile_flags) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_f16x3_skip: null flags");
    return conv_h_launch(x, C, nullptr, 0, B, D, H, W, scale_shift, packed_w_f16x3, Cout, relu, out, out_part, nullptr, nullptr, nullptr, stream,
                         GnIn{}, GnOut{}, tile_flags);
}


// does vt_conv3d_gcr_f16x3_final cover this layer?  (32 output channels, a shape of the specialised-wave kernel)
int vt_conv3d_final_fusable(int B, int D, int H, int W, int Cin, int Cout) {
    const Src src{nullptr, nullptr, Cin, 0, D, H, W};              // (the fused layer is a plain one: no `low`)
    return Cout == 32 && conv_h_specialised(conv_h_tz(B, D, H, W, Cin, Cout)) && conv_h_fits_32bit(src, B) ? 1 : 0;
}

// out[v][o] = fin_b[o] + sum_c fin_w[o][c] relu(conv(...))[v][c]: the last 'gcr' layer and the final 1x1x1 conv (32 -> 32) in one launch
int vt_conv3d_gcr_f16x3_final(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                              const float *scale_shift, const float *packed_w_f16x3, int Cout,
                              const float *final_packed_f16x3, const float *final_b, float *out, void *stream) {
    if (!final_packed_f16x3) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_f16x3_final: null argument");
    if (Cout != 32 || !conv_h_specialised(conv_h_tz(B, D, H, W, C1 + (low ? C2 : 0), Cout)))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_f16x3_final: needs Cout = 32 and a shape of the specialised-wave kernel; use vt_conv3d_gcr_f16x3 + vt_conv1x1_cl");
    return conv_h_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_f16x3, Cout, 1, out, nullptr, nullptr, final_packed_f16x3, final_b, stream);
}

// the same launch, and y = relu(conv(...)) itself to `y_keep` (the training forward: the layer's backward and the final conv's read it)
int vt_conv3d_gcr_f16x3_final_keep(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                                   const float *scale_shift, const float *packed_w_f16x3, int Cout,
                                   const float *final_packed_f16x3, const float *final_b, float *out, float *y_keep, void *stream) {
    if (!final_packed_f16x3 || !y_keep) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_f16x3_final_keep: null argument");
    if (Cout != 32 || !conv_h_specialised(conv_h_tz(B, D, H, W, C1 + (low ? C2 : 0), Cout)))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_f16x3_final_keep: needs Cout = 32 and a shape of the specialised-wave kernel");
    return conv_h_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_f16x3, Cout, 1, out, nullptr, nullptr, final_packed_f16x3, final_b, stream,
                         GnIn{}, GnOut{}, nullptr, ClsLink{}, nullptr, y_keep);
}

// w [32][32] (out, in) -> the A-operand fragments of dense32s<2>: [part: hi, lo][k-step 0, 1][64 lanes][8 halves], lane = (out row,
// k-group), element e of k-step s = input channel chan_of(8 s + e, k-group) -- the channel that register 8 s + e of the 3x3x3
// layer's accumulator holds in that half of the wave
__global__ void conv1x1_pack_h_kernel(const float *w, float *packed) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;            // fragment: part * 128 + s * 64 + lane
    if (f >= 256) return;
    const int l = f & 63, s = (f >> 6) & 1, part = f >> 7;
    const int o = l & 31, kg = l >> 5;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = w[o * 32 + chan_of(8 * s + e, kg)];
        const _Float16 h = (_Float16)x;
        v[e] = part ? (_Float16)(x - (float)h) : h;
    }
    reinterpret_cast<f16x8 *>(packed)[f] = v;
}

int vt_conv1x1_pack_f16x3(const float *w, int Cout, int Cin, float *packed, void *stream) {
    if (!w || !packed) return vt_fail(VT_ERR_INVALID, "vt_conv1x1_pack_f16x3: null argument");
    if (Cout != 32 || Cin != 32) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv1x1_pack_f16x3: the fused final conv is built for 32 -> 32 channels");
    hipLaunchKernelGGL(conv1x1_pack_h_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w, packed);
    return vt_check(hipGetLastError(), "vt_conv1x1_pack_f16x3");
}

static int conv_h_launch(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                         const float *scale_shift, const float *packed_w_f16x3, int Cout, int relu, float *out,
                         float *out_part, const float *in_absmax, const float *fin_w, const float *fin_b, void *stream,
                         const GnIn &stat_in, const GnOut &stat_out, const unsigned char *tile_skip, const ClsLink &cls, const float *stat_x,
                         float *fin_y) {
    HbArgs ha;
    ha.stat_x = stat_x;
    ha.c.stat_in = stat_in; ha.c.stat_out = stat_out;
    ha.in_absmax = in_absmax;
    ha.fin_w = fin_w; ha.fin_b = fin_b; ha.fin_y = fin_w ? fin_y : nullptr;
    ConvArgs &a = ha.c;
    a.s = Src{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(a.s, B) || !packed_w_f16x3 || !out) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_f16x3: bad argument");
    const int tz = conv_h_tz(B, D, H, W, a.s.C1 + a.s.C2, Cout);
    if (!tz) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_f16x3: shape not covered (see vt_conv3d_stat_blocks_f16x3); use vt_conv3d_gcr[_bf16x3]");
    a.scale_shift = scale_shift; a.wp = packed_w_f16x3; a.out = out; a.part = out_part; a.Cout = Cout; a.relu = relu;
    a.TX = a.TY = 8; a.TZ = tz;
    a.tiles_x = W / 8; a.tiles_y = H / 8; a.tiles_z = D / tz;
    ha.wgs_per_scene = conv_h_wgs_per_scene(B, D, H, W, Cout, tz);
    // skip flags: the specialised-wave kernel on a plain layer whose workgroups' tile lists fit
    if (tile_skip && !low && !in_absmax && !fin_w && conv_h_specialised(tz) && !conv_h_inline() &&
        (a.tiles_x * a.tiles_y * a.tiles_z + ha.wgs_per_scene - 1) / ha.wgs_per_scene <= HB_SKIP_LIST)
        ha.tile_skip = tile_skip;
    const dim3 grid((unsigned)(ha.wgs_per_scene * B), (unsigned)(Cout / 32));
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_h_kernel<8>), (int)hb_lds(8));
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_h_kernel<4>), (int)hb_lds(4));
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_h_kernel<2>), (int)hb_lds(2));
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_f16x3: hipFuncSetAttribute");
        attr = true;
    }
    const bool spec = conv_h_specialised(tz) && conv_h_fits_32bit(a.s, B);
    if (fin_w && !spec) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_f16x3_final: shape not on the specialised-wave kernel");
    if (!spec) ha.tile_skip = nullptr;
    if (cls.out || cls.in) {
        // (the caller asked conv_h_skip_accepts for both layers; a layer that reads class rows nobody wrote must not run)
        if (!ha.tile_skip || Cout != 32 || a.s.C1 + a.s.C2 > HB_MAX_CIN || (cls.in && (a.s.C1 != 32 || !stat_in.acc[0] || stat_in.groups != 8)) ||
            (cls.out && (!cls.w || !cls.gamma || !cls.beta)) || (cls.in && cls.out) || D < 16 || H < 16 || W < 16 || ha.wgs_per_scene < 32)
            return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_f16x3: class rows on a layer that does not take them");
        ha.cls_w = cls.w; ha.cls_gamma = cls.gamma; ha.cls_beta = cls.beta; ha.cls_out = cls.out; ha.cls_in = cls.in;
        if (cls.in) ha.skip_mask = 2u;
    }
    if (spec && conv_h_inline()) {               // the support work in the tap waves' MFMA gaps (VTACO_CONV_SPEC=2)
        bool attr_x = false;        // (vt_max_dyn_lds keeps the per-device record)
        if (!attr_x) {
            hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hx_kernel<8>), (int)hb_lds(8));
            if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hx_kernel<4>), (int)hb_lds(4));
            if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hx_kernel<8, true>), (int)hb_lds(8));
            if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hx_kernel<4, true>), (int)hb_lds(4));
            if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_f16x3: hipFuncSetAttribute");
            attr_x = true;
        }
        if (fin_w) {
            if (tz == 8) hipLaunchKernelGGL((conv3d_gcr_hx_kernel<8, true>), grid, dim3(64 * 8), hb_lds(8), (hipStream_t)stream, ha);
            else hipLaunchKernelGGL((conv3d_gcr_hx_kernel<4, true>), grid, dim3(64 * 4), hb_lds(4), (hipStream_t)stream, ha);
        } else {
            if (tz == 8) hipLaunchKernelGGL(conv3d_gcr_hx_kernel<8>, grid, dim3(64 * 8), hb_lds(8), (hipStream_t)stream, ha);
            else hipLaunchKernelGGL(conv3d_gcr_hx_kernel<4>, grid, dim3(64 * 4), hb_lds(4), (hipStream_t)stream, ha);
        }
        return vt_check(hipGetLastError(), "vt_conv3d_gcr_f16x3");
    }
    if (stat_x) {                                                  // data gradient with the GroupNorm backward's sums in the epilogue
        if (!spec || conv_h_inline() || fin_w || ha.tile_skip || !out_part || low)
            return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_f16x3_xstats: needs a plain layer on the specialised-wave kernel and a partial-sum buffer");
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hw_kernel<8, false, true>), (int)hb_lds_sparse(8));
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hw_kernel<4, false, true>), (int)hb_lds_sparse(4));
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_f16x3_xstats: hipFuncSetAttribute");
        if (tz == 8) hipLaunchKernelGGL((conv3d_gcr_hw_kernel<8, false, true>), grid, dim3(hb_threads(8)), hb_lds_sparse(8), (hipStream_t)stream, ha);
        else hipLaunchKernelGGL((conv3d_gcr_hw_kernel<4, false, true>), grid, dim3(hb_threads(4)), hb_lds_sparse(4), (hipStream_t)stream, ha);
        return vt_check(hipGetLastError(), "vt_conv3d_gcr_f16x3_xstats");
    }
    if (spec) {                                                    // specialised tap / loader waves (VTACO_CONV_SPEC=0: the uniform-wave kernel)
        bool attr_w = false;        // (vt_max_dyn_lds keeps the per-device record)
        if (!attr_w) {
            hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hw_kernel<8>), (int)hb_lds_sparse(8));
            if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hw_kernel<4>), (int)hb_lds_sparse(4));
            if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_f16x3: hipFuncSetAttribute");
            attr_w = true;
        }
        if (fin_w) {
            bool attr_f = false;        // (vt_max_dyn_lds keeps the per-device record)
            if (!attr_f) {
                hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hw_kernel<8, true>), (int)hb_lds(8));
                if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_hw_kernel<4, true>), (int)hb_lds(4));
                if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_f16x3_final: hipFuncSetAttribute");
                attr_f = true;
            }
            if (tz == 8) hipLaunchKernelGGL((conv3d_gcr_hw_kernel<8, true>), grid, dim3(hb_threads(8)), hb_lds(8), (hipStream_t)stream, ha);
            else hipLaunchKernelGGL((conv3d_gcr_hw_kernel<4, true>), grid, dim3(hb_threads(4)), hb_lds(4), (hipStream_t)stream, ha);
            return vt_check(hipGetLastError(), "vt_conv3d_gcr_f16x3_final");
        }
        if (tz == 8) hipLaunchKernelGGL(conv3d_gcr_hw_kernel<8>, grid, dim3(hb_threads(8)), hb_lds_sparse(8), (hipStream_t)stream, ha);
        else hipLaunchKernelGGL(conv3d_gcr_hw_kernel<4>, grid, dim3(hb_threads(4)), hb_lds_sparse(4), (hipStream_t)stream, ha);
        return vt_check(hipGetLastError(), "vt_conv3d_gcr_f16x3");
    }
    if (tz == 8) hipLaunchKernelGGL(conv3d_gcr_h_kernel<8>, grid, dim3(hb_threads(8)), hb_lds(8), (hipStream_t)stream, ha);
    else if (tz == 4) hipLaunchKernelGGL(conv3d_gcr_h_kernel<4>, grid, dim3(hb_threads(4)), hb_lds(4), (hipStream_t)stream, ha);
    else hipLaunchKernelGGL(conv3d_gcr_h_kernel<2>, grid, dim3(hb_threads(2)), hb_lds(2), (hipStream_t)stream, ha);
    return vt_check(hipGetLastError(), "vt_conv3d_gcr_f16x3");
}

static bool conv_h_skip_accepts(int B, int D, int H, int W, int Cin, int Cout) {
    const int tz = conv_h_tz(B, D, H, W, Cin, Cout);
    if (!tz || !conv_h_specialised(tz) || conv_h_inline()) return false;
    const Src src{nullptr, nullptr, Cin, 0, D, H, W};
    const int wgs = conv_h_wgs_per_scene(B, D, H, W, Cout, tz);
    return conv_h_fits_32bit(src, B) && ((W / 8) * (H / 8) * (D / tz) + wgs - 1) / wgs <= HB_SKIP_LIST;
}
static int conv_h_wgs_of(int B, int D, int H, int W, int Cin, int Cout) {
    const int tz = conv_h_tz(B, D, H, W, Cin, Cout);
    return tz ? conv_h_wgs_per_scene(B, D, H, W, Cout, tz) : 0;
}

// ---- decoder-entry layers in per-parity form (unet3d_up.inc) ---------------------------------------------------------------
// covered: a concat layer ([skip | upsample(low)]) of a shape the specialised-wave kernel takes, with both halves in multiples
// of 16 channels (an even number of 8-channel chunks per phase)
static int conv_up_tz(int C1, int C2, int B, int D, int H, int W, int Cout) {
    static const bool off = getenv("VTACO_CONV_UP") && getenv("VTACO_CONV_UP")[0] == '0';      // A/B knob: the 27-tap kernel on every layer
    if (off || C2 <= 0 || (C1 & 15) || (C2 & 15)) return 0;
    const int tz = conv_h_tz(B, D, H, W, C1 + C2, Cout);
    const Src src{nullptr, reinterpret_cast<const float *>(1), C1, C2, D, H, W};
    return (tz == 8 || tz == 4) && conv_h_specialised(tz) && !conv_h_inline() && conv_h_fits_32bit(src, B) ? tz : 0;
}

static int conv_up_launch(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                          const float *scale_shift, const float *packed_w_f16x3, const float *packed_up, int Cout, int relu, float *out,
                          float *out_part, void *stream, const GnIn &stat_in = GnIn{}, const GnOut &stat_out = GnOut{}) {
    HbArgs ha;
    ha.c.stat_in = stat_in; ha.c.stat_out = stat_out;
    ConvArgs &a = ha.c;
    a.s = Src{skip, low, C1, C2, D, H, W};
    if (!low || !src_ok(a.s, B) || !packed_w_f16x3 || !packed_up || !out) return vt_fail(VT_ERR_INVALID, "vt_conv3d_gcr_f16x3_up: bad argument");
    const int tz = conv_up_tz(C1, C2, B, D, H, W, Cout);
    if (!tz) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_gcr_f16x3_up: shape not covered (see vt_conv3d_up_covers); use vt_conv3d_gcr_f16x3");
    a.scale_shift = scale_shift; a.wp = packed_w_f16x3; a.out = out; a.part = out_part; a.Cout = Cout; a.relu = relu;
    ha.wp_up = packed_up;
    a.TX = a.TY = 8; a.TZ = tz;
    a.tiles_x = W / 8; a.tiles_y = H / 8; a.tiles_z = D / tz;
    ha.wgs_per_scene = conv_h_wgs_per_scene(B, D, H, W, Cout, tz);
    const dim3 grid((unsigned)(ha.wgs_per_scene * B), (unsigned)(Cout / 32));
    {
        hipError_t e = tz == 8 ? vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_up_kernel<8>), (int)up_lds(8))
                               : vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_gcr_up_kernel<4>), (int)up_lds(4));
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_gcr_f16x3_up: hipFuncSetAttribute");
    }
    if (tz == 8) hipLaunchKernelGGL(conv3d_gcr_up_kernel<8>, grid, dim3(hb_threads(8)), up_lds(8), (hipStream_t)stream, ha);
    else hipLaunchKernelGGL(conv3d_gcr_up_kernel<4>, grid, dim3(hb_threads(4)), up_lds(4), (hipStream_t)stream, ha);
    return vt_check(hipGetLastError(), "vt_conv3d_gcr_f16x3_up");
}

size_t vt_conv3d_up_packed_floats(int Cout, int C2) {
    if (Cout <= 0 || C2 <= 0 || (Cout & 31) || (C2 & 15)) return 0;
    return (size_t)(C2 / 8) * (Cout / 32) * (UP_CW / 4);
}

int vt_conv3d_pack_f16x3_up(const float *w, int Cout, int Cin, int C1, float *packed, void *stream) {
    if (!w || !packed) return vt_fail(VT_ERR_INVALID, "vt_conv3d_pack_f16x3_up: null argument");
    if (C1 <= 0 || C1 >= Cin || !vt_conv3d_up_packed_floats(Cout, Cin - C1))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_pack_f16x3_up: Cout must be a multiple of 32, the low channels a multiple of 16");
    const size_t frags = vt_conv3d_up_packed_floats(Cout, Cin - C1) / 4;
    size_t g = (frags + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv3d_pack_up_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, C1, packed, frags);
    return vt_check(hipGetLastError(), "vt_conv3d_pack_f16x3_up");
}

int vt_conv3d_up_covers(int C1, int C2, int B, int D, int H, int W, int Cout) {
    return conv_up_tz(C1, C2, B, D, H, W, Cout) != 0 ? 1 : 0;
}

int vt_conv3d_gcr_f16x3_up(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                           const float *scale_shift, const float *packed_w_f16x3, const float *packed_up, int Cout, int relu, float *out,
                           float *out_part, void *stream) {
    return conv_up_launch(skip, C1, low, C2, B, D, H, W, scale_shift, packed_w_f16x3, packed_up, Cout, relu, out, out_part, stream);
}

#ifdef VT_DIAG_HB
int vt_diag_hb_read(unsigned long long *host, size_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(vt_diag_hb_buf), n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

int vt_maxpool3d_cl(const float *x, int B, int D, int H, int W, int C, float *out, void *stream) {
    if (!x || !out || B <= 0 || C <= 0 || D < 2 || H < 2 || W < 2) return vt_fail(VT_ERR_INVALID, "vt_maxpool3d_cl: bad argument");
    const size_t total = (size_t)B * (D / 2) * (H / 2) * (W / 2) * C;
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(maxpool3d_cl_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, out, D, H, W, C, total);
    return vt_check(hipGetLastError(), "vt_maxpool3d_cl");
}

static int maxpool_stats_launch(const float *x, int B, int D, int H, int W, int C, float *out, int nblk, float *part, const GnOut &stat_out, void *stream) {
    if (!x || !out || (!part && !stat_out.acc) || B <= 0 || C <= 0 || (C & 31) || D < 2 || H < 2 || W < 2 || nblk <= 0)
        return vt_fail(VT_ERR_INVALID, "vt_maxpool3d_cl_stats: bad argument");
    hipLaunchKernelGGL(maxpool3d_cl_stats_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, out, D, H, W, C, nblk, part, stat_out);
    return vt_check(hipGetLastError(), "vt_maxpool3d_cl_stats");
}

int vt_maxpool3d_cl_stats(const float *x, int B, int D, int H, int W, int C, float *out, int nblk, float *part, void *stream) {
    if (!part) return vt_fail(VT_ERR_INVALID, "vt_maxpool3d_cl_stats: bad argument");
    return maxpool_stats_launch(x, B, D, H, W, C, out, nblk, part, GnOut{}, stream);
}

int vt_conv1x1_cl(const float *x, int64_t V, int Cin, const float *w, const float *bias, int Cout, float *out, void *stream) {
    if (!x || !w || !out || V <= 0 || Cin <= 0 || Cout <= 0) return vt_fail(VT_ERR_INVALID, "vt_conv1x1_cl: bad argument");
    if (!(Cin & 31) && !(Cout & 31)) {
        size_t g = ((size_t)V + 127) / 128;
        const size_t cap = (size_t)vt_num_cus() * 8;
        if (g > cap) g = cap;
        hipLaunchKernelGGL(conv1x1_mfma_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, Cin, Cout, (size_t)V);
        return vt_check(hipGetLastError(), "vt_conv1x1_cl");
    }
    const size_t lds = ((size_t)Cout * (Cin + 1) + Cout) * sizeof(float);
    if (lds > 64 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv1x1_cl: weights do not fit 64 KiB of LDS");
    size_t g = ((size_t)V * Cout + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(conv1x1_cl_kernel, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream, x, w, bias, out, Cin, Cout, (size_t)V);
    return vt_check(hipGetLastError(), "vt_conv1x1_cl");
}

// ---- the whole UNet3D.forward in one call (no host round trips between its ~45 launches) ---------
namespace {
struct Bump {
    char *base; size_t off;
    float *take(size_t floats) { float *p = base ? (float *)(base + off) : nullptr; off += (floats * sizeof(float) + 255) / 256 * 256; return p; }
};
struct Tensor { float *x; float *part; int nblk; int C; unsigned long long *acc = nullptr; unsigned *flag = nullptr; int rows = 1; };

// blocks of a statistics pass over V voxels (the Python mirror uses the same rule, ops.stat_blocks: the per-block float sums -- and with them
// the results -- are then the same whichever path computes them): 16-voxel blocks where that fills 1024 workgroups, 8-voxel blocks below
int stat_blocks(int64_t V) {
    const int64_t n = V / (V >= 16384 ? 16 : 8);
    return (int)(n < 1 ? 1 : (n > 1024 ? 1024 : n));
}

// which kernel family a layer runs on (0: the exact-f32 kernels, which leave partial rows only)
enum ConvKind { CONV_F32 = 0, CONV_HALF, CONV_KSPLIT, CONV_SPLIT };
ConvKind conv_kind(const vt_unet3d_conv &c, int B, int Ri) {
    if (c.packed_f16x3 && conv_h_tz(B, Ri, Ri, Ri, c.cin, c.cout) != 0) return CONV_HALF;
    if (!c.packed_bf16x3 && !c.packed_f16x3_thin) return CONV_F32;
    if (vt_conv3d_ksplit_workspace_bytes(B, Ri, Ri, Ri, c.cin, c.cout)) return CONV_KSPLIT;
    return conv_s_eligible(B, Ri, Ri, Ri, c.cin, c.cout) ? CONV_SPLIT : CONV_F32;
}
int conv_stat_blocks(ConvKind k, const vt_unet3d_conv &c, int B, int Ri) {
    return k == CONV_HALF ? vt_conv3d_stat_blocks_f16x3(B, Ri, Ri, Ri, c.cin, c.cout)
         : k == CONV_KSPLIT ? vt_conv3d_stat_blocks_ksplit(B, Ri, Ri, Ri, c.cin, c.cout)
         : k == CONV_SPLIT ? vt_conv3d_stat_blocks_bf16x3(B, Ri, Ri, Ri, c.cin, c.cout) : vt_conv3d_stat_blocks(B, Ri, Ri, Ri, c.cin, c.cout);
}

// GroupNorm statistics through accumulator rows (GnOut / GnIn) where every layer runs on a kernel family that carries them and
// the group / atom arithmetic divides; VTACO_GN_FOLD=0 keeps the finalising launches (A/B)
bool unet3d_fold_ok(int B, int R, const vt_unet3d_params *p) {
    const char *e = getenv("VTACO_GN_FOLD");
    if (e && e[0] == '0') return false;
    const int L = p->n_levels;
    auto ok = [&](const vt_unet3d_conv &c, int Ri, int c_skip) {
        const int groups = (c.cin >= p->groups) ? p->groups : 1;
        return conv_kind(c, B, Ri) != CONV_F32 && groups <= 64 && c.cin % groups == 0 && (c.cin / groups) % GN_ATOM == 0 &&
               c_skip % GN_ATOM == 0 && c.cout % 32 == 0 && c.cin <= GN_MAX_CIN;
    };
    for (int i = 0; i < L; ++i) for (int k = 0; k < 2; ++k) { if (!ok(p->enc[i][k], R >> i, 0)) return false; }
    for (int k = 0; k + 1 < L; ++k) {
        const int lvl = L - 2 - k;
        if (!ok(p->dec[k][0], R >> lvl, p->enc[lvl][1].cout) || !ok(p->dec[k][1], R >> lvl, 0)) return false;
    }
    return true;
}

// does the SECOND layer of level 0 take block flags too (the class rows of HbArgs::cls_*)?  By the shape alone.
bool unet3d_skip2_shape(int B, int R, const vt_unet3d_params *p) {
    const char *e2 = getenv("VTACO_CONV_SKIP2");                     // (A/B and tests: read per call)
    const bool off2 = e2 && e2[0] == '0';
    const vt_unet3d_conv &c0 = p->enc[0][0], &c1 = p->enc[0][1];
    return !off2 && R >= 16 && c0.cout == 32 && c1.cin == 32 && c1.cout == 32 && p->groups == 8 && p->n_levels > 1 &&
           conv_kind(c0, B, R) == CONV_HALF && conv_kind(c1, B, R) == CONV_HALF &&
           conv_h_skip_accepts(B, R, R, R, c0.cin, c0.cout) && conv_h_skip_accepts(B, R, R, R, c1.cin, c1.cout) &&
           conv_h_wgs_of(B, R, R, R, c0.cin, c0.cout) >= 32;
}

struct StatRegion { size_t off = 0, words = 0; };                  // the accumulators' place in the workspace (known after a planning pass)

// vt_unet3d_fwd with the statistics in accumulator rows: 13 launches fewer (the first finalisation stays: the statistics of x
// come as partial rows from its producer, and that launch also clears the accumulators)
int unet3d_run_fold(const float *x_cl, int B, int R, const vt_unet3d_params *p, char *wsbase, size_t *ws_need, float *out,
                    hipStream_t st, const float *in_part, int in_nblk, StatRegion *region, const unsigned char *tile_flags = nullptr) {
    const int L = p->n_levels;
    const bool plan = wsbase == nullptr;
    StatRegion reg;
    if (!plan) {
        const int rc = unet3d_run_fold(x_cl, B, R, p, nullptr, nullptr, nullptr, st, in_part, in_nblk, &reg, nullptr);
        if (rc) return rc;
    }
    Bump ws{wsbase, 0};
    unsigned long long *zbase = plan ? nullptr : reinterpret_cast<unsigned long long *>(wsbase + reg.off);
    size_t zoff = 0;
    // rows and flags of tensor t, written by `nblk` workgroups per scene and channel block
    auto cells = [&](Tensor &t, int nblk) -> GnOut {
        GnOut o;
        o.C = t.C;
        o.rows = 1;
        while (o.rows < 64 && o.rows * 8 < nblk) o.rows <<= 1;
        o.acc = zbase ? zbase + zoff : nullptr; zoff += gn_acc_words(B, o.rows, t.C);
        o.flag = zbase ? reinterpret_cast<unsigned *>(zbase + zoff) : nullptr; zoff += (size_t)(B + 1) / 2;
        t.part = nullptr; t.nblk = 0;
        t.acc = o.acc; t.flag = o.flag; t.rows = o.rows;
        return o;
    };
    auto stat_in = [&](const vt_unet3d_conv &c, const Tensor &a, const Tensor *low, int Ri) {
        GnIn in;
        in.acc[0] = a.acc; in.flag[0] = a.flag; in.C[0] = a.C; in.rows[0] = a.rows;
        if (low) { in.acc[1] = low->acc; in.flag[1] = low->flag; in.C[1] = low->C; in.rows[1] = low->rows; }
        in.gamma = c.gn_w; in.beta = c.gn_b;
        in.groups = (c.cin >= p->groups) ? p->groups : 1;
        in.eps = (float)p->eps;
        in.count = (double)Ri * Ri * Ri;
        return in;
    };
    int rc = 0;
    // one 'gcr' layer; ss: a finished scale / shift table (the first layer's), else the statistics come through GnIn
    auto gcr = [&](const vt_unet3d_conv &c, const float *ss, const Tensor &a, const Tensor *low, int Ri, Tensor &o, bool want_stats,
                   const unsigned char *skipf = nullptr, const ClsLink &cls = ClsLink{}) -> int {
        const int C2 = low ? low->C : 0;
        if (a.C + C2 != c.cin) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd: channel mismatch between levels");
        o.C = c.cout;
        o.x = ws.take((size_t)B * Ri * Ri * Ri * c.cout);
        const ConvKind kind = conv_kind(c, B, Ri);
        const size_t ksbytes = kind == CONV_KSPLIT ? vt_conv3d_ksplit_workspace_bytes(B, Ri, Ri, Ri, c.cin, c.cout) : 0;
        float *kws = ksbytes ? ws.take(ksbytes / sizeof(float)) : nullptr;
        GnOut so;
        if (want_stats) so = cells(o, conv_stat_blocks(kind, c, B, Ri));
        if (plan) return 0;
        const GnIn si = ss ? GnIn{} : stat_in(c, a, low, Ri);
        const float *lx = low ? low->x : nullptr;
        if (kind == CONV_HALF && low && c.packed_f16x3_up && conv_up_tz(a.C, C2, B, Ri, Ri, Ri, c.cout))     // decoder entry: per-parity form
            return conv_up_launch(a.x, a.C, lx, C2, B, Ri, Ri, Ri, ss, c.packed_f16x3, c.packed_f16x3_up, c.cout, 1, o.x, nullptr, st, si, so);
        if (kind == CONV_HALF)
            return conv_h_launch(a.x, a.C, lx, C2, B, Ri, Ri, Ri, ss, c.packed_f16x3, c.cout, 1, o.x, nullptr, nullptr, nullptr, nullptr, st, si, so, skipf, cls);
        // the thin levels: IEEE-half pairs where the caller packed them (a network whose large levels run the split-f16 kernels)
        const bool thin_half = c.packed_f16x3_thin != nullptr;
        const float *thin_w = thin_half ? c.packed_f16x3_thin : c.packed_bf16x3;
        if (kind == CONV_KSPLIT)
            return conv_sk_launch(a.x, a.C, lx, C2, B, Ri, Ri, Ri, ss, thin_w, c.cout, 1, o.x, nullptr, si, so, kws, ksbytes, st, thin_half);
        return conv_s_launch(a.x, a.C, lx, C2, B, Ri, Ri, Ri, ss, thin_w, c.cout, 1, o.x, nullptr, si, so, st, thin_half);
    };
    Tensor skips[VT_UNET_MAX_LEVELS];
    Tensor cur;
    cur.x = const_cast<float *>(x_cl); cur.C = p->enc[0][0].cin;
    float *ss0 = nullptr;
    {
        const vt_unet3d_conv &c0 = p->enc[0][0];
        const int64_t V = (int64_t)R * R * R;
        if (in_part) {                                              // the producer of x left partial rows: finalise them and clear the cells
            ss0 = ws.take((size_t)B * c0.cin * 2);
            if (!plan && (rc = gn_scale_shift_launch(in_part, in_nblk, cur.C, nullptr, 0, 0, B, V, (c0.cin >= p->groups) ? p->groups : 1,
                                                     c0.gn_w, c0.gn_b, p->eps, ss0, zbase, reg.words, st))) return rc;
        } else {
            if (!plan && (rc = vt_check(hipMemsetAsync(zbase, 0, reg.words * sizeof(unsigned long long), st), "vt_unet3d_fwd: hipMemsetAsync"))) return rc;
            const int nblk = stat_blocks(V);
            const GnOut so = cells(cur, nblk);
            if (!plan && (rc = channel_stats_launch(x_cl, B, V, cur.C, nblk, nullptr, so, st))) return rc;
        }
    }
    for (int i = 0; i < L; ++i) {
        const int Ri = R >> i;
        if (i > 0) {
            Tensor pooled;
            pooled.C = cur.C;
            pooled.x = ws.take((size_t)B * Ri * Ri * Ri * cur.C);
            const int64_t Vp = (int64_t)Ri * Ri * Ri;
            // pool and statistics in ONE pass on every level: with partial rows that paid only where the blocks fill the chip (unet3d_run);
            // accumulator rows have no finalisation that reads per-block rows, and 8-voxel blocks (stat_blocks) give the small levels the
            // workgroups (16^3 and 8^3: 5.7 + 5.3 and 4.8 + 5.6 us as two launches each, 5.2 and 6.9 us as one)
            const int nblk = stat_blocks(Vp);
            const GnOut so = cells(pooled, nblk);
            if (!plan && (rc = maxpool_stats_launch(cur.x, B, 2 * Ri, 2 * Ri, 2 * Ri, cur.C, pooled.x, nblk, nullptr, so, st))) return rc;
            cur = pooled;
        }
        Tensor t1, t2;
        // (the first layer reads x: where the caller flagged x as zero over a block's halo, the block's taps are skipped; where the
        // 12^3 halo is empty too the second layer's are as well: its output there follows from class rows the first launch leaves)
        ClsLink mk, use;
        const unsigned char *skip2 = nullptr;
        if (i == 0) {
            const vt_unet3d_conv &c1 = p->enc[0][1];
            const bool shape2 = unet3d_skip2_shape(B, R, p);
            if (shape2) {                                           // (sized by the shape alone: the planning pass has no flags)
                float *rows = ws.take((size_t)B * HB_CLS * 32 * HB_CLS_ROW);
                if (tile_flags) {
                    mk.w = c1.packed_f16x3; mk.gamma = c1.gn_w; mk.beta = c1.gn_b; mk.out = rows;
                    use.in = rows;
                    skip2 = tile_flags;
                }
            }
        }
        if ((rc = gcr(p->enc[i][0], i == 0 ? ss0 : nullptr, cur, nullptr, Ri, t1, true, i == 0 ? tile_flags : nullptr, mk))) return rc;
        if ((rc = gcr(p->enc[i][1], nullptr, t1, nullptr, Ri, t2, L > 1, skip2, use))) return rc;
        skips[i] = t2;
        cur = t2;
    }
    for (int k = 0; k + 1 < L; ++k) {
        const int lvl = L - 2 - k, Ri = R >> lvl;
        Tensor t1, t2;
        if ((rc = gcr(p->dec[k][0], nullptr, skips[lvl], &cur, Ri, t1, true))) return rc;
        const vt_unet3d_conv &last = p->dec[k][1];
        if (k + 2 == L && p->final_packed_f16x3 && last.packed_f16x3 && p->out_channels == 32 &&
            vt_conv3d_final_fusable(B, Ri, Ri, Ri, last.cin, last.cout)) {
            if (region) { region->off = ws.off; region->words = zoff; }
            ws.take(zoff * 2);
            if (ws_need) *ws_need = ws.off;
            if (plan) return 0;
            return conv_h_launch(t1.x, t1.C, nullptr, 0, B, Ri, Ri, Ri, nullptr, last.packed_f16x3, last.cout, 1, out, nullptr, nullptr,
                                 p->final_packed_f16x3, p->final_b, st, stat_in(last, t1, nullptr, Ri), GnOut{});
        }
        if ((rc = gcr(last, nullptr, t1, nullptr, Ri, t2, k + 2 < L))) return rc;
        cur = t2;
    }
    if (region) { region->off = ws.off; region->words = zoff; }
    ws.take(zoff * 2);
    if (ws_need) *ws_need = ws.off;
    if (plan) return 0;
    return vt_conv1x1_cl(cur.x, (int64_t)B * R * R * R, cur.C, p->final_w, p->final_b, p->out_channels, out, st);
}

// plan == true only sizes the workspace
int unet3d_run(const float *x_cl, int B, int R, const vt_unet3d_params *p, char *wsbase, size_t *ws_need, float *out,
               hipStream_t st, const float *in_part = nullptr, int in_nblk = 0, const unsigned char *tile_flags = nullptr) {
    const int L = p->n_levels;
    if (L < 1 || L > VT_UNET_MAX_LEVELS) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd: bad level count");
    if (R % (1 << (L - 1))) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd: resolution must be divisible by 2^(levels-1)");
    if (unet3d_fold_ok(B, R, p)) return unet3d_run_fold(x_cl, B, R, p, wsbase, ws_need, out, st, in_part, in_nblk, nullptr, tile_flags);
    const bool plan = wsbase == nullptr;
    Bump ws{wsbase, 0};
    int maxC = 0;
    for (int i = 0; i < L; ++i) for (int k = 0; k < 2; ++k) { if (p->enc[i][k].cin > maxC) maxC = p->enc[i][k].cin; }
    for (int i = 0; i + 1 < L; ++i) for (int k = 0; k < 2; ++k) { if (p->dec[i][k].cin > maxC) maxC = p->dec[i][k].cin; }
    float *ss = ws.take((size_t)B * maxC * 2);
    auto stats_of = [&](const float *x, int Ri, int C, Tensor &t) -> int {
        const int64_t V = (int64_t)Ri * Ri * Ri;
        t.nblk = stat_blocks(V);
        t.part = ws.take((size_t)B * t.nblk * C * 2);
        t.C = C;
        return plan ? 0 : vt_channel_stats(x, B, V, C, t.nblk, t.part, st);
    };
    auto gcr = [&](const vt_unet3d_conv &c, const Tensor &a, const Tensor *low, int Ri, Tensor &o, const unsigned char *skipf = nullptr) -> int {
        const int C2 = low ? low->C : 0;
        if (a.C + C2 != c.cin) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd: channel mismatch between levels");
        o.C = c.cout;
        o.x = ws.take((size_t)B * Ri * Ri * Ri * c.cout);
        const bool half = c.packed_f16x3 && conv_h_tz(B, Ri, Ri, Ri, c.cin, c.cout) != 0;
        const bool thin_half = c.packed_f16x3_thin != nullptr;
        const float *thin_w = thin_half ? c.packed_f16x3_thin : c.packed_bf16x3;
        const size_t ksbytes = (!half && thin_w) ? vt_conv3d_ksplit_workspace_bytes(B, Ri, Ri, Ri, c.cin, c.cout) : 0;
        const bool split = !half && !ksbytes && thin_w && conv_s_eligible(B, Ri, Ri, Ri, c.cin, c.cout);
        o.nblk = half ? vt_conv3d_stat_blocks_f16x3(B, Ri, Ri, Ri, c.cin, c.cout)
                      : ksbytes ? vt_conv3d_stat_blocks_ksplit(B, Ri, Ri, Ri, c.cin, c.cout)
                      : split ? vt_conv3d_stat_blocks_bf16x3(B, Ri, Ri, Ri, c.cin, c.cout) : vt_conv3d_stat_blocks(B, Ri, Ri, Ri, c.cin, c.cout);
        o.part = ws.take((size_t)B * o.nblk * c.cout * 2);
        float *kws = ksbytes ? ws.take(ksbytes / sizeof(float)) : nullptr;
        if (plan) return 0;
        const int groups = (c.cin >= p->groups) ? p->groups : 1;
        int rc = vt_gn_scale_shift(a.part, a.nblk, a.C, low ? low->part : nullptr, low ? low->nblk : 0, C2, B,
                                   (int64_t)Ri * Ri * Ri, groups, c.gn_w, c.gn_b, p->eps, ss, st);
        if (rc) return rc;
        if (half && low && c.packed_f16x3_up && conv_up_tz(a.C, C2, B, Ri, Ri, Ri, c.cout))
            return conv_up_launch(a.x, a.C, low->x, C2, B, Ri, Ri, Ri, ss, c.packed_f16x3, c.packed_f16x3_up, c.cout, 1, o.x, o.part, st);
        if (half)
            return conv_h_launch(a.x, a.C, low ? low->x : nullptr, C2, B, Ri, Ri, Ri, ss, c.packed_f16x3, c.cout, 1, o.x, o.part, nullptr, nullptr, nullptr, st,
                                 GnIn{}, GnOut{}, skipf);
        if (ksbytes)
            return conv_sk_launch(a.x, a.C, low ? low->x : nullptr, C2, B, Ri, Ri, Ri, ss, thin_w, c.cout, 1, o.x, o.part, GnIn{}, GnOut{},
                                  kws, ksbytes, st, thin_half);
        if (split)
            return conv_s_launch(a.x, a.C, low ? low->x : nullptr, C2, B, Ri, Ri, Ri, ss, thin_w, c.cout, 1, o.x, o.part, GnIn{}, GnOut{}, st, thin_half);
        return vt_conv3d_gcr(a.x, a.C, low ? low->x : nullptr, C2, B, Ri, Ri, Ri, ss, c.packed, c.cout, 1, o.x, o.part, st);
    };
    Tensor skips[VT_UNET_MAX_LEVELS];
    Tensor cur;
    cur.x = const_cast<float *>(x_cl); cur.C = p->enc[0][0].cin;
    int rc = 0;
    if (in_part) { cur.part = const_cast<float *>(in_part); cur.nblk = in_nblk; }      // the producer of x left its partial sums
    else rc = stats_of(x_cl, R, cur.C, cur);
    if (rc) return rc;
    for (int i = 0; i < L; ++i) {
        const int Ri = R >> i;
        if (i > 0) {
            Tensor pooled;
            pooled.C = cur.C;
            pooled.x = ws.take((size_t)B * Ri * Ri * Ri * cur.C);
            const int64_t Vp = (int64_t)Ri * Ri * Ri;
            if (Vp / 16 >= 1024) {
                // pool and statistics in one pass where the statistics have the blocks to fill the chip (64^3 -> 32^3: 17.6 -> 10.3 us);
                // with fewer blocks the fused kernel's eight loads per value sit on too few workgroups (measured slower)
                pooled.nblk = 1024;                                                         // as stats_of
                pooled.part = ws.take((size_t)B * pooled.nblk * pooled.C * 2);
                pooled.C = cur.C;
                if (!plan && (rc = vt_maxpool3d_cl_stats(cur.x, B, 2 * Ri, 2 * Ri, 2 * Ri, cur.C, pooled.x, pooled.nblk, pooled.part, st))) return rc;
            } else {
                if (!plan && (rc = vt_maxpool3d_cl(cur.x, B, 2 * Ri, 2 * Ri, 2 * Ri, cur.C, pooled.x, st))) return rc;
                if ((rc = stats_of(pooled.x, Ri, pooled.C, pooled))) return rc;
            }
            cur = pooled;
        }
        Tensor t1, t2;
        if ((rc = gcr(p->enc[i][0], cur, nullptr, Ri, t1, i == 0 ? tile_flags : nullptr))) return rc;
        if ((rc = gcr(p->enc[i][1], t1, nullptr, Ri, t2))) return rc;
        skips[i] = t2;
        cur = t2;
    }
    for (int k = 0; k + 1 < L; ++k) {
        const int lvl = L - 2 - k, Ri = R >> lvl;
        Tensor t1, t2;
        if ((rc = gcr(p->dec[k][0], skips[lvl], &cur, Ri, t1))) return rc;
        const vt_unet3d_conv &last = p->dec[k][1];
        if (k + 2 == L && p->final_packed_f16x3 && last.packed_f16x3 && p->out_channels == 32 &&
            vt_conv3d_final_fusable(B, Ri, Ri, Ri, last.cin, last.cout)) {
            // the last 'gcr' layer with the final 1x1x1 conv in its epilogue: one launch and a 67 MB round trip less
            if (ws_need) *ws_need = ws.off;
            if (plan) return 0;
            const int groups = (last.cin >= p->groups) ? p->groups : 1;
            if ((rc = vt_gn_scale_shift(t1.part, t1.nblk, t1.C, nullptr, 0, 0, B, (int64_t)Ri * Ri * Ri, groups, last.gn_w, last.gn_b, p->eps, ss, st))) return rc;
            return vt_conv3d_gcr_f16x3_final(t1.x, t1.C, nullptr, 0, B, Ri, Ri, Ri, ss, last.packed_f16x3, last.cout, p->final_packed_f16x3, p->final_b, out, st);
        }
        if ((rc = gcr(last, t1, nullptr, Ri, t2))) return rc;
        cur = t2;
    }
    if (ws_need) *ws_need = ws.off;
    if (plan) return 0;
    return vt_conv1x1_cl(cur.x, (int64_t)B * R * R * R, cur.C, p->final_w, p->final_b, p->out_channels, out, st);
}
}  // namespace

size_t vt_unet3d_workspace_bytes(int B, int R, const vt_unet3d_params *params_host) {
    size_t need = 0;
    if (!params_host || B <= 0 || R <= 0) return 0;
    if (unet3d_run(nullptr, B, R, params_host, nullptr, &need, nullptr, nullptr)) return 0;
    return need;
}

int vt_unet3d_fwd_stats(const float *x_cl, const float *in_part, int in_nblk, int B, int R, const vt_unet3d_params *params_host,
                        void *workspace, size_t workspace_bytes, float *out, void *stream) {
    if (!x_cl || !in_part || in_nblk <= 0 || !params_host || !workspace || !out) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd_stats: bad argument");
    const size_t need = vt_unet3d_workspace_bytes(B, R, params_host);
    if (!need) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd_stats: bad configuration");
    if (workspace_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_unet3d_fwd_stats: workspace too small");
    return unet3d_run(x_cl, B, R, params_host, (char *)workspace, nullptr, out, (hipStream_t)stream, in_part, in_nblk);
}

int vt_unet3d_skip_layers(int B, int R, const vt_unet3d_params *params_host) {
    if (!params_host || B <= 0 || R <= 0 || params_host->n_levels < 1) return 0;
    const vt_unet3d_conv &c0 = params_host->enc[0][0];
    if (!c0.packed_f16x3 || conv_kind(c0, B, R) != CONV_HALF || !conv_h_skip_accepts(B, R, R, R, c0.cin, c0.cout)) return 0;
    return unet3d_fold_ok(B, R, params_host) && unet3d_skip2_shape(B, R, params_host) ? 2 : 1;
}

int vt_unet3d_fwd_skip(const float *x_cl, const float *in_part, int in_nblk, const unsigned char *tile_flags, int B, int R,
                       const vt_unet3d_params *params_host, void *workspace, size_t workspace_bytes, float *out, void *stream) {
    if (!x_cl || !tile_flags || !params_host || !workspace || !out || (in_part && in_nblk <= 0))
        return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd_skip: bad argument");
    const size_t need = vt_unet3d_workspace_bytes(B, R, params_host);
    if (!need) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd_skip: bad configuration");
    if (workspace_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_unet3d_fwd_skip: workspace too small");
    return unet3d_run(x_cl, B, R, params_host, (char *)workspace, nullptr, out, (hipStream_t)stream, in_part, in_nblk, tile_flags);
}

int vt_unet3d_fwd(const float *x_cl, int B, int R, const vt_unet3d_params *params_host,
                  void *workspace, size_t workspace_bytes, float *out, void *stream) {
    if (!x_cl || !params_host || !workspace || !out) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd: null argument");
    const size_t need = vt_unet3d_workspace_bytes(B, R, params_host);
    if (!need) return vt_fail(VT_ERR_INVALID, "vt_unet3d_fwd: bad configuration");
    if (workspace_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_unet3d_fwd: workspace too small");
    return unet3d_run(x_cl, B, R, params_host, (char *)workspace, nullptr, out, (hipStream_t)stream);
}

}  // extern "C"

// =====================================================================================
// Backward of the 'gcr' block, max-pool and concat/upsample (training; reference: PyTorch
// autograd of src/encoder/unet3d.py:20-72, 219-238, 283-293 triggered by loss.backward(),
// training.py:79,89,96).  With y = relu(conv(xn)), xn = GroupNorm(x_cat):
//   g    = dy * (y > 0)                                   relu_mask_kernel
//   dxn  = conv3x3x3(g, W^T flipped)                       the FORWARD conv kernels on repacked weights
//   dW   = sum_v g[v] (x) xn[v + tap]                      conv3d_wgrad_kernel (f32 MFMA, K = voxels)
//   dx   = A_c dxn + B_g x + C_g,  dgamma, dbeta           gn_bwd_stats / gn_bwd_coeffs / gn_bwd_apply
// =====================================================================================
namespace {

// absmax (or null): max |g| as the bit pattern of a non-negative float, which orders like an unsigned integer: one atomicMax per
// workgroup on a cell the launcher zeroed -- exact and independent of the order of arrival (NaN gradients land above every
// finite value and stop the consumers' rescale, as a NaN maximum would)
__global__ void __launch_bounds__(256) relu_mask_kernel(const f32x4 *dy, const f32x4 *y, f32x4 *g, size_t n4, unsigned *absmax) {
    unsigned u = 0;
    auto bits = [](float x) { return __builtin_bit_cast(unsigned, x) & 0x7fffffffu; };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 a = dy[i], b = y[i];
        const f32x4 v = f32x4{b.x > 0.f ? a.x : 0.f, b.y > 0.f ? a.y : 0.f, b.z > 0.f ? a.z : 0.f, b.w > 0.f ? a.w : 0.f};
        g[i] = v;
        const unsigned m01 = max(bits(v.x), bits(v.y)), m23 = max(bits(v.z), bits(v.w));
        u = max(u, max(m01, m23));
    }
    if (!absmax) return;
    __shared__ unsigned wmax[4];
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)u, o); u = t > u ? t : u; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a0 = wmax[0] > wmax[1] ? wmax[0] : wmax[1], a1 = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
        atomicMax(absmax, a0 > a1 ? a0 : a1);
    }
}

// ---- backward of the final 1x1x1 conv (32 -> 32) behind the last 'gcr' layer ------------------------------------------------
// out[v][co] = sum_ci W[co][ci] y[v][ci] + bias[co] with y the last layer's ReLU output, so in ONE pass over dout and y:
//     g[v][ci]   = (y[v][ci] > 0) * sum_co dout[v][co] W[co][ci]     the last layer's masked output gradient, and max |g|
//     dW[co][ci] = sum_v dout[v][co] y[v][ci],   db[co] = sum_v dout[v][co]
// (the framework ran a GEMM for dy, vt_relu_mask_absmax, a batched GEMM + sum for dW and a reduction for db: 2.7 GB of traffic for
// eight 64^3 scenes, 0.8 GB here).  Exact f32 on v_mfma_f32_32x32x2_f32 (2 M voxels: 31 us of matrix time under 0.17 ms of memory):
// a wave takes 32 voxels per turn -- 16 MFMAs D[ci][v] += W^T[ci][co pair] dout^T[co pair][v] for g (accumulator layout: masked and
// stored with 16-byte accesses), 16 MFMAs D[co][ci] += dout^T[co][v pair] y[v pair][ci] into an accumulator it keeps for the whole
// launch.  The eight waves' dW / db meet in LDS in wave order, the workgroups' in conv1x1_bwd_reduce_kernel in workgroup order.
constexpr int F1_WAVES = 8, F1_WGS = 512, F1_PART = 32 * 32 + 32;
__global__ void __launch_bounds__(F1_WAVES * 64, 4)        // (second argument: waves per SIMD -> 128 registers, two workgroups per CU)
conv1x1_bwd_kernel(const float *dout, const float *y, const float *w, unsigned n, float *g, unsigned *absmax, float *partial) {
    __shared__ float red[F1_WAVES][F1_PART];
    __shared__ float wl[16][64];
    __shared__ unsigned wmax[F1_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kk = lane >> 5;
    for (int e = threadIdx.x; e < 1024; e += F1_WAVES * 64) {
        const int s = e >> 6, l = e & 63;
        wl[s][l] = w[(2 * s + (l >> 5)) * 32 + (l & 31)];                           // A[ci = l & 31][co = 2 s + (l >> 5)]
    }
    __syncthreads();
    f32x16 accw;
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = 0.0f;
    float dbs = 0.0f;
    unsigned u = 0;
    auto bits = [](float x) { return __builtin_bit_cast(unsigned, x) & 0x7fffffffu; };
    const unsigned ntile = (n + 31) / 32;
    for (unsigned t = blockIdx.x * F1_WAVES + wave; t < ntile; t += gridDim.x * F1_WAVES) {
        const unsigned v0 = t * 32, v = min(v0 + (unsigned)j, n - 1);
        const bool live = v0 + j < n;
        const f32x4 *row = reinterpret_cast<const f32x4 *>(dout + (size_t)v * 32);
        float x[16];                                                                // x[s] = dout[v][2 s + kk]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 q = row[i];
            x[2 * i] = kk ? q.y : q.x;
            x[2 * i + 1] = kk ? q.w : q.z;
        }
        const f32x16 yv = load_acc16(y + (size_t)v * 32, kk);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = mfma(wl[s][lane], x[s], acc);
        __builtin_amdgcn_sched_barrier(0);
        // the v-pair operands of dW: element 64 s + lane of the tile's 32 rows = row 2 s + kk, column j (4-byte loads of the lines the
        // 16-byte loads above brought in), requested under the MFMAs above
        const float *pa = dout + (size_t)v0 * 32 + lane, *pb = y + (size_t)v0 * 32 + lane;
        float a2[16], b2[16];
        if (v0 + 32 <= n) {
#pragma unroll
            for (int s = 0; s < 16; ++s) { a2[s] = pa[s * 64]; b2[s] = pb[s * 64]; }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bool ok = v0 + 2 * s + kk < n;
                a2[s] = ok ? pa[s * 64] : 0.0f;
                b2[s] = ok ? pb[s * 64] : 0.0f;
            }
        }
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[r] = yv[r] > 0.0f ? acc[r] : 0.0f;
            if (live) u = max(u, bits(o[r]));
        }
        if (live) store_acc16(g + (size_t)v * 32, o, kk);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            accw = mfma(a2[s], b2[s], accw);
            dbs += a2[s];
        }
    }
    // lane (ci = j, half kk), register r = dW[co = chan_of(r, kk)][ci]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][chan_of(r, kk) * 32 + j] = accw[r];
    dbs += __shfl_xor(dbs, 32);
    if (lane < 32) red[wave][1024 + j] = dbs;
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)u, o); u = t > u ? t : u; }
    if (lane == 0) wmax[wave] = u;
    __syncthreads();
    for (int e = threadIdx.x; e < F1_PART; e += F1_WAVES * 64) {
        float sum = red[0][e];
#pragma unroll
        for (int k = 1; k < F1_WAVES; ++k) sum += red[k][e];
        partial[(size_t)blockIdx.x * F1_PART + e] = sum;
    }
    if (threadIdx.x == 0 && absmax) {
        unsigned m = wmax[0];
        for (int k = 1; k < F1_WAVES; ++k) m = wmax[k] > m ? wmax[k] : m;
        atomicMax(absmax, m);
    }
}

// dW [32][32] and db [32] from the workgroups' partials: eight thread groups add contiguous eighths in workgroup order (eight loads in
// flight), the eighths meet in LDS in group order
__global__ void __launch_bounds__(256)
conv1x1_bwd_reduce_kernel(const float *partial, int nparts, float *dw, float *db) {
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5, e = blockIdx.x * 32 + o;
    const int p0 = nparts * grp / 8, p1 = nparts * (grp + 1) / 8;
    float sum = 0.0f;
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = partial[(size_t)(p + k) * F1_PART + e];
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += t[k];
    }
    for (; p < p1; ++p) sum += partial[(size_t)p * F1_PART + e];
    part[grp][o] = sum;
    __syncthreads();
    if (grp == 0) {
        float tot = part[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += part[k][o];
        if (e < 1024) { if (dw) dw[e] = tot; }
        else if (db) db[e - 1024] = tot;
    }
}

// ---- weight gradient ---------------------------------------------------------------------------
// One workgroup = one (cout block, cin block) pair and a chunk of 8x8x4 voxel tiles; per tile the
// normalised input (+halo) and the masked output gradient are staged in LDS and every tap is an
// MFMA product D[co][ci] += g^T[co][v] xn[v + tap][ci] over the tile's 256 voxels (K).  Seven waves
// split the 27 taps (4 each); accumulators live in registers across the chunk's tiles; the per-chunk
// partials are summed in chunk order by conv3d_wgrad_reduce_kernel (bit-reproducible).
constexpr int WG_WAVES = 7, WG_TAPS = 4;
constexpr int WG_TX = 8, WG_TY = 8, WG_TZ = 4, WG_VOX = WG_TX * WG_TY * WG_TZ;
constexpr int WG_PX = WG_TX + 2, WG_PY = WG_TY + 2, WG_PZ = WG_TZ + 2, WG_HALO = WG_PX * WG_PY * WG_PZ;

struct WgradArgs {
    ConvArgs c;              // s, scale_shift (forward's), tiles_*; TX.. unused
    const float *g;          // [B,D,H,W,Cout] masked output gradient
    float *partial;          // [chunks][nco][ncib][27][32 co][32 ci]
    int ntiles;              // B * tiles per scene
};

__global__ void __launch_bounds__(WG_WAVES * 64)
conv3d_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [WG_HALO][CPAD] xn, then [WG_VOX][CPAD] g
    float *xs = wl, *gs = wl + WG_HALO * CPAD;
    const Src &s = a.c.s;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int ncib = (s.C1 + s.C2) / 32, nco = a.c.Cout / 32;
    const int cob = blockIdx.y / ncib, cib = blockIdx.y % ncib;
    const int tap0 = wave * WG_TAPS;
    f32x16 acc[WG_TAPS];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    int toff[WG_TAPS];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t) {
        const int tap = min(tap0 + t, 26);
        toff[t] = (((tap / 9 - 1) * WG_PY + ((tap / 3) % 3 - 1)) * WG_PX + (tap % 3 - 1)) * CPAD;
    }
    const int tiles_per_scene = a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_scene;
        int t = tile - b * tiles_per_scene;
        const int tx = t % a.c.tiles_x; t /= a.c.tiles_x;
        const int ty = t % a.c.tiles_y, tz = t / a.c.tiles_y;
        const int x0 = tx * WG_TX, y0 = ty * WG_TY, z0 = tz * WG_TZ;
        __syncthreads();
        stage_tile(xs, a.c, b, cib, x0, y0, z0, WG_PX, WG_PY, WG_HALO, threadIdx.x, WG_WAVES * 64);
        for (int e = threadIdx.x; e < WG_VOX * 8; e += WG_WAVES * 64) {          // g tile: 8 float4 per voxel
            const int v = e >> 3, c4 = (e & 7) * 4;
            const int gx = x0 + (v & 7), gy = y0 + ((v >> 3) & 7), gz = z0 + (v >> 6);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (gx < s.W && gy < s.H && gz < s.D)
                val = *reinterpret_cast<const f32x4 *>(a.g + ((((size_t)b * s.D + gz) * s.H + gy) * s.W + gx) * a.c.Cout + cob * 32 + c4);
            float *d = gs + v * CPAD + c4;
            d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
        }
        __syncthreads();
#pragma unroll 1
        for (int zy = 0; zy < WG_TZ * WG_TY; ++zy) {
            const int z = zy >> 3, y = zy & 7;
            const float *grow = gs + (zy * 8 + kk) * CPAD + i;
            const float *xrow = xs + (((z + 1) * WG_PY + (y + 1)) * WG_PX + 1 + kk) * CPAD + i;
#pragma unroll
            for (int xp = 0; xp < 4; ++xp) {                              // voxels 2xp + kk of this row
                const float av = grow[2 * xp * CPAD];
#pragma unroll
                for (int t = 0; t < WG_TAPS; ++t) acc[t] = mfma(av, xrow[2 * xp * CPAD + toff[t]], acc[t]);
            }
        }
    }
    // partial[chunk][cob][cib][tap][co][ci]: lane (ci = i, h = kk) register r = co chan_of(r, kk)
    float *dst = a.partial + (((size_t)blockIdx.x * nco + cob) * ncib + cib) * 27 * 1024;
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t) {
        const int tap = tap0 + t;
        if (tap >= 27) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(size_t)tap * 1024 + chan_of(r, kk) * 32 + i] = acc[t][r];
    }
}

// dW[co][ci][tap] (torch layout [Cout][Cin][3][3][3]) = sum over chunks, in chunk order
__global__ void __launch_bounds__(256)
conv3d_wgrad_reduce_kernel(const float *partial, int chunks, int Cout, int Cin, float *dw) {
    const int nco = Cout / 32, ncib = Cin / 32;
    const size_t total = (size_t)Cout * Cin * 27, per_chunk = (size_t)nco * ncib * 27 * 1024;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        // e indexes the partial layout (coalesced reads): [cob][cib][tap][co][ci]
        const int ci = (int)(e & 31), co = (int)((e >> 5) & 31);
        size_t r = e >> 10;
        const int tap = (int)(r % 27); r /= 27;
        const int cib = (int)(r % ncib), cob = (int)(r / ncib);
        float sum = 0.0f;
        for (int c = 0; c < chunks; ++c) sum += partial[(size_t)c * per_chunk + e];
        dw[((size_t)(cob * 32 + co) * Cin + cib * 32 + ci) * 27 + tap] = sum;
    }
}

// ---- weight gradient on the f16 matrix core (split operands, "f16x3") ---------------------------------------------
// The contraction index of dW[co][ci][tap] = sum_v g[v][co] xn[v + tap][ci] is the VOXEL, so both MFMA operands want eight
// consecutive voxels of one channel per lane (v_mfma_f32_32x32x16_f16: lane = (channel, k-group), k-group 0 / 1 = two x-rows
// of eight voxels): the staging TRANSPOSES the channels-last tensors into per-channel planes of IEEE-half hi / lo pairs
// (hi = rtz_half(v), lo = half(v - hi): 21-22 mantissa bits; g is first scaled by the power of two that brings its largest
// element to ~2^10 -- output gradients sit far below the half range -- and the reduce kernel scales dW back, exactly).
// One workgroup = one (cout block, cin block) pair over a chunk of 8 x 8 x 2 tiles, four waves x seven taps (112 accumulator
// registers); per k-step (16 voxels) a wave reads the g fragment once and each halo row (y + dy, z + dz) of xn once: the
// row's ten voxels are five dwords, and the three dx windows are dwords 0-3, the byte-aligned middle, dwords 1-4.
// 79 KB of LDS and 4 waves: two workgroups per CU, one's staging under the other's MFMAs.
constexpr int WH_TZ = 2, WH_ROWS = (WH_TZ + 2) * 10;
constexpr int WH_ROWB = 24;                              // a halo row in one channel's plane: 10 voxels + 2 pad, 2 bytes each
constexpr int WH_XCH = WH_ROWS * WH_ROWB + 8;            // 968 B = 242 dwords per channel: the 16 lanes of an 8-byte read pass tile the 32 banks
constexpr int WH_XPLANE = 32 * WH_XCH;
constexpr int WH_GROWS = WH_TZ * 8;
constexpr int WH_GCH = WH_GROWS * 16 + 16;               // 272 B per channel (68 dwords: 16-byte reads of 8 lanes tile the banks)
constexpr int WH_GPLANE = 32 * WH_GCH;
constexpr size_t WH_LDS = 2 * (size_t)WH_XPLANE + 2 * (size_t)WH_GPLANE;
constexpr int WH_THREADS = 256, WH_XITEMS = WH_ROWS * 5 * 8, WH_GITEMS = WH_GROWS * 4 * 8;
constexpr int WH_XITERS = (WH_XITEMS + WH_THREADS - 1) / WH_THREADS, WH_GITERS = WH_GITEMS / WH_THREADS;
static_assert(2 * WH_LDS <= 160 * 1024, "two workgroups per CU");

struct WgradHArgs {
    WgradArgs w;
    const float *g_absmax;      // device scalar max |g| (or null: no rescale)
    int ss_cin;                 // channels per scene of scale_shift's rows when they are more than this launch's (0: its own Cin)
    const int *list;            // null: every tile.  Else [0] = n, [1 .. n] = the tiles to visit, ascending (wgrad_tile_list_kernel):
                                // the staged input is then x * scale WITHOUT the shift (zero wherever x is), and the shift's share of dW
                                // is the rank-one term the reduce kernel adds (vt_conv3d_wgrad_f16x3_sparse)
};

__device__ __forceinline__ float pow2_scale_for(const float *absmax) {
    if (!absmax) return 1.0f;
    const float m = *absmax;
    if (!(m > 0.0f && m < 3.0e38f)) return 1.0f;
    const int e = 10 - ilogbf(m);                                       // 2^e * m in [2^10, 2^11)
    return ldexpf(1.0f, e < -100 ? -100 : (e > 100 ? 100 : e));
}

// hi / lo half pairs of two values: hi by round-toward-zero (one v_cvt_pkrtz), lo = half(v - hi)
__device__ __forceinline__ void split_pair_h(float a, float b, unsigned &hi, unsigned &lo) {
    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// the k-steps of one staged tile for wave W (taps 7W .. 7W+6)
template <int W>
__device__ __forceinline__ void wgrad_h_ksteps(const char *xh, const char *xl, const char *gh, const char *gl, f32x16 (&acc)[7], int lane) {
    const int ch = lane & 31, kg = lane >> 5;
    constexpr int T0 = 7 * W, T1 = (T0 + 7 < 27) ? T0 + 7 : 27;
#pragma unroll 1
    for (int ks = 0; ks < WH_TZ * 4; ++ks) {
        const int z = ks >> 2, y = (ks & 3) * 2 + kg;
        const int goff = ch * WH_GCH + (z * 8 + y) * 16;
        const f16x8 ah = *reinterpret_cast<const f16x8 *>(gh + goff), al = *reinterpret_cast<const f16x8 *>(gl + goff);
        // the wave's halo rows one after the other, a scheduling barrier between them (the scheduler would otherwise hoist every
        // row's loads to the top: 40 registers the prefetched tile needs; requesting row r + 1 under row r's MFMAs spilled 63
        // registers -- the other workgroup on the SIMD covers the LDS latency instead)
        struct Row { unsigned dh[5], dl[5]; u32x4 ph, pl; };            // ph / pl: dwords 1-4 again, as an aligned register tuple
        auto load_row = [&](int r) {
            Row o;
            const int dz = r / 3 - 1, dy = r % 3 - 1;
            const int xoff = ch * WH_XCH + ((z + dz + 1) * 10 + (y + dy + 1)) * WH_ROWB;
            const u32x2 a0 = *reinterpret_cast<const u32x2 *>(xh + xoff), a1 = *reinterpret_cast<const u32x2 *>(xh + xoff + 8);
            const u32x2 b0 = *reinterpret_cast<const u32x2 *>(xl + xoff), b1 = *reinterpret_cast<const u32x2 *>(xl + xoff + 8);
            o.dh[0] = a0.x; o.dh[1] = a0.y; o.dh[2] = a1.x; o.dh[3] = a1.y; o.dh[4] = *reinterpret_cast<const unsigned *>(xh + xoff + 16);
            o.dl[0] = b0.x; o.dl[1] = b0.y; o.dl[2] = b1.x; o.dl[3] = b1.y; o.dl[4] = *reinterpret_cast<const unsigned *>(xl + xoff + 16);
            if (3 * r + 2 >= T0 && 3 * r + 2 < T1) {
                // the dx = +1 window (dwords 1-4) read a second time, 4-byte aligned (two ds_read2_b32): as registers 1-4 of the
                // five above it is not an MFMA operand tuple, and copying it into one cost eight v_mov per tap
                typedef unsigned u1 __attribute__((aligned(4)));
                const u1 *qh = reinterpret_cast<const u1 *>(xh + xoff + 4), *ql = reinterpret_cast<const u1 *>(xl + xoff + 4);
                o.ph = u32x4{qh[0], qh[1], qh[2], qh[3]}; o.pl = u32x4{ql[0], ql[1], ql[2], ql[3]};
            }
            return o;
        };
        constexpr int R0 = T0 / 3, R1 = (T1 - 1) / 3;                     // first and last halo row with a tap of this wave
#pragma unroll
        for (int r = R0; r <= R1; ++r) {
            const Row cur = load_row(r);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned *dh = cur.dh, *dl = cur.dl;
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int t = 3 * r + dxi;
                if (t < T0 || t >= T1) continue;
                u32x4 bh, bl;
                if (dxi == 0) { bh = u32x4{dh[0], dh[1], dh[2], dh[3]}; bl = u32x4{dl[0], dl[1], dl[2], dl[3]}; }
                else if (dxi == 2) { bh = cur.ph; bl = cur.pl; }
                else {
                    bh = u32x4{__builtin_amdgcn_alignbyte(dh[1], dh[0], 2), __builtin_amdgcn_alignbyte(dh[2], dh[1], 2),
                               __builtin_amdgcn_alignbyte(dh[3], dh[2], 2), __builtin_amdgcn_alignbyte(dh[4], dh[3], 2)};
                    bl = u32x4{__builtin_amdgcn_alignbyte(dl[1], dl[0], 2), __builtin_amdgcn_alignbyte(dl[2], dl[1], 2),
                               __builtin_amdgcn_alignbyte(dl[3], dl[2], 2), __builtin_amdgcn_alignbyte(dl[4], dl[3], 2)};
                }
                const f16x8 xhv = __builtin_bit_cast(f16x8, bh), xlv = __builtin_bit_cast(f16x8, bl);
                f32x16 &c = acc[t - T0];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xhv, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xlv, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xhv, c, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#ifndef VT_WH_WPS
#define VT_WH_WPS 2                 // waves per SIMD the register budget is sized for (A/B: 1 = 512 registers, one workgroup per CU)
#endif
// everything a wave does, instantiated per wave index: one straight tile loop per variant (a wave switch INSIDE the loop made the
// allocator shuffle the 112 accumulator registers at every join: 48 v_mov_b64 per k-step)
template <int W, bool LIST>
__device__ __forceinline__ void wgrad_h_wave(const WgradHArgs &ha, char *whl) {
    char *xh = whl, *xl = whl + WH_XPLANE, *gh = whl + 2 * WH_XPLANE, *gl = gh + WH_GPLANE;
    const WgradArgs &a = ha.w;
    const Src &s = a.c.s;
    const int lane = threadIdx.x & 63;
    constexpr int wave = W;
    const int Cin = s.C1 + s.C2, ncib = Cin / 32, nco = a.c.Cout / 32;
    const int cob = blockIdx.y / ncib, cib = blockIdx.y % ncib;
    const float pre = pow2_scale_for(ha.g_absmax);
    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // staging plan: item -> (halo row, x pair, four channels); the channel quad is the thread's for the whole kernel
    const int q = threadIdx.x & 7, chq = cib * 32 + q * 4;
    const bool from_low = chq >= s.C1;
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    const float *xbase = from_low ? s.low : s.skip;
    const int tiles_per_scene = a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;
    // A tile's global loads are issued one tile ahead (fetch: 18 x 16 bytes per thread, in flight under the previous tile's
    // k-steps) and turned into the half planes when the k-steps are done with the LDS (commit).
    f32x4 px0[WH_XITERS], px1[WH_XITERS], pg0[WH_GITERS], pg1[WH_GITERS], sc, sh;
    unsigned inmask = 0;
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_scene;
        int t = tile - b * tiles_per_scene;
        const int tx = t % a.c.tiles_x; t /= a.c.tiles_x;
        const int ty = t % a.c.tiles_y, tz = t / a.c.tiles_y;
        const int x0 = tx * 8, y0 = ty * 8, z0 = tz * WH_TZ;
        sc = f32x4{1.f, 1.f, 1.f, 1.f}; sh = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.c.scale_shift) {
            const float *ss = a.c.scale_shift + ((size_t)b * (ha.ss_cin ? ha.ss_cin : Cin) + chq) * 2;
            sc = f32x4{ss[0], ss[2], ss[4], ss[6]};
            if (!LIST) sh = f32x4{ss[1], ss[3], ss[5], ss[7]};
        }
        inmask = 0;
#pragma unroll
        for (int it = 0; it < WH_XITERS; ++it) {
            const int id = min((int)threadIdx.x + it * WH_THREADS, WH_XITEMS - 1);      // (the last iteration's idle threads repeat an item)
            const int rp = id >> 3, p = rp % 5, row = rp / 5, py = row % 10, pz = row / 10;
            const int gx = x0 - 1 + 2 * p, gy = y0 - 1 + py, gz = z0 - 1 + pz;        // gx is odd: the pair's `low` voxels are neighbours too
            const bool rowin = (unsigned)gy < (unsigned)s.H && (unsigned)gz < (unsigned)s.D;
            if (rowin && gx >= 0) inmask |= 1u << (2 * it);
            if (rowin && gx + 1 < s.W) inmask |= 2u << (2 * it);
            // branch-free: both loads always issue, from addresses clamped into the volume (the zero padding is a select at
            // commit); 32-bit element offsets from one uniform base pointer
            const int cz = min(max(gz, 0), s.D - 1), cy = min(max(gy, 0), s.H - 1), cx0 = max(gx, 0), cx1 = min(gx + 1, s.W - 1);
            unsigned voff0, voff1;
            if (from_low) {
                voff0 = (unsigned)(((b * D2 + (cz >> 1)) * H2 + (cy >> 1)) * W2 + (cx0 >> 1)) * (unsigned)s.C2 + (unsigned)(chq - s.C1);
                voff1 = voff0 + (unsigned)((cx1 >> 1) - (cx0 >> 1)) * (unsigned)s.C2;
            } else {
                voff0 = (unsigned)(((b * s.D + cz) * s.H + cy) * s.W + cx0) * (unsigned)s.C1 + (unsigned)chq;
                voff1 = voff0 + (unsigned)(cx1 - cx0) * (unsigned)s.C1;
            }
            px0[it] = *reinterpret_cast<const f32x4 *>(xbase + voff0);
            px1[it] = *reinterpret_cast<const f32x4 *>(xbase + voff1);
        }
#pragma unroll
        for (int it = 0; it < WH_GITERS; ++it) {
            const int id = threadIdx.x + it * WH_THREADS;
            const int rp = id >> 3, p = rp & 3, row = rp >> 2, y = row & 7, z = row >> 3;
            const unsigned goff = (unsigned)(((b * s.D + z0 + z) * s.H + y0 + y) * s.W + x0 + 2 * p) * (unsigned)a.c.Cout + (unsigned)(cob * 32 + q * 4);
            pg0[it] = *reinterpret_cast<const f32x4 *>(a.g + goff);
            pg1[it] = *reinterpret_cast<const f32x4 *>(a.g + goff + (unsigned)a.c.Cout);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int it = 0; it < WH_XITERS; ++it) {
            const int id = threadIdx.x + it * WH_THREADS;
            if (id >= WH_XITEMS) continue;
            const int rp = id >> 3, p = rp % 5, row = rp / 5;
            const bool in0 = inmask >> (2 * it) & 1u, in1 = inmask >> (2 * it + 1) & 1u;
            const int off = (q * 4) * WH_XCH + row * WH_ROWB + p * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned hi, lo;
                split_pair_h(in0 ? fmaf(px0[it][c], sc[c], sh[c]) : 0.0f, in1 ? fmaf(px1[it][c], sc[c], sh[c]) : 0.0f, hi, lo);
                *reinterpret_cast<unsigned *>(xh + off + c * WH_XCH) = hi;
                *reinterpret_cast<unsigned *>(xl + off + c * WH_XCH) = lo;
            }
        }
#pragma unroll
        for (int it = 0; it < WH_GITERS; ++it) {
            const int id = threadIdx.x + it * WH_THREADS;
            const int rp = id >> 3, p = rp & 3, row = rp >> 2;
            const int off = (q * 4) * WH_GCH + row * 16 + p * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned hi, lo;
                split_pair_h(pg0[it][c] * pre, pg1[it][c] * pre, hi, lo);
                *reinterpret_cast<unsigned *>(gh + off + c * WH_GCH) = hi;
                *reinterpret_cast<unsigned *>(gl + off + c * WH_GCH) = lo;
            }
        }
    };
    if constexpr (LIST) {
        // the list's tiles, every gridDim.x-th; an entry is read one visit ahead of its fetch
        const int *list = ha.list;
        const int nvisit = __builtin_amdgcn_readfirstlane(list[0]), G = gridDim.x;
        auto tile_at = [&](int i) { return __builtin_amdgcn_readfirstlane(list[1 + min(i, nvisit - 1)]); };
        int t_ahead = 0;
        if ((int)blockIdx.x < nvisit) { fetch(tile_at(blockIdx.x)); t_ahead = tile_at(blockIdx.x + G); }
        for (int i = blockIdx.x; i < nvisit; i += G) {
            __syncthreads();
            commit();
            __syncthreads();
            if (i + G < nvisit) { fetch(t_ahead); t_ahead = tile_at(i + 2 * G); }
            wgrad_h_ksteps<W>(xh, xl, gh, gl, acc, lane);
        }
    } else {
        if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
        for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
            __syncthreads();                                                   // the previous tile's k-steps are done with the planes
            commit();
            __syncthreads();
            if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
            wgrad_h_ksteps<W>(xh, xl, gh, gl, acc, lane);
        }
    }
    // partial[chunk][cob][cib][tap][co][ci]: lane (ci, half kk) register r = co chan_of(r, kk)
    float *dst = a.partial + (((size_t)blockIdx.x * nco + cob) * ncib + cib) * 27 * 1024;
    const int i = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int tap = wave * 7 + t;
        if (tap >= 27) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(size_t)tap * 1024 + chan_of(r, kk) * 32 + i] = acc[t][r];
    }
}

template <bool LIST = false>
__global__ void __launch_bounds__(WH_THREADS, VT_WH_WPS)
conv3d_wgrad_h_kernel(WgradHArgs ha) {
    extern __shared__ __attribute__((aligned(16))) char whl[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) wgrad_h_wave<0, LIST>(ha, whl);
    else if (wave == 1) wgrad_h_wave<1, LIST>(ha, whl);
    else if (wave == 2) wgrad_h_wave<2, LIST>(ha, whl);
    else wgrad_h_wave<3, LIST>(ha, whl);
}

// ---- weight gradient of a decoder-entry layer's UPSAMPLED channels, per output parity --------------------------------------
// xn[v + t] of a nearest-upsampled channel is xl[(v + t) >> 1] (xl = the normalised low-resolution field, zero outside), so with
// v = 2 u + p (p in {0,1}^3) the tap t reads xl[u + d], d = (p + t) >> 1 per axis: d in {-1, 0} for p = 0, {0, +1} for p = 1.
//     dW[t] = sum_p C_p[(p + t) >> 1],   C_p[d] = sum_u g[2 u + p] (x) xl[u + d]
// -- per parity class a 2 x 2 x 2-tap weight gradient over the LOW-resolution grid: 64 products of N / 8 voxels instead of 27 of N
// (0.30 of the MFMAs).  The kernel is conv3d_wgrad_h_kernel on the coarse grid: blockIdx.z = (p_z, p_y), both p_x in turn, the output
// gradient's tile gathered from the voxels of that class (x stride two), the halo of xl staged as there; wave w takes the rows
// d_z = p_z - 1 + (w >> 1), d_y = p_y - 1 + (w & 1) and both d_x.  partial[chunk][class][pair][w * 2 + i][co][ci];
// conv3d_wgrad_reduce_parity_kernel adds chunks and classes in order.
template <int W>
__device__ __forceinline__ void wgrad_hp_wave(const WgradHArgs &ha, char *whl) {
    char *xh = whl, *xl = whl + WH_XPLANE, *gh = whl + 2 * WH_XPLANE, *gl = gh + WH_GPLANE;
    const WgradArgs &a = ha.w;
    const Src &s = a.c.s;
    const int lane = threadIdx.x & 63;
    const int Cin = s.C1 + s.C2, ncib = s.C2 / 32, nco = a.c.Cout / 32;
    const int cob = blockIdx.y / ncib, cib = blockIdx.y % ncib;
    // a workgroup takes the two classes (p_z, p_y, 0) and (p_z, p_y, 1): they read the same halo rows of xl, staged once per tile
    const int pz = blockIdx.z >> 1, py = blockIdx.z & 1;
    const float pre = pow2_scale_for(ha.g_absmax);
    f32x16 acc[4];                                                          // [p_x][d_x - (p_x - 1)]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int q = threadIdx.x & 7, chq = cib * 32 + q * 4;                 // channel quad among the low channels
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    const int tiles_per_scene = a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;   // coarse tiles
    f32x4 px0[WH_XITERS], px1[WH_XITERS], pg0[WH_GITERS], pg1[WH_GITERS], sc, sh;
    unsigned inmask = 0;
    auto origin = [&](int tile, int &b, int &x0, int &y0, int &z0) {
        b = tile / tiles_per_scene;
        int t = tile - b * tiles_per_scene;
        const int tx = t % a.c.tiles_x; t /= a.c.tiles_x;
        const int ty = t % a.c.tiles_y, tz = t / a.c.tiles_y;
        x0 = tx * 8; y0 = ty * 8; z0 = tz * WH_TZ;                          // coarse
    };
    auto fetch_x = [&](int tile) {
        int b, x0, y0, z0;
        origin(tile, b, x0, y0, z0);
        {
            const float *ss = a.c.scale_shift + ((size_t)b * Cin + s.C1 + chq) * 2;
            sc = f32x4{ss[0], ss[2], ss[4], ss[6]}; sh = f32x4{ss[1], ss[3], ss[5], ss[7]};
        }
        inmask = 0;
#pragma unroll
        for (int it = 0; it < WH_XITERS; ++it) {
            const int id = min((int)threadIdx.x + it * WH_THREADS, WH_XITEMS - 1);
            const int rp = id >> 3, p = rp % 5, row = rp / 5, yy = row % 10, zz = row / 10;
            const int gx = x0 - 1 + 2 * p, gy = y0 - 1 + yy, gz = z0 - 1 + zz;
            const bool rowin = (unsigned)gy < (unsigned)H2 && (unsigned)gz < (unsigned)D2;
            if (rowin && gx >= 0) inmask |= 1u << (2 * it);
            if (rowin && gx + 1 < W2) inmask |= 2u << (2 * it);
            const int cz = min(max(gz, 0), D2 - 1), cy = min(max(gy, 0), H2 - 1), cx0 = max(gx, 0), cx1 = min(gx + 1, W2 - 1);
            const unsigned voff0 = (unsigned)(((b * D2 + cz) * H2 + cy) * W2 + cx0) * (unsigned)s.C2 + (unsigned)chq;
            const unsigned voff1 = voff0 + (unsigned)(cx1 - cx0) * (unsigned)s.C2;
            px0[it] = *reinterpret_cast<const f32x4 *>(s.low + voff0);
            px1[it] = *reinterpret_cast<const f32x4 *>(s.low + voff1);
        }
    };
    auto fetch_g = [&](int tile, int px) {
        int b, x0, y0, z0;
        origin(tile, b, x0, y0, z0);
#pragma unroll
        for (int it = 0; it < WH_GITERS; ++it) {
            const int id = threadIdx.x + it * WH_THREADS;
            const int rp = id >> 3, p = rp & 3, row = rp >> 2, y = row & 7, z = row >> 3;
            // the class's voxels: fine coordinate 2 * coarse + parity
            const unsigned goff = (unsigned)(((b * s.D + 2 * (z0 + z) + pz) * s.H + 2 * (y0 + y) + py) * s.W + 2 * (x0 + 2 * p) + px) * (unsigned)a.c.Cout
                                  + (unsigned)(cob * 32 + q * 4);
            pg0[it] = *reinterpret_cast<const f32x4 *>(a.g + goff);
            pg1[it] = *reinterpret_cast<const f32x4 *>(a.g + goff + 2u * (unsigned)a.c.Cout);
        }
    };
    auto commit_x = [&]() {
#pragma unroll
        for (int it = 0; it < WH_XITERS; ++it) {
            const int id = threadIdx.x + it * WH_THREADS;
            if (id >= WH_XITEMS) continue;
            const int rp = id >> 3, p = rp % 5, row = rp / 5;
            const bool in0 = inmask >> (2 * it) & 1u, in1 = inmask >> (2 * it + 1) & 1u;
            const int off = (q * 4) * WH_XCH + row * WH_ROWB + p * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned hi, lo;
                split_pair_h(in0 ? fmaf(px0[it][c], sc[c], sh[c]) : 0.0f, in1 ? fmaf(px1[it][c], sc[c], sh[c]) : 0.0f, hi, lo);
                *reinterpret_cast<unsigned *>(xh + off + c * WH_XCH) = hi;
                *reinterpret_cast<unsigned *>(xl + off + c * WH_XCH) = lo;
            }
        }
    };
    auto commit_g = [&]() {
#pragma unroll
        for (int it = 0; it < WH_GITERS; ++it) {
            const int id = threadIdx.x + it * WH_THREADS;
            const int rp = id >> 3, p = rp & 3, row = rp >> 2;
            const int off = (q * 4) * WH_GCH + row * 16 + p * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned hi, lo;
                split_pair_h(pg0[it][c] * pre, pg1[it][c] * pre, hi, lo);
                *reinterpret_cast<unsigned *>(gh + off + c * WH_GCH) = hi;
                *reinterpret_cast<unsigned *>(gl + off + c * WH_GCH) = lo;
            }
        }
    };
    const int dz = pz - 1 + (W >> 1), dy = py - 1 + (W & 1);
    const int ch = lane & 31, kg = lane >> 5;
    auto ksteps = [&](auto pxc, f32x16 &c0, f32x16 &c1) {
        constexpr int PX = decltype(pxc)::value;
#pragma unroll 1
        for (int ks = 0; ks < WH_TZ * 4; ++ks) {
            const int z = ks >> 2, y = (ks & 3) * 2 + kg;
            const int goff = ch * WH_GCH + (z * 8 + y) * 16;
            const f16x8 ah = *reinterpret_cast<const f16x8 *>(gh + goff), al = *reinterpret_cast<const f16x8 *>(gl + goff);
            const int xoff = ch * WH_XCH + ((z + dz + 1) * 10 + (y + dy + 1)) * WH_ROWB;
            // the row's ten voxels as five dwords (hi and lo planes); windows: d_x = -1 dwords 0-3, d_x = 0 the byte-aligned middle,
            // d_x = +1 dwords 1-4
            const u32x2 a0 = *reinterpret_cast<const u32x2 *>(xh + xoff), a1 = *reinterpret_cast<const u32x2 *>(xh + xoff + 8);
            const u32x2 b0 = *reinterpret_cast<const u32x2 *>(xl + xoff), b1 = *reinterpret_cast<const u32x2 *>(xl + xoff + 8);
            const unsigned a4 = *reinterpret_cast<const unsigned *>(xh + xoff + 16), b4 = *reinterpret_cast<const unsigned *>(xl + xoff + 16);
            const u32x4 mh = u32x4{__builtin_amdgcn_alignbyte(a0.y, a0.x, 2), __builtin_amdgcn_alignbyte(a1.x, a0.y, 2),
                                   __builtin_amdgcn_alignbyte(a1.y, a1.x, 2), __builtin_amdgcn_alignbyte(a4, a1.y, 2)};
            const u32x4 ml = u32x4{__builtin_amdgcn_alignbyte(b0.y, b0.x, 2), __builtin_amdgcn_alignbyte(b1.x, b0.y, 2),
                                   __builtin_amdgcn_alignbyte(b1.y, b1.x, 2), __builtin_amdgcn_alignbyte(b4, b1.y, 2)};
            const u32x4 eh = PX ? u32x4{a0.y, a1.x, a1.y, a4} : u32x4{a0.x, a0.y, a1.x, a1.y};       // the class's other window
            const u32x4 el = PX ? u32x4{b0.y, b1.x, b1.y, b4} : u32x4{b0.x, b0.y, b1.x, b1.y};
            // c0: d_x = p_x - 1, c1: d_x = p_x
            const f16x8 w0h = __builtin_bit_cast(f16x8, PX ? mh : eh), w0l = __builtin_bit_cast(f16x8, PX ? ml : el);
            const f16x8 w1h = __builtin_bit_cast(f16x8, PX ? eh : mh), w1l = __builtin_bit_cast(f16x8, PX ? el : ml);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w0h, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w1h, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w0l, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w1l, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w0h, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w1h, c1, 0, 0, 0);
        }
    };
    if ((int)blockIdx.x < a.ntiles) { fetch_x(blockIdx.x); fetch_g(blockIdx.x, 0); }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();                                                   // the previous k-steps are done with the planes
        commit_x();
        commit_g();
        __syncthreads();
        fetch_g(tile, 1);
        ksteps(std::integral_constant<int, 0>{}, acc[0], acc[1]);
        __syncthreads();
        commit_g();
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) { fetch_x(tile + gridDim.x); fetch_g(tile + gridDim.x, 0); }
        ksteps(std::integral_constant<int, 1>{}, acc[2], acc[3]);
    }
    // partial[chunk][class][pair][W * 2 + i][co][ci]: lane (ci, half kk) register r = co chan_of(r, kk)
    const int i = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        const int par = (int)blockIdx.z * 2 + px;
        float *dst = a.partial + ((((size_t)blockIdx.x * 8 + par) * (nco * ncib) + blockIdx.y) * 8 + W * 2) * 1024;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)t * 1024 + chan_of(r, kk) * 32 + i] = acc[px * 2 + t][r];
    }
}

__global__ void __launch_bounds__(WH_THREADS, VT_WH_WPS)
conv3d_wgrad_hp_kernel(WgradHArgs ha) {
    extern __shared__ __attribute__((aligned(16))) char whl[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) wgrad_hp_wave<0>(ha, whl);
    else if (wave == 1) wgrad_hp_wave<1>(ha, whl);
    else if (wave == 2) wgrad_hp_wave<2>(ha, whl);
    else wgrad_hp_wave<3>(ha, whl);
}

// dW[co][C1 + ci][tap] = (sum over chunks and parity classes, in that order, of the class's product for d = (p + t) >> 1) / scale:
// one thread per (co, ci, tap) of a (cout block, low cin block) pair
__global__ void __launch_bounds__(256)
conv3d_wgrad_reduce_parity_kernel(const float *partial, int chunks, int Cout, int C1, int C2, const float *g_absmax, float *dw) {
    const int nco = Cout / 32, ncib = C2 / 32, npairs = nco * ncib;
    const size_t total = (size_t)Cout * C2 * 27;
    const float post = 1.0f / pow2_scale_for(g_absmax);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int ci = (int)(e & 31), co = (int)((e >> 5) & 31);
        size_t r = e >> 10;
        const int tap = (int)(r % 27); r /= 27;
        const int pair = (int)r, cob = pair / ncib, cib = pair % ncib;
        const int tz = tap / 9 - 1, ty = (tap / 3) % 3 - 1, tx = tap % 3 - 1;
        float sum = 0.0f;
        for (int c = 0; c < chunks; ++c)
#pragma unroll
            for (int par = 0; par < 8; ++par) {
                const int pz = par >> 2, py = (par >> 1) & 1, px = par & 1;
                // row index per axis: d - (p - 1) with d = (p + t) >> 1 (arithmetic)
                const int rz = ((pz + tz) >> 1) - pz + 1, ry = ((py + ty) >> 1) - py + 1, rx = ((px + tx) >> 1) - px + 1;
                sum += partial[((((size_t)c * 8 + par) * npairs + pair) * 8 + (rz * 2 + ry) * 2 + rx) * 1024 + co * 32 + ci];
            }
        dw[((size_t)(cob * 32 + co) * (C1 + C2) + C1 + cib * 32 + ci) * 27 + tap] = sum * post;
    }
}

// ---- weight gradient of a layer whose input is mostly EXACT zeros (the network's first layer on a scene's mean grid) -------------
// xn = x * scale + shift inside the volume and 0 outside, so
//     dW[co][ci][tap] = sum_v g[v][co] (x * scale)[v + tap][ci]  +  shift[ci] * sum_{v : v + tap inside} g[v][co]
// and the first sum has no term from a tile over whose halo x is zero: the 8 x 8 x 2 tiles of a block flagged by the voxel sort
// (bit 0: no point in the block's 10^3 halo; unet3d's first-layer skip reads the same bytes).  wgrad_tile_list_kernel lists the other
// tiles in ascending order (so the kernel's summation order is fixed), wgrad_gsum_plane_kernel / wgrad_gsum_final_kernel leave the 27
// box sums of g per scene, the reduce kernel adds the rank-one term.
__global__ void __launch_bounds__(1024)
wgrad_tile_list_kernel(const unsigned char *flags, int ntiles, int tiles_x, int tiles_y, int tiles_z, int *list) {
    __shared__ int cnt[1024];
    const int per = (ntiles + 1023) / 1024, t0 = threadIdx.x * per, t1 = min(t0 + per, ntiles);
    const int tps = tiles_x * tiles_y * tiles_z, t8z = (tiles_z * WH_TZ) >> 3;
    auto keep = [&](int tile) {
        const int b = tile / tps;
        int t = tile - b * tps;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y, tz = t / tiles_y;
        return !(flags[(size_t)b * t8z * tiles_y * tiles_x + ((size_t)((tz * WH_TZ) >> 3) * tiles_y + ty) * tiles_x + tx] & 1);
    };
    int n = 0;
    for (int t = t0; t < t1; ++t) n += keep(t) ? 1 : 0;
    cnt[threadIdx.x] = n;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                           // inclusive scan
        const int v = threadIdx.x >= d ? cnt[threadIdx.x - d] : 0;
        __syncthreads();
        cnt[threadIdx.x] += v;
        __syncthreads();
    }
    int o = cnt[threadIdx.x] - n;
    for (int t = t0; t < t1; ++t)
        if (keep(t)) list[1 + o++] = t;
    if (threadIdx.x == 1023) list[0] = cnt[1023];
}

// P[b][z][yc][xc][Cout]: sums of g over plane z of scene b by row class yc (y = 0, inner, y = H - 1) and xc = (all x, x = 0, x = W - 1).
// Thread = (x lane 0..31, four channels); grid (D, B).
__global__ void __launch_bounds__(256)
wgrad_gsum_plane_kernel(const float *g, int D, int H, int W, int Cout, float *P) {
    __shared__ float red[4][9][8][4];
    const int z = blockIdx.x, b = blockIdx.y, q = threadIdx.x & 7, vl = threadIdx.x >> 3, wave = threadIdx.x >> 6;
    for (int cb = 0; cb < Cout; cb += 32) {
        f32x4 acc[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[i][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *pl = g + ((size_t)((b * D + z) * H) * W) * Cout + cb + q * 4;
        for (int x = vl; x < W; x += 32) {
            const float *col = pl + (size_t)x * Cout;
            const size_t pitch = (size_t)W * Cout;
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(col), r1 = *reinterpret_cast<const f32x4 *>(col + (size_t)(H - 1) * pitch);
            f32x4 mid = {0.f, 0.f, 0.f, 0.f};
            for (int y = 1; y < H - 1; y += 8) {                    // eight rows in flight per thread
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(col + (size_t)min(y + u, H - 2) * pitch);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (y + u < H - 1) mid += v[u];
            }
            const f32x4 rows[3] = {r0, mid, r1};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                acc[i][0] += rows[i];
                if (x == 0) acc[i][1] += rows[i];
                if (x == W - 1) acc[i][2] += rows[i];
            }
        }
        // lanes of a wave with the same q: bits 3..5 of the lane
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][k][e];
                    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                    acc[i][k][e] = v;
                }
        __syncthreads();
        if ((threadIdx.x & 63) < 8)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) red[wave][i * 3 + k][q][e] = acc[i][k][e];
        __syncthreads();
        for (int t = threadIdx.x; t < 9 * 32; t += 256) {
            const int cls = t >> 5, c = t & 31;
            const float v = ((red[0][cls][c >> 2][c & 3] + red[1][cls][c >> 2][c & 3]) + red[2][cls][c >> 2][c & 3]) + red[3][cls][c >> 2][c & 3];
            P[(((size_t)(b * D + z)) * 9 + cls) * Cout + cb + c] = v;
        }
    }
}

// S[b][tap][co] = sum of g over the voxels v of scene b with v + tap inside the volume, from the plane sums (planes in ascending z)
__global__ void __launch_bounds__(256)
wgrad_gsum_final_kernel(const float *P, int D, int Cout, float *S) {
    const int b = blockIdx.y;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < 27 * Cout; t += gridDim.x * 256) {
        const int tap = t / Cout, co = t - tap * Cout;
        const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
        const int z0 = dz < 0 ? 1 : 0, z1 = dz > 0 ? D - 1 : D;
        float sum = 0.0f;
        for (int z = z0; z < z1; ++z) {
            const float *p = P + ((size_t)(b * D + z)) * 9 * Cout + co;
            float pl = 0.0f;
#pragma unroll
            for (int yc = 0; yc < 3; ++yc) {
                if ((dy < 0 && yc == 0) || (dy > 0 && yc == 2)) continue;
                float v = p[(yc * 3 + 0) * Cout];
                if (dx < 0) v -= p[(yc * 3 + 1) * Cout];
                if (dx > 0) v -= p[(yc * 3 + 2) * Cout];
                pl += v;
            }
            sum += pl;
        }
        S[((size_t)b * 27 + tap) * Cout + co] = sum;
    }
}

// as conv3d_wgrad_reduce_kernel, with the power-of-two scale of the output gradient taken back out.  The 512 chunk partials of
// an entry are dealt over four thread groups (each adds its contiguous quarter in chunk order, eight loads in flight), the four
// sums meet in LDS and are added in group order: a fixed tree (bit-reproducible), 4x the workgroups of one thread per entry
// (56 MB of partials per layer: 54 -> ~20 us).
// ``rank_s`` (or null): S[b][tap][co] = the sum of the UNSCALED output gradient over the voxels whose tap neighbour lies inside the
// volume; dW[co][ci][tap] += sum_b shift_b[ci] S_b[tap][co] -- the share of the GroupNorm shift, which the list form of the kernel
// above leaves out of its staged input
__global__ void __launch_bounds__(256)
conv3d_wgrad_reduce_scaled_kernel(const float *partial, int chunks, int Cout, int Cin, const float *g_absmax, float *dw,
                                  const float *rank_s = nullptr, const float *scale_shift = nullptr, int B = 0, int cin_stride = 0) {
    __shared__ float quarter[4][64];
    const int nco = Cout / 32, ncib = Cin / 32;
    const size_t total = (size_t)Cout * Cin * 27, per_chunk = (size_t)nco * ncib * 27 * 1024;
    const float post = 1.0f / pow2_scale_for(g_absmax);
    const int grp = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int c0 = chunks * grp / 4, c1 = chunks * (grp + 1) / 4;
    for (size_t base = (size_t)blockIdx.x * 64; base < total; base += (size_t)gridDim.x * 64) {
        const size_t e = base + l;                                  // total is a multiple of 1024
        float sum = 0.0f;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(c + u) * per_chunk + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += t[u];
        }
        for (; c < c1; ++c) sum += partial[(size_t)c * per_chunk + e];
        quarter[grp][l] = sum;
        __syncthreads();
        if (grp == 0) {
            const float tot4 = ((quarter[0][l] + quarter[1][l]) + quarter[2][l]) + quarter[3][l];
            const int ci = (int)(e & 31), co = (int)((e >> 5) & 31);
            size_t r = e >> 10;
            const int tap = (int)(r % 27); r /= 27;
            const int cib = (int)(r % ncib), cob = (int)(r / ncib);
            float v = tot4 * post;
            if (rank_s)
                for (int b = 0; b < B; ++b)
                    v = fmaf(scale_shift[((size_t)b * Cin + cib * 32 + ci) * 2 + 1], rank_s[((size_t)b * 27 + tap) * Cout + cob * 32 + co], v);
            dw[((size_t)(cob * 32 + co) * (cin_stride ? cin_stride : Cin) + cib * 32 + ci) * 27 + tap] = v;      // (cin_stride: the first Cin of a wider dW)
        }
        __syncthreads();
    }
}

// ---- GroupNorm backward ------------------------------------------------------------------------
// part[b][blk][c] = (sum_v dxn, sum_v dxn * x) over the block's voxels; x is the virtual concat input.
// Thread = (voxel group of 32, four channels): 16-byte loads of dxn and x, 32-bit voxel arithmetic (the scalar-load form with
// 64-bit div / mod per element ran at half the memory rate: 1.23 ms of a 16 ms training step).
__global__ void __launch_bounds__(256)
gn_bwd_stats_kernel(Src s, const float *dxn, int nblk, float *part) {
    __shared__ float red[32][8][8];
    const int b = blockIdx.y, blk = blockIdx.x;
    const int C = s.C1 + s.C2;
    const unsigned V = (unsigned)s.D * s.H * s.W;
    const unsigned v0 = (unsigned)((size_t)V * blk / nblk), v1 = (unsigned)((size_t)V * (blk + 1) / nblk);
    const int q = threadIdx.x & 7, vg = threadIdx.x >> 3;
    const unsigned HW = (unsigned)s.H * s.W;
    const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
    for (int cb = 0; cb < C; cb += 32) {
        const int ch = cb + 4 * q;
        const bool from_low = ch >= s.C1;
        f32x4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = {0.f, 0.f, 0.f, 0.f};
        for (unsigned v = v0 + vg; v < v1; v += 32) {
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dxn + ((size_t)b * V + v) * C + ch);
            f32x4 x;
            if (!from_low) {
                x = *reinterpret_cast<const f32x4 *>(s.skip + ((size_t)b * V + v) * s.C1 + ch);
            } else {
                const unsigned z = v / HW, r = v - z * HW, y = r / (unsigned)s.W, xx = r - y * (unsigned)s.W;
                x = *reinterpret_cast<const f32x4 *>(s.low + ((((size_t)b * D2 + (z >> 1)) * H2 + (y >> 1)) * W2 + (xx >> 1)) * s.C2 + (ch - s.C1));
            }
            p1.x += d.x; p1.y += d.y; p1.z += d.z; p1.w += d.w;
            p2.x = fmaf(d.x, x.x, p2.x); p2.y = fmaf(d.y, x.y, p2.y); p2.z = fmaf(d.z, x.z, p2.z); p2.w = fmaf(d.w, x.w, p2.w);
        }
        float *r = red[vg][q];
        r[0] = p1.x; r[1] = p2.x; r[2] = p1.y; r[3] = p2.y; r[4] = p1.z; r[5] = p2.z; r[6] = p1.w; r[7] = p2.w;
        __syncthreads();
        if (threadIdx.x < 64) {                                       // threadIdx.x = (channel of the 32-block) * 2 + {sum, sum x}
            float a = 0.0f;
            for (int k = 0; k < 32; ++k) a += red[k][threadIdx.x >> 3][threadIdx.x & 7];
            part[(((size_t)b * nblk + blk) * C + cb) * 2 + threadIdx.x] = a;
        }
        __syncthreads();
    }
}

// one block per (scene, group): forward statistics again (same reduction as gn_finalize_kernel) and
// the backward coefficients coef[b][c] = (A_c, B_g, C_g): dx = A dxn + B x + C;
// dgb[b][c] = (dgamma, dbeta) contributions of this scene
__global__ void __launch_bounds__(256)
gn_bwd_coeffs_kernel(StatSrc s1, StatSrc s2, const float *bpart, int nblkb, int groups, double count,
                     const float *gamma, float eps, float *coef, float *dgb) {
    __shared__ double red[256][4];
    __shared__ double stat[4];
    const int b = blockIdx.x, g = blockIdx.y;
    const int C = s1.C + s2.C, cpg = C / groups;
    double sum = 0.0, sq = 0.0;
    for (int k = 0; k < cpg; ++k) {
        const int c = g * cpg + k;
        const StatSrc &s = (c < s1.C) ? s1 : s2;
        const int cc = (c < s1.C) ? c : c - s1.C;
        const double mult = (c < s1.C) ? 1.0 : 8.0;
        for (int blk = threadIdx.x; blk < s.nblk; blk += 256) {
            const float *p = s.part + (((size_t)b * s.nblk + blk) * s.C + cc) * 2;
            sum += mult * (double)p[0]; sq += mult * (double)p[1];
        }
    }
    // per-channel backward sums over the partial blocks, all the group's channels at once (fixed order: thread (row r, channel k) adds
    // the blocks r, r + R, ... in order, then channel k's thread adds the R row sums in order; one pass and two barriers -- the form
    // with a 256-wide tree per channel spent 8 barriers per channel: 22 us per launch for 64 KB of partials)
    __shared__ double chs[256][2];
    double p1 = 0.0, p2 = 0.0;
    {
        const int R = 256 / cpg, r = threadIdx.x / cpg, k = threadIdx.x - r * cpg;       // cpg <= 256 (checked by the launcher)
        double a1 = 0.0, a2 = 0.0;
        if (r < R) {
            const int c = g * cpg + k;
            for (int blk = r; blk < nblkb; blk += R) {
                const float *p = bpart + (((size_t)b * nblkb + blk) * C + c) * 2;
                a1 += (double)p[0]; a2 += (double)p[1];
            }
        }
        red[threadIdx.x][2] = a1; red[threadIdx.x][3] = a2;
        __syncthreads();
        if ((int)threadIdx.x < cpg) {
            double s1 = 0.0, s2 = 0.0;
            for (int rr = 0; rr < R; ++rr) { s1 += red[rr * cpg + threadIdx.x][2]; s2 += red[rr * cpg + threadIdx.x][3]; }
            chs[threadIdx.x][0] = s1; chs[threadIdx.x][1] = s2;
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < cpg) { p1 = chs[threadIdx.x][0]; p2 = chs[threadIdx.x][1]; }
    red[threadIdx.x][0] = sum; red[threadIdx.x][1] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x][0] += red[threadIdx.x + o][0]; red[threadIdx.x][1] += red[threadIdx.x + o][1]; }
        __syncthreads();
    }
    const double n = count * cpg;
    if (threadIdx.x == 0) {
        const double mean = red[0][0] / n;
        double var = red[0][1] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[0] = mean; stat[1] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    const double mean = stat[0], rstd = stat[1];
    // m1 = mean_g(gamma dxn), m2 = mean_g(gamma dxn xhat)
    double t1 = 0.0, t2 = 0.0;
    if ((int)threadIdx.x < cpg) {
        const double gm = (double)gamma[g * cpg + threadIdx.x];
        t1 = gm * p1; t2 = gm * rstd * (p2 - mean * p1);
    }
    red[threadIdx.x][2] = t1; red[threadIdx.x][3] = t2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < cpg; ++k) { a1 += red[k][2]; a2 += red[k][3]; }
        stat[2] = a1 / n; stat[3] = a2 / n;
    }
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        const int c = g * cpg + threadIdx.x;
        const double m1 = stat[2], m2 = stat[3];
        float *cf = coef + ((size_t)b * C + c) * 3;
        cf[0] = (float)(rstd * (double)gamma[c]);
        cf[1] = (float)(-rstd * rstd * m2);
        cf[2] = (float)(-rstd * m1 + rstd * rstd * m2 * mean);
        dgb[((size_t)b * C + c) * 2 + 0] = (float)(rstd * (p2 - mean * p1));
        dgb[((size_t)b * C + c) * 2 + 1] = (float)p1;
    }
}

// dskip[v][c] = A dxn + B x + C  (c < C1);   dlow[v2][c] = sum over the 8 children of the same (c >= C1)
// mask bit 0 / bit 1: `skip` / `low` is the ReLU output of the layer in front and this gradient is that layer's only one, so the
// layer's relu_mask pass happens here -- (x > 0 ? d : 0) with the running max |.| for its power-of-two rescale (amax_skip /
// amax_low: cells the launcher zeroed, one atomicMax per workgroup as relu_mask_kernel) -- instead of in a pass of its own
__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(Src s, const float *dxn, const float *coef, float *dskip, float *dlow, int mask, unsigned *amax_skip, unsigned *amax_low,
                    const float *dgb, float *dgb_sum, int B) {
    const int C = s.C1 + s.C2;
    const size_t V = (size_t)s.D * s.H * s.W;
    const int b = blockIdx.y;
    // (one workgroup also adds the scenes' (dgamma, dbeta) contributions in scene order: [2][C], what the framework's sum + transpose
    // + copy took two launches for)
    if (dgb_sum && blockIdx.x == 0 && b == 0)
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            const int c = i % C, q = i / C;
            float a = 0.0f;
            for (int bb = 0; bb < B; ++bb) a += dgb[((size_t)bb * C + c) * 2 + q];
            dgb_sum[i] = a;
        }
    auto bits = [](float x) { return __builtin_bit_cast(unsigned, x) & 0x7fffffffu; };
    __shared__ unsigned wmax[2][4];
    unsigned us = 0, ul = 0;
    if (dskip) {
        const size_t total = V * (s.C1 / 4);
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            const int c4 = (int)(e % (s.C1 / 4)) * 4;
            const size_t v = e / (s.C1 / 4);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dxn + ((size_t)b * V + v) * C + c4);
            const f32x4 x = *reinterpret_cast<const f32x4 *>(s.skip + ((size_t)b * V + v) * s.C1 + c4);
            const float *cf = coef + ((size_t)b * C + c4) * 3;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaf(cf[3 * k], d[k], fmaf(cf[3 * k + 1], x[k], cf[3 * k + 2]));
            if (mask & 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[k] = x[k] > 0.f ? o[k] : 0.f; us = max(us, bits(o[k])); }
            }
            *reinterpret_cast<f32x4 *>(dskip + ((size_t)b * V + v) * s.C1 + c4) = o;
        }
    }
    if (dlow && s.low) {
        const int D2 = s.D >> 1, H2 = s.H >> 1, W2 = s.W >> 1;
        const size_t V2 = (size_t)D2 * H2 * W2, total = V2 * (s.C2 / 4);
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
            const int c4 = (int)(e % (s.C2 / 4)) * 4;
            size_t v2 = e / (s.C2 / 4);
            const int x2 = (int)(v2 % W2); size_t r = v2 / W2;
            const int y2 = (int)(r % H2), z2 = (int)(r / H2);
            const f32x4 x = *reinterpret_cast<const f32x4 *>(s.low + ((size_t)b * V2 + v2) * s.C2 + c4);
            const float *cf = coef + ((size_t)b * C + s.C1 + c4) * 3;
            f32x4 dsum = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < 8; ++k) {
                const size_t v = ((size_t)(2 * z2 + (k >> 2)) * s.H + (2 * y2 + ((k >> 1) & 1))) * s.W + 2 * x2 + (k & 1);
                const f32x4 d = *reinterpret_cast<const f32x4 *>(dxn + ((size_t)b * V + v) * C + s.C1 + c4);
                dsum.x += d.x; dsum.y += d.y; dsum.z += d.z; dsum.w += d.w;
            }
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaf(cf[3 * k], dsum[k], 8.0f * fmaf(cf[3 * k + 1], x[k], cf[3 * k + 2]));
            if (mask & 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[k] = x[k] > 0.f ? o[k] : 0.f; ul = max(ul, bits(o[k])); }
            }
            *reinterpret_cast<f32x4 *>(dlow + ((size_t)b * V2 + v2) * s.C2 + c4) = o;
        }
    }
    if (!mask) return;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)us, o), u = (unsigned)__shfl_xor((int)ul, o);
        us = t > us ? t : us; ul = u > ul ? u : ul;
    }
    if ((threadIdx.x & 63) == 0) { wmax[0][threadIdx.x >> 6] = us; wmax[1][threadIdx.x >> 6] = ul; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned *w = wmax[threadIdx.x];
        const unsigned a0 = w[0] > w[1] ? w[0] : w[1], a1 = w[2] > w[3] ? w[2] : w[3];
        unsigned *dst = threadIdx.x ? amax_low : amax_skip;
        if (dst && ((mask >> threadIdx.x) & 1)) atomicMax(dst, a0 > a1 ? a0 : a1);
    }
}

// dx = dy routed to the first maximum of each 2x2x2 window (scan order z,y,x: ATen's max_pool3d)
__global__ void __launch_bounds__(256)
maxpool3d_cl_bwd_kernel(const float *x, const float *dy, float *dx, int D, int H, int W, int C, size_t total) {
    const int D2 = D / 2, H2 = H / 2, W2 = W / 2;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        size_t v = e / C;
        const int ox = (int)(v % W2); v /= W2;
        const int oy = (int)(v % H2); v /= H2;
        const int oz = (int)(v % D2);
        const size_t b = v / D2;
        float m = -INFINITY;
        int best = 0;
        size_t idx[8];
        for (int k = 0; k < 8; ++k) {
            const int z = 2 * oz + (k >> 2), y = 2 * oy + ((k >> 1) & 1), xx = 2 * ox + (k & 1);
            idx[k] = ((((size_t)b * D + z) * H + y) * W + xx) * C + c;
            const float t = x[idx[k]];
            if (t > m) { m = t; best = k; }
        }
        const float gval = dy[e];
        for (int k = 0; k < 8; ++k) dx[idx[k]] = (k == best) ? gval : 0.0f;
    }
}

// An encoder level's output y (a ReLU output) feeds the 2x2x2 max-pool AND the decoder's skip: its gradient is
// d_skip + maxpool_backward(d_pooled), and the layer that produced y masks it by (y > 0).  One pass for all three (instead of
// maxpool3d_cl_bwd_kernel + the framework's add + relu_mask_kernel: 2.1 GB of traffic per 64^3 level of eight scenes, 0.84 GB here):
// thread = (window, four channels) reads the window's eight y and d_skip rows, routes d_pooled to the FIRST maximum in scan order
// (z, y, x: ATen's max_pool3d), writes g = (y > 0 ? d_skip + routed : 0) and keeps max |g| (absmax as relu_mask_kernel).
__global__ void __launch_bounds__(256)
pool_fork_bwd_kernel(const f32x4 *y, const f32x4 *dskip, const f32x4 *dpool, f32x4 *g, int D, int H, int W, int C4, size_t total, unsigned *absmax) {
    const int D2 = D / 2, H2 = H / 2, W2 = W / 2;
    unsigned u = 0;
    auto bits = [](float x) { return __builtin_bit_cast(unsigned, x) & 0x7fffffffu; };
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C4);
        size_t v = e / C4;
        const int ox = (int)(v % W2); v /= W2;
        const int oy = (int)(v % H2); v /= H2;
        const int oz = (int)(v % D2);
        const size_t b = v / D2;
        size_t idx[8];
        f32x4 yv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int z = 2 * oz + (k >> 2), yy = 2 * oy + ((k >> 1) & 1), xx = 2 * ox + (k & 1);
            idx[k] = ((((size_t)b * D + z) * H + yy) * W + xx) * C4 + c;
            yv[k] = y[idx[k]];
        }
        const f32x4 dp = dpool[e];
        int best[4] = {0, 0, 0, 0};
        float m[4] = {yv[0].x, yv[0].y, yv[0].z, yv[0].w};
#pragma unroll
        for (int k = 1; k < 8; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (yv[k][q] > m[q]) { m[q] = yv[k][q]; best[q] = k; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 ds = dskip[idx[k]];
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = ds[q] + (best[q] == k ? dp[q] : 0.0f);
                o[q] = yv[k][q] > 0.0f ? t : 0.0f;
                u = max(u, bits(o[q]));
            }
            g[idx[k]] = o;
        }
    }
    if (!absmax) return;
    __shared__ unsigned wmax[4];
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)u, o); u = t > u ? t : u; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a0 = wmax[0] > wmax[1] ? wmax[0] : wmax[1], a1 = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
        atomicMax(absmax, a0 > a1 ? a0 : a1);
    }
}

}  // namespace

extern "C" {

int vt_relu_mask(const float *dy, const float *y, float *g, int64_t n, void *stream) {
    if (!dy || !y || !g || n <= 0 || (n & 3)) return vt_fail(VT_ERR_INVALID, "vt_relu_mask: bad argument (n must be a positive multiple of 4)");
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4 *>(dy), reinterpret_cast<const f32x4 *>(y), reinterpret_cast<f32x4 *>(g), (size_t)n / 4,
                       (unsigned *)nullptr);
    return vt_check(hipGetLastError(), "vt_relu_mask");
}

int vt_relu_mask_absmax(const float *dy, const float *y, float *g, int64_t n, float *absmax, void *stream) {
    if (!dy || !y || !g || !absmax || n <= 0 || (n & 3)) return vt_fail(VT_ERR_INVALID, "vt_relu_mask_absmax: bad argument (n must be a positive multiple of 4)");
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;                    // one atomicMax per workgroup on ONE address: 8192 of them cost 35 us per call
    const hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return vt_check(e, "vt_relu_mask_absmax: hipMemsetAsync");
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4 *>(dy), reinterpret_cast<const f32x4 *>(y), reinterpret_cast<f32x4 *>(g), (size_t)n / 4,
                       reinterpret_cast<unsigned *>(absmax));
    return vt_check(hipGetLastError(), "vt_relu_mask_absmax");
}

size_t vt_conv1x1_bwd_workspace_bytes(void) { return (size_t)F1_WGS * F1_PART * sizeof(float); }

int vt_conv1x1_bwd_masked(const float *dout, const float *y, const float *w, int64_t n, float *g, float *absmax, float *dw, float *db,
                          void *workspace, size_t workspace_bytes, void *stream) {
    if (!dout || !y || !w || !g || !workspace || n <= 0) return vt_fail(VT_ERR_INVALID, "vt_conv1x1_bwd_masked: bad argument");
    if (n >= ((int64_t)1 << 31) - 64) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv1x1_bwd_masked: more than 2^31 rows");
    if (workspace_bytes < vt_conv1x1_bwd_workspace_bytes()) return vt_fail(VT_ERR_WORKSPACE, "vt_conv1x1_bwd_masked: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (absmax) {
        const hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float), st);
        if (e != hipSuccess) return vt_check(e, "vt_conv1x1_bwd_masked: hipMemsetAsync");
    }
    const long long ntile = ((long long)n + 31) / 32;
    int wgs = (int)((ntile + F1_WAVES - 1) / F1_WAVES);
    if (wgs > F1_WGS) wgs = F1_WGS;
    hipLaunchKernelGGL(conv1x1_bwd_kernel, dim3((unsigned)wgs), dim3(F1_WAVES * 64), 0, st, dout, y, w, (unsigned)n, g,
                       reinterpret_cast<unsigned *>(absmax), (float *)workspace);
    if (dw || db)
        hipLaunchKernelGGL(conv1x1_bwd_reduce_kernel, dim3(F1_PART / 32), dim3(256), 0, st, (const float *)workspace, wgs, dw, db);
    return vt_check(hipGetLastError(), "vt_conv1x1_bwd_masked");
}

static int wgrad_chunks(int B, int D, int H, int W, int pairs) {
    const int ntiles = B * ((W + WG_TX - 1) / WG_TX) * ((H + WG_TY - 1) / WG_TY) * ((D + WG_TZ - 1) / WG_TZ);
    int chunks = 512 / pairs;
    if (chunks < 1) chunks = 1;
    if (chunks > ntiles) chunks = ntiles;
    return chunks;
}

size_t vt_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 31) || Cout <= 0 || (Cout & 31)) return 0;
    const int pairs = (Cin / 32) * (Cout / 32);
    return (size_t)wgrad_chunks(B, D, H, W, pairs) * pairs * 27 * 1024 * sizeof(float);
}

int vt_conv3d_wgrad(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                    const float *scale_shift, const float *g, int Cout, void *workspace, size_t workspace_bytes,
                    float *dw, void *stream) {
    WgradArgs a;
    a.c.s = Src{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(a.c.s, B) || !g || !workspace || !dw) return vt_fail(VT_ERR_INVALID, "vt_conv3d_wgrad: bad argument");
    if (Cout <= 0 || (Cout & 31)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_wgrad: Cout must be a multiple of 32");
    const int Cin = a.c.s.C1 + a.c.s.C2;
    if (workspace_bytes < vt_conv3d_wgrad_workspace_bytes(B, D, H, W, Cin, Cout)) return vt_fail(VT_ERR_WORKSPACE, "vt_conv3d_wgrad: workspace too small");
    a.c.scale_shift = scale_shift; a.c.wp = nullptr; a.c.out = nullptr; a.c.part = nullptr; a.c.Cout = Cout; a.c.relu = 0;
    a.c.TX = WG_TX; a.c.TY = WG_TY; a.c.TZ = WG_TZ;
    a.c.tiles_x = (W + WG_TX - 1) / WG_TX; a.c.tiles_y = (H + WG_TY - 1) / WG_TY; a.c.tiles_z = (D + WG_TZ - 1) / WG_TZ;
    a.g = g; a.partial = (float *)workspace;
    a.ntiles = B * a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;
    const int pairs = (Cin / 32) * (Cout / 32), chunks = wgrad_chunks(B, D, H, W, pairs);
    const size_t lds = (size_t)(WG_HALO + WG_VOX) * CPAD * sizeof(float);
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        const hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_wgrad_kernel), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_wgrad: hipFuncSetAttribute");
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv3d_wgrad_kernel, dim3((unsigned)chunks, (unsigned)pairs), dim3(WG_WAVES * 64), lds, st, a);
    const size_t total = (size_t)Cout * Cin * 27;
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float *)workspace, chunks, Cout, Cin, dw);
    return vt_check(hipGetLastError(), "vt_conv3d_wgrad");
}

static int wgrad_h_chunks(int B, int D, int H, int W, int pairs) {
    const int ntiles = B * (W / 8) * (H / 8) * (D / WH_TZ);
    int chunks = 512 / pairs;                                      // two workgroups per CU
    if (chunks < 1) chunks = 1;
    if (chunks > ntiles) chunks = ntiles;
    return chunks;
}

size_t vt_conv3d_wgrad_f16x3_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || (D % WH_TZ) || (H & 7) || (W & 7) || Cin <= 0 || (Cin & 31) || Cout <= 0 || (Cout & 31)) return 0;
    if ((size_t)B * D * H * W * (Cin > Cout ? Cin : Cout) >= ((size_t)1 << 32)) return 0;        // 32-bit element offsets
    const int pairs = (Cin / 32) * (Cout / 32);
    return (size_t)wgrad_h_chunks(B, D, H, W, pairs) * pairs * 27 * 1024 * sizeof(float);
}

int vt_conv3d_wgrad_f16x3(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                          const float *scale_shift, const float *g, int Cout, const float *g_absmax,
                          void *workspace, size_t workspace_bytes, float *dw, void *stream) {
    WgradHArgs ha;
    WgradArgs &a = ha.w;
    a.c.s = Src{skip, low, C1, low ? C2 : 0, D, H, W};
    if (!src_ok(a.c.s, B) || !g || !workspace || !dw) return vt_fail(VT_ERR_INVALID, "vt_conv3d_wgrad_f16x3: bad argument");
    const int Cin = a.c.s.C1 + a.c.s.C2;
    const size_t need = vt_conv3d_wgrad_f16x3_workspace_bytes(B, D, H, W, Cin, Cout);
    if (!need) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_wgrad_f16x3: shape not covered (sides in multiples of 8, channels of 32); use vt_conv3d_wgrad");
    if (workspace_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_conv3d_wgrad_f16x3: workspace too small");
    a.c.scale_shift = scale_shift; a.c.wp = nullptr; a.c.out = nullptr; a.c.part = nullptr; a.c.Cout = Cout; a.c.relu = 0;
    a.c.TX = 8; a.c.TY = 8; a.c.TZ = WH_TZ;
    a.c.tiles_x = W / 8; a.c.tiles_y = H / 8; a.c.tiles_z = D / WH_TZ;
    a.g = g; a.partial = (float *)workspace;
    a.ntiles = B * a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;
    ha.g_absmax = g_absmax;
    ha.list = nullptr; ha.ss_cin = 0;
    const int pairs = (Cin / 32) * (Cout / 32), chunks = wgrad_h_chunks(B, D, H, W, pairs);
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        const hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_wgrad_h_kernel<false>), (int)WH_LDS);
        if (e != hipSuccess) return vt_check(e, "vt_conv3d_wgrad_f16x3: hipFuncSetAttribute");
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv3d_wgrad_h_kernel<false>, dim3((unsigned)chunks, (unsigned)pairs), dim3(WH_THREADS), WH_LDS, st, ha);
    const size_t total = (size_t)Cout * Cin * 27;
    hipLaunchKernelGGL(conv3d_wgrad_reduce_scaled_kernel, dim3((unsigned)(total / 64 < 4096 ? total / 64 : 4096)), dim3(256), 0, st,
                       (const float *)workspace, chunks, Cout, Cin, g_absmax, dw);
    return vt_check(hipGetLastError(), "vt_conv3d_wgrad_f16x3");
}

// the weight gradient of a decoder-entry layer [skip | upsample(low)]: the skip channels on conv3d_wgrad_h_kernel, the upsampled ones
// per output parity class on conv3d_wgrad_hp_kernel (workspace = [skip partials][parity partials])
static int wgrad_hp_chunks(int B, int D, int H, int W, int pairs_l) {
    const int ntiles = B * (W / 16) * (H / 16) * (D / (2 * WH_TZ));
    int chunks = 512 / (pairs_l * 4);                              // a workgroup takes two of the eight classes
    if (chunks < 1) chunks = 1;
    if (chunks > ntiles) chunks = ntiles;
    return chunks;
}

static size_t wgrad_up_layout(int B, int D, int H, int W, int C1, int C2, int Cout, size_t *par_off) {
    if (C1 <= 0 || C2 <= 0 || (C1 & 31) || (C2 & 31) || Cout <= 0 || (Cout & 31) || (D % (2 * WH_TZ)) || (H & 15) || (W & 15)) return 0;
    const size_t dense = vt_conv3d_wgrad_f16x3_workspace_bytes(B, D, H, W, C1, Cout);
    if (!dense || (size_t)B * D * H * W * (C1 + C2 > Cout ? C1 + C2 : Cout) >= ((size_t)1 << 32)) return 0;
    const int pairs_l = (C2 / 32) * (Cout / 32);
    const size_t off = (dense + 255) / 256 * 256;
    if (par_off) *par_off = off;
    return off + (size_t)wgrad_hp_chunks(B, D, H, W, pairs_l) * 8 * pairs_l * 8 * 1024 * sizeof(float);
}

size_t vt_conv3d_wgrad_f16x3_up_workspace_bytes(int B, int D, int H, int W, int C1, int C2, int Cout) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return wgrad_up_layout(B, D, H, W, C1, C2, Cout, nullptr);
}

int vt_conv3d_wgrad_f16x3_up(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                             const float *scale_shift, const float *g, int Cout, const float *g_absmax,
                             void *workspace, size_t workspace_bytes, float *dw, void *stream) {
    WgradHArgs ha;
    WgradArgs &a = ha.w;
    a.c.s = Src{skip, low, C1, C2, D, H, W};
    if (!low || !src_ok(a.c.s, B) || !g || !workspace || !dw || !scale_shift) return vt_fail(VT_ERR_INVALID, "vt_conv3d_wgrad_f16x3_up: bad argument");
    size_t par_off;
    const size_t need = wgrad_up_layout(B, D, H, W, C1, C2, Cout, &par_off);
    if (!need) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_wgrad_f16x3_up: shape not covered (sides in multiples of 16, channels of 32); use vt_conv3d_wgrad_f16x3");
    if (workspace_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_conv3d_wgrad_f16x3_up: workspace too small");
    hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_wgrad_h_kernel<false>), (int)WH_LDS);
    if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_wgrad_hp_kernel), (int)WH_LDS);
    if (e != hipSuccess) return vt_check(e, "vt_conv3d_wgrad_f16x3_up: hipFuncSetAttribute");
    hipStream_t st = (hipStream_t)stream;
    a.c.scale_shift = scale_shift; a.c.wp = nullptr; a.c.out = nullptr; a.c.part = nullptr; a.c.Cout = Cout; a.c.relu = 0;
    a.c.TX = 8; a.c.TY = 8; a.c.TZ = WH_TZ;
    a.g = g;
    ha.g_absmax = g_absmax; ha.list = nullptr;
    {   // the skip channels: the dense kernel over [skip] alone, its scale / shift rows taken from the wider table
        a.c.s = Src{skip, nullptr, C1, 0, D, H, W};
        a.c.tiles_x = W / 8; a.c.tiles_y = H / 8; a.c.tiles_z = D / WH_TZ;
        a.ntiles = B * a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;
        a.partial = (float *)workspace;
        ha.ss_cin = C1 + C2;
        const int pairs = (C1 / 32) * (Cout / 32), chunks = wgrad_h_chunks(B, D, H, W, pairs);
        hipLaunchKernelGGL(conv3d_wgrad_h_kernel<false>, dim3((unsigned)chunks, (unsigned)pairs), dim3(WH_THREADS), WH_LDS, st, ha);
        const size_t total = (size_t)Cout * C1 * 27;
        hipLaunchKernelGGL(conv3d_wgrad_reduce_scaled_kernel, dim3((unsigned)(total / 64 < 4096 ? total / 64 : 4096)), dim3(256), 0, st,
                           (const float *)workspace, chunks, Cout, C1, g_absmax, dw, (const float *)nullptr, (const float *)nullptr, 0, C1 + C2);
    }
    {   // the upsampled channels, per parity class on the coarse grid
        a.c.s = Src{skip, low, C1, C2, D, H, W};
        a.c.tiles_x = W / 16; a.c.tiles_y = H / 16; a.c.tiles_z = D / (2 * WH_TZ);
        a.ntiles = B * a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;
        a.partial = reinterpret_cast<float *>((char *)workspace + par_off);
        ha.ss_cin = 0;
        const int pairs_l = (C2 / 32) * (Cout / 32), chunks = wgrad_hp_chunks(B, D, H, W, pairs_l);
        hipLaunchKernelGGL(conv3d_wgrad_hp_kernel, dim3((unsigned)chunks, (unsigned)pairs_l, 4u), dim3(WH_THREADS), WH_LDS, st, ha);
        const size_t total = (size_t)Cout * C2 * 27;
        hipLaunchKernelGGL(conv3d_wgrad_reduce_parity_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, st,
                           (const float *)a.partial, chunks, Cout, C1, C2, g_absmax, dw);
    }
    return vt_check(hipGetLastError(), "vt_conv3d_wgrad_f16x3_up");
}

// the same weight gradient for an input that is exactly zero over most of the volume (see wgrad_tile_list_kernel): workspace =
// [chunk partials][tile list][plane sums][box sums]
static size_t wgrad_sparse_layout(int B, int D, int H, int W, int Cin, int Cout, size_t *list_off, size_t *p_off, size_t *s_off) {
    const size_t part = vt_conv3d_wgrad_f16x3_workspace_bytes(B, D, H, W, Cin, Cout);
    if (!part || (D & 7)) return 0;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t off = up(part);
    if (list_off) *list_off = off;
    off += up((size_t)(1 + B * (W / 8) * (H / 8) * (D / WH_TZ)) * sizeof(int));
    if (p_off) *p_off = off;
    off += up((size_t)B * D * 9 * Cout * sizeof(float));
    if (s_off) *s_off = off;
    off += up((size_t)B * 27 * Cout * sizeof(float));
    return off;
}

size_t vt_conv3d_wgrad_f16x3_sparse_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout) {
    return wgrad_sparse_layout(B, D, H, W, Cin, Cout, nullptr, nullptr, nullptr);
}

int vt_conv3d_wgrad_f16x3_sparse(const float *x, int C, int B, int D, int H, int W, const float *scale_shift,
                                 const unsigned char *tile_flags, const float *g, int Cout, const float *g_absmax,
                                 void *workspace, size_t workspace_bytes, float *dw, void *stream) {
    WgradHArgs ha;
    WgradArgs &a = ha.w;
    a.c.s = Src{x, nullptr, C, 0, D, H, W};
    if (!src_ok(a.c.s, B) || !g || !workspace || !dw || !scale_shift || !tile_flags)
        return vt_fail(VT_ERR_INVALID, "vt_conv3d_wgrad_f16x3_sparse: bad argument");
    size_t list_off, p_off, s_off;
    const size_t need = wgrad_sparse_layout(B, D, H, W, C, Cout, &list_off, &p_off, &s_off);
    if (!need) return vt_fail(VT_ERR_UNSUPPORTED, "vt_conv3d_wgrad_f16x3_sparse: shape not covered (sides in multiples of 8, channels of 32)");
    if (workspace_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_conv3d_wgrad_f16x3_sparse: workspace too small");
    a.c.scale_shift = scale_shift; a.c.wp = nullptr; a.c.out = nullptr; a.c.part = nullptr; a.c.Cout = Cout; a.c.relu = 0;
    a.c.TX = 8; a.c.TY = 8; a.c.TZ = WH_TZ;
    a.c.tiles_x = W / 8; a.c.tiles_y = H / 8; a.c.tiles_z = D / WH_TZ;
    a.g = g; a.partial = (float *)workspace;
    a.ntiles = B * a.c.tiles_x * a.c.tiles_y * a.c.tiles_z;
    ha.g_absmax = g_absmax;
    int *list = reinterpret_cast<int *>((char *)workspace + list_off);
    float *P = reinterpret_cast<float *>((char *)workspace + p_off), *S = reinterpret_cast<float *>((char *)workspace + s_off);
    ha.list = list; ha.ss_cin = 0;
    const int pairs = (C / 32) * (Cout / 32), chunks = wgrad_h_chunks(B, D, H, W, pairs);
    const hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&conv3d_wgrad_h_kernel<true>), (int)WH_LDS);
    if (e != hipSuccess) return vt_check(e, "vt_conv3d_wgrad_f16x3_sparse: hipFuncSetAttribute");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_tile_list_kernel, dim3(1), dim3(1024), 0, st, tile_flags, a.ntiles, a.c.tiles_x, a.c.tiles_y, a.c.tiles_z, list);
    hipLaunchKernelGGL(wgrad_gsum_plane_kernel, dim3((unsigned)D, (unsigned)B), dim3(256), 0, st, g, D, H, W, Cout, P);
    hipLaunchKernelGGL(wgrad_gsum_final_kernel, dim3((unsigned)((27 * Cout + 255) / 256), (unsigned)B), dim3(256), 0, st, (const float *)P, D, Cout, S);
    hipLaunchKernelGGL(conv3d_wgrad_h_kernel<true>, dim3((unsigned)chunks, (unsigned)pairs), dim3(WH_THREADS), WH_LDS, st, ha);
    const size_t total = (size_t)Cout * C * 27;
    hipLaunchKernelGGL(conv3d_wgrad_reduce_scaled_kernel, dim3((unsigned)(total / 64 < 4096 ? total / 64 : 4096)), dim3(256), 0, st,
                       (const float *)workspace, chunks, Cout, C, g_absmax, dw, (const float *)S, scale_shift, B);
    return vt_check(hipGetLastError(), "vt_conv3d_wgrad_f16x3_sparse");
}

static int gn_bwd_impl(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                       const float *part1, int nblk1, const float *part2, int nblk2,
                       const float *dxn, int groups, const float *gamma, double eps,
                       float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow,
                       int mask_flags, float *absmax_skip, float *absmax_low, void *stream, bool bpart_ready, float *dgb_sum = nullptr);

int vt_gn_bwd_masked(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                     const float *part1, int nblk1, const float *part2, int nblk2,
                     const float *dxn, int groups, const float *gamma, double eps,
                     float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow,
                     int mask_flags, float *absmax_skip, float *absmax_low, float *dgb_sum, void *stream) {
    return gn_bwd_impl(skip, C1, low, C2, B, D, H, W, part1, nblk1, part2, nblk2, dxn, groups, gamma, eps, bpart, nblkb, coef, dgb, dskip, dlow,
                       mask_flags, absmax_skip, absmax_low, stream, false, dgb_sum);
}

int vt_gn_bwd_from_part(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                        const float *part1, int nblk1, const float *part2, int nblk2,
                        const float *dxn, int groups, const float *gamma, double eps,
                        const float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow,
                        int mask_flags, float *absmax_skip, float *absmax_low, float *dgb_sum, void *stream) {
    return gn_bwd_impl(skip, C1, low, C2, B, D, H, W, part1, nblk1, part2, nblk2, dxn, groups, gamma, eps, const_cast<float *>(bpart), nblkb, coef, dgb,
                       dskip, dlow, mask_flags, absmax_skip, absmax_low, stream, true, dgb_sum);
}

static int gn_bwd_impl(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                       const float *part1, int nblk1, const float *part2, int nblk2,
                       const float *dxn, int groups, const float *gamma, double eps,
                       float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow,
                       int mask_flags, float *absmax_skip, float *absmax_low, void *stream, bool bpart_ready, float *dgb_sum) {
    Src s{skip, low, C1, low ? C2 : 0, D, H, W};
    if (dgb_sum && !dskip && !dlow) return vt_fail(VT_ERR_INVALID, "vt_gn_bwd: the summed (dgamma, dbeta) come out of the pass that writes dskip / dlow");
    if (!src_ok(s, B) || !part1 || nblk1 <= 0 || !dxn || !gamma || !bpart || nblkb <= 0 || !coef || !dgb)
        return vt_fail(VT_ERR_INVALID, "vt_gn_bwd: bad argument");
    if (low && (!part2 || nblk2 <= 0)) return vt_fail(VT_ERR_INVALID, "vt_gn_bwd: statistics of `low` missing");
    if ((mask_flags & ~3) || ((mask_flags & 1) && (!dskip || !absmax_skip)) || ((mask_flags & 2) && (!dlow || !low || !absmax_low)))
        return vt_fail(VT_ERR_INVALID, "vt_gn_bwd_masked: a masked gradient needs its output and its absmax cell");
    const int C = s.C1 + s.C2;
    if (groups <= 0 || C % groups || C / groups > 256) return vt_fail(VT_ERR_INVALID, "vt_gn_bwd: bad group count");
    hipStream_t st = (hipStream_t)stream;
    // (bpart_ready: the data-gradient conv left the two sums per workgroup -- vt_conv3d_gcr_f16x3_xstats)
    if (!bpart_ready) hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(nblkb, B), dim3(256), 0, st, s, dxn, nblkb, bpart);
    StatSrc s1{part1, nblk1, s.C1}, s2{low ? part2 : nullptr, low ? nblk2 : 0, s.C2};
    hipLaunchKernelGGL(gn_bwd_coeffs_kernel, dim3(B, groups), dim3(256), 0, st, s1, s2, (const float *)bpart, nblkb, groups,
                       (double)D * H * W, gamma, (float)eps, coef, dgb);
    if (dskip || dlow) {
        const size_t V = (size_t)D * H * W;
        size_t blocks = (V * (size_t)(s.C1 / 4) + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (mask_flags) {
            // (one atomicMax per workgroup on one address: a thousand of them at most, as vt_relu_mask_absmax)
            const size_t cap = (size_t)(1024 / B > 1 ? 1024 / B : 1);
            if (blocks > cap) blocks = cap;
            hipError_t e = hipSuccess;
            if (mask_flags & 1) e = hipMemsetAsync(absmax_skip, 0, sizeof(float), st);
            if (e == hipSuccess && (mask_flags & 2)) e = hipMemsetAsync(absmax_low, 0, sizeof(float), st);
            if (e != hipSuccess) return vt_check(e, "vt_gn_bwd_masked: hipMemsetAsync");
        }
        hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)blocks, B), dim3(256), 0, st, s, dxn, (const float *)coef, dskip, dlow,
                           mask_flags, reinterpret_cast<unsigned *>(absmax_skip), reinterpret_cast<unsigned *>(absmax_low), (const float *)dgb, dgb_sum, B);
    }
    return vt_check(hipGetLastError(), "vt_gn_bwd");
}

int vt_gn_bwd(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
              const float *part1, int nblk1, const float *part2, int nblk2,
              const float *dxn, int groups, const float *gamma, double eps,
              float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow, void *stream) {
    return vt_gn_bwd_masked(skip, C1, low, C2, B, D, H, W, part1, nblk1, part2, nblk2, dxn, groups, gamma, eps, bpart, nblkb, coef, dgb,
                            dskip, dlow, 0, nullptr, nullptr, nullptr, stream);
}

int vt_maxpool3d_cl_bwd_fork(const float *y, const float *dskip, const float *dpooled, int B, int D, int H, int W, int C, float *g,
                             float *absmax, void *stream) {
    if (!y || !dskip || !dpooled || !g || B <= 0 || C <= 0 || (C & 3) || D < 2 || H < 2 || W < 2 || ((D | H | W) & 1))
        return vt_fail(VT_ERR_INVALID, "vt_maxpool3d_cl_bwd_fork: bad argument (even extents, channels in multiples of 4)");
    const size_t total = (size_t)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;                    // (one atomicMax per workgroup on one address, as vt_relu_mask_absmax)
    if (absmax) {
        const hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return vt_check(e, "vt_maxpool3d_cl_bwd_fork: hipMemsetAsync");
    }
    hipLaunchKernelGGL(pool_fork_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4 *>(y),
                       reinterpret_cast<const f32x4 *>(dskip), reinterpret_cast<const f32x4 *>(dpooled), reinterpret_cast<f32x4 *>(g), D, H, W, C / 4,
                       total, reinterpret_cast<unsigned *>(absmax));
    return vt_check(hipGetLastError(), "vt_maxpool3d_cl_bwd_fork");
}

int vt_maxpool3d_cl_bwd(const float *x, const float *dy, int B, int D, int H, int W, int C, float *dx, void *stream) {
    if (!x || !dy || !dx || B <= 0 || C <= 0 || D < 2 || H < 2 || W < 2 || ((D | H | W) & 1))
        return vt_fail(VT_ERR_INVALID, "vt_maxpool3d_cl_bwd: bad argument (even extents)");
    const size_t total = (size_t)B * (D / 2) * (H / 2) * (W / 2) * C;
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(maxpool3d_cl_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, dy, dx, D, H, W, C, total);
    return vt_check(hipGetLastError(), "vt_maxpool3d_cl_bwd");
}

}  // extern "C"
