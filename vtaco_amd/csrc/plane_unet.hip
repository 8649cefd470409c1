// plane_unet.hip -- the hand branch's 2-D U-Net (reference src/encoder/unet.py:52-233, built by LocalPoolPointnet with
// ``unet: True``, src/encoder/pointnet.py:49-50, 85-100) on hand-written kernels: forward (vt_plane_unet_fwd) and backward
// (vt_plane_unet_bwd).
//
// The net is small (three 32^2 planes per scene, depth 4, 32 filters: 0.92 GFLOP, 1.93 M parameters) and DEEP: 18 dependent
// steps -- 2 convs per encoder level, a transposed conv + 2 convs per decoder level, the 1x1 head.  Through the framework it is
// 33 launches of ~25 us each (MIOpen's solvers are built for images, not for 16 pixels x 256 channels:
// profiles/r05_bench_extra.jsonl); here every step is one launch of ONE kernel template, a PHASE:
//
//   D[co][px] += W[co][ci][tap] * X[px + tap][ci]       implicit GEMM on the f32 matrix core (v_mfma_f32_32x32x2f32: the
//                                                       reference's arithmetic, no operand splitting);
//                                                       A = weights (32 output channels), B = activations (32 pixels)
//
// A workgroup owns one output tile (32 pixels x 32 output channels) and splits K = taps x Cin over its KS = 4 or 8 waves by
// input channel (KS grows with Cin, so a wave issues 36-72 MFMAs whatever the layer: the phases are latency-, not
// throughput-bound).  Each wave stages its 8-channel chunks of the tile's halo in a private LDS region ([pixel][9 floats]: the B
// operand is one ds_read_b32 per MFMA, conflict-free across the 32 pixels), streams its weights from L2 in fragment order (one
// 16-byte load per lane, tap and chunk), and the waves' accumulators meet in LDS in a fixed order (bit-reproducible).  The
// epilogue adds the bias, applies the ReLU and writes channels-last activations; the 2x2 max-pool and the skip concat cost
// nothing of their own: they are how the NEXT phase's loader reads (max over the four source pixels; the channels of two
// sources side by side).  A device-side barrier between phases inside one persistent launch was built first and measured: 4.6 us
// per barrier with 256 workgroups even with relaxed atomics, 21-45 us with the agent-scope fences the hand-off needs
// (tools/probe/barrier_probe.hip) against 1.45 us for a kernel boundary -- so the phases are launches.
//
// The training forward is the same launch sequence with the activations kept (the workspace) for vt_plane_unet_bwd.
#include <algorithm>
#include "vt_common.h"
#include "decode_common.h"

namespace {

constexpr int PU_CH = 8;            // input channels per staged chunk (4 k-steps of the 32x32x2 MFMA per tap)
constexpr int PU_PITCH = 9;         // LDS floats per staged pixel (odd: the 32 pixels of a B operand hit 32 banks)
constexpr int PU_MAX_HALO = 104;    // staged pixels per wave: NI * (TH + 2) * (TW + 2) <= 102 for every tile shape below
constexpr int PU_MAX_DEPTH = 5;
constexpr int PU_MAX_PHASES = 5 * PU_MAX_DEPTH - 2;

struct PuDims {
    int depth, in_ch, start, classes;   // UNet(num_classes, in_channels, depth, start_filts)
    int n_img, H, W;                    // images (planes x scenes), each H x W
};

struct PuSrc {
    const float *ptr;        // [n_img * H * W][C] channels-last activations or, nchw, [n_img][C][H * W]
    int C, pool, nchw;
    int W, H;                // the source's own image size (twice the phase's when pool is set)
};

struct PuPhase {
    PuSrc a, b;              // input channels = a.C + b.C (b.C = 0: one source)
    const float *wfrag;      // [n_cb][Cin / 8][ntaps][64 lanes][4]: lane l, slot j = W[cb * 32 + l % 32][chunk * 8 + 2 j + l / 32][tap]
    const float *bias;       // [Cout]
    float *out;              // mode 0: [n_img * H * W][Cout]; mode 1 (transposed conv): [n_img * 2H * 2W][Cout];
                             // mode 2 (head): [n_img][Cout][H * W]
    int Cin, Cout, n_cb, ntaps, W, H, mode, relu, KS, n_img;
    int TW, TH, NI, tiles_x, tiles_y, n_tiles;
};

// ---- backward: the gradient of a phase's output is ASSEMBLED by the loader of that phase's data-gradient launch from the input
// gradients its consumers have already written (the backward runs the phases in reverse): the sum of up to two consumers (the skip
// concat and the pooled path of an encoder level), the pool's routing (the gradient goes to the first maximum of the 2x2 window,
// as torch's max_pool2d backward does) and the ReLU mask, both read off the forward's activations.  No pass of its own for any of them.
struct PuGradSrc {
    const float *ptr;        // a consumer's input gradient [n_img * H * W][C] (its whole concatenated input), or dOut (nchw)
    int C, off, pooled, nchw;   // row stride, first channel of this tensor's share; pooled: the consumer read max-pooled input
    int W, H;                // the gradient's own image size (half the phase's when pooled)
};
struct PuBwd {
    PuGradSrc g0, g1;        // g1.ptr = null: one consumer
    const float *act;        // the forward output of this phase [n_img * H * W][Cz] (ReLU mask, argmax of the pool), or null
    float *dz;               // side output: the assembled gradient [n_img * H * W][Cz] for the weight-gradient launch, or null
    int Cz;                  // channels of this phase's output
    int unshuffle;           // transposed conv: the GEMM's input pixel (y, x), channel par * Cz + co is dZ[(2y + dy, 2x + dx)][co]
};

__host__ __device__ inline int pu_width(const PuDims &d, int level) { return d.start << level; }
__host__ __device__ inline int pu_n_phases(const PuDims &d) { return 5 * d.depth - 2; }

// what phase i reads and writes (sizes only)
struct PuShape {
    int Cin, Cout, ntaps, mode, level_in, level_out, relu;   // level_out: the level whose resolution the OUTPUT has
};
__host__ __device__ inline PuShape pu_shape(const PuDims &d, int i) {
    PuShape s;
    const int D = d.depth;
    s.relu = 1;
    if (i < 2 * D) {                                       // encoder level l: conv a (i even), conv b
        const int l = i >> 1;
        s.Cout = pu_width(d, l);
        s.Cin = (i & 1) ? s.Cout : (l == 0 ? d.in_ch : pu_width(d, l - 1));
        s.ntaps = 9; s.mode = 0; s.level_in = s.level_out = l;
    } else if (i < 5 * D - 3) {
        const int u = (i - 2 * D) / 3, k = (i - 2 * D) % 3, lev = D - 2 - u;
        s.Cout = pu_width(d, lev);
        if (k == 0) { s.Cin = pu_width(d, lev + 1); s.ntaps = 1; s.mode = 1; s.level_in = lev + 1; s.level_out = lev; s.relu = 0; }
        else { s.Cin = k == 1 ? 2 * s.Cout : s.Cout; s.ntaps = 9; s.mode = 0; s.level_in = s.level_out = lev; }
    } else {
        s.Cin = pu_width(d, 0); s.Cout = d.classes; s.ntaps = 1; s.mode = 2; s.level_in = s.level_out = 0; s.relu = 0;
    }
    return s;
}
__host__ __device__ inline long long pu_out_floats(const PuDims &d, const PuShape &s) {
    return (long long)d.n_img * (d.H >> s.level_out) * (d.W >> s.level_out) * s.Cout;
}
__host__ __device__ inline long long pu_frag_floats(const PuShape &s) {
    return (long long)(s.mode == 1 ? 4 : 1) * s.Cout * s.Cin * s.ntaps;
}
// blob: per phase the weight fragments, then the bias [Cout] padded to 4 floats
__host__ __device__ inline long long pu_blob_offset(const PuDims &d, int i) {
    long long off = 0;
    for (int k = 0; k < i; ++k) { const PuShape s = pu_shape(d, k); off += pu_frag_floats(s) + (s.Cout + 3) / 4 * 4; }
    return off;
}
__host__ __device__ inline long long pu_blobt_offset(const PuDims &d, int i) {     // within the blob's second half
    long long off = 0;
    for (int k = 0; k < i; ++k) off += pu_frag_floats(pu_shape(d, k));
    return off;
}
// workspace: the phases' activations one after the other (the head writes the caller's `out`)
__host__ __device__ inline long long pu_ws_offset(const PuDims &d, int i) {
    long long off = 0;
    for (int k = 0; k < i; ++k) { const PuShape s = pu_shape(d, k); if (s.mode != 2) off += pu_out_floats(d, s); }
    return off;
}

// the K split inside a workgroup: waves per output tile (each takes Cin / KS input channels, a whole number of 8-channel chunks)
inline int pu_waves(int Cin, int ntaps) {
    const int target = ntaps == 9 ? Cin / 8 : Cin / 32;              // 8 channels x 9 taps = 36 MFMAs per wave
    return target >= 8 ? 8 : 4;           // (16 waves leave 128 registers per lane: the loader's loads in flight do not fit)
}

struct PuArgs {
    PuDims d;
    const float *x;          // [n_img][in_ch][H][W]
    const float *blob;       // packed weights + biases (vt_plane_unet_pack)
    float *ws;               // every phase's activations (channels-last)
    float *out;              // [n_img][classes][H][W]
};

inline PuSrc pu_src_of_phase(const PuArgs &a, int i, int pool) {
    const PuShape s = pu_shape(a.d, i);
    PuSrc r;
    r.ptr = a.ws + pu_ws_offset(a.d, i);
    r.C = s.Cout; r.pool = pool; r.nchw = 0;
    r.W = a.d.W >> s.level_out; r.H = a.d.H >> s.level_out;
    return r;
}

inline PuPhase pu_phase(const PuArgs &a, int i) {
    const PuDims &d = a.d;
    const int D = d.depth;
    const PuShape s = pu_shape(d, i);
    PuPhase p;
    p.b.ptr = nullptr; p.b.C = 0; p.b.pool = 0; p.b.nchw = 0; p.b.W = p.b.H = 0;
    if (i == 0) {
        p.a.ptr = a.x; p.a.C = d.in_ch; p.a.pool = 0; p.a.nchw = 1; p.a.W = d.W; p.a.H = d.H;
    } else if (i < 2 * D) {
        p.a = pu_src_of_phase(a, i - 1, (i & 1) ? 0 : 1);             // conv a of a level reads the pooled output of the level above
    } else if (i < 5 * D - 3) {
        const int u = (i - 2 * D) / 3, k = (i - 2 * D) % 3, lev = D - 2 - u;
        p.a = pu_src_of_phase(a, i - 1, 0);                           // k = 0: the level below; k = 1: the transposed conv; k = 2: conv a
        if (k == 1) p.b = pu_src_of_phase(a, 2 * lev + 1, 0);         // cat(up, skip) (unet.py:112-114)
    } else {
        p.a = pu_src_of_phase(a, i - 1, 0);
    }
    const long long boff = pu_blob_offset(d, i);
    p.wfrag = a.blob + boff;
    p.bias = a.blob + boff + pu_frag_floats(s);
    p.out = s.mode == 2 ? a.out : a.ws + pu_ws_offset(d, i);
    p.Cin = s.Cin; p.Cout = s.Cout; p.ntaps = s.ntaps; p.mode = s.mode; p.relu = s.relu; p.n_img = d.n_img;
    p.KS = pu_waves(s.Cin, s.ntaps);
    p.n_cb = (s.mode == 1 ? 4 : 1) * s.Cout / 32;
    p.W = d.W >> s.level_in; p.H = d.H >> s.level_in;
    p.TW = p.W < 32 ? p.W : 32;
    p.TH = p.H < 32 / p.TW ? p.H : 32 / p.TW;
    p.NI = 32 / (p.TW * p.TH);
    p.tiles_x = p.W / p.TW; p.tiles_y = p.H / p.TH;
    p.n_tiles = (d.n_img + p.NI - 1) / p.NI * p.tiles_x * p.tiles_y;
    return p;
}

inline bool pu_dims_ok(const PuDims &d) {
    if (d.depth < 2 || d.depth > PU_MAX_DEPTH || d.n_img <= 0) return false;
    if (d.in_ch <= 0 || d.in_ch % 32 || d.start <= 0 || d.start % 32 || d.classes <= 0 || d.classes % 32) return false;
    if (d.H <= 0 || d.W <= 0 || (d.H & (d.H - 1)) || (d.W & (d.W - 1))) return false;
    if ((d.H >> (d.depth - 1)) < 4 || (d.W >> (d.depth - 1)) < 4) return false;    // (a 2 x 2 level would stage 128 halo pixels per tile)
    if (pu_width(d, d.depth - 1) > 1024 || d.in_ch > 1024 || d.classes > 1024) return false;
    const long long big = d.in_ch > d.classes ? d.in_ch : d.classes, top = big > 2 * d.start ? big : 2 * d.start;
    if ((long long)d.n_img * d.H * d.W * top >= (1ll << 31)) return false;                      // 32-bit element offsets inside a tensor
    return true;
}

__device__ __forceinline__ void pu_wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- loaders.  Every load of a batch is UNCONDITIONAL (coordinates clamped into the tensor, the value dropped afterwards where
// the pixel lies outside the image): the loads of a wave then issue back to back and meet one wait.  With the kind of source a
// run-time branch (nchw / pooled / one or two gradient sources) the compiler emitted a load -> wait group per pixel and tap:
// 330 loads and 249 waits in the weight-gradient kernel, 30-55 us per layer for 4-6 us of MFMAs.  So the kind is a template argument.
enum : int {
    PU_L_PLAIN = 0,      // forward: channels-last source(s), cat(a, b) by chunk
    PU_L_POOL = 1,       // forward: channels-last source read through the 2x2 max-pool
    PU_L_NCHW = 2,       // forward: the network input
    PU_B_ACT = 3,        // backward: one consumer, ReLU mask
    PU_B_FORK = 4,       // backward: a pooled consumer + the skip's consumer, ReLU mask (encoder conv b above the bottom)
    PU_B_UNSHUF = 5,     // backward: transposed conv (the four parities of the consumer's gradient are the GEMM's channels)
    PU_B_NCHW = 6,       // backward: the head (dOut)
};

__device__ __forceinline__ f32x4 pu_ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 pu_max4(f32x4 a, f32x4 b) { return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])}; }

template <int LK> struct PuLoadN { static constexpr int n = LK == PU_L_POOL ? 4 : LK == PU_B_ACT ? 2 : LK == PU_B_FORK ? 6 : 1; };

// issue the loads of one task (4 channels from c of pixel (img, y, x), all in range) into v[0 .. n)
template <int LK>
__device__ __forceinline__ void pu_issue(f32x4 *v, const PuSrc &src, const PuBwd &bw, int img, int y, int x, int c, int W, int H) {
    if constexpr (LK == PU_L_PLAIN) {
        v[0] = pu_ld4(src.ptr + (unsigned)(((img * src.H + y) * src.W + x) * src.C + c));
    } else if constexpr (LK == PU_L_POOL) {
        const float *q = src.ptr + (unsigned)(((img * src.H + 2 * y) * src.W + 2 * x) * src.C + c);
        const unsigned row = (unsigned)(src.W * src.C);
        v[0] = pu_ld4(q); v[1] = pu_ld4(q + src.C); v[2] = pu_ld4(q + row); v[3] = pu_ld4(q + row + src.C);
    } else if constexpr (LK == PU_L_NCHW) {
        const unsigned cs = (unsigned)(src.H * src.W);
        const float *q = src.ptr + (unsigned)((img * src.C + c) * cs + y * src.W + x);
        v[0] = f32x4{q[0], q[cs], q[2 * cs], q[3 * cs]};
    } else if constexpr (LK == PU_B_ACT) {
        const PuGradSrc &g = bw.g0;
        v[0] = pu_ld4(g.ptr + (unsigned)(((img * g.H + y) * g.W + x) * g.C + g.off + c));
        v[1] = pu_ld4(bw.act + (unsigned)(((img * H + y) * W + x) * bw.Cz + c));
    } else if constexpr (LK == PU_B_FORK) {
        const PuGradSrc &g = bw.g0, &k = bw.g1;
        v[0] = pu_ld4(g.ptr + (unsigned)(((img * g.H + (y >> 1)) * g.W + (x >> 1)) * g.C + g.off + c));
        v[1] = pu_ld4(k.ptr + (unsigned)(((img * k.H + y) * k.W + x) * k.C + k.off + c));
        const float *w0 = bw.act + (unsigned)(((img * H + (y & ~1)) * W + (x & ~1)) * bw.Cz + c);
        const unsigned row = (unsigned)(W * bw.Cz);
        v[2] = pu_ld4(w0); v[3] = pu_ld4(w0 + bw.Cz); v[4] = pu_ld4(w0 + row); v[5] = pu_ld4(w0 + row + bw.Cz);
    } else if constexpr (LK == PU_B_UNSHUF) {
        const PuGradSrc &g = bw.g0;
        const int par = c / bw.Cz, co = c - par * bw.Cz;
        v[0] = pu_ld4(g.ptr + (unsigned)(((img * g.H + 2 * y + (par >> 1)) * g.W + 2 * x + (par & 1)) * g.C + g.off + co));
    } else {
        const PuGradSrc &g = bw.g0;
        const unsigned cs = (unsigned)(g.H * g.W);
        const float *q = g.ptr + (unsigned)((img * g.C + g.off + c) * cs + y * g.W + x);
        v[0] = f32x4{q[0], q[cs], q[2 * cs], q[3 * cs]};
    }
}

// what the loads of a task mean: the B operand's 4 values (forward: the input view; backward: the layer's assembled output gradient)
template <int LK>
__device__ __forceinline__ f32x4 pu_combine(const f32x4 *v, int y, int x) {
    if constexpr (LK == PU_L_POOL) return pu_max4(pu_max4(v[0], v[1]), pu_max4(v[2], v[3]));
    else if constexpr (LK == PU_B_ACT) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = v[1][j] > 0.f ? v[0][j] : 0.f;
        return r;
    } else if constexpr (LK == PU_B_FORK) {
        // the pooled consumer's gradient reaches this pixel only where it holds the window's FIRST maximum (torch's max_pool2d)
        const int me = (y & 1) * 2 + (x & 1);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float best = v[2][j];
            int arg = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (v[2 + q][j] > best) { best = v[2 + q][j]; arg = q; }
            const float mine = me == 0 ? v[2][j] : me == 1 ? v[3][j] : me == 2 ? v[4][j] : v[5][j];
            r[j] = mine > 0.f ? v[1][j] + (arg == me ? v[0][j] : 0.f) : 0.f;
        }
        return r;
    } else return v[0];
}

template <int KS, int NT, int LK>
__global__ void __launch_bounds__(KS * 64) plane_unet_phase_kernel(PuPhase p, PuBwd bw) {
    constexpr bool BWD = LK >= PU_B_ACT;
    constexpr int HALO = NT == 9 ? 1 : 0;
    constexpr int STAGE = KS * PU_MAX_HALO * PU_PITCH, RED = (KS > 8 ? KS / 2 : 4) * 16 * 64;
    __shared__ __attribute__((aligned(16))) float lds[STAGE > RED ? STAGE : RED];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, lp = lane & 31;   // (wave: scalar)
    const int HW_ = p.TW + 2 * HALO, HH_ = p.TH + 2 * HALO;           // staged tile: NI x HH_ x HW_ pixels
    const int n_tasks = p.NI * HH_ * HW_ * 2;                         // (pixel, half of the 8 channels)
    const int chunks_per_wave = p.Cin / (KS * PU_CH), n_chunks = p.Cin / PU_CH;
    const int il = lp / (p.TW * p.TH), ly = (lp / p.TW) % p.TH, lx = lp % p.TW;       // this lane's pixel inside the tile
    const int my_slot = (il * HH_ + ly + HALO) * HW_ + lx + HALO;
    float *st = lds + wave * (PU_MAX_HALO * PU_PITCH);
    const int cb = blockIdx.x % p.n_cb, tile = blockIdx.x / p.n_cb;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, ig = tile / (p.tiles_x * p.tiles_y);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ck = 0; ck < chunks_per_wave; ++ck) {
        const int chunk = wave * chunks_per_wave + ck, c0 = chunk * PU_CH;
        // this chunk's weights: one 16-byte load per lane and tap, in flight together with the staging loads
        f32x4 wv[NT];
        const f32x4 *wp = reinterpret_cast<const f32x4 *>(p.wfrag) + ((size_t)(cb * n_chunks + chunk) * NT) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t) wv[t] = wp[t * 64];
        const bool first = c0 < p.a.C;                                // (uniform: the wave's chunk lies in one source)
        const PuSrc &src = first ? p.a : p.b;
        const int cs = BWD ? c0 : first ? c0 : c0 - p.a.C;
        constexpr int NL = PuLoadN<LK>::n, TB = NL > 4 ? 2 : 4;       // tasks per batch: at most 16 16-byte loads in flight per lane
        pu_wave_lds_fence();                                          // the previous chunk's B operands have been read
#pragma unroll
        for (int k0 = 0; k0 < 4; k0 += TB) {
            f32x4 v[TB][NL];
            bool ok[TB];
            int yy[TB], xx[TB], im[TB];
#pragma unroll
            for (int k = 0; k < TB; ++k) {
                const int idx = lane + 64 * (k0 + k), slot = idx >> 1, half = idx & 1;
                const int hx = slot % HW_, hy = (slot / HW_) % HH_, si = slot / (HW_ * HH_);
                const int img = ig * p.NI + si, y = ty * p.TH + hy - HALO, x = tx * p.TW + hx - HALO;
                ok[k] = idx < n_tasks && img < p.n_img && y >= 0 && y < p.H && x >= 0 && x < p.W;
                im[k] = img < p.n_img ? img : p.n_img - 1;
                yy[k] = y < 0 ? 0 : y >= p.H ? p.H - 1 : y;
                xx[k] = x < 0 ? 0 : x >= p.W ? p.W - 1 : x;
                pu_issue<LK>(v[k], src, bw, im[k], yy[k], xx[k], cs + 4 * half, p.W, p.H);
            }
#pragma unroll
            for (int k = 0; k < TB; ++k) {
                const int idx = lane + 64 * (k0 + k), slot = idx >> 1, half = idx & 1;
                f32x4 m = pu_combine<LK>(v[k], yy[k], xx[k]);
                if (!ok[k]) m = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < n_tasks) {
                    float *d = st + slot * PU_PITCH + 4 * half;
                    d[0] = m[0]; d[1] = m[1]; d[2] = m[2]; d[3] = m[3];
                }
                if constexpr (LK == PU_B_ACT || LK == PU_B_FORK) {
                    // the assembled gradient itself, once: the tile's own pixels, by the workgroups of output block 0
                    const int hx = slot % HW_, hy = (slot / HW_) % HH_;
                    if (bw.dz && cb == 0 && ok[k] && hy >= HALO && hy < HALO + p.TH && hx >= HALO && hx < HALO + p.TW)
                        *reinterpret_cast<f32x4 *>(bw.dz + (unsigned)(((im[k] * p.H + yy[k]) * p.W + xx[k]) * bw.Cz + c0 + 4 * half)) = m;
                }
            }
        }
        pu_wave_lds_fence();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int off = NT == 9 ? ((t / 3 - 1) * HW_ + (t % 3 - 1)) : 0;
            const float *bp = st + (my_slot + off) * PU_PITCH + h;
            acc = mfma(wv[t][0], bp[0], acc);
            acc = mfma(wv[t][1], bp[2], acc);
            acc = mfma(wv[t][2], bp[4], acc);
            acc = mfma(wv[t][3], bp[6], acc);
        }
    }
    // ---- the waves' K shares meet in LDS: the upper half folds into the lower until four are left, then wave q sums quad q ----
    __syncthreads();                                                  // every wave is done with its staging region
    f32x4 *red4 = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
    for (int half = KS / 2; half >= 4; half /= 2) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red4[((wave - half) * 4 + q) * 64 + lane] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 o = red4[(wave * 4 + q) * 64 + lane];
                acc[4 * q] += o[0]; acc[4 * q + 1] += o[1]; acc[4 * q + 2] += o[2]; acc[4 * q + 3] += o[3];
            }
        }
        __syncthreads();
    }
    if (wave < 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red4[(wave * 4 + q) * 64 + lane] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    }
    __syncthreads();
    if (wave >= 4) return;
    const int q = wave;                                               // channels cb * 32 + 8 q + 4 h .. + 3 of pixel lp
    f32x4 sum = red4[(0 * 4 + q) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) sum += red4[(w * 4 + q) * 64 + lane];
    const int img = ig * p.NI + il, y = ty * p.TH + ly, x = tx * p.TW + lx;
    if (img >= p.n_img) return;
    int co0 = cb * 32;
    size_t pix = ((size_t)img * p.H + y) * p.W + x;
    if (p.mode == 1) {                                                // transposed conv 2x2 stride 2: block cb belongs to one output parity
        const int par = co0 / p.Cout;
        co0 -= par * p.Cout;
        pix = ((size_t)img * 2 * p.H + 2 * y + (par >> 1)) * (2 * p.W) + 2 * x + (par & 1);
    }
    co0 += 8 * q + 4 * h;
    if (p.bias) sum += *reinterpret_cast<const f32x4 *>(p.bias + co0);
    if (p.relu) sum = f32x4{fmaxf(sum[0], 0.f), fmaxf(sum[1], 0.f), fmaxf(sum[2], 0.f), fmaxf(sum[3], 0.f)};
    if (p.mode == 2) {
        const size_t hw = (size_t)p.H * p.W;
        float *o = p.out + ((size_t)img * p.Cout + co0) * hw + (size_t)y * p.W + x;
        o[0] = sum[0]; o[hw] = sum[1]; o[2 * hw] = sum[2]; o[3 * hw] = sum[3];
    } else *reinterpret_cast<f32x4 *>(p.out + pix * p.Cout + co0) = sum;
}

template <int KS, int NT, int LK>
void pu_launch_k(const PuPhase &p, const PuBwd &bw, hipStream_t s) {
    hipLaunchKernelGGL((plane_unet_phase_kernel<KS, NT, LK>), dim3((unsigned)(p.n_tiles * p.n_cb)), dim3(KS * 64), 0, s, p, bw);
}
template <int NT, int LK>
void pu_launch_nt(const PuPhase &p, const PuBwd &bw, hipStream_t s) {
    if (p.KS == 8) pu_launch_k<8, NT, LK>(p, bw, s);
    else pu_launch_k<4, NT, LK>(p, bw, s);
}
void pu_launch(const PuPhase &p, hipStream_t s) {
    const PuBwd none{};
    if (p.ntaps == 9) {
        if (p.a.nchw) pu_launch_nt<9, PU_L_NCHW>(p, none, s);
        else if (p.a.pool) pu_launch_nt<9, PU_L_POOL>(p, none, s);
        else pu_launch_nt<9, PU_L_PLAIN>(p, none, s);
    } else pu_launch_nt<1, PU_L_PLAIN>(p, none, s);
}
void pu_launch_bwd(const PuPhase &p, const PuBwd &bw, hipStream_t s) {
    if (bw.unshuffle) pu_launch_nt<1, PU_B_UNSHUF>(p, bw, s);
    else if (bw.g0.nchw) pu_launch_nt<1, PU_B_NCHW>(p, bw, s);
    else if (bw.g1.ptr) pu_launch_nt<9, PU_B_FORK>(p, bw, s);
    else pu_launch_nt<9, PU_B_ACT>(p, bw, s);
}

// ---- weights into fragment order -------------------------------------------------------------------------------------------------
struct PuParamPtrs {
    const float *down_w[PU_MAX_DEPTH][2], *down_b[PU_MAX_DEPTH][2];
    const float *up_tw[PU_MAX_DEPTH], *up_tb[PU_MAX_DEPTH], *up_w[PU_MAX_DEPTH][2], *up_b[PU_MAX_DEPTH][2];
    const float *final_w, *final_b;
};

__global__ void __launch_bounds__(256) plane_unet_pack_kernel(PuDims d, PuParamPtrs prm, float *blob) {
    const int i = blockIdx.y, D = d.depth;
    const PuShape s = pu_shape(d, i);
    const float *w, *b;
    if (i < 2 * D) { w = prm.down_w[i >> 1][i & 1]; b = prm.down_b[i >> 1][i & 1]; }
    else if (i < 5 * D - 3) {
        const int u = (i - 2 * D) / 3, k = (i - 2 * D) % 3;
        w = k == 0 ? prm.up_tw[u] : prm.up_w[u][k - 1];
        b = k == 0 ? prm.up_tb[u] : prm.up_b[u][k - 1];
    } else { w = prm.final_w; b = prm.final_b; }
    float *dst = blob + pu_blob_offset(d, i);
    const long long nf = pu_frag_floats(s);
    const int n_chunks = s.Cin / PU_CH;
    // the data-gradient launch's fragments (second half of the blob): out-channels = this phase's INPUT channels, in-channels = its
    // output channels (4 Cout for the transposed conv: parity-major), the 3x3 taps mirrored
    {
        float *dt = blob + pu_blob_offset(d, pu_n_phases(d)) + pu_blobt_offset(d, i);
        const int cin_t = (s.mode == 1 ? 4 : 1) * s.Cout, chunks_t = cin_t / PU_CH;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nf; e += (long long)gridDim.x * 256) {
            const int j = (int)(e & 3), l = (int)((e >> 2) & 63);
            long long q = e >> 8;
            const int t = (int)(q % s.ntaps); q /= s.ntaps;
            const int chunk = (int)(q % chunks_t), cb = (int)(q / chunks_t);
            const int kin = chunk * PU_CH + 2 * j + (l >> 5), ci = cb * 32 + (l & 31);    // kin: channel of the output gradient
            float v;
            if (s.mode == 1) {
                const int par = kin / s.Cout, co = kin - par * s.Cout;
                v = w[(((size_t)ci * s.Cout + co) * 2 + (par >> 1)) * 2 + (par & 1)];
            } else v = w[((size_t)kin * s.Cin + ci) * s.ntaps + (s.ntaps - 1 - t)];
            dt[e] = v;
        }
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nf + s.Cout; e += (long long)gridDim.x * 256) {
        if (e >= nf) { dst[e] = b[e - nf]; continue; }
        const int j = (int)(e & 3), l = (int)((e >> 2) & 63);
        long long q = e >> 8;
        const int t = (int)(q % s.ntaps); q /= s.ntaps;
        const int chunk = (int)(q % n_chunks), cb = (int)(q / n_chunks);
        const int ci = chunk * PU_CH + 2 * j + (l >> 5), coe = cb * 32 + (l & 31);
        float v;
        if (s.mode == 1) {                                           // ConvTranspose2d weight [Cin][Cout][2][2]
            const int par = coe / s.Cout, co = coe - par * s.Cout;
            v = w[(((size_t)ci * s.Cout + co) * 2 + (par >> 1)) * 2 + (par & 1)];
        } else v = w[((size_t)coe * s.Cin + ci) * s.ntaps + t];       // Conv2d weight [Cout][Cin][3][3] or [Cout][Cin][1][1]
        dst[e] = v;
    }
}

// ---- backward: workspace layout, phase descriptors ------------------------------------------------------------------------------
// [dX_j: the gradient of phase j's (concatenated, pooled-resolution) input, j = 1 .. n-1][dZ_j: the assembled output gradient of
// every 3x3 phase][weight-gradient partial sums]
__host__ __device__ inline long long pu_dx_floats(const PuDims &d, int j) {
    const PuShape s = pu_shape(d, j);
    return (long long)d.n_img * (d.H >> s.level_in) * (d.W >> s.level_in) * s.Cin;
}
inline long long pu_dx_offset(const PuDims &d, int j) {
    long long off = 0;
    for (int k = 1; k < j; ++k) off += pu_dx_floats(d, k);
    return off;
}
inline long long pu_dz_offset(const PuDims &d, int j) {
    long long off = pu_dx_offset(d, pu_n_phases(d));
    for (int k = 0; k < j; ++k) { const PuShape s = pu_shape(d, k); if (s.ntaps == 9) off += pu_out_floats(d, s); }
    return off;
}
struct PuWgPlan { int px_per_slice, n_slices, ntaps; long long part_floats, db_floats; };
// A phase's weight gradient costs tiles x pixels x taps MFMA pairs.  All phases share one launch, so the slices are cut for EQUAL work
// per workgroup -- the whole net's cost over ~3 workgroups per slot of the chip (2 per CU) -- instead of a fixed pixel count: fewer,
// longer walks where the weights are small (the partial sums a slice writes are ntaps x Cout x Cin floats whatever its length: 101 MB
// per backward of 24 planes with 128-pixel slices, half of it now), at least 64 pixels each.
inline long long pu_wg_cost(const PuDims &d, int j) {
    const PuShape s = pu_shape(d, j);
    const long long P = (long long)d.n_img * (d.H >> s.level_in) * (d.W >> s.level_in);
    return (long long)(s.Cout / 32) * (s.Cin / 32) * P * (s.mode == 1 ? 4 : s.ntaps);
}
inline PuWgPlan pu_wg_plan(const PuDims &d, int j) {
    const PuShape s = pu_shape(d, j);
    PuWgPlan w;
    const long long P = (long long)d.n_img * (d.H >> s.level_in) * (d.W >> s.level_in);
    w.ntaps = s.mode == 1 ? 4 : s.ntaps;
    long long total = 0;
    for (int k = 0; k < pu_n_phases(d); ++k) total += pu_wg_cost(d, k);
    long long share = total / 1536;                                  // pixel-taps per workgroup
    if (share < 64 * 9) share = 64 * 9;
    long long slices = (P * w.ntaps + share / 2) / share, most = (P + 63) / 64;
    if (slices > most) slices = most;
    if (slices < 1) slices = 1;
    w.px_per_slice = (int)(((P + slices - 1) / slices + 7) / 8 * 8);
    w.n_slices = (int)((P + w.px_per_slice - 1) / w.px_per_slice);
    w.part_floats = (long long)w.n_slices * w.ntaps * s.Cout * s.Cin;
    w.db_floats = (long long)w.n_slices * s.Cout;
    return w;
}
inline long long pu_part_offset(const PuDims &d, int j) {
    long long off = pu_dz_offset(d, pu_n_phases(d));
    for (int k = 0; k < j; ++k) { const PuWgPlan w = pu_wg_plan(d, k); off += w.part_floats + (w.db_floats + 3) / 4 * 4; }
    return off;
}

struct PuBwdArgs {
    PuArgs f;                // the forward's arguments (x, blob, its workspace with the activations)
    const float *dout;       // [n_img][classes][H][W]
    float *bws;              // backward workspace
    float *dx;               // [n_img][in_ch][H][W]
};

inline PuGradSrc pu_gsrc(const PuBwdArgs &a, int consumer, int off, int pooled) {
    const PuShape s = pu_shape(a.f.d, consumer);
    PuGradSrc g;
    g.ptr = a.bws + pu_dx_offset(a.f.d, consumer);
    g.C = s.Cin; g.off = off; g.pooled = pooled; g.nchw = 0;
    g.W = a.f.d.W >> s.level_in; g.H = a.f.d.H >> s.level_in;
    return g;
}

// the output gradient of phase j as its consumers left it (see PuBwd)
inline PuBwd pu_bwd_of(const PuBwdArgs &a, int j) {
    const PuDims &d = a.f.d;
    const int D = d.depth, n = pu_n_phases(d);
    const PuShape s = pu_shape(d, j);
    PuBwd bw{};
    bw.Cz = s.Cout; bw.unshuffle = s.mode == 1;
    bw.act = s.relu ? a.f.ws + pu_ws_offset(d, j) : nullptr;
    bw.dz = s.ntaps == 9 ? a.bws + pu_dz_offset(d, j) : nullptr;
    if (j == n - 1) {
        bw.g0.ptr = a.dout; bw.g0.C = d.classes; bw.g0.off = 0; bw.g0.pooled = 0; bw.g0.nchw = 1; bw.g0.W = d.W; bw.g0.H = d.H;
    } else if (j < 2 * D && (j & 1) && (j >> 1) < D - 1) {
        const int l = j >> 1;
        bw.g0 = pu_gsrc(a, 2 * l + 2, 0, 1);                           // the next level's conv a read this output max-pooled
        bw.g1 = pu_gsrc(a, 2 * D + 3 * (D - 2 - l) + 1, s.Cout, 0);    // the decoder's cat(up, skip): the skip is the second half
    } else {
        bw.g0 = pu_gsrc(a, j + 1, 0, 0);
    }
    return bw;
}

inline PuPhase pu_phase_bwd(const PuBwdArgs &a, int j) {
    const PuDims &d = a.f.d;
    const PuShape s = pu_shape(d, j);
    PuPhase p{};
    p.wfrag = a.f.blob + pu_blob_offset(d, pu_n_phases(d)) + pu_blobt_offset(d, j);
    p.bias = nullptr;
    p.out = j == 0 ? a.dx : a.bws + pu_dx_offset(d, j);
    p.Cin = (s.mode == 1 ? 4 : 1) * s.Cout; p.Cout = s.Cin; p.ntaps = s.ntaps; p.relu = 0; p.n_img = d.n_img;
    p.mode = j == 0 ? 2 : 0;
    p.KS = pu_waves(p.Cin, p.ntaps);
    p.n_cb = p.Cout / 32;
    p.W = d.W >> s.level_in; p.H = d.H >> s.level_in;
    p.TW = p.W < 32 ? p.W : 32;
    p.TH = p.H < 32 / p.TW ? p.H : 32 / p.TW;
    p.NI = 32 / (p.TW * p.TH);
    p.tiles_x = p.W / p.TW; p.tiles_y = p.H / p.TH;
    p.n_tiles = (d.n_img + p.NI - 1) / p.NI * p.tiles_x * p.tiles_y;
    return p;
}

// ---- weight gradients --------------------------------------------------------------------------------------------------------------
// dW[co][ci][tap] = sum over pixels of dZ[px][co] * X[px + tap][ci]: per launch a grid of (pixel slice, 32 output channels, 32 input
// channels); K = the slice's pixels, two per v_mfma_f32_32x32x2f32 (A = dZ^T: 32 channels of a pixel are one 128-byte load; B = the
// forward's input view -- pool and concat resolved by the loader -- at the tap's offset), one accumulator per tap.  The four waves
// take a quarter of the slice each and meet in LDS; partial sums per slice go to the workspace in accumulator order and one
// finalize launch for the whole net sums the slices into the nn.Conv2d / nn.ConvTranspose2d layouts (fixed order: reproducible).
struct PuWg {
    PuSrc a, b;              // V: the phase's forward input view
    PuGradSrc u;             // U: the phase's output gradient (the dz buffer; transposed conv: the consumer's input gradient; head: dOut)
    float *part;             // [n_slices][ntaps][Cout / 32][Cin / 32][16][64]
    float *dbpart;           // [n_slices][Cout]
    int Cin, Cout, W, H, n_img, px_per_slice, n_cob, n_cib;
};

template <int VK>       // PU_L_PLAIN / PU_L_POOL / PU_L_NCHW; coordinates in range
__device__ __forceinline__ float pu_read1(const PuSrc &s, int img, int y, int x, int c) {
    if constexpr (VK == PU_L_NCHW) return s.ptr[(unsigned)((img * s.C + c) * (s.H * s.W) + y * s.W + x)];
    else if constexpr (VK == PU_L_PLAIN) return s.ptr[(unsigned)(((img * s.H + y) * s.W + x) * s.C + c)];
    else {
        const float *q = s.ptr + (unsigned)(((img * s.H + 2 * y) * s.W + 2 * x) * s.C + c);
        const unsigned row = (unsigned)(s.W * s.C);
        return fmaxf(fmaxf(q[0], q[s.C]), fmaxf(q[row], q[row + s.C]));
    }
}

template <int MODE, int VK>      // MODE 0: 3x3 conv (9 taps); 1: transposed conv 2x2 stride 2 (4 parities); 2: 1x1 head
__device__ __forceinline__ void pu_wgrad_body(const PuWg &w, const int block, float *red, float *dbs) {
    constexpr int NT = MODE == 0 ? 9 : MODE == 1 ? 4 : 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, lc = lane & 31;
    const int cib = block % w.n_cib, cob = (block / w.n_cib) % w.n_cob, slice = block / (w.n_cib * w.n_cob);
    const int P = w.n_img * w.H * w.W;
    const int s0 = slice * w.px_per_slice + wave * (w.px_per_slice / 4);
    int s1 = s0 + w.px_per_slice / 4;
    if (s1 > P) s1 = P;
    const int co = cob * 32 + lc, cif = cib * 32 + lc;
    const bool first = cib * 32 < w.a.C;                              // a block of 32 input channels lies in one source (uniform)
    const PuSrc &vs = first ? w.a : w.b;
    const int ci = first ? cif : cif - w.a.C;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lw = __builtin_ctz(w.W), lh = __builtin_ctz(w.H);       // planes are powers of two
    float usum = 0.f;
    constexpr int SB = MODE != 0 ? 4 : VK == PU_L_POOL ? 1 : 2;        // pixel pairs per batch of operands (registers: 36 loads per pooled pair)
    constexpr int NU = MODE == 1 ? 4 : 1, NV = MODE == 0 ? 9 : 1;
    struct Ops { float u[SB][NU], v[SB][NV]; };
    // every load unconditional (pixel clamped into the slice, tap clamped into the image), the value dropped afterwards
    auto load = [&](Ops &o, int pb) {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int pr = pb + 2 * i + h;
            const bool in = pr < s1;
            const int pc = in ? pr : P - 1;
            const int x = pc & (w.W - 1), y = (pc >> lw) & (w.H - 1), img = pc >> (lw + lh);
            if (MODE == 1) {
#pragma unroll
                for (int par = 0; par < 4; ++par) {
                    const float u = w.u.ptr[(unsigned)(((img * w.u.H + 2 * y + (par >> 1)) * w.u.W + 2 * x + (par & 1)) * w.u.C + w.u.off + co)];
                    o.u[i][par] = in ? u : 0.f;
                }
            } else {
                const float u = MODE == 2 ? w.u.ptr[(unsigned)((img * w.u.C + w.u.off + co) * (w.H * w.W) + y * w.W + x)]
                                          : w.u.ptr[(unsigned)(pc * w.u.C + w.u.off + co)];
                o.u[i][0] = in ? u : 0.f;
            }
            if (MODE == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    const bool inside = in && yy >= 0 && yy < w.H && xx >= 0 && xx < w.W;
                    const float v = pu_read1<VK>(vs, img, yy < 0 ? 0 : yy >= w.H ? w.H - 1 : yy, xx < 0 ? 0 : xx >= w.W ? w.W - 1 : xx, ci);
                    o.v[i][t] = inside ? v : 0.f;
                }
            } else {
                const float v = pu_read1<VK>(vs, img, y, x, ci);
                o.v[i][0] = in ? v : 0.f;
            }
        }
    };
    auto fma = [&](const Ops &o) {
#pragma unroll
        for (int i = 0; i < SB; ++i) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = mfma(o.u[i][MODE == 1 ? t : 0], o.v[i][MODE == 0 ? t : 0], acc[t]);
                if (MODE == 1) usum += o.u[i][t];
            }
            if (MODE != 1) usum += o.u[i][0];
        }
    };
    // two batches in flight: the next batch's loads are issued before the current batch's MFMAs
    Ops o0, o1;
    load(o0, s0);
    for (int pb = s0; pb < s1; pb += 4 * SB) {
        load(o1, pb + 2 * SB);
        fma(o0);
        load(o0, pb + 4 * SB);
        if (pb + 2 * SB < s1) fma(o1);
    }
    // the four waves' shares, tap by tap, summed by wave 0 in a fixed order
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (wave) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
            float *o = w.part + ((((size_t)slice * NT + t) * w.n_cob + cob) * w.n_cib + cib) * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r * 64] = ((acc[t][r] + red[(0 * 16 + r) * 64 + lane]) + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (cib == 0) {                                                    // the bias gradient: the sum of dZ over the slice's pixels
        dbs[wave * 64 + lane] = usum;
        __syncthreads();
        if (threadIdx.x < 32) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += dbs[k * 64 + threadIdx.x] + dbs[k * 64 + 32 + threadIdx.x];
            w.dbpart[(size_t)slice * w.Cout + cob * 32 + threadIdx.x] = sum;
        }
    }
}

// Every phase's weight gradient in ONE launch behind the data-gradient chain (their inputs -- the forward activations, the dZ side
// outputs, the consumers' dX -- all survive to the end of the chain): a phase alone offers <= 256 workgroups of a serial walk over its
// slice, i.e. half the chip's slots at best for ~15 us each, nineteen times; together they are ~4000 workgroups, the longest first.
enum { PU_W_CONV = 0, PU_W_CONV_POOL, PU_W_CONV_NCHW, PU_W_CONVT, PU_W_HEAD };
struct PuWgAll { int n; int first[PU_MAX_PHASES + 1]; unsigned char kind[PU_MAX_PHASES + 1]; PuWg w[PU_MAX_PHASES]; };
static_assert(sizeof(PuWgAll) <= 4096, "the phase table travels as a kernel argument: 4 KB at most");

__global__ void __launch_bounds__(256, 2) plane_unet_wgrad_kernel(PuWgAll all) {
    __shared__ float red[3 * 16 * 64];
    __shared__ float dbs[4 * 64];
    int k = 0;
    while (k + 1 < all.n && (int)blockIdx.x >= all.first[k + 1]) ++k;
    const PuWg &w = all.w[k];
    const int block = (int)blockIdx.x - all.first[k];
    switch (all.kind[k]) {
    case PU_W_CONV: pu_wgrad_body<0, PU_L_PLAIN>(w, block, red, dbs); break;
    case PU_W_CONV_POOL: pu_wgrad_body<0, PU_L_POOL>(w, block, red, dbs); break;
    case PU_W_CONV_NCHW: pu_wgrad_body<0, PU_L_NCHW>(w, block, red, dbs); break;
    case PU_W_CONVT: pu_wgrad_body<1, PU_L_PLAIN>(w, block, red, dbs); break;
    default: pu_wgrad_body<2, PU_L_PLAIN>(w, block, red, dbs); break;
    }
}

struct PuFin { const float *part, *dbpart; float *dw, *db; int Cin, Cout, mode, ntaps, n_slices; };
struct PuFinAll { int n; PuFin f[PU_MAX_PHASES]; };

__global__ void __launch_bounds__(256) plane_unet_wgrad_finalize_kernel(PuFinAll all) {
    const PuFin &f = all.f[blockIdx.y];
    const int n_cob = f.Cout / 32, n_cib = f.Cin / 32;
    const long long nw = (long long)f.ntaps * f.Cout * f.Cin;          // = ntaps * tiles * 1024: walked in the partial sums' order
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nw + f.Cout; e += (long long)gridDim.x * 256) {
        if (e >= nw) {
            const int co = (int)(e - nw);
            float sum = 0.f;
            for (int s = 0; s < f.n_slices; ++s) sum += f.dbpart[(size_t)s * f.Cout + co];
            f.db[co] = sum;
            continue;
        }
        float sum = 0.f;
        for (int s = 0; s < f.n_slices; ++s) sum += f.part[(size_t)s * nw + e];
        const int lane = (int)(e & 63), r = (int)((e >> 6) & 15);
        const long long tt = e >> 10;
        const int tile = (int)(tt % (n_cob * n_cib)), t = (int)(tt / (n_cob * n_cib));
        const int co = (tile / n_cib) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), ci = (tile % n_cib) * 32 + (lane & 31);
        if (f.mode == 1) f.dw[(((size_t)ci * f.Cout + co) << 2) + t] = sum;            // ConvTranspose2d [Cin][Cout][2][2]
        else f.dw[((size_t)co * f.Cin + ci) * f.ntaps + t] = sum;                      // Conv2d [Cout][Cin][taps]
    }
}

inline PuDims pu_dims_of(const vt_plane_unet_params *p, int n_img, int H, int W) {
    PuDims d;
    d.depth = p->depth; d.in_ch = p->in_channels; d.start = p->start_filts; d.classes = p->num_classes;
    d.n_img = n_img; d.H = H; d.W = W;
    return d;
}

}  // namespace

extern "C" {

int vt_plane_unet_supported(int depth, int in_channels, int start_filts, int num_classes, int H, int W) {
    PuDims d{depth, in_channels, start_filts, num_classes, 1, H, W};
    return pu_dims_ok(d) ? 1 : 0;
}

size_t vt_plane_unet_blob_bytes(int depth, int in_channels, int start_filts, int num_classes) {
    PuDims d{depth, in_channels, start_filts, num_classes, 1, 64, 64};
    if (depth < 2 || depth > PU_MAX_DEPTH || in_channels % 32 || start_filts % 32 || num_classes % 32 || in_channels <= 0 || start_filts <= 0 ||
        num_classes <= 0)
        return 0;
    return (size_t)(pu_blob_offset(d, pu_n_phases(d)) + pu_blobt_offset(d, pu_n_phases(d))) * sizeof(float);
}

size_t vt_plane_unet_workspace_bytes(int depth, int in_channels, int start_filts, int num_classes, int n_img, int H, int W) {
    PuDims d{depth, in_channels, start_filts, num_classes, n_img, H, W};
    if (!pu_dims_ok(d)) return 0;
    return (size_t)pu_ws_offset(d, pu_n_phases(d)) * sizeof(float);
}

int vt_plane_unet_pack(const vt_plane_unet_params *p, float *blob, size_t blob_bytes, void *stream) {
    if (!p || !blob) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_pack: null argument");
    const size_t need = vt_plane_unet_blob_bytes(p->depth, p->in_channels, p->start_filts, p->num_classes);
    if (!need) return vt_fail(VT_ERR_UNSUPPORTED, "vt_plane_unet_pack: depth 2..5, channel counts multiples of 32");
    if (blob_bytes < need) return vt_fail(VT_ERR_WORKSPACE, "vt_plane_unet_pack: blob too small");
    const PuDims d = pu_dims_of(p, 1, 64, 64);
    PuParamPtrs prm{};
    for (int l = 0; l < d.depth; ++l)
        for (int k = 0; k < 2; ++k) {
            prm.down_w[l][k] = p->down_w[l][k]; prm.down_b[l][k] = p->down_b[l][k];
            if (!prm.down_w[l][k] || !prm.down_b[l][k]) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_pack: null encoder weight");
        }
    for (int u = 0; u + 1 < d.depth; ++u) {
        prm.up_tw[u] = p->up_tw[u]; prm.up_tb[u] = p->up_tb[u];
        if (!prm.up_tw[u] || !prm.up_tb[u]) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_pack: null transposed-conv weight");
        for (int k = 0; k < 2; ++k) {
            prm.up_w[u][k] = p->up_w[u][k]; prm.up_b[u][k] = p->up_b[u][k];
            if (!prm.up_w[u][k] || !prm.up_b[u][k]) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_pack: null decoder weight");
        }
    }
    prm.final_w = p->final_w; prm.final_b = p->final_b;
    if (!prm.final_w || !prm.final_b) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_pack: null head weight");
    hipLaunchKernelGGL(plane_unet_pack_kernel, dim3(64, pu_n_phases(d)), dim3(256), 0, (hipStream_t)stream, d, prm, blob);
    return vt_check(hipGetLastError(), "vt_plane_unet_pack");
}

int vt_plane_unet_fwd(const float *x, int n_img, int H, int W, const vt_plane_unet_params *p, const float *blob,
                      void *workspace, size_t workspace_bytes, float *out, void *stream) {
    if (!x || !p || !blob || !workspace || !out) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_fwd: null argument");
    const PuDims d = pu_dims_of(p, n_img, H, W);
    if (!pu_dims_ok(d))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_plane_unet_fwd: depth 2..5, channel counts multiples of 32, power-of-two planes of at least "
                                           "4 x 4 at the bottom level (vt_plane_unet_supported)");
    if (workspace_bytes < vt_plane_unet_workspace_bytes(d.depth, d.in_ch, d.start, d.classes, n_img, H, W))
        return vt_fail(VT_ERR_WORKSPACE, "vt_plane_unet_fwd: workspace too small");
    PuArgs a;
    a.d = d; a.x = x; a.blob = blob; a.out = out; a.ws = reinterpret_cast<float *>(workspace);
    for (int i = 0; i < pu_n_phases(d); ++i) pu_launch(pu_phase(a, i), (hipStream_t)stream);
    return vt_check(hipGetLastError(), "vt_plane_unet_fwd");
}

size_t vt_plane_unet_bwd_workspace_bytes(int depth, int in_channels, int start_filts, int num_classes, int n_img, int H, int W) {
    PuDims d{depth, in_channels, start_filts, num_classes, n_img, H, W};
    if (!pu_dims_ok(d)) return 0;
    return (size_t)pu_part_offset(d, pu_n_phases(d)) * sizeof(float);
}

int vt_plane_unet_bwd(const float *x, int n_img, int H, int W, const vt_plane_unet_params *p, const float *blob, const void *fwd_workspace,
                      const float *dout, void *workspace, size_t workspace_bytes, const vt_plane_unet_grads *grads, float *dx, void *stream) {
    if (!x || !p || !blob || !fwd_workspace || !dout || !workspace || !grads || !dx) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_bwd: null argument");
    const PuDims d = pu_dims_of(p, n_img, H, W);
    if (!pu_dims_ok(d)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_plane_unet_bwd: shape not covered (vt_plane_unet_supported)");
    if (workspace_bytes < vt_plane_unet_bwd_workspace_bytes(d.depth, d.in_ch, d.start, d.classes, n_img, H, W))
        return vt_fail(VT_ERR_WORKSPACE, "vt_plane_unet_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    PuBwdArgs a;
    a.f.d = d; a.f.x = x; a.f.blob = blob; a.f.out = nullptr;
    a.f.ws = const_cast<float *>(reinterpret_cast<const float *>(fwd_workspace));
    a.dout = dout; a.bws = reinterpret_cast<float *>(workspace); a.dx = dx;
    const int n = pu_n_phases(d), D = d.depth;
    PuFinAll fin{};
    fin.n = n;
    PuWg wg[PU_MAX_PHASES];
    int wg_kind[PU_MAX_PHASES], wg_blocks[PU_MAX_PHASES];
    long long wg_cost[PU_MAX_PHASES];
    for (int j = n - 1; j >= 0; --j) {
        const PuShape sh = pu_shape(d, j);
        const PuBwd bw = pu_bwd_of(a, j);
        pu_launch_bwd(pu_phase_bwd(a, j), bw, s);                      // the input gradient of phase j (and the assembled dZ_j)
        // its weight / bias gradient
        const PuPhase fp = pu_phase(a.f, j);
        const PuWgPlan plan = pu_wg_plan(d, j);
        PuWg w;
        w.a = fp.a; w.b = fp.b;
        if (sh.ntaps == 9) { w.u.ptr = bw.dz; w.u.C = sh.Cout; w.u.off = 0; w.u.pooled = 0; w.u.nchw = 0; w.u.W = fp.W; w.u.H = fp.H; }
        else w.u = bw.g0;
        float *part = a.bws + pu_part_offset(d, j);
        w.part = part; w.dbpart = part + plan.part_floats;
        w.Cin = sh.Cin; w.Cout = sh.Cout; w.W = fp.W; w.H = fp.H; w.n_img = n_img; w.px_per_slice = plan.px_per_slice;
        w.n_cob = sh.Cout / 32; w.n_cib = sh.Cin / 32;
        wg[j] = w;
        wg_blocks[j] = plan.n_slices * w.n_cob * w.n_cib;
        wg_kind[j] = sh.mode == 1 ? PU_W_CONVT : sh.ntaps == 1 ? PU_W_HEAD : fp.a.nchw ? PU_W_CONV_NCHW : fp.a.pool ? PU_W_CONV_POOL : PU_W_CONV;
        wg_cost[j] = (long long)plan.px_per_slice * plan.ntaps * (fp.a.pool ? 2 : 1);      // a workgroup's walk: the longest go first
        PuFin &f = fin.f[j];
        f.part = w.part; f.dbpart = w.dbpart; f.Cin = sh.Cin; f.Cout = sh.Cout; f.mode = sh.mode; f.ntaps = plan.ntaps; f.n_slices = plan.n_slices;
        if (j < 2 * D) { f.dw = grads->down_w[j >> 1][j & 1]; f.db = grads->down_b[j >> 1][j & 1]; }
        else if (j < 5 * D - 3) {
            const int u = (j - 2 * D) / 3, k = (j - 2 * D) % 3;
            f.dw = k == 0 ? grads->up_tw[u] : grads->up_w[u][k - 1];
            f.db = k == 0 ? grads->up_tb[u] : grads->up_b[u][k - 1];
        } else { f.dw = grads->final_w; f.db = grads->final_b; }
        if (!f.dw || !f.db) return vt_fail(VT_ERR_INVALID, "vt_plane_unet_bwd: null gradient buffer");
    }
    {
        PuWgAll all{};
        all.n = n;
        int idx[PU_MAX_PHASES];
        for (int j = 0; j < n; ++j) idx[j] = j;
        std::stable_sort(idx, idx + n, [&](int x, int y) { return wg_cost[x] > wg_cost[y]; });
        int total = 0;
        for (int k = 0; k < n; ++k) {
            all.first[k] = total; all.w[k] = wg[idx[k]]; all.kind[k] = (unsigned char)wg_kind[idx[k]];
            total += wg_blocks[idx[k]];
        }
        all.first[n] = total;
        hipLaunchKernelGGL(plane_unet_wgrad_kernel, dim3((unsigned)total), dim3(256), 0, s, all);
    }
    hipLaunchKernelGGL(plane_unet_wgrad_finalize_kernel, dim3(256, n), dim3(256), 0, s, fin);
    return vt_check(hipGetLastError(), "vt_plane_unet_bwd");
}

}  // extern "C"
