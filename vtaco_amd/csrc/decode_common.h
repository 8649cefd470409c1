// Device helpers shared by the decode forward (decode.hip) and backward (decode_bwd.hip)
// kernels: MFMA wrappers, fragment loads/stores, the reference's coordinate maths and the
// trilinear corner set-up.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// max(x,0) as ONE instruction (v_med3_f32 x, 0, +inf).  fmaxf() on an MFMA result costs a
// second, canonicalising v_max; inline asm is not an option: hipcc inserts the MFMA->VALU
// wait states only for instructions it can see, and a hand-written v_max read stale
// accumulators (measured: 6e-2 logit error).
__device__ __forceinline__ float relu1(float x) {
    return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff());
}

// Force the 16 values to exist in VGPRs here (stops LLVM sinking the FMAs that produce
// them past later loads, which would keep every load of the gather in flight at once).
__device__ __forceinline__ void pin16(f32x16 &v) {
    asm volatile("" : "+v"(v));
}

// one dense 32x32 layer; x[s] is the B operand of k-step s.  With RELU the 16 v_max run as
// one block IN FRONT of the MFMA chain (pinned): VALU instructions interleaved between
// dependent MFMAs cost ~13 % of the matrix rate (measured, tools/probe/mlp_probe.hip).
template <bool RELU>
__device__ __forceinline__ f32x16 dense32(f32x16 acc, const float *wl, const f32x16 &x, int lane) {
    f32x16 b = x;
    if (RELU) {
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = relu1(x[s]);
        asm volatile("" : "+v"(b));
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mfma(wl[s * 64 + lane], b[s], acc);
    return acc;
}

// ---- split 16-bit ("bf16x3" / "f16x3") dense layers -------------------------------------
// An f32 value v is carried as hi + lo, two 16-bit floats, and a product W x as
// W_lo x_hi + W_hi x_lo + W_hi x_hi  on the 16-bit matrix core with f32 accumulation:
// 6 x v_mfma_f32_32x32x16_{bf16,f16} (192 cycles) per 32x32 layer instead of
// 16 x v_mfma_f32_32x32x2_f32 (1024 cycles).  Same accumulator-as-operand chaining as the f32
// path: registers 8s..8s+7 of the previous layer's accumulator are the 8 k-elements of k-step s
// (weights pre-permuted).
//   P = 1, split-bf16: hi = bf16(v), lo = bf16(v - hi): 16 mantissa bits, f32's exponent range; the
//          dropped lo*lo term is 2^-16 relative.  relu + split costs ~4 VALU instructions per value.
//   P = 2, split-f16:  hi, lo are IEEE halves: 21-22 mantissa bits (dropped term 2^-21) for |v| < 65504
//          (larger magnitudes saturate; absolute resolution 6e-8 from the half subnormals).  The point
//          of this form is the VALU: relu + split is TWO instructions per value --
//              hi2 = v_cvt_pkrtz_f16_f32(a, b)      (round toward zero: hi <= v for v > 0)
//              hi2 = v_pk_max_f16(hi2, 0)           (relu on the packed halves)
//              lo  = v_fma_mix{lo,hi}_f16(-hi, 1.0, v) clamp
//          the mix instruction reads the half straight out of the packed register, subtracts in f32,
//          rounds to f16 into one half of the destination; its clamp to [0,1] is the relu of lo
//          (v > 0: 0 <= v - hi < ulp(hi) << 1 is left alone; v <= 0: hi = 0 and v - 0 <= 0 clamps to 0).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int P> struct MxT;
template <> struct MxT<1> { typedef bf16x8 v8; };
template <> struct MxT<2> { typedef f16x8 v8; };
template <int P> using mx8 = typename MxT<P>::v8;

__device__ __forceinline__ f32x16 mfma_s(const bf16x8 &a, const bf16x8 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_s(const f16x8 &a, const f16x8 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int P> struct SplitP {
    mx8<P> hi[2], lo[2];
};
typedef SplitP<1> Split16;

// lo = (half)fma((float)hi, m1, v) with m1 = -1 hidden from the optimiser (it would fold the fma to a subtract,
// which has no mix form) selects v_fma_mixlo_f16 / v_fma_mixhi_f16, and the packed min/max pair becomes their
// clamp bit -- provided the SLP vectoriser does not pair the two fmas into a v_pk_fma_f32 first: the split-f16
// kernels are compiled in their own translation unit with -fno-slp-vectorize (decode_f16.hip).
__device__ __forceinline__ float opaque_minus_one() {
    float m1 = -1.0f;
    asm("" : "+s"(m1));
    return m1;
}
template <bool RELU>
__device__ __forceinline__ void split_pair_f16(float a, float b, float m1, unsigned &hw, unsigned &lw) {
    f16x2 hp, lo;
    const f16x2 zero = {(_Float16)0.0f, (_Float16)0.0f}, one = {(_Float16)1.0f, (_Float16)1.0f};
    if (RELU) {
        hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
        hp = __builtin_elementwise_max(hp, zero);
    } else {
        hp = f16x2{(_Float16)a, (_Float16)b};
    }
    lo[0] = (_Float16)__builtin_fmaf((float)hp[0], m1, a);
    lo[1] = (_Float16)__builtin_fmaf((float)hp[1], m1, b);
    if (RELU) lo = __builtin_elementwise_min(__builtin_elementwise_max(lo, zero), one);
    hw = __builtin_bit_cast(unsigned, hp);
    lw = __builtin_bit_cast(unsigned, lo);
}

// (bf16: v - hi through v_dot2_f32_bf16 saves the expansion of hi back to f32 -- 6 instead of 8 instructions
// per pair -- but measured no faster, and its result is not the exact difference: parity failed.)
template <bool RELU, int P = 1>
__device__ __forceinline__ SplitP<P> split16(const f32x16 &x) {
    SplitP<P> r;
    if constexpr (P == 1) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = RELU ? relu1(x[8 * s + j]) : x[8 * s + j];
                const __bf16 hb = (__bf16)v;
                r.hi[s][j] = hb;
                r.lo[s][j] = (__bf16)(v - (float)hb);
            }
        }
    } else {
        const float m1 = opaque_minus_one();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 hw, lw;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h, l;
                split_pair_f16<RELU>(x[8 * s + 2 * q], x[8 * s + 2 * q + 1], m1, h, l);
                hw[q] = h; lw[q] = l;
            }
            r.hi[s] = __builtin_bit_cast(f16x8, hw);
            r.lo[s] = __builtin_bit_cast(f16x8, lw);
        }
    }
    return r;
}

// ---- range guard of the split-f16 layers ----------------------------------------------------------------------------
// hi = v_cvt_pkrtz_f16_f32(v) saturates at 65504 for |v| >= 65504 (round toward zero never produces an infinity): the layers
// then silently lose parity.  Every relu + split hands its FIRST hi pair (two of a lane's sixteen channels, all points, all
// layers) to a running unsigned maximum (one v_pk_max_u16 per split: the values are >= 0 after the relu); a wave that ends
// with the largest half in it sets bit 0 of the status word (vt_decode_range_status).  A sample, not a proof: a network whose
// activations leave the half range does so on whole channels and regions, not on one value.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void range_track(unsigned &rmax, unsigned hi_pair) {
    rmax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, rmax), __builtin_bit_cast(u16x2, hi_pair)));
}
// bit 0 (VT_RANGE_HALF): a sampled hi half reached 65504 (0x7bff) -- every half-precision form has lost parity there.
// bit 1 (VT_RANGE_FP8, only the "f16f8" kernel reports it): a sampled hi half reached 2^(8 + VT_F8_SX) = 1024 -- beyond that
// the fp8 copies of the correction products start to clip (x_lo 2^(11 - SX) < 2 x_hi 2^-SX against e4m3's 448; the x_hi copy
// itself saturates at 448 2^SX = 1792) and the result degrades towards two-product f16 accuracy (2^-11 relative): the 1e-4
// contract of that kernel no longer holds although nothing overflows.
// bit 2 (VT_RANGE_LOGIT, "f16f8" only): a logit beyond VT_F16F8_LOGIT_LIMIT was written.  The fp8 form's error is RELATIVE
// (~3e-5 |logit|: 4-bit corrections of 2^-11 terms), so its 1e-4 ABSOLUTE contract holds for |logit| <= 2.5 and no further.
constexpr unsigned VT_RANGE_HALF = 1u, VT_RANGE_FP8 = 2u, VT_RANGE_LOGIT = 4u;
constexpr float VT_F16F8_LOGIT_LIMIT = 2.5f;
__device__ __forceinline__ void range_report(unsigned rmax, unsigned *status, bool fp8_copies = false) {
    if (status == nullptr) return;
    const unsigned top = max(rmax & 0xffffu, rmax >> 16);
    constexpr unsigned FP8_LIMIT = (unsigned)(15 + 8 + VT_F8_SX) << 10;          // the half 2^(8 + SX)
    const unsigned bits = (top >= 0x7bffu ? VT_RANGE_HALF : 0u) | ((fp8_copies && top >= FP8_LIMIT) ? VT_RANGE_FP8 : 0u);
    if (bits) atomicOr(status, bits);
}

// ---- in-kernel clock stamps of the lattice kernels (bench.py's evidence of the clock the chip held) ------------------------
// Every workgroup is persistent over its share of the tiles.  Each reads the constant-rate counter (s_memrealtime: one time
// base for the whole chip) at both ends and leaves (start, end) in the device's status block: the spread of the starts and
// the ends is the launch's ramp and tail.  Workgroup 0 also reads the shader clock counter (s_memtime) at both ends: cycles
// over ticks is the clock the chip held under this kernel's load (vt_decode_last_clock).  A few scalar reads per workgroup
// and one 16-byte store: nothing measurable.
constexpr int VT_CLK_MAX_WGS = 512;                  // (start, end) pairs kept per launch
constexpr unsigned VT_TAIL_LDS_BYTES = 256;          // behind a lattice kernel's images: the tile counter (16 B) + the waves' end stamps
struct ClockStamp {
    unsigned long long t0, r0;
};
__device__ __forceinline__ ClockStamp clock_begin(const unsigned long long *clk) {
    ClockStamp c{0ull, 0ull};
    if (clk != nullptr) {                                    // scalar conditions: the stamps live in SGPRs across the kernel
        c.r0 = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 0) c.t0 = __builtin_readcyclecounter();
    }
    return c;
}
// wave_ends: LDS scratch of (workgroup waves) 64-bit words nobody else uses any more.  Every wave leaves its own end there
// and the workgroup meets at a barrier (finished waves wait instead of exiting: free), so the recorded end is the LAST wave's.
__device__ __forceinline__ void clock_end(unsigned long long *clk, const ClockStamp &c, unsigned long long *wave_ends) {
    if (clk != nullptr) {
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long dt = blockIdx.x == 0 ? __builtin_readcyclecounter() - c.t0 : 0ull;
        const int nwaves = (int)(blockDim.x >> 6);
        if ((threadIdx.x & 63) == 0) wave_ends[threadIdx.x >> 6] = r1;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long last = r1;
            for (int w = 1; w < nwaves; ++w) last = wave_ends[w] > last ? wave_ends[w] : last;
            if (blockIdx.x == 0) { clk[0] = dt; clk[1] = r1 - c.r0; clk[2] = gridDim.x; }      // wave 0's own lifetime: the clock
            if (blockIdx.x < VT_CLK_MAX_WGS) { clk[4 + 2 * blockIdx.x] = c.r0; clk[5 + 2 * blockIdx.x] = last; }
        }
    }
}

// wl: the layer's LDS image [part: hi, lo][k-step 0,1][lane] x 16 bytes
template <int P>
__device__ __forceinline__ f32x16 dense32s(f32x16 acc, const float *wl, const SplitP<P> &x, int lane) {
    const mx8<P> *w = reinterpret_cast<const mx8<P> *>(wl);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const mx8<P> wh = w[s * 64 + lane], wo = w[(2 + s) * 64 + lane];
        acc = mfma_s(wo, x.hi[s], acc);
        acc = mfma_s(wh, x.lo[s], acc);
        acc = mfma_s(wh, x.hi[s], acc);
    }
    return acc;
}

// one k-step (s = 0 or 1) of dense32s
template <int P>
__device__ __forceinline__ f32x16 dense32s_half(f32x16 acc, const float *wl, const SplitP<P> &x, int lane, int s) {
    const mx8<P> *w = reinterpret_cast<const mx8<P> *>(wl);
    const mx8<P> wh = w[s * 64 + lane], wo = w[(2 + s) * 64 + lane];
    acc = mfma_s(wo, x.hi[s], acc);
    acc = mfma_s(wh, x.lo[s], acc);
    acc = mfma_s(wh, x.hi[s], acc);
    return acc;
}

__device__ __forceinline__ f32x16 load_frag16(const float *p) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(p);
    f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    f32x16 r;
    r.s0 = a.x; r.s1 = a.y; r.s2 = a.z; r.s3 = a.w;
    r.s4 = b.x; r.s5 = b.y; r.s6 = b.z; r.s7 = b.w;
    r.s8 = c.x; r.s9 = c.y; r.sa = c.z; r.sb = c.w;
    r.sc = d.x; r.sd = d.y; r.se = d.z; r.sf = d.w;
    return r;
}

// reference src/common.py:293-309 followed by ATen's align_corners=True
// un-normalisation and border clip (decoder.py:62-68): returns the continuous
// grid coordinate in [0, R-1].
__device__ __forceinline__ float grid_coord(float v, float divisor, int R) {
    float q = v / divisor + 0.5f;
    q = (q >= 1.0f) ? 0.999f : q;
    q = (q < 0.0f) ? 0.0f : q;
    float g = 2.0f * q - 1.0f;
    float f = ((g + 1.0f) / 2.0f) * (float)(R - 1);
    return fminf(fmaxf(f, 0.0f), (float)(R - 1));
}

struct DecodeArgs {
    const float *grid;   // [B,R,R,R,32]
    const float *pts;    // [B,N,3] or null
    const float *c_img;  // [B,N,32] or null
    const float *blob;
    float *out;
    float *out2;
    float *save;         // [VT_SAVE_SLOTS][total][32] activations for the backward, or null
    const float *c_direct;  // [B,N,32] conditioning features given directly (no grid gather), or null
    const unsigned char *cimg_ids;   // [B,N] finger id per point (255 = none) with cimg_table, instead of c_img
    const float *cimg_table;         // [F][32] tactile feature per finger
    uint32_t cimg_nf;                // F: an id >= F (255 included) reads as "no feature" -- never as a row past the table
    int brick;           // lattice mode with tiles = 2x4x4 bricks (slab aligned to x-plane pairs, nx % 4 == 0)
    uint32_t N;          // points per batch element
    uint32_t total;      // B*N   (< 2^31, checked by the entry point)
    uint32_t lattice_first;
    int R;
    int nx;
    float box;
    float divisor;       // 1 + padding + 10e-4
    unsigned *status;    // device word of the range guard (VT_RANGE_* bits), or null
    unsigned long long *clk;   // the device's clock-stamp block (lattice kernels: clock_begin / clock_end), or null
    int claim;                 // lattice kernels: waves claim tiles from the workgroup's LDS counter (1) or walk a fixed share (0)
};


// accumulator layout (register r of lane-half h = channel (r&3)+8(r>>2)+4h) <-> a [32]-float row
__device__ __forceinline__ void store_acc16(float *row, const f32x16 &v, int h) {
    f32x4 *q = reinterpret_cast<f32x4 *>(row + 4 * h);
    q[0] = f32x4{v.s0, v.s1, v.s2, v.s3};
    q[2] = f32x4{v.s4, v.s5, v.s6, v.s7};
    q[4] = f32x4{v.s8, v.s9, v.sa, v.sb};
    q[6] = f32x4{v.sc, v.sd, v.se, v.sf};
}
__device__ __forceinline__ f32x16 load_acc16(const float *row, int h) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(row + 4 * h);
    const f32x4 a = q[0], b = q[2], c = q[4], d = q[6];
    f32x16 r;
    r.s0 = a.x; r.s1 = a.y; r.s2 = a.z; r.s3 = a.w;
    r.s4 = b.x; r.s5 = b.y; r.s6 = b.z; r.s7 = b.w;
    r.s8 = c.x; r.s9 = c.y; r.sa = c.z; r.sb = c.w;
    r.sc = d.x; r.sd = d.y; r.se = d.z; r.sf = d.w;
    return r;
}
// gather layout (register s of lane-half h = channel 16h+s)
__device__ __forceinline__ void store_gather16(float *row, const f32x16 &v, int h) {
    f32x4 *q = reinterpret_cast<f32x4 *>(row + 16 * h);
    q[0] = f32x4{v.s0, v.s1, v.s2, v.s3};
    q[1] = f32x4{v.s4, v.s5, v.s6, v.s7};
    q[2] = f32x4{v.s8, v.s9, v.sa, v.sb};
    q[3] = f32x4{v.sc, v.sd, v.se, v.sf};
}
__device__ __forceinline__ f32x16 relu16(const f32x16 &v) {
    f32x16 r;
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = relu1(v[s]);
    return r;
}

// box * linspace(-0.5, 0.5, nx)[i] per axis (src/common.py:178-197, generation.py:155-157)
__device__ __forceinline__ void lattice_point(const DecodeArgs &a, uint32_t ix, uint32_t iy, uint32_t iz,
                                              float &px, float &py, float &pz) {
    const float step = 1.0f / (float)(a.nx - 1);
    const int half = a.nx / 2;
    auto lin = [&](int i) {
        // ONE rounding per element, as torch's CPU linspace kernel evaluates start + step * i (a fused multiply-add: against
        // torch 2.10 -- the oracle's and the goldens' arithmetic -- 0 mismatching indices for nx = 32 .. 256; a separate multiply
        // and add differs in the last bit on 28 / 42 / 124 of the 64 / 128 / 256 indices of an axis, which moves sampled features by
        // up to 2e-5 on a rough grid)
        float v = (i < half) ? fmaf(step, (float)i, -0.5f) : fmaf(-step, (float)(a.nx - i - 1), 0.5f);
        return a.box * v;
    };
    px = lin((int)ix); py = lin((int)iy); pz = lin((int)iz);
}

// query point of global index g (points tensor or in-kernel lattice)
__device__ __forceinline__ void point_of(const DecodeArgs &a, uint32_t g, uint32_t n, float &px, float &py, float &pz) {
    if (a.pts) {
        const float *pp = a.pts + (size_t)g * 3;
        px = pp[0]; py = pp[1]; pz = pp[2];
    } else {
        // box * linspace(-0.5, 0.5, nx)[i] (src/common.py:178-197, generation.py:155-157)
        const uint32_t m = a.lattice_first + n;
        const uint32_t nx = (uint32_t)a.nx;
        const uint32_t t = m / nx;
        const uint32_t iz = m - t * nx;
        const uint32_t ix = t / nx;
        const uint32_t iy = t - ix * nx;
        lattice_point(a, ix, iy, iz, px, py, pz);
    }
}

// the 8 corners and weights of ATen's trilinear grid_sample (align_corners, border)
struct Tri {
    int x0, x1, y0, y1, z0, z1;
    float wx0, wx1, wy0, wy1, wz0, wz1;     // weights; the +1 weight is 0 when the corner is out of bounds
};
__device__ __forceinline__ Tri tri_setup(float px, float py, float pz, float divisor, int R) {
    Tri t;
    const float fx = grid_coord(px, divisor, R), fy = grid_coord(py, divisor, R), fz = grid_coord(pz, divisor, R);
    const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
    t.x0 = (int)x0f; t.y0 = (int)y0f; t.z0 = (int)z0f;
    t.wx0 = (x0f + 1.0f) - fx; t.wy0 = (y0f + 1.0f) - fy; t.wz0 = (z0f + 1.0f) - fz;
    t.x1 = min(t.x0 + 1, R - 1); t.y1 = min(t.y0 + 1, R - 1); t.z1 = min(t.z0 + 1, R - 1);
    // a corner beyond the border is skipped by ATen; its weight is 0 there anyway
    t.wx1 = (t.x0 + 1 <= R - 1) ? fx - x0f : 0.0f;
    t.wy1 = (t.y0 + 1 <= R - 1) ? fy - y0f : 0.0f;
    t.wz1 = (t.z0 + 1 <= R - 1) ? fz - z0f : 0.0f;
    return t;
}

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace
