// Marching cubes (Lewiner / "MC33") on gfx950, output identical -- vertex numbering
// included -- to what the reference gets from
//     skimage.measure.marching_cubes(value_grid, gradient_direction='ascent')
// (reference src/conv_onet/generation.py:268-273; scikit-image's Lewiner port).
//
// scikit-image sweeps the cells sequentially (array axis 0 outermost, axis 2
// innermost) and numbers a vertex the first time a triangle refers to it.  Every
// vertex sits on a grid edge (or is a cell's centre vertex), and every cell that
// touches a sign-changing edge uses it, so the first user of an edge is its
// lexicographically smallest adjacent cell: the edge's OWNER.  That turns the
// sequential numbering into a scan:
//   vertex id = (# vertices owned by earlier cells) + rank of the edge among the
//               owner's owned edges in the owner's triangle-list order.
// Pipeline (all on the caller's stream, no host round trip inside):
//   minmax    -> iso level 0.5*(min+max) when the caller asks for skimage's default
//   classify  -> per cell: Lewiner case/sub-case (face + interior tests in f64),
//                triangle-list offset, #triangles, owned-edge ranks; per-block sums
//   scan      -> exclusive offsets of the block sums, totals
//   vertices  -> owners write their vertices (inverse-|value| weighted, f64 -> f32)
//   faces     -> every triangle corner resolves (owner cell, owner-local edge) -> id
// HBM-class kernels, latency-dominated at 128^3 (8.4 MB volume).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>

#include "vt_common.h"

#define MC_TABLE_QUALIFIER __device__ const
#include "mc_tables.inc"

namespace {

constexpr double MC_EPS = 2.220446049250313e-16;   // skimage's "FLT_EPSILON" = np.spacing(1.0)
constexpr int CELLS_PER_BLOCK = 256;

struct McHeader {
    unsigned unused0;      // (formerly the atomic min / max keys)
    unsigned unused1;
    int nverts;
    int nfaces;
    double level;
    int reserved[10];
};
static_assert(sizeof(McHeader) == 64, "header is 64 bytes");

struct McDims {
    int n0, n1, n2;        // volume extents (axis 0,1,2); x runs along axis 2
    int c1, c2;            // cells along axis 1 and 2
    unsigned ncells;
};

struct McWs {
    McHeader *hdr;
    uint16_t *cnt;         // ntri | nnew << 8
    uint32_t *desc;        // offset of the cell's triangle list in MC_LUT (active cells only)
    uint64_t *rank;        // 13 nibbles: 1 + rank of owned edge e (0 = not owned / unused)
    uint32_t *vbase;       // first vertex id owned by the cell (active cells only)
    uint2 *bsum;           // per block (tri, vert) sums
    uint2 *boff;           // per block exclusive offsets
};

// per-workgroup (min, max) of the volume into part[blockIdx.x] (at most MC_MM_BLOCKS of them: the classify kernel
// reduces them again in every wave -- 2 KB from L2 -- which needs neither atomics nor a pre-initialised header)
constexpr int MC_MM_BLOCKS = 256;
__global__ void __launch_bounds__(256) mc_minmax_kernel(const float *vol, size_t n, float2 *part) {
    __shared__ float slo[4], shi[4];
    float lo = INFINITY, hi = -INFINITY;
    const size_t n4 = ((reinterpret_cast<uintptr_t>(vol) & 15) == 0) ? n / 4 : 0;   // 16-B loads need alignment
    const float4 *v4 = reinterpret_cast<const float4 *>(vol);
    // eight independent 16-byte loads in flight per thread: the pass is latency-bound (8 MB at 128^3), not bandwidth-bound
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const size_t i = i0 + k * stride;
            v[k] = i < n4 ? v4[i] : v4[i0];                          // past the end: repeat a value already counted
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            lo = fminf(fminf(lo, v[k].x), fminf(v[k].y, fminf(v[k].z, v[k].w)));
            hi = fmaxf(fmaxf(hi, v[k].x), fmaxf(v[k].y, fmaxf(v[k].z, v[k].w)));
        }
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = vol[i];                                // tail (or everything, if unaligned)
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = slo[0]; hi = shi[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { lo = fminf(lo, slo[i]); hi = fmaxf(hi, shi[i]); }
        part[blockIdx.x] = make_float2(lo, hi);
    }
}

// skimage's default level 0.5 * (volume.min() + volume.max()), the sum rounded in float32; `part`: the npart partial
// (min, max) pairs of mc_minmax_kernel (npart <= 256: four per lane)
__device__ __forceinline__ double iso_level(const float2 *part, int npart, double level, int auto_level) {
    if (!auto_level) return level;
    const int lane = threadIdx.x & 63;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = lane; i < npart; i += 64) { const float2 p = part[i]; lo = fminf(lo, p.x); hi = fmaxf(hi, p.y); }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    return 0.5 * (double)(lo + hi);
}

// corner k -> (dx,dy,dz), Lewiner order
__device__ __forceinline__ int cx(int k) { return (0x66 >> k) & 1; }   // 0,1,1,0,0,1,1,0
__device__ __forceinline__ int cy(int k) { return (0xCC >> k) & 1; }   // 0,0,1,1,0,0,1,1
__device__ __forceinline__ int cz(int k) { return k >> 2; }

__device__ __forceinline__ void load_cell(const float *vol, const McDims &d, int x, int y, int z, double level, double v[8]) {
    const size_t s1 = (size_t)d.n2, s0 = (size_t)d.n1 * d.n2;
    const float *p = vol + (size_t)z * s0 + (size_t)y * s1 + x;
    v[0] = (double)p[0] - level;          v[1] = (double)p[1] - level;
    v[3] = (double)p[s1] - level;         v[2] = (double)p[s1 + 1] - level;
    v[4] = (double)p[s0] - level;         v[5] = (double)p[s0 + 1] - level;
    v[7] = (double)p[s0 + s1] - level;    v[6] = (double)p[s0 + s1 + 1] - level;
}

// select v[k] with a runtime k without sending the array to scratch
__device__ __forceinline__ double pick(const double v[8], int k) {
    double r = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) r = (k == i) ? v[i] : r;
    return r;
}

__device__ bool test_face(const double v[8], int face) {
    const int f = face < 0 ? -face : face;
    // corners (A,B,C,D) of faces 1..6, packed 3 bits each
    const unsigned tab[7] = {0, 0 | 4 << 3 | 5 << 6 | 1 << 9, 1 | 5 << 3 | 6 << 6 | 2 << 9, 2 | 6 << 3 | 7 << 6 | 3 << 9,
                             3 | 7 << 3 | 4 << 6 | 0 << 9, 0 | 3 << 3 | 2 << 6 | 1 << 9, 4 | 7 << 3 | 6 << 6 | 5 << 9};
    const unsigned t = tab[f];
    const double A = pick(v, t & 7), B = pick(v, (t >> 3) & 7), C = pick(v, (t >> 6) & 7), D = pick(v, (t >> 9) & 7);
    const double acbd = A * C - B * D;
    if (fabs(acbd) < MC_EPS) return face >= 0;
    return face * A * acbd >= 0;
}

__device__ bool test_interior(const double v[8], int mc_case, int edge, int s) {
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0;
    if (mc_case == 4 || mc_case == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        if (edge < 0 || edge > 11) return s < 0;
        // reference edge P->Q and the three parallel edges B0->B1, C0->C1, D0->D1 (3 bits each)
        const unsigned char Pt[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
        const unsigned char Qt[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
        const unsigned char B0[12] = {3, 0, 1, 2, 7, 4, 5, 6, 3, 0, 1, 2};
        const unsigned char B1[12] = {2, 3, 0, 1, 6, 7, 4, 5, 7, 4, 5, 6};
        const unsigned char C0[12] = {7, 4, 5, 6, 3, 0, 1, 2, 2, 3, 0, 1};
        const unsigned char C1[12] = {6, 7, 4, 5, 2, 3, 0, 1, 6, 7, 4, 5};
        const unsigned char D0[12] = {4, 5, 6, 7, 0, 1, 2, 3, 1, 2, 3, 0};
        const unsigned char D1[12] = {5, 6, 7, 4, 1, 2, 3, 0, 5, 6, 7, 4};
        const double vp = pick(v, Pt[edge]), vq = pick(v, Qt[edge]);
        t = vp / (vp - vq + MC_EPS);
        const double b0 = pick(v, B0[edge]), c0 = pick(v, C0[edge]), d0 = pick(v, D0[edge]);
        Bt = b0 + (pick(v, B1[edge]) - b0) * t;
        Ct = c0 + (pick(v, C1[edge]) - c0) * t;
        Dt = d0 + (pick(v, D1[edge]) - d0) * t;
    }
    int test = 0;
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
        case 5: if (At * Ct - Bt * Dt < MC_EPS) return s > 0; break;
        case 10: if (At * Ct - Bt * Dt >= MC_EPS) return s > 0; break;
        case 7: case 11: case 13: case 14: case 15: return s < 0;
    }
    return false;   // scikit-image's answer for the two undecided patterns (see oracle/mc_lewiner.c)
}

#define OFF(name, cfg) (MC_OFF_##name + (cfg) * MC_ROW_##name)
#define OFF2(name, cfg, sub, len) (MC_OFF_##name + (cfg) * MC_ROW_##name + (sub) * (len))
#define TST(name, cfg, j) ((int)MC_LUT[MC_OFF_##name + (cfg) * MC_ROW_##name + (j)])

// Lewiner's case analysis: returns the offset of the triangle list in MC_LUT, sets ntri.
__device__ int classify_cell(const double v[8], int index, int &ntri) {
    const int cs = MC_LUT[MC_OFF_CASES + 2 * index], cf = MC_LUT[MC_OFF_CASES + 2 * index + 1];
    int sub = 0;
    switch (cs) {
        case 1: ntri = 1; return OFF(TILING1, cf);
        case 2: ntri = 2; return OFF(TILING2, cf);
        case 3:
            if (test_face(v, TST(TEST3, cf, 0))) { ntri = 4; return OFF(TILING3_2, cf); }
            ntri = 2; return OFF(TILING3_1, cf);
        case 4:
            if (test_interior(v, 4, -1, TST(TEST4, cf, 0))) { ntri = 2; return OFF(TILING4_1, cf); }
            ntri = 6; return OFF(TILING4_2, cf);
        case 5: ntri = 3; return OFF(TILING5, cf);
        case 6:
            if (test_face(v, TST(TEST6, cf, 0))) { ntri = 5; return OFF(TILING6_2, cf); }
            if (test_interior(v, 6, TST(TEST6, cf, 2), TST(TEST6, cf, 1))) { ntri = 3; return OFF(TILING6_1_1, cf); }
            ntri = 9; return OFF(TILING6_1_2, cf);
        case 7:
            if (test_face(v, TST(TEST7, cf, 0))) sub += 1;
            if (test_face(v, TST(TEST7, cf, 1))) sub += 2;
            if (test_face(v, TST(TEST7, cf, 2))) sub += 4;
            switch (sub) {
                case 0: ntri = 3; return OFF(TILING7_1, cf);
                case 1: ntri = 5; return OFF2(TILING7_2, cf, 0, 15);
                case 2: ntri = 5; return OFF2(TILING7_2, cf, 1, 15);
                case 3: ntri = 9; return OFF2(TILING7_3, cf, 0, 27);
                case 4: ntri = 5; return OFF2(TILING7_2, cf, 2, 15);
                case 5: ntri = 9; return OFF2(TILING7_3, cf, 1, 27);
                case 6: ntri = 9; return OFF2(TILING7_3, cf, 2, 27);
                default:
                    if (test_interior(v, 7, TST(TEST7, cf, 4), TST(TEST7, cf, 3))) { ntri = 9; return OFF(TILING7_4_2, cf); }
                    ntri = 5; return OFF(TILING7_4_1, cf);
            }
        case 8: ntri = 2; return OFF(TILING8, cf);
        case 9: ntri = 4; return OFF(TILING9, cf);
        case 10:
            if (test_face(v, TST(TEST10, cf, 0))) {
                if (test_face(v, TST(TEST10, cf, 1))) { ntri = 4; return OFF(TILING10_1_1_, cf); }
                ntri = 8; return OFF(TILING10_2, cf);
            }
            if (test_face(v, TST(TEST10, cf, 1))) { ntri = 8; return OFF(TILING10_2_, cf); }
            if (test_interior(v, 10, -1, TST(TEST10, cf, 2))) { ntri = 4; return OFF(TILING10_1_1, cf); }
            ntri = 8; return OFF(TILING10_1_2, cf);
        case 11: ntri = 4; return OFF(TILING11, cf);
        case 12:
            if (test_face(v, TST(TEST12, cf, 0))) {
                if (test_face(v, TST(TEST12, cf, 1))) { ntri = 4; return OFF(TILING12_1_1_, cf); }
                ntri = 8; return OFF(TILING12_2, cf);
            }
            if (test_face(v, TST(TEST12, cf, 1))) { ntri = 8; return OFF(TILING12_2_, cf); }
            if (test_interior(v, 12, TST(TEST12, cf, 3), TST(TEST12, cf, 2))) { ntri = 4; return OFF(TILING12_1_1, cf); }
            ntri = 8; return OFF(TILING12_1_2, cf);
        case 13: {
            for (int k = 0; k < 6; ++k)
                if (test_face(v, TST(TEST13, cf, k))) sub += 1 << k;
            const int sc = MC_LUT[MC_OFF_SUBCONFIG13 + sub];
            if (sc == 0) { ntri = 4; return OFF(TILING13_1, cf); }
            if (sc <= 6) { ntri = 6; return OFF2(TILING13_2, cf, sc - 1, 18); }
            if (sc <= 18) { ntri = 10; return OFF2(TILING13_3, cf, sc - 7, 30); }
            if (sc <= 22) { ntri = 12; return OFF2(TILING13_4, cf, sc - 19, 36); }
            if (sc <= 26) {
                const int e = MC_LUT[OFF2(TILING13_5_1, cf, sc - 23, 18)];
                if (test_interior(v, 13, e, TST(TEST13, cf, 6))) { ntri = 6; return OFF2(TILING13_5_1, cf, sc - 23, 18); }
                ntri = 10; return OFF2(TILING13_5_2, cf, sc - 23, 30);
            }
            if (sc <= 38) { ntri = 10; return OFF2(TILING13_3_, cf, sc - 27, 30); }
            if (sc <= 44) { ntri = 6; return OFF2(TILING13_2_, cf, sc - 39, 18); }
            if (sc == 45) { ntri = 4; return OFF(TILING13_1_, cf); }
            ntri = 0; return 0;
        }
        case 14: ntri = 4; return OFF(TILING14, cf);
        default: ntri = 0; return 0;
    }
}

// which of the 13 vertex slots (12 edges + centre) a cell owns: it is the first cell of the
// sweep that touches the edge
__device__ __forceinline__ unsigned own_mask(int x, int y, int z) {
    const unsigned X = x == 0, Y = y == 0, Z = z == 0;
    unsigned m = (1u << 6) | (1u << 5) | (1u << 10) | (1u << 12);
    m |= (Y & Z) << 0; m |= Z << 2; m |= Y << 4;          // x-edges 0,2,4 (6 always)
    m |= (X & Z) << 3; m |= Z << 1; m |= X << 7;          // y-edges 3,1,7 (5 always)
    m |= (X & Y) << 8; m |= Y << 9; m |= X << 11;         // z-edges 8,9,11 (10 always)
    return m;
}

__device__ __forceinline__ void cell_xyz(unsigned c, const McDims &d, int &x, int &y, int &z) {
    const unsigned t = c / (unsigned)d.c2;
    x = (int)(c - t * (unsigned)d.c2);
    z = (int)(t / (unsigned)d.c1);
    y = (int)(t - (unsigned)z * (unsigned)d.c1);
}

// (x,y,z) of cell `first + t` given (x0,y0,z0) of cell `first` (wave-uniform) and a small t:
// avoids two 32-bit integer divisions per thread
__device__ __forceinline__ void cell_from(int x0, int y0, int z0, unsigned t, const McDims &d, float inv_c2, int &x, int &y, int &z) {
    const unsigned u = (unsigned)x0 + t;                          // < c2 + 256: exact in float
    unsigned q = (unsigned)((float)u * inv_c2);
    int r = (int)(u - q * (unsigned)d.c2);
    if (r < 0) { --q; r += d.c2; }
    if (r >= d.c2) { ++q; r -= d.c2; }
    x = r;
    y = y0 + (int)q;
    z = z0;
    while (y >= d.c1) { y -= d.c1; ++z; }
}

// smallest float f with (double)f > level: then "(double)v - level > 0" is exactly "v >= f"
__device__ __forceinline__ float level_threshold(double level) {
    float t = (float)level;
    if (!((double)t > level)) t = nextafterf(t, INFINITY);
    return t;
}

// Surface cells are a minority (1-2 % for a trained field, ~16 % for noise) and scattered, so every
// wave would run the expensive branch with most lanes idle.  Each kernel therefore first COMPACTS its
// workgroup's active cells into an LDS list (ballot + prefix) and then processes the list densely.

// inclusive prefix over the wave on the DPP path (row shifts, then the row-15 / row-31 broadcasts): six adds, no LDS round trip
__device__ __forceinline__ unsigned wave_incl_scan_dpp(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2, 3
    return v;
}

// block-wide exclusive scan of (a,b) pairs over 256 threads; returns exclusive prefix, totals in tot
__device__ __forceinline__ uint2 block_exscan(unsigned a, unsigned b, uint2 *lds /*[4]*/, uint2 &tot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned ia = wave_incl_scan_dpp(a), ib = wave_incl_scan_dpp(b);
    if (lane == 63) lds[w] = make_uint2(ia, ib);
    __syncthreads();
    unsigned oa = 0, ob = 0, sa = 0, sb = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < w) { oa += lds[i].x; ob += lds[i].y; }
        sa += lds[i].x; sb += lds[i].y;
    }
    tot = make_uint2(sa, sb);
    __syncthreads();
    return make_uint2(oa + ia - a, ob + ib - b);
}

__device__ __forceinline__ unsigned block_compact(bool flag, unsigned payload, unsigned *list, unsigned *wave_cnt /*[4]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) wave_cnt[w] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int i = 0; i < 4; ++i) { if (i < w) base += wave_cnt[i]; total += wave_cnt[i]; }
    if (flag) list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = payload;
    __syncthreads();
    return total;
}

// One workgroup classifies CLS_CHUNKS consecutive 256-cell chunks (the chunk stays the unit of the count /
// offset tables the other kernels use).  The kernel is a chain of dependent memory round trips (header, volume,
// table look-ups), so its time is rounds x latency: four chunks per workgroup put four times the loads in
// flight per round and run the case analysis once over the combined active list.
constexpr int CLS_CHUNKS = 4;

__global__ void __launch_bounds__(CELLS_PER_BLOCK)
mc_classify_kernel(const float *vol, McDims d, McWs ws, double level_in, int auto_level, unsigned nblk, int npart) {
    __shared__ unsigned list[CLS_CHUNKS * CELLS_PER_BLOCK];
    __shared__ unsigned wave_cnt[CLS_CHUNKS * 4];
    __shared__ unsigned tot[CLS_CHUNKS][2];
    __shared__ int base_xyz[CLS_CHUNKS][3];
    const unsigned chunk0 = blockIdx.x * CLS_CHUNKS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the min/max partials sit in the (not yet written) block-offset table: the scan kernel fills it after this kernel
    const double level = iso_level(reinterpret_cast<const float2 *>(ws.boff), npart, level_in, auto_level);
    if (threadIdx.x < 2 * CLS_CHUNKS) tot[threadIdx.x >> 1][threadIdx.x & 1] = 0;
    const float inv_c2 = 1.0f / (float)d.c2;
    const float thr = level_threshold(level);
    const size_t s1 = (size_t)d.n2, s0 = (size_t)d.n1 * d.n2;
    // phase 1 (all cells): the 8-bit sign pattern with f32 compares against an exact threshold;
    // inactive cells are done.  All chunks' loads are issued before the first compare.
    typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
    f2 va[CLS_CHUNKS], vb[CLS_CHUNKS], ve[CLS_CHUNKS], vf[CLS_CHUNKS];
    bool inside[CLS_CHUNKS];
#pragma unroll
    for (int k = 0; k < CLS_CHUNKS; ++k) {
        const unsigned first = (chunk0 + k) * CELLS_PER_BLOCK, c = first + threadIdx.x;
        inside[k] = c < d.ncells;
        int bx0 = 0, by0 = 0, bz0 = 0;
        if (first < d.ncells) cell_xyz(first, d, bx0, by0, bz0);      // wave-uniform: scalar unit
        if (threadIdx.x == 0) { base_xyz[k][0] = bx0; base_xyz[k][1] = by0; base_xyz[k][2] = bz0; }
        va[k] = vb[k] = ve[k] = vf[k] = f2{0.0f, 0.0f};
        if (inside[k]) {
            int x, y, z;
            cell_from(bx0, by0, bz0, threadIdx.x, d, inv_c2, x, y, z);
            const float *p = vol + (size_t)z * s0 + (size_t)y * s1 + x;
            va[k] = *reinterpret_cast<const f2 *>(p);      vb[k] = *reinterpret_cast<const f2 *>(p + s1);
            ve[k] = *reinterpret_cast<const f2 *>(p + s0); vf[k] = *reinterpret_cast<const f2 *>(p + s0 + s1);
        }
    }
    bool active[CLS_CHUNKS];
    unsigned long long ball[CLS_CHUNKS];
#pragma unroll
    for (int k = 0; k < CLS_CHUNKS; ++k) {
        const f2 a = va[k], b = vb[k], e = ve[k], f = vf[k];
        const int index = (a.x >= thr) | (a.y >= thr) << 1 | (b.y >= thr) << 2 | (b.x >= thr) << 3 |
                          (e.x >= thr) << 4 | (e.y >= thr) << 5 | (f.y >= thr) << 6 | (f.x >= thr) << 7;
        active[k] = inside[k] && index != 0 && index != 255;
        if (inside[k] && !active[k]) ws.cnt[(chunk0 + k) * CELLS_PER_BLOCK + threadIdx.x] = 0;
        ball[k] = __ballot(active[k]);
        if (lane == 0) wave_cnt[k * 4 + w] = (unsigned)__popcll(ball[k]);
    }
    __syncthreads();
    // compact the active cells of all chunks into one list (chunk-major, then cell order)
    unsigned nact = 0;
    {
        unsigned run = 0, mine[CLS_CHUNKS];
#pragma unroll
        for (int i = 0; i < CLS_CHUNKS * 4; ++i) {
            if ((i & 3) == w) mine[i >> 2] = run;
            run += wave_cnt[i];
        }
        nact = run;
#pragma unroll
        for (int k = 0; k < CLS_CHUNKS; ++k)
            if (active[k]) list[mine[k] + (unsigned)__popcll(ball[k] & ((1ull << lane) - 1ull))] = (unsigned)k << 8 | threadIdx.x;
    }
    __syncthreads();
    // phase 2 (over the active list): Lewiner case analysis + owned-edge ranks.  The work is a chain of
    // dependent table look-ups (latency-bound), so the list is dealt round-robin over the block's four
    // waves: four times as many waves in flight beats packing it into one wave.
    for (unsigned slot = (threadIdx.x & 63u) * 4u + (threadIdx.x >> 6); slot < nact; slot += CELLS_PER_BLOCK) {
        const unsigned ent = list[slot], k = ent >> 8, t = ent & 255u;
        const unsigned cc = (chunk0 + k) * CELLS_PER_BLOCK + t;
        int x, y, z;
        cell_from(base_xyz[k][0], base_xyz[k][1], base_xyz[k][2], t, d, inv_c2, x, y, z);
        double v[8];
        load_cell(vol, d, x, y, z, level, v);
        int idx2 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) idx2 |= (v[q] > 0.0) << q;
        int nt = 0;
        const int off = classify_cell(v, idx2, nt);
        const unsigned own = own_mask(x, y, z);
        unsigned seen = 0, nnew = 0;
        uint64_t rk = 0;
        for (int i = 0; i < 3 * nt; ++i) {
            const int e = MC_LUT[off + i];
            if (!((seen >> e) & 1u)) {
                seen |= 1u << e;
                if ((own >> e) & 1u) { ++nnew; rk |= (uint64_t)nnew << (4 * e); }
            }
        }
        ws.desc[cc] = (uint32_t)off;
        ws.rank[cc] = rk;
        ws.cnt[cc] = (uint16_t)((unsigned)nt | (nnew << 8));
        atomicAdd(&tot[k][0], (unsigned)nt);
        atomicAdd(&tot[k][1], nnew);
    }
    __syncthreads();
    if (threadIdx.x < CLS_CHUNKS && chunk0 + threadIdx.x < nblk)
        ws.bsum[chunk0 + threadIdx.x] = make_uint2(tot[threadIdx.x][0], tot[threadIdx.x][1]);
    if (blockIdx.x == 0 && threadIdx.x == 0) ws.hdr->level = level;
}

// one block: exclusive scan of the per-block sums.  Every wave owns a contiguous 1/16 of the table and walks it in
// rows of 64 entries (one coalesced 512-byte load per row, SC_BATCH rows in flight): pass 1 adds its rows up for the
// wave's total, pass 2 -- after the 16 totals met in LDS -- re-reads them (L2) and writes the exclusive offsets, a
// DPP prefix per row plus a running carry.  (The table has 8 192 entries at 128^3 and 65 536 at 256^3; a thread
// walking its own 64 strided entries took 134 us there.)
constexpr int SC_BATCH = 16;

__global__ void __launch_bounds__(1024) mc_scan_kernel(McWs ws, unsigned nblk, McHeader *host_hdr, int host_seq, const int *seq_src) {
    __shared__ uint2 wtot[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned rows = ((nblk + 1023) / 1024), seg = rows * 64;            // rows per wave, entries per wave
    const unsigned lo = (unsigned)w * seg;
    unsigned a = 0, b = 0;
    // seq_src (captured scenes, vt_mc_count_echo): the number to write behind the counts is read from a page-locked word the host set
    // before the replay -- a kernel argument would be frozen into the graph.  Requested here, used at the end of the kernel.
    if (seq_src && threadIdx.x == 0) host_seq = __hip_atomic_load(seq_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (unsigned r0 = 0; r0 < rows; r0 += SC_BATCH) {
        uint2 v[SC_BATCH];
#pragma unroll
        for (int k = 0; k < SC_BATCH; ++k) {
            const unsigned i = lo + (r0 + k) * 64 + lane;
            v[k] = (r0 + k < rows && i < nblk) ? ws.bsum[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < SC_BATCH; ++k) { a += v[k].x; b += v[k].y; }
    }
    a = wave_incl_scan_dpp(a); b = wave_incl_scan_dpp(b);
    if (lane == 63) wtot[w] = make_uint2(a, b);
    __syncthreads();
    unsigned ca = 0, cb = 0, sa = 0, sb = 0;                                  // carry = totals of the waves before this one
    for (int i = 0; i < 16; ++i) {
        if (i < w) { ca += wtot[i].x; cb += wtot[i].y; }
        sa += wtot[i].x; sb += wtot[i].y;
    }
    for (unsigned r0 = 0; r0 < rows; r0 += SC_BATCH) {
        uint2 v[SC_BATCH];
#pragma unroll
        for (int k = 0; k < SC_BATCH; ++k) {
            const unsigned i = lo + (r0 + k) * 64 + lane;
            v[k] = (r0 + k < rows && i < nblk) ? ws.bsum[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < SC_BATCH; ++k) {
            const unsigned i = lo + (r0 + k) * 64 + lane;
            const unsigned ia = wave_incl_scan_dpp(v[k].x), ib = wave_incl_scan_dpp(v[k].y);
            if (r0 + k < rows && i < nblk) ws.boff[i] = make_uint2(ca + ia - v[k].x, cb + ib - v[k].y);
            ca += (unsigned)__builtin_amdgcn_readlane((int)ia, 63);
            cb += (unsigned)__builtin_amdgcn_readlane((int)ib, 63);
        }
    }
    if (threadIdx.x == 0) {
        ws.hdr->nfaces = (int)sa; ws.hdr->nverts = (int)sb;
        // vt_mc_count_notify: the counts (and the level the classify kernel left) straight into a page-locked host slot -- no copy command
        // between this kernel and the emit kernels behind it
        // between this kernel and the emit kernels behind it, and no event either: the host polls the sequence number, written last
        if (host_hdr) {
            host_hdr->nfaces = (int)sa; host_hdr->nverts = (int)sb; host_hdr->level = ws.hdr->level;
            __threadfence_system();
            __hip_atomic_store(&host_hdr->reserved[0], host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct McOut {
    float *verts; int max_verts;
    int *faces; int max_faces;
    int rescale; float shift, scale;
};

// The vertex kernel uses the same four-chunks-per-workgroup scheme as the classify kernel: the chunks' count
// loads are in flight together and the dependent tail (rank / volume look-ups) runs once per workgroup.
constexpr int EM_CHUNKS = 4;

__global__ void __launch_bounds__(CELLS_PER_BLOCK)
mc_vertices_kernel(const float *vol, McDims d, McWs ws, McOut o, unsigned nblk) {
    __shared__ unsigned wsum[EM_CHUNKS * 4];                       // vertices per (chunk, wave)
    __shared__ unsigned wave_cnt[EM_CHUNKS * 4];                   // vertex-owning cells per (chunk, wave)
    __shared__ unsigned list[EM_CHUNKS * CELLS_PER_BLOCK], lbase[EM_CHUNKS * CELLS_PER_BLOCK];
    __shared__ int base_xyz[EM_CHUNKS][3];
    const unsigned chunk0 = blockIdx.x * EM_CHUNKS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        // most workgroups see no surface at all: their chunks' (triangle, vertex) sums say so in one scalar load each
        unsigned any = 0;
#pragma unroll
        for (int k = 0; k < EM_CHUNKS; ++k)
            if (chunk0 + k < nblk) { const uint2 bs = ws.bsum[chunk0 + k]; any |= bs.x | bs.y; }
        if (!any) return;
    }
    const double level = ws.hdr->level;
    unsigned cn[EM_CHUNKS], boffv[EM_CHUNKS];
#pragma unroll
    for (int k = 0; k < EM_CHUNKS; ++k) {
        const unsigned c = (chunk0 + k) * CELLS_PER_BLOCK + threadIdx.x;
        cn[k] = (c < d.ncells) ? ws.cnt[c] : 0;
        boffv[k] = (chunk0 + k < nblk) ? ws.boff[chunk0 + k].y : 0;
        if (threadIdx.x == 0) {
            int bx0 = 0, by0 = 0, bz0 = 0;
            if ((chunk0 + k) * CELLS_PER_BLOCK < d.ncells) cell_xyz((chunk0 + k) * CELLS_PER_BLOCK, d, bx0, by0, bz0);
            base_xyz[k][0] = bx0; base_xyz[k][1] = by0; base_xyz[k][2] = bz0;
        }
    }
    unsigned incl[EM_CHUNKS];
    unsigned long long ball[EM_CHUNKS];
#pragma unroll
    for (int k = 0; k < EM_CHUNKS; ++k) {
        incl[k] = wave_incl_scan_dpp(cn[k] >> 8);
        ball[k] = __ballot((cn[k] >> 8) != 0);
        if (lane == 63) wsum[k * 4 + w] = incl[k];
        if (lane == 0) wave_cnt[k * 4 + w] = (unsigned)__popcll(ball[k]);
    }
    __syncthreads();
    unsigned nact = 0;
    {
        unsigned run = 0, mine[EM_CHUNKS];
#pragma unroll
        for (int i = 0; i < EM_CHUNKS * 4; ++i) {
            if ((i & 3) == w) mine[i >> 2] = run;
            run += wave_cnt[i];
        }
        nact = run;
#pragma unroll
        for (int k = 0; k < EM_CHUNKS; ++k) {
            unsigned before = 0;
            for (int i = 0; i < 4; ++i) if (i < w) before += wsum[k * 4 + i];
            const unsigned nv = cn[k] >> 8, vb = boffv[k] + before + incl[k] - nv;     // first vertex id of the cell
            if (cn[k]) ws.vbase[(chunk0 + k) * CELLS_PER_BLOCK + threadIdx.x] = vb;
            if (nv) {
                const unsigned pos = mine[k] + (unsigned)__popcll(ball[k] & ((1ull << lane) - 1ull));
                list[pos] = (unsigned)k << 8 | threadIdx.x;
                lbase[pos] = vb;
            }
        }
    }
    __syncthreads();
    const float inv_c2 = 1.0f / (float)d.c2;
    for (unsigned slot = (threadIdx.x & 63u) * 4u + (threadIdx.x >> 6); slot < nact; slot += CELLS_PER_BLOCK) {   // round-robin over the 4 waves
        const unsigned ent = list[slot], k = ent >> 8, li = ent & 255u;
        const unsigned cc = (chunk0 + k) * CELLS_PER_BLOCK + li, vbase = lbase[slot];
        int x, y, z;
        cell_from(base_xyz[k][0], base_xyz[k][1], base_xyz[k][2], li, d, inv_c2, x, y, z);
        double v[8];
        load_cell(vol, d, x, y, z, level, v);
        const uint64_t rk = ws.rank[cc];
        for (int e = 0; e < 13; ++e) {
            const unsigned nib = (unsigned)(rk >> (4 * e)) & 15u;
            if (!nib) continue;
            double fx = 0, fy = 0, fz = 0, ff = 0;
            if (e == 12) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double wq = 1.0 / (MC_EPS + fabs(v[q]));
                    fx += cx(q) * wq; fy += cy(q) * wq; fz += cz(q) * wq; ff += wq;
                }
            } else {
                const int a = (e < 8) ? e : e - 8;                       // E0
                const int b = (e < 8) ? ((e & 4) | ((e + 1) & 3)) : e - 4;  // E1
                const double wa = 1.0 / (MC_EPS + fabs(pick(v, a)));
                const double wb = 1.0 / (MC_EPS + fabs(pick(v, b)));
                fx += cx(a) * wa; fy += cy(a) * wa; fz += cz(a) * wa; ff += wa;
                fx += cx(b) * wb; fy += cy(b) * wb; fz += cz(b) * wb; ff += wb;
            }
            const unsigned vid = vbase + nib - 1;
            if (vid < (unsigned)o.max_verts) {
                // array-axis order (axis0, axis1, axis2) = (z, y, x), as skimage returns
                float p0 = (float)((double)z + fz / ff), p1 = (float)((double)y + fy / ff), p2 = (float)((double)x + fx / ff);
                if (o.rescale) { p0 = (p0 - o.shift) * o.scale; p1 = (p1 - o.shift) * o.scale; p2 = (p2 - o.shift) * o.scale; }
                float *dst = o.verts + (size_t)vid * 3;
                dst[0] = p0; dst[1] = p1; dst[2] = p2;
            }
        }
    }
}

// (one 256-cell chunk per workgroup here: the entry loop below is itself a chain of dependent look-ups per
// iteration, and four chunks per workgroup would run four times as many iterations back to back -- measured slower)
// one thread per (active cell, triangle corner): cells carry 1..12 triangles, so a thread per cell
// would leave most lanes waiting for the few 12-triangle cells
__global__ void __launch_bounds__(CELLS_PER_BLOCK)
mc_faces_kernel(McDims d, McWs ws, McOut o) {
    __shared__ uint2 red[4];
    __shared__ unsigned list[CELLS_PER_BLOCK], lpre[CELLS_PER_BLOCK];
    __shared__ unsigned wave_cnt[4];
    if (ws.bsum[blockIdx.x].x == 0) return;                       // no triangle in this chunk (one scalar load)
    const unsigned c = blockIdx.x * CELLS_PER_BLOCK + threadIdx.x;
    const unsigned cn = (c < d.ncells) ? ws.cnt[c] : 0;
    uint2 tot;
    const uint2 pre = block_exscan(cn & 0xff, cn >> 8, red, tot);
    // active list entry k: (local cell id, first triangle of the cell within the block)
    const unsigned nact = block_compact((cn & 0xff) != 0, threadIdx.x | pre.x << 8, list, wave_cnt);
    if (threadIdx.x < nact) lpre[threadIdx.x] = list[threadIdx.x] >> 8;
    __syncthreads();
    const unsigned nent = 3u * tot.x, tb0 = ws.boff[blockIdx.x].x;
    int bx0, by0, bz0;
    cell_xyz(blockIdx.x * CELLS_PER_BLOCK, d, bx0, by0, bz0);
    const float inv_c2 = 1.0f / (float)d.c2;
    for (unsigned ent = threadIdx.x; ent < nent; ent += CELLS_PER_BLOCK) {
        const unsigned tri = ent / 3u;
        // the active cell whose triangle range contains `tri`: last k with lpre[k] <= tri
        unsigned lo = 0, hi = nact;
        while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (lpre[mid] <= tri) lo = mid; else hi = mid; }
        const unsigned li = list[lo] & 0xff, cc = blockIdx.x * CELLS_PER_BLOCK + li;
        const unsigned i = ent - 3u * lpre[lo];                    // entry index inside the cell's list
        int x, y, z;
        cell_from(bx0, by0, bz0, li, d, inv_c2, x, y, z);
        const int e = MC_LUT[ws.desc[cc] + i];
        int ox = x, oy = y, oz = z, el = 12;
        if (e < 12) {
            // edge base point relative to the cell, and direction
            const int dir = (e >= 8) ? 2 : (e & 1);                 // 0: x, 1: y, 2: z
            int bx, by, bz;
            if (e >= 8) { const int k = e - 8; bx = (k == 1) | (k == 2); by = (k >> 1) & 1; bz = 0; }
            else { const int k = e & 3; bx = (k == 1); by = (k == 2); bz = e >> 2; }
            const int gx = x + bx, gy = y + by, gz = z + bz;
            if (dir == 0) {
                ox = gx; oy = max(gy - 1, 0); oz = max(gz - 1, 0);
                el = 2 * (gy - oy) + 4 * (gz - oz);                  // 0,2,4,6
            } else if (dir == 1) {
                ox = max(gx - 1, 0); oy = gy; oz = max(gz - 1, 0);
                const int dx = gx - ox, dz = gz - oz;
                el = (dx ? 1 : 3) + 4 * dz;                          // 3,1,7,5
            } else {
                ox = max(gx - 1, 0); oy = max(gy - 1, 0); oz = gz;
                const int dx = gx - ox, dy = gy - oy;
                el = 8 + (dy ? (dx ? 2 : 3) : dx);                   // 8,9,11,10
            }
        }
        const unsigned oc = ((unsigned)oz * (unsigned)d.c1 + (unsigned)oy) * (unsigned)d.c2 + (unsigned)ox;
        const unsigned nib = (unsigned)(ws.rank[oc] >> (4 * el)) & 15u;
        const unsigned vid = ws.vbase[oc] + nib - 1;
        const size_t fi = (size_t)tb0 * 3 + ent;
        if (fi < (size_t)o.max_faces * 3) o.faces[fi] = (int)vid;
    }
}

__host__ size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__host__ bool mc_layout(int n0, int n1, int n2, McDims &d, size_t off[8], size_t &total, unsigned &nblk) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return false;
    const uint64_t nc = (uint64_t)(n0 - 1) * (n1 - 1) * (n2 - 1);
    if (nc >= (1ull << 31)) return false;
    d.n0 = n0; d.n1 = n1; d.n2 = n2; d.c1 = n1 - 1; d.c2 = n2 - 1; d.ncells = (unsigned)nc;
    nblk = (unsigned)((nc + CELLS_PER_BLOCK - 1) / CELLS_PER_BLOCK);
    size_t p = 0;
    off[0] = p; p = align_up(p + sizeof(McHeader), 256);
    off[1] = p; p = align_up(p + nc * sizeof(uint16_t), 256);
    off[2] = p; p = align_up(p + nc * sizeof(uint32_t), 256);
    off[3] = p; p = align_up(p + nc * sizeof(uint64_t), 256);
    off[4] = p; p = align_up(p + nc * sizeof(uint32_t), 256);
    off[5] = p; p = align_up(p + (size_t)nblk * sizeof(uint2), 256);
    off[6] = p; p = align_up(p + (size_t)nblk * sizeof(uint2), 256);
    total = p;
    return true;
}

__host__ McWs mc_ws(void *base, const size_t off[8]) {
    char *b = (char *)base;
    McWs w;
    w.hdr = (McHeader *)(b + off[0]); w.cnt = (uint16_t *)(b + off[1]); w.desc = (uint32_t *)(b + off[2]);
    w.rank = (uint64_t *)(b + off[3]); w.vbase = (uint32_t *)(b + off[4]);
    w.bsum = (uint2 *)(b + off[5]); w.boff = (uint2 *)(b + off[6]);
    return w;
}

}  // namespace

extern "C" {

size_t vt_mc_workspace_bytes(int n0, int n1, int n2) {
    McDims d; size_t off[8], total; unsigned nblk;
    if (!mc_layout(n0, n1, n2, d, off, total, nblk)) return 0;
    return total;
}

static int mc_count_impl(const float *vol, int n0, int n1, int n2, double level, int auto_level,
                         void *workspace, size_t workspace_bytes, void *stream, McHeader *host_hdr, int host_seq, const int *seq_src = nullptr) {
    if (!vol || !workspace) return vt_fail(VT_ERR_INVALID, "vt_mc_count: null argument");
    McDims d; size_t off[8], total; unsigned nblk;
    if (!mc_layout(n0, n1, n2, d, off, total, nblk))
        return vt_fail(VT_ERR_INVALID, "vt_mc_count: volume must be at least 2x2x2 (and < 2^31 cells)");
    if (workspace_bytes < total) return vt_fail(VT_ERR_WORKSPACE, "vt_mc_count: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    McWs ws = mc_ws(workspace, off);
    unsigned g = 0;
    if (auto_level) {
        const size_t n = (size_t)n0 * n1 * n2;
        g = (unsigned)((n / 4 + 256 * 8 - 1) / (256 * 8));
        if (g > MC_MM_BLOCKS) g = MC_MM_BLOCKS;
        if (g > nblk) g = nblk;                                    // the partials borrow the block-offset table
        if (g < 1) g = 1;
        hipLaunchKernelGGL(mc_minmax_kernel, dim3(g), dim3(256), 0, s, vol, n, reinterpret_cast<float2 *>(ws.boff));
    }
    hipLaunchKernelGGL(mc_classify_kernel, dim3((nblk + CLS_CHUNKS - 1) / CLS_CHUNKS), dim3(CELLS_PER_BLOCK), 0, s,
                       vol, d, ws, level, auto_level, nblk, (int)g);
    hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, s, ws, nblk, host_hdr, host_seq, seq_src);
    return vt_check(hipGetLastError(), "vt_mc_count");
}

int vt_mc_count(const float *vol, int n0, int n1, int n2, double level, int auto_level,
                void *workspace, size_t workspace_bytes, void *stream) {
    return mc_count_impl(vol, n0, n1, n2, level, auto_level, workspace, workspace_bytes, stream, nullptr, 0);
}

int vt_mc_read_counts(const void *workspace, int *nverts_host, int *nfaces_host, double *level_host, void *stream) {
    if (!workspace || !nverts_host || !nfaces_host) return vt_fail(VT_ERR_INVALID, "vt_mc_read_counts: null argument");
    McHeader h;
    hipError_t e = hipMemcpyAsync(&h, workspace, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return vt_check(e, "vt_mc_read_counts");
    *nverts_host = h.nverts; *nfaces_host = h.nfaces;
    if (level_host) *level_host = h.level;
    return 0;
}

// The same read split in two, so that work launched in between overlaps it: _begin copies the header into a page-locked slot and
// records an event right behind the copy; _end waits for THAT event only (not for what was launched after it -- the speculative
// vertex / face kernels of a marching-cubes call) and returns the counts.  A ring of 16 slots; a token is valid for one _end.
namespace {
constexpr int MC_SLOTS = 16;
struct McSlot { McHeader *host; hipEvent_t ev; bool made, busy; int seq, polled; hipStream_t stream; };   // polled: the scan kernel writes `seq` into the slot (vt_mc_count_notify, on `stream`)
McSlot mc_slots[MC_SLOTS];                              // zero-initialised: nothing made, nothing busy
unsigned mc_slot_next = 0;
std::mutex mc_slots_mutex;                              // guards the slots' creation and their busy flags
}  // namespace

// the next page-locked slot no read-back is waiting in (made on first use), marked busy
static int mc_slot_acquire(int &t) {
    t = -1;
    std::lock_guard<std::mutex> lock(mc_slots_mutex);
    for (int i = 0; i < MC_SLOTS && t < 0; ++i) {
        const int c = (int)((mc_slot_next + i) % MC_SLOTS);
        if (!mc_slots[c].busy) t = c;
    }
    if (t < 0) return vt_fail(VT_ERR_INVALID, "vt_mc_read_counts_begin: 16 read-backs are already in flight (call vt_mc_read_counts_end)");
    McSlot &sl = mc_slots[t];
    if (!sl.made) {
        hipError_t e = sl.host ? hipSuccess : hipHostMalloc(reinterpret_cast<void **>(&sl.host), sizeof(McHeader), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming);
        if (e != hipSuccess) return vt_check(e, "vt_mc_read_counts_begin: page-locked slot");
        memset(sl.host, 0, sizeof(McHeader));               // the polled sequence number starts at 0 (no launch ever writes 0)
        sl.made = true;
    }
    sl.busy = true;
    mc_slot_next = (unsigned)t + 1;
    return 0;
}

int vt_mc_count_notify(const float *vol, int n0, int n1, int n2, double level, int auto_level,
                       void *workspace, size_t workspace_bytes, void *stream, int *token) {
    if (!token) return vt_fail(VT_ERR_INVALID, "vt_mc_count_notify: null argument");
    int t;
    int rc = mc_slot_acquire(t);
    if (rc) return rc;
    McSlot &sl = mc_slots[t];
    sl.seq = sl.seq == 0x7fffffff ? 1 : sl.seq + 1;             // never 0: a fresh slot reads 0
    sl.polled = 1;
    sl.stream = (hipStream_t)stream;
    rc = mc_count_impl(vol, n0, n1, n2, level, auto_level, workspace, workspace_bytes, stream, sl.host, sl.seq);
    if (rc) {
        std::lock_guard<std::mutex> lock(mc_slots_mutex);
        mc_slots[t].busy = false;
        return rc;
    }
    *token = t;
    return 0;
}

int vt_mc_read_counts_begin(const void *workspace, void *stream, int *token) {
    if (!workspace || !token) return vt_fail(VT_ERR_INVALID, "vt_mc_read_counts_begin: null argument");
    int t;
    const int arc = mc_slot_acquire(t);
    if (arc) return arc;
    mc_slots[t].polled = 0;
    hipError_t e = hipMemcpyAsync(mc_slots[t].host, workspace, sizeof(McHeader), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventRecord(mc_slots[t].ev, (hipStream_t)stream);
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> lock(mc_slots_mutex);
        mc_slots[t].busy = false;
        return vt_check(e, "vt_mc_read_counts_begin");
    }
    *token = t;
    return 0;
}

int vt_mc_read_counts_end(int token, int *nverts_host, int *nfaces_host, double *level_host) {
    if (token < 0 || token >= MC_SLOTS || !nverts_host || !nfaces_host)
        return vt_fail(VT_ERR_INVALID, "vt_mc_read_counts_end: bad token or null argument");
    {
        std::lock_guard<std::mutex> lock(mc_slots_mutex);
        if (!mc_slots[token].busy) return vt_fail(VT_ERR_INVALID, "vt_mc_read_counts_end: no read-back is in flight under this token");
    }
    hipError_t e = hipSuccess;
    if (mc_slots[token].polled) {
        // the scan kernel stores the sequence number behind the counts (system-scope release): spin on it -- a stream wait would need
        // an event packet between the scan and the emit kernels (~5 us of the ~85 the extraction takes)
        volatile int *seq = &mc_slots[token].host->reserved[0];
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != mc_slots[token].seq) {
            if ((++spins & 0xfffu) == 0) {
                if (hipPeekAtLastError() != hipSuccess) { e = hipGetLastError(); break; }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                    // a long backlog in front of the scan kernel (or a paused process): stop spinning and WAIT for the stream --
                    // the slot must not be handed out again while that kernel can still write into it
                    e = hipStreamSynchronize(mc_slots[token].stream);
                    if (e == hipSuccess && __atomic_load_n(seq, __ATOMIC_ACQUIRE) != mc_slots[token].seq) e = hipErrorUnknown;
                    break;
                }
            }
        }
        mc_slots[token].polled = 0;
    } else e = hipEventSynchronize(mc_slots[token].ev);
    if (e == hipSuccess) {
        *nverts_host = mc_slots[token].host->nverts; *nfaces_host = mc_slots[token].host->nfaces;
        if (level_host) *level_host = mc_slots[token].host->level;
    }
    {
        std::lock_guard<std::mutex> lock(mc_slots_mutex);
        mc_slots[token].busy = false;                       // the token is spent either way
    }
    return e == hipSuccess ? 0 : vt_check(e, "vt_mc_read_counts_end");
}

// ---- the counts of a CAPTURED scene without a copy between the scan and the emit kernels ---------------------------------------
// vt_mc_count_notify's sequence number is a kernel argument: frozen at capture.  Here the host owns a page-locked word per slot
// (`want`), sets it to a fresh number before every replay (vt_mc_echo_arm), the scan kernel of the replay reads it and writes it
// behind the counts into the slot's header, and vt_mc_echo_wait spins until the header carries that number.  The slot is made OUTSIDE
// the capture (page-locked allocation) and lives as long as the graph; one replay in flight per slot.
namespace {
constexpr int MC_ECHO_SLOTS = 64;
struct McEcho { McHeader *host; int *want; int nonce; bool made; bool free_; };
McEcho mc_echo[MC_ECHO_SLOTS];
int mc_echo_made = 0;
}  // namespace

int vt_mc_echo_slot(int *token) {
    if (!token) return vt_fail(VT_ERR_INVALID, "vt_mc_echo_slot: null argument");
    std::lock_guard<std::mutex> lock(mc_slots_mutex);
    for (int i = 0; i < mc_echo_made; ++i)
        if (mc_echo[i].free_) {                                     // a released slot (its graph is gone): the page-locked block is reused
            McEcho &sl = mc_echo[i];
            memset(sl.host, 0, sizeof(McHeader) + 64);
            sl.nonce = 0; sl.free_ = false;
            *token = i;
            return 0;
        }
    if (mc_echo_made >= MC_ECHO_SLOTS) return vt_fail(VT_ERR_INVALID, "vt_mc_echo_slot: all 64 slots are taken (one per captured scene shape)");
    McEcho &sl = mc_echo[mc_echo_made];
    char *mem = nullptr;
    const hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&mem), sizeof(McHeader) + 64, hipHostMallocDefault);
    if (e != hipSuccess) return vt_check(e, "vt_mc_echo_slot: page-locked slot");
    memset(mem, 0, sizeof(McHeader) + 64);
    sl.host = reinterpret_cast<McHeader *>(mem);
    sl.want = reinterpret_cast<int *>(mem + sizeof(McHeader));
    sl.nonce = 0; sl.made = true; sl.free_ = false;
    *token = mc_echo_made++;
    return 0;
}

int vt_mc_echo_release(int token) {
    std::lock_guard<std::mutex> lock(mc_slots_mutex);
    if (token < 0 || token >= mc_echo_made || mc_echo[token].free_) return vt_fail(VT_ERR_INVALID, "vt_mc_echo_release: bad token");
    mc_echo[token].free_ = true;
    return 0;
}

int vt_mc_count_echo(const float *vol, int n0, int n1, int n2, double level, int auto_level,
                     void *workspace, size_t workspace_bytes, void *stream, int token) {
    if (token < 0 || token >= mc_echo_made) return vt_fail(VT_ERR_INVALID, "vt_mc_count_echo: bad token");
    return mc_count_impl(vol, n0, n1, n2, level, auto_level, workspace, workspace_bytes, stream, mc_echo[token].host, 0, mc_echo[token].want);
}

int vt_mc_echo_arm(int token) {
    if (token < 0 || token >= mc_echo_made) return vt_fail(VT_ERR_INVALID, "vt_mc_echo_arm: bad token");
    McEcho &sl = mc_echo[token];
    sl.nonce = sl.nonce == 0x7fffffff ? 1 : sl.nonce + 1;           // never 0: a fresh slot reads 0
    __atomic_store_n(sl.want, sl.nonce, __ATOMIC_RELEASE);
    return 0;
}

int vt_mc_echo_wait(int token, void *stream, int *nverts_host, int *nfaces_host, double *level_host) {
    if (token < 0 || token >= mc_echo_made || !nverts_host || !nfaces_host) return vt_fail(VT_ERR_INVALID, "vt_mc_echo_wait: bad token or null argument");
    McEcho &sl = mc_echo[token];
    volatile int *seq = &sl.host->reserved[0];
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    hipError_t e = hipSuccess;
    while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != sl.nonce) {
        if ((++spins & 0xfffu) == 0) {
            if (hipPeekAtLastError() != hipSuccess) { e = hipGetLastError(); break; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                e = hipStreamSynchronize((hipStream_t)stream);          // a long backlog (or a paused process): wait for the stream itself
                if (e == hipSuccess && __atomic_load_n(seq, __ATOMIC_ACQUIRE) != sl.nonce) e = hipErrorUnknown;
                break;
            }
        }
    }
    if (e != hipSuccess) return vt_check(e, "vt_mc_echo_wait");
    *nverts_host = sl.host->nverts; *nfaces_host = sl.host->nfaces;
    if (level_host) *level_host = sl.host->level;
    return 0;
}

int vt_mc_emit(const float *vol, int n0, int n1, int n2, void *workspace,
               float *verts, int max_verts, int *faces, int max_faces,
               int rescale, float shift, float scale, void *stream) {
    if (!vol || !workspace || !verts || !faces) return vt_fail(VT_ERR_INVALID, "vt_mc_emit: null argument");
    if (max_verts < 0 || max_faces < 0) return vt_fail(VT_ERR_INVALID, "vt_mc_emit: negative capacity");
    McDims d; size_t off[8], total; unsigned nblk;
    if (!mc_layout(n0, n1, n2, d, off, total, nblk)) return vt_fail(VT_ERR_INVALID, "vt_mc_emit: bad volume shape");
    McWs ws = mc_ws(workspace, off);
    McOut o; o.verts = verts; o.max_verts = max_verts; o.faces = faces; o.max_faces = max_faces;
    o.rescale = rescale; o.shift = shift; o.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    const unsigned ngrp = (nblk + EM_CHUNKS - 1) / EM_CHUNKS;
    hipLaunchKernelGGL(mc_vertices_kernel, dim3(ngrp), dim3(CELLS_PER_BLOCK), 0, s, vol, d, ws, o, nblk);
    hipLaunchKernelGGL(mc_faces_kernel, dim3(nblk), dim3(CELLS_PER_BLOCK), 0, s, d, ws, o);
    return vt_check(hipGetLastError(), "vt_mc_emit");
}

}  // extern "C"
