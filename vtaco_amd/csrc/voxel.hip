// PointNet local-pool voxeliser for gfx950: voxel ids (K1), per-voxel channel-wise max
// gathered back to the points (K2) and per-voxel mean into the dense feature grid (K3),
// forward and backward.  Replaces torch_scatter.scatter_max + gather and scatter_mean at
// reference src/encoder/pointnet.py:116-132 and :102-110, plus the index math of
// src/common.py:293-309, 333-348.
//
// No dense [C, R^3] transient and no float atomics: one workgroup per scene radix-sorts the
// points by voxel id in LDS once per forward (the ids do not change between the
// four pooling rounds), which gives every point the contiguous range of its voxel-mates
// in sorted order.  Reductions then run over those ranges in ascending point order, so
// max/argmax and the mean are bit-reproducible run to run (the sum order equals the
// sequential scatter_add order of the CPU path).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "vt_common.h"

namespace {

constexpr int SORT_THREADS = 1024;
constexpr int MAX_T = 8192;          // points per scene the in-LDS sort covers

// reference src/common.py:293-309 + :333-348 in f32, truncating cast
// (planes: src/common.py:268-291, divisor 1 + padding + 10e-6 and upper clamp 1 - 10e-6 instead)
__device__ __forceinline__ int voxel_coord(float v, float divisor, float clamp_hi, int R) {
    float q = v / divisor + 0.5f;
    q = (q >= 1.0f) ? clamp_hi : q;
    q = (q < 0.0f) ? 0.0f : q;
    return (int)(q * (float)R);
}

// LDS image of the sort: cell ids by point, two permutation buffers, the (digit, chunk) count table
constexpr int RADIX_BITS = 6, RADIX = 1 << RADIX_BITS;
constexpr int MAX_CHUNKS = MAX_T / 64;                                  // a chunk = the 64 elements one wave ranks at a time
constexpr size_t SORT_LDS = (size_t)MAX_T * 4 + 2 * (size_t)MAX_T * 2 + (size_t)RADIX * MAX_CHUNKS * 2;     // 80 KB

// Stable LSD radix sort of the points by cell id, six bits per pass (three passes for 64^3 cells): the points start in
// ascending point order, so a stable sort by cell gives (cell, point) order.  Per pass every wave ranks its 64 elements
// among equal digits with six ballots (no atomics: the first lane of each digit class writes the class size into the
// (digit, chunk) table), one block-wide exclusive scan of the table in (digit, chunk) order turns the sizes into
// offsets, and the elements scatter to offset + rank.  Deterministic; 3 passes x ~3 us instead of 78 bitonic passes.
constexpr int FLAG_MAX_TILES = 4096;                               // 8^3 blocks of a volume of up to 128^3 voxels
// `tile_flags` (or null; volumes only): [B][(R/8)^3] bytes, bit 0 = no point of the scene in the 10^3 halo of that 8^3 block, bit 1 = none
// in its 12^3 halo either (vt_voxel_tile_flags)
__global__ void __launch_bounds__(SORT_THREADS)
voxel_build_kernel(const float *pts, int T, int nbits, int R, float divisor, float clamp_hi, int a0, int a1, int a2,
                   int *idx, int *order, int *seg_lo, int *seg_hi, int B, uint4 *fill, size_t fill16, unsigned char *tile_flags, int planes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    __shared__ unsigned char tflag[FLAG_MAX_TILES], tflag2[FLAG_MAX_TILES];
    // `planes` (vt_plane_build_multi): n | id_0 << 4 | id_1 << 6 | ...; workgroup pl * (B / n) + scene sorts the scene's points by the
    // cells of plane id_pl into the pl-th [B / n][T] slice of the outputs -- the planes of a scene side by side instead of one after the other
    int b = blockIdx.x;
    if (planes && (int)blockIdx.x < B) {
        const int scenes = B / (planes & 15), pl = blockIdx.x / scenes, plane = (planes >> (4 + 2 * pl)) & 3;
        b = blockIdx.x - pl * scenes;
        a0 = plane == 2 ? 1 : 0; a1 = plane == 1 ? 1 : 2;
        const size_t off = (size_t)pl * scenes * T;
        idx += off; order += off; seg_lo += off; seg_hi += off;
    }
    if ((int)blockIdx.x >= B) {
        // the workgroups behind the B sorting ones clear a buffer of the caller's (the grid the scatter-mean fills next): the sort keeps
        // one CU per scene busy for ~18 us, the other CUs stream 33 MB of zeros in that time instead of in a launch of their own
        const uint4 z = {0u, 0u, 0u, 0u};
        for (size_t i = (size_t)(blockIdx.x - B) * SORT_THREADS + threadIdx.x; i < fill16; i += (size_t)(gridDim.x - B) * SORT_THREADS) fill[i] = z;
        return;
    }
    unsigned *ids = reinterpret_cast<unsigned *>(sort_lds);                         // [MAX_T]
    unsigned short *perm_a = reinterpret_cast<unsigned short *>(ids + MAX_T);       // [MAX_T]
    unsigned short *perm_b = perm_a + MAX_T;                                         // [MAX_T]
    unsigned short *table = perm_b + MAX_T;                                          // [RADIX][nchunk]
    __shared__ unsigned wave_tot[SORT_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *p = pts + (size_t)b * T * 3;
    const int nround = (T + SORT_THREADS - 1) / SORT_THREADS, nchunk = nround * (SORT_THREADS / 64);
    const int nt1 = R >> 3;
    if (tile_flags) {
        for (int t = tid; t < nt1 * nt1 * nt1; t += SORT_THREADS) { tflag[t] = 1; tflag2[t] = 2; }
        __syncthreads();
    }
    for (int t = tid; t < T; t += SORT_THREADS) {
        // cell id = i(a0) + R * (i(a1) + R * i(a2)); a plane has no third axis (a2 < 0)
        const int ix = voxel_coord(p[3 * t + a0], divisor, clamp_hi, R);
        const int iy = voxel_coord(p[3 * t + a1], divisor, clamp_hi, R);
        const int iz = a2 >= 0 ? voxel_coord(p[3 * t + a2], divisor, clamp_hi, R) : 0;
        const int id = ix + R * (iy + R * iz);
        idx[(size_t)b * T + t] = id;
        ids[t] = (unsigned)id;
        perm_a[t] = (unsigned short)t;
        if (tile_flags) {                                           // the blocks whose halo [8 t - 1, 8 t + 8] holds the voxel
            for (int tz = max((iz - 1) >> 3, 0); tz <= min((iz + 1) >> 3, nt1 - 1); ++tz)
                for (int ty = max((iy - 1) >> 3, 0); ty <= min((iy + 1) >> 3, nt1 - 1); ++ty)
                    for (int tx = max((ix - 1) >> 3, 0); tx <= min((ix + 1) >> 3, nt1 - 1); ++tx) tflag[(tz * nt1 + ty) * nt1 + tx] = 0;
            for (int tz = max((iz - 2) >> 3, 0); tz <= min((iz + 2) >> 3, nt1 - 1); ++tz)      // ... whose halo [8 t - 2, 8 t + 9] does
                for (int ty = max((iy - 2) >> 3, 0); ty <= min((iy + 2) >> 3, nt1 - 1); ++ty)
                    for (int tx = max((ix - 2) >> 3, 0); tx <= min((ix + 2) >> 3, nt1 - 1); ++tx) tflag2[(tz * nt1 + ty) * nt1 + tx] = 0;
        }
    }
    __syncthreads();
    if (tile_flags)
        for (int t = tid; t < nt1 * nt1 * nt1; t += SORT_THREADS) tile_flags[(size_t)b * nt1 * nt1 * nt1 + t] = tflag[t] | tflag2[t];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int shift = 0; shift < nbits; shift += RADIX_BITS) {
        for (int e = tid; e < RADIX * nchunk; e += SORT_THREADS) table[e] = 0;
        __syncthreads();
        unsigned rank[MAX_T / SORT_THREADS], dig[MAX_T / SORT_THREADS], val[MAX_T / SORT_THREADS];
#pragma unroll
        for (int r = 0; r < MAX_T / SORT_THREADS; ++r) {
            if (r >= nround) break;
            const int i = r * SORT_THREADS + tid;
            const bool valid = i < T;
            const unsigned v = valid ? perm_a[i] : 0u;
            const unsigned d = valid ? (ids[v] >> shift) & (RADIX - 1) : 0u;
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < RADIX_BITS; ++bit) {
                const unsigned long long ones = __ballot((d >> bit) & 1u);
                same &= ((d >> bit) & 1u) ? ones : ~ones;
            }
            val[r] = v; dig[r] = d;
            rank[r] = (unsigned)__popcll(same & lt);
            if (valid && rank[r] == 0) table[d * nchunk + r * (SORT_THREADS / 64) + wave] = (unsigned short)__popcll(same);
        }
        __syncthreads();
        // exclusive scan of the table (digit-major, chunk order inside a digit): RADIX * nchunk <= 8192 entries, 8 per thread
        {
            const int per = (RADIX * nchunk + SORT_THREADS - 1) / SORT_THREADS;
            const int lo = tid * per, hi = min(lo + per, RADIX * nchunk);
            unsigned sum = 0;
            for (int e = lo; e < hi; ++e) sum += table[e];
            unsigned incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
            if (lane == 63) wave_tot[wave] = incl;
            __syncthreads();
            unsigned base = incl - sum;
            for (int w = 0; w < wave; ++w) base += wave_tot[w];
            for (int e = lo; e < hi; ++e) { const unsigned c = table[e]; table[e] = (unsigned short)base; base += c; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAX_T / SORT_THREADS; ++r) {
            if (r >= nround) break;
            const int i = r * SORT_THREADS + tid;
            if (i < T) perm_b[table[dig[r] * nchunk + r * (SORT_THREADS / 64) + wave] + rank[r]] = (unsigned short)val[r];
        }
        __syncthreads();
        unsigned short *sw = perm_a; perm_a = perm_b; perm_b = sw;
    }
    // segment bounds: every point learns [lo, hi) of its cell in sorted order.  A volume's cells hold one to three points: the cell's
    // head walks it and writes its members' bounds (17.7 us per sort; the scan below: 20.0).
    if (a2 >= 0) {
        for (int j = tid; j < T; j += SORT_THREADS) {
            const int t = perm_a[j];
            const unsigned id = ids[t];
            order[(size_t)b * T + j] = t;
            const bool first = (j == 0) || (ids[perm_a[j - 1]] != id);
            if (first) {
                int e = j + 1;
                while (e < T && ids[perm_a[e]] == id) ++e;
                for (int q = j; q < e; ++q) {
                    const int tq = perm_a[q];
                    seg_lo[(size_t)b * T + tq] = j;
                    seg_hi[(size_t)b * T + tq] = e;
                }
            }
        }
        return;
    }
    // A plane's cells hold tens of points, and that walk cost ~10 us of 25: there the bounds in parallel -- lo = the last head at or before
    // a position (a max-scan over contiguous ranges, the carry through one wave scan), a cell's end = the next head's position, left by
    // that head at the previous cell's lo.  perm_b and the count table are free after the last pass.
    {
        unsigned short *lo_arr = perm_b, *end_arr = table;
        const int per = (T + SORT_THREADS - 1) / SORT_THREADS;
        const int j0 = min(tid * per, T), j1 = min(j0 + per, T);
        int run = 0;
        for (int j = j0; j < j1; ++j) {
            if (j == 0 || ids[perm_a[j - 1]] != ids[perm_a[j]]) run = j;
            lo_arr[j] = (unsigned short)run;                        // (0 in front of the range's first head: the carry fills it in)
        }
        int incl = run;
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl = max(incl, up); }
        __syncthreads();                                            // (wave_tot's readers of the last pass are done)
        if (lane == 63) wave_tot[wave] = (unsigned)incl;
        __syncthreads();
        int carry = __shfl_up(incl, 1);
        if (lane == 0) carry = 0;
        for (int w = 0; w < wave; ++w) carry = max(carry, (int)wave_tot[w]);
        for (int j = j0; j < j1; ++j) lo_arr[j] = (unsigned short)max(carry, (int)lo_arr[j]);
        __syncthreads();
        for (int j = j0; j < j1; ++j) {
            if (j > 0 && lo_arr[j] == j) end_arr[lo_arr[j - 1]] = (unsigned short)j;
            if (j == T - 1) end_arr[lo_arr[j]] = (unsigned short)T;
        }
        __syncthreads();
        for (int j = tid; j < T; j += SORT_THREADS) {
            const int t = perm_a[j], lo = lo_arr[j];
            order[(size_t)b * T + j] = t;
            seg_lo[(size_t)b * T + t] = lo;
            seg_hi[(size_t)b * T + t] = end_arr[lo];
        }
    }
}

// ---- more than MAX_T points per scene: the same stable LSD radix sort through global memory -------------------------------
// Five bits per pass, a 64-element chunk per wave, the (digit, chunk) table scanned by one workgroup per scene.  The call's
// outputs double as its scratch: `order` and `seg_lo` hold the two permutations in turn (the pass count decides which one
// starts, so the last pass lands in `order`), `seg_hi` the table -- 32 * ceil(T/64) <= T entries.  Segment bounds by binary
// search over the sorted ids (a cell of 10^5 points costs the same 2 x 17 probes per point as a cell of one).
constexpr int GBITS = 5, GRADIX = 1 << GBITS;

__global__ void __launch_bounds__(256)
voxel_ids_kernel(const float *pts, int T, int R, float divisor, float clamp_hi, int a0, int a1, int a2, int *idx, int *perm) {
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const float *p = pts + ((size_t)b * T + t) * 3;
    const int ix = voxel_coord(p[a0], divisor, clamp_hi, R);
    const int iy = voxel_coord(p[a1], divisor, clamp_hi, R);
    const int iz = a2 >= 0 ? voxel_coord(p[a2], divisor, clamp_hi, R) : 0;
    idx[(size_t)b * T + t] = ix + R * (iy + R * iz);
    perm[(size_t)b * T + t] = t;
}

// every wave counts the digits of its chunk: lane d < 32 intersects the five bit ballots into the class of digit d
__global__ void __launch_bounds__(256)
gsort_count_kernel(const int *idx, const int *perm_in, int T, int nchunk, int shift, int *table) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= nchunk) return;
    const int i = chunk * 64 + lane;
    const bool valid = i < T;
    const unsigned d = valid ? ((unsigned)idx[(size_t)b * T + perm_in[(size_t)b * T + i]] >> shift) & (GRADIX - 1) : 0u;
    unsigned long long cls = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < GBITS; ++bit) {
        const unsigned long long ones = __ballot((d >> bit) & 1u);
        cls &= ((lane >> bit) & 1) ? ones : ~ones;
    }
    if (lane < GRADIX) table[(size_t)b * T + (size_t)lane * nchunk + chunk] = __popcll(cls);
}

// exclusive scan of one scene's table in (digit, chunk) order: tiles of 1024 x 4 entries with a running carry
__global__ void __launch_bounds__(1024)
gsort_scan_kernel(int *table, int T, int n) {
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned carry_s;
    int *tb = table + (size_t)blockIdx.x * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int e0 = base + tid * 4;
        unsigned v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = e0 + k < n ? (unsigned)tb[e0 + k] : 0u; sum += v[k]; }
        unsigned incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned pre = carry + incl - sum;
        for (int w = 0; w < wave; ++w) pre += wave_tot[w];
        if (tid == 1023) carry_s = pre + sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (e0 + k < n) tb[e0 + k] = (int)pre; pre += v[k]; }
        __syncthreads();
        carry = carry_s;
    }
}

__global__ void __launch_bounds__(256)
gsort_scatter_kernel(const int *idx, const int *perm_in, int T, int nchunk, int shift, const int *table, int *perm_out) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= nchunk) return;
    const int i = chunk * 64 + lane;
    const bool valid = i < T;
    const int v = valid ? perm_in[(size_t)b * T + i] : 0;
    const unsigned d = valid ? ((unsigned)idx[(size_t)b * T + v] >> shift) & (GRADIX - 1) : 0u;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < GBITS; ++bit) {
        const unsigned long long ones = __ballot((d >> bit) & 1u);
        same &= ((d >> bit) & 1u) ? ones : ~ones;
    }
    const int rank = __popcll(same & ((1ull << lane) - 1ull));            // equal digits keep their order: stable
    if (valid) perm_out[(size_t)b * T + table[(size_t)b * T + (size_t)d * nchunk + chunk] + rank] = v;
}

__global__ void __launch_bounds__(256)
segment_bounds_kernel(const int *idx, const int *order, int T, int *seg_lo, int *seg_hi) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= T) return;
    const int *ib = idx + (size_t)b * T, *ob = order + (size_t)b * T;
    const int t = ob[j], id = ib[t];
    int lo = 0, hi = j;                                                   // first position whose id is not below `id`
    while (lo < hi) { const int m = (lo + hi) >> 1; if (ib[ob[m]] < id) lo = m + 1; else hi = m; }
    int lo2 = j + 1, hi2 = T;                                             // first position whose id is above `id`
    while (lo2 < hi2) { const int m = (lo2 + hi2) >> 1; if (ib[ob[m]] <= id) lo2 = m + 1; else hi2 = m; }
    seg_lo[(size_t)b * T + t] = lo;
    seg_hi[(size_t)b * T + t] = lo2;
}

// The per-point kernels below use one thread per (point, channel): a 256-thread block covers
// 256/CL points x CL channel lanes (CL = min(C,256)), so there is no integer division per thread
// and a point's feature row is read as one coalesced run.
//
// ONE PASS PER SEGMENT.  A segment (the points of one cell) of at most SEG_SMALL points is reduced by the threads of its
// first point (its "head") and the result written to every member -- O(length) per segment, O(T C) per call.  A longer
// segment (dense cells: hand planes at 32^2, degenerate clouds with thousands of points in one cell) is reduced by a whole
// workgroup: the extra blocks of the launch each look at one SEG_SMALL-aligned position of the sorted order, and the block
// whose position is the first aligned one inside a long segment takes that segment (every long segment contains exactly one
// such position).  Max / argmax are order-independent up to ties, and ties go to the smallest sorted position in both
// paths (= the smallest point index = torch's first maximum); sums of long segments are accumulated as 256/CL interleaved
// partial sums combined in a fixed order -- deterministic, not the sequential order of the short path.
constexpr int SEG_SMALL = 32;

__device__ __forceinline__ bool point_lane(int C, uint32_t npts, uint32_t &bt, int &c) {
    const int cl = C < 256 ? C : 256;                          // channel lanes per point (C <= 256 here)
    const int ppb = 256 / cl;                                  // points per block
    const int lp = threadIdx.x / cl;
    c = threadIdx.x - lp * cl;
    bt = blockIdx.x * ppb + lp;
    return lp < ppb && bt < npts;
}

// Large-segment role of block `tile` (0-based among the extra blocks): (scene b, sorted range [lo,hi)) or false
__device__ __forceinline__ bool large_segment(uint32_t tile, int T, const int *order, const int *seg_lo, const int *seg_hi,
                                              uint32_t &b, int &lo, int &hi) {
    const uint32_t tiles = (uint32_t)((T + SEG_SMALL - 1) / SEG_SMALL);
    b = tile / tiles;
    const int pos = (int)(tile - b * tiles) * SEG_SMALL;
    const int t = order[(size_t)b * T + pos];
    lo = seg_lo[(size_t)b * T + t];
    hi = seg_hi[(size_t)b * T + t];
    return hi - lo > SEG_SMALL && lo > pos - SEG_SMALL;       // long, and `pos` is the first aligned position inside it
}

// out[b,t,c] = max over the voxel-mates of t of feat[b,.,c]; argmax = the point that wins
__global__ void __launch_bounds__(256)
pool_max_fwd_kernel(const float *feat, const int *order, const int *seg_lo, const int *seg_hi,
                    float *out, int *argmax, int T, int C, uint32_t npts, uint32_t small_blocks) {
    if (blockIdx.x < small_blocks) {
        uint32_t bt; int c0;
        if (!point_lane(C, npts, bt, c0)) return;
        const uint32_t b = bt / (uint32_t)T;
        const int t = (int)(bt - b * (uint32_t)T);
        const int lo = seg_lo[bt], hi = seg_hi[bt];
        const int *ord = order + (size_t)b * T;
        if (hi - lo > SEG_SMALL || ord[lo] != t) return;       // long segments: the cooperative blocks; otherwise the head works
        const float *fb = feat + (size_t)b * T * C;
        for (int c = c0; c < C; c += 256) {
            int best = t;
            float m = fb[(size_t)t * C + c];
            for (int j = lo + 1; j < hi; ++j) {
                const int t2 = ord[j];
                const float v = fb[(size_t)t2 * C + c];
                if (v > m) { m = v; best = t2; }
            }
            for (int j = lo; j < hi; ++j) {
                const size_t o = ((size_t)b * T + ord[j]) * C + c;
                out[o] = m;
                if (argmax) argmax[o] = best;
            }
        }
        return;
    }
    __shared__ float red_m[256];
    __shared__ int red_j[256];
    uint32_t b; int lo, hi;
    if (!large_segment(blockIdx.x - small_blocks, T, order, seg_lo, seg_hi, b, lo, hi)) return;
    const int cl = C < 256 ? C : 256, G = 256 / cl, grp = threadIdx.x / cl, c0 = threadIdx.x - grp * cl;
    const int *ord = order + (size_t)b * T;
    const float *fb = feat + (size_t)b * T * C;
    for (int c = c0; c < C; c += 256) {
        float m = 0.0f;
        int bj = -1;
        if (grp < G)
            for (int j = lo + grp; j < hi; j += G) {
                const float v = fb[(size_t)ord[j] * C + c];
                if (bj < 0 || v > m) { m = v; bj = j; }
            }
        red_m[threadIdx.x] = m; red_j[threadIdx.x] = bj;
        __syncthreads();
        if (grp == 0) {
            for (int g = 1; g < G; ++g) {
                const float v = red_m[g * cl + c0];
                const int j = red_j[g * cl + c0];
                if (j >= 0 && (v > m || (v == m && j < bj))) { m = v; bj = j; }
            }
            red_m[c0] = m; red_j[c0] = bj;
        }
        __syncthreads();
        m = red_m[c0];
        const int best = ord[red_j[c0]];
        if (grp < G)
            for (int j = lo + grp; j < hi; j += G) {
                const size_t o = ((size_t)b * T + ord[j]) * C + c;
                out[o] = m;
                if (argmax) argmax[o] = best;
            }
        __syncthreads();
    }
}

// sum over the sorted range [lo,hi) of src[b, ord[j], c] by the whole block (threads = channel lanes x groups); every
// thread of channel lane c0 returns the total.  red: 256 floats of LDS.
__device__ __forceinline__ float block_segment_sum(const float *src_b, const int *ord, int lo, int hi, int C, int c, int cl, int G,
                                                   int grp, int c0, float *red) {
    float s = 0.0f;
    if (grp < G)
        for (int j = lo + grp; j < hi; j += G) s += src_b[(size_t)ord[j] * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    float tot = 0.0f;
    for (int g = 0; g < G; ++g) tot += red[g * cl + c0];
    __syncthreads();
    return tot;
}

// pool_local with scatter_type = 'mean' (pointnet.py:64-69, 116-132: torch_scatter.scatter_mean, gathered back): every point gets
// the mean of the features of its cell.  The cell's first point (in sorted order) sums in that order and writes every member.
// The backward is the same map applied to the gradient (the mean over a cell is self-adjoint).
__global__ void __launch_bounds__(256)
pool_mean_kernel(const float *feat, const int *order, const int *seg_lo, const int *seg_hi, float *out, int T, int C, uint32_t npts) {
    uint32_t bt; int c0;
    if (!point_lane(C, npts, bt, c0)) return;
    const uint32_t b = bt / (uint32_t)T;
    const int t = (int)(bt - b * (uint32_t)T);
    const int lo = seg_lo[bt], hi = seg_hi[bt];
    const int *ord = order + (size_t)b * T;
    if (ord[lo] != t) return;                                   // the head of the cell works
    const float *fb = feat + (size_t)b * T * C;
    const float inv = 1.0f / (float)(hi - lo);
    for (int c = c0; c < C; c += 256) {
        float a = 0.0f;
        for (int j = lo; j < hi; ++j) a += fb[(size_t)ord[j] * C + c];
        a *= inv;
        for (int j = lo; j < hi; ++j) out[((size_t)b * T + ord[j]) * C + c] = a;
    }
}


// grad_feat[b,t,c] = sum over voxel-mates of grad_out[b,.,c] if t is the arg-max, else 0
__global__ void __launch_bounds__(256)
pool_max_bwd_kernel(const float *grad_out, const int *argmax, const int *order, const int *seg_lo, const int *seg_hi,
                    float *grad_feat, int T, int C, uint32_t npts, uint32_t small_blocks) {
    if (blockIdx.x < small_blocks) {
        uint32_t bt; int c0;
        if (!point_lane(C, npts, bt, c0)) return;
        const uint32_t b = bt / (uint32_t)T;
        const int t = (int)(bt - b * (uint32_t)T);
        const int lo = seg_lo[bt], hi = seg_hi[bt];
        if (hi - lo > SEG_SMALL) return;
        const int *ord = order + (size_t)b * T;
        const float *gb = grad_out + (size_t)b * T * C;
        for (int c = c0; c < C; c += 256) {
            float g = 0.0f;
            if (argmax[(size_t)bt * C + c] == t)
                for (int j = lo; j < hi; ++j) g += gb[(size_t)ord[j] * C + c];
            grad_feat[(size_t)bt * C + c] = g;
        }
        return;
    }
    __shared__ float red[256];
    uint32_t b; int lo, hi;
    if (!large_segment(blockIdx.x - small_blocks, T, order, seg_lo, seg_hi, b, lo, hi)) return;
    const int cl = C < 256 ? C : 256, G = 256 / cl, grp = threadIdx.x / cl, c0 = threadIdx.x - grp * cl;
    const int *ord = order + (size_t)b * T;
    for (int c = c0; c < C; c += 256) {
        const float tot = block_segment_sum(grad_out + (size_t)b * T * C, ord, lo, hi, C, c, cl, G, grp, c0, red);
        if (grp < G)
            for (int j = lo + grp; j < hi; j += G) {
                const int t = ord[j];
                const size_t o = ((size_t)b * T + t) * C + c;
                grad_feat[o] = argmax[o] == t ? tot : 0.0f;
            }
    }
}

// pool_local over SEVERAL cell partitions of the same points, summed (the hand encoder pools its point features over the xz, xy and yz
// planes and adds the three, pointnet.py:116-132): out[b][t][c] = sum_k max over t's cell in partition k, in the order k = 0, 1, ...
// (the order of the reference's `c += pooled`).  Gather form -- every (point, channel) walks its own cell in sorted order, strict '>'
// from the cell's head, so all members of a cell find the same maximum and the same first arg-max as vt_voxel_pool_max_fwd's head does
// -- because the partitions group the points differently: no thread could own an output element across them otherwise.  One launch
// instead of K pool launches and K - 1 framework adds per PointNet block; a cell of n points costs n^2 row reads (L2), which only
// matters for degenerate clouds.
constexpr int POOL_MAX_SETS = 4;
struct PoolSets { const int *order[POOL_MAX_SETS], *seg_lo[POOL_MAX_SETS], *seg_hi[POOL_MAX_SETS]; int *argmax[POOL_MAX_SETS]; int K; };

__global__ void __launch_bounds__(256)
pool_max_sum_fwd_kernel(const float *feat, PoolSets s, float *out, int T, int C, uint32_t npts) {
    uint32_t bt; int c0;
    if (!point_lane(C, npts, bt, c0)) return;
    const uint32_t b = bt / (uint32_t)T;
    const float *fb = feat + (size_t)b * T * C;
    for (int c = c0; c < C; c += 256) {
        float sum = 0.0f;
        for (int k = 0; k < s.K; ++k) {
            const int lo = s.seg_lo[k][bt], hi = s.seg_hi[k][bt];
            const int *ord = s.order[k] + (size_t)b * T;
            int best = ord[lo];
            float m = fb[(size_t)best * C + c];
            for (int j = lo + 1; j < hi; ++j) {
                const int t2 = ord[j];
                const float v = fb[(size_t)t2 * C + c];
                if (v > m) { m = v; best = t2; }
            }
            sum = k ? sum + m : m;
            if (s.argmax[k]) s.argmax[k][(size_t)bt * C + c] = best;
        }
        out[(size_t)bt * C + c] = sum;
    }
}

// its backward: grad_feat[b][t][c] = sum_k [t is the arg-max of its cell in partition k] * (sum of grad_out over that cell), k in order
__global__ void __launch_bounds__(256)
pool_max_sum_bwd_kernel(const float *grad_out, PoolSets s, float *grad_feat, int T, int C, uint32_t npts) {
    uint32_t bt; int c0;
    if (!point_lane(C, npts, bt, c0)) return;
    const uint32_t b = bt / (uint32_t)T;
    const int t = (int)(bt - b * (uint32_t)T);
    const float *gb = grad_out + (size_t)b * T * C;
    for (int c = c0; c < C; c += 256) {
        float sum = 0.0f;
        for (int k = 0; k < s.K; ++k) {
            if (s.argmax[k][(size_t)bt * C + c] != t) continue;
            const int lo = s.seg_lo[k][bt], hi = s.seg_hi[k][bt];
            const int *ord = s.order[k] + (size_t)b * T;
            float g = 0.0f;
            for (int j = lo; j < hi; ++j) g += gb[(size_t)ord[j] * C + c];
            sum += g;
        }
        grad_feat[(size_t)bt * C + c] = sum;
    }
}

// grid cell = mean of feat over the cell's points (grid pre-zeroed).  CL = false: NCDHW / NCHW [b][c][cell];
// CL = true: channels-last [b][cell][c] (what the UNet3D kernels read)
template <bool CL>
__global__ void __launch_bounds__(256)
scatter_mean_fwd_kernel(const float *feat, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                        float *grid, int T, int C, size_t V, uint32_t npts, uint32_t small_blocks, uint32_t feat_scenes = 0) {   // V = cells per channel
    // feat_scenes: the index arrays hold several partitions of the SAME feat_scenes scenes one after the other (vt_plane_scatter_mean_multi_fwd)
    auto cell = [&](uint32_t b, int c, int id) -> size_t {
        return CL ? ((size_t)b * V + (size_t)id) * C + c : ((size_t)b * C + c) * V + (size_t)id;
    };
    if (blockIdx.x < small_blocks) {
        uint32_t bt; int c0;
        if (!point_lane(C, npts, bt, c0)) return;
        const uint32_t b = bt / (uint32_t)T;
        const int t = (int)(bt - b * (uint32_t)T);
        const int lo = seg_lo[bt], hi = seg_hi[bt];
        const int *ord = order + (size_t)b * T;
        if (hi - lo > SEG_SMALL || ord[lo] != t) return;       // the cell's first point writes
        const float *fb = feat + (size_t)(feat_scenes ? b % feat_scenes : b) * T * C;
        for (int c = c0; c < C; c += 256) {
            float s = 0.0f;
            for (int j = lo; j < hi; ++j) s += fb[(size_t)ord[j] * C + c];
            grid[cell(b, c, idx[bt])] = s / (float)(hi - lo);
        }
        return;
    }
    __shared__ float red[256];
    uint32_t b; int lo, hi;
    if (!large_segment(blockIdx.x - small_blocks, T, order, seg_lo, seg_hi, b, lo, hi)) return;
    const int cl = C < 256 ? C : 256, G = 256 / cl, grp = threadIdx.x / cl, c0 = threadIdx.x - grp * cl;
    const int *ord = order + (size_t)b * T;
    const int id = idx[(size_t)b * T + ord[lo]];
    for (int c = c0; c < C; c += 256) {
        const float tot = block_segment_sum(feat + (size_t)(feat_scenes ? b % feat_scenes : b) * T * C, ord, lo, hi, C, c, cl, G, grp, c0, red);
        if (grp == 0) grid[cell(b, c, id)] = tot / (float)(hi - lo);
    }
}

template <bool CL>
__global__ void __launch_bounds__(256)
scatter_mean_bwd_kernel(const float *grad_grid, const int *idx, const int *seg_lo, const int *seg_hi,
                        float *grad_feat, int T, int C, size_t V, uint32_t npts) {
    uint32_t bt; int c0;
    if (!point_lane(C, npts, bt, c0)) return;
    const uint32_t b = bt / (uint32_t)T;
    const float inv = 1.0f / (float)(seg_hi[bt] - seg_lo[bt]);
    for (int c = c0; c < C; c += 256) {
        const size_t g = CL ? ((size_t)b * V + (size_t)idx[bt]) * C + c : ((size_t)b * C + c) * V + (size_t)idx[bt];
        grad_feat[(size_t)bt * C + c] = grad_grid[g] * inv;
    }
}

// the backward of n partitions' scatter-means of the same features: grad_feat[b][t][c] = sum_k grad_plane_k[b][c][cell_k(t)] / count_k, k in order
__global__ void __launch_bounds__(256)
scatter_mean_multi_bwd_kernel(const float *grad_planes, const int *idx, const int *seg_lo, const int *seg_hi, float *grad_feat,
                              int T, int C, size_t V, uint32_t npts, int n) {
    uint32_t bt; int c0;
    if (!point_lane(C, npts, bt, c0)) return;
    const uint32_t b = bt / (uint32_t)T, scenes = npts / (uint32_t)T;
    for (int c = c0; c < C; c += 256) {
        float sum = 0.0f;
        for (int k = 0; k < n; ++k) {
            const size_t kt = (size_t)k * npts + bt;
            const float g = grad_planes[(((size_t)k * scenes + b) * C + c) * V + (size_t)idx[kt]] * (1.0f / (float)(seg_hi[kt] - seg_lo[kt]));
            sum = k ? sum + g : g;
        }
        grad_feat[(size_t)bt * C + c] = sum;
    }
}

// the extra blocks of a launch that look for long segments: one per SEG_SMALL sorted positions per scene
inline unsigned large_blocks(int B, int T) { return (unsigned)B * (unsigned)((T + SEG_SMALL - 1) / SEG_SMALL); }

inline unsigned point_blocks(int C, size_t npts) {
    const int cl = C < 256 ? C : 256;
    const size_t ppb = 256 / cl;
    return (unsigned)((npts + ppb - 1) / ppb);
}

inline unsigned blocks_for(size_t total) {
    size_t g = (total + 255) / 256;
    const size_t cap = (size_t)vt_num_cus() * 8;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// flags[b][tile]: bit 0 where no point of scene b lies in the 10^3 halo of the 8^3 voxel block `tile` (the voxels a 3x3x3 conv over the
// block reads): the scatter-mean grid is zero there; bit 1 where none lies in its 12^3 halo either (what a second 3x3x3 conv behind the
// first depends on), so the values are 0, 1 and 3.  One workgroup per scene; the flags of a scene fit its LDS (R <= 128: 4096 blocks).
__global__ void __launch_bounds__(256)
tile_flags_kernel(const int *idx, int T, int R, unsigned char *flags) {
    __shared__ unsigned char f[FLAG_MAX_TILES], f2[FLAG_MAX_TILES];
    const int b = blockIdx.x, nt1 = R >> 3, nt = nt1 * nt1 * nt1;
    for (int t = threadIdx.x; t < nt; t += 256) { f[t] = 1; f2[t] = 2; }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += 256) {
        const int id = idx[(size_t)b * T + i];
        const int x = id % R, y = (id / R) % R, z = id / (R * R);
        // the blocks whose halo [8 t - 1, 8 t + 8] contains the voxel: t from (v - 8) / 8 rounded up to (v + 1) / 8
        const int x0 = max((x - 1) >> 3, 0), x1 = min((x + 1) >> 3, nt1 - 1);
        const int y0 = max((y - 1) >> 3, 0), y1 = min((y + 1) >> 3, nt1 - 1);
        const int z0 = max((z - 1) >> 3, 0), z1 = min((z + 1) >> 3, nt1 - 1);
        for (int tz = z0; tz <= z1; ++tz)
            for (int ty = y0; ty <= y1; ++ty)
                for (int tx = x0; tx <= x1; ++tx) f[(tz * nt1 + ty) * nt1 + tx] = 0;         // (racing stores of the same value)
        for (int tz = max((z - 2) >> 3, 0); tz <= min((z + 2) >> 3, nt1 - 1); ++tz)
            for (int ty = max((y - 2) >> 3, 0); ty <= min((y + 2) >> 3, nt1 - 1); ++ty)
                for (int tx = max((x - 2) >> 3, 0); tx <= min((x + 2) >> 3, nt1 - 1); ++tx) f2[(tz * nt1 + ty) * nt1 + tx] = 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += 256) flags[(size_t)b * nt + t] = f[t] | f2[t];
}

int build_launch(const char *who, const float *pts, int B, int T, int R, float divisor, float clamp_hi,
                 int a0, int a1, int a2, int *idx, int *order, int *seg_lo, int *seg_hi, void *stream,
                 void *fill = nullptr, size_t fill_bytes = 0, unsigned char *tile_flags = nullptr, int planes = 0) {
    char msg[96];
    auto fail = [&](int code, const char *what) { snprintf(msg, sizeof msg, "%s: %s", who, what); return vt_fail(code, msg); };
    if (!pts || !idx || !order || !seg_lo || !seg_hi) return fail(VT_ERR_INVALID, "null argument");
    if (B <= 0 || T <= 0 || R < 1 || R > 1024) return fail(VT_ERR_INVALID, "bad size");
    if ((unsigned long long)B * (unsigned long long)T > 0x7fffffffull) return fail(VT_ERR_UNSUPPORTED, "more than 2^31 points per call");
    // bits of the largest cell id
    unsigned long long cells = (unsigned long long)R * R * (a2 >= 0 ? (unsigned long long)R : 1ull);
    int nbits = 1;
    while (nbits < 32 && (1ull << nbits) < cells) ++nbits;
    static const bool force_global = getenv("VTACO_VOXEL_GLOBAL_SORT") != nullptr;     // tests: the large-cloud path on small clouds
    if (fill_bytes && (!fill || (fill_bytes & 15) || ((size_t)fill & 15))) return fail(VT_ERR_INVALID, "the buffer to clear must be 16-byte aligned and sized");
    if (tile_flags && (a2 < 0 || R < 8 || (R & 7) || R > 128)) return fail(VT_ERR_UNSUPPORTED, "block flags: a volume whose resolution is a multiple of 8, at most 128");
    if (planes && (T > MAX_T || (force_global && T >= 64))) {       // large clouds: plane after plane through the global-memory sort
        const int n = planes & 15, scenes = B / n;
        for (int pl = 0; pl < n; ++pl) {
            const int plane = (planes >> (4 + 2 * pl)) & 3;
            const size_t off = (size_t)pl * scenes * T;
            const int rc = build_launch(who, pts, scenes, T, R, divisor, clamp_hi, plane == 2 ? 1 : 0, plane == 1 ? 1 : 2, -1,
                                        idx + off, order + off, seg_lo + off, seg_hi + off, stream);
            if (rc) return rc;
        }
        return 0;
    }
    if (T > MAX_T || (force_global && T >= 64)) {
        hipStream_t s = (hipStream_t)stream;
        if (fill_bytes) { const int frc = vt_fill32(fill, 0u, fill_bytes, s); if (frc) return frc; }
        const int nchunk = (T + 63) / 64, passes = (nbits + GBITS - 1) / GBITS;
        int *cur = (passes & 1) ? seg_lo : order, *oth = (passes & 1) ? order : seg_lo;   // an odd pass count ends in the other buffer
        const dim3 pg((unsigned)((T + 255) / 256), (unsigned)B), cg((unsigned)((nchunk + 3) / 4), (unsigned)B);
        hipLaunchKernelGGL(voxel_ids_kernel, pg, dim3(256), 0, s, pts, T, R, divisor, clamp_hi, a0, a1, a2, idx, cur);
        for (int pass = 0; pass < passes; ++pass) {
            hipLaunchKernelGGL(gsort_count_kernel, cg, dim3(256), 0, s, idx, cur, T, nchunk, pass * GBITS, seg_hi);
            hipLaunchKernelGGL(gsort_scan_kernel, dim3(B), dim3(1024), 0, s, seg_hi, T, GRADIX * nchunk);
            hipLaunchKernelGGL(gsort_scatter_kernel, cg, dim3(256), 0, s, idx, cur, T, nchunk, pass * GBITS, seg_hi, oth);
            int *sw = cur; cur = oth; oth = sw;
        }
        hipLaunchKernelGGL(segment_bounds_kernel, pg, dim3(256), 0, s, idx, order, T, seg_lo, seg_hi);
        if (tile_flags) hipLaunchKernelGGL(tile_flags_kernel, dim3((unsigned)B), dim3(256), 0, s, idx, T, R, tile_flags);
        return vt_check(hipGetLastError(), who);
    }
    bool attr_set = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr_set) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&voxel_build_kernel), (int)SORT_LDS);
        if (e != hipSuccess) return vt_check(e, "vt_voxel_build: hipFuncSetAttribute");
        attr_set = true;
    }
    const size_t fill16 = fill_bytes / 16;
    size_t fb = (fill16 + SORT_THREADS * 16 - 1) / (SORT_THREADS * 16);                 // >= 16 stores per thread, at most two rounds of the chip
    if (fb > (size_t)vt_num_cus() * 2) fb = (size_t)vt_num_cus() * 2;
    hipLaunchKernelGGL(voxel_build_kernel, dim3((unsigned)(B + fb)), dim3(SORT_THREADS), SORT_LDS, (hipStream_t)stream,
                       pts, T, nbits, R, divisor, clamp_hi, a0, a1, a2, idx, order, seg_lo, seg_hi, B, (uint4 *)fill, fill16, tile_flags, planes);
    return vt_check(hipGetLastError(), who);
}

}  // namespace

extern "C" {

int vt_voxel_tile_flags(const int *idx, int B, int T, int R, unsigned char *flags, void *stream) {
    if (!idx || !flags || B <= 0 || T <= 0) return vt_fail(VT_ERR_INVALID, "vt_voxel_tile_flags: bad argument");
    if (R < 8 || (R & 7) || R > 128) return vt_fail(VT_ERR_UNSUPPORTED, "vt_voxel_tile_flags: resolution must be a multiple of 8, at most 128");
    hipLaunchKernelGGL(tile_flags_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, idx, T, R, flags);
    return vt_check(hipGetLastError(), "vt_voxel_tile_flags");
}

int vt_voxel_build(const float *pts, int B, int T, int R, double padding,
                   int *idx, int *order, int *seg_lo, int *seg_hi, void *stream) {
    return build_launch("vt_voxel_build", pts, B, T, R, (float)(1.0 + padding + 10e-4), 0.999f, 0, 1, 2,
                        idx, order, seg_lo, seg_hi, stream);
}

int vt_voxel_build_clear(const float *pts, int B, int T, int R, double padding,
                         int *idx, int *order, int *seg_lo, int *seg_hi, void *clear, size_t clear_bytes, void *stream) {
    return build_launch("vt_voxel_build_clear", pts, B, T, R, (float)(1.0 + padding + 10e-4), 0.999f, 0, 1, 2,
                        idx, order, seg_lo, seg_hi, stream, clear, clear_bytes);
}

int vt_voxel_build_clear_flags(const float *pts, int B, int T, int R, double padding, int *idx, int *order, int *seg_lo, int *seg_hi,
                               void *clear, size_t clear_bytes, unsigned char *tile_flags, void *stream) {
    if (!tile_flags) return vt_fail(VT_ERR_INVALID, "vt_voxel_build_clear_flags: null flags");
    return build_launch("vt_voxel_build_clear_flags", pts, B, T, R, (float)(1.0 + padding + 10e-4), 0.999f, 0, 1, 2,
                        idx, order, seg_lo, seg_hi, stream, clear, clear_bytes, tile_flags);
}

int vt_plane_build(const float *pts, int B, int T, int R, double padding, int plane,
                   int *idx, int *order, int *seg_lo, int *seg_hi, void *stream) {
    if (plane < 0 || plane > 2) return vt_fail(VT_ERR_INVALID, "vt_plane_build: plane must be 0 (xz), 1 (xy) or 2 (yz)");
    const int a0 = plane == 2 ? 1 : 0, a1 = plane == 1 ? 1 : 2;
    return build_launch("vt_plane_build", pts, B, T, R, (float)(1.0 + padding + 10e-6), (float)(1.0 - 10e-6), a0, a1, -1,
                        idx, order, seg_lo, seg_hi, stream);
}

int vt_plane_build_multi(const float *pts, int B, int T, int R, double padding, int n_planes, const int *planes,
                         int *idx, int *order, int *seg_lo, int *seg_hi, void *stream) {
    if (n_planes < 1 || n_planes > 3 || !planes) return vt_fail(VT_ERR_INVALID, "vt_plane_build_multi: one to three planes");
    int code = n_planes;
    for (int k = 0; k < n_planes; ++k) {
        if (planes[k] < 0 || planes[k] > 2) return vt_fail(VT_ERR_INVALID, "vt_plane_build_multi: plane must be 0 (xz), 1 (xy) or 2 (yz)");
        code |= planes[k] << (4 + 2 * k);
    }
    if ((unsigned long long)B * n_planes > 0x7fffffffull / (unsigned long long)(T > 0 ? T : 1)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_plane_build_multi: more than 2^31 points per call");
    return build_launch("vt_plane_build_multi", pts, B * n_planes, T, R, (float)(1.0 + padding + 10e-6), (float)(1.0 - 10e-6), 0, 2, -1,
                        idx, order, seg_lo, seg_hi, stream, nullptr, 0, nullptr, code);
}

int vt_plane_scatter_mean_fwd(const float *feat, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *plane, void *stream) {
    if (!feat || !idx || !order || !seg_lo || !seg_hi || !plane)
        return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_fwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_fwd: bad size");
    const size_t V = (size_t)R * R;
    int frc = vt_fill32(plane, 0u, (size_t)B * C * V * sizeof(float), (hipStream_t)stream);
    if (frc) return frc;
    const unsigned sb = point_blocks(C, (size_t)B * T);
    hipLaunchKernelGGL(scatter_mean_fwd_kernel<false>, dim3(sb + large_blocks(B, T)), dim3(256), 0, (hipStream_t)stream,
                       feat, idx, order, seg_lo, seg_hi, plane, T, C, V, (uint32_t)((size_t)B * T), sb);
    return vt_check(hipGetLastError(), "vt_plane_scatter_mean_fwd");
}

// n partitions of the same scenes (index arrays [n][B][T] as vt_plane_build_multi leaves them) in one launch each way: planes [n][B][C][R^2]
int vt_plane_scatter_mean_multi_fwd(const float *feat, int n, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                                    int B, int T, int C, int R, float *planes, void *stream) {
    if (!feat || !idx || !order || !seg_lo || !seg_hi || !planes) return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_multi_fwd: null argument");
    if (n < 1 || n > 3 || B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_multi_fwd: bad size");
    if ((unsigned long long)n * B * T > 0x7fffffffull) return vt_fail(VT_ERR_UNSUPPORTED, "vt_plane_scatter_mean_multi_fwd: more than 2^31 points per call");
    const size_t V = (size_t)R * R;
    const int frc = vt_fill32(planes, 0u, (size_t)n * B * C * V * sizeof(float), (hipStream_t)stream);
    if (frc) return frc;
    const unsigned sb = point_blocks(C, (size_t)n * B * T);
    hipLaunchKernelGGL(scatter_mean_fwd_kernel<false>, dim3(sb + large_blocks(n * B, T)), dim3(256), 0, (hipStream_t)stream,
                       feat, idx, order, seg_lo, seg_hi, planes, T, C, V, (uint32_t)((size_t)n * B * T), sb, (uint32_t)B);
    return vt_check(hipGetLastError(), "vt_plane_scatter_mean_multi_fwd");
}

int vt_plane_scatter_mean_multi_bwd(const float *grad_planes, int n, const int *idx, const int *seg_lo, const int *seg_hi,
                                    int B, int T, int C, int R, float *grad_feat, void *stream) {
    if (!grad_planes || !idx || !seg_lo || !seg_hi || !grad_feat) return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_multi_bwd: null argument");
    if (n < 1 || n > 3 || B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_multi_bwd: bad size");
    hipLaunchKernelGGL(scatter_mean_multi_bwd_kernel, dim3(point_blocks(C, (size_t)B * T)), dim3(256), 0, (hipStream_t)stream,
                       grad_planes, idx, seg_lo, seg_hi, grad_feat, T, C, (size_t)R * R, (uint32_t)((size_t)B * T), n);
    return vt_check(hipGetLastError(), "vt_plane_scatter_mean_multi_bwd");
}

int vt_plane_scatter_mean_bwd(const float *grad_plane, const int *idx, const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *grad_feat, void *stream) {
    if (!grad_plane || !idx || !seg_lo || !seg_hi || !grad_feat)
        return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_bwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_plane_scatter_mean_bwd: bad size");
    hipLaunchKernelGGL(scatter_mean_bwd_kernel<false>, dim3(point_blocks(C, (size_t)B * T)), dim3(256), 0, (hipStream_t)stream,
                       grad_plane, idx, seg_lo, seg_hi, grad_feat, T, C, (size_t)R * R, (uint32_t)((size_t)B * T));
    return vt_check(hipGetLastError(), "vt_plane_scatter_mean_bwd");
}

int vt_voxel_pool_max_fwd(const float *feat, const int *order, const int *seg_lo, const int *seg_hi,
                          int B, int T, int C, float *out, int *argmax, void *stream) {
    if (!feat || !order || !seg_lo || !seg_hi || !out) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_fwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_fwd: bad size");
    const size_t npts = (size_t)B * T;
    const unsigned sb = point_blocks(C, npts);
    hipLaunchKernelGGL(pool_max_fwd_kernel, dim3(sb + large_blocks(B, T)), dim3(256), 0, (hipStream_t)stream,
                       feat, order, seg_lo, seg_hi, out, argmax, T, C, (uint32_t)npts, sb);
    return vt_check(hipGetLastError(), "vt_voxel_pool_max_fwd");
}

static int pool_sets(PoolSets &s, const char *who, int K, const int *const *order, const int *const *seg_lo, const int *const *seg_hi,
                     int *const *argmax, bool need_argmax) {
    if (K < 1 || K > POOL_MAX_SETS || !order || !seg_lo || !seg_hi || (need_argmax && !argmax)) return vt_fail(VT_ERR_INVALID, who);
    s.K = K;
    for (int k = 0; k < POOL_MAX_SETS; ++k) {
        s.order[k] = k < K ? order[k] : nullptr; s.seg_lo[k] = k < K ? seg_lo[k] : nullptr; s.seg_hi[k] = k < K ? seg_hi[k] : nullptr;
        s.argmax[k] = (k < K && argmax) ? argmax[k] : nullptr;
        if (k < K && (!s.order[k] || !s.seg_lo[k] || !s.seg_hi[k] || (need_argmax && !s.argmax[k]))) return vt_fail(VT_ERR_INVALID, who);
    }
    return 0;
}

int vt_voxel_pool_max_sum_fwd(const float *feat, int K, const int *const *order, const int *const *seg_lo, const int *const *seg_hi,
                              int B, int T, int C, float *out, int *const *argmax, void *stream) {
    if (!feat || !out) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_sum_fwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_sum_fwd: bad size");
    PoolSets s;
    if (const int rc = pool_sets(s, "vt_voxel_pool_max_sum_fwd: 1..4 partitions with their three index arrays", K, order, seg_lo, seg_hi, argmax, false)) return rc;
    const size_t npts = (size_t)B * T;
    hipLaunchKernelGGL(pool_max_sum_fwd_kernel, dim3(point_blocks(C, npts)), dim3(256), 0, (hipStream_t)stream, feat, s, out, T, C, (uint32_t)npts);
    return vt_check(hipGetLastError(), "vt_voxel_pool_max_sum_fwd");
}

int vt_voxel_pool_max_sum_bwd(const float *grad_out, int K, int *const *argmax, const int *const *order, const int *const *seg_lo,
                              const int *const *seg_hi, int B, int T, int C, float *grad_feat, void *stream) {
    if (!grad_out || !grad_feat) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_sum_bwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_sum_bwd: bad size");
    PoolSets s;
    if (const int rc = pool_sets(s, "vt_voxel_pool_max_sum_bwd: 1..4 partitions with their index arrays and arg-maxima", K, order, seg_lo, seg_hi, argmax, true)) return rc;
    const size_t npts = (size_t)B * T;
    hipLaunchKernelGGL(pool_max_sum_bwd_kernel, dim3(point_blocks(C, npts)), dim3(256), 0, (hipStream_t)stream, grad_out, s, grad_feat, T, C, (uint32_t)npts);
    return vt_check(hipGetLastError(), "vt_voxel_pool_max_sum_bwd");
}

int vt_voxel_pool_mean(const float *feat, const int *order, const int *seg_lo, const int *seg_hi, int B, int T, int C, float *out, void *stream) {
    if (!feat || !order || !seg_lo || !seg_hi || !out) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_mean: null argument");
    if (B <= 0 || T <= 0 || C <= 0) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_mean: bad size");
    const size_t npts = (size_t)B * T;
    hipLaunchKernelGGL(pool_mean_kernel, dim3(point_blocks(C, npts)), dim3(256), 0, (hipStream_t)stream, feat, order, seg_lo, seg_hi, out, T, C,
                       (uint32_t)npts);
    return vt_check(hipGetLastError(), "vt_voxel_pool_mean");
}

int vt_voxel_pool_max_bwd(const float *grad_out, const int *argmax, const int *order, const int *seg_lo, const int *seg_hi,
                          int B, int T, int C, float *grad_feat, void *stream) {
    if (!grad_out || !argmax || !order || !seg_lo || !seg_hi || !grad_feat)
        return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_bwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0) return vt_fail(VT_ERR_INVALID, "vt_voxel_pool_max_bwd: bad size");
    const size_t npts = (size_t)B * T;
    const unsigned sb = point_blocks(C, npts);
    hipLaunchKernelGGL(pool_max_bwd_kernel, dim3(sb + large_blocks(B, T)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, argmax, order, seg_lo, seg_hi, grad_feat, T, C, (uint32_t)npts, sb);
    return vt_check(hipGetLastError(), "vt_voxel_pool_max_bwd");
}

int vt_voxel_scatter_mean_fwd(const float *feat, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *grid, void *stream) {
    if (!feat || !idx || !order || !seg_lo || !seg_hi || !grid)
        return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_fwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_fwd: bad size");
    const size_t V = (size_t)R * R * R, total = (size_t)B * T * C;
    int frc = vt_fill32(grid, 0u, (size_t)B * C * V * sizeof(float), (hipStream_t)stream);
    if (frc) return frc;
    const unsigned sb = point_blocks(C, (size_t)B * T);
    hipLaunchKernelGGL(scatter_mean_fwd_kernel<false>, dim3(sb + large_blocks(B, T)), dim3(256), 0, (hipStream_t)stream,
                       feat, idx, order, seg_lo, seg_hi, grid, T, C, V, (uint32_t)((size_t)B * T), sb);
    return vt_check(hipGetLastError(), "vt_voxel_scatter_mean_fwd");
}

int vt_voxel_scatter_mean_bwd(const float *grad_grid, const int *idx, const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *grad_feat, void *stream) {
    if (!grad_grid || !idx || !seg_lo || !seg_hi || !grad_feat)
        return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_bwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_bwd: bad size");
    const size_t V = (size_t)R * R * R, total = (size_t)B * T * C;
    hipLaunchKernelGGL(scatter_mean_bwd_kernel<false>, dim3(point_blocks(C, (size_t)B * T)), dim3(256), 0, (hipStream_t)stream,
                       grad_grid, idx, seg_lo, seg_hi, grad_feat, T, C, V, (uint32_t)((size_t)B * T));
    return vt_check(hipGetLastError(), "vt_voxel_scatter_mean_bwd");
}

// the same scatter straight into / out of a channels-last grid [B,R,R,R,C] (the layout the UNet3D kernels read)
int vt_voxel_scatter_mean_cl_fwd(const float *feat, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                                 int B, int T, int C, int R, float *grid_cl, void *stream) {
    if (!feat || !idx || !order || !seg_lo || !seg_hi || !grid_cl) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_cl_fwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_cl_fwd: bad size");
    const size_t V = (size_t)R * R * R;
    int frc = vt_fill32(grid_cl, 0u, (size_t)B * C * V * sizeof(float), (hipStream_t)stream);
    if (frc) return frc;
    const unsigned sb = point_blocks(C, (size_t)B * T);
    hipLaunchKernelGGL(scatter_mean_fwd_kernel<true>, dim3(sb + large_blocks(B, T)), dim3(256), 0, (hipStream_t)stream,
                       feat, idx, order, seg_lo, seg_hi, grid_cl, T, C, V, (uint32_t)((size_t)B * T), sb);
    return vt_check(hipGetLastError(), "vt_voxel_scatter_mean_cl_fwd");
}

int vt_voxel_scatter_mean_cl_bwd(const float *grad_grid_cl, const int *idx, const int *seg_lo, const int *seg_hi,
                                 int B, int T, int C, int R, float *grad_feat, void *stream) {
    if (!grad_grid_cl || !idx || !seg_lo || !seg_hi || !grad_feat) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_cl_bwd: null argument");
    if (B <= 0 || T <= 0 || C <= 0 || R < 1) return vt_fail(VT_ERR_INVALID, "vt_voxel_scatter_mean_cl_bwd: bad size");
    hipLaunchKernelGGL(scatter_mean_bwd_kernel<true>, dim3(point_blocks(C, (size_t)B * T)), dim3(256), 0, (hipStream_t)stream,
                       grad_grid_cl, idx, seg_lo, seg_hi, grad_feat, T, C, (size_t)R * R * R, (uint32_t)((size_t)B * T));
    return vt_check(hipGetLastError(), "vt_voxel_scatter_mean_cl_bwd");
}

}  // extern "C"
