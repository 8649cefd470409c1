// Shared by the TransformerFusion forward (fusion.hip) and backward (fusion_bwd.hip) kernels: the epilogue weight
// blob, the train-mode dropout rule and the layout of what the training forward leaves for the backward.
#pragma once
#include "decode_common.h"

namespace {

constexpr int FU_WT = 0, FU_W1A = 1024, FU_W1B = 2048, FU_W2A = 3072, FU_W2B = 4096;
constexpr int FU_BIAS = 5120;           // b1a, b1b, b2, gamma, beta fragments [2][16] each
constexpr int FU_BLOB = FU_BIAS + 5 * 32;

struct FusionUnitDev {
    const float *WK, *WQ, *WV, *Wt, *l1w, *l1b, *l2w, *l2b, *lnw, *lnb;
};

// The tactile rows of the decoder's self-attention unit given by finger id instead of a dense [B][N][32] tensor (vt_fusion_fwd_ids:
// what the reference gathers on the host, generation.py:159-255): point n of batch element b reads table[ids[row(b)][n]], a zero
// row where the id is 255 or not a row of the table; row(b) = chunk[b] (the chunks a generator picked out of a lattice) or b.
__device__ const float vt_fusion_zero_row[32] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
struct XIds {
    const unsigned char *ids = nullptr;     // [rows][N]
    const float *table = nullptr;           // [F][32]
    const int *chunk = nullptr;
    int N = 0;
    unsigned F = 0;                         // rows of the table: an id >= F (255 included) reads the zero row
    __device__ __forceinline__ const float *row(size_t p) const {
        const size_t b = p / (size_t)N, n = p - b * (size_t)N;
        const unsigned id = ids[(chunk ? (size_t)chunk[b] : b) * (size_t)N + n];
        return id >= F ? vt_fusion_zero_row : table + (size_t)id * 32;
    }
};

// epilogue weights -> accumulator-fed fragment order (k = chan_of(s, h))
__global__ void fusion_pack_kernel(FusionUnitDev u, float *blob) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < FU_BLOB; e += gridDim.x * blockDim.x) {
        float v;
        if (e < FU_BIAS) {
            const int L = e >> 10, s = (e >> 6) & 15, l = e & 63, i = l & 31, h = l >> 5, k = chan_of(s, h);
            if (L == 0) v = u.Wt[i * 32 + k];
            else if (L == 1) v = u.l1w[i * 32 + k];               // linear1 rows 0..31
            else if (L == 2) v = u.l1w[(32 + i) * 32 + k];        // linear1 rows 32..63
            else if (L == 3) v = u.l2w[i * 64 + k];               // linear2 cols 0..31
            else v = u.l2w[i * 64 + 32 + k];                      // linear2 cols 32..63
        } else {
            const int q = e - FU_BIAS, j = q >> 5, h = (q >> 4) & 1, r = q & 15, o = chan_of(r, h);
            v = j == 0 ? u.l1b[o] : j == 1 ? u.l1b[32 + o] : j == 2 ? u.l2b[o] : j == 3 ? u.lnw[o] : u.lnb[o];
        }
        blob[e] = v;
    }
}

// ---- train-mode dropout of TransNonlinear (reference TransformerFusion.py:13-25: ``dropout`` on relu(linear1) and
// ``dropout2`` on linear2's output, p = 0.1) -------------------------------------------------------------------------
// The keep decision of element (call, which, point, channel) is a pure function of a 64-bit seed, so the backward
// replays the forward's masks without storing them (and vt_fusion_dropout_mask materialises them for tests).
// call: 0 = encoder self-attention on c, 1 = decoder self-attention on c_img, 2 = cross-attention; which: 0 = dropout (64 ch),
// 1 = dropout2 (32 ch).  torch's own generator cannot be matched bit for bit (its CPU and GPU streams differ as well).
struct DropCfg {
    unsigned long long seed;
    uint32_t thresh;        // keep iff hash >= thresh; 0 = no dropout
    float scale;            // 1 / (1 - p)
    uint32_t call;
    uint32_t shift;         // the point's place in the hashed word: 6 at d_model 32 (channels < 64), 8 at the wider models (< 256)
};
__host__ __device__ inline DropCfg drop_cfg(float p, unsigned long long seed, uint32_t call, uint32_t shift = 6) {
    DropCfg d;
    d.seed = seed;
    d.call = call;
    d.shift = shift;
    if (!(p > 0.0f)) { d.thresh = 0; d.scale = 1.0f; return d; }
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    d.scale = 1.0f / (1.0f - p);
    return d;
}
__device__ __forceinline__ float drop_mask(const DropCfg &d, uint32_t which, uint32_t point, uint32_t channel) {
    if (d.thresh == 0) return 1.0f;
    unsigned long long z = d.seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)(d.call * 2 + which) << 40) + ((unsigned long long)point << d.shift) + channel + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((uint32_t)(z >> 32) >= d.thresh) ? d.scale : 0.0f;
}
__device__ __forceinline__ f32x16 drop16(const f32x16 &v, const DropCfg &d, uint32_t which, uint32_t point, int h, int chan_base) {
    f32x16 r;
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = v[s] * drop_mask(d, which, point, (uint32_t)(chan_base + chan_of(s, h)));
    return r;
}

// ---- what vt_fusion_fwd_train leaves for vt_fusion_bwd: per attention call the softmax row sums (as 1/l), the column sums s,
// the value projections V, the attention outputs O and the pre-InstanceNorm sums Z; plus the two inner results (mem, tgt).
// No N x N tensor: the backward recomputes the scores tile by tile like the forward.
struct FusionSaved {
    float *linv[3], *s[3], *V[3], *O[3], *Z[3], *M, *T;
};
inline size_t fusion_saved_layout(int B, int N, FusionSaved *sv, char *base, int C = 32) {
    const size_t P = (size_t)B * N;
    size_t off = 0;
    auto take = [&](size_t floats) { float *p = base ? (float *)(base + off) : nullptr; off += (floats * 4 + 255) / 256 * 256; return p; };
    FusionSaved t;
    for (int c = 0; c < 3; ++c) { t.linv[c] = take(P); t.s[c] = take(P); t.V[c] = take(P * C); t.O[c] = take(P * C); t.Z[c] = take(P * C); }
    t.M = take(P * C); t.T = take(P * C);
    if (sv) *sv = t;
    return off;
}

// ---- workgroup -> (row block, chunk): the row blocks of ONE chunk on ONE XCD --------------------------------------------
// The N x N passes (forward and backward) launch nrb = ceil(N / rows per workgroup) workgroups per chunk, each streaming the
// chunk's complete other operand (N rows of 192 or 256 B).  Workgroups are dealt to the eight XCDs round robin by linear id,
// so with the row block as the fast grid index the eight workgroups of a 2048-point chunk landed on eight different L2s and
// every one of them fetched the same rows from the fabric (profiles/r03_fusion_pmc_summary.csv: 2.2-4.4 x the unique bytes
// per launch).  With a 1-D grid of nrb * B workgroups, id w runs on XCD w & 7 and is the (w >> 3)-th workgroup there: XCD x
// takes the chunks x, x + 8, x + 16, ... and walks each chunk's row blocks back to back, so a chunk's streamed rows are
// fetched once into ONE L2 and hit there for the other row blocks (consecutive ids of an XCD are resident together).
__device__ __forceinline__ void chunk_of_workgroup(int nrb, int B, int &rb, int &b) {
    const unsigned w = blockIdx.x;
    if ((B & 7) == 0) {
        const unsigned idx = w >> 3;
        rb = (int)(idx % (unsigned)nrb);
        b = (int)(idx / (unsigned)nrb) * 8 + (int)(w & 7u);
    } else {                                                    // fewer than eight chunks or a ragged count: plain order
        rb = (int)(w % (unsigned)nrb);
        b = (int)(w / (unsigned)nrb);
    }
}

inline FusionUnitDev unit_of(const vt_fusion_unit &u) {
    return FusionUnitDev{u.WK, u.WQ, u.WV, u.trans_conv, u.linear1_w, u.linear1_b, u.linear2_w, u.linear2_b, u.norm2_w, u.norm2_b};
}

}  // namespace
