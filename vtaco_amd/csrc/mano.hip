// MANO hand layer forward for gfx950: pose -> 778 skinned vertices + 21 joints, one workgroup per hand.
// Replaces ManoLayer.forward (reference src/encoder/manolayer.py:160-364) in the configuration the shipped
// configs build (configs/VTacO/VTacO_YCB.yaml:46-56: axis-angle root and joints, use_pca False,
// flat_hand_mean False, the model's own betas, no translation, right hand): Rodrigues through the
// normalised quaternion (manopth/rodrigues_layer.py:15-60), pose blend shapes, the three-level kinematic
// chain off the wrist, linear blend skinning, finger-tip vertices appended as joints, centring on a joint.
//
// The work per hand is 0.95 MFLOP (135 pose blend shapes x 2334 coordinates dominate) over a 1.3 MB
// model that stays in L2 across the batch: latency-bound, so everything runs in one launch with the
// intermediate results (rotations, chain, posed vertices) in LDS and the blend-shape matrix stored
// transposed ([135][2336]) so that consecutive lanes read consecutive coordinates.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vt_common.h"
#include "vtaco_hip.h"

namespace {

constexpr int NV = 778, NJ = 16, NC = NV * 3, NCP = 2336, NPD = 135;
constexpr int OFF_MEAN = 0, OFF_J = 48, OFF_VS = 96, OFF_W = OFF_VS + NCP, OFF_PD = OFF_W + NV * NJ;
constexpr int BLOB_FLOATS = OFF_PD + NPD * NCP;
static_assert(BLOB_FLOATS == VT_MANO_BLOB_FLOATS, "header and kernel disagree on the MANO blob size");
constexpr int THREADS = 256;

__constant__ int PARENT[NJ] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__constant__ int TIPS[5] = {745, 317, 444, 556, 673};                                  // right hand (manolayer.py:328)
// the left hand's middle-finger tip is vertex 445 (manolayer.py:330): the blob's unused mean slot 47 carries the side (0 right, 1 left)
constexpr int OFF_SIDE = 47;
__device__ __forceinline__ int tip_vertex(const float *blob, int i) { return TIPS[i] + ((i == 2 && blob[OFF_SIDE] != 0.0f) ? 1 : 0); }
// output joint order (manolayer.py:339): entries < 16 index the chain joints, 16..20 the five tips
__constant__ int JORDER[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

__global__ void __launch_bounds__(THREADS)
mano_pack_kernel(const float *v_template, const float *shapedirs, const float *betas, const float *posedirs,
                 const float *j_regressor, const float *weights, const float *hands_mean, float *blob, int left) {
    __shared__ float vs[NCP];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        for (int e = tid; e < NCP; e += THREADS) {
            float v = 0.0f;
            if (e < NC) {
                if (betas) for (int k = 0; k < 10; ++k) v += shapedirs[(size_t)e * 10 + k] * betas[k];
                v += v_template[e];
            }
            vs[e] = v;
            blob[OFF_VS + e] = v;
        }
        __syncthreads();
        if (tid < 48) {
            const int j = tid / 3, c = tid - 3 * j;
            float s = 0.0f;
            for (int v = 0; v < NV; ++v) s += j_regressor[(size_t)j * NV + v] * vs[3 * v + c];
            blob[OFF_J + tid] = s;
            blob[OFF_MEAN + tid] = tid < 45 ? hands_mean[tid] : (tid == OFF_SIDE && left ? 1.0f : 0.0f);
        }
        for (int e = tid; e < NV * NJ; e += THREADS) blob[OFF_W + e] = weights[e];
    }
    // posedirs [2334][135] -> [135][2336]
    for (size_t i = (size_t)blockIdx.x * THREADS + tid; i < (size_t)NPD * NCP; i += (size_t)gridDim.x * THREADS) {
        const int k = (int)(i / NCP), e = (int)(i - (size_t)k * NCP);
        blob[OFF_PD + i] = e < NC ? posedirs[(size_t)e * NPD + k] : 0.0f;
    }
}

// What both kernels start from, per hand (all 256 threads call it; it ends with a barrier): the joint rotations (Rodrigues through
// the re-normalised quaternion) with the pose map of the 15 finger joints, the kinematic chain G_j = [Rg_j | tg_j] off the wrist, the
// skinning transforms A_j = [Rg_j | tg_j - Rg_j J_j] and the posed rest vertices vp = v_shaped + posedirs . pose_map.  `qa` (or null)
// receives per joint (a[3], |a + 1e-8|, sin, cos of half the angle, |quaternion| before normalisation) for the backward.
// mano_chain: everything but the blend shapes (16 threads' worth of work); mano_blend: vp[e] for the coordinates [e0, e1) -- per
// element the sum runs over k in order whichever thread or workgroup computes it, four rows of posedirs requested at a time (a
// workgroup that streams the 1.26 MB matrix one row per round trip is bound by the latency of a cold row: 285 us per launch in the
// training step, where other kernels evict the model from L2 between two calls).
__device__ __forceinline__ void mano_blend(const float *blob, const float *pm, float *vp, int e0, int e1) {
    const float *vs = blob + OFF_VS, *pd = blob + OFF_PD;
    constexpr int NE = 4, NK = 4;                                   // coordinates per thread and pass, rows per request round: 16 loads in flight
    for (int base = e0 + (int)threadIdx.x; base < e1; base += THREADS * NE) {
        float acc[NE];
        int e[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) { acc[i] = 0.0f; e[i] = min(base + THREADS * i, e1 - 1); }      // (a clamped lane repeats the last coordinate and does not store)
        int k = 0;
        for (; k + NK <= NPD; k += NK) {
            float r[NK][NE];
#pragma unroll
            for (int u = 0; u < NK; ++u)
#pragma unroll
                for (int i = 0; i < NE; ++i) r[u][i] = pd[(size_t)(k + u) * NCP + e[i]];
#pragma unroll
            for (int u = 0; u < NK; ++u)
#pragma unroll
                for (int i = 0; i < NE; ++i) acc[i] = fmaf(r[u][i], pm[k + u], acc[i]);
        }
        for (; k < NPD; ++k)
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[i] = fmaf(pd[(size_t)k * NCP + e[i]], pm[k], acc[i]);
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (base + THREADS * i < e1) vp[base + THREADS * i] = vs[base + THREADS * i] + acc[i];
    }
}

__device__ __forceinline__ void mano_chain(const float *p, const float *blob, float (*rot)[9], float *pm, float (*G)[12], float (*A)[12],
                                           float (*qa)[8]) {
    const int tid = threadIdx.x;
    const float *J = blob + OFF_J;
    if (tid < NJ) {
        // full pose = [root axis-angle | hands_mean + joint angles]  (manolayer.py:191)
        float a[3];
        for (int c = 0; c < 3; ++c) {
            const int i = 3 * tid + c;
            a[c] = i < 3 ? p[i] : blob[OFF_MEAN + i - 3] + p[i];
        }
        const float e0 = a[0] + 1e-8f, e1 = a[1] + 1e-8f, e2 = a[2] + 1e-8f;
        const float angle = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        const float half = angle * 0.5f, sn = sinf(half), cs = cosf(half);
        float w = cs, x = sn * (a[0] / angle), y = sn * (a[1] / angle), z = sn * (a[2] / angle);
        const float qn = sqrtf(w * w + x * x + y * y + z * z);
        if (qa) { qa[tid][0] = a[0]; qa[tid][1] = a[1]; qa[tid][2] = a[2]; qa[tid][3] = angle; qa[tid][4] = sn; qa[tid][5] = cs; qa[tid][6] = qn; }
        w /= qn; x /= qn; y /= qn; z /= qn;
        const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
        const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
        float *r = rot[tid];
        r[0] = w2 + x2 - y2 - z2; r[1] = 2 * xy - 2 * wz;     r[2] = 2 * wy + 2 * xz;
        r[3] = 2 * wz + 2 * xy;     r[4] = w2 - x2 + y2 - z2; r[5] = 2 * yz - 2 * wx;
        r[6] = 2 * xz - 2 * wy;     r[7] = 2 * wx + 2 * yz;     r[8] = w2 - x2 - y2 + z2;
        if (tid > 0)
            for (int k = 0; k < 9; ++k) pm[9 * (tid - 1) + k] = r[k] - ((k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f);
    }
    __syncthreads();
    // kinematic chain, one level per step: G_j = G_parent * [R_j | J_j - J_parent]  (manolayer.py:264-303)
    for (int level = 0; level < 4; ++level) {
        if (tid < NJ) {
            const int par = PARENT[tid];
            const int my_level = tid == 0 ? 0 : ((tid - 1) % 3) + 1;
            if (my_level == level) {
                float t[3];
                for (int c = 0; c < 3; ++c) t[c] = par < 0 ? J[c] : J[3 * tid + c] - J[3 * par + c];
                if (par < 0) {
                    for (int r = 0; r < 3; ++r) {
                        for (int c = 0; c < 3; ++c) G[0][4 * r + c] = rot[0][3 * r + c];
                        G[0][4 * r + 3] = t[r];
                    }
                } else {
                    for (int r = 0; r < 3; ++r) {
                        const float *gp = &G[par][4 * r];
                        for (int c = 0; c < 3; ++c)
                            G[tid][4 * r + c] = gp[0] * rot[tid][c] + gp[1] * rot[tid][3 + c] + gp[2] * rot[tid][6 + c];
                        G[tid][4 * r + 3] = gp[0] * t[0] + gp[1] * t[1] + gp[2] * t[2] + gp[3];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid < NJ) {
        // skinning uses A_j = [R | t - R J_j]  (manolayer.py:305-307); the joints are the chain translations G[j][.][3]
        for (int r = 0; r < 3; ++r) {
            const float *g = &G[tid][4 * r];
            float *o = &A[tid][4 * r];
            o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
            o[3] = g[3] - (g[0] * J[3 * tid] + g[1] * J[3 * tid + 1] + g[2] * J[3 * tid + 2]);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void mano_state(const float *p, const float *blob, float (*rot)[9], float *pm, float (*G)[12], float (*A)[12],
                                           float *vp, float (*qa)[8]) {
    mano_chain(p, blob, rot, pm, G, A, qa);
    // pose blend shapes: v_posed = v_shaped + posedirs . pose_map, coalesced rows
    mano_blend(blob, pm, vp, 0, NCP);
    __syncthreads();
}

// One workgroup per (hand, slice of MANO_VS vertices): every workgroup repeats the chain (tiny) and computes the blend shapes and
// the skinning of its own vertices only -- 158 KB of posedirs instead of 1.26 MB per workgroup, eight times the workgroups; slice 0
// writes the chain joints, the slice that owns a finger-tip vertex that tip's joint row.  A centre joint that is a tip (center_idx
// 4, 8, 12, 16, 20) is one more vertex that every slice computes for itself.  Same arithmetic per element as one workgroup per hand.
constexpr int MANO_SLICES = 8, MANO_VS = (NV + MANO_SLICES - 1) / MANO_SLICES;
__global__ void __launch_bounds__(THREADS)
mano_fwd_kernel(const float *pose, const float *blob, int center_idx, float *verts, float *joints) {
    __shared__ float rot[NJ][9];        // per-joint rotations
    __shared__ float pm[NPD + 1];       // pose map: (R_j - I) of the 15 finger joints, row-major
    __shared__ float Gc[NJ][12];        // chain transforms [Rg | tg]: the translations are the joints
    __shared__ float G[NJ][12];         // skinning transforms A_j
    __shared__ float vp[3 * (MANO_VS + 1)];     // posed rest vertices of the slice (+ the centre tip), then the skinned ones
    __shared__ float ctr_s[3];
    const int tid = threadIdx.x, b = blockIdx.x / MANO_SLICES, sl = blockIdx.x - b * MANO_SLICES;
    const int v0 = sl * MANO_VS, nv = min(MANO_VS, NV - v0);
    const int csrc = center_idx >= 0 ? JORDER[center_idx] : -1, ctip = csrc >= NJ ? tip_vertex(blob, csrc - NJ) : -1;     // the centre, if it is a tip vertex
    mano_chain(pose + (size_t)b * 48, blob, rot, pm, Gc, G, nullptr);
    mano_blend(blob, pm, vp - 3 * v0, 3 * v0, 3 * (v0 + nv));
    if (ctip >= 0) mano_blend(blob, pm, vp + 3 * nv - 3 * ctip, 3 * ctip, 3 * ctip + 3);
    __syncthreads();
    // linear blend skinning: T_v = sum_j w[v][j] A_j, vertex = T_v [v_posed; 1]
    const float *W = blob + OFF_W;
    float out[3] = {0.0f, 0.0f, 0.0f};
    const bool mine = tid < nv || (ctip >= 0 && tid == nv);
    if (mine) {
        const int v = tid < nv ? v0 + tid : ctip;
        float T[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) T[q] = 0.0f;
        for (int j = 0; j < NJ; ++j) {
            const float w = W[v * NJ + j];
#pragma unroll
            for (int q = 0; q < 12; ++q) T[q] = fmaf(w, G[j][q], T[q]);
        }
        const float x = vp[3 * tid], y = vp[3 * tid + 1], z = vp[3 * tid + 2];
        for (int r = 0; r < 3; ++r) out[r] = T[4 * r] * x + T[4 * r + 1] * y + T[4 * r + 2] * z + T[4 * r + 3];
    }
    __syncthreads();
    if (mine) for (int r = 0; r < 3; ++r) vp[3 * tid + r] = out[r];
    if (tid < 3) ctr_s[tid] = center_idx < 0 ? 0.0f : (csrc < NJ ? Gc[csrc][4 * tid + 3] : 0.0f);
    __syncthreads();
    if (ctip >= 0 && tid < 3) ctr_s[tid] = vp[3 * nv + tid];
    __syncthreads();
    const float ctr[3] = {ctr_s[0], ctr_s[1], ctr_s[2]};
    for (int e = tid; e < 3 * nv; e += THREADS) verts[(size_t)b * NC + 3 * v0 + e] = vp[e] - ctr[e % 3];
    if (tid < 63) {
        const int i = tid / 3, r = tid - 3 * i, src = JORDER[i];
        if (src < NJ) { if (sl == 0) joints[(size_t)b * 63 + tid] = Gc[src][4 * r + 3] - ctr[r]; }
        else {
            const int tv = tip_vertex(blob, src - NJ);
            if (tv >= v0 && tv < v0 + nv) joints[(size_t)b * 63 + tid] = vp[3 * (tv - v0) + r] - ctr[r];
        }
    }
}


// ---- backward: d(pose) from d(verts), d(joints) (training: loss_mano / loss_pc, reference autograd through
// manolayer.py:186-347).  One workgroup per hand again: the forward's intermediates are recomputed into LDS (rotations,
// chain, posed vertices), then the chain rule in reverse --
//   centring and the joint / tip selection            d out_v, d tg_j
//   skinning     out_v = (sum_j w_vj A_j) [vp_v; 1]    d A_j = sum_v w_vj d out_v (x) [vp_v; 1]  (thread = (joint, entry), v in order)
//                                                      d vp_v = (sum_j w_vj Rg_j)^T d out_v
//   A_j = [Rg_j | tg_j - Rg_j J_j], the chain Rg_j = Rg_p R_j, tg_j = Rg_p t_j + tg_p, leaves first (the root's five children
//   added by one thread in joint order)                 d R_j, j = 0..15
//   pose blend shapes  vp = vs + pd^T pm               d pm_k = pd_k . d vp   (a wave per row, lanes over coordinates)
//   Rodrigues through the normalised quaternion         d a_j
// No atomics: every sum has a fixed order (bit-reproducible).
__device__ __forceinline__ float wave_sum_fixed(float x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

__global__ void __launch_bounds__(THREADS)
mano_bwd_kernel(const float *pose, const float *blob, int center_idx, const float *dverts, const float *djoints, float *dpose) {
    __shared__ float rot[NJ][9];
    __shared__ float pm[NPD + 1];
    __shared__ float G[NJ][12];         // chain [Rg | tg]
    __shared__ float A[NJ][12];         // skinning [Rg | tg - Rg J]
    __shared__ float vp[NCP];           // posed rest vertices
    __shared__ float dout[NCP];         // d out_v, later d vp_v
    __shared__ float dA[NJ][12];        // [d Rg (from skinning) | d ta]
    __shared__ float dtg[NJ][3];
    __shared__ float dR[NJ][9];         // d R_j (local rotations)
    __shared__ float dpm[NPD + 1];
    __shared__ float red[4][3];
    __shared__ float qa[NJ][8];         // per joint: a[3], angle, sin(half), cos(half), |u|
    const int tid = threadIdx.x, b = blockIdx.x;
    const float *p = pose + (size_t)b * 48;
    const float *J = blob + OFF_J;

    mano_state(p, blob, rot, pm, G, A, vp, qa);                     // the forward's intermediates
    // ---------------- d out_v, d tg_j from the outputs' gradients ----------------
    const float *dV = dverts + (size_t)b * NC, *dJ = djoints + (size_t)b * 63;
    float s3[3] = {0.0f, 0.0f, 0.0f};
    for (int v = tid; v < NV; v += THREADS)
        for (int r = 0; r < 3; ++r) { const float g = dV[3 * v + r]; dout[3 * v + r] = g; s3[r] += g; }
    if (tid < NCP - NC) dout[NC + tid] = 0.0f;
    for (int r = 0; r < 3; ++r) { s3[r] = wave_sum_fixed(s3[r]); if ((tid & 63) == 0) red[tid >> 6][r] = s3[r]; }
    if (tid < NJ) for (int r = 0; r < 3; ++r) dtg[tid][r] = 0.0f;
    __syncthreads();
    if (tid == 0) {
        float dj[21][3], dctr[3] = {0.0f, 0.0f, 0.0f};
        for (int i = 0; i < 21; ++i) for (int r = 0; r < 3; ++r) { dj[i][r] = dJ[3 * i + r]; dctr[r] -= dj[i][r]; }
        if (center_idx >= 0) for (int r = 0; r < 3; ++r) dj[center_idx][r] += dctr[r] - (((red[0][r] + red[1][r]) + red[2][r]) + red[3][r]);
        for (int i = 0; i < 21; ++i) {
            const int src = JORDER[i];
            for (int r = 0; r < 3; ++r) {
                if (src < NJ) dtg[src][r] += dj[i][r];
                else dout[3 * tip_vertex(blob, src - NJ) + r] += dj[i][r];
            }
        }
    }
    __syncthreads();
    // ---------------- skinning: d A_j (thread = (joint, entry of the 3x4), vertices in order) ----------------
    const float *W = blob + OFF_W;
    {
        const int j = tid >> 4, q = tid & 15;
        if (q < 12) {
            const int r = q >> 2, c = q & 3;
            float acc = 0.0f;
            for (int v = 0; v < NV; ++v) acc = fmaf(W[v * NJ + j] * dout[3 * v + r], c < 3 ? vp[3 * v + c] : 1.0f, acc);
            dA[j][q] = acc;
        }
    }
    __syncthreads();
    // d vp_v = (sum_j w_vj Rg_j)^T d out_v, into dout's place (a thread touches its own vertices only)
    for (int v = tid; v < NV; v += THREADS) {
        float T[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) T[q] = 0.0f;
        for (int j = 0; j < NJ; ++j) {
            const float w = W[v * NJ + j];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) T[3 * r + c] = fmaf(w, A[j][4 * r + c], T[3 * r + c]);
        }
        const float g0 = dout[3 * v], g1 = dout[3 * v + 1], g2 = dout[3 * v + 2];
        for (int c = 0; c < 3; ++c) dout[3 * v + c] = T[c] * g0 + T[3 + c] * g1 + T[6 + c] * g2;
    }
    // ---------------- A_j -> chain: d Rg_j (kept in dA[j][4r+c]), d tg_j ----------------
    if (tid < NJ) {
        for (int r = 0; r < 3; ++r) {
            const float dta = dA[tid][4 * r + 3];
            dtg[tid][r] += dta;
            for (int c = 0; c < 3; ++c) dA[tid][4 * r + c] -= dta * J[3 * tid + c];
        }
    }
    __syncthreads();
    for (int level = 3; level >= 1; --level) {
        // a joint hands d Rg, d tg to its parent; levels 3 and 2 have one child per parent, the root's five children go in joint order
        if (level > 1 ? (tid < NJ && tid > 0 && ((tid - 1) % 3) + 1 == level) : tid == 0) {
            for (int jj = (level > 1 ? tid : 1); jj < (level > 1 ? tid + 1 : NJ); jj += (level > 1 ? 1 : 3)) {
                const int par = PARENT[jj];
                float t[3];
                for (int c = 0; c < 3; ++c) t[c] = J[3 * jj + c] - J[3 * par + c];
                for (int r = 0; r < 3; ++r) {
                    // d R_j = Rg_p^T d Rg_j
                    for (int c = 0; c < 3; ++c)
                        dR[jj][3 * r + c] = G[par][r] * dA[jj][c] + G[par][4 + r] * dA[jj][4 + c] + G[par][8 + r] * dA[jj][8 + c];
                }
                for (int r = 0; r < 3; ++r) {
                    // d Rg_p += d Rg_j R_j^T + d tg_j (x) t_j ;  d tg_p += d tg_j
                    for (int c = 0; c < 3; ++c)
                        dA[par][4 * r + c] += dA[jj][4 * r] * rot[jj][3 * c] + dA[jj][4 * r + 1] * rot[jj][3 * c + 1] + dA[jj][4 * r + 2] * rot[jj][3 * c + 2]
                                              + dtg[jj][r] * t[c];
                    dtg[par][r] += dtg[jj][r];
                }
            }
        }
        __syncthreads();
    }
    if (tid < 9) dR[0][tid] = dA[0][4 * (tid / 3) + tid % 3];
    // ---------------- pose blend shapes: d pm_k = pd_k . d vp ----------------
    {
        const int wave = tid >> 6, lane = tid & 63;
        const float *pd = blob + OFF_PD;
        for (int k = wave; k < NPD; k += 4) {
            const float *row = pd + (size_t)k * NCP;
            float acc = 0.0f;
            int e = lane;
            for (; e + 64 * 7 < NC; e += 64 * 8) {                 // (eight requests in flight; the sum keeps its order)
                float r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) r[u] = row[e + 64 * u];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fmaf(r[u], dout[e + 64 * u], acc);
            }
            for (; e < NC; e += 64) acc = fmaf(row[e], dout[e], acc);
            acc = wave_sum_fixed(acc);
            if (lane == 0) dpm[k] = acc;
        }
    }
    __syncthreads();
    // ---------------- Rodrigues: d a_j ----------------
    if (tid < NJ) {
        float d[9];
        for (int k = 0; k < 9; ++k) d[k] = dR[tid][k] + (tid > 0 ? dpm[9 * (tid - 1) + k] : 0.0f);
        const float a0 = qa[tid][0], a1 = qa[tid][1], a2 = qa[tid][2], th = qa[tid][3], sn = qa[tid][4], cs = qa[tid][5], qn = qa[tid][6];
        const float uw = cs, ux = sn * (a0 / th), uy = sn * (a1 / th), uz = sn * (a2 / th);
        const float w = uw / qn, x = ux / qn, y = uy / qn, z = uz / qn;
        const float dw = 2 * w * (d[0] + d[4] + d[8]) - 2 * z * d[1] + 2 * y * d[2] + 2 * z * d[3] - 2 * x * d[5] - 2 * y * d[6] + 2 * x * d[7];
        const float dx = 2 * x * (d[0] - d[4] - d[8]) + 2 * y * d[1] + 2 * z * d[2] + 2 * y * d[3] - 2 * w * d[5] + 2 * z * d[6] + 2 * w * d[7];
        const float dy = 2 * y * (-d[0] + d[4] - d[8]) + 2 * x * d[1] + 2 * w * d[2] + 2 * x * d[3] + 2 * z * d[5] - 2 * w * d[6] + 2 * z * d[7];
        const float dz = 2 * z * (-d[0] - d[4] + d[8]) - 2 * w * d[1] + 2 * x * d[2] + 2 * w * d[3] + 2 * y * d[5] + 2 * x * d[6] + 2 * y * d[7];
        // q = u / |u|
        const float dot = w * dw + x * dx + y * dy + z * dz;
        const float duw = (dw - w * dot) / qn, dux = (dx - x * dot) / qn, duy = (dy - y * dot) / qn, duz = (dz - z * dot) / qn;
        // u = (cos h, sin h * a / theta), theta = |a + 1e-8|, h = theta / 2
        const float e[3] = {(a0 + 1e-8f) / th, (a1 + 1e-8f) / th, (a2 + 1e-8f) / th};       // d theta / d a_c
        const float av[3] = {a0, a1, a2}, du[3] = {dux, duy, duz};
        const float adu = (a0 * dux + a1 * duy + a2 * duz) / th;                             // sum_i du_i a_i / theta
        for (int c = 0; c < 3; ++c) {
            const float da = duw * (-0.5f * sn * e[c]) + adu * (0.5f * cs * e[c]) + sn * (du[c] / th - adu * e[c] / th);
            dpose[(size_t)b * 48 + 3 * tid + c] = da;
            (void)av;
        }
    }
}

}  // namespace

extern "C" {

int vt_mano_pack_side(const float *v_template, const float *shapedirs, const float *betas, const float *posedirs,
                      const float *j_regressor, const float *weights, const float *hands_mean, int left, float *blob, void *stream) {
    if (!v_template || !posedirs || !j_regressor || !weights || !hands_mean || !blob)
        return vt_fail(VT_ERR_INVALID, "vt_mano_pack: null argument");
    if (betas && !shapedirs) return vt_fail(VT_ERR_INVALID, "vt_mano_pack: betas without shapedirs");
    hipLaunchKernelGGL(mano_pack_kernel, dim3(256), dim3(THREADS), 0, (hipStream_t)stream,
                       v_template, shapedirs, betas, posedirs, j_regressor, weights, hands_mean, blob, left ? 1 : 0);
    return vt_check(hipGetLastError(), "vt_mano_pack");
}

int vt_mano_pack(const float *v_template, const float *shapedirs, const float *betas, const float *posedirs,
                 const float *j_regressor, const float *weights, const float *hands_mean, float *blob, void *stream) {
    return vt_mano_pack_side(v_template, shapedirs, betas, posedirs, j_regressor, weights, hands_mean, 0, blob, stream);
}

int vt_mano_fwd(const float *pose, int B, const float *blob, int center_idx, float *verts, float *joints, void *stream) {
    if (B == 0) return 0;
    if (!pose || !blob || !verts || !joints) return vt_fail(VT_ERR_INVALID, "vt_mano_fwd: null argument");
    if (B < 0 || center_idx < -1 || center_idx > 20) return vt_fail(VT_ERR_INVALID, "vt_mano_fwd: bad size or centre joint");
    hipLaunchKernelGGL(mano_fwd_kernel, dim3((unsigned)B * MANO_SLICES), dim3(THREADS), 0, (hipStream_t)stream, pose, blob, center_idx, verts, joints);
    return vt_check(hipGetLastError(), "vt_mano_fwd");
}

int vt_mano_bwd(const float *pose, int B, const float *blob, int center_idx, const float *dverts, const float *djoints,
                float *dpose, void *stream) {
    if (B == 0) return 0;
    if (!pose || !blob || !dverts || !djoints || !dpose) return vt_fail(VT_ERR_INVALID, "vt_mano_bwd: null argument");
    if (B < 0 || center_idx < -1 || center_idx > 20) return vt_fail(VT_ERR_INVALID, "vt_mano_bwd: bad size or centre joint");
    hipLaunchKernelGGL(mano_bwd_kernel, dim3(B), dim3(THREADS), 0, (hipStream_t)stream, pose, blob, center_idx, dverts, djoints, dpose);
    return vt_check(hipGetLastError(), "vt_mano_bwd");
}

}  // extern "C"
