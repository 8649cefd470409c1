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
// output joint order (manolayer.py:339): entries < 16 index the chain joints, 16..20 the five tips
__constant__ int JORDER[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

__global__ void __launch_bounds__(THREADS)
mano_pack_kernel(const float *v_template, const float *shapedirs, const float *betas, const float *posedirs,
                 const float *j_regressor, const float *weights, const float *hands_mean, float *blob) {
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
            blob[OFF_MEAN + tid] = tid < 45 ? hands_mean[tid] : 0.0f;
        }
        for (int e = tid; e < NV * NJ; e += THREADS) blob[OFF_W + e] = weights[e];
    }
    // posedirs [2334][135] -> [135][2336]
    for (size_t i = (size_t)blockIdx.x * THREADS + tid; i < (size_t)NPD * NCP; i += (size_t)gridDim.x * THREADS) {
        const int k = (int)(i / NCP), e = (int)(i - (size_t)k * NCP);
        blob[OFF_PD + i] = e < NC ? posedirs[(size_t)e * NPD + k] : 0.0f;
    }
}

__global__ void __launch_bounds__(THREADS)
mano_fwd_kernel(const float *pose, const float *blob, int center_idx, float *verts, float *joints) {
    __shared__ float rot[NJ][9];        // per-joint rotations
    __shared__ float pm[NPD + 1];       // pose map: (R_j - I) of the 15 finger joints, row-major
    __shared__ float G[NJ][12];         // chain transforms [R | t], then rest-pose-removed in place (A)
    __shared__ float gt[NJ][3];         // chain translations before the rest pose is removed (= joints)
    __shared__ float vp[NCP];           // posed rest vertices, then the skinned vertices
    __shared__ float jout[21][3];
    const int tid = threadIdx.x, b = blockIdx.x;
    const float *p = pose + (size_t)b * 48;

    if (tid < NJ) {
        // full pose = [root axis-angle | hands_mean + joint angles]  (manolayer.py:191)
        float a[3];
        for (int c = 0; c < 3; ++c) {
            const int i = 3 * tid + c;
            a[c] = i < 3 ? p[i] : blob[OFF_MEAN + i - 3] + p[i];
        }
        const float e0 = a[0] + 1e-8f, e1 = a[1] + 1e-8f, e2 = a[2] + 1e-8f;
        const float angle = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        const float half = angle * 0.5f, sn = sinf(half);
        float w = cosf(half), x = sn * (a[0] / angle), y = sn * (a[1] / angle), z = sn * (a[2] / angle);
        const float qn = sqrtf(w * w + x * x + y * y + z * z);
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
    const float *J = blob + OFF_J;
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
        // joints are the chain translations; skinning uses A_j = [R | t - R J_j]  (manolayer.py:305-307)
        for (int r = 0; r < 3; ++r) {
            float *g = &G[tid][4 * r];
            gt[tid][r] = g[3];
            g[3] = g[3] - (g[0] * J[3 * tid] + g[1] * J[3 * tid + 1] + g[2] * J[3 * tid + 2]);
        }
    }
    // pose blend shapes: v_posed = v_shaped + posedirs . pose_map, ten coordinates per thread, coalesced rows
    {
        float acc[10];
        const float *vs = blob + OFF_VS, *pd = blob + OFF_PD;
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[i] = 0.0f;
        for (int k = 0; k < NPD; ++k) {
            const float m = pm[k];
            const float *row = pd + (size_t)k * NCP;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int e = tid + THREADS * i;
                if (e < NCP) acc[i] = fmaf(row[e], m, acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int e = tid + THREADS * i;
            if (e < NCP) vp[e] = vs[e] + acc[i];
        }
    }
    __syncthreads();
    // linear blend skinning: T_v = sum_j w[v][j] A_j, vertex = T_v [v_posed; 1]
    const float *W = blob + OFF_W;
    float out[4][3];
    for (int i = 0; i < 4; ++i) {
        const int v = tid + THREADS * i;
        if (v < NV) {
            float T[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) T[q] = 0.0f;
            for (int j = 0; j < NJ; ++j) {
                const float w = W[v * NJ + j];
#pragma unroll
                for (int q = 0; q < 12; ++q) T[q] = fmaf(w, G[j][q], T[q]);
            }
            const float x = vp[3 * v], y = vp[3 * v + 1], z = vp[3 * v + 2];
            for (int r = 0; r < 3; ++r) out[i][r] = T[4 * r] * x + T[4 * r + 1] * y + T[4 * r + 2] * z + T[4 * r + 3];
        }
    }
    __syncthreads();
    for (int i = 0; i < 4; ++i) {
        const int v = tid + THREADS * i;
        if (v < NV) for (int r = 0; r < 3; ++r) vp[3 * v + r] = out[i][r];
    }
    __syncthreads();
    if (tid < 21) {
        const int src = JORDER[tid];
        for (int r = 0; r < 3; ++r) jout[tid][r] = src < NJ ? gt[src][r] : vp[3 * TIPS[src - NJ] + r];
    }
    __syncthreads();
    float ctr[3] = {0.0f, 0.0f, 0.0f};
    if (center_idx >= 0) for (int r = 0; r < 3; ++r) ctr[r] = jout[center_idx][r];
    for (int e = tid; e < NC; e += THREADS) verts[(size_t)b * NC + e] = vp[e] - ctr[e % 3];
    if (tid < 63) joints[(size_t)b * 63 + tid] = jout[tid / 3][tid % 3] - ctr[tid % 3];
}

}  // namespace

extern "C" {

int vt_mano_pack(const float *v_template, const float *shapedirs, const float *betas, const float *posedirs,
                 const float *j_regressor, const float *weights, const float *hands_mean, float *blob, void *stream) {
    if (!v_template || !posedirs || !j_regressor || !weights || !hands_mean || !blob)
        return vt_fail(VT_ERR_INVALID, "vt_mano_pack: null argument");
    if (betas && !shapedirs) return vt_fail(VT_ERR_INVALID, "vt_mano_pack: betas without shapedirs");
    hipLaunchKernelGGL(mano_pack_kernel, dim3(256), dim3(THREADS), 0, (hipStream_t)stream,
                       v_template, shapedirs, betas, posedirs, j_regressor, weights, hands_mean, blob);
    return vt_check(hipGetLastError(), "vt_mano_pack");
}

int vt_mano_fwd(const float *pose, int B, const float *blob, int center_idx, float *verts, float *joints, void *stream) {
    if (B == 0) return 0;
    if (!pose || !blob || !verts || !joints) return vt_fail(VT_ERR_INVALID, "vt_mano_fwd: null argument");
    if (B < 0 || center_idx < -1 || center_idx > 20) return vt_fail(VT_ERR_INVALID, "vt_mano_fwd: bad size or centre joint");
    hipLaunchKernelGGL(mano_fwd_kernel, dim3(B), dim3(THREADS), 0, (hipStream_t)stream, pose, blob, center_idx, verts, joints);
    return vt_check(hipGetLastError(), "vt_mano_fwd");
}

}  // extern "C"
