// Generalized winding number of query points against a triangle mesh, for the VTacO (t2d) training step: the reference labels
// its re-sampled query points with igl.fast_winding_number_for_meshes(V, F, Q) (src/conv_onet/training.py:723, 862) -- libigl's
// hierarchical APPROXIMATION of w(q) = 1/(4 pi) sum_f Omega_f(q).  libigl is an un-vendored dependency that is absent here, so
// this kernel evaluates the exact sum it approximates (Van Oosterom-Strackee solid angles, float64): 1 inside a closed,
// outward-oriented mesh, 0 outside, fractional for open meshes.  Sixteen lanes per query point, faces through LDS in tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vt_common.h"
#include "vtaco_hip.h"

namespace {

constexpr int WN_THREADS = 256;
constexpr int WN_SUB = 16;                       // lanes per query point: each takes every 16th face of a tile
constexpr int WN_POINTS = WN_THREADS / WN_SUB;   // 16 points per workgroup (a training step has ~16 k points: 1024 workgroups)

// one scene's mesh in a batch (vt_winding_number_scenes): device pointers and counts
struct WnScene { const float *verts; const int32_t *faces; int V, F; };

__global__ void __launch_bounds__(WN_THREADS)
winding_kernel(const float *verts, int V, const int32_t *faces, int F, const float *pts, int64_t N, float *out, const WnScene *scenes) {
    __shared__ float tri[WN_THREADS][9];
    if (scenes) {                                 // blockIdx.y = scene: its own mesh, its N points
        const WnScene sc = scenes[blockIdx.y];
        verts = sc.verts; V = sc.V; faces = sc.faces; F = sc.F;
        pts += (size_t)blockIdx.y * N * 3; out += (size_t)blockIdx.y * N;
    }
    const int sub = threadIdx.x % WN_SUB;
    const int64_t n = (int64_t)blockIdx.x * WN_POINTS + threadIdx.x / WN_SUB;
    double px = 0, py = 0, pz = 0;
    if (n < N) { px = pts[3 * n]; py = pts[3 * n + 1]; pz = pts[3 * n + 2]; }
    double sum = 0.0;
    for (int f0 = 0; f0 < F; f0 += WN_THREADS) {
        __syncthreads();
        const int f = f0 + threadIdx.x;
        if (f < F) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int vi = faces[3 * f + k];
                vi = vi < 0 ? 0 : (vi >= V ? V - 1 : vi);
                tri[threadIdx.x][3 * k] = verts[3 * vi]; tri[threadIdx.x][3 * k + 1] = verts[3 * vi + 1]; tri[threadIdx.x][3 * k + 2] = verts[3 * vi + 2];
            }
        }
        __syncthreads();
        const int cnt = min(WN_THREADS, F - f0);
        for (int j = sub; j < cnt; j += WN_SUB) {
            const float *t = tri[j];
            const double ax = t[0] - px, ay = t[1] - py, az = t[2] - pz;
            const double bx = t[3] - px, by = t[4] - py, bz = t[5] - pz;
            const double cx = t[6] - px, cy = t[7] - py, cz = t[8] - pz;
            const double la = sqrt(ax * ax + ay * ay + az * az), lb = sqrt(bx * bx + by * by + bz * bz), lc = sqrt(cx * cx + cy * cy + cz * cz);
            const double num = ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx);
            const double den = la * lb * lc + (ax * bx + ay * by + az * bz) * lc + (bx * cx + by * cy + bz * cz) * la +
                               (cx * ax + cy * ay + cz * az) * lb;
            sum += 2.0 * atan2(num, den);
        }
    }
    // the 16 partial sums of a point, combined in a fixed butterfly order (deterministic)
    for (int o = WN_SUB / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (n < N && sub == 0) out[n] = (float)(sum / (4.0 * 3.14159265358979323846));
}

}  // namespace

extern "C" {

int vt_winding_number(const float *verts, int V, const int32_t *faces, int F, const float *pts, int64_t N, float *out, void *stream) {
    if (N == 0) return 0;
    if (!verts || !faces || !pts || !out || V <= 0 || F < 0 || N < 0) return vt_fail(VT_ERR_INVALID, "vt_winding_number: bad argument");
    hipLaunchKernelGGL(winding_kernel, dim3((unsigned)((N + WN_POINTS - 1) / WN_POINTS)), dim3(WN_THREADS), 0, (hipStream_t)stream,
                       verts, V, faces, F, pts, N, out, (const WnScene *)nullptr);
    return vt_check(hipGetLastError(), "vt_winding_number");
}

int vt_winding_number_scenes(const void *scenes, int B, const float *pts, int64_t N, float *out, void *stream) {
    if (B == 0 || N == 0) return 0;
    if (!scenes || !pts || !out || B < 0 || N < 0) return vt_fail(VT_ERR_INVALID, "vt_winding_number_scenes: bad argument");
    hipLaunchKernelGGL(winding_kernel, dim3((unsigned)((N + WN_POINTS - 1) / WN_POINTS), (unsigned)B), dim3(WN_THREADS), 0, (hipStream_t)stream,
                       (const float *)nullptr, 0, (const int32_t *)nullptr, 0, pts, N, out, reinterpret_cast<const WnScene *>(scenes));
    return vt_check(hipGetLastError(), "vt_winding_number_scenes");
}

}  // extern "C"
