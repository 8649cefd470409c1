// Tactile feature assignment (SURVEY.md section 8f "next" row 2, K11): which finger's tactile
// feature a query point receives.  Replaces the CPU glue of reference
// src/conv_onet/generation.py:186-200 (VTacOH: nearest fingertip within 0.05) and :245-255
// (VTacO: within 0.015 of a finger's <=128-point contact cloud; later fingers overwrite), which
// builds a dense c_img_all [1, nx^3, 32] on the GPU from scipy cdist masks on the CPU.  Here the
// result is ONE byte per point (finger id, 255 = none); the decode kernel looks the 32-d feature
// up in a [F,32] table, so the 2.1 GB dense tensor of the 256^3 configuration never exists.
// Distances are evaluated in double like scipy's cdist.
#include "decode_common.h"

namespace {

struct AssignArgs {
    DecodeArgs d;                 // pts / lattice description
    const float *anchors;         // [F][K][3]
    const int *count;             // [F] valid anchors per finger (<= K)
    const unsigned char *success; // [F] touch_success
    int F, K, mode;               // mode 0: nearest fingertip (K = 1), 1: any contact point within radius
    double radius;
    unsigned char *ids;           // [total]
};

__global__ void __launch_bounds__(256) tactile_assign_kernel(AssignArgs a) {
    extern __shared__ float anc[];                                 // anchors staged in LDS
    __shared__ float box[256][6];                                  // mode 1: per finger, the bounds of its valid anchors
    for (int i = threadIdx.x; i < a.F * a.K * 3; i += 256) anc[i] = a.anchors[i];
    __syncthreads();
    if (a.mode == 1 && (int)threadIdx.x < a.F) {
        // A contact cloud is a few millimetres across and the radius two lattice cells: a point outside the cloud's bounds grown by
        // the radius cannot be within it of any anchor (|d| >= |dx|), so all but a few hundred of the 2 M lattice points skip the
        // finger's anchor loop after six compares (1.2 ms -> 0.03 ms at 128^3; the result is the loop's, bit for bit)
        const float *q = anc + (size_t)threadIdx.x * a.K * 3;
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int k = 0; k < a.count[threadIdx.x]; ++k)
            for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], q[3 * k + c]); hi[c] = fmaxf(hi[c], q[3 * k + c]); }
        for (int c = 0; c < 3; ++c) { box[threadIdx.x][c] = lo[c]; box[threadIdx.x][3 + c] = hi[c]; }
    }
    __syncthreads();
    const double grow = a.radius * (1.0 + 1e-9);
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < a.d.total; g += gridDim.x * 256) {
        float px, py, pz;
        const uint32_t b = g / a.d.N;
        point_of(a.d, g, g - b * a.d.N, px, py, pz);
        int id = 255;
        if (a.mode == 0) {
            double best = 1e300;
            int arg = 0;
            for (int f = 0; f < a.F; ++f) {
                const double dx = (double)px - (double)anc[f * 3], dy = (double)py - (double)anc[f * 3 + 1], dz = (double)pz - (double)anc[f * 3 + 2];
                const double dist = sqrt(dx * dx + dy * dy + dz * dz);
                if (dist < best) { best = dist; arg = f; }          // first minimum, like np.argmin
            }
            if (best < a.radius && a.success[arg]) id = arg;
        } else {
            for (int f = 0; f < a.F; ++f) {
                if (!a.success[f]) continue;
                if ((double)px < (double)box[f][0] - grow || (double)px > (double)box[f][3] + grow ||
                    (double)py < (double)box[f][1] - grow || (double)py > (double)box[f][4] + grow ||
                    (double)pz < (double)box[f][2] - grow || (double)pz > (double)box[f][5] + grow) continue;
                const float *q = anc + (size_t)f * a.K * 3;
                bool hit = false;
                for (int k = 0; k < a.count[f] && !hit; ++k) {
                    const double dx = (double)q[3 * k] - (double)px, dy = (double)q[3 * k + 1] - (double)py, dz = (double)q[3 * k + 2] - (double)pz;
                    hit = sqrt(dx * dx + dy * dy + dz * dz) < a.radius;
                }
                if (hit) id = f;                                    // later fingers overwrite earlier ones
            }
        }
        a.ids[g] = (unsigned char)id;
    }
}

}  // namespace

extern "C" int vt_tactile_assign(const float *pts, int B, int64_t N, int lattice_nx, float lattice_box, int64_t lattice_first,
                                 const float *anchors, const int *count, const unsigned char *success, int F, int K,
                                 int mode, double radius, unsigned char *ids, void *stream) {
    if (!anchors || !count || !success || !ids) return vt_fail(VT_ERR_INVALID, "vt_tactile_assign: null argument");
    if (B <= 0 || N <= 0 || F <= 0 || F > 254 || K <= 0 || (mode != 0 && mode != 1)) return vt_fail(VT_ERR_INVALID, "vt_tactile_assign: bad argument");
    if (mode == 0 && K != 1) return vt_fail(VT_ERR_INVALID, "vt_tactile_assign: nearest-fingertip mode takes one anchor per finger");
    if ((int64_t)B * N >= (int64_t)1 << 31) return vt_fail(VT_ERR_UNSUPPORTED, "vt_tactile_assign: B*N must be < 2^31");
    if (!pts && lattice_nx < 2) return vt_fail(VT_ERR_INVALID, "vt_tactile_assign: lattice mode needs nx >= 2");
    const size_t lds = (size_t)F * K * 3 * sizeof(float);
    if (lds > 64 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_tactile_assign: anchor set does not fit 64 KiB of LDS");
    AssignArgs a;
    a.d.c_direct = nullptr; a.d.brick = 0; a.d.grid = nullptr; a.d.pts = pts; a.d.c_img = nullptr; a.d.blob = nullptr;
    a.d.out = nullptr; a.d.out2 = nullptr; a.d.save = nullptr; a.d.cimg_ids = nullptr; a.d.cimg_table = nullptr; a.d.cimg_nf = 0;
    a.d.N = (uint32_t)N; a.d.total = (uint32_t)((int64_t)B * N); a.d.lattice_first = (uint32_t)lattice_first;
    a.d.R = 2; a.d.nx = lattice_nx; a.d.box = lattice_box; a.d.divisor = 1.0f;
    a.anchors = anchors; a.count = count; a.success = success; a.F = F; a.K = K; a.mode = mode; a.radius = radius; a.ids = ids;
    size_t g = ((size_t)a.d.total + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(tactile_assign_kernel, dim3((unsigned)g), dim3(256), lds, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_tactile_assign");
}
