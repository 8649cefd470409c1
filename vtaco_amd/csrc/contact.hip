// Contact clouds of the tactile sensors from their depth images, for the VTacO (t2d) training step and generator: reference
// src/conv_onet/training.py:817-853 / generation.py:224-244 do this per scene and sensor in numpy on the host -- threshold the
// depth image against the sensor's flat reading, np.where, unproject the touched pixels through the pinhole model, keep at most
// 128 of them (np.random.randint), move them to the world with the sample's camera pose, normalise like the object cloud --
// 40 passes over 76 800 pixels per step of eight scenes (5-20 ms of host time).  Here:
//   vt_contact_scan    one workgroup per image: the touched pixels' indices in ascending order (= np.where's) and their count;
//   (host)             reads the 5 B counts, draws the reference's np.random.randint indices where a count exceeds 128 -- the
//                      draws stay on the host so that a seeded run consumes numpy's generator exactly as the reference does --
//                      and inverts the camera poses (a 4 x 4 each);
//   vt_contact_points  one thread per kept pixel: unprojection, pose, normalisation in float64 in the reference's operation
//                      order, written as float32 rows of the step's query-point tensor.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vt_common.h"
#include "vtaco_hip.h"

namespace {

constexpr int CS_THREADS = 1024;

// touched[i] = |depth[i] - origin[i]| > threshold, evaluated as numpy does (float32 - float64 -> float64)
__global__ void __launch_bounds__(CS_THREADS)
contact_scan_kernel(const float *depth, const double *origin, const unsigned char *touch, int npix, double threshold, int *index, int *count) {
    __shared__ int wave_tot[CS_THREADS / 64];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *z = depth + (size_t)img * npix;
    int *out = index + (size_t)img * npix;
    if (touch && !touch[img]) {                                        // the sensor did not touch: no cloud (the reference skips it)
        if (tid == 0) count[img] = 0;
        return;
    }
    const int per = (npix + CS_THREADS - 1) / CS_THREADS, lo = tid * per, hi = min(lo + per, npix);
    int mine = 0;
    for (int i = lo; i < hi; ++i) mine += fabs((double)z[i] - origin[i]) > threshold ? 1 : 0;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    for (int i = lo; i < hi; ++i)
        if (fabs((double)z[i] - origin[i]) > threshold) out[base++] = i;
    if (tid == CS_THREADS - 1) count[img] = base;
}

struct ContactPose { double m[9], t[3], centroid[3], scale; };         // inverse pose's linear part, translation; norm_pc_1 of the scene

__global__ void __launch_bounds__(256)
contact_points_kernel(const float *depth, const int *index, const int *sel, const int *kept, const int *row0, const ContactPose *pose,
                      int npix, int width, int height, double f, int max_points, int S, float *p_sample, long long *finger) {
    const int img = blockIdx.x, b = img / 5, t = img - 5 * b;
    const int n = kept[img];
    const ContactPose ps = pose[img];
    for (int j = threadIdx.x; j < n; j += 256) {
        const int s = sel ? sel[(size_t)img * max_points + j] : j;
        const int pix = index[(size_t)img * npix + s];
        const int px = pix % width, py = pix / width;
        const double z = (double)depth[(size_t)img * npix + pix];
        // cam = (z, -(px - w/2) z / f, -(py - h/2) z / f) as numpy evaluates it: ((-(px - w/2)) * z) / f
        const double c0 = z, c1 = (-((double)px - width / 2.0)) * z / f, c2 = (-((double)py - height / 2.0)) * z / f;
        double w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = fma(ps.m[3 * i + 2], c2, fma(ps.m[3 * i + 1], c1, ps.m[3 * i] * c0)) + ps.t[i];
        const size_t row = (size_t)b * S + row0[img] + j;
#pragma unroll
        for (int i = 0; i < 3; ++i) p_sample[row * 3 + i] = (float)((w[i] - ps.centroid[i]) / ps.scale);
        if (finger) finger[row] = t;
    }
}

}  // namespace

extern "C" {

int vt_contact_scan(const float *depth, const double *depth_origin, const unsigned char *touch_success, int n_images, int n_pixels,
                    double threshold, int *index, int *count, void *stream) {
    if (n_images == 0) return 0;
    if (!depth || !depth_origin || !index || !count || n_images < 0 || n_pixels <= 0) return vt_fail(VT_ERR_INVALID, "vt_contact_scan: bad argument");
    hipLaunchKernelGGL(contact_scan_kernel, dim3((unsigned)n_images), dim3(CS_THREADS), 0, (hipStream_t)stream, depth, depth_origin,
                       touch_success, n_pixels, threshold, index, count);
    return vt_check(hipGetLastError(), "vt_contact_scan");
}

int vt_contact_points(const float *depth, const int *index, const int *sel, const int *kept, const int *row0, const double *pose,
                      int n_images, int n_pixels, int width, int height, double fov_deg, int max_points, int S,
                      float *p_sample, long long *finger, void *stream) {
    if (n_images == 0) return 0;
    if (!depth || !index || !kept || !row0 || !pose || !p_sample || n_images < 0 || n_images % 5 || width * height != n_pixels || max_points <= 0)
        return vt_fail(VT_ERR_INVALID, "vt_contact_points: bad argument");
    const double f = height / (2.0 * tan(fov_deg * 3.14159265358979323846 / 180.0 / 2.0));
    hipLaunchKernelGGL(contact_points_kernel, dim3((unsigned)n_images), dim3(256), 0, (hipStream_t)stream, depth, index, sel, kept, row0,
                       reinterpret_cast<const ContactPose *>(pose), n_pixels, width, height, f, max_points, S, p_sample, finger);
    return vt_check(hipGetLastError(), "vt_contact_points");
}

}  // extern "C"
