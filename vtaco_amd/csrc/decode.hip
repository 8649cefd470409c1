// Fused trilinear feature gather + per-point conditioned ResNet MLP (forward),
// gfx950 only.  Replaces LocalDecoder.forward / forward_img / forward_contact
// (reference src/conv_onet/models/decoder.py:135-161, 71-103, 105-133).
//
// Mapping.  One wave owns a tile of 32 query points.  Every dense layer
// y = W x (32x32) is computed transposed on the matrix core,
//     D[out][point] += A[out][k] * B[k][point]
// (exact f32: v_mfma_f32_32x32x2_f32, described here; split-bf16: decode_common.h),
// so the accumulator of one layer (point on the lane, 16 output channels in the
// 16 accumulator registers) IS the B operand of the next layer with no lane
// movement: register r of lane-half h holds channel o(r,h) = (r&3)+8(r>>2)+4h,
// and the weight fragments are pre-permuted into that k order by vt_decoder_pack.
// Weights live in LDS (one ds_read_b32 per MFMA); biases enter as the initial
// accumulator; the residual adds are the accumulator chain itself.
//
// 242 MFMAs per 32 points (258 with tactile concat): 15 dense 32x32 layers x 16
// + fc_p (K=3 padded to 4: 2).  The trilinear gather reads the channels-last grid:
// lane (p,h) loads channels 16h..16h+15 of each of its point's 8 corners.
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "decode_common.h"

namespace {

// ---- everything after the gather: fc_p / fc_p_img, 5 x (fc_c + ResnetBlockFC), output heads ----
// c: the sampled features in gather layout (register s of lane-half h = channel 16h+s).
template <bool SAVE, int P>
__device__ __forceinline__ void mlp_and_heads(const DecodeArgs &a, const float *L, const f32x16 &c, float px, float py, float pz,
                                              uint32_t g, bool live, int lane, int h, bool with_img, unsigned &rmax) {
    const size_t slot = (size_t)a.total * 32;           // one saved tensor
    float *srow = SAVE ? a.save + (size_t)g * 32 : nullptr;
    if (SAVE && live) store_gather16(srow, c, h);        // slot 0: c

    // ---- net = fc_p(p) + fc_c[0](c) (+ fc_p_img's c_img columns) ----
    f32x16 net = load_frag16(L + VT_OFF_BIAS + 0 * 32 + h * 16);
    {
        const float k0 = h ? py : px;          // k = 2s+h : s=0 -> x|y
        const float k1 = h ? 0.0f : pz;        //            s=1 -> z|pad
        net = mfma(L[VT_OFF_WP + lane], k0, net);
        net = mfma(L[VT_OFF_WP + 64 + lane], k1, net);
    }
    if (with_img) {
        if (a.cimg_ids) {
            // tactile feature by finger id: most tiles touch no finger and skip the 16 MFMAs
            const unsigned id = a.cimg_ids[g];
            if (__ballot(id < a.cimg_nf) != 0ull) {
                f32x16 ci;
#pragma unroll
                for (int s = 0; s < 16; ++s) ci[s] = 0.0f;
                if (id < a.cimg_nf) ci = load_frag16(a.cimg_table + (size_t)id * 32 + 16 * h);
                if constexpr (P != 0) net = dense32s<P>(net, L + VT_OFF_WPI, split16<false, P>(ci), lane);
                else net = dense32<false>(net, L + VT_OFF_WPI, ci, lane);
            }
        } else {
            const f32x16 ci = load_frag16(a.c_img + (size_t)g * 32 + 16 * h);
            if constexpr (P != 0) net = dense32s<P>(net, L + VT_OFF_WPI, split16<false, P>(ci), lane);
            else net = dense32<false>(net, L + VT_OFF_WPI, ci, lane);
        }
    }
    if constexpr (P != 0) {
        // ---- split 16-bit layers: c is split once and feeds all five fc_c ----
        const SplitP<P> cs = split16<false, P>(c);
        net = dense32s<P>(net, L + VT_OFF_WL, cs, lane);
#pragma unroll 1
        for (int i = 0; i < 5; ++i) {
            const float *wl = L + VT_OFF_WL + (1 + 3 * i) * 1024;
            f32x16 hid = load_frag16(L + VT_OFF_BIAS + (1 + 2 * i) * 32 + h * 16);
            // same accumulation order as the two-brick kernel: net + cond(k-step 0) + bias + cond(k-step 1) + fc_1(..),
            // the conditioning MFMAs placed where the VALU is busy splitting relu(net) / relu(hid)
            const SplitP<P> sn = split16<true, P>(net);
            if constexpr (P == 2) range_track(rmax, __builtin_bit_cast(u32x4, sn.hi[0])[0]);
            if (i < 4) net = dense32s_half<P>(net, wl + 2048, cs, lane, 0);
            const f32x16 bb = load_frag16(L + VT_OFF_BIAS + (2 + 2 * i) * 32 + h * 16);
            net = net + bb;
            hid = dense32s<P>(hid, wl, sn, lane);
            const SplitP<P> sh = split16<true, P>(hid);
            if constexpr (P == 2) range_track(rmax, __builtin_bit_cast(u32x4, sh.hi[0])[0]);
            if (i < 4) net = dense32s_half<P>(net, wl + 2048, cs, lane, 1);
            net = dense32s<P>(net, wl + 1024, sh, lane);
        }
    } else {
    net = dense32<false>(net, L + VT_OFF_WL, c, lane);

    // ---- 5 x (ResnetBlockFC + next block's fc_c) ----
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
        const float *wl = L + VT_OFF_WL + (1 + 3 * i) * 1024;
        f32x16 hid = load_frag16(L + VT_OFF_BIAS + (1 + 2 * i) * 32 + h * 16);
        if (SAVE && live) store_acc16(srow + (1 + i) * slot, relu16(net), h);    // slots 1..5: relu(x_i)
        hid = dense32<true>(hid, wl, net, lane);
        if (SAVE && live) store_acc16(srow + (6 + i) * slot, relu16(hid), h);    // slots 6..10: relu(h_i)
        net = dense32<true>(net, wl + 1024, hid, lane);
        if (i < 4) net = dense32<false>(net, wl + 2048, c, lane);
        const f32x16 bb = load_frag16(L + VT_OFF_BIAS + (2 + 2 * i) * 32 + h * 16);
        net = net + bb;
    }
    }

    if (SAVE && live) store_acc16(srow + 11 * slot, relu16(net), h);            // slot 11: relu(net_5)
    // ---- heads: out = fc_out(relu(net)) ----
    {
        const f32x16 wo = load_frag16(L + VT_OFF_OUT + h * 16);
        float acc = 0.0f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = fmaf(relu1(net[s]), wo[s], acc);
        acc += __shfl_xor(acc, 32);
        acc += L[VT_OFF_OUT + 64];
        if (live && h == 0) a.out[g] = acc;
        if (a.out2) {
            const f32x16 wo2 = load_frag16(L + VT_OFF_OUT + 32 + h * 16);
            float acc2 = 0.0f;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc2 = fmaf(relu1(net[s]), wo2[s], acc2);
            acc2 += __shfl_xor(acc2, 32);
            acc2 += L[VT_OFF_OUT + 65];
            if (live && h == 0) a.out2[g] = acc2;
        }
    }
}

// P selects the dense layers (decode_common.h): 0 exact f32, 1 split-bf16, 2 split-f16, and expects the
// blob of vt_decoder_pack / _bf16x3 / _f16x3; everything around the 16 dense layers is shared.
template <int THREADS, bool SAVE, int P>
__global__ void __launch_bounds__(THREADS, VT_WAVES_PER_SIMD)
decode_fwd_kernel(DecodeArgs a) {
    static_assert(!(SAVE && P != 0), "the training forward keeps the exact-f32 layers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // stage the packed weights (identical for every block) into LDS
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.blob);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = threadIdx.x; i < VT_BLOB_FLOATS / 4; i += THREADS) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int pl = lane & 31;
    const int h = lane >> 5;
    // wave-uniform by construction: tell the compiler, so tile/brick index maths runs on the SALU
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WPB = THREADS / 64;
    const uint32_t ntiles = (a.total + 31u) >> 5;
    const bool with_img = a.c_img != nullptr || a.cimg_ids != nullptr;
    const int R = a.R;

    // XCD-aware tile order: workgroups b and b+8 share an XCD (and its 4 MiB L2), so each XCD
    // gets ONE contiguous eighth of the tiles -- at 128^3 a 16-plane x-slab of the lattice, i.e.
    // an eighth of every grid row -- instead of every XCD pulling the whole 33.5 MB grid
    // through its own L2.  Placement only changes speed, never results.
    uint32_t t_begin = 0, t_end = ntiles, w_idx = blockIdx.x * WPB + wave, w_cnt = gridDim.x * WPB;
    if ((gridDim.x & 7u) == 0 && ntiles >= 8u * WPB) {
        const uint32_t chunk = (ntiles + 7u) >> 3, xcd = blockIdx.x & 7u;
        t_begin = min(xcd * chunk, ntiles);
        t_end = min(t_begin + chunk, ntiles);
        w_idx = (blockIdx.x >> 3) * WPB + wave;
        w_cnt = (gridDim.x >> 3) * WPB;
    }
    unsigned rmax = 0;                                                   // range guard (decode_common.h)
    for (uint32_t tile = t_begin + w_idx; tile < t_end; tile += w_cnt) {
        // Re-derive the LDS base every tile behind an opaque asm so the (loop-invariant)
        // weight reads are not hoisted out of the tile loop into ~120 extra VGPRs.
        unsigned lds_off = 0;
        asm volatile("" : "+v"(lds_off));
        const float *L = lds + lds_off;
        uint32_t g, b;
        bool live = true;
        float px, py, pz;
        if (a.brick) {
            // lattice tile = 2 x 4 x 4 brick of points (x,y,z): 18-27 distinct corner lines per
            // tile instead of 68 for 32 points along z, and 16-B runs in the output
            const uint32_t tpb = a.N >> 5, q4 = (uint32_t)a.nx >> 2;
            b = tile / tpb;
            const uint32_t t = tile - b * tpb;
            const uint32_t pp = t / (q4 * q4), rem = t - pp * q4 * q4;
            const uint32_t by = rem / q4, bz = rem - by * q4;
            const uint32_t ixl = 2u * pp + (uint32_t)(pl >> 4), iy = 4u * by + (uint32_t)((pl >> 2) & 3), iz = 4u * bz + (uint32_t)(pl & 3);
            const uint32_t n = (ixl * (uint32_t)a.nx + iy) * (uint32_t)a.nx + iz;
            g = b * a.N + n;
            lattice_point(a, a.lattice_first / ((uint32_t)a.nx * (uint32_t)a.nx) + ixl, iy, iz, px, py, pz);
        } else {
            g = tile * 32u + pl;
            live = g < a.total;
            if (!live) g = a.total - 1u;
            b = g / a.N;
            point_of(a, g, g - b * a.N, px, py, pz);
        }

        // ---- trilinear gather: c[s] = feature channel 16h+s of this lane's point ----
        f32x16 c;
#pragma unroll
        for (int s = 0; s < 16; ++s) c[s] = 0.0f;
        if (a.c_direct) {
            c = load_frag16(a.c_direct + (size_t)g * 32 + 16 * h);
        } else {
            const Tri t = tri_setup(px, py, pz, a.divisor, R);
            const float *gb = a.grid + (size_t)b * R * R * R * 32 + 16 * h;
            // one z-plane (4 corners, 16 x 16-byte loads per lane) at a time: the scheduling
            // barrier keeps the second plane's loads from being hoisted above the first
            // plane's FMAs, which would double the VGPRs held by loads in flight.
            auto plane = [&](int zz, float wz) {
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int yy = dy ? t.y1 : t.y0;
                    const float wy = dy ? t.wy1 : t.wy0;
                    const size_t row = ((size_t)zz * R + yy) * R;
                    const f32x16 v0 = load_frag16(gb + (row + t.x0) * 32);
                    const f32x16 v1 = load_frag16(gb + (row + t.x1) * 32);
                    const float w0 = (t.wx0 * wy) * wz;
                    const float w1 = (t.wx1 * wy) * wz;
#pragma unroll
                    for (int s = 0; s < 16; ++s) c[s] = fmaf(v0[s], w0, c[s]);
#pragma unroll
                    for (int s = 0; s < 16; ++s) c[s] = fmaf(v1[s], w1, c[s]);
                }
            };
            plane(t.z0, t.wz0);
            pin16(c);
            __builtin_amdgcn_sched_barrier(0);
            plane(t.z1, t.wz1);
            pin16(c);
            __builtin_amdgcn_sched_barrier(0);
        }
        mlp_and_heads<SAVE, P>(a, L, c, px, py, pz, g, live, lane, h, with_img, rmax);
    }
    if constexpr (P == 2) range_report(rmax, a.status);
}

// ---- lattice decode with the gather staged through LDS -------------------------------------
// The direct gather above is bound by the L1's address processing: each lane's 16-byte load is
// its own cache access (46 accesses per wave-instruction measured, 97.5 M per 128^3 launch, one
// per cycle per CU = the whole 0.17 ms).  Here every wave first copies the 3 x 4 x 4 voxel
// footprint of its 2 x 4 x 4 point brick (48 rows of 128 B; 16 runs of 384 contiguous bytes)
// into a private LDS image with six fully coalesced loads, issued one tile ahead so their latency
// hides behind the previous tile's MLP, and gathers the 8 corners from LDS.  Per-axis corner
// indices and weights come from a table (all three axes share it: same nx, box, R), so the
// coordinate maths -- three exact divisions per point in the direct path -- is done nx times
// per block, not per point.  Same FMA order as the direct path: results are bit-identical.
// Host-side conditions (decode_launch): brick-aligned lattice, voxels per lattice step < 2/3
// (the footprint bound), R >= 4, table fits.
constexpr int ST_ROW_BYTES = 144;                 // 128 B of channels + 16 B pad: rotates the LDS banks row by row
constexpr int ST_ROWS = 48;
constexpr int ST_WAVE_FLOATS = ST_ROWS * ST_ROW_BYTES / 4;
constexpr int ST_THREADS = 768;                   // 12 waves: blob + table + 12 images fit the CU's 160 KiB

struct AxisEnt {                                   // one lattice index along one axis
    int i0, i1;                                    // lower / upper corner voxel (border-clamped)
    float w0, w1;                                  // their weights (w1 = 0 where ATen skips the corner)
};

template <int P>
__global__ void __launch_bounds__(ST_THREADS)
decode_fwd_staged_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.blob);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = threadIdx.x; i < VT_BLOB_FLOATS / 4; i += ST_THREADS) dst[i] = src[i];
    }
    const int R = a.R;
    AxisEnt *tab = reinterpret_cast<AxisEnt *>(lds + VT_BLOB_FLOATS);
    for (int i = threadIdx.x; i < a.nx; i += ST_THREADS) {
        float p, unused0, unused1;
        lattice_point(a, (uint32_t)i, 0u, 0u, p, unused0, unused1);
        const float f = grid_coord(p, a.divisor, R);
        const float f0 = floorf(f);
        AxisEnt e;
        e.i0 = (int)f0;
        e.w0 = (f0 + 1.0f) - f;
        e.i1 = min(e.i0 + 1, R - 1);
        e.w1 = (e.i0 + 1 <= R - 1) ? f - f0 : 0.0f;
        tab[i] = e;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int pl = lane & 31;
    const int h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WPB = ST_THREADS / 64;
    const uint32_t ntiles = a.total >> 5;
    const bool with_img = a.c_img != nullptr || a.cimg_ids != nullptr;
    const uint32_t nx = (uint32_t)a.nx;
    const uint32_t tpb = a.N >> 5, q4 = nx >> 2;
    const uint32_t plane0 = a.lattice_first / (nx * nx);
    char *stage = reinterpret_cast<char *>(lds + VT_BLOB_FLOATS + 4 * a.nx) + wave * (ST_WAVE_FLOATS * 4);

    // the six 1-KiB pieces of a footprint: piece k, lane -> 16-byte chunk m = 64k + lane of the
    // 16 runs x 24 chunks; run = (dz, dy), chunk q = 8 dx + (16-byte column)
    uint32_t src_off[6], dst_off[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int m = 64 * k + lane, run = m / 24, q = m - 24 * run;
        const int dz = run >> 2, dy = run & 3;
        src_off[k] = (uint32_t)((dz * R + dy) * R) * 128u + (uint32_t)q * 16u;
        dst_off[k] = (uint32_t)(((dz * 4 + dy) * 3 + (q >> 3)) * ST_ROW_BYTES + (q & 7) * 16);
    }

    uint32_t t_begin = 0, t_end = ntiles, w_idx = blockIdx.x * WPB + wave, w_cnt = gridDim.x * WPB;
    if ((gridDim.x & 7u) == 0 && ntiles >= 8u * WPB) {                  // XCD-aware order, as decode_fwd_kernel
        const uint32_t chunk = (ntiles + 7u) >> 3, xcd = blockIdx.x & 7u;
        t_begin = min(xcd * chunk, ntiles);
        t_end = min(t_begin + chunk, ntiles);
        w_idx = (blockIdx.x >> 3) * WPB + wave;
        w_cnt = (gridDim.x >> 3) * WPB;
    }

    // brick -> first lattice indices (wave-uniform)
    auto brick_of = [&](uint32_t tile, uint32_t &b, uint32_t &X0, uint32_t &Y0, uint32_t &Z0) {
        b = tile / tpb;
        const uint32_t t = tile - b * tpb;
        const uint32_t pp = t / (q4 * q4), rem = t - pp * q4 * q4;
        const uint32_t by = rem / q4, bz = rem - by * q4;
        X0 = 2u * pp; Y0 = 4u * by; Z0 = 4u * bz;
    };
    // footprint origin (clamped so that the 3 x 4 x 4 block stays inside the grid) and its loads
    f32x4 pre[6];
    int ox = 0, oy = 0, oz = 0;
    auto fetch = [&](uint32_t tile, int &fx, int &fy, int &fz) {
        uint32_t b, X0, Y0, Z0;
        brick_of(tile, b, X0, Y0, Z0);
        fx = min(__builtin_amdgcn_readfirstlane(tab[plane0 + X0].i0), R - 3);
        fy = min(__builtin_amdgcn_readfirstlane(tab[Y0].i0), R - 4);
        fz = min(__builtin_amdgcn_readfirstlane(tab[Z0].i0), R - 4);
        const uint64_t bp = reinterpret_cast<uint64_t>(a.grid + ((((size_t)b * R + fz) * R + fy) * R + fx) * 32);
        // wave-uniform by construction: keep it in SGPRs so the loads use the saddr + 32-bit voffset form
        const char *base = reinterpret_cast<const char *>(
            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bp >> 32)) << 32) |
            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bp));
#pragma unroll
        for (int k = 0; k < 6; ++k) pre[k] = *reinterpret_cast<const f32x4 *>(base + src_off[k]);
    };

    unsigned rmax = 0;                                                   // range guard (decode_common.h)
    uint32_t tile = t_begin + w_idx;
    if (tile < t_end) fetch(tile, ox, oy, oz);
    for (; tile < t_end; tile += w_cnt) {
        unsigned lds_off = 0;
        asm volatile("" : "+v"(lds_off));                                // see decode_fwd_kernel
        const float *L = lds + lds_off;

        // ---- this tile's footprint: registers -> the wave's LDS image ----
#pragma unroll
        for (int k = 0; k < 6; ++k) *reinterpret_cast<f32x4 *>(stage + dst_off[k]) = pre[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        uint32_t b, X0, Y0, Z0;
        brick_of(tile, b, X0, Y0, Z0);
        const uint32_t ixl = X0 + (uint32_t)(pl >> 4), iy = Y0 + (uint32_t)((pl >> 2) & 3), iz = Z0 + (uint32_t)(pl & 3);
        const uint32_t g = b * a.N + (ixl * nx + iy) * nx + iz;
        float px, py, pz;
        lattice_point(a, plane0 + ixl, iy, iz, px, py, pz);
        const AxisEnt ex = tab[plane0 + ixl], ey = tab[iy], ez = tab[iz];

        // ---- trilinear gather from the image (same corner and FMA order as the direct path) ----
        f32x16 c;
#pragma unroll
        for (int s = 0; s < 16; ++s) c[s] = 0.0f;
        const int rx0 = ex.i0 - ox, rx1 = ex.i1 - ox;
        const int ry0 = (ey.i0 - oy) * 3, ry1 = (ey.i1 - oy) * 3;
        const int rz0 = (ez.i0 - oz) * 12, rz1 = (ez.i1 - oz) * 12;
        const char *img = stage + 64 * h;
        auto plane = [&](int rz, float wz) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int ry = dy ? ry1 : ry0;
                const float wy = dy ? ey.w1 : ey.w0;
                const f32x16 v0 = load_frag16(reinterpret_cast<const float *>(img + (rz + ry + rx0) * ST_ROW_BYTES));
                const f32x16 v1 = load_frag16(reinterpret_cast<const float *>(img + (rz + ry + rx1) * ST_ROW_BYTES));
                const float w0 = (ex.w0 * wy) * wz;
                const float w1 = (ex.w1 * wy) * wz;
#pragma unroll
                for (int s = 0; s < 16; ++s) c[s] = fmaf(v0[s], w0, c[s]);
#pragma unroll
                for (int s = 0; s < 16; ++s) c[s] = fmaf(v1[s], w1, c[s]);
            }
        };
        plane(rz0, ez.w0);
        pin16(c);
        __builtin_amdgcn_sched_barrier(0);
        plane(rz1, ez.w1);
        pin16(c);
        __builtin_amdgcn_sched_barrier(0);

        // ---- next tile's footprint goes into flight before this tile's MLP ----
        if (tile + w_cnt < t_end) fetch(tile + w_cnt, ox, oy, oz);

        mlp_and_heads<false, P>(a, L, c, px, py, pz, g, true, lane, h, with_img, rmax);
    }
    if constexpr (P == 2) range_report(rmax, a.status);
}

// ---- two bricks per wave (split-bf16 lattice: visual-only, tactile concat, finger ids) -------------
// Same staged gather, but a wave owns a 2 x 4 x 8 double brick = two 32-point MFMA column groups A and B
// that share every weight fragment (half the LDS weight reads per point) and whose dependent
// MFMA / split chains interleave inside the wave: while A's six MFMAs of a layer are in the matrix
// pipe, B's relu + hi/lo split issues on the VALU, without relying on another wave being in the
// complementary phase.  Footprint 3 x 4 x 6 voxels (72 rows: 24 runs of 384 bytes, nine coalesced
// loads); eight waves per CU use the same LDS as the twelve single-brick images.  Same per-point
// operation sequence as the other paths; the logits agree with them to the last bit for ~96 % of the
// points and to 1 ulp for the rest (running the two MLPs one after the other instead of interleaved is
// bit-identical -- and no faster).  Conditions: nx % 8 == 0, voxels per step < 0.55.
constexpr int ST2_ROWS = 72;
constexpr int ST2_WAVE_BYTES = ST2_ROWS * ST_ROW_BYTES;
constexpr int ST2_THREADS = 512;                  // 8 waves per CU
constexpr int ST2_PIECES = 9;

template <int P>
__device__ __forceinline__ void dense32s2(f32x16 &accA, f32x16 &accB, const float *wl, const SplitP<P> &xA, const SplitP<P> &xB, int lane) {
    const mx8<P> *w = reinterpret_cast<const mx8<P> *>(wl);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const mx8<P> wh = w[s * 64 + lane], wo = w[(2 + s) * 64 + lane];
        accA = mfma_s(wo, xA.hi[s], accA);
        accB = mfma_s(wo, xB.hi[s], accB);
        accA = mfma_s(wh, xA.lo[s], accA);
        accB = mfma_s(wh, xB.lo[s], accB);
        accA = mfma_s(wh, xA.hi[s], accA);
        accB = mfma_s(wh, xB.hi[s], accB);
    }
}

// one k-step (s = 0 or 1) of dense32s2: six of the layer's twelve MFMAs
template <int P>
__device__ __forceinline__ void dense32s2_half(f32x16 &accA, f32x16 &accB, const float *wl, const SplitP<P> &xA, const SplitP<P> &xB,
                                               int lane, int s) {
    const mx8<P> *w = reinterpret_cast<const mx8<P> *>(wl);
    const mx8<P> wh = w[s * 64 + lane], wo = w[(2 + s) * 64 + lane];
    accA = mfma_s(wo, xA.hi[s], accA);
    accB = mfma_s(wo, xB.hi[s], accB);
    accA = mfma_s(wh, xA.lo[s], accA);
    accB = mfma_s(wh, xB.lo[s], accB);
    accA = mfma_s(wh, xA.hi[s], accA);
    accB = mfma_s(wh, xB.hi[s], accB);
}

// exact-f32 layer for two column groups: one weight read per k-step feeds both chains
template <bool RELU>
__device__ __forceinline__ void dense32x2(f32x16 &accA, f32x16 &accB, const float *wl, const f32x16 &xA, const f32x16 &xB, int lane) {
    f32x16 bA = xA, bB = xB;
    if (RELU) {
#pragma unroll
        for (int s = 0; s < 16; ++s) { bA[s] = relu1(xA[s]); bB[s] = relu1(xB[s]); }
        asm volatile("" : "+v"(bA), "+v"(bB));
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float w = wl[s * 64 + lane];
        accA = mfma(w, bA[s], accA);
        accB = mfma(w, bB[s], accB);
    }
}

template <int P>
__global__ void __launch_bounds__(ST2_THREADS)
decode_fwd_staged2_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ClockStamp stamp = clock_begin(a.clk);
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.blob);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = threadIdx.x; i < VT_BLOB_FLOATS / 4; i += ST2_THREADS) dst[i] = src[i];
    }
    const int R = a.R;
    AxisEnt *tab = reinterpret_cast<AxisEnt *>(lds + VT_BLOB_FLOATS);
    for (int i = threadIdx.x; i < a.nx; i += ST2_THREADS) {
        float p, unused0, unused1;
        lattice_point(a, (uint32_t)i, 0u, 0u, p, unused0, unused1);
        const float f = grid_coord(p, a.divisor, R);
        const float f0 = floorf(f);
        AxisEnt e;
        e.i0 = (int)f0;
        e.w0 = (f0 + 1.0f) - f;
        e.i1 = min(e.i0 + 1, R - 1);
        e.w1 = (e.i0 + 1 <= R - 1) ? f - f0 : 0.0f;
        tab[i] = e;
    }
    // the workgroup's tile counter and the waves' end stamps: VT_TAIL_LDS_BYTES behind the images
    unsigned *claim_ctr = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(lds + VT_BLOB_FLOATS + 4 * a.nx) + (ST2_THREADS / 64) * ST2_WAVE_BYTES);
    if (threadIdx.x == 0) *claim_ctr = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int pl = lane & 31;
    const int h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WPB = ST2_THREADS / 64;
    const uint32_t ntiles = a.total >> 6;                                // double bricks
    const uint32_t nx = (uint32_t)a.nx;
    const uint32_t tpb = a.N >> 6, q4 = nx >> 2, q8 = nx >> 3;
    const uint32_t plane0 = a.lattice_first / (nx * nx);
    char *stage = reinterpret_cast<char *>(lds + VT_BLOB_FLOATS + 4 * a.nx) + wave * ST2_WAVE_BYTES;

    uint32_t src_off[ST2_PIECES], dst_off[ST2_PIECES];
#pragma unroll
    for (int k = 0; k < ST2_PIECES; ++k) {
        const int m = 64 * k + lane, run = m / 24, q = m - 24 * run;      // 24 runs (dz 0..5, dy 0..3) x 24 chunks
        const int dz = run >> 2, dy = run & 3;
        src_off[k] = (uint32_t)((dz * R + dy) * R) * 128u + (uint32_t)q * 16u;
        dst_off[k] = (uint32_t)(((dz * 4 + dy) * 3 + (q >> 3)) * ST_ROW_BYTES + (q & 7) * 16);
    }

    uint32_t t_begin = 0, t_end = ntiles, w_idx = blockIdx.x * WPB + wave, w_cnt = gridDim.x * WPB;
    if ((gridDim.x & 7u) == 0 && ntiles >= 8u * WPB) {                  // XCD-aware order, as decode_fwd_kernel
        const uint32_t chunk = (ntiles + 7u) >> 3, xcd = blockIdx.x & 7u;
        t_begin = min(xcd * chunk, ntiles);
        t_end = min(t_begin + chunk, ntiles);
        w_idx = (blockIdx.x >> 3) * WPB + wave;
        w_cnt = (gridDim.x >> 3) * WPB;
    }
    auto brick_of = [&](uint32_t tile, uint32_t &b, uint32_t &X0, uint32_t &Y0, uint32_t &Z0) {
        b = tile / tpb;
        const uint32_t t = tile - b * tpb;
        const uint32_t pp = t / (q4 * q8), rem = t - pp * q4 * q8;
        const uint32_t by = rem / q8, bz = rem - by * q8;
        X0 = 2u * pp; Y0 = 4u * by; Z0 = 8u * bz;
    };
    f32x4 pre[ST2_PIECES];
    int ox = 0, oy = 0, oz = 0;
    auto fetch = [&](uint32_t tile, int &fx, int &fy, int &fz) {
        uint32_t b, X0, Y0, Z0;
        brick_of(tile, b, X0, Y0, Z0);
        fx = min(__builtin_amdgcn_readfirstlane(tab[plane0 + X0].i0), R - 3);
        fy = min(__builtin_amdgcn_readfirstlane(tab[Y0].i0), R - 4);
        fz = min(__builtin_amdgcn_readfirstlane(tab[Z0].i0), R - 6);
        const uint64_t bp = reinterpret_cast<uint64_t>(a.grid + ((((size_t)b * R + fz) * R + fy) * R + fx) * 32);
        const char *base = reinterpret_cast<const char *>(
            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bp >> 32)) << 32) |
            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bp));
#pragma unroll
        for (int k = 0; k < ST2_PIECES; ++k) pre[k] = *reinterpret_cast<const f32x4 *>(base + src_off[k]);
    };

    // B operand of the bias MFMA: ones in the three k-slots that carry the bias's hi / mid / lo parts
    auto make_ones = [&]() {
        if constexpr (P == 2) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((h == 0 && e < 3) ? 1.0f : 0.0f);
            return o;
        } else {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)((h == 0 && e < 3) ? 1.0f : 0.0f);
            return o;
        }
    };
    const auto ones = make_ones();

    unsigned rmax = 0;                                                   // range guard (decode_common.h)
    // tiles are claimed from an LDS counter, not pre-assigned: the two waves of a SIMD do not share it evenly (VALU issue is
    // arbitrated by age), and with fixed shares the older wave finished early and left its partner to run alone -- see
    // decode_st3.h.  The workgroup's set of tiles is the one the fixed assignment gave it.
    const uint32_t wg_first = t_begin + w_idx - (uint32_t)wave;
    unsigned fixed_next = (unsigned)wave;
    auto claim = [&]() -> uint32_t {
        unsigned i = 0;
        if (!a.claim) { i = fixed_next; fixed_next += (unsigned)WPB; }                      // A/B: the fixed share (wave w: w, w + WPB, ...)
        else {
            if (lane == 0) i = __hip_atomic_fetch_add(claim_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            i = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
        }
        const uint32_t t = wg_first + (i % (unsigned)WPB) + (i / (unsigned)WPB) * w_cnt;
        return i < 0x10000u ? t : t_end;
    };
    uint32_t tile = claim();
    if (tile < t_end) fetch(tile, ox, oy, oz);
    uint32_t next_tile = t_end;
    for (; tile < t_end; tile = next_tile) {
        unsigned lds_off = 0;
        asm volatile("" : "+v"(lds_off));
        const float *L = lds + lds_off;
#pragma unroll
        for (int k = 0; k < ST2_PIECES; ++k) *reinterpret_cast<f32x4 *>(stage + dst_off[k]) = pre[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        uint32_t b, X0, Y0, Z0;
        brick_of(tile, b, X0, Y0, Z0);
        const uint32_t ixl = X0 + (uint32_t)(pl >> 4), iy = Y0 + (uint32_t)((pl >> 2) & 3), izA = Z0 + (uint32_t)(pl & 3), izB = izA + 4u;
        const uint32_t gA = b * a.N + (ixl * nx + iy) * nx + izA, gB = gA + 4u;
        float px, py, pzA, pzB, unused0, unused1;
        lattice_point(a, plane0 + ixl, iy, izA, px, py, pzA);
        lattice_point(a, plane0 + ixl, iy, izB, unused0, unused1, pzB);
        const AxisEnt ex = tab[plane0 + ixl], ey = tab[iy], ezA = tab[izA], ezB = tab[izB];
        const int rx0 = ex.i0 - ox, rx1 = ex.i1 - ox;
        const int ry0 = (ey.i0 - oy) * 3, ry1 = (ey.i1 - oy) * 3;
        const char *img = stage + 64 * h;
        auto gather = [&](const AxisEnt &ez) {
            f32x16 c;
#pragma unroll
            for (int s = 0; s < 16; ++s) c[s] = 0.0f;
            const int rz0 = (ez.i0 - oz) * 12, rz1 = (ez.i1 - oz) * 12;
            auto plane = [&](int rz, float wz) {
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int ry = dy ? ry1 : ry0;
                    const float wy = dy ? ey.w1 : ey.w0;
                    const f32x16 v0 = load_frag16(reinterpret_cast<const float *>(img + (rz + ry + rx0) * ST_ROW_BYTES));
                    const f32x16 v1 = load_frag16(reinterpret_cast<const float *>(img + (rz + ry + rx1) * ST_ROW_BYTES));
                    const float w0 = (ex.w0 * wy) * wz;
                    const float w1 = (ex.w1 * wy) * wz;
#pragma unroll
                    for (int s = 0; s < 16; ++s) c[s] = fmaf(v0[s], w0, c[s]);
#pragma unroll
                    for (int s = 0; s < 16; ++s) c[s] = fmaf(v1[s], w1, c[s]);
                }
            };
            plane(rz0, ez.w0);
            pin16(c);
            __builtin_amdgcn_sched_barrier(0);
            plane(rz1, ez.w1);
            pin16(c);
            __builtin_amdgcn_sched_barrier(0);
            return c;
        };
        const f32x16 cA = gather(ezA);
        const f32x16 cB = gather(ezB);
        next_tile = claim();
        if (next_tile < t_end) fetch(next_tile, ox, oy, oz);

        // ---- MLP on both column groups (mlp_and_heads<false, true>, visual-only, two chains) ----
        const f32x16 b0 = load_frag16(L + VT_OFF_BIAS + h * 16);
        f32x16 netA = b0, netB = b0;
        {
            const float k0 = h ? py : px;
            const float wp0 = L[VT_OFF_WP + lane], wp1 = L[VT_OFF_WP + 64 + lane];
            netA = mfma(wp0, k0, netA);
            netB = mfma(wp0, k0, netB);
            netA = mfma(wp1, h ? 0.0f : pzA, netA);
            netB = mfma(wp1, h ? 0.0f : pzB, netB);
        }
        f32x16 ciA, ciB;
        bool has_img = false;
        if (a.cimg_ids) {
            // tactile feature by finger id: most double bricks touch no finger and skip the layer
            const unsigned idA = a.cimg_ids[gA], idB = a.cimg_ids[gB];
            if (__ballot(idA < a.cimg_nf || idB < a.cimg_nf) != 0ull) {
#pragma unroll
                for (int s = 0; s < 16; ++s) { ciA[s] = 0.0f; ciB[s] = 0.0f; }
                if (idA < a.cimg_nf) ciA = load_frag16(a.cimg_table + (size_t)idA * 32 + 16 * h);
                if (idB < a.cimg_nf) ciB = load_frag16(a.cimg_table + (size_t)idB * 32 + 16 * h);
                has_img = true;
            }
        } else if (a.c_img) {
            ciA = load_frag16(a.c_img + (size_t)gA * 32 + 16 * h);
            ciB = load_frag16(a.c_img + (size_t)gB * 32 + 16 * h);
            has_img = true;
        }
        if constexpr (P != 0) {
            constexpr int Q = P ? P : 1;
            if (has_img) dense32s2<Q>(netA, netB, L + VT_OFF_WPI, split16<false, Q>(ciA), split16<false, Q>(ciB), lane);
            const SplitP<Q> csA = split16<false, Q>(cA), csB = split16<false, Q>(cB);
            dense32s2<Q>(netA, netB, L + VT_OFF_WL, csA, csB, lane);
            // The block's c-conditioning and bias MFMAs (net += fc_c{i+1}(c) + b_1) depend on nothing the block
            // computes, only on net having been READ by the relu: they issue while the VALU splits relu(net) and
            // relu(hid), instead of back to back behind fc_1 with the VALU idle (net = net + cond + fc_1(..)).
            auto block = [&](int i, auto cond_tag) {
                constexpr bool COND = decltype(cond_tag)::value;
                const float *wl = L + VT_OFF_WL + (1 + 3 * i) * 1024;
                const f32x16 hb = load_frag16(L + VT_OFF_BIAS + (1 + 2 * i) * 32 + h * 16);
                const mx8<Q> bf = reinterpret_cast<const mx8<Q> *>(L + VT_OFF_BFRAG + i * 256)[lane];
                if constexpr (P == 1) {
                    f32x16 rA, rB;
#pragma unroll
                    for (int q = 0; q < 16; ++q) { rA[q] = relu1(netA[q]); rB[q] = relu1(netB[q]); }
                    asm volatile("" : "+v"(rA), "+v"(rB));
                    if constexpr (COND) dense32s2_half<Q>(netA, netB, wl + 2048, csA, csB, lane, 0);
                    netA = mfma_s(bf, ones, netA);
                    netB = mfma_s(bf, ones, netB);
                    f32x16 hidA = hb, hidB = hb;
                    dense32s2<Q>(hidA, hidB, wl, split16<false, Q>(rA), split16<false, Q>(rB), lane);
#pragma unroll
                    for (int q = 0; q < 16; ++q) { rA[q] = relu1(hidA[q]); rB[q] = relu1(hidB[q]); }
                    asm volatile("" : "+v"(rA), "+v"(rB));
                    if constexpr (COND) dense32s2_half<Q>(netA, netB, wl + 2048, csA, csB, lane, 1);
                    dense32s2<Q>(netA, netB, wl + 1024, split16<false, Q>(rA), split16<false, Q>(rB), lane);
                } else {
                    // split-f16: relu + split reads net directly (two instructions per value); the conditioning and
                    // bias MFMAs may only overwrite net once the split has read it
                    SplitP<Q> sA = split16<true, Q>(netA), sB = split16<true, Q>(netB);
                    range_track(rmax, __builtin_bit_cast(u32x4, sA.hi[0])[0]);
                    range_track(rmax, __builtin_bit_cast(u32x4, sB.hi[0])[0]);
                    asm volatile("" : "+v"(sA.hi[0]), "+v"(sA.hi[1]), "+v"(sA.lo[0]), "+v"(sA.lo[1]));
                    asm volatile("" : "+v"(sB.hi[0]), "+v"(sB.hi[1]), "+v"(sB.lo[0]), "+v"(sB.lo[1]));
                    if constexpr (COND) dense32s2_half<Q>(netA, netB, wl + 2048, csA, csB, lane, 0);
                    netA = mfma_s(bf, ones, netA);
                    netB = mfma_s(bf, ones, netB);
                    f32x16 hidA = hb, hidB = hb;
                    dense32s2<Q>(hidA, hidB, wl, sA, sB, lane);
                    if constexpr (COND) dense32s2_half<Q>(netA, netB, wl + 2048, csA, csB, lane, 1);
                    sA = split16<true, Q>(hidA);
                    sB = split16<true, Q>(hidB);
                    range_track(rmax, __builtin_bit_cast(u32x4, sA.hi[0])[0]);
                    range_track(rmax, __builtin_bit_cast(u32x4, sB.hi[0])[0]);
                    dense32s2<Q>(netA, netB, wl + 1024, sA, sB, lane);
                }
            };
#pragma unroll 1
            for (int i = 0; i < 4; ++i) block(i, std::true_type{});
            block(4, std::false_type{});
        } else {
            if (has_img) dense32x2<false>(netA, netB, L + VT_OFF_WPI, ciA, ciB, lane);
            dense32x2<false>(netA, netB, L + VT_OFF_WL, cA, cB, lane);
#pragma unroll 1
            for (int i = 0; i < 5; ++i) {
                const float *wl = L + VT_OFF_WL + (1 + 3 * i) * 1024;
                const f32x16 hb = load_frag16(L + VT_OFF_BIAS + (1 + 2 * i) * 32 + h * 16);
                f32x16 hidA = hb, hidB = hb;
                dense32x2<true>(hidA, hidB, wl, netA, netB, lane);
                dense32x2<true>(netA, netB, wl + 1024, hidA, hidB, lane);
                if (i < 4) dense32x2<false>(netA, netB, wl + 2048, cA, cB, lane);
                const f32x16 bb = load_frag16(L + VT_OFF_BIAS + (2 + 2 * i) * 32 + h * 16);
                netA = netA + bb;
                netB = netB + bb;
            }
        }
        {
            const f32x16 wo = load_frag16(L + VT_OFF_OUT + h * 16);
            float accA = 0.0f, accB = 0.0f;
#pragma unroll
            for (int s = 0; s < 16; ++s) { accA = fmaf(relu1(netA[s]), wo[s], accA); accB = fmaf(relu1(netB[s]), wo[s], accB); }
            accA += __shfl_xor(accA, 32);
            accB += __shfl_xor(accB, 32);
            const float ob = L[VT_OFF_OUT + 64];
            if (h == 0) { a.out[gA] = accA + ob; a.out[gB] = accB + ob; }
        }
    }
    if constexpr (P == 2) range_report(rmax, a.status);
    clock_end(a.clk, stamp, reinterpret_cast<unsigned long long *>(claim_ctr + 4));
}

}  // namespace
#ifdef VT_DECODE_F16_TU
#include "decode_st3.h"       // decode_fwd_staged3_kernel: the slot-pipelined form of the split-f16 lattice decode
#endif
namespace {

#ifndef VT_DECODE_F16_TU      // the remaining kernels exist once, in decode.o
// ---- trilinear gather only: feat[b,n,:] = grid sampled at the query point -----------------
__global__ void __launch_bounds__(256) sample_grid_kernel(DecodeArgs a, float *feat, int C) {
    const int lane = threadIdx.x & 63, pl = lane & 31, h = lane >> 5;
    const uint32_t ntiles = (a.total + 31u) >> 5;
    const int R = a.R;
    for (uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += gridDim.x * 4) {
        uint32_t g = tile * 32u + pl;
        const bool live = g < a.total;
        if (!live) g = a.total - 1u;
        const uint32_t b = g / a.N, n = g - b * a.N;
        float px, py, pz;
        point_of(a, g, n, px, py, pz);
        const Tri t = tri_setup(px, py, pz, a.divisor, R);
        for (int cb = 0; cb < C; cb += 32) {                     // c_dim in blocks of 32 channels (one block at the shipped shape)
            const float *gb = a.grid + (size_t)b * R * R * R * C + cb + 16 * h;
            f32x16 c;
#pragma unroll
            for (int s = 0; s < 16; ++s) c[s] = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int zz = (k & 4) ? t.z1 : t.z0, yy = (k & 2) ? t.y1 : t.y0, xx = (k & 1) ? t.x1 : t.x0;
                const float w = (((k & 1) ? t.wx1 : t.wx0) * ((k & 2) ? t.wy1 : t.wy0)) * ((k & 4) ? t.wz1 : t.wz0);
                const f32x16 v = load_frag16(gb + (((size_t)zz * R + yy) * R + xx) * C);
#pragma unroll
                for (int s = 0; s < 16; ++s) c[s] = fmaf(v[s], w, c[s]);
            }
            if (live) store_gather16(feat + (size_t)g * C + cb, c, h);
        }
    }
}

// ---- weight packing -------------------------------------------------------------
struct PackArgs {
    vt_decoder_params p;
    float *blob;
    int split;            // 1: dense layers as split-bf16 fragments (vt_decoder_pack_bf16x3), 2: split-f16 (vt_decoder_pack_f16x3),
                          // 3: f16 hi parts + fp8 correction fragments (vt_decoder_pack_f16f8)
};

// v as hi + mid + lo 16-bit floats (round to nearest each time); fmt 1 = bf16, 2 = IEEE half
__device__ __forceinline__ void split3_bits(float v, int fmt, unsigned &hi, unsigned &mid, unsigned &lo) {
    if (fmt == 2) {
        const _Float16 h = (_Float16)v;
        const float r1 = v - (float)h;
        _Float16 m = (_Float16)r1;
        const _Float16 l = (_Float16)(r1 - (float)m);
        hi = __builtin_bit_cast(unsigned short, h); mid = __builtin_bit_cast(unsigned short, m); lo = __builtin_bit_cast(unsigned short, l);
    } else {
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const __bf16 l = (__bf16)(r1 - (float)m);
        hi = __builtin_bit_cast(unsigned short, h); mid = __builtin_bit_cast(unsigned short, m); lo = __builtin_bit_cast(unsigned short, l);
    }
}

// f32 -> OCP fp8 e4m3 (round to nearest even, saturating at 448; no NaN is produced for finite input)
__device__ __forceinline__ unsigned f32_to_e4m3(float v) {
    const unsigned sign = (__builtin_bit_cast(unsigned, v) >> 31) << 7;
    float a = fabsf(v);
    if (!(a < 448.0f)) return sign | 0x7eu;
    int e;
    (void)frexpf(a, &e);                               // a = f * 2^e, f in [0.5, 1)
    e = max(e - 1, -6);                                // exponent of the leading bit, clamped into the subnormal range
    const float step = ldexpf(1.0f, e - 3);
    int n = (int)rintf(a / step);                      // 8..16 for a normal, 0..8 for a subnormal
    if (n >= 16) { n = 8; ++e; }
    const unsigned bits = (n >= 8) ? (unsigned)(((e + 7) << 3) | (n - 8)) : (unsigned)n;
    return sign | bits;
}

__global__ void decoder_pack_kernel(PackArgs a) {
    const vt_decoder_params &p = a.p;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < VT_BLOB_FLOATS; e += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (e >= VT_OFF_PFRAG) {
            // fc_p's coordinate columns for the one-MFMA form (vt_common.h): lane (i, kg), dword m = halves 2m, 2m+1
            const int q = e - VT_OFF_PFRAG, l = (q >> 2) & 63, m = q & 3, i = l & 31, kg = l >> 5;
            unsigned bits = 0;
            if (a.split >= 2) {
                unsigned hi[3], lo[3], unused;
                for (int k = 0; k < 3; ++k) split3_bits(p.fc_p_w[i * p.p_in + k], 2, hi[k], lo[k], unused);
                if (kg == 0) bits = (m < 3) ? (hi[m] | (hi[m] << 16)) : (lo[0] | (lo[1] << 16));
                else bits = (m == 0) ? lo[2] : 0u;
            }
            a.blob[e] = __builtin_bit_cast(float, bits);
            continue;
        }
        if (e >= VT_OFF_BFRAG) {
            // bias fragment of block blk: lane (i, kg), elements 0..2 of kg = 0 carry hi / mid / lo
            const int q = e - VT_OFF_BFRAG, blk = q >> 8, l = (q >> 2) & 63, m = q & 3, i = l & 31, kg = l >> 5;
            unsigned bits = 0;
            if (a.split && kg == 0 && m < 2) {
                const float bv = p.fc1_b[blk][i] + ((blk < 4) ? p.fc_c_b[blk + 1][i] : 0.0f);
                unsigned hb, mb, lb;
                split3_bits(bv, a.split == 1 ? 1 : 2, hb, mb, lb);
                bits = (m == 0) ? (hb | (mb << 16)) : lb;
            }
            a.blob[e] = __builtin_bit_cast(float, bits);
            continue;
        }
        if (a.split && e >= VT_OFF_WL) {
            // layer image [part hi|lo][k-step][lane][8 bf16]; one float slot = elements 2m, 2m+1.
            // k of element j: the accumulator register 8s+j of lane-half h (or, for the layers
            // fed by the gather / c_img, gather register 8s+j = channel 16h+8s+j).
            const int q = e - VT_OFF_WL;
            const int L = q >> 10, part = (q >> 9) & 1, s = (q >> 8) & 1, l = (q >> 2) & 63, m = q & 3;
            const int i = l & 31, h = l >> 5;
            const float *w;
            int ld = 32, koff = 0;
            bool gather_fed;
            if (L == 15) { w = p.fc_p_w; ld = p.p_in; koff = 3; gather_fed = true; }
            else if (L == 0) { w = p.fc_c_w[0]; gather_fed = true; }
            else {
                const int blk = (L - 1) / 3, kind = (L - 1) % 3;
                if (kind == 0) { w = p.fc0_w[blk]; gather_fed = false; }
                else if (kind == 1) { w = p.fc1_w[blk]; gather_fed = false; }
                else { w = p.fc_c_w[blk + 1]; gather_fed = true; }
            }
            unsigned bits = 0;
            if (a.split == 3 && part) {
                // "f16f8": the layer's second half is the fp8 (e4m3) A fragment of the correction MFMA, eight dwords per lane as
                // two 16-byte fragments: byte j < 16 pairs W_lo 2^(11+SW) with the k of accumulator register j, byte 16 + j
                // W_hi 2^SW with the same k (decode_st3.h); W_hi is the half the main product uses, W_lo = W - W_hi
                const int d8 = 4 * s + m;
                for (int b = 0; b < 4; ++b) {
                    const int j = 4 * d8 + b, r = j & 15;
                    const int k = gather_fed ? (16 * h + r) : chan_of(r, h);
                    const float wv = (L == 15 && p.p_in <= 3) ? 0.0f : w[i * ld + koff + k];
                    const float whi = (float)(_Float16)wv;
                    const float v = (j < 16) ? (wv - whi) * (float)(1 << (11 + VT_F8_SW)) : whi * (float)(1 << VT_F8_SW);
                    bits |= f32_to_e4m3(v) << (8 * b);
                }
                a.blob[e] = __builtin_bit_cast(float, bits);
                continue;
            }
            for (int z = 0; z < 2; ++z) {
                const int j = 2 * m + z;
                const int k = gather_fed ? (16 * h + 8 * s + j) : chan_of(8 * s + j, h);
                const float wv = (L == 15 && p.p_in <= 3) ? 0.0f : w[i * ld + koff + k];
                unsigned hb, mb, lb;
                split3_bits(wv, a.split == 1 ? 1 : 2, hb, mb, lb);
                bits |= (part ? mb : hb) << (16 * z);
            }
            a.blob[e] = __builtin_bit_cast(float, bits);
            continue;
        }
        if (e >= VT_OFF_WL && e < VT_OFF_WPI) {
            const int q = e - VT_OFF_WL;
            const int L = q >> 10, s = (q >> 6) & 15, l = q & 63, i = l & 31, h = l >> 5;
            const float *w;
            bool gather_fed;
            if (L == 0) { w = p.fc_c_w[0]; gather_fed = true; }
            else {
                const int blk = (L - 1) / 3, kind = (L - 1) % 3;
                if (kind == 0) { w = p.fc0_w[blk]; gather_fed = false; }
                else if (kind == 1) { w = p.fc1_w[blk]; gather_fed = false; }
                else { w = p.fc_c_w[blk + 1]; gather_fed = true; }
            }
            const int k = gather_fed ? (16 * h + s) : chan_of(s, h);
            v = w[i * 32 + k];
        } else if (e >= VT_OFF_WP && e < VT_OFF_WL) {
            const int q = e - VT_OFF_WP, s = q >> 6, l = q & 63, i = l & 31, h = l >> 5;
            const int k = 2 * s + h;
            v = (k < 3) ? p.fc_p_w[i * p.p_in + k] : 0.0f;
        } else if (e >= VT_OFF_WPI) {
            const int q = e - VT_OFF_WPI, s = q >> 6, l = q & 63, i = l & 31, h = l >> 5;
            v = (p.p_in > 3) ? p.fc_p_w[i * p.p_in + 3 + 16 * h + s] : 0.0f;
        } else if (e < VT_OFF_OUT) {
            const int q = e - VT_OFF_BIAS, j = q >> 5, h = (q >> 4) & 1, r = q & 15;
            const int o = chan_of(r, h);
            if (j == 0) v = p.fc_p_b[o] + p.fc_c_b[0][o];
            else if (j & 1) v = p.fc0_b[(j - 1) >> 1][o];
            else {
                const int blk = (j - 2) >> 1;
                v = p.fc1_b[blk][o] + ((blk < 4) ? p.fc_c_b[blk + 1][o] : 0.0f);
            }
        } else {
            const int q = e - VT_OFF_OUT;
            if (q < 32) v = p.fc_out_w[chan_of(q & 15, q >> 4)];
            else if (q < 64) v = p.fc_out2_w ? p.fc_out2_w[chan_of(q & 15, (q >> 4) & 1)] : 0.0f;
            else if (q == 64) v = p.fc_out_b[0];
            else if (q == 65) v = p.fc_out2_b ? p.fc_out2_b[0] : 0.0f;
        }
        a.blob[e] = v;
    }
}

// ---- NCDHW <-> NDHWC ---------------------------------------------------------------
// one block transposes a [C=32] x [64 voxels] tile through LDS
__global__ void __launch_bounds__(256) grid_to_cl_kernel(const float *src, float *dst, int C, int64_t V) {
    __shared__ float t[32][65];
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 32;
    const float *s = src + ((int64_t)b * C + c0) * V;
    float *d = dst + (int64_t)b * V * C;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, v = i & 63;
        if (c0 + c < C && v0 + v < V) t[c][v] = s[(int64_t)c * V + v0 + v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int v = i >> 5, c = i & 31;
        if (c0 + c < C && v0 + v < V) d[(v0 + v) * C + c0 + c] = t[c][v];
    }
}

__global__ void __launch_bounds__(256) grid_from_cl_kernel(const float *src, float *dst, int C, int64_t V) {
    __shared__ float t[64][33];
    const int64_t v0 = (int64_t)blockIdx.x * 64;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * 32;
    const float *s = src + (int64_t)b * V * C;
    float *d = dst + ((int64_t)b * C + c0) * V;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int v = i >> 5, c = i & 31;
        if (c0 + c < C && v0 + v < V) t[v][c] = s[(v0 + v) * C + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, v = i & 63;
        if (c0 + c < C && v0 + v < V) d[(int64_t)c * V + v0 + v] = t[v][c];
    }
}

#endif  // !VT_DECODE_F16_TU

}  // namespace

// =====================================================================================
// C ABI.  This file is compiled twice: as decode.o (everything except the split-f16 kernels) and, through
// decode_f16.hip with VT_DECODE_F16_TU and -fno-slp-vectorize, as decode_f16.o (vt_decode_fwd_f16x3 only).
// =====================================================================================
#ifndef VT_DECODE_F16_TU
// ---- range guard of the half-precision decodes: one status block PER DEVICE (kernels of a process that drives several
// GPUs must not atomicOr into another GPU's memory), created under a mutex at a device's first use.
//   bytes 0..3: VT_RANGE_* bits; from byte 8: the last lattice kernel's clock stamps (clock_end, decode_common.h):
//   [cycles of workgroup 0, its ticks, workgroups, -, then (start tick, end tick) per workgroup]
constexpr int VT_MAX_DEVICES = 64;
constexpr size_t VT_STATUS_BYTES = 8 + (4 + 2 * VT_CLK_MAX_WGS) * 8;
static std::mutex g_status_mutex;
static std::atomic<unsigned *> g_decode_status[VT_MAX_DEVICES];
unsigned *vt_decode_status_dev() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VT_MAX_DEVICES) return nullptr;
    unsigned *p = g_decode_status[dev].load(std::memory_order_acquire);
    if (p) return p;
    std::lock_guard<std::mutex> lock(g_status_mutex);
    p = g_decode_status[dev].load(std::memory_order_relaxed);
    if (!p) {
        // first use on this device: an allocation (the packers call this too, so that it never falls into a stream capture)
        if (hipMalloc(&p, VT_STATUS_BYTES) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, VT_STATUS_BYTES) != hipSuccess) { (void)hipFree(p); return nullptr; }
        g_decode_status[dev].store(p, std::memory_order_release);
    }
    return p;
}
extern "C" {

size_t vt_decoder_blob_bytes(int hidden, int c_dim, int n_blocks) {
    if (hidden != 32 || c_dim != 32 || n_blocks != 5) return 0;
    return (size_t)VT_BLOB_FLOATS * sizeof(float);
}

int vt_decoder_pack(const vt_decoder_params *p, float *blob, size_t blob_bytes, void *stream) {
    if (!p || !blob) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack: null argument");
    if (p->hidden != 32 || p->c_dim != 32 || p->n_blocks != 5)
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decoder_pack: only hidden=32, c_dim=32, n_blocks=5 (the shipped VTacO configs) are built");
    if (p->p_in != 3 && p->p_in != 3 + p->c_dim) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack: p_in must be 3 or 3+c_dim");
    if (blob_bytes < (size_t)VT_BLOB_FLOATS * sizeof(float)) return vt_fail(VT_ERR_WORKSPACE, "vt_decoder_pack: blob too small");
    if (!p->fc_p_w || !p->fc_p_b || !p->fc_out_w || !p->fc_out_b) return vt_fail(VT_ERR_INVALID, "vt_decoder_pack: null weight");
    for (int i = 0; i < 5; ++i)
        if (!p->fc_c_w[i] || !p->fc_c_b[i] || !p->fc0_w[i] || !p->fc0_b[i] || !p->fc1_w[i] || !p->fc1_b[i])
            return vt_fail(VT_ERR_INVALID, "vt_decoder_pack: null block weight");
    (void)vt_decode_status_dev();          // the device's status block exists before any launch can be captured into a graph
    PackArgs a;
    a.p = *p;
    a.blob = blob;
    a.split = 0;
    hipLaunchKernelGGL(decoder_pack_kernel, dim3(17), dim3(1024), 0, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decoder_pack");
}

int vt_decoder_pack_bf16x3(const vt_decoder_params *p, float *blob, size_t blob_bytes, void *stream) {
    // same argument checks and blob size as the f32 pack; only the dense-layer images differ
    const int rc = vt_decoder_pack(p, blob, blob_bytes, stream);
    if (rc) return rc;
    PackArgs a;
    a.p = *p;
    a.blob = blob;
    a.split = 1;
    hipLaunchKernelGGL(decoder_pack_kernel, dim3(17), dim3(1024), 0, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decoder_pack_bf16x3");
}

int vt_decoder_pack_f16x3(const vt_decoder_params *p, float *blob, size_t blob_bytes, void *stream) {
    const int rc = vt_decoder_pack(p, blob, blob_bytes, stream);
    if (rc) return rc;
    (void)vt_decode_status_dev();
    PackArgs a;
    a.p = *p;
    a.blob = blob;
    a.split = 2;
    hipLaunchKernelGGL(decoder_pack_kernel, dim3(17), dim3(1024), 0, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decoder_pack_f16x3");
}

int vt_decoder_pack_f16f8(const vt_decoder_params *p, float *blob, size_t blob_bytes, void *stream) {
    const int rc = vt_decoder_pack(p, blob, blob_bytes, stream);
    if (rc) return rc;
    (void)vt_decode_status_dev();
    PackArgs a;
    a.p = *p;
    a.blob = blob;
    a.split = 3;
    hipLaunchKernelGGL(decoder_pack_kernel, dim3(17), dim3(1024), 0, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decoder_pack_f16f8");
}

int vt_decode_range_status(unsigned *host_status, int reset, void *stream) {
    if (!host_status && !reset) return vt_fail(VT_ERR_INVALID, "vt_decode_range_status: null argument");
    unsigned *d = vt_decode_status_dev();
    if (!d) return vt_fail(VT_ERR_INVALID, "vt_decode_range_status: no device memory for the status word");
    if (!host_status) return vt_fill32(d, 0u, sizeof(unsigned), (hipStream_t)stream);      // clear without a read-back: asynchronous
    hipError_t e = hipMemcpyAsync(host_status, d, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return vt_check(e, "vt_decode_range_status");
    if (reset && *host_status) return vt_fill32(d, 0u, sizeof(unsigned), (hipStream_t)stream);
    return 0;
}

int vt_decode_last_clock(unsigned long long *shader_cycles, unsigned long long *ref_ticks, int *ref_khz,
                         unsigned long long *wg_ticks, int max_wgs, int *n_wgs, void *stream) {
    if (!shader_cycles || !ref_ticks || !ref_khz) return vt_fail(VT_ERR_INVALID, "vt_decode_last_clock: null argument");
    if (max_wgs < 0 || (max_wgs > 0 && (!wg_ticks || !n_wgs))) return vt_fail(VT_ERR_INVALID, "vt_decode_last_clock: bad workgroup buffer");
    unsigned *d = vt_decode_status_dev();
    if (!d) return vt_fail(VT_ERR_INVALID, "vt_decode_last_clock: no device memory for the status block");
    static thread_local unsigned long long h[4 + 2 * VT_CLK_MAX_WGS];
    hipError_t e = hipMemcpyAsync(h, status_clk(d), sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    int dev = 0, khz = 0;
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);   // the rate of s_memrealtime
    if (e != hipSuccess) return vt_check(e, "vt_decode_last_clock");
    *shader_cycles = h[0]; *ref_ticks = h[1]; *ref_khz = khz;
    if (max_wgs > 0) {
        int n = (int)(h[2] < (unsigned long long)VT_CLK_MAX_WGS ? h[2] : VT_CLK_MAX_WGS);
        if (n > max_wgs) n = max_wgs;
        for (int i = 0; i < 2 * n; ++i) wg_ticks[i] = h[4 + i];
        *n_wgs = n;
    }
    return 0;
}

int vt_grid_to_channels_last(const float *src, float *dst, int B, int C, int D, int H, int W, void *stream) {
    if (!src || !dst || B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return vt_fail(VT_ERR_INVALID, "vt_grid_to_channels_last: bad argument");
    const int64_t V = (int64_t)D * H * W;
    dim3 grid((unsigned)((V + 63) / 64), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(grid_to_cl_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, V);
    return vt_check(hipGetLastError(), "vt_grid_to_channels_last");
}

int vt_grid_from_channels_last(const float *src, float *dst, int B, int C, int D, int H, int W, void *stream) {
    if (!src || !dst || B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return vt_fail(VT_ERR_INVALID, "vt_grid_from_channels_last: bad argument");
    const int64_t V = (int64_t)D * H * W;
    dim3 grid((unsigned)((V + 63) / 64), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(grid_from_cl_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, V);
    return vt_check(hipGetLastError(), "vt_grid_from_channels_last");
}

}  // extern "C"
#endif  // !VT_DECODE_F16_TU

#ifdef VT_DECODE_F16_TU
// the slot-pipelined lattice kernels (decode_st3.h) cover: whole 2 x 4 x 8 double bricks (nx % 8 == 0, a slab of x-plane pairs),
// less than 0.55 voxels per lattice step (the 3 x 4 x 6 footprint), blob + tables + eight images within the CU's 160 KiB
static bool st3_covers(int nx, int R, double s_vox) {
    const size_t lds = ((size_t)VT_BLOB_FLOATS + 5u * (size_t)nx) * sizeof(float) + (size_t)(ST3_THREADS / 64) * ST2_WAVE_BYTES + VT_TAIL_LDS_BYTES;
    return (nx & 7) == 0 && R >= 6 && s_vox > 0.0 && s_vox < 0.55 && lds <= 160u * 1024u;
}
static int st3_launch(const DecodeArgs &a_in, int variant, void *stream) {
    DecodeArgs a = a_in;
    a.clk = status_clk(a.status);
    const size_t lds = ((size_t)VT_BLOB_FLOATS + 5u * (size_t)a.nx) * sizeof(float) + (size_t)(ST3_THREADS / 64) * ST2_WAVE_BYTES + VT_TAIL_LDS_BYTES;   // + the tile counter and the end stamps
    const int64_t nt = (int64_t)a.total / 64;
    int64_t blocks = (nt + ST3_THREADS / 64 - 1) / (ST3_THREADS / 64);
    if (blocks > vt_num_cus()) blocks = vt_num_cus();
    if (blocks > 8) blocks &= ~7ll;
    bool attr = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_staged3_kernel<0>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_staged3_kernel<1>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_staged3_kernel<2>), 160 * 1024);
        if (e == hipSuccess) e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_staged3_kernel<3>), 160 * 1024);
        if (e != hipSuccess) return vt_check(e, "vt_decode_fwd: hipFuncSetAttribute (staged3)");
        attr = true;
    }
    if (variant == 0) hipLaunchKernelGGL(decode_fwd_staged3_kernel<0>, dim3((unsigned)blocks), dim3(ST3_THREADS), lds, (hipStream_t)stream, a);
    else if (variant == 1) hipLaunchKernelGGL(decode_fwd_staged3_kernel<1>, dim3((unsigned)blocks), dim3(ST3_THREADS), lds, (hipStream_t)stream, a);
    else if (variant == 2) hipLaunchKernelGGL(decode_fwd_staged3_kernel<2>, dim3((unsigned)blocks), dim3(ST3_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(decode_fwd_staged3_kernel<3>, dim3((unsigned)blocks), dim3(ST3_THREADS), lds, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decode_fwd");
}
static double lattice_step_voxels(int R, int nx, float box, double padding) {
    return (double)(R - 1) * (double)box / ((double)(nx - 1) * (double)(float)(1.0 + padding + 10e-4));
}
// vt_sample_grid over a lattice slab the staged double-brick kernel covers (whole x-plane pairs, nx % 8 == 0, < 0.55 voxels per
// step): the gather of decode_fwd_staged3_kernel alone -- footprints by LDS-DMA, corners from LDS -- writing the features.  Same
// corner and FMA order as sample_grid_kernel: the same bits.  *covered = 0: not such a slab, nothing launched.
int vt_st3_sample_lattice(const float *grid_cl, int B, int R, int C, int64_t N, int nx, float box, int64_t first, double padding,
                          float *feat, void *stream, int *covered) {
    *covered = 0;
    if (C != 32 || nx < 8 || N <= 0 || (int64_t)B * N >= (int64_t)1 << 31) return 0;
    const int64_t pair = 2ll * nx * nx;
    if (first % pair != 0 || N % pair != 0) return 0;
    if (!st3_covers(nx, R, lattice_step_voxels(R, nx, box, padding))) return 0;
    static const bool off = getenv("VTACO_SAMPLE_STAGED") != nullptr && atoi(getenv("VTACO_SAMPLE_STAGED")) == 0;
    if (off) return 0;
    DecodeArgs a;
    a.status = nullptr; a.clk = nullptr; a.claim = 1;
    a.c_direct = nullptr; a.grid = grid_cl; a.pts = nullptr; a.brick = 1; a.cimg_ids = nullptr; a.cimg_table = nullptr; a.cimg_nf = 0; a.c_img = nullptr;
    a.blob = nullptr; a.out = feat; a.out2 = nullptr; a.save = nullptr;
    a.N = (uint32_t)N; a.total = (uint32_t)((int64_t)B * N); a.lattice_first = (uint32_t)first;
    a.R = R; a.nx = nx; a.box = box; a.divisor = (float)(1.0 + padding + 10e-4);
    *covered = 1;
    return st3_launch(a, 3, stream);
}
#endif

// P: 0 exact f32, 1 split-bf16, 2 split-f16 (the dense layers; decode_common.h)
template <int P>
static int decode_launch(const float *grid_cl, const float *c_direct, int B, int R, int C, const float *pts, int64_t N,
                         int lattice_nx, float lattice_box, int64_t lattice_first,
                         const float *c_img, const unsigned char *cimg_ids, const float *cimg_table, int cimg_nf,
                         const float *blob, double padding,
                         float *out, float *out2, float *save, void *stream) {
    if ((!grid_cl && !c_direct) || !blob || !out) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd: null argument");
    if (P != 0 && save) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_bf16x3 / _f16x3: the training forward (save) is exact-f32 only");
    if (B <= 0 || R < 2 || N < 0) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd: bad size");
    if (C != 32) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd: c_dim must be 32");
    if (!pts) {
        if (lattice_nx < 2) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd: lattice mode needs nx >= 2");
        const int64_t all = (int64_t)lattice_nx * lattice_nx * lattice_nx;
        if (lattice_first < 0 || lattice_first + N > all) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd: lattice range out of bounds");
    }
    if (N == 0) return 0;
    if ((int64_t)B * N >= (int64_t)1 << 31) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd: B*N must be < 2^31");
    DecodeArgs a;
    unsigned *const status_block = vt_decode_status_dev();
    a.status = (P == 2) ? status_block : nullptr;
    a.clk = nullptr;                                       // set for the lattice kernels below
    static const int claim_tiles = !(getenv("VTACO_DECODE_CLAIM") && atoi(getenv("VTACO_DECODE_CLAIM")) == 0);
    a.claim = claim_tiles && P != 0;                       // (same box, claims / fixed shares: f16x3 163.2 / 167.2 us, f16f8 149.5 / 155.0,
                                                           // bf16x3 186.0 / 190.1 -- and exact f32 502.9 / 496.3: its waves are matrix-bound, not issue-bound)
    a.c_direct = c_direct; a.grid = grid_cl; a.pts = pts; a.brick = 0; a.cimg_ids = cimg_ids; a.cimg_table = cimg_table; a.cimg_nf = cimg_ids ? (uint32_t)(cimg_nf < 255 ? cimg_nf : 255) : 0u; a.c_img = c_img; a.blob = blob; a.out = out; a.out2 = out2; a.save = save;
    a.N = (uint32_t)N; a.total = (uint32_t)((int64_t)B * N); a.lattice_first = (uint32_t)lattice_first;
    a.R = R; a.nx = lattice_nx; a.box = lattice_box;
    a.divisor = (float)(1.0 + padding + 10e-4);   // src/common.py:302, rounded to f32 as torch does
    if (!pts && !save && (lattice_nx & 3) == 0) {
        const int64_t pair = 2ll * lattice_nx * lattice_nx;            // two x-planes
        if (lattice_first % pair == 0 && N % pair == 0) a.brick = 1;
    }
    // LDS-staged gather (decode_fwd_staged_kernel) when the brick footprint is bounded by 3 x 4 x 4 voxels:
    // voxels per lattice step s = (R-1) * box / ((nx-1) * divisor) < 2/3 (0.496 at 128^3 / R=64)
    static const bool force_direct = getenv("VTACO_DECODE_DIRECT") != nullptr;       // A/B knob for tests and benches
    if (a.brick && !force_direct && R >= 4 && lattice_nx <= 512 && !c_direct) {
        const double s_vox = (double)(R - 1) * (double)lattice_box / ((double)(lattice_nx - 1) * (double)a.divisor);
        const size_t lds_st = ((size_t)VT_BLOB_FLOATS + 4u * (size_t)lattice_nx + (size_t)(ST_THREADS / 64) * ST_WAVE_FLOATS) * sizeof(float);
        const size_t lds_st2 = ((size_t)VT_BLOB_FLOATS + 4u * (size_t)lattice_nx) * sizeof(float) + (size_t)(ST2_THREADS / 64) * ST2_WAVE_BYTES + VT_TAIL_LDS_BYTES;
        static const bool no_pair = getenv("VTACO_DECODE_NO_PAIR") != nullptr;    // A/B knob
#ifdef VT_DECODE_F16_TU
        if constexpr (P == 2) {
            // slot-pipelined double-brick kernel (decode_st3.h); VTACO_DECODE_ST3=0 falls back to the compiler-scheduled one,
            // VTACO_DECODE_ST3_SPLIT=0 selects the 4-instruction relu/split (mix-to-half) instead of the 5-instruction one
            static const bool no_st3 = getenv("VTACO_DECODE_ST3") != nullptr && atoi(getenv("VTACO_DECODE_ST3")) == 0;
            static const bool mix_half = getenv("VTACO_DECODE_ST3_SPLIT") != nullptr && atoi(getenv("VTACO_DECODE_ST3_SPLIT")) == 0;
            if (!no_st3 && !no_pair && !out2 && st3_covers(lattice_nx, R, s_vox)) return st3_launch(a, mix_half ? 0 : 1, stream);
        }
#endif
        if (!no_pair && !out2 && (lattice_nx & 7) == 0 && R >= 6 && s_vox > 0.0 && s_vox < 0.55 &&
            lds_st2 <= 160u * 1024u) {
            const int64_t nt = (int64_t)a.total / 64;
            int64_t blocks = (nt + ST2_THREADS / 64 - 1) / (ST2_THREADS / 64);
            if (blocks > vt_num_cus()) blocks = vt_num_cus();
            if (blocks > 8) blocks &= ~7ll;
            static bool st2_attr = false;
            if (!st2_attr) {
                const hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_staged2_kernel<P>), 160 * 1024);
                if (e != hipSuccess) return vt_check(e, "vt_decode_fwd: hipFuncSetAttribute (staged2)");
                st2_attr = true;
            }
            a.clk = status_clk(status_block);
            hipLaunchKernelGGL(decode_fwd_staged2_kernel<P>, dim3((unsigned)blocks), dim3(ST2_THREADS), lds_st2, (hipStream_t)stream, a);
            return vt_check(hipGetLastError(), "vt_decode_fwd");
        }
        if (s_vox > 0.0 && s_vox < 0.66 && lds_st <= 160u * 1024u) {
            const int64_t nt = (int64_t)a.total / 32;
            int64_t blocks = (nt + ST_THREADS / 64 - 1) / (ST_THREADS / 64);
            if (blocks > vt_num_cus()) blocks = vt_num_cus();
            if (blocks > 8) blocks &= ~7ll;
            bool st_attr = false;        // (vt_max_dyn_lds keeps the per-device record)
            if (!st_attr) {
                const hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_staged_kernel<P>), 160 * 1024);
                if (e != hipSuccess) return vt_check(e, "vt_decode_fwd: hipFuncSetAttribute (staged)");
                st_attr = true;
            }
            hipLaunchKernelGGL((decode_fwd_staged_kernel<P>), dim3((unsigned)blocks), dim3(ST_THREADS), lds_st, (hipStream_t)stream, a);
            return vt_check(hipGetLastError(), "vt_decode_fwd");
        }
    }
    constexpr int THREADS = 512;
    const int64_t ntiles = ((int64_t)a.total + 31) / 32;
    int64_t blocks = (ntiles + THREADS / 64 - 1) / (THREADS / 64);
    const int64_t cap = (int64_t)(1024 / THREADS) * vt_num_cus();
    if (blocks > cap) blocks = cap;
    if (blocks > 8) blocks &= ~7ll;                                     // whole workgroups per XCD
    const size_t lds_bytes = (size_t)VT_BLOB_FLOATS * sizeof(float);
    bool attr_set = false;        // (vt_max_dyn_lds keeps the per-device record)
    if (!attr_set) {
        hipError_t e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_kernel<THREADS, false, P>), (int)lds_bytes);
        if constexpr (P == 0) {
            if (e == hipSuccess)
                e = vt_max_dyn_lds(reinterpret_cast<const void *>(&decode_fwd_kernel<THREADS, true, 0>), (int)lds_bytes);
        }
        if (e != hipSuccess) return vt_check(e, "vt_decode_fwd: hipFuncSetAttribute");
        attr_set = true;
    }
    if constexpr (P == 0) {
        if (save) {
            hipLaunchKernelGGL((decode_fwd_kernel<THREADS, true, 0>), dim3((unsigned)blocks), dim3(THREADS), lds_bytes, (hipStream_t)stream, a);
            return vt_check(hipGetLastError(), "vt_decode_fwd");
        }
    }
    hipLaunchKernelGGL((decode_fwd_kernel<THREADS, false, P>), dim3((unsigned)blocks), dim3(THREADS), lds_bytes, (hipStream_t)stream, a);
    return vt_check(hipGetLastError(), "vt_decode_fwd");
}

extern "C" {

#ifdef VT_DECODE_F16_TU
int vt_decode_f16f8_covers(int R, int C, int lattice_nx, float lattice_box, int64_t lattice_first, int64_t N, double padding) {
    if (C != 32 || lattice_nx < 8 || N <= 0) return 0;
    const int64_t pair = 2ll * lattice_nx * lattice_nx;
    if (lattice_first % pair != 0 || N % pair != 0) return 0;
    return st3_covers(lattice_nx, R, lattice_step_voxels(R, lattice_nx, lattice_box, padding)) ? 1 : 0;
}

int vt_decode_fwd_f16f8(const float *grid_cl, int B, int R, int C, int64_t N, int lattice_nx, float lattice_box, int64_t lattice_first,
                        const float *c_img, const unsigned char *finger_ids, const float *finger_feats, int F,
                        const float *blob_f16f8, double padding, float *out, void *stream) {
    if (!grid_cl || !blob_f16f8 || !out) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_f16f8: null argument");
    if (c_img && finger_ids) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_f16f8: give c_img or finger ids, not both");
    if (finger_ids && (!finger_feats || F <= 0)) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_f16f8: finger ids without a feature table");
    if (B <= 0 || N < 0) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_f16f8: bad size");
    if (N == 0) return 0;
    if ((int64_t)B * N >= (int64_t)1 << 31) return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_f16f8: B*N must be < 2^31");
    if (!vt_decode_f16f8_covers(R, C, lattice_nx, lattice_box, lattice_first, N, padding))
        return vt_fail(VT_ERR_UNSUPPORTED, "vt_decode_fwd_f16f8: covers lattice slabs of whole x-plane pairs with nx % 8 == 0 and < 0.55 voxels per "
                                           "step (vt_decode_f16f8_covers); use vt_decode_fwd_f16x3 otherwise");
    DecodeArgs a;
    a.status = vt_decode_status_dev();
    a.clk = nullptr;
    a.claim = !(getenv("VTACO_DECODE_CLAIM") && atoi(getenv("VTACO_DECODE_CLAIM")) == 0);
    a.c_direct = nullptr; a.grid = grid_cl; a.pts = nullptr; a.brick = 1; a.cimg_ids = finger_ids; a.cimg_table = finger_ids ? finger_feats : nullptr; a.cimg_nf = finger_ids ? (uint32_t)(F < 255 ? F : 255) : 0u;
    a.c_img = c_img; a.blob = blob_f16f8; a.out = out; a.out2 = nullptr; a.save = nullptr;
    a.N = (uint32_t)N; a.total = (uint32_t)((int64_t)B * N); a.lattice_first = (uint32_t)lattice_first;
    a.R = R; a.nx = lattice_nx; a.box = lattice_box; a.divisor = (float)(1.0 + padding + 10e-4);
    return st3_launch(a, 2, stream);
}

int vt_decode_fwd_f16x3(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                        int lattice_nx, float lattice_box, int64_t lattice_first,
                        const float *c_img, const unsigned char *finger_ids, const float *finger_feats, int F,
                        const float *blob_f16x3, double padding, float *out, float *out2, void *stream) {
    if (!grid_cl) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_f16x3: null grid");
    if (c_img && finger_ids) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_f16x3: give c_img or finger ids, not both");
    if (finger_ids && (!finger_feats || F <= 0)) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_f16x3: finger ids without a feature table");
    return decode_launch<2>(grid_cl, nullptr, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, c_img,
                            finger_ids, finger_ids ? finger_feats : nullptr, F, blob_f16x3, padding, out, out2, nullptr, stream);
}

// vt_decode_mlp_fwd with split-f16 layers (inference: the MLP behind the TransformerFusion of AttentionDecoder.forward_img)
int vt_decode_mlp_fwd_f16x3(const float *c, int B, int C, const float *pts, int64_t N,
                            int lattice_nx, float lattice_box, int64_t lattice_first,
                            const float *blob_f16x3, float *out, void *stream) {
    if (!c) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_fwd_f16x3: null features");
    return decode_launch<2>(nullptr, c, B, 2, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, nullptr, nullptr, 0, blob_f16x3, 0.1,
                            out, nullptr, nullptr, stream);
}
#else
int vt_decode_fwd(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                  int lattice_nx, float lattice_box, int64_t lattice_first,
                  const float *c_img, const float *blob, double padding,
                  float *out, float *out2, float *save, void *stream) {
    if (!grid_cl) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd: null grid");
    return decode_launch<0>(grid_cl, nullptr, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, c_img, nullptr, nullptr, 0,
                            blob, padding, out, out2, save, stream);
}

int vt_decode_fwd_ids(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                      int lattice_nx, float lattice_box, int64_t lattice_first,
                      const unsigned char *finger_ids, const float *finger_feats, int F,
                      const float *blob, double padding, float *out, void *stream) {
    if (!grid_cl || !finger_ids || !finger_feats || F <= 0) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_ids: null argument");
    return decode_launch<0>(grid_cl, nullptr, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, finger_ids, finger_feats, F,
                            blob, padding, out, nullptr, nullptr, stream);
}

int vt_decode_fwd_bf16x3(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                         int lattice_nx, float lattice_box, int64_t lattice_first,
                         const float *c_img, const unsigned char *finger_ids, const float *finger_feats, int F,
                         const float *blob_bf16x3, double padding, float *out, float *out2, void *stream) {
    if (!grid_cl) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_bf16x3: null grid");
    if (c_img && finger_ids) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_bf16x3: give c_img or finger ids, not both");
    if (finger_ids && (!finger_feats || F <= 0)) return vt_fail(VT_ERR_INVALID, "vt_decode_fwd_bf16x3: finger ids without a feature table");
    return decode_launch<1>(grid_cl, nullptr, B, R, C, pts, N, lattice_nx, lattice_box, lattice_first, c_img,
                            finger_ids, finger_ids ? finger_feats : nullptr, F, blob_bf16x3, padding, out, out2, nullptr, stream);
}

int vt_decode_mlp_fwd(const float *c, int B, int C, const float *pts, int64_t N,
                      int lattice_nx, float lattice_box, int64_t lattice_first,
                      const float *blob, float *out, void *stream) {
    if (!c) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_fwd: null features");
    return decode_launch<0>(nullptr, c, B, 2, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, nullptr, nullptr, 0, blob, 0.1,
                            out, nullptr, nullptr, stream);
}

int vt_decode_mlp_fwd_train(const float *c, int B, int C, const float *pts, int64_t N,
                            int lattice_nx, float lattice_box, int64_t lattice_first,
                            const float *blob, float *out, float *save, void *stream) {
    if (!c || !save) return vt_fail(VT_ERR_INVALID, "vt_decode_mlp_fwd_train: null argument");
    return decode_launch<0>(nullptr, c, B, 2, C, pts, N, lattice_nx, lattice_box, lattice_first, nullptr, nullptr, nullptr, 0, blob, 0.1,
                            out, nullptr, save, stream);
}

int vt_sample_grid(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                   int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                   float *feat, void *stream) {
    if (!grid_cl || !feat) return vt_fail(VT_ERR_INVALID, "vt_sample_grid: null argument");
    if (B <= 0 || R < 2 || N < 0) return vt_fail(VT_ERR_INVALID, "vt_sample_grid: bad size");
    if (C <= 0 || (C & 31)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_sample_grid: c_dim must be a multiple of 32");
    if (!pts && lattice_nx < 2) return vt_fail(VT_ERR_INVALID, "vt_sample_grid: lattice mode needs nx >= 2");
    if (N == 0) return 0;
    if ((int64_t)B * N >= (int64_t)1 << 31) return vt_fail(VT_ERR_UNSUPPORTED, "vt_sample_grid: B*N must be < 2^31");
    if (!pts) {                                           // lattice slabs of the staged kernel's shape: its gather alone (decode_f16.o)
        int covered = 0;
        const int rc = vt_st3_sample_lattice(grid_cl, B, R, C, N, lattice_nx, lattice_box, lattice_first, padding, feat, stream, &covered);
        if (rc || covered) return rc;
    }
    DecodeArgs a;
    a.status = nullptr; a.clk = nullptr; a.claim = 0;
    a.c_direct = nullptr; a.brick = 0; a.cimg_ids = nullptr; a.cimg_table = nullptr; a.cimg_nf = 0; a.grid = grid_cl; a.pts = pts; a.c_img = nullptr; a.blob = nullptr; a.out = nullptr; a.out2 = nullptr; a.save = nullptr;
    a.N = (uint32_t)N; a.total = (uint32_t)((int64_t)B * N); a.lattice_first = (uint32_t)lattice_first;
    a.R = R; a.nx = lattice_nx; a.box = lattice_box; a.divisor = (float)(1.0 + padding + 10e-4);
    int64_t blocks = (((int64_t)a.total + 31) / 32 + 3) / 4;
    const int64_t cap = 8 * vt_num_cus();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sample_grid_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, feat, C);
    return vt_check(hipGetLastError(), "vt_sample_grid");
}
#endif  // VT_DECODE_F16_TU

}  // extern "C"
