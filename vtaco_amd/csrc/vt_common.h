// Shared internals of libvtaco_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/vtaco_hip.h"

// ---- packed decoder blob layout (floats); see decode.hip -------------------------
// Small fragments first, so that every LDS read of the visual-only path has a byte offset
// below 64 KiB and fits the ds_read immediate (no address VALU next to the MFMAs).
// 11 bias fragments [2 halves][16 regs]: 0 = fc_p.b + fc_c0.b, 1+2i = fc_0_i.b, 2+2i = fc_1_i.b + fc_c{i+1}.b
constexpr int VT_OFF_BIAS = 0;
// fc_out fragment [2][16], fc_out_contact fragment [2][16], fc_out.b, fc_out_contact.b, pad
constexpr int VT_OFF_OUT = VT_OFF_BIAS + 11 * 32;
// fc_p (or fc_p_img columns 0..2), K padded to 4: [2 k-steps][64 lanes]
constexpr int VT_OFF_WP = VT_OFF_OUT + 64 + 32;
// 15 dense layers x [16 k-steps][64 lanes]  : fc_c0, then per block fc_0, fc_1, fc_c{i+1}
constexpr int VT_OFF_WL = VT_OFF_WP + 128;
// fc_p_img columns 3..34 (tactile concat): [16][64]
constexpr int VT_OFF_WPI = VT_OFF_WL + 15 * 1024;
// split-bf16 blob only: the five block-end biases as A fragments of one 32x32x16 bf16 MFMA each
// ([64 lanes][8 bf16]; k-slots 0..2 = bf16 hi/mid/lo of the bias, against a B operand of ones)
constexpr int VT_OFF_BFRAG = VT_OFF_WPI + 1024;
// split-f16 blob only: fc_p's coordinate columns as the A fragment of ONE 32x32x16 f16 MFMA ([64 lanes][8 halves]):
// lane half 0 = [Wx_hi, Wx_hi, Wy_hi, Wy_hi, Wz_hi, Wz_hi, Wx_lo, Wy_lo], half 1 = [Wz_lo, 0 x 7], against the B operand
// [x_hi, x_mid, y_hi, y_mid, z_hi, z_mid, x_hi, y_hi | z_hi, 0 x 7] of the slot-pipelined lattice kernel (decode_st3.h)
constexpr int VT_OFF_PFRAG = VT_OFF_BFRAG + 5 * 256;
constexpr int VT_BLOB_FLOATS = VT_OFF_PFRAG + 256;
// "f16f8" blob (vt_decoder_pack_f16f8): the lo half of every dense layer image is the fp8 (e4m3) A fragment of the correction MFMA;
// weights enter it as W 2^VT_F8_SW, activations as x 2^-VT_F8_SX (decode_st3.h)
constexpr int VT_F8_SW = 4;
constexpr int VT_F8_SX = 2;
static_assert(VT_BLOB_FLOATS % 4 == 0, "blob is copied as float4");
static_assert((VT_OFF_WPI) * 4 <= 65536, "visual-only fragments must sit below the 64 KiB ds_read offset limit");

int vt_fail(int code, const char *msg);
int vt_check(hipError_t e, const char *where);
int vt_num_cus();
// hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), once per (device, kernel): the attribute belongs to
// the device the call is made on, so a process that drives several GPUs must set it on each (a process-wide `static bool` would
// leave the second device's > 64 KiB launches failing)
hipError_t vt_max_dyn_lds(const void *fn, int bytes);
// the device word of the half-precision decodes' range guard (decode_common.h; read by vt_decode_range_status), allocated at first use
// (one block per device: bytes 0..3 the VT_RANGE_* bits, bytes 8..23 the last lattice kernel's clock stamps)
unsigned *vt_decode_status_dev();
static inline unsigned long long *status_clk(unsigned *status) {
    return status ? reinterpret_cast<unsigned long long *>(status + 2) : nullptr;
}
// the staged lattice gather alone (decode_f16.o; vt_sample_grid tries it first): *covered = 0 when the slab is not of its shape
__attribute__((visibility("hidden"))) int vt_st3_sample_lattice(const float *grid_cl, int B, int R, int C, int64_t N, int nx, float box, int64_t first,
                                                                   double padding, float *feat, void *stream, int *covered);
// capture-safe replacement of hipMemsetAsync (memset nodes misbehaved under hipGraph replay on ROCm 7.0/7.2):
// fills `bytes` (multiple of 4) at `dst` with the 32-bit pattern
int vt_fill32(void *dst, unsigned pattern, size_t bytes, hipStream_t stream);

// ---- activations the training forward saves for the backward: [slot][point][32] ---------
// 0: c (trilinear features)   1..5: relu(x_i) (block inputs)   6..10: relu(h_i)   11: relu(net_5)
constexpr int VT_SAVE_SLOTS = 12;
// ---- output-side gradients vt_decode_bwd leaves for the weight-gradient pass -------------
// 0: d x_0 (fc_p / fc_c0 output)   1..5: d h_i (fc_0 output)   6..10: d block_i output (fc_1, fc_c{i+1})
constexpr int VT_GWS_SLOTS = 11;
