/*
 * vtaco_hip.h -- C ABI of libvtaco_hip.so, the MI355X (gfx950) implementation
 * of VTacO's occupancy hot path.
 *
 * The reference (jeffsonyu/VTacO) is pure Python: its "plugin API" for this path
 * is the nn.Module registry (decoder_dict / encoder_dict).  There is no FFI in
 * the reference, so each entry point below names the reference *call site* whose
 * third-party native op it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - no allocation, no synchronisation, no host<->device copy inside any call
 *     (graph-capture safe) unless stated;
 *   - return value: 0 = ok, >0 = a hipError_t from the runtime, <0 = VT_ERR_*;
 *     vt_last_error() returns a static, thread-local description;
 *   - all float data is IEEE float32, all layouts are dense row-major in the
 *     stated dimension order.
 */
#ifndef VTACO_HIP_H
#define VTACO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VT_ABI_VERSION 1

#define VT_ERR_INVALID      (-1)  /* bad argument (NULL, negative size, ...)        */
#define VT_ERR_UNSUPPORTED  (-2)  /* shape outside what the gfx950 kernels cover    */
#define VT_ERR_WORKSPACE    (-3)  /* caller-provided workspace too small            */

#define VT_MAX_BLOCKS 8

int         vt_abi_version(void);
const char *vt_last_error(void);

/* ------------------------------------------------------------------------- */
/* Decoder weights.                                                            */
/* Replaces: nothing in the reference; it is the one-off repack of the          */
/* nn.Linear parameters of LocalDecoder (src/conv_onet/models/decoder.py:27-44) */
/* into the MFMA-fragment order vt_decode_fwd reads from LDS.                   */
/* ------------------------------------------------------------------------- */
typedef struct vt_decoder_params {
    int32_t hidden;      /* hidden_size: 32 (vt_decoder_pack*), 32..256 (_pack_wide)  */
    int32_t c_dim;       /* c_dim: 32 (vt_decoder_pack*), 32..256 (_pack_wide)        */
    int32_t n_blocks;    /* n_blocks: 5 (vt_decoder_pack*), 1..8 (_pack_wide)         */
    int32_t p_in;        /* columns of fc_p_w: 3 (fc_p) or 3+c_dim (fc_p_img)        */
    const float *fc_p_w; /* [hidden, p_in]  fc_p.weight or fc_p_img.weight           */
    const float *fc_p_b; /* [hidden]                                                 */
    const float *fc_c_w[VT_MAX_BLOCKS]; /* [hidden, c_dim] fc_c.{i}.weight           */
    const float *fc_c_b[VT_MAX_BLOCKS]; /* [hidden]                                  */
    const float *fc0_w[VT_MAX_BLOCKS];  /* [hidden, hidden] blocks.{i}.fc_0.weight   */
    const float *fc0_b[VT_MAX_BLOCKS];
    const float *fc1_w[VT_MAX_BLOCKS];  /* [hidden, hidden] blocks.{i}.fc_1.weight   */
    const float *fc1_b[VT_MAX_BLOCKS];
    const float *fc_out_w;  /* [1, hidden] fc_out.weight                             */
    const float *fc_out_b;  /* [1]                                                   */
    const float *fc_out2_w; /* fc_out_contact.weight or NULL                         */
    const float *fc_out2_b; /* fc_out_contact.bias   or NULL                         */
} vt_decoder_params;

/* Size in bytes of the packed blob for a given (hidden, c_dim, n_blocks). */
size_t vt_decoder_blob_bytes(int hidden, int c_dim, int n_blocks);

/* params_host: HOST pointer to the struct (its members are device pointers). */
int vt_decoder_pack(const vt_decoder_params *params_host, float *blob, size_t blob_bytes, void *stream);

/* ------------------------------------------------------------------------- */
/* Feature grid layout.                                                        */
/* vt_decode_* read the grid channels-last: grid_cl[b][z][y][x][c].            */
/* Replaces: the implicit NCDHW read of F.grid_sample (decoder.py:62-68).      */
/* ------------------------------------------------------------------------- */
int vt_grid_to_channels_last(const float *grid_ncdhw, float *grid_cl,
                             int B, int C, int D, int H, int W, void *stream);
int vt_grid_from_channels_last(const float *grid_cl, float *grid_ncdhw,
                               int B, int C, int D, int H, int W, void *stream);

/* ------------------------------------------------------------------------- */
/* Fused trilinear gather + conditioned ResNet MLP, forward.                    */
/* Replaces: LocalDecoder.forward / forward_img / forward_contact               */
/*   (decoder.py:135-161, 71-103, 105-133) = normalize_3d_coordinate            */
/*   (src/common.py:293-309) + F.grid_sample (decoder.py:62-68) + fc_p +        */
/*   5 x (fc_c[i] add, ResnetBlockFC src/layers.py:41-50) + fc_out.             */
/*                                                                             */
/*   pts      [B,N,3] query points, or NULL for lattice mode;                   */
/*   lattice mode: the n-th point of every batch element is the                 */
/*     (lattice_first + n)-th point of box * make_3d_grid((-.5,)*3,(.5,)*3,     */
/*     (nx,)*3) (src/common.py:178-197, generation.py:155-157): axis 0 slowest; */
/*   c_img    [B,N,c_dim] tactile features (forward_img) or NULL;               */
/*   blob     from vt_decoder_pack (packed with p_in = 3+c_dim iff c_img);      */
/*   out      [B,N] logits; out2 [B,N] contact logits or NULL;                  */
/*   save     NULL (inference) or vt_decode_save_bytes(B*N) bytes: the training  */
/*            forward leaves the activations vt_decode_bwd needs there.          */
/* ------------------------------------------------------------------------- */
int vt_decode_fwd(const float *grid_cl, int B, int R, int C,
                  const float *pts, int64_t N,
                  int lattice_nx, float lattice_box, int64_t lattice_first,
                  const float *c_img, const float *blob, double padding,
                  float *out, float *out2, float *save, void *stream);

/* Split-bf16 ("bf16x3") inference variant of vt_decode_fwd / vt_decode_fwd_ids for the same   */
/* reference functions.  Every f32 operand v of the 16 dense 32x32 layers is carried as           */
/* bf16(v) + bf16(v - bf16(v)) and every product as W_lo*x_hi + W_hi*x_lo + W_hi*x_hi on the       */
/* bf16 matrix core with f32 accumulation (error ~2^-16 relative per product: 2e-5 abs on the      */
/* O(1) logits of the golden vectors, inside the 1e-4 parity bar; plain bf16 is ~1e-2).  The        */
/* gather, fc_p, biases, residuals and the output heads stay f32.  blob_bf16x3 comes from           */
/* vt_decoder_pack_bf16x3 (same size as the f32 blob).  c_img and finger_ids are alternatives         */
/* (both NULL = visual-only); no `save`: the training forward is the exact-f32 vt_decode_fwd.          */
int vt_decoder_pack_bf16x3(const vt_decoder_params *params_host, float *blob, size_t blob_bytes, void *stream);
int vt_decode_fwd_bf16x3(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                         int lattice_nx, float lattice_box, int64_t lattice_first,
                         const float *c_img, const unsigned char *finger_ids, const float *finger_feats, int F,
                         const float *blob_bf16x3, double padding, float *out, float *out2, void *stream);

/* Split-f16 ("f16x3") variant: the same three-product scheme with IEEE half hi / lo parts on      */
/* v_mfma_f32_32x32x16_f16 (the same matrix rate as bf16).  hi + lo carries 21-22 mantissa bits     */
/* (logit error ~1e-6 on the goldens against ~1.6e-5 for bf16x3) for operand magnitudes below       */
/* 65504 -- activations beyond that saturate, so networks with huge hidden activations belong on     */
/* bf16x3 or f32 -- and relu + split costs two VALU instructions per value instead of four            */
/* (v_cvt_pkrtz_f16_f32, v_pk_max_f16, v_fma_mix{lo,hi}_f16 with clamp): the lattice decode's       */
/* fastest form.  Same arguments and blob size as the bf16x3 pair; blob from vt_decoder_pack_f16x3.  */
/* Replaces the same reference functions (decoder.py:135-161, 71-103 on no-grad paths).              */
int vt_decoder_pack_f16x3(const vt_decoder_params *params_host, float *blob, size_t blob_bytes, void *stream);
int vt_decode_fwd_f16x3(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                        int lattice_nx, float lattice_box, int64_t lattice_first,
                        const float *c_img, const unsigned char *finger_ids, const float *finger_feats, int F,
                        const float *blob_f16x3, double padding, float *out, float *out2, void *stream);

/* "f16f8": the dense lattice decode with the two correction products of every layer on ONE fp8   */
/* (e4m3) 32x32x64 MFMA: W x = W_hi x_hi (two f16 MFMAs, as f16x3) + [W_lo 2^15 | W_hi 2^4] .       */
/* [x_hi 2^-2 | x_lo 2^9] with the MFMA's block scales undoing the shifts -- 128 matrix cycles per   */
/* layer instead of 192 and ~40 % less matrix-pipe energy (the kernel is power-limited).  The        */
/* corrections carry 4 significant bits per operand: logits within ~4e-5 of the f32 reference on    */
/* the golden decoder (inside the 1e-4 parity bar; f16x3: ~1e-6); activations beyond 448 * 4          */
/* saturate in the correction (never NaN).  Lattice mode only, for slabs the slot-pipelined kernel   */
/* covers (vt_decode_f16f8_covers: whole x-plane pairs, nx % 8 == 0, < 0.55 voxels per lattice       */
/* step, c_dim 32); everything else is vt_decode_fwd_f16x3's.  blob from vt_decoder_pack_f16f8       */
/* (same size).  Replaces the same reference functions on the generator's no-grad path               */
/* (decoder.py:135-161, 71-103 via generation.py:338-383).                                            */
int vt_decoder_pack_f16f8(const vt_decoder_params *params_host, float *blob, size_t blob_bytes, void *stream);
int vt_decode_f16f8_covers(int R, int C, int lattice_nx, float lattice_box, int64_t lattice_first, int64_t N, double padding);
int vt_decode_fwd_f16f8(const float *grid_cl, int B, int R, int C, int64_t N,
                        int lattice_nx, float lattice_box, int64_t lattice_first,
                        const float *c_img, const unsigned char *finger_ids, const float *finger_feats, int F,
                        const float *blob_f16f8, double padding, float *out, void *stream);

/* Range guard of the half-precision decodes (vt_decode_fwd_f16x3 / _f16f8): their hi operand     */
/* saturates at 65504 (round toward zero: never an infinity), after which the logits silently lose  */
/* parity.  Every launch samples its relu'd hi operands (two of a lane's sixteen channels, every     */
/* point and layer) and sets bits of a per-DEVICE status word (the current device's):                 */
/*   bit 0: a sampled half reached 65504 -- re-run with vt_decode_fwd_bf16x3 or the exact-f32 kernel; */
/*   bit 1 (vt_decode_fwd_f16f8 only): a sampled half reached 1024, where the fp8 copies of the       */
/*          correction products begin to clip and the 1e-4 contract of that kernel ends -- re-run     */
/*          with vt_decode_fwd_f16x3;                                                                  */
/*   bit 2 (vt_decode_fwd_f16f8 only): a logit beyond 2.5 in magnitude was written -- that kernel's    */
/*          error is relative (~3e-5 |logit|), its 1e-4 absolute contract ends there: re-run likewise.  */
/* vt_decode_range_status copies that word to the host (synchronises `stream`) and, with `reset`,    */
/* clears it (the host mirror's Generator3D does the re-runs); host_status NULL with `reset`: clear    */
/* only, asynchronously on `stream` (the start of a scene).  No reference counterpart: the             */
/* reference is f32 (decoder.py:135-161).                                                              */
int vt_decode_range_status(unsigned *host_status, int reset, void *stream);

/* Measurement aid (bench.py's clock evidence; no reference counterpart): the lattice kernels of      */
/* vt_decode_fwd* stamp every workgroup's lifetime with the constant-rate counter and workgroup 0's     */
/* also with the shader-clock counter; this returns the last launch's differences of workgroup 0 on     */
/* the current device and the constant counter's rate in kHz (synchronises `stream`):                   */
/* shader MHz = cycles / ticks * ref_khz / 1000; and, with max_wgs > 0, the (start, end) ticks of the   */
/* first min(max_wgs, workgroups, 512) workgroups -- the launch's ramp and tail.                         */
int vt_decode_last_clock(unsigned long long *shader_cycles, unsigned long long *ref_ticks, int *ref_khz,
                         unsigned long long *wg_ticks, int max_wgs, int *n_wgs, void *stream);

/* LocalDecoder beyond the shipped shape.  Replaces the same reference functions (decoder.py:135-161, */
/* 71-103, 105-133) for hidden_size and c_dim any multiples of 32 up to 256 (the class defaults are      */
/* 256 / 128, decoder.py:24), n_blocks <= VT_MAX_BLOCKS, and `leaky`: leaky_relu(0.2) in front of the       */
/* output heads (decoder.py:46-49, 157; the ResnetBlockFC activations are ReLU regardless, layers.py:33).    */
/* Exact f32 (v_mfma_f32_32x32x2_f32); training: below.  The weights stream from L2 in the fragment order      */
/* vt_decoder_pack_wide writes (params_host->hidden / c_dim / n_blocks / p_in describe the shape; p_in = 3,    */
/* or 3 + c_dim for forward_img); blob_bytes from vt_decoder_wide_blob_bytes (0: shape not covered).           */
/* vt_decode_fwd_wide: the arguments of vt_decode_fwd (pts or lattice, optional c_img [B,N,c_dim], out2 for     */
/* the contact head) plus the shape the blob was packed for; C = c_dim = the grid's channel count; flags:        */
/* VT_WIDE_LEAKY (leaky=True), VT_WIDE_NEAREST (sample_mode='nearest': F.grid_sample mode 'nearest', i.e. the     */
/* voxel at the half-to-even rounded coordinate, decoder.py:62-68).                                               */
#define VT_WIDE_LEAKY 1
#define VT_WIDE_NEAREST 2
size_t vt_decoder_wide_blob_bytes(int hidden, int c_dim, int n_blocks, int p_in);
int vt_decoder_pack_wide(const vt_decoder_params *params_host, float *blob, size_t blob_bytes, void *stream);
int vt_decode_fwd_wide(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                       int lattice_nx, float lattice_box, int64_t lattice_first,
                       const float *c_img, const float *blob_wide, int hidden, int n_blocks, int flags, double padding,
                       float *out, float *out2, void *stream);
/* Split-f16 form of the same forward (inference): W x = W_lo x_hi + W_hi x_lo + W_hi x_hi on v_mfma_f32_32x32x16_f16 with f32     */
/* accumulation, operands as IEEE-half hi + lo pairs (21-22 mantissa bits: f32-level logits while the hidden activations stay      */
/* inside the half range -- watched like the shipped-shape kernels': vt_decode_range_status).  A workgroup owns 64 points as two   */
/* groups that share every streamed weight fragment.  Blob: vt_decoder_pack_wide_f16x3 (its own fragment format and size).         */
/* Same arguments and coverage as vt_decode_fwd_wide.  256 / 128 / 5 at 128^3: see profiles/ (exact f32: 36 ms).                   */
size_t vt_decoder_wide_blob_f16x3_bytes(int hidden, int c_dim, int n_blocks, int p_in);
int vt_decoder_pack_wide_f16x3(const vt_decoder_params *params_host, float *blob, size_t blob_bytes, void *stream);
int vt_decode_fwd_wide_f16x3(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                             int lattice_nx, float lattice_box, int64_t lattice_first,
                             const float *c_img, const float *blob_wide_f16x3, int hidden, int n_blocks, int flags, double padding,
                             float *out, float *out2, void *stream);
/* The two forwards above with the tactile feature given per point as a finger id (255 = none) and the [n_fingers][C] table of   */
/* finger features instead of the dense c_img tensor (reference generation.py:159-255 builds c_img_all [1, nx^3, C] on the host:   */
/* 8.6 GB at 256^3 / c_dim 128); the blob is the one packed with fc_p_img (p_in = 3 + C).  Equal to the dense form bit for bit.   */
int vt_decode_fwd_wide_ids(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                           int lattice_nx, float lattice_box, int64_t lattice_first,
                           const unsigned char *finger_ids, const float *finger_feats, int n_fingers,
                           const float *blob_wide, int hidden, int n_blocks, int flags, double padding,
                           float *out, float *out2, void *stream);
/* vt_decode_fwd_wide_f16x3 with a workspace.  At hidden 64 / c_dim 32 / n_blocks <= 5 without tactile input columns (c_img NULL)   */
/* the whole network fits the registers of one workgroup: 2 n_blocks waves each keep 32 rows of one block's three layers as MFMA    */
/* operands for the whole launch and the 32-point tiles move through them as a pipeline (three barriers per tick, no weight         */
/* traffic; decode_wide_pipe.inc), fed with per-point features.  With `workspace` of vt_decode_wide_f16x3_workspace_bytes(B N,      */
/* ...) bytes (0: the shape does not use one) a pre-pass leaves the grid's samples there (same sums, same order) and the pipeline   */
/* runs on them; without it (or through vt_decode_fwd_wide_f16x3) the streaming kernel runs.  vt_decode_mlp_fwd_wide_f16x3, whose    */
/* features are given, takes the pipeline by itself.  decoder.py:24-51, 71-103.                                                     */
size_t vt_decode_wide_f16x3_workspace_bytes(int64_t total_points, int hidden, int c_dim, int n_blocks, int tactile);
int vt_decode_fwd_wide_f16x3_ws(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                                int lattice_nx, float lattice_box, int64_t lattice_first,
                                const float *c_img, const float *blob, int hidden, int n_blocks, int flags, double padding,
                                float *out, float *out2, void *workspace, size_t workspace_bytes, void *stream);
int vt_decode_fwd_wide_f16x3_ids(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                                 int lattice_nx, float lattice_box, int64_t lattice_first,
                                 const unsigned char *finger_ids, const float *finger_feats, int n_fingers,
                                 const float *blob_wide_f16x3, int hidden, int n_blocks, int flags, double padding,
                                 float *out, float *out2, void *stream);
/* The conditioned MLP of the wide shapes on features given per point, c [B][N][C], instead of the grid gather: what runs behind the  */
/* fuser in AttentionDecoder.forward_img (decoder.py:259-271) at the reference's default widths; blob packed with fc_p (p_in = 3).  */
int vt_decode_mlp_fwd_wide(const float *c, int B, int C, const float *pts, int64_t N,
                           int lattice_nx, float lattice_box, int64_t lattice_first,
                           const float *blob_wide, int hidden, int n_blocks, int flags, float *out, float *out2, void *stream);
int vt_decode_mlp_fwd_wide_f16x3(const float *c, int B, int C, const float *pts, int64_t N,
                                 int lattice_nx, float lattice_box, int64_t lattice_first,
                                 const float *blob_wide_f16x3, int hidden, int n_blocks, int flags, float *out, float *out2, void *stream);

/* The same shapes under autograd (the reference trains them through torch autograd: decoder.py:24-51,     */
/* 135-161 called from training.py:476-489, 734-740, 879).                                                   */
/*   vt_decode_fwd_wide_train  vt_decode_fwd_wide on query points that also keeps, point-major, the inputs     */
/*                       of every layer: save = vt_decode_wide_save_floats floats laid out as                    */
/*                       c [P][c_dim] | per block relu(net + fc_c(c)) [P][H], relu(fc_0(.)) [P][H] | actvn(net) [P][H]. */
/*   vt_decoder_pack_wide_t    the transposed weight fragments the backward streams (W1^T, W0^T, Wc^T per block,    */
/*                       fc_p_img's c_img columns transposed, the two head vectors).                               */
/*   vt_decode_bwd_wide  grad_out [P] (and grad_out2 for the contact head, or NULL) -> d grid (channels-last, f32     */
/*                       atomics into a ZEROED buffer; NULL: not wanted), d c_img [P][c_dim] (NULL: not wanted), and  */
/*                       gws = vt_decode_wide_gws_floats floats: dN_i [P][H] for i = 0..n_blocks (gradient of the     */
/*                       residual stream in front of block i; dN_nb: behind the last block) then dH_i [P][H]          */
/*                       (gradient of fc_0_i's output).  The weight gradients are vt_rows_wgrad over these rows:       */
/*                       fc_p <- (dN_0, [p | c_img]); fc_c_i <- (dN_i, c); fc_0_i <- (dH_i, save a0_i);                */
/*                       fc_1_i <- (dN_{i+1}, save a1_i); fc_out <- (grad_out, save actvn(net)).  No gradient with     */
/*                       respect to the query points (the reference never reads it: SURVEY.md 8a row A14).             */
size_t vt_decode_wide_save_floats(int64_t total_points, int hidden, int c_dim, int n_blocks);
size_t vt_decode_wide_gws_floats(int64_t total_points, int hidden, int c_dim, int n_blocks);
int vt_decode_fwd_wide_train(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                             const float *c_img, const float *blob_wide, int hidden, int n_blocks, int flags, double padding,
                             float *out, float *out2, float *save, void *stream);
size_t vt_decoder_wide_blob_t_bytes(int hidden, int c_dim, int n_blocks);
int vt_decoder_pack_wide_t(const vt_decoder_params *params_host, float *blob_t, size_t blob_bytes, void *stream);
int vt_decode_bwd_wide(int B, int R, int C, const float *pts, int64_t N, const float *blob_wide_t, int hidden, int n_blocks, int flags,
                       double padding, const float *grad_out, const float *grad_out2, const float *save, float *gws,
                       float *grad_grid_cl, float *grad_c_img, void *stream);
/* The conditioned MLP alone under autograd at these widths (AttentionDecoder.forward_img behind its fuser, decoder.py:259-271 with  */
/* c_dim 64 / 96 / 128): vt_decode_mlp_fwd_wide on query points that also fills `save` (same layout; its c slot is left unwritten --   */
/* the caller holds c), and the data pass that returns d c per point [B][N][C] instead of a grid scatter.  gws and the weight          */
/* gradients as vt_decode_bwd_wide's, with fc_c_i <- (dN_i, the caller's c).  vt_sample_grid_bwd[_sorted] take any c_dim % 32 == 0.    */
int vt_decode_mlp_fwd_wide_train(const float *c, int B, int C, const float *pts, int64_t N, const float *blob_wide, int hidden, int n_blocks,
                                 int flags, float *out, float *out2, float *save, void *stream);
int vt_decode_mlp_bwd_wide(int B, int C, const float *pts, int64_t N, const float *blob_wide_t, int hidden, int n_blocks, int flags,
                           const float *grad_out, const float *grad_out2, const float *save, float *gws, float *grad_c, void *stream);

/* Tactile feature assignment and decode by finger id (SURVEY.md section 8f "next" row 2).        */
/* Replaces: the scipy cdist + np.where glue that fills the dense c_img_all [1,N,32] at            */
/*   src/conv_onet/generation.py:186-200 (mode 0: nearest fingertip, radius 0.05, only if that      */
/*   finger's touch succeeded) and :245-255 (mode 1: within radius 0.015 of any of the finger's      */
/*   contact points; fingers in ascending order, later ones overwrite).                              */
/*   anchors [F,K,3], count [F] (valid anchors per finger), success [F] (u8); ids [B,N] u8, 255=none. */
/* vt_decode_fwd_ids = vt_decode_fwd with c_img[b,n,:] = finger_feats[ids[b,n]] (0 where 255).        */
/* In every *_ids entry an id >= n_fingers (F) reads as 255 does -- a zero feature -- never as a row  */
/* past the table (the host gather this replaces, table[row], raised on such an index).              */
int vt_tactile_assign(const float *pts, int B, int64_t N, int lattice_nx, float lattice_box, int64_t lattice_first,
                      const float *anchors, const int *count, const unsigned char *success, int F, int K,
                      int mode, double radius, unsigned char *ids, void *stream);
int vt_decode_fwd_ids(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                      int lattice_nx, float lattice_box, int64_t lattice_first,
                      const unsigned char *finger_ids, const float *finger_feats, int F,
                      const float *blob, double padding, float *out, void *stream);

/* The two halves of vt_decode_fwd on their own, for AttentionDecoder.forward_img      */
/* (decoder.py:237-271), which transforms the sampled features before the MLP:         */
/*   vt_sample_grid    feat[B,N,C] = trilinear sample only (decoder.py:62-68);          */
/*   vt_decode_mlp_fwd the MLP on given features c[B,N,C] (decoder.py:255-269).         */
int vt_sample_grid(const float *grid_cl, int B, int R, int C, const float *pts, int64_t N,
                   int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                   float *feat, void *stream);
int vt_decode_mlp_fwd(const float *c, int B, int C, const float *pts, int64_t N,
                      int lattice_nx, float lattice_box, int64_t lattice_first,
                      const float *blob, float *out, void *stream);
/* the same with split-f16 layers (blob from vt_decoder_pack_f16x3; inference: 2.6x the f32 form's rate at f32-level logits) */
int vt_decode_mlp_fwd_f16x3(const float *c, int B, int C, const float *pts, int64_t N,
                            int lattice_nx, float lattice_box, int64_t lattice_first,
                            const float *blob_f16x3, float *out, void *stream);
/* ... and their training forms (autograd of decoder.py:237-271 around the fuser): the MLP forward   */
/* that saves activations (save: vt_decode_save_bytes), its backward to the features it was given      */
/* (grad_c [B,N,C]; parameter gradients: vt_decode_wgrad on the same save/gws), and the backward of     */
/* vt_sample_grid (scatter-add of grad_feat into grad_grid_cl, which the caller zeroes).                 */
int vt_decode_mlp_fwd_train(const float *c, int B, int C, const float *pts, int64_t N,
                            int lattice_nx, float lattice_box, int64_t lattice_first,
                            const float *blob, float *out, float *save, void *stream);
int vt_decode_mlp_bwd(int B, int C, const float *pts, int64_t N,
                      int lattice_nx, float lattice_box, int64_t lattice_first,
                      const float *blob_t, const float *grad_out, const float *save, float *gws,
                      float *grad_c, void *stream);
int vt_sample_grid_bwd(int B, int R, int C, const float *pts, int64_t N,
                       int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                       const float *grad_feat, float *grad_grid_cl, void *stream);
/* vt_sample_grid_bwd with the points grouped by trilinear cell: order / seg_lo / seg_hi from vt_voxel_build(pts, B, N, R - 1,       */
/* padding, ...) (a point's bin at resolution R - 1 is the cell whose eight corners it touches).  One wave per cell sums its points'    */
/* contributions in ascending point order and issues its atomics once per cell: the form for clustered query points (the contact        */
/* clouds of training.py:817-866 put up to 128 points of a scene into a few cells, whose per-point atomics collide).  Same sums.         */
int vt_sample_grid_bwd_sorted(int B, int R, int C, const float *pts, int64_t N, double padding, const float *grad_feat,
                              const int *order, const int *seg_lo, const int *seg_hi, float *grad_grid_cl, void *stream);
/* vt_decode_bwd_contact that leaves d c (the gradient of the sampled features, [B*N][32]) in `grad_c` instead of scattering it:  */
/* the caller scatters with vt_sample_grid_bwd_sorted (query points given explicitly; grad_out2 / grad_c_img may be NULL).            */
int vt_decode_bwd_dc(int B, int R, int C, const float *pts, int64_t N, double padding,
                     const float *blob_t, const float *grad_out, const float *grad_out2, const float *save, float *gws,
                     float *grad_c, float *grad_c_img, void *stream);

/* ------------------------------------------------------------------------- */
/* TransformerFusion forward (eval mode).                                       */
/* Replaces: self.fuser(c_img, 1, c, 1) at decoder.py:258, i.e.                  */
/*   TransformerFusion.forward (src/TransformerFusion.py:311-333) with           */
/*   num_layers=1, d_model=32, key_feature_dim=64, with_pos_embed=False:          */
/*   RelationUnit (:92-113, incl. the column re-normalisation :104),              */
/*   TransNonlinear (:21-25), InstanceNorm1d over the N points of the chunk.       */
/* self_attn is shared by the encoder layer and the decoder layer's               */
/* self-attention (the reference builds both from ONE module, :291-309).           */
/* c_img, c, out: [B,N,32].  Attention couples the N points of a chunk: N is part   */
/* of the function's definition.                             */
/* ------------------------------------------------------------------------- */
typedef struct vt_fusion_unit {
    const float *WK;         /* [64,32] head.0.WK.weight                              */
    const float *WQ;         /* [64,32] head.0.WQ.weight                              */
    const float *WV;         /* [32,32] head.0.WV.weight                              */
    const float *trans_conv; /* [32,32] head.0.trans_conv.weight                      */
    const float *linear1_w;  /* [64,32] extra_nonlinear.0.linear1.weight              */
    const float *linear1_b;  /* [64]                                                  */
    const float *linear2_w;  /* [32,64] extra_nonlinear.0.linear2.weight              */
    const float *linear2_b;  /* [32]                                                  */
    const float *norm2_w;    /* [32] extra_nonlinear.0.norm2.weight                   */
    const float *norm2_b;    /* [32]                                                  */
} vt_fusion_unit;

typedef struct vt_fusion_params {
    int32_t d_model;         /* 32 (every entry point) or 64 / 96 / 128 (vt_fusion_fwd only) */
    int32_t key_dim;         /* must be 64 */
    vt_fusion_unit self_attn;   /* fuser.encoder.layers.0.self_attn (== decoder.layers.0.self_attn) */
    vt_fusion_unit cross_attn;  /* fuser.decoder.layers.0.cross_attn                                 */
} vt_fusion_params;

size_t vt_fusion_workspace_bytes(int B, int N);
/* d_model beyond 32 (the reference's AttentionDecoder defaults to c_dim = d_model = 128, decoder.py:176-207; key_feature_dim stays 64):  */
/* vt_fusion_fwd takes d_model in {32, 64, 96, 128} (c_img, c, out: [B][N][d_model]; vt_fusion_unit's shapes with 32 -> d_model: WK / WQ  */
/* [64][d], WV / trans_conv [d][d], linear1 [64][d], linear2 [d][64], norm2 [d]) with a workspace of vt_fusion_workspace_bytes_wide.      */
/* Eval mode; the wider rows run generic-width projection / epilogue kernels around the same N x N passes.                                */
size_t vt_fusion_workspace_bytes_wide(int B, int N, int d_model);
int vt_fusion_fwd(const float *c_img, const float *c, int B, int N, const vt_fusion_params *params_host,
                  void *workspace, size_t workspace_bytes, float *out, void *stream);
/* The same forward with the tactile features of the decoder's self-attention given per point as a finger id (255 = none: a zero  */
/* row) and the [n_fingers][32] table -- what the reference gathers into c_img_all on the host (generation.py:159-255, consumed by */
/* AttentionDecoder.forward_img, decoder.py:237-271).  finger_ids [rows][N]; batch element b reads row chunk_index[b] (device      */
/* int32 [B]: the chunks a generator picked out of a lattice) or row b when chunk_index is NULL.  Equal to vt_fusion_fwd on the    */
/* gathered tensor bit for bit.                                                                                                    */
int vt_fusion_fwd_ids(const unsigned char *finger_ids, const float *finger_feats, int n_fingers, const int *chunk_index,
                      const float *c, int B, int N, const vt_fusion_params *params_host,
                      void *workspace, size_t workspace_bytes, float *out, void *stream);

/* ------------------------------------------------------------------------- */
/* TransformerFusion under autograd (training).                                  */
/* Replaces: PyTorch autograd of self.fuser(c_img, 1, c, 1) (decoder.py:258)      */
/*   triggered by loss.backward() at src/conv_onet/training.py:79,89,96:          */
/*   RelationUnit.forward (src/TransformerFusion.py:92-113: l2-normalised WK/WQ,  */
/*   softmax over keys, the column re-normalisation :104, WV, trans_conv, relu),  */
/*   TransNonlinear.forward (:21-25, with its two train-mode dropouts :13-19),     */
/*   the encoder / decoder layers' InstanceNorm1d + relu (:144-145, :209-218).      */
/* vt_fusion_fwd_train = vt_fusion_fwd that (a) applies the dropouts with          */
/*   probability p_drop (0 = eval), masks a pure function of `seed`, and            */
/*   (b) leaves, in `saved` (vt_fusion_saved_bytes, caller-owned, kept until the     */
/*   backward), per attention call the softmax row sums, column sums, value rows,     */
/*   attention outputs and pre-InstanceNorm sums -- O(N) state, never N x N.          */
/* vt_fusion_bwd: d_out [B,N,32] -> d_c_img, d_c [B,N,32] (overwritten) and the     */
/*   gradient of every parameter, written to the buffers of `grads` (overwritten;   */
/*   the self-attention unit is ONE module used twice, :291-309: its gradient is     */
/*   the sum of both uses).  The N x N scores are recomputed tile by tile on the      */
/*   matrix core; reductions are in a fixed order (bit-reproducible).                 */
/* vt_fusion_dropout_mask: the factors (0 or 1/(1-p)) the kernels apply for           */
/*   (call 0 = encoder self-attention, 1 = decoder self-attention, 2 = cross;          */
/*   which 0 = TransNonlinear.dropout [points,64], 1 = dropout2 [points,32]) --         */
/*   lets a test feed the same masks to a reference implementation.                     */
/* ------------------------------------------------------------------------- */
typedef struct vt_fusion_unit_grads {
    float *WK, *WQ, *WV, *trans_conv, *linear1_w, *linear1_b, *linear2_w, *linear2_b, *norm2_w, *norm2_b;
} vt_fusion_unit_grads;

typedef struct vt_fusion_grads {
    vt_fusion_unit_grads self_attn;
    vt_fusion_unit_grads cross_attn;
} vt_fusion_grads;

size_t vt_fusion_saved_bytes(int B, int N);
size_t vt_fusion_bwd_workspace_bytes(int B, int N);
int vt_fusion_fwd_train(const float *c_img, const float *c, int B, int N, const vt_fusion_params *params_host, float p_drop,
                        unsigned long long seed, void *workspace, size_t workspace_bytes, void *saved, size_t saved_bytes,
                        float *out, void *stream);
int vt_fusion_bwd(const float *d_out, const float *c_img, const float *c, int B, int N, const vt_fusion_params *params_host,
                  float p_drop, unsigned long long seed, const void *saved, size_t saved_bytes, void *workspace,
                  size_t workspace_bytes, float *d_c_img, float *d_c, const vt_fusion_grads *grads_host, void *stream);
int vt_fusion_dropout_mask(float p_drop, unsigned long long seed, int call, int which, int points, float *mask, void *stream);
/* d_model 64 / 96 / 128 under autograd (the reference trains AttentionDecoder at its defaults c_dim 128 / hidden 256 like any module,   */
/* decoder.py:176-207): vt_fusion_fwd_train / vt_fusion_bwd take those widths too (tensors [B][N][d_model], the unit's shapes with 32 ->   */
/* d_model), with `saved` of vt_fusion_saved_bytes_wide, workspaces of vt_fusion_workspace_bytes_wide / vt_fusion_bwd_workspace_bytes_wide.  */
/* The backward's three N x N passes are linear in their payload and run once per 32-channel slice of it (per-key / per-query scalars    */
/* enter with the first slice); the per-point parts are one-wave-per-point kernels with the weights in LDS; the weight gradients are        */
/* vt_rows_wgrad products.  Dropout masks of these widths: vt_fusion_dropout_mask_wide (which 1: [points][d_model]).                        */
size_t vt_fusion_saved_bytes_wide(int B, int N, int d_model);
size_t vt_fusion_bwd_workspace_bytes_wide(int B, int N, int d_model);
int vt_fusion_dropout_mask_wide(float p_drop, unsigned long long seed, int call, int which, int points, int d_model, float *mask, void *stream);

/* ------------------------------------------------------------------------- */
/* Backward of vt_decode_fwd (training).                                        */
/* Replaces: PyTorch autograd of LocalDecoder.forward / forward_img, triggered   */
/*   by loss.backward() at src/conv_onet/training.py:79,89,96 (grid_sampler_3d   */
/*   backward, addmm backward, relu backward).                                   */
/*   vt_decoder_pack_t : transposed-weight blob for the data gradient;           */
/*   vt_decode_bwd     : grad_out [B,N] -> grad_grid_cl [B,R,R,R,C] (ACCUMULATED  */
/*                       with f32 atomics: zero it first; may be NULL),           */
/*                       grad_c_img [B,N,C] (or NULL), and per-layer output       */
/*                       gradients in `gws` (vt_decode_gws_bytes);                */
/*   vt_decode_wgrad   : all parameter gradients into `grads`                     */
/*                       (vt_decode_wgrad_floats(p_in) floats, nn.Linear layout): */
/*                       fc_p.w[32*p_in] fc_p.b[32] fc_c.w[5][32*32] fc_c.b[5][32] */
/*                       fc_0.w[5][..] fc_0.b[5][32] fc_1.w[5][..] fc_1.b[5][32]   */
/*                       fc_out.w[32] fc_out.b[1]; p_in = 35 iff c_img != NULL.   */
/*                       Partials are summed in a fixed order: bit-reproducible.  */
/* ------------------------------------------------------------------------- */
size_t vt_decoder_blob_t_bytes(int hidden, int c_dim, int n_blocks);
int vt_decoder_pack_t(const vt_decoder_params *params_host, float *blob_t, size_t blob_bytes, void *stream);
size_t vt_decode_save_bytes(int64_t total_points);
size_t vt_decode_gws_bytes(int64_t total_points);
int vt_decode_bwd(int B, int R, int C, const float *pts, int64_t N,
                  int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                  const float *blob_t, const float *grad_out, const float *save, float *gws,
                  float *grad_grid_cl, float *grad_c_img, void *stream);
size_t vt_decode_wgrad_workspace_bytes(int64_t total_points);
size_t vt_decode_wgrad_floats(int p_in);
int vt_decode_wgrad(int B, const float *pts, int64_t N, int lattice_nx, float lattice_box, int64_t lattice_first,
                    const float *c_img, const float *grad_out, const float *save, const float *gws,
                    void *workspace, size_t workspace_bytes, float *grads, void *stream);

/* The same two calls with the contact head (LocalDecoder.forward_contact under autograd,  */
/* decoder.py:105-133 -> training.py:896-948): grad_out2 [B,N] is the gradient of the      */
/* contact logits; `grads` has vt_decode_wgrad_floats_contact(p_in) floats: the layout     */
/* above followed by fc_out_contact.w[32] fc_out_contact.b[1].  blob_t must come from a    */
/* vt_decoder_pack_t call whose params carry fc_out2_w.  grad_out2 NULL = the plain calls. */
int vt_decode_bwd_contact(int B, int R, int C, const float *pts, int64_t N,
                          int lattice_nx, float lattice_box, int64_t lattice_first, double padding,
                          const float *blob_t, const float *grad_out, const float *grad_out2, const float *save, float *gws,
                          float *grad_grid_cl, float *grad_c_img, void *stream);
size_t vt_decode_wgrad_floats_contact(int p_in);
int vt_decode_wgrad_contact(int B, const float *pts, int64_t N, int lattice_nx, float lattice_box, int64_t lattice_first,
                            const float *c_img, const float *grad_out, const float *grad_out2, const float *save, const float *gws,
                            void *workspace, size_t workspace_bytes, float *grads, void *stream);

/* ------------------------------------------------------------------------- */
/* Marching cubes (Lewiner), vertex numbering identical to scikit-image's.      */
/* Replaces: skimage.measure.marching_cubes(value_grid,                         */
/*   gradient_direction='ascent') + the rescale `v -= nx/2; v *= 1.1/nx`        */
/*   (src/conv_onet/generation.py:268-272; src/conv_onet/inferencing.py:172-179).*/
/*                                                                             */
/* Two phases, because the output size is data dependent:                       */
/*   vt_mc_count  classifies the cells and scans the counts into `workspace`;   */
/*   vt_mc_read_counts copies {nverts, nfaces, level} to the host (this one      */
/*                SYNCHRONISES the stream -- skip it when capacities are known); */
/*   vt_mc_emit   writes verts [nverts,3] f32 (array-axis order, like skimage)   */
/*                and faces [nfaces,3] i32; entries beyond the capacities are    */
/*                dropped.  rescale != 0 applies (v - shift) * scale in f32.     */
/* vol is [n0,n1,n2] f32; auto_level != 0 uses skimage's default                */
/* 0.5*(min+max) instead of `level`.  workspace: vt_mc_workspace_bytes().       */
/* ------------------------------------------------------------------------- */
size_t vt_mc_workspace_bytes(int n0, int n1, int n2);
int vt_mc_count(const float *vol, int n0, int n1, int n2, double level, int auto_level,
                void *workspace, size_t workspace_bytes, void *stream);
int vt_mc_read_counts(const void *workspace, int *nverts_host, int *nfaces_host, double *level_host, void *stream);
/* the same read in two halves: _begin queues the copy (page-locked slot, an event right behind it) and returns a token; _end      */
/* waits for that event only -- launches made between the two (the speculative vt_mc_emit) run under the wait instead of in front  */
/* of it.  Sixteen tokens may be outstanding; a seventeenth _begin fails (VT_ERR_INVALID), _end spends its token.                 */
int vt_mc_read_counts_begin(const void *workspace, void *stream, int *token);
int vt_mc_read_counts_end(int token, int *nverts_host, int *nfaces_host, double *level_host);
/* vt_mc_count whose last kernel also writes the counts, the level and then a sequence number into a page-locked host slot: no copy */
/* command and no event in the stream (the emit kernels queued behind it start when the scan ends); returns the token               */
/* vt_mc_read_counts_end takes, which then polls the slot for that sequence number instead of waiting for an event.                 */
int vt_mc_count_notify(const float *vol, int n0, int n1, int n2, double level, int auto_level,
                       void *workspace, size_t workspace_bytes, void *stream, int *token);
/* The same for a scene whose launches are CAPTURED in a hipGraph (a kernel argument -- vt_mc_count_notify's sequence number -- would  */
/* be frozen into the graph): vt_mc_echo_slot makes a page-locked slot (outside the capture; it lives as long as the graph),            */
/* vt_mc_count_echo is vt_mc_count whose scan kernel reads the number to echo from that slot, vt_mc_echo_arm sets a fresh number        */
/* before every replay and vt_mc_echo_wait spins until the slot's header carries it, then returns the counts.  One replay in flight     */
/* per slot.  (No copy command and no event between the scan and the emit kernels: 19 us of a 0.9 ms scene.)                             */
int vt_mc_echo_slot(int *token);
/* vt_mc_echo_release hands a slot back once the graph that echoes into it has been destroyed (no replay in flight): the next   */
/* vt_mc_echo_slot reuses its page-locked block.                                                                                */
int vt_mc_echo_release(int token);
int vt_mc_count_echo(const float *vol, int n0, int n1, int n2, double level, int auto_level,
                     void *workspace, size_t workspace_bytes, void *stream, int token);
int vt_mc_echo_arm(int token);
int vt_mc_echo_wait(int token, void *stream, int *nverts_host, int *nfaces_host, double *level_host);
int vt_mc_emit(const float *vol, int n0, int n1, int n2, void *workspace,
               float *verts, int max_verts, int *faces, int max_faces,
               int rescale, float shift, float scale, void *stream);

/* ------------------------------------------------------------------------- */
/* PointNet local-pool voxeliser.                                              */
/* Replaces: normalize_3d_coordinate + coordinate2index (src/common.py:293-309, */
/*   333-348; call site src/encoder/pointnet.py:151-152), torch_scatter          */
/*   scatter_max + gather in pool_local (pointnet.py:116-132) and scatter_mean   */
/*   in generate_grid_features (pointnet.py:102-110), and their autograd.        */
/*                                                                             */
/* vt_voxel_build runs once per forward: idx[b,t] = ix + R*(iy + R*iz) (bit-exact */
/* with the reference's f32 maths), order[b,j] = point ids sorted by (voxel,      */
/* point), seg_lo/seg_hi[b,t] = the sorted range of t's voxel.  Up to 8192 points  */
/* per scene sort inside one workgroup's LDS (one launch); larger clouds take the  */
/* same stable radix sort through global memory (2 + 3 launches per five id bits). */
/* pool_max: out[b,t,c] = max_{t' in voxel(t)} feat[b,t',c], argmax = winning t'.  */
/* scatter_mean: grid[b,c,z,y,x] (NCDHW, zero-filled inside) = per-voxel mean.     */
/* All reductions run in ascending point order: bit-reproducible.                 */
/* ------------------------------------------------------------------------- */
int vt_voxel_build(const float *pts, int B, int T, int R, double padding,
                   int *idx, int *order, int *seg_lo, int *seg_hi, void *stream);
/* vt_voxel_build that also zero-fills `clear` (clear_bytes, both multiples of 16; e.g. the grid the scatter-mean fills next) with */
/* the workgroups the sort leaves idle: the torch.zeros of generate_grid_features (pointnet.py:102-110) without a launch of its own. */
int vt_voxel_build_clear(const float *pts, int B, int T, int R, double padding,
                         int *idx, int *order, int *seg_lo, int *seg_hi, void *clear, size_t clear_bytes, void *stream);
/* flags [B][(R/8)^3] bytes: bit 0 where no point of the scene lies in the 10^3 halo of that 8^3 voxel block, i.e. the mean grid  */
/* of generate_grid_features (pointnet.py:102-110) is zero over everything a 3x3x3 conv of the block reads; bit 1 where none lies */
/* in its 12^3 halo either (what a second 3x3x3 conv behind the first depends on): values 0, 1 and 3 (idx from vt_voxel_build;    */
/* R a multiple of 8, at most 128).  Consumed by vt_unet3d_fwd_skip / vt_conv3d_gcr_f16x3_skip (any non-zero flag skips there).  */
int vt_voxel_tile_flags(const int *idx, int B, int T, int R, unsigned char *flags, void *stream);
/* vt_voxel_build_clear (clear may be NULL with clear_bytes 0) that leaves those flags as well: the sorting workgroup marks the blocks */
/* while it computes the voxel ids (no launch of its own, no second pass over the points).                                       */
int vt_voxel_build_clear_flags(const float *pts, int B, int T, int R, double padding, int *idx, int *order, int *seg_lo, int *seg_hi,
                               void *clear, size_t clear_bytes, unsigned char *tile_flags, void *stream);
int vt_voxel_pool_max_fwd(const float *feat, const int *order, const int *seg_lo, const int *seg_hi,
                          int B, int T, int C, float *out, int *argmax, void *stream);
/* pool_local over K <= 4 cell partitions of the same points, summed in the order k = 0, 1, ... (the hand encoder's xz + xy + yz planes,    */
/* pointnet.py:116-132: `c += pooled`): out[b][t][c] = sum_k max of feat over t's cell in partition k; order / seg_lo / seg_hi / argmax:    */
/* host arrays of K device pointers (argmax, or its entries, may be NULL in the forward).  Same maxima and first arg-maxima as K calls of   */
/* vt_voxel_pool_max_fwd + K - 1 adds; one launch.  The backward sums, per partition in order, the cell's grad_out at its arg-max point.     */
int vt_voxel_pool_max_sum_fwd(const float *feat, int K, const int *const *order, const int *const *seg_lo, const int *const *seg_hi,
                              int B, int T, int C, float *out, int *const *argmax, void *stream);
int vt_voxel_pool_max_sum_bwd(const float *grad_out, int K, int *const *argmax, const int *const *order, const int *const *seg_lo,
                              const int *const *seg_hi, int B, int T, int C, float *grad_feat, void *stream);
/* pool_local with scatter_type = 'mean' (pointnet.py:64-69, 116-132: scatter_mean over the cells, gathered back to the points):  */
/* out[b][t][c] = mean of feat over the points of t's cell (cells of a volume or a plane: the segments of vt_voxel_build /          */
/* vt_plane_build).  Its backward is the same call on the gradient.                                                              */
int vt_voxel_pool_mean(const float *feat, const int *order, const int *seg_lo, const int *seg_hi, int B, int T, int C, float *out, void *stream);
int vt_voxel_pool_max_bwd(const float *grad_out, const int *argmax, const int *order,
                          const int *seg_lo, const int *seg_hi,
                          int B, int T, int C, float *grad_feat, void *stream);
int vt_voxel_scatter_mean_fwd(const float *feat, const int *idx, const int *order,
                              const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *grid, void *stream);
int vt_voxel_scatter_mean_bwd(const float *grad_grid, const int *idx, const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *grad_feat, void *stream);

/* ------------------------------------------------------------------------- */
/* UNet3D forward (inference), channels-last.                                   */
/* Replaces (SURVEY.md section 8f "next" row 1): the SingleConv 'gcr' blocks     */
/*   GroupNorm -> Conv3d(3,pad 1,no bias) -> ReLU (src/encoder/unet3d.py:20-72),  */
/*   MaxPool3d(2) (:219-238), nearest upsample + concat (:283-293, 321-323) and   */
/*   the final 1x1x1 conv (:440-441) of UNet3D.forward (:449-474).                */
/* All activations are [B,D,H,W,C] f32; channel counts multiples of 32.           */
/*   vt_conv3d_pack      Conv3d weight [Cout,Cin,3,3,3] -> MFMA fragment order;    */
/*   vt_gn_scale_shift   GroupNorm statistics of the (virtually concatenated)      */
/*                       input -> scale_shift[B][Cin][2];                          */
/*   vt_conv3d_gcr       out = relu?(conv3x3x3(x * scale + shift)), zero padding   */
/*                       applied after the normalisation; input = skip [..,C1],    */
/*                       optionally followed by `low` [B,D/2,H/2,W/2,C2]            */
/*                       nearest-upsampled (neither is materialised);               */
/*   vt_maxpool3d_cl, vt_conv1x1_cl: the remaining two layer types;                 */
/*   vt_voxel_scatter_mean_cl_fwd: scatter-mean writing the channels-last grid.     */
/* ------------------------------------------------------------------------- */
size_t vt_conv3d_packed_floats(int Cout, int Cin);
int vt_conv3d_pack(const float *w, int Cout, int Cin, float *packed, void *stream);
/* GroupNorm statistics travel as per-block partial sums part[B][nblk][C][2] = (sum, sumsq)   */
/* written by the PRODUCER of a tensor: vt_conv3d_gcr's epilogue (nblk =                       */
/* vt_conv3d_stat_blocks) or vt_channel_stats (any nblk) for pooled tensors / the input.       */
/* vt_gn_scale_shift reduces them in a fixed order into scale_shift[B][C1+C2][2]; part2 (may   */
/* be NULL) is the nearest-upsampled `low` source, whose sums count 8 times.                    */
size_t vt_stats_floats(int B, int D, int H, int W, int C);
int vt_conv3d_stat_blocks(int B, int D, int H, int W, int Cin, int Cout);
int vt_channel_stats(const float *x, int B, int64_t V, int C, int nblk, float *part, void *stream);
int vt_gn_scale_shift(const float *part1, int nblk1, int C1, const float *part2, int nblk2, int C2,
                      int B, int64_t voxels, int groups, const float *gamma, const float *beta, double eps,
                      float *scale_shift, void *stream);
int vt_conv3d_gcr(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                  const float *scale_shift, const float *packed_w, int Cout, int relu, float *out,
                  float *out_part, void *stream);
/* The whole UNet3D.forward (unet3d.py:449-474) in one call: encoder levels (max-pool from level 1  */
/* on, two gcr convs each), decoder levels (virtual upsample+concat, two gcr convs), final 1x1x1     */
/* conv.  `packed` = vt_conv3d_pack of the level's Conv3d weight.  x_cl [B,R,R,R,enc[0][0].cin].      */
#define VT_UNET_MAX_LEVELS 6
typedef struct vt_unet3d_conv {
    const float *gn_w;    /* groupnorm.weight [cin] */
    const float *gn_b;    /* groupnorm.bias   [cin] */
    const float *packed;  /* vt_conv3d_pack(conv.weight [cout,cin,3,3,3]) */
    int32_t cin, cout;
    const float *packed_bf16x3;  /* vt_conv3d_pack_bf16x3(conv.weight) or NULL: run this conv on the bf16 matrix core with */
                                 /* split-bf16 operands where vt_conv3d_stat_blocks_bf16x3(...) != 0, exact f32 elsewhere */
    const float *packed_f16x3;   /* vt_conv3d_pack_f16x3(conv.weight) or NULL: where vt_conv3d_stat_blocks_f16x3(...) != 0 this conv runs */
                                 /* the persistent split-f16 kernel (takes precedence over packed_bf16x3)                              */
    const float *packed_f16x3_thin; /* vt_conv3d_pack_f16x3_thin(conv.weight) or NULL: the shapes of packed_bf16x3 (the thin-tile and */
                                 /* K-split kernels) on IEEE-half pairs instead of bf16 pairs; takes precedence over packed_bf16x3     */
    const float *packed_f16x3_up;   /* decoder-entry layers (dec[k][0]) only, or NULL: vt_conv3d_pack_f16x3_up(conv.weight, C1 = the skip's */
                                 /* channels); where vt_conv3d_up_covers(...) the layer runs in per-parity form (vt_conv3d_gcr_f16x3_up)  */
} vt_unet3d_conv;
typedef struct vt_unet3d_params {
    int32_t n_levels;     /* len(f_maps) */
    int32_t groups;       /* num_groups (1 is used when a layer has fewer channels) */
    double eps;           /* GroupNorm eps */
    vt_unet3d_conv enc[VT_UNET_MAX_LEVELS][2];   /* encoders.{i}.basic_module.SingleConv{1,2} */
    vt_unet3d_conv dec[VT_UNET_MAX_LEVELS][2];   /* decoders.{k}.basic_module.SingleConv{1,2} */
    const float *final_w; /* final_conv.weight [out_channels, f_maps[0]] */
    const float *final_b; /* final_conv.bias or NULL */
    int32_t out_channels;
    const float *final_packed_f16x3; /* vt_conv1x1_pack_f16x3(final_w) or NULL: the final conv runs in the last layer's epilogue */
                                     /* where that layer is on the specialised-wave split-f16 kernel and both are 32 channels wide */
} vt_unet3d_params;
/* Split-bf16 form of vt_conv3d_pack / vt_conv3d_stat_blocks / vt_conv3d_gcr (same arguments): the 3x3x3 convolution  */
/* as W_lo*x_hi + W_hi*x_lo + W_hi*x_hi on the bf16 matrix core with f32 accumulation (hi = bf16(v), lo = bf16(v - hi));  */
/* GroupNorm, ReLU and the output statistics stay f32.  Covers volumes whose sides are multiples of 8 and that give at     */
/* least 64 workgroups of 8x8x8 / 8x8x4 (large levels) or 8x8x2 (16^3-class levels) tiles per cout block                   */
/* (stat_blocks returns 0 and gcr VT_ERR_UNSUPPORTED otherwise: use the f32 form).  6.6e-5 abs on the UNet3D golden       */
/* (scale 2.4), 4e-5 on the logits decoded from the resulting grid.                                                       */
int vt_conv3d_pack_bf16x3(const float *w, int Cout, int Cin, float *packed, void *stream);
int vt_conv3d_stat_blocks_bf16x3(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_gcr_bf16x3(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                         const float *scale_shift, const float *packed_w_bf16x3, int Cout, int relu, float *out,
                         float *out_part, void *stream);
/* K-split form of vt_conv3d_gcr_bf16x3 for the levels whose output tiles cannot fill the chip (the 16^3 and 8^3 levels of  */
/* ONE scene, unet3d.py:449-474 at num_levels 4): the 16-channel blocks of the input are dealt over 2-8 workgroups per     */
/* output tile, the raw partial sums go through `workspace` and a second launch adds them in slice order (bit-reproducible), */
/* applies the ReLU and writes the GroupNorm partial sums of every 128-voxel block (stat_blocks = D*H*W/128).  Same packed   */
/* weights as vt_conv3d_gcr_bf16x3.  workspace_bytes / stat_blocks return 0 where the plain kernels already fill the chip   */
/* (fewer than eight 16-channel blocks, or VTACO_CONV_KSPLIT=0): use them there.  One scene, both launches: 384->128 at   */
/* 16^3 77 -> 40 us, 128->128 at 16^3 29 -> 22 us, 128->128 / 128->256 at 8^3 33 -> 13 us (f32 K-split kernel before).          */
/* The same two kernels on IEEE-half hi + lo pairs (21-22 mantissa bits instead of 16; same fragment layout, sizes, coverage and    */
/* arguments): for the thin levels of a network whose large levels run the split-f16 kernels -- with bf16 pairs there they were  */
/* most of the encoder's remaining drift against the f32 reference.  Inputs are GroupNorm outputs: inside the half range.        */
int vt_conv3d_pack_f16x3_thin(const float *w, int Cout, int Cin, float *packed, void *stream);
int vt_conv3d_gcr_f16x3_thin(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                             const float *scale_shift, const float *packed_w_f16x3_thin, int Cout, int relu, float *out,
                             float *out_part, void *stream);
int vt_conv3d_gcr_f16x3_thin_ksplit(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                                    const float *scale_shift, const float *packed_w_f16x3_thin, int Cout, int relu, float *out,
                                    float *out_part, void *workspace, size_t workspace_bytes, void *stream);
size_t vt_conv3d_ksplit_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_stat_blocks_ksplit(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_gcr_bf16x3_ksplit(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                                const float *scale_shift, const float *packed_w_bf16x3, int Cout, int relu, float *out,
                                float *out_part, void *workspace, size_t workspace_bytes, void *stream);
/* Split-f16 form ("f16x3") for the large volumes (same reference layer: unet3d.py:20-72 SingleConv 'gcr'): operands as */
/* IEEE-half hi + lo pairs (21-22 mantissa bits: f32-rounding-level error on GroupNorm outputs and conv weights) on      */
/* v_mfma_f32_32x32x16_f16, in a persistent, double-buffered kernel: input channels in chunks of eight with a PAIR of     */
/* taps per MFMA k-step, the next chunk's image and weight fragments written to the other LDS buffer while the current    */
/* chunk's taps run, one barrier per chunk, workgroups walking their tiles of a scene without per-tile prologues; the       */
/* output's GroupNorm partial sums are per workgroup (stat_blocks = workgroups per scene).  Covers side lengths that are     */
/* multiples of 8 with >= 256 tiles of 8x8x8 or >= 128 tiles of 8x8x4 per launch (stat_blocks returns 0 otherwise).          */
/* The packed blob has its own size (28 instead of 27 tap slots per channel pair): vt_conv3d_packed_floats_f16x3.            */
/* vt_conv3d_gcr_f16x3 on a plain layer (no `low`) whose input is ZERO before the normalisation over the halo of the flagged 8^3    */
/* blocks (tile_flags [B][(D/8)(H/8)(W/8)], vt_voxel_tile_flags): the normalised input there is the per-channel shift, so a block's */
/* output is, per voxel, the sum over the taps inside the volume of T[tap][cout] = sum_cin W[cout][cin][tap] shift[cin] -- the     */
/* workgroups of a scene deal the blocks that need their taps among themselves and fill the others from T (the UNet3D's first      */
/* layer, unet3d.py:449-474 on the grid of pointnet.py:102-110: a 3000-point cloud leaves 3/4 of the 512 blocks of a 64^3 grid     */
/* empty).  Equal to the dense kernel to f32 rounding of T (the dense kernel sums the same products in another order).  Shapes    */
/* the specialised-wave kernel does not run, or lists of more than 64 blocks per workgroup, fall back to the dense walk.           */
int vt_conv3d_gcr_f16x3_skip(const float *x, int C, int B, int D, int H, int W, const float *scale_shift, const float *packed_w_f16x3,
                             int Cout, int relu, const unsigned char *tile_flags, float *out, float *out_part, void *stream);
size_t vt_conv3d_packed_floats_f16x3(int Cout, int Cin);
int vt_conv3d_pack_f16x3(const float *w, int Cout, int Cin, float *packed, void *stream);
/* the fragments of the conv that computes this layer's DATA GRADIENT (autograd of unet3d.py:20-72: conv_transpose = the forward    */
/* kernels on W with channels swapped and taps flipped) straight from w [Cout][Cin][27]: vt_conv3d_packed_floats_f16x3(Cin, Cout)   */
/* floats, equal to vt_conv3d_pack_f16x3 of w.flip(2,3,4).transpose(0,1) without that copy.                                         */
int vt_conv3d_pack_f16x3_t(const float *w, int Cout, int Cin, float *packed, void *stream);
int vt_conv3d_stat_blocks_f16x3(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_gcr_f16x3(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                        const float *scale_shift, const float *packed_w_f16x3, int Cout, int relu, float *out,
                        float *out_part, void *stream);
/* The same kernel for inputs far below the half range (the data-gradient convolution of the backward: dxn =           */
/* conv(g, W^T flipped) with g = dy * [y > 0], autograd of unet3d.py:20-72): `in_absmax` is a device scalar holding         */
/* max |input|; the kernel multiplies the input by the power of two that brings it to ~2^10 before the split and            */
/* divides the result by it (both exact), so the split keeps ~21 bits relative to the tensor's largest element.             */
int vt_conv3d_gcr_f16x3_scaled(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                               const float *scale_shift, const float *packed_w_f16x3, int Cout, int relu, float *out,
                               float *out_part, const float *in_absmax, void *stream);
/* Decoder-entry layers in per-parity form (reference unet3d.py:195-293: Decoder.forward = nearest upsample of the lower level,   */
/* torch.cat((encoder_features, x)), DoubleConv; :449-474 the call order).  The layer reads the virtual [skip | upsample(low)]; over   */
/* the upsampled channels a 3x3x3 conv is, per output parity class (x&1, y&1, z&1), a 2x2x2 conv over `low` itself with the taps that  */
/* share a low voxel summed in the weights -- 8 instead of 27 taps and a 6^3 instead of a 10^3 halo for those channels.  `packed_up`   */
/* = vt_conv3d_pack_f16x3_up(w [Cout][C1+C2][3][3][3], C1): the eight classes' merged taps of the low channels as half pairs           */
/* (vt_conv3d_up_packed_floats(Cout, C2) floats); the skip channels keep vt_conv3d_pack_f16x3's fragments (`packed_w_f16x3`, the       */
/* whole layer's blob).  Same arguments, statistics blocks and output as vt_conv3d_gcr_f16x3, equal to it to f32 rounding of the merged */
/* weights.  Covers what vt_conv3d_up_covers reports: a shape of the specialised-wave kernel with C1, C2 multiples of 16.              */
size_t vt_conv3d_up_packed_floats(int Cout, int C2);
int vt_conv3d_pack_f16x3_up(const float *w, int Cout, int Cin, int C1, float *packed, void *stream);
int vt_conv3d_up_covers(int C1, int C2, int B, int D, int H, int W, int Cout);
int vt_conv3d_gcr_f16x3_up(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                           const float *scale_shift, const float *packed_w_f16x3, const float *packed_up, int Cout, int relu, float *out,
                           float *out_part, void *stream);
/* The last 'gcr' layer of the UNet3D together with final_conv (unet3d.py:470-474, a 1x1x1 conv 32 -> 32): relu(conv) stays in      */
/* registers, is split into half pairs and multiplied by the packed final weight in the epilogue -- no intermediate tensor, no     */
/* second launch.  Cout must be 32 and the shape one the specialised-wave kernel covers (VT_ERR_UNSUPPORTED otherwise).            */
int vt_conv3d_final_fusable(int B, int D, int H, int W, int Cin, int Cout);   /* 1 where _final covers the layer */
int vt_conv1x1_pack_f16x3(const float *w, int Cout, int Cin, float *packed /* 1024 floats */, void *stream);
int vt_conv3d_gcr_f16x3_final(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                              const float *scale_shift, const float *packed_w_f16x3, int Cout,
                              const float *final_packed_f16x3, const float *final_b, float *out, void *stream);
/* the same launch, and relu(conv(...)) itself to `y_keep` [B,D,H,W,32]: the training forward keeps the last layer's output for its     */
/* backward and for vt_conv1x1_bwd_masked (reference: the same two modules under autograd, src/conv_onet/training.py:757-894)            */
int vt_conv3d_gcr_f16x3_final_keep(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                                   const float *scale_shift, const float *packed_w_f16x3, int Cout,
                                   const float *final_packed_f16x3, const float *final_b, float *out, float *y_keep, void *stream);
size_t vt_unet3d_workspace_bytes(int B, int R, const vt_unet3d_params *params_host);
int vt_unet3d_fwd(const float *x_cl, int B, int R, const vt_unet3d_params *params_host,
                  void *workspace, size_t workspace_bytes, float *out, void *stream);
/* the same with the input's GroupNorm partial sums supplied by its producer (in_part [B][in_nblk][C][2], e.g. from                 */
/* vt_pointnet_mlp_fused): no statistics pass over the input grid.                                                              */
int vt_unet3d_fwd_stats(const float *x_cl, const float *in_part, int in_nblk, int B, int R, const vt_unet3d_params *params_host,
                        void *workspace, size_t workspace_bytes, float *out, void *stream);
/* vt_unet3d_fwd / vt_unet3d_fwd_stats (in_part may be NULL: statistics pass over x) with the first layer's empty blocks skipped:   */
/* tile_flags [B][(R/8)^3] from vt_voxel_tile_flags on the cloud that x was scattered from (vt_conv3d_gcr_f16x3_skip).  Where the    */
/* first DoubleConv is 32 -> 32 -> 32 channels with 8 groups (unet3d.py:96-127, the shipped f_maps) the SECOND layer skips the       */
/* blocks whose 12^3 halo is empty (flag bit 1) as well: the first layer's output is a constant per border class around them, so    */
/* the second's is one per class of a two-voxel rim (125 classes) -- sums over taps of W2 gamma2 relu(K1), W2 gamma2 and W2 beta2     */
/* that the first launch leaves per GroupNorm group, combined with the second GroupNorm's statistics by the second launch.          */
/* vt_unet3d_skip_layers: how many layers of this network take the flags at this batch and resolution (0, 1 or 2).                 */
int vt_unet3d_skip_layers(int B, int R, const vt_unet3d_params *params_host);
int vt_unet3d_fwd_skip(const float *x_cl, const float *in_part, int in_nblk, const unsigned char *tile_flags, int B, int R,
                       const vt_unet3d_params *params_host, void *workspace, size_t workspace_bytes, float *out, void *stream);
int vt_maxpool3d_cl(const float *x, int B, int D, int H, int W, int C, float *out, void *stream);
/* the same max-pool and, from the same pass, the pooled tensor's GroupNorm partial sums as vt_channel_stats would leave them   */
/* (bit-identical: same blocks, same order): part [B][nblk][C][2].                                                          */
int vt_maxpool3d_cl_stats(const float *x, int B, int D, int H, int W, int C, float *out, int nblk, float *part, void *stream);
int vt_conv1x1_cl(const float *x, int64_t V, int Cin, const float *w, const float *bias, int Cout, float *out, void *stream);
int vt_voxel_scatter_mean_cl_fwd(const float *feat, const int *idx, const int *order,
                                 const int *seg_lo, const int *seg_hi,
                                 int B, int T, int C, int R, float *grid_cl, void *stream);
int vt_voxel_scatter_mean_cl_bwd(const float *grad_grid_cl, const int *idx, const int *seg_lo, const int *seg_hi,
                                 int B, int T, int C, int R, float *grad_feat, void *stream);


/* ------------------------------------------------------------------------- */
/* UNet3D backward (training), channels-last.                                   */
/* Replaces: PyTorch autograd of the 'gcr' SingleConv, MaxPool3d and the        */
/*   upsample+concat of src/encoder/unet3d.py:20-72, 219-238, 283-293, run by    */
/*   loss.backward() (src/conv_onet/training.py:79,89,96).  For one block         */
/*   y = relu(conv(xn)), xn = GroupNorm([skip | upsample(low)]) and a gradient dy: */
/*   vt_relu_mask      g = dy where y > 0, else 0;                                 */
/*   (data gradient)   dxn = vt_conv3d_gcr[_bf16x3](g, pack(W^T flipped), no norm, */
/*                     no relu): the forward kernels on repacked weights;           */
/*   vt_conv3d_wgrad   dW[Cout][Cin][3][3][3] = sum_v g[v] (x) xn[v+tap] (xn is      */
/*                     re-normalised from skip/low with the forward scale_shift);     */
/*   vt_gn_bwd         GroupNorm backward from dxn: dskip [B,D,H,W,C1], dlow           */
/*                     [B,D/2,H/2,W/2,C2] (children summed; either may be NULL),        */
/*                     dgb[B][C][2] = per-scene (dgamma, dbeta); part1/part2 are the     */
/*                     FORWARD statistics of skip/low; bpart [B][nblkb][C][2] and coef    */
/*                     [B][C][3] are scratch;                                              */
/*   vt_maxpool3d_cl_bwd  routes dy to the first maximum of each 2x2x2 window.             */
/* ------------------------------------------------------------------------- */
int vt_relu_mask(const float *dy, const float *y, float *g, int64_t n, void *stream);
/* the same, and *absmax = max |g| (device scalar; feeds the power-of-two rescale of the split-half data / weight gradient kernels) */
int vt_relu_mask_absmax(const float *dy, const float *y, float *g, int64_t n, float *absmax, void *stream);
/* Backward of the UNet3D's final 1x1x1 conv (32 -> 32 channels; reference: src/encoder/unet3d.py final_conv under autograd,           */
/* src/conv_onet/training.py:757-894) fused with the ReLU mask of the layer in front of it: dout, y [n,32] (y = that layer's ReLU        */
/* output = the conv's input), w [32,32] (out, in).  g[v] = (y[v] > 0) * (dout[v] W), *absmax = max |g| (device scalar, or NULL),         */
/* dw [32,32] = dout^T y, db [32] = column sums of dout (either may be NULL).  Exact f32 (v_mfma_f32_32x32x2_f32), fixed summation        */
/* order.  One pass over dout and y instead of the framework's GEMM + vt_relu_mask_absmax + batched GEMM + two reductions.               */
size_t vt_conv1x1_bwd_workspace_bytes(void);
int vt_conv1x1_bwd_masked(const float *dout, const float *y, const float *w, int64_t n, float *g, float *absmax, float *dw, float *db,
                          void *workspace, size_t workspace_bytes, void *stream);
size_t vt_conv3d_wgrad_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_wgrad(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                    const float *scale_shift, const float *g, int Cout, void *workspace, size_t workspace_bytes,
                    float *dw, void *stream);
/* vt_conv3d_wgrad on the f16 matrix core with split (hi + lo IEEE-half) operands, f32 accumulation: the same dW to f32     */
/* rounding level (<= 1e-5 relative on the block tests), ~3x faster at 64^3 (MFMA rate 16x the f32 core's, three products).     */
/* `g_absmax`: device scalar max |g| (or NULL) -- g is scaled by the power of two that brings it to ~2^10 before the split        */
/* (output gradients sit far below the half range) and dW scaled back, both exactly.  Sides: D even, H and W multiples of 8        */
/* (workspace_bytes returns 0 otherwise: use vt_conv3d_wgrad).  Chunk-ordered reduction: bit-reproducible.                          */
size_t vt_conv3d_wgrad_f16x3_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_wgrad_f16x3(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                          const float *scale_shift, const float *g, int Cout, const float *g_absmax,
                          void *workspace, size_t workspace_bytes, float *dw, void *stream);
/* The same dW for a decoder-entry layer [skip | nearest-upsample(low)] (reference: src/encoder/unet3d.py:283-293 under autograd):     */
/* the skip channels as vt_conv3d_wgrad_f16x3; the upsampled channels per output parity class -- tap t of voxel 2 u + p reads the     */
/* low-resolution voxel u + ((p + t) >> 1), so each class is a 2 x 2 x 2-tap weight gradient over the low-resolution grid (64          */
/* products of N / 8 voxels instead of 27 of N), summed into the 27 taps by the reduction in a fixed order.  Sides in multiples of     */
/* 16 (D: 4), channels of 32 (workspace_bytes returns 0 otherwise: use vt_conv3d_wgrad_f16x3).                                          */
size_t vt_conv3d_wgrad_f16x3_up_workspace_bytes(int B, int D, int H, int W, int C1, int C2, int Cout);
int vt_conv3d_wgrad_f16x3_up(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                             const float *scale_shift, const float *g, int Cout, const float *g_absmax,
                             void *workspace, size_t workspace_bytes, float *dw, void *stream);
/* The same dW for a layer whose input x [B,D,H,W,C] is EXACTLY zero over most of the volume -- the UNet3D's first layer on a scene's */
/* mean grid (reference: src/encoder/pointnet.py:102-114 leaves >= 98.8 % of the grid zero; its gradient, src/conv_onet/training.py   */
/* :757-894).  xn = x * scale + shift inside the volume, 0 outside, so dW = sum_v g[v] (x * scale)[v + tap] + shift * (sum of g over   */
/* the voxels whose tap neighbour is inside): the first sum runs over the 8 x 8 x 2 tiles of the blocks `tile_flags` does not mark     */
/* (vt_voxel_build_clear_flags: bit 0 = no point in the block's 10^3 halo) only, in ascending tile order (bit-reproducible); the       */
/* second is 27 box sums of g per scene times the shift, added by the reduction.  Sides in multiples of 8, channels of 32.            */
size_t vt_conv3d_wgrad_f16x3_sparse_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_wgrad_f16x3_sparse(const float *x, int C, int B, int D, int H, int W, const float *scale_shift,
                                 const unsigned char *tile_flags, const float *g, int Cout, const float *g_absmax,
                                 void *workspace, size_t workspace_bytes, float *dw, void *stream);
int vt_gn_bwd(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
              const float *part1, int nblk1, const float *part2, int nblk2,
              const float *dxn, int groups, const float *gamma, double eps,
              float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow, void *stream);
/* The data-gradient conv of a plain 'gcr' layer (autograd of unet3d.py:20-72: the forward kernels on the weight with channels      */
/* swapped and taps flipped, vt_conv3d_pack_f16x3_t) that also leaves the GroupNorm backward's two sums: out_part                  */
/* [B][vt_conv3d_xstats_blocks][Cout][2] = per workgroup (sum dxn, sum dxn * stat_x) with stat_x [B][D][H][W][Cout] the layer's     */
/* input -- what vt_gn_bwd's statistics pass over dxn and x computes, from the conv's epilogue.  vt_gn_bwd_from_part is            */
/* vt_gn_bwd_masked on such sums (bpart, nblkb = the blocks count) without that pass.  vt_conv3d_xstats_blocks: 0 where the shape    */
/* is not on the specialised-wave kernel (use vt_conv3d_gcr_f16x3_scaled + vt_gn_bwd).                                              */
int vt_conv3d_xstats_blocks(int B, int D, int H, int W, int Cin, int Cout);
int vt_conv3d_gcr_f16x3_xstats(const float *g, int C, int B, int D, int H, int W, const float *packed_w_f16x3, int Cout,
                               const float *in_absmax, const float *stat_x, float *out, float *out_part, void *stream);
int vt_gn_bwd_from_part(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                        const float *part1, int nblk1, const float *part2, int nblk2,
                        const float *dxn, int groups, const float *gamma, double eps,
                        const float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow,
                        int mask_flags, float *absmax_skip, float *absmax_low, float *dgb_sum, void *stream);
/* vt_gn_bwd that also does the relu_mask pass of the layer(s) in front (autograd of unet3d.py:20-72: ReLU behind the conv whose   */
/* output this GroupNorm reads).  mask_flags bit 0: `skip` is such a ReLU output and dskip is its only gradient -- dskip comes out  */
/* as (skip > 0 ? dskip : 0) with max |dskip| in the device scalar absmax_skip, exactly what vt_relu_mask_absmax(dskip, skip)      */
/* would leave; bit 1: the same for `low` / dlow / absmax_low.  The layer in front then skips its vt_relu_mask_absmax.             */
/* dgb_sum (or NULL; needs dskip or dlow): [2][C] = (dgamma, dbeta) summed over the scenes in scene order (dgb keeps them per scene). */
int vt_gn_bwd_masked(const float *skip, int C1, const float *low, int C2, int B, int D, int H, int W,
                     const float *part1, int nblk1, const float *part2, int nblk2,
                     const float *dxn, int groups, const float *gamma, double eps,
                     float *bpart, int nblkb, float *coef, float *dgb, float *dskip, float *dlow,
                     int mask_flags, float *absmax_skip, float *absmax_low, float *dgb_sum, void *stream);
int vt_maxpool3d_cl_bwd(const float *x, const float *dy, int B, int D, int H, int W, int C, float *dx, void *stream);
/* The gradient of an encoder level's output y (unet3d.py:449-474: it feeds the next level's max-pool AND the decoder's skip) in one  */
/* pass: g = (y > 0 ? dskip + maxpool_backward(dpooled) : 0) -- vt_maxpool3d_cl_bwd, the framework's add of the two gradients and    */
/* the vt_relu_mask_absmax of the layer that produced y -- with max |g| in the device scalar absmax (or NULL).                        */
int vt_maxpool3d_cl_bwd_fork(const float *y, const float *dskip, const float *dpooled, int B, int D, int H, int W, int C, float *g,
                             float *absmax, void *stream);

/* ------------------------------------------------------------------------- */
/* Hand branch (SURVEY.md section 8f "next" row 3).                              */
/* Plane-mode PointNet: replaces normalize_coordinate + coordinate2index('2d')   */
/*   (src/common.py:268-291, 333-345; call sites src/encoder/pointnet.py:141-149) */
/*   and the scatter_mean of generate_plane_features (pointnet.py:85-95).         */
/*   vt_plane_build: plane 0 = 'xz', 1 = 'xy', 2 = 'yz'; idx[b,t] = i(first axis)  */
/*   + R * i(second axis) with the plane constants (divisor 1 + padding + 10e-6,    */
/*   upper clamp 1 - 10e-6); order/seg_lo/seg_hi as vt_voxel_build, so the          */
/*   vt_voxel_pool_max_* kernels serve every plane (pool_local, pointnet.py:116-132, */
/*   sums the per-plane results).  vt_plane_scatter_mean_*: plane [B,C,R,R].          */
/* MANO layer: replaces ManoLayer.forward (src/encoder/manolayer.py:160-364) for the  */
/*   shipped configuration (axis-angle root + joints, use_pca False, hands_mean added, */
/*   model betas, no translation; right hand, or left with vt_mano_pack_side).  vt_mano_pack lays the model out once: */
/*   v_template [778,3], shapedirs [778,3,10] + betas [10] (both may be NULL: betas 0), */
/*   posedirs [778,3,135], j_regressor [16,778] dense, weights [778,16], hands_mean [45] */
/*   -> blob[VT_MANO_BLOB_FLOATS].  vt_mano_fwd: pose [B,48] = root axis-angle + 45 joint */
/*   angles -> verts [B,778,3], joints [B,21,3], both minus joint center_idx (-1: none).   */
/* ------------------------------------------------------------------------- */
#define VT_MANO_BLOB_FLOATS 330240
int vt_plane_build(const float *pts, int B, int T, int R, double padding, int plane,
                   int *idx, int *order, int *seg_lo, int *seg_hi, void *stream);
/* The planes of a scene side by side in ONE launch: `planes` = n_planes ids (0 xz, 1 xy, 2 yz); idx / order / seg_lo / seg_hi are       */
/* [n_planes][B][T], slice k = what vt_plane_build(plane = planes[k]) writes (the hand encoder builds all three per forward).            */
int vt_plane_build_multi(const float *pts, int B, int T, int R, double padding, int n_planes, const int *planes,
                         int *idx, int *order, int *seg_lo, int *seg_hi, void *stream);
/* generate_plane_features (pointnet.py:85-95) for those planes in one launch each way: planes / grad_planes [n_planes][B][C][R*R] (the      */
/* layout of torch.cat over the planes' [B,C,R,R] tensors: what the hand encoder hands its U-Net), index arrays as above; the backward sums */
/* the planes' shares per point in plane order.                                                                                              */
int vt_plane_scatter_mean_multi_fwd(const float *feat, int n_planes, const int *idx, const int *order, const int *seg_lo, const int *seg_hi,
                                    int B, int T, int C, int R, float *planes, void *stream);
int vt_plane_scatter_mean_multi_bwd(const float *grad_planes, int n_planes, const int *idx, const int *seg_lo, const int *seg_hi,
                                    int B, int T, int C, int R, float *grad_feat, void *stream);
int vt_plane_scatter_mean_fwd(const float *feat, const int *idx, const int *order,
                              const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *plane, void *stream);
int vt_plane_scatter_mean_bwd(const float *grad_plane, const int *idx, const int *seg_lo, const int *seg_hi,
                              int B, int T, int C, int R, float *grad_feat, void *stream);
int vt_mano_pack(const float *v_template, const float *shapedirs, const float *betas, const float *posedirs,
                 const float *j_regressor, const float *weights, const float *hands_mean, float *blob, void *stream);
/* the same with the hand's side: left != 0 marks a MANO_LEFT model, whose middle-finger tip is vertex 445 instead of 444            */
/* (manolayer.py:327-330); the side travels in the blob, vt_mano_fwd / vt_mano_bwd read it there                                      */
int vt_mano_pack_side(const float *v_template, const float *shapedirs, const float *betas, const float *posedirs,
                      const float *j_regressor, const float *weights, const float *hands_mean, int left, float *blob, void *stream);
int vt_mano_fwd(const float *pose, int B, const float *blob, int center_idx, float *verts, float *joints, void *stream);
/* Backward of vt_mano_fwd (PyTorch autograd through manolayer.py:186-347 under loss_mano / loss_pc, training.py:59-60):   */
/* dpose [B,48] from dverts [B,778,3] and djoints [B,21,3]; one workgroup per hand, the forward's intermediates recomputed,  */
/* every sum in a fixed order (bit-reproducible).                                                                          */
int vt_mano_bwd(const float *pose, int B, const float *blob, int center_idx, const float *dverts, const float *djoints,
                float *dpose, void *stream);

/* ------------------------------------------------------------------------- */
/* The hand encoder's 2-D U-Net over its feature planes.    */
/* Replaces: UNet.forward (src/encoder/unet.py:220-233: DownConv :52-79,         */
/*   UpConv :82-120, conv_final) as LocalPoolPointnet builds and calls it         */
/*   (src/encoder/pointnet.py:49-50, 96-99): UNet(c_dim, in_channels=c_dim,        */
/*   depth, start_filts, merge_mode 'concat', up_mode 'transpose') on the planes   */
/*   [B, c_dim, R, R] of a scene -- here on n_img images at once (the net has no   */
/*   cross-image operation): x [n_img][in_channels][H][W] -> out [n_img][classes]  */
/*   [H][W], both NCHW like the reference's tensors.                               */
/* One launch per layer of ONE kernel template (implicit GEMM on the f32 matrix     */
/*   core: a workgroup per 32 pixels x 32 output channels, K split over its waves); */
/*   bias + ReLU in the epilogue, the 2x2 max-pool and the skip concat in the next  */
/*   layer's loader (no pass of their own).  Weights in nn.Conv2d /                 */
/*   nn.ConvTranspose2d layout; vt_plane_unet_pack lays them out in fragment order */
/*   (blob of vt_plane_unet_blob_bytes; repack after every weight update).         */
/* Covered (vt_plane_unet_supported): depth 2..5; in_channels, start_filts and     */
/*   num_classes multiples of 32; H, W powers of two with at least 4 x 4 pixels at */
/*   the bottom level.  After the call the workspace holds every layer's           */
/*   channels-last activations: vt_plane_unet_bwd reads them (the training forward */
/*   is this same call).                                                           */
/* ------------------------------------------------------------------------- */
#define VT_PLANE_UNET_MAX_DEPTH 5
typedef struct vt_plane_unet_params {
    int32_t depth, in_channels, start_filts, num_classes;
    const float *down_w[VT_PLANE_UNET_MAX_DEPTH][2];   /* down_convs.{l}.conv{1,2}.weight [Cout][Cin][3][3] */
    const float *down_b[VT_PLANE_UNET_MAX_DEPTH][2];
    const float *up_tw[VT_PLANE_UNET_MAX_DEPTH];        /* up_convs.{u}.upconv.weight [Cin][Cout][2][2]      */
    const float *up_tb[VT_PLANE_UNET_MAX_DEPTH];
    const float *up_w[VT_PLANE_UNET_MAX_DEPTH][2];      /* up_convs.{u}.conv{1,2}.weight                     */
    const float *up_b[VT_PLANE_UNET_MAX_DEPTH][2];
    const float *final_w, *final_b;                     /* conv_final [classes][start_filts][1][1]           */
} vt_plane_unet_params;
int vt_plane_unet_supported(int depth, int in_channels, int start_filts, int num_classes, int H, int W);
size_t vt_plane_unet_blob_bytes(int depth, int in_channels, int start_filts, int num_classes);
size_t vt_plane_unet_workspace_bytes(int depth, int in_channels, int start_filts, int num_classes, int n_img, int H, int W);
int vt_plane_unet_pack(const vt_plane_unet_params *params_host, float *blob, size_t blob_bytes, void *stream);
int vt_plane_unet_fwd(const float *x, int n_img, int H, int W, const vt_plane_unet_params *dims_host, const float *blob,
                      void *workspace, size_t workspace_bytes, float *out, void *stream);
/* Backward of vt_plane_unet_fwd (PyTorch autograd through UNet.forward under the hand encoder's losses, loss_mano / loss_pc,   */
/* src/conv_onet/training.py:59-60, 79-96): from dout [n_img][classes][H][W], the input x, the packed weights and the forward's   */
/* workspace (its activations) -> dx [n_img][in_channels][H][W] and the gradient of every weight and bias in the parameter's own   */
/* layout (vt_plane_unet_grads: the pointers of vt_plane_unet_params, written, not accumulated).  Per layer, in reverse: one       */
/* data-gradient launch of the forward's kernel template (mirrored taps, transposed weights: the second half of the blob) whose     */
/* loader assembles the layer's output gradient from what its consumers left -- the skip concat's and the pooled path's sum, the     */
/* max-pool's routing to the window's first maximum, the ReLU mask -- and one weight-gradient launch (pixels as K, partial sums per   */
/* pixel slice); one finalize launch sums the slices.  Every sum in a fixed order: bit-reproducible.                                   */
typedef struct vt_plane_unet_grads {
    float *down_w[VT_PLANE_UNET_MAX_DEPTH][2], *down_b[VT_PLANE_UNET_MAX_DEPTH][2];
    float *up_tw[VT_PLANE_UNET_MAX_DEPTH], *up_tb[VT_PLANE_UNET_MAX_DEPTH];
    float *up_w[VT_PLANE_UNET_MAX_DEPTH][2], *up_b[VT_PLANE_UNET_MAX_DEPTH][2];
    float *final_w, *final_b;
} vt_plane_unet_grads;
size_t vt_plane_unet_bwd_workspace_bytes(int depth, int in_channels, int start_filts, int num_classes, int n_img, int H, int W);
int vt_plane_unet_bwd(const float *x, int n_img, int H, int W, const vt_plane_unet_params *dims_host, const float *blob,
                      const void *fwd_workspace, const float *dout, void *workspace, size_t workspace_bytes,
                      const vt_plane_unet_grads *grads_host, float *dx, void *stream);

/* ------------------------------------------------------------------------- */
/* PointNet per-point MLP (inference).  Replaces the nn.Linear / ResnetBlockFC   */
/* calls of LocalPoolPointnet.forward (src/encoder/pointnet.py:154-162;           */
/* src/layers.py:8-50): rows are points, weights in nn.Linear layout [out][in].   */
/*   vt_linear_rows  out[n] = b + W x[n]            (fc_pos, fc_c; b may be NULL)   */
/*   vt_resblock_fc  x = [x1[n] | x2[n]] (x2 may be NULL: no concat);                */
/*                   h = b0 + W0 relu(x); out[n] = b1 + W1 relu(h) + Ws x           */
/*                   (ws NULL: identity shortcut, needs C1 + C2 == O).               */
/* At most 256 hidden / output channels; weights must fit 64 KiB of LDS.             */
/* ------------------------------------------------------------------------- */
int vt_linear_rows(const float *x, const float *w, const float *b, int64_t N, int Cin, int Cout, float *out, void *stream);
int vt_resblock_fc(const float *x1, int C1, const float *x2, int C2, int64_t N,
                   const float *w0, const float *b0, const float *w1, const float *b1, const float *ws,
                   int H, int O, float *out, void *stream);
/* The whole per-point MLP of LocalPoolPointnet.forward (pointnet.py:154-162) in one launch for inference with ONE voxel index:     */
/* fc_pos -> block 0 -> 4 x (pool_local, concat, block) -> fc_c.  A workgroup owns every cell whose first sorted point falls into  */
/* its window of 8-16 sorted positions -- complete cells of any size -- so pooling needs no grid-wide step; same arithmetic and        */
/* summation order as vt_linear_rows / vt_resblock_fc / vt_voxel_pool_max_fwd: bit-identical features.  block_w: 25 device pointers, */
/* per block fc_0.weight [32][64], fc_0.bias, fc_1.weight [32][32], fc_1.bias, shortcut.weight [32][64]; hidden must be 32, c_dim     */
/* <= 64 (VT_ERR_UNSUPPORTED otherwise: use the per-layer kernels); scratch [B,T,32] floats; out [B,T,c_dim] by point (or NULL).     */
/* With grid_cl != NULL the kernel is also generate_grid_features (pointnet.py:102-110, scatter_mean): every cell's mean feature,     */
/* summed in ascending point order like vt_voxel_scatter_mean_cl_fwd, goes into the ZERO-FILLED channels-last grid [B,R,R,R,c_dim]   */
/* (idx [B,T] = the points' cell ids) and grid_part [B][vt_pointnet_mlp_stat_blocks(B,T)][c_dim][2] receives the GroupNorm partial    */
/* sums of that grid (the empty voxels contribute nothing): what vt_channel_stats would compute in a pass over the whole grid.        */
int vt_pointnet_mlp_stat_blocks(int B, int T);
int vt_pointnet_mlp_fused(const float *pts, int B, int T, const int *order, const int *seg_lo, const int *seg_hi,
                          const float *pos_w, const float *pos_b, const float *const *block_w, int hidden,
                          const float *c_w, const float *c_b, int c_dim, float *scratch, float *out,
                          const int *idx, int R, float *grid_cl, float *grid_part, void *stream);
/* Backward of the two (training; PyTorch autograd of layers.py:8-50 and of the nn.Linear calls at    */
/* pointnet.py:154-162 under loss.backward(), training.py:79,89,96):                                    */
/*   vt_resblock_fc_bwd  d out [N][O] -> d x1 [N][C1], d x2 [N][C2] (NULL: not wanted), and the two       */
/*                       [N][H] tensors the weight gradients contract over: act = relu(h) (recomputed)   */
/*                       and dh = (W1^T d out) . [h > 0];                                                 */
/*   vt_rows_wgrad       dW [M][K] = sum_n G[n][m] X[n][k], db [M] = sum_n G[n][m] (db may be NULL) with    */
/*                       X = [x1 | x2] (x2 may be NULL), relu'd when relu_x: f32 MFMA outer products over  */
/*                       1024-point chunks, partials summed in chunk order (bit-reproducible).             */
/*                       fc_1: (d out, act); fc_0: (dh, relu [x1|x2]); shortcut: (d out, [x1|x2]);          */
/*                       a plain linear layer: (d out, x).  Its data gradient is vt_linear_rows on W^T.     */
int vt_resblock_fc_bwd(const float *x1, int C1, const float *x2, int C2, int64_t N,
                       const float *w0, const float *b0, const float *w1, const float *ws, int H, int O,
                       const float *dout, float *dx1, float *dx2, float *act, float *dh, void *stream);
size_t vt_rows_wgrad_workspace_bytes(int64_t N, int M, int K);
int vt_rows_wgrad(const float *G, int M, const float *x1, int C1, const float *x2, int C2, int relu_x, int64_t N,
                  void *workspace, size_t workspace_bytes, float *dW, float *db, void *stream);
/* The three weight gradients of a ResnetBlockFC (layers.py:8-50: fc_1 from (dout, relu(h)), fc_0 from (dh, relu(x)), the shortcut  */
/* from (dout, x); x = [x1 | x2]) in one pair of launches instead of three: the same tiles, partial sums and chunk-ordered          */
/* reduction as vt_rows_wgrad (bit for bit).  act / dh: vt_resblock_fc_bwd's outputs; dws NULL without a shortcut layer.           */
size_t vt_resblock_wgrad_workspace_bytes(int64_t N, int C, int H, int O, int has_shortcut);
int vt_resblock_wgrad(const float *x1, int C1, const float *x2, int C2, int64_t N, const float *act, const float *dh, const float *dout,
                      int H, int O, void *workspace, size_t workspace_bytes,
                      float *dw0, float *db0, float *dw1, float *db1, float *dws, void *stream);

/* ------------------------------------------------------------------------- */
/* Generalized winding number of query points against a triangle mesh.          */
/* Stands where the VTacO (t2d) training step calls                              */
/*   igl.fast_winding_number_for_meshes(V, F, Q) (src/conv_onet/training.py:723,  */
/*   862) to label its re-sampled query points.  libigl is an un-vendored,        */
/*   absent dependency and its routine is a hierarchical approximation: this is    */
/*   the exact sum it approximates, w(q) = 1/(4 pi) sum_f Omega_f(q) (float64        */
/*   solid angles): 1 inside a closed outward-oriented mesh, 0 outside.               */
/*   verts [V,3] f32, faces [F,3] i32, pts [N,3] f32 -> out [N] f32.                    */
/* ------------------------------------------------------------------------- */
int vt_winding_number(const float *verts, int V, const int32_t *faces, int F, const float *pts, int64_t N, float *out, void *stream);
/* The same for a batch of scenes in ONE launch (the reference loops over the batch, training.py:723, 862): scenes = device array of  */
/* B records {const float *verts; const int32_t *faces; int32 V; int32 F} (24 bytes each, the meshes anywhere in device memory), */
/* pts [B][N][3], out [B][N].                                                                                                     */
int vt_winding_number_scenes(const void *scenes, int B, const float *pts, int64_t N, float *out, void *stream);

/* Contact clouds of the tactile sensors from their depth images (SURVEY.md section 8f "next" row 2: the training side).          */
/* Replaces: the per-scene, per-sensor numpy passes of src/conv_onet/training.py:817-853 and generation.py:224-244               */
/*   (np.where(abs(depth - depth_origin) > 1e-4), depth_2_camera_pointcloud, np.random.randint, pc_cam_to_world, norm_pc_1).        */
/* vt_contact_scan: depth [n_images][n_pixels] f32, depth_origin [n_pixels] f64 (the sensor's flat reading), touch_success          */
/*   [n_images] u8 or NULL -> index [n_images][n_pixels] i32 (the touched pixels in ascending order = np.where's) and count         */
/*   [n_images]; the comparison in float64 as numpy's.  The host reads the counts, draws np.random.randint(count, size=128) where a  */
/*   count exceeds 128 (the draws stay on the host: a seeded run consumes numpy's generator as the reference does) and inverts the   */
/*   4 x 4 camera poses.                                                                                                             */
/* vt_contact_points: per image kept[i] points (all `count` in order when sel is NULL, else index[sel[i][j]]), pose [n_images][16]  */
/*   f64 = {inverse pose's 3x3 row-major, translation 3, cloud centroid 3, cloud scale 1}: pinhole unprojection (f = height /        */
/*   (2 tan(fov / 2)), principal point at the centre, axes (z, -x, -y)), world = M cam + t, (world - centroid) / scale in float64,    */
/*   written as float32 rows row0[i] .. row0[i] + kept[i] of scene i / 5 in p_sample [B][S][3] (finger [B][S] i64 gets i % 5; may be  */
/*   NULL).  n_images = 5 B.                                                                                                          */
int vt_contact_scan(const float *depth, const double *depth_origin, const unsigned char *touch_success, int n_images, int n_pixels,
                    double threshold, int *index, int *count, void *stream);
int vt_contact_points(const float *depth, const int *index, const int *sel, const int *kept, const int *row0, const double *pose,
                      int n_images, int n_pixels, int width, int height, double fov_deg, int max_points, int S,
                      float *p_sample, long long *finger, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VTACO_HIP_H */
