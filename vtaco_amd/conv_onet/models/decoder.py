"""Implicit occupancy decoders (drop-in for reference src/conv_onet/models/decoder.py).

Same constructor kwargs, parameter names and call signatures as the reference's
``LocalDecoder`` (decoder.py:9-161); the arithmetic -- coordinate normalisation,
trilinear sampling of the feature grid, fc_p, 5 x (fc_c + ResnetBlockFC), fc_out --
is ONE fused HIP kernel reached through the C ABI (``vt_decode_fwd``).  There is
no torch fallback: CPU tensors raise.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from ... import ops
from ..._lib import VtError
from ...layers import ResnetBlockFC
from ...transformer_fusion import TransformerFusion


class _DecodeFn(torch.autograd.Function):
    """Differentiable fused decode: forward = vt_decode_fwd (saving activations), backward =
    vt_decode_bwd + vt_decode_wgrad.  Gradients flow to the feature grid, c_img and every
    decoder parameter; not to the query points (the reference never reads p.grad either,
    SURVEY.md section 8a row A14)."""

    @staticmethod
    def forward(ctx, dec, p, grid, c_img, *params):
        img = c_img is not None
        B, N = p.shape[0], p.shape[1]
        save = ops.decode_save_buffer(B * N, grid.device)
        out = ops.decode_fwd(grid, dec._blob(img=img), pts=p, c_img=c_img, padding=dec.padding, save=save)
        ctx.dec, ctx.img, ctx.save, ctx.grid_shape = dec, img, save, tuple(grid.shape)
        ctx.p, ctx.c_img = p.detach(), (c_img.detach() if img else None)
        ctx.need_grid = grid.requires_grad
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dec, img = ctx.dec, ctx.img
        ggrid, gimg, flat = ops.decode_bwd(ctx.grid_shape, dec._blob_t(img=img), grad_out, ctx.save, pts=ctx.p,
                                           with_c_img=img, c_img=ctx.c_img, padding=dec.padding,
                                           want_grid_grad=ctx.need_grid)
        g = ops.split_decoder_grads(flat, 3 + dec.c_dim if img else 3)
        grads = []
        for name in dec._param_order(img):
            key, idx = name
            grads.append(g[key] if idx is None else g[key][idx])
        return (None, None, ggrid, gimg, *grads)


class _DecodeContactFn(torch.autograd.Function):
    """forward_contact, differentiable (decoder.py:105-133; training.py:896-948): the fused decode with both heads
    (vt_decode_fwd with out2 and the saved activations), backward = vt_decode_bwd_contact + vt_decode_wgrad_contact."""

    @staticmethod
    def forward(ctx, dec, p, grid, *params):
        B, N = p.shape[0], p.shape[1]
        save = ops.decode_save_buffer(B * N, grid.device)
        out, out2 = ops.decode_fwd(grid, dec._blob(contact=True), pts=p, padding=dec.padding, save=save, want_contact=True)
        ctx.dec, ctx.save, ctx.grid_shape, ctx.p, ctx.need_grid = dec, save, tuple(grid.shape), p.detach(), grid.requires_grad
        return out, out2

    @staticmethod
    def backward(ctx, grad_out, grad_out2):
        dec = ctx.dec
        ggrid, _, flat = ops.decode_bwd(ctx.grid_shape, dec._blob_t(contact=True), grad_out, ctx.save, pts=ctx.p,
                                        padding=dec.padding, want_grid_grad=ctx.need_grid, grad_out2=grad_out2)
        g = ops.split_decoder_grads(flat, 3)
        grads = [g[key] if idx is None else g[key][idx] for key, idx in dec._param_order(False)]
        return (None, None, ggrid, *grads, g["fc_out_contact.weight"], g["fc_out_contact.bias"])


class _DecodeWideFn(torch.autograd.Function):
    """The decoder at the shapes beyond 32 / 32 (hidden_size / c_dim multiples of 32 up to 256, `leaky`, 'nearest'), differentiable:
    forward = vt_decode_fwd_wide_train (keeps every layer's input), backward = vt_decode_bwd_wide (data gradients on transposed
    weight fragments, the grid gradient by atomics) + vt_rows_wgrad per layer.  Gradients to the feature grid, c_img and every
    parameter; with ``contact`` both heads (forward_contact)."""

    @staticmethod
    def forward(ctx, dec, p, grid, c_img, contact, *params):
        img = c_img is not None
        nearest = dec.sample_mode == 'nearest'
        out, out2, save = ops.decode_fwd_wide_train(grid, dec._blob(img=img, contact=contact), p, c_img, dec.hidden_size, dec.n_blocks,
                                                    dec.leaky, nearest, dec.padding, want_contact=contact)
        ctx.dec, ctx.img, ctx.contact, ctx.save, ctx.grid_shape = dec, img, contact, save, tuple(grid.shape)
        ctx.p, ctx.c_img = p.detach(), (c_img.detach() if img else None)
        ctx.need_grid = grid.requires_grad
        ctx.need_img = img and c_img.requires_grad
        return (out, out2) if contact else out

    @staticmethod
    def backward(ctx, grad_out, grad_out2=None):
        dec, img = ctx.dec, ctx.img
        first = dec.fc_p_img if img else dec.fc_p
        blob_t = ops.pack_decoder_wide_t(first.weight, [l.weight for l in dec.fc_c], [(b.fc_0.weight, b.fc_1.weight) for b in dec.blocks],
                                         dec.fc_out.weight, dec.fc_out_contact.weight if ctx.contact else None)
        if ctx.contact and grad_out2 is None:
            grad_out2 = torch.zeros_like(grad_out)
        ggrid, gimg, g = ops.decode_bwd_wide(ctx.grid_shape, blob_t, grad_out, ctx.save, ctx.p, dec.hidden_size, dec.n_blocks, dec.leaky,
                                             dec.sample_mode == 'nearest', dec.padding, c_img=ctx.c_img, want_grid_grad=ctx.need_grid,
                                             grad_out2=grad_out2 if ctx.contact else None)
        grads = [g[key] if idx is None else g[key][idx] for key, idx in dec._param_order(img)]
        if ctx.contact:
            grads += [g["fc_out_contact.weight"], g["fc_out_contact.bias"]]
        return (None, None, ggrid, gimg if ctx.need_img else None, None, *grads)


class _SampleGridFn(torch.autograd.Function):
    """Trilinear sampling alone (vt_sample_grid / vt_sample_grid_bwd), differentiable in the grid."""

    @staticmethod
    def forward(ctx, grid, p, padding):
        ctx.shape, ctx.p, ctx.padding = tuple(grid.shape), p.detach(), padding
        return ops.sample_grid(grid, p, padding)

    @staticmethod
    def backward(ctx, grad_feat):
        return ops.sample_grid_bwd(ctx.shape, ctx.p, grad_feat, ctx.padding), None, None


class _DecodeMlpFn(torch.autograd.Function):
    """The conditioned MLP on given features (vt_decode_mlp_fwd_train; backward vt_decode_mlp_bwd +
    vt_decode_wgrad): gradients to the features and to every decoder parameter (fc_p form)."""

    @staticmethod
    def forward(ctx, dec, p, c, *params):
        out, save = ops.decode_mlp_fwd_train(c, dec._blob(), p)
        ctx.dec, ctx.save, ctx.p = dec, save, p.detach()
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dec = ctx.dec
        grad_c, flat = ops.decode_mlp_bwd(dec._blob_t(img=False), grad_out, ctx.save, ctx.p, dec.c_dim)
        g = ops.split_decoder_grads(flat, 3)
        grads = []
        for key, idx in dec._param_order(False):
            grads.append(g[key] if idx is None else g[key][idx])
        return (None, None, grad_c, *grads)


class _DecodeMlpWideFn(torch.autograd.Function):
    """_DecodeMlpFn at the widths beyond 32 / 32 (vt_decode_mlp_fwd_wide_train; backward vt_decode_mlp_bwd_wide + vt_rows_wgrad per
    layer): gradients to the given features and to every decoder parameter (fc_p form)."""

    @staticmethod
    def forward(ctx, dec, p, c, *params):
        out, save = ops.decode_mlp_fwd_wide_train(c, dec._blob(), p, dec.hidden_size, dec.n_blocks, dec.leaky)
        ctx.dec, ctx.save, ctx.p, ctx.c = dec, save, p.detach(), c.detach()
        return out

    @staticmethod
    def backward(ctx, grad_out):
        dec = ctx.dec
        blob_t = ops.pack_decoder_wide_t(dec.fc_p.weight, [l.weight for l in dec.fc_c], [(b.fc_0.weight, b.fc_1.weight) for b in dec.blocks],
                                         dec.fc_out.weight, None)
        grad_c, g = ops.decode_mlp_bwd_wide(blob_t, grad_out, ctx.save, ctx.p, ctx.c, dec.hidden_size, dec.n_blocks, dec.leaky)
        grads = [g[key] if idx is None else g[key][idx] for key, idx in dec._param_order(False)]
        return (None, None, grad_c, *grads)


class LocalDecoder(nn.Module):
    """Decoder conditioned on a local 3-D feature grid.

    Args mirror the reference (decoder.py:23-24): dim, c_dim, hidden_size,
    n_blocks, leaky, sample_mode, padding, with_contact.
    """

    def __init__(self, dim=3, c_dim=128, hidden_size=256, n_blocks=5, leaky=False,
                 sample_mode='bilinear', padding=0.1, with_contact=False, **kwargs):
        super().__init__()
        if dim != 3:
            raise VtError("LocalDecoder: only dim=3 is built")
        if sample_mode not in ('bilinear', 'nearest'):
            raise VtError(f"LocalDecoder: sample_mode must be 'bilinear' or 'nearest' (F.grid_sample's modes for 5-D input), got {sample_mode!r}")
        self.c_dim, self.n_blocks, self.hidden_size = c_dim, n_blocks, hidden_size
        # `leaky`: leaky_relu(0.2) in front of the output heads (reference decoder.py:46-49, 157; the blocks stay ReLU).  The shipped
        # shape (32 / 32, relu) runs on the LDS-resident kernels of decode.hip, training included; every other shape -- hidden_size
        # and c_dim multiples of 32 up to 256, e.g. the class defaults 256 / 128 -- on the weight-streaming kernels of
        # decode_wide.hip: vt_decode_fwd_wide (exact f32) or vt_decode_fwd_wide_f16x3 (split-f16 operands, for the half-precision
        # settings of ``precision``), and under autograd vt_decode_fwd_wide_train / vt_decode_bwd_wide / vt_rows_wgrad (_DecodeWideFn)
        self.leaky = bool(leaky)
        self._wide = self.leaky or hidden_size != 32 or c_dim != 32 or sample_mode == 'nearest'
        self.sample_mode, self.padding = sample_mode, padding
        self.fc_c = nn.ModuleList(nn.Linear(c_dim, hidden_size) for _ in range(n_blocks))
        self.fc_p = nn.Linear(dim, hidden_size)
        self.fc_p_img = nn.Linear(dim + c_dim, hidden_size)
        self.blocks = nn.ModuleList(ResnetBlockFC(hidden_size) for _ in range(n_blocks))
        self.fc_out = nn.Linear(hidden_size, 1)
        if with_contact:
            self.fc_out_contact = nn.Linear(hidden_size, 1)
        self._blobs = {}
        # arithmetic of the 16 dense layers on the no-grad paths: "f32" (exact-f32 MFMA), "bf16x3" / "f16x3" (split 16-bit MFMA
        # operands: ~2e-5 / ~1e-6 abs on O(1) logits) or "f16f8" (f16 products + fp8 corrections, lattice slabs only: ~4e-5, the
        # fastest; point queries then run as "f16x3"); training is always "f32"
        self.precision = os.environ.get("VTACO_DECODE_PRECISION", "f32")

    def _point_precision(self):
        return "f16x3" if self.precision == "f16f8" else self.precision

    # -- weights -> MFMA-fragment blob, cached until a parameter changes ---------
    def _blob(self, img=False, contact=False, precision="f32"):
        head2 = (self.fc_out_contact.weight, self.fc_out_contact.bias) if contact else None
        first = self.fc_p_img if img else self.fc_p
        params = [first.weight, first.bias, self.fc_out.weight, self.fc_out.bias]
        for lin, blk in zip(self.fc_c, self.blocks):
            params += [lin.weight, lin.bias, *blk.packed()]
        if head2:
            params += list(head2)
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        if self._wide:
            precision = self._wide_precision(precision)
        hit = self._blobs.get((img, contact, precision))
        if hit is not None and hit[0] == stamp:
            return hit[1]
        blob = ops.pack_decoder(first.weight, first.bias,
                                [(l.weight, l.bias) for l in self.fc_c],
                                [b.packed() for b in self.blocks],
                                (self.fc_out.weight, self.fc_out.bias), head2,
                                out=hit[1] if hit is not None else None, precision=precision)
        self._blobs[(img, contact, precision)] = (stamp, blob)
        return blob

    def _blob_t(self, img=False, contact=False):
        first = self.fc_p_img if img else self.fc_p
        out2 = (self.fc_out_contact.weight, self.fc_out_contact.bias) if contact else None
        return ops.pack_decoder(first.weight, first.bias, [(l.weight, l.bias) for l in self.fc_c],
                                [b.packed() for b in self.blocks], (self.fc_out.weight, self.fc_out.bias), out2,
                                transposed=True)

    def _param_order(self, img):
        """(gradient key, index) for every tensor passed to _DecodeFn, in order."""
        order = [("fc_p.weight", None), ("fc_p.bias", None)]
        for i in range(self.n_blocks):
            order += [("fc_c.weight", i), ("fc_c.bias", i), ("fc_0.weight", i), ("fc_0.bias", i),
                      ("fc_1.weight", i), ("fc_1.bias", i)]
        return order + [("fc_out.weight", None), ("fc_out.bias", None)]

    def _params(self, img):
        first = self.fc_p_img if img else self.fc_p
        ps = [first.weight, first.bias]
        for lin, blk in zip(self.fc_c, self.blocks):
            ps += [lin.weight, lin.bias, *blk.packed()]
        return ps + [self.fc_out.weight, self.fc_out.bias]

    def _wants_grad(self, grid, c_img=None):
        if not torch.is_grad_enabled():
            return False
        return grid.requires_grad or (c_img is not None and c_img.requires_grad) or any(
            p.requires_grad for p in self.parameters())

    def _wide_train(self, p, grid, c_img=None, contact=False):
        """The wide shapes under autograd: _DecodeWideFn (HIP forward and backward)."""
        if not grid.is_cuda:
            raise VtError(f"LocalDecoder: inputs must live on a HIP device (got {grid.device})")
        params = self._params(c_img is not None)
        if contact:
            params = params + [self.fc_out_contact.weight, self.fc_out_contact.bias]
        return _DecodeWideFn.apply(self, p.float(), grid, None if c_img is None else c_img.float(), contact, *params)

    @staticmethod
    def _wide_precision(precision):
        """The kernel of the shapes beyond 32 / 32 for a requested arithmetic: the split-f16 forward (vt_decode_fwd_wide_f16x3) for the
        half-precision forms, the exact-f32 one (vt_decode_fwd_wide) for "f32" and for "bf16x3" (the range guard's way out)."""
        return "wide_f16x3" if precision in ("f16x3", "f16f8", "wide_f16x3") else "wide"

    def _wide_fwd(self, grid, precision=None, **kw):
        wp = self._wide_precision(precision or self._point_precision())
        img = kw.get("c_img") is not None or kw.get("finger_ids") is not None
        return ops.decode_fwd(grid, self._blob(img=img, contact=kw.get("want_contact", False), precision=wp),
                              padding=self.padding, precision=wp,
                              wide=(self.hidden_size, self.n_blocks, self.leaky, self.sample_mode == 'nearest'), **kw)

    @staticmethod
    def _grid_of(c_plane):
        if set(c_plane.keys()) != {'grid'}:
            raise VtError("LocalDecoder: only the 'grid' feature volume is built (plane features belong to "
                          "the hand branch, out of scope: SURVEY.md section 2 row 10)")
        return c_plane['grid']

    # -- reference call signatures ---------------------------------------------
    def forward(self, p, c_plane, **kwargs):
        """logits [B,N] for points p [B,N,3] (decoder.py:135-161)."""
        grid = self._grid_of(c_plane)
        if self._wants_grad(grid):
            return self._wide_train(p, grid) if self._wide else _DecodeFn.apply(self, p, grid, None, *self._params(False))
        if self._wide:
            return self._wide_fwd(grid, pts=p)
        prec = self._point_precision()
        return ops.decode_fwd(grid, self._blob(precision=prec), pts=p, padding=self.padding, precision=prec)

    def forward_img(self, p, c_plane, c_img, **kwargs):
        """Tactile concat variant (decoder.py:71-103): fc_p_img([p; c_img])."""
        grid = self._grid_of(c_plane)
        if self._wants_grad(grid, c_img):
            return self._wide_train(p, grid, c_img) if self._wide else _DecodeFn.apply(self, p, grid, c_img, *self._params(True))
        if self._wide:
            return self._wide_fwd(grid, pts=p, c_img=c_img.float())
        prec = self._point_precision()
        return ops.decode_fwd(grid, self._blob(img=True, precision=prec), pts=p, c_img=c_img, padding=self.padding, precision=prec)

    def forward_contact(self, p, c_plane, **kwargs):
        """(occupancy logits, contact logits) (decoder.py:105-133)."""
        grid = self._grid_of(c_plane)
        if self._wants_grad(grid) and self._wide:
            return self._wide_train(p, grid, contact=True)
        if self._wants_grad(grid):
            # training with the contact head: the fused decode kernel with both heads and its HIP backward
            return _DecodeContactFn.apply(self, p.float(), grid, *self._params(False),
                                          self.fc_out_contact.weight, self.fc_out_contact.bias)
        if self._wide:
            return self._wide_fwd(grid, pts=p, want_contact=True)
        prec = self._point_precision()
        return ops.decode_fwd(grid, self._blob(contact=True, precision=prec), pts=p, padding=self.padding, want_contact=True,
                              precision=prec)

    # -- dense fast path: the lattice is generated in-kernel ---------------------
    def decode_lattice(self, grid, nx, box=1.1, first=0, count=None, c_img=None, out=None, precision=None):
        """Logits of ``box * make_3d_grid((-.5,)*3,(.5,)*3,(nx,)*3)[first:first+count]``
        (generation.py:155-157 + eval_points) without materialising the points."""
        count = nx ** 3 - first if count is None else count
        if self._wide:
            return self._wide_fwd(grid, precision=precision or self.precision, lattice=(nx, box, first, count), out=out,
                                  **({} if c_img is None else {"c_img": c_img.float()}))
        precision = precision or self.precision
        if precision == "f16f8" and not ops.f16f8_covers(grid, (nx, box, first, count), self.padding):
            precision = "f16x3"                   # slabs the fp8-corrected kernel does not cover
        return ops.decode_fwd(grid, self._blob(img=c_img is not None, precision=precision), c_img=c_img,
                              padding=self.padding, lattice=(nx, box, first, count), out=out, precision=precision)


def _decode_lattice_ids(self, grid, nx, finger_ids, finger_feats, box=1.1, first=0, count=None, out=None, precision=None):
    """``decode_lattice`` with the tactile feature given as (finger id per point, [F,c_dim] table)
    instead of a dense c_img tensor (what the 256^3 configuration needs: 16.7 MB of ids instead of
    2.1 GB of c_img_all)."""
    count = nx ** 3 - first if count is None else count
    if self._wide:
        # the wide kernels read the ids themselves (vt_decode_fwd_wide[_f16x3]_ids): no dense [B, count, c_dim] tensor
        return self._wide_fwd(grid, precision=precision or self.precision, lattice=(nx, box, first, count), out=out,
                              finger_ids=finger_ids.reshape(grid.shape[0], count), finger_feats=finger_feats)
    precision = precision or self.precision
    if precision == "f16f8" and not ops.f16f8_covers(grid, (nx, box, first, count), self.padding):
        precision = "f16x3"
    return ops.decode_fwd_ids(grid, self._blob(img=True, precision=precision), finger_ids, finger_feats,
                              padding=self.padding, lattice=(nx, box, first, count), out=out, precision=precision)


LocalDecoder.decode_lattice_ids = _decode_lattice_ids


class AttentionDecoder(LocalDecoder):
    """``attention_local`` (reference decoder.py:163-329): ``forward_img`` replaces the sampled
    grid features by TransformerFusion(c_img, c) -- attention + InstanceNorm across the N query
    points of the call -- before the same conditioned MLP (with fc_p, not fc_p_img).
    ``forward`` / ``forward_contact`` are LocalDecoder's."""

    def __init__(self, dim=3, c_dim=128, input_size=2048, hidden_size=256, n_blocks=5, leaky=False,
                 sample_mode='bilinear', padding=0.1, with_contact=False, **kwargs):
        super().__init__(dim=dim, c_dim=c_dim, hidden_size=hidden_size, n_blocks=n_blocks, leaky=leaky,
                         sample_mode=sample_mode, padding=padding, with_contact=with_contact)
        if self._wide and (c_dim not in (32, 64, 96, 128) or sample_mode != 'bilinear'):
            raise VtError("AttentionDecoder: the TransformerFusion kernels take c_dim (= d_model) 32, 64, 96 or 128 (the reference's "
                          "default) and the trilinear sample")
        self.mlp_precision = os.environ.get("VTACO_ATTENTION_MLP_PRECISION", "f16x3")
        self.fuser = TransformerFusion(use_xyz=True, input_size=input_size, d_model=c_dim, num_layers=1,
                                       key_feature_dim=64, with_pos_embed=False,
                                       encoder_pos_embed_input_dim=3, decoder_pos_embed_input_dim=3)
        # the reference registers fuser before fc_out_contact; order is irrelevant for load_state_dict

    def forward_img(self, p, c_plane, c_img, **kwargs):
        grid = self._grid_of(c_plane)
        if self._wants_grad(grid, c_img):
            # under autograd every stage is HIP, forward and backward: vt_sample_grid[_bwd], vt_fusion_fwd_train / vt_fusion_bwd
            # (train-mode dropout replayed from a seed), vt_decode_mlp_fwd_train / vt_decode_mlp_bwd / vt_decode_wgrad -- at the
            # widths beyond 32 / 32 vt_decode_mlp_fwd_wide_train / vt_decode_mlp_bwd_wide / vt_rows_wgrad
            c = _SampleGridFn.apply(grid, p, self.padding)
            c = self.fuser.forward_train(c_img, c)
            return (_DecodeMlpWideFn if self._wide else _DecodeMlpFn).apply(self, p, c, *self._params(False))
        c = ops.sample_grid(grid, p, self.padding)
        c = self.fuser(c_img, 1, c, 1)
        return self._mlp_fwd(c, p)

    def _mlp_fwd(self, c, p):
        """The conditioned MLP behind the fusion (inference): split-f16 layers by default (vt_decode_mlp_fwd_f16x3: f32-level logits at
        2.6x the exact-f32 kernel's rate; `mlp_precision = "f32"` / VTACO_ATTENTION_MLP_PRECISION for the exact form, which the
        generator's range guard also falls back to)."""
        prec = self.mlp_precision
        if self._wide:                                  # widths beyond 32 / 32 (the reference's defaults are 128 / 256): decode_wide.hip
            wp = self._wide_precision(prec)
            return ops.decode_mlp_fwd(c, self._blob(precision=wp), p, precision=wp, wide=(self.hidden_size, self.n_blocks, self.leaky))
        return ops.decode_mlp_fwd(c, self._blob(precision=prec), p, precision=prec)
