"""TransformerFusion parameter container (reference src/TransformerFusion.py:269-333).

Keeps the reference's parameter tree -- ``encoder.layers.0.self_attn.head.0.{WK,WQ,WV,
trans_conv,after_norm}``, ``...extra_nonlinear.0.{linear1,linear2,norm2}``,
``decoder.layers.0.{self_attn,cross_attn}...`` -- including the fact that the encoder
layer and the decoder layer's self-attention are the SAME module (state_dict lists it
under both names).  Without autograd the forward is the HIP pipeline ``vt_fusion_fwd`` (eval
mode: the reference's dropout layers are identity there); under autograd it is ``_FusionFn``:
``vt_fusion_fwd_train`` (the same pipeline with TransNonlinear's two train-mode dropouts, masks a
function of a per-call seed drawn from torch's generator) and ``vt_fusion_bwd`` (gradients of
c_img, c and all twenty parameter tensors; the shared self-attention's are the sum of its two uses).
There is no host-operator form of the fuser in this package (the oracle under ``oracle/`` is the checker).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import ops
from ._lib import VtError


class RelationUnit(nn.Module):
    def __init__(self, feature_dim=512, key_feature_dim=64):
        super().__init__()
        self.temp = 1
        self.WK = nn.Linear(feature_dim, key_feature_dim, bias=False)
        self.WQ = nn.Linear(feature_dim, key_feature_dim, bias=False)
        self.WV = nn.Linear(feature_dim, feature_dim, bias=False)
        self.after_norm = nn.BatchNorm1d(feature_dim)          # never called by the reference either
        self.trans_conv = nn.Linear(feature_dim, feature_dim, bias=False)
        for lin in (self.WK, self.WQ, self.WV):
            lin.weight.data.normal_(0, math.sqrt(2.0 / lin.out_features))


class TransNonlinear(nn.Module):
    def __init__(self, d_model, dim_feedforward, dropout=0.1):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p_drop = dropout                 # the reference's `dropout` and `dropout2` (TransformerFusion.py:13-23)


class MultiheadAttention(nn.Module):
    def __init__(self, feature_dim=512, n_head=8, key_feature_dim=64, extra_nonlinear=True):
        super().__init__()
        if n_head != 1 or not extra_nonlinear:
            raise VtError("MultiheadAttention: the fuser is built with n_head=1 and the extra non-linearity")
        self.Nh = n_head
        self.head = nn.ModuleList([RelationUnit(feature_dim, key_feature_dim)])
        self.extra_nonlinear = nn.ModuleList([TransNonlinear(feature_dim, key_feature_dim)])

    def unit_tensors(self):
        h, e = self.head[0], self.extra_nonlinear[0]
        return {"WK": h.WK.weight, "WQ": h.WQ.weight, "WV": h.WV.weight, "trans_conv": h.trans_conv.weight,
                "linear1_w": e.linear1.weight, "linear1_b": e.linear1.bias, "linear2_w": e.linear2.weight,
                "linear2_b": e.linear2.bias, "norm2_w": e.norm2.weight, "norm2_b": e.norm2.bias}


class _Layer(nn.Module):
    def __init__(self, self_attn, cross_attn=None):
        super().__init__()
        self.self_attn = self_attn
        if cross_attn is not None:
            self.cross_attn = cross_attn


class _Stack(nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.layers = nn.ModuleList([layer])


class _FusionFn(torch.autograd.Function):
    """fuse(c_img, c) under autograd on the HIP kernels; ``tensors`` = the ten tensors of the self-attention unit followed by the
    ten of the cross-attention unit (ops.FUSION_TENSORS order)."""

    @staticmethod
    def forward(ctx, c_img, c, p_drop, seed, *tensors):
        n = len(ops.FUSION_TENSORS)
        sa, ca = dict(zip(ops.FUSION_TENSORS, tensors[:n])), dict(zip(ops.FUSION_TENSORS, tensors[n:]))
        out, saved = ops.fusion_fwd_train(c_img, c, sa, ca, p_drop, seed)
        ctx.save_for_backward(c_img, c, saved, *tensors)
        ctx.cfg = (p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, d_out):
        c_img, c, saved, *tensors = ctx.saved_tensors
        n = len(ops.FUSION_TENSORS)
        sa, ca = dict(zip(ops.FUSION_TENSORS, tensors[:n])), dict(zip(ops.FUSION_TENSORS, tensors[n:]))
        p_drop, seed = ctx.cfg
        d_c_img, d_c, g_sa, g_ca = ops.fusion_bwd(d_out, c_img, c, sa, ca, saved, p_drop, seed)
        grads = [g_sa[k] for k in ops.FUSION_TENSORS] + [g_ca[k] for k in ops.FUSION_TENSORS]
        return (d_c_img, d_c, None, None, *grads)


class TransformerFusion(nn.Module):
    def __init__(self, use_xyz=True, input_size=2048, d_model=32, num_layers=1, key_feature_dim=128,
                 with_pos_embed=True, encoder_pos_embed_input_dim=3, decoder_pos_embed_input_dim=3):
        super().__init__()
        if num_layers != 1 or with_pos_embed:
            raise VtError("TransformerFusion: only num_layers=1, with_pos_embed=False (what AttentionDecoder builds)")
        self.d_model, self.input_size = d_model, input_size
        shared = MultiheadAttention(feature_dim=d_model, n_head=1, key_feature_dim=key_feature_dim)
        self.encoder = _Stack(_Layer(shared))
        self.decoder = _Stack(_Layer(shared, MultiheadAttention(feature_dim=d_model, n_head=1,
                                                                key_feature_dim=key_feature_dim)))

    def forward(self, search_feature, search_coord, template_feature, template_coord):
        """fuse(search=c_img [B,N,C], template=c [B,N,C]) -> [B,N,C]  (TransformerFusion.py:311-333)."""
        layer = self.decoder.layers[0]
        sa, ca = layer.self_attn.unit_tensors(), layer.cross_attn.unit_tensors()
        if torch.is_grad_enabled() and (search_feature.requires_grad or template_feature.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            return self.forward_train(search_feature, template_feature)
        if self.training and self.p_drop > 0:
            # train mode without autograd (no_grad evaluation of a model left in train()): dropout still applies
            return ops.fusion_fwd_train(search_feature, template_feature, sa, ca, self.p_drop, self._draw_seed())[0]
        return ops.fusion_fwd(search_feature, template_feature, sa, ca)

    def forward_ids(self, finger_ids, finger_feats, template_feature, chunk_index=None):
        """``forward`` in eval mode with the search features given by finger id (ops.fusion_fwd_ids): the generator's lattice
        chunks, whose tactile rows the reference gathers on the host (generation.py:159-255)."""
        layer = self.decoder.layers[0]
        if self.training and self.p_drop > 0:
            raise VtError("TransformerFusion.forward_ids is the eval-mode path (dropout would apply in train mode)")
        if template_feature.shape[-1] != 32:
            # d_model beyond 32 (vt_fusion_fwd's generic-width kernels take the dense rows): gather them from the table
            feats = finger_feats.float()
            table = torch.cat([feats, feats.new_zeros(1, feats.shape[1])])
            rows = finger_ids if chunk_index is None else finger_ids[chunk_index.long()]
            gathered = table[torch.where(rows >= feats.shape[0], torch.full_like(rows, feats.shape[0]), rows).long()]   # 255 or past the table: zero row
            return ops.fusion_fwd(gathered, template_feature, layer.self_attn.unit_tensors(), layer.cross_attn.unit_tensors())
        return ops.fusion_fwd_ids(finger_ids, finger_feats, template_feature, layer.self_attn.unit_tensors(),
                                  layer.cross_attn.unit_tensors(), chunk_index=chunk_index)

    @property
    def p_drop(self):
        return self.decoder.layers[0].self_attn.extra_nonlinear[0].p_drop

    @staticmethod
    def _draw_seed():
        """A fresh 63-bit seed from torch's CPU generator (so ``torch.manual_seed`` makes a run repeatable)."""
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    def forward_train(self, search_feature, template_feature, seed=None):
        """Differentiable HIP path (vt_fusion_fwd_train / vt_fusion_bwd); dropout active in training mode only."""
        layer = self.decoder.layers[0]
        sa, ca = layer.self_attn.unit_tensors(), layer.cross_attn.unit_tensors()
        p = self.p_drop if self.training else 0.0
        if seed is None:
            seed = self._draw_seed() if p > 0 else 0
        self.last_seed = seed
        return _FusionFn.apply(search_feature, template_feature, p, seed,
                               *[sa[k] for k in ops.FUSION_TENSORS], *[ca[k] for k in ops.FUSION_TENSORS])
