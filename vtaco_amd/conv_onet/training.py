"""Trainer for the object branch of the path (reference src/conv_onet/training.py:35-104, 454-500).

``train_step(data)`` consumes the dictionaries ``vtaco_amd.data`` (or the reference loader) produces:
``inputs`` [B,T,3] -> encode_inputs -> feature grid; ``points`` [B,N,3] -> decode -> logits;
``loss = F.l1_loss(logits, points.occ)`` exactly as the reference's ``compute_loss`` (:487-497).  With a hand
encoder (``model.encoder_hand``) its two other terms are added as the reference does (:493-498):
``loss_mano = mse(mano_param, points.mano)`` and ``loss_pc = mse(mano_verts, points.pc_hand)``; without one
they are reported as 0.  Forward and backward of the whole step (voxeliser, UNet3D, decoder; plane pooling
and scatter of the hand encoder) run on the HIP kernels; the MANO layer and the 2-D U-Net differentiate
through host PyTorch ops.  The tactile variants (``with_img`` / ``encode_t2d``) need the per-point
contact features the reference assembles with CPU geometry (igl / cdist, training.py:817-866): feed them
at model level (``model.decode_img(p, c, c_img)``) or as finger ids (``vt_tactile_assign``) instead.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.nn import functional as F

from .._lib import VtError
from ..eval import compute_iou


class Trainer:
    def __init__(self, model, optimizer, device=None, input_type='pointcloud', vis_dir=None, threshold=0.5,
                 eval_sample=False, num_sample=2048, with_img=False, with_contact=False, train_tactile=False,
                 encode_t2d=False, pretrained_t2d=True, grad_sync=None):
        if with_img or encode_t2d or with_contact or train_tactile:
            raise VtError("Trainer: only the visual object branch is built (with_img / encode_t2d / with_contact / "
                          "train_tactile need the reference's CPU tactile-assembly glue; use the model-level API)")
        self.model, self.optimizer, self.device = model, optimizer, device
        self.input_type, self.threshold = input_type, threshold
        # data-parallel training (one process per GPU): a callable run between backward and the optimizer step,
        # e.g. vtaco_amd.dist.GradAllReduce(model.parameters()) -- one flat-bucket RCCL all-reduce per step that
        # also covers the parameters a step leaves without gradient (fc_p_img, the contact head)
        self.grad_sync = grad_sync

    def compute_loss(self, data):
        """(loss, loss_mano, loss_pc) -- the last two are the hand branch's (0 without a hand encoder)."""
        p = data.get('points').to(self.device)
        occ = data.get('points.occ').to(self.device)
        inputs = data.get('inputs').to(self.device)
        c = self.model.encode_inputs(inputs)
        logits = self.model.decode(p, c).logits
        loss = F.l1_loss(logits, occ)
        if getattr(self.model, 'encoder_hand', None) is None:
            zero = logits.new_zeros(())
            return loss, zero, zero
        c_hand = self.model.encode_hand_inputs(inputs)
        loss_mano = F.mse_loss(c_hand['mano_param'], data.get('points.mano').to(self.device).float())
        loss_pc = F.mse_loss(c_hand['mano_verts'], data.get('points.pc_hand').to(self.device).float())
        return loss + loss_mano + loss_pc, loss_mano, loss_pc

    def train_step(self, data, vf_dict=None):
        self.model.train()
        self.optimizer.zero_grad()
        loss, loss_mano, loss_pc = self.compute_loss(data)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        self.optimizer.step()
        return loss.item(), loss_mano.item(), loss_pc.item()

    def eval_step(self, data, vf_dict=None):
        """{'loss', 'iou'}: L1 loss on ``points`` and the reference's IoU (compute_iou: both sides cut at the
        mean ground-truth occupancy) on ``points_iou``."""
        self.model.eval()
        with torch.no_grad():
            inputs = data.get('inputs').to(self.device)
            c = self.model.encode_inputs(inputs)
            logits = self.model.decode(data.get('points').to(self.device), c).logits
            out = {'loss': F.l1_loss(logits, data.get('points.occ').to(self.device)).item()}
            if data.get('points_iou') is not None:
                occ_hat = self.model.decode(data.get('points_iou').to(self.device), c).probs
                occ_iou = data.get('points_iou.occ')
                out['iou'] = float(np.mean(compute_iou(occ_hat.cpu().numpy(), occ_iou.numpy(), self.threshold)))
        return out

    def evaluate(self, val_loader):
        """Mean of eval_step over a loader."""
        sums, n = {}, 0
        for batch in val_loader:
            for k, v in self.eval_step(batch).items():
                sums[k] = sums.get(k, 0.0) + v
            n += 1
        return {k: v / max(n, 1) for k, v in sums.items()}
