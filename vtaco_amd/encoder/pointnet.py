"""PointNet local-pool encoder -> 3-D feature grid (drop-in for reference
src/encoder/pointnet.py:12-210, ``plane_type='grid'``).

The per-point MLP (fc_pos, 5 ResnetBlockFC, fc_c: 0.16 GFLOP/scene) is host
PyTorch-ROCm; the voxel bookkeeping, the 4 local max-pool rounds and the final
scatter-mean -- torch_scatter in the reference -- are HIP kernels (vt_voxel_*),
wrapped in autograd Functions whose backward is also HIP.  Plane features
('xz','xy','yz') belong to the hand branch and are not built.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from .. import ops
from .._lib import VtError
from ..layers import ResnetBlockFC
from .unet3d import UNet3D


class _PoolMax(torch.autograd.Function):
    """pool_local (pointnet.py:116-132): per-voxel channel max, gathered back."""

    @staticmethod
    def forward(ctx, feat, vi):
        out, arg = ops.voxel_pool_max_fwd(feat, vi, want_argmax=True)
        ctx.vi, ctx.arg = vi, arg
        return out

    @staticmethod
    def backward(ctx, grad):
        return ops.voxel_pool_max_bwd(grad, ctx.arg, ctx.vi), None


class _ScatterMean(torch.autograd.Function):
    """generate_grid_features' scatter (pointnet.py:102-110)."""

    @staticmethod
    def forward(ctx, feat, vi):
        ctx.vi, ctx.C = vi, feat.shape[2]
        return ops.voxel_scatter_mean_fwd(feat, vi)

    @staticmethod
    def backward(ctx, grad):
        return ops.voxel_scatter_mean_bwd(grad, ctx.vi, ctx.C), None


class _ScatterMeanCL(torch.autograd.Function):
    """Same scatter, channels-last: returns a [B,C,R,R,R]-shaped tensor with channels_last_3d
    strides (what the UNet3D's channels-last convs and the decode kernel both want)."""

    @staticmethod
    def forward(ctx, feat, vi):
        ctx.vi, ctx.C = vi, feat.shape[2]
        return ops.voxel_scatter_mean_cl_fwd(feat, vi).permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, grad):
        return ops.voxel_scatter_mean_cl_bwd(grad.permute(0, 2, 3, 4, 1).contiguous(), ctx.vi, ctx.C), None


class LocalPoolPointnet(nn.Module):
    """Args as the reference (pointnet.py:32-35); only scatter_type='max',
    plane_type='grid' (or ['grid']) are built."""

    def __init__(self, c_dim=128, dim=3, hidden_dim=128, scatter_type='max', unet=False, unet_kwargs=None,
                 unet3d=False, unet3d_kwargs=None, plane_resolution=None, grid_resolution=None,
                 plane_type='xz', padding=0.1, n_blocks=5, out_mano=False, out_dim=None,
                 manolayer_kwargs=None, **kwargs):
        super().__init__()
        if scatter_type != 'max':
            if scatter_type == 'mean':
                raise VtError("LocalPoolPointnet: scatter_type='mean' pooling is not built (configs use 'max')")
            raise ValueError('incorrect scatter type')
        planes = [plane_type] if isinstance(plane_type, str) else list(plane_type)
        if planes != ['grid'] or unet or out_mano or manolayer_kwargs is not None:
            raise VtError("LocalPoolPointnet: only plane_type='grid' without the hand/MANO head is built "
                          "(plane features / MANO: out of scope, SURVEY.md section 2 rows 10, 17)")
        if grid_resolution is None:
            raise VtError("LocalPoolPointnet: grid_resolution is required")
        self.c_dim, self.hidden_dim = c_dim, hidden_dim
        self.fc_pos = nn.Linear(dim, 2 * hidden_dim)
        self.blocks = nn.ModuleList(ResnetBlockFC(2 * hidden_dim, hidden_dim) for _ in range(n_blocks))
        self.fc_c = nn.Linear(hidden_dim, c_dim)
        self.unet = None
        # channels_last_3d parameters: MIOpen's f32 conv3d is ~14x faster in that layout on gfx950
        # (26 vs 382 ms fwd+bwd for two 64^3 scenes), and it is the layout the HIP kernels use
        self.unet3d = UNet3D(**unet3d_kwargs).to(memory_format=torch.channels_last_3d) if unet3d else None
        self.reso_plane, self.reso_grid = plane_resolution, grid_resolution
        self.plane_type, self.padding = plane_type, padding
        # UNet3D under autograd: "hip" = vt_* forward and backward kernels (UNet3D.forward_channels_last_train),
        # "host" = PyTorch-ROCm autograd (MIOpen; fast only in find mode, torch.backends.cudnn.benchmark = True)
        self.train_unet3d = os.environ.get("VTACO_TRAIN_UNET3D", "hip")

    def point_features(self, p, vi):
        """fc_pos -> block0 -> 4 x (local max-pool, concat, block) -> fc_c  (pointnet.py:154-162)."""
        net = self.blocks[0](self.fc_pos(p))
        for blk in self.blocks[1:]:
            pooled = _PoolMax.apply(net, vi)
            net = blk(torch.cat([net, pooled], dim=2))
        return self.fc_c(net)

    def forward(self, p):
        if not p.is_cuda:
            raise VtError(f"LocalPoolPointnet: inputs must live on a HIP device (got {p.device})")
        vi = ops.VoxelIndex(p, self.reso_grid, self.padding)
        feat = self.point_features(p.float(), vi)
        if self.unet3d is not None and not torch.is_grad_enabled() and self.unet3d.hip_supported():
            # inference: scatter straight into a channels-last grid, UNet3D on the HIP conv kernels,
            # and hand the decoder the layout it samples (shape [B,C,R,R,R], channels-last strides)
            grid = self.unet3d.forward_channels_last(ops.voxel_scatter_mean_cl_fwd(feat, vi))
            return {'grid': grid.permute(0, 4, 1, 2, 3)}
        if self.unet3d is not None and self.unet3d.hip_supported() and self.train_unet3d == "hip":
            # training on the HIP kernels: differentiable channels-last forward, HIP backward
            grid_cl = _ScatterMeanCL.apply(feat, vi).permute(0, 2, 3, 4, 1)
            return {'grid': self.unet3d.forward_channels_last_train(grid_cl).permute(0, 4, 1, 2, 3)}
        if self.unet3d is not None:
            # training through host PyTorch-ROCm autograd (MIOpen), channels-last end to end
            return {'grid': self.unet3d(_ScatterMeanCL.apply(feat, vi))}
        return {'grid': _ScatterMean.apply(feat, vi)}
