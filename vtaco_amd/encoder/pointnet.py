"""PointNet local-pool encoder -> 3-D feature grid (drop-in for reference
src/encoder/pointnet.py:12-210, ``plane_type='grid'``).

The per-point MLP (fc_pos, 5 ResnetBlockFC, fc_c: 0.16 GFLOP/scene) is host
PyTorch-ROCm; the voxel bookkeeping, the 4 local max-pool rounds and the final
scatter-mean -- torch_scatter in the reference -- are HIP kernels (vt_voxel_*),
wrapped in autograd Functions whose backward is also HIP.

The hand branch (``encoder_hand``: plane_type ['xz','xy','yz'], 2-D U-Net, ``out_mano``) uses the
same pooling kernels over per-plane cell ids (vt_plane_build), scatter-means into [B,C,R,R] planes
(vt_plane_scatter_mean_*), and ends in the MANO layer (manolayer.py).
"""
from __future__ import annotations

import os

import torch
from torch import nn

from .. import ops
from .._lib import VtError
from ..layers import ResnetBlockFC, tall_linear
from .manolayer import ManoLayer
from .unet import UNet
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


class _PoolMaxSum(torch.autograd.Function):
    """The sum of pool_local over several index sets (the hand encoder's three planes, pointnet.py:116-132) in one launch each way
    (vt_voxel_pool_max_sum_fwd / _bwd) instead of a pool per set and the framework's adds."""

    @staticmethod
    def forward(ctx, feat, vis):
        out, args = ops.voxel_pool_max_sum_fwd(feat, vis, want_argmax=True)
        ctx.vis, ctx.args = vis, args
        return out

    @staticmethod
    def backward(ctx, grad):
        return ops.voxel_pool_max_sum_bwd(grad, ctx.args, ctx.vis), None


class _PoolMean(torch.autograd.Function):
    """pool_local with scatter_type='mean' (pointnet.py:64-69, 116-132): per-cell mean, gathered back; self-adjoint."""

    @staticmethod
    def forward(ctx, feat, vi):
        ctx.vi = vi
        return ops.voxel_pool_mean(feat, vi)

    @staticmethod
    def backward(ctx, grad):
        return ops.voxel_pool_mean(grad, ctx.vi), None


class _LinearRowsFn(torch.autograd.Function):
    """nn.Linear over the rows of [B,T,Cin] on the HIP kernels: forward vt_linear_rows; backward vt_linear_rows on the
    transposed weight (data gradient) and vt_rows_wgrad (weight / bias gradients)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.linear_rows(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = ops.linear_rows(dy, weight.t().contiguous(), None) if ctx.needs_input_grad[0] else None
        dw, db = ops.rows_wgrad(dy, x, want_bias=ctx.has_bias)
        return dx, dw, db


class _ResBlockFcFn(torch.autograd.Function):
    """ResnetBlockFC on the virtual concat [x1 | x2] (layers.py:8-50): forward vt_resblock_fc, backward
    vt_resblock_fc_bwd (data gradient; h is recomputed) + three vt_rows_wgrad calls (fc_1, fc_0, shortcut)."""

    @staticmethod
    def forward(ctx, x1, x2, w0, b0, w1, b1, ws):
        ctx.save_for_backward(x1, x2, w0, b0, w1, ws)
        lib = ops._lib.load()
        C1, C2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
        H, O = w0.shape[0], w1.shape[0]
        out = torch.empty(x1.shape[:-1] + (O,), dtype=torch.float32, device=x1.device)
        c = ops._c
        ops.check(lib.vt_resblock_fc(ops.dev_ptr(c(x1), "x1"), C1, ops.dev_ptr(c(x2) if x2 is not None else None, "x2"), C2,
                                     x1.numel() // C1, ops.dev_ptr(c(w0), "w0"), ops.dev_ptr(c(b0), "b0"), ops.dev_ptr(c(w1), "w1"),
                                     ops.dev_ptr(c(b1), "b1"), ops.dev_ptr(c(ws) if ws is not None else None, "ws"), H, O,
                                     ops.dev_ptr(out, "out"), ops.stream_ptr()), "vt_resblock_fc")
        return out

    @staticmethod
    def backward(ctx, dout):
        x1, x2, w0, b0, w1, ws = ctx.saved_tensors
        dx1, dx2, act, dh = ops.resblock_fc_bwd(x1, x2, w0, b0, w1, ws, dout, want_dx2=x2 is not None and ctx.needs_input_grad[1])
        grads = ops.resblock_wgrad(x1, x2, act, dh, dout, ws is not None)       # the three products in one pair of launches
        if grads is not None:
            dw0, db0, dw1, db1, dws = grads
            return dx1, dx2, dw0, db0, dw1, db1, dws
        dw1, db1 = ops.rows_wgrad(dout, act)
        dw0, db0 = ops.rows_wgrad(dh, x1, x2, relu_x=True)
        dws = ops.rows_wgrad(dout, x1, x2, want_bias=False)[0] if ws is not None else None
        return dx1, dx2, dw0, db0, dw1, db1, dws


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


class _ScatterMeanPlane(torch.autograd.Function):
    """generate_plane_features' scatter (pointnet.py:85-95): [B,T,C] -> [B,C,R,R]."""

    @staticmethod
    def forward(ctx, feat, pi):
        ctx.pi, ctx.C = pi, feat.shape[2]
        return ops.plane_scatter_mean_fwd(feat, pi)

    @staticmethod
    def backward(ctx, grad):
        return ops.plane_scatter_mean_bwd(grad, ctx.pi, ctx.C), None


class _ScatterMeanPlanes(torch.autograd.Function):
    """_ScatterMeanPlane for the planes of one ops.plane_indices call in one launch each way: [B,T,C] -> [n * B, C, R, R], the planes one
    after the other (what torch.cat of the per-plane tensors gives, and what the U-Net takes)."""

    @staticmethod
    def forward(ctx, feat, pis):
        ctx.pis, ctx.C = pis, feat.shape[2]
        return ops.plane_scatter_mean_multi_fwd(feat, pis)

    @staticmethod
    def backward(ctx, grad):
        return ops.plane_scatter_mean_multi_bwd(grad, ctx.pis, ctx.C), None


class LocalPoolPointnet(nn.Module):
    """Args as the reference (pointnet.py:32-35).  Built: scatter_type 'max' (the configs') and 'mean' with plane_type 'grid'
    (object encoder) or any of 'xz','xy','yz' (hand encoder; 'grid' and planes are not mixed)."""

    def __init__(self, c_dim=128, dim=3, hidden_dim=128, scatter_type='max', unet=False, unet_kwargs=None,
                 unet3d=False, unet3d_kwargs=None, plane_resolution=None, grid_resolution=None,
                 plane_type='xz', padding=0.1, n_blocks=5, out_mano=False, out_dim=None,
                 manolayer_kwargs=None, **kwargs):
        super().__init__()
        if scatter_type not in ('max', 'mean'):
            raise ValueError('incorrect scatter type')
        self.scatter_type = scatter_type
        planes = [plane_type] if isinstance(plane_type, str) else list(plane_type)
        # the reference walks the keys in this fixed order whatever the list says (pointnet.py:141-176)
        self.planes = [k for k in ('grid', 'xz', 'xy', 'yz') if k in planes]
        if not self.planes or len(self.planes) != len(planes):
            raise VtError(f"LocalPoolPointnet: plane_type {plane_type!r} has entries other than 'grid','xz','xy','yz'")
        if 'grid' in self.planes and len(self.planes) > 1:
            raise VtError("LocalPoolPointnet: 'grid' mixed with canonical planes is not built (no config uses it)")
        if self.planes == ['grid'] and grid_resolution is None:
            raise VtError("LocalPoolPointnet: grid_resolution is required")
        if self.planes != ['grid'] and plane_resolution is None:
            raise VtError("LocalPoolPointnet: plane_resolution is required")
        if self.planes == ['grid'] and unet:
            raise VtError("LocalPoolPointnet: unet=True needs canonical planes (plane_type 'xz'/'xy'/'yz')")
        self.c_dim, self.hidden_dim = c_dim, hidden_dim
        self.fc_pos = nn.Linear(dim, 2 * hidden_dim)
        self.blocks = nn.ModuleList(ResnetBlockFC(2 * hidden_dim, hidden_dim) for _ in range(n_blocks))
        self.fc_c = nn.Linear(hidden_dim, c_dim)
        self.unet = UNet(c_dim, in_channels=c_dim, **(unet_kwargs or {})) if unet else None
        self.out_mano, self.out_dim = out_mano, out_dim
        if manolayer_kwargs is not None:
            self.mano_layer = ManoLayer(**manolayer_kwargs)
        if out_mano:
            if out_dim is None:
                raise VtError("LocalPoolPointnet: out_mano=True needs out_dim")
            # pointnet.py:78-82: the three-plane head reads 3*c_dim pooled channels, the grid head c_dim
            if self.planes == ['xz', 'xy', 'yz']:
                self.fc_mano = nn.Linear(c_dim * 3, out_dim)
            elif self.planes == ['grid']:
                self.fc_mano = nn.Linear(c_dim, out_dim)
            else:
                raise VtError("LocalPoolPointnet: out_mano=True is defined for the three planes or the grid only")
            if out_dim > 30 and manolayer_kwargs is None:
                raise VtError("LocalPoolPointnet: out_dim > 30 runs the MANO layer (pointnet.py:194-201): pass manolayer_kwargs")
        # (parameters in the default layout: the HIP kernels read [Cout,Cin,3,3,3] as it is -- channels_last_3d weights, which MIOpen's
        # f32 conv3d wanted when the modules ran the net, cost three re-layout copies per layer and step on the packing kernels' way)
        self.unet3d = UNet3D(**unet3d_kwargs) if unet3d else None
        self.reso_plane, self.reso_grid = plane_resolution, grid_resolution
        self.plane_type, self.padding = plane_type, padding
        # inference: the UNet3D's first layer skips the 8^3 blocks of the mean grid that no point comes near (ops.voxel_tile_flags)
        self.skip_empty = os.environ.get("VTACO_UNET_SKIP", "1") != "0"
        # UNet3D under autograd: "hip" = vt_* forward and backward kernels (UNet3D.forward_channels_last_train),
        # "host" = PyTorch-ROCm autograd (MIOpen; fast only in find mode, torch.backends.cudnn.benchmark = True)
        self.train_unet3d = os.environ.get("VTACO_TRAIN_UNET3D", "hip")
        # the per-point MLP under autograd: "hip" = vt_linear_rows / vt_resblock_fc forward, vt_resblock_fc_bwd / vt_rows_wgrad
        # backward; "host" = nn.Linear (hipBLASLt) with the split-K weight gradient of layers._TallLinear
        self.train_mlp = os.environ.get("VTACO_TRAIN_POINTNET_MLP", "hip")

    def point_features(self, p, vi):
        """fc_pos -> block0 -> 4 x (local max-pool, concat, block) -> fc_c  (pointnet.py:154-162).
        ``vi``: one VoxelIndex, or a list of PlaneIndex whose pooled features are summed (pointnet.py:116-132)."""
        if not torch.is_grad_enabled() and self._fused_mlp_fits():
            return self._point_features_fused(p, vi)
        hip = self._fused_mlp_fits() and self.train_mlp == "hip"

        def block(blk, x1, x2=None):
            if hip:          # HIP forward and backward (the concat with the pooled features is read in place)
                return _ResBlockFcFn.apply(x1, x2, blk.fc_0.weight, blk.fc_0.bias, blk.fc_1.weight, blk.fc_1.bias,
                                           blk.shortcut.weight if blk.shortcut is not None else None)
            return blk(x1 if x2 is None else torch.cat([x1, x2], dim=2))
        def linear(lin, x):
            if hip and self._linear_fits(lin):
                return _LinearRowsFn.apply(x, lin.weight, lin.bias)
            return tall_linear(lin, x)
        net = block(self.blocks[0], linear(self.fc_pos, p))
        pool = _PoolMax.apply if self.scatter_type == 'max' else _PoolMean.apply
        for blk in self.blocks[1:]:
            if isinstance(vi, (list, tuple)):
                if self.scatter_type == 'max' and 1 < len(vi) <= 4:
                    pooled = _PoolMaxSum.apply(net, list(vi))
                else:
                    pooled = pool(net, vi[0])
                    for other in vi[1:]:
                        pooled = pooled + pool(net, other)
            else:
                pooled = pool(net, vi)
            net = block(blk, net, pooled)
        return linear(self.fc_c, net)

    def _fused_mlp_fits(self):
        """vt_resblock_fc keeps a block's three weight matrices in 64 KiB of LDS (hidden_dim <= 48 or so: the shipped
        configs use 32); wider PointNets keep the nn.Linear (hipBLASLt) path."""
        h = self.hidden_dim
        floats = 2 * h * h + h * h + 2 * h * h + (256 // h if h <= 256 else 0) * 3 * h
        return h <= 256 and floats * 4 <= 64 * 1024 and 2 * h <= 256

    @staticmethod
    def _linear_fits(lin):
        """vt_linear_rows: at most 256 output channels, the transposed weight and a few input rows in 64 KiB of LDS (the t2d hand
        encoder's fc_c -- 32 -> 512 -- does not: it keeps nn.Linear)."""
        cout, cin = lin.weight.shape
        return cout <= 256 and (cin * (cout | 1) + max(1, 256 // cout) * cin) * 4 <= 64 * 1024

    def _fused_weights(self):
        """The one-launch kernel's weight pointers, gathered again only when a parameter changed its storage (29 tensors)."""
        ts = [self.fc_pos.weight, self.fc_pos.bias, self.fc_c.weight, self.fc_c.bias] + [
            t for b in self.blocks for t in (b.fc_0.weight, b.fc_0.bias, b.fc_1.weight, b.fc_1.bias, b.shortcut.weight)]
        stamp = tuple((id(t), t.data_ptr()) for t in ts)
        hit = getattr(self, "_fused_w", None)
        if hit is None or hit[0] != stamp:
            hit = self._fused_w = (stamp, ops.pointnet_mlp_weights(self.fc_pos, self.blocks, self.fc_c))
        return hit[1]

    def _one_launch_fits(self, vi):
        """vt_pointnet_mlp_fused: one voxel index (the object grid; the hand encoder sums three planes' pools), the shipped widths
        (hidden 32, five blocks with shortcut layers, c_dim <= 64).  VTACO_POINTNET_ONE_LAUNCH=0: the launch-per-layer path."""
        if isinstance(vi, (list, tuple)) or os.environ.get("VTACO_POINTNET_ONE_LAUNCH", "1") == "0" or self.scatter_type != 'max':
            return False
        return (self.hidden_dim == 32 and len(self.blocks) == 5 and all(b.shortcut is not None for b in self.blocks)
                and tuple(self.fc_pos.weight.shape) == (64, 3) and self.fc_c.weight.shape[0] <= 64 and self.fc_c.weight.shape[1] == 32
                and self.fc_pos.bias is not None and self.fc_c.bias is not None
                and all(w.data_ptr() % 16 == 0 for b in self.blocks for w in (b.fc_0.weight, b.fc_1.weight, b.shortcut.weight)))

    def _point_features_fused(self, p, vi):
        """The same layers without autograd: one HIP launch per linear layer / ResnetBlockFC (vt_linear_rows,
        vt_resblock_fc, the concat with the pooled features read in place) instead of ~9 framework launches per block."""
        if self._one_launch_fits(vi):
            return ops.pointnet_mlp_fused(p, vi, self.fc_pos, self.blocks, self.fc_c, weights=self._fused_weights())
        lin = lambda l, x: ops.linear_rows(x, l.weight, l.bias) if self._linear_fits(l) else l(x)
        net = lin(self.fc_pos, p)
        b0 = self.blocks[0]
        net = ops.resblock_fc(net, None, b0.fc_0, b0.fc_1, b0.shortcut)
        pool = (lambda f, v: ops.voxel_pool_max_fwd(f, v, want_argmax=False)[0]) if self.scatter_type == 'max' else ops.voxel_pool_mean
        for blk in self.blocks[1:]:
            if isinstance(vi, (list, tuple)):
                if self.scatter_type == 'max' and 1 < len(vi) <= 4:
                    pooled = ops.voxel_pool_max_sum_fwd(net, list(vi), want_argmax=False)[0]
                else:
                    pooled = pool(net, vi[0])
                    for other in vi[1:]:
                        pooled = pooled + pool(net, other)
            else:
                pooled = pool(net, vi)
            net = ops.resblock_fc(net, pooled, blk.fc_0, blk.fc_1, blk.shortcut)
        return lin(self.fc_c, net)

    def _mano_head(self, fea):
        """out_mano (pointnet.py:179-201): pooled plane/grid features -> mano_param (-> MANO layer)."""
        cat = torch.cat([fea[k] for k in fea], dim=1)
        pooled = cat.mean(dim=tuple(range(2, cat.dim())))
        param = self.fc_mano(pooled)
        out = {'mano_param': param}
        if self.out_dim > 30:
            # the wrist position slot is zeroed and what follows the six wrist numbers is the hand pose (:195-197)
            full = torch.cat([torch.zeros_like(param[:, :3]), param[:, 6:]], dim=1)
            out.update(self.forward_mano(full))
        return out

    def forward_mano(self, fea_m_full):
        """pointnet.py:204-212."""
        verts, joints = self.mano_layer(fea_m_full)[:2]
        return {'mano_verts': verts, 'mano_joints': joints, 'mano_faces': self.mano_layer.th_faces}

    def forward_planes(self, p):
        pis = ops.plane_indices(p, self.reso_plane, self.padding, self.planes)      # the planes' sorts in one launch
        feat = self.point_features(p.float(), pis)
        if len(pis) > 1 and ops.plane_group(pis) is not None:
            stacked = _ScatterMeanPlanes.apply(feat, pis)           # the planes' scatter-means in one launch, already stacked
        else:
            stacked = torch.cat([_ScatterMeanPlane.apply(feat, pi) for pi in pis], dim=0)
        if self.unet is not None:
            # one U-Net pass over the planes stacked on the batch axis: the net has no cross-sample op
            # (no normalisation layers), so this equals the reference's per-plane calls with a third of the launches
            stacked = self.unet(stacked)
        return dict(zip(self.planes, stacked.split(p.shape[0], dim=0)))

    def forward(self, p):
        if not p.is_cuda:
            raise VtError(f"LocalPoolPointnet: inputs must live on a HIP device (got {p.device})")
        fea = self.forward_planes(p) if self.planes != ['grid'] else self.forward_grid(p)
        return self._mano_head(fea) if self.out_mano else fea

    def forward_grid(self, p):
        one_launch = (self.unet3d is not None and not torch.is_grad_enabled() and self.unet3d.hip_supported() and self._fused_mlp_fits()
                      and self.unet3d.encoders[0].basic_module.SingleConv1.conv.in_channels == self.c_dim)
        # the mean grid is cleared by the voxel sort's idle workgroups (one launch less; unused if a cell overflows the one-launch MLP)
        R = self.reso_grid
        zeroed = torch.empty((p.shape[0], R, R, R, self.c_dim), dtype=torch.float32, device=p.device) if one_launch else None
        # the blocks of the grid no point comes near are zero, and the UNet3D's first layer skips their taps (VTACO_UNET_SKIP=0: dense)
        skip = self.skip_empty and self.unet3d is not None and self.unet3d.hip_supported() and (not torch.is_grad_enabled() or self.train_unet3d == "hip")
        vi = ops.VoxelIndex(p, self.reso_grid, self.padding, clear=zeroed, want_tile_flags=skip)
        flags = vi.tile_flags
        if one_launch and self._one_launch_fits(vi):
            # inference: the per-point MLP, the voxeliser's mean and the grid's GroupNorm statistics from one launch, then the UNet3D
            grid, stats = ops.pointnet_mlp_fused(p.float(), vi, self.fc_pos, self.blocks, self.fc_c, want_grid=True,
                                                 weights=self._fused_weights(), zeroed_grid=zeroed)
            return {'grid': self.unet3d.forward_channels_last(grid, in_stats=stats, tile_flags=flags).permute(0, 4, 1, 2, 3)}
        feat = self.point_features(p.float(), vi)
        if self.unet3d is not None and not torch.is_grad_enabled() and self.unet3d.hip_supported():
            # inference: scatter straight into a channels-last grid, UNet3D on the HIP conv kernels,
            # and hand the decoder the layout it samples (shape [B,C,R,R,R], channels-last strides)
            grid = self.unet3d.forward_channels_last(ops.voxel_scatter_mean_cl_fwd(feat, vi), tile_flags=flags)
            return {'grid': grid.permute(0, 4, 1, 2, 3)}
        if self.unet3d is not None and self.unet3d.hip_supported() and self.train_unet3d == "hip":
            # training on the HIP kernels: differentiable channels-last forward, HIP backward
            grid_cl = _ScatterMeanCL.apply(feat, vi).permute(0, 2, 3, 4, 1)
            return {'grid': self.unet3d.forward_channels_last_train(grid_cl, tile_flags=flags).permute(0, 4, 1, 2, 3)}
        if self.unet3d is not None:
            # training through host PyTorch-ROCm autograd (MIOpen), channels-last end to end
            return {'grid': self.unet3d(_ScatterMeanCL.apply(feat, vi))}
        return {'grid': _ScatterMean.apply(feat, vi)}
