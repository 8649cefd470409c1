"""2-D U-Net over the hand encoder's 32x32 (or 64x64) feature planes (drop-in for reference
src/encoder/unet.py:52-233; built by LocalPoolPointnet when ``unet: True``, pointnet.py:49-50).

On a HIP device the shipped shapes (concat merge, transposed-conv upsampling, channel counts multiples of 32, power-of-two
planes: ``ops.plane_unet_supported``) run as ONE persistent launch, ``vt_plane_unet_fwd`` (csrc/plane_unet.hip: every conv a phase
of the kernel, bias / ReLU / max-pool / concat in the consuming phase's loader); other configurations (``up_mode='upsample'``,
``merge_mode='add'``, odd channel counts) keep the nn.Conv2d modules below (host PyTorch-ROCm / MIOpen).
Parameter names follow the reference checkpoint: ``down_convs.{i}.conv{1,2}``,
``up_convs.{i}.{upconv,conv1,conv2}``, ``conv_final``.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops

# A/B knob: "0" keeps the nn.Conv2d modules (MIOpen) on every shape
_HIP_UNET = os.environ.get("VTACO_PLANE_UNET", "1") != "0"
# A/B knob: "host" trains through the nn.Conv2d modules' autograd (MIOpen) while inference stays on the HIP kernel
_HIP_UNET_TRAIN = os.environ.get("VTACO_PLANE_UNET_TRAIN", "hip") != "host"


class _PlaneUNetFn(torch.autograd.Function):
    """UNet.forward under autograd on the HIP kernels: vt_plane_unet_fwd with its workspace kept, vt_plane_unet_bwd.  ``params`` = the
    net's parameters in ``named_parameters()`` order (so autograd hands the gradients back to them)."""

    @staticmethod
    def forward(ctx, x, net, *params):
        blob = net._blob()
        ws = ops.plane_unet_workspace(net, x.shape[0], x.shape[2], x.shape[3], fresh=True)
        out = ops.plane_unet_fwd(x, net, blob, ws)
        ctx.net, ctx.blob, ctx.ws = net, blob, ws
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, = ctx.saved_tensors
        dx, grads = ops.plane_unet_bwd(x, ctx.net, ctx.blob, ctx.ws, dout)
        return (dx, None) + tuple(grads[name] for name, _ in ctx.net.named_parameters())


class DownConv(nn.Module):
    """conv3x3-ReLU twice, then an optional 2x2 max-pool; returns (pooled, before_pool)."""

    def __init__(self, in_channels, out_channels, pooling=True):
        super().__init__()
        self.in_channels, self.out_channels, self.pooling = in_channels, out_channels, pooling
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if pooling:
            self.pool = nn.MaxPool2d(2, 2)

    def forward(self, x):
        skip = F.relu(self.conv2(F.relu(self.conv1(x))))
        return (self.pool(skip) if self.pooling else skip), skip


class UpConv(nn.Module):
    """2x2 stride-2 transposed conv (or bilinear upsample + 1x1), merge with the skip, conv3x3-ReLU twice."""

    def __init__(self, in_channels, out_channels, merge_mode="concat", up_mode="transpose"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.merge_mode, self.up_mode = merge_mode, up_mode
        if up_mode == "transpose":
            self.upconv = nn.ConvTranspose2d(in_channels, out_channels, 2, stride=2)
        else:
            self.upconv = nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2),
                                        nn.Conv2d(in_channels, out_channels, 1))
        self.conv1 = nn.Conv2d(2 * out_channels if merge_mode == "concat" else out_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)

    def forward(self, from_down, from_up):
        up = self.upconv(from_up)
        x = torch.cat((up, from_down), 1) if self.merge_mode == "concat" else up + from_down
        return F.relu(self.conv2(F.relu(self.conv1(x))))


class UNet(nn.Module):
    """Args as the reference (unet.py:146-148).  No activation after ``conv_final``."""

    def __init__(self, num_classes, in_channels=3, depth=4, start_filts=32, up_mode="transpose",
                 merge_mode="concat", **kwargs):
        super().__init__()
        if up_mode not in ("transpose", "upsample"):
            raise ValueError(f'"{up_mode}" is not a valid mode for upsampling. Only "transpose" and "upsample" are allowed.')
        if merge_mode not in ("concat", "add"):
            raise ValueError(f'"{merge_mode}" is not a valid mode for merging up and down paths. '
                             'Only "concat" and "add" are allowed.')
        if up_mode == "upsample" and merge_mode == "add":
            raise ValueError('up_mode "upsample" is incompatible with merge_mode "add"')
        self.num_classes, self.in_channels, self.start_filts, self.depth = num_classes, in_channels, start_filts, depth
        self.up_mode, self.merge_mode = up_mode, merge_mode
        widths = [start_filts * 2 ** i for i in range(depth)]
        self.down_convs = nn.ModuleList(
            DownConv(in_channels if i == 0 else widths[i - 1], widths[i], pooling=i < depth - 1) for i in range(depth))
        self.up_convs = nn.ModuleList(
            UpConv(widths[i], widths[i - 1], up_mode=up_mode, merge_mode=merge_mode) for i in range(depth - 1, 0, -1))
        self.conv_final = nn.Conv2d(widths[0], num_classes, 1)
        self.reset_params()

    def reset_params(self):
        # unet.py:200-209: Xavier-normal weights and zero biases for every Conv2d (transposed convs keep the default)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)

    def hip_supported(self, x):
        return (_HIP_UNET and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and self.up_mode == "transpose"
                and self.merge_mode == "concat" and ops.plane_unet_supported(self, x.shape[2], x.shape[3]))

    def _blob(self):
        """The packed weights, repacked when a parameter changed (storage or version)."""
        stamp = tuple((p.data_ptr(), p._version) for p in self.parameters())
        hit = getattr(self, "_blob_cache", None)
        if hit is None or hit[0] != stamp:
            hit = self._blob_cache = (stamp, ops.plane_unet_pack(self))
        return hit[1]

    def forward(self, x):
        if self.hip_supported(x):
            if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
                if _HIP_UNET_TRAIN:
                    return _PlaneUNetFn.apply(x, self, *self.parameters())
                return self.forward_modules(x)
            return ops.plane_unet_fwd(x, self, self._blob())
        return self.forward_modules(x)

    def forward_modules(self, x):
        """The nn.Conv2d modules one by one (host PyTorch-ROCm / MIOpen): shapes the HIP kernel does not cover."""
        skips = []
        for down in self.down_convs:
            x, skip = down(x)
            skips.append(skip)
        for i, up in enumerate(self.up_convs):
            x = up(skips[-(i + 2)], x)
        return self.conv_final(x)
