"""Parameter containers and host-PyTorch layers of the hot path.

``ResnetBlockFC`` keeps the reference's parameter names (fc_0, fc_1, shortcut;
reference src/layers.py:8-50) so checkpoints load unchanged.  Inside the decoder
its arithmetic runs in the fused HIP kernel (vtaco_amd/csrc/decode.hip); the
``forward`` here is the plain PyTorch-ROCm path used by the PointNet encoder's
tiny per-point MLP (SURVEY.md K4, 0.16 GFLOP/scene, stays host PyTorch).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import functional as F


class _TallLinear(torch.autograd.Function):
    """nn.Linear over a tall, skinny activation matrix ([tens of thousands of points] x [32..64 channels]) whose weight
    gradient dW = dy^T x is a 32 x 64 GEMM with K = 24 000: hipBLASLt runs that as one workgroup-starved kernel (0.5 ms,
    0.2 TFLOP/s -- the third largest item of the training step's profile).  The backward here splits K into batches
    (a batched GEMM of [S, out, N/S] x [S, N/S, in], then a sum over S): same arithmetic, all CUs busy."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        if ctx.needs_input_grad[0]:
            dx = (dy2 @ weight).view_as(x)
        if ctx.needs_input_grad[1]:
            n = x2.shape[0]
            s = next((k for k in (128, 96, 64, 48, 32, 24, 16, 8, 4, 2) if n % k == 0), 1)
            dw = torch.bmm(dy2.view(s, n // s, -1).transpose(1, 2), x2.view(s, n // s, -1)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0)
        return dx, dw, db


def tall_linear(lin, x):
    """``lin(x)``; under autograd on tall inputs (>= 4096 rows) with the split-K weight gradient of _TallLinear."""
    if torch.is_grad_enabled() and x.is_cuda and x.numel() // x.shape[-1] >= 4096 and (lin.weight.requires_grad or x.requires_grad):
        return _TallLinear.apply(x, lin.weight, lin.bias)
    return lin(x)


class ResnetBlockFC(nn.Module):
    """x -> shortcut(x) + fc_1(relu(fc_0(relu(x)))); fc_1.weight starts at zero."""

    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.shortcut = None if size_in == size_out else nn.Linear(size_in, size_out, bias=False)
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        dx = tall_linear(self.fc_1, F.relu(tall_linear(self.fc_0, F.relu(x))))
        return (x if self.shortcut is None else tall_linear(self.shortcut, x)) + dx

    def packed(self):
        """(fc_0.weight, fc_0.bias, fc_1.weight, fc_1.bias) for vt_decoder_pack."""
        return self.fc_0.weight, self.fc_0.bias, self.fc_1.weight, self.fc_1.bias


class _ConvPair(nn.Module):
    """Two 3x3 convs sharing ONE BatchNorm module, ReLU after each (the reference's
    DownConv / UpConv bodies, src/layers.py:246-319, reuse `self.bn` twice)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.bn = nn.BatchNorm2d(cout)

    def _pair(self, x):
        x = F.relu(self.bn(self.conv1(x)))
        return F.relu(self.bn(self.conv2(x)))


class DownConv(_ConvPair):
    def __init__(self, in_channels, out_channels, pooling=True):
        super().__init__(in_channels, out_channels)
        self.pooling = pooling
        if pooling:
            self.pool = nn.MaxPool2d(2, 2)

    def forward(self, x):
        skip = self._pair(x)
        return (self.pool(skip) if self.pooling else skip), skip


class UpConv(_ConvPair):
    def __init__(self, in_channels, out_channels):
        super().__init__(2 * out_channels, out_channels)
        self.upconv = nn.ConvTranspose2d(in_channels, out_channels, 2, stride=2)

    def forward(self, from_down, from_up):
        return self._pair(torch.cat((self.upconv(from_up), from_down), dim=1))


class TactileUNet(nn.Module):
    """Tactile depth estimator (reference ``UNet``, src/layers.py:322-450): `depth`
    DownConvs (last without pooling), depth-1 UpConvs (transpose-conv up, concat),
    1x1 conv, sigmoid.  Host PyTorch-ROCm."""

    def __init__(self, num_classes=1, in_channels=3, depth=4, start_filts=32, up_mode='transpose',
                 merge_mode='concat', **kwargs):
        super().__init__()
        if up_mode != 'transpose' or merge_mode != 'concat':
            raise ValueError("TactileUNet: only up_mode='transpose', merge_mode='concat' (the shipped config) are built")
        self.num_classes, self.in_channels, self.start_filts, self.depth = num_classes, in_channels, start_filts, depth
        widths = [start_filts * 2 ** i for i in range(depth)]
        self.down_convs = nn.ModuleList(
            DownConv(in_channels if i == 0 else widths[i - 1], w, pooling=i < depth - 1) for i, w in enumerate(widths))
        self.up_convs = nn.ModuleList(UpConv(widths[i], widths[i - 1]) for i in range(depth - 1, 0, -1))
        self.conv_final = nn.Conv2d(widths[0], num_classes, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        skips = []
        for down in self.down_convs:
            x, skip = down(x)
            skips.append(skip)
        for i, up in enumerate(self.up_convs):
            x = up(skips[-(i + 2)], x)
        return torch.sigmoid(self.conv_final(x)) * 1


def _bn_scenes(bn, x, scenes):
    """``bn(x)`` for a batch of ``scenes`` groups of images ordered image-major (image f of scene b at row f * scenes + b), with the
    statistics of every scene's images alone -- what the reference's per-scene calls compute (models/__init__.py:115-136: train-mode
    BatchNorm sees the five images of ONE scene) -- from one pass: in that order [F * S, C, H, W] is [F, S * C, H, W], a batch of F
    images with S * C channels.  The running statistics move as the S calls in scene order would move them."""
    if scenes <= 1 or not bn.training:
        return bn(x)                                                  # (eval mode: the running statistics, whatever the order)
    N, C, H, W = x.shape
    xv = x.contiguous().view(N // scenes, scenes * C, H, W)
    w = bn.weight.repeat(scenes) if bn.weight is not None else None
    b = bn.bias.repeat(scenes) if bn.bias is not None else None
    if bn.track_running_stats and bn.running_mean is not None:
        mean, var = x.new_zeros(scenes * C), x.new_ones(scenes * C)
        y = F.batch_norm(xv, mean, var, w, b, True, 1.0, bn.eps)      # momentum 1: the buffers receive the batch statistics themselves
        with torch.no_grad():
            bn.num_batches_tracked += scenes
            if bn.momentum is None:
                raise NotImplementedError("_bn_scenes: cumulative-average BatchNorm (momentum=None) is not on this path")
            m = float(bn.momentum)
            # (kept on the module: a tensor made from host values would be a copy inside a stream capture)
            key = (scenes, m, x.dtype, x.device)
            cache = bn.__dict__.setdefault("_scene_coef", {})
            coef = cache.get(key)
            if coef is None:
                coef = cache[key] = torch.tensor([m * (1.0 - m) ** (scenes - 1 - k) for k in range(scenes)], dtype=x.dtype, device=x.device)
            bn.running_mean.mul_((1.0 - m) ** scenes).add_((coef[:, None] * mean.view(scenes, C)).sum(0))
            bn.running_var.mul_((1.0 - m) ** scenes).add_((coef[:, None] * var.view(scenes, C)).sum(0))
    else:
        y = F.batch_norm(xv, None, None, w, b, True, 0.0, bn.eps)
    return y.view(N, C, H, W)


class _ResidualPair(nn.Module):
    """Two 3x3 conv + BatchNorm stages with an identity (or 1x1-projected) skip: the basic ResNet block
    (reference ``BasicBlock``, src/layers.py:52-82; parameter names conv1/bn1/conv2/bn2/downsample)."""
    expansion = 1

    def __init__(self, in_channel, out_channel, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channel, out_channel, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_channel)
        self.conv2 = nn.Conv2d(out_channel, out_channel, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channel)
        self.downsample = downsample

    def forward(self, x, scenes=1):
        if self.downsample is None:
            skip = x
        elif scenes > 1:
            skip = _bn_scenes(self.downsample[1], self.downsample[0](x), scenes)
        else:
            skip = self.downsample(x)
        y = _bn_scenes(self.bn2, self.conv2(F.relu(_bn_scenes(self.bn1, self.conv1(x), scenes))), scenes)
        return F.relu(y + skip)


class TactileResNet(nn.Module):
    """Tactile feature encoder of the shipped VTacO / VTacOH configs (``encoder_img: Resnet18``; reference ``ResNet`` with
    BasicBlocks, src/layers.py:127-195): 7x7/2 stem, 3x3/2 max-pool, four stages of residual pairs (64, 128, 256, 512; stride 2
    from the second), global average pool, Linear(512, 100), Linear(100, num_classes) -- no activation between the two.
    Host PyTorch-ROCm (MIOpen), like the tactile depth U-Net: five 320x240 images per scene."""

    def __init__(self, blocks_num=(2, 2, 2, 2), num_classes=32):
        super().__init__()
        self.in_channel = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, blocks_num[0], 1)
        self.layer2 = self._stage(128, blocks_num[1], 2)
        self.layer3 = self._stage(256, blocks_num[2], 2)
        self.layer4 = self._stage(512, blocks_num[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.linear = nn.Linear(512, 100)
        self.fc = nn.Linear(100, num_classes)
        for m in self.modules():                                       # src/layers.py:150-152
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _stage(self, channel, count, stride):
        project = None
        if stride != 1 or self.in_channel != channel:
            project = nn.Sequential(nn.Conv2d(self.in_channel, channel, 1, stride=stride, bias=False), nn.BatchNorm2d(channel))
        blocks = [_ResidualPair(self.in_channel, channel, stride=stride, downsample=project)]
        self.in_channel = channel
        blocks += [_ResidualPair(channel, channel) for _ in range(1, count)]
        return nn.Sequential(*blocks)

    def forward(self, x, scenes=1):
        x = self.maxpool(F.relu(_bn_scenes(self.bn1, self.conv1(x), scenes)))
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for block in stage:
                x = block(x, scenes)
        return self.fc(self.linear(torch.flatten(self.avgpool(x), 1)))

    def forward_scenes(self, imgs):
        """imgs [S, F, 3, H, W] -> [S, F, num_classes]: the reference's loop ``cat([self(imgs[s]) for s])`` (models/__init__.py:115-136)
        as ONE pass over the S * F images -- the convolutions at batch S * F (MIOpen at batch five runs at a third of that rate), every
        BatchNorm with the statistics of each scene's images alone (_bn_scenes), so values, gradients and running statistics are the
        loop's."""
        S, Fn = imgs.shape[:2]
        x = imgs.transpose(0, 1).reshape(Fn * S, *imgs.shape[2:])       # image-major
        return self.forward(x, scenes=S).view(Fn, S, -1).transpose(0, 1)


def Resnet18(num_classes=32):
    return TactileResNet((2, 2, 2, 2), num_classes=num_classes)


def Resnet34(num_classes=32):
    return TactileResNet((3, 4, 6, 3), num_classes=num_classes)
