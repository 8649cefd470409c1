"""3-D U-Net that refines the scattered feature grid.

Two paths over one parameter tree: inference (no grad) runs the HIP UNet3D of
``csrc/unet3d.hip`` through ``vt_unet3d_fwd`` (channels-last implicit-GEMM conv3d with fused
GroupNorm / ReLU / upsample / concat; ``precision`` selects split-bf16 or exact-f32 matrix
cores), training runs host PyTorch-ROCm autograd (MIOpen conv3d / group_norm; north_star
names no HIP kernel for the UNet3D backward).  Same parameter tree as the reference's ``UNet3D`` (src/encoder/unet3d.py:361-491:
``encoders.{i}.basic_module.SingleConv{1,2}.{groupnorm,conv}``, ``decoders.{i}...``,
``final_conv``) so checkpoints load unchanged; only what the shipped configs use is
built: DoubleConv blocks, layer order 'gcr', max-pool down, nearest-neighbour up +
concat, 1x1x1 final conv, no final activation (``testing=False``).
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops
from ..layers import _TallLinear


def _norm_conv_relu(cin, cout, order, num_groups):
    """One 'SingleConv': modules named by role, in the order the string gives."""
    if 'c' not in order or order[0] in 'rle':
        raise ValueError(f"bad layer order {order!r}")
    has_norm = ('g' in order) or ('b' in order)
    mods = OrderedDict()
    for pos, ch in enumerate(order):
        if ch == 'c':
            mods['conv'] = nn.Conv3d(cin, cout, 3, padding=1, bias=not has_norm)
        elif ch == 'g':
            n = cin if pos < order.index('c') else cout
            mods['groupnorm'] = nn.GroupNorm(num_groups if n >= num_groups else 1, n)
        elif ch == 'b':
            mods['batchnorm'] = nn.BatchNorm3d(cin if pos < order.index('c') else cout)
        elif ch == 'r':
            mods['ReLU'] = nn.ReLU(inplace=True)
        elif ch == 'l':
            mods['LeakyReLU'] = nn.LeakyReLU(0.1, inplace=True)
        elif ch == 'e':
            mods['ELU'] = nn.ELU(inplace=True)
        else:
            raise ValueError(f"unsupported layer type {ch!r}")
    return nn.Sequential(mods)


class _TwoConvs(nn.Sequential):
    def __init__(self, cin, cout, encoder, order, num_groups):
        if encoder:
            mid = max(cout // 2, cin)
            widths = [(cin, mid), (mid, cout)]
        else:
            widths = [(cin, cout), (cout, cout)]
        super().__init__(OrderedDict(
            (f'SingleConv{i + 1}', _norm_conv_relu(a, b, order, num_groups)) for i, (a, b) in enumerate(widths)))


class _Down(nn.Module):
    def __init__(self, cin, cout, pool, order, num_groups):
        super().__init__()
        self.pooling = nn.MaxPool3d(2) if pool else None
        self.basic_module = _TwoConvs(cin, cout, True, order, num_groups)

    def forward(self, x):
        return self.basic_module(x if self.pooling is None else self.pooling(x))


class _Up(nn.Module):
    def __init__(self, cin, cout, order, num_groups):
        super().__init__()
        self.basic_module = _TwoConvs(cin, cout, False, order, num_groups)

    def forward(self, skip, x):
        x = F.interpolate(x, size=skip.shape[2:], mode='nearest')
        return self.basic_module(torch.cat((skip, x), dim=1))


_WGRAD_F16 = os.environ.get("VTACO_UNET_WGRAD_PRECISION", "f16x3") != "f32"     # A/B knob: "f32" keeps the exact-f32 weight-gradient kernel


_UP_TRAIN = os.environ.get("VTACO_CONV_UP_TRAIN", os.environ.get("VTACO_CONV_UP", "1")) != "0"      # A/B knob: per-parity forward convs in training


class _MaskLink:
    """Between a 'gcr' layer and the ONE layer that reads its output (forward_channels_last_train wires them): the reader's
    GroupNorm backward leaves the gradient already masked by (output > 0), with its max |.| (ops.gn_bwd(mask_...)), and says so
    here; the layer then skips its own relu_mask pass."""
    __slots__ = ("ready", "gmax")

    def __init__(self):
        self.ready, self.gmax = False, None


_MASK_FUSE = os.environ.get("VTACO_UNET_MASK_FUSE", "1") != "0"     # A/B knob
_XSTATS = os.environ.get("VTACO_UNET_DGRAD_XSTATS", "1") != "0"      # A/B knob: GroupNorm-backward sums from the data-gradient conv's epilogue
_FIN_FWD = os.environ.get("VTACO_UNET_FIN_FWD", "1") != "0"         # A/B knob: the final conv's training forward in the last layer's epilogue
_FIN_FUSE = os.environ.get("VTACO_UNET_FIN_BWD", "1") != "0"        # A/B knob: the final 1x1x1 conv's backward + the last layer's mask in one pass
_WGRAD_SPARSE = os.environ.get("VTACO_UNET_WGRAD_SPARSE", "1") != "0"   # A/B knob: the first layer's weight gradient without the blocks whose input is zero


class _GcrFn(torch.autograd.Function):
    """One 'gcr' SingleConv on channels-last tensors through the C ABI, differentiable:
    forward vt_gn_scale_shift + vt_conv3d_gcr[_bf16x3]; backward vt_relu_mask, the forward conv
    kernels on the transposed/flipped weight (data gradient), vt_conv3d_wgrad and vt_gn_bwd.
    ``x_part`` / ``low_part`` are the producers' GroupNorm partial sums (not differentiated:
    vt_gn_bwd accounts for the statistics' dependence on x)."""

    @staticmethod
    def forward(ctx, x, low, gamma, beta, weight, x_part, low_part, groups, eps, precision, tile_flags=None, out_link=None,
                x_link=None, low_link=None, fin=None):
        B, D, H, W, C1 = x.shape
        C2 = low.shape[-1] if low is not None else 0
        Cout = weight.shape[0]
        x_st = (x_part, x_part.shape[1])
        low_st = (low_part, low_part.shape[1]) if low is not None else None
        ss = ops.gn_scale_shift(x_st, low_st, C1, C2, B, D * H * W, gamma, beta, groups, eps, x.device)
        split = ops.conv3d_pack(weight, "bf16x3") if precision == "bf16x3" else None
        # "f16x3": the forward conv on split-f16 operands where that kernel covers the shape (its inputs are GroupNorm outputs:
        # inside the half range, error at f32 rounding level)
        half = ops.conv3d_pack(weight, "f16x3") if precision == "f16x3" else None
        ctx.flags = None
        if fin is not None and half is not None and low is None and Cout == 32 and ops.final_fusable(x, Cout):
            # the last layer: the final 1x1x1 conv rides in its epilogue (fin = (weight [32,32], bias, stash)); y is kept as well, no
            # statistics (nothing normalises it)
            y, fin[2].out = ops.conv3d_gcr_final_keep(x, ss, half, ops.conv1x1_pack_f16x3(fin[0].detach()), fin[1].detach())
            part = x_part.new_empty(0)
        elif tile_flags is not None and half is not None and low is None and ops.conv3d_skip_covers(x, Cout):
            # the network's first layer on a mean grid: the blocks no point comes near are filled from the border-class constants
            y, (part, _) = ops.conv3d_gcr_skip(x, ss, half, Cout, tile_flags)
            ctx.flags = tile_flags                                          # (the weight gradient leaves the same blocks out)
        else:
            # a decoder entry [skip | upsample(low)]: the upsampled channels as a 2x2x2 conv per output parity class (forward only:
            # the data gradient reads a dense output gradient)
            up = (ops.conv3d_pack_up(weight, C1) if half is not None and low is not None and _UP_TRAIN
                  and ops.conv3d_up_covers(C1, C2, B, D, H, W, Cout) else None)
            y, (part, _) = ops.conv3d_gcr(x, low, ss, lambda: ops.conv3d_pack(weight), Cout, True, split, packed_w_f16x3=half,
                                          packed_w_up=up)                                  # f32 pack on demand
        ctx.save_for_backward(x, low, gamma, weight, ss, y, x_part, low_part)
        ctx.cfg = (groups, eps, precision)
        ctx.links = (out_link, x_link, low_link)
        ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    def backward(ctx, dy, _dpart):
        x, low, gamma, weight, ss, y, x_part, low_part = ctx.saved_tensors
        groups, eps, precision = ctx.cfg
        out_link, x_link, low_link = ctx.links
        if out_link is not None and out_link.ready:
            # the layer that reads y masked this gradient in its GroupNorm backward (and took its maximum)
            g, gmax = (dy if dy.is_contiguous() else dy.contiguous()), out_link.gmax
            out_link.ready, out_link.gmax = False, None
        elif precision == "f16x3":
            g, gmax = ops.relu_mask(dy, y, want_absmax=True)                # max |g| from the same pass (the kernels' power-of-two rescale)
        else:
            g, gmax = ops.relu_mask(dy, y), None
        # dxn = conv of g with the weight's channels swapped and taps flipped ([Cin,Cout,3,3,3]); the copy only where a kernel wants it
        w_tc = []

        def w_t():
            if not w_tc:
                w_tc.append(weight.flip(2, 3, 4).transpose(0, 1).contiguous())
            return w_tc[0]
        split = ops.conv3d_pack(w_t(), "bf16x3") if precision == "bf16x3" else None
        half = None
        if precision == "f16x3":
            # output gradients sit many orders of magnitude below the half range: the kernel scales them by a power of two
            # taken from their largest element before the split (exact), so the data gradient keeps f32-level accuracy
            half = ops.conv3d_pack_t(weight)
        # a plain layer on the split-f16 kernel: the GroupNorm backward's sums (sum dxn, sum dxn x) come out of the conv's epilogue
        fused = ops.conv3d_dgrad_xstats(g, half, weight.shape[1], gmax, x) if (_XSTATS and half is not None and low is None) else None
        if fused is not None:
            dxn, bpart = fused
        else:
            bpart = None
            dxn, _ = ops.conv3d_gcr(g, None, None, lambda: ops.conv3d_pack(w_t()), weight.shape[1], False, split, want_stats=False,
                                    packed_w_f16x3=half, in_absmax=gmax)
        # weight gradient: split-half operands as well (K = voxels; g under the same power-of-two rescale)
        dw = None
        if ctx.needs_input_grad[4]:
            # x zero over most blocks (the first layer): the taps over the other blocks + the GroupNorm shift's rank-one share
            if ctx.flags is not None and _WGRAD_SPARSE and _WGRAD_F16 and gmax is not None:
                dw = ops.conv3d_wgrad_sparse(x, ss, g, ctx.flags, g_absmax=gmax)
            if dw is None:
                dw = ops.conv3d_wgrad(x, low, ss, g, precision="f16x3" if precision == "f16x3" and _WGRAD_F16 else "f32", g_absmax=gmax)
        x_st = (x_part, x_part.shape[1])
        low_st = (low_part, low_part.shape[1]) if low is not None else None
        m_skip = x_link is not None and ctx.needs_input_grad[0]
        m_low = low_link is not None and low is not None and ctx.needs_input_grad[1]
        res = ops.gn_bwd(x, x_st, low, low_st, dxn, gamma, groups, eps, want_skip=ctx.needs_input_grad[0],
                         want_low=low is not None and ctx.needs_input_grad[1], mask_skip=m_skip, mask_low=m_low, bpart=bpart)
        dskip, dlow, dgamma, dbeta = res[:4]
        if m_skip:
            x_link.ready, x_link.gmax = True, res[4]
        if m_low:
            low_link.ready, low_link.gmax = True, res[5]
        return dskip, dlow, dgamma, dbeta, dw, None, None, None, None, None, None, None, None, None, None


class _FinStash:
    """The final conv's output when the last layer's launch computed it in its epilogue (handed from _GcrFn to _FinConvFn)."""
    out = None


class _FinConvFn(torch.autograd.Function):
    """The final 1x1x1 conv (32 -> 32) behind the last 'gcr' layer under autograd: forward = the matmul, backward = ONE pass
    (ops.conv1x1_bwd_masked) that leaves the last layer's MASKED output gradient with its maximum (handed over through ``link``,
    as a reader's GroupNorm backward does for the inner layers), dW and db."""

    @staticmethod
    def forward(ctx, y, weight, bias, link, stash):
        ctx.save_for_backward(y, weight)
        ctx.link = link
        out, stash.out = stash.out, None                              # the last layer's launch already left it (its epilogue)
        return out if out is not None else F.linear(y, weight, bias)

    @staticmethod
    def backward(ctx, dout):
        y, weight = ctx.saved_tensors
        g, gmax, dw, db = ops.conv1x1_bwd_masked(dout, y, weight, want_dw=ctx.needs_input_grad[1], want_db=ctx.needs_input_grad[2])
        ctx.link.ready, ctx.link.gmax = True, gmax
        return g, dw, db, None, None


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool3d_cl(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool3d_cl_bwd(x, dy)


class _PoolForkFn(torch.autograd.Function):
    """An encoder level's output y -> (y for the decoder's skip, maxpool(y) for the next level).  The two gradients meet here: one
    pass routes the pooled gradient to the windows' first maxima, adds the skip's and applies the (y > 0) mask of the layer that
    produced y (ops.maxpool3d_cl_bwd_fork), which that layer then skips (``link``)."""

    @staticmethod
    def forward(ctx, y, link):
        ctx.save_for_backward(y)
        ctx.link = link
        # (the pooled tensor's GroupNorm partial sums from the same pass: what channel_stats(pooled) would leave, bit for bit)
        pooled, (part, _) = ops.maxpool3d_cl_stats(y)
        ctx.mark_non_differentiable(part)
        return y.view_as(y), pooled, part

    @staticmethod
    def backward(ctx, dskip, dpooled, _dpart):
        (y,) = ctx.saved_tensors
        if dskip is None or dpooled is None:        # one branch unused: the plain forms
            dx = ops.maxpool3d_cl_bwd(y, dpooled) if dpooled is not None else dskip
            return dx, None
        g, gmax = ops.maxpool3d_cl_bwd_fork(y, dskip, dpooled)
        if ctx.link is not None:
            ctx.link.ready, ctx.link.gmax = True, gmax
            return g, None
        return g, None


class UNet3D(nn.Module):
    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order='gcr',
                 num_groups=8, num_levels=4, is_segmentation=True, testing=False, **kwargs):
        super().__init__()
        if isinstance(f_maps, int):
            f_maps = [f_maps * 2 ** k for k in range(num_levels)]
        self.encoders = nn.ModuleList(
            _Down(in_channels if i == 0 else f_maps[i - 1], f, i > 0, layer_order, num_groups)
            for i, f in enumerate(f_maps))
        rev = f_maps[::-1]
        self.decoders = nn.ModuleList(
            _Up(rev[i] + rev[i + 1], rev[i + 1], layer_order, num_groups) for i in range(len(rev) - 1))
        self.final_conv = nn.Conv3d(f_maps[0], out_channels, 1)
        self.testing = testing
        self.layer_order = layer_order
        self._pack_cache = {}
        # arithmetic of the 3x3x3 convolutions on the HIP inference path: "f16x3" = split-f16 operands in the persistent
        # double-buffered kernel on the 64^3 / 32^3-class levels (f32-rounding-level error) and split-bf16 thin tiles on
        # the 16^3-class levels; "bf16x3" = split-bf16 operands on all of those (6.6e-5 abs on the golden grid, 4e-5 on
        # the decoded logits); "f32" = exact-f32 matrix core everywhere.  Training (host autograd) is unaffected.
        self.precision = os.environ.get("VTACO_UNET_PRECISION", "f16x3")
        # the differentiable HIP path (forward_channels_last_train): "f16x3" (the default) = split-f16 forward, data-gradient
        # (input rescaled by a power of two) AND weight-gradient convolutions -- all three at f32 accumulation-order level against
        # the exact-f32 kernels (tests/test_unet3d_gpu.py; VTACO_UNET_WGRAD_PRECISION=f32 keeps the exact-f32 weight gradient);
        # "f32" = everything on the exact-f32 matrix core; "bf16x3" = split-bf16 forward and data gradient (at random init this
        # network's gradients move by ~1 % (L2) under a 1e-6 input perturbation -- ReLU / max-pool decisions -- and the 2e-5
        # deviations of the split-bf16 form flip more of them, ~2 %).  Settable per model: the constructor's ``train_precision`` /
        # ``precision`` keywords (``unet3d_kwargs`` in the config), then these attributes, then the environment.
        self.train_precision = os.environ.get("VTACO_UNET_TRAIN_PRECISION", "f16x3")
        # inference with precision "f16x3": the thin levels (16^3 / 8^3 of one scene) on IEEE-half pairs too (VTACO_UNET_THIN_HALF=0:
        # bf16 pairs there, as in rounds 1-3)
        self.thin_half = os.environ.get("VTACO_UNET_THIN_HALF", "1") != "0"
        for key in ("precision", "train_precision"):            # model-level setting (config: model.encoder_kwargs.unet3d_kwargs)
            if kwargs.get(key) is not None:
                if kwargs[key] not in ("f32", "bf16x3", "f16x3"):
                    raise ValueError(f"UNet3D: {key} must be 'f32', 'bf16x3' or 'f16x3' (got {kwargs[key]!r})")
                setattr(self, key, kwargs[key])
        self.final_activation = (nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)) if is_segmentation else None

    # ---- HIP inference path (channels-last, vt_conv3d_gcr) ------------------------------
    def hip_supported(self):
        """The HIP kernels cover the shipped configuration: 'gcr' blocks, max-pool, channel
        counts that are multiples of 32."""
        if self.layer_order != 'gcr':
            return False
        for m in self.modules():
            if isinstance(m, nn.Conv3d) and m.kernel_size == (3, 3, 3):
                if m.in_channels % 32 or m.out_channels % 32 or m.bias is not None:
                    return False
        return True

    def _packed(self, conv, precision="f32"):
        key = (id(conv), precision)
        stamp = (conv.weight.data_ptr(), conv.weight._version)
        hit = self._pack_cache.get(key)
        if hit is None or hit[0] != stamp:
            hit = (stamp, ops.conv3d_pack(conv.weight.detach(), precision=precision))
            self._pack_cache[key] = hit
        return hit[1]

    def _packed_up(self, conv, c_skip):
        """Merged class weights of a decoder-entry conv (ops.conv3d_pack_up), cached like _packed; None where not applicable
        (precision other than "f16x3", VTACO_CONV_UP=0, channel counts the kernel does not take)."""
        if self.precision != "f16x3" or os.environ.get("VTACO_CONV_UP", "1") == "0":
            return None
        key = (id(conv), "f16x3_up", c_skip)
        stamp = (conv.weight.data_ptr(), conv.weight._version)
        hit = self._pack_cache.get(key)
        if hit is None or hit[0] != stamp:
            hit = (stamp, ops.conv3d_pack_up(conv.weight.detach(), c_skip))
            self._pack_cache[key] = hit
        return hit[1]

    def _gcr(self, single, x, x_stats, low=None, low_stats=None):
        gn, conv = single.groupnorm, single.conv
        # "f16x3": IEEE-half pairs on every level -- the persistent kernels where they cover the shape, the thin-tile / K-split
        # kernels with half-pair fragments below (self.thin_half; bf16 pairs there were most of the encoder's drift)
        thin_half = self.precision == "f16x3" and self.thin_half
        split = self._packed(conv, "f16x3_thin" if thin_half else "bf16x3") if self.precision in ("bf16x3", "f16x3") else None
        half = self._packed(conv, "f16x3") if self.precision == "f16x3" else None
        up = self._packed_up(conv, x.shape[-1]) if low is not None else None
        return ops.gn_conv3d_relu(x, x_stats, low, low_stats, gn.weight.detach(), gn.bias.detach(), gn.num_groups,
                                  self._packed(conv), conv.out_channels, eps=gn.eps, relu=True, packed_w_bf16x3=split,
                                  packed_w_f16x3=half, thin_half=thin_half, packed_w_up=up)

    def _hip_params(self):
        """vt_unet3d_params for the current weights (re-packed only when a conv weight changed); the filled structure itself is
        kept while no parameter changed (stamps of the ~45 tensors: 15 us instead of 70 us of packing-cache lookups per encode)."""
        from .. import _lib
        tensors = getattr(self, "_prm_tensors", None)
        if tensors is None or tensors[0] is not self.final_conv.weight:
            tensors = self._prm_tensors = [self.final_conv.weight, self.final_conv.bias] + [
                t for blk in list(self.encoders) + list(self.decoders)
                for single in (blk.basic_module.SingleConv1, blk.basic_module.SingleConv2)
                for t in (single.groupnorm.weight, single.groupnorm.bias, single.conv.weight)]
        stamp = (self.precision, self.thin_half, os.environ.get("VTACO_UNET_FUSED_FINAL", "1"), os.environ.get("VTACO_CONV_UP", "1")) + tuple(
            (id(t), t.data_ptr(), t._version) for t in tensors if t is not None)
        hit = getattr(self, "_prm_cache", None)
        if hit is not None and hit[0] == stamp:
            return hit[1], hit[2]
        prm, keep = self._build_hip_params()
        self._prm_cache = (stamp, prm, keep)
        return prm, keep

    def _build_hip_params(self):
        from .. import _lib
        prm = _lib.UnetParams()
        keep = []

        def fill(dst, single, c_skip=None):
            gn, conv = single.groupnorm, single.conv
            tensors = (gn.weight.detach().contiguous(), gn.bias.detach().contiguous(), self._packed(conv))
            keep.extend(tensors)
            dst.gn_w, dst.gn_b, dst.packed = (t.data_ptr() for t in tensors)
            dst.cin, dst.cout = conv.in_channels, conv.out_channels
            if self.precision == "f16x3" and self.thin_half:
                thin = self._packed(conv, "f16x3_thin")
                keep.append(thin)
                dst.packed_f16x3_thin = thin.data_ptr()
            elif self.precision in ("bf16x3", "f16x3"):
                split = self._packed(conv, "bf16x3")
                keep.append(split)
                dst.packed_bf16x3 = split.data_ptr()
            if self.precision == "f16x3":
                half = self._packed(conv, "f16x3")
                keep.append(half)
                dst.packed_f16x3 = half.data_ptr()
                up = self._packed_up(conv, c_skip) if c_skip else None      # decoder entry: [skip | upsample(low)] in per-parity form
                if up is not None:
                    keep.append(up)
                    dst.packed_f16x3_up = up.data_ptr()
        prm.n_levels = len(self.encoders)
        first_gn = self.encoders[-1].basic_module.SingleConv1.groupnorm
        prm.groups, prm.eps = first_gn.num_groups, first_gn.eps
        for i, enc in enumerate(self.encoders):
            fill(prm.enc[i][0], enc.basic_module.SingleConv1)
            fill(prm.enc[i][1], enc.basic_module.SingleConv2)
        for k, dec in enumerate(self.decoders):
            skip_c = self.encoders[len(self.encoders) - 2 - k].basic_module.SingleConv2.conv.out_channels
            fill(prm.dec[k][0], dec.basic_module.SingleConv1, c_skip=skip_c)
            fill(prm.dec[k][1], dec.basic_module.SingleConv2)
        fw = self.final_conv.weight.detach().reshape(self.final_conv.out_channels, -1).contiguous()
        keep.append(fw)
        prm.final_w = fw.data_ptr()
        if self.final_conv.bias is not None:
            fb = self.final_conv.bias.detach().contiguous()
            keep.append(fb)
            prm.final_b = fb.data_ptr()
        prm.out_channels = self.final_conv.out_channels
        fp = self._final_packed()
        if fp is not None:
            keep.append(fp)
            prm.final_packed_f16x3 = fp.data_ptr()
        return prm, keep

    def _final_packed(self):
        """The final 1x1x1 conv's weight packed for the epilogue of the last 'gcr' layer (used where that layer runs on the
        specialised-wave split-f16 kernel; VTACO_UNET_FUSED_FINAL=0 keeps the separate launch), re-packed when it changes; None
        when the fusion does not apply."""
        w = self.final_conv.weight
        if self.precision != "f16x3" or tuple(w.shape[:2]) != (32, 32) or os.environ.get("VTACO_UNET_FUSED_FINAL", "1") == "0":
            return None
        key = ("final", "f16x3")
        stamp = (w.data_ptr(), w._version)
        hit = self._pack_cache.get(key)
        if hit is None or hit[0] != stamp:
            hit = (stamp, ops.conv1x1_pack_f16x3(w.detach().reshape(32, 32)))
            self._pack_cache[key] = hit
        return hit[1]

    def forward_channels_last(self, x, in_stats=None, tile_flags=None):
        """x [B,D,H,W,C] channels-last -> [B,D,H,W,out_channels]; inference only (no autograd).
        One C-ABI call (vt_unet3d_fwd) runs every launch of the network back to back.  ``in_stats`` = (part, nblk): the input's
        GroupNorm partial sums when its producer already has them (the one-launch PointNet MLP).  ``tile_flags``
        (ops.voxel_tile_flags): the 8^3 blocks over whose halo x is zero; the first layer skips their taps."""
        if x.shape[1] == x.shape[2] == x.shape[3]:
            prm, keep = self._hip_params()
            y = ops.unet3d_fwd(x.contiguous(), prm, keep, in_stats=in_stats, tile_flags=tile_flags)
            if self.testing and self.final_activation is not None:
                y = self.final_activation(y) if isinstance(self.final_activation, nn.Sigmoid) else torch.softmax(y, dim=-1)
            return y
        return self.forward_channels_last_layers(x)

    def forward_channels_last_layers(self, x):
        """Layer-by-layer variant of the same computation (non-cubic volumes, debugging).
        GroupNorm statistics are produced by whoever writes a tensor (conv epilogue / a stats pass
        for the input and the pooled tensors) and consumed by the next conv's prologue."""
        skips = []
        st = None
        for i, enc in enumerate(self.encoders):
            if i > 0:
                # pool + statistics in one pass where the statistics have their 1024 blocks (as vt_unet3d_fwd chooses)
                fused = x.shape[-1] % 32 == 0 and (x.shape[1] // 2) * (x.shape[2] // 2) * (x.shape[3] // 2) // 16 >= 1024
                x, st = ops.maxpool3d_cl_stats(x) if fused else (ops.maxpool3d_cl(x), None)
            if st is None:
                st = ops.channel_stats(x)
            x, st = self._gcr(enc.basic_module.SingleConv1, x, st)
            x, st = self._gcr(enc.basic_module.SingleConv2, x, st)
            skips.append((x, st))
        fused_final = self._final_packed()
        for k, (dec, (skip, skip_st)) in enumerate(zip(self.decoders, skips[-2::-1])):
            x, st = self._gcr(dec.basic_module.SingleConv1, skip, skip_st, low=x, low_stats=st)
            last = dec.basic_module.SingleConv2
            if (k + 1 == len(self.decoders) and fused_final is not None and self.final_conv.out_channels == 32
                    and ops.final_fusable(x, last.conv.out_channels)):
                # as vt_unet3d_fwd: the final 1x1x1 conv in the last layer's epilogue
                gn = last.groupnorm
                ss = ops.gn_scale_shift(st, None, x.shape[-1], 0, x.shape[0], x.shape[1] * x.shape[2] * x.shape[3], gn.weight.detach(),
                                        gn.bias.detach(), gn.num_groups, gn.eps, x.device)
                x = ops.conv3d_gcr_final(x, ss, self._packed(last.conv, "f16x3"), fused_final,
                                         self.final_conv.bias.detach().contiguous() if self.final_conv.bias is not None else None)
                break
            x, st = self._gcr(last, x, st)
        else:
            x = ops.conv1x1_cl(x, self.final_conv.weight.detach(), self.final_conv.bias.detach()
                               if self.final_conv.bias is not None else None)
        if self.testing and self.final_activation is not None:
            x = self.final_activation(x) if isinstance(self.final_activation, nn.Sigmoid) else torch.softmax(x, dim=-1)
        return x

    def forward_channels_last_train(self, x, tile_flags=None):
        """Differentiable channels-last forward on the HIP kernels (training): same layers as
        ``forward_channels_last_layers`` as autograd Functions whose backward is HIP too (data gradient
        = the forward conv kernels on flipped weights, vt_conv3d_wgrad, vt_gn_bwd, vt_maxpool3d_cl_bwd);
        the final 1x1x1 conv is a plain matmul.  x [B,D,H,W,C] contiguous."""
        def stats(t):
            return ops.channel_stats(t.detach())[0]

        def gcr(single, t, part, low=None, low_part=None, flags=None, out_link=None, x_link=None, low_link=None, fin=None):
            gn, conv = single.groupnorm, single.conv
            return _GcrFn.apply(t, low, gn.weight, gn.bias, conv.weight, part, low_part, gn.num_groups, gn.eps, self.train_precision, flags,
                                out_link, x_link, low_link, fin)

        def link():
            return _MaskLink() if _MASK_FUSE else None
        # (a layer whose output has ONE reader hands its relu_mask pass to that reader's GroupNorm backward: the first conv of every
        # DoubleConv, and the last conv of the bottom level and of every decoder level but the last, whose output is the `low` of the
        # next decoder's entry conv; an encoder level's output feeds the pool AND its skip: two gradients, summed by autograd first)
        skips = []
        part = None
        n_enc = len(self.encoders)
        low_link = None
        pooled = None
        for i, enc in enumerate(self.encoders):
            if i > 0:
                x = pooled if pooled is not None else _MaxPoolFn.apply(x)
            if pooled is not None:
                part = pooled_part
            elif i > 0 or part is None:
                part = stats(x)
            l12 = link()
            x, part = gcr(enc.basic_module.SingleConv1, x, part, flags=tile_flags if i == 0 else None, out_link=l12)
            last = i == n_enc - 1
            # the level's output: `low` of the first decoder (bottom level), else the pool's and the skip's input -- their two gradients
            # meet in _PoolForkFn, which also applies this layer's mask
            out_link = link() if (not last or len(self.decoders) > 0) else None
            x, part = gcr(enc.basic_module.SingleConv2, x, part, x_link=l12, out_link=out_link)
            low_link = out_link if last else None
            pooled = None
            if not last and _MASK_FUSE:
                x, pooled, pooled_part = _PoolForkFn.apply(x, out_link)
            skips.append((x, part))
        n_dec = len(self.decoders)
        for k, (dec, (skip, skip_part)) in enumerate(zip(self.decoders, skips[-2::-1])):
            l12 = link()
            x, part = gcr(dec.basic_module.SingleConv1, skip, skip_part, low=x, low_part=part, out_link=l12, low_link=low_link)
            # (the last layer's output has one reader too: the final conv, whose backward masks)
            fin = (k + 1 == n_dec and _FIN_FUSE and _MASK_FUSE and self.train_precision == "f16x3" and self.final_conv.bias is not None
                   and tuple(self.final_conv.weight.shape[:2]) == (32, 32) and x.shape[-1] == 32)
            low_link = link() if (k + 1 < n_dec or fin) else None
            stash = _FinStash() if fin else None
            w = self.final_conv.weight.reshape(self.final_conv.out_channels, -1)
            x, part = gcr(dec.basic_module.SingleConv2, x, part, x_link=l12, out_link=low_link,
                          fin=(w, self.final_conv.bias, stash) if fin and _FIN_FWD else None)
        w = self.final_conv.weight.reshape(self.final_conv.out_channels, -1)
        if n_dec and fin:
            x = _FinConvFn.apply(x, w, self.final_conv.bias, low_link, stash)
        else:
            # 2 M voxels x 32 channels: the weight gradient is a 32 x 32 GEMM with K = 2 M (hipBLASLt: one 2.6 ms kernel) -> split-K
            x = _TallLinear.apply(x, w, self.final_conv.bias) if x.numel() // x.shape[-1] >= 4096 else F.linear(x, w, self.final_conv.bias)
        if self.testing and self.final_activation is not None:
            x = self.final_activation(x) if isinstance(self.final_activation, nn.Sigmoid) else torch.softmax(x, dim=-1)
        return x

    def forward(self, x):
        skips = []
        for enc in self.encoders:
            x = enc(x)
            skips.append(x)
        for dec, skip in zip(self.decoders, skips[-2::-1]):
            x = dec(skip, x)
        x = self.final_conv(x)
        if self.testing and self.final_activation is not None:
            x = self.final_activation(x)
        return x
