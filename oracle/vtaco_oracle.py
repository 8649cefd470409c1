"""CPU oracle for the VTacO occupancy hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vtaco_amd/`` may import this file;
it is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

It restates, as plain functions over a ``state_dict`` (same key names as the
reference checkpoint), the arithmetic of the reference's
hot path.  All maths is float32 on the CPU (torch CPU tensors), every function
cites the reference file:line it follows.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
golden vectors produced by importing the *real* reference in the build
container (``tests/golden/make_goldens.py``).  Marching cubes lives in
``oracle/mc_lewiner.c`` and is pinned against scikit-image 0.18.3 outputs.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# coordinate helpers
# --------------------------------------------------------------------------


def make_3d_grid(bb_min, bb_max, shape):
    """Query lattice, axis 0 slowest / axis 2 fastest (src/common.py:178-197)."""
    ax = [torch.linspace(bb_min[k], bb_max[k], shape[k]) for k in range(3)]
    gx, gy, gz = torch.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
    return torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], dim=1)


def normalize_3d_coordinate(p, padding=0.1):
    """[-0.55,0.55] -> [0,1) with clamps (src/common.py:293-309).

    Divisor is 1 + padding + 10e-4 (=1.101); values >= 1 become 1 - 10e-4,
    values < 0 become 0.  (The reference mutates a clone in place; this is the
    functional form.)
    """
    q = p / (1 + padding + 10e-4) + 0.5
    q = torch.where(q >= 1, torch.full_like(q, 1 - 10e-4), q)
    q = torch.where(q < 0, torch.zeros_like(q), q)
    return q


def coordinate2index_3d(x, reso):
    """Voxel id = ix + R*(iy + R*iz), truncating cast (src/common.py:333-348)."""
    xi = (x * reso).long()
    return xi[..., 0] + reso * (xi[..., 1] + reso * xi[..., 2])


def voxel_index(p, reso, padding=0.1):
    """K1: points -> voxel ids (src/encoder/pointnet.py:151-152)."""
    return coordinate2index_3d(normalize_3d_coordinate(p, padding), reso)


# --------------------------------------------------------------------------
# trilinear gather (K6)
# --------------------------------------------------------------------------


def nearest_sample(grid, p, padding=0.1):
    """``F.grid_sample(c, 2*p_nor-1, padding_mode='border', align_corners=True, mode='nearest')`` (decoder.py:62-68 with
    ``sample_mode='nearest'``): the voxel at the coordinate rounded half-to-even (ATen's nearbyint) after the border clip.
    Returns [B,N,C]."""
    B, C, D, H, W = grid.shape
    v = 2.0 * normalize_3d_coordinate(p.float(), padding) - 1.0

    def index(coord, size):
        return torch.round(torch.clamp(((coord + 1.0) / 2) * (size - 1), 0, size - 1)).long()   # torch.round: half to even

    lin = (index(v[..., 2], D) * H + index(v[..., 1], H)) * W + index(v[..., 0], W)
    gcl = grid.permute(0, 2, 3, 4, 1).reshape(B, D * H * W, C)
    return torch.gather(gcl, 1, lin.unsqueeze(-1).expand(-1, -1, C))


def _sample(grid, p, padding, sample_mode):
    return nearest_sample(grid, p, padding) if sample_mode == "nearest" else trilinear_sample(grid, p, padding)


def trilinear_sample(grid, p, padding=0.1):
    """Trilinear interpolation of ``grid`` [B,C,D,H,W] at points ``p`` [B,N,3].

    Restates ``LocalDecoder.sample_grid_feature`` (decoder.py:62-68), i.e.
    ``F.grid_sample(c, 2*p_nor-1, padding_mode='border', align_corners=True)``
    as the explicit 8-corner formula: x <-> W (last dim), y <-> H, z <-> D.
    Returns [B,N,C].
    """
    B, C, D, H, W = grid.shape
    pn = normalize_3d_coordinate(p.float(), padding)
    v = 2.0 * pn - 1.0

    def unnorm(coord, size):
        f = ((coord + 1.0) / 2) * (size - 1)
        return torch.clamp(f, 0, size - 1)

    fx, fy, fz = unnorm(v[..., 0], W), unnorm(v[..., 1], H), unnorm(v[..., 2], D)
    x0, y0, z0 = torch.floor(fx), torch.floor(fy), torch.floor(fz)
    tx, ty, tz = fx - x0, fy - y0, fz - z0          # weight of the +1 corner
    ux, uy, uz = (x0 + 1) - fx, (y0 + 1) - fy, (z0 + 1) - fz
    x0, y0, z0 = x0.long(), y0.long(), z0.long()
    gcl = grid.permute(0, 2, 3, 4, 1).reshape(B, D * H * W, C)
    out = torch.zeros(B, p.shape[1], C, dtype=grid.dtype)
    for dz, wz in ((0, uz), (1, tz)):
        for dy, wy in ((0, uy), (1, ty)):
            for dx, wx in ((0, ux), (1, tx)):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                inb = (xi <= W - 1) & (yi <= H - 1) & (zi <= D - 1)
                lin = (zi.clamp(max=D - 1) * H + yi.clamp(max=H - 1)) * W + xi.clamp(max=W - 1)
                vals = torch.gather(gcl, 1, lin.unsqueeze(-1).expand(-1, -1, C))
                w = ((wx * wy * wz) * inb.to(wx.dtype)).to(grid.dtype)   # corner weights are f32 arithmetic by definition
                out = out + vals * w.unsqueeze(-1)
    return out


# --------------------------------------------------------------------------
# per-point MLPs
# --------------------------------------------------------------------------


def _lin(sd, name, x, bias=True):
    y = x @ sd[name + ".weight"].t()
    if bias and (name + ".bias") in sd:
        y = y + sd[name + ".bias"]
    return y


def resnet_block_fc(sd, prefix, x):
    """``ResnetBlockFC.forward`` (src/layers.py:41-50)."""
    net = _lin(sd, prefix + ".fc_0", F.relu(x))
    dx = _lin(sd, prefix + ".fc_1", F.relu(net))
    if (prefix + ".shortcut.weight") in sd:
        xs = x @ sd[prefix + ".shortcut.weight"].t()
    else:
        xs = x
    return xs + dx


def _n_blocks(sd, prefix="blocks."):
    n = 0
    while f"{prefix}{n}.fc_0.weight" in sd:
        n += 1
    return n


def decoder_mlp(sd, net, c):
    """Conditioned ResNet MLP + head shared by all decoder variants
    (decoder.py:152-159)."""
    for i in range(_n_blocks(sd)):
        net = net + _lin(sd, f"fc_c.{i}", c)
        net = resnet_block_fc(sd, f"blocks.{i}", net)
    return net


def _head_actvn(x, leaky):
    """``LocalDecoder.actvn`` (decoder.py:46-49): relu, or leaky_relu(0.2) with ``leaky`` -- applied in front of the output heads
    only; the ResnetBlockFC activations are nn.ReLU whatever ``leaky`` says (layers.py:33)."""
    return F.leaky_relu(x, 0.2) if leaky else F.relu(x)


def local_decoder_forward(sd, p, grid, padding=0.1, leaky=False, sample_mode="bilinear"):
    """``LocalDecoder.forward`` (decoder.py:135-161): logits [B,N]."""
    c = _sample(grid, p, padding, sample_mode)
    net = decoder_mlp(sd, _lin(sd, "fc_p", p.to(sd["fc_p.weight"].dtype)), c)
    return _lin(sd, "fc_out", _head_actvn(net, leaky)).squeeze(-1)


def local_decoder_forward_img(sd, p, grid, c_img, padding=0.1, leaky=False, sample_mode="bilinear"):
    """``LocalDecoder.forward_img`` (decoder.py:71-103): tactile concat."""
    c = _sample(grid, p, padding, sample_mode)
    net = _lin(sd, "fc_p_img", torch.cat((p.to(c_img.dtype), c_img), dim=2))
    net = decoder_mlp(sd, net, c)
    return _lin(sd, "fc_out", _head_actvn(net, leaky)).squeeze(-1)


def local_decoder_forward_contact(sd, p, grid, padding=0.1, leaky=False, sample_mode="bilinear"):
    """``LocalDecoder.forward_contact`` (decoder.py:105-133)."""
    c = _sample(grid, p, padding, sample_mode)
    net = decoder_mlp(sd, _lin(sd, "fc_p", p.to(sd["fc_p.weight"].dtype)), c)
    a = _head_actvn(net, leaky)
    return _lin(sd, "fc_out", a).squeeze(-1), _lin(sd, "fc_out_contact", a).squeeze(-1)


# --------------------------------------------------------------------------
# TransformerFusion (K8)
# --------------------------------------------------------------------------


def _relation_unit(sd, pre, q, k, v):
    """``RelationUnit.forward`` (src/TransformerFusion.py:92-113); tensors are
    [B,N,C] here (the reference permutes to N,B,C and back)."""
    wk = F.normalize(k @ sd[pre + ".WK.weight"].t(), p=2, dim=-1)
    wq = F.normalize(q @ sd[pre + ".WQ.weight"].t(), p=2, dim=-1)
    dot = torch.bmm(wq, wk.transpose(1, 2))                  # B, Nq, Nk
    aff = F.softmax(dot, dim=-1)
    aff = aff / (1e-9 + aff.sum(dim=1, keepdim=True))        # column re-norm (:104)
    out = torch.bmm(aff, v @ sd[pre + ".WV.weight"].t())
    return F.relu((q - out) @ sd[pre + ".trans_conv.weight"].t())


def _trans_nonlinear(sd, pre, x, masks=None):
    """``TransNonlinear.forward`` (TransformerFusion.py:21-25).  Eval mode by default; ``masks = (m1 [..,64], m2 [..,32])``
    are the factors (0 or 1/(1-p)) of its two train-mode dropouts (:13-19), given explicitly so that a test can replay
    the masks another implementation drew."""
    h = F.relu(_lin(sd, pre + ".linear1", x))
    if masks is not None:
        h = h * masks[0]
    y = _lin(sd, pre + ".linear2", h)
    if masks is not None:
        y = y * masks[1]
    x = x + y
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"], 1e-5)


def _mha1(sd, pre, q, k, v, masks=None):
    """Single-head ``MultiheadAttention`` (TransformerFusion.py:42-62)."""
    return _trans_nonlinear(sd, pre + ".extra_nonlinear.0", _relation_unit(sd, pre + ".head.0", q, k, v), masks)


def _inorm_relu(x):
    """InstanceNorm1d over the N axis (no affine, biased var, eps 1e-5) + ReLU
    (TransformerFusion.py:144-145, 209-210, 216-218)."""
    m = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True)
    return F.relu((x - m) / torch.sqrt(var + 1e-5))


def transformer_fusion(sd, c_img, c, masks=None):
    """``TransformerFusion.forward(search=c_img, template=c)`` with
    ``num_layers=1, with_pos_embed=False`` (TransformerFusion.py:311-333); eval mode unless ``masks`` (three pairs of
    dropout factors: encoder self-attention, decoder self-attention, cross-attention) is given.
    ``sd`` keys are relative to ``fuser.``."""
    masks = masks or (None, None, None)
    sa = "encoder.layers.0.self_attn"                       # shared with decoder's self_attn
    mem = _inorm_relu(c + _mha1(sd, sa, c, c, c, masks[0]))
    sa_d = "decoder.layers.0.self_attn"
    tgt = _inorm_relu(c_img + _mha1(sd, sa_d, c_img, c_img, c_img, masks[1]))
    ca = "decoder.layers.0.cross_attn"
    return _inorm_relu(tgt + _mha1(sd, ca, tgt, mem, mem, masks[2]))


def attention_decoder_forward_img(sd, p, grid, c_img, padding=0.1, leaky=False):
    """``AttentionDecoder.forward_img`` (decoder.py:237-271; ``leaky``: decoder.py:210-213 -- the head's activation only)."""
    c = trilinear_sample(grid, p, padding)
    fsd = {k[len("fuser."):]: v for k, v in sd.items() if k.startswith("fuser.")}
    c = transformer_fusion(fsd, c_img, c)
    net = decoder_mlp(sd, _lin(sd, "fc_p", p.to(sd["fc_p.weight"].dtype)), c)
    return _lin(sd, "fc_out", _head_actvn(net, leaky)).squeeze(-1)


# --------------------------------------------------------------------------
# PointNet local-pool encoder (K1-K4) and UNet3D (K5)
# --------------------------------------------------------------------------


def segment_pool_max(feat, index):
    """``pool_local`` for the 'grid' key (pointnet.py:116-132): per-voxel
    channel-wise max gathered back to the points.  feat [B,T,C], index [B,T]."""
    out = torch.empty_like(feat)
    for b in range(feat.shape[0]):
        uniq, inv = torch.unique(index[b], return_inverse=True)
        seg = torch.full((uniq.numel(), feat.shape[2]), -float("inf"), dtype=feat.dtype)
        seg = seg.scatter_reduce(0, inv.unsqueeze(-1).expand_as(feat[b]), feat[b], "amax", include_self=True)
        out[b] = seg[inv]
    return out


def segment_pool_mean(feat, index):
    """``pool_local`` with ``scatter_type='mean'`` (pointnet.py:64-69, 116-132: torch_scatter.scatter_mean over the cells,
    gathered back to the points).  feat [B,T,C], index [B,T]."""
    out = torch.empty_like(feat)
    for b in range(feat.shape[0]):
        uniq, inv = torch.unique(index[b], return_inverse=True)
        acc = torch.zeros((uniq.numel(), feat.shape[2]), dtype=feat.dtype).index_add_(0, inv, feat[b])
        cnt = torch.zeros(uniq.numel(), dtype=feat.dtype).index_add_(0, inv, torch.ones(index.shape[1], dtype=feat.dtype))
        out[b] = (acc / cnt.unsqueeze(-1))[inv]
    return out


def scatter_mean_grid(feat, index, reso):
    """``generate_grid_features`` scatter part (pointnet.py:102-110):
    per-voxel mean, empty voxels 0; returns [B,C,R,R,R] (dims z,y,x)."""
    B, T, C = feat.shape
    grid = torch.zeros(B, reso ** 3, C, dtype=feat.dtype)
    cnt = torch.zeros(B, reso ** 3, 1, dtype=feat.dtype)
    grid.scatter_add_(1, index.unsqueeze(-1).expand(-1, -1, C), feat)
    cnt.scatter_add_(1, index.unsqueeze(-1), torch.ones(B, T, 1, dtype=feat.dtype))
    grid = grid / cnt.clamp(min=1)
    return grid.permute(0, 2, 1).reshape(B, C, reso, reso, reso)


def pointnet_point_features(sd, p, reso, padding=0.1, return_stages=False, scatter_type="max"):
    """``LocalPoolPointnet.forward`` up to ``fc_c`` (pointnet.py:135-162).  The voxel ids are f32 arithmetic by
    definition (a float64 run of the oracle -- the yardstick tests measure f32 rounding noise against -- keeps them)."""
    idx = voxel_index(p.float(), reso, padding)
    net = _lin(sd, "fc_pos", p)
    net = resnet_block_fc(sd, "blocks.0", net)
    stages = [net]
    for i in range(1, _n_blocks(sd)):
        pooled = segment_pool_max(net, idx) if scatter_type == "max" else segment_pool_mean(net, idx)
        net = resnet_block_fc(sd, f"blocks.{i}", torch.cat([net, pooled], dim=2))
        stages.append(net)
    c = _lin(sd, "fc_c", net)
    if return_stages:
        return c, idx, stages
    return c, idx


def _gcr(sd, pre, x, groups=8):
    """SingleConv order 'gcr': GroupNorm -> Conv3d(no bias) -> ReLU
    (unet3d.py:20-72)."""
    ch = x.shape[1]
    g = groups if ch >= groups else 1
    x = F.group_norm(x, g, sd[pre + ".groupnorm.weight"], sd[pre + ".groupnorm.bias"], 1e-5)
    # .contiguous(): a state_dict taken from a module kept in channels_last_3d hands F.conv3d strided weights, and torch's CPU
    # conv then takes a different (and, measured against a float64 run at R = 64 / 4 levels, 20-300x less accurate, thread-count
    # dependent) path than the reference's contiguous nn.Conv3d weights do; with contiguous weights this IS the reference's op
    x = F.conv3d(x, sd[pre + ".conv.weight"].contiguous(), None, padding=1)
    return F.relu(x)


def unet3d_forward(sd, x):
    """``UNet3D.forward`` (unet3d.py:449-474), DoubleConv blocks, nearest
    upsample + concat; final sigmoid NOT applied (testing=False)."""
    n_enc = 0
    while f"encoders.{n_enc}.basic_module.SingleConv1.conv.weight" in sd:
        n_enc += 1
    feats = []
    for i in range(n_enc):
        if i > 0:
            x = F.max_pool3d(x, 2)
        x = _gcr(sd, f"encoders.{i}.basic_module.SingleConv1", x)
        x = _gcr(sd, f"encoders.{i}.basic_module.SingleConv2", x)
        feats.insert(0, x)
    for i, skip in enumerate(feats[1:]):
        x = F.interpolate(x, size=skip.shape[2:], mode="nearest")
        x = torch.cat((skip, x), dim=1)
        x = _gcr(sd, f"decoders.{i}.basic_module.SingleConv1", x)
        x = _gcr(sd, f"decoders.{i}.basic_module.SingleConv2", x)
    return F.conv3d(x, sd["final_conv.weight"].contiguous(), sd["final_conv.bias"])


def pointnet_encoder_forward(sd, p, reso, padding=0.1, unet3d=True):
    """Full ``LocalPoolPointnet.forward`` with plane_type='grid'
    (pointnet.py:135-166)."""
    c, idx = pointnet_point_features(sd, p, reso, padding)
    grid = scatter_mean_grid(c, idx, reso)
    if unet3d:
        usd = {k[len("unet3d."):]: v for k, v in sd.items() if k.startswith("unet3d.")}
        grid = unet3d_forward(usd, grid)
    return grid


# --------------------------------------------------------------------------
# tactile UNet depth estimator (K9)
# --------------------------------------------------------------------------


def _bn2d(sd, pre, x, training):
    if training:
        return F.batch_norm(x, None, None, sd[pre + ".weight"], sd[pre + ".bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"],
                        sd[pre + ".weight"], sd[pre + ".bias"], False, 0.1, 1e-5)


def tactile_unet_forward(sd, x, training=False):
    """Tactile ``UNet.forward`` (src/layers.py:430-450) with DownConv/UpConv
    (:246-319): ONE BatchNorm module is applied after both convs of a block."""
    depth = 0
    while f"down_convs.{depth}.conv1.weight" in sd:
        depth += 1
    skips = []
    for i in range(depth):
        pre = f"down_convs.{i}"
        x = F.relu(_bn2d(sd, pre + ".bn", F.conv2d(x, sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"], padding=1), training))
        x = F.relu(_bn2d(sd, pre + ".bn", F.conv2d(x, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1), training))
        skips.append(x)
        if i < depth - 1:
            x = F.max_pool2d(x, 2, 2)
    for i in range(depth - 1):
        pre = f"up_convs.{i}"
        up = F.conv_transpose2d(x, sd[pre + ".upconv.weight"], sd[pre + ".upconv.bias"], stride=2)
        x = torch.cat((up, skips[-(i + 2)]), dim=1)
        x = F.relu(_bn2d(sd, pre + ".bn", F.conv2d(x, sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"], padding=1), training))
        x = F.relu(_bn2d(sd, pre + ".bn", F.conv2d(x, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1), training))
    x = F.conv2d(x, sd["conv_final.weight"], sd["conv_final.bias"])
    return torch.sigmoid(x) * 1


# --------------------------------------------------------------------------
# dense evaluation (A11) and mesh post-processing (A12)
# --------------------------------------------------------------------------


def eval_points_dense(sd, grid, nx, padding=0.1, chunk=100000):
    """``Generator3D.eval_points`` over the ``generate_obj_mesh_wnf`` lattice
    (generation.py:119-120,155-157,338-383): 100k-point chunks, logits [nx^3]."""
    pts = (1 + padding) * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
    outs = []
    for pi in torch.split(pts, chunk):
        outs.append(local_decoder_forward(sd, pi.unsqueeze(0), grid, padding).squeeze(0))
    return torch.cat(outs, dim=0)


def mesh_rescale(verts, nx, padding=0.1):
    """``vertices -= nx/2; vertices *= 1.1/nx`` (generation.py:271-272) in f32."""
    import numpy as np
    v = verts.astype(np.float32) - np.array([nx / 2] * 3, dtype=np.float32)
    return v * np.float32(1.1 / nx) if padding == 0.1 else v * np.float32((1 + padding) / nx)


def mc_level(vol):
    """skimage's default iso level ``0.5 * (volume.min() + volume.max())``
    (skimage/measure/_marching_cubes_lewiner.py, 0.18.3): the sum is rounded
    in float32 (two np.float32 scalars), the halving is exact."""
    import numpy as np
    return 0.5 * float(np.float32(vol.min()) + np.float32(vol.max()))


# --------------------------------------------------------------------------
# tactile feature assignment (K11)
# --------------------------------------------------------------------------


def tactile_assign_nearest(pts, tips, success, radius=0.05):
    """VTacOH rule (generation.py:186-200): finger = argmin distance to the 5 fingertips, kept if
    that distance < radius and that finger's touch succeeded.  float64 distances like scipy's cdist.
    Returns int64 ids [N], 255 = none."""
    import numpy as np
    p = np.asarray(pts, dtype=np.float64)
    t = np.asarray(tips, dtype=np.float64)
    d = np.sqrt(((p[:, None, :] - t[None, :, :]) ** 2).sum(-1))
    ids = np.full(len(p), 255, dtype=np.int64)
    for f in range(len(t)):
        if success[f]:
            sel = (d.min(1) < radius) & (d.argmin(1) == f)
            ids[sel] = f
    return ids


def tactile_assign_within(pts, clouds, counts, success, radius=0.015):
    """VTacO rule (generation.py:245-255): a point within radius of any contact point of finger t
    takes finger t; fingers in ascending order, later ones overwrite."""
    import numpy as np
    p = np.asarray(pts, dtype=np.float64)
    ids = np.full(len(p), 255, dtype=np.int64)
    for f in range(len(clouds)):
        if not success[f] or counts[f] == 0:
            continue
        c = np.asarray(clouds[f][:counts[f]], dtype=np.float64)
        d = np.sqrt(((c[:, None, :] - p[None, :, :]) ** 2).sum(-1))
        ids[(d < radius).any(0)] = f
    return ids


# --------------------------------------------------------------------------
# hand branch: plane-mode PointNet, 2-D UNet, MANO head
# --------------------------------------------------------------------------

PLANE_AXES = {"xz": (0, 2), "xy": (0, 1), "yz": (1, 2)}


def normalize_coordinate(p, padding=0.1, plane="xz"):
    """Plane projection -> [0,1) (src/common.py:268-291): divisor 1 + padding + 10e-6,
    values >= 1 become 1 - 10e-6 (NOT the 3-D variant's 10e-4 constants)."""
    a, b = PLANE_AXES[plane]
    q = torch.stack([p[..., a], p[..., b]], dim=-1) / (1 + padding + 10e-6) + 0.5
    q = torch.where(q >= 1, torch.full_like(q, 1 - 10e-6), q)
    q = torch.where(q < 0, torch.zeros_like(q), q)
    return q


def plane_index(p, reso, padding=0.1, plane="xz"):
    """``coordinate2index(.., '2d')`` (src/common.py:333-345): first projected axis fastest."""
    xi = (normalize_coordinate(p, padding, plane) * reso).long()
    return xi[..., 0] + reso * xi[..., 1]


def scatter_mean_plane(feat, index, reso):
    """``generate_plane_features`` scatter part (pointnet.py:85-95): per-cell mean, [B,C,R,R]."""
    B, T, C = feat.shape
    acc = torch.zeros(B, reso * reso, C)
    cnt = torch.zeros(B, reso * reso, 1)
    acc.scatter_add_(1, index.unsqueeze(-1).expand(-1, -1, C), feat)
    cnt.scatter_add_(1, index.unsqueeze(-1), torch.ones(B, T, 1))
    return (acc / cnt.clamp(min=1)).permute(0, 2, 1).reshape(B, C, reso, reso)


def unet2d_forward(sd, x):
    """``UNet.forward`` (src/encoder/unet.py:218-233): depth x {conv3x3+ReLU twice, maxpool except last},
    depth-1 x {ConvTranspose2d 2x2 s2, cat(up, skip), conv3x3+ReLU twice}, conv1x1.  merge 'concat'."""
    depth = 0
    while f"down_convs.{depth}.conv1.weight" in sd:
        depth += 1
    skips = []
    for i in range(depth):
        pre = f"down_convs.{i}."
        x = F.relu(F.conv2d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=1))
        x = F.relu(F.conv2d(x, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], padding=1))
        skips.append(x)
        if i < depth - 1:
            x = F.max_pool2d(x, 2)
    for i in range(depth - 1):
        pre = f"up_convs.{i}."
        up = F.conv_transpose2d(x, sd[pre + "upconv.weight"], sd[pre + "upconv.bias"], stride=2)
        x = torch.cat((up, skips[-(i + 2)]), dim=1)
        x = F.relu(F.conv2d(x, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=1))
        x = F.relu(F.conv2d(x, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], padding=1))
    return F.conv2d(x, sd["conv_final.weight"], sd["conv_final.bias"])


def plane_pointnet_forward(sd, p, reso, padding=0.1, planes=("xz", "xy", "yz"), return_stages=False, scatter_type="max"):
    """``LocalPoolPointnet.forward`` with plane_type=['xz','xy','yz'] (pointnet.py:135-176):
    pool_local SUMS the per-plane max-pools (:116-132); one scatter-mean plane (+ shared UNet) per key.
    Returns {plane: [B,C,R,R]} in the reference's dict order (xz, xy, yz)."""
    idx = {k: plane_index(p, reso, padding, k) for k in planes}
    net = _lin(sd, "fc_pos", p)
    net = resnet_block_fc(sd, "blocks.0", net)
    stages = [net]
    for i in range(1, _n_blocks(sd)):
        pool = segment_pool_max if scatter_type == "max" else segment_pool_mean
        pooled = sum(pool(net, idx[k]) for k in planes)
        net = resnet_block_fc(sd, f"blocks.{i}", torch.cat([net, pooled], dim=2))
        stages.append(net)
    c = _lin(sd, "fc_c", net)
    usd = {k[len("unet."):]: v for k, v in sd.items() if k.startswith("unet.")}
    fea = {}
    for k in planes:
        f = scatter_mean_plane(c, idx[k], reso)
        fea[k] = unet2d_forward(usd, f) if usd else f
    if return_stages:
        return fea, idx, stages, c
    return fea


def mano_param_head(sd, fea):
    """out_mano head (pointnet.py:179-192): channel-concat the planes, global average pool, fc_mano."""
    cat = torch.cat([fea[k] for k in fea], dim=1)
    return _lin(sd, "fc_mano", cat.mean(dim=(2, 3)))


def rodrigues(axisang):
    """manopth batch_rodrigues (src/encoder/manopth/rodrigues_layer.py:49-60 with quat2mat :15-46):
    angle = ||v + 1e-8||, quaternion (cos a/2, sin a/2 * v/angle) re-normalised, then the rotation matrix."""
    angle = torch.norm(axisang + 1e-8, p=2, dim=1, keepdim=True)
    axis = axisang / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]        # three joints per finger off the wrist
MANO_TIPS_RIGHT = [745, 317, 444, 556, 673]
MANO_TIPS_LEFT = [745, 317, 445, 556, 673]            # manolayer.py:327-330: the left hand's middle-finger tip vertex
MANO_JOINT_ORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def mano_forward(model, pose, center_idx=9, side="right"):
    """``ManoLayer.forward`` (src/encoder/manolayer.py:160-364) for the shipped configuration
    (axis-angle root + joints, use_pca False, flat_hand_mean False, betas = the model's zeros,
    th_trans = 0; ``side`` picks the tip vertices, manolayer.py:327-330).  ``model``: dict of f32 tensors v_template [778,3], shapedirs [778,3,10],
    posedirs [778,3,135], J_regressor [16,778], weights [778,16], hands_mean [45], betas [10].
    pose [B,48] -> (verts [B,778,3], joints [B,21,3]) centred on joint ``center_idx``."""
    B = pose.shape[0]
    full = torch.cat([pose[:, :3], model["hands_mean"].unsqueeze(0) + pose[:, 3:48]], dim=1)
    rots = rodrigues(full.reshape(-1, 3)).view(B, 16, 3, 3)
    pose_map = (rots[:, 1:] - torch.eye(3)).reshape(B, 135)
    v_shaped = torch.matmul(model["shapedirs"], model["betas"]) + model["v_template"]          # [778,3]
    J = torch.matmul(model["J_regressor"], v_shaped)                                            # [16,3]
    v_posed = v_shaped.unsqueeze(0) + torch.matmul(model["posedirs"], pose_map.t()).permute(2, 0, 1)
    # kinematic chain (manolayer.py:264-303): G_root = [R_0 | J_0], G_j = G_parent * [R_j | J_j - J_parent]
    G = [None] * 16
    for j in range(16):
        par = MANO_PARENTS[j]
        t = J[j] if par < 0 else J[j] - J[par]
        local = torch.zeros(B, 4, 4, dtype=pose.dtype)                 # (float64 poses give a float64 reference for gradient checks)
        local[:, :3, :3] = rots[:, j]
        local[:, :3, 3] = t
        local[:, 3, 3] = 1.0
        G[j] = local if par < 0 else torch.matmul(G[par], local)
    G = torch.stack(G, dim=1)                                                                   # [B,16,4,4]
    # remove the rest pose: A_j = G_j with translation G_j[:3,3] - G_j[:3,:3] J_j  (manolayer.py:305-307)
    A = G.clone()
    A[:, :, :3, 3] = G[:, :, :3, 3] - torch.matmul(G[:, :, :3, :3], J.view(1, 16, 3, 1)).squeeze(-1)
    T = torch.einsum("vj,bjrc->bvrc", model["weights"], A)                                      # [B,778,4,4]
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=pose.dtype)], dim=2)
    verts = torch.einsum("bvrc,bvc->bvr", T, vh)[:, :, :3]
    jtr = torch.cat([G[:, :, :3, 3], verts[:, MANO_TIPS_RIGHT if side == "right" else MANO_TIPS_LEFT]], dim=1)[:, MANO_JOINT_ORDER]
    centre = jtr[:, center_idx:center_idx + 1] if center_idx is not None else torch.zeros(B, 1, 3, dtype=pose.dtype)
    return verts - centre, jtr - centre


def hand_encoder_forward(sd, model, p, reso, padding=0.1, out_dim=51, center_idx=9):
    """The hand encoder end to end (pointnet.py:135-206, out_mano=True): plane features -> mano_param;
    for out_dim > 30 the MANO layer runs on [0,0,0 | mano_param[6:]] (wrist position zeroed, :195-197)."""
    fea = plane_pointnet_forward(sd, p, reso, padding)
    param = mano_param_head(sd, fea)
    out = {"mano_param": param}
    if out_dim > 30:
        full = torch.cat([torch.zeros(param.shape[0], 3), param[:, 6:]], dim=1)
        out["mano_verts"], out["mano_joints"] = mano_forward(model, full, center_idx)
    return out


def rot_from_pyr(angles):
    """``R_from_PYR`` (src/common.py:591-604): R_pitch(angles[1]) @ R_yaw(angles[2]) @ R_roll(angles[0]) with the
    reference's sign conventions (roll about z; pitch about x and yaw about y, both transposed)."""
    import numpy as np
    roll, pitch, yaw = angles
    r_roll = np.array([[np.cos(roll), -np.sin(roll), 0], [np.sin(roll), np.cos(roll), 0], [0, 0, 1]])
    r_pitch = np.array([[1, 0, 0], [0, np.cos(pitch), np.sin(pitch)], [0, -np.sin(pitch), np.cos(pitch)]])
    r_yaw = np.array([[np.cos(yaw), 0, -np.sin(yaw)], [0, 1, 0], [np.sin(yaw), 0, np.cos(yaw)]])
    return r_pitch @ r_yaw @ r_roll


def hand_mesh_vertices(verts, mano_param, pc_ply):
    """Vertex post-processing of ``Generator3D.generate_hand_mesh`` (generation.py:88-110): remove the MANO
    frame offset and the fixed frame rotation, undo the predicted wrist rotation (rotation vector -> 'XYZ'
    Euler angles -> R_from_PYR), add the wrist position, normalise like the object cloud (norm_pc_1,
    common.py:606-612).  numpy in, float64 [V,3] out."""
    import numpy as np
    from scipy.spatial.transform import Rotation
    wrist_pos, wrist_rotvec = mano_param[:3], mano_param[3:6]
    euler = Rotation.from_rotvec(wrist_rotvec).as_euler("XYZ", degrees=False)
    v = verts.astype(np.float32) - np.array([0.11, 0.005, 0], dtype=np.float32)
    v = np.linalg.inv(rot_from_pyr(np.array([-np.pi / 2, np.pi / 2, 0]))) @ v.T
    v = np.linalg.inv(rot_from_pyr(np.array(euler))) @ v
    v = v.T + wrist_pos
    centroid = np.mean(pc_ply, axis=0)
    m = np.max(np.sqrt(np.sum((pc_ply - centroid) ** 2, axis=1)))
    return (v - centroid) / (2 * m)


# --------------------------------------------------------------------------
# tactile training-sample assembly of the VTacOH trainer (A13, training.py:502-626)
# --------------------------------------------------------------------------

MANO_TIP_JOINTS = [4, 8, 12, 16, 20]


def hand_tips_world(joints, wrist_pos, wrist_euler, pc_ply):
    """Fingertip joints in the object's normalised frame (training.py:547-556): the same frame change as
    ``hand_mesh_vertices`` but with the GROUND-TRUTH wrist position and the dataset's wrist Euler angles.
    numpy: joints [21,3] f32, wrist_pos [3], wrist_euler [3], pc_ply [M,3] -> float32 [5,3] (the reference stores the
    float64 result back into a float32 array)."""
    import numpy as np
    t = joints[MANO_TIP_JOINTS].astype(np.float32) - np.array([0.11, 0.005, 0], dtype=np.float32)
    t = np.linalg.inv(rot_from_pyr(np.array([-np.pi / 2, np.pi / 2, 0]))) @ t.T
    t = np.linalg.inv(rot_from_pyr(np.array(wrist_euler))) @ t
    t = t.T + wrist_pos
    centroid = np.mean(pc_ply, axis=0)
    m = np.max(np.sqrt(np.sum((pc_ply - centroid) ** 2, axis=1)))
    return ((t - centroid) / (2 * m)).astype(np.float32)


def trainer_img_assembly(p, occ, tips, touch_success, num_sample, rng=None):
    """Which query points a VTacOH training step decodes and which finger's tactile feature each carries
    (training.py:559-611).  Per scene: points within 0.05 of their nearest fingertip take that finger (if its touch
    succeeded; at most 512 per finger, drawn with ``choice`` -- with replacement -- beyond that), listed finger by finger;
    the remaining rows are drawn with ``randint(len(rest))`` and -- as the reference does -- used as indices into ALL
    points, not into the rest.  ``rng``: numpy's global generator by default (the reference's), consumed in the same
    order: every scene's ``choice`` calls first, then one ``randint`` per scene.
    numpy in; returns (rows [B,S] int64 point index of every sample row, finger [B,S] int64, -1 = no tactile feature)."""
    import numpy as np
    rng = np.random if rng is None else rng
    B, N = p.shape[:2]
    picked = []
    for b in range(B):
        d = np.sqrt(((p[b].astype(np.float64)[:, None, :] - tips[b].astype(np.float64)[None, :, :]) ** 2).sum(-1))
        idx_b, fin_b = [], []
        for f in range(5):
            if touch_success[b, f]:
                idx = np.where((d.min(1) < 0.05) & (d.argmin(1) == f))[0]
                if idx.shape[0] > 512:
                    idx = idx[rng.choice(idx.shape[0], 512)]
                idx_b += list(idx)
                fin_b += [f] * len(idx)
        picked.append((idx_b, fin_b))
    rows = np.zeros((B, num_sample), dtype=np.int64)
    finger = np.full((B, num_sample), -1, dtype=np.int64)
    everything = np.arange(N)
    for b in range(B):
        idx_b, fin_b = picked[b]
        k = len(idx_b)
        rows[b, :k] = idx_b
        finger[b, :k] = fin_b
        rest = everything[~np.isin(everything, idx_b)]
        rows[b, k:] = rng.randint(len(rest), size=num_sample - k)
    return rows, finger


# --------------------------------------------------------------------------
# VTacO (t2d) contact clouds from the tactile depth images (generation.py:202-257, training.py:817-853)
# --------------------------------------------------------------------------

T2D_W, T2D_H, T2D_NEAR, T2D_FAR, T2D_FOV = 240, 320, 0.019, 0.022, 60        # generation.py:18-19, 148-151


def depth_to_camera_cloud(depth):
    """``RFUniverseCamera.depth_2_camera_pointcloud`` (src/common.py:553-588), the unfiltered cloud: pinhole unprojection with
    f = H / (2 tan(fov/2)) and the principal point at (W/2, H/2), axes reordered to (z, -x, -y).  depth [H,W] -> [H*W,3] f64."""
    import numpy as np
    f = T2D_H / (2 * math.tan(math.radians(T2D_FOV / 2)))
    xmap, ymap = np.meshgrid(np.arange(T2D_W), np.arange(T2D_H))
    z = depth
    x = (xmap - T2D_W / 2) * z / f
    y = (ymap - T2D_H / 2) * z / f
    return np.stack([z, -x, -y], axis=-1).reshape(-1, 3)


def cam_to_world(pc, rot, trans):
    """``pc_cam_to_world`` (src/common.py:614-640): the three factor matrices as the reference writes them (its ``rot_z`` is not
    a rotation matrix), composed rot_z @ rot_x @ rot_y, the 3x3 block of the INVERSE of [R | t; 0 1] applied, then + t."""
    import numpy as np
    dx, dy, dz = rot
    rx = np.array([[np.cos(dx), 0, np.sin(dx)], [0, 1, 0], [-np.sin(dx), 0, np.cos(dx)]])
    ry = np.array([[np.cos(dy), -np.sin(dy), 0], [np.sin(dy), np.cos(dy), 0], [0, 0, 1]])
    rz = np.array([[0, 0, 1], [np.cos(dz), np.sin(dz), 0], [-np.sin(dz), np.cos(dz), 0]])
    ext = np.zeros((4, 4))
    ext[:3, :3] = rz @ rx @ ry
    ext[:3, 3] = trans
    ext[3, 3] = 1
    r_inv = np.linalg.inv(ext)[:3, :3]
    return (r_inv @ pc.T).T + np.asarray(trans)


def t2d_contact_clouds(depths, depth_origin, cam_pos, cam_rot, pc_ply, touch_success, rng=None):
    """Per finger of ONE scene: the contact point cloud in the object's normalised frame (generation.py:224-244).  Pixels whose
    depth differs from the flat reading ``depth_origin`` by more than 1e-4 are unprojected, at most 128 kept (``randint`` draws,
    with replacement, from numpy's global generator unless ``rng`` is given), moved to the world with the sample's camera pose
    (rotation + [-pi/2, 0, pi/2]) and normalised like the object cloud.  depths [5,H*W] f32 -> list of 5 arrays [k,3] f64 (k = 0 for
    failed touches)."""
    import numpy as np
    rng = np.random if rng is None else rng
    centroid = np.mean(pc_ply, axis=0)
    m = np.max(np.sqrt(np.sum((pc_ply - centroid) ** 2, axis=1)))
    clouds = []
    for t in range(5):
        if not touch_success[t]:
            clouds.append(np.zeros((0, 3)))
            continue
        depth = depths[t].reshape(T2D_H, T2D_W)
        idx = np.where(np.abs(depth.reshape(-1) - depth_origin) > 0.0001)[0]
        pc = depth_to_camera_cloud(depth)[idx]
        if pc.shape[0] > 128:
            pc = pc[rng.randint(pc.shape[0], size=128)]
        world = cam_to_world(pc, np.asarray(cam_rot[t]) + np.array([-np.pi / 2, 0, np.pi / 2]), cam_pos[t])
        clouds.append((world - centroid) / (2 * m))
    return clouds


# --------------------------------------------------------------------------
# VTacO (t2d) training step (A13, training.py:757-894): occupancy labels and sample assembly
# --------------------------------------------------------------------------


def winding_number(verts, faces, pts, chunk=2048):
    """Generalized winding number w(q) = 1/(4 pi) sum_f Omega_f(q) (Van Oosterom-Strackee solid angles, float64).
    The reference calls ``igl.fast_winding_number_for_meshes`` (training.py:723, 862), libigl's hierarchical approximation of
    this sum; libigl is un-vendored and absent here -> this restates the quantity it approximates (**parity unpinned** against
    igl itself; pinned by known answers: 1 inside / 0 outside closed meshes, 0.5 on a face of a cube from inside the face plane
    limit, additivity over a split mesh).  numpy: verts [V,3], faces [F,3] int, pts [N,3] -> float64 [N]."""
    import numpy as np
    v = np.asarray(verts, dtype=np.float64)
    tri = v[np.asarray(faces, dtype=np.int64)]                                    # [F,3,3]
    q = np.asarray(pts, dtype=np.float64)
    out = np.zeros(len(q))
    for lo in range(0, len(q), chunk):
        d = tri[None, :, :, :] - q[lo:lo + chunk, None, None, :]                     # [n,F,3,3]
        a, b, c = d[:, :, 0], d[:, :, 1], d[:, :, 2]
        la, lb, lc = (np.linalg.norm(x, axis=-1) for x in (a, b, c))
        num = np.einsum("nfi,nfi->nf", a, np.cross(b, c))
        den = la * lb * lc + (a * b).sum(-1) * lc + (b * c).sum(-1) * la + (c * a).sum(-1) * lb
        out[lo:lo + chunk] = (2.0 * np.arctan2(num, den)).sum(-1) / (4.0 * np.pi)
    return out


def trainer_t2d_assembly(p, depths, depth_origin, cam_pos, cam_rot, pc_ply, touch_success, c_img, meshes, num_sample, rng=None):
    """Sample assembly of the VTacO (t2d) training step (training.py:809-866).  Per scene: the contact clouds of the fingers whose
    touch succeeded (``t2d_contact_clouds``; their ``randint`` draws come first), stored as float32, become the first query
    points and carry the finger's tactile feature; the remaining ``num_sample - k`` points are ``randint(N)`` draws from the
    scene's points (with replacement) and carry ONES (the reference fills c_img_all with ones, not zeros); the occupancy target of
    every row is the winding number of the scene's mesh.  ``meshes``: list of (verts, faces) per scene.
    numpy; returns (p_sample [B,S,3] f32, c_img_all [B,S,C] f32, occ_new [B,S] f64)."""
    import numpy as np
    rng = np.random if rng is None else rng
    B, N = p.shape[:2]
    C = c_img.shape[2]
    p_sample = np.zeros((B, num_sample, 3), dtype=np.float32)
    feats = np.ones((B, num_sample, C), dtype=np.float32)
    occ_new = np.zeros((B, num_sample))
    for b in range(B):
        clouds = t2d_contact_clouds(depths[b], depth_origin, cam_pos[b], cam_rot[b], pc_ply[b], touch_success[b], rng)
        k = 0
        for t in range(5):
            n = len(clouds[t])
            if touch_success[b][t]:
                feats[b, k:k + n] = c_img[b, t]
                p_sample[b, k:k + n] = clouds[t].astype(np.float32)
                k += n
        p_sample[b, k:] = p[b][rng.randint(N, size=num_sample - k)]
        occ_new[b] = winding_number(meshes[b][0], meshes[b][1], p_sample[b])
    return p_sample, feats, occ_new
