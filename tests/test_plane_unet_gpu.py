"""GPU: the hand encoder's 2-D U-Net as one persistent HIP launch (vt_plane_unet_fwd / vt_plane_unet_bwd, csrc/plane_unet.hip)
against the oracle's restatement of reference src/encoder/unet.py (oracle.unet2d_forward, pinned by g10 / g19)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _net(depth, in_ch, start, classes, seed):
    from vtaco_amd.encoder.unet import UNet
    torch.manual_seed(seed)
    net = UNet(classes, in_channels=in_ch, depth=depth, start_filts=start)
    with torch.no_grad():                                   # reset_params zeroes every bias: give them values
        for name, p in net.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape) * 0.05)
    return net


SHAPES = [  # depth, in_ch, start, classes, n_img, H, W
    (4, 32, 32, 32, 3, 32, 32),        # the shipped hand encoder: three planes of one scene
    (4, 32, 32, 32, 24, 32, 32),       # ... of a training batch of eight scenes
    (3, 64, 32, 96, 5, 16, 16),
    (2, 32, 32, 32, 9, 8, 8),          # bottom level 4 x 4: two images per pixel tile, a ragged last tile
    (4, 32, 32, 64, 2, 64, 64),        # 64 x 64 planes: two tiles per row
    (3, 32, 64, 32, 2, 64, 32),        # non-square, 64 filters
    (4, 512, 32, 512, 1, 64, 64),      # the t2d digit encoder's planes (c_dim 512, start_filts 32 through the config's typo)
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_plane_unet_forward_against_the_oracle(shape):
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    depth, in_ch, start, classes, n_img, H, W = shape
    net = _net(depth, in_ch, start, classes, seed=depth * 7 + n_img)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n_img, in_ch, H, W, generator=g)
    x[x.abs() < 0.3] = 0.0                                   # planes are mostly empty cells
    with torch.no_grad():
        ref = orc.unet2d_forward({k: v.detach() for k, v in net.state_dict().items()}, x)
    net = net.to(DEV)
    assert net.hip_supported(x.to(DEV))
    with torch.no_grad():
        got = net(x.to(DEV))
        again = net(x.to(DEV))
    scale = max(1.0, float(ref.abs().max()))
    err = float((got.cpu() - ref).abs().max())
    assert err <= 2e-5 * scale, (err, scale)
    assert torch.equal(got, again)                           # fixed summation order: bit-reproducible
    # the nn.Conv2d modules (MIOpen) agree as well, and a changed weight is picked up (the blob is repacked)
    with torch.no_grad():
        mods = net.forward_modules(x.to(DEV))
        assert float((mods - got).abs().max()) <= 1e-4 * scale
        net.conv_final.bias.add_(1.0)
        assert float((net(x.to(DEV)) - got - 1.0).abs().max()) <= 1e-5 * scale


def test_plane_unet_refused_shapes_keep_the_modules():
    from vtaco_amd.encoder.unet import UNet
    torch.manual_seed(0)
    x = torch.randn(1, 32, 32, 32, device=DEV)
    for kw in (dict(start_filts=16), dict(merge_mode="add"), dict(up_mode="upsample"), dict(depth=5, start_filts=32)):
        net = UNet(32, in_channels=32, **dict(dict(depth=3), **kw)).to(DEV)
        assert not net.hip_supported(x)
        with torch.no_grad():
            assert net(x).shape == (1, 32, 32, 32)


BWD_SHAPES = [(4, 32, 32, 32, 3, 32, 32), (4, 32, 32, 32, 24, 32, 32), (3, 64, 32, 96, 5, 16, 16), (2, 32, 32, 32, 9, 8, 8),
              (3, 32, 64, 32, 2, 64, 32), (4, 512, 32, 512, 1, 64, 64)]


@pytest.mark.parametrize("shape", BWD_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_plane_unet_backward_against_the_oracle_autograd(shape):
    """vt_plane_unet_bwd against torch-CPU autograd through the oracle's restatement: the input gradient and every parameter's
    gradient, with the framework's conv / pool operators made to raise on the device (nothing may fall back to MIOpen)."""
    from oracle import vtaco_oracle as orc
    depth, in_ch, start, classes, n_img, H, W = shape
    net = _net(depth, in_ch, start, classes, seed=depth * 5 + n_img)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n_img, in_ch, H, W, generator=g)
    x[x.abs() < 0.3] = 0.0
    wout = torch.randn(n_img, classes, H, W, generator=g)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    (orc.unet2d_forward(sd, xr) * wout).sum().backward()
    net = net.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    import torch.nn.functional as F
    saved = F.conv2d, F.conv_transpose2d, F.max_pool2d

    def boom(*a, **k):
        raise AssertionError("a framework convolution / pool ran under the HIP U-Net")
    F.conv2d = F.conv_transpose2d = F.max_pool2d = boom
    try:
        out = net(xd)
        (out * wout.to(DEV)).sum().backward()
    finally:
        F.conv2d, F.conv_transpose2d, F.max_pool2d = saved

    # The net is piecewise linear: where the CPU's forward and the kernel's differ in the last bit, a ReLU at ~0 or a max-pool window
    # with two near-equal entries can take the other branch, and the gradients then differ on that unit's receptive field (seen on
    # the two large cases: one window in 400 000).  Small cases must agree everywhere; the large ones everywhere but on < 10 % of dx (a bottom-level unit of a 64 x 64 plane is seen by a
    # quarter of the image), and
    # to 3 % in the norm of every gradient -- a parameter's sums the flipped unit's error over its pixels, so all its entries move a
    # little (a wrong term or sign is O(1) in both measures).
    strict = n_img * H * W * max(in_ch, start) <= 6144 * 64       # (~ the number of ReLU / pool units)

    def close(a, b, tag):
        scale = max(1e-6, float(b.abs().max()))
        err = (a.cpu() - b).abs()
        if strict:
            assert float(err.max()) <= 3e-5 * scale, (tag, float(err.max()), scale)
        else:
            if tag == "dx":                                  # the flipped unit's receptive field only
                frac = float((err > 3e-5 * scale).float().mean())
                assert frac < 0.10, (tag, frac, float(err.max()), scale)
            assert float((a.cpu() - b).norm()) <= 3e-2 * float(b.norm()), (tag, float((a.cpu() - b).norm()), float(b.norm()))
    close(xd.grad, xr.grad, "dx")
    for name, p in net.named_parameters():
        close(p.grad, sd[name].grad, name)
    # written, not accumulated into stale buffers; a second backward accumulates through autograd as usual
    g1 = {n: p.grad.clone() for n, p in net.named_parameters()}
    (net(xd) * wout.to(DEV)).sum().backward()
    for n, p in net.named_parameters():
        assert float((p.grad - 2 * g1[n]).abs().max()) <= 1e-6 * max(1.0, float(g1[n].abs().max())), n


def test_plane_unet_backward_routes_pool_ties_to_the_first_maximum():
    """Every 2x2 window an exact tie (conv weights zero, positive biases: constant activations): the max-pool's gradient must reach the
    window's FIRST entry, as torch's max_pool2d backward does -- compared with torch-CPU autograd exactly (the values tie on both sides)."""
    from oracle import vtaco_oracle as orc
    net = _net(3, 32, 32, 32, seed=2)
    with torch.no_grad():
        for l in (0, 1):
            for conv in (net.down_convs[l].conv1, net.down_convs[l].conv2):
                conv.weight.zero_()
                conv.bias.fill_(0.25 * (l + 1))
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 32, 16, 16, generator=g)
    wout = torch.randn(2, 32, 16, 16, generator=g)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    (orc.unet2d_forward(sd, x) * wout).sum().backward()
    net = net.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    (net(xd) * wout.to(DEV)).sum().backward()
    for name, p in net.named_parameters():
        ref = sd[name].grad
        assert float((p.grad.cpu() - ref).abs().max()) <= 3e-5 * max(1e-6, float(ref.abs().max())), name
    assert float(net.down_convs[0].conv2.bias.grad.abs().max()) > 0      # the tied windows did carry gradient


def _g19_net():
    """vtaco_amd's UNet with the parameters the reference module drew for g19 (same seeds; every tensor's sums re-checked)."""
    from conftest import GOLDEN
    from vtaco_amd.encoder.unet import UNet
    z = np.load(os.path.join(GOLDEN, "g19_plane_unet.npz"))
    seed, bias_seed = int(z["seeds"][0]), int(z["seeds"][1])
    torch.manual_seed(seed)
    net = UNet(32, in_channels=32, depth=4, start_filts=32, merge_mode="concat")
    g = torch.Generator().manual_seed(bias_seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    for name, p in net.named_parameters():
        ref = z[f"psum.{name}"]
        assert float(p.detach().double().sum()) == ref[0] and float(p.detach().double().abs().sum()) == ref[1], name
    return net, z


def test_plane_unet_against_the_reference_module_golden():
    """g19: the REAL reference src/encoder/unet.py at the shipped shape (depth 4, 32 filters, three 32 x 32 planes), forward and
    backward: output, dL/dx, and every parameter gradient's 64 sampled entries + sum against vt_plane_unet_fwd / _bwd."""
    net, z = _g19_net()
    net = net.to(DEV)
    x = torch.from_numpy(z["x"]).to(DEV).requires_grad_(True)
    out = net(x)
    (out * torch.from_numpy(z["w"]).to(DEV)).sum().backward()
    scale = float(np.abs(z["out"]).max())
    assert float((out.detach().cpu() - torch.from_numpy(z["out"])).abs().max()) <= 2e-5 * scale
    # (gradients: the flip-aware criterion of the oracle test above -- the reference ran on the CPU, a ReLU at ~0 or a near-tied pool
    # window may take the other branch here; a flipped unit moves dx on its receptive field and every parameter's sum a little)
    dscale = float(np.abs(z["dx"]).max())
    derr = (x.grad.cpu() - torch.from_numpy(z["dx"])).abs()
    assert float((derr > 3e-5 * dscale).float().mean()) < 0.10 and float(derr.norm()) <= 3e-2 * float(np.linalg.norm(z["dx"]))
    sample_seed = int(z["seeds"][3])
    for name, p in net.named_parameters():
        gr = p.grad.double().reshape(-1).cpu()
        idx = torch.randint(0, gr.numel(), (64,), generator=torch.Generator().manual_seed(sample_seed + sum(map(ord, name))))
        ref = torch.from_numpy(z[f"gsample.{name}"]).double()
        gs = z[f"gsum.{name}"]
        typical = max(1e-9, float(gs[1]) / gr.numel())                 # the gradient's mean |entry|
        assert float((gr[idx] - ref).abs().max()) <= 3e-2 * max(typical * 8, float(ref.abs().max())), name
        assert abs(float(gr.sum()) - gs[0]) <= 3e-2 * gs[1] + 1e-9, name
