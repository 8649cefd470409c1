"""GPU parity of the HIP UNet3D forward (channels-last implicit-GEMM conv3d + fused GroupNorm /
upsample / concat) against the oracle (torch-CPU restatement of reference unet3d.py:449-474)
and against the host PyTorch-ROCm path of the same module."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _unet(f_maps, levels, seed):
    from vtaco_amd.encoder.unet3d import UNet3D
    torch.manual_seed(seed)
    net = UNet3D(in_channels=32, out_channels=32, f_maps=f_maps, num_levels=levels)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "groupnorm" in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.2)
    return net


@pytest.mark.parametrize("R,levels,B", [(16, 3, 2), (32, 4, 1)])
def test_hip_unet3d_vs_oracle_and_torch_path(R, levels, B):
    from oracle import vtaco_oracle as orc
    net = _unet(32, levels, R)
    assert net.hip_supported()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 32, R, R, R, generator=g) * (torch.rand(B, 1, R, R, R, generator=g) < 0.05)   # sparse, like the scattered grid
    ref = orc.unet3d_forward({k: v.detach() for k, v in net.state_dict().items()}, x)
    net = net.to(DEV)
    with torch.no_grad():
        x_cl = x.to(DEV).permute(0, 2, 3, 4, 1).contiguous()
        got = net.forward_channels_last(x_cl).permute(0, 4, 1, 2, 3)          # one C-ABI call
        layered = net.forward_channels_last_layers(x_cl).permute(0, 4, 1, 2, 3)  # same kernels, per layer
        assert torch.equal(got, layered)
        host = net(x.to(DEV))
    scale = float(ref.abs().max())
    assert float((got.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, scale)
    assert float((host.cpu() - ref).abs().max()) <= 2e-4 * max(1.0, scale)


def test_encoder_inference_uses_hip_unet_and_matches_training_path():
    from vtaco_amd import ops
    from vtaco_amd.encoder import encoder_dict
    torch.manual_seed(0)
    enc = encoder_dict['pointnet_local_pool'](c_dim=32, dim=3, hidden_dim=32, unet3d=True, grid_resolution=32, plane_type='grid',
                                              unet3d_kwargs=dict(num_levels=3, f_maps=32, in_channels=32, out_channels=32)).to(DEV)
    g = torch.Generator().manual_seed(1)
    d = torch.randn(2, 3000, 3, generator=g)
    p = (0.3 * d / d.norm(dim=-1, keepdim=True) + 0.005 * torch.randn(2, 3000, 3, generator=g)).to(DEV)
    with torch.no_grad():
        fast = enc(p)['grid']
    assert ops.is_channels_last_grid(fast)
    slow = enc(p)['grid']                      # grad enabled -> host PyTorch-ROCm UNet3D
    assert float((fast - slow).abs().max()) <= 2e-4 * max(1.0, float(slow.abs().max()))


def test_split_bf16_conv_layers_match_f32_kernel():
    """vt_conv3d_gcr_bf16x3 against vt_conv3d_gcr on the shapes the 64^3 level uses: plain 32->32, the
    virtual concat [skip | upsample(low)] 96->32, and a 64-wide output (two cout blocks per workgroup)."""
    from vtaco_amd import ops
    g = torch.Generator().manual_seed(11)
    R = 64
    for C1, C2, Cout in ((32, 0, 32), (32, 64, 32), (32, 0, 64)):
        x = (torch.randn(1, R, R, R, C1, generator=g) * (torch.rand(1, R, R, R, 1, generator=g) < 0.3)).to(DEV)
        low = torch.randn(1, R // 2, R // 2, R // 2, C2, generator=g).to(DEV) if C2 else None
        w = (torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) * 0.05).to(DEV)
        gamma = (1 + 0.2 * torch.randn(C1 + C2, generator=g)).to(DEV)
        beta = (0.2 * torch.randn(C1 + C2, generator=g)).to(DEV)
        xs = ops.channel_stats(x)
        ls = ops.channel_stats(low) if C2 else None
        pf, ps = ops.conv3d_pack(w), ops.conv3d_pack(w, precision="bf16x3")
        ref, (rp, rn) = ops.gn_conv3d_relu(x, xs, low, ls, gamma, beta, 8, pf, Cout)
        got, (gp, gn) = ops.gn_conv3d_relu(x, xs, low, ls, gamma, beta, 8, pf, Cout, packed_w_bf16x3=ps)
        assert gn == (R // 8) ** 3 and rn != gn                     # it really took the split kernel
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert 0.0 < err <= 3e-5 * max(1.0, scale), (C1, C2, Cout, err, scale)
        # the epilogue's GroupNorm partial sums describe the same tensor
        assert float((gp.sum(1) - rp.sum(1)).abs().max()) <= 1e-3 * float(rp.sum(1).abs().max())


def test_hip_unet3d_at_64_split_bf16_vs_f32_and_oracle():
    from oracle import vtaco_oracle as orc
    net = _unet(32, 3, 64)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 32, 64, 64, 64, generator=g) * (torch.rand(1, 1, 64, 64, 64, generator=g) < 0.02)
    ref = orc.unet3d_forward({k: v.detach() for k, v in net.state_dict().items()}, x)
    net = net.to(DEV)
    x_cl = x.to(DEV).permute(0, 2, 3, 4, 1).contiguous()
    outs = {}
    with torch.no_grad():
        for prec in ("f32", "bf16x3"):
            net.precision = prec
            outs[prec] = net.forward_channels_last(x_cl)
            assert torch.equal(outs[prec], net.forward_channels_last_layers(x_cl))
    scale = max(1.0, float(ref.abs().max()))
    e32 = float((outs["f32"].permute(0, 4, 1, 2, 3).cpu() - ref).abs().max())
    es = float((outs["bf16x3"].permute(0, 4, 1, 2, 3).cpu() - ref).abs().max())
    assert e32 <= 1e-4 * scale and es <= 1e-4 * scale, (e32, es, scale)
    assert not torch.equal(outs["f32"], outs["bf16x3"])
