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
