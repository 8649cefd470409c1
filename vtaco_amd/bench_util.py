"""Synthetic scenes for bench.py / smoke (SURVEY.md section 8d inputs)."""
from __future__ import annotations

import torch

from .conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict


def randomise_fc1(module, seed):
    """fc_1.weight ~ N(0, 0.1^2): the default zero init would make every ResNet
    block a no-op ('degenerate random init')."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in module.named_parameters():
            if name.endswith("fc_1.weight"):
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.1)


def sphere_cloud(seed, T=3000, r=0.3, sigma=0.005):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(1, T, 3, generator=g)
    return r * d / d.norm(dim=-1, keepdim=True) + sigma * torch.randn(1, T, 3, generator=g)


def build_scene(seed, device, R=64):
    torch.manual_seed(0)
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1)
    randomise_fc1(dec, 1)
    model = ConvolutionalOccupancyNetwork(dec, None, device=device)
    try:
        from .encoder import encoder_dict
    except ImportError:
        encoder_dict = None
    g = torch.Generator().manual_seed(1000 + seed)
    cloud = sphere_cloud(seed)
    if encoder_dict is not None and 'pointnet_local_pool' in encoder_dict:
        torch.manual_seed(0)
        enc = encoder_dict['pointnet_local_pool'](
            dim=3, c_dim=32, padding=0.1, hidden_dim=32, plane_type='grid', grid_resolution=R,
            unet3d=True, unet3d_kwargs=dict(num_levels=4, f_maps=32, in_channels=32, out_channels=32))
        randomise_fc1(enc, 2)
        model.encoder = enc.to(device)
        with torch.no_grad():
            grid = model.encode_inputs(cloud.to(device))['grid']
    else:
        grid = torch.randn(1, 32, R, R, R, generator=g).to(device)
    from . import ops
    grid = ops.grid_to_channels_last(grid)
    sd_cpu = {k: v.detach().cpu() for k, v in dec.state_dict().items()}

    def c_img(nx):
        gg = torch.Generator().manual_seed(2000 + seed)
        n = nx ** 3
        mask = (torch.rand(1, n, 1, generator=gg) < 0.02).float()
        return (torch.randn(1, 1, 32, generator=gg) * mask).to(device)

    return {"model": model, "grid": grid, "grid_cpu": grid.detach().cpu().contiguous(),
            "sd_decoder_cpu": sd_cpu, "c_img": c_img, "cloud": cloud}
