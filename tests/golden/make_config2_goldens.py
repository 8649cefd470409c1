#!/usr/bin/env python3
"""Golden vectors at the SHIPPED shape of BASELINE config 2 (g15_config2.npz), from the REAL reference modules
(src/encoder/pointnet.py:135-200, src/encoder/unet3d.py:449-474, src/conv_onet/models/decoder.py:135-161).  Build container only.

The networks' parameters (PointNet + 4-level UNet3D with f_maps 32 at R = 64: ~4 M numbers; LocalDecoder) are not stored: both sides
fill them with ``tests/seeded_fill.py`` in state_dict order.  The fixture holds the input cloud, samples of the reference's feature
grid, the reference's logits on a seeded sample of the 128^3 lattice, and two stand-alone UNet3D cases (f_maps = 32, so the HIP
network -- not the host path -- is what a test of them exercises).

    python tests/golden/make_config2_goldens.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_goldens as mg          # noqa: E402
from seeded_fill import keys_of, seeded_fill   # noqa: E402


def sparse_volume(seed, C, R, density=0.02):
    """A seeded volume like the encoder's scatter-mean grid: ~2 % occupied voxels."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, C, R, R, R, generator=g) * (torch.rand(1, 1, R, R, R, generator=g) < density)


def sphere_cloud(seed, T=3000, r=0.3, sigma=0.005):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(1, T, 3, generator=g)
    return r * d / d.norm(dim=-1, keepdim=True) + sigma * torch.randn(1, T, 3, generator=g)


def main():
    mg._install_stubs()
    from src.common import make_3d_grid
    decoder = importlib.import_module("src.conv_onet.models.decoder")
    pointnet = importlib.import_module("src.encoder.pointnet")
    from src.encoder.unet3d import UNet3D
    torch.set_num_threads(8)
    out = {}

    # ---- stand-alone UNet3D, f_maps 32: 3 levels at 16^3 (stored whole), 4 levels at 32^3 (seeded sample of the output) ----
    for tag, levels, R, seed in (("u16", 3, 16, 150), ("u32", 4, 32, 151)):
        net = seeded_fill(UNet3D(num_levels=levels, f_maps=32, in_channels=32, out_channels=32), seed).eval()
        x = sparse_volume(seed + 10, 32, R)
        with torch.no_grad():
            y = net(x)
        out[f"{tag}_keys"] = np.array(keys_of(net))
        out[f"{tag}_xsum"] = np.array([float(x.double().sum()), float(x.double().abs().sum())])
        if R == 16:
            out[f"{tag}_y"] = y.numpy()
        else:
            g = torch.Generator().manual_seed(seed + 20)
            vox = torch.randperm(R ** 3, generator=g)[:4096]
            out[f"{tag}_vox"] = vox.numpy()
            out[f"{tag}_y_at"] = y[0].reshape(32, -1)[:, vox].numpy()
        out[f"{tag}_ystat"] = np.array([float(y.double().mean()), float(y.double().abs().mean()), float(y.abs().max())])

    # ---- config 2 at the shipped shape: cloud -> encoder (R=64, 4 levels, f_maps 32) -> grid; 128^3 lattice sample -> logits ----
    enc = pointnet.LocalPoolPointnet(c_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet3d=True,
                                     unet3d_kwargs=dict(num_levels=4, f_maps=32, in_channels=32, out_channels=32),
                                     grid_resolution=64, plane_type="grid", padding=0.1, n_blocks=5)
    dec = decoder.LocalDecoder(dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1, sample_mode="bilinear")
    seeded_fill(enc, 160)
    seeded_fill(dec, 161)
    enc.eval(), dec.eval()
    cloud = sphere_cloud(0)
    nx = 128
    g = torch.Generator().manual_seed(162)
    sample = torch.sort(torch.randperm(nx ** 3, generator=g)[:65536]).values
    lattice = 1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
    # a second sample near the surface of the cloud (|p| ~ 0.3: where the features are not the empty-space constant)
    rad = lattice.norm(dim=1)
    near = torch.nonzero((rad - 0.3).abs() < 0.03).squeeze(1)
    near = near[torch.randperm(near.numel(), generator=g)[:32768]].sort().values
    with torch.no_grad():
        grid = enc(cloud)["grid"]
        logits = dec(lattice[sample].unsqueeze(0), {"grid": grid})[0]
        logits_near = dec(lattice[near].unsqueeze(0), {"grid": grid})[0]
    vox = torch.randperm(64 ** 3, generator=g)[:8192]
    occ = torch.nonzero(grid[0].abs().amax(0).reshape(-1) > 0.5 * float(grid.abs().mean())).squeeze(1)
    out.update(enc_keys=np.array(keys_of(enc)), dec_keys=np.array(keys_of(dec)), cloud=cloud.numpy(),
               grid_vox=vox.numpy(), grid_at=grid[0].reshape(32, -1)[:, vox].numpy(),
               grid_stat=np.array([float(grid.double().mean()), float(grid.double().abs().mean()), float(grid.abs().max())]),
               grid_chan_mean=grid[0].double().mean(dim=(1, 2, 3)).numpy(),
               sample=sample.numpy().astype(np.int64), logits=logits.numpy(),
               near=near.numpy().astype(np.int64), logits_near=logits_near.numpy())
    print("grid |mean| %.4f max %.3f; logits range [%.3f, %.3f], near range [%.3f, %.3f] std %.3f" % (
        out["grid_stat"][1], out["grid_stat"][2], float(logits.min()), float(logits.max()),
        float(logits_near.min()), float(logits_near.max()), float(logits_near.std())))
    mg._save("g15_config2.npz", **out)


if __name__ == "__main__":
    main()
