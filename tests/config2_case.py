"""BASELINE config 2 at the SHIPPED shape (R = 64, UNet3D num_levels 4 / f_maps 32, 128^3 lattice): builders shared by the CPU
test that pins the oracle against the reference-made fixture g15_config2.npz and the GPU test that runs the HIP path on it.
Parameters come from tests/seeded_fill.py (the fixture stores none)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from seeded_fill import keys_of, seeded_fill

UKW = dict(num_levels=4, f_maps=32, in_channels=32, out_channels=32)


def fixture():
    return np.load(os.path.join(GOLDEN, "g15_config2.npz"))


def sparse_volume(seed, C, R, density=0.02):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, C, R, R, R, generator=g) * (torch.rand(1, 1, R, R, R, generator=g) < density)


def unet_case(z, tag):
    """(mirror UNet3D module filled like the reference's, input volume) of the stand-alone cases 'u16' / 'u32'."""
    from vtaco_amd.encoder.unet3d import UNet3D
    levels, R, seed = {"u16": (3, 16, 150), "u32": (4, 32, 151)}[tag]
    net = seeded_fill(UNet3D(num_levels=levels, f_maps=32, in_channels=32, out_channels=32), seed).eval()
    assert keys_of(net) == list(z[f"{tag}_keys"])                  # the reference's checkpoint names, shapes and order
    x = sparse_volume(seed + 10, 32, R)
    assert abs(float(x.double().sum()) - z[f"{tag}_xsum"][0]) < 1e-9 and abs(float(x.double().abs().sum()) - z[f"{tag}_xsum"][1]) < 1e-9
    return net, x


def models(z):
    """Mirror encoder + decoder with the reference's seeded parameters (state_dict keys checked against the fixture)."""
    from vtaco_amd.conv_onet.models import decoder_dict
    from vtaco_amd.encoder import encoder_dict
    enc = encoder_dict['pointnet_local_pool'](c_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet3d=True, unet3d_kwargs=UKW,
                                              grid_resolution=64, plane_type="grid", padding=0.1, n_blocks=5)
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1, sample_mode="bilinear")
    seeded_fill(enc, 160)
    seeded_fill(dec, 161)
    assert keys_of(enc) == list(z["enc_keys"]) and keys_of(dec) == list(z["dec_keys"])
    return enc.eval(), dec.eval()


def cpu_sd(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def lattice_points(idx, nx=128, box=1.1):
    """Coordinates of lattice points ``idx`` (x-major order of make_3d_grid, common.py:178-197) without the 2 M-row table."""
    lin = box * torch.linspace(-0.5, 0.5, nx)
    idx = torch.as_tensor(idx)
    return torch.stack([lin[idx // (nx * nx)], lin[(idx // nx) % nx], lin[idx % nx]], dim=1)
