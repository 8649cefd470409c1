#!/usr/bin/env python3
"""Golden vectors for the decoder shapes beyond the shipped 32/32 (tests/golden/g16_decode_wide.npz), from the REAL reference:
LocalDecoder.forward / forward_img / forward_contact (src/conv_onet/models/decoder.py:71-161) at

    A: hidden_size 64,  c_dim 32,  n_blocks 5, leaky=True  (leaky_relu(0.2) in front of the heads), with_contact
    B: hidden_size 256, c_dim 128, n_blocks 3, leaky=False (the class defaults' widths)
    C: hidden_size 32,  c_dim 32,  n_blocks 2, sample_mode='nearest' (F.grid_sample's other mode for 5-D input)

on random points (both clamps hit) and on an 8^3 lattice.  Runs only in the build container (/root/reference).  Weights, grids and
c_img are rounded to f16-representable values so the fixture stores them in half the bytes without changing the arithmetic.

    python tests/golden/make_wide_goldens.py
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import _install_stubs, _randomise, _save  # noqa: E402


def main():
    _install_stubs()
    from src.common import make_3d_grid
    decoder = importlib.import_module("src.conv_onet.models.decoder")
    torch.set_num_threads(8)
    out = {}
    for tag, hidden, c_dim, nb, leaky, seed, mode in (("A", 64, 32, 5, True, 160, "bilinear"), ("B", 256, 128, 3, False, 161, "bilinear"),
                                                     ("C", 32, 32, 2, False, 162, "nearest")):
        torch.manual_seed(seed)
        dec = decoder.LocalDecoder(dim=3, c_dim=c_dim, hidden_size=hidden, n_blocks=nb, leaky=leaky, padding=0.1,
                                   sample_mode=mode, with_contact=True)
        _randomise(dec, seed + 10)
        with torch.no_grad():
            for prm in dec.parameters():
                prm.copy_(prm.half().float())
        g = torch.Generator().manual_seed(seed + 20)
        R, nx = 8, 8
        grid = torch.randn(2, c_dim, R, R, R, generator=g).half().float()
        prand = (torch.rand(2, 333, 3, generator=g) - 0.5) * 1.3
        c_img = (torch.randn(2, 333, c_dim, generator=g) * (torch.rand(2, 333, 1, generator=g) < 0.3)).half().float()
        pts = (1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).unsqueeze(0)
        with torch.no_grad():
            lo = dec(prand, {"grid": grid})
            lo_img = dec.forward_img(prand, {"grid": grid}, c_img)
            lo_c, lo_cc = dec.forward_contact(prand, {"grid": grid})
            lo_lat = dec(pts, {"grid": grid[:1]})
        out.update({f"{tag}.grid": grid.numpy().astype(np.float16), f"{tag}.prand": prand.numpy(),
                    f"{tag}.c_img": c_img.numpy().astype(np.float16), f"{tag}.logits": lo.numpy(), f"{tag}.logits_img": lo_img.numpy(),
                    f"{tag}.logits_contact": lo_c.numpy(), f"{tag}.logits_contact2": lo_cc.numpy(), f"{tag}.logits_lattice": lo_lat.numpy(),
                    f"{tag}.shape": np.array([hidden, c_dim, nb, int(leaky), nx, int(mode == "nearest")], dtype=np.int64)})
        out.update({f"sd.{tag}.{k}": v.detach().numpy().astype(np.float16) for k, v in dec.state_dict().items()})
    _save("g16_decode_wide.npz", **out)


if __name__ == "__main__":
    main()
