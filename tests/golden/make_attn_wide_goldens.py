#!/usr/bin/env python3
"""Golden vectors for AttentionDecoder beyond the shipped 32 / 32 (tests/golden/g17_attention_wide.npz), from the REAL reference:
AttentionDecoder.forward_img (src/conv_onet/models/decoder.py:176-271: grid sample -> TransformerFusion(d_model = c_dim,
key_feature_dim = 64) -> fc_p + the conditioned ResnetBlockFC stack) and its fuser alone (src/TransformerFusion.py:311-333) at

    D: c_dim 128, hidden_size 256, n_blocks 5 (the reference's class defaults), one chunk of N = 512 points
    E: c_dim 64,  hidden_size 64,  n_blocks 2, two chunks of N = 300 points (a ragged chunk: not a multiple of 32)

Eval mode.  Runs only in the build container (/root/reference).  Weights, grids and c_img are rounded to f16-representable values so
that the fixture stores them in half the bytes without changing the arithmetic.

    python tests/golden/make_attn_wide_goldens.py
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
    decoder = importlib.import_module("src.conv_onet.models.decoder")
    torch.set_num_threads(8)
    out = {}
    for tag, c_dim, hidden, nb, B, N, seed in (("D", 128, 256, 5, 1, 512, 170), ("E", 64, 64, 2, 2, 300, 171)):
        torch.manual_seed(seed)
        dec = decoder.AttentionDecoder(dim=3, c_dim=c_dim, hidden_size=hidden, n_blocks=nb, padding=0.1).eval()
        _randomise(dec, seed + 10)
        with torch.no_grad():
            for prm in dec.parameters():
                prm.copy_(prm.half().float())
        g = torch.Generator().manual_seed(seed + 20)
        R = 8
        grid = torch.randn(B, c_dim, R, R, R, generator=g).half().float()
        p = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.2
        # tactile features as the generator makes them: a finger's row on a fraction of the points, zeros elsewhere
        c_img = (torch.randn(B, N, c_dim, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.3)).half().float()
        with torch.no_grad():
            c = dec.sample_grid_feature(p, grid).transpose(1, 2)
            fused = dec.fuser(c_img, 1, c, 1)
            logits = dec.forward_img(p, {"grid": grid}, c_img)
        out.update({f"{tag}.grid": grid.numpy().astype(np.float16), f"{tag}.p": p.numpy(), f"{tag}.c_img": c_img.numpy().astype(np.float16),
                    f"{tag}.c": c.numpy(), f"{tag}.fused": fused.numpy(), f"{tag}.logits_img": logits.numpy(),
                    f"{tag}.shape": np.array([c_dim, hidden, nb, B, N], dtype=np.int64)})
        out.update({f"sd.{tag}.{k}": v.detach().numpy().astype(np.float16) for k, v in dec.state_dict().items()})
    _save("g17_attention_wide.npz", **out)


if __name__ == "__main__":
    main()
