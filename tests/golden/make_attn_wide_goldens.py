#!/usr/bin/env python3
"""Golden vectors for AttentionDecoder beyond the shipped 32 / 32 (tests/golden/g17_attention_wide.npz), from the REAL reference:
AttentionDecoder.forward_img (src/conv_onet/models/decoder.py:176-271: grid sample -> TransformerFusion(d_model = c_dim,
key_feature_dim = 64) -> fc_p + the conditioned ResnetBlockFC stack) and its fuser alone (src/TransformerFusion.py:311-333) at

    D: c_dim 128, hidden_size 256, n_blocks 5 (the reference's class defaults), one chunk of N = 512 points
    E: c_dim 64,  hidden_size 64,  n_blocks 2, two chunks of N = 300 points (a ragged chunk: not a multiple of 32)

and (tests/golden/g20_attention_wide_grads.npz) the gradients of L = sum(logits * w) through the same forward_img under the reference's
own autograd: d grid, d c_img, and per parameter the gradient's sum, abs-sum and 64 sampled entries.

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


SAMPLE_SEED = 2000


def sample_index(name, numel):
    g = torch.Generator().manual_seed(SAMPLE_SEED + sum(map(ord, name)))
    return torch.randint(0, numel, (64,), generator=g)


def main():
    _install_stubs()
    decoder = importlib.import_module("src.conv_onet.models.decoder")
    torch.set_num_threads(8)
    out, grads = {}, {}
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
        # the same call under autograd (g20): the inputs and parameters are those of g17
        w = torch.randn(B, N, generator=torch.Generator().manual_seed(seed + 30))
        grid.requires_grad_(True)
        c_img.requires_grad_(True)
        (dec.forward_img(p, {"grid": grid}, c_img) * w).sum().backward()
        grads.update({f"{tag}.w": w.numpy(), f"{tag}.d_grid": grid.grad.numpy(), f"{tag}.d_c_img": c_img.grad.numpy()})
        for name, prm in dec.named_parameters():
            if prm.grad is None:                                    # fc_p_img / after_norm / fc_out_contact: not on this path
                continue
            gr = prm.grad.double().reshape(-1)
            grads[f"{tag}.gsum.{name}"] = np.array([float(gr.sum()), float(gr.abs().sum())])
            grads[f"{tag}.gsample.{name}"] = gr[sample_index(name, gr.numel())].numpy().astype(np.float32)
    _save("g17_attention_wide.npz", **out)
    _save("g20_attention_wide_grads.npz", **grads)


if __name__ == "__main__":
    main()
