"""Error of the TransformerFusion kernels on the reference-made golden vectors (g5_fusion.npz) and on random inputs against the
float64 oracle: the numbers behind DESIGN.md's fusion.hip paragraph.  python tools/probe/fusion_err.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, sub_sd  # noqa: E402
from oracle import vtaco_oracle as orc  # noqa: E402
from vtaco_amd.conv_onet.models import decoder_dict  # noqa: E402

DEV = "cuda:0"
T = torch.from_numpy
a, sd = load_golden("g5_fusion.npz")
dec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32)
dec.load_state_dict({k: v for k, v in sd.items()}, strict=False)
dec = dec.to(DEV).eval()
with torch.no_grad():
    for n in (256, 2048):
        out = dec.fuser(T(a[f"c_img{n}"]).to(DEV), 1, T(a[f"c{n}"]).to(DEV), 1)
        print(f"golden fused{n}: max |err| {float((out.cpu() - T(a[f'fused{n}'])).abs().max()):.2e}")
    lo = dec.forward_img(T(a["p"]).to(DEV), {"grid": T(a["grid"]).to(DEV)}, T(a["c_img256"]).to(DEV))
    print(f"golden logits: max |err| {float((lo.cpu() - T(a['logits'])).abs().max()):.2e}")
    sd64 = {k: v.double() for k, v in sub_sd(sd, "fuser.").items()}
    for (B, N) in ((2, 1000), (1, 2048)):
        g = torch.Generator().manual_seed(N)
        ci = torch.randn(B, N, 32, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.3)
        cc = torch.randn(B, N, 32, generator=g)
        ref = orc.transformer_fusion(sd64, ci.double(), cc.double())
        out = dec.fuser(ci.to(DEV), 1, cc.to(DEV), 1)
        print(f"random B={B} N={N} vs float64 oracle: max |err| {float((out.cpu().double() - ref).abs().max()):.2e}")
