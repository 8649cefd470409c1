"""Diagnostic: vt_decode_fwd_wide against the oracle with parts of the network zeroed, to localise an error."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vtaco_oracle as orc
from vtaco_amd.conv_onet.models.decoder import LocalDecoder
DEV = "cuda:0"
for hidden, c_dim, nb, leaky in ((32, 32, 1, True), (64, 32, 1, False), (64, 32, 2, False), (64, 64, 1, False)):
    for zero in ("none", "fc_c", "blocks", "fc_c+blocks", "fc1"):
        torch.manual_seed(0)
        dec = LocalDecoder(c_dim=c_dim, hidden_size=hidden, n_blocks=nb, leaky=leaky, with_contact=True)
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for n, p in dec.named_parameters():
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
                if ("fc_c" in zero and n.startswith("fc_c")) or ("blocks" in zero and n.startswith("blocks")) or (zero == "fc1" and "fc_1" in n):
                    p.zero_()
        sd = {k: v.detach().clone() for k, v in dec.state_dict().items()}
        dec = dec.to(DEV)
        grid = torch.randn(1, c_dim, 8, 8, 8, generator=g)
        p = (torch.rand(1, 64, 3, generator=g) - 0.5)
        with torch.no_grad():
            got = dec(p.to(DEV), {"grid": grid.to(DEV)}).cpu()
        ref = orc.local_decoder_forward(sd, p, grid, leaky=leaky)
        print(hidden, c_dim, nb, leaky, zero, "err", float((got - ref).abs().max()), "scale", float(ref.abs().max()))
