import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden, sub_sd
from oracle import vtaco_oracle as orc
from vtaco_amd.conv_onet.models import decoder_dict
DEV="cuda:0"
_, sd = load_golden("g5_fusion.npz")
dec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32)
dec.load_state_dict({k: v for k, v in sd.items()}, strict=False)
dec = dec.to(DEV).eval()
sd64 = {k: v.double() for k, v in sub_sd(sd, "fuser.").items()}
for (B, N) in ((1, 33), (3, 100), (2, 1000), (1, 64), (1, 32)):
    g = torch.Generator().manual_seed(N)
    ci = torch.randn(B, N, 32, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.3)
    cc = torch.randn(B, N, 32, generator=g)
    ref = orc.transformer_fusion(sub_sd(sd, "fuser."), ci, cc)
    ref64 = orc.transformer_fusion(sd64, ci.double(), cc.double())
    with torch.no_grad():
        out = dec.fuser(ci.to(DEV), 1, cc.to(DEV), 1).cpu()
    print(os.environ.get("VTACO_FUSION_SCORE_TERMS", "f8"), B, N, "vs f32 oracle %.2e" % float((out - ref).abs().max()), "vs f64 %.2e" % float((out.double() - ref64).abs().max()), "f32 oracle vs f64 %.2e" % float((ref.double()-ref64).abs().max()))
