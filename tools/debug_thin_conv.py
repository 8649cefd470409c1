"""Layer-level check of the split-bf16 conv on thin (8x8x2) tiles against torch conv3d and the f32 kernel, with stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, R, Cin, Cout, C2) in ((2, 16, 64, 64, 0), (1, 16, 128, 128, 256), (2, 8, 128, 256, 0), (1, 16, 64, 128, 0), (2, 16, 32, 32, 0), (3, 24, 32, 64, 0)):
    C1 = Cin
    x = torch.randn(B, R, R, R, C1, generator=g)
    low = torch.randn(B, R // 2, R // 2, R // 2, C2, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) * (1.0 / (27 * (C1 + C2)) ** 0.5)
    xin = x if low is None else torch.cat([x, low.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)], dim=-1)
    ref = torch.relu(torch.nn.functional.conv3d(xin.permute(0, 4, 1, 2, 3).double(), w.double(), padding=1)).permute(0, 2, 3, 4, 1).float()
    pw, ps = ops.conv3d_pack(w.to(dev), "f32"), ops.conv3d_pack(w.to(dev), "bf16x3")
    xd, ld = x.to(dev), (low.to(dev) if low is not None else None)
    o32, st32 = ops.conv3d_gcr(xd, ld, None, pw, Cout)
    o16, st16 = ops.conv3d_gcr(xd, ld, None, pw, Cout, packed_w_bf16x3=ps)
    s32 = st32[0].sum(1).cpu(); s16 = st16[0].sum(1).cpu()
    refsum = torch.stack([ref.sum((1, 2, 3)), (ref * ref).sum((1, 2, 3))], dim=-1)
    print(f"B{B} R{R} {C1}+{C2}->{Cout}: nblk f32 {st32[1]} split {st16[1]} | f32 err {float((o32.cpu()-ref).abs().max()):.2e} split err {float((o16.cpu()-ref).abs().max()):.2e}"
          f" (|ref| {float(ref.abs().max()):.2f}) | stats err f32 {float((s32-refsum).abs().max()/refsum.abs().max()):.1e} split {float((s16-refsum).abs().max()/refsum.abs().max()):.1e}")
