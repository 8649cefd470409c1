"""A/B of the split-f16 conv variants on the encoder's large layers (one process per VTACO_CONV_SPEC value: the switch is read once).
Prints, per layer shape, microseconds per launch and a checksum of the output bits and of the statistics.
Usage: VTACO_CONV_SPEC=1 python tools/probe/conv_ab.py ; VTACO_CONV_SPEC=2 python tools/probe/conv_ab.py"""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("AB_B", "1"))
shapes = [(64, 32, 0, 32), (64, 32, 64, 32), (32, 32, 0, 64), (32, 64, 0, 64), (32, 64, 128, 64)]
for R, C1, C2, Cout in shapes:
    g = torch.Generator().manual_seed(R + C1 + C2)
    x = torch.randn(B, R, R, R, C1, generator=g).to(dev)
    low = torch.randn(B, R // 2, R // 2, R // 2, C2, generator=g).to(dev) if C2 else None
    w = (torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) * 0.05).to(dev)
    gamma, beta = (torch.rand(C1 + C2, generator=g) + 0.5).to(dev), (torch.randn(C1 + C2, generator=g) * 0.1).to(dev)
    xs = ops.channel_stats(x)
    ls = ops.channel_stats(low) if C2 else None
    pf, ph = ops.conv3d_pack(w), ops.conv3d_pack(w, precision="f16x3")
    ss = ops.gn_scale_shift(xs, ls, C1, C2, B, R ** 3, gamma, beta, 8, 1e-5, dev)
    fn = lambda: ops.conv3d_gcr(x, low, ss, pf, Cout, True, None, packed_w_f16x3=ph)
    out, st = fn()
    torch.cuda.synchronize()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record(); torch.cuda.synchronize()
    h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
    hs = hashlib.sha256(st[0].cpu().numpy().tobytes()).hexdigest()[:12] if st is not None else "-"
    print(f"spec={os.environ.get('VTACO_CONV_SPEC', 'default')} R={R} {C1}+{C2}->{Cout} B={B}: {e0.elapsed_time(e1) * 10:.1f} us  out {h} stats {hs}  finite {bool(torch.isfinite(out).all())}", flush=True)
