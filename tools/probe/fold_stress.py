"""Stress of the GroupNorm accumulator rows (GnOut / GnIn): the shipped encoder on B = 1 and B = 4 scenes, 300 eager encodes and 300
graph replays each -- every output must equal the first bit for bit (integer accumulators: no dependence on the order of arrival),
and the launch path (VTACO_GN_FOLD=0) must agree to rounding."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd.bench_util import build_scene, sphere_cloud
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
enc = sc["model"].encoder
for B in (1, 4):
    pc = torch.cat([sphere_cloud(10 + b).to(dev) for b in range(B)]) if B > 1 else sc["cloud"].to(dev)
    with torch.no_grad():
        first = enc(pc)["grid"].clone()
        bad = 0
        for _ in range(300):
            bad += int(not torch.equal(enc(pc)["grid"], first))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            enc(pc)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = enc(pc)["grid"]
        badg = 0
        for _ in range(300):
            g.replay()
            torch.cuda.synchronize()
            badg += int(not torch.equal(out, first))
        os.environ["VTACO_GN_FOLD"] = "0"
        ref = enc(pc)["grid"].clone()
        os.environ.pop("VTACO_GN_FOLD")
    print(f"B={B}: eager mismatches {bad}/300, graph mismatches {badg}/300, |rows - launches| max {float((first - ref).abs().max()):.2e} "
          f"(scale {float(ref.abs().max()):.2f})")
