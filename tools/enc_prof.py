import os, sys, torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.bench_util import build_scene
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
model = sc["model"]; pc = sc["cloud"].to(dev)
with torch.no_grad():
    for _ in range(3): model.encode_inputs(pc)
    torch.cuda.synchronize()
    import time; t0=time.perf_counter()
    for _ in range(20): model.encode_inputs(pc)
    torch.cuda.synchronize(); print("encode ms", (time.perf_counter()-t0)/20*1e3)
