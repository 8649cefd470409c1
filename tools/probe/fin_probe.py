"""vt_conv1x1_bwd_masked over eight 64^3 scenes: time per call (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd import ops
dev = "cuda:0"
n = 8 * 64 ** 3
y = torch.relu(torch.randn(n, 32, device=dev)); dout = torch.randn(n, 32, device=dev) * 1e-6; w = torch.randn(32, 32, device=dev) * 0.2
for _ in range(3):
    ops.conv1x1_bwd_masked(dout, y, w)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20):
    ops.conv1x1_bwd_masked(dout, y, w)
e1.record(); torch.cuda.synchronize()
print("mode", os.environ.get("VTACO_F1_MODE"), "ms per call: %.4f" % (e0.elapsed_time(e1) / 20))
