"""Is the tactile Resnet18's share of a training step launch-bound?  Forward + backward of the per-scene loop (8 scenes x five
320x240 images, train-mode BatchNorm): eager against torch.cuda.make_graphed_callables (forward and backward as hipGraphs),
with the gradients and the running statistics compared."""
import copy, sys, time
import torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.encoder import encoder_dict
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = encoder_dict["Resnet18"](num_classes=32).to(dev).train()
ref = copy.deepcopy(net)
imgs = torch.rand(8, 5, 3, 320, 240, device=dev)

def loop(m, call):
    m.zero_grad(set_to_none=True)
    out = torch.cat([call(imgs[b]).reshape(1, 5, -1) for b in range(8)])
    out.sum().backward()
    return out

def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

print(f"eager: {timed(lambda: loop(ref, ref)):.2f} ms")
g = torch.cuda.make_graphed_callables(net, (imgs[0].clone(),))
print(f"graphed: {timed(lambda: loop(net, g)):.2f} ms")
# same state, one step each: outputs, gradients, running statistics
net2 = encoder_dict["Resnet18"](num_classes=32).to(dev).train()
ref2 = copy.deepcopy(net2)
g2 = torch.cuda.make_graphed_callables(net2, (imgs[0].clone(),))
ref2.load_state_dict(net2.state_dict())           # (the capture's warm-up iterations moved the running statistics)
o1, o0 = loop(net2, g2), loop(ref2, ref2)
print("out", float((o1 - o0).abs().max()), "grad", max(float((a.grad - b.grad).abs().max()) for a, b in zip(net2.parameters(), ref2.parameters())),
      "running", max(float((a - b).abs().max()) for a, b in zip(net2.buffers(), ref2.buffers())))
