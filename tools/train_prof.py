"""Where one VTacO training step (bench.py's train_step section, BASELINE config 4's per-GPU share) spends its time: wall-clock of
the step's phases with a device synchronisation after each (so phases do not overlap as they do in the real step), and cProfile's
view of the host side.  Run under rocprofv3 --kernel-trace --stats for the kernel view."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd.bench_util import build_train_case  # noqa: E402

dev = torch.device("cuda:0")
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, grad_sync=False)
np.random.seed(0)
for _ in range(3):
    trainer.train_step(batch, vf)
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, 1e3 * (time.perf_counter() - t0)


model.train()
res = {}
trainer.optimizer.zero_grad()
s, res["t2d_samples (t2d net fwd + host contact clouds + winding numbers)"] = timed(lambda: trainer._t2d_samples(batch, vf, normalise_depth=False))
c_img, res["encode_img (Resnet18, 40 images 320x240)"] = timed(lambda: model.encode_img_inputs(s['imgs']))
c, res["encode_inputs (PointNet + UNet3D, 8 scenes)"] = timed(lambda: model.encode_inputs(s['inputs']))
feat = torch.gather(c_img, 1, s['finger'].clamp(min=0).unsqueeze(-1).expand(-1, -1, c_img.shape[2]))
c_img_all = torch.where((s['finger'] >= 0).unsqueeze(-1), feat, torch.ones_like(feat))
logits, res["decode_img"] = timed(lambda: model.decode_img(s['p_sample'], c, c_img_all).logits)
c_hand, res["encode_hand (plane PointNet + 2-D U-Net + MANO)"] = timed(lambda: model.encode_hand_inputs(s['inputs']))
loss = torch.nn.functional.l1_loss(logits, s['occ']) + torch.nn.functional.mse_loss(c_hand['mano_param'], batch['points.mano'].to(dev).float()) \
    + torch.nn.functional.mse_loss(c_hand['mano_verts'], batch['points.pc_hand'].to(dev).float())
_, res["backward"] = timed(lambda: loss.backward())
_, res["optimizer.step (Adam, 20 M parameters)"] = timed(lambda: trainer.optimizer.step())
_, res["whole train_step (overlapped)"] = timed(lambda: trainer.train_step(batch, vf))
for k, v in res.items():
    print(f"{v:9.2f} ms  {k}")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    trainer.train_step(batch, vf)
torch.cuda.synchronize()
pr.disable()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(35)
print(buf.getvalue()[:6000])
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_callers("'item'|'cpu'|synchronize")     # who waits for the device
print(buf.getvalue()[:5000])

# ---- the tactile feature encoder alone (host PyTorch-ROCm / MIOpen): default vs find mode, NCHW vs channels_last
if os.environ.get("VTACO_PROF_RESNET"):
    from vtaco_amd.encoder import encoder_dict
    imgs = batch["inputs.img"].to(dev)

    def run(net, x):
        net.zero_grad()
        out = torch.cat([net(x[b]).reshape(1, 5, -1) for b in range(x.shape[0])])
        out.sum().backward()
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for cl in (False, True):
            net = encoder_dict["Resnet18"](num_classes=32).to(dev).train()
            x = imgs
            if cl:
                net = net.to(memory_format=torch.channels_last)
                x = imgs.contiguous(memory_format=torch.channels_last_3d) if False else imgs
            t0 = time.perf_counter()
            run(net, x)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            run(net, x)
            _, ms = timed(lambda: run(net, x))
            print(f"Resnet18 fwd+bwd 8x5 images: benchmark={bench} channels_last={cl}: {ms:.1f} ms (first call {first:.2f} s)")
