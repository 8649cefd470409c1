import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.bench_util import build_train_case
dev = torch.device("cuda:0")
model, trainer, batch, vf = build_train_case(dev, 0, scenes=8, grad_sync=False)
np.random.seed(0)
for _ in range(2): trainer.train_step(batch, vf)
torch.cuda.synchronize()
print("=====STEP=====", file=sys.stderr, flush=True)
trainer.train_step(batch, vf)
torch.cuda.synchronize()
print("=====END=====", file=sys.stderr, flush=True)
