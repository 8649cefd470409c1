import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd.conv_onet.models import decoder_dict
from vtaco_amd.bench_util import randomise_fc1
dev = "cuda:0"
torch.manual_seed(0)
adec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32).to(dev).eval()
B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 2048
ci = torch.randn(B, N, 32, device=dev) * (torch.rand(B, N, 1, device=dev) < 0.1)
c = torch.randn(B, N, 32, device=dev)
for _ in range(3):
    adec.fuser(ci, 1, c, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    adec.fuser(ci, 1, c, 1)
torch.cuda.synchronize(); print("ms per fuse (B=%d,N=2048), VTACO_PROJ_TILES=%s:" % (B, os.environ.get("VTACO_PROJ_TILES")), (time.perf_counter() - t0) / 5 * 1e3)
