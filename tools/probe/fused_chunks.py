"""Generator3D._eval_lattice_tactile with decoder attention_local over a 128^3 lattice (points_batch_size 2048): whole chunks batched
per call (FUSED_CHUNKS_PER_CALL = 256) against one chunk per call.  python tools/probe/fused_chunks.py"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from vtaco_amd.bench_util import build_scene, randomise_fc1
from vtaco_amd.conv_onet.generation import Generator3D
from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
dev = torch.device("cuda:0")
sc = build_scene(0, dev); grid = sc["grid"]
torch.manual_seed(0)
adec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32).to(dev).eval(); randomise_fc1(adec, 3)
nx, N = 128, 2048
agen = Generator3D(ConvolutionalOccupancyNetwork(adec, None, device=dev), device=dev, resolution0=nx // 4, padding=0.1, points_batch_size=N, with_img=True)
gs = torch.Generator().manual_seed(2); tips = torch.randn(5, 1, 3, generator=gs)
setup = {'feats': torch.randn(5, 32, generator=gs), 'anchors': 0.3 * tips / tips.norm(dim=-1, keepdim=True), 'success': torch.tensor([1, 0, 1, 1, 1], dtype=torch.uint8), 'mode': 'nearest', 'radius': 0.08, 'count': torch.ones(5, dtype=torch.int32)}
for per in (256, 1):
    agen.FUSED_CHUNKS_PER_CALL = per
    with torch.no_grad():
        agen._eval_lattice_tactile({"grid": grid}, nx, setup); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): agen._eval_lattice_tactile({"grid": grid}, nx, setup)
        torch.cuda.synchronize(); print(per, "chunks per call:", (time.perf_counter() - t0) / 2 * 1e3, "ms")
