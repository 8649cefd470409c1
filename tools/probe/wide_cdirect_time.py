"""Where the 256 / 128 / 5 decoder's time goes: the whole lattice pass (in-kernel gather) against the gather alone (vt_sample_grid) and the
MLP alone on given features (vt_decode_mlp_fwd_wide_f16x3).  GPU box: python3 tools/probe/wide_cdirect_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd import ops
from vtaco_amd.bench_util import randomise_fc1
from vtaco_amd.conv_onet.models import decoder_dict
dev = torch.device("cuda:0")
nx = 128


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for hidden, cd in ((256, 128), (128, 64)):
    torch.manual_seed(1)
    dec = decoder_dict['simple_local'](dim=3, c_dim=cd, hidden_size=hidden, n_blocks=5).to(dev).eval()
    randomise_fc1(dec, 4)
    grid = ops.grid_to_channels_last(torch.randn(1, cd, 64, 64, 64, device=dev))
    with torch.no_grad():
        t_all = timed(lambda: dec.decode_lattice(grid, nx, precision="f16x3"))
        t_s = timed(lambda: ops.sample_grid(grid, None, lattice=(nx, 1.1, 0, nx ** 3)))
        c = ops.sample_grid(grid, None, lattice=(nx, 1.1, 0, nx ** 3))
        from vtaco_amd.common import make_3d_grid
        pts = (1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).to(dev).unsqueeze(0)
        blob = dec._blob(precision="wide_f16x3")
        t_m = timed(lambda: ops.decode_mlp_fwd(c, blob, pts, precision="wide_f16x3", wide=(hidden, 5, False)))
        a = dec.decode_lattice(grid, nx, precision="f16x3").reshape(-1)
        b = ops.decode_mlp_fwd(c, blob, pts, precision="wide_f16x3", wide=(hidden, 5, False)).reshape(-1)
    print(f"{hidden} / {cd} / 5 over 128^3: whole pass {t_all:.2f} ms; gather alone (vt_sample_grid) {t_s:.2f} ms; MLP alone on given features {t_m:.2f} ms; "
          f"max |pass - (gather, MLP)| = {float((a - b).abs().max()):.2e}")
