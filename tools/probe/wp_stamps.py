"""Phase stamps of decode_wide_p_kernel's stage waves (variant build: bash tools/build_variant.sh wp "-DVT_DIAG_WP";
VTACO_HIP_LIB=variants/lib_wp.so VTACO_WP_PRINT=1 python3 tools/probe/wp_stamps.py): one 128^3 lattice at 64 / 32 / 5."""
import sys
import torch
sys.path.insert(0, '/root/repo')
from vtaco_amd.conv_onet.models import decoder_dict
dev = torch.device('cuda:0')
torch.manual_seed(1)
dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=64, n_blocks=5).to(dev).eval()
grid = torch.randn(1, 32, 64, 64, 64, device=dev)
with torch.no_grad():
    dec.decode_lattice(grid, 128, precision="f16x3")
    torch.cuda.synchronize()
    print("---- second launch ----", file=sys.stderr, flush=True)
    dec.decode_lattice(grid, 128, precision="f16x3")
    torch.cuda.synchronize()
