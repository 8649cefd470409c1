"""Per-launch timeline of ONE forward (and, with `bwd`, backward) of the hand encoder's 2-D U-Net, from a rocprofv3 kernel trace.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o e -- python3 /root/repo/tools/probe/plane_unet_tl.py run [n_img] [bwd]
  python3 /root/repo/tools/probe/plane_unet_tl.py /tmp/pt/e_kernel_trace.csv
"""
import csv, os, sys
if len(sys.argv) > 1 and sys.argv[1] != "run":
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "vt_fill32" in r["Kernel_Name"]]
    rows = rows[marks[-2] + 1:marks[-1]]
    t0 = int(rows[0]["Start_Timestamp"])
    tot = 0.0
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += dur
        print(f"{name:46s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  dur {dur:7.1f} us  grid {r.get('Grid_Size', '?'):>8s} wg {r.get('Workgroup_Size', '?'):>5s}")
    print(f"kernel time {tot:.1f} us over {len(rows)} launches, span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd import ops, _lib
from vtaco_amd.encoder.unet import UNet
dev = torch.device("cuda:0")
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bwd = len(sys.argv) > 3 and sys.argv[3] == "bwd"
torch.manual_seed(0)
net = UNet(32, in_channels=32, depth=4, start_filts=32).to(dev)
x = torch.randn(n_img, 32, 32, 32, device=dev, requires_grad=bwd)
mark = torch.zeros(64, device=dev)


def fill():
    _lib.check(_lib.load().vt_fill32(mark.data_ptr(), 0, 256, _lib.stream_ptr()), "fill") if hasattr(_lib.load(), "vt_fill32") else None


def one():
    if bwd:
        for prm in net.parameters():        # as after optimizer.zero_grad(): the gradients are written, not accumulated
            prm.grad = None
        x.grad = None
        y = net(x)
        y.sum().backward()
    else:
        with torch.no_grad():
            net(x)


for _ in range(4):
    one()
torch.cuda.synchronize()
from vtaco_amd.ops import decode_range_clear
decode_range_clear()          # vt_fill32 marker
one()
decode_range_clear()
torch.cuda.synchronize()
