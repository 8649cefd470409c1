"""Per-launch kernel timeline of ONE encode (PointNet + UNet3D), from a rocprofv3 kernel trace.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/et -o e -- python3 /root/repo/tools/enc_timeline.py
  python3 /root/repo/tools/enc_timeline.py /tmp/et/e_kernel_trace.csv
"""
import csv
import os
import sys

if len(sys.argv) > 1:
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the script brackets the last encode with two vt_fill32-free marker launches of a tiny torch kernel: take the
    # launches after the LAST occurrence of the first encode kernel (voxel_build)
    starts = [i for i, r in enumerate(rows) if "voxel_build" in r["Kernel_Name"]]
    rows = rows[starts[-1]:]
    t0 = int(rows[0]["Start_Timestamp"])
    tot = 0.0
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += dur
        g = r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")
        print(f"{name:46s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  dur {dur:7.1f} us  grid {g[0]:>8s} wg {g[1]:>5s}")
    print(f"kernel time {tot:.1f} us over {len(rows)} launches, span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd.bench_util import build_scene
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
model, pc = sc["model"], sc["cloud"].to(dev)
with torch.no_grad():
    for _ in range(4):
        model.encode_inputs(pc)
    torch.cuda.synchronize()
