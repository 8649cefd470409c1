"""Marching-cubes call under rocprofv3 --kernel-trace: run `rocprofv3 --kernel-trace --output-format csv -d out -o m -- python3 tools/mc_timeline.py`
then `python3 tools/mc_timeline.py out/m_kernel_trace.csv` prints the kernel start/end offsets of the last call."""
import os, sys
if len(sys.argv) > 1 and sys.argv[1].endswith(".csv"):
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mc_" in r["Kernel_Name"] or "fill" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-7:]
    t0 = int(last[0]["Start_Timestamp"])
    for r in last:
        print("%-28s start %7.1f us  dur %6.1f us" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-28:], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops
from vtaco_amd.bench_util import build_scene
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
vol = sc["model"].decoder.decode_lattice(sc["grid"], 128, precision="bf16x3").view(128, 128, 128)
for _ in range(20):
    ops.marching_cubes(vol, None, rescale=(64, 1.1 / 128))
torch.cuda.synchronize()
