"""Timeline of ONE replay of the visual scene graph (Generator3D.generate_mesh_graphed: encode + decode + marching-cubes count, then the
emit kernels): run under ``rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o s`` this script generates the
traces; called with the two CSVs (kernel trace, memory-copy trace) it prints the last scene's nodes in start order with the gaps."""
import csv, os, sys
if len(sys.argv) > 1:
    rows = []
    for f in sys.argv[1:]:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Name") or r.get("Direction") or "copy"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]))
    rows.sort()
    # the last scene: back from the last row to the last voxel_build_kernel
    last = max(i for i, r in enumerate(rows) if "voxel_build" in r[2])
    first = last
    while first > 0 and rows[first][0] - rows[first - 1][1] < 20000 and "mc_faces" not in rows[first - 1][2]:
        first -= 1
    seq = rows[first:]
    t0, prev = seq[0][0], seq[0][0]
    busy = 0
    for s, e, n in seq:
        print(f"{n:50s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {(s - prev) / 1e3:6.1f} us")
        prev = e; busy += e - s
    print(f"span {(seq[-1][1] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us over {len(seq)} nodes")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd.bench_util import build_scene
from vtaco_amd.conv_onet.generation import Generator3D
dev = torch.device("cuda:0")
scene = build_scene(0, dev)
gen = Generator3D(scene["model"], device=dev, resolution0=32, padding=0.1, decode_precision="f16x3")
pc = scene["cloud"].to(dev)
for _ in range(6):
    gen.generate_mesh_graphed(pc)
torch.cuda.synchronize()
