"""What a kernel boundary costs on this runtime, and what the HIP runtime's knobs change: (a) N back-to-back launches of a tiny
kernel (torch's fill of 256 words), eager and as one hipGraph; (b) one encode (50 launches) eager / graphed; (c) the graphed scene.
Run once per environment (the knobs are read when the runtime starts):
  python tools/probe/launch_gap.py                       # this process
  python tools/probe/launch_gap.py --sweep               # child processes with HIP_FORCE_DEV_KERNARG / DEBUG_CLR_GRAPH_PACKET_CAPTURE set
"""
import os
import subprocess
import sys
import time

KNOBS = [{}, {"HIP_FORCE_DEV_KERNARG": "1"}, {"HIP_FORCE_DEV_KERNARG": "0"}, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
         {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}, {"HIP_FORCE_DEV_KERNARG": "1", "DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"},
         {"GPU_MAX_HW_QUEUES": "1"}, {"ROC_SIGNAL_POOL_SIZE": "4096"}]
if "--sweep" in sys.argv:
    for k in KNOBS:
        env = dict(os.environ); env.update(k)
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        print(k or "default", "->", out.stdout.strip().replace("\n", " | ") or out.stderr[-300:], flush=True)
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vtaco_amd import ops  # noqa: E402
from vtaco_amd.bench_util import build_scene  # noqa: E402
from vtaco_amd.conv_onet.generation import Generator3D  # noqa: E402

dev = torch.device("cuda:0")


def ev_ms(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


buf = torch.zeros(256, dtype=torch.int32, device=dev)
N = 200


def tiny():
    for _ in range(N):
        buf.zero_()


e = ev_ms(tiny)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    tiny()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        tiny()
gms = ev_ms(g.replay)
print(f"tiny kernel: eager {1e3 * e / N:.2f} us/launch, graph {1e3 * gms / N:.2f} us/launch")

sc = build_scene(0, dev)
model, pc = sc["model"], sc["cloud"].to(dev)
model.eval()
with torch.no_grad():
    enc = ev_ms(lambda: model.encode_inputs(pc))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        model.encode_inputs(pc)
        torch.cuda.synchronize()
        with torch.cuda.graph(g2, stream=s):
            out = model.encode_inputs(pc)
    encg = ev_ms(g2.replay)
    gen = Generator3D(model, device=dev, resolution0=32, padding=0.1)
    scene = ev_ms(lambda: gen.generate_mesh_graphed(pc), reps=20)
print(f"encode: eager {enc:.3f} ms, graph {encg:.3f} ms; graphed scene (with emit + read-back) {scene:.3f} ms")

# ---- where the graphed scene's time goes: host cost of the replay call, device span of the graph, the emit tail
nx = 128
gs = gen._scene_graph(pc.shape, nx)
torch.cuda.synchronize()
host, span, emit = [], [], []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record(); gs["graph"].replay(); b.record()
    t1 = time.perf_counter()
    b.synchronize()
    t2 = time.perf_counter()
    v, f, _ = ops.mc_emit(gs["vol"], gs["ws"], rescale=(nx / 2, 1.1 / nx))
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    host.append(1e3 * (t1 - t0)); span.append(a.elapsed_time(b)); emit.append(1e3 * (t3 - t2))
med = lambda x: sorted(x)[len(x) // 2]
print(f"scene graph: replay() host call {med(host):.3f} ms, device span {med(span):.3f} ms, mc_emit (read-back + 2 launches + sync) {med(emit):.3f} ms")
