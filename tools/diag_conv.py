"""Diagnostic (VT_DIAG_HB build only: the stamps are no-op macros in the product build): per-wave shader-clock sums of the phases of the
persistent split-f16 conv kernels on one layer (default 32 -> 32 at 64^3).
bash tools/build_variant.sh hb "-DVT_DIAG_HB"; VTACO_HIP_LIB=variants/lib_hb.so [VTACO_CONV_SPEC=2] python tools/diag_conv.py [R C1 Cout]
Timing-only ablations of conv3d_gcr_hx_kernel (wrong results): add -DVT_HX_ABL=<bits> to the variant's flags (see unet3d.hip)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import _lib, ops
dev = torch.device("cuda:0")
R, C1, Cout = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), (int(sys.argv[2]) if len(sys.argv) > 2 else 32), (int(sys.argv[3]) if len(sys.argv) > 3 else 32)
g = torch.Generator().manual_seed(1)
x = torch.randn(1, R, R, R, C1, generator=g).to(dev)
if os.environ.get('DIAG_ZERO'):
    x = x * (torch.rand(1, R, R, R, 1, generator=g) < float(os.environ['DIAG_ZERO'])).to(dev)     # keep that fraction of the voxels, zero the rest
w = (torch.randn(Cout, C1, 3, 3, 3, generator=g) * 0.05).to(dev)
gamma, beta = torch.ones(C1, device=dev), torch.zeros(C1, device=dev)
xs = ops.channel_stats(x)
pf, ph = ops.conv3d_pack(w), ops.conv3d_pack(w, precision="f16x3")
fn = lambda: ops.gn_conv3d_relu(x, xs, None, None, gamma, beta, 8, pf, Cout, packed_w_f16x3=ph)
if os.environ.get("DIAG_SKIP"):
    # the first layer of the encoder on the bench scene's kind of input: zero away from a sphere shell, the empty blocks flagged
    from types import SimpleNamespace
    from vtaco_amd.bench_util import sphere_cloud
    pc = sphere_cloud(0)[0] / 1.1 + 0.5
    v = (pc * R).long().clamp(0, R - 1)
    idx = (v[:, 0] + R * (v[:, 1] + R * v[:, 2])).int().reshape(1, -1).to(dev)
    occ = torch.zeros(R ** 3, dtype=torch.bool, device=dev)
    occ[idx[0].long()] = True
    x = x * occ.reshape(1, R, R, R, 1)
    xs = ops.channel_stats(x)
    flags = ops.voxel_tile_flags(SimpleNamespace(idx=idx, B=1, T=idx.shape[1], R=R))
    if os.environ["DIAG_SKIP"] == "0":
        flags = torch.zeros_like(flags)
    print(f"flagged blocks: {int(flags.sum())} of {flags.numel()}")
    ss = ops.gn_scale_shift(xs, None, C1, 0, 1, R ** 3, gamma, beta, 8, 1e-5, dev)
    fn = lambda: ops.conv3d_gcr_skip(x, ss, ph, Cout, flags)
for _ in range(50):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fn()
e1.record(); torch.cuda.synchronize()
print(f"R={R} {C1}->{Cout}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per layer (GroupNorm finalisation + conv, with stamps)")
lib = _lib.load()
HX = os.environ.get("VTACO_CONV_SPEC", "1") == "2"
SL = 16                                 # per wave: phase sums 0..13, lifetime in 100 MHz ticks (14) and in shader-clock counts (15)
n = 8192 * SL
buf = (ctypes.c_ulonglong * n)()
lib.vt_diag_hb_read.restype = ctypes.c_int
assert lib.vt_diag_hb_read(buf, ctypes.c_size_t(n)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, SL).astype(np.float64)
a = a[a[:, 15] > 0]
ghz = float(np.median(a[:, 15] / (a[:, 14] * 10.0)))
print(f"shader-clock counter: {ghz:.2f} GHz (counts per 100 MHz tick, median over waves); wave lifetime {np.median(a[:, 14]) / 100:.1f} us")
def table(rows, title, names):
    print(f"{title}: {rows.shape[0]} waves")
    tot = rows[:, 15].sum()
    for i, nm in enumerate(names + ["-"] * (15 - len(names)) + ["TOTAL"]):
        if nm == "-":
            continue
        print(f"  {nm:44s} median {np.median(rows[:, i]):9.0f} counts per wave   ({100 * rows[:, i].sum() / tot:5.1f} %)  {np.median(rows[:, i]) / ghz / 1e3:6.2f} us"
              + (f"   (max {rows[:, i].max() / ghz / 1e3:.2f} us)" if nm == "TOTAL" or rows[:, i].max() > 2 * np.median(rows[:, i]) + 1 else ""))
if HX:
    # the support work in the tap waves' MFMA gaps (conv3d_gcr_hx_kernel)
    table(a, "waves (taps + their share of the staging)",
          ["setup to the first barrier", "prologue: second chunk's requests", "GroupNorm statistics -> scale / shift", "commit of chunk 0 (first round trip)",
           "wait: weights of chunk 0", "first barrier", "taps + pieces (14 k-steps x chunks)", "epilogue: relu + stores", "epilogue: statistics",
           "wait (vmcnt) before the chunk barrier", "chunk barrier", "prologue: item tables, tile origin", "prologue: weights DMA issue",
           "prologue: first chunk's requests"])
elif os.environ.get("VTACO_CONV_SPEC", "1") != "0":
    # specialised waves (conv3d_gcr_hw_kernel): in every workgroup the first half of the waves run taps, the second half load
    per_wg = 16 if R >= 64 else 8
    w = np.arange(a.shape[0]) % per_wg
    table(a[w < per_wg // 2], "tap waves", ["prologue (to the first barrier)", "taps (14 k-steps x chunks)", "epilogue: relu + stores", "epilogue: statistics", "barrier", "constant blocks: values"])
    table(a[w >= per_wg // 2], "loader waves", ["prologue: the first barrier", "commit (normalise, split, LDS)", "weights DMA + request issue", "wait (vmcnt)", "barrier",
                                                "-", "-", "-", "-", "prologue: start -> item tables, first tile", "prologue: weights DMA + two chunks' requests",
                                                "prologue: GroupNorm statistics -> table", "prologue: commit of chunk 0 + third request", "prologue: wait (weights of chunk 0)"])
else:
    print("the uniform-wave kernel carries no stamps")
