"""Diagnostic (VT_DIAG_HB build only): per-wave shader-clock sums of the phases of the per-parity decoder-entry conv
(conv3d_gcr_up_kernel) on one layer (default 96 -> 32 at 64^3: C1 = 32 skip + C2 = 64 upsampled channels).
bash tools/build_variant.sh hb "-DVT_DIAG_HB"; VTACO_HIP_LIB=variants/lib_hb.so python tools/diag_conv_up.py [R C1 C2 Cout]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import _lib, ops
dev = torch.device("cuda:0")
R, C1, C2, Cout = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 32, 64, 32))]
g = torch.Generator().manual_seed(1)
x = torch.randn(1, R, R, R, C1, generator=g).to(dev)
low = torch.randn(1, R // 2, R // 2, R // 2, C2, generator=g).to(dev)
w = (torch.randn(Cout, C1 + C2, 3, 3, 3, generator=g) * 0.05).to(dev)
gamma, beta = torch.ones(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev)
xs, ls = ops.channel_stats(x), ops.channel_stats(low)
pf, ph, pu = ops.conv3d_pack(w), ops.conv3d_pack(w, precision="f16x3"), ops.conv3d_pack_up(w, C1)
if os.environ.get("DIAG_OLD"):
    pu = None
fn = lambda: ops.gn_conv3d_relu(x, xs, low, ls, gamma, beta, 8, pf, Cout, packed_w_f16x3=ph, packed_w_up=pu)
for _ in range(50):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fn()
e1.record(); torch.cuda.synchronize()
print(f"R={R} {C1}+{C2}->{Cout}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per layer (GroupNorm finalisation + conv, with stamps)")
lib = _lib.load()
if not hasattr(lib, "vt_diag_hb_read"):
    sys.exit(0)
SL = 16
n = 8192 * SL
buf = (ctypes.c_ulonglong * n)()
lib.vt_diag_hb_read.restype = ctypes.c_int
assert lib.vt_diag_hb_read(buf, ctypes.c_size_t(n)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, SL).astype(np.float64)
a = a[a[:, 15] > 0]
ghz = float(np.median(a[:, 15] / (a[:, 14] * 10.0)))
print(f"shader-clock counter: {ghz:.2f} GHz; wave lifetime {np.median(a[:, 14]) / 100:.1f} us")
def table(rows, title, names):
    print(f"{title}: {rows.shape[0]} waves")
    tot = rows[:, 15].sum()
    for i, nm in enumerate(names + ["-"] * (15 - len(names)) + ["TOTAL"]):
        if nm == "-":
            continue
        print(f"  {nm:44s} median {np.median(rows[:, i]):9.0f} counts per wave   ({100 * rows[:, i].sum() / tot:5.1f} %)  {np.median(rows[:, i]) / ghz / 1e3:6.2f} us")
per_wg = 16 if R >= 64 else 8
wv = np.arange(a.shape[0]) % per_wg
table(a[wv < per_wg // 2], "tap waves", ["prologue (to the first barrier)", "skip chunks: taps", "epilogue: relu + stores", "epilogue: statistics",
                                         "barrier after a skip chunk", "low chunks: taps", "-", "-", "barrier after a low chunk"])
table(a[wv >= per_wg // 2], "loader waves", ["prologue", "skip chunk: commit", "skip chunk: weights DMA + request issue", "wait (vmcnt)", "barrier (staging a skip chunk)",
                                            "-", "low chunk: commit", "low chunk: weights DMA + request issue", "barrier (staging a low chunk)"])
print("per wave index (median counts): " + " ".join(f"s{i}" for i in range(9)))
for wi in range(per_wg):
    r = a[wv == wi]
    print(f"  wave {wi:2d}: " + " ".join(f"{np.median(r[:, i]):7.0f}" for i in range(9)) + f"  total {np.median(r[:, 15]):8.0f}")
