"""Diagnostic (VT_DIAG_HB build only; the stamps live in tools/probe/conv_diag.patch: `patch vtaco_amd/csrc/unet3d.hip tools/probe/conv_diag.patch`, build the variant, `patch -R` afterwards): per-wave shader-clock sums of the phases of the persistent split-f16 conv on one
32->32 layer at 64^3.  bash tools/build_variant.sh hb "-DVT_DIAG_HB"; VTACO_HIP_LIB=variants/lib_hb.so python tools/diag_conv.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import _lib, ops
dev = torch.device("cuda:0")
R, C1, Cout = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), (int(sys.argv[2]) if len(sys.argv) > 2 else 32), (int(sys.argv[3]) if len(sys.argv) > 3 else 32)
g = torch.Generator().manual_seed(1)
x = torch.randn(1, R, R, R, C1, generator=g).to(dev)
w = (torch.randn(Cout, C1, 3, 3, 3, generator=g) * 0.05).to(dev)
gamma, beta = torch.ones(C1, device=dev), torch.zeros(C1, device=dev)
xs = ops.channel_stats(x)
pf, ph = ops.conv3d_pack(w), ops.conv3d_pack(w, precision="f16x3")
fn = lambda: ops.gn_conv3d_relu(x, xs, None, None, gamma, beta, 8, pf, Cout, packed_w_f16x3=ph)
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
n = 8192 * 8
buf = (ctypes.c_ulonglong * n)()
lib.vt_diag_hb_read.restype = ctypes.c_int
assert lib.vt_diag_hb_read(buf, ctypes.c_size_t(n)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
raw = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)
raw = raw[raw[:, 7] > 0]
a = a[a[:, 7] > 0]
names = ["prologue", "commit + DMA issue", "fetch issue", "taps", "tile epilogue", "wait DMA", "barrier", "TOTAL"]
tap_names = ["prologue (to the first barrier)", "taps (14 k-steps x chunks)", "epilogue: relu + stores", "epilogue: statistics", "barrier", "-", "-", "TOTAL"]
load_names = ["prologue", "commit (normalise, split, LDS)", "weights DMA + fetch issue", "wait (vmcnt)", "barrier", "-", "-", "TOTAL"]
def table(rows, title, names=names):
    print(f"{title}: {rows.shape[0]} waves")
    for i, nm in enumerate(names):
        if nm == "-":
            continue
        print(f"  {nm:20s} median {np.median(rows[:, i]):9.0f} cycles per wave   ({100 * rows[:, i].sum() / rows[:, 7].sum():5.1f} %)")
if os.environ.get("VTACO_CONV_SPEC", "1") != "0":
    # specialised waves: in every workgroup the first half of the waves run taps (slots 3 = taps, 4 = taps + tile epilogue of the
    # tile's last chunk, 6 = barrier), the second half load (1 = commit + DMA issue, 2 = fetch issue, 5 = wait, 6 = barrier)
    per_wg = 16 if R >= 64 else 8
    w = np.arange(a.shape[0]) % per_wg
    table(a[w < per_wg // 2], "tap waves", tap_names)
    table(a[w >= per_wg // 2], "loader waves", load_names)
else:
    table(a, "all waves")
