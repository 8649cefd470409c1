"""Diagnostic (VT_DIAG_PHASES build only): per-wave shader-clock sums of the phases of the two-brick lattice decode.
Build: (cd vtaco_amd/csrc && make clean && make CXXFLAGS_EXTRA=-DVT_DIAG_PHASES OUT=../variants/lib_phases.so), run with
VTACO_HIP_LIB=vtaco_amd/variants/lib_phases.so python tools/diag_phases.py [precision]."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import _lib
from vtaco_amd.bench_util import build_scene
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0")
sc = build_scene(0, dev)
dec, grid = sc["model"].decoder, sc["grid"]
nx = 128
for _ in range(300):
    dec.decode_lattice(grid, nx, precision=prec)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(50):
    dec.decode_lattice(grid, nx, precision=prec)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 50
lib = _lib.load()
n = 2048 * 8
buf = (ctypes.c_ulonglong * n)()
read = lib.vt_diag_phases_read_f16 if prec in ("f16x3", "f16f8") else lib.vt_diag_phases_read      # one buffer per translation unit
read.restype = ctypes.c_int
rc = read(buf, ctypes.c_size_t(n))
assert rc == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
a = a[a[:, 7] > 0]
names = ["loop/tail", "regs->LDS+idx", "gather", "fetch issue", "fc_p,c split,fc_c0", "5 blocks", "head+store", "TOTAL"]
if os.environ.get("VTACO_DECODE_ST3", "1") != "0" and prec in ("f16x3", "f16f8"):
    names = ["indices + DMA wait", "gather", "fetch issue", "fc_p operands, img", "prologue (c split, fc_p, fc_c0)", "blocks 0-3", "block 4 + heads", "TOTAL"]
tiles = 32768 / a.shape[0]
print(f"{prec}: {ms:.4f} ms per launch (with stamps), {a.shape[0]} waves, {tiles:.1f} double bricks per wave")
for i, nm in enumerate(names):
    print(f"  {nm:22s} median {np.median(a[:, i]) / tiles:9.0f} cycles per double brick   ({100 * a[:, i].sum() / a[:, 7].sum():5.1f} %)")
