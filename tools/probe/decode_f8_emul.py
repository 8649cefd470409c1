"""CPU emulation of candidate arithmetic for the lattice decode's dense layers on the golden decoder (tests/golden/g1_decode.npz):
every layer y = W x + b is evaluated as  W_hi x_hi  (f16 operands, exact products, wide accumulation)  +  a correction
W_lo x_hi + W_hi x_lo  whose operands are rounded to a narrow format (fp8 e4m3 with a fixed power-of-two pre-scale, fp6 e2m3
with a per-point block scale, or f16 = the shipped split-f16 scheme).  Prints the max abs logit error against float64."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vtaco_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def f16_rtz(x):
    h = x.astype(np.float16)
    hf = h.astype(np.float64)
    over = np.abs(hf) > np.abs(x)
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float64)


def q_e4m3(v):
    """OCP e4m3fn, round to nearest even, saturating at 448."""
    a = np.abs(v)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    q = np.round(a / step) * step
    q = np.minimum(q, 448.0)
    return np.sign(v) * q


def q_e2m3_block(v):
    """fp6 e2m3 with one power-of-two scale per row (last axis = the block)."""
    amax = np.max(np.abs(v), axis=-1, keepdims=True)
    s = 2.0 ** np.ceil(np.log2(np.maximum(amax, 1e-300) / 7.5))
    a = np.abs(v) / s
    e = np.clip(np.floor(np.log2(np.maximum(a, 1e-300))), 0, 2)
    step = 2.0 ** (e - 3)
    q = np.minimum(np.round(a / step) * step, 7.5)
    return np.sign(v) * q * s


def run(mode, sx=0, sw=0, wlo_bits=11):
    z = np.load(os.path.join(ROOT, "tests", "golden", "g1_decode.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    grid = torch.from_numpy(z["grid"])
    nx = 32
    pts = (1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).unsqueeze(0)
    c = orc.trilinear_sample(grid, pts, 0.1)[0].numpy().astype(np.float64)          # [N, 32]
    p = pts[0].numpy().astype(np.float64)
    W = {k: v.numpy().astype(np.float64) for k, v in sd.items()}

    def dense(name, x, relu):
        w, b = W[name + ".weight"], W[name + ".bias"]
        if relu:
            x = np.maximum(x, 0.0)
        if mode == "f64":
            return x @ w.T + b
        x = x.astype(np.float32).astype(np.float64)
        xh = f16_rtz(x); xl = x - xh
        wh = w.astype(np.float16).astype(np.float64); wl = w - wh
        main = xh @ wh.T
        if mode == "f16x3":
            xl16 = f16_rtz(xl)
            wl16 = wl.astype(np.float16).astype(np.float64)
            corr = xh @ wl16.T + xl16 @ wh.T
        elif mode == "f16x2":
            corr = 0.0
        elif mode == "f8":
            # B side: x_hi * 2^-sx and x_lo * 2^(11-sx); A side: W_lo * 2^(11+sw) and W_hi * 2^sw; result * 2^(sx-sw-11)
            bxh = q_e4m3(xh * 2.0 ** -sx); bxl = q_e4m3(xl * 2.0 ** (11 - sx))
            awl = q_e4m3(wl * 2.0 ** (11 + sw)); awh = q_e4m3(wh * 2.0 ** sw)
            corr = (bxh @ awl.T + bxl @ awh.T) * 2.0 ** (sx - sw - 11)
        elif mode == "f6":
            # per-point block = the 32 k-slots of a lane half: 16 channels' x_hi and x_lo * 2^11 (two blocks per point)
            corr = 0.0
            for half in (slice(0, 16), slice(16, 32)):
                blk = q_e2m3_block(np.concatenate([xh[:, half], xl[:, half] * 2.0 ** 11], axis=1))
                wblk = q_e2m3_block(np.concatenate([wl[:, half] * 2.0 ** 11, wh[:, half]], axis=1))
                corr = corr + (blk @ wblk.T) * 2.0 ** -11
        return main + corr + b

    net = p @ W["fc_p.weight"].T + W["fc_p.bias"]
    for i in range(5):
        net = net + dense(f"fc_c.{i}", c, False)
        h = dense(f"blocks.{i}.fc_0", net, True)
        net = net + dense(f"blocks.{i}.fc_1", h, True)
    out = np.maximum(net, 0.0) @ W["fc_out.weight"].T + W["fc_out.bias"]
    return out[:, 0], net


ref, net_ref = run("f64")
print("logit range", ref.min(), ref.max(), " |net| max", np.abs(net_ref).max())
for mode, kw in [("f16x3", {}), ("f16x2", {}), ("f6", {}), ("f8", dict(sx=0, sw=0)), ("f8", dict(sx=2, sw=0)), ("f8", dict(sx=4, sw=0)),
                 ("f8", dict(sx=0, sw=2)), ("f8", dict(sx=2, sw=2)), ("f8", dict(sx=-2, sw=2)), ("f8", dict(sx=0, sw=4))]:
    got, _ = run(mode, **kw)
    print(f"{mode:6s} {kw}: max abs logit error {np.abs(got - ref).max():.3e}   rms {np.sqrt(np.mean((got - ref) ** 2)):.3e}")
