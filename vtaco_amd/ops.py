"""Host-side launchers for the HIP kernels behind the C ABI (include/vtaco_hip.h).

Thin and allocation-explicit: every function takes torch HIP tensors, hands raw
device pointers + the current stream to libvtaco_hip.so and returns torch
tensors.  No function here computes anything with torch ops.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import VtError, check, dev_ptr, stream_ptr


# ---- hipGraph capture support -------------------------------------------------------------------------------------
# A captured graph holds RAW POINTERS to everything its launches read: derived buffers (packed weights, the decoder
# blob, the UNet3D workspace) live in caches that re-allocate when a weight changes or another shape runs.  While a
# capture is being prepared every launcher hands such tensors to ``keep_for_graph``; the graph's owner stores the list
# next to the graph, so the memory cannot be recycled under a live graph.
_graph_keep = None


class graph_keepalive:
    """``with ops.graph_keepalive() as keep:`` -- collects every derived tensor the launchers inside touch."""

    def __enter__(self):
        global _graph_keep
        self._prev, _graph_keep = _graph_keep, []
        return _graph_keep

    def __exit__(self, *exc):
        global _graph_keep
        _graph_keep = self._prev
        return False


def keep_for_graph(*tensors):
    if _graph_keep is not None:
        _graph_keep.extend(t for t in tensors if t is not None)


def blob_floats(hidden=32, c_dim=32, n_blocks=5):
    n = _lib.load().vt_decoder_blob_bytes(hidden, c_dim, n_blocks)
    if n == 0:
        raise VtError(f"decoder shape hidden={hidden}, c_dim={c_dim}, n_blocks={n_blocks} is not built "
                      "(gfx950 kernels cover the shipped VTacO shape 32/32/5)")
    return n // 4


PRECISIONS = ("f32", "bf16x3", "f16x3", "f16f8")
SPLIT_PRECISIONS = ("bf16x3", "f16x3", "f16f8")   # dense layers on the 16-bit matrix core with hi + lo operands
# "f16f8": f16 hi products + ONE fp8 MFMA for both correction products of a layer (vt_decode_fwd_f16f8: ~4e-5 on the golden logits,
# the fastest form); it exists for lattice slabs only -- see f16f8_covers -- and LocalDecoder falls back to "f16x3" elsewhere


def f16f8_covers(grid, lattice, padding=0.1):
    """True if vt_decode_fwd_f16f8 covers this lattice slab (whole x-plane pairs, nx % 8 == 0, < 0.55 voxels per step)."""
    if lattice is None:
        return False
    nx, box, first, count = lattice
    B, C, D, H, W = grid.shape
    return bool(_lib.load().vt_decode_f16f8_covers(D, C, int(nx), float(box), int(first), int(count), float(padding)))


def pack_decoder(fc_p_w, fc_p_b, fc_c, blocks, fc_out, fc_out2=None, out=None, transposed=False, precision="f32"):
    """Repack decoder parameters into the MFMA-fragment blob (vt_decoder_pack; with
    ``precision="bf16x3"`` / ``"f16x3"`` the split blob of vt_decoder_pack_bf16x3 / _f16x3), or with
    ``transposed=True`` into the transposed-weight blob of the backward (vt_decoder_pack_t).

    fc_c: list of (weight, bias); blocks: list of (fc0_w, fc0_b, fc1_w, fc1_b);
    fc_out / fc_out2: (weight, bias).  ``fc_p_w`` is fc_p.weight [H,3] or
    fc_p_img.weight [H,3+C].
    """
    lib = _lib.load()
    hidden, p_in = fc_p_w.shape
    c_dim = fc_c[0][0].shape[1]
    nb = len(blocks)
    if nb > _lib.VT_MAX_BLOCKS:
        raise VtError(f"n_blocks={nb} exceeds VT_MAX_BLOCKS")
    keep = []

    def ptr(t, name):
        t = t.detach()
        if not t.is_contiguous():
            t = t.contiguous()
        keep.append(t)
        return dev_ptr(t, name)

    prm = _lib.DecoderParams()
    prm.hidden, prm.c_dim, prm.n_blocks, prm.p_in = hidden, c_dim, nb, p_in
    prm.fc_p_w, prm.fc_p_b = ptr(fc_p_w, "fc_p.weight"), ptr(fc_p_b, "fc_p.bias")
    for i, (w, b) in enumerate(fc_c):
        prm.fc_c_w[i], prm.fc_c_b[i] = ptr(w, f"fc_c.{i}.weight").value, ptr(b, f"fc_c.{i}.bias").value
    for i, (w0, b0, w1, b1) in enumerate(blocks):
        prm.fc0_w[i], prm.fc0_b[i] = ptr(w0, "fc_0.weight").value, ptr(b0, "fc_0.bias").value
        prm.fc1_w[i], prm.fc1_b[i] = ptr(w1, "fc_1.weight").value, ptr(b1, "fc_1.bias").value
    prm.fc_out_w, prm.fc_out_b = ptr(fc_out[0], "fc_out.weight"), ptr(fc_out[1], "fc_out.bias")
    if fc_out2 is not None:
        prm.fc_out2_w, prm.fc_out2_b = ptr(fc_out2[0], "fc_out_contact.weight"), ptr(fc_out2[1], "fc_out_contact.bias")
    if transposed:
        n = lib.vt_decoder_blob_t_bytes(hidden, c_dim, nb) // 4
        if n == 0:
            raise VtError("decoder shape not built (32/32/5 only)")
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=fc_p_w.device)
        check(lib.vt_decoder_pack_t(ctypes.byref(prm), dev_ptr(out, "blob_t"), n * 4, stream_ptr()), "vt_decoder_pack_t")
        return out
    if precision == "wide_f16x3":
        # the same shapes on the f16 matrix core with split operands (vt_decode_fwd_wide_f16x3): its own fragment format
        n = lib.vt_decoder_wide_blob_f16x3_bytes(hidden, c_dim, nb, p_in) // 4
        if n == 0:
            raise VtError(f"decoder shape hidden={hidden}, c_dim={c_dim}, n_blocks={nb} is not built: hidden_size and c_dim must be "
                          f"multiples of 32 up to 256, n_blocks <= {_lib.VT_MAX_BLOCKS}")
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=fc_p_w.device)
        check(lib.vt_decoder_pack_wide_f16x3(ctypes.byref(prm), dev_ptr(out, "blob"), n * 4, stream_ptr()), "vt_decoder_pack_wide_f16x3")
        return out
    if precision == "wide":
        # the general-shape kernel (vt_decode_fwd_wide): hidden / c_dim multiples of 32 up to 256, weights streamed in fragment order
        n = lib.vt_decoder_wide_blob_bytes(hidden, c_dim, nb, p_in) // 4
        if n == 0:
            raise VtError(f"decoder shape hidden={hidden}, c_dim={c_dim}, n_blocks={nb} is not built: hidden_size and c_dim must be "
                          f"multiples of 32 up to 256, n_blocks <= {_lib.VT_MAX_BLOCKS}")
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=fc_p_w.device)
        check(lib.vt_decoder_pack_wide(ctypes.byref(prm), dev_ptr(out, "blob"), n * 4, stream_ptr()), "vt_decoder_pack_wide")
        return out
    if precision not in PRECISIONS:
        raise VtError(f"precision must be one of {PRECISIONS} (got {precision!r})")
    n = blob_floats(hidden, c_dim, nb)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=fc_p_w.device)
    if precision in SPLIT_PRECISIONS:
        name = "vt_decoder_pack_" + precision
        check(getattr(lib, name)(ctypes.byref(prm), dev_ptr(out, "blob"), n * 4, stream_ptr()), name)
    else:
        check(lib.vt_decoder_pack(ctypes.byref(prm), dev_ptr(out, "blob"), n * 4, stream_ptr()), "vt_decoder_pack")
    return out


RANGE_HALF, RANGE_FP8, RANGE_LOGIT = 1, 2, 4


def decode_range_status(reset=True):
    """The current device's range-guard word (vt_decode_range_status).  Bit 0 (RANGE_HALF): a half-precision lattice decode
    ("f16x3" / "f16f8") since the last reset met activations at the edge of the half range (its hi operand saturated at 65504:
    the logits of those launches are not to be trusted).  Bit 1 (RANGE_FP8): an "f16f8" decode met activations >= 1024, where
    the fp8 copies of its correction products begin to clip, bit 2 (RANGE_LOGIT): an "f16f8" decode wrote a logit beyond 2.5 in
    magnitude (its error is relative, ~3e-5 |logit|) -- in both cases its 1e-4 contract ends and "f16x3" is the form to use.
    Synchronises the stream."""
    word = ctypes.c_uint32(0)
    check(_lib.load().vt_decode_range_status(ctypes.byref(word), int(bool(reset)), stream_ptr()), "vt_decode_range_status")
    return int(word.value)


def decode_range_clear():
    """Clear the current device's range-guard word without reading it (an asynchronous 4-byte fill on the current stream: what
    the generator does when a scene begins, so that bits left by earlier launches are not attributed to it)."""
    check(_lib.load().vt_decode_range_status(None, 1, stream_ptr()), "vt_decode_range_status")


def decode_last_clock(workgroups=False):
    """Clock evidence of the last lattice decode launch on the current device (vt_decode_last_clock): workgroup 0's lifetime in
    shader cycles and in ticks of the constant-rate counter -> {"shader_mhz", "wg0_us", "shader_cycles"}; with
    ``workgroups=True`` also the launch's shape in time from every workgroup's (start, end) stamps: "span_us" (first start to
    last end: the kernel's duration as the chip saw it), "start_spread_us" (the dispatch ramp), "wg_us_min/median/max" and
    the raw stamps as "wg_ticks" ([n, 2] list, ticks of "ref_khz").  Synchronises."""
    cyc, ref, khz, n = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int(0), ctypes.c_int(0)
    cap = 512 if workgroups else 0
    buf = (ctypes.c_uint64 * (2 * cap))() if cap else None
    check(_lib.load().vt_decode_last_clock(ctypes.byref(cyc), ctypes.byref(ref), ctypes.byref(khz), buf, cap, ctypes.byref(n),
                                           stream_ptr()), "vt_decode_last_clock")
    if ref.value == 0 or khz.value <= 0:
        return None
    tick_us = 1e3 / khz.value
    us = ref.value * tick_us
    res = {"shader_mhz": cyc.value / us, "wg0_us": us, "shader_cycles": int(cyc.value), "ref_khz": int(khz.value)}
    if workgroups and n.value > 0:
        st = [(buf[2 * i], buf[2 * i + 1]) for i in range(n.value)]
        t0 = min(a for a, _ in st)
        dur = sorted((b - a) * tick_us for a, b in st)
        res.update({"workgroups": n.value, "span_us": (max(b for _, b in st) - t0) * tick_us,
                    "start_spread_us": (max(a for a, _ in st) - t0) * tick_us,
                    "wg_us_min": dur[0], "wg_us_median": dur[len(dur) // 2], "wg_us_max": dur[-1],
                    "wg_ticks": [[int(a - t0), int(b - t0)] for a, b in st]})
    return res


def is_channels_last_grid(grid):
    """True if ``grid`` [B,C,D,H,W] is laid out b,z,y,x,c in memory."""
    B, C, D, H, W = grid.shape
    return grid.stride() == (D * H * W * C, 1, H * W * C, W * C, C)


def grid_to_channels_last(grid):
    """[B,C,D,H,W] contiguous -> tensor of the SAME shape whose memory is
    [B,D,H,W,C] (torch.channels_last_3d), via vt_grid_to_channels_last."""
    if is_channels_last_grid(grid):
        return grid
    g = grid.detach()
    if not g.is_contiguous():
        g = g.contiguous()
    B, C, D, H, W = g.shape
    out = torch.empty((B, D, H, W, C), dtype=torch.float32, device=g.device)
    check(_lib.load().vt_grid_to_channels_last(dev_ptr(g, "grid"), dev_ptr(out, "grid_cl"), B, C, D, H, W, stream_ptr()),
          "vt_grid_to_channels_last")
    return out.permute(0, 4, 1, 2, 3)


def grid_from_channels_last(grid_cl):
    """Inverse of :func:`grid_to_channels_last`: returns a contiguous NCDHW tensor."""
    B, C, D, H, W = grid_cl.shape
    if not is_channels_last_grid(grid_cl):
        raise VtError("grid_from_channels_last: input is not channels-last")
    out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=grid_cl.device)
    src = grid_cl.permute(0, 2, 3, 4, 1)
    check(_lib.load().vt_grid_from_channels_last(dev_ptr(src, "grid_cl"), dev_ptr(out, "grid"), B, C, D, H, W, stream_ptr()),
          "vt_grid_from_channels_last")
    return out


def _cl_storage(grid):
    """Device pointer of the channels-last storage of a [B,C,R,R,R] grid."""
    g = grid_to_channels_last(grid)
    return g, dev_ptr(g.permute(0, 2, 3, 4, 1), "grid")


_wide_ws = {}          # per (device, stream): the sampling workspace of the 64 / 32 pipeline (vt_decode_fwd_wide_f16x3_ws)
WIDE_SLICE = 1 << 19   # points per launch pair of that pipeline


def decode_fwd(grid, blob, pts=None, c_img=None, padding=0.1, lattice=None, want_contact=False, out=None, save=None,
               precision="f32", wide=None, finger_ids=None, finger_feats=None):
    """Fused trilinear gather + conditioned MLP (vt_decode_fwd; ``precision="bf16x3"`` / ``"f16x3"``:
    vt_decode_fwd_bf16x3 / vt_decode_fwd_f16x3 with a blob packed for it; ``precision="wide"`` /
    ``"wide_f16x3"`` with ``wide=(hidden_size, n_blocks, leaky[, nearest])``: vt_decode_fwd_wide[_f16x3], the exact-f32 / split-f16
    kernels of the shapes beyond 32/32).

    grid  [B,C,R,R,R] (any layout; converted to channels-last if needed)
    pts   [B,N,3] or None with lattice=(nx, box, first, count)
    c_img [B,N,C] or None.  Returns logits [B,N] (and contact logits).
    """
    lib = _lib.load()
    B, C, D, H, W = grid.shape
    if not (D == H == W):
        raise VtError("feature grid must be cubic")
    keep, gptr = _cl_storage(grid)
    if pts is not None:
        pts = pts.detach().float()
        if not pts.is_contiguous():
            pts = pts.contiguous()
        if pts.shape[0] != B or pts.shape[-1] != 3:
            raise VtError(f"pts must be [B,N,3] with B={B} (got {tuple(pts.shape)})")
        N = pts.shape[1]
        nx, box, first = 0, 0.0, 0
    else:
        nx, box, first, N = lattice
    if c_img is not None:
        c_img = c_img.detach()
        if not c_img.is_contiguous():
            c_img = c_img.contiguous()
        if tuple(c_img.shape) != (B, N, C):
            raise VtError(f"c_img must be [B,N,C]=({B},{N},{C}) (got {tuple(c_img.shape)})")
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=grid.device)
    out2 = torch.empty((B, N), dtype=torch.float32, device=grid.device) if want_contact else None
    if N == 0:                                   # empty query set: nothing to launch
        return (out, out2) if want_contact else out
    keep_for_graph(blob, keep)
    if precision in ("wide", "wide_f16x3"):
        if save is not None or wide is None:
            raise VtError("decode_fwd: precision 'wide' is inference only and needs wide=(hidden_size, n_blocks, leaky)")
        hidden, nb, leaky = wide[:3]
        flags = (1 if leaky else 0) | (2 if len(wide) > 3 and wide[3] else 0)          # VT_WIDE_LEAKY | VT_WIDE_NEAREST
        name = "vt_decode_fwd_wide" if precision == "wide" else "vt_decode_fwd_wide_f16x3"
        if finger_ids is not None:
            # the tactile feature by finger id (uint8 [B,N], 255 = none) and the [F,C] table: no dense [B,N,C] tensor
            ids, feats = _c(finger_ids), _c(finger_feats.detach().float())
            if c_img is not None or ids.dtype != torch.uint8 or ids.numel() != B * N or feats.dim() != 2 or feats.shape[1] != C:
                raise VtError(f"decode_fwd: finger ids must be uint8 [B,N] with a [F,{C}] feature table (and no c_img)")
            keep_for_graph(ids, feats)
            check(getattr(lib, name + "_ids")(gptr, B, D, C, dev_ptr(pts, "pts"), N, nx, box, first, dev_ptr(ids, "finger_ids", torch.uint8),
                                              dev_ptr(feats, "finger_feats"), int(feats.shape[0]), dev_ptr(blob, "blob"), int(hidden),
                                              int(nb), flags, float(padding), dev_ptr(out, "out"), dev_ptr(out2, "out2"), stream_ptr()),
                  name + "_ids")
            return (out, out2) if want_contact else out
        wsb = lib.vt_decode_wide_f16x3_workspace_bytes(B * N, int(hidden), C, int(nb), 0 if c_img is None else 1) if precision == "wide_f16x3" else 0
        if wsb:
            # 64 / 32 / <= 5: the register-resident pipeline on the grid's samples, which a pre-pass leaves in a workspace.  The call
            # runs scene by scene in slices of WIDE_SLICE points (the workspace is 128 bytes per point: 67 MB per slice instead of 268 MB
            # for a 128^3 lattice and 2.1 GB for 256^3; a slice's samples are still in the Infinity Cache when the pipeline reads them),
            # one workspace per (device, stream) -- two streams decoding at once do not share it -- released when a smaller call comes
            step = min(N, WIDE_SLICE)
            wsb = lib.vt_decode_wide_f16x3_workspace_bytes(step, int(hidden), C, int(nb), 0 if c_img is None else 1)
            key = (grid.device, stream_ptr().value)
            ws = _wide_ws.get(key)
            if ws is None or ws.numel() < wsb or ws.numel() > 4 * wsb:
                ws = _wide_ws[key] = torch.empty(wsb, dtype=torch.uint8, device=grid.device)
            keep_for_graph(ws)
            cl = keep.permute(0, 2, 3, 4, 1)                           # [B, R, R, R, C] contiguous: scene b starts at cl[b]
            for b in range(B):
                for lo in range(0, N, step):
                    n = min(step, N - lo)
                    sl = (slice(b, b + 1), slice(lo, lo + n))
                    check(lib.vt_decode_fwd_wide_f16x3_ws(dev_ptr(cl[b], "grid"), 1, D, C, dev_ptr(pts[sl] if pts is not None else None, "pts"), n,
                                                          nx, box, first + lo if pts is None else 0,
                                                          dev_ptr(c_img[sl] if c_img is not None else None, "c_img"), dev_ptr(blob, "blob"),
                                                          int(hidden), int(nb), flags, float(padding), dev_ptr(out[sl], "out"),
                                                          dev_ptr(out2[sl] if out2 is not None else None, "out2"),
                                                          ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream_ptr()),
                          "vt_decode_fwd_wide_f16x3_ws")
            return (out, out2) if want_contact else out
        check(getattr(lib, name)(gptr, B, D, C, dev_ptr(pts, "pts"), N, nx, box, first, dev_ptr(c_img, "c_img"),
                                 dev_ptr(blob, "blob"), int(hidden), int(nb), flags, float(padding),
                                 dev_ptr(out, "out"), dev_ptr(out2, "out2"), stream_ptr()), name)
    elif precision == "f16f8":
        if pts is not None or want_contact or save is not None:
            raise VtError("decode_fwd: precision 'f16f8' covers lattice slabs only (ops.f16f8_covers); use 'f16x3'")
        check(lib.vt_decode_fwd_f16f8(gptr, B, D, C, N, nx, box, first, dev_ptr(c_img, "c_img"), None, None, 0,
                                      dev_ptr(blob, "blob"), float(padding), dev_ptr(out, "out"), stream_ptr()), "vt_decode_fwd_f16f8")
    elif precision in SPLIT_PRECISIONS:
        if save is not None:
            raise VtError("decode_fwd: the training forward (save) is exact-f32 only")
        name = "vt_decode_fwd_" + precision
        check(getattr(lib, name)(gptr, B, D, C, dev_ptr(pts, "pts"), N, nx, box, first,
                                 dev_ptr(c_img, "c_img"), None, None, 0, dev_ptr(blob, "blob"), float(padding),
                                 dev_ptr(out, "out"), dev_ptr(out2, "out2"), stream_ptr()), name)
    elif precision == "f32":
        check(lib.vt_decode_fwd(gptr, B, D, C, dev_ptr(pts, "pts"), N, nx, box, first,
                                dev_ptr(c_img, "c_img"), dev_ptr(blob, "blob"), float(padding),
                                dev_ptr(out, "out"), dev_ptr(out2, "out2"), dev_ptr(save, "save"), stream_ptr()), "vt_decode_fwd")
    else:
        raise VtError(f"precision must be one of {PRECISIONS} (got {precision!r})")
    return (out, out2) if want_contact else out


_mc_ws_cache = {}


def _mc_workspace(vol):
    lib = _lib.load()
    n0, n1, n2 = vol.shape
    nbytes = lib.vt_mc_workspace_bytes(n0, n1, n2)
    if nbytes == 0:
        raise ValueError("Input array must be at least 2x2x2.")
    key = (vol.device, nbytes)
    ws = _mc_ws_cache.get(key)
    if ws is None:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=vol.device)
        _mc_ws_cache.clear()
        _mc_ws_cache[key] = ws
    return ws, nbytes


def mc_count(vol, level=None):
    """Phase 1 of marching cubes (vt_mc_count): classify + scan into the workspace.  Asynchronous
    and graph-capturable; returns the workspace tensor."""
    if vol.dim() != 3:
        raise VtError("marching_cubes: volume must be [n0,n1,n2]")
    if not vol.is_contiguous():
        raise VtError("marching_cubes: volume must be contiguous")
    ws, nbytes = _mc_workspace(vol)
    n0, n1, n2 = vol.shape
    check(_lib.load().vt_mc_count(dev_ptr(vol, "vol"), n0, n1, n2, 0.0 if level is None else float(level), int(level is None),
                                  ctypes.c_void_p(ws.data_ptr()), nbytes, stream_ptr()), "vt_mc_count")
    return ws


def mc_count_notify(vol, level=None):
    """``mc_count`` whose scan kernel also writes the counts into a page-locked host slot (vt_mc_count_notify): returns
    (workspace, token); hand the token to ``mc_emit`` -- the count read then needs no copy command between the count and the emit
    kernels.  Not for graph capture: the slot's sequence number is baked into the scan kernel's arguments, so a replay would write
    a number nobody waits for; captured scenes use ``mc_count``."""
    if vol.dim() != 3 or not vol.is_contiguous():
        raise VtError("marching_cubes: volume must be a contiguous [n0,n1,n2] tensor")
    ws, nbytes = _mc_workspace(vol)
    n0, n1, n2 = vol.shape
    tok = ctypes.c_int()
    check(_lib.load().vt_mc_count_notify(dev_ptr(vol, "vol"), n0, n1, n2, 0.0 if level is None else float(level), int(level is None),
                                         ctypes.c_void_p(ws.data_ptr()), nbytes, stream_ptr(), ctypes.byref(tok)), "vt_mc_count_notify")
    return ws, tok.value


def mc_echo_slot():
    """A page-locked slot for the counts of a CAPTURED scene (vt_mc_echo_slot): make it outside the capture, hand it to
    ``mc_count_echo`` inside, ``mc_echo_arm`` it before every replay and pass it to ``mc_emit(echo=...)``."""
    tok = ctypes.c_int()
    check(_lib.load().vt_mc_echo_slot(ctypes.byref(tok)), "vt_mc_echo_slot")
    return tok.value


def mc_echo_release(slot):
    """Hand a slot back (vt_mc_echo_release) once the graph that echoes into it is gone; the next ``mc_echo_slot`` reuses it."""
    check(_lib.load().vt_mc_echo_release(int(slot)), "vt_mc_echo_release")


def mc_count_echo(vol, level, slot):
    """``mc_count`` for graph capture whose scan kernel echoes the slot's current number behind the counts into the slot's page-locked
    header (vt_mc_count_echo): the replayed scene needs no copy command between the count and the emit kernels."""
    if vol.dim() != 3 or not vol.is_contiguous():
        raise VtError("marching_cubes: volume must be a contiguous [n0,n1,n2] tensor")
    ws, nbytes = _mc_workspace(vol)
    n0, n1, n2 = vol.shape
    check(_lib.load().vt_mc_count_echo(dev_ptr(vol, "vol"), n0, n1, n2, 0.0 if level is None else float(level), int(level is None),
                                       ctypes.c_void_p(ws.data_ptr()), nbytes, stream_ptr(), int(slot)), "vt_mc_count_echo")
    return ws


def mc_echo_arm(slot):
    """A fresh number into the slot (vt_mc_echo_arm): before every replay of the graph that holds ``mc_count_echo``."""
    check(_lib.load().vt_mc_echo_arm(int(slot)), "vt_mc_echo_arm")


_mc_guess = {}          # volume shape -> (vertex, face) capacity that covered the last extraction there


def mc_emit(vol, ws, rescale=None, capacity=None, token=None, echo=None):
    """Phase 2 (vt_mc_emit + vt_mc_read_counts).  With ``capacity=(V,F)`` nothing is read back
    (no stream sync; the counts stay in the workspace).  Otherwise the outputs are sized by the
    counts, which costs one host read: after the first extraction of a shape the emit kernels are
    launched SPECULATIVELY into buffers 25 % larger than the previous result before that read, so
    the read is the only synchronisation of the call; if the surface outgrew the guess the emit is
    repeated at the exact size (the kernels never write past their capacity)."""
    lib = _lib.load()
    if capacity is not None and (token is not None or echo is not None):
        raise VtError("mc_emit: a token (mc_count_notify) is handed back by reading the counts; with capacity= nothing is read -- "
                      "use mc_count for fixed-capacity extraction")
    n0, n1, n2 = vol.shape
    st = stream_ptr()
    wp = ctypes.c_void_p(ws.data_ptr())
    shift, scale = rescale if rescale is not None else (0.0, 1.0)

    def emit(cap_v, cap_f):
        verts = torch.empty((cap_v, 3), dtype=torch.float32, device=vol.device)
        faces = torch.empty((cap_f, 3), dtype=torch.int32, device=vol.device)
        check(lib.vt_mc_emit(dev_ptr(vol, "vol"), n0, n1, n2, wp, dev_ptr(verts, "verts"), cap_v,
                             dev_ptr(faces, "faces", torch.int32), cap_f, int(rescale is not None), shift, scale, st),
              "vt_mc_emit")
        return verts, faces

    if capacity is not None:
        verts, faces = emit(*capacity)
        return verts, faces, ws          # counts stay on the device: ws[8:16] = (nverts, nfaces) int32
    key = (vol.device.index, n0, n1, n2)
    guess = _mc_guess.get(key)
    # the copy of the counts is queued FIRST (it needs the classify / scan launches only) and waited for by its own event, so
    # the speculative emit kernels run under that wait instead of in front of the copy
    tok = ctypes.c_int(-1 if token is None else token)
    if token is None and echo is None:
        check(lib.vt_mc_read_counts_begin(wp, st, ctypes.byref(tok)), "vt_mc_read_counts_begin")
    nv, nf, lvl = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    try:
        spec = emit(*guess) if guess is not None else None
    finally:                             # the token is handed back whatever the emit did (sixteen exist)
        if echo is not None:             # (a captured scene's slot, armed before the replay: the scan kernel echoes its number)
            rc = lib.vt_mc_echo_wait(int(echo), st, ctypes.byref(nv), ctypes.byref(nf), ctypes.byref(lvl))
        else:
            rc = lib.vt_mc_read_counts_end(tok.value, ctypes.byref(nv), ctypes.byref(nf), ctypes.byref(lvl))
    check(rc, "vt_mc_read_counts_end")
    if nv.value == 0:
        raise RuntimeError("No surface found at the given iso value.")
    _mc_guess[key] = (nv.value + nv.value // 4 + 1024, nf.value + nf.value // 4 + 1024)
    if spec is not None and nv.value <= guess[0] and nf.value <= guess[1]:
        return spec[0][:nv.value], spec[1][:nf.value], lvl.value
    verts, faces = emit(nv.value, nf.value)
    return verts, faces, lvl.value


def marching_cubes(vol, level=None, rescale=None, capacity=None):
    """Lewiner marching cubes of a device volume [n0,n1,n2] (vt_mc_count/emit).

    Returns (verts f32 [V,3] in array-axis order, faces i32 [F,3], level) as device
    tensors, numbered exactly as skimage.measure.marching_cubes(vol,
    gradient_direction='ascent') numbers them.  ``rescale=(shift, scale)`` fuses the
    reference's `v -= shift; v *= scale` (generation.py:271-272).  ``capacity=(V,F)``
    skips the host read of the counts (no stream sync); otherwise the counts are read
    back once to size the outputs.  Raises RuntimeError('No surface found ...') like
    skimage when the level misses the data.
    """
    if vol.dim() != 3:
        raise VtError("marching_cubes: volume must be [n0,n1,n2]")
    vol = vol.detach()
    if not vol.is_contiguous():
        vol = vol.contiguous()
    if capacity is not None or torch.cuda.is_current_stream_capturing():
        return mc_emit(vol, mc_count(vol, level), rescale, capacity)
    ws, tok = mc_count_notify(vol, level)            # the counts arrive in a page-locked slot: no copy between count and emit
    return mc_emit(vol, ws, rescale, None, token=tok)


# --------------------------------------------------------------------------------------
# PointNet local-pool voxeliser (vt_voxel_*)
# --------------------------------------------------------------------------------------
I32 = torch.int32


class VoxelIndex:
    """Per-forward voxel bookkeeping of a point cloud [B,T,3] (vt_voxel_build).  ``clear``: a contiguous float tensor the same
    launch zero-fills with the workgroups the sort leaves idle (vt_voxel_build_clear: the mean grid, without a fill launch)."""

    def __init__(self, pts, reso, padding=0.1, clear=None, want_tile_flags=False):
        pts = pts.detach().float()
        if not pts.is_contiguous():
            pts = pts.contiguous()
        B, T, _ = pts.shape
        self.B, self.T, self.R = B, T, reso
        dev = pts.device
        self.idx = torch.empty((B, T), dtype=I32, device=dev)
        self.order = torch.empty((B, T), dtype=I32, device=dev)
        self.seg_lo = torch.empty((B, T), dtype=I32, device=dev)
        self.seg_hi = torch.empty((B, T), dtype=I32, device=dev)
        # ``want_tile_flags``: uint8 [B, (reso/8)^3], 1 where no point lies in the 10^3 halo of that 8^3 block (voxel_tile_flags),
        # marked by the same launch (vt_voxel_build_clear_flags); None where the resolution is not covered
        self.tile_flags = None
        if want_tile_flags and reso % 8 == 0 and 8 <= reso <= 128:
            self.tile_flags = torch.empty((B, (reso // 8) ** 3), dtype=torch.uint8, device=dev)
            check(_lib.load().vt_voxel_build_clear_flags(dev_ptr(pts, "pts"), B, T, reso, float(padding),
                                                         dev_ptr(self.idx, "idx", I32), dev_ptr(self.order, "order", I32),
                                                         dev_ptr(self.seg_lo, "seg_lo", I32), dev_ptr(self.seg_hi, "seg_hi", I32),
                                                         dev_ptr(clear, "clear") if clear is not None else None,
                                                         clear.numel() * clear.element_size() if clear is not None else 0,
                                                         dev_ptr(self.tile_flags, "tile_flags", torch.uint8), stream_ptr()),
                  "vt_voxel_build_clear_flags")
            return
        if clear is not None:
            check(_lib.load().vt_voxel_build_clear(dev_ptr(pts, "pts"), B, T, reso, float(padding),
                                                   dev_ptr(self.idx, "idx", I32), dev_ptr(self.order, "order", I32),
                                                   dev_ptr(self.seg_lo, "seg_lo", I32), dev_ptr(self.seg_hi, "seg_hi", I32),
                                                   dev_ptr(clear, "clear"), clear.numel() * clear.element_size(), stream_ptr()),
                  "vt_voxel_build_clear")
            return
        check(_lib.load().vt_voxel_build(dev_ptr(pts, "pts"), B, T, reso, float(padding),
                                         dev_ptr(self.idx, "idx", I32), dev_ptr(self.order, "order", I32),
                                         dev_ptr(self.seg_lo, "seg_lo", I32), dev_ptr(self.seg_hi, "seg_hi", I32),
                                         stream_ptr()), "vt_voxel_build")


def voxel_tile_flags(vi):
    """uint8 [B, (R/8)^3]: 1 where no point of the scene lies in the 10^3 halo of that 8^3 voxel block (vt_voxel_tile_flags) -- the
    mean grid is zero over everything a 3x3x3 conv of the block reads, so the UNet3D's first layer can skip the block's taps
    (unet3d_fwd(tile_flags=...)).  None where the resolution is not covered (not a multiple of 8, or above 128)."""
    if vi.R % 8 or vi.R < 8 or vi.R > 128:
        return None
    flags = torch.empty((vi.B, (vi.R // 8) ** 3), dtype=torch.uint8, device=vi.idx.device)
    check(_lib.load().vt_voxel_tile_flags(dev_ptr(vi.idx, "idx", I32), vi.B, vi.T, vi.R, dev_ptr(flags, "flags", torch.uint8), stream_ptr()),
          "vt_voxel_tile_flags")
    return flags


def _c(t):
    t = t.detach()
    return t if t.is_contiguous() else t.contiguous()


def voxel_pool_max_fwd(feat, vi, want_argmax=True):
    feat = _c(feat)
    B, T, C = feat.shape
    out = torch.empty_like(feat)
    arg = torch.empty((B, T, C), dtype=I32, device=feat.device) if want_argmax else None
    check(_lib.load().vt_voxel_pool_max_fwd(dev_ptr(feat, "feat"), dev_ptr(vi.order, "order", I32),
                                            dev_ptr(vi.seg_lo, "seg_lo", I32), dev_ptr(vi.seg_hi, "seg_hi", I32),
                                            B, T, C, dev_ptr(out, "out"), dev_ptr(arg, "argmax", I32), stream_ptr()),
          "vt_voxel_pool_max_fwd")
    return out, arg


def _ptr_array(tensors, name):
    """A host array of device pointers (ctypes c_void_p * K) for the K int32 tensors, or None entries."""
    arr = (ctypes.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = dev_ptr(t, name, I32).value if t is not None else None
    return arr


def voxel_pool_max_sum_fwd(feat, vis, want_argmax=True):
    """Sum over the index sets ``vis`` of the per-cell channel max, gathered back to the points (vt_voxel_pool_max_sum_fwd): one launch
    for the hand encoder's three planes.  Returns (out [B,T,C], list of arg-max tensors or None)."""
    feat = _c(feat)
    B, T, C = feat.shape
    K = len(vis)
    out = torch.empty_like(feat)
    args = [torch.empty((B, T, C), dtype=I32, device=feat.device) for _ in range(K)] if want_argmax else None
    check(_lib.load().vt_voxel_pool_max_sum_fwd(dev_ptr(feat, "feat"), K, _ptr_array([v.order for v in vis], "order"),
                                                _ptr_array([v.seg_lo for v in vis], "seg_lo"), _ptr_array([v.seg_hi for v in vis], "seg_hi"),
                                                B, T, C, dev_ptr(out, "out"), _ptr_array(args, "argmax") if args else None, stream_ptr()),
          "vt_voxel_pool_max_sum_fwd")
    return out, args


def voxel_pool_max_sum_bwd(grad_out, args, vis):
    grad_out = _c(grad_out)
    B, T, C = grad_out.shape
    g = torch.empty_like(grad_out)
    check(_lib.load().vt_voxel_pool_max_sum_bwd(dev_ptr(grad_out, "grad_out"), len(vis), _ptr_array(args, "argmax"),
                                                _ptr_array([v.order for v in vis], "order"), _ptr_array([v.seg_lo for v in vis], "seg_lo"),
                                                _ptr_array([v.seg_hi for v in vis], "seg_hi"), B, T, C, dev_ptr(g, "grad_feat"), stream_ptr()),
          "vt_voxel_pool_max_sum_bwd")
    return g


def voxel_pool_mean(feat, vi):
    """pool_local with scatter_type='mean' (vt_voxel_pool_mean): every point gets the mean of the features of its cell; its
    backward is the same call on the gradient."""
    feat = _c(feat.float())
    B, T, C = feat.shape
    out = torch.empty_like(feat)
    check(_lib.load().vt_voxel_pool_mean(dev_ptr(feat, "feat"), dev_ptr(vi.order, "order", I32), dev_ptr(vi.seg_lo, "seg_lo", I32),
                                         dev_ptr(vi.seg_hi, "seg_hi", I32), B, T, C, dev_ptr(out, "out"), stream_ptr()), "vt_voxel_pool_mean")
    return out


def voxel_pool_max_bwd(grad_out, argmax, vi):
    grad_out = _c(grad_out)
    B, T, C = grad_out.shape
    g = torch.empty_like(grad_out)
    check(_lib.load().vt_voxel_pool_max_bwd(dev_ptr(grad_out, "grad_out"), dev_ptr(argmax, "argmax", I32),
                                            dev_ptr(vi.order, "order", I32), dev_ptr(vi.seg_lo, "seg_lo", I32),
                                            dev_ptr(vi.seg_hi, "seg_hi", I32), B, T, C, dev_ptr(g, "grad_feat"), stream_ptr()),
          "vt_voxel_pool_max_bwd")
    return g


def voxel_scatter_mean_fwd(feat, vi):
    feat = _c(feat)
    B, T, C = feat.shape
    R = vi.R
    grid = torch.empty((B, C, R, R, R), dtype=torch.float32, device=feat.device)
    check(_lib.load().vt_voxel_scatter_mean_fwd(dev_ptr(feat, "feat"), dev_ptr(vi.idx, "idx", I32),
                                                dev_ptr(vi.order, "order", I32), dev_ptr(vi.seg_lo, "seg_lo", I32),
                                                dev_ptr(vi.seg_hi, "seg_hi", I32), B, T, C, R, dev_ptr(grid, "grid"),
                                                stream_ptr()), "vt_voxel_scatter_mean_fwd")
    return grid


def voxel_scatter_mean_bwd(grad_grid, vi, C):
    grad_grid = _c(grad_grid)
    B, T = vi.B, vi.T
    g = torch.empty((B, T, C), dtype=torch.float32, device=grad_grid.device)
    check(_lib.load().vt_voxel_scatter_mean_bwd(dev_ptr(grad_grid, "grad_grid"), dev_ptr(vi.idx, "idx", I32),
                                                dev_ptr(vi.seg_lo, "seg_lo", I32), dev_ptr(vi.seg_hi, "seg_hi", I32),
                                                B, T, C, vi.R, dev_ptr(g, "grad_feat"), stream_ptr()),
          "vt_voxel_scatter_mean_bwd")
    return g


# --------------------------------------------------------------------------------------
# generalized winding number (vt_winding_number)
# --------------------------------------------------------------------------------------
def winding_number(verts, faces, pts):
    """w(q) of every query point pts [..., 3] against the mesh (verts [V,3] f32, faces [F,3] int): 1 inside a closed,
    outward-oriented mesh, 0 outside (the exact sum igl.fast_winding_number_for_meshes approximates)."""
    verts, pts = _c(verts.float()), _c(pts.float())
    faces = _c(faces.to(I32))
    out = torch.empty(pts.shape[:-1], dtype=torch.float32, device=pts.device)
    check(_lib.load().vt_winding_number(dev_ptr(verts, "verts"), verts.shape[0], dev_ptr(faces, "faces", I32), faces.shape[0],
                                        dev_ptr(pts, "pts"), pts.numel() // 3, dev_ptr(out, "out"), stream_ptr()), "vt_winding_number")
    return out


def contact_scan(depth, origin, touch, threshold=1e-4):
    """(index [n_images, n_pixels] i32, count [n_images] i32): per depth image the pixels whose reading departs from the sensor's
    flat one by more than ``threshold``, ascending (np.where's order); images whose ``touch`` byte is 0 count 0 (vt_contact_scan).
    depth [n_images, n_pixels] f32, origin [n_pixels] f64, touch [n_images] u8 or None."""
    depth = _c(depth)
    n_img, n_pix = depth.shape
    index = torch.empty((n_img, n_pix), dtype=I32, device=depth.device)
    count = torch.empty((n_img,), dtype=I32, device=depth.device)
    check(_lib.load().vt_contact_scan(dev_ptr(depth, "depth"), dev_ptr(origin, "depth_origin", torch.float64),
                                      dev_ptr(touch, "touch_success", torch.uint8) if touch is not None else None, n_img, n_pix,
                                      float(threshold), dev_ptr(index, "index", I32), dev_ptr(count, "count", I32), stream_ptr()), "vt_contact_scan")
    return index, count


def contact_points(depth, index, sel, kept, row0, pose, width, height, fov, max_points, p_sample, finger=None):
    """Writes the contact rows of ``p_sample`` [B, S, 3] in place (vt_contact_points): per image ``kept`` pixels (``index[sel]``),
    unprojected, posed and normalised in float64 (``pose`` [n_images, 16] f64: inverse pose 3 x 3, translation, cloud centroid, scale)."""
    depth = _c(depth)
    n_img, n_pix = depth.shape
    check(_lib.load().vt_contact_points(dev_ptr(depth, "depth"), dev_ptr(index, "index", I32), dev_ptr(sel, "sel", I32) if sel is not None else None,
                                        dev_ptr(kept, "kept", I32), dev_ptr(row0, "row0", I32), dev_ptr(pose, "pose", torch.float64),
                                        n_img, n_pix, int(width), int(height), float(fov), int(max_points), p_sample.shape[1],
                                        dev_ptr(p_sample, "p_sample"), dev_ptr(finger, "finger", torch.int64) if finger is not None else None,
                                        stream_ptr()), "vt_contact_points")
    return p_sample


def winding_number_scenes(meshes, pts):
    """w(q) for a batch of scenes in one launch (vt_winding_number_scenes): ``meshes`` = [(verts [V,3] f32, faces [F,3] i32)] device
    tensors per scene, pts [B, N, 3] -> [B, N].  The 24-byte records go up in one small copy."""
    import struct
    pts = _c(pts.float())
    B, N = pts.shape[:2]
    if len(meshes) != B:
        raise VtError(f"winding_number_scenes: {len(meshes)} meshes for {B} scenes")
    rec = bytearray()
    for v, f in meshes:
        if v.dtype != torch.float32 or f.dtype != I32 or not v.is_contiguous() or not f.is_contiguous() or not v.is_cuda or not f.is_cuda:
            raise VtError("winding_number_scenes: meshes must be contiguous device tensors (verts f32 [V,3], faces i32 [F,3])")
        rec += struct.pack("<QQii", v.data_ptr(), f.data_ptr(), v.shape[0], f.shape[0])
    table = torch.frombuffer(rec, dtype=torch.uint8).to(pts.device, non_blocking=True)
    out = torch.empty((B, N), dtype=torch.float32, device=pts.device)
    check(_lib.load().vt_winding_number_scenes(ctypes.c_void_p(table.data_ptr()), B, dev_ptr(pts, "pts"), N, dev_ptr(out, "out"), stream_ptr()),
          "vt_winding_number_scenes")
    return out


# --------------------------------------------------------------------------------------
# PointNet per-point MLP, inference (vt_linear_rows, vt_resblock_fc)
# --------------------------------------------------------------------------------------
def linear_rows(x, weight, bias=None):
    """nn.Linear over the rows of x [..., Cin] -> [..., Cout]."""
    x = _c(x)
    Cout, Cin = weight.shape
    out = torch.empty(x.shape[:-1] + (Cout,), dtype=torch.float32, device=x.device)
    check(_lib.load().vt_linear_rows(dev_ptr(x, "x"), dev_ptr(_c(weight), "weight"),
                                     dev_ptr(_c(bias) if bias is not None else None, "bias"),
                                     x.numel() // Cin, Cin, Cout, dev_ptr(out, "out"), stream_ptr()), "vt_linear_rows")
    return out


def resblock_fc(x1, x2, fc_0, fc_1, shortcut):
    """ResnetBlockFC (layers.py:8-50) on the rows of cat([x1, x2], -1) (x2 may be None); the arguments after x2 are
    the block's nn.Linear modules (shortcut None = identity)."""
    x1 = _c(x1)
    x2 = _c(x2) if x2 is not None else None
    C1, C2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
    H, O = fc_0.weight.shape[0], fc_1.weight.shape[0]
    out = torch.empty(x1.shape[:-1] + (O,), dtype=torch.float32, device=x1.device)
    check(_lib.load().vt_resblock_fc(dev_ptr(x1, "x1"), C1, dev_ptr(x2, "x2"), C2, x1.numel() // C1,
                                     dev_ptr(_c(fc_0.weight), "fc_0.weight"), dev_ptr(_c(fc_0.bias), "fc_0.bias"),
                                     dev_ptr(_c(fc_1.weight), "fc_1.weight"), dev_ptr(_c(fc_1.bias), "fc_1.bias"),
                                     dev_ptr(_c(shortcut.weight) if shortcut is not None else None, "shortcut.weight"),
                                     H, O, dev_ptr(out, "out"), stream_ptr()), "vt_resblock_fc")
    return out


def resblock_fc_bwd(x1, x2, fc_0_w, fc_0_b, fc_1_w, shortcut_w, dout, want_dx2=True):
    """vt_resblock_fc_bwd: (dx1, dx2 or None, act = relu(h) [.., H], dh [.., H])."""
    x1, dout = _c(x1), _c(dout)
    x2 = _c(x2) if x2 is not None else None
    C1, C2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
    H, O = fc_0_w.shape[0], fc_1_w.shape[0]
    N = x1.numel() // C1
    dev = x1.device
    dx1 = torch.empty_like(x1)
    dx2 = torch.empty_like(x2) if (x2 is not None and want_dx2) else None
    act = torch.empty(x1.shape[:-1] + (H,), dtype=torch.float32, device=dev)
    dh = torch.empty_like(act)
    check(_lib.load().vt_resblock_fc_bwd(dev_ptr(x1, "x1"), C1, dev_ptr(x2, "x2"), C2, N, dev_ptr(_c(fc_0_w), "fc_0.weight"),
                                         dev_ptr(_c(fc_0_b), "fc_0.bias"), dev_ptr(_c(fc_1_w), "fc_1.weight"),
                                         dev_ptr(_c(shortcut_w) if shortcut_w is not None else None, "shortcut.weight"), H, O,
                                         dev_ptr(dout, "dout"), dev_ptr(dx1, "dx1"), dev_ptr(dx2, "dx2"), dev_ptr(act, "act"),
                                         dev_ptr(dh, "dh"), stream_ptr()), "vt_resblock_fc_bwd")
    return dx1, dx2, act, dh


def rows_wgrad(g, x1, x2=None, relu_x=False, want_bias=True):
    """vt_rows_wgrad: dW [M, K] = g^T [x1 | x2] over the rows (x relu'd when ``relu_x``), db [M] = column sums of g."""
    g, x1 = _c(g), _c(x1)
    x2 = _c(x2) if x2 is not None else None
    M, C1, C2 = g.shape[-1], x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
    N = g.numel() // M
    lib = _lib.load()
    dev = g.device
    wsb = lib.vt_rows_wgrad_workspace_bytes(N, M, C1 + C2)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    dW = torch.empty((M, C1 + C2), dtype=torch.float32, device=dev)
    db = torch.empty((M,), dtype=torch.float32, device=dev) if want_bias else None
    check(lib.vt_rows_wgrad(dev_ptr(g, "g"), M, dev_ptr(x1, "x1"), C1, dev_ptr(x2, "x2"), C2, int(relu_x), N,
                            ctypes.c_void_p(ws.data_ptr()), wsb, dev_ptr(dW, "dW"), dev_ptr(db, "db"), stream_ptr()), "vt_rows_wgrad")
    return dW, db


def resblock_wgrad(x1, x2, act, dh, dout, has_shortcut):
    """The three weight gradients of a ResnetBlockFC in one pair of launches (vt_resblock_wgrad): returns (dw0, db0, dw1, db1, dws or
    None), or None where the block is too wide for it (use rows_wgrad per product)."""
    x1, act, dh, dout = _c(x1), _c(act), _c(dh), _c(dout)
    x2 = _c(x2) if x2 is not None else None
    C1, C2 = x1.shape[-1], (x2.shape[-1] if x2 is not None else 0)
    H, O = act.shape[-1], dout.shape[-1]
    N = x1.numel() // C1
    lib = _lib.load()
    wsb = lib.vt_resblock_wgrad_workspace_bytes(N, C1 + C2, H, O, 1 if has_shortcut else 0)
    if not wsb:
        return None
    dev = x1.device
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    dw0 = torch.empty((H, C1 + C2), dtype=torch.float32, device=dev)
    db0 = torch.empty((H,), dtype=torch.float32, device=dev)
    dw1 = torch.empty((O, H), dtype=torch.float32, device=dev)
    db1 = torch.empty((O,), dtype=torch.float32, device=dev)
    dws = torch.empty((O, C1 + C2), dtype=torch.float32, device=dev) if has_shortcut else None
    check(lib.vt_resblock_wgrad(dev_ptr(x1, "x1"), C1, dev_ptr(x2, "x2"), C2, N, dev_ptr(act, "act"), dev_ptr(dh, "dh"), dev_ptr(dout, "dout"),
                                H, O, ctypes.c_void_p(ws.data_ptr()), wsb, dev_ptr(dw0, "dw0"), dev_ptr(db0, "db0"), dev_ptr(dw1, "dw1"),
                                dev_ptr(db1, "db1"), dev_ptr(dws, "dws"), stream_ptr()), "vt_resblock_wgrad")
    return dw0, db0, dw1, db1, dws


# --------------------------------------------------------------------------------------
# hand branch: plane bookkeeping (vt_plane_*) and the MANO layer (vt_mano_*)
# --------------------------------------------------------------------------------------
PLANES = {"xz": 0, "xy": 1, "yz": 2}


class PlaneIndex(VoxelIndex):
    """VoxelIndex of one canonical plane (vt_plane_build): same fields, R^2 cells, so the
    voxel_pool_max_* wrappers take it unchanged."""

    def __init__(self, pts, reso, padding=0.1, plane="xz"):
        if plane not in PLANES:
            raise _lib.VtError(f"PlaneIndex: unknown plane {plane!r} (one of {sorted(PLANES)})")
        pts = pts.detach().float()
        if not pts.is_contiguous():
            pts = pts.contiguous()
        B, T, _ = pts.shape
        self.B, self.T, self.R, self.plane = B, T, reso, plane
        dev = pts.device
        self.idx, self.order, self.seg_lo, self.seg_hi = (torch.empty((B, T), dtype=I32, device=dev) for _ in range(4))
        check(_lib.load().vt_plane_build(dev_ptr(pts, "pts"), B, T, reso, float(padding), PLANES[plane],
                                         dev_ptr(self.idx, "idx", I32), dev_ptr(self.order, "order", I32),
                                         dev_ptr(self.seg_lo, "seg_lo", I32), dev_ptr(self.seg_hi, "seg_hi", I32),
                                         stream_ptr()), "vt_plane_build")


def plane_indices(pts, reso, padding=0.1, planes=("xz", "xy", "yz")):
    """[PlaneIndex(pts, reso, padding, k) for k in planes] from ONE launch (vt_plane_build_multi): the planes' sorts run side by side."""
    for k in planes:
        if k not in PLANES:
            raise _lib.VtError(f"PlaneIndex: unknown plane {k!r} (one of {sorted(PLANES)})")
    if not 1 <= len(planes) <= 3:
        return [PlaneIndex(pts, reso, padding, k) for k in planes]
    pts = pts.detach().float()
    if not pts.is_contiguous():
        pts = pts.contiguous()
    B, T, _ = pts.shape
    n = len(planes)
    buf = torch.empty((4, n, B, T), dtype=I32, device=pts.device)
    ids = (ctypes.c_int * n)(*[PLANES[k] for k in planes])
    check(_lib.load().vt_plane_build_multi(dev_ptr(pts, "pts"), B, T, reso, float(padding), n, ids, dev_ptr(buf[0], "idx", I32),
                                           dev_ptr(buf[1], "order", I32), dev_ptr(buf[2], "seg_lo", I32), dev_ptr(buf[3], "seg_hi", I32),
                                           stream_ptr()), "vt_plane_build_multi")
    out = []
    for i, k in enumerate(planes):
        pi = PlaneIndex.__new__(PlaneIndex)
        pi.B, pi.T, pi.R, pi.plane, pi.tile_flags = B, T, reso, k, None
        pi.idx, pi.order, pi.seg_lo, pi.seg_hi = buf[0, i], buf[1, i], buf[2, i], buf[3, i]
        pi.group = (buf, i, n)                                      # the planes' arrays side by side: the *_multi entries take them whole
        out.append(pi)
    return out


def plane_group(pis):
    """The shared [4, n, B, T] index buffer of ``pis`` if they are exactly the planes of one plane_indices call, in order; else None."""
    g = getattr(pis[0], "group", None)
    if g is None or g[2] != len(pis) or any(getattr(p, "group", (None,))[0] is not g[0] or p.group[1] != i for i, p in enumerate(pis)):
        return None
    return g[0]


def plane_scatter_mean_multi_fwd(feat, pis):
    """generate_plane_features for the planes of one plane_indices call in one launch (vt_plane_scatter_mean_multi_fwd): [n * B, C, R, R],
    the planes one after the other (= torch.cat of the per-plane tensors)."""
    buf = plane_group(pis)
    feat = _c(feat)
    B, T, C = feat.shape
    n, R = len(pis), pis[0].R
    planes = torch.empty((n * B, C, R, R), dtype=torch.float32, device=feat.device)
    check(_lib.load().vt_plane_scatter_mean_multi_fwd(dev_ptr(feat, "feat"), n, dev_ptr(buf[0], "idx", I32), dev_ptr(buf[1], "order", I32),
                                                      dev_ptr(buf[2], "seg_lo", I32), dev_ptr(buf[3], "seg_hi", I32), B, T, C, R,
                                                      dev_ptr(planes, "planes"), stream_ptr()), "vt_plane_scatter_mean_multi_fwd")
    return planes


def plane_scatter_mean_multi_bwd(grad_planes, pis, C):
    buf = plane_group(pis)
    grad_planes = _c(grad_planes)
    B, T, n, R = pis[0].B, pis[0].T, len(pis), pis[0].R
    g = torch.empty((B, T, C), dtype=torch.float32, device=grad_planes.device)
    check(_lib.load().vt_plane_scatter_mean_multi_bwd(dev_ptr(grad_planes, "grad_planes"), n, dev_ptr(buf[0], "idx", I32),
                                                      dev_ptr(buf[2], "seg_lo", I32), dev_ptr(buf[3], "seg_hi", I32), B, T, C, R,
                                                      dev_ptr(g, "grad_feat"), stream_ptr()), "vt_plane_scatter_mean_multi_bwd")
    return g


def plane_scatter_mean_fwd(feat, pi):
    feat = _c(feat)
    B, T, C = feat.shape
    plane = torch.empty((B, C, pi.R, pi.R), dtype=torch.float32, device=feat.device)
    check(_lib.load().vt_plane_scatter_mean_fwd(dev_ptr(feat, "feat"), dev_ptr(pi.idx, "idx", I32),
                                                dev_ptr(pi.order, "order", I32), dev_ptr(pi.seg_lo, "seg_lo", I32),
                                                dev_ptr(pi.seg_hi, "seg_hi", I32), B, T, C, pi.R, dev_ptr(plane, "plane"),
                                                stream_ptr()), "vt_plane_scatter_mean_fwd")
    return plane


def plane_scatter_mean_bwd(grad_plane, pi, C):
    grad_plane = _c(grad_plane)
    g = torch.empty((pi.B, pi.T, C), dtype=torch.float32, device=grad_plane.device)
    check(_lib.load().vt_plane_scatter_mean_bwd(dev_ptr(grad_plane, "grad_plane"), dev_ptr(pi.idx, "idx", I32),
                                                dev_ptr(pi.seg_lo, "seg_lo", I32), dev_ptr(pi.seg_hi, "seg_hi", I32),
                                                pi.B, pi.T, C, pi.R, dev_ptr(g, "grad_feat"), stream_ptr()),
          "vt_plane_scatter_mean_bwd")
    return g


MANO_BLOB_FLOATS = 330240


def mano_pack(v_template, shapedirs, betas, posedirs, j_regressor, weights, hands_mean, left=False):
    """Model arrays (f32, on the device) -> the blob vt_mano_fwd reads (vt_mano_pack_side; ``left``: a MANO_LEFT model)."""
    dev = v_template.device
    want = {"v_template": (v_template, (778, 3)), "posedirs": (posedirs, (778, 3, 135)),
            "j_regressor": (j_regressor, (16, 778)), "weights": (weights, (778, 16)), "hands_mean": (hands_mean, (45,))}
    if betas is not None:
        want["shapedirs"], want["betas"] = (shapedirs, (778, 3, 10)), (betas, (10,))
    arrs = {}
    for name, (t, shape) in want.items():
        if tuple(t.shape) != shape:
            raise _lib.VtError(f"mano_pack: {name} has shape {tuple(t.shape)}, expected {shape}")
        arrs[name] = _c(t.float())
    blob = torch.empty(MANO_BLOB_FLOATS, dtype=torch.float32, device=dev)
    check(_lib.load().vt_mano_pack_side(dev_ptr(arrs["v_template"], "v_template"), dev_ptr(arrs.get("shapedirs"), "shapedirs"),
                                        dev_ptr(arrs.get("betas"), "betas"), dev_ptr(arrs["posedirs"], "posedirs"),
                                        dev_ptr(arrs["j_regressor"], "j_regressor"), dev_ptr(arrs["weights"], "weights"),
                                        dev_ptr(arrs["hands_mean"], "hands_mean"), int(bool(left)), dev_ptr(blob, "blob"), stream_ptr()),
          "vt_mano_pack_side")
    return blob


def mano_fwd(pose, blob, center_idx=9):
    """pose [B,48] -> (verts [B,778,3], joints [B,21,3]) (vt_mano_fwd)."""
    pose = _c(pose.float())
    if pose.dim() != 2 or pose.shape[1] != 48:
        raise _lib.VtError(f"mano_fwd: pose must be [B,48] (root axis-angle + 45 joint angles), got {tuple(pose.shape)}")
    B = pose.shape[0]
    verts = torch.empty((B, 778, 3), dtype=torch.float32, device=pose.device)
    joints = torch.empty((B, 21, 3), dtype=torch.float32, device=pose.device)
    check(_lib.load().vt_mano_fwd(dev_ptr(pose, "pose"), B, dev_ptr(blob, "blob"),
                                  -1 if center_idx is None else int(center_idx),
                                  dev_ptr(verts, "verts"), dev_ptr(joints, "joints"), stream_ptr()), "vt_mano_fwd")
    return verts, joints


def pointnet_mlp_weights(fc_pos, blocks, fc_c):
    """The weight pointers vt_pointnet_mlp_fused takes, gathered once: (ctypes pointer array of the 25 block tensors, those tensors,
    [fc_pos.weight, fc_pos.bias, fc_c.weight, fc_c.bias]); valid while the parameters keep their storage."""
    ws = []
    for blk in blocks:
        ws += [_c(blk.fc_0.weight), _c(blk.fc_0.bias), _c(blk.fc_1.weight), _c(blk.fc_1.bias), _c(blk.shortcut.weight)]
    ptrs = (ctypes.c_void_p * len(ws))(*[t.data_ptr() for t in ws])
    return ptrs, ws, [_c(fc_pos.weight), _c(fc_pos.bias), _c(fc_c.weight), _c(fc_c.bias)]


def pointnet_mlp_fused(p, vi, fc_pos, blocks, fc_c, want_grid=False, weights=None, zeroed_grid=None):
    """fc_pos -> block 0 -> 4 x (pool over the point's cell, concat, block) -> fc_c for one voxel index in ONE launch
    (vt_pointnet_mlp_fused; inference): [B,T,c_dim], bit-identical to the launch-per-layer path.  ``want_grid``: instead of the
    point features, the voxeliser's channels-last mean grid [B,R,R,R,c_dim] and its GroupNorm partial sums (part, nblk) from
    the same kernel (scatter_mean + channel_stats without their launches and the pass over the grid); ``zeroed_grid``: that grid,
    already cleared (VoxelIndex(clear=...))."""
    p = _c(p.float())
    B, T, _ = p.shape
    c_dim = fc_c.weight.shape[0]
    ptrs, ws, keep = weights if weights is not None else pointnet_mlp_weights(fc_pos, blocks, fc_c)
    lib = _lib.load()
    scratch = torch.empty((B, T, 32), dtype=torch.float32, device=p.device)
    out = grid = part = None
    nblk = 0
    if want_grid:
        R = vi.R
        if zeroed_grid is not None and tuple(zeroed_grid.shape) != (B, R, R, R, c_dim):
            raise VtError(f"pointnet_mlp_fused: zeroed_grid must be {(B, R, R, R, c_dim)}, got {tuple(zeroed_grid.shape)}")
        grid = zeroed_grid if zeroed_grid is not None else torch.zeros((B, R, R, R, c_dim), dtype=torch.float32, device=p.device)
        nblk = lib.vt_pointnet_mlp_stat_blocks(B, T)
        part = torch.empty((B, nblk, c_dim, 2), dtype=torch.float32, device=p.device)
    else:
        out = torch.empty((B, T, c_dim), dtype=torch.float32, device=p.device)
    check(lib.vt_pointnet_mlp_fused(dev_ptr(p, "p"), B, T, dev_ptr(vi.order, "order", I32), dev_ptr(vi.seg_lo, "seg_lo", I32),
                                    dev_ptr(vi.seg_hi, "seg_hi", I32), dev_ptr(keep[0], "fc_pos.weight"), dev_ptr(keep[1], "fc_pos.bias"),
                                    ptrs, 32, dev_ptr(keep[2], "fc_c.weight"), dev_ptr(keep[3], "fc_c.bias"), c_dim,
                                    dev_ptr(scratch, "scratch"), dev_ptr(out, "out"), dev_ptr(vi.idx, "idx", I32) if want_grid else None,
                                    vi.R if want_grid else 0, dev_ptr(grid, "grid"), dev_ptr(part, "part"), stream_ptr()), "vt_pointnet_mlp_fused")
    keep_for_graph(scratch, *ws, *keep)
    return (grid, (part, nblk)) if want_grid else out


def mano_bwd(pose, blob, center_idx, dverts, djoints):
    """d pose [B,48] of mano_fwd from d verts [B,778,3] and d joints [B,21,3] (vt_mano_bwd)."""
    pose = _c(pose.float())
    B = pose.shape[0]
    dverts = _c(dverts.float()) if dverts is not None else torch.zeros((B, 778, 3), dtype=torch.float32, device=pose.device)
    djoints = _c(djoints.float()) if djoints is not None else torch.zeros((B, 21, 3), dtype=torch.float32, device=pose.device)
    dpose = torch.empty((B, 48), dtype=torch.float32, device=pose.device)
    check(_lib.load().vt_mano_bwd(dev_ptr(pose, "pose"), B, dev_ptr(blob, "blob"), -1 if center_idx is None else int(center_idx),
                                  dev_ptr(dverts, "dverts"), dev_ptr(djoints, "djoints"), dev_ptr(dpose, "dpose"), stream_ptr()), "vt_mano_bwd")
    return dpose


# --------------------------------------------------------------------------------------
# decode backward (training)
# --------------------------------------------------------------------------------------
def decode_save_buffer(total_points, device):
    n = _lib.load().vt_decode_save_bytes(total_points) // 4
    return torch.empty(n, dtype=torch.float32, device=device)


def decode_bwd(grid_shape, blob_t, grad_out, save, pts=None, lattice=None, with_c_img=False, c_img=None,
               padding=0.1, want_grid_grad=True, grad_out2=None):
    """vt_decode_bwd + vt_decode_wgrad (with ``grad_out2``, the contact head's logit gradient: the _contact forms).
    Returns (grad_grid [B,C,R,R,R] channels-last strided or None, grad_c_img [B,N,C] or None, flat parameter gradients)."""
    lib = _lib.load()
    B, C, R = grid_shape[0], grid_shape[1], grid_shape[2]
    grad_out = _c(grad_out.float())
    dev = grad_out.device
    if pts is not None:
        pts = _c(pts.float())
        N = pts.shape[1]
        nx, box, first = 0, 0.0, 0
    else:
        nx, box, first, N = lattice
    total = B * N
    ggrid = torch.zeros((B, R, R, R, C), dtype=torch.float32, device=dev) if want_grid_grad else None
    gimg = torch.empty((B, N, C), dtype=torch.float32, device=dev) if with_c_img else None
    g2 = _c(grad_out2.float()) if grad_out2 is not None else None
    if total == 0:                                                  # an empty query set: every gradient is zero, no launch
        p_in = 3 + C if with_c_img else 3
        nflat = lib.vt_decode_wgrad_floats_contact(p_in) if g2 is not None else lib.vt_decode_wgrad_floats(p_in)
        return ((ggrid.permute(0, 4, 1, 2, 3) if ggrid is not None else None), gimg,
                torch.zeros(nflat, dtype=torch.float32, device=dev))
    gws = torch.empty(lib.vt_decode_gws_bytes(total) // 4, dtype=torch.float32, device=dev)
    st = stream_ptr()
    if ggrid is not None and pts is not None and GRID_SCATTER_SORTED and R >= 3:
        # grid gradient by cell (vt_sample_grid_bwd_sorted): the data pass leaves d c, the points are binned by trilinear cell
        # (vt_voxel_build at R - 1) and every cell scatters once -- training points cluster (contact clouds), and per-point f32
        # atomics that collide were 0.8 of the 0.9 ms this call took in a training step
        dc = torch.empty((total, C), dtype=torch.float32, device=dev)
        check(lib.vt_decode_bwd_dc(B, R, C, dev_ptr(pts, "pts"), N, float(padding), dev_ptr(blob_t, "blob_t"),
                                   dev_ptr(grad_out, "grad_out"), dev_ptr(g2, "grad_out2"), dev_ptr(save, "save"), dev_ptr(gws, "gws"),
                                   dev_ptr(dc, "grad_c"), dev_ptr(gimg, "grad_c_img"), st), "vt_decode_bwd_dc")
        sample_grid_bwd_sorted_into(ggrid, pts, dc, padding)
    else:
        check(lib.vt_decode_bwd_contact(B, R, C, dev_ptr(pts, "pts"), N, nx, box, first, float(padding),
                                        dev_ptr(blob_t, "blob_t"), dev_ptr(grad_out, "grad_out"), dev_ptr(g2, "grad_out2"),
                                        dev_ptr(save, "save"), dev_ptr(gws, "gws"), dev_ptr(ggrid, "grad_grid"),
                                        dev_ptr(gimg, "grad_c_img"), st), "vt_decode_bwd")
    wsb = lib.vt_decode_wgrad_workspace_bytes(total)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    p_in = 3 + C if with_c_img else 3
    nflat = lib.vt_decode_wgrad_floats_contact(p_in) if g2 is not None else lib.vt_decode_wgrad_floats(p_in)
    flat = torch.empty(nflat, dtype=torch.float32, device=dev)
    ci = _c(c_img) if with_c_img else None
    check(lib.vt_decode_wgrad_contact(B, dev_ptr(pts, "pts"), N, nx, box, first, dev_ptr(ci, "c_img"),
                                      dev_ptr(grad_out, "grad_out"), dev_ptr(g2, "grad_out2"), dev_ptr(save, "save"),
                                      dev_ptr(gws, "gws"), ctypes.c_void_p(ws.data_ptr()), wsb, dev_ptr(flat, "grads"), st),
          "vt_decode_wgrad")
    return (ggrid.permute(0, 4, 1, 2, 3) if ggrid is not None else None), gimg, flat


def split_decoder_grads(flat, p_in, hidden=32, c_dim=32, nb=5):
    """Views of the flat gradient buffer in the order documented in vtaco_hip.h."""
    o = 0

    def take(n, shape):
        nonlocal o
        v = flat[o:o + n].view(shape)
        o += n
        return v
    g = {"fc_p.weight": take(hidden * p_in, (hidden, p_in)), "fc_p.bias": take(hidden, (hidden,))}
    g["fc_c.weight"] = take(nb * hidden * c_dim, (nb, hidden, c_dim))
    g["fc_c.bias"] = take(nb * hidden, (nb, hidden))
    g["fc_0.weight"] = take(nb * hidden * hidden, (nb, hidden, hidden))
    g["fc_0.bias"] = take(nb * hidden, (nb, hidden))
    g["fc_1.weight"] = take(nb * hidden * hidden, (nb, hidden, hidden))
    g["fc_1.bias"] = take(nb * hidden, (nb, hidden))
    g["fc_out.weight"] = take(hidden, (1, hidden))
    g["fc_out.bias"] = take(1, (1,))
    if flat.numel() - o >= hidden + 1:                  # the _contact layout
        g["fc_out_contact.weight"] = take(hidden, (1, hidden))
        g["fc_out_contact.bias"] = take(1, (1,))
    return g


# ---- the wide decoder (hidden_size / c_dim beyond 32 / 32) under autograd ---------------------------------------------
def _wide_flags(leaky, nearest):
    return (1 if leaky else 0) | (2 if nearest else 0)          # VT_WIDE_LEAKY | VT_WIDE_NEAREST


def decode_fwd_wide_train(grid, blob, pts, c_img, hidden, nb, leaky, nearest, padding=0.1, want_contact=False):
    """vt_decode_fwd_wide_train: logits [B,N] (and the contact logits) plus the saved layer inputs (opaque f32 tensor)."""
    lib = _lib.load()
    B, C, D, H, W = grid.shape
    keep, gptr = _cl_storage(grid)
    pts = _c(pts.detach().float())
    N = pts.shape[1]
    ci = _c(c_img.detach().float()) if c_img is not None else None
    dev = grid.device
    out = torch.empty((B, N), dtype=torch.float32, device=dev)
    out2 = torch.empty((B, N), dtype=torch.float32, device=dev) if want_contact else None
    # (an empty query set is not 'shape not built': the shape is judged on one point, the save of no points is empty)
    if lib.vt_decode_wide_save_floats(1, int(hidden), C, int(nb)) == 0:
        raise VtError(f"decoder shape hidden={hidden}, c_dim={C}, n_blocks={nb} is not built (multiples of 32 up to 256)")
    nsave = lib.vt_decode_wide_save_floats(B * N, int(hidden), C, int(nb)) if B * N else 0
    save = torch.empty(nsave, dtype=torch.float32, device=dev)
    if N:
        check(lib.vt_decode_fwd_wide_train(gptr, B, D, C, dev_ptr(pts, "pts"), N, dev_ptr(ci, "c_img"), dev_ptr(blob, "blob"),
                                           int(hidden), int(nb), _wide_flags(leaky, nearest), float(padding), dev_ptr(out, "out"),
                                           dev_ptr(out2, "out2"), dev_ptr(save, "save"), stream_ptr()), "vt_decode_fwd_wide_train")
    return out, out2, save


def pack_decoder_wide_t(fc_p_w, fc_c, blocks, fc_out_w, fc_out2_w=None):
    """vt_decoder_pack_wide_t: the transposed weight fragments vt_decode_bwd_wide streams."""
    lib = _lib.load()
    hidden, p_in = fc_p_w.shape
    c_dim, nb = fc_c[0].shape[1], len(blocks)
    keep = []

    def ptr(t, name):
        t = t.detach()
        t = t if t.is_contiguous() else t.contiguous()
        keep.append(t)
        return dev_ptr(t, name)
    prm = _lib.DecoderParams()
    prm.hidden, prm.c_dim, prm.n_blocks, prm.p_in = hidden, c_dim, nb, p_in
    prm.fc_p_w = ptr(fc_p_w, "fc_p.weight")
    for i, w in enumerate(fc_c):
        prm.fc_c_w[i] = ptr(w, f"fc_c.{i}.weight").value
    for i, (w0, w1) in enumerate(blocks):
        prm.fc0_w[i], prm.fc1_w[i] = ptr(w0, "fc_0.weight").value, ptr(w1, "fc_1.weight").value
    prm.fc_out_w = ptr(fc_out_w, "fc_out.weight")
    if fc_out2_w is not None:
        prm.fc_out2_w = ptr(fc_out2_w, "fc_out_contact.weight")
    n = lib.vt_decoder_wide_blob_t_bytes(hidden, c_dim, nb)
    if n == 0:
        raise VtError(f"decoder shape hidden={hidden}, c_dim={c_dim}, n_blocks={nb} is not built (multiples of 32 up to 256)")
    out = torch.empty(n // 4, dtype=torch.float32, device=fc_p_w.device)
    check(lib.vt_decoder_pack_wide_t(ctypes.byref(prm), dev_ptr(out, "blob_t"), n, stream_ptr()), "vt_decoder_pack_wide_t")
    return out


def _wide_zero_grads(H, C, nb, p_in, dev, contact):
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    g = {"fc_p.weight": z(H, p_in), "fc_p.bias": z(H), "fc_c.weight": [z(H, C) for _ in range(nb)], "fc_c.bias": [z(H) for _ in range(nb)],
         "fc_0.weight": [z(H, H) for _ in range(nb)], "fc_0.bias": [z(H) for _ in range(nb)],
         "fc_1.weight": [z(H, H) for _ in range(nb)], "fc_1.bias": [z(H) for _ in range(nb)],
         "fc_out.weight": z(1, H), "fc_out.bias": z(1)}
    if contact:
        g["fc_out_contact.weight"], g["fc_out_contact.bias"] = z(1, H), z(1)
    return g


def _wide_wgrads(save, gws, P, H, C, nb, pts, c_rows, c_img, grad_out, g2):
    """The decoder's parameter gradients from the rows the data pass leaves (vt_rows_wgrad per layer; layouts: decode_wide.hip
    wide_save_layout / wide_gws_layout).  ``c_rows`` [P, C]: the conditioning features (the save's c slot, or the caller's)."""
    sv_blk = save[P * C:P * C + 2 * nb * P * H].view(nb, 2, P, H)
    sv_af = save[P * C + 2 * nb * P * H:].view(P, H)
    dn = gws[:(nb + 1) * P * H].view(nb + 1, P, H)
    dh = gws[(nb + 1) * P * H:].view(nb, P, H)
    g = {}
    x2 = _c(c_img.float()).view(P, C) if c_img is not None else None
    g["fc_p.weight"], g["fc_p.bias"] = rows_wgrad(dn[0], pts.view(P, 3), x2)
    wc, bc, w0, b0, w1, b1 = [], [], [], [], [], []
    for i in range(nb):
        a, b = rows_wgrad(dn[i], c_rows); wc.append(a); bc.append(b)
        a, b = rows_wgrad(dh[i], sv_blk[i, 0]); w0.append(a); b0.append(b)
        a, b = rows_wgrad(dn[i + 1], sv_blk[i, 1]); w1.append(a); b1.append(b)
    g["fc_c.weight"], g["fc_c.bias"] = wc, bc
    g["fc_0.weight"], g["fc_0.bias"], g["fc_1.weight"], g["fc_1.bias"] = w0, b0, w1, b1
    g["fc_out.weight"], g["fc_out.bias"] = rows_wgrad(grad_out.view(P, 1), sv_af)
    if g2 is not None:
        g["fc_out_contact.weight"], g["fc_out_contact.bias"] = rows_wgrad(g2.view(P, 1), sv_af)
    return g


def decode_bwd_wide(grid_shape, blob_t, grad_out, save, pts, hidden, nb, leaky, nearest, padding=0.1, c_img=None,
                    want_grid_grad=True, grad_out2=None):
    """vt_decode_bwd_wide + the weight gradients (vt_rows_wgrad over the saved layer inputs and the output gradients the data pass
    leaves).  Returns (grad_grid channels-last strided [B,C,R,R,R] or None, grad_c_img [B,N,C] or None, dict of parameter gradients
    keyed like split_decoder_grads)."""
    lib = _lib.load()
    B, C, R = grid_shape[0], grid_shape[1], grid_shape[2]
    H = int(hidden)
    grad_out = _c(grad_out.float())
    dev = grad_out.device
    pts = _c(pts.float())
    N = pts.shape[1]
    P = B * N
    g2 = _c(grad_out2.float()) if grad_out2 is not None else None
    ggrid = torch.zeros((B, R, R, R, C), dtype=torch.float32, device=dev) if want_grid_grad else None
    gimg = torch.empty((B, N, C), dtype=torch.float32, device=dev) if c_img is not None else None
    if P == 0:                                                      # an empty query set: every gradient is zero, no launch
        g = _wide_zero_grads(H, C, nb, 3 + C if c_img is not None else 3, dev, g2 is not None)
        return (ggrid.permute(0, 4, 1, 2, 3) if ggrid is not None else None), gimg, g
    gws = torch.empty(lib.vt_decode_wide_gws_floats(P, H, C, int(nb)), dtype=torch.float32, device=dev)
    check(lib.vt_decode_bwd_wide(B, R, C, dev_ptr(pts, "pts"), N, dev_ptr(blob_t, "blob_t"), H, int(nb), _wide_flags(leaky, nearest),
                                 float(padding), dev_ptr(grad_out, "grad_out"), dev_ptr(g2, "grad_out2"), dev_ptr(save, "save"),
                                 dev_ptr(gws, "gws"), dev_ptr(ggrid, "grad_grid"), dev_ptr(gimg, "grad_c_img"), stream_ptr()),
          "vt_decode_bwd_wide")
    g = _wide_wgrads(save, gws, P, H, C, int(nb), pts, save[:P * C].view(P, C), c_img, grad_out, g2)
    return (ggrid.permute(0, 4, 1, 2, 3) if ggrid is not None else None), gimg, g


def decode_mlp_fwd_wide_train(c, blob, pts, hidden, nb, leaky):
    """vt_decode_mlp_fwd_wide_train: the conditioned MLP on given features c [B,N,C] at the wide shapes, keeping every layer's input."""
    lib = _lib.load()
    c = _c(c.detach().float())
    pts = _c(pts.detach().float())
    B, N, C = c.shape
    if lib.vt_decode_wide_save_floats(1, int(hidden), C, int(nb)) == 0:
        raise VtError(f"decoder shape hidden={hidden}, c_dim={C}, n_blocks={nb} is not built (multiples of 32 up to 256)")
    out = torch.empty((B, N), dtype=torch.float32, device=c.device)
    save = torch.empty(lib.vt_decode_wide_save_floats(B * N, int(hidden), C, int(nb)) if B * N else 0, dtype=torch.float32, device=c.device)
    if B * N:
        check(lib.vt_decode_mlp_fwd_wide_train(dev_ptr(c, "c"), B, C, dev_ptr(pts, "pts"), N, dev_ptr(blob, "blob"), int(hidden), int(nb),
                                               _wide_flags(leaky, False), dev_ptr(out, "out"), None, dev_ptr(save, "save"), stream_ptr()),
              "vt_decode_mlp_fwd_wide_train")
    return out, save


def decode_mlp_bwd_wide(blob_t, grad_out, save, pts, c, hidden, nb, leaky):
    """vt_decode_mlp_bwd_wide + the weight gradients: (grad_c [B,N,C], dict of parameter gradients keyed like split_decoder_grads)."""
    lib = _lib.load()
    grad_out = _c(grad_out.float())
    pts = _c(pts.float())
    c = _c(c.float())
    B, N, C = c.shape
    H, P, dev = int(hidden), B * N, grad_out.device
    grad_c = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    if P == 0:
        return grad_c, _wide_zero_grads(H, C, nb, 3, dev, False)
    gws = torch.empty(lib.vt_decode_wide_gws_floats(P, H, C, int(nb)), dtype=torch.float32, device=dev)
    check(lib.vt_decode_mlp_bwd_wide(B, C, dev_ptr(pts, "pts"), N, dev_ptr(blob_t, "blob_t"), H, int(nb), _wide_flags(leaky, False),
                                     dev_ptr(grad_out, "grad_out"), None, dev_ptr(save, "save"), dev_ptr(gws, "gws"),
                                     dev_ptr(grad_c, "grad_c"), stream_ptr()), "vt_decode_mlp_bwd_wide")
    return grad_c, _wide_wgrads(save, gws, P, H, C, int(nb), pts, c.view(P, C), None, grad_out, None)


# --------------------------------------------------------------------------------------
# AttentionDecoder pieces: sample-only, MLP-only, TransformerFusion
# --------------------------------------------------------------------------------------
def sample_grid(grid, pts=None, padding=0.1, lattice=None):
    """Trilinear features [B,N,C] of ``grid`` at ``pts`` [B,N,3], or with ``lattice=(nx, box, first, count)`` at the points
    ``box * make_3d_grid(...)[first:first+count]`` generated in the kernel (vt_sample_grid; slabs of whole x-plane pairs with
    nx % 8 == 0 and < 0.55 voxels per step run the LDS-staged gather: the same bits, ~2.5x the rate)."""
    B, C, D, H, W = grid.shape
    keep, gptr = _cl_storage(grid)
    if pts is not None:
        pts = _c(pts.float())
        N, nx, box, first = pts.shape[1], 0, 0.0, 0
    else:
        nx, box, first, N = lattice
    feat = torch.empty((B, N, C), dtype=torch.float32, device=grid.device)
    if N:
        check(_lib.load().vt_sample_grid(gptr, B, D, C, dev_ptr(pts, "pts"), N, int(nx), float(box), int(first), float(padding),
                                         dev_ptr(feat, "feat"), stream_ptr()), "vt_sample_grid")
    return feat


GRID_SCATTER_SORTED = os.environ.get("VTACO_GRID_SCATTER", "sorted") != "points"     # "points": one set of atomics per point (round 1-3)


def sample_grid_bwd_sorted_into(ggrid_cl, pts, grad_feat, padding=0.1):
    """Scatter d feat [B,N,C] of query points pts [B,N,3] into the zeroed channels-last grid gradient [B,R,R,R,C], the points grouped
    by trilinear cell (vt_voxel_build at resolution R - 1 + vt_sample_grid_bwd_sorted)."""
    B, R, C = ggrid_cl.shape[0], ggrid_cl.shape[1], ggrid_cl.shape[4]
    N = pts.shape[1]
    if B * N == 0:                                                  # no points: the zeroed gradient is the answer
        return
    vi = VoxelIndex(pts, R - 1, padding)
    check(_lib.load().vt_sample_grid_bwd_sorted(B, R, C, dev_ptr(pts, "pts"), N, float(padding), dev_ptr(grad_feat, "grad_feat"),
                                                dev_ptr(vi.order, "order", I32), dev_ptr(vi.seg_lo, "seg_lo", I32),
                                                dev_ptr(vi.seg_hi, "seg_hi", I32), dev_ptr(ggrid_cl, "grad_grid"), stream_ptr()),
          "vt_sample_grid_bwd_sorted")


def sample_grid_bwd(grid_shape, pts, grad_feat, padding=0.1):
    """Backward of :func:`sample_grid` w.r.t. the grid (vt_sample_grid_bwd): [B,C,R,R,R] with channels-last strides."""
    B, C, R = grid_shape[0], grid_shape[1], grid_shape[2]
    pts = _c(pts.float())
    grad_feat = _c(grad_feat.float())
    ggrid = torch.zeros((B, R, R, R, C), dtype=torch.float32, device=grad_feat.device)
    if GRID_SCATTER_SORTED and R >= 3:
        sample_grid_bwd_sorted_into(ggrid, pts, grad_feat, padding)
        return ggrid.permute(0, 4, 1, 2, 3)
    check(_lib.load().vt_sample_grid_bwd(B, R, C, dev_ptr(pts, "pts"), pts.shape[1], 0, 0.0, 0, float(padding),
                                         dev_ptr(grad_feat, "grad_feat"), dev_ptr(ggrid, "grad_grid"), stream_ptr()),
          "vt_sample_grid_bwd")
    return ggrid.permute(0, 4, 1, 2, 3)


def decode_mlp_fwd_train(c, blob, pts):
    """:func:`decode_mlp_fwd` that also returns the activations its backward needs (vt_decode_mlp_fwd_train)."""
    lib = _lib.load()
    c = _c(c)
    pts = _c(pts.float())
    B, N, C = c.shape
    out = torch.empty((B, N), dtype=torch.float32, device=c.device)
    save = torch.empty(lib.vt_decode_save_bytes(B * N) // 4, dtype=torch.float32, device=c.device)
    check(lib.vt_decode_mlp_fwd_train(dev_ptr(c, "c"), B, C, dev_ptr(pts, "pts"), N, 0, 0.0, 0,
                                      dev_ptr(blob, "blob"), dev_ptr(out, "out"), dev_ptr(save, "save"), stream_ptr()),
          "vt_decode_mlp_fwd_train")
    return out, save


def decode_mlp_bwd(blob_t, grad_out, save, pts, C=32):
    """vt_decode_mlp_bwd + vt_decode_wgrad: (grad_c [B,N,C], flat parameter gradients with p_in = 3)."""
    lib = _lib.load()
    pts = _c(pts.float())
    grad_out = _c(grad_out.float())
    B, N = grad_out.shape
    dev = grad_out.device
    total = B * N
    gws = torch.empty(lib.vt_decode_gws_bytes(total) // 4, dtype=torch.float32, device=dev)
    grad_c = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    st = stream_ptr()
    check(lib.vt_decode_mlp_bwd(B, C, dev_ptr(pts, "pts"), N, 0, 0.0, 0, dev_ptr(blob_t, "blob_t"), dev_ptr(grad_out, "grad_out"),
                                dev_ptr(save, "save"), dev_ptr(gws, "gws"), dev_ptr(grad_c, "grad_c"), st), "vt_decode_mlp_bwd")
    wsb = lib.vt_decode_wgrad_workspace_bytes(total)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    flat = torch.empty(lib.vt_decode_wgrad_floats(3), dtype=torch.float32, device=dev)
    check(lib.vt_decode_wgrad(B, dev_ptr(pts, "pts"), N, 0, 0.0, 0, None, dev_ptr(grad_out, "grad_out"), dev_ptr(save, "save"),
                              dev_ptr(gws, "gws"), ctypes.c_void_p(ws.data_ptr()), wsb, dev_ptr(flat, "grads"), st), "vt_decode_wgrad")
    return grad_c, flat


def decode_mlp_fwd(c, blob, pts, precision="f32", wide=None):
    """The conditioned MLP on given features c [B,N,C] (vt_decode_mlp_fwd; ``precision="f16x3"`` with a blob packed for it:
    vt_decode_mlp_fwd_f16x3).  ``precision="wide"`` / ``"wide_f16x3"`` with ``wide=(hidden_size, n_blocks, leaky)``: the shapes
    beyond 32 / 32 (vt_decode_mlp_fwd_wide[_f16x3], blob from pack_decoder(..., precision="wide" / "wide_f16x3"))."""
    if precision not in ("f32", "f16x3", "wide", "wide_f16x3"):
        raise VtError(f"decode_mlp_fwd: precision must be 'f32', 'f16x3', 'wide' or 'wide_f16x3' (got {precision!r})")
    c = _c(c.float())
    pts = _c(pts.float())
    B, N, C = c.shape
    out = torch.empty((B, N), dtype=torch.float32, device=c.device)
    if N and precision in ("wide", "wide_f16x3"):
        if wide is None:
            raise VtError("decode_mlp_fwd: precision 'wide' needs wide=(hidden_size, n_blocks, leaky)")
        hidden, nb, leaky = wide[:3]
        name = "vt_decode_mlp_fwd_wide" if precision == "wide" else "vt_decode_mlp_fwd_wide_f16x3"
        check(getattr(_lib.load(), name)(dev_ptr(c, "c"), B, C, dev_ptr(pts, "pts"), N, 0, 0.0, 0, dev_ptr(blob, "blob"),
                                         int(hidden), int(nb), 1 if leaky else 0, dev_ptr(out, "out"), None, stream_ptr()), name)
        return out
    if N:
        name = "vt_decode_mlp_fwd" if precision == "f32" else "vt_decode_mlp_fwd_f16x3"
        check(getattr(_lib.load(), name)(dev_ptr(c, "c"), B, C, dev_ptr(pts, "pts"), N, 0, 0.0, 0,
                                         dev_ptr(blob, "blob"), dev_ptr(out, "out"), stream_ptr()), name)
    return out


FUSION_TENSORS = tuple(n for n, _t in _lib.FusionUnit._fields_)


def _fusion_params(self_attn, cross_attn, C, keep):
    def unit(d):
        u = _lib.FusionUnit()
        for name in FUSION_TENSORS:
            t = _c(d[name].detach())
            keep.append(t)
            setattr(u, name, dev_ptr(t, name).value)
        return u
    prm = _lib.FusionParams()
    prm.d_model, prm.key_dim = C, self_attn["WK"].shape[0]
    prm.self_attn, prm.cross_attn = unit(self_attn), unit(cross_attn)
    return prm


def fusion_fwd(c_img, c, self_attn, cross_attn):
    """TransformerFusion forward, eval mode (vt_fusion_fwd).  ``self_attn`` / ``cross_attn``:
    dicts with the ten tensors of a vt_fusion_unit."""
    lib = _lib.load()
    c_img, c = _c(c_img.float()), _c(c.float())
    B, N, C = c.shape
    if tuple(c_img.shape) != (B, N, C):
        raise VtError(f"fusion: c_img {tuple(c_img.shape)} and c {tuple(c.shape)} must match")
    keep = []
    prm = _fusion_params(self_attn, cross_attn, C, keep)
    nbytes = lib.vt_fusion_workspace_bytes_wide(B, N, C)              # (d_model 32, or the generic-width kernels up to 128)
    if nbytes == 0:
        raise VtError(f"fusion: d_model = {C} is not built (32, 64, 96 or 128 with key_feature_dim 64)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=c.device)
    out = torch.empty((B, N, C), dtype=torch.float32, device=c.device)
    check(lib.vt_fusion_fwd(dev_ptr(c_img, "c_img"), dev_ptr(c, "c"), B, N, ctypes.byref(prm),
                            ctypes.c_void_p(ws.data_ptr()), nbytes, dev_ptr(out, "out"), stream_ptr()), "vt_fusion_fwd")
    return out


def fusion_fwd_ids(finger_ids, finger_feats, c, self_attn, cross_attn, chunk_index=None):
    """TransformerFusion forward, eval mode, with the tactile rows by finger id (vt_fusion_fwd_ids): ``finger_ids`` uint8 [rows, N]
    (255 = none), ``finger_feats`` [F, C]; batch element b of ``c`` [B, N, C] reads ids row ``chunk_index[b]`` (int32 [B] on the
    device) or row b.  No [B, N, C] tensor of gathered features exists."""
    lib = _lib.load()
    c = _c(c.float())
    B, N, C = c.shape
    ids, feats = _c(finger_ids), _c(finger_feats.detach().float())
    if ids.dtype != torch.uint8 or ids.dim() != 2 or ids.shape[1] != N or feats.dim() != 2 or feats.shape[1] != C:
        raise VtError(f"fusion_fwd_ids: finger ids must be uint8 [rows,{N}] with a [F,{C}] table (got {tuple(ids.shape)}, {tuple(feats.shape)})")
    if chunk_index is None and ids.shape[0] != B:
        raise VtError(f"fusion_fwd_ids: {ids.shape[0]} id rows for {B} chunks and no chunk_index")
    ci = _c(chunk_index) if chunk_index is not None else None
    if ci is not None and (ci.dtype != torch.int32 or ci.numel() != B):
        raise VtError("fusion_fwd_ids: chunk_index must be int32 [B]")
    keep = []
    prm = _fusion_params(self_attn, cross_attn, C, keep)
    nbytes = lib.vt_fusion_workspace_bytes(B, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=c.device)
    out = torch.empty((B, N, C), dtype=torch.float32, device=c.device)
    check(lib.vt_fusion_fwd_ids(dev_ptr(ids, "finger_ids", torch.uint8), dev_ptr(feats, "finger_feats"), int(feats.shape[0]),
                                dev_ptr(ci, "chunk_index", torch.int32), dev_ptr(c, "c"), B, N, ctypes.byref(prm),
                                ctypes.c_void_p(ws.data_ptr()), nbytes, dev_ptr(out, "out"), stream_ptr()), "vt_fusion_fwd_ids")
    return out


def fusion_fwd_train(c_img, c, self_attn, cross_attn, p_drop=0.0, seed=0):
    """TransformerFusion forward for training (vt_fusion_fwd_train): dropout with probability ``p_drop`` (masks a function
    of ``seed``) and the O(N) state the backward needs.  Returns (out [B,N,C], saved: opaque uint8 tensor)."""
    lib = _lib.load()
    c_img, c = _c(c_img.float()), _c(c.float())
    B, N, C = c.shape
    if tuple(c_img.shape) != (B, N, C):
        raise VtError(f"fusion: c_img {tuple(c_img.shape)} and c {tuple(c.shape)} must match")
    keep = []
    prm = _fusion_params(self_attn, cross_attn, C, keep)
    nbytes, sbytes = lib.vt_fusion_workspace_bytes_wide(B, N, C), lib.vt_fusion_saved_bytes_wide(B, N, C)
    if not nbytes or not sbytes:
        raise VtError(f"fusion_fwd_train: d_model {C} is not built (32, 64, 96, 128)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=c.device)
    saved = torch.empty(sbytes, dtype=torch.uint8, device=c.device)
    out = torch.empty((B, N, C), dtype=torch.float32, device=c.device)
    check(lib.vt_fusion_fwd_train(dev_ptr(c_img, "c_img"), dev_ptr(c, "c"), B, N, ctypes.byref(prm), float(p_drop), int(seed),
                                  ctypes.c_void_p(ws.data_ptr()), nbytes, ctypes.c_void_p(saved.data_ptr()), sbytes,
                                  dev_ptr(out, "out"), stream_ptr()), "vt_fusion_fwd_train")
    return out, saved


def fusion_bwd(d_out, c_img, c, self_attn, cross_attn, saved, p_drop=0.0, seed=0):
    """Backward of ``fusion_fwd_train`` (vt_fusion_bwd).  Returns (d_c_img, d_c, grads_self, grads_cross): the last two are
    dicts name -> gradient tensor with the shapes of the unit's parameters (the self unit's: the sum of its two uses)."""
    lib = _lib.load()
    d_out, c_img, c = _c(d_out.float()), _c(c_img.float()), _c(c.float())
    B, N, C = c.shape
    keep = []
    prm = _fusion_params(self_attn, cross_attn, C, keep)
    grads = _lib.FusionGrads()
    outs = []
    for unit, src in ((grads.self_attn, self_attn), (grads.cross_attn, cross_attn)):
        g = {name: torch.empty_like(src[name], memory_format=torch.contiguous_format) for name in FUSION_TENSORS}
        for name in FUSION_TENSORS:
            setattr(unit, name, dev_ptr(g[name], "grad " + name).value)
        outs.append(g)
    d_c_img, d_c = torch.empty_like(c), torch.empty_like(c)
    nbytes = lib.vt_fusion_bwd_workspace_bytes_wide(B, N, C)
    if not nbytes:
        raise VtError(f"fusion_bwd: d_model {C} is not built (32, 64, 96, 128)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=c.device)
    check(lib.vt_fusion_bwd(dev_ptr(d_out, "d_out"), dev_ptr(c_img, "c_img"), dev_ptr(c, "c"), B, N, ctypes.byref(prm), float(p_drop),
                            int(seed), ctypes.c_void_p(saved.data_ptr()), saved.numel(), ctypes.c_void_p(ws.data_ptr()), nbytes,
                            dev_ptr(d_c_img, "d_c_img"), dev_ptr(d_c, "d_c"), ctypes.byref(grads), stream_ptr()), "vt_fusion_bwd")
    return d_c_img, d_c, outs[0], outs[1]


def fusion_dropout_mask(p_drop, seed, call, which, points, device, d_model=32):
    """The dropout factors (0 or 1/(1-p)) the fusion kernels apply: [points, 64] for which=0, [points, d_model] for which=1."""
    out = torch.empty((points, 64 if which == 0 else d_model), dtype=torch.float32, device=device)
    if d_model == 32:
        check(_lib.load().vt_fusion_dropout_mask(float(p_drop), int(seed), int(call), int(which), int(points), dev_ptr(out, "mask"),
                                                stream_ptr()), "vt_fusion_dropout_mask")
    else:
        check(_lib.load().vt_fusion_dropout_mask_wide(float(p_drop), int(seed), int(call), int(which), int(points), int(d_model),
                                                     dev_ptr(out, "mask"), stream_ptr()), "vt_fusion_dropout_mask_wide")
    return out


# --------------------------------------------------------------------------------------
# UNet3D forward, channels-last (vt_conv3d_* / vt_gn_* / vt_maxpool3d_cl / vt_conv1x1_cl)
# --------------------------------------------------------------------------------------
def voxel_scatter_mean_cl_fwd(feat, vi):
    """Scatter-mean into a channels-last grid [B,R,R,R,C]."""
    feat = _c(feat)
    B, T, C = feat.shape
    R = vi.R
    grid = torch.empty((B, R, R, R, C), dtype=torch.float32, device=feat.device)
    check(_lib.load().vt_voxel_scatter_mean_cl_fwd(dev_ptr(feat, "feat"), dev_ptr(vi.idx, "idx", I32),
                                                   dev_ptr(vi.order, "order", I32), dev_ptr(vi.seg_lo, "seg_lo", I32),
                                                   dev_ptr(vi.seg_hi, "seg_hi", I32), B, T, C, R, dev_ptr(grid, "grid"),
                                                   stream_ptr()), "vt_voxel_scatter_mean_cl_fwd")
    return grid


def voxel_scatter_mean_cl_bwd(grad_grid_cl, vi, C):
    """grad of voxel_scatter_mean_cl_fwd: grad_grid_cl [B,R,R,R,C] contiguous -> [B,T,C]."""
    g = torch.empty((vi.B, vi.T, C), dtype=torch.float32, device=grad_grid_cl.device)
    check(_lib.load().vt_voxel_scatter_mean_cl_bwd(dev_ptr(_c(grad_grid_cl), "grad_grid"), dev_ptr(vi.idx, "idx", I32),
                                                   dev_ptr(vi.seg_lo, "seg_lo", I32), dev_ptr(vi.seg_hi, "seg_hi", I32),
                                                   vi.B, vi.T, C, vi.R, dev_ptr(g, "grad_feat"), stream_ptr()),
          "vt_voxel_scatter_mean_cl_bwd")
    return g


def conv3d_pack(weight, precision="f32"):
    """Fragment-ordered copy of a [Cout,Cin,3,3,3] conv weight: f32 (vt_conv3d_pack), split-bf16 hi/lo fragments
    (vt_conv3d_pack_bf16x3; same size) or split-f16 tap-pair fragments (vt_conv3d_pack_f16x3; its own size)."""
    lib = _lib.load()
    Cout, Cin = weight.shape[0], weight.shape[1]
    n = lib.vt_conv3d_packed_floats_f16x3(Cout, Cin) if precision == "f16x3" else lib.vt_conv3d_packed_floats(Cout, Cin)
    if n == 0 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise VtError(f"conv3d_pack: unsupported weight shape {tuple(weight.shape)}")
    w = _c(weight)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    if precision == "f16x3_thin":      # the thin-tile / K-split kernels' fragments (vt_conv3d_pack_bf16x3's layout) with IEEE-half pairs
        check(lib.vt_conv3d_pack_f16x3_thin(dev_ptr(w, "w"), Cout, Cin, dev_ptr(out, "packed"), stream_ptr()), "vt_conv3d_pack_f16x3_thin")
        return out
    if precision not in PRECISIONS:
        raise VtError(f"precision must be one of {PRECISIONS} (got {precision!r})")
    if precision in SPLIT_PRECISIONS:
        name = "vt_conv3d_pack_" + precision
        check(getattr(lib, name)(dev_ptr(w, "w"), Cout, Cin, dev_ptr(out, "packed"), stream_ptr()), name)
    else:
        check(lib.vt_conv3d_pack(dev_ptr(w, "w"), Cout, Cin, dev_ptr(out, "packed"), stream_ptr()), "vt_conv3d_pack")
    return out


def conv3d_pack_t(weight):
    """vt_conv3d_pack_f16x3_t: the split-f16 fragments of the data-gradient conv of a [Cout,Cin,3,3,3] weight (channels swapped, taps
    flipped) without materialising weight.flip(2, 3, 4).transpose(0, 1)."""
    lib = _lib.load()
    Cout, Cin = weight.shape[0], weight.shape[1]
    n = lib.vt_conv3d_packed_floats_f16x3(Cin, Cout)
    if n == 0 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise VtError(f"conv3d_pack_t: unsupported weight shape {tuple(weight.shape)}")
    w = _c(weight)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.vt_conv3d_pack_f16x3_t(dev_ptr(w, "w"), Cout, Cin, dev_ptr(out, "packed"), stream_ptr()), "vt_conv3d_pack_f16x3_t")
    return out


def conv3d_pack_up(weight, c_skip):
    """The merged class weights of a decoder-entry conv's upsampled channels (vt_conv3d_pack_f16x3_up): ``weight``
    [Cout, c_skip + C2, 3, 3, 3] of the layer that reads [skip | upsample(low)]; None where the per-parity kernel does not
    take the channel counts."""
    lib = _lib.load()
    Cout, Cin = weight.shape[0], weight.shape[1]
    n = lib.vt_conv3d_up_packed_floats(Cout, Cin - c_skip) if 0 < c_skip < Cin else 0
    if n == 0 or tuple(weight.shape[2:]) != (3, 3, 3):
        return None
    w = _c(weight)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.vt_conv3d_pack_f16x3_up(dev_ptr(w, "w"), Cout, Cin, c_skip, dev_ptr(out, "packed"), stream_ptr()), "vt_conv3d_pack_f16x3_up")
    return out


def conv3d_up_covers(C1, C2, B, D, H, W, Cout):
    """Does the per-parity kernel take this decoder-entry layer (vt_conv3d_up_covers)?"""
    return bool(_lib.load().vt_conv3d_up_covers(int(C1), int(C2), int(B), int(D), int(H), int(W), int(Cout)))


def conv3d_gcr_final(x, ss, packed_w_f16x3, final_packed, final_bias):
    """relu(conv3x3x3(x * scale + shift)) followed by the final 1x1x1 conv (32 -> 32) in the same launch
    (vt_conv3d_gcr_f16x3_final); check ``final_fusable`` first."""
    B, D, H, W, C1 = x.shape
    out = torch.empty((B, D, H, W, 32), dtype=torch.float32, device=x.device)
    check(_lib.load().vt_conv3d_gcr_f16x3_final(dev_ptr(x, "x"), C1, None, 0, B, D, H, W, dev_ptr(ss, "scale_shift"),
                                                dev_ptr(packed_w_f16x3, "packed_w"), 32, dev_ptr(final_packed, "final_packed"),
                                                dev_ptr(final_bias, "final_bias"), dev_ptr(out, "out"), stream_ptr()),
          "vt_conv3d_gcr_f16x3_final")
    return out


def conv3d_gcr_final_keep(x, ss, packed_w_f16x3, final_packed, final_bias):
    """As conv3d_gcr_final, returning (y, out): y = relu(conv3x3x3(x * scale + shift)) is stored too
    (vt_conv3d_gcr_f16x3_final_keep: the training forward of the last layer + final conv)."""
    B, D, H, W, C1 = x.shape
    y = torch.empty((B, D, H, W, 32), dtype=torch.float32, device=x.device)
    out = torch.empty((B, D, H, W, 32), dtype=torch.float32, device=x.device)
    check(_lib.load().vt_conv3d_gcr_f16x3_final_keep(dev_ptr(x, "x"), C1, None, 0, B, D, H, W, dev_ptr(ss, "scale_shift"),
                                                     dev_ptr(packed_w_f16x3, "packed_w"), 32, dev_ptr(final_packed, "final_packed"),
                                                     dev_ptr(_c(final_bias), "final_bias"), dev_ptr(out, "out"), dev_ptr(y, "y_keep"),
                                                     stream_ptr()), "vt_conv3d_gcr_f16x3_final_keep")
    return y, out


def conv3d_skip_covers(x, Cout):
    """Does the persistent split-f16 kernel (the one that takes block flags) run a plain layer of this shape?"""
    B, D, H, W, C = x.shape
    return bool(_lib.load().vt_conv3d_stat_blocks_f16x3(B, D, H, W, C, Cout))


def conv3d_gcr_skip(x, ss, packed_w_f16x3, Cout, tile_flags, relu=True):
    """relu?(conv3x3x3(x * scale + shift)) with the taps of the flagged 8^3 blocks skipped (vt_conv3d_gcr_f16x3_skip; plain layers on
    the persistent split-f16 kernel): returns (out, (part, nblk)).  ``tile_flags`` [B, (D/8)(H/8)(W/8)] uint8, 1 = x is zero over the
    block's halo."""
    lib = _lib.load()
    B, D, H, W, C = x.shape
    nblk = lib.vt_conv3d_stat_blocks_f16x3(B, D, H, W, C, Cout)
    if not nblk:
        raise VtError("conv3d_gcr_skip: shape not covered by the split-f16 kernel")
    out = torch.empty((B, D, H, W, Cout), dtype=torch.float32, device=x.device)
    part = torch.empty((B, nblk, Cout, 2), dtype=torch.float32, device=x.device)
    check(lib.vt_conv3d_gcr_f16x3_skip(dev_ptr(x, "x"), C, B, D, H, W, dev_ptr(ss, "scale_shift"), dev_ptr(packed_w_f16x3, "packed_w"), Cout,
                                       int(relu), dev_ptr(tile_flags, "tile_flags", torch.uint8), dev_ptr(out, "out"), dev_ptr(part, "part"),
                                       stream_ptr()), "vt_conv3d_gcr_f16x3_skip")
    return out, (part, nblk)


def final_fusable(x, Cout):
    B, D, H, W, C1 = x.shape
    return bool(_lib.load().vt_conv3d_final_fusable(B, D, H, W, C1, Cout))


def conv1x1_pack_f16x3(weight):
    """Split-half A-operand fragments of a [32,32(,1,1,1)] final conv weight (vt_conv1x1_pack_f16x3), for the fused epilogue of
    vt_conv3d_gcr_f16x3_final."""
    w = _c(weight).reshape(weight.shape[0], -1)
    out = torch.empty(1024, dtype=torch.float32, device=w.device)
    check(_lib.load().vt_conv1x1_pack_f16x3(dev_ptr(w, "w"), w.shape[0], w.shape[1], dev_ptr(out, "packed"), stream_ptr()),
          "vt_conv1x1_pack_f16x3")
    return out


def stat_blocks(V):
    """Blocks of a GroupNorm statistics pass over V voxels: vt_unet3d_fwd's rule (unet3d.hip::stat_blocks), so that the per-layer path
    sums the same blocks in the same order."""
    return max(1, min(1024, V // (16 if V >= 16384 else 8)))


def channel_stats(x):
    """Per-block partial (sum, sumsq) of a channels-last tensor: (part, nblk)."""
    B, D, H, W, C = x.shape
    V = D * H * W
    nblk = stat_blocks(V)
    part = torch.empty((B, nblk, C, 2), dtype=torch.float32, device=x.device)
    check(_lib.load().vt_channel_stats(dev_ptr(x, "x"), B, V, C, nblk, dev_ptr(part, "part"), stream_ptr()), "vt_channel_stats")
    return part, nblk


def gn_scale_shift(x_stats, low_stats, C1, C2, B, voxels, gamma, beta, groups, eps, device):
    """GroupNorm statistics of [x | upsample(low)] from the producers' partial sums -> scale_shift [B,C,2]."""
    ss = torch.empty((B, C1 + C2, 2), dtype=torch.float32, device=device)
    p2, n2 = low_stats if low_stats is not None else (None, 0)
    check(_lib.load().vt_gn_scale_shift(dev_ptr(x_stats[0], "part1"), x_stats[1], C1, dev_ptr(p2, "part2"), n2, C2, B, voxels,
                                        groups, dev_ptr(_c(gamma), "gamma"), dev_ptr(_c(beta), "beta"), float(eps),
                                        dev_ptr(ss, "scale_shift"), stream_ptr()), "vt_gn_scale_shift")
    return ss


def conv3d_gcr(x, low, ss, packed_w, Cout, relu=True, packed_w_bf16x3=None, want_stats=True, packed_w_f16x3=None,
               in_absmax=None, thin_half=False, packed_w_up=None):
    """relu?(conv3x3x3(x_cat * scale + shift)) on channels-last tensors (``ss`` None: no normalisation);
    returns (out, (part, nblk) or None).  With ``packed_w_f16x3`` / ``packed_w_bf16x3`` the convolution runs on the
    16-bit matrix core with split operands where that kernel covers the shape (f16x3 first).  ``thin_half``: ``packed_w_bf16x3``
    holds conv3d_pack(..., "f16x3_thin") fragments and the thin-tile / K-split kernels run on IEEE-half pairs."""
    lib = _lib.load()
    B, D, H, W, C1 = x.shape
    C2 = low.shape[-1] if low is not None else 0
    dev = x.device
    st = stream_ptr()
    out = torch.empty((B, D, H, W, Cout), dtype=torch.float32, device=dev)
    if callable(packed_w) and packed_w_f16x3 is None and packed_w_bf16x3 is None:
        packed_w = packed_w()
    fn, name, pw = lib.vt_conv3d_gcr, "vt_conv3d_gcr", packed_w      # (a callable: packed on demand, only if the f32 kernel runs)
    nblk = lib.vt_conv3d_stat_blocks_f16x3(B, D, H, W, C1 + C2, Cout) if packed_w_f16x3 is not None else 0
    if nblk:
        fn, name, pw = lib.vt_conv3d_gcr_f16x3, "vt_conv3d_gcr_f16x3", packed_w_f16x3
    else:
        ksbytes = lib.vt_conv3d_ksplit_workspace_bytes(B, D, H, W, C1 + C2, Cout) if packed_w_bf16x3 is not None else 0
        if ksbytes:
            # thin level (16^3 / 8^3 of one scene): the input channels dealt over several workgroups per output tile
            nblk = lib.vt_conv3d_stat_blocks_ksplit(B, D, H, W, C1 + C2, Cout)
            part = torch.empty((B, nblk, Cout, 2), dtype=torch.float32, device=dev) if want_stats else None
            ws = torch.empty(ksbytes // 4, dtype=torch.float32, device=dev)
            kfn = lib.vt_conv3d_gcr_f16x3_thin_ksplit if thin_half else lib.vt_conv3d_gcr_bf16x3_ksplit
            check(kfn(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W, dev_ptr(ss, "scale_shift"),
                      dev_ptr(packed_w_bf16x3, "packed_w"), Cout, int(relu), dev_ptr(out, "out"),
                      dev_ptr(part, "part"), ctypes.c_void_p(ws.data_ptr()), ksbytes, st),
                  "vt_conv3d_gcr_f16x3_thin_ksplit" if thin_half else "vt_conv3d_gcr_bf16x3_ksplit")
            return out, ((part, nblk) if want_stats else None)
        nblk = lib.vt_conv3d_stat_blocks_bf16x3(B, D, H, W, C1 + C2, Cout) if packed_w_bf16x3 is not None else 0
        if nblk:
            fn, name, pw = lib.vt_conv3d_gcr_bf16x3, "vt_conv3d_gcr_bf16x3", packed_w_bf16x3
            if thin_half:
                fn, name = lib.vt_conv3d_gcr_f16x3_thin, "vt_conv3d_gcr_f16x3_thin"
        else:
            nblk = lib.vt_conv3d_stat_blocks(B, D, H, W, C1 + C2, Cout)
            if callable(pw):
                pw = pw()
    part = torch.empty((B, nblk, Cout, 2), dtype=torch.float32, device=dev) if want_stats else None
    if (packed_w_up is not None and low is not None and in_absmax is None and name == "vt_conv3d_gcr_f16x3"
            and lib.vt_conv3d_up_covers(C1, C2, B, D, H, W, Cout)):
        # decoder entry [skip | upsample(low)]: the low channels as a 2x2x2 conv per output parity class (conv3d_pack_up)
        check(lib.vt_conv3d_gcr_f16x3_up(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W, dev_ptr(ss, "scale_shift"),
                                         dev_ptr(pw, "packed_w"), dev_ptr(packed_w_up, "packed_up"), Cout, int(relu),
                                         dev_ptr(out, "out"), dev_ptr(part, "part"), st), "vt_conv3d_gcr_f16x3_up")
        return out, ((part, nblk) if want_stats else None)
    if in_absmax is not None and name == "vt_conv3d_gcr_f16x3":
        # input far below the half range (output gradients): the kernel rescales it by a power of two around the split
        check(lib.vt_conv3d_gcr_f16x3_scaled(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W, dev_ptr(ss, "scale_shift"),
                                             dev_ptr(pw, "packed_w"), Cout, int(relu), dev_ptr(out, "out"), dev_ptr(part, "part"),
                                             dev_ptr(in_absmax, "in_absmax"), st), "vt_conv3d_gcr_f16x3_scaled")
        return out, ((part, nblk) if want_stats else None)
    check(fn(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W, dev_ptr(ss, "scale_shift"),
             dev_ptr(pw, "packed_w"), Cout, int(relu), dev_ptr(out, "out"), dev_ptr(part, "part"), st), name)
    return out, ((part, nblk) if want_stats else None)


def gn_conv3d_relu(x, x_stats, low, low_stats, gamma, beta, groups, packed_w, Cout, eps=1e-5, relu=True,
                   packed_w_bf16x3=None, packed_w_f16x3=None, thin_half=False, packed_w_up=None):
    """relu(conv3x3x3(GroupNorm([x | upsample(low)]))) on channels-last tensors; the statistics
    come from the producers' partial sums.  Returns (out, out_stats)."""
    B, D, H, W, C1 = x.shape
    C2 = low.shape[-1] if low is not None else 0
    ss = gn_scale_shift(x_stats, low_stats if low is not None else None, C1, C2, B, D * H * W, gamma, beta, groups, eps, x.device)
    return conv3d_gcr(x, low, ss, packed_w, Cout, relu, packed_w_bf16x3, packed_w_f16x3=packed_w_f16x3, thin_half=thin_half,
                      packed_w_up=packed_w_up)


def relu_mask(dy, y, want_absmax=False):
    """g = dy where y > 0 else 0 (vt_relu_mask); with ``want_absmax`` also max |g| as a device scalar [1] from the same pass
    (vt_relu_mask_absmax): returns (g, absmax)."""
    dy = _c(dy)
    g = torch.empty_like(dy)
    if want_absmax:
        m = torch.empty(1, dtype=torch.float32, device=dy.device)
        check(_lib.load().vt_relu_mask_absmax(dev_ptr(dy, "dy"), dev_ptr(y, "y"), dev_ptr(g, "g"), dy.numel(), dev_ptr(m, "absmax"),
                                              stream_ptr()), "vt_relu_mask_absmax")
        return g, m
    check(_lib.load().vt_relu_mask(dev_ptr(dy, "dy"), dev_ptr(y, "y"), dev_ptr(g, "g"), dy.numel(), stream_ptr()), "vt_relu_mask")
    return g


def conv1x1_bwd_masked(dout, y, w, want_dw=True, want_db=True):
    """Backward of a 32 -> 32 pointwise conv out = y W^T + b whose input y is the ReLU output of the layer in front of it
    (vt_conv1x1_bwd_masked): returns (g, gmax, dw, db) with g = (y > 0) * (dout W) -- that layer's masked output gradient -- its
    max |g| as a device scalar, dw [32,32] and db [32] (None where not wanted)."""
    lib = _lib.load()
    if w.shape != (32, 32) or y.shape[-1] != 32 or dout.shape != y.shape:
        raise VtError(f"conv1x1_bwd_masked: built for 32 -> 32 channels over equal-shaped dout / y, got {tuple(w.shape)}, {tuple(dout.shape)}, {tuple(y.shape)}")
    dout, y, w = _c(dout), _c(y), _c(w)
    n = y.numel() // 32
    g = torch.empty_like(y)
    gmax = torch.empty(1, dtype=torch.float32, device=y.device)
    dw = torch.empty((32, 32), dtype=torch.float32, device=y.device) if want_dw else None
    db = torch.empty(32, dtype=torch.float32, device=y.device) if want_db else None
    nbytes = lib.vt_conv1x1_bwd_workspace_bytes()
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=y.device)
    check(lib.vt_conv1x1_bwd_masked(dev_ptr(dout, "dout"), dev_ptr(y, "y"), dev_ptr(w, "w"), n, dev_ptr(g, "g"), dev_ptr(gmax, "absmax"),
                                    dev_ptr(dw, "dw"), dev_ptr(db, "db"), ctypes.c_void_p(ws.data_ptr()), nbytes, stream_ptr()),
          "vt_conv1x1_bwd_masked")
    return g, gmax, dw, db


_WGRAD_UP = os.environ.get("VTACO_UNET_WGRAD_UP", "1") != "0"       # A/B knob: decoder-entry weight gradients per parity class


def conv3d_wgrad(x, low, ss, g, precision="f32", g_absmax=None):
    """dW [Cout,Cin,3,3,3] of the 3x3x3 conv over xn = [x | upsample(low)] * scale + shift (vt_conv3d_wgrad).
    ``precision="f16x3"``: vt_conv3d_wgrad_f16x3 where it covers the shape (split-half operands on the f16 matrix core;
    ``g_absmax`` = device scalar max |g| for its power-of-two rescale of g), the f32 kernel elsewhere."""
    lib = _lib.load()
    B, D, H, W, C1 = x.shape
    C2 = low.shape[-1] if low is not None else 0
    Cout = g.shape[-1]
    ubytes = (lib.vt_conv3d_wgrad_f16x3_up_workspace_bytes(B, D, H, W, C1, C2, Cout)
              if precision == "f16x3" and low is not None and _WGRAD_UP else 0)
    if ubytes:
        # a decoder entry: the upsampled channels per output parity class (2 x 2 x 2 taps over the low-resolution grid)
        ws = torch.empty(ubytes // 4, dtype=torch.float32, device=x.device)
        dw = torch.empty((Cout, C1 + C2, 3, 3, 3), dtype=torch.float32, device=x.device)
        check(lib.vt_conv3d_wgrad_f16x3_up(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W, dev_ptr(ss, "scale_shift"),
                                           dev_ptr(g, "g"), Cout, dev_ptr(g_absmax, "g_absmax"), ctypes.c_void_p(ws.data_ptr()), ubytes,
                                           dev_ptr(dw, "dw"), stream_ptr()), "vt_conv3d_wgrad_f16x3_up")
        return dw
    hbytes = lib.vt_conv3d_wgrad_f16x3_workspace_bytes(B, D, H, W, C1 + C2, Cout) if precision == "f16x3" else 0
    if hbytes:
        ws = torch.empty(hbytes // 4, dtype=torch.float32, device=x.device)
        dw = torch.empty((Cout, C1 + C2, 3, 3, 3), dtype=torch.float32, device=x.device)
        check(lib.vt_conv3d_wgrad_f16x3(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W, dev_ptr(ss, "scale_shift"),
                                        dev_ptr(g, "g"), Cout, dev_ptr(g_absmax, "g_absmax"), ctypes.c_void_p(ws.data_ptr()), hbytes,
                                        dev_ptr(dw, "dw"), stream_ptr()), "vt_conv3d_wgrad_f16x3")
        return dw
    nbytes = lib.vt_conv3d_wgrad_workspace_bytes(B, D, H, W, C1 + C2, Cout)
    if nbytes == 0:
        raise VtError("conv3d_wgrad: unsupported shape")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((Cout, C1 + C2, 3, 3, 3), dtype=torch.float32, device=x.device)
    check(lib.vt_conv3d_wgrad(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W, dev_ptr(ss, "scale_shift"),
                              dev_ptr(g, "g"), Cout, ctypes.c_void_p(ws.data_ptr()), nbytes, dev_ptr(dw, "dw"), stream_ptr()),
          "vt_conv3d_wgrad")
    return dw


def conv3d_wgrad_sparse(x, ss, g, tile_flags, g_absmax=None):
    """dW of a layer whose input is exactly zero over the blocks ``tile_flags`` marks (vt_conv3d_wgrad_f16x3_sparse: the taps over the
    other blocks' tiles + the GroupNorm shift's rank-one share); None where the shape is not on that kernel."""
    lib = _lib.load()
    B, D, H, W, C = x.shape
    Cout = g.shape[-1]
    nbytes = lib.vt_conv3d_wgrad_f16x3_sparse_workspace_bytes(B, D, H, W, C, Cout)
    if not nbytes or tile_flags is None or tile_flags.numel() != B * (D // 8) * (H // 8) * (W // 8):
        return None
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty((Cout, C, 3, 3, 3), dtype=torch.float32, device=x.device)
    check(lib.vt_conv3d_wgrad_f16x3_sparse(dev_ptr(x, "x"), C, B, D, H, W, dev_ptr(ss, "scale_shift"),
                                           dev_ptr(tile_flags, "tile_flags", torch.uint8), dev_ptr(g, "g"), Cout, dev_ptr(g_absmax, "g_absmax"), ctypes.c_void_p(ws.data_ptr()), nbytes,
                                           dev_ptr(dw, "dw"), stream_ptr()), "vt_conv3d_wgrad_f16x3_sparse")
    return dw


def conv3d_dgrad_xstats(g, packed_t, Cin, g_absmax, x):
    """The data gradient of a plain 'gcr' layer with the GroupNorm backward's sums from its epilogue (vt_conv3d_gcr_f16x3_xstats):
    ``g`` [B,D,H,W,Cout] the masked output gradient, ``packed_t`` = conv3d_pack_t(weight), ``x`` [B,D,H,W,Cin] the layer's input.
    Returns (dxn, (bpart, nblk)) or None where the shape is not on that kernel."""
    lib = _lib.load()
    B, D, H, W, C = g.shape
    nblk = lib.vt_conv3d_xstats_blocks(B, D, H, W, C, int(Cin)) if g_absmax is not None else 0
    if not nblk or tuple(x.shape) != (B, D, H, W, Cin):
        return None
    dxn = torch.empty((B, D, H, W, Cin), dtype=torch.float32, device=g.device)
    part = torch.empty((B, nblk, Cin, 2), dtype=torch.float32, device=g.device)
    check(lib.vt_conv3d_gcr_f16x3_xstats(dev_ptr(_c(g), "g"), C, B, D, H, W, dev_ptr(packed_t, "packed_w"), int(Cin), dev_ptr(g_absmax, "in_absmax"),
                                         dev_ptr(_c(x), "x"), dev_ptr(dxn, "out"), dev_ptr(part, "part"), stream_ptr()), "vt_conv3d_gcr_f16x3_xstats")
    return dxn, (part, nblk)


def gn_bwd(x, x_stats, low, low_stats, dxn, gamma, groups, eps, want_skip=True, want_low=True, mask_skip=False, mask_low=False, bpart=None):
    """GroupNorm backward of xn = GN([x | upsample(low)]) given dxn (vt_gn_bwd): returns
    (dskip or None, dlow or None, dgamma [C], dbeta [C]).  ``mask_skip`` / ``mask_low`` (vt_gn_bwd_masked): x / low is the ReLU
    output of the layer in front and this is its only gradient -- the gradient comes out masked by (x > 0) and the call returns
    (dskip, dlow, dgamma, dbeta, absmax_skip, absmax_low) with the device scalars max |gradient| (None where not asked): what that
    layer's relu_mask(..., want_absmax=True) would compute in a pass of its own.  ``bpart`` = (part, nblk) from conv3d_dgrad_xstats:
    the statistics pass over dxn and x is not launched (vt_gn_bwd_from_part)."""
    B, D, H, W, C1 = x.shape
    C2 = low.shape[-1] if low is not None else 0
    C = C1 + C2
    dev = x.device
    V = D * H * W
    have_part = bpart is not None
    if have_part:
        bpart, nblkb = bpart
    else:
        nblkb = max(1, min(1024, V // 64))
        bpart = torch.empty((B, nblkb, C, 2), dtype=torch.float32, device=dev)
    coef = torch.empty((B, C, 3), dtype=torch.float32, device=dev)
    dgb = torch.empty((B, C, 2), dtype=torch.float32, device=dev)
    dskip = torch.empty_like(x) if want_skip else None
    dlow = torch.empty_like(low) if (low is not None and want_low) else None
    p2, n2 = low_stats if low is not None else (None, 0)
    mask_skip = bool(mask_skip and dskip is not None)
    mask_low = bool(mask_low and dlow is not None)
    am_s = torch.empty(1, dtype=torch.float32, device=dev) if mask_skip else None
    am_l = torch.empty(1, dtype=torch.float32, device=dev) if mask_low else None
    # (dgamma, dbeta) summed over the scenes by the pass that writes the gradients, where there is one
    gsum = torch.empty((2, C), dtype=torch.float32, device=dev) if (dskip is not None or dlow is not None) else None
    fn = _lib.load().vt_gn_bwd_from_part if have_part else _lib.load().vt_gn_bwd_masked
    check(fn(dev_ptr(x, "x"), C1, dev_ptr(low, "low"), C2, B, D, H, W,
             dev_ptr(x_stats[0], "part1"), x_stats[1], dev_ptr(p2, "part2"), n2,
             dev_ptr(_c(dxn), "dxn"), groups, dev_ptr(_c(gamma), "gamma"), float(eps),
             dev_ptr(bpart, "bpart"), nblkb, dev_ptr(coef, "coef"), dev_ptr(dgb, "dgb"),
             dev_ptr(dskip, "dskip"), dev_ptr(dlow, "dlow"), (1 if mask_skip else 0) | (2 if mask_low else 0),
             dev_ptr(am_s, "absmax_skip"), dev_ptr(am_l, "absmax_low"), dev_ptr(gsum, "dgb_sum"), stream_ptr()),
          "vt_gn_bwd_from_part" if have_part else "vt_gn_bwd_masked")
    g = gsum if gsum is not None else dgb.sum(0).t().contiguous()              # [2, C]: dgamma, dbeta as rows
    if mask_skip or mask_low:
        return dskip, dlow, g[0], g[1], am_s, am_l
    return dskip, dlow, g[0], g[1]


def maxpool3d_cl_bwd(x, dy):
    B, D, H, W, C = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().vt_maxpool3d_cl_bwd(dev_ptr(x, "x"), dev_ptr(_c(dy), "dy"), B, D, H, W, C, dev_ptr(dx, "dx"), stream_ptr()),
          "vt_maxpool3d_cl_bwd")
    return dx


def maxpool3d_cl_bwd_fork(y, dskip, dpooled, want_absmax=True):
    """g = (y > 0 ? dskip + maxpool_backward(dpooled) : 0) for a tensor y that feeds a 2x2x2 max-pool and a skip connection
    (vt_maxpool3d_cl_bwd_fork); with ``want_absmax`` also the device scalar max |g|: returns (g, absmax or None)."""
    B, D, H, W, C = y.shape
    dskip, dpooled = _c(dskip), _c(dpooled)
    g = torch.empty_like(y)
    m = torch.empty(1, dtype=torch.float32, device=y.device) if want_absmax else None
    check(_lib.load().vt_maxpool3d_cl_bwd_fork(dev_ptr(y, "y"), dev_ptr(dskip, "dskip"), dev_ptr(dpooled, "dpooled"), B, D, H, W, C,
                                               dev_ptr(g, "g"), dev_ptr(m, "absmax"), stream_ptr()), "vt_maxpool3d_cl_bwd_fork")
    return g, m


def maxpool3d_cl(x):
    B, D, H, W, C = x.shape
    out = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    check(_lib.load().vt_maxpool3d_cl(dev_ptr(x, "x"), B, D, H, W, C, dev_ptr(out, "out"), stream_ptr()), "vt_maxpool3d_cl")
    return out


def maxpool3d_cl_stats(x):
    """2x2x2 max-pool and the pooled tensor's GroupNorm partial sums from one pass: (out, (part, nblk)) -- what maxpool3d_cl followed by
    channel_stats returns, bit for bit (vt_maxpool3d_cl_stats)."""
    B, D, H, W, C = x.shape
    out = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    V = (D // 2) * (H // 2) * (W // 2)
    nblk = stat_blocks(V)
    part = torch.empty((B, nblk, C, 2), dtype=torch.float32, device=x.device)
    check(_lib.load().vt_maxpool3d_cl_stats(dev_ptr(x, "x"), B, D, H, W, C, dev_ptr(out, "out"), nblk, dev_ptr(part, "part"), stream_ptr()),
          "vt_maxpool3d_cl_stats")
    return out, (part, nblk)


def conv1x1_cl(x, weight, bias):
    B, D, H, W, Cin = x.shape
    Cout = weight.shape[0]
    out = torch.empty((B, D, H, W, Cout), dtype=torch.float32, device=x.device)
    w = _c(weight).reshape(Cout, Cin)
    check(_lib.load().vt_conv1x1_cl(dev_ptr(x, "x"), B * D * H * W, Cin, dev_ptr(w, "w"),
                                    dev_ptr(_c(bias) if bias is not None else None, "bias"), Cout,
                                    dev_ptr(out, "out"), stream_ptr()), "vt_conv1x1_cl")
    return out


_unet_ws = {}


def unet3d_skip_layers(B, R, params):
    """How many layers of this UNet3D take the block flags at this batch and resolution (vt_unet3d_skip_layers): 0, 1 (the first layer)
    or 2 (the first DoubleConv: the second layer over the blocks whose 12^3 halo is empty)."""
    return int(_lib.load().vt_unet3d_skip_layers(int(B), int(R), ctypes.byref(params)))


def unet3d_fwd(x_cl, params, keep, in_stats=None, tile_flags=None):
    """Whole UNet3D forward (vt_unet3d_fwd).  ``params``: a filled _lib.UnetParams; ``keep``: the
    tensors its pointers refer to (kept alive by the caller).  ``in_stats`` = (part, nblk): GroupNorm partial sums of the
    input that its producer already has (vt_unet3d_fwd_stats: no statistics pass over the input).  ``tile_flags``
    (voxel_tile_flags): the 8^3 blocks over whose halo x is zero -- the first layer skips their taps (vt_unet3d_fwd_skip)."""
    lib = _lib.load()
    B, R = x_cl.shape[0], x_cl.shape[1]
    need = lib.vt_unet3d_workspace_bytes(B, R, ctypes.byref(params))
    if need == 0:
        raise VtError("unet3d_fwd: unsupported configuration: " + lib.vt_last_error().decode())
    key = (x_cl.device, need)
    ws = _unet_ws.get(key)
    if ws is None:
        _unet_ws.clear()
        ws = _unet_ws[key] = torch.empty(need, dtype=torch.uint8, device=x_cl.device)
    keep_for_graph(ws, *keep)
    out = torch.empty((B, R, R, R, params.out_channels), dtype=torch.float32, device=x_cl.device)
    if tile_flags is not None:
        keep_for_graph(tile_flags, *([in_stats[0]] if in_stats is not None else []))
        check(lib.vt_unet3d_fwd_skip(dev_ptr(x_cl, "x"), dev_ptr(in_stats[0], "in_part") if in_stats is not None else None,
                                     int(in_stats[1]) if in_stats is not None else 0, dev_ptr(tile_flags, "tile_flags", torch.uint8), B, R,
                                     ctypes.byref(params), ctypes.c_void_p(ws.data_ptr()), need, dev_ptr(out, "out"), stream_ptr()),
              "vt_unet3d_fwd_skip")
        return out
    if in_stats is not None:
        keep_for_graph(in_stats[0])
        check(lib.vt_unet3d_fwd_stats(dev_ptr(x_cl, "x"), dev_ptr(in_stats[0], "in_part"), int(in_stats[1]), B, R, ctypes.byref(params),
                                      ctypes.c_void_p(ws.data_ptr()), need, dev_ptr(out, "out"), stream_ptr()), "vt_unet3d_fwd_stats")
        return out
    check(lib.vt_unet3d_fwd(dev_ptr(x_cl, "x"), B, R, ctypes.byref(params), ctypes.c_void_p(ws.data_ptr()), need,
                            dev_ptr(out, "out"), stream_ptr()), "vt_unet3d_fwd")
    return out


# --------------------------------------------------------------------------------------
# tactile feature assignment by finger id (vt_tactile_assign / vt_decode_fwd_ids)
# --------------------------------------------------------------------------------------
U8 = torch.uint8


def tactile_assign(anchors, success, mode, radius, pts=None, lattice=None, count=None, B=1):
    """Finger id per query point (uint8, 255 = none).  anchors [F,K,3] f32; success [F];
    mode 'nearest' (K=1, generation.py:186-200) or 'within' (generation.py:245-255)."""
    anchors = _c(anchors.float())
    F, K = anchors.shape[0], anchors.shape[1]
    dev = anchors.device
    if count is None:
        count = torch.full((F,), K, dtype=I32, device=dev)
    count = _c(count.to(I32))
    success = _c(success.to(U8))
    if pts is not None:
        pts = _c(pts.float())
        B, N = pts.shape[0], pts.shape[1]
        nx, box, first = 0, 0.0, 0
    else:
        nx, box, first, N = lattice
    ids = torch.empty((B, N), dtype=U8, device=dev)
    check(_lib.load().vt_tactile_assign(dev_ptr(pts, "pts"), B, N, nx, box, first, dev_ptr(anchors, "anchors"),
                                        dev_ptr(count, "count", I32), dev_ptr(success, "success", U8), F, K,
                                        {"nearest": 0, "within": 1}[mode], float(radius), dev_ptr(ids, "ids", U8), stream_ptr()),
          "vt_tactile_assign")
    return ids


def decode_fwd_ids(grid, blob, ids, feats, pts=None, lattice=None, padding=0.1, out=None, precision="f32"):
    """vt_decode_fwd_ids: forward_img with c_img[b,n] = feats[ids[b,n]] (zeros where ids == 255)."""
    B, C, D, H, W = grid.shape
    keep, gptr = _cl_storage(grid)
    feats = _c(feats.float())
    if pts is not None:
        pts = _c(pts.float())
        N = pts.shape[1]
        nx, box, first = 0, 0.0, 0
    else:
        nx, box, first, N = lattice
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=grid.device)
    if precision == "f16f8":
        if pts is not None:
            raise VtError("decode_fwd_ids: precision 'f16f8' covers lattice slabs only (ops.f16f8_covers); use 'f16x3'")
        check(_lib.load().vt_decode_fwd_f16f8(gptr, B, D, C, N, nx, box, first, None, dev_ptr(_c(ids), "ids", U8), dev_ptr(feats, "feats"),
                                              feats.shape[0], dev_ptr(blob, "blob"), float(padding), dev_ptr(out, "out"), stream_ptr()),
              "vt_decode_fwd_f16f8")
        return out
    if precision in SPLIT_PRECISIONS:
        name = "vt_decode_fwd_" + precision
        check(getattr(_lib.load(), name)(gptr, B, D, C, dev_ptr(pts, "pts"), N, nx, box, first, None,
                                         dev_ptr(_c(ids), "ids", U8), dev_ptr(feats, "feats"), feats.shape[0],
                                         dev_ptr(blob, "blob"), float(padding), dev_ptr(out, "out"), None, stream_ptr()), name)
        return out
    check(_lib.load().vt_decode_fwd_ids(gptr, B, D, C, dev_ptr(pts, "pts"), N, nx, box, first, dev_ptr(_c(ids), "ids", U8),
                                        dev_ptr(feats, "feats"), feats.shape[0], dev_ptr(blob, "blob"), float(padding),
                                        dev_ptr(out, "out"), stream_ptr()), "vt_decode_fwd_ids")
    return out


# ---- the hand encoder's 2-D U-Net as one persistent launch (plane_unet.hip) --------------------------------------------

def plane_unet_params(net):
    """(PlaneUnetParams, tensors it points to) of an ``encoder.unet.UNet`` (depth, channel counts, nn.Conv2d / nn.ConvTranspose2d
    weights in their own layout)."""
    prm = _lib.PlaneUnetParams()
    prm.depth, prm.in_channels, prm.start_filts, prm.num_classes = net.depth, net.in_channels, net.start_filts, net.num_classes
    keep = []

    def ptr(t, name):
        t = t.detach()
        if not t.is_contiguous():
            t = t.contiguous()
        keep.append(t)
        return dev_ptr(t, name).value
    for l, d in enumerate(net.down_convs):
        for k, conv in enumerate((d.conv1, d.conv2)):
            prm.down_w[l][k], prm.down_b[l][k] = ptr(conv.weight, "down conv weight"), ptr(conv.bias, "down conv bias")
    for u, up in enumerate(net.up_convs):
        prm.up_tw[u], prm.up_tb[u] = ptr(up.upconv.weight, "upconv weight"), ptr(up.upconv.bias, "upconv bias")
        for k, conv in enumerate((up.conv1, up.conv2)):
            prm.up_w[u][k], prm.up_b[u][k] = ptr(conv.weight, "up conv weight"), ptr(conv.bias, "up conv bias")
    prm.final_w, prm.final_b = ptr(net.conv_final.weight, "conv_final.weight"), ptr(net.conv_final.bias, "conv_final.bias")
    return prm, keep


def plane_unet_supported(net, H, W):
    return bool(_lib.load().vt_plane_unet_supported(net.depth, net.in_channels, net.start_filts, net.num_classes, int(H), int(W)))


def plane_unet_pack(net):
    """The net's weights in fragment order + its biases: the blob vt_plane_unet_fwd reads (vt_plane_unet_pack)."""
    lib = _lib.load()
    n = lib.vt_plane_unet_blob_bytes(net.depth, net.in_channels, net.start_filts, net.num_classes)
    if n == 0:
        raise VtError("plane U-Net shape not built: depth 2..5, in_channels / start_filts / num_classes multiples of 32")
    prm, keep = plane_unet_params(net)
    blob = torch.empty(n // 4, dtype=torch.float32, device=keep[0].device)
    check(lib.vt_plane_unet_pack(ctypes.byref(prm), dev_ptr(blob, "blob"), n, stream_ptr()), "vt_plane_unet_pack")
    return blob


_plane_unet_ws = {}    # (device, dims, images, H, W) -> workspace (every phase's activations)


def plane_unet_workspace(net, n_img, H, W, fresh=False):
    """Workspace of vt_plane_unet_fwd (every phase's channels-last activations).  Inference calls share one per shape (torch's current
    stream orders them); ``fresh`` makes a new one (the training forward keeps it for the backward)."""
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, net.depth, net.in_channels, net.start_filts, net.num_classes, int(n_img), int(H), int(W))
    if not fresh and key in _plane_unet_ws:
        ws = _plane_unet_ws[key]
        keep_for_graph(ws)
        return ws
    n = lib.vt_plane_unet_workspace_bytes(net.depth, net.in_channels, net.start_filts, net.num_classes, int(n_img), int(H), int(W))
    if n == 0:
        raise VtError("plane U-Net shape not built (vt_plane_unet_supported)")
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    if not fresh:
        if len(_plane_unet_ws) >= 8:
            _plane_unet_ws.pop(next(iter(_plane_unet_ws)))
        _plane_unet_ws[key] = ws
    keep_for_graph(ws)
    return ws


def plane_unet_fwd(x, net, blob, ws=None):
    """UNet.forward on the HIP kernel: x [n_img, in_channels, H, W] -> [n_img, num_classes, H, W] (vt_plane_unet_fwd)."""
    x = _c(x)
    n_img, C, H, W = x.shape
    if C != net.in_channels:
        raise VtError(f"plane_unet_fwd: input has {C} channels, the net takes {net.in_channels}")
    if ws is None:
        ws = plane_unet_workspace(net, n_img, H, W)
    prm = _lib.PlaneUnetParams()
    prm.depth, prm.in_channels, prm.start_filts, prm.num_classes = net.depth, net.in_channels, net.start_filts, net.num_classes
    out = torch.empty((n_img, net.num_classes, H, W), dtype=torch.float32, device=x.device)
    keep_for_graph(blob)
    check(_lib.load().vt_plane_unet_fwd(dev_ptr(x, "x"), n_img, H, W, ctypes.byref(prm), dev_ptr(blob, "blob"),
                                        ctypes.c_void_p(ws.data_ptr()), ws.numel(), dev_ptr(out, "out"), stream_ptr()), "vt_plane_unet_fwd")
    return out


def plane_unet_bwd(x, net, blob, fwd_ws, dout):
    """Backward of plane_unet_fwd (vt_plane_unet_bwd): (dx, {parameter name: gradient}) from dout, the input, the packed weights and the
    workspace the forward filled.  Gradients are written, not accumulated."""
    lib = _lib.load()
    x, dout = _c(x), _c(dout)
    n_img, C, H, W = x.shape
    n = lib.vt_plane_unet_bwd_workspace_bytes(net.depth, net.in_channels, net.start_filts, net.num_classes, n_img, H, W)
    if n == 0:
        raise VtError("plane U-Net shape not built (vt_plane_unet_supported)")
    ws = torch.empty(n, dtype=torch.uint8, device=x.device)
    prm = _lib.PlaneUnetParams()
    prm.depth, prm.in_channels, prm.start_filts, prm.num_classes = net.depth, net.in_channels, net.start_filts, net.num_classes
    g = _lib.PlaneUnetGrads()
    grads = {}

    def buf(name, like):
        t = torch.empty(like.shape, dtype=torch.float32, device=x.device)
        grads[name] = t
        return dev_ptr(t, name).value
    for l, d in enumerate(net.down_convs):
        for k, cname in enumerate(("conv1", "conv2")):
            conv = getattr(d, cname)
            g.down_w[l][k], g.down_b[l][k] = buf(f"down_convs.{l}.{cname}.weight", conv.weight), buf(f"down_convs.{l}.{cname}.bias", conv.bias)
    for u, up in enumerate(net.up_convs):
        g.up_tw[u], g.up_tb[u] = buf(f"up_convs.{u}.upconv.weight", up.upconv.weight), buf(f"up_convs.{u}.upconv.bias", up.upconv.bias)
        for k, cname in enumerate(("conv1", "conv2")):
            conv = getattr(up, cname)
            g.up_w[u][k], g.up_b[u][k] = buf(f"up_convs.{u}.{cname}.weight", conv.weight), buf(f"up_convs.{u}.{cname}.bias", conv.bias)
    g.final_w, g.final_b = buf("conv_final.weight", net.conv_final.weight), buf("conv_final.bias", net.conv_final.bias)
    dx = torch.empty_like(x)
    check(lib.vt_plane_unet_bwd(dev_ptr(x, "x"), n_img, H, W, ctypes.byref(prm), dev_ptr(blob, "blob"), ctypes.c_void_p(fwd_ws.data_ptr()),
                                dev_ptr(dout, "dout"), ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.byref(g), dev_ptr(dx, "dx"),
                                stream_ptr()), "vt_plane_unet_bwd")
    return dx, grads

