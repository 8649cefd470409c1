"""GPU parity of vt_decode_fwd_wide -- LocalDecoder beyond the shipped 32 / 32 shape: hidden_size and c_dim multiples of 32 up
to 256, leaky heads (reference src/conv_onet/models/decoder.py:24-161) -- through the module the reference's users call:
against the reference's own outputs (tests/golden/g16_decode_wide.npz) and against the oracle on seeded shapes in between.
Exact-f32 arithmetic on v_mfma_f32_32x32x2_f32: the bar is 1e-5 on O(1) logits (summation order only)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _case(tag):
    a, sd = load_golden("g16_decode_wide.npz")
    arrs = {k[2:]: v for k, v in a.items() if k.startswith(tag + ".")}
    sdc = {k[2:]: v.float() for k, v in sd.items() if k.startswith(tag + ".")}
    hidden, c_dim, nb, leaky, nx, nearest = (int(x) for x in arrs["shape"])
    return arrs, sdc, hidden, c_dim, nb, bool(leaky), nx, "nearest" if nearest else "bilinear"


def _decoder(hidden, c_dim, nb, leaky, sd=None, seed=0, contact=True, mode="bilinear"):
    from vtaco_amd.conv_onet.models.decoder import LocalDecoder
    torch.manual_seed(seed)
    dec = LocalDecoder(dim=3, c_dim=c_dim, hidden_size=hidden, n_blocks=nb, leaky=leaky, padding=0.1, with_contact=contact,
                       sample_mode=mode)
    if sd is not None:
        dec.load_state_dict(sd)
    else:
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n, p in dec.named_parameters():
                p.add_(torch.randn(p.shape, generator=g) * (0.1 if n.endswith("fc_1.weight") else 0.03))
    return dec.to(DEV)


def _err(got, ref):
    return float((got.cpu() - torch.as_tensor(ref)).abs().max())


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_wide_decoder_against_the_reference_fixture(tag):
    a, sd, hidden, c_dim, nb, leaky, nx, mode = _case(tag)
    dec = _decoder(hidden, c_dim, nb, leaky, sd, mode=mode)
    assert dec._wide
    grid = T(a["grid"].astype(np.float32)).to(DEV)
    p, c_img = T(a["prand"]).to(DEV), T(a["c_img"].astype(np.float32)).to(DEV)
    with torch.no_grad():
        assert _err(dec(p, {"grid": grid}), a["logits"]) <= 1e-5
        assert _err(dec.forward_img(p, {"grid": grid}, c_img), a["logits_img"]) <= 1e-5
        o, oc = dec.forward_contact(p, {"grid": grid})
        assert _err(o, a["logits_contact"]) <= 1e-5 and _err(oc, a["logits_contact2"]) <= 1e-5
        lat = dec.decode_lattice(grid[:1], nx)
        assert _err(lat, a["logits_lattice"]) <= 1e-5
        # a slab of the lattice, and the same through explicit points
        half = dec.decode_lattice(grid[:1], nx, first=nx * nx * 3, count=nx * nx * 2)
        assert torch.equal(half, lat[:, nx * nx * 3: nx * nx * 5])


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("hidden,c_dim,nb,leaky,B,N,R", [(32, 32, 5, True, 2, 1000, 16), (96, 64, 2, False, 1, 33, 8),
                                                        (128, 256, 1, True, 3, 257, 8), (256, 32, 8, False, 1, 64, 4),
                                                        (64, 96, 3, False, 2, 1, 8)])
def test_wide_decoder_against_the_oracle(hidden, c_dim, nb, leaky, B, N, R, mode):
    """Shapes between the fixture's: a leaky 32 / 32 decoder (routed to the wide kernel), c_dim > hidden (fc_p_img's K = 3 + c_dim
    exceeds the hidden width), one and eight blocks, ragged point counts (one point; a tile and a bit)."""
    from oracle import vtaco_oracle as orc
    dec = _decoder(hidden, c_dim, nb, leaky, seed=hidden + c_dim + nb, mode=mode)
    kw = dict(leaky=leaky, sample_mode=mode)
    sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    g = torch.Generator().manual_seed(N)
    grid = torch.randn(B, c_dim, R, R, R, generator=g)
    p = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.3
    c_img = torch.randn(B, N, c_dim, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.4)
    with torch.no_grad():
        got = dec(p.to(DEV), {"grid": grid.to(DEV)})
        got_img = dec.forward_img(p.to(DEV), {"grid": grid.to(DEV)}, c_img.to(DEV))
        got_c, got_cc = dec.forward_contact(p.to(DEV), {"grid": grid.to(DEV)})
    ref = orc.local_decoder_forward(sd, p, grid, **kw)
    scale = max(1.0, float(ref.abs().max()))
    assert _err(got, ref) <= 1e-5 * scale
    assert _err(got_img, orc.local_decoder_forward_img(sd, p, grid, c_img, **kw)) <= 1e-5 * scale
    rc, rcc = orc.local_decoder_forward_contact(sd, p, grid, **kw)
    assert _err(got_c, rc) <= 1e-5 * scale and _err(got_cc, rcc) <= 1e-5 * scale


def test_wide_decoder_finger_ids_and_errors():
    from vtaco_amd._lib import VtError
    from vtaco_amd.conv_onet.models.decoder import AttentionDecoder, LocalDecoder
    dec = _decoder(64, 32, 2, True, seed=5)
    g = torch.Generator().manual_seed(6)
    grid = torch.randn(1, 32, 8, 8, 8, generator=g).to(DEV)
    nx = 8
    ids = torch.randint(0, 6, (1, nx ** 3), generator=g).to(torch.uint8)
    ids[ids == 5] = 255                                           # no finger
    feats = torch.randn(5, 32, generator=g)
    with torch.no_grad():
        got = dec.decode_lattice_ids(grid, nx, ids.to(DEV), feats.to(DEV))
        table = torch.cat([feats, torch.zeros(1, 32)])
        c_img = table[torch.where(ids == 255, torch.full_like(ids, 5), ids).long()]
        ref = dec.decode_lattice(grid, nx, c_img=c_img.to(DEV))
        # the split-f16 kernel reads the ids itself as well; a slab; two scenes with their own ids
        got_h = dec.decode_lattice_ids(grid, nx, ids.to(DEV), feats.to(DEV), precision="f16x3")
        ref_h = dec.decode_lattice(grid, nx, c_img=c_img.to(DEV), precision="f16x3")
        first, count = nx * nx * 2, nx * nx * 3
        got_s = dec.decode_lattice_ids(grid, nx, ids[:, first:first + count].contiguous().to(DEV), feats.to(DEV), first=first, count=count)
        grid2 = torch.cat([grid, grid.flip(2)])
        ids2 = torch.cat([ids, ids.flip(1)])
        got_2 = dec.decode_lattice_ids(grid2, nx, ids2.to(DEV), feats.to(DEV))
        ref_2 = dec.decode_lattice(grid2, nx, c_img=torch.cat([c_img, c_img.flip(1)]).to(DEV))
    assert torch.equal(got, ref) and torch.equal(got_h, ref_h) and not torch.equal(got_h, got)
    assert torch.equal(got_s, ref[:, first:first + count]) and torch.equal(got_2, ref_2)
    for bad in (dict(hidden_size=48, c_dim=32), dict(hidden_size=288, c_dim=32), dict(hidden_size=64, c_dim=16)):
        d = LocalDecoder(n_blocks=2, **bad).to(DEV)
        with torch.no_grad(), pytest.raises(VtError, match="multiples of 32 up to 256"):
            d(torch.zeros(1, 4, 3, device=DEV), {"grid": torch.zeros(1, bad["c_dim"], 4, 4, 4, device=DEV)})
    AttentionDecoder(c_dim=64, hidden_size=64)                     # built since round 5 (tests/test_fusion_gpu.py pins it on a reference fixture)
    with pytest.raises(VtError, match="c_dim"):
        AttentionDecoder(c_dim=160, hidden_size=64)


def _forbid_framework_ops(monkeypatch):
    """LocalDecoder must not reach F.grid_sample / F.linear on any path: make them raise for the duration of a test."""
    import torch.nn.functional as F

    def boom(*a, **k):
        raise AssertionError("a framework operator was reached from LocalDecoder")
    monkeypatch.setattr(F, "grid_sample", boom)
    monkeypatch.setattr(F, "linear", boom)


@pytest.mark.parametrize("hidden,c_dim,nb,leaky,mode,npts", [(96, 64, 2, True, "bilinear", 200), (256, 128, 3, False, "bilinear", 333),
                                                              (32, 160, 2, True, "bilinear", 65), (64, 32, 2, False, "nearest", 100)])
def test_wide_decoder_trains_on_hip_kernels(hidden, c_dim, nb, leaky, mode, npts, monkeypatch):
    """Shapes beyond 32 / 32 under autograd (the reference trains its class defaults 256 / 128 through torch autograd,
    decoder.py:24-51, 135-161): forward vt_decode_fwd_wide_train, backward vt_decode_bwd_wide + vt_rows_wgrad -- no framework
    operator (F.grid_sample / F.linear raise while the test runs).  Logits equal the inference kernel's; the gradients of the
    grid, of c_img and of every parameter equal the oracle's autograd (torch CPU) to 1e-4 of their scale: forward_img, forward
    and forward_contact."""
    from oracle import vtaco_oracle as orc
    _forbid_framework_ops(monkeypatch)
    dec = _decoder(hidden, c_dim, nb, leaky, seed=11, mode=mode)
    g = torch.Generator().manual_seed(12)
    grid = torch.randn(2, c_dim, 8, 8, 8, generator=g)
    p = (torch.rand(2, npts, 3, generator=g) - 0.5) * 1.2
    c_img = torch.randn(2, npts, c_dim, generator=g)
    sd0 = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}

    def check(tag, run_hip, run_ref, with_cimg):
        dec.zero_grad(set_to_none=True)
        gd = grid.to(DEV).requires_grad_(True)
        cd = c_img.to(DEV).requires_grad_(True) if with_cimg else None
        outs = run_hip(gd, cd)
        outs = outs if isinstance(outs, tuple) else (outs,)
        assert all(o.requires_grad for o in outs)
        loss = sum((k + 1) * o.square().sum() for k, o in enumerate(outs))
        loss.backward()
        with monkeypatch.context() as m:                              # the oracle is torch CPU: it may use its operators
            m.undo()
            sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
            gc = grid.clone().requires_grad_(True)
            cc = c_img.clone().requires_grad_(True) if with_cimg else None
            refs = run_ref(sd, gc, cc)
            refs = refs if isinstance(refs, tuple) else (refs,)
            sum((k + 1) * r.square().sum() for k, r in enumerate(refs)).backward()
        for o, r in zip(outs, refs):
            assert _err(o.detach(), r.detach()) <= 2e-5 * max(1.0, float(r.abs().max())), tag
        assert _err(gd.grad, gc.grad) <= 1e-4 * float(gc.grad.abs().max()), tag
        if with_cimg:
            assert _err(cd.grad, cc.grad) <= 1e-4 * float(cc.grad.abs().max()), tag
        seen = 0
        for name, prm in dec.named_parameters():
            if sd[name].grad is None:
                assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, (tag, name)
                continue
            seen += 1
            assert prm.grad is not None, (tag, name)
            assert _err(prm.grad, sd[name].grad) <= 1e-4 * max(1.0, float(sd[name].grad.abs().max())), (tag, name)
        assert seen >= 4 + 6 * nb
        return outs

    pd = p.to(DEV)
    out = check("forward_img", lambda gd, cd: dec.forward_img(pd, {"grid": gd}, cd),
                lambda sd, gc, cc: orc.local_decoder_forward_img(sd, p, gc, cc, leaky=leaky, sample_mode=mode), True)[0]
    with torch.no_grad():
        fast = dec.forward_img(pd, {"grid": grid.to(DEV)}, c_img.to(DEV))
    assert torch.equal(out.detach(), fast)                            # the training forward IS the inference kernel plus stores
    check("forward", lambda gd, cd: dec(pd, {"grid": gd}),
          lambda sd, gc, cc: orc.local_decoder_forward(sd, p, gc, leaky=leaky, sample_mode=mode), False)
    check("forward_contact", lambda gd, cd: dec.forward_contact(pd, {"grid": gd}),
          lambda sd, gc, cc: orc.local_decoder_forward_contact(sd, p, gc, leaky=leaky, sample_mode=mode), False)


# ---- the split-f16 form (vt_decode_fwd_wide_f16x3): LocalDecoder.precision = "f16x3" / decode_lattice(precision="f16x3") on the wide shapes.
# Half hi + lo operands carry 21-22 mantissa bits: the bar is f32 rounding over the layer widths, far inside north_star's 1e-4.
@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_wide_split_f16_against_the_reference_fixture(tag):
    a, sd, hidden, c_dim, nb, leaky, nx, mode = _case(tag)
    dec = _decoder(hidden, c_dim, nb, leaky, sd, mode=mode)
    dec.precision = "f16x3"
    grid = T(a["grid"].astype(np.float32)).to(DEV)
    p, c_img = T(a["prand"]).to(DEV), T(a["c_img"].astype(np.float32)).to(DEV)
    from vtaco_amd import ops
    ops.decode_range_status(reset=True)
    with torch.no_grad():
        exact = dec.decode_lattice(grid[:1], nx, precision="f32")
        assert _err(dec(p, {"grid": grid}), a["logits"]) <= 2e-5
        assert _err(dec.forward_img(p, {"grid": grid}, c_img), a["logits_img"]) <= 2e-5
        o, oc = dec.forward_contact(p, {"grid": grid})
        assert _err(o, a["logits_contact"]) <= 2e-5 and _err(oc, a["logits_contact2"]) <= 2e-5
        lat = dec.decode_lattice(grid[:1], nx)
        assert _err(lat, a["logits_lattice"]) <= 2e-5
        assert not torch.equal(lat, exact)                          # really the other kernel
        assert torch.equal(dec.decode_lattice(grid[:1], nx, precision="bf16x3"), exact)      # the range guard's way out: the exact kernel
        half = dec.decode_lattice(grid[:1], nx, first=nx * nx * 3, count=nx * nx * 2)
        assert _err(half, lat[:, nx * nx * 3: nx * nx * 5].cpu()) <= 1e-6
    assert ops.decode_range_status() == 0


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("hidden,c_dim,nb,leaky,B,N,R", [(32, 32, 5, True, 2, 1000, 16), (96, 64, 2, False, 1, 33, 8),
                                                        (128, 256, 1, True, 3, 257, 8), (256, 32, 8, False, 1, 64, 4),
                                                        (64, 96, 3, False, 2, 1, 8), (256, 128, 5, False, 1, 4097, 16)])
def test_wide_split_f16_against_the_oracle(hidden, c_dim, nb, leaky, B, N, R, mode):
    from oracle import vtaco_oracle as orc
    dec = _decoder(hidden, c_dim, nb, leaky, seed=hidden + c_dim + nb, mode=mode)
    dec.precision = "f16x3"
    kw = dict(leaky=leaky, sample_mode=mode)
    sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    g = torch.Generator().manual_seed(N)
    grid = torch.randn(B, c_dim, R, R, R, generator=g)
    p = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.3
    c_img = torch.randn(B, N, c_dim, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.4)
    with torch.no_grad():
        got = dec(p.to(DEV), {"grid": grid.to(DEV)})
        got_img = dec.forward_img(p.to(DEV), {"grid": grid.to(DEV)}, c_img.to(DEV))
        got_c, got_cc = dec.forward_contact(p.to(DEV), {"grid": grid.to(DEV)})
    ref = orc.local_decoder_forward(sd, p, grid, **kw)
    scale = max(1.0, float(ref.abs().max()))
    assert _err(got, ref) <= 2e-5 * scale
    assert _err(got_img, orc.local_decoder_forward_img(sd, p, grid, c_img, **kw)) <= 2e-5 * scale
    rc, rcc = orc.local_decoder_forward_contact(sd, p, grid, **kw)
    assert _err(got_c, rc) <= 2e-5 * scale and _err(got_cc, rcc) <= 2e-5 * scale


@pytest.mark.parametrize("nb,leaky,B,N,R,mode", [(5, False, 1, 70001, 16, "bilinear"), (5, True, 2, 1000, 8, "bilinear"), (1, False, 1, 1, 8, "bilinear"),
                                                 (2, True, 3, 31, 8, "nearest"), (3, False, 1, 33, 8, "bilinear"), (4, True, 1, 8191, 16, "nearest")])
def test_register_resident_pipeline_at_64_32_equals_the_streaming_kernel(nb, leaky, B, N, R, mode):
    """hidden 64 / c_dim 32 / n_blocks <= 5 (decode_wide_pipe.inc): every layer's weights stay in the registers of a pipeline of waves
    and the grid's samples come from a pre-pass in the caller's workspace (vt_decode_fwd_wide_f16x3_ws, what ops.decode_fwd calls).
    Against the streaming split-f16 kernel (vt_decode_fwd_wide_f16x3: same products, the heads summed in another order) and the
    exact-f32 kernel: ragged tile counts (fewer tiles than pipeline stages, one point), batches, both sample modes, the contact
    head, query points and lattice ranges; and the MLP on given features (vt_decode_mlp_fwd_wide_f16x3) takes the pipeline too."""
    import ctypes
    from vtaco_amd import _lib, ops
    lib = _lib.load()
    assert lib.vt_decode_wide_f16x3_workspace_bytes(B * N, 64, 32, nb, 0) == B * N * 32 * 4
    assert lib.vt_decode_wide_f16x3_workspace_bytes(B * N, 64, 32, nb, 1) == 0          # tactile input columns: the streaming kernel
    assert lib.vt_decode_wide_f16x3_workspace_bytes(B * N, 96, 32, nb, 0) == 0
    dec = _decoder(64, 32, nb, leaky, seed=7 + nb, mode=mode)
    g = torch.Generator().manual_seed(N + nb)
    grid = torch.randn(B, 32, R, R, R, generator=g).to(DEV)
    p = ((torch.rand(B, N, 3, generator=g) - 0.5) * 1.3).to(DEV)
    wide = (64, nb, leaky, mode == "nearest")
    flags = (1 if leaky else 0) | (2 if mode == "nearest" else 0)
    with torch.no_grad():
        blob = dec._blob(img=False, contact=True, precision="wide_f16x3")
        got, got2 = ops.decode_fwd(grid, blob, pts=p, padding=0.1, precision="wide_f16x3", wide=wide, want_contact=True)
        exact, exact2 = ops.decode_fwd(grid, dec._blob(img=False, contact=True, precision="wide"), pts=p, padding=0.1, precision="wide",
                                       wide=wide, want_contact=True)
        keep, gptr = ops._cl_storage(grid)
        old, old2 = torch.empty_like(got), torch.empty_like(got)
        ops.check(lib.vt_decode_fwd_wide_f16x3(gptr, B, R, 32, ops.dev_ptr(p, "pts"), N, 0, 0.0, 0, None, ops.dev_ptr(blob, "blob"), 64, nb, flags, 0.1,
                                               ops.dev_ptr(old, "out"), ops.dev_ptr(old2, "out2"), ops.stream_ptr()), "vt_decode_fwd_wide_f16x3")
        scale = max(1.0, float(exact.abs().max()), float(exact2.abs().max()))
        for a, b, c in ((got, old, exact), (got2, old2, exact2)):
            assert float((a - b).abs().max()) <= 2e-6 * scale and float((a - c).abs().max()) <= 5e-6 * scale, (nb, N, float((a - b).abs().max()), float((a - c).abs().max()))
        assert not torch.equal(got, old) or N < 4                      # really another kernel
        # a lattice range that starts inside a tile
        nx = 24
        first, count = 5 * nx + 3, min(nx ** 3 - 5 * nx - 3, 40000)
        lat = ops.decode_fwd(grid[:1], blob, lattice=(nx, 1.1, first, count), padding=0.1, precision="wide_f16x3", wide=wide)
        lat_exact = ops.decode_fwd(grid[:1], dec._blob(img=False, contact=True, precision="wide"), lattice=(nx, 1.1, first, count), padding=0.1,
                                   precision="wide", wide=wide)
        assert float((lat - lat_exact).abs().max()) <= 5e-6 * max(1.0, float(lat_exact.abs().max()))
        # the MLP behind a fuser: features given per point
        c = torch.randn(B, N, 32, generator=g).to(DEV)
        mlp = ops.decode_mlp_fwd(c, dec._blob(img=False, contact=False, precision="wide_f16x3"), p, precision="wide_f16x3", wide=(64, nb, leaky))
        mlp_exact = ops.decode_mlp_fwd(c, dec._blob(img=False, contact=False, precision="wide"), p, precision="wide", wide=(64, nb, leaky))
        assert float((mlp - mlp_exact).abs().max()) <= 5e-6 * max(1.0, float(mlp_exact.abs().max()))


def test_wide_split_f16_reports_activations_at_the_half_limit():
    """Hidden activations beyond 65504 saturate the hi halves: the kernel raises RANGE_HALF in the device's status word (the generator's
    guard then moves to the exact kernel), and stays silent on ordinary weights."""
    from vtaco_amd import ops
    dec = _decoder(64, 32, 2, False, seed=3)
    dec.precision = "f16x3"
    g = torch.Generator().manual_seed(4)
    grid = torch.randn(1, 32, 8, 8, 8, generator=g).to(DEV)
    p = ((torch.rand(1, 500, 3, generator=g) - 0.5)).to(DEV)
    ops.decode_range_status(reset=True)
    with torch.no_grad():
        dec(p, {"grid": grid})
        assert ops.decode_range_status(reset=True) == 0
        dec.fc_c[0].weight.mul_(3e5)
        dec(p, {"grid": grid})
    assert ops.decode_range_status(reset=True) & ops.RANGE_HALF


def test_wide_decoder_by_finger_id_at_the_reference_default_widths_needs_no_dense_tensor():
    """256 / 128 (the reference's class defaults, decoder.py:24) over a 64^3 slab by finger id: the dense c_img the round-4 path
    built with torch indexing would be B * count * c_dim floats (8.6 GB at 256^3); the kernels read ids and the [F, C] table.
    Equal to the dense form bit for bit, and the call allocates nothing of the dense tensor's size."""
    dec = _decoder(256, 128, 2, False, seed=7)
    g = torch.Generator().manual_seed(8)
    nx, first, count = 64, 64 * 64 * 16, 64 * 64 * 8
    grid = torch.randn(1, 128, 8, 8, 8, generator=g).to(DEV)
    ids = torch.randint(0, 6, (1, count), generator=g).to(torch.uint8)
    ids[ids == 5] = 255
    feats = torch.randn(5, 128, generator=g)
    with torch.no_grad():
        idd, fd = ids.to(DEV), feats.to(DEV)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        got = dec.decode_lattice_ids(grid, nx, idd, fd, first=first, count=count, precision="f16x3")
        torch.cuda.synchronize()
        extra = torch.cuda.max_memory_allocated() - base
        dense_bytes = count * 128 * 4
        assert extra < dense_bytes // 4, (extra, dense_bytes)       # logits + blobs only
        table = torch.cat([feats, torch.zeros(1, 128)])
        c_img = table[torch.where(ids == 255, torch.full_like(ids, 5), ids).long()].to(DEV)
        ref = dec.decode_lattice(grid, nx, first=first, count=count, c_img=c_img, precision="f16x3")
    assert torch.equal(got, ref)


def test_empty_query_sets_under_autograd():
    """N = 0 in the training paths (shipped shape and wide): empty logits, zero gradients, no 'shape not built'."""
    from vtaco_amd.conv_onet.models.decoder import LocalDecoder
    for kw in (dict(hidden_size=32, c_dim=32, n_blocks=5), dict(hidden_size=64, c_dim=32, n_blocks=2)):
        torch.manual_seed(0)
        dec = LocalDecoder(dim=3, padding=0.1, with_contact=True, **kw).to(DEV)
        grid = torch.randn(2, 32, 8, 8, 8, device=DEV, requires_grad=True)
        p = torch.zeros(2, 0, 3, device=DEV)
        out = dec(p, {"grid": grid})
        assert tuple(out.shape) == (2, 0)
        oi = dec.forward_img(p, {"grid": grid}, torch.zeros(2, 0, 32, device=DEV, requires_grad=True))
        o1, o2 = dec.forward_contact(p, {"grid": grid})
        (out.sum() + oi.sum() + o1.sum() + o2.sum()).backward()
        assert grid.grad is not None and float(grid.grad.abs().max()) == 0.0
        for prm in dec.parameters():
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0
    from vtaco_amd import ops
    gg = ops.sample_grid_bwd((1, 32, 8, 8, 8), torch.zeros(1, 0, 3, device=DEV), torch.zeros(1, 0, 32, device=DEV))
    assert float(gg.abs().max()) == 0.0


def test_wide_split_f16_tile_fits_the_lds_at_every_width():
    """Regression (found by tests/stress_gpu.py): narrow hidden layers let spare waves take further point-group pairs, and the tile's
    activation planes grow with them -- at hidden 32 / c_dim 256 four pairs would need 270 KB of LDS.  The launcher caps the pairs by
    what fits; every corner of the (hidden, c_dim) range must launch and agree with the exact kernel."""
    g = torch.Generator().manual_seed(31)
    for hidden, c_dim in ((32, 256), (32, 32), (64, 256), (96, 160), (128, 256), (256, 32)):
        dec = _decoder(hidden, c_dim, 1, True, seed=hidden + c_dim)
        grid = torch.randn(1, c_dim, 4, 4, 4, generator=g).to(DEV)
        p = ((torch.rand(1, 300, 3, generator=g) - 0.5) * 1.1).to(DEV)
        c_img = torch.randn(1, 300, c_dim, generator=g).to(DEV)
        with torch.no_grad():
            dec.precision = "f32"
            ref = dec.forward_img(p, {"grid": grid}, c_img)
            dec.precision = "f16x3"
            got = dec.forward_img(p, {"grid": grid}, c_img)
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (hidden, c_dim)
