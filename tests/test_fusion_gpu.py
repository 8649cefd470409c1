"""GPU parity of TransformerFusion / AttentionDecoder.forward_img against the
reference-generated golden g5 (N=256 and N=2048) and the oracle on ragged N."""
import pytest
import torch

from conftest import load_golden, sub_sd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy


def _adec(sd):
    from vtaco_amd.conv_onet.models import decoder_dict
    dec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32)
    missing = dec.load_state_dict(sd, strict=True)
    return dec.to(DEV).eval()


def test_state_dict_keys_match_reference():
    _, sd = load_golden("g5_fusion.npz")
    from vtaco_amd.conv_onet.models import decoder_dict
    dec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32)
    mine = {k for k in dec.state_dict() if "num_batches_tracked" not in k}
    assert mine == set(sd.keys())


@pytest.mark.parametrize("n", [256, 2048])
def test_fusion_vs_golden(n):
    a, sd = load_golden("g5_fusion.npz")
    dec = _adec(sd)
    with torch.no_grad():
        out = dec.fuser(T(a[f"c_img{n}"]).to(DEV), 1, T(a[f"c{n}"]).to(DEV), 1)
    assert float((out.cpu() - T(a[f"fused{n}"])).abs().max()) <= 5e-5


def test_attention_decoder_forward_img_vs_golden():
    a, sd = load_golden("g5_fusion.npz")
    dec = _adec(sd)
    with torch.no_grad():
        lo = dec.forward_img(T(a["p"]).to(DEV), {"grid": T(a["grid"]).to(DEV)}, T(a["c_img256"]).to(DEV))
    assert float((lo.cpu() - T(a["logits"])).abs().max()) <= 1e-4


@pytest.mark.parametrize("B,N", [(1, 33), (3, 100), (2, 1000)])
def test_fusion_ragged_sizes_vs_oracle(B, N):
    from oracle import vtaco_oracle as orc
    _, sd = load_golden("g5_fusion.npz")
    dec = _adec(sd)
    g = torch.Generator().manual_seed(N)
    ci = torch.randn(B, N, 32, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.3)
    cc = torch.randn(B, N, 32, generator=g)
    ref = orc.transformer_fusion(sub_sd(sd, "fuser."), ci, cc)
    with torch.no_grad():
        out = dec.fuser(ci.to(DEV), 1, cc.to(DEV), 1)
    assert float((out.cpu() - ref).abs().max()) <= 5e-5


def test_sample_grid_and_mlp_split_equals_fused_decode():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    from vtaco_amd.conv_onet.models import decoder_dict
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, with_contact=True)
    dec.load_state_dict(sd, strict=True)
    dec.to(DEV)
    grid, p = T(a["grid2"]).to(DEV), T(a["prand"]).to(DEV)
    with torch.no_grad():
        feat = ops.sample_grid(grid, p)
        assert float((feat.cpu().transpose(1, 2) - T(a["feat_rand"])).abs().max()) <= 2e-6
        split = ops.decode_mlp_fwd(feat, dec._blob(), p)
        fused = dec(p, {"grid": grid})
    assert float((split - fused).abs().max()) <= 1e-6


def test_attention_decoder_forward_img_backward_vs_oracle_autograd():
    """AttentionDecoder.forward_img under autograd, every stage HIP forward and backward: sampling (vt_sample_grid / _bwd),
    the fuser (vt_fusion_fwd_train / vt_fusion_bwd) and the conditioned MLP (vt_decode_mlp_fwd_train / vt_decode_mlp_bwd /
    vt_decode_wgrad), against torch-CPU autograd of the oracle: logits, d grid, d c_img and every parameter gradient, 1e-4
    relative."""
    from oracle import vtaco_oracle as orc
    a, sd = load_golden("g5_fusion.npz")
    dec = _adec(sd)                                   # eval mode: no dropout, as the oracle
    p, grid, c_img = T(a["p"]), T(a["grid"]), T(a["c_img256"])
    wgt = torch.randn(p.shape[:2], generator=torch.Generator().manual_seed(2))
    sdr = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    gr, cr = grid.clone().requires_grad_(), c_img.clone().requires_grad_()
    ref = orc.attention_decoder_forward_img(sdr, p, gr, cr)
    (ref * wgt).sum().backward()
    gh, ch = grid.to(DEV).requires_grad_(), c_img.to(DEV).requires_grad_()
    out = dec.forward_img(p.to(DEV), {"grid": gh}, ch)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 1e-4
    (out * wgt.to(DEV)).sum().backward()
    rel = lambda x, y: float((x - y).abs().max()) / max(float(y.abs().max()), 1e-12)
    assert rel(gh.grad.cpu(), gr.grad) <= 1e-4
    assert rel(ch.grad.cpu(), cr.grad) <= 1e-4
    gscale = max(float(v.grad.abs().max()) for v in sdr.values() if v.grad is not None)
    checked = 0
    for name, prm in dec.named_parameters():
        want = sdr[name].grad
        twin = name.replace("fuser.encoder.layers.0.self_attn", "fuser.decoder.layers.0.self_attn")
        if twin != name and sdr[twin].grad is not None:      # ONE module under two state_dict names: its gradient is the sum
            want = want + sdr[twin].grad if want is not None else sdr[twin].grad
        if want is None:                                   # fc_p_img, after_norm: unused by this forward in the reference too
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0
            continue
        # (a LayerNorm bias in front of an InstanceNorm over the points has an exactly-zero gradient: rounding noise only)
        # (such exactly-zero gradients are f32 rounding noise of sums whose terms have the size of the other gradients:
        # the floor is 1e-6 of the largest gradient)
        assert float((prm.grad.cpu() - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1e-2 * gscale), name
        checked += 1
    assert checked >= 40


def test_attention_decoder_trains_with_dropout():
    """Train mode (dropout in TransNonlinear active): one Adam step lowers nothing in particular, but every
    used parameter receives a finite gradient and the eval-mode HIP forward still runs afterwards."""
    a, sd = load_golden("g5_fusion.npz")
    dec = _adec(sd).train()
    p, grid, c_img = T(a["p"]).to(DEV), T(a["grid"]).to(DEV).requires_grad_(), T(a["c_img256"]).to(DEV)
    opt = torch.optim.Adam(dec.parameters(), lr=1e-4)
    loss = torch.nn.functional.l1_loss(dec.forward_img(p, {"grid": grid}, c_img), torch.rand(p.shape[:2], device=DEV))
    loss.backward()
    assert torch.isfinite(grid.grad).all() and float(grid.grad.abs().max()) > 0
    assert all(torch.isfinite(q.grad).all() for q in dec.parameters() if q.grad is not None)
    opt.step()
    dec.eval()
    with torch.no_grad():
        assert torch.isfinite(dec.forward_img(p, {"grid": grid.detach()}, c_img)).all()


def _fusion_case(B, N, seed, C=32):
    """A seeded fuser + inputs; the mirror module on the device and its state_dict for the oracle."""
    from vtaco_amd.transformer_fusion import TransformerFusion
    torch.manual_seed(seed)
    fuser = TransformerFusion(d_model=C, key_feature_dim=64, with_pos_embed=False)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, prm in fuser.named_parameters():
            if "norm2" in name or name.endswith(".bias"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.2)
    c_img = torch.randn(B, N, C, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.3)
    c = torch.randn(B, N, C, generator=g)
    wgt = torch.randn(B, N, C, generator=g)
    return fuser, c_img, c, wgt


def _check_fusion_grads(fuser, sdr, c_img_dev, c_dev, cr, cc, tol):
    rel = lambda x, y: float((x - y).abs().max()) / max(float(y.abs().max()), 1e-12)
    assert rel(c_img_dev.grad.cpu(), cr.grad) <= tol, ("d c_img", rel(c_img_dev.grad.cpu(), cr.grad))
    assert rel(c_dev.grad.cpu(), cc.grad) <= tol, ("d c", rel(c_dev.grad.cpu(), cc.grad))
    gscale = max(float(v.grad.abs().max()) for v in sdr.values() if v.grad is not None)
    checked = 0
    for name, prm in fuser.named_parameters():
        if "after_norm" in name:                              # never called, by the reference either
            continue
        want = sdr[name].grad
        twin = name.replace("encoder.layers.0.self_attn", "decoder.layers.0.self_attn")
        if twin != name and sdr[twin].grad is not None:       # ONE module under two names: the sum of both uses
            want = want + sdr[twin].grad if want is not None else sdr[twin].grad
        assert want is not None, name
        err = float((prm.grad.cpu() - want).abs().max())
        assert err <= tol * max(float(want.abs().max()), 1e-2 * gscale), (name, err, float(want.abs().max()))
        checked += 1
    assert checked == 20


@pytest.mark.parametrize("B,N", [(1, 256), (2, 300), (1, 2048), (3, 77)])
def test_fusion_backward_vs_oracle_autograd(B, N):
    """vt_fusion_fwd_train / vt_fusion_bwd (eval mode: no dropout) against torch-CPU autograd of the oracle: the fused features,
    d c_img, d c and all twenty parameter gradients (the shared self-attention's = the sum of its two uses), 1e-4 relative;
    N a multiple of the 32-row tiles, ragged, and the reference's chunk size."""
    from oracle import vtaco_oracle as orc
    fuser, c_img, c, wgt = _fusion_case(B, N, 40 + N)
    sdr = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in fuser.state_dict().items()}
    cr, cc = c_img.clone().requires_grad_(), c.clone().requires_grad_()
    ref = orc.transformer_fusion(sdr, cr, cc)
    (ref * wgt).sum().backward()
    fuser = fuser.to(DEV).eval()
    ch, gh = c_img.to(DEV).requires_grad_(), c.to(DEV).requires_grad_()
    out = fuser(ch, 1, gh, 1)
    assert out.requires_grad
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 5e-5
    (out * wgt.to(DEV)).sum().backward()
    _check_fusion_grads(fuser, sdr, ch, gh, cr, cc, 1e-4)
    # bit-reproducible: a second backward gives the same numbers
    first = {n: p.grad.clone() for n, p in fuser.named_parameters() if p.grad is not None}
    fuser.zero_grad()
    ch2, gh2 = c_img.to(DEV).requires_grad_(), c.to(DEV).requires_grad_()
    (fuser(ch2, 1, gh2, 1) * wgt.to(DEV)).sum().backward()
    assert torch.equal(ch2.grad, ch.grad) and all(torch.equal(p.grad, first[n]) for n, p in fuser.named_parameters() if p.grad is not None)


def test_fusion_train_mode_dropout_replayed_against_the_oracle():
    """Train mode: TransNonlinear's two dropouts (p = 0.1) are applied from a seed; the masks the kernels used are materialised
    (vt_fusion_dropout_mask) and handed to the oracle, whose output and autograd gradients the HIP path must then match.
    Also: ~10 % of the factors are 0 and the rest 1/0.9; another seed gives another mask; the same seed the same output."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    B, N = 2, 200
    fuser, c_img, c, wgt = _fusion_case(B, N, 7)
    sdr = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in fuser.state_dict().items()}
    fuser = fuser.to(DEV).train()
    ch, gh = c_img.to(DEV).requires_grad_(), c.to(DEV).requires_grad_()
    out = fuser.forward_train(ch, gh, seed=12345)
    masks = [tuple(ops.fusion_dropout_mask(fuser.p_drop, 12345, call, which, B * N, DEV).cpu().reshape(B, N, -1) for which in (0, 1))
             for call in range(3)]
    m = torch.cat([x.reshape(-1) for pair in masks for x in pair])
    zero_frac = float((m == 0).float().mean())
    assert 0.08 <= zero_frac <= 0.12 and bool(((m == 0) | ((m - 1 / 0.9).abs() < 1e-6)).all())
    assert not torch.equal(masks[0][0], masks[1][0])
    cr, cc = c_img.clone().requires_grad_(), c.clone().requires_grad_()
    ref = orc.transformer_fusion(sdr, cr, cc, masks=masks)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 5e-5
    (ref * wgt).sum().backward()
    (out * wgt.to(DEV)).sum().backward()
    _check_fusion_grads(fuser, sdr, ch, gh, cr, cc, 1e-4)
    with torch.no_grad():
        again = fuser.forward_train(c_img.to(DEV), c.to(DEV), seed=12345) if False else ops.fusion_fwd_train(
            c_img.to(DEV), c.to(DEV), fuser.decoder.layers[0].self_attn.unit_tensors(), fuser.decoder.layers[0].cross_attn.unit_tensors(),
            fuser.p_drop, 12345)[0]
        other = ops.fusion_fwd_train(c_img.to(DEV), c.to(DEV), fuser.decoder.layers[0].self_attn.unit_tensors(),
                                     fuser.decoder.layers[0].cross_attn.unit_tensors(), fuser.p_drop, 999)[0]
    assert torch.equal(again, out.detach()) and not torch.equal(other, out.detach())


def test_fusion_by_finger_id_equals_the_gathered_tensor_bit_for_bit():
    """vt_fusion_fwd_ids (the tactile rows of the decoder's self-attention by finger id + table, what generation.py:159-255 gathers on
    the host) against vt_fusion_fwd on the gathered [B, N, C] tensor: identical bits -- whole chunks (fp8-corrected tiles, N >= 512), a
    short ragged chunk (half-pair tiles), and chunks picked out of a larger id array through chunk_index."""
    from vtaco_amd import ops
    from vtaco_amd.transformer_fusion import TransformerFusion
    torch.manual_seed(3)
    fuser = TransformerFusion(use_xyz=True, input_size=2048, d_model=32, num_layers=1, key_feature_dim=64, with_pos_embed=False,
                              encoder_pos_embed_input_dim=3, decoder_pos_embed_input_dim=3).to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(5, 32, generator=g).to(DEV)
    table = torch.cat([feats, torch.zeros(1, 32, device=DEV)])
    for rows, N in ((6, 1024), (1, 77)):
        ids = torch.randint(0, 6, (rows, N), generator=g).to(torch.uint8)
        ids[ids == 5] = 255
        ids[0] = 255                                                   # a chunk no finger touches
        c = torch.randn(rows, N, 32, generator=g).to(DEV)
        idd = ids.to(DEV)
        gathered = table[torch.where(idd == 255, torch.full_like(idd, 5), idd).long()]
        with torch.no_grad():
            ref = fuser(gathered, 1, c, 1)
            got = fuser.forward_ids(idd, feats, c)
            assert torch.equal(got, ref)
            assert float(got[0].abs().max()) == 0.0                    # fuse(0, c) = 0 exactly (the generator's shortcut relies on it)
            if rows > 2:
                pick = torch.tensor([4, 1, 3], dtype=torch.int32, device=DEV)
                got_p = fuser.forward_ids(idd, feats, c[pick.long()], chunk_index=pick)
                assert torch.equal(got_p, ref[pick.long()])
    fuser.train()
    with pytest.raises(Exception, match="eval-mode"):
        fuser.forward_ids(idd, feats, c)


@pytest.mark.parametrize("tag", ["D", "E"])
def test_attention_decoder_beyond_the_shipped_widths_against_the_reference_fixture(tag):
    """AttentionDecoder at c_dim 128 / hidden_size 256 / 5 blocks (the reference's class defaults, decoder.py:176-207; one chunk of
    512 points) and at 64 / 64 / 2 on two ragged chunks of 300 points: the fuser's output and the logits of forward_img against the
    REAL reference's (tests/golden/g17_attention_wide.npz, make_attn_wide_goldens.py).  vt_fusion_fwd at d_model 64 / 128 (generic-width
    projection / epilogue kernels around the N x N passes), vt_sample_grid at c_dim > 32, vt_decode_mlp_fwd_wide[_f16x3]."""
    from conftest import load_golden
    from oracle import vtaco_oracle as orc
    from vtaco_amd.conv_onet.models.decoder import AttentionDecoder
    a, sd = load_golden("g17_attention_wide.npz")
    arrs = {k[2:]: v for k, v in a.items() if k.startswith(tag + ".")}
    sdc = {k[2:]: v.float() for k, v in sd.items() if k.startswith(tag + ".")}
    c_dim, hidden, nb, B, N = (int(x) for x in arrs["shape"])
    dec = AttentionDecoder(dim=3, c_dim=c_dim, hidden_size=hidden, n_blocks=nb, padding=0.1)
    dec.load_state_dict(sdc)
    dec = dec.to(DEV).eval()
    T = torch.from_numpy
    grid, p, c_img = T(arrs["grid"].astype("float32")).to(DEV), T(arrs["p"]).to(DEV), T(arrs["c_img"].astype("float32")).to(DEV)
    with torch.no_grad():
        from vtaco_amd import ops
        c = ops.sample_grid(grid, p, 0.1)
        assert float((c.cpu() - T(arrs["c"])).abs().max()) <= 2e-6
        fused = dec.fuser(c_img, 1, c, 1)
        ref_f = T(arrs["fused"])
        assert float((fused.cpu() - ref_f).abs().max()) <= 5e-5 * max(1.0, float(ref_f.abs().max()))
        for prec in ("f16x3", "f32"):
            dec.mlp_precision = prec
            logits = dec.forward_img(p, {"grid": grid}, c_img)
            ref = T(arrs["logits_img"])
            assert float((logits.cpu() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max())), prec
    # the oracle (width-generic restatement) agrees with the fixture as well: it is the checker of the shapes in between
    o = orc.attention_decoder_forward_img(sdc, T(arrs["p"]), T(arrs["grid"].astype("float32")), T(arrs["c_img"].astype("float32")))
    assert float((o - T(arrs["logits_img"])).abs().max()) <= 2e-5


@pytest.mark.parametrize("tag", ["D", "E"])
def test_attention_decoder_beyond_the_shipped_widths_trains_like_the_reference(tag):
    """The same two decoders under autograd against the REAL reference's own gradients (g20, make_attn_wide_goldens.py): d grid,
    d c_img and every parameter's gradient (64 sampled entries + its sum) of L = sum(logits * w) through vt_sample_grid[_bwd] at
    c_dim 64 / 128, vt_fusion_fwd_train / vt_fusion_bwd at d_model 64 / 128 and vt_decode_mlp_fwd_wide_train / vt_decode_mlp_bwd_wide /
    vt_rows_wgrad; the framework's linear / grid_sample operators made to raise (nothing falls back)."""
    import os
    import numpy as np
    import torch.nn.functional as F
    from conftest import GOLDEN, load_golden
    from test_oracle_golden import _g20_check_param_grads
    from vtaco_amd.conv_onet.models.decoder import AttentionDecoder
    a, sd = load_golden("g17_attention_wide.npz")
    z = dict(np.load(os.path.join(GOLDEN, "g20_attention_wide_grads.npz")))
    arrs = {k[2:]: v for k, v in a.items() if k.startswith(tag + ".")}
    sdc = {k[2:]: v.float() for k, v in sd.items() if k.startswith(tag + ".")}
    c_dim, hidden, nb, B, N = (int(x) for x in arrs["shape"])
    dec = AttentionDecoder(dim=3, c_dim=c_dim, hidden_size=hidden, n_blocks=nb, padding=0.1)
    dec.load_state_dict(sdc)
    dec = dec.to(DEV).eval()
    T = torch.from_numpy
    grid = T(arrs["grid"].astype("float32")).to(DEV).requires_grad_(True)
    c_img = T(arrs["c_img"].astype("float32")).to(DEV).requires_grad_(True)
    p, w = T(arrs["p"]).to(DEV), T(z[f"{tag}.w"]).to(DEV)
    saved = F.linear, F.grid_sample

    def boom(*a, **k):
        raise AssertionError("a framework operator ran under the HIP decoder")
    F.linear = F.grid_sample = boom
    try:
        out = dec.forward_img(p, {"grid": grid}, c_img)
        (out * w).sum().backward()
    finally:
        F.linear, F.grid_sample = saved
    ref = T(arrs["logits_img"])
    assert float((out.detach().cpu() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max()))
    # The reference ran on the CPU: a ReLU of the MLP or of the fuser at ~0 may take the other branch on a point here, which moves
    # that point's row of d c_img (seen in the fuser's own test: one row in ~2000), and d grid and every parameter's gradient a little.  At most two rows beyond the tolerance; the parameters to 2e-4, or 2e-2 when a row did flip.
    rows_bad = 0
    for got, key in ((c_img.grad, "d_c_img"), (grid.grad, "d_grid")):
        refg = T(z[f"{tag}.{key}"])
        err = (got.cpu() - refg).abs()
        tol = 1e-4 * float(refg.abs().max())
        if key == "d_c_img":
            rows_bad = int((err.reshape(B * N, -1).max(dim=1).values > tol).sum())
            assert rows_bad <= 2, (tag, key, rows_bad, float(err.max()), tol)
        else:
            # (a flipped row reaches every corner of the grid a little: its key / value rows enter every point's attention)
            assert rows_bad > 0 or float(err.max()) <= tol, (tag, key, float(err.max()), tol)
        assert float((got.cpu() - refg).norm()) <= (1e-4 if rows_bad == 0 else 2e-2) * float(refg.norm()), (tag, key)
    _g20_check_param_grads(z, tag, {n: prm.grad for n, prm in dec.named_parameters()}, 2e-4 if rows_bad == 0 else 2e-2)
    # a second backward accumulates through autograd as usual
    g1 = {n: prm.grad.clone() for n, prm in dec.named_parameters() if prm.grad is not None}
    (dec.forward_img(p, {"grid": grid}, c_img) * w).sum().backward()
    for n, prm in dec.named_parameters():
        if prm.grad is not None:
            assert float((prm.grad - 2 * g1[n]).abs().max()) <= 1e-5 * max(1e-6, float(g1[n].abs().max())), n


@pytest.mark.parametrize("c_dim,hidden,nb,B,N,leaky", [(32, 64, 3, 2, 200, False), (96, 128, 3, 1, 256, True), (128, 32, 1, 2, 65, False)])
def test_attention_decoder_wide_backward_vs_oracle_autograd(c_dim, hidden, nb, B, N, leaky):
    """The widths between the fixtures (a 32-wide fuser in front of a 64-wide MLP; 96 / 128 with leaky_relu in front of the head; a
    128-wide fuser in front of a 32-wide one-block MLP on ragged chunks) against torch-CPU autograd through the oracle, eval mode;
    then train mode: the dropout of the fuser is drawn from a seed, so the same seed gives the same bits and another seed other ones."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd.conv_onet.models.decoder import AttentionDecoder
    torch.manual_seed(40 + c_dim + hidden)
    dec = AttentionDecoder(dim=3, c_dim=c_dim, hidden_size=hidden, n_blocks=nb, padding=0.1, leaky=leaky).eval()
    g = torch.Generator().manual_seed(41)
    with torch.no_grad():
        for n, prm in dec.named_parameters():
            if n.endswith(".bias") or n.endswith("norm2.weight"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
    R = 6
    grid = torch.randn(B, c_dim, R, R, R, generator=g)
    p = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.2
    c_img = torch.randn(B, N, c_dim, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.3)
    w = torch.randn(B, N, generator=g)
    sdr = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in dec.state_dict().items()}
    gr, cr = grid.clone().requires_grad_(), c_img.clone().requires_grad_()
    ref = orc.attention_decoder_forward_img(sdr, p, gr, cr, leaky=leaky)
    (ref * w).sum().backward()
    dec = dec.to(DEV)
    gd, cd = grid.to(DEV).requires_grad_(), c_img.to(DEV).requires_grad_()
    out = dec.forward_img(p.to(DEV), {"grid": gd}, cd)
    (out * w.to(DEV)).sum().backward()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 5e-5 * max(1.0, float(ref.detach().abs().max()))
    rows_bad = 0
    for got, refg, key in ((cd.grad, cr.grad, "d c_img"), (gd.grad, gr.grad, "d grid")):
        err = (got.cpu() - refg).abs()
        tol = 1e-4 * float(refg.abs().max())
        if key == "d c_img":
            rows_bad = int((err.reshape(B * N, -1).max(dim=1).values > tol).sum())
            assert rows_bad <= 2, (key, rows_bad, float(err.max()), tol)
        assert float((got.cpu() - refg).norm()) <= (1e-4 if rows_bad == 0 else 2e-2) * float(refg.norm()), (key, float(err.max()), tol)
    ptol = 2e-4 if rows_bad == 0 else 2e-2
    shared = "fuser.decoder.layers.0.self_attn."
    for n, prm in dec.named_parameters():
        refp = sdr[n].grad
        if n.startswith("fuser.encoder.layers.0.self_attn."):       # the shared unit: both uses
            other = sdr[n.replace("fuser.encoder.", "fuser.decoder.")].grad
            refp = refp + other if other is not None else refp
        if refp is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, n
            continue
        if n.endswith("norm2.bias"):                                # zero in exact arithmetic (InstanceNorm follows): noise on both sides
            continue
        scale = max(1e-6, float(refp.abs().max()))
        assert float((prm.grad.cpu() - refp).abs().max()) <= ptol * scale, (n, float((prm.grad.cpu() - refp).abs().max()), scale)
    assert not any(n.startswith(shared) for n, _ in dec.named_parameters())
    # train mode
    dec.train()
    outs = []
    for seed in (7, 7, 8):
        torch.manual_seed(seed)
        for prm in dec.parameters():
            prm.grad = None
        o = dec.forward_img(p.to(DEV), {"grid": gd}, cd)
        (o * w.to(DEV)).sum().backward()
        outs.append((o.detach().clone(), dec.fc_p.weight.grad.clone()))
        assert bool(torch.isfinite(o).all()) and all(bool(torch.isfinite(q.grad).all()) for q in dec.parameters() if q.grad is not None)
    assert torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[0][0], outs[2][0])
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-5 * float(outs[0][1].abs().max())      # (atomics in the grid scatter only)


def test_wide_fusion_against_the_oracle_on_a_whole_chunk():
    """d_model 96 (three 32-channel slices, three waves in the epilogue) on a chunk of 2048 points and a short one, against the oracle."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd.transformer_fusion import TransformerFusion
    torch.manual_seed(5)
    for N, B in ((2048, 1), (50, 2)):
        fuser = TransformerFusion(use_xyz=True, input_size=2048, d_model=96, num_layers=1, key_feature_dim=64, with_pos_embed=False,
                                  encoder_pos_embed_input_dim=3, decoder_pos_embed_input_dim=3).eval()
        g = torch.Generator().manual_seed(6)
        with torch.no_grad():
            for n, prm in fuser.named_parameters():
                if n.endswith("norm2.weight") or n.endswith("norm2.bias") or n.endswith(".bias"):
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
        c = torch.randn(B, N, 96, generator=g)
        c_img = torch.randn(B, N, 96, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.3)
        ref = orc.transformer_fusion({k: v.detach() for k, v in fuser.state_dict().items()}, c_img, c)
        fd = fuser.to(DEV)
        with torch.no_grad():
            got = fd(c_img.to(DEV), 1, c.to(DEV), 1)
        assert float((got.cpu() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max())), (N, B)


@pytest.mark.parametrize("C,B,N", [(64, 2, 300), (96, 1, 256), (128, 1, 512), (128, 3, 77)])
def test_wide_fusion_backward_vs_oracle_autograd(C, B, N):
    """vt_fusion_fwd_train / vt_fusion_bwd at d_model 64 / 96 / 128 (the reference's AttentionDecoder default is 128) against torch-CPU
    autograd of the oracle: fused features, d c_img, d c and all twenty parameter gradients; eval mode, then train mode with the
    dropout masks the kernels used (vt_fusion_dropout_mask_wide) handed to the oracle."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    for train in (False, True):
        fuser, c_img, c, wgt = _fusion_case(B, N, 300 + C + N, C)
        sdr = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in fuser.state_dict().items()}
        cr, cc = c_img.clone().requires_grad_(), c.clone().requires_grad_()
        fuser = fuser.to(DEV).train(train)
        ch, gh = c_img.to(DEV).requires_grad_(), c.to(DEV).requires_grad_()
        masks = None
        if train:
            out = fuser.forward_train(ch, gh, seed=4242)
            masks = [tuple(ops.fusion_dropout_mask(fuser.p_drop, 4242, call, which, B * N, DEV, d_model=C).cpu().reshape(B, N, -1)
                           for which in (0, 1)) for call in range(3)]
            m = torch.cat([x.reshape(-1) for pair in masks for x in pair])
            assert 0.08 <= float((m == 0).float().mean()) <= 0.12
            assert masks[0][1].shape[-1] == C and not torch.equal(masks[0][1][..., :32], masks[0][1][..., C - 32:])
        else:
            out = fuser(ch, 1, gh, 1)
        assert out.requires_grad
        ref = orc.transformer_fusion(sdr, cr, cc, masks=masks)
        assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max())), (C, train)
        (ref * wgt).sum().backward()
        (out * wgt.to(DEV)).sum().backward()
        # The fuser is piecewise linear in its ReLUs (trans_conv, linear1, the InstanceNorm's): a unit at ~0 that the split-f16 forward
        # and the f32 oracle decide differently moves the gradient of THAT point's row by ~1e-2 of the scale and everything else by
        # ~1 / (B N) of it (tools/probe/fusion_wide_bwd_err.py: one seed in eight; 5e-6 otherwise).  So: at most two rows of the input
        # gradients beyond the tolerance, and the parameters to 2e-4 when no row is, 2e-2 when one is.
        flipped = 0
        for got, want, tag in ((ch.grad.cpu(), cr.grad, "d c_img"), (gh.grad.cpu(), cc.grad, "d c")):
            bad = ((got - want).abs() > 2e-4 * float(want.abs().max())).any(dim=-1)
            assert int(bad.sum()) <= 2, (tag, C, train, int(bad.sum()))
            flipped += int(bad.sum())
        tol = 2e-4 if flipped == 0 else 2e-2
        gscale = max(float(v.grad.abs().max()) for v in sdr.values() if v.grad is not None)
        checked = 0
        for name, prm in fuser.named_parameters():
            if "after_norm" in name:
                continue
            want = sdr[name].grad
            twin = name.replace("encoder.layers.0.self_attn", "decoder.layers.0.self_attn")
            if twin != name and sdr[twin].grad is not None:
                want = want + sdr[twin].grad if want is not None else sdr[twin].grad
            err = float((prm.grad.cpu() - want).abs().max())
            assert err <= tol * max(float(want.abs().max()), 1e-2 * gscale), (name, C, train, err, float(want.abs().max()), flipped)
            checked += 1
        assert checked == 20
