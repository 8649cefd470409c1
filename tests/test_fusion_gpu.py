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
    """AttentionDecoder.forward_img under autograd: HIP sampling (vt_sample_grid / _bwd) and conditioned MLP
    (vt_decode_mlp_fwd_train / vt_decode_mlp_bwd / vt_decode_wgrad) around the fuser's host-PyTorch form,
    against torch-CPU autograd of the oracle: logits, d grid, d c_img and every parameter gradient."""
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
    assert rel(gh.grad.cpu(), gr.grad) <= 1e-3
    assert rel(ch.grad.cpu(), cr.grad) <= 1e-3
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
        assert float((prm.grad.cpu() - want).abs().max()) <= 1e-3 * max(float(want.abs().max()), 1e-3 * gscale), name
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
