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
