"""GPU parity of the tactile feature assignment (finger id per point) and of decoding with
(ids, feature table) against the dense c_img_all path the reference builds."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lattice(nx):
    from oracle import vtaco_oracle as orc
    return 1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)


def test_nearest_fingertip_rule_bit_exact():
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    g = torch.Generator().manual_seed(0)
    nx = 32
    pts = _lattice(nx)
    d = torch.randn(5, 3, generator=g)
    tips = 0.3 * d / d.norm(dim=-1, keepdim=True)
    success = torch.tensor([1, 0, 1, 1, 1])
    ref = orc.tactile_assign_nearest(pts.numpy(), tips.numpy(), success.numpy())
    ids = ops.tactile_assign(tips.view(5, 1, 3).to(DEV), success.to(DEV), "nearest", 0.05, lattice=(nx, 1.1, 0, nx ** 3))
    assert np.array_equal(ids.cpu().numpy()[0].astype(np.int64), ref)
    assert (ref != 255).sum() > 10
    ids_p = ops.tactile_assign(tips.view(5, 1, 3).to(DEV), success.to(DEV), "nearest", 0.05, pts=pts.unsqueeze(0).to(DEV))
    assert torch.equal(ids, ids_p)


def test_contact_cloud_rule_bit_exact():
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    g = torch.Generator().manual_seed(1)
    nx = 32
    pts = _lattice(nx)
    centers = 0.3 * torch.nn.functional.normalize(torch.randn(5, 1, 3, generator=g), dim=-1)
    clouds = centers + 0.02 * torch.randn(5, 128, 3, generator=g)
    counts = torch.tensor([128, 40, 0, 128, 7])
    success = torch.tensor([1, 1, 1, 0, 1])
    ref = orc.tactile_assign_within(pts.numpy(), clouds.numpy(), counts.numpy(), success.numpy())
    ids = ops.tactile_assign(clouds.to(DEV), success.to(DEV), "within", 0.015, lattice=(nx, 1.1, 0, nx ** 3), count=counts.to(DEV))
    assert np.array_equal(ids.cpu().numpy()[0].astype(np.int64), ref)
    assert (ref != 255).sum() > 10


def test_decode_by_finger_id_equals_dense_c_img():
    from vtaco_amd import ops
    from vtaco_amd.conv_onet.models import decoder_dict
    a, sd = load_golden("g1_decode.npz")
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, with_contact=True)
    dec.load_state_dict(sd, strict=True)
    dec.to(DEV)
    g = torch.Generator().manual_seed(2)
    nx = 32
    grid = torch.from_numpy(a["grid"]).to(DEV)
    ids = torch.full((1, nx ** 3), 255, dtype=torch.uint8)
    pick = torch.rand(nx ** 3, generator=g) < 0.05
    ids[0, pick] = torch.randint(0, 5, (int(pick.sum()),), generator=g).to(torch.uint8)
    feats = torch.randn(5, 32, generator=g)
    dense = torch.zeros(1, nx ** 3, 32)
    sel = ids[0] != 255
    dense[0, sel] = feats[ids[0, sel].long()]
    with torch.no_grad():
        by_id = dec.decode_lattice_ids(grid, nx, ids.to(DEV), feats.to(DEV))
        ref = dec.decode_lattice(grid, nx, c_img=dense.to(DEV))
    assert torch.equal(by_id, ref)
