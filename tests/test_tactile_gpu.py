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


def test_generator_tactile_mesh_equals_dense_c_img_all_path():
    """Generator3D.generate_obj_mesh_tactile (finger ids + feature table) produces the mesh of the reference-style path
    that materialises c_img_all [1, nx^3, C] from the oracle's assignment rule and decodes with forward_img."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    from vtaco_amd.encoder import encoder_dict
    a, sd_e = load_golden("g3_pointnet.npz")
    _, sd_d = load_golden("g1_decode.npz")
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, with_contact=True)
    dec.load_state_dict(sd_d, strict=True)
    enc = encoder_dict['pointnet_local_pool'](c_dim=32, dim=3, hidden_dim=32, grid_resolution=16, plane_type='grid')
    enc.load_state_dict(sd_e, strict=True)
    model = ConvolutionalOccupancyNetwork(dec, enc, device=DEV)
    gen = Generator3D(model, device=DEV, resolution0=8, padding=0.1, with_img=True)
    g = torch.Generator().manual_seed(4)
    nx = 32
    d = torch.randn(5, 40, 3, generator=g)
    clouds = 0.3 * d / d.norm(dim=-1, keepdim=True) + 0.01 * torch.randn(5, 40, 3, generator=g)
    count = torch.tensor([40, 25, 40, 1, 33])
    success = torch.tensor([1, 1, 0, 1, 1])
    feats = torch.randn(5, 32, generator=g)
    p = torch.from_numpy(a["p"])[:1]
    mesh = gen.generate_obj_mesh_tactile({"inputs": p}, feats, clouds, success, mode="within", count=count)
    ids = orc.tactile_assign_within(_lattice(nx).numpy(), clouds.numpy(), count.numpy(), success.numpy())
    dense = torch.zeros(1, nx ** 3, 32)
    hit = torch.from_numpy(ids != 255)
    dense[0, hit] = feats[torch.from_numpy(ids[ids != 255]).long()]
    assert int(hit.sum()) > 20
    ref = gen.generate_obj_mesh_wnf({"inputs": p}, c_img_all=dense.to(DEV))
    assert torch.equal(mesh.faces, ref.faces) and torch.equal(mesh.vertices, ref.vertices)
