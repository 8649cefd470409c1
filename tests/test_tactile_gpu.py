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


def test_vtaco_t2d_route_matches_the_reference_generator():
    """Generator3D's VTacO branch (generation.py:202-257) against the real reference generator's assignment (g12_t2d.npz):
    contact clouds from the depth images -> vt_tactile_assign('within') on the 128^3 lattice gives the same finger per lattice
    point, and generate_obj_mesh_wnf(with_img, encode_t2d) runs the chain end to end (at another lattice size, which the
    reference cannot)."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from vtaco_amd import ops
    from vtaco_amd.common import contact_clouds_from_depth
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    from vtaco_amd.encoder import encoder_dict
    from vtaco_amd._lib import VtError
    z = np.load(os.path.join(GOLDEN, "g12_t2d.npz"))
    dev = torch.device("cuda:0")
    state = np.random.get_state()
    try:
        np.random.seed(int(z["seed"]))
        anchors, count = contact_clouds_from_depth(z["depths"], z["depth_origin"], z["cam_pos"][0], z["cam_rot"][0], z["pc_ply"][0],
                                                   z["touch"][0])
    finally:
        np.random.set_state(state)
    nx = int(z["nx"])
    ids = ops.tactile_assign(torch.from_numpy(anchors).float().to(dev), torch.from_numpy((count > 0).astype(np.uint8)).to(dev), "within",
                             0.015, lattice=(nx, 1.1, 0, nx ** 3), count=torch.from_numpy(count).int().to(dev))
    assert torch.equal(ids[0].cpu(), torch.from_numpy(z["ids"]))
    # end to end on a small model
    torch.manual_seed(3)
    dec = decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32)
    enc = encoder_dict["pointnet_local_pool"](c_dim=32, dim=3, hidden_dim=32, grid_resolution=16, plane_type="grid", unet3d=False)
    for blk in list(dec.blocks) + list(enc.blocks):
        torch.nn.init.normal_(blk.fc_1.weight, 0, 0.1)
    img = encoder_dict["UNet"](num_classes=1, in_channels=3, depth=2, start_filts=8)
    model = ConvolutionalOccupancyNetwork(dec, enc, None, img, None, device=dev)
    g = torch.Generator().manual_seed(4)
    d = torch.randn(1, 3000, 3, generator=g)
    data = {"inputs": 0.3 * d / d.norm(dim=-1, keepdim=True), "inputs.img": torch.rand(1, 5, 3, 8, 4, generator=g),
            "inputs.depth": torch.from_numpy(z["depths"])[None], "inputs.touch_success": torch.from_numpy(z["touch"]),
            "inputs.pc_ply": torch.from_numpy(z["pc_ply"]), "points.cam_pos": torch.from_numpy(z["cam_pos"]),
            "points.cam_rot": torch.from_numpy(z["cam_rot"])}
    gen = Generator3D(model, device=dev, resolution0=16, padding=0.1, with_img=True, encode_t2d=True, decode_precision="f32",
                      depth_origin=z["depth_origin"])
    mesh = gen.generate_obj_mesh_wnf(data)
    assert mesh.vertices.shape[0] > 0 and mesh.faces.shape[1] == 3
    plain = Generator3D(model, device=dev, resolution0=16, padding=0.1, decode_precision="f32").generate_obj_mesh_wnf(data)
    assert plain.vertices.shape != mesh.vertices.shape or not torch.equal(plain.vertices, mesh.vertices)   # the touch changed the surface
    with pytest.raises(VtError, match="depth_origin"):
        Generator3D(model, device=dev, resolution0=16, with_img=True, encode_t2d=True,
                    depth_origin="/nonexistent/depth_origin.txt").generate_obj_mesh_wnf(data)


def test_tactile_generation_with_the_attention_decoder():
    """BASELINE config 3's decoder (attention_local) through generate_obj_mesh_tactile: chunked like eval_points (the chunk is
    part of the function), per-chunk features gathered from the finger ids -- equal to eval_points on the dense c_img_all."""
    import numpy as np
    from vtaco_amd import ops
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    from vtaco_amd.encoder import encoder_dict
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    dec = decoder_dict["attention_local"](dim=3, c_dim=32, hidden_size=32)
    enc = encoder_dict["pointnet_local_pool"](c_dim=32, dim=3, hidden_dim=32, grid_resolution=16, plane_type="grid", unet3d=False)
    for blk in list(dec.blocks) + list(enc.blocks):
        torch.nn.init.normal_(blk.fc_1.weight, 0, 0.1)
    model = ConvolutionalOccupancyNetwork(dec, enc, device=dev).eval()
    g = torch.Generator().manual_seed(6)
    d = torch.randn(1, 3000, 3, generator=g)
    data = {"inputs": 0.3 * d / d.norm(dim=-1, keepdim=True)}
    feats = torch.randn(5, 32, generator=g)
    anchors = (torch.rand(5, 16, 3, generator=g) - 0.5) * 0.8
    success = torch.tensor([1, 1, 0, 1, 1], dtype=torch.uint8)
    gen = Generator3D(model, device=dev, resolution0=4, padding=0.1, with_img=True, points_batch_size=1024, decode_precision="f32")
    nx = 16
    mesh = gen.generate_obj_mesh_tactile(data, feats, anchors, success, mode="within", radius=0.08)
    ids = ops.tactile_assign(anchors.to(dev), success.to(dev), "within", 0.08, lattice=(nx, 1.1, 0, nx ** 3))
    assert int((ids != 255).sum()) > 20
    dense = torch.zeros(nx ** 3, 32, device=dev)
    sel = ids[0] != 255
    dense[sel] = feats.to(dev)[ids[0][sel].long()]
    from oracle import vtaco_oracle as orc
    pts = 1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
    with torch.no_grad():
        c = model.encode_inputs(data["inputs"].to(dev))
    vals = gen.eval_points(pts, c, dense.unsqueeze(0))                    # the reference's chunk loop, dense features
    ref = gen.extract_mesh(vals.to(dev).reshape(nx, nx, nx))
    assert torch.equal(mesh.faces, ref.faces) and torch.equal(mesh.vertices, ref.vertices)
