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


def test_finger_id_outside_the_table_reads_as_no_feature():
    """An id >= n_fingers (a table with fewer rows than the ids were assigned against) must never index past the table: every
    by-id kernel (shipped decoder in all four arithmetics, the wide decoder exact and split-f16, the fuser) treats it like 255 --
    the result equals the one with those ids rewritten to 255, bit for bit.  (The host gather this replaces raised instead; the
    table here is the LAST allocation of a fresh segment so that a read past it would at least fetch other bytes.)"""
    from vtaco_amd import ops
    from vtaco_amd.conv_onet.models import decoder_dict
    from vtaco_amd.transformer_fusion import TransformerFusion
    a, sd = load_golden("g1_decode.npz")
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, with_contact=True)
    dec.load_state_dict(sd, strict=True)
    dec.to(DEV)
    g = torch.Generator().manual_seed(7)
    nx = 32
    grid = torch.from_numpy(a["grid"]).to(DEV)
    ids = torch.full((1, nx ** 3), 255, dtype=torch.uint8)
    pick = torch.rand(nx ** 3, generator=g) < 0.1
    ids[0, pick] = torch.randint(0, 9, (int(pick.sum()),), generator=g).to(torch.uint8)      # ids 3..8 are outside a 3-row table
    clean = torch.where(ids >= 3, torch.full_like(ids, 255), ids)
    assert int(((ids >= 3) & (ids != 255)).sum()) > 50
    feats = torch.randn(3, 32, generator=g).to(DEV)
    with torch.no_grad():
        for prec in ops.PRECISIONS:
            got = dec.decode_lattice_ids(grid, nx, ids.to(DEV), feats, precision=prec)
            ref = dec.decode_lattice_ids(grid, nx, clean.to(DEV), feats, precision=prec)
            assert torch.equal(got, ref), prec
        torch.manual_seed(5)
        wdec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=64, n_blocks=3).to(DEV).eval()
        for prec in ("f32", "f16x3"):
            got = wdec.decode_lattice_ids(grid, nx, ids.to(DEV), feats, precision=prec)
            ref = wdec.decode_lattice_ids(grid, nx, clean.to(DEV), feats, precision=prec)
            assert torch.equal(got, ref), ("wide", prec)
        torch.manual_seed(3)
        fuser = TransformerFusion(use_xyz=True, input_size=2048, d_model=32, num_layers=1, key_feature_dim=64, with_pos_embed=False,
                                  encoder_pos_embed_input_dim=3, decoder_pos_embed_input_dim=3).to(DEV).eval()
        c = torch.randn(4, 512, 32, generator=g).to(DEV)
        fid, fclean = ids[0, :2048].reshape(4, 512).to(DEV), clean[0, :2048].reshape(4, 512).to(DEV)
        assert torch.equal(fuser.forward_ids(fid, feats, c), fuser.forward_ids(fclean, feats, c))


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


def test_winding_number_kernel_against_the_oracle():
    import numpy as np
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    # a closed, outward-oriented mesh with many faces: a subdivided cube surface projected on a sphere
    n = 9
    g = np.linspace(-1, 1, n)
    verts, faces, index = [], [], {}

    def vid(p):
        key = tuple(np.round(p, 9))
        if key not in index:
            index[key] = len(verts)
            verts.append(np.array(p) / np.linalg.norm(p) * 0.4)
        return index[key]

    for axis in range(3):
        for sign in (-1.0, 1.0):
            for i in range(n - 1):
                for j in range(n - 1):
                    def pt(a, b):
                        q = [0.0, 0.0, 0.0]
                        q[axis] = sign
                        q[(axis + 1) % 3], q[(axis + 2) % 3] = g[a], g[b]
                        return q
                    quad = [vid(pt(i, j)), vid(pt(i + 1, j)), vid(pt(i + 1, j + 1)), vid(pt(i, j + 1))]
                    if sign < 0:
                        quad = quad[::-1]
                    faces += [[quad[0], quad[1], quad[2]], [quad[0], quad[2], quad[3]]]
    verts, faces = np.array(verts, dtype=np.float32), np.array(faces, dtype=np.int64)
    pts = ((rs.rand(3001, 3) - 0.5) * 1.2).astype(np.float32)
    ref = orc.winding_number(verts, faces, pts)
    got = ops.winding_number(torch.from_numpy(verts).to(dev), torch.from_numpy(faces).to(dev), torch.from_numpy(pts).to(dev)).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-6
    inside = np.linalg.norm(pts, axis=1) < 0.39
    outside = np.linalg.norm(pts, axis=1) > 0.41
    assert np.allclose(ref[inside], 1, atol=1e-9) and np.allclose(ref[outside], 0, atol=1e-9) and inside.sum() > 100
    assert ops.winding_number(torch.from_numpy(verts).to(dev), torch.from_numpy(faces).to(dev), torch.zeros(0, 3, device=dev)).shape == (0,)


def test_trainer_compute_loss_t2d_img_matches_the_reference_trainer():
    """Trainer.compute_loss_t2d_img (VTacO step) on stand-in encoders returning the fixture's tensors: under the reference run's
    numpy seed, decode_img receives the same points and features and the losses equal the real reference Trainer's -- with and
    without the t2d net's own losses (g13_trainer_t2d.npz; libigl's call answered by the exact winding number there)."""
    import os
    import types
    import numpy as np
    from conftest import GOLDEN
    from vtaco_amd.conv_onet.training import Trainer
    z = np.load(os.path.join(GOLDEN, "g13_trainer_t2d.npz"))
    z12 = np.load(os.path.join(GOLDEN, "g12_t2d.npz"))
    dev = torch.device("cuda:0")
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    depths = torch.from_numpy(np.stack([z12["depths"], np.roll(z12["depths"], 7, axis=0)]))
    pred_depth = (torch.linspace(0, 1, 240 * 320).view(1, 1, -1) * torch.tensor([0.2, 0.4, 0.6, 0.8, 1.0]).view(1, 5, 1)).expand(2, 5, -1)
    seen = {}

    class StandIn(object):
        def train(self):
            return self

        def encode_t2d(self, inputs, imgs):
            return pred_depth.to(dev), {"mano_param": t("digit")}

        def encode_inputs(self, inputs):
            return "c"

        def encode_hand_inputs(self, inputs):
            return {"mano_param": t("mano_param"), "mano_verts": t("mano_verts")}

        def encode_img_inputs(self, imgs):
            return t("c_img")

        def decode(self, p_sample, c, **kw):
            seen["p_sample_plain"] = p_sample
            return types.SimpleNamespace(logits=p_sample.sum(-1) * 0.5)

        def decode_img(self, p_sample, c, c_img_all, **kw):
            seen["p_sample"], seen["c_img_all"] = p_sample, c_img_all
            return types.SimpleNamespace(logits=p_sample.sum(-1) * 0.5 + c_img_all.sum(-1) * 0.01)

    data = {"points": t("p"), "points.mano": t("mano"), "points.pc_hand": t("pc_hand"), "points.name": ["cube", "tet"],
            "points.cam_pos": t("cam_pos"), "points.cam_rot": t("cam_rot"), "inputs": torch.zeros(2, 16, 3), "inputs.pc_ply": t("pc_ply"),
            "inputs.img": torch.zeros(2, 5, 3, 8, 6), "inputs.depth": depths, "inputs.touch_success": t("touch")}
    vf = {"cube": {"v": z["cube_v"], "f": z["cube_f"]}, "tet": {"v": z["tet_v"], "f": z["tet_f"]}}
    for pretrained, key in ((True, "loss_pretrained"), (False, "loss_joint")):
        trainer = Trainer(StandIn(), None, device=dev, num_sample=int(z["num_sample"]), with_img=True, encode_t2d=True,
                          pretrained_t2d=pretrained, depth_origin=z12["depth_origin"])
        state = np.random.get_state()
        try:
            np.random.seed(int(z["seed"]))
            out = trainer.compute_loss_t2d_img(data, vf)
        finally:
            np.random.set_state(state)
        assert torch.equal(seen["p_sample"].cpu(), torch.from_numpy(z["p_sample"]))
        assert torch.equal(seen["c_img_all"].cpu(), torch.from_numpy(z["c_img_all"]))
        for got, ref in zip(out, z[key]):
            assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (key, float(got), float(ref))
    # the variant without tactile features (compute_loss_t2d): it normalises the depth images before looking for contact pixels
    for pretrained, key in ((True, "loss_plain_pretrained"), (False, "loss_plain_joint")):
        trainer = Trainer(StandIn(), None, device=dev, num_sample=int(z["num_sample"]), with_img=False, encode_t2d=True,
                          pretrained_t2d=pretrained, depth_origin=z12["depth_origin"])
        state = np.random.get_state()
        try:
            np.random.seed(int(z["seed"]))
            out = trainer.compute_loss_t2d(data, vf)
        finally:
            np.random.set_state(state)
        assert torch.equal(seen["p_sample_plain"].cpu(), torch.from_numpy(z["p_sample_plain"]))
        for got, ref in zip(out, z[key]):
            assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (key, float(got), float(ref))


def test_trainer_vtaco_t2d_step_with_the_shipped_module_types(tmp_path):
    """One VTacO training step built by get_model / get_trainer from a config shaped like configs/VTacO/VTacO_YCB.yaml: object
    encoder + UNet3D, hand encoder + MANO layer, Resnet18 tactile features, the t2d net (depth U-Net + digit-pose regressor),
    concat decoder; contact clouds from the depth images, winding-number targets from meshes read from .off / .obj files."""
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import synth_mano
    from conftest import GOLDEN
    from vtaco_amd.conv_onet import config as cfgmod
    from vtaco_amd.data import load_mesh_dict
    z12 = np.load(os.path.join(GOLDEN, "g12_t2d.npz"))
    z13 = np.load(os.path.join(GOLDEN, "g13_trainer_t2d.npz"))
    dev = torch.device("cuda:0")
    synth_mano.write_pkl(synth_mano.make_asset(0), str(tmp_path / "mano"))
    (tmp_path / "cube.off").write_text("OFF\n8 12 0\n" + "\n".join(" ".join(str(float(c)) for c in v) for v in z13["cube_v"]) + "\n" +
                                       "\n".join("3 " + " ".join(str(int(i)) for i in f) for f in z13["cube_f"]) + "\n")
    (tmp_path / "tet.obj").write_text("\n".join("v " + " ".join(str(float(c)) for c in v) for v in z13["tet_v"]) + "\n" +
                                      "\n".join("f " + " ".join(str(int(i) + 1) for i in f) for f in z13["tet_f"]) + "\n")
    vf = load_mesh_dict(str(tmp_path), ["cube", "tet"])
    mano_kw = dict(center_idx=9, flat_hand_mean=False, ncomps=45, side="right", mano_root=str(tmp_path / "mano"), use_pca=False,
                   root_rot_mode="axisang", joint_rot_mode="axisang", robust_rot=False, return_transf=False)
    hand = {"hidden_dim": 32, "plane_type": ["xz", "xy", "yz"], "plane_resolution": 32, "unet": True,
            "unet_kwargs": {"depth": 2, "merge_mode": "concat", "start_filts": 8}, "out_mano": True}
    cfg = {"data": {"dim": 3, "padding": 0.1, "input_type": "pointcloud", "num_sample": 640}, "test": {"threshold": 0.5},
           "model": {"c_dim": 32, "decoder": "simple_local", "decoder_kwargs": {"sample_mode": "bilinear", "hidden_size": 32},
                     "encoder": "pointnet_local_pool",
                     "encoder_kwargs": {"hidden_dim": 32, "plane_type": "grid", "grid_resolution": 32, "unet3d": True,
                                        "unet3d_kwargs": {"num_levels": 3, "f_maps": 32, "in_channels": 32, "out_channels": 32}},
                     "encoder_hand": "pointnet_local_pool", "encoder_hand_kwargs": dict(hand, out_dim=51, manolayer_kwargs=mano_kw),
                     "with_img": True, "encoder_img": "Resnet18", "encoder_img_kwargs": {"num_classes": 32},
                     "encoder_t2d": True,
                     "encoder_t2d_kwargs": {"pretrained": False, "encoder_img": "UNet",
                                            "encoder_img_kwargs": {"num_classes": 1, "in_channel": 3, "start_filts": 8, "depth": 2},
                                            "encoder_hand": "pointnet_local_pool", "encoder_hand_kwargs": dict(hand, c_dim=16, out_dim=30)}}}
    torch.manual_seed(0)
    model = cfgmod.get_model(cfg, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    trainer = cfgmod.get_trainer(model, opt, cfg, dev, depth_origin=z12["depth_origin"])
    assert trainer.encode_t2d and trainer.with_img and not trainer.pretrained_t2d and trainer.num_sample == 640
    g = torch.Generator().manual_seed(2)
    d = torch.randn(2, 300, 3, generator=g)
    batch = {"inputs": 0.3 * d / d.norm(dim=-1, keepdim=True), "points": (torch.rand(2, 900, 3, generator=g) - 0.5) * 1.1,
             "points.mano": torch.randn(2, 51, generator=g) * 0.2, "points.pc_hand": torch.randn(2, 778, 3, generator=g) * 0.05,
             "points.name": ["cube", "tet"], "points.cam_pos": torch.from_numpy(z13["cam_pos"]), "points.cam_rot": torch.from_numpy(z13["cam_rot"]),
             "inputs.pc_ply": torch.from_numpy(z13["pc_ply"]), "inputs.img": torch.rand(2, 5, 3, 320, 240, generator=g),
             "inputs.depth": torch.from_numpy(np.stack([z12["depths"], np.roll(z12["depths"], 7, axis=0)])),
             "inputs.touch_success": torch.from_numpy(z13["touch"])}
    np.random.seed(0)
    first = trainer.train_step(batch, vf)
    for _ in range(4):
        last = trainer.train_step(batch, vf)
    assert all(np.isfinite(x) for x in first + last) and last[0] < first[0], (first, last)
    for name in ("encoder", "encoder_hand", "encoder_img", "decoder"):
        got = [p.grad is not None for n, p in getattr(model, name).named_parameters() if "fc_out_contact" not in n and "fc_p." not in n]
        assert all(got), name
    assert any(p.grad is not None and p.grad.abs().sum() > 0 for p in model.encoder_t2d.parameters())       # joint training of the t2d net
    from vtaco_amd._lib import VtError
    with pytest.raises(VtError, match="vf_dict"):
        trainer.train_step(batch)


def test_trainer_train_tactile_step():
    """The t2d net trained on its own (training.py:950-986): depth L1 + digit-pose MSE, against the same two terms written out."""
    import numpy as np
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork
    from vtaco_amd.conv_onet.training import Trainer
    from vtaco_amd.encoder import encoder_dict
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    depth_net = encoder_dict["UNet"](num_classes=1, in_channels=3, depth=2, start_filts=8)
    digits = encoder_dict["pointnet_local_pool"](dim=3, c_dim=16, padding=0.1, hidden_dim=32, plane_type=["xz", "xy", "yz"],
                                                 plane_resolution=32, unet=False, out_mano=True, out_dim=30)
    model = ConvolutionalOccupancyNetwork(None, None, digits, depth_net, None, device=dev)
    g = torch.Generator().manual_seed(2)
    data = {"inputs": torch.randn(2, 300, 3, generator=g) * 0.2, "inputs.img": torch.rand(2, 5, 3, 16, 12, generator=g),
            "inputs.depth": 0.019 + 0.003 * torch.rand(2, 5, 16 * 12, generator=g), "points.cam_pos": torch.randn(2, 5, 3, generator=g) * 0.1,
            "points.cam_rot": torch.randn(2, 5, 3, generator=g)}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    trainer = Trainer(model, opt, device=dev, train_tactile=True)
    model.train()
    loss, ld, lg = trainer.compute_loss_tactile(data)
    d = data["inputs.depth"].to(dev)
    d = (d - d.min()) / (d.max() - d.min())
    ref_d = torch.nn.functional.l1_loss(model.encode_img_inputs(data["inputs.img"].to(dev)), d)
    ref_g = torch.nn.functional.mse_loss(model.encode_hand_inputs(data["inputs"].to(dev))["mano_param"],
                                         torch.cat((data["points.cam_pos"].reshape(2, -1), data["points.cam_rot"].reshape(2, -1)), 1).to(dev))
    f = lambda t: float(t.detach())
    assert abs(f(ld) - f(ref_d)) <= 1e-6 and abs(f(lg) - f(ref_g)) <= 1e-6 and abs(f(loss) - f(ref_d + ref_g)) <= 1e-6
    first = trainer.train_step(data)
    for _ in range(10):
        last = trainer.train_step(data)
    assert len(first) == 3 and last[0] < first[0]
    model.encoder_hand = None
    assert len(Trainer(model, torch.optim.Adam(model.parameters(), lr=1e-3), device=dev, train_tactile=True).train_step(data)) == 2


def test_contact_clouds_on_the_device_equal_the_host_rule():
    """vt_contact_scan + vt_contact_points against ``contact_clouds_from_depth`` (the reference's numpy rule, pinned by g12 / g13):
    the touched pixels (count and order), the same ``randint`` draws under the same seed, the points to float32 rounding -- on the
    g12 fixture's real depth images and on synthetic ones with more than 128 touched pixels and an untouched sensor."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from vtaco_amd import ops
    from vtaco_amd.common import contact_clouds_from_depth, contact_clouds_on_device
    z = np.load(os.path.join(GOLDEN, "g12_t2d.npz"))
    H, W = 320, 240
    g = torch.Generator().manual_seed(3)
    cases = [(z["depths"][None].astype(np.float32), z["depth_origin"].astype(np.float64), z["cam_pos"].astype(np.float64),
              z["cam_rot"].astype(np.float64), z["pc_ply"].astype(np.float32), np.asarray(z["touch"]).reshape(1, 5))]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    depth = torch.full((3, 5, H * W), 0.02)
    for b in range(3):
        for t in range(5):
            r = (4, 9, 15, 30, 2)[(b + t) % 5]
            disc = ((yy - 60 - 40 * t) ** 2 + (xx - 50 - 30 * b) ** 2) < r * r
            depth[b, t][disc.reshape(-1)] = 0.02 - 0.002 * torch.rand(int(disc.sum()), generator=g) - 0.0005
    d = torch.randn(3, 5, 3, generator=g)
    touch = np.array([[1, 1, 0, 1, 1], [1, 0, 1, 1, 1], [1, 1, 1, 1, 0]], dtype=np.uint8)
    cases.append((depth.numpy(), np.full(H * W, 0.02), (0.32 * d / d.norm(dim=-1, keepdim=True)).double().numpy(),
                  (torch.rand(3, 5, 3, generator=g) * 2 - 1).double().numpy(), (torch.randn(3, 500, 3, generator=g) * 0.2).numpy(), touch))
    for depths, origin, cam_pos, cam_rot, pc_ply, touch in cases:
        B, S, N = depths.shape[0], 1024, 4000
        p_host = torch.rand(B, N, 3, generator=g).numpy()
        state = np.random.get_state()
        try:
            np.random.seed(21)
            ref = np.zeros((B, S, 3), dtype=np.float32)
            ref_f = np.full((B, S), -1, dtype=np.int64)
            for b in range(B):
                anchors, count = contact_clouds_from_depth(depths[b], origin, cam_pos[b], cam_rot[b], pc_ply[b], touch[b])
                k = 0
                for t in range(5):
                    if touch[b][t]:
                        n = int(count[t])
                        ref[b, k:k + n], ref_f[b, k:k + n] = anchors[t, :n].astype(np.float32), t
                        k += n
                ref[b, k:] = p_host[b][np.random.randint(N, size=S - k)]
            np.random.seed(21)
            buf = np.zeros((B, S, 3), dtype=np.float32)
            got, got_f = contact_clouds_on_device(torch.from_numpy(depths).to(DEV), torch.from_numpy(origin).to(DEV), cam_pos, cam_rot, pc_ply,
                                                  touch, buf, p_host, S)
            after = np.random.randint(1 << 30)                           # ... and both consumed the generator equally
            np.random.seed(21)
            for b in range(B):
                contact_clouds_from_depth(depths[b], origin, cam_pos[b], cam_rot[b], pc_ply[b], touch[b])
                np.random.randint(N, size=S - int((ref_f[b] >= 0).sum()))
            assert after == np.random.randint(1 << 30)
        finally:
            np.random.set_state(state)
        assert np.array_equal(got_f, ref_f) and int((ref_f >= 0).sum()) > 20
        err = np.abs(got.cpu().numpy() - ref)
        assert float(err.max()) <= 1e-6 * max(1.0, float(np.abs(ref).max())), float(err.max())
        assert float((err > 0).mean()) < 0.01                               # float64 arithmetic in the same order: equal but for rare ties
