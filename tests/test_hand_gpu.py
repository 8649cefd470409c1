"""GPU parity of the hand branch through the C ABI: plane cell ids and scatter-mean planes
(vt_plane_*), the shared max-pool kernels over planes, the MANO layer (vt_mano_*), and the
LocalPoolPointnet hand encoder end to end -- against the oracle and the reference-made golden
g10_hand.npz (synthetic MANO asset of tests/synth_mano.py)."""
import numpy as np
import pytest
import torch

import synth_mano
from conftest import load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy
PLANES = ("xz", "xy", "yz")
MANO_KW = dict(center_idx=9, flat_hand_mean=False, ncomps=45, side="right", use_pca=False,
               root_rot_mode="axisang", joint_rot_mode="axisang", robust_rot=False, return_transf=False)


def maxdiff(a, b):
    return float((torch.as_tensor(a).cpu() - torch.as_tensor(b).cpu()).abs().max())


def test_plane_ids_bit_exact_and_scatter_mean():
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    a, _ = load_golden("g10_hand.npz")
    dev = torch.device("cuda:0")
    p = T(a["p"]).to(dev)
    g = torch.Generator().manual_seed(7)
    feat = torch.randn(2, 3000, 32, generator=g)
    for k in PLANES:
        pi = ops.PlaneIndex(p, 32, 0.1, k)
        assert torch.equal(pi.idx.cpu(), T(a["idx_" + k]))              # golden ids from the reference, bit-exact
        got = ops.plane_scatter_mean_fwd(feat.to(dev), pi)
        ref = orc.scatter_mean_plane(feat, T(a["idx_" + k]).long(), 32)
        assert maxdiff(got, ref) <= 1e-6
        assert torch.equal(got.cpu() == 0, ref == 0)                    # empty cells exactly zero
        # backward: d feat = d plane[cell] / count
        gp = torch.randn(2, 32, 32, 32, generator=g)
        f = feat.clone().requires_grad_(True)
        (orc.scatter_mean_plane(f, T(a["idx_" + k]).long(), 32) * gp).sum().backward()
        assert maxdiff(ops.plane_scatter_mean_bwd(gp.to(dev), pi, 32), f.grad) <= 1e-6
    # ragged sizes, other resolutions, points far outside the box
    for (B, Tn, R) in ((1, 1, 8), (3, 257, 64), (2, 8192, 128)):
        pts = (torch.rand(B, Tn, 3, generator=g) - 0.5) * 1.6
        for k in PLANES:
            pi = ops.PlaneIndex(pts.to(dev), R, 0.1, k)
            assert torch.equal(pi.idx.cpu().long(), orc.plane_index(pts, R, 0.1, k))
    from vtaco_amd._lib import VtError
    with pytest.raises(VtError):
        ops.PlaneIndex(p, 32, 0.1, "zx")


def test_pool_local_sums_the_three_planes():
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    a, _ = load_golden("g10_hand.npz")
    dev = torch.device("cuda:0")
    p = T(a["p"])
    feat = torch.randn(2, 3000, 32, generator=torch.Generator().manual_seed(8))
    ref = sum(orc.segment_pool_max(feat, T(a["idx_" + k]).long()) for k in PLANES)
    got = 0
    for k in PLANES:
        got = got + ops.voxel_pool_max_fwd(feat.to(dev), ops.PlaneIndex(p.to(dev), 32, 0.1, k))[0]
    assert maxdiff(got, ref) <= 1e-6


def test_plane_indices_in_one_launch_equal_the_single_builds():
    """vt_plane_build_multi (the three planes' sorts side by side) writes what three vt_plane_build calls write: ids, sorted order, segment
    bounds -- on the golden cloud, a ragged one, one with every point in one cell (a 700-point segment) and through the large-cloud path."""
    import os
    from vtaco_amd import ops
    a, _ = load_golden("g10_hand.npz")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    clouds = [T(a["p"]), 0.5 * (torch.rand(3, 1237, 3, generator=g) - 0.5), 0.001 * torch.randn(2, 700, 3, generator=g),
              0.6 * (torch.rand(1, 9000, 3, generator=g) - 0.5)]
    for p in clouds:
        p = p.to(dev)
        for planes in (("xz", "xy", "yz"), ("yz", "xz"), ("xy",)):
            many = ops.plane_indices(p, 32, 0.1, planes)
            for pi, k in zip(many, planes):
                one = ops.PlaneIndex(p, 32, 0.1, k)
                for f in ("idx", "order", "seg_lo", "seg_hi"):
                    assert torch.equal(getattr(pi, f), getattr(one, f)), (tuple(p.shape), planes, k, f)
                # the bounds are those of the sorted ids: [lo, hi) is the maximal run of the point's cell
                ids = torch.gather(one.idx, 1, one.order.long())
                lo = torch.gather(one.seg_lo, 1, one.order.long()).long()
                hi = torch.gather(one.seg_hi, 1, one.order.long()).long()
                pos = torch.arange(p.shape[1], device=dev).expand_as(ids)
                assert bool(((lo <= pos) & (pos < hi)).all()) and bool((torch.gather(ids, 1, lo) == ids).all()) and bool((torch.gather(ids, 1, hi - 1) == ids).all())
                assert bool(((lo == 0) | (torch.gather(ids, 1, (lo - 1).clamp(min=0)) != ids)).all())
                assert bool(((hi == p.shape[1]) | (torch.gather(ids, 1, hi.clamp(max=p.shape[1] - 1)) != ids)).all())


def test_plane_scatter_means_in_one_launch():
    """vt_plane_scatter_mean_multi_fwd / _bwd against the per-plane calls: the stacked planes are torch.cat of the three, the gradient
    their sum in plane order -- the same bits; through the autograd Function as the hand encoder calls it."""
    from vtaco_amd import ops
    from vtaco_amd.encoder.pointnet import _ScatterMeanPlanes
    a, _ = load_golden("g10_hand.npz")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    for p in (T(a["p"]), 0.001 * torch.randn(2, 700, 3, generator=g), 0.5 * (torch.rand(3, 1237, 3, generator=g) - 0.5)):
        p = p.to(dev)
        B, N = p.shape[0], p.shape[1]
        feat = torch.randn(B, N, 32, generator=g).to(dev)
        for planes in (("xz", "xy", "yz"), ("yz", "xz")):
            pis = ops.plane_indices(p, 32, 0.1, planes)
            singles = [ops.PlaneIndex(p, 32, 0.1, k) for k in planes]
            want = torch.cat([ops.plane_scatter_mean_fwd(feat, pi) for pi in singles], dim=0)
            got = ops.plane_scatter_mean_multi_fwd(feat, pis)
            assert torch.equal(got, want), (tuple(p.shape), planes)
            go = torch.randn(want.shape, generator=g).to(dev)
            gs = [ops.plane_scatter_mean_bwd(go[i * B:(i + 1) * B], pi, 32) for i, pi in enumerate(singles)]
            ref_g = gs[0] + gs[1] if len(gs) == 2 else (gs[0] + gs[1]) + gs[2]
            assert torch.equal(ops.plane_scatter_mean_multi_bwd(go, pis, 32), ref_g), (tuple(p.shape), planes)
            f2 = feat.clone().requires_grad_()
            _ScatterMeanPlanes.apply(f2, pis).backward(go)
            assert torch.equal(f2.grad, ref_g)
    assert ops.plane_group(singles) is None                        # separately built indices do not share a buffer: the per-plane path


def test_multi_plane_entries_refuse_bad_arguments():
    """The C ABI's checks of the round-6 plane entries: unknown plane ids, more partitions than the kernels take, null tables."""
    import ctypes
    from vtaco_amd import _lib, ops
    from vtaco_amd._lib import VtError
    dev = torch.device("cuda:0")
    p = torch.rand(1, 64, 3, device=dev) - 0.5
    with pytest.raises(VtError):
        ops.plane_indices(p, 32, 0.1, ("xz", "zx"))
    lib = _lib.load()
    buf = torch.empty((4, 1, 64), dtype=torch.int32, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    bad = (ctypes.c_int * 2)(0, 3)
    assert lib.vt_plane_build_multi(ptr(p), 1, 64, 32, 0.1, 2, bad, ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), ptr(buf[3]), None) != 0
    assert lib.vt_plane_build_multi(ptr(p), 1, 64, 32, 0.1, 4, bad, ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), ptr(buf[3]), None) != 0
    feat = torch.randn(1, 64, 32, device=dev)
    pis = ops.plane_indices(p, 32, 0.1, ("xz", "xy", "yz"))
    with pytest.raises(VtError):
        ops.voxel_pool_max_sum_fwd(feat, pis + pis[:2])              # five partitions
    out = torch.empty_like(feat)
    assert lib.vt_voxel_pool_max_sum_fwd(ptr(feat), 3, None, None, None, 1, 64, 32, ptr(out), None, None) != 0
    torch.cuda.synchronize()


def test_pool_over_the_three_planes_in_one_launch():
    """vt_voxel_pool_max_sum_fwd / _bwd (the hand encoder's `c += pooled` over xz, xy, yz in one launch each way) against three
    vt_voxel_pool_max_fwd / _bwd calls summed in the same order: the same bits for the values and the first arg-maxima (ties included:
    a block of duplicated rows), the same routing and sums for the gradients; the oracle's values; a degenerate cloud (every point in
    one cell) as well."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    from vtaco_amd.encoder.pointnet import _PoolMaxSum
    a, _ = load_golden("g10_hand.npz")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(18)
    for case in ("golden", "one cell"):
        p = T(a["p"]) if case == "golden" else 0.001 * torch.randn(2, 700, 3, generator=g)
        B, N = p.shape[0], p.shape[1]
        feat = torch.randn(B, N, 32, generator=g)
        feat[:, 100:140] = feat[:, 60:100]                          # exact ties between points that share cells with their neighbours
        pis = [ops.PlaneIndex(p.to(dev), 32, 0.1, k) for k in PLANES]
        fd = feat.to(dev)
        out, args = ops.voxel_pool_max_sum_fwd(fd, pis)
        singles = [ops.voxel_pool_max_fwd(fd, pi) for pi in pis]
        want = (singles[0][0] + singles[1][0]) + singles[2][0]
        assert torch.equal(out, want)
        for k in range(3):
            assert torch.equal(args[k], singles[k][1]), (case, k)
        assert torch.equal(ops.voxel_pool_max_sum_fwd(fd, pis, want_argmax=False)[0], out)
        go = torch.randn(B, N, 32, generator=g).to(dev)
        gsum = ops.voxel_pool_max_sum_bwd(go, args, pis)
        gs = [ops.voxel_pool_max_bwd(go, singles[k][1], pis[k]) for k in range(3)]
        # (cells of more than 32 points: the single-partition kernel sums them cooperatively, in another order)
        ref_g = (gs[0] + gs[1]) + gs[2]
        assert maxdiff(gsum, ref_g) <= 1e-5 * float(ref_g.abs().max()), case
        assert torch.equal(gsum == 0, ref_g == 0)                   # the same routing: only arg-max points receive gradient
        if case == "golden":                                        # the oracle: values and, through autograd, the gradient's routing
            fr = feat.clone().requires_grad_()
            ref = sum(orc.segment_pool_max(fr, T(a["idx_" + k]).long()) for k in PLANES)
            assert maxdiff(out, ref.detach()) <= 1e-6
        f2 = fd.clone().requires_grad_()
        _PoolMaxSum.apply(f2, pis).backward(go)
        assert torch.equal(f2.grad, gsum)


@pytest.mark.parametrize("center_idx", [9, None, 0])
def test_mano_layer_kernel(center_idx, tmp_path):
    from oracle import vtaco_oracle as orc
    from vtaco_amd.encoder.manolayer import ManoLayer
    a, _ = load_golden("g10_hand.npz")
    asset = synth_mano.make_asset(0)
    synth_mano.write_pkl(asset, str(tmp_path))
    dev = torch.device("cuda:0")
    layer = ManoLayer(**dict(MANO_KW, mano_root=str(tmp_path), center_idx=center_idx)).to(dev)
    pose = T(a["pose"])
    with torch.no_grad():
        v, j = layer(pose.to(dev))
    if center_idx == 9:                                            # the reference's own outputs
        assert maxdiff(v, a["pose_verts"]) <= 1e-6 and maxdiff(j, a["pose_joints"]) <= 1e-6
    rv, rj = orc.mano_forward(synth_mano.as_model(asset), pose, center_idx)
    assert maxdiff(v, rv) <= 1e-6 and maxdiff(j, rj) <= 1e-6
    # seeded batch incl. B not a multiple of anything, large angles
    big = torch.randn(37, 48, generator=torch.Generator().manual_seed(2)) * 2.0
    with torch.no_grad():
        v, j = layer(big.to(dev))
    rv, rj = orc.mano_forward(synth_mano.as_model(asset), big, center_idx)
    assert maxdiff(v, rv) <= 2e-6 and maxdiff(j, rj) <= 2e-6
    v0, j0 = layer(torch.zeros(0, 48, device=dev))                 # empty batch
    assert v0.shape == (0, 778, 3) and j0.shape == (0, 21, 3)
    # the differentiable host form is the same function, and its gradient matches the oracle's autograd
    pg = pose.clone().to(dev).requires_grad_(True)
    vt, jt = layer(pg)
    assert vt.requires_grad
    assert maxdiff(vt.detach(), orc.mano_forward(synth_mano.as_model(asset), pose, center_idx)[0]) <= 1e-6
    w = torch.randn(4, 778, 3, generator=torch.Generator().manual_seed(3))
    (vt * w.to(dev)).sum().backward()
    pr = pose.clone().requires_grad_(True)
    (orc.mano_forward(synth_mano.as_model(asset), pr, center_idx)[0] * w).sum().backward()
    assert maxdiff(pg.grad, pr.grad) <= 1e-4 * float(pr.grad.abs().max())
    # vt_mano_bwd with verts AND joints gradients (tips, centring), B = 37, large angles, against the oracle's autograd in float64
    gen = torch.Generator().manual_seed(4)
    wv, wj = torch.randn(37, 778, 3, generator=gen), torch.randn(37, 21, 3, generator=gen) * 20.0
    pg = big.clone().to(dev).requires_grad_(True)
    vt, jt = layer(pg)
    ((vt * wv.to(dev)).sum() + (jt * wj.to(dev)).sum()).backward()
    pr = big.clone().double().requires_grad_(True)
    model64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in synth_mano.as_model(asset).items()}
    rv, rj = orc.mano_forward(model64, pr, center_idx)
    ((rv * wv.double()).sum() + (rj * wj.double()).sum()).backward()
    assert maxdiff(pg.grad, pr.grad.float()) <= 2e-5 * float(pr.grad.abs().max())
    again = big.clone().to(dev).requires_grad_(True)
    v2, j2 = layer(again)
    ((v2 * wv.to(dev)).sum() + (j2 * wj.to(dev)).sum()).backward()
    assert torch.equal(again.grad, pg.grad)                        # fixed summation order
    # joints only (d verts absent)
    pj = big.clone().to(dev).requires_grad_(True)
    (layer(pj)[1] * wj.to(dev)).sum().backward()
    pr2 = big.clone().double().requires_grad_(True)
    (orc.mano_forward(model64, pr2, center_idx)[1] * wj.double()).sum().backward()
    assert maxdiff(pj.grad, pr2.grad.float()) <= 2e-5 * float(pr2.grad.abs().max())


@pytest.mark.parametrize("center_idx", [9, 12])
def test_mano_layer_left_hand(center_idx, tmp_path):
    """side='left' (manolayer.py:112-118, 327-330: MANO_LEFT.pkl, middle-finger tip = vertex 445) on the kernels: forward and backward
    against the oracle with the left tips, centred on a chain joint and on the tip that differs (joint 12 of the output order)."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd.encoder.manolayer import ManoLayer
    asset = synth_mano.make_asset(3)
    synth_mano.write_pkl(asset, str(tmp_path), side="left")
    dev = torch.device("cuda:0")
    layer = ManoLayer(**dict(MANO_KW, mano_root=str(tmp_path), center_idx=center_idx, side="left")).to(dev)
    model = synth_mano.as_model(asset)
    pose = torch.randn(9, 48, generator=torch.Generator().manual_seed(12)) * 0.8
    with torch.no_grad():
        v, j = layer(pose.to(dev))
    rv, rj = orc.mano_forward(model, pose, center_idx, side="left")
    assert maxdiff(v, rv) <= 2e-6 and maxdiff(j, rj) <= 2e-6
    rv_r, rj_r = orc.mano_forward(model, pose, center_idx, side="right")
    assert maxdiff(rj, rj_r) > 1e-4                                  # the two tip tables do differ on this model
    gen = torch.Generator().manual_seed(13)
    wv, wj = torch.randn(9, 778, 3, generator=gen), torch.randn(9, 21, 3, generator=gen) * 5.0
    pg = pose.clone().to(dev).requires_grad_(True)
    vt, jt = layer(pg)
    ((vt * wv.to(dev)).sum() + (jt * wj.to(dev)).sum()).backward()
    pr = pose.clone().double().requires_grad_(True)
    model64 = {k: (t.double() if torch.is_tensor(t) and t.is_floating_point() else t) for k, t in model.items()}
    r64 = orc.mano_forward(model64, pr, center_idx, side="left")
    ((r64[0] * wv.double()).sum() + (r64[1] * wj.double()).sum()).backward()
    assert maxdiff(pg.grad, pr.grad.float()) <= 2e-5 * float(pr.grad.abs().max())


def test_pca_pose_space_and_refused_arguments(tmp_path):
    from oracle import vtaco_oracle as orc
    from vtaco_amd._lib import VtError
    from vtaco_amd.encoder.manolayer import ManoLayer
    asset = synth_mano.make_asset(0)
    synth_mano.write_pkl(asset, str(tmp_path))
    dev = torch.device("cuda:0")
    layer = ManoLayer(**dict(MANO_KW, mano_root=str(tmp_path), use_pca=True, ncomps=6)).to(dev)
    coeffs = torch.randn(3, 9, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        v, _ = layer(coeffs.to(dev))
    comps = T(asset["hands_components"][:6].astype(np.float32))
    pose = torch.cat([coeffs[:, :3], coeffs[:, 3:].mm(comps)], dim=1)          # manolayer.py:186-188
    assert maxdiff(v, orc.mano_forward(synth_mano.as_model(asset), pose)[0]) <= 1e-6
    with pytest.raises(VtError):
        layer(coeffs.to(dev), th_trans=torch.zeros(3, 3, device=dev))
    with pytest.raises(VtError):
        layer(torch.zeros(3, 5, device=dev))


def test_hand_encoder_end_to_end_golden_and_training_gradients(tmp_path):
    from oracle import vtaco_oracle as orc
    from vtaco_amd.encoder import encoder_dict
    a, sd = load_golden("g10_hand.npz")
    asset = synth_mano.make_asset(0)
    synth_mano.write_pkl(asset, str(tmp_path))
    dev = torch.device("cuda:0")
    enc = encoder_dict["pointnet_local_pool"](
        dim=3, c_dim=32, padding=0.1, hidden_dim=32, plane_type=["xz", "xy", "yz"], plane_resolution=32, unet=True,
        unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=16), out_mano=True, out_dim=51,
        manolayer_kwargs=dict(MANO_KW, mano_root=str(tmp_path)))
    enc.load_state_dict(sd, strict=False)                          # MANO buffers come from the asset
    enc = enc.to(dev).eval()
    p = T(a["p"]).to(dev)
    with torch.no_grad():
        out = enc(p)
        enc.out_mano = False
        planes = enc(p)
        enc.out_mano = True
    for k in PLANES:
        assert maxdiff(planes[k], a["plane_" + k]) <= 1e-4
    assert maxdiff(out["mano_param"], a["mano_param"]) <= 1e-4
    assert maxdiff(out["mano_verts"], a["mano_verts"]) <= 1e-4
    assert maxdiff(out["mano_joints"], a["mano_joints"]) <= 1e-4
    assert torch.equal(out["mano_faces"].cpu(), T(a["mano_faces"]).long())
    # training: loss_mano + loss_pc (training.py:493-494) backward through MANO layer, U-Net, HIP scatter / pool kernels
    g = torch.Generator().manual_seed(5)
    mano_gt, pc_gt = torch.randn(2, 51, generator=g) * 0.3, torch.randn(2, 778, 3, generator=g) * 0.05
    enc.train()
    enc.zero_grad()
    out = enc(p)
    loss = torch.nn.functional.mse_loss(out["mano_param"], mano_gt.to(dev)) + \
        torch.nn.functional.mse_loss(out["mano_verts"], pc_gt.to(dev))
    loss.backward()
    # reference gradients: autograd of the oracle on the CPU
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ro = orc.hand_encoder_forward(leaves, synth_mano.as_model(asset), T(a["p"]), 32)
    rl = torch.nn.functional.mse_loss(ro["mano_param"], mano_gt) + torch.nn.functional.mse_loss(ro["mano_verts"], pc_gt)
    rl.backward()
    assert abs(float(loss.detach()) - float(rl.detach())) <= 1e-5 * max(1.0, abs(float(rl.detach())))
    checked = 0
    for name, prm in enc.named_parameters():
        ref = leaves[name].grad
        assert ref is not None, name
        scale = float(ref.abs().max())
        assert maxdiff(prm.grad, ref) <= 2e-3 * scale + 1e-7, name
        checked += 1
    assert checked == len(sd)


def _hand_encoder(tmp_path, sd, dev):
    from vtaco_amd.encoder import encoder_dict
    synth_mano.write_pkl(synth_mano.make_asset(0), str(tmp_path))
    enc = encoder_dict["pointnet_local_pool"](
        dim=3, c_dim=32, padding=0.1, hidden_dim=32, plane_type=["xz", "xy", "yz"], plane_resolution=32, unet=True,
        unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=16), out_mano=True, out_dim=51,
        manolayer_kwargs=dict(MANO_KW, mano_root=str(tmp_path)))
    enc.load_state_dict(sd, strict=False)
    return enc.to(dev)


def test_generator_hand_mesh_matches_the_reference_generator(tmp_path):
    from vtaco_amd.conv_onet import models
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd._lib import VtError
    a, sd = load_golden("g10_hand.npz")
    dev = torch.device("cuda:0")
    model = models.ConvolutionalOccupancyNetwork(None, None, _hand_encoder(tmp_path, sd, dev), device=dev)
    gen = Generator3D(model, device=dev)
    mesh = gen.generate_hand_mesh({"inputs": T(a["p"])[:1], "inputs.pc_ply": T(a["pc_ply"])})
    assert mesh.vertices.dtype == torch.float64 and mesh.vertices.shape == (778, 3)
    assert maxdiff(mesh.vertices, a["hand_mesh_verts"]) <= 2e-4           # /(2m) with m ~ 0.5 scales the 1e-4 encoder bar
    assert torch.equal(mesh.faces.cpu(), T(a["hand_mesh_faces"]).long())
    with pytest.raises(VtError, match="one scene"):
        gen.generate_hand_mesh({"inputs": T(a["p"]), "inputs.pc_ply": T(a["pc_ply"])})
    # the MANO layer alone through the container (models/__init__.py:104-112)
    out = model.encode_hand_mano(T(a["pose"]).to(dev))
    assert maxdiff(out["mano_verts"], a["pose_verts"]) <= 1e-6 and out["mano_faces"].shape == (1538, 3)


def test_trainer_with_hand_encoder_on_synthetic_dataset(tmp_path):
    """Object branch + hand branch in one Trainer step (training.py:454-500): loss = l1 + loss_mano + loss_pc."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from synth_dataset import make_cfg, make_synthetic_dataset
    from vtaco_amd import data
    from vtaco_amd.config import get_dataset
    from vtaco_amd.conv_onet import config as cfgmod
    dev = torch.device("cuda:0")
    os.makedirs(tmp_path / "ds")
    make_synthetic_dataset(str(tmp_path / "ds"), seed=5)
    synth_mano.write_pkl(synth_mano.make_asset(0), str(tmp_path / "mano"))
    cfg = make_cfg(str(tmp_path / "ds"), points_subsample=128)
    cfg["model"] = {"decoder": "simple_local", "encoder": "pointnet_local_pool", "c_dim": 32,
                    "decoder_kwargs": {"sample_mode": "bilinear", "hidden_size": 32},
                    "encoder_kwargs": {"hidden_dim": 32, "plane_type": "grid", "grid_resolution": 32, "unet3d": True,
                                       "unet3d_kwargs": {"num_levels": 3, "f_maps": 32, "in_channels": 32, "out_channels": 32}},
                    "encoder_hand": "pointnet_local_pool",
                    "encoder_hand_kwargs": {"hidden_dim": 32, "plane_type": ["xz", "xy", "yz"], "plane_resolution": 32,
                                            "unet": True, "unet_kwargs": {"depth": 3, "merge_mode": "concat", "start_filts": 16},
                                            "out_mano": True, "out_dim": 51,
                                            "manolayer_kwargs": dict(MANO_KW, mano_root=str(tmp_path / "mano"))}}
    cfg["test"] = {"threshold": 0.5}
    torch.manual_seed(0)
    model = cfgmod.get_model(cfg, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    trainer = cfgmod.get_trainer(model, opt, cfg, dev)
    np.random.seed(0)
    batch = next(iter(torch.utils.data.DataLoader(get_dataset("train", cfg), batch_size=3, collate_fn=data.collate_remove_none)))
    first = trainer.train_step(batch)
    for _ in range(25):
        last = trainer.train_step(batch)
    assert all(np.isfinite(x) for x in first) and first[1] > 0 and first[2] > 0
    assert last[0] < 0.8 * first[0] and last[1] < first[1], (first, last)
    assert all(p.grad is not None for p in model.encoder_hand.parameters())


def test_trainer_compute_loss_img_matches_the_reference_trainer():
    """Trainer.compute_loss_img (VTacOH step) on stand-in encoders that return the fixture's tensors: with numpy seeded as the
    reference run was, the three losses equal the real reference Trainer's (g11_trainer_img.npz) and decode_img receives the
    same points and tactile features."""
    import os
    import types
    from conftest import GOLDEN
    from vtaco_amd.conv_onet.training import Trainer
    z = np.load(os.path.join(GOLDEN, "g11_trainer_img.npz"))
    dev = torch.device("cuda:0")
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    seen = {}

    class StandIn(object):
        encoder_hand = encoder_img = object()

        def train(self):
            return self

        def encode_inputs(self, inputs):
            return "c"

        def encode_hand_inputs(self, inputs):
            return {"mano_param": t("mano_param"), "mano_verts": t("mano_verts"), "mano_joints": t("mano_joints")}

        def encode_img_inputs(self, imgs):
            return t("c_img")

        def decode_img(self, p_sample, c, c_img_all, **kw):
            seen["p_sample"], seen["c_img_all"] = p_sample, c_img_all
            return types.SimpleNamespace(logits=p_sample.sum(-1) * 0.5 + c_img_all.sum(-1) * 0.1)

    data = {"points": t("p"), "points.occ": t("occ"), "points.mano": t("mano"), "points.pc_hand": t("pc_hand"),
            "points.wrist": t("wrist"), "inputs": torch.zeros(2, 16, 3), "inputs.pc_ply": t("pc_ply"),
            "inputs.img": torch.zeros(2, 5, 3, 8, 6), "inputs.touch_success": t("touch")}
    trainer = Trainer(StandIn(), None, device=dev, num_sample=int(z["num_sample"]), with_img=True)
    state = np.random.get_state()
    try:
        np.random.seed(int(z["seed"]))
        loss, loss_mano, loss_pc = trainer.compute_loss_img(data)
    finally:
        np.random.set_state(state)
    assert torch.equal(seen["p_sample"].cpu(), torch.from_numpy(z["p_sample"]))
    assert torch.equal(seen["c_img_all"].cpu(), torch.from_numpy(z["c_img_all"]))
    for got, ref in zip((loss, loss_mano, loss_pc), z["loss"]):
        assert abs(float(got) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))


def test_trainer_vtacoh_step_on_synthetic_dataset(tmp_path):
    """The whole VTacOH training step on real modules: object encoder + UNet3D, hand encoder + MANO layer, tactile U-Net on the
    five images, forward_img decoder; loss = l1 + loss_mano + loss_pc goes down on a fixed batch and every branch gets gradients."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from synth_dataset import make_cfg, make_synthetic_dataset
    from vtaco_amd import data
    from vtaco_amd.config import get_dataset
    from vtaco_amd.conv_onet import config as cfgmod
    dev = torch.device("cuda:0")
    os.makedirs(tmp_path / "ds")
    make_synthetic_dataset(str(tmp_path / "ds"), seed=6)
    synth_mano.write_pkl(synth_mano.make_asset(0), str(tmp_path / "mano"))
    cfg = make_cfg(str(tmp_path / "ds"), points_subsample=512)
    cfg["data"]["num_sample"] = 256
    cfg["model"] = {"decoder": "simple_local", "encoder": "pointnet_local_pool", "c_dim": 32, "with_img": True,
                    "decoder_kwargs": {"sample_mode": "bilinear", "hidden_size": 32},
                    "encoder_kwargs": {"hidden_dim": 32, "plane_type": "grid", "grid_resolution": 32, "unet3d": True,
                                       "unet3d_kwargs": {"num_levels": 3, "f_maps": 32, "in_channels": 32, "out_channels": 32}},
                    "encoder_hand": "pointnet_local_pool",
                    "encoder_hand_kwargs": {"hidden_dim": 32, "plane_type": ["xz", "xy", "yz"], "plane_resolution": 32,
                                            "unet": True, "unet_kwargs": {"depth": 3, "merge_mode": "concat", "start_filts": 16},
                                            "out_mano": True, "out_dim": 51,
                                            "manolayer_kwargs": dict(MANO_KW, mano_root=str(tmp_path / "mano"))},
                    "encoder_img": "UNet", "encoder_img_kwargs": {"num_classes": 1, "in_channels": 3, "depth": 3, "start_filts": 8}}
    cfg["test"] = {"threshold": 0.5}
    torch.manual_seed(0)
    model = cfgmod.get_model(cfg, device=dev)
    batch = next(iter(torch.utils.data.DataLoader(get_dataset("train", cfg), batch_size=2, collate_fn=data.collate_remove_none)))
    # the tactile U-Net maps an image to one channel per pixel: its flattened output is the finger's feature, c_dim wide
    H, W = batch["inputs.img"].shape[-2:]
    if H * W != 32:
        batch["inputs.img"] = torch.nn.functional.interpolate(batch["inputs.img"].flatten(0, 1), size=(8, 4)).unflatten(0, (2, 5))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    trainer = cfgmod.get_trainer(model, opt, cfg, dev)
    assert trainer.with_img and trainer.num_sample == 256
    np.random.seed(0)
    first = trainer.train_step(batch)
    for _ in range(15):
        last = trainer.train_step(batch)
    assert all(np.isfinite(x) for x in first) and last[0] < first[0], (first, last)
    for name in ("encoder", "encoder_hand", "decoder"):
        assert all(p.grad is not None for n, p in getattr(model, name).named_parameters() if "fc_out_contact" not in n and "fc_p." not in n), name
    batch["points_iou"], batch["points_iou.occ"] = batch["points"], batch["points.occ"]
    ev = trainer.eval_step(batch)                                   # tactile-aware evaluation: features by the generator's rule
    assert np.isfinite(ev["loss"]) and 0.0 <= ev["iou"] <= 1.0


def test_generator_vtacoh_route_equals_dense_c_img_all(tmp_path):
    """generate_obj_mesh_wnf with with_img=True (generation.py:161-200): the finger-id route (fingertips from the hand encoder,
    vt_tactile_assign, decode by id) against the reference's construction -- the dense c_img_all assembled with the oracle's
    cdist rule -- decoded by the same model: identical logits, identical mesh."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import vtaco_oracle as orc
    from synth_dataset import make_cfg, make_synthetic_dataset
    from vtaco_amd import data as vdata
    from vtaco_amd.config import get_dataset
    from vtaco_amd.conv_onet import config as cfgmod
    dev = torch.device("cuda:0")
    os.makedirs(tmp_path / "ds")
    make_synthetic_dataset(str(tmp_path / "ds"), seed=7)
    synth_mano.write_pkl(synth_mano.make_asset(0), str(tmp_path / "mano"))
    cfg = make_cfg(str(tmp_path / "ds"), points_subsample=64)
    cfg["model"] = {"decoder": "simple_local", "encoder": "pointnet_local_pool", "c_dim": 32, "with_img": True,
                    "decoder_kwargs": {"sample_mode": "bilinear", "hidden_size": 32},
                    "encoder_kwargs": {"hidden_dim": 32, "plane_type": "grid", "grid_resolution": 16, "unet3d": False},
                    "encoder_hand": "pointnet_local_pool",
                    "encoder_hand_kwargs": {"hidden_dim": 32, "plane_type": ["xz", "xy", "yz"], "plane_resolution": 32,
                                            "unet": False, "out_mano": True, "out_dim": 51,
                                            "manolayer_kwargs": dict(MANO_KW, mano_root=str(tmp_path / "mano"))},
                    "encoder_img": "UNet", "encoder_img_kwargs": {"num_classes": 1, "in_channels": 3, "depth": 2, "start_filts": 8}}
    cfg["test"] = {"threshold": 0.5}
    cfg["generation"] = {"resolution_0": 8, "upsampling_steps": 0}
    torch.manual_seed(1)
    model = cfgmod.get_model(cfg, device=dev)
    for blk in list(model.decoder.blocks) + list(model.encoder.blocks):
        torch.nn.init.normal_(blk.fc_1.weight, 0, 0.1)
    gen = cfgmod.get_generator(model, cfg, dev)
    gen.decode_precision = "f32"
    batch = next(iter(torch.utils.data.DataLoader(get_dataset("test", cfg), batch_size=1, collate_fn=vdata.collate_remove_none)))
    batch["inputs.img"] = torch.nn.functional.interpolate(batch["inputs.img"].flatten(0, 1), size=(8, 4)).unflatten(0, (1, 5))
    batch["inputs.touch_success"] = torch.tensor([[True, False, True, True, True]])
    # put the wrist where the first fingertip lands at the lattice centre: the synthetic sample's pose is arbitrary
    from vtaco_amd.common import fingertips_in_object_frame
    with torch.no_grad():
        joints = model.encode_hand_inputs(batch["inputs"].to(dev))["mano_joints"].cpu().numpy()
    cloud = batch["inputs.pc_ply"][0].numpy()
    m = np.max(np.sqrt(np.sum((cloud - cloud.mean(0)) ** 2, axis=1)))
    tips0 = fingertips_in_object_frame(joints, np.zeros((1, 3)), batch["points.wrist"].numpy(), batch["inputs.pc_ply"].numpy())
    batch["points.mano"][0, :3] = torch.from_numpy(-tips0[0, 0] * 2 * m).float()
    mesh = gen.generate_obj_mesh_wnf(batch)
    # the reference's construction, with the oracle's assignment rule on the CPU
    nx = 32
    with torch.no_grad():
        c = model.encode_inputs(batch["inputs"].to(dev))
        c_hand = model.encode_hand_inputs(batch["inputs"].to(dev))
        c_img = model.encode_img_inputs(batch["inputs.img"].to(dev))
    tips = orc.hand_tips_world(c_hand["mano_joints"][0].cpu().numpy(), batch["points.mano"][0, :3].numpy(),
                               batch["points.wrist"][0].numpy(), batch["inputs.pc_ply"][0].numpy())
    pts = 1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
    ids = orc.tactile_assign_nearest(pts.numpy(), tips, batch["inputs.touch_success"][0].numpy(), radius=0.05)
    dense = torch.zeros(1, nx ** 3, 32)
    hit = torch.from_numpy(ids != 255)
    dense[0, hit] = c_img[0].cpu()[torch.from_numpy(ids[ids != 255])]
    with torch.no_grad():
        ref_vals = model.decoder.decode_lattice(c["grid"], nx, c_img=dense.to(dev), precision="f32")
    ref_mesh = gen.extract_mesh(ref_vals.reshape(nx, nx, nx))
    assert torch.equal(mesh.faces, ref_mesh.faces) and torch.equal(mesh.vertices, ref_mesh.vertices)
    assert int(hit.sum()) >= 8                                     # the ball of radius 0.05 around the centred fingertip
    with torch.no_grad():
        plain = model.decoder.decode_lattice(c["grid"], nx, c_img=torch.zeros_like(dense).to(dev), precision="f32")
    assert not torch.equal(plain, ref_vals)                        # ... and its feature really changes the logits
