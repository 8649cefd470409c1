"""Synthetic scenes for bench.py / smoke (SURVEY.md section 8d inputs)."""
from __future__ import annotations

import os

import torch

from .conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict


def randomise_fc1(module, seed):
    """fc_1.weight ~ N(0, 0.1^2): the default zero init would make every ResNet
    block a no-op ('degenerate random init')."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in module.named_parameters():
            if name.endswith("fc_1.weight"):
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.1)


def sphere_cloud(seed, T=3000, r=0.3, sigma=0.005):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(1, T, 3, generator=g)
    return r * d / d.norm(dim=-1, keepdim=True) + sigma * torch.randn(1, T, 3, generator=g)


def build_scene(seed, device, R=64):
    torch.manual_seed(0)
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1)
    randomise_fc1(dec, 1)
    model = ConvolutionalOccupancyNetwork(dec, None, device=device)
    try:
        from .encoder import encoder_dict
    except ImportError:
        encoder_dict = None
    g = torch.Generator().manual_seed(1000 + seed)
    cloud = sphere_cloud(seed)
    if encoder_dict is not None and 'pointnet_local_pool' in encoder_dict:
        torch.manual_seed(0)
        enc = encoder_dict['pointnet_local_pool'](
            dim=3, c_dim=32, padding=0.1, hidden_dim=32, plane_type='grid', grid_resolution=R,
            unet3d=True, unet3d_kwargs=dict(num_levels=4, f_maps=32, in_channels=32, out_channels=32))
        randomise_fc1(enc, 2)
        model.encoder = enc.to(device)
        with torch.no_grad():
            grid = model.encode_inputs(cloud.to(device))['grid']
    else:
        grid = torch.randn(1, 32, R, R, R, generator=g).to(device)
    from . import ops
    grid = ops.grid_to_channels_last(grid)
    sd_cpu = {k: v.detach().cpu() for k, v in dec.state_dict().items()}

    def c_img(nx):
        gg = torch.Generator().manual_seed(2000 + seed)
        n = nx ** 3
        mask = (torch.rand(1, n, 1, generator=gg) < 0.02).float()
        return (torch.randn(1, 1, 32, generator=gg) * mask).to(device)

    return {"model": model, "grid": grid, "grid_cpu": grid.detach().cpu().contiguous(),
            "sd_decoder_cpu": sd_cpu, "c_img": c_img, "cloud": cloud}


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4: the VTacO training step on synthetic data (shapes of configs/VTacO/VTacO_YCB.yaml and SURVEY.md
# Appendix B: 3000-point clouds, 5 tactile images of 320 x 240, 5 depth images of 76 800 pixels, 2048 query points per scene)
# ---------------------------------------------------------------------------------------------------------------------

def vtaco_cfg(mano_root, num_sample=2048):
    """A config dict with the shipped VTacO model section (20 011 415 parameters); the t2d net stays at its init
    (``pretrained`` false in the factory: there is no checkpoint to load on a synthetic run)."""
    mano_kw = dict(center_idx=9, flat_hand_mean=False, ncomps=45, side="right", mano_root=mano_root, use_pca=False,
                   root_rot_mode="axisang", joint_rot_mode="axisang", robust_rot=False, return_transf=False, return_full_pose=True)
    hand = dict(hidden_dim=32, plane_type=["xz", "xy", "yz"], unet=True, out_mano=True)
    return {"method": "vtaco",
            "data": {"dim": 3, "padding": 0.1, "input_type": "pointcloud", "num_sample": num_sample},
            "test": {"threshold": 0.5},
            "model": {"c_dim": 32, "decoder": "simple_local", "decoder_kwargs": {"sample_mode": "bilinear", "hidden_size": 32},
                      "encoder": "pointnet_local_pool",
                      "encoder_kwargs": {"hidden_dim": 32, "plane_type": "grid", "grid_resolution": 64, "unet3d": True,
                                         "unet3d_kwargs": {"num_levels": 4, "f_maps": 32, "in_channels": 32, "out_channels": 32}},
                      "encoder_hand": "pointnet_local_pool",
                      "encoder_hand_kwargs": dict(hand, plane_resolution=32, out_dim=51, manolayer_kwargs=mano_kw,
                                                  unet_kwargs={"depth": 4, "merge_mode": "concat", "start_filts": 32}),
                      "with_img": True, "with_contact": False, "train_tactile": False,
                      "encoder_img": "Resnet18", "encoder_img_kwargs": {"num_classes": 32},
                      "encoder_t2d": True,
                      "encoder_t2d_kwargs": {"pretrained": False, "encoder_img": "UNet",
                                             "encoder_img_kwargs": {"num_classes": 1, "in_channel": 3, "start_filts": 32, "depth": 3},
                                             "encoder_hand": "pointnet_local_pool",
                                             "encoder_hand_kwargs": dict(hand, c_dim=512, plane_resolution=64, out_dim=30, manolayer_kwargs=mano_kw,
                                                                         unet_kwargs={"depth": 4, "merge_mode": "concat", "start_flits": 32})}}}


def icosphere(radius=0.3, level=3):
    """A closed triangle mesh (the scene's object: winding-number targets).  (verts f64 [V,3], faces i32 [F,3])."""
    import numpy as np
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, dtype=np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(level):
        mid, nf = {}, []

        def midpoint(a, b):
            key = (min(a, b), max(a, b))
            if key not in mid:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                mid[key] = len(v) - 1
            return mid[key]
        for a, b, c in f:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return radius * np.stack(v), np.asarray(f, dtype=np.int32)


def build_train_case(device, rank=0, scenes=8, num_sample=2048, n_points=8192, grad_sync=True, pretrained_t2d=True):
    """(model, trainer, batch, vf_dict) of one rank's share of config 4: the shipped VTacO model built by ``get_model`` on a synthetic
    MANO-format asset, ``Trainer(with_img, encode_t2d)`` (compute_loss_t2d_img: contact clouds from the depth images, winding-number
    targets from the object mesh, tactile features from Resnet18, hand terms), Adam(1e-4), and a seeded synthetic batch of ``scenes``
    scenes.  Every rank builds the same weights (seed 0) and its own data (seed by rank)."""
    import tempfile

    import numpy as np

    from . import synth_mano
    from .conv_onet import config as cfgmod
    from .dist import GradAllReduce
    root = tempfile.mkdtemp(prefix="vt_mano_")
    synth_mano.write_pkl(synth_mano.make_asset(0), root)
    cfg = vtaco_cfg(root, num_sample)
    torch.manual_seed(0)
    model = cfgmod.get_model(cfg, device=device)
    if pretrained_t2d:
        # the shipped configuration (configs/VTacO/VTacO_YCB.yaml:65 `pretrained: True`): the t2d net comes from a checkpoint and is
        # not trained (training.py:749-752: its losses are dropped, its outputs reach the step detached).  The checkpoint here is a
        # synthetic one -- the freshly initialised t2d net saved in the reference's format -- loaded through the factory's own path
        ck = os.path.join(root, "t2d_pretrained.pt")
        torch.save({"model": model.encoder_t2d.state_dict()}, ck)
        cfg["model"]["encoder_t2d_kwargs"].update(pretrained=True, model_file=ck)
        torch.manual_seed(0)
        model = cfgmod.get_model(cfg, device=device)
    randomise_fc1(model.decoder, 1)
    randomise_fc1(model.encoder, 2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    H, W = 320, 240
    depth_origin = np.full(H * W, 0.02, dtype=np.float64)
    trainer = cfgmod.get_trainer(model, opt, cfg, device, depth_origin=depth_origin)
    if grad_sync:
        trainer.grad_sync = GradAllReduce(model.parameters())
    g = torch.Generator().manual_seed(5000 + rank)
    B = scenes
    cloud = torch.cat([sphere_cloud(1000 * rank + i) for i in range(B)])
    # depth images: the flat reading with a pressed-in disc of a few hundred pixels on the touched sensors
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    depth = torch.full((B, 5, H * W), 0.02)
    touch = (torch.rand(B, 5, generator=g) < 0.7)
    touch[:, 0] = True
    for b in range(B):
        for t in range(5):
            if touch[b, t]:
                cy, cx = int(torch.randint(60, H - 60, (1,), generator=g)), int(torch.randint(60, W - 60, (1,), generator=g))
                disc = ((yy - cy) ** 2 + (xx - cx) ** 2) < 12 ** 2
                depth[b, t][disc.reshape(-1)] = 0.02 - 0.002 * torch.rand(int(disc.sum()), generator=g) - 0.0005
    d = torch.randn(B, 5, 3, generator=g)
    verts, faces = icosphere(0.3, 3)
    batch = {"inputs": cloud, "points": (torch.rand(B, n_points, 3, generator=g) - 0.5) * 1.1,
             "points.occ": torch.rand(B, n_points, generator=g),           # read by the visual-only step only (bench.py ms_hip_part)
             "points.mano": torch.randn(B, 51, generator=g) * 0.2, "points.pc_hand": torch.randn(B, 778, 3, generator=g) * 0.05,
             "points.name": ["ico"] * B, "points.cam_pos": (0.32 * d / d.norm(dim=-1, keepdim=True)).double(),
             "points.cam_rot": (torch.rand(B, 5, 3, generator=g) * 2 - 1).double(),
             "inputs.pc_ply": cloud.clone(), "inputs.img": torch.rand(B, 5, 3, H, W, generator=g) / 255.0,
             "inputs.depth": depth, "inputs.touch_success": touch.to(torch.uint8)}
    if torch.cuda.is_available():
        # page-locked host tensors, as a DataLoader(pin_memory=True) hands them over (50 MB of images and depths per step)
        batch = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in batch.items()}
    return model, trainer, batch, {"ico": {"v": verts, "f": faces}}


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs 3 and 5: ONE synthetic scene for Generator3D's tactile routes (generate_obj_mesh_wnf with_img):
# VTacO / t2d (contact clouds from the depth images, generation.py:202-257) and VTacOH (fingertips from the hand encoder's
# MANO joints, generation.py:161-200).  Shapes of configs/VTacO/VTacO_YCB.yaml / VTacOH_YCB.yaml and SURVEY.md Appendix B.
# ---------------------------------------------------------------------------------------------------------------------

def build_tactile_scene(device, variant="vtaco", decoder="simple_local", seed=0):
    """(model, data, depth_origin): the shipped VTacO (``variant="vtaco"``: encoder_t2d present) or VTacOH model section built by
    ``get_model`` on a synthetic MANO-format asset, with ``decoder`` 'simple_local' (tactile concat, what the shipped configs use) or
    'attention_local' (TransformerFusion, BASELINE config 3's decoder; c_dim = hidden = 32), and one scene's sample dictionary with
    the keys the reference generator reads (generation.py:123-143).  The wrist of the VTacOH sample is placed so that the predicted
    fingertips lie on the object (the synthetic pose is arbitrary); the VTacO sample's sensors look at the sphere from 0.32."""
    import tempfile

    import numpy as np

    from . import synth_mano
    from .common import fingertips_in_object_frame
    from .conv_onet import config as cfgmod
    root = tempfile.mkdtemp(prefix="vt_mano_")
    synth_mano.write_pkl(synth_mano.make_asset(0), root)
    cfg = vtaco_cfg(root)
    m = cfg["model"]
    m["decoder"] = decoder
    if variant == "vtacoh":
        m["encoder_t2d"] = False
        m.pop("encoder_t2d_kwargs")
    torch.manual_seed(0)
    model = cfgmod.get_model(cfg, device=device).eval()
    randomise_fc1(model.decoder, 1)
    randomise_fc1(model.encoder, 2)
    H, W = 320, 240
    depth_origin = np.full(H * W, 0.02, dtype=np.float64)
    g = torch.Generator().manual_seed(7000 + seed)
    cloud = sphere_cloud(seed)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    depth = torch.full((1, 5, H * W), 0.02)
    touch = torch.tensor([[1, 1, 0, 1, 1]], dtype=torch.uint8)
    for t in range(5):
        if touch[0, t]:
            cy, cx = int(torch.randint(60, H - 60, (1,), generator=g)), int(torch.randint(60, W - 60, (1,), generator=g))
            disc = ((yy - cy) ** 2 + (xx - cx) ** 2) < 12 ** 2
            depth[0, t][disc.reshape(-1)] = 0.02 - 0.002 * torch.rand(int(disc.sum()), generator=g) - 0.0005
    d = torch.randn(1, 5, 3, generator=g)
    data = {"inputs": cloud, "inputs.pc_ply": cloud.clone(), "inputs.img": torch.rand(1, 5, 3, H, W, generator=g) / 255.0,
            "inputs.depth": depth, "inputs.touch_success": touch,
            "points.cam_pos": (0.32 * d / d.norm(dim=-1, keepdim=True)).double(),
            "points.cam_rot": (torch.rand(1, 5, 3, generator=g) * 2 - 1).double(),
            "points.mano": torch.zeros(1, 51), "points.wrist": torch.zeros(1, 3)}
    if variant == "vtacoh":
        with torch.no_grad():
            joints = model.encode_hand_inputs(cloud.to(device))["mano_joints"].float().cpu().numpy()
        pc = cloud.numpy()
        scale = 2 * np.max(np.sqrt(np.sum((pc[0] - pc[0].mean(0)) ** 2, axis=1)))
        tips0 = fingertips_in_object_frame(joints, np.zeros((1, 3)), data["points.wrist"].numpy(), pc)
        # the tips' centroid goes onto the cloud the encoder sees (the sphere of radius 0.3)
        target = np.array([0.3, 0.0, 0.0])
        data["points.mano"][0, :3] = torch.from_numpy((target - tips0[0].mean(0)) * scale).float()
    return model, data, depth_origin
