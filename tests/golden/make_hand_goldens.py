#!/usr/bin/env python3
"""Golden vectors for the hand branch (g10_hand.npz) from the REAL reference.

Build container only (needs /root/reference).  A harness around the unmodified reference files
(src/encoder/pointnet.py plane mode + out_mano, src/encoder/unet.py, src/encoder/manolayer.py):
stand-ins for the third-party modules that are not installed (torch_scatter, pykdtree, pybullet as in
make_goldens.py; chumpy and cv2 only as far as MANO's loader touches them), seeded inputs, outputs
stored.  The MANO asset is the synthetic one of tests/synth_mano.py (the licensed MANO_RIGHT.pkl is
never stored); when the real asset is present the script also reports, without storing anything, how
far the oracle is from the reference ManoLayer on it.

    python tests/golden/make_hand_goldens.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_goldens as mg          # noqa: E402  (the shared import recipe)
import synth_mano                  # noqa: E402


def _install_mano_stubs():
    class Arr(np.ndarray):
        @property
        def r(self):
            return np.asarray(self)

    def array(x):
        return np.asarray(x, dtype=np.float64).view(Arr)

    class Ch(object):
        pass

    def mat_vec_mult(mtx, vec):
        return array(np.asarray(mtx.dot(np.asarray(vec))).ravel())

    ch = types.ModuleType("chumpy")
    ch.array, ch.Ch = array, Ch
    ch.vstack = lambda xs: array(np.vstack([np.asarray(x) for x in xs]))
    ch.concatenate = lambda xs: array(np.concatenate([np.asarray(x) for x in xs]))
    ch.eye, ch.zeros = (lambda n: array(np.eye(n))), (lambda n: array(np.zeros(n)))
    chch = types.ModuleType("chumpy.ch")
    chch.MatVecMult, chch.Ch = mat_vec_mult, Ch
    ch.ch = chch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = ch, chch

    def cv_rodrigues(v):
        from scipy.spatial.transform import Rotation
        return Rotation.from_rotvec(np.asarray(v, dtype=np.float64).ravel()).as_matrix(), None

    cv2 = types.ModuleType("cv2")
    cv2.Rodrigues = cv_rodrigues
    sys.modules["cv2"] = cv2


def _reference_hand_mesh(out, pc_ply):
    """Generator3D.generate_hand_mesh (generation.py:74-115) of the real reference on given encoder outputs.
    generation.py imports trimesh / skimage (not installed) and reads a dataset file at import: stand-ins
    for the two modules, and np.loadtxt answers that one read with zeros (the array is unused here)."""
    import importlib

    class Trimesh(object):
        def __init__(self, vertices, faces):
            self.vertices, self.faces = np.asarray(vertices), np.asarray(faces)

    tm = types.ModuleType("trimesh")
    tm.Trimesh = Trimesh
    sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.measure")
    sk.measure = skm
    sys.modules.update({"trimesh": tm, "skimage": sk, "skimage.measure": skm})
    loadtxt = np.loadtxt
    np.loadtxt = lambda *a, **k: np.zeros((320, 240))
    try:
        generation = importlib.import_module("src.conv_onet.generation")
    finally:
        np.loadtxt = loadtxt

    class FakeModel(object):
        def to(self, device):
            return self

        def eval(self):
            return self

        def encode_hand_inputs(self, inputs):
            return out

    gen = generation.Generator3D(FakeModel(), device="cpu")
    B = out["mano_param"].shape[0]
    assert B == 1
    data = {"inputs": torch.zeros(1, 4, 3), "inputs.pc_ply": pc_ply, "points.mano": torch.zeros(1, 51),
            "points.wrist": torch.zeros(1, 3)}
    mesh = gen.generate_hand_mesh(data)
    return mesh.vertices, mesh.faces


MANO_KW = dict(center_idx=9, flat_hand_mean=False, ncomps=45, side="right", use_pca=False,
               root_rot_mode="axisang", joint_rot_mode="axisang", robust_rot=False, return_transf=False)


def main():
    mg._install_stubs()
    _install_mano_stubs()
    import importlib
    pointnet = importlib.import_module("src.encoder.pointnet")
    from oracle import vtaco_oracle as orc
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp(prefix="vt_mano_")
    asset = synth_mano.make_asset(0)
    synth_mano.write_pkl(asset, tmp)

    # hand encoder as configs/VTacO/VTacO_YCB.yaml:33-56 builds it (c_dim 32, hidden 32, 3 planes @32, 2-D U-Net, MANO head);
    # the U-Net is narrower / shallower than the config's (depth 3, 16 filters: 0.5 MB of weights instead of 7.7 MB)
    torch.manual_seed(10)
    R = 32
    enc = pointnet.LocalPoolPointnet(c_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet=True,
                                     unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=16),
                                     plane_resolution=R, plane_type=["xz", "xy", "yz"], padding=0.1, n_blocks=5,
                                     out_mano=True, out_dim=51, manolayer_kwargs=dict(MANO_KW, mano_root=tmp))
    mg._randomise(enc, 11)
    enc.eval()
    g = torch.Generator().manual_seed(12)
    d = torch.randn(2, 3000, 3, generator=g)
    p_in = 0.3 * d / d.norm(dim=-1, keepdim=True) + 0.005 * torch.randn(2, 3000, 3, generator=g)
    p_in[:, :40] = (torch.rand(2, 40, 3, generator=g) - 0.5) * 1.4                   # outliers -> both clamps
    fea_m_get = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device                               # the reference calls .to(get_device()): -1 on CPU
    try:
        with torch.no_grad():
            out = enc(p_in)
            # the plane features the head consumed: rerun the encoder body without the head
            enc.out_mano = False
            planes = enc(p_in)
            enc.out_mano = True
    finally:
        torch.Tensor.get_device = fea_m_get
    idx = {k: pointnet.coordinate2index(pointnet.normalize_coordinate(p_in.clone(), plane=k, padding=0.1), R).squeeze(1)
           for k in ("xz", "xy", "yz")}
    # MANO layer alone on seeded poses (incl. a zero pose and a large one)
    pose = torch.randn(4, 48, generator=g) * 0.6
    pose[0] = 0.0
    pose[3] *= 3.0
    with torch.no_grad():
        mv, mj = enc.mano_layer(pose)
    # hand mesh post-processing of the generator on scene 0 (wrist pose un-rotation, normalisation by the object cloud)
    pc_ply = torch.randn(1, 500, 3, generator=g) * 0.2 + 0.1
    hv, hf = _reference_hand_mesh({k: (v[:1] if k != "mano_faces" else v) for k, v in out.items()}, pc_ply)
    sd = {k: v for k, v in mg._sd(enc, "sd.").items() if "mano_layer" not in k}
    mg._save("g10_hand.npz", p=p_in.numpy(),
             idx_xz=idx["xz"].numpy().astype(np.int32), idx_xy=idx["xy"].numpy().astype(np.int32),
             idx_yz=idx["yz"].numpy().astype(np.int32),
             plane_xz=planes["xz"].numpy(), plane_xy=planes["xy"].numpy(), plane_yz=planes["yz"].numpy(),
             mano_param=out["mano_param"].numpy(), mano_verts=out["mano_verts"].numpy(),
             mano_joints=out["mano_joints"].numpy(), mano_faces=out["mano_faces"].numpy().astype(np.int32),
             pose=pose.numpy(), pose_verts=mv.numpy(), pose_joints=mj.numpy(),
             pc_ply=pc_ply.numpy(), hand_mesh_verts=np.asarray(hv, dtype=np.float64), hand_mesh_faces=np.asarray(hf, dtype=np.int32),
             **sd)

    # oracle vs the reference on the synthetic asset
    model = synth_mano.as_model(asset)
    ov, oj = orc.mano_forward(model, pose)
    print("oracle vs reference ManoLayer (synthetic asset): verts %.2e joints %.2e (|verts| up to %.3f)"
          % (float((ov - mv).abs().max()), float((oj - mj).abs().max()), float(mv.abs().max())))
    real = "/root/reference/src/encoder/assets/mano"
    if os.path.exists(os.path.join(real, "MANO_RIGHT.pkl")):
        # the real asset holds one chumpy object (shapedirs): read it chumpy-free, hand the reference a plain copy
        from vtaco_amd.encoder.manolayer import load_mano_pkl
        dd = load_mano_pkl(os.path.join(real, "MANO_RIGHT.pkl"))
        tmp2 = tempfile.mkdtemp(prefix="vt_mano_real_")
        synth_mano.write_pkl(dict(dd, bs_type="lrotmin", bs_style="lbs", hands_coeffs=np.zeros((1, 45))), tmp2)
        ref_layer = pointnet.ManoLayer(**dict(MANO_KW, mano_root=tmp2))
        with torch.no_grad():
            rv, rj = ref_layer(pose)
        ov, oj = orc.mano_forward(synth_mano.as_model(dd), pose)
        print("oracle vs reference ManoLayer (real MANO_RIGHT.pkl, nothing stored): verts %.2e joints %.2e (|verts| up to %.3f)"
              % (float((ov - rv).abs().max()), float((oj - rj).abs().max()), float(rv.abs().max())))


if __name__ == "__main__":
    main()
