"""Host-side coordinate helpers with the reference's names (src/common.py).  The HIP
kernels re-implement this arithmetic in-kernel; these torch versions exist for callers
that want the explicit point tensors (e.g. ``eval_points`` on arbitrary points)."""
from __future__ import annotations

import torch


def make_3d_grid(bb_min, bb_max, shape):
    """Lattice with axis 0 slowest, axis 2 fastest (src/common.py:178-197)."""
    axes = [torch.linspace(bb_min[k], bb_max[k], shape[k]) for k in range(3)]
    mesh = torch.meshgrid(*axes, indexing="ij")
    return torch.stack([m.reshape(-1) for m in mesh], dim=1)


def normalize_3d_coordinate(p, padding=0.1):
    """src/common.py:293-309 (functional; the reference mutates a clone)."""
    q = p / (1 + padding + 10e-4) + 0.5
    q = torch.where(q >= 1, torch.full_like(q, 1 - 10e-4), q)
    return torch.where(q < 0, torch.zeros_like(q), q)


def coordinate2index(x, reso, coord_type='3d'):
    """src/common.py:333-348, returns [B,1,T] int64."""
    xi = (x * reso).long()
    if coord_type == '2d':
        index = xi[:, :, 0] + reso * xi[:, :, 1]
    else:
        index = xi[:, :, 0] + reso * (xi[:, :, 1] + reso * xi[:, :, 2])
    return index[:, None, :]


def rot_from_pyr(roll, pitch, yaw):
    """The reference's wrist-frame convention (R_from_PYR, src/common.py:591-604): a z rotation by ``roll`` applied first,
    then the TRANSPOSED y rotation by ``yaw``, then the transposed x rotation by ``pitch``.  3x3 numpy, host side."""
    import numpy as np
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    about_z = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    about_x_t = np.array([[1, 0, 0], [0, cp, sp], [0, -sp, cp]])
    about_y_t = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
    return about_x_t @ about_y_t @ about_z


def fingertips_in_object_frame(mano_joints, wrist_pos, wrist_euler, pc_ply):
    """The five fingertip joints (MANO joints 4, 8, 12, 16, 20) of every scene in the object's normalised frame
    (training.py:543-556, generation.py:178-184): out of the MANO frame (fixed offset and rotation), out of the wrist
    rotation, plus the wrist position, normalised like the object cloud (norm_pc_1, common.py:606-612).  5 x 3 numbers
    per scene, host-side numpy in the reference's operation order.  [B,21,3], [B,3], [B,3], [B,M,3] -> float64 [B,5,3]."""
    import numpy as np
    joints = np.asarray(mano_joints, dtype=np.float32)[:, [4, 8, 12, 16, 20]]
    fixed = np.linalg.inv(rot_from_pyr(-np.pi / 2, np.pi / 2, 0.0))
    tips = np.empty(joints.shape, dtype=np.float64)
    for b in range(joints.shape[0]):
        t = joints[b] - np.array([0.11, 0.005, 0], dtype=np.float32)
        t = np.linalg.inv(rot_from_pyr(*np.asarray(wrist_euler[b]))) @ (fixed @ t.T)
        t = t.T + np.asarray(wrist_pos[b])
        cloud = np.asarray(pc_ply[b])
        centroid = np.mean(cloud, axis=0)
        m = np.max(np.sqrt(np.sum((cloud - centroid) ** 2, axis=1)))
        tips[b] = (t - centroid) / (2 * m)
    return tips


def contact_clouds_from_depth(depths, depth_origin, cam_pos, cam_rot, pc_ply, touch_success, width=240, height=320, fov=60.0,
                              max_points=128, threshold=1e-4):
    """Contact point clouds of one scene's five tactile sensors in the object's normalised frame -- the VTacO (t2d) rule
    (generation.py:224-244, training.py:817-853): pixels whose depth departs from the sensor's flat reading ``depth_origin``
    by more than ``threshold`` are unprojected through the pinhole model (f = height / (2 tan(fov/2)), principal point at the
    image centre, axes (z, -x, -y)), at most ``max_points`` of them are kept (``np.random.randint`` draws, with replacement,
    from numpy's GLOBAL generator, as the reference: a seeded run picks the same pixels), moved to the world with the sample's
    camera pose and normalised like the object cloud (norm_pc_1).  Host-side numpy on 5 x 76 800 pixels, float64 like the
    reference.  depths [5, H*W]; cam_pos, cam_rot [5,3]; returns (anchors [5,max_points,3] float64, count [5] int)."""
    import math
    import numpy as np
    depths = np.asarray(depths, dtype=np.float32)
    origin = np.asarray(depth_origin, dtype=np.float64).reshape(-1)
    if depths.shape != (5, width * height) or origin.shape[0] != width * height:
        raise ValueError(f"contact_clouds_from_depth: depths {depths.shape} / depth_origin {origin.shape} do not match a "
                         f"{height}x{width} sensor image")
    cloud = np.asarray(pc_ply, dtype=np.float32)
    centroid = np.mean(cloud, axis=0)
    scale = 2 * np.max(np.sqrt(np.sum((cloud - centroid) ** 2, axis=1)))
    f = height / (2 * math.tan(math.radians(fov / 2)))
    px, py = np.meshgrid(np.arange(width), np.arange(height))
    anchors = np.zeros((5, max_points, 3))
    count = np.zeros(5, dtype=np.int64)
    for t in range(5):
        if not bool(touch_success[t]):
            continue
        z = depths[t]
        touched = np.where(np.abs(z - origin) > threshold)[0]
        # unproject the touched pixels only (same arithmetic and dtypes as unprojecting the whole image and selecting)
        zt, pxt, pyt = z[touched], px.reshape(-1)[touched], py.reshape(-1)[touched]
        cam = np.stack([zt, -(pxt - width / 2) * zt / f, -(pyt - height / 2) * zt / f], axis=-1)
        if cam.shape[0] > max_points:
            cam = cam[np.random.randint(cam.shape[0], size=max_points)]
        # camera -> world (pc_cam_to_world, common.py:614-640, with the sample rotation + [-pi/2, 0, pi/2]): the reference composes
        # three hand-written factor matrices (the last is not a rotation) and applies the linear part of the INVERSE pose, plus t
        ax, ay, az = np.asarray(cam_rot[t], dtype=np.float64) + np.array([-np.pi / 2, 0.0, np.pi / 2])
        m_x = np.array([[np.cos(ax), 0, np.sin(ax)], [0, 1, 0], [-np.sin(ax), 0, np.cos(ax)]])
        m_y = np.array([[np.cos(ay), -np.sin(ay), 0], [np.sin(ay), np.cos(ay), 0], [0, 0, 1]])
        m_z = np.array([[0, 0, 1], [np.cos(az), np.sin(az), 0], [-np.sin(az), np.cos(az), 0]])
        pose = np.zeros((4, 4))
        pose[:3, :3] = m_z @ m_x @ m_y
        pose[:3, 3] = np.asarray(cam_pos[t], dtype=np.float64)
        pose[3, 3] = 1
        world = (np.linalg.inv(pose)[:3, :3] @ cam.T).T + pose[:3, 3]
        k = world.shape[0]
        anchors[t, :k] = (world - centroid) / scale
        count[t] = k
    return anchors, count
