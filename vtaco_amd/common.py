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


def sensor_pose_inverse(cam_pos_t, cam_rot_t):
    """(linear part of the inverse pose [3,3], translation [3]) of one tactile sensor, float64: camera -> world as the reference does it
    (pc_cam_to_world, common.py:614-640, with the sample rotation + [-pi/2, 0, pi/2]): three hand-written factor matrices (the last
    is not a rotation) composed into a 4 x 4 pose, the linear part of whose INVERSE is applied, plus t."""
    import numpy as np
    ax, ay, az = np.asarray(cam_rot_t, dtype=np.float64) + np.array([-np.pi / 2, 0.0, np.pi / 2])
    m_x = np.array([[np.cos(ax), 0, np.sin(ax)], [0, 1, 0], [-np.sin(ax), 0, np.cos(ax)]])
    m_y = np.array([[np.cos(ay), -np.sin(ay), 0], [np.sin(ay), np.cos(ay), 0], [0, 0, 1]])
    m_z = np.array([[0, 0, 1], [np.cos(az), np.sin(az), 0], [-np.sin(az), np.cos(az), 0]])
    pose = np.zeros((4, 4))
    pose[:3, :3] = m_z @ m_x @ m_y
    pose[:3, 3] = np.asarray(cam_pos_t, dtype=np.float64)
    pose[3, 3] = 1
    return np.linalg.inv(pose)[:3, :3], pose[:3, 3]


def cloud_norm(pc_ply):
    """(centroid [3] f32, scale f32) of norm_pc_1 (common.py:606-612): centre of the object cloud, twice its largest radius."""
    import numpy as np
    cloud = np.asarray(pc_ply, dtype=np.float32)
    centroid = np.mean(cloud, axis=0)
    return centroid, 2 * np.max(np.sqrt(np.sum((cloud - centroid) ** 2, axis=1)))


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
    centroid, scale = cloud_norm(pc_ply)
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
        m_inv, trans = sensor_pose_inverse(cam_pos[t], cam_rot[t])
        world = (m_inv @ cam.T).T + trans
        k = world.shape[0]
        anchors[t, :k] = (world - centroid) / scale
        count[t] = k
    return anchors, count


def contact_clouds_on_device(depths_dev, origin_dev, cam_pos, cam_rot, pc_ply, touch_success, p_sample_host, p_host, num_sample,
                             width=240, height=320, fov=60.0, max_points=128, threshold=1e-4, pack=True):
    """The batch form of ``contact_clouds_from_depth`` + the VTacO step's row assembly (training.py:809-866) with the pixel work on the
    device (vt_contact_scan / vt_contact_points): depths_dev [B,5,H*W] f32 and origin_dev [H*W] f64 on the device; camera poses, the
    object clouds and ``touch_success`` on the host.  Fills ``p_sample_host`` [B,S,3] f32 with the rows that are NOT contact points
    (``randint`` draws from ``p_host`` [B,N,3]) and returns (p_sample device tensor [B,S,3] with the contact rows written by the
    kernel, finger [B,S] int64 host array).  numpy's global generator is consumed exactly as the reference's loop does: per scene the
    sensors' ``randint(count, size=128)`` draws in sensor order, then the scene's fill draw.  ``pack=False`` (the generator): sensor t's
    points at rows t * max_points .. instead of one after the other, no fill rows (``p_host`` unused)."""
    import numpy as np
    import torch

    from . import ops
    from ._lib import VtError
    B, S = depths_dev.shape[0], num_sample
    N = p_host.shape[1] if p_host is not None else 0
    dev = depths_dev.device
    touch = np.asarray(touch_success).astype(bool).reshape(B, 5)
    index, count = ops.contact_scan(depths_dev.reshape(B * 5, -1), origin_dev, torch.from_numpy(touch.astype(np.uint8).reshape(-1)).to(dev), threshold)
    # the host side that does not depend on the counts runs while the scan does
    pose = np.zeros((B * 5, 16))
    for b in range(B):
        centroid, scale = cloud_norm(pc_ply[b])
        for t in range(5):
            if touch[b, t]:
                m_inv, trans = sensor_pose_inverse(cam_pos[b][t], cam_rot[b][t])
                pose[b * 5 + t, :9], pose[b * 5 + t, 9:12] = m_inv.reshape(-1), trans
            pose[b * 5 + t, 12:15], pose[b * 5 + t, 15] = centroid, scale
    cnt = count.cpu().numpy().reshape(B, 5)                          # the one synchronisation of the assembly: 5 B integers
    sel = np.zeros((B * 5, max_points), dtype=np.int32)
    kept = np.zeros(B * 5, dtype=np.int32)
    row0 = np.zeros(B * 5, dtype=np.int32)
    finger = np.full((B, S), -1, dtype=np.int64)
    any_sel = False
    for b in range(B):
        k = 0
        for t in range(5):
            if not touch[b, t]:
                continue
            n = int(cnt[b, t])
            if n > max_points:
                sel[b * 5 + t] = np.random.randint(n, size=max_points)
                n, any_sel = max_points, True
            else:
                sel[b * 5 + t, :n] = np.arange(n)
            if k + n > S:
                raise VtError(f"Trainer: {k + n} contact points do not fit num_sample = {S}")
            if not pack:
                k = t * max_points
            kept[b * 5 + t], row0[b * 5 + t] = n, k
            finger[b, k:k + n] = t
            k += n
        if pack:
            p_sample_host[b, k:] = p_host[b][np.random.randint(N, size=S - k)]
    p_sample = torch.from_numpy(p_sample_host).to(dev, non_blocking=True)
    ops.contact_points(depths_dev.reshape(B * 5, -1), index, torch.from_numpy(sel).to(dev), torch.from_numpy(kept).to(dev),
                       torch.from_numpy(row0).to(dev), torch.from_numpy(pose).to(dev), width, height, fov, max_points, p_sample)
    return p_sample, finger

