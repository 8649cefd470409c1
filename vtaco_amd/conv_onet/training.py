"""Trainer for the object branch of the path (reference src/conv_onet/training.py:35-104, 454-500).

``train_step(data)`` consumes the dictionaries ``vtaco_amd.data`` (or the reference loader) produces:
``inputs`` [B,T,3] -> encode_inputs -> feature grid; ``points`` [B,N,3] -> decode -> logits;
``loss = F.l1_loss(logits, points.occ)`` exactly as the reference's ``compute_loss`` (:487-497).  With a hand
encoder (``model.encoder_hand``) its two other terms are added as the reference does (:493-498):
``loss_mano = mse(mano_param, points.mano)`` and ``loss_pc = mse(mano_verts, points.pc_hand)``; without one
they are reported as 0.  Forward and backward of the whole step (voxeliser, UNet3D, decoder; plane pooling
and scatter of the hand encoder) run on the HIP kernels; the MANO layer and the 2-D U-Net differentiate
through host PyTorch ops.  ``with_img=True`` selects the VTacOH step (``compute_loss_img``, training.py:502-626): fingertips from
the MANO joints, nearest-fingertip assignment of the query points on the device (vt_tactile_assign), the reference's re-sampling
of the query points (same numpy draws under the same seed), ``decode_img`` with the per-point tactile features.
``with_img=True, encode_t2d=True`` selects the VTacO step (``compute_loss_t2d_img``, training.py:757-894): contact clouds from the
depth images as query points, winding-number occupancy targets from the object mesh (vt_winding_number).  The tactile variants (``with_img`` / ``encode_t2d``) need the per-point
contact features the reference assembles with CPU geometry (igl / cdist, training.py:817-866): feed them
at model level (``model.decode_img(p, c, c_img)``) or as finger ids (``vt_tactile_assign``) instead.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.nn import functional as F

from .._lib import VtError
from ..common import fingertips_in_object_frame
from ..eval import compute_iou


# A/B knob: "host" keeps the contact clouds of the VTacO step on the host (numpy, the reference's loop); default: the pixel work on the device
_DEVICE_CLOUDS = os.environ.get("VTACO_CONTACT_CLOUDS", "device") != "host"


class Trainer:
    def __init__(self, model, optimizer, device=None, input_type='pointcloud', vis_dir=None, threshold=0.5,
                 eval_sample=False, num_sample=2048, with_img=False, with_contact=False, train_tactile=False,
                 encode_t2d=False, pretrained_t2d=True, grad_sync=None, depth_origin=None):
        if with_contact and (encode_t2d or train_tactile):
            raise VtError("Trainer: with_contact combines with the visual / VTacOH branches only (as the reference's train_step)")
        self.train_tactile, self.with_contact = train_tactile, with_contact
        self.model, self.optimizer, self.device = model, optimizer, device
        self.input_type, self.threshold = input_type, threshold
        self.with_img, self.num_sample = with_img, num_sample
        self.encode_t2d, self.pretrained_t2d = encode_t2d, pretrained_t2d
        # the tactile sensor's flat depth reading [240*320] (array or path; None = the reference's ./data/VTacO_mesh/depth_origin.txt)
        self.depth_origin = depth_origin
        # data-parallel training (one process per GPU): a callable run between backward and the optimizer step,
        # e.g. vtaco_amd.dist.GradAllReduce(model.parameters()) -- one flat-bucket RCCL all-reduce per step that
        # also covers the parameters a step leaves without gradient (fc_p_img, the contact head)
        self.grad_sync = grad_sync

    def compute_loss(self, data):
        """(loss, loss_mano, loss_pc) -- the last two are the hand branch's (0 without a hand encoder)."""
        p = data.get('points').to(self.device)
        occ = data.get('points.occ').to(self.device)
        inputs = data.get('inputs').to(self.device)
        hand = self._hand_branch(inputs, data) if getattr(self.model, 'encoder_hand', None) is not None else None
        c = self.model.encode_inputs(inputs)
        logits = self.model.decode(p, c).logits
        loss = F.l1_loss(logits, occ)
        if hand is None:
            zero = logits.new_zeros(())
            return loss, zero, zero
        loss_mano, loss_pc = hand()
        return loss + loss_mano + loss_pc, loss_mano, loss_pc

    def _hand_branch(self, inputs, data):
        """The hand branch of a step -- plane PointNet, 2-D U-Net, MANO layer, loss_mano and loss_pc (training.py:476-489) -- queued NOW;
        returns the function that hands over (loss_mano, loss_pc).  In a single process it runs on a side stream: ~120 small launches
        forward and backward that do not depend on the shape branch and hide under its convolutions (autograd runs a node's backward on
        its forward's stream)."""
        dev = self.device

        def run():
            c_hand = self.model.encode_hand_inputs(inputs)
            return (F.mse_loss(c_hand['mano_param'], data.get('points.mano').to(dev).float()),
                    F.mse_loss(c_hand['mano_verts'], data.get('points.pc_hand').to(dev).float()))
        side = self._side_stream(1)
        if side is None:
            return run
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            losses = run()

        def join():
            cur.wait_stream(side)
            for t in losses:
                t.record_stream(cur)
            return losses
        return join

    # -- VTacOH: tactile features concatenated to the points near a fingertip (training.py:502-626) -------------------
    def fingertips(self, mano_joints, mano_gt, wrist_euler, pc_ply):
        """The five fingertip joints of every scene in the object's normalised frame (training.py:543-556), stored as
        float32 like the reference's ``tips_pos`` array."""
        tips = fingertips_in_object_frame(mano_joints.detach().float().cpu().numpy(), mano_gt.detach().cpu().numpy()[:, :3],
                                          wrist_euler.detach().cpu().numpy(), pc_ply.detach().float().cpu().numpy())
        return tips.astype(np.float32)

    def tactile_rows(self, p, tips, touch_success):
        """(rows [B,S], finger [B,S]) on the host: which points a step decodes and whose tactile feature each carries
        (training.py:559-608).  The nearest-fingertip test (cdist + argmin + radius 0.05 + touch success) is vt_tactile_assign
        on the device; what remains is index bookkeeping that consumes numpy's global generator exactly as the reference
        does (``choice`` beyond 512 points per finger, then ``randint`` for the rest -- used, as there, as indices into all
        points), so a seeded run draws the same samples."""
        from .. import ops
        B, N = p.shape[:2]
        S = self.num_sample
        ids = torch.stack([ops.tactile_assign(torch.from_numpy(tips[b]).to(p.device).unsqueeze(1), touch_success[b], 'nearest',
                                              0.05, pts=p[b:b + 1])[0] for b in range(B)]).cpu().numpy()
        picked = []
        for b in range(B):
            idx_b, fin_b = [], []
            for f in range(5):
                idx = np.where(ids[b] == f)[0]
                if idx.shape[0] > 512:
                    idx = idx[np.random.choice(idx.shape[0], 512)]
                idx_b += list(idx)
                fin_b += [f] * len(idx)
            picked.append((idx_b, fin_b))
        rows = np.zeros((B, S), dtype=np.int64)
        finger = np.full((B, S), -1, dtype=np.int64)
        everything = np.arange(N)
        for b in range(B):
            idx_b, fin_b = picked[b]
            k = len(idx_b)
            if k > S:
                raise VtError(f"Trainer.compute_loss_img: {k} tactile points do not fit num_sample = {S}")
            rows[b, :k], finger[b, :k] = idx_b, fin_b
            rest = everything[~np.isin(everything, idx_b)]
            rows[b, k:] = np.random.randint(len(rest), size=S - k)
        return rows, finger

    def compute_loss_img(self, data):
        """(loss, loss_mano, loss_pc) of the VTacOH step: object encoder, hand encoder + MANO layer, tactile encoder on the five
        images; the points near a fingertip whose touch succeeded take that finger's feature; LocalDecoder.forward_img."""
        dev = self.device
        p = data.get('points').to(dev)
        occ = data.get('points.occ').to(dev)
        inputs = data.get('inputs').to(dev)
        if getattr(self.model, 'encoder_hand', None) is None or getattr(self.model, 'encoder_img', None) is None:
            raise VtError("Trainer.compute_loss_img needs model.encoder_hand (fingertips) and model.encoder_img (tactile features)")
        c = self.model.encode_inputs(inputs)
        c_hand = self.model.encode_hand_inputs(inputs)
        if 'mano_joints' not in c_hand:
            raise VtError("Trainer.compute_loss_img: the hand encoder has no MANO layer (out_dim <= 30)")
        c_img = self.model.encode_img_inputs(data.get('inputs.img').to(dev))                     # [B,5,C]
        tips = self.fingertips(c_hand['mano_joints'], data.get('points.mano'), data.get('points.wrist'), data.get('inputs.pc_ply'))
        rows, finger = self.tactile_rows(p, tips, data.get('inputs.touch_success').to(dev))
        rows_t, finger_t = torch.from_numpy(rows).to(dev), torch.from_numpy(finger).to(dev)
        p_sample = torch.gather(p, 1, rows_t.unsqueeze(-1).expand(-1, -1, 3))
        occ_new = torch.gather(occ, 1, rows_t)
        feat = torch.gather(c_img, 1, finger_t.clamp(min=0).unsqueeze(-1).expand(-1, -1, c_img.shape[2]))
        c_img_all = feat * (finger_t >= 0).unsqueeze(-1).to(feat.dtype)
        logits = self.model.decode_img(p_sample, c, c_img_all).logits
        loss_l1 = F.l1_loss(logits, occ_new)
        loss_mano = F.mse_loss(c_hand['mano_param'], data.get('points.mano').to(dev).float())
        loss_pc = F.mse_loss(c_hand['mano_verts'], data.get('points.pc_hand').to(dev).float())
        return loss_l1 + loss_mano + loss_pc, loss_mano, loss_pc

    # -- VTacO (t2d): contact clouds from the tactile depth images become query points (training.py:757-894) --------------
    def _depth_origin(self):
        src = self.depth_origin
        if src is None:
            src = "./data/VTacO_mesh/depth_origin.txt"
        if isinstance(src, str):
            import os
            if not os.path.exists(src):
                raise VtError(f"Trainer: the VTacO branch needs the sensor's flat depth reading; {src} not found "
                              "(pass depth_origin=<array or path>)")
            src = np.loadtxt(src)
            self.depth_origin = src
        return np.asarray(src, dtype=np.float64).reshape(-1)

    def _device_mesh(self, vf_dict, name):
        """(verts f32 [V,3], faces i32 [F,3]) of ``vf_dict[name]`` on the device, uploaded once per mesh object (the dataset's
        meshes do not change during a run; an entry whose arrays were replaced is uploaded again)."""
        cache = self.__dict__.setdefault("_mesh_cache", {})
        m = vf_dict[name]
        hit = cache.get(name)
        if hit is None or hit[0] is not m['v'] or hit[1] is not m['f']:
            v = torch.as_tensor(np.ascontiguousarray(m['v'])).float().to(self.device).contiguous()
            f = torch.as_tensor(np.ascontiguousarray(np.asarray(m['f']).astype(np.int32))).to(self.device).contiguous()
            if len(cache) >= 4096:
                cache.pop(next(iter(cache)))
            hit = cache[name] = (m['v'], m['f'], v, f)
        return hit[2], hit[3]

    def _t2d_samples(self, data, vf_dict, normalise_depth):
        """Query points, finger per row and winding-number targets of a VTacO step (training.py:809-866 / 672-733), plus the t2d
        net's outputs and the depth images as the calling variant uses them."""
        from .. import ops
        from ..common import contact_clouds_from_depth
        dev = self.device
        p = data.get('points')                                        # only its shape and a host view are used here
        B, N = p.shape[:2]
        S = self.num_sample
        inputs = data.get('inputs').to(dev)
        imgs = data.get('inputs.img').to(dev)
        # the depth images stay where the loader left them (host) for the contact clouds: what the device needs of them is the
        # depth loss's target, which only a t2d net under training has -- 12 MB up and down per step otherwise
        depths_src = data.get('inputs.depth')
        need_dev = normalise_depth or not self.pretrained_t2d
        depths = depths_src.to(dev, non_blocking=True).float() if need_dev else None
        if normalise_depth:
            depths = (depths - depths.min()) / (depths.max() - depths.min())
        cam_pos = data.get('points.cam_pos').reshape(B, 5, 3)
        cam_rot = data.get('points.cam_rot').reshape(B, 5, 3)
        if self.pretrained_t2d:
            # the reference runs the t2d net here in every case (training.py:776-778) and then uses its two outputs only in the
            # depth / digit-pose losses, which a pretrained (frozen) t2d net does not have (:887-891): skipped -- 40 U-Net passes
            pred_depth, c_hand_d = None, {'mano_param': None}
        else:
            pred_depth, c_hand_d = self.model.encode_t2d(inputs, imgs)
        origin = self._depth_origin()
        p_host = data.get('points').detach().float().cpu().numpy()      # (a host tensor as the loader hands it over: no copy)
        pc_ply = data.get('inputs.pc_ply').float().cpu().numpy()
        touch = data.get('inputs.touch_success').cpu().numpy()
        p_sample = np.zeros((B, S, 3), dtype=np.float32)
        if _DEVICE_CLOUDS:
            # threshold + np.where + unprojection + pose of the 5 B depth images on the device (vt_contact_scan / vt_contact_points);
            # the randint draws and the 4 x 4 pose inverses stay on the host, in the reference's order
            from ..common import contact_clouds_on_device
            dd = depths if depths is not None else depths_src.to(dev, non_blocking=True).float()
            if getattr(self, "_origin_dev", None) is None or self._origin_dev[0] is not self.depth_origin:
                self._origin_dev = (self.depth_origin, torch.from_numpy(origin).to(dev))
            p_sample_t, finger = contact_clouds_on_device(dd, self._origin_dev[1], cam_pos.cpu().numpy(), cam_rot.cpu().numpy(), pc_ply, touch,
                                                          p_sample, p_host, S)
        else:
            depths_host = (depths if normalise_depth else depths_src).detach().float().cpu().numpy()
            finger = np.full((B, S), -1, dtype=np.int64)
            for b in range(B):
                anchors, count = contact_clouds_from_depth(depths_host[b], origin, cam_pos[b].cpu().numpy(), cam_rot[b].cpu().numpy(),
                                                           pc_ply[b], touch[b])
                k = 0
                for t in range(5):
                    n = int(count[t])
                    if touch[b][t]:
                        if k + n > S:
                            raise VtError(f"Trainer: {k + n} contact points do not fit num_sample = {S}")
                        p_sample[b, k:k + n] = anchors[t, :n].astype(np.float32)
                        finger[b, k:k + n] = t
                        k += n
                p_sample[b, k:] = p_host[b][np.random.randint(N, size=S - k)]
            p_sample_t = torch.from_numpy(p_sample).to(dev)
        names = data.get('points.name')
        occ_new = ops.winding_number_scenes([self._device_mesh(vf_dict, names[b]) for b in range(B)], p_sample_t)    # one launch
        cam_info = torch.cat((cam_pos.reshape(B, -1), cam_rot.reshape(B, -1)), dim=1).to(dev).float()
        return {'inputs': inputs, 'imgs': imgs, 'p_sample': p_sample_t, 'finger': torch.from_numpy(finger).to(dev), 'occ': occ_new,
                'pred_depth': pred_depth, 'digit': c_hand_d['mano_param'], 'depths': depths, 'cam_info': cam_info}

    def _side_stream(self, which=0):
        """Side stream 0 (the tactile feature encoder of the VTacO step) or 1 (the hand branch), or None: VTACO_TRAIN_OVERLAP=0, a CPU
        device, or gradient synchronisation over more than one process (GradAllReduce's buckets mix the encoders' parameters and are launched from
        whichever hook fires last: the multi-process path keeps the single stream it was verified on)."""
        import os
        syncing = self.grad_sync is not None and getattr(self.grad_sync, "_active", lambda: True)()    # (a sync object without peers is idle)
        if os.environ.get("VTACO_TRAIN_OVERLAP", "1") == "0" or syncing or torch.device(self.device).type != "cuda":
            return None
        if getattr(self, "_sides", None) is None:
            self._sides = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        return self._sides[which]

    def _t2d_losses(self, data, s, logits, depth_target, hand=None):
        loss_mano, loss_pc = (hand or self._hand_branch(s['inputs'], data))()
        loss = F.l1_loss(logits, s['occ']) + loss_mano + loss_pc
        if not self.pretrained_t2d:
            loss = loss + F.l1_loss(s['pred_depth'], depth_target) + F.mse_loss(s['digit'], s['cam_info'])
        return loss, loss_mano, loss_pc

    def compute_loss_t2d_img(self, data, vf_dict):
        """(loss, loss_mano, loss_pc) of the VTacO step.  Per scene the contact clouds of the successful touches
        (vtaco_amd.common.contact_clouds_from_depth: the sample's depth images against the sensor's flat reading, the reference's
        numpy draws) are the first query points and carry their finger's tactile feature; the rest are ``randint`` draws from
        the scene's points and carry ONES, as in the reference; every row's occupancy target is the winding number of the
        scene's mesh ``vf_dict[name]`` -- vt_winding_number, the exact sum where the reference calls libigl's fast
        approximation.  With ``pretrained_t2d=False`` the depth and digit-pose losses of the t2d net are added (:887-891)."""
        s = self._t2d_samples(data, vf_dict, normalise_depth=False)
        hand = self._hand_branch(s['inputs'], data)

        def tactile():
            c_img = self.model.encode_img_inputs(s['imgs'])                                 # [B,5,C]
            feat = torch.gather(c_img, 1, s['finger'].clamp(min=0).unsqueeze(-1).expand(-1, -1, c_img.shape[2]))
            return torch.where((s['finger'] >= 0).unsqueeze(-1), feat, torch.ones_like(feat))    # ones where there is no touch
        side = self._side_stream()
        if side is not None:
            # the tactile feature encoder (Resnet18 over the step's 5 B images: the framework's / MIOpen's kernels) and the shape encoder
            # (this repository's kernels) do not depend on each other: the first on a side stream, forward AND backward (autograd runs a
            # node's backward on its forward's stream and orders the streams where gradients cross), joined in front of the decoder
            cur = torch.cuda.current_stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                c_img_all = tactile()
            c = self.model.encode_inputs(s['inputs'])
            cur.wait_stream(side)
            c_img_all.record_stream(cur)
        else:
            c_img_all = tactile()
            c = self.model.encode_inputs(s['inputs'])
        logits = self.model.decode_img(s['p_sample'], c, c_img_all).logits
        d = s['depths']
        return self._t2d_losses(data, s, logits, None if self.pretrained_t2d else (d - d.min()) / (d.max() - d.min()), hand)

    def compute_loss_t2d(self, data, vf_dict):
        """The VTacO step without tactile features (training.py:628-755): the same sample assembly decoded by the plain decoder.
        As in the reference this variant normalises the depth images to [0, 1] BEFORE looking for contact pixels (:643-644), so
        nearly every pixel counts as touched -- reproduced as is."""
        s = self._t2d_samples(data, vf_dict, normalise_depth=True)
        hand = self._hand_branch(s['inputs'], data)
        c = self.model.encode_inputs(s['inputs'])
        logits = self.model.decode(s['p_sample'], c).logits
        return self._t2d_losses(data, s, logits, s['depths'], hand)

    def compute_loss_contact(self, data):
        """(loss, loss_mano, loss_pc, loss_contact) with the decoder's contact head (training.py:896-948): L1 on the occupancy
        logits + binary cross-entropy of the contact logits against ``points.contact`` + the hand terms."""
        dev = self.device
        p = data.get('points').to(dev)
        inputs = data.get('inputs').to(dev)
        c = self.model.encode_inputs(inputs)
        p_r, pred_contact = self.model.decode_contact(p, c)
        loss_l1 = F.l1_loss(p_r.logits, data.get('points.occ').to(dev))
        loss_contact = F.binary_cross_entropy_with_logits(pred_contact, data.get('points.contact').to(dev).float(), reduction='mean')
        zero = loss_l1.new_zeros(())
        loss_mano = loss_pc = zero
        if getattr(self.model, 'encoder_hand', None) is not None:
            c_hand = self.model.encode_hand_inputs(inputs)
            loss_mano = F.mse_loss(c_hand['mano_param'], data.get('points.mano').to(dev).float())
            loss_pc = F.mse_loss(c_hand['mano_verts'], data.get('points.pc_hand').to(dev).float())
        return loss_contact + loss_l1 + loss_mano + loss_pc, loss_mano, loss_pc, loss_contact

    def compute_loss_tactile(self, data):
        """(loss, loss_depth, loss_digit) of the t2d net trained on its own (training.py:950-986; the model is the t2d
        ``ConvolutionalOccupancyNetwork``: depth U-Net as encoder_img, digit-pose regressor as encoder_hand): L1 between the
        predicted depth images and the batch-normalised ground truth, MSE between the regressed digit poses and the cameras'.
        ``loss_digit`` is None without a hand encoder."""
        dev = self.device
        inputs = data.get('inputs').to(dev)
        depths = data.get('inputs.depth').to(dev).float()
        B = inputs.shape[0]
        depths = (depths - depths.min()) / (depths.max() - depths.min())
        loss_depth = F.l1_loss(self.model.encode_img_inputs(data.get('inputs.img').to(dev)), depths)
        if getattr(self.model, 'encoder_hand', None) is None:
            return loss_depth, loss_depth, None
        cam_info = torch.cat((data.get('points.cam_pos').reshape(B, -1), data.get('points.cam_rot').reshape(B, -1)), dim=1).to(dev).float()
        loss_digit = F.mse_loss(self.model.encode_hand_inputs(inputs)['mano_param'], cam_info)
        return loss_depth + loss_digit, loss_depth, loss_digit

    def _model_train(self):
        """``self.model.train()`` (training.py: every step) without walking the module tree through ``nn.Module.__setattr__`` each time:
        the same flag on the same modules (the list is taken again every 256 steps and whenever the model object changes) -- 332
        modules cost ~0.5 ms of host time per step the plain way, which shows where the host, not the GPU, paces the step."""
        mods = getattr(self, "_train_mods", None)
        self._train_calls = getattr(self, "_train_calls", 0) + 1
        if mods is None or mods[0] is not self.model or self._train_calls % 256 == 1:
            mods = self._train_mods = list(self.model.modules())
            # a module whose class overrides train() (a frozen sub-net that keeps its BatchNorm in eval mode, ...) must see the call
            self._train_plain = any(type(m).train is not torch.nn.Module.train for m in mods)
        if self._train_plain:
            self.model.train()
            return
        for m in mods:
            m.__dict__["training"] = True

    def train_step(self, data, vf_dict=None):
        self._model_train()
        self.optimizer.zero_grad()
        if self.train_tactile:
            loss, loss_depth, loss_digit = self.compute_loss_tactile(data)
            loss.backward()
            if self.grad_sync is not None:
                self.grad_sync()
            self.optimizer.step()
            return (loss.item(), loss_depth.item(), loss_digit.item()) if loss_digit is not None else (loss.item(), loss_depth.item())
        if self.with_contact:
            loss, loss_mano, loss_pc, loss_contact = self.compute_loss_contact(data)
            loss.backward()
            if self.grad_sync is not None:
                self.grad_sync()
            self.optimizer.step()
            return loss.item(), loss_mano.item(), loss_pc.item(), loss_contact.item()
        if self.encode_t2d:
            if vf_dict is None:
                raise VtError("Trainer.train_step: the VTacO branch needs vf_dict (object meshes by name, vtaco_amd.data.load_mesh_dict)")
            loss, loss_mano, loss_pc = (self.compute_loss_t2d_img if self.with_img else self.compute_loss_t2d)(data, vf_dict)
        else:
            loss, loss_mano, loss_pc = self.compute_loss_img(data) if self.with_img else self.compute_loss(data)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        self.optimizer.step()
        return loss.item(), loss_mano.item(), loss_pc.item()

    def _tactile_features_at(self, data, pts):
        """c_img [B,N,C] for arbitrary query points with the configured tactile branch's assignment rule (the generator's:
        nearest successful fingertip within 0.05 for VTacOH, within 0.015 of a contact point for VTacO), zeros elsewhere."""
        from .. import ops
        from ..common import contact_clouds_from_depth
        dev = self.device
        inputs = data.get('inputs').to(dev)
        B = pts.shape[0]
        c_img = self.model.encode_img_inputs(data.get('inputs.img').to(dev))
        touch = data.get('inputs.touch_success')
        if self.encode_t2d:
            origin = self._depth_origin()
            cam_pos, cam_rot = data.get('points.cam_pos').reshape(B, 5, 3), data.get('points.cam_rot').reshape(B, 5, 3)
            ids = []
            for b in range(B):
                anchors, count = contact_clouds_from_depth(data.get('inputs.depth')[b].float().cpu().numpy(), origin, cam_pos[b].cpu().numpy(),
                                                           cam_rot[b].cpu().numpy(), data.get('inputs.pc_ply')[b].float().cpu().numpy(),
                                                           touch[b].cpu().numpy())
                ids.append(ops.tactile_assign(torch.from_numpy(anchors).float().to(dev), torch.from_numpy((count > 0).astype(np.uint8)).to(dev),
                                              'within', 0.015, pts=pts[b:b + 1], count=torch.from_numpy(count).int().to(dev))[0])
        else:
            c_hand = self.model.encode_hand_inputs(inputs)
            tips = self.fingertips(c_hand['mano_joints'], data.get('points.mano'), data.get('points.wrist'), data.get('inputs.pc_ply'))
            ids = [ops.tactile_assign(torch.from_numpy(tips[b]).to(dev).unsqueeze(1), touch[b].to(dev), 'nearest', 0.05, pts=pts[b:b + 1])[0]
                   for b in range(B)]
        ids = torch.stack(ids).long()
        feat = torch.gather(c_img, 1, ids.clamp(max=4).unsqueeze(-1).expand(-1, -1, c_img.shape[2]))
        return feat * (ids != 255).unsqueeze(-1).to(feat.dtype)

    def eval_step(self, data, vf_dict=None):
        """{'loss', 'iou'}: L1 loss on ``points`` and the reference's IoU (compute_iou: both sides cut at the mean ground-truth
        occupancy) on ``points_iou``; with ``with_img`` the query points carry tactile features assigned by the generator's rule
        (the reference's eval_step re-labels its points with libigl winding numbers instead, training.py:105-452: not mirrored).
        ``train_tactile``: {'loss', 'loss_depth'} of the t2d net (training.py:424-452)."""
        self.model.eval()
        dev = self.device
        with torch.no_grad():
            if self.train_tactile:
                loss, loss_depth, _ = self.compute_loss_tactile(data)
                return {'loss': loss.item(), 'loss_depth': loss_depth.item()}
            inputs = data.get('inputs').to(dev)
            c = self.model.encode_inputs(inputs)

            def logits_at(pts):
                pts = pts.to(dev)
                if self.with_img:
                    return self.model.decode_img(pts, c, self._tactile_features_at(data, pts)).logits
                return self.model.decode(pts, c).logits
            out = {'loss': F.l1_loss(logits_at(data.get('points')), data.get('points.occ').to(dev)).item()}
            if data.get('points_iou') is not None:
                occ_hat = torch.sigmoid(logits_at(data.get('points_iou')))
                occ_iou = data.get('points_iou.occ')
                out['iou'] = float(np.mean(compute_iou(occ_hat.cpu().numpy(), occ_iou.numpy(), self.threshold)))
        return out

    def evaluate(self, val_loader):
        """Mean of eval_step over a loader."""
        sums, n = {}, 0
        for batch in val_loader:
            for k, v in self.eval_step(batch).items():
                sums[k] = sums.get(k, 0.0) + v
            n += 1
        return {k: v / max(n, 1) for k, v in sums.items()}
