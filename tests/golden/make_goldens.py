#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference mounted read-only).
It is a harness around the unmodified reference files: it imports them with
stand-ins for the third-party modules that are not installed here
(torch_scatter, pykdtree, pybullet), feeds seeded inputs, and stores inputs +
weights + outputs as small .npz fixtures.  No reference code is stored.

    python tests/golden/make_goldens.py            # torch part (python3.10)
    /opt/conda/bin/python3.9 tests/golden/make_mc_goldens.py   # skimage part

Fixtures:
  g1_decode.npz      LocalDecoder.forward / forward_img / forward_contact, 32^3 lattice, R=16
  g3_pointnet.npz    LocalPoolPointnet (no UNet3D): stage features, fc_c, voxel ids, scatter-mean grid
  g4_unet3d.npz      small UNet3D in/out + full encoder (PointNet + UNet3D) grid
  g5_fusion.npz      TransformerFusion N=256 / N=2048 (eval) + AttentionDecoder.forward_img
  g6_tactile.npz     tactile UNet eval + train-mode BN
  g8_trainstep.npz   one fwd+bwd: loss, gradients of decoder/encoder params, of grid and c_img
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    # torch_scatter==2.0.9 (requirements.txt:30) is not installed: the two ops the
    # path uses, written with torch primitives.
    def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
        index = index.expand_as(src)
        if out is None:
            size = list(src.shape)
            size[dim] = dim_size or int(index.max()) + 1
            out = src.new_zeros(size)
        out = out.scatter_add(dim, index, src)
        cnt = torch.zeros_like(out).scatter_add(dim, index, torch.ones_like(src)).clamp_(min=1)
        return out / cnt

    def scatter_max(src, index, dim=-1, out=None, dim_size=None):
        index = index.expand_as(src)
        size = list(src.shape)
        size[dim] = dim_size or int(index.max()) + 1
        return src.new_zeros(size).scatter_reduce(dim, index, src, reduce="amax", include_self=False), None

    stub("torch_scatter", scatter_mean=scatter_mean, scatter_max=scatter_max)
    stub("pykdtree")
    stub("pykdtree.kdtree", KDTree=object)
    stub("pybullet", computeProjectionMatrixFOV=lambda *a: [0.0] * 16)
    sys.path.insert(0, REF)
    import src  # noqa: F401
    for name, path in [("src.conv_onet", REF + "/src/conv_onet"),
                       ("src.conv_onet.models", REF + "/src/conv_onet/models")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        sys.modules[name] = pkg


def _randomise(module, seed, scale=0.1):
    """Seeded re-randomisation: default init leaves ResnetBlockFC.fc_1.weight at
    zero (layers.py:39), which would make the blocks test nothing."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in module.named_parameters():
            if name.endswith("fc_1.weight") or name.endswith("norm2.weight") or name.endswith("norm2.bias") \
                    or "groupnorm" in name or ".bn." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * scale)
            elif name.endswith(".bias"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)


def _sd(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()
            if "num_batches_tracked" not in k}


def _save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrs)} arrays")


def main():
    _install_stubs()
    from src.common import make_3d_grid
    from src.layers import UNet as TactileUNet
    from src.TransformerFusion import TransformerFusion
    decoder = importlib.import_module("src.conv_onet.models.decoder")
    pointnet = importlib.import_module("src.encoder.pointnet")
    from src.encoder.unet3d import UNet3D
    torch.set_num_threads(8)

    # ---- G1/G2: LocalDecoder on the 32^3 lattice, R=16 -------------------------
    torch.manual_seed(0)
    dec = decoder.LocalDecoder(dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1,
                               sample_mode="bilinear", with_contact=True)
    _randomise(dec, 1)
    g = torch.Generator().manual_seed(2)
    grid = torch.randn(1, 32, 16, 16, 16, generator=g)
    nx = 32
    pts = (1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).unsqueeze(0)
    c_img = torch.randn(1, nx ** 3, 32, generator=g) * (torch.rand(1, nx ** 3, 1, generator=g) < 0.1)
    c_img = c_img.half().float()          # exactly representable in the f16 the fixture stores
    # random (non-lattice) points incl. out-of-range ones that hit both clamps
    prand = (torch.rand(2, 777, 3, generator=g) - 0.5) * 1.3
    grid2 = torch.randn(2, 32, 16, 16, 16, generator=g)
    with torch.no_grad():
        lo = dec(pts, {"grid": grid})
        lo_img = dec.forward_img(pts, {"grid": grid}, c_img)
        lo_c, lo_cc = dec.forward_contact(pts, {"grid": grid})
        lo_rand = dec(prand, {"grid": grid2})
        feat_rand = dec.sample_grid_feature(prand, grid2)
    _save("g1_decode.npz", grid=grid.numpy(), pts=pts.numpy(), c_img=c_img.numpy().astype(np.float16),
          logits=lo.numpy(), logits_img=lo_img.numpy(), logits_contact=lo_c.numpy(),
          logits_contact2=lo_cc.numpy(), prand=prand.numpy(), grid2=grid2.numpy(),
          logits_rand=lo_rand.numpy(), feat_rand=feat_rand.numpy(), **_sd(dec, "sd."))

    # ---- G3: PointNet local pool, no UNet3D -------------------------------------
    torch.manual_seed(3)
    R = 16
    enc = pointnet.LocalPoolPointnet(c_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet3d=False,
                                     grid_resolution=R, plane_type="grid", padding=0.1, n_blocks=5)
    _randomise(enc, 4)
    g = torch.Generator().manual_seed(5)
    d = torch.randn(2, 3000, 3, generator=g)
    p_in = 0.3 * d / d.norm(dim=-1, keepdim=True) + 0.005 * torch.randn(2, 3000, 3, generator=g)
    p_in[:, :40] = (torch.rand(2, 40, 3, generator=g) - 0.5) * 1.4      # outliers -> clamps
    stages = []
    hooks = [blk.register_forward_hook(lambda m, i, o: stages.append(o.detach().clone())) for blk in enc.blocks]
    fcc = []
    hooks.append(enc.fc_c.register_forward_hook(lambda m, i, o: fcc.append(o.detach().clone())))
    with torch.no_grad():
        fea = enc(p_in)["grid"]
    for h in hooks:
        h.remove()
    from src.common import normalize_3d_coordinate, coordinate2index
    idx = coordinate2index(normalize_3d_coordinate(p_in.clone(), padding=0.1), R, coord_type="3d")[:, 0]
    occ = [torch.nonzero(fea[b].abs().sum(0).reshape(-1)).squeeze(1) for b in range(2)]
    _save("g3_pointnet.npz", p=p_in.numpy(), idx=idx.numpy().astype(np.int64),
          **{f"stage{i}": s.numpy() for i, s in enumerate(stages)}, fc_c=fcc[0].numpy(),
          grid=fea.numpy().astype(np.float32), occ0=occ[0].numpy(), occ1=occ[1].numpy(), **_sd(enc, "sd."))

    # ---- G4: small UNet3D and the full encoder ----------------------------------
    torch.manual_seed(6)
    ukw = dict(num_levels=3, f_maps=8, in_channels=32, out_channels=32)
    enc_u = pointnet.LocalPoolPointnet(c_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet3d=True,
                                       unet3d_kwargs=ukw, grid_resolution=R, plane_type="grid",
                                       padding=0.1, n_blocks=5)
    _randomise(enc_u, 7)
    g = torch.Generator().manual_seed(8)
    xin = torch.randn(1, 32, R, R, R, generator=g) * (torch.rand(1, 1, R, R, R, generator=g) < 0.05)
    with torch.no_grad():
        u_out = enc_u.unet3d(xin)
        full = enc_u(p_in[:1])["grid"]
    _save("g4_unet3d.npz", x=xin.numpy(), y=u_out.numpy(), p=p_in[:1].numpy(), grid=full.numpy(),
          **_sd(enc_u, "sd."))

    # ---- G5: TransformerFusion / AttentionDecoder -------------------------------
    torch.manual_seed(9)
    adec = decoder.AttentionDecoder(dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1)
    _randomise(adec, 10)
    adec.eval()
    g = torch.Generator().manual_seed(11)
    out = {}
    for N in (256, 2048):
        ci = torch.randn(2, N, 32, generator=g) * (torch.rand(2, N, 1, generator=g) < 0.3)
        cc = torch.randn(2, N, 32, generator=g)
        with torch.no_grad():
            fo = adec.fuser(ci, 1, cc, 1)
        out[f"c_img{N}"] = ci.numpy()
        out[f"c{N}"] = cc.numpy()
        out[f"fused{N}"] = fo.numpy()
    pa = (torch.rand(2, 256, 3, generator=g) - 0.5) * 1.1
    with torch.no_grad():
        la = adec.forward_img(pa, {"grid": grid2}, torch.from_numpy(out["c_img256"]))
    _save("g5_fusion.npz", p=pa.numpy(), grid=grid2.numpy(), logits=la.numpy(), **out, **_sd(adec, "sd."))

    # ---- G6: tactile UNet ---------------------------------------------------------
    torch.manual_seed(12)
    tun = TactileUNet(num_classes=1, in_channels=3, depth=3, start_filts=8)
    _randomise(tun, 13)
    g = torch.Generator().manual_seed(14)
    with torch.no_grad():
        for m in tun.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    imgs = torch.rand(5, 3, 64, 48, generator=g)
    sd_t = _sd(tun, "sd.")
    tun.eval()
    with torch.no_grad():
        y_eval = tun(imgs)
    tun.train()
    with torch.no_grad():
        y_train = tun(imgs)
    _save("g6_tactile.npz", x=imgs.numpy(), y_eval=y_eval.numpy(), y_train=y_train.numpy(), **sd_t)

    # ---- G8: one forward+backward step --------------------------------------------
    torch.manual_seed(15)
    dec8 = decoder.LocalDecoder(dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1)
    enc8 = pointnet.LocalPoolPointnet(c_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet3d=False,
                                      grid_resolution=R, plane_type="grid", padding=0.1, n_blocks=5)
    _randomise(dec8, 16)
    _randomise(enc8, 17)
    g = torch.Generator().manual_seed(18)
    N = 2048
    pq = (torch.rand(2, N, 3, generator=g) - 0.5) * 1.1
    occ_t = torch.rand(2, N, generator=g)
    ci8 = (torch.randn(2, N, 32, generator=g) * (torch.rand(2, N, 1, generator=g) < 0.2)).requires_grad_(True)
    cgrid = enc8(p_in)["grid"]
    cgrid.retain_grad()
    logits = dec8.forward_img(pq, {"grid": cgrid}, ci8)
    loss = torch.nn.functional.l1_loss(logits, occ_t)
    loss.backward()
    grads = {}
    for pre, mod in (("dec.", dec8), ("enc.", enc8)):
        for n, prm in mod.named_parameters():
            grads["g." + pre + n] = (prm.grad if prm.grad is not None else torch.zeros_like(prm)).numpy()
    # visual-only variant (decoder.forward) for the fc_p gradient
    dec8.zero_grad()
    logits_v = dec8(pq, {"grid": cgrid.detach()})
    loss_v = torch.nn.functional.l1_loss(logits_v, occ_t)
    loss_v.backward()
    gv = {"gv." + n: (prm.grad if prm.grad is not None else torch.zeros_like(prm)).numpy()
          for n, prm in dec8.named_parameters()}
    occv = torch.nonzero(cgrid.grad.abs().sum(1).reshape(2, -1))
    _save("g8_trainstep.npz", p_in=p_in.numpy(), pq=pq.numpy(), occ=occ_t.numpy(),
          c_img=ci8.detach().numpy().astype(np.float32), loss=np.float32(loss.item()),
          loss_v=np.float32(loss_v.item()), logits=logits.detach().numpy(),
          grid_grad_idx=occv.numpy().astype(np.int32),
          grid_grad_val=cgrid.grad.permute(0, 2, 3, 4, 1).reshape(2, -1, 32)[occv[:, 0], occv[:, 1]].numpy(),
          c_img_grad=ci8.grad.numpy(), **grads, **gv, **_sd(dec8, "sd.dec."), **_sd(enc8, "sd.enc."))

    # ---- volumes for the marching-cubes goldens (consumed by make_mc_goldens.py) --
    with torch.no_grad():
        vol = dec(pts, {"grid": grid}).reshape(nx, nx, nx).numpy()
    np.save(os.path.join(OUT, "_mc_vol_logits32.npy"), vol)


if __name__ == "__main__":
    main()

