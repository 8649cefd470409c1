"""CPU: the oracle (oracle/vtaco_oracle.py) against golden vectors produced by
the real reference (tests/golden/make_goldens.py).  This is what pins it."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub_sd
from oracle import vtaco_oracle as orc

T = torch.from_numpy


def maxdiff(a, b):
    return float((torch.as_tensor(a) - torch.as_tensor(b)).abs().max())


def test_lattice_matches_reference_points():
    a, _ = load_golden("g1_decode.npz")
    pts = 1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (32,) * 3)
    assert torch.equal(pts, T(a["pts"])[0])


def test_local_decoder_forward_variants():
    a, sd = load_golden("g1_decode.npz")
    grid, pts = T(a["grid"]), T(a["pts"])
    assert maxdiff(orc.local_decoder_forward(sd, pts, grid), a["logits"]) <= 2e-5
    c_img = T(a["c_img"].astype(np.float32))
    assert maxdiff(orc.local_decoder_forward_img(sd, pts, grid, c_img), a["logits_img"]) <= 2e-5
    o, oc = orc.local_decoder_forward_contact(sd, pts, grid)
    assert maxdiff(o, a["logits_contact"]) <= 2e-5 and maxdiff(oc, a["logits_contact2"]) <= 2e-5


def _wide_case(tag):
    """One case of g16_decode_wide.npz: (arrays of the case, its state dict in f32, hidden, c_dim, n_blocks, leaky, nx)."""
    a, sd = load_golden("g16_decode_wide.npz")
    arrs = {k[2:]: v for k, v in a.items() if k.startswith(tag + ".")}
    sdc = {k[2:]: v.float() for k, v in sd.items() if k.startswith(tag + ".")}
    hidden, c_dim, nb, leaky, nx, nearest = (int(x) for x in arrs["shape"])
    return arrs, sdc, hidden, c_dim, nb, bool(leaky), nx, "nearest" if nearest else "bilinear"


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_local_decoder_beyond_the_shipped_shape(tag):
    """The reference's LocalDecoder at 64/32/5 with leaky heads, at 256/128/3 and with sample_mode='nearest' (decoder.py:24-68): the
    oracle's restatement is shape-generic and takes ``leaky`` / ``sample_mode`` -- pinned here against the reference's own outputs."""
    a, sd, hidden, c_dim, nb, leaky, nx, mode = _wide_case(tag)
    kw = dict(leaky=leaky, sample_mode=mode)
    grid, p, c_img = T(a["grid"].astype(np.float32)), T(a["prand"]), T(a["c_img"].astype(np.float32))
    assert sd["fc_p.weight"].shape == (hidden, 3) and sd["fc_c.0.weight"].shape == (hidden, c_dim)
    assert maxdiff(orc.local_decoder_forward(sd, p, grid, **kw), a["logits"]) <= 2e-5
    assert maxdiff(orc.local_decoder_forward_img(sd, p, grid, c_img, **kw), a["logits_img"]) <= 2e-5
    o, oc = orc.local_decoder_forward_contact(sd, p, grid, **kw)
    assert maxdiff(o, a["logits_contact"]) <= 2e-5 and maxdiff(oc, a["logits_contact2"]) <= 2e-5
    pts = (1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).unsqueeze(0)
    assert maxdiff(orc.local_decoder_forward(sd, pts, grid[:1], **kw), a["logits_lattice"]) <= 2e-5
    if leaky:                                                    # the flag matters on this fixture
        assert maxdiff(orc.local_decoder_forward(sd, p, grid, leaky=False), a["logits"]) > 1e-3
    if mode == "nearest":
        assert maxdiff(orc.local_decoder_forward(sd, p, grid), a["logits"]) > 1e-3


def test_trilinear_with_clamped_points():
    a, sd = load_golden("g1_decode.npz")
    feat = orc.trilinear_sample(T(a["grid2"]), T(a["prand"]))
    assert maxdiff(feat.transpose(1, 2), a["feat_rand"]) <= 2e-6
    assert maxdiff(orc.local_decoder_forward(sd, T(a["prand"]), T(a["grid2"])), a["logits_rand"]) <= 2e-5


def test_pointnet_stages_and_grid():
    a, sd = load_golden("g3_pointnet.npz")
    p = T(a["p"])
    c, idx, stages = orc.pointnet_point_features(sd, p, 16, return_stages=True)
    assert torch.equal(idx, T(a["idx"]))                      # voxel ids bit-exact
    for i, s in enumerate(stages):
        assert maxdiff(s, a[f"stage{i}"]) <= 1e-5
    assert maxdiff(c, a["fc_c"]) <= 1e-5
    grid = orc.scatter_mean_grid(c, idx, 16)
    assert maxdiff(grid, a["grid"]) <= 1e-5
    for b in range(2):                                        # empty voxels exactly zero
        occ = torch.nonzero(grid[b].abs().sum(0).reshape(-1)).squeeze(1)
        assert torch.equal(occ, T(a[f"occ{b}"]))


def test_pointnet_with_mean_pooling():
    """pool_local with scatter_type='mean' (pointnet.py:64-69, 116-132) on the object grid and on the three planes, against the
    reference's own outputs (g18, make_pointnet_mean_goldens.py): a cloud with outliers and 300 points in one cell."""
    a, sd = load_golden("g18_pointnet_mean.npz")
    p = T(a["p"])
    c, idx, stages = orc.pointnet_point_features(sub_sd(sd, "grid."), p, 16, return_stages=True, scatter_type="mean")
    assert maxdiff(stages[0], a["grid.stage0"]) <= 1e-5 and maxdiff(stages[4], a["grid.stage4"]) <= 1e-5
    assert maxdiff(c, a["grid.fc_c"]) <= 1e-5
    assert maxdiff(orc.scatter_mean_grid(c, idx, 16), a["grid.fea.grid"]) <= 1e-5
    fea, _, stages, c = orc.plane_pointnet_forward(sub_sd(sd, "planes."), p, 16, return_stages=True, scatter_type="mean")
    assert maxdiff(stages[4], a["planes.stage4"]) <= 1e-5 and maxdiff(c, a["planes.fc_c"]) <= 1e-5
    for k in ("xz", "xy", "yz"):
        assert maxdiff(fea[k], a[f"planes.fea.{k}"]) <= 1e-5
    # and it is not the max pool
    cm, _ = orc.pointnet_point_features(sub_sd(sd, "grid."), p, 16)
    assert maxdiff(cm, a["grid.fc_c"]) > 1e-3


def test_unet3d_and_full_encoder():
    a, sd = load_golden("g4_unet3d.npz")
    y = orc.unet3d_forward(sub_sd(sd, "unet3d."), T(a["x"]))
    assert maxdiff(y, a["y"]) <= 1e-4
    grid = orc.pointnet_encoder_forward(sd, T(a["p"]), 16)
    assert maxdiff(grid, a["grid"]) <= 1e-4


def test_transformer_fusion_and_attention_decoder():
    a, sd = load_golden("g5_fusion.npz")
    fsd = sub_sd(sd, "fuser.")
    for n in (256, 2048):
        out = orc.transformer_fusion(fsd, T(a[f"c_img{n}"]), T(a[f"c{n}"]))
        assert maxdiff(out, a[f"fused{n}"]) <= 2e-5
    lo = orc.attention_decoder_forward_img(sd, T(a["p"]), T(a["grid"]), T(a["c_img256"]))
    assert maxdiff(lo, a["logits"]) <= 5e-5


def test_tactile_unet_eval_and_train_bn():
    a, sd = load_golden("g6_tactile.npz")
    x = T(a["x"])
    assert maxdiff(orc.tactile_unet_forward(sd, x, training=False), a["y_eval"]) <= 1e-5
    assert maxdiff(orc.tactile_unet_forward(sd, x, training=True), a["y_train"]) <= 1e-5


def test_train_step_loss_and_grads_via_autograd_of_oracle():
    """The oracle is differentiable torch code: its autograd must reproduce the
    reference's gradients (A14) so it can check the HIP backward kernels."""
    a, sd = load_golden("g8_trainstep.npz")
    dsd = {k: v.clone().requires_grad_(True) for k, v in sub_sd(sd, "dec.").items()}
    esd = {k: v.clone().requires_grad_(True) for k, v in sub_sd(sd, "enc.").items()}
    c, idx = orc.pointnet_point_features(esd, T(a["p_in"]), 16)
    grid = orc.scatter_mean_grid(c, idx, 16)
    grid.retain_grad()
    c_img = T(a["c_img"]).clone().requires_grad_(True)
    logits = orc.local_decoder_forward_img(dsd, T(a["pq"]), grid, c_img)
    loss = torch.nn.functional.l1_loss(logits, T(a["occ"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(a["loss"])) <= 1e-6
    assert maxdiff(logits.detach(), a["logits"]) <= 2e-5
    for k, v in dsd.items():
        ref = a["g.dec." + k]
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert maxdiff(got, ref) <= 1e-6 + 1e-4 * float(np.abs(ref).max()), k
    for k, v in esd.items():
        ref = a["g.enc." + k]
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert maxdiff(got, ref) <= 1e-6 + 1e-4 * float(np.abs(ref).max()), k
    assert maxdiff(c_img.grad, a["c_img_grad"]) <= 1e-7
    gi = a["grid_grad_idx"]
    gg = grid.grad.permute(0, 2, 3, 4, 1).reshape(2, -1, 32)
    assert maxdiff(gg[gi[:, 0], gi[:, 1]], a["grid_grad_val"]) <= 1e-7


def test_oracle_at_the_shipped_shape_of_config2():
    """g15_config2.npz (real reference; R = 64, UNet3D 4 levels x f_maps 32, LocalDecoder on the 128^3 lattice): the oracle's
    encoder grid and logits, and its UNet3D alone on the two f_maps-32 cases the HIP network is tested with."""
    import config2_case as c2
    z = c2.fixture()
    for tag in ("u16", "u32"):
        net, x = c2.unet_case(z, tag)
        y = orc.unet3d_forward(c2.cpu_sd(net), x)
        scale = float(z[f"{tag}_ystat"][2])
        if tag == "u16":
            assert maxdiff(y, z["u16_y"]) <= 2e-5 * scale
        else:
            assert maxdiff(y[0].reshape(32, -1)[:, T(z["u32_vox"])], z["u32_y_at"]) <= 2e-5 * scale
    enc, dec = c2.models(z)
    grid = orc.pointnet_encoder_forward(c2.cpu_sd(enc), T(z["cloud"]), 64)
    gmax = float(z["grid_stat"][2])
    assert maxdiff(grid[0].reshape(32, -1)[:, T(z["grid_vox"])], z["grid_at"]) <= 2e-5 * gmax
    assert maxdiff(grid[0].double().mean(dim=(1, 2, 3)), z["grid_chan_mean"]) <= 1e-6
    dsd = c2.cpu_sd(dec)
    for name in ("sample", "near"):
        p = c2.lattice_points(z[name]).unsqueeze(0)
        key = "logits" if name == "sample" else "logits_near"
        assert maxdiff(orc.local_decoder_forward(dsd, p, grid)[0], z[key]) <= 1e-5


def test_oracle_plane_unet_matches_the_reference_module_golden():
    """g19 (the real reference src/encoder/unet.py, shipped hand-encoder shape): the oracle's unet2d_forward and torch autograd through
    it reproduce the reference's output, input gradient and parameter-gradient samples -- the oracle the GPU kernels are checked against
    on other shapes (tests/test_plane_unet_gpu.py)."""
    import os
    import numpy as np
    import torch
    from conftest import GOLDEN
    from vtaco_amd.encoder.unet import UNet
    z = np.load(os.path.join(GOLDEN, "g19_plane_unet.npz"))
    torch.manual_seed(int(z["seeds"][0]))
    net = UNet(32, in_channels=32, depth=4, start_filts=32, merge_mode="concat")
    g = torch.Generator().manual_seed(int(z["seeds"][1]))
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    for name in sd:
        assert float(sd[name].detach().double().sum()) == z[f"psum.{name}"][0], name
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    out = orc.unet2d_forward(sd, x)
    (out * torch.from_numpy(z["w"])).sum().backward()
    assert float((out.detach() - torch.from_numpy(z["out"])).abs().max()) <= 1e-5 * float(np.abs(z["out"]).max())
    assert float((x.grad - torch.from_numpy(z["dx"])).abs().max()) <= 1e-5 * float(np.abs(z["dx"]).max())
    for name in sd:
        gr = sd[name].grad.double().reshape(-1)
        idx = torch.randint(0, gr.numel(), (64,), generator=torch.Generator().manual_seed(int(z["seeds"][3]) + sum(map(ord, name))))
        ref = torch.from_numpy(z[f"gsample.{name}"]).double()
        assert float((gr[idx] - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1e-6) + 1e-9, name


G20_SAMPLE_SEED = 2000


def _g20_check_param_grads(z, tag, grads, tol, sample_seed=G20_SAMPLE_SEED):
    """Per parameter: the 64 sampled entries and the sum of its gradient against g20 (a parameter absent from g20 is one the reference
    left without gradient on this path)."""
    seen = 0
    for name, gr in grads.items():
        key = f"{tag}.gsum.{name}"
        if key not in z:
            assert gr is None or float(gr.abs().max()) == 0.0, name
            continue
        seen += 1
        gr = gr.double().reshape(-1).cpu()
        if name.endswith("norm2.bias"):
            # a per-channel constant in front of InstanceNorm (TransformerFusion.py:259-262, 299-304): its gradient is zero in exact
            # arithmetic and rounding noise in the reference (1e-6 of norm2.weight's) -- noise here as well, nothing to compare digit by digit
            wsum = z[key.replace("norm2.bias", "norm2.weight")][1]
            assert float(gr.abs().sum()) <= 1e-4 * wsum and z[key][1] <= 1e-4 * wsum, (tag, name)
            continue
        idx = torch.randint(0, gr.numel(), (64,), generator=torch.Generator().manual_seed(sample_seed + sum(map(ord, name))))
        ref = T(z[f"{tag}.gsample.{name}"]).double()
        gs = z[key]
        typical = max(1e-12, float(gs[1]) / gr.numel())
        assert float((gr[idx] - ref).abs().max()) <= tol * max(8 * typical, float(ref.abs().max())), (tag, name)
        assert abs(float(gr.sum()) - gs[0]) <= tol * gs[1] + 1e-9, (tag, name)
    assert seen == sum(1 for k in z if k.startswith(f"{tag}.gsum."))


@pytest.mark.parametrize("tag", ["D", "E"])
def test_oracle_attention_decoder_autograd_matches_the_reference_gradients(tag):
    """g20 (the REAL reference's autograd through AttentionDecoder.forward_img at c_dim 128 / hidden 256 / 5 blocks and 64 / 64 / 2,
    make_attn_wide_goldens.py): torch autograd through the oracle's restatement gives the same d grid, d c_img and parameter
    gradients -- what makes it the checker of the HIP backward at the widths in between (tests/test_fusion_gpu.py)."""
    import os
    from conftest import GOLDEN
    a, sd = load_golden("g17_attention_wide.npz")
    z = dict(np.load(os.path.join(GOLDEN, "g20_attention_wide_grads.npz")))
    arrs = {k[2:]: v for k, v in a.items() if k.startswith(tag + ".")}
    sdc = {k[2:]: v.float().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items() if k.startswith(tag + ".")}
    grid = T(arrs["grid"].astype("float32")).requires_grad_(True)
    c_img = T(arrs["c_img"].astype("float32")).requires_grad_(True)
    out = orc.attention_decoder_forward_img(sdc, T(arrs["p"]), grid, c_img)
    (out * T(z[f"{tag}.w"])).sum().backward()
    for got, key in ((grid.grad, "d_grid"), (c_img.grad, "d_c_img")):
        ref = T(z[f"{tag}.{key}"])
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (tag, key)
    # the fuser's self-attention unit is ONE module used by its encoder and its decoder layer (TransformerFusion.py:282-296): the
    # reference's named_parameters lists it once, under the encoder's name, with the sum of both uses' gradients
    grads = {k: v.grad for k, v in sdc.items() if v.dtype.is_floating_point and "running" not in k}
    for k in [k for k in grads if k.startswith("fuser.decoder.layers.0.self_attn.")]:
        g = grads.pop(k)
        enc = k.replace("fuser.decoder.", "fuser.encoder.")
        if g is not None:
            grads[enc] = g if grads[enc] is None else grads[enc] + g
    _g20_check_param_grads(z, tag, grads, 2e-5)
