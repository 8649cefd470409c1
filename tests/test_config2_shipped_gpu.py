"""BASELINE config 2 END TO END at the SHIPPED shape against the reference: cloud [1,3000,3] -> HIP encoder (PointNet, voxeliser,
UNet3D with num_levels 4 / f_maps 32 at R = 64: reference configs/VTacO/VTacO_YCB.yaml:22-31, src/encoder/pointnet.py:135-200,
unet3d.py:449-474) -> feature grid -> HIP decode of the 128^3 lattice (decoder.py:135-161) -> logits.

Checked against (1) g15_config2.npz, made by the REAL reference in the build container (tests/golden/make_config2_goldens.py:
grid samples, logits on two lattice samples) and (2) the oracle run here on the host cores on the same inputs (whole grid).
Encoder drift and logit error are reported SEPARATELY, for the exact-f32 and the split-bf16 arithmetic; north_star's bar is
1e-4 absolute on the logits.  The two stand-alone UNet3D cases of the fixture (f_maps 32) go through unet3d.hip, not MIOpen."""
import json
import os

import pytest
import torch

import config2_case as c2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy
REPORT = {}


def _flush_report():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "config2_shipped_parity.json"), "w") as fh:
            json.dump(REPORT, fh, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.mark.parametrize("tag", ["u16", "u32"])
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_hip_unet3d_against_reference_golden(tag, precision):
    """A reference-made UNet3D golden that the HIP network (not the MIOpen host path) answers."""
    z = c2.fixture()
    net, x = c2.unet_case(z, tag)
    assert net.hip_supported()
    net = net.to(DEV)
    net.precision = precision
    with torch.no_grad():
        y = net.forward_channels_last(x.to(DEV).permute(0, 2, 3, 4, 1).contiguous()).permute(0, 4, 1, 2, 3).cpu()
    scale = float(z[f"{tag}_ystat"][2])
    if tag == "u16":
        err = float((y - T(z["u16_y"])).abs().max())
    else:
        err = float((y[0].reshape(32, -1)[:, T(z["u32_vox"])] - T(z["u32_y_at"])).abs().max())
    REPORT[f"unet3d_{tag}_{precision}"] = {"max_abs_err": err, "output_max": scale}
    _flush_report()
    assert err <= (2e-5 if precision == "f32" else 1e-4) * scale, (tag, precision, err, scale)


@pytest.fixture(scope="module")
def shipped():
    from oracle import vtaco_oracle as orc
    z = c2.fixture()
    enc, dec = c2.models(z)
    cloud = T(z["cloud"])
    esd, dsd = c2.cpu_sd(enc), c2.cpu_sd(dec)
    grid_orc = orc.pointnet_encoder_forward(esd, cloud, 64)          # the oracle on this box's host cores (~2-5 s)
    # the oracle itself against the reference-made fixture, on THIS box (thread count / CPU differ from the build container)
    o_err = float((grid_orc[0].reshape(32, -1)[:, T(z["grid_vox"])] - T(z["grid_at"])).abs().max())
    assert o_err <= 2e-5 * float(z["grid_stat"][2]), o_err
    return {"z": z, "enc": enc.to(DEV), "dec": dec.to(DEV), "cloud": cloud, "dsd": dsd, "grid_orc": grid_orc, "orc": orc, "oracle_vs_fixture": o_err}


@pytest.mark.parametrize("enc_precision,dec_precision", [("f32", "f32"), ("bf16x3", "f32"), ("bf16x3", "bf16x3"), ("f16x3", "f32"), ("f16x3", "f16x3"),
                                                         ("f16x3", "f16f8")])
def test_config2_end_to_end_at_the_shipped_shape(shipped, enc_precision, dec_precision):
    s, z, orc = shipped, shipped["z"], shipped["orc"]
    enc, dec = s["enc"], s["dec"]
    enc.unet3d.precision = enc_precision
    nx = 128
    with torch.no_grad():
        # this test is what pins the encoder's work-skipping paths against the reference fixture, so make sure they are the ones
        # that run: the first layer's empty blocks (flags non-zero on this scene) and, with half pairs, the per-parity decoder entries
        from vtaco_amd import _lib, ops as _ops
        assert enc.skip_empty and enc.unet3d.hip_supported()
        flags = _ops.VoxelIndex(s["cloud"].to(DEV), enc.reso_grid, enc.padding, want_tile_flags=True).tile_flags
        assert flags is not None and 0 < int((flags != 0).sum()) < flags.numel(), "the scene must have empty AND occupied 8^3 blocks"
        if enc_precision == "f16x3":
            up = enc.unet3d.decoders[-1].basic_module.SingleConv1.conv
            assert enc.unet3d._packed_up(up, up.out_channels) is not None
            assert _lib.load().vt_conv3d_up_covers(up.out_channels, up.in_channels - up.out_channels, 1, 64, 64, 64, up.out_channels) == 1
        grid = enc(s["cloud"].to(DEV))["grid"]                                         # HIP: pointnet.hip, voxel.hip, unet3d.hip
        logits = dec.decode_lattice(grid, nx, box=1.1, precision=dec_precision).reshape(-1)   # HIP: decode.hip, whole 128^3 lattice
        # the same decode kernel on the ORACLE's grid: the decoder's own error, free of encoder drift
        from vtaco_amd import ops
        grid_orc_dev = ops.grid_to_channels_last(s["grid_orc"].to(DEV))
        logits_on_orc = dec.decode_lattice(grid_orc_dev, nx, box=1.1, precision=dec_precision).reshape(-1)
    gmax = float(z["grid_stat"][2])
    g_cpu = grid.cpu()
    rep = {"grid_max": gmax,
           "encoder_drift_vs_oracle_whole_grid": float((g_cpu - s["grid_orc"]).abs().max()),
           "encoder_drift_vs_reference_sample": float((g_cpu[0].reshape(32, -1)[:, T(z["grid_vox"])] - T(z["grid_at"])).abs().max()),
           "oracle_vs_reference_sample": s["oracle_vs_fixture"]}
    lo_cpu, lo_orc_cpu = logits.cpu(), logits_on_orc.cpu()
    for name, key in (("sample", "logits"), ("near", "logits_near")):
        idx, ref = T(z[name]), T(z[key])
        rep[f"logit_err_end_to_end_{name}"] = float((lo_cpu[idx] - ref).abs().max())
        rep[f"logit_err_decoder_only_{name}"] = float((lo_orc_cpu[idx] - ref).abs().max())
        rep[f"logit_absmax_{name}"] = float(ref.abs().max())
    # the oracle's logits on its own grid for a third, independent sample of this box's choosing (whole rows of the lattice)
    g = torch.Generator().manual_seed(7)
    idx = torch.randperm(nx ** 3, generator=g)[:65536].sort().values
    ref = orc.local_decoder_forward(s["dsd"], c2.lattice_points(idx).unsqueeze(0), s["grid_orc"])[0]
    rep["logit_err_end_to_end_vs_oracle_65536"] = float((lo_cpu[idx] - ref).abs().max())
    REPORT[f"config2_enc_{enc_precision}_dec_{dec_precision}"] = rep
    _flush_report()
    print(json.dumps({f"{enc_precision}/{dec_precision}": rep}))
    # north_star's bar: 1e-4 absolute on the logits, end to end, against the reference
    bar = 1e-4
    assert rep["logit_err_end_to_end_sample"] <= bar and rep["logit_err_end_to_end_near"] <= bar, rep
    assert rep["logit_err_end_to_end_vs_oracle_65536"] <= bar, rep
    assert rep["logit_err_decoder_only_sample"] <= {"f32": 5e-6, "f16f8": 9e-5}.get(dec_precision, 5e-5), rep      # measured 2.2e-6 / 5.5e-6
    # encoder drift, reported separately: f32 convs stay at f32 rounding noise, the split-bf16 convs within 1e-4 of the grid's scale
    assert rep["encoder_drift_vs_oracle_whole_grid"] <= (2e-5 if enc_precision == "f32" else 1e-4) * gmax, rep
    if enc_precision == "f16x3":
        # half pairs on EVERY level since round 4 (the thin levels too): measured 2.5e-5 drift (6.1e-5 with bf16 pairs there) and
        # 5.2e-6 end to end with the default decode -- a regression to bf16 pairs on the thin levels would show here
        assert rep["encoder_drift_vs_oracle_whole_grid"] <= 4e-5 * gmax, rep
        if dec_precision in ("f32", "f16x3"):
            assert rep["logit_err_end_to_end_vs_oracle_65536"] <= 1e-5, rep
