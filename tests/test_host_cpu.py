"""CPU: the host mirror of the reference interface -- registry names, constructor kwargs,
state_dict keys/shapes (against the reference-generated goldens), the host-PyTorch modules'
numerics (UNet3D, tactile UNet), config factories, and the no-CPU-fallback contract."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub_sd

T = torch.from_numpy


def _keys(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items() if "num_batches_tracked" not in k}


def test_registries_have_the_reference_names():
    from vtaco_amd.conv_onet.models import decoder_dict
    from vtaco_amd.encoder import encoder_dict
    assert {"simple_local", "attention_local"} <= set(decoder_dict)
    assert {"pointnet_local_pool", "UNet"} <= set(encoder_dict)


def test_local_decoder_state_dict_matches_reference():
    from vtaco_amd.conv_onet.models import decoder_dict
    _, sd = load_golden("g1_decode.npz")
    dec = decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1, with_contact=True,
                                       sample_mode="bilinear")
    assert _keys(dec) == {k: tuple(v.shape) for k, v in sd.items()}
    dec.load_state_dict(sd, strict=True)


def test_attention_decoder_state_dict_matches_reference():
    from vtaco_amd.conv_onet.models import decoder_dict
    _, sd = load_golden("g5_fusion.npz")
    dec = decoder_dict["attention_local"](dim=3, c_dim=32, hidden_size=32)
    assert _keys(dec) == {k: tuple(v.shape) for k, v in sd.items()}
    # encoder layer and decoder self-attention are ONE module in the reference
    assert dec.fuser.encoder.layers[0].self_attn is dec.fuser.decoder.layers[0].self_attn


def test_pointnet_and_unet3d_state_dict_and_host_numerics():
    from vtaco_amd.encoder import encoder_dict
    a, sd = load_golden("g4_unet3d.npz")
    enc = encoder_dict["pointnet_local_pool"](dim=3, c_dim=32, padding=0.1, hidden_dim=32, plane_type="grid", grid_resolution=16,
                                              unet3d=True, unet3d_kwargs=dict(num_levels=3, f_maps=8, in_channels=32, out_channels=32))
    assert _keys(enc) == {k: tuple(v.shape) for k, v in sd.items()}
    enc.load_state_dict(sd, strict=True)
    with torch.no_grad():
        y = enc.unet3d(T(a["x"]))
    assert float((y - T(a["y"])).abs().max()) <= 1e-4
    _, sd3 = load_golden("g3_pointnet.npz")
    enc3 = encoder_dict["pointnet_local_pool"](dim=3, c_dim=32, hidden_dim=32, plane_type=["grid"], grid_resolution=16)
    assert _keys(enc3) == {k: tuple(v.shape) for k, v in sd3.items()}


def test_tactile_unet_matches_reference_eval_and_train():
    from vtaco_amd.encoder import encoder_dict
    a, sd = load_golden("g6_tactile.npz")
    net = encoder_dict["UNet"](num_classes=1, in_channels=3, depth=3, start_filts=8, in_channel=3, start_flits=8)  # typo kwargs are swallowed, as in the reference
    assert _keys(net) == {k: tuple(v.shape) for k, v in sd.items()}
    net.load_state_dict(sd, strict=False)
    net.eval()
    with torch.no_grad():
        assert float((net(T(a["x"])) - T(a["y_eval"])).abs().max()) <= 1e-5
    net.train()
    with torch.no_grad():
        assert float((net(T(a["x"])) - T(a["y_train"])).abs().max()) <= 1e-5


def test_unsupported_configurations_raise():
    from vtaco_amd._lib import VtError
    from vtaco_amd.conv_onet.models import decoder_dict
    from vtaco_amd.encoder import encoder_dict
    with pytest.raises(ValueError, match="incorrect scatter type"):
        encoder_dict["pointnet_local_pool"](scatter_type="median", plane_type="grid", grid_resolution=8)
    with pytest.raises(VtError):
        encoder_dict["pointnet_local_pool"](plane_type=["xz", "xy", "yz"])                      # planes need plane_resolution
    with pytest.raises(VtError):
        encoder_dict["pointnet_local_pool"](plane_type=["xz", "grid"], plane_resolution=32, grid_resolution=32)
    with pytest.raises(VtError):
        encoder_dict["pointnet_local_pool"](plane_type=["xz", "xy", "yz"], plane_resolution=32, out_mano=True, out_dim=51)  # no MANO asset given
    with pytest.raises(VtError):
        decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32, leaky=True)
    with pytest.raises(KeyError):
        decoder_dict["simple_local_crop"]


def test_get_model_and_generator_from_config():
    from vtaco_amd.conv_onet import config
    cfg = {
        "data": {"dim": 3, "padding": 0.1, "input_type": "pointcloud"},
        "model": {"decoder": "simple_local", "encoder": "pointnet_local_pool", "encoder_hand": False, "c_dim": 32,
                  "decoder_kwargs": {"sample_mode": "bilinear", "hidden_size": 32},
                  "encoder_kwargs": {"hidden_dim": 32, "plane_type": "grid", "grid_resolution": 16, "unet3d": True,
                                     "unet3d_kwargs": {"num_levels": 3, "f_maps": 32, "in_channels": 32, "out_channels": 32}},
                  "with_img": False, "with_contact": False, "encoder_t2d": False},
        "generation": {"resolution_0": 8, "upsampling_steps": 0}, "test": {"threshold": 0.5},
    }
    model = config.get_model(cfg, device="cpu")
    gen = config.get_generator(model, cfg, device="cpu")
    assert gen.resolution0 == 8 and gen.padding == 0.1 and gen.points_batch_size == 100000
    assert hasattr(model, "encode_inputs") and hasattr(model, "decode_img") and hasattr(model, "decode_contact")
    assert model.encoder.unet3d.hip_supported()


def test_common_helpers_match_oracle():
    from oracle import vtaco_oracle as orc
    from vtaco_amd import common
    g = torch.Generator().manual_seed(0)
    p = (torch.rand(2, 500, 3, generator=g) - 0.5) * 1.4
    assert torch.equal(common.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (8,) * 3), orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (8,) * 3))
    q = common.normalize_3d_coordinate(p)
    assert torch.equal(q, orc.normalize_3d_coordinate(p))
    assert torch.equal(common.coordinate2index(q, 16)[:, 0], orc.coordinate2index_3d(q, 16))


def test_tactile_resnet18_against_the_reference_module():
    """encoder_dict['Resnet18'] (the shipped configs' tactile feature encoder) against the reference module's outputs
    (g14_resnet.npz); the 11 M parameters are rebuilt on both sides by the same seeded fill, in state_dict order."""
    import os
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_resnet_goldens import deterministic_fill
    from vtaco_amd.encoder import encoder_dict
    z = np.load(os.path.join(GOLDEN, "g14_resnet.npz"))
    net = encoder_dict["Resnet18"](num_classes=32)
    keys = [f"{k}:{tuple(v.shape)}" for k, v in net.state_dict().items() if "num_batches_tracked" not in k]
    assert keys == list(z["keys"])                                   # same checkpoint names, shapes and ORDER
    deterministic_fill(net, 90)
    x = T(z["x"])
    net.eval()
    with torch.no_grad():
        y = net(x)
    assert y.shape == (2, 32) and float((y - T(z["y_eval"])).abs().max()) <= 1e-4 * float(T(z["y_eval"]).abs().max())
    net.train()
    with torch.no_grad():
        y = net(x)
    assert float((y - T(z["y_train"])).abs().max()) <= 1e-4 * float(T(z["y_train"]).abs().max())
    assert sum(p.numel() for p in encoder_dict["Resnet34"](num_classes=32).parameters()) > sum(p.numel() for p in net.parameters())
