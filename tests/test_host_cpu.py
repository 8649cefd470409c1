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
    assert decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32, leaky=True)._wide       # leaky heads: the wide kernel (round 3)
    assert decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32, sample_mode="nearest")._wide   # F.grid_sample's other mode: wide kernel
    with pytest.raises(VtError):
        decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32, sample_mode="bicubic")         # not a 5-D grid_sample mode
    assert decoder_dict["attention_local"](dim=3, c_dim=128, hidden_size=256)._wide               # the reference's default widths: built (round 5)
    with pytest.raises(VtError):
        decoder_dict["attention_local"](dim=3, c_dim=160, hidden_size=64)                         # fusion kernels: d_model 32 / 64 / 96 / 128
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


VTACO_YCB_SHAPED_YAML = """
# a config with the KEYS of the reference's shipped configs/VTacO/VTacO_YCB.yaml (method, model.*, encoder_t2d_kwargs ...),
# small sizes where a size does not matter to the factory; written by this test, not copied
method: vtaco
data:
  dataset: Shapes3D
  input_type: pointcloud
  classes: null
  path: {root}
  dim: 3
  padding: 0.1
  train_split: train
  val_split: val
  test_split: val
  multi_files: null
  pointcloud_n: 150
  pointcloud_noise: 0.005
  points_subsample: 128
  num_sample: 64
  points_file: points.npz
  points_iou_file: points.npz
  voxels_file: null
  pointcloud_file: pointcloud.npz
  points_unpackbits: False
model:
  train_tactile: False
  with_img: True
  with_contact: False
  encoder: pointnet_local_pool
  encoder_kwargs:
    hidden_dim: 32
    plane_type: 'grid'
    grid_resolution: 64
    unet3d: True
    unet3d_kwargs: {{num_levels: 4, f_maps: 32, in_channels: 32, out_channels: 32}}
  encoder_hand: pointnet_local_pool
  encoder_hand_kwargs:
    hidden_dim: 32
    plane_type: ['xz', 'xy', 'yz']
    plane_resolution: 32
    unet: True
    unet_kwargs: {{depth: 4, merge_mode: concat, start_filts: 32}}
    out_mano: True
    out_dim: 51
    manolayer_kwargs: &manolayer_k
      center_idx: 9
      flat_hand_mean: False
      ncomps: 45
      side: right
      mano_root: {mano}
      use_pca: False
      root_rot_mode: axisang
      joint_rot_mode: axisang
      robust_rot: False
      return_transf: False
      return_full_pose: True
  encoder_img: Resnet18
  encoder_img_kwargs: {{num_classes: 32}}
  encoder_t2d: True
  encoder_t2d_kwargs:
    pretrained: True
    model_file: {ckpt}
    encoder_img: UNet
    encoder_img_kwargs: {{num_classes: 1, in_channel: 3, start_filts: 32, depth: 3}}
    encoder_hand: pointnet_local_pool
    encoder_hand_kwargs:
      c_dim: 512
      hidden_dim: 32
      plane_type: ['xz', 'xy', 'yz']
      plane_resolution: 64
      unet: True
      unet_kwargs: {{depth: 4, merge_mode: concat, start_flits: 32}}
      out_mano: True
      out_dim: 30
      manolayer_kwargs: *manolayer_k
  decoder: simple_local
  decoder_kwargs: {{sample_mode: bilinear, hidden_size: 32}}
  c_dim: 32
training: {{out_dir: {out}, batch_size: 3}}
test: {{threshold: 0.5}}
generation: {{resolution_0: 32, upsampling_steps: 0}}
"""


def test_shipped_shape_yaml_builds_dataset_model_trainer_and_loads_the_t2d_checkpoint(tmp_path):
    """``method: vtaco`` (reference src/config.py:7-9) resolves; ``get_model`` builds every module of the shipped VTacO config
    (20 011 415 parameters) and LOADS ``encoder_t2d_kwargs.model_file`` into the t2d net
    (reference conv_onet/config.py:131-133: CheckpointIO -> state['model']); a missing file raises as the reference does."""
    import os
    import sys
    import yaml
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import synth_dataset
    import synth_mano
    from vtaco_amd import config as topcfg
    from vtaco_amd._lib import VtError
    root = str(tmp_path / "data")
    os.makedirs(root)
    synth_dataset.make_synthetic_dataset(root)
    synth_mano.write_pkl(synth_mano.make_asset(0), str(tmp_path / "mano"))
    out = tmp_path / "out"
    out.mkdir()
    text = VTACO_YCB_SHAPED_YAML.format(root=root, mano=str(tmp_path / "mano"), ckpt="t2d/model.pt", out=str(out))
    cfg = yaml.safe_load(text)
    assert cfg["method"] == "vtaco" and topcfg.method_dict["vtaco"] is topcfg.method_dict["conv_onet"]
    ds = topcfg.get_dataset("train", cfg)
    assert len(ds) > 0 and "points" in ds[0] and "inputs" in ds[0]
    method = topcfg.method_dict[cfg["method"]]
    with pytest.raises(FileNotFoundError):                 # pretrained: True and nothing to load
        method.get_model(cfg, device=None)
    # a checkpoint as the reference's CheckpointIO.save writes it: {'model': state_dict, scalars...} relative to out_dir
    cfg_np = yaml.safe_load(text)
    cfg_np["model"]["encoder_t2d_kwargs"]["pretrained"] = False
    torch.manual_seed(3)
    donor = method.get_model(cfg_np, device=None)
    assert sum(p.numel() for p in donor.parameters()) == 20011415
    (out / "t2d").mkdir()
    torch.save({"model": donor.encoder_t2d.state_dict(), "epoch_it": 7, "loss_val_best": 0.5}, str(out / "t2d" / "model.pt"))
    torch.manual_seed(4)
    model = method.get_model(cfg, device=None)
    for (k, a), (_, b) in zip(model.encoder_t2d.state_dict().items(), donor.encoder_t2d.state_dict().items()):
        assert torch.equal(a, b), k
    assert not torch.equal(model.decoder.fc_p.weight, donor.decoder.fc_p.weight)      # everything else is a fresh init
    torch.save({"epoch_it": 7}, str(out / "t2d" / "model.pt"))
    with pytest.raises(VtError):                              # a checkpoint without a 'model' entry
        method.get_model(cfg, device=None)
    trainer = method.get_trainer(donor, torch.optim.Adam(donor.parameters(), lr=1e-4), cfg, None)
    assert trainer.encode_t2d and trainer.with_img and trainer.pretrained_t2d and trainer.num_sample == 64
    gen = method.get_generator(donor, cfg, None)
    assert gen.with_img and gen.encode_t2d and gen.resolution0 == 32


def test_generator_weight_stamps_and_eval_mode_follow_the_model():
    """Generator3D's per-scene checks (the captured scene graph is replayed only while they are unchanged) walk cached module
    tables instead of nn.Module's recursive generators: a weight updated in place, moved / cast (``.to()`` swaps the storage under the
    same Parameter), replaced by a new Parameter, or a buffer written in place must all change the stamps; ``_eval_mode`` must
    put every submodule in eval mode whichever one was switched to train."""
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    from vtaco_amd.encoder import encoder_dict
    dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1)
    model = ConvolutionalOccupancyNetwork(dec, None, device='cpu')
    model.encoder = encoder_dict['pointnet_local_pool'](dim=3, c_dim=32, padding=0.1, hidden_dim=32, plane_type='grid', grid_resolution=16,
                                                        unet3d=True, unet3d_kwargs=dict(num_levels=2, f_maps=32, in_channels=32, out_channels=32))
    model.register_buffer("some_buffer", torch.zeros(3))
    gen = Generator3D(model, device='cpu', resolution0=8, padding=0.1)
    want = tuple((id(t), t.data_ptr(), t._version) for t in list(model.parameters()) + list(model.buffers()))
    got = gen._weight_stamps()
    assert sorted(zip(*got)) == sorted(want)                          # the same tensors as the recursive walk finds
    s0 = gen._weight_stamps()
    assert gen._weight_stamps() == s0
    with torch.no_grad():
        model.decoder.fc_out.weight.add_(1.0)                         # optimizer.step / load_state_dict
    s1 = gen._weight_stamps()
    assert s1 != s0
    model.decoder.fc_out.weight = torch.nn.Parameter(torch.zeros_like(model.decoder.fc_out.weight))
    s2 = gen._weight_stamps()
    assert s2 != s1
    model.double()                                                    # .to(): new storage under the same Parameter objects
    s3 = gen._weight_stamps()
    assert s3 != s2
    model.some_buffer.add_(1.0)
    assert gen._weight_stamps() != s3
    model.train()
    gen._eval_mode()
    assert not any(m.training for m in model.modules())
    model.encoder.unet3d.train()                                      # one branch only
    gen._eval_mode()
    assert not any(m.training for m in model.modules())


def test_range_guard_walks_f16f8_to_f16x3_to_bf16x3_and_stays_rank_local(monkeypatch):
    """Generator3D's range guard (host logic; the device word is played by a stub): RANGE_FP8 / RANGE_LOGIT move an 'f16f8'
    generator to 'f16x3', RANGE_HALF moves either half-precision form to 'bf16x3', each with a warning and ONE regeneration per
    move; a clean word changes nothing; the non-sharded entry points never touch torch.distributed (a collective hidden there
    would hang a rank-0-only export)."""
    import warnings
    import torch.distributed as dist
    from vtaco_amd import ops
    from vtaco_amd.conv_onet import generation as gen_mod

    class Dummy:
        def __init__(self, precision):
            self.decode_precision, self.device, self.calls = precision, None, []
            self.model = type("M", (), {"decoder": type("D", (), {})()})()

        _set_decode_precision = gen_mod.Generator3D._set_decode_precision

        @gen_mod._range_guarded
        def make(self, x):
            self.calls.append(self.decode_precision)
            return (x, self.decode_precision)

    def forbid(*a, **k):
        raise AssertionError("a collective was reached from a non-sharded entry point")
    for name in ("all_reduce", "broadcast", "all_gather", "barrier"):
        monkeypatch.setattr(dist, name, forbid)
    words, clears = [], []
    monkeypatch.setattr(ops, "decode_range_status", lambda reset=True: words.pop(0) if words else 0)
    monkeypatch.setattr(ops, "decode_range_clear", lambda: clears.append(1))     # the scene-start clear (asynchronous on the device)
    # clean
    g = Dummy("f16f8")
    assert g.make(1) == (1, "f16f8") and g.calls == ["f16f8"] and clears == [1]    # one clear per guarded scene, before its launches
    # the fp8 copies clipped, then the half range too: two moves, three generations
    g, words[:] = Dummy("f16f8"), [ops.RANGE_FP8, ops.RANGE_HALF, 0]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert g.make(2) == (2, "bf16x3")
    assert g.calls == ["f16f8", "f16x3", "bf16x3"] and len(w) == 2 and g.model.decoder.__dict__ == {}
    # a logit beyond 2.5: only 'f16f8' cares
    g, words[:] = Dummy("f16f8"), [ops.RANGE_LOGIT, 0]
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert g.make(3) == (3, "f16x3") and g.calls == ["f16f8", "f16x3"]
    g, words[:] = Dummy("f16x3"), [ops.RANGE_LOGIT | ops.RANGE_FP8]
    assert g.make(4) == (4, "f16x3") and g.calls == ["f16x3"]
    # the half range from the default precision; the attention decoder's MLP goes back to the exact kernel with it
    g, words[:] = Dummy("f16x3"), [ops.RANGE_HALF, 0]
    g.model.decoder.mlp_precision = "f16x3"
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert g.make(5) == (5, "bf16x3") and g.model.decoder.mlp_precision == "f32"
    # other precisions are not guarded at all (no status read)
    g, words[:] = Dummy("f32"), [ops.RANGE_HALF]
    n_clears = len(clears)
    assert g.make(6) == (6, "f32") and words == [ops.RANGE_HALF] and len(clears) == n_clears


def test_wide_decoder_precision_routing_and_encoder_skip_switch(monkeypatch):
    """Host logic without a device: a decoder beyond 32 / 32 maps the half-precision settings to the split-f16 wide kernel and "f32" /
    "bf16x3" (the range guard's way out) to the exact one; the encoder's block skipping is on by default and VTACO_UNET_SKIP=0 turns
    it off; block flags exist only for resolutions the 8^3 blocks tile."""
    from types import SimpleNamespace
    from vtaco_amd import ops
    from vtaco_amd.conv_onet.models.decoder import LocalDecoder
    from vtaco_amd.encoder import encoder_dict
    dec = LocalDecoder(dim=3, c_dim=128, hidden_size=256, n_blocks=2)
    assert dec._wide
    assert [dec._wide_precision(p) for p in ("f16x3", "f16f8", "f32", "bf16x3")] == ["wide_f16x3", "wide_f16x3", "wide", "wide"]
    assert not LocalDecoder(dim=3, c_dim=32, hidden_size=32)._wide
    kw = dict(c_dim=32, dim=3, hidden_dim=32, unet3d=True, grid_resolution=64, plane_type='grid',
              unet3d_kwargs=dict(num_levels=3, f_maps=32, in_channels=32, out_channels=32))
    assert encoder_dict['pointnet_local_pool'](**kw).skip_empty
    monkeypatch.setenv("VTACO_UNET_SKIP", "0")
    assert not encoder_dict['pointnet_local_pool'](**kw).skip_empty
    for R in (12, 4, 136):                                         # not a multiple of 8 / below a block / beyond the flags' LDS table
        assert ops.voxel_tile_flags(SimpleNamespace(R=R, B=1, T=1, idx=None)) is None


def test_tactile_resnet_over_all_scenes_at_once_equals_the_per_scene_loop():
    """TactileResNet.forward_scenes (one pass over S x F images, every BatchNorm with the statistics of each scene's images alone) against
    the reference's loop over the scenes (models/__init__.py:115-136): outputs, parameter gradients and the running statistics after
    the S sequential updates; eval mode too."""
    import copy
    from vtaco_amd.layers import Resnet18
    torch.manual_seed(5)
    net = Resnet18(8).train()
    ref = copy.deepcopy(net)
    imgs = torch.rand(3, 2, 3, 64, 48)
    a = net.forward_scenes(imgs)
    b = torch.cat([ref(imgs[s]).reshape(1, 2, -1) for s in range(3)])
    assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    w = torch.randn(a.shape, generator=torch.Generator().manual_seed(6))
    (a * w).sum().backward()
    (b * w).sum().backward()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 2e-5 * max(1e-6, float(q.grad.abs().max()))
    for (n1, p), (_, q) in zip(net.named_buffers(), ref.named_buffers()):
        assert float((p.float() - q.float()).abs().max()) <= 2e-6 * max(1.0, float(q.float().abs().max())), n1
    net.eval(); ref.eval()
    with torch.no_grad():
        e = net.forward_scenes(imgs)
        f = torch.cat([ref(imgs[s]).reshape(1, 2, -1) for s in range(3)])
    assert float((e - f).abs().max()) <= 2e-6 * max(1.0, float(f.abs().max()))


def test_trainer_train_mode_respects_a_train_override():
    """Trainer.train_step puts the model in train mode through a cached module list (332 modules: 0.5 ms of host time per step the plain
    way) -- unless a module's class overrides train(), e.g. a frozen sub-net that keeps its BatchNorm in eval mode: then model.train() runs."""
    import torch
    from vtaco_amd.conv_onet.training import Trainer

    class Frozen(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = torch.nn.BatchNorm1d(4)

        def train(self, mode=True):
            return super().train(False)

    plain = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4)).eval()
    tr = Trainer(plain, None, device="cpu")
    tr._model_train()
    assert all(m.training for m in plain.modules())
    mixed = torch.nn.Sequential(torch.nn.Linear(4, 4), Frozen()).eval()
    tr = Trainer(mixed, None, device="cpu")
    tr._model_train()
    assert mixed[0].training and not mixed[1].training and not mixed[1].bn.training
