"""GPU parity of the fp8-corrected lattice decode (vt_decode_fwd_f16f8, through the C ABI): f16 hi products plus ONE fp8 (e4m3)
MFMA for both correction products of every dense layer.  Same golden vectors and the same 1e-4 bar as the other decode kernels
(BASELINE.json north_star; reference decoder.py:135-161, 71-103); lattice slabs only -- everything else stays on the split-f16
kernel, which these tests also check the fall-back to."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _blob(sd, dev, img=False, precision="f16f8"):
    from vtaco_amd import ops
    g = lambda k: sd[k].to(dev)
    pw, pb = (g("fc_p_img.weight"), g("fc_p_img.bias")) if img else (g("fc_p.weight"), g("fc_p.bias"))
    fc_c = [(g(f"fc_c.{i}.weight"), g(f"fc_c.{i}.bias")) for i in range(5)]
    blocks = [(g(f"blocks.{i}.fc_0.weight"), g(f"blocks.{i}.fc_0.bias"),
               g(f"blocks.{i}.fc_1.weight"), g(f"blocks.{i}.fc_1.bias")) for i in range(5)]
    return ops.pack_decoder(pw, pb, fc_c, blocks, (g("fc_out.weight"), g("fc_out.bias")), None, precision=precision)


def test_golden_lattice_and_slabs():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device(DEV)
    grid = torch.from_numpy(a["grid"]).to(dev)
    ref = torch.from_numpy(a["logits"])
    lat = (32, 1.1, 0, 32 ** 3)
    assert ops.f16f8_covers(grid, lat)
    blob = _blob(sd, dev)
    got = ops.decode_fwd(grid, blob, lattice=lat, precision="f16f8").cpu()
    err = float((got - ref).abs().max())
    print("f16f8 golden lattice: max abs logit error", err)
    assert err <= TOL
    # it IS another arithmetic than the split-f16 kernel's (whose error is ~1e-6): the fp8 corrections leave 1e-6 .. 1e-4
    f16 = ops.decode_fwd(grid, _blob(sd, dev, precision="f16x3"), lattice=lat, precision="f16x3").cpu()
    assert float((f16 - ref).abs().max()) < 5e-6 < err
    # slabs of whole x-plane pairs are the whole lattice's values bit for bit
    pair = 2 * 32 * 32
    parts = [ops.decode_fwd(grid, blob, lattice=(32, 1.1, first, cnt), precision="f16f8")
             for first, cnt in ((0, 3 * pair), (3 * pair, 5 * pair), (8 * pair, 8 * pair))]
    assert torch.equal(torch.cat(parts, dim=1).cpu(), got)


def test_what_it_does_not_cover_is_refused_and_the_decoder_falls_back():
    from vtaco_amd import ops
    from vtaco_amd._lib import VtError
    from vtaco_amd.conv_onet.models import decoder_dict
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device(DEV)
    grid = torch.from_numpy(a["grid"]).to(dev)
    blob = _blob(sd, dev)
    assert not ops.f16f8_covers(grid, (32, 1.1, 7, 32 ** 3 - 7))            # not whole plane pairs
    assert not ops.f16f8_covers(grid, (36, 1.1, 0, 36 ** 3))                # nx % 8
    assert not ops.f16f8_covers(grid, (16, 1.1, 0, 16 ** 3))                # a lattice step of a whole voxel
    with pytest.raises(VtError):
        ops.decode_fwd(grid, blob, lattice=(32, 1.1, 7, 32 ** 3 - 7), precision="f16f8")
    with pytest.raises(VtError):
        ops.decode_fwd(grid, blob, pts=torch.from_numpy(a["pts"]).to(dev), precision="f16f8")
    dec = decoder_dict["simple_local"](dim=3, c_dim=32, hidden_size=32, n_blocks=5, padding=0.1)
    dec.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc_out_contact")}, strict=False)
    dec = dec.to(dev)
    dec.precision = "f16f8"
    ref = torch.from_numpy(a["logits"])
    with torch.no_grad():
        pts_logits = dec(torch.from_numpy(a["pts"]).to(dev), {"grid": grid}).cpu()           # point queries: split-f16
        odd = dec.decode_lattice(grid, 32, first=7, count=32 ** 3 - 7).cpu()                  # unaligned slab: split-f16
    assert float((pts_logits - ref).abs().max()) <= 5e-6
    assert float((odd - ref[:, 7:]).abs().max()) <= 5e-6


def test_tactile_concat_dense_and_by_finger_id():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device(DEV)
    grid = torch.from_numpy(a["grid"]).to(dev)
    nx = 32
    lat = (nx, 1.1, 0, nx ** 3)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(5, 32, generator=g).to(dev)
    ids = torch.full((1, nx ** 3), 255, dtype=torch.uint8)
    hit = torch.rand(nx ** 3, generator=g) < 0.03
    ids[0, hit] = torch.randint(0, 5, (int(hit.sum()),), generator=g).to(torch.uint8)
    ids = ids.to(dev)
    dense = torch.zeros(1, nx ** 3, 32, device=dev)
    dense[0, hit.to(dev)] = feats[ids[0, hit.to(dev)].long()]
    blob = _blob(sd, dev, img=True)
    by_id = ops.decode_fwd_ids(grid, blob, ids, feats, lattice=lat, precision="f16f8")
    by_dense = ops.decode_fwd(grid, blob, c_img=dense, lattice=lat, precision="f16f8")
    assert torch.equal(by_id, by_dense)
    exact = ops.decode_fwd(grid, _blob(sd, dev, img=True, precision="f32"), c_img=dense, lattice=lat)
    err = float((by_id - exact).abs().max())
    print("f16f8 tactile concat: max abs error against the exact-f32 kernel", err)
    assert err <= TOL


def test_full_size_lattices_against_the_exact_f32_kernel():
    """128^3 and 256^3 on the bench scene's encoder grid (R = 64): every logit within the bar of the exact-f32 kernel."""
    from vtaco_amd.bench_util import build_scene
    dev = torch.device(DEV)
    sc = build_scene(0, dev)
    dec, grid = sc["model"].decoder, sc["grid"]
    for nx in (128, 256):
        with torch.no_grad():
            exact = dec.decode_lattice(grid, nx, precision="f32")
            fast = dec.decode_lattice(grid, nx, precision="f16f8")
        err = float((fast - exact).abs().max())
        print(f"f16f8 {nx}^3: max abs error against the exact-f32 kernel {err:.3e} (logits up to {float(exact.abs().max()):.2f})")
        assert err <= TOL
        assert torch.isfinite(fast).all()


def test_huge_activations_saturate_the_correction_instead_of_poisoning_it():
    """Activations beyond the fp8 copies' range (448 * 4): the conversions saturate (MODE.FP16_OVFL), nothing turns NaN, and the
    result is what two-product f16 arithmetic gives there (relative error ~2^-11), not garbage."""
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device(DEV)
    grid = torch.from_numpy(a["grid"]).to(dev)
    big = dict(sd)
    big["fc_p.bias"] = sd["fc_p.bias"] + 3.0e4
    lat = (32, 1.1, 0, 32 ** 3)
    ops.decode_range_status(reset=True)
    got = ops.decode_fwd(grid, _blob(big, dev), lattice=lat, precision="f16f8")
    assert ops.decode_range_status(reset=True) == 0            # 3e4 is beyond the fp8 copies' range, not beyond the half range
    ref = ops.decode_fwd(grid, _blob(big, dev, precision="f32"), lattice=lat, precision="f32")
    assert torch.isfinite(got).all()
    rel = float(((got - ref).abs() / ref.abs().clamp_min(1.0)).max())
    print("f16f8 with activations ~3e4: max relative error", rel)
    assert rel <= 2e-3


def test_the_generator_falls_back_when_the_range_guard_trips():
    """A decoder whose hidden activations leave the half range: the launch reports it (vt_decode_range_status), the generator
    warns, switches its lattice decode to 'bf16x3' and generates the scene again."""
    import warnings
    from vtaco_amd import ops
    from vtaco_amd.bench_util import build_scene
    from vtaco_amd.conv_onet.generation import Generator3D
    dev = torch.device(DEV)
    sc = build_scene(0, dev)
    model = sc["model"]
    with torch.no_grad():
        model.decoder.fc_p.bias += 1.0e5
        model.decoder.fc_out.weight *= 1e-5                       # keeps the logits (and the surface) at a sane scale
    pc = sc["cloud"].to(dev)
    ref = Generator3D(model, device=dev, resolution0=16, padding=0.1, decode_precision="bf16x3").generate_obj_mesh_wnf({"inputs": pc})
    gen = Generator3D(model, device=dev, resolution0=16, padding=0.1)
    assert gen.decode_precision == "f16f8"
    ops.decode_range_status(reset=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        mesh = gen.generate_obj_mesh_wnf({"inputs": pc})
    assert gen.decode_precision == "bf16x3" and any("half-precision range" in str(x.message) for x in w)
    # the scene it hands back is the split-bf16 generator's, vertex for vertex
    assert torch.equal(mesh.faces, ref.faces) and torch.equal(mesh.vertices, ref.vertices)
