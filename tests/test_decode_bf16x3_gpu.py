"""GPU parity of the split 16-bit decodes (vt_decode_fwd_bf16x3 and vt_decode_fwd_f16x3, through the C ABI):
the same golden vectors and seeded oracle comparisons as the exact-f32 kernel, same 1e-4 bar
(BASELINE.json north_star), plus how far each sits from the exact-f32 kernel.  Every test runs once per
split precision (the autouse fixture sets the module's P)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4
P = "bf16x3"


@pytest.fixture(params=["bf16x3", "f16x3"], autouse=True)
def split_precision(request):
    globals()["P"] = request.param
    return request.param


def _blob(sd, dev, img=False, contact=False, precision=None):
    from vtaco_amd import ops
    precision = precision or P
    g = lambda k: sd[k].to(dev)
    pw, pb = (g("fc_p_img.weight"), g("fc_p_img.bias")) if img else (g("fc_p.weight"), g("fc_p.bias"))
    fc_c = [(g(f"fc_c.{i}.weight"), g(f"fc_c.{i}.bias")) for i in range(5)]
    blocks = [(g(f"blocks.{i}.fc_0.weight"), g(f"blocks.{i}.fc_0.bias"),
               g(f"blocks.{i}.fc_1.weight"), g(f"blocks.{i}.fc_1.bias")) for i in range(5)]
    out2 = (g("fc_out_contact.weight"), g("fc_out_contact.bias")) if contact else None
    return ops.pack_decoder(pw, pb, fc_c, blocks, (g("fc_out.weight"), g("fc_out.bias")), out2, precision=precision)


def test_golden_points_lattice_slab():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    grid = torch.from_numpy(a["grid"]).to(dev)
    pts = torch.from_numpy(a["pts"]).to(dev)
    blob = _blob(sd, dev)
    ref = torch.from_numpy(a["logits"])
    got = ops.decode_fwd(grid, blob, pts=pts, precision=P).cpu()
    err = float((got - ref).abs().max())
    assert err <= TOL
    assert err > 0.0                      # it IS a different arithmetic; 0 would mean the f32 kernel ran
    got_l = ops.decode_fwd(grid, blob, lattice=(32, 1.1, 0, 32 ** 3), precision=P).cpu()   # brick tiles
    assert float((got_l - ref).abs().max()) <= TOL
    first, cnt = 5 * 32 * 32 + 7, 3 * 32 * 32 + 11
    got_s = ops.decode_fwd(grid, blob, lattice=(32, 1.1, first, cnt), precision=P).cpu()
    assert float((got_s - ref[:, first:first + cnt]).abs().max()) <= TOL


def test_golden_forward_img_contact_and_random_points():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    grid = torch.from_numpy(a["grid"]).to(dev)
    pts = torch.from_numpy(a["pts"]).to(dev)
    c_img = torch.from_numpy(a["c_img"].astype(np.float32)).to(dev)
    got = ops.decode_fwd(grid, _blob(sd, dev, img=True), pts=pts, c_img=c_img, precision=P).cpu()
    assert float((got - torch.from_numpy(a["logits_img"])).abs().max()) <= TOL
    o, oc = ops.decode_fwd(grid, _blob(sd, dev, contact=True), pts=pts, want_contact=True, precision=P)
    assert float((o.cpu() - torch.from_numpy(a["logits_contact"])).abs().max()) <= TOL
    assert float((oc.cpu() - torch.from_numpy(a["logits_contact2"])).abs().max()) <= TOL
    got = ops.decode_fwd(torch.from_numpy(a["grid2"]).to(dev), _blob(sd, dev),
                         pts=torch.from_numpy(a["prand"]).to(dev), precision=P).cpu()
    assert float((got - torch.from_numpy(a["logits_rand"])).abs().max()) <= TOL


@pytest.mark.parametrize("B,N,R", [(1, 1, 8), (3, 31, 8), (2, 33, 16), (1, 100000, 32), (4, 2048, 64)])
def test_seeded_vs_oracle_ragged_sizes(B, N, R):
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    _, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1000 + N)
    grid = torch.randn(B, 32, R, R, R, generator=g)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.3          # some beyond the padded box: clamps
    ref = orc.local_decoder_forward(sd, pts, grid)
    got = ops.decode_fwd(grid.to(dev), _blob(sd, dev), pts=pts.to(dev), precision=P).cpu()
    assert float((got - ref).abs().max()) <= TOL


def test_finger_ids_equal_dense_c_img():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    grid = torch.from_numpy(a["grid"]).to(dev)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(5, 32, generator=g).to(dev)
    ids = torch.full((1, 32 ** 3), 255, dtype=torch.uint8)
    hit = torch.rand(32 ** 3, generator=g) < 0.03
    ids[0, hit] = torch.randint(0, 5, (int(hit.sum()),), generator=g, dtype=torch.uint8)
    ids = ids.to(dev)
    dense = torch.zeros(1, 32 ** 3, 32, device=dev)
    dense[0, ids[0] != 255] = feats[ids[0][ids[0] != 255].long()]
    blob = _blob(sd, dev, img=True)
    lat = (32, 1.1, 0, 32 ** 3)
    by_id = ops.decode_fwd_ids(grid, blob, ids, feats, lattice=lat, precision=P)
    by_dense = ops.decode_fwd(grid, blob, c_img=dense, lattice=lat, precision=P)
    assert torch.equal(by_id, by_dense)
    exact = ops.decode_fwd(grid, _blob(sd, dev, img=True, precision="f32"), c_img=dense, lattice=lat)
    assert float((by_id - exact).abs().max()) <= TOL


def test_full_size_128_lattice_against_exact_f32_kernel():
    """2 097 152 points (BASELINE config 2): the split kernel against the exact-f32 kernel on the same
    inputs, and slab decomposition is bit-identical to the whole lattice (what sharding relies on)."""
    from vtaco_amd import ops
    from vtaco_amd.bench_util import build_scene
    scene = build_scene(0, torch.device("cuda:0"))
    dec, grid = scene["model"].decoder, scene["grid"]
    nx = 128
    exact = dec.decode_lattice(grid, nx, precision="f32")
    fast = dec.decode_lattice(grid, nx, precision=P)
    err = float((fast - exact).abs().max())
    assert 0.0 < err <= TOL, err
    half = nx ** 3 // 2
    lo = dec.decode_lattice(grid, nx, first=0, count=half, precision=P)
    hi = dec.decode_lattice(grid, nx, first=half, count=half, precision=P)
    assert torch.equal(torch.cat([lo, hi], dim=1), fast)


@pytest.mark.parametrize("precision", ["f32", "split"])
@pytest.mark.parametrize("nx", [128, 256])
def test_lds_staged_gather_is_bit_identical_to_direct_gather(nx, precision):
    """Whole brick-aligned lattices at nx >= 2R run the LDS-staged kernel (coalesced footprint copy +
    per-axis coordinate table); a slab that starts on an odd x-plane is not brick-aligned and runs the
    direct-gather kernel.  Same corner and FMA order: the logits must be bit-identical."""
    from vtaco_amd.bench_util import build_scene
    if precision == "f32" and P != "bf16x3":
        pytest.skip("the exact-f32 case does not depend on the split precision")
    precision = P if precision == "split" else precision
    scene = build_scene(0, torch.device("cuda:0"))
    dec, grid = scene["model"].decoder, scene["grid"]
    whole = dec.decode_lattice(grid, nx, precision=precision)
    plane = nx * nx
    # the split whole-lattice paths interleave two bricks per wave and add the block-end bias on the matrix core; their
    # logits match the other kernels to a few ulp (bf16x3: bit-identical for ~96 % of the points), the exact-f32 paths
    # are bit-identical throughout
    same = (lambda x, y: torch.equal(x, y)) if precision == "f32" else \
           (lambda x, y: float((x - y).abs().max()) <= 2e-6 and (precision != "bf16x3" or float((x == y).float().mean()) >= 0.9))
    for first_plane, planes in ((1, 3), (nx - 3, 3), (nx // 2 - 1, 2)):
        part = dec.decode_lattice(grid, nx, first=first_plane * plane, count=planes * plane, precision=precision)
        assert same(part, whole[:, first_plane * plane:(first_plane + planes) * plane])
    # batch of two grids, staged path
    g2 = torch.cat([grid, grid.flip(2)], dim=0).contiguous(memory_format=torch.channels_last_3d)
    both = dec.decode_lattice(g2, nx, precision=precision)
    assert torch.equal(both[0:1], whole)
    odd = dec.decode_lattice(g2, nx, first=plane, count=plane, precision=precision)
    assert same(odd, both[:, plane:2 * plane])


def test_training_forward_refuses_split_blob_path():
    from vtaco_amd import ops
    from vtaco_amd._lib import VtError
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    grid = torch.from_numpy(a["grid"]).to(dev)
    pts = torch.from_numpy(a["pts"]).to(dev)
    save = torch.empty(12 * 32 * pts.shape[1], device=dev)          # VT_SAVE_SLOTS x [N,32]
    with pytest.raises(VtError):
        ops.decode_fwd(grid, _blob(sd, dev), pts=pts, save=save, precision=P)
    with pytest.raises(VtError):
        ops.decode_fwd(grid, _blob(sd, dev), pts=pts, precision="fp8")


def test_f16x3_is_closer_to_f32_than_bf16x3():
    """hi + lo halves carry 21-22 mantissa bits against 16 for the bf16 pair: on the golden lattice the split-f16
    logits must sit within 5e-6 of the reference (bf16x3: ~1.6e-5) -- which also shows that the half subnormals
    of the lo parts (weights' lo ~1e-5, below the smallest normal half 6.1e-5) survive the matrix core."""
    from vtaco_amd import ops
    if P != "f16x3":
        pytest.skip("one comparison")
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    grid = torch.from_numpy(a["grid"]).to(dev)
    ref = torch.from_numpy(a["logits"])
    lat = (32, 1.1, 0, 32 ** 3)
    err = {p: float((ops.decode_fwd(grid, _blob(sd, dev, precision=p), lattice=lat, precision=p).cpu() - ref).abs().max())
           for p in ("f32", "bf16x3", "f16x3")}
    print("max |logit - reference| on g1:", err)
    assert err["f16x3"] <= 5e-6 and err["f16x3"] < err["bf16x3"], err


def test_f16x3_large_activations_saturate_not_overflow():
    """Activations beyond the half range: hi saturates at 65504 (round toward zero never produces inf), so the
    logits stay finite; the documented contract is |hidden activation| < 65504 for parity."""
    from vtaco_amd import ops
    if P != "f16x3":
        pytest.skip("one comparison")
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    big = {k: v.clone() for k, v in sd.items()}
    big["fc_p.bias"] = big["fc_p.bias"] + 1.0e5            # pushes net_0 past 65504
    grid = torch.from_numpy(a["grid"]).to(dev)
    pts = torch.from_numpy(a["pts"]).to(dev)
    ops.decode_range_status(reset=True)
    ops.decode_fwd(grid, _blob(sd, dev), lattice=(32, 1.1, 0, 32 ** 3), precision=P)
    assert ops.decode_range_status(reset=True) == 0            # ordinary activations leave the range guard alone
    # the guard is LOUD: every f16 kernel form (slot-pipelined lattice, unaligned slab on the generic kernel, point queries) reports
    for kw in (dict(lattice=(32, 1.1, 0, 32 ** 3)), dict(lattice=(32, 1.1, 7, 5000)), dict(pts=pts)):
        got = ops.decode_fwd(grid, _blob(big, dev), precision=P, **kw)
        assert bool(torch.isfinite(got).all())
        assert ops.decode_range_status(reset=True) & 1, kw
        assert ops.decode_range_status(reset=True) == 0        # reading with reset clears the word
