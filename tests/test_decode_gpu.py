"""GPU parity: vt_decode_fwd (through the C ABI) against the oracle and the
reference-generated golden vectors.  Tolerance: 1e-4 abs on f32 logits
(BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _blob(sd, dev, img=False, contact=False):
    from vtaco_amd import ops
    g = lambda k: sd[k].to(dev)
    pw, pb = (g("fc_p_img.weight"), g("fc_p_img.bias")) if img else (g("fc_p.weight"), g("fc_p.bias"))
    fc_c = [(g(f"fc_c.{i}.weight"), g(f"fc_c.{i}.bias")) for i in range(5)]
    blocks = [(g(f"blocks.{i}.fc_0.weight"), g(f"blocks.{i}.fc_0.bias"),
               g(f"blocks.{i}.fc_1.weight"), g(f"blocks.{i}.fc_1.bias")) for i in range(5)]
    out2 = (g("fc_out_contact.weight"), g("fc_out_contact.bias")) if contact else None
    return ops.pack_decoder(pw, pb, fc_c, blocks, (g("fc_out.weight"), g("fc_out.bias")), out2)


def test_golden_lattice_32_points_mode_and_lattice_mode():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    grid = torch.from_numpy(a["grid"]).to(dev)
    pts = torch.from_numpy(a["pts"]).to(dev)
    blob = _blob(sd, dev)
    ref = torch.from_numpy(a["logits"])
    got = ops.decode_fwd(grid, blob, pts=pts).cpu()
    assert float((got - ref).abs().max()) <= TOL
    got_l = ops.decode_fwd(grid, blob, lattice=(32, 1.1, 0, 32 ** 3)).cpu()
    assert float((got_l - ref).abs().max()) <= TOL
    # a slab of the lattice (what a rank evaluates when the lattice is sharded)
    first, cnt = 5 * 32 * 32 + 7, 3 * 32 * 32 + 11
    got_s = ops.decode_fwd(grid, blob, lattice=(32, 1.1, first, cnt)).cpu()
    assert float((got_s - ref[:, first:first + cnt]).abs().max()) <= TOL


def test_golden_forward_img_and_contact():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    grid = torch.from_numpy(a["grid"]).to(dev)
    pts = torch.from_numpy(a["pts"]).to(dev)
    c_img = torch.from_numpy(a["c_img"].astype(np.float32)).to(dev)
    got = ops.decode_fwd(grid, _blob(sd, dev, img=True), pts=pts, c_img=c_img).cpu()
    assert float((got - torch.from_numpy(a["logits_img"])).abs().max()) <= TOL
    o, oc = ops.decode_fwd(grid, _blob(sd, dev, contact=True), pts=pts, want_contact=True)
    assert float((o.cpu() - torch.from_numpy(a["logits_contact"])).abs().max()) <= TOL
    assert float((oc.cpu() - torch.from_numpy(a["logits_contact2"])).abs().max()) <= TOL


def test_golden_random_points_with_clamps_batch2():
    from vtaco_amd import ops
    a, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    got = ops.decode_fwd(torch.from_numpy(a["grid2"]).to(dev), _blob(sd, dev), pts=torch.from_numpy(a["prand"]).to(dev)).cpu()
    assert float((got - torch.from_numpy(a["logits_rand"])).abs().max()) <= TOL


@pytest.mark.parametrize("B,N,R", [(1, 1, 8), (3, 31, 8), (2, 33, 16), (1, 100000, 32), (4, 2048, 64)])
def test_seeded_vs_oracle_ragged_sizes(B, N, R):
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    _, sd = load_golden("g1_decode.npz")
    g = torch.Generator().manual_seed(100 + N)
    grid = torch.randn(B, 32, R, R, R, generator=g)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 1.2
    c_img = torch.randn(B, N, 32, generator=g)
    dev = torch.device("cuda:0")
    ref = orc.local_decoder_forward(sd, pts, grid)
    got = ops.decode_fwd(grid.to(dev), _blob(sd, dev), pts=pts.to(dev)).cpu()
    assert float((got - ref).abs().max()) <= TOL
    ref_i = orc.local_decoder_forward_img(sd, pts, grid, c_img)
    got_i = ops.decode_fwd(grid.to(dev), _blob(sd, dev, img=True), pts=pts.to(dev), c_img=c_img.to(dev)).cpu()
    assert float((got_i - ref_i).abs().max()) <= TOL


def test_channels_last_roundtrip_and_empty():
    from vtaco_amd import ops
    dev = torch.device("cuda:0")
    g = torch.randn(2, 32, 5, 6, 7, device=dev)
    cl = ops.grid_to_channels_last(g)
    assert cl.shape == g.shape and ops.is_channels_last_grid(cl)
    assert torch.equal(cl, g)
    assert torch.equal(ops.grid_from_channels_last(cl), g)
    _, sd = load_golden("g1_decode.npz")
    out = ops.decode_fwd(torch.randn(1, 32, 8, 8, 8, device=dev), _blob(sd, dev), pts=torch.zeros(1, 0, 3, device=dev))
    assert out.shape == (1, 0)


def test_full_size_128_lattice_properties():
    """BASELINE size: 128^3 lattice.  Size-independent checks: lattice mode ==
    points mode on the same coordinates (bit-exact), slab sharding == whole, and a
    random 4096-point subsample against the oracle."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    _, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    grid = torch.randn(1, 32, 64, 64, 64, generator=g)
    gd = ops.grid_to_channels_last(grid.to(dev))
    blob = _blob(sd, dev)
    nx = 128
    whole = ops.decode_fwd(gd, blob, lattice=(nx, 1.1, 0, nx ** 3))
    parts = [ops.decode_fwd(gd, blob, lattice=(nx, 1.1, r * nx ** 3 // 8, nx ** 3 // 8)) for r in range(8)]
    assert torch.equal(torch.cat(parts, dim=1), whole)
    pts = 1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
    via_pts = ops.decode_fwd(gd, blob, pts=pts.unsqueeze(0).to(dev))
    assert float((via_pts - whole).abs().max()) <= 2e-5       # linspace rounding only
    sel = torch.randint(0, nx ** 3, (4096,), generator=g)
    ref = orc.local_decoder_forward(sd, pts[sel].unsqueeze(0), grid)
    assert float((whole.cpu()[:, sel] - ref).abs().max()) <= TOL


@pytest.mark.parametrize("nx", [8, 12, 20, 30])
def test_lattice_batch2_and_odd_sizes(nx):
    """Lattice mode for B=2 scenes and lattice sizes that are / are not multiples of 4 (brick tiles
    vs linear tiles) against points mode on the same coordinates."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    _, sd = load_golden("g1_decode.npz")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(nx)
    grid = torch.randn(2, 32, 16, 16, 16, generator=g).to(dev)
    blob = _blob(sd, dev)
    pts = (1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).to(dev)
    lat = ops.decode_fwd(grid, blob, lattice=(nx, 1.1, 0, nx ** 3))
    via = ops.decode_fwd(grid, blob, pts=pts.unsqueeze(0).expand(2, -1, -1).contiguous())
    assert lat.shape == (2, nx ** 3)
    assert float((lat - via).abs().max()) <= 2e-5
    ref = orc.local_decoder_forward(sd, pts.cpu().unsqueeze(0).expand(2, -1, -1), grid.cpu())
    assert float((lat.cpu() - ref).abs().max()) <= TOL


def test_unsupported_shapes_fail_loudly():
    from vtaco_amd import ops
    from vtaco_amd._lib import VtError
    from vtaco_amd.conv_onet.models import decoder_dict
    dev = torch.device("cuda:0")
    dec = decoder_dict["simple_local"](dim=3, c_dim=120, hidden_size=256).to(dev)      # widths must be multiples of 32 (<= 256)
    with torch.no_grad(), pytest.raises(VtError, match="multiples of 32"):
        dec(torch.zeros(1, 4, 3, device=dev), {"grid": torch.zeros(1, 120, 4, 4, 4, device=dev)})
    dec = decoder_dict["simple_local"](dim=3, c_dim=128, hidden_size=256).to(dev)      # the class defaults: vt_decode_fwd_wide (round 3)
    with torch.no_grad():
        assert dec(torch.zeros(1, 4, 3, device=dev), {"grid": torch.zeros(1, 128, 4, 4, 4, device=dev)}).shape == (1, 4)
    _, sd = load_golden("g1_decode.npz")
    with pytest.raises(VtError):
        ops.decode_fwd(torch.zeros(1, 32, 4, 5, 6, device=dev), _blob(sd, dev), pts=torch.zeros(1, 4, 3, device=dev))


def test_sample_grid_over_lattice_slabs_equals_the_point_form():
    """vt_sample_grid in lattice mode (points generated in the kernel; slabs of the staged kernel's shape take its LDS-staged
    gather) returns the features of the explicit points, bit for bit: whole 64^3 and 128^3 lattices on an R = 32 / 64 grid, an
    aligned slab, and an unaligned range (direct gather)."""
    from vtaco_amd import ops
    from vtaco_amd.common import make_3d_grid
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for nx, R in ((64, 32), (128, 64)):
        grid = ops.grid_to_channels_last(torch.randn(1, 32, R, R, R, generator=g).to(dev))
        pts = (1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).to(dev)
        pair = 2 * nx * nx
        for first, count in ((0, nx ** 3), (3 * pair, 5 * pair), (7, 1000)):
            want = ops.sample_grid(grid, pts[first:first + count].unsqueeze(0))
            got = ops.sample_grid(grid, None, lattice=(nx, 1.1, first, count))
            assert got.shape == want.shape and torch.equal(got, want), (nx, first, count)


def test_lattice_decode_equals_point_decode_at_full_size():
    """The in-kernel lattice is torch's linspace to the bit (one rounding per element, as its CPU kernel evaluates start + step i), so
    the exact-f32 lattice decode of 64^3 / 128^3 / 256^3 slabs equals the decode of the explicit points of generation.py:155-157 --
    logit for logit -- not only at the goldens' 32^3 (where a separate multiply and add happens to round the same way)."""
    from vtaco_amd import ops
    from vtaco_amd.bench_util import build_scene
    from vtaco_amd.common import make_3d_grid
    dev = torch.device("cuda:0")
    sc = build_scene(0, dev)
    dec, grid = sc["model"].decoder, sc["grid"]
    for nx in (64, 128, 256):
        pts = (1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3))
        first, count = (nx // 2) * nx * nx, 4 * nx * nx                # four x-planes from the middle of the lattice
        with torch.no_grad():
            lat = dec.decode_lattice(grid, nx, first=first, count=count, precision="f32").reshape(-1)
            pnt = dec(pts[first:first + count].unsqueeze(0).to(dev), {"grid": grid}).reshape(-1)
        assert torch.equal(lat, pnt), nx


def test_lattice_launches_leave_their_clock_stamps():
    """bench.py's clock evidence (vt_decode_last_clock): every lattice launch stamps its workgroups' lifetimes with the constant-rate
    counter and workgroup 0's with the shader clock.  The stamps of the last launch: a clock between 0.5 and 3 GHz, all workgroups
    started within a few microseconds, and a span that agrees with the launch's HIP-event time."""
    from vtaco_amd import ops
    from vtaco_amd.bench_util import build_scene
    dev = torch.device("cuda:0")
    sc = build_scene(0, dev)
    dec, grid = sc["model"].decoder, sc["grid"]
    for prec in ("f16x3", "f32"):
        seen = []
        for attempt in range(5):                                    # (the event pair also times the host's launch: a hiccup there is retried)
            with torch.no_grad():
                for _ in range(20):
                    dec.decode_lattice(grid, 128, precision=prec)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dec.decode_lattice(grid, 128, precision=prec)
                e1.record()
                torch.cuda.synchronize()
            c = ops.decode_last_clock(workgroups=True)
            assert c is not None and c["workgroups"] >= 8 and len(c["wg_ticks"]) == c["workgroups"]
            assert 500.0 < c["shader_mhz"] < 3000.0, c["shader_mhz"]
            assert c["start_spread_us"] < 20.0 and c["wg_us_min"] > 0.0 and c["wg_us_max"] <= c["span_us"] + 1e-6
            ms = e0.elapsed_time(e1)
            seen.append((c["span_us"], ms))
            assert c["span_us"] * 1e-3 <= 1.05 * ms, (prec, seen)   # the kernel's own span never exceeds what the events saw
            if 0.5 * ms <= c["span_us"] * 1e-3:
                break
        else:
            raise AssertionError((prec, seen))
