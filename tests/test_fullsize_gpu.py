"""Full-size forms of BASELINE configs 3 and 5 inside the GPU suite (they used to live only in tools/bench_extra.py):
marching cubes at 256^3 against the C oracle, decode by finger id over the whole 256^3 lattice against the dense-c_img decode on
slabs, and the attention decoder over a whole 64^3 lattice in 2048-point chunks against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def scene():
    from vtaco_amd.bench_util import build_scene
    return build_scene(0, DEV)


def test_marching_cubes_256_vs_c_oracle(scene):
    """Config 5's mesh step at full size: the decoder's 256^3 logit grid -> HIP marching cubes, against the C restatement of
    scikit-image's Lewiner algorithm on the same volume: faces (vertex numbering) bit-exact, coordinates <= 1e-5."""
    from oracle import mc
    from vtaco_amd import ops
    nx = 256
    with torch.no_grad():
        vol = scene["model"].decoder.decode_lattice(scene["grid"], nx, box=1.1, precision="bf16x3").reshape(nx, nx, nx)
    v, f, lvl = ops.marching_cubes(vol, None)
    rv, rf, rl = mc.marching_cubes(vol.cpu().numpy())
    assert lvl == rl and f.shape[0] > 1_000_000
    assert np.array_equal(f.cpu().numpy(), rf)
    assert np.abs(v.cpu().numpy() - rv).max() <= 1e-5
    # and at an explicit level, with the generator's rescale applied on the device
    v2, f2, _ = ops.marching_cubes(vol, 0.3, rescale=(nx / 2, 1.1 / nx))
    rv2, rf2, _ = mc.marching_cubes(vol.cpu().numpy(), 0.3)
    assert np.array_equal(f2.cpu().numpy(), rf2)
    assert np.abs(v2.cpu().numpy() - (rv2 - nx / 2) * (1.1 / nx)).max() <= 1e-6


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_decode_by_finger_id_256_equals_dense_on_slabs(scene, precision):
    """Config 5's decode at full size: finger ids for all 16.7 M lattice points (vt_tactile_assign) + decode by id
    (vt_decode_fwd_ids) against the dense [1, n, 32] c_img_all the reference would build, on three slabs of the lattice
    (first planes, a slab through the fingertips, the last planes): bit-identical logits."""
    from vtaco_amd import ops
    nx = 256
    dec, grid = scene["model"].decoder, scene["grid"]
    g = torch.Generator().manual_seed(3)
    tips = torch.randn(5, 1, 3, generator=g)
    tips = (0.3 * tips / tips.norm(dim=-1, keepdim=True)).to(DEV)
    success = torch.tensor([1, 1, 0, 1, 1], dtype=torch.uint8, device=DEV)
    feats = torch.randn(5, 32, generator=g).to(DEV)
    with torch.no_grad():
        ids = ops.tactile_assign(tips, success, 'nearest', 0.05, lattice=(nx, 1.1, 0, nx ** 3))
        whole = dec.decode_lattice_ids(grid, nx, ids, feats, box=1.1, precision=precision).reshape(-1)
        assert int((ids != 255).sum()) > 1000 and int((ids[0] == 2).sum()) == 0          # the failed touch assigns nothing
        table = torch.cat([feats, feats.new_zeros(1, 32)])
        hit = torch.nonzero(ids[0] != 255).squeeze(1)
        mid = int(hit[hit.numel() // 2]) // (2 * nx * nx) * (2 * nx * nx)
        for first in (0, mid, nx ** 3 - 4 * nx * nx):
            count = 4 * nx * nx
            row = ids[0, first:first + count].long()
            c_img = table[torch.where(row == 255, torch.full_like(row, 5), row)].unsqueeze(0)
            dense = dec.decode_lattice(grid, nx, box=1.1, first=first, count=count, c_img=c_img, precision=precision).reshape(-1)
            assert torch.equal(dense, whole[first:first + count]), first
        assert bool((ids[0, mid:mid + 4 * nx * nx] != 255).any())


@pytest.mark.parametrize("nx,R", [(64, 32), (128, 64)])
def test_attention_decoder_over_a_whole_lattice_vs_oracle(nx, R):
    """Config 3's decoder at lattice scale (128^3 = BASELINE config 3 at its literal size: 1024 chunks): ``decoder: attention_local``
    evaluates the 64^3 lattice in 128 chunks of 2048 points
    (TransformerFusion couples the points of a chunk), features by finger id; the oracle (torch CPU) re-computes eight of the
    chunks -- first, last, and the ones holding the most touched points -- from the same inputs: <= 1e-4 on the logits.

    A chunk NO touched point falls into is ill-conditioned by construction: its tactile features are all zero, the decoder's
    self-attention output is constant over the chunk, and InstanceNorm divides an exactly-zero variance by sqrt(1e-5) -- f32
    rounding noise times 316.  The oracle itself, run in float32, misses its own float64 value by 1e-4 .. 2e-4 there (measured),
    so the yardstick is the oracle in FLOAT64 and the bar max(1e-4, 3 x the f32 oracle's own error)."""
    from oracle import vtaco_oracle as orc
    from vtaco_amd import ops
    from vtaco_amd.bench_util import randomise_fc1
    from vtaco_amd.common import make_3d_grid
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    torch.manual_seed(0)
    adec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32).eval()
    randomise_fc1(adec, 3)
    g = torch.Generator().manual_seed(4)
    grid = torch.randn(1, 32, R, R, R, generator=g)
    model = ConvolutionalOccupancyNetwork(adec, None, device=DEV)
    gen = Generator3D(model, device=DEV, resolution0=nx // 4, padding=0.1, points_batch_size=2048, with_img=True)
    chunk = 2048
    tips = torch.randn(5, 1, 3, generator=g)
    tips = 0.3 * tips / tips.norm(dim=-1, keepdim=True)
    success = torch.tensor([1, 0, 1, 1, 1], dtype=torch.uint8)
    feats = torch.randn(5, 32, generator=g)
    setup = {'feats': feats, 'anchors': tips, 'success': success, 'mode': 'nearest', 'radius': 0.08,
             'count': torch.ones(5, dtype=torch.int32)}
    with torch.no_grad():
        c = {"grid": ops.grid_to_channels_last(grid.to(DEV))}
        got = gen._eval_lattice_tactile(c, nx, setup).cpu()
        ids = ops.tactile_assign(tips.to(DEV), success.to(DEV), 'nearest', 0.08, lattice=(nx, 1.1, 0, nx ** 3))[0].cpu().long()
    assert got.shape == (nx ** 3,) and int((ids != 255).sum()) > 500
    per_chunk = (ids != 255).reshape(-1, chunk).sum(1)
    picks = sorted(set([0, nx ** 3 // chunk - 1] + [int(i) for i in torch.argsort(per_chunk, descending=True)[:6]]))
    pts = 1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
    table = torch.cat([feats, torch.zeros(1, 32)])
    sd = {k: v.detach().cpu() for k, v in adec.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd.items()}
    seen_touched = 0
    for ch in picks:
        lo = ch * chunk
        row = ids[lo:lo + chunk]
        c_img = table[torch.where(row == 255, torch.full_like(row, 5), row)].unsqueeze(0)
        p = pts[lo:lo + chunk].unsqueeze(0)
        ref64 = orc.attention_decoder_forward_img(sd64, p, grid.double(), c_img.double())[0]
        ref32 = orc.attention_decoder_forward_img(sd, p, grid, c_img)[0]
        own = float((ref32.double() - ref64).abs().max())            # the reference arithmetic's own rounding error here
        err = float((got[lo:lo + chunk].double() - ref64).abs().max())
        touched = int(per_chunk[ch])
        if touched:
            seen_touched += 1
            assert own <= 2e-5 and err <= 1e-4, (ch, touched, own, err)
        else:
            assert err <= max(1e-4, 3 * own), (ch, touched, own, err)
    assert seen_touched >= 4


def test_attention_decoder_chunks_in_batches_equal_chunks_one_by_one():
    """``_eval_lattice_fused`` hands the kernels whole chunks as a batch (FUSED_CHUNKS_PER_CALL at a time) and the ragged last chunk
    on its own: the same logits, bit for bit, as one ``forward_img`` call per chunk (what the reference's ``eval_points`` loop does) --
    a slab of 7 chunks and a tail with 3 chunks per call, and with all of them in one call."""
    from vtaco_amd import ops
    from vtaco_amd.bench_util import randomise_fc1
    from vtaco_amd.common import make_3d_grid
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    torch.manual_seed(1)
    adec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32).eval()
    randomise_fc1(adec, 5)
    g = torch.Generator().manual_seed(9)
    model = ConvolutionalOccupancyNetwork(adec, None, device=DEV)
    nx, chunk = 32, 512
    first, count = 3 * chunk, 7 * chunk + 100
    gen = Generator3D(model, device=DEV, resolution0=8, padding=0.1, points_batch_size=chunk, with_img=True)
    c = {"grid": ops.grid_to_channels_last(torch.randn(1, 32, 16, 16, 16, generator=g).to(DEV))}
    feats = torch.randn(5, 32, generator=g).to(DEV)
    ids = torch.randint(0, 6, (1, count), generator=g).to(torch.uint8)
    ids[ids == 5] = 255
    ids = ids.to(DEV)
    pts = ((1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3))[first:first + count]).to(DEV)
    table = torch.cat([feats, feats.new_zeros(1, 32)])
    row = torch.where(ids[0] == 255, torch.full_like(ids[0], 5), ids[0]).long()
    with torch.no_grad():
        one_by_one = torch.cat([model.decoder.forward_img(pts[lo:lo + chunk].unsqueeze(0), c, table[row[lo:lo + chunk]].unsqueeze(0))[0]
                                for lo in range(0, count, chunk)])
        for per_call in (3, 256):
            gen.FUSED_CHUNKS_PER_CALL = per_call
            got = gen._eval_lattice_fused(c, nx, ids, feats, first, count)
            assert got.shape == (count,) and torch.equal(got, one_by_one), per_call
            # the generic entry point (the reference's eval_points: arbitrary points, dense features, a CPU tensor back)
            ev = gen.eval_points(pts.cpu(), c, table[row].unsqueeze(0).cpu())
            assert ev.device.type == "cpu" and torch.equal(ev, one_by_one.cpu()), per_call


def test_chunks_without_tactile_features_skip_the_fuser_bit_for_bit():
    """fuse(0, c) = 0: a chunk no point of which carries a tactile feature needs no attention (InstanceNorm over a chunk whose
    rows are all equal leaves exact zeros, twice -- Generator3D._eval_lattice_fused).  The 64^3 lattice with four fingertips
    touches a minority of its 128 chunks; the logits with the shortcut are the logits of sending every chunk through the three
    attention units, bit for bit, and the fuser itself returns exact zeros on an untouched chunk."""
    from vtaco_amd import ops
    from vtaco_amd.bench_util import randomise_fc1
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    torch.manual_seed(0)
    adec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32).eval()
    randomise_fc1(adec, 3)
    g = torch.Generator().manual_seed(4)
    nx, R, chunk = 64, 32, 2048
    grid = torch.randn(1, 32, R, R, R, generator=g)
    model = ConvolutionalOccupancyNetwork(adec, None, device=DEV)
    gen = Generator3D(model, device=DEV, resolution0=nx // 4, padding=0.1, points_batch_size=chunk, with_img=True)
    tips = torch.randn(5, 1, 3, generator=g)
    setup = {'feats': torch.randn(5, 32, generator=g), 'anchors': 0.3 * tips / tips.norm(dim=-1, keepdim=True),
             'success': torch.tensor([1, 0, 1, 1, 1], dtype=torch.uint8), 'mode': 'nearest', 'radius': 0.08,
             'count': torch.ones(5, dtype=torch.int32)}
    with torch.no_grad():
        c = {"grid": ops.grid_to_channels_last(grid.to(DEV))}
        assert gen.skip_untouched_chunks
        fast = gen._eval_lattice_tactile(c, nx, setup)
        gen.skip_untouched_chunks = False
        dense = gen._eval_lattice_tactile(c, nx, setup)
        ids = ops.tactile_assign(setup['anchors'].to(DEV), setup['success'].to(DEV), 'nearest', 0.08, lattice=(nx, 1.1, 0, nx ** 3))[0]
        touched = (ids != 255).reshape(-1, chunk).any(dim=1)
        print(f"{int(touched.sum())} of {touched.numel()} chunks carry tactile features")
        assert 0 < int(touched.sum()) < touched.numel() // 2
        assert torch.equal(fast, dense)
        # the fuser on chunks without features: exact zeros for every chunk size the kernels distinguish (fp8-corrected tiles at
        # N >= 512, half pairs below, a ragged size)
        for n in (2048, 512, 256, 100):
            feat = torch.randn(3, n, 32, generator=g).to(DEV)
            z = adec.fuser(torch.zeros(3, n, 32, device=DEV), 1, feat, 1)
            assert float(z.abs().max()) == 0.0, n


@pytest.mark.parametrize("variant", ["vtaco", "vtacoh"])
def test_tactile_routes_with_overlapped_encoders_equal_the_sequential_ones(variant):
    """generate_obj_mesh_wnf(with_img) replays the scene's independent encoders (shape, tactile features, hand) on three HIP streams at
    once (Generator3D._generate_tactile); the mesh must be the one the sequential route (VTACO_SCENE_OVERLAP=0) gives, call after
    call -- eager first calls, the capturing call and graph replays all included -- on the shipped model sections."""
    import os
    import numpy as np
    from vtaco_amd.bench_util import build_tactile_scene
    from vtaco_amd.conv_onet.generation import Generator3D
    model, data, depth_origin = build_tactile_scene(torch.device(DEV), variant=variant)
    kw = dict(device=torch.device(DEV), resolution0=16, padding=0.1, with_img=True, encode_t2d=variant == "vtaco", depth_origin=depth_origin)

    def run(gen, n):
        out = []
        for _ in range(n):
            np.random.seed(11)                                      # the t2d rule draws from numpy's global generator
            m = gen.generate_obj_mesh_wnf(data)
            out.append((m.vertices.clone(), m.faces.clone()))
        return out
    saved = os.environ.get("VTACO_SCENE_OVERLAP")
    try:
        os.environ["VTACO_SCENE_OVERLAP"] = "0"
        ref = run(Generator3D(model, **kw), 2)
        os.environ["VTACO_SCENE_OVERLAP"] = "1"
        gen = Generator3D(model, **kw)
        got = run(gen, 8)
        assert getattr(gen, "_sides", None) is not None             # the side streams were used
    finally:
        if saved is None:
            os.environ.pop("VTACO_SCENE_OVERLAP", None)
        else:
            os.environ["VTACO_SCENE_OVERLAP"] = saved
    # (the tactile features come from MIOpen's Resnet18, whose sums move in the last bits from call to call -- sequential route included,
    # tools/probe/overlap_diag.py -- so the vertices are compared to 1e-5; the faces and the vertex count exactly)
    assert ref[0][0].shape[0] > 0
    for v, f in ref[1:] + got:
        assert v.shape == ref[0][0].shape and torch.equal(f, ref[0][1])
        assert float((v - ref[0][0]).abs().max()) <= 1e-5
