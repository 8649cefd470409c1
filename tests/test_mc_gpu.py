"""GPU parity: HIP marching cubes (vt_mc_count / vt_mc_emit through the C ABI) against
scikit-image's own outputs (g7_mc.npz) and the C oracle.  Faces -- i.e. the vertex
numbering -- bit-exact; vertex coordinates <= 1e-5 (north_star: indices bit-exact)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
_Z = np.load(GOLDEN + "/g7_mc.npz")
Z = {k: _Z[k] for k in _Z.files}
CASES = sorted({k.rsplit(".", 1)[0] for k in Z} - {"cells"})
DEV = "cuda:0"


def run(vol, level=None, **kw):
    from vtaco_amd import ops
    v, f, lvl = ops.marching_cubes(torch.from_numpy(np.ascontiguousarray(vol, np.float32)).to(DEV), level, **kw)
    return v.cpu().numpy(), f.cpu().numpy(), lvl


@pytest.mark.parametrize("name", CASES)
def test_matches_skimage_golden(name):
    vol, level = Z[name + ".vol"], float(Z[name + ".level"])
    v, f, lvl = run(vol, level if "@" in name else None)
    assert lvl == level
    assert f.shape == Z[name + ".faces"].shape and np.array_equal(f, Z[name + ".faces"])
    assert v.shape == Z[name + ".verts"].shape and np.abs(v - Z[name + ".verts"]).max() <= 1e-5


def test_single_cells_all_subcases():
    vols, nf, nv = Z["cells.vols"], Z["cells.nf"], Z["cells.nv"]
    for i in range(0, len(vols), 3):
        if nf[i] == 0:
            with pytest.raises(RuntimeError):
                run(vols[i], 0.0)
            continue
        v, f, _ = run(vols[i], 0.0)
        assert np.array_equal(f, Z["cells.faces"][i, :nf[i]]), i
        assert np.abs(v - Z["cells.verts"][i, :nv[i]]).max() <= 1e-5, i


@pytest.mark.parametrize("shape,seed", [((64, 64, 64), 0), ((128, 128, 128), 1), ((33, 70, 17), 2), ((2, 2, 300), 3)])
def test_noisy_volumes_vs_oracle(shape, seed):
    """Random-noise volumes hit the ambiguous MC33 cases in ~17 % of the cells."""
    from oracle import mc
    rng = np.random.RandomState(seed)
    vol = rng.randn(*shape).astype(np.float32)
    if shape[0] == 128:           # smoother field, like a logit grid: blur once
        vol = (vol + np.roll(vol, 1, 0) + np.roll(vol, 1, 1) + np.roll(vol, 1, 2)) / 4
    rv, rf, rl = mc.marching_cubes(vol)
    v, f, lvl = run(vol)
    assert lvl == rl
    assert np.array_equal(f, rf)
    assert np.abs(v - rv).max() <= 1e-5


def test_rescale_explicit_level_and_capacity_mode():
    from oracle import mc, vtaco_oracle as orc
    from vtaco_amd import ops
    vol = Z["logits32.vol"]
    rv, rf, _ = mc.marching_cubes(vol, -0.1)
    v, f, _ = run(vol, -0.1, rescale=(16.0, 1.1 / 32))
    assert np.array_equal(f, rf)
    assert np.abs(v - orc.mesh_rescale(rv, 32)).max() <= 1e-6
    # capacity mode: no host sync; counts stay in the workspace header
    vd, fd, ws = ops.marching_cubes(torch.from_numpy(vol).to(DEV), -0.1, capacity=(len(rv) + 100, len(rf) + 100))
    counts = ws[8:16].view(torch.int32).cpu().numpy()
    assert counts[0] == len(rv) and counts[1] == len(rf)
    assert np.array_equal(fd[:len(rf)].cpu().numpy(), rf)


def test_errors():
    from vtaco_amd import ops
    with pytest.raises(RuntimeError):
        run(np.zeros((4, 4, 4), np.float32), 1.0)
    with pytest.raises(ValueError):
        ops.marching_cubes(torch.zeros(1, 4, 4, device=DEV))


def test_speculative_emit_small_large_small_surface():
    """Same volume shape three times: the first call sizes the outputs from the counts, the second
    (a far larger surface) overflows the speculative buffers and is re-emitted at the exact size, the
    third runs entirely in the speculative buffers -- all three numbered exactly as the oracle."""
    from oracle import mc
    from vtaco_amd import ops
    ops._mc_guess.clear()
    n = 40
    zz, yy, xx = np.meshgrid(*(np.arange(n, dtype=np.float32),) * 3, indexing="ij")
    ball = lambda r: (r - np.sqrt((xx - 19.3) ** 2 + (yy - 20.1) ** 2 + (zz - 18.7) ** 2)).astype(np.float32)
    noise = np.random.RandomState(4).randn(n, n, n).astype(np.float32)
    for vol in (ball(5.0), noise, ball(7.5), ball(4.0)):
        rv, rf, _ = mc.marching_cubes(vol, 0.0)
        v, f, _ = run(vol, 0.0)
        assert np.array_equal(f, rf) and v.shape == rv.shape and np.abs(v - rv).max() <= 1e-5
    assert (0, n, n, n) in ops._mc_guess or (torch.cuda.current_device(), n, n, n) in ops._mc_guess


def test_read_back_tokens_are_counted():
    """vt_mc_read_counts_begin hands out sixteen page-locked slots: a seventeenth outstanding read-back is refused instead of
    overwriting one that is still waited for, a spent token is refused, and every slot comes back into use."""
    import ctypes
    from vtaco_amd import _lib, ops
    lib = _lib.load()
    vol = torch.randn(12, 12, 12, device=DEV)
    ws = ops.mc_count(vol, 0.0)
    wp, st = ctypes.c_void_p(ws.data_ptr()), ops.stream_ptr()
    want = ops.marching_cubes(vol, 0.0)
    toks = []
    for _ in range(16):
        t = ctypes.c_int(-1)
        assert lib.vt_mc_read_counts_begin(wp, st, ctypes.byref(t)) == 0
        toks.append(t.value)
    assert sorted(toks) == list(range(16))
    t = ctypes.c_int(-1)
    assert lib.vt_mc_read_counts_begin(wp, st, ctypes.byref(t)) != 0 and b"in flight" in lib.vt_last_error()
    nv, nf, lvl = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    for k in toks:
        assert lib.vt_mc_read_counts_end(k, ctypes.byref(nv), ctypes.byref(nf), ctypes.byref(lvl)) == 0
        assert (nv.value, nf.value) == (want[0].shape[0], want[1].shape[0])
    assert lib.vt_mc_read_counts_end(toks[3], ctypes.byref(nv), ctypes.byref(nf), ctypes.byref(lvl)) != 0
    got = ops.marching_cubes(vol, 0.0)                  # the ring is whole again
    assert torch.equal(got[1], want[1])


def test_counts_by_notification_equal_the_copied_counts():
    """vt_mc_count_notify (the scan kernel writes the counts and a sequence number into a page-locked slot the host polls) against
    vt_mc_count + vt_mc_read_counts: the same counts, level and mesh over many calls on alternating shapes -- slots are reused, their
    sequence numbers advance, an event-waited read-back in between leaves the polled ones alone."""
    import ctypes
    from vtaco_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    vols = [torch.randn(n, n, n, generator=g).to(DEV) for n in (12, 20, 33)]
    for it in range(40):
        vol = vols[it % 3]
        level = None if it % 2 else 0.1
        ws = ops.mc_count(vol, level)
        nv, nf, lvl = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        assert lib.vt_mc_read_counts(ctypes.c_void_p(ws.data_ptr()), ctypes.byref(nv), ctypes.byref(nf), ctypes.byref(lvl), ops.stream_ptr()) == 0
        ref_v, ref_f, _ = ops.mc_emit(vol, ws, None, capacity=(nv.value, nf.value))
        ref_v, ref_f = ref_v.clone(), ref_f.clone()
        ws2, tok = ops.mc_count_notify(vol, level)
        nv2, nf2, lvl2 = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        assert lib.vt_mc_read_counts_end(tok, ctypes.byref(nv2), ctypes.byref(nf2), ctypes.byref(lvl2)) == 0
        assert (nv2.value, nf2.value, lvl2.value) == (nv.value, nf.value, lvl.value)
        v, f, l3 = ops.marching_cubes(vol, level)                    # notify + speculative emit
        assert l3 == lvl.value and torch.equal(f, ref_f) and torch.equal(v, ref_v)


def test_echo_slots_are_reused_after_release():
    """The 64 page-locked echo slots of a process are a pool: a released slot (its graph is gone) is what the next request gets, so a
    long run that re-captures scenes (LRU eviction, a generator per validation pass) never exhausts them."""
    from vtaco_amd import ops
    from vtaco_amd._lib import VtError
    held = []
    try:
        while True:
            held.append(ops.mc_echo_slot())
    except VtError:
        pass
    assert 1 <= len(held) <= 64
    victim = held.pop(len(held) // 2)
    ops.mc_echo_release(victim)
    assert ops.mc_echo_slot() == victim                    # the freed block, not a 65th
    held.append(victim)
    with pytest.raises(VtError):
        ops.mc_echo_slot()
    for t in held:
        ops.mc_echo_release(t)
    with pytest.raises(VtError):
        ops.mc_echo_release(held[0])                        # already free
