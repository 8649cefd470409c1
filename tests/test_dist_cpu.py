"""CPU, world_size 2, gloo: the N>1 plumbing (slab sharding + all-gather of logits,
flat-bucket gradient all-reduce with parameters that got no gradient)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vtaco_amd.dist import GradAllReduce, all_gather_slabs, decode_lattice_sharded, lattice_align, slab_of


def test_slabs_partition_exactly():
    for total in (1, 31, 32, 1000, 32 ** 3, 128 ** 3, 17 ** 3):
        for world in (1, 2, 3, 8):
            spans = [slab_of(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1 and (c0 % 32 == 0 or f1 == total)


def test_slabs_with_empty_tails_still_partition():
    """Rounding the share up to the alignment leaves trailing ranks empty (nx=32 over 7 ranks in plane pairs; a
    100 000-point chunk alignment at 128^3 over 8 ranks): the spans still tile [0,total) and the empty ones are (total, 0)."""
    seen_empty = 0
    for nx in (8, 11, 16, 32, 128):
        for world in range(1, 9):
            for align in (lattice_align(nx, world), 32, 2048, 100000):
                total = nx ** 3
                spans = [slab_of(total, r, world, align) for r in range(world)]
                assert sum(c for _, c in spans) == total and spans[0][0] == 0
                for (f0, c0), (f1, c1) in zip(spans, spans[1:]):
                    assert f0 + c0 == f1
                    if c0 == 0:
                        assert c1 == 0 and f0 == total            # once empty, empty to the end
                seen_empty += sum(c == 0 for _, c in spans)
    assert slab_of(32 ** 3, 6, 7, lattice_align(32, 7))[1] == 0      # the advisor's example
    assert slab_of(128 ** 3, 7, 8, 100000) == (128 ** 3, 0)            # 100 000-point chunks at 128^3 over 8 ranks
    assert seen_empty > 0


def test_lattice_slabs_are_whole_plane_pairs_when_possible():
    """128^3 / 256^3 over 2, 4, 8 ranks: every slab starts and ends on a pair of x-planes (the brick-tiled,
    LDS-staged decode kernels' alignment); tiny lattices fall back to the 32-point tile."""
    for nx, world in ((128, 2), (128, 8), (256, 8), (128, 3), (32, 8)):
        align = lattice_align(nx, world)
        assert align == 2 * nx * nx
        spans = [slab_of(nx ** 3, r, world, align) for r in range(world)]
        assert sum(c for _, c in spans) == nx ** 3
        assert all(f % align == 0 and (c % align == 0) for f, c in spans)
    assert lattice_align(8, 8) == 32 and lattice_align(11, 2) == 2 * 121


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Net(torch.nn.Module):
    """Two heads (only one is used per step, like fc_p / fc_p_img), a layer used twice (shared weights, like the
    fuser's encoder / decoder self-attention) and a parameter no step ever touches."""

    def __init__(self):
        super().__init__()
        self.head_a, self.head_b = torch.nn.Linear(6, 16), torch.nn.Linear(6, 16)
        self.shared = torch.nn.Linear(16, 16)
        self.out = torch.nn.Linear(16, 1)
        self.unused = torch.nn.Parameter(torch.ones(3))

    def forward(self, x, use_b):
        h = (self.head_b if use_b else self.head_a)(x)
        h = self.shared(torch.relu(self.shared(torch.relu(h))))
        return self.out(h).squeeze(-1)


def _ddp_steps(rank, world, steps=4):
    """Bucketed, hook-launched all-reduce under a real backward + Adam: after every step the weights equal the ones a single
    process reaches with the mean of the per-rank gradients; tiny buckets (1 KB) force several collectives per step, the
    bucket order is rebuilt from the observed hook order after step 1, the unused head / parameter count as zeros."""
    def data(r, step):
        g = torch.Generator().manual_seed(100 * step + r)
        return torch.randn(32, 6, generator=g), torch.randn(32, generator=g)
    torch.manual_seed(7)
    net, ref = _Net(), _Net()
    ref.load_state_dict(net.state_dict())
    opt, opt_ref = torch.optim.Adam(net.parameters(), lr=1e-2), torch.optim.Adam(ref.parameters(), lr=1e-2)
    sync = GradAllReduce(net.parameters(), bucket_bytes=1024)
    ok = True
    for step in range(steps):
        use_b = step % 2 == 1
        opt.zero_grad(set_to_none=(step != 2))             # both zero_grad flavours
        x, y = data(rank, step)
        torch.nn.functional.l1_loss(net(x, use_b), y).backward()
        sync()
        opt.step()
        opt_ref.zero_grad()
        for r in range(world):
            x, y = data(r, step)
            (torch.nn.functional.l1_loss(ref(x, use_b), y) / world).backward()
        for p in ref.parameters():                         # parameters without gradient count as zeros on the DDP side
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        opt_ref.step()
        ok = ok and all(torch.allclose(a, b, atol=1e-6, rtol=1e-5) for a, b in zip(net.parameters(), ref.parameters()))
        ok = ok and all(p.grad is not None for p in net.parameters())
    ok = ok and sync.stats["buckets"] >= 3 and sync._rebuilt and sync.stats["launched_in_backward"] > 0
    # the ranks hold identical weights
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    return ok and all(torch.equal(both[0], t) for t in both)


def _accumulation_steps(rank, world):
    """Two backward() calls per step: inside ``no_sync()`` the first one accumulates locally and the second launches the buckets
    with the totals (the synchronised gradient is the mean over ranks of the SUM of both micro-batches, with .grad already a view
    of its bucket from the step before); outside ``no_sync()`` the second backward raises instead of synchronising half a step."""
    def data(r, step, micro):
        g = torch.Generator().manual_seed(1000 * step + 10 * micro + r)
        return torch.randn(16, 6, generator=g), torch.randn(16, generator=g)
    torch.manual_seed(11)
    net, ref = _Net(), _Net()
    ref.load_state_dict(net.state_dict())
    sync = GradAllReduce(net.parameters(), bucket_bytes=1024)
    ok = True
    for step in range(3):
        net.zero_grad(set_to_none=False) if step else None
        ref.zero_grad()
        x, y = data(rank, step, 0)
        with sync.no_sync():
            torch.nn.functional.l1_loss(net(x, False), y).backward()
        x, y = data(rank, step, 1)
        torch.nn.functional.l1_loss(net(x, True), y).backward()
        sync()
        for r in range(world):
            for micro, use_b in ((0, False), (1, True)):
                x, y = data(r, step, micro)
                (torch.nn.functional.l1_loss(ref(x, use_b), y) / world).backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            want = q.grad if q.grad is not None else torch.zeros_like(q)
            ok = ok and torch.allclose(p.grad, want, atol=1e-6, rtol=1e-5)
    # the misuse is loud
    net.zero_grad(set_to_none=False)
    x, y = data(rank, 9, 0)
    torch.nn.functional.l1_loss(net(x, False), y).backward()
    try:
        torch.nn.functional.l1_loss(net(x, False), y).backward()
        ok = False
    except RuntimeError as e:
        ok = ok and "no_sync" in str(e)
    sync.remove_hooks()
    return ok


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok1 = True
        def slab(first, count):
            assert count > 0                              # an empty slab must not reach the kernels
            return torch.arange(first, first + count, dtype=torch.float32) * 0.5
        for nx in (11, 8, 16):                            # ragged plane-pair slabs, 32-point slabs, even plane pairs
            full = decode_lattice_sharded(slab, nx)
            ok1 = ok1 and torch.equal(full, torch.arange(nx ** 3, dtype=torch.float32) * 0.5)
        # alignment larger than half the lattice: rank 1's slab is empty, it still joins the all-gather
        full = decode_lattice_sharded(slab, 8, align=400)
        ok1 = ok1 and slab_of(512, 1, 2, 400) == (400, 112) and torch.equal(full, torch.arange(512, dtype=torch.float32) * 0.5)
        full = decode_lattice_sharded(slab, 8, align=512)
        ok1 = ok1 and slab_of(512, 1, 2, 512) == (512, 0) and torch.equal(full, torch.arange(512, dtype=torch.float32) * 0.5)
        # gradient all-reduce: rank-dependent grads, one parameter without a gradient on rank 1
        torch.manual_seed(0)
        a, b, c = (torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
                   torch.nn.Parameter(torch.zeros(2, 2)))
        a.grad = torch.full((5, 3), float(rank + 1))
        b.grad = torch.arange(7.0) * (rank + 1)
        if rank == 0:
            c.grad = torch.ones(2, 2)
        GradAllReduce([a, b, c], overlap=False)()
        ok2 = torch.allclose(a.grad, torch.full((5, 3), 1.5)) and torch.allclose(b.grad, torch.arange(7.0) * 1.5) \
            and torch.allclose(c.grad, torch.full((2, 2), 0.5))
        ok2 = ok2 and _ddp_steps(rank, world)
        ok2 = ok2 and _accumulation_steps(rank, world)
        q.put((rank, bool(ok1), bool(ok2)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """``bench.py --gpus 2`` without a launcher starts two ranks of itself before touching any GPU (here: --dry-run, the
    launcher / rendezvous / max-over-ranks plumbing only -- no kernel runs and the value is null) and reports n_gpus == 2;
    under a launcher a --gpus that disagrees with WORLD_SIZE is an error, not a silent one-rank run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["dry_run"] is True and res["value"] is None and res["steps"] == 4
    assert res["wall_s"] >= 0.02                       # the max over the ranks (rank 1 sleeps 20 ms), not rank 0's 10 ms
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def _guard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vtaco_amd.conv_onet.generation import _reduce_or
        # whole group: the OR of the ranks' words
        got_all = _reduce_or(1 if rank == 0 else 4, None, None)
        # a sub-group of rank 0 alone (every rank creates it, only its member uses it): no other rank is waited for
        g0 = dist.new_group([0])
        got_sub = _reduce_or(2, g0, None) if rank == 0 else None
        dist.barrier()
        q.put((rank, got_all, got_sub))
    finally:
        dist.destroy_process_group()


def test_range_guard_word_is_reduced_over_the_given_group_only():
    """generate_obj_mesh_sharded's flag exchange: bitwise OR over the ranks of the group the call was given (per-bit MAX: RCCL
    has no BOR), a one-rank sub-group exchanges nothing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 5, 2), (1, 5, None)]


def _agree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        from types import SimpleNamespace
        from vtaco_amd import ops
        from vtaco_amd.conv_onet import generation as gen
        # no GPU here: the device's status word is a local variable, the scene a counter
        state = {"word": 0, "runs": 0}
        ops_clear, ops_status = ops.decode_range_clear, ops.decode_range_status
        ops.decode_range_clear = lambda: state.__setitem__("word", 0)
        def _status(reset=True):
            w = state["word"]
            if reset:
                state["word"] = 0
            return w
        ops.decode_range_status = _status

        class Fake:
            device = None
            def __init__(self, precision):
                self.decode_precision = precision
                self.model = SimpleNamespace(decoder=SimpleNamespace(mlp_precision="f16x3"))
            _set_decode_precision = gen.Generator3D._set_decode_precision

            @gen._range_guarded(collective=True)
            def sharded(self, data, group=None):
                state["runs"] += 1
                # rank 1's decode trips the half range on its first run in a half precision
                if rank == 1 and self.decode_precision in ("f16x3", "f16f8") and state["runs"] == 1:
                    state["word"] |= ops.RANGE_HALF
                t = torch.tensor([float(gen._PRECISION_ORDER.index(self.decode_precision))])
                dist.all_reduce(t, group=group)                     # the slabs' all-gather stands in: every rank must arrive
                return self.decode_precision, float(t.item())

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # (1) rank 0 was downgraded earlier by a rank-local guard: the ranks agree on bf16x3 BEFORE any decode, no hang
            a = Fake("bf16x3" if rank == 0 else "f16x3")
            r1 = a.sharded(None, None)
            # (2) both in f16x3, rank 1 trips RANGE_HALF: both regenerate in bf16x3
            state["runs"] = 0
            b = Fake("f16x3")
            r2 = b.sharded(None, None)
            runs2 = state["runs"]
            # (3) a stale bit from an earlier launch on this device is not this scene's
            state["runs"] = 5; state["word"] = ops.RANGE_HALF
            c = Fake("f16x3")
            r3 = c.sharded(None, None)
        ops.decode_range_clear, ops.decode_range_status = ops_clear, ops_status
        dist.barrier()
        q.put((rank, r1, a.model.decoder.mlp_precision, r2, runs2, r3))
    finally:
        dist.destroy_process_group()


def test_sharded_guard_agrees_on_one_precision_and_every_rank_enters_every_collective():
    """a rank whose precision was downgraded by a rank-local guard must not skip the sharded entry point's
    reduction (mismatched collective = hang), and the slabs of one value grid must be decoded in ONE arithmetic."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r1, mlp, r2, runs2, r3 in res:
        assert r1 == ("bf16x3", 4.0)                 # both ranks decoded in bf16x3 (index 2 + 2)
        assert mlp == ("f32" if rank == 1 else "f16x3")   # the rank that FOLLOWED also took its attention MLP to the exact kernel
        assert r2 == ("bf16x3", 4.0) and runs2 == 2    # tripped on rank 1 only: BOTH regenerate once
        assert r3 == ("f16x3", 2.0)                  # the stale bit was cleared when the scene began
