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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok1 = True
        for nx in (11, 8, 16):                            # ragged plane-pair slabs, 32-point slabs, even plane pairs
            full = decode_lattice_sharded(lambda first, count: torch.arange(first, first + count, dtype=torch.float32) * 0.5, nx)
            ok1 = ok1 and torch.equal(full, torch.arange(nx ** 3, dtype=torch.float32) * 0.5)
        # gradient all-reduce: rank-dependent grads, one parameter without a gradient on rank 1
        torch.manual_seed(0)
        a, b, c = (torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
                   torch.nn.Parameter(torch.zeros(2, 2)))
        a.grad = torch.full((5, 3), float(rank + 1))
        b.grad = torch.arange(7.0) * (rank + 1)
        if rank == 0:
            c.grad = torch.ones(2, 2)
        GradAllReduce([a, b, c])()
        ok2 = torch.allclose(a.grad, torch.full((5, 3), 1.5)) and torch.allclose(b.grad, torch.arange(7.0) * 1.5) \
            and torch.allclose(c.grad, torch.full((2, 2), 0.5))
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
