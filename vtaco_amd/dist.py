"""Multi-GPU plumbing for the hot path: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  The reference has no distributed code; this is new.

* Inference shards by QUERY POINTS: every point depends only on the (replicated) feature
  grid and weights, so a rank evaluates a contiguous slab of the lattice
  (``slab_of``) with no data-path collective; one all_gather of the logit slabs feeds
  marching cubes (which needs the whole value grid).
* Training is replicas + ONE collective: the gradient all-reduce.  ``GradAllReduce``
  flattens every parameter gradient into a single bucket (20 M f32 = 80 MB for VTacO: one
  large collective suits xGMI's per-link-bound rings better than many small ones) and
  treats parameters that received no gradient (fc_p vs fc_p_img, the frozen t2d net;
  SURVEY.md section 7) as zeros, so ranks never disagree on the bucket layout.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def slab_of(total, rank, world, align=32):
    """Contiguous [first, first+count) share of `total` units for `rank`; all but the last
    slab are multiples of `align` (the decode kernel's 32-point tile)."""
    per = -(-total // world)
    per = -(-per // align) * align
    first = min(rank * per, total)
    return first, max(0, min(per, total - first))


def lattice_align(nx, world):
    """Slab granularity for an nx^3 lattice: whole pairs of x-planes (what the brick-tiled decode kernels want)
    when every rank still gets work that way, else the 32-point tile."""
    pair = 2 * nx * nx
    return pair if pair * world <= nx ** 3 else 32


def all_gather_slabs(local, total, group=None, align=32):
    """Concatenate per-rank 1-D slabs (possibly ragged) into the full [total] tensor on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    counts = [slab_of(total, r, world, align)[1] for r in range(world)]
    width = max(counts)
    pad = torch.zeros(width, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)])


def decode_lattice_sharded(decode_slab, nx, group=None, align=None):
    """`decode_slab(first, count) -> [count]` logits of that lattice slab; returns the whole
    [nx^3] value grid on every rank.  With one rank this is a plain call.  ``align``: slab granularity
    (default: whole x-plane pairs; a decoder that couples the points of a chunk passes its chunk size, so
    that no chunk is split between ranks)."""
    total = nx ** 3
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    align = lattice_align(nx, world) if align is None else align
    first, count = slab_of(total, rank, world, align)
    local = decode_slab(first, count)
    if world == 1:
        return local
    return all_gather_slabs(local, total, group, align)


class GradAllReduce:
    """Average the gradients of `params` over the group with one flat bucket."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self._flat = None

    def __call__(self):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1 or not self.params:
            return
        ref = self.params[0]
        if self._flat is None or self._flat.device != ref.device:
            self._flat = torch.empty(self.numel, dtype=torch.float32, device=ref.device)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self._flat[off:off + n].zero_()
            else:
                self._flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self._flat.div_(dist.get_world_size(self.group))
        off = 0
        for p in self.params:
            n = p.numel()
            g = self._flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
