"""Multi-GPU plumbing for the hot path: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  The reference has no distributed code; this is new.

* Inference shards by QUERY POINTS: every point depends only on the (replicated) feature
  grid and weights, so a rank evaluates a contiguous slab of the lattice
  (``slab_of``) with no data-path collective; one all_gather of the logit slabs feeds
  marching cubes (which needs the whole value grid).
* Training is replicas + ONE kind of collective: the gradient all-reduce.  ``GradAllReduce``
  keeps the gradients in ~20 MB buckets (20 M f32 = 80 MB for VTacO: four buckets; large
  messages suit xGMI's per-link-bound rings), launches each bucket's all-reduce from a
  post-accumulate-grad hook as soon as its last gradient is final, so the collective runs
  under the rest of backward, and treats parameters that received no gradient (fc_p vs
  fc_p_img, the frozen t2d net) as zeros, so ranks never disagree on
  the bucket layout.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def slab_of(total, rank, world, align=32):
    """Contiguous [first, first+count) share of `total` units for `rank`; all but the last non-empty
    slab are multiples of `align` (the decode kernel's 32-point tile).  Rounding the share up to `align` can
    leave the LAST ranks with ``count == 0`` (nx=32 over 7 ranks; a 100 000-point chunk alignment at 128^3 over
    8 ranks): callers must skip their kernels for an empty slab and still join the collective."""
    per = -(-total // world)
    per = -(-per // align) * align
    first = min(rank * per, total)
    return first, max(0, min(per, total - first))


def lattice_align(nx, world):
    """Slab granularity for an nx^3 lattice: whole pairs of x-planes (what the brick-tiled decode kernels want)
    when there are at least as many plane pairs as ranks, else the 32-point tile.  (When the pair count is not a
    multiple of the world size the rounded-up share can still leave trailing ranks empty, see ``slab_of``.)"""
    pair = 2 * nx * nx
    return pair if pair * world <= nx ** 3 else 32


def all_gather_slabs(local, total, group=None, align=32, out=None):
    """Concatenate per-rank 1-D slabs into the full [total] tensor on every rank: ONE collective.  Equal slabs (the
    plane-pair slabs of 128^3 / 256^3 over 2, 4, 8 ranks) are gathered straight into the output (``all_gather_into_tensor``:
    no staging copies); ragged ones go through equal-width padded rows."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    counts = [slab_of(total, r, world, align)[1] for r in range(world)]
    if min(counts) == max(counts) and dist.get_backend(group) == "nccl":
        if out is None:
            out = torch.empty(total, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    width = max(counts)
    pad = torch.zeros(width, dtype=local.dtype, device=local.device)
    pad[:local.numel()] = local
    rows = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(rows, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(rows, counts)])


def decode_lattice_sharded(decode_slab, nx, group=None, align=None, device=None):
    """`decode_slab(first, count) -> [count]` logits of that lattice slab; returns the whole
    [nx^3] value grid on every rank.  With one rank this is a plain call.  ``align``: slab granularity
    (default: whole x-plane pairs; a decoder that couples the points of a chunk passes its chunk size, so
    that no chunk is split between ranks).  A rank whose slab is empty calls no kernel (``decode_slab`` is
    not invoked) and contributes an empty f32 tensor on ``device`` to the all-gather."""
    total = nx ** 3
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    align = lattice_align(nx, world) if align is None else align
    first, count = slab_of(total, rank, world, align)
    if count > 0:
        local = decode_slab(first, count)
    else:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
        local = torch.empty(0, dtype=torch.float32, device=device)
    if world == 1:
        return local
    return all_gather_slabs(local, total, group, align)


class GradAllReduce:
    """Average the gradients of `params` over the group: size-capped buckets whose all-reduces are launched
    from post-accumulate-grad hooks while the rest of backward still runs (RCCL runs the collective on its
    own stream; the bucket's gradients are final when its last hook fires).

    * Buckets are filled in the order the gradients become ready.  The first synchronised step runs with
      buckets in REVERSE registration order (the usual approximation) and records the order in which the
      hooks actually fired; rank 0's order is then broadcast and the buckets are rebuilt once, so every rank
      holds the same layout.  Buckets are launched strictly in index order, so the collectives of different
      ranks always pair up, whatever order the hooks fire in on a given rank.
    * A parameter a step leaves without gradient (fc_p vs fc_p_img, the frozen t2d net, the contact head;
      SURVEY.md section 7) contributes zeros: ``__call__`` (run between backward and the optimizer step)
      zero-fills what never arrived, launches the remaining buckets and waits.
    * No per-parameter copy loop: a gradient reaches its bucket slice with ONE copy inside its hook (none
      when ``.grad`` already is the slice -- after a step every ``.grad`` is a view of its bucket, so with
      ``zero_grad(set_to_none=False)`` the steady state is copy-free); averaging is one multiply per bucket.
    ``overlap=False`` keeps the buckets but launches everything from ``__call__`` (one ``_foreach_copy_``).

    Several ``backward()`` calls per step (gradient accumulation, two losses backpropagated one after the other): wrap all but
    the LAST one in ``with sync.no_sync():`` -- the hooks then leave the gradients to accumulate locally and the last backward
    launches the buckets with the totals.  A hook that finds its gradient already handed to a bucket (a second backward outside
    ``no_sync``: the bucket's all-reduce may be in flight, and with ``.grad`` a view of the bucket autograd has just accumulated
    into the very buffer RCCL is reducing) raises instead of synchronising half a step's gradients.
    """

    def __init__(self, params, group=None, bucket_bytes=20 << 20, overlap=True):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.group, self.bucket_bytes, self.overlap = group, int(bucket_bytes), overlap
        self.numel = sum(p.numel() for p in self.params)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._order = list(range(len(self.params)))[::-1]        # reverse registration order until measured
        self._rebuilt = False
        self._fired = []                                         # hook order of the current backward
        self._buckets = None
        self._hooks = []
        self._enabled = True                                     # False inside no_sync(): hooks and __call__ do nothing
        self.stats = {"buckets": 0, "launched_in_backward": 0, "launched_at_sync": 0}
        if overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- layout -------------------------------------------------------------------------------------
    def _active(self):
        return dist.is_initialized() and dist.get_world_size(self.group) > 1 and bool(self.params)

    def _build(self):
        ref = self.params[0]
        cap = max(1, self.bucket_bytes // 4)
        groups, cur, n = [], [], 0
        for i in self._order:
            sz = self.params[i].numel()
            if cur and n + sz > cap:
                groups.append(cur)
                cur, n = [], 0
            cur.append(i)
            n += sz
        if cur:
            groups.append(cur)
        self._buckets, self._slot = [], {}
        for b, idxs in enumerate(groups):
            flat = torch.zeros(sum(self.params[i].numel() for i in idxs), dtype=torch.float32, device=ref.device)
            views, off = [], 0
            for i in idxs:
                p = self.params[i]
                views.append(flat[off:off + p.numel()].view_as(p))
                self._slot[i] = (b, len(views) - 1)
                off += p.numel()
            self._buckets.append({"idx": idxs, "flat": flat, "views": views, "ready": [False] * len(idxs),
                                  "pending": len(idxs), "work": None})
        self._next = 0
        self.stats["buckets"] = len(self._buckets)

    def _reset(self):
        for bk in self._buckets:
            bk["ready"] = [False] * len(bk["idx"])
            bk["pending"], bk["work"] = len(bk["idx"]), None
        self._next = 0
        self._fired = []

    # -- hooks ----------------------------------------------------------------------------------------
    def _launch_ready(self, in_backward):
        while self._next < len(self._buckets) and self._buckets[self._next]["pending"] == 0:
            bk = self._buckets[self._next]
            bk["work"] = dist.all_reduce(bk["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.stats["launched_in_backward" if in_backward else "launched_at_sync"] += 1
            self._next += 1

    def _on_grad(self, p):
        if not self._enabled or not self._active():
            return
        if self._buckets is None or self._buckets[0]["flat"].device != p.device:
            self._build()
        i = self._index[id(p)]
        b, k = self._slot[i]
        bk = self._buckets[b]
        if bk["ready"][k]:
            raise RuntimeError("GradAllReduce: a parameter received a second gradient before grad_sync() ran -- with several "
                               "backward() calls per step wrap all but the last one in `with grad_sync.no_sync():`")
        view = bk["views"][k]
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
        bk["ready"][k] = True
        bk["pending"] -= 1
        self._fired.append(i)
        self._launch_ready(True)

    def no_sync(self):
        """Context manager for every backward() of a step except the last: gradients accumulate locally, nothing is launched."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            was, self._enabled = self._enabled, False
            try:
                yield self
            finally:
                self._enabled = was
        return ctx()

    # -- between backward and optimizer.step() -------------------------------------------------------------
    def __call__(self):
        if not self._enabled or not self._active():
            return
        if self._buckets is None or self._buckets[0]["flat"].device != self.params[0].device:
            self._build()
        world = dist.get_world_size(self.group)
        # whatever did not arrive through a hook: its gradient if it has one (overlap=False), zeros otherwise
        src, dst = [], []
        for bk in self._buckets:
            for k, i in enumerate(bk["idx"]):
                if bk["ready"][k]:
                    continue
                p, view = self.params[i], bk["views"][k]
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != view.data_ptr():
                    src.append(p.grad)
                    dst.append(view)
                bk["ready"][k] = True
                bk["pending"] -= 1
        if src:
            torch._foreach_copy_(dst, src)
        self._launch_ready(False)
        for bk in self._buckets:
            bk["work"].wait()
            bk["flat"].mul_(1.0 / world)
            for k, i in enumerate(bk["idx"]):
                self.params[i].grad = bk["views"][k]
        fired = self._fired
        self._reset()
        if self.overlap and not self._rebuilt:
            self._rebuild_from(fired)

    def _rebuild_from(self, fired):
        """Once, after the first synchronised step: bucket order := the order the hooks fired in on rank 0
        (parameters that never fired go last), agreed by broadcast so that every rank holds the same layout."""
        seen = set(fired)
        order = list(fired) + [i for i in range(len(self.params))[::-1] if i not in seen]
        dev = self._buckets[0]["flat"].device
        t = torch.tensor(order, dtype=torch.int64, device=dev)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(t, src=src, group=self.group)
        order = [int(v) for v in t.cpu()]
        self._rebuilt = True
        if order != self._order:
            self._order = order
            old = {i: self.params[i].grad for i in range(len(self.params))}
            self._build()
            # the gradients of this step live in the old buckets; move them so that .grad stays a bucket view
            for bk in self._buckets:
                torch._foreach_copy_(bk["views"], [old[i] for i in bk["idx"]])
                for k, i in enumerate(bk["idx"]):
                    self.params[i].grad = bk["views"][k]

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
