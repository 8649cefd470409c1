"""Mesh generation (drop-in for the hot-path part of reference
src/conv_onet/generation.py: ``Generator3D.eval_points`` :338-383 and the dense
evaluation + marching cubes of ``generate_obj_mesh_wnf`` :119-120, 155-157, 257-273).

The reference builds the nx^3 lattice on the CPU, ships it to the GPU in 100k-point
chunks, copies every chunk's logits back and runs scikit-image's marching cubes on
the CPU.  Here the lattice is generated inside the decode kernel, the logits stay on
the device and marching cubes is a HIP kernel; the results (logits within 1e-4,
vertex numbering bit-exact) are the same.  ``generate_hand_mesh`` (:74-115) runs the hand
encoder + MANO layer on the device and the reference's wrist-frame post-processing on the
778 vertices.  ``c_img_all`` can be passed in pre-built or as finger ids (generate_obj_mesh_tactile).
"""
from __future__ import annotations

import operator
import os
from collections import namedtuple

import torch

from .. import ops
from .._lib import VtError

# A/B knob: "0" reads a captured scene's marching-cubes counts through a copy + event behind the graph instead of the page-locked slot
# the scan kernel writes (ops.mc_count_echo)
_MC_ECHO = os.environ.get("VTACO_MC_ECHO", "1") != "0"

Mesh = namedtuple("Mesh", ["vertices", "faces"])
_tensor_version = operator.attrgetter("_version")


_PRECISION_ORDER = ("f16f8", "f16x3", "bf16x3", "f32")            # least to most conservative


def _guard_step(word, precision):
    """What the range word of a finished scene asks of a half-precision decode: (next precision, reason) or None."""
    if precision not in ("f16x3", "f16f8"):
        return None
    if word & ops.RANGE_HALF:
        return "bf16x3", "reach the half-precision range limit (65504)"
    if (word & ops.RANGE_FP8) and precision == "f16f8":
        return "f16x3", "reach 1024, where the fp8 correction products of 'f16f8' begin to clip"
    if (word & ops.RANGE_LOGIT) and precision == "f16f8":
        return "f16x3", "produce logits beyond 2.5, where the relative error of 'f16f8' (~3e-5 |logit|) leaves the 1e-4 bar"
    return None


def _agree_precision(precision, group, device):
    """The most conservative decode precision held by any rank of ``group`` (one MAX all-reduce of its rank in _PRECISION_ORDER);
    ``precision`` itself without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return precision
    mine = _PRECISION_ORDER.index(precision) if precision in _PRECISION_ORDER else len(_PRECISION_ORDER) - 1
    if dist.get_backend(group) == "nccl":
        dev = device if device is not None and torch.device(device).type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    else:
        dev = "cpu"
    t = torch.tensor([mine], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return _PRECISION_ORDER[int(t.item())] if int(t.item()) != mine else precision


def _range_guarded(method=None, *, collective=False):
    """A generation entry point under the half-precision decodes' range guard (ops.decode_range_status, one word per device).
    "f16f8" that met activations >= 1024 (RANGE_FP8: its fp8 correction copies begin to clip) or wrote a logit beyond 2.5
    (RANGE_LOGIT: its error is relative, ~3e-5 |logit|) has left its 1e-4 contract and moves to "f16x3"; a half-precision form that met activations at the edge of the half range (RANGE_HALF: 65504, hi operands saturate)
    moves to "bf16x3" (f32's exponent range) -- for good, with a warning -- and the scene is generated again.  The word is
    cleared (asynchronously) when the outermost guarded call begins, so that only THIS scene's launches are judged, and read
    back once per scene behind the synchronisation the mesh extraction has just done.

    The decision is RANK-LOCAL for every entry point except the sharded one: a rank-0-only mesh export during distributed
    training or per-rank evaluation with uneven scene counts must not meet a collective here.  ``collective=True``
    (generate_obj_mesh_sharded) is a collective on the ``group`` the call was given whatever the ranks' precisions are: the
    ranks first agree on the most conservative precision any of them holds (a rank-local guard may have moved one rank
    earlier: the slabs of one value grid must be decoded in one arithmetic, and every rank must enter the same reductions),
    then every round reduces the word over the group, so that all ranks take the same decision."""
    import functools
    import warnings

    def wrap(method):
        @functools.wraps(method)
        def guarded(self, *args, **kwargs):
            if getattr(self, "_guard_depth", 0):
                return method(self, *args, **kwargs)
            group = kwargs.get("group", args[1] if len(args) > 1 else None) if collective else None
            if collective:
                agreed = _agree_precision(self.decode_precision, group, self.device)
                if agreed != self.decode_precision:
                    warnings.warn(f"Generator3D: another rank of the group decodes in {agreed!r}; this rank follows "
                                  f"(decode_precision {self.decode_precision!r} -> {agreed!r})")
                    self._set_decode_precision(agreed)
            if self.decode_precision not in ("f16x3", "f16f8"):
                return method(self, *args, **kwargs)
            self._guard_depth = 1
            try:
                ops.decode_range_clear()                             # bits left by earlier launches on this device are not this scene's
                out = method(self, *args, **kwargs)
                for _ in range(2):                                   # f16f8 -> f16x3 -> bf16x3 at most
                    if self.decode_precision not in ("f16x3", "f16f8"):
                        break
                    word = ops.decode_range_status(reset=True)
                    if collective:
                        word = _reduce_or(word, group, self.device)
                    step = _guard_step(word, self.decode_precision)
                    if step is None:
                        break
                    nxt, why = step
                    warnings.warn(f"Generator3D: the decoder's activations {why}; "
                                  f"decode_precision {self.decode_precision!r} -> {nxt!r} and the scene is generated again")
                    self._set_decode_precision(nxt)
                    out = method(self, *args, **kwargs)
                return out
            finally:
                self._guard_depth = 0
        return guarded
    return wrap if method is None else wrap(method)


def _reduce_or(word, group, device):
    """Bitwise OR of a small status word over the ranks of ``group`` (None = the default group); the word itself without an
    initialised process group.  With the nccl (RCCL) backend the scratch tensor lives on this rank's device."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return word
    if dist.get_backend(group) == "nccl":
        dev = device if device is not None and torch.device(device).type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    else:
        dev = "cpu"
    t = torch.tensor([(word >> b) & 1 for b in range(8)], dtype=torch.int32, device=dev)      # RCCL has no bitwise OR: MAX per bit
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return sum(int(v) << b for b, v in enumerate(t.tolist()))


class Generator3D(object):
    """Constructor arguments as the reference (generation.py:42-52)."""

    MAX_SCENE_GRAPHS = 4
    FUSED_CHUNKS_PER_CALL = 256          # attention_local: chunks of points_batch_size points evaluated per launch sequence (see _fused_chunks_per_call)
    # generate_obj_mesh_wnf replays the visual branch as a captured hipGraph (VTACO_SCENE_GRAPH=0: eager launches)
    scene_graph = os.environ.get("VTACO_SCENE_GRAPH", "1") != "0"

    def __init__(self, model, points_batch_size=100000, threshold=0.5, refinement_step=0, device=None,
                 resolution0=16, upsampling_steps=3, with_normals=False, padding=0.1, sample=False,
                 input_type=None, vol_info=None, vol_bound=None, simplify_nfaces=None, alpha=0.2,
                 with_img=False, encode_t2d=False, decode_precision="f16x3", depth_origin=None):
        self.model = model.to(device)
        # arithmetic of the dense lattice decode (eval_lattice): "f16x3" = split-f16 MFMA (the default: f32-level logit error,
        # ~1e-6 on the goldens, for hidden activations below 65504 -- guarded, see _range_guarded), "bf16x3" = split-bf16 MFMA
        # (f32's exponent range, ~1.6e-5), "f32" = exact-f32 MFMA, "f16f8" = f16 products + fp8 correction products (opt-in:
        # fewer matrix cycles, but its error is RELATIVE, ~3e-5 |logit| -- inside the 1e-4 bar for |logit| <~ 2.5 only -- and
        # the mesh is not vertex-for-vertex the f32 path's; slabs it does not cover run as "f16x3").  eval_points follows the
        # decoder's own ``precision`` attribute (default "f32").
        self.decode_precision = decode_precision
        self.points_batch_size = points_batch_size
        self.threshold, self.refinement_step = threshold, refinement_step
        self.device = device
        self.resolution0, self.upsampling_steps = resolution0, upsampling_steps
        self.with_normals, self.input_type, self.padding, self.sample = with_normals, input_type, padding, sample
        self.simplify_nfaces, self.alpha = simplify_nfaces, alpha
        self.with_img, self.encode_t2d = with_img, encode_t2d
        # the tactile sensor's flat depth reading [240*320] (the VTacO branch compares every depth image with it): an array, a
        # path, or None = the reference's ./data/VTacO_mesh/depth_origin.txt (generation.py:17), read when first needed
        self.depth_origin = depth_origin
        self.vol_bound = vol_bound
        if input_type == 'pointcloud_crop':
            raise VtError("Generator3D: crop / sliding-window mode is not built (no shipped config uses it)")

    # -- reference API: arbitrary points, chunked, result on the CPU ---------------
    def eval_points(self, p, c=None, c_img_all=None, vol_bound=None, **kwargs):
        """Occupancy logits [N] (CPU tensor, as the reference returns) for points p [N,3]."""
        p = p.to(self.device)
        ci = None if c_img_all is None else c_img_all.reshape(-1, c_img_all.shape[-1]).to(self.device)
        outs = []
        with torch.no_grad():
            lo0 = 0
            if self.with_img and ci is not None and hasattr(self.model.decoder, 'fuser'):
                # decoder attention_local: the whole chunks as batches of chunks (see _eval_lattice_fused), the ragged rest below
                dec, chunk = self.model.decoder, self.points_batch_size
                grid = dec._grid_of(c)
                full = p.shape[0] // chunk
                per_call = self._fused_chunks_per_call(chunk)
                for lo in range(0, full, per_call):
                    nb = min(per_call, full - lo)
                    sl = slice(lo * chunk, (lo + nb) * chunk)
                    pb = p[sl].float()
                    feat = ops.sample_grid(grid, pb.unsqueeze(0), dec.padding).reshape(nb, chunk, -1)
                    fused = dec.fuser(ci[sl].float().reshape(nb, chunk, -1), 1, feat, 1)
                    outs.append(dec._mlp_fwd(fused, pb.reshape(nb, chunk, 3)).reshape(-1))
                lo0 = full * chunk
            for lo in range(lo0, p.shape[0], self.points_batch_size):
                pi = p[lo:lo + self.points_batch_size].unsqueeze(0)
                if self.with_img and ci is not None:
                    occ = self.model.decode_img(pi, c, ci[lo:lo + self.points_batch_size].unsqueeze(0), **kwargs).logits
                else:
                    occ = self.model.decode(pi, c, **kwargs).logits
                outs.append(occ.squeeze(0))
        return torch.cat(outs, dim=0).detach().cpu()

    def _fused_chunks_per_call(self, chunk):
        """Chunks of ``chunk`` points per launch sequence of the attention decoder: FUSED_CHUNKS_PER_CALL, but at most 2^19 points
        (the fusion workspace is ~1.3 KB per point: 0.7 GB there, whatever the chunk size)."""
        return max(1, min(self.FUSED_CHUNKS_PER_CALL, (1 << 19) // max(int(chunk), 1)))

    # -- fast path: the nx^3 lattice never exists as a tensor ----------------------
    def eval_lattice(self, c, nx, c_img_all=None, first=0, count=None, out=None):
        """Logits of the (1+padding)-box lattice, device tensor [count] (whole: nx^3)."""
        grid = c['grid'] if isinstance(c, dict) else c
        if grid.shape[0] != 1:
            raise VtError("eval_lattice: one scene at a time (the lattice is per scene)")
        if count == 0:                                   # an empty slab of a sharded lattice: no kernel
            return torch.empty(0, dtype=torch.float32, device=grid.device)
        return self.model.decoder.decode_lattice(grid, nx, box=1 + self.padding, first=first, count=count,
                                                 c_img=c_img_all, out=out, precision=self.decode_precision).reshape(-1)

    def extract_mesh(self, value_grid, level=None):
        """``measure.marching_cubes(value_grid, gradient_direction='ascent')`` followed by
        ``vertices -= nx/2; vertices *= (1+padding)/nx`` (generation.py:270-272), on the device."""
        nx = value_grid.shape[0]
        verts, faces, _ = ops.marching_cubes(value_grid, level, rescale=(nx / 2, (1 + self.padding) / nx))
        return Mesh(verts, faces)

    # -- whole scene as ONE hipGraph replay (encode + dense decode + marching-cubes count) --------
    def _module_tables(self):
        """The model's modules and their parameter / buffer dicts, listed once per model object: the per-scene checks below
        walk these lists instead of ``nn.Module``'s recursive generators (0.3 ms each for the ~110 tensors of the shipped model,
        more than a quarter of a graphed scene).  A module's dict is updated in place when a weight is replaced, so the lists
        stay valid; submodules (or a module's first parameter) added after the first call are not followed."""
        tab = getattr(self, "_tables", None)
        if tab is None or tab[0] is not self.model:
            mods = list(self.model.modules())
            tab = self._tables = (self.model, mods, [d for m in mods for d in (m._parameters, m._buffers) if d])
        return tab

    def _set_decode_precision(self, precision):
        """Move the lattice decode to ``precision`` (the range guard's downgrade); "bf16x3" / "f32" also take the attention decoder's
        MLP back to the exact kernel."""
        self.decode_precision = precision
        if precision in ("bf16x3", "f32") and hasattr(self.model.decoder, "mlp_precision"):
            self.model.decoder.mlp_precision = "f32"

    def _eval_mode(self):
        """``self.model.eval()`` (generation.py:66, 131), skipped when every module already is in eval mode."""
        if any(m.training for m in self._module_tables()[1]):
            self.model.eval()

    def _weight_stamps(self):
        """(object, storage address, version counter) of every parameter and buffer: changes when a weight is updated in place
        (optimizer.step, load_state_dict), moved (``.to()`` swaps the storage under the same Parameter) or replaced."""
        ts = [t for d in self._module_tables()[2] for t in d.values() if t is not None]
        return tuple(map(id, ts)), tuple(map(torch.Tensor.data_ptr, ts)), tuple(map(_tensor_version, ts))

    def _captured(self, key, shapes, run):
        """The captured graph of ``run(*static_inputs)`` for one key (kind, input shapes, ...): ``{"graph", "in": static inputs, "out":
        what run returned, ...}``.  A graph holds raw pointers to derived buffers -- packed conv weights, the decoder blob, the UNet3D
        workspace -- that live in caches keyed on the weights' versions and on the last shape run; so the entry (a) keeps every such
        tensor alive next to the graph (``ops.graph_keepalive``) and (b) records the weight stamps at capture: a replay after
        ``optimizer.step()`` / ``load_state_dict`` / ``.to()`` finds different stamps and captures afresh instead of reading stale
        or recycled memory.  At most MAX_SCENE_GRAPHS entries stay captured, least recently used first out."""
        self._graphs = getattr(self, "_graphs", {})
        hit = self._graphs.get(key)
        stamps = self._weight_stamps()
        if hit is not None and hit["stamps"] == stamps:
            if len(self._graphs) > 1:
                self._graphs[key] = self._graphs.pop(key)   # most recently used last
            return hit
        self._drop_graph(key)                            # stale: drop the old graph (and its keep-alive list) first
        while len(self._graphs) >= self.MAX_SCENE_GRAPHS:   # every graph pins its workspaces (~0.5 GB at 128^3): keep a few shapes
            self._drop_graph(next(iter(self._graphs)))
        static = [torch.zeros(shape, dtype=torch.float32, device=self.device) for shape in shapes]
        with ops.graph_keepalive() as keep:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):                      # warm-up: fills every cache / workspace outside the capture
                    run(*static)
            torch.cuda.current_stream().wait_stream(side)
            # the stamps the entry is valid for are taken AFTER the warm-up (a module that bumps a buffer's version during its
            # forward would otherwise look changed at every call and be captured again each time) ...
            stamps = self._weight_stamps()
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph):
                out = run(*static)
            # ... and must not move while the graph is recorded: a capture that changes a weight cannot be replayed safely
            if self._weight_stamps() != stamps:
                raise VtError("Generator3D: the model's parameters or buffers changed while its launches were being captured; "
                              "set scene_graph = False for models whose forward updates state")
        self._graphs[key] = {"graph": graph, "in": static, "out": out, "keep": list(keep), "stamps": stamps}
        return self._graphs[key]

    def _drop_graph(self, key):
        """Forget a captured graph; its marching-cubes echo slot (a page-locked block, 64 per process) goes back to the pool once the
        device has drained the graph's last replay."""
        hit = self._graphs.pop(key, None)
        if hit is not None and hit.get("echo") is not None:
            torch.cuda.synchronize()
            hit.pop("graph", None)
            ops.mc_echo_release(hit["echo"])

    def __del__(self):
        try:
            for key in list(getattr(self, "_graphs", {})):
                self._drop_graph(key)
        except Exception:                                 # noqa: BLE001 -- interpreter shutdown: the library may be gone
            pass

    def _scene_graph(self, shape, nx):
        """Encode + dense decode + marching-cubes classification of one (input shape, lattice size) as one graph."""
        key = (tuple(shape), nx, self.decode_precision)
        hit = getattr(self, "_graphs", {}).get(key)
        # the counts of a replay arrive in a page-locked slot the scan kernel writes (no copy command between it and the emit kernels);
        # the slot is made before the capture and belongs to this graph
        if hit is not None and "echo" in hit:
            echo = hit.pop("echo")                        # (a stale entry is re-captured below: its slot moves to the new graph)
        else:
            try:
                echo = ops.mc_echo_slot() if _MC_ECHO else None
            except VtError:                                           # every slot taken (64 captured shapes): the copy + event form
                echo = None

        def run(static_in):
            c = self.model.encode_inputs(static_in)
            vol = self.eval_lattice(c, nx).reshape(nx, nx, nx)
            return vol, (ops.mc_count_echo(vol, None, echo) if echo is not None else ops.mc_count(vol))
        g = self._captured(key, [shape], run)
        g["vol"], g["ws"] = g["out"]
        g["echo"] = echo
        return g

    def _graphs_allowed(self):
        """Captured graphs are used in single-process runs only: with several ranks per node the launches stay eager -- two processes
        replaying graphs on ONE device (the gloo dry-run of the multi-rank bench) took 180 ms per replay, and the one-process-per-GPU
        case could not be measured on this pool (the encoder is then 1.0 instead of 0.7 ms per scene)."""
        if not self.scene_graph:
            return False
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _worth_capturing(self, key):
        """A capture costs two warm-up runs and a recording (tens of milliseconds): a shape is captured when it comes back, not the
        first time it is seen -- callers whose clouds all differ in size stay on plain launches instead of recording a graph per call."""
        if key in getattr(self, "_graphs", {}):
            return True
        seen = self.__dict__.setdefault("_seen_shapes", {})
        if key in seen:
            return True
        if len(seen) >= 64:
            seen.pop(next(iter(seen)))
        seen[key] = True
        return False

    def _replay(self, kind, tensors, run):
        """``run(*tensors)`` through a captured graph per (kind, shapes) when ``self.scene_graph`` is on.  LIFETIME of the result:
        it lives in the graph's static buffers -- the next replay of the same (kind, shapes) overwrites it in place, and the entry
        can be evicted (MAX_SCENE_GRAPHS, least recently used) -- so it is for use within the current scene; a caller that keeps
        encoder outputs across scenes (``c`` of one cloud next to the ``c`` of another of the same shape) clones them.  The
        generator's own entry points consume the result before they return.  Plain call otherwise."""
        key = (kind,) + tuple(tuple(t.shape) for t in tensors)
        if not self._graphs_allowed() or not self._worth_capturing(key):
            with torch.no_grad():
                return run(*[t.to(self.device) for t in tensors])
        g = self._captured(key, [t.shape for t in tensors], run)
        for dst, src in zip(g["in"], tensors):
            dst.copy_(src.to(self.device), non_blocking=True)
        g["graph"].replay()
        return g["out"]

    @_range_guarded
    def generate_mesh_graphed(self, inputs):
        """Same result as ``generate_obj_mesh_wnf({'inputs': inputs})`` for the visual branch, with the
        ~110 launches of encode + decode + marching-cubes classification replayed as one hipGraph
        (launch-bound otherwise); only the data-dependent output sizing leaves the graph.  Safe across weight
        updates and interleaved eager calls of other shapes (see ``_scene_graph``)."""
        self._eval_mode()
        nx = self.resolution0 * 4
        g = self._scene_graph(inputs.shape, nx)
        g["in"][0].copy_(inputs.to(self.device), non_blocking=True)
        if g.get("echo") is not None:
            ops.mc_echo_arm(g["echo"])                               # the number this replay's scan kernel writes behind the counts
        g["graph"].replay()
        verts, faces, _ = ops.mc_emit(g["vol"], g["ws"], rescale=(nx / 2, (1 + self.padding) / nx), echo=g.get("echo"))
        return Mesh(verts, faces)

    @_range_guarded(collective=True)
    def generate_obj_mesh_sharded(self, data, group=None):
        """``generate_obj_mesh_wnf`` with the lattice split over the ranks of a process group (one process per GPU):
        every rank encodes the scene (cheap, deterministic: no broadcast), decodes its slab of x-plane pairs with no
        data-path collective, one all_gather of the logit slabs (8.4 MB at 128^3, 67 MB at 256^3) rebuilds the value
        grid, and every rank extracts the (identical) mesh.  Without an initialised group this is the single-GPU path."""
        from .. import dist as vdist
        self._eval_mode()
        nx = self.resolution0 * 4
        inputs = data.get('inputs').to(self.device)
        c = self._replay("encode_inputs", [inputs], self.model.encode_inputs)        # the encoder's launches as one graph
        with torch.no_grad():
            if self.with_img:
                # the tactile branches (VTacOH fingertips / VTacO contact clouds): every rank assigns finger ids to ITS slab and
                # decodes by id; the VTacO clouds are drawn with numpy's generator, so rank 0's go to everybody (a few KB) -- and so
                # do rank 0's tactile FEATURES (F x C floats): they come out of the host framework's convolutions (MIOpen), whose
                # algorithm choice is made per process by timing, so two ranks may hold features that differ in the last bits and
                # the slabs of one value grid would not belong to one function
                setup = self._tactile_setup(data)
                if vdist.dist.is_initialized() and vdist.dist.get_world_size(group) > 1:
                    setup['feats'] = setup['feats'].float().contiguous()
                    for key in ('anchors', 'count', 'success', 'feats'):
                        t = setup[key].to(self.device)
                        vdist.dist.broadcast(t, src=vdist.dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                        setup[key] = t
                fused = hasattr(self.model.decoder, 'fuser')
                values = vdist.decode_lattice_sharded(
                    lambda first, count: self._eval_lattice_tactile(c, nx, setup, first, count), nx, group,
                    align=self.points_batch_size if fused else None, device=self.device)
            else:
                values = vdist.decode_lattice_sharded(lambda first, count: self.eval_lattice(c, nx, first=first, count=count), nx, group,
                                                      device=self.device)
        return self.extract_mesh(values.reshape(nx, nx, nx))

    @_range_guarded
    def generate_obj_mesh_tactile(self, data, finger_feats, anchors, success, mode='within', radius=None, count=None):
        """Tactile branch of ``generate_obj_mesh_wnf`` (generation.py:159-257) without the dense
        ``c_img_all [1,nx^3,C]`` tensor and its CPU cdist glue: every lattice point gets the id of the finger whose
        contact points lie within ``radius`` (``mode='within'``, radius 0.015: VTacO, :245-255) or of the nearest
        successful fingertip (``mode='nearest'``, radius 0.05: VTacOH, :186-200) from ``vt_tactile_assign``, and the
        decoder reads ``finger_feats [F,C]`` by id (``vt_decode_fwd_ids``): 1 byte per point instead of 4*C.
        ``anchors [F,K,3]`` (K = 1 for 'nearest'), ``count [F]`` valid anchors per finger, ``success [F]``."""
        self._eval_mode()
        nx = self.resolution0 * 4
        setup = {'feats': finger_feats, 'anchors': anchors, 'success': success, 'mode': mode,
                 'radius': (0.015 if mode == 'within' else 0.05) if radius is None else radius,
                 'count': count if count is not None else torch.full((anchors.shape[0],), anchors.shape[1], dtype=torch.int32)}
        inputs = data.get('inputs').to(self.device)
        c = self._replay("encode_inputs", [inputs], self.model.encode_inputs)        # the encoder's ~55 launches as one graph
        with torch.no_grad():
            values = self._eval_lattice_tactile(c, nx, setup)
        return self.extract_mesh(values.reshape(nx, nx, nx))

    def _eval_lattice_tactile(self, c, nx, setup, first=0, count=None):
        """Logits of lattice points [first, first+count) with the tactile features of ``setup`` (finger features, anchors, rule)."""
        count = nx ** 3 - first if count is None else count
        grid = c['grid'] if isinstance(c, dict) else c
        if count == 0:                                   # an empty slab of a sharded lattice: no kernel
            return torch.empty(0, dtype=torch.float32, device=grid.device)
        ids = ops.tactile_assign(setup['anchors'].to(self.device), setup['success'].to(self.device), setup['mode'], setup['radius'],
                                 lattice=(nx, 1 + self.padding, first, count), count=setup['count'].to(self.device))
        feats = setup['feats'].to(self.device)
        if hasattr(self.model.decoder, 'fuser'):
            return self._eval_lattice_fused(c, nx, ids, feats, first, count)
        return self.model.decoder.decode_lattice_ids(grid, nx, ids, feats, box=1 + self.padding, first=first, count=count,
                                                     precision=self.decode_precision).reshape(-1)

    def _eval_lattice_fused(self, c, nx, ids, finger_feats, first=0, count=None):
        """``decoder: attention_local`` over the lattice (BASELINE config 3's decoder): TransformerFusion couples the points
        of a chunk (attention + InstanceNorm over the chunk), so the chunk is part of the function -- ``points_batch_size``
        points at a time in lattice order, exactly as ``eval_points`` walks them (the reference default of 100 000 needs a
        40 GB attention matrix; 2048 is the workable setting).  The per-chunk tactile features are gathered from the finger ids.
        A slab must start on a chunk boundary (sharded generation aligns its slabs to the chunk)."""
        from ..common import make_3d_grid
        chunk = self.points_batch_size
        count = nx ** 3 - first if count is None else count
        if first % chunk:
            raise VtError(f"_eval_lattice_fused: slab start {first} splits a chunk of {chunk} points")
        # the lattice as the reference builds it (host arithmetic, generation.py:155-157), kept on the device per (nx, padding):
        # 25 MB at 128^3 that every scene of a run shares
        cache = self.__dict__.setdefault("_lattice_points", {})
        key = (nx, float(self.padding))
        if key not in cache:
            if len(cache) >= 2:
                cache.pop(next(iter(cache)))
            cache[key] = ((1 + self.padding) * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).to(self.device)
        pts = cache[key][first:first + count]
        feats = finger_feats.float().contiguous()
        out = torch.empty(count, dtype=torch.float32, device=self.device)
        dec = self.model.decoder
        grid = dec._grid_of(c)
        # whole chunks go through the kernels as a BATCH of chunks (each is its own attention / InstanceNorm problem, exactly as
        # in a call of its own: the same logits bit for bit) -- one chunk per call is ~20 launches per 2048 points, 100 ms of
        # launches for a 128^3 lattice whose arithmetic takes 15
        full = count // chunk
        per_call = self._fused_chunks_per_call(chunk)
        C = feats.shape[1]
        # the finger ids of the whole chunks, [full, chunk] (a view): the fusion kernels read the tactile rows by id
        # (vt_fusion_fwd_ids) -- no gathered [chunks, chunk, C] tensor, no framework indexing between vt_tactile_assign and the fuser
        P3, I2, O2 = pts[:full * chunk].reshape(full, chunk, 3), ids.reshape(-1)[:full * chunk].view(full, chunk), out[:full * chunk].view(full, chunk)

        def fused_chunks(sel):
            """The attention decoder on the chunks ``sel`` (a slice of consecutive chunks, or an index tensor)."""
            p = P3[sel]
            nb = p.shape[0]
            if isinstance(sel, slice):                      # consecutive chunks: the lattice range itself (points generated in the kernel)
                feat = ops.sample_grid(grid, None, dec.padding, lattice=(nx, 1 + self.padding, first + sel.start * chunk, nb * chunk))
                fused = dec.fuser.forward_ids(I2[sel], feats, feat.reshape(nb, chunk, -1))
            else:                                           # picked chunks: their id rows through the index list, in the kernel
                feat = ops.sample_grid(grid, p.reshape(1, -1, 3), dec.padding).reshape(nb, chunk, -1)
                fused = dec.fuser.forward_ids(I2, feats, feat, chunk_index=sel.to(torch.int32))
            O2[sel] = dec._mlp_fwd(fused, p)

        # (the shortcut below holds in eval mode only: under model.train() the reference's dropout breaks the 'constant row ->
        # InstanceNorm -> 0' argument; generation always runs in eval mode (_eval_mode), asserted here.  It costs one host read per
        # lattice (torch.nonzero), so a lattice that takes it cannot be captured into a hipGraph.)
        if full and self.skip_untouched_chunks and not dec.training:
            # A chunk NO point of which carries a tactile feature (the fingers touch a few per cent of a scene's chunks) needs no
            # attention at all: with c_img = 0 on the whole chunk the decoder's self-attention block returns the same vector for
            # every point, InstanceNorm over the chunk turns that into zeros, the cross-attention of an all-zero query block is
            # constant over the points again, and the last InstanceNorm leaves fuse(0, c) = 0 -- in exact arithmetic and, bit for
            # bit, in these kernels (constant rows leave the norm as exact zeros: fusion.hip, fusion_inorm_relu*; asserted in
            # tests/test_fusion_gpu.py; the reference's own f32 evaluation leaves rounding noise of ~1e-5 there, TransformerFusion.py
            # :144, 209-218).  Such chunks go through the conditioned MLP with zero features; the others through the whole fuser.
            touched = (I2 != 255).any(dim=1)
            t_idx, u_idx = torch.nonzero(touched).flatten(), torch.nonzero(~touched).flatten()      # one host read per lattice
            for lo in range(0, u_idx.numel(), per_call):
                sel = u_idx[lo:lo + per_call]
                O2[sel] = dec._mlp_fwd(self._zero_features(sel.numel() * chunk, C).view(-1, chunk, C), P3[sel])
            for lo in range(0, t_idx.numel(), per_call):
                fused_chunks(t_idx[lo:lo + per_call])
        else:
            for lo in range(0, full, per_call):
                fused_chunks(slice(lo, min(lo + per_call, full)))
        if full * chunk < count:                          # the ragged last chunk
            sl = slice(full * chunk, count)
            tail = ids.reshape(-1)[sl].view(1, -1)
            feat = ops.sample_grid(grid, pts[sl].unsqueeze(0), dec.padding)
            out[sl] = dec._mlp_fwd(dec.fuser.forward_ids(tail, feats, feat), pts[sl].unsqueeze(0))[0]
        return out

    # attention_local over a lattice: chunks without any tactile feature skip the fuser (see _eval_lattice_fused; VTACO_FUSION_SKIP_UNTOUCHED=0
    # or ``gen.skip_untouched_chunks = False`` sends every chunk through the three attention units)
    skip_untouched_chunks = os.environ.get("VTACO_FUSION_SKIP_UNTOUCHED", "1") != "0"

    def _zero_features(self, rows, C):
        """[rows, C] zeros on the device, never written: the fused features of chunks without tactile input."""
        z = self.__dict__.get("_zero_feat")
        if z is None or z.shape[0] < rows or z.shape[1] != C or z.device != torch.device(self.device):
            z = self._zero_feat = torch.zeros((rows, C), dtype=torch.float32, device=self.device)
        return z[:rows]

    def generate_hand_mesh(self, data):
        """Hand mesh of one scene (generation.py:74-115): encoder_hand -> MANO vertices, then out of the MANO
        frame (fixed offset and rotation), out of the predicted wrist rotation (rotation vector -> 'XYZ' Euler
        -> the reference's R_from_PYR convention, common.py:591-604), plus the wrist position, normalised by
        the object's ``inputs.pc_ply`` cloud (norm_pc_1, common.py:606-612).  Returns Mesh(vertices [778,3]
        f64, faces [1538,3] i64) on the device (the reference wraps the same arrays in a trimesh.Trimesh)."""
        import numpy as np
        from scipy.spatial.transform import Rotation
        self._eval_mode()
        inputs = data.get('inputs').to(self.device)
        pc_ply = data.get('inputs.pc_ply').to(self.device)
        if inputs.shape[0] != 1:
            raise VtError(f"generate_hand_mesh: one scene at a time (got a batch of {inputs.shape[0]})")
        c_hand = self._replay("encode_hand_inputs", [inputs], self.model.encode_hand_inputs)   # ~70 launches as one graph
        if 'mano_verts' not in c_hand:
            raise VtError("generate_hand_mesh: the hand encoder has no MANO layer (out_dim <= 30 regresses digit poses only)")
        param = c_hand['mano_param'][0].double().cpu().numpy()

        def pyr(roll, pitch, yaw):                                  # 3x3, host side: nine numbers
            cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
            about_z = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
            about_x_t = np.array([[1, 0, 0], [0, cp, sp], [0, -sp, cp]])
            about_y_t = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
            return about_x_t @ about_y_t @ about_z

        euler = Rotation.from_rotvec(param[3:6]).as_euler('XYZ', degrees=False)
        undo = np.linalg.inv(pyr(*euler)) @ np.linalg.inv(pyr(-np.pi / 2, np.pi / 2, 0.0))
        dev = c_hand['mano_verts'].device
        undo_t = torch.from_numpy(undo).to(dev)
        v = (c_hand['mano_verts'][0] - torch.tensor([0.11, 0.005, 0.0], device=dev)).double()
        v = v @ undo_t.t() + torch.from_numpy(param[:3]).to(dev)
        cloud = pc_ply[0].float()
        centroid = cloud.mean(dim=0)
        m = (cloud - centroid).pow(2).sum(dim=1).sqrt().max()
        v = (v - centroid.double()) / (2.0 * m.double())
        return Mesh(v, c_hand['mano_faces'])

    @_range_guarded
    def generate_obj_mesh_wnf(self, data, c_img_all=None):
        """Encode -> dense decode -> marching cubes for one scene; ``data['inputs']`` is the
        point cloud [1,T,3].  Returns Mesh(vertices [V,3] f32, faces [F,3] i32) on the device."""
        self._eval_mode()
        nx = self.resolution0 * 4                       # generation.py:120
        inputs = data.get('inputs').to(self.device)
        if self.with_img and c_img_all is None:
            return self._generate_tactile(data)
        if (not self.with_img and self._graphs_allowed() and inputs.dim() == 3 and inputs.shape[0] == 1
                and self._worth_capturing((tuple(inputs.shape), nx, self.decode_precision))):
            # the visual branch: the same launches replayed as one hipGraph per (cloud shape, lattice) -- 1.1 instead of 1.5 ms
            return self.generate_mesh_graphed(inputs)
        with torch.no_grad():
            c = self.model.encode_inputs(inputs)
            values = self.eval_lattice(c, nx, c_img_all=c_img_all if self.with_img else None)
        return self.extract_mesh(values.reshape(nx, nx, nx))

    def _depth_origin(self):
        import numpy as np
        src = self.depth_origin
        if src is None:
            src = "./data/VTacO_mesh/depth_origin.txt"
        if isinstance(src, str):
            import os
            if not os.path.exists(src):
                raise VtError(f"Generator3D: the VTacO branch needs the sensor's flat depth reading; {src} not found "
                              "(pass depth_origin=<array or path>)")
            src = np.loadtxt(src)
            self.depth_origin = src
        return np.asarray(src.cpu() if torch.is_tensor(src) else src, dtype=np.float64).reshape(-1)

    def _tactile_setup(self, data, sides=None):
        """Finger features, anchors and assignment rule of the configured tactile branch (VTacO t2d or VTacOH).  ``sides``: two HIP
        streams that take the branch's encoders (see _generate_tactile); None: everything on the current stream."""
        return self._setup_vtaco_t2d(data, sides) if self.encode_t2d else self._setup_vtacoh(data, sides)

    def _side_streams(self):
        """Two side streams per generator for the tactile branch's encoders, or None where graphs / overlap are off
        (VTACO_SCENE_OVERLAP=0: the three encoders one after the other on the caller's stream)."""
        if os.environ.get("VTACO_SCENE_OVERLAP", "1") == "0" or not self._graphs_allowed():
            return None
        if getattr(self, "_sides", None) is None:
            # (plain priority: with high-priority side streams the VTacOH route measured 3.59 against 3.71 ms, the t2d route 4.7 against
            # 2.25 -- the shape encoder's convs then wait behind ~60 small launches)
            prio = int(os.environ.get("VTACO_SCENE_SIDE_PRIORITY", "0"))
            self._sides = (torch.cuda.Stream(device=self.device, priority=prio), torch.cuda.Stream(device=self.device, priority=prio))
        return self._sides

    def _generate_tactile(self, data):
        """generate_obj_mesh_tactile with the branch's own setup.  The scene's encoders do not depend on each other -- the shape
        encoder (0.6 ms, the whole chip for its 64^3 convs), the tactile feature encoder (Resnet18 on five images: 1.2-1.3 ms of
        launch-bound kernels on a few CUs) and, in the VTacOH branch, the hand encoder (0.44 ms) -- so their graphs are replayed on
        three HIP streams at once and joined in front of the decode: 2.3 ms of encoders one after the other become the longest one.
        The setup's host side (contact clouds in numpy, the fingertips' frame change) runs under them."""
        self._eval_mode()
        nx = self.resolution0 * 4
        inputs = data.get('inputs').to(self.device)
        if inputs.shape[0] != 1:
            raise VtError(f"generate_obj_mesh_wnf: one scene at a time (got a batch of {inputs.shape[0]})")
        sides = self._side_streams()
        if sides is not None:
            cur = torch.cuda.current_stream(self.device)
            for side in sides:                                      # (the side streams see everything queued so far: the inputs' uploads,
                side.wait_stream(cur)                               #  the previous scene's reads of the graphs' static outputs)
            pending = self._tactile_setup(data, sides)              # queues the tactile encoders; returns the host part still to do
            c = self._replay("encode_inputs", [inputs], self.model.encode_inputs)
            setup = pending()
            for side in sides:
                cur.wait_stream(side)
        else:
            c = self._replay("encode_inputs", [inputs], self.model.encode_inputs)
            setup = self._tactile_setup(data)
        with torch.no_grad():
            values = self._eval_lattice_tactile(c, nx, setup)
        return self.extract_mesh(values.reshape(nx, nx, nx))

    def _setup_vtaco_t2d(self, data, sides=None):
        """The VTacO branch of generate_obj_mesh_wnf (generation.py:202-257): per finger whose touch succeeded, the contact cloud
        unprojected from the sample's depth image (as the reference: the dataset's depth, not the predicted one, and the dataset's
        camera poses), at most 128 points; every lattice point within 0.015 of a contact point takes that finger's tactile feature
        (later fingers overwrite earlier ones) -- by finger id (vt_tactile_assign 'within' + vt_decode_fwd_ids) at any lattice
        size, where the reference builds a dense [1, 128^3, C] tensor with eight CPU cdist passes hard-wired to 128^3."""
        from ..common import contact_clouds_from_depth
        if getattr(self.model, 'encoder_img', None) is None:
            raise VtError("generate_obj_mesh_wnf(with_img, encode_t2d): the model needs encoder_img (tactile features)")
        inputs = data.get('inputs').to(self.device)
        if inputs.shape[0] != 1:
            raise VtError(f"generate_obj_mesh_wnf: one scene at a time (got a batch of {inputs.shape[0]})")
        self._eval_mode()
        # [1,5,C]; the feature encoder (Resnet18 in eval mode: ~60 launch-bound MIOpen / ATen kernels) replayed as a graph
        c_img = self._replay_on(sides[0] if sides else None, "encode_img", [data.get('inputs.img')], self.model.encode_img_inputs)
        # (one scene's five images stay on the host: 0.5 ms of numpy that runs under the shape encoder's replay; the device kernels of
        # the training step -- vt_contact_scan / vt_contact_points, 40 images per step -- need the counts back on the host in between,
        # which would wait for that replay: measured 1.78 against 1.26 ms for this setup)
        def host_part():
            anchors, count = contact_clouds_from_depth(
                data.get('inputs.depth')[0].float().cpu().numpy(), self._depth_origin(),
                data.get('points.cam_pos').reshape(1, 5, 3)[0].cpu().numpy(), data.get('points.cam_rot').reshape(1, 5, 3)[0].cpu().numpy(),
                data.get('inputs.pc_ply')[0].float().cpu().numpy(), data.get('inputs.touch_success')[0].cpu().numpy())
            return {'feats': c_img[0], 'anchors': torch.from_numpy(anchors).float(), 'success': torch.from_numpy((count > 0).astype('uint8')),
                    'mode': 'within', 'radius': 0.015, 'count': torch.from_numpy(count).int()}
        return host_part if sides else host_part()

    def _replay_on(self, stream, kind, tensors, run):
        """A clone of ``_replay(kind, tensors, run)``'s result, queued on ``stream`` (None: the current one).  The clone is marked as
        used by the caller's stream, which reads it behind its ``wait_stream``."""
        if stream is None:
            return self._replay(kind, tensors, run).clone()
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            out = self._replay(kind, tensors, run).clone()
        out.record_stream(cur)
        return out

    def _setup_vtacoh(self, data, sides=None):
        """The VTacOH branch of generate_obj_mesh_wnf (generation.py:161-200): fingertips from the hand encoder's MANO joints
        in the object's frame (ground-truth wrist position and wrist Euler angles from the sample), every lattice point
        within 0.05 of its nearest fingertip takes that finger's tactile feature if its touch succeeded -- by finger id
        (vt_tactile_assign + vt_decode_fwd_ids) instead of the reference's dense [1, nx^3, C] tensor and CPU cdist."""
        from ..common import fingertips_in_object_frame
        if getattr(self.model, 'encoder_hand', None) is None or getattr(self.model, 'encoder_img', None) is None:
            raise VtError("generate_obj_mesh_wnf(with_img): the model needs encoder_hand (fingertips) and encoder_img "
                          "(tactile features), or pass c_img_all / use generate_obj_mesh_tactile")
        inputs = data.get('inputs').to(self.device)
        if inputs.shape[0] != 1:
            raise VtError(f"generate_obj_mesh_wnf: one scene at a time (got a batch of {inputs.shape[0]})")
        self._eval_mode()
        # the tactile features first (the longest of the scene's encoders), then the hand encoder: plane PointNet + 2-D U-Net + MANO as one graph
        c_img = self._replay_on(sides[0] if sides else None, "encode_img", [data.get('inputs.img')], self.model.encode_img_inputs)    # [1,5,C]
        hand_stream = sides[1] if sides else torch.cuda.current_stream(self.device)
        with torch.cuda.stream(hand_stream):
            c_hand = self._replay("encode_hand_inputs", [inputs], self.model.encode_hand_inputs)
            if 'mano_joints' not in c_hand:
                raise VtError("generate_obj_mesh_wnf(with_img): the hand encoder has no MANO layer (out_dim <= 30)")
            joints_dev = c_hand['mano_joints'].float()

        def host_part():
            with torch.cuda.stream(hand_stream):                    # (the copy back waits for the hand encoder's stream only)
                joints = joints_dev.cpu().numpy()
            tips = fingertips_in_object_frame(joints, data.get('points.mano').cpu().numpy()[:, :3],
                                              data.get('points.wrist').cpu().numpy(), data.get('inputs.pc_ply').float().cpu().numpy())
            anchors = torch.from_numpy(tips[0]).float().unsqueeze(1)                               # [5,1,3]
            return {'feats': c_img[0], 'anchors': anchors, 'success': data.get('inputs.touch_success')[0].to(torch.uint8),
                    'mode': 'nearest', 'radius': 0.05, 'count': torch.ones(5, dtype=torch.int32)}
        return host_part if sides else host_part()
