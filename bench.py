#!/usr/bin/env python3
"""bench.py -- occupancy query-points/s at 128^3 on MI355X (BASELINE.json metric).

A "step" is one pass of the decode hot path (lattice -> logits grid on device:
in-kernel lattice generation, trilinear gather of the channels-last feature grid,
per-point conditioned ResNet MLP) over one synthetic scene; weights and feature
grid are resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W

* ``--gpus N`` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU and starts
  N ranks (``python -m torch.distributed.run``, one process per GPU, RCCL) of itself; under a launcher
  (WORLD_SIZE set) ``--gpus`` must agree with it.
* ``--scaling weak`` (default): every rank decodes its own scene -- the unit is the query point, there is
  no data-path collective -- value = N * nx^3 * steps / max-over-ranks time.
  ``--scaling strong``: ONE scene, every rank decodes its slab of the lattice (whole x-plane pairs) and one
  all-gather rebuilds the value grid on every rank; value = nx^3 * steps / time (N = 1: the same number).
* Unless ``--decode-only``: the same JSON line carries ``sharded_scene`` (BASELINE config 5's shape of work:
  redundant encode, slab decode, one all-gather, marching cubes on rank 0; at 128^3 and 256^3) and
  ``train_step`` (config 4: the VTacO training step, 8 scenes x 2048 points per GPU, bucketed gradient
  all-reduce over RCCL overlapped with backward; ms/step and the all-reduce's share), and at N = 1 the
  mesh-extract latency, the stage times and the CPU baseline.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 31488        # 30 976 (16 linear layers) + 512 (8 corners x 32 ch FMA)
FLOP_PER_POINT_IMG = 33536    # with the tactile concat (forward_img)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
KERNEL_OF = {"f32": "decode_fwd_staged2_kernel<0>", "bf16x3": "decode_fwd_staged2_kernel<1>", "f16x3": "decode_fwd_staged3_kernel<1>",
             "f16f8": "decode_fwd_staged3_kernel<2>"}
SPLIT = ("bf16x3", "f16x3", "f16f8")   # dense layers on the 16-bit matrix core, operands as hi + lo
MIN_WARM_S = 0.25             # launches before any timed region, whatever --warmup says (clocks settle)
MIN_TIMED_S = 0.05            # the timed region is REPS x --steps steps, REPS chosen so that it lasts at least this long


# ---------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a launcher around us
# ---------------------------------------------------------------------------------------------------------------------

def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n ranks of this script as CHILD processes (this process has not touched the GPU and never will; a process
    that has must not be replaced by exec on this pool) and exit with their code.  Rank 0's JSON line passes through."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    if env.get("VTACO_BENCH_BACKEND", "nccl") == "nccl" and "--dry-run" not in argv:
        import torch                                   # (device_count() does not initialise the GPU on this image)
        have = torch.cuda.device_count()
        if 0 < have < n:
            sys.exit(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {have} (RCCL refuses ranks that share a device; "
                     "VTACO_BENCH_BACKEND=gloo dry-runs the multi-process path on fewer)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


# ---------------------------------------------------------------------------------------------------------------------
# pieces
# ---------------------------------------------------------------------------------------------------------------------

def cpu_baseline(scene, nx, budget_s=15.0):
    """The oracle (a port, not the reference itself) timed on the host cores: the decode metric on a bounded sample -- whole
    100k-point chunks of the same lattice until ~budget_s -- and, beside it, the other two stages of the scene on the same
    inputs (encode / decode in 100k chunks / C marching cubes): the PointNet + UNet3D encode of the bench cloud
    (median of 3) and oracle/mc_lewiner.c on the oracle's own 128^3 logit grid of a coarser lattice pass (median of 3)."""
    import numpy as np
    import torch
    from oracle import vtaco_oracle as orc
    sd, grid_cpu = scene["sd_decoder_cpu"], scene["grid_cpu"]
    # torch's CPU kernels on 100k x 32 operands stop scaling (and then collapse) beyond a few
    # tens of threads; use what the host has, capped at 32, and report that number as `cores`
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    pts = 1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
    chunks = torch.split(pts, 100000)
    orc.local_decoder_forward(sd, chunks[0][:1000].unsqueeze(0), grid_cpu)   # warm
    done, t0 = 0, time.perf_counter()
    for ch in chunks:
        orc.local_decoder_forward(sd, ch.unsqueeze(0), grid_cpu)
        done += ch.shape[0]
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    res = {"value": done / dt, "unit": "query-points/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{done} of {nx ** 3} lattice points in 100k-point chunks, oracle/vtaco_oracle.py (torch CPU f32)"}
    stages = {}
    try:
        enc = scene["model"].encoder
        if enc is not None:
            esd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                orc.pointnet_encoder_forward(esd, scene["cloud"], 64)
                ts.append(time.perf_counter() - t0)
            stages["encode_pointnet_unet3d_ms"] = 1e3 * sorted(ts)[1]
        # (the sample is the whole lattice when it fits the time budget: then this is a measurement, otherwise it is scaled up)
        stages["decode_lattice_ms" if done >= nx ** 3 else "decode_lattice_ms_scaled_from_sample"] = 1e3 * dt * nx ** 3 / done
        from oracle import mc as omc
        vol = scene.get("logits_cpu")                                      # the GPU's logit grid of this scene (same surface)
        if vol is not None:
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                v, f = omc.marching_cubes(np.ascontiguousarray(vol))[:2]
                ts.append(time.perf_counter() - t0)
            stages["marching_cubes_ms"] = 1e3 * sorted(ts)[1]
            stages["marching_cubes_note"] = f"oracle/mc_lewiner.c, one thread, {len(v)} vertices / {len(f)} faces"
        stages["threads_torch"] = torch.get_num_threads()
    except Exception as e:                                                 # noqa: BLE001 -- the decode baseline stands on its own
        stages["error"] = f"{type(e).__name__}: {e}"[:300]
    res["stages_ms"] = stages
    return res


def mesh_extract_stats(vol, nx, runs=100):
    """mesh-extract latency (BASELINE.json metric, second half): HIP marching cubes on the
    device-resident logit grid -> device verts/faces, including the one host read of the
    counts that sizes the outputs.  p50 over `runs`."""
    import torch
    from vtaco_amd import ops
    lat = []
    for i in range(runs + 5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v, f, lvl = ops.marching_cubes(vol, None, rescale=(nx / 2, 1.1 / nx))
        torch.cuda.synchronize()
        if i >= 5:
            lat.append(1e3 * (time.perf_counter() - t0))
    lat.sort()
    p50 = lat[len(lat) // 2]
    nbytes = 4 * nx ** 3 + 12 * v.shape[0] + 12 * f.shape[0]      # algorithmic bytes
    return {"p50_ms": p50, "p90_ms": lat[int(len(lat) * 0.9)], "min_ms": lat[0], "runs": runs, "verts": v.shape[0],
            "faces": f.shape[0], "level": lvl, "algorithmic_GBps": nbytes / (p50 * 1e-3) / 1e9,
            "note": "latency-dominated (five dependent launches, the counts polled from a page-locked slot); 8.4 MB volume = 1.3 us at HBM rate"}


def stage_times(scene, dec, grid, nx, out, dev, precision):
    """End-to-end stages for one scene (reported beside the metric, not part of it)."""
    import torch

    def timed(fn, n=5):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return sorted(ts)[len(ts) // 2]
    res = {}
    model = scene["model"]
    if model.encoder is not None:
        pc = scene["cloud"].to(dev)
        with torch.no_grad():
            res["encode_pointnet_unet3d"] = timed(lambda: model.encode_inputs(pc))
    res["decode_lattice"] = timed(lambda: dec.decode_lattice(grid, nx, box=1.1, out=out, precision=precision))
    from vtaco_amd import ops
    res["marching_cubes"] = timed(lambda: ops.marching_cubes(out.view(nx, nx, nx), None, rescale=(nx / 2, 1.1 / nx)))
    res["end_to_end"] = sum(res.values())
    if model.encoder is not None:
        from vtaco_amd.conv_onet.generation import Generator3D
        gen = Generator3D(model, device=dev, resolution0=nx // 4, padding=0.1, decode_precision=precision)
        gen.generate_mesh_graphed(pc)               # builds + captures
        res["end_to_end_hipgraph"] = timed(lambda: gen.generate_mesh_graphed(pc), 20)
        # the reference's entry point (generation.py:117-273): replays the same graph by default
        res["generate_obj_mesh_wnf"] = timed(lambda: gen.generate_obj_mesh_wnf({"inputs": pc}), 20)
    return res


PMC_SUMMARY = os.path.join("profiles", "r06_pmc_summary.csv")
EXTRAS_LIMIT_S = 420            # the sections after the headline (sharded scene, training step, CPU baseline) take well under a minute
DECODE_SOURCES = ("decode.hip", "decode_f16.hip", "decode_common.h", "decode_st3.h", "decode_st3_f16x3.inc", "decode_st3_f16f8.inc",
                  "vt_common.h", "Makefile")


WIDE_SOURCES = ("decode_wide.hip", "decode_wide_pipe.inc", "decode_common.h", "vt_common.h", "Makefile")
PMC_WIDE_SUMMARY = os.path.join("profiles", "r06_pmc_wide_summary.csv")


def source_hash(names):
    import hashlib
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(ROOT, "vtaco_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def decode_source_hash():
    """sha256 over the sources the decode kernels are built from (tools/src_hash.py prints the same): the committed counter
    summary carries it in its first line, and counters collected on other sources are not reported."""
    return source_hash(DECODE_SOURCES)


def pmc_wide_table():
    """{kernel key: {counter: mean per launch}} of the general-shape decoder's kernels (tools/pmc_wide.sh), or {} when the committed
    summary was collected on other sources than this tree's (its first line carries their hash)."""
    tab, stamp = {}, None
    try:
        for line in open(os.path.join(ROOT, PMC_WIDE_SUMMARY)).read().splitlines():
            if line.startswith("#"):
                if "wide_sources=" in line:
                    stamp = line.split("wide_sources=", 1)[1].strip()
                continue
            if line.startswith("kernel,"):
                continue
            k, c, n, mean = line.split(",")
            tab.setdefault(k, {})[c] = float(mean)
    except Exception:
        return {}
    return tab if stamp == source_hash(WIDE_SOURCES) else {}


_pmc_cache = {}


def pmc_table():
    """{kernel key: {counter: mean per launch}} from the committed PMC passes -- only if the file's stamp (`# decode_sources=<hash>`,
    written by tools/collect_profiles.sh on the box that collected them) equals the hash of the sources in this tree; a summary
    collected on other kernel sources gives {} (the line then carries no `pmc` / `traffic` instead of stale ones)."""
    if "t" in _pmc_cache:
        return _pmc_cache["t"]
    tab, stamp = {}, None
    try:
        for line in open(os.path.join(ROOT, PMC_SUMMARY)).read().splitlines():
            if line.startswith("#"):
                if "decode_sources=" in line:
                    stamp = line.split("decode_sources=", 1)[1].strip()
                continue
            if line.startswith("kernel,"):
                continue
            k, c, n, mean = line.split(",")
            tab.setdefault(k, {})[c] = float(mean)
        if stamp != decode_source_hash():
            tab = {}
    except Exception:
        tab = {}
    _pmc_cache["t"] = tab
    return tab


def pmc_counters(precision):
    """Per-launch counter sums of this precision's decode kernel from the committed PMC passes (or {})."""
    return pmc_table().get("decode_" + precision, {})


def measured_traffic(precision):
    """HBM-side bytes per decode launch from the committed PMC passes (profiles/, same command
    as this bench): (2 x FETCH_SIZE + WRITE_SIZE) KB -- the x2 is the guide's gfx950 correction
    for 16-B-per-lane reads; None if no profile is committed for this kernel."""
    vals = pmc_counters(precision)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def roofline_of(precision, flop_pt, npts, kern_ms, clock=None):
    """Roofline object of one decode kernel: algorithmic FLOP / HIP-event time against the dense
    MFMA peak of the matrix-core input type it runs on."""
    achieved = flop_pt * npts / (kern_ms * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if precision in SPLIT else PEAK_F32_MFMA_TFLOPS
    r = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
         "traffic": measured_traffic(precision),
         "traffic_note": "bytes/launch at the L2's memory side from rocprofv3 FETCH_SIZE/WRITE_SIZE passes "
                         f"({PMC_SUMMARY}, stamped with the hash of the kernel sources it was collected on; null when that is "
                         "not this tree's); algorithmic = 33.5 MB grid + 8.4 MB logits",
         "kernel": KERNEL_OF[precision], "kernel_ms": kern_ms, "flop_per_point": flop_pt}
    if clock is not None:
        r["clock"] = clock
    pmc = pmc_counters(precision)
    if "GRBM_GUI_ACTIVE" in pmc and "SQ_VALU_MFMA_BUSY_CYCLES" in pmc:
        simd_cycles = 1024.0 * pmc["GRBM_GUI_ACTIVE"] / 8.0                 # 256 CUs x 4 SIMDs x cycles per launch (8 XCDs summed)
        r["pmc"] = {"matrix_pipe_busy": pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                    "valu_issue": 4.0 * pmc.get("SQ_INSTS_VALU", 0.0) / simd_cycles,
                    "source": PMC_SUMMARY + " (profiled launches of the same command on the same kernel sources)"}
    if precision == "f16f8":
        r["note"] = ("f16 hi products (2 MFMAs per 32x32x32 layer half) + ONE fp8 32x32x64 MFMA for both correction products: 128 matrix "
                     "cycles per layer against 192 for three f16 products and 64 for a plain f16 layer, so frac <= 0.50 by construction; "
                     "the rest is VALU issue (relu, hi/lo split, fp8 conversion: see `pmc`) and the clock the chip holds under this load; "
                     "DESIGN.md section 4")
        r["vs_f32_mfma_roofline"] = achieved / PEAK_F32_MFMA_TFLOPS
    elif precision in SPLIT:
        r["note"] = ("split 16-bit operands: each f32 product = 3 MFMA products (lo*hi + hi*lo + hi*hi) on the bf16 / f16 matrix "
                     "core (same rate), so the matrix pipe executes ~3x the algorithmic FLOP: frac <= 0.33 by construction; the "
                     "rest is VALU issue (relu, hi/lo split, trilinear FMAs: see `pmc`); DESIGN.md section 4")
        r["vs_f32_mfma_roofline"] = achieved / PEAK_F32_MFMA_TFLOPS     # SURVEY.md 8d's binding roofline for f32 results
    return r


class Fences:
    """barrier + synchronize on both sides of a timed region; max over ranks of a host time."""

    def __init__(self, dist, dev, backend):
        self.dist, self.dev, self.backend = dist, dev, backend

    def fence(self):
        import torch
        if self.dist is not None:
            self.dist.barrier()
        if self.dev is not None:
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        import torch
        if self.dist is None:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def warm_up(step, min_steps, fx, min_s=MIN_WARM_S, collective=False):
    """At least `min_steps` untimed steps AND at least `min_s` seconds of them.  A step that contains a collective must run
    the SAME number of times on every rank: the ranks then agree on "enough" after every round of 8 steps (max over ranks
    of the elapsed time) instead of each consulting its own clock."""
    import torch
    t0, n = time.perf_counter(), 0
    while True:
        for _ in range(8):
            step()
        n += 8
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if collective:
            elapsed = fx.max_over_ranks(elapsed)
        if n >= min_steps and elapsed >= min_s:
            return n


def clock_of_last_launch(vops):
    """In-kernel stamps of the last lattice launch (ops.decode_last_clock) without the raw per-workgroup list."""
    c = vops.decode_last_clock(workgroups=True)
    if c is not None:
        c.pop("wg_ticks", None)
        c["note"] = ("shader_mhz = workgroup 0's s_memtime cycles / its s_memrealtime ticks (the clock the chip held under this kernel); "
                     "span_us = first workgroup start to last workgroup end of that launch, start_spread_us = the dispatch ramp, wg_us_* = "
                     "workgroup lifetimes (persistent workgroups, equal tile counts)")
    return c


def timed_region(step, steps, fx, collective=False):
    """The timed region of one kernel: REPS back-to-back repetitions of EXACTLY `steps` steps between two barrier +
    synchronize fences, REPS = the smallest count that makes the region last MIN_TIMED_S (a 20-step region of this kernel is
    ~3 ms: the fences' own ~0.3 ms was a tenth of it in round 3's record).  One HIP event per repetition on the launch stream
    gives the per-repetition kernel time: their spread is in the line.  Returns (wall seconds max over ranks, timed steps,
    stats of the per-repetition ms per step)."""
    import math
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 8
    if collective or fx.dist is not None:
        est = fx.max_over_ranks(est)                       # every rank times the SAME number of steps (value = all ranks' units / max time)
    reps = max(1, min(4096, int(math.ceil(MIN_TIMED_S / max(est * steps, 1e-9)))))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    fx.fence()
    t0 = time.perf_counter()
    ev[0].record()
    for r in range(reps):
        for _ in range(steps):
            step()
        ev[r + 1].record()
    fx.fence()
    wall = fx.max_over_ranks(time.perf_counter() - t0)
    per = [ev[r].elapsed_time(ev[r + 1]) / steps for r in range(reps)]
    mean = sum(per) / reps
    std = (sum((x - mean) ** 2 for x in per) / reps) ** 0.5
    stats = {"reps": reps, "steps_per_rep": steps, "timed_steps": reps * steps, "kernel_ms_mean": mean, "kernel_ms_std": std,
             "kernel_ms_min": min(per), "kernel_ms_max": max(per), "std_over_mean": std / mean if mean > 0 else None,
             "kernel_ms_total_events": ev[0].elapsed_time(ev[reps]) / (reps * steps),
             "wall_over_events": wall * 1e3 / ev[0].elapsed_time(ev[reps]) if ev[0].elapsed_time(ev[reps]) > 0 else None}
    return wall, reps * steps, stats


def sharded_scene(scene, dev, fx, rank, world, dist, precision, sizes=(128, 256), iters=10):
    """BASELINE config 5's shape of work, strong scaling over the ranks: ONE scene; every rank encodes it (cheap, deterministic:
    no broadcast), decodes its slab of x-plane pairs, one all-gather of the logit slabs rebuilds the value grid, rank 0 extracts
    the mesh.  Times are per scene, max over ranks; the stage breakdown is rank 0's (HIP events on the launch stream)."""
    import torch
    from vtaco_amd import dist as vdist
    from vtaco_amd.conv_onet.generation import Generator3D
    model = scene["model"]
    pc = scene["cloud"].to(dev)
    res = {}
    for nx in sizes:
        gen = Generator3D(model, device=dev, resolution0=nx // 4, padding=0.1, decode_precision=precision)
        total = nx ** 3
        align = vdist.lattice_align(nx, world)
        first, count = vdist.slab_of(total, rank, world, align)
        full = torch.empty(total, dtype=torch.float32, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        mesh = [None]

        def one(record=False):
            with torch.no_grad():
                if record:
                    ev[0].record()
                c = gen._replay("encode_inputs", [pc], model.encode_inputs)      # as generate_obj_mesh_sharded: the encoder as one graph
                if record:
                    ev[1].record()
                local = gen.eval_lattice(c, nx, first=first, count=count) if count else torch.empty(0, dtype=torch.float32, device=dev)
                if record:
                    ev[2].record()
                vol = vdist.all_gather_slabs(local, total, None, align, out=full) if world > 1 else local
                if record:
                    ev[3].record()
                if rank == 0:
                    mesh[0] = gen.extract_mesh(vol.reshape(nx, nx, nx))
                if record:
                    ev[4].record()
        warm_up(one, 2, fx, 0.1, collective=world > 1)
        fx.fence()
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        fx.fence()
        ms = 1e3 * fx.max_over_ranks(time.perf_counter() - t0) / iters
        one(True)
        torch.cuda.synchronize()
        st = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
        per_rank = None
        if world > 1:
            # every rank's stage times in the line: a sub-linear curve can then be read without a re-run
            mine = torch.tensor(st, dtype=torch.float64, device=dev if fx.backend == "nccl" else "cpu")
            rows = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(rows, mine)
            per_rank = [{"encode": float(r[0]), "decode_slab": float(r[1]), "all_gather": float(r[2]), "marching_cubes": float(r[3])}
                        for r in rows]
        res[str(nx)] = {"ms_per_scene": ms, "points_per_s": total / (ms * 1e-3), "slab_points_per_rank": count,
                        "rank0_stage_ms": {"encode": st[0], "decode_slab": st[1], "all_gather": st[2], "marching_cubes": st[3]},
                        "per_rank_stage_ms": per_rank,
                        "all_gather_bytes": 4 * total,
                        "verts": int(mesh[0].vertices.shape[0]) if rank == 0 else None}
    return res


def train_step_section(dev, fx, rank, world, dist, steps=8, scenes=8):
    """BASELINE config 4: the VTacO training step (forward + backward + gradient all-reduce + Adam) on `scenes` synthetic scenes x
    2048 query points per GPU; weak scaling (global batch = 8 N: 64 scenes at N = 8).  The shipped configuration: the t2d net is
    PRETRAINED (configs/VTacO/VTacO_YCB.yaml:65; here a synthetic checkpoint loaded through the factory) and therefore not trained
    (training.py:749-752).  Beside it: the all-reduce's share = (ms with the bucketed all-reduce) - (ms of the same step without
    any gradient exchange); at N = 1 also the step with the t2d net trained (`pretrained: False`, what round 2 timed) and the part
    of the step that runs on this repository's kernels (the same model's visual + hand branches alone: PointNet, voxeliser, UNet3D,
    decoder, hand encoder forward + backward + Adam -- the tactile U-Net and Resnet18 stay host PyTorch by north_star)."""
    import numpy as np
    import torch
    from vtaco_amd.bench_util import build_train_case
    B = scenes
    model, trainer, batch, vf = build_train_case(dev, rank, scenes=B, pretrained_t2d=True)
    sync = trainer.grad_sync
    n_param = sum(p.numel() for p in model.parameters())
    np.random.seed(1234 + rank)

    def timed_steps(tr, n, warm=2):
        def step():
            tr.train_step(batch, vf)
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        first_s = time.perf_counter() - t0
        for _ in range(warm):                                           # untimed: MIOpen's find / first-touch work is over
            step()
        fx.fence()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        fx.fence()
        return 1e3 * fx.max_over_ranks(time.perf_counter() - t0) / n, first_s
    ms_sync, first_s = timed_steps(trainer, steps)
    res = {"ms_per_step": ms_sync, "scenes_per_s": world * B / (ms_sync * 1e-3), "global_batch": world * B,
           "points_per_scene": 2048, "parameters": n_param, "allreduce_bytes": 4 * sync.numel, "first_step_s": first_s,
           "t2d_pretrained": True,
           "workload": "Trainer(with_img, encode_t2d).train_step: shipped VTacO model (get_model) with the pretrained (frozen) t2d net, "
                       "contact clouds from depth images, winding-number targets, Adam 1e-4; synthetic batch; the tactile Resnet18 "
                       "(host PyTorch / MIOpen) runs once over all scenes' images with every BatchNorm on each scene's statistics alone "
                       "(= the reference's per-scene loop; VTACO_TACTILE_SCENE_BATCH=0 runs the loop)"}
    if world > 1:
        res["buckets"] = dict(sync.stats)
        # the all-reduce alone: the same buckets, nothing to overlap with
        flats = [bk["flat"] for bk in sync._buckets]
        fx.fence()
        t0 = time.perf_counter()
        for _ in range(5):
            works = [dist.all_reduce(f, async_op=True) for f in flats]
            for w in works:
                w.wait()
        fx.fence()
        res["allreduce_alone_ms"] = 1e3 * fx.max_over_ranks(time.perf_counter() - t0) / 5
        res["allreduce_alone_GBps_per_gpu"] = 2 * (world - 1) / world * 4 * sync.numel / (res["allreduce_alone_ms"] * 1e-3) / 1e9
        # assertion-free diagnostic: where the measured all-reduce sits between the two wire-time bounds of a fully connected xGMI node
        # (7 links x ~153 GB/s per GPU): a RING moves 2 (w-1)/w S bytes over ONE link per GPU; a DIRECT
        # reduce-scatter + all-gather sends S/w to every peer at once, twice
        link, S = 153e9, 4.0 * sync.numel
        ring_ms, direct_ms = 1e3 * 2 * (world - 1) / world * S / link, 1e3 * 2 * (S / world) / link
        res["allreduce_alone_vs_xgmi_bounds"] = {
            "ring_bound_ms": ring_ms, "direct_bound_ms": direct_ms, "measured_over_ring": res["allreduce_alone_ms"] / ring_ms,
            "measured_over_direct": res["allreduce_alone_ms"] / direct_ms,
            "reads_as": "ring-like (per-link bound)" if res["allreduce_alone_ms"] > 0.7 * ring_ms else "uses several links at once",
            "note": f"{world} ranks, {S / 1e6:.1f} MB of gradients in {len(flats)} buckets launched together; wire time only (no reduction arithmetic, no launch latency)"}
        # the same step with no gradient exchange at all: the hooks are switched off too (they would launch the buckets)
        with sync.no_sync():
            ms_local, _ = timed_steps(trainer, steps, warm=1)
        res["ms_per_step_without_allreduce"] = ms_local
        res["allreduce_share"] = max(0.0, (ms_sync - ms_local) / ms_sync)
    else:
        try:
            from vtaco_amd.conv_onet.training import Trainer
            vis = Trainer(model, trainer.optimizer, device=dev, input_type="pointcloud", threshold=0.5, num_sample=2048,
                          with_img=False, encode_t2d=False)
            res["ms_hip_part"], _ = timed_steps(vis, max(2, steps // 2), warm=1)
            res["ms_hip_part_note"] = "Trainer(with_img=False, encode_t2d=False).train_step on the same model and batch"
        except Exception as e:                                           # noqa: BLE001
            res["ms_hip_part_error"] = f"{type(e).__name__}: {e}"[:300]
        del trainer, model
        torch.cuda.empty_cache()
        model2, trainer2, batch, vf = build_train_case(dev, rank, scenes=B, pretrained_t2d=False)
        res["ms_per_step_t2d_trained"], _ = timed_steps(trainer2, max(2, steps // 2), warm=1)
    return res


def _median_ms(fn, n=7, warm=2):
    import torch
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]


def _route_stages(gen, model, data, nx, n=7):
    """generate_obj_mesh_wnf(with_img) end to end (median wall ms, device idle on both sides) and its stages timed one by one the
    same way: shape encoder (a graph replay), the branch's setup (hand encoder / Resnet18 replays + the host-side anchors), finger ids
    + decode of the lattice, marching cubes."""
    import torch
    res = {"end_to_end_ms": _median_ms(lambda: gen.generate_obj_mesh_wnf(data), n)}
    inputs = data["inputs"].to(gen.device)
    box = {}

    def enc():
        box["c"] = gen._replay("encode_inputs", [inputs], model.encode_inputs)

    def setup():
        box["setup"] = gen._tactile_setup(data)

    def lattice():
        with torch.no_grad():
            box["vals"] = gen._eval_lattice_tactile(box["c"], nx, box["setup"])

    def mc():
        box["mesh"] = gen.extract_mesh(box["vals"].reshape(nx, nx, nx))
    res["stage_ms"] = {"encode_pointnet_unet3d": _median_ms(enc, n), "tactile_setup": _median_ms(setup, n),
                       "finger_ids_and_decode": _median_ms(lattice, n), "marching_cubes": _median_ms(mc, n)}
    res["stage_note"] = ("stages timed ONE BY ONE; end_to_end_ms is the entry point itself, which replays the scene's independent encoders "
                         "(shape, tactile features, hand) on three HIP streams at once (Generator3D._generate_tactile): it is less than the "
                         "stages' sum")
    res["verts"], res["faces"] = int(box["mesh"].vertices.shape[0]), int(box["mesh"].faces.shape[0])
    return res, box


def _kernel_ms(fn, name_part):
    """(average device ms per launch, launches) of the kernels whose name contains ``name_part`` during one call of fn (torch.profiler
    = roctracer on this image; rocprofv3 of tools/pmc_fusion.sh gives the same table offline: profiles/r06_fusion_kernel_stats.csv)."""
    import torch
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA and name_part in e.key]
    n = sum(e.count for e in ev)
    return (sum(e.device_time_total for e in ev) / 1e3 / n, n) if n else (None, 0)


def config3_section(dev, precision, nx=128):
    """BASELINE config 3 -- full VTacO at 128^3 on one GPU, four numbers:
    (a) the tactile-concat decode of the lattice (LocalDecoder.forward_img, the SHIPPED VTacO decoder): dense c_img and by finger id;
    (b) AttentionDecoder.forward_img over the lattice in chunks of 2048 points, EVERY chunk through the three attention units, with a
        roofline object for the pass (HIP events) and one for its dominant kernel, fusion_attend_kernel (192 N^2 FLOP per chunk:
        2 N^2 64 for the scores + 2 N^2 32 for A V'; SURVEY.md 8d's 576 N per point is three of these);
    (c) Generator3D(with_img, encode_t2d).generate_obj_mesh_wnf -- the reference's entry point (generation.py:202-257, 268-273) --
        end to end on the shipped VTacO model with either decoder, with stage times; beside it the t2d net's forward, which the
        reference runs and whose result it discards (:212-217, 226: the dataset's depth overwrites the predicted one);
    (d) the oracle on the host cores for the stages it has (a bounded sample: 4 chunks of the attention decoder, the assignment
        rule on 1/64 of the lattice)."""
    import numpy as np
    import torch
    from vtaco_amd import ops
    from vtaco_amd.bench_util import build_tactile_scene
    from vtaco_amd.conv_onet.generation import Generator3D
    res = {}
    npts = nx ** 3
    # ---- (c) first: it builds the models the other parts reuse ----
    routes = {}
    keep = {}
    for decoder in ("simple_local", "attention_local"):
        model, data, origin = build_tactile_scene(dev, "vtaco", decoder)
        gen = Generator3D(model, device=dev, resolution0=nx // 4, padding=0.1, with_img=True, encode_t2d=True,
                          points_batch_size=2048 if decoder == "attention_local" else 100000, decode_precision=precision,
                          depth_origin=origin)
        np.random.seed(11)
        r, box = _route_stages(gen, model, data, nx)
        ids = ops.tactile_assign(box["setup"]["anchors"].to(dev), box["setup"]["success"].to(dev), "within", 0.015,
                                 lattice=(nx, 1.1, 0, npts), count=box["setup"]["count"].to(dev))
        r["lattice_points_with_a_tactile_feature"] = int((ids != 255).sum())
        if decoder == "attention_local":
            r["chunks"] = npts // 2048
            r["chunks_with_a_tactile_feature"] = int((ids.reshape(-1, 2048) != 255).any(dim=1).sum())
            r["note"] = "chunks no finger touches skip the fuser (fuse(0, c) = 0 exactly: Generator3D._eval_lattice_fused); (b) is the pass without that shortcut"
        routes[decoder] = r
        keep[decoder] = (model, gen, data, box, ids)
    model, gen, data, box, ids = keep["simple_local"]
    imgs, inputs = data["inputs.img"].to(dev), data["inputs"].to(dev)
    with torch.no_grad():
        t2d_ms = _median_ms(lambda: model.encode_t2d(inputs, imgs), 5, 2)
    res["generate_obj_mesh_wnf_t2d"] = dict(routes, encode_t2d_forward_ms=t2d_ms,
                                            encode_t2d_note="tactile U-Net (5 images of 320 x 240, host PyTorch / MIOpen by north_star) + digit-pose "
                                                            "encoder: the reference runs it in this branch and discards both results; "
                                                            "this generator does not run it (same mesh)")
    # ---- (a) tactile concat over the lattice ----
    dec, grid = model.decoder, box["c"]["grid"]
    feats = box["setup"]["feats"].to(dev).float()
    dense = torch.zeros(1, npts, 32, device=dev)
    hit = ids[0] != 255
    dense[0, hit] = feats[ids[0][hit].long()]
    out = torch.empty((1, npts), dtype=torch.float32, device=dev)
    with torch.no_grad():
        t_dense = _median_ms(lambda: dec.decode_lattice(grid, nx, c_img=dense, out=out, precision=precision), 20, 5)
        t_ids = _median_ms(lambda: dec.decode_lattice_ids(grid, nx, ids, feats, out=out, precision=precision), 20, 5)
    res["tactile_concat_decode"] = {
        "dense_c_img_ms": t_dense, "by_finger_id_ms": t_ids, "points_per_s_by_id": npts / (t_ids * 1e-3), "precision": precision,
        "tflops_dense": FLOP_PER_POINT_IMG * npts / (t_dense * 1e-3) / 1e12,
        "note": "LocalDecoder.forward_img over the 128^3 lattice: dense = [1, nx^3, 32] f32 c_img_all (268 MB read per pass), by id = one byte "
                "per point + the [5, 32] table (tiles no finger touches skip fc_p_img's c_img columns)"}
    del dense
    # ---- (b) the attention decoder, every chunk through the fuser ----
    amodel, agen, adata, abox, aids = keep["attention_local"]
    adec, agrid = amodel.decoder, abox["c"]["grid"]
    afeats = abox["setup"]["feats"].to(dev).float()
    N, chunks = 2048, npts // 2048
    agen.skip_untouched_chunks = False
    setup = abox["setup"]
    with torch.no_grad():
        run = lambda: agen._eval_lattice_fused(abox["c"], nx, aids, afeats)
        t_pass = _median_ms(run, 5, 2)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(3):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        t_pass_ev = ev[0].elapsed_time(ev[1]) / 3
        att_ms, att_n = _kernel_ms(run, "fusion_attend_kernel")
        exp_ms, exp_n = _kernel_ms(run, "fusion_expsum")
    agen.skip_untouched_chunks = True
    flop_pt = 30976 + 576 * N + 61440
    tf = flop_pt * npts / (t_pass_ev * 1e-3) / 1e12
    att = None
    if att_ms:
        chunks_per_launch = 3 * chunks / att_n                       # three attention units per chunk
        att_tf = 192.0 * N * N * chunks_per_launch / (att_ms * 1e-3) / 1e12
        att = {"bound": "mfma", "achieved": att_tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": att_tf / PEAK_BF16_MFMA_TFLOPS,
               "traffic": None, "kernel": "fusion_attend_kernel", "kernel_ms": att_ms, "launches_per_pass": att_n,
               "chunks_per_launch": chunks_per_launch, "flop_per_chunk": 192 * N * N,
               "note": "algorithmic 192 N^2 FLOP per chunk and unit (scores once + A V' once) over the kernel's average device time "
                       "(torch.profiler, live); the kernel recomputes the score tile with split operands (3 half products or f16 + fp8 "
                       "corrections), so the matrix pipe executes ~2.3x this; counters: profiles/r06_fusion_pmc_summary.csv"}
    res["attention_decoder_dense"] = {
        "ms_per_lattice": t_pass, "ms_per_lattice_hip_events": t_pass_ev, "points_per_s": npts / (t_pass * 1e-3), "chunk_points": N, "chunks": chunks,
        "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_MFMA_TFLOPS,
                     "traffic": None, "flop_per_point": flop_pt,
                     "note": "SURVEY.md 8d: 30 976 + 576 N + 61 440 FLOP per point at N = 2048, over the HIP-event time of the whole pass "
                             "(grid sample, three attention units, conditioned MLP)"},
        "attend_kernel_roofline": att,
        "expsum_kernels": {"avg_ms": exp_ms, "launches_per_pass": exp_n}}
    # ---- (d) the oracle on the host cores, bounded samples ----
    try:
        from oracle import vtaco_oracle as orc
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        cpu = {"cores": torch.get_num_threads(), "kind": "port"}
        sd = {k: v.detach().cpu() for k, v in adec.state_dict().items()}
        gcpu = agrid.detach().cpu().contiguous()
        pts = 1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)
        k = 4
        ci = torch.zeros(1, N, 32)
        t0 = time.perf_counter()
        for j in range(k):
            orc.attention_decoder_forward_img(sd, pts[j * N:(j + 1) * N].unsqueeze(0), gcpu, ci)
        dt = time.perf_counter() - t0
        cpu["attention_decoder_ms_per_lattice_scaled"] = 1e3 * dt * chunks / k
        cpu["attention_decoder_sample"] = f"{k} of {chunks} chunks of {N} points, oracle.attention_decoder_forward_img (torch CPU f32)"
        anchors, count = setup["anchors"].double().cpu().numpy(), setup["count"].cpu().numpy()
        sub = pts[:npts // 64].numpy()
        t0 = time.perf_counter()
        orc.tactile_assign_within(sub, anchors, count, setup["success"].cpu().numpy())
        cpu["tactile_assign_ms_per_lattice_scaled"] = 1e3 * (time.perf_counter() - t0) * 64
        cpu["tactile_assign_sample"] = "1/64 of the lattice, oracle.tactile_assign_within (numpy cdist rule, one thread)"
        res["cpu_baseline"] = cpu
    except Exception as e:                                                 # noqa: BLE001
        res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


def config5_section(dev, precision, nx=256):
    """BASELINE config 5 on ONE GPU -- the VTacOH route of generate_obj_mesh_wnf at 256^3, end to end (generation.py:161-200, 257,
    268-273): shape encoder -> hand encoder (plane PointNet + 2-D U-Net + MANO layer) -> fingertips -> Resnet18 tactile features ->
    vt_tactile_assign (nearest fingertip within 0.05) -> decode by finger id -> marching cubes.  (The sharded form of the same work
    is `sharded_scene`; the reference walks 168 chunks of 100 000 points through a dense [1, 256^3, 32] c_img_all = 2.1 GB.)"""
    import numpy as np
    import torch
    from vtaco_amd import ops
    from vtaco_amd.bench_util import build_tactile_scene
    from vtaco_amd.conv_onet.generation import Generator3D
    model, data, _ = build_tactile_scene(dev, "vtacoh", "simple_local")
    gen = Generator3D(model, device=dev, resolution0=nx // 4, padding=0.1, with_img=True, encode_t2d=False, decode_precision=precision)
    res, box = _route_stages(gen, model, data, nx)
    setup, inputs, imgs = box["setup"], data["inputs"].to(dev), data["inputs.img"].to(dev)
    with torch.no_grad():
        ids = ops.tactile_assign(setup["anchors"].to(dev), setup["success"].to(dev), "nearest", 0.05, lattice=(nx, 1.1, 0, nx ** 3),
                                 count=setup["count"].to(dev))
        feats, out = setup["feats"].to(dev).float(), torch.empty((1, nx ** 3), dtype=torch.float32, device=dev)
        grid = box["c"]["grid"]
        res["setup_stage_ms"] = {
            "hand_encoder_unet_mano": _median_ms(lambda: gen._replay("encode_hand_inputs", [inputs], model.encode_hand_inputs)),
            "resnet18_tactile_features": _median_ms(lambda: gen._replay("encode_img", [data["inputs.img"]], model.encode_img_inputs)),
            "note": "graph replays; the rest of tactile_setup is the host side (fingertips into the object frame: 5 x 3 numbers, numpy)"}
        res["lattice_stage_ms"] = {
            "vt_tactile_assign": _median_ms(lambda: ops.tactile_assign(setup["anchors"].to(dev), setup["success"].to(dev), "nearest", 0.05,
                                                                       lattice=(nx, 1.1, 0, nx ** 3), count=setup["count"].to(dev))),
            "decode_by_finger_id": _median_ms(lambda: model.decoder.decode_lattice_ids(grid, nx, ids, feats, out=out, precision=precision))}
    res["lattice_points_with_a_tactile_feature"] = int((ids != 255).sum())
    res["points_per_s_end_to_end"] = nx ** 3 / (res["end_to_end_ms"] * 1e-3)
    res["nx"], res["precision"] = nx, precision
    try:
        from oracle import vtaco_oracle as orc
        pts = (1.1 * orc.make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3))[:nx ** 3 // 64].numpy()
        tips = setup["anchors"][:, 0].double().cpu().numpy()
        t0 = time.perf_counter()
        orc.tactile_assign_nearest(pts, tips, setup["success"].cpu().numpy(), radius=0.05)
        res["cpu_baseline"] = {"tactile_assign_ms_per_lattice_scaled": 1e3 * (time.perf_counter() - t0) * 64, "cores": 1, "kind": "port",
                               "sample": "1/64 of the 256^3 lattice, oracle.tactile_assign_nearest (numpy cdist rule); decode / encode / marching cubes: "
                                         "the headline's cpu_baseline.stages_ms at 128^3"}
    except Exception as e:                                                 # noqa: BLE001
        res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


# ---------------------------------------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=128)
    ap.add_argument("--mode", choices=["visual", "img"], default="visual")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16x3", "f16f8"], default="f16x3",
                    help="arithmetic of the 16 dense layers: exact-f32 MFMA, split-bf16 / split-f16 MFMA (default: f32-level logits, "
                         "the fewest shader cycles), or f16 products with fp8 correction products (opt-in: relative error ~3e-5 |logit|)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: one scene per rank, no collective; strong: one scene, slab decode + one all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-only", action="store_true", help="only the decode metric (perf experiments)")
    ap.add_argument("--no-train", action="store_true", help="skip the config-4 training-step section")
    ap.add_argument("--no-configs35", action="store_true", help="skip the config-3 (full VTacO, 128^3) and config-5 (VTacOH route, 256^3) sections")
    ap.add_argument("--train-scenes", type=int, default=8, help="scenes per GPU of the training-step section (config 4: 8)")
    ap.add_argument("--train-steps", type=int, default=8, help="timed steps of the training-step section")
    ap.add_argument("--sharded-sizes", default="128,256", help="lattice sizes of the sharded-scene section (config 5: 128,256)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only: no kernel runs, value is null (CPU tests of --gpus N)")
    args = ap.parse_args()

    # ---- ranks ---------------------------------------------------------------------------------------------------
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            launch_ranks(args.gpus, sys.argv[1:])          # does not return
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    backend = os.environ.get("VTACO_BENCH_BACKEND", "nccl")
    dist = None
    if args.dry_run:
        dev = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        backend = "gloo"
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
        # one rank per GPU; VTACO_BENCH_BACKEND=gloo lets the multi-process path be dry-run on a box with fewer
        # GPUs than ranks (ranks then share devices; RCCL refuses that)
        dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        if world > 1:
            import torch.distributed as dist
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
    fx = Fences(dist, dev, backend)
    nx = args.nx
    npts = nx ** 3

    if args.dry_run:
        fx.fence()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        fx.fence()
        wall = fx.max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"metric": "occupancy query-points/sec at 128^3 (decode stage, lattice -> logits on device)",
                              "value": None, "unit": "query-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                              "dry_run": True, "wall_s": wall}))
        if dist is not None:
            dist.destroy_process_group()
        return

    from vtaco_amd import dist as vdist
    from vtaco_amd.bench_util import build_scene
    strong = args.scaling == "strong"
    scene = build_scene(0 if strong else rank, dev)
    model, grid = scene["model"], scene["grid"]
    dec = model.decoder
    c_img = scene["c_img"](nx) if args.mode == "img" else None
    out = torch.empty((1, npts), dtype=torch.float32, device=dev)
    if strong and world > 1:
        if args.mode == "img":
            raise SystemExit("bench.py: --scaling strong covers the visual decode")
        align = vdist.lattice_align(nx, world)
        first, count = vdist.slab_of(npts, rank, world, align)
        slab = torch.empty((1, max(count, 1)), dtype=torch.float32, device=dev)
        full = out.view(-1)

        def step():
            if count:
                dec.decode_lattice(grid, nx, box=1.1, first=first, count=count, out=slab[:, :count], precision=args.precision)
            vdist.all_gather_slabs(slab[0, :count], npts, None, align, out=full)
    else:
        def step():
            dec.decode_lattice(grid, nx, box=1.1, c_img=c_img, out=out, precision=args.precision)

    from vtaco_amd import ops as vops
    warm_steps = warm_up(step, args.warmup, fx, collective=strong and world > 1)
    wall, timed_steps, timing = timed_region(step, args.steps, fx, collective=strong and world > 1)
    kern_ms = timing["kernel_ms_total_events"]          # HIP events on the launch stream, over the whole timed region
    clock = clock_of_last_launch(vops)                  # stamps of the LAST launch: the clock the chip held, ramp and tail

    res = None
    if rank == 0:
        flop_pt = FLOP_PER_POINT_IMG if args.mode == "img" else FLOP_PER_POINT
        units = npts if strong else world * npts
        res = {
            "metric": "occupancy query-points/sec at 128^3 (decode stage, lattice -> logits on device)",
            "value": units * timed_steps / wall,
            "unit": "query-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": warm_steps,
            "ms_per_step": 1e3 * wall / timed_steps,
            "timing": dict(timing, note="the timed region is `reps` back-to-back repetitions of exactly `steps` steps between ONE pair "
                                        f"of barrier + synchronize fences (reps = what makes it last >= {MIN_TIMED_S} s); value and "
                                        "ms_per_step are over all of them (wall clock, max over ranks); kernel_ms_* are HIP-event times "
                                        "per repetition on the launch stream"),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"visual-only PointNet encoder + LocalDecoder, {nx}^3 lattice, "
                                   + ("ONE scene, slab per GPU + one all-gather of the logits, "
                                      if strong else "1 scene/GPU, no collective, ")
                                   + f"R=64 c_dim=32 hidden=32 n_blocks=5, mode={args.mode}, random-init weights "
                                   "(fc_1 re-randomised); f32 in / f32 out, dense layers on "
                                   + ({"bf16x3": "the bf16 matrix core with split-bf16 (hi+lo) operands and f32 accumulation",
                                       "f16x3": "the f16 matrix core with split-f16 (hi+lo) operands and f32 accumulation",
                                       "f16f8": "the f16 matrix core (hi parts) + one fp8 MFMA per layer for the two correction products, f32 accumulation",
                                       "f32": "the f32 matrix core"}[args.precision])
                                   + "; parity bar 1e-4 vs the f32 oracle",
                       "nx": nx, "points_per_step_per_gpu": npts if not strong else npts // world, "mode": args.mode,
                       "precision": args.precision},
            "per_gpu": units * timed_steps / wall / world,
        }
        if not strong or world == 1:
            res["roofline"] = roofline_of(args.precision, flop_pt, npts, kern_ms, clock)
        if world == 1 and args.precision in SPLIT:
            # the other kernels on the same inputs: exact f32 (what the training forward and precision="f32" run) and the
            # other split form
            def side(prec):
                fn = lambda: dec.decode_lattice(grid, nx, box=1.1, c_img=c_img, out=out, precision=prec)
                warm_up(fn, 5, fx)
                w, n, tm = timed_region(fn, args.steps, fx)
                return {"value": npts * n / w, "unit": "query-points/s",
                        "timing": {k: tm[k] for k in ("reps", "timed_steps", "kernel_ms_mean", "kernel_ms_std", "std_over_mean", "wall_over_events")},
                        "roofline": roofline_of(prec, flop_pt, npts, tm["kernel_ms_total_events"], clock_of_last_launch(vops))}
            res["exact_f32_kernel"] = side("f32")
            res["value_f32"] = res["exact_f32_kernel"]["value"]
            for other in ("f16x3", "f16f8", "bf16x3"):              # the other 16-bit forms: fp8-corrected (opt-in), round 1's split-bf16
                if other != args.precision:
                    key = ("f16_fp8_corrected" if other == "f16f8" else "split_" + other) + "_kernel"
                    res[key] = side(other)
                    res["value_" + other] = res[key]["value"]
            step()                                                   # leave the headline kernel's logits in `out`
        if world == 1 and not args.decode_only:
            res["mesh_extract"] = mesh_extract_stats(out.view(nx, nx, nx), nx)
            res["stages_ms"] = stage_times(scene, dec, grid, nx, out, dev, args.precision)
    # The headline is measured; the sections below (sharded scene, training step: the multi-rank ones run collectives) must not be
    # able to lose it: an exception is recorded in the line instead of ending the run, and a rank that is still waiting in a
    # collective after EXTRAS_LIMIT_S (a peer failed) prints what it has and leaves.
    print_lock = threading.Lock()
    printed = [False]

    def print_line_once():
        with print_lock:
            if rank == 0 and not printed[0]:
                printed[0] = True
                print(json.dumps(res), flush=True)

    def give_up():
        # a rank stuck in a collective (a peer died, RCCL deadlocked): the headline still goes out -- once -- and the process
        # leaves with a NON-ZERO status, so that the launcher and the harness see the failure and not only `extras_error`
        if rank == 0 and not printed[0]:
            res["extras_error"] = f"the sections after the headline did not finish within {EXTRAS_LIMIT_S} s"
        print_line_once()
        os._exit(3)
    watchdog = threading.Timer(EXTRAS_LIMIT_S, give_up)
    watchdog.daemon = True
    watchdog.start()
    if not args.decode_only and args.mode == "visual":
        try:
            if strong or world == 1:
                one_scene = scene
            else:
                one_scene = build_scene(0, dev)                          # every rank the SAME scene
            sizes = tuple(int(v) for v in args.sharded_sizes.split(",") if v)
            sh = sharded_scene(one_scene, dev, fx, rank, world, dist, args.precision, sizes=sizes)
            if rank == 0:
                res["sharded_scene"] = sh
            tr = None if args.no_train else train_step_section(dev, fx, rank, world, dist, steps=args.train_steps, scenes=args.train_scenes)
            if rank == 0 and tr is not None:
                res["train_step"] = tr
            if world == 1 and not args.no_configs35:
                # BASELINE configs 3 and 5 on one GPU (each in its own try: one failing must not lose the other)
                for key, fn in (("config3", config3_section), ("config5", config5_section)):
                    try:
                        t0 = time.perf_counter()
                        res[key] = fn(dev, args.precision)
                        res[key]["section_s"] = time.perf_counter() - t0
                    except Exception as e:                                   # noqa: BLE001 -- reported in the line
                        res[key] = {"error": f"{type(e).__name__}: {e}"[:400]}
                    torch.cuda.empty_cache()
        except Exception as e:                                           # noqa: BLE001 -- reported in the line
            if rank == 0:
                res["extras_error"] = f"{type(e).__name__}: {e}"[:400]
            else:
                sys.stderr.write(f"bench.py rank {rank}: {type(e).__name__}: {e}\n")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                scene["logits_cpu"] = out.view(nx, nx, nx).cpu().numpy()
                res["cpu_baseline"] = cpu_baseline(scene, nx)
            except Exception as e:                                       # noqa: BLE001
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        print_line_once()
    if dist is not None:
        try:
            dist.destroy_process_group()                                 # still under the watchdog: a failed peer may never arrive
        except Exception:                                                # noqa: BLE001 -- a peer that failed above cannot take the line back
            pass
    watchdog.cancel()


if __name__ == "__main__":
    main()
