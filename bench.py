#!/usr/bin/env python3
"""bench.py -- occupancy query-points/s at 128^3 on MI355X (BASELINE.json metric).

A "step" is one pass of the decode hot path (lattice -> logits grid on device:
in-kernel lattice generation, trilinear gather of the channels-last feature grid,
per-point conditioned ResNet MLP) over one synthetic scene; weights and feature
grid are resident in HBM when the timed region starts.
With --gpus N every rank decodes its own scene (the unit is the query point; no
data-path collective), so value = N * nx^3 * steps / max-over-ranks time ("weak").

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 31488        # 30 976 (16 linear layers) + 512 (8 corners x 32 ch FMA)
FLOP_PER_POINT_IMG = 33536    # with the tactile concat (forward_img)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
KERNEL_OF = {"f32": "decode_fwd_staged2_kernel<false>", "bf16x3": "decode_fwd_staged2_kernel<true>"}


def synthetic_scene(seed, device, R=64):
    """Seeded synthetic inputs: sphere point cloud -> encoder -> grid."""
    from vtaco_amd.bench_util import build_scene
    return build_scene(seed, device, R)


def cpu_baseline(sd, grid_cpu, nx, budget_s=15.0):
    """The oracle (a port, not the reference itself) timed on the host cores on a
    bounded sample: whole 100k-point chunks of the same lattice until ~budget_s."""
    from oracle import vtaco_oracle as orc
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
    return {"value": done / dt, "unit": "query-points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} of {nx ** 3} lattice points in 100k-point chunks, oracle/vtaco_oracle.py (torch CPU f32)"}


def mesh_extract_stats(vol, nx, runs=100):
    """mesh-extract latency (BASELINE.json metric, second half): HIP marching cubes on the
    device-resident logit grid -> device verts/faces, including the one host read of the
    counts that sizes the outputs.  p50 over `runs`."""
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
            "note": "latency-dominated (5 launches + 1 sync readback); 8.4 MB volume = 1.3 us at HBM rate"}


def stage_times(scene, dec, grid, nx, out, dev, precision):
    """End-to-end stages for one scene (reported beside the metric, not part of it)."""
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
    return res


PMC_SUMMARY = os.path.join("profiles", "r01i_pmc_summary.csv")


def pmc_counters(precision):
    """Per-launch counter sums of this precision's decode kernel from the committed PMC passes (or {})."""
    vals = {}
    try:
        for line in open(os.path.join(ROOT, PMC_SUMMARY)).read().splitlines()[1:]:
            k, c, n, mean = line.split(",")
            if k == "decode_" + precision:
                vals[c] = float(mean)
    except Exception:
        pass
    return vals


def measured_traffic(precision):
    """HBM-side bytes per decode launch from the committed PMC passes (profiles/, same command
    as this bench): (2 x FETCH_SIZE + WRITE_SIZE) KB -- the x2 is the guide's gfx950 correction
    for 16-B-per-lane reads; None if no profile is committed for this kernel."""
    try:
        vals = {}
        for line in open(os.path.join(ROOT, PMC_SUMMARY)).read().splitlines()[1:]:
            k, c, n, mean = line.split(",")
            if k == "decode_" + precision:
                vals[c] = float(mean)
        return (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def roofline_of(precision, flop_pt, npts, kern_ms):
    """Roofline object of one decode kernel: algorithmic FLOP / HIP-event time against the dense
    MFMA peak of the matrix-core input type it runs on."""
    achieved = flop_pt * npts / (kern_ms * 1e-3) / 1e12
    peak = PEAK_BF16_MFMA_TFLOPS if precision == "bf16x3" else PEAK_F32_MFMA_TFLOPS
    r = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
         "traffic": measured_traffic(precision),
         "traffic_note": "bytes/launch at the L2's memory side from rocprofv3 FETCH_SIZE/WRITE_SIZE passes "
                         f"({PMC_SUMMARY}); algorithmic = 33.5 MB grid + 8.4 MB logits",
         "kernel": KERNEL_OF[precision], "kernel_ms": kern_ms, "flop_per_point": flop_pt}
    pmc = pmc_counters(precision)
    if "GRBM_GUI_ACTIVE" in pmc and "SQ_VALU_MFMA_BUSY_CYCLES" in pmc:
        simd_cycles = 1024.0 * pmc["GRBM_GUI_ACTIVE"] / 8.0                 # 256 CUs x 4 SIMDs x cycles per launch (8 XCDs summed)
        r["pmc"] = {"matrix_pipe_busy": pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                    "valu_issue": 4.0 * pmc.get("SQ_INSTS_VALU", 0.0) / simd_cycles,
                    "source": PMC_SUMMARY + " (profiled launches of the same command)"}
    if precision == "bf16x3":
        r["note"] = ("split-bf16: each f32 product = 3 bf16 MFMA products (lo*hi + hi*lo + hi*hi), so the matrix pipe "
                     "executes ~3x the algorithmic FLOP; the kernel is bound by VALU issue (relu, hi/lo split, trilinear FMAs: "
                     "see `pmc`), not by the matrix pipe; DESIGN.md section 4")
        r["vs_f32_mfma_roofline"] = achieved / PEAK_F32_MFMA_TFLOPS     # SURVEY.md 8d's binding roofline for f32 results
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=128)
    ap.add_argument("--mode", choices=["visual", "img"], default="visual")
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default="bf16x3",
                    help="arithmetic of the 16 dense layers: exact-f32 MFMA or split-bf16 MFMA (both inside the 1e-4 bar)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-only", action="store_true", help="skip the mesh-extract / stage timings (perf experiments)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # one rank per GPU; VTACO_BENCH_BACKEND=gloo lets the multi-process path be dry-run on a box with fewer
    # GPUs than ranks (ranks then share devices; RCCL refuses that)
    backend = os.environ.get("VTACO_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from vtaco_amd import ops
    scene = synthetic_scene(rank, dev)
    model, grid = scene["model"], scene["grid"]
    nx = args.nx
    npts = nx ** 3
    dec = model.decoder
    c_img = scene["c_img"](nx) if args.mode == "img" else None
    out = torch.empty((1, npts), dtype=torch.float32, device=dev)

    def step():
        dec.decode_lattice(grid, nx, box=1.1, c_img=c_img, out=out, precision=args.precision)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # HIP events on the launch stream
    t = torch.tensor([wall], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())

    if rank == 0:
        flop_pt = FLOP_PER_POINT_IMG if args.mode == "img" else FLOP_PER_POINT
        res = {
            "metric": "occupancy query-points/sec at 128^3 (decode stage, lattice -> logits on device)",
            "value": world * npts * args.steps / wall,
            "unit": "query-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3" if args.precision == "bf16x3" else "f32", "data": "synthetic",
            "config": {"workload": f"visual-only PointNet encoder + LocalDecoder, {nx}^3 lattice, 1 scene/GPU, "
                                   f"R=64 c_dim=32 hidden=32 n_blocks=5, mode={args.mode}, random-init weights "
                                   "(fc_1 re-randomised); f32 in / f32 out, dense layers on "
                                   + ("the bf16 matrix core with split-bf16 (hi+lo) operands and f32 accumulation"
                                      if args.precision == "bf16x3" else "the f32 matrix core")
                                   + "; parity bar 1e-4 vs the f32 oracle",
                       "nx": nx, "points_per_step_per_gpu": npts, "mode": args.mode, "precision": args.precision},
            "per_gpu": npts * args.steps / wall,
            "roofline": roofline_of(args.precision, flop_pt, npts, kern_ms),
        }
        if world == 1 and args.precision == "bf16x3":
            # the exact-f32 kernel on the same inputs (what the training forward and precision="f32" run)
            for _ in range(5):
                dec.decode_lattice(grid, nx, box=1.1, c_img=c_img, out=out, precision="f32")
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(args.steps):
                dec.decode_lattice(grid, nx, box=1.1, c_img=c_img, out=out, precision="f32")
            ev1.record()
            torch.cuda.synchronize()
            ms32 = ev0.elapsed_time(ev1) / args.steps
            res["exact_f32_kernel"] = {"value": npts / (ms32 * 1e-3), "unit": "query-points/s",
                                       "roofline": roofline_of("f32", flop_pt, npts, ms32)}
            step()                                                   # leave the bf16x3 logits in `out`
        if world == 1 and not args.decode_only:
            res["mesh_extract"] = mesh_extract_stats(out.view(nx, nx, nx), nx)
            res["stages_ms"] = stage_times(scene, dec, grid, nx, out, dev, args.precision)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(scene["sd_decoder_cpu"], scene["grid_cpu"], nx)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
