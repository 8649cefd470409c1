"""Two ranks on ONE GPU (gloo carries the HIP tensors through the host): the multi-process code path of the sharded
mesh generation and of a data-parallel training step, with real kernels.  RCCL itself needs one GPU per rank and is
exercised by the driver's scaling bench only.

The worker processes must be started by a parent that has NOT initialised the GPU (a process that has may not be the
origin of an exec on the GPU boxes), hence the file name -- pytest collects it first -- and the skip below."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_golden
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
    from vtaco_amd.conv_onet.training import Trainer
    from vtaco_amd.dist import GradAllReduce
    from vtaco_amd.encoder import encoder_dict
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = "cuda:0"
        a, sd_e = load_golden("g3_pointnet.npz")
        _, sd_d = load_golden("g1_decode.npz")
        dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, with_contact=True)
        dec.load_state_dict(sd_d, strict=True)
        enc = encoder_dict['pointnet_local_pool'](c_dim=32, dim=3, hidden_dim=32, grid_resolution=16, plane_type='grid')
        enc.load_state_dict(sd_e, strict=True)
        model = ConvolutionalOccupancyNetwork(dec, enc, device=dev)
        gen = Generator3D(model, device=dev, resolution0=8, padding=0.1)
        p = torch.from_numpy(a["p"])[:1]
        plain = gen.generate_obj_mesh_wnf({"inputs": p})
        shard = gen.generate_obj_mesh_sharded({"inputs": p})
        ok_mesh = torch.equal(plain.faces, shard.faces) and torch.equal(plain.vertices, shard.vertices)
        why = [] if ok_mesh else ["visual: sharded != plain"]
        # the range guard of the (default) half-precision decode is rank-local outside the sharded entry point: a mesh that only
        # rank 0 exports -- during distributed training, or with uneven scene counts -- must meet no collective ...
        assert gen.decode_precision == "f16x3"
        if rank == 0:
            solo = gen.generate_obj_mesh_wnf({"inputs": p})
            ok_mesh = ok_mesh and torch.equal(solo.faces, plain.faces) and torch.equal(solo.vertices, plain.vertices)
        # ... and sharded generation on a SUB-group reduces its flag over that group only (here: rank 0 alone)
        g0 = dist.new_group([0])
        if rank == 0:
            sub = gen.generate_obj_mesh_sharded({"inputs": p}, group=g0)
            ok_mesh = ok_mesh and torch.equal(sub.faces, plain.faces) and torch.equal(sub.vertices, plain.vertices)
        # tactile (VTacO t2d) generation sharded: the contact clouds are drawn with numpy's generator -- rank 0's draw is
        # broadcast, so ranks seeded differently still assemble rank 0's mesh
        import numpy as np
        from conftest import GOLDEN
        z = np.load(os.path.join(GOLDEN, "g12_t2d.npz"))
        torch.manual_seed(0)
        img = encoder_dict["UNet"](num_classes=1, in_channels=3, depth=2, start_filts=8)
        model_t = ConvolutionalOccupancyNetwork(dec, enc, None, img, None, device=dev)
        gen_t = Generator3D(model_t, device=dev, resolution0=8, padding=0.1, with_img=True, encode_t2d=True,
                            depth_origin=z["depth_origin"])
        gi = torch.Generator().manual_seed(1)
        data_t = {"inputs": p, "inputs.img": torch.rand(1, 5, 3, 8, 4, generator=gi), "inputs.depth": torch.from_numpy(z["depths"])[None],
                  "inputs.touch_success": torch.from_numpy(z["touch"]), "inputs.pc_ply": torch.from_numpy(z["pc_ply"]),
                  "points.cam_pos": torch.from_numpy(z["cam_pos"]), "points.cam_rot": torch.from_numpy(z["cam_rot"])}
        np.random.seed(7)
        plain_t = gen_t.generate_obj_mesh_wnf(data_t)
        np.random.seed(7 if rank == 0 else 99)
        shard_t = gen_t.generate_obj_mesh_sharded(data_t)
        # rank 0's clouds AND tactile features are broadcast: the sharded mesh is rank 0's single-process mesh bit for bit.  Another
        # rank's own single-process mesh may differ in the last bits (its features come out of MIOpen convolutions whose algorithm
        # is chosen per process by timing): there the comparison allows 1e-5 on the vertices
        if rank == 0:
            ok_t = torch.equal(plain_t.faces, shard_t.faces) and torch.equal(plain_t.vertices, shard_t.vertices)
        else:
            ok_t = (plain_t.faces.shape == shard_t.faces.shape and plain_t.vertices.shape == shard_t.vertices.shape
                    and float((plain_t.vertices - shard_t.vertices).abs().max()) <= 1e-5)
        meshes = [None, None]
        dist.all_gather_object(meshes, (shard_t.vertices.cpu(), shard_t.faces.cpu()))
        ok_t = ok_t and torch.equal(meshes[0][0], meshes[1][0]) and torch.equal(meshes[0][1], meshes[1][1])   # every rank holds the SAME mesh
        if not ok_t:
            why.append(f"tactile: sharded != plain ({plain_t.vertices.shape[0]} vs {shard_t.vertices.shape[0]} vertices)")
        ok_mesh = ok_mesh and ok_t
        # data-parallel step: different batches per rank, one flat all-reduce -> identical parameters afterwards
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        trainer = Trainer(model, opt, device=dev, grad_sync=GradAllReduce(model.parameters()))
        g = torch.Generator().manual_seed(100 + rank)
        batch = {"inputs": torch.from_numpy(a["p"])[rank:rank + 1], "points": (torch.rand(1, 512, 3, generator=g) - 0.5),
                 "points.occ": torch.rand(1, 512, generator=g)}
        trainer.train_step(batch)
        flat = torch.cat([q.detach().reshape(-1) for q in model.parameters()]).cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ok_sync = all(torch.equal(gathered[0], t) for t in gathered[1:])
        if why:
            sys.stderr.write(f"rank {rank}: {why}; lattice precisions now {gen.decode_precision!r} / {gen_t.decode_precision!r}\n")
        q.put((rank, bool(ok_mesh), bool(ok_sync)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu_gloo():
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run this file on its own (or first)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]


def test_bench_gpus_2_starts_two_ranks_and_strong_equals_the_lattice():
    """``python bench.py --gpus 2`` with no launcher around it: the parent starts two ranks (here sharing the one GPU over gloo),
    rank 0 prints ONE JSON line with n_gpus == 2; strong scaling = one scene, slab decode + one all-gather."""
    import json
    import subprocess
    import sys
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run this file on its own (or first)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VTACO_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    for scaling in ("strong", "weak"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                              "--scaling", scaling, "--decode-only"], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        res = json.loads(lines[0])
        assert res["n_gpus"] == 2 and res["scaling"] == scaling and res["steps"] == 3 and res["value"] > 1e8
        assert res["config"]["points_per_step_per_gpu"] == (128 ** 3 // 2 if scaling == "strong" else 128 ** 3)


def test_bench_gpus_2_runs_the_multi_rank_sections():
    """Pre-flight of the driver's scaling run: ``bench.py --gpus 2`` WITHOUT --decode-only (two ranks sharing the one GPU over gloo,
    small knobs) executes the sections that only exist with world > 1 -- the sharded scene's all-gather with every rank's stage
    times, the training step's bucketed gradient all-reduce, its share -- and leaves no ``extras_error``."""
    import json
    import subprocess
    import sys
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run this file on its own (or first)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VTACO_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--train-scenes", "1", "--train-steps", "1", "--sharded-sizes", "128"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and "extras_error" not in res, res.get("extras_error")
    sh = res["sharded_scene"]["128"]
    assert sh["slab_points_per_rank"] == 128 ** 3 // 2 and sh["verts"] > 0
    assert len(sh["per_rank_stage_ms"]) == 2 and all(r["decode_slab"] > 0 and r["all_gather"] > 0 for r in sh["per_rank_stage_ms"])
    tr = res["train_step"]
    assert tr["global_batch"] == 2 and tr["t2d_pretrained"] is True
    assert tr["buckets"]["buckets"] >= 2 and tr["buckets"]["launched_in_backward"] + tr["buckets"]["launched_at_sync"] > 0
    assert 0.0 <= tr["allreduce_share"] <= 1.0 and tr["ms_per_step_without_allreduce"] > 0 and tr["allreduce_alone_ms"] > 0
