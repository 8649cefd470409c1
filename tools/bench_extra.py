"""Secondary workloads of BASELINE.json (configs 3-5) on ONE MI355X; prints one JSON line each.
Not the headline metric (bench.py is); numbers are quoted in DESIGN.md.
  fusion : AttentionDecoder.forward_img (TransformerFusion) over the 128^3 lattice in chunks of 2048, EVERY chunk through the three
           attention units (the kernels' own number: what tools/pmc_fusion.sh profiles)
  fusion_product : the same lattice through Generator3D (finger ids; chunks without tactile features skip the fuser), and with the
           skip switched off
  img    : LocalDecoder.forward_img (tactile concat, the shipped VTacO config) 128^3
  train  : one training step fwd+bwd+Adam, 8 scenes x 2048 points per GPU (config 4's per-GPU share)
  dense256 : 256^3 decode + marching cubes (config 5 on one GPU)
  hand   : the hand encoder (plane PointNet + 2-D U-Net + MANO layer) and the MANO kernel alone
  wide   : LocalDecoder at the reference's class-default widths (hidden 256, c_dim 128; vt_decode_fwd_wide) over the 128^3 lattice
"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtaco_amd import ops
from vtaco_amd.bench_util import build_scene, randomise_fc1, sphere_cloud
from vtaco_amd.conv_onet.models import decoder_dict

dev = torch.device("cuda:0")
SECTIONS = set(sys.argv[1:]) or {"img", "fusion", "fusion_product", "train", "dense256", "hand", "wide"}
# MIOpen only picks its fast f32 conv3d kernels for channels_last_3d tensors in find mode, and only if
# the flag is set before the first convolution of the process (26 ms vs 382 ms fwd+bwd at B=2)
torch.backends.cudnn.benchmark = True


def timed(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


sc = build_scene(0, dev)
model, grid = sc["model"], sc["grid"]
dec = model.decoder
nx = 128
out = {}

# --- img: tactile concat over the lattice
c_img = sc["c_img"](nx)
if "img" in SECTIONS:
    for prec in ("f32", "bf16x3", "f16x3", "f16f8"):
        t = timed(lambda: dec.decode_lattice(grid, nx, c_img=c_img, precision=prec), 50, 5)
        print(json.dumps({"workload": f"forward_img (tactile concat) 128^3 lattice, {prec}", "ms": t * 1e3,
                          "points_per_s": nx ** 3 / t, "tflops": 33536 * nx ** 3 / t / 1e12}))

# --- wide: the general-shape decoder (weights streamed from L2): exact f32 and split-f16
if "wide" in SECTIONS:
    for hidden, cd in ((256, 128), (64, 32)):
        torch.manual_seed(1)
        wdec = decoder_dict['simple_local'](dim=3, c_dim=cd, hidden_size=hidden, n_blocks=5).to(dev).eval()
        randomise_fc1(wdec, 4)
        wgrid = torch.randn(1, cd, 64, 64, 64, device=dev)
        flop = 2 * (3 * hidden + 5 * (cd + 2 * hidden) * hidden + hidden)
        with torch.no_grad():
            exact = wdec.decode_lattice(wgrid, nx, precision="f32")
            for prec, kernel, peak in (("f32", "vt_decode_fwd_wide, exact f32", 157.3), ("f16x3", "vt_decode_fwd_wide_f16x3, split f16", 2500.0)):
                t = timed(lambda: wdec.decode_lattice(wgrid, nx, precision=prec), 5, 1)
                err = float((wdec.decode_lattice(wgrid, nx, precision=prec) - exact).abs().max())
                roof = {"bound": "mfma", "achieved": flop * nx ** 3 / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                        "frac": flop * nx ** 3 / t / 1e12 / peak, "traffic": None}
                # counters of tools/pmc_wide.sh (profiles/r06_pmc_wide_summary.csv; dropped when collected on other sources): HBM-side bytes
                # per launch ((2 x FETCH_SIZE + WRITE_SIZE) KB, the guide's gfx950 correction), matrix pipe busy, L2 requests
                import bench as _bench
                pmc = _bench.pmc_wide_table().get(f"wide_{prec}_{hidden}", {})
                if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                    roof["traffic"] = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
                    roof["algorithmic_bytes"] = 4.0 * (wgrid.numel() + nx ** 3)
                if "GRBM_GUI_ACTIVE" in pmc and "SQ_VALU_MFMA_BUSY_CYCLES" in pmc:
                    simd_cycles = 1024.0 * pmc["GRBM_GUI_ACTIVE"] / 8.0
                    roof["pmc"] = {"matrix_pipe_busy": pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                                   "valu_issue": 4.0 * pmc.get("SQ_INSTS_VALU", 0.0) / simd_cycles,
                                   "l2_requests": pmc.get("TCC_REQ_sum"), "l2_hit_rate": (pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
                                                                                        if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc else None),
                                   "source": _bench.PMC_WIDE_SUMMARY}
                print(json.dumps({"workload": f"LocalDecoder hidden {hidden} / c_dim {cd} / 5 blocks, 128^3 lattice ({kernel})",
                                  "ms": t * 1e3, "points_per_s": nx ** 3 / t, "tflops": flop * nx ** 3 / t / 1e12,
                                  "max_abs_vs_exact_f32": err, "logit_absmax": float(exact.abs().max()), "roofline": roof}))

# --- fusion: attention decoder, chunks of 2048 points as a batch of 1024 "scenes" sharing one grid
torch.manual_seed(0)
adec = decoder_dict['attention_local'](dim=3, c_dim=32, hidden_size=32).to(dev).eval()
randomise_fc1(adec, 3)
N, chunks = 2048, nx ** 3 // 2048
from vtaco_amd.common import make_3d_grid
pts = (1.1 * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).to(dev)
ci = c_img.reshape(chunks, N, 32)
pc = pts.reshape(chunks, N, 3)


def fusion_pass(cb=256):
    outs = []
    with torch.no_grad():
        for lo in range(0, chunks, cb):
            p = pc[lo:lo + cb]
            c = ops.sample_grid(grid, None, lattice=(nx, 1.1, lo * N, p.shape[0] * N)).reshape(-1, N, 32)   # the lattice range: staged gather
            f = adec.fuser(ci[lo:lo + cb], 1, c, 1)
            outs.append(adec._mlp_fwd(f, p))
    return outs


if "fusion" in SECTIONS:
    t = timed(fusion_pass, 3, 1)
    flop_pt = 30976 + 576 * N + 61440
    tf = flop_pt * nx ** 3 / t / 1e12
    print(json.dumps({"workload": "AttentionDecoder.forward_img 128^3, chunk N=2048 (1024 chunks)", "ms": t * 1e3,
                      "points_per_s": nx ** 3 / t, "tflops": tf,
                      "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0, "traffic": None,
                                   "flop_per_point": flop_pt,
                                   "note": "algorithmic FLOP (SURVEY.md 8d: 30 976 + 576 N + 61 440 per point at chunk size N = 2048) over the "
                                           "wall time of the whole pass (sample, 3 attention units, MLP) against the dense 16-bit MFMA peak; "
                                           "the kernels recompute every N x N score tile three times (row sums, column sums, attend) with "
                                           "split operands, so the matrix pipe executes several times the algorithmic FLOP: per-kernel time "
                                           "and counters in profiles/r03*_fusion_*"}}))
if "fusion_product" in SECTIONS:
    # the same lattice through the product path: Generator3D (decoder: attention_local, points_batch_size 2048) assigning finger
    # ids to the lattice and decoding it chunk by chunk (whole chunks batched per call; chunks no finger touches skip the fuser)
    from vtaco_amd.conv_onet.generation import Generator3D
    from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork
    agen = Generator3D(ConvolutionalOccupancyNetwork(adec, None, device=dev), device=dev, resolution0=nx // 4, padding=0.1,
                       points_batch_size=N, with_img=True)
    gs = torch.Generator().manual_seed(2)
    tips = torch.randn(5, 1, 3, generator=gs)
    setup = {'feats': torch.randn(5, 32, generator=gs), 'anchors': 0.3 * tips / tips.norm(dim=-1, keepdim=True),
             'success': torch.tensor([1, 0, 1, 1, 1], dtype=torch.uint8), 'mode': 'nearest', 'radius': 0.08,
             'count': torch.ones(5, dtype=torch.int32)}
    cg = {"grid": grid}
    with torch.no_grad():
        ids = ops.tactile_assign(setup['anchors'].to(dev), setup['success'].to(dev), 'nearest', 0.08, lattice=(nx, 1.1, 0, nx ** 3))
        touched = int((ids[0] != 255).reshape(-1, N).any(dim=1).sum())
        for skip in (True, False):
            agen.skip_untouched_chunks = skip
            t = timed(lambda: agen._eval_lattice_tactile(cg, nx, setup), 3, 1)
            print(json.dumps({"workload": "Generator3D._eval_lattice_tactile, decoder attention_local, 128^3, points_batch_size 2048 (finger ids + "
                                          "batched chunks; four fingertips, radius 0.08), " +
                                          ("chunks without tactile features skip the fuser" if skip else "every chunk through the fuser"),
                              "ms": t * 1e3, "points_per_s": nx ** 3 / t, "chunks": chunks, "chunks_with_tactile_features": touched}))

# --- train: 8 scenes/GPU, N=2048, fwd+bwd+Adam through encoder (PointNet + UNet3D host path) and decoder
B = 8
model.train()
cloud = torch.cat([sphere_cloud(i) for i in range(B)]).to(dev)
g = torch.Generator().manual_seed(1)
pq = ((torch.rand(B, 2048, 3, generator=g) - 0.5) * 1.1).to(dev)
occ = torch.rand(B, 2048, generator=g).to(dev)
cimg_t = (torch.randn(B, 2048, 32, generator=g) * (torch.rand(B, 2048, 1, generator=g) < 0.2)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)


def train_step(with_encoder=True):
    opt.zero_grad(set_to_none=True)
    if with_encoder:
        c = model.encode_inputs(cloud)
    else:
        c = {"grid": grid.expand(B, -1, -1, -1, -1).contiguous(memory_format=torch.channels_last_3d)}
    logits = model.decode_img(pq, c, cimg_t).logits
    loss = torch.nn.functional.l1_loss(logits, occ)
    loss.backward()
    opt.step()


if "train" in SECTIONS:
    t_dec = timed(lambda: train_step(False), 10, 2)
    print(json.dumps({"workload": "train step, decoder only (fwd+bwd+Adam), 8 scenes x 2048 pts", "ms": t_dec * 1e3,
                      "scenes_per_s": B / t_dec}))
    t0 = time.perf_counter()
    train_step(True)                     # first call pays MIOpen's kernel search
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    t_all = timed(lambda: train_step(True), 5, 1)
    print(json.dumps({"workload": "train step incl. PointNet+UNet3D, 8 scenes x 2048 pts; UNet3D under autograd = "
                                  + model.encoder.train_unet3d + (" (convs " + model.encoder.unet3d.train_precision + ")"
                                                                   if model.encoder.train_unet3d == "hip" else " (MIOpen find mode)"),
                      "ms": t_all * 1e3, "scenes_per_s": B / t_all, "first_call_s": t_first}))
model.eval()

# --- dense256: config 5 on one GPU
if "dense256" in SECTIONS:
    nx2 = 256
    buf = torch.empty((1, nx2 ** 3), dtype=torch.float32, device=dev)
    for prec in ("f32", "bf16x3", "f16x3", "f16f8"):
        t = timed(lambda: dec.decode_lattice(grid, nx2, out=buf, precision=prec), 10, 2)
        t_mc = timed(lambda: ops.marching_cubes(buf.view(nx2, nx2, nx2), None, rescale=(nx2 / 2, 1.1 / nx2)), 10, 2)
        v, f, _ = ops.marching_cubes(buf.view(nx2, nx2, nx2), None)
        print(json.dumps({"workload": f"256^3 decode + marching cubes, {prec}", "decode_ms": t * 1e3,
                          "points_per_s": nx2 ** 3 / t, "mc_ms": t_mc * 1e3, "verts": v.shape[0], "faces": f.shape[0]}))

# --- hand: the hand encoder as configs/VTacO/VTacO_YCB.yaml:33-56 builds it, on a synthetic MANO-format asset
if "hand" in SECTIONS:
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import synth_mano
    from vtaco_amd.encoder import encoder_dict
    root = tempfile.mkdtemp(prefix="vt_mano_")
    synth_mano.write_pkl(synth_mano.make_asset(0), root)
    torch.manual_seed(0)
    henc = encoder_dict["pointnet_local_pool"](
        dim=3, c_dim=32, padding=0.1, hidden_dim=32, plane_type=["xz", "xy", "yz"], plane_resolution=32, unet=True,
        unet_kwargs=dict(depth=4, merge_mode="concat", start_filts=32), out_mano=True, out_dim=51,
        manolayer_kwargs=dict(center_idx=9, flat_hand_mean=False, ncomps=45, side="right", mano_root=root, use_pca=False,
                              root_rot_mode="axisang", joint_rot_mode="axisang", robust_rot=False, return_transf=False)
    ).to(dev).eval()
    hcloud = sphere_cloud(0).to(dev)
    with torch.no_grad():
        t_enc = timed(lambda: henc(hcloud), 50, 5)
        pis = [ops.PlaneIndex(hcloud, 32, 0.1, k) for k in ("xz", "xy", "yz")]
        t_idx = timed(lambda: [ops.PlaneIndex(hcloud, 32, 0.1, k) for k in ("xz", "xy", "yz")], 50, 5)
        feat = torch.randn(1, 3000, 32, device=dev)
        t_pool = timed(lambda: [ops.voxel_pool_max_fwd(feat, pi, want_argmax=False) for pi in pis], 50, 5)
        for Bh in (1, 64, 1024):
            pose = torch.randn(Bh, 48, device=dev) * 0.5
            t_m = timed(lambda: henc.mano_layer(pose), 100, 10)
            print(json.dumps({"workload": f"MANO layer (vt_mano_fwd), {Bh} hands", "ms": t_m * 1e3, "hands_per_s": Bh / t_m,
                              "gflops": 0.95e-3 * Bh / t_m}))
    print(json.dumps({"workload": "hand encoder: 3000 pts -> 3 planes @32^2 -> U-Net (depth 4, 32 filters, vt_plane_unet_fwd: 18 launches) -> MANO",
                      "ms": t_enc * 1e3, "plane_build_x3_ms": t_idx * 1e3, "pool_max_x3_ms": t_pool * 1e3}))
