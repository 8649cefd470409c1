import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
from conftest import load_golden
from vtaco_amd.conv_onet.generation import Generator3D
from vtaco_amd.conv_onet.models import ConvolutionalOccupancyNetwork, decoder_dict
from vtaco_amd.encoder import encoder_dict
from vtaco_amd import ops
DEV="cuda:0"
a, sd_e = load_golden("g3_pointnet.npz"); _, sd_d = load_golden("g1_decode.npz")
dec = decoder_dict['simple_local'](dim=3, c_dim=32, hidden_size=32, with_contact=True); dec.load_state_dict(sd_d)
enc = encoder_dict['pointnet_local_pool'](c_dim=32, dim=3, hidden_dim=32, unet3d=False, grid_resolution=16, plane_type='grid'); enc.load_state_dict(sd_e)
model = ConvolutionalOccupancyNetwork(dec, enc, device=DEV)
gen = Generator3D(model, device=DEV, resolution0=8, padding=0.1)
T=torch.from_numpy
for b in (0,1,0):
    p = T(a["p"])[b:b+1]
    with torch.no_grad():
        c = model.encode_inputs(p.to(DEV)); vol_e = gen.eval_lattice(c, 32).reshape(32,32,32).clone()
    eager = gen.generate_obj_mesh_wnf({"inputs": p})
    graph, static_in, vol, ws = gen._scene_graph(p.shape, 32)
    static_in.copy_(p.to(DEV)); graph.replay(); torch.cuda.synchronize()
    print("after replay: vol finite", bool(torch.isfinite(vol).all()), "vol==eager", torch.equal(vol, vol_e), float((vol - vol_e).abs().max()),
          "hdr", ws[:24].view(torch.int32).tolist(), "static_in==p", torch.equal(static_in.cpu(), p))
    try:
        fast = gen.generate_mesh_graphed(p)
    except Exception as e:
        print("graphed failed:", e); continue
    print(b, "vol equal", torch.equal(vol, vol_e), float((vol-vol_e).abs().max()), "V", eager.vertices.shape, fast.vertices.shape,
          "faces eq", eager.faces.shape == fast.faces.shape and torch.equal(eager.faces, fast.faces),
          "verts eq", eager.vertices.shape == fast.vertices.shape and torch.equal(eager.vertices, fast.vertices))
