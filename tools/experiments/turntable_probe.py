import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from neuralbody_amd import novel_view as nv, ops
from tests import synthetic as syn
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
H = W = 512
sd, body, net, rend, bd, n_rays = bench.build_scene(dev, H, W, 64, None)
K, R, T = syn.full_coverage_camera(body, H, W)
train = []
for yaw in (0.0, 0.8, 1.6, 2.4):
    _, Rv, Tv = syn.full_coverage_camera(body, H, W, yaw=yaw)
    train.append(np.concatenate([np.concatenate([Rv, Tv.reshape(3, 1)], 1), [[0, 0, 0, 1.0]]], 0))
path = nv.gen_path(train, 6, center=body["world_verts"].mean(0).astype(np.float64))
frame = {k: v for k, v in bd.items() if k in ("coord", "out_sh", "bounds", "R", "Th", "latent_index")}
nvr = nv.NovelViewRenderer(rend, H, W, dev)
def tm(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for i, RT in enumerate(path):
    b, t_ray = tm(lambda: nvr.view_batch(K, RT, body["can_bounds"], frame))
    sp, t_sp = tm(lambda: rend.prepare_sp_input(b))
    vols, t_enc = tm(lambda: net.encode_sparse_voxels(sp))
    order, t_ord = tm(lambda: rend._tile_order(b, b["ray_o"].shape[1], 0, b["ray_o"].shape[1]))
    out, t_all = tm(lambda: rend.render(b))
    _, t_all2 = tm(lambda: rend.render(b))
    _, t_asm = tm(lambda: ops.image_assemble(b["mask_at_box"][0], out["rgb_map"][0].contiguous(), out["depth_map"][0].contiguous()))
    d = np.linalg.norm(np.linalg.inv(RT)[:3, 3] - body["world_verts"].mean(0))
    print("view %d: rays %d cam dist %.2f | raygen %.1f sp %.1f enc %.1f order %.1f render %.1f / again %.1f asm %.1f ms" %
          (i, b["ray_o"].shape[1], d, t_ray, t_sp, t_enc, t_ord, t_all, t_all2, t_asm), flush=True)
