import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, "f16f6")
pose = bench.build_poses(dev, body, bd, 512, 512, n_poses=2)[1]
stash = []
enc = net.encode_sparse_voxels
def spy(sp, save=None):
    v = enc(sp, save)
    stash.append([x.clone() for x in v])
    return v
net.encode_sparse_voxels = spy
with torch.no_grad():
    out = rend.render(pose)
    out2 = rend.render(pose)
    vols = enc(rend.prepare_sp_input(pose))
torch.cuda.synchronize()
print("render vs render again: rgb diff %.3e" % float((out["rgb_map"] - out2["rgb_map"]).abs().max()))
for li in range(4):
    print("level %d: render-internal vs later call %.3e" % (li, float((stash[0][li] - vols[li]).abs().max())))
sp = rend.prepare_sp_input(pose)
with torch.no_grad():
    m = net.render_rays(pose["ray_o"][0], pose["ray_d"][0], pose["near"][0], pose["far"][0], vols, sp, 64)
    m2 = net.render_rays(pose["ray_o"][0], pose["ray_d"][0], pose["near"][0], pose["far"][0], stash[0], sp, 64)
torch.cuda.synchronize()
print("manual march on later volumes vs render: %.3e ; on stashed volumes vs render: %.3e" % (
    float((m["rgb_map"] - out["rgb_map"][0]).abs().max()), float((m2["rgb_map"] - out["rgb_map"][0]).abs().max())))
