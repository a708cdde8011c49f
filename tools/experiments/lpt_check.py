"""Does the order in which the 8 x 8 pixel tiles are handed to the dispatcher matter?  (round 4)
Tiles sorted by the opacity a first render found in them (dense tiles first: long voxel lists, dear workgroups; background last:
a short tail), the reverse, and a random order against the row-major default: march time of the 512 x 512 x 64 bench view."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuralbody_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, None)
pose = bench.build_poses(dev, body, bd, 512, 512)[1]
with torch.no_grad():
    fv = net.encode_sparse_voxels(rend.prepare_sp_input(pose))
    out = rend.render(pose, feature_volume=fv)
    ref = out["rgb_map"].clone()
    slots = rend._tile_order(pose, n_rays, 0, n_rays).clone()  # [tiles * 64]
    groups = slots.view(-1, 64)
    acc = out["acc_map"][0]
    cost = acc[groups.clamp_min(0).long()].mean(1)  # full-coverage view: every slot is a ray
    orders = {"row-major (default)": torch.arange(groups.shape[0], device=dev),
              "dense tiles first": torch.argsort(cost, descending=True),
              "dense tiles last": torch.argsort(cost),
              "random": torch.randperm(groups.shape[0], device=dev)}
    key = next(iter(rend._order_full))
    for rep in range(2):
        for name, perm in orders.items():
            rend._order_full[key] = groups[perm].reshape(-1).contiguous()
            for _ in range(3):
                o = rend.render(pose, feature_volume=fv)
            torch.cuda.synchronize()
            ops.MARCH_EVENTS = []
            for _ in range(10):
                o = rend.render(pose, feature_volume=fv)
            torch.cuda.synchronize()
            ev, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
            ms = sorted(a.elapsed_time(b) for a, b in ev)
            print("%-22s march %.3f ms (min %.3f)  max |rgb - default| %.1e" % (name, ms[len(ms) // 2], ms[0], float((o["rgb_map"] - ref).abs().max())), flush=True)
