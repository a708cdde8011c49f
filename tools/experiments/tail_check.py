"""March time against the number of workgroups (512 slots = one generation): how much of a launch is its tail?  (round 4)
Views of 512 rows x W columns, W = 64 .. 576: 8 W workgroups of 64 rays each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuralbody_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for W in (64, 128, 256, 384, 448, 480, 512, 544, 576, 640, 1024):
    sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, W, 64, None)
    with torch.no_grad():
        fv = net.encode_sparse_voxels(rend.prepare_sp_input(bd))
        for _ in range(3):
            rend.render(bd, feature_volume=fv)
        torch.cuda.synchronize()
        ops.MARCH_EVENTS = []
        for _ in range(8):
            rend.render(bd, feature_volume=fv)
        torch.cuda.synchronize()
        ev, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
    ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
    wgs = n_rays // 64
    print("W %4d: %5d workgroups = %5.2f generations, march %.3f ms = %.3f ms per generation" % (W, wgs, wgs / 512, ms, ms / (wgs / 512)), flush=True)
