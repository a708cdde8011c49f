"""Cycle budget of one depth step of the M-split f16f6 march (experiment build -DMS6_TIMING: wave 0 of the first 32 workgroups
stamps the cycle counter at its phase boundaries into the `raw` output).
NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_ms6TIMING.so python tools/experiments/ms6_phase_times.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = ["gather L3 + convert + ring prime", "barrier", "MFMA fc_0 K phase A (48)", "barrier", "gather L2 + convert + prime", "barrier",
         "MFMA phase B (48)", "barrier", "gather L0, L1 + convert + prime", "barrier", "MFMA phase C (48)",
         "publish fc_0 (barrier, convert, barrier)", "MFMA fc_1 (96)", "publish fc_1", "MFMA fc_2 (96)", "publish fc_2",
         "alpha partial sums", "MFMA folded view layer (48)", "barrier", "encodings + convert", "barrier", "MFMA view_fc over PE (24)",
         "rgb partial sums", "barrier", "heads + composite"]

dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, "f16f6")
with torch.no_grad():
    sp = rend.prepare_sp_input(bd)
    vols = net.encode_sparse_voxels(sp)
    order = rend._tile_order(bd, n, 0, n)
    for _ in range(2):
        out = net.render_rays(bd["ray_o"][0], bd["ray_d"][0], bd["near"][0], bd["far"][0], vols, sp, 64, want_raw=True, ray_order=order)
torch.cuda.synchronize()
t = out["raw"].view(torch.int32).reshape(-1)[:32 * 64 * 32].cpu().numpy().astype(np.int64).reshape(32, 64, 32)[:, :, :26]
d = np.diff(t, axis=2) & 0xffffffff
d = d[:, 2:-1].reshape(-1, 25)
step = (np.diff(t[:, :, 0], axis=1) & 0xffffffff)[:, 2:-1]
print("| phase | mean cycles | share |\n|---|---|---|")
tot = d.sum(1).mean()
for i, nm in enumerate(NAMES):
    print("| %s | %.0f | %.1f %% |" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / tot))
print("| stamped part of a depth step | %.0f | |" % tot)
print("| step to step | %.0f | |" % step.mean())
x = out["raw"].view(torch.int32).reshape(-1)[:32 * 64 * 32].cpu().numpy().astype(np.int64).reshape(32, 64, 32)[:, 2:-1]
if os.environ.get("MS6_SUB", "A") == "B":  # library built with -DMS6_TIMING=2
    sub = [("ray record + next z", 19, 26), ("grid coordinates", 26, 27), ("wave box (DPP reductions)", 27, 28), ("voxel box", 28, 29),
           ("issue the level-3 DMA", 29, 30), ("ring prime", 30, 31), ("store the encodings' operands", 31, 20)]
    print("inside 'encodings + convert' (= preparation of the next depth step):",
          ", ".join("%s %.0f" % (nm, ((x[:, :, e] - x[:, :, b]) & 0xffffffff).mean()) for nm, b, e in sub))
    sys.exit(0)
sub = [("tile wait", 4, 26), ("blend level 2", 26, 27), ("boxes of levels 0, 1", 27, 28), ("convert", 28, 29), ("issue the DMAs", 29, 30),
       ("store operands", 30, 31), ("ring prime", 31, 5)]
print("inside 'gather L2 + convert + prime':", ", ".join("%s %.0f" % (nm, ((x[:, :, e] - x[:, :, b]) & 0xffffffff).mean()) for nm, b, e in sub))
mf = [2, 6, 10, 12, 14, 17, 21]
print("MFMA phases %.0f, barriers %.0f, gather %.0f, publish %.0f" % (d[:, mf].sum(1).mean(), d[:, [1, 3, 5, 7, 9, 18, 20, 23]].sum(1).mean(),
                                                                       d[:, [0, 4, 8]].sum(1).mean(), d[:, [11, 13, 15]].sum(1).mean()))
