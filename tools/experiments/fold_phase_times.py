"""Cycle budget of one depth step of the fc_0-folded march (experiment build -DFOLD_TIMING: wave 0 of the first 32 workgroups
stamps the cycle counter at its phase boundaries into the `raw` output).
    NB_EXTRA_FLAGS="-DFOLD_TIMING -include tools/experiments/fold_instrument.h" NB_LIB_SUFFIX=_timing python -m neuralbody_amd.build
    NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_timing.so python tools/experiments/fold_phase_times.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = ["barrier (Wt visible)", "bias + folded fc_0: U^T . Wt MFMAs (12 per 16 voxels)", "ring prime + publish fc_0", "MFMA fc_1 (96) + encodings",
         "publish fc_1", "MFMA fc_2 (96)", "next step: depth, grid coords, boxes", "publish fc_2", "alpha sums + next step: K list, grid lookups",
         "MFMA colour head (48)", "table store, barrier, encodings' operands, barrier", "MFMA view_fc over the encodings (24)", "rgb sums",
         "barrier", "next step: U DMA + trilinear weights", "heads + composite"]

dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, "f16f6")
with torch.no_grad():
    sp = rend.prepare_sp_input(bd)
    vols = net.encode_sparse_voxels(sp)
    order = rend._tile_order(bd, n, 0, n)
    for _ in range(2):
        out = net.render_rays(bd["ray_o"][0], bd["ray_d"][0], bd["near"][0], bd["far"][0], vols, sp, 64, want_raw=True, ray_order=order)
torch.cuda.synchronize()
t = out["raw"].view(torch.int32).reshape(-1)[:32 * 64 * 32].cpu().numpy().astype(np.int64).reshape(32, 64, 32)[:, :, :17]
d = np.diff(t, axis=2) & 0xffffffff
d = d[:, 2:-1].reshape(-1, 16)
step = (np.diff(t[:, :, 0], axis=1) & 0xffffffff)[:, 2:-1]
print("| phase | mean cycles | share |\n|---|---|---|")
tot = d.sum(1).mean()
for i, nm in enumerate(NAMES):
    print("| %s | %.0f | %.1f %% |" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / tot))
print("| stamped part of a depth step | %.0f | |" % tot)
print("| step to step | %.0f | |" % step.mean())
x = out["raw"].view(torch.int32).reshape(-1)[:32 * 64 * 32].cpu().numpy().astype(np.int64).reshape(32, 64, 32)[:, 2:-1]
sub = [("bias, K list header", 1, 17), ("wait for the first chunk's DMA", 17, 18), ("read its fragments", 18, 19), ("chunk loop", 19, 20), ("tail", 20, 2)]
ok = x[:, :, 21] > 0
print("inside the folded fc_0 (steps on the one-pass path, %.2f chunks on average):" % x[:, :, 21][ok].mean(),
      ", ".join("%s %.0f" % (nm, ((x[:, :, e] - x[:, :, b]) & 0xffffffff)[ok].mean()) for nm, b, e in sub))
mf = [1, 3, 5, 9, 11]
print("MFMA phases %.0f, publishes %.0f, next-step preparation %.0f, barriers (explicit) %.0f" % (
    d[:, mf].sum(1).mean(), d[:, [2, 4, 7]].sum(1).mean(), d[:, [6, 8, 14]].sum(1).mean(), d[:, [0, 13]].sum(1).mean()))
