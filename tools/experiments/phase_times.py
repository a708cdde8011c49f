"""Cycle budget of one depth step of the f16f8 march (experiment build -DF_TIMING: wave 0 of workgroup 0 stamps s_memtime at
its phase boundaries into the `raw` output).   NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_f16TIMING.so python tools/experiments/phase_times.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = {1: "grid coords + wave box", 2: "issue the tile fetches of 4 levels", 3: "level 0: tile, blend, convert", 4: "level 1",
         5: "level 2", 6: "level 3", 13: "fc_0 (184 rec) + tail conversion", 14: "fc_1 (128 rec) + tail conversion",
         15: "fc_2 (128 rec) + alpha + tail", 16: "merged layer (64 rec) + tail", 17: "view_fc over g (64 rec)",
         18: "xyz encodings + conversion", 19: "view_fc over PE (28 rec) + rgb", 20: "composite, weight store, loop"}
STAMPS = sorted(NAMES)

dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, "f16f8")
with torch.no_grad():
    sp = rend.prepare_sp_input(bd)
    vols = net.encode_sparse_voxels(sp)
    order = rend._tile_order(bd, n, 0, n)
    for _ in range(2):
        out = net.render_rays(bd["ray_o"][0], bd["ray_d"][0], bd["near"][0], bd["far"][0], vols, sp, 64, want_raw=True, ray_order=order)
torch.cuda.synchronize()
t = out["raw"].view(torch.int32).reshape(-1)[:64 * 32].cpu().numpy().astype(np.int64).reshape(64, 32)[:, :21]
t = t[:, [0] + STAMPS]
d = np.diff(t, axis=1) & 0xffffffff
d = d[2:-1]  # skip the first steps (cold caches) and the last
print("| phase | mean cycles | share |\n|---|---|---|")
tot = d.sum(1).mean()
for i, st in enumerate(STAMPS):
    print("| %s | %.0f | %.1f %% |" % (NAMES[st], d[:, i].mean(), 100 * d[:, i].mean() / tot))
print("| one depth step | %.0f | |" % tot)

# per-record trace of fc_1 (F_TIMING): cycles between consecutive records, averaged over the steps
tr = out["raw"].view(torch.int32).reshape(-1)[8192:8192 + 64 * 128].cpu().numpy().astype(np.int64).reshape(64, 128)
dr = (np.diff(tr, axis=1) & 0xffffffff)[2:-1]
m = dr.mean(0)
print("fc_1 record-to-record cycles (records 1..127; a pair = 32 records: 16 main, then 16 cross; page turns every 12 stream records):")
for i in range(0, 127, 16):
    print(" ".join("%4d" % v for v in m[i:i + 16]))
