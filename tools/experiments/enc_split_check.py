"""Compare the encoder's fp16-split convolution path with the exact-fp32 kernels on the bench scene (full out_sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import neuralbody_amd.network as nw  # noqa: E402

dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, "f16f6")
sp = rend.prepare_sp_input(bd)
out = {}
for split in (True, False, True):
    nw.ENC_SPLIT = split
    with torch.no_grad():
        v = [x.clone() for x in net.encode_sparse_voxels(sp)]
    torch.cuda.synchronize()
    out.setdefault(split, []).append(v)
for li in range(4):
    a, b, a2 = out[True][0][li], out[False][0][li], out[True][1][li]
    print("level %d: shape %s max|fp32| %.3e max|split| %.3e  split-vs-fp32 rel %.3e  split run1-vs-run2 abs %.3e  nan %d inf %d" % (
        li, tuple(a.shape), float(b.abs().max()), float(a.abs().max()), float((a - b).abs().max() / b.abs().max()),
        float((a - a2).abs().max()), int(torch.isnan(a).sum()), int(torch.isinf(a).sum())))
