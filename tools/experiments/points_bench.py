"""nb_decode_points timing per arithmetic: the sample points of a 512x512x64 view, 2 M of them, colour + density
and density only.   python tools/experiments/points_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuralbody_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, "f16f6")
with torch.no_grad():
    sp = rend.prepare_sp_input(bd)
    vols = net.encode_sparse_voxels(sp)
    scene = net.make_scene(vols, sp)
    lb = net.latent_bias(sp["latent_index"])
    sel = torch.arange(0, n, 8, device=dev)  # every 8th ray, all of its 64 samples, ray-major like get_pixel_value
    wpts, _ = rend.get_sampling_points(bd["ray_o"][:, sel], bd["ray_d"][:, sel], bd["near"][:, sel], bd["far"][:, sel])
    w = wpts.reshape(-1, 3).contiguous()
    vd = (bd["ray_d"][0, sel] / bd["ray_d"][0, sel].norm(dim=-1, keepdim=True))[:, None].repeat(1, 64, 1).reshape(-1, 3).contiguous()
    ref = None
    for prec in ("f32", "bf16x3", "f16f6"):
        pk = net.packed_weights(prec)
        for density_only in (False, True):
            for _ in range(2):
                out = ops.decode_points(scene, pk, lb, w, None if density_only else vd, density_only=density_only, precision=prec)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = ops.decode_points(scene, pk, lb, w, None if density_only else vd, density_only=density_only, precision=prec)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            if prec == "f32" and not density_only:
                ref = out
            err = float((out - ref).abs().max()) if (ref is not None and not density_only) else float("nan")
            print("%-7s %-12s %8d points  %7.3f ms  %.3e points/s  max |raw - f32| %.2e" % (
                prec, "density" if density_only else "raw", w.shape[0], ms, w.shape[0] / ms * 1e3, err))
