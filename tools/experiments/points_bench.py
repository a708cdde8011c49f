"""nb_decode_points timing per arithmetic: the sample points of a 512x512x64 view, 2 M of them (ray-major: 64 consecutive points are the
samples of ONE ray, ~3 cm apart — the worst order for the voxel-list march), colour + density and density only.   python tools/experiments/points_bench.py"""
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
    scene32 = net.make_scene(vols, sp)
    scene16 = net.make_scene(vols, sp, "f16f6")
    lb = net.latent_bias(sp["latent_index"])
    sel = torch.arange(0, n, 8, device=dev)  # every 8th ray, all of its 64 samples, ray-major like get_pixel_value
    wpts, _ = rend.get_sampling_points(bd["ray_o"][:, sel], bd["ray_d"][:, sel], bd["near"][:, sel], bd["far"][:, sel])
    w = wpts.reshape(-1, 3).contiguous()
    vd = (bd["ray_d"][0, sel] / bd["ray_d"][0, sel].norm(dim=-1, keepdim=True))[:, None].repeat(1, 64, 1).reshape(-1, 3).contiguous()
    ref = None
    for prec in ("f32", "f16f6"):
        pk = net.packed_weights(prec)
        scene = scene16 if prec == "f16f6" else scene32
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

    # the Network API (calculate_density_color / calculate_density): sorts the points spatially first
    for name, fn in (("raw", lambda: net.calculate_density_color(w[None], vd[None], vols, sp)), ("density", lambda: net.calculate_density(w[None], vols, sp))):
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        err = float((out[0] - ref).abs().max()) if name == "raw" else float("nan")
        print("Network API (%s, spatial sort + decode + scatter) %-8s %7.3f ms  max |raw - f32| %.2e" % (net.march_precision(), name, e0.elapsed_time(e1) / 5, err))
    # where the time goes: the sort, and the kernel on the sorted points
    def timed(fn, reps=5):
        for _ in range(2):
            r = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r
    t_sort, order = timed(lambda: net._spatial_order(w, sp))
    ws, vds = w[order].contiguous(), vd[order].contiguous()
    t_k, _ = timed(lambda: ops.decode_points(scene16, net.packed_weights("f16f6"), lb, ws, vds, precision="f16f6"))
    print("spatial sort %.3f ms; f16f6 kernel on the sorted points %.3f ms" % (t_sort, t_k))
