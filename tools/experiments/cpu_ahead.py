"""Is the host ahead of the device in the bench loop?  Per-step host time of Renderer.render (enqueue only) against the device
step time; and where the host spends it (encoder / march glue)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, None)
with torch.no_grad():
    for _ in range(3):
        rend.render(bd)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(12):
        a = time.perf_counter()
        rend.render(bd)
        host.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("host ms per render():", " ".join("%.2f" % (1e3 * h) for h in host))
    print("enqueue of 12 steps took %.1f ms, device finished after %.1f ms (%.2f ms / step)" % (1e3 * t_enq, 1e3 * t_all, 1e3 * t_all / 12))
    # split: encoder only / march only host time
    sp = rend.prepare_sp_input(bd)
    torch.cuda.synchronize()
    a = time.perf_counter(); vols = net.encode_sparse_voxels(sp); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    print("encoder: host enqueue %.2f ms, device done after %.2f ms" % (1e3 * (b - a), 1e3 * (c - a)))
