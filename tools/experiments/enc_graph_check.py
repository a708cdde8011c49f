"""The prefetched encoder pass eager vs as a HIP graph, back to back on an idle device: ms per pass (HIP events) and host ms per pass."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, None)
side = torch.cuda.Stream()
with torch.no_grad():
    rend.render(bd)
    sp = rend.prepare_sp_input(bd)
    for mode in ("eager", "graph"):
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fv = rend._encode_for_ticket(dict(sp))
            run = g.replay
        else:
            run = lambda: rend._encode_for_ticket(sp)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        host = (time.perf_counter() - t0) / 20 * 1e3
        torch.cuda.synchronize()
        print("%s: %.3f ms per pass on the device, %.3f ms of host time per pass" % (mode, e0.elapsed_time(e1) / 20, host))
